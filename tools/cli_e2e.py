"""End-to-end rate of the drop-in CLI on a generated VOC-layout tree (GPU box): JPEG files + PNG ground truth + GPT class
table on disk -> `PnP_OVSS_0514_updated_segmentation.py --data_type voc` -> histograms on disk, i.e. everything a user of the
reference's Run_seg.sh runs, next to `bench.py`'s device-resident number.

    python tools/cli_e2e.py [--images 350] [--batch 35] [--dtype bf16] [--profile] [--device_jpeg 1] [--pipelines 1]

--profile runs the loop under cProfile and prints the host functions by cumulative time.
"""
import argparse
import cProfile
import io
import json
import os
import pstats
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "pnp-ovss_amd"))
sys.path.insert(0, str(ROOT / "tools"))

VOC = ["aeroplane", "bicycle", "bird", "boat", "bottle", "bus", "car", "cat", "chair", "cow", "diningtable", "dog", "horse",
       "motorbike", "person", "pottedplant", "sheep", "sofa", "train", "tvmonitor"]


def build_tree(home, n, seed=0):
    from PIL import Image
    from jpeg_bench import synth_photo
    root = home / "VOCdevkit" / "VOC2012"
    for d in ("JPEGImages", "SegmentationClass", "ImageSets/Segmentation"):
        (root / d).mkdir(parents=True)
    (home / "GPT4o_classification").mkdir()
    rng = np.random.default_rng(seed)
    ids, table = [], {}
    for i in range(n):
        h, w = ((375, 500), (500, 375), (333, 500), (375, 500))[i % 4]
        name = f"2008_{i:06d}"
        Image.fromarray(synth_photo(rng, h, w)).save(root / "JPEGImages" / f"{name}.jpg", quality=90)
        gt = np.zeros((h, w), np.uint8)
        cls = rng.choice(20, size=3, replace=False)
        for k, c in enumerate(cls[:2]):
            gt[h // 4 * k:h // 4 * (k + 2), w // 3 * k:w // 3 * (k + 2)] = c + 1
        Image.fromarray(gt).save(root / "SegmentationClass" / f"{name}.png")
        ids.append(name)
        table[name] = "[" + ", ".join(f"{int(c) + 1}: '{VOC[int(c)]}'" for c in cls) + "], [95%, 80%, 60%]"
    (root / "ImageSets/Segmentation/val.txt").write_text("\n".join(ids) + "\n")
    (home / "GPT4o_classification" / "voc_classification_noboundary.json").write_text(json.dumps(table))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=350)
    ap.add_argument("--batch", type=int, default=35)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--device_jpeg", type=int, default=1)
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--pipelines", type=int, default=1)
    a = ap.parse_args()
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        home = Path(tmp)
        t0 = time.perf_counter()
        build_tree(home, a.images)
        print(f"tree of {a.images} images built in {time.perf_counter() - t0:.1f} s", flush=True)
        import importlib.util
        spec = importlib.util.spec_from_file_location("pnp_cli", ROOT / "pnp-ovss_amd" / "PnP_OVSS_0514_updated_segmentation.py")
        cli = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(cli)
        argv = ["--batch_size", str(a.batch), "--home_dir", str(home), "--save_path", str(home / "out"), "--img_size", "336",
                "--del_patch_num", "sort_thresh005", "--max_att_block_num", "8", "--drop_iter", "4", "--prune_att_head", "9",
                "--sort_threshold", "0.05", "--threshold", "0.15", "--postprocess", "blur+crf", "--data_type", "voc",
                "--world_size", "1", "--dtype", a.dtype, "--device_jpeg", str(a.device_jpeg), "--pipelines", str(a.pipelines)]
        args = cli.get_args_parser().parse_args(argv)
        buf = io.StringIO()
        real = sys.stdout
        sys.stdout = buf                                # the CLI prints one line per batch
        try:
            if a.profile:
                pr = cProfile.Profile()
                pr.enable()
            cli.main(0, 1, args)
            if a.profile:
                pr.disable()
        finally:
            sys.stdout = real
        lines = [ln for ln in buf.getvalue().splitlines() if ln.startswith("{")]
        print(lines[-1] if lines else buf.getvalue()[-2000:])
        if a.profile:
            st = pstats.Stats(pr)
            st.sort_stats("cumulative").print_stats(45)


if __name__ == "__main__":
    main()
