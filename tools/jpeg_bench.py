"""Throughput of the device JPEG decoder against Pillow on the same synthetic files (GPU box only).

    python tools/jpeg_bench.py [--images 35] [--size 375x500] [--quality 90] [--reps 5] [--out gpurun_out/jpeg_bench.json]

Images are smooth colour fields plus texture noise (compressed size close to a VOC / COCO photograph of that size),
written with Pillow at 4:2:0.  Prints one JSON line: host marker walk (pack_batch), device time per batch by kernel
(HIP events), Pillow single-thread (what the reference's DataLoader does) and 8-thread decode times.
"""
import argparse
import io
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pnp-ovss_amd"))


def synth_photo(rng, H, W):
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    img = np.zeros((H, W, 3), np.float32)
    for c in range(3):
        for _ in range(6):
            fx, fy, ph = rng.uniform(0.002, 0.05), rng.uniform(0.002, 0.05), rng.uniform(0, 6.28)
            img[..., c] += rng.uniform(10, 40) * np.sin(fx * xx + fy * yy + ph)
    img += 128 + rng.normal(0, 12, (H, W, 1)).astype(np.float32) + rng.normal(0, 4, (H, W, 3)).astype(np.float32)
    return np.clip(img, 0, 255).astype(np.uint8)


def main():
    from PIL import Image
    import torch
    from pnp_ovss import hip, jpeg as J
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=35)
    ap.add_argument("--size", default="375x500")
    ap.add_argument("--quality", type=int, default=90)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--restart", type=int, default=0, help="restart interval in MCU rows (0 = none, what the datasets ship)")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    H, W = map(int, a.size.split("x"))
    rng = np.random.default_rng(0)
    files = []
    for _ in range(a.images):
        b = io.BytesIO()
        kw = dict(restart_marker_rows=a.restart) if a.restart else {}
        Image.fromarray(synth_photo(rng, H, W)).save(b, "JPEG", quality=a.quality, **kw)
        files.append(b.getvalue())
    nbytes = sum(len(f) for f in files)

    def pil_one(f):
        return np.asarray(Image.open(io.BytesIO(f)).convert("RGB"))
    ref = [pil_one(f) for f in files]
    t0 = time.perf_counter()
    for _ in range(a.reps):
        for f in files:
            pil_one(f)
    t_pil1 = (time.perf_counter() - t0) / a.reps
    with ThreadPoolExecutor(8) as ex:
        list(ex.map(pil_one, files))
        t0 = time.perf_counter()
        for _ in range(a.reps):
            list(ex.map(pil_one, files))
        t_pil8 = (time.perf_counter() - t0) / a.reps

    out = hip.jpeg_decode_batch(files)
    exact = all(np.array_equal(o.cpu().numpy(), r) for o, r in zip(out, ref))
    t0 = time.perf_counter()
    for _ in range(a.reps):
        J.pack_batch(files)
    t_pack = (time.perf_counter() - t0) / a.reps
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        hip.jpeg_decode_batch(files)
    torch.cuda.synchronize()
    t_total = (time.perf_counter() - t0) / a.reps
    rec = dict(images=a.images, size=[H, W], quality=a.quality, restart_rows=a.restart, jpeg_bytes=nbytes, bit_exact_vs_pillow=bool(exact),
               pillow_1thread_ms=t_pil1 * 1e3, pillow_8thread_ms=t_pil8 * 1e3, host_pack_ms=t_pack * 1e3,
               device_total_ms=t_total * 1e3, device_only_ms=(t_total - t_pack) * 1e3,
               images_per_s=dict(pillow_1thread=a.images / t_pil1, pillow_8thread=a.images / t_pil8, device=a.images / t_total))
    line = json.dumps(rec)
    print(line)
    if a.out:
        with open(a.out, "w") as f:
            f.write(line + "\n")


if __name__ == "__main__":
    main()
