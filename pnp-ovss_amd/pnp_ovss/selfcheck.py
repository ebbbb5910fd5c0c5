"""Self-check of a compute mode against the reference's own outputs: the committed pipeline fixtures
(tests/golden/pipeline_{voc,psc,voc_large}.npz -- inputs by seed, and the label maps the REFERENCE's
save_img_union_attention produced for them, PnP.py:290-521; `pipeline_voc_large.npz` is the benchmarked geometry:
BLIP-ITM-large 336^2 with the full 20-class prompt) run through the HIP engine in the given mode; returns how many label pixels differ.

Used by bench.py to print, next to the headline number, the label-flip fraction of the benchmarked mode against the
reference fixtures (tests/test_hip_parity.py::test_end_to_end_labels_vs_reference_run asserts the same comparison with the
near-tie rule).  Product code only: no oracle import; the fixtures are data."""
import json
import os

import numpy as np
import torch

from . import config as C
from . import host, synth
from .hip import Engine
from .tokenizer import SynthTokenizer


def fixture_label_flips(mode, golden_dir, fixtures=("pipeline_voc.npz", "pipeline_psc.npz", "pipeline_voc_large.npz"), device=0):
    """-> {"pixels": total compared, "differing": count, "frac": ..., "per_fixture": {...}} over both branches (1-drop,
    N-drop) x both un-CRF'd post-process forms the fixtures hold (blur, none) of every fixture image."""
    dev = torch.device("cuda", device)
    out = {"pixels": 0, "differing": 0, "per_fixture": {}}
    for fname in fixtures:
        g = np.load(os.path.join(golden_dir, fname), allow_pickle=False)
        cfg = C.ModelCfg(**json.loads(str(g["cfg"])))
        data_type = str(g["data_type"])
        cats = {int(k): v for k, v in json.loads(str(g["cats"])).items()}
        nms = list(cats.values())
        sizes = [tuple(int(v) for v in s) for s in g["sizes"]]
        B = len(sizes)
        _, imgs = synth.synth_images(B, cfg.img_size, seed=int(g["image_seed"]))
        rng = np.random.default_rng(int(g["org_seed"]))
        org = [rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8) for h, w in sizes]
        gpt = json.loads(str(g["gpt"]))
        tok = SynthTokenizer(cfg.vocab)
        best, caps = [], []
        for k in [str(s) for s in g["img_ids"]]:
            b, _, cap = host.parse_gpt_classes(gpt[k], nms)
            best.append(b)
            caps.append(cap)
        enc = tok(caps, padding="max_length", max_length=500)
        ids, mask = enc.input_ids.numpy(), enc.attention_mask.numpy()
        L = int(mask.sum(1).max())
        e = Engine(cfg, max_batch=4, max_text_len=32, stash_layer=7, mode=mode, device=device)
        try:
            e.load_state_dict(synth.synth_state_dict(cfg, int(g["weight_seed"])))
            e.post_reserve(max(B, 4), max(sum(h * w for h, w in sizes), 4 * 128 * 128), max(max(h * w for h, w in sizes), 128 * 128),
                           max(max(len(b) for b in best) + 1, 8), 0)
            g0, agg, _, _ = e.drop_loop(torch.from_numpy(imgs).to(dev), torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev),
                                        L, 9, 4)
            plans, luts, bgs = [], [], []
            for i in range(B):
                bg = host.has_background(data_type, len(best[i]))
                plans.append(host.merge_plan(host.caption_pieces(tok, ids[i]), len(best[i])))
                luts.append(host.remap_lut(best[i], bg, len(best[i]) + int(bg)))
                bgs.append(bg)
            d_rgb = torch.from_numpy(np.concatenate([r.reshape(-1) for r in org])).to(dev)
            e.post_prepare(sizes, plans, luts, bgs, rgb=d_rgb, gt=None, want_crf=False)
            n = d = 0
            for name, src, scale01 in (("1drop", g0, True), ("ndrop", agg, False)):
                for pp in ("blur", None):
                    labels = e.split_labels(e.postprocess(src, 0.15, scale01, pp))
                    torch.cuda.synchronize()
                    for i in range(B):
                        ref = g[f"labels_{name}_{pp or 'none'}_{i}"]
                        d += int((labels[i].cpu().numpy() != ref).sum())
                        n += ref.size
        finally:
            e.close()
        out["per_fixture"][fname] = {"pixels": n, "differing": d}
        out["pixels"] += n
        out["differing"] += d
    out["frac"] = out["differing"] / max(out["pixels"], 1)
    return out
