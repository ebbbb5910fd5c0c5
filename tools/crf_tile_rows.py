"""GPU box: how many DISTINCT lattice value rows does a TW x TH pixel tile of the DenseCRF update touch (6 bilateral + 3 Gaussian
vertices per pixel)?  The update gathers 9 rows per pixel; what a tile could stage once is the distinct count.
  python tools/crf_tile_rows.py [img=336] [B=4] [photo|blocks]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pnp-ovss_amd"))
import numpy as np
import torch
from pnp_ovss import config as C, synth
from pnp_ovss.hip import Engine

IMG = int(sys.argv[1]) if len(sys.argv) > 1 else 336
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
KIND = sys.argv[3] if len(sys.argv) > 3 else "blocks"
K = 21
cfg = C.blip_itm_small(IMG)
if KIND == "photo":
    rgb, _ = synth.synth_photo_images(B, IMG, seed=1234)
else:
    rgb, _ = synth.synth_images(B, IMG, seed=1234, noise=4)
rgb = np.ascontiguousarray(np.asarray(rgb, dtype=np.uint8).reshape(B, IMG, IMG, 3))
d_rgb = torch.from_numpy(rgb.reshape(-1)).cuda()
plans = [[([i], 1) for i in range(K - 1)]] * B
luts = [list(range(K))] * B
e = Engine(cfg, max_batch=B, max_text_len=32, stash_layer=7, bf16=True)
e.post_reserve(B, B * IMG * IMG, IMG * IMG, K, 0)
e.post_prepare([(IMG, IMG)] * B, plans, luts, [True] * B, rgb=d_rgb, gt=None, want_crf=True)
torch.cuda.synchronize()
og = e.buffer("crf_offset_gauss", torch.int32)[: B * IMG * IMG * 3].cpu().numpy().reshape(B, IMG, IMG, 3)
ob = e.buffer("crf_offset_bilateral", torch.int32)[: B * IMG * IMG * 6].cpu().numpy().reshape(B, IMG, IMG, 6)
for tw, th in ((4, 4), (8, 4), (8, 8), (16, 8), (16, 16)):
    dg, db = [], []
    for b in range(min(B, 2)):
        for y0 in range(0, IMG - th + 1, th * 3):
            for x0 in range(0, IMG - tw + 1, tw * 3):
                dg.append(len(np.unique(og[b, y0:y0 + th, x0:x0 + tw])))
                db.append(len(np.unique(ob[b, y0:y0 + th, x0:x0 + tw])))
    dg, db = np.array(dg), np.array(db)
    tp = tw * th
    print(f"{KIND} {IMG}^2 tile {tw:2d}x{th:2d} ({tp:3d} px): gathers {9 * tp:5d} rows; distinct Gaussian {dg.mean():6.1f} (max {dg.max()}), "
          f"bilateral {db.mean():6.1f} (p90 {np.percentile(db, 90):.0f}, max {db.max()}) -> {9 * tp / (dg.mean() + db.mean()):.1f}x fewer rows", flush=True)
