"""Scratch probe (GPU box): CRF/post-process time vs crf_chunk and synthetic-image noise level."""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pnp-ovss_amd"))
import numpy as np, torch
from pnp_ovss import config as C, synth
from pnp_ovss.hip import Engine

B, IMG, K = 35, 336, 21
cfg = C.blip_itm_small(IMG)      # post-process only needs the grid (P = 21)
rng = np.random.default_rng(0)
maps = (rng.random((B, 24, 21, 21), dtype=np.float32) ** 4)
d_maps = torch.from_numpy(maps).cuda()
plans = [[([i], 1) for i in range(20)]] * B
luts = [list(range(21))] * B
for noise in (4,):
    g = np.random.default_rng([1234, 7])
    nb = IMG // 8
    coarse = g.integers(0, 256, size=(B, nb, nb, 3))
    fine = g.integers(-noise, noise + 1, size=(B, IMG, IMG, 3))
    rgb = np.clip(np.repeat(np.repeat(coarse, 8, 1), 8, 2) + fine, 0, 255).astype(np.uint8)
    d_rgb = torch.from_numpy(rgb.reshape(-1)).cuda()
    for chunk in (35, 24, 18, 16, 12, 8):
        e = Engine(cfg, max_batch=B, max_text_len=32, stash_layer=7, bf16=True)
        e.post_reserve(B, B * IMG * IMG, IMG * IMG, K, chunk)
        e.post_prepare([(IMG, IMG)] * B, plans, luts, [True] * B, rgb=d_rgb, gt=None, want_crf=True)
        idb = e.buffer("crf_idbase_bilateral", torch.int32)[: B + 1].cpu().numpy()
        idg = e.buffer("crf_idbase_gauss", torch.int32)[: B + 1].cpu().numpy()
        e.merge_tokens(d_maps); e.threshold_upsample(0.15, False); e.blur_minmax()
        torch.cuda.synchronize()
        for rep in range(2):
            t0 = time.perf_counter(); e.densecrf(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        t0 = time.perf_counter()
        e.post_prepare([(IMG, IMG)] * B, plans, luts, [True] * B, rgb=d_rgb, gt=None, want_crf=True)
        torch.cuda.synchronize(); tp = time.perf_counter() - t0
        print(json.dumps(dict(noise=noise, chunk=chunk, crf_ms=dt * 1e3, prepare_ms=tp * 1e3, M_bilateral_per_img=float((idb[-1]) / B), M_gauss_per_img=float(idg[-1] / B))), flush=True)
        e.close(); del e
