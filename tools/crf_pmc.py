"""Scratch probe (GPU box): one densecrf() call on the bench-shaped batch, for rocprofv3 --pmc runs
(restrict with --kernel-include-regex crf_)."""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pnp-ovss_amd"))
import numpy as np, torch
from pnp_ovss import config as C
from pnp_ovss.hip import Engine

B, IMG = 35, 336
K = int(sys.argv[3]) if len(sys.argv) > 3 else 21        # channels: K - 1 classes + background
ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cfg = C.blip_itm_small(IMG)
rng = np.random.default_rng(0)
d_maps = torch.from_numpy(rng.random((B, K + 3, 21, 21), dtype=np.float32) ** 4).cuda()
plans = [[([i], 1) for i in range(K - 1)]] * B
luts = [list(range(K))] * B
g = np.random.default_rng([1234, 7])
coarse = g.integers(0, 256, size=(B, IMG // 8, IMG // 8, 3))
fine = g.integers(-4, 5, size=(B, IMG, IMG, 3))
rgb = np.clip(np.repeat(np.repeat(coarse, 8, 1), 8, 2) + fine, 0, 255).astype(np.uint8)
d_rgb = torch.from_numpy(rgb.reshape(-1)).cuda()
e = Engine(cfg, max_batch=B, max_text_len=max(32, K + 8), stash_layer=7, bf16=True)
e.post_reserve(B, B * IMG * IMG, IMG * IMG, K, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
e.post_prepare([(IMG, IMG)] * B, plans, luts, [True] * B, rgb=d_rgb, gt=None, want_crf=True)
e.merge_tokens(d_maps); e.threshold_upsample(0.15, False); e.blur_minmax()
torch.cuda.synchronize()
for rep in range(ITERS):
    t0 = time.perf_counter(); e.densecrf(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
idb = e.buffer("crf_idbase_bilateral", torch.int32)[: B + 1].cpu().numpy()
idg = e.buffer("crf_idbase_gauss", torch.int32)[: B + 1].cpu().numpy()
print(json.dumps(dict(crf_ms=dt * 1e3, M_bilateral=int(idb[-1]), M_gauss=int(idg[-1]))))
