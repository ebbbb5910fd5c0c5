"""GPU box: the bench's paired post-processing (1-drop | N-drop as two channel groups of one DenseCRF run) on the
bench-shaped batch, without the model: `reps` calls of post_prepare + postprocess_pair, wall time per call and the
live mean-field time.  Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split (tools/crf_pair_probe.sh).

  python3 tools/crf_pair_probe.py [reps=3] [noise=4 | photo] [K=21] [chunk=0]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pnp-ovss_amd"))
import numpy as np
import torch
from pnp_ovss import config as C, synth, hip
from pnp_ovss.hip import Engine

if os.environ.get("PNP_DEV_LIB"):                          # DEV build: the PNP_CRF_* knobs of csrc/crf.hip
    _name = os.environ["PNP_DEV_LIB"] if os.environ["PNP_DEV_LIB"].endswith(".so") else "libpnp_hip_dev.so"
    hip.LIB_PATH = os.path.join(os.path.dirname(hip.LIB_PATH), _name)

B, IMG = 35, 336
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 3
NOISE = (sys.argv[2] if sys.argv[2] == "photo" else int(sys.argv[2])) if len(sys.argv) > 2 else 4
K = int(sys.argv[3]) if len(sys.argv) > 3 else 21        # channels: K - 1 classes + background
CHUNK = int(sys.argv[4]) if len(sys.argv) > 4 else 0     # images per DenseCRF chunk (0 = the whole batch)
cfg = C.blip_itm_small(IMG)
rng = np.random.default_rng(0)
g0 = torch.from_numpy(rng.random((B, K + 3, 21, 21), dtype=np.float32) ** 4).cuda()
agg = torch.from_numpy(rng.random((B, K + 3, 21, 21), dtype=np.float32) ** 4).cuda()
plans = [[([i], 1) for i in range(K - 1)]] * B
luts = [list(range(K))] * B
rgb, _ = synth.synth_photo_images(B, IMG, seed=1234) if NOISE == "photo" else synth.synth_images(B, IMG, seed=1234, noise=NOISE)
d_rgb = torch.from_numpy(rgb.reshape(-1)).cuda()
e = Engine(cfg, max_batch=B, max_text_len=max(32, K + 8), stash_layer=1, bf16=True)
e.post_reserve(B, B * IMG * IMG, IMG * IMG, K, CHUNK)
sizes = [(IMG, IMG)] * B
out = {}
for rep in range(REPS + 1):
    if rep == 1:
        torch.cuda.synchronize()
        e.profile_enable(1)
        t0 = time.perf_counter()
    e.post_prepare(sizes, plans, luts, [True] * B, rgb=d_rgb, gt=None, want_crf=True)
    l1, ln = e.postprocess_pair(g0, agg, 0.15, 21, None, None)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / REPS
crf = e.profile_read_stage(1)
idb = e.buffer("crf_idbase_bilateral", torch.int32)[: B + 1].cpu().numpy()
idg = e.buffer("crf_idbase_gauss", torch.int32)[: B + 1].cpu().numpy()
print(json.dumps(dict(post_ms=dt * 1e3, crf=crf, M_bilateral=int(idb[-1]), M_gauss=int(idg[-1]),
                      ppp=float(idb[-1]) / (B * IMG * IMG), noise=NOISE, K=K,
                      label_sum=[int(l1.long().sum()), int(ln.long().sum())])))
