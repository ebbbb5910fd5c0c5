"""GPU box: how local are the two-axis blur's neighbour ids in the spatial numbering of the lattice?  For the bench-shaped batch
(or 12-level noise: argv[1] = fine noise amplitude) prints, per lattice, the share of present neighbours and the share of
neighbour references that fall inside a window of T points + H halo points on either side around the referencing point's tile."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pnp-ovss_amd"))
import numpy as np, torch
from pnp_ovss import config as C
from pnp_ovss.hip import Engine

B, IMG, K = 35, 336, 21
NOISE = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = C.blip_itm_small(IMG)
rng = np.random.default_rng(0)
d_maps = torch.from_numpy(rng.random((B, K + 3, 21, 21), dtype=np.float32) ** 4).cuda()
plans = [[([i], 1) for i in range(K - 1)]] * B
luts = [list(range(K))] * B
g = np.random.default_rng([1234, 7])
coarse = g.integers(0, 256, size=(B, IMG // 8, IMG // 8, 3))
fine = g.integers(-NOISE, NOISE + 1, size=(B, IMG, IMG, 3))
rgb = np.clip(np.repeat(np.repeat(coarse, 8, 1), 8, 2) + fine, 0, 255).astype(np.uint8)
d_rgb = torch.from_numpy(rgb.reshape(-1)).cuda()
e = Engine(cfg, max_batch=B, max_text_len=32, stash_layer=7, bf16=True)
e.post_reserve(B, B * IMG * IMG, IMG * IMG, K, 0)
e.post_prepare([(IMG, IMG)] * B, plans, luts, [True] * B, rgb=d_rgb, gt=None, want_crf=True)
e.merge_tokens(d_maps); e.threshold_upsample(0.15, False); e.blur_minmax(); e.densecrf()
torch.cuda.synchronize()
for name, pairs in (("gauss", 1), ("bilateral", 3)):
    idb = e.buffer(f"crf_idbase_{name}", torch.int32)[: B + 1].cpu().numpy()
    nb = e.buffer(f"crf_nbr8_{name}", torch.int32).cpu().numpy().reshape(pairs, -1, 8)
    print(name, "points", int(idb[-1]), "cap", nb.shape[1], flush=True)
    for b in (0, 17):
        lo, hi = int(idb[b]), int(idb[b + 1])
        for pr in range(pairs):
            t = nb[pr, lo:hi]
            if (t < 0).all():
                continue
            ids = np.arange(hi - lo)[:, None]
            present = t >= 0
            line = f"  image {b} pair {pr}: M {hi - lo} present {present.mean():.3f}"
            d = np.abs(t - ids)[present]
            line += "  |delta| pct 50/90/99: %d %d %d" % tuple(np.percentile(d, [50, 90, 99]))
            for T, H in ((512, 128), (768, 128), (1024, 256), (2048, 512), (4096, 1024)):
                tile_lo = ids // T * T
                inside = (t >= tile_lo - H) & (t < tile_lo + T + H) & present
                line += f"  T{T}+{H}: {inside.sum() / present.sum():.3f}"
            print(line, flush=True)
