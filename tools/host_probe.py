import os, sys, time
ROOT="/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pnp-ovss_amd"))
import numpy as np, torch
from pnp_ovss import config as C, synth
from pnp_ovss.hip import Engine
cfg = C.blip_itm_large(336); B=35
e = Engine(cfg, max_batch=B, max_text_len=32, stash_layer=7, mode="bf16")
e.load_state_dict({k: torch.from_numpy(v).cuda() for k, v in synth.synth_state_dict(cfg, 0).items()})
_, imgs = synth.synth_images(B, 336, seed=1234, noise=4)
d_img = torch.from_numpy(imgs).cuda()
ids, mask = synth.synth_tokens(cfg, [20]*B, seed=1234); L=int(mask.sum(1).max())
d_ids, d_mask = torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda()
for _ in range(2): e.drop_loop(d_img, d_ids, d_mask, L, 9, 4)
torch.cuda.synchronize()
for rep in range(3):
    t0=time.perf_counter(); e.drop_loop(d_img, d_ids, d_mask, L, 9, 4); t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
    print(f"drop_loop: host enqueue {1e3*(t1-t0):.1f} ms, until done {1e3*(t2-t0):.1f} ms")
t0=time.perf_counter(); e.vit_forward(d_img); t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
print(f"vit_forward: host {1e3*(t1-t0):.2f} ms, done {1e3*(t2-t0):.2f} ms")
t0=time.perf_counter(); lg=e.text_forward(d_ids, d_mask, L); e.xattn_grad(B, L); t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
print(f"text fwd+bwd: host {1e3*(t1-t0):.2f} ms, done {1e3*(t2-t0):.2f} ms")
