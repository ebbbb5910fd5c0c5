"""In-kernel clock stamps of the stream-K tail (DEV library: make -C pnp-ovss_amd/csrc DEV=1 OBJDIR=build_dev OUT=../pnp_ovss/libpnp_hip_dev.so;
PNP_GEMM_STAMPS=1).  Per workgroup: start, end of the whole-tile part, producing part (main loop end -> partial tile stored and
flag raised), owning part (main loop end -> partial tiles of the workgroups before it added), workgroup done.
usage: PNP_GEMM_STAMPS=1 python tools/gemm_x3_streamk_stamps.py [M N K kind]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pnp-ovss_amd"))
import numpy as np
import torch
from pnp_ovss import hip

hip.LIB_PATH = os.path.join(os.path.dirname(hip.LIB_PATH), "libpnp_hip_dev.so")
lib = hip.load_library()
args = [a for a in sys.argv[1:]]
M, N, K = (int(args[0]), int(args[1]), int(args[2])) if len(args) >= 3 else (35 * 442, 3072, 1024)
kind = args[3] if len(args) > 3 else "bias"


def split(t):
    hi = torch.empty(t.shape, device="cuda", dtype=torch.bfloat16)
    lo = torch.empty_like(hi)
    assert lib.pnp_op_split(t.data_ptr(), hi.data_ptr(), lo.data_ptr(), t.numel(), None) == 0
    return hi, lo


torch.manual_seed(0)
A = torch.randn(M, K, device="cuda")
B = 0.02 * torch.randn(N, K, device="cuda")
(Ah, Al), (Bh, Bl) = split(A), split(B)
bias = torch.randn(N, device="cuda")
out = torch.empty(M, N, device="cuda")
p = lambda t: t.data_ptr() if t is not None else None
call = lambda: lib.pnp_op_gemm_x3(p(Ah), p(Al), K, p(Bh), p(Bl), K, M, N, K, p(bias), 0, None, 0, p(out), N, None, None, 0, 0, 0, 0, None)
for mode in (0, 2):
    hip.set_tuning("streamk", mode)
    for _ in range(3):
        assert call() == 0
    torch.cuda.synchronize()
    nb = 512
    st = np.zeros((nb, 8), dtype=np.uint64)
    assert lib.pnp_dbg_gemm_stamps(st.ctypes.data, nb) == 0
    wall = st[:, 4:].astype(np.int64)
    nwg = int((wall[:256, 0] > 0).sum()) if mode == 0 else 256        # whole-tile launches of fewer tiles than CUs use fewer workgroups
    t0 = wall[:nwg, 0].min()
    us = (wall - t0) / 100.0
    g, k = us[:nwg], us[256:]
    print(f"mode {mode}: M={M} N={N} K={K}: span {g[:, 3].max():.1f} us; start spread {g[:, 0].max():.1f}; first tile main loop end (median) {np.median(g[:, 2]):.1f}, "
          f"first epilogue done {np.median(g[:, 1]):.1f}; workgroup done: min {g[:, 3].min():.1f} median {np.median(g[:, 3]):.1f} max {g[:, 3].max():.1f}")
    if mode == 2:
        prod = k[:, 1] > 0
        own = k[:, 3] > 0
        print(f"   producing parts ({prod.sum()} workgroups): main loop end at {np.median(k[prod, 0]):.1f} us (median), store + drain + flag {np.median(k[prod, 1] - k[prod, 0]):.1f} us "
              f"(min {np.min(k[prod, 1] - k[prod, 0]):.1f}, max {np.max(k[prod, 1] - k[prod, 0]):.1f})")
        print(f"   owning parts ({own.sum()} workgroups): main loop end at {np.median(k[own, 2]):.1f} us (median), wait + add of the parts before {np.median(k[own, 3] - k[own, 2]):.1f} us "
              f"(min {np.min(k[own, 3] - k[own, 2]):.1f}, max {np.max(k[own, 3] - k[own, 2]):.1f}); epilogue + exit {np.median(g[own, 3] - k[own, 3]):.1f} us")
        both = prod & own
        if both.any():
            print(f"   workgroups with both: flag raised -> own main loop end {np.median(k[both, 2] - k[both, 1]):.1f} us")
        np.save(os.path.join(ROOT, "gpurun_out", f"sk_stamps_{M}_{N}_{K}.npy"), us)
hip.set_tuning("streamk", 1)
