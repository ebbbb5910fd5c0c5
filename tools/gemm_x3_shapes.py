"""Split-bf16 ("bf16x3") wide GEMM at the four ViT-L block shapes of the bench batch: back-to-back launch time, issued and
algorithmic rate; with the DEV library (make -C pnp-ovss_amd/csrc DEV=1 OBJDIR=build_dev OUT=../pnp_ovss/libpnp_hip_dev.so,
PNP_GEMM_STAMPS=1, --dev) also the in-kernel clock and the main-loop cycles per slab from the workgroup stamps."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pnp-ovss_amd"))
import numpy as np
import torch
from pnp_ovss import hip

DEV = "--dev" in sys.argv
ONLY = [a for a in sys.argv[1:] if not a.startswith("--")]
if DEV:
    hip.LIB_PATH = os.path.join(os.path.dirname(hip.LIB_PATH), "libpnp_hip_dev.so")
if "--lib" in sys.argv:                                     # a named library variant next to the product one (A/B runs)
    hip.LIB_PATH = os.path.join(os.path.dirname(hip.LIB_PATH), sys.argv[sys.argv.index("--lib") + 1])
    ONLY = [a for a in ONLY if not a.endswith(".so")]
lib = hip.load_library()


def split(t):
    hi = torch.empty(t.shape, device="cuda", dtype=torch.bfloat16)
    lo = torch.empty_like(hi)
    assert lib.pnp_op_split(t.data_ptr(), hi.data_ptr(), lo.data_ptr(), t.numel(), None) == 0
    return hi, lo


def run(tag, M, N, K, kind):
    if ONLY and tag not in ONLY:
        return
    torch.manual_seed(0)
    A = torch.randn(M, K, device="cuda")
    B = 0.02 * torch.randn(N, K, device="cuda")
    (Ah, Al), (Bh, Bl) = split(A), split(B)
    bias = torch.randn(N, device="cuda")
    p = lambda t: t.data_ptr() if t is not None else None
    out = hi = lo = None
    if kind == "resid":
        out = torch.randn(M, N, device="cuda")
        call = lambda: lib.pnp_op_gemm_x3(p(Ah), p(Al), K, p(Bh), p(Bl), K, M, N, K, p(bias), 0, p(out), N, p(out), N, None, None, 0, 0, 0, 0, None)
    elif kind == "bias":
        out = torch.empty(M, N, device="cuda")
        call = lambda: lib.pnp_op_gemm_x3(p(Ah), p(Al), K, p(Bh), p(Bl), K, M, N, K, p(bias), 0, None, 0, p(out), N, None, None, 0, 0, 0, 0, None)
    else:
        hi = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        lo = torch.empty_like(hi)
        call = lambda: lib.pnp_op_gemm_x3(p(Ah), p(Al), K, p(Bh), p(Bl), K, M, N, K, p(bias), 0, None, 0, None, 0, p(hi), p(lo), N,
                                          1 if kind == "gelu" else 0, 0, 0, None)
    for _ in range(3):
        assert call() == 0
    torch.cuda.synchronize()
    if kind == "bias":
        ref = A.double() @ B.double().t() + bias.double()
        err = float((out.double() - ref).abs().max() / ref.abs().max())
    else:
        err = -1.0
    n = 100
    t0 = time.perf_counter()
    for _ in range(n):
        call()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    fl = 2.0 * M * N * K
    print(f"{tag:6s} M={M} N={N} K={K}: {dt * 1e6:7.1f} us  algorithmic {fl / dt / 1e12:6.0f} TF  issued {3 * fl / dt / 1e12:6.0f} TF "
          f"({3 * fl / dt / 2.5e15:.3f} of 2.5 PF)  relerr {err:.1e}", flush=True)
    if DEV and os.environ.get("PNP_GEMM_STAMPS"):
        nb = min(((M + 255) // 256) * ((N + 255) // 256), 256)
        st = np.zeros((nb, 8), dtype=np.uint64)
        assert lib.pnp_dbg_gemm_stamps(st.ctypes.data, nb) == 0
        cyc = st[:, :4].astype(np.int64)
        wall = st[:, 4:].astype(np.int64)
        us = (wall - wall[:, 0].min()) / 100.0
        # slots: 0 start, 2 first tile's main loop end, 1 first tile's epilogue issued, 3 workgroup done
        main_us = us[:, 2] - us[:, 0]
        main_clk = cyc[:, 2] - cyc[:, 0]
        ghz = (cyc[:, 3] - cyc[:, 0]) / ((us[:, 3] - us[:, 0]) * 1e3)
        nk = K // 32
        print(f"       span {us[:, 3].max():.1f} us, clock {np.median(ghz):.2f} GHz; first tile: prologue + main loop {np.median(main_us):.1f} us = "
              f"{np.median(main_clk) / nk:.0f} clk per 32-deep slab (MFMA floor 3072), epilogue {np.median(us[:, 1] - us[:, 2]):.1f} us; "
              f"tiles per workgroup {((M + 255) // 256) * ((N + 255) // 256) / nb:.2f}", flush=True)


M = 15470
run("qkv", M, 3072, 1024, "split")
run("fc1", M, 4096, 1024, "gelu")
run("fc2", M, 1024, 4096, "resid")
run("proj", M, 1024, 1024, "resid")
run("crossk", M, 9216, 1024, "bias")
