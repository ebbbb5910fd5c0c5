"""The text-side (M = B*L = 875 rows) GEMM shapes: the split-bf16 kernel (gemm_nt_small_x3_kernel, fp32 activations, weight as a
(hi, lo) bf16 pair) next to the bf16 and the exact-fp32 kernels.  `--dev`: DEV library (PNP_SMALL_NW=4|8 selects the wave count)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pnp-ovss_amd"))
import torch
from pnp_ovss import hip

if "--dev" in sys.argv:
    hip.LIB_PATH = os.path.join(os.path.dirname(hip.LIB_PATH), "libpnp_hip_dev.so")
lib = hip.load_library()
M = int([a for a in sys.argv[1:] if a.isdigit()][0]) if any(a.isdigit() for a in sys.argv[1:]) else 875


def timeit(call, n=200):
    for _ in range(5):
        assert call() == 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        call()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


tot = {"x3": 0.0, "bf16": 0.0, "f32": 0.0}
# (N, K, launches per text layer forward + backward at the bench's stash layer)
for N, K, w in ((2304, 768, 1), (768, 768, 6), (3072, 768, 2), (768, 3072, 2), (768, 2304, 1)):
    A = torch.randn(M, K, device="cuda")
    B = 0.02 * torch.randn(N, K, device="cuda")
    bi = torch.randn(N, device="cuda")
    Bh = torch.empty(N, K, device="cuda", dtype=torch.bfloat16)
    Bl = torch.empty_like(Bh)
    assert lib.pnp_op_split(B.data_ptr(), Bh.data_ptr(), Bl.data_ptr(), B.numel(), None) == 0
    out = torch.empty(M, N, device="cuda")
    tx = timeit(lambda: lib.pnp_op_gemm_x3a(A.data_ptr(), K, Bh.data_ptr(), Bl.data_ptr(), K, M, N, K, bi.data_ptr(), None, 0, out.data_ptr(), N, 0, None, 0, None))
    ref = A.double() @ B.double().t() + bi.double()
    err = float((out.double() - ref).abs().max() / ref.abs().max())
    Ab, Bb = A.to(torch.bfloat16), B.to(torch.bfloat16)
    ot = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    tb = timeit(lambda: lib.pnp_op_gemm_ex(1, Ab.data_ptr(), K, Bb.data_ptr(), K, M, N, K, bi.data_ptr(), None, 0, None, 0, ot.data_ptr(), N, 0, None))
    tf = timeit(lambda: lib.pnp_op_gemm_ex(0, A.data_ptr(), K, B.data_ptr(), K, M, N, K, bi.data_ptr(), None, 0, out.data_ptr(), N, None, 0, 0, None))
    tot["x3"] += w * tx; tot["bf16"] += w * tb; tot["f32"] += w * tf
    print(f"M={M} N={N} K={K}: split-bf16 {tx * 1e6:6.1f} us (relerr {err:.1e})   bf16 {tb * 1e6:6.1f} us   f32 {tf * 1e6:6.1f} us", flush=True)
print("weighted per layer (us):", {k: round(v * 1e6, 1) for k, v in tot.items()})
