"""Scratch probe: the text-side (M = B*L = 875 rows) GEMM shapes; PNP_GEMM_SMALL_NS selects the ring depth."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pnp-ovss_amd"))
import torch
from pnp_ovss import hip
lib = hip.load_library()
M = 875
for N, K in ((2304, 768), (768, 768), (3072, 768), (768, 3072)):
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16); B = (0.02 * torch.randn(N, K, device="cuda")).to(torch.bfloat16)
    bi = torch.randn(N, device="cuda"); ot = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    call = lambda: lib.pnp_op_gemm_ex(1, A.data_ptr(), K, B.data_ptr(), K, M, N, K, bi.data_ptr(), None, 0, None, 0, ot.data_ptr(), N, 0, None)
    for _ in range(5): assert call() == 0
    torch.cuda.synchronize(); n = 200; t0 = time.perf_counter()
    for _ in range(n): call()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    ref = A.float() @ B.float().t() + bi
    print(f"M={M} N={N} K={K}: {dt*1e6:6.1f} us  {2*M*N*K/dt/1e12:6.1f} TF  relerr {float((ot.float()-ref).abs().max()/ref.abs().max()):.1e}", flush=True)
