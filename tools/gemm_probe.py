"""Scratch probe (GPU box): stand-alone GEMM throughput through the C ABI."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pnp-ovss_amd"))
import torch
from pnp_ovss import hip
lib = hip.load_library()
for (M, N, K) in [(4096, 4096, 4096), (8192, 8192, 8192), (15470, 4096, 1024), (15470, 1024, 4096), (15470, 2048, 1024), (15470, 1024, 1024)]:
    A = (torch.randn(M, K, device="cuda") ).to(torch.bfloat16)
    B = (torch.randn(N, K, device="cuda") ).to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda")
    for it in range(3):
        lib.pnp_op_gemm(1, A.data_ptr(), K, B.data_ptr(), K, M, N, K, None, None, 0, out.data_ptr(), N, 0, None)
    torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    for it in range(n):
        lib.pnp_op_gemm(1, A.data_ptr(), K, B.data_ptr(), K, M, N, K, None, None, 0, out.data_ptr(), N, 0, None)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    ref = (A[:64].float() @ B[:64].float().T)
    err = (out[:64, :64] - ref).abs().max().item()
    t1 = time.perf_counter()
    for it in range(n):
        C = A @ B.T
    torch.cuda.synchronize()
    dt2 = (time.perf_counter() - t1) / n
    print(f"M={M} N={N} K={K}: ours {2*M*N*K/dt/1e12:.0f} TF ({dt*1e6:.0f} us), torch.matmul {2*M*N*K/dt2/1e12:.0f} TF, err {err:.3g}", flush=True)
