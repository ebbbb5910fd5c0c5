"""Scratch probe: epilogue cost of the GEMM on the fc1 / fc2 / qk shapes."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pnp-ovss_amd"))
import torch
from pnp_ovss import hip
lib = hip.load_library()
def run(M, N, K, bias, resid, f32out, tout, mode, tag):
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16); B = (0.02 * torch.randn(N, K, device="cuda")).to(torch.bfloat16)
    bi = torch.randn(N, device="cuda") if bias else None
    rs = torch.randn(M, N, device="cuda") if resid else None
    of = torch.empty(M, N, device="cuda") if f32out else None
    ot = torch.empty(M, N, device="cuda", dtype=torch.bfloat16) if tout else None
    p = lambda t: t.data_ptr() if t is not None else None
    def call():
        return lib.pnp_op_gemm_ex(1, A.data_ptr(), K, B.data_ptr(), K, M, N, K, p(bi), p(rs), N, p(of), N, p(ot), N, mode, None)
    for _ in range(3): assert call() == 0
    torch.cuda.synchronize(); n = 20; t0 = time.perf_counter()
    for _ in range(n): call()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"{tag:34s} M={M} N={N} K={K}: {2*M*N*K/dt/1e12:6.0f} TF ({dt*1e6:.0f} us)", flush=True)
M = 15470
run(M, 4096, 1024, False, False, True, False, 0, "fc1 shape, f32 out, plain")
run(M, 4096, 1024, False, False, False, True, 0, "fc1 shape, bf16 out, plain")
run(M, 4096, 1024, True, False, False, True, 0, "fc1 shape, bf16 out, bias")
run(M, 4096, 1024, True, False, False, True, 1, "fc1 shape, bf16 out, bias+gelu")
run(M, 1024, 4096, True, True, True, False, 0, "fc2 shape, f32 out, bias+resid")
run(M, 1024, 1024, True, True, True, False, 0, "proj shape, f32 out, bias+resid")
run(M, 2048, 1024, True, False, False, True, 0, "qk shape, bf16 out, bias")
