"""Throughput and error of the ViT-L block GEMM shapes at the bench batch (PNP_GEMM_VARIANT forces a kernel)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pnp-ovss_amd"))
import torch
from pnp_ovss import hip
lib = hip.load_library()           # PNP_HIP_LIB=<dev build> + PNP_GEMM_ABLATE=3: time the 16x16x32 MFMA shape (results are garbage)
CHECK = "exp" not in os.environ.get("PNP_HIP_LIB", "")
def run(M, N, K, bias, resid, f32out, tout, mode, tag, check=True):
    torch.manual_seed(0)
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16); B = (0.02 * torch.randn(N, K, device="cuda")).to(torch.bfloat16)
    bi = torch.randn(N, device="cuda") if bias else None
    rs = torch.randn(M, N, device="cuda") if resid else None
    of = torch.empty(M, N, device="cuda") if f32out else None
    ot = torch.empty(M, N, device="cuda", dtype=torch.bfloat16) if tout else None
    p = lambda t: t.data_ptr() if t is not None else None
    def call():
        return lib.pnp_op_gemm_ex(1, A.data_ptr(), K, B.data_ptr(), K, M, N, K, p(bi), p(rs), N, p(of), N, p(ot), N, mode, None)
    for _ in range(3): assert call() == 0
    torch.cuda.synchronize()
    err = -1.0
    if check and CHECK:
        ref = A.float() @ B.float().t()
        if bias: ref += bi
        if mode == 1: ref = torch.nn.functional.gelu(ref)
        if resid: ref += rs
        got = of if f32out else ot.float()
        err = float((got - ref).abs().max() / ref.abs().max())
    n = 200; t0 = time.perf_counter()
    for _ in range(n): call()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"{tag:30s} M={M} N={N} K={K}: {2*M*N*K/dt/1e12:6.0f} TF ({dt*1e6:.0f} us) relerr {err:.1e}", flush=True)
M = 15470
run(M, 3072, 1024, True, False, False, True, 0, "qkv  bf16 out, bias")
run(M, 4096, 1024, True, False, False, True, 1, "fc1  bf16 out, bias+gelu")
run(M, 1024, 4096, True, True, True, False, 0, "fc2  f32 out, bias+resid")
run(M, 1024, 1024, True, True, True, False, 0, "proj f32 out, bias+resid")
run(M, 9216, 1024, True, False, False, True, 0, "cross K (12 layers x 768), bias", check=False)
run(1000, 1000, 64, True, False, True, False, 0, "edge M,N ragged, K=64")
run(300, 520, 192, False, False, False, True, 0, "edge small")
