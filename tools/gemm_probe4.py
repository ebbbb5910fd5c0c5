"""Scratch probe: does a non-power-of-two leading dimension change GEMM throughput (L2 channel camping)?"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pnp-ovss_amd"))
import torch
from pnp_ovss import hip
lib = hip.load_library()
def run(M, N, K, pad_a, pad_b, tag):
    A = torch.randn(M, K + pad_a, device="cuda").to(torch.bfloat16); B = (0.02 * torch.randn(N, K + pad_b, device="cuda")).to(torch.bfloat16)
    ot = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    def call():
        return lib.pnp_op_gemm_ex(1, A.data_ptr(), K + pad_a, B.data_ptr(), K + pad_b, M, N, K, None, None, 0, None, 0, ot.data_ptr(), N, 0, None)
    for _ in range(3): assert call() == 0
    torch.cuda.synchronize(); n = 20; t0 = time.perf_counter()
    for _ in range(n): call()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"{tag:8s} M={M} N={N} K={K} padA={pad_a} padB={pad_b}: {2*M*N*K/dt/1e12:6.0f} TF ({dt*1e6:.0f} us)", flush=True)
M = 15470
for pa, pb in ((0, 0), (64, 64), (32, 32), (8, 8), (64, 0), (0, 64), (128, 128)):
    run(M, 3072, 1024, pa, pb, "qkv")
    run(M, 1024, 4096, pa, pb, "fc2")
