import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pnp-ovss_amd"))
import torch
from pnp_ovss import hip
lib = hip.load_library()
M, N, K = 15470, 4096, 1024
A = torch.randn(M, K, device="cuda").to(torch.bfloat16); B = torch.randn(N, K, device="cuda").to(torch.bfloat16)
out = torch.empty(M, N, device="cuda")
for it in range(5):
    lib.pnp_op_gemm(1, A.data_ptr(), K, B.data_ptr(), K, M, N, K, None, None, 0, out.data_ptr(), N, 0, None)
torch.cuda.synchronize()
