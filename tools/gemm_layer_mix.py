"""One ViT-L block's four dense launches at the bench shape (35 images x 442 tokens), replayed REPS times:
fused q|k|v (N=3072), proj (+residual), fc1 (+GELU), fc2 (+residual).  Used under
`rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE --kernel-include-regex gemm_nt_wide` for roofline.traffic."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pnp-ovss_amd"))
import torch
from pnp_ovss import hip
lib = hip.load_library()
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B, N, D, F = 35, 442, 1024, 4096
M, Npad = B * N, 448
bf = torch.bfloat16
xn = torch.randn(M, D, device="cuda").to(bf); ctx = torch.randn(M, D, device="cuda").to(bf)
h1 = torch.randn(M, F, device="cuda").to(bf)
wqkv = (0.02 * torch.randn(3 * D, D, device="cuda")).to(bf); wproj = (0.02 * torch.randn(D, D, device="cuda")).to(bf)
wfc1 = (0.02 * torch.randn(F, D, device="cuda")).to(bf); wfc2 = (0.02 * torch.randn(D, F, device="cuda")).to(bf)
bq = torch.randn(3 * D, device="cuda"); bp = torch.randn(D, device="cuda"); b1 = torch.randn(F, device="cuda"); b2 = torch.randn(D, device="cuda")
x = torch.randn(M, D, device="cuda")
qkv = torch.empty(M, 3 * D, device="cuda", dtype=bf)
h1o = torch.empty(M, F, device="cuda", dtype=bf)
p = lambda t: t.data_ptr()
for _ in range(REPS):
    assert lib.pnp_op_gemm_ex(1, p(xn), D, p(wqkv), D, M, 3 * D, D, p(bq), None, 0, None, 0, p(qkv), 3 * D, 0, None) == 0
    assert lib.pnp_op_gemm_ex(1, p(ctx), D, p(wproj), D, M, D, D, p(bp), p(x), D, p(x), D, None, 0, 0, None) == 0
    assert lib.pnp_op_gemm_ex(1, p(xn), D, p(wfc1), D, M, F, D, p(b1), None, 0, None, 0, p(h1o), F, 1, None) == 0
    assert lib.pnp_op_gemm_ex(1, p(h1), F, p(wfc2), F, M, D, F, p(b2), p(x), D, p(x), D, None, 0, 0, None) == 0
torch.cuda.synchronize()
print("ok")
