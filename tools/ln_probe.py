"""GPU box: LayerNorm kernel at the ViT shape of the bench (15470 x 1024 fp32 in, fp32 out through the operator entry): us per call, TB/s."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pnp-ovss_amd"))
import torch
from pnp_ovss import hip
lib = hip.load_library()
rows, D = 15470, 1024
x = torch.randn(rows, D, device="cuda"); w = torch.randn(D, device="cuda"); b = torch.randn(D, device="cuda")
y = torch.empty(rows, D, device="cuda")
def call():
    return lib.pnp_op_layernorm(x.data_ptr(), w.data_ptr(), b.data_ptr(), 1e-6, rows, D, y.data_ptr(), None)
for _ in range(5): assert call() == 0
torch.cuda.synchronize()
n = 500; t0 = time.perf_counter()
for _ in range(n): call()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"layernorm {rows}x{D}: {dt*1e6:.1f} us, {(rows*D*8)/dt/1e12:.2f} TB/s")
ref = torch.nn.functional.layer_norm(x, (D,), w, b, 1e-6)
print("max err vs torch:", float((y - ref).abs().max()))
