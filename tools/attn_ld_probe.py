"""Split-bf16 ViT attention against the row stride of the fused q|k|v buffer (does the K/V tile fetch -- 64 rows x 128 B at the
row stride -- depend on how the rows map to L2 channels / cache sets?)."""
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
B, H, N = 35, 16, 442
D = H * 64
for pad in (0, 64, 0, 64, 128, 0, 256, 0):
    ld = 3 * D + pad
    torch.manual_seed(0)
    qkv = torch.randn(B * N, ld, device="cuda")
    hi = torch.empty(qkv.shape, device="cuda", dtype=torch.bfloat16)
    lo = torch.empty_like(hi)
    assert lib.pnp_op_split(qkv.data_ptr(), hi.data_ptr(), lo.data_ptr(), qkv.numel(), None) == 0
    ch = torch.empty(B * N, D, device="cuda", dtype=torch.bfloat16)
    cl = torch.empty_like(ch)
    call = lambda: lib.pnp_op_vit_attention_x3(hi.data_ptr(), lo.data_ptr(), ld, D, ch.data_ptr(), cl.data_ptr(), B, H, N, 0.125, None)
    for _ in range(3):
        assert call() == 0
    torch.cuda.synchronize()
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(30):
            call()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 30)
    print(f"row stride {ld * 2} B (pad {pad * 2} B): {min(ts) * 1e6:7.1f} us", flush=True)
