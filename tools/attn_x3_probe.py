"""Split-bf16 ViT attention at the bench shapes: back-to-back launch time and error against float64 attention of the original
fp32 q, k, v.  `--dev` uses the DEV library, where PNP_ATTN_OLD=1 selects the un-pipelined kernel for comparison."""
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


def run(B, H, N):
    D = H * 64
    torch.manual_seed(0)
    qkv = torch.randn(B * N, 3 * D, device="cuda")
    hi = torch.empty(qkv.shape, device="cuda", dtype=torch.bfloat16)
    lo = torch.empty_like(hi)
    assert lib.pnp_op_split(qkv.data_ptr(), hi.data_ptr(), lo.data_ptr(), qkv.numel(), None) == 0
    ch = torch.empty(B * N, D, device="cuda", dtype=torch.bfloat16)
    cl = torch.empty_like(ch)
    call = lambda: lib.pnp_op_vit_attention_x3(hi.data_ptr(), lo.data_ptr(), 3 * D, D, ch.data_ptr(), cl.data_ptr(), B, H, N, 0.125, None)
    for _ in range(3):
        assert call() == 0
    torch.cuda.synchronize()
    q = qkv[:N, :64].double()
    k = qkv[:N, D:D + 64].double()
    v = qkv[:N, 2 * D:2 * D + 64].double()
    ref = torch.softmax(q @ k.t() * 0.125, dim=-1) @ v
    got = (ch[:N, :64].float() + cl[:N, :64].float()).double()
    err = float((got - ref).abs().max() / ref.abs().max())
    n = 50
    t0 = time.perf_counter()
    for _ in range(n):
        call()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    fl = 4.0 * N * N * 64 * H * B
    print(f"B={B} H={H} N={N}: {dt * 1e6:7.1f} us  algorithmic {fl / dt / 1e12:6.0f} TF  issued {3 * fl / dt / 1e12:6.0f} TF  relerr {err:.1e}", flush=True)


run(35, 16, 442)
if "--all" in sys.argv:
    run(8, 16, 2305)
    run(3, 2, 17)

