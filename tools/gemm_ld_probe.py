"""Split-bf16 wide GEMM against the row strides of its operands: power-of-two strides (2 KB / 8 KB rows) against the same rows
padded by 128 B ... (L2 channel / cache-set mapping of a k-slab's 256 rows x 64 B)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pnp-ovss_amd"))
import torch
from pnp_ovss import hip

lib = hip.load_library()


def split_into(t, ld):
    hi = torch.zeros(t.shape[0], ld, device="cuda", dtype=torch.bfloat16)
    lo = torch.zeros_like(hi)
    h = torch.empty(t.shape, device="cuda", dtype=torch.bfloat16)
    l = torch.empty_like(h)
    assert lib.pnp_op_split(t.data_ptr(), h.data_ptr(), l.data_ptr(), t.numel(), None) == 0
    hi[:, :t.shape[1]] = h
    lo[:, :t.shape[1]] = l
    return hi, lo


def run(tag, M, N, K, kind, pa, pb, po):
    torch.manual_seed(0)
    A = torch.randn(M, K, device="cuda")
    B = 0.02 * torch.randn(N, K, device="cuda")
    (Ah, Al), (Bh, Bl) = split_into(A, K + pa), split_into(B, K + pb)
    bias = torch.randn(N, device="cuda")
    p = lambda t: t.data_ptr() if t is not None else None
    ldo = N + po
    if kind == "resid":
        out = torch.randn(M, ldo, device="cuda")
        call = lambda: lib.pnp_op_gemm_x3(p(Ah), p(Al), K + pa, p(Bh), p(Bl), K + pb, M, N, K, p(bias), 0, p(out), ldo, p(out), ldo, None, None, 0, 0, 0, 0, None)
    else:
        hi = torch.empty(M, ldo, device="cuda", dtype=torch.bfloat16)
        lo = torch.empty_like(hi)
        call = lambda: lib.pnp_op_gemm_x3(p(Ah), p(Al), K + pa, p(Bh), p(Bl), K + pb, M, N, K, p(bias), 0, None, 0, None, 0, p(hi), p(lo), ldo,
                                          1 if kind == "gelu" else 0, 0, 0, None)
    for _ in range(3):
        assert call() == 0
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(40):
            call()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 40)
    fl = 2.0 * M * N * K
    print(f"{tag:5s} pad A {pa * 2:4d} B  W {pb * 2:4d} B  out {po:4d} el: {best * 1e6:7.1f} us  issued {3 * fl / best / 2.5e15:.3f} of 2.5 PF", flush=True)


M = 15470
for tag, N, K, kind in (("qkv", 3072, 1024, "split"), ("fc1", 4096, 1024, "gelu"), ("fc2", 1024, 4096, "resid"), ("proj", 1024, 1024, "resid")):
    for pa, pb, po in ((0, 0, 0), (64, 0, 0), (0, 64, 0), (64, 64, 0), (64, 64, 64), (128, 128, 0), (32, 32, 0)):
        run(tag, M, N, K, kind, pa, pb, po)
