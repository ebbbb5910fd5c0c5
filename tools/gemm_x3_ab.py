"""A/B of split-bf16 wide-GEMM builds IN ONE PROCESS on one GPU (boxes differ by several percent in clock, so variants are
only comparable inside a run): each variant is a library file next to the product one (e.g. built from another commit or with
`-DPNP_X3_ABLATE=n`, the timing-only ablations of csrc/gemm_x3.hip: garbage results) plus optional environment knobs of a DEV
build, timed interleaved over several rounds, best-of reported.  This is how round 4 compared the 16x16x32 kernel with the
32x32x16 kernel of round 3 (numbers in DESIGN.md section 5).

    python tools/gemm_x3_ab.py [shape ...] name=lib[:ENV=V,...] ...
    e.g.  new=libpnp_hip.so other=libpnp_hip_other.so nodma=libpnp_hip_abl1.so"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "pnp-ovss_amd", "pnp_ovss")
import torch

args = sys.argv[1:]
specs = [a for a in args if "=" in a.split(":")[0]] or ["product=libpnp_hip.so", "dev=libpnp_hip_dev.so"]
only = [a for a in args if "=" not in a]
KNOBS = ("PNP_GEMM_GM", "PNP_GEMM_GRID")
vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
variants = []
for sp in specs:
    name, rest = sp.split("=", 1)
    libname, _, env = rest.partition(":")
    lib = C.CDLL(os.path.join(LIBDIR, libname))
    lib.pnp_op_split.restype = i32
    lib.pnp_op_split.argtypes = [vp, vp, vp, i64, vp]
    lib.pnp_op_gemm_x3.restype = i32
    lib.pnp_op_gemm_x3.argtypes = [vp, vp, i32, vp, vp, i32, i32, i32, i32, vp, i32, vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, vp]
    variants.append((name, lib, env))
lib0 = variants[0][1]


def split(t):
    hi = torch.empty(t.shape, device="cuda", dtype=torch.bfloat16)
    lo = torch.empty_like(hi)
    assert lib0.pnp_op_split(t.data_ptr(), hi.data_ptr(), lo.data_ptr(), t.numel(), None) == 0
    return hi, lo


def setenv(spec):
    for k in KNOBS:
        os.environ.pop(k, None)
    for kv in filter(None, spec.split(",")):
        k, v = kv.split("=")
        os.environ[k] = v


M = 15470
for tag, N, K, kind in (("qkv", 3072, 1024, "split"), ("fc1", 4096, 1024, "gelu"), ("fc2", 1024, 4096, "resid"), ("proj", 1024, 1024, "resid"),
                        ("crossk", 9216, 1024, "bias")):
    if only and tag not in only:
        continue
    torch.manual_seed(0)
    A = torch.randn(M, K, device="cuda")
    B = 0.02 * torch.randn(N, K, device="cuda")
    (Ah, Al), (Bh, Bl) = split(A), split(B)
    bias = torch.randn(N, device="cuda")
    p = lambda t: t.data_ptr() if t is not None else None
    if kind == "resid":
        out = torch.randn(M, N, device="cuda")
        call = lambda l: l.pnp_op_gemm_x3(p(Ah), p(Al), K, p(Bh), p(Bl), K, M, N, K, p(bias), 0, p(out), N, p(out), N, None, None, 0, 0, 0, 0, None)
    elif kind == "bias":
        out = torch.empty(M, N, device="cuda")
        call = lambda l: l.pnp_op_gemm_x3(p(Ah), p(Al), K, p(Bh), p(Bl), K, M, N, K, p(bias), 0, None, 0, p(out), N, None, None, 0, 0, 0, 0, None)
    else:
        hi = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        lo = torch.empty_like(hi)
        call = lambda l: l.pnp_op_gemm_x3(p(Ah), p(Al), K, p(Bh), p(Bl), K, M, N, K, p(bias), 0, None, 0, None, 0, p(hi), p(lo), N,
                                           1 if kind == "gelu" else 0, 0, 0, None)
    best = {v[0]: 1e9 for v in variants}
    for rnd in range(4):
        for name, lib, env in variants:
            setenv(env)
            for _ in range(5):
                assert call(lib) == 0
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(40):
                call(lib)
            torch.cuda.synchronize()
            best[name] = min(best[name], (time.perf_counter() - t0) / 40)
    fl = 6.0 * M * N * K
    print(f"{tag:6s} " + "  ".join(f"{n} {best[n] * 1e6:6.1f} us ({fl / best[n] / 2.5e15:.3f})" for n, _, _ in variants), flush=True)
