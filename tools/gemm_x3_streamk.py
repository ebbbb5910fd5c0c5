"""Stream-K tail of the persistent split-bf16 GEMM (csrc/gemm_x3.hip), A/B per launch shape on the GPU box:
`pnp_set_tuning("streamk", 0)` (whole tiles only) against 2 (the last partial tile round cut along K over all CUs) and 1 (what
the cost model picks), back-to-back launch time through the op-level entry point, result against float64 on a sample of rows,
run-to-run bit-identity of the split form, and the give-up word of its bounded spins.

usage: python tools/gemm_x3_streamk.py [--reps 60] [tags...]      -> table on stdout (profiles/r06_streamk.txt)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pnp-ovss_amd"))
import torch
from pnp_ovss import hip

REPS = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 60
ONLY = [a for a in sys.argv[1:] if not a.startswith("--") and not a.isdigit()]
if "--lib" in sys.argv:                                     # a named library variant next to the product one (timing-only ablations)
    hip.LIB_PATH = os.path.join(os.path.dirname(hip.LIB_PATH), sys.argv[sys.argv.index("--lib") + 1])
    ONLY = [a for a in ONLY if not a.endswith(".so")]
CHECK = "--no-check" not in sys.argv
lib = hip.load_library()
CUS = torch.cuda.get_device_properties(0).multi_processor_count


def split(t):
    hi = torch.empty(t.shape, device="cuda", dtype=torch.bfloat16)
    lo = torch.empty_like(hi)
    assert lib.pnp_op_split(t.data_ptr(), hi.data_ptr(), lo.data_ptr(), t.numel(), None) == 0
    return hi, lo


def run(tag, M, N, K, kind):
    if ONLY and tag not in ONLY:
        return
    torch.manual_seed(0)
    A = torch.randn(M, K, device="cuda")
    B = 0.02 * torch.randn(N, K, device="cuda")
    (Ah, Al), (Bh, Bl) = split(A), split(B)
    bias = torch.randn(N, device="cuda")
    p = lambda t: t.data_ptr() if t is not None else None
    res0 = torch.randn(M, N, device="cuda") if kind == "resid" else None
    out = torch.empty(M, N, device="cuda") if kind in ("resid", "bias") else None
    hi = torch.empty(M, N, device="cuda", dtype=torch.bfloat16) if out is None else None
    lo = torch.empty_like(hi) if hi is not None else None
    if kind == "resid":
        call = lambda: lib.pnp_op_gemm_x3(p(Ah), p(Al), K, p(Bh), p(Bl), K, M, N, K, p(bias), 0, p(res0), N, p(out), N, None, None, 0, 0, 0, 0, None)
    elif kind == "bias":
        call = lambda: lib.pnp_op_gemm_x3(p(Ah), p(Al), K, p(Bh), p(Bl), K, M, N, K, p(bias), 0, None, 0, p(out), N, None, None, 0, 0, 0, 0, None)
    else:
        call = lambda: lib.pnp_op_gemm_x3(p(Ah), p(Al), K, p(Bh), p(Bl), K, M, N, K, p(bias), 0, None, 0, None, 0, p(hi), p(lo), N,
                                          1 if kind == "gelu" else 0, 0, 0, None)
    rows = torch.randint(0, M, (64,), device="cuda")
    ref = A[rows].double() @ B.double().t() + bias.double()
    if kind == "resid":
        ref = ref + res0[rows].double()
    if kind == "gelu":
        ref = torch.nn.functional.gelu(ref)
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    times, errs, same, used = {}, {}, {}, {}
    for mode in (0, 2, 1):
        hip.set_tuning("streamk", mode)
        n0 = hip.streamk_status_ops()[0]
        for _ in range(3):
            assert call() == 0
        torch.cuda.synchronize()
        got = out[rows].double() if out is not None else hi[rows].double() + lo[rows].double()
        errs[mode] = float((got - ref).abs().max() / ref.abs().max())
        first = (out if out is not None else hi).clone()
        first_lo = lo.clone() if lo is not None else None
        assert call() == 0
        torch.cuda.synchronize()
        same[mode] = bool(torch.equal(first, out if out is not None else hi)) and (lo is None or bool(torch.equal(first_lo, lo)))
        t0 = time.perf_counter()
        for _ in range(REPS):
            call()
        torch.cuda.synchronize()
        times[mode] = (time.perf_counter() - t0) / REPS * 1e6
        used[mode] = hip.streamk_status_ops()[0] - n0 > 0
    gave_up = hip.streamk_status_ops()[1]
    hip.set_tuning("streamk", 1)
    fl = 2.0 * M * N * K
    print(f"{tag:12s} M={M:6d} N={N:5d} K={K:5d} tiles {tiles:5d} = {tiles / CUS:5.2f} rounds | whole tiles {times[0]:7.1f} us ({fl / times[0] / 1e6:5.0f} TF) | "
          f"stream-K {times[2]:7.1f} us ({100 * (times[2] / times[0] - 1):+5.1f} %) | auto {'SK' if used[1] else '--'} {times[1]:7.1f} us | "
          f"relerr {errs[0]:.1e} / {errs[2]:.1e} | bit-identical reruns {same[0]} / {same[2]} | gave up {gave_up}", flush=True)
    if CHECK:       # a (hi, lo) bf16 pair carries 16 significant bits: 2^-17 of the row maximum
        assert gave_up == 0 and same[2] and errs[2] < (1.2e-5 if kind in ("split", "gelu") else 5e-6), (tag, gave_up, same, errs)


print(f"# {torch.cuda.get_device_name(0)}, {CUS} CUs; {REPS} back-to-back launches per cell; error = max |got - float64| / max |ref| over 64 sampled rows")
M = 35 * 442                      # the bench batch: 60.4 row tiles
if not ONLY:
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):           # clock / allocator warm-up, not reported
        run("warmup", M, 3072, 1024, "split")
run("qkv", M, 3072, 1024, "split")
run("proj", M, 1024, 1024, "resid")
run("fc1", M, 4096, 1024, "gelu")
run("fc2", M, 1024, 4096, "resid")
run("crosskv", M, 18432, 1024, "bias")
M = 8 * 2305                      # ADE20K shape, 8 images: 72.03 row tiles
run("ade8-qkv", M, 3072, 1024, "split")
run("ade8-proj", M, 1024, 1024, "resid")
run("ade8-fc1", M, 4096, 1024, "gelu")
run("ade8-fc2", M, 1024, 4096, "resid")
M = 7 * 2305                      # 7 images: 63.03 row tiles (what round 5 benchmarked to dodge the partial round)
run("ade7-qkv", M, 3072, 1024, "split")
run("ade7-proj", M, 1024, 1024, "resid")
run("ade7-fc2", M, 1024, 4096, "resid")
for b in (12, 20, 29):            # ragged last batches of a shard
    run(f"b{b}-qkv", b * 442, 3072, 1024, "split")
    run(f"b{b}-proj", b * 442, 1024, 1024, "resid")
    run(f"b{b}-fc1", b * 442, 4096, 1024, "gelu")
    run(f"b{b}-fc2", b * 442, 1024, 4096, "resid")
