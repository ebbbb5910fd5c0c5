"""Scratch probe: per-workgroup phase timing of the wide-tile GEMM.  Needs the DEV build of the library
(make -C pnp-ovss_amd/csrc DEV=1 OBJDIR=build_dev OUT=../pnp_ovss/libpnp_hip_dev.so) and PNP_GEMM_STAMPS=1."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pnp-ovss_amd"))
import numpy as np, torch
from pnp_ovss import hip
hip.LIB_PATH = os.path.join(os.path.dirname(hip.LIB_PATH), "libpnp_hip_dev.so")
lib = hip.load_library()
def run(M, N, K, bias, resid, f32out, tout, mode, tag):
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
    nb = ((M + 255) // 256) * ((N + 255) // 256)
    if os.environ.get("PNP_GEMM_PERSIST", "1") != "0" and int(os.environ.get("TILE", "256")) == 256: nb = min(nb, 256)
    st = np.zeros((nb, 8), dtype=np.uint64)
    assert lib.pnp_dbg_gemm_stamps(st.ctypes.data, nb) == 0
    cyc = st[:, :4].astype(np.int64); wall = st[:, 4:].astype(np.int64)
    t0 = wall[:, 0].min()
    us = (wall - t0) / 100.0
    d = np.diff(us, axis=1)
    dc = np.diff(cyc, axis=1)
    ghz = dc.sum(1) / (d.sum(1) * 1e3)
    nk = K // 64
    print(f"{tag}: M={M} N={N} K={K} blocks={nb}  kernel span {us[:, 3].max():.1f} us; clock {np.median(ghz):.2f} GHz")
    print(f"  per block median us: prologue {np.median(d[:,0]):.2f}  main {np.median(d[:,1]):.2f} ({np.median(d[:,1])/nk:.3f}/slab = {np.median(dc[:,1])/nk:.0f} clk)  epilogue {np.median(d[:,2]):.2f}")
    order = np.argsort(us[:, 0])
    print(f"  first tile: main loop ends {np.median(us[:,2]-us[:,0]):.2f}, epilogue (stage + issue) {np.median(us[:,1]-us[:,2]):.2f} us")
    ntile_all = ((M + 255) // 256) * ((N + 255) // 256)
    print(f"  persistent: tiles/block {ntile_all / nb:.2f}; block total median {np.median(us[:,3]-us[:,0]):.1f} us, max {np.max(us[:,3]-us[:,0]):.1f}; first tile main-end at {np.median(us[:,2]-us[:,0]):.1f}")
    print("  start times (us) quantiles", np.round(np.quantile(us[:, 0], [0, .25, .5, .75, 1]), 1), " end", np.round(np.quantile(us[:, 3], [0, .25, .5, .75, 1]), 1))
M = 15470
run(M, 3072, 1024, True, False, False, True, 0, "qkv")
run(M, 4096, 1024, True, False, False, True, 1, "fc1")
run(M, 1024, 4096, True, True, True, False, 0, "fc2")
run(M, 1024, 4096, False, False, False, True, 0, "fc2 plain")
run(M, 1024, 1024, True, True, True, False, 0, "proj")
