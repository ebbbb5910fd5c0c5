"""Operator-level check of pnp_op_xattn (all three modes) against a torch fp64 reference."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pnp-ovss_amd"))
import torch
from pnp_ovss.hip import load_library

def run(N, L, heads=12, B=1, zero_rows=False, seed=0):
    lib = load_library()
    g = torch.Generator().manual_seed(seed)
    H = heads * 64
    Npad = (N + 63) // 64 * 64
    K = torch.randn(B, N, H, generator=g) * 0.5
    V = torch.randn(B, N, H, generator=g) * 0.5
    q = torch.randn(B, L, H, generator=g)
    dctx = torch.randn(B, L, H, generator=g)
    if zero_rows:
        dctx[:, 1:] = 0
    def tr(a):                     # [B,N,H] -> [H, B*Npad]
        t = torch.zeros(H, B, Npad)
        t[:, :, :N] = a.permute(2, 0, 1)
        return t.reshape(H, B * Npad).contiguous()
    d = lambda a: a.contiguous().cuda()
    Kn, Vn, Kt, Vt, qd, dcd = d(K.reshape(B * N, H)), d(V.reshape(B * N, H)), d(tr(K)), d(tr(V)), d(q.reshape(B * L, H)), d(dctx.reshape(B * L, H))
    P = torch.zeros(B, heads, L, Npad, device="cuda")
    dP = torch.zeros(B, heads, L, Npad, device="cuda")
    ctx = torch.zeros(B * L, H, device="cuda")
    dq = torch.zeros(B * L, H, device="cuda")
    p = lambda t: t.data_ptr()
    assert lib.pnp_op_xattn(0, 0, p(Kn), H, p(Vt), B * Npad, Npad, p(qd), H, p(ctx), H, p(P), Npad, B, L, N, heads, None) == 0
    assert lib.pnp_op_xattn(0, 2, p(Vn), H, None, 0, Npad, p(dcd), H, None, 0, p(dP), Npad, B, L, N, heads, None) == 0
    assert lib.pnp_op_xattn(0, 1, p(Vn), H, p(Kt), B * Npad, Npad, p(dcd), H, p(dq), H, p(P), Npad, B, L, N, heads, None) == 0
    torch.cuda.synchronize()
    # reference
    f = lambda a: a.double().view(B, -1, heads, 64).permute(0, 2, 1, 3)
    Kh, Vh, qh, dch = f(K), f(V), f(q), f(dctx)
    S = qh @ Kh.transpose(-1, -2) / 8
    Pr = S.softmax(-1)
    ctxr = (Pr @ Vh).permute(0, 2, 1, 3).reshape(B * L, H)
    dPr = dch @ Vh.transpose(-1, -2)
    dS = Pr * (dPr - (dPr * Pr).sum(-1, keepdim=True))
    dqr = (dS @ Kh / 8).permute(0, 2, 1, 3).reshape(B * L, H)
    e = lambda a, b: float((a.cpu().double() - b).abs().max())
    print(f"N={N} L={L} zero_rows={zero_rows}: P {e(P[..., :N], Pr):.2e} ctx {e(ctx, ctxr):.2e} dP {e(dP[..., :N], dPr):.2e} "
          f"dq {e(dq, dqr):.2e} (|dq| max {float(dqr.abs().max()):.2e}) nan {int(torch.isnan(dq).sum())}")
    if torch.isnan(dq).any():
        idx = torch.isnan(dq).nonzero().cpu()
        print("  nan rows", idx[:, 0].unique().tolist(), "heads", (idx[:, 1] // 64).unique().tolist())

for a in sys.argv[1:]:
    N, L, z = a.split(",")
    run(int(N), int(L), zero_rows=bool(int(z)))
