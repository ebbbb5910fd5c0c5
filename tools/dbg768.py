import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pnp-ovss_amd"))
import numpy as np, torch
from pnp_ovss import config as C, synth
from pnp_ovss.hip import Engine
from oracle import blip_itm_np as OM
IMG=int(sys.argv[2]) if len(sys.argv)>2 else 768
cfg = C.blip_itm_small(IMG)
W = synth.synth_state_dict(cfg, 2)
_, imgs = synth.synth_images(1, IMG, seed=9)
ids, mask = synth.synth_tokens(cfg, [6], seed=3)
L = int(mask.sum(1).max())
SL=int(sys.argv[1]) if len(sys.argv)>1 else 7
e = Engine(cfg, max_batch=1, max_text_len=16, stash_layer=SL, bf16=False)
e.load_state_dict(W)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
out, logits = e.compute_gradcam(d(imgs), d(ids), d(mask), L, 9)
torch.cuda.synchronize()
maps, ref_logits, raw = OM.compute_gradcam(W, cfg, imgs, ids, mask, layers=[SL])
N = cfg.n_img_tokens; nst = (N + 63) // 64 * 64
emb = e.buffer("image_embeds")[: N * cfg.vit_dim].view(N, cfg.vit_dim).cpu().numpy()
ref_emb = OM.vit_forward(W, cfg, imgs)[0]
print("emb err", np.abs(emb - ref_emb).max())
print("logits", logits.cpu().numpy(), ref_logits)
P = e.buffer("P")[: 12 * L * nst].view(12, L, nst)[..., :N].cpu().numpy()
dP = e.buffer("dP")[: 12 * L * nst].view(12, L, nst)[..., :N].cpu().numpy()
print("P err", np.abs(P - raw[SL][0][0]).max(), "P max", P.max(), raw[SL][0][0].max())
print("dP err", np.abs(dP - raw[SL][1][0]).max(), "dP absmax", np.abs(dP).max(), np.abs(raw[SL][1][0]).max())
print("map err", np.abs(out.cpu().numpy() - maps[SL][:, 9]).max())
if SL < 11:
    H = cfg.txt_hidden
    nKt = 11 - SL
    Kt = e.buffer("Kt")[: nKt * H * nst].view(nKt, H, nst).cpu().numpy()
    pre = "text_encoder.encoder.layer.11.crossattention.self.key."
    Kref = ref_emb @ np.asarray(W[pre + "weight"]).T + np.asarray(W[pre + "bias"])
    print("Kt nan", np.isnan(Kt).sum(), "pad absmax", np.abs(Kt[..., N:]).max(), "err", np.abs(Kt[-1][:, :N] - Kref.T).max())
    dq = e.buffer("dq_xattn")[: L * H].view(L, H).cpu().numpy()
    dc = e.buffer("dctx_xattn")[: L * H].view(L, H).cpu().numpy()
    print("dq nan", np.isnan(dq).sum(), "dctx nan", np.isnan(dc).sum(), np.abs(dq[~np.isnan(dq)]).max() if (~np.isnan(dq)).any() else None)
    print("dq nan rows/cols", np.unique(np.where(np.isnan(dq))[0]), np.unique(np.where(np.isnan(dq))[1])[:20])

hl = e.buffer("h_last")[: L * cfg.txt_hidden].view(L, cfg.txt_hidden).cpu().numpy()
print("h_last nan", np.isnan(hl).sum(), np.unique(np.where(np.isnan(hl))[0]))

Pl = e.buffer("P_last")[: 12 * L * nst].view(12, L, nst).cpu().numpy()
print("P_last nan", np.isnan(Pl).sum(), "inf", np.isinf(Pl).sum(), "rowsum range", Pl[..., :N].sum(-1).min(), Pl[..., :N].sum(-1).max(), "pad", np.abs(Pl[..., N:]).max())
bad = ~np.isfinite(Pl)
print("bad heads", np.unique(np.where(bad)[0]), "rows", np.unique(np.where(bad)[1]), "cols", np.unique(np.where(bad)[2])[:10], np.unique(np.where(bad)[2])[-10:])
