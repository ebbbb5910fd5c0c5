"""CPU emulation (numpy oracle) of candidate ViT arithmetic for the parity mode: does split-bf16 ("bf16x3": a = a_hi +
a_lo in bf16, a.b ~ a_hi.b_hi + a_hi.b_lo + a_lo.b_hi on the bf16 MFMA with fp32 accumulation) keep the salience picks
of the fp32 reference?  Variants of the ViT (text stack + backward stay fp32 in all of them):
  f32        reference arithmetic
  bf16       operands rounded to bf16 (Linear and attention)
  x3         Linears split-bf16, attention fp32
  x3_att16   Linears split-bf16, attention operands (q, k, v, P) rounded to bf16
  x3_attx3   Linears and attention products split-bf16
usage: python tools/precision_emulate.py [B] [variants...]      (BLIP-ITM-large 336^2; minutes per variant on 8 cores)
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pnp-ovss_amd")):
    sys.path.insert(0, p)
import numpy as np

from pnp_ovss import config as C, synth
from oracle import blip_itm_np as M
from oracle import pipeline_np as OP

F32 = np.float32


def bf16(x):
    u = np.ascontiguousarray(x, dtype=F32).view(np.uint32)
    r = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return r.view(F32)


def split(x):
    hi = bf16(x)
    return hi, bf16(x - hi)


def mm_x3(a, b):
    ah, al = split(a)
    bh, bl = split(b)
    return (ah @ bh + (ah @ bl + al @ bh)).astype(F32)


def make_vit(linear_mode, att_mode):
    def lin(x, w, b=None):
        if linear_mode == "f32":
            y = x @ w.T
        elif linear_mode == "bf16":
            y = bf16(x) @ bf16(w).T
        else:
            y = mm_x3(x, w.T)
        return (y + b).astype(F32) if b is not None else y.astype(F32)

    def att_mm(a, b):
        if att_mode == "f32":
            return a @ b
        if att_mode == "bf16":
            return bf16(a) @ bf16(b)
        return mm_x3(a, b)

    def vit(W, cfg, img):
        v = "visual_encoder."
        B = img.shape[0]
        P, D, H = cfg.grid, cfg.vit_dim, cfg.vit_heads
        dh = D // H
        ps = cfg.patch
        x = img.reshape(B, 3, P, ps, P, ps).transpose(0, 2, 4, 1, 3, 5).reshape(B, P * P, 3 * ps * ps)
        x = lin(x, W[v + "patch_embed.proj.weight"].reshape(D, -1), W[v + "patch_embed.proj.bias"])
        cls = np.broadcast_to(W[v + "cls_token"], (B, 1, D))
        x = (np.concatenate([cls, x], axis=1) + W[v + "pos_embed"][:, : P * P + 1]).astype(F32)
        N = x.shape[1]
        scale = F32(dh ** -0.5)
        for i in range(cfg.vit_depth):
            b = f"{v}blocks.{i}."
            h, _, _ = M.layer_norm(x, W[b + "norm1.weight"], W[b + "norm1.bias"], cfg.vit_ln_eps)
            qkv = lin(h, W[b + "attn.qkv.weight"], W[b + "attn.qkv.bias"])
            if linear_mode == "bf16":
                qkv = bf16(qkv)
            qkv = qkv.reshape(B, N, 3, H, dh).transpose(2, 0, 3, 1, 4)
            q, k, vv = qkv[0], qkv[1], qkv[2]
            att = M.softmax(att_mm(q, k.transpose(0, 1, 3, 2)) * scale)
            ctx = att_mm(att, vv).transpose(0, 2, 1, 3).reshape(B, N, D)
            x = x + lin(ctx, W[b + "attn.proj.weight"], W[b + "attn.proj.bias"])
            h, _, _ = M.layer_norm(x, W[b + "norm2.weight"], W[b + "norm2.bias"], cfg.vit_ln_eps)
            h = M.gelu(lin(h, W[b + "mlp.fc1.weight"], W[b + "mlp.fc1.bias"]))
            x = (x + lin(h, W[b + "mlp.fc2.weight"], W[b + "mlp.fc2.bias"])).astype(F32)
        out, _, _ = M.layer_norm(x, W[v + "norm.weight"], W[v + "norm.bias"], cfg.vit_ln_eps)
        return out
    return vit


VARIANTS = {"f32": ("f32", "f32"), "bf16": ("bf16", "bf16"), "x3": ("x3", "f32"), "x3_att16": ("x3", "bf16"),
            "x3_attx3": ("x3", "x3")}

if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    names = sys.argv[2:] or ["f32", "x3", "x3_att16"]
    if "f32" not in names:
        names = ["f32"] + names
    cfg = C.blip_itm_large(336)
    W = synth.synth_state_dict(cfg, 0)
    _, imgs = synth.synth_images(B, 336, seed=1234, noise=4)
    ids, mask = synth.synth_tokens(cfg, [20] * B, seed=1234)
    res = {}
    orig = M.vit_forward
    for nm in names:
        M.vit_forward = make_vit(*VARIANTS[nm])
        t0 = time.time()
        g0, agg, picks = OP.drop_loop(W, cfg, imgs, ids, mask, 4, 7, 9)
        res[nm] = (g0, agg, picks)
        print(nm, f"{time.time() - t0:.0f} s", flush=True)
    M.vit_forward = orig
    ref = res["f32"]
    out = {}
    for nm in names[1:]:
        g0, agg, picks = res[nm]
        same = [[set(sum((ref[2][i][b] for i in range(it + 1)), [])) == set(sum((picks[i][b] for i in range(it + 1)), []))
                 for b in range(B)] for it in range(4)]
        out[nm] = {"pick_sets_equal_per_iter": [int(sum(s)) for s in same], "of": B,
                   "g0_rel_err": float(np.linalg.norm(g0 - ref[0]) / np.linalg.norm(ref[0])),
                   "agg_rel_err": float(np.linalg.norm(agg - ref[1]) / np.linalg.norm(ref[1]))}
    print(json.dumps(out, indent=1))
