"""ORACLE (test infrastructure, not product code): numpy fp32 restatement of the reference's
BLIP-ITM forward + GradCAM path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import this; the product path (pnp-ovss_amd/) never does.

Restates, citing the reference files under `/root/reference/Files to replace for BLIP/` (B/):
  * ViT-L/16 forward                     B/vit.py:274-290, Attention :91-121, Block :164-167, Mlp :45-51
  * BERT text stack, mode="multimodal"    B/med.py:88-123 (embeddings), :191-311 (attention),
                                          :321-325 (self-output), :393-411 (FFN), :444-519 (layer)
  * additive masks (1-m)*-10000           B/med.py:776-852
  * ITM head + enc-token substitution     B/blip_image_text_matching.py:217-249
  * loss = logits[:,1].sum(); backward    B/blip_image_text_matching.py:399-404
    -> here an *analytic* backward through the text layers only (dL/dP of the cross-attention
       probabilities is independent of ViT / weight gradients)
  * gradcam = P[...,1:] * relu(dP[...,1:]) * mask, clamp, drop [ENC] row
                                          B/blip_image_text_matching.py:411-435

  * checkpoint pos-embed re-tiling        B/base_model.py:44-73 (interpolate_pos_embed: bicubic, align_corners=False)

Pinned against golden vectors generated from the reference itself (tests/golden/make_golden.py).
"""
import math

import numpy as np
from scipy.special import erf

F32 = np.float32


def _p(W, name):
    return W[name]


def layer_norm(x, w, b, eps):
    mu = x.mean(axis=-1, keepdims=True, dtype=F32)
    xc = x - mu
    var = (xc * xc).mean(axis=-1, keepdims=True, dtype=F32)
    rstd = (1.0 / np.sqrt(var + F32(eps))).astype(F32)
    xhat = xc * rstd
    return (xhat * w + b).astype(F32), xhat.astype(F32), rstd


def layer_norm_bwd(dy, w, xhat, rstd):
    dxh = dy * w
    m1 = dxh.mean(axis=-1, keepdims=True, dtype=F32)
    m2 = (dxh * xhat).mean(axis=-1, keepdims=True, dtype=F32)
    return (rstd * (dxh - m1 - xhat * m2)).astype(F32)


def gelu(x):
    return (0.5 * x * (1.0 + erf(x / math.sqrt(2.0)))).astype(F32)


def gelu_grad(x):
    cdf = 0.5 * (1.0 + erf(x / math.sqrt(2.0)))
    pdf = np.exp(-0.5 * x * x) / math.sqrt(2.0 * math.pi)
    return (cdf + x * pdf).astype(F32)


def softmax(x):
    m = x.max(axis=-1, keepdims=True)
    e = np.exp(x - m)
    return (e / e.sum(axis=-1, keepdims=True, dtype=F32)).astype(F32)


def linear(x, w, b=None):
    y = x @ w.T
    if b is not None:
        y = y + b
    return y.astype(F32)


# ------------------------------------------------------------------------------------- ViT

def vit_forward(W, cfg, img):
    """img (B,3,S,S) fp32 -> image_embeds (B,N,D).  B/vit.py:274-290."""
    v = "visual_encoder."
    B = img.shape[0]
    P, D, H = cfg.grid, cfg.vit_dim, cfg.vit_heads
    dh = D // H
    ps = cfg.patch
    x = img.reshape(B, 3, P, ps, P, ps).transpose(0, 2, 4, 1, 3, 5).reshape(B, P * P, 3 * ps * ps)
    x = linear(x, _p(W, v + "patch_embed.proj.weight").reshape(D, -1), _p(W, v + "patch_embed.proj.bias"))
    cls = np.broadcast_to(_p(W, v + "cls_token"), (B, 1, D))
    x = np.concatenate([cls, x], axis=1) + _p(W, v + "pos_embed")[:, : P * P + 1]
    x = x.astype(F32)
    N = x.shape[1]
    scale = F32(dh ** -0.5)
    for i in range(cfg.vit_depth):
        b = f"{v}blocks.{i}."
        h, _, _ = layer_norm(x, _p(W, b + "norm1.weight"), _p(W, b + "norm1.bias"), cfg.vit_ln_eps)
        qkv = linear(h, _p(W, b + "attn.qkv.weight"), _p(W, b + "attn.qkv.bias"))
        qkv = qkv.reshape(B, N, 3, H, dh).transpose(2, 0, 3, 1, 4)
        q, k, vv = qkv[0], qkv[1], qkv[2]
        att = softmax((q @ k.transpose(0, 1, 3, 2)) * scale)
        ctx = (att @ vv).transpose(0, 2, 1, 3).reshape(B, N, D)
        x = x + linear(ctx, _p(W, b + "attn.proj.weight"), _p(W, b + "attn.proj.bias"))
        h, _, _ = layer_norm(x, _p(W, b + "norm2.weight"), _p(W, b + "norm2.bias"), cfg.vit_ln_eps)
        h = gelu(linear(h, _p(W, b + "mlp.fc1.weight"), _p(W, b + "mlp.fc1.bias")))
        x = (x + linear(h, _p(W, b + "mlp.fc2.weight"), _p(W, b + "mlp.fc2.bias"))).astype(F32)
    out, _, _ = layer_norm(x, _p(W, v + "norm.weight"), _p(W, v + "norm.bias"), cfg.vit_ln_eps)
    return out


# ------------------------------------------------------------------------------------- text

def _heads(x, nh):
    B, L, Hd = x.shape
    return x.reshape(B, L, nh, Hd // nh).transpose(0, 2, 1, 3)


def _merge(x):
    B, nh, L, dh = x.shape
    return x.transpose(0, 2, 1, 3).reshape(B, L, nh * dh)


def text_forward(W, cfg, ids, att_mask, image_embeds):
    """ids/att_mask (B,L) (already cut to the longest caption; ids[:,0] is replaced by enc_token_id
    as B/blip_image_text_matching.py:238-239 does).  Returns (logits (B,2), caches)."""
    t = "text_encoder."
    B, L = ids.shape
    nh = cfg.txt_heads
    dh = cfg.txt_hidden // nh
    ids = ids.copy()
    ids[:, 0] = cfg.enc_token_id
    emb = _p(W, t + "embeddings.word_embeddings.weight")[ids] + \
        _p(W, t + "embeddings.position_embeddings.weight")[None, :L]
    h, _, _ = layer_norm(emb.astype(F32), _p(W, t + "embeddings.LayerNorm.weight"),
                         _p(W, t + "embeddings.LayerNorm.bias"), cfg.txt_ln_eps)
    ext = ((1.0 - att_mask.astype(F32)) * F32(-10000.0))[:, None, None, :]
    inv_sqrt = F32(1.0 / math.sqrt(dh))
    caches = []
    for i in range(cfg.txt_layers):
        b = f"{t}encoder.layer.{i}."
        c = {"h_in": h}
        # self attention
        a = b + "attention."
        q = _heads(linear(h, _p(W, a + "self.query.weight"), _p(W, a + "self.query.bias")), nh)
        k = _heads(linear(h, _p(W, a + "self.key.weight"), _p(W, a + "self.key.bias")), nh)
        v = _heads(linear(h, _p(W, a + "self.value.weight"), _p(W, a + "self.value.bias")), nh)
        Ps = softmax((q @ k.transpose(0, 1, 3, 2)) * inv_sqrt + ext)
        ctx = _merge(Ps @ v)
        so = linear(ctx, _p(W, a + "output.dense.weight"), _p(W, a + "output.dense.bias"))
        a_out, a_hat, a_rstd = layer_norm(so + h, _p(W, a + "output.LayerNorm.weight"),
                                          _p(W, a + "output.LayerNorm.bias"), cfg.txt_ln_eps)
        c.update(q=q, k=k, v=v, Ps=Ps, a_hat=a_hat, a_rstd=a_rstd)
        # cross attention (image mask is all ones -> additive 0)
        x = b + "crossattention."
        qc = _heads(linear(a_out, _p(W, x + "self.query.weight"), _p(W, x + "self.query.bias")), nh)
        kc = _heads(linear(image_embeds, _p(W, x + "self.key.weight"), _p(W, x + "self.key.bias")), nh)
        vc = _heads(linear(image_embeds, _p(W, x + "self.value.weight"), _p(W, x + "self.value.bias")), nh)
        Pc = softmax((qc @ kc.transpose(0, 1, 3, 2)) * inv_sqrt)
        ctxc = _merge(Pc @ vc)
        co = linear(ctxc, _p(W, x + "output.dense.weight"), _p(W, x + "output.dense.bias"))
        c_out, c_hat, c_rstd = layer_norm(co + a_out, _p(W, x + "output.LayerNorm.weight"),
                                          _p(W, x + "output.LayerNorm.bias"), cfg.txt_ln_eps)
        c.update(kc=kc, vc=vc, Pc=Pc, c_hat=c_hat, c_rstd=c_rstd)
        # FFN
        u = linear(c_out, _p(W, b + "intermediate.dense.weight"), _p(W, b + "intermediate.dense.bias"))
        o = linear(gelu(u), _p(W, b + "output.dense.weight"), _p(W, b + "output.dense.bias"))
        h, o_hat, o_rstd = layer_norm(o + c_out, _p(W, b + "output.LayerNorm.weight"),
                                      _p(W, b + "output.LayerNorm.bias"), cfg.txt_ln_eps)
        c.update(u=u, o_hat=o_hat, o_rstd=o_rstd)
        caches.append(c)
    logits = linear(h[:, 0, :], _p(W, "itm_head.weight"), _p(W, "itm_head.bias"))
    return logits, caches


def _softmax_bwd(P, dP):
    return (P * (dP - (dP * P).sum(axis=-1, keepdims=True, dtype=F32))).astype(F32)


def xattn_grads(W, cfg, caches, layer_min=0):
    """Analytic d(sum_b logits[b,1]) / d(cross-attention probs) for text layers
    layer_min..txt_layers-1 (what `attention_probs.register_hook(save_attn_gradients)`
    records, B/med.py:280-283 + B/blip_image_text_matching.py:399-404).
    Returns {layer: dP (B,heads,L,N)}."""
    t = "text_encoder."
    nh = cfg.txt_heads
    dh = cfg.txt_hidden // nh
    inv_sqrt = F32(1.0 / math.sqrt(dh))
    B, L, Hd = caches[0]["h_in"].shape
    dh_out = np.zeros((B, L, Hd), dtype=F32)
    dh_out[:, 0, :] = _p(W, "itm_head.weight")[1]
    out = {}
    for i in range(cfg.txt_layers - 1, layer_min - 1, -1):
        b = f"{t}encoder.layer.{i}."
        c = caches[i]
        d_pre = layer_norm_bwd(dh_out, _p(W, b + "output.LayerNorm.weight"), c["o_hat"], c["o_rstd"])
        dg = d_pre @ _p(W, b + "output.dense.weight")
        du = dg * gelu_grad(c["u"])
        dc = d_pre + du @ _p(W, b + "intermediate.dense.weight")
        x = b + "crossattention."
        d_cpre = layer_norm_bwd(dc.astype(F32), _p(W, x + "output.LayerNorm.weight"), c["c_hat"], c["c_rstd"])
        dctxc = _heads((d_cpre @ _p(W, x + "output.dense.weight")).astype(F32), nh)
        dPc = (dctxc @ c["vc"].transpose(0, 1, 3, 2)).astype(F32)
        out[i] = dPc
        if i == layer_min:
            break
        dSc = _softmax_bwd(c["Pc"], dPc)
        dqc = _merge((dSc @ c["kc"]) * inv_sqrt)
        da = d_cpre + dqc @ _p(W, x + "self.query.weight")
        a = b + "attention."
        d_apre = layer_norm_bwd(da.astype(F32), _p(W, a + "output.LayerNorm.weight"), c["a_hat"], c["a_rstd"])
        dctx = _heads((d_apre @ _p(W, a + "output.dense.weight")).astype(F32), nh)
        dPs = dctx @ c["v"].transpose(0, 1, 3, 2)
        dv = c["Ps"].transpose(0, 1, 3, 2) @ dctx
        dSs = _softmax_bwd(c["Ps"], dPs)
        dq = (dSs @ c["k"]) * inv_sqrt
        dk = (dSs.transpose(0, 1, 3, 2) @ c["q"]) * inv_sqrt
        dh_out = (d_apre + _merge(dq) @ _p(W, a + "self.query.weight")
                  + _merge(dk) @ _p(W, a + "self.key.weight")
                  + _merge(dv) @ _p(W, a + "self.value.weight")).astype(F32)
    return out


def gradcam_from(P, dP, mask_L, grid):
    """B/blip_image_text_matching.py:427-433 for one layer: returns (B,heads,L-1,grid,grid)."""
    B, nh, L, N = P.shape
    g = P[:, :, :, 1:].reshape(B, nh, L, grid, grid) * \
        np.maximum(dP[:, :, :, 1:], 0).reshape(B, nh, L, grid, grid) * \
        mask_L.astype(F32)[:, None, :, None, None]
    g = np.where(g < 0, F32(0), g).astype(F32)
    return g[:, :, 1:]


def compute_gradcam(W, cfg, imgs, ids500, mask500, layers=None):
    """Restatement of compute_gradcam_ensemble (B/blip_image_text_matching.py:386-457).
    ids500/mask500: the caller's max_length-padded tokenisation; the forward uses the batch's
    longest caption length L (padding="longest", :230-236).
    Returns ({layer: gradcam (B,heads,L-1,P,P)}, logits, {layer: (P, dP)})."""
    L = int(mask500.sum(axis=1).max())
    ids = ids500[:, :L]
    m = mask500[:, :L]
    emb = vit_forward(W, cfg, imgs)
    logits, caches = text_forward(W, cfg, ids, m, emb)
    if layers is None:
        layers = list(range(cfg.txt_layers))
    dps = xattn_grads(W, cfg, caches, layer_min=min(layers))
    out, raw = {}, {}
    for l in layers:
        out[l] = gradcam_from(caches[l]["Pc"], dps[l], m, cfg.grid)
        raw[l] = (caches[l]["Pc"], dps[l])
    return out, logits, raw


# ------------------------------------------------------------------------------------- weights in

def _cubic_taps(in_size, out_size):
    """torch upsample_bicubic2d, align_corners=False: source x = (dst + 0.5) * in/out - 0.5 (not clamped), taps at
    floor(x) - 1 .. floor(x) + 2 clamped to the border, cubic-convolution weights with A = -0.75."""
    A = -0.75
    scale = in_size / out_size
    x = (np.arange(out_size) + 0.5) * scale - 0.5
    ix = np.floor(x).astype(np.int64)
    t = (x - ix).astype(np.float64)

    def c1(v):
        return ((A + 2) * v - (A + 3)) * v * v + 1

    def c2(v):
        return ((A * v - 5 * A) * v + 8 * A) * v - 4 * A
    w = np.stack([c2(t + 1), c1(t), c1(1 - t), c2(2 - t)], axis=1)
    idx = np.clip(ix[:, None] + np.arange(-1, 3)[None, :], 0, in_size - 1)
    return idx, w


def interpolate_pos_embed(pos, new_grid):
    """B/base_model.py:44-73: checkpoint pos_embed (1, 1 + g*g, D) -> (1, 1 + new_grid^2, D); the class token is kept,
    the g x g grid is resized bicubically (torch F.interpolate(mode="bicubic", align_corners=False))."""
    pos = np.asarray(pos, dtype=F32)
    D = pos.shape[-1]
    g = int(round((pos.shape[-2] - 1) ** 0.5))
    if g == new_grid:
        return pos
    grid = pos[0, 1:].reshape(g, g, D).astype(np.float64)
    iy, wy = _cubic_taps(g, new_grid)
    tmp = np.einsum("oi,oixd->oxd", wy, grid[iy])             # rows
    out = np.einsum("pj,opjd->opd", wy, tmp[:, iy])           # columns (square: same taps)
    return np.concatenate([pos[:, :1], out.reshape(1, new_grid * new_grid, D).astype(F32)], axis=1)
