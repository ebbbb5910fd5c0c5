"""Seeded synthetic weights and inputs (there is no checkpoint / dataset on the GPU box).

Parameter names follow the reference's state-dict keys (module attribute paths of
`Files to replace for BLIP/blip_image_text_matching.py`:39-57, `vit.py`:176-258, `med.py`:56-524)
so the same dict loads into the reference modules (golden generation), the numpy oracle and the
HIP engine.  Every tensor is drawn from its own generator seeded by (seed, crc32(name)) so any
subset can be regenerated independently and identically anywhere.
"""
import zlib
from collections import OrderedDict

import numpy as np

from .config import ModelCfg


def param_shapes(cfg: ModelCfg) -> "OrderedDict[str, tuple]":
    D, H, I = cfg.vit_dim, cfg.txt_hidden, cfg.txt_inter
    s = OrderedDict()
    v = "visual_encoder."
    s[v + "cls_token"] = (1, 1, D)
    s[v + "pos_embed"] = (1, cfg.n_img_tokens, D)
    s[v + "patch_embed.proj.weight"] = (D, 3, cfg.patch, cfg.patch)
    s[v + "patch_embed.proj.bias"] = (D,)
    for i in range(cfg.vit_depth):
        b = f"{v}blocks.{i}."
        s[b + "norm1.weight"] = (D,)
        s[b + "norm1.bias"] = (D,)
        s[b + "attn.qkv.weight"] = (3 * D, D)
        s[b + "attn.qkv.bias"] = (3 * D,)
        s[b + "attn.proj.weight"] = (D, D)
        s[b + "attn.proj.bias"] = (D,)
        s[b + "norm2.weight"] = (D,)
        s[b + "norm2.bias"] = (D,)
        s[b + "mlp.fc1.weight"] = (cfg.vit_mlp_ratio * D, D)
        s[b + "mlp.fc1.bias"] = (cfg.vit_mlp_ratio * D,)
        s[b + "mlp.fc2.weight"] = (D, cfg.vit_mlp_ratio * D)
        s[b + "mlp.fc2.bias"] = (D,)
    s[v + "norm.weight"] = (D,)
    s[v + "norm.bias"] = (D,)
    t = "text_encoder."
    s[t + "embeddings.word_embeddings.weight"] = (cfg.vocab, H)
    s[t + "embeddings.position_embeddings.weight"] = (cfg.max_pos, H)
    s[t + "embeddings.LayerNorm.weight"] = (H,)
    s[t + "embeddings.LayerNorm.bias"] = (H,)
    for i in range(cfg.txt_layers):
        b = f"{t}encoder.layer.{i}."
        for att, kw in (("attention", H), ("crossattention", D)):
            s[b + att + ".self.query.weight"] = (H, H)
            s[b + att + ".self.query.bias"] = (H,)
            s[b + att + ".self.key.weight"] = (H, kw)
            s[b + att + ".self.key.bias"] = (H,)
            s[b + att + ".self.value.weight"] = (H, kw)
            s[b + att + ".self.value.bias"] = (H,)
            s[b + att + ".output.dense.weight"] = (H, H)
            s[b + att + ".output.dense.bias"] = (H,)
            s[b + att + ".output.LayerNorm.weight"] = (H,)
            s[b + att + ".output.LayerNorm.bias"] = (H,)
        s[b + "intermediate.dense.weight"] = (I, H)
        s[b + "intermediate.dense.bias"] = (I,)
        s[b + "output.dense.weight"] = (H, I)
        s[b + "output.dense.bias"] = (H,)
        s[b + "output.LayerNorm.weight"] = (H,)
        s[b + "output.LayerNorm.bias"] = (H,)
    s["itm_head.weight"] = (2, H)
    s["itm_head.bias"] = (2,)
    return s


def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.default_rng([seed, zlib.crc32(name.encode())])


def synth_tensor(name: str, shape, seed: int) -> np.ndarray:
    """One synthetic parameter.  Scales: weights N(0, 0.02) like the reference's own init
    (vit.py:262-269 trunc_normal 0.02; BERT normal 0.02) except attention q/k projections which are
    widened so softmaxes are not flat; LayerNorm gains 1+N(0,0.1); every bias N(0,0.02)."""
    g = _rng(seed, name)
    x = g.standard_normal(size=shape, dtype=np.float32)
    if name.endswith("LayerNorm.weight") or name.endswith("norm1.weight") or \
            name.endswith("norm2.weight") or name.endswith("norm.weight"):
        return (1.0 + 0.1 * x).astype(np.float32)
    if name.endswith(".bias"):
        return (0.02 * x).astype(np.float32)
    if name.endswith("query.weight") or name.endswith("key.weight"):
        return (0.06 * x).astype(np.float32)
    if name.endswith("attn.qkv.weight"):
        x *= 0.02
        d = shape[1]
        x[: 2 * d] *= 3.0          # q and k rows of the fused ViT qkv
        return x.astype(np.float32)
    if name == "itm_head.weight":
        return (0.5 * x).astype(np.float32)
    return (0.02 * x).astype(np.float32)


def synth_state_dict(cfg: ModelCfg, seed: int = 0, names=None) -> "OrderedDict[str, np.ndarray]":
    shapes = param_shapes(cfg)
    out = OrderedDict()
    for n, shp in shapes.items():
        if names is not None and n not in names:
            continue
        out[n] = synth_tensor(n, shp, seed)
    return out


def synth_checkpoint(cfg_model: ModelCfg, cfg_ckpt: ModelCfg, seed: int = 7) -> "OrderedDict[str, np.ndarray]":
    """A synthetic BLIP checkpoint state dict fine-tuned at another resolution (pos_embed of cfg_ckpt's grid, like the
    384-px flickr checkpoint of blip_itm_large.yaml:10), with the extra heads a real one carries and ONE
    shape-mismatched key (itm_head.bias) that load_checkpoint must drop (base_model.py:116-119)."""
    sd = synth_state_dict(cfg_ckpt, seed)
    sd["vision_proj.weight"] = np.zeros((8, cfg_model.vit_dim), dtype=np.float32)
    sd["text_proj.weight"] = np.zeros((8, cfg_model.txt_hidden), dtype=np.float32)
    sd["itm_head.bias"] = np.zeros((3,), dtype=np.float32)
    return sd


# ----------------------------------------------------------------------------- inputs

CLIP_MEAN = np.array([0.48145466, 0.4578275, 0.40821073], dtype=np.float32)   # Dataset.py:434-443
CLIP_STD = np.array([0.26862954, 0.26130258, 0.27577711], dtype=np.float32)


def synth_images(batch: int, size: int, seed: int = 1234, block: int = 8, noise: int = 12):
    """Seeded block-smoothed uint8 RGB noise: 8x8 blocks of uniform colour plus
    +-`noise` levels of per-pixel noise (sensor-noise stand-in; it sets how many bilateral lattice
    points an image occupies: ~3.6 / pixel at 12, ~0.9 / pixel at 4).  Returns
    (rgb uint8 (B,H,W,3) used by the CRF bilateral term, normalised fp32 (B,3,H,W) network input)."""
    g = np.random.default_rng([seed, 7])
    nb = (size + block - 1) // block
    coarse = g.integers(0, 256, size=(batch, nb, nb, 3), dtype=np.int64)
    fine = g.integers(-noise, noise + 1, size=(batch, size, size, 3), dtype=np.int64)
    rgb = np.repeat(np.repeat(coarse, block, axis=1), block, axis=2)[:, :size, :size]
    rgb = np.clip(rgb + fine, 0, 255).astype(np.uint8)
    x = rgb.astype(np.float32) / np.float32(255.0)
    x = (x - CLIP_MEAN) / CLIP_STD
    return rgb, np.ascontiguousarray(x.transpose(0, 3, 1, 2)).astype(np.float32)


def synth_photo_images(batch: int, size: int, seed: int = 1234, sigma: float = 3.0):
    """Photograph-like seeded RGB images (the third DenseCRF operating point of bench.py): a few large regions with soft
    edges (objects), smooth illumination gradients, band-limited texture of region-dependent strength and per-pixel sensor
    noise of `sigma` grey levels -- what decides how many bilateral lattice points (cells of 50 px x 5 grey levels per
    channel) an image occupies.  The block images of synth_images() are the two ends: flat 8 x 8 blocks + noise 4 -> 0.9
    points per pixel, + noise 12 -> 3.6; this generator lands in between (bench.py reports the measured density).
    Same return convention as synth_images()."""
    g = np.random.default_rng([seed, 23])
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32)
    out = np.empty((batch, size, size, 3), dtype=np.float32)
    for b in range(batch):
        # 5 soft-edged regions (Voronoi cells of random sites, blended over ~6 px), each with its own colour and texture gain
        sites = g.uniform(0, size, size=(5, 2)).astype(np.float32)
        d2 = (yy[None] - sites[:, 0, None, None]) ** 2 + (xx[None] - sites[:, 1, None, None]) ** 2
        w = np.exp(-(np.sqrt(d2) - np.sqrt(d2.min(axis=0, keepdims=True))) / 6.0)
        w /= w.sum(axis=0, keepdims=True)
        colour = g.uniform(40, 215, size=(5, 3)).astype(np.float32)
        gain = g.uniform(0.0, 14.0, size=5).astype(np.float32)
        img = np.einsum("rhw,rc->hwc", w, colour)
        # illumination: two low-frequency gradients shared by the channels
        illum = 18 * np.sin(xx / g.uniform(60, 140) + g.uniform(0, 6.28)) + 14 * np.cos(yy / g.uniform(60, 140) + g.uniform(0, 6.28))
        # texture: sum of 6 mid-frequency waves, scaled per region
        tex = np.zeros((size, size), np.float32)
        for _ in range(6):
            fx, fy = g.uniform(0.08, 0.5, size=2)
            tex += np.sin(fx * xx + fy * yy + g.uniform(0, 6.28)).astype(np.float32)
        tex *= np.einsum("rhw,r->hw", w, gain) / 3.0
        img = img + (illum + tex)[..., None] + g.normal(0, sigma, size=(size, size, 3)).astype(np.float32)
        out[b] = img
    rgb = np.clip(np.rint(out), 0, 255).astype(np.uint8)
    x = rgb.astype(np.float32) / np.float32(255.0)
    x = (x - CLIP_MEAN) / CLIP_STD
    return rgb, np.ascontiguousarray(x.transpose(0, 3, 1, 2)).astype(np.float32)


def synth_tokens(cfg: ModelCfg, n_classes_per_image, seed: int = 1234, max_length: int = 500):
    """Synthetic token ids `[CLS] a picture of t1..tC [SEP]` padded to `max_length`
    (the caller's `padding="max_length", max_length=500` tokenisation at
    PnP_OVSS_0514_updated_segmentation.py:317) with one word-piece per class."""
    g = np.random.default_rng([seed, 11])
    B = len(n_classes_per_image)
    ids = np.zeros((B, max_length), dtype=np.int64)
    lo = 110 if cfg.vocab > 400 else 3
    for b, c in enumerate(n_classes_per_image):
        toks = [101, lo + 1, lo + 2, lo + 3] + list(g.integers(lo + 10, cfg.vocab - 2, size=c)) + [cfg.sep_token_id]
        ids[b, : len(toks)] = toks
    mask = (ids != cfg.pad_token_id).astype(np.int64)
    return ids, mask


def inject_outliers(W, cfg, gain, seed=99, n=6, jitter=0.0, compensate=True):
    """Outlier channels of the kind trained ViT-L checkpoints carry, as a FUNCTION-PRESERVING re-parametrisation: per ViT block
    `n` channels of each LayerNorm (gain and bias) are scaled by `gain` and the matching input columns of the consuming Linear
    (qkv / fc1) by 1 / gain; `n` value channels of qkv (rows + bias) by `gain` and the matching proj columns by 1 / gain; `n`
    channels of the final LayerNorm by `gain` and the matching columns of all 12 cross-attention key / value projections by
    1 / gain.  In exact arithmetic the model is unchanged; element-wise-relative arithmetic (fp32, split bf16) sees the same
    relative errors; per-row-scaled integer slices lose log2(gain) bits on every other channel of the row.  (The GELU between
    fc1 and fc2 does not commute with a scale, so fc2's input carries no injected outliers.)

    With a power-of-two `gain` the scaled model is BIT-identical in any binary floating-point arithmetic (scaling by 2^k is
    exact), which makes it a control, not a stress.  `jitter` > 0 draws a separate gain per channel, log-uniform in
    [gain / (1 + jitter), gain * (1 + jitter)]: 1 / g is then rounded, every product and partial sum rounds differently from
    the plain model's, and the two arithmetics under comparison part ways where they can.  `compensate=False` leaves the
    consuming weights alone: the outlier channels then dominate every dot product they enter (a DIFFERENT model, with the
    massive activations of a trained checkpoint), so the 2^-16 relative error of a split-bf16 product on the large terms is
    an absolute error that the small, informative terms have to live with."""
    g = np.random.default_rng(seed)
    W = {k: v for k, v in W.items()}
    D = cfg.vit_dim

    def gains():
        if jitter > 0:
            return (np.float32(gain) * np.exp(g.uniform(-np.log1p(jitter), np.log1p(jitter), size=n))).astype(np.float32)
        return np.full(n, gain, np.float32)

    def scale(name, ch, f, axis):
        w = W[name].copy()
        if axis == 0:
            w[ch] = w[ch] * (f if w.ndim == 1 else f[:, None])
        else:
            w[:, ch] = w[:, ch] * f[None, :]
        W[name] = w

    for i in range(cfg.vit_depth):
        b = f"visual_encoder.blocks.{i}."
        for nm, cons in (("norm1", "attn.qkv.weight"), ("norm2", "mlp.fc1.weight")):
            ch = g.choice(D, n, replace=False)
            f = gains()
            for part in (".weight", ".bias"):
                scale(b + nm + part, ch, f, 0)
            if compensate:
                scale(b + cons, ch, np.float32(1) / f, 1)
        ch = g.choice(D, n, replace=False)
        f = gains()
        scale(b + "attn.qkv.weight", 2 * D + ch, f, 0)
        scale(b + "attn.qkv.bias", 2 * D + ch, f, 0)
        if compensate:
            scale(b + "attn.proj.weight", ch, np.float32(1) / f, 1)
    ch = g.choice(D, n, replace=False)
    f = gains()
    for part in (".weight", ".bias"):
        scale("visual_encoder.norm" + part, ch, f, 0)
    if compensate:
        for i in range(cfg.txt_layers):
            for kv in ("key", "value"):
                scale(f"text_encoder.encoder.layer.{i}.crossattention.self.{kv}.weight", ch, np.float32(1) / f, 1)
    return W
