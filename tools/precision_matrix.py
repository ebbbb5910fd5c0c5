"""Go / no-go matrix for cutting the 3x MFMA issue multiplier of the parity mode (task 1, step A).

The benchmarked mode ("bf16x3") forms every dense product of the ViT from (hi, lo) bf16 operand pairs with THREE bf16 MFMAs
(a_hi.b_hi + a_hi.b_lo + a_lo.b_hi, fp32 accumulate).  This tool emulates cheaper arithmetic for the five GEMM families of a
drop iteration -- qkv, proj, fc1, fc2 (B/vit.py:45-51, 93-117) and the cross-attention K/V projections (B/med.py:208-211) -- in
torch (float64 where exactness of an integer product matters) and reports, against the exact-fp32 run of the same inputs:
  * picks kept: (image, iteration) pairs whose cumulative top-10 patch sets equal the fp32 run's (PnP.py:638-647);
    on the reference's own fixture (tests/golden/droploop_large.npz, 6 pairs) and on a B-image batch of the bench's generator;
  * the per-token min-max normalised error of the aggregated map (what the threshold of PnP.py:350-355 sees);
  * label pixels that differ from the fp32 run's after threshold / upsample / argmax (postprocess "none": PnP.py:348-379).

Candidates (cost in bf16-MFMA equivalents per product; int8 MFMAs run at twice the bf16 rate on gfx950):
  x3            3.0  the product today
  x2w8 / x2a8   2.0  one cross term dropped: a.b_hi (weights at 8 bits) / a_hi.b (activations at 8 bits), per family or everywhere
  i8_2x2_3      1.5  Ozaki-style int8 slicing, per-row scales, 2 slices each (15 bits + sign of the row maximum), lo.lo dropped
  i8_2x2_4      2.0  the same with all four slice products
  i8_3x2_5      2.5  3 activation slices x 2 weight slices, the five products with i + j <= 2 ... (2,1) dropped
  i8_3x3_6      3.0  3 x 3 slices, i + j <= 2: the int8 form with no saving (calibration row)
  h<...>             the same after an orthonormal Hadamard rotation of both operands along K (spreads outlier channels)
Weights: the seeded synthetic state dict, optionally with OUTLIER channels injected (--outliers GAIN) as a function-preserving
re-parametrisation (inject_outliers) -- real ViT-L checkpoints carry such channels, seeded Gaussians do not.  `f64` is a
calibration row: the same model with every product in float64.

Runs on the GPU box through torch (fp64 GEMMs; seconds per candidate at B = 35) or on the CPU at small B (minutes per candidate).
Test infrastructure: never imported by the product.

usage: python tools/precision_matrix.py [--batch 35] [--outliers 16] [--cands x3,i8_2x2_3,...] [--families all|qkv,proj,...] [--out f.json]
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pnp-ovss_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch

from pnp_ovss import config as C, synth
from oracle import pipeline_np as OP

DEV = "cuda" if torch.cuda.is_available() else "cpu"
torch.backends.cuda.matmul.allow_tf32 = False
F32, F64 = torch.float32, torch.float64
FAMILIES = ("qkv", "proj", "fc1", "fc2", "crosskv")


def bf16r(x):
    return x.to(torch.bfloat16).to(F32)


def split(x):
    hi = bf16r(x)
    return hi, bf16r(x - hi)


def split16(x):
    """(hi, lo) fp16 pair of an fp32 tensor, as fp32 values: hi = fp16(x) (11 significant bits), lo = fp16(x - hi) (11 more where the
    exponent range allows: fp16 subnormals below 6.1e-5 carry fewer bits, values below 6e-8 vanish -- torch's conversion keeps
    subnormals, as v_mfma_f32_16x16x32_f16 does)."""
    hi = x.to(torch.float16).to(F32)
    return hi, (x - hi).to(torch.float16).to(F32)


def mm_x3(a, bt):
    """a (.., M, K) @ bt (.., K, N), both split; small terms first like the kernels."""
    ah, al = split(a)
    bh, bl = split(bt)
    return (ah @ bl + al @ bh) + ah @ bh


_HAD = {}


def hadamard(k):
    if k not in _HAD:
        h = torch.ones(1, 1, dtype=F64, device=DEV)
        while h.shape[0] < k:
            h = torch.cat([torch.cat([h, h], 1), torch.cat([h, -h], 1)], 0)
        assert h.shape[0] == k, "K must be a power of two for the rotation"
        _HAD[k] = h / math.sqrt(k)
    return _HAD[k]


def int_slices(x, p):
    """x (R, K) float64 -> (slices [p] of integer-valued float64 in [-128, 128], step (R, 1)): x ~ step * sum_i 256^(p-1-i) s_i.
    Signed digits: the leading slice holds 7 bits + sign of the row maximum, every further slice 8 more bits.  (A digit of
    +128 -- one value past int8 -- can appear below the leading slice when the remainder rounds up; a kernel would clamp it,
    an error of one step of that digit on < 0.4 % of the elements; the emulation keeps it.)"""
    amax = x.abs().amax(dim=1, keepdim=True).clamp_min(1e-300)
    Q = float(sum(127 * 256 ** i for i in range(p)))
    step = amax / Q
    rem = torch.round(x / step)
    out = []
    for i in range(p):
        w = float(256 ** (p - 1 - i))
        s = torch.floor(rem / w + 0.5)
        rem = rem - s * w
        out.append(s)
    return out, step


_TERMS = {"2x2_3": (2, 2, [(0, 0), (0, 1), (1, 0)]), "2x2_4": (2, 2, [(0, 0), (0, 1), (1, 0), (1, 1)]),
          "3x2_5": (3, 2, [(0, 0), (0, 1), (1, 0), (1, 1), (2, 0)]),
          "3x3_6": (3, 3, [(0, 0), (0, 1), (1, 0), (1, 1), (2, 0), (0, 2)])}
COST = {"f32": 16.0, "f64": None, "x3": 3.0, "x2w8": 2.0, "x2a8": 2.0, "h1": 1.0, "h2w11": 2.0, "h2a11": 2.0, "h3": 3.0, "i8_2x2_3": 1.5, "i8_2x2_4": 2.0, "i8_3x2_5": 2.5, "i8_3x3_6": 3.0}


class Gemm:
    """y = a @ w.T in an emulated arithmetic; the weight-side preparation is cached per weight tensor."""

    def __init__(self):
        self.cache = {}

    def _w(self, w, key, fn):
        k = (w.data_ptr(), key)
        if k not in self.cache:
            self.cache[k] = fn(w)
        return self.cache[k]

    def __call__(self, mode, a, w):
        shp = a.shape[:-1]
        a2 = a.reshape(-1, a.shape[-1])
        y = self._mm(mode, a2, w)
        return y.reshape(*shp, w.shape[0])

    def _mm(self, mode, a, w):
        if mode == "f32":
            return a @ w.T
        if mode == "f64":
            return (a.to(F64) @ w.to(F64).T).to(F32)
        if mode in ("x3", "x2w8", "x2a8"):
            ah, al = split(a)
            bh, bl = self._w(w, "split", lambda t: tuple(s.T.contiguous() for s in split(t)))
            if mode == "x3":
                return (ah @ bl + al @ bh) + ah @ bh
            if mode == "x2w8":
                return al @ bh + ah @ bh
            return ah @ bl + ah @ bh
        if mode in ("h1", "h2w11", "h2a11", "h3"):
            # fp16 operands (the fp16 MFMA issues at the bf16 rate): h1 = hi x hi, both operands at 11 bits (one MFMA per product);
            # h2w11 = activations (hi, lo) x weights hi (weights at 11 bits); h2a11 = activations hi x weights (hi, lo); h3 = three
            # terms like the split-bf16 form (22-bit operands; no saving -- calibration)
            ah, al = split16(a)
            bh, bl = self._w(w, "split16", lambda t: tuple(s.T.contiguous() for s in split16(t)))
            if mode == "h1":
                return ah @ bh
            if mode == "h2w11":
                return al @ bh + ah @ bh
            if mode == "h2a11":
                return ah @ bl + ah @ bh
            return (ah @ bl + al @ bh) + ah @ bh
        rot = mode.startswith("hi8")
        pa, pb, terms = _TERMS[mode[4:] if rot else mode[3:]]
        K = a.shape[1]
        ad = a.to(F64)
        if rot:
            ad = ad @ hadamard(K)

        def prep_w(t):
            td = t.to(F64)
            if rot:
                td = td @ hadamard(K)
            sl, st = int_slices(td, pb)
            return [s.T.contiguous() for s in sl], st.T.contiguous()
        bs, bstep = self._w(w, ("i8", pb, rot), prep_w)
        as_, astep = int_slices(ad, pa)
        acc = None
        for (i, j) in terms:
            t = (as_[i] @ bs[j]) * float(256 ** (pa - 1 - i) * 256 ** (pb - 1 - j))
            acc = t if acc is None else acc + t
        return (acc * astep * bstep).to(F32)


def layer_norm(x, w, b, eps):
    return torch.nn.functional.layer_norm(x, (x.shape[-1],), w, b, eps)


def gelu(x):
    return torch.nn.functional.gelu(x)


class Model:
    """BLIP-ITM forward + dL/dP of one text layer in torch, every dense product through `Gemm` in the family's mode.
    Restates oracle/blip_itm_np.py (B/vit.py:274-290, B/med.py:191-311, B/blip_image_text_matching.py:217-249, 399-435)."""

    def __init__(self, Wnp, cfg, modes, att_mode, text_mode):
        self.W = {k: torch.from_numpy(np.ascontiguousarray(v)).to(DEV) for k, v in Wnp.items()}
        self.cfg, self.modes, self.att_mode, self.text_mode = cfg, modes, att_mode, text_mode
        self.g = Gemm()

    def lin(self, fam, x, wname, bname):
        y = self.g(self.modes.get(fam, "x3" if self.att_mode != "f32" else "f32"), x, self.W[wname])
        return y + self.W[bname]

    def att_mm(self, a, b):
        if self.att_mode == "f64":
            return (a.to(F64) @ b.to(F64)).to(F32)
        return a @ b if self.att_mode == "f32" else mm_x3(a, b)

    def vit(self, img):
        W, cfg, v = self.W, self.cfg, "visual_encoder."
        B = img.shape[0]
        P, D, H = cfg.grid, cfg.vit_dim, cfg.vit_heads
        dh, ps = D // H, cfg.patch
        x = img.reshape(B, 3, P, ps, P, ps).permute(0, 2, 4, 1, 3, 5).reshape(B, P * P, 3 * ps * ps)
        wp = W[v + "patch_embed.proj.weight"].reshape(D, -1)
        x = (x @ wp.T if self.att_mode in ("f32", "f64") else mm_x3(x, wp.T)) + W[v + "patch_embed.proj.bias"]
        x = torch.cat([W[v + "cls_token"].expand(B, 1, D), x], 1) + W[v + "pos_embed"][:, : P * P + 1]
        N = x.shape[1]
        scale = dh ** -0.5
        for i in range(cfg.vit_depth):
            b = f"{v}blocks.{i}."
            h = layer_norm(x, W[b + "norm1.weight"], W[b + "norm1.bias"], cfg.vit_ln_eps)
            qkv = self.lin("qkv", h, b + "attn.qkv.weight", b + "attn.qkv.bias").reshape(B, N, 3, H, dh).permute(2, 0, 3, 1, 4)
            att = torch.softmax(self.att_mm(qkv[0], qkv[1].transpose(-1, -2)) * scale, dim=-1)
            ctx = self.att_mm(att, qkv[2]).permute(0, 2, 1, 3).reshape(B, N, D)
            x = x + self.lin("proj", ctx, b + "attn.proj.weight", b + "attn.proj.bias")
            h = layer_norm(x, W[b + "norm2.weight"], W[b + "norm2.bias"], cfg.vit_ln_eps)
            h = gelu(self.lin("fc1", h, b + "mlp.fc1.weight", b + "mlp.fc1.bias"))
            x = x + self.lin("fc2", h, b + "mlp.fc2.weight", b + "mlp.fc2.bias")
        return layer_norm(x, W[v + "norm.weight"], W[v + "norm.bias"], cfg.vit_ln_eps)

    def tlin(self, x, wname, bname):
        w, b = self.W[wname], self.W[bname]
        return (x @ w.T if self.text_mode == "f32" else _X3Linear.apply(x, w)) + b

    def gradcam(self, img, ids, mask, layer, head):
        """-> (B, L-1, P, P) = gradcam_blocklist[layer][head] of the reference, logits (B, 2)."""
        W, cfg, t = self.W, self.cfg, "text_encoder."
        with torch.no_grad():
            emb_img = self.vit(img)
        B, L = ids.shape
        nh = cfg.txt_heads
        dh = cfg.txt_hidden // nh
        ids = ids.clone()
        ids[:, 0] = cfg.enc_token_id
        emb = W[t + "embeddings.word_embeddings.weight"][ids] + W[t + "embeddings.position_embeddings.weight"][None, :L]
        h = layer_norm(emb, W[t + "embeddings.LayerNorm.weight"], W[t + "embeddings.LayerNorm.bias"], cfg.txt_ln_eps)
        ext = ((1.0 - mask.to(F32)) * -10000.0)[:, None, None, :]
        inv = 1.0 / math.sqrt(dh)

        def heads(x):
            return x.reshape(x.shape[0], x.shape[1], nh, dh).permute(0, 2, 1, 3)

        def merge(x):
            return x.permute(0, 2, 1, 3).reshape(x.shape[0], x.shape[2], nh * dh)
        Pkeep = None
        with torch.enable_grad():
            for i in range(cfg.txt_layers):
                b = f"{t}encoder.layer.{i}."
                a = b + "attention."
                q = heads(self.tlin(h, a + "self.query.weight", a + "self.query.bias"))
                k = heads(self.tlin(h, a + "self.key.weight", a + "self.key.bias"))
                vv = heads(self.tlin(h, a + "self.value.weight", a + "self.value.bias"))
                Ps = torch.softmax((q @ k.transpose(-1, -2)) * inv + ext, dim=-1)
                so = self.tlin(merge(Ps @ vv), a + "output.dense.weight", a + "output.dense.bias")
                a_out = layer_norm(so + h, W[a + "output.LayerNorm.weight"], W[a + "output.LayerNorm.bias"], cfg.txt_ln_eps)
                x = b + "crossattention."
                qc = heads(self.tlin(a_out, x + "self.query.weight", x + "self.query.bias"))
                kc = heads(self.lin("crosskv", emb_img, x + "self.key.weight", x + "self.key.bias"))
                vc = heads(self.lin("crosskv", emb_img, x + "self.value.weight", x + "self.value.bias"))
                Pc = torch.softmax((qc @ kc.transpose(-1, -2)) * inv, dim=-1)
                if i == layer:
                    Pc = Pc.detach().requires_grad_(True)
                    Pkeep = Pc
                co = self.tlin(merge(Pc @ vc), x + "output.dense.weight", x + "output.dense.bias")
                c_out = layer_norm(co + a_out, W[x + "output.LayerNorm.weight"], W[x + "output.LayerNorm.bias"], cfg.txt_ln_eps)
                u = self.tlin(c_out, b + "intermediate.dense.weight", b + "intermediate.dense.bias")
                o = self.tlin(gelu(u), b + "output.dense.weight", b + "output.dense.bias")
                h = layer_norm(o + c_out, W[b + "output.LayerNorm.weight"], W[b + "output.LayerNorm.bias"], cfg.txt_ln_eps)
            logits = h[:, 0, :] @ W["itm_head.weight"].T + W["itm_head.bias"]
            logits[:, 1].sum().backward()
        dP = Pkeep.grad
        Pn = cfg.grid
        g = Pkeep.detach()[:, head, :, 1:] * torch.relu(dP[:, head, :, 1:]) * mask.to(F32)[:, :, None]
        return g[:, 1:].reshape(B, L - 1, Pn, Pn).contiguous(), logits.detach()


class _X3Linear(torch.autograd.Function):
    """Text-side Linear of the product's parity mode: split-bf16 forward, and the backward on the transposed weight."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(w)
        return mm_x3(x, w.T)

    @staticmethod
    def backward(ctx, dy):
        (w,) = ctx.saved_tensors
        return mm_x3(dy, w), None


def drop_loop(model, imgs, ids, mask, drop_iter, layer, head, grid, patch):
    """oracle/pipeline_np.py:drop_loop (PnP.py:564-722) on device tensors."""
    B = imgs.shape[0]
    x = imgs.clone()
    picks = [[] for _ in range(B)]
    preds, newp = [], []
    for it in range(drop_iter):
        for b in range(B):
            for p in picks[b]:
                r, c = (p // grid) * patch, (p % grid) * patch
                x[b, :, r:r + patch, c:c + patch] = 0
        g, _ = model.gradcam(x, ids, mask, layer, head)
        pred = g.clone()
        for b in range(B):
            for p in picks[b]:
                pred[b, :, p // grid, p % grid] = 0
        preds.append(pred)
        gn = g.cpu().numpy()
        cur = []
        for b in range(B):
            sal = gn[b, 3:-1].sum(axis=0, dtype=np.float32)
            sel = OP.select_topk(sal, picks[b], 10)
            picks[b].extend(sel)
            cur.append(sel)
        newp.append(cur)
    agg = preds[0].clone()
    for it in range(drop_iter):
        agg = agg + preds[it]
    return preds[0].cpu().numpy(), agg.cpu().numpy(), newp


inject_outliers = synth.inject_outliers          # the re-parametrisation lives with the other seeded generators


def nerr(a, b):
    """max over token rows of |minmax(a) - minmax(b)| (rows = leading dims, maps = last two)."""
    a = a.reshape(-1, a.shape[-2] * a.shape[-1]).astype(np.float64)
    b = b.reshape(-1, b.shape[-2] * b.shape[-1]).astype(np.float64)
    keep = (b.max(1) - b.min(1)) > 0
    a, b = a[keep], b[keep]
    na = (a - a.min(1, keepdims=True)) / np.maximum(a.max(1, keepdims=True) - a.min(1, keepdims=True), 1e-300)
    nb = (b - b.min(1, keepdims=True)) / (b.max(1, keepdims=True) - b.min(1, keepdims=True))
    return float(np.abs(na - nb).max())


def labels_none(agg, ncls):
    out = []
    for b, n in enumerate(ncls):
        m = OP.threshold_upsample(agg[b, 3:3 + n], 336, 336, 0.15, False, True)
        out.append(np.argmax(m, axis=0).astype(np.uint8))
    return np.stack(out)


def pick_pairs(ref, got, B):
    ok = 0
    for it in range(4):
        for b in range(B):
            ok += set(sum((ref[i][b] for i in range(it + 1)), [])) == set(sum((got[i][b] for i in range(it + 1)), []))
    return ok


def build_modes(cand, families):
    if cand == "f32":
        return {f: "f32" for f in FAMILIES}, "f32", "f32"
    if cand == "f64":                 # calibration: how far is an fp32 run from a higher-precision run of the same model
        return {f: "f64" for f in FAMILIES}, "f64", "f32"
    return {f: (cand if f in families else "x3") for f in FAMILIES}, "x3", "x3"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=35 if DEV == "cuda" else 2)
    ap.add_argument("--outliers", type=float, default=0.0, help="gain of the injected outlier channels (0: none)")
    ap.add_argument("--cands", default="x3,x2w8,x2a8,i8_2x2_3,i8_2x2_4,i8_3x2_5,i8_3x3_6,hi8_2x2_3,hi8_2x2_4,hi8_3x2_5")
    ap.add_argument("--families", default="all", help="'all', 'each' (one family at a time), or a comma list")
    ap.add_argument("--no-fixture", action="store_true")
    ap.add_argument("--label-images", type=int, default=8)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    cfg = C.blip_itm_large(336)
    W0 = synth.synth_state_dict(cfg, 0)
    if a.outliers:
        W0 = inject_outliers(W0, cfg, a.outliers)
    B = a.batch
    work = []
    if not a.no_fixture:
        g = np.load(os.path.join(ROOT, "tests", "golden", "droploop_large.npz"), allow_pickle=True)
        ncls = [int(x) for x in g["n_classes"]]
        _, im = synth.synth_images(2, 336, seed=int(g["image_seed"]))
        ids, mask = synth.synth_tokens(cfg, ncls, seed=int(g["token_seed"]))
        work.append(("fixture", im, ids, mask, ncls, g))
    _, im = synth.synth_images(B, 336, seed=515)
    ids, mask = synth.synth_tokens(cfg, [20] * B, seed=515)
    work.append((f"batch{B}", im, ids, mask, [20] * B, None))

    fam_sets = {"all": [FAMILIES]}.get(a.families)
    if a.families == "each":
        fam_sets = [(f,) for f in FAMILIES]
    elif fam_sets is None:
        fam_sets = [tuple(a.families.split(","))]
    runs = [("f32", FAMILIES)]
    for c in a.cands.split(","):
        if c in ("x3",):
            runs.append((c, FAMILIES))
        else:
            runs += [(c, fs) for fs in fam_sets]
    rows, ref = [], {}
    for cand, fams in runs:
        modes, att, txt = build_modes(cand, fams)
        model = Model(W0, cfg, modes, att, txt)
        row = {"cand": cand, "families": "all" if tuple(fams) == FAMILIES else ",".join(fams),
               "bf16_equiv": COST.get(cand.lstrip("h") if cand.startswith("hi8") else cand, None)}
        t0 = time.time()
        for (nm, im, ids, mask, ncls, gold) in work:
            L = int(mask.sum(1).max())
            g0, agg, picks = drop_loop(model, torch.from_numpy(im).to(DEV), torch.from_numpy(ids[:, :L]).to(DEV),
                                       torch.from_numpy(mask[:, :L]).to(DEV), 4, 7, 9, cfg.grid, cfg.patch)
            nb = len(ncls)
            lab = labels_none(agg[: a.label_images], ncls[: a.label_images])
            if cand == "f32":
                ref[nm] = (agg, picks, lab)
            r_agg, r_picks, r_lab = ref[nm]
            rec = {"pairs_kept": pick_pairs(r_picks, picks, nb), "pairs": 4 * nb,
                   "nerr_agg": max(nerr(agg[b, 3:3 + n], r_agg[b, 3:3 + n]) for b, n in enumerate(ncls)),
                   "max_abs_agg": float(np.abs(agg - r_agg).max()),
                   "label_flips": float((lab != r_lab).mean())}
            if gold is not None and not a.outliers:
                z = gold["zeroed"]
                kept = 0
                for it in range(1, 4):
                    for b in range(nb):
                        got = set(sum((picks[i][b] for i in range(it)), []))
                        kept += got == set(np.nonzero(z[it, b])[0].tolist())
                rec["reference_pairs_kept"] = kept
                rec["reference_pairs"] = 3 * nb
                rec["nerr_agg_vs_reference"] = max(nerr(agg[b, 3:3 + n], gold["agg"][b, 3:3 + n]) for b, n in enumerate(ncls))
            row[nm] = rec
        row["seconds"] = round(time.time() - t0, 1)
        rows.append(row)
        print(json.dumps(row), flush=True)
        del model
        if DEV == "cuda":
            torch.cuda.empty_cache()
    if a.out:
        with open(a.out, "w") as f:
            json.dump({"outliers": a.outliers, "batch": B, "device": DEV, "rows": rows}, f, indent=1)


if __name__ == "__main__":
    main()
