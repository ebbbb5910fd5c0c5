"""ORACLE (test infrastructure, not product code): numpy restatement of the reference driver's
hot-path orchestration.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this.

Shorthand: PnP.py = /root/reference/PnP_OVSS_0514_updated_segmentation.py
  parse_gpt_classes        PnP.py:726-787   Load_predicted_classes
  drop_loop                PnP.py:564-722   Inference_BLIP_filteredcaption
  merge_tokens             PnP.py:810-853   Mean_over_filtered_label_tokens
  threshold_upsample       PnP.py:348-379 / 424-455 (+ Scale_0_1 :1078-1094)
  bilinear_align_corners   torch F.interpolate(mode='bilinear', align_corners=True) generic CPU kernel
  gaussian_blur / blurring PnP.py:1149-1153 (scipy.ndimage.gaussian_filter, reflect, truncate 4)
  densecrf                 PnP.py:1030-1074 (C restatement in densecrf_ref.c; parity unpinned)
  postprocess              PnP.py:1002-1028
  remap_labels             PnP.py:390-399 / 468-480
  fast_hist, scores        PnP.py:1106-1146
COCO driver (PnPc.py = /root/reference/PnP_OVSS_0514_updated_segmentation_coco.py):
  parse_gpt_classes_coco   PnPc.py:858-963  Load_predicted_classes (category id -> position in `cats`)
  coco rules in segment_batch: 1-drop branch only when drop_iter < 3 (:420), Scale_0_1 on the N-drop branch too
  (:527), background rule (:446-450, :470-473), remap through cats[..]['id'] (:458-463, :482-489, :549-556, :577-584)
Pinned against tests/golden/*.npz produced by running the reference's own functions.
"""
import ctypes
import os

import numpy as np

from . import blip_itm_np as M

F32 = np.float32
_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libpnp_oracle.so")
        if not os.path.exists(path):
            import subprocess
            subprocess.check_call(["make", "-C", _HERE, "-s"])
        _LIB = ctypes.CDLL(path)
        _LIB.pnp_oracle_densecrf.restype = ctypes.c_int
    return _LIB


# --------------------------------------------------------------------------- a-8

def parse_gpt_classes(per_img_cls: str, nms):
    """PnP.py:746-783.  Returns (best_class_idx, class names, caption).  Raises like the reference
    (IndexError / ValueError) on strings it cannot split."""
    parts = per_img_cls.replace(']\n\n[', '], [').replace('],\n\n[', '], [').replace('], \n[', '], [ ') \
        .replace(']\n[', '], [ ').replace('],\n[', '], [ ').strip("][").split("], [")
    cls_list = parts[0].split(",")
    if len(parts) == 1 and parts[0] == '':
        cls_list = ["1: 'wall'" for _ in range(len(cls_list))]
        prob_list = [100 for _ in range(len(cls_list))]
    else:
        prob_list = [int(p.split(":")[-1].split("%")[0]) for p in parts[1].split(",")]
    idx = [int(cls_list[i].split(":")[0]) for i, p in enumerate(prob_list) if p > 70]
    best = [i - 1 for i in idx]
    names = [nms[i - 1] for i in idx]
    if not best:
        best, names = [0], [nms[0]]
    return best, names, "A picture of " + " ".join(names)


def parse_gpt_classes_coco(per_img_cls: str, cats, nms, data_type="coco_object"):
    """PnPc.py:870-905 (coco_object) / :918-958 (coco_stuff).  `cats` = list of {'id', 'name'} in pycocotools
    order, nms[j] the blank/dash-stripped name of cats[j] (PnPc.py:1399-1400).  The class number GPT-4o printed is a
    COCO category id; best_class_idx is its position in `cats`; ids that are no category are skipped."""
    parts = per_img_cls.replace(']\n\n[', '], [').replace('],\n\n[', '], [').replace('], \n[', '], [ ') \
        .replace('],\n[', '], [ ').replace(']\n[', '], [ ').strip("][").split("], [")
    cls_list = parts[0].split(",")
    if len(parts) == 1 and parts[0] == '':
        cls_list = ["1: 'person'" for _ in range(len(cls_list))]
        prob_list = [100 for _ in range(len(cls_list))]
    elif len(parts) == 1:
        prob_list = [100 for _ in range(len(cls_list))]
    else:
        prob_list = [int(p.split(":")[-1].split("%")[0]) for p in parts[1].split(",")]
    if data_type == "coco_stuff":
        prob_list = prob_list[:len(cls_list)]
    ids = []
    for i, p in enumerate(prob_list):
        if p > 70:
            if data_type == "coco_stuff":
                try:
                    ids.append(int(cls_list[i].split(":")[0]))
                except Exception:          # noqa: BLE001  (PnPc.py:943-946 swallows everything here)
                    pass
            else:
                ids.append(int(cls_list[i].split(":")[0]))
    best, names = [], []
    for v in ids:
        for j, cat in enumerate(cats):
            if cat["id"] == v:
                best.append(j)
                names.append(nms[j])
                break
    if not best:
        best, names = [0], [nms[0]]
    return best, names, "A picture of " + " ".join(names)


# --------------------------------------------------------------------------- a-7

def zero_patches(imgs, picks, grid, patch=16):
    """PnP.py:597-603: zero the 16x16 pixel blocks of every picked patch (normalised space)."""
    for b, pk in enumerate(picks):
        for p in pk:
            r, c = (p // grid) * patch, (p % grid) * patch
            imgs[b, :, r:r + patch, c:c + patch] = 0
    return imgs


def select_topk(sal, picked, k=10):
    """PnP.py:638-647: previously picked cells -> 0, then the k largest (np.argsort, ascending).
    The reference's default quicksort leaves ties (only ever among exact zeros) implementation
    defined; the build fixes them as a stable ascending sort (larger index wins the tail)."""
    s = sal.flatten().copy()
    for p in picked:
        s[p] = 0
    return [int(i) for i in np.argsort(s, kind="stable")[-k:]]


def drop_loop(W, cfg, imgs, ids500, mask500, drop_iter, layer, head, gradcam_fn=None):
    """Returns (g0 (B,L-1,P,P), agg or None, picks_per_iter: list[iter][b] -> new picks)."""
    if gradcam_fn is None:
        def gradcam_fn(x):
            maps, _, _ = M.compute_gradcam(W, cfg, x, ids500, mask500, layers=[layer])
            return maps[layer][:, head]
    B = imgs.shape[0]
    if drop_iter == 1:
        return gradcam_fn(imgs).copy(), None, []
    x = imgs.copy()
    picks = [[] for _ in range(B)]
    new_picks = []
    preds = []
    for it in range(drop_iter):
        zero_patches(x, picks, cfg.grid, cfg.patch)
        g = gradcam_fn(x)
        pred = g.copy()
        for b in range(B):
            for p in picks[b]:
                pred[b, :, p // cfg.grid, p % cfg.grid] = 0
        preds.append(pred)
        cur = []
        for b in range(B):
            sal = g[b, 3:-1].sum(axis=0, dtype=F32)
            sel = select_topk(sal, picks[b], 10)
            picks[b].extend(sel)
            cur.append(sel)
        new_picks.append(cur)
    g0 = preds[0].copy()
    agg = preds[0].copy()
    for it in range(drop_iter):
        agg = (agg + preds[it]).astype(F32)
    return g0, agg, new_picks


# --------------------------------------------------------------------------- a-9

def merge_plan(pieces, n_classes):
    """pieces: decoded word-piece strings of the caption AFTER '[ENC] a picture of' and before
    [SEP].  Returns per class (token index list, divisor) reproducing PnP.py:820-853 exactly,
    or None when #pieces == #classes (fast path `map[:C]`)."""
    if len(pieces) == n_classes:
        return None
    plan = [([], 1) for _ in range(n_classes)]
    it, ic, wl = 0, 0, 1
    n = len(pieces)
    while it < n:
        if not pieces[it].startswith("##"):
            plan[ic] = ([it], 1)            # IndexError here mirrors the reference's overflow
            if it + 1 < n and not pieces[it + 1].startswith("##"):
                ic += 1
            it += 1
            wl = 1
        else:
            wl += 1
            toks, _ = plan[ic]
            div = 1
            nxt = it + 1 < n and not pieces[it + 1].startswith("##")
            if nxt:
                div = wl
            plan[ic] = (toks + [it], div)
            if nxt:
                ic += 1
            it += 1
    return plan


def merge_tokens(maps, pieces, n_classes):
    """maps (L-1,P,P) one image's gradcam (rows: 'a','picture','of', class pieces..., last)."""
    g = maps[3:-1]
    plan = merge_plan(pieces, n_classes)
    if plan is None:
        return g[:n_classes].copy()
    out = np.zeros((n_classes,) + g.shape[1:], dtype=F32)
    for c, (toks, div) in enumerate(plan):
        if not toks:
            continue
        acc = g[toks[0]].copy()
        for t in toks[1:]:
            acc = (acc + g[t]).astype(F32)
        if div != 1:
            acc = (acc / F32(div)).astype(F32)
        out[c] = acc
    return out


# --------------------------------------------------------------------------- a-10

def _src_index(in_size, out_size):
    scale = F32(in_size - 1) / F32(out_size - 1) if out_size > 1 else F32(0)
    real = (scale * np.arange(out_size, dtype=F32)).astype(F32)
    i0 = np.minimum(np.floor(real).astype(np.int64), in_size - 1)
    lam = np.clip((real - i0.astype(F32)).astype(F32), 0, 1).astype(F32)
    i1 = i0 + (i0 < in_size - 1)
    return i0, i1, (F32(1) - lam).astype(F32), lam


def _fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(F32)


def bilinear_align_corners(x, H, W):
    """(C,h,w) -> (C,H,W): torch's upsample_generic_Nd_kernel_impl<2> arithmetic
    (out = fma(top, w0h, bot*w1h), top = fma(v00, w0w, v01*w1w)); bit-exact vs this container's
    torch 2.10 CPU build for H+W > 128 (the path every dataset image takes)."""
    h0, h1, wh0, wh1 = _src_index(x.shape[1], H)
    w0, w1, ww0, ww1 = _src_index(x.shape[2], W)
    v00 = x[:, h0][:, :, w0]
    v01 = x[:, h0][:, :, w1]
    v10 = x[:, h1][:, :, w0]
    v11 = x[:, h1][:, :, w1]
    a, b = wh0[None, :, None], wh1[None, :, None]
    c, d = ww0[None, None, :], ww1[None, None, :]
    top = _fma(v00, c, (v01 * d).astype(F32))
    bot = _fma(v10, c, (v11 * d).astype(F32))
    return _fma(top, a, (bot * b).astype(F32))


def scale_0_1(x):
    """PnP.py:1078-1094 (3-D branch): per-channel x -= min; x /= max."""
    if x.ndim == 2:
        return x
    c = x.shape[0]
    f = x.reshape(c, -1).copy()
    f = (f - f.min(axis=1, keepdims=True)).astype(F32)
    with np.errstate(all="ignore"):
        f = (f / f.max(axis=1, keepdims=True)).astype(F32)
    return f.reshape(x.shape)


def threshold_upsample(merged, H, W, threshold, apply_scale01, add_background):
    """PnP.py:348-379 (apply_scale01=True, 1-drop branch) / :424-455 (False, N-drop branch).
    merged (C,P,P) -> (K,H,W) float32, K = C (+1 background as channel 0)."""
    C = merged.shape[0]
    norm = np.empty_like(merged)
    with np.errstate(all="ignore"):
        for i in range(C):
            mn, mx = merged[i].min(), merged[i].max()
            norm[i] = (merged[i] - mn) / (mx - mn)
    keep = norm >= F32(threshold)
    pred = (merged * keep).astype(F32)
    up = bilinear_align_corners(pred, H, W)
    if C == 1:
        up2 = up[0]                       # .squeeze() drops the channel dim too (PnP.py:364-368)
        if apply_scale01:
            up2 = scale_0_1(up2)          # 2-D: returned unchanged
        mx = up2
        up = up2[None]
    else:
        if apply_scale01:
            up = scale_0_1(up)
        mx = up.max(axis=0)
    background = (mx == 0)[None].astype(F32)
    if add_background:
        return np.concatenate([background, up], axis=0).astype(F32)
    return up.astype(F32)


# --------------------------------------------------------------------------- a-11

def gaussian_kernel1d(sigma, truncate=4.0):
    """scipy.ndimage._filters._gaussian_kernel1d (order 0); float64."""
    radius = int(truncate * float(sigma) + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    return phi / phi.sum(), radius


def _correlate1d_reflect(x, w, radius, axis):
    """scipy NI_Correlate1D symmetric branch: double accumulation
    tmp = x[i]*w[0] + sum_{j=r..1} (x[i-j] + x[i+j]) * w[j]; result cast to float32."""
    x = np.moveaxis(x, axis, -1).astype(np.float64)
    n = x.shape[-1]
    idx = np.arange(-radius, n + radius)
    period = 2 * n
    idx = np.mod(idx, period)
    idx = np.where(idx >= n, period - 1 - idx, idx)     # half-sample symmetric (d c b a | a b c d | d c b a)
    ext = x[..., idx]
    c = radius
    out = ext[..., c:c + n] * w[radius]
    for j in range(radius, 0, -1):
        out = out + (ext[..., c - j:c - j + n] + ext[..., c + j:c + j + n]) * w[radius - j]
    return np.moveaxis(out.astype(F32), -1, axis)


def gaussian_blur(x, sigma):
    """scipy.ndimage.gaussian_filter(x(float32 2-D), sigma): axis 0 then axis 1, float32 between."""
    w, r = gaussian_kernel1d(sigma)
    y = _correlate1d_reflect(x.astype(F32), w, r, 0)
    return _correlate1d_reflect(y, w, r, 1)


def blurring(x, img_shape, scale=0.05):
    """PnP.py:1149-1153."""
    y = gaussian_blur(x, scale * max(img_shape))
    y = (y - y.min()).astype(F32)
    with np.errstate(all="ignore"):
        return (y / y.max()).astype(F32)


# --------------------------------------------------------------------------- a-12

CRF_PARAMS = dict(iters=10, pos_w=7.0, pos_xy=3.0, bi_w=10.0, bi_xy=50.0, bi_rgb=5.0)   # PnP.py:1036-1041


def densecrf(rgb, maps, want_q=False, **kw):
    """PnP.py:1030-1074.  rgb (H,W,3) uint8, maps (K,H,W) float32 -> (H,W) float32 label idx."""
    p = dict(CRF_PARAMS)
    p.update(kw)
    K, H, W = maps.shape
    maps = np.ascontiguousarray(maps, dtype=F32)
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    out = np.empty((H, W), dtype=F32)
    q = np.empty((K, H, W), dtype=F32) if want_q else None
    stats = np.zeros(2, dtype=np.int32)
    fp = ctypes.POINTER(ctypes.c_float)
    _lib().pnp_oracle_densecrf(
        maps.ctypes.data_as(fp), rgb.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)),
        ctypes.c_int(K), ctypes.c_int(H), ctypes.c_int(W), ctypes.c_int(p["iters"]),
        ctypes.c_float(p["pos_w"]), ctypes.c_float(p["pos_xy"]), ctypes.c_float(p["bi_w"]),
        ctypes.c_float(p["bi_xy"]), ctypes.c_float(p["bi_rgb"]),
        q.ctypes.data_as(fp) if want_q else None, out.ctypes.data_as(fp),
        stats.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    if want_q:
        return out, q, stats
    return out


def densecrf_variant(rgb, maps, variant=0, unary=None, **kw):
    """The same mean-field with the arithmetic the oracle fixes by definition swapped for what the reference's own stack may
    use (variant bit 0: libm exp / log; bit 1: float32 two-rounding lattice blur) and / or with externally computed unary
    energies (K,H,W) -- sensitivity tests only (tests/test_oracle_golden.py).  Returns (labels, marginals)."""
    p = dict(CRF_PARAMS)
    p.update(kw)
    K, H, W = maps.shape
    maps = np.ascontiguousarray(maps, dtype=F32)
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    out = np.empty((H, W), dtype=F32)
    q = np.empty((K, H, W), dtype=F32)
    fp = ctypes.POINTER(ctypes.c_float)
    u = np.ascontiguousarray(unary, dtype=F32) if unary is not None else None
    fn = _lib().pnp_oracle_densecrf_variant
    fn.restype = ctypes.c_int
    fn(maps.ctypes.data_as(fp), rgb.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), ctypes.c_int(K), ctypes.c_int(H),
       ctypes.c_int(W), ctypes.c_int(p["iters"]), ctypes.c_float(p["pos_w"]), ctypes.c_float(p["pos_xy"]), ctypes.c_float(p["bi_w"]),
       ctypes.c_float(p["bi_xy"]), ctypes.c_float(p["bi_rgb"]), q.ctypes.data_as(fp), out.ctypes.data_as(fp), ctypes.c_int(variant),
       u.ctypes.data_as(fp) if u is not None else None)
    return out, q


def crf_unary(maps):
    """The oracle's unary stage alone: softmax over channels, -log(clip(p, 1e-5, 1)) with include/pnp_math.h's exp / log."""
    K, H, W = maps.shape
    maps = np.ascontiguousarray(maps, dtype=F32)
    out = np.empty((K, H, W), dtype=F32)
    fp = ctypes.POINTER(ctypes.c_float)
    _lib().pnp_oracle_unary(maps.ctypes.data_as(fp), ctypes.c_int(K), ctypes.c_int(H * W), out.ctypes.data_as(fp))
    return out


def postprocess(mode, maps, rgb, hw):
    """PnP.py:1002-1028.  mode: 'blur+crf' | 'crf' | 'blur' | None."""
    if mode is None:
        return np.argmax(maps, axis=0).astype(F32)
    if "blur" in mode:
        maps = np.stack([blurring(maps[i], hw) for i in range(maps.shape[0])])
    if "crf" in mode:
        return densecrf(rgb, maps)
    return np.argmax(maps, axis=0).astype(F32)


# --------------------------------------------------------------------------- a-13 / a-14

def remap_labels(label_map, best_class_idx, has_background, class_ids=None):
    """PnP.py:390-399: descending i, in place, order-dependent collisions kept.  COCO (PnPc.py:458-463 ...):
    the target is cats[best_class_idx[i]]['id'] (`class_ids[j]` = id of cats[j]) instead of index + 1."""
    out = label_map.copy()
    for i in range(len(best_class_idx) - 1, -1, -1):
        src = i + 1 if has_background else i
        out[out == src] = best_class_idx[i] + 1 if class_ids is None else class_ids[best_class_idx[i]]
    return out


def has_background(data_type, n_sel):
    """PnP.py:373-379; PnPc.py:446-450 / :470-473 (coco_object always, coco_stuff like the context datasets)."""
    return data_type in ("voc", "coco_object") or n_sel < 3


def fast_hist(label_true, label_pred, n_class):
    mask = (label_true >= 0) & (label_true < n_class)
    return np.bincount(n_class * label_true[mask].astype(int) + label_pred[mask].astype(int),
                       minlength=n_class ** 2).reshape(n_class, n_class)


def scores(label_trues, label_preds, n_class):
    hist = np.zeros((n_class, n_class))
    for lt, lp in zip(label_trues, label_preds):
        hist += fast_hist(lt.flatten(), lp.flatten(), n_class)
    with np.errstate(all="ignore"):
        acc = np.diag(hist).sum() / hist.sum()
        acc_cls = np.nanmean(np.diag(hist) / hist.sum(axis=1))
        iu = np.diag(hist) / (hist.sum(axis=1) + hist.sum(axis=0) - np.diag(hist))
        valid = hist.sum(axis=1) > 0
        mean_iu = np.nanmean(iu[valid])
        freq = hist.sum(axis=1) / hist.sum()
        fwavacc = (freq[freq > 0] * iu[freq > 0]).sum()
    return {"Pixel Accuracy": acc, "Mean Accuracy": acc_cls, "Frequency Weighted IoU": fwavacc,
            "Mean IoU": mean_iu, "Class IoU": iu}, hist


# --------------------------------------------------------------------------- whole image batch

def segment_batch(W, cfg, imgs, ids500, mask500, pieces_per_img, best_idx_per_img, rgbs, sizes,
                  data_type="voc", drop_iter=4, layer=7, head=9, threshold=0.15, mode="blur+crf",
                  run_1drop=True, gradcam_fn=None, class_ids=None):
    """save_img_union_attention (PnP.py:290-521; COCO driver PnPc.py:338-642) minus file I/O: returns
    (labels_1drop list|None, labels_ndrop list|None, aux dict).  data_type coco_object / coco_stuff: the 1-drop
    branch runs only when drop_iter < 3, the N-drop branch is Scale_0_1'ed too, labels are COCO category ids
    (`class_ids[j]` = cats[j]['id'])."""
    g0, agg, picks = drop_loop(W, cfg, imgs, ids500, mask500, drop_iter, layer, head, gradcam_fn)
    out1, outn, pre = [], [], {"1": [], "n": []}
    B = imgs.shape[0]
    coco = data_type.startswith("coco")
    if coco and drop_iter >= 3:
        run_1drop = False
    for branch, src, scale01, dst in (("1", g0, True, out1), ("n", agg, coco, outn)):
        if src is None or (branch == "1" and not run_1drop):
            continue
        for b in range(B):
            best = best_idx_per_img[b]
            merged = merge_tokens(src[b], pieces_per_img[b], len(best))
            H, Wd = sizes[b]
            bg = has_background(data_type, len(best))
            maps = threshold_upsample(merged, H, Wd, threshold, scale01, bg)
            pre[branch].append(maps)
            lab = postprocess(mode, maps, rgbs[b], (H, Wd))
            dst.append(remap_labels(lab, best, bg, class_ids))
    return (out1 or None), (outn or None), dict(g0=g0, agg=agg, picks=picks, pre=pre)
