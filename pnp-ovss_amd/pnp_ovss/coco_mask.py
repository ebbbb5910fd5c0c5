"""COCO segmentation -> binary mask on the host, for the coco_object ground truth (PnP_OVSS_0514_updated_segmentation_
coco.py:1099-1110 calls `coco_thing.annToMask(ann)`).  pycocotools is an un-vendored dependency of the reference and
is not installed here, so this restates its published algorithm (cocoapi common/maskApi.c: rleFrPoly, rleMerge-by-
union for multi-part polygons, rleDecode, rleFrString) in numpy.  It only feeds the confusion-matrix ground truth --
nothing on the device path -- and has no pycocotools fixture yet ("unpinned"; tests check rasterisation properties).

Masks are column-major run-length encoded: counts alternate 0-runs / 1-runs over pixel index x*h + y.
"""
import numpy as np


def _rle_from_polygon(xy, h, w):
    """maskApi.c rleFrPoly: 5x upsampled boundary walk -> y-boundary crossings -> run lengths."""
    k = len(xy) // 2
    scale = 5.0
    x = [int(scale * xy[2 * j] + .5) for j in range(k)]
    y = [int(scale * xy[2 * j + 1] + .5) for j in range(k)]
    x.append(x[0])
    y.append(y[0])
    u, v = [], []
    for j in range(k):
        xs, xe, ys, ye = x[j], x[j + 1], y[j], y[j + 1]
        dx, dy = abs(xe - xs), abs(ys - ye)
        flip = (dx >= dy and xs > xe) or (dx < dy and ys > ye)
        if flip:
            xs, xe, ys, ye = xe, xs, ye, ys
        if dx >= dy:
            s = (ye - ys) / dx if dx else 0.0
            for d in range(dx + 1):
                t = dx - d if flip else d
                u.append(t + xs)
                v.append(int(ys + s * t + .5))
        else:
            s = (xe - xs) / dy
            for d in range(dy + 1):
                t = dy - d if flip else d
                v.append(t + ys)
                u.append(int(xs + s * t + .5))
    a = []
    for j in range(1, len(u)):
        if u[j] != u[j - 1]:
            xd = float(u[j] if u[j] < u[j - 1] else u[j] - 1)
            xd = (xd + .5) / scale - .5
            if np.floor(xd) != xd or xd < 0 or xd > w - 1:
                continue
            yd = float(v[j] if v[j] < v[j - 1] else v[j - 1])
            yd = (yd + .5) / scale - .5
            yd = 0.0 if yd < 0 else (float(h) if yd > h else yd)
            a.append(int(xd) * h + int(np.ceil(yd)))
    a.append(h * w)
    a.sort()
    counts, p = [], 0
    for t in a:
        counts.append(t - p)
        p = t
    b = [counts[0]]
    j = 1
    while j < len(counts):
        if counts[j] > 0:
            b.append(counts[j])
            j += 1
        else:
            j += 1
            if j < len(counts):
                b[-1] += counts[j]
                j += 1
    return b


def _rle_from_string(s):
    """maskApi.c rleFrString: LEB128-like 5-bit groups, deltas against the count two positions back."""
    cnts, p = [], 0
    s = s.encode() if isinstance(s, str) else bytes(s)
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = s[p] - 48
            x |= (c & 0x1f) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(cnts) > 2:
            x += cnts[-2]
        cnts.append(x)
    return cnts


def _decode(counts, h, w):
    flat = np.zeros(h * w, dtype=np.uint8)
    pos, val = 0, 0
    for c in counts:
        if val:
            flat[pos:pos + c] = 1
        pos += c
        val ^= 1
    return flat.reshape(w, h).T                      # column-major -> (h, w)


def ann_to_mask(ann, h, w):
    """pycocotools COCO.annToMask: polygons (list of flat xy lists, merged by union), uncompressed RLE
    ({'counts': [..], 'size': [h, w]}) or compressed RLE ({'counts': str}).  Returns uint8 (h, w)."""
    seg = ann["segmentation"]
    if isinstance(seg, list):
        m = np.zeros((h, w), dtype=np.uint8)
        for poly in seg:
            m |= _decode(_rle_from_polygon(poly, h, w), h, w)
        return m
    counts = seg["counts"]
    if isinstance(counts, list):
        return _decode(counts, h, w)
    return _decode(_rle_from_string(counts), h, w)
