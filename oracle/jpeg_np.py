"""ORACLE (test infrastructure, not product code): baseline JPEG decoder in numpy / plain Python, restating what the
reference's input side gets from Pillow: `Image.open(path).convert('RGB')` (Dataset.py:349-445 PascalVOC.__getitem__,
PnP_OVSS_0514_updated_segmentation.py:929-955 load_OrgImage).  Pillow decodes with libjpeg(-turbo) defaults, an
un-vendored third-party library, so this follows its published algorithm: sequential Huffman baseline (ITU T.81), the
"islow" integer inverse DCT (jidctint.c: 13-bit constants, two passes, DESCALE rounding), "fancy" triangle-filter chroma
upsampling (jdsample.c h2v1 / h2v2) and the fixed-point YCbCr -> RGB tables of jdcolor.c.  Pinned bit-exact against Pillow
itself on JPEGs encoded here (tests/test_oracle_golden.py::test_jpeg_oracle_matches_pillow).

The parser half (`parse_jpeg`) is shared with the product host code path through pnp_ovss.jpeg (same tables go to the
device kernels); the arithmetic half below is only the checker.
"""
import numpy as np

ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21,
                   28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61,
                   54, 47, 55, 62, 63], dtype=np.int32)


class JpegError(ValueError):
    pass


def parse_jpeg(data: bytes):
    """Markers of a baseline (SOF0 / SOF1, 8-bit, Huffman, non-progressive) JFIF stream -> dict with frame geometry,
    quantisation tables (natural order), Huffman tables (counts + symbols), restart interval and the entropy-coded
    segment.  Raises JpegError for anything else (progressive, arithmetic, 12-bit, CMYK ...)."""
    if data[:2] != b"\xff\xd8":
        raise JpegError("not a JPEG (no SOI)")
    pos = 2
    qt, ht = {}, {}
    frame = None
    dri = 0
    n = len(data)
    while pos < n:
        if data[pos] != 0xFF:
            raise JpegError(f"marker expected at {pos}")
        while pos < n and data[pos] == 0xFF:
            pos += 1
        m = data[pos]
        pos += 1
        if m == 0xD9:
            break
        if m == 0x01 or 0xD0 <= m <= 0xD7:
            continue
        L = (data[pos] << 8) | data[pos + 1]
        seg = data[pos + 2:pos + L]
        if m == 0xDB:
            p = 0
            while p < len(seg):
                pq, tq = seg[p] >> 4, seg[p] & 15
                p += 1
                if pq:
                    vals = [(seg[p + 2 * i] << 8) | seg[p + 2 * i + 1] for i in range(64)]
                    p += 128
                else:
                    vals = list(seg[p:p + 64])
                    p += 64
                t = np.zeros(64, dtype=np.int32)
                t[ZIGZAG] = vals
                qt[tq] = t
        elif m in (0xC0, 0xC1):
            if seg[0] != 8:
                raise JpegError("only 8-bit samples")
            H, W, nc = (seg[1] << 8) | seg[2], (seg[3] << 8) | seg[4], seg[5]
            comps = [dict(id=seg[6 + 3 * i], h=seg[7 + 3 * i] >> 4, v=seg[7 + 3 * i] & 15, tq=seg[8 + 3 * i]) for i in range(nc)]
            frame = dict(H=H, W=W, comps=comps)
        elif 0xC2 <= m <= 0xCF and m not in (0xC4, 0xC8, 0xCC):
            raise JpegError(f"unsupported JPEG process (SOF{m - 0xC0}: progressive / lossless / arithmetic)")
        elif m == 0xC4:
            p = 0
            while p < len(seg):
                tc, th = seg[p] >> 4, seg[p] & 15
                counts = list(seg[p + 1:p + 17])
                ns = sum(counts)
                ht[(tc, th)] = (counts, list(seg[p + 17:p + 17 + ns]))
                p += 17 + ns
        elif m == 0xDD:
            dri = (seg[0] << 8) | seg[1]
        elif m == 0xDA:
            ns = seg[0]
            if frame is None or ns != len(frame["comps"]):
                raise JpegError("only single-scan (interleaved) baseline files")
            for i in range(ns):
                cs, t = seg[1 + 2 * i], seg[2 + 2 * i]
                c = [c for c in frame["comps"] if c["id"] == cs][0]
                c["td"], c["ta"] = t >> 4, t & 15
            if seg[1 + 2 * ns] != 0 or seg[2 + 2 * ns] != 63:
                raise JpegError("spectral selection in a baseline scan")
            start = pos + L
            end = data.rfind(b"\xff\xd9")
            if end < start:
                end = n
            return dict(frame=frame, qt=qt, ht=ht, dri=dri, scan=data[start:end])
        pos += L
    raise JpegError("no SOS marker")


def geometry(frame):
    comps = frame["comps"]
    if len(comps) not in (1, 3):
        raise JpegError("only grayscale / YCbCr")
    hmax = max(c["h"] for c in comps)
    vmax = max(c["v"] for c in comps)
    if len(comps) == 3 and not (comps[1]["h"] == comps[2]["h"] == 1 and comps[1]["v"] == comps[2]["v"] == 1 and
                                (comps[0]["h"], comps[0]["v"]) in ((1, 1), (2, 1), (2, 2))):
        raise JpegError("only 4:4:4 / 4:2:2 / 4:2:0 sampling")
    if len(comps) == 1:
        hmax = vmax = 1
        comps[0]["h"] = comps[0]["v"] = 1
    mcux = -(-frame["W"] // (8 * hmax))
    mcuy = -(-frame["H"] // (8 * vmax))
    return hmax, vmax, mcux, mcuy


def huff_lookup(counts, symbols):
    """codes by (length, code) -> symbol; returned as dict for the Python decoder and as the canonical
    (mincode, maxcode, valptr) arrays of T.81 F.2.2.3 that the device decoder walks."""
    code, k = 0, 0
    table = {}
    mincode, maxcode, valptr = [0] * 17, [-1] * 17, [0] * 17
    for l in range(1, 17):
        valptr[l] = k
        mincode[l] = code
        for _ in range(counts[l - 1]):
            table[(l, code)] = symbols[k]
            code += 1
            k += 1
        maxcode[l] = code - 1 if counts[l - 1] else -1
        code <<= 1
    return table, mincode, maxcode, valptr


class _Bits:
    def __init__(self, data):
        self.d, self.p, self.acc, self.n = data, 0, 0, 0

    def bit(self):
        if self.n == 0:
            if self.p >= len(self.d):
                b = 0
            else:
                b = self.d[self.p]
                self.p += 1
                if b == 0xFF:
                    nxt = self.d[self.p] if self.p < len(self.d) else 0
                    if nxt == 0:
                        self.p += 1
                    else:                 # a marker inside the data: feed zeros (libjpeg does the same)
                        self.p -= 1
                        b = 0
            self.acc, self.n = b, 8
        self.n -= 1
        return (self.acc >> self.n) & 1

    def bits(self, k):
        v = 0
        for _ in range(k):
            v = (v << 1) | self.bit()
        return v

    def restart(self):
        self.n = 0
        while self.p + 1 < len(self.d) and not (self.d[self.p] == 0xFF and 0xD0 <= self.d[self.p + 1] <= 0xD7):
            self.p += 1
        self.p += 2


def _extend(v, t):
    return v if t == 0 or v >= (1 << (t - 1)) else v - (1 << t) + 1


def decode_coefficients(j):
    """Entropy decode -> per component int32 array [blocks_y, blocks_x, 64] of DEQUANTISED coefficients, natural order."""
    frame = j["frame"]
    hmax, vmax, mcux, mcuy = geometry(frame)
    comps = frame["comps"]
    tabs = {k: huff_lookup(*v)[0] for k, v in j["ht"].items()}
    out = [np.zeros((mcuy * c["v"], mcux * c["h"], 64), dtype=np.int32) for c in comps]
    br = _Bits(j["scan"])
    pred = [0] * len(comps)

    def sym(tab):
        code = 0
        for l in range(1, 17):
            code = (code << 1) | br.bit()
            s = tab.get((l, code))
            if s is not None:
                return s
        raise JpegError("bad Huffman code")
    count = 0
    for my in range(mcuy):
        for mx in range(mcux):
            if j["dri"] and count and count % j["dri"] == 0:
                br.restart()
                pred = [0] * len(comps)
            count += 1
            for ci, c in enumerate(comps):
                dc_t, ac_t, q = tabs[(0, c["td"])], tabs[(1, c["ta"])], j["qt"][c["tq"]]
                for by in range(c["v"]):
                    for bx in range(c["h"]):
                        blk = out[ci][my * c["v"] + by, mx * c["h"] + bx]
                        t = sym(dc_t)
                        pred[ci] += _extend(br.bits(t), t)
                        blk[0] = pred[ci] * q[0]
                        k = 1
                        while k < 64:
                            rs = sym(ac_t)
                            r, s = rs >> 4, rs & 15
                            if s == 0:
                                if r != 15:
                                    break
                                k += 16
                                continue
                            k += r
                            if k > 63:
                                raise JpegError("coefficient index past 63")
                            z = ZIGZAG[k]
                            blk[z] = _extend(br.bits(s), s) * q[z]
                            k += 1
    return out


# jidctint.c constants (CONST_BITS = 13, PASS1_BITS = 2)
_C = dict(f0298=2446, f0390=3196, f0541=4433, f0765=6270, f0899=7373, f1175=9633, f1501=12299, f1847=15137, f1961=16069,
          f2053=16819, f2562=20995, f3072=25172)


def _idct_1d(x0, x1, x2, x3, x4, x5, x6, x7, first):
    """One pass of jpeg_idct_islow on eight int64 arrays; first=True: column pass (<< 13, descale 11), else row pass
    (descale 18)."""
    c = _C
    z1 = (x2 + x6) * c["f0541"]
    tmp2 = z1 + x6 * (-c["f1847"])
    tmp3 = z1 + x2 * c["f0765"]
    tmp0 = (x0 + x4) << 13
    tmp1 = (x0 - x4) << 13
    tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
    t0, t1, t2, t3 = x7, x5, x3, x1
    z1, z2, z3, z4 = t0 + t3, t1 + t2, t0 + t2, t1 + t3
    z5 = (z3 + z4) * c["f1175"]
    t0, t1, t2, t3 = t0 * c["f0298"], t1 * c["f2053"], t2 * c["f3072"], t3 * c["f1501"]
    z1, z2, z3, z4 = z1 * (-c["f0899"]), z2 * (-c["f2562"]), z3 * (-c["f1961"]) + z5, z4 * (-c["f0390"]) + z5
    t0, t1, t2, t3 = t0 + z1 + z3, t1 + z2 + z4, t2 + z2 + z3, t3 + z1 + z4
    n = 11 if first else 18
    r = 1 << (n - 1)
    return [(v + r) >> n for v in (tmp10 + t3, tmp11 + t2, tmp12 + t1, tmp13 + t0, tmp13 - t0, tmp12 - t1, tmp11 - t2, tmp10 - t3)]


def idct_islow(blocks):
    """[..., 64] dequantised coefficients -> [..., 8, 8] samples 0..255 (jidctint.c jpeg_idct_islow)."""
    b = blocks.astype(np.int64).reshape(blocks.shape[:-1] + (8, 8))
    cols = _idct_1d(*[b[..., i, :] for i in range(8)], first=True)            # along rows index (vertical pass)
    ws = np.stack(cols, axis=-2)
    rows = _idct_1d(*[ws[..., :, i] for i in range(8)], first=False)
    out = np.stack(rows, axis=-1) + 128
    return np.clip(out, 0, 255).astype(np.uint8)


def _plane(blocks):
    by, bx = blocks.shape[:2]
    return idct_islow(blocks).transpose(0, 2, 1, 3).reshape(by * 8, bx * 8)


def upsample_h2v1_fancy(p, out_w):
    """jdsample.c h2v1_fancy_upsample on the first ceil(out_w / 2) columns of every row."""
    w = -(-out_w // 2)
    a = p[:, :w].astype(np.int32)
    prev = np.concatenate([a[:, :1], a[:, :-1]], axis=1)
    nxt = np.concatenate([a[:, 1:], a[:, -1:]], axis=1)
    even = (3 * a + prev + 1) >> 2
    odd = (3 * a + nxt + 2) >> 2
    even[:, 0] = a[:, 0]
    odd[:, -1] = a[:, -1]
    out = np.empty((a.shape[0], 2 * w), dtype=np.int32)
    out[:, 0::2], out[:, 1::2] = even, odd
    return out[:, :out_w].astype(np.uint8)


def upsample_h2v2_fancy(p, out_h, out_w):
    """jdsample.c h2v2_fancy_upsample: vertical 3:1 blend towards the nearer row (edge rows replicated), then the
    horizontal triangle filter on column sums with the +8 / +7 rounding pair."""
    h, w = -(-out_h // 2), -(-out_w // 2)
    a = p[:h, :w].astype(np.int32)
    up = np.concatenate([a[:1], a[:-1]], axis=0)
    dn = np.concatenate([a[1:], a[-1:]], axis=0)
    rows = np.empty((2 * h, w), dtype=np.int32)
    rows[0::2] = 3 * a + up
    rows[1::2] = 3 * a + dn
    prev = np.concatenate([rows[:, :1], rows[:, :-1]], axis=1)
    nxt = np.concatenate([rows[:, 1:], rows[:, -1:]], axis=1)
    even = (3 * rows + prev + 8) >> 4
    odd = (3 * rows + nxt + 7) >> 4
    even[:, 0] = (rows[:, 0] * 4 + 8) >> 4
    odd[:, -1] = (rows[:, -1] * 4 + 7) >> 4
    out = np.empty((2 * h, 2 * w), dtype=np.int32)
    out[:, 0::2], out[:, 1::2] = even, odd
    return out[:out_h, :out_w].astype(np.uint8)


def ycc_to_rgb(y, cb, cr):
    """jdcolor.c ycc_rgb_convert with its 16-bit fixed-point tables."""
    y, cb, cr = y.astype(np.int32), cb.astype(np.int32) - 128, cr.astype(np.int32) - 128
    r = y + ((91881 * cr + 32768) >> 16)
    g = y + ((-22554 * cb + 32768 - 46802 * cr) >> 16)
    b = y + ((116130 * cb + 32768) >> 16)
    return np.clip(np.stack([r, g, b], axis=-1), 0, 255).astype(np.uint8)


def decode(data: bytes):
    """bytes of a baseline JPEG -> RGB uint8 [H, W, 3], as Pillow's Image.open(...).convert('RGB')."""
    j = parse_jpeg(data)
    frame = j["frame"]
    H, W = frame["H"], frame["W"]
    coefs = decode_coefficients(j)
    planes = [_plane(c) for c in coefs]
    if len(planes) == 1:
        y = planes[0][:H, :W]
        return np.stack([y, y, y], axis=-1)
    c0 = frame["comps"][0]
    y = planes[0][:H, :W]
    if (c0["h"], c0["v"]) == (1, 1):
        cb, cr = planes[1][:H, :W], planes[2][:H, :W]
    elif (c0["h"], c0["v"]) == (2, 1):
        cb, cr = upsample_h2v1_fancy(planes[1][:H], W), upsample_h2v1_fancy(planes[2][:H], W)
    else:
        cb, cr = upsample_h2v2_fancy(planes[1], H, W), upsample_h2v2_fancy(planes[2], H, W)
    return ycc_to_rgb(y, cb, cr)
