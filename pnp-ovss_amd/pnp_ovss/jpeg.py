"""Host half of the device JPEG decoder (csrc/jpeg.hip, `pnp_jpeg_decode`): the reference reads every image with
`Image.open(path).convert('RGB')` (Dataset.py:349-445, PnP_OVSS_0514_updated_segmentation.py:929-955); here the host only
walks the JPEG markers (byte work: frame geometry, quantisation and Huffman tables, restart intervals) and hands the
entropy-coded bytes plus descriptors to the GPU, which does the Huffman decode, inverse DCT, chroma upsampling and colour
conversion for the whole batch.  Baseline sequential files (grayscale / YCbCr 4:4:4, 4:2:2, 4:2:0) are supported --
what VOC / COCO / ADE20K ship; anything else (progressive, arithmetic, CMYK) raises `UnsupportedJpeg` and the dataset
decodes that one file with Pillow.
"""
import ctypes as C

import numpy as np

ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21,
                   28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61,
                   54, 47, 55, 62, 63], dtype=np.int32)


class UnsupportedJpeg(ValueError):
    pass


class JpegImage(C.Structure):
    _fields_ = [("data_off", C.c_int64), ("coef_off", C.c_int64 * 3), ("plane_off", C.c_int64 * 3), ("rgb_off", C.c_int64)] + \
               [(n, C.c_int32) for n in ("data_len", "H", "W", "ncomp", "hmax", "vmax", "mcux", "mcuy")] + \
               [(n, C.c_int32 * 3) for n in ("h", "v", "tq", "td", "ta", "bx", "by")] + [("tab", C.c_int32), ("pad", C.c_int32 * 2)]


class JpegTables(C.Structure):
    _fields_ = [("counts", (C.c_uint8 * 16) * 4), ("vals", (C.c_uint8 * 256) * 4), ("quant", (C.c_int32 * 64) * 4)]


class JpegSegment(C.Structure):
    _fields_ = [("byte_off", C.c_int64), ("clean_off", C.c_int64)] + \
               [(n, C.c_int32) for n in ("image", "mcu0", "nmcu", "raw_len", "clean_cap", "sub_bits")] + [("pad", C.c_int32 * 2)]


SUBSEQUENCES = 1024     # threads per restart segment of the device decoder (csrc/jpeg.hip JPEG_T)


def _u16(b, p):
    return (b[p] << 8) | b[p + 1]


def parse(data: bytes):
    """Marker walk of one file -> dict(H, W, comps[{h, v, tq, td, ta}], qt{id: int32[64] natural order},
    ht{(class, id): (counts[16], symbols)}, dri, scan_start, scan_end).  Anything the device decoder does not cover --
    including truncated / malformed marker segments and 3-component files that are NOT YCbCr (Adobe APP14 transform 0,
    or component ids 'R','G','B': libjpeg would not colour-convert them) -- raises UnsupportedJpeg, so the caller falls
    back to Pillow instead of failing or silently converting the wrong colour space."""
    try:
        return _parse(data)
    except UnsupportedJpeg:
        raise
    except (IndexError, ValueError, KeyError) as exc:
        raise UnsupportedJpeg(f"malformed marker segment ({type(exc).__name__}: {exc})") from None


def _parse(data: bytes):
    if len(data) < 4 or data[0] != 0xFF or data[1] != 0xD8:
        raise UnsupportedJpeg("not a JPEG stream")
    pos, n = 2, len(data)
    qt, ht, comps, H, W, dri = {}, {}, None, 0, 0, 0
    adobe_transform = None
    while pos + 4 <= n:
        if data[pos] != 0xFF:
            raise UnsupportedJpeg("marker expected")
        while pos < n and data[pos] == 0xFF:
            pos += 1
        m = data[pos]
        pos += 1
        if m == 0xD9:
            break
        if m == 0x01 or 0xD0 <= m <= 0xD7:
            continue
        L = _u16(data, pos)
        seg = data[pos + 2:pos + L]
        if m == 0xDB:
            p = 0
            while p < len(seg):
                pq, tq = seg[p] >> 4, seg[p] & 15
                p += 1
                if pq:
                    vals = [_u16(seg, p + 2 * i) for i in range(64)]
                    p += 128
                else:
                    vals = list(seg[p:p + 64])
                    p += 64
                t = np.zeros(64, dtype=np.int32)
                t[ZIGZAG] = vals
                qt[tq] = t
        elif m in (0xC0, 0xC1):
            if seg[0] != 8:
                raise UnsupportedJpeg("sample precision is not 8 bits")
            H, W, nc = _u16(seg, 1), _u16(seg, 3), seg[5]
            comps = [dict(id=seg[6 + 3 * i], h=seg[7 + 3 * i] >> 4, v=seg[7 + 3 * i] & 15, tq=seg[8 + 3 * i]) for i in range(nc)]
        elif 0xC2 <= m <= 0xCF and m not in (0xC4, 0xC8, 0xCC):
            raise UnsupportedJpeg(f"SOF{m - 0xC0}: progressive / lossless / arithmetic coding")
        elif m == 0xC4:
            p = 0
            while p < len(seg):
                tc, th = seg[p] >> 4, seg[p] & 15
                counts = list(seg[p + 1:p + 17])
                ns = sum(counts)
                ht[(tc, th)] = (counts, list(seg[p + 17:p + 17 + ns]))
                p += 17 + ns
        elif m == 0xDD:
            dri = _u16(seg, 0)
        elif m == 0xEE and len(seg) >= 12 and bytes(seg[:5]) == b"Adobe":
            adobe_transform = seg[11]                  # 0: no colour transform (RGB / CMYK), 1: YCbCr, 2: YCCK
        elif m == 0xDA:
            if comps is None:
                raise UnsupportedJpeg("scan before frame header")
            ns = seg[0]
            if ns != len(comps):
                raise UnsupportedJpeg("non-interleaved scans")
            for i in range(ns):
                c = [c for c in comps if c["id"] == seg[1 + 2 * i]][0]
                c["td"], c["ta"] = seg[2 + 2 * i] >> 4, seg[2 + 2 * i] & 15
            start = pos + L
            end = data.rfind(b"\xff\xd9")
            if end < start:
                end = n
            if len(comps) == 1:
                comps[0]["h"] = comps[0]["v"] = 1
            elif len(comps) == 3:
                if adobe_transform == 0 or [c["id"] for c in comps] == [82, 71, 66]:
                    raise UnsupportedJpeg("3-component file stored as RGB, not YCbCr (Adobe transform 0 / component ids R,G,B)")
                c0 = comps[0]
                if not (comps[1]["h"] == comps[2]["h"] == comps[1]["v"] == comps[2]["v"] == 1 and
                        (c0["h"], c0["v"]) in ((1, 1), (2, 1), (2, 2))):
                    raise UnsupportedJpeg("chroma sampling other than 4:4:4 / 4:2:2 / 4:2:0")
            else:
                raise UnsupportedJpeg(f"{len(comps)} components")
            for c in comps:
                if c["tq"] not in qt or (0, c["td"]) not in ht or (1, c["ta"]) not in ht or c["td"] > 1 or c["ta"] > 1 or c["tq"] > 3:
                    raise UnsupportedJpeg("missing table")
            return dict(H=H, W=W, comps=comps, qt=qt, ht=ht, dri=dri, scan_start=start, scan_end=end)
        pos += L
    raise UnsupportedJpeg("no scan")


def _fill_tables(tab, j):
    """The file's DHT (code counts per length, symbols) and DQT tables as they are; the device builds its look-ahead tables
    (T.81 F.2.2.3 canonical codes) from them."""
    for (tc, th), (counts, symbols) in j["ht"].items():
        if th > 1:
            continue
        if sum(counts) > 256 or len(symbols) != sum(counts):
            raise UnsupportedJpeg("bad Huffman table")
        t = tc * 2 + th
        tab.counts[t][:] = counts
        tab.vals[t][:len(symbols)] = symbols
    for tq, q in j["qt"].items():
        if tq < 4:
            tab.quant[tq][:] = q.tolist()


def _table_key(j):
    return (tuple(sorted((k, tuple(c), tuple(v)) for k, (c, v) in j["ht"].items())), tuple(sorted((k, q.tobytes()) for k, q in j["qt"].items())))


def pack_batch(files):
    """list of JPEG byte strings -> (data uint8 array, JpegImage[], JpegTables[], JpegSegment[], sizes, totals) ready to
    upload.  totals = dict(coef_elems, plane_bytes, rgb_bytes, clean_bytes, max_blocks, max_pixels).  Files with the same
    DHT + DQT bytes (every file an encoder wrote with its default tables at one quality) share one JpegTables entry."""
    n = len(files)
    imgs = (JpegImage * n)()
    tab_index, tab_list = {}, []
    segs = []
    chunks, data_off = [], 0
    coef = plane = rgb = clean = 0
    max_blocks = max_pixels = 0
    sizes = []
    for i, f in enumerate(files):
        j = parse(f)
        comps = j["comps"]
        hmax = max(c["h"] for c in comps)
        vmax = max(c["v"] for c in comps)
        mcux = -(-j["W"] // (8 * hmax))
        mcuy = -(-j["H"] // (8 * vmax))
        scan = np.frombuffer(f, dtype=np.uint8, count=j["scan_end"] - j["scan_start"], offset=j["scan_start"])
        im = imgs[i]
        im.data_off, im.data_len = data_off, len(scan)
        im.H, im.W, im.ncomp, im.hmax, im.vmax, im.mcux, im.mcuy = j["H"], j["W"], len(comps), hmax, vmax, mcux, mcuy
        blocks = 0
        for ci, c in enumerate(comps):
            im.h[ci], im.v[ci], im.tq[ci], im.td[ci], im.ta[ci] = c["h"], c["v"], c["tq"], c["td"], c["ta"]
            im.bx[ci], im.by[ci] = mcux * c["h"], mcuy * c["v"]
            nb = im.bx[ci] * im.by[ci]
            im.coef_off[ci], im.plane_off[ci] = coef, plane
            coef += nb * 64
            plane += nb * 64
            blocks += nb
        im.rgb_off = rgb
        key = _table_key(j)
        if key not in tab_index:
            tab_index[key] = len(tab_list)
            t = JpegTables()
            _fill_tables(t, j)
            tab_list.append(t)
        im.tab = tab_index[key]
        rgb += j["H"] * j["W"] * 3
        sizes.append((j["H"], j["W"]))
        max_blocks = max(max_blocks, blocks)
        max_pixels = max(max_pixels, j["H"] * j["W"])
        # restart intervals: every RSTn marker starts an independently decodable segment
        nmcu = mcux * mcuy
        starts, ends = [0], []
        if j["dri"]:
            ff = np.flatnonzero(scan[:-1] == 0xFF)
            rst = ff[(scan[ff + 1] >= 0xD0) & (scan[ff + 1] <= 0xD7)]
            starts += [int(p) + 2 for p in rst]
            ends = [int(p) for p in rst]
        ends.append(len(scan))
        for k, (s0, s1) in enumerate(zip(starts, ends)):
            m0 = k * j["dri"] if j["dri"] else 0
            if m0 >= nmcu:
                break
            raw = s1 - s0
            cap = ((raw + 15) & ~15) + 32
            sub = max(128, (-(-raw * 8 // SUBSEQUENCES) + 31) & ~31)
            segs.append((s0, clean, i, m0, min(j["dri"], nmcu - m0) if j["dri"] else nmcu, raw, cap, sub))
            clean += cap
        chunks.append(scan)
        pad = (-len(scan)) % 16
        if pad:
            chunks.append(np.zeros(pad, dtype=np.uint8))
        data_off += len(scan) + pad
    tabs = (JpegTables * len(tab_list))(*tab_list)
    sg = (JpegSegment * len(segs))()
    for k, (off, coff, i, m0, nm, raw, cap, sub) in enumerate(segs):
        g = sg[k]
        g.byte_off, g.clean_off, g.image, g.mcu0, g.nmcu, g.raw_len, g.clean_cap, g.sub_bits = off, coff, i, m0, nm, raw, cap, sub
    data = np.concatenate(chunks) if chunks else np.zeros(16, dtype=np.uint8)
    return data, imgs, tabs, sg, sizes, dict(coef_elems=coef, plane_bytes=plane, rgb_bytes=rgb, clean_bytes=clean, max_blocks=max_blocks,
                                             max_pixels=max_pixels)
