// Baseline JPEG decode on gfx950 -- the last host-side piece of the input path: the reference decodes
// with Pillow on the main thread (Dataset.py:349-445 `Image.open(...).convert('RGB')`, PnP_OVSS_0514_updated_
// segmentation.py:929-955 load_OrgImage; DataLoader num_workers = 0).  Pillow's arithmetic is libjpeg(-turbo)'s defaults,
// an un-vendored library, restated from its published algorithm (oracle/jpeg_np.py, pinned bit-exact against Pillow):
//   jpeg_unstuff_kernel  one workgroup per restart segment: drops the 0x00 that follows every 0xFF of the entropy-coded
//                        bytes (block-wide compaction), so the decoder reads plain big-endian words
//   jpeg_huffman_kernel  Huffman entropy decode (ITU T.81 F.2.2), one workgroup of 1024 threads per restart segment.  The
//                        bit stream is serial only in its decoder STATE (bit position, block-in-MCU, zig-zag index), and
//                        Huffman streams self-synchronise: the segment is cut into up to 1024 sub-sequences of equal
//                        bit length, every thread decodes its own from a guessed state, then repeatedly takes over the
//                        exit state of its predecessor and re-decodes until no entry state changes (a fixed point of
//                        S[i+1] = f(S[i]) with S[0] known is the sequential decode; typically 2-4 rounds).  Block counts and
//                        DC differences per sub-sequence are prefix-summed, and a last pass writes the coefficients.
//                        Look-ahead tables (10 bits) are built in LDS from the file's own DHT bytes.
//   jpeg_idct_kernel     one thread per 8 x 8 block: dequantise + jidctint.c "islow" (13-bit constants, DESCALE)
//   jpeg_color_kernel    one thread per pixel: jdsample.c "fancy" h2v1 / h2v2 chroma upsampling + jdcolor.c fixed-point
//                        YCbCr -> RGB, written as the concatenated HWC uint8 buffer the resize and the CRF read
// Integer / byte work, HBM-trivial (a 500 x 375 image: ~85 KB in, 0.56 MB out).
#include "common.h"
#include "kernels.h"

namespace pnp {

__constant__ int kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

constexpr int JPEG_T = 1024;          // threads = sub-sequences per restart segment
constexpr int JPEG_LA = 10;           // look-ahead bits of the code tables
constexpr int JPEG_MAXB = 6;          // blocks per MCU (4:2:0)

// exclusive prefix sum over the workgroup (tid order); wsum: one int per wave
__device__ __forceinline__ int block_excl_scan(int v, int* wsum, int* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    int x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_up(x, d);
        if (lane >= d) x += y;
    }
    __syncthreads();                                      // wsum may still be read from the previous scan
    if (lane == 63) wsum[w] = x;
    __syncthreads();
    int off = 0, tot = 0;
    for (int i = 0; i < nw; i++) {
        if (i < w) off += wsum[i];
        tot += wsum[i];
    }
    if (total) *total = tot;
    return off + x - v;
}

// ------------------------------------------------------------------------------------------ byte un-stuffing
__global__ __launch_bounds__(256) void jpeg_unstuff_kernel(const uint8_t* __restrict__ data, const JpegImage* __restrict__ imgs,
                                                           const JpegSegment* __restrict__ segs, uint8_t* __restrict__ clean,
                                                           int* __restrict__ seg_bits) {
    __shared__ int wsum[4];
    const JpegSegment sg = segs[blockIdx.x];
    const uint8_t* src = data + imgs[sg.image].data_off + sg.byte_off;
    uint8_t* dst = clean + sg.clean_off;
    const int n = sg.raw_len, tid = threadIdx.x;
    int base = 0;
    for (int c0 = 0; c0 < n; c0 += 256 * 16) {
        const int o = c0 + tid * 16;
        uint8_t b[16];
        unsigned keep = 0;
        int prev = (o > 0 && o <= n) ? src[o - 1] : 0;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const bool in = o + j < n;
            b[j] = in ? src[o + j] : 0;
            if (in && !(b[j] == 0 && prev == 0xFF)) keep |= 1u << j;
            prev = b[j];
        }
        int total;
        int w = base + block_excl_scan(__popc(keep), wsum, &total);
#pragma unroll
        for (int j = 0; j < 16; j++)
            if (keep >> j & 1) dst[w++] = b[j];
        base += total;
    }
    // zeros behind the data (what libjpeg feeds after the end), up to the capacity the host reserved
    for (int i = base + tid; i < sg.clean_cap; i += 256) dst[i] = 0;
    if (tid == 0) seg_bits[blockIdx.x] = base * 8;
}

// ------------------------------------------------------------------------------------------ entropy decode
struct JpegLds {
    uint16_t fast[4][1 << JPEG_LA];
    int mincode[4][17], maxcode[4][17], valptr[4][17];
    uint8_t vals[4][256];
    int zz[64];
    int mb_info[JPEG_MAXB], mb_by[JPEG_MAXB], mb_bx[JPEG_MAXB];      // per block of the MCU: component | DC table << 2 | AC table << 4
    int g_h[3], g_v[3], g_bxc[3];
    long g_coef[3];
    unsigned Ep[JPEG_T];
    uint16_t Ebk[JPEG_T];
    int Ecnt[JPEG_T], Edc[3][JPEG_T];
    int wsum[JPEG_T / 64];
    int done;
};

struct JpegGeom {            // uniform per workgroup
    int bpm, mcux, mcu0, total_blocks;
};

__device__ __forceinline__ uint32_t jpeg_word(const uint32_t* __restrict__ w, uint32_t i) { return __builtin_bswap32(w[i]); }

// Decode from state (p, b, k) up to the first code boundary at or behind p_end.  WRITE = false: only the exit state, the
// number of blocks completed and the sum of DC differences per component (the synchronisation rounds).  WRITE = true: n is the
// index of the block in progress, pred the DC predictors at the entry; coefficients go to global memory.
template <bool WRITE>
__device__ __forceinline__ void jpeg_run(const JpegLds& S, const JpegGeom& G, const uint32_t* __restrict__ words, unsigned& p, int& b,
                                         int& k, unsigned p_end, int& cnt, int& dc0, int& dc1, int& dc2, int n, int16_t* __restrict__ coef,
                                         int* __restrict__ err, int* done) {
    uint32_t wp = p >> 5;
    uint64_t acc = ((uint64_t)jpeg_word(words, wp) << 32) | jpeg_word(words, wp + 1);
    const int sh = p & 31;
    acc <<= sh;
    int nb = 64 - sh;
    wp += 2;
    uint32_t nxt = jpeg_word(words, wp);                 // one word ahead of the bit buffer (hides the load latency)
    int16_t* blk = nullptr;
    auto block_ptr = [&](int nblk, int bi) -> int16_t* {
        const int mcu = G.mcu0 + nblk / G.bpm;
        const int my = mcu / G.mcux, mx = mcu - my * G.mcux;
        const int c = S.mb_info[bi] & 3;
        return coef + S.g_coef[c] + ((long)(my * S.g_v[c] + S.mb_by[bi]) * S.g_bxc[c] + (mx * S.g_h[c] + S.mb_bx[bi])) * 64;
    };
    if (WRITE) blk = block_ptr(n, b);
    int info = S.mb_info[b];
    while (p < p_end) {
        if (nb < 32) {
            acc |= (uint64_t)nxt << (32 - nb);
            nb += 32;
            wp++;
            nxt = jpeg_word(words, wp);
        }
        const int c = info & 3;
        const int t = k == 0 ? (info >> 2) & 1 : 2 + ((info >> 4) & 1);
        const unsigned top = (unsigned)(acc >> 48);
        const unsigned f = S.fast[t][top >> (16 - JPEG_LA)];
        int len = f >> 8, sym = f & 255;
        if (len == 0) {                                  // code longer than the look-ahead: T.81 F.2.2.3 scan
            len = 16;
            sym = 0;
            bool found = false;
            for (int l = JPEG_LA + 1; l <= 16; l++) {
                const int code = (int)(top >> (16 - l));
                if (S.maxcode[t][l] >= 0 && code <= S.maxcode[t][l] && code >= S.mincode[t][l]) {
                    len = l;
                    sym = S.vals[t][(S.valptr[t][l] + code - S.mincode[t][l]) & 255];
                    found = true;
                    break;
                }
            }
            if (WRITE && !found) atomicExch(err, 1);
        }
        const int s = sym & 15, r = k == 0 ? 0 : sym >> 4;
        if (WRITE && k == 0 && sym > 11) atomicExch(err, 1);
        acc <<= len;
        const unsigned vb = s ? (unsigned)(acc >> (64 - s)) : 0u;
        acc <<= s;
        nb -= len + s;
        p += len + s;
        const int v = (s == 0 || vb >= (1u << (s - 1))) ? (int)vb : (int)vb - (1 << s) + 1;
        if (k == 0) {
            dc0 += c == 0 ? v : 0;                       // (selects, not an indexed array: that would live in scratch)
            dc1 += c == 1 ? v : 0;
            dc2 += c == 2 ? v : 0;
            if (WRITE) blk[0] = (int16_t)(c == 0 ? dc0 : (c == 1 ? dc1 : dc2));
            k = 1;
        } else if (s == 0) {
            k = r == 15 ? k + 16 : 64;
        } else {
            k += r;
            if (k > 63) {
                if (WRITE) atomicExch(err, 1);
            } else if (WRITE) {
                blk[S.zz[k]] = (int16_t)v;
            }
            k++;
        }
        if (k >= 64) {
            k = 0;
            b = b + 1 == G.bpm ? 0 : b + 1;
            info = S.mb_info[b];
            cnt++;
            if (WRITE) {
                n++;
                if (n >= G.total_blocks) {
                    *done = 1;
                    break;
                }
                blk = block_ptr(n, b);
            }
        }
    }
}

__global__ __launch_bounds__(JPEG_T) void jpeg_huffman_kernel(const uint8_t* __restrict__ clean, const JpegImage* __restrict__ imgs,
                                                              const JpegTables* __restrict__ tabs, const JpegSegment* __restrict__ segs,
                                                              const int* __restrict__ seg_bits, int16_t* __restrict__ coef,
                                                              int* __restrict__ err) {
    __shared__ JpegLds S;
    const JpegSegment sg = segs[blockIdx.x];
    const JpegImage* const ip = imgs + sg.image;
    const JpegTables& T = tabs[ip->tab];
    const int tid = threadIdx.x;
    // canonical code ranges (one thread per table), then the look-ahead entries (all threads)
    if (tid < 4) {
        int code = 0, kk = 0;
        for (int l = 1; l <= 16; l++) {
            const int cnt = T.counts[tid][l - 1];
            S.valptr[tid][l] = kk;
            S.mincode[tid][l] = code;
            code += cnt;
            kk += cnt;
            S.maxcode[tid][l] = cnt ? code - 1 : -1;
            code <<= 1;
        }
    }
    for (int i = tid; i < 4 * 256; i += JPEG_T) (&S.vals[0][0])[i] = (&T.vals[0][0])[i];
    if (tid < 64) S.zz[tid] = kZigzag[tid];
    JpegGeom G;
    G.mcux = ip->mcux;
    G.mcu0 = sg.mcu0;
    const int ncomp = ip->ncomp;
    int bpm = 0;
    for (int c = 0; c < ncomp && c < 3; c++) {
        const int hc = ip->h[c], vc = ip->v[c];
        if (tid == 0) {
            S.g_h[c] = hc;
            S.g_v[c] = vc;
            S.g_bxc[c] = ip->bx[c];
            S.g_coef[c] = ip->coef_off[c];
            for (int y = 0; y < vc; y++)
                for (int x = 0; x < hc; x++) {
                    const int bi = bpm + y * hc + x;
                    if (bi < JPEG_MAXB) {
                        S.mb_info[bi] = c | (ip->td[c] & 1) << 2 | (ip->ta[c] & 1) << 4;
                        S.mb_by[bi] = y;
                        S.mb_bx[bi] = x;
                    }
                }
        }
        bpm += hc * vc;
    }
    G.bpm = bpm;
    G.total_blocks = sg.nmcu * bpm;
    if (bpm > JPEG_MAXB || ncomp > 3) {                   // the host refuses such files; never index past the per-block tables
        if (tid == 0) atomicExch(err, 1);
        return;
    }
    __syncthreads();
    for (int i = tid; i < 4 << JPEG_LA; i += JPEG_T) {
        const int t = i >> JPEG_LA, x = i & ((1 << JPEG_LA) - 1);
        unsigned e = 0;
        for (int l = 1; l <= JPEG_LA; l++) {
            const int code = x >> (JPEG_LA - l);
            if (S.maxcode[t][l] >= 0 && code <= S.maxcode[t][l] && code >= S.mincode[t][l]) {
                e = ((unsigned)l << 8) | S.vals[t][(S.valptr[t][l] + code - S.mincode[t][l]) & 255];
                break;
            }
        }
        S.fast[t][x] = (uint16_t)e;
    }
    if (tid == 0) S.done = 0;
    __syncthreads();

    const uint32_t* words = reinterpret_cast<const uint32_t*>(clean + sg.clean_off);        // clean_off is 16-byte aligned
    const unsigned total_bits = (unsigned)seg_bits[blockIdx.x];
    const unsigned sub = (unsigned)sg.sub_bits;                                             // multiple of 32, >= total_bits / JPEG_T
    const int nsub = (int)((total_bits + sub - 1) / sub);
    const bool live = tid < nsub;
    const unsigned p_end = live ? min((unsigned)(tid + 1) * sub, total_bits) : 0u;
    // entry state of this sub-sequence: exact for thread 0, a guess (block start at the cut) for the others
    unsigned sp = (unsigned)tid * sub;
    int sb = 0, sk = 0;
    bool dirty = live;
    for (int round = 0; round <= JPEG_T; round++) {
        if (dirty) {
            unsigned p = sp;
            int b = sb, k = sk, cnt = 0, d0 = 0, d1 = 0, d2 = 0;
            jpeg_run<false>(S, G, words, p, b, k, p_end, cnt, d0, d1, d2, 0, nullptr, nullptr, nullptr);
            S.Ep[tid] = p;
            S.Ebk[tid] = (uint16_t)(b * 64 + k);
            S.Ecnt[tid] = cnt;
            S.Edc[0][tid] = d0;
            S.Edc[1][tid] = d1;
            S.Edc[2][tid] = d2;
        }
        __syncthreads();
        dirty = false;
        if (live && tid > 0) {
            const unsigned np = S.Ep[tid - 1];
            const int nbk = S.Ebk[tid - 1];
            if (np != sp || nbk != sb * 64 + sk) {
                sp = np;
                sb = nbk >> 6;
                sk = nbk & 63;
                dirty = true;
            }
        }
        if (!__syncthreads_or(dirty ? 1 : 0)) break;
    }
    // absolute block index and DC predictors at every entry state, then the writing pass
    const int n0 = block_excl_scan(live ? S.Ecnt[tid] : 0, S.wsum, nullptr);
    int p0 = block_excl_scan(live ? S.Edc[0][tid] : 0, S.wsum, nullptr);
    int p1 = block_excl_scan(live ? S.Edc[1][tid] : 0, S.wsum, nullptr);
    int p2 = block_excl_scan(live ? S.Edc[2][tid] : 0, S.wsum, nullptr);
    if (live && n0 < G.total_blocks && sp < total_bits) {
        unsigned p = sp;
        int b = sb, k = sk, cnt = 0;
        if (b != n0 % G.bpm) atomicExch(err, 1);
        jpeg_run<true>(S, G, words, p, b, k, p_end, cnt, p0, p1, p2, n0, coef, err, &S.done);
    }
    __syncthreads();
    if (tid == 0 && !S.done && G.total_blocks > 0) atomicExch(err, 1);       // the stream ended before the last block
}

// ------------------------------------------------------------------------------------------ inverse DCT
__device__ __forceinline__ void idct_islow_1d(int x0, int x1, int x2, int x3, int x4, int x5, int x6, int x7, int shift, int* o, int stride) {
    int z1 = (x2 + x6) * 4433;
    const int tmp2 = z1 + x6 * (-15137);
    const int tmp3 = z1 + x2 * 6270;
    const int tmp0 = (x0 + x4) << 13, tmp1 = (x0 - x4) << 13;
    const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    int t0 = x7, t1 = x5, t2 = x3, t3 = x1;
    z1 = t0 + t3;
    int z2 = t1 + t2, z3 = t0 + t2, z4 = t1 + t3;
    const int z5 = (z3 + z4) * 9633;
    t0 *= 2446; t1 *= 16819; t2 *= 25172; t3 *= 12299;
    z1 *= -7373; z2 *= -20995; z3 = z3 * (-16069) + z5; z4 = z4 * (-3196) + z5;
    t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4;
    const int r = 1 << (shift - 1);
    o[0 * stride] = (tmp10 + t3 + r) >> shift;
    o[7 * stride] = (tmp10 - t3 + r) >> shift;
    o[1 * stride] = (tmp11 + t2 + r) >> shift;
    o[6 * stride] = (tmp11 - t2 + r) >> shift;
    o[2 * stride] = (tmp12 + t1 + r) >> shift;
    o[5 * stride] = (tmp12 - t1 + r) >> shift;
    o[3 * stride] = (tmp13 + t0 + r) >> shift;
    o[4 * stride] = (tmp13 - t0 + r) >> shift;
}

__global__ __launch_bounds__(256) void jpeg_idct_kernel(const JpegImage* __restrict__ imgs, const JpegTables* __restrict__ tabs,
                                                        const int16_t* __restrict__ coef, uint8_t* __restrict__ planes) {
    const JpegImage* const ip = imgs + blockIdx.y;
    const int ncomp = ip->ncomp;
    const int n0 = ip->bx[0] * ip->by[0], n1 = ncomp > 1 ? ip->bx[1] * ip->by[1] : 0, n2 = ncomp > 2 ? ip->bx[2] * ip->by[2] : 0;
    const int total = n0 + n1 + n2;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        const int c = t < n0 ? 0 : (t < n0 + n1 ? 1 : 2);
        const int b = t - (c == 0 ? 0 : (c == 1 ? n0 : n0 + n1));
        const int* qc = tabs[ip->tab].quant[ip->tq[c]];
        const int bxc = ip->bx[c];
        const int16_t* in = coef + ip->coef_off[c] + (long)b * 64;
        int ws[64];
#pragma unroll
        for (int col = 0; col < 8; col++) {
            int x[8];
#pragma unroll
            for (int r = 0; r < 8; r++) x[r] = (int)in[r * 8 + col] * qc[r * 8 + col];
            idct_islow_1d(x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7], 11, ws + col, 8);
        }
        const int by = b / bxc, bx = b - by * bxc;
        uint8_t* out = planes + ip->plane_off[c] + ((long)by * 8) * (bxc * 8) + bx * 8;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            int o[8];
            idct_islow_1d(ws[r * 8], ws[r * 8 + 1], ws[r * 8 + 2], ws[r * 8 + 3], ws[r * 8 + 4], ws[r * 8 + 5], ws[r * 8 + 6], ws[r * 8 + 7], 18, o, 1);
            chunk8 pk;
            unsigned lo = 0, hi = 0;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                int a = o[e] + 128, d = o[4 + e] + 128;
                a = a < 0 ? 0 : (a > 255 ? 255 : a);
                d = d < 0 ? 0 : (d > 255 ? 255 : d);
                lo |= (unsigned)a << (8 * e);
                hi |= (unsigned)d << (8 * e);
            }
            pk[0] = lo;
            pk[1] = hi;
            *reinterpret_cast<chunk8*>(out + (long)r * (bxc * 8)) = pk;
        }
    }
}

// ------------------------------------------------------------------------------------------ upsample + colour
__global__ __launch_bounds__(256) void jpeg_color_kernel(const JpegImage* __restrict__ imgs, const uint8_t* __restrict__ planes,
                                                         uint8_t* __restrict__ rgb) {
    const JpegImage im = imgs[blockIdx.y];
    const int H = im.H, W = im.W;
    const uint8_t* Y = planes + im.plane_off[0];
    const int ys = im.bx[0] * 8;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < H * W; t += gridDim.x * blockDim.x) {
        const int y = t / W, x = t - y * W;
        const int yv = Y[(long)y * ys + x];
        uint8_t* o = rgb + im.rgb_off + (long)t * 3;
        if (im.ncomp == 1) {
            o[0] = o[1] = o[2] = (uint8_t)yv;
            continue;
        }
        int cc[2];
        const int cs = im.bx[1] * 8;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const uint8_t* P = planes + im.plane_off[1 + k];
            if (im.hmax == 1) {
                cc[k] = P[(long)y * cs + x];
            } else if (im.vmax == 1) {          // h2v1 fancy
                const int w = (W + 1) >> 1, cx = x >> 1;
                const int cur = P[(long)y * cs + cx];
                if (x & 1) cc[k] = cx == w - 1 ? cur : (3 * cur + P[(long)y * cs + cx + 1] + 2) >> 2;
                else cc[k] = cx == 0 ? cur : (3 * cur + P[(long)y * cs + cx - 1] + 1) >> 2;
            } else {                            // h2v2 fancy
                const int w = (W + 1) >> 1, h = (H + 1) >> 1, cx = x >> 1, cy = y >> 1;
                int far = (y & 1) ? cy + 1 : cy - 1;
                far = far < 0 ? 0 : (far > h - 1 ? h - 1 : far);
                const uint8_t* rn = P + (long)cy * cs;
                const uint8_t* rf = P + (long)far * cs;
                const int cur = 3 * rn[cx] + rf[cx];
                if (x & 1) cc[k] = cx == w - 1 ? (cur * 4 + 7) >> 4 : (3 * cur + 3 * rn[cx + 1] + rf[cx + 1] + 7) >> 4;
                else cc[k] = cx == 0 ? (cur * 4 + 8) >> 4 : (3 * cur + 3 * rn[cx - 1] + rf[cx - 1] + 8) >> 4;
            }
        }
        const int cb = cc[0] - 128, cr = cc[1] - 128;
        int r = yv + ((91881 * cr + 32768) >> 16);
        int g = yv + ((-22554 * cb + 32768 - 46802 * cr) >> 16);
        int b = yv + ((116130 * cb + 32768) >> 16);
        o[0] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
        o[1] = (uint8_t)(g < 0 ? 0 : (g > 255 ? 255 : g));
        o[2] = (uint8_t)(b < 0 ? 0 : (b > 255 ? 255 : b));
    }
}

// ------------------------------------------------------------------------------------------ host
int jpeg_decode(const uint8_t* d_data, const JpegImage* d_imgs, const JpegTables* d_tabs, const JpegSegment* d_segs, int n_images,
                int n_segments, uint8_t* d_clean, int* d_seg_bits, int16_t* d_coef, size_t coef_elems, uint8_t* d_planes, uint8_t* d_rgb,
                int max_blocks, int max_pixels, int* d_err, hipStream_t s) {
    if (n_images <= 0 || n_segments < n_images) return PNP_ERR_ARG;
    if (hipMemsetAsync(d_coef, 0, coef_elems * sizeof(int16_t), s) != hipSuccess) return PNP_ERR_HIP;
    hipLaunchKernelGGL(jpeg_unstuff_kernel, dim3(n_segments), dim3(256), 0, s, d_data, d_imgs, d_segs, d_clean, d_seg_bits);
    hipLaunchKernelGGL(jpeg_huffman_kernel, dim3(n_segments), dim3(JPEG_T), 0, s, d_clean, d_imgs, d_tabs, d_segs, d_seg_bits, d_coef, d_err);
    const int nb = (max_blocks + 255) / 256;
    hipLaunchKernelGGL(jpeg_idct_kernel, dim3(nb < 1 ? 1 : nb, n_images), dim3(256), 0, s, d_imgs, d_tabs, d_coef, d_planes);
    int np = (max_pixels + 255) / 256;
    np = np > 1024 ? 1024 : (np < 1 ? 1 : np);
    hipLaunchKernelGGL(jpeg_color_kernel, dim3(np, n_images), dim3(256), 0, s, d_imgs, d_planes, d_rgb);
    return hipGetLastError() == hipSuccess ? PNP_OK : PNP_ERR_HIP;
}

}  // namespace pnp
