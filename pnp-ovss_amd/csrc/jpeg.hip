// Baseline JPEG decode on gfx950 -- the last host-side piece of the input path: the reference decodes
// with Pillow on the main thread (Dataset.py:349-445 `Image.open(...).convert('RGB')`, PnP_OVSS_0514_updated_
// segmentation.py:929-955 load_OrgImage; DataLoader num_workers = 0).  Pillow's arithmetic is libjpeg(-turbo)'s defaults,
// an un-vendored library, restated from its published algorithm (oracle/jpeg_np.py, pinned bit-exact against Pillow):
//   jpeg_huffman_kernel  sequential Huffman entropy decode (ITU T.81 F.2.2), ONE WAVE PER RESTART SEGMENT of an image:
//                        the bit stream is inherently serial, the parallelism is the batch (35 images per step) --
//                        byte-stuffed stream staged through an LDS ring by all 64 lanes, 8-bit look-ahead code tables
//                        in LDS, every lane runs the same (uniform) decode, lane 0 stores quantised coefficients
//   jpeg_idct_kernel     one thread per 8 x 8 block: dequantise + jidctint.c "islow" (13-bit constants, DESCALE)
//   jpeg_color_kernel    one thread per pixel: jdsample.c "fancy" h2v1 / h2v2 chroma upsampling + jdcolor.c fixed-point
//                        YCbCr -> RGB, written as the concatenated HWC uint8 buffer the resize and the CRF read
// Integer / byte work, HBM-trivial (a 500 x 375 image: ~100 KB in, 0.56 MB out); the entropy decode is latency-bound
// (~100 clk per symbol, a few ms per image) and overlaps the previous batch's model work on the prefetch stream.
#include "common.h"
#include "kernels.h"

namespace pnp {

__constant__ int kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

constexpr int JPEG_RING = 16384;      // LDS bytes of stream staged at a time (two halves of 8 KB)

// ------------------------------------------------------------------------------------------ entropy decode
__global__ __launch_bounds__(64) void jpeg_huffman_kernel(const uint8_t* __restrict__ data, const JpegImage* __restrict__ imgs,
                                                          const JpegTables* __restrict__ tabs, const JpegSegment* __restrict__ segs,
                                                          int16_t* __restrict__ coef, int* __restrict__ err) {
    __shared__ __attribute__((aligned(16))) uint8_t ring[JPEG_RING];
    __shared__ uint16_t fast[4][256];
    __shared__ int mincode[4][17], maxcode[4][17], valptr[4][17];
    __shared__ uint8_t vals[4][256];
    __shared__ int zz[64];
    const JpegSegment sg = segs[blockIdx.x];
    const JpegImage* const ip = imgs + sg.image;        // per-component fields are read through memory (runtime index)
    const JpegImage im = *ip;
    const JpegTables& T = tabs[im.tab];
    const int lane = threadIdx.x;
    for (int i = lane; i < 4 * 256; i += 64) {
        (&fast[0][0])[i] = (&T.fast[0][0])[i];
        (&vals[0][0])[i] = (&T.vals[0][0])[i];
    }
    for (int i = lane; i < 4 * 17; i += 64) {
        (&mincode[0][0])[i] = (&T.mincode[0][0])[i];
        (&maxcode[0][0])[i] = (&T.maxcode[0][0])[i];
        (&valptr[0][0])[i] = (&T.valptr[0][0])[i];
    }
    zz[lane] = kZigzag[lane];
    const uint8_t* src = data + im.data_off;
    const long end = im.data_len;
    long fill = sg.byte_off & ~(long)15;       // stream bytes [fill - JPEG_RING, fill) are in the ring (position p at p % RING)
    // cooperative refill of one 8 KB half starting at stream offset `from` (16-byte aligned)
    auto refill = [&](long from) {
#pragma unroll
        for (int i = 0; i < JPEG_RING / 2 / 16 / 64; i++) {
            const long o = from + (long)(i * 64 + lane) * 16;
            chunk16 v = {0u, 0u, 0u, 0u};
            if (o + 16 <= end) {
                v = *reinterpret_cast<const chunk16*>(src + o);           // data_off is 16-byte aligned by the host
            } else if (o < end) {
                uint8_t tmp[16];
                for (int b = 0; b < 16; b++) tmp[b] = o + b < end ? src[o + b] : 0;
                v = *reinterpret_cast<const chunk16*>(tmp);
            }
            *reinterpret_cast<chunk16*>(ring + (o & (JPEG_RING - 1))) = v;
        }
    };
    refill(fill);
    refill(fill + JPEG_RING / 2);
    fill += JPEG_RING;
    __syncthreads();

    long p = sg.byte_off;                       // next stream byte to enter the bit buffer
    uint64_t acc = 0;                           // valid bits at the top
    int nbits = 0;
    bool hit_marker = false;
    auto fill_bits = [&]() {                    // keep >= 32 valid bits (zeros past a marker / the end, like libjpeg)
        while (nbits <= 56) {
            if (p >= fill - JPEG_RING / 2) {    // the read position has left the older half of the ring: recycle that half
                __syncthreads();
                refill(fill);
                fill += JPEG_RING / 2;
                __syncthreads();
            }
            unsigned b = 0;
            if (!hit_marker && p < end) {
                b = ring[p & (JPEG_RING - 1)];
                if (b == 0xFF) {
                    const unsigned n = p + 1 < end ? ring[(p + 1) & (JPEG_RING - 1)] : 0xD9;
                    if (n == 0) p += 2;
                    else { hit_marker = true; b = 0; }
                } else {
                    p += 1;
                }
            }
            acc |= (uint64_t)b << (56 - nbits);
            nbits += 8;
        }
    };
    auto take = [&](int n) -> unsigned {        // n <= 16 bits off the top
        const unsigned v = n ? (unsigned)(acc >> (64 - n)) : 0u;
        acc <<= n;
        nbits -= n;
        return v;
    };
    auto symbol = [&](int t) -> int {
        if (nbits < 32) fill_bits();
        const unsigned f = fast[t][acc >> 56];
        if (f >> 8) {
            take(f >> 8);
            return f & 255;
        }
        int code = (int)(acc >> 55);            // 9 bits
        for (int l = 9; l <= 16; l++) {
            if (maxcode[t][l] >= 0 && code <= maxcode[t][l] && code >= mincode[t][l]) {
                take(l);
                return vals[t][valptr[t][l] + code - mincode[t][l]];
            }
            code = (int)(acc >> (64 - l - 1));
        }
        if (lane == 0) atomicExch(err, 1);
        take(16);
        return 0;
    };
    auto extend = [](int v, int t) { return (t == 0 || v >= (1 << (t - 1))) ? v : v - (1 << t) + 1; };

    int pred0 = 0, pred1 = 0, pred2 = 0;          // DC predictors (named: a runtime-indexed array would live in scratch)
    for (int m = sg.mcu0; m < sg.mcu0 + sg.nmcu; m++) {
        const int my = m / im.mcux, mx = m - my * im.mcux;
        for (int c = 0; c < im.ncomp; c++) {
            const int tdc = ip->td[c], tac = 2 + ip->ta[c], vc = ip->v[c], hc = ip->h[c], bxc = ip->bx[c];
            const long coff = ip->coef_off[c];
            for (int by = 0; by < vc; by++)
                for (int bx = 0; bx < hc; bx++) {
                    int16_t* blk = coef + coff + ((long)(my * vc + by) * bxc + (mx * hc + bx)) * 64;
                    const int t = symbol(tdc);
                    if (nbits < 32) fill_bits();
                    const int diff = extend((int)take(t), t);
                    int pc;
                    if (c == 0) pc = (pred0 += diff);
                    else if (c == 1) pc = (pred1 += diff);
                    else pc = (pred2 += diff);
                    if (lane == 0) blk[0] = (int16_t)pc;
                    int k = 1;
                    while (k < 64) {
                        const int rs = symbol(tac);
                        const int r = rs >> 4, s = rs & 15;
                        if (s == 0) {
                            if (r != 15) break;
                            k += 16;
                            continue;
                        }
                        k += r;
                        if (nbits < 32) fill_bits();
                        const int v = extend((int)take(s), s);
                        if (k > 63) {
                            if (lane == 0) atomicExch(err, 1);
                            break;
                        }
                        if (lane == 0) blk[zz[k]] = (int16_t)v;
                        k++;
                    }
                }
        }
    }
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
                int n_segments, int16_t* d_coef, size_t coef_elems, uint8_t* d_planes, uint8_t* d_rgb, int max_blocks, int max_pixels,
                int* d_err, hipStream_t s) {
    if (n_images <= 0 || n_segments < n_images) return PNP_ERR_ARG;
    if (hipMemsetAsync(d_coef, 0, coef_elems * sizeof(int16_t), s) != hipSuccess) return PNP_ERR_HIP;
    hipLaunchKernelGGL(jpeg_huffman_kernel, dim3(n_segments), dim3(64), 0, s, d_data, d_imgs, d_tabs, d_segs, d_coef, d_err);
    const int nb = (max_blocks + 255) / 256;
    hipLaunchKernelGGL(jpeg_idct_kernel, dim3(nb < 1 ? 1 : nb, n_images), dim3(256), 0, s, d_imgs, d_tabs, d_coef, d_planes);
    int np = (max_pixels + 255) / 256;
    np = np > 1024 ? 1024 : (np < 1 ? 1 : np);
    hipLaunchKernelGGL(jpeg_color_kernel, dim3(np, n_images), dim3(256), 0, s, d_imgs, d_planes, d_rgb);
    return hipGetLastError() == hipSuccess ? PNP_OK : PNP_ERR_HIP;
}

}  // namespace pnp
