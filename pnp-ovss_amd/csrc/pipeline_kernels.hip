// GradCAM gather, the device-resident salience-drop bookkeeping, word-piece merge, and the
// HBM-bound post-process kernels (threshold + bilinear upsample, Scale_0_1, background,
// separable Gaussian blur + min-max, softmax/unary, argmax + remap + confusion histogram).
// Reference (PnP.py = PnP_OVSS_0514_updated_segmentation.py):
//   gather      blip_image_text_matching.py:411-435
//   drop loop   PnP.py:587-647, 716-721
//   merge       PnP.py:810-853
//   threshold / upsample / Scale_0_1 / background   PnP.py:348-379, 424-455, 1078-1094
//   blur        PnP.py:1149-1153 (scipy.ndimage.gaussian_filter: reflect, truncate 4, double acc)
//   remap/hist  PnP.py:390-399, 1106-1146
#include <stdlib.h>

#include "common.h"
#include "kernels.h"
#include "../../include/pnp_math.h"

namespace pnp {

// ------------------------------------------------------------------------------------------
// G[b, t, p] = max(P[b,h,t+1,1+p] * relu(dP[b,h,t+1,1+p]) * mask[b,t+1], 0)
__global__ void gradcam_gather_kernel(const float* __restrict__ P, const float* __restrict__ dP,
                                      const int64_t* __restrict__ mask, int ld_mask, float* __restrict__ out, int B,
                                      int nheads, int head, int L, int Nst, int PP) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int T = L - 1;
    if (idx >= B * T * PP) return;
    const int p = idx % PP, t = (idx / PP) % T, b = idx / (PP * T);
    const size_t o = (((size_t)b * nheads + head) * L + (t + 1)) * Nst + 1 + p;
    const float g = P[o] * fmaxf(dP[o], 0.f) * (float)mask[(size_t)b * ld_mask + t + 1];
    out[idx] = g < 0.f ? 0.f : g;
}

// ------------------------------------------------------------------------------------------
// One salience-drop iteration for one image (one workgroup per image), fully on device:
//   pred = G with already-picked cells zeroed  -> g0 (iter 0) and agg (l0 + l0 + l1 + ...)
//   salience = sum_{t=3}^{T-2} G[t]  (un-zeroed copy), picked cells -> 0, take the `npick` largest
//   (ties: larger index wins, i.e. a stable ascending argsort's tail), append to picks and flag
//   them in `dropped` for the next forward's patch gather.
__device__ __forceinline__ uint64_t sal_key(float v, int idx) {
    uint32_t u = __float_as_uint(v);
    u = (v != v) ? 0xffffffffu : ((u & 0x80000000u) ? ~u : (u | 0x80000000u));   // order-preserving, NaN largest
    return ((uint64_t)u << 32) | (uint32_t)idx;
}

// the element-wise half of the step, over the whole (B, T, P^2) array: g0 (first iteration) and the running aggregate take the
// map with the already-dropped cells zeroed.  Runs before drop_step_kernel of the same iteration (which flags the new picks).
__global__ __launch_bounds__(256) void drop_accumulate_kernel(const float* __restrict__ G, float* __restrict__ g0,
                                                              float* __restrict__ agg, const uint8_t* __restrict__ dropped,
                                                              int iter, int T, int PP) {
    const int b = blockIdx.y;
    const uint8_t* dr = dropped + (size_t)b * PP;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < T * PP; i += gridDim.x * 256) {
        const int p = i % PP;
        const size_t o = (size_t)b * T * PP + i;
        const float v = dr[p] ? 0.f : G[o];
        if (iter == 0) {
            if (g0) g0[o] = v;
            agg[o] = v + v;
        } else {
            agg[o] += v;
        }
    }
}

__global__ __launch_bounds__(256) void drop_step_kernel(const float* __restrict__ G, uint8_t* __restrict__ dropped,
                                                        int32_t* __restrict__ picks, int iter, int T, int PP,
                                                        int npick, int max_picks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sal = reinterpret_cast<float*>(smem);                 // [PP]
    __shared__ uint64_t red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* Gb = G + (size_t)b * T * PP;
    uint8_t* dr = dropped + (size_t)b * PP;
    for (int p = tid; p < PP; p += 256) {
        float s = 0.f;
        int t = 3;
        for (; t + 8 <= T - 1; t += 8) {                         // eight plane loads in flight, summed in order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = Gb[(size_t)(t + u) * PP + p];
#pragma unroll
            for (int u = 0; u < 8; u++) s += v[u];
        }
        for (; t < T - 1; t++) s += Gb[(size_t)t * PP + p];
        sal[p] = dr[p] ? 0.f : s;
    }
    __syncthreads();
    if (npick <= 0) return;
    // npick rounds of block-wide arg-max on (value, index) keys; results are emitted in ascending
    // order like np.argsort(...)[-npick:] (order is irrelevant downstream, kept for parity dumps)
    for (int k = 0; k < npick; k++) {
        uint64_t best = 0;
        for (int p = tid; p < PP; p += 256) {
            const uint64_t key = sal_key(sal[p], p);
            const bool taken = (__float_as_uint(sal[p]) == 0xffc00001u);
            if (!taken && key > best) best = key;
        }
        for (int o = 32; o > 0; o >>= 1) {
            const uint64_t other = __shfl_xor(best, o, 64);
            best = other > best ? other : best;
        }
        if ((tid & 63) == 0) red[tid >> 6] = best;
        __syncthreads();
        if (tid == 0) {
            uint64_t m = red[0];
            for (int w = 1; w < 4; w++) m = red[w] > m ? red[w] : m;
            const int p = (int)(m & 0xffffffffu);
            const int slot = iter * npick + (npick - 1 - k);
            if (slot < max_picks) picks[(size_t)b * max_picks + slot] = p;
            sal[p] = __uint_as_float(0xffc00001u);               // tombstone (a NaN payload we never produce)
        }
        __syncthreads();
    }
    for (int k = tid; k < npick; k += 256) {
        const int slot = iter * npick + k;
        if (slot < max_picks) dr[picks[(size_t)b * max_picks + slot]] = 1;
    }
}

// ------------------------------------------------------------------------------------------
// Word-piece -> class merge, table driven (host builds the plan once per caption):
// merged[b, c] = (sum_{k in tokens(c), in order} src[b, 3 + tok_k]) / div(c)   (div applied only if != 1)
__global__ void merge_tokens_kernel(const float* __restrict__ src, const int32_t* __restrict__ cls_off,
                                    const int32_t* __restrict__ tok_idx, const int32_t* __restrict__ cls_div,
                                    const int32_t* __restrict__ img_cls_off, float* __restrict__ out, int T, int PP,
                                    int Cmax) {
    const int b = blockIdx.y, c = blockIdx.x;
    const int c0 = img_cls_off[b], nc = img_cls_off[b + 1] - c0;
    if (c >= nc) return;
    const int k0 = cls_off[c0 + c], k1 = cls_off[c0 + c + 1];
    const int div = cls_div[c0 + c];
    const float* sb = src + (size_t)b * T * PP;
    for (int p = threadIdx.x; p < PP; p += blockDim.x) {
        float acc = 0.f;
        if (k1 > k0) {
            acc = sb[(size_t)(3 + tok_idx[k0]) * PP + p];
            for (int k = k0 + 1; k < k1; k++) acc += sb[(size_t)(3 + tok_idx[k]) * PP + p];
            if (div != 1) acc = acc / (float)div;
        }
        out[((size_t)b * Cmax + c) * PP + p] = acc;
    }
}

// ------------------------------------------------------------------------------------------
// per (image, class): (m - min) / (max - min) >= threshold -> m * mask   (NaN -> mask false)
__global__ __launch_bounds__(256) void threshold_kernel(const float* __restrict__ merged, const int32_t* __restrict__ img_cls_off,
                                                        float* __restrict__ out, float threshold, int PP, int Cmax) {
    const int b = blockIdx.y, c = blockIdx.x, tid = threadIdx.x;
    const int nc = img_cls_off[b + 1] - img_cls_off[b];
    if (c >= nc) return;
    __shared__ float smn[4], smx[4];
    const float* m = merged + ((size_t)b * Cmax + c) * PP;
    float mn = INFINITY, mx = -INFINITY;
    int nan = 0;
    for (int p = tid; p < PP; p += 256) {
        const float v = m[p];
        nan |= (v != v);
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
    for (int o = 32; o > 0; o >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, o, 64));
        mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    }
    if ((tid & 63) == 0) { smn[tid >> 6] = mn; smx[tid >> 6] = mx; }
    nan = __syncthreads_or(nan);
    mn = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
    mx = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
    const float den = nan ? __uint_as_float(0x7fc00000u) : mx - mn;   // torch min()/max() propagate NaN
    for (int p = tid; p < PP; p += 256) {
        const float v = m[p];
        const float nrm = (v - mn) / den;
        out[((size_t)b * Cmax + c) * PP + p] = (nrm >= threshold) ? v : v * 0.0f;   // v*False keeps NaN/inf semantics
    }
}

// Bilinear align_corners=True upsample, torch generic-kernel arithmetic:
// out = fma(top, w0h, bot*w1h), top = fma(v00, w0w, v01*w1w), bot = fma(v10, w0w, v11*w1w).
__device__ __forceinline__ void src_index(int dst, int in_size, float scale, int& i0, int& i1, float& w0, float& w1) {
    const float real = __fmul_rn(scale, (float)dst);
    int f = (int)floorf(real);
    f = f < in_size - 1 ? f : in_size - 1;
    float lam = __fsub_rn(real, (float)f);
    lam = fminf(fmaxf(lam, 0.f), 1.f);
    i0 = f;
    i1 = f + (f < in_size - 1 ? 1 : 0);
    w1 = lam;
    w0 = __fsub_rn(1.f, lam);
}

__global__ void upsample_kernel(const float* __restrict__ src, const PostDesc* __restrict__ desc, float* __restrict__ maps,
                                int P, int Cmax) {
    const int b = blockIdx.z, c = blockIdx.y;
    const PostDesc d = desc[b];
    if (c >= d.C) return;
    const int H = d.H, W = d.W;
    const float sh = H > 1 ? __fdiv_rn((float)(P - 1), (float)(H - 1)) : 0.f;
    const float sw = W > 1 ? __fdiv_rn((float)(P - 1), (float)(W - 1)) : 0.f;
    const float* s = src + ((size_t)b * Cmax + c) * P * P;
    float* o = maps + d.off + (size_t)(c + d.has_bg) * H * W;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < H * W; i += gridDim.x * blockDim.x) {
        const int y = i / W, x = i - y * W;
        int h0, h1, w0i, w1i;
        float a, bb, cc, dd;
        src_index(y, P, sh, h0, h1, a, bb);
        src_index(x, P, sw, w0i, w1i, cc, dd);
        const float v00 = s[h0 * P + w0i], v01 = s[h0 * P + w1i], v10 = s[h1 * P + w0i], v11 = s[h1 * P + w1i];
        const float top = fmaf(v00, cc, __fmul_rn(v01, dd));
        const float bot = fmaf(v10, cc, __fmul_rn(v11, dd));
        o[i] = fmaf(top, a, __fmul_rn(bot, bb));
    }
}

// per (image, channel) min / max over H*W -> stats[(b*Kmax + k)*2 + {0,1}]
__global__ __launch_bounds__(256) void minmax_kernel(const float* __restrict__ maps, const PostDesc* __restrict__ desc,
                                                     float* __restrict__ stats, int Kmax, int first_k_is_class) {
    const int b = blockIdx.y, k = blockIdx.x, tid = threadIdx.x;
    const PostDesc d = desc[b];
    const int kk = first_k_is_class ? k + d.has_bg : k;       // Scale_0_1 touches class channels only
    if (kk >= d.K) return;
    if (first_k_is_class && d.C == 1) return;                 // Scale_0_1 on a squeezed 2-D map is a no-op (PnP.py:1079-1080)
    __shared__ float smn[4], smx[4];
    const float* m = maps + d.off + (size_t)kk * d.H * d.W;
    float mn = INFINITY, mx = -INFINITY;
    bool nan = false;
    const int n = d.H * d.W;
    for (int p0 = tid; p0 < n; p0 += 8 * 256) {               // eight independent loads in flight per thread
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int p = p0 + u * 256;
            v[u] = p < n ? m[p] : m[p0];                        // out-of-range slots repeat an in-range sample
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            nan |= (v[u] != v[u]);
            mn = fminf(mn, v[u]);
            mx = fmaxf(mx, v[u]);
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, o, 64));
        mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        nan |= (bool)__shfl_xor((int)nan, o, 64);
    }
    __shared__ int snan[4];
    if ((tid & 63) == 0) { smn[tid >> 6] = mn; smx[tid >> 6] = mx; snan[tid >> 6] = nan; }
    __syncthreads();
    if (tid == 0) {
        mn = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
        mx = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
        if (snan[0] | snan[1] | snan[2] | snan[3]) mn = mx = __uint_as_float(0x7fc00000u);   // numpy/torch min/max propagate NaN
        stats[((size_t)b * Kmax + kk) * 2 + 0] = mn;
        stats[((size_t)b * Kmax + kk) * 2 + 1] = mx;
    }
}

// x = (x - min) / (max - min)   [x -= min ; x /= x.max()  in fp32, PnP.py:1151-1152 / :1086-1087]
__global__ void normalize_kernel(float* __restrict__ maps, const PostDesc* __restrict__ desc, const float* __restrict__ stats,
                                 int Kmax, int first_k_is_class) {
    const int b = blockIdx.z, k = blockIdx.y;
    const PostDesc d = desc[b];
    const int kk = first_k_is_class ? k + d.has_bg : k;
    if (kk >= d.K) return;
    if (first_k_is_class && d.C == 1) return;
    const float mn = stats[((size_t)b * Kmax + kk) * 2], mx = stats[((size_t)b * Kmax + kk) * 2 + 1];
    const float den = __fsub_rn(mx, mn);
    float* m = maps + d.off + (size_t)kk * d.H * d.W;
    const int n = d.H * d.W;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        m[i] = __fdiv_rn(__fsub_rn(m[i], mn), den);
}

// background channel 0 = (max over class channels == 0)
__global__ void background_kernel(float* __restrict__ maps, const PostDesc* __restrict__ desc) {
    const int b = blockIdx.y;
    const PostDesc d = desc[b];
    if (!d.has_bg) return;
    const int n = d.H * d.W;
    float* m = maps + d.off;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float mx = m[(size_t)n + i];
        bool nan = mx != mx;
        int c = 1;
        for (; c + 8 <= d.C; c += 8) {                       // eight channel planes in flight
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = m[(size_t)(1 + c + u) * n + i];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                nan |= (v[u] != v[u]);
                mx = fmaxf(mx, v[u]);
            }
        }
        for (; c < d.C; c++) {
            const float v = m[(size_t)(1 + c) * n + i];
            nan |= (v != v);
            mx = fmaxf(mx, v);
        }
        m[i] = (!nan && mx == 0.f) ? 1.f : 0.f;          // torch.max propagates NaN; NaN == 0 is False
    }
}

// ------------------------------------------------------------------------------------------
// scipy.ndimage.correlate1d (symmetric kernel, mode='reflect') along one axis, double
// accumulation in scipy's order: tmp = x[i]*w[0]; for j = r..1: tmp += (x[i-j] + x[i+j]) * w[j].
// wts[j] = weight at distance j (j = 0..radius), float64.  fp32 in / fp32 out.
__device__ __forceinline__ int reflect_idx(int i, int n) {
    // half-sample symmetric: (d c b a | a b c d | d c b a), valid for any offset
    const int period = 2 * n;
    int m = i % period;
    if (m < 0) m += period;
    return m >= n ? period - 1 - m : m;
}

// Sliding-window evaluation: a thread owns R = 8 CONSECUTIVE outputs along the filtered axis.  For
// tap distance j the eight left samples x[o_i - j] and the eight right samples x[o_i + j] form two
// register windows that shift by one element when j decreases, so each tap costs two LDS reads (plus
// the broadcast weight) for eight outputs instead of sixteen.  The j loop is unrolled by 8 so the
// window rotation is compile-time register renaming.  Per output the sum is still
// x[o]*w[0] + sum_{j=r..1} (x[o-j] + x[o+j]) * w[j] in scipy's order, in fp64 (samples are staged
// as fp32 -- their storage type -- and widened on the way into the window, which is exact).
//   ctr: LDS address of output 0 of the thread, S: element stride along the filtered axis.
__device__ __forceinline__ void blur_window8(const float* __restrict__ ctr, int S, const double* __restrict__ wl, int radius,
                                             double* __restrict__ acc) {
    constexpr int R = 8;
    double lo[R], hi[R];
#pragma unroll
    for (int i = 0; i < R; i++) {
        acc[i] = __dmul_rn((double)ctr[i * S], wl[0]);
        lo[i] = (double)ctr[(i - radius) * S];
        hi[i] = (double)ctr[(i + radius) * S];
    }
    for (int j = radius; j >= 1; j -= R) {
#pragma unroll
        for (int jj = 0; jj < R; jj++) {
            const int jc = j - jj;                      // current tap distance (may run past 1: predicated)
            if (jc >= 1) {
                const double wj = wl[jc];
                // three passes over the eight outputs (sums, products, accumulations) instead of one add-mul-add chain per
                // output through a single temporary: eight independent fp64 operations between dependent ones
                double t8[R];
#pragma unroll
                for (int i = 0; i < R; i++)             // logical window slot i lives in lo[(i + jj) % R], hi[(i - jj) & 7]
                    t8[i] = __dadd_rn(lo[(i + jj) % R], hi[(i - jj + R) % R]);
#pragma unroll
                for (int i = 0; i < R; i++) t8[i] = __dmul_rn(t8[i], wj);
#pragma unroll
                for (int i = 0; i < R; i++) acc[i] = __dadd_rn(acc[i], t8[i]);
                // shift: lo gains x[o_{R-1} - (jc-1)], hi gains x[o_0 + (jc-1)]
                lo[jj % R] = (double)ctr[(R - jc) * S];
                hi[(R - 1 - jj) % R] = (double)ctr[(jc - 1) * S];
            }
        }
    }
}

// half-sample symmetric reflection; the branch-free form covers |overshoot| < n (always true here
// unless the image is smaller than the kernel radius)
__device__ __forceinline__ int reflect_fast(int i, int n) {
    if (i >= 0 && i < n) return i;
    const int m = i < 0 ? -i - 1 : 2 * n - 1 - i;
    return (m >= 0 && m < n) ? m : reflect_idx(i, n);
}

// LDS-tiled separable pass: a workgroup stages its output tile + halo once.  The tile is NT sub-tiles long ALONG the filtered
// axis (a halo of 2 x radius samples is paid once per tile: at radius 67 a 32-row tile stages 5.2 samples per output, a
// 128-row tile 2.0; at radius 154 -- ADE20K at 768^2 -- 10.6 against 3.4), 256 threads x 8 consecutive outputs per sub-tile.
//   axis 0 (vertical):   xs[(32 NT + 2r)][64],     thread = (column c, row group of 8) per sub-tile
//   axis 1 (horizontal): xs[32][colsP], colsP = 64 NT + 2r rounded up to odd (conflict-free row stride),
//                        thread = (row, column group of 8) per sub-tile
template <int AXIS>
__global__ __launch_bounds__(1024) void blur_axis_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                        const PostDesc* __restrict__ desc, const double* __restrict__ wts,
                                                        const int32_t* __restrict__ wt_off, int max_radius, int NT) {
    constexpr int axis = AXIS;                                           // one kernel per pass: each keeps only its own staging code
    constexpr int TH = 32, TW = 64;
    extern __shared__ __attribute__((aligned(16))) double tile[];
    double* wl = tile;                                                   // [max_radius + 1] taps
    float* xs = reinterpret_cast<float*>(tile + ((max_radius + 3) & ~1));  // staged samples (16-byte aligned: the vertical pass stages 16-byte pieces)
    const int b = blockIdx.z, k = blockIdx.y;
    const PostDesc d = desc[b];
    if (k >= d.K) return;
    const int H = d.H, W = d.W;
    const int THe = axis == 0 ? TH * NT : TH, TWe = axis == 1 ? TW * NT : TW;     // tile extent (outputs)
    const int tiles_x = (W + TWe - 1) / TWe, tiles_y = (H + THe - 1) / THe;
    const double* w = wts + wt_off[b];
    const int radius = wt_off[b + 1] - wt_off[b] - 1;
    const float* src = in + d.off + (size_t)k * H * W;
    float* dst = out + d.off + (size_t)k * H * W;
    const int tid = threadIdx.x, nthr = blockDim.x;                      // 256 threads per sub-tile: NT * 256
    const int sub = tid >> 8, t = tid & 255;
    for (int j = tid; j <= radius; j += nthr) wl[j] = w[j];
    const int colsP = (TWe + 2 * radius) | 1;
    for (int tl = blockIdx.x; tl < tiles_x * tiles_y; tl += gridDim.x) {
        const int ty = tl / tiles_x, tx = tl - ty * tiles_x;
        const int y0 = ty * THe, x0 = tx * TWe;
        __syncthreads();
        double acc[8];
        if constexpr (AXIS == 0) {
            const int rows = THe + 2 * radius;           // xs[rows][TW]
            const int c = tid & 63;
            const int x = x0 + c < W ? x0 + c : W - 1;
            // eight independent loads in flight per thread (a one-load-per-iteration loop made the whole kernel
            // wait on global latency: the arithmetic of a tile is ~1 us, its staging was ~20 us).  Round 6: 16-byte pieces where
            // the rows allow it (W a multiple of 4, a full-width tile, an aligned plane) -- with 4-byte loads a workgroup had 8 KB in
            // flight and staged its 42 KB tile + halo in six latency-bound passes, half of the launch
            if ((W & 3) == 0 && x0 + TW <= W && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
                const int c4 = (tid & 15) * 4;
                const int rp = nthr >> 4;                // rows staged per pass of the workgroup
                for (int r0 = tid >> 4; r0 < rows; r0 += 8 * rp) {
                    f32x4 v[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const int r = r0 + rp * u;
                        v[u] = r < rows ? *reinterpret_cast<const f32x4*>(src + (size_t)reflect_fast(y0 - radius + r, H) * W + x0 + c4)
                                        : f32x4{0.f, 0.f, 0.f, 0.f};
                    }
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const int r = r0 + rp * u;
                        if (r < rows) *reinterpret_cast<f32x4*>(xs + r * TW + c4) = v[u];
                    }
                }
            } else {
            const int rp = nthr >> 6;                    // rows staged per pass of the workgroup
            for (int r0 = tid >> 6; r0 < rows; r0 += 8 * rp) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int r = r0 + rp * u;
                    v[u] = r < rows ? src[(size_t)reflect_fast(y0 - radius + r, H) * W + x] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int r = r0 + rp * u;
                    if (r < rows) xs[r * TW + c] = v[u];
                }
            }
            }
            __syncthreads();
            {
                const int rg = sub * 4 + (t >> 6);
                if (y0 + rg * 8 >= H) continue;
                blur_window8(xs + (size_t)(rg * 8 + radius) * TW + c, TW, wl, radius, acc);
                if (x0 + c < W) {
#pragma unroll
                    for (int i = 0; i < 8; i++)
                        if (y0 + rg * 8 + i < H) dst[(size_t)(y0 + rg * 8 + i) * W + x0 + c] = (float)acc[i];
                }
            }
        } else {
            // flat (row, column) index so all 256 threads load, eight independent loads in flight each
            const int total = TH * colsP;
            constexpr int NF = 16;                       // loads in flight per thread (round 6: 16, was 8: 198-float rows at arbitrary
                                                         // offsets with reflected ends stay 4-byte loads, so depth is what is left)
            for (int i0 = tid; i0 < total; i0 += NF * nthr) {
                float v[NF];
#pragma unroll
                for (int u = 0; u < NF; u++) {
                    const int i = i0 + u * nthr;
                    const int r = i / colsP, c = i - r * colsP;
                    const int y = y0 + r < H ? y0 + r : H - 1;
                    v[u] = i < total ? src[(size_t)y * W + reflect_fast(x0 - radius + c, W)] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < NF; u++) {
                    const int i = i0 + u * nthr;
                    if (i < total) xs[i] = v[u];
                }
            }
            __syncthreads();
            const int rr = t & 31;
            {
                const int cg = sub * 8 + (t >> 5);
                if (x0 + cg * 8 >= W) continue;
                blur_window8(xs + (size_t)rr * colsP + radius + cg * 8, 1, wl, radius, acc);
                if (y0 + rr < H) {
                    float* o = dst + (size_t)(y0 + rr) * W + x0 + cg * 8;
                    if (x0 + cg * 8 + 7 < W && ((reinterpret_cast<uintptr_t>(o) & 15) == 0)) {
                        reinterpret_cast<f32x4*>(o)[0] = f32x4{(float)acc[0], (float)acc[1], (float)acc[2], (float)acc[3]};
                        reinterpret_cast<f32x4*>(o)[1] = f32x4{(float)acc[4], (float)acc[5], (float)acc[6], (float)acc[7]};
                    } else {
#pragma unroll
                        for (int i = 0; i < 8; i++)
                            if (x0 + cg * 8 + i < W) o[i] = (float)acc[i];
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Input side: Pillow's 8-bit bicubic resampler + ToTensor + Normalize (Dataset.py:434-443), integer exact.
// Tables: per output index (first tap, tap count, taps[ksize]) in 22-bit fixed point, built on the host.
struct PreImage {           // mirrors pnp_pre_image (include/pnp_hip.h)
    int64_t src_off, tmp_off;
    int32_t H, W, kx_off, kx_size, ky_off, ky_size;
};
__device__ __forceinline__ int clip8_fixed(int acc) {
    const int v = (acc + (1 << 21)) >> 22;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}
// horizontal: tmp[y][xo][c] for all rows of the image; one thread per (row, output column)
__global__ __launch_bounds__(256) void pre_resize_h_kernel(const uint8_t* __restrict__ rgb, const PreImage* __restrict__ desc,
                                                           int S, const int32_t* __restrict__ coef, uint8_t* __restrict__ tmp) {
    const PreImage d = desc[blockIdx.z];
    const int xo = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (xo >= S || y >= d.H) return;
    const int32_t* t = coef + d.kx_off + (size_t)xo * (2 + d.kx_size);
    const int x0 = t[0], n = t[1];
    const uint8_t* row = rgb + d.src_off + ((size_t)y * d.W + x0) * 3;
    int a0 = 0, a1 = 0, a2 = 0;
    for (int i = 0; i < n; i++) {
        const int w = t[2 + i];
        a0 += w * row[3 * i];
        a1 += w * row[3 * i + 1];
        a2 += w * row[3 * i + 2];
    }
    uint8_t* o = tmp + d.tmp_off + ((size_t)y * S + xo) * 3;
    o[0] = (uint8_t)clip8_fixed(a0);
    o[1] = (uint8_t)clip8_fixed(a1);
    o[2] = (uint8_t)clip8_fixed(a2);
}
// vertical + ToTensor + Normalize: out[b][c][yo][xo]
__global__ __launch_bounds__(256) void pre_resize_v_kernel(const uint8_t* __restrict__ tmp, const PreImage* __restrict__ desc,
                                                           int S, const int32_t* __restrict__ coef, float m0, float m1, float m2,
                                                           float s0, float s1, float s2, float* __restrict__ out) {
    const PreImage d = desc[blockIdx.z];
    const int xo = blockIdx.x * 64 + (threadIdx.x & 63), yo = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (xo >= S || yo >= S) return;
    const int32_t* t = coef + d.ky_off + (size_t)yo * (2 + d.ky_size);
    const int y0 = t[0], n = t[1];
    const uint8_t* col = tmp + d.tmp_off + ((size_t)y0 * S + xo) * 3;
    int a0 = 0, a1 = 0, a2 = 0;
    for (int i = 0; i < n; i++) {
        const int w = t[2 + i];
        const uint8_t* p = col + (size_t)i * S * 3;
        a0 += w * p[0];
        a1 += w * p[1];
        a2 += w * p[2];
    }
    float* o = out + (size_t)blockIdx.z * 3 * S * S + (size_t)yo * S + xo;
    o[0] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)clip8_fixed(a0), 255.0f), m0), s0);
    o[(size_t)S * S] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)clip8_fixed(a1), 255.0f), m1), s1);
    o[(size_t)2 * S * S] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)clip8_fixed(a2), 255.0f), m2), s2);
}

int preprocess_images(const uint8_t* rgb, const void* desc, int B, int S, int max_H, const int32_t* coef, uint8_t* tmp,
                      const float* mean3, const float* std3, float* out, hipStream_t s) {
    if (B <= 0 || S <= 0 || max_H <= 0) return PNP_ERR_ARG;
    const PreImage* d = reinterpret_cast<const PreImage*>(desc);
    hipLaunchKernelGGL(pre_resize_h_kernel, dim3((S + 63) / 64, (max_H + 3) / 4, B), dim3(256), 0, s, rgb, d, S, coef, tmp);
    hipLaunchKernelGGL(pre_resize_v_kernel, dim3((S + 63) / 64, (S + 3) / 4, B), dim3(256), 0, s, tmp, d, S, coef, mean3[0],
                       mean3[1], mean3[2], std3[0], std3[1], std3[2], out);
    return hipGetLastError() == hipSuccess ? PNP_OK : PNP_ERR_HIP;
}

// ------------------------------------------------------------------------------------------
// softmax over channels + unary = -log(clip(p, 1e-5, 1)), written pixel-major [n][K]
// (densecrf's value layout).  include/pnp_math.h defines exp/log bit-exactly for host and device.
__global__ __launch_bounds__(256) void unary_kernel(const float* __restrict__ maps, const PostDesc* __restrict__ desc,
                                                    float* __restrict__ unary, int group) {
    // tile of 256 pixels: channel-major reads (coalesced along pixels), results parked in LDS [pixel][K + 1], then
    // written as whole 16-byte chunks of the pixel-major rows (the per-thread row writes were 4 bytes at a 96-byte stride).
    // A row holds G groups of K floats back to back (+ zero pad to a multiple of 4): this call owns floats [f0, f1) of
    // every row.  A chunk it shares with the previous group (f0 not a multiple of 4) is read back and completed -- the
    // groups of a row are written by successive launches on one stream; floats behind f1 are written as zeros (the next
    // group's launch, or nobody: the pad).
    extern __shared__ __attribute__((aligned(16))) float utile[];
    const int b = blockIdx.y;
    const PostDesc d = desc[b];
    const int n = d.H * d.W, K = d.K, ldt = K + 1, R4 = d.Kp >> 2;
    const int f0 = group * d.Kg, f1 = f0 + K;
    const int c_lo = f0 >> 2, c_hi = group == d.G - 1 ? R4 : (f1 + 3) >> 2, nc = c_hi - c_lo;
    const float* m = maps + d.off;
    f32x4* u4 = reinterpret_cast<f32x4*>(unary + d.qoff);
    const int tid = threadIdx.x;
    for (int p0 = blockIdx.x * 256; p0 < n; p0 += gridDim.x * 256) {
        const int np = (n - p0) < 256 ? (n - p0) : 256;
        if (tid < np) {
            const int i = p0 + tid;
            const float* mi = m + i;
            // channel planes in batches of eight loads in flight (a per-channel loop waits for every load in turn: 21 + 21
            // dependent round trips per pixel); maxima and sums stay the sequential per-pixel ones
            float mx = mi[0];
            int k = 1;
            for (; k + 8 <= K; k += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = mi[(size_t)(k + u) * n];
#pragma unroll
                for (int u = 0; u < 8; u++)
                    if (v[u] > mx || v[u] != v[u]) mx = v[u];
            }
            for (; k < K; k++) {
                const float v = mi[(size_t)k * n];
                if (v > mx || v != v) mx = v;
            }
            float s = 0.f;
            float* row = utile + tid * ldt;
            for (k = 0; k + 8 <= K; k += 8) {                            // the exponentials are parked in the tile: one exp per
                float v[8];                                             // element instead of two (same values)
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = mi[(size_t)(k + u) * n];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const float e = pnp_expf(__fsub_rn(v[u], mx));
                    row[k + u] = e;
                    s = __fadd_rn(s, e);
                }
            }
            for (; k < K; k++) {
                const float e = pnp_expf(__fsub_rn(mi[(size_t)k * n], mx));
                row[k] = e;
                s = __fadd_rn(s, e);
            }
            for (int k = 0; k < K; k++) {
                float p = __fdiv_rn(row[k], s);
                if (p < 1e-5f) p = 1e-5f;
                else if (p > 1.0f) p = 1.0f;
                row[k] = -pnp_logf(p);
            }
        }
        __syncthreads();
        for (int item = tid; item < np * nc; item += 256) {
            const int pl = item / nc, c = c_lo + (item - pl * nc);
            const float* r = utile + pl * ldt;
            f32x4* dst = u4 + (size_t)(p0 + pl) * R4 + c;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (4 * c < f0) v = *dst;                                    // the previous group's floats of a shared chunk
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int f = 4 * c + j;
                if (f >= f0) v[j] = f < f1 ? r[f - f0] : 0.f;
            }
            *dst = v;
        }
        __syncthreads();
    }
}

// The same unary for wide rows (K >= 32: the 59 / 80 / 150-class prompts).  Parking a pixel's K exponentials in LDS leaves
// 256 / (K + 1) KB-sized tiles -- one 4-wave workgroup per CU at K = 150, every thread walking 150 channel planes with a
// handful of loads in flight: 10.5 ms per launch at ADE20K size, latency-bound.  Here nothing is parked: a wave owns 64
// pixels, reads the planes three times (maximum; sum of exponentials; the exponentials again for p = e / s -- the same
// values, pnp_expf is a pure function), eight loads in flight per pass, and only a 64 x 32-float transpose tile per wave goes
// through LDS so the pixel-major rows leave as 16-byte chunks, eight chunks (128 B) of a row at a time.  8.4 KB of LDS per
// wave: four waves per SIMD with eight loads in flight each.  Sums and maxima are the sequential per-pixel ones of unary_kernel.
__global__ __launch_bounds__(256) void unary_wide_kernel(const float* __restrict__ maps, const PostDesc* __restrict__ desc,
                                                         float* __restrict__ unary, int group) {
    __shared__ float tiles[4][64][33];
    const int b = blockIdx.y;
    const int n = desc[b].H * desc[b].W, K = desc[b].K, R4 = desc[b].Kp >> 2, G = desc[b].G;
    const int f0 = group * desc[b].Kg, f1 = f0 + K;
    const int c_lo = f0 >> 2, c_hi = group == G - 1 ? R4 : (f1 + 3) >> 2;
    const float* m = maps + desc[b].off;
    f32x4* u4 = reinterpret_cast<f32x4*>(unary + desc[b].qoff);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float (*tile)[33] = tiles[wave];
    for (int p0 = (blockIdx.x * 4 + wave) * 64; p0 < n; p0 += gridDim.x * 256) {
        const int i = p0 + lane < n ? p0 + lane : n - 1;                  // lanes past the end redo the last pixel, store nothing
        const float* mi = m + i;
        float mx = mi[0];
        int k = 1;
        for (; k + 8 <= K; k += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = mi[(size_t)(k + u) * n];
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (v[u] > mx || v[u] != v[u]) mx = v[u];
        }
        for (; k < K; k++) {
            const float v = mi[(size_t)k * n];
            if (v > mx || v != v) mx = v;
        }
        float s = 0.f;
        for (k = 0; k + 8 <= K; k += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = mi[(size_t)(k + u) * n];
#pragma unroll
            for (int u = 0; u < 8; u++) s = __fadd_rn(s, pnp_expf(__fsub_rn(v[u], mx)));
        }
        for (; k < K; k++) s = __fadd_rn(s, pnp_expf(__fsub_rn(mi[(size_t)k * n], mx)));
        // blocks of 8 chunks = 32 floats of the row
        for (int cb = c_lo; cb < c_hi; cb += 8) {
            const int fb = 4 * cb;                                         // first row float of the block
            const int ka = (fb > f0 ? fb : f0) - f0, kb = (fb + 32 < f1 ? fb + 32 : f1) - f0;   // channels [ka, kb) fall in it
            for (k = ka; k < kb; k += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = k + u < kb ? mi[(size_t)(k + u) * n] : 0.f;
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    if (k + u < kb) {
                        float p = __fdiv_rn(pnp_expf(__fsub_rn(v[u], mx)), s);
                        if (p < 1e-5f) p = 1e-5f;
                        else if (p > 1.0f) p = 1.0f;
                        tile[lane][f0 + k + u - fb] = -pnp_logf(p);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            const int ncb = c_hi - cb < 8 ? c_hi - cb : 8;
            for (int item = lane; item < 64 * ncb; item += 64) {
                const int pl = item / ncb, c = item - pl * ncb;
                if (p0 + pl >= n) continue;
                f32x4* dst = u4 + (size_t)(p0 + pl) * R4 + cb + c;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (4 * (cb + c) < f0) v = *dst;                           // the previous group's floats of a shared chunk
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int f = fb + 4 * c + j;
                    if (f >= f0) v[j] = f < f1 ? tile[pl][f - fb] : 0.f;
                }
                *dst = v;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// argmax over channels.  pixel_major: Q[n][Kp] (CRF marginals, rows padded to Kp) else maps[K][n].
// np.argmax semantics: first maximum, NaN counts as maximum.
__global__ void argmax_kernel(const float* __restrict__ q, const PostDesc* __restrict__ desc, const int32_t* __restrict__ lut,
                              int lut_stride, uint8_t* __restrict__ labels, const size_t* __restrict__ label_off,
                              int pixel_major, int group) {
    const int b = blockIdx.y;
    const PostDesc d = desc[b];
    const int n = d.H * d.W, K = d.K, Kp = d.Kp;
    const float* p = q + (pixel_major ? d.qoff : d.off);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        int best = 0;
        float bv = pixel_major ? p[(size_t)i * Kp + group * d.Kg] : p[i];
        if (pixel_major) {                          // rows are G groups of Kg = K floats back to back: aligned 16-byte chunks
            const int f0 = group * d.Kg;            // covering floats [f0, f0 + K)
            const f32x4* row = reinterpret_cast<const f32x4*>(p + (size_t)i * Kp);
            for (int c = f0 >> 2; c < (f0 + K + 3) >> 2; c++) {
                const f32x4 v4 = row[c];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int k = c * 4 + j - f0;
                    const float v = v4[j];
                    if (k >= 1 && k < K && bv == bv && (v > bv || v != v)) { best = k; bv = v; }
                }
            }
        } else {
            for (int k = 1; k < K; k++) {
                const float v = p[(size_t)k * n + i];
                if (bv == bv && (v > bv || v != v)) { best = k; bv = v; }
            }
        }
        labels[label_off[b] + i] = (uint8_t)lut[b * lut_stride + best];
    }
}

// confusion histogram: hist[n_class * gt + pred] += 1 for 0 <= gt < n_class  (PnP.py:1106-1112).
// Counts are privatised per workgroup in LDS (integer adds: order-independent, exact) and flushed
// with one global atomic per non-zero bin; n_class^2 > 8192 bins falls back to global atomics.
__global__ __launch_bounds__(256) void hist_kernel(const uint8_t* __restrict__ labels, const float* __restrict__ gt,
                                                   const PostDesc* __restrict__ desc, const size_t* __restrict__ label_off,
                                                   unsigned long long* __restrict__ hist, int n_class) {
    __shared__ unsigned int bins[8192];
    const int b = blockIdx.y;
    const PostDesc d = desc[b];
    const int n = d.H * d.W, nb = n_class * n_class;
    const bool use_lds = nb <= 8192;
    // one copy of the bins per wave while they fit (21 classes: 441 bins x 4): the pixels of an image hit a handful of
    // (truth, prediction) pairs, and LDS atomics on one address serialise
    const int copies = nb * 4 <= 8192 ? 4 : nb * 2 <= 8192 ? 2 : 1;
    unsigned int* const mybins = bins + ((threadIdx.x >> 6) % copies) * nb;
    if (use_lds) {
        for (int i = threadIdx.x; i < nb * copies; i += 256) bins[i] = 0;
        __syncthreads();
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float g = gt[label_off[b] + i];
        if (g >= 0.f && g < (float)n_class) {
            const int gi = (int)g, pi = labels[label_off[b] + i];
            if (pi < n_class) {
                if (use_lds) atomicAdd(&mybins[gi * n_class + pi], 1u);
                else atomicAdd(&hist[(size_t)gi * n_class + pi], 1ULL);
            }
        }
    }
    if (use_lds) {
        __syncthreads();
        for (int i = threadIdx.x; i < nb; i += 256) {
            unsigned int t = bins[i];
            for (int c = 1; c < copies; c++) t += bins[c * nb + i];
            if (t) atomicAdd(&hist[i], (unsigned long long)t);
        }
    }
}

// ------------------------------------------------------------------------------------------ host
static inline int ok() { return hipGetLastError() == hipSuccess ? PNP_OK : PNP_ERR_HIP; }

int gradcam_gather(const float* P, const float* dP, const int64_t* mask, int ld_mask, float* out, int B, int nheads,
                   int head, int L, int Nst, int PP, hipStream_t s) {
    const int total = B * (L - 1) * PP;
    if (total <= 0) return PNP_ERR_ARG;
    hipLaunchKernelGGL(gradcam_gather_kernel, dim3((total + 255) / 256), dim3(256), 0, s, P, dP, mask, ld_mask, out, B,
                       nheads, head, L, Nst, PP);
    return ok();
}

int drop_step(const float* G, float* g0, float* agg, uint8_t* dropped, int32_t* picks, int iter, int B, int T, int PP,
              int npick, int max_picks, hipStream_t s) {
    if (PP * sizeof(float) > 48 * 1024) return PNP_ERR_ARG;
    const int nb = (T * PP + 1023) / 1024 < 256 ? (T * PP + 1023) / 1024 : 256;
    hipLaunchKernelGGL(drop_accumulate_kernel, dim3(nb, B), dim3(256), 0, s, G, g0, agg, dropped, iter, T, PP);
    hipLaunchKernelGGL(drop_step_kernel, dim3(B), dim3(256), PP * sizeof(float), s, G, dropped, picks, iter, T, PP, npick,
                       max_picks);
    return ok();
}

int merge_tokens(const float* src, const int32_t* cls_off, const int32_t* tok_idx, const int32_t* cls_div,
                 const int32_t* img_cls_off, float* out, int B, int T, int PP, int Cmax, hipStream_t s) {
    hipLaunchKernelGGL(merge_tokens_kernel, dim3(Cmax, B), dim3(256), 0, s, src, cls_off, tok_idx, cls_div, img_cls_off,
                       out, T, PP, Cmax);
    return ok();
}

int threshold_maps(const float* merged, const int32_t* img_cls_off, float* out, float threshold, int B, int PP, int Cmax,
                   hipStream_t s) {
    hipLaunchKernelGGL(threshold_kernel, dim3(Cmax, B), dim3(256), 0, s, merged, img_cls_off, out, threshold, PP, Cmax);
    return ok();
}

int upsample_maps(const float* src, const PostDesc* desc, float* maps, int B, int P, int Cmax, int maxHW, hipStream_t s) {
    const int nb = (maxHW + 255) / 256 < 64 ? (maxHW + 255) / 256 : 64;
    hipLaunchKernelGGL(upsample_kernel, dim3(nb, Cmax, B), dim3(256), 0, s, src, desc, maps, P, Cmax);
    return ok();
}

int minmax_normalize(float* maps, const PostDesc* desc, float* stats, int B, int Kmax, int maxHW, int class_channels_only,
                     hipStream_t s) {
    hipLaunchKernelGGL(minmax_kernel, dim3(Kmax, B), dim3(256), 0, s, maps, desc, stats, Kmax, class_channels_only);
    const int nb = (maxHW + 255) / 256 < 64 ? (maxHW + 255) / 256 : 64;
    hipLaunchKernelGGL(normalize_kernel, dim3(nb, Kmax, B), dim3(256), 0, s, maps, desc, stats, Kmax, class_channels_only);
    return ok();
}

int background_channel(float* maps, const PostDesc* desc, int B, int maxHW, hipStream_t s) {
    const int nb = (maxHW + 255) / 256 < 128 ? (maxHW + 255) / 256 : 128;
    hipLaunchKernelGGL(background_kernel, dim3(nb, B), dim3(256), 0, s, maps, desc);
    return ok();
}

int blur_maps(const float* in, float* tmp, float* out, const PostDesc* desc, const double* wts, const int32_t* wt_off,
              int B, int Kmax, int maxH, int maxW, int max_radius, hipStream_t s) {
    // sub-tiles per tile along the filtered axis.  Small radii (67 at 336^2: 43 KB per 32-row tile, three workgroups per CU
    // overlap each other's staging) keep one; once the halo alone leaves one workgroup per CU (radius 154 at 768^2: 87 KB)
    // nothing overlaps the staging any more and the tile grows to as many as fit ~120 KB (at most 4): 23.5 -> 18.2 ms per
    // launch at ADE20K size, where a 1-sub-tile launch at radius 67 is 6 % faster than a 4-sub-tile one
    auto lds_for = [&](int n, size_t& l0, size_t& l1) {
        l0 = (size_t)(max_radius + 4) * sizeof(double) + (size_t)(32 * n + 2 * max_radius) * 64 * sizeof(float);
        l1 = (size_t)(max_radius + 4) * sizeof(double) + (size_t)32 * ((64 * n + 2 * max_radius) | 1) * sizeof(float);
    };
    size_t lds0, lds1;
    lds_for(1, lds0, lds1);
    int nt = lds0 > 80 * 1024 ? 4 : 1;
#ifdef PNP_DEV
    if (getenv("PNP_BLUR_NT")) nt = atoi(getenv("PNP_BLUR_NT"));
#endif
    lds_for(nt, lds0, lds1);
    while (nt > 1 && (lds0 > 120 * 1024 || lds1 > 120 * 1024)) lds_for(--nt, lds0, lds1);
    if (lds0 > 160 * 1024 || lds1 > 160 * 1024) return PNP_ERR_ARG;            // radius <= ~300 (images up to ~1500 px on the long side)
    int nt0 = nt, nt1 = nt;
    while (nt0 > 1 && 32 * (nt0 - 1) >= maxH) nt0--;                          // no taller / wider than the image
    while (nt1 > 1 && 64 * (nt1 - 1) >= maxW) nt1--;
    lds_for(nt0, lds0, lds1);
    size_t d0, d1;
    lds_for(nt1, d0, d1);
    lds1 = d1;
    static std::atomic<uint32_t> opted0{0}, opted1{0};  // per device ordinal (common.h: lds_opt_in)
    if (lds_opt_in(opted0, reinterpret_cast<const void*>(blur_axis_kernel<0>), 160 * 1024) != PNP_OK) return PNP_ERR_HIP;
    if (lds_opt_in(opted1, reinterpret_cast<const void*>(blur_axis_kernel<1>), 160 * 1024) != PNP_OK) return PNP_ERR_HIP;
    const int tiles0 = ((maxW + 63) / 64) * ((maxH + 32 * nt0 - 1) / (32 * nt0));
    const int tiles1 = ((maxW + 64 * nt1 - 1) / (64 * nt1)) * ((maxH + 31) / 32);
    hipLaunchKernelGGL(blur_axis_kernel<0>, dim3(tiles0, Kmax, B), dim3(256 * nt0), lds0, s, in, tmp, desc, wts, wt_off, max_radius, nt0);
    hipLaunchKernelGGL(blur_axis_kernel<1>, dim3(tiles1, Kmax, B), dim3(256 * nt1), lds1, s, tmp, out, desc, wts, wt_off, max_radius, nt1);
    return ok();
}

int unary_from_maps(const float* maps, const PostDesc* desc, float* unary, int B, int maxHW, int max_kp, int group, hipStream_t s) {
    const int nb = (maxHW + 255) / 256 < 256 ? (maxHW + 255) / 256 : 256;
    const size_t smem = (size_t)256 * (max_kp + 1) * sizeof(float);
    static std::atomic<uint32_t> opted{0};            // per device ordinal (common.h: lds_opt_in)
    if (lds_opt_in(opted, reinterpret_cast<const void*>(unary_kernel), 160 * 1024) != PNP_OK) return PNP_ERR_HIP;
    if (max_kp >= 32) {                                                    // wide rows: nothing parked in LDS
        const int nbw = (maxHW + 255) / 256 < 2048 ? (maxHW + 255) / 256 : 2048;
        hipLaunchKernelGGL(unary_wide_kernel, dim3(nbw, B), dim3(256), 0, s, maps, desc, unary, group);
        return ok();
    }
    hipLaunchKernelGGL(unary_kernel, dim3(nb, B), dim3(256), smem, s, maps, desc, unary, group);
    return ok();
}

int argmax_remap(const float* q, const PostDesc* desc, const int32_t* lut, int lut_stride, uint8_t* labels,
                 const size_t* label_off, int pixel_major, int group, int B, int maxHW, hipStream_t s) {
    const int nb = (maxHW + 255) / 256 < 256 ? (maxHW + 255) / 256 : 256;
    hipLaunchKernelGGL(argmax_kernel, dim3(nb, B), dim3(256), 0, s, q, desc, lut, lut_stride, labels, label_off, pixel_major, group);
    return ok();
}

int confusion_hist(const uint8_t* labels, const float* gt, const PostDesc* desc, const size_t* label_off,
                   unsigned long long* hist, int n_class, int B, int maxHW, hipStream_t s) {
    // bins in LDS (n_class^2 <= 8192): every workgroup pays for zeroing and flushing its bins, so few, long-running workgroups
    // per image (32 x 3500 pixels at 336^2, not 441 x 256: 181 us per launch was that overhead); global-atomic form: as before
    const int cap = n_class * n_class <= 8192 ? 32 : 256;
    const int nb = (maxHW + 255) / 256 < cap ? (maxHW + 255) / 256 : cap;
    hipLaunchKernelGGL(hist_kernel, dim3(nb, B), dim3(256), 0, s, labels, gt, desc, label_off, hist, n_class);
    return ok();
}

}  // namespace pnp
