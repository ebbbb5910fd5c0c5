// Dense-CRF mean-field on gfx950 (replaces the pydensecrf calls at
// PnP_OVSS_0514_updated_segmentation.py:1063-1073: DenseCRF2D + addPairwiseGaussian(sxy=3,w=7) +
// addPairwiseBilateral(sxy=50,srgb=5,w=10) + inference(10)).
//
// Permutohedral lattice, built WITHOUT a hash table: every (pixel, simplex-vertex) pair emits a
// packed 64-bit lattice key; a stable segmented radix sort groups equal keys, a prefix sum over
// segment heads numbers the lattice points, blur neighbours are found by binary search in the
// sorted unique keys.  Because the sort is stable, each lattice point's contributor list is in
// ascending pixel order, so the splat is a gather that adds in exactly the order of the sequential
// CPU algorithm: results are run-to-run deterministic and bit-comparable with the oracle
// (oracle/densecrf_ref.c).  All images of a batch go through every kernel together
// (blockIdx.y = image).  HBM-bound: values are streamed, nothing is reshaped into a GEMM.
#include <hipcub/hipcub.hpp>

#include "common.h"
#include "kernels.h"
#include "../../include/pnp_math.h"

namespace pnp {

template <int D> struct KeyPack;
template <> struct KeyPack<2> { static constexpr int BITS = 16; };
template <> struct KeyPack<5> { static constexpr int BITS = 12; };

template <int D>
__device__ __forceinline__ uint64_t pack_key(const int* c) {
    constexpr int BITS = KeyPack<D>::BITS;
    uint64_t k = 0;
#pragma unroll
    for (int i = 0; i < D; i++) k = (k << BITS) | (uint64_t)((c[i] + (1 << (BITS - 1))) & ((1 << BITS) - 1));
    return k;
}
template <int D>
__device__ __forceinline__ void unpack_key(uint64_t k, int* c) {
    constexpr int BITS = KeyPack<D>::BITS;
#pragma unroll
    for (int i = D - 1; i >= 0; i--) {
        c[i] = (int)(k & ((1 << BITS) - 1)) - (1 << (BITS - 1));
        k >>= BITS;
    }
}

// ------------------------------------------------------------------------------------------
// Per pixel: features -> elevate -> nearest remainder-0 point -> rank -> barycentric -> d+1 keys.
// Mirrors oracle/densecrf_ref.c::lattice_init statement by statement (fp contraction is off for
// this translation unit) so barycentric weights and keys are bit-identical.
template <int D>
__global__ void lattice_embed_kernel(const PostDesc* __restrict__ imgs, const uint8_t* __restrict__ rgb, float sxy, float srgb,
                                     float* __restrict__ bary, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                     int* __restrict__ range_err) {
    const int b = blockIdx.y;
    const PostDesc im = imgs[b];
    const int n = im.H * im.W;
    float scale_factor[D];
    {
        const float inv_std_dev = (float)(sqrt(2.0 / 3.0) * (D + 1));
#pragma unroll
        for (int i = 0; i < D; i++) scale_factor[i] = (float)(1.0 / sqrt((double)((i + 2) * (i + 1))) * inv_std_dev);
    }
    for (int pix = blockIdx.x * blockDim.x + threadIdx.x; pix < n; pix += gridDim.x * blockDim.x) {
        const int y = pix / im.W, x = pix - y * im.W;
        float f[D];
        f[0] = __fdiv_rn((float)x, sxy);
        f[1] = __fdiv_rn((float)y, sxy);
        if constexpr (D == 5) {
            const uint8_t* c = rgb + ((size_t)im.pix0 + pix) * 3;
            f[2] = __fdiv_rn((float)c[0], srgb);
            f[3] = __fdiv_rn((float)c[1], srgb);
            f[4] = __fdiv_rn((float)c[2], srgb);
        }
        float elevated[D + 1], rem0[D + 1], barycentric[D + 2];
        int rank[D + 1];
        float sm = 0.f;
#pragma unroll
        for (int j = D; j > 0; j--) {
            const float cf = __fmul_rn(f[j - 1], scale_factor[j - 1]);
            elevated[j] = __fsub_rn(sm, __fmul_rn((float)j, cf));
            sm = __fadd_rn(sm, cf);
        }
        elevated[0] = sm;
        const float down_factor = 1.0f / (D + 1);
        const float up_factor = (float)(D + 1);
        int sum = 0;
#pragma unroll
        for (int i = 0; i <= D; i++) {
            const float v = __fmul_rn(down_factor, elevated[i]);
            const float up = __fmul_rn(ceilf(v), up_factor);
            const float down = __fmul_rn(floorf(v), up_factor);
            int rd2;
            if (__fsub_rn(up, elevated[i]) < __fsub_rn(elevated[i], down)) rd2 = (int)(short)up;
            else rd2 = (int)(short)down;
            rem0[i] = (float)rd2;
            sum = (int)__fadd_rn((float)sum, __fmul_rn((float)rd2, down_factor));
        }
#pragma unroll
        for (int i = 0; i <= D; i++) rank[i] = 0;
#pragma unroll
        for (int i = 0; i < D; i++) {
            const double di = (double)__fsub_rn(elevated[i], rem0[i]);
#pragma unroll
            for (int j = i + 1; j <= D; j++) {
                if (di < (double)__fsub_rn(elevated[j], rem0[j])) rank[i]++;
                else rank[j]++;
            }
        }
#pragma unroll
        for (int i = 0; i <= D; i++) {
            rank[i] += sum;
            if (rank[i] < 0) {
                rank[i] += D + 1;
                rem0[i] = __fadd_rn(rem0[i], (float)(D + 1));
            } else if (rank[i] > D) {
                rank[i] -= D + 1;
                rem0[i] = __fsub_rn(rem0[i], (float)(D + 1));
            }
        }
#pragma unroll
        for (int i = 0; i <= D + 1; i++) barycentric[i] = 0.f;
#pragma unroll
        for (int i = 0; i <= D; i++) {
            const float v = __fmul_rn(__fsub_rn(elevated[i], rem0[i]), down_factor);
            // barycentric[D - rank[i]] += v ; barycentric[D - rank[i] + 1] -= v   (static indexing)
#pragma unroll
            for (int s = 0; s <= D + 1; s++) {
                if (s == D - rank[i]) barycentric[s] = __fadd_rn(barycentric[s], v);
                if (s == D - rank[i] + 1) barycentric[s] = __fsub_rn(barycentric[s], v);
            }
        }
        barycentric[0] = __fadd_rn(barycentric[0], __fadd_rn(1.0f, barycentric[D + 1]));

        const size_t e0 = (size_t)(im.pix0 + pix) * (D + 1);
#pragma unroll
        for (int rem = 0; rem <= D; rem++) {
            int key[D];
            bool bad = false;
#pragma unroll
            for (int i = 0; i < D; i++) {
                // canonical[rem][rank[i]] = rem if rank[i] <= D - rem else rem - (D+1)
                const int canon = (rank[i] <= D - rem) ? rem : rem - (D + 1);
                key[i] = (int)(short)(rem0[i] + (float)canon);
                bad |= (key[i] <= -(1 << (KeyPack<D>::BITS - 1)) + D + 1) || (key[i] >= (1 << (KeyPack<D>::BITS - 1)) - D - 1);
            }
            if (bad) atomicExch(range_err, 1);
            keys[e0 + rem] = pack_key<D>(key);
            vals[e0 + rem] = (uint32_t)(e0 + rem);
            bary[e0 + rem] = barycentric[rem];
        }
    }
}

// head[i] = 1 where sorted entry i starts a new lattice point (first entry of the image or new key)
__global__ void mark_heads_kernel(const uint64_t* __restrict__ keys, const PostDesc* __restrict__ imgs, int D1,
                                  int* __restrict__ head) {
    const int b = blockIdx.y;
    const PostDesc im = imgs[b];
    const size_t e0 = (size_t)im.pix0 * D1, n = (size_t)im.H * im.W * D1;
    for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        head[e0 + i] = (i == 0 || keys[e0 + i] != keys[e0 + i - 1]) ? 1 : 0;
}

// ids: inclusive scan of head.  offset[pv] = lattice id; seg_start[id] = first sorted entry;
// ukeys[id] = key; idbase[b] = id of the image's first entry; idbase[B] = seg_start[M] sentinel.
__global__ void scatter_ids_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                   const int* __restrict__ head, const int* __restrict__ incl, const PostDesc* __restrict__ imgs,
                                   int D1, int B, size_t ent_total, int* __restrict__ offset, int* __restrict__ seg_start,
                                   uint64_t* __restrict__ ukeys, int* __restrict__ idbase) {
    const int b = blockIdx.y;
    const PostDesc im = imgs[b];
    const size_t e0 = (size_t)im.pix0 * D1, n = (size_t)im.H * im.W * D1;
    for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t g = e0 + i;
        const int id = incl[g] - 1;
        offset[vals[g]] = id;
        if (head[g]) {
            seg_start[id] = (int)g;
            ukeys[id] = keys[g];
        }
        if (i == 0) idbase[b] = id;
        if (b == B - 1 && i == n - 1) {
            idbase[B] = id + 1;
            seg_start[id + 1] = (int)ent_total;
        }
    }
}

// blur neighbours along each of the d+1 lattice axes: n1 = key - 1 (coord j: + d), n2 = key + 1 (coord j: - d)
template <int D>
__global__ void neighbors_kernel(const uint64_t* __restrict__ ukeys, const int* __restrict__ idbase, size_t cap,
                                 int* __restrict__ n1, int* __restrict__ n2) {
    const int b = blockIdx.y;
    const int lo = idbase[b], hi = idbase[b + 1];
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < (hi - lo) * (D + 1); t += gridDim.x * blockDim.x) {
        const int id = lo + t / (D + 1), j = t % (D + 1);
        int c[D], a[D], d2[D];
        unpack_key<D>(ukeys[id], c);
#pragma unroll
        for (int k = 0; k < D; k++) {
            a[k] = c[k] - 1;
            d2[k] = c[k] + 1;
        }
#pragma unroll
        for (int k = 0; k < D; k++)
            if (k == j) {
                a[k] = c[k] + D;
                d2[k] = c[k] - D;
            }
        // binary search inside this image's sorted unique keys [lo, hi)
        int r1 = -1, r2 = -1;
        {
            constexpr int BITS = KeyPack<D>::BITS;
            bool in1 = true, in2 = true;
#pragma unroll
            for (int k = 0; k < D; k++) {
                in1 &= (a[k] >= -(1 << (BITS - 1)) && a[k] < (1 << (BITS - 1)));
                in2 &= (d2[k] >= -(1 << (BITS - 1)) && d2[k] < (1 << (BITS - 1)));
            }
            if (in1) {
                const uint64_t k1 = pack_key<D>(a);
                int l = lo, h = hi;
                while (l < h) {
                    const int mid = (l + h) >> 1;
                    if (ukeys[mid] < k1) l = mid + 1;
                    else h = mid;
                }
                if (l < hi && ukeys[l] == k1) r1 = l;
            }
            if (in2) {
                const uint64_t k2 = pack_key<D>(d2);
                int l = lo, h = hi;
                while (l < h) {
                    const int mid = (l + h) >> 1;
                    if (ukeys[mid] < k2) l = mid + 1;
                    else h = mid;
                }
                if (l < hi && ukeys[l] == k2) r2 = l;
            }
        }
        n1[(size_t)j * cap + id] = r1;
        n2[(size_t)j * cap + id] = r2;
    }
}

// ------------------------------------------------------------------------------------------
// splat: val[id, k] = sum over the lattice point's contributors (ascending pixel) of
//        bary * (Q[pixel, k] * norm[pixel])          (Q == nullptr: the all-ones vector, K = 1)
__global__ void crf_splat_kernel(const CrfLattice L, const PostDesc* __restrict__ imgs, const float* __restrict__ Q,
                                 const float* __restrict__ norm, float* __restrict__ val, int img0, int force_k1) {
    const int b = img0 + blockIdx.y;
    const PostDesc im = imgs[b];
    const int K = force_k1 ? 1 : im.K;
    const int lo = L.idbase[b], hi = L.idbase[b + 1];
    const size_t vbase = force_k1 ? (size_t)lo : im.voff[L.which] ;
    const int total = (hi - lo) * K;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        const int idl = t / K, k = t - idl * K;
        const int id = lo + idl;
        const int e0 = L.seg_start[id], e1 = L.seg_start[id + 1];
        float acc = 0.f;
        for (int e = e0; e < e1; e++) {
            const uint32_t pv = L.vals[e];
            const uint32_t pixel = pv / (uint32_t)L.D1;             // global pixel index
            float in = 1.0f;
            if (Q) in = Q[im.off + (size_t)(pixel - im.pix0) * K + k];
            if (norm) in = __fmul_rn(in, norm[pixel]);
            acc = __fadd_rn(acc, __fmul_rn(L.bary[pv], in));
        }
        val[vbase + (size_t)idl * K + k] = acc;
    }
}

// one axis of the lattice blur: new = old + 0.5 * (n1 + n2), absent neighbours contribute 0
__global__ void crf_blur_kernel(const CrfLattice L, const PostDesc* __restrict__ imgs, const float* __restrict__ src,
                                float* __restrict__ dst, int axis, int img0, int force_k1) {
    const int b = img0 + blockIdx.y;
    const PostDesc im = imgs[b];
    const int K = force_k1 ? 1 : im.K;
    const int lo = L.idbase[b], hi = L.idbase[b + 1];
    const size_t vbase = force_k1 ? (size_t)lo : im.voff[L.which];
    const int total = (hi - lo) * K;
    const int* n1 = L.n1 + (size_t)axis * L.cap;
    const int* n2 = L.n2 + (size_t)axis * L.cap;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        const int idl = t / K, k = t - idl * K;
        const int a = n1[lo + idl], c = n2[lo + idl];
        const float va = a >= 0 ? src[vbase + (size_t)(a - lo) * K + k] : 0.f;
        const float vc = c >= 0 ? src[vbase + (size_t)(c - lo) * K + k] : 0.f;
        const float old = src[vbase + (size_t)idl * K + k];
        dst[vbase + (size_t)idl * K + k] = (float)__dadd_rn((double)old, __dmul_rn(0.5, (double)__fadd_rn(va, vc)));
    }
}

// slice + symmetric normalisation + Potts compatibility, accumulated into the mean-field sum:
//   f = norm * alpha-scaled slice ;  tmp = (first ? -U : tmp) - (-w * f)
// mode 2: write norm[pixel] = 1 / sqrt(slice + 1e-20)  (lattice applied to the ones vector)
__global__ void crf_slice_kernel(const CrfLattice L, const PostDesc* __restrict__ imgs, const float* __restrict__ val,
                                 const float* __restrict__ norm, const float* __restrict__ unary, float* __restrict__ tmp,
                                 float* __restrict__ norm_out, float w, float alpha, int first, int img0, int mode) {
    const int b = img0 + blockIdx.y;
    const PostDesc im = imgs[b];
    const int K = mode == 2 ? 1 : im.K;
    const int lo = L.idbase[b];
    const size_t vbase = mode == 2 ? (size_t)lo : im.voff[L.which];
    const int n = im.H * im.W;
    const int total = n * K;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        const int pix = t / K, k = t - pix * K;
        const size_t pv0 = (size_t)(im.pix0 + pix) * L.D1;
        float out = 0.f;
        for (int v = 0; v < L.D1; v++) {
            const int o = L.offset[pv0 + v];
            const float wv = L.bary[pv0 + v];
            out = __fadd_rn(out, __fmul_rn(__fmul_rn(wv, val[vbase + (size_t)(o - lo) * K + k]), alpha));
        }
        if (mode == 2) {
            norm_out[im.pix0 + pix] = (float)(1.0 / sqrt((double)out + 1e-20));
        } else {
            const float f = __fmul_rn(-w, __fmul_rn(out, norm[im.pix0 + pix]));
            const size_t qi = im.off + (size_t)pix * K + k;
            const float base = first ? -unary[qi] : tmp[qi];
            tmp[qi] = __fsub_rn(base, f);
        }
    }
}

// Q = exp(x - max) / sum over the K labels of each pixel (x = -U when `neg`), NaN-propagating max
__global__ void crf_softmax_kernel(const PostDesc* __restrict__ imgs, const float* __restrict__ x, float* __restrict__ Q,
                                   int neg, int img0) {
    const int b = img0 + blockIdx.y;
    const PostDesc im = imgs[b];
    const int n = im.H * im.W, K = im.K;
    for (int pix = blockIdx.x * blockDim.x + threadIdx.x; pix < n; pix += gridDim.x * blockDim.x) {
        const float* xi = x + im.off + (size_t)pix * K;
        float* qi = Q + im.off + (size_t)pix * K;
        float m = neg ? -xi[0] : xi[0];
        for (int k = 1; k < K; k++) {
            const float v = neg ? -xi[k] : xi[k];
            if (v > m || v != v) m = v;
        }
        float s = 0.f;
        for (int k = 0; k < K; k++) {
            const float e = pnp_expf(__fsub_rn(neg ? -xi[k] : xi[k], m));
            qi[k] = e;
            s = __fadd_rn(s, e);
        }
        for (int k = 0; k < K; k++) qi[k] = __fdiv_rn(qi[k], s);
    }
}

// ------------------------------------------------------------------------------------------ host
static inline int ok() { return hipGetLastError() == hipSuccess ? PNP_OK : PNP_ERR_HIP; }

size_t crf_sort_temp_bytes(size_t max_entries, int max_images) {
    size_t a = 0, b = 0;
    (void)hipcub::DeviceSegmentedRadixSort::SortPairs(nullptr, a, (const uint64_t*)nullptr, (uint64_t*)nullptr,
                                                (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)max_entries, max_images,
                                                (const int*)nullptr, (const int*)nullptr, 0, 64, 0);
    (void)hipcub::DeviceScan::InclusiveSum(nullptr, b, (const int*)nullptr, (int*)nullptr, (int)max_entries, 0);
    return (a > b ? a : b) + 256;
}

// Build one lattice (D = 2: Gaussian xy/sxy ; D = 5: bilateral xy/sxy, rgb/srgb) for images [0,B).
// Scratch arrays (keys/vals double buffers, head, incl, temp) are caller-provided.
int crf_build_lattice(int D, const CrfLattice& L, const PostDesc* d_imgs, const uint8_t* d_rgb, float sxy, float srgb,
                      int B, size_t ent_total, int max_pixels, const int* d_seg_begin, const int* d_seg_end,
                      uint64_t* keys_a, uint64_t* keys_b, uint32_t* vals_a, int* head, int* incl, void* temp,
                      size_t temp_bytes, int* d_range_err, hipStream_t s) {
    const int nb = (max_pixels + 255) / 256 < 512 ? (max_pixels + 255) / 256 : 512;
    if (D == 2)
        hipLaunchKernelGGL((lattice_embed_kernel<2>), dim3(nb, B), dim3(256), 0, s, d_imgs, d_rgb, sxy, srgb, L.bary, keys_a, vals_a, d_range_err);
    else if (D == 5)
        hipLaunchKernelGGL((lattice_embed_kernel<5>), dim3(nb, B), dim3(256), 0, s, d_imgs, d_rgb, sxy, srgb, L.bary, keys_a, vals_a, d_range_err);
    else
        return PNP_ERR_ARG;
    const int bits = D == 2 ? 32 : 60;
    size_t tb = temp_bytes;
    if (hipcub::DeviceSegmentedRadixSort::SortPairs(temp, tb, keys_a, keys_b, vals_a, L.vals, (int)ent_total, B, d_seg_begin,
                                                    d_seg_end, 0, bits, s) != hipSuccess)
        return PNP_ERR_HIP;
    const int nbe = 1024;
    hipLaunchKernelGGL(mark_heads_kernel, dim3(nbe, B), dim3(256), 0, s, keys_b, d_imgs, D + 1, head);
    tb = temp_bytes;
    if (hipcub::DeviceScan::InclusiveSum(temp, tb, head, incl, (int)ent_total, s) != hipSuccess) return PNP_ERR_HIP;
    hipLaunchKernelGGL(scatter_ids_kernel, dim3(nbe, B), dim3(256), 0, s, keys_b, L.vals, head, incl, d_imgs, D + 1, B,
                       ent_total, L.offset, L.seg_start, L.ukeys, L.idbase);
    if (D == 2)
        hipLaunchKernelGGL((neighbors_kernel<2>), dim3(nbe, B), dim3(256), 0, s, L.ukeys, L.idbase, L.cap, L.n1, L.n2);
    else
        hipLaunchKernelGGL((neighbors_kernel<5>), dim3(nbe, B), dim3(256), 0, s, L.ukeys, L.idbase, L.cap, L.n1, L.n2);
    return ok();
}

static float crf_alpha(int D) { return 1.0f / (1 + powf(2, (float)-D)); }

// norm = 1 / sqrt(lattice(ones) + 1e-20) for images [0,B); va/vb: scratch of >= cap floats each.
int crf_lattice_norm(const CrfLattice& L, const PostDesc* d_imgs, int B, float* va, float* vb, float* norm_out,
                     hipStream_t s) {
    const int D = L.D1 - 1;
    const int nb = 1024;
    hipLaunchKernelGGL(crf_splat_kernel, dim3(nb, B), dim3(256), 0, s, L, d_imgs, (const float*)nullptr,
                       (const float*)nullptr, va, 0, 1);
    float* src = va;
    float* dst = vb;
    for (int j = 0; j <= D; j++) {
        hipLaunchKernelGGL(crf_blur_kernel, dim3(nb, B), dim3(256), 0, s, L, d_imgs, src, dst, j, 0, 1);
        float* t = src;
        src = dst;
        dst = t;
    }
    hipLaunchKernelGGL(crf_slice_kernel, dim3(nb, B), dim3(256), 0, s, L, d_imgs, src, (const float*)nullptr,
                       (const float*)nullptr, (float*)nullptr, norm_out, 0.f, crf_alpha(D), 0, 0, 2);
    return ok();
}

// One pairwise term of one mean-field iteration for images [img0, img0+nimg):
//   tmp = (first ? -U : tmp) - (-w) * norm * lattice(norm * Q)
int crf_pairwise(const CrfLattice& L, const PostDesc* d_imgs, int img0, int nimg, const float* Q, const float* norm,
                 const float* unary, float* tmp, float* va, float* vb, float w, int first, hipStream_t s) {
    const int D = L.D1 - 1;
    const int nb = 2048;
    hipLaunchKernelGGL(crf_splat_kernel, dim3(nb, nimg), dim3(256), 0, s, L, d_imgs, Q, norm, va, img0, 0);
    float* src = va;
    float* dst = vb;
    for (int j = 0; j <= D; j++) {
        hipLaunchKernelGGL(crf_blur_kernel, dim3(nb, nimg), dim3(256), 0, s, L, d_imgs, src, dst, j, img0, 0);
        float* t = src;
        src = dst;
        dst = t;
    }
    hipLaunchKernelGGL(crf_slice_kernel, dim3(nb, nimg), dim3(256), 0, s, L, d_imgs, src, norm, unary, tmp,
                       (float*)nullptr, w, crf_alpha(D), first, img0, 0);
    return ok();
}

int crf_softmax(const PostDesc* d_imgs, int img0, int nimg, const float* x, float* Q, int neg, int max_pixels,
                hipStream_t s) {
    const int nb = (max_pixels + 255) / 256 < 512 ? (max_pixels + 255) / 256 : 512;
    hipLaunchKernelGGL(crf_softmax_kernel, dim3(nb, nimg), dim3(256), 0, s, d_imgs, x, Q, neg, img0);
    return ok();
}

}  // namespace pnp
