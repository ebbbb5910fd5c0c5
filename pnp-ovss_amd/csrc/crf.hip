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
// (oracle/densecrf_ref.c).  All images of a batch go through every kernel together (build kernels:
// blockIdx.y = image; iteration kernels: whole images per XCD, grids sized to what is resident).
// Nothing is reshaped into a GEMM: the iteration kernels stream 176..608-byte value rows through gathers and are
// bound by memory latency x occupancy and by vector-memory issue, so they are written around
// their dependent-load chains: contributor / neighbour / slice records, requests of the next item behind the
// gathers of the current one, results stored one item later.
#include <mutex>

#include <stdlib.h>

#include "common.h"
#include "kernels.h"
#include "../../include/pnp_math.h"

namespace pnp {

// Streaming accesses of the mean-field kernels (data with no reuse inside a launch: contributor records, the value rows a splat
// writes, the unary rows an update reads and the Q rows it writes) are marked non-temporal, so that they do not displace the
// rows that ARE reused (a pixel's Q row is gathered by its 3 + 6 lattice points) from L2.  Round 5, same-box A/B
// (tools/crf_nt_ab.sh, two alternations): mean-field per step 30.9 / 30.6 -> 30.3 / 30.6 ms on the headline batch (neutral),
// 74.1 / 75.0 -> 72.5 / 72.9 ms at K = 59, 72.3 / 72.3 -> 67.3 / 71.1 ms at 3.6 lattice points per pixel, 252 / 257 -> 253 / 253 ms
// at ADE20K size: small, never negative; the stores of the two-axis lattice blur on top of that: 30.6 / 30.8 -> 30.3 / 29.6 ms
// (headline), 74.8 / 74.2 -> 73.1 / 73.1 (K = 59), 71.6 / 71.5 -> 69.7 / 66.6 (3.6 points per pixel).  Loads / stores are otherwise
// identical: results unchanged.
#ifdef PNP_CRF_NO_NT            // A/B opt-out (tools/crf_nt_ab.sh builds its `base` variant with it): plain accesses
template <typename T> __device__ __forceinline__ T ld_stream(const T* p) { return *p; }
template <typename T> __device__ __forceinline__ void st_stream(T* p, const T& v) { *p = v; }
#else
template <typename T> __device__ __forceinline__ T ld_stream(const T* p) { return __builtin_nontemporal_load(p); }
template <typename T> __device__ __forceinline__ void st_stream(T* p, const T& v) { __builtin_nontemporal_store(v, p); }
#endif
__device__ __forceinline__ CrfEntry ld_entry(const CrfEntry* p) {
    return __builtin_bit_cast(CrfEntry, ld_stream(reinterpret_cast<const chunk16*>(p)));
}

template <int D> struct KeyPack;
template <> struct KeyPack<2> { static constexpr int BITS = 16; };
template <> struct KeyPack<5> { static constexpr int BITS = 11; };
// The image index of the batch sits above the coordinate bits (bit IMG_SHIFT..).  It is NOT sorted on: the entries arrive
// image by image, so crf_build_lattice runs sort.hip's stable radix sort segmented by image over the coordinate bits alone;
// the image bits stay in the keys for what reads them afterwards (mark_heads: a run never spans two images; the neighbour
// search compares whole keys).  5 x 11 + 6 bits <= 61: at most 64 images per prepared batch.
constexpr int IMG_SHIFT = 55;
constexpr int IMG_BITS = 6;

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
        c[i] = (int)(k & ((1 << BITS) - 1)) - (1 << (BITS - 1));      // image bits above D*BITS are ignored
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
            keys[e0 + rem] = pack_key<D>(key) | ((uint64_t)b << IMG_SHIFT);
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

// blur neighbours along each of the d+1 lattice axes: n1 = key - 1 (coord j: + d), n2 = key + 1 (coord j: - d).
// A neighbour's key is the point's key plus a constant of the axis (the fields are positional digits and a key whose neighbour
// would leave a field's range is flagged absent), so along the sorted unique keys of an image the neighbour positions are
// monotone: a thread takes a RUN of consecutive points of one axis, finds the first neighbour by binary search (17 dependent
// loads at 100 k points) and the others by galloping from the previous position (2-3 loads).
constexpr int kNbrRun = 8;
__device__ __forceinline__ int lower_bound_from(const uint64_t* __restrict__ ukeys, int from, int hi, uint64_t key) {
    // first position in [from, hi) whose key is >= key, given that every position before `from` holds a smaller key
    int l = from, step = 1;
    while (l + step <= hi && ukeys[l + step - 1] < key) {      // gallop: the answer is beyond l + step - 1
        l += step;
        step <<= 1;
    }
    int h = l + step - 1 < hi ? l + step - 1 : hi;             // answer in [l, h]
    while (l < h) {
        const int mid = (l + h) >> 1;
        if (ukeys[mid] < key) l = mid + 1;
        else h = mid;
    }
    return l;
}
template <int D>
__global__ void neighbors_kernel(const uint64_t* __restrict__ ukeys, const int* __restrict__ idbase, size_t cap,
                                 int* __restrict__ n1, int* __restrict__ n2) {
    constexpr int BITS = KeyPack<D>::BITS;
    const int b = blockIdx.y;
    const int lo = idbase[b], hi = idbase[b + 1];
    const int nruns = (hi - lo + kNbrRun - 1) / kNbrRun;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < nruns * (D + 1); t += gridDim.x * blockDim.x) {
        const int run = t / (D + 1), j = t % (D + 1);
        const int id0 = lo + run * kNbrRun, id1 = id0 + kNbrRun < hi ? id0 + kNbrRun : hi;
        int p1 = lo, p2 = lo;                                   // positions reached so far (nothing before them can match)
        bool first1 = true, first2 = true;
        for (int id = id0; id < id1; id++) {
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
            int r1 = -1, r2 = -1;
            bool in1 = true, in2 = true;
#pragma unroll
            for (int k = 0; k < D; k++) {
                in1 &= (a[k] >= -(1 << (BITS - 1)) && a[k] < (1 << (BITS - 1)));
                in2 &= (d2[k] >= -(1 << (BITS - 1)) && d2[k] < (1 << (BITS - 1)));
            }
            if (in1) {
                const uint64_t k1 = pack_key<D>(a) | ((uint64_t)b << IMG_SHIFT);
                int l;
                if (first1) {                                   // binary search inside this image's sorted unique keys [lo, hi)
                    int ll = lo, h = hi;
                    while (ll < h) {
                        const int mid = (ll + h) >> 1;
                        if (ukeys[mid] < k1) ll = mid + 1;
                        else h = mid;
                    }
                    l = ll;
                    first1 = false;
                } else {
                    l = lower_bound_from(ukeys, p1, hi, k1);
                }
                p1 = l;
                if (l < hi && ukeys[l] == k1) r1 = l;
            }
            if (in2) {
                const uint64_t k2 = pack_key<D>(d2) | ((uint64_t)b << IMG_SHIFT);
                int l;
                if (first2) {
                    int ll = lo, h = hi;
                    while (ll < h) {
                        const int mid = (ll + h) >> 1;
                        if (ukeys[mid] < k2) ll = mid + 1;
                        else h = mid;
                    }
                    l = ll;
                    first2 = false;
                } else {
                    l = lower_bound_from(ukeys, p2, hi, k2);
                }
                p2 = l;
                if (l < hi && ukeys[l] == k2) r2 = l;
            }
            n1[(size_t)j * cap + id] = r1;
            n2[(size_t)j * cap + id] = r2;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Scalar (K = 1) lattice pass used once per batch for the normaliser: norm = 1/sqrt(L(1) + 1e-20).
// One element of a lattice axis blur, densecrf's `old + 0.5 * (n1 + n2)` (double literal): (n1 + n2) is a float sum; the
// product by 0.5 is exact; and the double sum old + 0.5 t, rounded to float, equals the single rounding of the exact sum
// (it is exact in double whenever the exponents are within 28, and beyond that the small term is below a quarter ulp of the
// float result on either path) -- i.e. one float fma.  Two VALU instructions per element instead of two conversions, an
// fp64 multiply, an fp64 add and a conversion back; bit-identical to the oracle's double form (tests: GPU == oracle).
__device__ __forceinline__ float crf_axis(float old, float n1, float n2) { return __fmaf_rn(0.5f, __fadd_rn(n1, n2), old); }

__global__ void crf_splat1_kernel(const CrfLattice L, const PostDesc* __restrict__ imgs, float* __restrict__ val) {
    const int b = blockIdx.y;
    const int lo = L.idbase[b], hi = L.idbase[b + 1];
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < hi - lo; t += gridDim.x * blockDim.x) {
        const int id = lo + t;
        const int e0 = L.seg_lo[id], e1 = L.seg_hi[id];
        float acc = 0.f;
        for (int e = e0; e < e1; e++) acc = __fadd_rn(acc, L.bary[L.vals[e]]);
        val[id] = acc;
    }
}

__global__ void crf_blur1_kernel(const CrfLattice L, const PostDesc* __restrict__ imgs, const float* __restrict__ src,
                                 float* __restrict__ dst, int axis) {
    const int b = blockIdx.y;
    const int lo = L.idbase[b], hi = L.idbase[b + 1];
    const int* n1 = L.n1 + (size_t)axis * L.cap;
    const int* n2 = L.n2 + (size_t)axis * L.cap;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < hi - lo; t += gridDim.x * blockDim.x) {
        const int id = lo + t;
        const int a = n1[id], c = n2[id];
        const float va = a >= 0 ? src[a] : 0.f;
        const float vc = c >= 0 ? src[c] : 0.f;
        dst[id] = crf_axis(src[id], va, vc);
    }
}

__global__ void crf_norm_kernel(const CrfLattice L, const PostDesc* __restrict__ imgs, const float* __restrict__ val,
                                float* __restrict__ norm_out, float alpha) {
    const int b = blockIdx.y;
    const PostDesc im = imgs[b];
    const int n = im.H * im.W;
    for (int pix = blockIdx.x * blockDim.x + threadIdx.x; pix < n; pix += gridDim.x * blockDim.x) {
        const size_t pv0 = (size_t)(im.pix0 + pix) * L.D1;
        float out = 0.f;
        for (int v = 0; v < L.D1; v++)
            out = __fadd_rn(out, __fmul_rn(__fmul_rn(L.bary[pv0 + v], val[L.offset[pv0 + v]]), alpha));
        norm_out[im.pix0 + pix] = (float)(1.0 / sqrt((double)out + 1e-20));
    }
}

// contributor records get the normaliser of their pixel (the splat multiplies Q * norm before the barycentric weight)
__global__ void entry_norm_kernel(CrfEntry* __restrict__ ent, size_t n, const float* __restrict__ norm) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        ent[i].nr = norm[ent[i].pixel];
}

// ------------------------------------------------------------------------------------------
// Mean-field iteration kernels.  Values / Q / unary rows are padded to Kp = 4*ceil(K/4) floats so
// every access is one 16-byte vector per lane (pad channels carry zeros and are never read back as
// labels).  Arithmetic per channel is exactly the scalar sequence of the oracle.
//
// Work distribution of the iteration kernels: workgroups with the same blockIdx % 8 share an XCD
// (round-robin dispatch; a speed assumption only), so each XCD is given whole images (b = xcd,
// xcd + 8, ...) and sweeps them one after another: the ~10 MB value rows of the image an XCD is
// working on are re-read (neighbour / contributor gathers) out of L2 / Infinity Cache instead of HBM.
// Inside an image LPP lanes share one lattice point (one 16-byte channel chunk each); LPP = 8, 16 or 32 is the
// smallest that covers the widest row of the batch, so the contributor list of a point is walked once
// (with LPP = 8 a 36-float row walked it twice: +43 % on the whole mean-field pass).
//
// Image schedule of an XCD: whole images img0 + xcd, + 8, ... while a full round of 8 remains; the last nimg % 8 images
// are cut into 8 equal slices, one per XCD (with 35 images three XCDs would otherwise sweep a fifth image while five
// idle: 12.5 % of every iteration kernel).
__device__ __forceinline__ bool xcd_work(int i, int xcd, int img0, int nimg, int& b, int& part, int& parts) {
    const int full = nimg >> 3;
    if (i < full) {
        b = img0 + i * 8 + xcd;
        part = 0;
        parts = 1;
        return true;
    }
    if (i - full < (nimg & 7)) {
        b = img0 + full * 8 + (i - full);
        part = xcd;
        parts = 8;
        return true;
    }
    return false;
}

// splat: val[id] = sum over the lattice point's contributors (ascending pixel) of bary * (Q * norm).
// LPP lanes share one lattice point (one 16-byte channel chunk each).  The contributor records of the point -- (pixel, bary,
// norm) triples stored in list order by the lattice build -- are loaded LPP at a time, one 16-byte record per lane (one
// coalesced load instead of a dependent index -> weight -> norm chain per contributor), handed round the group with lane
// shuffles, and up to eight Q-row gathers are in flight per lane before the ordered accumulation.  The kernel is bound by
// memory latency x occupancy, not by bytes, so the chain segment -> records -> rows is software-pipelined over the
// grid-stride loop: while the rows of point t are gathered, the records of point t+1 and the segment of point t+2 are
// already in flight (one exposed latency per point instead of three).  Per-channel arithmetic and order are those of the
// sequential CPU algorithm (oracle/densecrf_ref.c::lattice_compute).  WHICH only names the instantiation in profiles.
template <int LPP, int WHICH, bool MULTI>
__global__ __launch_bounds__(256) void crf_splat_kernel(const CrfLattice L, const PostDesc* __restrict__ imgs,
                                                        const float* __restrict__ Q, float* __restrict__ val, int img0, int nimg) {
    constexpr int PPB = 256 / LPP;                       // lattice points per workgroup pass
    constexpr int G = 8;                                 // row gathers in flight per lane
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, bpx = gridDim.x >> 3;
    const int c0 = threadIdx.x & (LPP - 1), pl = threadIdx.x / LPP;
    const int stride = bpx * PPB;
    const CrfEntry none = {0u, 0.f, 0.f, 0u};
    int b, part, parts;
    for (int wi = 0; xcd_work(wi, xcd, img0, nimg, b, part, parts); wi++) {
        const PostDesc im = imgs[b];
        const int K4 = im.Kp >> 2;
        const int lo = L.idbase[b], hi = L.idbase[b + 1];
        // rows are addressed as base + 32-bit byte offset (an image's Q block is < 4 GB; pixels per image < 2^24)
        const char* const Qb = reinterpret_cast<const char*>(Q + im.qoff);
        const uint32_t rowb = (uint32_t)K4 * 16u;
        f32x4* V4 = reinterpret_cast<f32x4*>(val + im.voff[L.which]);
        const int r0 = (int)((long)(hi - lo) * part / parts), r1 = (int)((long)(hi - lo) * (part + 1) / parts);
        int idl = r0 + slot * PPB + pl;
        // pipeline prologue: segment of points t and t+1, first record chunk of point t
        int e0 = 0, e1 = 0, e0n = 0, e1n = 0;
        if (idl < r1) {
            e0 = L.seg_lo[lo + idl];
            e1 = L.seg_hi[lo + idl];
        }
        if (idl + stride < r1) {
            e0n = L.seg_lo[lo + idl + stride];
            e1n = L.seg_hi[lo + idl + stride];
        }
        CrfEntry first = none;
        if (c0 < e1 - e0) first = ld_entry(L.ent + e0 + c0);
        f32x4 prev_acc = {0.f, 0.f, 0.f, 0.f};
        size_t prev_at = 0;
        bool have_prev = false;
        // up to G row gathers of a record chunk (records j0 .. of `mine`, n valid), then their ordered accumulation
        auto gather = [&](const CrfEntry& mine, int j0, int n, uint32_t cofs, f32x4* in) {
#pragma unroll
            for (int j = 0; j < G; j++) {
                if (j == 0 || j0 + j < n) {
                    const uint32_t px = (uint32_t)__shfl((int)mine.pixel, j0 + j, LPP) - (uint32_t)im.pix0;
                    in[j] = *reinterpret_cast<const f32x4*>(Qb + (__umul24(px, rowb) + cofs));
                }
            }
        };
        auto accumulate = [&](const CrfEntry& mine, int j0, int n, const f32x4* in, f32x4& acc) {
#pragma unroll
            for (int j = 0; j < G; j++) {
                if (j == 0 || j0 + j < n) {
                    const float w = __shfl(mine.w, j0 + j, LPP), nr = __shfl(mine.nr, j0 + j, LPP);
#pragma unroll
                    for (int i = 0; i < 4; i++) acc[i] = __fadd_rn(acc[i], __fmul_rn(w, __fmul_rn(in[j][i], nr)));
                }
            }
        };
        for (; idl < r1; idl += stride) {
            // segment of point t+2 and first records of point t+1 are requested right behind the first gather group of point
            // t (straight-line code, nothing pending in front of it), so the three latencies overlap: the compiler's wait in
            // front of the accumulation covers them all
            int e0nn = 0, e1nn = 0;
            CrfEntry firstn = none;
            auto prefetch = [&]() {
                if (idl + 2 * stride < r1) {
                    e0nn = L.seg_lo[lo + idl + 2 * stride];
                    e1nn = L.seg_hi[lo + idl + 2 * stride];
                }
                if (c0 < e1n - e0n) firstn = ld_entry(L.ent + e0n + c0);
            };
            if (MULTI) prefetch();
            // MULTI: rows wider than LPP chunks take several passes over the point's records (cb loop; the prefetch then sits
            // in front of the loop and is waited for first); the common case is one straight-line pass
            for (int cb = 0; cb < (MULTI ? K4 : 1); cb += LPP) {
                const uint32_t cofs = (uint32_t)(cb + c0 < K4 ? cb + c0 : K4 - 1) * 16u;   // lanes past the row keep a valid chunk (they carry records)
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                f32x4 in[G];
                int n = e1 - e0 < LPP ? e1 - e0 : LPP;                    // (the same in all lanes of the group)
                gather(first, 0, n, cofs, in);
                if (!MULTI) {
                    prefetch();
                    if (have_prev) st_stream(V4 + prev_at, prev_acc);     // the previous point's row, stored under this point's gathers
                    asm volatile("" ::: "memory");                        // (keeps every request above in front of the wait below)
                }
                accumulate(first, 0, n, in, acc);
                for (int j0 = G; j0 < n; j0 += G) {
                    gather(first, j0, n, cofs, in);
                    accumulate(first, j0, n, in, acc);
                }
                for (int eb = e0 + LPP; eb < e1; eb += LPP) {
                    n = e1 - eb < LPP ? e1 - eb : LPP;
                    const CrfEntry mine = c0 < n ? ld_entry(L.ent + eb + c0) : none;
                    for (int j0 = 0; j0 < n; j0 += G) {
                        gather(mine, j0, n, cofs, in);
                        accumulate(mine, j0, n, in, acc);
                    }
                }
                if (MULTI) {
                    if (cb + c0 < K4) st_stream(V4 + (size_t)idl * K4 + cb + c0, acc);
                } else {                                                  // stored one point later (a store waited for at the loop
                    prev_acc = acc;                                       // head would expose its acknowledge latency)
                    prev_at = (size_t)idl * K4 + c0;
                    have_prev = c0 < K4;
                }
            }
            e0 = e0n; e1 = e1n; e0n = e0nn; e1n = e1nn;
            first = firstn;
        }
        if (have_prev) st_stream(V4 + prev_at, prev_acc);
    }
}

// one axis of the lattice blur: new = old + 0.5 * (n1 + n2), absent neighbours contribute 0.
// Work items are (lattice point, 16-byte channel chunk) pairs in linear order, so a wave touches 1 KB of
// consecutive value bytes whatever Kp is (no idle lanes at Kp = 24), and every thread keeps two items -- six
// independent value gathers behind four index loads -- in flight.  (The same item order made the splat
// slower: its lanes then diverge on the contributor counts of more points per wave.)
__global__ __launch_bounds__(256) void crf_blur4_kernel(const CrfLattice L, const PostDesc* __restrict__ imgs,
                                                        const float* __restrict__ src, float* __restrict__ dst, int axis,
                                                        int img0, int nimg) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, bpx = gridDim.x >> 3;
    const int* n1 = L.n1 + (size_t)axis * L.cap;
    const int* n2 = L.n2 + (size_t)axis * L.cap;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    int b, part, parts;
    for (int wi = 0; xcd_work(wi, xcd, img0, nimg, b, part, parts); wi++) {
        const PostDesc im = imgs[b];
        const int K4 = im.Kp >> 2;
        const int lo = L.idbase[b], hi = L.idbase[b + 1];
        const f32x4* S4 = reinterpret_cast<const f32x4*>(src + im.voff[L.which]);
        f32x4* D4 = reinterpret_cast<f32x4*>(dst + im.voff[L.which]);
        const int first = (int)((long)(hi - lo) * part / parts) * K4;
        const int nitem = (int)((long)(hi - lo) * (part + 1) / parts) * K4, stride = bpx * 256;
        for (int it0 = first + slot * 256 + threadIdx.x; it0 < nitem; it0 += 2 * stride) {
            const int it1 = it0 + stride;
            const bool two = it1 < nitem;
            const int p0 = it0 / K4, p1 = two ? it1 / K4 : p0;
            const int a0 = n1[lo + p0], d0 = n2[lo + p0], a1 = n1[lo + p1], d1 = n2[lo + p1];
            const int c0 = it0 - p0 * K4, c1 = (two ? it1 : it0) - p1 * K4;
            const f32x4 va0 = a0 >= 0 ? S4[(size_t)(a0 - lo) * K4 + c0] : zero;
            const f32x4 vd0 = d0 >= 0 ? S4[(size_t)(d0 - lo) * K4 + c0] : zero;
            const f32x4 old0 = S4[it0];
            const f32x4 va1 = a1 >= 0 ? S4[(size_t)(a1 - lo) * K4 + c1] : zero;
            const f32x4 vd1 = d1 >= 0 ? S4[(size_t)(d1 - lo) * K4 + c1] : zero;
            const f32x4 old1 = S4[two ? it1 : it0];
            f32x4 o0, o1;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                o0[i] = crf_axis(old0[i], va0[i], vd0[i]);
                o1[i] = crf_axis(old1[i], va1[i], vd1[i]);
            }
            D4[it0] = o0;
            if (two) D4[it1] = o1;
        }
    }
}

// Two consecutive axes of the lattice blur in one pass: out = B_{axis+1}(B_axis(src)).  The intermediate B_axis(src) is
// recomputed at the point and at its two axis+1 neighbours with exactly the single-axis arithmetic (each rounded to fp32 as
// the stored intermediate would be), so the result equals two crf_blur4 passes bit for bit while the value array is
// streamed from / to HBM once instead of twice; the 8 neighbour rows of a point sit close to it in the spatial numbering
// and are served by L2.  Same (point, 16-byte chunk) item order and XCD schedule as crf_blur4.
__device__ __forceinline__ f32x4 crf_blur1(const f32x4& old, const f32x4& va, const f32x4& vd) {
    f32x4 o;
#pragma unroll
    for (int i = 0; i < 4; i++) o[i] = crf_axis(old[i], va[i], vd[i]);
    return o;
}
// The eight neighbour ids of a point for the axis pair (2p, 2p+1) come from one 32-byte record (CrfNbr8, image-local ids,
// built once per lattice) instead of two dependent rounds of index loads; rows are addressed with 32-bit byte offsets; the
// record of the thread's next item is requested behind the row gathers of the current one and the previous result is
// stored there too, so an item exposes one memory latency (the kernel is bound by latency x occupancy and by the 64 B/clk
// of the CU's vector memory path -- nine rows in, one out -- not by HBM bytes).
__global__ __launch_bounds__(256) void crf_blur4x2_kernel(const CrfLattice L, const PostDesc* __restrict__ imgs,
                                                          const float* __restrict__ src, float* __restrict__ dst, int pair,
                                                          int img0, int nimg) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, bpx = gridDim.x >> 3;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    int b, part, parts;
    for (int wi = 0; xcd_work(wi, xcd, img0, nimg, b, part, parts); wi++) {
        const PostDesc im = imgs[b];
        const int K4 = im.Kp >> 2;
        const int lo = L.idbase[b], hi = L.idbase[b + 1];
        const char* const Sb = reinterpret_cast<const char*>(src + im.voff[L.which]);
        f32x4* const D4 = reinterpret_cast<f32x4*>(dst + im.voff[L.which]);
        const CrfNbr8* const T = L.nbr8 + (size_t)pair * L.cap + lo;
        const uint32_t rowb = (uint32_t)K4 * 16u;
        const int first = (int)((long)(hi - lo) * part / parts) * K4;
        const int nitem = (int)((long)(hi - lo) * (part + 1) / parts) * K4, stride = bpx * 256;
        const int sq = stride / K4, sr = stride - sq * K4;              // item -> (point, chunk) advances without a division
        int it = first + slot * 256 + threadIdx.x;
        int p = it / K4, c = it - p * K4;
        CrfNbr8 t = {-1, -1, -1, -1, -1, -1, -1, -1};
        if (it < nitem) t = T[p];
        f32x4 prev = zero;
        int prev_it = -1;
        for (; it < nitem; it += stride) {
            const uint32_t cofs = (uint32_t)c * 16u;
#define PNP_ROW(x) ((x) >= 0 ? *reinterpret_cast<const f32x4*>(Sb + (__umul24((uint32_t)(x), rowb) + cofs)) : zero)
            const f32x4 r_i = *reinterpret_cast<const f32x4*>(Sb + (__umul24((uint32_t)p, rowb) + cofs));
            const f32x4 r_ia = PNP_ROW(t.ia), r_id = PNP_ROW(t.id);
            const f32x4 r_a = PNP_ROW(t.pa), r_aa = PNP_ROW(t.aa), r_ad = PNP_ROW(t.ad);
            const f32x4 r_d = PNP_ROW(t.pd), r_da = PNP_ROW(t.da), r_dd = PNP_ROW(t.dd);
#undef PNP_ROW
            // next item's neighbour record and the previous item's result go out behind the gathers
            int pn = p + sq, cn = c + sr;
            if (cn >= K4) {
                cn -= K4;
                pn++;
            }
            CrfNbr8 tn = {-1, -1, -1, -1, -1, -1, -1, -1};
            if (it + stride < nitem) tn = T[pn];
            if (prev_it >= 0) st_stream(D4 + prev_it, prev);   // (streamed: the next pass reads it back from beyond L2 either way)
            asm volatile("" ::: "memory");
            const f32x4 m_i = crf_blur1(r_i, r_ia, r_id);
            const f32x4 m_a = t.pa >= 0 ? crf_blur1(r_a, r_aa, r_ad) : zero;
            const f32x4 m_d = t.pd >= 0 ? crf_blur1(r_d, r_da, r_dd) : zero;
            prev = crf_blur1(m_i, m_a, m_d);
            prev_it = it;
            t = tn;
            p = pn;
            c = cn;
        }
        if (prev_it >= 0) st_stream(D4 + prev_it, prev);
    }
}

// Fused tail of a mean-field iteration for a tile of up to 256 pixels:
//   t = -U - (-w_g * norm_g * slice_g) - (-w_b * norm_b * slice_b)   (both lattices already blurred)
//   Q = exp(t - max_k t) / sum_k          (per pixel, staged through LDS so global I/O stays 16-B wide)
// pairwise == 0: Q = softmax(-U) (the initial marginals).
// A tile is a TW x TH BLOCK of pixels, not a run of a raster line: the simplex vertices of a pixel are shared with its
// neighbours in both directions (Gaussian cells are 3 px wide, lattice points are numbered along a Z-order curve), so a block
// touches about half the distinct value rows of a strip of the same size and the slice gathers hit L1 / L2 accordingly.
constexpr int CRF_TP = 256;
// channels per group from which the update's softmax runs wave-parallel (see there).  Measured (tools/cfg_kstats.sh): K = 150 (32-pixel
// tiles, 64 rows: the thread-per-row form keeps ONE wave busy) 1350 -> 1169 us per launch; K = 81 / 59 (64 / 128 rows) 1654 -> 2514 /
// 1947 -> 3029 us -- a row's dependent chain (LDS read, six shuffles, exponential, store) is too long where rows are many and short
constexpr int CRF_WAVE_SOFTMAX_K = 128;
__host__ __device__ inline int crf_tile_w(int tp) { return tp >= 128 ? 16 : tp >= 32 ? 8 : 4; }
// WIDE (rows of >= CRF_WAVE_SOFTMAX_K channels per group): the wave-parallel softmax below; such tiles are LDS-bound to three
// workgroups per CU, so the kernel may use 128 registers (the narrow form is held to 96 for five waves per SIMD)
template <bool WIDE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WIDE ? 4 : 5))) void crf_update_kernel(const CrfLattice Lg, const CrfLattice Lb, const PostDesc* __restrict__ imgs,
                                                         const float* __restrict__ vg, const float* __restrict__ vb,
                                                         const float* __restrict__ norm_g, const float* __restrict__ norm_b,
                                                         const float* __restrict__ unary, float* __restrict__ Q, float w_g,
                                                         float w_b, float alpha_g, float alpha_b, int pairwise, int img0,
                                                         int nimg, int tp_cap, int tile_w, const CrfLabelOut lout) {
    extern __shared__ __attribute__((aligned(16))) float tile[];        // [TP][Kp + 1], then the slice records [TP][20]
    // XCD-affine sweep like the splat / blur kernels: the slice gathers of an image hit the value rows
    // its XCD has just blurred
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, bpx = gridDim.x >> 3;
    const int tid = threadIdx.x;
    int b, part, parts;
    for (int wi = 0; xcd_work(wi, xcd, img0, nimg, b, part, parts); wi++) {
        const PostDesc im = imgs[b];
        const int K = im.K, Kp = im.Kp, K4 = Kp >> 2, ldt = Kp + 1;
        f32x4* Q4 = reinterpret_cast<f32x4*>(Q + im.qoff);
        const int lo_g = pairwise ? Lg.idbase[b] : 0, lo_b = pairwise ? Lb.idbase[b] : 0;
        const int TPe = CRF_TP / im.G < tp_cap ? CRF_TP / im.G : tp_cap; // one softmax thread per (pixel, group)
        const int TW = tile_w > 0 ? tile_w : crf_tile_w(TPe), TH = TPe / TW, TP = TW * TH;   // TW: a power of two
        const int tw_shift = __ffs(TW) - 1;
        const int tiles_x = (im.W + TW - 1) / TW, tiles_y = (im.H + TH - 1) / TH, ntiles = tiles_x * tiles_y;
        const int tfirst = (int)((long)ntiles * part / parts), tend = (int)((long)ntiles * (part + 1) / parts);
        // rows of the pixel-major arrays and of the lattice value arrays as base + 32-bit byte offset
        const char* const Ub = reinterpret_cast<const char*>(unary + im.qoff);
        const char* const Gb = reinterpret_cast<const char*>(vg + im.voff[0]);
        const char* const Bb = reinterpret_cast<const char*>(vb + im.voff[1]);
        const uint32_t rowb = (uint32_t)K4 * 16u;
        int* const rec_ob = reinterpret_cast<int*>(tile + TP * ldt);                   // [TP][6] bilateral ids, then ...
        float* const rec_wb = reinterpret_cast<float*>(rec_ob + TP * 6);
        int* const rec_og = reinterpret_cast<int*>(rec_wb + TP * 6);
        float* const rec_wg = reinterpret_cast<float*>(rec_og + TP * 3);
        float* const rec_nb = rec_wg + TP * 3;
        float* const rec_ng = rec_nb + TP;
        const int q256 = 256 / K4, r256 = 256 - q256 * K4;             // item -> (pixel, chunk) advance without a division
        const int pl0 = tid / K4, c0 = tid - pl0 * K4;
        for (int tl = tfirst + slot; tl < tend; tl += bpx) {
            const int ty = tl / tiles_x, tx = tl - ty * tiles_x;
            const int x0 = tx * TW, y0 = ty * TH;
            // slots of an edge tile that fall outside the image redo its last column / row (branch-free main loop) and
            // store nothing
            auto pixel_of = [&](int pl) {
                const int x = x0 + (pl & (TW - 1)), y = y0 + (pl >> tw_shift);
                return (y < im.H ? y : im.H - 1) * im.W + (x < im.W ? x : im.W - 1);
            };
            auto inside = [&](int pl) { return x0 + (pl & (TW - 1)) < im.W && y0 + (pl >> tw_shift) < im.H; };
            // the tile's per-pixel slice records -- 6 + 3 image-local lattice ids and barycentric weights, two normalisers --
            // are staged once through LDS (coalesced loads) instead of being fetched by every channel-chunk lane of the
            // pixel: 10 vector-memory instructions per (pixel, chunk) item instead of 30 (the kernel is issue-bound there)
            if (pairwise) {
                for (int i = tid; i < TP * 6; i += 256) {
                    const int pl = i / 6;
                    const size_t g = ((size_t)im.pix0 + pixel_of(pl)) * 6 + (i - pl * 6);
                    rec_ob[i] = Lb.offset[g] - lo_b;
                    rec_wb[i] = Lb.bary[g];
                }
                for (int i = tid; i < TP * 3; i += 256) {
                    const int pl = i / 3;
                    const size_t g = ((size_t)im.pix0 + pixel_of(pl)) * 3 + (i - pl * 3);
                    rec_og[i] = Lg.offset[g] - lo_g;
                    rec_wg[i] = Lg.bary[g];
                }
                for (int pl = tid; pl < TP; pl += 256) {
                    const size_t g = (size_t)im.pix0 + pixel_of(pl);
                    rec_nb[pl] = norm_b[g];
                    rec_ng[pl] = norm_g[g];
                }
                __syncthreads();
            }
            {
                int pl = pl0, c = c0;
                for (int item = tid; item < TP * K4; item += 256) {
                    const int px = pixel_of(pl);
                    {
                        const uint32_t cofs = (uint32_t)c * 16u;
                        const f32x4 u = ld_stream(reinterpret_cast<const f32x4*>(Ub + (__umul24((uint32_t)px, rowb) + cofs)));
                        f32x4 t = {-u[0], -u[1], -u[2], -u[3]};
                        if (pairwise) {
                            f32x4 vg3[3], vb6[6];
#pragma unroll
                            for (int v = 0; v < 3; v++)
                                vg3[v] = *reinterpret_cast<const f32x4*>(Gb + (__umul24((uint32_t)rec_og[pl * 3 + v], rowb) + cofs));
#pragma unroll
                            for (int v = 0; v < 6; v++)
                                vb6[v] = *reinterpret_cast<const f32x4*>(Bb + (__umul24((uint32_t)rec_ob[pl * 6 + v], rowb) + cofs));
                            {
                                f32x4 out = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                                for (int v = 0; v < 3; v++) {
                                    const float wv = rec_wg[pl * 3 + v];
#pragma unroll
                                    for (int i = 0; i < 4; i++) out[i] = __fadd_rn(out[i], __fmul_rn(__fmul_rn(wv, vg3[v][i]), alpha_g));
                                }
                                const float nr = rec_ng[pl];
#pragma unroll
                                for (int i = 0; i < 4; i++) t[i] = __fsub_rn(t[i], __fmul_rn(-w_g, __fmul_rn(out[i], nr)));
                            }
                            {
                                f32x4 out = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                                for (int v = 0; v < 6; v++) {
                                    const float wv = rec_wb[pl * 6 + v];
#pragma unroll
                                    for (int i = 0; i < 4; i++) out[i] = __fadd_rn(out[i], __fmul_rn(__fmul_rn(wv, vb6[v][i]), alpha_b));
                                }
                                const float nr = rec_nb[pl];
#pragma unroll
                                for (int i = 0; i < 4; i++) t[i] = __fsub_rn(t[i], __fmul_rn(-w_b, __fmul_rn(out[i], nr)));
                            }
                        }
#pragma unroll
                        for (int i = 0; i < 4; i++) tile[pl * ldt + 4 * c + i] = t[i];
                    }
                    pl += q256;
                    c += r256;
                    if (c >= K4) {
                        c -= K4;
                        pl++;
                    }
                }
            }
            __syncthreads();
            if (WIDE && K >= CRF_WAVE_SOFTMAX_K) {                       // (per image: the rows of a batch differ in width)
                // Wide rows (round 6): the softmax of a tile's rows by WAVE, not by thread.  With one thread per (pixel, group) a
                // 32-pixel tile of 2 x 150 channels keeps ONE wave busy for ~6000 dependent instructions per lane (150 exponentials
                // in a row) while the other three idle.  Here each wave owns RW = TP * G / 4 rows and works through them without any
                // workgroup barrier: maxima and exponentials with the lanes over the CHANNELS of one row at a time (the maximum is
                // order-independent, pnp_expf is a pure function), the sums -- the one order-dependent step, k = 0 .. K - 1 as
                // densecrf's expAndNormalize adds them -- with one LANE PER ROW, sequentially, all rows of the wave at once; then the
                // divisions with the lanes over the channels again.  Element by element the arithmetic of the thread-per-row form:
                // bit-identical marginals and labels.  (Narrower rows keep that form: measured slower there, see CRF_WAVE_SOFTMAX_K.)  Cross-lane hand-over through LDS inside one wave needs no barrier: a wave's LDS
                // operations execute in order.
                const int lane = tid & 63, wave = tid >> 6;
                const int R = TP * im.G, RW = R >> 2;                   // rows (pixel, group) of the tile / per wave: 4 .. 64
                const int rbase = wave * RW;
                auto row_ptr = [&](int rr) {                            // row rr of the tile: group-major like the thread form
                    const int grp = rr / TP, px = rr - grp * TP;
                    return tile + px * ldt + grp * im.Kg;
                };
                float s_mine = 0.f;                                     // lane j < RW: sum of row rbase + j
                // pass 1: per row, maximum (NaN if any NaN) and exponentials, lanes over channels; two rows at a time (RW is even), so
                // that one row's shuffle chain runs under the other's loads and exponentials
                for (int j = 0; j < RW; j += 2) {
                    float* const row0 = row_ptr(rbase + j);
                    float* const row1 = row_ptr(rbase + j + 1);
                    float m0 = -INFINITY, m1 = -INFINITY;
                    bool nan0 = false, nan1 = false;
                    for (int k = lane; k < K; k += 64) {
                        const float v0 = row0[k], v1 = row1[k];
                        nan0 |= v0 != v0;
                        nan1 |= v1 != v1;
                        m0 = v0 > m0 ? v0 : m0;
                        m1 = v1 > m1 ? v1 : m1;
                    }
#pragma unroll
                    for (int o = 32; o >= 1; o >>= 1) {
                        const float a0 = __shfl_xor(m0, o, 64), a1 = __shfl_xor(m1, o, 64);
                        m0 = a0 > m0 ? a0 : m0;
                        m1 = a1 > m1 ? a1 : m1;
                    }
                    if (__any(nan0)) m0 = __builtin_nanf("");
                    if (__any(nan1)) m1 = __builtin_nanf("");
                    for (int k = lane; k < K; k += 64) {
                        const float e0 = pnp_expf(__fsub_rn(row0[k], m0)), e1 = pnp_expf(__fsub_rn(row1[k], m1));
                        row0[k] = e0;
                        row1[k] = e1;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                // pass 2: sums in channel order, one lane per row
                if (lane < RW) {
                    const float* row = row_ptr(rbase + lane);
                    float sm = 0.f;
                    for (int k = 0; k < K; k++) sm = __fadd_rn(sm, row[k]);
                    s_mine = sm;
                }
                // pass 3: divisions, lanes over channels
                for (int j = 0; j < RW; j++) {
                    float* row = row_ptr(rbase + j);
                    const float sm = __shfl(s_mine, j, 64);
                    for (int k = lane; k < K; k += 64) row[k] = __fdiv_rn(row[k], sm);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                if (lane < RW) {
                    const int rr = rbase + lane, grp = rr / TP, px = rr - grp * TP;
                    float* row = tile + px * ldt + grp * im.Kg;
                    if (lout.lab[grp]) {                                // last iteration: first maximum, NaN counts as maximum
                        int best = 0;
                        float bv = row[0];
                        for (int k = 1; k < K; k++) {
                            const float v = row[k];
                            if (bv == bv && (v > bv || v != v)) {
                                best = k;
                                bv = v;
                            }
                        }
                        if (inside(px)) lout.lab[grp][lout.label_off[b] + pixel_of(px)] = (uint8_t)lout.lut[b * lout.lut_stride + best];
                    }
                    if (grp == im.G - 1)                                // pad floats behind the last group stay zero
                        for (int k = im.G * im.Kg; k < Kp; k++) tile[px * ldt + k] = 0.f;
                }
            } else if (tid < TP * im.G) {                               // one softmax per (pixel, channel group)
                const int grp = tid / TP, px = tid - grp * TP;
                const int p = pixel_of(px);
                {
                    float* row = tile + px * ldt + grp * im.Kg;
                    float m = row[0];
                    for (int k = 1; k < K; k++) {
                        const float v = row[k];
                        if (v > m || v != v) m = v;
                    }
                    float s = 0.f;
                    for (int k = 0; k < K; k++) {
                        const float e = pnp_expf(__fsub_rn(row[k], m));
                        row[k] = e;
                        s = __fadd_rn(s, e);
                    }
                    for (int k = 0; k < K; k++) row[k] = __fdiv_rn(row[k], s);
                    if (lout.lab[grp]) {
                        // last iteration: the label of this (pixel, group) straight from the marginals in LDS -- first maximum,
                        // NaN counts as maximum (np.argmax, as argmax_kernel reads them back from memory) -- through the LUT
                        int best = 0;
                        float bv = row[0];
                        for (int k = 1; k < K; k++) {
                            const float v = row[k];
                            if (bv == bv && (v > bv || v != v)) {
                                best = k;
                                bv = v;
                            }
                        }
                        if (inside(px)) lout.lab[grp][lout.label_off[b] + p] = (uint8_t)lout.lut[b * lout.lut_stride + best];
                    }
                    if (grp == im.G - 1)                                // pad floats behind the last group stay zero
                        for (int k = im.G * im.Kg; k < Kp; k++) tile[px * ldt + k] = 0.f;
                }
            }
            __syncthreads();
            for (int item = tid; item < TP * K4; item += 256) {
                const int pl = item / K4, c = item - pl * K4;
                if (inside(pl)) {
                    const float* r = tile + pl * ldt + 4 * c;
                    st_stream(Q4 + (size_t)pixel_of(pl) * K4 + c, f32x4{r[0], r[1], r[2], r[3]});
                }
            }
            __syncthreads();
        }
    }
}

// ---- spatial renumbering.  The key sort numbers lattice points in (colour-major) key order; the
// iteration kernels are gather-bound, so the points of each image are renumbered by the position of their
// FIRST contributor pixel along a Z-order curve over the image: contributors, simplex vertices of neighbouring
// pixels and blur neighbours (all within about one spatial cell, sxy pixels, in both directions) then sit in
// nearby value rows and the gathers hit L2.  Against row-major pixel order: the same at 0.9 lattice points per pixel,
// 7 % less mean-field time at 3.6.  Lattice ids are internal: results do not depend on them.
__device__ __forceinline__ uint32_t morton_spread(uint32_t v) {          // 16 bits -> every second bit
    v = (v | (v << 8)) & 0x00FF00FFu;
    v = (v | (v << 4)) & 0x0F0F0F0Fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}
// sort key of a lattice point: (image | position of its first contributor pixel along a Z-order curve over the image | vertex)
__global__ void first_contrib_kernel(const uint32_t* __restrict__ vals, const int* __restrict__ seg_start, int M,
                                     const PostDesc* __restrict__ imgs, const int* __restrict__ idbase, int B, int D1,
                                     uint64_t* __restrict__ fkey, uint32_t* __restrict__ fid) {
    for (int id = blockIdx.x * blockDim.x + threadIdx.x; id < M; id += gridDim.x * blockDim.x) {
        const uint32_t pv = vals[seg_start[id]];
        int lo = 0, hi = B;                                  // image of the point: idbase[lo] <= id < idbase[lo + 1]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (idbase[mid] <= id) lo = mid;
            else hi = mid;
        }
        const uint32_t pixel = pv / (uint32_t)D1, v = pv - pixel * (uint32_t)D1;
        const uint32_t local = pixel - (uint32_t)imgs[lo].pix0;
        const uint32_t W = (uint32_t)imgs[lo].W, y = local / W, x = local - y * W;
        const uint64_t pos = (uint64_t)morton_spread(x) | ((uint64_t)morton_spread(y) << 1);
        fkey[id] = ((uint64_t)lo << 40) | (pos << 3) | v;
        fid[id] = (uint32_t)id;
    }
}
__global__ void rank_kernel(const uint32_t* __restrict__ sorted_id, int M, int* __restrict__ rank) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < M; i += gridDim.x * blockDim.x) rank[sorted_id[i]] = i;
}
__global__ void renumber_points_kernel(const int* __restrict__ rank, const int* __restrict__ seg_start,
                                       const int* __restrict__ n1k, const int* __restrict__ n2k, size_t cap, int M, int D1,
                                       int* __restrict__ seg_lo, int* __restrict__ seg_hi, int* __restrict__ n1,
                                       int* __restrict__ n2) {
    for (int id = blockIdx.x * blockDim.x + threadIdx.x; id < M; id += gridDim.x * blockDim.x) {
        const int nid = rank[id];
        seg_lo[nid] = seg_start[id];
        seg_hi[nid] = seg_start[id + 1];
        for (int j = 0; j < D1; j++) {
            const int a = n1k[(size_t)j * cap + id], b = n2k[(size_t)j * cap + id];
            n1[(size_t)j * cap + nid] = a >= 0 ? rank[a] : -1;
            n2[(size_t)j * cap + nid] = b >= 0 ? rank[b] : -1;
        }
    }
}
__global__ void renumber_entries_kernel(const int* __restrict__ rank, size_t n, int* __restrict__ offset) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        offset[i] = rank[offset[i]];
}

// ---- neighbour records of the two-axis blur: for axis pair (2p, 2p+1) the point's neighbours along axis 2p (ia, id), along
// axis 2p+1 (pa, pd) and the axis-2p neighbours of those two (aa, ad, da, dd), as image-local ids (-1 = absent)
__global__ void nbr8_kernel(const int* __restrict__ n1, const int* __restrict__ n2, size_t cap, const int* __restrict__ idbase,
                            int npairs, CrfNbr8* __restrict__ out) {
    const int b = blockIdx.y;
    const int lo = idbase[b], hi = idbase[b + 1];
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < (hi - lo) * npairs; t += gridDim.x * blockDim.x) {
        const int id = lo + t / npairs, pr = t % npairs;
        const int* n1a = n1 + (size_t)(2 * pr) * cap;
        const int* n2a = n2 + (size_t)(2 * pr) * cap;
        const int* n1b = n1 + (size_t)(2 * pr + 1) * cap;
        const int* n2b = n2 + (size_t)(2 * pr + 1) * cap;
        const int pa = n1b[id], pd = n2b[id];
        CrfNbr8 r;
        r.ia = n1a[id];
        r.id = n2a[id];
        r.pa = pa;
        r.pd = pd;
        r.aa = pa >= 0 ? n1a[pa] : -1;
        r.ad = pa >= 0 ? n2a[pa] : -1;
        r.da = pd >= 0 ? n1a[pd] : -1;
        r.dd = pd >= 0 ? n2a[pd] : -1;
        int* q = &r.ia;
#pragma unroll
        for (int k = 0; k < 8; k++) q[k] = q[k] >= 0 ? q[k] - lo : -1;
        out[(size_t)pr * cap + id] = r;
    }
}

// ---- contributor lists in lattice-id order.  The key sort leaves each point's contributor segment where its KEY
// sorted; after the spatial renumbering consecutive lattice ids own segments scattered over the whole list.  The
// segments are moved (each keeps its ascending-pixel order) so that id order = list order: the splat then walks the
// list front to back, and a range of lattice points is a contiguous range of contributors.
__global__ void seg_len_kernel(const int* __restrict__ seg_lo, const int* __restrict__ seg_hi, int M, int* __restrict__ len) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < M; i += gridDim.x * blockDim.x) len[i] = seg_hi[i] - seg_lo[i];
}
__global__ void move_segments_kernel(const uint32_t* __restrict__ vals, const int* __restrict__ new_start, int M, int D1,
                                     const float* __restrict__ bary, int* __restrict__ seg_lo, int* __restrict__ seg_hi,
                                     uint32_t* __restrict__ vals2, CrfEntry* __restrict__ ent) {
    // eight lanes per lattice point: the entries of a segment are copied side by side (one thread per point walked its
    // segment alone: 16-byte records at scattered addresses, 0.49 ms per build)
    const long nthr = (long)gridDim.x * blockDim.x;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < (long)M * 8; t += nthr) {
        const int i = (int)(t >> 3), j0 = (int)(t & 7);
        const int a = seg_lo[i], n = seg_hi[i] - a, d = new_start[i];
        for (int j = j0; j < n; j += 8) {
            const uint32_t pv = vals[a + j];
            vals2[d + j] = pv;
            ent[d + j] = CrfEntry{pv / (uint32_t)D1, bary[pv], 0.f, 0u};  // the splat's contributor record (nr: crf_lattice_norm)
        }
    }
}
// (the segment bounds are rewritten once every lane group has read the old ones: separate launch)
__global__ void set_segments_kernel(const int* __restrict__ new_start, int M, int* __restrict__ seg_lo, int* __restrict__ seg_hi) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < M; i += gridDim.x * blockDim.x) {
        const int n = seg_hi[i] - seg_lo[i], d = new_start[i];
        seg_lo[i] = d;
        seg_hi[i] = d + n;
    }
}

// ------------------------------------------------------------------------------------------ host
static inline int ok() { return hipGetLastError() == hipSuccess ? PNP_OK : PNP_ERR_HIP; }

size_t crf_sort_temp_bytes(size_t max_entries, int max_images) {
    (void)max_images;
    return sort_temp_bytes(max_entries) + 256;
}

// Build one lattice (D = 2: Gaussian xy/sxy ; D = 5: bilateral xy/sxy, rgb/srgb) for images [0,B).
// Scratch arrays (keys/vals double buffers, head, incl, temp) are caller-provided.
int crf_build_lattice(int D, const CrfLattice& L, const PostDesc* d_imgs, const PostDesc* h_imgs, const uint8_t* d_rgb, float sxy, float srgb,
                      int B, size_t ent_total, int max_pixels,
                      uint64_t* keys_a, uint64_t* keys_b, uint32_t* vals_a, int* head, int* incl, int* n1k, int* n2k,
                      void* temp, size_t temp_bytes, int* d_range_err, int* h_range_err, int* h_points, hipStream_t s) {
    const int nb = (max_pixels + 255) / 256 < 512 ? (max_pixels + 255) / 256 : 512;
    if (D == 2)
        hipLaunchKernelGGL((lattice_embed_kernel<2>), dim3(nb, B), dim3(256), 0, s, d_imgs, d_rgb, sxy, srgb, L.bary, keys_a, vals_a, d_range_err);
    else if (D == 5)
        hipLaunchKernelGGL((lattice_embed_kernel<5>), dim3(nb, B), dim3(256), 0, s, d_imgs, d_rgb, sxy, srgb, L.bary, keys_a, vals_a, d_range_err);
    else
        return PNP_ERR_ARG;
    if (B > (1 << IMG_BITS)) return PNP_ERR_ARG;
    // bits actually populated: coordinates (D * BITS, low) + image index (IMG_SHIFT..)
    // stable sort by (image, coordinates).  The entries of image b are the run [pix0 * (D+1), (pix0 + H W) * (D+1)) on entry, so
    // the sort is segmented by image over the coordinate bits [0, D * BITS) alone (sort.hip; the image bits at IMG_SHIFT stay in
    // the keys for the kernels below; the inputs keys_a / vals_a are scratch from here on)
    const int coord_bits = D == 2 ? 2 * KeyPack<2>::BITS : 5 * KeyPack<5>::BITS;
    size_t seg_off[(1 << IMG_BITS) + 1];
    for (int b = 0; b < B; b++) {
        seg_off[b] = (size_t)h_imgs[b].pix0 * (D + 1);
        if (b && seg_off[b] != seg_off[b - 1] + (size_t)h_imgs[b - 1].H * h_imgs[b - 1].W * (D + 1)) return PNP_ERR_ARG;
    }
    seg_off[B] = ent_total;
    if (B < 1 || seg_off[0] != 0 || seg_off[B - 1] + (size_t)h_imgs[B - 1].H * h_imgs[B - 1].W * (D + 1) != ent_total) return PNP_ERR_ARG;
    {
        const int r = radix_sort_pairs(keys_a, keys_b, vals_a, L.vals, ent_total, 0, coord_bits, seg_off, B, temp, temp_bytes, s);
        if (r != PNP_OK) return r;
    }
    const int nbe = 1024;
    hipLaunchKernelGGL(mark_heads_kernel, dim3(nbe, B), dim3(256), 0, s, keys_b, d_imgs, D + 1, head);
    if (const int r = device_scan_i32(head, incl, ent_total, true, temp, temp_bytes, s); r != PNP_OK) return r;
    hipLaunchKernelGGL(scatter_ids_kernel, dim3(nbe, B), dim3(256), 0, s, keys_b, L.vals, head, incl, d_imgs, D + 1, B,
                       ent_total, L.offset, L.seg_start, L.ukeys, L.idbase);
    if (D == 2)
        hipLaunchKernelGGL((neighbors_kernel<2>), dim3(nbe, B), dim3(256), 0, s, L.ukeys, L.idbase, L.cap, n1k, n2k);
    else
        hipLaunchKernelGGL((neighbors_kernel<5>), dim3(nbe, B), dim3(256), 0, s, L.ukeys, L.idbase, L.cap, n1k, n2k);
    // spatial renumbering (needs the lattice size on the host: one small read-back per build)
    int M = 0;
    int h_idbase[(1 << IMG_BITS) + 1];                    // first lattice id of every image (the second sort's segments)
    // (the key-range flag of the embed kernel rides on the same synchronisation)
    int err = 0;
    if (hipMemcpyAsync(h_idbase, L.idbase, sizeof(int) * (B + 1), hipMemcpyDeviceToHost, s) != hipSuccess) return PNP_ERR_HIP;
    if (hipMemcpyAsync(&err, d_range_err, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess) return PNP_ERR_HIP;
    if (hipStreamSynchronize(s) != hipSuccess) return PNP_ERR_HIP;
    M = h_idbase[B];
    if (err && h_range_err) *h_range_err = 1;
    if (M <= 0 || (size_t)M > ent_total) return PNP_ERR_STATE;
    if (h_points) *h_points = M;                          // lattice points of the batch (bench.py: lattice term of the byte model)
    uint64_t* fkey = keys_a;
    uint64_t* skey = keys_b;
    uint32_t* fid = vals_a;                               // (the first sort's input: free by now)
    uint32_t* sid = reinterpret_cast<uint32_t*>(incl);    // (rewritten by the segment scan below)
    int* rank = head;
    hipLaunchKernelGGL(first_contrib_kernel, dim3(1024), dim3(256), 0, s, L.vals, L.seg_start, M, d_imgs, L.idbase, B, D + 1, fkey, fid);
    {   // lattice points are numbered image by image: segmented by image over (Z-order position, vertex) = bits [0, 40)
        size_t seg2[(1 << IMG_BITS) + 1];
        for (int b = 0; b <= B; b++) seg2[b] = (size_t)h_idbase[b];
        if (const int r = radix_sort_pairs(fkey, skey, fid, sid, (size_t)M, 0, 40, seg2, B, temp, temp_bytes, s); r != PNP_OK) return r;
    }
    hipLaunchKernelGGL(rank_kernel, dim3(1024), dim3(256), 0, s, sid, M, rank);
    hipLaunchKernelGGL(renumber_points_kernel, dim3(1024), dim3(256), 0, s, rank, L.seg_start, n1k, n2k, L.cap, M, D + 1,
                       L.seg_lo, L.seg_hi, L.n1, L.n2);
    hipLaunchKernelGGL(renumber_entries_kernel, dim3(2048), dim3(256), 0, s, rank, ent_total, L.offset);
    hipLaunchKernelGGL(nbr8_kernel, dim3(256, B), dim3(256), 0, s, L.n1, L.n2, L.cap, L.idbase, (D + 1) / 2, L.nbr8);
    // contributor segments into lattice-id order (scratch: keys_a = lengths, incl = new starts, keys_b = moved list)
    int* len = reinterpret_cast<int*>(keys_a);
    uint32_t* vals2 = reinterpret_cast<uint32_t*>(keys_b);
    hipLaunchKernelGGL(seg_len_kernel, dim3(1024), dim3(256), 0, s, L.seg_lo, L.seg_hi, M, len);
    if (const int r = device_scan_i32(len, incl, (size_t)M, false, temp, temp_bytes, s); r != PNP_OK) return r;
    hipLaunchKernelGGL(move_segments_kernel, dim3(4096), dim3(256), 0, s, L.vals, incl, M, D + 1, L.bary, L.seg_lo, L.seg_hi, vals2, L.ent);
    hipLaunchKernelGGL(set_segments_kernel, dim3(1024), dim3(256), 0, s, incl, M, L.seg_lo, L.seg_hi);
    if (hipMemcpyAsync(L.vals, vals2, ent_total * sizeof(uint32_t), hipMemcpyDeviceToDevice, s) != hipSuccess) return PNP_ERR_HIP;
    return ok();
}

static float crf_alpha(int D) { return 1.0f / (1 + powf(2, (float)-D)); }

// Grid of the XCD-affine iteration kernels: 8 XCD groups x the workgroups that are RESIDENT per XCD for this kernel
// (occupancy query x CUs per XCD).  Each workgroup walks its share of every image of its XCD, so a grid larger than what is
// resident runs as a second wave of workgroups after the first has swept all its images: unbalanced, and the second wave
// re-fetches from HBM every Q / value row the first one had in L2.
template <typename F>
static int resident_grid(F kernel, int threads, size_t smem) {
    int cus = device_cu_count(), occ = 0;
    if (cus <= 0) cus = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, threads, smem) != hipSuccess || occ < 1) occ = 1;
    const int per_xcd = (cus / 8 > 0 ? cus / 8 : 1) * occ;
    return 8 * (per_xcd < 256 ? per_xcd : 256);
}

// norm = 1 / sqrt(lattice(ones) + 1e-20) for images [0,B); va/vb: scratch of >= (number of lattice points) floats.
int crf_lattice_norm(const CrfLattice& L, const PostDesc* d_imgs, int B, int max_pixels, size_t ent_total, float* va, float* vb,
                     float* norm_out, hipStream_t s) {
    const int D = L.D1 - 1;
    const int nb = 1024;
    hipLaunchKernelGGL(crf_splat1_kernel, dim3(nb, B), dim3(256), 0, s, L, d_imgs, va);
    float* src = va;
    float* dst = vb;
    for (int j = 0; j <= D; j++) {
        hipLaunchKernelGGL(crf_blur1_kernel, dim3(nb, B), dim3(256), 0, s, L, d_imgs, src, dst, j);
        float* t = src;
        src = dst;
        dst = t;
    }
    const int nbp = (max_pixels + 255) / 256 < 512 ? (max_pixels + 255) / 256 : 512;
    hipLaunchKernelGGL(crf_norm_kernel, dim3(nbp, B), dim3(256), 0, s, L, d_imgs, src, norm_out, crf_alpha(D));
    hipLaunchKernelGGL(entry_norm_kernel, dim3(2048), dim3(256), 0, s, L.ent, ent_total, norm_out);
    return ok();
}

// lattice(norm * Q) for images [img0, img0+nimg): splat + (d+1) blurs (the normaliser rides in the contributor records).
// Returns the buffer holding the result.
int crf_filter(const CrfLattice& L, const PostDesc* d_imgs, int img0, int nimg, const float* Q,
               float* va, float* vb, const float** result, int max_kp, hipStream_t s) {
    const int D = L.D1 - 1;
    const int k4 = (max_kp + 3) / 4;
#define PNP_SPLAT(LPP_, MULTI_)                                                                                                  \
    do {                                                                                                                      \
        static PerDeviceInt g0_, g1_;                        /* per device ordinal (common.h) */                              \
        const int nbs0 = g0_.get([] { return resident_grid(crf_splat_kernel<LPP_, 0, MULTI_>, 256, 0); });                    \
        const int nbs1 = g1_.get([] { return resident_grid(crf_splat_kernel<LPP_, 1, MULTI_>, 256, 0); });                    \
        if (L.which == 0) hipLaunchKernelGGL((crf_splat_kernel<LPP_, 0, MULTI_>), dim3(nbs0), dim3(256), 0, s, L, d_imgs, Q, va, img0, nimg); \
        else hipLaunchKernelGGL((crf_splat_kernel<LPP_, 1, MULTI_>), dim3(nbs1), dim3(256), 0, s, L, d_imgs, Q, va, img0, nimg);              \
    } while (0)
    if (k4 <= 8) PNP_SPLAT(8, false);
    else if (k4 <= 16) PNP_SPLAT(16, false);
    else if (k4 <= 32) PNP_SPLAT(32, false);
    else PNP_SPLAT(32, true);
#undef PNP_SPLAT
    static PerDeviceInt gb2, gb1;                            // per device ordinal (common.h)
    const int nb2 = gb2.get([] { return resident_grid(crf_blur4x2_kernel, 256, 0); });
    const int nb1 = gb1.get([] { return resident_grid(crf_blur4_kernel, 256, 0); });
    float* src = va;
    float* dst = vb;
    // axes in pairs through the fused two-axis kernel (bilateral: 3 passes for 6 axes; Gaussian: one pair + one single):
    // mean-field 41.6 -> 37.0 ms per bench step, results bit-identical
    for (int j = 0; j <= D;) {
        if (j + 1 <= D) {
            hipLaunchKernelGGL(crf_blur4x2_kernel, dim3(nb2), dim3(256), 0, s, L, d_imgs, src, dst, j / 2, img0, nimg);
            j += 2;
        } else {
            hipLaunchKernelGGL(crf_blur4_kernel, dim3(nb1), dim3(256), 0, s, L, d_imgs, src, dst, j, img0, nimg);
            j += 1;
        }
        float* t = src;
        src = dst;
        dst = t;
    }
    *result = src;
    return ok();
}

// Q <- softmax(-U - pairwise terms) (pairwise != 0) or softmax(-U) (pairwise == 0)
int crf_update(const CrfLattice& Lg, const CrfLattice& Lb, const PostDesc* d_imgs, int img0, int nimg, const float* vg,
               const float* vb, const float* norm_g, const float* norm_b, const float* unary, float* Q, float w_g,
               float w_b, int pairwise, int max_pixels, int max_kp, int groups, hipStream_t s, const CrfLabelOut& labels) {
    // tile + per-pixel slice records; wide rows (150 classes) take fewer pixels per tile to stay inside the CU's 160 KB
    size_t tp = (size_t)(CRF_TP / (groups > 0 ? groups : 1));
    const size_t per_pixel = ((size_t)max_kp + 1 + 20) * sizeof(float);
    if (groups == 2 && tp > 64) tp = 64;                     // smaller tiles, more resident workgroups: 643 -> 624 us on the paired bench rows
    // wide rows: keep a tile under ~48 KB so that three workgroups stay resident per CU (the kernel waits on its row gathers;
    // measured per bench step: COCO-Object K = 81 70 -> 58 ms at 64 pixels, ADE20K K = 2 x 150 348 -> 278 ms at 32 pixels;
    // below 32 pixels the per-tile overheads win: 295 ms at 16, 389 ms at 8)
    while (tp > 16 && tp * per_pixel > 48 * 1024) tp >>= 1;
    if (tp * per_pixel > 158 * 1024) tp = 158 * 1024 / per_pixel;
    int tile_w = 0;                                          // 0: crf_tile_w(pixels per tile)
#ifdef PNP_DEV
    if (getenv("PNP_CRF_TP") && (size_t)atoi(getenv("PNP_CRF_TP")) < tp) tp = (size_t)atoi(getenv("PNP_CRF_TP"));
    if (getenv("PNP_CRF_TW")) tile_w = atoi(getenv("PNP_CRF_TW"));   // a power of two <= tp (tp: the whole strip, as before)
#endif
    if (tp < 4) return PNP_ERR_ARG;
    const size_t smem = tp * per_pixel;
    const bool wide = max_kp / (groups > 0 ? groups : 1) >= CRF_WAVE_SOFTMAX_K;      // some image of the batch may have such rows
    const void* const kern = wide ? reinterpret_cast<const void*>(crf_update_kernel<true>) : reinterpret_cast<const void*>(crf_update_kernel<false>);
    if (smem > 64 * 1024) {
        if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return PNP_ERR_HIP;
    }
    (void)max_pixels;
    // (the occupancy depends on the tile's LDS bytes: queried per distinct size, a handful per process)
    static size_t grid_smem[16];
    static int grid_dev[16], grid_nb[16], grid_wide[16], grid_n = 0;
    static std::mutex grid_mu;                               // engines of different host threads / devices share the table
    int nbu = 0;
    {
        const int dev = current_device();
        std::lock_guard<std::mutex> lk(grid_mu);
        for (int i = 0; i < grid_n; i++)
            if (grid_smem[i] == smem && grid_dev[i] == dev && grid_wide[i] == (int)wide) nbu = grid_nb[i];
        if (!nbu) {
            nbu = wide ? resident_grid(crf_update_kernel<true>, 256, smem) : resident_grid(crf_update_kernel<false>, 256, smem);
            if (grid_n < 16) {
                grid_smem[grid_n] = smem;
                grid_dev[grid_n] = dev;
                grid_wide[grid_n] = (int)wide;
                grid_nb[grid_n++] = nbu;
            }
        }
    }
    if (wide)
        hipLaunchKernelGGL(crf_update_kernel<true>, dim3(nbu), dim3(256), smem, s, Lg, Lb, d_imgs, vg, vb, norm_g, norm_b, unary, Q,
                           w_g, w_b, crf_alpha(2), crf_alpha(5), pairwise, img0, nimg, (int)tp, tile_w, labels);
    else
        hipLaunchKernelGGL(crf_update_kernel<false>, dim3(nbu), dim3(256), smem, s, Lg, Lb, d_imgs, vg, vb, norm_g, norm_b, unary, Q,
                           w_g, w_b, crf_alpha(2), crf_alpha(5), pairwise, img0, nimg, (int)tp, tile_w, labels);
    return ok();
}

}  // namespace pnp
