// Shared device/host helpers for the PnP-OVSS gfx950 engine.  CDNA4 only: 64-wide wavefronts,
// MFMA 16x16 tiles, 160 KB LDS.  No portability layer on purpose.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>

namespace pnp {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;   // one 32x32 MFMA accumulator tile

typedef __bf16 bf16;

// ---- element traits: T is the storage/compute-input type of the dense contractions
template <typename T> struct Elem;
template <> struct Elem<bf16> {
    static constexpr int kBytes = 2;
    static constexpr int kPerChunk = 8;   // elements per 16-byte chunk
};
template <> struct Elem<float> {
    static constexpr int kBytes = 4;
    static constexpr int kPerChunk = 4;
};

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float v) { return (bf16)v; }   // v_cvt_pk_bf16_f32 (RNE, NaN-safe)

// ---- fragment: the 8 k-elements one lane feeds to a 16x16 MFMA "k-step" (32 k-values per step).
// bf16: one v_mfma_f32_16x16x32_bf16, lane (r=l&15, q=l>>4) element j <-> k = 8q + j.
// f32 : eight v_mfma_f32_16x16x4_f32 (exact f32 fma chain), element j of quad q <-> any k as long
//       as A and B fragments use the same (q, j) -> k map (they are loaded by the same code).
template <typename T> struct Frag;
template <> struct Frag<bf16> { bf16x8 v; };
template <> struct Frag<float> { float v[8]; };

__device__ __forceinline__ void mma16(f32x4& acc, const Frag<bf16>& a, const Frag<bf16>& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma16(f32x4& acc, const Frag<float>& a, const Frag<float>& b) {
#pragma unroll
    for (int j = 0; j < 8; j++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j], b.v[j], acc, 0, 0, 0);
}

// 16-byte vector for global/LDS moves
typedef __attribute__((ext_vector_type(4))) uint32_t chunk16;
typedef __attribute__((ext_vector_type(2))) uint32_t chunk8;

// ---- LDS XOR swizzle over 16-byte chunks so that the 16 lanes of an MFMA operand read (16
// consecutive rows, same logical chunk) hit 16 distinct 16-byte slots of the 256-byte bank row.
// ROWB = bytes per LDS row (128 or 256).
template <int ROWB> __device__ __forceinline__ int swz_chunk(int row, int chunk);
template <> __device__ __forceinline__ int swz_chunk<128>(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }
template <> __device__ __forceinline__ int swz_chunk<256>(int row, int chunk) { return chunk ^ (row & 15); }

template <int ROWB> __device__ __forceinline__ int lds_off(int row, int chunk) {
    return row * ROWB + swz_chunk<ROWB>(row, chunk) * 16;
}

// load a lane's fragment from a swizzled LDS tile whose rows hold one k-block.
// For bf16 the fragment is logical chunk `c` (16 B).  For f32 it is chunks c and c+4 of a 128-B
// row (k = 4q.. and 16+4q..).
template <int ROWB>
__device__ __forceinline__ void lds_frag(Frag<bf16>& f, const char* tile, int row, int kstep, int q) {
    f.v = *reinterpret_cast<const bf16x8*>(tile + lds_off<ROWB>(row, kstep * 4 + q));
}
template <int ROWB>
__device__ __forceinline__ void lds_frag(Frag<float>& f, const char* tile, int row, int kstep, int q) {
    const f32x4 lo = *reinterpret_cast<const f32x4*>(tile + lds_off<ROWB>(row, kstep * 8 + q));
    const f32x4 hi = *reinterpret_cast<const f32x4*>(tile + lds_off<ROWB>(row, kstep * 8 + 4 + q));
#pragma unroll
    for (int j = 0; j < 4; j++) { f.v[j] = lo[j]; f.v[4 + j] = hi[j]; }
}

// ---- fragment loads straight from global memory (same (q, j) -> k map as lds_frag)
__device__ __forceinline__ void glb_frag(Frag<bf16>& f, const bf16* row, int kstep, int q) {
    f.v = *reinterpret_cast<const bf16x8*>(row + kstep * 32 + q * 8);
}
__device__ __forceinline__ void glb_frag(Frag<float>& f, const float* row, int kstep, int q) {
    const f32x4 lo = *reinterpret_cast<const f32x4*>(row + (kstep * 8 + q) * 4);
    const f32x4 hi = *reinterpret_cast<const f32x4*>(row + (kstep * 8 + 4 + q) * 4);
#pragma unroll
    for (int j = 0; j < 4; j++) { f.v[j] = lo[j]; f.v[4 + j] = hi[j]; }
}
// fragment whose elements 0..3 / 4..7 are 4 consecutive values at pa / pb ("accumulator order":
// exactly the keys a lane of two S^T accumulator tiles holds for its query column)
__device__ __forceinline__ void glb_frag_pair(Frag<bf16>& f, const bf16* pa, const bf16* pb) {
    const bf16x4 lo = *reinterpret_cast<const bf16x4*>(pa);
    const bf16x4 hi = *reinterpret_cast<const bf16x4*>(pb);
#pragma unroll
    for (int j = 0; j < 4; j++) { f.v[j] = lo[j]; f.v[4 + j] = hi[j]; }
}
__device__ __forceinline__ void glb_frag_pair(Frag<float>& f, const float* pa, const float* pb) {
    const f32x4 lo = *reinterpret_cast<const f32x4*>(pa);
    const f32x4 hi = *reinterpret_cast<const f32x4*>(pb);
#pragma unroll
    for (int j = 0; j < 4; j++) { f.v[j] = lo[j]; f.v[4 + j] = hi[j]; }
}
// "Accumulator order" k map from a swizzled LDS tile: element j<4 <-> k = 32u + 4q + j,
// j>=4 <-> k = 32u + 16 + 4q + (j-4).
template <int ROWB>
__device__ __forceinline__ void lds_frag_acc_order(Frag<bf16>& f, const char* tile, int row, int u, int q) {
    const bf16x4 lo = *reinterpret_cast<const bf16x4*>(tile + lds_off<ROWB>(row, 4 * u + (q >> 1)) + (q & 1) * 8);
    const bf16x4 hi = *reinterpret_cast<const bf16x4*>(tile + lds_off<ROWB>(row, 4 * u + 2 + (q >> 1)) + (q & 1) * 8);
#pragma unroll
    for (int j = 0; j < 4; j++) { f.v[j] = lo[j]; f.v[4 + j] = hi[j]; }
}
template <int ROWB>
__device__ __forceinline__ void lds_frag_acc_order(Frag<float>& f, const char* tile, int row, int u, int q) {
    lds_frag<ROWB>(f, tile, row, u, q);      // chunks 8u+q and 8u+4+q: k = 32u+4q.. and 32u+16+4q..
}
__device__ __forceinline__ void pack_p(Frag<bf16>& f, const f32x4& a, const f32x4& b) {
#pragma unroll
    for (int j = 0; j < 4; j++) { f.v[j] = (bf16)a[j]; f.v[4 + j] = (bf16)b[j]; }
}
__device__ __forceinline__ void pack_p(Frag<float>& f, const f32x4& a, const f32x4& b) {
#pragma unroll
    for (int j = 0; j < 4; j++) { f.v[j] = a[j]; f.v[4 + j] = b[j]; }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7): one exp + one rcp instead of the ~40-instruction
// libm erff.  Used where the result is rounded to bf16 anyway (bf16 mode GEMM epilogues).
__device__ __forceinline__ float gelu_erf_fast(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __frcp_rn(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = 1.0f - p * t * __expf(-z * z);           // erf(|x| / sqrt 2)
    return 0.5f * x * (1.0f + copysignf(e, x));
}
// the same on two values at once: the polynomial runs on the packed fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32 --
// two lanes' worth of fp32 per instruction); rcp and exp stay one per value.  Same operations per element as above.
typedef __attribute__((ext_vector_type(2))) float pnp_f32x2;
__device__ __forceinline__ pnp_f32x2 gelu_erf_fast2(pnp_f32x2 x) {
    const pnp_f32x2 ax = {fabsf(x[0]), fabsf(x[1])};
    const pnp_f32x2 z = ax * 0.70710678118654752f;
    const pnp_f32x2 d = __builtin_elementwise_fma(pnp_f32x2{0.3275911f, 0.3275911f}, z, pnp_f32x2{1.0f, 1.0f});
    const pnp_f32x2 t = {__frcp_rn(d[0]), __frcp_rn(d[1])};
    pnp_f32x2 p = __builtin_elementwise_fma(pnp_f32x2{1.061405429f, 1.061405429f}, t, pnp_f32x2{-1.453152027f, -1.453152027f});
    p = __builtin_elementwise_fma(p, t, pnp_f32x2{1.421413741f, 1.421413741f});
    p = __builtin_elementwise_fma(p, t, pnp_f32x2{-0.284496736f, -0.284496736f});
    p = __builtin_elementwise_fma(p, t, pnp_f32x2{0.254829592f, 0.254829592f});
    const pnp_f32x2 nz2 = -(z * z);
    const pnp_f32x2 ex = {__expf(nz2[0]), __expf(nz2[1])};
    const pnp_f32x2 e = pnp_f32x2{1.0f, 1.0f} - p * t * ex;                 // erf(|x| / sqrt 2)
    const pnp_f32x2 es = {copysignf(e[0], x[0]), copysignf(e[1], x[1])};
    return (x * 0.5f) * (pnp_f32x2{1.0f, 1.0f} + es);
}
// x * Phi(x) with Phi(x) ~ sigmoid(x (c0 + c1 x^2 + c2 x^4)): minimax fit of the erf form on [-8, 8]
// (tools/fit_gelu.py), max |error| 2.6e-5 -- 1/150 of a bf16 ulp at 1.0.  Seven VALU + exp2 + rcp; used by
// the wide GEMM's GELU epilogue only, whose output is stored as bf16.
__device__ __forceinline__ float gelu_logistic_fit(float x) {
    const float xc = __builtin_amdgcn_fmed3f(x, -8.0f, 8.0f);
    const float x2 = xc * xc;
    // coefficients pre-multiplied by -log2(e): e = 2^(-u log2 e) = exp(-u)
    float p = fmaf(1.01426309e-3f, x2, -1.06775727e-1f);
    p = fmaf(p, x2, -2.30112133f);
    const float e = __builtin_amdgcn_exp2f(p * xc);
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
    const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

}  // namespace pnp

// ---- host-side error plumbing (no C++ exceptions cross the C ABI)
#define PNP_OK 0
#define PNP_ERR_ARG (-22)
#define PNP_ERR_HIP (-5)
#define PNP_ERR_STATE (-1)
#define PNP_ERR_NOMEM (-12)

// ---- per-device one-time set-up of a launcher.  A process may hold engines on several devices (pnp_config.device) and drives
// them from several host threads: what a launcher needs once -- the opt-in of a kernel to more than 64 KB of dynamic LDS, the
// CU count behind a persistent grid, an occupancy query -- is kept per DEVICE ORDINAL of the calling thread's current
// device, and a failure is returned, never cached.
namespace pnp {
constexpr int kMaxDevices = 32;

static inline int current_device() {
    int d = -1;
    return (hipGetDevice(&d) == hipSuccess && d >= 0 && d < kMaxDevices) ? d : -1;
}

// `done`: one bit per device ordinal, owned by the call site (function-local static).  hipFuncSetAttribute is idempotent, so two
// threads that race on the first call of a device both set it.
static inline int lds_opt_in(std::atomic<uint32_t>& done, const void* kernel, int bytes) {
    const int d = current_device();
    if (d < 0) return -5;
    if (done.load(std::memory_order_acquire) & (1u << d)) return 0;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return -5;
    done.fetch_or(1u << d, std::memory_order_release);
    return 0;
}

// compute units of the current device (0: the query failed; not cached then)
static inline int device_cu_count() {
    static std::atomic<int> cus[kMaxDevices];
    const int d = current_device();
    if (d < 0) return 0;
    int n = cus[d].load(std::memory_order_relaxed);
    if (n) return n;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, d) != hipSuccess) return 0;
    n = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    cus[d].store(n, std::memory_order_relaxed);
    return n;
}

// a small per-device cache of one int per call site (occupancy-derived grids): 0 = not yet computed on this device
struct PerDeviceInt {
    std::atomic<int> v[kMaxDevices];
    template <typename F> int get(F compute) {
        const int d = current_device();
        if (d < 0) return compute();
        int n = v[d].load(std::memory_order_relaxed);
        if (!n) {
            n = compute();
            v[d].store(n, std::memory_order_relaxed);
        }
        return n;
    }
};
}  // namespace pnp
