// ViT-side kernels: patch gather (im2col with salience-dropped patches zeroed), cls/pos rows,
// LayerNorm, and the fused patch self-attention (flash-style, MFMA QK^T / PV, LDS-staged K / V^T).
// Reference: vit.py:274-290 (forward), :91-121 (Attention), :164-167 (Block);
//            PnP_OVSS_0514_updated_segmentation.py:597-603 (zeroing dropped 16x16 blocks).
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace pnp {

// ------------------------------------------------------------------------------------------
// im2col for the 16x16/16 patch conv.  One thread per (patch row m, channel c, kernel row i):
// reads 16 contiguous fp32 pixels, writes 16 T.  Patches flagged in `dropped` are written as
// zeros: the reference zeroes those pixels in normalised space before the conv (PnP.py:602), so
// their embedding is bias + pos only.
template <typename T>
__global__ void patchify_kernel(const float* __restrict__ img, const uint8_t* __restrict__ dropped,
                                T* __restrict__ out, int B, int S, int P) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = B * P * P * 3 * 16;
    if (idx >= total) return;
    const int i = idx & 15;
    const int c = (idx >> 4) % 3;
    const int m = idx / 48;
    const int b = m / (P * P), p = m - b * P * P;
    const int py = p / P, px = p - py * P;
    T* o = out + (size_t)m * 768 + c * 256 + i * 16;
    const bool drop = dropped && dropped[m];
    const float* src = img + (((size_t)b * 3 + c) * S + (py * 16 + i)) * S + px * 16;
#pragma unroll
    for (int v = 0; v < 4; v++) {
        f32x4 x = drop ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(src + v * 4);
#pragma unroll
        for (int e = 0; e < 4; e++) o[v * 4 + e] = from_f32<T>(x[e]);
    }
}

__global__ void cls_rows_kernel(const float* __restrict__ cls, const float* __restrict__ pos,
                                float* __restrict__ x, int B, int N, int D) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * D) return;
    const int b = idx / D, d = idx - b * D;
    x[(size_t)b * N * D + d] = cls[d] + pos[d];
}

// ------------------------------------------------------------------------------------------
// LayerNorm over the last dim (D <= 1024, D % 4 == 0): one wave per row, two-pass in registers.
// Optional outputs: y (fp32), yt (T), xhat (fp32) and rstd (fp32 per row) for the backward.
template <typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ b, float eps, int rows, int D,
                                                        float* __restrict__ y, T* __restrict__ yt,
                                                        float* __restrict__ xhat, float* __restrict__ rstd_out,
                                                        bf16* __restrict__ yt_lo = nullptr) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int nv = D >> 2;
    const f32x4* xr = reinterpret_cast<const f32x4*>(x + (size_t)row * D);
    f32x4 v[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int c = lane + i * 64;
        v[i] = c < nv ? xr[c] : f32x4{0.f, 0.f, 0.f, 0.f};
        s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    }
    const float mean = wave_sum(s) / (float)D;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int c = lane + i * 64;
        if (c < nv) {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float d = v[i][e] - mean;
                ss += d * d;
            }
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)D + eps);
    if (rstd_out && lane == 0) rstd_out[row] = rstd;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int c = lane + i * 64;
        if (c >= nv) continue;
        const f32x4 wv = reinterpret_cast<const f32x4*>(w)[c];
        const f32x4 bv = reinterpret_cast<const f32x4*>(b)[c];
        f32x4 h, o;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            h[e] = (v[i][e] - mean) * rstd;
            o[e] = h[e] * wv[e] + bv[e];
        }
        if (xhat) reinterpret_cast<f32x4*>(xhat + (size_t)row * D)[c] = h;
        if (y) reinterpret_cast<f32x4*>(y + (size_t)row * D)[c] = o;
        if (yt) {
            T* p = yt + (size_t)row * D + c * 4;
            if constexpr (sizeof(T) == 2) {          // one 8-byte store per lane: 512 contiguous bytes per wave
                const bf16x4 pk = {(bf16)o[0], (bf16)o[1], (bf16)o[2], (bf16)o[3]};
                *reinterpret_cast<bf16x4*>(p) = pk;
                if (yt_lo) {                     // split-bf16 operand pair: lo = bf16(o - hi)
                    const bf16x4 pl = {(bf16)(o[0] - (float)pk[0]), (bf16)(o[1] - (float)pk[1]), (bf16)(o[2] - (float)pk[2]),
                                       (bf16)(o[3] - (float)pk[3])};
                    *reinterpret_cast<bf16x4*>(yt_lo + (size_t)row * D + c * 4) = pl;
                }
            } else {
                *reinterpret_cast<f32x4*>(p) = o;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// ViT self-attention, head_dim 64, one workgroup = 64 query rows of one (image, head); each of the
// 4 waves owns 16 queries.  Per 64-key tile: S^T = K.Q^T (key on the accumulator row, query on the
// lane column) -> online softmax entirely per lane -> O^T += V^T.P^T with P taken straight from
// the S^T accumulators.  N x N scores are never materialised (the reference does: vit.py:106-108).
//   qk : [B*N, ld_qk]  (q of head h at column h*64, k at column D + h*64)
//   vt : [D, ld_vt]    row h*64+d, column b*Npad + key   (V^T written by the transposed-V GEMM; fp32 mode only:
//                      the bf16 engine uses vit_attn32_kernel below)
//   ctx: [B*N, D]
template <typename T>
__global__ __launch_bounds__(256) void vit_attn_kernel(const T* __restrict__ qk, int ld_qk, int D,
                                                       const T* __restrict__ vt, int ld_vt, int Npad,
                                                       T* __restrict__ ctx, int N, float scale,
                                                       bf16* __restrict__ ctx_hi = nullptr, bf16* __restrict__ ctx_lo = nullptr) {
    constexpr int ROWB = 64 * Elem<T>::kBytes;
    constexpr int CPR = ROWB / 16;                 // 16-byte chunks per LDS row
    constexpr int CPT = 64 * CPR / 256;            // chunks per thread per tile
    __shared__ __attribute__((aligned(16))) char Ks[64 * ROWB];
    __shared__ __attribute__((aligned(16))) char Vs[64 * ROWB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * 64;
    const size_t row0 = (size_t)b * N;

    int qrow = q0 + wave * 16 + r;
    const bool q_valid = qrow < N;
    qrow = q_valid ? qrow : N - 1;
    Frag<T> fq[2];
    {
        const T* qp = qk + (row0 + qrow) * ld_qk + h * 64;
        glb_frag(fq[0], qp, 0, q);
        glb_frag(fq[1], qp, 1, q);
    }
    f32x4 acc_o[4];
#pragma unroll
    for (int i = 0; i < 4; i++) acc_o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;

    const int ntiles = (N + 63) / 64;
    chunk16 rk[CPT], rv[CPT];
#define PNP_LOAD_TILE(t)                                                                                       \
    _Pragma("unroll") for (int i = 0; i < CPT; i++) {                                                          \
        const int ci = tid + i * 256, row = ci / CPR, c = ci % CPR;                                            \
        int key = (t) * 64 + row;                                                                              \
        key = key < N ? key : N - 1;                                                                           \
        rk[i] = *reinterpret_cast<const chunk16*>(                                                             \
            reinterpret_cast<const char*>(qk + (row0 + key) * ld_qk + D + h * 64) + c * 16);                   \
        rv[i] = *reinterpret_cast<const chunk16*>(                                                             \
            reinterpret_cast<const char*>(vt + (size_t)(h * 64 + row) * ld_vt + (size_t)b * Npad + (t) * 64) + \
            c * 16);                                                                                           \
    }
    PNP_LOAD_TILE(0)
    for (int t = 0; t < ntiles; t++) {
        if (t > 0) __syncthreads();
#pragma unroll
        for (int i = 0; i < CPT; i++) {
            const int ci = tid + i * 256, row = ci / CPR, c = ci % CPR;
            *reinterpret_cast<chunk16*>(Ks + lds_off<ROWB>(row, c)) = rk[i];
            *reinterpret_cast<chunk16*>(Vs + lds_off<ROWB>(row, c)) = rv[i];
        }
        __syncthreads();
        if (t + 1 < ntiles) { PNP_LOAD_TILE(t + 1) }

        // S^T tiles: rows = keys (kt*16 + 4q + e), col = query r
        f32x4 s[4];
#pragma unroll
        for (int kt = 0; kt < 4; kt++) {
            s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                Frag<T> fk;
                lds_frag<ROWB>(fk, Ks, kt * 16 + r, ks, q);
                mma16(s[kt], fk, fq[ks]);
            }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; kt++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int key = t * 64 + kt * 16 + q * 4 + e;
                const float v = key < N ? s[kt][e] * scale : -INFINITY;
                s[kt][e] = v;
                mx = fmaxf(mx, v);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __expf(m_run - m_new);
        float ls = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; kt++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float p = __expf(s[kt][e] - m_new);
                s[kt][e] = p;
                ls += p;
            }
        ls += __shfl_xor(ls, 16, 64);
        ls += __shfl_xor(ls, 32, 64);
        l_run = l_run * alpha + ls;
        m_run = m_new;
#pragma unroll
        for (int i = 0; i < 4; i++) acc_o[i] *= alpha;
        // O^T += V^T . P^T
#pragma unroll
        for (int u = 0; u < 2; u++) {
            Frag<T> fp;
            pack_p(fp, s[2 * u], s[2 * u + 1]);
#pragma unroll
            for (int dt = 0; dt < 4; dt++) {
                Frag<T> fv;
                lds_frag_acc_order<ROWB>(fv, Vs, dt * 16 + r, u, q);
                mma16(acc_o[dt], fv, fp);
            }
        }
    }
#undef PNP_LOAD_TILE
    if (q_valid) {
        const float inv = 1.0f / l_run;
        T* o = ctx + (row0 + qrow) * D + h * 64 + q * 4;
#pragma unroll
        for (int dt = 0; dt < 4; dt++) {
            if constexpr (sizeof(T) == 2) {
                bf16x4 pk = {(bf16)(acc_o[dt][0] * inv), (bf16)(acc_o[dt][1] * inv), (bf16)(acc_o[dt][2] * inv),
                             (bf16)(acc_o[dt][3] * inv)};
                *reinterpret_cast<bf16x4*>(o + dt * 16) = pk;
            } else {
                f32x4 ov = acc_o[dt] * inv;
                if (ctx_hi) {                    // split-bf16 mode: the output feeds the proj GEMM as a (hi, lo) bf16 pair
                    const size_t off = (row0 + qrow) * D + h * 64 + q * 4 + dt * 16;
                    const bf16x4 ph = {(bf16)ov[0], (bf16)ov[1], (bf16)ov[2], (bf16)ov[3]};
                    const bf16x4 pl = {(bf16)(ov[0] - (float)ph[0]), (bf16)(ov[1] - (float)ph[1]), (bf16)(ov[2] - (float)ph[2]),
                                       (bf16)(ov[3] - (float)ph[3])};
                    *reinterpret_cast<bf16x4*>(ctx_hi + off) = ph;
                    *reinterpret_cast<bf16x4*>(ctx_lo + off) = pl;
                } else {
                    *reinterpret_cast<f32x4*>(o + dt * 16) = ov;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// bf16 ViT self-attention, second generation.  One wave = 32 queries of one (image, head) with
// v_mfma_f32_32x32x16_bf16; a workgroup is up to 8 such waves sharing the K / V tiles, which arrive by
// LDS-DMA into a two-slot ring (one s_barrier per 64-key tile) straight from the fused q|k|v rows.
// Against the first kernel (16 queries per wave, 16x16 tiles, register-staged tiles, V^T from its own GEMM)
// this halves the LDS fragment bytes per FLOP and per score element spends one fma + one exp2 + one add
// + half a max3 + half a cvt:
//   S^T = K.Q^T (keys on accumulator rows, the query on the lane column) -> running max per lane pair
//   -> p = exp2(s c - m c), c = scale log2 e -> O^T += V^T.P^T with P taken from the S^T registers in
//   "accumulator order"; the V^T fragment comes out of the row-major V tile through the transposing LDS
//   read ds_read_b64_tr_b16 (two 4-key blocks per lane = the same key order P has), so no V^T is ever
//   materialised in memory.
__global__ __launch_bounds__(512) void vit_attn32_kernel(const bf16* __restrict__ qk, int ld_qk, int D,
                                                         const bf16* __restrict__ vt, int ld_vt, int Npad,
                                                         bf16* __restrict__ ctx, int N, float scale, int nqw) {
    constexpr int ROWB = 128, TILE = 64 * ROWB;         // K tile [64 keys][64 d], V^T tile [64 d][64 keys]
    __shared__ __attribute__((aligned(16))) char ring[2][2 * TILE];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwaves = blockDim.x >> 6;
    const int l32 = lane & 31, hi = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const size_t row0 = (size_t)b * N;
    const int qw = blockIdx.x * nwaves + wave;           // this wave's 32-query group
    int qrow = qw * 32 + l32;
    const bool q_valid = qw < nqw && qrow < N;
    qrow = qrow < N ? qrow : N - 1;

    bf16x8 fq[4];                                       // Q as the B operand: lane = query, 8 d per k16 step
    {
        const bf16* qp = qk + (row0 + qrow) * ld_qk + h * 64 + hi * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ks++) fq[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);
    }

    // DMA: 16 pieces of 8 rows per tile pair (8 K + 8 V^T), dealt round-robin to the waves
    const char* kbase = reinterpret_cast<const char*>(qk + row0 * ld_qk + D + h * 64);
    const char* vbase = reinterpret_cast<const char*>(vt + row0 * ld_vt + h * 64);      // V rows of this image (natural layout)
    const int prow = lane >> 3, pc = lane & 7;
    auto issue_tile = [&](int t) {
        char* dst = ring[t & 1];
        for (int p = wave; p < 16; p += nwaves) {
            const int row = (p & 7) * 8 + prow;
            const int sc = pc ^ ((row >> 1) & 7);
            int key = t * 64 + row;
            key = key < N ? key : N - 1;             // keys >= N get p = 0 (masked scores), their V rows are finite copies
            const char* src = p < 8 ? kbase + (size_t)key * ld_qk * 2 + sc * 16 : vbase + (size_t)key * ld_vt * 2 + sc * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dst + p * 1024), 16, 0, 0);
        }
    };

    f32x16 o[2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int e = 0; e < 16; e++) o[i][e] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;               // l_run: this lane's share of the row sum
    const float c = scale * 1.4426950408889634f;
    const int sw = (l32 >> 1) & 7;
    const int ntiles = (N + 63) / 64;

    issue_tile(0);
    for (int t = 0; t < ntiles; t++) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                   // tile t landed; nobody reads the other slot any more
        if (t + 1 < ntiles) issue_tile(t + 1);
        const char* Ks = ring[t & 1];
        const char* Vs = Ks + TILE;

        // all fragment reads of the tile up front: the V^T reads land while the softmax runs
        bf16x8 fk[2][4];
#pragma unroll
        for (int kt = 0; kt < 2; kt++)
#pragma unroll
            for (int ks = 0; ks < 4; ks++)
                fk[kt][ks] = *reinterpret_cast<const bf16x8*>(Ks + (kt * 32 + l32) * ROWB + (((ks * 2 + hi) ^ sw) << 4));
        f32x16 s[2];
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < 2; kt++) {
            s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fk[kt][0], fq[0], zero16, 0, 0, 0);
#pragma unroll
            for (int ks = 1; ks < 4; ks++) s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fk[kt][ks], fq[ks], s[kt], 0, 0, 0);
        }
        if (t == ntiles - 1 && (N & 63)) {              // ragged last tile: keys >= N never win
#pragma unroll
            for (int kt = 0; kt < 2; kt++)
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const int key = t * 64 + kt * 32 + (e >> 2) * 8 + hi * 4 + (e & 3);
                    if (key >= N) s[kt][e] = -INFINITY;
                }
        }
        float mx;
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mx) : "v"(s[0][0]), "v"(s[0][1]), "v"(s[0][2]));
#pragma unroll
        for (int e = 3; e + 1 < 16; e += 2) asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mx) : "v"(mx), "v"(s[0][e]), "v"(s[0][e + 1]));
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mx) : "v"(mx), "v"(s[0][15]), "v"(s[1][0]));
#pragma unroll
        for (int e = 1; e + 1 < 16; e += 2) asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mx) : "v"(mx), "v"(s[1][e]), "v"(s[1][e + 1]));
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mx) : "v"(mx), "v"(s[1][15]), "v"(m_run));
        const float m_new = fmaxf(mx, __shfl_xor(mx, 32, 64));     // both halves of this query's keys (and the running max)
        {   // unconditional rescale: a branch around it makes the compiler shuttle the 32 O registers through copies
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
#pragma unroll
            for (int i = 0; i < 2; i++) o[i] *= alpha;
            l_run *= alpha;
            m_run = m_new;
        }
        const float mc = -m_run * c;
        bf16x8 fp[4];
#pragma unroll
        for (int kt = 0; kt < 2; kt++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const float p = __builtin_amdgcn_exp2f(fmaf(s[kt][e], c, mc));
                l_run += p;
                fp[kt * 2 + (e >> 3)][e & 7] = (bf16)p;
            }
        // O^T += V^T . P^T : k16 step j covers keys 16j .. 16j+15, lane half hi holds keys 16j + 4hi + (0..3) and
        // 16j + 8 + 4hi + (0..3) (accumulator order).  Transposing read: the 16 lanes of a group fetch a 4-key x 16-d
        // block (lane 4q+p supplies the address of key q, d 4p..4p+3) and lane i receives d = i of the 4 keys.
        const int grp_d = ((lane >> 4) & 1) * 16, tq = (lane >> 2) & 3, tp = lane & 3;
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int dt = 0; dt < 2; dt++) {
                typedef __attribute__((ext_vector_type(4))) short s16x4;
                s16x4 part[2];
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const int key = 16 * j + 8 * u + 4 * hi + tq;                       // row of the V tile
                    const int col = dt * 32 + grp_d + 4 * tp;                           // first of 4 d's
                    const char* a = Vs + key * ROWB + ((((col >> 3)) ^ ((key >> 1) & 7)) << 4) + (col & 7) * 2;
                    part[u] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a);
                }
                const bf16x4 lo = __builtin_bit_cast(bf16x4, part[0]), hh = __builtin_bit_cast(bf16x4, part[1]);
                const bf16x8 fv = {lo[0], lo[1], lo[2], lo[3], hh[0], hh[1], hh[2], hh[3]};
                o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fv, fp[j], o[dt], 0, 0, 0);
            }
    }
    l_run += __shfl_xor(l_run, 32, 64);
    if (q_valid) {
        const float inv = 1.0f / l_run;
        bf16* op = ctx + (row0 + qrow) * D + h * 64 + hi * 4;
#pragma unroll
        for (int dt = 0; dt < 2; dt++)
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) {
                const bf16x4 pk = {(bf16)(o[dt][g4 * 4] * inv), (bf16)(o[dt][g4 * 4 + 1] * inv), (bf16)(o[dt][g4 * 4 + 2] * inv),
                                   (bf16)(o[dt][g4 * 4 + 3] * inv)};
                *reinterpret_cast<bf16x4*>(op + dt * 32 + g4 * 8) = pk;
            }
    }
}

#ifdef PNP_DEV
// DEV diagnostics: per-(wave, key tile) issue-time clock stamps of workgroup (0, 0, 0) of the split-bf16 attention
// (tools/attn_x3_probe.py --stamps): [wave][tile][6] = loop top, barrier passed, DMA issued, S issued, softmax done, P.V issued
__device__ unsigned long long* g_attn_stamps = nullptr;
__device__ int g_attn_ablate = 0;
extern "C" int pnp_dev_attn_ablate(int on) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_attn_ablate), &on, sizeof(on)) == hipSuccess ? PNP_OK : PNP_ERR_HIP;
}
extern "C" int pnp_dev_attn_stamps(unsigned long long* d_buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_attn_stamps), &d_buf, sizeof(d_buf)) == hipSuccess ? PNP_OK : PNP_ERR_HIP;
}
#define PNP_ATTN_STAMP(k)                                                                                       \
    do {                                                                                                        \
        if (stamps && lane == 0) {                                                                              \
            __builtin_amdgcn_sched_barrier(0);                                                                  \
            stamps[((size_t)wave * ntiles + t) * 6 + (k)] = __builtin_readcyclecounter();                       \
            __builtin_amdgcn_sched_barrier(0);                                                                  \
        }                                                                                                       \
    } while (0)
#else
#define PNP_ATTN_STAMP(k) do {} while (0)
#endif

// Split-bf16 ("bf16x3") form of the kernel above for compute mode 2: q, k, v arrive as (hi, lo) bf16 pairs (the fused
// q|k|v rows of the split-output GEMM), both products run as three bf16 MFMA passes -- S = K_hi.Q_hi + K_hi.Q_lo + K_lo.Q_hi,
// O += V_hi.P_hi + V_hi.P_lo + V_lo.P_hi with P split after the exp2 -- and the context leaves as a (hi, lo) pair for the
// proj GEMM.  Same tiling, ring and softmax; the ring slot holds four tiles (K_hi, V_hi, K_lo, V_lo).
__global__ __launch_bounds__(768) void vit_attn32_x3_kernel(const bf16* __restrict__ qk, const bf16* __restrict__ qk_lo, int ld_qk, int D,
                                                            const bf16* __restrict__ vt, const bf16* __restrict__ vt_lo, int ld_vt,
                                                            bf16* __restrict__ ctx, bf16* __restrict__ ctx_lo, int N, float scale, int nqw) {
    constexpr int ROWB = 128, TILE = 64 * ROWB;         // K tile [64 keys][64 d], V tile [64 keys][64 d] (transposed on the read)
    __shared__ __attribute__((aligned(16))) char ring[2][4 * TILE];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwaves = blockDim.x >> 6;
    const int l32 = lane & 31, hi = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const size_t row0 = (size_t)b * N;
    const int qw = blockIdx.x * nwaves + wave;
    int qrow = qw * 32 + l32;
    const bool q_valid = qw < nqw && qrow < N;
    qrow = qrow < N ? qrow : N - 1;

    bf16x8 fq[4], fql[4];
    {
        const size_t off = (row0 + qrow) * ld_qk + h * 64 + hi * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            fq[ks] = *reinterpret_cast<const bf16x8*>(qk + off + ks * 16);
            fql[ks] = *reinterpret_cast<const bf16x8*>(qk_lo + off + ks * 16);
        }
    }
    const size_t koff = (row0 * ld_qk + D + h * 64) * 2, voff = (row0 * ld_vt + h * 64) * 2;
    // A DMA piece = 8 key rows x 128 B of one of the four operand tiles (pieces 0-7 K_hi, 8-15 V_hi, 16-23 K_lo, 24-31 V_lo of
    // key tile t; wave w owns pieces w, w + nwaves, ...).  Its address is a UNIFORM 64-bit base (operand, first row of the
    // piece: scalar registers) + a 32-bit lane offset (row within the piece, swizzled 16-byte chunk).  The four operand bases
    // are formed by integer masks from one base and three uniform differences -- never by indexing or selecting pointers:
    // hipcc keeps such pointers (the `kb[2]` / `vb[2]` arrays this kernel had) in scratch memory and loads the selected one per
    // piece, and that scratch load then waits -- vmcnt is one in-order counter -- for every DMA piece issued before it: ~700
    // cycles per piece, 3600 of the 9500 cycles a key tile took (tools/attn_x3_probe.py --dev --stamps): 190 -> 149 us at 442
    // tokens, 758 -> 606 us at 2305.  Rows past N - 1 (ragged last tile) re-read row N - 1: their scores are masked, their V
    // rows multiply exact zeros and must be finite.
    const uint64_t k0 = reinterpret_cast<uint64_t>(qk) + koff;
    const uint64_t d_klo = reinterpret_cast<uint64_t>(qk_lo) + koff - k0;
    const uint64_t d_v = reinterpret_cast<uint64_t>(vt) + voff - k0;
    const uint64_t d_vlo = reinterpret_cast<uint64_t>(vt_lo) + voff - k0;
    const int prow = lane >> 3, pc = lane & 7;
    const uint32_t sc_even = (uint32_t)(pc ^ ((prow >> 1) & 7)) << 4, sc_odd = (uint32_t)(pc ^ ((4 + (prow >> 1)) & 7)) << 4;
    auto issue_tile = [&](int t) {
        for (int p = wave; p < 32; p += nwaves) {
            const uint64_t m_v = 0ull - (uint64_t)((p >> 3) & 1), m_lo = 0ull - (uint64_t)(p >> 4);
            const uint64_t d = (m_v & m_lo & d_vlo) | (m_v & ~m_lo & d_v) | (~m_v & m_lo & d_klo);
            const uint32_t ld2 = ((uint32_t)ld_qk + ((uint32_t)m_v & (uint32_t)(ld_vt - ld_qk))) * 2u;
            int rs = t * 64 + (p & 7) * 8;
#ifdef PNP_DEV
            if (g_attn_ablate) rs = (p & 7) * 8;          // timing-only (results garbage): every tile fetches the first one's rows (cache-hot)
#endif
            rs = rs < N - 1 ? rs : N - 1;
            const char* const sbase = reinterpret_cast<const char*>(k0 + d + (uint64_t)((uint32_t)rs * ld2));
            int lr = rs + prow;
            lr = (lr < N ? lr : N - 1) - rs;
            const uint32_t loff = (uint32_t)lr * ld2 + ((p & 1) ? sc_odd : sc_even);
            // LDS-DMA as inline asm (round 4): hipcc models __builtin_amdgcn_global_load_lds as a FLAT access that may touch
            // both memories, and while one is pending EVERY wait it inserts for an LDS read is forced to lgkmcnt(0)
            // (SIInsertWaitcnts "pending flat") -- with a tile always in flight that is every wait of this loop: each K / V
            // fragment read was waited for right where it was issued.  Hidden from the compiler its LDS waits are counted;
            // the landing of the pieces is covered by the explicit vmcnt(0) in front of the tile barrier.
            const uint32_t lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(ring[t & 1] + p * 1024);
            const uint64_t sb = reinterpret_cast<uint64_t>(sbase);           // uniform; made scalar explicitly for the "s" operand
            const uint64_t sbu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(sb >> 32)) << 32) |
                                 (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)sb);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(loff), "s"(sbu),
                         "s"(__builtin_amdgcn_readfirstlane((int)lds))
                         : "memory");      // (m0 is a reserved register to hipcc: not nameable as a clobber; the shipped code
                                           //  object is checked for foreign m0 uses by tests/test_cabi_cpu.py)
        }
    };

    f32x16 o[2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int e = 0; e < 16; e++) o[i][e] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float c = scale * 1.4426950408889634f;
    const int sw = (l32 >> 1) & 7;
    const int ntiles = (N + 63) / 64;

#ifdef PNP_DEV
    unsigned long long* const stamps = (blockIdx.x | blockIdx.y | blockIdx.z) == 0 ? g_attn_stamps : nullptr;
#endif
    issue_tile(0);
    for (int t = 0; t < ntiles; t++) {
        PNP_ATTN_STAMP(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        PNP_ATTN_STAMP(1);
        if (t + 1 < ntiles) issue_tile(t + 1);
        PNP_ATTN_STAMP(2);
        const char* Ks = ring[t & 1];
        const char* Vs = Ks + TILE;
        const char* Kl = Ks + 2 * TILE;
        const char* Vl = Ks + 3 * TILE;

        f32x16 s[2];
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < 2; kt++) {
            s[kt] = zero16;
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {            // small terms first, the hi.hi term last
                const int a = (kt * 32 + l32) * ROWB + (((ks * 2 + hi) ^ sw) << 4);
                const bf16x8 kl = *reinterpret_cast<const bf16x8*>(Kl + a);
                const bf16x8 kh = *reinterpret_cast<const bf16x8*>(Ks + a);
                s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, fq[ks], s[kt], 0, 0, 0);
                s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, fql[ks], s[kt], 0, 0, 0);
                s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, fq[ks], s[kt], 0, 0, 0);
            }
        }
        PNP_ATTN_STAMP(3);
        if (t == ntiles - 1 && (N & 63)) {
#pragma unroll
            for (int kt = 0; kt < 2; kt++)
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const int key = t * 64 + kt * 32 + (e >> 2) * 8 + hi * 4 + (e & 3);
                    if (key >= N) s[kt][e] = -INFINITY;
                }
        }
        float mx = m_run;
#pragma unroll
        for (int kt = 0; kt < 2; kt++)
#pragma unroll
            for (int e = 0; e < 16; e++) mx = fmaxf(mx, s[kt][e]);
        const float m_new = fmaxf(mx, __shfl_xor(mx, 32, 64));
        {
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
#pragma unroll
            for (int i = 0; i < 2; i++) o[i] *= alpha;
            l_run *= alpha;
            m_run = m_new;
        }
        const float mc = -m_run * c;
        bf16x8 fp[4], fpl[4];
#pragma unroll
        for (int kt = 0; kt < 2; kt++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const float p = __builtin_amdgcn_exp2f(fmaf(s[kt][e], c, mc));
                l_run += p;
                const bf16 ph = (bf16)p;
                fp[kt * 2 + (e >> 3)][e & 7] = ph;
                fpl[kt * 2 + (e >> 3)][e & 7] = (bf16)(p - (float)ph);
            }
        PNP_ATTN_STAMP(4);
        const int grp_d = ((lane >> 4) & 1) * 16, tq = (lane >> 2) & 3, tp = lane & 3;
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int dt = 0; dt < 2; dt++) {
                typedef __attribute__((ext_vector_type(4))) short s16x4;
                bf16x8 fv[2];
#pragma unroll
                for (int part = 0; part < 2; part++) {
                    const char* V = part ? Vl : Vs;
                    s16x4 pr[2];
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        const int key = 16 * j + 8 * u + 4 * hi + tq;
                        const int col = dt * 32 + grp_d + 4 * tp;
                        const char* a = V + key * ROWB + ((((col >> 3)) ^ ((key >> 1) & 7)) << 4) + (col & 7) * 2;
                        pr[u] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a);
                    }
                    const bf16x4 lo = __builtin_bit_cast(bf16x4, pr[0]), hh = __builtin_bit_cast(bf16x4, pr[1]);
                    fv[part] = bf16x8{lo[0], lo[1], lo[2], lo[3], hh[0], hh[1], hh[2], hh[3]};
                }
                o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fv[1], fp[j], o[dt], 0, 0, 0);
                o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fv[0], fpl[j], o[dt], 0, 0, 0);
                o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fv[0], fp[j], o[dt], 0, 0, 0);
            }
        PNP_ATTN_STAMP(5);
    }
    l_run += __shfl_xor(l_run, 32, 64);
    if (q_valid) {
        const float inv = 1.0f / l_run;
        const size_t off = (row0 + qrow) * D + h * 64 + hi * 4;
#pragma unroll
        for (int dt = 0; dt < 2; dt++)
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) {
                bf16x4 ph, pl;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const float v = o[dt][g4 * 4 + e] * inv;
                    ph[e] = (bf16)v;
                    pl[e] = (bf16)(v - (float)ph[e]);
                }
                *reinterpret_cast<bf16x4*>(ctx + off + dt * 32 + g4 * 8) = ph;
                *reinterpret_cast<bf16x4*>(ctx_lo + off + dt * 32 + g4 * 8) = pl;
            }
    }
}

// ------------------------------------------------------------------------------------------ host
// ctx_lo != nullptr (fp32 kernel only): the output is written as a split-bf16 pair (ctx = hi, ctx_lo = lo)
int vit_attention(int bf, const void* qk, int ld_qk, int D, const void* vt, int ld_vt, int Npad, void* ctx, int B,
                  int H, int N, float scale, hipStream_t s, void* ctx_lo) {
    if (D != H * 64 || Npad % 64 || Npad < N) return PNP_ERR_ARG;
    if (bf) {       // bf16: `vt` is V in the NATURAL layout [B*N, ld_vt] (the fused q|k|v rows), transposed on the LDS read
        if (ctx_lo) return PNP_ERR_ARG;
        const int nqw = (N + 31) / 32;                          // 32-query waves per (image, head)
        const int max_wpb = 8;
        const int nblk = (nqw + max_wpb - 1) / max_wpb, wpb = (nqw + nblk - 1) / nblk;
        hipLaunchKernelGGL(vit_attn32_kernel, dim3(nblk, H, B), dim3(wpb * 64), 0, s, (const bf16*)qk, ld_qk, D,
                           (const bf16*)vt, ld_vt, Npad, (bf16*)ctx, N, scale, nqw);
        return hipGetLastError() == hipSuccess ? PNP_OK : PNP_ERR_HIP;
    }
    dim3 grid((N + 63) / 64, H, B);
    hipLaunchKernelGGL((vit_attn_kernel<float>), grid, dim3(256), 0, s, (const float*)qk, ld_qk, D, (const float*)vt, ld_vt,
                       Npad, ctx_lo ? nullptr : (float*)ctx, N, scale, ctx_lo ? (bf16*)ctx : nullptr, (bf16*)ctx_lo);
    return hipGetLastError() == hipSuccess ? PNP_OK : PNP_ERR_HIP;
}

// split-bf16 attention (compute mode 2): q|k|v rows as (hi, lo) bf16 pairs [B*N, ld_qk] each, context as a pair
int vit_attention_x3(const void* qkv_hi, const void* qkv_lo, int ld_qk, int D, void* ctx_hi, void* ctx_lo, int B, int H, int N,
                     float scale, hipStream_t s) {
    if (D != H * 64 || !qkv_hi || !qkv_lo || !ctx_hi || !ctx_lo) return PNP_ERR_ARG;
    const int nqw = (N + 31) / 32;
    // one workgroup per CU (157 registers: 12 wave slots, two 7-wave workgroups do not fit and smaller ones lose more to the
    // per-tile hand-over than they win): long sequences take up to 12 waves per workgroup (-3 % at 2305 tokens), 442 tokens
    // are 14 waves = two workgroups of 7
    int max_wpb = nqw > 24 ? 12 : 8;
#ifdef PNP_DEV
    if (getenv("PNP_ATTN_WPB")) max_wpb = atoi(getenv("PNP_ATTN_WPB"));   // <= 12 (launch bound 768 threads)
#endif
    const int nblk = (nqw + max_wpb - 1) / max_wpb, wpb = (nqw + nblk - 1) / nblk;
    const bf16 *qh = (const bf16*)qkv_hi, *ql = (const bf16*)qkv_lo;
    hipLaunchKernelGGL(vit_attn32_x3_kernel, dim3(nblk, H, B), dim3(wpb * 64), 0, s, qh, ql, ld_qk, D, qh + 2 * D, ql + 2 * D, ld_qk,
                       (bf16*)ctx_hi, (bf16*)ctx_lo, N, scale, nqw);
    return hipGetLastError() == hipSuccess ? PNP_OK : PNP_ERR_HIP;
}

int patchify(int bf, const float* img, const uint8_t* dropped, void* out, int B, int S, int P, hipStream_t s) {
    const int total = B * P * P * 48;
    const int nb = (total + 255) / 256;
    if (bf) hipLaunchKernelGGL((patchify_kernel<bf16>), dim3(nb), dim3(256), 0, s, img, dropped, (bf16*)out, B, S, P);
    else hipLaunchKernelGGL((patchify_kernel<float>), dim3(nb), dim3(256), 0, s, img, dropped, (float*)out, B, S, P);
    return hipGetLastError() == hipSuccess ? PNP_OK : PNP_ERR_HIP;
}

int cls_rows(const float* cls, const float* pos, float* x, int B, int N, int D, hipStream_t s) {
    hipLaunchKernelGGL(cls_rows_kernel, dim3((B * D + 255) / 256), dim3(256), 0, s, cls, pos, x, B, N, D);
    return hipGetLastError() == hipSuccess ? PNP_OK : PNP_ERR_HIP;
}

// yt_lo != nullptr (with bf != 0): yt / yt_lo receive the split-bf16 pair of the output
int layernorm(int bf, const float* x, const float* w, const float* b, float eps, int rows, int D, float* y, void* yt,
              float* xhat, float* rstd, hipStream_t s, void* yt_lo) {
    if (D > 1024 || D % 4) return PNP_ERR_ARG;
    if (yt_lo && !bf) return PNP_ERR_ARG;
    const int nb = (rows + 3) / 4;
    if (bf) hipLaunchKernelGGL((layernorm_kernel<bf16>), dim3(nb), dim3(256), 0, s, x, w, b, eps, rows, D, y, (bf16*)yt, xhat, rstd, (bf16*)yt_lo);
    else hipLaunchKernelGGL((layernorm_kernel<float>), dim3(nb), dim3(256), 0, s, x, w, b, eps, rows, D, y, (float*)yt, xhat, rstd, (bf16*)nullptr);
    return hipGetLastError() == hipSuccess ? PNP_OK : PNP_ERR_HIP;
}

}  // namespace pnp
