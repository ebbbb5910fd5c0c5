// Text-side (BERT + cross-attention) kernels and the analytic backward that yields
// d(sum_b logit[b,1]) / d(cross-attention probabilities).
// Reference: med.py:88-123 (embeddings), :191-311 (attention; probs stash + grad hook at :280-283),
//            :321-325 / :393-411 (post-LN dense blocks), :776-852 (additive masks),
//            blip_image_text_matching.py:238-249 (enc token, itm head), :399-404 (loss + backward).
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace pnp {

// ------------------------------------------------------------------------------------------
// word + position embeddings (token 0 of every caption is replaced by [ENC]).
__global__ void text_embed_kernel(const int64_t* __restrict__ ids, int ld_ids, const float* __restrict__ word,
                                  const float* __restrict__ pos, float* __restrict__ out, int B, int L, int H,
                                  int enc_id, int vocab) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int hv = H >> 2;
    if (idx >= B * L * hv) return;
    const int c = idx % hv, row = idx / hv;
    const int b = row / L, l = row - b * L;
    int64_t id = l == 0 ? (int64_t)enc_id : ids[(size_t)b * ld_ids + l];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const f32x4 w = reinterpret_cast<const f32x4*>(word + (size_t)id * H)[c];
    const f32x4 p = reinterpret_cast<const f32x4*>(pos + (size_t)l * H)[c];
    reinterpret_cast<f32x4*>(out + (size_t)row * H)[c] = w + p;
}

// ------------------------------------------------------------------------------------------
// Text self-attention (L <= 192 tokens -- longer captions: text_self_attn_long_kernel below --, head_dim 64): one workgroup per (head, image), k/v staged
// in LDS as fp32 (row stride 65 -> conflict-free column walks), one wave per query row.
// qkv: [B*L, 3H] (q | k | v), mask: (B, ld_mask) int64 (1 = attend), additive -10000 like
// med.py:851.  ctx: [B*L, H] T.  probs (optional): fp32 [B, heads, L, L] stash for the backward.
constexpr int TXT_MAX_L = 192;

template <typename T, int NW = 4>
__global__ __launch_bounds__(NW * 64) void text_self_attn_kernel(const T* __restrict__ qkv, const int64_t* __restrict__ mask,
                                                                 int ld_mask, T* __restrict__ ctx,
                                                                 float* __restrict__ probs, int L, int H) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ks = reinterpret_cast<float*>(smem);          // [L][65]
    float* vs = ks + L * 65;                             // [L][65]
    float* pw = vs + L * 65;                             // [NW waves][TXT_MAX_L]
    float* madd = pw + NW * TXT_MAX_L;                   // [L]
    float* qw = madd + TXT_MAX_L;                        // [NW waves][64]
    const int h = blockIdx.x, b = blockIdx.y, nh = gridDim.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t row0 = (size_t)b * L;
    if constexpr (std::is_same<T, float>::value) {
        // K / V rows in 16-byte pieces, four of each in flight per thread (round 6: with one 4-byte load per array and trip the
        // staging of 2 x L x 64 floats was ~40 dependent round trips to L2 -- most of the kernel at L = 85 ... 155)
        const int npc = L * 16;                              // 16-byte pieces per array
        for (int i0 = tid; i0 < npc; i0 += 4 * NW * 64) {
            f32x4 kv[4], vv[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = i0 + u * NW * 64;
                const int j = (i < npc ? i : npc - 1) >> 4, c4 = (i & 15) * 4;
                const float* src = reinterpret_cast<const float*>(qkv) + (row0 + j) * 3 * H + H + h * 64 + c4;
                kv[u] = *reinterpret_cast<const f32x4*>(src);
                vv[u] = *reinterpret_cast<const f32x4*>(src + H);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = i0 + u * NW * 64;
                if (i < npc) {
                    const int j = i >> 4, c4 = (i & 15) * 4;
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        ks[j * 65 + c4 + e] = kv[u][e];
                        vs[j * 65 + c4 + e] = vv[u][e];
                    }
                }
            }
        }
    } else {
        for (int i = tid; i < L * 64; i += NW * 64) {
            const int j = i >> 6, d = i & 63;
            ks[j * 65 + d] = to_f32(qkv[(row0 + j) * 3 * H + H + h * 64 + d]);
            vs[j * 65 + d] = to_f32(qkv[(row0 + j) * 3 * H + 2 * H + h * 64 + d]);
        }
    }
    for (int j = tid; j < L; j += NW * 64) madd[j] = (1.0f - (float)mask[(size_t)b * ld_mask + j]) * -10000.0f;
    __syncthreads();
    float* myp = pw + wave * TXT_MAX_L;
    // query rows are dealt over the waves of gridDim.z workgroups of the same (head, image) (each stages the head's K / V:
    // a handful of workgroups cannot fill the chip when B * heads < CUs -- ADE20K at 768^2: 8 images x 12 heads, L = 155)
    for (int i = wave + NW * blockIdx.z; i < L; i += NW * gridDim.z) {
        float* myq = qw + wave * 64;
        myq[lane] = to_f32(qkv[(row0 + i) * 3 * H + h * 64 + lane]);
        __builtin_amdgcn_wave_barrier();
        float s[3];
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int j = lane + c * 64;
            float acc = 0.f;
            if (j < L) {
                // four partial sums (d mod 4): the kernel runs one to three waves per SIMD and was waiting on ONE dependent fma
                // chain of 64 links per score (round 6)
                float a4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int d = 0; d < 64; d += 4) {
#pragma unroll
                    for (int u = 0; u < 4; u++) a4[u] += myq[d + u] * ks[j * 65 + d + u];
                }
                acc = (a4[0] + a4[1]) + (a4[2] + a4[3]);
                acc = acc * 0.125f + madd[j];
            } else {
                acc = -INFINITY;
            }
            s[c] = acc;
            mx = fmaxf(mx, acc);
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            s[c] = (lane + c * 64 < L) ? __expf(s[c] - mx) : 0.f;
            sum += s[c];
        }
        sum = wave_sum(sum);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int j = lane + c * 64;
            if (j < L) {
                const float p = s[c] * inv;
                myp[j] = p;
                if (probs) probs[(((size_t)b * nh + h) * L + i) * L + j] = p;
            }
        }
        __builtin_amdgcn_wave_barrier();
        float o4[4] = {0.f, 0.f, 0.f, 0.f};                   // likewise: four partial sums over the keys (j mod 4)
        int j = 0;
        for (; j + 4 <= L; j += 4) {
#pragma unroll
            for (int u = 0; u < 4; u++) o4[u] += myp[j + u] * vs[(j + u) * 65 + lane];
        }
        for (; j < L; j++) o4[j & 3] += myp[j] * vs[j * 65 + lane];
        const float o = (o4[0] + o4[1]) + (o4[2] + o4[3]);
        ctx[(row0 + i) * H + h * 64 + lane] = from_f32<T>(o);
        __builtin_amdgcn_wave_barrier();
    }
}

// Backward of the text self-attention: given dctx [B*L, H] (fp32), the stashed probs and q/k/v,
// writes dqkv [B*L, 3H] (T).  Same staging; dS goes through a global scratch [B, heads, L, L].
// PHASE 0: both phases in one workgroup per (head, image).  PHASE 1 / 2: phase A / phase B alone, as two launches whose rows
// (query rows in A, key rows in B) are dealt over gridDim.z workgroups per (head, image) -- for batches too small to fill the
// chip with one workgroup each (the launch boundary is the grid-wide hand-over of dS).
template <typename T, int PHASE = 0>
__global__ __launch_bounds__(256) void text_self_attn_bwd_kernel(const T* __restrict__ qkv, const float* __restrict__ dctx,
                                                                 const float* __restrict__ probs,
                                                                 float* __restrict__ ds_scratch, T* __restrict__ dqkv,
                                                                 int L, int H) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* as = reinterpret_cast<float*>(smem);          // [L][65]  k, later q
    float* bs = as + L * 65;                             // [L][65]  v, later dctx
    float* pw = bs + L * 65;                             // [4][TXT_MAX_L]
    float* qw = pw + 5 * TXT_MAX_L;                      // [4 waves][64]
    const int h = blockIdx.x, b = blockIdx.y, nh = gridDim.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t row0 = (size_t)b * L;
    const float* P = probs + ((size_t)b * nh + h) * L * L;
    float* dS = ds_scratch + ((size_t)b * nh + h) * L * L;
    float* myp = pw + wave * TXT_MAX_L;
    if (PHASE != 2) {
    for (int i = tid; i < L * 64; i += 256) {
        const int j = i >> 6, d = i & 63;
        as[j * 65 + d] = to_f32(qkv[(row0 + j) * 3 * H + H + h * 64 + d]);
        bs[j * 65 + d] = to_f32(qkv[(row0 + j) * 3 * H + 2 * H + h * 64 + d]);
    }
    __syncthreads();
    // phase A: per query row i: dP_ij = dctx_i . v_j ; dS = P (dP - sum_j dP P) ; dq_i = dS k / 8
    for (int i = wave + 4 * blockIdx.z; i < L; i += 4 * gridDim.z) {
        float* myg = qw + wave * 64;
        myg[lane] = dctx[(row0 + i) * H + h * 64 + lane];
        __builtin_amdgcn_wave_barrier();
        float dp[3], pv[3];
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int j = lane + c * 64;
            float acc = 0.f;
            if (j < L) {
                float a4[4] = {0.f, 0.f, 0.f, 0.f};          // four partial sums, as in the forward kernel (round 6)
#pragma unroll
                for (int d = 0; d < 64; d += 4) {
#pragma unroll
                    for (int u = 0; u < 4; u++) a4[u] += myg[d + u] * bs[j * 65 + d + u];
                }
                acc = (a4[0] + a4[1]) + (a4[2] + a4[3]);
                pv[c] = P[(size_t)i * L + j];
            } else {
                pv[c] = 0.f;
            }
            dp[c] = acc;
            dot += acc * pv[c];
        }
        dot = wave_sum(dot);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int j = lane + c * 64;
            if (j < L) {
                const float v = pv[c] * (dp[c] - dot);
                myp[j] = v;
                dS[(size_t)i * L + j] = v;
            }
        }
        __builtin_amdgcn_wave_barrier();
        float o4[4] = {0.f, 0.f, 0.f, 0.f};
        int j = 0;
        for (; j + 4 <= L; j += 4) {
#pragma unroll
            for (int u = 0; u < 4; u++) o4[u] += myp[j + u] * as[(j + u) * 65 + lane];
        }
        for (; j < L; j++) o4[j & 3] += myp[j] * as[j * 65 + lane];
        const float o = (o4[0] + o4[1]) + (o4[2] + o4[3]);
        dqkv[(row0 + i) * 3 * H + h * 64 + lane] = from_f32<T>(o * 0.125f);
        __builtin_amdgcn_wave_barrier();
    }
    if (PHASE == 1) return;
    __syncthreads();
    }
    // phase B: restage q and dctx, then per key row j: dk_j = dS^T q / 8 ; dv_j = P^T dctx
    for (int i = tid; i < L * 64; i += 256) {
        const int j = i >> 6, d = i & 63;
        as[j * 65 + d] = to_f32(qkv[(row0 + j) * 3 * H + h * 64 + d]);
        bs[j * 65 + d] = dctx[(row0 + j) * H + h * 64 + d];
    }
    __threadfence_block();
    __syncthreads();
    for (int j = wave + 4 * blockIdx.z; j < L; j += 4 * gridDim.z) {
        float k4[4] = {0.f, 0.f, 0.f, 0.f}, v4[4] = {0.f, 0.f, 0.f, 0.f};
        int i = 0;
        for (; i + 4 <= L; i += 4) {
            float ds[4], pp[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                ds[u] = dS[(size_t)(i + u) * L + j];
                pp[u] = P[(size_t)(i + u) * L + j];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                k4[u] += ds[u] * as[(i + u) * 65 + lane];
                v4[u] += pp[u] * bs[(i + u) * 65 + lane];
            }
        }
        for (; i < L; i++) {
            k4[i & 3] += dS[(size_t)i * L + j] * as[i * 65 + lane];
            v4[i & 3] += P[(size_t)i * L + j] * bs[i * 65 + lane];
        }
        const float dk = (k4[0] + k4[1]) + (k4[2] + k4[3]), dv = (v4[0] + v4[1]) + (v4[2] + v4[3]);
        dqkv[(row0 + j) * 3 * H + H + h * 64 + lane] = from_f32<T>(dk * 0.125f);
        dqkv[(row0 + j) * 3 * H + 2 * H + h * 64 + lane] = from_f32<T>(dv);
    }
}

// ------------------------------------------------------------------------------------------
// Long captions, TXT_MAX_L < L <= TXT_LONG_L (the reference tokenises to max_length = 500 and BERT's position table ends at 512:
// PnP.py:271,318, B/blip_image_text_matching.py:48,234).  Two fp32 [L][65] arrays no longer fit the 160 KB of LDS, so the same
// one-wave-per-row scheme runs in PHASES over ONE staged array; what a later phase needs of an earlier one goes through global
// memory -- the probabilities through the layer's stash (or a scratch of the same shape), dS through the scratch the short
// kernel uses too.  Same arithmetic per row as the short kernels (which, since round 6, add their dot products in four partial sums).
constexpr int TXT_LONG_L = 512;
constexpr int TXT_LONG_CH = TXT_LONG_L / 64;

template <typename T>
__device__ __forceinline__ void stage_rows65(float* dst, const T* __restrict__ src, size_t ld, int L, int tid) {
    for (int i = tid; i < L * 64; i += 256) {
        const int j = i >> 6, d = i & 63;
        dst[j * 65 + d] = to_f32(src[(size_t)j * ld + d]);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void text_self_attn_long_kernel(const T* __restrict__ qkv, const int64_t* __restrict__ mask,
                                                                  int ld_mask, T* __restrict__ ctx, float* __restrict__ pglob,
                                                                  int L, int H) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* st = reinterpret_cast<float*>(smem);          // [L][65]: keys, then values
    float* pw = st + L * 65;                             // [4 waves][TXT_LONG_L]
    float* madd = pw + 4 * TXT_LONG_L;                   // [TXT_LONG_L]
    float* qw = madd + TXT_LONG_L;                       // [4 waves][64]
    const int h = blockIdx.x, b = blockIdx.y, nh = gridDim.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t row0 = (size_t)b * L;
    float* P = pglob + ((size_t)b * nh + h) * L * L;
    stage_rows65(st, qkv + row0 * 3 * H + H + h * 64, (size_t)3 * H, L, tid);
    for (int j = tid; j < L; j += 256) madd[j] = (1.0f - (float)mask[(size_t)b * ld_mask + j]) * -10000.0f;
    __syncthreads();
    float* myq = qw + wave * 64;
    for (int i = wave; i < L; i += 4) {                  // phase 1: scores, softmax, probabilities -> global
        myq[lane] = to_f32(qkv[(row0 + i) * 3 * H + h * 64 + lane]);
        __builtin_amdgcn_wave_barrier();
        float sc[TXT_LONG_CH];
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < TXT_LONG_CH; c++) {
            const int j = lane + c * 64;
            float acc = -INFINITY;
            if (j < L) {
                acc = 0.f;
                for (int d = 0; d < 64; d++) acc += myq[d] * st[j * 65 + d];
                acc = acc * 0.125f + madd[j];
            }
            sc[c] = acc;
            mx = fmaxf(mx, acc);
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < TXT_LONG_CH; c++) {
            sc[c] = (lane + c * 64 < L) ? __expf(sc[c] - mx) : 0.f;
            sum += sc[c];
        }
        sum = wave_sum(sum);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int c = 0; c < TXT_LONG_CH; c++) {
            const int j = lane + c * 64;
            if (j < L) P[(size_t)i * L + j] = sc[c] * inv;
        }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();                                     // every wave is done with the keys
    stage_rows65(st, qkv + row0 * 3 * H + 2 * H + h * 64, (size_t)3 * H, L, tid);
    __syncthreads();
    float* myp = pw + wave * TXT_LONG_L;
    for (int i = wave; i < L; i += 4) {                  // phase 2: ctx = P.V (a lane reads back exactly what it wrote)
        for (int j = lane; j < L; j += 64) myp[j] = P[(size_t)i * L + j];
        __builtin_amdgcn_wave_barrier();
        float o = 0.f;
        for (int j = 0; j < L; j++) o += myp[j] * st[j * 65 + lane];
        ctx[(row0 + i) * H + h * 64 + lane] = from_f32<T>(o);
        __builtin_amdgcn_wave_barrier();
    }
}

template <typename T>
__global__ __launch_bounds__(256) void text_self_attn_bwd_long_kernel(const T* __restrict__ qkv, const float* __restrict__ dctx,
                                                                      const float* __restrict__ probs,
                                                                      float* __restrict__ ds_scratch, T* __restrict__ dqkv,
                                                                      int L, int H) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* st = reinterpret_cast<float*>(smem);          // [L][65]: v, k, q, dctx in turn
    float* pw = st + L * 65;                             // [4 waves][TXT_LONG_L]
    float* qw = pw + 4 * TXT_LONG_L;                     // [4 waves][64]
    const int h = blockIdx.x, b = blockIdx.y, nh = gridDim.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t row0 = (size_t)b * L;
    const float* P = probs + ((size_t)b * nh + h) * L * L;
    float* dS = ds_scratch + ((size_t)b * nh + h) * L * L;
    float* myp = pw + wave * TXT_LONG_L;
    float* myg = qw + wave * 64;
    // phase A1 (values staged): dP_ij = dctx_i . v_j ; dS = P (dP - sum_j dP P) -> global
    stage_rows65(st, qkv + row0 * 3 * H + 2 * H + h * 64, (size_t)3 * H, L, tid);
    __syncthreads();
    for (int i = wave; i < L; i += 4) {
        myg[lane] = dctx[(row0 + i) * H + h * 64 + lane];
        __builtin_amdgcn_wave_barrier();
        float dp[TXT_LONG_CH], pv[TXT_LONG_CH];
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < TXT_LONG_CH; c++) {
            const int j = lane + c * 64;
            float acc = 0.f;
            pv[c] = 0.f;
            if (j < L) {
                for (int d = 0; d < 64; d++) acc += myg[d] * st[j * 65 + d];
                pv[c] = P[(size_t)i * L + j];
            }
            dp[c] = acc;
            dot += acc * pv[c];
        }
        dot = wave_sum(dot);
#pragma unroll
        for (int c = 0; c < TXT_LONG_CH; c++) {
            const int j = lane + c * 64;
            if (j < L) dS[(size_t)i * L + j] = pv[c] * (dp[c] - dot);
        }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // phase A2 (keys staged): dq_i = dS_i . k / 8 (a lane reads back the dS entries it wrote)
    stage_rows65(st, qkv + row0 * 3 * H + H + h * 64, (size_t)3 * H, L, tid);
    __syncthreads();
    for (int i = wave; i < L; i += 4) {
        for (int j = lane; j < L; j += 64) myp[j] = dS[(size_t)i * L + j];
        __builtin_amdgcn_wave_barrier();
        float o = 0.f;
        for (int j = 0; j < L; j++) o += myp[j] * st[j * 65 + lane];
        dqkv[(row0 + i) * 3 * H + h * 64 + lane] = from_f32<T>(o * 0.125f);
        __builtin_amdgcn_wave_barrier();
    }
    __threadfence_block();                               // dS rows of all waves visible to all waves below
    __syncthreads();
    // phase B1 (queries staged): dk_j = dS^T q / 8
    stage_rows65(st, qkv + row0 * 3 * H + h * 64, (size_t)3 * H, L, tid);
    __syncthreads();
    for (int j = wave; j < L; j += 4) {
        float dk = 0.f;
        for (int i = 0; i < L; i++) dk += dS[(size_t)i * L + j] * st[i * 65 + lane];
        dqkv[(row0 + j) * 3 * H + H + h * 64 + lane] = from_f32<T>(dk * 0.125f);
    }
    __syncthreads();
    // phase B2 (dctx staged): dv_j = P^T dctx
    stage_rows65(st, dctx + row0 * H + h * 64, (size_t)H, L, tid);
    __syncthreads();
    for (int j = wave; j < L; j += 4) {
        float dv = 0.f;
        for (int i = 0; i < L; i++) dv += P[(size_t)i * L + j] * st[i * 65 + lane];
        dqkv[(row0 + j) * 3 * H + 2 * H + h * 64 + lane] = from_f32<T>(dv);
    }
}

// ------------------------------------------------------------------------------------------
// Cross-attention over the image tokens with MFMA, whole rows in registers (no online rescale so
// the probabilities can be stashed exactly as med.py:280 saves them).
//   MODE 0 (forward):  S^T = K.q^T / 8 -> softmax -> stash P (fp32 [B,heads,L,Nst]) -> ctx = P.V
//   MODE 1 (backward): dP^T = V.dctx^T ; dS = P (dP - rowsum(dP P)) ; dq = dS.K / 8
//   MODE 2 (backward, target layer): dP^T = V.dctx^T -> store dP (fp32 [B,heads,L,Nst])
// a1 : "natural" operand  [B*N, ld1] (+ column offset applied by the host), row = image token
// a2t: transposed operand [64 rows per head..., ld2], element (h*64+d, b*Npad + n)
// x  : per-token operand  [B*L, ldx] (q or dctx), T
// out: [B*L, ldo] T (ctx or dq)
template <typename T, int MODE, int NW, int TPW, int QT>
__global__ __launch_bounds__(NW * 64) void xattn_kernel(const T* __restrict__ a1, int ld1, const T* __restrict__ a2t,
                                                        int ld2, int Npad, const T* __restrict__ x, int ldx,
                                                        T* __restrict__ out, int ldo, float* __restrict__ pbuf,
                                                        int Nst, int L, int N, int nheads) {
    __shared__ float red[NW][QT * 16];
    __shared__ float red2[NW][QT * 16];
    __shared__ __attribute__((aligned(16))) float obuf[64][QT * 16 + 1];
    const int h = blockIdx.x, b = blockIdx.y, l0 = blockIdx.z * (QT * 16);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int nkt = (N + 15) / 16;

    // per-token operand fragments for the two 16-query tiles of this block
    Frag<T> fx[QT][2];
    int lrow[QT];
#pragma unroll
    for (int qt = 0; qt < QT; qt++) {
        int l = l0 + qt * 16 + r;
        lrow[qt] = l;
        l = l < L ? l : L - 1;
        const T* xp = x + ((size_t)b * L + l) * ldx + h * 64;
        glb_frag(fx[qt][0], xp, 0, q);
        glb_frag(fx[qt][1], xp, 1, q);
    }
    // first product: rows = image tokens of this wave's tiles, cols = queries
    f32x4 s[TPW][QT];
#pragma unroll
    for (int i = 0; i < TPW; i++) {
        const int kt = wave * TPW + i;
#pragma unroll
        for (int qt = 0; qt < QT; qt++) s[i][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (kt < nkt) {
            int n = kt * 16 + r;
            n = n < N ? n : N - 1;
            const T* ap = a1 + ((size_t)b * N + n) * ld1 + h * 64;
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                Frag<T> fa;
                glb_frag(fa, ap, ks, q);
#pragma unroll
                for (int qt = 0; qt < QT; qt++) mma16(s[i][qt], fa, fx[qt][ks]);
            }
        }
    }
    float* prow[QT];
#pragma unroll
    for (int qt = 0; qt < QT; qt++) {
        const int l = lrow[qt] < L ? lrow[qt] : L - 1;
        prow[qt] = pbuf ? pbuf + (((size_t)b * nheads + h) * L + l) * Nst : nullptr;
    }

    if (MODE == 2) {
#pragma unroll
        for (int i = 0; i < TPW; i++) {
            const int kt = wave * TPW + i;
            if (kt >= nkt) continue;
#pragma unroll
            for (int qt = 0; qt < QT; qt++)
                if (lrow[qt] < L) *reinterpret_cast<f32x4*>(prow[qt] + kt * 16 + q * 4) = s[i][qt];
        }
        return;
    }

    if (MODE == 0) {
        // softmax over all N keys of each query column
        float mx[QT];
#pragma unroll
        for (int qt = 0; qt < QT; qt++) mx[qt] = -INFINITY;
#pragma unroll
        for (int i = 0; i < TPW; i++) {
            const int kt = wave * TPW + i;
#pragma unroll
            for (int qt = 0; qt < QT; qt++)
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int n = kt * 16 + q * 4 + e;
                    const float v = (kt < nkt && n < N) ? s[i][qt][e] * 0.125f : -INFINITY;
                    s[i][qt][e] = v;
                    mx[qt] = fmaxf(mx[qt], v);
                }
        }
#pragma unroll
        for (int qt = 0; qt < QT; qt++) {
            mx[qt] = fmaxf(mx[qt], __shfl_xor(mx[qt], 16, 64));
            mx[qt] = fmaxf(mx[qt], __shfl_xor(mx[qt], 32, 64));
            if (q == 0) red[wave][qt * 16 + r] = mx[qt];
        }
        __syncthreads();
        float sum[QT];
#pragma unroll
        for (int qt = 0; qt < QT; qt++) {
            float m = -INFINITY;
            sum[qt] = 0.f;
#pragma unroll
            for (int w = 0; w < NW; w++) m = fmaxf(m, red[w][qt * 16 + r]);
#pragma unroll
            for (int i = 0; i < TPW; i++)
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const float p = __expf(s[i][qt][e] - m);
                    s[i][qt][e] = p;
                    sum[qt] += p;
                }
            sum[qt] += __shfl_xor(sum[qt], 16, 64);
            sum[qt] += __shfl_xor(sum[qt], 32, 64);
            if (q == 0) red2[wave][qt * 16 + r] = sum[qt];
        }
        __syncthreads();
#pragma unroll
        for (int qt = 0; qt < QT; qt++) {
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < NW; w++) tot += red2[w][qt * 16 + r];
            const float inv = 1.0f / tot;
#pragma unroll
            for (int i = 0; i < TPW; i++) {
                s[i][qt] *= inv;
                const int kt = wave * TPW + i;
                if (pbuf && kt < nkt && lrow[qt] < L)
                    *reinterpret_cast<f32x4*>(prow[qt] + kt * 16 + q * 4) = s[i][qt];
            }
        }
    } else {
        // MODE 1: dS = P * (dP - rowsum(dP * P))
        float dot[QT];
#pragma unroll
        for (int qt = 0; qt < QT; qt++) dot[qt] = 0.f;
        f32x4 pv[TPW][QT];
#pragma unroll
        for (int i = 0; i < TPW; i++) {
            const int kt = wave * TPW + i;
#pragma unroll
            for (int qt = 0; qt < QT; qt++) {
                pv[i][qt] = (kt < nkt) ? *reinterpret_cast<const f32x4*>(prow[qt] + kt * 16 + q * 4)
                                       : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int n = kt * 16 + q * 4 + e;
                    if (!(kt < nkt && n < N)) pv[i][qt][e] = 0.f;
                    dot[qt] += s[i][qt][e] * pv[i][qt][e];
                }
            }
        }
#pragma unroll
        for (int qt = 0; qt < QT; qt++) {
            dot[qt] += __shfl_xor(dot[qt], 16, 64);
            dot[qt] += __shfl_xor(dot[qt], 32, 64);
            if (q == 0) red[wave][qt * 16 + r] = dot[qt];
        }
        __syncthreads();
#pragma unroll
        for (int qt = 0; qt < QT; qt++) {
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < NW; w++) tot += red[w][qt * 16 + r];
#pragma unroll
            for (int i = 0; i < TPW; i++)
#pragma unroll
                for (int e = 0; e < 4; e++) s[i][qt][e] = pv[i][qt][e] * (s[i][qt][e] - tot) * 0.125f;
        }
    }

    // second product: out^T[d][l] = sum_n a2t[d][n] * s^T[n][l], k-steps pair this wave's tiles
    f32x4 o[4][QT];
#pragma unroll
    for (int dt = 0; dt < 4; dt++) {
#pragma unroll
        for (int qt = 0; qt < QT; qt++) o[dt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < TPW; i += 2) {
        const int ktA = wave * TPW + i, ktB = ktA + 1;
        if (ktA >= nkt) continue;
        const bool hasB = (i + 1 < TPW) && (ktB < nkt);
        Frag<T> fp[QT];
#pragma unroll
        for (int qt = 0; qt < QT; qt++) {
            if (i + 1 < TPW) pack_p(fp[qt], s[i][qt], hasB ? s[i + 1][qt] : f32x4{0.f, 0.f, 0.f, 0.f});
            else pack_p(fp[qt], s[i][qt], f32x4{0.f, 0.f, 0.f, 0.f});
        }
#pragma unroll
        for (int dt = 0; dt < 4; dt++) {
            const T* vp = a2t + (size_t)(h * 64 + dt * 16 + r) * ld2 + (size_t)b * Npad;
            Frag<T> fv;
            glb_frag_pair(fv, vp + ktA * 16 + q * 4, vp + (hasB ? ktB : ktA) * 16 + q * 4);
#pragma unroll
            for (int qt = 0; qt < QT; qt++) mma16(o[dt][qt], fv, fp[qt]);
        }
    }
    // deterministic cross-wave reduction through LDS
    for (int w = 0; w < NW; w++) {
        if (wave == w) {
#pragma unroll
            for (int dt = 0; dt < 4; dt++)
#pragma unroll
                for (int qt = 0; qt < QT; qt++)
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        float* p = &obuf[dt * 16 + q * 4 + e][qt * 16 + r];
                        *p = (w == 0 ? 0.f : *p) + o[dt][qt][e];
                    }
        }
        __syncthreads();
    }
    for (int i = tid; i < QT * 16 * 64; i += NW * 64) {
        const int l = i >> 6, d = i & 63;
        if (l0 + l < L) out[((size_t)b * L + l0 + l) * ldo + h * 64 + d] = from_f32<T>(obuf[d][l]);
    }
}

// ------------------------------------------------------------------------------------------
// LayerNorm backward (weights are not differentiated): dx = rstd (g - mean(g) - xhat mean(g xhat)),
// g = dy * w.  One wave per row; writes fp32 and optional T copy.
template <typename T>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                            const float* __restrict__ xhat,
                                                            const float* __restrict__ rstd, int rows, int D,
                                                            float* __restrict__ dx, T* __restrict__ dxt) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int nv = D >> 2;
    f32x4 gv[4], hv[4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int c = lane + i * 64;
        if (c < nv) {
            const f32x4 d = reinterpret_cast<const f32x4*>(dy + (size_t)row * D)[c];
            const f32x4 ww = reinterpret_cast<const f32x4*>(w)[c];
            hv[i] = reinterpret_cast<const f32x4*>(xhat + (size_t)row * D)[c];
            gv[i] = d * ww;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                s1 += gv[i][e];
                s2 += gv[i][e] * hv[i][e];
            }
        }
    }
    const float m1 = wave_sum(s1) / (float)D, m2 = wave_sum(s2) / (float)D;
    const float rs = rstd[row];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int c = lane + i * 64;
        if (c >= nv) continue;
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; e++) o[e] = rs * (gv[i][e] - m1 - hv[i][e] * m2);
        if (dx) reinterpret_cast<f32x4*>(dx + (size_t)row * D)[c] = o;
        if (dxt) {
            T* p = dxt + (size_t)row * D + c * 4;
#pragma unroll
            for (int e = 0; e < 4; e++) p[e] = from_f32<T>(o[e]);
        }
    }
}

// logits[b, c] = h[b, 0, :] . W[c, :] + bias[c]  (itm_head, blip_image_text_matching.py:248)
__global__ void itm_head_kernel(const float* __restrict__ hlast, const float* __restrict__ w,
                                const float* __restrict__ bias, float* __restrict__ logits, int L, int H) {
    const int b = blockIdx.x, c = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float acc = 0.f;
    for (int d = lane; d < H; d += 64) acc += hlast[(size_t)b * L * H + d] * w[c * H + d];
    acc = wave_sum(acc);
    if (lane == 0) logits[b * 2 + c] = acc + bias[c];
}

// seed of the backward: d(sum_b logits[b,1]) / d h_last = itm_head.weight[1] at token 0, else 0
__global__ void itm_grad_seed_kernel(const float* __restrict__ w, float* __restrict__ dh, int B, int L, int H) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * L * H) return;
    const int d = idx % H, l = (idx / H) % L;
    dh[idx] = l == 0 ? w[H + d] : 0.f;
}

template <typename T>
__global__ void cast_kernel(const float* __restrict__ in, T* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = from_f32<T>(in[i]);
}

__global__ void split_kernel(const float* __restrict__ in, bf16* __restrict__ hi, bf16* __restrict__ lo, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float v = in[i];
        const bf16 h = (bf16)v;
        hi[i] = h;
        lo[i] = (bf16)(v - (float)h);
    }
}

// ------------------------------------------------------------------------------------------ host
static inline int ok() { return hipGetLastError() == hipSuccess ? PNP_OK : PNP_ERR_HIP; }

int text_embed(const int64_t* ids, int ld_ids, const float* word, const float* pos, float* out, int B, int L, int H,
               int enc_id, int vocab, hipStream_t s) {
    const int total = B * L * (H >> 2);
    hipLaunchKernelGGL(text_embed_kernel, dim3((total + 255) / 256), dim3(256), 0, s, ids, ld_ids, word, pos, out, B, L,
                       H, enc_id, vocab);
    return ok();
}

static size_t self_attn_smem(int L, int nw = 4) { return (size_t)(2 * L * 65 + (nw + 1) * TXT_MAX_L + nw * 64) * sizeof(float); }
// workgroups per (head, image) for the row-split launches: the nz in [1, 4] that minimises rounds(pairs * nz) / nz, a round being
// what is RESIDENT at once: CUs x workgroups per CU at this caption length's LDS use (about 100 KB at L = 155: one per CU, 96 pairs
// -> nz = 2, one round of half the rows; 49 KB at L = 85: three per CU, 420 pairs -> nz = 3, two rounds of a third)
static int row_split(int pairs, size_t smem) {
    const int cus = device_cu_count() > 0 ? device_cu_count() : 256;
    int per_cu = (int)(160 * 1024 / (smem ? smem : 1));
    per_cu = per_cu < 1 ? 1 : per_cu > 8 ? 8 : per_cu;
    const int slots = cus * per_cu;
#ifdef PNP_DEV
    if (getenv("PNP_TXT_NZ")) return atoi(getenv("PNP_TXT_NZ"));
#endif
    int best = 1;
    double cost = 1e30;
    for (int nz = 1; nz <= 4; nz++) {
        const double c = (double)((pairs * nz + slots - 1) / slots) / nz;
        if (c < cost - 1e-9) {
            cost = c;
            best = nz;
        }
    }
    return best;
}
static size_t self_attn_long_smem(int L) { return (size_t)(L * 65 + 5 * TXT_LONG_L + 4 * 64) * sizeof(float); }

// one-time per device: the kernels' opt-in to more than 64 KB of dynamic LDS (common.h: lds_opt_in)
#define PNP_OPT_IN(kernel, bytes)                                                               \
    do {                                                                                        \
        static std::atomic<uint32_t> opted{0};                                                  \
        if (lds_opt_in(opted, reinterpret_cast<const void*>(kernel), (int)(bytes)) != PNP_OK) return PNP_ERR_HIP; \
    } while (0)

int text_self_attn(int bf, const void* qkv, const int64_t* mask, int ld_mask, void* ctx, float* probs, float* scratch, int B, int L,
                   int H, hipStream_t s) {
    if (L > TXT_LONG_L || H % 64) return PNP_ERR_ARG;
    dim3 grid(H / 64, B);
    if (L > TXT_MAX_L) {
        // long captions: the probabilities of a row pass through global memory (the stash, else `scratch`: [B, heads, L, L])
        float* pg = probs ? probs : scratch;
        if (!pg) return PNP_ERR_ARG;
        const size_t smem = self_attn_long_smem(L);
        if (bf) {
            PNP_OPT_IN(text_self_attn_long_kernel<bf16>, self_attn_long_smem(TXT_LONG_L));
            hipLaunchKernelGGL((text_self_attn_long_kernel<bf16>), grid, dim3(256), smem, s, (const bf16*)qkv, mask, ld_mask, (bf16*)ctx, pg, L, H);
        } else {
            PNP_OPT_IN(text_self_attn_long_kernel<float>, self_attn_long_smem(TXT_LONG_L));
            hipLaunchKernelGGL((text_self_attn_long_kernel<float>), grid, dim3(256), smem, s, (const float*)qkv, mask, ld_mask, (float*)ctx, pg, L, H);
        }
        return ok();
    }
    const size_t smem = self_attn_smem(L);
    if (L > 64) grid.z = row_split((int)grid.x * (int)grid.y, smem);   // few (head, image) pairs: query rows over up to 4 workgroups each
    if (bf) {
        PNP_OPT_IN(text_self_attn_kernel<bf16>, self_attn_smem(TXT_MAX_L));
        hipLaunchKernelGGL((text_self_attn_kernel<bf16>), grid, dim3(256), smem, s, (const bf16*)qkv, mask, ld_mask,
                           (bf16*)ctx, probs, L, H);
    } else if (L <= 64) {
        // short captions (one wave per query row, L / waves rows each): 8 waves per (image, head)
        PNP_OPT_IN((text_self_attn_kernel<float, 8>), self_attn_smem(64, 8));
        hipLaunchKernelGGL((text_self_attn_kernel<float, 8>), grid, dim3(512), self_attn_smem(L, 8), s, (const float*)qkv, mask, ld_mask,
                           (float*)ctx, probs, L, H);
    } else {
        PNP_OPT_IN(text_self_attn_kernel<float>, self_attn_smem(TXT_MAX_L));
        hipLaunchKernelGGL((text_self_attn_kernel<float>), grid, dim3(256), smem, s, (const float*)qkv, mask, ld_mask,
                           (float*)ctx, probs, L, H);
    }
    return ok();
}

int text_self_attn_bwd(int bf, const void* qkv, const float* dctx, const float* probs, float* ds_scratch, void* dqkv,
                       int B, int L, int H, hipStream_t s) {
    if (L > TXT_LONG_L || H % 64) return PNP_ERR_ARG;
    dim3 grid(H / 64, B);
    if (L > TXT_MAX_L) {
        const size_t smem = self_attn_long_smem(L);
        if (bf) {
            PNP_OPT_IN(text_self_attn_bwd_long_kernel<bf16>, self_attn_long_smem(TXT_LONG_L));
            hipLaunchKernelGGL((text_self_attn_bwd_long_kernel<bf16>), grid, dim3(256), smem, s, (const bf16*)qkv, dctx, probs, ds_scratch, (bf16*)dqkv, L, H);
        } else {
            PNP_OPT_IN(text_self_attn_bwd_long_kernel<float>, self_attn_long_smem(TXT_LONG_L));
            hipLaunchKernelGGL((text_self_attn_bwd_long_kernel<float>), grid, dim3(256), smem, s, (const float*)qkv, dctx, probs, ds_scratch, (float*)dqkv, L, H);
        }
        return ok();
    }
    const size_t smem = self_attn_smem(L);
    if (!bf && L > 64 && row_split((int)grid.x * (int)grid.y, smem) > 1) {
        // too few (head, image) pairs for the chip: phase A and phase B as two launches, rows dealt over up to 4 workgroups each
        grid.z = row_split((int)grid.x * (int)grid.y, smem);
        PNP_OPT_IN((text_self_attn_bwd_kernel<float, 1>), self_attn_smem(TXT_MAX_L));
        hipLaunchKernelGGL((text_self_attn_bwd_kernel<float, 1>), grid, dim3(256), smem, s, (const float*)qkv, dctx, probs,
                           ds_scratch, (float*)dqkv, L, H);
        PNP_OPT_IN((text_self_attn_bwd_kernel<float, 2>), self_attn_smem(TXT_MAX_L));
        hipLaunchKernelGGL((text_self_attn_bwd_kernel<float, 2>), grid, dim3(256), smem, s, (const float*)qkv, dctx, probs,
                           ds_scratch, (float*)dqkv, L, H);
        return ok();
    }
    if (bf) {
        PNP_OPT_IN(text_self_attn_bwd_kernel<bf16>, self_attn_smem(TXT_MAX_L));
        hipLaunchKernelGGL((text_self_attn_bwd_kernel<bf16>), grid, dim3(256), smem, s, (const bf16*)qkv, dctx, probs,
                           ds_scratch, (bf16*)dqkv, L, H);
    } else {
        PNP_OPT_IN(text_self_attn_bwd_kernel<float>, self_attn_smem(TXT_MAX_L));
        hipLaunchKernelGGL((text_self_attn_bwd_kernel<float>), grid, dim3(256), smem, s, (const float*)qkv, dctx, probs,
                           ds_scratch, (float*)dqkv, L, H);
    }
    return ok();
}
#undef PNP_OPT_IN

template <typename T, int MODE>
static int xattn_launch(const void* a1, int ld1, const void* a2t, int ld2, int Npad, const void* x, int ldx, void* out,
                        int ldo, float* pbuf, int Nst, int B, int L, int N, int nheads, hipStream_t s) {
    const int nkt = (N + 15) / 16;
    if (nkt <= 4 * 8) {
        dim3 grid(nheads, B, (L + 31) / 32);
        // (8 waves of 4 key tiles run 38.7 -> 35.1 us per launch -- twice the K / V rows requested at a time -- but sum the softmax
        // denominators in another order, and the normalised-map error against the reference's drop-loop golden moves from 3.0e-4
        // to 7.3e-4, past its 2x-measured bound: not taken.  16 waves of 2 tiles: 48 us.)
        hipLaunchKernelGGL((xattn_kernel<T, MODE, 4, 8, 2>), grid, dim3(256), 0, s, (const T*)a1, ld1, (const T*)a2t, ld2,
                           Npad, (const T*)x, ldx, (T*)out, ldo, pbuf, Nst, L, N, nheads);
    } else if (nkt <= 16 * 10) {
        dim3 grid(nheads, B, (L + 15) / 16);
        hipLaunchKernelGGL((xattn_kernel<T, MODE, 16, 10, 1>), grid, dim3(1024), 0, s, (const T*)a1, ld1, (const T*)a2t,
                           ld2, Npad, (const T*)x, ldx, (T*)out, ldo, pbuf, Nst, L, N, nheads);
    } else {
        return PNP_ERR_ARG;
    }
    return ok();
}

int xattn(int bf, int mode, const void* a1, int ld1, const void* a2t, int ld2, int Npad, const void* x, int ldx,
          void* out, int ldo, float* pbuf, int Nst, int B, int L, int N, int nheads, hipStream_t s) {
    if (Nst % 4 || Nst < ((N + 15) / 16) * 16) return PNP_ERR_ARG;
#define PNP_X(TT, MM) return xattn_launch<TT, MM>(a1, ld1, a2t, ld2, Npad, x, ldx, out, ldo, pbuf, Nst, B, L, N, nheads, s)
    if (bf) {
        if (mode == 0) PNP_X(bf16, 0);
        if (mode == 1) PNP_X(bf16, 1);
        if (mode == 2) PNP_X(bf16, 2);
    } else {
        if (mode == 0) PNP_X(float, 0);
        if (mode == 1) PNP_X(float, 1);
        if (mode == 2) PNP_X(float, 2);
    }
#undef PNP_X
    return PNP_ERR_ARG;
}

int layernorm_bwd(int bf, const float* dy, const float* w, const float* xhat, const float* rstd, int rows, int D,
                  float* dx, void* dxt, hipStream_t s) {
    if (D > 1024 || D % 4) return PNP_ERR_ARG;
    const int nb = (rows + 3) / 4;
    if (bf) hipLaunchKernelGGL((layernorm_bwd_kernel<bf16>), dim3(nb), dim3(256), 0, s, dy, w, xhat, rstd, rows, D, dx, (bf16*)dxt);
    else hipLaunchKernelGGL((layernorm_bwd_kernel<float>), dim3(nb), dim3(256), 0, s, dy, w, xhat, rstd, rows, D, dx, (float*)dxt);
    return ok();
}

int itm_head(const float* hlast, const float* w, const float* bias, float* logits, int B, int L, int H, hipStream_t s) {
    hipLaunchKernelGGL(itm_head_kernel, dim3(B), dim3(128), 0, s, hlast, w, bias, logits, L, H);
    return ok();
}

int itm_grad_seed(const float* w, float* dh, int B, int L, int H, hipStream_t s) {
    const int total = B * L * H;
    hipLaunchKernelGGL(itm_grad_seed_kernel, dim3((total + 255) / 256), dim3(256), 0, s, w, dh, B, L, H);
    return ok();
}

int cast_f32(int bf, const float* in, void* out, size_t n, hipStream_t s) {
    const unsigned nb = (unsigned)((n + 255) / 256);
    if (bf) hipLaunchKernelGGL((cast_kernel<bf16>), dim3(nb), dim3(256), 0, s, in, (bf16*)out, n);
    else hipLaunchKernelGGL((cast_kernel<float>), dim3(nb), dim3(256), 0, s, in, (float*)out, n);
    return ok();
}

int split_f32(const float* in, void* hi, void* lo, size_t n, hipStream_t s) {
    const unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(split_kernel, dim3(nb), dim3(256), 0, s, in, (bf16*)hi, (bf16*)lo, n);
    return ok();
}

}  // namespace pnp
