// NT GEMM for gfx950:  C[M,N] = A[M,K] * B[N,K]^T  (+ fused epilogue), A and B both K-contiguous.
//
// Every dense Linear of the path goes through this kernel (reference call sites: vit.py:93-117
// qkv/proj, vit.py:45-51 fc1/fc2, timm PatchEmbed conv16x16/16 as a GEMM, med.py:201-228 q/k/v,
// med.py:321-325 / :393-411 dense layers; the analytic backward re-uses it with pre-transposed
// weights).  T = bf16 (v_mfma_f32_16x16x32_bf16, fp32 accumulate) or float
// (v_mfma_f32_16x16x4_f32: exact fp32 fma chain, the parity mode).
//
// Tiling: BM x BN block tile, 256 threads = 4 waves (2x2), 128-byte k-slab per stage
// (64 bf16 / 32 f32), LDS double-buffered and XOR-swizzled per 16-byte chunk so the MFMA operand
// reads (16 consecutive rows, same logical chunk) are conflict-free ds_read_b128.
// Operands are swapped at the MFMA (D = Btile * Atile^T) so each lane owns 4 CONSECUTIVE output
// columns of one row: the epilogue does one 8/16-byte store per fragment and vector bias loads.
#include "common.h"
#include "gemm.h"

namespace pnp {

template <typename T, int BM, int BN>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const GemmArgs g) {
    constexpr int ROWB = 128;                       // bytes of k per LDS row per stage
    constexpr int BK = ROWB / Elem<T>::kBytes;      // 64 bf16 / 32 f32
    constexpr int WTM = BM / 2, WTN = BN / 2;       // wave tile
    constexpr int TM = WTM / 16, TN = WTN / 16;     // 16x16 tiles per wave
    constexpr int A_CHUNKS = BM * 8 / 256;          // 16-byte chunks per thread per stage
    constexpr int B_CHUNKS = BN * 8 / 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STAGE = (BM + BN) * ROWB;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware tile order: blocks b and b+8 share an XCD/L2 (round-robin dispatch), so give each
    // XCD a contiguous run of tiles that walk M fastest (they re-use one B panel out of L2).
    const int nbm = (g.M + BM - 1) / BM, nbn = g.N / BN;
    const int nwg = nbm * nbn;
    int bid = blockIdx.x;
    {
        const int qd = nwg >> 3, rm = nwg & 7, x = bid & 7, i = bid >> 3;
        bid = (x < rm ? x * (qd + 1) : rm * (qd + 1) + (x - rm) * qd) + i;
    }
    const int bm = bid % nbm, bn = bid / nbm;
    const int m0 = bm * BM, n0 = bn * BN;

    const char* Ab = reinterpret_cast<const char*>(g.A);
    const char* Bb = reinterpret_cast<const char*>(g.B);
    const size_t lda_b = (size_t)g.lda * Elem<T>::kBytes, ldb_b = (size_t)g.ldb * Elem<T>::kBytes;

    // per-thread staging assignment: chunk ci -> (row = ci / 8, chunk = ci % 8)
    const char* a_src[A_CHUNKS];
    int a_dst[A_CHUNKS];
#pragma unroll
    for (int i = 0; i < A_CHUNKS; i++) {
        const int ci = tid + i * 256, row = ci >> 3, c = ci & 7;
        int gr = m0 + row;
        gr = gr < g.M ? gr : g.M - 1;
        a_src[i] = Ab + (size_t)gr * lda_b + c * 16;
        a_dst[i] = lds_off<ROWB>(row, c);
    }
    const char* b_src[B_CHUNKS];
    int b_dst[B_CHUNKS];
#pragma unroll
    for (int i = 0; i < B_CHUNKS; i++) {
        const int ci = tid + i * 256, row = ci >> 3, c = ci & 7;
        int gr = n0 + row;
        gr = gr < g.Nvalid ? gr : g.Nvalid - 1;
        b_src[i] = Bb + (size_t)gr * ldb_b + c * 16;
        b_dst[i] = lds_off<ROWB>(row, c);
    }

    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; i++)
#pragma unroll
        for (int j = 0; j < TM; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    chunk16 ra[A_CHUNKS], rb[B_CHUNKS];
    const int nk = g.K / BK;
#pragma unroll
    for (int i = 0; i < A_CHUNKS; i++) ra[i] = *reinterpret_cast<const chunk16*>(a_src[i]);
#pragma unroll
    for (int i = 0; i < B_CHUNKS; i++) rb[i] = *reinterpret_cast<const chunk16*>(b_src[i]);
#pragma unroll
    for (int i = 0; i < A_CHUNKS; i++) *reinterpret_cast<chunk16*>(smem + a_dst[i]) = ra[i];
#pragma unroll
    for (int i = 0; i < B_CHUNKS; i++) *reinterpret_cast<chunk16*>(smem + BM * ROWB + b_dst[i]) = rb[i];
    __syncthreads();

    for (int kt = 0; kt < nk; kt++) {
        const int cur = kt & 1;
        const char* cA = smem + cur * STAGE;
        const char* cB = cA + BM * ROWB;
        char* nA = smem + (cur ^ 1) * STAGE;
        char* nB = nA + BM * ROWB;
        if (kt + 1 < nk) {
            const size_t koff = (size_t)(kt + 1) * ROWB;
#pragma unroll
            for (int i = 0; i < A_CHUNKS; i++) ra[i] = *reinterpret_cast<const chunk16*>(a_src[i] + koff);
#pragma unroll
            for (int i = 0; i < B_CHUNKS; i++) rb[i] = *reinterpret_cast<const chunk16*>(b_src[i] + koff);
        }
        constexpr int KSTEPS = BK / 32;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ks++) {
            Frag<T> fa[TM], fb[TN];
#pragma unroll
            for (int j = 0; j < TM; j++) lds_frag<ROWB>(fa[j], cA, wm * WTM + j * 16 + r, ks, q);
#pragma unroll
            for (int i = 0; i < TN; i++) lds_frag<ROWB>(fb[i], cB, wn * WTN + i * 16 + r, ks, q);
#pragma unroll
            for (int i = 0; i < TN; i++)
#pragma unroll
                for (int j = 0; j < TM; j++) mma16(acc[i][j], fb[i], fa[j]);
        }
        if (kt + 1 < nk) {
#pragma unroll
            for (int i = 0; i < A_CHUNKS; i++) *reinterpret_cast<chunk16*>(nA + a_dst[i]) = ra[i];
#pragma unroll
            for (int i = 0; i < B_CHUNKS; i++) *reinterpret_cast<chunk16*>(nB + b_dst[i]) = rb[i];
        }
        __syncthreads();
    }

    // ---- epilogue: lane owns rows m = .. + r, columns n = .. + 4q .. 4q+3 of each 16x16 tile
#pragma unroll
    for (int j = 0; j < TM; j++) {
        const int m = m0 + wm * WTM + j * 16 + r;
        if (m >= g.M) continue;
        int orow = m;
        const float* resid_row = nullptr;
        if (g.row_div > 0) {                         // patch-embed rows -> token rows (skip cls)
            const int b = m / g.row_div, p = m - b * g.row_div;
            orow = b * (g.row_div + 1) + 1 + p;
            if (g.resid) resid_row = g.resid + (size_t)(1 + p) * g.ldr;   // pos_embed[1+p]
        } else if (g.resid) {
            resid_row = g.resid + (size_t)m * g.ldr;
        }
        const float bias_row = (g.bias && g.bias_on_rows) ? g.bias[m] : 0.f;
#pragma unroll
        for (int i = 0; i < TN; i++) {
            const int n = n0 + wn * WTN + i * 16 + q * 4;
            if (n >= g.Nvalid) continue;
            f32x4 v = acc[i][j];
            if (g.bias) {
                if (g.bias_on_rows) {
                    v += bias_row;
                } else {
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(g.bias + n);
                    v += bv;
                }
            }
            if (g.mode == GEMM_EPI_GELU) {
                if (g.aux) *reinterpret_cast<f32x4*>(g.aux + (size_t)orow * g.ld_aux + n) = v;
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = gelu_erf(v[e]);
            } else if (g.mode == GEMM_EPI_GELU_GRAD) {
                const f32x4 u = *reinterpret_cast<const f32x4*>(g.aux + (size_t)orow * g.ld_aux + n);
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] *= gelu_erf_grad(u[e]);
            }
            if (resid_row) {
                const f32x4 rv = *reinterpret_cast<const f32x4*>(resid_row + n);
                v += rv;
            }
            // column remap (token index -> per-image padded index) for transposed outputs
            size_t ocol = n;
            bool contiguous = true;
            if (g.col_div > 0) {
                const int b = n / g.col_div, t = n - b * g.col_div;
                ocol = (size_t)b * g.col_pad + t;
                contiguous = (t + 3 < g.col_div) && (n + 3 < g.Nvalid);
            } else {
                contiguous = (n + 3 < g.Nvalid);
            }
            if (contiguous) {
                if (g.out_f32) *reinterpret_cast<f32x4*>(g.out_f32 + (size_t)orow * g.ldo + ocol) = v;
                if (g.out_t) {
                    T* o = reinterpret_cast<T*>(g.out_t) + (size_t)orow * g.ldo_t + ocol;
                    if constexpr (sizeof(T) == 2) {
                        bf16x4 pk = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                        *reinterpret_cast<bf16x4*>(o) = pk;
                    } else {
                        *reinterpret_cast<f32x4*>(o) = v;
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int ne = n + e;
                    if (ne >= g.Nvalid) break;
                    size_t oc = ne;
                    if (g.col_div > 0) {
                        const int b = ne / g.col_div, t = ne - b * g.col_div;
                        oc = (size_t)b * g.col_pad + t;
                    }
                    if (g.out_f32) g.out_f32[(size_t)orow * g.ldo + oc] = v[e];
                    if (g.out_t) reinterpret_cast<T*>(g.out_t)[(size_t)orow * g.ldo_t + oc] = from_f32<T>(v[e]);
                }
            }
        }
    }
}

template <typename T, int BM, int BN>
static int launch_cfg(const GemmArgs& g, hipStream_t s) {
    const int nbm = (g.M + BM - 1) / BM, nbn = g.N / BN;
    const size_t smem = 2 * (BM + BN) * 128;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_kernel<T, BM, BN>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return PNP_ERR_HIP;
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_nt_kernel<T, BM, BN>), dim3(nbm * nbn), dim3(256), smem, s, g);
    return hipGetLastError() == hipSuccess ? PNP_OK : PNP_ERR_HIP;
}

// Host entry.  N is rounded up to the tile internally (loads clamp, stores mask on Nvalid).
int gemm_nt(int dtype_bf16, GemmArgs g, hipStream_t s) {
    if (g.M <= 0 || g.N <= 0 || g.K <= 0) return PNP_ERR_ARG;
    const int bk = dtype_bf16 ? 64 : 32;
    if (g.K % bk) return PNP_ERR_ARG;
    if ((g.lda * (dtype_bf16 ? 2 : 4)) % 16 || (g.ldb * (dtype_bf16 ? 2 : 4)) % 16) return PNP_ERR_ARG;
    g.Nvalid = g.N;
    // small problems (text side: M = B*L rows) use 64x64 tiles to fill more CUs
    const long tiles128 = (long)((g.M + 127) / 128) * ((g.N + 127) / 128);
    const bool small = tiles128 < 192;
    if (small) {
        g.N = (g.N + 63) / 64 * 64;
        return dtype_bf16 ? launch_cfg<bf16, 64, 64>(g, s) : launch_cfg<float, 64, 64>(g, s);
    }
    GemmProfile& pf = gemm_profile();
    const bool timed = pf.on && pf.used < GemmProfile::kMax;
    if (timed) {
        while (pf.created <= pf.used) {
            if (hipEventCreate(&pf.ev0[pf.created]) != hipSuccess || hipEventCreate(&pf.ev1[pf.created]) != hipSuccess)
                return PNP_ERR_HIP;
            pf.created++;
        }
        (void)hipEventRecord(pf.ev0[pf.used], s);
    }
    const double fl = 2.0 * g.M * (double)g.N * g.K;
    g.N = (g.N + 127) / 128 * 128;
    const int r = dtype_bf16 ? launch_cfg<bf16, 128, 128>(g, s) : launch_cfg<float, 128, 128>(g, s);
    if (timed) {
        (void)hipEventRecord(pf.ev1[pf.used], s);
        pf.used++;
        pf.launches++;
        pf.flops += fl;
    }
    return r;
}

GemmProfile& gemm_profile() {
    static GemmProfile p;
    return p;
}

}  // namespace pnp
