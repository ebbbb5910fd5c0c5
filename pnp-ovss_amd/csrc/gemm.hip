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
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "gemm.h"
#include "gemm_wide.h"

namespace pnp {

// ---- shared epilogue: one fragment = row m, 4 consecutive columns n..n+3
struct RowCtx {
    int orow;
    const float* resid_row;
    float bias_row;
};

__device__ __forceinline__ RowCtx row_ctx(const GemmArgs& g, int m) {
    RowCtx rc;
    rc.orow = m;
    rc.resid_row = nullptr;
    if (g.row_div > 0) {                         // patch-embed rows -> token rows (skip cls)
        const int b = m / g.row_div, p = m - b * g.row_div;
        rc.orow = b * (g.row_div + 1) + 1 + p;
        if (g.resid) rc.resid_row = g.resid + (size_t)(1 + p) * g.ldr;   // pos_embed[1+p]
    } else if (g.resid) {
        rc.resid_row = g.resid + (size_t)m * g.ldr;
    }
    rc.bias_row = (g.bias && g.bias_on_rows) ? g.bias[m] : 0.f;
    return rc;
}

// preloaded: bias (and, in LINEAR mode, the residual) were already folded into the accumulator's
// initial value, so the epilogue issues no loads.
template <typename T>
__device__ __forceinline__ void store_frag(const GemmArgs& g, const RowCtx& rc, f32x4 v, int m, int n, bool preloaded = false) {
    const int orow = rc.orow;
    if (g.bias && !preloaded) {
        if (g.bias_on_rows) {
            v += rc.bias_row;
        } else {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(g.bias + n);
            v += bv;
        }
    }
    if (g.mode == GEMM_EPI_GELU) {
        if (g.aux) *reinterpret_cast<f32x4*>(g.aux + (size_t)orow * g.ld_aux + n) = v;
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] = sizeof(T) == 2 ? gelu_erf_fast(v[e]) : gelu_erf(v[e]);
    } else if (g.mode == GEMM_EPI_GELU_GRAD) {
        const f32x4 u = *reinterpret_cast<const f32x4*>(g.aux + (size_t)orow * g.ld_aux + n);
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] *= gelu_erf_grad(u[e]);
    }
    if (rc.resid_row && !(preloaded && g.mode == GEMM_EPI_LINEAR)) {
        const f32x4 rv = *reinterpret_cast<const f32x4*>(rc.resid_row + n);
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

// ------------------------------------------------------------------------------------------
// Large-problem kernel: 256 x 128 block tile, 8 waves (4 x 2, 64 x 64 each), 128-byte k-slab,
// operands streamed global -> LDS by `global_load_lds_dwordx4` (no VGPR staging) into a 3-deep
// ring so two k-slabs are always in flight; one raw s_barrier per k-slab with a COUNTED vmcnt
// (never 0 in the main loop).  The LDS image is lane-linear per DMA (8 rows x 128 B), so the XOR
// swizzle is applied to the per-lane SOURCE address and again on the fragment read.

// BM x BN block tile, (BM/WTM) x (BN/WTN) waves of WTM x WTN, NS-slot LDS ring.
template <typename T, int BM, int BN, int WTM, int WTN, int NS, bool PIPE = true>
__global__ __launch_bounds__((BM / WTM) * (BN / WTN) * 64) void gemm_nt_big_kernel(const GemmArgs g) {
    constexpr int ROWB = 128;
    constexpr int BK = ROWB / Elem<T>::kBytes;
    constexpr int STAGE = (BM + BN) * ROWB;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr int WN = BN / WTN, NWAVES = (BM / WTM) * WN;
    constexpr int A_DMA = BM / 8 / NWAVES, B_DMA = BN / 8 / NWAVES, NDMA = A_DMA + B_DMA;   // per wave per slab
    static_assert(BM % (8 * NWAVES) == 0 && BN % (8 * NWAVES) == 0, "tile rows must split evenly over the waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int wm = wave / WN, wn = wave % WN;

    const int nbm = (g.M + BM - 1) / BM, nbn = g.N / BN;
    int bm, bn;
    tile_coords<(BM >= 256 ? 4 : 8)>(blockIdx.x, nbm, nbn, bm, bn);
    const int m0 = bm * BM, n0 = bn * BN;

    // DMA assignment: wave w moves A rows [8*A_DMA*w, ..) and B rows [8*B_DMA*w, ..), 8 rows per DMA.
    // Per-lane sources are kept as 32-bit byte offsets from the (uniform) operand bases.
    const char* Ab = reinterpret_cast<const char*>(g.A);
    const char* Bb = reinterpret_cast<const char*>(g.B);
    uint32_t soff[NDMA];
    {
        const uint32_t lda_b = (uint32_t)g.lda * Elem<T>::kBytes, ldb_b = (uint32_t)g.ldb * Elem<T>::kBytes;
        const int pc = lane & 7;
#pragma unroll
        for (int i = 0; i < A_DMA; i++) {
            const int row = (wave * A_DMA + i) * 8 + (lane >> 3);
            int gr = m0 + row;
            gr = gr < g.M ? gr : g.M - 1;
            soff[i] = (uint32_t)gr * lda_b + swz_chunk<ROWB>(row, pc) * 16;
        }
#pragma unroll
        for (int i = 0; i < B_DMA; i++) {
            const int row = (wave * B_DMA + i) * 8 + (lane >> 3);
            int gr = n0 + row;
            gr = gr < g.Nvalid ? gr : g.Nvalid - 1;
            soff[A_DMA + i] = (uint32_t)gr * ldb_b + swz_chunk<ROWB>(row, pc) * 16;
        }
    }
    auto issue = [&](int kt) {
#ifdef PNP_DEV
        if (g.ablate == 1 && kt >= NS) return;           // timing ablation: no steady-state DMA
#endif
        char* stage = smem + (kt % NS) * STAGE;
        const uint32_t koff = (uint32_t)kt * ROWB;
#pragma unroll
        for (int i = 0; i < NDMA; i++) {
            const char* base = i < A_DMA ? Ab : Bb;
            const int d = i < A_DMA ? (wave * A_DMA + i) * 8 * ROWB : BM * ROWB + (wave * B_DMA + (i - A_DMA)) * 8 * ROWB;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (size_t)(soff[i] + koff)),
                                             (__attribute__((address_space(3))) void*)(stage + d), 16, 0, 0);
        }
    };

    // accumulators start at bias (+ residual in LINEAR mode): those loads fly during the DMA prologue
    // and the epilogue becomes store-only
    f32x4 acc[TN][TM];
#pragma unroll
    for (int j = 0; j < TM; j++) {
        const int m = m0 + wm * WTM + j * 16 + r;
        const bool mv = m < g.M;
        const RowCtx rc = row_ctx(g, mv ? m : g.M - 1);
#pragma unroll
        for (int i = 0; i < TN; i++) {
            const int n = n0 + wn * WTN + i * 16 + q * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (mv && n + 3 < g.Nvalid) {
                if (g.bias) {
                    if (g.bias_on_rows) v += rc.bias_row;
                    else v = *reinterpret_cast<const f32x4*>(g.bias + n);
                }
                if (rc.resid_row && g.mode == GEMM_EPI_LINEAR) v += *reinterpret_cast<const f32x4*>(rc.resid_row + n);
            }
            acc[i][j] = v;
        }
    }

    auto stamp = [&](int slot) {
        if (g.stamps && tid == 0) {
            g.stamps[(size_t)blockIdx.x * 8 + slot] = __builtin_readcyclecounter();
            g.stamps[(size_t)blockIdx.x * 8 + 4 + slot] = wall_clock64();
        }
    };
    stamp(0);
    const int nk = g.K / BK;
    constexpr int KSTEPS = BK / 32;
    auto read_frags = [&](Frag<T>* fa, Frag<T>* fb, int kt, int ks) {
        const char* cA = smem + (kt % NS) * STAGE;
        const char* cB = cA + BM * ROWB;
#pragma unroll
        for (int j = 0; j < TM; j++) lds_frag<ROWB>(fa[j], cA, wm * WTM + j * 16 + r, ks, q);
#pragma unroll
        for (int i = 0; i < TN; i++) lds_frag<ROWB>(fb[i], cB, wn * WTN + i * 16 + r, ks, q);
    };
    auto mma_step = [&](const Frag<T>* fa, const Frag<T>* fb) {
#pragma unroll
        for (int i = 0; i < TN; i++)
#pragma unroll
            for (int j = 0; j < TM; j++) mma16(acc[i][j], fb[i], fa[j]);
    };
    if constexpr (KSTEPS == 2 && PIPE) {
        // Pipeline (bf16): every MFMA cluster overlaps the fragment reads of the NEXT k-step, and the
        // slab hand-over (counted vmcnt + one s_barrier + next DMA issue) sits between two clusters
        // whose operands are already in registers.  After the barrier of iteration kt slab kt lives
        // entirely in registers, so its ring slot is refilled with slab kt+NS: slabs kt+1 (landed)
        // .. kt+NS-1 (in flight) occupy the other slots.
        // A fragment j is dead after the TN MFMAs of its row, so the next step's A fragment j is read
        // right behind them (same registers); only the TN B fragments are double-buffered.
        auto step = [&](Frag<T>* fa_cur, const Frag<T>* fb_cur, Frag<T>* fb_nxt, int kt_n, int ks_n, bool have_next) {
            const char* cA = smem + (kt_n % NS) * STAGE;
            const char* cB = cA + BM * ROWB;
#pragma unroll
            for (int j = 0; j < TM; j++) {
#ifdef PNP_DEV
                if (g.ablate == 2) {                         // timing ablation: no MFMA, operands kept alive
#pragma unroll
                    for (int i = 0; i < TN; i++) asm volatile("" ::"v"(fb_cur[i].v), "v"(fa_cur[j].v));
                } else
#endif
                {
#pragma unroll
                    for (int i = 0; i < TN; i++) mma16(acc[i][j], fb_cur[i], fa_cur[j]);
                }
                if (have_next) {
                    lds_frag<ROWB>(fa_cur[j], cA, wm * WTM + j * 16 + r, ks_n, q);
                    if (j < TN) lds_frag<ROWB>(fb_nxt[j], cB, wn * WTN + j * 16 + r, ks_n, q);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, TN, 0);                  // TN MFMAs
                if (j < TN) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);       // then this row's refill reads
                else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        };
#pragma unroll
        for (int i = 0; i < NS; i++)
            if (i < nk) issue(i);
        // slab 0 landed: at most min(NS, nk) - 1 younger slabs may stay in flight
        if (nk >= NS) PNP_WAIT_VM((NS - 1) * NDMA);
        else if (NS > 2 && nk == NS - 1) PNP_WAIT_VM((NS > 2 ? NS - 2 : 0) * NDMA);
        else PNP_WAIT_VM(0);
        __builtin_amdgcn_s_barrier();
        stamp(1);
        Frag<T> fa[TM], fb0[TN], fb1[TN];
        read_frags(fa, fb0, 0, 0);
        for (int kt = 0; kt < nk; kt++) {
            step(fa, fb0, fb1, kt, 1, true);                 // MFMAs of (kt, 0) || reads of (kt, 1)
            if (kt + 1 < nk) {
                // own reads of slab kt are complete, slab kt+1 has landed (younger slabs stay in flight)
                if (kt + NS - 1 < nk) PNP_WAIT_VM_LGKM((NS - 2) * NDMA);
                else PNP_WAIT_VM_LGKM(0);
                __builtin_amdgcn_s_barrier();
                if (kt + NS < nk) issue(kt + NS);
                step(fa, fb1, fb0, kt + 1, 0, true);         // MFMAs of (kt, 1) || reads of (kt+1, 0)
            } else {
                step(fa, fb1, fb0, kt, 0, false);
            }
        }
    } else {
        // simple ring (fp32 parity mode: one k-step per slab; large-tile bf16 variant: fragments are
        // not double-buffered, the co-resident wave of the SIMD covers the LDS latency)
        issue(0);
        if (nk > 1 && NS > 2) issue(1);
        for (int kt = 0; kt < nk; kt++) {
            if (NS > 2 && kt + 1 < nk) PNP_WAIT_VM(NDMA);
            else PNP_WAIT_VM(0);
            __builtin_amdgcn_s_barrier();
            if (NS > 2) {
                if (kt + 2 < nk) issue(kt + 2);
            } else if (kt + 1 < nk) {
                issue(kt + 1);
            }
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ks++) {
                Frag<T> fa[TM], fb[TN];
                read_frags(fa, fb, kt, ks);
                __builtin_amdgcn_s_setprio(1);
                mma_step(fa, fb);
                __builtin_amdgcn_s_setprio(0);
            }
            if (NS == 2) {                     // the single spare slot is refilled next iteration: finish reading first
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
    }
    stamp(2);
#pragma unroll
    for (int j = 0; j < TM; j++) {
        const int m = m0 + wm * WTM + j * 16 + r;
        if (m >= g.M) continue;
        const RowCtx rc = row_ctx(g, m);
#pragma unroll
        for (int i = 0; i < TN; i++) {
            const int n = n0 + wn * WTN + i * 16 + q * 4;
            if (n >= g.Nvalid) continue;
            store_frag<T>(g, rc, acc[i][j], m, n, n + 3 < g.Nvalid);
        }
    }
    if (g.stamps) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(3);
    }
}

template <typename T, int BM, int BN, int WTM, int WTN, int NS, bool PIPE = true>
static int launch_big(const GemmArgs& g, hipStream_t s) {
    const int nbm = (g.M + BM - 1) / BM, nbn = g.N / BN;
    const size_t smem = (size_t)NS * (BM + BN) * 128;
    static std::atomic<uint32_t> opted{0};            // per device ordinal (common.h: lds_opt_in)
    if (lds_opt_in(opted, reinterpret_cast<const void*>(gemm_nt_big_kernel<T, BM, BN, WTM, WTN, NS, PIPE>), (int)smem) != PNP_OK) return PNP_ERR_HIP;
    hipLaunchKernelGGL((gemm_nt_big_kernel<T, BM, BN, WTM, WTN, NS, PIPE>), dim3(nbm * nbn), dim3((BM / WTM) * (BN / WTN) * 64), smem, s, g);
    return hipGetLastError() == hipSuccess ? PNP_OK : PNP_ERR_HIP;
}

// ------------------------------------------------------------------------------------------
// Wide-tile bf16 kernel: 256 x 256 block tile, 8 waves (2 x 4) of 128 x 64, v_mfma_f32_32x32x16_bf16,
// PERSISTENT: one workgroup per CU walks tiles blockIdx.x, blockIdx.x + gridDim.x, ...
//
// Why this shape: one CU takes in at most 64 B/clk of global data (TA / vector L1 path) and every
// LDS-DMA piece stalls its issuing wave for 60+ cycles, so a 128 x 128 tile (32 KB per 512 MFMA
// cycles) is ingest-bound.  At 256 x 256 the same 64 KB slab feeds 2048 MFMA cycles, the 32x32x16
// form halves the LDS fragment bytes per FLOP and leaves 24 free issue cycles per MFMA, and two
// waves per SIMD (256 registers each: 128 accumulators + two fragment sets) cover each other's
// DMA-issue stalls.
//
// Schedule per 64-deep slab (4 k16 sub-steps, fragments double-buffered in registers):
//   MFMA(s0) || read(s1);  MFMA(s1) || read(s2);  MFMA(s2) || read(s3);
//   wait own reads + own DMA of slab t+1; s_barrier      <- the only barrier per slab
//   issue DMA of slab t+2 into the slot just vacated || read(s0 of slab t+1) || MFMA(s3)
// so the MFMAs of s3 (operands already in registers) cover the barrier hand-over, the DMA issue and
// the first fragment reads of the next slab, and every DMA has a whole slab time to land.
//
// Tile hand-over: after the last slab both ring slots are free.  Slot 0 immediately receives slab 0 of
// the NEXT tile while the accumulators of this tile leave through a staging area that overlays slot 1
// (clock stamps: a fresh workgroup costs ~3 us of launch gap + 1 us of first-slab latency per tile).
//
// Epilogue through LDS: the accumulator layout gives a lane 4 consecutive columns in each of 32
// different rows, i.e. 16-byte fragments of 32 cache lines per store instruction (measured: as long
// as the whole K = 1024 main loop).  Each wave transposes one 32-row quarter of its tile at a time
// (fp32, rows padded by 16 B against bank conflicts) so that 16 consecutive lanes own one 64-column
// row segment and every global access is a full 128/256-byte line.  vmcnt counts loads and stores in
// one in-order counter, so a quarter's residual rows are requested BEFORE its first store and no
// wait ever sits behind a store; epilogues are compile-time variants (no generic branches).

// EPI: 0 = +bias -> bf16 | 1 = +bias, GELU -> bf16 | 2 = +bias +residual -> fp32 |
//      3 = +per-row bias, token columns remapped to per-image padded columns -> bf16 (transposed cross-attention K / V)
// The split-bf16 ("bf16x3") launches of the same tile frame, and their fp32-facing epilogues (4 = +bias -> fp32 | 5 = as 3 with
// fp32 output | 6 = +bias, erf-GELU -> (hi, lo) bf16 pair | 7 = +bias -> pair), live in gemm_x3.hip (v_mfma_f32_16x16x32_bf16).
template <int EPI>
__global__ __launch_bounds__(512) void gemm_nt_wide_kernel(const GemmArgs g) {
    constexpr int BM = 256, BN = 256, ROWB = 128, BK = 64, STAGE = (BM + BN) * ROWB;
    constexpr int TM = 4, TN = 2;                   // 32 x 32 tiles per wave: 128 (m) x 64 (n)
    constexpr int A_DMA = 4, B_DMA = 4, NDMA = 8;   // 8-row pieces per wave per slab
    constexpr int SROW = kWideStageRow;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l32 = lane & 31, hi = lane >> 5;
    const int wm = wave >> 2, wn = wave & 3;
    const int nbm = (g.M + BM - 1) / BM, nbn = g.N / BN;
    const int ntiles = nbm * nbn;
    const int nk = g.K / BK;
    const char* Ab = reinterpret_cast<const char*>(g.A);
    const char* Bb = reinterpret_cast<const char*>(g.B);

    uint32_t soff[NDMA];
    auto set_tile = [&](int tile, int& m0, int& n0) {
        int bm, bn;
        tile_coords<4>(tile, nbm, nbn, bm, bn);
        m0 = bm * BM;
        n0 = bn * BN;
        const uint32_t lda_b = (uint32_t)g.lda * 2, ldb_b = (uint32_t)g.ldb * 2;
        {
            const int pc = lane & 7;
#pragma unroll
            for (int i = 0; i < A_DMA; i++) {
                const int row = (wave * A_DMA + i) * 8 + (lane >> 3);
                int gr = m0 + row;
                gr = gr < g.M ? gr : g.M - 1;
                soff[i] = (uint32_t)gr * lda_b + swz_chunk<ROWB>(row, pc) * 16;
            }
#pragma unroll
            for (int i = 0; i < B_DMA; i++) {
                const int row = (wave * B_DMA + i) * 8 + (lane >> 3);
                int gr = n0 + row;
                gr = gr < g.Nvalid ? gr : g.Nvalid - 1;
                soff[A_DMA + i] = (uint32_t)gr * ldb_b + swz_chunk<ROWB>(row, pc) * 16;
            }
        }
    };
    auto issue_one = [&](int kt, int i) {
#ifdef PNP_DEV
        if (g.ablate == 1 && kt >= 2) return;           // timing ablation: no steady-state DMA
#endif
        char* stage = smem + (kt & 1) * STAGE;
        {
            const uint32_t koff = (uint32_t)kt * ROWB;
            const char* base = i < A_DMA ? Ab : Bb;
            const int d = i < A_DMA ? (wave * A_DMA + i) * 8 * ROWB : BM * ROWB + (wave * B_DMA + (i - A_DMA)) * 8 * ROWB;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (size_t)(soff[i] + koff)),
                                             (__attribute__((address_space(3))) void*)(stage + d), 16, 0, 0);
        }
    };
    auto stamp = [&](int slot) {
        if (g.stamps && tid == 0) {
            g.stamps[(size_t)blockIdx.x * 8 + slot] = __builtin_readcyclecounter();
            g.stamps[(size_t)blockIdx.x * 8 + 4 + slot] = wall_clock64();
        }
    };

    // fragment addresses: row = tile row + l32, logical chunk = 2 ks + hi; the swizzle term depends on
    // l32 only (tile row offsets are multiples of 32), so one XOR per k16 sub-step serves all tiles
    const int sw = (l32 >> 1) & 7;
    const int a_row_off = (wm * 128 + l32) * ROWB;
    const int b_row_off = BM * ROWB + (wn * 64 + l32) * ROWB;
    float* const stg = reinterpret_cast<float*>(smem + 65536) + wave * (32 * SROW);

    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    int m0, n0;
    bool drain32 = false;                           // the previous tile's epilogue left >= 32 stores behind this tile's slab 0
    stamp(0);
    set_tile(tile, m0, n0);
#pragma unroll
    for (int i = 0; i < NDMA; i++) issue_one(0, i);

    for (;;) {
        // lane owns row m = .. + l32 and, per 4-register group gq, columns 8 gq + 4 hi .. + 3 of each 32 x 32
        // tile (operands are swapped at the MFMA: D = Btile * Atile^T)
        f32x16 acc[TN][TM];
#pragma unroll
        for (int j = 0; j < TM; j++)
#pragma unroll
            for (int i = 0; i < TN; i++)
#pragma unroll
                for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

        auto read_frags = [&](bf16x8* fa, bf16x8* fb, int kt, int ks) {
            const char* st = smem + (kt & 1) * STAGE + (((ks * 2 + hi) ^ sw) << 4);
#pragma unroll
            for (int j = 0; j < TM; j++) fa[j] = *reinterpret_cast<const bf16x8*>(st + a_row_off + j * 32 * ROWB);
#pragma unroll
            for (int i = 0; i < TN; i++) fb[i] = *reinterpret_cast<const bf16x8*>(st + b_row_off + i * 32 * ROWB);
        };
        auto mma_all = [&](const bf16x8* fa, const bf16x8* fb) {
#pragma unroll
            for (int j = 0; j < TM; j++)
#pragma unroll
                for (int i = 0; i < TN; i++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[i], fa[j], acc[i][j], 0, 0, 0);
        };
        // interleave hint: one LDS read behind each of the first six MFMAs of a sub-step
        auto sched_plain = [&]() {
#pragma unroll
            for (int u = 0; u < TM + TN; u++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, TM * TN - (TM + TN), 0);
        };

        // slab 0 of this tile is in flight (issued before the previous tile's epilogue, or above).  vmcnt counts in
        // issue order, so after a full-tile epilogue that issued >= 32 stores behind the DMA pieces, "at most 32
        // outstanding" already means the slab has landed: the stores of the previous tile keep draining under the first
        // sub-steps of this one instead of stalling every wave here
        if (drain32) PNP_WAIT_VM(32);
        else PNP_WAIT_VM(0);
        __builtin_amdgcn_s_barrier();               // slab 0 complete; staging area (slot 1) no longer read
        {
            if (nk > 1) {
    #pragma unroll
                for (int i = 0; i < NDMA; i++) issue_one(1, i);
            }
            bf16x8 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
            read_frags(fa0, fb0, 0, 0);
            // sub-steps s0..s2 of slab kt: MFMAs on one fragment set, reads of the next sub-step into the other
            auto body012 = [&](int kt) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                read_frags(fa1, fb1, kt, 1);
                mma_all(fa0, fb0);
                sched_plain();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                read_frags(fa0, fb0, kt, 2);
                mma_all(fa1, fb1);
                sched_plain();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                read_frags(fa1, fb1, kt, 3);
                mma_all(fa0, fb0);
                sched_plain();
            };
            int kt = 0;
            for (; kt + 2 < nk; kt++) {                 // steady state: slab kt+2 exists
                body012(kt);
                PNP_WAIT_VM_LGKM(0);                    // own s3 fragments in registers, own pieces of slab kt+1 landed
                __builtin_amdgcn_s_barrier();           // slab kt+1 complete; nobody reads slab kt's slot any more
                read_frags(fa0, fb0, kt + 1, 0);
    #pragma unroll
                for (int i = 0; i < NDMA; i++) issue_one(kt + 2, i);
                mma_all(fa1, fb1);
                // LDS-DMA writes may not be reordered against the LDS reads: reads lead, DMA pieces follow
    #pragma unroll
                for (int u = 0; u < 3; u++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
    #pragma unroll
                for (int u = 0; u < 4; u++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            }
            if (kt + 1 < nk) {                          // second-to-last slab: nothing left to fetch
                body012(kt);
                PNP_WAIT_VM_LGKM(0);
                __builtin_amdgcn_s_barrier();
                read_frags(fa0, fb0, kt + 1, 0);
                mma_all(fa1, fb1);
                sched_plain();
                kt++;
            }
            body012(kt);                                // last slab
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            mma_all(fa1, fb1);
        }
        if (tile == (int)blockIdx.x) stamp(2);

        __syncthreads();                            // every wave is done reading the last slab: both slots are free
        const int em0 = m0, en0 = n0;
        // epilogue operands are requested BEFORE the next tile's first slab: vmcnt is one in-order counter, a
        // wait for a load issued behind the DMA pieces would also wait for those (HBM latency, ~2 us)
        const int n = en0 + wn * 64 + (lane & 15) * 4;          // this lane's 4 output columns (same for every row)
        const bool nv = n < g.Nvalid;                           // N is a multiple of 4 on the row-major epilogues
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        f32x4 rv[8];
        f32x4 bacc[TN][4];                          // bf16 epilogues: bias in the accumulator layout
        if constexpr (EPI == WIDE_BF16 || EPI == WIDE_GELU_BF16) {
#pragma unroll
            for (int i = 0; i < TN; i++)
#pragma unroll
                for (int gq = 0; gq < 4; gq++) {
                    const int nn = en0 + wn * 64 + i * 32 + gq * 8 + hi * 4;
                    bacc[i][gq] = (g.bias && nn < g.Nvalid) ? *reinterpret_cast<const f32x4*>(g.bias + nn) : bv;
                }
        }
        if constexpr (EPI == WIDE_RESID_F32) {
            if (g.bias && nv) bv = *reinterpret_cast<const f32x4*>(g.bias + n);
            {
                const int mrow0 = em0 + wm * 128 + (lane >> 4);
#pragma unroll
                for (int it = 0; it < 8; it++) {
                    const int m = mrow0 + it * 4;
                    rv[it] = (m < g.M && nv) ? *reinterpret_cast<const f32x4*>(g.resid + (size_t)m * g.ldr + n) : bv;
                }
            }
        }
        float brow[TM] = {0.f, 0.f, 0.f, 0.f};        // TOKCOLS: bias of this lane's row in each 32-row tile
        if constexpr (EPI == WIDE_TOKCOLS_BF16) {
#pragma unroll
            for (int j = 0; j < TM; j++) {
                const int m = em0 + wm * 128 + j * 32 + l32;
                if (g.bias && m < g.M) brow[j] = g.bias[m];
            }
        }
        const int next = tile + gridDim.x;
        if (next < ntiles) {                        // slab 0 of the next tile flies during this tile's epilogue
            set_tile(next, m0, n0);
#pragma unroll
            for (int i = 0; i < NDMA; i++) issue_one(0, i);
        }

        if constexpr (EPI == WIDE_TOKCOLS_BF16) {
            // per-row bias and bf16 rounding in the accumulator layout, bf16 staging in two 64-row halves; on the way
            // out a lane owns a PAIR of token columns (4-byte stores, 128 contiguous bytes per row, two rows per
            // instruction) when col_div is even -- pairs then never straddle an image -- else single tokens
            constexpr int HROW = 68;
            bf16* const stgh = reinterpret_cast<bf16*>(smem + 65536) + wave * (64 * HROW);
            const bool pairs = (g.col_div & 1) == 0;
            const int tl0 = pairs ? 2 * l32 : lane;                      // token column inside the wave's 64
            const int tok = en0 + wn * 64 + tl0;
            size_t ocol = tok;
            if (g.col_div > 0) {
                const int b = tok / g.col_div;
                int tl = tok - b * g.col_div;
                ocol = (size_t)b * g.col_pad + tl;
            }
            bf16* const ocolp = reinterpret_cast<bf16*>(g.out_t) + ocol;
            const bool tv = tok < g.Nvalid;         // Nvalid is even whenever col_div is (whole images)
#pragma unroll
            for (int half = 0; half < 2; half++) {
                const int mbase = em0 + wm * 128 + half * 64;
#pragma unroll
                for (int jj = 0; jj < 2; jj++)
#pragma unroll
                    for (int i = 0; i < TN; i++)
#pragma unroll
                        for (int gq = 0; gq < 4; gq++) {
                            const f32x16& a = acc[i][half * 2 + jj];
                            const float br = brow[half * 2 + jj];
                            const bf16x4 pk = {(bf16)(a[gq * 4] + br), (bf16)(a[gq * 4 + 1] + br), (bf16)(a[gq * 4 + 2] + br),
                                               (bf16)(a[gq * 4 + 3] + br)};
                            *reinterpret_cast<bf16x4*>(stgh + (jj * 32 + l32) * HROW + i * 32 + gq * 8 + hi * 4) = pk;
                        }
                if (pairs) {
                    uint32_t sv[32];
#pragma unroll
                    for (int it = 0; it < 32; it++) sv[it] = *reinterpret_cast<const uint32_t*>(stgh + (it * 2 + hi) * HROW + tl0);
#pragma unroll
                    for (int it = 0; it < 32; it++) {
                        const int m = mbase + it * 2 + hi;
                        if (tv && m < g.M) *reinterpret_cast<uint32_t*>(ocolp + (size_t)m * g.ldo_t) = sv[it];
                    }
                } else {
#pragma unroll 8
                    for (int row = 0; row < 64; row++) {
                        const bf16 v = stgh[row * HROW + lane];
                        if (tv && mbase + row < g.M) ocolp[(size_t)(mbase + row) * g.ldo_t] = v;
                    }
                }
            }
        } else if constexpr (EPI == WIDE_BF16 || EPI == WIDE_GELU_BF16) {
            // bias (+GELU) and the bf16 rounding happen in the accumulator layout; the tile is staged as bf16
            // (half the LDS bytes: LDS stores run at ~80 B/clk/CU) in two 64-row halves, 136-byte rows so the
            // 8-byte stores of 16 lanes (16 rows, same column) fall in 16 different bank pairs
            constexpr int HROW = 68;                // bf16 per staged row (64 + 4 pad)
            bf16* const stgh = reinterpret_cast<bf16*>(smem + 65536) + wave * (64 * HROW);
            const bool full = (em0 + BM <= g.M) && (en0 + BN <= g.Nvalid);
            bf16* const obase = reinterpret_cast<bf16*>(g.out_t) + (size_t)(em0 + wm * 128 + (lane >> 4)) * g.ldo_t + n;
#pragma unroll
            for (int half = 0; half < 2; half++) {
#pragma unroll
                for (int jj = 0; jj < 2; jj++)
#pragma unroll
                    for (int i = 0; i < TN; i++)
#pragma unroll
                        for (int gq = 0; gq < 4; gq++) {
                            const f32x16& a = acc[i][half * 2 + jj];
                            f32x4 v = {a[gq * 4], a[gq * 4 + 1], a[gq * 4 + 2], a[gq * 4 + 3]};
                            v += bacc[i][gq];
                            if constexpr (EPI == WIDE_GELU_BF16) {
#pragma unroll
                                for (int e = 0; e < 4; e++) v[e] = gelu_logistic_fit(v[e]);
                            }
                            const bf16x4 pk = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                            *reinterpret_cast<bf16x4*>(stgh + (jj * 32 + l32) * HROW + i * 32 + gq * 8 + hi * 4) = pk;
                        }
                bf16x4 sv[16];                      // all LDS reads of the half in flight before the first store
#pragma unroll
                for (int it = 0; it < 16; it++)
                    sv[it] = *reinterpret_cast<const bf16x4*>(stgh + (it * 4 + (lane >> 4)) * HROW + (lane & 15) * 4);
                if (full) {
#pragma unroll
                    for (int it = 0; it < 16; it++)
                        *reinterpret_cast<bf16x4*>(obase + (size_t)(half * 64 + it * 4) * g.ldo_t) = sv[it];
                } else {
#pragma unroll
                    for (int it = 0; it < 16; it++) {
                        const int m = em0 + wm * 128 + half * 64 + it * 4 + (lane >> 4);
                        if (m < g.M && nv) *reinterpret_cast<bf16x4*>(obase + (size_t)(half * 64 + it * 4) * g.ldo_t) = sv[it];
                    }
                }
            }
        } else {
            // fp32 residual epilogue, 32-row quarters staged as fp32.  The residual rows of quarter q+1 are
            // requested before the stores of quarter q (two register sets), so each wait has a whole quarter
            // of work in front of it and never sits behind a store.
            constexpr bool kResid = EPI == WIDE_RESID_F32;
            const bool full = (em0 + BM <= g.M) && (en0 + BN <= g.Nvalid);
            const float* const rbase = kResid ? g.resid + (size_t)(em0 + wm * 128 + (lane >> 4)) * g.ldr + n : nullptr;
            float* const obase = g.out_f32 + (size_t)(em0 + wm * 128 + (lane >> 4)) * g.ldo + n;
            f32x4 rw[8];
#pragma unroll
            for (int qd = 0; qd < 4; qd++) {
                f32x4* const rcur = (qd & 1) ? rw : rv;
                f32x4* const rnxt = (qd & 1) ? rv : rw;
#pragma unroll
                for (int i = 0; i < TN; i++)
#pragma unroll
                    for (int gq = 0; gq < 4; gq++) {
                        const f32x16& a = acc[i][qd];
                        const f32x4 v = {a[gq * 4], a[gq * 4 + 1], a[gq * 4 + 2], a[gq * 4 + 3]};
                        *reinterpret_cast<f32x4*>(stg + l32 * SROW + i * 32 + gq * 8 + hi * 4) = v;
                    }
                if (kResid && qd < 3) {
#pragma unroll
                    for (int it = 0; it < 8; it++) {
                        const int m = em0 + wm * 128 + (qd + 1) * 32 + it * 4 + (lane >> 4);
                        rnxt[it] = (full || (m < g.M && nv)) ? *reinterpret_cast<const f32x4*>(rbase + (size_t)((qd + 1) * 32 + it * 4) * g.ldr) : bv;
                    }
                }
                f32x4 sv[8];                        // all LDS reads of the quarter in flight before the first use
#pragma unroll
                for (int it = 0; it < 8; it++)
                    sv[it] = *reinterpret_cast<const f32x4*>(stg + (it * 4 + (lane >> 4)) * SROW + (lane & 15) * 4);
#pragma unroll
                for (int it = 0; it < 8; it++) {
                    f32x4 v = sv[it] + bv;
                    if constexpr (kResid) v += rcur[it];
                    const int m = em0 + wm * 128 + qd * 32 + it * 4 + (lane >> 4);
                    if (full || (m < g.M && nv)) *reinterpret_cast<f32x4*>(obase + (size_t)(qd * 32 + it * 4) * g.ldo) = v;
                }
            }
        }
        if (tile == (int)blockIdx.x) stamp(1);     // diagnostics: first tile's epilogue done (stores issued)
        if (next >= ntiles) break;
        tile = next;
        // every lane of a full tile executes all of the epilogue's stores (32 per wave, 64 for the split pair)
        drain32 = EPI != WIDE_TOKCOLS_BF16 && (em0 + BM <= g.M) && (en0 + BN <= g.Nvalid);
    }
    if (g.stamps) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(3);                                   // whole workgroup (all its tiles) done
    }
}

// the wide kernel's compile-time epilogues cover the ViT block's dense layers; anything else (row / column
// remaps, per-row bias, stashes, dual outputs) stays on the generic kernels
static int wide_epilogue_kind(const GemmArgs& g) {
    if (g.aux || g.row_div) return -1;
    if (g.A_lo || g.B_lo) {                         // split-bf16 operands: the fp32-facing epilogues
        if (!g.A_lo || !g.B_lo) return -1;
        if (g.mode == GEMM_EPI_LINEAR && !g.resid && g.out_f32 && !g.out_t && (g.bias_on_rows || !g.bias) && (g.col_div > 0 || g.bias_on_rows))
            return WIDE_TOKCOLS_F32;
        if (g.col_div || g.bias_on_rows || (g.Nvalid & 3)) return -1;
        if (g.mode == GEMM_EPI_LINEAR && g.resid && g.out_f32 && !g.out_t) return WIDE_RESID_F32;
        if (g.mode == GEMM_EPI_LINEAR && !g.resid && g.out_f32 && !g.out_t) return WIDE_BIAS_F32;
        if (g.mode == GEMM_EPI_GELU && !g.resid && g.out_t && g.out_lo && !g.out_f32) return WIDE_GELU_SPLIT;
        if (g.mode == GEMM_EPI_LINEAR && !g.resid && g.out_t && g.out_lo && !g.out_f32) return WIDE_SPLIT;
        return -1;
    }
    if (g.mode == GEMM_EPI_LINEAR && !g.resid && g.out_t && !g.out_f32 && (g.bias_on_rows || !g.bias) && (g.col_div > 0 || g.bias_on_rows))
        return WIDE_TOKCOLS_BF16;
    if (g.col_div || g.bias_on_rows || (g.Nvalid & 3)) return -1;
    if (g.mode == GEMM_EPI_LINEAR && !g.resid && g.out_t && !g.out_f32) return WIDE_BF16;
    if (g.mode == GEMM_EPI_GELU && !g.resid && g.out_t && !g.out_f32) return WIDE_GELU_BF16;
    if (g.mode == GEMM_EPI_LINEAR && g.resid && g.out_f32 && !g.out_t) return WIDE_RESID_F32;
    return -1;
}

template <int EPI>
static int launch_wide(const GemmArgs& g, hipStream_t s) {
    const int nbm = (g.M + 255) / 256, nbn = g.N / 256;
    const int n_cu = device_cu_count();
    if (!n_cu) return PNP_ERR_HIP;
    static std::atomic<uint32_t> opted{0};            // per device ordinal (common.h: lds_opt_in)
    if (lds_opt_in(opted, reinterpret_cast<const void*>(gemm_nt_wide_kernel<EPI>), kWideSmem) != PNP_OK) return PNP_ERR_HIP;
    const int ntiles = nbm * nbn;
    int cap = n_cu;
#ifdef PNP_DEV
    if (getenv("PNP_GEMM_GRID")) cap = atoi(getenv("PNP_GEMM_GRID"));
#endif
    const int grid = ntiles > cap ? cap : ntiles;        // one workgroup per CU (LDS-limited) walks the tiles
    hipLaunchKernelGGL((gemm_nt_wide_kernel<EPI>), dim3(grid), dim3(512), kWideSmem, s, g);
    return hipGetLastError() == hipSuccess ? PNP_OK : PNP_ERR_HIP;
}

// ------------------------------------------------------------------------------------------
// Text-side GEMM of the split-bf16 mode: C[M,N] = A[M,K] . B[N,K]^T with A = fp32 activations (M = B*L rows, a few hundred
// to a few thousand) and B = a weight stored as a bf16 (hi | lo) pair.  The fp32 MFMA runs at 1/16 of the bf16 rate, so the
// exact-fp32 form of these launches is MFMA-bound at ~40 us each (396 launches per 35-image bench step); here the activation tile is
// split into (hi, lo) bf16 ONCE by the threads that stage it (global -> registers -> two 16-byte LDS stores per 8 values),
// the weight tile arrives by LDS-DMA already split, and each fragment pair issues three v_mfma_f32_16x16x32_bf16
// (a_hi.b_lo, a_lo.b_hi, a_hi.b_hi: fp32-class product, see gemm_x3.hip).
// Tile 64 x 64, 4 waves of 32 x 32, 32-deep k-slabs, 4-slot ring (three slabs in flight): slot = A_hi | A_lo | B_hi | B_lo, each 64 rows x 64 bytes,
// 16-byte chunks swizzled chunk ^= (-(row >> 2)) & 3 (the 16 lanes a ds_read_b128 services together -- 4 rows of one q
// and 8 rows of the next -- then hit 16 different slots).  Epilogue: the generic store_frag (bias, residual, GELU + stash,
// GELU', fp32 outputs), same accumulator layout as gemm_nt_big_kernel.
// NW = 4 (wave tile 32 x 32) or 8 waves (32 x 16: half the instructions per wave and slab, two waves per SIMD to overlap).
// KS = k32 steps per slab (1: 32-deep, 64-byte LDS rows; 2: 64-deep, 128-byte rows swizzled like the wide bf16 kernel): these
// launches move ~260 MB through L2 for a few GFLOP and run at the latency x bytes-in-flight of the CUs they occupy (168-672
// workgroups), so the deeper slab (twice the bytes in flight per workgroup, half the barriers) is what the product launches.
template <int NW, int KS, int NS = 4>
__global__ __launch_bounds__(NW * 64) void gemm_nt_small_x3_kernel(const GemmArgs g) {
    constexpr int BM = 64, BN = 64, ROWB = 64 * KS, ARR = 64 * ROWB, SLOT = 4 * ARR;
    constexpr int LEAD = NS - 1;                       // slabs in flight ahead of the one being multiplied (3 | 2)
    static_assert(NS == 3 || NS == 4, "ring depth");
    constexpr int WN = NW / 2, TN = 2 / (NW / 4);      // waves along n, 16-column tiles per wave (2 | 1); 2 row tiles per wave
    constexpr int EPT = 32 * KS / NW;                  // staged A values per thread and slab (NW threads per row)
    constexpr int PROWS = 1024 / ROWB;                 // rows of a 1 KB DMA piece (16 | 8)
    constexpr int BP = 2 * (64 / PROWS) / NW;          // B DMA pieces per wave and slab (hi and lo arrays)
    constexpr int VMI = (EPT + 3) / 4 + BP;            // vector-memory operations a wave issues per slab (16-byte A loads + B pieces)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int wm = wave / WN, wn = wave % WN;
    // XCD-grouped, column-major tile order: the workgroups of one XCD (blockIdx.x % 8) take a contiguous run of tile ids =
    // a few column tiles x all row tiles, so an XCD's L2 holds its slice of the weight and the activations instead of
    // every XCD streaming the whole weight from the Infinity Cache
    const int nbm = (g.M + BM - 1) / BM, nbn = g.N / BN;
    int bm, bn;
    tile_coords<16>(blockIdx.x, nbm, nbn, bm, bn);
    const int m0 = bm * BM, n0 = bn * BN;
    const int nk = g.K / (32 * KS);
    auto swz = [](int row) { return KS == 1 ? ((-(row >> 2)) & 3) : ((row >> 1) & 7); };

    // A staging: thread -> row tid / NW, EPT consecutive k values of the slab (fp32)
    const int a_row = tid / NW, a_e = (tid % NW) * EPT;
    int a_gr = m0 + a_row;
    a_gr = a_gr < g.M ? a_gr : g.M - 1;
    const float* a_src = reinterpret_cast<const float*>(g.A) + (size_t)a_gr * g.lda + a_e;
    // B DMA: pieces of PROWS rows; piece p of the slab: array (hi | lo) = p / (64 / PROWS), rows PROWS (p % (64 / PROWS)) ..
    const char* Bh = reinterpret_cast<const char*>(g.B);
    const char* Bl = reinterpret_cast<const char*>(g.B_lo);
    constexpr int PPA = 64 / PROWS, LPR = ROWB / 16;   // pieces per array, lanes per row
    uint32_t b_off[BP];
#pragma unroll
    for (int i = 0; i < BP; i++) {
        const int p = wave * BP + i;
        const int row = (p % PPA) * PROWS + lane / LPR;
        int gr = n0 + row;
        gr = gr < g.Nvalid ? gr : g.Nvalid - 1;
        const int c = (lane % LPR) ^ swz(row);
        b_off[i] = (uint32_t)gr * (uint32_t)g.ldb * 2 + c * 16;
    }
    auto issue_b = [&](int kt) {
        char* st = smem + (kt % NS) * SLOT + 2 * ARR;
        const uint32_t koff = (uint32_t)kt * ROWB;
#pragma unroll
        for (int i = 0; i < BP; i++) {
            const int p = wave * BP + i;               // uniform
            const char* base = p < PPA ? Bh : Bl;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (size_t)(b_off[i] + koff)),
                                             (__attribute__((address_space(3))) void*)(st + (p / PPA) * ARR + (p % PPA) * 1024), 16, 0, 0);
        }
    };
    // staged A values: two register sets, slab kt + 1 (stored to LDS in iteration kt) and slab kt + 2; a load has two
    // iterations to land
    typedef __attribute__((ext_vector_type(EPT))) float fvec;
    fvec ra[2];
    auto load_a = [&](int kt) { ra[kt & 1] = *reinterpret_cast<const fvec*>(a_src + (size_t)kt * (32 * KS)); };
    auto store_a = [&](int kt) {                      // x = hi + lo, both bf16 (round to nearest even)
        char* st = smem + (kt % NS) * SLOT + a_row * ROWB;
        if constexpr (EPT >= 8) {
#pragma unroll
            for (int c8 = 0; c8 < EPT / 8; c8++) {
                bf16x8 h, l;
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    h[e] = (bf16)ra[kt & 1][c8 * 8 + e];
                    l[e] = (bf16)(ra[kt & 1][c8 * 8 + e] - (float)h[e]);
                }
                const int off = (((a_e >> 3) + c8) ^ swz(a_row)) << 4;
                *reinterpret_cast<bf16x8*>(st + off) = h;
                *reinterpret_cast<bf16x8*>(st + ARR + off) = l;
            }
        } else {
            bf16x4 h, l;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                h[e] = (bf16)ra[kt & 1][e];
                l[e] = (bf16)(ra[kt & 1][e] - (float)h[e]);
            }
            const int off = (((a_e >> 3) ^ swz(a_row)) << 4) + (a_e & 7) * 2;
            *reinterpret_cast<bf16x4*>(st + off) = h;
            *reinterpret_cast<bf16x4*>(st + ARR + off) = l;
        }
    };

    f32x4 acc[TN][2];
#pragma unroll
    for (int i = 0; i < TN; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment offsets: row of the 16-row tile = r, k = 32 ks + 8 q .. + 7 -> logical chunk 4 ks + q; tile row offsets are
    // multiples of 16, which leave both swizzles unchanged
    const int fa_row = (wm * 32 + r) * ROWB, fb_row = 2 * ARR + (wn * 16 * TN + r) * ROWB;
    const int fsw = swz(r);

    // prologue: slab 0 (A written at once); slabs 1 .. LEAD - 1 in flight (A in registers, B by DMA)
    load_a(0);
    issue_b(0);
    store_a(0);
    if (nk > 1) {
        load_a(1);
        issue_b(1);
    }
    if (LEAD > 2 && nk > 2) {
        load_a(2);
        issue_b(2);
    }
    // one k-slab; REST = slabs behind kt that exist (3+: steady state, compile-time so the loop body has no branches)
    auto iter = [&](int kt, auto rest) {
        constexpr int REST = decltype(rest)::value;
        // slab kt complete for everybody: own B pieces of slab kt landed and the A values of slab kt + 1 loaded; still in
        // flight (issue order): B(kt+1) | A(kt+2), B(kt+2)
        if constexpr (REST >= 2 && LEAD == 3) PNP_WAIT_VM_LGKM(BP + VMI);
        else if constexpr (REST >= 1) PNP_WAIT_VM_LGKM(BP);
        else PNP_WAIT_VM_LGKM(0);
        __builtin_amdgcn_s_barrier();
        // fragment reads first: their latency runs under the conversion of the next slab's A values
        const char* st = smem + (kt % NS) * SLOT;
        bf16x8 ah[KS][2], al[KS][2], bh[KS][TN], bl[KS][TN];
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            const int co = ((ks * 4 + q) ^ fsw) << 4;
#pragma unroll
            for (int j = 0; j < 2; j++) {
                ah[ks][j] = *reinterpret_cast<const bf16x8*>(st + fa_row + j * 16 * ROWB + co);
                al[ks][j] = *reinterpret_cast<const bf16x8*>(st + ARR + fa_row + j * 16 * ROWB + co);
            }
#pragma unroll
            for (int i = 0; i < TN; i++) {
                bh[ks][i] = *reinterpret_cast<const bf16x8*>(st + fb_row + i * 16 * ROWB + co);
                bl[ks][i] = *reinterpret_cast<const bf16x8*>(st + ARR + fb_row + i * 16 * ROWB + co);
            }
        }
        if constexpr (REST >= 1) store_a(kt + 1);     // slot (kt+1) % NS was last read in iteration kt + 1 - NS
        if constexpr (REST >= LEAD) {
            load_a(kt + LEAD);                        // (LEAD = 3: into the register set store_a has just emptied)
            issue_b(kt + LEAD);                       // slot (kt+LEAD) % NS was last read in iteration kt-1: free since this barrier
        }
#pragma unroll
        for (int ks = 0; ks < KS; ks++)
#pragma unroll
            for (int i = 0; i < TN; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[ks][i], ah[ks][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[ks][i], al[ks][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[ks][i], ah[ks][j], acc[i][j], 0, 0, 0);
                }
    };
    constexpr std::integral_constant<int, 3> r3{};
    constexpr std::integral_constant<int, 2> r2{};
    constexpr std::integral_constant<int, 1> r1{};
    constexpr std::integral_constant<int, 0> r0{};
    int kt = 0;
    for (; kt + LEAD + 1 < nk; kt += 2) {             // two iterations per trip: the register set indices are compile-time
        iter(kt, r3);                                 // (REST >= LEAD is all the steady state needs to know)
        iter(kt + 1, r3);
    }
    for (; kt < nk; kt++) {                           // the last (up to four) slabs
        const int rest = nk - 1 - kt;
        if (rest >= 3) iter(kt, r3);
        else if (rest == 2) iter(kt, r2);
        else if (rest == 1) iter(kt, r1);
        else iter(kt, r0);
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int m = m0 + wm * 32 + j * 16 + r;
        if (m >= g.M) continue;
        const RowCtx rc = row_ctx(g, m);
#pragma unroll
        for (int i = 0; i < TN; i++) {
            const int n = n0 + wn * 16 * TN + i * 16 + q * 4;
            if (n >= g.Nvalid) continue;
            store_frag<float>(g, rc, acc[i][j], m, n);
        }
    }
}

template <int NW, int KS, int NS = 4>
static int launch_small_x3_t(const GemmArgs& g, hipStream_t s) {
    const int nbm = (g.M + 63) / 64, nbn = g.N / 64;
    constexpr int smem = NS * 4 * 64 * 64 * KS;
    static std::atomic<uint32_t> opted{0};            // per device ordinal (common.h: lds_opt_in)
    if (lds_opt_in(opted, reinterpret_cast<const void*>(gemm_nt_small_x3_kernel<NW, KS, NS>), smem) != PNP_OK) return PNP_ERR_HIP;
    hipLaunchKernelGGL((gemm_nt_small_x3_kernel<NW, KS, NS>), dim3(nbm * nbn), dim3(NW * 64), smem, s, g);
    return hipGetLastError() == hipSuccess ? PNP_OK : PNP_ERR_HIP;
}

static int launch_small_x3(const GemmArgs& g, hipStream_t s) {
    int nw = 8, ks = g.K % 64 == 0 ? 2 : 1;
    // many more workgroups than CUs (the N = 3072 launches at M = 875: 672): three co-resident workgroups per CU on 48 KB rings
    // (3 slots of 32-deep slabs) run the launch as ONE round instead of three rounds of one 128 KB workgroup per CU: 28.9 -> 25.2 us.
    // Not more than that, and nothing at 504 workgroups (21.2 against 21.4 us) or on the deep-K shapes (58 against 29 us): these
    // launches run at what a CU ingests from L2 (~20-27 B/clk, tools/micro/lds_dma_rate.hip), whoever is resident on it.
    int variant = (long)((g.M + 63) / 64) * (g.N / 64) > 600 && g.K <= 1024 ? 1 : 0;
#ifdef PNP_DEV
    if (getenv("PNP_SMALL_NW")) nw = atoi(getenv("PNP_SMALL_NW"));
    if (getenv("PNP_SMALL_KS") && atoi(getenv("PNP_SMALL_KS")) == 1) ks = 1;
    if (getenv("PNP_SMALL_VARIANT")) variant = atoi(getenv("PNP_SMALL_VARIANT"));
#endif
    if (variant == 1) return launch_small_x3_t<8, 1, 3>(g, s);
    if (variant == 2) return launch_small_x3_t<8, 2, 3>(g, s);          // 96 KB: one per CU, for comparison
    if (ks == 2) return nw == 4 ? launch_small_x3_t<4, 2>(g, s) : launch_small_x3_t<8, 2>(g, s);
    return nw == 4 ? launch_small_x3_t<4, 1>(g, s) : launch_small_x3_t<8, 1>(g, s);
}

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

    const int nbm = (g.M + BM - 1) / BM, nbn = g.N / BN;
    int bm, bn;
    tile_coords<8>(blockIdx.x, nbm, nbn, bm, bn);
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
        const RowCtx rc = row_ctx(g, m);
#pragma unroll
        for (int i = 0; i < TN; i++) {
            const int n = n0 + wn * WTN + i * 16 + q * 4;
            if (n >= g.Nvalid) continue;
            store_frag<T>(g, rc, acc[i][j], m, n, n + 3 < g.Nvalid);
        }
    }
}

template <typename T, int BM, int BN>
static int launch_cfg(const GemmArgs& g, hipStream_t s) {
    const int nbm = (g.M + BM - 1) / BM, nbn = g.N / BN;
    const size_t smem = 2 * (BM + BN) * 128;
    static std::atomic<uint32_t> opted{0};            // per device ordinal (common.h: lds_opt_in)
    if (lds_opt_in(opted, reinterpret_cast<const void*>(gemm_nt_kernel<T, BM, BN>), (int)smem) != PNP_OK) return PNP_ERR_HIP;
    hipLaunchKernelGGL((gemm_nt_kernel<T, BM, BN>), dim3(nbm * nbn), dim3(256), smem, s, g);
    return hipGetLastError() == hipSuccess ? PNP_OK : PNP_ERR_HIP;
}

static constexpr int kStampBlocks = 8192;
static unsigned long long*& stamp_buf() {
    static unsigned long long* p = nullptr;
    return p;
}
int gemm_read_stamps(unsigned long long* host_out, int max_blocks) {
    if (!stamp_buf()) return PNP_ERR_STATE;              // product builds never allocate it (see PNP_DEV above)
    const int n = max_blocks < kStampBlocks ? max_blocks : kStampBlocks;
    if (hipDeviceSynchronize() != hipSuccess) return PNP_ERR_HIP;
    return hipMemcpy(host_out, stamp_buf(), (size_t)n * 64, hipMemcpyDeviceToHost) == hipSuccess ? PNP_OK : PNP_ERR_HIP;
}

// Development knobs (timing ablations, forced tile variants, in-kernel clock stamps) exist only in builds made with
// `make DEV=1` (-DPNP_DEV); the product library reads no environment variable.
#ifdef PNP_DEV
static int dev_env(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}
#endif

// Host entry.  N is rounded up to the tile internally (loads clamp, stores mask on Nvalid).
int gemm_nt(int dtype_bf16, GemmArgs g, hipStream_t s) {
    if (g.M <= 0 || g.N <= 0 || g.K <= 0) return PNP_ERR_ARG;
    if (g.a_f32) {
        // split-bf16 with fp32 activations (text side): A is split by the kernel, B is a (hi, lo) bf16 pair; fp32 epilogues
        if (!g.B_lo || g.A_lo || g.K % 32 || (g.lda * 4) % 16 || (g.ldb * 2) % 16 || g.row_div || g.col_div || g.bias_on_rows ||
            (g.N & 3))
            return PNP_ERR_ARG;
        g.Nvalid = g.N;
        g.N = (g.N + 63) / 64 * 64;
        return launch_small_x3(g, s);
    }
    const bool x3 = g.A_lo || g.B_lo;                      // split-bf16 operands (bf16 pairs), fp32-facing epilogues
    if (x3) dtype_bf16 = 1;
    const int bk = dtype_bf16 ? 64 : 32;
    if (g.K % bk) return PNP_ERR_ARG;
    if ((g.lda * (dtype_bf16 ? 2 : 4)) % 16 || (g.ldb * (dtype_bf16 ? 2 : 4)) % 16) return PNP_ERR_ARG;
    g.Nvalid = g.N;
    int variant = 0;
#ifdef PNP_DEV
    variant = dev_env("PNP_GEMM_VARIANT", 0);              // 1 / 2 / 3 / 4: generic 256x128 / generic 256x256 / generic 128x128 / wide
    g.ablate = dev_env("PNP_GEMM_ABLATE", 0);
    if (getenv("PNP_GEMM_GM")) g.ablate = 100 + dev_env("PNP_GEMM_GM", 4);   // tile-order experiment (gemm_x3.hip)
    if (dev_env("PNP_GEMM_STAMPS", 0)) {
        if (!stamp_buf() && hipMalloc(&stamp_buf(), kStampBlocks * 64) != hipSuccess) return PNP_ERR_HIP;
        g.stamps = stamp_buf();
    }
#endif
    // small problems (text side: M = B*L rows) use 64x64 tiles to fill more CUs
    const long tiles128 = (long)((g.M + 127) / 128) * ((g.N + 127) / 128);
    const bool small = tiles128 < 192 && !x3;
    if (small) {
        // text side (M = B*L rows): 64 x 64 tiles, 4 waves of 32 x 32, 3-slot DMA ring -- the deep
        // prefetch matters more than tile efficiency for these latency-bound launches
        g.N = (g.N + 63) / 64 * 64;
#ifdef PNP_DEV
        if (dev_env("PNP_GEMM_SMALL", 0) == 1) return dtype_bf16 ? launch_cfg<bf16, 64, 64>(g, s) : launch_cfg<float, 64, 64>(g, s);
        const int small_ns = dev_env("PNP_GEMM_SMALL_NS", 3);
        if (dtype_bf16 && small_ns == 4) return launch_big<bf16, 64, 64, 32, 32, 4>(g, s);
        if (dtype_bf16 && small_ns == 6) return launch_big<bf16, 64, 64, 32, 32, 6>(g, s);
#endif
        return dtype_bf16 ? launch_big<bf16, 64, 64, 32, 32, 3>(g, s) : launch_big<float, 64, 64, 32, 32, 3>(g, s);
    }
    static GemmProfile never_on;                           // launches outside an engine (pnp_op_*) are not timed
    GemmProfile& pf = g.prof ? *g.prof : never_on;
    // every `period`-th launch of the family is bracketed (1 = all of them: ~2.5 us of stream serialisation per launch)
    const bool timed = pf.on && pf.used < GemmProfile::kMax && (pf.seq++ % pf.period) == 0;
    if (timed) {
        while (pf.created <= pf.used) {
            if (hipEventCreate(&pf.ev0[pf.created]) != hipSuccess || hipEventCreate(&pf.ev1[pf.created]) != hipSuccess)
                return PNP_ERR_HIP;
            pf.created++;
        }
        (void)hipEventRecord(pf.ev0[pf.used], s);
    }
    const double fl = 2.0 * g.M * (double)g.N * g.K;       // algorithmic FLOPs (the split-bf16 form issues 3x as MFMA work)
    g.N = (g.N + 127) / 128 * 128;
    // Tile choice (in-kernel clock stamps, tools/gemm_stamps.py on a DEV build):
    //   bf16 ViT-block epilogues (bias -> bf16 | bias+GELU -> bf16 | bias+residual -> f32), >= 128 tiles:
    //       gemm_nt_wide_kernel, 256 x 256, 32x32x16 MFMA; main loop ~2350 clk per 64-deep slab against
    //       2048 MFMA clk, epilogue staged through LDS (full-line stores)
    //   split-bf16 operands       : always gemm_nt_x3_kernel (gemm_x3.hip)
    //   other bf16 with K >= 2048 : generic 256 x 256 (16x16x32 MFMA, simple ring)
    //   otherwise                 : generic 128 x 128, two workgroups per CU
    int r;
    const bool big_k = g.K >= 2048 && g.Nvalid >= 512;
    const int wide = dtype_bf16 ? wide_epilogue_kind(g) : -1;
    const long tiles256 = (long)((g.M + 255) / 256) * ((g.Nvalid + 255) / 256);
    if (x3) {
        if (wide < 0) return PNP_ERR_ARG;
        g.N = (g.Nvalid + 255) / 256 * 256;
        r = launch_x3_wide(wide, g, s);
    } else if (wide >= 0 && (variant == 4 || (variant == 0 && tiles256 >= 128))) {
        g.N = (g.Nvalid + 255) / 256 * 256;
        r = wide == WIDE_BF16 ? launch_wide<WIDE_BF16>(g, s)
            : wide == WIDE_GELU_BF16 ? launch_wide<WIDE_GELU_BF16>(g, s)
            : wide == WIDE_RESID_F32 ? launch_wide<WIDE_RESID_F32>(g, s) : launch_wide<WIDE_TOKCOLS_BF16>(g, s);
    } else if (variant == 1) {
        r = dtype_bf16 ? launch_big<bf16, 256, 128, 64, 64, 3>(g, s) : launch_big<float, 256, 128, 64, 64, 3>(g, s);
    } else if (dtype_bf16 && (variant == 2 || (variant == 0 && big_k))) {
        g.N = (g.Nvalid + 255) / 256 * 256;
        r = launch_big<bf16, 256, 256, 128, 64, 2, false>(g, s);
    } else {
        r = dtype_bf16 ? launch_big<bf16, 128, 128, 64, 64, 2>(g, s) : launch_big<float, 128, 128, 64, 64, 2>(g, s);
    }
    if (timed) {
        (void)hipEventRecord(pf.ev1[pf.used], s);
        pf.used++;
        pf.launches++;
        pf.flops += fl;
    }
    return r;
}

}  // namespace pnp
