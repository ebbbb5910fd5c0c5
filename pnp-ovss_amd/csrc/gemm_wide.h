// Pieces shared by the two persistent 256 x 256 kernels: gemm_nt_wide_kernel (gemm.hip: bf16, 32x32x16 MFMA) and
// gemm_nt_x3_kernel (gemm_x3.hip: split-bf16, 16x16x32 MFMA).
#pragma once
#include "common.h"
#include "gemm.h"

namespace pnp {

// ---- tile rasterisation.  Blocks b and b+8 share an XCD (round-robin dispatch), so each XCD gets
// a contiguous run of tile ids; inside the run tiles are walked in groups of GM row-tiles x all
// column tiles, column-major inside the group, so the ~32 workgroups resident on one XCD cover a
// GM x (32/GM) patch: they share GM A-panels and 32/GM B-panels out of the 4 MB L2 instead of
// streaming 32 different A-panels from Infinity Cache / HBM.
template <int GM>
__device__ __forceinline__ void tile_coords(int bid, int nbm, int nbn, int& bm, int& bn) {
    const int nwg = nbm * nbn;
    const int qd = nwg >> 3, rm = nwg & 7, x = bid & 7, i = bid >> 3;
    const int id = (x < rm ? x * (qd + 1) : rm * (qd + 1) + (x - rm) * qd) + i;
    const int per_group = GM * nbn;
    const int rg = id / per_group, rem = id - rg * per_group;
    const int rows = (nbm - rg * GM) < GM ? (nbm - rg * GM) : GM;
    bn = rem / rows;
    bm = rg * GM + (rem - bn * rows);
}

// the same with the group height as a run-time value (DEV builds: the tile-order experiment of tools/gemm_window_traffic.sh)
__device__ __forceinline__ void tile_coords_rt(int gm, int bid, int nbm, int nbn, int& bm, int& bn) {
    const int nwg = nbm * nbn;
    const int qd = nwg >> 3, rm = nwg & 7, x = bid & 7, i = bid >> 3;
    const int id = (x < rm ? x * (qd + 1) : rm * (qd + 1) + (x - rm) * qd) + i;
    const int per_group = gm * nbn;
    const int rg = id / per_group, rem = id - rg * per_group;
    const int rows = (nbm - rg * gm) < gm ? (nbm - rg * gm) : gm;
    bn = rem / rows;
    bm = rg * gm + (rem - bn * rows);
}

#define PNP_WAIT_VM_LGKM(n) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(n) : "memory")
#define PNP_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

// compile-time epilogues of the wide kernels (see gemm.hip for the full description)
enum { WIDE_BF16 = 0, WIDE_GELU_BF16 = 1, WIDE_RESID_F32 = 2, WIDE_TOKCOLS_BF16 = 3, WIDE_BIAS_F32 = 4, WIDE_TOKCOLS_F32 = 5,
       WIDE_GELU_SPLIT = 6, WIDE_SPLIT = 7 };
constexpr int kWideStageRow = 68;                                  // floats per staged row (64 + 4 pad)
constexpr int kWideStageBytes = 8 * 32 * kWideStageRow * 4;        // 8 waves x 32 rows
constexpr int kWideSmem = 65536 + kWideStageBytes;                 // slot 0 | slot 1 overlaid by the staging area

// split-bf16 launches (gemm_x3.hip); epi = one of the fp32-facing WIDE_* kinds
int launch_x3_wide(int epi, const GemmArgs& g, hipStream_t s);

}  // namespace pnp
