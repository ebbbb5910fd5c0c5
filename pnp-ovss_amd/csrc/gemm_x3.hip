// Split-bf16 ("bf16x3") wide GEMM for gfx950:  C[M,N] = A[M,K] * B[N,K]^T with both operands given as (hi, lo) bf16 pairs
// (a = a_hi + a_lo), every product accumulated as a_hi.b_lo + a_lo.b_hi + a_hi.b_hi on the bf16 MFMA in fp32: an fp32-class
// result (2^-16 per product) at a third of the bf16 MFMA rate, ~5x the fp32 MFMA rate.  These are the launches the
// benchmarked mode spends most of its time in: every Linear of the ViT block (B/vit.py:93-117 qkv / proj, :45-51 fc1 / fc2)
// and the cross-attention K / V projections of all text layers (B/med.py:208-211) at M = B * N_img rows.
//
// Same frame as gemm_nt_wide_kernel (gemm.hip): PERSISTENT 256 x 256 tiles, one 512-thread workgroup per CU, 8 waves (2 x 4)
// of 128 x 64, operands streamed global -> LDS by `global_load_lds_dwordx4` into a two-slot ring of 32-deep k-slabs
// (slot = A_hi | A_lo | B_hi | B_lo, 256 rows x 64 bytes each), ONE s_barrier per slab, slab 0 of the next tile requested
// before the epilogue, epilogues staged through LDS so that every global access is a full line.
//
// What is different (round 4): the matrix instruction is v_mfma_f32_16x16x32_bf16 instead of 32x32x16.  At equal FLOPs and
// equal cycles per FLOP the chip holds a higher clock on the 16x16x32 form (MI355X_MICROARCH.md, DVFS give-back (7): 1.12-1.15x
// in a bare loop; a timing-only swap of the opcode in the 32x32 kernel ran 5-7 % faster on every ViT shape), but hipcc does
// not keep 4-register accumulators in place across a builtin MFMA chain (round 1: copies and spills made the port slower
// than what it replaced).  So the MFMAs of one 16-row m-tile -- 4 n-tiles x 3 products = 12 instructions -- are ONE
// `asm volatile` block whose accumulators are tied in/out operands ("+v"): in place by construction, issued back to back,
// and a fixed point of the schedule: the fragment reads and DMA pieces the compiler emits stay between the blocks where
// the source puts them.  The compiler still tracks every LDS read (its counted lgkmcnt waits in front of each block).
//   per slab and wave: 8 blocks (m-tiles) x 12 MFMAs; the A fragments (hi, lo) of m-tile j+3 are read behind block j into
//   a 4-slot rotation, the B fragments of the NEXT slab (4 n-tiles x (hi, lo)) behind the slab hand-over into the other of
//   two register sets; the hand-over (own reads done, own DMA pieces of slab t+1 landed, s_barrier) sits behind block 5; the
//   8 DMA pieces of slab t+2 are issued two per block behind blocks 5, 6 and blocks 0, 1 of the next slab.
//   registers: 128 accumulators + 64 B fragments + 32 A fragments + addresses.
// LDS image: 16-byte chunks swizzled chunk ^= (-(row >> 2)) & 3, so the 16 lanes a ds_read_b128 services together (4 rows of
// one chunk column and 8 + 4 rows of the next) hit 16 different 16-byte slots of the 256-byte bank row; applied to the
// per-lane SOURCE address of the DMA pieces and again on the fragment reads.
// Accumulator layout (operands swapped at the MFMA, D = Btile * Atile^T): lane (r = l & 15, q = l >> 4) of tile (jm, in)
// owns row m = 16 jm + r and the 4 consecutive columns n = 16 in + 4 q .. + 3.
//
// Stream-K tail (round 6, the SK instantiations): tiles come in rounds of one per CU, and the last round is rarely full
// (732 / 976 / 244 tiles on 256 CUs at the bench batch; 292 / 876 / 1168 at 8 images of 768^2, where a round of 36 tiles costs
// a whole tile time).  With SK the tiles of the whole rounds are walked as before and the TAIL tiles are cut along K into
// 64-deep units (slab pairs) that are dealt over ALL workgroups in unit order, `sk_upw` units each: a workgroup's run covers
// the end of one tile and / or the start of the next.  A part that does not reach its tile's end leaves its 256 x 256 fp32
// partial sums in the workgroup's own 256 KB slot of a workspace (accumulator layout, 16-byte write-through stores, one
// flag word per workgroup); the workgroup that holds the END of a tile adds the partial sums of the workgroups before it
// in a fixed order (nearest first) and runs the tile's epilogue -- no atomics, results independent of timing.  A workgroup
// runs its producing part BEFORE its owning part and a producer never waits, so an owner only ever waits for workgroups
// of its own launch with lower block ids.  Those were handed to the dispatcher before it, but each XCD deals its share of the
// grid at its own pace, so "lower id" means "started, or waiting for a CU that something else holds": progress then depends on
// that something not being another launch of this kind waiting the other way round.  The launcher therefore keeps ONE
// stream-K launch in flight per device (an event chain across streams, launch_x3), beside which only kernels that never
// spin can hold CUs; and every spin is bounded all the same, a give-up recorded in a word the host reads
// (pnp_streamk_status) instead of a hung device.  The hand-off is the guide's
// write-through form (cdna_hip_programming.md Guideline 16, R1): every payload store `sc1`, every storing wave drains
// vmcnt, one lane stores the flag `sc1`; the consumer polls that word with `sc1` loads, joins a workgroup barrier, and every
// load of the payload is an `sc1` buffer load (no L1 copy can be stale, no agent-scope fence is needed).  The owner resets
// the flag it consumed, so the state is clean for the next launch (and under graph replay).
#include <stdlib.h>

#include <mutex>
#include <type_traits>

#include "common.h"
#include "gemm.h"
#include "gemm_wide.h"

// epilogue stores: -DPNP_EPI_NT builds mark them non-temporal (A/B experiment: do the output tiles evict the operand panels from L2?)
#ifdef PNP_EPI_NT
#define PNP_EPI_STORE(ptr, val) __builtin_nontemporal_store((val), (ptr))
#else
#define PNP_EPI_STORE(ptr, val) (*(ptr) = (val))
#endif

namespace pnp {

typedef __attribute__((ext_vector_type(4))) uint32_t frag16;      // one lane's 8 bf16 of a 16x16x32 operand

// 6 MFMAs of one m-tile and two n-tiles: c[i] += bl[i].ah + bh[i].al + bh[i].ah (small terms first), accumulators in place
__device__ __forceinline__ void x3_half(f32x4& c0, f32x4& c1, const frag16& ah, const frag16& al, const frag16& bh0, const frag16& bh1,
                                        const frag16& bl0, const frag16& bl1) {
    asm volatile(
        "v_mfma_f32_16x16x32_bf16 %0, %6, %2, %0\n\t"
        "v_mfma_f32_16x16x32_bf16 %1, %7, %2, %1\n\t"
        "v_mfma_f32_16x16x32_bf16 %0, %4, %3, %0\n\t"
        "v_mfma_f32_16x16x32_bf16 %1, %5, %3, %1\n\t"
        "v_mfma_f32_16x16x32_bf16 %0, %4, %2, %0\n\t"
        "v_mfma_f32_16x16x32_bf16 %1, %5, %2, %1"
        : "+v"(c0), "+v"(c1)
        : "v"(ah), "v"(al), "v"(bh0), "v"(bh1), "v"(bl0), "v"(bl1));
}

// EPI: WIDE_RESID_F32 (+bias +residual -> fp32) | WIDE_BIAS_F32 (+bias -> fp32) | WIDE_TOKCOLS_F32 (per-row bias, token columns
// remapped to per-image padded columns -> fp32) | WIDE_GELU_SPLIT (+bias, erf-GELU -> (hi, lo) bf16 pair) | WIDE_SPLIT (+bias -> pair)
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((address_space(1))) uint32_t gu32;
#ifdef PNP_SK_PLAIN            // experiment: plain payload + agent-scope release / acquire fences instead of write-through stores
#define PNP_SK_AUX 0
#else
#define PNP_SK_AUX 16          // sc1
#endif

template <int EPI, bool SK>
__global__ __launch_bounds__(512) void gemm_nt_x3_kernel(const GemmArgs g) {
    constexpr int BM = 256, BN = 256, SLOT = 65536, ARR = 16384;
    constexpr int JM = 8, IN = 4;                   // 16 x 16 tiles per wave: 128 (m) x 64 (n)
    constexpr int NDMA = 8;
    constexpr int kDrain = (EPI == WIDE_GELU_SPLIT || EPI == WIDE_SPLIT) ? 32 : 24;
    constexpr int SROW = kWideStageRow;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int wm = wave >> 2, wn = wave & 3;
    const int nbm = (g.M + BM - 1) / BM, nbn = g.N / BN;
    const int ntiles = nbm * nbn;
    const int nk = g.K / 32;                        // even, >= 2 (gemm_nt: K % 64 == 0)
    const char* Ab = reinterpret_cast<const char*>(g.A);
    const char* Bb = reinterpret_cast<const char*>(g.B);
    const char* Alo = reinterpret_cast<const char*>(g.A_lo);
    const char* Blo = reinterpret_cast<const char*>(g.B_lo);

    // DMA piece = 16 rows of ONE array (uniform base, per-lane 32-bit offset): a wave's hi and lo piece of the same rows
    // share an offset.  Lane l lands at row l >> 2, physical chunk l & 3 of the piece.
    uint32_t soff[4];
    auto set_tile = [&](int tile, int& m0, int& n0, int lane) {
        int bm, bn;
#ifdef PNP_DEV
        if (g.ablate >= 100) {                               // PNP_GEMM_GM: group height of the tile order (0 = plain row-major ids)
            if (g.ablate == 100) {
                bm = tile / nbn;
                bn = tile - bm * nbn;
            } else {
                tile_coords_rt(g.ablate - 100, tile, nbm, nbn, bm, bn);
            }
        } else
#endif
        tile_coords<4>(tile, nbm, nbn, bm, bn);
        m0 = bm * BM;
        n0 = bn * BN;
        const uint32_t lda_b = (uint32_t)g.lda * 2, ldb_b = (uint32_t)g.ldb * 2;
        const int c = (lane & 3) ^ ((-(lane >> 4)) & 3);     // logical chunk this lane's 16 bytes land as (row & 15 = lane >> 2)
#pragma unroll
        for (int i = 0; i < 2; i++) {                        // two 16-row groups per wave and operand
            const int row = (wave * 2 + i) * 16 + (lane >> 2);
            int gr = m0 + row;
            gr = gr < g.M ? gr : g.M - 1;
            soff[i] = (uint32_t)gr * lda_b + c * 16;
            gr = n0 + row;
            gr = gr < g.Nvalid ? gr : g.Nvalid - 1;
            soff[2 + i] = (uint32_t)gr * ldb_b + c * 16;
        }
    };
    // on = non-zero / zero (wave-uniform): a piece issued under EXEC = 0 moves nothing, so the slabs at either end of the
    // k range run the SAME instruction stream as the steady state (no peeled copies of the slab body whose register
    // assignment the allocator then has to reconcile with the loop's through scratch memory)
    auto issue_one = [&](int kt, int i, uint32_t on) {  // piece i: operand = i >> 2, row group = (i >> 1) & 1, array (hi | lo) = i & 1
#ifdef PNP_X3_ABLATE                                // timing-only builds (results are garbage; tools/gemm_x3_ab.py): 1 = no steady-state
        if (kt >= 2) return;                        // DMA, 3 = also no slab barrier, 4 = also no fragment reads.  Compile-time:
#endif                                              // a run-time test inside the slab loop changes the schedule it is meant to time
        const int op = i >> 2, rg = (i >> 1) & 1, lo = i & 1;
        const char* base = op ? (lo ? Blo : Bb) : (lo ? Alo : Ab);
        const int d = (kt & 1) * SLOT + op * (2 * ARR) + lo * ARR + (wave * 2 + rg) * 1024;
        // LDS-DMA as inline asm, not as __builtin_amdgcn_global_load_lds: hipcc treats the builtin like a FLAT access that
        // may touch both memories, and while one is pending every wait it inserts for an LDS read is forced to lgkmcnt(0)
        // (SIInsertWaitcnts "pending flat") -- i.e. a fragment read issued right before an MFMA block would be waited for
        // there.  Hidden from the compiler, its ds_read waits are counted (only the fragments a block consumes), and the
        // landing of the pieces is covered by the explicit vmcnt(0) in front of each slab's barrier.
        const uint32_t lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(smem + d);
        // the per-lane source offset is formed inside the block (one dead temporary) so that the compiler cannot hoist the
        // eight adds of a slab to the top of the loop and hold their results in registers the accumulators need
        uint32_t voff;
        uint64_t saved;
        asm volatile(
            "s_mov_b64 %1, exec\n\t"
            "s_cmp_lg_u32 %6, 0\n\t"
            "s_cselect_b64 exec, exec, 0\n\t"
            "v_add_u32 %0, %3, %2\n\t"
            "s_mov_b32 m0, %5\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %0, %4\n\t"
            "s_mov_b64 exec, %1"
            : "=&v"(voff), "=&s"(saved)
            : "v"(soff[op * 2 + rg]), "s"((uint32_t)kt * 64), "s"(base), "s"(lds), "s"(on)
            : "memory", "scc");        // m0: hipcc treats it as RESERVED -- it cannot be named in a clobber list ("inline asm
                                       // clobber list contains reserved registers", the entry is ignored) and the compiler itself
                                       // writes it only for its own LDS-DMA / GPR-indexing / LDS-param instructions, none of which this
                                       // kernel contains: tests/test_cabi_cpu.py::test_m0_is_written_only_by_the_asm_lds_dma checks the
                                       // disassembly of the shipped code object for exactly that
    };
    auto stamp_sk = [&](int slot) {                 // stream-K phases of this workgroup, in the stamp rows behind the grid's own
        if (g.stamps && tid == 0) {
            g.stamps[((size_t)gridDim.x + blockIdx.x) * 8 + slot] = __builtin_readcyclecounter();
            g.stamps[((size_t)gridDim.x + blockIdx.x) * 8 + 4 + slot] = wall_clock64();
        }
    };
    auto stamp = [&](int slot) {
        if (g.stamps && tid == 0) {
            g.stamps[(size_t)blockIdx.x * 8 + slot] = __builtin_readcyclecounter();
            g.stamps[(size_t)blockIdx.x * 8 + 4 + slot] = wall_clock64();
        }
    };

    // fragment addresses: row = tile row + r, logical chunk q; the swizzle term depends on r only (tile rows are multiples of 16)
    const int fo = r * 64 + ((q ^ ((-(r >> 2)) & 3)) << 4);
    // one base register per (operand, ring slot); everything else of a fragment address fits the 16-bit offset field of
    // ds_read_b128.  The slot-1 bases are made opaque to the compiler: folded into constants they exceed that field and it
    // materialises a separate address register per read (20 registers that the accumulators need)
    int fa_off[2], fb_off[2];
    fa_off[0] = wm * (128 * 64) + fo;                                   // A_hi (lo: + ARR)
    fb_off[0] = 2 * ARR + wn * (64 * 64) + fo;                          // B_hi
    fa_off[1] = fa_off[0] + SLOT;
    fb_off[1] = fb_off[0] + SLOT;
    asm volatile("" : "+v"(fa_off[1]), "+v"(fb_off[1]));
    float* const stg = reinterpret_cast<float*>(smem + SLOT) + wave * (32 * SROW);

    // ---- work list of this workgroup (everything wave-uniform).  Whole rounds: tiles blockIdx.x + i * gridDim.x below `full`
    // (= every tile when the stream-K tail is off).  Tail (SK): units [bid * upw, (bid + 1) * upw) of the tail tiles' slab
    // pairs; P = the part of that run that does not reach its tile's end (partial sums -> workspace), O = the part that does
    // (adds the parts of the workgroups before it, then the epilogue).  Order: whole tiles, P, O
    // Nothing of the list is kept in registers across the tile loop (this kernel has no scalar register to spare: what does not fit
    // is parked in vector registers, and those belong to the accumulators): the item is recomputed from the kernel arguments
    const int bid = blockIdx.x;
    enum { kWhole = 0, kProduce = 1, kOwn = 2 };
    auto get_work = [&](int wi, int& t, int& k0, int& k1, int& kind) {
        const int grid = gridDim.x;
        const int full = SK ? g.sk_full : ntiles;
        t = bid + wi * grid;
        if (t < full) {
            k0 = 0;
            k1 = nk;
            kind = kWhole;
            return true;
        }
        if constexpr (SK) {
            const int j = wi - g.sk_full / grid;             // every workgroup walks the same number of whole tiles
            const int nk2 = nk >> 1;
            const int T = (ntiles - full) * nk2;
            const int u0 = bid * g.sk_upw;
            const int u1 = u0 + g.sk_upw < T ? u0 + g.sk_upw : T;
            if (u0 >= u1 || j > 1) return false;
            const int ta = u0 / nk2, bnd = (ta + 1) * nk2;
            const bool owns = u1 >= bnd, two = u1 > bnd;     // reaches the end of tile ta | goes on into tile ta + 1
            if (j == 0 && (two || !owns)) {                  // the producing part first
                t = full + ta + (two ? 1 : 0);
                k0 = two ? 0 : 2 * (u0 - ta * nk2);
                k1 = 2 * (u1 - (two ? bnd : ta * nk2));
                kind = kProduce;
                return true;
            }
            if (owns && (j == 0 || two)) {
                t = full + ta;
                k0 = 2 * (u0 - ta * nk2);
                k1 = nk;
                kind = k0 ? kOwn : kWhole;
                return true;
            }
        }
        return false;
    };
    int wi = 0, tile, k0, k1, kind;
    if (!get_work(0, tile, k0, k1, kind)) return;
    int m0, n0;
    bool drain32 = false;                           // the previous tile's epilogue left >= kDrain stores behind this tile's slab 0
    stamp(0);
    set_tile(tile, m0, n0, lane);
#pragma unroll
    for (int i = 0; i < NDMA; i++) issue_one(k0, i, 1u);

    for (;;) {
        f32x4 acc[JM][IN];
#pragma unroll
        for (int j = 0; j < JM; j++)
#pragma unroll
            for (int i = 0; i < IN; i++) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

        frag16 ah[4], al[4], bh[2][IN], bl[2][IN];
        auto rd_a = [&](int slot, int s, int jm) {
#if defined(PNP_X3_ABLATE) && PNP_X3_ABLATE >= 4
            return;
#endif
            const char* p = smem + fa_off[s] + jm * 1024;
            ah[slot] = *reinterpret_cast<const frag16*>(p);
            al[slot] = *reinterpret_cast<const frag16*>(p + ARR);
        };
        auto rd_b = [&](int set, int s, int in) {
#if defined(PNP_X3_ABLATE) && PNP_X3_ABLATE >= 4
            return;
#endif
            const char* p = smem + fb_off[s] + in * 1024;
            bh[set][in] = *reinterpret_cast<const frag16*>(p);
            bl[set][in] = *reinterpret_cast<const frag16*>(p + ARR);
        };

        // slab 0 of this tile is in flight (requested during the previous tile's epilogue, or above).  vmcnt counts in issue
        // order, so after a full-tile epilogue that issued >= kDrain stores behind the DMA pieces, "at most kDrain outstanding"
        // already means the slab has landed and the rest of those stores drain under the first blocks of this tile.  The
        // pieces are requested behind quarter 0 of the fp32 epilogues (24 stores follow) and in front of all 64 stores of the
        // split ones
        if (drain32) PNP_WAIT_VM(kDrain);
        else PNP_WAIT_VM(0);
        __builtin_amdgcn_s_barrier();               // slab 0 complete; staging area (slot 1) no longer read
#pragma unroll
        for (int i = 0; i < NDMA; i++) issue_one(k0 + 1, i, 1u);      // k0 is even and every part is at least one slab pair
        rd_a(0, 0, 0);
        rd_a(1, 0, 1);
        rd_a(2, 0, 2);
#pragma unroll
        for (int i = 0; i < IN; i++) rd_b(0, 0, i);

        // one slab; S = its ring slot / B register set (compile-time).  16 half blocks (j, h) = 6 MFMAs on m-tile j, n-tiles
        // 2h, 2h+1, each followed by a small cluster:
        //   (j, 0), j <= 4 : A(j + 3)                       (0,0) (0,1) (1,0) (1,1): + pieces 4..7 of slab kt + 1 ("head")
        //   (5, 0)         : hand-over, A'(0), piece 0 of slab kt + 2 ("tail")
        //   (5, 1)         : B'(0), piece 1      (6,0): A'(1) B'(1), piece 2      (6,1): B'(2), piece 3      (7,0): A'(2) B'(3)
        // i.e. at most ONE DMA piece per cluster with six MFMAs of the wave's own between two pieces: a piece holds its wave
        // for 100+ cycles at issue, and only the SIMD partner's MFMAs can use the matrix pipe meanwhile -- two pieces in a row
        // outlast the partner's half block.  Every read is at least three half blocks old when its consumer starts (the
        // compiler's waits are counted); every piece has seven half blocks or more to land.  Slab 1 is requested whole in the
        // prologue (no head in slab 0); behind the last slab nothing is requested (masks) and the hand-over and the reads of
        // the "next" slab run on whatever the other slot holds: harmless, and the same code for every slab.
        auto slab = [&](int kt, auto par) {
            constexpr int S = decltype(par)::value;
            const uint32_t head = __builtin_amdgcn_readfirstlane((kt >= k0 + 1 && kt + 1 < k1) ? 1 : 0);
            const uint32_t tail = __builtin_amdgcn_readfirstlane((kt + 2 < k1) ? 1 : 0);
#pragma unroll
            for (int j = 0; j < JM; j++) {
                x3_half(acc[j][0], acc[j][1], ah[j & 3], al[j & 3], bh[S][0], bh[S][1], bl[S][0], bl[S][1]);
                if (j + 3 < JM) {
                    rd_a((j + 3) & 3, S, j + 3);
                    if (j < 2) issue_one(kt + 1, 4 + 2 * j, head);
                } else {
                    if (j == JM - 3) {
                        // own reads of slab kt complete, own pieces of slab kt+1 landed (a builtin, not asm: the compiler's
                        // scoreboard sees it and adds no wait of its own behind the reads that follow)
#if !defined(PNP_X3_ABLATE) || PNP_X3_ABLATE < 3
                        __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0) lgkmcnt(0)
                        __builtin_amdgcn_s_barrier();         // slab kt+1 complete; nobody reads slab kt's slot any more
#endif
                    }
                    rd_a((j + 3) & 3, S ^ 1, j + 3 - JM);
                    if (j == JM - 2) rd_b(S ^ 1, S ^ 1, 1);
                    if (j == JM - 1) rd_b(S ^ 1, S ^ 1, 3);
                    if (j == JM - 3) issue_one(kt + 2, 0, tail);
                    if (j == JM - 2) issue_one(kt + 2, 2, tail);
                }
                x3_half(acc[j][2], acc[j][3], ah[j & 3], al[j & 3], bh[S][2], bh[S][3], bl[S][2], bl[S][3]);
                if (j < 2) issue_one(kt + 1, 5 + 2 * j, head);
                if (j == JM - 3) {
                    rd_b(S ^ 1, S ^ 1, 0);
                    issue_one(kt + 2, 1, tail);
                }
                if (j == JM - 2) {
                    rd_b(S ^ 1, S ^ 1, 2);
                    issue_one(kt + 2, 3, tail);
                }
            }
        };
        constexpr std::integral_constant<int, 0> s0{};
        constexpr std::integral_constant<int, 1> s1{};
        for (int kt = k0; kt < k1; kt += 2) {       // k0, k1 even
            slab(kt, s0);
            slab(kt + 1, s1);
        }
        // the compiler does not see the MFMAs inside the asm blocks: cover the matrix-pipe -> VALU read hazard of the last ones
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        if (wi == 0) stamp(2);

        __syncthreads();                            // every wave is done reading the last slab: both slots are free
        int ntile, nk0, nk1, nkind;
        const bool have_next = get_work(wi + 1, ntile, nk0, nk1, nkind);
        // slab 0 of the next part flies during this one's epilogue (see request_next below for where it is issued)
        auto request_next = [&]() {
            if (have_next) {
                set_tile(ntile, m0, n0, lane);
#pragma unroll
                for (int i = 0; i < NDMA; i++) issue_one(nk0, i, 1u);
            }
        };
        if constexpr (SK) {
            if (kind == kProduce) {
                // partial sums of this part -> own workspace slot, accumulator layout: store (jm, in) of lane t at
                // float4 index (jm * 4 + in) * 512 + t, i.e. every store instruction writes 1 KB per wave, contiguous
                stamp_sk(0);
                request_next();
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(g.sk_part + (size_t)bid * (BM * BN), 0, BM * BN * 4, 0x00020000);
#if !defined(PNP_SK_ABLATE) || PNP_SK_ABLATE < 1 || PNP_SK_ABLATE == 3     // timing-only builds (tools/gemm_x3_streamk.py --lib): 1 = no
                {                                     // partial-tile traffic, 2 = also no flag wait, 3 = stores only, 4 = loads only
                    // ONE lane-offset register walks the 32 pieces (8 KB apart): offsets folded into 32 constants would cost 32 scalar
                    // registers this kernel does not have (the allocator then parks accumulators in scratch memory)
                    int voff = tid * 16;
#pragma unroll
                    for (int j = 0; j < JM; j++)
#pragma unroll
                        for (int i = 0; i < IN; i++) {
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[j][i]), rs, voff, 0, PNP_SK_AUX);   // aux 16 = sc1
                            voff += 8192;
                            asm volatile("" : "+v"(voff));
                        }
                }
#endif
#ifdef PNP_SK_PLAIN
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
#endif
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every storing wave drains before the flag is raised
                __syncthreads();
                if (tid == 0) __hip_atomic_store((gu32*)(g.sk_flag + bid), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                stamp_sk(1);
            }
            if (kind == kOwn) {
                const int np = ((k0 >> 1) + g.sk_upw - 1) / g.sk_upw;       // workgroups bid - 1 .. bid - np hold the parts before k0
                stamp_sk(2);
                for (int pj = 1; pj <= np; pj++) {
                    const int src = bid - pj;
                    if (wave == 0) {
                        gu32* const fl = (gu32*)(g.sk_flag + src);
                        uint32_t spins = 0;
#if defined(PNP_SK_ABLATE) && PNP_SK_ABLATE >= 2
                        while (false) {
#else
                        while (__hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1u) {
#endif
                            __builtin_amdgcn_s_sleep(8);
                            if (++spins > (1u << 22)) {                      // ~1 s: give up loudly instead of hanging the device
                                if (lane == 0) __hip_atomic_store((gu32*)g.sk_tmo, 1u + (uint32_t)bid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                break;
                            }
                        }
                        if (lane == 0) __hip_atomic_store(fl, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // clean for the next launch
                    }
                    __syncthreads();
                    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(g.sk_part + (size_t)src * (BM * BN), 0, BM * BN * 4, 0x00020000);
#ifdef PNP_SK_PLAIN
                    if (tid == 0) {
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                    __syncthreads();
#endif
#if !defined(PNP_SK_ABLATE) || PNP_SK_ABLATE < 1 || PNP_SK_ABLATE == 4
                    int voff = tid * 16;
#pragma unroll
                    for (int grp = 0; grp < 4; grp++) {
                        u32x4 t[8];
#pragma unroll
                        for (int x = 0; x < 8; x++) {
                            t[x] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, PNP_SK_AUX);
                            voff += 8192;
                            asm volatile("" : "+v"(voff));
                        }
#pragma unroll
                        for (int x = 0; x < 8; x++) acc[(grp * 8 + x) / IN][(grp * 8 + x) % IN] += __builtin_bit_cast(f32x4, t[x]);
                    }
#endif
                }
                stamp_sk(3);
            }
        }
        bool next_drain = false;
        if (!SK || kind != kProduce) {              // (one exit of the tile body for both kinds of part: a `continue` out of the middle
                                                    // left the allocator two loop edges to reconcile, through scratch memory)
        // everything the epilogue derives from the lane id is RE-derived here from an opaque copy of the thread id: computed
        // once in front of the k loop (where the compiler would otherwise put it) these ~15 values are live across a loop that
        // has no register to spare, and are spilled around it -- with scratch reloads whose vmcnt waits also wait for the DMA
        // pieces in flight
        int etid = tid;
        asm volatile("" : "+v"(etid));
        const int lane = etid & 63, r = lane & 15, q = lane >> 4;
        const int em0 = m0, en0 = n0;
        // epilogue operands are requested BEFORE the next tile's first slab: vmcnt is one in-order counter, a wait for a
        // load issued behind the DMA pieces would also wait for those
        const int n = en0 + wn * 64 + (lane & 15) * 4;          // this lane's 4 output columns on the way out (same for every row)
        const bool nv = n < g.Nvalid;                           // N is a multiple of 4 on the row-major epilogues
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        f32x4 rv[8];
        f32x4 bacc[IN];                             // split epilogues: bias in the accumulator layout
        if constexpr (EPI == WIDE_GELU_SPLIT || EPI == WIDE_SPLIT) {
#pragma unroll
            for (int i = 0; i < IN; i++) {
                const int nn = en0 + wn * 64 + i * 16 + q * 4;
                bacc[i] = (g.bias && nn < g.Nvalid) ? *reinterpret_cast<const f32x4*>(g.bias + nn) : bv;
            }
        }
        if constexpr (EPI == WIDE_BIAS_F32 || EPI == WIDE_RESID_F32) {
            if (g.bias && nv) bv = *reinterpret_cast<const f32x4*>(g.bias + n);
        }
        if constexpr (EPI == WIDE_RESID_F32) {
            const int mrow0 = em0 + wm * 128 + (lane >> 4);
#pragma unroll
            for (int it = 0; it < 8; it++) {
                const int m = mrow0 + it * 4;
                rv[it] = (m < g.M && nv) ? *reinterpret_cast<const f32x4*>(g.resid + (size_t)m * g.ldr + n) : bv;
            }
        }
        float brow[JM] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};       // TOKCOLS: bias of this lane's row in each 16-row tile
        if constexpr (EPI == WIDE_TOKCOLS_F32) {
#pragma unroll
            for (int j = 0; j < JM; j++) {
                const int m = em0 + wm * 128 + j * 16 + r;
                if (g.bias && m < g.M) brow[j] = g.bias[m];
            }
        }
        // slab 0 of the next tile flies during this tile's epilogue.  It is requested right BEHIND the first use of the operands
        // loaded above (bias, first residual rows): vmcnt is one in-order counter and the compiler does not see the pieces (asm),
        // so its wait for those loads would count the pieces as landed-before -- issued in front of that wait they would be
        // waited for with it (~3 us per tile measured); loads issued later wait for the pieces anyway, by then long landed

        if constexpr (EPI == WIDE_TOKCOLS_F32) {
            // 32-row quarters staged as fp32 with the row bias added in the accumulator layout; on the way out a lane owns a
            // PAIR of token columns (8-byte stores, 256 contiguous bytes per row) when col_div is even -- pairs then never
            // straddle an image -- else single tokens
            const int l32 = lane & 31, hi = lane >> 5;
            const bool pairs = (g.col_div & 1) == 0;
            const int tl0 = pairs ? 2 * l32 : lane;
            const int tok = en0 + wn * 64 + tl0;
            size_t ocol = tok;
            if (g.col_div > 0) {
                const int b = tok / g.col_div;
                const int tl = tok - b * g.col_div;
                ocol = (size_t)b * g.col_pad + tl;
            }
            float* const ocolp = g.out_f32 + ocol;
            const bool tv = tok < g.Nvalid;
#pragma unroll
            for (int qd = 0; qd < 4; qd++) {
                const int mbase = em0 + wm * 128 + qd * 32;
#pragma unroll
                for (int j2 = 0; j2 < 2; j2++)
#pragma unroll
                    for (int i = 0; i < IN; i++) {
                        const float br = brow[qd * 2 + j2];
                        const f32x4 v = acc[qd * 2 + j2][i] + br;
                        *reinterpret_cast<f32x4*>(stg + (j2 * 16 + r) * SROW + i * 16 + q * 4) = v;
                    }
                if (qd == 0) request_next();
                if (pairs) {
                    f32x2 sv[16];
#pragma unroll
                    for (int it = 0; it < 16; it++) sv[it] = *reinterpret_cast<const f32x2*>(stg + (it * 2 + hi) * SROW + tl0);
#pragma unroll
                    for (int it = 0; it < 16; it++) {
                        const int m = mbase + it * 2 + hi;
                        if (tv && m < g.M) *reinterpret_cast<f32x2*>(ocolp + (size_t)m * g.ldo) = sv[it];
                    }
                } else {
#pragma unroll 8
                    for (int row = 0; row < 32; row++) {
                        const float v = stg[row * SROW + lane];
                        if (tv && mbase + row < g.M) ocolp[(size_t)(mbase + row) * g.ldo] = v;
                    }
                }
            }
        } else if constexpr (EPI == WIDE_GELU_SPLIT || EPI == WIDE_SPLIT) {
            // per 64-row half: bias (+ erf-form GELU, |erf error| <= 1.5e-7) in the accumulator layout, then two passes over a
            // bf16 staging of the half: hi = bf16(v) -> out_t, lo = bf16(v - hi) -> out_lo (the pair carries 16 significant
            // bits of v).  Half-outer order: the stores of half 0 drain while the GELU of half 1 runs on the VALU.
            constexpr int HROW = 68;                // bf16 per staged row (64 + 4 pad)
            bf16* const stgh = reinterpret_cast<bf16*>(smem + SLOT) + wave * (64 * HROW);
            const bool full = (em0 + BM <= g.M) && (en0 + BN <= g.Nvalid);
#pragma unroll
            for (int half = 0; half < 2; half++) {
#pragma unroll
                for (int j4 = 0; j4 < 4; j4++)
#pragma unroll
                    for (int i = 0; i < IN; i++)
#pragma unroll
                        for (int e = 0; e < 4; e += 2) {
                            f32x4& a = acc[half * 4 + j4][i];
                            pnp_f32x2 v = {a[e] + bacc[i][e], a[e + 1] + bacc[i][e + 1]};
                            if constexpr (EPI == WIDE_GELU_SPLIT) v = gelu_erf_fast2(v);
                            a[e] = v[0];
                            a[e + 1] = v[1];
                        }
                if (half == 0) request_next();
#pragma unroll
                for (int part = 0; part < 2; part++) {
                    bf16* const obase = reinterpret_cast<bf16*>(part ? g.out_lo : g.out_t) + (size_t)(em0 + wm * 128 + (lane >> 4)) * g.ldo_t + n;
#pragma unroll
                    for (int j4 = 0; j4 < 4; j4++)
#pragma unroll
                        for (int i = 0; i < IN; i++) {
                            const f32x4& a = acc[half * 4 + j4][i];
                            bf16x4 pk;
#pragma unroll
                            for (int e = 0; e < 4; e++) {
                                const float v = a[e];
                                const bf16 h = (bf16)v;
                                pk[e] = part ? (bf16)(v - (float)h) : h;
                            }
                            *reinterpret_cast<bf16x4*>(stgh + (j4 * 16 + r) * HROW + i * 16 + q * 4) = pk;
                        }
                    bf16x4 sv[16];
#pragma unroll
                    for (int it = 0; it < 16; it++)
                        sv[it] = *reinterpret_cast<const bf16x4*>(stgh + (it * 4 + (lane >> 4)) * HROW + (lane & 15) * 4);
#pragma unroll
                    for (int it = 0; it < 16; it++) {
                        const int m = em0 + wm * 128 + half * 64 + it * 4 + (lane >> 4);
                        if (full || (m < g.M && nv)) PNP_EPI_STORE(reinterpret_cast<bf16x4*>(obase + (size_t)(half * 64 + it * 4) * g.ldo_t), sv[it]);
                    }
                }
            }
        } else {
            // fp32 residual / bias epilogue, 32-row quarters staged as fp32.  The residual rows of quarter q+1 are requested
            // before the stores of quarter q (two register sets), so each wait has a whole quarter of work in front of it
            // and never sits behind a store.
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
                for (int j2 = 0; j2 < 2; j2++)
#pragma unroll
                    for (int i = 0; i < IN; i++)
                        *reinterpret_cast<f32x4*>(stg + (j2 * 16 + r) * SROW + i * 16 + q * 4) = acc[qd * 2 + j2][i];
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
                    if (full || (m < g.M && nv)) PNP_EPI_STORE(reinterpret_cast<f32x4*>(obase + (size_t)(qd * 32 + it * 4) * g.ldo), v);
                }
                if (qd == 0) request_next();
            }
        }
        // every lane of a full tile executes all of the epilogue's stores (8 per quarter and wave, 64 for the split pair)
        next_drain = (EPI != WIDE_TOKCOLS_F32) && (em0 + BM <= g.M) && (en0 + BN <= g.Nvalid);
        }
        if (wi == 0) stamp(1);                     // diagnostics: first tile's epilogue done (stores issued)
        if (!have_next) break;
        wi++;
        tile = ntile;
        k0 = nk0;
        k1 = nk1;
        kind = nkind;
        drain32 = next_drain;
    }
    if (g.stamps) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(3);                                   // whole workgroup (all its tiles) done
    }
}

// Stream-K tail policy, in units of one slab pair of one workgroup (4.2 us at the clock these launches hold).  Measured with the
// in-kernel stamps (tools/gemm_x3_streamk_stamps.py, profiles/r06_streamk_stamps.txt) and per launch shape
// (tools/gemm_x3_streamk.py, profiles/r06_streamk.txt):
//   whole tiles : the tail round costs a tile, nk2 pairs + ~2 of epilogue -- x 0.82 when the tail holds under a quarter of the CUs
//                 (the few workgroups of such a round run beside idle CUs, at a higher rate: 58 against 75 us at K = 1024);
//   stream-K    : 1.05 x upw pairs in two parts, each with its own pipeline fill (~2 pairs each); the partial tile stored and
//                 the flag raised: 3-7 us when up to ~64 tiles are split, 19 us when most workgroups store 256 KB at the same time
//                 beside the others' operand streams (220 producers of the 732-tile launch); ~4 us per partial tile an owner
//                 adds (np of them, one after the other); the epilogue.
// So the hand-off costs 8-12 pairs = 35-50 us: it pays for a SMALL tail of DEEP tiles (8 x 2305 rows, fc2: 292 tiles of K = 4096,
// 501 -> 382 us; 12 x 442 rows: 84 tiles, 202 -> 146 us), is worth +-5 % for small tails at K = 1024 (left off), and loses 10-35 us
// on the bench batch's launches (732 / 976 / 244 tiles), where a whole tile is 75 us.  mode: 0 = never, 1 = when this model says it pays, 2 = whenever
// there is a tail (tests, A/B runs); set process-wide by pnp_set_tuning("streamk", mode).
static std::atomic<int> g_streamk_mode{1};
void set_streamk_mode(int m) { g_streamk_mode.store(m, std::memory_order_relaxed); }
int streamk_mode() { return g_streamk_mode.load(std::memory_order_relaxed); }
static bool streamk_pays(int tail, int cap, int nk2, int upw, int np) {
    // a tail round that leaves CUs idle runs its tiles FASTER than a full round (clock and memory system to itself): 0.73-0.82 of
    // the full-round tile time up to ~60 % of the CUs, 0.96 at 80 %, 1.05 at 95 % (K = 4096 launches of 84 ... 244 tiles)
    const double x = (double)tail / cap;
    const double whole = (nk2 + 2.0) * (x <= 0.6 ? 0.75 : 0.75 + (x - 0.6) * 0.857);
    const double sk = 1.05 * upw + 4.0 + (tail <= 64 ? 1.0 : 4.5) + 0.75 * np + 1.5;
    return sk * 1.05 + 0.5 < whole;                 // at least 5 % of the tail round, or it is not worth a second code path
}

template <int EPI>
static int launch_x3(GemmArgs g, hipStream_t s) {
    const int nbm = (g.M + 255) / 256, nbn = g.N / 256;
    const int n_cu = device_cu_count();
    if (!n_cu) return PNP_ERR_HIP;
    static std::atomic<uint32_t> opted{0}, opted_sk{0};            // per device ordinal (common.h: lds_opt_in)
    const int ntiles = nbm * nbn;
    int cap = n_cu;
#ifdef PNP_DEV
    if (getenv("PNP_GEMM_GRID")) cap = atoi(getenv("PNP_GEMM_GRID"));
#endif
    const int mode = streamk_mode();
    StreamKWs* const ws = g.sk;
    if (mode && ws && ws->part && ws->wgs >= cap) {
        const int rounds = ntiles / cap, tail = ntiles - rounds * cap, nk2 = g.K / 64;
        if (tail > 0) {
            const int upw = (int)(((long)tail * nk2 + cap - 1) / cap);
            const int np = (nk2 - 1 + upw - 1) / upw;                    // parts in front of the last one of a tile, at most
            // (a stream that is being captured into a graph keeps whole tiles: the event chain below reaches across streams, which
            // a capture must not, and one-launch-at-a-time cannot be promised for replays)
            hipStreamCaptureStatus cap_st = hipStreamCaptureStatusNone;
            const bool capturing = hipStreamIsCapturing(s, &cap_st) == hipSuccess && cap_st != hipStreamCaptureStatusNone;
            if (!capturing && (mode == 2 || streamk_pays(tail, cap, nk2, upw, np))) {
                if (lds_opt_in(opted_sk, reinterpret_cast<const void*>(gemm_nt_x3_kernel<EPI, true>), kWideSmem) != PNP_OK) return PNP_ERR_HIP;
                g.sk_part = ws->part;
                g.sk_flag = ws->flag;
                g.sk_tmo = ws->flag + ws->wgs;
                g.sk_full = rounds * cap;
                g.sk_upw = upw;
                ws->launches++;
                // ONE stream-K launch in flight per device: an owner spins on workgroups of its own launch with lower ids, which
                // the dispatcher started before it -- true within a launch, but two such launches from two streams, each resident on
                // part of the chip, could hold the CUs the other's missing producers need.  Every launch waits for the event behind the
                // previous one (whatever stream it was on) and leaves its own; kernels that never spin overlap with it as before
                static std::mutex mu;
                static hipEvent_t last[kMaxDevices];
                const int d = current_device();
                if (d < 0) return PNP_ERR_HIP;
                std::lock_guard<std::mutex> lk(mu);
                if (!last[d]) {
                    if (hipEventCreateWithFlags(&last[d], hipEventDisableTiming) != hipSuccess) return PNP_ERR_HIP;
                } else if (hipStreamWaitEvent(s, last[d], 0) != hipSuccess) {
                    return PNP_ERR_HIP;
                }
                hipLaunchKernelGGL((gemm_nt_x3_kernel<EPI, true>), dim3(cap), dim3(512), kWideSmem, s, g);
                const bool ok = hipGetLastError() == hipSuccess && hipEventRecord(last[d], s) == hipSuccess;
                return ok ? PNP_OK : PNP_ERR_HIP;
            }
        }
    }
    if (lds_opt_in(opted, reinterpret_cast<const void*>(gemm_nt_x3_kernel<EPI, false>), kWideSmem) != PNP_OK) return PNP_ERR_HIP;
    const int grid = ntiles > cap ? cap : ntiles;        // one workgroup per CU (LDS-limited) walks the tiles
    hipLaunchKernelGGL((gemm_nt_x3_kernel<EPI, false>), dim3(grid), dim3(512), kWideSmem, s, g);
    return hipGetLastError() == hipSuccess ? PNP_OK : PNP_ERR_HIP;
}

int streamk_ws_create(StreamKWs* ws, int wgs) {
    if (!ws || wgs <= 0) return PNP_ERR_ARG;
    const size_t fbytes = ((size_t)(wgs + 1) * 4 + 15) / 16 * 16;
    if (hipMalloc(&ws->part, (size_t)wgs * 256 * 256 * 4) != hipSuccess) return PNP_ERR_HIP;
    if (hipMalloc(&ws->flag, fbytes) != hipSuccess || hipMemset(ws->flag, 0, fbytes) != hipSuccess) {
        (void)hipFree(ws->part);
        ws->part = nullptr;
        return PNP_ERR_HIP;
    }
    ws->wgs = wgs;
    return PNP_OK;
}

void streamk_ws_destroy(StreamKWs* ws) {
    if (!ws) return;
    if (ws->part) (void)hipFree(ws->part);
    if (ws->flag) (void)hipFree(ws->flag);
    *ws = StreamKWs();
}

int streamk_ws_timeouts(StreamKWs* ws, unsigned* out) {
    if (!ws || !out) return PNP_ERR_ARG;
    *out = 0;
    if (!ws->flag) return PNP_OK;
    return hipMemcpy(out, ws->flag + ws->wgs, 4, hipMemcpyDeviceToHost) == hipSuccess ? PNP_OK : PNP_ERR_HIP;
}

StreamKWs* streamk_ws_default() {
    static StreamKWs ws[kMaxDevices];
    static std::mutex mu;
    const int d = current_device();
    if (d < 0) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    if (!ws[d].part) {
        const int n = device_cu_count();
        if (!n || streamk_ws_create(&ws[d], n) != PNP_OK) return nullptr;
    }
    return &ws[d];
}

int launch_x3_wide(int epi, const GemmArgs& g, hipStream_t s) {
    switch (epi) {
        case WIDE_RESID_F32: return launch_x3<WIDE_RESID_F32>(g, s);
        case WIDE_BIAS_F32: return launch_x3<WIDE_BIAS_F32>(g, s);
        case WIDE_GELU_SPLIT: return launch_x3<WIDE_GELU_SPLIT>(g, s);
        case WIDE_SPLIT: return launch_x3<WIDE_SPLIT>(g, s);
        case WIDE_TOKCOLS_F32: return launch_x3<WIDE_TOKCOLS_F32>(g, s);
    }
    return PNP_ERR_ARG;
}

}  // namespace pnp
