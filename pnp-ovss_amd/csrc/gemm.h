// Host-visible description of one fused NT GEMM launch (see gemm.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace pnp {

enum { GEMM_EPI_LINEAR = 0, GEMM_EPI_GELU = 1, GEMM_EPI_GELU_GRAD = 2 };

struct GemmProfile;

// Workspace of the in-launch reductions (gemm_x3.hip stream-K tail; the text-side split-K): one per engine -- engines run
// concurrently on their own streams -- sized for `wgs` workgroups.  part: wgs x 256 KB of fp32 partial tiles; flag: wgs ready
// words + one give-up word (zeroed at allocation; every consumer resets the word it consumed, so launches need no memset).
struct StreamKWs {
    float* part = nullptr;
    unsigned* flag = nullptr;
    int wgs = 0;
    long long launches = 0;        // launches that used the workspace (diagnostics)
};
int streamk_ws_create(StreamKWs* ws, int wgs);
void streamk_ws_destroy(StreamKWs* ws);
int streamk_ws_timeouts(StreamKWs* ws, unsigned* out);     // synchronises the device; *out = give-up word (0 = none)
StreamKWs* streamk_ws_default();                          // per-device workspace of the op-level entry points (one stream at a time)
void set_streamk_mode(int mode);
int streamk_mode();

struct GemmArgs {
    const void* A = nullptr;   // [M, lda] T
    const void* B = nullptr;   // [N, ldb] T
    int M = 0, N = 0, K = 0, lda = 0, ldb = 0;
    int Nvalid = 0;            // set by gemm_nt (N before tile rounding)
    int mode = GEMM_EPI_LINEAR;
    const float* bias = nullptr;   // [N] (or [M] when bias_on_rows)
    int bias_on_rows = 0;
    const float* resid = nullptr;  // f32 [M, ldr] added after activation (or pos_embed when row_div>0)
    int ldr = 0;
    float* out_f32 = nullptr;      // optional f32 output [*, ldo]
    int ldo = 0;
    void* out_t = nullptr;         // optional T output [*, ldo_t]
    int ldo_t = 0;
    float* aux = nullptr;          // GELU: pre-activation stash (f32) ; GELU_GRAD: pre-activation input
    int ld_aux = 0;
    // split-bf16 ("bf16x3") form of the wide kernel: A = A (hi) + A_lo, B = B (hi) + B_lo, all bf16, and the product is
    // accumulated as A.B + A.B_lo + A_lo.B on the bf16 MFMA in fp32 (the dropped A_lo.B_lo term is 2^-18 relative)
    const void* A_lo = nullptr;    // [M, lda] bf16
    const void* B_lo = nullptr;    // [N, ldb] bf16
    void* out_lo = nullptr;        // optional bf16 low part of a split output [*, ldo_t] (out_t holds the high part)
    int a_f32 = 0;                 // 1: A is fp32 [M, lda] and is split by the kernel (text-side form: B = bf16 hi, B_lo = bf16 lo,
                                   // outputs fp32 -- out_t, if given, is a float array like out_f32)
    int row_div = 0;               // >0: patch rows -> token rows b*(row_div+1)+1+p, resid = pos_embed
    int col_div = 0, col_pad = 0;  // >0: output column n -> (n / col_div) * col_pad + n % col_div
    struct GemmProfile* prof = nullptr;     // live timing ring of the calling engine (bench.py roofline), null = none
    StreamKWs* sk = nullptr;                // workspace of the calling engine for the stream-K tail (null = data-parallel tiles only)
    float* sk_part = nullptr;               // set by the launcher from `sk` when the tail is split: partial tiles, flags, give-up word,
    unsigned* sk_flag = nullptr;            // tiles of the whole rounds, slab pairs per workgroup in the tail
    unsigned* sk_tmo = nullptr;
    int sk_full = 0, sk_upw = 0;
    unsigned long long* stamps = nullptr;   // diagnostics (DEV builds, PNP_GEMM_STAMPS): per-workgroup clock stamps, 8 per block
    int ablate = 0;                // DEV builds only (PNP_GEMM_ABLATE): 1 = no steady-state DMA, 2 = no MFMA
};

int gemm_nt(int dtype_bf16, GemmArgs g, hipStream_t s);

// Optional live timing of the big-tile GEMM launches (bench.py roofline): a ring of event pairs.  One per engine (engines
// are driven from different host threads when several batches are in flight); gemm_nt touches it through GemmArgs::prof.
struct GemmProfile {
    bool on = false;
    static constexpr int kMax = 8192;
    hipEvent_t ev0[kMax], ev1[kMax];
    int created = 0, used = 0;
    int period = 1;         // bracket every period-th launch
    long long seq = 0;
    long long launches = 0;
    double flops = 0;
};
int gemm_read_stamps(unsigned long long* host_out, int max_blocks);

}  // namespace pnp
