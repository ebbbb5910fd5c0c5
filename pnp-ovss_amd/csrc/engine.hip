// The engine behind include/pnp_hip.h: owns weights + workspace on one MI355X, sequences the
// hand-written kernels of this directory on the caller's stream.  Host logic only; every FLOP and
// byte of the hot path is in the .hip kernels.  No allocation / sync inside hot-path calls.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/pnp_hip.h"
#include "common.h"
#include "kernels.h"

using namespace pnp;

namespace {

struct Buf {
    void* p = nullptr;
    size_t bytes = 0;
};

struct TextLayerW {
    // forward (T = compute dtype)
    void *qkv_w = nullptr, *so_w = nullptr, *cq_w = nullptr, *co_w = nullptr, *i_w = nullptr, *o_w = nullptr;
    float *qkv_b = nullptr, *so_b = nullptr, *sln_w = nullptr, *sln_b = nullptr, *cq_b = nullptr, *co_b = nullptr,
          *cln_w = nullptr, *cln_b = nullptr, *i_b = nullptr, *o_b = nullptr, *oln_w = nullptr, *oln_b = nullptr;
    // backward (transposed copies, layers >= stash_layer)
    void *o_wT = nullptr, *i_wT = nullptr, *co_wT = nullptr, *cq_wT = nullptr, *so_wT = nullptr, *qkv_wT = nullptr;
};

struct TextLayerA {
    void* qkv = nullptr;       // T [R,3H]
    float* Ps = nullptr;       // [B,nh,L,L]
    float *a_hat = nullptr, *a_rstd = nullptr, *a_out = nullptr;
    void* a_outT = nullptr;
    void* qc = nullptr;        // T [R,H]
    float* Pc = nullptr;       // [B,nh,L,Nst]  (layers >= stash only)
    float *c_hat = nullptr, *c_rstd = nullptr, *c_out = nullptr;
    void* c_outT = nullptr;
    float* u = nullptr;        // [R,I]
    float *o_hat = nullptr, *o_rstd = nullptr, *h_out = nullptr;
    void* h_outT = nullptr;
};

struct VitLayerW {
    float *n1w, *n1b, *n2w, *n2b, *qkv_b, *proj_b, *fc1_b, *fc2_b;
    void *qkv_w, *proj_w, *fc1_w, *fc2_w;
};

// Device allocations that hold WEIGHTS (converted GEMM weights, biases, LayerNorm parameters, embeddings): read-only once
// finalized, so several engines of a process may use one copy (pnp_create_shared); freed with the last engine that holds it.
struct WeightStore {
    int device = 0;
    std::vector<void*> allocs;
    size_t bytes = 0;
    ~WeightStore() {
        (void)hipSetDevice(device);
        for (void* p : allocs) (void)hipFree(p);
    }
};

}  // namespace

struct pnp_engine {
    pnp_config c{};
    std::shared_ptr<WeightStore> wstore;       // owner(s) of every weight allocation below
    bool to_store = false;                     // dalloc target: the weight store (true) or this engine's own list
    bool shares_weights = false;               // created by pnp_create_shared: the weights belong to a donor's store
    int bf = 0;                // 1: bf16 storage / bf16 MFMA everywhere
    int x3 = 0;                // 1: split-bf16 ("bf16x3") ViT Linears + cross K/V projections, everything else as fp32 mode
    size_t esz = 4;
    int P = 0, PP = 0, N = 0, Npad = 0, D = 0, H = 0, I = 0, TL = 0, nh = 0, Nst = 0, SL = 0;
    char err[512] = {0};
    std::vector<void*> allocs;
    size_t alloc_bytes = 0;
    std::map<std::string, Buf> named;          // raw fp32 device copies of small params + test buffers
    std::map<std::string, bool> loaded;
    bool finalized = false;

    // ---- weights
    float *cls = nullptr, *pos = nullptr, *patch_b = nullptr, *vnorm_w = nullptr, *vnorm_b = nullptr;
    void* patch_w = nullptr;
    std::vector<VitLayerW> vit;
    float *word = nullptr, *tpos = nullptr, *eln_w = nullptr, *eln_b = nullptr, *itm_w = nullptr, *itm_b = nullptr;
    std::vector<TextLayerW> txt;
    void *ck_w = nullptr, *cv_w = nullptr;     // cross K / V weights of all layers: [TL*H, D]
    float *ck_b = nullptr, *cv_b = nullptr;    // [TL*H]
    std::vector<Buf> staging;                  // fp32 staging of GEMM weights until finalize

    // ---- activations
    int ldq = 0;               // row stride (elements) of the fused q|k|v buffer
    float* x0 = nullptr;       // [M, D] token embeddings of drop iteration 0 (patch embed + pos, cls row): later iterations reuse them
    // what the reusable state was computed from (embed == 1 call): an embed == 2 call that does not match recomputes instead.
    // CONTRACT: the record catches stale POINTERS and SHAPES, not stale CONTENTS -- a caller that refills the same image / id
    // buffers in place between an embed == 1 and an embed == 2 call gets the old embeddings (the one caller, pnp_drop_loop_layer,
    // always starts a batch with embed == 1; pnp_vit_forward / pnp_text_forward_xattn invalidate the record)
    struct { const float* images = nullptr; const int64_t* ids = nullptr; const int64_t* mask = nullptr; int B = 0, L = 0, ld = 0;
             bool vit = false, text = false; } reuse;
    void *patches = nullptr, *xn = nullptr, *qk = nullptr, *vt = nullptr, *ctx = nullptr, *h1 = nullptr, *embT = nullptr;
    float *x = nullptr, *emb32 = nullptr;
    void *Knat = nullptr, *Vnat = nullptr, *Kt = nullptr, *Vt = nullptr;
    std::vector<TextLayerA> ta;
    float *temb = nullptr, *h0 = nullptr, *tmp = nullptr;
    void *h0T = nullptr, *ctx_s = nullptr, *ctx_c = nullptr, *g = nullptr;
    // backward
    float *dh = nullptr, *d_pre = nullptr, *dc = nullptr, *d_cpre = nullptr, *da = nullptr, *d_apre = nullptr,
          *dctx_s = nullptr, *dS = nullptr, *dPc = nullptr;
    void *d_preT = nullptr, *dg = nullptr, *d_cpreT = nullptr, *dctxc = nullptr, *dqc = nullptr, *d_apreT = nullptr,
         *dqkv = nullptr;
    int grad_layer = -1;       // text layer whose dL/dP the dPc buffer currently holds (-1: none)
    // drop loop
    float* G = nullptr;
    uint8_t* dropped = nullptr;
    float* logits_scratch = nullptr;

    // ---- post-process state
    struct Post {
        bool reserved = false, prepared = false, has_crf = false;
        int groups_cap = 1;                   // 2: the CRF arrays were sized for the paired (two-group) run
        int maxB = 0, maxK = 0, max_pix_img = 0, chunk = 0;
        int64_t max_total_pix = 0;
        int B = 0, Cmax = 0, Kmax = 0, maxHW = 0;
        int64_t total_pix = 0;
        std::vector<PostDesc> desc, desc_pair;
        PostDesc* d_desc = nullptr;
        PostDesc* d_desc_pair = nullptr;      // same batch with two channel groups per row (1-drop | N-drop)
        int32_t *d_img_cls_off = nullptr, *d_cls_off = nullptr, *d_tok_idx = nullptr, *d_cls_div = nullptr, *d_lut = nullptr;
        int lut_stride = 0;
        size_t* d_label_off = nullptr;
        double* d_wts = nullptr;
        int32_t* d_wt_off = nullptr;
        float *merged = nullptr, *thr = nullptr, *maps = nullptr, *maps2 = nullptr, *maps3 = nullptr, *stats = nullptr;
        float *unary = nullptr, *Q = nullptr, *va = nullptr, *vb = nullptr, *vga = nullptr, *vgb = nullptr, *norm[2] = {nullptr, nullptr};
        int Kpmax = 0, Kpmax_pair = 0, maxH = 0, maxW = 0, max_radius = 0, maxKp = 0;
        size_t valg_cap = 0;
        std::vector<int> gauss_sig;      // (H, W) list the Gaussian lattice was last built for
        const uint8_t* d_rgb = nullptr;
        const float* d_gt = nullptr;
        CrfLattice lat[2]{};
        uint64_t *keys_a = nullptr, *keys_b = nullptr;
        uint32_t* vals_a = nullptr;
        int *n1k = nullptr, *n2k = nullptr;
        int *head = nullptr, *incl = nullptr, *range_err = nullptr;
        int range_err_host = 0;                // key-range flag of the lattices built by the last prepare
        int lat_points[2] = {0, 0};            // lattice points of the prepared batch (Gaussian, bilateral)
        std::vector<size_t> h_label_off;       // host staging of the per-batch tables (see pnp_post_prepare)
        std::vector<int32_t> h_wt_off;
        std::vector<double> h_wts;
        void* sort_tmp = nullptr;
        size_t sort_tmp_bytes = 0;
        size_t cap[2] = {0, 0};
        size_t val_cap = 0;
        int lut_cap = 0, cls_cap = 0, tok_cap = 0, wts_cap = 0;
        bool maps_in_2 = false;   // where the current maps live after blur
    } post;

    GemmProfile* gemm_prof = nullptr;          // live timing ring of this engine's dense GEMM launches (allocated on first enable)
    StreamKWs sk_ws;                           // partial tiles + flags of the in-launch reductions (split-bf16 mode: gemm_x3.hip)
    // live timing of one pipeline stage (bench.py's hbm roofline record for the DenseCRF mean-field): event pairs
    // on the launch stream around the stage, algorithmic bytes summed beside them
    struct StageProfile {
        bool on = false;
        std::vector<hipEvent_t> ev0, ev1;
        int used = 0;
        long long launches = 0;
        double work = 0;                       // SURVEY.md 8d bytes of the bracketed iterations
        double lattice = 0;                    // lattice-blur term of the same brackets (reported separately)
    } crf_prof;
};

namespace {

int fail(pnp_engine* e, int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(e->err, sizeof(e->err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIPCHK(e, call)                                                                      \
    do {                                                                                     \
        hipError_t _s = (call);                                                              \
        if (_s != hipSuccess) return fail(e, PNP_ERR_HIP, "%s: %s", #call, hipGetErrorString(_s)); \
    } while (0)
#define KCHK(e, call)                                                         \
    do {                                                                      \
        int _r = (call);                                                      \
        if (_r != PNP_OK) return fail(e, _r, "%s failed (%d): %s", #call, _r, hipGetErrorString(hipGetLastError())); \
    } while (0)

template <typename T>
int dalloc(pnp_engine* e, T** out, size_t count, bool zero = false) {
    void* p = nullptr;
    const size_t bytes = (count * sizeof(T) + 255) / 256 * 256;
    if (hipMalloc(&p, bytes ? bytes : 256) != hipSuccess) return fail(e, PNP_ERR_NOMEM, "hipMalloc(%zu) failed", bytes);
    if (zero && hipMemset(p, 0, bytes ? bytes : 256) != hipSuccess) return fail(e, PNP_ERR_HIP, "hipMemset failed");
    if (e->to_store && e->wstore) {
        e->wstore->allocs.push_back(p);
        e->wstore->bytes += bytes;
    } else {
        e->allocs.push_back(p);
    }
    e->alloc_bytes += bytes;
    *out = reinterpret_cast<T*>(p);
    return PNP_OK;
}
int dalloc_t(pnp_engine* e, void** out, size_t count, bool zero = false) {   // `count` elements of the compute type
    char* p = nullptr;
    int r = dalloc<char>(e, &p, count * e->esz, zero);
    *out = p;
    return r;
}

bool ends_with(const std::string& s, const char* suf) {
    const size_t n = strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

__global__ void transpose_cast_kernel_f32(const float* __restrict__ in, float* __restrict__ out, int rows, int cols) {
    __shared__ float t[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    for (int i = threadIdx.y; i < 32; i += 8) {
        const int r = by + i, c = bx + threadIdx.x;
        t[i][threadIdx.x] = (r < rows && c < cols) ? in[(size_t)r * cols + c] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += 8) {
        const int c = bx + i, r = by + threadIdx.x;
        if (r < rows && c < cols) out[(size_t)c * rows + r] = t[threadIdx.x][i];
    }
}

// The two extra layouts the analytic backward reads (V token-major for text layers >= stash_layer, K^T feature-major for
// layers > stash_layer) as 32 x 32 LDS transposes of what the forward's two cross K / V GEMMs have just written, instead of
// two more GEMMs over the same products (split-bf16 mode: 0.56 ms of GEMM per forward against 0.18 ms of copies; the copies
// are also bit-identical to the forward's values, which two GEMMs with swapped operand roles are not).
//   feat_to_tok: src [R features, ld_src] with token column b * n_pad + t  ->  dst [(b * N + t), R]
__global__ void feat_to_tok_kernel(const float* __restrict__ src, int ld_src, int n_pad, float* __restrict__ dst, int R, int N) {
    __shared__ float t[32][33];
    const int b = blockIdx.z, r0 = blockIdx.x * 32, t0 = blockIdx.y * 32;
    for (int i = threadIdx.y; i < 32; i += 8) {
        const int r = r0 + i, tok = t0 + threadIdx.x;
        t[i][threadIdx.x] = (r < R && tok < N) ? src[(size_t)r * ld_src + (size_t)b * n_pad + tok] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += 8) {
        const int tok = t0 + i, r = r0 + threadIdx.x;
        if (r < R && tok < N) dst[((size_t)b * N + tok) * R + r] = t[threadIdx.x][i];
    }
}
//   tok_to_feat: src [(b * N + t), ld_src] columns c0 .. c0 + R  ->  dst [R features, ld_dst] at token column b * n_pad + t
__global__ void tok_to_feat_kernel(const float* __restrict__ src, int ld_src, float* __restrict__ dst, int ld_dst, int n_pad, int R, int N) {
    __shared__ float t[32][33];
    const int b = blockIdx.z, r0 = blockIdx.x * 32, t0 = blockIdx.y * 32;
    for (int i = threadIdx.y; i < 32; i += 8) {
        const int tok = t0 + i, r = r0 + threadIdx.x;
        t[i][threadIdx.x] = (r < R && tok < N) ? src[((size_t)b * N + tok) * ld_src + r] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += 8) {
        const int r = r0 + i, tok = t0 + threadIdx.x;
        if (r < R && tok < N) dst[(size_t)r * ld_dst + (size_t)b * n_pad + tok] = t[threadIdx.x][i];
    }
}

// fp32 [rows, cols] staging -> compute-type device weight (optionally transposed); split: a bf16 (hi | lo) pair in
// the same bytes as the fp32 copy, hi first (split-bf16 mode, weights of the wide GEMMs)
// Patch embeddings of a later drop iteration: the images are the ones of iteration 0 with more 16 x 16 blocks zeroed
// (PnP.py:597-603), so a token's embedding is either what iteration 0 computed or, for a dropped patch, bias + pos exactly as the
// GEMM epilogue forms it from a zero accumulator ((0 + bias[n]) + pos[t][n]); the cls row never changes.  One pass over x instead of
// patchify + the patch GEMM.
__global__ void embed_reuse_kernel(const float* __restrict__ x0, const uint8_t* __restrict__ dropped, const float* __restrict__ bias,
                                   const float* __restrict__ pos, float* __restrict__ x, int rows, int N, int D4) {
    const size_t total = (size_t)rows * D4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / D4), c = (int)(i - (size_t)m * D4);
        const int b = m / N, t = m - b * N;
        f32x4 v;
        if (t > 0 && dropped[(size_t)b * (N - 1) + (t - 1)]) {
            const f32x4 bv = reinterpret_cast<const f32x4*>(bias)[c];
            const f32x4 pv = reinterpret_cast<const f32x4*>(pos)[(size_t)t * D4 + c];
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = __fadd_rn(__fadd_rn(0.0f, bv[e]), pv[e]);
        } else {
            v = reinterpret_cast<const f32x4*>(x0)[i];
        }
        reinterpret_cast<f32x4*>(x)[i] = v;
    }
}

int make_weight(pnp_engine* e, const float* src32, int rows, int cols, bool transpose, void** out, bool split = false) {
    KCHK(e, dalloc_t(e, out, (size_t)rows * cols));
    const float* s = src32;
    float* tmp = nullptr;
    if (transpose) {
        HIPCHK(e, hipMalloc((void**)&tmp, (size_t)rows * cols * 4));
        hipLaunchKernelGGL(transpose_cast_kernel_f32, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(32, 8), 0, 0, src32, tmp,
                           rows, cols);
        s = tmp;
    }
    int r = split ? split_f32(s, *out, (char*)*out + (size_t)rows * cols * 2, (size_t)rows * cols, 0)
                  : cast_f32(e->bf, s, *out, (size_t)rows * cols, 0);
    hipError_t st = hipDeviceSynchronize();
    if (tmp) (void)hipFree(tmp);
    if (r != PNP_OK || st != hipSuccess) return fail(e, PNP_ERR_HIP, "weight conversion failed");
    return PNP_OK;
}

GemmArgs G_(const void* A, int lda, const void* B, int ldb, int M, int N, int K) {
    GemmArgs g;
    g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.M = M; g.N = N; g.K = K;
    return g;
}

// every GEMM of an engine goes through here: the launch is timed into the engine's own ring when profiling is on
int egemm(pnp_engine* e, int bf, GemmArgs g, hipStream_t s) {
    g.prof = e->gemm_prof;
    g.sk = e->sk_ws.part ? &e->sk_ws : nullptr;
    return gemm_nt(bf, g, s);
}

// a text-side Linear (M = B*L rows): in the split-bf16 mode the weight is a (hi | lo) bf16 pair and the fp32 activations are
// split by the kernel (gemm_nt_small_x3_kernel); otherwise the compute type's generic kernel
int tgemm(pnp_engine* e, GemmArgs g, hipStream_t s) {
    if (e->x3) {
        g.a_f32 = 1;
        g.B_lo = (const char*)g.B + (size_t)g.N * g.K * 2;      // make_weight(split): lo array behind the hi array
    }
    return egemm(e, e->bf, g, s);
}

const char* kVitNames[] = {"norm1.weight", "norm1.bias", "attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight",
                           "attn.proj.bias", "norm2.weight", "norm2.bias", "mlp.fc1.weight", "mlp.fc1.bias",
                           "mlp.fc2.weight", "mlp.fc2.bias"};

}  // namespace

// =========================================================================================== lifetime

extern "C" size_t pnp_workspace_bytes(const pnp_config* c) {
    if (!c) return 0;
    const size_t es = c->compute_bf16 == 1 ? 2 : 4;
    const size_t P = c->img_size / c->patch, N = P * P + 1, Npad = (N + 63) / 64 * 64, D = c->vit_dim, H = c->txt_hidden,
                 I = c->txt_inter, TL = c->txt_layers, B = c->max_batch, L = c->max_text_len;
    const size_t M = B * N, R = B * L;
    size_t w = (size_t)c->vit_depth * (12 * D * D) * es + TL * (4 * H * H + 2 * H * D + 2 * H * I) * es * 2 +
               ((size_t)c->vocab + c->max_pos) * H * 4;
    size_t a = M * (768 + D * 2 + 3 * D + 64 + D + 4 * D) * es + M * D * 12 + D * B * Npad * es + M * TL * H * es * 2 +
               2 * TL * H * B * Npad * es;            // (incl. the padded q|k|v rows and the drop loop's copy of the token embeddings)
    size_t t = TL * (R * (3 * H + 2 * H) * es + R * (H * 8 + I) * 4 + B * (H / 64) * L * (L + Npad) * 4) + R * I * (4 + es) * 2 +
               R * H * 64;
    return w + a + t + (64u << 20);
}

extern "C" const char* pnp_last_error(const pnp_engine* e) { return e ? e->err : "null engine"; }

extern "C" void pnp_destroy(pnp_engine* e) {
    if (!e) return;
    (void)hipSetDevice(e->c.device);
    (void)hipDeviceSynchronize();
    for (void* p : e->allocs) (void)hipFree(p);
    for (auto& b : e->staging) if (b.p) (void)hipFree(b.p);
    if (e->gemm_prof) {
        for (int i = 0; i < e->gemm_prof->created; i++) {
            (void)hipEventDestroy(e->gemm_prof->ev0[i]);
            (void)hipEventDestroy(e->gemm_prof->ev1[i]);
        }
        delete e->gemm_prof;
    }
    for (hipEvent_t ev : e->crf_prof.ev0) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : e->crf_prof.ev1) (void)hipEventDestroy(ev);
    streamk_ws_destroy(&e->sk_ws);
    delete e;
}

static int create_impl(const pnp_config* cfg, pnp_engine* donor, pnp_engine** out);
extern "C" int pnp_create(const pnp_config* cfg, pnp_engine** out) { return create_impl(cfg, nullptr, out); }
extern "C" int pnp_create_shared(const pnp_config* cfg, pnp_engine* donor, pnp_engine** out) {
    if (!donor) return PNP_ERR_ARG;
    return create_impl(cfg, donor, out);
}
static int create_impl(const pnp_config* cfg, pnp_engine* donor, pnp_engine** out) {
    if (!cfg || !out) return PNP_ERR_ARG;
    pnp_engine* e = new pnp_engine();
    *out = e;
    e->c = *cfg;
    if (donor) {
        const pnp_config& d = donor->c;
        if (!donor->finalized) return fail(e, PNP_ERR_STATE, "pnp_create_shared: the donor's weights are not finalized");
        if (d.device != cfg->device || d.compute_bf16 != cfg->compute_bf16 || d.img_size != cfg->img_size || d.patch != cfg->patch ||
            d.vit_dim != cfg->vit_dim || d.vit_depth != cfg->vit_depth || d.vit_heads != cfg->vit_heads ||
            d.vit_mlp_ratio != cfg->vit_mlp_ratio || d.txt_hidden != cfg->txt_hidden || d.txt_layers != cfg->txt_layers ||
            d.txt_heads != cfg->txt_heads || d.txt_inter != cfg->txt_inter || d.vocab != cfg->vocab || d.max_pos != cfg->max_pos ||
            d.vit_ln_eps != cfg->vit_ln_eps || d.txt_ln_eps != cfg->txt_ln_eps)
            return fail(e, PNP_ERR_ARG, "pnp_create_shared: device, compute mode and model geometry must equal the donor's");
        if (cfg->stash_layer < d.stash_layer)
            return fail(e, PNP_ERR_ARG, "pnp_create_shared: stash_layer %d below the donor's %d (its transposed backward weights "
                        "exist for layers >= %d only)", cfg->stash_layer, d.stash_layer, d.stash_layer);
    }
    const pnp_config& c = e->c;
    if (c.compute_bf16 < 0 || c.compute_bf16 > 2) return fail(e, PNP_ERR_ARG, "compute_bf16 must be 0 (fp32), 1 (bf16) or 2 (split-bf16)");
    e->bf = c.compute_bf16 == 1 ? 1 : 0;
    e->x3 = c.compute_bf16 == 2 ? 1 : 0;
    e->esz = e->bf ? 2 : 4;
    if (c.patch != 16) return fail(e, PNP_ERR_ARG, "patch must be 16");
    if (c.img_size % 16 || c.img_size <= 0) return fail(e, PNP_ERR_ARG, "img_size must be a positive multiple of 16");
    if (c.vit_dim != c.vit_heads * 64 || c.txt_hidden != c.txt_heads * 64)
        return fail(e, PNP_ERR_ARG, "head_dim must be 64 (vit_dim=%d heads=%d, txt_hidden=%d heads=%d)", c.vit_dim,
                    c.vit_heads, c.txt_hidden, c.txt_heads);
    if (c.vit_dim % 128 || c.txt_hidden % 128 || c.txt_inter % 128 || (c.vit_dim * c.vit_mlp_ratio) % 128)
        return fail(e, PNP_ERR_ARG, "widths must be multiples of 128");
    if (c.vit_dim > 1024 || c.txt_hidden > 1024) return fail(e, PNP_ERR_ARG, "LayerNorm width > 1024 unsupported");
    if (c.max_text_len < 5 || c.max_text_len > 512 || c.max_text_len > c.max_pos)
        return fail(e, PNP_ERR_ARG, "max_text_len must be in [5, min(512, max_pos = %d)]", c.max_pos);
    if (c.stash_layer < 0 || c.stash_layer >= c.txt_layers) return fail(e, PNP_ERR_ARG, "stash_layer out of range");
    if (c.max_batch <= 0) return fail(e, PNP_ERR_ARG, "max_batch must be positive");
    HIPCHK(e, hipSetDevice(c.device));
    e->P = c.img_size / 16;
    e->PP = e->P * e->P;
    e->N = e->PP + 1;
    e->Npad = (e->N + 63) / 64 * 64;
    e->Nst = e->Npad;
    e->D = c.vit_dim;
    e->H = c.txt_hidden;
    e->I = c.txt_inter;
    e->TL = c.txt_layers;
    e->nh = c.txt_heads;
    e->SL = c.stash_layer;
    if ((e->N + 15) / 16 > 160) return fail(e, PNP_ERR_ARG, "too many image tokens (%d) for the cross-attention kernel", e->N);
    const size_t B = c.max_batch, M = B * e->N, D = e->D, H = e->H, I = e->I, L = c.max_text_len, R = B * L, TL = e->TL;
    const size_t ldv = B * e->Npad;
    e->vit.resize(c.vit_depth);
    e->txt.resize(TL);
    e->ta.resize(TL);
    if (e->x3) {                               // one 256 KB partial-tile slot + one flag per CU (64 MB): not shared between engines
        const int n_cu = device_cu_count();
        if (!n_cu || streamk_ws_create(&e->sk_ws, n_cu) != PNP_OK) return fail(e, PNP_ERR_HIP, "stream-K workspace allocation failed");
        e->alloc_bytes += (size_t)n_cu * 256 * 256 * 4;
    }
    // activations
    KCHK(e, dalloc_t(e, &e->patches, B * e->PP * 768));
    KCHK(e, dalloc(e, &e->x, M * D));
    KCHK(e, dalloc(e, &e->x0, M * D));
    KCHK(e, dalloc_t(e, &e->xn, M * D));
    // bf16 / split-bf16 modes: fused q|k|v rows at a stride of 3D + 64 elements; fp32 mode: q|k rows (v goes to vt).
    // The pad matters: an attention K / V tile is 64 rows x 128 B at the row stride, and at 6144 B (3D bf16, D = 1024) those
    // rows crowd a few L2 channels -- 151 us per launch against 136-138 us at 6272 ... 6656 B (tools/attn_ld_probe.py)
    e->ldq = 3 * (int)D + 64;
    KCHK(e, dalloc_t(e, &e->qk, M * (size_t)e->ldq));
    KCHK(e, dalloc_t(e, &e->vt, D * ldv, true));
    KCHK(e, dalloc_t(e, &e->ctx, M * D));
    KCHK(e, dalloc_t(e, &e->h1, M * D * c.vit_mlp_ratio));
    KCHK(e, dalloc_t(e, &e->embT, M * D));
    KCHK(e, dalloc(e, &e->emb32, M * D));
    const int nVn = e->TL - e->SL, nKt = e->TL - e->SL - 1;
    KCHK(e, dalloc_t(e, &e->Knat, M * TL * H));
    KCHK(e, dalloc_t(e, &e->Vnat, M * (size_t)nVn * H));
    KCHK(e, dalloc_t(e, &e->Vt, TL * H * ldv, true));
    if (nKt > 0) KCHK(e, dalloc_t(e, &e->Kt, (size_t)nKt * H * ldv, true));
    KCHK(e, dalloc(e, &e->temb, R * H));
    KCHK(e, dalloc(e, &e->h0, R * H));
    KCHK(e, dalloc_t(e, &e->h0T, R * H));
    KCHK(e, dalloc(e, &e->tmp, R * H));
    KCHK(e, dalloc_t(e, &e->ctx_s, R * H));
    KCHK(e, dalloc_t(e, &e->ctx_c, R * H));
    KCHK(e, dalloc_t(e, &e->g, R * I));
    for (size_t i = 0; i < TL; i++) {
        TextLayerA& a = e->ta[i];
        KCHK(e, dalloc_t(e, &a.qkv, R * 3 * H));
        KCHK(e, dalloc(e, &a.a_hat, R * H));
        KCHK(e, dalloc(e, &a.a_rstd, R));
        KCHK(e, dalloc(e, &a.a_out, R * H));
        KCHK(e, dalloc_t(e, &a.a_outT, R * H));
        KCHK(e, dalloc_t(e, &a.qc, R * H));
        KCHK(e, dalloc(e, &a.c_hat, R * H));
        KCHK(e, dalloc(e, &a.c_rstd, R));
        KCHK(e, dalloc(e, &a.c_out, R * H));
        KCHK(e, dalloc_t(e, &a.c_outT, R * H));
        KCHK(e, dalloc(e, &a.u, R * I));
        KCHK(e, dalloc(e, &a.o_hat, R * H));
        KCHK(e, dalloc(e, &a.o_rstd, R));
        KCHK(e, dalloc(e, &a.h_out, R * H));
        KCHK(e, dalloc_t(e, &a.h_outT, R * H));
        if ((int)i >= e->SL) {
            KCHK(e, dalloc(e, &a.Ps, B * e->nh * L * L));
            KCHK(e, dalloc(e, &a.Pc, B * e->nh * L * (size_t)e->Nst, true));
        }
    }
    KCHK(e, dalloc(e, &e->dh, R * H));
    KCHK(e, dalloc(e, &e->d_pre, R * H));
    KCHK(e, dalloc_t(e, &e->d_preT, R * H));
    KCHK(e, dalloc_t(e, &e->dg, R * I));
    KCHK(e, dalloc(e, &e->dc, R * H));
    KCHK(e, dalloc(e, &e->d_cpre, R * H));
    KCHK(e, dalloc_t(e, &e->d_cpreT, R * H));
    KCHK(e, dalloc_t(e, &e->dctxc, R * H));
    KCHK(e, dalloc_t(e, &e->dqc, R * H));
    KCHK(e, dalloc(e, &e->da, R * H));
    KCHK(e, dalloc(e, &e->d_apre, R * H));
    KCHK(e, dalloc_t(e, &e->d_apreT, R * H));
    KCHK(e, dalloc(e, &e->dctx_s, R * H));
    KCHK(e, dalloc(e, &e->dS, B * e->nh * L * L));
    KCHK(e, dalloc(e, &e->dPc, B * e->nh * L * (size_t)e->Nst, true));
    KCHK(e, dalloc_t(e, &e->dqkv, R * 3 * H));
    KCHK(e, dalloc(e, &e->G, B * L * (size_t)e->PP));
    KCHK(e, dalloc(e, &e->dropped, B * (size_t)e->PP, true));
    KCHK(e, dalloc(e, &e->logits_scratch, B * 2));
    if (donor) {
        // weights: the donor's device copies (read-only after finalize); this engine owns activations and workspace only
        e->wstore = donor->wstore;
        e->shares_weights = true;
        e->cls = donor->cls; e->pos = donor->pos; e->patch_b = donor->patch_b; e->vnorm_w = donor->vnorm_w; e->vnorm_b = donor->vnorm_b;
        e->word = donor->word; e->tpos = donor->tpos; e->eln_w = donor->eln_w; e->eln_b = donor->eln_b;
        e->itm_w = donor->itm_w; e->itm_b = donor->itm_b; e->ck_b = donor->ck_b; e->cv_b = donor->cv_b;
        e->patch_w = donor->patch_w; e->ck_w = donor->ck_w; e->cv_w = donor->cv_w;
        e->vit = donor->vit;
        e->txt = donor->txt;
        e->finalized = true;
        return PNP_OK;
    }
    e->wstore = std::make_shared<WeightStore>();
    e->wstore->device = c.device;
    e->to_store = true;
    // small fp32 params
    KCHK(e, dalloc(e, &e->cls, D));
    KCHK(e, dalloc(e, &e->pos, (size_t)e->N * D));
    KCHK(e, dalloc(e, &e->patch_b, D));
    KCHK(e, dalloc(e, &e->vnorm_w, D));
    KCHK(e, dalloc(e, &e->vnorm_b, D));
    KCHK(e, dalloc(e, &e->word, (size_t)c.vocab * H));
    KCHK(e, dalloc(e, &e->tpos, (size_t)c.max_pos * H));
    KCHK(e, dalloc(e, &e->eln_w, H));
    KCHK(e, dalloc(e, &e->eln_b, H));
    KCHK(e, dalloc(e, &e->itm_w, 2 * H));
    KCHK(e, dalloc(e, &e->itm_b, 2));
    KCHK(e, dalloc(e, &e->ck_b, TL * H));
    KCHK(e, dalloc(e, &e->cv_b, TL * H));
    for (auto& v : e->vit) {
        KCHK(e, dalloc(e, &v.n1w, D)); KCHK(e, dalloc(e, &v.n1b, D)); KCHK(e, dalloc(e, &v.n2w, D)); KCHK(e, dalloc(e, &v.n2b, D));
        KCHK(e, dalloc(e, &v.qkv_b, 3 * D)); KCHK(e, dalloc(e, &v.proj_b, D));
        KCHK(e, dalloc(e, &v.fc1_b, D * c.vit_mlp_ratio)); KCHK(e, dalloc(e, &v.fc2_b, D));
        v.qkv_w = v.proj_w = v.fc1_w = v.fc2_w = nullptr;
    }
    for (auto& t : e->txt) {
        KCHK(e, dalloc(e, &t.qkv_b, 3 * H)); KCHK(e, dalloc(e, &t.so_b, H)); KCHK(e, dalloc(e, &t.sln_w, H)); KCHK(e, dalloc(e, &t.sln_b, H));
        KCHK(e, dalloc(e, &t.cq_b, H)); KCHK(e, dalloc(e, &t.co_b, H)); KCHK(e, dalloc(e, &t.cln_w, H)); KCHK(e, dalloc(e, &t.cln_b, H));
        KCHK(e, dalloc(e, &t.i_b, I)); KCHK(e, dalloc(e, &t.o_b, H)); KCHK(e, dalloc(e, &t.oln_w, H)); KCHK(e, dalloc(e, &t.oln_b, H));
    }
    e->to_store = false;
    return PNP_OK;
}

// =========================================================================================== weights

namespace {

struct Slot {
    float* small = nullptr;    // direct fp32 destination (biases, LN, embeddings)
    size_t small_off = 0;      // element offset inside `small`
    bool gemm = false;         // GEMM weight: staged fp32 until finalize
    int rows = 0, cols = 0;
};

// resolve a reference state-dict key
bool resolve(pnp_engine* e, const std::string& n, Slot& s) {
    const int D = e->D, H = e->H, I = e->I;
    auto small = [&](float* p, int count, size_t off = 0) { s.small = p; s.small_off = off; s.rows = 1; s.cols = count; return true; };
    auto gemm = [&](int r, int c) { s.gemm = true; s.rows = r; s.cols = c; return true; };
    if (n == "visual_encoder.cls_token") return small(e->cls, D);
    if (n == "visual_encoder.pos_embed") return small(e->pos, e->N * D);
    if (n == "visual_encoder.patch_embed.proj.weight") return gemm(D, 768);
    if (n == "visual_encoder.patch_embed.proj.bias") return small(e->patch_b, D);
    if (n == "visual_encoder.norm.weight") return small(e->vnorm_w, D);
    if (n == "visual_encoder.norm.bias") return small(e->vnorm_b, D);
    if (n == "text_encoder.embeddings.word_embeddings.weight") return small(e->word, e->c.vocab * H);
    if (n == "text_encoder.embeddings.position_embeddings.weight") return small(e->tpos, e->c.max_pos * H);
    if (n == "text_encoder.embeddings.LayerNorm.weight") return small(e->eln_w, H);
    if (n == "text_encoder.embeddings.LayerNorm.bias") return small(e->eln_b, H);
    if (n == "itm_head.weight") return small(e->itm_w, 2 * H);
    if (n == "itm_head.bias") return small(e->itm_b, 2);
    int li = -1;
    char rest[128];
    if (sscanf(n.c_str(), "visual_encoder.blocks.%d.%127s", &li, rest) == 2 && li >= 0 && li < e->c.vit_depth) {
        VitLayerW& v = e->vit[li];
        const std::string r(rest);
        const int F = D * e->c.vit_mlp_ratio;
        if (r == "norm1.weight") return small(v.n1w, D);
        if (r == "norm1.bias") return small(v.n1b, D);
        if (r == "norm2.weight") return small(v.n2w, D);
        if (r == "norm2.bias") return small(v.n2b, D);
        if (r == "attn.qkv.bias") return small(v.qkv_b, 3 * D);
        if (r == "attn.proj.bias") return small(v.proj_b, D);
        if (r == "mlp.fc1.bias") return small(v.fc1_b, F);
        if (r == "mlp.fc2.bias") return small(v.fc2_b, D);
        if (r == "attn.qkv.weight") return gemm(3 * D, D);
        if (r == "attn.proj.weight") return gemm(D, D);
        if (r == "mlp.fc1.weight") return gemm(F, D);
        if (r == "mlp.fc2.weight") return gemm(D, F);
        return false;
    }
    if (sscanf(n.c_str(), "text_encoder.encoder.layer.%d.%127s", &li, rest) == 2 && li >= 0 && li < e->TL) {
        TextLayerW& t = e->txt[li];
        const std::string r(rest);
        if (r == "attention.self.query.bias") return small(t.qkv_b, H, 0);
        if (r == "attention.self.key.bias") return small(t.qkv_b, H, H);
        if (r == "attention.self.value.bias") return small(t.qkv_b, H, 2 * H);
        if (r == "attention.output.dense.bias") return small(t.so_b, H);
        if (r == "attention.output.LayerNorm.weight") return small(t.sln_w, H);
        if (r == "attention.output.LayerNorm.bias") return small(t.sln_b, H);
        if (r == "crossattention.self.query.bias") return small(t.cq_b, H);
        if (r == "crossattention.self.key.bias") return small(e->ck_b, H, (size_t)li * H);
        if (r == "crossattention.self.value.bias") return small(e->cv_b, H, (size_t)li * H);
        if (r == "crossattention.output.dense.bias") return small(t.co_b, H);
        if (r == "crossattention.output.LayerNorm.weight") return small(t.cln_w, H);
        if (r == "crossattention.output.LayerNorm.bias") return small(t.cln_b, H);
        if (r == "intermediate.dense.bias") return small(t.i_b, I);
        if (r == "output.dense.bias") return small(t.o_b, H);
        if (r == "output.LayerNorm.weight") return small(t.oln_w, H);
        if (r == "output.LayerNorm.bias") return small(t.oln_b, H);
        if (r == "attention.self.query.weight" || r == "attention.self.key.weight" || r == "attention.self.value.weight" ||
            r == "attention.output.dense.weight" || r == "crossattention.self.query.weight" ||
            r == "crossattention.output.dense.weight")
            return gemm(H, H);
        if (r == "crossattention.self.key.weight" || r == "crossattention.self.value.weight") return gemm(H, D);
        if (r == "intermediate.dense.weight") return gemm(I, H);
        if (r == "output.dense.weight") return gemm(H, I);
        return false;
    }
    return false;
}

}  // namespace

extern "C" int pnp_load_weight(pnp_engine* e, const char* name, const float* data, const int64_t* shape, int32_t ndim,
                               int32_t on_device) {
    if (!e || !name || !data || !shape) return PNP_ERR_ARG;
    if (e->shares_weights) return fail(e, PNP_ERR_STATE, "this engine uses a donor's weights (pnp_create_shared)");
    if (e->finalized) return fail(e, PNP_ERR_STATE, "weights already finalized");
    HIPCHK(e, hipSetDevice(e->c.device));
    Slot s;
    if (!resolve(e, name, s)) return PNP_OK;          // strict=False: not a tensor of this path
    size_t count = 1;
    for (int i = 0; i < ndim; i++) count *= (size_t)shape[i];
    if (count != (size_t)s.rows * s.cols) return fail(e, PNP_ERR_ARG, "%s: expected %d elements, got %zu", name, s.rows * s.cols, count);
    const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    if (s.small) {
        HIPCHK(e, hipMemcpy(s.small + s.small_off, data, count * 4, kind));
    } else {
        Buf b;
        b.bytes = count * 4;
        HIPCHK(e, hipMalloc(&b.p, b.bytes));
        HIPCHK(e, hipMemcpy(b.p, data, b.bytes, kind));
        e->staging.push_back(b);
        e->named[std::string("stage:") + name] = b;
    }
    e->loaded[name] = true;
    return PNP_OK;
}

extern "C" int pnp_finalize_weights(pnp_engine* e) {
    if (!e) return PNP_ERR_ARG;
    if (e->finalized) return PNP_OK;
    HIPCHK(e, hipSetDevice(e->c.device));
    e->to_store = true;                        // everything allocated from here to the end of the call is a weight
    struct StoreOff { pnp_engine* e; ~StoreOff() { e->to_store = false; } } store_off{e};
    const int D = e->D, H = e->H, I = e->I, TL = e->TL;
    auto stage = [&](const std::string& n) -> const float* {
        auto it = e->named.find("stage:" + n);
        return it == e->named.end() ? nullptr : (const float*)it->second.p;
    };
#define NEED(ptr, nm)                                                         \
    const float* ptr = stage(nm);                                             \
    if (!ptr) return fail(e, PNP_ERR_STATE, "missing weight %s", std::string(nm).c_str());
    // every small tensor must have arrived too
    {
        const char* smalls[] = {"visual_encoder.cls_token", "visual_encoder.pos_embed", "visual_encoder.patch_embed.proj.bias",
                                "visual_encoder.norm.weight", "visual_encoder.norm.bias",
                                "text_encoder.embeddings.word_embeddings.weight", "text_encoder.embeddings.position_embeddings.weight",
                                "text_encoder.embeddings.LayerNorm.weight", "text_encoder.embeddings.LayerNorm.bias",
                                "itm_head.weight", "itm_head.bias"};
        for (const char* s : smalls)
            if (!e->loaded.count(s)) return fail(e, PNP_ERR_STATE, "missing weight %s", s);
    }
    {
        NEED(pw, "visual_encoder.patch_embed.proj.weight");
        KCHK(e, make_weight(e, pw, D, 768, false, &e->patch_w));
    }
    const int F = D * e->c.vit_mlp_ratio;
    for (int i = 0; i < e->c.vit_depth; i++) {
        const std::string b = "visual_encoder.blocks." + std::to_string(i) + ".";
        for (const char* k : kVitNames)
            if (!e->loaded.count(b + k)) return fail(e, PNP_ERR_STATE, "missing weight %s%s", b.c_str(), k);
        NEED(w0, b + "attn.qkv.weight");
        NEED(w1, b + "attn.proj.weight");
        NEED(w2, b + "mlp.fc1.weight");
        NEED(w3, b + "mlp.fc2.weight");
        KCHK(e, make_weight(e, w0, 3 * D, D, false, &e->vit[i].qkv_w, e->x3));
        KCHK(e, make_weight(e, w1, D, D, false, &e->vit[i].proj_w, e->x3));
        KCHK(e, make_weight(e, w2, F, D, false, &e->vit[i].fc1_w, e->x3));
        KCHK(e, make_weight(e, w3, D, F, false, &e->vit[i].fc2_w, e->x3));
    }
    // cross-attention K / V weights of all layers, layer-major, so one GEMM projects every layer
    KCHK(e, dalloc_t(e, &e->ck_w, (size_t)TL * H * D));
    KCHK(e, dalloc_t(e, &e->cv_w, (size_t)TL * H * D));
    for (int i = 0; i < TL; i++) {
        const std::string b = "text_encoder.encoder.layer." + std::to_string(i) + ".";
        TextLayerW& t = e->txt[i];
        const char* req[] = {"attention.self.query.bias", "attention.self.key.bias", "attention.self.value.bias",
                             "attention.output.dense.bias", "attention.output.LayerNorm.weight", "attention.output.LayerNorm.bias",
                             "crossattention.self.query.bias", "crossattention.self.key.bias", "crossattention.self.value.bias",
                             "crossattention.output.dense.bias", "crossattention.output.LayerNorm.weight",
                             "crossattention.output.LayerNorm.bias", "intermediate.dense.bias", "output.dense.bias",
                             "output.LayerNorm.weight", "output.LayerNorm.bias"};
        for (const char* k : req)
            if (!e->loaded.count(b + k)) return fail(e, PNP_ERR_STATE, "missing weight %s%s", b.c_str(), k);
        NEED(wq, b + "attention.self.query.weight");
        NEED(wk, b + "attention.self.key.weight");
        NEED(wv, b + "attention.self.value.weight");
        NEED(wso, b + "attention.output.dense.weight");
        NEED(wcq, b + "crossattention.self.query.weight");
        NEED(wck, b + "crossattention.self.key.weight");
        NEED(wcv, b + "crossattention.self.value.weight");
        NEED(wco, b + "crossattention.output.dense.weight");
        NEED(wi, b + "intermediate.dense.weight");
        NEED(wo, b + "output.dense.weight");
        // fused self q|k|v [3H, H]
        float* fused = nullptr;
        HIPCHK(e, hipMalloc((void**)&fused, (size_t)3 * H * H * 4));
        HIPCHK(e, hipMemcpy(fused, wq, (size_t)H * H * 4, hipMemcpyDeviceToDevice));
        HIPCHK(e, hipMemcpy(fused + (size_t)H * H, wk, (size_t)H * H * 4, hipMemcpyDeviceToDevice));
        HIPCHK(e, hipMemcpy(fused + (size_t)2 * H * H, wv, (size_t)H * H * 4, hipMemcpyDeviceToDevice));
        int r = make_weight(e, fused, 3 * H, H, false, &t.qkv_w, e->x3);
        if (r == PNP_OK && i > e->SL) r = make_weight(e, fused, 3 * H, H, true, &t.qkv_wT, e->x3);
        (void)hipFree(fused);
        if (r != PNP_OK) return r;
        KCHK(e, make_weight(e, wso, H, H, false, &t.so_w, e->x3));
        KCHK(e, make_weight(e, wcq, H, H, false, &t.cq_w, e->x3));
        KCHK(e, make_weight(e, wco, H, H, false, &t.co_w, e->x3));
        KCHK(e, make_weight(e, wi, I, H, false, &t.i_w, e->x3));
        KCHK(e, make_weight(e, wo, H, I, false, &t.o_w, e->x3));
        if (e->x3) {               // (hi | lo) halves over all layers: hi[TL*H*D] then lo[TL*H*D]
            const size_t half = (size_t)TL * H * D * 2, off = (size_t)i * H * D * 2;
            KCHK(e, split_f32(wck, (char*)e->ck_w + off, (char*)e->ck_w + half + off, (size_t)H * D, 0));
            KCHK(e, split_f32(wcv, (char*)e->cv_w + off, (char*)e->cv_w + half + off, (size_t)H * D, 0));
        } else {
            KCHK(e, cast_f32(e->bf, wck, (char*)e->ck_w + (size_t)i * H * D * e->esz, (size_t)H * D, 0));
            KCHK(e, cast_f32(e->bf, wcv, (char*)e->cv_w + (size_t)i * H * D * e->esz, (size_t)H * D, 0));
        }
        if (i >= e->SL) {
            KCHK(e, make_weight(e, wo, H, I, true, &t.o_wT, e->x3));      // [I][H]
            KCHK(e, make_weight(e, wi, I, H, true, &t.i_wT, e->x3));      // [H][I]
            KCHK(e, make_weight(e, wco, H, H, true, &t.co_wT, e->x3));
        }
        if (i > e->SL) {
            KCHK(e, make_weight(e, wcq, H, H, true, &t.cq_wT, e->x3));
            KCHK(e, make_weight(e, wso, H, H, true, &t.so_wT, e->x3));
        }
    }
#undef NEED
    HIPCHK(e, hipDeviceSynchronize());
    for (auto& b : e->staging) if (b.p) (void)hipFree(b.p);
    e->staging.clear();
    for (auto it = e->named.begin(); it != e->named.end();) {
        if (it->first.rfind("stage:", 0) == 0) it = e->named.erase(it);
        else ++it;
    }
    e->finalized = true;
    return PNP_OK;
}

// =========================================================================================== model

// embed: 0 = compute the token embeddings (the operator form) | 1 = compute them and keep a copy (drop iteration 0) | 2 = the
// images are those of the last embed-1 call with the patches of d_dropped zeroed: reuse the copy (embed_reuse_kernel)
static int vit_forward_impl(pnp_engine* e, const float* d_images, const uint8_t* d_dropped, int32_t B, void* stream, int embed);
extern "C" int pnp_vit_forward(pnp_engine* e, const float* d_images, const uint8_t* d_dropped, int32_t B, void* stream) {
    if (e) e->reuse.vit = false;                 // the operator form writes e->x: x0 no longer describes what follows
    return vit_forward_impl(e, d_images, d_dropped, B, stream, 0);
}
static int vit_forward_impl(pnp_engine* e, const float* d_images, const uint8_t* d_dropped, int32_t B, void* stream, int embed) {
    if (!e || !d_images) return PNP_ERR_ARG;
    if (!e->finalized) return fail(e, PNP_ERR_STATE, "weights not finalized");
    if (B <= 0 || B > e->c.max_batch) return fail(e, PNP_ERR_ARG, "batch %d out of range (max %d)", B, e->c.max_batch);
    hipStream_t s = (hipStream_t)stream;
    const int D = e->D, N = e->N, M = B * N, F = D * e->c.vit_mlp_ratio, bf = e->bf;
    const int ldv = e->c.max_batch * e->Npad;
    if (embed == 2 && !(e->reuse.vit && e->reuse.images == d_images && e->reuse.B == B)) embed = 0;   // stale: recompute
    if (embed == 2 && d_dropped) {
        hipLaunchKernelGGL(embed_reuse_kernel, dim3(2048), dim3(256), 0, s, (const float*)e->x0, d_dropped, (const float*)e->patch_b,
                           (const float*)e->pos, e->x, M, N, D / 4);
        if (hipGetLastError() != hipSuccess) return fail(e, PNP_ERR_HIP, "embed_reuse launch");
    } else {
    KCHK(e, patchify(bf, d_images, d_dropped, e->patches, B, e->c.img_size, e->P, s));
    KCHK(e, cls_rows(e->cls, e->pos, e->x, B, N, D, s));
    {
        GemmArgs g = G_(e->patches, 768, e->patch_w, 768, B * e->PP, D, 768);
        g.bias = e->patch_b; g.resid = e->pos; g.ldr = D; g.out_f32 = e->x; g.ldo = D; g.row_div = e->PP;
        KCHK(e, egemm(e, bf, g, s));
    }
    if (embed == 1) {
        HIPCHK(e, hipMemcpyAsync(e->x0, e->x, (size_t)M * D * sizeof(float), hipMemcpyDeviceToDevice, s));
        e->reuse.images = d_images;
        e->reuse.B = B;
        e->reuse.vit = true;
    }
    }
    const float scale = 1.0f / sqrtf(64.f);
    if (e->x3) {
        // split-bf16 ("bf16x3"): every Linear of the block is a wide-kernel launch on (hi, lo) bf16 operand pairs -- three
        // bf16 MFMA passes per product, fp32-class result; LayerNorm, the attention kernel and the GELU epilogue hand the
        // next GEMM its operand already split; attention runs in the split form of the bf16 kernel (vit_attn32_x3_kernel).
        const size_t cap = (size_t)e->c.max_batch * N;                    // row capacity of the activation buffers
        void* const xn_lo = (char*)e->xn + cap * D * 2;
        void* const ctx_lo = (char*)e->ctx + cap * D * 2;
        void* const h1_lo = (char*)e->h1 + cap * F * 2;
        void* const qk_lo = (char*)e->qk + cap * e->ldq * 2;
        for (int l = 0; l < e->c.vit_depth; l++) {
            const VitLayerW& w = e->vit[l];
            const char* const qkv_lo = (const char*)w.qkv_w + (size_t)3 * D * D * 2;
            KCHK(e, layernorm(1, e->x, w.n1w, w.n1b, e->c.vit_ln_eps, M, D, nullptr, e->xn, nullptr, nullptr, s, xn_lo));
            {   // q | k | v in one launch as a (hi, lo) bf16 pair: [M, 3D] each, halves of the qk buffer
                GemmArgs g = G_(e->xn, D, w.qkv_w, D, M, 3 * D, D);
                g.A_lo = xn_lo; g.B_lo = qkv_lo;
                g.bias = w.qkv_b; g.out_t = e->qk; g.out_lo = qk_lo; g.ldo_t = e->ldq;
                KCHK(e, egemm(e, 1, g, s));
            }
            KCHK(e, vit_attention_x3(e->qk, qk_lo, e->ldq, D, e->ctx, ctx_lo, B, e->c.vit_heads, N, scale, s));
            {
                GemmArgs g = G_(e->ctx, D, w.proj_w, D, M, D, D);
                g.A_lo = ctx_lo; g.B_lo = (const char*)w.proj_w + (size_t)D * D * 2;
                g.bias = w.proj_b; g.resid = e->x; g.ldr = D; g.out_f32 = e->x; g.ldo = D;
                KCHK(e, egemm(e, 1, g, s));
            }
            KCHK(e, layernorm(1, e->x, w.n2w, w.n2b, e->c.vit_ln_eps, M, D, nullptr, e->xn, nullptr, nullptr, s, xn_lo));
            {
                GemmArgs g = G_(e->xn, D, w.fc1_w, D, M, F, D);
                g.A_lo = xn_lo; g.B_lo = (const char*)w.fc1_w + (size_t)F * D * 2;
                g.bias = w.fc1_b; g.mode = GEMM_EPI_GELU; g.out_t = e->h1; g.out_lo = h1_lo; g.ldo_t = F;
                KCHK(e, egemm(e, 1, g, s));
            }
            {
                GemmArgs g = G_(e->h1, F, w.fc2_w, F, M, D, F);
                g.A_lo = h1_lo; g.B_lo = (const char*)w.fc2_w + (size_t)D * F * 2;
                g.bias = w.fc2_b; g.resid = e->x; g.ldr = D; g.out_f32 = e->x; g.ldo = D;
                KCHK(e, egemm(e, 1, g, s));
            }
        }
        KCHK(e, layernorm(1, e->x, e->vnorm_w, e->vnorm_b, e->c.vit_ln_eps, M, D, e->emb32, e->embT, nullptr, nullptr, s,
                          (char*)e->embT + cap * D * 2));
        return pnp_cross_kv(e, B, stream);
    }
    for (int l = 0; l < e->c.vit_depth; l++) {
        const VitLayerW& w = e->vit[l];
        KCHK(e, layernorm(bf, e->x, w.n1w, w.n1b, e->c.vit_ln_eps, M, D, nullptr, e->xn, nullptr, nullptr, s));
        if (bf) {   // q | k | v natural in one launch: [M, 3D]; the attention kernel transposes V on its LDS reads
            GemmArgs g = G_(e->xn, D, w.qkv_w, D, M, 3 * D, D);
            g.bias = w.qkv_b; g.out_t = e->qk; g.ldo_t = e->ldq;
            KCHK(e, egemm(e, bf, g, s));
            KCHK(e, vit_attention(bf, e->qk, e->ldq, D, (const char*)e->qk + (size_t)2 * D * e->esz, e->ldq, e->Npad, e->ctx, B,
                                  e->c.vit_heads, N, scale, s));
        } else {
        {   // q | k  natural: [M, 2D]
            GemmArgs g = G_(e->xn, D, w.qkv_w, D, M, 2 * D, D);
            g.bias = w.qkv_b; g.out_t = e->qk; g.ldo_t = 2 * D;
            KCHK(e, egemm(e, bf, g, s));
        }
        {   // V^T: [D, B*Npad] = Wv . xn^T, token columns padded per image
            GemmArgs g = G_((const char*)w.qkv_w + (size_t)2 * D * D * e->esz, D, e->xn, D, D, M, D);
            g.bias = w.qkv_b + 2 * D; g.bias_on_rows = 1; g.out_t = e->vt; g.ldo_t = ldv; g.col_div = N; g.col_pad = e->Npad;
            KCHK(e, egemm(e, bf, g, s));
        }
        KCHK(e, vit_attention(bf, e->qk, 2 * D, D, e->vt, ldv, e->Npad, e->ctx, B, e->c.vit_heads, N, scale, s));
        }
        {
            GemmArgs g = G_(e->ctx, D, w.proj_w, D, M, D, D);
            g.bias = w.proj_b; g.resid = e->x; g.ldr = D; g.out_f32 = e->x; g.ldo = D;
            KCHK(e, egemm(e, bf, g, s));
        }
        KCHK(e, layernorm(bf, e->x, w.n2w, w.n2b, e->c.vit_ln_eps, M, D, nullptr, e->xn, nullptr, nullptr, s));
        {
            GemmArgs g = G_(e->xn, D, w.fc1_w, D, M, F, D);
            g.bias = w.fc1_b; g.mode = GEMM_EPI_GELU; g.out_t = e->h1; g.ldo_t = F;
            KCHK(e, egemm(e, bf, g, s));
        }
        {
            GemmArgs g = G_(e->h1, F, w.fc2_w, F, M, D, F);
            g.bias = w.fc2_b; g.resid = e->x; g.ldr = D; g.out_f32 = e->x; g.ldo = D;
            KCHK(e, egemm(e, bf, g, s));
        }
    }
    KCHK(e, layernorm(bf, e->x, e->vnorm_w, e->vnorm_b, e->c.vit_ln_eps, M, D, e->emb32, e->embT, nullptr, nullptr, s));
    return pnp_cross_kv(e, B, stream);
}

// encoder_hidden_states -> key / value of every text layer's cross-attention (B/med.py:208-211): the reference
// projects image_embeds once per layer inside BertSelfAttention.forward; here all 12 layers are projected in four
// GEMMs (natural + transposed layouts) right after the ViT, from the compute-type copy of image_embeds.
extern "C" int pnp_cross_kv(pnp_engine* e, int32_t B, void* stream) {
    if (!e) return PNP_ERR_ARG;
    if (!e->finalized) return fail(e, PNP_ERR_STATE, "weights not finalized");
    if (B <= 0 || B > e->c.max_batch) return fail(e, PNP_ERR_ARG, "batch %d out of range (max %d)", B, e->c.max_batch);
    hipStream_t s = (hipStream_t)stream;
    const int D = e->D, N = e->N, M = B * N, bf = e->bf;
    const int ldv = e->c.max_batch * e->Npad;
    const int H = e->H, TL = e->TL, SL = e->SL;
    if (e->x3) {
        const size_t cap = (size_t)e->c.max_batch * N, whalf = (size_t)TL * H * D * 2;
        const void* const emb_lo = (const char*)e->embT + cap * D * 2;
        const char *ck = (const char*)e->ck_w, *cv = (const char*)e->cv_w;
        {
            GemmArgs g = G_(e->embT, D, ck, D, M, TL * H, D);
            g.A_lo = emb_lo; g.B_lo = ck + whalf;
            g.bias = e->ck_b; g.out_f32 = (float*)e->Knat; g.ldo = TL * H;
            KCHK(e, egemm(e, 1, g, s));
        }
        {
            GemmArgs g = G_(cv, D, e->embT, D, TL * H, M, D);
            g.A_lo = cv + whalf; g.B_lo = emb_lo;
            g.bias = e->cv_b; g.bias_on_rows = 1; g.out_f32 = (float*)e->Vt; g.ldo = ldv; g.col_div = N; g.col_pad = e->Npad;
            KCHK(e, egemm(e, 1, g, s));
        }
        {   // V token-major of layers >= SL from V^T, K^T of layers > SL from K token-major (see feat_to_tok_kernel)
            const int nVn = TL - SL, R = nVn * H;
            hipLaunchKernelGGL(feat_to_tok_kernel, dim3((R + 31) / 32, (N + 31) / 32, B), dim3(32, 8), 0, s,
                               (const float*)e->Vt + (size_t)SL * H * ldv, ldv, e->Npad, (float*)e->Vnat, R, N);
            const int nKt = TL - SL - 1;
            if (nKt > 0)
                hipLaunchKernelGGL(tok_to_feat_kernel, dim3((nKt * H + 31) / 32, (N + 31) / 32, B), dim3(32, 8), 0, s,
                                   (const float*)e->Knat + (size_t)(SL + 1) * H, TL * H, (float*)e->Kt, ldv, e->Npad, nKt * H, N);
            if (hipGetLastError() != hipSuccess) return fail(e, PNP_ERR_HIP, "cross K / V layout kernels failed to launch");
        }
        return PNP_OK;
    }
    {
        GemmArgs g = G_(e->embT, D, e->ck_w, D, M, TL * H, D);
        g.bias = e->ck_b; g.out_t = e->Knat; g.ldo_t = TL * H;
        KCHK(e, egemm(e, bf, g, s));
    }
    {
        GemmArgs g = G_(e->cv_w, D, e->embT, D, TL * H, M, D);
        g.bias = e->cv_b; g.bias_on_rows = 1; g.out_t = e->Vt; g.ldo_t = ldv; g.col_div = N; g.col_pad = e->Npad;
        KCHK(e, egemm(e, bf, g, s));
    }
    {
        const int nVn = TL - SL;
        GemmArgs g = G_(e->embT, D, (const char*)e->cv_w + (size_t)SL * H * D * e->esz, D, M, nVn * H, D);
        g.bias = e->cv_b + (size_t)SL * H; g.out_t = e->Vnat; g.ldo_t = nVn * H;
        KCHK(e, egemm(e, bf, g, s));
    }
    if (TL - SL - 1 > 0) {
        const int nKt = TL - SL - 1;
        GemmArgs g = G_((const char*)e->ck_w + (size_t)(SL + 1) * H * D * e->esz, D, e->embT, D, nKt * H, M, D);
        g.bias = e->ck_b + (size_t)(SL + 1) * H; g.bias_on_rows = 1; g.out_t = e->Kt; g.ldo_t = ldv; g.col_div = N; g.col_pad = e->Npad;
        KCHK(e, egemm(e, bf, g, s));
    }
    return PNP_OK;
}

// reuse_prefix: the token ids / mask are those of the previous call on this engine (drop iterations 1..): the embeddings and the
// self-attention sub-layer of text layer 0 do not see the image -- their activations of the previous call are still in place
static int text_forward_impl(pnp_engine* e, const int64_t* d_ids, const int64_t* d_mask, int32_t ld, int32_t B, int32_t L,
                             float* d_logits, void* stream, bool reuse_prefix);
extern "C" int pnp_text_forward_xattn(pnp_engine* e, const int64_t* d_ids, const int64_t* d_mask, int32_t ld, int32_t B,
                                      int32_t L, float* d_logits, void* stream) {
    if (e) e->reuse.text = false;
    return text_forward_impl(e, d_ids, d_mask, ld, B, L, d_logits, stream, false);
}
static int text_forward_impl(pnp_engine* e, const int64_t* d_ids, const int64_t* d_mask, int32_t ld, int32_t B, int32_t L,
                             float* d_logits, void* stream, bool reuse_prefix) {
    if (!e || !d_ids || !d_mask) return PNP_ERR_ARG;
    if (!e->finalized) return fail(e, PNP_ERR_STATE, "weights not finalized");
    if (B <= 0 || B > e->c.max_batch || L < 5 || L > e->c.max_text_len || ld < L)
        return fail(e, PNP_ERR_ARG, "text batch %d x %d (ld %d) out of range (max %d x %d)", B, L, ld, e->c.max_batch, e->c.max_text_len);
    hipStream_t s = (hipStream_t)stream;
    const int H = e->H, I = e->I, TL = e->TL, R = B * L, bf = e->bf, N = e->N, D = e->D;
    const int ldv = e->c.max_batch * e->Npad;
    (void)D;
    if (reuse_prefix && !(e->reuse.text && e->reuse.ids == d_ids && e->reuse.mask == d_mask && e->reuse.B == B && e->reuse.L == L &&
                          e->reuse.ld == ld))
        reuse_prefix = false;                    // not the captions whose prefix is in place: compute it
    if (!reuse_prefix) {
        KCHK(e, text_embed(d_ids, ld, e->word, e->tpos, e->temb, B, L, H, e->c.enc_token_id, e->c.vocab, s));
        KCHK(e, layernorm(bf, e->temb, e->eln_w, e->eln_b, e->c.txt_ln_eps, R, H, e->h0, e->h0T, nullptr, nullptr, s));
    }
    const float* h = e->h0;
    const void* hT = e->h0T;
    for (int i = 0; i < TL; i++) {
        const TextLayerW& w = e->txt[i];
        TextLayerA& a = e->ta[i];
        const bool stash = i >= e->SL;
        if (!(reuse_prefix && i == 0)) {
        {
            GemmArgs g = G_(hT, H, w.qkv_w, H, R, 3 * H, H);
            g.bias = w.qkv_b; g.out_t = a.qkv; g.ldo_t = 3 * H;
            KCHK(e, tgemm(e, g, s));
        }
        KCHK(e, text_self_attn(bf, a.qkv, d_mask, ld, e->ctx_s, stash ? a.Ps : nullptr, e->dS, B, L, H, s));
        {
            GemmArgs g = G_(e->ctx_s, H, w.so_w, H, R, H, H);
            g.bias = w.so_b; g.resid = h; g.ldr = H; g.out_f32 = e->tmp; g.ldo = H;
            KCHK(e, tgemm(e, g, s));
        }
        KCHK(e, layernorm(bf, e->tmp, w.sln_w, w.sln_b, e->c.txt_ln_eps, R, H, a.a_out, a.a_outT, a.a_hat, a.a_rstd, s));
        }
        {
            GemmArgs g = G_(a.a_outT, H, w.cq_w, H, R, H, H);
            g.bias = w.cq_b; g.out_t = a.qc; g.ldo_t = H;
            KCHK(e, tgemm(e, g, s));
        }
        KCHK(e, xattn(bf, 0, (const char*)e->Knat + (size_t)i * H * e->esz, TL * H,
                      (const char*)e->Vt + (size_t)i * H * ldv * e->esz, ldv, e->Npad, a.qc, H, e->ctx_c, H,
                      stash ? a.Pc : nullptr, e->Nst, B, L, N, e->nh, s));
        {
            GemmArgs g = G_(e->ctx_c, H, w.co_w, H, R, H, H);
            g.bias = w.co_b; g.resid = a.a_out; g.ldr = H; g.out_f32 = e->tmp; g.ldo = H;
            KCHK(e, tgemm(e, g, s));
        }
        KCHK(e, layernorm(bf, e->tmp, w.cln_w, w.cln_b, e->c.txt_ln_eps, R, H, a.c_out, a.c_outT, a.c_hat, a.c_rstd, s));
        {
            GemmArgs g = G_(a.c_outT, H, w.i_w, H, R, I, H);
            g.bias = w.i_b; g.mode = GEMM_EPI_GELU; g.aux = a.u; g.ld_aux = I; g.out_t = e->g; g.ldo_t = I;
            KCHK(e, tgemm(e, g, s));
        }
        {
            GemmArgs g = G_(e->g, I, w.o_w, I, R, H, I);
            g.bias = w.o_b; g.resid = a.c_out; g.ldr = H; g.out_f32 = e->tmp; g.ldo = H;
            KCHK(e, tgemm(e, g, s));
        }
        KCHK(e, layernorm(bf, e->tmp, w.oln_w, w.oln_b, e->c.txt_ln_eps, R, H, a.h_out, a.h_outT, a.o_hat, a.o_rstd, s));
        h = a.h_out;
        hT = a.h_outT;
    }
    KCHK(e, itm_head(h, e->itm_w, e->itm_b, d_logits ? d_logits : e->logits_scratch, B, L, H, s));
    return PNP_OK;
}

extern "C" int pnp_xattn_grad(pnp_engine* e, int32_t B, int32_t L, void* stream) {
    return pnp_xattn_grad_layer(e, B, L, e ? e->SL : 0, stream);
}

extern "C" int pnp_xattn_grad_layer(pnp_engine* e, int32_t B, int32_t L, int32_t layer, void* stream) {
    if (!e) return PNP_ERR_ARG;
    if (!e->finalized) return fail(e, PNP_ERR_STATE, "weights not finalized");
    if (B <= 0 || B > e->c.max_batch || L < 5 || L > e->c.max_text_len) return fail(e, PNP_ERR_ARG, "bad B/L");
    if (layer < e->SL || layer >= e->TL)
        return fail(e, PNP_ERR_ARG, "layer %d: cross-attention maps are kept for text layers %d..%d (stash_layer)", layer, e->SL, e->TL - 1);
    hipStream_t s = (hipStream_t)stream;
    const int H = e->H, I = e->I, TL = e->TL, R = B * L, bf = e->bf, N = e->N, SL = e->SL;
    const int ldv = e->c.max_batch * e->Npad, nVn = TL - SL;
    e->grad_layer = layer;
    KCHK(e, itm_grad_seed(e->itm_w, e->dh, B, L, H, s));
    for (int i = TL - 1; i >= layer; i--) {
        const TextLayerW& w = e->txt[i];
        TextLayerA& a = e->ta[i];
        KCHK(e, layernorm_bwd(bf, e->dh, w.oln_w, a.o_hat, a.o_rstd, R, H, e->d_pre, e->d_preT, s));
        {   // dg = (d_pre . Wo2) * gelu'(u)
            GemmArgs g = G_(e->d_preT, H, w.o_wT, H, R, I, H);
            g.mode = GEMM_EPI_GELU_GRAD; g.aux = a.u; g.ld_aux = I; g.out_t = e->dg; g.ldo_t = I;
            KCHK(e, tgemm(e, g, s));
        }
        {   // dc = d_pre + dg . Wi
            GemmArgs g = G_(e->dg, I, w.i_wT, I, R, H, I);
            g.resid = e->d_pre; g.ldr = H; g.out_f32 = e->dc; g.ldo = H;
            KCHK(e, tgemm(e, g, s));
        }
        KCHK(e, layernorm_bwd(bf, e->dc, w.cln_w, a.c_hat, a.c_rstd, R, H, e->d_cpre, e->d_cpreT, s));
        {
            GemmArgs g = G_(e->d_cpreT, H, w.co_wT, H, R, H, H);
            g.out_t = e->dctxc; g.ldo_t = H;
            KCHK(e, tgemm(e, g, s));
        }
        const char* vnat = (const char*)e->Vnat + (size_t)(i - SL) * H * e->esz;
        if (i == layer) {
            KCHK(e, xattn(bf, 2, vnat, nVn * H, nullptr, 0, e->Npad, e->dctxc, H, nullptr, 0, e->dPc, e->Nst, B, L, N, e->nh, s));
            break;
        }
        KCHK(e, xattn(bf, 1, vnat, nVn * H, (const char*)e->Kt + (size_t)(i - SL - 1) * H * ldv * e->esz, ldv, e->Npad,
                      e->dctxc, H, e->dqc, H, a.Pc, e->Nst, B, L, N, e->nh, s));
        {
            GemmArgs g = G_(e->dqc, H, w.cq_wT, H, R, H, H);
            g.resid = e->d_cpre; g.ldr = H; g.out_f32 = e->da; g.ldo = H;
            KCHK(e, tgemm(e, g, s));
        }
        KCHK(e, layernorm_bwd(bf, e->da, w.sln_w, a.a_hat, a.a_rstd, R, H, e->d_apre, e->d_apreT, s));
        {
            GemmArgs g = G_(e->d_apreT, H, w.so_wT, H, R, H, H);
            g.out_f32 = e->dctx_s; g.ldo = H;
            KCHK(e, tgemm(e, g, s));
        }
        KCHK(e, text_self_attn_bwd(bf, a.qkv, e->dctx_s, a.Ps, e->dS, e->dqkv, B, L, H, s));
        {
            GemmArgs g = G_(e->dqkv, 3 * H, w.qkv_wT, 3 * H, R, H, 3 * H);
            g.resid = e->d_apre; g.ldr = H; g.out_f32 = e->dh; g.ldo = H;
            KCHK(e, tgemm(e, g, s));
        }
    }
    return PNP_OK;
}

extern "C" int pnp_gradcam_gather(pnp_engine* e, const int64_t* d_mask, int32_t ld, int32_t B, int32_t L, int32_t head,
                                  float* d_out, void* stream) {
    if (!e || !d_mask || !d_out) return PNP_ERR_ARG;
    if (head < 0 || head >= e->nh) return fail(e, PNP_ERR_ARG, "head %d out of range", head);
    if (B <= 0 || B > e->c.max_batch || L < 5 || L > e->c.max_text_len || ld < L) return fail(e, PNP_ERR_ARG, "bad B/L/ld");
    if (e->grad_layer < 0) return fail(e, PNP_ERR_STATE, "call pnp_xattn_grad first");
    // P of the layer the last backward stopped at, and its dL/dP
    KCHK(e, gradcam_gather(e->ta[e->grad_layer].Pc, e->dPc, d_mask, ld, d_out, B, e->nh, head, L, e->Nst, e->PP, (hipStream_t)stream));
    return PNP_OK;
}

static int compute_gradcam_impl(pnp_engine* e, const float* d_images, const uint8_t* d_dropped, const int64_t* d_ids,
                                const int64_t* d_mask, int32_t ld, int32_t B, int32_t L, int32_t layer, int32_t head,
                                float* d_out, float* d_logits, void* stream, int embed);
extern "C" int pnp_compute_gradcam_layer(pnp_engine* e, const float* d_images, const uint8_t* d_dropped, const int64_t* d_ids,
                                         const int64_t* d_mask, int32_t ld, int32_t B, int32_t L, int32_t layer, int32_t head,
                                         float* d_out, float* d_logits, void* stream) {
    return compute_gradcam_impl(e, d_images, d_dropped, d_ids, d_mask, ld, B, L, layer, head, d_out, d_logits, stream, 0);
}
static int compute_gradcam_impl(pnp_engine* e, const float* d_images, const uint8_t* d_dropped, const int64_t* d_ids,
                                const int64_t* d_mask, int32_t ld, int32_t B, int32_t L, int32_t layer, int32_t head,
                                float* d_out, float* d_logits, void* stream, int embed) {
    int r = vit_forward_impl(e, d_images, d_dropped, B, stream, embed);
    if (r) return r;
    // drop iterations 1..: same captions as iteration 0 (embed == 2 only comes from the drop loop; a stash layer of 0 keeps the
    // probabilities of layer 0's self-attention, which are in place as well)
    r = text_forward_impl(e, d_ids, d_mask, ld, B, L, d_logits, stream, embed == 2);
    if (r) return r;
    if (embed == 1) {                            // the prefix now in place belongs to these captions
        e->reuse.ids = d_ids; e->reuse.mask = d_mask; e->reuse.L = L; e->reuse.ld = ld;
        e->reuse.text = true;
    } else if (embed == 0) {
        e->reuse.text = false;
    }
    r = pnp_xattn_grad_layer(e, B, L, layer, stream);
    if (r) return r;
    return pnp_gradcam_gather(e, d_mask, ld, B, L, head, d_out, stream);
}

extern "C" int pnp_compute_gradcam(pnp_engine* e, const float* d_images, const uint8_t* d_dropped, const int64_t* d_ids,
                                   const int64_t* d_mask, int32_t ld, int32_t B, int32_t L, int32_t head, float* d_out,
                                   float* d_logits, void* stream) {
    return pnp_compute_gradcam_layer(e, d_images, d_dropped, d_ids, d_mask, ld, B, L, e ? e->SL : 0, head, d_out, d_logits, stream);
}

extern "C" int pnp_drop_step(pnp_engine* e, const float* d_gradcam, float* d_g0, float* d_agg, uint8_t* d_dropped,
                             int32_t* d_picks, int32_t iter, int32_t B, int32_t T, int32_t npick, int32_t max_picks,
                             void* stream) {
    if (!e || !d_gradcam || !d_agg || !d_dropped || (npick > 0 && !d_picks)) return PNP_ERR_ARG;
    if (T < 4) return fail(e, PNP_ERR_ARG, "T=%d too short", T);
    KCHK(e, drop_step(d_gradcam, d_g0, d_agg, d_dropped, d_picks, iter, B, T, e->PP, npick, max_picks, (hipStream_t)stream));
    return PNP_OK;
}

extern "C" int pnp_drop_loop(pnp_engine* e, const float* d_images, const int64_t* d_ids, const int64_t* d_mask, int32_t ld,
                             int32_t B, int32_t L, int32_t head, int32_t drop_iter, int32_t npick, float* d_g0,
                             float* d_agg, int32_t* d_picks, float* d_logits, void* stream) {
    return pnp_drop_loop_layer(e, d_images, d_ids, d_mask, ld, B, L, e ? e->SL : 0, head, drop_iter, npick, d_g0, d_agg, d_picks,
                               d_logits, stream);
}

extern "C" int pnp_drop_loop_layer(pnp_engine* e, const float* d_images, const int64_t* d_ids, const int64_t* d_mask, int32_t ld,
                                   int32_t B, int32_t L, int32_t layer, int32_t head, int32_t drop_iter, int32_t npick, float* d_g0,
                                   float* d_agg, int32_t* d_picks, float* d_logits, void* stream) {
    if (!e || !d_images || !d_ids || !d_mask || !d_g0) return PNP_ERR_ARG;
    if (drop_iter < 1) return fail(e, PNP_ERR_ARG, "drop_iter must be >= 1");
    if (drop_iter == 1)   // PnP.py:565-575: single call, no aggregate
        return pnp_compute_gradcam_layer(e, d_images, nullptr, d_ids, d_mask, ld, B, L, layer, head, d_g0, d_logits, stream);
    if (!d_agg || !d_picks) return PNP_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(e, hipMemsetAsync(e->dropped, 0, (size_t)B * e->PP, s));
    for (int it = 0; it < drop_iter; it++) {
        // the token embeddings of iteration 0 serve the later iterations (same images, more patches zeroed)
        int r = compute_gradcam_impl(e, d_images, e->dropped, d_ids, d_mask, ld, B, L, layer, head, e->G, d_logits, stream, it == 0 ? 1 : 2);
        if (r) return r;
        r = pnp_drop_step(e, e->G, d_g0, d_agg, e->dropped, d_picks, it, B, L - 1, npick, drop_iter * npick, stream);
        if (r) return r;
    }
    return PNP_OK;
}

// =========================================================================================== post-process

extern "C" int pnp_post_reserve(pnp_engine* e, int32_t max_batch, int64_t max_total_pixels, int32_t max_pixels_per_image,
                                int32_t max_channels, int32_t crf_chunk) {
    if (!e) return PNP_ERR_ARG;
    if (e->post.reserved) return fail(e, PNP_ERR_STATE, "post-process workspace already reserved");
    if (max_batch <= 0 || max_batch > 64 || max_total_pixels <= 0 || max_pixels_per_image <= 0 || max_channels <= 0 ||
        max_channels > 255)
        return fail(e, PNP_ERR_ARG, "bad post-process bounds (at most 64 images per batch, 255 channels)");
    HIPCHK(e, hipSetDevice(e->c.device));
    auto& p = e->post;
    p.maxB = max_batch;
    p.max_total_pix = max_total_pixels;
    p.max_pix_img = max_pixels_per_image;
    p.maxK = max_channels;
    p.chunk = crf_chunk > 0 ? crf_chunk : max_batch;
    const size_t TP = (size_t)max_total_pixels, K = max_channels, B = max_batch;
    KCHK(e, dalloc(e, &p.d_desc, B));
    KCHK(e, dalloc(e, &p.d_desc_pair, B));
    KCHK(e, dalloc(e, &p.d_img_cls_off, B + 1));
    p.cls_cap = (int)(B * K + 1);
    p.tok_cap = (int)(B * e->c.max_text_len + 1);
    KCHK(e, dalloc(e, &p.d_cls_off, p.cls_cap + 1));
    KCHK(e, dalloc(e, &p.d_cls_div, p.cls_cap));
    KCHK(e, dalloc(e, &p.d_tok_idx, p.tok_cap));
    p.lut_cap = (int)(B * (K + 1));
    KCHK(e, dalloc(e, &p.d_lut, p.lut_cap));
    KCHK(e, dalloc(e, &p.d_label_off, B + 1));
    p.wts_cap = (int)(B * 4096);
    KCHK(e, dalloc(e, &p.d_wts, p.wts_cap));
    KCHK(e, dalloc(e, &p.d_wt_off, B + 1));
    KCHK(e, dalloc(e, &p.merged, B * K * e->PP));
    KCHK(e, dalloc(e, &p.thr, B * K * e->PP));
    KCHK(e, dalloc(e, &p.maps, TP * K));
    KCHK(e, dalloc(e, &p.maps2, TP * K));
    KCHK(e, dalloc(e, &p.maps3, TP * K));
    KCHK(e, dalloc(e, &p.stats, B * K * 2));
    // CRF
    const size_t Kp = (K + 3) / 4 * 4;
    p.maxKp = (int)Kp;
    // the paired 1-drop | N-drop run keeps two channel groups per row: size the CRF arrays for it unless that would
    // take more than a third of the free device memory (many classes x large images); pnp_postprocess_pair then runs
    // the two branches one after the other
    {
        const size_t chunk_pix0 = (size_t)std::min<int64_t>((int64_t)(crf_chunk > 0 ? crf_chunk : B) * max_pixels_per_image, max_total_pixels);
        const size_t single = (2 * TP * Kp + 2 * chunk_pix0 * 6 * Kp + 2 * chunk_pix0 * 3 * Kp) * sizeof(float);
        // ... of what the device has free right now (the rest of this function needs about as much again for maps and lattices)
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = (size_t)96 << 30;
        p.groups_cap = 2 * single <= free_b / 3 ? 2 : 1;
    }
    const size_t G2 = (size_t)p.groups_cap;
    KCHK(e, dalloc(e, &p.unary, TP * Kp * G2));
    KCHK(e, dalloc(e, &p.Q, TP * Kp * G2));
    KCHK(e, dalloc(e, &p.norm[0], TP));
    KCHK(e, dalloc(e, &p.norm[1], TP));
    for (int t = 0; t < 2; t++) {
        const int D1 = t == 0 ? 3 : 6;
        const size_t cap = TP * D1;
        p.cap[t] = cap;
        CrfLattice& L = p.lat[t];
        L.D1 = D1;
        L.which = t;
        L.cap = cap;
        KCHK(e, dalloc(e, &L.bary, cap));
        KCHK(e, dalloc(e, &L.vals, cap));
        KCHK(e, dalloc(e, &L.ent, cap));
        KCHK(e, dalloc(e, &L.offset, cap));
        KCHK(e, dalloc(e, &L.seg_start, cap + 1));
        KCHK(e, dalloc(e, &L.seg_lo, cap));
        KCHK(e, dalloc(e, &L.seg_hi, cap));
        KCHK(e, dalloc(e, &L.ukeys, cap));
        KCHK(e, dalloc(e, &L.idbase, B + 1));
        KCHK(e, dalloc(e, &L.n1, cap * D1));
        KCHK(e, dalloc(e, &L.n2, cap * D1));
        KCHK(e, dalloc(e, &L.nbr8, cap * (D1 / 2)));
    }
    const size_t cap6 = p.cap[1];
    KCHK(e, dalloc(e, &p.keys_a, cap6));
    KCHK(e, dalloc(e, &p.keys_b, cap6));
    KCHK(e, dalloc(e, &p.vals_a, cap6));
    KCHK(e, dalloc(e, &p.head, cap6));
    KCHK(e, dalloc(e, &p.incl, cap6));
    KCHK(e, dalloc(e, &p.n1k, cap6 * 6));
    KCHK(e, dalloc(e, &p.n2k, cap6 * 6));
    KCHK(e, dalloc(e, &p.range_err, 1, true));
    p.sort_tmp_bytes = crf_sort_temp_bytes(cap6, max_batch);
    char* st = nullptr;
    KCHK(e, dalloc(e, &st, p.sort_tmp_bytes));
    p.sort_tmp = st;
    // lattice value buffers: the images of one chunk, bilateral upper bound (6 entries per pixel) x K
    const size_t chunk_pix = (size_t)std::min<int64_t>((int64_t)p.chunk * max_pixels_per_image, max_total_pixels);
    p.val_cap = std::max(chunk_pix * 6 * Kp * G2, cap6);
    KCHK(e, dalloc(e, &p.va, p.val_cap));
    KCHK(e, dalloc(e, &p.vb, p.val_cap));
    p.valg_cap = chunk_pix * 3 * Kp * G2;
    KCHK(e, dalloc(e, &p.vga, p.valg_cap));
    KCHK(e, dalloc(e, &p.vgb, p.valg_cap));
    p.reserved = true;
    return PNP_OK;
}

extern "C" int pnp_post_prepare(pnp_engine* e, const pnp_post_batch* b, int32_t want_crf, void* stream) {
    if (!e || !b) return PNP_ERR_ARG;
    auto& p = e->post;
    if (!p.reserved) return fail(e, PNP_ERR_STATE, "call pnp_post_reserve first");
    if (b->B <= 0 || b->B > p.maxB) return fail(e, PNP_ERR_ARG, "post batch %d out of range", b->B);
    if (!b->H || !b->W || !b->n_classes || !b->has_bg || !b->img_cls_off || !b->cls_off || !b->tok_idx || !b->cls_div || !b->lut)
        return fail(e, PNP_ERR_ARG, "post batch has null tables");
    if (want_crf && !b->d_rgb) return fail(e, PNP_ERR_ARG, "CRF needs d_rgb");
    hipStream_t s = (hipStream_t)stream;
    const int B = b->B;
    p.prepared = false;
    p.desc.assign(B, PostDesc{});
    // host staging of the small tables lives in the engine until the next prepare (pageable memory: the runtime copies it
    // into its own staging buffer at enqueue time, so rewriting it on the next call is safe without a synchronisation)
    std::vector<size_t>& label_off = p.h_label_off;
    std::vector<int32_t>& wt_off = p.h_wt_off;
    std::vector<double>& wts = p.h_wts;
    label_off.assign(B + 1, 0);
    wt_off.assign(B + 1, 0);
    wts.clear();
    size_t off = 0, qoff = 0;
    int64_t pix = 0;
    p.Cmax = p.Kmax = p.maxHW = p.Kpmax = p.maxH = p.maxW = p.max_radius = 0;
    for (int i = 0; i < B; i++) {
        PostDesc& d = p.desc[i];
        d.H = b->H[i]; d.W = b->W[i]; d.C = b->n_classes[i]; d.has_bg = b->has_bg[i] ? 1 : 0; d.K = d.C + d.has_bg;
        if (d.H <= 0 || d.W <= 0 || d.C <= 0 || d.K > p.maxK) return fail(e, PNP_ERR_ARG, "image %d: bad H/W/classes (K=%d max %d)", i, d.K, p.maxK);
        if ((int64_t)d.H * d.W > p.max_pix_img) return fail(e, PNP_ERR_ARG, "image %d: %dx%d exceeds max_pixels_per_image", i, d.H, d.W);
        if (b->img_cls_off[i + 1] - b->img_cls_off[i] != d.C) return fail(e, PNP_ERR_ARG, "image %d: merge plan has %d classes, expected %d", i, b->img_cls_off[i + 1] - b->img_cls_off[i], d.C);
        d.pix0 = (int)pix;
        d.off = off;
        d.Kp = (d.K + 3) / 4 * 4;
        d.G = 1;
        d.Kg = d.K;
        d.qoff = qoff;
        label_off[i] = (size_t)pix;
        off += (size_t)d.K * d.H * d.W;
        qoff += (size_t)d.Kp * d.H * d.W;
        p.Kpmax = std::max(p.Kpmax, d.Kp);
        p.maxH = std::max(p.maxH, d.H);
        p.maxW = std::max(p.maxW, d.W);
        pix += (int64_t)d.H * d.W;
        p.Cmax = std::max(p.Cmax, d.C);
        p.Kmax = std::max(p.Kmax, d.K);
        p.maxHW = std::max(p.maxHW, d.H * d.W);
        // scipy _gaussian_kernel1d(sigma = 0.05 * max(H, W)), truncate 4.0  (PnP.py:1150, scale=0.05)
        const double sigma = 0.05 * (double)std::max(d.H, d.W);
        const int radius = (int)(4.0 * sigma + 0.5);
        std::vector<double> phi(2 * radius + 1);
        double sum = 0;
        for (int x = -radius; x <= radius; x++) {
            phi[x + radius] = std::exp(-0.5 / (sigma * sigma) * (double)(x * x));
            sum += phi[x + radius];
        }
        p.max_radius = std::max(p.max_radius, (b->blur_wts && b->blur_wt_off) ? b->blur_wt_off[i + 1] - b->blur_wt_off[i] - 1 : radius);
        wt_off[i] = (int32_t)wts.size();
        if (b->blur_wts && b->blur_wt_off) {
            for (int j = b->blur_wt_off[i]; j < b->blur_wt_off[i + 1]; j++) wts.push_back(b->blur_wts[j]);
        } else {
            for (int j = 0; j <= radius; j++) wts.push_back(phi[radius - j] / sum);   // weight at distance j (w[-j] == w[j])
        }
    }
    wt_off[B] = (int32_t)wts.size();
    label_off[B] = (size_t)pix;
    if (pix > p.max_total_pix) return fail(e, PNP_ERR_ARG, "batch has %lld pixels, reserved %lld", (long long)pix, (long long)p.max_total_pix);
    if ((int)wts.size() > p.wts_cap) return fail(e, PNP_ERR_ARG, "blur taps exceed reserved capacity");
    const int ncls = b->img_cls_off[B], ntok = b->cls_off[ncls];
    if (ncls > p.cls_cap || ntok > p.tok_cap) return fail(e, PNP_ERR_ARG, "merge plan exceeds reserved capacity");
    if (b->lut_stride < p.Kmax || B * b->lut_stride > p.lut_cap) return fail(e, PNP_ERR_ARG, "bad lut_stride");
    // lattice value offsets per chunk (upper bound: entries * K)
    // the paired run (two channel groups per row, 1-drop | N-drop) packs the groups back to back: rows of
    // 4 * ceil(2 K / 4) floats (K = 21: 44 instead of 2 x 24), its own row-strided offsets
    std::vector<size_t> voff_pair(2 * (size_t)B, 0);
    for (int c0 = 0; c0 < B; c0 += p.chunk) {
        size_t v[2] = {0, 0}, v2[2] = {0, 0};
        for (int i = c0; i < std::min(B, c0 + p.chunk); i++)
            for (int t = 0; t < 2; t++) {
                p.desc[i].voff[t] = v[t];
                v[t] += (size_t)p.desc[i].H * p.desc[i].W * (t == 0 ? 3 : 6) * p.desc[i].Kp;
                voff_pair[2 * (size_t)i + t] = v2[t];
                v2[t] += (size_t)p.desc[i].H * p.desc[i].W * (t == 0 ? 3 : 6) * ((2 * p.desc[i].K + 3) / 4 * 4);
            }
        if (p.groups_cap * v[1] > p.val_cap || p.groups_cap * v[0] > p.valg_cap)     // incl. room for the paired run
            return fail(e, PNP_ERR_ARG, "CRF chunk needs %zu value floats, reserved %zu", p.groups_cap * v[1], p.val_cap);
    }
    p.B = B;
    p.total_pix = pix;
    p.lut_stride = b->lut_stride;
    p.d_rgb = b->d_rgb;
    p.d_gt = b->d_gt;
    HIPCHK(e, hipMemcpyAsync(p.d_desc, p.desc.data(), sizeof(PostDesc) * B, hipMemcpyHostToDevice, s));
    p.desc_pair = p.desc;
    p.Kpmax_pair = 0;
    {
        size_t qoff2 = 0;
        for (int i = 0; i < B; i++) {          // rows of two groups
            PostDesc& d = p.desc_pair[i];
            d.G = 2;
            d.Kp = (2 * d.K + 3) / 4 * 4;
            d.qoff = qoff2;
            qoff2 += (size_t)d.Kp * d.H * d.W;
            d.voff[0] = voff_pair[2 * (size_t)i];
            d.voff[1] = voff_pair[2 * (size_t)i + 1];
            p.Kpmax_pair = std::max(p.Kpmax_pair, d.Kp);
        }
    }
    HIPCHK(e, hipMemcpyAsync(p.d_desc_pair, p.desc_pair.data(), sizeof(PostDesc) * B, hipMemcpyHostToDevice, s));
    HIPCHK(e, hipMemcpyAsync(p.d_img_cls_off, b->img_cls_off, 4 * (B + 1), hipMemcpyHostToDevice, s));
    HIPCHK(e, hipMemcpyAsync(p.d_cls_off, b->cls_off, 4 * (ncls + 1), hipMemcpyHostToDevice, s));
    HIPCHK(e, hipMemcpyAsync(p.d_cls_div, b->cls_div, 4 * std::max(ncls, 1), hipMemcpyHostToDevice, s));
    HIPCHK(e, hipMemcpyAsync(p.d_tok_idx, b->tok_idx, 4 * std::max(ntok, 1), hipMemcpyHostToDevice, s));
    HIPCHK(e, hipMemcpyAsync(p.d_lut, b->lut, 4 * (size_t)B * b->lut_stride, hipMemcpyHostToDevice, s));
    HIPCHK(e, hipMemcpyAsync(p.d_label_off, label_off.data(), sizeof(size_t) * (B + 1), hipMemcpyHostToDevice, s));
    HIPCHK(e, hipMemcpyAsync(p.d_wts, wts.data(), 8 * wts.size(), hipMemcpyHostToDevice, s));
    HIPCHK(e, hipMemcpyAsync(p.d_wt_off, wt_off.data(), 4 * (B + 1), hipMemcpyHostToDevice, s));
    p.has_crf = false;
    // the table copies above read caller / engine host memory asynchronously: they are complete behind the lattice build's own
    // read-back (one synchronisation per batch, inside crf_build_lattice); without a lattice to build, wait here
    bool synced = false;
    if (want_crf) {
        // PnP.py:1036-1041: POS_XY_STD = 3, Bi_XY_STD = 50, Bi_RGB_STD = 5 (features are fixed per batch)
        std::vector<int> sig;
        for (int i = 0; i < B; i++) { sig.push_back(p.desc[i].H); sig.push_back(p.desc[i].W); }
        for (int t = 0; t < 2; t++) {
            // the Gaussian (xy-only) lattice depends on the image sizes alone: keep it across batches
            if (t == 0 && sig == p.gauss_sig) continue;
            const int D1 = t == 0 ? 3 : 6;
            KCHK(e, crf_build_lattice(D1 - 1, p.lat[t], p.d_desc, p.desc.data(), p.d_rgb, t == 0 ? 3.0f : 50.0f, 5.0f, B, (size_t)pix * D1,
                                      p.maxHW, p.keys_a, p.keys_b, p.vals_a, p.head, p.incl,
                                      p.n1k, p.n2k, p.sort_tmp, p.sort_tmp_bytes, p.range_err, &p.range_err_host, &p.lat_points[t], s));
            KCHK(e, crf_lattice_norm(p.lat[t], p.d_desc, B, p.maxHW, (size_t)pix * D1, p.va, p.vb, p.norm[t], s));
            synced = true;
        }
        if (p.range_err_host) {                // (read back with the lattice size inside crf_build_lattice: no extra sync)
            p.range_err_host = 0;                           // leave no stale state behind: the next prepare rebuilds both lattices
            p.gauss_sig.clear();
            (void)hipMemsetAsync(p.range_err, 0, 4, s);
            return fail(e, PNP_ERR_ARG, "lattice key out of packing range (image too large for the 64-bit key)");
        }
        p.gauss_sig = sig;
        p.has_crf = true;
    }
    if (!synced) HIPCHK(e, hipStreamSynchronize(s));
    p.prepared = true;
    return PNP_OK;
}

#define POST_READY(e)                                                                     \
    if (!(e)) return PNP_ERR_ARG;                                                         \
    if (!(e)->post.prepared) return fail(e, PNP_ERR_STATE, "call pnp_post_prepare first");

extern "C" int pnp_merge_tokens(pnp_engine* e, const float* d_gradcam, int32_t T, void* stream) {
    POST_READY(e);
    auto& p = e->post;
    if (!d_gradcam || T < 5) return fail(e, PNP_ERR_ARG, "bad gradcam / T");
    KCHK(e, merge_tokens(d_gradcam, p.d_cls_off, p.d_tok_idx, p.d_cls_div, p.d_img_cls_off, p.merged, p.B, T, e->PP, p.Cmax,
                         (hipStream_t)stream));
    return PNP_OK;
}

extern "C" int pnp_threshold_upsample(pnp_engine* e, float threshold, int32_t scale01, void* stream) {
    POST_READY(e);
    auto& p = e->post;
    hipStream_t s = (hipStream_t)stream;
    KCHK(e, threshold_maps(p.merged, p.d_img_cls_off, p.thr, threshold, p.B, e->PP, p.Cmax, s));
    KCHK(e, upsample_maps(p.thr, p.d_desc, p.maps, p.B, e->P, p.Cmax, p.maxHW, s));
    if (scale01) KCHK(e, minmax_normalize(p.maps, p.d_desc, p.stats, p.B, p.Cmax, p.maxHW, 1, s));
    KCHK(e, background_channel(p.maps, p.d_desc, p.B, p.maxHW, s));
    p.maps_in_2 = false;
    return PNP_OK;
}

extern "C" int pnp_blur_minmax(pnp_engine* e, void* stream) {
    POST_READY(e);
    auto& p = e->post;
    hipStream_t s = (hipStream_t)stream;
    KCHK(e, blur_maps(p.maps, p.maps3, p.maps2, p.d_desc, p.d_wts, p.d_wt_off, p.B, p.Kmax, p.maxH, p.maxW, p.max_radius, s));
    KCHK(e, minmax_normalize(p.maps2, p.d_desc, p.stats, p.B, p.Kmax, p.maxHW, 0, s));
    p.maps_in_2 = true;
    return PNP_OK;
}

// mean-field iterations over the unary rows already in p.unary; desc = single- or two-group descriptors
static int crf_iterate_body(pnp_engine* e, const PostDesc* desc, int kp_max, int32_t iters, float pos_w, float bi_w, hipStream_t s,
                            const CrfLabelOut& labels);

// labels: when set, the last update also writes the label maps (argmax + LUT fused into it: no separate pass over Q)
static int crf_iterate(pnp_engine* e, const PostDesc* desc, int kp_max, int32_t iters, float pos_w, float bi_w, hipStream_t s,
                       const CrfLabelOut& labels = CrfLabelOut()) {
    auto& pf = e->crf_prof;
    const bool timed = pf.on && pf.used < 4096;
    if (timed) {
        if ((int)pf.ev0.size() <= pf.used) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return fail(e, PNP_ERR_HIP, "hipEventCreate failed");
            pf.ev0.push_back(a);
            pf.ev1.push_back(b);
        }
        (void)hipEventRecord(pf.ev0[pf.used], s);
    }
    const int r = crf_iterate_body(e, desc, kp_max, iters, pos_w, bi_w, s, labels);
    if (timed) {
        (void)hipEventRecord(pf.ev1[pf.used], s);
        pf.used++;
        pf.launches++;
        // `work` = SURVEY.md 8d, nothing else: per mean-field iteration splat + slice of (3 + 6) simplex vertices per pixel and
        // channel, each a 4-byte read or write, plus Q read and written once: (2 * 9 + 2) * K * H * W * 4 bytes.  The term that grows
        // with the lattice instead of the pixels is kept BESIDE it (`lattice`, stage 2 of pnp_profile_read_stage): the axis blurs
        // read and write the value array of every lattice point (M_g + M_b points of K floats) once per PASS of two axes -- 2 passes
        // for the Gaussian lattice (d + 1 = 3 axes), 3 for the bilateral one (6 axes).  Neither figure is what the HBM counters
        // see (part of both is served by L2: the kernels are gather-bound)
        const auto& p = e->post;
        const int groups = desc == p.d_desc_pair ? 2 : 1;
        double kpix = 0, pix = 0;
        for (int i = 0; i < p.B; i++) {
            kpix += (double)p.desc[i].K * p.desc[i].H * p.desc[i].W;
            pix += (double)p.desc[i].H * p.desc[i].W;
        }
        const double kavg = pix > 0 ? kpix / pix : 0;
        pf.work += (double)iters * 20.0 * 4.0 * groups * kpix;
        pf.lattice += (double)iters * 2.0 * 4.0 * groups * kavg * (2.0 * p.lat_points[0] + 3.0 * p.lat_points[1]);
    }
    return r;
}

static int crf_iterate_body(pnp_engine* e, const PostDesc* desc, int kp_max, int32_t iters, float pos_w, float bi_w, hipStream_t s,
                            const CrfLabelOut& labels) {
    auto& p = e->post;
    const int groups = desc == p.d_desc_pair ? 2 : 1;
    for (int c0 = 0; c0 < p.B; c0 += p.chunk) {
        const int n = std::min(p.chunk, p.B - c0);
        KCHK(e, crf_update(p.lat[0], p.lat[1], desc, c0, n, p.vga, p.va, p.norm[0], p.norm[1], p.unary, p.Q, pos_w, bi_w, 0,
                           p.maxHW, kp_max, groups, s));
        for (int it = 0; it < iters; it++) {
            const float *rg = nullptr, *rb = nullptr;
            KCHK(e, crf_filter(p.lat[0], desc, c0, n, p.Q, p.vga, p.vgb, &rg, kp_max, s));
            KCHK(e, crf_filter(p.lat[1], desc, c0, n, p.Q, p.va, p.vb, &rb, kp_max, s));
            KCHK(e, crf_update(p.lat[0], p.lat[1], desc, c0, n, rg, rb, p.norm[0], p.norm[1], p.unary, p.Q, pos_w, bi_w, 1,
                               p.maxHW, kp_max, groups, s, it == iters - 1 ? labels : CrfLabelOut()));
        }
    }
    return PNP_OK;
}

extern "C" int pnp_densecrf(pnp_engine* e, int32_t iters, float pos_w, float pos_xy, float bi_w, float bi_xy, float bi_rgb,
                            void* stream) {
    POST_READY(e);
    auto& p = e->post;
    if (!p.has_crf) return fail(e, PNP_ERR_STATE, "batch was prepared without CRF lattices");
    if (pos_xy != 3.0f || bi_xy != 50.0f || bi_rgb != 5.0f)
        return fail(e, PNP_ERR_ARG, "lattices are built for sxy=3 / sxy=50, srgb=5 (PnP.py:1036-1041)");
    hipStream_t s = (hipStream_t)stream;
    const float* maps = p.maps_in_2 ? p.maps2 : p.maps;
    KCHK(e, unary_from_maps(maps, p.d_desc, p.unary, p.B, p.maxHW, p.maxKp, 0, s));
    return crf_iterate(e, p.d_desc, p.Kpmax, iters, pos_w, bi_w, s);
}

extern "C" int pnp_remap_hist(pnp_engine* e, int32_t from_crf, uint8_t* d_labels, unsigned long long* d_hist, int32_t n_class,
                              void* stream) {
    POST_READY(e);
    auto& p = e->post;
    if (!d_labels) return fail(e, PNP_ERR_ARG, "d_labels is null");
    hipStream_t s = (hipStream_t)stream;
    const float* src = from_crf ? p.Q : (p.maps_in_2 ? p.maps2 : p.maps);
    KCHK(e, argmax_remap(src, p.d_desc, p.d_lut, p.lut_stride, d_labels, p.d_label_off, from_crf ? 1 : 0, 0, p.B, p.maxHW, s));
    if (d_hist && p.d_gt) KCHK(e, confusion_hist(d_labels, p.d_gt, p.d_desc, p.d_label_off, d_hist, n_class, p.B, p.maxHW, s));
    return PNP_OK;
}

extern "C" int pnp_postprocess(pnp_engine* e, const float* d_gradcam, int32_t T, float threshold, int32_t scale01, int32_t mode,
                               uint8_t* d_labels, unsigned long long* d_hist, int32_t n_class, void* stream) {
    int r = pnp_merge_tokens(e, d_gradcam, T, stream);
    if (r) return r;
    r = pnp_threshold_upsample(e, threshold, scale01, stream);
    if (r) return r;
    if (mode & 1) {
        r = pnp_blur_minmax(e, stream);
        if (r) return r;
    }
    if (mode & 2) {
        r = pnp_densecrf(e, 10, 7.0f, 3.0f, 10.0f, 50.0f, 5.0f, stream);
        if (r) return r;
    }
    return pnp_remap_hist(e, (mode & 2) ? 1 : 0, d_labels, d_hist, n_class, stream);
}

// 1-drop and N-drop "blur+crf" post-processing of one batch in ONE mean-field run (PnP.py:348-403 and 424-481 run the
// same DenseCRF twice per image on the same RGB image): the two problems become two channel groups of every row, so the
// lattice index walks -- contributor lists, neighbour ids, simplex offsets -- are shared.  Per-channel arithmetic is
// untouched: labels and histograms equal two pnp_postprocess calls bit for bit.
extern "C" int pnp_postprocess_pair(pnp_engine* e, const float* d_gradcam_1drop, const float* d_gradcam_ndrop, int32_t T,
                                    float threshold, int32_t scale01_mask, uint8_t* d_labels_1drop, unsigned long long* d_hist_1drop,
                                    uint8_t* d_labels_ndrop, unsigned long long* d_hist_ndrop, int32_t n_class, void* stream) {
    POST_READY(e);
    auto& p = e->post;
    if (!p.has_crf) return fail(e, PNP_ERR_STATE, "batch was prepared without CRF lattices");
    if (!d_labels_1drop || !d_labels_ndrop) return fail(e, PNP_ERR_ARG, "label outputs are null");
    if (p.groups_cap < 2) {                     // arrays sized for one group only: same results, two passes
        int r = pnp_postprocess(e, d_gradcam_1drop, T, threshold, scale01_mask & 1, 3, d_labels_1drop, d_hist_1drop, n_class, stream);
        if (r) return r;
        return pnp_postprocess(e, d_gradcam_ndrop, T, threshold, (scale01_mask >> 1) & 1, 3, d_labels_ndrop, d_hist_ndrop, n_class, stream);
    }
    hipStream_t s = (hipStream_t)stream;
    for (int grp = 0; grp < 2; grp++) {
        int r = pnp_merge_tokens(e, grp == 0 ? d_gradcam_1drop : d_gradcam_ndrop, T, stream);
        if (r) return r;
        r = pnp_threshold_upsample(e, threshold, (scale01_mask >> grp) & 1, stream);   // PnP.py: Scale_0_1 in the 1-drop branch only; PnPc.py: both
        if (r) return r;
        r = pnp_blur_minmax(e, stream);
        if (r) return r;
        KCHK(e, unary_from_maps(p.maps2, p.d_desc_pair, p.unary, p.B, p.maxHW, p.maxKp, grp, s));
    }
    CrfLabelOut lout;
    lout.lab[0] = d_labels_1drop;
    lout.lab[1] = d_labels_ndrop;
    lout.lut = p.d_lut;
    lout.lut_stride = p.lut_stride;
    lout.label_off = p.d_label_off;
    int r = crf_iterate(e, p.d_desc_pair, p.Kpmax_pair, 10, 7.0f, 10.0f, s, lout);
    if (r) return r;
    for (int grp = 0; grp < 2; grp++) {
        uint8_t* lab = grp == 0 ? d_labels_1drop : d_labels_ndrop;
        unsigned long long* hist = grp == 0 ? d_hist_1drop : d_hist_ndrop;
        if (hist && p.d_gt) KCHK(e, confusion_hist(lab, p.d_gt, p.d_desc, p.d_label_off, hist, n_class, p.B, p.maxHW, s));
    }
    return PNP_OK;
}

// =========================================================================================== introspection

extern "C" int pnp_get_buffer(pnp_engine* e, const char* name, void** d_ptr, size_t* bytes) {
    if (!e || !name || !d_ptr || !bytes) return PNP_ERR_ARG;
    const std::string n(name);
    const size_t B = e->c.max_batch, L = e->c.max_text_len;
    auto& p = e->post;
    auto set = [&](void* ptr, size_t b) { *d_ptr = ptr; *bytes = b; return PNP_OK; };
    if (n == "image_embeds") return set(e->emb32, B * e->N * (size_t)e->D * 4);
    if (n == "image_embeds_t") return set(e->embT, B * e->N * (size_t)e->D * e->esz);
    if (n == "x") return set(e->x, B * e->N * (size_t)e->D * 4);
    if (n == "P") return set(e->ta[e->grad_layer >= 0 ? e->grad_layer : e->SL].Pc, B * e->nh * L * (size_t)e->Nst * 4);
    if (n == "dP") return set(e->dPc, B * e->nh * L * (size_t)e->Nst * 4);
    if (n == "h_last") return set(e->ta[e->TL - 1].h_out, B * L * (size_t)e->H * 4);
    if (n == "dropped") return set(e->dropped, B * (size_t)e->PP);
    if (n == "P_last" && e->ta[e->TL - 1].Pc) return set(e->ta[e->TL - 1].Pc, B * e->nh * L * (size_t)e->Nst * 4);
    if (n == "Kt" && e->Kt) return set(e->Kt, (size_t)(e->TL - e->SL - 1) * e->H * B * e->Npad * e->esz);
    if (n == "Vt") return set(e->Vt, (size_t)e->TL * e->H * B * e->Npad * e->esz);
    if (n == "dq_xattn") return set(e->dqc, B * L * (size_t)e->H * e->esz);
    if (n == "dctx_xattn") return set(e->dctxc, B * L * (size_t)e->H * e->esz);
    if (p.reserved) {
        const size_t mk = (size_t)p.max_total_pix * p.maxK * 4;
        const size_t mq = (size_t)p.max_total_pix * p.maxKp * 4;
        if (n == "merged") return set(p.merged, (size_t)p.maxB * p.maxK * e->PP * 4);
        if (n == "maps") return set(p.maps_in_2 ? p.maps2 : p.maps, mk);
        if (n == "maps_pre_blur") return set(p.maps, mk);
        if (n == "unary") return set(p.unary, mq);
        if (n == "crf_q") return set(p.Q, mq);
        if (n == "crf_idbase_gauss") return set(p.lat[0].idbase, ((size_t)p.maxB + 1) * 4);
        if (n == "crf_idbase_bilateral") return set(p.lat[1].idbase, ((size_t)p.maxB + 1) * 4);
        if (n == "crf_norm_gauss") return set(p.norm[0], (size_t)p.max_total_pix * 4);
        if (n == "crf_norm_bilateral") return set(p.norm[1], (size_t)p.max_total_pix * 4);
        if (n == "crf_offset_gauss") return set(p.lat[0].offset, (size_t)p.max_total_pix * 3 * 4);          // per pixel: 3 lattice ids
        if (n == "crf_offset_bilateral") return set(p.lat[1].offset, (size_t)p.max_total_pix * 6 * 4);  // per pixel: 6 lattice ids
        if (n == "crf_nbr8_gauss") return set(p.lat[0].nbr8, (size_t)(p.lat[0].D1 / 2) * p.lat[0].cap * sizeof(CrfNbr8));
        if (n == "crf_nbr8_bilateral") return set(p.lat[1].nbr8, (size_t)(p.lat[1].D1 / 2) * p.lat[1].cap * sizeof(CrfNbr8));
    }
    return fail(e, PNP_ERR_ARG, "unknown buffer %s", name);
}

extern "C" size_t pnp_allocated_bytes(const pnp_engine* e) { return e ? e->alloc_bytes : 0; }

extern "C" int pnp_profile_enable(pnp_engine* e, int32_t on) {
    if (!e) return PNP_ERR_ARG;
    if (!e->gemm_prof) {
        if (!on) {
            e->crf_prof.on = false;
            return PNP_OK;
        }
        e->gemm_prof = new GemmProfile();
    }
    GemmProfile& pf = *e->gemm_prof;
    pf.on = on != 0;
    pf.period = on > 1 ? on : 1;
    pf.seq = 0;
    pf.used = 0;
    pf.launches = 0;
    pf.flops = 0;
    e->crf_prof.on = on != 0;
    e->crf_prof.used = 0;
    e->crf_prof.launches = 0;
    e->crf_prof.work = 0;
    e->crf_prof.lattice = 0;
    return PNP_OK;
}

extern "C" int pnp_profile_read_stage(pnp_engine* e, int32_t stage, int64_t* launches, double* work, double* ms) {
    if (!e || !launches || !work || !ms) return PNP_ERR_ARG;
    if (stage == 0) return pnp_profile_read(e, launches, work, ms);
    if (stage != 1 && stage != 2) return fail(e, PNP_ERR_ARG, "unknown profile stage %d", stage);
    auto& pf = e->crf_prof;
    double total = 0;
    for (int i = 0; i < pf.used; i++) {
        HIPCHK(e, hipEventSynchronize(pf.ev1[i]));
        float t = 0;
        HIPCHK(e, hipEventElapsedTime(&t, pf.ev0[i], pf.ev1[i]));
        total += t;
    }
    *launches = pf.launches;
    *work = stage == 1 ? pf.work : pf.lattice;
    *ms = total;
    return PNP_OK;
}

extern "C" int pnp_profile_read(pnp_engine* e, int64_t* launches, double* flops, double* ms) {
    if (!e || !launches || !flops || !ms) return PNP_ERR_ARG;
    if (!e->gemm_prof) {
        *launches = 0; *flops = 0; *ms = 0;
        return PNP_OK;
    }
    GemmProfile& pf = *e->gemm_prof;
    double total = 0;
    for (int i = 0; i < pf.used; i++) {
        HIPCHK(e, hipEventSynchronize(pf.ev1[i]));
        float t = 0;
        HIPCHK(e, hipEventElapsedTime(&t, pf.ev0[i], pf.ev1[i]));
        total += t;
    }
    *launches = pf.launches;
    *flops = pf.flops;
    *ms = total;
    return PNP_OK;
}

extern "C" int pnp_op_gemm(int32_t bf, const void* d_A, int32_t lda, const void* d_B, int32_t ldb, int32_t M, int32_t N,
                           int32_t K, const float* d_bias, const float* d_resid, int32_t ldr, float* d_out_f32, int32_t ldo,
                           int32_t gelu, void* stream) {
    GemmArgs g = G_(d_A, lda, d_B, ldb, M, N, K);
    g.bias = d_bias; g.resid = d_resid; g.ldr = ldr; g.out_f32 = d_out_f32; g.ldo = ldo;
    g.mode = gelu ? GEMM_EPI_GELU : GEMM_EPI_LINEAR;
    return gemm_nt(bf, g, (hipStream_t)stream);
}

extern "C" int pnp_op_gemm_ex(int32_t bf, const void* d_A, int32_t lda, const void* d_B, int32_t ldb, int32_t M, int32_t N,
                              int32_t K, const float* d_bias, const float* d_resid, int32_t ldr, float* d_out_f32, int32_t ldo,
                              void* d_out_t, int32_t ldo_t, int32_t mode, void* stream) {
    GemmArgs g = G_(d_A, lda, d_B, ldb, M, N, K);
    g.bias = d_bias; g.resid = d_resid; g.ldr = ldr; g.out_f32 = d_out_f32; g.ldo = ldo; g.out_t = d_out_t; g.ldo_t = ldo_t;
    g.mode = mode ? GEMM_EPI_GELU : GEMM_EPI_LINEAR;
    return gemm_nt(bf, g, (hipStream_t)stream);
}

extern "C" int pnp_op_split(const float* d_in, void* d_hi, void* d_lo, int64_t n, void* stream) {
    if (!d_in || !d_hi || !d_lo || n <= 0) return PNP_ERR_ARG;
    return split_f32(d_in, d_hi, d_lo, (size_t)n, (hipStream_t)stream);
}

extern "C" int pnp_op_gemm_x3(const void* d_A_hi, const void* d_A_lo, int32_t lda, const void* d_B_hi, const void* d_B_lo,
                              int32_t ldb, int32_t M, int32_t N, int32_t K, const float* d_bias, int32_t bias_on_rows,
                              const float* d_resid, int32_t ldr, float* d_out_f32, int32_t ldo, void* d_out_hi, void* d_out_lo,
                              int32_t ldo_t, int32_t gelu, int32_t col_div, int32_t col_pad, void* stream) {
    if (!d_A_hi || !d_A_lo || !d_B_hi || !d_B_lo) return PNP_ERR_ARG;
    GemmArgs g = G_(d_A_hi, lda, d_B_hi, ldb, M, N, K);
    g.A_lo = d_A_lo; g.B_lo = d_B_lo;
    g.bias = d_bias; g.bias_on_rows = bias_on_rows; g.resid = d_resid; g.ldr = ldr; g.out_f32 = d_out_f32; g.ldo = ldo;
    g.out_t = d_out_hi; g.out_lo = d_out_lo; g.ldo_t = ldo_t; g.col_div = col_div; g.col_pad = col_pad;
    g.mode = gelu ? GEMM_EPI_GELU : GEMM_EPI_LINEAR;
    g.sk = streamk_mode() ? streamk_ws_default() : nullptr;     // op level: the per-device workspace (calls on one stream at a time)
    return gemm_nt(1, g, (hipStream_t)stream);
}

extern "C" int pnp_set_tuning(const char* key, int32_t value) {
    if (!key) return PNP_ERR_ARG;
    if (!strcmp(key, "streamk")) {
        if (value < 0 || value > 2) return PNP_ERR_ARG;
        set_streamk_mode(value);
        return PNP_OK;
    }
    return PNP_ERR_ARG;
}

extern "C" int pnp_streamk_status(pnp_engine* e, int64_t* launches, uint32_t* gave_up) {
    if (!launches || !gave_up) return PNP_ERR_ARG;
    StreamKWs* ws = e ? &e->sk_ws : streamk_ws_default();
    *launches = ws ? ws->launches : 0;
    *gave_up = 0;
    if (!ws || !ws->part) return PNP_OK;
    if (e) HIPCHK(e, hipSetDevice(e->c.device));
    if (hipDeviceSynchronize() != hipSuccess) return PNP_ERR_HIP;
    return streamk_ws_timeouts(ws, gave_up);
}

extern "C" int pnp_op_gemm_x3a(const float* d_A, int32_t lda, const void* d_B_hi, const void* d_B_lo, int32_t ldb, int32_t M,
                               int32_t N, int32_t K, const float* d_bias, const float* d_resid, int32_t ldr, float* d_out_f32,
                               int32_t ldo, int32_t mode, float* d_aux, int32_t ld_aux, void* stream) {
    if (!d_A || !d_B_hi || !d_B_lo || !d_out_f32 || mode < 0 || mode > 2) return PNP_ERR_ARG;
    GemmArgs g = G_(d_A, lda, d_B_hi, ldb, M, N, K);
    g.B_lo = d_B_lo; g.a_f32 = 1;
    g.bias = d_bias; g.resid = d_resid; g.ldr = ldr; g.out_f32 = d_out_f32; g.ldo = ldo;
    g.mode = mode; g.aux = d_aux; g.ld_aux = ld_aux;
    return gemm_nt(0, g, (hipStream_t)stream);
}

extern "C" int pnp_op_gemm_tokcols(int32_t bf, const void* d_A, int32_t lda, const void* d_B, int32_t ldb, int32_t M, int32_t N,
                                   int32_t K, const float* d_bias_rows, void* d_out_t, int32_t ldo_t, int32_t col_div,
                                   int32_t col_pad, void* stream) {
    if (!d_out_t || col_div < 0 || (col_div > 0 && col_pad < col_div)) return PNP_ERR_ARG;
    GemmArgs g = G_(d_A, lda, d_B, ldb, M, N, K);
    g.bias = d_bias_rows; g.bias_on_rows = 1; g.out_t = d_out_t; g.ldo_t = ldo_t; g.col_div = col_div; g.col_pad = col_pad;
    return gemm_nt(bf, g, (hipStream_t)stream);
}

extern "C" int pnp_op_vit_attention(int32_t bf, const void* d_qk, int32_t ld_qk, int32_t D, const void* d_vt, int32_t ld_vt,
                                    int32_t n_pad, void* d_ctx, int32_t B, int32_t heads, int32_t N, float scale, void* stream) {
    if (!d_qk || !d_vt || !d_ctx || B <= 0 || heads <= 0 || N <= 0) return PNP_ERR_ARG;
    return vit_attention(bf, d_qk, ld_qk, D, d_vt, ld_vt, n_pad, d_ctx, B, heads, N, scale, (hipStream_t)stream);
}

extern "C" int pnp_op_vit_attention_x3(const void* d_qkv_hi, const void* d_qkv_lo, int32_t ld_qkv, int32_t D, void* d_ctx_hi,
                                       void* d_ctx_lo, int32_t B, int32_t heads, int32_t N, float scale, void* stream) {
    if (B <= 0 || heads <= 0 || N <= 0 || ld_qkv < 3 * D) return PNP_ERR_ARG;
    return vit_attention_x3(d_qkv_hi, d_qkv_lo, ld_qkv, D, d_ctx_hi, d_ctx_lo, B, heads, N, scale, (hipStream_t)stream);
}

extern "C" int pnp_op_layernorm(const float* d_x, const float* d_w, const float* d_b, float eps, int32_t rows, int32_t D,
                                float* d_y, void* stream) {
    return layernorm(0, d_x, d_w, d_b, eps, rows, D, d_y, nullptr, nullptr, nullptr, (hipStream_t)stream);
}

extern "C" int pnp_op_xattn(int32_t bf16, int32_t mode, const void* d_nat, int32_t ld_nat, const void* d_tr, int32_t ld_tr,
                            int32_t n_pad, const void* d_x, int32_t ldx, void* d_out, int32_t ldo, float* d_probs,
                            int32_t n_stride, int32_t B, int32_t L, int32_t N, int32_t heads, void* stream) {
    if (!d_nat || !d_x || !d_probs || mode < 0 || mode > 2 || (mode != 2 && (!d_tr || !d_out))) return PNP_ERR_ARG;
    if (B <= 0 || L <= 0 || N <= 0 || heads <= 0) return PNP_ERR_ARG;
    return xattn(bf16, mode, d_nat, ld_nat, d_tr, ld_tr, n_pad, d_x, ldx, d_out, ldo, d_probs, n_stride, B, L, N, heads,
                 (hipStream_t)stream);
}

extern "C" int pnp_dbg_gemm_stamps(uint64_t* host_out, int32_t max_blocks) {
    return gemm_read_stamps((unsigned long long*)host_out, max_blocks);
}

extern "C" int pnp_preprocess_images(const uint8_t* d_rgb, const pnp_pre_image* d_desc, int32_t B, int32_t S, int32_t max_H,
                                     const int32_t* d_coef, uint8_t* d_tmp, const float* mean3, const float* std3,
                                     float* d_out, void* stream) {
    if (!d_rgb || !d_desc || !d_coef || !d_tmp || !mean3 || !std3 || !d_out) return PNP_ERR_ARG;
    static_assert(sizeof(pnp_pre_image) == 40, "pnp_pre_image layout is part of the ABI");
    return preprocess_images(d_rgb, d_desc, B, S, max_H, d_coef, d_tmp, mean3, std3, d_out, (hipStream_t)stream);
}

extern "C" int pnp_jpeg_decode(const uint8_t* d_data, const pnp_jpeg_image* d_images, const pnp_jpeg_tables* d_tables,
                               const pnp_jpeg_segment* d_segments, int32_t n_images, int32_t n_segments, uint8_t* d_clean,
                               int32_t* d_seg_bits, int16_t* d_coef, int64_t coef_elems, uint8_t* d_planes, uint8_t* d_rgb,
                               int32_t max_blocks_per_image, int32_t max_pixels_per_image, int32_t* d_err, void* stream) {
    if (!d_data || !d_images || !d_tables || !d_segments || !d_clean || !d_seg_bits || !d_coef || !d_planes || !d_rgb || !d_err ||
        coef_elems <= 0)
        return PNP_ERR_ARG;
    static_assert(sizeof(pnp_jpeg_image) == sizeof(JpegImage) && sizeof(pnp_jpeg_tables) == sizeof(JpegTables) &&
                      sizeof(pnp_jpeg_segment) == sizeof(JpegSegment), "JPEG descriptor layouts are part of the ABI");
    return jpeg_decode(d_data, reinterpret_cast<const JpegImage*>(d_images), reinterpret_cast<const JpegTables*>(d_tables),
                       reinterpret_cast<const JpegSegment*>(d_segments), n_images, n_segments, d_clean, d_seg_bits, d_coef,
                       (size_t)coef_elems, d_planes, d_rgb, max_blocks_per_image, max_pixels_per_image, d_err, (hipStream_t)stream);
}

// operator forms of the lattice build's sort / scan (tests); the scratch is allocated per call
extern "C" int pnp_op_sort_pairs(uint64_t* d_keys_in, uint64_t* d_keys_out, uint32_t* d_vals_in, uint32_t* d_vals_out, int64_t n,
                                 int32_t begin_bit, int32_t end_bit, const size_t* h_seg_off, int32_t n_seg, void* stream) {
    if (n < 0 || (n && (!d_keys_in || !d_keys_out || !d_vals_in || !d_vals_out))) return PNP_ERR_ARG;
    if (!n) return PNP_OK;
    const size_t tb = sort_temp_bytes((size_t)n);
    void* temp = nullptr;
    if (hipMalloc(&temp, tb) != hipSuccess) return PNP_ERR_NOMEM;
    int r = radix_sort_pairs(d_keys_in, d_keys_out, d_vals_in, d_vals_out, (size_t)n, begin_bit, end_bit, h_seg_off, n_seg, temp, tb, (hipStream_t)stream);
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess && r == PNP_OK) r = PNP_ERR_HIP;
    (void)hipFree(temp);
    return r;
}
extern "C" int pnp_op_scan_i32(const int32_t* d_in, int32_t* d_out, int64_t n, int32_t inclusive, void* stream) {
    if (n < 0 || (n && (!d_in || !d_out))) return PNP_ERR_ARG;
    if (!n) return PNP_OK;
    const size_t tb = sort_temp_bytes((size_t)n);
    void* temp = nullptr;
    if (hipMalloc(&temp, tb) != hipSuccess) return PNP_ERR_NOMEM;
    int r = device_scan_i32(d_in, d_out, (size_t)n, inclusive != 0, temp, tb, (hipStream_t)stream);
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess && r == PNP_OK) r = PNP_ERR_HIP;
    (void)hipFree(temp);
    return r;
}

extern "C" int pnp_op_cast(int32_t to_bf16, const float* d_in, void* d_out, int64_t n, void* stream) {
    return cast_f32(to_bf16, d_in, d_out, (size_t)n, (hipStream_t)stream);
}
