// Host launchers of the hand-written gfx950 kernels (one .hip per group).  All return PNP_OK or a
// negative errno-style code; none allocates or synchronises (graph-capture safe).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "gemm.h"

namespace pnp {

// Per-image descriptor of the post-process batch (images may have different original sizes and
// class counts).  `off` = offset (in floats) of this image's K*H*W block inside the flat map /
// unary / Q / tmp buffers; `pix0` = global index of its first pixel; `voff[t]` = offset of its
// lattice value block (t = 0 Gaussian, 1 bilateral).
struct PostDesc {
    int H, W, K, C, has_bg, pix0;
    int Kp;            // row stride of the CRF arrays in floats: G groups of Kg back to back, zero-padded to a multiple of 4
    int G, Kg;         // channel groups per row (1; 2 = the 1-drop and N-drop problems of one batch side by side, which
                       // share every lattice index walk) and floats per group (= K)
    size_t off;        // K*H*W blocks of the (K,H,W) map buffers
    size_t qoff;       // Kp*H*W blocks of the pixel-major CRF arrays (unary, Q)
    size_t voff[2];
};

// contributor record of a lattice point, stored in list order: global pixel index, barycentric weight, the pixel's
// normaliser 1/sqrt(lattice(1) + 1e-20) (filled in by crf_lattice_norm); one 16-byte load per contributor
struct alignas(16) CrfEntry {
    uint32_t pixel;
    float w;
    float nr;
    uint32_t pad;
};

// neighbour record of the two-axis lattice blur (crf_blur4x2_kernel): image-local lattice ids, -1 = absent
struct alignas(16) CrfNbr8 {
    int ia, id;        // neighbours along the pair's first axis
    int pa, pd;        // neighbours along its second axis ...
    int aa, ad;        // ... and their neighbours along the first axis
    int da, dd;
};

// One permutohedral lattice type for a whole image batch (device pointers).
struct CrfLattice {
    int D1;            // d + 1
    int which;         // 0 Gaussian, 1 bilateral
    size_t cap;        // capacity (entries) = stride of the neighbour tables
    float* bary;       // [entries] barycentric weight per (pixel, vertex)
    uint32_t* vals;    // [entries] sorted (pixel, vertex) indices = contributor lists
    CrfEntry* ent;     // [entries] the same lists as (pixel, weight, norm) records (what the splat streams)
    int* offset;       // [entries] lattice id per (pixel, vertex)
    int* seg_start;    // [M+1] first sorted entry of each lattice point, key order (build scratch)
    int* seg_lo;       // [M] contributor range of each lattice point (final, spatial numbering)
    int* seg_hi;
    uint64_t* ukeys;   // [M] sorted unique packed keys
    int* idbase;       // [B+1] first lattice id of each image
    int* n1;           // [(d+1) * cap]
    int* n2;
    CrfNbr8* nbr8;     // [(d+1)/2 * cap] per axis pair
};

// vit_kernels.hip
int patchify(int bf, const float* img, const uint8_t* dropped, void* out, int B, int S, int P, hipStream_t s);
int cls_rows(const float* cls, const float* pos, float* x, int B, int N, int D, hipStream_t s);
int layernorm(int bf, const float* x, const float* w, const float* b, float eps, int rows, int D, float* y, void* yt,
              float* xhat, float* rstd, hipStream_t s, void* yt_lo = nullptr);
int vit_attention(int bf, const void* qk, int ld_qk, int D, const void* vt, int ld_vt, int Npad, void* ctx, int B,
                  int H, int N, float scale, hipStream_t s, void* ctx_lo = nullptr);

int vit_attention_x3(const void* qkv_hi, const void* qkv_lo, int ld_qk, int D, void* ctx_hi, void* ctx_lo, int B, int H, int N,
                     float scale, hipStream_t s);

// text_kernels.hip
int text_embed(const int64_t* ids, int ld_ids, const float* word, const float* pos, float* out, int B, int L, int H,
               int enc_id, int vocab, hipStream_t s);
// scratch: [B, heads, L, L] floats, used when L > 192 and probs is null (the long-caption form passes a row's probabilities
// through global memory)
int text_self_attn(int bf, const void* qkv, const int64_t* mask, int ld_mask, void* ctx, float* probs, float* scratch, int B, int L,
                   int H, hipStream_t s);
int text_self_attn_bwd(int bf, const void* qkv, const float* dctx, const float* probs, float* ds_scratch, void* dqkv,
                       int B, int L, int H, hipStream_t s);
int xattn(int bf, int mode, const void* a1, int ld1, const void* a2t, int ld2, int Npad, const void* x, int ldx,
          void* out, int ldo, float* pbuf, int Nst, int B, int L, int N, int nheads, hipStream_t s);
int layernorm_bwd(int bf, const float* dy, const float* w, const float* xhat, const float* rstd, int rows, int D,
                  float* dx, void* dxt, hipStream_t s);
int itm_head(const float* hlast, const float* w, const float* bias, float* logits, int B, int L, int H, hipStream_t s);
int itm_grad_seed(const float* w, float* dh, int B, int L, int H, hipStream_t s);
int cast_f32(int bf, const float* in, void* out, size_t n, hipStream_t s);
int split_f32(const float* in, void* hi, void* lo, size_t n, hipStream_t s);     // x ~ hi + lo, both bf16 (round to nearest even)

// pipeline_kernels.hip
int gradcam_gather(const float* P, const float* dP, const int64_t* mask, int ld_mask, float* out, int B, int nheads,
                   int head, int L, int Nst, int PP, hipStream_t s);
int drop_step(const float* G, float* g0, float* agg, uint8_t* dropped, int32_t* picks, int iter, int B, int T, int PP,
              int npick, int max_picks, hipStream_t s);
int merge_tokens(const float* src, const int32_t* cls_off, const int32_t* tok_idx, const int32_t* cls_div,
                 const int32_t* img_cls_off, float* out, int B, int T, int PP, int Cmax, hipStream_t s);
int threshold_maps(const float* merged, const int32_t* img_cls_off, float* out, float threshold, int B, int PP, int Cmax,
                   hipStream_t s);
int upsample_maps(const float* src, const PostDesc* desc, float* maps, int B, int P, int Cmax, int maxHW, hipStream_t s);
int minmax_normalize(float* maps, const PostDesc* desc, float* stats, int B, int Kmax, int maxHW, int class_channels_only,
                     hipStream_t s);
int background_channel(float* maps, const PostDesc* desc, int B, int maxHW, hipStream_t s);
int blur_maps(const float* in, float* tmp, float* out, const PostDesc* desc, const double* wts, const int32_t* wt_off,
              int B, int Kmax, int maxH, int maxW, int max_radius, hipStream_t s);
int preprocess_images(const uint8_t* rgb, const void* desc, int B, int S, int max_H, const int32_t* coef, uint8_t* tmp,
                      const float* mean3, const float* std3, float* out, hipStream_t s);
int unary_from_maps(const float* maps, const PostDesc* desc, float* unary, int B, int maxHW, int max_kp, int group, hipStream_t s);
int argmax_remap(const float* q, const PostDesc* desc, const int32_t* lut, int lut_stride, uint8_t* labels,
                 const size_t* label_off, int pixel_major, int group, int B, int maxHW, hipStream_t s);
int confusion_hist(const uint8_t* labels, const float* gt, const PostDesc* desc, const size_t* label_off,
                   unsigned long long* hist, int n_class, int B, int maxHW, hipStream_t s);

// jpeg.hip -- descriptors shared with the host (pnp_ovss/jpeg.py mirrors them with ctypes; see include/pnp_hip.h)
struct JpegImage {
    int64_t data_off;       // offset of the entropy-coded segment in d_data (16-byte aligned)
    int64_t coef_off[3];    // int16 elements: component coefficient blocks [blocks_y][blocks_x][64], natural order
    int64_t plane_off[3];   // bytes: component sample planes, row stride blocks_x * 8
    int64_t rgb_off;        // bytes: output RGB, H * W * 3
    int32_t data_len, H, W, ncomp, hmax, vmax, mcux, mcuy;
    int32_t h[3], v[3], tq[3], td[3], ta[3], bx[3], by[3];
    int32_t tab, pad[2];
};
struct JpegTables {
    uint8_t counts[4][16];  // [DC0, DC1, AC0, AC1]: the file's DHT segments as they are (codes per length 1..16, symbols)
    uint8_t vals[4][256];
    int32_t quant[4][64];   // natural (row-major) order
};
struct JpegSegment {
    int64_t byte_off;       // first entropy-coded byte of the restart interval, relative to the image's data_off
    int64_t clean_off;      // where its un-stuffed bytes go in d_clean (16-byte aligned)
    int32_t image, mcu0, nmcu;
    int32_t raw_len;        // entropy-coded bytes of the interval (markers excluded)
    int32_t clean_cap;      // bytes reserved at clean_off: >= raw_len + 32
    int32_t sub_bits;       // sub-sequence length of the parallel decode: a multiple of 32, >= 8 * raw_len / 1024
    int32_t pad[2];
};
int jpeg_decode(const uint8_t* d_data, const JpegImage* d_imgs, const JpegTables* d_tabs, const JpegSegment* d_segs, int n_images,
                int n_segments, uint8_t* d_clean, int* d_seg_bits, int16_t* d_coef, size_t coef_elems, uint8_t* d_planes, uint8_t* d_rgb,
                int max_blocks, int max_pixels, int* d_err, hipStream_t s);

// crf.hip
// sort.hip: stable LSD radix sort of (u64 key, u32 value) pairs and int32 prefix sum (the lattice build's device-wide primitives)
size_t sort_temp_bytes(size_t n);
int radix_sort_pairs(uint64_t* keys_in, uint64_t* keys_out, uint32_t* vals_in, uint32_t* vals_out, size_t n, int begin_bit, int end_bit,
                     const size_t* seg_off, int nseg, void* temp, size_t temp_bytes, hipStream_t s);
int device_scan_i32(const int* in, int* out, size_t n, bool inclusive, void* temp, size_t temp_bytes, hipStream_t s);
size_t crf_sort_temp_bytes(size_t max_entries, int max_images);
int crf_build_lattice(int D, const CrfLattice& L, const PostDesc* d_imgs, const PostDesc* h_imgs, const uint8_t* d_rgb, float sxy, float srgb,
                      int B, size_t ent_total, int max_pixels,
                      uint64_t* keys_a, uint64_t* keys_b, uint32_t* vals_a, int* head, int* incl, int* n1k, int* n2k,
                      void* temp, size_t temp_bytes, int* d_range_err, int* h_range_err, int* h_points, hipStream_t s);
int crf_lattice_norm(const CrfLattice& L, const PostDesc* d_imgs, int B, int max_pixels, size_t ent_total, float* va, float* vb,
                     float* norm_out, hipStream_t s);
int crf_filter(const CrfLattice& L, const PostDesc* d_imgs, int img0, int nimg, const float* Q,
               float* va, float* vb, const float** result, int max_kp, hipStream_t s);
// label output of the LAST mean-field update (argmax over the marginals it has just normalised, np.argmax semantics, remapped
// through the per-image LUT): lab[g] = label map of channel group g, null = no labels from this launch
struct CrfLabelOut {
    uint8_t* lab[2] = {nullptr, nullptr};
    const int32_t* lut = nullptr;
    int lut_stride = 0;
    const size_t* label_off = nullptr;
};
int crf_update(const CrfLattice& Lg, const CrfLattice& Lb, const PostDesc* d_imgs, int img0, int nimg, const float* vg,
               const float* vb, const float* norm_g, const float* norm_b, const float* unary, float* Q, float w_g,
               float w_b, int pairwise, int max_pixels, int max_kp, int groups, hipStream_t s, const CrfLabelOut& labels = CrfLabelOut());

}  // namespace pnp
