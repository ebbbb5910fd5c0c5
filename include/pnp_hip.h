/* pnp_hip.h -- C ABI of libpnp_hip.so, the MI355X (gfx950) engine behind the PnP-OVSS hot path.
 *
 * The reference (letitiabanana/PnP-OVSS) is pure Python and has NO native boundary of its own; the
 * closest analogue is pydensecrf's DenseCRF2D object (PnP_OVSS_0514_updated_segmentation.py:
 * 1066-1071).  Each entry point below therefore cites the reference *Python* interface it replaces
 * (PnP.py = PnP_OVSS_0514_updated_segmentation.py, B/ = "Files to replace for BLIP/").
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types; every call returns 0 or a negative
 *     errno-style code, message via pnp_last_error(); no exception crosses the boundary.
 *   - "d_" pointers are DEVICE pointers owned by the caller (e.g. torch tensor .data_ptr()),
 *     contiguous row-major; `stream` is a hipStream_t (NULL = default stream).
 *   - All device memory the engine needs is allocated in pnp_create / pnp_post_reserve; the hot
 *     path calls allocate nothing and never synchronise the device (hipGraph-capturable) except
 *     where stated.
 *   - One process per device (PnP.py:1439).  An engine is not thread-safe: calls on one engine must not overlap.  Distinct
 *     engines of a process are independent (own workspace, caller's stream) and may be driven concurrently from different
 *     host threads: that is how a host keeps several batches in flight on one GPU (bench.py --pipelines, the CLI's
 *     --pipelines; tests/test_hip_parity.py::test_engines_in_flight_match_one_at_a_time).
 */
#ifndef PNP_HIP_H
#define PNP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pnp_engine pnp_engine;

/* Geometry of `load_model_and_preprocess("blip_image_text_matching", "large")`
 * (B/blip_itm_large.yaml, B/vit.py:511-523, LAVIS med_large_config.json). */
typedef struct pnp_config {
    int32_t img_size, patch, vit_dim, vit_depth, vit_heads, vit_mlp_ratio;
    float vit_ln_eps;
    int32_t txt_hidden, txt_layers, txt_heads, txt_inter;
    float txt_ln_eps;
    int32_t vocab, max_pos, enc_token_id;
    int32_t max_batch;      /* images per step (--batch_size, PnP.py:58) */
    int32_t max_text_len;   /* longest tokenised caption the engine must hold (<= 512 = BERT's position table; the reference
                             * tokenises to max_length 500, PnP.py:271,318) */
    int32_t stash_layer;    /* args.max_att_block_num - 1 (PnP.py:619): text layer whose P and dL/dP are kept */
    int32_t compute_bf16;   /* arithmetic of the dense contractions:
                             *   0  exact fp32 MFMA everywhere (the reference's arithmetic; parity mode)
                             *   1  bf16 storage + bf16 MFMA with fp32 accumulation (throughput mode; ~1 % error on image_embeds)
                             *   2  split-bf16 ("bf16x3", the benchmarked mode): EVERY dense contraction of the model -- the ViT Linears and
                             *      the ViT self-attention, the cross K/V projections, the text-side Linears and their analytic
                             *      backward -- on (hi, lo) bf16 operand pairs, three bf16 MFMA passes per product with fp32
                             *      accumulation: fp32-class results (maps within 1e-4 of mode 0, same patch picks) at ~5x the
                             *      fp32 MFMA rate.  Softmax, LayerNorm, the text self- / cross-attention kernels (fp32 operands)
                             *      and all post-processing are the mode-0 code */
    int32_t device;         /* HIP device ordinal */
} pnp_config;

/* ---- lifetime ---------------------------------------------------------------------------- */
int pnp_create(const pnp_config* cfg, pnp_engine** out);
/* A second engine on the donor's WEIGHTS (same device, compute mode and model geometry; cfg->stash_layer >= the donor's;
 * max_batch / max_text_len free): allocates activations and workspace only, never calls pnp_load_weight /
 * pnp_finalize_weights, and keeps the weights alive past the donor's pnp_destroy (shared ownership).  What `--pipelines P`
 * / bench.py's batches in flight run on: P engines, one weight copy (the reference holds one replica per process,
 * PnP.py:1212-1218; several batches in flight per GPU are this build's addition). */
int pnp_create_shared(const pnp_config* cfg, pnp_engine* donor, pnp_engine** out);
void pnp_destroy(pnp_engine* e);
const char* pnp_last_error(const pnp_engine* e);          /* valid until the next call on e */
size_t pnp_workspace_bytes(const pnp_config* cfg);         /* device bytes pnp_create will allocate */
size_t pnp_allocated_bytes(const pnp_engine* e);           /* device bytes the engine holds now (create + post_reserve + weights) */

/* Weights in: replaces BaseModel.load_checkpoint + load_state_dict (B/base_model.py:86-125).
 * `name` is the reference state-dict key ("visual_encoder.blocks.3.attn.qkv.weight", ...), data is
 * fp32 (host pointer, or device pointer when on_device != 0, e.g. after an RCCL broadcast).
 * Unknown names are ignored (strict=False), shape mismatches are errors. */
int pnp_load_weight(pnp_engine* e, const char* name, const float* data, const int64_t* shape, int32_t ndim,
                    int32_t on_device);
/* Builds fused / transposed device copies; fails with the first missing tensor name. Synchronises. */
int pnp_finalize_weights(pnp_engine* e);

/* ---- model: replaces BlipITM.forward(match_head="itm") + loss.backward() + hooks -------- */
/* VisionTransformer.forward (B/vit.py:274-290).  d_images: (B,3,S,S) fp32 normalised.
 * d_dropped: optional (B, P*P) uint8, 1 = patch zeroed by the salience drop (PnP.py:597-603). */
int pnp_vit_forward(pnp_engine* e, const float* d_images, const uint8_t* d_dropped, int32_t B, void* stream);
/* key / value Linear(1024 -> 768) of every text layer's cross-attention applied to image_embeds (B/med.py:208-211,
 * encoder_hidden_states branch), all layers at once.  pnp_vit_forward ends with this call; it is exported on its own
 * so the projections can be re-run on the engine's current image_embeds (precision probes, tests). */
int pnp_cross_kv(pnp_engine* e, int32_t B, void* stream);
/* BertModel.forward(mode="multimodal") + itm_head (B/med.py:854-1024, B/blip_image_text_matching.py:
 * 238-249); stashes the cross-attention probabilities of layers >= stash_layer (B/med.py:280-283).
 * d_ids / d_mask: (B, ld) int64, the first L columns are used (padding="longest"). */
int pnp_text_forward_xattn(pnp_engine* e, const int64_t* d_ids, const int64_t* d_mask, int32_t ld, int32_t B,
                           int32_t L, float* d_logits, void* stream);
/* loss = logits[:,1].sum(); loss.backward() restricted to what reaches attention_probs of
 * stash_layer (B/blip_image_text_matching.py:399-404, hook B/med.py:164-168). */
int pnp_xattn_grad(pnp_engine* e, int32_t B, int32_t L, void* stream);
/* The same backward stopped at any kept layer (stash_layer <= layer < txt_layers): after it "P" / "dP" and
 * pnp_gradcam_gather refer to that layer.  With stash_layer = 0 every [layer][head] entry of compute_gradcam_ensemble's
 * return value (B/blip_image_text_matching.py:411-435; the drivers' --ensemble_blocks / layer-head sweep reads them) is
 * available from one forward: one pnp_xattn_grad_layer + 12 gathers per layer. */
int pnp_xattn_grad_layer(pnp_engine* e, int32_t B, int32_t L, int32_t layer, void* stream);
/* cams * relu(grads) * mask for one head, [ENC] row and image-CLS column dropped
 * (B/blip_image_text_matching.py:427-433).  d_out: (B, L-1, P, P) fp32. */
int pnp_gradcam_gather(pnp_engine* e, const int64_t* d_mask, int32_t ld, int32_t B, int32_t L, int32_t head,
                       float* d_out, void* stream);
/* compute_gradcam_ensemble for [stash_layer][head] (B/blip_image_text_matching.py:386-457). */
int pnp_compute_gradcam(pnp_engine* e, const float* d_images, const uint8_t* d_dropped, const int64_t* d_ids,
                        const int64_t* d_mask, int32_t ld, int32_t B, int32_t L, int32_t head, float* d_out,
                        float* d_logits, void* stream);

/* ... for [layer][head], stash_layer <= layer (args.max_att_block_num - 1 chosen per call: layer / head sweeps). */
int pnp_compute_gradcam_layer(pnp_engine* e, const float* d_images, const uint8_t* d_dropped, const int64_t* d_ids,
                              const int64_t* d_mask, int32_t ld, int32_t B, int32_t L, int32_t layer, int32_t head,
                              float* d_out, float* d_logits, void* stream);

/* ---- salience-drop loop: replaces Inference_BLIP_filteredcaption (PnP.py:564-722) -------- */
/* One bookkeeping step (PnP.py:619-647 + the running sum of :716-721) on a gathered map. */
int pnp_drop_step(pnp_engine* e, const float* d_gradcam, float* d_g0, float* d_agg, uint8_t* d_dropped,
                  int32_t* d_picks, int32_t iter, int32_t B, int32_t T, int32_t npick, int32_t max_picks,
                  void* stream);
/* The whole loop, device resident, no host sync.  d_g0 / d_agg: (B, L-1, P, P); d_picks: (B, drop_iter*npick)
 * int32 patch ids in pick order; drop_iter == 1 -> single forward, d_agg untouched. */
int pnp_drop_loop(pnp_engine* e, const float* d_images, const int64_t* d_ids, const int64_t* d_mask, int32_t ld,
                  int32_t B, int32_t L, int32_t head, int32_t drop_iter, int32_t npick, float* d_g0, float* d_agg,
                  int32_t* d_picks, float* d_logits, void* stream);

/* The loop on the maps of another kept text layer (stash_layer <= layer). */
int pnp_drop_loop_layer(pnp_engine* e, const float* d_images, const int64_t* d_ids, const int64_t* d_mask, int32_t ld,
                        int32_t B, int32_t L, int32_t layer, int32_t head, int32_t drop_iter, int32_t npick, float* d_g0,
                        float* d_agg, int32_t* d_picks, float* d_logits, void* stream);

/* ---- post-process: replaces PnP.py:348-403 / 424-481 + postprocess/densecrf/scores ------ */
typedef struct pnp_post_batch {
    int32_t B;
    const int32_t* H;            /* [B] original image height (label_trues[img].shape[0]) */
    const int32_t* W;            /* [B] */
    const int32_t* n_classes;    /* [B] selected classes (len(best_class_idx_list[img])) */
    const int32_t* has_bg;       /* [B] background channel prepended (PnP.py:373-379) */
    /* Mean_over_filtered_label_tokens plan (PnP.py:810-853): class c of image b sums tokens
     * tok_idx[cls_off[img_cls_off[b]+c] .. cls_off[img_cls_off[b]+c+1]) (indices into map[3:-1]),
     * then divides by cls_div when != 1 */
    const int32_t* img_cls_off;  /* [B+1] */
    const int32_t* cls_off;      /* [total_classes+1] */
    const int32_t* tok_idx;      /* [total_tokens] */
    const int32_t* cls_div;      /* [total_classes] */
    const int32_t* lut;          /* [B*lut_stride] argmax index -> dataset class id (PnP.py:390-399 folded) */
    int32_t lut_stride;
    const uint8_t* d_rgb;        /* device: concatenated original RGB images (sum H*W*3), CRF bilateral */
    const float* d_gt;           /* device: concatenated ground-truth label maps (sum H*W) or NULL */
    /* optional host-provided Gaussian taps (float64), image b uses blur_wts[blur_wt_off[b] + j] for
     * distance j = 0..radius, radius = blur_wt_off[b+1] - blur_wt_off[b] - 1.  NULL: computed by the
     * engine as scipy's _gaussian_kernel1d(0.05 * max(H, W), truncate 4).  The Python host passes
     * numpy-computed taps so the blur is bit-identical to scipy.ndimage.gaussian_filter. */
    const double* blur_wts;
    const int32_t* blur_wt_off;  /* [B+1] */
} pnp_post_batch;

/* Reserve device memory for post-processing batches up to these bounds (call once; allocates). */
int pnp_post_reserve(pnp_engine* e, int32_t max_batch, int64_t max_total_pixels, int32_t max_pixels_per_image,
                     int32_t max_channels, int32_t crf_chunk);
/* Upload descriptors/plan of a batch, compute blur taps, and (when want_crf) build both
 * permutohedral lattices + normalisers for its images.  Synchronises once (range check). */
int pnp_post_prepare(pnp_engine* e, const pnp_post_batch* batch, int32_t want_crf, void* stream);
/* Stage calls operate on the prepared batch and the engine's internal map buffers. */
int pnp_merge_tokens(pnp_engine* e, const float* d_gradcam, int32_t T, void* stream);          /* PnP.py:810-853 */
int pnp_threshold_upsample(pnp_engine* e, float threshold, int32_t scale01, void* stream);      /* PnP.py:348-379 */
int pnp_blur_minmax(pnp_engine* e, void* stream);                                               /* PnP.py:1149-1153 */
int pnp_densecrf(pnp_engine* e, int32_t iters, float pos_w, float pos_xy, float bi_w, float bi_xy, float bi_rgb,
                 void* stream);                                                                  /* PnP.py:1030-1074 */
/* argmax (+CRF marginals when from_crf) -> remap -> uint8 labels (concatenated, sum H*W) and, when
 * d_gt was given, hist[n_class*gt + pred] += 1 (PnP.py:1106-1146). d_hist: n_class*n_class uint64. */
int pnp_remap_hist(pnp_engine* e, int32_t from_crf, uint8_t* d_labels, unsigned long long* d_hist, int32_t n_class,
                   void* stream);
/* merge -> threshold/upsample -> [blur] -> [crf] -> argmax/remap/hist.  mode: bit0 blur, bit1 crf. */
int pnp_postprocess(pnp_engine* e, const float* d_gradcam, int32_t T, float threshold, int32_t scale01, int32_t mode,
                    uint8_t* d_labels, unsigned long long* d_hist, int32_t n_class, void* stream);

/* Both post-processing branches of a batch -- 1-drop (PnP.py:348-403, with Scale_0_1) and N-drop (PnP.py:424-481) --
 * with "blur+crf" in one DenseCRF run: the two problems share the image's lattices, so they are iterated side by side
 * (two channel groups per row).  Results equal two pnp_postprocess(..., mode 3, ...) calls bit for bit.
 * scale01_mask: bit 0 = Scale_0_1 on the 1-drop maps, bit 1 = on the N-drop maps (PnP.py: 1; the COCO driver
 * PnP_OVSS_0514_updated_segmentation_coco.py:436,527 scales both: 3). */
int pnp_postprocess_pair(pnp_engine* e, const float* d_gradcam_1drop, const float* d_gradcam_ndrop, int32_t T, float threshold,
                         int32_t scale01_mask, uint8_t* d_labels_1drop, unsigned long long* d_hist_1drop, uint8_t* d_labels_ndrop,
                         unsigned long long* d_hist_ndrop, int32_t n_class, void* stream);

/* ---- input side ("next" row: the tensors the reference's datasets hand to the model) -------
 * Dataset.py:434-443: transforms.Resize((S, S), BICUBIC) on the PIL image -> ToTensor -> Normalize(mean, std).
 * Pillow's 8-bit two-pass resampler (src/libImaging/Resample.c) in integer arithmetic, bit-exact: per axis and
 * output index o the host supplies (first tap, tap count) and 22-bit fixed-point taps (pnp_ovss.hip
 * .resample_table, the double-precision recipe of precompute_coeffs / normalize_coeffs_8bpc);
 * out = clip8((2^21 + sum taps * pixel) >> 22), horizontal pass first with a uint8 intermediate; then
 * (v / 255 - mean[c]) / std[c] in fp32.  Stateless; all pointers are device pointers except mean3 / std3. */
typedef struct pnp_pre_image {
    int64_t src_off;            /* byte offset of the image in d_rgb (H*W*3, row-major HWC uint8) */
    int64_t tmp_off;            /* byte offset of its H x S x 3 intermediate in d_tmp */
    int32_t H, W;
    int32_t kx_off, kx_size;    /* horizontal table at d_coef[kx_off]: S x (2 + kx_size) int32 = first, count, taps... */
    int32_t ky_off, ky_size;    /* vertical table, same layout */
} pnp_pre_image;
int pnp_preprocess_images(const uint8_t* d_rgb, const pnp_pre_image* d_desc, int32_t B, int32_t S, int32_t max_H,
                          const int32_t* d_coef, uint8_t* d_tmp, const float* mean3, const float* std3, float* d_out,
                          void* stream);

/* Dataset.py:349-445 / PnP.py:929-955 `Image.open(path).convert('RGB')` on device, for baseline (sequential Huffman, 8-bit,
 * grayscale or YCbCr 4:4:4 / 4:2:2 / 4:2:0) JPEG files: Pillow = libjpeg(-turbo) defaults, restated bit-exactly (islow
 * integer IDCT, fancy chroma upsampling, fixed-point YCbCr -> RGB).  The host parses the markers (pnp_ovss/jpeg.py) and
 * hands over the entropy-coded bytes, the file's own DHT / DQT tables and these descriptors.  Entropy decode: one workgroup
 * per restart segment; the segment is cut into up to 1024 sub-sequences decoded by one thread each from a guessed decoder
 * state, then re-decoded from the predecessor's exit state until no entry state changes (Huffman streams self-synchronise;
 * the fixed point IS the sequential decode), block counts and DC differences are prefix-summed and a last pass writes the
 * coefficients.  Output: the images' RGB bytes (H*W*3, HWC) at rgb_off -- the buffer pnp_preprocess_images and
 * pnp_post_batch.d_rgb read.  Workspaces: d_clean (sum of clean_cap bytes: the un-stuffed streams), d_seg_bits (n_segments
 * ints).  *d_err is set to 1 on a corrupt or truncated stream.  Stateless. */
typedef struct pnp_jpeg_image {
    int64_t data_off;           /* offset of the entropy-coded segment in d_data, a multiple of 16 */
    int64_t coef_off[3];        /* int16 elements: quantised coefficient blocks [blocks_y][blocks_x][64] per component */
    int64_t plane_off[3];       /* bytes: component sample planes, row stride blocks_x * 8 */
    int64_t rgb_off;            /* bytes */
    int32_t data_len, H, W, ncomp, hmax, vmax, mcux, mcuy;
    int32_t h[3], v[3], tq[3], td[3], ta[3], bx[3], by[3];
    int32_t tab, pad[2];
} pnp_jpeg_image;
typedef struct pnp_jpeg_tables {
    uint8_t counts[4][16];      /* [DC0, DC1, AC0, AC1]: DHT code counts per length 1..16 ... */
    uint8_t vals[4][256];       /* ... and symbols, as in the file (the look-ahead tables are built on the device) */
    int32_t quant[4][64];       /* natural (row-major) order */
} pnp_jpeg_tables;
typedef struct pnp_jpeg_segment {
    int64_t byte_off;           /* first byte of the restart interval, relative to data_off */
    int64_t clean_off;          /* offset of its un-stuffed copy in d_clean, a multiple of 16 */
    int32_t image, mcu0, nmcu;
    int32_t raw_len;            /* entropy-coded bytes of the interval, markers excluded */
    int32_t clean_cap;          /* bytes reserved at clean_off, >= raw_len + 32 */
    int32_t sub_bits;           /* sub-sequence length in bits: a multiple of 32, >= 8 * raw_len / 1024 */
    int32_t pad[2];
} pnp_jpeg_segment;
int pnp_jpeg_decode(const uint8_t* d_data, const pnp_jpeg_image* d_images, const pnp_jpeg_tables* d_tables,
                    const pnp_jpeg_segment* d_segments, int32_t n_images, int32_t n_segments, uint8_t* d_clean, int32_t* d_seg_bits,
                    int16_t* d_coef, int64_t coef_elems, uint8_t* d_planes, uint8_t* d_rgb, int32_t max_blocks_per_image,
                    int32_t max_pixels_per_image, int32_t* d_err, void* stream);

/* ---- introspection (tests / profiling) --------------------------------------------------- */
/* Named internal device buffers: "image_embeds" (fp32 B*N*D), "maps" (fp32 post-process maps),
 * "crf_q", "P", "dP", "crf_M" (int32 [2][B+1] lattice id bases), ... */
int pnp_get_buffer(pnp_engine* e, const char* name, void** d_ptr, size_t* bytes);
/* Live kernel timing for bench.py's roofline line: while enabled, every launch of the dominant
 * kernel family (the dense NT GEMMs with M = B*N rows: gemm_nt_x3_kernel<EPI> in the benchmarked split-bf16 mode
 * (csrc/gemm_x3.hip; three bf16 MFMAs per product, so the MFMA work issued is 3x the FLOPs reported here),
 * gemm_nt_wide_kernel<EPI> in the bf16 throughput mode, gemm_nt_big_kernel<float> in the exact-fp32 mode) is
 * bracketed by hipEvents on the launch stream.  pnp_profile_read
 * synchronises those events and returns launches, summed algorithmic FLOPs (2*M*N*K) and summed
 * kernel milliseconds since the last enable.  Off by default (no events on the hot path).  on = 1 brackets every launch,
 * on = n > 1 every n-th launch of the family (an event pair costs ~2.5 us of stream serialisation; sums cover the bracketed
 * launches only). */
int pnp_profile_enable(pnp_engine* e, int32_t on);
int pnp_profile_read(pnp_engine* e, int64_t* launches, double* flops, double* ms);
/* Same for a pipeline stage: 0 = the dense GEMMs (as pnp_profile_read); 1 = the DenseCRF mean-field iterations
 * (PnP.py:1066-1072: splat / lattice blur / slice + update kernels of one batch, bracketed as a whole) with `work` =
 * SURVEY.md 8d's algorithmic bytes and nothing else: per mean-field iteration (2 x 9 + 2) x K x H x W x 4 (splat + slice over
 * the 3 + 6 simplex vertices of a pixel, Q read + written), summed over the iterations of the bracketed batches; 2 = the same
 * brackets (same launches, same ms) with `work` = the lattice term that 8d leaves out: 2 x (2 x M_gauss + 3 x M_bilateral) x K x 4
 * per iteration (the value arrays of the M_* lattice points read and written once per two-axis blur pass). */
int pnp_profile_read_stage(pnp_engine* e, int32_t stage, int64_t* launches, double* work, double* ms);
/* Process-wide tuning switches (diagnostics and A/B measurements; the defaults are what is benchmarked).
 *   "streamk": the stream-K tail of the persistent split-bf16 GEMM (the ViT Linears of B/vit.py:45-51, 93-117 at M = B * N rows):
 *              0 = whole tiles only, 1 (default) = the last partial round of tiles is cut along K over all CUs when the cost
 *              model says it pays, 2 = whenever a launch has a partial last round.  Results are deterministic in every
 *              mode (fixed-order fix-up, no atomics) and differ between modes only by fp32 summation order.  Launches on a
 *              stream that is being captured into a hipGraph always take whole tiles (the tail's one-launch-at-a-time event
 *              chain reaches across streams). */
int pnp_set_tuning(const char* key, int32_t value);
/* Launches that used the engine's (e = NULL: the op-level entry points') stream-K workspace, and the give-up word of its
 * bounded spins (0 = no owner ever gave up waiting for a partial tile; anything else is a bug report).  Synchronises. */
int pnp_streamk_status(pnp_engine* e, int64_t* launches, uint32_t* gave_up);
/* Stand-alone operator entry points used by the parity tests (device pointers, see csrc/). */
int pnp_op_gemm(int32_t bf16, const void* d_A, int32_t lda, const void* d_B, int32_t ldb, int32_t M, int32_t N, int32_t K,
                const float* d_bias, const float* d_resid, int32_t ldr, float* d_out_f32, int32_t ldo, int32_t gelu,
                void* stream);
/* Same with the compute-type (bf16 / fp32) output and epilogue mode (0 linear, 1 GELU) exposed. */
int pnp_op_gemm_ex(int32_t bf16, const void* d_A, int32_t lda, const void* d_B, int32_t ldb, int32_t M, int32_t N, int32_t K,
                   const float* d_bias, const float* d_resid, int32_t ldr, float* d_out_f32, int32_t ldo, void* d_out_t,
                   int32_t ldo_t, int32_t mode, void* stream);
/* The transposed K / V projection form (B/vit.py:93-117 value, B/med.py:201-228 key/value with the image tokens
 * as output columns): out[m, (n / col_div) * col_pad + n % col_div] = A[m,:] . B[n,:] + bias_rows[m], output in
 * the compute type; col_div = 0 keeps n. */
int pnp_op_gemm_tokcols(int32_t bf16, const void* d_A, int32_t lda, const void* d_B, int32_t ldb, int32_t M, int32_t N,
                        int32_t K, const float* d_bias_rows, void* d_out_t, int32_t ldo_t, int32_t col_div,
                        int32_t col_pad, void* stream);
/* Split-bf16 ("bf16x3") form of the Linear (compute mode 2): operands are bf16 pairs x = hi + lo (pnp_op_split),
 * out = A.B^T accumulated as A_hi.B_hi + A_hi.B_lo + A_lo.B_hi in fp32, then one of the fp32-facing epilogues:
 *   d_out_f32, no col_div / bias_on_rows : + bias[n] (+ resid)            -> fp32 [M, ldo]
 *   d_out_f32, bias_on_rows / col_div    : + bias[m], token-column remap  -> fp32 [M, ldo]   (as pnp_op_gemm_tokcols)
 *   gelu, d_out_hi + d_out_lo            : + bias[n], erf GELU            -> split bf16 pair [M, ldo_t] */
int pnp_op_split(const float* d_in, void* d_hi, void* d_lo, int64_t n, void* stream);
int pnp_op_gemm_x3(const void* d_A_hi, const void* d_A_lo, int32_t lda, const void* d_B_hi, const void* d_B_lo, int32_t ldb,
                   int32_t M, int32_t N, int32_t K, const float* d_bias, int32_t bias_on_rows, const float* d_resid, int32_t ldr,
                   float* d_out_f32, int32_t ldo, void* d_out_hi, void* d_out_lo, int32_t ldo_t, int32_t gelu, int32_t col_div,
                   int32_t col_pad, void* stream);
/* Text-side form of the split-bf16 Linear (B/med.py:201-228 query / key / value, :321-325 and :393-411 dense layers at
 * M = B*L rows, and their backward): A is fp32 [M, lda] and is split into (hi, lo) bf16 by the kernel, the weight is a
 * (hi, lo) bf16 pair; out[m, n] = act(A[m,:] . B[n,:] + bias[n]) (+ resid[m, n]) in fp32.
 *   mode 0 linear | 1 erf GELU (pre-activation stashed to d_aux when given) | 2 multiply by GELU'(d_aux[m, n]). */
int pnp_op_gemm_x3a(const float* d_A, int32_t lda, const void* d_B_hi, const void* d_B_lo, int32_t ldb, int32_t M, int32_t N,
                    int32_t K, const float* d_bias, const float* d_resid, int32_t ldr, float* d_out_f32, int32_t ldo, int32_t mode,
                    float* d_aux, int32_t ld_aux, void* stream);
/* ViT self-attention of one block (B/vit.py:93-117): ctx = softmax(q k^T * scale) v per (image, head), head_dim 64.
 * d_qk [B*N, ld_qk]: q of head h at column h*64, k at column D + h*64; d_ctx [B*N, D].
 * fp32 mode: d_vt [D, ld_vt] = V^T, row h*64+d, column b*n_pad + token (n_pad a multiple of 64, pad columns zero).
 * bf16 mode: d_vt [B*N, ld_vt] = V in the natural layout, head h at column h*64 (e.g. the v third of fused q|k|v
 * rows); it is transposed by the kernel's LDS reads and n_pad is not used. */
int pnp_op_vit_attention(int32_t bf16, const void* d_qk, int32_t ld_qk, int32_t D, const void* d_vt, int32_t ld_vt,
                         int32_t n_pad, void* d_ctx, int32_t B, int32_t heads, int32_t N, float scale, void* stream);
/* The same attention in split-bf16 form (compute mode 2): d_qkv_hi / d_qkv_lo [B*N, ld_qkv] hold q | k | v of every head as
 * bf16 pairs (q of head h at column h*64, k at D + h*64, v at 2D + h*64); the context leaves as a pair [B*N, D]. */
int pnp_op_vit_attention_x3(const void* d_qkv_hi, const void* d_qkv_lo, int32_t ld_qkv, int32_t D, void* d_ctx_hi, void* d_ctx_lo,
                            int32_t B, int32_t heads, int32_t N, float scale, void* stream);
int pnp_op_layernorm(const float* d_x, const float* d_w, const float* d_b, float eps, int32_t rows, int32_t D,
                     float* d_y, void* stream);
/* Cross-attention over the image tokens as one operator (B/med.py:229-283 forward, its autograd backward):
 *   mode 0: probs = softmax(x . K^T / 8) -> d_probs ; out = probs . V          (d_nat = K rows, d_tr = V^T)
 *   mode 1: dP = dctx . V^T ; dS = probs (dP - rowsum(dP probs)) ; out = dS . K / 8   (d_nat = V rows, d_tr = K^T,
 *           d_probs read)
 *   mode 2: d_probs = dctx . V^T                                               (d_nat = V rows)
 * d_nat [B*N, ld_nat] token-major; d_tr [heads*64, ld_tr] with element (h*64+d, b*n_pad + n); x / out [B*L, ld]
 * in the compute type (bf16 or f32); d_probs fp32 [B, heads, L, n_stride]. head_dim is 64. */
int pnp_op_xattn(int32_t bf16, int32_t mode, const void* d_nat, int32_t ld_nat, const void* d_tr, int32_t ld_tr,
                 int32_t n_pad, const void* d_x, int32_t ldx, void* d_out, int32_t ldo, float* d_probs, int32_t n_stride,
                 int32_t B, int32_t L, int32_t N, int32_t heads, void* stream);
/* Diagnostics, development builds only (`make DEV=1`, libpnp_hip_dev.so: the product library reads no environment
 * variable and records no stamps -- there this returns PNP_ERR_STATE).  A DEV build started with PNP_GEMM_STAMPS=1
 * records, per workgroup of the wide-tile GEMM, the shader clock (slots 0-3) and the 100 MHz wall clock (slots 4-7) at
 * kernel start, after the first slab landed, after the main loop and after the epilogue; this copies the last launch's
 * stamps to the host. */
int pnp_dbg_gemm_stamps(uint64_t* host_out, int32_t max_blocks);
int pnp_op_cast(int32_t to_bf16, const float* d_in, void* d_out, int64_t n, void* stream);
/* The device-wide primitives of the DenseCRF lattice build as operators (the reference reaches them through pydensecrf's
 * hash table, PnP_OVSS_0514_updated_segmentation.py:1066-1070: here the lattice is built by a STABLE radix sort of the
 * (pixel, vertex) keys and prefix sums, csrc/sort.hip).  pnp_op_sort_pairs sorts n (u64 key, u32 value) pairs by the key
 * bits [begin_bit, end_bit) into d_keys_out / d_vals_out, equal keys keeping their input order; the input arrays are
 * overwritten.  h_seg_off (HOST array of n_seg + 1 ascending item offsets from 0 to n, n_seg <= 64; NULL = one segment):
 * the items of a segment are sorted among themselves and stay in its range (the images of a batch).
 * pnp_op_scan_i32: d_out[i] = sum of d_in[0..i] (inclusive) or d_in[0..i-1] (exclusive); in place allowed. */
int pnp_op_sort_pairs(uint64_t* d_keys_in, uint64_t* d_keys_out, uint32_t* d_vals_in, uint32_t* d_vals_out, int64_t n,
                      int32_t begin_bit, int32_t end_bit, const size_t* h_seg_off, int32_t n_seg, void* stream);
int pnp_op_scan_i32(const int32_t* d_in, int32_t* d_out, int64_t n, int32_t inclusive, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PNP_HIP_H */
