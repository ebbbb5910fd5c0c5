/* ORACLE (test infrastructure, not product code): plain-C restatement of the dense-CRF
 * mean-field step the reference calls at PnP_OVSS_0514_updated_segmentation.py:1030-1074
 * (`densecrf`): softmax -> unary_from_softmax -> DenseCRF2D(W,H,K) + addPairwiseGaussian(sxy=3,
 * compat=7) + addPairwiseBilateral(sxy=50, srgb=5, compat=10) -> inference(10) -> argmax.
 *
 * PARITY UNPINNED: the arithmetic lives in the un-vendored third-party package
 * lucasb-eyer/pydensecrf (Cython over Kraehenbuehl & Koltun's densecrf C++ + Eigen; version not
 * pinned by the reference: README.md:28-30).  Its sources are absent from /root/reference and it is
 * not installed, so this file restates the PUBLISHED algorithm (Kraehenbuehl & Koltun NIPS'11
 * mean-field; Adams et al. 2010 permutohedral lattice as used by densecrf: sequential splat over
 * pixels in index order, d+1 axis blurs v + 0.5*(n1+n2), slice scaled by 1/(1+2^-d); Potts
 * compatibility; DIAG_KERNEL; NORMALIZE_SYMMETRIC with norm = 1/sqrt(K*1 + 1e-20)) and is anchored
 * on the reference's call site only.  exp/log are include/pnp_math.h's fixed fmaf sequences (Eigen's
 * vectorised exp is not reproducible anyway) so the HIP path can be compared bit-for-bit.
 * Closed point (was open until round 3): densecrf's Permutohedral::compute has a scalar path, followed here (the blur is
 * `old + 0.5 * (n1 + n2)` with a double literal), and an SSE path with the same structure in float32 (`0.5f * (n1 + n2)`).
 * They cannot differ: (n1 + n2) is float + float and is rounded to float in both forms, 0.5 * x is exact, and old + t is exact
 * in double, so its single rounding to float IS the float addition -- tests/test_oracle_golden.py asserts bit-identical marginals
 * for the two forms.  The same test bounds what a different exp / log (libm here; torch / numpy / Eigen in the reference's
 * stack) does to the labels: 0 flips on 158 016 pixels, |dQ| <= 1.8e-3.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#include "../include/pnp_math.h"

/* Sensitivity variants, for tests/test_oracle_golden.py only (pnp_oracle_densecrf_variant): what the arithmetic this file
 * FIXES BY DEFINITION could be in the reference's own stack, so that the label flips it can cause are measured, not guessed.
 *   bit 0: libm expf / logf in place of include/pnp_math.h (torch's softmax, numpy's log and Eigen's exp each use their own)
 *   bit 1: float32 lattice blur `old + 0.5f * (n1 + n2)` (densecrf's SSE path) in place of the scalar path (provably identical)
 * 0 = the oracle the HIP path is compared with. */
static int g_variant = 0;
static float oracle_expf(float x) { return (g_variant & 1) ? expf(x) : pnp_expf(x); }
static float oracle_logf(float x) { return (g_variant & 1) ? logf(x) : pnp_logf(x); }

typedef struct {
    int d, N, M;
    int *offset;        /* N*(d+1) lattice point id per simplex vertex */
    float *bary;        /* N*(d+1) */
    int *n1, *n2;       /* (d+1)*M blur neighbours, -1 = absent */
} lattice_t;

/* ------------------------------------------------------------------ hash of d-short keys */
typedef struct {
    int key_size, filled, capacity;
    short *keys;
    int *table;
} hash_t;

static size_t hash_key(const hash_t *h, const short *k) {
    size_t r = 0;
    for (int i = 0; i < h->key_size; i++) {
        r += (size_t)(long)k[i];
        r *= 1664525u;
    }
    return r;
}

static void hash_init(hash_t *h, int key_size, int n_elements) {
    h->key_size = key_size;
    h->filled = 0;
    h->capacity = 2 * n_elements + 16;
    h->keys = (short *)malloc(sizeof(short) * (size_t)(h->capacity / 2 + 10) * key_size);
    h->table = (int *)malloc(sizeof(int) * h->capacity);
    for (int i = 0; i < h->capacity; i++) h->table[i] = -1;
}

static void hash_grow(hash_t *h) {
    int old_cap = h->capacity;
    h->capacity *= 2;
    h->keys = (short *)realloc(h->keys, sizeof(short) * (size_t)(old_cap + 10) * h->key_size);
    free(h->table);
    h->table = (int *)malloc(sizeof(int) * h->capacity);
    for (int i = 0; i < h->capacity; i++) h->table[i] = -1;
    for (int i = 0; i < h->filled; i++) {
        size_t e = hash_key(h, h->keys + (size_t)i * h->key_size) % (size_t)h->capacity;
        while (h->table[e] >= 0) {
            e++;
            if (e == (size_t)h->capacity) e = 0;
        }
        h->table[e] = i;
    }
}

static int hash_find(hash_t *h, const short *k, int create) {
    if (2 * h->filled >= h->capacity) hash_grow(h);
    size_t e = hash_key(h, k) % (size_t)h->capacity;
    for (;;) {
        int id = h->table[e];
        if (id == -1) {
            if (!create) return -1;
            memcpy(h->keys + (size_t)h->filled * h->key_size, k, sizeof(short) * h->key_size);
            h->table[e] = h->filled;
            return h->filled++;
        }
        if (memcmp(h->keys + (size_t)id * h->key_size, k, sizeof(short) * h->key_size) == 0) return id;
        e++;
        if (e == (size_t)h->capacity) e = 0;
    }
}

/* ------------------------------------------------------------------ lattice construction */
/* feature: column-major d x N (feature[i*d + j] = coordinate j of pixel i). */
static void lattice_init(lattice_t *L, const float *feature, int d, int N) {
    L->d = d;
    L->N = N;
    L->offset = (int *)malloc(sizeof(int) * (size_t)N * (d + 1));
    L->bary = (float *)malloc(sizeof(float) * (size_t)N * (d + 1));
    hash_t ht;
    hash_init(&ht, d, N * (d + 1));

    float scale_factor[16], elevated[17], rem0[17], barycentric[18];
    short rank[17], key[17];
    short canonical[17 * 17];
    for (int i = 0; i <= d; i++) {
        for (int j = 0; j <= d - i; j++) canonical[i * (d + 1) + j] = (short)i;
        for (int j = d - i + 1; j <= d; j++) canonical[i * (d + 1) + j] = (short)(i - (d + 1));
    }
    float inv_std_dev = (float)(sqrt(2.0 / 3.0) * (d + 1));
    for (int i = 0; i < d; i++) scale_factor[i] = (float)(1.0 / sqrt((double)((i + 2) * (i + 1))) * inv_std_dev);

    for (int k = 0; k < N; k++) {
        const float *f = feature + (size_t)k * d;
        float sm = 0;
        for (int j = d; j > 0; j--) {
            float cf = f[j - 1] * scale_factor[j - 1];
            elevated[j] = sm - j * cf;
            sm += cf;
        }
        elevated[0] = sm;

        float down_factor = 1.0f / (d + 1);
        float up_factor = (float)(d + 1);
        int sum = 0;
        for (int i = 0; i <= d; i++) {
            float v = down_factor * elevated[i];
            float up = ceilf(v) * up_factor;
            float down = floorf(v) * up_factor;
            int rd2;
            if (up - elevated[i] < elevated[i] - down) rd2 = (short)up;
            else rd2 = (short)down;
            rem0[i] = (float)rd2;
            sum = (int)((float)sum + rd2 * down_factor);
        }
        for (int i = 0; i <= d; i++) rank[i] = 0;
        for (int i = 0; i < d; i++) {
            double di = elevated[i] - rem0[i];
            for (int j = i + 1; j <= d; j++) {
                if (di < elevated[j] - rem0[j]) rank[i]++;
                else rank[j]++;
            }
        }
        for (int i = 0; i <= d; i++) {
            rank[i] += (short)sum;
            if (rank[i] < 0) {
                rank[i] += (short)(d + 1);
                rem0[i] += (float)(d + 1);
            } else if (rank[i] > d) {
                rank[i] -= (short)(d + 1);
                rem0[i] -= (float)(d + 1);
            }
        }
        for (int i = 0; i <= d + 1; i++) barycentric[i] = 0;
        for (int i = 0; i <= d; i++) {
            float v = (elevated[i] - rem0[i]) * down_factor;
            barycentric[d - rank[i]] += v;
            barycentric[d - rank[i] + 1] -= v;
        }
        barycentric[0] += 1.0f + barycentric[d + 1];

        for (int remainder = 0; remainder <= d; remainder++) {
            for (int i = 0; i < d; i++) key[i] = (short)(rem0[i] + canonical[remainder * (d + 1) + rank[i]]);
            L->offset[(size_t)k * (d + 1) + remainder] = hash_find(&ht, key, 1);
            L->bary[(size_t)k * (d + 1) + remainder] = barycentric[remainder];
        }
    }
    int M = ht.filled;
    L->M = M;
    L->n1 = (int *)malloc(sizeof(int) * (size_t)(d + 1) * M);
    L->n2 = (int *)malloc(sizeof(int) * (size_t)(d + 1) * M);
    short n1[17], n2[17];
    for (int j = 0; j <= d; j++) {
        for (int i = 0; i < M; i++) {
            const short *kk = ht.keys + (size_t)i * d;
            for (int k = 0; k < d; k++) {
                n1[k] = (short)(kk[k] - 1);
                n2[k] = (short)(kk[k] + 1);
            }
            if (j < d) {
                n1[j] = (short)(kk[j] + d);
                n2[j] = (short)(kk[j] - d);
            }
            L->n1[(size_t)j * M + i] = hash_find(&ht, n1, 0);
            L->n2[(size_t)j * M + i] = hash_find(&ht, n2, 0);
        }
    }
    free(ht.keys);
    free(ht.table);
}

static void lattice_free(lattice_t *L) {
    free(L->offset);
    free(L->bary);
    free(L->n1);
    free(L->n2);
}

/* out/in: N x vs, pixel-major (in[i*vs + k]); may alias. */
static void lattice_compute(const lattice_t *L, float *out, const float *in, int vs) {
    int d = L->d, N = L->N, M = L->M;
    size_t sz = (size_t)(M + 2) * vs;
    float *values = (float *)calloc(sz, sizeof(float));
    float *new_values = (float *)calloc(sz, sizeof(float));
    for (int i = 0; i < N; i++) {
        for (int j = 0; j <= d; j++) {
            int o = L->offset[(size_t)i * (d + 1) + j] + 1;
            float w = L->bary[(size_t)i * (d + 1) + j];
            for (int k = 0; k < vs; k++) values[(size_t)o * vs + k] += w * in[(size_t)i * vs + k];
        }
    }
    for (int j = 0; j <= d; j++) {
        for (int i = 0; i < M; i++) {
            float *old_val = values + (size_t)(i + 1) * vs;
            float *new_val = new_values + (size_t)(i + 1) * vs;
            int a = L->n1[(size_t)j * M + i] + 1;
            int b = L->n2[(size_t)j * M + i] + 1;
            float *n1_val = values + (size_t)a * vs;
            float *n2_val = values + (size_t)b * vs;
            if (g_variant & 2) {
                for (int k = 0; k < vs; k++) {
                    float t = n1_val[k] + n2_val[k];
                    t = 0.5f * t;
                    new_val[k] = old_val[k] + t;
                }
            } else {
                for (int k = 0; k < vs; k++) new_val[k] = (float)(old_val[k] + 0.5 * (n1_val[k] + n2_val[k]));
            }
        }
        float *t = values;
        values = new_values;
        new_values = t;
    }
    float alpha = 1.0f / (1 + powf(2, (float)-d));
    for (int i = 0; i < N; i++) {
        for (int k = 0; k < vs; k++) out[(size_t)i * vs + k] = 0;
        for (int j = 0; j <= d; j++) {
            int o = L->offset[(size_t)i * (d + 1) + j] + 1;
            float w = L->bary[(size_t)i * (d + 1) + j];
            for (int k = 0; k < vs; k++) out[(size_t)i * vs + k] += w * values[(size_t)o * vs + k] * alpha;
        }
    }
    free(values);
    free(new_values);
}

static float *lattice_norm(const lattice_t *L) {
    int N = L->N;
    float *ones = (float *)malloc(sizeof(float) * N);
    for (int i = 0; i < N; i++) ones[i] = 1.0f;
    lattice_compute(L, ones, ones, 1);
    for (int i = 0; i < N; i++) ones[i] = (float)(1.0 / sqrt(ones[i] + 1e-20));
    return ones;
}

static void exp_and_normalize(float *Q, const float *in, int K, int N) {
    for (int i = 0; i < N; i++) {
        const float *b = in + (size_t)i * K;
        float m = b[0];
        for (int k = 1; k < K; k++)
            if (b[k] > m || b[k] != b[k]) m = b[k];
        float s = 0;
        for (int k = 0; k < K; k++) {
            float e = oracle_expf(b[k] - m);
            Q[(size_t)i * K + k] = e;
            s += e;
        }
        for (int k = 0; k < K; k++) Q[(size_t)i * K + k] = Q[(size_t)i * K + k] / s;
    }
}

static void pairwise_apply(const lattice_t *L, const float *norm, float w, float *out, const float *Q, int K) {
    int N = L->N;
    for (int i = 0; i < N; i++)
        for (int k = 0; k < K; k++) out[(size_t)i * K + k] = Q[(size_t)i * K + k] * norm[i];
    lattice_compute(L, out, out, K);
    for (int i = 0; i < N; i++)
        for (int k = 0; k < K; k++) out[(size_t)i * K + k] = -w * (out[(size_t)i * K + k] * norm[i]);
}

/* F.softmax(dim=0) then unary_from_softmax: -log(clip(p, 1e-5, 1)); maps (K,N) channel-major -> unary (N,K) pixel-major */
static void unary_stage(const float *maps, int K, int N, float *unary) {
    for (int i = 0; i < N; i++) {
        float m = maps[i];
        for (int k = 1; k < K; k++) {
            float v = maps[(size_t)k * N + i];
            if (v > m || v != v) m = v;
        }
        float s = 0;
        float e[1024];
        for (int k = 0; k < K; k++) {
            e[k] = oracle_expf(maps[(size_t)k * N + i] - m);
            s += e[k];
        }
        for (int k = 0; k < K; k++) {
            float p = e[k] / s;
            if (p < 1e-5f) p = 1e-5f;
            else if (p > 1.0f) p = 1.0f;
            unary[(size_t)i * K + k] = -oracle_logf(p);
        }
    }
}

/* the unary stage alone, (K,N) in -> (K,N) out (tests: against torch F.softmax + numpy -log(clip)) */
int pnp_oracle_unary(const float *maps, int K, int N, float *unary_kn) {
    float *u = (float *)malloc(sizeof(float) * (size_t)N * K);
    g_variant = 0;
    unary_stage(maps, K, N, u);
    for (int i = 0; i < N; i++)
        for (int k = 0; k < K; k++) unary_kn[(size_t)k * N + i] = u[(size_t)i * K + k];
    free(u);
    return 0;
}

/* maps: (K,H,W) blurred score maps; rgb: (H,W,3) uint8.
 * q_out (optional): (K,H,W) marginals; map_out: (H,W) float32 argmax labels;
 * stats_out (optional): [M_gauss, M_bilateral]; unary_kn (optional): (K,N) unary energies to use instead of the unary stage. */
static int densecrf_impl(const float *maps, const uint8_t *rgb, int K, int H, int W, int iters,
                         float pos_w, float pos_xy, float bi_w, float bi_xy, float bi_rgb,
                         float *q_out, float *map_out, int *stats_out, const float *unary_kn) {
    int N = H * W;
    float *unary = (float *)malloc(sizeof(float) * (size_t)N * K);
    if (unary_kn) {
        for (int i = 0; i < N; i++)
            for (int k = 0; k < K; k++) unary[(size_t)i * K + k] = unary_kn[(size_t)k * N + i];
    } else {
        unary_stage(maps, K, N, unary);
    }
    float *fg = (float *)malloc(sizeof(float) * (size_t)N * 2);
    float *fb = (float *)malloc(sizeof(float) * (size_t)N * 5);
    for (int j = 0; j < H; j++)
        for (int i = 0; i < W; i++) {
            int n = j * W + i;
            fg[n * 2 + 0] = i / pos_xy;
            fg[n * 2 + 1] = j / pos_xy;
            fb[n * 5 + 0] = i / bi_xy;
            fb[n * 5 + 1] = j / bi_xy;
            fb[n * 5 + 2] = rgb[n * 3 + 0] / bi_rgb;
            fb[n * 5 + 3] = rgb[n * 3 + 1] / bi_rgb;
            fb[n * 5 + 4] = rgb[n * 3 + 2] / bi_rgb;
        }
    lattice_t Lg, Lb;
    lattice_init(&Lg, fg, 2, N);
    lattice_init(&Lb, fb, 5, N);
    free(fg);
    free(fb);
    if (stats_out) {
        stats_out[0] = Lg.M;
        stats_out[1] = Lb.M;
    }
    float *ng = lattice_norm(&Lg);
    float *nb = lattice_norm(&Lb);

    float *Q = (float *)malloc(sizeof(float) * (size_t)N * K);
    float *tmp1 = (float *)malloc(sizeof(float) * (size_t)N * K);
    float *tmp2 = (float *)malloc(sizeof(float) * (size_t)N * K);
    for (size_t t = 0; t < (size_t)N * K; t++) tmp1[t] = -unary[t];
    exp_and_normalize(Q, tmp1, K, N);
    for (int it = 0; it < iters; it++) {
        for (size_t t = 0; t < (size_t)N * K; t++) tmp1[t] = -unary[t];
        pairwise_apply(&Lg, ng, pos_w, tmp2, Q, K);
        for (size_t t = 0; t < (size_t)N * K; t++) tmp1[t] -= tmp2[t];
        pairwise_apply(&Lb, nb, bi_w, tmp2, Q, K);
        for (size_t t = 0; t < (size_t)N * K; t++) tmp1[t] -= tmp2[t];
        exp_and_normalize(Q, tmp1, K, N);
    }
    for (int i = 0; i < N; i++) {
        int best = 0;
        for (int k = 1; k < K; k++) {
            float a = Q[(size_t)i * K + best], b = Q[(size_t)i * K + k];
            if (a == a && (b > a || b != b)) best = k;
        }
        map_out[i] = (float)best;
        if (q_out)
            for (int k = 0; k < K; k++) q_out[(size_t)k * N + i] = Q[(size_t)i * K + k];
    }
    free(Q);
    free(tmp1);
    free(tmp2);
    free(ng);
    free(nb);
    free(unary);
    lattice_free(&Lg);
    lattice_free(&Lb);
    return 0;
}

int pnp_oracle_densecrf(const float *maps, const uint8_t *rgb, int K, int H, int W, int iters,
                        float pos_w, float pos_xy, float bi_w, float bi_xy, float bi_rgb,
                        float *q_out, float *map_out, int *stats_out) {
    g_variant = 0;
    return densecrf_impl(maps, rgb, K, H, W, iters, pos_w, pos_xy, bi_w, bi_xy, bi_rgb, q_out, map_out, stats_out, NULL);
}

/* the same mean-field under a sensitivity variant (see g_variant) and / or with externally computed unary energies */
int pnp_oracle_densecrf_variant(const float *maps, const uint8_t *rgb, int K, int H, int W, int iters,
                                float pos_w, float pos_xy, float bi_w, float bi_xy, float bi_rgb,
                                float *q_out, float *map_out, int variant, const float *unary_kn) {
    g_variant = variant;
    int r = densecrf_impl(maps, rgb, K, H, W, iters, pos_w, pos_xy, bi_w, bi_xy, bi_rgb, q_out, map_out, NULL, unary_kn);
    g_variant = 0;
    return r;
}
