"""ORACLE (test infrastructure, not product code): BRUTE-FORCE dense CRF in float64 -- independent evidence for the
permutohedral restatement in densecrf_ref.c (pydensecrf, the reference's library, is not installable here).

Same model as the reference's call site (PnP_OVSS_0514_updated_segmentation.py:1030-1074): unary = -log(clip(softmax(maps),
1e-5, 1)); pairwise Potts terms  w_g * k_g + w_b * k_b  with the fully connected Gaussian kernels of Kraehenbuehl &
Koltun (NIPS'11, eq. 2/3 as implemented by densecrf: feature vectors divided by their standard deviations,
k(f_i, f_j) = exp(-|f_i - f_j|^2 / 2), self term included as the lattice filter includes it),
   k_g: f = (x, y) / 3            (addPairwiseGaussian(sxy=3, compat=7))
   k_b: f = (x, y) / 50, rgb / 5  (addPairwiseBilateral(sxy=50, srgb=5, compat=10))
NORMALIZE_SYMMETRIC: K~ = D^-1/2 K D^-1/2 with D = diag(K 1); mean field: Q <- softmax(-U + sum_k w_k K~_k Q), 10 times.
Every kernel entry is evaluated explicitly (N x N), so only images up to ~48 x 48 are practical.  The lattice is an
approximation of these kernels (Adams et al. 2010), so agreement is statistical: tests state the bounds they measured.
"""
import numpy as np


def densecrf_exact(rgb, maps, iters=10, pos_w=7.0, pos_xy=3.0, bi_w=10.0, bi_xy=50.0, bi_rgb=5.0):
    """rgb (H,W,3) uint8, maps (K,H,W) float -> (labels (H,W) int, Q (K,H,W) float64)."""
    K, H, W = maps.shape
    N = H * W
    m = np.asarray(maps, dtype=np.float64).reshape(K, N)
    p = np.exp(m - m.max(axis=0, keepdims=True))
    p /= p.sum(axis=0, keepdims=True)
    U = -np.log(np.clip(p, 1e-5, 1.0))                       # (K,N)
    yy, xx = np.mgrid[0:H, 0:W]
    xy = np.stack([xx.reshape(-1), yy.reshape(-1)], axis=1).astype(np.float64)
    col = np.asarray(rgb, dtype=np.float64).reshape(N, 3)

    def kernel(f):
        sq = (f * f).sum(1)
        d2 = sq[:, None] + sq[None, :] - 2.0 * (f @ f.T)
        Km = np.exp(-0.5 * np.maximum(d2, 0.0))
        nrm = 1.0 / np.sqrt(Km.sum(1) + 1e-20)
        return Km * nrm[:, None] * nrm[None, :]
    Kg = kernel(xy / pos_xy)
    Kb = kernel(np.concatenate([xy / bi_xy, col / bi_rgb], axis=1))

    def softmax(t):
        e = np.exp(t - t.max(axis=0, keepdims=True))
        return e / e.sum(axis=0, keepdims=True)
    Q = softmax(-U)
    for _ in range(iters):
        Q = softmax(-U + pos_w * (Q @ Kg) + bi_w * (Q @ Kb))   # kernels are symmetric
    return Q.argmax(axis=0).reshape(H, W), Q.reshape(K, H, W)
