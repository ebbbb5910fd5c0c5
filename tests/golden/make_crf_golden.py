#!/usr/bin/env python
"""Pin the DenseCRF oracle to the reference's own library -- for a maintainer whose machine HAS pydensecrf
(lucasb-eyer/pydensecrf, the package the reference's README.md:28-30 asks for; it is not installable in the build
container: no network, no Eigen).

    pip install pydensecrf        # or: git clone + python setup.py install, as the reference's README says
    python tests/golden/make_crf_golden.py            # writes tests/golden/crf_pydensecrf.npz

It runs EXACTLY the call sequence of the reference's `densecrf(image, mask)` (PnP_OVSS_0514_updated_segmentation.py:
1030-1074: softmax over channels -> unary_from_softmax -> DenseCRF2D(W, H, C) -> addPairwiseGaussian(sxy=3, compat=7)
-> addPairwiseBilateral(sxy=50, srgb=5, rgbim=image, compat=10) -> inference(10) -> argmax) on seeded inputs and stores
inputs, marginals and label maps.  tests/test_oracle_golden.py::test_crf_oracle_vs_pydensecrf_fixture consumes the file
when it exists (and is reported as SKIPPED when it does not); commit the .npz to close the "parity unpinned" note.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "pnp-ovss_amd"))


def crf_cases():
    """Seeded (rgb, maps) cases shared by this script and the tests: block-coloured noisy images and blurred-blob score
    maps in [0, 1] (what the blur + min-max stage hands to densecrf), incl. a non-square and a 21-channel case."""
    from pnp_ovss import synth
    cases = []
    for seed, (H, W, K, noise) in enumerate([(40, 40, 4, 4), (32, 48, 3, 12), (36, 36, 6, 4), (40, 40, 21, 4), (24, 40, 2, 12),
                                              (96, 128, 5, 8)]):
        rng = np.random.default_rng(seed)
        rgb, _ = synth.synth_images(1, 128, seed=seed, noise=noise)
        rgb = np.ascontiguousarray(rgb[0, :H, :W])
        maps = np.zeros((K, H, W), np.float32)
        yy, xx = np.mgrid[0:H, 0:W]
        for k in range(K):
            cy, cx, s = rng.uniform(0, H), rng.uniform(0, W), rng.uniform(4, 10)
            m = np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * s * s))
            maps[k] = (m - m.min()) / (m.max() - m.min())
        cases.append((rgb, maps))
    return cases


def main():
    import torch
    import torch.nn.functional as F
    import pydensecrf.densecrf as dcrf
    import pydensecrf.utils as utils
    out = {}
    for i, (rgb, maps) in enumerate(crf_cases()):
        # --- PnP_OVSS_0514_updated_segmentation.py:1036-1073, argument for argument
        MAX_ITER, POS_W, POS_XY_STD, Bi_W, Bi_XY_STD, Bi_RGB_STD = 10, 7, 3, 10, 50, 5
        h, w = rgb.shape[:2]
        mask = F.softmax(torch.from_numpy(maps), dim=0).numpy()
        c = mask.shape[0]
        U = np.ascontiguousarray(utils.unary_from_softmax(mask))
        image = np.ascontiguousarray(rgb)
        d = dcrf.DenseCRF2D(w, h, c)
        d.setUnaryEnergy(U)
        d.addPairwiseGaussian(sxy=POS_XY_STD, compat=POS_W)
        d.addPairwiseBilateral(sxy=Bi_XY_STD, srgb=Bi_RGB_STD, rgbim=image, compat=Bi_W)
        Q = np.array(d.inference(MAX_ITER)).reshape((c, h, w))
        out[f"rgb_{i}"], out[f"maps_{i}"] = rgb, maps
        out[f"Q_{i}"] = Q.astype(np.float32)
        out[f"labels_{i}"] = np.argmax(Q, axis=0).astype(np.uint8)
    out["n"] = np.int32(len(crf_cases()))
    np.savez_compressed(os.path.join(HERE, "crf_pydensecrf.npz"), **out)
    print("wrote crf_pydensecrf.npz with", int(out["n"]), "cases")


if __name__ == "__main__":
    main()
