#!/usr/bin/env python
"""Same role as the reference's Calculate_mIoU.py (:204-256): sum the per-batch confusion matrices
a run saved under {save_path}/all_drop_hist_with_filtered_caption/ and print the metrics."""
import argparse
import glob
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pnp_ovss import host  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--save_path", required=True)
    ap.add_argument("--data_type", default="voc")
    ap.add_argument("--subdir", default="all_drop_hist_with_filtered_caption")
    a = ap.parse_args()
    files = sorted(glob.glob(os.path.join(a.save_path, a.subdir, "*.npy")))
    if not files:
        raise SystemExit(f"no .npy histograms under {a.save_path}/{a.subdir}")
    hist = sum(np.load(f) for f in files)
    s = host.scores_from_hist(hist)
    print(f"{len(files)} batches")
    for k in ("Pixel Accuracy", "Mean Accuracy", "Frequency Weighted IoU", "Mean IoU"):
        print(f"{k}: {s[k]:.6f}")
