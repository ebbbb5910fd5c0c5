#!/usr/bin/env python
"""bench.py -- PnP-OVSS hot path on MI355X: images/sec at 336^2, drop_iter=4, blur+CRF.

One "step" = one pass of the whole hot path over one batch of synthetic images resident in HBM:
  4 x [ViT-L/16 + BERT/cross-attention forward, analytic dL/dP backward, GradCAM gather, salience
  drop] -> word-piece merge -> threshold + bilinear upsample (+ Scale_0_1) -> Gaussian blur ->
  DenseCRF (10 mean-field iterations, lattices rebuilt for the batch) -> argmax/remap/histogram,
  for BOTH branches the VOC driver runs (1-drop and N-drop: PnP_OVSS_0514_updated_segmentation.py:
  348-403 and :424-481), i.e. two blur+CRF passes per image like the reference.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 35] [--dtype bf16|f32] [--no-cpu-baseline]
For N > 1 launch with torch.distributed.run (one rank per GPU, RCCL): images are sharded across
ranks (weak scaling: fixed per-GPU batch), weights are broadcast from rank 0 over RCCL, the
confusion histogram is all-reduced and label maps gathered to rank 0 after the timed region.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "pnp-ovss_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

N_CLASSES = 20
IMG = 336
LAYER, HEAD, DROP_ITER, THRESH = 7, 9, 4, 0.15
NOISE = 4          # +-4 grey levels of per-pixel noise on the 8x8-block synthetic images (see synth.synth_images)
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}     # /opt/skills/guides/MI355X_MICROARCH.md (dense)


def cpu_baseline(cfg, seed_w, n_images=1):
    """The oracle (numpy + C restatement of the reference path) timed on this box's host cores on a
    bounded sample of the same workload: `n_images` image(s), full path (4 drop iterations, both
    branches, blur + CRF).  Reported next to the GPU number; not the optimisation target."""
    from pnp_ovss import synth
    from oracle import pipeline_np as OP
    W = synth.synth_state_dict(cfg, seed_w)
    rgb, imgs = synth.synth_images(n_images, IMG, seed=1234, noise=NOISE)
    ids, mask = synth.synth_tokens(cfg, [N_CLASSES] * n_images, seed=1234)
    pieces = [[f"t{i}" for i in range(N_CLASSES)]] * n_images
    best = [list(range(N_CLASSES))] * n_images
    t0 = time.perf_counter()
    OP.segment_batch(W, cfg, imgs, ids, mask, pieces, best, list(rgb), [(IMG, IMG)] * n_images, data_type="voc",
                     drop_iter=DROP_ITER, layer=LAYER, head=HEAD, threshold=THRESH, mode="blur+crf")
    dt = time.perf_counter() - t0
    return {"value": n_images / dt, "unit": "images/sec", "cores": os.cpu_count(), "kind": "port",
            "sample": f"{n_images} image(s) 336x336, 20-class prompt, drop_iter=4, 1-drop + N-drop blur+CRF, "
                      f"numpy/OpenBLAS + gcc oracle, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=35, help="images per step per GPU (--batch_size 35, Run_seg.sh)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--crf-chunk", type=int, default=0)
    ap.add_argument("--skip-1drop", action="store_true", help="PnPc.py behaviour (COCO driver): N-drop branch only")
    ap.add_argument("--separate-crf", action="store_true", help="run the 1-drop and N-drop DenseCRF as two passes (default: paired)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--overlap", action="store_true",
                    help="software-pipeline batches over two HIP streams (drop loop of batch i+1 beside the post-process "
                         "of batch i); +2%% images/sec, off by default so the per-kernel event timing stays undisturbed")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = "RANK" in os.environ          # launched by torch.distributed.run (also for N = 1)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from pnp_ovss import config as C, synth
    from pnp_ovss.hip import Engine
    cfg = C.blip_itm_large(IMG)
    B = a.batch
    e = Engine(cfg, max_batch=B, max_text_len=32, stash_layer=LAYER, bf16=(a.dtype == "bf16"), device=local)

    # ---- weights: rank 0 materialises the seeded weights, RCCL broadcast over xGMI to the others
    shapes = synth.param_shapes(cfg)
    total = sum(int(np.prod(s)) for s in shapes.values())
    flat = torch.empty(total, device=dev, dtype=torch.float32)
    if rank == 0:
        o = 0
        for n, shp in shapes.items():
            w = synth.synth_tensor(n, shp, 0)
            flat[o:o + w.size].copy_(torch.from_numpy(w.reshape(-1)))
            o += w.size
    if distributed:
        dist.broadcast(flat, src=0)
    sd, o = {}, 0
    for n, shp in shapes.items():
        k = int(np.prod(shp))
        sd[n] = flat[o:o + k].view(*shp)
        o += k
    e.load_state_dict(sd)
    del sd, flat
    torch.cuda.empty_cache()

    # ---- synthetic inputs, resident in HBM before the timed region (different images per rank)
    rgb, imgs = synth.synth_images(B, IMG, seed=1234 + rank, noise=NOISE)
    ids, mask = synth.synth_tokens(cfg, [N_CLASSES] * B, seed=1234 + rank)
    L = int(mask.sum(1).max())
    gt = np.random.default_rng(rank).integers(0, 21, size=(B, IMG, IMG)).astype(np.float32)
    d_img = torch.from_numpy(imgs).to(dev)
    d_ids, d_mask = torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev)
    d_rgb = torch.from_numpy(rgb.reshape(-1)).to(dev)
    d_gt = torch.from_numpy(gt.reshape(-1)).to(dev)
    sizes = [(IMG, IMG)] * B
    plans = [[([i], 1) for i in range(N_CLASSES)]] * B         # one word-piece per class
    luts = [list(range(N_CLASSES + 1))] * B                    # index i -> class id (background 0)
    e.post_reserve(B, B * IMG * IMG, IMG * IMG, N_CLASSES + 1, a.crf_chunk)
    hist1 = torch.zeros(21 * 21, device=dev, dtype=torch.int64)
    histn = torch.zeros(21 * 21, device=dev, dtype=torch.int64)
    state = {}

    # Two HIP streams (model / post-process).  With --overlap the drop loop of batch i+1 is enqueued
    # before the post-process of batch i; every batch still runs the complete path inside the timed region.
    if a.overlap:
        s_model, s_post = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    else:
        s_model = s_post = torch.cuda.current_stream(dev)
    keep = []

    def model_part():
        with torch.cuda.stream(s_model):
            g0, agg, picks, _ = e.drop_loop(d_img, d_ids, d_mask, L, HEAD, DROP_ITER)
            ev = torch.cuda.Event()
            ev.record(s_model)
        return g0, agg, ev

    def post_part(g0, agg, ev):
        with torch.cuda.stream(s_post):
            s_post.wait_event(ev)
            e.post_prepare(sizes, plans, luts, [True] * B, rgb=d_rgb, gt=d_gt, want_crf=True)
            if a.skip_1drop:
                state["ln"] = e.postprocess(agg, THRESH, False, "blur+crf", 21, histn)
            elif a.separate_crf:
                state["l1"] = e.postprocess(g0, THRESH, True, "blur+crf", 21, hist1)
                state["ln"] = e.postprocess(agg, THRESH, False, "blur+crf", 21, histn)
            else:       # both branches in one DenseCRF run (two channel groups per row; identical results)
                state["l1"], state["ln"] = e.postprocess_pair(g0, agg, THRESH, 21, hist1, histn)
        keep.append((g0, agg))

    def run(n):
        if not a.overlap:
            for _ in range(n):
                post_part(*model_part())
            return
        nxt = model_part()
        for i in range(n):
            cur = nxt
            if i + 1 < n:
                nxt = model_part()          # enqueue batch i+1's drop loop before batch i's post-process
            post_part(*cur)

    def sync():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    run(a.warmup)
    sync()
    keep.clear()
    e.profile_enable(True)
    t0 = time.perf_counter()
    run(a.steps)
    sync()
    dt = time.perf_counter() - t0
    launches, flops, ms = e.profile_read()
    e.profile_enable(False)
    if distributed:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        dist.all_reduce(histn)                                   # cross-rank metric reduce (replaces the
        dist.all_reduce(hist1)                                   # reference's .npy files, PnP.py:513-520)
        gathered = [torch.empty_like(state["ln"]) for _ in range(world)] if rank == 0 else None
        dist.gather(state["ln"], gathered, dst=0)                # mask-gather of uint8 label maps to rank 0

    if rank == 0:
        # HBM-side bytes per launch of the same kernels from the committed rocprofv3 PMC passes
        # (FETCH_SIZE / WRITE_SIZE cannot be read live; see profiles/r01_gemm_traffic.json)
        traffic = None
        tf = os.path.join(ROOT, "profiles", "r01_gemm_traffic.json")
        if a.dtype == "bf16" and os.path.exists(tf):
            ks = json.load(open(tf))["kernels"].values()
            n = sum(k["launches_profiled"] for k in ks)
            traffic = sum(k["traffic_bytes_per_launch"] * k["launches_profiled"] for k in ks) / max(n, 1)
        achieved = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        peak = MFMA_PEAK_TFLOPS[a.dtype]
        out = {
            "metric": "images/sec (336^2, drop_iter=4, blur+CRF)",
            "value": world * B * a.steps / dt,
            "unit": "images/sec",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * dt / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": "Pascal-VOC-shaped: 336x336 RGB, 20-class prompt (L=25 tokens, K=21 channels), "
                                   "BLIP-ITM-large random weights, layer 8 head 9, drop_iter 4, threshold 0.15, "
                                   + ("N-drop" if a.skip_1drop else "1-drop + N-drop") + " blur+CRF",
                       "images_per_step_per_gpu": B, "sharding": "images across ranks, no per-step collective"},
            "roofline": {"bound": "mfma", "kernel": ("gemm_nt_wide_kernel (persistent 256x256 bf16 LDS-DMA ring GEMM, 32x32x16 MFMA; all non-small launches timed)" if a.dtype == "bf16" else "gemm_nt_big_kernel<float> (128x128 fp32 LDS-DMA ring GEMM, 16x16x4 MFMA)"), "achieved": achieved,
                         "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
                         "traffic_unit": "HBM-side bytes per launch (rocprofv3 PMC, offline pass)",
                         "launches": launches, "avg_launch_ms": ms / max(launches, 1),
                         "algorithmic_flop_per_launch": flops / max(launches, 1)},
        }
        if not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, 0, 1)
        print(json.dumps(out))
    e.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
