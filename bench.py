#!/usr/bin/env python
"""bench.py -- PnP-OVSS hot path on MI355X: images/sec at 336^2, drop_iter=4, blur+CRF.

One "step" = one pass of the whole hot path over one batch of synthetic images resident in HBM:
  4 x [ViT-L/16 + BERT/cross-attention forward, analytic dL/dP backward, GradCAM gather, salience
  drop] -> word-piece merge -> threshold + bilinear upsample (+ Scale_0_1) -> Gaussian blur ->
  DenseCRF (10 mean-field iterations, lattices rebuilt for the batch) -> argmax/remap/histogram,
  for BOTH branches the VOC driver runs (1-drop and N-drop: PnP_OVSS_0514_updated_segmentation.py:
  348-403 and :424-481), i.e. two blur+CRF passes per image like the reference.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 35] [--dtype bf16|f32]

Multi-GPU (`--gpus N`, N > 1): one process per GPU.  Launched under torch.distributed.run (RANK in the
environment) this process IS a rank; launched bare, this process only starts N rank processes of itself
(subprocess, never exec, never touching the GPU), forwards rank 0's JSON line and exits with the worst
child status -- the counterpart of the reference's mp.spawn (PnP.py:1439).  Images are sharded across
ranks (weak scaling: fixed per-GPU batch, no per-step collective), weights are broadcast from rank 0
over RCCL, the confusion histogram is all-reduced and label maps gathered to rank 0 after the timed
region.  `--backend gloo --dry-run` drives the same launcher and collectives on CPU tensors (tests).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "pnp-ovss_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

N_CLASSES = 20
IMG = 336
LAYER, HEAD, DROP_ITER, THRESH = 7, 9, 4, 0.15
NOISE = 4          # +-4 grey levels of per-pixel noise on the 8x8-block synthetic images (see synth.synth_images)
NOISE_HARD = 12    # second operating point: ~4x the bilateral lattice points per pixel
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3, "bf16x3": 2500.0}     # /opt/skills/guides/MI355X_MICROARCH.md (dense)
HBM_PEAK_GBS = 8000.0


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=35, help="images per step per GPU (--batch_size 35, Run_seg.sh)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32", "bf16x3"],
                    help="bf16: throughput mode (BASELINE config 2); f32: the reference's arithmetic; bf16x3: split-bf16, "
                         "fp32-class results on the bf16 MFMA")
    ap.add_argument("--crf-chunk", type=int, default=0)
    ap.add_argument("--noise", type=int, default=NOISE, help="per-pixel noise amplitude of the synthetic images")
    ap.add_argument("--skip-1drop", action="store_true", help="PnPc.py behaviour (COCO driver): N-drop branch only")
    ap.add_argument("--separate-crf", action="store_true", help="run the 1-drop and N-drop DenseCRF as two passes (default: paired)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-mode", action="store_true", help="skip the fp32 (parity mode) timing record")
    ap.add_argument("--no-noise12", action="store_true", help="skip the second (noise +-12) operating point")
    ap.add_argument("--parity-steps", type=int, default=2)
    ap.add_argument("--cpu-images", type=int, default=3, help="images of the cpu_baseline sample (after one warm-up image)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU work: exercise launcher + rendezvous + broadcast / all-reduce / gather on CPU tensors")
    ap.add_argument("--dry-run-fail-rank", type=int, default=-1, help=argparse.SUPPRESS)   # launcher test: this rank exits 3
    ap.add_argument("--share-gpu", action="store_true",
                    help="diagnostic: every rank uses cuda:0 (rehearses the N > 1 path on a one-GPU box with --backend gloo; "
                         "RCCL itself refuses two ranks on one device)")
    ap.add_argument("--no-events", action="store_true", help="diagnostic: no hipEvents around the GEMM / CRF launches (no roofline records)")
    ap.add_argument("--event-period", type=int, default=5,
                    help="bracket every n-th launch of the dense GEMM family with hipEvents (5 is coprime to the 4-GEMM layer "
                         "cycle: all shapes sampled equally; 1 = every launch, +2.3 %% step time)")
    ap.add_argument("--pipelines", type=int, default=3,
                    help="batches in flight per GPU: P engines, each with its own HIP stream and host thread, take the timed "
                         "steps round-robin (step = one 35-image batch through the whole path); the latency-bound text side and "
                         "kernel tails of one batch run beside the dense kernels of another.  1 = one batch at a time.  The "
                         "per-kernel roofline records always come from a one-batch-at-a-time pass of the same workload")
    ap.add_argument("--overlap", action="store_true",
                    help="software-pipeline batches over two HIP streams (drop loop of batch i+1 beside the post-process "
                         "of batch i); +2%% images/sec, off by default so the per-kernel event timing stays undisturbed")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------ launcher

def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, argv):
    """Start n rank processes of this script (LOCAL_RANK = RANK = GPU ordinal) and wait for them.  The parent never
    initialises the GPU and never execs; it forwards rank 0's stdout (the JSON line) and returns the worst status."""
    port = int(os.environ.get("MASTER_PORT", 0)) or _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out0 or "")
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        sys.stderr.write(f"bench.py: rank(s) failed: {bad}\n")
        return max(abs(c) for _, c in bad) or 1
    return 0


# ------------------------------------------------------------------------------------------ CPU baseline

def cpu_baseline(cfg, seed_w, n_images=3, noise=NOISE):
    """The oracle (numpy + C restatement of the reference path) timed on this box's host cores on a
    bounded sample of the same workload: one warm-up image, then `n_images` images, full path (4 drop
    iterations, both branches, blur + CRF).  Reported next to the GPU number; not the optimisation target."""
    from pnp_ovss import synth
    from oracle import pipeline_np as OP
    W = synth.synth_state_dict(cfg, seed_w)
    pieces1 = [[f"t{i}" for i in range(N_CLASSES)]]
    best1 = [list(range(N_CLASSES))]

    def run(n, seed):
        rgb, imgs = synth.synth_images(n, IMG, seed=seed, noise=noise)
        ids, mask = synth.synth_tokens(cfg, [N_CLASSES] * n, seed=seed)
        t0 = time.perf_counter()
        OP.segment_batch(W, cfg, imgs, ids, mask, pieces1 * n, best1 * n, list(rgb), [(IMG, IMG)] * n, data_type="voc",
                         drop_iter=DROP_ITER, layer=LAYER, head=HEAD, threshold=THRESH, mode="blur+crf")
        return time.perf_counter() - t0

    warm = run(1, 99)
    dt = run(n_images, 1234)
    return {"value": n_images / dt, "unit": "images/sec", "cores": os.cpu_count(), "kind": "port",
            "sample": f"{n_images} image(s) 336x336 in one batch after a 1-image warm-up ({warm:.1f} s), 20-class prompt, "
                      f"drop_iter=4, 1-drop + N-drop blur+CRF, numpy/OpenBLAS + gcc oracle, {dt:.1f} s"}


# ------------------------------------------------------------------------------------------ one rank

class Collectives:
    """The three exchanges of the path, all outside the steady state."""

    def __init__(self, distributed, rank, world):
        self.on, self.rank, self.world = distributed, rank, world

    def broadcast_weights(self, flat):
        import torch.distributed as dist
        if self.on:
            dist.broadcast(flat, src=0)                               # DDP ctor broadcast, PnP.py:1218

    def barrier(self):
        import torch.distributed as dist
        if self.on:
            dist.barrier()

    def max_time(self, dt, dev):
        import torch
        import torch.distributed as dist
        if not self.on:
            return dt
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def reduce_results(self, hists, labels):
        """all-reduce of the int64 confusion matrices (replaces the reference's .npy files, PnP.py:513-520 ->
        Calculate_mIoU.py:215-219) and gather of the uint8 label maps to rank 0 (mask-gather)."""
        import torch
        import torch.distributed as dist
        if not self.on:
            return [labels]
        for h in hists:
            dist.all_reduce(h)
        gathered = [torch.empty_like(labels) for _ in range(self.world)] if self.rank == 0 else None
        dist.gather(labels, gathered, dst=0)
        return gathered


def dry_run(a, coll, rank, world):
    """CPU stand-in for a rank: same rendezvous and the same three collectives on small CPU tensors."""
    import torch
    if rank == a.dry_run_fail_rank:
        os._exit(3)
    flat = torch.full((1000,), float(rank + 1))
    coll.broadcast_weights(flat)
    assert float(flat[0]) == 1.0 and float(flat[-1]) == 1.0, "weight broadcast did not deliver rank 0's buffer"
    hist = torch.zeros(21 * 21, dtype=torch.int64)
    hist[rank] = a.batch
    labels = torch.full((a.batch * 4,), rank, dtype=torch.uint8)
    coll.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * a.steps)
    dt = coll.max_time(time.perf_counter() - t0, torch.device("cpu"))
    gathered = coll.reduce_results([hist], labels)
    if rank == 0:
        assert int(hist.sum()) == world * a.batch
        assert [int(g[0]) for g in gathered] == list(range(world))
        import torch.distributed as dist
        print(json.dumps({"metric": "images/sec (336^2, drop_iter=4, blur+CRF)", "value": world * a.batch * a.steps / dt,
                          "unit": "images/sec", "n_gpus": world, "n_ranks": dist.get_world_size() if coll.on else 1,
                          "images_per_rank": a.batch * a.steps, "steps": a.steps, "warmup": a.warmup, "dry_run": True,
                          "backend": a.backend, "hist_total": int(hist.sum()), "gathered_ranks": [int(g[0]) for g in gathered]}))


def run_rank(a):
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = "RANK" in os.environ          # a rank of torch.distributed.run / of launch_ranks (also for N = 1)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(a.backend, rank=rank, world_size=world)
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}")
    coll = Collectives(distributed, rank, world)
    if a.dry_run:
        dry_run(a, coll, rank, world)
        if distributed:
            dist.destroy_process_group()
        return
    if a.share_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from pnp_ovss import config as C, synth
    from pnp_ovss.hip import Engine
    cfg = C.blip_itm_large(IMG)
    B = a.batch

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
    coll.broadcast_weights(flat)

    def make_engine(dtype):
        e = Engine(cfg, max_batch=B, max_text_len=32, stash_layer=LAYER, mode=dtype, device=local)
        sd, o = {}, 0
        for n, shp in shapes.items():
            k = int(np.prod(shp))
            sd[n] = flat[o:o + k].view(*shp)
            o += k
        e.load_state_dict(sd)
        e.post_reserve(B, B * IMG * IMG, IMG * IMG, N_CLASSES + 1, a.crf_chunk)
        return e

    sizes = [(IMG, IMG)] * B
    plans = [[([i], 1) for i in range(N_CLASSES)]] * B         # one word-piece per class
    luts = [list(range(N_CLASSES + 1))] * B                    # index i -> class id (background 0)
    ids, mask = synth.synth_tokens(cfg, [N_CLASSES] * B, seed=1234 + rank)
    L = int(mask.sum(1).max())
    d_ids, d_mask = torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev)
    gt = np.random.default_rng(rank).integers(0, 21, size=(B, IMG, IMG)).astype(np.float32)
    d_gt = torch.from_numpy(gt.reshape(-1)).to(dev)

    def timed_run_pipelined(engines, noise, steps, warmup):
        """`steps` timed passes (each the whole path over one 35-image batch resident in HBM), dealt round-robin to
        len(engines) pipelines: one engine + HIP stream + host thread each, `warmup` untimed passes per pipeline first.
        No per-kernel events (kernels of different pipelines interleave).  Returns (seconds [max over ranks], states)."""
        import threading
        rgb, imgs = synth.synth_images(B, IMG, seed=1234 + rank, noise=noise)
        d_img = torch.from_numpy(imgs).to(dev)
        d_rgb = torch.from_numpy(rgb.reshape(-1)).to(dev)
        P = len(engines)
        streams = [torch.cuda.Stream(device=dev) for _ in range(P)]
        states = [{"hist1": torch.zeros(21 * 21, device=dev, dtype=torch.int64),
                   "histn": torch.zeros(21 * 21, device=dev, dtype=torch.int64)} for _ in range(P)]
        errors = []

        def one_step(p):
            e, st = engines[p], states[p]
            g0, agg, _, _ = e.drop_loop(d_img, d_ids, d_mask, L, HEAD, DROP_ITER)
            e.post_prepare(sizes, plans, luts, [True] * B, rgb=d_rgb, gt=d_gt, want_crf=True)
            if a.skip_1drop:
                st["ln"] = e.postprocess(agg, THRESH, False, "blur+crf", 21, st["histn"])
            else:
                st["l1"], st["ln"] = e.postprocess_pair(g0, agg, THRESH, 21, st["hist1"], st["histn"])
            st["keep"] = (g0, agg)

        def worker(p, n_steps):
            try:
                torch.cuda.set_device(dev)
                with torch.cuda.stream(streams[p]):
                    for _ in range(n_steps):
                        one_step(p)
            except Exception as ex:          # noqa: BLE001 -- reported by the caller
                errors.append((p, repr(ex)))

        def run(counts):
            ths = [threading.Thread(target=worker, args=(p, counts[p])) for p in range(P) if counts[p] > 0]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
            if errors:
                raise RuntimeError(f"pipeline(s) failed: {errors}")

        torch.cuda.synchronize()
        run([warmup] * P)
        coll.barrier()
        torch.cuda.synchronize()
        for e in engines:
            e.profile_enable(False)
        t0 = time.perf_counter()
        run([steps // P + (1 if p < steps % P else 0) for p in range(P)])
        coll.barrier()
        torch.cuda.synchronize()
        dt = coll.max_time(time.perf_counter() - t0, dev)
        return dt, states

    def timed_run(e, noise, steps, warmup, overlap=False):
        """`warmup` untimed + `steps` timed passes over one synthetic batch resident in HBM.
        Returns (seconds [max over ranks], state, gemm profile, crf profile, lattice points per pixel)."""
        rgb, imgs = synth.synth_images(B, IMG, seed=1234 + rank, noise=noise)
        d_img = torch.from_numpy(imgs).to(dev)
        d_rgb = torch.from_numpy(rgb.reshape(-1)).to(dev)
        hist1 = torch.zeros(21 * 21, device=dev, dtype=torch.int64)
        histn = torch.zeros(21 * 21, device=dev, dtype=torch.int64)
        state = {"hist1": hist1, "histn": histn}
        # Two HIP streams (model / post-process).  With --overlap the drop loop of batch i+1 is enqueued
        # before the post-process of batch i; every batch still runs the complete path inside the timed region.
        if overlap:
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
            if not overlap:
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
            coll.barrier()
            torch.cuda.synchronize()

        run(warmup)
        sync()
        keep.clear()
        e.profile_enable(0 if a.no_events else max(1, a.event_period))
        t0 = time.perf_counter()
        run(steps)
        sync()
        dt = time.perf_counter() - t0
        gemm = e.profile_read_stage(0)
        crf = e.profile_read_stage(1)
        e.profile_enable(False)
        dt = coll.max_time(dt, dev)
        idb = e.buffer("crf_idbase_bilateral", torch.int32)[: B + 1].cpu().numpy()
        ppp = float(idb[B] - idb[0]) / float(B * IMG * IMG)
        return dt, state, gemm, crf, ppp

    def roofline_gemm(dtype, gemm):
        launches, flops, ms = gemm
        achieved = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        peak = MFMA_PEAK_TFLOPS[dtype]
        kern = {"bf16": "gemm_nt_wide_kernel (persistent 256x256 bf16 LDS-DMA ring GEMM, 32x32x16 MFMA; all launches with "
                        "M = B*N rows timed)",
                "f32": "gemm_nt_big_kernel<float> (128x128 fp32 LDS-DMA ring GEMM, 16x16x4 MFMA)",
                "bf16x3": "gemm_nt_wide_kernel<.., X3> (the bf16 kernel on (hi, lo) operand pairs: 3 MFMA passes per product; "
                          "achieved counts ALGORITHMIC flops 2MNK, so the ceiling of frac is 1/3)"}[dtype]
        return {"bound": "mfma", "kernel": kern,
                "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                "launches": launches, "sampling": f"every {max(1, a.event_period)}-th launch of the family bracketed by hipEvents",
                "avg_launch_ms": ms / max(launches, 1),
                "algorithmic_flop_per_launch": flops / max(launches, 1)}

    def roofline_crf(crf, steps, ppp):
        runs, nbytes, ms = crf
        achieved = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        return {"bound": "hbm", "kernel": "DenseCRF mean-field (crf_splat / crf_blur / crf_update, 10 iterations, "
                                          "both channel groups), bracketed by hipEvents per batch",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "ms_per_step": ms / max(steps, 1), "algorithmic_bytes_per_step": nbytes / max(steps, 1),
                "bilateral_lattice_points_per_pixel": ppp}

    e = make_engine(a.dtype)
    dt, state, gemm, crf, ppp = timed_run(e, a.noise, a.steps, a.warmup, a.overlap)
    seq = {"value": world * B * a.steps / dt, "ms_per_step": 1e3 * dt / a.steps}
    P = max(1, a.pipelines)
    engines = [e]
    if P > 1:
        # headline: P batches in flight (same K timed steps, same work per step); the pass above, one batch at a
        # time with events around the dense GEMM / mean-field launches, supplies the per-kernel records
        engines += [make_engine(a.dtype) for _ in range(P - 1)]
        dt, states = timed_run_pipelined(engines, a.noise, a.steps, a.warmup)
        for st in states[1:]:
            states[0]["histn"] += st["histn"]
            states[0]["hist1"] += st["hist1"]
        state = {k: states[0][k] for k in ("histn", "hist1")}
        state["ln"] = next(st["ln"] for st in states if "ln" in st)
    gathered = coll.reduce_results([state["histn"], state["hist1"]], state["ln"])

    if rank == 0:
        # HBM-side bytes per launch of the same kernels from the committed rocprofv3 PMC passes
        # (FETCH_SIZE / WRITE_SIZE cannot be read live; see profiles/*_gemm_traffic.json)
        traffic = None
        for tf in ("r02_gemm_traffic.json", "r01_gemm_traffic.json"):
            tf = os.path.join(ROOT, "profiles", tf)
            if a.dtype == "bf16" and os.path.exists(tf):
                ks = json.load(open(tf))["kernels"].values()
                n = sum(k["launches_profiled"] for k in ks)
                traffic = sum(k["traffic_bytes_per_launch"] * k["launches_profiled"] for k in ks) / max(n, 1)
                break
        roof = roofline_gemm(a.dtype, gemm)
        roof["traffic"] = traffic
        roof["traffic_unit"] = "HBM-side bytes per launch (rocprofv3 PMC, offline pass)"
        out = {
            "metric": "images/sec (336^2, drop_iter=4, blur+CRF)",
            "value": world * B * a.steps / dt,
            "unit": "images/sec",
            "n_gpus": world, "n_ranks": dist.get_world_size() if distributed else 1,
            "images_per_rank": B * a.steps, "gathered_label_maps": len(gathered),
            "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * dt / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": "Pascal-VOC-shaped: 336x336 RGB, 20-class prompt (L=25 tokens, K=21 channels), "
                                   "BLIP-ITM-large random weights, layer 8 head 9, drop_iter 4, threshold 0.15, "
                                   + ("N-drop" if a.skip_1drop else "1-drop + N-drop") + " blur+CRF",
                       "images_per_step_per_gpu": B, "sharding": "images across ranks, no per-step collective",
                       "image_noise": a.noise},
            "pipelines": {"batches_in_flight": P,
                          "note": "P engines / HIP streams / host threads take the timed steps round-robin; every step is "
                                  "the whole path over one 35-image batch",
                          "one_batch_at_a_time": seq},
            "roofline": roof,
            "crf": roofline_crf(crf, a.steps, ppp),
        }
        out["roofline"]["measured_in"] = out["crf"]["measured_in"] = (
            "one-batch-at-a-time pass of the same workload in this run (%d timed steps, %.1f ms per step)" % (a.steps, seq["ms_per_step"]))
        if world == 1 and not a.no_noise12 and a.noise != NOISE_HARD:
            n2 = max(1, min(a.steps, 2))
            dt2, _, _, crf2, ppp2 = timed_run(e, NOISE_HARD, n2, 1)
            out["noise12"] = {"value": B * n2 / dt2, "unit": "images/sec", "ms_per_step": 1e3 * dt2 / n2, "steps": n2,
                              "image_noise": NOISE_HARD, "batches_in_flight": 1, "crf": roofline_crf(crf2, n2, ppp2)}
            if P > 1:
                dt2p, _ = timed_run_pipelined(engines, NOISE_HARD, a.steps, 1)
                out["noise12"].update({"value": B * a.steps / dt2p, "ms_per_step": 1e3 * dt2p / a.steps, "steps": a.steps,
                                       "batches_in_flight": P, "one_batch_at_a_time": {"value": B * n2 / dt2, "ms_per_step": 1e3 * dt2 / n2}})
        for ex in engines[1:]:
            ex.close()
        engines = [e]
        if world == 1 and not a.no_parity_mode and a.dtype == "bf16":
            # the modes whose outputs meet north_star's tolerances against the reference's fp32 run (tests/test_hip_parity.py:
            # maps < 1e-4, identical patch picks): split-bf16 (fp32-class products on the bf16 MFMA) and exact fp32
            for pm in ("bf16x3", "f32"):
                e.close()
                del e
                torch.cuda.empty_cache()
                e = make_engine(pm)
                dt3, _, gemm3, crf3, _ = timed_run(e, a.noise, a.parity_steps, 1)
                rec = {"dtype": pm, "value": B * a.parity_steps / dt3, "unit": "images/sec",
                       "ms_per_step": 1e3 * dt3 / a.parity_steps, "steps": a.parity_steps, "warmup": 1, "batches_in_flight": 1,
                       "roofline": roofline_gemm(pm, gemm3), "crf_ms_per_step": crf3[2] / a.parity_steps}
                if pm == "bf16x3" and P > 1:
                    # the parity mode with the headline's batches in flight (per-kernel records: the pass above)
                    more = [make_engine(pm) for _ in range(P - 1)]
                    nst = max(a.parity_steps, P)
                    dt3p, _ = timed_run_pipelined([e] + more, a.noise, nst, 1)
                    for ex in more:
                        ex.close()
                    rec.update({"value": B * nst / dt3p, "ms_per_step": 1e3 * dt3p / nst, "steps": nst, "batches_in_flight": P,
                                "one_batch_at_a_time": {"value": B * a.parity_steps / dt3, "ms_per_step": 1e3 * dt3 / a.parity_steps}})
                if pm == "bf16x3":
                    out["parity_mode"] = rec
                else:
                    out["parity_mode"]["f32"] = rec
        e.close()
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, 0, a.cpu_images, a.noise)
        print(json.dumps(out))
        sys.stdout.flush()
    else:
        for ex in engines:
            ex.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


def main():
    a = parse_args()
    if a.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(a.gpus, sys.argv[1:]))
    run_rank(a)


if __name__ == "__main__":
    main()
