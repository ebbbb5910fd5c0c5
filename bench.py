#!/usr/bin/env python
"""bench.py -- PnP-OVSS hot path on MI355X: images/sec at 336^2, drop_iter=4, blur+CRF.

One "step" = one pass of the whole hot path over one batch of synthetic images resident in HBM:
  4 x [ViT-L/16 + BERT/cross-attention forward, analytic dL/dP backward, GradCAM gather, salience
  drop] -> word-piece merge -> threshold + bilinear upsample (+ Scale_0_1) -> Gaussian blur ->
  DenseCRF (10 mean-field iterations, lattices rebuilt for the batch) -> argmax/remap/histogram,
  for BOTH branches the VOC driver runs (1-drop and N-drop: PnP_OVSS_0514_updated_segmentation.py:
  348-403 and :424-481), i.e. two blur+CRF passes per image like the reference.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config voc|psc59|coco80|ade768] [--dtype bf16x3|f32|bf16]

The headline (`value`, `dtype`, `roofline`) is measured in a mode that reproduces the reference's fp32 results
(`bf16x3`, split-bf16: tests/test_hip_parity.py hold it to maps < 1e-4 and identical patch picks); plain `bf16` is
reported as a nested `throughput_mode` record together with how far its label maps are from the headline's.

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

LAYER, HEAD, DROP_ITER, THRESH = 7, 9, 4, 0.15
NOISE = 4          # +-4 grey levels of per-pixel noise on the 8x8-block synthetic images (see synth.synth_images)
NOISE_HARD = 12    # second operating point: ~4x the bilateral lattice points per pixel
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3, "bf16x3": 2500.0}     # /opt/skills/guides/MI355X_MICROARCH.md (dense)
HBM_PEAK_GBS = 8000.0
COCO_IDS = [i for i in range(1, 91) if i not in (12, 26, 29, 30, 45, 66, 68, 69, 71, 83)]   # the 80 category ids in 1..90

# Workloads (BASELINE.json configs 2-5, single-GPU shapes).  `voc` is the headline (the configuration the metric is quoted
# on); the others are records of the same path at the other datasets' shapes.  classes = words in the caption (one word-piece
# each: L = 1 + 3 + classes + 1 tokens), hist = rows of the confusion matrix, scale01 = Scale_0_1 on (1-drop, N-drop).
CONFIGS = {
    "voc": dict(img=336, classes=20, data_type="voc", hist=21, batch=35, skip_1drop=False, scale01=(True, False), crf_chunk=0, pipelines=3,
                what="Pascal-VOC-shaped (BASELINE config 2): 336x336 RGB, 20-class prompt (L=25 tokens, K=21 channels)"),
    "psc59": dict(img=336, classes=59, data_type="psc", hist=60, batch=35, skip_1drop=False, scale01=(True, False), crf_chunk=0, pipelines=3,
                  what="Pascal-Context-shaped (BASELINE config 3): 336x336, 59-class prompt (L=64, K=59, no background channel)"),
    "coco80": dict(img=336, classes=80, data_type="coco_object", hist=91, batch=35, skip_1drop=True, scale01=(True, True), crf_chunk=0,
                   pipelines=3,
                   what="COCO-Object-shaped (BASELINE config 4, one GPU's share): 336x336, batch 35, 80-class prompt (L=85, K=81), "
                        "COCO driver rules: N-drop branch only (drop_iter >= 3), Scale_0_1, category-id labels, 91-row histogram"),
    # batch 8 again (round 6).  Round 5 ran 7: 8 x 2305 = 18 440 token rows are 72.03 -> 73 row tiles of the persistent 256 x 256 GEMM,
    # i.e. 292 / 876 / 1168 tiles = 2 / 4 / 5 rounds on 256 CUs where 7 images pay exactly 1 / 3 / 4.  The stream-K tail of the
    # split-bf16 GEMM (csrc/gemm_x3.hip) now cuts fc2's 36-tile second round over all CUs (501 -> 382 us per launch); per image the
    # two batch sizes are level (14.65 / 14.73 images/s, tools/gemm_x3_streamk.py, profiles/r06_ade_batch.txt)
    "ade768": dict(img=768, classes=150, data_type="ade20k", hist=151, batch=8, skip_1drop=False, scale01=(True, False), crf_chunk=1,
                   pipelines=2,
                   what="ADE20K-shaped (BASELINE config 5, one GPU's share): 768x768 (2305 image tokens, pos-embed grid 48x48), "
                        "8 images per step, 150-class prompt (L=155, K=150), blur radius 154, DenseCRF one image per launch group"),
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="voc", choices=sorted(CONFIGS),
                    help="workload of the headline line (default voc = the configuration BASELINE.json's metric is quoted on)")
    ap.add_argument("--batch", type=int, default=0, help="images per step per GPU (0: the config's, 35 = --batch_size 35 of Run_seg.sh)")
    ap.add_argument("--dtype", default="bf16x3", choices=["bf16x3", "f32", "bf16"],
                    help="bf16x3 (default): split-bf16, fp32-class products on the bf16 MFMA -- reproduces the reference's fp32 "
                         "results (maps < 1e-4, identical patch picks); f32: exact fp32 MFMA; bf16: throughput mode, does NOT "
                         "reproduce the reference's patch picks (reported as the nested throughput_mode record)")
    ap.add_argument("--crf-chunk", type=int, default=-1, help="images per DenseCRF launch group (-1: the config's, 0 = whole batch)")
    ap.add_argument("--noise", type=int, default=NOISE, help="per-pixel noise amplitude of the synthetic images")
    ap.add_argument("--skip-1drop", action="store_true", help="PnPc.py behaviour (COCO driver): N-drop branch only")
    ap.add_argument("--separate-crf", action="store_true", help="run the 1-drop and N-drop DenseCRF as two passes (default: paired)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-modes", "--no-parity-mode", dest="no_other_modes", action="store_true",
                    help="skip the nested f32 (second parity mode) and bf16 (throughput mode) records")
    ap.add_argument("--no-noise12", action="store_true", help="skip the second (noise +-12) and third (photograph-like) operating points")
    ap.add_argument("--no-fixture-check", action="store_true",
                    help="skip the label comparison of the mode against the reference's fixtures (profiling runs: it adds small launches)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short records of the other BASELINE configs")
    ap.add_argument("--other-steps", type=int, default=3, help="timed steps of each other-config record (after one warm-up)")
    ap.add_argument("--cpu-images", type=int, default=20, help="images of the cpu_baseline sample, rounded up to a whole number per "
                    "worker process (SURVEY 8d: >= 20 after warm-ups)")
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
    ap.add_argument("--pipelines", type=int, default=-1,
                    help="batches in flight per GPU (-1: the config's, 3 for voc): P engines, each with its own HIP stream and host "
                         "thread, take the timed steps round-robin (step = one batch through the whole path); the latency-bound "
                         "text side and kernel tails of one batch run beside the dense kernels of another.  1 = one batch at a "
                         "time.  The per-kernel roofline records always come from a one-batch-at-a-time pass of the same workload")
    ap.add_argument("--overlap", action="store_true",
                    help="software-pipeline batches over two HIP streams (drop loop of batch i+1 beside the post-process "
                         "of batch i); off by default so the per-kernel event timing stays undisturbed")
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

def host_cpu_share():
    """Host cores this process may actually use: the cgroup CPU quota when one is set (a GPU box hands a one-GPU job a share
    of the host, not the cores os.cpu_count() lists), else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(cfg, seed_w, n_images=20, noise=NOISE, cores=None, W=None):
    """The oracle (numpy + C restatement of the reference path) timed on this box's host cores, IMAGE-PARALLEL (SURVEY.md 8d:
    all host cores, >= 20 images after warm-ups): a pool of worker processes (oracle/cpu_worker.py), two BLAS threads each,
    sharing one memory-mapped copy of the weights; every worker runs a warm-up image, then all start together and process
    their share of the images one at a time through the full path (4 drop iterations, 1-drop + N-drop branches, blur + CRF).
    images/sec = images / wall time from the common start to the last worker's end.  Reported next to the GPU number; not
    the optimisation target.  The in-container timing of the ACTUAL reference code (SURVEY.md 6 / BASELINE.md 2: it cannot
    travel to the GPU box) rides along as a labelled constant."""
    import shutil
    import tempfile
    from pnp_ovss import synth
    from oracle import cpu_worker as CW
    share = cores or host_cpu_share()
    threads = 2 if share >= 4 else 1
    workers = max(1, min(share // threads, 32))
    per = max(1, -(-n_images // workers))                     # images per worker; every worker gets the same count
    total = per * workers
    if W is None:
        W = synth.synth_state_dict(cfg, seed_w)
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    d = tempfile.mkdtemp(prefix="pnp_cpu_baseline_", dir=base)
    procs = []
    try:
        CW.save_weights(W, d)
        nc, img = CONFIGS["voc"]["classes"], CONFIGS["voc"]["img"]
        for i in range(workers):
            procs.append(subprocess.Popen([sys.executable, "-m", "oracle.cpu_worker", "--weights", d, "--images", str(per), "--seed",
                                           str(1234 + i), "--threads", str(threads), "--noise", str(noise), "--classes", str(nc),
                                           "--img", str(img)], cwd=ROOT, stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True))
        for p in procs:
            if p.stdout.readline().strip() != "ready":
                raise RuntimeError("cpu_baseline worker did not come up")
        t0 = time.perf_counter()
        for p in procs:
            p.stdin.write("go\n")
            p.stdin.flush()
        outs = [json.loads(p.stdout.readline()) for p in procs]
        dt = time.perf_counter() - t0
        for p in procs:
            p.wait()
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        shutil.rmtree(d, ignore_errors=True)
    per_img = [t for o in outs for t in o["seconds"]]
    warm = [o["warmup_seconds"] for o in outs]
    return {"value": total / dt, "unit": "images/sec", "cores": workers * threads, "kind": "port",
            "workers": workers, "threads_per_worker": threads, "host_cores_available": share, "host_cores_listed": os.cpu_count(),
            "sample": f"{total} images 336x336 ({per} per worker, one at a time) on {workers} worker processes x {threads} BLAS threads "
                      f"after one warm-up image per worker (drop_iter 1, {min(warm):.1f}-{max(warm):.1f} s), 20-class prompt, drop_iter=4, "
                      f"1-drop + N-drop blur+CRF, numpy/OpenBLAS + gcc oracle: {dt:.1f} s wall from the common start to the last "
                      f"worker's end; {min(per_img):.1f}-{max(per_img):.1f} s per image inside a worker",
            "reference_in_container": {
                "value": 1.0 / (4 * 9.07 + 0.38 + 1.23), "unit": "images/sec", "cores": 8, "kind": "reference",
                "machine": "build container (8 host cores, torch 2.10 CPU), NOT this box",
                "sample": "the reference's own compute_gradcam_ensemble (forward + full autograd backward + 144 gathers) 9.07 s per "
                          "call x 4 drop iterations + threshold/upsample 0.38 s + scipy blur 1.23 s, one 336x336 image, 25-token "
                          "caption; DenseCRF not included (pydensecrf cannot be installed there): BASELINE.md section 2"}}


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

    def ranks_seen(self, dev, rank):
        """Every rank contributes its id to one all-gather: the line then shows which ranks the communicator really joined."""
        import torch
        import torch.distributed as dist
        if not self.on:
            return [rank]
        mine = torch.tensor([rank], device=dev, dtype=torch.int64)
        allr = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(allr, mine)
        return sorted(int(x.item()) for x in allr)

    def gather_rates(self, rate, dev):
        """Each rank's own images/s (own clock) on every rank: a scaling run is self-checking."""
        import torch
        import torch.distributed as dist
        if not self.on:
            return [float(rate)]
        mine = torch.tensor([rate], device=dev, dtype=torch.float64)
        allr = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(allr, mine)
        return [float(x.item()) for x in allr]

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


# measured on MI355X (round 5, `pipelines.engine_device_bytes`): the first engine of a workload owns the converted weights, the others share them
ENGINE_GIB = {"voc": (27.2, 25.4), "psc59": (51.0, 49.2), "coco80": (66.8, 65.0), "ade768": (41.5, 39.7)}
HBM_GIB = 288 * 1e9 / 2 ** 30


def resource_budget(world, name, P, host_cores=None):
    """What N ranks x P engines ask of one node, checked BEFORE anything is allocated (the driver's 8-GPU run is the first time
    eight ranks meet; it must not die of plumbing): device memory per rank (flat fp32 weights + P engines) against one GPU's HBM,
    host threads (per rank: main + P step threads) against the cores the node grants, host memory per rank (rank 0 synthesises
    the weights one tensor at a time; the cpu_baseline / other-mode / other-config legs run at N = 1 only)."""
    first, more = ENGINE_GIB[name]
    dev_gib = 1.78e9 / 2 ** 30 + first + more * (P - 1)
    cores = host_cores or len(os.sched_getaffinity(0))
    b = {"ranks": world, "engines_per_rank": P, "device_GiB_per_rank": round(dev_gib, 1), "device_GiB_available": round(HBM_GIB, 1),
         "host_threads_total": world * (1 + P), "host_cores_granted": cores,
         "host_GiB_per_rank": 4.0, "legs_run_at_n_gt_1": "headline record only (one-batch-at-a-time pass + P batches in flight)"}
    assert dev_gib < 0.9 * HBM_GIB, f"{P} engines of {name} need {dev_gib:.0f} GiB per GPU"
    b["host_threads_oversubscribed"] = b["host_threads_total"] > cores       # reported, not fatal: step threads mostly wait on the GPU
    return b


def dry_run(a, coll, rank, world):
    """CPU stand-in for a rank: same rendezvous and the same three collectives on small CPU tensors."""
    import torch
    if rank == a.dry_run_fail_rank:
        os._exit(3)
    batch = a.batch or CONFIGS[a.config]["batch"]
    flat = torch.full((1000,), float(rank + 1))
    coll.broadcast_weights(flat)
    assert float(flat[0]) == 1.0 and float(flat[-1]) == 1.0, "weight broadcast did not deliver rank 0's buffer"
    hist = torch.zeros(21 * 21, dtype=torch.int64)
    hist[rank] = batch
    labels = torch.full((batch * 4,), rank, dtype=torch.uint8)
    coll.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * a.steps)
    dt = coll.max_time(time.perf_counter() - t0, torch.device("cpu"))
    gathered = coll.reduce_results([hist], labels)
    seen = coll.ranks_seen(torch.device("cpu"), rank)
    per_rank = coll.gather_rates(batch * a.steps / dt, torch.device("cpu"))
    if rank == 0:
        assert int(hist.sum()) == world * batch
        assert seen == list(range(world)) and len(per_rank) == world
        P = max(1, CONFIGS[a.config]["pipelines"] if a.pipelines < 0 else a.pipelines)
        assert [int(g[0]) for g in gathered] == list(range(world))
        import torch.distributed as dist
        print(json.dumps({"metric": "images/sec (336^2, drop_iter=4, blur+CRF)", "value": world * batch * a.steps / dt,
                          "unit": "images/sec", "n_gpus": world, "n_ranks": dist.get_world_size() if coll.on else 1,
                          "images_per_rank": batch * a.steps, "steps": a.steps, "warmup": a.warmup, "dry_run": True,
                          "backend": a.backend, "hist_total": int(hist.sum()), "gathered_ranks": [int(g[0]) for g in gathered],
                          "rccl_ranks_seen": seen, "per_rank_images_per_sec": per_rank,
                          "budget": resource_budget(world, a.config, P)}))


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

    from pnp_ovss import config as C, host, synth
    from pnp_ovss.hip import Engine

    # ---- weights: rank 0 materialises the seeded weights, RCCL broadcast over xGMI to the others (DDP ctor, PnP.py:1218).
    # pos_embed is generated per geometry (336: 21x21 grid, 768: 48x48); everything else is shared by the configs.
    flats = {}

    def weights_for(cfg):
        key = cfg.img_size
        if key in flats:
            return flats[key]
        shapes = synth.param_shapes(cfg)
        total = sum(int(np.prod(s)) for s in shapes.values())
        flat = torch.empty(total, device=dev, dtype=torch.float32)
        if rank == 0:
            o = 0
            for n, shp in shapes.items():
                w = synth.synth_tensor(n, shp, 0)
                flat[o:o + w.size].copy_(torch.from_numpy(w.reshape(-1)))
                o += w.size
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        coll.broadcast_weights(flat)
        torch.cuda.synchronize()
        flats[key] = (flat, shapes, {"weight_broadcast_ms": 1e3 * (time.perf_counter() - t0), "weight_bytes": total * 4})
        return flats[key]

    class Workload:
        """One config's engines, inputs and timed passes."""

        def __init__(self, name, dtype, batch=0, crf_chunk=-1, skip_1drop=None):
            w = CONFIGS[name]
            self.name, self.w, self.dtype = name, w, dtype
            self.img, self.nc, self.nh = w["img"], w["classes"], w["hist"]
            self.B = batch or w["batch"]
            self.chunk = w["crf_chunk"] if crf_chunk < 0 else crf_chunk
            self.skip_1drop = w["skip_1drop"] if skip_1drop is None else skip_1drop
            self.cfg = C.blip_itm_large(self.img)
            B, img, nc = self.B, self.img, self.nc
            self.bg = host.has_background(w["data_type"], nc)
            self.K = nc + int(self.bg)
            self.sizes = [(img, img)] * B
            self.plans = [[([i], 1) for i in range(nc)]] * B            # one word-piece per class
            ids_tab = COCO_IDS if w["data_type"].startswith("coco") else None
            self.luts = [host.remap_lut(list(range(nc)), self.bg, self.K, ids_tab)] * B
            ids, mask = synth.synth_tokens(self.cfg, [nc] * B, seed=1234 + rank)
            self.L = int(mask.sum(1).max())
            self.d_ids, self.d_mask = torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev)
            gt = np.random.default_rng(rank).integers(0, self.nh, size=(B, img, img)).astype(np.float32)
            self.d_gt = torch.from_numpy(gt.reshape(-1)).to(dev)
            self.engines = []
            self._inputs = {}

        def make_engine(self):
            flat, shapes, _ = weights_for(self.cfg)
            # the engines of a workload (batches in flight) run on ONE converted weight copy: the first one loads it, the
            # others are created on it (pnp_create_shared: activations + workspace of their own)
            e = Engine(self.cfg, max_batch=self.B, max_text_len=max(32, (self.L + 7) // 8 * 8), stash_layer=LAYER, mode=self.dtype,
                       device=local, share_weights_with=self.engines[0] if self.engines else None)
            if not self.engines:
                sd, o = {}, 0
                for n, shp in shapes.items():
                    k = int(np.prod(shp))
                    sd[n] = flat[o:o + k].view(*shp)
                    o += k
                e.load_state_dict(sd)
            pb = self.B if self.B <= 64 else self.w["batch"]             # images per prepared post-process batch
            e.post_reserve(pb, pb * self.img * self.img, self.img * self.img, self.K, self.chunk)
            self.engines.append(e)
            return e

        def close(self):
            for e in self.engines:
                e.close()
            self.engines = []
            self._inputs = {}

        def inputs(self, noise):
            """noise: +-grey levels of the block images, or "photo": the photograph-like generator (synth.synth_photo_images)."""
            if noise not in self._inputs:
                if noise == "photo":
                    rgb, imgs = synth.synth_photo_images(self.B, self.img, seed=1234 + rank)
                else:
                    rgb, imgs = synth.synth_images(self.B, self.img, seed=1234 + rank, noise=noise)
                self._inputs[noise] = (torch.from_numpy(imgs).to(dev), torch.from_numpy(rgb.reshape(-1)).to(dev))
            return self._inputs[noise]

        def one_step(self, e, st, d_img, d_rgb, separate=False):
            """The whole path over one batch: drop loop -> prepare (lattices of the batch) -> post-process of the branch(es)."""
            g0, agg, _, _ = e.drop_loop(d_img, self.d_ids, self.d_mask, self.L, HEAD, DROP_ITER)
            if self.B > 64:
                # --batch above what one prepared post-process batch holds (64 images: the image index has 6 bits in the lattice
                # keys): the model runs on all B images, the post-processing in groups of the config's batch (35)
                G = self.w["batch"]
                px = self.img * self.img
                keep1, keepn = [], []
                for a0 in range(0, self.B, G):
                    a1 = min(a0 + G, self.B)
                    e.post_prepare(self.sizes[a0:a1], self.plans[a0:a1], self.luts[a0:a1], [self.bg] * (a1 - a0),
                                   rgb=d_rgb[a0 * px * 3: a1 * px * 3], gt=self.d_gt[a0 * px: a1 * px], want_crf=True)
                    if self.skip_1drop:
                        keepn.append(e.postprocess(agg[a0:a1], THRESH, self.w["scale01"][1], "blur+crf", self.nh, st["histn"]).clone())
                    else:
                        l1, ln = e.postprocess_pair(g0[a0:a1], agg[a0:a1], THRESH, self.nh, st["hist1"], st["histn"], self.w["scale01"])
                        keep1.append(l1.clone())
                        keepn.append(ln.clone())
                st["ln"] = torch.cat(keepn)
                if keep1:
                    st["l1"] = torch.cat(keep1)
                st["keep"] = (g0, agg)
                return
            e.post_prepare(self.sizes, self.plans, self.luts, [self.bg] * self.B, rgb=d_rgb, gt=self.d_gt, want_crf=True)
            s01 = self.w["scale01"]
            if self.skip_1drop:
                st["ln"] = e.postprocess(agg, THRESH, s01[1], "blur+crf", self.nh, st["histn"])
            elif separate:
                st["l1"] = e.postprocess(g0, THRESH, s01[0], "blur+crf", self.nh, st["hist1"])
                st["ln"] = e.postprocess(agg, THRESH, s01[1], "blur+crf", self.nh, st["histn"])
            else:           # both branches in one DenseCRF run (two channel groups per row; identical results)
                st["l1"], st["ln"] = e.postprocess_pair(g0, agg, THRESH, self.nh, st["hist1"], st["histn"], s01)
            st["keep"] = (g0, agg)

        def new_state(self):
            return {"hist1": torch.zeros(self.nh * self.nh, device=dev, dtype=torch.int64),
                    "histn": torch.zeros(self.nh * self.nh, device=dev, dtype=torch.int64)}

        def timed_run_pipelined(self, P, noise, steps, warmup):
            """`steps` timed passes (each the whole path over one batch resident in HBM), dealt round-robin to P pipelines:
            one engine + HIP stream + host thread each, `warmup` untimed passes per pipeline first.  No per-kernel events
            (kernels of different pipelines interleave).  Returns (seconds [max over ranks], states)."""
            import threading
            while len(self.engines) < P:
                self.make_engine()
            engines = self.engines[:P]
            d_img, d_rgb = self.inputs(noise)
            streams = [torch.cuda.Stream(device=dev) for _ in range(P)]
            states = [self.new_state() for _ in range(P)]
            errors = []

            def worker(p, n_steps):
                try:
                    torch.cuda.set_device(dev)
                    with torch.cuda.stream(streams[p]):
                        for _ in range(n_steps):
                            self.one_step(engines[p], states[p], d_img, d_rgb)
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

            for e in engines:
                e.profile_enable(False)          # the event ring is not for interleaved streams
            torch.cuda.synchronize()
            run([warmup] * P)
            coll.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run([steps // P + (1 if p < steps % P else 0) for p in range(P)])
            coll.barrier()
            torch.cuda.synchronize()
            self.local_dt = time.perf_counter() - t0      # this rank's own time (the line's value uses the max over ranks)
            dt = coll.max_time(self.local_dt, dev)
            return dt, states

        def timed_run(self, noise, steps, warmup, overlap=False, separate=False):
            """`warmup` untimed + `steps` timed passes, one batch at a time, hipEvents around the dense GEMM launches and the
            mean-field.  Returns (seconds [max over ranks], state, gemm profile, crf profile, lattice points per pixel)."""
            e = self.engines[0] if self.engines else self.make_engine()
            d_img, d_rgb = self.inputs(noise)
            state = self.new_state()
            if overlap:
                s_model, s_post = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)

            def run(n):
                if not overlap:
                    for _ in range(n):
                        self.one_step(e, state, d_img, d_rgb, separate)
                    return
                # two HIP streams: the drop loop of batch i+1 is enqueued before the post-process of batch i
                def model_part():
                    with torch.cuda.stream(s_model):
                        g0, agg, _, _ = e.drop_loop(d_img, self.d_ids, self.d_mask, self.L, HEAD, DROP_ITER)
                        ev = torch.cuda.Event()
                        ev.record(s_model)
                    return g0, agg, ev

                def post_part(g0, agg, ev):
                    with torch.cuda.stream(s_post):
                        s_post.wait_event(ev)
                        e.post_prepare(self.sizes, self.plans, self.luts, [self.bg] * self.B, rgb=d_rgb, gt=self.d_gt, want_crf=True)
                        if self.skip_1drop:
                            state["ln"] = e.postprocess(agg, THRESH, self.w["scale01"][1], "blur+crf", self.nh, state["histn"])
                        else:
                            state["l1"], state["ln"] = e.postprocess_pair(g0, agg, THRESH, self.nh, state["hist1"], state["histn"],
                                                                          self.w["scale01"])
                    state.setdefault("keepall", []).append((g0, agg))
                nxt = model_part()
                for i in range(n):
                    cur = nxt
                    if i + 1 < n:
                        nxt = model_part()
                    post_part(*cur)

            def sync():
                coll.barrier()
                torch.cuda.synchronize()

            run(warmup)
            sync()
            e.profile_enable(0 if a.no_events else max(1, a.event_period))
            t0 = time.perf_counter()
            run(steps)
            sync()
            dt = time.perf_counter() - t0
            self.local_dt = dt
            gemm = e.profile_read_stage(0)
            crf = e.profile_read_stage(1) + (e.profile_read_stage(2)[1],)      # (runs, SURVEY 8d bytes, ms, lattice-term bytes)
            e.profile_enable(False)
            dt = coll.max_time(dt, dev)
            pb = self.B if self.B <= 64 else (self.B - 1) % self.w["batch"] + 1          # images of the last prepared group
            idb = e.buffer("crf_idbase_bilateral", torch.int32)[: pb + 1].cpu().numpy()
            ppp = float(idb[pb] - idb[0]) / float(pb * self.img * self.img)
            return dt, state, gemm, crf, ppp

    def roofline_gemm(dtype, gemm):
        launches, flops, ms = gemm
        achieved = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        peak = MFMA_PEAK_TFLOPS[dtype]
        kern = {"bf16": "gemm_nt_wide_kernel (persistent 256x256 bf16 LDS-DMA ring GEMM, 32x32x16 MFMA; all launches with "
                        "M = B*N rows timed)",
                "f32": "gemm_nt_big_kernel<float> (128x128 fp32 LDS-DMA ring GEMM, 16x16x4 MFMA)",
                "bf16x3": "gemm_nt_x3_kernel (persistent 256x256 split-bf16 GEMM: one K sweep over (hi, lo) operand pairs, three "
                          "v_mfma_f32_16x16x32_bf16 per fragment pair; `achieved` counts ALGORITHMIC flops 2MNK, the MFMA "
                          "work issued is 3x that, so the ceiling of frac is 1/3; issued_frac = 3 x frac)"}[dtype]
        out = {"bound": "mfma", "kernel": kern,
               "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
               "launches": launches, "sampling": f"every {max(1, a.event_period)}-th launch of the family bracketed by hipEvents",
               "excluded": "the text-side Linears (M = B*L rows: gemm_nt_small_x3_kernel / 64x64 tiles) are not in this record",
               "avg_launch_ms": ms / max(launches, 1),
               "algorithmic_flop_per_launch": flops / max(launches, 1)}
        if dtype == "bf16x3":
            out["issued_frac"] = 3 * achieved / peak
        return out

    def crf_counters(name, noise):
        """Fabric-side bytes and L2 requests per step of the mean-field kernels (the ones inside the event bracket: splat, lattice
        blur, update) from the committed rocprofv3 PMC passes of the same workload (profiles/: FETCH_SIZE doubled + WRITE_SIZE per
        the guide's gfx950 correction; TCC_HIT + TCC_MISS).  Only the configs' own image noise has such a pass."""
        if noise != NOISE:
            return None
        inside = lambda k: ("crf_splat_kernel" in k) or ("crf_blur4" in k) or ("crf_update" in k)
        for tag in ("r06", "r05"):
            tf = os.path.join(ROOT, "profiles", f"{tag}_hbm_traffic.json" if name == "voc" else f"{tag}_{name}_crf_traffic.json")
            if not os.path.exists(tf):
                continue
            d = json.load(open(tf))
            nsteps = d.get("steps_profiled", 2)                      # bench.py --steps 1 --warmup 1 under the profiler
            ks = {k: v for k, v in d["kernels"].items() if inside(k)}
            out = {"source": os.path.relpath(tf, ROOT),
                   "fabric_bytes_per_step": sum(v["traffic_bytes_per_launch"] * v["launches_profiled"] for v in ks.values()) / nsteps}
            hf = os.path.join(ROOT, "profiles", f"{tag}_crf_l2_hit.json" if name == "voc" else f"{tag}_{name}_crf_l2_hit.json")
            if os.path.exists(hf):
                h = json.load(open(hf))
                hk = {k: v for k, v in h.items() if isinstance(v, dict) and inside(k)}
                out["l2_hit_rate"] = {k.replace("void ", ""): round(v["l2_hit_rate"], 3) for k, v in hk.items()}
                if hk and all("tcc_req_per_launch" in v for v in hk.values()):
                    out["l2_requests_per_step"] = sum(v["tcc_req_per_launch"] * v["launches_profiled"] for v in hk.values()) / h.get("steps_profiled", 2)
                    out["l2_source"] = os.path.relpath(hf, ROOT)
            return out
        return None

    def roofline_crf(crf, steps, ppp, name=None, noise=None):
        runs, nbytes, ms, lattice = crf
        sec = ms * 1e-3
        achieved = nbytes / sec / 1e9 if ms > 0 else 0.0
        rec = {"bound": "hbm", "kernel": "DenseCRF mean-field (crf_splat / crf_blur / crf_update, 10 iterations, "
                                         "all channel groups), bracketed by hipEvents per batch",
               "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
               "ms_per_step": ms / max(steps, 1), "algorithmic_bytes_per_step": nbytes / max(steps, 1),
               "byte_model": "SURVEY 8d and nothing else: per mean-field iteration (2*9 + 2) * K*H*W*4 (splat + slice of 3 + 6 simplex "
                             "vertices, Q read and written)",
               "lattice_term": {"bytes_per_step": lattice / max(steps, 1),
                                "what": "NOT in frac: 2 * (2*M_gauss + 3*M_bilateral) * K*4 per iteration, the lattice value arrays read "
                                        "and written once per pass of two blur axes",
                                "frac_with_it": (nbytes + lattice) / sec / 1e9 / HBM_PEAK_GBS if ms > 0 else 0.0},
               "bilateral_lattice_points_per_pixel": ppp,
               "what_bounds_it": "neither memory bandwidth: counter_frac (bytes that actually crossed the fabric: part of the algorithmic "
                                 "bytes is served by L2, hit rates below) sits at or below frac and far from the 0.79 (6.3 of 8 TB/s) the guide "
                                 "calls achievable, and the L2 request rate (l2_gather) is about a quarter of the 34 TB/s the L2s deliver.  The "
                                 "kernels are dependent-load chains (contributor list -> Q rows; neighbour record -> nine value rows) and run at "
                                 "gather latency x the occupancy their registers allow (DESIGN.md section 3)"}
        cn = crf_counters(name, noise) if name else None
        if cn and ms > 0:
            per_step_s = sec / max(steps, 1)
            rec["counter_frac"] = cn["fabric_bytes_per_step"] / per_step_s / 1e9 / HBM_PEAK_GBS
            rec["counter_bytes_per_step"] = cn["fabric_bytes_per_step"]
            rec["counter_source"] = cn["source"] + " (offline PMC pass of the same workload: FETCH_SIZE x 2 + WRITE_SIZE) over this run's live time"
            if "l2_hit_rate" in cn:
                rec["l2_hit_rate"] = cn["l2_hit_rate"]
            if "l2_requests_per_step" in cn:
                rec["l2_gather"] = {"requests_per_sec": cn["l2_requests_per_step"] / per_step_s,
                                    "GBps_at_128B_per_request": cn["l2_requests_per_step"] * 128 / per_step_s / 1e9,
                                    "source": cn["l2_source"] + " (TCC_HIT_sum + TCC_MISS_sum; request width uncalibrated, 128-byte lines assumed)"}
        else:
            rec["counter_frac"] = None
        return rec

    def traffic_for(dtype):
        """HBM-side bytes per launch of the dense GEMM family from the committed rocprofv3 PMC passes (FETCH_SIZE /
        WRITE_SIZE cannot be read live; see profiles/*_gemm_traffic*.json)."""
        names = {"bf16": ("r05_gemm_traffic.json", "r04_gemm_traffic.json", "r03_gemm_traffic.json", "r02_gemm_traffic.json", "r01_gemm_traffic.json"),
                 "bf16x3": ("r05_gemm_traffic_bf16x3.json", "r04_gemm_traffic_bf16x3.json", "r03_gemm_traffic_bf16x3.json")}.get(dtype, ())
        for tf in names:
            tf = os.path.join(ROOT, "profiles", tf)
            if os.path.exists(tf):
                ks = json.load(open(tf))["kernels"].values()
                n = sum(k["launches_profiled"] for k in ks)
                return sum(k["traffic_bytes_per_launch"] * k["launches_profiled"] for k in ks) / max(n, 1)
        return None

    def record(wl, P, noise, steps, warmup, headline=False):
        """One workload in one compute mode: a one-batch-at-a-time pass with the per-kernel records, then (P > 1) the same
        steps with P batches in flight.  Returns (line fields, final state)."""
        dt, state, gemm, crf, ppp = wl.timed_run(noise, steps, warmup, a.overlap if headline else False, a.separate_crf if headline else False)
        seq = {"value": world * wl.B * steps / dt, "ms_per_step": 1e3 * dt / steps}
        if P > 1:
            dt, states = wl.timed_run_pipelined(P, noise, steps, warmup)
            for st in states[1:]:
                states[0]["histn"] += st["histn"]
                states[0]["hist1"] += st["hist1"]
            state = {k: states[0][k] for k in ("histn", "hist1")}
            state["ln"] = next(st["ln"] for st in states if "ln" in st)
        roof = roofline_gemm(wl.dtype, gemm)
        rec = {"value": world * wl.B * steps / dt, "unit": "images/sec", "ms_per_step": 1e3 * dt / steps, "steps": steps, "warmup": warmup,
               "dtype": wl.dtype,
               "pipelines": {"batches_in_flight": P,
                             "note": "P engines / HIP streams / host threads take the timed steps round-robin; every step is "
                                     "the whole path over one batch; the engines share one weight copy", "one_batch_at_a_time": seq,
                             "engine_device_bytes": [int(e.allocated_bytes()) for e in wl.engines[:max(P, 1)]]},
               "roofline": roof, "crf": roofline_crf(crf, steps, ppp, wl.name, noise)}
        if wl.dtype == "bf16x3":
            # the stream-K tail of the split-bf16 GEMM (csrc/gemm_x3.hip): launches that took it, and the give-up word of its bounded
            # spins -- an owner that gave up waiting for a partial tile produced garbage, so anything but 0 fails the run loudly
            sk = [e.streamk_status() for e in wl.engines[:max(P, 1)]]
            rec["streamk"] = {"launches": int(sum(n for n, _ in sk)), "gave_up": int(max(t for _, t in sk))}
            if rec["streamk"]["gave_up"]:
                raise SystemExit(f"bench.py: a stream-K owner gave up waiting for a partial tile (word {rec['streamk']['gave_up']}): results invalid")
        rec["roofline"]["measured_in"] = rec["crf"]["measured_in"] = (
            "one-batch-at-a-time pass of the same workload in this run (%d timed steps, %.1f ms per step)" % (steps, seq["ms_per_step"]))
        return rec, state

    def workload_cfg(wl, noise):
        return {"workload": wl.w["what"] + ", BLIP-ITM-large random weights, layer 8 head 9, drop_iter 4, threshold 0.15, "
                            + ("N-drop" if wl.skip_1drop else "1-drop + N-drop") + " blur+CRF",
                "name": wl.name, "images_per_step_per_gpu": wl.B, "sharding": "images across ranks, no per-step collective",
                "image_noise": noise, "crf_launch_group_images": wl.chunk or wl.B}

    wc = CONFIGS[a.config]
    P = max(1, wc["pipelines"] if a.pipelines < 0 else a.pipelines)
    budget = resource_budget(world, a.config, P)
    free_b, total_b = torch.cuda.mem_get_info(dev)
    budget["device_GiB_free_at_start"] = round(free_b / 2 ** 30, 1)
    if free_b / 2 ** 30 < budget["device_GiB_per_rank"] and not a.share_gpu:
        raise SystemExit(f"bench.py: rank {rank} sees {free_b / 2 ** 30:.0f} GiB free on cuda:{local}, {budget['device_GiB_per_rank']} needed for {P} engines")
    wl = Workload(a.config, a.dtype, a.batch, a.crf_chunk, True if a.skip_1drop else None)
    head, state = record(wl, P, a.noise, a.steps, a.warmup, headline=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gathered = coll.reduce_results([state["histn"], state["hist1"]], state["ln"])
    torch.cuda.synchronize()
    t_reduce = time.perf_counter() - t0
    per_rank = coll.gather_rates(wl.B * a.steps / wl.local_dt, dev) if distributed else None     # own clock, not the max over ranks
    seen = coll.ranks_seen(dev, rank)

    if rank == 0:
        head["roofline"]["traffic"] = traffic_for(a.dtype)
        head["roofline"]["traffic_unit"] = "HBM-side bytes per launch (rocprofv3 PMC, offline pass; profiles/)"
        out = {
            "metric": "images/sec (336^2, drop_iter=4, blur+CRF)" if wl.img == 336 else f"images/sec ({wl.img}^2, drop_iter=4, blur+CRF)",
            "value": head["value"], "unit": "images/sec",
            "n_gpus": world, "n_ranks": dist.get_world_size() if distributed else 1,
            "rccl_ranks_seen": seen, "budget": budget,
            "images_per_rank": wl.B * a.steps, "gathered_label_maps": len(gathered),
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": head["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": a.dtype, "data": "synthetic",
            "parity": {"bf16x3": "reproduces the reference's fp32 run: maps < 1e-4, identical patch picks (tests/test_hip_parity.py)",
                       "f32": "the reference's arithmetic (exact fp32 MFMA)",
                       "bf16": "does NOT reproduce the reference's patch picks"}[a.dtype],
            "config": workload_cfg(wl, a.noise),
            "pipelines": head["pipelines"], "roofline": head["roofline"], "crf": head["crf"], "streamk": head.get("streamk"),
            "collectives": dict(weights_for(wl.cfg)[2], hist_allreduce_and_label_gather_ms=1e3 * t_reduce,
                                per_rank_images_per_sec=per_rank,
                                hist_ndrop_total=int(state["histn"].sum().item()),       # all-reduced: pixels counted by ALL ranks
                                gathered_label_bytes=[int(g.numel()) for g in gathered],
                                note="outside the timed region (start-up broadcast, end-of-run reduce); RCCL when n_ranks > 1"),
        }
        single = world == 1
        if a.dtype != "bf16" and not a.no_fixture_check:
            # the benchmarked mode against the REFERENCE's own label maps (committed fixtures, small geometry): fraction of
            # label pixels that differ, both branches, blur and no post-process (asserted with the near-tie rule in
            # tests/test_hip_parity.py::test_end_to_end_labels_vs_reference_run)
            from pnp_ovss import selfcheck
            fl = selfcheck.fixture_label_flips(a.dtype, os.path.join(ROOT, "tests", "golden"), device=local)
            out["label_pixels_differing_from_reference_fixtures"] = {
                "frac": fl["frac"], "differing": fl["differing"], "pixels": fl["pixels"], "per_fixture": fl["per_fixture"],
                "what": "pipeline_voc.npz + pipeline_psc.npz (small geometry) + pipeline_voc_large.npz (the HEADLINE geometry: BLIP-ITM-large "
                        "336^2, 20-class prompt): the reference's save_img_union_attention label maps (1-drop and N-drop, blur and none) "
                        "vs this mode's, same inputs; differences sit at float near-ties of the two best channels"}
        if single and not a.no_noise12 and a.noise != NOISE_HARD:
            n2 = max(1, min(a.steps, 2))
            dt2, _, _, crf2, ppp2 = wl.timed_run(NOISE_HARD, n2, 1)
            out["noise12"] = {"value": wl.B * n2 / dt2, "unit": "images/sec", "ms_per_step": 1e3 * dt2 / n2, "steps": n2,
                              "image_noise": NOISE_HARD, "batches_in_flight": 1, "crf": roofline_crf(crf2, n2, ppp2, wl.name, NOISE_HARD)}
            out["crf"]["second_operating_point"] = out["noise12"]["crf"]
            if P > 1:
                dt2p, _ = wl.timed_run_pipelined(P, NOISE_HARD, a.steps, 1)
                out["noise12"].update({"value": wl.B * a.steps / dt2p, "ms_per_step": 1e3 * dt2p / a.steps, "steps": a.steps,
                                       "batches_in_flight": P, "one_batch_at_a_time": {"value": wl.B * n2 / dt2, "ms_per_step": 1e3 * dt2 / n2}})
        if single and not a.no_noise12:
            # third operating point: photograph-like images (soft-edged regions, illumination gradients, texture, sensor
            # noise of 3 grey levels) -- fewer distinct colours than the random 8x8 blocks, so FEWER bilateral lattice points
            n3 = max(1, min(a.steps, 2))
            dt3, _, _, crf3, ppp3 = wl.timed_run("photo", n3, 1)
            out["photo"] = {"value": wl.B * n3 / dt3, "unit": "images/sec", "ms_per_step": 1e3 * dt3 / n3, "steps": n3,
                            "images": "synth.synth_photo_images(sigma=3)", "batches_in_flight": 1, "crf": roofline_crf(crf3, n3, ppp3, wl.name, "photo")}
            out["crf"]["photo_like_operating_point"] = out["photo"]["crf"]
        labels_head = state["ln"].clone()
        wl.close()
        torch.cuda.empty_cache()
        if single and not a.no_other_modes:
            # the other two compute modes on the headline workload, full --steps / --warmup each
            for pm in [m for m in ("f32", "bf16", "bf16x3") if m != a.dtype][:2]:
                w2 = Workload(a.config, pm, a.batch, a.crf_chunk, True if a.skip_1drop else None)
                r2, st2 = record(w2, P if pm != "f32" else 1, a.noise, a.steps, a.warmup)
                r2["label_pixels_differing_from_headline"] = float((st2["ln"] != labels_head).float().mean().item())
                w2.close()
                torch.cuda.empty_cache()
                if pm == "bf16":
                    r2["note"] = ("plain bf16 MFMA (BASELINE config 2's dtype): ~1 % error on image_embeds moves near-tie patch picks, "
                                  "so its label maps differ from the parity modes' (fraction above; bounded in "
                                  "tests/test_hip_parity.py::test_bf16_vs_f32_divergence_is_bounded)")
                    out["throughput_mode"] = r2
                else:
                    out.setdefault("parity_mode", {})[pm] = r2
        if single and not a.no_other_configs and a.config == "voc":
            # BASELINE configs 3-5 at their single-GPU shapes, same compute mode, short records
            out["other_configs"] = {}
            for name in ("psc59", "coco80", "ade768"):
                w3 = Workload(name, a.dtype)
                r3, _ = record(w3, CONFIGS[name]["pipelines"], a.noise, a.other_steps, 1)
                r3["config"] = workload_cfg(w3, a.noise)
                w3.close()
                torch.cuda.empty_cache()
                out["other_configs"][name] = r3
        if single and not a.no_cpu_baseline:
            flat, shapes, _ = weights_for(C.blip_itm_large(336))          # the same seeded weights the engines run on
            host, W_cpu, o = flat.cpu().numpy(), {}, 0
            for n, shp in shapes.items():
                k = int(np.prod(shp))
                W_cpu[n] = host[o:o + k].reshape(shp)
                o += k
            out["cpu_baseline"] = cpu_baseline(C.blip_itm_large(336), 0, a.cpu_images, a.noise, W=W_cpu)
        print(json.dumps(out))
        sys.stdout.flush()
    else:
        wl.close()
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
