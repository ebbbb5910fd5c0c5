#!/usr/bin/env python
"""Drop-in for the reference's `PnP_OVSS_0514_updated_segmentation.py` command line on MI355X.

Same flags (reference get_args_parser, :57-106), same one-process-per-GPU layout (:1439 mp.spawn ->
here torchrun or the built-in spawn), same outputs: per-batch confusion matrices saved as
`{save_path}/hist_withfiltered_caption/img_{id}_max_blocknum_{L}_atthead_{h}.npy` and
`{save_path}/all_drop_hist_with_filtered_caption/...` (:505-520) so the reference's own
Calculate_mIoU.py reads them.  New: `--data_type synthetic` (no dataset on disk), `--dtype`,
`--checkpoint`, `--vocab`; the final histogram is additionally all-reduced over RCCL and printed.

Multi-rank (the reference: DDP-constructor broadcast PnP.py:1218, file-system reduce :513-520):
  * `--weights_sync broadcast` (default): rank 0 loads the checkpoint, every rank receives rank 0's weights as one flat fp32
    buffer over RCCL and the ranks compare a digest of what they hold -- a mismatch stops the run on every rank;
    `checksum`: every rank loads the checkpoint itself (what the reference does in front of its DDP wrapper) and only the
    digest is compared; `none`: no collective at start-up;
  * the image count and the confusion matrix of the summary line are all-reduced;
  * `--gather_labels`: the final (N-drop) uint8 label maps of every rank are gathered on rank 0 over RCCL and written to
    `{save_path}/label_maps.npz` (image id -> H x W), north_star's mask-gather.

Device work happens in libpnp_hip.so via pnp_ovss.model.Segmenter; this file is host orchestration.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from pnp_ovss import host, synth  # noqa: E402
from pnp_ovss.datasets import make_dataset, prefetch, wait_ready  # noqa: E402
from pnp_ovss.model import Segmenter  # noqa: E402


def get_args_parser():
    p = argparse.ArgumentParser("image caption localization with ITM", add_help=True)
    p.add_argument("--batch_size", default=2, type=int)
    p.add_argument("--num_workers", default=0, type=int)
    p.add_argument("--gen_multiplecap_withpnpvqa", default="label")
    p.add_argument("--save_path", default="Eval_test_ddp")
    p.add_argument("--home_dir", default="/home/letitiabanana/LAVIS/")
    p.add_argument("--master_port", default="12355")
    p.add_argument("--existing_att_path", default=None)
    p.add_argument("--cam_out_dir", default=None, type=str)
    p.add_argument("--del_patch_num", default=None)
    p.add_argument("--max_att_block_num", default=10, type=int)
    p.add_argument("--img_size", default=768, type=int)
    p.add_argument("--world_size", default=4, type=int)
    p.add_argument("--ensemble_blocks", default=None, type=str)
    p.add_argument("--drop_iter", default=10, type=int)
    p.add_argument("--prune_att_head", default=None)
    p.add_argument("--sort_threshold", default=None, type=float)
    p.add_argument("--edge_map_for_clip", action="store_true")
    p.add_argument("--final_att_threshold", default=0.05)
    p.add_argument("--search", default=None)
    p.add_argument("--layer", default=None, type=str)
    p.add_argument("--cal_token_sim_forall_layerhead", action="store_true")
    p.add_argument("--in_the_wild", action="store_true")
    p.add_argument("--data_type", default=None, type=str, help="voc, psc, ade20k, coco_object, coco_stuff or synthetic")
    p.add_argument("--postprocess", default=None, type=str, help="blur or crf or blur+crf")
    p.add_argument("--threshold", default=None, type=float)
    # additions
    p.add_argument("--dtype", default="f32", choices=["f32", "bf16x3", "bf16"],
                   help="f32 (default): the reference's arithmetic; bf16x3: split-bf16 -- fp32-class products on the bf16 MFMA, "
                        "same patch picks as f32 in every test, 2x its throughput; bf16: 3.8x the throughput of f32, but ~1%% error "
                        "on image_embeds moves near-tie patch picks, so ~5-10%% of label pixels differ "
                        "(tests/test_hip_parity.py::test_bf16_vs_f32_divergence_is_bounded)")
    p.add_argument("--crf_chunk", default=0, type=int, help="images per DenseCRF launch group (0 = the whole batch)")
    p.add_argument("--checkpoint", default=None, help="BLIP ITM-large checkpoint (.pth); default: seeded synthetic weights")
    p.add_argument("--vocab", default=None, help="bert-base-uncased vocab.txt")
    p.add_argument("--device_jpeg", default=1, type=int, help="1: decode JPEG files on the GPU (baseline files; others fall back to Pillow)")
    p.add_argument("--pipelines", default=1, type=int,
                   help="batches in flight on this GPU: P model replicas, each with its own HIP stream and host thread, take the "
                        "batches as they come (the text side and kernel tails of one batch run beside another batch's dense "
                        "kernels: +10-14 %% images/s at P = 3; results are identical, the per-batch lines may print out of order)")
    p.add_argument("--weights_sync", default="broadcast", choices=["broadcast", "checksum", "none"],
                   help="world_size > 1: RCCL broadcast of rank 0's weights + digest check (default), digest check of per-rank loads, or nothing")
    p.add_argument("--gather_labels", action="store_true",
                   help="gather every rank's final label maps on rank 0 (RCCL) and write {save_path}/label_maps.npz")
    p.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend (nccl = RCCL on ROCm)")
    p.add_argument("--share_gpu", action="store_true",
                   help="diagnostic: every rank uses cuda:0 (rehearses world_size > 1 on a one-GPU box with --backend gloo; RCCL "
                        "refuses two ranks on one device)")
    p.add_argument("--synthetic_images", default=70, type=int)
    p.add_argument("--max_batches", default=0, type=int)
    return p


def ddp_setup(args, rank, world_size):
    """reference :45-54 (NCCL on localhost) -> RCCL via torch.distributed; gloo when no GPU."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(args.master_port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    backend = args.backend if torch.cuda.is_available() else "gloo"
    dist.init_process_group(backend=backend, rank=rank, world_size=world_size)


def make_weights_sync(args, rank, world_size):
    """The start-up collective on the flat fp32 weight buffer (pnp_ovss.model.build_model(sync=...)): broadcast of rank 0's
    weights (the reference's DDP(model) constructor, PnP.py:1218) and / or a digest comparison that fails on EVERY rank."""
    from pnp_ovss.model import weights_digest
    info = {}

    def sync(flat):
        if world_size <= 1 or args.weights_sync == "none":
            return
        t0 = time.perf_counter()
        if args.weights_sync == "broadcast":
            dist.broadcast(flat, src=0)
        dg = weights_digest(flat)
        every = [torch.zeros_like(dg) for _ in range(world_size)]
        dist.all_gather(every, dg)
        if flat.is_cuda:
            torch.cuda.synchronize()
        info.update(mode=args.weights_sync, bytes=int(flat.numel()) * 4, ms=1e3 * (time.perf_counter() - t0),
                    digest=[int(v) for v in dg.cpu()])
        bad = [r for r, d in enumerate(every) if not torch.equal(d, every[0])]
        if bad:
            raise SystemExit(f"rank {rank}: weights differ from rank 0's on rank(s) {bad} after --weights_sync {args.weights_sync} "
                             f"(digests {[[int(v) for v in d.cpu()] for d in every]})")
    return sync, info


def gather_label_maps(kept, rank, world_size, dev, save_path):
    """north_star's mask-gather: every rank's final uint8 label maps to rank 0 in ONE padded device-to-device gather
    (RCCL; the ids / sizes travel as a small object gather), written as {save_path}/label_maps.npz (image id -> H x W).
    Replaces nothing in the reference, which keeps label maps only inside save_img_union_attention (PnP.py:390-399)."""
    ids = sorted(kept)
    shapes = [tuple(int(v) for v in kept[i].shape) for i in ids]
    buf = torch.cat([kept[i].reshape(-1) for i in ids]).to(dev) if ids else torch.zeros(0, dtype=torch.uint8, device=dev)
    metas = [(ids, shapes)]
    bufs = [buf]
    if world_size > 1:
        metas = [None] * world_size
        dist.all_gather_object(metas, (ids, shapes))
        cap = max(sum(h * w for h, w in m[1]) for m in metas)
        pad = torch.zeros(max(cap, 1), dtype=torch.uint8, device=dev)
        pad[: buf.numel()] = buf
        bufs = [torch.empty_like(pad) for _ in range(world_size)] if rank == 0 else None
        dist.gather(pad, bufs, dst=0)
    if rank != 0:
        return None
    out = {}
    for (r_ids, r_shapes), b in zip(metas, bufs):
        host_b, o = b.cpu().numpy(), 0
        for i, (h, w) in zip(r_ids, r_shapes):
            out[i] = host_b[o:o + h * w].reshape(h, w)
            o += h * w
    np.savez_compressed(os.path.join(save_path, "label_maps.npz"), **out)
    return {"images": len(out), "bytes_per_rank": [int(sum(h * w for h, w in m[1])) for m in metas]}


def main(rank, world_size, args):
    tic = time.perf_counter()
    if world_size > 1:
        ddp_setup(args, rank, world_size)
    dev_idx = 0 if args.share_gpu else rank
    args.device_index = dev_idx                 # what the dataset's device-side decode / resize runs on
    torch.cuda.set_device(dev_idx)
    if args.prune_att_head is None:
        raise SystemExit("--prune_att_head is required (reference :277)")
    if args.del_patch_num is None or "sort_thresh" not in args.del_patch_num:
        raise SystemExit('--del_patch_num must contain "sort_thresh" (reference :645-647)')
    ds = make_dataset(args, rank, world_size)
    from lavis.models import load_model_and_preprocess
    stash_layer = args.max_att_block_num - 1
    if args.ensemble_blocks is not None and "saveall" in args.ensemble_blocks:
        stash_layer = int(args.layer) - 1 if args.layer else 0         # lowest layer of the sweep (default: all 12)
    sync, sync_info = make_weights_sync(args, rank, world_size)
    receive_only = world_size > 1 and args.weights_sync == "broadcast" and rank != 0
    model, vis_processors, text_processors = load_model_and_preprocess(
        "blip_image_text_matching", "large", device=dev_idx, is_eval=True, img_size=args.img_size,
        max_batch=args.batch_size, stash_layer=stash_layer, mode=args.dtype,
        checkpoint=args.checkpoint, vocab=args.vocab, max_text_len=ds.max_text_len,
        sync=sync if world_size > 1 and args.weights_sync != "none" else None, receive_only=receive_only)
    if rank == 0 and sync_info:
        print(f"weights: --weights_sync {sync_info['mode']} of {sync_info['bytes'] / 1e9:.2f} GB over {world_size} ranks in "
              f"{sync_info['ms']:.0f} ms, digests equal", flush=True)
    coco = args.data_type in ("coco_object", "coco_stuff")
    n_class = host.coco_n_class(args.data_type) if coco else len(ds.cats) + 1        # PnPc.py:597-600 / PnP.py:1115
    seg = Segmenter(model, args.data_type if args.data_type != "synthetic" else "voc", n_class, threshold=args.threshold,
                    postprocess=args.postprocess, max_pixels_per_image=ds.max_pixels, max_channels=ds.max_channels,
                    crf_chunk=args.crf_chunk, class_ids=ds.class_ids)
    # --ensemble_blocks saveall (reference get_grad_cam_labelascaption: `for block in range(0, 12): for head in
    # range(0, 12)` around save_img_union_attention): one evaluation per (text layer, head), histograms saved under
    # that pair's file name; the model keeps maps from the lowest swept layer up (stash_layer), one forward per pair
    sweep = args.ensemble_blocks is not None and "saveall" in args.ensemble_blocks
    pairs = [(b + 1, h) for b in range(stash_layer, 12) for h in range(12)] if sweep else [(args.max_att_block_num, int(args.prune_att_head))]
    for d in ("hist_withfiltered_caption", "all_drop_hist_with_filtered_caption"):
        Path(f"{args.save_path}/{d}/").mkdir(parents=True, exist_ok=True)
    replicas = [(model, seg)]          # --pipelines: P engines on ONE weight copy (pnp_create_shared: the first model is the donor)
    for _ in range(max(0, args.pipelines - 1)):
        m2, _, _ = load_model_and_preprocess(
            "blip_image_text_matching", "large", device=dev_idx, is_eval=True, img_size=args.img_size,
            max_batch=args.batch_size, stash_layer=stash_layer, mode=args.dtype,
            vocab=args.vocab, max_text_len=ds.max_text_len, donor=model)
        replicas.append((m2, Segmenter(m2, args.data_type if args.data_type != "synthetic" else "voc", n_class,
                                       threshold=args.threshold, postprocess=args.postprocess,
                                       max_pixels_per_image=ds.max_pixels, max_channels=ds.max_channels,
                                       crf_chunk=args.crf_chunk, class_ids=ds.class_ids)))
    if args.pipelines > 1 and rank == 0:
        free_b, total_b = torch.cuda.mem_get_info()
        used = [m.engine.allocated_bytes() for m, _ in replicas]
        print(f"--pipelines {args.pipelines}: {sum(used) / 2**30:.1f} GiB in {len(used)} engines "
              f"(the first holds the one weight copy: {used[0] / 2**30:.1f} GiB; the others workspace only: {min(used) / 2**30:.1f} GiB), "
              f"{free_b / 2**30:.1f} GiB of {total_b / 2**30:.0f} GiB free", flush=True)
        if free_b < 0.05 * total_b:
            print("warning: less than 5 % of device memory free: lower --pipelines or --batch_size", flush=True)
    n_img = 0
    t_loop = time.perf_counter()
    # Two batches in flight: the host half of batch i+1 (class lookup, tokenisation, merge plans, uploads) and the
    # bookkeeping of batch i-1 (histogram read-back, .npy files, the per-batch line) run while the GPU works on batch i,
    # so each batch accumulates into its own pair of confusion matrices.
    hist_ring = [(torch.zeros_like(seg.hist_1drop), torch.zeros_like(seg.hist_ndrop), torch.cuda.Event()) for _ in range(2)]

    def finish(job):
        img_ids, layer, head, l1, ln, (h1d, hnd, done) = job
        done.synchronize()
        h1 = h1d.cpu().numpy().reshape(n_class, n_class).astype(np.float64)
        hn = hnd.cpu().numpy().reshape(n_class, n_class).astype(np.float64)
        first = img_ids[0]
        if l1:                                  # the COCO driver skips the 1-drop branch when drop_iter >= 3 (PnPc.py:420,633)
            np.save(f"{args.save_path}/hist_withfiltered_caption/img_{first}_max_blocknum_{layer}_atthead_{head}.npy", h1)
        if ln:
            np.save(f"{args.save_path}/all_drop_hist_with_filtered_caption/img_{first}_max_blocknum_{layer}_atthead_{head}.npy", hn)
        print(img_ids[:3], f"layer {layer} head {head}", "miou filtered_caption",
              host.scores_from_hist(h1)["Mean IoU"] if l1 else None,
              "miou all_drop", host.scores_from_hist(hn)["Mean IoU"] if ln else None, flush=True)
        return hn if ln else h1

    # --gather_labels: image id -> uint8 label map (last (layer, head) wins).  The maps leave the device batch by batch -- an
    # asynchronous copy into pinned host memory on the launch's own stream -- so a 2000-image ADE20K sweep at 768^2 holds its
    # 1.2 GB of label maps on the host, not beside the engines in HBM; they go back to the device only for the final RCCL gather
    kept = {}

    def keep_labels(img_ids, maps):
        for i, m in zip(img_ids, maps):
            src = m.to(torch.uint8)                          # (a device temporary; the launch's views are only valid until the next launch)
            dst = torch.empty(src.shape, dtype=torch.uint8, pin_memory=True)
            dst.copy_(src, non_blocking=True)                # ordered behind the launch on the current stream; read after synchronize()
            kept[str(i)] = dst

    if args.pipelines > 1:
        # P replicas (model + Segmenter + stream + host thread) consume the batch stream; each batch still runs the whole
        # path on one replica, into that replica's own pair of confusion matrices (bench.py --pipelines is the same scheme)
        import threading
        it = enumerate(prefetch(ds.batches(args.batch_size), depth=2 * args.pipelines))
        lock, errors, counts = threading.Lock(), [], [0]

        def worker(rep):
            try:
                torch.cuda.set_device(dev_idx)
                sg = rep[1]
                ring = (torch.zeros_like(sg.hist_1drop), torch.zeros_like(sg.hist_ndrop), torch.cuda.Event())
                with torch.cuda.stream(torch.cuda.Stream()):
                    while True:
                        with lock:
                            if errors:                        # another replica failed: stop taking batches
                                return
                            bi, batch = next(it, (None, None))
                            if batch is None or (args.max_batches and bi >= args.max_batches):
                                return
                        wait_ready(batch)                     # produced on the prefetch thread's stream, consumed on this one
                        best, caps = [], []
                        for img_id in batch["img_ids"]:
                            b, names, cap = ds.predicted_classes(img_id)
                            best.append(b)
                            caps.append(cap)
                        prep = sg.prepare(caps, best, batch["org_images"], batch["label_trues"], batch.get("gt_dev"))
                        last = None
                        for layer, head in pairs:
                            pargs = argparse.Namespace(**{**vars(args), "max_att_block_num": layer, "prune_att_head": str(head)})
                            ring[0].zero_()
                            ring[1].zero_()
                            l1, ln = sg.launch(pargs, batch["imgs"], prep, run_1drop=True, hists=ring[:2])
                            if args.gather_labels:
                                keep_labels(batch["img_ids"], ln if ln is not None else l1)      # (dict stores of distinct keys: no lock)
                            ring[2].record()
                            last = finish((batch["img_ids"], layer, head, l1 is not None, ln is not None, ring))
                        with lock:
                            ds.total_hist += last       # summary line: the last (layer, head) of a batch's sweep
                            counts[0] += len(batch["img_ids"])
            except BaseException as ex:      # noqa: BLE001 -- re-raised below
                errors.append(ex)

        threads = [threading.Thread(target=worker, args=(rep,)) for rep in replicas]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
        n_img = counts[0]
    pending, slot = None, 0
    for bi, batch in enumerate(prefetch(ds.batches(args.batch_size), depth=2) if args.pipelines <= 1 else ()):
        if args.max_batches and bi >= args.max_batches:
            break
        best, caps = [], []
        for img_id in batch["img_ids"]:
            b, names, cap = ds.predicted_classes(img_id)
            best.append(b)
            caps.append(cap)
        wait_ready(batch)
        prep = seg.prepare(caps, best, batch["org_images"], batch["label_trues"], batch.get("gt_dev"))
        for layer, head in pairs:
            pargs = argparse.Namespace(**{**vars(args), "max_att_block_num": layer, "prune_att_head": str(head)})
            ring = hist_ring[slot]
            slot ^= 1
            ring[0].zero_()
            ring[1].zero_()
            l1, ln = seg.launch(pargs, batch["imgs"], prep, run_1drop=True, hists=ring[:2])
            if args.gather_labels:
                keep_labels(batch["img_ids"], ln if ln is not None else l1)
            ring[2].record()
            job = (batch["img_ids"], layer, head, l1 is not None, ln is not None, ring)
            if pending is not None:
                last = finish(pending)          # the previous launch: long done, or at most the one before this one
                if pending[0] is not batch["img_ids"]:
                    ds.total_hist += last       # summary line: the last (layer, head) of a batch's sweep
            pending = job
        n_img += len(batch["img_ids"])
    if pending is not None:
        ds.total_hist += finish(pending)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t_loop
    for m_, _ in replicas:                      # split-bf16 engines: the stream-K tail's bounded spins must never have given up
        st = getattr(m_.engine, "streamk_status", None)
        if st is not None and st()[1]:
            raise RuntimeError(f"a stream-K owner gave up waiting for a partial tile (word {st()[1]}): results invalid")
    dev = torch.device("cuda", dev_idx)
    total = torch.from_numpy(ds.total_hist).to(dev)
    count = torch.tensor([n_img], device=dev, dtype=torch.int64)
    if world_size > 1:
        dist.all_reduce(total)                  # RCCL reduce of the confusion matrix (the reference sums files offline)
        dist.all_reduce(count)                  # images actually processed (DistributedSampler pads: a few count twice, as in the reference)
    gathered = None
    if args.gather_labels:
        gathered = gather_label_maps(kept, rank, world_size, dev, args.save_path)
    if rank == 0:
        s = host.scores_from_hist(total.cpu().numpy())
        line = {"images": int(count.item()), "images_rank0": n_img, "ranks": world_size, "images_per_sec_rank0": n_img / dt,
                "Mean IoU": float(s["Mean IoU"]), "Pixel Accuracy": float(s["Pixel Accuracy"]),
                "FWIoU": float(s["Frequency Weighted IoU"]), "pixels": float(total.sum().item())}
        if sync_info:
            line["weights_sync"] = {k: sync_info[k] for k in ("mode", "bytes", "ms")}
        if gathered is not None:
            line["gathered_label_maps"] = gathered
        print(json.dumps(line))
        print(f"Time: total running time for {n_img} images/rank {time.perf_counter() - tic:0.4f} seconds")
    if world_size > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    args = get_args_parser().parse_args()
    if "RANK" in os.environ:                    # launched by torchrun: one rank per GPU
        main(int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), args)
    elif args.world_size > 1:
        import torch.multiprocessing as mp
        mp.spawn(main, args=(args.world_size, args), nprocs=args.world_size)
    else:
        main(0, 1, args)
