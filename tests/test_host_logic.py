"""CPU: the product's host-side logic (pnp_ovss.host / tokenizer / datasets sharding) against the
golden vectors produced by the reference and against the oracle; multi-rank pieces over gloo."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from pnp_ovss import host
from pnp_ovss.tokenizer import SynthTokenizer
from oracle import pipeline_np as OP

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_parse_gpt_matches_reference(golden_dir):
    data = json.load(open(os.path.join(golden_dir, "gpt_parse.json")))
    for dt, d in data.items():
        for k, case in d["cases"].items():
            if "error" in case:
                with pytest.raises(Exception):
                    host.parse_gpt_classes(case["raw"], d["nms"])
                continue
            best, names, cap = host.parse_gpt_classes(case["raw"], d["nms"])
            assert (best, names, cap) == (case["best_class_idx"], case["classes"], case["caption"]), k


def test_merge_plan_reproduces_reference_merge(golden_dir):
    g = np.load(os.path.join(golden_dir, "merge_tokens.npz"))
    pieces = json.loads(str(g["pieces"]))
    tok = SynthTokenizer(1024)
    for i, cap in enumerate(str(c) for c in g["captions"]):
        n_cls = len(cap.split()[3:])
        enc = tok([cap], padding="max_length", max_length=500)
        pc = host.caption_pieces(tok, enc.input_ids[0].numpy())
        assert pc == pieces[i][4:-1]
        plan = host.merge_plan(pc, n_cls)
        src = g["maps"][i][3:-1]
        out = np.zeros((n_cls,) + src.shape[1:], np.float32)
        for c, (toks, div) in enumerate(plan):
            if toks:
                acc = src[toks[0]].copy()
                for t in toks[1:]:
                    acc = (acc + src[t]).astype(np.float32)
                out[c] = acc / np.float32(div) if div != 1 else acc
        np.testing.assert_array_equal(out, g[f"merged_{i}"])


def test_remap_lut_equals_sequential_remap():
    rng = np.random.default_rng(0)
    for _ in range(50):
        n = int(rng.integers(1, 8))
        best = [int(v) for v in rng.integers(0, 9, size=n)]          # small ids force index/id collisions
        for bg in (True, False):
            K = n + int(bg)
            lab = rng.integers(0, K, size=(13, 7)).astype(np.float32)
            lut = np.array(host.remap_lut(best, bg, K))
            np.testing.assert_array_equal(lut[lab.astype(int)].astype(np.float32), OP.remap_labels(lab, best, bg))


def test_shard_indices_matches_distributed_sampler():
    torch = pytest.importorskip("torch")
    from torch.utils.data.distributed import DistributedSampler
    for n, w in [(1449, 8), (10, 4), (7, 2), (5, 8)]:
        seen = []
        for r in range(w):
            ref = list(DistributedSampler(range(n), num_replicas=w, rank=r))
            assert host.shard_indices(n, r, w) == ref
            seen += ref
        assert set(seen) == set(range(n))


def test_scores_from_hist_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "hist_cases.npz"))
    s = host.scores_from_hist(g["hist"])
    assert s["Mean IoU"] == float(g["miou"]) and s["Pixel Accuracy"] == float(g["pixacc"])
    assert s["Frequency Weighted IoU"] == float(g["fwiou"]) and s["Mean Accuracy"] == float(g["macc"])


def test_gaussian_taps_equal_scipy_kernel():
    from pnp_ovss.hip import gaussian_taps
    from scipy.ndimage import _filters
    for h, w in [(336, 336), (375, 500), (768, 768), (33, 21)]:
        sigma = 0.05 * max(h, w)
        radius = int(4.0 * sigma + 0.5)
        ref = _filters._gaussian_kernel1d(sigma, 0, radius)[::-1]
        np.testing.assert_array_equal(gaussian_taps(h, w), ref[radius:])


_GLOO_WORKER = r'''
import os, sys, json
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "pnp-ovss_amd"))
from pnp_ovss import host
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
n = 11
idx = host.shard_indices(n, rank, world)
# each rank "segments" its shard: histogram of (gt, pred) pairs derived from the image index
hist = torch.zeros(21 * 21, dtype=torch.int64)
for i in idx:
    hist[(i % 21) * 21 + (3 * i) % 21] += 1
labels = torch.tensor(idx, dtype=torch.uint8)
dist.all_reduce(hist)                                   # the reduce bench.py / the CLI do over RCCL
gathered = [torch.empty_like(labels) for _ in range(world)] if rank == 0 else None
dist.gather(labels, gathered, dst=0)                    # mask-gather to rank 0
flat = torch.tensor([float(rank + 1)] * 5)
dist.broadcast(flat, src=0)                             # weight broadcast from rank 0
if rank == 0:
    print(json.dumps({"hist_sum": int(hist.sum()), "gathered": [g.tolist() for g in gathered], "bcast": flat.tolist()}))
dist.destroy_process_group()
'''


def test_world_size_2_sharding_reduce_gather_over_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_GLOO_WORKER)
    port = 29500 + os.getpid() % 2000
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script), ROOT],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["hist_sum"] == 12                         # 11 images padded to 12 (one duplicate), like the reference sampler
    assert sorted(res["gathered"][0] + res["gathered"][1]) == sorted(host.shard_indices(11, 0, 2) + host.shard_indices(11, 1, 2))
    assert res["bcast"] == [1.0] * 5


_CLI_COLLECTIVES_WORKER = r'''
import argparse, importlib.util, json, os, sys
import numpy as np, torch, torch.distributed as dist
root, save, mode = sys.argv[1], sys.argv[2], sys.argv[3]
sys.path.insert(0, os.path.join(root, "pnp-ovss_amd"))
spec = importlib.util.spec_from_file_location("pnp_cli", os.path.join(root, "pnp-ovss_amd", "PnP_OVSS_0514_updated_segmentation.py"))
cli = importlib.util.module_from_spec(spec); spec.loader.exec_module(cli)
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
args = argparse.Namespace(weights_sync="broadcast" if mode != "checksum_bad" else "checksum")
sync, info = cli.make_weights_sync(args, rank, world)
flat = torch.arange(1000, dtype=torch.float32) * (1.0 if rank == 0 else 3.0)     # rank 1 starts with other weights
if mode == "checksum_bad":
    try:
        sync(flat)
        print(json.dumps({"rank": rank, "raised": False}))
    except SystemExit as ex:
        print(json.dumps({"rank": rank, "raised": True, "msg": str(ex)[:80]}))
    dist.destroy_process_group()
    sys.exit(0)
sync(flat)
assert torch.equal(flat, torch.arange(1000, dtype=torch.float32)), "broadcast did not deliver rank 0's weights"
kept = {f"img{rank}_{i}": torch.full((3 + rank, 4 + i), 10 * rank + i, dtype=torch.uint8) for i in range(2 + rank)}
got = cli.gather_label_maps(kept, rank, world, torch.device("cpu"), save)
if rank == 0:
    z = dict(np.load(os.path.join(save, "label_maps.npz")))
    print(json.dumps({"info": {k: info[k] for k in ("mode", "bytes")}, "gathered": got,
                      "maps": {k: [list(v.shape), int(v.min()), int(v.max())] for k, v in z.items()}}))
dist.destroy_process_group()
'''


def _run_cli_collectives(tmp_path, mode):
    script = tmp_path / "cw.py"
    script.write_text(_CLI_COLLECTIVES_WORKER)
    port = 29500 + (os.getpid() + 31) % 2000
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script), ROOT, str(tmp_path), mode],
                          capture_output=True, text=True, timeout=300)


def test_cli_weights_broadcast_and_label_gather_world_size_2_gloo(tmp_path):
    """The CLI's own start-up and end-of-run collectives (PnP_OVSS_0514_updated_segmentation.py: make_weights_sync,
    gather_label_maps) with two ranks over gloo on CPU tensors: rank 1's buffer is overwritten by rank 0's (the DDP-constructor
    broadcast of PnP.py:1218), the digests agree afterwards, and rank 0 ends up with every rank's ragged label maps."""
    out = _run_cli_collectives(tmp_path, "broadcast")
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert res["info"] == {"mode": "broadcast", "bytes": 4000}
    assert res["gathered"] == {"images": 5, "bytes_per_rank": [3 * 4 + 3 * 5, 4 * 4 + 4 * 5 + 4 * 6]}
    assert res["maps"] == {"img0_0": [[3, 4], 0, 0], "img0_1": [[3, 5], 1, 1], "img1_0": [[4, 4], 10, 10],
                           "img1_1": [[4, 5], 11, 11], "img1_2": [[4, 6], 12, 12]}


def test_cli_weights_digest_mismatch_stops_every_rank(tmp_path):
    """`--weights_sync checksum` with ranks that loaded different weights: every rank raises (nobody runs on a replica that
    differs from rank 0's)."""
    out = _run_cli_collectives(tmp_path, "checksum_bad")
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 2 and all(l["raised"] for l in lines), lines


def test_weights_digest_sees_single_bit_flips_and_swaps():
    import torch
    from pnp_ovss.model import weights_digest
    a = torch.randn(100_000, generator=torch.Generator().manual_seed(0))
    b = a.clone()
    assert torch.equal(weights_digest(a), weights_digest(b))
    b.view(torch.int32)[77_777] ^= 1
    assert not torch.equal(weights_digest(a), weights_digest(b))
    c = a.clone()
    c[[5, 6]] = c[[6, 5]]                                  # same multiset of values: only the position-weighted sum moves
    d0, d1 = weights_digest(a), weights_digest(c)
    assert d0[0] == d1[0] and d0[1] != d1[1]


def _bench(*args, timeout=300):
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True, text=True,
                          timeout=timeout, env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})


def test_bench_launches_its_own_ranks_dry_run_gloo():
    """`python bench.py --gpus 2` (no torchrun) starts two rank processes itself; the dry run drives the same
    rendezvous, weight broadcast, histogram all-reduce and label gather on CPU tensors over gloo."""
    out = _bench("--gpus", "2", "--backend", "gloo", "--dry-run", "--steps", "2")
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert res["n_gpus"] == 2 and res["n_ranks"] == 2 and res["gathered_ranks"] == [0, 1]
    assert res["hist_total"] == 2 * 35 and res["images_per_rank"] == 70


def test_bench_eight_rank_dress_rehearsal_gloo():
    """The driver's 8-GPU scaling run rehearsed without an 8-GPU node (PnP.py:45-54 process group, :1218 weight broadcast,
    :1439 one process per GPU): `bench.py --gpus 8 --dry-run --backend gloo` -- launcher, rendezvous at 127.0.0.1, the three
    collectives of the path, the per-rank rate all-gather and the rank census on CPU tensors; the line carries the ranks the
    communicator really joined and the memory / thread budget of 8 ranks x 3 engines, asserted against one GPU's HBM."""
    out = _bench("--gpus", "8", "--backend", "gloo", "--dry-run", "--steps", "2", timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert res["n_gpus"] == 8 and res["n_ranks"] == 8
    assert res["rccl_ranks_seen"] == list(range(8)) and res["gathered_ranks"] == list(range(8))
    assert len(res["per_rank_images_per_sec"]) == 8 and res["hist_total"] == 8 * 35
    b = res["budget"]
    assert b["ranks"] == 8 and b["engines_per_rank"] == 3 and b["host_threads_total"] == 32
    assert b["device_GiB_per_rank"] < 0.9 * b["device_GiB_available"]


def test_bench_budget_refuses_engines_that_do_not_fit():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.resource_budget(8, "coco80", 3)["device_GiB_per_rank"] > 150
    with pytest.raises(AssertionError):
        bench.resource_budget(8, "coco80", 5)


def test_bench_launcher_propagates_a_failed_rank():
    out = _bench("--gpus", "2", "--backend", "gloo", "--dry-run", "--dry-run-fail-rank", "1", timeout=600)
    assert out.returncode != 0


def test_bench_under_torchrun_is_a_rank_not_a_launcher(tmp_path):
    port = 29500 + (os.getpid() + 7) % 2000
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--backend", "gloo", "--dry-run"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert res["n_ranks"] == 2


def test_wordpiece_tokenizer_matches_hf_bert_tokenizer_golden():
    """pnp_ovss.tokenizer.WordPieceTokenizer against ids produced by HF BertTokenizer + LAVIS' [DEC] / [ENC] additions
    on the committed tiny vocabulary (tests/golden/make_golden.py:gen_tokenizer): class-name captions of all five
    datasets (so ## splits decide the merge plan), punctuation, accents, unknown words, an over-long word, an empty
    caption; both call forms of the drivers and the per-id decode strings."""
    from pnp_ovss.tokenizer import WordPieceTokenizer
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "tokenizer_cases.json")))
    tok = WordPieceTokenizer(os.path.join(ROOT, "tests", "golden", "tiny_vocab.txt"))
    assert tok.vocab_size == g["vocab_size"] and tok.enc_token_id == g["enc_token_id"]
    caps = g["captions"]
    e = tok(caps, padding="max_length", max_length=500, return_tensors="pt")
    assert e.input_ids.shape == (len(caps), 500)
    for i, ref in enumerate(g["max_length_500"]["input_ids"]):
        n = int(e.attention_mask[i].sum())
        assert e.input_ids[i, :n].tolist() == ref, (i, caps[i])
        assert int(e.input_ids[i, n:].abs().sum()) == 0
    e2 = tok(caps, padding="longest", truncation=True, max_length=500, return_tensors="pt")
    assert list(e2.input_ids.shape) == g["longest"]["shape"]
    assert e2.input_ids.tolist() == g["longest"]["input_ids"] and e2.attention_mask.tolist() == g["longest"]["attention_mask"]
    e3 = tok(caps[:3], padding="longest", truncation=True, max_length=16, return_tensors="pt")
    assert e3.input_ids.tolist() == g["truncate_16"]["input_ids"] and e3.attention_mask.tolist() == g["truncate_16"]["attention_mask"]
    for i, s in g["decode"].items():
        assert tok.decode([int(i)]) == s, i
    # the merge plan the device kernel is fed with follows from those pieces (PnP.py:812-853)
    pieces = host.caption_pieces(tok, e.input_ids[8].numpy())          # "A picture of pottedplant tvmonitor"
    assert pieces == ["potted", "##plant", "tv", "##mon", "##itor"]
    assert host.merge_plan(pieces, 2) == [([0, 1], 2), ([2, 3, 4], 1)]


def test_pos_embed_retile_matches_reference_interpolate_pos_embed():
    """base_model.py:44-73 on the two re-tilings the configs need (24^2 -> 21^2 for 336 px, 24^2 -> 48^2 for 768 px):
    the product's host function and the oracle's numpy bicubic against the reference's own output."""
    from pnp_ovss.model import _resize_pos_embed
    from oracle import blip_itm_np as OM
    g = np.load(os.path.join(ROOT, "tests", "golden", "pos_embed_cases.npz"))
    for grid in (21, 48, 24):
        ref = g[f"pos_{grid}_from_24"]
        got = _resize_pos_embed(g["pos_24"], grid).numpy()
        assert got.shape == ref.shape == (1, 1 + grid * grid, 24)
        np.testing.assert_array_equal(got[:, 0], g["pos_24"][:, 0])          # class token untouched
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-6)
        np.testing.assert_allclose(OM.interpolate_pos_embed(g["pos_24"], grid), ref, rtol=0, atol=2e-5)   # float64 taps vs torch float32


def test_merge_checkpoint_matches_reference_load_checkpoint():
    """BaseModel.load_checkpoint semantics (base_model.py:86-125) on a synthetic checkpoint saved at another
    resolution with one shape-mismatched key: golden = the reference model's state after ITS load_checkpoint."""
    from pnp_ovss import config as C, synth
    from pnp_ovss.model import merge_checkpoint
    g = np.load(os.path.join(ROOT, "tests", "golden", "checkpoint_small.npz"))
    cfg = C.ModelCfg(**json.loads(str(g["cfg"])))
    cfg_ck = C.ModelCfg(**json.loads(str(g["cfg_ckpt"])))
    init = synth.synth_state_dict(cfg, int(g["init_seed"]))
    ck = synth.synth_checkpoint(cfg, cfg_ck, int(g["ckpt_seed"]))
    import torch
    state, dropped, missing = merge_checkpoint(cfg, {k: torch.from_numpy(v) for k, v in ck.items()}, init)
    assert dropped == ["itm_head.bias"] and missing == []
    assert "itm_head.bias" in [str(k) for k in g["missing"]]                  # the reference reports it as not loaded too
    np.testing.assert_allclose(np.asarray(state["visual_encoder.pos_embed"]), g["pos_embed"], rtol=0, atol=1e-6)
    np.testing.assert_array_equal(np.asarray(state["itm_head.bias"]), g["itm_head_bias"])       # init value kept
    np.testing.assert_array_equal(np.asarray(state["itm_head.weight"]), ck["itm_head.weight"])  # checkpoint value taken


def test_host_tables_equal_sequential_reference_semantics_randomised():
    """Property tests of the two host-side tables the device kernels are fed with, against direct restatements of the
    reference's sequential code on random inputs: remap_lut == the in-place descending remap applied to every pixel value
    (PnP.py:390-399 / PnPc.py:458-463, collisions included), merge_plan == the oracle's token walk (PnP.py:820-853)."""
    from oracle import pipeline_np as OP
    rng = np.random.default_rng(2024)
    for _ in range(300):
        n = int(rng.integers(1, 12))
        best = [int(v) for v in rng.integers(0, 20, size=n)]
        bg = bool(rng.integers(0, 2))
        ids = [int(v) for v in rng.choice(np.arange(1, 91), size=20, replace=False)] if rng.integers(0, 2) else None
        K = n + int(bg)
        lut = host.remap_lut(best, bg, K, ids)
        lab = np.arange(K, dtype=np.float32)
        ref = OP.remap_labels(lab, best, bg, ids)
        assert lut == [int(v) for v in ref], (best, bg, ids)
    words = ["cat", "dog", "aeroplane", "pottedplant", "tvmonitor", "diningtable", "bus", "refrigerator"]
    from pnp_ovss.tokenizer import SynthTokenizer
    tok = SynthTokenizer(2048)
    for _ in range(200):
        names = [words[i] for i in rng.integers(0, len(words), size=int(rng.integers(1, 7)))]
        pieces = tok.tokenize(" ".join(names))
        ours = host.merge_plan(pieces, len(names))
        ref = OP.merge_plan(pieces, len(names))
        if ref is None:
            assert ours == [([i], 1) for i in range(len(names))]
        else:
            assert ours == ref, (names, pieces)


def wordpiece_merge_cases():
    """(caption index, pieces, class words, per-word piece spans) for the class-name captions of tokenizer_cases.json:
    every VOC / Pascal-Context / ADE20K / COCO class word as HF's BertTokenizer splits it on the committed vocabulary
    (ids pinned by the fixture).  Shared with the GPU twin in test_hip_parity.py."""
    from pnp_ovss.tokenizer import WordPieceTokenizer
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "tokenizer_cases.json")))
    tok = WordPieceTokenizer(os.path.join(ROOT, "tests", "golden", "tiny_vocab.txt"))
    cases = []
    for i, cap in enumerate(g["captions"][:9]):
        ids = g["max_length_500"]["input_ids"][i]                 # HF's own ids, not re-tokenised
        pieces = host.caption_pieces(tok, ids)
        words = cap.split()[3:]
        spans, p = [], 0
        for w in words:                                           # a word = one plain piece + its ## continuations
            q = p + 1
            while q < len(pieces) and pieces[q].startswith("##"):
                q += 1
            assert "".join(x[2:] if x.startswith("##") else x for x in pieces[p:q]) == w, (w, pieces[p:q])
            spans.append((p, q))
            p = q
        assert p == len(pieces)
        cases.append((i, pieces, words, spans))
    return cases


def expected_word_merge(maps, spans):
    """What Mean_over_filtered_label_tokens (PnP.py:810-853) computes, stated per class WORD instead of as a token walk:
    a word's map is the in-order sum of its pieces' rows of map[3:-1], divided by the piece count -- except for the LAST
    word of the caption when it is split (the walk only divides when another word follows: summed, not averaged); when
    no word is split the reference returns map[3:-1][:C] untouched."""
    g = maps[3:-1]
    out = np.zeros((len(spans),) + g.shape[1:], dtype=np.float32)
    for c, (p, q) in enumerate(spans):
        acc = g[p].copy()
        for t in range(p + 1, q):
            acc = (acc + g[t]).astype(np.float32)
        if q - p > 1 and c + 1 < len(spans):
            acc = (acc / np.float32(q - p)).astype(np.float32)
        out[c] = acc
    return out


def test_merge_plan_on_real_tokenizer_splits_of_dataset_class_names():
    """Word-piece splits as a real BertTokenizer produces them (tokenizer_cases.json: `pottedplant`, `tvmonitor`,
    `bedclothes`, ... split into up to a dozen `##` pieces) through host.merge_plan, against the word-level statement of
    the reference's merge above AND the oracle's token walk -- the cases the tiny synthetic vocabulary of the other
    fixtures only touches with two words.  (The device kernel is fed these plans: GPU twin in test_hip_parity.py.)"""
    from oracle import pipeline_np as OP
    rng = np.random.default_rng(7)
    n_split_words = 0
    for i, pieces, words, spans in wordpiece_merge_cases():
        C = len(words)
        plan = host.merge_plan(pieces, C)
        maps = rng.random((3 + len(pieces) + 1, 5, 5), dtype=np.float32)
        want = expected_word_merge(maps, spans)
        got = np.zeros_like(want)
        for c, (toks, div) in enumerate(plan):
            acc = maps[3:-1][toks[0]].copy()
            for t in toks[1:]:
                acc = (acc + maps[3:-1][t]).astype(np.float32)
            got[c] = acc if div == 1 else (acc / np.float32(div)).astype(np.float32)
        np.testing.assert_array_equal(got, want, err_msg=f"caption {i}")
        np.testing.assert_array_equal(OP.merge_tokens(maps, pieces, C), want, err_msg=f"oracle, caption {i}")
        for c, (p, q) in enumerate(spans):
            assert plan[c][0] == list(range(p, q))
            n_split_words += q - p > 1
    assert n_split_words >= 140                                   # the fixture really is about split words (149 of 267)


def test_inject_outliers_is_function_preserving_on_the_oracle():
    """synth.inject_outliers (the weights of tests/golden/droploop_large_outliers.npz): a power-of-two gain with the consuming weights
    scaled back leaves the oracle's maps BIT-identical (scaling by 2^k is exact in binary floating point: a control, not a stress);
    jittered gains change them only in the last bits; without compensation the model is a different one."""
    from pnp_ovss import config as C, synth
    from oracle import blip_itm_np as OM
    cfg = C.blip_itm_small(64)
    W = synth.synth_state_dict(cfg, 3)
    _, imgs = synth.synth_images(2, cfg.img_size, seed=5)
    ids, mask = synth.synth_tokens(cfg, [4, 2], seed=1)
    m0, l0, _ = OM.compute_gradcam(W, cfg, imgs, ids, mask, layers=[7])
    m1, l1, _ = OM.compute_gradcam(synth.inject_outliers(W, cfg, 16.0), cfg, imgs, ids, mask, layers=[7])
    assert np.array_equal(m1[7], m0[7]) and np.array_equal(l1, l0)
    m2, l2, _ = OM.compute_gradcam(synth.inject_outliers(W, cfg, 16.0, jitter=0.5), cfg, imgs, ids, mask, layers=[7])
    assert 0 < np.abs(m2[7] - m0[7]).max() < 1e-5 * np.abs(m0[7]).max() + 1e-6
    m3, _, _ = OM.compute_gradcam(synth.inject_outliers(W, cfg, 8.0, jitter=0.5, compensate=False), cfg, imgs, ids, mask, layers=[7])
    assert np.abs(m3[7] - m0[7]).max() > 1e-3
    # the three variants of the fixture draw different channels / gains (same seed: reparam64 would be reparam16 x 4, bit-identical)
    g = np.load(os.path.join(ROOT, "tests", "golden", "droploop_large_outliers.npz"), allow_pickle=False)
    v = json.loads(str(g["variants"]))
    assert v["reparam16"].get("seed", 99) != v["reparam64"].get("seed", 99)
    assert not np.array_equal(g["reparam16_agg"], g["reparam64_agg"])
