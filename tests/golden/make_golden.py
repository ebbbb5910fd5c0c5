#!/usr/bin/env python
"""Generate golden vectors by RUNNING THE REFERENCE ITSELF (build container only; needs
/root/reference).  Outputs small .npz / .json fixtures next to this script; tests and the GPU box
only ever read those fixtures.

    python tests/golden/make_golden.py [--only NAME] [--skip-large]

Fixtures (inputs are regenerated from seeds by pnp_ovss.synth; only outputs are stored):
  gradcam_small.npz   compute_gradcam_ensemble, small geometry, B=2 ragged captions, all 12x12 maps
  gradcam_large.npz   compute_gradcam_ensemble, BLIP-ITM-large 336^2, B=1, L=25: map [7][9] + logits
  droploop_small.npz  Inference_BLIP_filteredcaption (drop_iter 4 and 1), picks per iteration
  droploop_large.npz  the same at BLIP-ITM-large 336^2, B=2 ragged captions, drop_iter 4
  droploop_large_outliers.npz  the same with outlier channels injected into the seeded weights (synth.inject_outliers)
  merge_tokens.npz    Mean_over_filtered_label_tokens on split / unsplit captions
  pipeline_voc.npz    save_img_union_attention end to end (blur / no post-process; CRF is not
  pipeline_psc.npz    importable here -> parity unpinned for CRF), hist .npy contents
  pipeline_voc_large.npz  the same at the HEADLINE geometry: BLIP-ITM-large 336^2, 20-class prompt, labels + hists only
  pipeline_coco_object.npz / pipeline_coco_stuff.npz   the COCO driver's save_img_union_attention
                      (PnP_OVSS_0514_updated_segmentation_coco.py) at drop_iter 4 (N-drop only) and 2 (both branches)
  gpt_parse_coco.json the COCO driver's Load_predicted_classes on sampled strings (category id -> cats position)
  gpt_parse.json      Load_predicted_classes on sampled GPT-4o strings of the shipped JSON files
  blur_cases.npz      `blurring` (scipy gaussian_filter + min-max) incl. non-square and NaN cases
  hist_cases.npz      _fast_hist / scores on hand-made label maps
"""
import argparse
import json
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "pnp-ovss_amd"))
sys.path.insert(0, HERE)

from pnp_ovss import config as C            # noqa: E402
from pnp_ovss import synth                  # noqa: E402
from pnp_ovss.tokenizer import SynthTokenizer  # noqa: E402
import _ref_loader as RL                    # noqa: E402

torch.set_grad_enabled(True)


def _model(cfg, seed):
    tok = SynthTokenizer(cfg.vocab)
    sd = synth.synth_state_dict(cfg, seed)
    m, itm = RL.build_reference_model(cfg, sd, tok)
    return m, itm, tok


def gen_gradcam_small():
    cfg = C.blip_itm_small(64)
    m, itm, tok = _model(cfg, seed=3)
    _, imgs = synth.synth_images(2, cfg.img_size, seed=5)
    caps = ["A picture of cat aeroplane dog sheep boat", "A picture of bus tvmonitor"]
    tok500 = tok(caps, padding="max_length", max_length=500, return_tensors="pt")
    args = argparse.Namespace(img_size=cfg.img_size)
    g, _, out = itm.compute_gradcam_ensemble(args, m, torch.from_numpy(imgs), caps, tok500)
    maps = np.stack([np.stack([g[l][h].numpy() for h in range(12)]) for l in range(12)])
    layer = m.text_encoder.base_model.base_model.encoder.layer
    P7 = layer[7].crossattention.self.get_attention_map().detach().numpy()
    dP7 = layer[7].crossattention.self.get_attn_gradients().detach().numpy()
    dP0 = layer[0].crossattention.self.get_attn_gradients().detach().numpy()
    np.savez_compressed(os.path.join(HERE, "gradcam_small.npz"),
                        cfg=json.dumps(cfg.as_dict()), weight_seed=3, image_seed=5,
                        captions=np.array(caps), input_ids=tok500.input_ids.numpy(),
                        attention_mask=tok500.attention_mask.numpy(),
                        maps=maps, logits=out.detach().numpy(), P7=P7, dP7=dP7, dP0=dP0)
    print("gradcam_small: maps", maps.shape, "max", maps.max())


def gen_gradcam_large():
    cfg = C.blip_itm_large(336)
    m, itm, tok = _model(cfg, seed=0)
    _, imgs = synth.synth_images(1, 336, seed=1234)
    ids, mask = synth.synth_tokens(cfg, [20], seed=1234)

    class Fixed:                      # tokenizer stand-in returning the synthetic ids
        enc_token_id = cfg.enc_token_id
        pad_token_id = 0

        def __call__(self, caps, padding="longest", max_length=None, **kw):
            from pnp_ovss.tokenizer import Encoding
            L = max_length if padding == "max_length" else int(mask.sum(1).max())
            return Encoding(torch.from_numpy(ids[:, :L]), torch.from_numpy(mask[:, :L]))
    m.tokenizer = Fixed()
    tok500 = m.tokenizer(["x"], padding="max_length", max_length=500)
    args = argparse.Namespace(img_size=336)
    g, _, out = itm.compute_gradcam_ensemble(args, m, torch.from_numpy(imgs), ["x"], tok500)
    layer = m.text_encoder.base_model.base_model.encoder.layer
    P7 = layer[7].crossattention.self.get_attention_map().detach().numpy()
    dP7 = layer[7].crossattention.self.get_attn_gradients().detach().numpy()
    np.savez_compressed(os.path.join(HERE, "gradcam_large.npz"),
                        cfg=json.dumps(cfg.as_dict()), weight_seed=0, image_seed=1234, token_seed=1234,
                        n_classes=20, map_7_9=g[7][9].numpy(), map_7_0=g[7][0].numpy(),
                        map_11_3=g[11][3].numpy(), map_9_3=g[9][3].numpy(), map_10_5=g[10][5].numpy(),
                        logits=out.detach().numpy(), P7_h9=P7[:, 9], dP7_h9=dP7[:, 9])
    print("gradcam_large: map", g[7][9].shape, "max", g[7][9].max().item())


def gen_gradcam_large_768():
    """BASELINE config 5 geometry at full model size: BLIP-ITM-large at img_size 768 (48 x 48 patches, 2305 image tokens),
    B = 1, an ADE20K-sized prompt (40 classes, L = 45): the selected map [7][9], one more (layer, head), logits."""
    cfg = C.blip_itm_large(768)
    m, itm, tok = _model(cfg, seed=0)
    _, imgs = synth.synth_images(1, 768, seed=4321)
    ids, mask = synth.synth_tokens(cfg, [40], seed=4321)

    class Fixed:
        enc_token_id = cfg.enc_token_id
        pad_token_id = 0

        def __call__(self, caps, padding="longest", max_length=None, **kw):
            from pnp_ovss.tokenizer import Encoding
            L = max_length if padding == "max_length" else int(mask.sum(1).max())
            return Encoding(torch.from_numpy(ids[:, :L]), torch.from_numpy(mask[:, :L]))
    m.tokenizer = Fixed()
    tok500 = m.tokenizer(["x"], padding="max_length", max_length=500)
    args = argparse.Namespace(img_size=768)
    g, _, out = itm.compute_gradcam_ensemble(args, m, torch.from_numpy(imgs), ["x"], tok500)
    np.savez_compressed(os.path.join(HERE, "gradcam_large_768.npz"), cfg=json.dumps(cfg.as_dict()), weight_seed=0,
                        image_seed=4321, token_seed=4321, n_classes=40, map_7_9=g[7][9].numpy().astype(np.float32),
                        map_11_3=g[11][3].numpy().astype(np.float32), map_9_3=g[9][3].numpy().astype(np.float32),
                        logits=out.detach().numpy())
    print("gradcam_large_768: map", g[7][9].shape, "max", g[7][9].max().item())


def _driver_ns(itm, extra=None):
    from pathlib import Path
    import scipy.ndimage as filters
    g = dict(torch=torch, np=np, json=json, os=os, Path=Path, filters=filters,
             compute_gradcam_ensemble=itm.compute_gradcam_ensemble)
    if extra:
        g.update(extra)
    return g


def gen_droploop_small():
    cfg = C.blip_itm_small(128)
    m, itm, tok = _model(cfg, seed=4)
    B = 3
    _, imgs = synth.synth_images(B, cfg.img_size, seed=6)
    caps = ["A picture of cat aeroplane dog sheep boat", "A picture of bus tvmonitor",
            "A picture of person"]
    img_ids = ["2007_000033", "2007_000042", "2007_000061"]
    rec = {"zeroed": []}
    real = itm.compute_gradcam_ensemble

    def spy(args, model, visual_input, text_input, tokenized_text, drop_iter=0):
        x = visual_input.detach().numpy()
        P = cfg.grid
        blk = x.reshape(x.shape[0], 3, P, 16, P, 16)
        rec["zeroed"].append((np.abs(blk).sum(axis=(1, 3, 5)) == 0).reshape(x.shape[0], -1))
        return real(args, model, visual_input, text_input, tokenized_text, drop_iter)
    ns = RL.load_driver_functions(["Inference_BLIP_filteredcaption"], _driver_ns(itm))
    ns["compute_gradcam_ensemble"] = spy
    out = {}
    for di in (4, 1):
        rec["zeroed"] = []
        args = argparse.Namespace(img_size=cfg.img_size, drop_iter=di, max_att_block_num=8,
                                  prune_att_head="9", del_patch_num="sort_thresh005")
        tok500 = tok(caps, padding="max_length", max_length=500, return_tensors="pt")
        norm_imgs = torch.zeros(B, cfg.img_size, cfg.img_size, 3)
        g0, agg = ns["Inference_BLIP_filteredcaption"](args, RL.DDPLike(m), tok500, torch.from_numpy(imgs.copy()),
                                                       norm_imgs, img_ids, caps, [c.split()[3:] for c in caps], "cpu")
        out[f"g0_d{di}"] = g0.numpy()
        if agg is not None:
            out[f"agg_d{di}"] = agg.numpy()
        out[f"zeroed_d{di}"] = np.stack(rec["zeroed"])
    tok500 = tok(caps, padding="max_length", max_length=500, return_tensors="pt")
    np.savez_compressed(os.path.join(HERE, "droploop_small.npz"), cfg=json.dumps(cfg.as_dict()),
                        weight_seed=4, image_seed=6, captions=np.array(caps),
                        input_ids=tok500.input_ids.numpy(), attention_mask=tok500.attention_mask.numpy(), **out)
    print("droploop_small:", {k: v.shape for k, v in out.items()})


def _droploop_large(mutate=None):
    """One reference run of Inference_BLIP_filteredcaption (PnP.py:564-722) at BLIP-ITM-large 336^2, B = 2 ragged captions, drop_iter 4;
    `mutate(state_dict, cfg)` edits the seeded weights before they are loaded into the reference modules."""
    cfg = C.blip_itm_large(336)
    tok = SynthTokenizer(cfg.vocab)
    sd = synth.synth_state_dict(cfg, 0)
    if mutate is not None:
        sd = mutate(sd, cfg)
    m, itm = RL.build_reference_model(cfg, sd, tok)
    B, ncls = 2, [20, 12]
    _, imgs = synth.synth_images(B, 336, seed=2024)
    ids, mask = synth.synth_tokens(cfg, ncls, seed=2024)

    class Fixed:                      # tokenizer stand-in returning the synthetic ids (one word-piece per class)
        enc_token_id = cfg.enc_token_id
        pad_token_id = 0

        def __call__(self, caps, padding="longest", max_length=None, **kw):
            from pnp_ovss.tokenizer import Encoding
            L = max_length if padding == "max_length" else int(mask.sum(1).max())
            return Encoding(torch.from_numpy(ids[:, :L].copy()), torch.from_numpy(mask[:, :L].copy()))

        def decode(self, token_ids):                 # PnP.py:658 (word strings of the dead visualisation block)
            return f"t{int(token_ids[0])}"
    m.tokenizer = Fixed()
    rec = {"zeroed": []}
    real = itm.compute_gradcam_ensemble

    def spy(args, model, visual_input, text_input, tokenized_text, drop_iter=0):
        x = visual_input.detach().numpy()
        P = cfg.grid
        blk = x.reshape(x.shape[0], 3, P, 16, P, 16)
        rec["zeroed"].append((np.abs(blk).sum(axis=(1, 3, 5)) == 0).reshape(x.shape[0], -1))
        return real(args, model, visual_input, text_input, tokenized_text, drop_iter)
    ns = RL.load_driver_functions(["Inference_BLIP_filteredcaption"], _driver_ns(itm))
    ns["compute_gradcam_ensemble"] = spy
    args = argparse.Namespace(img_size=336, drop_iter=4, max_att_block_num=8, prune_att_head="9", del_patch_num="sort_thresh005")
    caps = ["x"] * B
    tok500 = m.tokenizer(caps, padding="max_length", max_length=500)
    norm_imgs = torch.zeros(B, 336, 336, 3)
    g0, agg = ns["Inference_BLIP_filteredcaption"](args, RL.DDPLike(m), tok500, torch.from_numpy(imgs.copy()), norm_imgs,
                                                   ["2007_000033", "2007_000042"], caps, [["c"] * n for n in ncls], "cpu")
    return cfg, ncls, g0.numpy().astype(np.float32), agg.numpy().astype(np.float32), np.stack(rec["zeroed"])


def gen_droploop_large():
    """Inference_BLIP_filteredcaption (PnP.py:564-722) at FULL model size: BLIP-ITM-large 336^2, B = 2 with ragged captions
    (20 and 12 classes: the shorter row carries [SEP] and zero pad rows inside the [3:-1] salience slice), drop_iter 4:
    gradcam_0, gradcam_agg and the patches zeroed in front of every iteration (= the picks so far)."""
    cfg, ncls, g0, agg, zeroed = _droploop_large()
    np.savez_compressed(os.path.join(HERE, "droploop_large.npz"), cfg=json.dumps(cfg.as_dict()), weight_seed=0, image_seed=2024,
                        token_seed=2024, n_classes=np.array(ncls), g0=g0, agg=agg, zeroed=zeroed)
    print("droploop_large:", g0.shape, agg.shape, zeroed.sum(axis=(1, 2)))


# heavy-tailed weights (the closest stand-in for a trained checkpoint this container offers): synth.inject_outliers variants of
# the BLIP-large seed; the keyword sets are stored in the fixture so the tests rebuild the same weights
OUTLIER_VARIANTS = {
    "reparam16": dict(gain=16.0, jitter=0.5),                 # function-preserving, per-channel gains 10.7 .. 24 (not powers of two)
    "reparam64": dict(gain=64.0, jitter=0.5, seed=7),        # own channel / gain draws (same seed: reparam16 x 4, bit-identical)
    "massive8": dict(gain=8.0, jitter=0.5, compensate=False),  # consuming weights untouched: outlier channels dominate the dot products
}


def gen_droploop_large_outliers():
    """gen_droploop_large's run with outlier channels injected into the seeded weights (OUTLIER_VARIANTS)."""
    out = {}
    for name, kw in OUTLIER_VARIANTS.items():
        cfg, ncls, g0, agg, zeroed = _droploop_large(lambda sd, cfg_, kw=kw: synth.inject_outliers(sd, cfg_, **kw))
        out[f"{name}_g0"], out[f"{name}_agg"], out[f"{name}_zeroed"] = g0, agg, zeroed
        print("droploop_large_outliers", name, float(np.abs(g0).max()), float(np.abs(agg).max()), zeroed.sum(axis=(1, 2)), flush=True)
    np.savez_compressed(os.path.join(HERE, "droploop_large_outliers.npz"), cfg=json.dumps(cfg.as_dict()), weight_seed=0,
                        image_seed=2024, token_seed=2024, n_classes=np.array(ncls), variants=json.dumps(OUTLIER_VARIANTS), **out)


def gen_merge_tokens():
    tok = SynthTokenizer(1024)
    ns = RL.load_driver_functions(["Mean_over_filtered_label_tokens"], dict(torch=torch, np=np))

    class M:
        tokenizer = tok
    caps = ["A picture of cat pottedplant dog",              # split word in the middle
            "A picture of cat dog tvmonitor",                 # split word last (summed, not averaged)
            "A picture of aeroplane",                         # single split word
            "A picture of cat dog bus",                       # no split: fast path
            "A picture of diningtable pottedplant"]           # two split words in a row
    classes = [c.split()[3:] for c in caps]
    tk = tok(caps, padding="max_length", max_length=500, return_tensors="pt")
    Lmax = int(tk.attention_mask.sum(1).max())
    g = torch.Generator().manual_seed(0)
    maps = torch.rand(len(caps), Lmax - 1, 5, 5, generator=g)
    out = {}
    for i in range(len(caps)):
        r = ns["Mean_over_filtered_label_tokens"](RL.DDPLike(M), tk, maps[i], classes, i)
        out[f"merged_{i}"] = r.numpy()
    pieces = [[tok.decode([t]) for t in tk.input_ids[i][: int(tk.attention_mask[i].sum())]] for i in range(len(caps))]
    np.savez_compressed(os.path.join(HERE, "merge_tokens.npz"), captions=np.array(caps),
                        input_ids=tk.input_ids.numpy(), attention_mask=tk.attention_mask.numpy(),
                        maps=maps.numpy(), pieces=json.dumps(pieces), **out)
    print("merge_tokens:", {k: v.shape for k, v in out.items()})


def _psc_cats():
    import ast as _ast
    tree = _ast.parse(open(RL.REF + "/Load_datasets.py").read())
    for fn in tree.body:
        if isinstance(fn, _ast.FunctionDef) and fn.name == "load_psc":
            for node in _ast.walk(fn):
                if isinstance(node, _ast.Assign) and getattr(node.targets[0], "id", "") == "cats":
                    return _ast.literal_eval(node.value)


VOC_CATS = {1: "aeroplane", 2: "bicycle", 3: "bird", 4: "boat", 5: "bottle", 6: "bus", 7: "car", 8: "cat",
            9: "chair", 10: "cow", 11: "table", 12: "dog", 13: "horse", 14: "motorbike", 15: "person",
            16: "pottedplant", 17: "sheep", 18: "sofa", 19: "train", 20: "tvmonitor"}


def _gen_pipeline(data_type, img_ids, cats, fname, cfg=None, weight_seed=4, image_seed=6,
                  sizes=((90, 120), (128, 128), (75, 100)), gpt_override=None, keep_prepost=True):
    """save_img_union_attention end to end on synthetic images of different original sizes
    (GPT-4o class strings are the reference's shipped JSON for the chosen ids unless `gpt_override` supplies strings of
    the same format for some ids -- then the reference reads a JSON file holding the merged dict from a temp home_dir)."""
    cfg = cfg or C.blip_itm_small(128)
    m, itm, tok = _model(cfg, seed=weight_seed)
    sizes = [tuple(s) for s in sizes]            # H+W > 128: torch generic bilinear kernel
    B = len(sizes)
    assert B == len(img_ids)
    _, imgs = synth.synth_images(B, cfg.img_size, seed=image_seed)
    gpt = json.load(open(RL.REF + f"/GPT4o_classification/{data_type}_classification_noboundary.json"))
    home = RL.REF
    if gpt_override:
        gpt = dict(gpt)
        gpt.update(gpt_override)
        home = tempfile.mkdtemp()
        os.makedirs(os.path.join(home, "GPT4o_classification"))
        with open(os.path.join(home, "GPT4o_classification", f"{data_type}_classification_noboundary.json"), "w") as f:
            json.dump({k: gpt[k] for k in img_ids}, f)
    rng = np.random.default_rng(99)
    org = [rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8) for h, w in sizes]
    gts = [rng.integers(0, len(cats) + 1, size=(h, w)).astype(np.float32) for h, w in sizes]
    nms = list(cats.values())
    names = ["save_img_union_attention", "Inference_BLIP_filteredcaption", "Load_predicted_classes",
             "Mean_over_filtered_label_tokens", "postprocess", "blurring", "Scale_0_1", "_fast_hist", "scores"]
    ns = RL.load_driver_functions(names, _driver_ns(itm))
    ns["load_OrgImage"] = lambda args, ids: org
    ns["Load_GroundTruth"] = lambda args, ids: gts
    out = {}
    real_sc = ns["scores"]
    for pp in ("blur", None):
        rec, labs = [], []
        real_pp = ns["postprocess"]
        if pp:
            def spy(args, pred, org_img_list, label_trues, img, _r=rec, _f=real_pp):
                _r.append(pred.detach().clone().numpy().astype(np.float32))
                return _f(args, pred, org_img_list, label_trues, img)
            ns["postprocess"] = spy

        def spy_scores(lt, lp, cats_, n_class, _l=labs):
            _l.append([np.array(x).astype(np.uint8) for x in lp])     # final (remapped) label maps
            return real_sc(lt, lp, cats_, n_class)
        ns["scores"] = spy_scores
        tmp = tempfile.mkdtemp()
        args = argparse.Namespace(img_size=cfg.img_size, drop_iter=4, max_att_block_num=8, prune_att_head="9",
                                  del_patch_num="sort_thresh005", data_type=data_type, postprocess=pp,
                                  threshold=0.15, home_dir=home, save_path=tmp)
        tok500 = tok(["x"] * B, padding="max_length", max_length=500, return_tensors="pt")
        norm_imgs = torch.zeros(B, cfg.img_size, cfg.img_size, 3)
        with np.errstate(all="ignore"):
            ns["save_img_union_attention"](RL.DDPLike(m), torch.from_numpy(imgs.copy()), None, args, None, img_ids,
                                           4, norm_imgs, None, cats, nms, tok500, "cpu", att_head=9, max_block_num=8)
        ns["postprocess"] = real_pp
        ns["scores"] = real_sc
        tag = pp or "none"
        for d in ("hist_withfiltered_caption", "all_drop_hist_with_filtered_caption"):
            f = os.path.join(tmp, d, f"img_{img_ids[0]}_max_blocknum_8_atthead_9.npy")
            out[f"{d}_{tag}"] = np.load(f)
        for i, r in enumerate(rec if keep_prepost else []):
            out[f"prepost_{tag}_{i}"] = r          # order: 1-drop imgs 0..B-1 then N-drop imgs 0..B-1
        for br, name in enumerate(("1drop", "ndrop")):
            for i in range(B):
                out[f"labels_{name}_{tag}_{i}"] = labs[br][i]
    np.savez_compressed(os.path.join(HERE, fname), cfg=json.dumps(cfg.as_dict()), weight_seed=weight_seed,
                        image_seed=image_seed, img_ids=np.array(img_ids), gpt=json.dumps({k: gpt[k] for k in img_ids}),
                        cats=json.dumps(cats), data_type=data_type, sizes=np.array(sizes), org_seed=99, **out)
    print(fname, {k: getattr(v, "shape", None) for k, v in out.items()})


def gen_pipeline_voc():
    _gen_pipeline("voc", ["2007_008374", "2007_000129", "2007_009419"], VOC_CATS, "pipeline_voc.npz")


def gen_pipeline_voc_large():
    """The HEADLINE geometry end to end through the reference: BLIP-ITM-large 336^2, VOC rules, drop_iter 4, B = 2 --
    image 0 carries the full 20-class prompt of the benchmark (a GPT-4o-format string naming every VOC class at 95 %) at
    the benchmark's 336 x 336 original size, image 1 the reference's shipped string of a real id at a 375 x 500 original.
    Label maps and .npy histograms only (the K x H x W pre-post maps would be 10 MB each)."""
    all20 = "[" + ", ".join(f"{i}: {n}" for i, n in VOC_CATS.items()) + "], [" + ", ".join(["95%"] * 20) + "]"
    _gen_pipeline("voc", ["2007_000033", "2007_000129"], VOC_CATS, "pipeline_voc_large.npz", cfg=C.blip_itm_large(336),
                  weight_seed=0, image_seed=77, sizes=((336, 336), (375, 500)), gpt_override={"2007_000033": all20},
                  keep_prepost=False)


def gen_pipeline_psc():
    _gen_pipeline("psc", ["2008_000009", "2008_000003", "2008_000032"], _psc_cats(), "pipeline_psc.npz")


# COCO category tables (pycocotools loadCats order = ascending id).  Data, not code: ids / names of
# instances_val2017.json + stuff_val2017.json; every (id, name) pair that occurs in the reference's shipped
# GPT4o_classification/coco_*_classification_noboundary.json strings is cross-checked below.
COCO_THINGS = {1: "person", 2: "bicycle", 3: "car", 4: "motorcycle", 5: "airplane", 6: "bus", 7: "train", 8: "truck", 9: "boat",
               10: "traffic light", 11: "fire hydrant", 13: "stop sign", 14: "parking meter", 15: "bench", 16: "bird", 17: "cat",
               18: "dog", 19: "horse", 20: "sheep", 21: "cow", 22: "elephant", 23: "bear", 24: "zebra", 25: "giraffe",
               27: "backpack", 28: "umbrella", 31: "handbag", 32: "tie", 33: "suitcase", 34: "frisbee", 35: "skis",
               36: "snowboard", 37: "sports ball", 38: "kite", 39: "baseball bat", 40: "baseball glove", 41: "skateboard",
               42: "surfboard", 43: "tennis racket", 44: "bottle", 46: "wine glass", 47: "cup", 48: "fork", 49: "knife",
               50: "spoon", 51: "bowl", 52: "banana", 53: "apple", 54: "sandwich", 55: "orange", 56: "broccoli", 57: "carrot",
               58: "hot dog", 59: "pizza", 60: "donut", 61: "cake", 62: "chair", 63: "couch", 64: "potted plant", 65: "bed",
               67: "dining table", 70: "toilet", 72: "tv", 73: "laptop", 74: "mouse", 75: "remote", 76: "keyboard",
               77: "cell phone", 78: "microwave", 79: "oven", 80: "toaster", 81: "sink", 82: "refrigerator", 84: "book",
               85: "clock", 86: "vase", 87: "scissors", 88: "teddy bear", 89: "hair drier", 90: "toothbrush"}
COCO_STUFF = {92: "banner", 93: "blanket", 94: "branch", 95: "bridge", 96: "building-other", 97: "bush", 98: "cabinet", 99: "cage",
              100: "cardboard", 101: "carpet", 102: "ceiling-other", 103: "ceiling-tile", 104: "cloth", 105: "clothes",
              106: "clouds", 107: "counter", 108: "cupboard", 109: "curtain", 110: "desk-stuff", 111: "dirt", 112: "door-stuff",
              113: "fence", 114: "floor-marble", 115: "floor-other", 116: "floor-stone", 117: "floor-tile", 118: "floor-wood",
              119: "flower", 120: "fog", 121: "food-other", 122: "fruit", 123: "furniture-other", 124: "grass", 125: "gravel",
              126: "ground-other", 127: "hill", 128: "house", 129: "leaves", 130: "light", 131: "mat", 132: "metal",
              133: "mirror-stuff", 134: "moss", 135: "mountain", 136: "mud", 137: "napkin", 138: "net", 139: "paper",
              140: "pavement", 141: "pillow", 142: "plant-other", 143: "plastic", 144: "platform", 145: "playingfield",
              146: "railing", 147: "railroad", 148: "river", 149: "road", 150: "rock", 151: "roof", 152: "rug", 153: "salad",
              154: "sand", 155: "sea", 156: "shelf", 157: "sky-other", 158: "skyscraper", 159: "snow", 160: "solid-other",
              161: "stairs", 162: "stone", 163: "straw", 164: "structural-other", 165: "table", 166: "tent",
              167: "textile-other", 168: "towel", 169: "tree", 170: "vegetable", 171: "wall-brick", 172: "wall-concrete",
              173: "wall-other", 174: "wall-panel", 175: "wall-stone", 176: "wall-tile", 177: "wall-wood", 178: "water-other",
              179: "waterdrops", 180: "window-blind", 181: "window-other", 182: "wood", 183: "other"}


def _coco_cats(data_type):
    """cats / nms as the COCO driver's main() builds them (PnP..._coco.py:1381-1400)."""
    things = [{"id": i, "name": n} for i, n in COCO_THINGS.items()]
    stuff = [{"id": i, "name": n} for i, n in COCO_STUFF.items()]
    cats = things + stuff if data_type == "coco_stuff" else things
    nms = [c["name"] for c in cats]
    for i, classname in enumerate(nms):
        nms[i] = "".join("".join(classname.split(" ")).split("-"))
    return cats, nms


def _check_coco_tables():
    import re
    import collections
    for dt, table in (("coco_object", COCO_THINGS), ("coco_stuff", {**COCO_THINGS, **COCO_STUFF})):
        d = json.load(open(RL.REF + f"/GPT4o_classification/{dt}_classification_noboundary.json"))
        seen = collections.defaultdict(collections.Counter)
        for v in d.values():
            for m in re.finditer(r"(\d+):\s*'?([A-Za-z \-]+)'?", v.split("], [")[0]):
                seen[int(m.group(1))][m.group(2).strip()] += 1
        for i, names in seen.items():
            if i in table:
                assert names.most_common(1)[0][0] == table[i], (dt, i, names.most_common(2), table[i])


def _gen_pipeline_coco(data_type, img_ids, fname):
    """The COCO driver's save_img_union_attention end to end on 3 synthetic images (GPT-4o strings: the shipped
    JSON entries of `img_ids`).  drop_iter 4 -> N-drop branch only (with Scale_0_1); drop_iter 2 -> both."""
    _check_coco_tables()
    from PIL import Image
    cfg = C.blip_itm_small(128)
    m, itm, tok = _model(cfg, seed=4)
    B = 3
    _, imgs = synth.synth_images(B, cfg.img_size, seed=6)
    sizes = [(90, 120), (128, 128), (75, 100)]
    rng = np.random.default_rng(99)
    org = [rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8) for h, w in sizes]
    cats, nms = _coco_cats(data_type)
    n_class = 91 if data_type == "coco_object" else 183
    gts = [rng.integers(0, n_class, size=(h, w)).astype(np.float32) for h, w in sizes]
    names = ["save_img_union_attention", "Inference_BLIP_filteredcaption", "Load_predicted_classes",
             "Mean_over_filtered_label_tokens", "postprocess", "blurring", "Scale_0_1", "_fast_hist", "scores",
             "getClassName", "print_iou"]
    S = cfg.img_size
    ns = RL.load_driver_functions(names, _driver_ns(itm, dict(Image=Image, getAttMap=lambda *a, **k: np.zeros((S, S, 3)),
                                                              Draw_Segmentation_map=lambda *a, **k: None)), path=RL.PNPC)
    ns["load_OrgImage"] = lambda args, ids, coco: org
    ns["Load_GroundTruth"] = lambda args, ids, coco: gts
    out = {}
    real_sc = ns["scores"]
    real_pp = ns["postprocess"]
    for di, pp in ((4, "blur"), (4, None), (2, "blur")):
        rec, labs = [], []
        if pp:
            def spy(args, pred, org_img_list, label_trues, img, _r=rec, _f=real_pp):
                _r.append(pred.detach().clone().numpy().astype(np.float32))
                return _f(args, pred, org_img_list, label_trues, img)
            ns["postprocess"] = spy

        def spy_scores(lt, lp, cats_, n_class, _l=labs):
            _l.append([np.array(x).astype(np.uint8) for x in lp])
            return real_sc(lt, lp, cats_, n_class)
        ns["scores"] = spy_scores
        tmp = tempfile.mkdtemp()
        args = argparse.Namespace(img_size=S, drop_iter=di, max_att_block_num=8, prune_att_head="9",
                                  del_patch_num="sort_thresh005", data_type=data_type, postprocess=pp, threshold=0.15,
                                  home_dir=RL.REF, save_path=tmp, in_the_wild=False)
        tok500 = tok(["x"] * B, padding="max_length", max_length=500, return_tensors="pt")
        norm_imgs = torch.zeros(B, S, S, 3)
        with np.errstate(all="ignore"):
            ns["save_img_union_attention"](None, RL.DDPLike(m), torch.from_numpy(imgs.copy()), None, args, [[]] * B, img_ids, di,
                                           norm_imgs, None, cats, nms, tok500, "cpu", att_head=9, max_block_num=8)
        ns["postprocess"] = real_pp
        ns["scores"] = real_sc
        tag = f"d{di}_{pp or 'none'}"
        branches = ("1drop", "ndrop") if di < 3 else ("ndrop",)
        dirs = {"1drop": "hist_withfiltered_caption", "ndrop": "all_drop_hist_with_filtered_caption"}
        for br, name in enumerate(branches):
            out[f"hist_{name}_{tag}"] = np.load(os.path.join(tmp, dirs[name], f"img_{img_ids[0]}_max_blocknum_8_atthead_9.npy"))
            for i in range(B):
                out[f"labels_{name}_{tag}_{i}"] = labs[br][i]
                if pp:
                    out[f"prepost_{name}_{tag}_{i}"] = rec[br * B + i]
        if di >= 3:
            assert not os.path.exists(os.path.join(tmp, dirs["1drop"], f"img_{img_ids[0]}_max_blocknum_8_atthead_9.npy"))
    gpt = json.load(open(RL.REF + f"/GPT4o_classification/{data_type}_classification_noboundary.json"))
    keys = [str(int(i)).rjust(12, "0") for i in img_ids]
    np.savez_compressed(os.path.join(HERE, fname), cfg=json.dumps(cfg.as_dict()), weight_seed=4, image_seed=6,
                        img_ids=np.array(img_ids), gpt=json.dumps({k: gpt[k] for k in keys}), cats=json.dumps(cats),
                        data_type=data_type, sizes=np.array(sizes), org_seed=99, n_class=n_class, **out)
    print(fname, {k: getattr(v, "shape", None) for k, v in out.items()})


def gen_pipeline_coco_object():
    _gen_pipeline_coco("coco_object", [397133, 252219, 37777], "pipeline_coco_object.npz")


def gen_pipeline_coco_stuff():
    _gen_pipeline_coco("coco_stuff", [397133, 502136, 37777], "pipeline_coco_stuff.npz")


def gen_gpt_parse_coco():
    ns = RL.load_driver_functions(["Load_predicted_classes"], dict(json=json), path=RL.PNPC)
    res = {}
    for dt in ("coco_object", "coco_stuff"):
        cats, nms = _coco_cats(dt)
        data = json.load(open(RL.REF + f"/GPT4o_classification/{dt}_classification_noboundary.json"))
        keys = sorted(data.keys())
        pick = keys[::max(1, len(keys) // 50)][:50]
        odd = [k for k in keys if ("\n" in data[k] or data[k] == "" or "], [" not in data[k] or "%" not in data[k])][:20]
        exp = {}
        for k in sorted(set(pick + odd)):
            args = argparse.Namespace(home_dir=RL.REF, data_type=dt)
            try:
                b, c, cap = ns["Load_predicted_classes"](args, nms, cats, [], [], [], [[]], [int(k)], 0, pred_path=None)
                exp[k] = {"raw": data[k], "best_class_idx": b[0], "classes": c[0], "caption": cap[0]}
            except Exception as e:          # noqa: BLE001  (the reference raises on some malformed strings)
                exp[k] = {"raw": data[k], "error": type(e).__name__}
        res[dt] = {"cats": cats, "cases": exp}
    json.dump(res, open(os.path.join(HERE, "gpt_parse_coco.json"), "w"), indent=0)
    print("gpt_parse_coco:", {k: (len(v["cases"]), sum("error" in c for c in v["cases"].values())) for k, v in res.items()})


def _ref_base_model_functions():
    """interpolate_pos_embed and BaseModel.load_checkpoint exec'd from the reference's base_model.py (the module itself
    needs omegaconf / LAVIS dist utils, the two functions need only torch / os / logging)."""
    import ast as _ast
    import logging
    path = RL.REF_B + "base_model.py"
    tree = _ast.parse(open(path).read())
    ns = dict(torch=torch, os=os, logging=logging, is_url=lambda p: False, download_cached_file=None)
    for node in tree.body:
        if isinstance(node, _ast.FunctionDef) and node.name == "interpolate_pos_embed":
            exec(compile(_ast.Module(body=[node], type_ignores=[]), path, "exec"), ns)
        if isinstance(node, _ast.ClassDef) and node.name == "BaseModel":
            for sub in node.body:
                if isinstance(sub, _ast.FunctionDef) and sub.name == "load_checkpoint":
                    exec(compile(_ast.Module(body=[sub], type_ignores=[]), path, "exec"), ns)
    return ns["interpolate_pos_embed"], ns["load_checkpoint"]


def gen_pos_embed():
    """interpolate_pos_embed (base_model.py:44-73) for the two re-tilings the configs need: the 384-px flickr
    checkpoint's 24 x 24 grid -> 21 x 21 (336 px) and -> 48 x 48 (768 px, BASELINE config 5)."""
    interp, _ = _ref_base_model_functions()
    rng = np.random.default_rng(21)
    D = 24
    pos = rng.standard_normal((1, 1 + 24 * 24, D)).astype(np.float32)
    out = {"pos_24": pos}
    for g in (21, 48, 24):
        class V:
            pass
        v = V()
        v.patch_embed = V()
        v.patch_embed.num_patches = g * g
        v.pos_embed = torch.zeros(1, 1 + g * g, D)
        out[f"pos_{g}_from_24"] = interp(torch.from_numpy(pos.copy()), v).numpy()
    np.savez_compressed(os.path.join(HERE, "pos_embed_cases.npz"), **out)
    print("pos_embed_cases:", {k: v.shape for k, v in out.items()})


def gen_checkpoint_small():
    """BaseModel.load_checkpoint (base_model.py:86-125) on the reference model: seeded init (seed 3) -> load a synthetic
    checkpoint saved at grid 8 (128 px) into the 64-px model -> compute_gradcam_ensemble.  Golden: resized pos_embed,
    the surviving itm_head.bias (mismatched key dropped -> init value kept), maps [7][9], logits."""
    _, load_checkpoint = _ref_base_model_functions()
    cfg = C.blip_itm_small(64)
    cfg_ck = C.blip_itm_small(128)
    m, itm, tok = _model(cfg, seed=3)
    ck = synth.synth_checkpoint(cfg, cfg_ck, 7)
    tmp = tempfile.mkdtemp()
    path = os.path.join(tmp, "ckpt.pth")
    torch.save({"model": {k: torch.from_numpy(v.copy()) for k, v in ck.items()}}, path)
    msg = load_checkpoint(m, path)
    _, imgs = synth.synth_images(2, cfg.img_size, seed=5)
    caps = ["A picture of cat aeroplane dog sheep boat", "A picture of bus tvmonitor"]
    tok500 = tok(caps, padding="max_length", max_length=500, return_tensors="pt")
    args = argparse.Namespace(img_size=cfg.img_size)
    g, _, out = itm.compute_gradcam_ensemble(args, m, torch.from_numpy(imgs), caps, tok500)
    sd = m.state_dict()
    np.savez_compressed(os.path.join(HERE, "checkpoint_small.npz"), cfg=json.dumps(cfg.as_dict()),
                        cfg_ckpt=json.dumps(cfg_ck.as_dict()), init_seed=3, ckpt_seed=7, image_seed=5, captions=np.array(caps),
                        pos_embed=sd["visual_encoder.pos_embed"].numpy(), itm_head_bias=sd["itm_head.bias"].numpy(),
                        missing=np.array(sorted(msg.missing_keys)), unexpected=np.array(sorted(msg.unexpected_keys)),
                        map_7_9=g[7][9].numpy(), logits=out.detach().numpy())
    print("checkpoint_small: missing", msg.missing_keys, "unexpected", msg.unexpected_keys)


def gen_tokenizer():
    """LAVIS BlipBase.init_tokenizer (blip_image_text_matching.py:42): HF BertTokenizer +
    add_special_tokens bos "[DEC]" + additional "[ENC]", run on a small committed vocabulary laid out like
    bert-base-uncased's (specials at 0 / 100-103).  Golden: ids / masks for the two call forms of the drivers
    (padding="max_length" PnP.py:317 and padding="longest", truncation blip_image_text_matching.py:230-236) and the
    per-id decode strings the merge walk reads (PnP.py:812-819)."""
    from transformers import BertTokenizer
    from pnp_ovss import datasets as DS
    words = ("a picture of aero ##plane bi ##cycle bird boat bottle bus car cat chair cow table dog horse motor ##bike person "
             "potted ##plant sheep sofa train tv ##mon ##itor wall sky floor tree ceiling road bed window ##pane grass cabinet "
             "side ##walk ground door mountain plant curtain water painting shelf house sea mirror rug field arm ##chair seat "
             "fence desk rock ward ##robe lamp bath ##tub rail ##ing cushion base box pillar sign ##board chest ##of ##draw ##ers "
             "traffic ##light fire ##hy ##dran ##t stop parking meter bench ele ##phant bear zebra gi ##raf ##fe back ##pack "
             "umbrella hand ##bag tie suit ##case fr ##is ##bee ski ##s snow sports ##ball kite base ##ball bat glove skate "
             "surf tennis racket wine ##glass cup fork knife spoon bowl banana apple sand ##wich orange bro ##cco ##li carrot "
             "hot ##dog pizza don ##ut cake couch dining toilet laptop mouse remote key cell ##phone micro ##wave oven toast "
             "##er sink re ##fr ##iger ##ator book clock vase sci ##sso ##rs teddy hair ##dr ##ier tooth ##brush it").split()
    vocab = ["[PAD]"] + [f"[unused{i}]" for i in range(1, 100)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"]
    vocab += list("abcdefghijklmnopqrstuvwxyz") + ["##" + ch for ch in "abcdefghijklmnopqrstuvwxyz"] + list("0123456789")
    vocab += list(",.-'!?()/")
    for w in words:
        if w not in vocab:
            vocab.append(w)
    open(os.path.join(HERE, "tiny_vocab.txt"), "w").write("\n".join(vocab) + "\n")
    t = BertTokenizer(vocab={w: i for i, w in enumerate(vocab)})
    t.add_special_tokens({"bos_token": "[DEC]"})
    t.add_special_tokens({"additional_special_tokens": ["[ENC]"]})
    coco = ["".join("".join(n.split(" ")).split("-")) for n in list(COCO_THINGS.values()) + list(COCO_STUFF.values())]
    ade = ["".join(n.split(" ")) for n in DS.ADE_NAMES]
    caps = ["A picture of " + " ".join(VOC_CATS.values()),
            "A picture of " + " ".join(DS.PSC_NAMES[:30]), "A picture of " + " ".join(DS.PSC_NAMES[30:]),
            "A picture of " + " ".join(ade[:40]), "A picture of " + " ".join(ade[100:150]),
            "A picture of " + " ".join(coco[:45]), "A picture of " + " ".join(coco[80:130]),
            "A picture of cat", "A picture of pottedplant tvmonitor",
            "A Picture of  Dog-x, it's caf\u00e9 time!  (sofa/tv) 12 stra\u00dfe", "", "a " + "x" * 120 + " b"]
    out = {"captions": caps, "vocab_size": len(t), "enc_token_id": t.convert_tokens_to_ids("[ENC]"),
           "dec_token_id": t.convert_tokens_to_ids("[DEC]")}
    e = t(caps, padding="max_length", max_length=500, return_tensors="pt")
    out["max_length_500"] = {"input_ids": [r[: int(m.sum())].tolist() for r, m in zip(e.input_ids, e.attention_mask)]}
    e2 = t(caps, padding="longest", truncation=True, max_length=500, return_tensors="pt")
    out["longest"] = {"shape": list(e2.input_ids.shape), "input_ids": e2.input_ids.tolist(), "attention_mask": e2.attention_mask.tolist()}
    e3 = t(caps[:3], padding="longest", truncation=True, max_length=16, return_tensors="pt")
    out["truncate_16"] = {"input_ids": e3.input_ids.tolist(), "attention_mask": e3.attention_mask.tolist()}
    ids = sorted({i for r in out["max_length_500"]["input_ids"] for i in r} | {0, 100, out["enc_token_id"]})
    out["decode"] = {str(i): t.decode([i]) for i in ids}
    json.dump(out, open(os.path.join(HERE, "tokenizer_cases.json"), "w"), indent=0)
    print("tokenizer_cases:", len(caps), "captions, vocab", len(t), "longest", out["longest"]["shape"])


def gen_gpt_parse():
    ns = RL.load_driver_functions(["Load_predicted_classes"], dict(json=json))
    res = {}
    root = RL.REF + "/GPT4o_classification/"
    tables = {
        "voc": ["aeroplane", "bicycle", "bird", "boat", "bottle", "bus", "car", "cat", "chair", "cow", "table", "dog",
                "horse", "motorbike", "person", "pottedplant", "sheep", "sofa", "train", "tvmonitor"],
    }
    # nms tables for psc / ade20k come from the reference loaders (Load_datasets.py) at generation time
    import ast as _ast
    src = open(RL.REF + "/Load_datasets.py").read()
    tree = _ast.parse(src)
    for fn in tree.body:
        if isinstance(fn, _ast.FunctionDef) and fn.name == "load_psc":
            for node in _ast.walk(fn):
                if isinstance(node, _ast.Assign) and getattr(node.targets[0], "id", "") == "cats":
                    tables["psc"] = list(_ast.literal_eval(node.value).values())
    for dt in ("voc", "psc"):
        data = json.load(open(f"{root}{dt}_classification_noboundary.json"))
        keys = sorted(data.keys())
        step = max(1, len(keys) // 60)
        pick = keys[::step][:60]
        # add odd-looking strings explicitly
        odd = [k for k in keys if ("\n" in data[k] or data[k] == "" or "%" not in data[k])][:15]
        pick = sorted(set(pick + odd))
        nms = tables[dt]
        exp = {}
        for k in pick:
            args = argparse.Namespace(home_dir=RL.REF, data_type=dt)
            try:
                b, c, cap = ns["Load_predicted_classes"](args, nms, [], [], [], None, [k], 0, pred_path=None)
                exp[k] = {"raw": data[k], "best_class_idx": b[0], "classes": c[0], "caption": cap[0]}
            except Exception as e:          # the reference raises on some malformed strings: record that
                exp[k] = {"raw": data[k], "error": type(e).__name__}
        res[dt] = {"nms": nms, "cases": exp}
    json.dump(res, open(os.path.join(HERE, "gpt_parse.json"), "w"), indent=0)
    print("gpt_parse:", {k: len(v["cases"]) for k, v in res.items()})


def gen_blur_cases():
    import scipy.ndimage as filters
    ns = RL.load_driver_functions(["blurring"], dict(filters=filters, np=np, torch=torch))
    rng = np.random.default_rng(5)
    out = {}
    for i, (h, w) in enumerate([(64, 64), (75, 100), (33, 21), (336, 336)]):
        x = rng.random((h, w), dtype=np.float32)
        x[x < 0.6] = 0
        out[f"in_{i}"] = x
        out[f"out_{i}"] = ns["blurring"](torch.from_numpy(x), (h, w), scale=0.05)
    z = np.zeros((40, 30), dtype=np.float32)
    out["in_nan"] = z
    with np.errstate(all="ignore"):
        out["out_nan"] = ns["blurring"](torch.from_numpy(z), (40, 30), scale=0.05)
    np.savez_compressed(os.path.join(HERE, "blur_cases.npz"), **out)
    print("blur_cases ok; nan case all-nan:", bool(np.isnan(out["out_nan"]).all()))


def gen_hist_cases():
    ns = RL.load_driver_functions(["_fast_hist", "scores"], dict(np=np))
    rng = np.random.default_rng(8)
    cats = {i: f"c{i}" for i in range(1, 21)}
    lt = [rng.integers(0, 21, size=(30, 40)).astype(np.float32), rng.integers(0, 21, size=(17, 9)).astype(np.float32)]
    lp = [rng.integers(0, 21, size=(30, 40)).astype(np.float32), rng.integers(0, 21, size=(17, 9)).astype(np.float32)]
    lt[1][0, :] = 255            # out-of-range gt is ignored by the mask
    with np.errstate(all="ignore"):
        acc, hist = ns["scores"](lt, lp, cats, n_class=21)
    np.savez_compressed(os.path.join(HERE, "hist_cases.npz"), lt0=lt[0], lt1=lt[1], lp0=lp[0], lp1=lp[1], hist=hist,
                        miou=acc["Mean IoU"], pixacc=acc["Pixel Accuracy"], fwiou=acc["Frequency Weighted IoU"],
                        macc=acc["Mean Accuracy"])
    print("hist_cases: mIoU", acc["Mean IoU"])


def gen_preprocess_cases():
    """Dataset.py:434-443 on synthetic RGB images: Pillow bicubic resize (the library call behind
    transforms.Resize on a PIL image) -> /255 -> (x - mean) / std in float32, as torchvision's ToTensor /
    Normalize compute them.  Inputs are small; outputs are stored as the resized uint8 image (the float tensor is
    a deterministic float32 function of it, re-derived in the tests)."""
    from PIL import Image
    rng = np.random.default_rng(11)
    out = {}
    cases = [(75, 100, 64), (100, 67, 64), (24, 32, 48), (64, 64, 64), (57, 100, 96), (160, 107, 48), (17, 19, 16)]
    for i, (H, W, S) in enumerate(cases):
        coarse = rng.integers(0, 256, size=((H + 7) // 8, (W + 7) // 8, 3))
        img = np.clip(np.repeat(np.repeat(coarse, 8, 0), 8, 1)[:H, :W] + rng.integers(-20, 21, size=(H, W, 3)), 0, 255).astype(np.uint8)
        res = np.asarray(Image.fromarray(img).resize((S, S), Image.BICUBIC))
        out[f"res_bilinear{i}"] = np.asarray(Image.fromarray(img).resize((S, S), Image.BILINEAR))    # ADE20K: Dataset.py:1263
        t = torch.from_numpy(res.copy()).permute(2, 0, 1).float().div(255)
        mean = torch.tensor([0.48145466, 0.4578275, 0.40821073]).view(3, 1, 1)
        std = torch.tensor([0.26862954, 0.26130258, 0.27577711]).view(3, 1, 1)
        t = (t - mean) / std
        out[f"img{i}"], out[f"res{i}"], out[f"S{i}"] = img, res, np.int32(S)
        out[f"tensor_sha{i}"] = np.frombuffer(__import__("hashlib").sha256(t.numpy().tobytes()).digest(), dtype=np.uint8)
    out["n"] = np.int32(len(cases))
    import PIL
    out["pillow_version"] = np.array(PIL.__version__)
    np.savez_compressed(os.path.join(HERE, "preprocess_cases.npz"), **out)
    print("preprocess_cases:", len(cases), "cases, Pillow", PIL.__version__)


GENS = dict(pipeline_voc_large=gen_pipeline_voc_large, gradcam_small=gen_gradcam_small, gradcam_large=gen_gradcam_large, gradcam_large_768=gen_gradcam_large_768, droploop_small=gen_droploop_small,
            merge_tokens=gen_merge_tokens, pipeline_voc=gen_pipeline_voc, pipeline_psc=gen_pipeline_psc,
            pipeline_coco_object=gen_pipeline_coco_object, pipeline_coco_stuff=gen_pipeline_coco_stuff,
            gpt_parse_coco=gen_gpt_parse_coco, gpt_parse=gen_gpt_parse, tokenizer=gen_tokenizer,
            pos_embed=gen_pos_embed, checkpoint_small=gen_checkpoint_small,
            droploop_large=gen_droploop_large, droploop_large_outliers=gen_droploop_large_outliers, blur_cases=gen_blur_cases, hist_cases=gen_hist_cases, preprocess_cases=gen_preprocess_cases)

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--skip-large", action="store_true")
    a = ap.parse_args()
    for name, fn in GENS.items():
        if a.only and name != a.only:
            continue
        if a.skip_large and name in ("gradcam_large", "gradcam_large_768", "droploop_large", "droploop_large_outliers", "pipeline_voc_large"):
            continue
        fn()
