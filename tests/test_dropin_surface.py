"""GPU: the reference's Python surface (lavis import paths, compute_gradcam_ensemble return shape,
hook accessors, CLI flags + .npy outputs) served by the HIP engine."""
import argparse
import glob
import json
import os
import subprocess
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def test_compute_gradcam_ensemble_surface_matches_reference_golden(golden_dir):
    from pnp_ovss import config as C, synth
    from pnp_ovss.model import build_model
    from lavis.models.blip_models.blip_image_text_matching import compute_gradcam_ensemble
    g = np.load(os.path.join(golden_dir, "gradcam_small.npz"))
    cfg = C.ModelCfg(**json.loads(str(g["cfg"])))
    model = build_model(cfg=cfg, max_batch=2, max_text_len=32, stash_layer=7, bf16=False, seed=int(g["weight_seed"]))
    _, imgs = synth.synth_images(2, cfg.img_size, seed=int(g["image_seed"]))
    caps = [str(c) for c in g["captions"]]
    tok500 = model.module.tokenizer(caps, padding="max_length", max_length=500, return_tensors="pt")
    np.testing.assert_array_equal(tok500.input_ids.numpy(), g["input_ids"])          # same synthetic tokenizer
    for blk in range(7, 9):                                                          # the driver toggles these (PnP.py:294-298)
        model.module.text_encoder.base_model.base_model.encoder.layer[blk].crossattention.self.save_attention = False
    args = argparse.Namespace(img_size=cfg.img_size)
    blocks, cams, logits = compute_gradcam_ensemble(args, model.module, torch.from_numpy(imgs), caps, tok500)
    m = blocks[7][9]
    assert cams == [] and m.device.type == "cpu" and m.dtype == torch.float32 and len(blocks) == 12 and len(blocks[7]) == 12
    assert np.abs(m.numpy() - g["maps"][7, 9]).max() < 1e-4
    np.testing.assert_allclose(logits.cpu().numpy(), g["logits"], atol=5e-3)
    hook = model.module.text_encoder.base_model.base_model.encoder.layer[7].crossattention.self
    np.testing.assert_allclose(hook.get_attention_map().cpu().numpy(), g["P7"], atol=2e-5)
    np.testing.assert_allclose(hook.get_attn_gradients().cpu().numpy(), g["dP7"], atol=3e-4)
    with pytest.raises(RuntimeError):
        blocks[3][0]
    model.engine.close()


def test_reference_driver_literal_call_sequence(golden_dir, tmp_path, monkeypatch):
    """The reference driver's own lines, verbatim, through the import shim: PnP.py:1212-1213 (load_model_and_preprocess with
    the reference's four arguments and nothing else), :1218 (DDP wrapper), :271 (tokenizer(...).to(rank)), :294-298 (hook
    flags), :567-575 (compute_gradcam_ensemble under torch.inference_mode(), [layer][head].detach().clone()).  The model must
    come up in the parity mode (fp32 arithmetic), take its geometry from args.img_size and its kept layer from
    args.max_att_block_num, and reproduce the reference's own map (gradcam_small.npz).  The model geometry is the "yaml" of
    this build: PNP_OVSS_MODEL_CONFIG names the reduced configuration the golden vectors were made with."""
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    from pnp_ovss import config as C, synth
    g = np.load(os.path.join(golden_dir, "gradcam_small.npz"))
    cfgd = json.loads(str(g["cfg"]))
    cfg = C.ModelCfg(**cfgd)
    yaml = dict(cfgd, weight_seed=int(g["weight_seed"]))
    (tmp_path / "model.json").write_text(json.dumps(yaml))
    monkeypatch.setenv("PNP_OVSS_MODEL_CONFIG", str(tmp_path / "model.json"))
    for k in ("PNP_OVSS_DTYPE", "PNP_OVSS_CHECKPOINT", "PNP_OVSS_VOCAB", "PNP_OVSS_STASH_LAYER"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", "29533")
    monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    _, imgs = synth.synth_images(2, cfg.img_size, seed=int(g["image_seed"]))
    imgs_in = torch.from_numpy(imgs)
    caption_filtered_list = [str(c) for c in g["captions"]]
    args = argparse.Namespace(img_size=cfg.img_size, max_att_block_num=8, prune_att_head="9", drop_iter=1, batch_size=2)
    rank = 0
    dist.init_process_group("nccl", rank=rank, world_size=1)                   # ddp_setup, PnP.py:45-54
    try:
        torch.cuda.set_device(rank)
        from lavis.models import load_model_and_preprocess
        from lavis.models.blip_models.blip_image_text_matching import compute_gradcam_ensemble
        model_textloc, vis_processors_textloc, text_processors_textloc = load_model_and_preprocess(
            "blip_image_text_matching", "large", device=rank, is_eval=True)
        assert model_textloc._engine is None                                   # nothing sized yet: the first call decides
        model_textloc = DDP(model_textloc, device_ids=[rank])
        txt_tokens = model_textloc.module.tokenizer(caption_filtered_list, padding="max_length", max_length=500,
                                                    return_tensors="pt").to(rank)
        np.testing.assert_array_equal(txt_tokens.input_ids.cpu().numpy(), g["input_ids"])
        for block_num in range(7, 9):
            model_textloc.module.text_encoder.base_model.base_model.encoder.layer[
                block_num
            ].crossattention.self.save_attention = False
        with torch.inference_mode():
            gradcam_filterd_ensemble, cam_filterd_ensemble, filterd_output = compute_gradcam_ensemble(args,
                                                                                                      model_textloc.module,
                                                                                                      imgs_in.to(rank),
                                                                                                      caption_filtered_list,
                                                                                                      txt_tokens)
            layer = int(args.max_att_block_num) - 1
            head = int(args.prune_att_head)
            gradcam_0_filtered = gradcam_filterd_ensemble[layer][head].detach().clone()
        eng = model_textloc.module.engine
        assert eng.mode == "f32" and eng.cfg.img_size == cfg.img_size and eng.stash_layer == 7 and eng.max_batch == 2
        assert cam_filterd_ensemble == [] and gradcam_0_filtered.device.type == "cpu"
        assert np.abs(gradcam_0_filtered.numpy() - g["maps"][7, 9]).max() < 1e-4
        np.testing.assert_allclose(filterd_output.cpu().numpy(), g["logits"], atol=5e-3)
        # a second call with a larger batch / another layer re-creates the engine instead of failing or silently truncating
        args2 = argparse.Namespace(img_size=cfg.img_size, max_att_block_num=4, prune_att_head="2", drop_iter=1, batch_size=2)
        with pytest.warns(UserWarning, match="re-creating"):
            ens2, _, _ = compute_gradcam_ensemble(args2, model_textloc.module, imgs_in.to(rank), caption_filtered_list, txt_tokens)
        assert np.abs(ens2[3][2].numpy() - g["maps"][3, 2]).max() < 1e-4
        model_textloc.module.engine.close()
    finally:
        dist.destroy_process_group()


def test_cli_synthetic_end_to_end(tmp_path):
    save = tmp_path / "out"
    cmd = [sys.executable, os.path.join(ROOT, "pnp-ovss_amd", "PnP_OVSS_0514_updated_segmentation.py"),
           "--save_path", str(save), "--world_size", "1", "--img_size", "336", "--del_patch_num", "sort_thresh005",
           "--batch_size", "3", "--max_att_block_num", "8", "--drop_iter", "2", "--prune_att_head", "9",
           "--sort_threshold", "0.05", "--threshold", "0.15", "--postprocess", "blur+crf", "--data_type", "synthetic",
           "--synthetic_images", "6"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    files = sorted(glob.glob(str(save / "all_drop_hist_with_filtered_caption" / "*_max_blocknum_8_atthead_9.npy")))
    assert len(files) == 2 and len(glob.glob(str(save / "hist_withfiltered_caption" / "*.npy"))) == 2
    total = sum(np.load(f) for f in files)
    assert total.shape == (21, 21) and total.sum() == 6 * 336 * 336
    summary = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert summary["images"] == 6 and 0.0 <= summary["Mean IoU"] <= 1.0
    r = subprocess.run([sys.executable, os.path.join(ROOT, "pnp-ovss_amd", "Calculate_mIoU.py"), "--save_path", str(save)],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "Mean IoU" in r.stdout


def test_cli_pipelines_write_the_same_files(tmp_path):
    """--pipelines 2 (two model replicas taking the batches on their own streams / host threads) must leave exactly the
    files a one-replica run leaves: same names, same confusion matrices."""
    outs = []
    for P in (1, 2):
        save = tmp_path / f"out{P}"
        cmd = [sys.executable, os.path.join(ROOT, "pnp-ovss_amd", "PnP_OVSS_0514_updated_segmentation.py"),
               "--save_path", str(save), "--world_size", "1", "--img_size", "336", "--del_patch_num", "sort_thresh005",
               "--batch_size", "2", "--max_att_block_num", "8", "--drop_iter", "2", "--prune_att_head", "9",
               "--sort_threshold", "0.05", "--threshold", "0.15", "--postprocess", "blur+crf", "--data_type", "synthetic",
               "--synthetic_images", "8", "--pipelines", str(P)]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-3000:]
        files = {}
        for d in ("hist_withfiltered_caption", "all_drop_hist_with_filtered_caption"):
            for f in sorted(glob.glob(str(save / d / "*.npy"))):
                files[d + "/" + os.path.basename(f)] = np.load(f)
        outs.append((files, json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])))
    (f1, s1), (f2, s2) = outs
    assert len(f1) == 8 and sorted(f1) == sorted(f2)
    for k in f1:
        np.testing.assert_array_equal(f1[k], f2[k])
    assert s1["images"] == s2["images"] == 8 and s1["Mean IoU"] == s2["Mean IoU"]


def test_voc_dataset_device_preprocess_equals_host_pillow_path(tmp_path):
    """`--data_type voc` input side on a tiny fake VOCdevkit: the batch tensor produced by the device resize +
    normalise equals the reference's host recipe (PIL bicubic resize -> /255 -> (x - mean) / std, Dataset.py:434-443)
    byte for byte; ids / RGB / ground truth come through as the driver expects (PnP.py:901-955)."""
    import types
    from PIL import Image
    from pnp_ovss import datasets, synth
    root = tmp_path / "VOCdevkit" / "VOC2012"
    for d in ("JPEGImages", "SegmentationClass", "ImageSets/Segmentation"):
        (root / d).mkdir(parents=True)
    (tmp_path / "GPT4o_classification").mkdir()
    rng = np.random.default_rng(3)
    ids, table = [], {}
    for i, (h, w) in enumerate(((75, 100), (100, 67), (64, 64))):
        name = f"2008_{i:06d}"
        Image.fromarray(rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)).save(root / "JPEGImages" / f"{name}.jpg", quality=95)
        gt = rng.integers(0, 21, size=(h, w)).astype(np.uint8)
        gt[0, 0] = 255
        Image.fromarray(gt).save(root / "SegmentationClass" / f"{name}.png")
        ids.append(name)
        table[name] = "[1: 'aeroplane', 12: 'dog'], [95%, 80%]"
    (root / "ImageSets/Segmentation/val.txt").write_text("\n".join(ids) + "\n")
    (tmp_path / "GPT4o_classification" / "voc_classification_noboundary.json").write_text(json.dumps(table))
    args = types.SimpleNamespace(home_dir=str(tmp_path), img_size=64, data_type="voc")
    ds = datasets.make_dataset(args, 0, 1)
    batches = list(ds.batches(2))
    assert sum(len(b["img_ids"]) for b in batches) == 3
    mean = np.array(synth.CLIP_MEAN, dtype=np.float32).reshape(3, 1, 1)
    std = np.array(synth.CLIP_STD, dtype=np.float32).reshape(3, 1, 1)
    for b in batches:
        got = b["imgs"].cpu().numpy()
        for j, name in enumerate(b["img_ids"]):
            img = Image.open(root / "JPEGImages" / f"{name}.jpg").convert("RGB")
            x = np.asarray(img.resize((64, 64), Image.BICUBIC), dtype=np.float32).transpose(2, 0, 1) / np.float32(255.0)
            assert np.array_equal(got[j], (x - mean) / std), name
            org = b["org_images"][j]
            assert isinstance(org, torch.Tensor) and org.is_cuda              # decoded on the device (hip.jpeg_decode_batch)
            assert np.array_equal(org.cpu().numpy(), np.asarray(img))
            assert b["label_trues"][j].dtype == np.float32 and b["label_trues"][j][0, 0] == 0      # 255 -> 0 (PnP.py:908)


def test_ade20k_dataset_layout_and_device_preprocess(tmp_path):
    """`--data_type ade20k` on a tiny fake tree laid out as the reference expects (validation.odgt list,
    ADEChallengeData2016/{images,annotations}/validation/ADE_val_%08d, GPT table keyed the same way): ids are the
    zero-stripped number (Dataset.py:1270), the tensor is PIL bilinear resize + ToTensor only (Dataset.py:1263-1275),
    class names lose their blanks (Load_datasets.py:87)."""
    import types
    from PIL import Image
    from pnp_ovss import datasets
    (tmp_path / "semantic-segmentation-pytorch-master/data").mkdir(parents=True)
    (tmp_path / "ADEChallengeData2016/images/validation").mkdir(parents=True)
    (tmp_path / "ADEChallengeData2016/annotations/validation").mkdir(parents=True)
    (tmp_path / "GPT4o_classification").mkdir()
    rng = np.random.default_rng(4)
    recs, table = [], {}
    for num, (h, w) in ((7, (60, 80)), (123, (90, 64))):
        stem = f"ADE_val_{num:08d}"
        Image.fromarray(rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)).save(tmp_path / f"ADEChallengeData2016/images/validation/{stem}.jpg", quality=95)
        Image.fromarray(rng.integers(0, 151, size=(h, w)).astype(np.uint8)).save(tmp_path / f"ADEChallengeData2016/annotations/validation/{stem}.png")
        recs.append({"fpath_img": f"ADEChallengeData2016/images/validation/{stem}.jpg",
                     "fpath_segm": f"ADEChallengeData2016/annotations/validation/{stem}.png", "width": w, "height": h})
        table[stem] = "[45: 'chest of drawers', 3: 'sky', 1: 'wall'], [90%, 85%, 40%]"
    (tmp_path / "semantic-segmentation-pytorch-master/data/validation.odgt").write_text("\n".join(json.dumps(r) for r in recs) + "\n")
    (tmp_path / "GPT4o_classification/ade20k_classification_noboundary.json").write_text(json.dumps(table))
    args = types.SimpleNamespace(home_dir=str(tmp_path), img_size=48, data_type="ade20k")
    ds = datasets.make_dataset(args, 0, 1)
    assert len(ds.cats) == 150 and ds.nms[44] == "chestofdrawers" and ds.total_hist.shape == (151, 151)
    (b,) = list(ds.batches(4))
    assert sorted(b["img_ids"]) == ["123", "7"]
    got = b["imgs"].cpu().numpy()
    for j, img_id in enumerate(b["img_ids"]):
        stem = "ADE_val_" + img_id.rjust(8, "0")
        img = Image.open(tmp_path / f"ADEChallengeData2016/images/validation/{stem}.jpg").convert("RGB")
        x = np.asarray(img.resize((48, 48), Image.BILINEAR), dtype=np.float32).transpose(2, 0, 1) / np.float32(255.0)
        assert np.array_equal(got[j], x), img_id
        best, names, cap = ds.predicted_classes(img_id)
        assert best == [44, 2] and names == ["chestofdrawers", "sky"] and cap == "A picture of chestofdrawers sky"


def test_coco_dataset_layout_and_rules(tmp_path):
    """`--data_type coco_object` on a tiny fake tree laid out as the COCO driver expects (coco/annotations/
    instances_val2017.json, coco/images/val2017, GPT table keyed by the 12-digit id): categories / image order from the
    JSON (pycocotools index order), GT painted from instance masks in annotation order with the first annotation
    winning a pixel (PnP_OVSS_0514_updated_segmentation_coco.py:1099-1110), class positions via cats[..]['id']."""
    import types
    from PIL import Image
    from pnp_ovss import datasets, host
    (tmp_path / "coco/annotations").mkdir(parents=True)
    (tmp_path / "coco/images/val2017").mkdir(parents=True)
    (tmp_path / "GPT4o_classification").mkdir()
    rng = np.random.default_rng(5)
    cats = [{"id": 1, "name": "person", "supercategory": "x"}, {"id": 3, "name": "car", "supercategory": "x"},
            {"id": 10, "name": "traffic light", "supercategory": "x"}]
    images, anns, table = [], [], {}
    for k, (iid, (h, w)) in enumerate(((139, (40, 60)), (285, (48, 48)))):
        Image.fromarray(rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)).save(tmp_path / f"coco/images/val2017/{iid:012d}.jpg", quality=95)
        images.append({"id": iid, "file_name": f"{iid:012d}.jpg", "height": h, "width": w})
        anns.append({"id": 10 * k, "image_id": iid, "category_id": 3, "iscrowd": 0, "segmentation": [[2, 1, 10, 1, 10, 6, 2, 6]]})
        anns.append({"id": 10 * k + 1, "image_id": iid, "category_id": 10, "iscrowd": 0, "segmentation": [[5, 3, 20, 3, 20, 9, 5, 9]]})
        table[f"{iid:012d}"] = "[10: traffic light, 3: car, 77: cell phone, 1: person], [90, 80, 99, 40]"
    (tmp_path / "coco/annotations/instances_val2017.json").write_text(json.dumps({"categories": cats, "images": images, "annotations": anns}))
    (tmp_path / "GPT4o_classification/coco_object_classification_noboundary.json").write_text(json.dumps(table))
    args = types.SimpleNamespace(home_dir=str(tmp_path), img_size=32, data_type="coco_object")
    ds = datasets.make_dataset(args, 0, 1)
    assert ds.nms == ["person", "car", "trafficlight"] and ds.class_ids == [1, 3, 10] and ds.total_hist.shape == (91, 91)
    (b,) = list(ds.batches(4))
    assert sorted(b["img_ids"]) == [139, 285] and b["imgs"].shape == (2, 3, 32, 32)
    gt = b["label_trues"][0]
    assert gt[1, 2] == 3 and gt[3, 7] == 3 and gt[4, 12] == 10 and gt[0, 0] == 0 and gt[6, 5] == 10      # overlap keeps "car"
    best, names, cap = ds.predicted_classes(139)
    assert best == [2, 1] and names == ["trafficlight", "car"] and cap == "A picture of trafficlight car"   # id 77: not a category here
    assert host.remap_lut(best, True, 3, ds.class_ids) == [0, 10, 3]


def test_build_model_from_checkpoint_matches_reference_load_checkpoint(tmp_path, golden_dir):
    """a-16: `build_model(checkpoint=...)` (torch.load -> pos-embed re-tile -> drop mismatched keys -> engine) against the
    reference model after its own BaseModel.load_checkpoint of the same synthetic .pth (golden checkpoint_small.npz)."""
    import argparse as ap
    from pnp_ovss import config as C, synth
    from pnp_ovss.model import build_model
    from lavis.models.blip_models.blip_image_text_matching import compute_gradcam_ensemble
    g = np.load(os.path.join(golden_dir, "checkpoint_small.npz"))
    cfg = C.ModelCfg(**json.loads(str(g["cfg"])))
    cfg_ck = C.ModelCfg(**json.loads(str(g["cfg_ckpt"])))
    ck = synth.synth_checkpoint(cfg, cfg_ck, int(g["ckpt_seed"]))
    path = str(tmp_path / "ckpt.pth")
    torch.save({"model": {k: torch.from_numpy(v.copy()) for k, v in ck.items()}}, path)
    with pytest.warns(UserWarning, match="itm_head.bias"):
        model = build_model(cfg=cfg, max_batch=2, max_text_len=32, stash_layer=7, bf16=False, checkpoint=path,
                            seed=int(g["init_seed"]))
    _, imgs = synth.synth_images(2, cfg.img_size, seed=int(g["image_seed"]))
    caps = [str(c) for c in g["captions"]]
    tok500 = model.module.tokenizer(caps, padding="max_length", max_length=500, return_tensors="pt")
    blocks, _, logits = compute_gradcam_ensemble(ap.Namespace(img_size=cfg.img_size), model.module, torch.from_numpy(imgs), caps, tok500)
    assert np.abs(blocks[7][9].numpy() - g["map_7_9"]).max() < 1e-4
    np.testing.assert_allclose(logits.cpu().numpy(), g["logits"], atol=5e-3)
    model.engine.close()


def test_lazy_model_from_checkpoint_retiles_pos_embed_at_first_call(tmp_path, golden_dir, monkeypatch):
    """The reference's argument-less load (PnP.py:1212) with a checkpoint: the lazy model keeps the checkpoint's own pos-embed
    grid in its buffers and re-tiles it (base_model.py:108-110) when the first compute_gradcam_ensemble(args, ...) fixes the
    geometry -- same map as the reference model after its own load_checkpoint (golden checkpoint_small.npz)."""
    import argparse as ap
    from pnp_ovss import config as C, synth
    from lavis.models import load_model_and_preprocess
    from lavis.models.blip_models.blip_image_text_matching import compute_gradcam_ensemble
    g = np.load(os.path.join(golden_dir, "checkpoint_small.npz"))
    cfgd = json.loads(str(g["cfg"]))
    cfg = C.ModelCfg(**cfgd)
    cfg_ck = C.ModelCfg(**json.loads(str(g["cfg_ckpt"])))
    ck = synth.synth_checkpoint(cfg, cfg_ck, int(g["ckpt_seed"]))
    path = str(tmp_path / "ckpt.pth")
    torch.save({"model": {k: torch.from_numpy(v.copy()) for k, v in ck.items()}}, path)
    (tmp_path / "model.json").write_text(json.dumps(dict(cfgd, weight_seed=int(g["init_seed"]))))
    monkeypatch.setenv("PNP_OVSS_MODEL_CONFIG", str(tmp_path / "model.json"))
    monkeypatch.setenv("PNP_OVSS_CHECKPOINT", path)
    monkeypatch.delenv("PNP_OVSS_DTYPE", raising=False)
    with pytest.warns(UserWarning, match="itm_head.bias"):
        model, _, _ = load_model_and_preprocess("blip_image_text_matching", "large", device=0, is_eval=True)
    assert model._engine is None and model.pos_embed_raw.shape[1] == cfg_ck.n_img_tokens      # the checkpoint's grid, not yet re-tiled
    _, imgs = synth.synth_images(2, cfg.img_size, seed=int(g["image_seed"]))
    caps = [str(c) for c in g["captions"]]
    tok500 = model.module.tokenizer(caps, padding="max_length", max_length=500, return_tensors="pt")
    args = ap.Namespace(img_size=cfg.img_size, max_att_block_num=8, prune_att_head="9", batch_size=2)
    blocks, _, logits = compute_gradcam_ensemble(args, model.module, torch.from_numpy(imgs), caps, tok500)
    assert model.engine.mode == "f32" and model.engine.cfg.img_size == cfg.img_size
    assert np.abs(blocks[7][9].numpy() - g["map_7_9"]).max() < 1e-4
    np.testing.assert_allclose(logits.cpu().numpy(), g["logits"], atol=5e-3)
    model.engine.close()


def test_compute_gradcam_ensemble_full_return_value_all_layers_and_heads(golden_dir):
    """f-4 (layer / head sweep): with stash_layer = 0 every [layer][head] entry of compute_gradcam_ensemble's return value
    (blip_image_text_matching.py:411-435, 12 x 12 maps) comes out of ONE forward -- against the reference's own 144 maps;
    hook accessors follow the layer; layers below stash_layer fail loudly."""
    from pnp_ovss import config as C, synth
    from pnp_ovss.model import build_model
    from lavis.models.blip_models.blip_image_text_matching import compute_gradcam_ensemble
    g = np.load(os.path.join(golden_dir, "gradcam_small.npz"))
    cfg = C.ModelCfg(**json.loads(str(g["cfg"])))
    model = build_model(cfg=cfg, max_batch=2, max_text_len=32, stash_layer=0, mode="f32", seed=int(g["weight_seed"]))
    _, imgs = synth.synth_images(2, cfg.img_size, seed=int(g["image_seed"]))
    caps = [str(c) for c in g["captions"]]
    tok500 = model.module.tokenizer(caps, padding="max_length", max_length=500, return_tensors="pt")
    blocks, _, _ = compute_gradcam_ensemble(argparse.Namespace(img_size=cfg.img_size), model.module, torch.from_numpy(imgs), caps, tok500)
    worst = 0.0
    for layer in (11, 0, 7, 3, 8, 1, 2, 4, 5, 6, 9, 10):            # any order: each layer re-runs the (text-only) backward
        for head in range(12):
            worst = max(worst, float(np.abs(blocks[layer][head].numpy() - g["maps"][layer, head]).max()))
    assert worst < 1e-4, worst
    hook0 = model.module.text_encoder.base_model.base_model.encoder.layer[0].crossattention.self
    np.testing.assert_allclose(hook0.get_attn_gradients().cpu().numpy(), g["dP0"], atol=3e-4)
    model.engine.close()
    model = build_model(cfg=cfg, max_batch=2, max_text_len=32, stash_layer=7, mode="f32", seed=int(g["weight_seed"]))
    blocks, _, _ = compute_gradcam_ensemble(argparse.Namespace(img_size=cfg.img_size), model.module, torch.from_numpy(imgs), caps, tok500)
    assert np.abs(blocks[9][3].numpy() - g["maps"][9, 3]).max() < 1e-4      # layers above stash_layer are there too
    with pytest.raises(RuntimeError):
        blocks[3][0]
    model.engine.close()


def test_cli_layer_head_sweep_and_crf_only(tmp_path):
    """`--ensemble_blocks saveall --layer 12` (sweep restricted to the last text layer: 12 heads) with `--postprocess crf`
    (PnP.py:1013-1026, CRF on the un-blurred maps): one histogram file per (layer, head) and batch, pixel totals exact."""
    save = tmp_path / "sweep"
    cmd = [sys.executable, os.path.join(ROOT, "pnp-ovss_amd", "PnP_OVSS_0514_updated_segmentation.py"),
           "--save_path", str(save), "--world_size", "1", "--img_size", "336", "--del_patch_num", "sort_thresh005",
           "--batch_size", "2", "--max_att_block_num", "8", "--drop_iter", "2", "--prune_att_head", "9", "--threshold", "0.15",
           "--postprocess", "crf", "--data_type", "synthetic", "--synthetic_images", "2", "--ensemble_blocks", "saveall",
           "--layer", "12", "--dtype", "bf16x3"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    files = sorted(glob.glob(str(save / "all_drop_hist_with_filtered_caption" / "*.npy")))
    assert len(files) == 12 and all("_max_blocknum_12_atthead_" in f for f in files)
    for f in files:
        h = np.load(f)
        assert h.shape == (21, 21) and h.sum() == 2 * 336 * 336


def test_bench_two_ranks_on_one_gpu_rehearsal():
    """The N > 1 path of bench.py under its real launcher with the real engines, on the one GPU a test box has: two rank
    processes share cuda:0 over gloo (RCCL refuses two ranks on one device; the driver's scaling job is where RCCL itself
    runs).  Mirrors PnP.py:1193-1225 (one process per rank, DDP-ctor weight broadcast) and :1439 (spawn): both ranks must
    take part in the weight broadcast, the histogram all-reduce and the label gather -- the all-reduced confusion matrix
    holds every pixel of both ranks' images, rank 0 ends up with two label-map buffers."""
    steps, warm, B, S = 2, 1, 35, 336
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--backend", "gloo", "--steps", str(steps),
           "--warmup", str(warm), "--pipelines", "1", "--no-cpu-baseline", "--no-other-modes", "--no-other-configs", "--no-noise12"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["n_ranks"] == 2 and line["scaling"] == "weak"
    assert line["images_per_rank"] == B * steps and line["gathered_label_maps"] == 2
    col = line["collectives"]
    assert col["gathered_label_bytes"] == [B * S * S] * 2                     # one uint8 label map set per rank
    assert len(col["per_rank_images_per_sec"]) == 2 and min(col["per_rank_images_per_sec"]) > 0
    # every step (warm-up included: the histogram state lives across them) counts each pixel once, on each rank
    assert col["hist_ndrop_total"] == 2 * (steps + warm) * B * S * S
    assert col["weight_bytes"] > 1.7e9 and line["value"] > 0


def test_validate_real_kit_self_test(tmp_path):
    """tools/validate_real.py (the one-command real-data check: CLI per parity mode -> Calculate_mIoU.py, per-image label
    differences f32 vs bf16x3, dumped [7][9] maps) rehearsed end to end on a generated VOC-layout tree of JPEG files with a
    BLIP-ITM-large-shaped .pth checkpoint and a word-piece vocabulary file -- the kit must run where the real files do not
    exist, so that it works for whoever has them.  The split-bf16 mode must agree with the fp32 mode to north_star's 1e-4
    on the dumped maps and on all but a fraction of a percent of the label pixels."""
    out = tmp_path / "vr"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "validate_real.py"), "--self_test", "--max_batches", "2",
                        "--out", str(out)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    rep = json.load(open(out / "report.json"))
    for mode in ("f32", "bf16x3"):
        m = rep["miou"][mode]
        assert m["cli_summary"]["images"] == 70 and 0 < m["n_drop"]["Mean IoU"] < 1 and "Mean IoU" in m["1_drop"]
        assert abs(m["n_drop"]["Mean IoU"] - m["cli_summary"]["Mean IoU"]) < 1e-6       # files on disk == the run's own reduce
    d = rep["f32_vs_bf16x3"]
    assert d["images"] == 70 and d["map_7_9_max_abs_f32_vs_bf16x3"] < 1e-4
    assert d["n_drop_label_pixels_differing"]["mean"] < 1e-3 and d["n_drop_label_pixels_differing"]["max"] < 2e-2
    assert abs(rep["miou"]["f32"]["n_drop"]["Mean IoU"] - rep["miou"]["bf16x3"]["n_drop"]["Mean IoU"]) < 1e-3
    z = np.load(out / "maps.npz")
    assert z["map_7_9_f32"].shape == z["map_7_9_bf16x3"].shape and z["map_7_9_f32"].shape[0] == 3


def test_cli_two_ranks_on_one_gpu_equal_one_rank(tmp_path):
    """The CLI's multi-rank path (VERDICT r04 task 5) under its own `--world_size 2` spawn on the one GPU a test box has
    (`--share_gpu --backend gloo`; RCCL refuses two ranks on one device): rank 1 loads nothing and receives rank 0's weights
    in the start-up broadcast (PnP.py:1218), the digests are compared, the image count and the confusion matrix are
    all-reduced (PnP.py:513-520 -> Calculate_mIoU.py:215-219 in the reference), the label maps are gathered on rank 0.
    A 12-image VOC-layout tree of JPEG files, one image per batch (the reference's results depend on what else is in a
    batch -- the longest caption sets L, PnP.py:639 -- so one image per batch makes the sharding invisible): the summed
    histogram, the per-batch .npy files and every gathered label map must equal the one-rank run's."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import cli_e2e
    home = tmp_path / "home"
    home.mkdir()
    cli_e2e.build_tree(home, 12, seed=3)
    runs = {}
    for W in (1, 2):
        save = tmp_path / f"out{W}"
        cmd = [sys.executable, os.path.join(ROOT, "pnp-ovss_amd", "PnP_OVSS_0514_updated_segmentation.py"),
               "--home_dir", str(home), "--save_path", str(save), "--world_size", str(W), "--img_size", "336", "--del_patch_num",
               "sort_thresh005", "--batch_size", "1", "--max_att_block_num", "8", "--drop_iter", "4", "--prune_att_head", "9",
               "--sort_threshold", "0.05", "--threshold", "0.15", "--postprocess", "blur+crf", "--data_type", "voc",
               "--dtype", "bf16x3", "--gather_labels", "--master_port", str(_free_port())]
        if W > 1:
            cmd += ["--share_gpu", "--backend", "gloo"]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
            env.pop(k, None)
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, env=env)
        assert out.returncode == 0, out.stderr[-3000:]
        line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
        files = {}
        for d in ("hist_withfiltered_caption", "all_drop_hist_with_filtered_caption"):
            for f in sorted(glob.glob(str(save / d / "*.npy"))):
                files[d + "/" + os.path.basename(f)] = np.load(f)
        runs[W] = (line, files, dict(np.load(save / "label_maps.npz")), out.stdout)
    (l1, f1, m1, _), (l2, f2, m2, o2) = runs[1], runs[2]
    assert l1["images"] == l2["images"] == 12 and l2["ranks"] == 2 and l2["images_rank0"] == 6
    assert l2["weights_sync"]["mode"] == "broadcast" and l2["weights_sync"]["bytes"] > 1.7e9 and "digests equal" in o2
    assert l2["gathered_label_maps"]["images"] == 12 and len(l2["gathered_label_maps"]["bytes_per_rank"]) == 2
    assert l1["pixels"] == l2["pixels"] and l1["Mean IoU"] == l2["Mean IoU"] and l1["Pixel Accuracy"] == l2["Pixel Accuracy"]
    assert len(f1) == 24 and sorted(f1) == sorted(f2)
    for k in f1:
        np.testing.assert_array_equal(f1[k], f2[k])
    assert sorted(m1) == sorted(m2) and len(m1) == 12
    for k in m1:
        np.testing.assert_array_equal(m1[k], m2[k])


def test_bench_under_the_drivers_launcher_with_rccl_world_1():
    """The driver's literal multi-GPU launch form -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...` -- at the N a one-GPU box allows, with the real backend
    ("nccl" = RCCL): the rank initialises an RCCL process group and every collective of the path (weight broadcast, timing
    max, confusion-matrix all-reduce, label gather, per-rank all-gather) goes through RCCL calls, so an API-level mistake
    in the N > 1 path shows here and not first in the scaling job (PnP.py:45-54, :1218)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--pipelines", "2",
           "--no-cpu-baseline", "--no-other-modes", "--no-other-configs", "--no-noise12", "--no-fixture-check"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["n_ranks"] == 1 and line["gathered_label_maps"] == 1 and line["value"] > 0
    col = line["collectives"]
    assert col["weight_bytes"] > 1.7e9 and col["weight_broadcast_ms"] >= 0 and len(col["per_rank_images_per_sec"]) == 1
    assert col["gathered_label_bytes"] == [35 * 336 * 336]
    assert len(line["pipelines"]["engine_device_bytes"]) == 2
    assert line["pipelines"]["engine_device_bytes"][1] < line["pipelines"]["engine_device_bytes"][0]     # the second engine holds no weights
