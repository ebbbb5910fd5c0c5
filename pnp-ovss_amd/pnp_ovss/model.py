"""Python host mirror of the reference's model interface for the hot path, backed by the HIP engine.

Reference surface kept:
  * `load_model_and_preprocess("blip_image_text_matching", "large", device=..., is_eval=True)`
    -> (model, vis_processors, txt_processors)                     (PnP.py:1212-1213; lavis shim)
  * `model.tokenizer(...)`, `model.tokenizer.decode([id])`          (PnP.py:271,317,608,813)
  * `model.text_encoder.base_model.base_model.encoder.layer[i].crossattention.self.save_attention`
    + get_attention_map / get_attn_gradients                       (PnP.py:294-298; med.py:162-180)
  * `compute_gradcam_ensemble(args, model, visual_input, text_input, tokenized_text, drop_iter=0)`
    -> (gradcam_blocklist[layer][head] -> CPU (B,L-1,P,P) fp32, [], logits)
                                                                    (blip_image_text_matching.py:386-457)
  * `Inference_BLIP_filteredcaption`-equivalent `drop_loop` and the batch segmenter that replaces
    save_img_union_attention's device work                          (PnP.py:290-521, 564-722)
There is no eager / CPU fallback: every tensor op below is plumbing around pnp_ovss.hip.Engine.
"""
import os
import warnings

import numpy as np
import torch

from . import config as C
from . import host
from . import synth
from .hip import Engine
from .tokenizer import SynthTokenizer, WordPieceTokenizer


class _CrossSelf:
    """`layer[i].crossattention.self` accessor object (med.py:162-180)."""

    def __init__(self, model, idx):
        self._m, self._i = model, idx
        self.save_attention = False

    def _check(self):
        eng = self._m.engine
        if self._i < eng.stash_layer:
            raise RuntimeError(f"the engine keeps cross-attention maps for text layers >= {eng.stash_layer} "
                               f"(max_att_block_num - 1); re-create the model with stash_layer={self._i}")
        self._m._grad_to(self._i)

    def get_attention_map(self):
        self._check()
        return self._m._stash("P")

    def get_attn_gradients(self):
        self._check()
        return self._m._stash("dP")


class _Obj:
    pass


class BlipITM:
    """BLIP image-text matching model whose forward / GradCAM run in libpnp_hip.so."""

    def __init__(self, cfg, engine, tokenizer):
        self.cfg, self.engine, self.tokenizer = cfg, engine, tokenizer
        self.max_txt_len = 500                                     # blip_image_text_matching.py:48
        layers = []
        for i in range(cfg.txt_layers):
            lay = _Obj()
            lay.crossattention = _Obj()
            lay.crossattention.self = _CrossSelf(self, i)
            layers.append(lay)
        enc = _Obj()
        enc.layer = layers
        bm2 = _Obj()
        bm2.encoder = enc
        bm1 = _Obj()
        bm1.base_model = bm2
        self.text_encoder = _Obj()
        self.text_encoder.base_model = bm1
        self.visual_encoder = _Obj()
        self.visual_encoder.vision_width = cfg.vit_dim
        self.module = self                                         # DDP-wrapper attribute the driver dereferences
        self._last = None
        self._grad_layer = None

    # nn.Module-ish no-ops the driver calls
    def eval(self):
        return self

    def to(self, device):
        return self

    def zero_grad(self):
        pass

    @property
    def device(self):
        return self.engine.device

    def _tok_longest(self, captions):
        return self.tokenizer(captions, padding="longest", truncation=True, max_length=self.max_txt_len,
                              return_tensors="pt")

    def _grad_to(self, layer):
        """Make P / dP / gradcam_gather refer to `layer`: the analytic backward re-run down to it when the last one
        stopped elsewhere (activations of the forward are still in the engine)."""
        if self._grad_layer != layer:
            B, L = self._last
            self.engine.xattn_grad(B, L, layer)
            self._grad_layer = layer

    def _stash(self, name):
        B, L = self._last
        N, nst = self.cfg.n_img_tokens, (self.cfg.n_img_tokens + 63) // 64 * 64
        flat = self.engine.buffer(name)
        return flat[: B * self.cfg.txt_heads * L * nst].view(B, self.cfg.txt_heads, L, nst)[..., :N]

    def __call__(self, samples, match_head="itm"):
        """BlipITM.forward(match_head="itm") (blip_image_text_matching.py:217-249) -> logits (B,2)."""
        if match_head != "itm":
            raise NotImplementedError("only the ITM head is on the hot path")
        image = samples["image"].to(self.device, torch.float32).contiguous()
        text = self._tok_longest(samples["text_input"]).to(self.device)
        L = text.input_ids.shape[1]
        self.engine.vit_forward(image)
        logits = self.engine.text_forward(text.input_ids.contiguous(), text.attention_mask.contiguous(), L)
        self._last = (image.shape[0], L)
        self._grad_layer = None
        return logits


class _LazyHeads:
    def __init__(self, fn):
        self._fn, self._cache = fn, {}

    def __getitem__(self, head):
        if head not in self._cache:
            self._cache[head] = self._fn(head)
        return self._cache[head]

    def __len__(self):
        return 12


class _LazyBlocks:
    """gradcam_blocklist[layer][head]: materialised on demand (the reference builds all 144 maps and
    the driver reads one, PnP.py:619-621).  Layers >= the engine's stash_layer are available -- all 12 x 12 with
    stash_layer = 0 (the layer / head sweep the full return value exists for); a layer other than the last one
    asked for costs one more analytic backward (text layers only), no second forward."""

    def __init__(self, model, mask, L):
        self._m, self._mask, self._L = model, mask, L

    def __getitem__(self, layer):
        eng = self._m.engine
        if layer < eng.stash_layer or layer >= self._m.cfg.txt_layers:
            raise RuntimeError(f"GradCAM maps are kept for text layers >= {eng.stash_layer}; "
                               f"re-create the model with stash_layer={layer}")

        def head_map(h):
            self._m._grad_to(layer)
            return eng.gradcam_gather(self._mask, self._L, h).cpu()
        return _LazyHeads(head_map)

    def __len__(self):
        return 12


def compute_gradcam_ensemble(args, model, visual_input, text_input, tokenized_text, drop_iter=0):
    """Drop-in for blip_image_text_matching.py:386-457."""
    m = model.module if hasattr(model, "module") else model
    eng = m.engine
    image = visual_input.to(m.device, torch.float32).contiguous()
    text = m._tok_longest(text_input).to(m.device)
    L = text.input_ids.shape[1]
    B = image.shape[0]
    eng.vit_forward(image)
    logits = eng.text_forward(text.input_ids.contiguous(), text.attention_mask.contiguous(), L)
    eng.xattn_grad(B, L)
    m._last = (B, L)
    m._grad_layer = eng.stash_layer
    mask = tokenized_text.attention_mask.to(m.device).contiguous()          # the caller's 500-padded mask (:415-416)
    return _LazyBlocks(m, mask, L), [], logits


def drop_loop(args, model, txt_tokens, imgs_in, caption_list):
    """Inference_BLIP_filteredcaption (PnP.py:564-722): returns (gradcam_0, gradcam_agg) on device."""
    m = model.module if hasattr(model, "module") else model
    image = imgs_in.to(m.device, torch.float32).contiguous()
    text = m._tok_longest(caption_list).to(m.device)
    L = text.input_ids.shape[1]
    mask = txt_tokens.attention_mask.to(m.device).contiguous()
    ids = txt_tokens.input_ids.to(m.device).contiguous()
    layer = int(args.max_att_block_num) - 1 if getattr(args, "max_att_block_num", None) is not None else m.engine.stash_layer
    g0, agg, picks, _ = m.engine.drop_loop(image, ids, mask, L, int(args.prune_att_head), int(args.drop_iter), layer=layer)
    m._last = (image.shape[0], L)
    m._grad_layer = layer
    return g0, agg


def _resize_pos_embed(pos, n_new_grid):
    """interpolate_pos_embed (base_model.py:44-73): bicubic, align_corners=False, cls token kept."""
    pos = torch.as_tensor(pos, dtype=torch.float32)
    D = pos.shape[-1]
    n_old = int((pos.shape[-2] - 1) ** 0.5)
    if n_old == n_new_grid:
        return pos
    extra, grid = pos[:, :1], pos[:, 1:]
    grid = grid.reshape(-1, n_old, n_old, D).permute(0, 3, 1, 2)
    grid = torch.nn.functional.interpolate(grid, size=(n_new_grid, n_new_grid), mode="bicubic", align_corners=False)
    grid = grid.permute(0, 2, 3, 1).flatten(1, 2)
    return torch.cat((extra, grid), dim=1)


def merge_checkpoint(cfg, ckpt_state, init_state):
    """BaseModel.load_checkpoint (base_model.py:86-125) as a pure host function: `ckpt_state` (the checkpoint's
    "model" dict) over `init_state` (the module's values before loading).  The pos_embed grid is re-tiled to cfg.grid
    (:108-110), keys whose shape differs from the model's are dropped (:116-119), everything else overwrites
    (strict=False: keys the path does not have are ignored).  Returns (state, dropped, missing): `missing` = path
    tensors the checkpoint did not provide (they keep their init value, as in the reference)."""
    shapes = synth.param_shapes(cfg)
    ck = dict(ckpt_state)
    if "visual_encoder.pos_embed" in ck:
        ck["visual_encoder.pos_embed"] = _resize_pos_embed(ck["visual_encoder.pos_embed"], cfg.grid)
    state, dropped, missing = {}, [], []
    for k, shp in shapes.items():
        if k in ck and tuple(ck[k].shape) == tuple(shp):
            state[k] = ck[k]
        else:
            if k in ck:
                dropped.append(k)
            else:
                missing.append(k)
            state[k] = init_state[k]
    return state, dropped, missing


def build_model(model_type="large", img_size=336, device=0, max_batch=35, max_text_len=64, stash_layer=7, bf16=True,
                checkpoint=None, vocab=None, seed=0, cfg=None, mode=None):
    """from_config + load_checkpoint (blip_image_text_matching.py:297-314, base_model.py:86-125)."""
    if cfg is None:
        if model_type != "large":
            raise ValueError("only blip_image_text_matching/large is on the hot path")
        cfg = C.blip_itm_large(img_size)
    checkpoint = checkpoint or os.environ.get("PNP_OVSS_CHECKPOINT")
    vocab = vocab or os.environ.get("PNP_OVSS_VOCAB")
    dev = device if isinstance(device, int) else (torch.device(device).index or 0)
    eng = Engine(cfg, max_batch=max_batch, max_text_len=max_text_len, stash_layer=stash_layer, bf16=bf16, device=dev, mode=mode)
    init = synth.synth_state_dict(cfg, seed)                      # stands in for the module's initialisation
    if checkpoint:
        sd = torch.load(checkpoint, map_location="cpu")
        sd = sd["model"] if "model" in sd else sd
        state, dropped, missing = merge_checkpoint(cfg, sd, init)
        if dropped or missing:
            warnings.warn(f"checkpoint {checkpoint}: dropped (shape mismatch) {dropped}, not provided {missing}: "
                          f"those tensors keep their initial values, like load_state_dict(strict=False)")
        eng.load_state_dict(state)
    else:
        warnings.warn("no BLIP checkpoint given (PNP_OVSS_CHECKPOINT): using seeded synthetic weights")
        eng.load_state_dict(init)
    tok = WordPieceTokenizer(vocab) if vocab else SynthTokenizer(cfg.vocab)
    return BlipITM(cfg, eng, tok)


class Segmenter:
    """Device work of save_img_union_attention (PnP.py:290-521; COCO driver PnPc.py:338-642) for one batch: drop
    loop, merge, threshold/upsample, blur, CRF, argmax/remap, histogram.

    data_type "voc" | "psc" | "ade20k": PnP.py rules (both branches; Scale_0_1 on the 1-drop branch only; labels are
    class positions + 1).  "coco_object" | "coco_stuff": the COCO driver's rules -- the 1-drop branch runs only when
    drop_iter < 3 (PnPc.py:420), Scale_0_1 on both branches (:436, :527), background rule of :446-450 / :470-473,
    labels are COCO category ids (`class_ids[j]` = cats[j]['id'], :458-463 ...), n_class 91 / 183 (:597-600)."""

    def __init__(self, model, data_type, n_class, threshold=0.15, postprocess="blur+crf", max_pixels_per_image=600 * 600,
                 max_channels=24, crf_chunk=0, class_ids=None):
        self.m = model.module if hasattr(model, "module") else model
        self.data_type, self.n_class, self.threshold, self.mode = data_type, n_class, threshold, postprocess
        self.coco = data_type.startswith("coco")
        self.class_ids = list(class_ids) if class_ids is not None else None
        if self.coco and self.class_ids is None:
            raise ValueError("COCO data types need class_ids (cats[j]['id'])")
        if n_class > 256:
            raise ValueError("label maps are uint8: at most 256 classes")
        eng = self.m.engine
        eng.post_reserve(eng.max_batch, eng.max_batch * max_pixels_per_image, max_pixels_per_image, max_channels, crf_chunk)
        self.hist_1drop = torch.zeros(n_class * n_class, device=eng.device, dtype=torch.int64)
        self.hist_ndrop = torch.zeros(n_class * n_class, device=eng.device, dtype=torch.int64)

    def prepare(self, captions, best_class_idx, org_images, label_trues, gt_dev=None):
        """Host-only half of a batch (no dependence on the model's results): tokenisation, word-piece merge plans, label
        LUTs, the concatenated RGB / ground-truth buffers on the device.  A driver that pipelines batches calls this for
        batch i+1 while the GPU still works on batch i."""
        m, dev = self.m, self.m.engine.device
        tok500 = m.tokenizer(captions, padding="max_length", max_length=500, return_tensors="pt")
        ids = tok500.input_ids.numpy()
        sizes = [(int(x.shape[0]), int(x.shape[1])) for x in org_images]
        plans, luts, bgs = [], [], []
        for i, best in enumerate(best_class_idx):
            pieces = host.caption_pieces(m.tokenizer, ids[i])
            bg = host.has_background(self.data_type, len(best))
            plans.append(host.merge_plan(pieces, len(best)))
            luts.append(host.remap_lut(best, bg, len(best) + int(bg), self.class_ids))
            bgs.append(bg)
        if all(isinstance(x, torch.Tensor) and x.is_cuda for x in org_images):      # decoded on the device (hip.jpeg_decode_batch)
            rgb = torch.cat([x.reshape(-1) for x in org_images])
        else:
            rgb = torch.from_numpy(np.concatenate([np.ascontiguousarray(x.cpu().numpy() if isinstance(x, torch.Tensor) else x,
                                                                        dtype=np.uint8).reshape(-1) for x in org_images])).to(dev)
        gt = gt_dev
        if gt is None and label_trues is not None:
            gt = torch.from_numpy(np.concatenate([np.asarray(x, dtype=np.float32).reshape(-1) for x in label_trues])).to(dev)
        return dict(tok500=tok500, captions=captions, sizes=sizes, plans=plans, luts=luts, bgs=bgs, rgb=rgb, gt=gt)

    def launch(self, args, imgs_in, prep, run_1drop=True, hists=None):
        """Device half: drop loop, merge, threshold / upsample, blur, CRF, argmax / remap, histogram into `hists`
        (default: self.hist_1drop / self.hist_ndrop).  Returns the two label-map lists (views of engine buffers: valid
        until the next launch); the tail of the work is still queued on the stream when this returns."""
        m, eng = self.m, self.m.engine
        h1, hn = hists if hists is not None else (self.hist_1drop, self.hist_ndrop)
        g0, agg = drop_loop(args, m, prep["tok500"], imgs_in, prep["captions"])
        if self.coco and int(args.drop_iter) >= 3:
            run_1drop = False
        scale01 = (True, self.coco)                                 # Scale_0_1 on (1-drop, N-drop)
        eng.post_prepare(prep["sizes"], prep["plans"], prep["luts"], prep["bgs"], rgb=prep["rgb"], gt=prep["gt"],
                         want_crf=bool(self.mode and "crf" in self.mode))
        out1 = outn = None
        if run_1drop and agg is not None and self.mode == "blur+crf":
            # both branches share the image lattices: one DenseCRF run over two channel groups (same results)
            l1, ln = eng.postprocess_pair(g0, agg, self.threshold, self.n_class, h1, hn, scale01)
            return eng.split_labels(l1), eng.split_labels(ln)
        if run_1drop or agg is None:
            out1 = eng.split_labels(eng.postprocess(g0, self.threshold, scale01[0], self.mode, self.n_class, h1))
        if agg is not None:
            outn = eng.split_labels(eng.postprocess(agg, self.threshold, scale01[1], self.mode, self.n_class, hn))
        return out1, outn

    def run(self, args, imgs_in, captions, best_class_idx, org_images, label_trues, run_1drop=True, gt_dev=None):
        return self.launch(args, imgs_in, self.prepare(captions, best_class_idx, org_images, label_trues, gt_dev), run_1drop)
