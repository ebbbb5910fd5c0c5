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


MAX_TEXT_TOKENS = 512        # longest caption the text kernels take (csrc/text_kernels.hip TXT_LONG_L = BERT's max_position_embeddings;
                             # the reference tokenises to max_length = 500, PnP.py:271,318)


class BlipITM(torch.nn.Module):
    """BLIP image-text matching model whose forward / GradCAM run in libpnp_hip.so.

    An `nn.Module` so that the reference driver's `DDP(model, device_ids=[rank])` (PnP.py:1218) accepts it: DDP needs one
    parameter that requires grad (`ddp_anchor`, never used) and broadcasts the parameters AND buffers of rank 0 at
    construction -- which is the reference's weight broadcast, so a lazily built model keeps its fp32 weights in buffers
    (`weights_flat`, `pos_embed_raw`) until the engine is created.

    Two ways to get one (build_model): eager -- geometry, batch, text length, stash layer and compute mode given, engine
    created at once (CLI, bench, tests) -- or lazy: the reference's literal
    `load_model_and_preprocess("blip_image_text_matching", "large", device=rank, is_eval=True)` carries none of them, so
    the engine is created by the first `compute_gradcam_ensemble(args, ...)` / `drop_loop(args, ...)` / forward call from
    `args.img_size` (B/blip_image_text_matching.py:408), `args.max_att_block_num`, `args.batch_size` and the batch at hand,
    in the parity mode ("f32" unless PNP_OVSS_DTYPE says otherwise), and re-created if a later call does not fit it."""

    def __init__(self, cfg, engine, tokenizer, lazy=None, device=None):
        super().__init__()
        self.cfg, self.tokenizer = cfg, tokenizer
        self._engine = engine
        self._lazy = lazy                                          # None (eager) or dict(mode=, seed=, index=)
        self._device = torch.device("cuda", device if device is not None else 0) if engine is None else engine.device
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
        self.ddp_anchor = torch.nn.Parameter(torch.zeros(1, device=self._device))
        self._last = None
        self._grad_layer = None

    @property
    def module(self):                                              # `model.module.…` also works without the DDP wrapper
        return self

    @property
    def device(self):
        return self._engine.device if self._engine is not None else self._device

    @property
    def engine(self):
        if self._engine is None:
            self.ensure_engine()
        return self._engine

    # ------------------------------------------------------------------ lazy engine
    def ensure_engine(self, img_size=None, batch=None, stash_layer=None, text_len=None):
        """Create (or, for a lazily built model, re-create) the engine so that it fits the call at hand.  Eager models
        only check and fail loudly, as before."""
        if text_len is not None and text_len > MAX_TEXT_TOKENS:
            # before anything is torn down: no engine can serve the call (BERT's position table ends at 512; the reference's
            # tokenizer truncates at max_length = 500)
            raise RuntimeError(f"caption of {text_len} tokens: the text stack takes at most {MAX_TEXT_TOKENS} "
                               f"(max_position_embeddings)")
        eng = self._engine
        if eng is not None:
            fits = ((img_size is None or img_size == eng.cfg.img_size) and (batch is None or batch <= eng.max_batch) and
                    (stash_layer is None or stash_layer >= eng.stash_layer) and (text_len is None or text_len <= eng.max_text_len))
            if fits:
                return eng
            if self._lazy is None:
                raise RuntimeError(f"engine built for img_size={eng.cfg.img_size}, batch<={eng.max_batch}, text<={eng.max_text_len}, "
                                   f"layers>={eng.stash_layer}; the call needs img_size={img_size}, batch={batch}, text={text_len}, "
                                   f"layer={stash_layer}: re-create the model with matching arguments")
            warnings.warn("re-creating the HIP engine: the call does not fit the one built by the first call")
            img_size = img_size or eng.cfg.img_size
            batch = max(batch or 0, eng.max_batch)
            stash_layer = min(stash_layer if stash_layer is not None else eng.stash_layer, eng.stash_layer)
            text_len = max(text_len or 0, eng.max_text_len)
            eng.close()
            self._engine = None
        if self._lazy is None:
            raise RuntimeError("model has no engine")
        import dataclasses
        lz = self._lazy
        cfg = dataclasses.replace(self.cfg, img_size=int(img_size or self.cfg.img_size))
        if stash_layer is None:
            stash_layer = int(os.environ.get("PNP_OVSS_STASH_LAYER", 0))      # 0: all 12 x 12 maps, like the reference's return value
        eng = Engine(cfg, max_batch=int(batch or lz["max_batch"]), max_text_len=min(MAX_TEXT_TOKENS, max(64, int(text_len or 0))),
                     stash_layer=int(stash_layer), mode=lz["mode"], device=self._device.index or 0)
        flat = self.weights_flat
        sd = {}
        for n, (o, shp) in lz["index"].items():
            sd[n] = flat[o:o + int(np.prod(shp))].view(*shp)
        sd["visual_encoder.pos_embed"] = _resize_pos_embed(self.pos_embed_raw.cpu(), cfg.grid).to(flat.device)   # base_model.py:108-110
        eng.load_state_dict(sd)
        if flat.is_cuda:
            # the engine now holds its own (converted) copy: the 1.8 GB of fp32 leave the module's buffers -- a DDP wrapper
            # would otherwise re-broadcast them on every DDP.forward under its default broadcast_buffers=True -- and stay
            # on the host for the case that a later call needs a larger engine.  (A DDP wrapper built before the first
            # call keeps its own reference to the device buffer: pass broadcast_buffers=False to release that one too.)
            torch.cuda.synchronize()
            host_copy = flat.cpu()
            delattr(self, "weights_flat")                          # nn.Module.__delattr__: out of _buffers
            self.weights_flat = host_copy                          # plain attribute now
        self.cfg, self._engine = cfg, eng
        self._last = self._grad_layer = None
        return eng

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

    def forward(self, samples, match_head="itm"):
        """BlipITM.forward(match_head="itm") (blip_image_text_matching.py:217-249) -> logits (B,2)."""
        if match_head != "itm":
            raise NotImplementedError("only the ITM head is on the hot path")
        text = self._tok_longest(samples["text_input"])
        L = text.input_ids.shape[1]
        eng = self.ensure_engine(img_size=int(samples["image"].shape[-1]), batch=int(samples["image"].shape[0]), text_len=L)
        image = samples["image"].to(self.device, torch.float32).contiguous()
        text = text.to(self.device)
        eng.vit_forward(image)
        logits = eng.text_forward(text.input_ids.contiguous(), text.attention_mask.contiguous(), L)
        self._last = (image.shape[0], L)
        self._grad_layer = None
        return logits


def _call_geometry(args, m, image, L):
    """What a call needs of the engine: the reference reads args.img_size inside compute_gradcam_ensemble (B/…:408); the
    layer it will index is args.max_att_block_num - 1 (PnP.py:619-621); batches come in args.batch_size (PnP.py:59)."""
    img_size = int(getattr(args, "img_size", 0) or image.shape[-1])
    if img_size != int(image.shape[-1]):
        raise ValueError(f"args.img_size={img_size} but the images are {tuple(image.shape)}")
    layer = getattr(args, "max_att_block_num", None)
    layer = int(layer) - 1 if layer is not None else None
    batch = max(int(image.shape[0]), int(getattr(args, "batch_size", 0) or 0))
    if m._engine is not None and m._lazy is None:
        layer = None                              # eager engines serve layers >= their stash_layer; the accessors check
    return m.ensure_engine(img_size=img_size, batch=batch, stash_layer=layer, text_len=L)


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
    text = m._tok_longest(text_input)
    L = text.input_ids.shape[1]
    eng = _call_geometry(args, m, visual_input, L)
    image = visual_input.to(m.device, torch.float32).contiguous()
    text = text.to(m.device)
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
    text = m._tok_longest(caption_list)
    L = text.input_ids.shape[1]
    eng = _call_geometry(args, m, imgs_in, L)
    image = imgs_in.to(m.device, torch.float32).contiguous()
    mask = txt_tokens.attention_mask.to(m.device).contiguous()
    ids = txt_tokens.input_ids.to(m.device).contiguous()
    layer = int(args.max_att_block_num) - 1 if getattr(args, "max_att_block_num", None) is not None else eng.stash_layer
    g0, agg, picks, _ = eng.drop_loop(image, ids, mask, L, int(args.prune_att_head), int(args.drop_iter), layer=layer)
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


def _model_config(model_type, img_size):
    """The model yaml of the reference (B/blip_itm_large.yaml + med_large_config.json) as this build reads it: the
    built-in BLIP-ITM-large geometry, or a JSON file of ModelCfg fields named by PNP_OVSS_MODEL_CONFIG (tests: the
    reduced geometry of the golden vectors; "weight_seed" selects the seeded synthetic weights)."""
    path = os.environ.get("PNP_OVSS_MODEL_CONFIG")
    seed = None
    if path:
        import json
        d = json.load(open(path))
        seed = d.pop("weight_seed", None)
        cfg = C.ModelCfg(**d)
        if img_size:
            import dataclasses
            cfg = dataclasses.replace(cfg, img_size=int(img_size))
        return cfg, seed
    if model_type != "large":
        raise ValueError("only blip_image_text_matching/large is on the hot path")
    return C.blip_itm_large(int(img_size or 336)), seed            # image_size: 336 (blip_itm_large.yaml:17)


def weights_digest(flat):
    """Two order-independent 64-bit integer sums over the bit patterns of a flat fp32 weight buffer (device or host): equal
    buffers give equal digests on every rank, and the sums are exact (wrap-around int64), so ranks compare them with ==."""
    bits = flat.view(torch.int32)
    s0 = torch.zeros((), dtype=torch.int64, device=flat.device)
    s1 = torch.zeros((), dtype=torch.int64, device=flat.device)
    step = 1 << 26
    for o in range(0, bits.numel(), step):
        b = bits[o:o + step].to(torch.int64)
        s0 += b.sum()
        s1 += (b * (torch.arange(o, o + b.numel(), device=flat.device, dtype=torch.int64) % 65521 + 1)).sum()
    return torch.stack([s0, s1])


def build_model(model_type="large", img_size=None, device=0, max_batch=None, max_text_len=None, stash_layer=None, bf16=None,
                checkpoint=None, vocab=None, seed=None, cfg=None, mode=None, sync=None, receive_only=False, donor=None):
    """from_config + load_checkpoint (blip_image_text_matching.py:297-314, base_model.py:86-125).

    With the engine's sizing given (img_size or cfg, max_batch, stash_layer) the engine is created here.  Without -- the
    reference's own call, PnP.py:1212 -- the model is lazy (see BlipITM): weights are loaded into device buffers now, the
    engine follows the first call.  Compute mode: `mode` ("f32" | "bf16x3" | "bf16"), else PNP_OVSS_DTYPE, else "f32" --
    the reference's arithmetic; "bf16" is never a default.

    Multi-rank (eager models): `sync(flat)` is called with every weight of the path in one flat fp32 DEVICE buffer before
    the engine takes them -- the driver broadcasts rank 0's buffer over RCCL there and / or checks a digest across ranks
    (the construction-time broadcast of the reference's DDP wrapper, PnP.py:1218); `receive_only` ranks skip the checkpoint
    and the initialisation and only receive.  `donor`: another model of the same geometry, device and mode whose engine
    lends its (converted, read-only) weights: the new engine allocates activations and workspace only (--pipelines)."""
    cfg_seed = None
    if cfg is None:
        cfg, cfg_seed = _model_config(model_type, img_size)
    if seed is None:
        seed = int(cfg_seed if cfg_seed is not None else os.environ.get("PNP_OVSS_SEED", 0))
    if mode is None:
        mode = ("bf16" if bf16 else "f32") if bf16 is not None else os.environ.get("PNP_OVSS_DTYPE", "f32")
    checkpoint = checkpoint or os.environ.get("PNP_OVSS_CHECKPOINT")
    vocab = vocab or os.environ.get("PNP_OVSS_VOCAB")
    dev = device if isinstance(device, int) else (torch.device(device).index or 0)
    tok = WordPieceTokenizer(vocab) if vocab else SynthTokenizer(cfg.vocab)
    eager = max_batch is not None and stash_layer is not None
    if donor is not None:
        if not eager:
            raise ValueError("donor= needs an eager model (max_batch and stash_layer given)")
        d = donor.module if hasattr(donor, "module") else donor
        eng = Engine(cfg, max_batch=max_batch, max_text_len=max_text_len or 64, stash_layer=stash_layer, device=dev, mode=mode,
                     share_weights_with=d.engine)
        return BlipITM(cfg, eng, tok)
    if receive_only:
        if not (eager and sync is not None):
            raise ValueError("receive_only needs an eager model and a sync callable that fills the weights")
        shapes = synth.param_shapes(cfg)
        torch.cuda.set_device(dev)
        flat = torch.empty(sum(int(np.prod(s)) for s in shapes.values()), dtype=torch.float32, device=torch.device("cuda", dev))
        sync(flat)
        eng = Engine(cfg, max_batch=max_batch, max_text_len=max_text_len or 64, stash_layer=stash_layer, device=dev, mode=mode)
        sd, o = {}, 0
        for n, shp in shapes.items():
            k = int(np.prod(shp))
            sd[n] = flat[o:o + k].view(*shp)
            o += k
        eng.load_state_dict(sd)
        return BlipITM(cfg, eng, tok)
    init = synth.synth_state_dict(cfg, seed)                      # stands in for the module's initialisation
    if checkpoint:
        sd = torch.load(checkpoint, map_location="cpu")
        sd = sd["model"] if "model" in sd else sd
        raw_pos = sd.get("visual_encoder.pos_embed")
        state, dropped, missing = merge_checkpoint(cfg, sd, init)
        if dropped or missing:
            warnings.warn(f"checkpoint {checkpoint}: dropped (shape mismatch) {dropped}, not provided {missing}: "
                          f"those tensors keep their initial values, like load_state_dict(strict=False)")
    else:
        warnings.warn("no BLIP checkpoint given (PNP_OVSS_CHECKPOINT): using seeded synthetic weights")
        state, raw_pos = init, None
    if eager:
        if sync is not None:
            # one flat device buffer in param_shapes order (what receive_only ranks allocate): the driver's collective runs on it
            shapes = synth.param_shapes(cfg)
            torch.cuda.set_device(dev)
            flat = torch.empty(sum(int(np.prod(s)) for s in shapes.values()), dtype=torch.float32, device=torch.device("cuda", dev))
            o, sd = 0, {}
            for n, shp in shapes.items():
                k = int(np.prod(shp))
                t = state[n]
                flat[o:o + k].copy_(torch.as_tensor(np.asarray(t) if not isinstance(t, torch.Tensor) else t, dtype=torch.float32).reshape(-1))
                sd[n] = flat[o:o + k].view(*shp)
                o += k
            sync(flat)
            state = sd
        eng = Engine(cfg, max_batch=max_batch, max_text_len=max_text_len or 64, stash_layer=stash_layer, device=dev, mode=mode)
        eng.load_state_dict(state)
        return BlipITM(cfg, eng, tok)
    # lazy: fp32 weights in device buffers (what DDP broadcasts), the checkpoint's own pos_embed grid kept for re-tiling
    torch.cuda.set_device(dev)
    index, total = {}, 0
    for n, t in state.items():
        if n == "visual_encoder.pos_embed":
            continue
        shp = tuple(int(x) for x in np.shape(t))
        index[n] = (total, shp)
        total += int(np.prod(shp))
    flat = torch.empty(total, dtype=torch.float32)
    for n, (o, shp) in index.items():
        flat[o:o + int(np.prod(shp))].copy_(torch.as_tensor(np.asarray(state[n]) if not isinstance(state[n], torch.Tensor) else state[n],
                                                             dtype=torch.float32).reshape(-1))
    pos = raw_pos if raw_pos is not None else state["visual_encoder.pos_embed"]
    model = BlipITM(cfg, None, tok, lazy=dict(mode=mode, index=index, max_batch=int(max_batch or 35)), device=dev)
    model.register_buffer("weights_flat", flat.to(model._device), persistent=False)
    model.register_buffer("pos_embed_raw", torch.as_tensor(np.asarray(pos) if not isinstance(pos, torch.Tensor) else pos,
                                                           dtype=torch.float32).to(model._device), persistent=False)
    return model


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
        self._reserve = (max_pixels_per_image, max_channels, crf_chunk)
        self._reserved_eng = None
        self._reserve_on(eng)
        self.hist_1drop = torch.zeros(n_class * n_class, device=eng.device, dtype=torch.int64)
        self.hist_ndrop = torch.zeros(n_class * n_class, device=eng.device, dtype=torch.int64)

    def _reserve_on(self, eng):
        """Post-processing workspace of `eng` (a lazily built model may replace its engine between launches, e.g. a
        (layer, head) sweep that reaches below the kept layers: the new engine needs its own reserve and prepare)."""
        if self._reserved_eng is not eng:
            px, ch, chunk = self._reserve
            eng.post_reserve(eng.max_batch, eng.max_batch * px, px, ch, chunk)
            self._reserved_eng = eng
            self._prepared = None

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
        m = self.m
        h1, hn = hists if hists is not None else (self.hist_1drop, self.hist_ndrop)
        g0, agg = drop_loop(args, m, prep["tok500"], imgs_in, prep["captions"])
        eng = m.engine                                              # read AFTER the drop loop: it may have re-created the engine
        self._reserve_on(eng)
        if self.coco and int(args.drop_iter) >= 3:
            run_1drop = False
        scale01 = (True, self.coco)                                 # Scale_0_1 on (1-drop, N-drop)
        if getattr(self, "_prepared", None) is not prep:            # once per batch and engine: a (layer, head) sweep re-launches the same prep
            eng.post_prepare(prep["sizes"], prep["plans"], prep["luts"], prep["bgs"], rgb=prep["rgb"], gt=prep["gt"],
                             want_crf=bool(self.mode and "crf" in self.mode))
            self._prepared = prep
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
