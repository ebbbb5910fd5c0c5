"""Generation-time only: import the reference's own hot-path modules from /root/reference
(read-only, exists only in the build container) so golden vectors can be produced from the
reference itself.  Nothing here is used by tests or the product at run time, and no reference
source is copied: un-vendored dependencies (lavis, timm, fairscale) are replaced by the minimal
in-memory stand-ins SURVEY.md §8c lists, and driver-script functions are exec'd straight from the
read-only file by name.
"""
import ast
import importlib.util
import sys
import types

import torch
import torch.nn as nn

REF = "/root/reference"
REF_B = REF + "/Files to replace for BLIP/"
PNP = REF + "/PnP_OVSS_0514_updated_segmentation.py"
PNPC = REF + "/PnP_OVSS_0514_updated_segmentation_coco.py"


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs():
    # transformers probes importlib for "timm": import it before the stand-ins are registered
    import transformers.modeling_utils as mu
    import transformers.pytorch_utils as pu

    class _Registry:
        def register_model(self, name):
            return lambda c: c

        def register_processor(self, name):
            return lambda c: c

    class BaseModel(nn.Module):
        @property
        def device(self):
            return list(self.parameters())[0].device

    class BaseEncoder(nn.Module):
        @property
        def device(self):
            return list(self.parameters())[0].device

    class BlipBase(BaseModel):
        @classmethod
        def init_tokenizer(cls):
            return None

    class PatchEmbed(nn.Module):
        def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
            super().__init__()
            self.img_size = (img_size, img_size)
            self.patch_size = (patch_size, patch_size)
            self.grid_size = (img_size // patch_size, img_size // patch_size)
            self.num_patches = self.grid_size[0] * self.grid_size[1]
            self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

        def forward(self, x):
            return self.proj(x).flatten(2).transpose(1, 2)

    class DropPath(nn.Module):
        def __init__(self, p=0.0):
            super().__init__()

        def forward(self, x):
            assert not self.training
            return x

    _mod("lavis")
    _mod("lavis.common")
    _mod("lavis.common.registry", registry=_Registry())
    _mod("lavis.common.utils", get_abs_path=lambda p: p, is_url=lambda p: False)
    _mod("lavis.models")
    _mod("lavis.models.base_model", BaseModel=BaseModel, BaseEncoder=BaseEncoder)
    _mod("lavis.models.blip_models")
    _mod("lavis.models.blip_models.blip", BlipBase=BlipBase)
    _mod("timm")
    _mod("timm.models")
    _mod("timm.models.vision_transformer", _cfg=lambda **k: k, PatchEmbed=PatchEmbed)
    _mod("timm.models.layers", trunc_normal_=torch.nn.init.trunc_normal_, DropPath=DropPath)
    _mod("timm.models.registry", register_model=lambda f: f)
    _mod("timm.models.helpers", named_apply=None, adapt_input_conv=None)
    _mod("fairscale")
    _mod("fairscale.nn")
    _mod("fairscale.nn.checkpoint")
    _mod("fairscale.nn.checkpoint.checkpoint_activations", checkpoint_wrapper=lambda m: m)
    mu.apply_chunking_to_forward = pu.apply_chunking_to_forward
    mu.prune_linear_layer = pu.prune_linear_layer
    mu.find_pruneable_heads_and_indices = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError())


def _load(name, fn):
    spec = importlib.util.spec_from_file_location(name, REF_B + fn)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


_CACHE = {}


def load_reference_model_modules():
    if "mods" in _CACHE:
        return _CACHE["mods"]
    install_stubs()
    med = _load("lavis.models.med", "med.py")
    vit = _load("lavis.models.vit", "vit.py")
    itm = _load("lavis.models.blip_models.blip_image_text_matching", "blip_image_text_matching.py")
    med.BertPreTrainedModel.init_weights = lambda self: self.apply(self._init_weights)
    med.BertModel.get_head_mask = lambda self, hm, n, *a: [None] * n
    _CACHE["mods"] = (med, vit, itm)
    return _CACHE["mods"]


def build_reference_model(cfg, state_dict_np, tokenizer):
    """Reference BlipITM with `cfg` geometry, loaded with the given numpy state dict (strict)."""
    med, vit, itm = load_reference_model_modules()
    from transformers.models.bert.configuration_bert import BertConfig
    v = vit.VisionTransformerEncoder(img_size=cfg.img_size, patch_size=cfg.patch, embed_dim=cfg.vit_dim,
                                     depth=cfg.vit_depth, num_heads=cfg.vit_heads, drop_path_rate=0.1)
    v.vision_width = cfg.vit_dim
    bc = BertConfig(vocab_size=cfg.vocab, hidden_size=cfg.txt_hidden, num_hidden_layers=cfg.txt_layers,
                    num_attention_heads=cfg.txt_heads, intermediate_size=cfg.txt_inter,
                    hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1,
                    max_position_embeddings=cfg.max_pos, type_vocab_size=2,
                    layer_norm_eps=cfg.txt_ln_eps, pad_token_id=cfg.pad_token_id)
    bc.encoder_width = cfg.vit_dim
    bc.add_cross_attention = True
    bc.add_type_embeddings = False
    t = med.XBertEncoder(config=bc, add_pooling_layer=False)
    itm.BlipITM.init_tokenizer = classmethod(lambda cls: tokenizer)
    m = itm.BlipITM(image_encoder=v, text_encoder=t, embed_dim=256).eval()
    sd = {k: torch.from_numpy(a.copy()) for k, a in state_dict_np.items()}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    allowed = {"vision_proj.weight", "vision_proj.bias", "text_proj.weight", "text_proj.bias",
               "text_encoder.embeddings.position_ids"}
    assert set(missing) <= allowed, missing
    assert not unexpected, unexpected
    return m, itm


def load_driver_functions(names, extra_globals, path=PNP):
    """exec the named top-level functions of a reference driver script (never copied: read from
    the read-only file at generation time) into a fresh namespace."""
    src = open(path).read()
    tree = ast.parse(src)
    ns = dict(extra_globals)
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            code = compile(ast.Module(body=[node], type_ignores=[]), path, "exec")
            exec(code, ns)
    missing = [n for n in names if n not in ns]
    assert not missing, missing
    return ns


class DDPLike:
    """`model_textloc.module` holder (the driver only ever touches `.module`)."""
    def __init__(self, m):
        self.module = m
