"""Model / run configuration for the PnP-OVSS hot path.

Mirrors the values the reference reads from its yaml + json configs:
  * `Files to replace for BLIP/blip_itm_large.yaml`:6-33  (image_size 336, vit_type large)
  * `Files to replace for BLIP/vit.py`:511-523             (large = 1024 / 24 layers / 16 heads)
  * LAVIS `configs/models/med_large_config.json` (un-vendored; values in SURVEY.md App. B):
    hidden 768, 12 layers, 12 heads, intermediate 3072, encoder_width 1024, LN eps 1e-12,
    max_position_embeddings 512, vocab 30524, add_type_embeddings false.
"""
from dataclasses import dataclass, asdict


@dataclass(frozen=True)
class ModelCfg:
    img_size: int = 336
    patch: int = 16
    vit_dim: int = 1024
    vit_depth: int = 24
    vit_heads: int = 16
    vit_mlp_ratio: int = 4
    vit_ln_eps: float = 1e-6          # vit.py:218
    txt_hidden: int = 768
    txt_layers: int = 12
    txt_heads: int = 12
    txt_inter: int = 3072
    txt_ln_eps: float = 1e-12
    vocab: int = 30524
    max_pos: int = 512
    enc_token_id: int = 30523         # BlipBase.init_tokenizer: "[ENC]"
    sep_token_id: int = 102           # literal at PnP_OVSS_0514_updated_segmentation.py:814
    pad_token_id: int = 0

    @property
    def grid(self) -> int:            # P: patches per side (blip_image_text_matching.py:408)
        return self.img_size // self.patch

    @property
    def n_img_tokens(self) -> int:    # N = P*P + 1 (cls)
        return self.grid * self.grid + 1

    @property
    def head_dim(self) -> int:
        return self.vit_dim // self.vit_heads

    def as_dict(self):
        return asdict(self)


def blip_itm_large(img_size: int = 336) -> ModelCfg:
    """`load_model_and_preprocess('blip_image_text_matching', 'large')` geometry."""
    return ModelCfg(img_size=img_size)


def blip_itm_small(img_size: int = 64) -> ModelCfg:
    """Reduced geometry used by the parity tests: head_dim stays 64 (what the HIP attention
    kernels are tiled for), the text side keeps 12 layers x 12 heads because the reference's
    compute_gradcam_ensemble hard-codes both (blip_image_text_matching.py:388,427)."""
    return ModelCfg(img_size=img_size, vit_dim=128, vit_depth=2, vit_heads=2,
                    txt_hidden=768, txt_layers=12, txt_heads=12, txt_inter=1024,
                    vocab=1024, max_pos=512, enc_token_id=1023)
