"""`lavis.models.load_model_and_preprocess` for the one model the hot path uses."""
import numpy as np
import torch

from pnp_ovss.model import build_model


class BlipImageEvalProcessor:
    """Files to replace for BLIP/blip_processors.py:166-194: the eval processor is reduced to
    Normalize only (resize / ToTensor removed); the drivers never apply it (datasets deliver tensors)."""
    mean = (0.48145466, 0.4578275, 0.40821073)
    std = (0.26862954, 0.26130258, 0.27577711)

    def __init__(self, image_size=336):
        self.image_size = image_size

    def __call__(self, x):
        x = torch.as_tensor(np.asarray(x), dtype=torch.float32)
        m = torch.tensor(self.mean).view(3, 1, 1)
        s = torch.tensor(self.std).view(3, 1, 1)
        return (x - m) / s


class BlipCaptionProcessor:
    """blip_processors.py:28-68 (prompt + whitespace clean-up; the file's default cap is 500 words, :30)."""

    def __init__(self, prompt="", max_words=500):
        self.prompt, self.max_words = prompt, max_words

    def __call__(self, caption):
        import re
        c = re.sub(r"([.!\"()*#:;~])", " ", caption.lower())
        c = re.sub(r"\s{2,}", " ", c).rstrip("\n").strip(" ")
        words = c.split(" ")
        if len(words) > self.max_words:
            c = " ".join(words[: self.max_words])
        return self.prompt + c


def load_model_and_preprocess(name, model_type, is_eval=False, device="cpu", **kw):
    """Same call as PnP_OVSS_0514_updated_segmentation.py:1212-1213.  With only the reference's arguments the model
    is built lazily in the parity mode (fp32 arithmetic) and takes its geometry from the first
    compute_gradcam_ensemble(args, ...) call; extra keyword arguments (img_size, max_batch, stash_layer, mode,
    checkpoint, vocab, ...) go to pnp_ovss.model.build_model and size the engine at once."""
    if name != "blip_image_text_matching":
        raise ValueError(f"unknown model {name!r}: only blip_image_text_matching is on the hot path")
    model = build_model(model_type=model_type, device=device, **kw)
    if is_eval:
        model.eval()
    return model, {"eval": BlipImageEvalProcessor(model.cfg.img_size)}, {"eval": BlipCaptionProcessor()}
