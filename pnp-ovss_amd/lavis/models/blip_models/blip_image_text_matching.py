"""`compute_gradcam_ensemble` under its reference import path
(Files to replace for BLIP/blip_image_text_matching.py:386)."""
from pnp_ovss.model import BlipITM, compute_gradcam_ensemble  # noqa: F401
