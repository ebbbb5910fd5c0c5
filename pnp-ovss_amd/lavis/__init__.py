"""Import-path shim: the reference driver does `from lavis.models import load_model_and_preprocess`
and `from lavis.models.blip_models.blip_image_text_matching import compute_gradcam_ensemble`
(PnP_OVSS_0514_updated_segmentation.py:4-8).  Putting `pnp-ovss_amd/` on sys.path makes those
imports resolve to the MI355X engine instead of an un-vendored LAVIS checkout."""
