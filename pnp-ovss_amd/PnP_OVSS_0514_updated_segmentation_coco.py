#!/usr/bin/env python
"""Drop-in for the reference's second driver, `PnP_OVSS_0514_updated_segmentation_coco.py` (Run_seg_coco.sh):
`--data_type coco_object | coco_stuff`.  Same flags and output files as the reference script; the device work and the
COCO-specific rules (1-drop branch only for drop_iter < 3, Scale_0_1 on the N-drop branch, background rule, category-id
remap, 91 / 183-class confusion matrix) live in pnp_ovss.model.Segmenter / pnp_ovss.datasets.CocoDataset, shared with
the VOC / Pascal-Context / ADE20K command line next to this file."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

import PnP_OVSS_0514_updated_segmentation as _cli  # noqa: E402

if __name__ == "__main__":
    args = _cli.get_args_parser().parse_args()
    if args.data_type not in ("coco_object", "coco_stuff"):
        raise SystemExit("--data_type must be coco_object or coco_stuff (reference: Run_seg_coco.sh)")
    if "RANK" in os.environ:
        _cli.main(int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), args)
    elif args.world_size > 1:
        import torch.multiprocessing as mp
        mp.spawn(_cli.main, args=(args.world_size, args), nprocs=args.world_size)
    else:
        _cli.main(0, 1, args)
