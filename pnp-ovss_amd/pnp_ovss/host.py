"""Host-side logic of the hot path (pure Python, no device work): the pieces of the reference driver
that are string / table manipulation and therefore stay on the host, restated behind the same
names and argument meaning.  PnP.py = PnP_OVSS_0514_updated_segmentation.py.

  Load_predicted_classes   PnP.py:726-787  -> parse_gpt_classes / load_predicted_classes
  Mean_over_filtered_label_tokens (the token walk, PnP.py:812-853) -> merge_plan
  index -> class-id remap  PnP.py:390-399 / 468-480 -> remap_lut (folded into a lookup table)
  background rule          PnP.py:373-379 -> has_background
  DistributedSampler split LD.py:25 -> shard_indices
  scores / mIoU            PnP.py:1115-1146, Calculate_mIoU.py:204-256 -> scores_from_hist
COCO driver (PnPc.py = PnP_OVSS_0514_updated_segmentation_coco.py): parse_gpt_classes_coco (:858-963), the
background rule of :446-450 / :470-473 and the cats[..]['id'] remap of :458-463 / :482-489 (remap_lut(class_ids=)).
"""
import json

import numpy as np

VOC_CATS = {1: "aeroplane", 2: "bicycle", 3: "bird", 4: "boat", 5: "bottle", 6: "bus", 7: "car", 8: "cat", 9: "chair",
            10: "cow", 11: "table", 12: "dog", 13: "horse", 14: "motorbike", 15: "person", 16: "pottedplant",
            17: "sheep", 18: "sofa", 19: "train", 20: "tvmonitor"}                  # Load_datasets.py:8-10


def parse_gpt_classes(per_img_cls: str, nms):
    """GPT-4o string "[id: 'name', ...], [p%, ...]" -> (best_class_idx, names, caption).
    Keeps classes with probability > 70, falls back to class 0, raises on unsplittable strings
    exactly like the reference (PnP.py:746-783)."""
    parts = per_img_cls.replace(']\n\n[', '], [').replace('],\n\n[', '], [').replace('], \n[', '], [ ') \
        .replace(']\n[', '], [ ').replace('],\n[', '], [ ').strip("][").split("], [")
    cls_list = parts[0].split(",")
    if len(parts) == 1 and parts[0] == '':
        cls_list = ["1: 'wall'" for _ in range(len(cls_list))]
        prob_list = [100 for _ in range(len(cls_list))]
    else:
        prob_list = [int(p.split(":")[-1].split("%")[0]) for p in parts[1].split(",")]
    idx = [int(cls_list[i].split(":")[0]) for i, p in enumerate(prob_list) if p > 70]
    best = [i - 1 for i in idx]
    names = [nms[i - 1] for i in idx]
    if not best:
        best, names = [0], [nms[0]]
    return best, names, "A picture of " + " ".join(names)


def parse_gpt_classes_coco(per_img_cls: str, cats, nms, data_type="coco_object"):
    """COCO variant (PnPc.py:870-905 coco_object, :918-958 coco_stuff): the number GPT-4o printed is a COCO category
    id, best_class_idx is its position in `cats` (list of {'id', 'name'}, pycocotools order); ids that are no
    category are skipped; a string without the probability list keeps every class ("no prob output")."""
    parts = per_img_cls.replace(']\n\n[', '], [').replace('],\n\n[', '], [').replace('], \n[', '], [ ') \
        .replace('],\n[', '], [ ').replace(']\n[', '], [ ').strip("][").split("], [")
    cls_list = parts[0].split(",")
    if len(parts) == 1 and parts[0] == '':
        cls_list = ["1: 'person'" for _ in range(len(cls_list))]
        prob_list = [100 for _ in range(len(cls_list))]
    elif len(parts) == 1:
        prob_list = [100 for _ in range(len(cls_list))]
    else:
        prob_list = [int(p.split(":")[-1].split("%")[0]) for p in parts[1].split(",")]
    stuff = data_type == "coco_stuff"
    if stuff:
        prob_list = prob_list[:len(cls_list)]
    ids = []
    for i, p in enumerate(prob_list):
        if p > 70:
            try:
                ids.append(int(cls_list[i].split(":")[0]))
            except Exception:              # noqa: BLE001
                if not stuff:              # only the coco_stuff branch swallows malformed entries (:943-946)
                    raise
    pos = {c["id"]: j for j, c in reversed(list(enumerate(cats)))}      # first match wins, like the reference loop
    best = [pos[v] for v in ids if v in pos]
    names = [nms[j] for j in best]
    if not best:
        best, names = [0], [nms[0]]
    return best, names, "A picture of " + " ".join(names)


class GptClassTable:
    """One of GPT4o_classification/*.json, read once (the reference re-opens it per image)."""

    def __init__(self, path, data_type):
        with open(path) as f:
            self.table = json.load(f)
        self.data_type = data_type

    def lookup(self, img_id, nms, cats=None):
        if self.data_type.startswith("coco"):
            return parse_gpt_classes_coco(self.table[str(int(img_id)).rjust(12, "0")], cats, nms, self.data_type)
        key = "ADE_val_" + str(img_id).rjust(8, "0") if self.data_type == "ade20k" else str(img_id)
        return parse_gpt_classes(self.table[key], nms)


def merge_plan(pieces, n_classes):
    """Token walk of Mean_over_filtered_label_tokens (PnP.py:820-853) as a table: for every class the
    list of word-piece rows (indices into map[3:-1]) that are summed in order and the divisor applied
    afterwards (1 = none; a trailing split word is summed but NOT averaged, like the reference).
    #pieces == #classes is the reference's fast path map[:C]."""
    if len(pieces) == n_classes:
        return [([i], 1) for i in range(n_classes)]
    plan = [([], 1) for _ in range(n_classes)]
    it, ic, wl, n = 0, 0, 1, len(pieces)
    while it < n:
        nxt_plain = it + 1 < n and not pieces[it + 1].startswith("##")
        if not pieces[it].startswith("##"):
            plan[ic] = ([it], 1)                   # IndexError on overflow mirrors the reference
            if nxt_plain:
                ic += 1
            wl = 1
        else:
            wl += 1
            toks, _ = plan[ic]
            plan[ic] = (toks + [it], wl if nxt_plain else 1)
            if nxt_plain:
                ic += 1
        it += 1
    return plan


def caption_pieces(tokenizer, input_ids_row):
    """Decoded word pieces between '[ENC] a picture of' and [SEP] (PnP.py:812-818)."""
    out = []
    for t in input_ids_row[1:]:
        t = int(t)
        if t == 102:
            break
        out.append(tokenizer.decode([t]))
    return out[3:]


def has_background(data_type, n_selected):
    """PnP.py:373-379: object datasets always get a background channel, context datasets only when
    fewer than 3 classes were selected.  COCO (PnPc.py:446-450 / :470-473): coco_object always,
    coco_stuff like the context datasets."""
    return data_type in ("voc", "coco_object") or n_selected < 3


def remap_lut(best_class_idx, with_background, n_channels, class_ids=None):
    """Fold the in-place, descending, collision-prone remap of PnP.py:390-399 into a table
    argmax index -> dataset class id (every pixel value follows the same chain of rewrites).
    COCO (PnPc.py:458-463, :482-489, :549-556, :577-584): the target is cats[best_class_idx[i]]['id'],
    passed as class_ids[j] = cats[j]['id']."""
    lut = []
    for v0 in range(n_channels):
        v = v0
        for i in range(len(best_class_idx) - 1, -1, -1):
            src = i + 1 if with_background else i
            if v == src:
                v = best_class_idx[i] + 1 if class_ids is None else class_ids[best_class_idx[i]]
        lut.append(int(v))
    return lut


def coco_n_class(data_type):
    """PnPc.py:597-600: the confusion matrix spans category ids, not positions."""
    return 91 if data_type == "coco_object" else 183


def coco_class_names(cats):
    """PnPc.py:1399-1400: class names with blanks and dashes removed."""
    return ["".join("".join(c["name"].split(" ")).split("-")) for c in cats]


def shard_indices(n, rank, world_size, seed=0, shuffle=True):
    """torch DistributedSampler(dataset) defaults (LD.py:25): seed-0 permutation, padded with the
    leading indices to a multiple of world_size, rank takes indices[rank::world_size]."""
    import torch
    if shuffle:
        g = torch.Generator()
        g.manual_seed(seed)
        idx = torch.randperm(n, generator=g).tolist()
    else:
        idx = list(range(n))
    total = -(-n // world_size) * world_size
    pad = total - len(idx)
    if pad > 0:
        idx += (idx * (-(-pad // len(idx))))[:pad]
    return idx[rank:total:world_size]


def scores_from_hist(hist):
    """PnP.py:1124-1131 / Calculate_mIoU.py:230-256 on an accumulated confusion matrix."""
    hist = np.asarray(hist, dtype=np.float64)
    with np.errstate(all="ignore"):
        acc = np.diag(hist).sum() / hist.sum()
        acc_cls = np.nanmean(np.diag(hist) / hist.sum(axis=1))
        iu = np.diag(hist) / (hist.sum(axis=1) + hist.sum(axis=0) - np.diag(hist))
        valid = hist.sum(axis=1) > 0
        mean_iu = np.nanmean(iu[valid])
        freq = hist.sum(axis=1) / hist.sum()
        fwavacc = (freq[freq > 0] * iu[freq > 0]).sum()
    return {"Pixel Accuracy": acc, "Mean Accuracy": acc_cls, "Frequency Weighted IoU": fwavacc, "Mean IoU": mean_iu,
            "Class IoU": iu}
