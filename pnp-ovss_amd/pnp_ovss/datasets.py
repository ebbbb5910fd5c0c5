"""Input side of the hot path (host, row a-15 of SURVEY.md §8): the tensors the reference's datasets
deliver to the driver (`img0`, `img_id`, original RGB for the CRF, ground-truth label map), sharded
across ranks like `DataLoader(..., sampler=DistributedSampler(dataset))` (Load_datasets.py:15-26).

JPEG files are decoded on the device per batch (hip.jpeg_decode_batch -> csrc/jpeg.hip: Pillow-exact baseline decoder;
`--device_jpeg 0` or a file the decoder does not cover -> Pillow on the host); PNG ground truth stays on the host.
  synthetic  seeded images + one-word-piece-per-class captions (no files needed; bench / CI)
  voc        Dataset.py:349-445 (bicubic resize to img_size, CLIP mean/std: on device, bit-identical to the
             Pillow / torchvision host path -- pnp_preprocess_images), PnP.py:901-955 (GT / RGB)
  psc        Dataset.py:889-991 (same transform), Pascal-Context 59 classes (Load_datasets.py:30-44)
  ade20k     Dataset.py:1181-1296 (PIL *bilinear* resize, ToTensor only), Load_datasets.py:60-104 (150 classes, names
             with blanks removed, validation.odgt list), PnP.py:917-923 / 945-952 (GT / RGB by "ADE_val_%08d")
  coco_object / coco_stuff   the COCO driver (PnPc.py = PnP_OVSS_0514_updated_segmentation_coco.py): Dataset.py:1373-1491
             (same bicubic + CLIP-normalise transform), categories / image list / instance annotations read from the
             annotation JSON with the standard library (pycocotools' COCO index: PnPc.py:1376-1400, Dataset.py:1341-1345),
             PnPc.py:1095-1125 (GT: instance masks painted in annotation order for coco_object -- `coco_mask.ann_to_mask`
             restates pycocotools' annToMask --, stuff PNG + 1 with 255 -> 0 for coco_stuff), PnPc.py:1127-1138 (RGB)
"""
import json
import os

import numpy as np
import torch

from . import host, synth

PSC_NAMES = ("aeroplane bag bed bedclothes bench bicycle bird boat book bottle building bus cabinet car cat ceiling "
             "chair cloth computer cow cup curtain dog door fence floor flower food grass ground horse keyboard light "
             "motorbike mountain mouse person plate platform pottedplant road rock sheep shelves sidewalk sign sky snow "
             "sofa table track train tree truck tvmonitor wall water window wood").split()


ADE_NAMES = ("wall,building,sky,floor,tree,ceiling,road,bed,windowpane,grass,cabinet,sidewalk,person,ground,door,table,mountain,"
             "plant,curtain,chair,car,water,painting,sofa,shelf,house,sea,mirror,rug,field,armchair,seat,fence,desk,rock,wardrobe,"
             "lamp,bathtub,railing,cushion,base,box,pillar,signboard,chest of drawers,counter,sand,sink,skyscraper,fireplace,"
             "refrigerator,grandstand,path,stairs,runway,case,billiard table,pillow,screen,stairway,river,bridge,bookcase,blind,"
             "coffee table,toilet,flower,book,hill,bench,countertop,stove,palm,kitchen island,computer,swivel chair,boat,bar,"
             "arcade machine,hovel,bus,towel,light,truck,tower,chandelier,sunshade,streetlight,booth,television receiver,airplane,"
             "dirt track,apparel,pole,land,bannister,escalator,ottoman,bottle,buffet,poster,stage,van,ship,fountain,conveyer belt,"
             "canopy,washer,toy,swimming pool,stool,barrel,basket,waterfall,tent,bag,motorbike,cradle,oven,ball,food,stair,tank,"
             "marque,microwave,pot,animal,bicycle,lake,dishwasher,screen,blanket,sculpture,hood,sconce,vase,trafficlight,tray,"
             "trash can,fan,pier,crt screen,plate,monitor,bulletinboard,shower,radiator,glass,clock,flag").split(",")


class _Base:
    max_text_len = 64
    max_channels = 24
    class_ids = None          # COCO: cats[j]['id'] (labels / confusion matrix are in category-id space)

    def __init__(self, args, rank, world_size, cats):
        self.args, self.rank, self.world = args, rank, world_size
        self.cats = cats
        self.nms = list(cats.values())
        n = len(cats) + 1
        self.total_hist = np.zeros((n, n), dtype=np.float64)

    def _indices(self, n):
        return host.shard_indices(n, self.rank, self.world, seed=0, shuffle=True)

    resample = ("bicubic", synth.CLIP_MEAN, synth.CLIP_STD)      # Dataset.py:434-443

    def _decode_on_device(self, items):
        """Items whose image is still a JPEG byte string are decoded together on the GPU; a file the device decoder does
        not cover (progressive, arithmetic, CMYK) is decoded by Pillow on the host, loudly."""
        todo = [k for k, it in enumerate(items) if isinstance(it[2], (bytes, bytearray))]
        if not todo:
            return items
        import io
        import warnings
        from . import hip
        from .jpeg import UnsupportedJpeg
        items = [list(it) for it in items]
        good = []
        def on_host(k, why):
            from PIL import Image
            warnings.warn(f"image {items[k][1]}: {why}: decoded on the host")
            items[k][2] = np.asarray(Image.open(io.BytesIO(items[k][2])).convert("RGB"))

        for k in todo:
            try:
                from . import jpeg as J
                J.parse(items[k][2])
                good.append(k)
            except UnsupportedJpeg as exc:
                on_host(k, exc)
        if good:
            try:
                dec = hip.jpeg_decode_batch([items[k][2] for k in good])
                for k, t in zip(good, dec):
                    items[k][2] = t
            except RuntimeError as exc:          # a corrupt entropy stream fails the whole device batch: file by file on the host
                for k in good:
                    on_host(k, f"device decode of the batch failed ({exc})")
        if any(isinstance(it[2], np.ndarray) for it in items) and any(isinstance(it[2], torch.Tensor) for it in items):
            for it in items:                                       # mixed batch: everything onto the device
                if isinstance(it[2], np.ndarray):
                    it[2] = torch.from_numpy(np.ascontiguousarray(it[2])).cuda()
        return [tuple(it) for it in items]

    def batches(self, batch_size):
        """Items whose first element is None get their model tensor from the device-side resize + normalise
        (hip.preprocess_images: Pillow-exact, Dataset.py:434-443 / :1263) over the decoded RGB the CRF uses anyway."""
        if torch.cuda.is_available():
            # this generator may run in a prefetch thread; device = rank unless the driver says otherwise (--share_gpu)
            torch.cuda.set_device(int(getattr(self.args, "device_index", self.rank)))
        idx = self._indices(len(self))
        # JPEG / PNG decode of a batch on a small thread pool (Pillow releases the GIL while decoding): at ~300 images/s
        # per GPU one decoding thread (2-4 ms per VOC-sized image) would be the bottleneck (the reference decodes on the
        # main thread, num_workers = 0, PnP.py:61)
        workers = int(getattr(self.args, "num_workers", 0) or 0) or min(8, os.cpu_count() or 1)
        pool = None
        if workers > 1 and not isinstance(self, SyntheticDataset):
            from concurrent.futures import ThreadPoolExecutor
            pool = ThreadPoolExecutor(max_workers=workers)
        for o in range(0, len(idx), batch_size):
            ids = idx[o:o + batch_size]
            items = list(pool.map(self.__getitem__, ids)) if pool else [self[i] for i in ids]
            items = self._decode_on_device(items)
            if items[0][0] is None:
                from . import hip
                filt, mean, std = self.resample
                imgs = hip.preprocess_images([it[2] for it in items], self.args.img_size, mean, std, filt=filt)
            else:
                imgs = torch.stack([it[0] for it in items])
            batch = {"imgs": imgs, "img_ids": [it[1] for it in items],
                     "org_images": [it[2] for it in items], "label_trues": [it[3] for it in items]}
            if torch.cuda.is_available() and all(it[3] is not None for it in items):
                # the concatenated ground truth goes up here (this generator usually runs in the prefetch thread), not on
                # the driver's critical path between the drop loop and the post-processing
                batch["gt_dev"] = torch.from_numpy(np.concatenate([np.asarray(it[3], dtype=np.float32).reshape(-1) for it in items])).cuda()
            if torch.cuda.is_available():
                # the device tensors above (decoded RGB, resized + normalised images, ground truth) were produced on THIS
                # thread's current stream; a consumer that works on another stream (CLI --pipelines workers) must order
                # itself behind this event before it touches them (wait_ready)
                batch["ready"] = torch.cuda.Event()
                batch["ready"].record()
            yield batch


def wait_ready(batch):
    """Order the caller's current stream behind the producer of `batch` (see batches): no host synchronisation."""
    ev = batch.get("ready")
    if ev is not None:
        torch.cuda.current_stream().wait_event(ev)


def _read_rgb(path, device_jpeg):
    """The image as the datasets hand it on: the JPEG file's bytes when the batch is decoded on the device
    (hip.jpeg_decode_batch), else `Image.open(path).convert('RGB')` as a numpy array (Dataset.py:349-445)."""
    if device_jpeg and path.lower().endswith((".jpg", ".jpeg")):
        with open(path, "rb") as f:
            return f.read()
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGB"))


def _device_jpeg(args):
    return bool(getattr(args, "device_jpeg", True)) and torch.cuda.is_available()


class SyntheticDataset(_Base):
    """`--data_type synthetic`: 336^2-style seeded images, all-VOC-class captions."""

    def __init__(self, args, rank, world_size):
        super().__init__(args, rank, world_size, dict(host.VOC_CATS))
        self.n = int(args.synthetic_images)
        self.max_pixels = args.img_size * args.img_size
        self.names = [f"c{i}" for i in range(20)]          # short names: one word piece each with SynthTokenizer

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        S = self.args.img_size
        rgb, img = synth.synth_images(1, S, seed=10_000 + i, noise=4)
        gt = np.random.default_rng(i).integers(0, 21, size=(S, S)).astype(np.float32)
        return torch.from_numpy(img[0]), f"{2007 + i % 6}_{i:06d}", rgb[0], gt

    def predicted_classes(self, img_id):
        k = 1 + int(img_id.split("_")[1]) % 20
        best = list(range(k))
        return best, [self.names[b] for b in best], "A picture of " + " ".join(self.names[b] for b in best)


class VocLikeDataset(_Base):
    def __init__(self, args, rank, world_size, kind):
        if kind == "psc":                                        # 59 classes: up to 59 channels, ~2.4 word pieces per name
            self.max_channels, self.max_text_len = 60, 160
        from PIL import Image                                    # noqa: F401  (fail early if missing)
        cats = dict(host.VOC_CATS) if kind == "voc" else {i + 1: n for i, n in enumerate(PSC_NAMES)}
        super().__init__(args, rank, world_size, cats)
        self.kind = kind
        home = args.home_dir
        if kind == "voc":
            self.img_dir = f"{home}/VOCdevkit/VOC2012/JPEGImages"
            self.gt_dir = f"{home}/VOCdevkit/VOC2012/SegmentationClass"
            split = f"{home}/VOCdevkit/VOC2012/ImageSets/Segmentation/val.txt"
        else:
            self.img_dir = f"{home}/VOCdevkit/VOC2012/JPEGImages"
            self.gt_dir = f"{home}/mmsegmentation/data/VOCdevkit/VOC2010/SegmentationClassContext"
            split = f"{home}/mmsegmentation/data/VOCdevkit/VOC2010/ImageSets/SegmentationContext/val.txt"
        with open(split) as f:
            self.ids = [l.strip() for l in f if l.strip()]
        self.gpt = host.GptClassTable(f"{home}/GPT4o_classification/{kind}_classification_noboundary.json", kind)
        self.max_pixels = 512 * 512
        self.mean = np.array(synth.CLIP_MEAN, dtype=np.float32).reshape(3, 1, 1)
        self.std = np.array(synth.CLIP_STD, dtype=np.float32).reshape(3, 1, 1)

    def __len__(self):
        return len(self.ids)

    def __getitem__(self, i):
        from PIL import Image
        img_id = self.ids[i]
        org = _read_rgb(os.path.join(self.img_dir, img_id + ".jpg"), _device_jpeg(self.args))    # bytes -> decoded per batch on the GPU
        gt = np.float32(Image.open(os.path.join(self.gt_dir, img_id + ".png")))
        if self.kind == "voc":
            gt[gt == 255] = 0                                    # PnP.py:908
        return None, img_id, org, gt                             # tensor: device resize + normalise per batch

    def predicted_classes(self, img_id):
        return self.gpt.lookup(img_id, self.nms)


class Ade20kDataset(_Base):
    """`--data_type ade20k`: 150 classes, 768-pixel inputs, up to 150 channels per image."""
    max_text_len = 192
    max_channels = 152
    resample = ("bilinear", (0.0, 0.0, 0.0), (1.0, 1.0, 1.0))    # imresize(..., 'bilinear') + ToTensor only (Dataset.py:1263-1275)

    def __init__(self, args, rank, world_size):
        from PIL import Image                                    # noqa: F401
        super().__init__(args, rank, world_size, {i + 1: n for i, n in enumerate(ADE_NAMES)})
        self.nms = ["".join(n.split(" ")) for n in self.cats.values()]           # Load_datasets.py:87
        home = args.home_dir
        with open(f"{home}/semantic-segmentation-pytorch-master/data/validation.odgt") as f:
            self.records = [json.loads(l) for l in f if l.strip()]
        self.gpt = host.GptClassTable(f"{home}/GPT4o_classification/ade20k_classification_noboundary.json", "ade20k")
        self.max_pixels = int(getattr(args, "max_pixels", 0)) or 1024 * 1024     # reserve bound per image; larger images fail loudly

    def __len__(self):
        return len(self.records)

    def __getitem__(self, i):
        from PIL import Image
        rec = self.records[i]
        img_id = rec["fpath_img"].split(".")[0].split("/")[-1].split("_")[-1].lstrip("0")     # Dataset.py:1270
        stem = "ADE_val_" + img_id.rjust(8, "0")
        home = self.args.home_dir
        org = _read_rgb(f"{home}/ADEChallengeData2016/images/validation/{stem}.jpg", _device_jpeg(self.args))
        gt = np.float32(Image.open(f"{home}/ADEChallengeData2016/annotations/validation/{stem}.png"))      # PnP.py:917-923
        return None, img_id, org, gt

    def predicted_classes(self, img_id):
        return self.gpt.lookup(img_id, self.nms)


class CocoDataset(_Base):
    """`--data_type coco_object | coco_stuff` (the reference's second driver script).  Everything pycocotools' COCO
    object provides to that driver is read from the two annotation files with `json`: `cats` in file order
    (loadCats(getCatIds()), PnPc.py:1383-1395), the image list in file order (Dataset.py:1345), per-image
    instance annotations (PnPc.py:1100-1110)."""
    max_text_len = 192
    max_channels = 96

    def __init__(self, args, rank, world_size, kind):
        from PIL import Image                                    # noqa: F401
        home = args.home_dir
        with open(f"{home}/coco/annotations/instances_val2017.json") as f:
            thing = json.load(f)
        cats = [{"id": c["id"], "name": c["name"]} for c in thing["categories"]]
        if kind == "coco_stuff":
            with open(f"{home}/coco/annotations/stuff_val2017.json") as f:
                cats += [{"id": c["id"], "name": c["name"]} for c in json.load(f)["categories"]]
            self.max_channels = 184
            self.max_text_len = 320          # 171 class names with their word-piece splits (the text kernels take up to 512 tokens)
        self.kind = kind
        self.rank, self.world, self.args = rank, world_size, args
        self.cats = cats                                          # list of dicts, like the reference's `cats`
        self.nms = host.coco_class_names(cats)
        self.class_ids = [c["id"] for c in cats]
        n = host.coco_n_class(kind)
        self.total_hist = np.zeros((n, n), dtype=np.float64)
        self.images = thing["images"]
        self.anns = {}
        if kind == "coco_object":
            for a in thing["annotations"]:                        # pycocotools imgToAnns: file order per image
                self.anns.setdefault(a["image_id"], []).append(a)
        self.gpt = host.GptClassTable(f"{home}/GPT4o_classification/{kind}_classification_noboundary.json", kind)
        self.max_pixels = 640 * 640

    def __len__(self):
        return len(self.images)

    def __getitem__(self, i):
        from PIL import Image
        from . import coco_mask
        rec = self.images[i]
        home = self.args.home_dir
        org = _read_rgb(f"{home}/coco/images/val2017/{rec['file_name']}", _device_jpeg(self.args))
        if self.kind == "coco_object":                            # PnPc.py:1099-1110: first annotation wins a pixel
            gt = np.zeros((rec["height"], rec["width"]), dtype=np.float32)
            for a in self.anns.get(rec["id"], []):
                m = coco_mask.ann_to_mask(a, rec["height"], rec["width"])
                gt[np.logical_and(m, gt == 0)] = a["category_id"]
        else:                                                     # PnPc.py:1112-1122
            gt = np.float32(Image.open(f"{home}/coco_stuff164k/annotations/val2017/{int(rec['id']):012d}.png"))
            gt = np.where(gt == 255, np.float32(0), gt + 1).astype(np.float32)
        return None, int(rec["id"]), org, gt

    def predicted_classes(self, img_id):
        return self.gpt.lookup(img_id, self.nms, self.cats)


def prefetch(iterable, depth=2):
    """Run `iterable` in a background thread, `depth` items ahead: JPEG decode / ground-truth loading / the device-side
    resize of batch i+1 overlap the model and CRF work of batch i (the reference's DataLoader runs with
    num_workers=0 on the main thread, PnP.py:61).  Exceptions of the producer are re-raised in the consumer; when the
    consumer stops early (break, exception, generator close) the producer is told to stop and joined, so no thread
    is left blocked on the queue holding device tensors."""
    import queue
    import threading
    q = queue.Queue(maxsize=max(1, depth))
    end = object()
    stop = threading.Event()

    def put(item):
        while not stop.is_set():
            try:
                q.put(item, timeout=0.1)
                return True
            except queue.Full:
                continue
        return False

    def work():
        try:
            for item in iterable:
                if not put(item):
                    return
            put(end)
        except BaseException as exc:       # noqa: BLE001  (handed to the consumer)
            put(exc)

    t = threading.Thread(target=work, daemon=True)
    t.start()
    try:
        while True:
            item = q.get()
            if item is end:
                return
            if isinstance(item, BaseException):
                raise item
            yield item
    finally:
        stop.set()
        while True:                        # unblock a producer waiting in put()
            try:
                q.get_nowait()
            except queue.Empty:
                break
        t.join(timeout=30)


def make_dataset(args, rank, world_size):
    if args.data_type == "synthetic":
        return SyntheticDataset(args, rank, world_size)
    if args.data_type in ("voc", "psc"):
        return VocLikeDataset(args, rank, world_size, args.data_type)
    if args.data_type == "ade20k":
        return Ade20kDataset(args, rank, world_size)
    if args.data_type in ("coco_object", "coco_stuff"):
        return CocoDataset(args, rank, world_size, args.data_type)
    raise SystemExit(f"--data_type {args.data_type!r}: supported: synthetic, voc, psc, ade20k, coco_object, coco_stuff")
