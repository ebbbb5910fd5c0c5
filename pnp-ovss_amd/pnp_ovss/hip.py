"""ctypes binding of libpnp_hip.so (include/pnp_hip.h) + a thin `Engine` wrapper.

PyTorch is plumbing here: it owns the device buffers handed to the C ABI (tensor.data_ptr()) and
the current HIP stream.  There is NO fallback: if the library is missing or a call fails this
module raises -- the product path never silently runs on the CPU or in eager PyTorch.
"""
import ctypes as C
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PNP_HIP_LIB") or os.path.join(_HERE, "libpnp_hip.so")     # override: development builds of the library


class PnpConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("img_size", "patch", "vit_dim", "vit_depth", "vit_heads", "vit_mlp_ratio")] + \
               [("vit_ln_eps", C.c_float)] + \
               [(n, C.c_int32) for n in ("txt_hidden", "txt_layers", "txt_heads", "txt_inter")] + \
               [("txt_ln_eps", C.c_float)] + \
               [(n, C.c_int32) for n in ("vocab", "max_pos", "enc_token_id", "max_batch", "max_text_len",
                                         "stash_layer", "compute_bf16", "device")]


class PnpPostBatch(C.Structure):
    _fields_ = [("B", C.c_int32),
                ("H", C.POINTER(C.c_int32)), ("W", C.POINTER(C.c_int32)),
                ("n_classes", C.POINTER(C.c_int32)), ("has_bg", C.POINTER(C.c_int32)),
                ("img_cls_off", C.POINTER(C.c_int32)), ("cls_off", C.POINTER(C.c_int32)),
                ("tok_idx", C.POINTER(C.c_int32)), ("cls_div", C.POINTER(C.c_int32)),
                ("lut", C.POINTER(C.c_int32)), ("lut_stride", C.c_int32),
                ("d_rgb", C.c_void_p), ("d_gt", C.c_void_p),
                ("blur_wts", C.POINTER(C.c_double)), ("blur_wt_off", C.POINTER(C.c_int32))]


_lib = None


def load_library():
    """dlopen the in-tree library.  Raises (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           f"or `make -C pnp-ovss_amd/csrc` (there is no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    sig = {
        "pnp_create": (i32, [C.POINTER(PnpConfig), C.POINTER(vp)]),
        "pnp_create_shared": (i32, [C.POINTER(PnpConfig), vp, C.POINTER(vp)]),
        "pnp_destroy": (None, [vp]),
        "pnp_last_error": (C.c_char_p, [vp]),
        "pnp_workspace_bytes": (C.c_size_t, [C.POINTER(PnpConfig)]),
        "pnp_allocated_bytes": (C.c_size_t, [vp]),
        "pnp_load_weight": (i32, [vp, C.c_char_p, vp, C.POINTER(i64), i32, i32]),
        "pnp_finalize_weights": (i32, [vp]),
        "pnp_vit_forward": (i32, [vp, vp, vp, i32, vp]),
        "pnp_cross_kv": (i32, [vp, i32, vp]),
        "pnp_text_forward_xattn": (i32, [vp, vp, vp, i32, i32, i32, vp, vp]),
        "pnp_xattn_grad": (i32, [vp, i32, i32, vp]),
        "pnp_xattn_grad_layer": (i32, [vp, i32, i32, i32, vp]),
        "pnp_compute_gradcam_layer": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp]),
        "pnp_drop_loop_layer": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp]),
        "pnp_gradcam_gather": (i32, [vp, vp, i32, i32, i32, i32, vp, vp]),
        "pnp_compute_gradcam": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp]),
        "pnp_drop_step": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
        "pnp_drop_loop": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp]),
        "pnp_post_reserve": (i32, [vp, i32, i64, i32, i32, i32]),
        "pnp_post_prepare": (i32, [vp, C.POINTER(PnpPostBatch), i32, vp]),
        "pnp_merge_tokens": (i32, [vp, vp, i32, vp]),
        "pnp_threshold_upsample": (i32, [vp, f32, i32, vp]),
        "pnp_blur_minmax": (i32, [vp, vp]),
        "pnp_densecrf": (i32, [vp, i32, f32, f32, f32, f32, f32, vp]),
        "pnp_remap_hist": (i32, [vp, i32, vp, vp, i32, vp]),
        "pnp_postprocess": (i32, [vp, vp, i32, f32, i32, i32, vp, vp, i32, vp]),
        "pnp_postprocess_pair": (i32, [vp, vp, vp, i32, f32, i32, vp, vp, vp, vp, i32, vp]),
        "pnp_get_buffer": (i32, [vp, C.c_char_p, C.POINTER(vp), C.POINTER(C.c_size_t)]),
        "pnp_profile_enable": (i32, [vp, i32]),
        "pnp_profile_read": (i32, [vp, C.POINTER(i64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        "pnp_profile_read_stage": (i32, [vp, i32, C.POINTER(i64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        "pnp_op_gemm": (i32, [i32, vp, i32, vp, i32, i32, i32, i32, vp, vp, i32, vp, i32, i32, vp]),
        "pnp_op_gemm_ex": (i32, [i32, vp, i32, vp, i32, i32, i32, i32, vp, vp, i32, vp, i32, vp, i32, i32, vp]),
        "pnp_op_gemm_tokcols": (i32, [i32, vp, i32, vp, i32, i32, i32, i32, vp, vp, i32, i32, i32, vp]),
        "pnp_op_vit_attention": (i32, [i32, vp, i32, i32, vp, i32, i32, vp, i32, i32, i32, f32, vp]),
        "pnp_op_vit_attention_x3": (i32, [vp, vp, i32, i32, vp, vp, i32, i32, i32, f32, vp]),
        "pnp_preprocess_images": (i32, [vp, vp, i32, i32, i32, vp, vp, C.POINTER(C.c_float), C.POINTER(C.c_float), vp, vp]),
        "pnp_op_layernorm": (i32, [vp, vp, vp, f32, i32, i32, vp, vp]),
        "pnp_op_cast": (i32, [i32, vp, vp, i64, vp]),
        "pnp_jpeg_decode": (i32, [vp, vp, vp, vp, i32, i32, vp, vp, vp, i64, vp, vp, i32, i32, vp, vp]),
        "pnp_op_split": (i32, [vp, vp, vp, i64, vp]),
        "pnp_op_gemm_x3": (i32, [vp, vp, i32, vp, vp, i32, i32, i32, i32, vp, i32, vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, vp]),
        "pnp_op_gemm_x3a": (i32, [vp, i32, vp, vp, i32, i32, i32, i32, vp, vp, i32, vp, i32, i32, vp, i32, vp]),
        "pnp_dbg_gemm_stamps": (i32, [vp, i32]),
        "pnp_op_xattn": (i32, [i32, i32, vp, i32, vp, i32, i32, vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, vp]),
        "pnp_op_sort_pairs": (i32, [vp, vp, vp, vp, i64, i32, i32, vp, i32, vp]),
        "pnp_op_scan_i32": (i32, [vp, vp, i64, i32, vp]),
        "pnp_set_tuning": (i32, [C.c_char_p, i32]),
        "pnp_streamk_status": (i32, [vp, C.POINTER(i64), C.POINTER(C.c_uint32)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)          # AttributeError here = ABI drift, fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


EXPORTED = ["pnp_create", "pnp_create_shared", "pnp_destroy", "pnp_last_error", "pnp_workspace_bytes", "pnp_load_weight",
            "pnp_finalize_weights", "pnp_vit_forward", "pnp_text_forward_xattn", "pnp_xattn_grad",
            "pnp_gradcam_gather", "pnp_compute_gradcam", "pnp_drop_step", "pnp_drop_loop", "pnp_post_reserve",
            "pnp_post_prepare", "pnp_merge_tokens", "pnp_threshold_upsample", "pnp_blur_minmax", "pnp_densecrf",
            "pnp_remap_hist", "pnp_postprocess", "pnp_postprocess_pair", "pnp_get_buffer", "pnp_profile_enable", "pnp_profile_read", "pnp_op_gemm", "pnp_op_gemm_ex", "pnp_op_layernorm", "pnp_op_cast", "pnp_op_xattn", "pnp_dbg_gemm_stamps", "pnp_op_gemm_tokcols", "pnp_op_vit_attention", "pnp_preprocess_images",
            "pnp_cross_kv", "pnp_profile_read_stage", "pnp_op_split", "pnp_op_gemm_x3", "pnp_op_gemm_x3a",
            "pnp_xattn_grad_layer", "pnp_compute_gradcam_layer", "pnp_drop_loop_layer", "pnp_allocated_bytes",
            "pnp_op_vit_attention_x3", "pnp_jpeg_decode", "pnp_op_sort_pairs", "pnp_op_scan_i32", "pnp_set_tuning",
            "pnp_streamk_status"]


class _DevView:
    """Expose a raw device pointer to torch through __cuda_array_interface__."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "device-contiguous tensor required"
    return C.c_void_p(t.data_ptr())


def _i32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


class PnpPreImage(C.Structure):
    _fields_ = [("src_off", C.c_int64), ("tmp_off", C.c_int64), ("H", C.c_int32), ("W", C.c_int32),
                ("kx_off", C.c_int32), ("kx_size", C.c_int32), ("ky_off", C.c_int32), ("ky_size", C.c_int32)]


_RESAMPLE_CACHE = {}


def resample_table(in_size, out_size, filt="bicubic"):
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc for the bicubic / bilinear filter over the whole axis
    (src/libImaging/Resample.c), as one int32 array [out_size, 2 + ksize] = (first tap, tap count, taps).
    Double precision, then 22-bit fixed point -- the kernel's only inputs besides the pixels."""
    key = (int(in_size), int(out_size), filt)
    if filt not in ("bicubic", "bilinear"):
        raise ValueError(f"unknown resampling filter {filt!r}")
    if key in _RESAMPLE_CACHE:
        return _RESAMPLE_CACHE[key]
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = (2.0 if filt == "bicubic" else 1.0) * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    tab = np.zeros((out_size, 2 + ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    a = -0.5
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        x = np.abs((np.arange(xmax) + xmin - center + 0.5) * ss)
        if filt == "bicubic":
            w = np.where(x < 1.0, ((a + 2.0) * x - (a + 3.0)) * x * x + 1, np.where(x < 2.0, (((x - 5) * x + 8) * x - 4) * a, 0.0))
        else:
            w = np.where(x < 1.0, 1.0 - x, 0.0)
        ww = 0.0
        for v in w:                                   # sequential sum, as the C loop
            ww += float(v)
        if ww != 0.0:
            w = w / ww
        tab[xx, 0], tab[xx, 1] = xmin, xmax
        tab[xx, 2:2 + xmax] = [int(-0.5 + float(v) * (1 << 22)) if v < 0 else int(0.5 + float(v) * (1 << 22)) for v in w]
    _RESAMPLE_CACHE[key] = (tab, ksize)
    return tab, ksize


def preprocess_images(images, S, mean=(0.0, 0.0, 0.0), std=(1.0, 1.0, 1.0), device=None, filt="bicubic"):
    """Dataset.py:434-443 on device: list of RGB uint8 arrays [H, W, 3] (any sizes) -> float32 tensor [B, 3, S, S]
    = Normalize(ToTensor(PIL resize)), bit-identical to the Pillow / torchvision host path.  ADE20K (Dataset.py:1263,
    1272-1275) is filt="bilinear" with the default mean 0 / std 1 ((v - 0) / 1 is exact)."""
    if not torch.cuda.is_available():
        raise RuntimeError("pnp_ovss.hip.preprocess_images needs a HIP device (no CPU fallback)")
    lib = load_library()
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    B = len(images)
    desc = (PnpPreImage * B)()
    coef_parts, coef_off, cache = [], 0, {}
    src_off = tmp_off = 0
    for i, im in enumerate(images):
        H, W = int(im.shape[0]), int(im.shape[1])
        for axis, n in (("x", W), ("y", H)):
            if (n, S) not in cache:
                tab, ks = resample_table(n, S, filt)
                cache[(n, S)] = (coef_off, ks)
                coef_parts.append(tab.reshape(-1))
                coef_off += tab.size
        kx, ky = cache[(W, S)], cache[(H, S)]
        desc[i] = PnpPreImage(src_off, tmp_off, H, W, kx[0], kx[1], ky[0], ky[1])
        src_off += H * W * 3
        tmp_off += H * S * 3
    if all(isinstance(im, torch.Tensor) for im in images):      # already on the device (pnp_jpeg_decode output)
        rgb = torch.cat([im.reshape(-1) for im in images]) if len(images) > 1 else images[0].reshape(-1)
        assert rgb.is_cuda and rgb.dtype == torch.uint8
    else:
        rgb = torch.from_numpy(np.concatenate([np.ascontiguousarray(im.cpu().numpy() if isinstance(im, torch.Tensor) else im,
                                                                    dtype=np.uint8).reshape(-1) for im in images])).to(dev)
    coef = torch.from_numpy(np.concatenate(coef_parts)).to(dev)
    d_desc = torch.frombuffer(bytearray(bytes(desc)), dtype=torch.uint8).to(dev)
    tmp = torch.empty(tmp_off, dtype=torch.uint8, device=dev)
    out = torch.empty(B, 3, S, S, dtype=torch.float32, device=dev)
    m3 = (C.c_float * 3)(*[float(np.float32(v)) for v in mean])
    s3 = (C.c_float * 3)(*[float(np.float32(v)) for v in std])
    r = lib.pnp_preprocess_images(rgb.data_ptr(), d_desc.data_ptr(), B, S, max(int(im.shape[0]) for im in images), coef.data_ptr(),
                                  tmp.data_ptr(), m3, s3, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    if r != 0:
        raise RuntimeError(f"pnp_preprocess_images failed ({r})")
    return out


def jpeg_decode_batch(files, device=None):
    """`Image.open(f).convert('RGB')` for a batch of baseline JPEG byte strings, on the device (pnp_jpeg_decode): returns a
    list of uint8 device tensors (H, W, 3), views of one concatenated buffer in batch order -- the layout the resize
    kernel and the CRF read.  Raises pnp_ovss.jpeg.UnsupportedJpeg for progressive / arithmetic / CMYK files (the caller
    decodes those with Pillow) and RuntimeError for a corrupt stream."""
    if not torch.cuda.is_available():
        raise RuntimeError("pnp_ovss.hip.jpeg_decode_batch needs a HIP device (no CPU fallback)")
    from . import jpeg as J
    lib = load_library()
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    data, imgs, tabs, segs, sizes, tot = J.pack_batch(files)

    def up(buf):
        return torch.frombuffer(bytearray(bytes(buf)), dtype=torch.uint8).to(dev)
    d_data = torch.from_numpy(data).to(dev)
    d_imgs, d_tabs, d_segs = up(imgs), up(tabs), up(segs)
    d_clean = torch.empty(tot["clean_bytes"], dtype=torch.uint8, device=dev)
    d_bits = torch.empty(len(segs), dtype=torch.int32, device=dev)
    d_coef = torch.empty(tot["coef_elems"], dtype=torch.int16, device=dev)
    d_planes = torch.empty(tot["plane_bytes"], dtype=torch.uint8, device=dev)
    d_rgb = torch.empty(tot["rgb_bytes"], dtype=torch.uint8, device=dev)
    d_err = torch.zeros(1, dtype=torch.int32, device=dev)
    r = lib.pnp_jpeg_decode(d_data.data_ptr(), d_imgs.data_ptr(), d_tabs.data_ptr(), d_segs.data_ptr(), len(files), len(segs),
                            d_clean.data_ptr(), d_bits.data_ptr(), d_coef.data_ptr(), tot["coef_elems"], d_planes.data_ptr(), d_rgb.data_ptr(), tot["max_blocks"],
                            tot["max_pixels"], d_err.data_ptr(), torch.cuda.current_stream().cuda_stream)
    if r != 0:
        raise RuntimeError(f"pnp_jpeg_decode failed ({r})")
    if int(d_err.item()):
        raise RuntimeError("pnp_jpeg_decode: corrupt entropy-coded data")
    out, o = [], 0
    for h, w in sizes:
        out.append(d_rgb[o:o + h * w * 3].view(h, w, 3))
        o += h * w * 3
    return out


def gaussian_taps(H, W, scale=0.05, truncate=4.0):
    """scipy.ndimage._filters._gaussian_kernel1d (order 0) for sigma = scale * max(H, W)
    (PnP_OVSS_0514_updated_segmentation.py:1150) -> taps at distance 0..radius (float64)."""
    sigma = scale * max(H, W)
    radius = int(truncate * float(sigma) + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    phi = phi / phi.sum()
    return np.ascontiguousarray(phi[radius:][: radius + 1])


class Engine:
    """One libpnp_hip engine on one GPU."""

    MODES = {"f32": 0, "bf16": 1, "bf16x3": 2}

    def __init__(self, cfg, max_batch, max_text_len=64, stash_layer=7, bf16=False, device=0, mode=None, share_weights_with=None):
        """mode: "f32" (the reference's arithmetic; the default), "bf16x3" (split-bf16: fp32-class results on the bf16
        MFMA), "bf16" (throughput; does not reproduce the reference's patch picks); `bf16=True/False` is the older
        spelling of "bf16" / "f32".  share_weights_with: a finalized Engine of the same geometry, device and mode whose
        weights this one uses (pnp_create_shared: activations and workspace of its own, no load_state_dict)."""
        if not torch.cuda.is_available():
            raise RuntimeError("pnp_ovss.hip.Engine needs a HIP device (no CPU fallback)")
        self.lib = load_library()
        self.cfg = cfg
        self.device = torch.device("cuda", device)
        if mode is None:
            mode = "bf16" if bf16 else "f32"
        if mode not in self.MODES:
            raise ValueError(f"unknown compute mode {mode!r}")
        self.mode = mode
        self.bf16 = mode == "bf16"
        self.max_batch, self.max_text_len, self.stash_layer = max_batch, max_text_len, stash_layer
        c = PnpConfig(cfg.img_size, cfg.patch, cfg.vit_dim, cfg.vit_depth, cfg.vit_heads, cfg.vit_mlp_ratio,
                      cfg.vit_ln_eps, cfg.txt_hidden, cfg.txt_layers, cfg.txt_heads, cfg.txt_inter, cfg.txt_ln_eps,
                      cfg.vocab, cfg.max_pos, cfg.enc_token_id, max_batch, max_text_len, stash_layer,
                      self.MODES[mode], device)
        self._c = c
        self.h = C.c_void_p()
        torch.cuda.set_device(device)
        if share_weights_with is not None:
            r = self.lib.pnp_create_shared(C.byref(c), share_weights_with.h, C.byref(self.h))
        else:
            r = self.lib.pnp_create(C.byref(c), C.byref(self.h))
        if r != 0:
            msg = self.lib.pnp_last_error(self.h).decode() if self.h else "?"
            if self.h:
                self.lib.pnp_destroy(self.h)
                self.h = C.c_void_p()
            raise RuntimeError(f"pnp_create{'_shared' if share_weights_with is not None else ''} failed ({r}): {msg}")
        self.shares_weights = share_weights_with is not None
        self._keep = []

    # ------------------------------------------------------------------ plumbing
    def _chk(self, r, what):
        if r != 0:
            raise RuntimeError(f"{what} failed ({r}): {self.lib.pnp_last_error(self.h).decode()}")

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            torch.cuda.synchronize()
            self.lib.pnp_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def grid(self):
        return self.cfg.grid

    def allocated_bytes(self):
        return int(self.lib.pnp_allocated_bytes(self.h))

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, sd, finalize=True):
        """sd: name -> numpy fp32 array or torch tensor (CPU, or CUDA e.g. after an RCCL broadcast)."""
        for name, w in sd.items():
            if isinstance(w, torch.Tensor):
                t = w.detach().to(torch.float32).contiguous()
                shape = (C.c_int64 * t.dim())(*t.shape)
                r = self.lib.pnp_load_weight(self.h, name.encode(), C.c_void_p(t.data_ptr()), shape, t.dim(),
                                             1 if t.is_cuda else 0)
            else:
                a = np.ascontiguousarray(w, dtype=np.float32)
                shape = (C.c_int64 * a.ndim)(*a.shape)
                r = self.lib.pnp_load_weight(self.h, name.encode(), C.c_void_p(a.ctypes.data), shape, a.ndim, 0)
            self._chk(r, f"pnp_load_weight({name})")
        if finalize:
            self._chk(self.lib.pnp_finalize_weights(self.h), "pnp_finalize_weights")

    # ------------------------------------------------------------------ model
    def vit_forward(self, images, dropped=None):
        B = images.shape[0]
        self._chk(self.lib.pnp_vit_forward(self.h, _ptr(images), _ptr(dropped), B, _stream()), "pnp_vit_forward")

    def cross_kv(self, B):
        self._chk(self.lib.pnp_cross_kv(self.h, B, _stream()), "pnp_cross_kv")

    def text_forward(self, ids, mask, L):
        B, ld = ids.shape
        logits = torch.empty(B, 2, device=self.device, dtype=torch.float32)
        self._chk(self.lib.pnp_text_forward_xattn(self.h, _ptr(ids), _ptr(mask), ld, B, L, _ptr(logits), _stream()),
                  "pnp_text_forward_xattn")
        return logits

    def xattn_grad(self, B, L, layer=None):
        """Backward down to `layer` (default: stash_layer); gradcam_gather / buffer("P") / buffer("dP") then refer to it."""
        layer = self.stash_layer if layer is None else int(layer)
        self._chk(self.lib.pnp_xattn_grad_layer(self.h, B, L, layer, _stream()), "pnp_xattn_grad_layer")

    def gradcam_gather(self, mask, L, head):
        B, ld = mask.shape
        out = torch.empty(B, L - 1, self.grid, self.grid, device=self.device, dtype=torch.float32)
        self._chk(self.lib.pnp_gradcam_gather(self.h, _ptr(mask), ld, B, L, head, _ptr(out), _stream()),
                  "pnp_gradcam_gather")
        return out

    def compute_gradcam(self, images, ids, mask, L, head, dropped=None, layer=None):
        B, ld = ids.shape
        out = torch.empty(B, L - 1, self.grid, self.grid, device=self.device, dtype=torch.float32)
        logits = torch.empty(B, 2, device=self.device, dtype=torch.float32)
        layer = self.stash_layer if layer is None else int(layer)
        self._chk(self.lib.pnp_compute_gradcam_layer(self.h, _ptr(images), _ptr(dropped), _ptr(ids), _ptr(mask), ld, B, L,
                                                     layer, head, _ptr(out), _ptr(logits), _stream()), "pnp_compute_gradcam_layer")
        return out, logits

    def drop_step(self, gradcam, g0, agg, dropped, picks, it, npick=10):
        """One bookkeeping step of the drop loop (PnP.py:619-647, 716-721) on a gathered map: updates g0 (it == 0), the
        running sum agg, the dropped-patch mask and the pick list, all on device."""
        B, T = gradcam.shape[0], gradcam.shape[1]
        self._chk(self.lib.pnp_drop_step(self.h, _ptr(gradcam), _ptr(g0), _ptr(agg), _ptr(dropped), _ptr(picks), it, B, T, npick,
                                         picks.shape[1], _stream()), "pnp_drop_step")

    def drop_loop(self, images, ids, mask, L, head, drop_iter, npick=10, layer=None):
        B, ld = ids.shape
        layer = self.stash_layer if layer is None else int(layer)
        g0 = torch.empty(B, L - 1, self.grid, self.grid, device=self.device, dtype=torch.float32)
        agg = torch.empty_like(g0) if drop_iter > 1 else None
        picks = torch.full((B, max(drop_iter, 1) * npick), -1, device=self.device, dtype=torch.int32)
        logits = torch.empty(B, 2, device=self.device, dtype=torch.float32)
        self._chk(self.lib.pnp_drop_loop_layer(self.h, _ptr(images), _ptr(ids), _ptr(mask), ld, B, L, layer, head, drop_iter,
                                               npick, _ptr(g0), _ptr(agg), _ptr(picks), _ptr(logits), _stream()), "pnp_drop_loop_layer")
        return g0, agg, picks, logits

    # ------------------------------------------------------------------ post-process
    def post_reserve(self, max_batch, max_total_pixels, max_pixels_per_image, max_channels, crf_chunk=0):
        self._chk(self.lib.pnp_post_reserve(self.h, max_batch, int(max_total_pixels), int(max_pixels_per_image),
                                            max_channels, crf_chunk), "pnp_post_reserve")

    def post_prepare(self, sizes, plans, luts, has_bg, rgb=None, gt=None, want_crf=True):
        """sizes: [(H, W)], plans: per image list of (token index list, divisor) per class,
        luts: per image list mapping argmax index -> class id, rgb/gt: concatenated device tensors."""
        B = len(sizes)
        H = np.array([s[0] for s in sizes], dtype=np.int32)
        W = np.array([s[1] for s in sizes], dtype=np.int32)
        ncls = np.array([len(p) for p in plans], dtype=np.int32)
        hb = np.array([1 if b else 0 for b in has_bg], dtype=np.int32)
        img_cls_off = np.zeros(B + 1, dtype=np.int32)
        img_cls_off[1:] = np.cumsum(ncls)
        cls_off, tok_idx, cls_div = [0], [], []
        for p in plans:
            for toks, div in p:
                tok_idx.extend(int(t) for t in toks)
                cls_off.append(len(tok_idx))
                cls_div.append(int(div))
        cls_off = np.array(cls_off, dtype=np.int32)
        tok_idx = np.array(tok_idx or [0], dtype=np.int32)
        cls_div = np.array(cls_div or [1], dtype=np.int32)
        stride = int(max(len(l) for l in luts))
        lut = np.zeros((B, stride), dtype=np.int32)
        for i, l in enumerate(luts):
            lut[i, : len(l)] = l
        taps = [gaussian_taps(h, w) for h, w in sizes]
        wt_off = np.zeros(B + 1, dtype=np.int32)
        wt_off[1:] = np.cumsum([len(t) for t in taps])
        wts = np.ascontiguousarray(np.concatenate(taps), dtype=np.float64)
        pb = PnpPostBatch(B, _i32p(H), _i32p(W), _i32p(ncls), _i32p(hb), _i32p(img_cls_off), _i32p(cls_off),
                          _i32p(tok_idx), _i32p(cls_div), _i32p(lut), stride,
                          C.c_void_p(rgb.data_ptr()) if rgb is not None else None,
                          C.c_void_p(gt.data_ptr()) if gt is not None else None,
                          wts.ctypes.data_as(C.POINTER(C.c_double)), _i32p(wt_off))
        self._post_keep = (rgb, gt)          # the engine keeps the device pointers for the batch
        self._post_sizes = list(sizes)
        self._post_K = [int(n + b) for n, b in zip(ncls, hb)]
        self._chk(self.lib.pnp_post_prepare(self.h, C.byref(pb), 1 if want_crf else 0, _stream()), "pnp_post_prepare")

    def merge_tokens(self, gradcam):
        T = gradcam.shape[1]
        self._chk(self.lib.pnp_merge_tokens(self.h, _ptr(gradcam), T, _stream()), "pnp_merge_tokens")

    def threshold_upsample(self, threshold, scale01):
        self._chk(self.lib.pnp_threshold_upsample(self.h, float(threshold), 1 if scale01 else 0, _stream()),
                  "pnp_threshold_upsample")

    def blur_minmax(self):
        self._chk(self.lib.pnp_blur_minmax(self.h, _stream()), "pnp_blur_minmax")

    def densecrf(self, iters=10, pos_w=7.0, pos_xy=3.0, bi_w=10.0, bi_xy=50.0, bi_rgb=5.0):
        self._chk(self.lib.pnp_densecrf(self.h, iters, pos_w, pos_xy, bi_w, bi_xy, bi_rgb, _stream()), "pnp_densecrf")

    def remap_hist(self, from_crf, n_class=0, hist=None):
        total = sum(h * w for h, w in self._post_sizes)
        labels = torch.empty(total, device=self.device, dtype=torch.uint8)
        self._chk(self.lib.pnp_remap_hist(self.h, 1 if from_crf else 0, _ptr(labels), _ptr(hist), n_class, _stream()),
                  "pnp_remap_hist")
        return labels

    def postprocess(self, gradcam, threshold, scale01, mode, n_class=0, hist=None):
        """mode: 'blur+crf' | 'crf' | 'blur' | None  (--postprocess, PnP.py:103)."""
        m = 0
        if mode:
            m = (1 if "blur" in mode else 0) | (2 if "crf" in mode else 0)
        total = sum(h * w for h, w in self._post_sizes)
        labels = torch.empty(total, device=self.device, dtype=torch.uint8)
        self._chk(self.lib.pnp_postprocess(self.h, _ptr(gradcam), gradcam.shape[1], float(threshold),
                                           1 if scale01 else 0, m, _ptr(labels), _ptr(hist), n_class, _stream()),
                  "pnp_postprocess")
        return labels

    def postprocess_pair(self, gradcam_1drop, gradcam_ndrop, threshold, n_class=0, hist_1drop=None, hist_ndrop=None,
                         scale01=(True, False)):
        """Both "blur+crf" branches of a batch in one DenseCRF run; returns (labels_1drop, labels_ndrop), identical
        to two postprocess() calls.  scale01 = Scale_0_1 on (1-drop, N-drop): (True, False) is PnP.py, the COCO
        driver scales both."""
        total = sum(h * w for h, w in self._post_sizes)
        l1 = torch.empty(total, device=self.device, dtype=torch.uint8)
        ln = torch.empty(total, device=self.device, dtype=torch.uint8)
        assert gradcam_1drop.shape == gradcam_ndrop.shape
        self._chk(self.lib.pnp_postprocess_pair(self.h, _ptr(gradcam_1drop), _ptr(gradcam_ndrop), gradcam_1drop.shape[1],
                                                float(threshold), (1 if scale01[0] else 0) | (2 if scale01[1] else 0),
                                                _ptr(l1), _ptr(hist_1drop), _ptr(ln), _ptr(hist_ndrop),
                                                n_class, _stream()), "pnp_postprocess_pair")
        return l1, ln

    def split_labels(self, labels):
        out, o = [], 0
        for h, w in self._post_sizes:
            out.append(labels[o:o + h * w].view(h, w))
            o += h * w
        return out

    # ------------------------------------------------------------------ live kernel timing (bench roofline)
    def profile_enable(self, on=True):
        """on: False / True (every launch of the dense GEMM family bracketed by hipEvents) / n > 1 (every n-th launch)."""
        self._chk(self.lib.pnp_profile_enable(self.h, int(on)), "pnp_profile_enable")

    def profile_read(self):
        n, fl, ms = C.c_int64(), C.c_double(), C.c_double()
        self._chk(self.lib.pnp_profile_read(self.h, C.byref(n), C.byref(fl), C.byref(ms)), "pnp_profile_read")
        return n.value, fl.value, ms.value

    def profile_read_stage(self, stage):
        """stage 0: dense GEMMs (launches, FLOPs, ms); stage 1: DenseCRF mean-field (runs, SURVEY 8d bytes, ms);
        stage 2: the same brackets with the lattice-blur bytes that 8d leaves out."""
        n, w, ms = C.c_int64(), C.c_double(), C.c_double()
        self._chk(self.lib.pnp_profile_read_stage(self.h, stage, C.byref(n), C.byref(w), C.byref(ms)), "pnp_profile_read_stage")
        return n.value, w.value, ms.value

    def streamk_status(self):
        """(launches of this engine that split their last tile round along K, give-up word of the bounded spins: 0 = none)."""
        n, t = C.c_int64(), C.c_uint32()
        self._chk(self.lib.pnp_streamk_status(self.h, C.byref(n), C.byref(t)), "pnp_streamk_status")
        return n.value, t.value

    # ------------------------------------------------------------------ introspection
    def buffer(self, name, dtype=torch.float32):
        p, n = C.c_void_p(), C.c_size_t()
        self._chk(self.lib.pnp_get_buffer(self.h, name.encode(), C.byref(p), C.byref(n)), f"pnp_get_buffer({name})")
        item = torch.empty(0, dtype=dtype).element_size()
        typestr = {torch.float32: "<f4", torch.int32: "<i4", torch.uint8: "|u1"}[dtype]
        return torch.as_tensor(_DevView(p.value, (n.value // item,), typestr), device=self.device)

    def buffer_ptr(self, name):
        """(device pointer, bytes) of a named internal buffer."""
        p, n = C.c_void_p(), C.c_size_t()
        self._chk(self.lib.pnp_get_buffer(self.h, name.encode(), C.byref(p), C.byref(n)), f"pnp_get_buffer({name})")
        return p.value, n.value

    def post_maps(self, name="maps"):
        """Per-image (K,H,W) views of an internal post-process map buffer."""
        flat = self.buffer(name)
        out, o = [], 0
        for (h, w), k in zip(self._post_sizes, self._post_K):
            out.append(flat[o:o + k * h * w].view(k, h, w))
            o += k * h * w
        return out

    def post_q(self):
        """Per-image CRF marginals, pixel-major (H*W, K) (rows are padded to a multiple of 4 on device)."""
        flat = self.buffer("crf_q")
        out, o = [], 0
        for (h, w), k in zip(self._post_sizes, self._post_K):
            kp = (k + 3) // 4 * 4
            out.append(flat[o:o + kp * h * w].view(h * w, kp)[:, :k])
            o += kp * h * w
        return out


def set_tuning(key, value):
    """Process-wide tuning switch (include/pnp_hip.h: pnp_set_tuning), e.g. set_tuning("streamk", 0 | 1 | 2)."""
    r = load_library().pnp_set_tuning(key.encode(), int(value))
    if r != 0:
        raise ValueError(f"pnp_set_tuning({key!r}, {value}) -> {r}")


def streamk_status_ops():
    """pnp_streamk_status of the op-level entry points' workspace."""
    n, t = C.c_int64(), C.c_uint32()
    r = load_library().pnp_streamk_status(None, C.byref(n), C.byref(t))
    if r != 0:
        raise RuntimeError(f"pnp_streamk_status -> {r}")
    return n.value, t.value
