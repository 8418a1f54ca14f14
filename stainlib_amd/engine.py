"""Batched device-level operators: torch uint8 NHWC tensors in HBM -> the C ABI.

PyTorch is plumbing here (device memory, streams); every computation is a call into
``libstainlib_hip.so``.  All functions enqueue on torch's current stream and return
device tensors without synchronising.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _ffi


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _check_tiles(rgb: torch.Tensor):
    if not (isinstance(rgb, torch.Tensor) and rgb.is_cuda and rgb.dtype == torch.uint8 and rgb.dim() == 4
            and rgb.shape[-1] == 3 and rgb.is_contiguous()):
        raise ValueError("expected a contiguous CUDA uint8 tensor of shape (N, H, W, 3)")
    n, h, w, _ = rgb.shape
    return n, h, w


def _f64(x, shape, device):
    if not isinstance(x, torch.Tensor):
        import numpy as np
        x = np.asarray(x, dtype=np.float64)
    t = torch.as_tensor(x, dtype=torch.float64, device=device).reshape(shape)
    return t.contiguous()


def make_params(**kw) -> _ffi.SlParams:
    p = _ffi.default_params()
    for k, v in kw.items():
        if v is not None:
            setattr(p, k, v)
    return p


def attach_fallbacks(params: _ffi.SlParams, n: int, device="cuda") -> torch.Tensor:
    """Give ``params`` a device buffer of n int32 that sl_macenko_* / sl_vahadane_* fill with the per-tile count of order
    statistics that needed the slow exact selection (SlParams.fallbacks_out; diagnostics only).  Returns the tensor --
    keep it alive as long as ``params`` is used."""
    t = torch.zeros((n,), dtype=torch.int32, device=device)
    params.fallbacks_out = t.data_ptr()
    return t


class Workspace:
    """Caller-owned scratch the library asks for via sl_workspace_bytes (grown on demand).

    One Workspace must only ever be in use on ONE stream at a time: the kernels of a call keep their per-tile state in it.
    Pass your own (``ws=``) to pin a buffer to a pipeline stage; without one every call takes a fresh block from torch's
    caching allocator, which is stream-ordered -- two streams or threads can then never share scratch."""

    def __init__(self):
        self.buf = None

    def get(self, op: int, n: int, h: int, w: int, device, params=None) -> torch.Tensor:
        need = _ws_need(op, n, h, w, params)
        if self.buf is None or self.buf.numel() < need or self.buf.device != device:
            self.buf = torch.empty(max(need, 256), dtype=torch.uint8, device=device)
        return self.buf


def _ws_need(op, n, h, w, params=None) -> int:
    """what THIS call needs (sl_workspace_bytes_for: the schedule its SlParams select), not the maximum over every SlParams"""
    import ctypes as C
    return int(_ffi.lib().sl_workspace_bytes_for(op, n, h, w, C.byref(params) if params is not None else None))


def _scratch(ws, op, n, h, w, device, params=None) -> torch.Tensor:
    """The workspace of one call: the caller's Workspace, or a block of torch's caching allocator owned by the current
    stream for the duration of the call's kernels (freed blocks are reused on the same stream only after them)."""
    if ws is not None:
        return ws.get(op, n, h, w, device, params)
    need = _ws_need(op, n, h, w, params)
    return torch.empty(max(need, 256), dtype=torch.uint8, device=device)


class Graphed:
    """A sequence of engine calls captured ONCE into a HIP graph and replayed.  Every fit / transform / apply / augment
    entry point of the C ABI is capture-safe -- kernel launches on the caller's stream, no allocation, no synchronisation,
    no host read-back (the pooled slide mode, which reads histograms back between stages, is not).  What it buys is the
    LATENCY of an isolated small call on the one-launch-per-phase schedule (7 launches for Macenko, 11 for Vahadane), whose
    first kernels otherwise wait for the host to issue the next launch: 1024^2 Macenko transform, call-to-completion, 16
    tiles 274 -> 239 us, 128 tiles 671 -> 623 us.  Calls queued back to back gain nothing (the host already runs ahead
    of the device: 16 tiles 0.237 vs 0.235 ms per call).

        g = engine.Graphed(lambda: engine.macenko_transform(tiles, M_t, maxC_t, out=out, ws=ws))
        ...   # refill `tiles` in place (same tensors), then
        out, M, maxC, status = g.replay()

    fn must use only tensors that stay alive and in place (pass out= and ws=); it runs once for warm-up and once under
    capture.  replay() enqueues the graph on the current stream and returns what fn returned (the same tensors)."""

    def __init__(self, fn):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()                                     # warm-up outside the capture: workspace growth, lazy initialisation
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=side):
            self.result = fn()

    def replay(self):
        self.graph.replay()
        return self.result


def normalize_apply(rgb, M_src, maxC_src, M_tgt, maxC_tgt, lasso_lambda=0.01, out=None, want_prequant=False):
    """OD + reconstruction pass (sl_normalize_apply).  Returns out, or (out, prequant)."""
    n, h, w = _check_tiles(rgb)
    dev = rgb.device
    M_src = _f64(M_src, (n, 2, 3), dev)
    maxC_src = _f64(maxC_src, (n, 2), dev)
    M_tgt = _f64(M_tgt, (2, 3), dev)
    maxC_tgt = _f64(maxC_tgt, (2,), dev)
    if out is None:
        out = torch.empty_like(rgb)
    pre = torch.empty((n, h, w, 3), dtype=torch.float32, device=dev) if want_prequant else None
    _ffi.check(_ffi.lib().sl_normalize_apply(_ptr(rgb), _ptr(out), n, h, w, _ptr(M_src), _ptr(maxC_src),
                                             _ptr(M_tgt), _ptr(maxC_tgt), float(lasso_lambda), _ptr(pre),
                                             _stream()), "sl_normalize_apply")
    return (out, pre) if want_prequant else out


def _fit(fn_name, op, rgb, params, ws, with_sweeps=False):
    n, h, w = _check_tiles(rgb)
    dev = rgb.device
    p = params if params is not None else _ffi.default_params()
    M = torch.empty((n, 2, 3), dtype=torch.float64, device=dev)
    maxC = torch.empty((n, 2), dtype=torch.float64, device=dev)
    status = torch.empty((n,), dtype=torch.int32, device=dev)
    wsb = _scratch(ws, op, n, h, w, dev, p)
    fn = getattr(_ffi.lib(), fn_name)
    if with_sweeps:
        sweeps = torch.empty((n,), dtype=torch.int32, device=dev)
        code = fn(_ptr(rgb), n, h, w, C.byref(p), _ptr(M), _ptr(maxC), _ptr(status), _ptr(sweeps), _ptr(wsb),
                  wsb.numel(), _stream())
        _ffi.check(code, fn_name)
        return M, maxC, status, sweeps
    code = fn(_ptr(rgb), n, h, w, C.byref(p), _ptr(M), _ptr(maxC), _ptr(status), _ptr(wsb), wsb.numel(), _stream())
    _ffi.check(code, fn_name)
    return M, maxC, status


def macenko_fit(rgb, params=None, ws=None):
    """Per-tile Macenko stain matrix (N,2,3) f64, 99th-percentile concentrations (N,2) f64, status (N,) i32."""
    return _fit("sl_macenko_fit", _ffi.OP_MACENKO_FIT, rgb, params, ws)


def vahadane_fit(rgb, params=None, ws=None):
    """As macenko_fit with the sparse-NMF dictionary; also returns sweeps used per tile."""
    return _fit("sl_vahadane_fit", _ffi.OP_VAHADANE_FIT, rgb, params, ws, with_sweeps=True)


def _transform(fn_name, op, rgb, M_tgt, maxC_tgt, params, out, ws):
    n, h, w = _check_tiles(rgb)
    dev = rgb.device
    p = params if params is not None else _ffi.default_params()
    M_tgt = _f64(M_tgt, (2, 3), dev)
    maxC_tgt = _f64(maxC_tgt, (2,), dev)
    if out is None:
        out = torch.empty_like(rgb)
    M = torch.empty((n, 2, 3), dtype=torch.float64, device=dev)
    maxC = torch.empty((n, 2), dtype=torch.float64, device=dev)
    status = torch.empty((n,), dtype=torch.int32, device=dev)
    wsb = _scratch(ws, op, n, h, w, dev, p)
    code = getattr(_ffi.lib(), fn_name)(_ptr(rgb), _ptr(out), n, h, w, C.byref(p), _ptr(M_tgt), _ptr(maxC_tgt),
                                        _ptr(M), _ptr(maxC), _ptr(status), _ptr(wsb), wsb.numel(), _stream())
    _ffi.check(code, fn_name)
    return out, M, maxC, status


def macenko_transform(rgb, M_tgt, maxC_tgt, params=None, out=None, ws=None):
    """Batched ExtractiveStainNormalizer('macenko').transform -> (out, M_src, maxC_src, status)."""
    return _transform("sl_macenko_transform", _ffi.OP_MACENKO_TRANSFORM, rgb, M_tgt, maxC_tgt, params, out, ws)


def vahadane_transform(rgb, M_tgt, maxC_tgt, params=None, out=None, ws=None):
    return _transform("sl_vahadane_transform", _ffi.OP_VAHADANE_TRANSFORM, rgb, M_tgt, maxC_tgt, params, out, ws)


def hed_augment(rgb, sigma, bias, cutoff=(0.05, 0.95), skimage_mode=0, out=None, ws=None, want_sums=False):
    """Batched HedColorAugmenter.transform for uint8 tiles -> (out, applied (N,) i32)[, exact byte sums (N,) int64]."""
    n, h, w = _check_tiles(rgb)
    dev = rgb.device
    sigma = _f64(sigma, (n, 3), dev)
    bias = _f64(bias, (n, 3), dev)
    if out is None:
        out = torch.empty_like(rgb)
    applied = torch.empty((n,), dtype=torch.int32, device=dev)
    wsb = _scratch(ws, _ffi.OP_HED_AUGMENT, n, h, w, dev)
    _ffi.check(_ffi.lib().sl_hed_augment(_ptr(rgb), _ptr(out), n, h, w, _ptr(sigma), _ptr(bias), float(cutoff[0]),
                                         float(cutoff[1]), int(skimage_mode), _ptr(applied), _ptr(wsb), wsb.numel(),
                                         _stream()), "sl_hed_augment")
    if want_sums:
        return out, applied, wsb[:8 * n].view(torch.int64).clone()
    return out, applied


def hed_augment_float(patches, sigma, bias, cutoff=(0.05, 0.95), skimage_mode=0):
    """Float branch: (N,H,W,3) float64 CUDA tensor in [0,1] -> (out float64, applied)."""
    if not (patches.is_cuda and patches.dtype == torch.float64 and patches.dim() == 4 and patches.is_contiguous()):
        raise ValueError("expected a contiguous CUDA float64 tensor of shape (N, H, W, 3)")
    n, h, w, _ = patches.shape
    dev = patches.device
    sigma = _f64(sigma, (n, 3), dev)
    bias = _f64(bias, (n, 3), dev)
    out = torch.empty_like(patches)
    applied = torch.empty((n,), dtype=torch.int32, device=dev)
    wsb = torch.empty(max(8 * n, 256), dtype=torch.uint8, device=dev)
    _ffi.check(_ffi.lib().sl_hed_augment_f64(_ptr(patches), _ptr(out), n, h, w, _ptr(sigma), _ptr(bias), float(cutoff[0]),
                                             float(cutoff[1]), int(skimage_mode), _ptr(applied), _ptr(wsb), wsb.numel(),
                                             _stream()), "sl_hed_augment_f64")
    return out, applied


def rgb_to_od(rgb):
    """convert_RGB_to_OD materialised: (N,H,W,3) float64."""
    n, h, w = _check_tiles(rgb)
    od = torch.empty((n, h, w, 3), dtype=torch.float64, device=rgb.device)
    _ffi.check(_ffi.lib().sl_rgb_to_od(_ptr(rgb), n, h, w, _ptr(od), _stream()), "sl_rgb_to_od")
    return od


def stain_augment(rgb, M, alpha_beta, augment_background=False, params=None, out=None):
    """Batched StainAugmentor.pop given per-tile M (N,2,3) and (alpha0,beta0,alpha1,beta1) (N,4)."""
    n, h, w = _check_tiles(rgb)
    dev = rgb.device
    M = _f64(M, (n, 2, 3), dev)
    ab = _f64(alpha_beta, (n, 4), dev)
    p = params if params is not None else _ffi.default_params()
    if out is None:
        out = torch.empty_like(rgb)
    _ffi.check(_ffi.lib().sl_stain_augment(_ptr(rgb), _ptr(out), n, h, w, _ptr(M), _ptr(ab),
                                           1 if augment_background else 0, C.byref(p), _stream()), "sl_stain_augment")
    return out


def grayscale_augment(rgb, alpha_beta, out=None):
    """GrayscaleAugmentor.pop on a batch: alpha_beta (n, 2) -> (n, H, W, 3) uint8 with three equal channels."""
    n, h, w = _check_tiles(rgb)
    ab = _f64(alpha_beta, (n, 2), rgb.device)
    if out is None:
        out = torch.empty_like(rgb)
    _ffi.check(_ffi.lib().sl_grayscale_augment(_ptr(rgb), _ptr(out), n, h, w, _ptr(ab), _stream()), "sl_grayscale_augment")
    return out


def tissue_mask(rgb, luminosity_threshold=0.8, want_mask=True):
    """(mask (N,H,W) uint8 or None, counts (N,) int64)."""
    n, h, w = _check_tiles(rgb)
    dev = rgb.device
    mask = torch.empty((n, h, w), dtype=torch.uint8, device=dev) if want_mask else None
    counts = torch.empty((n,), dtype=torch.int64, device=dev)
    _ffi.check(_ffi.lib().sl_tissue_mask(_ptr(rgb), n, h, w, float(luminosity_threshold), _ptr(mask), _ptr(counts),
                                         _stream()), "sl_tissue_mask")
    return mask, counts


def concentrations(rgb, M, lasso_lambda=0.01):
    """get_concentrations materialised: (N, H*W, 2) float32."""
    n, h, w = _check_tiles(rgb)
    dev = rgb.device
    M = _f64(M, (n, 2, 3), dev)
    Cout = torch.empty((n, h * w, 2), dtype=torch.float32, device=dev)
    _ffi.check(_ffi.lib().sl_concentrations(_ptr(rgb), n, h, w, _ptr(M), float(lasso_lambda), _ptr(Cout), _stream()),
               "sl_concentrations")
    return Cout


def od_to_rgb(od):
    """convert_OD_to_RGB on a float64 CUDA tensor of any shape -> (uint8 tensor of the same shape, negative flag (1,) i32)."""
    if not (od.is_cuda and od.dtype == torch.float64 and od.is_contiguous()):
        raise ValueError("expected a contiguous CUDA float64 tensor")
    out = torch.empty(od.shape, dtype=torch.uint8, device=od.device)
    neg = torch.zeros((1,), dtype=torch.int32, device=od.device)
    _ffi.check(_ffi.lib().sl_od_to_rgb(_ptr(od), od.numel(), _ptr(out), _ptr(neg), _stream()), "sl_od_to_rgb")
    return out, neg


# ---- OpenCV 8-bit Lab family (lab.hip) ---------------------------------------------------------------
def rgb_to_lab8(rgb):
    """cv2.cvtColor(COLOR_RGB2LAB) on uint8 tiles -> (N,H,W,3) uint8."""
    n, h, w = _check_tiles(rgb)
    out = torch.empty_like(rgb)
    _ffi.check(_ffi.lib().sl_rgb_to_lab8(_ptr(rgb), _ptr(out), n, h, w, _stream()), "sl_rgb_to_lab8")
    return out


def lab8_to_rgb(lab):
    """cv2.cvtColor(COLOR_LAB2RGB) on uint8 tiles."""
    n, h, w = _check_tiles(lab)
    out = torch.empty_like(lab)
    _ffi.check(_ffi.lib().sl_lab8_to_rgb(_ptr(lab), _ptr(out), n, h, w, _stream()), "sl_lab8_to_rgb")
    return out


def lab_split(rgb):
    """lab_split: three (N,H,W) float32 planes L8/2.55, a8-128, b8-128."""
    n, h, w = _check_tiles(rgb)
    I1, I2, I3 = (torch.empty((n, h, w), dtype=torch.float32, device=rgb.device) for _ in range(3))
    _ffi.check(_ffi.lib().sl_lab_split(_ptr(rgb), n, h, w, _ptr(I1), _ptr(I2), _ptr(I3), _stream()), "sl_lab_split")
    return I1, I2, I3


def lab_merge(I1, I2, I3):
    """merge_back: three (N,H,W) planes of one float dtype (float32 or float64) -> (N,H,W,3) uint8 RGB."""
    if not (I1.dtype == I2.dtype == I3.dtype and I1.dtype in (torch.float32, torch.float64) and I1.shape == I2.shape == I3.shape
            and I1.dim() == 3 and all(t.is_cuda and t.is_contiguous() for t in (I1, I2, I3))):
        raise ValueError("expected three contiguous CUDA (N, H, W) planes of one float dtype")
    n, h, w = I1.shape
    out = torch.empty((n, h, w, 3), dtype=torch.uint8, device=I1.device)
    _ffi.check(_ffi.lib().sl_lab_merge(_ptr(I1), _ptr(I2), _ptr(I3), 1 if I1.dtype == torch.float64 else 0, n, h, w, _ptr(out),
                                       _stream()), "sl_lab_merge")
    return out


def standardize_brightness(rgb, out=None, ws=None):
    """standardize_brightness per tile -> (out, p90 (N,) f64)."""
    n, h, w = _check_tiles(rgb)
    if out is None:
        out = torch.empty_like(rgb)
    p = torch.empty((n,), dtype=torch.float64, device=rgb.device)
    wsb = _scratch(ws, _ffi.OP_LAB_STATS, n, h, w, rgb.device)
    _ffi.check(_ffi.lib().sl_standardize_brightness(_ptr(rgb), _ptr(out), n, h, w, _ptr(p), _ptr(wsb), wsb.numel(), _stream()),
               "sl_standardize_brightness")
    return out, p


def reinhard_stats(rgb, standardize=True, ws=None):
    """(N, 8) float64 per tile: p90, mean L/a/b, std L/a/b (cv2.meanStdDev of the lab_split planes), tissue count."""
    n, h, w = _check_tiles(rgb)
    st = torch.empty((n, 8), dtype=torch.float64, device=rgb.device)
    wsb = _scratch(ws, _ffi.OP_LAB_STATS, n, h, w, rgb.device)
    _ffi.check(_ffi.lib().sl_reinhard_stats(_ptr(rgb), n, h, w, 1 if standardize else 0, _ptr(st), _ptr(wsb), wsb.numel(), _stream()),
               "sl_reinhard_stats")
    return st


def reinhard_transform(rgb, target_means, target_stds, mask_background=False, luminosity_threshold=0.8, out=None, ws=None):
    """Batched ReinhardStainNormalizer.transform -> (out, stats (N, 8))."""
    n, h, w = _check_tiles(rgb)
    dev = rgb.device
    tm, ts = _f64(target_means, (3,), dev), _f64(target_stds, (3,), dev)
    if out is None:
        out = torch.empty_like(rgb)
    st = torch.empty((n, 8), dtype=torch.float64, device=dev)
    wsb = _scratch(ws, _ffi.OP_LAB_STATS, n, h, w, dev)
    _ffi.check(_ffi.lib().sl_reinhard_transform(_ptr(rgb), _ptr(out), n, h, w, _ptr(tm), _ptr(ts), 1 if mask_background else 0,
                                                float(luminosity_threshold), _ptr(st), _ptr(wsb), wsb.numel(), _stream()),
               "sl_reinhard_transform")
    return out, st


def luminosity_standardize(rgb, percentile=95, out=None, ws=None):
    """Batched LuminosityStandardizer.standardize -> (out, p (N,) f64)."""
    n, h, w = _check_tiles(rgb)
    if out is None:
        out = torch.empty_like(rgb)
    p = torch.empty((n,), dtype=torch.float64, device=rgb.device)
    wsb = _scratch(ws, _ffi.OP_LAB_STATS, n, h, w, rgb.device)
    _ffi.check(_ffi.lib().sl_luminosity_standardize(_ptr(rgb), _ptr(out), n, h, w, float(percentile), _ptr(p), _ptr(wsb),
                                                    wsb.numel(), _stream()), "sl_luminosity_standardize")
    return out, p


# ---- pooled slide-level mode: per-process reductions (combined over ranks in stainlib_amd.distributed) -------------
def tile_moments(rgb, params=None, ws=None):
    """(n, 10) float64 per tile: tissue count, sum od[3], sum od od^T [xx, xy, xz, yy, yz, zz]  (sl_tile_moments)."""
    n, h, w = _check_tiles(rgb)
    p = params if params is not None else _ffi.default_params()
    out = torch.empty((n, 10), dtype=torch.float64, device=rgb.device)
    wsb = _scratch(ws, _ffi.OP_TILE_MOMENTS, n, h, w, rgb.device)
    _ffi.check(_ffi.lib().sl_tile_moments(_ptr(rgb), n, h, w, C.byref(p), _ptr(out), _ptr(wsb), wsb.numel(), _stream()),
               "sl_tile_moments")
    return out


def _basis6(basis):
    import numpy as np
    b = np.ascontiguousarray(np.asarray(basis, dtype=np.float64).reshape(6))
    return b, b.ctypes.data_as(C.POINTER(C.c_double))


def slide_key_histogram(rgb, keyset, basis, prefixes, prefix_bits, hist=None, params=None):
    """Accumulate into hist ((2, 256) int64, device), for both targets of the key set, the next 8 key bits of this
    process's pixels whose key starts with prefixes[t]."""
    n, h, w = _check_tiles(rgb)
    p = params if params is not None else _ffi.default_params()
    if hist is None:
        hist = torch.zeros((2, 256), dtype=torch.int64, device=rgb.device)
    keep, bp = _basis6(basis)
    pre = (C.c_uint32 * 2)(int(prefixes[0]) & 0xffffffff, int(prefixes[1]) & 0xffffffff)
    _ffi.check(_ffi.lib().sl_slide_key_histogram(_ptr(rgb), n, h, w, C.byref(p), int(keyset), bp, pre, int(prefix_bits),
                                                 _ptr(hist), _stream()), "sl_slide_key_histogram")
    return hist


def slide_key_histogram16(rgb, keyset, basis, prefixes16, hist=None, params=None):
    """Accumulate into hist ((2, 65536) int64, device), for both targets, the LOW 16 key bits of this process's pixels
    whose key's top 16 bits equal prefixes16[t] (the last two radix rounds in one sweep)."""
    n, h, w = _check_tiles(rgb)
    p = params if params is not None else _ffi.default_params()
    if hist is None:
        hist = torch.zeros((2, 65536), dtype=torch.int64, device=rgb.device)
    keep, bp = _basis6(basis)
    pre = (C.c_uint32 * 2)(int(prefixes16[0]) & 0xffff, int(prefixes16[1]) & 0xffff)
    _ffi.check(_ffi.lib().sl_slide_key_histogram16(_ptr(rgb), n, h, w, C.byref(p), int(keyset), bp, pre, _ptr(hist), _stream()),
               "sl_slide_key_histogram16")
    return hist


def slide_key_histogram_sampled(rgb, keyset, basis, prefixes, prefix_bits, sample_log2, params=None):
    """slide_key_histogram over a stratified pixel sample (one 64-chunk row in 2**sample_log2); returns a fresh (2, 256) int64."""
    n, h, w = _check_tiles(rgb)
    p = params if params is not None else _ffi.default_params()
    hist = torch.zeros((2, 256), dtype=torch.int64, device=rgb.device)
    keep, bp = _basis6(basis)
    pre = (C.c_uint32 * 2)(int(prefixes[0]) & 0xffffffff, int(prefixes[1]) & 0xffffffff)
    _ffi.check(_ffi.lib().sl_slide_key_histogram_sampled(_ptr(rgb), n, h, w, C.byref(p), int(keyset), bp, pre, int(prefix_bits),
                                                         int(sample_log2), _ptr(hist), _stream()), "sl_slide_key_histogram_sampled")
    return hist


def slide_key_window(rgb, keyset, basis, window_lo, params=None):
    """Per target: histogram of key - window_lo[t] over this process's keys inside [window_lo[t], window_lo[t] + 65536) and the
    number of its keys below the window.  Returns one (2 * 65536 + 2,) int64 device tensor: hist[0], hist[1], below[0], below[1]."""
    n, h, w = _check_tiles(rgb)
    p = params if params is not None else _ffi.default_params()
    buf = torch.zeros((2 * 65536 + 2,), dtype=torch.int64, device=rgb.device)
    keep, bp = _basis6(basis)
    lo = (C.c_uint32 * 2)(int(window_lo[0]) & 0xffffffff, int(window_lo[1]) & 0xffffffff)
    _ffi.check(_ffi.lib().sl_slide_key_window(_ptr(rgb), n, h, w, C.byref(p), int(keyset), bp, lo, _ptr(buf), _stream()),
               "sl_slide_key_window")
    return buf


# ---- device-driven pooled statistics (sl_pool_*): every step is enqueued, nothing is read back -------------------------------
def pool_begin(moments11, state=None, params=None):
    """moments11: device float64 (11,) = the tile moments summed over all tiles and ranks + the pixel count.  Returns the state tensor."""
    p = params if params is not None else _ffi.default_params()
    if state is None:
        state = torch.empty((_ffi.POOL_STATE_DOUBLES,), dtype=torch.float64, device=moments11.device)
    _ffi.check(_ffi.lib().sl_pool_begin(_ptr(moments11), C.byref(p), _ptr(state), _stream()), "sl_pool_begin")
    return state


def pool_histogram(rgb, keyset, state, rnd, sample_log2, hist, params=None):
    """This process's sampled (2, 256) histogram of radix round `rnd` under the prefixes in `state`, accumulated into hist (zeroed by the caller)."""
    n, h, w = _check_tiles(rgb)
    p = params if params is not None else _ffi.default_params()
    _ffi.check(_ffi.lib().sl_pool_histogram(_ptr(rgb), n, h, w, C.byref(p), int(keyset), _ptr(state), int(rnd), int(sample_log2), _ptr(hist),
                                            _stream()), "sl_pool_histogram")
    return hist


def pool_pick(state, keyset, rnd, hist_reduced):
    _ffi.check(_ffi.lib().sl_pool_pick(_ptr(state), int(keyset), int(rnd), _ptr(hist_reduced), _stream()), "sl_pool_pick")


def pool_window(rgb, keyset, state, buf, params=None):
    """This process's window histogram + counts below ((2 * 65536 + 2,) int64, zeroed by the caller) around the windows in `state`."""
    n, h, w = _check_tiles(rgb)
    p = params if params is not None else _ffi.default_params()
    _ffi.check(_ffi.lib().sl_pool_window(_ptr(rgb), n, h, w, C.byref(p), int(keyset), _ptr(state), _ptr(buf), _stream()), "sl_pool_window")
    return buf


def pool_resolve(state, keyset, window_reduced, params=None):
    p = params if params is not None else _ffi.default_params()
    _ffi.check(_ffi.lib().sl_pool_resolve(_ptr(state), int(keyset), _ptr(window_reduced), C.byref(p), _stream()), "sl_pool_resolve")


# ---- the pooled statistics in ONE full sweep (sl_pool2_*): see include/stainlib_hip.h ------------------------------------------------
def pool2_workspace(n, h, w, sample_log2, device) -> torch.Tensor:
    need = int(_ffi.lib().sl_pool2_workspace_bytes(int(n), int(h), int(w), int(sample_log2)))
    if need == 0:
        raise ValueError("sl_pool2_workspace_bytes: bad arguments")
    return torch.empty(need, dtype=torch.uint8, device=device)


def pool2_sample(rgb, sample_log2, ws, params=None):
    """S1: this process's sample (packed list in ws) and its moment sums -> (16,) float64 to be all-reduced."""
    n, h, w = _check_tiles(rgb)
    p = params if params is not None else _ffi.default_params()
    out = torch.empty((16,), dtype=torch.float64, device=rgb.device)
    _ffi.check(_ffi.lib().sl_pool2_sample(_ptr(rgb), n, h, w, C.byref(p), int(sample_log2), _ptr(ws), ws.numel(), _ptr(out), _stream()),
               "sl_pool2_sample")
    return out


def pool2_begin(moments16, sample_log2, state=None, params=None):
    p = params if params is not None else _ffi.default_params()
    if state is None:
        state = torch.empty((_ffi.POOL2_STATE_DOUBLES,), dtype=torch.float64, device=moments16.device)
    _ffi.check(_ffi.lib().sl_pool2_begin(_ptr(moments16), C.byref(p), int(sample_log2), _ptr(state), _stream()), "sl_pool2_begin")
    return state


def pool2_hist(which, keyset, mode, shape, sample_log2, state, ws, hist, params=None):
    """A pass over the sample list (which=0, mode=0: a uniform grid) or the candidate list (which=1, mode=1: a window) of ws: the
    histogram of the key set under the constants in `state`, written into hist ((POOL2_HIST_WORDS,) int64)."""
    n, h, w = shape
    p = params if params is not None else _ffi.default_params()
    _ffi.check(_ffi.lib().sl_pool2_hist(int(which), int(keyset), int(mode), int(n), int(h), int(w), C.byref(p), int(sample_log2), _ptr(state),
                                        _ptr(ws), ws.numel(), _ptr(hist), _stream()), "sl_pool2_hist")
    return hist


def pool2_bands(state, keyset, hist_reduced):
    _ffi.check(_ffi.lib().sl_pool2_bands(_ptr(state), int(keyset), _ptr(hist_reduced), _stream()), "sl_pool2_bands")


def pool2_sweep(rgb, sample_log2, state, ws, params=None):
    """THE full sweep: exact moment sums + the raw candidates (into ws) -> (16,) float64 to be all-reduced."""
    n, h, w = _check_tiles(rgb)
    p = params if params is not None else _ffi.default_params()
    out = torch.empty((16,), dtype=torch.float64, device=rgb.device)
    _ffi.check(_ffi.lib().sl_pool2_sweep(_ptr(rgb), n, h, w, C.byref(p), int(sample_log2), _ptr(state), _ptr(ws), ws.numel(), _ptr(out),
                                         _stream()), "sl_pool2_sweep")
    return out


def pool2_exact(totals16, state):
    _ffi.check(_ffi.lib().sl_pool2_exact(_ptr(totals16), _ptr(state), _stream()), "sl_pool2_exact")


def pool2_local(rgb, sample_log2, ws, state=None, params=None):
    """The whole one-sweep chain on ONE process, enqueued by one call (sl_pool2_local).  Returns the state tensor."""
    n, h, w = _check_tiles(rgb)
    p = params if params is not None else _ffi.default_params()
    if state is None:
        state = torch.empty((_ffi.POOL2_STATE_DOUBLES,), dtype=torch.float64, device=rgb.device)
    _ffi.check(_ffi.lib().sl_pool2_local(_ptr(rgb), n, h, w, C.byref(p), int(sample_log2), _ptr(ws), ws.numel(), _ptr(state), _stream()),
               "sl_pool2_local")
    return state


def pool2_step(state, keyset, hist_reduced):
    """One level of the exact selection on the candidates (see sl_pool2_step)."""
    _ffi.check(_ffi.lib().sl_pool2_step(_ptr(state), int(keyset), _ptr(hist_reduced), _stream()), "sl_pool2_step")


def slide_key_next_above(rgb, keyset, basis, key_ords, params=None):
    """Per target: smallest key (ordered uint32, Python ints) above key_ords[t] among this process's pixels; 0xffffffff if none."""
    n, h, w = _check_tiles(rgb)
    p = params if params is not None else _ffi.default_params()
    mn = torch.full((2,), -1, dtype=torch.int32, device=rgb.device)        # 0xffffffff
    keep, bp = _basis6(basis)
    ko = (C.c_uint32 * 2)(int(key_ords[0]) & 0xffffffff, int(key_ords[1]) & 0xffffffff)
    _ffi.check(_ffi.lib().sl_slide_key_next_above(_ptr(rgb), n, h, w, C.byref(p), int(keyset), bp, ko, _ptr(mn), _stream()),
               "sl_slide_key_next_above")
    v = mn.cpu().tolist()
    return [int(v[0]) & 0xffffffff, int(v[1]) & 0xffffffff]
