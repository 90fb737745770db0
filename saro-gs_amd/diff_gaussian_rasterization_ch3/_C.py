"""`_C` -- the native binding layer of the drop-in package.

In the reference this module is a pybind11 torch extension (ext.cpp:15-19) exposing
``rasterize_gaussians``, ``rasterize_gaussians_backward`` and ``mark_visible``
(rasterize_points.cu:35-215).  Here the same three callables, with the same positional argument
lists and the same return tuples, marshal torch tensors onto the C ABI of ``include/gsrast.h``
(``libgsrast_hip.so``, hand-written HIP for gfx950) through ctypes: raw device pointers, sizes and
the current HIP stream -- no torch types cross the boundary.

There is NO fallback: if the shared library is missing or a tensor is not on a GPU the call raises.
"""
from __future__ import annotations

import ctypes as C
import itertools
import os
import threading
from typing import List, Optional, Tuple

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgsrast_hip.so")

NUM_CHANNELS = 3  # reference config.h:15
ABI_VERSION = 5   # include/gsrast.h: GSRAST_ABI_VERSION this binding was written against

_ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)
_lib: Optional[C.CDLL] = None

# every symbol include/gsrast.h declares (tests check the library exports all of them)
EXPORTS = (
    "gsrast_forward", "gsrast_backward", "gsrast_mark_visible", "gsrast_geometry_bytes",
    "gsrast_binning_bytes", "gsrast_image_bytes", "gsrast_debug_export", "gsrast_set_option",
    "gsrast_get_option", "gsrast_profile_kernel_count", "gsrast_profile_kernel_name",
    "gsrast_profile_collect", "gsrast_profile_read", "gsrast_profile_reset", "gsrast_last_error",
    "gsrast_abi_version", "gsrast_loss_scratch_bytes", "gsrast_loss_forward", "gsrast_loss_backward",
    "gsrast_sh_grad_combine", "gsrast_sh_grad_combine_rows", "gsrast_sh_grad_combine_union", "gsrast_rows_pack", "gsrast_rows_unpack", "gsrast_grad_rows_pack", "gsrast_grad_rows_clear", "gsrast_grad_rows_add", "gsrast_touched_rows", "gsrast_activate_forward", "gsrast_activate_backward", "gsrast_adam_step",
    "gsrast_knn_scratch_bytes", "gsrast_knn3_mean_dist2",
    "gsrast_hexplane_scratch_bytes", "gsrast_hexplane_forward", "gsrast_hexplane_backward",
    "gsrast_options_init", "gsrast_context_create", "gsrast_context_destroy", "gsrast_context_query", "gsrast_policy_event",
    "gsrast_forward_ex", "gsrast_backward_ex", "gsrast_forward_raw", "gsrast_backward_raw", "gsrast_alloc_prealloc",
)


class OptionsStruct(C.Structure):
    """gsrast_options (include/gsrast.h): everything that changes what ONE call computes / how it is scheduled."""
    _fields_ = [("exp_mode", C.c_int), ("binning", C.c_int), ("tile_clip", C.c_int), ("cull", C.c_int), ("lpt", C.c_int),
                ("speculative", C.c_int), ("fwd_pixels_per_lane", C.c_int), ("bwd_pixels_per_lane", C.c_int),
                ("sh_grad_factors", C.c_int), ("side_stream", C.c_int), ("grads_zeroed", C.c_int), ("backward_phase", C.c_int), ("depth_sort", C.c_int),
                ("forward_only", C.c_int), ("no_order_hint", C.c_int), ("dense_backward", C.c_int), ("no_list_cut", C.c_int)]


# Per-call options are kept PER HOST THREAD on the Python side and travel with every call (gsrast_forward_ex /
# gsrast_backward_ex): two threads rendering on two streams with different options never see each other's settings.
PER_CALL_OPTIONS = ("exp_mode", "binning", "tile_clip", "cull", "lpt", "speculative", "fwd_pixels_per_lane", "bwd_pixels_per_lane", "side_stream", "depth_sort", "forward_only", "no_order_hint", "dense_backward", "no_list_cut")
_OPTION_DEFAULTS = dict(exp_mode=0, binning=0, tile_clip=1, cull=1, lpt=1, speculative=1, fwd_pixels_per_lane=0, bwd_pixels_per_lane=0, side_stream=1, depth_sort=0, forward_only=0, no_order_hint=0, dense_backward=0, no_list_cut=0)
_OPTION_RANGE = dict(exp_mode=(0, 1, 2), binning=(0, 1), depth_sort=(0, 1), fwd_pixels_per_lane=(0, 1, 2, 4), bwd_pixels_per_lane=(0, 1, 2, 4))
_tls = threading.local()


def _thread_options() -> dict:
    o = getattr(_tls, "options", None)
    if o is None:
        o = _tls.options = dict(_OPTION_DEFAULTS)
    return o


def current_options() -> dict:
    """A copy of the calling thread's per-call options.  The autograd node stores it at forward time: autograd runs the
    backward on ITS OWN worker thread, which must use the options of the thread that issued the forward."""
    return dict(_thread_options())


_options_cache: dict = {}
_options_epoch = itertools.count(1)


def _options_struct(sh_grad_factors: bool = False, options: Optional[dict] = None, grads_zeroed: bool = False,
                    backward_phase: int = 0, forward_only: bool = False) -> OptionsStruct:
    """The gsrast_options value of one call.  The library copies it at entry, so one struct per distinct combination is built once and
    reused (round 6: fourteen setattr on a fresh ctypes Structure were ~8 us of every call's host time in front of the first launch)."""
    src = _thread_options() if options is None else options
    # (the calling thread's own dict is identified by its version -- bumped by set_option -- instead of by its fourteen items)
    ident = ("tls", getattr(_tls, "options_version", 0)) if options is None else tuple(src.items())      # (0 = the defaults; every set_option draws a process-wide unique number)
    key = (ident, bool(sh_grad_factors), bool(grads_zeroed), int(backward_phase), bool(forward_only))
    o = _options_cache.get(key)
    if o is None:
        o = OptionsStruct()
        for k, v in src.items():
            setattr(o, k, int(v))
        o.forward_only = int(bool(forward_only) or bool(o.forward_only))
        o.sh_grad_factors = int(bool(sh_grad_factors))
        o.grads_zeroed = int(bool(grads_zeroed))
        o.backward_phase = int(backward_phase)
        if len(_options_cache) > 256:
            _options_cache.clear()
        _options_cache[key] = o
    return o


class RawInputsStruct(C.Structure):
    """gsrast_raw_inputs (include/gsrast.h)."""
    _fields_ = [(n, C.c_void_p) for n in ("xyz", "motion_res", "rotation", "rot_res", "scaling", "opacity_logit", "trbf",
                                          "features_dc", "features_rest", "shs_res")]


class RawGradsStruct(C.Structure):
    """gsrast_raw_grads (include/gsrast.h)."""
    _fields_ = [(n, C.c_void_p) for n in ("dL_dmean2D", "d_xyz", "d_rotation", "d_scaling", "d_rot_res", "d_opacity_logit", "d_trbf",
                                          "d_features_dc", "d_features_rest", "d_shs_res", "d_sh_factor")]


class AdamGroupStruct(C.Structure):
    """gsrast_adam_group (include/gsrast.h)."""
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("lr_rows", C.c_void_p), ("lr", C.c_float), ("rows", C.c_int), ("width", C.c_int)]


class PlaneStruct(C.Structure):
    """gsrast_plane (include/gsrast.h)."""
    _fields_ = [("tex", C.c_void_p), ("grad_tex", C.c_void_p), ("W", C.c_int), ("H", C.c_int), ("cu", C.c_int), ("cv", C.c_int),
                ("max_mip_level", C.c_int), ("out_offset", C.c_int)]


def lib() -> C.CDLL:
    """Load libgsrast_hip.so (built by saro-gs_amd/build.py / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python saro-gs_amd/build.py` "
            "(hipcc --offload-arch=gfx950).  There is no CPU or PyTorch fallback for the rasterizer.")
    L = C.CDLL(LIB_PATH)
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    L.gsrast_forward.restype = ci
    L.gsrast_forward.argtypes = [_ALLOC_FN, vp, _ALLOC_FN, vp, _ALLOC_FN, vp, ci, ci, ci, vp, ci, ci, vp, vp, vp, vp,
                                 vp, cf, vp, vp, vp, vp, vp, cf, cf, ci, vp, vp, vp, vp]
    L.gsrast_backward.restype = ci
    L.gsrast_backward.argtypes = [ci, ci, ci, ci, vp, ci, ci, vp, vp, vp, vp, cf, vp, vp, vp, vp, vp, cf, cf, vp,
                                  vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.gsrast_forward_ex.restype = ci
    L.gsrast_forward_ex.argtypes = [vp, C.POINTER(OptionsStruct)] + L.gsrast_forward.argtypes
    L.gsrast_backward_ex.restype = ci
    L.gsrast_backward_ex.argtypes = [C.POINTER(OptionsStruct)] + L.gsrast_backward.argtypes
    L.gsrast_forward_raw.restype = ci
    L.gsrast_forward_raw.argtypes = [vp, C.POINTER(OptionsStruct), _ALLOC_FN, vp, _ALLOC_FN, vp, _ALLOC_FN, vp, ci, ci, ci, vp, ci, ci,
                                     C.POINTER(RawInputsStruct), cf, vp, vp, vp, cf, cf, vp, vp, vp, vp]
    L.gsrast_backward_raw.restype = ci
    L.gsrast_backward_raw.argtypes = [C.POINTER(OptionsStruct), ci, ci, ci, ci, vp, ci, ci, C.POINTER(RawInputsStruct), cf, vp, vp, vp, cf, cf,
                                      vp, vp, vp, vp, vp, C.POINTER(RawGradsStruct), vp]
    L.gsrast_options_init.restype = None
    L.gsrast_options_init.argtypes = [C.POINTER(OptionsStruct)]
    L.gsrast_context_create.restype = vp
    L.gsrast_context_destroy.restype = None
    L.gsrast_context_destroy.argtypes = [vp]
    L.gsrast_context_query.restype = ci
    L.gsrast_context_query.argtypes = [vp, C.c_char_p]
    L.gsrast_policy_event.restype = ci
    L.gsrast_policy_event.argtypes = [vp, C.c_char_p, ci, ci, ci]
    L.gsrast_mark_visible.restype = ci
    L.gsrast_mark_visible.argtypes = [ci, vp, vp, vp, vp, vp]
    for name in ("gsrast_geometry_bytes",):
        getattr(L, name).restype = C.c_size_t
        getattr(L, name).argtypes = [ci]
    L.gsrast_binning_bytes.restype = C.c_size_t
    L.gsrast_binning_bytes.argtypes = [ci, ci, ci]
    L.gsrast_image_bytes.restype = C.c_size_t
    L.gsrast_image_bytes.argtypes = [ci, ci]
    global _PREALLOC_CB
    _PREALLOC_CB = _ALLOC_FN(C.cast(L.gsrast_alloc_prealloc, C.c_void_p).value)      # the library's own C callback over pre-allocated memory (no trip into Python)
    L.gsrast_debug_export.restype = ci
    L.gsrast_debug_export.argtypes = [ci, ci, ci, ci] + [vp] * 16
    L.gsrast_set_option.restype = ci
    L.gsrast_set_option.argtypes = [C.c_char_p, ci]
    L.gsrast_get_option.restype = ci
    L.gsrast_get_option.argtypes = [C.c_char_p]
    L.gsrast_profile_kernel_count.restype = ci
    L.gsrast_profile_kernel_name.restype = C.c_char_p
    L.gsrast_profile_kernel_name.argtypes = [ci]
    L.gsrast_profile_collect.restype = ci
    L.gsrast_profile_read.restype = ci
    L.gsrast_profile_read.argtypes = [ci, C.POINTER(C.c_double), C.POINTER(C.c_longlong)]
    L.gsrast_profile_reset.restype = None
    L.gsrast_loss_scratch_bytes.restype = C.c_size_t
    L.gsrast_loss_scratch_bytes.argtypes = [ci, ci, ci]
    L.gsrast_loss_forward.restype = ci
    L.gsrast_loss_forward.argtypes = [ci, ci, ci, vp, vp, cf, vp, vp, vp]
    L.gsrast_loss_backward.restype = ci
    L.gsrast_loss_backward.argtypes = [ci, ci, ci, vp, vp, cf, vp, vp, vp, vp]
    L.gsrast_sh_grad_combine.restype = ci
    L.gsrast_sh_grad_combine.argtypes = [ci, ci, ci, ci, vp, vp, C.c_size_t, cf, vp, vp]
    L.gsrast_touched_rows.restype = ci
    L.gsrast_touched_rows.argtypes = [ci, vp, vp, vp]
    L.gsrast_sh_grad_combine_rows.restype = ci
    L.gsrast_sh_grad_combine_rows.argtypes = [ci, ci, ci, ci, vp, vp, C.c_size_t, ci, vp, cf, vp, vp, vp, vp]
    for fn in (L.gsrast_rows_pack, L.gsrast_rows_unpack):
        fn.restype = ci
        fn.argtypes = [C.c_longlong, vp, ci, C.POINTER(vp), C.POINTER(ci), vp, vp]
    L.gsrast_grad_rows_pack.restype = ci
    L.gsrast_grad_rows_pack.argtypes = [ci, vp, C.POINTER(vp), vp, vp, C.c_uint32, vp]
    L.gsrast_grad_rows_clear.restype = ci
    L.gsrast_grad_rows_clear.argtypes = [ci, vp, ci, C.c_size_t, C.c_uint32, C.POINTER(vp), ci, vp, vp, vp, vp]
    L.gsrast_grad_rows_add.restype = ci
    L.gsrast_grad_rows_add.argtypes = [ci, vp, C.c_uint32, C.POINTER(vp), ci, ci, vp, cf, vp, vp, vp, vp]
    L.gsrast_sh_grad_combine_union.restype = ci
    L.gsrast_sh_grad_combine_union.argtypes = [ci, ci, ci, ci, vp, vp, C.c_size_t, ci, vp, cf, vp, vp, vp, vp]
    L.gsrast_activate_forward.restype = ci
    L.gsrast_activate_forward.argtypes = [ci, ci] + [vp] * 16
    L.gsrast_activate_backward.restype = ci
    L.gsrast_activate_backward.argtypes = [ci] + [vp] * 14
    L.gsrast_knn_scratch_bytes.restype = C.c_size_t
    L.gsrast_knn_scratch_bytes.argtypes = [ci]
    L.gsrast_knn3_mean_dist2.restype = ci
    L.gsrast_knn3_mean_dist2.argtypes = [ci, vp, vp, vp, vp]
    L.gsrast_adam_step.restype = ci
    L.gsrast_adam_step.argtypes = [ci, C.POINTER(AdamGroupStruct), C.c_double, C.c_double, C.c_double, ci, vp]
    L.gsrast_hexplane_scratch_bytes.restype = C.c_size_t
    L.gsrast_hexplane_scratch_bytes.argtypes = [ci, C.POINTER(PlaneStruct), ci, ci]
    L.gsrast_hexplane_forward.restype = ci
    L.gsrast_hexplane_forward.argtypes = [ci, ci, ci, ci, ci, C.POINTER(PlaneStruct), vp, vp, vp, vp, vp]
    L.gsrast_hexplane_backward.restype = ci
    L.gsrast_hexplane_backward.argtypes = [ci, ci, ci, ci, ci, C.POINTER(PlaneStruct), vp, vp, vp, vp, vp, ci, vp, vp]
    L.gsrast_last_error.restype = C.c_char_p
    L.gsrast_abi_version.restype = ci
    if L.gsrast_abi_version() != ABI_VERSION:
        raise ImportError("libgsrast_hip.so: ABI version mismatch")
    _lib = L
    return L


def _err(code: int, where: str) -> RuntimeError:
    msg = lib().gsrast_last_error()
    return RuntimeError(f"{where} failed ({code}): {msg.decode() if msg else ''}")


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    """Device pointer, or NULL for an absent optional input (the reference passes CPU
    ``torch.Tensor([])`` and tests for a null data pointer, __init__.py:173-183)."""
    if t is None or t.numel() == 0:
        return None
    return t.data_ptr()


def _dev_f32(t: torch.Tensor, name: str, device: torch.device) -> torch.Tensor:
    if t.dtype is torch.float32 and t.device == device and t.is_contiguous():      # (the usual case: nothing to do)
        return t
    if t.numel() == 0:
        return t
    if t.device != device:
        raise RuntimeError(f"{name} must live on {device} (got {t.device})")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be float32 (got {t.dtype})")
    return t.contiguous()


def _require_gpu(t: torch.Tensor) -> torch.device:
    if not t.is_cuda:
        raise RuntimeError("gsrast: tensors must be on a GPU (HIP) device; there is no CPU fallback")
    return t.device


class GradArena:
    """Optional zero-copy gradient bucket for the multi-GPU path (view_parallel.py).

    When installed with ``set_grad_arena``, ``rasterize_gaussians_backward`` carves the leaf
    gradients it returns (means3D, sh, opacities, scales, rotations) out of ONE flat fp32 buffer, in
    that order, so the cross-rank exchange is a single in-place all-reduce of ``flat`` with no pack /
    unpack copies (the same idea as DDP's gradient-as-bucket-view).  The screen-space gradient
    (means2D) is deliberately NOT in the bucket: the reference only ever uses its per-view norm
    (train.py:212), which is reduced separately as a [P] statistic.  Not part of the reference's _C:
    without an arena the outputs are ordinary tensors, exactly as before.

    Contract: the arena holds the gradients of ONE step.  The first backward after ``zero_grad()`` writes its leaf gradients
    into the arena (autograd adopts the views as ``.grad``); a further backward before the next ``zero_grad()`` -- the
    reference's batch loop, or ``views_of_rank`` yielding several views per rank -- gets ordinary tensors, which autograd
    ADDS into those ``.grad`` views (the sum of the views' gradients, as ``cache_gradient`` keeps it,
    scene/saro_gaussian.py:242-247).  With ``sh_factors=True`` a second backward raises: the factor buffer holds exactly one
    view's factor.  Call ``zero_grad()`` (after ``optimizer.zero_grad(set_to_none=True)``) at the start of every step."""

    ORDER = (("means3D", 3), ("sh", None), ("opacity", 1), ("scales", 3), ("rotations", 4))
    # sh_factors=True: the dense part first (one contiguous all-reduce), dL/dsh last -- it is not exchanged at all: the
    # backward writes the per-view FACTOR g[P,3] of dL/dsh (include/gsrast.h, gsrast_sh_grad_combine) into `factor`,
    # ranks all-gather the factors (+ their camera positions) and every rank recombines dL/dsh locally.
    ORDER_FACTORS = (("means3D", 3), ("opacity", 1), ("scales", 3), ("rotations", 4), ("sh", None))
    # raw=True (round 4): the bucket of GaussianRasterizerRaw's six LEAVES -- SaRO-GS's own call pattern, where the rasterizer's
    # `shs` is cat(features_dc, features_rest) [+ residual], never a leaf (scene/saro_gaussian.py:836-845): dense part first, the
    # two SH leaves last (with sh_factors they are completed by gsrast_sh_grad_combine_rows after the exchange)
    ORDER_RAW = (("xyz", 3), ("opacity_logit", 1), ("scaling", 3), ("rotation", 4), ("features_dc", 3), ("features_rest", -1))

    def __init__(self, P: int, M: int, device: torch.device, sh_factors: bool = False, world: int = 1, raw: bool = False):
        self.P, self.M, self.sh_factors, self.world, self.raw = P, M, bool(sh_factors), int(world), bool(raw)
        order = self.ORDER_RAW if raw else (self.ORDER_FACTORS if sh_factors else self.ORDER)
        self.widths = {name: (M * 3 if w is None else ((M - 1) * 3 if w == -1 else w)) for name, w in order}
        self.offsets, o = {}, 0
        for name, _ in order:
            self.offsets[name] = o
            o += ((P * self.widths[name] + 3) // 4) * 4        # every segment starts on a 16-byte boundary (float4 stores of dL/drot)
        self.flat = torch.zeros(o, dtype=torch.float32, device=device)
        self.dirty = False                                     # a backward has written this step's gradients
        self.dense_names = tuple(n for n, _ in order if n not in ("sh", "features_dc", "features_rest"))
        if sh_factors:
            self.dense = self.flat[: self.offsets["features_dc" if raw else "sh"]]   # 11 floats / Gaussian: the all-reduced part
            self.chunk = ((3 * P + 3 + 3) // 4) * 4                            # [3P g | 3 campos | pad], 16-byte multiple
            self.factor = torch.zeros(self.chunk, dtype=torch.float32, device=device)
            self.gathered = torch.zeros(self.world * self.chunk, dtype=torch.float32, device=device)
        self.last_degree = 0

    def zero_grad(self) -> None:
        """Start of a step: the next backward writes into the arena again.  (Does not touch memory: the backward overwrites
        every element; set the leaves' .grad to None first, or autograd would add the arena to itself.)"""
        self.dirty = False

    def dense_segments(self):
        """The dense (non-SH) gradient arrays as [P, w] views of the bucket, in bucket order."""
        return [self.flat[self.offsets[n]: self.offsets[n] + self.P * self.widths[n]].view(self.P, self.widths[n]) for n in self.dense_names]

    def take(self, name: str, shape, zero: bool) -> torch.Tensor:
        n = self.P * self.widths[name]
        v = self.flat[self.offsets[name]: self.offsets[name] + n].view(shape)   # a fresh view every call
        if zero:
            v.zero_()
        return v


def _export_touched(ar: "GradArena", P: int, geomBuffer: torch.Tensor, dev: torch.device) -> None:
    """gsrast_touched_rows into the arena (round 5): one byte per Gaussian, 1 = some pixel of THIS view consumed it -- what the sparse
    exchange (view_parallel.exchange_gradients(sparse=True)) takes the union over ranks of, instead of scanning the gradient arrays."""
    pending = getattr(ar, "touched_reader_event", None)      # somebody still reads the previous step's flags on another stream
    if pending is not None:
        pending.synchronize()
        ar.touched_reader_event = None
    t = getattr(ar, "touched", None)
    if t is None or t.numel() != P or t.device != dev:
        t = ar.touched = torch.empty(P, dtype=torch.uint8, device=dev)
    with _on_device(dev):
        rc = lib().gsrast_touched_rows(P, _ptr(geomBuffer), t.data_ptr(), _stream_of(dev))
    if rc != 0:
        raise _err(rc, "gsrast_touched_rows")
    ar.touched_fresh = True
    ar.touched_seq = getattr(ar, "touched_seq", 0) + 1       # which backward these flags belong to (view_parallel._touched_hook tags its capacity event with it)


_grad_arena: Optional[GradArena] = None
_factor_ready_hook = None
_touched_ready_hook = None


def set_grad_arena(arena: Optional[GradArena]) -> None:
    global _grad_arena
    _grad_arena = arena


def set_factor_ready_hook(fn) -> None:
    """fn(arena) is called INSIDE the backward of a factor-mode arena, between the blend backward (after which arena.factor --
    the view's factor of dL/dsh and its camera position -- is final on the current stream) and the per-Gaussian backward:
    view_parallel starts the asynchronous all-gather of the factors there, so that it runs beside the second phase."""
    global _factor_ready_hook
    _factor_ready_hook = fn


def set_touched_ready_hook(fn) -> None:
    """fn(arena) is called INSIDE the backward of a factor-mode arena BEFORE any of its kernels is enqueued, right after arena.touched
    (one byte per Gaussian: some pixel of this view consumed it) has been written on the current stream."""
    global _touched_ready_hook
    _touched_ready_hook = fn


POISON_STATE_BUFFERS = bool(int(os.environ.get("GSRAST_POISON_STATE", "0")))      # tests: every state buffer is handed out filled with 0xFF bytes (NaN as floats, all-ones as bits / indices)


_get_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream_of(dev: torch.device) -> int:
    """The raw handle of torch's current stream on `dev` (what every launch of a call goes to).  torch.cuda.current_stream(dev).cuda_stream builds a
    Stream object per call (~3 us); torch's own raw accessor -- the one its compiled-code launchers use -- returns the handle directly."""
    if _get_raw_stream is not None and dev.index is not None:
        return _get_raw_stream(dev.index)
    return torch.cuda.current_stream(dev).cuda_stream


class PreallocStruct(C.Structure):
    """gsrast_prealloc (include/gsrast.h)."""
    _fields_ = [("ptr", C.c_void_p), ("capacity", C.c_size_t), ("requested", C.c_size_t)]


_PREALLOC_CB = None
PREALLOC_STATE = True       # geometry / image state buffers are allocated BEFORE the forward's C call and handed over through gsrast_alloc_prealloc
                            # (round 6: two Python callbacks fewer in front of the first launch); False: all three through Python callbacks, as rounds 1-5
_size_cache: dict = {}


def _state_bytes(P: int, W: int, H: int):
    k = (P, W, H)
    v = _size_cache.get(k)
    if v is None:
        L = lib()
        if len(_size_cache) > 64:
            _size_cache.clear()
        v = _size_cache[k] = (int(L.gsrast_geometry_bytes(P)), int(L.gsrast_image_bytes(W, H)))
    return v


class _on_device:
    """`with torch.cuda.device(dev)` without its cost when `dev` already is the current device (the usual case: two runtime calls and a
    Python context manager's bookkeeping per entry point, in front of the first launch)."""
    __slots__ = ("dev", "inner")

    def __init__(self, dev: torch.device):
        self.dev, self.inner = dev, None

    def __enter__(self):
        idx = self.dev.index
        if idx is not None and idx != torch.cuda.current_device():
            self.inner = torch.cuda.device(self.dev)
            self.inner.__enter__()
        return self

    def __exit__(self, *exc):
        if self.inner is not None:
            return self.inner.__exit__(*exc)
        return False


class _Arena:
    """The three resizable state buffers of the reference (rasterize_points.cu:27-33, :71-78):
    each allocation callback creates one uint8 tensor that is later saved for backward."""

    def __init__(self, device: torch.device):
        self.device = device
        self.buffers: List[Optional[torch.Tensor]] = [None, None, None]
        self.callbacks = [_ALLOC_FN(self._make(i)) for i in range(3)]  # keep references alive
        self.pre = (PreallocStruct * 2)()                              # geometry, image (gsrast_alloc_prealloc)

    def forward_allocators(self, P: int, W: int, H: int):
        """(geometry_alloc, geometry_ctx, binning_alloc, binning_ctx, image_alloc, image_ctx) of one forward.  With PREALLOC_STATE the geometry
        and image buffers are allocated here, their sizes being known (gsrast_geometry_bytes / gsrast_image_bytes), and the library's own C
        callback hands them out; the binning buffer, whose size the library decides, keeps the Python callback."""
        if not PREALLOC_STATE or P <= 0:
            return (self.callbacks[0], None, self.callbacks[1], None, self.callbacks[2], None)
        gb, ib = _state_bytes(P, W, H)
        geom = torch.empty(gb, dtype=torch.uint8, device=self.device)
        img = torch.empty(ib, dtype=torch.uint8, device=self.device)
        if POISON_STATE_BUFFERS:
            geom.fill_(255); img.fill_(255)
        self.buffers[0], self.buffers[2] = geom, img
        pre = self.pre
        pre[0].ptr, pre[0].capacity = geom.data_ptr(), gb
        pre[1].ptr, pre[1].capacity = img.data_ptr(), ib
        return (_PREALLOC_CB, C.addressof(pre[0]), self.callbacks[1], None, _PREALLOC_CB, C.addressof(pre[1]))

    @staticmethod
    def acquire(device: torch.device) -> "_Arena":
        """An arena of the calling thread's pool (round 6): building three ctypes callbacks per forward cost ~20 us of host time in front of
        the first launch; a pooled arena keeps its callbacks and only ever holds buffers between acquire() and close()."""
        pool = getattr(_tls, "arena_pool", None)
        if pool is None:
            pool = _tls.arena_pool = []
        if pool:
            a = pool.pop()
            a.device = device
            return a
        return _Arena(device)

    def _make(self, slot: int):
        def alloc(_ctx, nbytes):
            try:
                buf = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
                if POISON_STATE_BUFFERS:
                    buf.fill_(255)
            except Exception:  # out of memory -> NULL -> GSRAST_E_ALLOC
                return None
            self.buffers[slot] = buf
            return buf.data_ptr()
        return alloc

    def tensor(self, slot: int) -> torch.Tensor:
        b = self.buffers[slot]
        return b if b is not None else torch.empty(0, dtype=torch.uint8, device=self.device)

    def close(self) -> None:
        """Drop the callbacks and the buffer references.  The callbacks are closures over `self`, so the arena sits in
        a reference cycle: without this the three state buffers (hundreds of MB) stay alive until Python's CYCLIC
        collector happens to run, the caching allocator sees them freed at irregular times and occasionally has to
        hipMalloc fresh blocks in the middle of a training loop (a ~250 ms stall)."""
        self.buffers = [None, None, None]       # (the callbacks stay: the arena goes back to its thread's pool, holding nothing)
        pool = getattr(_tls, "arena_pool", None)
        if pool is not None and len(pool) < 8:
            pool.append(self)
        else:
            self.callbacks = None


def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier,
                        cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height,
                        image_width, sh, degree, campos, prefiltered, *, forward_only: bool = False
                        ) -> Tuple[int, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """Forward.  Mirrors RasterizeGaussiansCUDA (rasterize_points.cu:35-115): returns
    ``(num_rendered, out_color[3,H,W], radii[P] int32, geomBuffer, binningBuffer, imgBuffer,
    out_depth[1,H,W])``.  `forward_only` (not in the reference): no backward will follow on the returned state (the autograd
    node passes it when no input requires a gradient): the library skips what it only prepares for the backward."""
    if means3D.ndim != 2 or means3D.shape[1] != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")  # rasterize_points.cu:56-58
    dev = _require_gpu(means3D)
    L = lib()
    P, H, W = int(means3D.shape[0]), int(image_height), int(image_width)
    f = lambda t, n: _dev_f32(t, n, dev)  # noqa: E731
    background, means3D, colors, opacity = f(background, "bg"), f(means3D, "means3D"), f(colors, "colors_precomp"), f(opacity, "opacities")
    scales, rotations, cov3D_precomp = f(scales, "scales"), f(rotations, "rotations"), f(cov3D_precomp, "cov3D_precomp")
    viewmatrix, projmatrix, sh, campos = f(viewmatrix, "viewmatrix"), f(projmatrix, "projmatrix"), f(sh, "shs"), f(campos, "campos")
    M = int(sh.shape[1]) if sh.numel() != 0 else 0  # rasterize_points.cu:83-87

    out_color = torch.empty((NUM_CHANNELS, H, W), dtype=torch.float32, device=dev)
    out_depth = torch.empty((1, H, W), dtype=torch.float32, device=dev)
    radii = torch.empty((P,), dtype=torch.int32, device=dev)
    arena = _Arena.acquire(dev)
    try:
        with _on_device(dev):
            stream = _stream_of(dev)
            rendered = L.gsrast_forward_ex(
                _current_context(), C.byref(_options_struct(forward_only=forward_only)),       # context: the innermost `with Context()` of the calling thread, else the thread's own
                *arena.forward_allocators(P, W, H),
                P, int(degree), M, background.data_ptr(), W, H, means3D.data_ptr(), _ptr(sh), _ptr(colors), opacity.data_ptr(),
                _ptr(scales), float(scale_modifier), _ptr(rotations), _ptr(cov3D_precomp), viewmatrix.data_ptr(),
                projmatrix.data_ptr(), _ptr(campos), float(tan_fovx), float(tan_fovy), int(bool(prefiltered)),
                out_color.data_ptr(), out_depth.data_ptr(), _ptr(radii), stream)
        if rendered < 0:
            raise _err(rendered, "gsrast_forward")
        return rendered, out_color, radii, arena.tensor(0), arena.tensor(1), arena.tensor(2), out_depth
    finally:
        arena.close()       # break the arena <-> callback cycle now, not whenever the cyclic GC runs


def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier,
                                 cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color,
                                 sh, degree, campos, geomBuffer, R, binningBuffer, imageBuffer, *, options: Optional[dict] = None,
                                 first_backward: bool = False):
    """Backward.  Mirrors RasterizeGaussiansBackwardCUDA (rasterize_points.cu:117-194): returns
    ``(dL_dmeans2D[P,3], dL_dcolors[P,3], dL_dopacity[P,1], dL_dmeans3D[P,3], dL_dcov3D[P,6],
    dL_dsh[P,M,3], dL_dscales[P,3], dL_drotations[P,4])``.  `options` (not in the reference): the per-call options to use
    instead of the calling thread's (current_options() captured at forward time); `first_backward`: no backward has touched
    geomBuffer since its forward, whose gradient records are therefore still zero (the library skips its zero-fill)."""
    dev = _require_gpu(means3D)
    L = lib()
    P = int(means3D.shape[0])
    H, W = int(dL_dout_color.shape[1]), int(dL_dout_color.shape[2])  # rasterize_points.cu:141-142
    f = lambda t, n: _dev_f32(t, n, dev)  # noqa: E731
    background, means3D, colors = f(background, "bg"), f(means3D, "means3D"), f(colors, "colors_precomp")
    scales, rotations, cov3D_precomp = f(scales, "scales"), f(rotations, "rotations"), f(cov3D_precomp, "cov3D_precomp")
    viewmatrix, projmatrix, sh, campos = f(viewmatrix, "viewmatrix"), f(projmatrix, "projmatrix"), f(sh, "shs"), f(campos, "campos")
    dL_dout_color = f(dL_dout_color, "dL_dout_color")
    M = int(sh.shape[1]) if sh.numel() != 0 else 0
    opts = dict(dtype=torch.float32, device=dev)
    use_sh = sh.numel() != 0 and colors.numel() == 0
    use_sr = cov3D_precomp.numel() == 0
    ar = _grad_arena
    if ar is not None and not (ar.P == P and ar.M == M and use_sh and use_sr and ar.flat.device == dev):
        ar = None                       # the arena only serves the SH + scale/rotation training path
    if ar is not None and ar.dirty:     # a second backward of the same step (GradArena docstring)
        if ar.sh_factors:
            raise RuntimeError("GradArena(sh_factors=True): a second backward before zero_grad() would overwrite the first view's "
                               "factor; use one view per rank and exchange, or the plain arena (which accumulates)")
        ar = None                       # ordinary tensors: autograd adds them into the arena views it adopted as .grad
    if ar is not None:
        ar.dirty = True

    def out(name, shape, zero):
        if ar is not None and name in ar.offsets:
            return ar.take(name, shape, zero)
        return torch.zeros(shape, **opts) if zero else torch.empty(shape, **opts)

    # Nothing is zero-filled here: the library accumulates into its own 64-byte-per-Gaussian records inside geomBuffer and
    # writes every returned array exactly once (include/gsrast.h).  dL_dconic (an intermediate of the reference) is not
    # requested at all.
    dL_dmeans2D = torch.empty((P, 3), **opts)
    dL_dopacity = out("opacity", (P, 1), False)
    # with SH colours dL_dcolors is an intermediate nobody reads (the node returns None for the absent colors_precomp input):
    # not written -- an empty tensor in its place, like dL_dcov3D below (12 B / Gaussian of the bandwidth-bound per-Gaussian backward)
    dL_dcolors = torch.empty((0, NUM_CHANNELS) if use_sh else (P, NUM_CHANNELS), **opts)
    dL_dmeans3D = out("means3D", (P, 3), False)
    # with scales / rotations dL_dcov3D is an intermediate nobody reads (the autograd node returns None for the absent
    # cov3D_precomp input): not computed, not written -- the binding returns an empty tensor in its place
    dL_dcov3D = torch.empty((0, 6) if use_sr else (P, 6), **opts)
    dL_dsh = out("sh", (P, M, 3), not use_sh)
    factors = ar is not None and ar.sh_factors
    if factors:
        # dL_dsh (a view of the arena) is returned to autograd as usual but only becomes valid after
        # sh_grad_combine(); the kernel writes this view's factor g[P,3] (+ the camera position behind it)
        ar.factor[3 * P: 3 * P + 3].copy_(campos.reshape(-1)[:3])
        ar.last_degree = int(degree)
    dL_dscales = out("scales", (P, 3), not use_sr)
    dL_drotations = out("rotations", (P, 4), not use_sr)
    if P != 0:
        with _on_device(dev):
            stream = _stream_of(dev)
            sh_out = ar.factor.data_ptr() if factors else _ptr(dL_dsh)
            radii_c = radii.contiguous()

            def call(phase):      # options travel per call: no process-wide switch is flipped
                return L.gsrast_backward_ex(
                    C.byref(_options_struct(sh_grad_factors=factors, options=options, grads_zeroed=first_backward, backward_phase=phase)),
                    P, int(degree), M, int(R), _ptr(background), W, H, _ptr(means3D), _ptr(sh), _ptr(colors),
                    _ptr(scales), float(scale_modifier), _ptr(rotations), _ptr(cov3D_precomp), _ptr(viewmatrix),
                    _ptr(projmatrix), _ptr(campos), float(tan_fovx), float(tan_fovy), _ptr(radii_c),
                    _ptr(geomBuffer), _ptr(binningBuffer), _ptr(imageBuffer), _ptr(dL_dout_color),
                    dL_dmeans2D.data_ptr(), None, dL_dopacity.data_ptr(), _ptr(dL_dcolors),
                    dL_dmeans3D.data_ptr(), _ptr(dL_dcov3D), sh_out, dL_dscales.data_ptr(),
                    dL_drotations.data_ptr(), stream)

            if factors:
                # which rows this view can touch is known since the forward's blend (its untouched bits): exported BEFORE the backward
                # is enqueued, so that a caller's hook can start on it -- the all-gather exchange agrees on its row capacity beside
                # the backward instead of waiting for it (view_parallel._touched_hook)
                _export_touched(ar, P, geomBuffer, dev)
                if _touched_ready_hook is not None:
                    _touched_ready_hook(ar)
            if factors and _factor_ready_hook is not None:
                rc = call(1)                     # blend backward + the factors
                if rc == 0:
                    _factor_ready_hook(ar)       # e.g. the asynchronous all-gather of the factors
                    rc = call(2)                 # the per-Gaussian backward, beside it
            else:
                rc = call(0)
        if rc != 0:
            raise _err(rc, "gsrast_backward")
    return dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations


# ---- raw-parameter entry points (include/gsrast.h: gsrast_forward_raw / gsrast_backward_raw; no counterpart in the reference's _C) ----
RAW_NAMES = ("xyz", "motion_res", "rotation", "rot_res", "scaling", "opacity_logit", "trbf", "features_dc", "features_rest", "shs_res")


def _raw_struct(raw: dict, dev: torch.device, P: int):
    """raw: name -> tensor or None (RAW_NAMES).  Returns (struct, the contiguous tensors it points into, M)."""
    keep = {}
    for n in RAW_NAMES:
        t = raw.get(n)
        if t is not None and t.numel() == 0 and P != 0:
            t = None
        keep[n] = None if t is None else _dev_f32(t, n, dev)
    for n in ("xyz", "rotation", "scaling", "opacity_logit", "features_dc", "features_rest"):
        if keep[n] is None:
            raise RuntimeError(f"rasterize_gaussians_raw: {n} is required")
    if keep["xyz"].ndim != 2 or keep["xyz"].shape[1] != 3:
        raise RuntimeError("xyz must have dimensions (num_points, 3)")
    M = 1 + int(keep["features_rest"].shape[1])
    shapes = dict(motion_res=(P, 3), rotation=(P, 4), rot_res=(P, 7), scaling=(P, 3), features_dc=(P, 1, 3), features_rest=(P, M - 1, 3),
                  shs_res=(P, M, 3))
    for n, shp in shapes.items():
        if keep[n] is not None and tuple(keep[n].shape) != shp:
            raise RuntimeError(f"rasterize_gaussians_raw: {n} must be {list(shp)} (got {list(keep[n].shape)})")
    for n in ("opacity_logit", "trbf"):
        if keep[n] is not None and keep[n].numel() != P:
            raise RuntimeError(f"rasterize_gaussians_raw: {n} must hold one value per Gaussian")
    st = RawInputsStruct(**{n: _ptr(keep[n]) for n in RAW_NAMES})
    return st, keep, M


def rasterize_gaussians_raw(background, raw: dict, scale_modifier, viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width,
                            degree, campos, *, forward_only: bool = False):
    """rasterize_gaussians taking the model's raw leaves + optional residuals (`raw`: RAW_NAMES -> tensor / None); the activations
    of scene/saro_gaussian.py:39-47, :807-847 run inside the per-Gaussian kernels.  Same return tuple."""
    dev = _require_gpu(raw["xyz"])
    L = lib()
    P, H, W = int(raw["xyz"].shape[0]), int(image_height), int(image_width)
    st, keep, M = _raw_struct(raw, dev, P)
    f = lambda t, n: _dev_f32(t, n, dev)  # noqa: E731
    background, viewmatrix, projmatrix, campos = f(background, "bg"), f(viewmatrix, "viewmatrix"), f(projmatrix, "projmatrix"), f(campos, "campos")
    out_color = torch.empty((NUM_CHANNELS, H, W), dtype=torch.float32, device=dev)
    out_depth = torch.empty((1, H, W), dtype=torch.float32, device=dev)
    radii = torch.empty((P,), dtype=torch.int32, device=dev)
    arena = _Arena.acquire(dev)
    try:
        with _on_device(dev):
            rendered = L.gsrast_forward_raw(
                _current_context(), C.byref(_options_struct(forward_only=forward_only)),
                *arena.forward_allocators(P, W, H),
                P, int(degree), M, _ptr(background), W, H, C.byref(st), float(scale_modifier), _ptr(viewmatrix), _ptr(projmatrix), _ptr(campos),
                float(tan_fovx), float(tan_fovy), out_color.data_ptr(), out_depth.data_ptr(), _ptr(radii),
                _stream_of(dev))
        if rendered < 0:
            raise _err(rendered, "gsrast_forward_raw")
        return rendered, out_color, radii, arena.tensor(0), arena.tensor(1), arena.tensor(2), out_depth
    finally:
        arena.close()


def rasterize_gaussians_raw_backward(background, raw: dict, radii, scale_modifier, viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color,
                                     degree, campos, geomBuffer, R, binningBuffer, imageBuffer, *, options: Optional[dict] = None,
                                     first_backward: bool = False) -> dict:
    """Gradients of the raw leaves: dict with dL_dmeans2D [P,3], xyz (= motion_res), rotation, scaling, opacity_logit [P,1], features_dc,
    features_rest, and -- when the residual was given -- rot_res [P,7], trbf [P,1], shs_res [P,M,3]."""
    dev = _require_gpu(raw["xyz"])
    L = lib()
    P = int(raw["xyz"].shape[0])
    H, W = int(dL_dout_color.shape[1]), int(dL_dout_color.shape[2])
    st, keep, M = _raw_struct(raw, dev, P)
    f = lambda t, n: _dev_f32(t, n, dev)  # noqa: E731
    background, viewmatrix, projmatrix, campos = f(background, "bg"), f(viewmatrix, "viewmatrix"), f(projmatrix, "projmatrix"), f(campos, "campos")
    dL_dout_color = f(dL_dout_color, "dL_dout_color")
    o = dict(dtype=torch.float32, device=dev)
    ar = _grad_arena
    if ar is not None and not (getattr(ar, "raw", False) and ar.P == P and ar.M == M and ar.flat.device == dev):
        ar = None                       # (the bucket of another call shape)
    if ar is not None and ar.dirty:     # a second backward of the same step (GradArena docstring)
        if ar.sh_factors:
            raise RuntimeError("GradArena(sh_factors=True): a second backward before zero_grad() would overwrite the first view's "
                               "factor; use one view per rank and exchange, or the plain arena (which accumulates)")
        ar = None
    factors = ar is not None and ar.sh_factors
    if factors and keep["shs_res"] is not None:
        raise RuntimeError("GradArena(raw=True, sh_factors=True) cannot serve a call with shs_residual: every rank needs the whole "
                           "gradient of its own residual; use GradArena(raw=True) + view_parallel.allreduce_mean_inplace")
    if ar is not None:
        ar.dirty = True

    def out(name, shape):
        return ar.take(name, shape, False) if ar is not None else torch.empty(shape, **o)

    g = dict(dL_dmeans2D=torch.empty((P, 3), **o), xyz=out("xyz", (P, 3)), rotation=out("rotation", (P, 4)), scaling=out("scaling", (P, 3)),
             opacity_logit=out("opacity_logit", (P, 1)))
    if keep["rot_res"] is not None:
        g["rot_res"] = torch.empty((P, 7), **o)
    if keep["trbf"] is not None:
        g["trbf"] = torch.empty((P, 1), **o)
    if keep["shs_res"] is not None:
        g["shs_res"] = torch.empty((P, M, 3), **o)
    # the two SH leaves get their own contiguous gradients also when the residual's gradient holds the same rows: autograd would copy
    # strided slices of it into the leaves' .grad (two elementwise kernels, 96 us at 1 M), the kernel writes them for a third of that
    g["features_dc"], g["features_rest"] = out("features_dc", (P, 1, 3)), out("features_rest", (P, M - 1, 3))
    p_dc, p_rest = g["features_dc"].data_ptr(), _ptr(g["features_rest"])
    p_fac = None
    if factors:
        # the two SH leaves' gradients (views of the bucket) are returned to autograd as usual but only become valid after
        # sh_grad_combine(); the kernel writes this view's factor g[P,3], the camera position goes behind it
        ar.factor[3 * P: 3 * P + 3].copy_(campos.reshape(-1)[:3])
        ar.last_degree = int(degree)
        p_dc, p_rest, p_fac = None, None, ar.factor.data_ptr()
    gs = RawGradsStruct(dL_dmean2D=g["dL_dmeans2D"].data_ptr(), d_xyz=g["xyz"].data_ptr(), d_rotation=g["rotation"].data_ptr(),
                        d_scaling=g["scaling"].data_ptr(), d_rot_res=_ptr(g.get("rot_res")), d_opacity_logit=g["opacity_logit"].data_ptr(),
                        d_trbf=_ptr(g.get("trbf")), d_features_dc=p_dc, d_features_rest=p_rest, d_shs_res=_ptr(g.get("shs_res")),
                        d_sh_factor=p_fac)
    if P != 0:
        radii_c = radii.contiguous()
        with _on_device(dev):
            def call(phase):
                return L.gsrast_backward_raw(
                    C.byref(_options_struct(options=options, grads_zeroed=first_backward, backward_phase=phase)), P, int(degree), M, int(R),
                    _ptr(background), W, H, C.byref(st), float(scale_modifier), _ptr(viewmatrix), _ptr(projmatrix), _ptr(campos),
                    float(tan_fovx), float(tan_fovy), _ptr(radii_c), _ptr(geomBuffer), _ptr(binningBuffer), _ptr(imageBuffer),
                    _ptr(dL_dout_color), C.byref(gs), _stream_of(dev))

            if factors:
                # which rows this view can touch is known since the forward's blend (its untouched bits): exported BEFORE the backward
                # is enqueued, so that a caller's hook can start on it -- the all-gather exchange agrees on its row capacity beside
                # the backward instead of waiting for it (view_parallel._touched_hook)
                _export_touched(ar, P, geomBuffer, dev)
                if _touched_ready_hook is not None:
                    _touched_ready_hook(ar)
            if factors and _factor_ready_hook is not None:
                rc = call(1)                     # blend backward + the factors
                if rc == 0:
                    _factor_ready_hook(ar)       # e.g. the asynchronous all-gather of the factors
                    rc = call(2)                 # the per-Gaussian backward, beside it
            else:
                rc = call(0)
        if rc != 0:
            raise _err(rc, "gsrast_backward_raw")
    if keep["motion_res"] is not None:
        g["motion_res"] = g["xyz"]
    return g


def sh_grad_combine(arena: "GradArena", means3D: torch.Tensor, chunks: torch.Tensor, n_views: int, scale: float,
                    rows: Optional[int] = None, row_of: Optional[torch.Tensor] = None, chunk_stride: Optional[int] = None,
                    idx: Optional[torch.Tensor] = None):
    """dL/dsh of `n_views` views from their factors (include/gsrast.h: gsrast_sh_grad_combine / _rows / _union), written into the
    arena's SH region(s) -- the tensor(s) autograd already handed out as shs.grad, or as features_dc.grad / features_rest.grad for a
    raw arena.  rows / row_of: the records hold only `rows` factors, Gaussian i's is row row_of[i] (int32 [P], -1 = not sent).

    idx (int64 [rows], ascending, distinct; round 5) instead of row_of: record row j is Gaussian idx[j]'s, and ONLY those rows of the
    SH region are written.  The region is kept zero elsewhere from one step to the next: the rows of the previous union are cleared
    first (the whole region once after a dense combine or a fresh arena).  Whoever writes into those .grad tensors in a way that makes
    a zero row non-zero must set arena.sh_rows_known = False."""
    L = lib()
    P, M = arena.P, arena.M
    if getattr(arena, "raw", False):
        whole, dc, rest = None, arena.take("features_dc", (P, 1, 3), False), arena.take("features_rest", (P, M - 1, 3), False)
    else:
        whole, dc, rest = arena.take("sh", (P, M, 3), False), None, None
    if P == 0:
        return whole if whole is not None else (dc, rest)
    dev = means3D.device
    union = idx is not None and (M * 3) % 4 == 0 and M * 3 <= 48
    if union:
        views = [v.view(P, -1) for v in (whole, dc, rest) if v is not None and v.numel()]
        prev = getattr(arena, "_sh_union", None)
        arena._rows_prev = None                      # (the all-gather exchange keeps its own record of the rows it wrote)
        if not getattr(arena, "sh_rows_known", False) or prev is None:
            for v in views:
                v.zero_()
        else:
            for v in views:
                v.index_fill_(0, prev, 0.0)
        arena._sh_union, arena.sh_rows_known = idx, True
        with _on_device(dev):
            rc = L.gsrast_sh_grad_combine_union(P, int(arena.last_degree), M, int(n_views), means3D.data_ptr(), chunks.data_ptr(),
                                                int(chunk_stride if chunk_stride is not None else arena.chunk), int(idx.numel()),
                                                idx.data_ptr(), float(scale), _ptr(whole), _ptr(dc), _ptr(rest),
                                                _stream_of(dev))
        if rc != 0:
            raise _err(rc, "gsrast_sh_grad_combine_union")
        return whole if whole is not None else (dc, rest)
    if idx is not None:                  # (an M the union kernel does not take: the row map form)
        row_of = torch.full((P,), -1, dtype=torch.int32, device=dev)
        row_of[idx] = torch.arange(idx.numel(), dtype=torch.int32, device=dev)
        rows = int(idx.numel())
    arena.sh_rows_known = False
    with _on_device(dev):
        rc = L.gsrast_sh_grad_combine_rows(P, int(arena.last_degree), M, int(n_views), means3D.data_ptr(), chunks.data_ptr(),
                                           int(chunk_stride if chunk_stride is not None else arena.chunk), int(P if rows is None else rows),
                                           None if row_of is None else row_of.data_ptr(), float(scale), _ptr(whole), _ptr(dc), _ptr(rest),
                                           _stream_of(dev))
    if rc != 0:
        raise _err(rc, "gsrast_sh_grad_combine_rows")
    return whole if whole is not None else (dc, rest)


def rows_pack(idx: torch.Tensor, arrays, packed: torch.Tensor, unpack: bool = False) -> torch.Tensor:
    """Rows `idx` (int64) of the [P, w_k] float32 arrays side by side in packed [n, sum w_k] (include/gsrast.h: gsrast_rows_pack), or --
    unpack=True -- back into those rows.  One launch either way."""
    n, k = int(idx.numel()), len(arrays)
    widths = [int(a.shape[1]) for a in arrays]
    if packed.shape != (n, sum(widths)) or not packed.is_contiguous() or any(not a.is_contiguous() for a in arrays):
        raise ValueError("rows_pack: packed must be a contiguous [n, sum of widths] array, the arrays contiguous")
    ptrs = (C.c_void_p * k)(*[a.data_ptr() for a in arrays])
    wid = (C.c_int * k)(*widths)
    dev = packed.device
    with _on_device(dev):
        fn = lib().gsrast_rows_unpack if unpack else lib().gsrast_rows_pack
        rc = fn(n, idx.data_ptr(), k, ptrs, wid, packed.data_ptr(), _stream_of(dev))
    if rc != 0:
        raise _err(rc, "gsrast_rows_pack")
    return packed


# ---- the all-gather gradient exchange (include/gsrast.h: gsrast_grad_rows_*; csrc/gsrast_exchange.h) -----------------------------------
GRAD_ROW_WORDS = 16


def _arena_sh_arrays(arena: "GradArena"):
    P, M = arena.P, arena.M
    if getattr(arena, "raw", False):
        return None, arena.take("features_dc", (P, 1, 3), False), (arena.take("features_rest", (P, M - 1, 3), False) if M > 1 else None)
    return arena.take("sh", (P, M, 3), False), None, None


def _dense_ptrs(arena: "GradArena"):
    segs = arena.dense_segments()
    if [int(sg.shape[1]) for sg in segs] != [3, 1, 3, 4]:
        raise ValueError("the gradient rows hold mean 3 | opacity 1 | scale 3 | rotation 4: the arena's dense segments differ")
    return (C.c_void_p * 4)(*[sg.data_ptr() for sg in segs])


def grad_rows_pack(arena: "GradArena", touched: torch.Tensor, rows: torch.Tensor) -> None:
    """This rank's touched gradient rows into rows[1:] (int32 [1 + cap, 16]; rows[0, 0], zeroed by the caller, counts them)."""
    dev = rows.device
    with _on_device(dev):
        rc = lib().gsrast_grad_rows_pack(arena.P, touched.data_ptr(), _dense_ptrs(arena), arena.factor.data_ptr(), rows.data_ptr(),
                                         int(rows.shape[0]) - 1, _stream_of(dev))
    if rc != 0:
        raise _err(rc, "gsrast_grad_rows_pack")


def grad_rows_clear(arena: "GradArena", chunks: torch.Tensor, dense: bool, sh: bool) -> None:
    """Zero the rows the chunks (int32 [n, 1 + cap, 16]) name: of the dense arrays and / or of the SH region(s)."""
    n, cap = int(chunks.shape[0]), int(chunks.shape[1]) - 1
    whole, dc, rest = _arena_sh_arrays(arena) if sh else (None, None, None)
    dev = chunks.device
    with _on_device(dev):
        rc = lib().gsrast_grad_rows_clear(arena.P, chunks.data_ptr(), n, (1 + cap) * GRAD_ROW_WORDS, cap, _dense_ptrs(arena) if dense else None, arena.M,
                                          _ptr(whole), _ptr(dc), _ptr(rest), _stream_of(dev))
    if rc != 0:
        raise _err(rc, "gsrast_grad_rows_clear")


def grad_rows_add(arena: "GradArena", chunk: torch.Tensor, means3D: torch.Tensor, scale: float) -> None:
    """One rank's chunk (int32 [1 + cap, 16]) added into the arena: the dense rows and, recombined from the factor, dL/dsh."""
    whole, dc, rest = _arena_sh_arrays(arena)
    dev = chunk.device
    with _on_device(dev):
        rc = lib().gsrast_grad_rows_add(arena.P, chunk.data_ptr(), int(chunk.shape[0]) - 1, _dense_ptrs(arena), int(arena.last_degree), arena.M,
                                        means3D.data_ptr(), float(scale), _ptr(whole), _ptr(dc), _ptr(rest),
                                        _stream_of(dev))
    if rc != 0:
        raise _err(rc, "gsrast_grad_rows_add")


def mark_visible(means3D, viewmatrix, projmatrix) -> torch.Tensor:
    """Mirrors markVisible (rasterize_points.cu:196-215): bool[P], view-space z > 0.2."""
    dev = _require_gpu(means3D)
    P = int(means3D.shape[0])
    present = torch.zeros((P,), dtype=torch.bool, device=dev)
    if P != 0:
        means3D, viewmatrix, projmatrix = (_dev_f32(t, n, dev) for t, n in
                                           ((means3D, "means3D"), (viewmatrix, "viewmatrix"), (projmatrix, "projmatrix")))
        with _on_device(dev):
            rc = lib().gsrast_mark_visible(P, means3D.data_ptr(), viewmatrix.data_ptr(), projmatrix.data_ptr(),
                                           present.data_ptr(), _stream_of(dev))
        if rc != 0:
            raise _err(rc, "gsrast_mark_visible")
    return present


# ---- not part of the reference's _C: options, profiling and the parity-test state export --------
def set_option(name: str, value: int) -> None:
    """Per-call options (PER_CALL_OPTIONS; "pixels_per_lane" sets both directions) are set for the CALLING THREAD and passed
    with each of its calls; the process-wide diagnostics ("profile", "debug_sync", "hexplane_scatter", ...) go to the library."""
    value = int(value)
    names = ("fwd_pixels_per_lane", "bwd_pixels_per_lane") if name == "pixels_per_lane" else (name,)
    if all(n in _OPTION_DEFAULTS for n in names):
        for n in names:
            if n in _OPTION_RANGE:
                if value not in _OPTION_RANGE[n]:
                    raise ValueError(f"gsrast: unknown option or bad value: {name}={value}")
                _thread_options()[n] = value
            else:
                _thread_options()[n] = 1 if value else 0
        _tls.options_version = next(_options_epoch)          # (a new, process-wide unique identity for the cached option structs)
        return
    if lib().gsrast_set_option(name.encode(), value) != 0:
        raise ValueError(f"gsrast: unknown option or bad value: {name}={value}")


def get_option(name: str) -> int:
    if name == "pixels_per_lane":
        name = "fwd_pixels_per_lane"
    if name in _OPTION_DEFAULTS:
        return int(_thread_options()[name])
    return int(lib().gsrast_get_option(name.encode()))


class Context:
    """A caller-owned gsrast_context (include/gsrast.h: gsrast_context_create / _destroy): the state a sequence of similar views shares --
    capacity hints of the speculative launch, the depth sort's history, the pose table with its launch-order hints and cut depths, the
    side streams.  By default every host thread has one of its own; a caller that keeps SEVERAL views in flight on several streams of one
    thread (view_parallel.distributed_step(views_in_flight=n)) gives each lane its own, so that the lanes' forwards do not share side
    streams, gate words and adaptive state:

        ctx = _C.Context()
        with ctx:                      # forwards issued by this thread inside the block use ctx
            color, radii, depth = GaussianRasterizer(settings)(...)
        ctx.close()                    # frees its device memory (or let the object die)

    Results never depend on which context a call runs in."""

    def __init__(self):
        self._h = lib().gsrast_context_create()
        if not self._h:
            raise MemoryError("gsrast_context_create failed")

    @property
    def handle(self):
        if not self._h:
            raise RuntimeError("gsrast: the context has been closed")
        return self._h

    def query(self, name: str) -> int:
        v = int(lib().gsrast_context_query(self.handle, name.encode()))
        if v < 0:
            raise ValueError(f"gsrast: unknown context query: {name}")
        return v

    def policy_event(self, what: str, a: int = 0, b: int = 0, c: int = 0) -> int:
        """gsrast_policy_event (include/gsrast.h): one host-side decision of the context, no device involved."""
        v = int(lib().gsrast_policy_event(self.handle, what.encode(), int(a), int(b), int(c)))
        if v < 0:
            raise ValueError(f"gsrast: unknown policy event: {what}")
        return v

    def close(self) -> None:
        h, self._h = self._h, None
        if h:
            lib().gsrast_context_destroy(h)     # (waits for the context's own streams)

    def __enter__(self):
        stack = getattr(_tls, "contexts", None)
        if stack is None:
            stack = _tls.contexts = []
        stack.append(self)
        return self

    def __exit__(self, *exc):
        _tls.contexts.pop()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:       # noqa: BLE001 -- interpreter shutdown
            pass


def policy_event(what: str, a: int = 0, b: int = 0, c: int = 0) -> int:
    """gsrast_policy_event on the calling thread's current context."""
    v = int(lib().gsrast_policy_event(_current_context(), what.encode(), int(a), int(b), int(c)))
    if v < 0:
        raise ValueError(f"gsrast: unknown policy event: {what}")
    return v


def _current_context():
    """The handle of the innermost `with Context():` block of the calling thread, or None = the thread's own default context."""
    stack = getattr(_tls, "contexts", None)
    return stack[-1].handle if stack else None


def context_query(name: str) -> int:
    """gsrast_context_query on the calling thread's current context (include/gsrast.h): "last_instances", "last_runs", "redo_count",
    "bucket_skip", "last_late", "last_early_runs", "cut_pause", "cut_fallbacks", "completion_passes", "cut_margin_x4", "tau_req", ..."""
    v = int(lib().gsrast_context_query(_current_context(), name.encode()))
    if v < 0:
        raise ValueError(f"gsrast: unknown context query: {name}")
    return v


def profile_read() -> dict:
    """{kernel name: (total_ms, launches)} gathered while option 'profile' is 1."""
    L = lib()
    L.gsrast_profile_collect()
    out = {}
    for k in range(L.gsrast_profile_kernel_count()):
        ms, n = C.c_double(0), C.c_longlong(0)
        L.gsrast_profile_read(k, C.byref(ms), C.byref(n))
        out[L.gsrast_profile_kernel_name(k).decode()] = (ms.value, n.value)
    return out


def profile_reset() -> None:
    lib().gsrast_profile_reset()


def debug_export(P: int, R: int, W: int, H: int, geomBuffer, binningBuffer, imageBuffer) -> dict:
    """Copy the opaque state out in the reference's array layout (parity tests only)."""
    dev = geomBuffer.device
    T = ((W + 15) // 16) * ((H + 15) // 16)
    f32 = dict(dtype=torch.float32, device=dev)
    out = dict(
        depths=torch.zeros(P, **f32), means2D=torch.zeros((P, 2), **f32), cov3D=torch.zeros((P, 6), **f32),
        conic_opacity=torch.zeros((P, 4), **f32), rgb=torch.zeros((P, 3), **f32),
        clamped=torch.zeros((P, 3), dtype=torch.uint8, device=dev),
        tiles_touched=torch.zeros(P, dtype=torch.int32, device=dev),
        keys_sorted=torch.zeros(max(R, 1), dtype=torch.int64, device=dev)[:R],
        point_list=torch.zeros(max(R, 1), dtype=torch.int32, device=dev)[:R],
        ranges=torch.zeros((T, 2), dtype=torch.int32, device=dev),
        final_T=torch.zeros((H, W), **f32), n_contrib=torch.zeros((H, W), dtype=torch.int32, device=dev))
    # cov3D is kept by the forward only under the process-wide option "debug_state" (24 B / Gaussian nothing else reads)
    keep_cov = int(lib().gsrast_get_option(b"debug_state")) != 0
    if not keep_cov:
        del out["cov3D"]
    with _on_device(dev):
        rc = lib().gsrast_debug_export(
            P, R, W, H, _ptr(geomBuffer), _ptr(binningBuffer), _ptr(imageBuffer), out["depths"].data_ptr(),
            out["means2D"].data_ptr(), out["cov3D"].data_ptr() if keep_cov else None, out["conic_opacity"].data_ptr(),
            out["rgb"].data_ptr(), out["clamped"].data_ptr(), out["tiles_touched"].data_ptr(),
            _ptr(out["keys_sorted"]), _ptr(out["point_list"]), out["ranges"].data_ptr(),
            out["final_T"].data_ptr(), out["n_contrib"].data_ptr(), _stream_of(dev))
    if rc != 0:
        raise _err(rc, "gsrast_debug_export")
    return out
