"""CPU: the C-ABI shared library loads without a GPU, exports every symbol include/gsrast.h declares,
and rejects bad arguments before touching a device (no compute calls here)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "gsrast.h")


@pytest.fixture(scope="module")
def L(rast):
    return rast._C.lib()


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gsrast_[a-z_0-9]+)\s*\(", text)) - {"gsrast_alloc_fn"})


def test_header_and_library_agree(rast, L):
    names = declared_functions()
    assert len(names) >= 16
    raw = C.CDLL(rast._C.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"{n} declared in gsrast.h but not exported"
    assert sorted(rast._C.EXPORTS) == names
    assert L.gsrast_abi_version() == rast._C.ABI_VERSION == 5
    assert re.search(r"#define GSRAST_ABI_VERSION 5\b", open(HEADER).read())


def test_no_torch_or_cxx_types_in_the_boundary():
    text = open(HEADER).read()
    assert 'extern "C"' in text
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)          # comments cite the reference's C++ types
    for banned in ("torch", "at::", "std::", "Tensor", "#include <vector>", "#include <functional>"):
        assert banned not in text


def test_state_buffer_sizes(L):
    g = [L.gsrast_geometry_bytes(p) for p in (0, 1, 1000, 100000, 3000000)]
    assert all(b >= a for a, b in zip(g, g[1:])) and g[0] > 0 and g[2] > g[1]
    assert g[-1] / 3000000 < 345          # ~145 B per Gaussian of forward state + the backward's 64-byte gradient record and 36 B of
                                          # colour / view-direction derivatives + the bucket depth sort's slabs (32-64 B per Gaussian)
                                          # + the list cut's compact early set (2 x 4 B per bucket slot: 22 B per Gaussian at 3 M) and flag byte
    b = [L.gsrast_binning_bytes(r, 1920, 1080) for r in (0, 10, 10**6, 5 * 10**7)]
    assert all(y >= x for x, y in zip(b, b[1:])) and b[2] > b[1]
    assert b[-1] / (5 * 10**7) < 20       # 16 B per instance + histograms
    i = L.gsrast_image_bytes(1920, 1080)
    assert 8 * 1920 * 1080 <= i <= 12 * 1920 * 1080     # 8 B per pixel + per-tile arrays (ranges, work-bucket lists; round 5: the predicted cut's
                                                         # opacity-mass table, 8 copies x 16 depth bins x 4 B per tile = 2 B per pixel)
    assert all(x % 256 == 0 for x in g + b + [i])


def test_options_round_trip(L, rast):
    for mode in (0, 1, 2, 0):
        rast._C.set_option("exp_mode", mode)
        assert rast._C.get_option("exp_mode") == mode
    with pytest.raises(ValueError):
        rast._C.set_option("exp_mode", 7)
    with pytest.raises(ValueError):
        rast._C.set_option("no_such_option", 1)
    n = L.gsrast_profile_kernel_count()
    names = [L.gsrast_profile_kernel_name(k).decode() for k in range(n)]
    assert "blend_fwd" in names and "blend_bwd" in names and "preprocess_fwd" in names


def test_bad_arguments_fail_before_any_device_work(L, rast):
    ALLOC = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)
    cb = ALLOC(lambda ctx, n: None)
    # negative P / zero-size image / SH degree out of range / NULL allocators
    base = dict(P=10, D=3, M=16, W=64, H=64)
    for bad in (dict(P=-1), dict(W=0), dict(D=4)):
        a = dict(base, **bad)
        rc = L.gsrast_forward(cb, None, cb, None, cb, None, a["P"], a["D"], a["M"], None, a["W"], a["H"], None, None,
                              None, None, None, 1.0, None, None, None, None, None, 0.5, 0.5, 0, None, None, None, None)
        assert rc == -1 and L.gsrast_last_error()
    rc = L.gsrast_backward(-3, 3, 16, 0, None, 64, 64, None, None, None, None, 1.0, None, None, None, None, None, 0.5, 0.5,
                           None, None, None, None, None, None, None, None, None, None, None, None, None, None, None)
    assert rc == -1
    assert L.gsrast_mark_visible(-1, None, None, None, None, None) == -1


def test_widened_rows_reject_bad_arguments_without_a_device(rast, L):
    """hexplane lookup / Linear weight gradient: argument checks run before any HIP call (no GPU here)."""
    PS = rast._C.PlaneStruct
    ok = (PS * 1)(PS(None, None, 64, 64, 0, 1, 7, 0))
    n = L.gsrast_hexplane_scratch_bytes(1, ok, 32, 1000)
    assert n > 2 * 1365 * 32 * 4                               # value + gradient stacks of levels >= 1 (1365 texels), + the sorted pairs
    assert L.gsrast_hexplane_scratch_bytes(1, ok, 32, 2000) > n
    odd = (PS * 1)(PS(None, None, 12, 10, 0, 1, 7, 0))         # 6 x 5 cannot be halved
    assert L.gsrast_hexplane_scratch_bytes(1, odd, 32, 1000) == 0
    assert L.gsrast_hexplane_scratch_bytes(1, ok, 12, 1000) == 0   # channels: power of two in [4, 64]
    assert L.gsrast_hexplane_scratch_bytes(0, ok, 32, 1000) == 0
    assert L.gsrast_hexplane_forward(10, 4, 32, 32, 1, odd, None, None, None, None, None) == -1
    assert b"odd extent" in L.gsrast_last_error()
    far = (PS * 1)(PS(None, None, 64, 64, 0, 5, 7, 0))         # coordinate column outside a 4-float point row
    assert L.gsrast_hexplane_forward(10, 4, 32, 32, 1, far, None, None, None, None, None) == -1
    off = (PS * 1)(PS(None, None, 64, 64, 0, 1, 7, 16))        # feature block [16, 48) outside a 32-float row
    assert L.gsrast_hexplane_backward(10, 4, 32, 32, 1, off, None, None, None, None, None, 0, None, None) == -1


def test_raw_entry_points_reject_bad_arguments_before_any_device_work(L, rast):
    """gsrast_forward_raw / gsrast_backward_raw validate their pointer sets on the host (no GPU needed for the refusals)."""
    _C = rast._C
    ALLOC = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)
    cb = ALLOC(lambda ctx, n: None)
    one = C.c_void_p(16)
    opts = _C.OptionsStruct()
    L.gsrast_options_init(C.byref(opts))
    assert opts.forward_only == 0 and opts.tile_clip == 1
    ins = _C.RawInputsStruct()
    rc = L.gsrast_forward_raw(None, C.byref(opts), cb, None, cb, None, cb, None, 10, 3, 16, one, 64, 64, C.byref(ins), 1.0, one, one, one, 1.0, 1.0, one, one, one, None)
    assert rc < 0 and b"raw" in L.gsrast_last_error()
    ok = dict(xyz=16, rotation=16, scaling=16, opacity_logit=16, features_dc=16, features_rest=16)
    ins = _C.RawInputsStruct(**ok)
    rc = L.gsrast_forward_raw(None, C.byref(opts), cb, None, cb, None, cb, None, 10, 3, 9, one, 64, 64, C.byref(ins), 1.0, one, one, one, 1.0, 1.0, one, one, one, None)
    assert rc < 0 and b"M must be" in L.gsrast_last_error()
    ins = _C.RawInputsStruct(**dict(ok, features_rest=20))
    rc = L.gsrast_forward_raw(None, C.byref(opts), cb, None, cb, None, cb, None, 10, 3, 16, one, 64, 64, C.byref(ins), 1.0, one, one, one, 1.0, 1.0, one, one, one, None)
    assert rc < 0 and b"aligned" in L.gsrast_last_error()
    ins = _C.RawInputsStruct(**ok)
    gr = _C.RawGradsStruct()
    rc = L.gsrast_backward_raw(C.byref(opts), 10, 3, 16, 5, one, 64, 64, C.byref(ins), 1.0, one, one, one, 1.0, 1.0, one, one, one, one, one, C.byref(gr), None)
    assert rc < 0 and b"NULL required gradient" in L.gsrast_last_error()
    gr = _C.RawGradsStruct(dL_dmean2D=16, d_xyz=16, d_rotation=16, d_scaling=16, d_opacity_logit=16, d_shs_res=16)
    rc = L.gsrast_backward_raw(C.byref(opts), 10, 3, 16, 5, one, 64, 64, C.byref(ins), 1.0, one, one, one, 1.0, 1.0, one, one, one, one, one, C.byref(gr), None)
    assert rc < 0 and b"d_shs_res" in L.gsrast_last_error()


def test_prealloc_callback_hands_out_what_fits_and_nothing_else():
    """gsrast_alloc_prealloc (include/gsrast.h, round 6): the library's own allocation callback over memory the caller already holds --
    pure host code, callable without a GPU.  Returns the pointer when the request fits, NULL otherwise, and records the request."""
    import ctypes as C
    from diff_gaussian_rasterization_ch3 import _C
    L = _C.lib()
    L.gsrast_alloc_prealloc.restype = C.c_void_p
    L.gsrast_alloc_prealloc.argtypes = [C.c_void_p, C.c_size_t]
    p = _C.PreallocStruct()
    p.ptr, p.capacity = 0x1000, 4096
    assert L.gsrast_alloc_prealloc(C.addressof(p), 4096) == 0x1000 and p.requested == 4096
    assert L.gsrast_alloc_prealloc(C.addressof(p), 100) == 0x1000 and p.requested == 100
    assert L.gsrast_alloc_prealloc(C.addressof(p), 4097) is None and p.requested == 4097
    assert L.gsrast_alloc_prealloc(None, 16) is None
    assert _C._PREALLOC_CB is not None
    # the sizes a caller pre-allocates are the ones the forward will ask for: monotone, 256-byte aligned arrays inside
    gb, ib = _C._state_bytes(1000, 640, 480)
    assert gb == L.gsrast_geometry_bytes(1000) and ib == L.gsrast_image_bytes(640, 480) and gb > 0 and ib > 0
