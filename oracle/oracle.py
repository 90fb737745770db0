"""ctypes front-end of the CPU oracle (oracle/gsrast_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg; never by the product package under saro-gs_amd/.  See the C file's header for what the
oracle restates and for its pinning status (no running reference here; SH colour, camera conventions, point projection
and cov3D pinned against the reference's own Python, everything else -- including every backward formula -- against an
independent torch-autograd derivation, tests/test_oracle_independent.py).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libgsrast_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile).  Safe to call repeatedly."""
    src = os.path.join(_HERE, "gsrast_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        for pfx in ("orc32_", "orc64_"):
            getattr(_lib, pfx + "bin_count").restype = C.c_int64
            getattr(_lib, pfx + "expf").restype = C.c_float
            getattr(_lib, pfx + "expf").argtypes = [C.c_float]
            getattr(_lib, pfx + "higher_msb").restype = C.c_uint32
            getattr(_lib, pfx + "higher_msb").argtypes = [C.c_uint32]
    return _lib


def _p(a: Optional[np.ndarray]):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"], "oracle arrays must be contiguous"
    return a.ctypes.data_as(C.c_void_p)


def _f32(a) -> Optional[np.ndarray]:
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def set_exp_mode(mode: int) -> None:
    """0 = fixed-sequence fp32 exp (bit-reproducible on the GPU), 1 = libm expf."""
    lib().orc32_set_exp_mode(C.c_int(mode))
    lib().orc64_set_exp_mode(C.c_int(mode))


def expf(x: float) -> float:
    return float(lib().orc32_expf(C.c_float(x)))


def higher_msb(n: int) -> int:
    return int(lib().orc32_higher_msb(C.c_uint32(n)))


def sh_to_rgb(deg: int, pos: np.ndarray, campos: np.ndarray, sh: np.ndarray, f64: bool = False) -> np.ndarray:
    """Unclamped colour (SH + 0.5) for n points; sh is [n, M, 3]."""
    pos, campos, sh = _f32(pos), _f32(campos), _f32(sh)
    n, M = sh.shape[0], sh.shape[1]
    out = np.zeros((n, 3), dtype=np.float64 if f64 else np.float32)
    fn = lib().orc64_sh_to_rgb if f64 else lib().orc32_sh_to_rgb
    fn(C.c_int(n), C.c_int(deg), C.c_int(M), _p(pos), _p(campos), _p(sh), _p(out))
    return out


def mark_visible(means3D, viewmatrix, projmatrix) -> np.ndarray:
    means3D = _f32(means3D)
    out = np.zeros(means3D.shape[0], dtype=np.uint8)
    lib().orc32_mark_visible(C.c_int(means3D.shape[0]), _p(means3D), _p(_f32(viewmatrix)),
                             _p(_f32(projmatrix)), _p(out))
    return out.astype(bool)


def forward(scene: Dict, cam: Dict, *, colors_precomp=None, cov3D_precomp=None,
            st32: Optional[Dict] = None) -> Dict:
    """Full forward.  With st32=None runs the fp32 build (returns every intermediate array).
    With st32 = the fp32 result, runs the fp64 "truth" build replaying st32's discrete decisions."""
    f64 = st32 is not None
    L = lib()
    pfx = "orc64_" if f64 else "orc32_"
    rdt = np.float64 if f64 else np.float32
    means3D = _f32(scene["means3D"])
    P = means3D.shape[0]
    W, H = int(cam["image_width"]), int(cam["image_height"])
    shs = _f32(scene.get("shs")) if colors_precomp is None else None
    M = 0 if shs is None else shs.shape[1]
    D = int(scene.get("sh_degree", 0))
    scales = _f32(scene.get("scales")) if cov3D_precomp is None else None
    rots = _f32(scene.get("rotations")) if cov3D_precomp is None else None
    opac = _f32(scene["opacities"]).reshape(-1)
    colors_precomp = _f32(colors_precomp)
    cov3D_precomp = _f32(cov3D_precomp)
    view, proj, campos = _f32(cam["viewmatrix"]), _f32(cam["projmatrix"]), _f32(cam["campos"])
    bg = _f32(scene.get("bg", np.zeros(3)))
    T = ((W + 15) // 16) * ((H + 15) // 16)

    st = dict(P=P, W=W, H=H, D=D, M=M)
    st["radii"] = np.zeros(P, np.int32)
    st["means2D"] = np.zeros((P, 2), rdt)
    st["depths"] = np.zeros(P, rdt)
    st["cov3D"] = np.zeros((P, 6), rdt)
    st["rgb"] = np.zeros((P, 3), rdt)
    st["conic_opacity"] = np.zeros((P, 4), rdt)
    st["clamped"] = np.zeros((P, 3), np.uint8)
    st["tiles_touched"] = np.zeros(P, np.uint32)
    getattr(L, pfx + "preprocess")(
        C.c_int(P), C.c_int(D), C.c_int(M), _p(means3D), _p(scales), C.c_float(cam.get("scale_modifier", 1.0)),
        _p(rots), _p(opac), _p(shs), _p(cov3D_precomp), _p(colors_precomp), _p(view), _p(proj), _p(campos),
        C.c_int(W), C.c_int(H), C.c_float(cam["tanfovx"]), C.c_float(cam["tanfovy"]),
        _p(st32["radii"]) if f64 else None, _p(st32["clamped"]) if f64 else None,
        _p(st["radii"]), _p(st["means2D"]), _p(st["depths"]), _p(st["cov3D"]), _p(st["rgb"]),
        _p(st["conic_opacity"]), _p(st["clamped"]), _p(st["tiles_touched"]))
    if cov3D_precomp is not None:
        st["cov3D"] = cov3D_precomp.astype(rdt)
    colors = st["rgb"] if colors_precomp is None else np.ascontiguousarray(colors_precomp.astype(rdt))
    st["colors"] = colors

    if f64:
        for k in ("point_offsets", "R", "keys_unsorted", "keys_sorted", "point_list", "ranges"):
            st[k] = st32[k]
        m32, c32 = st32["means2D"], st32["conic_opacity"]
    else:
        st["point_offsets"] = np.zeros(P, np.uint32)
        R = int(L.orc32_bin_count(C.c_int(P), _p(st["tiles_touched"]), _p(st["point_offsets"])))
        st["R"] = R
        st["keys_unsorted"] = np.zeros(max(R, 1), np.uint64)[:R]
        st["keys_sorted"] = np.zeros(max(R, 1), np.uint64)[:R]
        st["point_list"] = np.zeros(max(R, 1), np.uint32)[:R]
        st["ranges"] = np.zeros((T, 2), np.uint32)
        L.orc32_bin_fill(C.c_int(P), C.c_int(W), C.c_int(H), _p(st["means2D"]), _p(st["depths"]),
                         _p(st["radii"]), _p(st["point_offsets"]), C.c_int64(R),
                         _p(np.ascontiguousarray(st["keys_unsorted"])) if R else None,
                         _p(st["keys_sorted"]) if R else None, _p(st["point_list"]) if R else None,
                         _p(st["ranges"]))
        m32, c32 = st["means2D"], st["conic_opacity"]

    st["out_color"] = np.zeros((3, H, W), rdt)
    st["out_depth"] = np.zeros((1, H, W), rdt)
    st["final_T"] = np.zeros((H, W), rdt)
    st["n_contrib"] = np.zeros((H, W), np.uint32)
    st["pair_hash"] = np.zeros((H, W), np.uint32)   # which list entries passed every test, per pixel
    if P > 0:
        getattr(L, pfx + "blend_forward")(
            C.c_int(W), C.c_int(H), _p(st["ranges"]), _p(st["point_list"]) if st["R"] else None,
            _p(st["means2D"]), _p(colors), _p(st["conic_opacity"]), _p(st["depths"]),
            _p(m32), _p(c32), _p(bg), _p(st["out_color"]), _p(st["out_depth"]), _p(st["final_T"]),
            _p(st["n_contrib"]), _p(st["pair_hash"]))
    st["_ctl_means2D"], st["_ctl_conic"] = m32, c32
    return st


def backward(scene: Dict, cam: Dict, st: Dict, dL_dcolor: np.ndarray, *, colors_precomp=None,
             cov3D_precomp=None) -> Dict:
    """Backward for the forward state st (fp32 or fp64 build, decided by st's dtype)."""
    f64 = st["means2D"].dtype == np.float64
    L = lib()
    pfx = "orc64_" if f64 else "orc32_"
    rdt = np.float64 if f64 else np.float32
    P, W, H, D, M = st["P"], st["W"], st["H"], st["D"], st["M"]
    means3D = _f32(scene["means3D"])
    shs = _f32(scene.get("shs")) if colors_precomp is None else None
    scales = _f32(scene.get("scales")) if cov3D_precomp is None else None
    rots = _f32(scene.get("rotations")) if cov3D_precomp is None else None
    view, proj, campos = _f32(cam["viewmatrix"]), _f32(cam["projmatrix"]), _f32(cam["campos"])
    bg = _f32(scene.get("bg", np.zeros(3)))
    dpix = _f32(dL_dcolor)
    g = dict(
        dL_dmeans2D=np.zeros((P, 3), rdt), dL_dconic=np.zeros((P, 4), rdt), dL_dopacity=np.zeros((P, 1), rdt),
        dL_dcolors=np.zeros((P, 3), rdt), dL_dmeans3D=np.zeros((P, 3), rdt), dL_dcov3D=np.zeros((P, 6), rdt),
        dL_dsh=np.zeros((P, M, 3), rdt), dL_dscales=np.zeros((P, 3), rdt), dL_drotations=np.zeros((P, 4), rdt))
    if P == 0:
        return g
    getattr(L, pfx + "blend_backward")(
        C.c_int(P), C.c_int(W), C.c_int(H), C.c_int64(st["R"]), _p(st["ranges"]),
        _p(st["point_list"]) if st["R"] else None, _p(bg), _p(st["means2D"]), _p(st["conic_opacity"]),
        _p(st["colors"]), _p(st["_ctl_means2D"]), _p(st["_ctl_conic"]), _p(st["final_T"]), _p(st["n_contrib"]),
        _p(dpix), _p(g["dL_dmeans2D"]), _p(g["dL_dconic"]), _p(g["dL_dopacity"]), _p(g["dL_dcolors"]))
    getattr(L, pfx + "preprocess_backward")(
        C.c_int(P), C.c_int(D), C.c_int(M), _p(means3D), _p(st["radii"]), _p(shs), _p(st["clamped"]),
        _p(scales), _p(rots), C.c_float(cam.get("scale_modifier", 1.0)), _p(np.ascontiguousarray(st["cov3D"])),
        _p(view), _p(proj), C.c_int(W), C.c_int(H), C.c_float(cam["tanfovx"]), C.c_float(cam["tanfovy"]),
        _p(campos), _p(g["dL_dmeans2D"]), _p(g["dL_dconic"]), _p(g["dL_dcolors"]), _p(g["dL_dmeans3D"]),
        _p(g["dL_dcov3D"]), _p(g["dL_dsh"]), _p(g["dL_dscales"]), _p(g["dL_drotations"]))
    return g


def render(scene: Dict, cam: Dict, dL_dcolor: Optional[np.ndarray] = None, *, f64: bool = False,
           colors_precomp=None, cov3D_precomp=None) -> Dict:
    """Convenience: forward (+ backward when dL_dcolor is given).  f64=True returns the truth build
    (control flow replayed from the fp32 build)."""
    st = forward(scene, cam, colors_precomp=colors_precomp, cov3D_precomp=cov3D_precomp)
    if f64:
        st = forward(scene, cam, colors_precomp=colors_precomp, cov3D_precomp=cov3D_precomp, st32=st)
    out = dict(st)
    if dL_dcolor is not None:
        out.update(backward(scene, cam, st, dL_dcolor, colors_precomp=colors_precomp, cov3D_precomp=cov3D_precomp))
    return out
