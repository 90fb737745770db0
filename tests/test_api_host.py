"""CPU: the host-side mirror of the reference's Python interface (no GPU needed)."""
import inspect

import pytest
import torch


def test_settings_fields_and_order(rast):
    """GaussianRasterizationSettings: the reference's eleven fields, in the reference's order
    (diff_gaussian_rasterization_ch3/__init__.py:134-145), so positional construction keeps working."""
    assert rast.GaussianRasterizationSettings._fields == (
        "image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix",
        "sh_degree", "campos", "prefiltered")


def test_forward_signature_matches_reference(rast):
    sig = inspect.signature(rast.GaussianRasterizer.forward)
    assert list(sig.parameters) == ["self", "means3D", "means2D", "opacities", "shs", "colors_precomp", "scales",
                                    "rotations", "cov3D_precomp"]
    for k in ("shs", "colors_precomp", "scales", "rotations", "cov3D_precomp"):
        assert sig.parameters[k].default is None
    assert hasattr(rast.GaussianRasterizer, "markVisible")
    assert issubclass(rast.GaussianRasterizer, torch.nn.Module)


def _settings(rast):
    z = torch.zeros
    return rast.GaussianRasterizationSettings(32, 32, 0.5, 0.5, z(3), 1.0, torch.eye(4), torch.eye(4), 3, z(3), False)


def test_exactly_one_of_checks(rast):
    """Same two argument-combination errors as the reference (__init__.py:167-171)."""
    r = rast.GaussianRasterizer(_settings(rast))
    P = 4
    m3, m2, op = torch.zeros(P, 3), torch.zeros(P, 3), torch.zeros(P, 1)
    sh, col, sc, rot, cov = torch.zeros(P, 16, 3), torch.zeros(P, 3), torch.ones(P, 3), torch.zeros(P, 4), torch.zeros(P, 6)
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        r(m3, m2, op, scales=sc, rotations=rot)
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        r(m3, m2, op, shs=sh, colors_precomp=col, scales=sc, rotations=rot)
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(m3, m2, op, shs=sh)
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(m3, m2, op, shs=sh, scales=sc)                      # rotations missing
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(m3, m2, op, shs=sh, scales=sc, rotations=rot, cov3D_precomp=cov)


def test_no_cpu_fallback(rast):
    """CPU tensors must fail loudly: the product path is the HIP library, nothing else."""
    r = rast.GaussianRasterizer(_settings(rast))
    P = 4
    with pytest.raises(RuntimeError, match="GPU"):
        r(torch.zeros(P, 3), torch.zeros(P, 3), torch.zeros(P, 1), shs=torch.zeros(P, 16, 3), scales=torch.ones(P, 3),
          rotations=torch.zeros(P, 4))
    with pytest.raises(RuntimeError, match="num_points, 3"):
        rast._C.rasterize_gaussians(torch.zeros(3), torch.zeros(5, 2), torch.empty(0), torch.zeros(5, 1), torch.empty(0),
                                    torch.empty(0), 1.0, torch.empty(0), torch.eye(4), torch.eye(4), 0.5, 0.5, 8, 8,
                                    torch.empty(0), 0, torch.zeros(3), False)


def test_product_does_not_import_the_oracle():
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "saro-gs_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(d, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle|#include\s+[\"<].*oracle", text, flags=re.M), f


def test_state_arena_is_freed_by_refcounting_not_by_the_cyclic_gc():
    """The allocation callbacks are closures over the arena: close() must break that cycle, otherwise the three state
    buffers (hundreds of MB on the GPU) live until the cyclic collector happens to run."""
    import gc
    import weakref
    import torch
    from diff_gaussian_rasterization_ch3 import _C
    gc.disable()
    try:
        arena = _C._Arena(torch.device("cpu"))
        ptr = arena.callbacks[1](None, 1024)                 # what libgsrast does: ask for 1 KiB of binning state
        assert ptr and arena.tensor(1).numel() == 1024
        buf_ref, arena_ref = weakref.ref(arena.tensor(1)), weakref.ref(arena)
        arena.close()
        del arena
        assert arena_ref() is None and buf_ref() is None     # gone without gc.collect()
    finally:
        gc.enable()


def test_bench_refuses_a_world_size_other_than_gpus():
    """bench.py never prints a line whose n_gpus differs from --gpus: a launcher that started another number of ranks is refused
    with a non-zero exit code (checked before anything touches a GPU, so this runs on the CPU box)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 2 and "refusing" in out.stderr and '{"metric"' not in out.stdout
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 2 and '{"metric"' not in out.stdout
