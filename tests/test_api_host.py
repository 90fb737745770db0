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
        arena = _C._Arena.acquire(torch.device("cpu"))
        ptr = arena.callbacks[1](None, 1024)                 # what libgsrast does: ask for 1 KiB of binning state
        assert ptr and arena.tensor(1).numel() == 1024
        buf_ref = weakref.ref(arena.tensor(1))
        arena.close()
        assert buf_ref() is None                             # the buffer is gone without gc.collect() ...
        again = _C._Arena.acquire(torch.device("cpu"))
        assert again is arena and again.buffers == [None, None, None] and again.callbacks is not None   # ... and the arena, holding nothing, serves the next call (round 6)
        again.close()
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


def test_bench_final_line_is_compact_and_carries_the_contract():
    """the LAST stdout line of bench.py must parse.  Round 5's line had grown to 20.7 KB and the driver's parser gave up
    (BENCH_r05.parsed = null).  bench.compact_line() is what main() prints last: compact JSON, at most bench.LINE_LIMIT (4096) bytes, with
    `roofline` and `cpu_baseline` inside -- checked here on a result of the real shape (the roofline objects come from bench.roofline_of
    over the committed counter passes; the numbers are a 3 M run's)."""
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    st = dict(N=1920 * 1080, R_eff=2_462_923, R=42_300_000, R_listed=25_000_000, Q=8_700_000, P_vis=2_990_000, T=8160)
    stage = {k: 0.0123 for k in ("preprocess_fwd", "preprocess_color", "sort_depth", "scan_tiles", "emit_instances", "sort_tile", "tile_ranges", "blend_fwd",
                                 "blend_bwd", "preprocess_bwd", "late_rows_zero", "sh_dir_derivs", "cut_redo", "grec_zero_touched")}
    result = {
        "metric": "rendered views/s (fwd+bwd) at 1080p vs #Gaussians", "value": 801.234, "unit": "views/s", "n_gpus": 1, "steps": 20, "warmup": 5,
        "ms_per_step": 1.2481, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "preroll_steps": 240, "mean_ms_per_step": 1.2611, "value_warm": 861.5, "ms_per_step_warm": 1.1608, "warmup_warm": 24, "value_no_list_cut": 640.2,
        "pipelined": {"views_per_s_cold": 842.1, "views_per_s_warm": 900.3, "note": "K steps back to back, one sync (rounds 1-5's headline): throughput, not the metric"},
        "config": {"workload": "BASELINE configs[4] stress-1080p: synth(P=3000000, seed 0) SH3, 1920x1080, one view per GPU, fwd+bwd",
                   "protocol": "SURVEY 8d: 1 / median of per-call device-synchronised wall-clock fwd+bwd", "pose_table": "off for value (every pose a first visit); on for value_warm",
                   "gaussians": 3_000_000, "width": 1920, "height": 1080, "sh_degree": 3, "exp_mode": 0, "poses_per_rank": 8, "views_per_step": 1,
                   "R": st["R"], "R_listed": st["R_listed"], "Q": st["Q"], "R_eff": st["R_eff"], "visible": st["P_vis"], "late_gaussians_cold": 1_290_000,
                   "late_gaussians_warm": 2_700_000, "word_fork": 0, "exchange": "gather", "exchange_backend": "nccl", "rccl_world": 8, "rccl_version": "2.26.6",
                   "rank_devices": list(range(8)), "exchange_bytes_per_rank_and_step": {"allreduce": 4, "allgather": 9_600_064, "rows": 150_000}},
        "roofline": bench.roofline_of(st, 0.4009, 3_000_000),
        "roofline_fwd": bench.roofline_of(st, 0.2601, 3_000_000, kernel="blend_fwd_cull_kernel"),
        "stage_ms": stage, "host": {"enqueue_ms_median": 0.6123, "device_mallocs_in_timed_steps": 0}, "per_step_ms": {"min": 1.2011, "max": 1.4012},
        "sweep_1080p_cold": {"100000": 1901.2, "300000": 1650.3, "1000000": 1310.4, "3000000": 801.234},
        "cpu_baseline": {"value": 0.2612, "unit": "views/s", "cores": 256, "kind": "port",
                         "sample": "4 view(s) fwd+bwd of the same workload (P=3000000, 1920x1080, SH3), 15.5 s of CPU work"},
        "extras": "gpurun_out/bench_report.json",
    }
    line = bench.compact_line(result)
    assert "\n" not in line and len(line) <= bench.LINE_LIMIT < 8192, len(line)
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert r["algorithmic_bytes_per_launch"] == st["N"] * 20 + st["R_eff"] * 76
    assert d["roofline_fwd"]["algorithmic_bytes_per_launch"] == st["N"] * 24 + st["R_eff"] * 44
    assert r["traffic"] and r["valu_issue_slot_frac"] and r["valu_issue_slot_frac_guide_2cyc_fma"] < r["valu_issue_slot_frac"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in d["cpu_baseline"], k
    assert d["config"]["workload"] and "model" not in d["config"]
    # a result that outgrows the limit loses optional diagnostics, never the contract's keys
    fat = dict(result, stage_ms={f"kernel_{i}": 0.0123 for i in range(400)})
    d2 = json.loads(bench.compact_line(fat))
    assert "stage_ms" not in d2 and "roofline" in d2 and "cpu_baseline" in d2 and len(bench.compact_line(fat)) <= bench.LINE_LIMIT


def test_bench_extras_and_tools_import_without_a_gpu():
    """tools/bench_extras.py (the side legs behind `bench.py --extras`) imports `bench` and must not rot unnoticed on the CPU box: it imports, its entry
    point and the legs exist, and bench's Workload / Deformation / timing helpers it relies on are there."""
    import importlib
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tools")):
        if p not in sys.path:
            sys.path.insert(0, p)
    bench = importlib.import_module("bench")
    extras = importlib.import_module("bench_extras")
    for name in ("run", "training_like_row", "eval_fps_row", "in_flight_row", "measure_point", "loss_row", "epilogue_row", "adam_row", "knn_row", "hexplane_row",
                 "iteration_row", "exp_mode2_row"):
        assert callable(getattr(extras, name)), name
    for name in ("Workload", "Deformation", "timed", "timed_sync", "roofline_of", "stage_table", "step_bytes", "cpu_baseline", "sweep_point", "compact_line", "parse"):
        assert hasattr(bench, name), name
    a = bench.parse(["--gpus", "1", "--steps", "20", "--warmup", "5"])
    assert a.steps == 20 and a.warmup == 5 and a.extras is None and a.exchange == "gather" and a.gaussians == 3_000_000
    assert bench.parse(["--extras"]).extras.endswith("bench_report.json")
