#!/usr/bin/env python
"""bench.py -- rendered views/s (forward + backward) of the Gaussian rasterizer hot path.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, or by this script itself)

A "step" is one GaussianRasterizer forward + backward over one synthetic 1080p view of synth(P = 3e6) (BASELINE.json configs[4]); with
N > 1 every rank renders its own view of the same scene (one view per GPU, "weak" scaling) and the step also contains the RCCL exchange
of the per-Gaussian gradients.  Inputs are resident in HBM before the timed region.

The LAST line rank 0 prints is ONE compact JSON object (< 4 KB; tests/test_api_host.py checks the size and the keys):
  value        SURVEY.md 8(d)'s metric as written: N / median over the K timed steps of the per-call DEVICE-SYNCHRONISED wall-clock time of
               one forward + backward (max over ranks), measured with the context's pose table OFF -- every forward is a first visit of its
               camera pose, the number that transfers to SaRO-GS's time-varying scenes (`config.pose_table` says so)
  value_warm   the same protocol with the pose table on (every pose seen before: a fixed rig over an unchanged scene)
  pipelined    K steps enqueued back to back, one synchronisation at the end (rounds 1-5's headline): throughput, not the stated metric
  roofline     the dominant kernel (per-tile backward blend) against the HBM roof: algorithmic bytes per launch (SURVEY.md 8(d) formula with
               the MEASURED R_eff) / its mean duration measured with HIP events on the launch stream inside the timed region; the binding
               resource (VALU issue) beside it;  roofline_fwd: the same for the forward blend
  cpu_baseline the CPU oracle (a port, OpenMP on the host cores) timed on a bounded sample of the same workload
Everything else rounds 1-5 printed in that line (per-stage tables, training-like / eval legs, the SURVEY 8f rows, a second scene) is
`--extras`: tools/bench_extras.py, written to gpurun_out/bench_report.json or the path given and printed as an EARLIER line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "saro-gs_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s (6.29 TB/s measured copy)
LINE_LIMIT = 4096      # bytes of the final JSON line (the driver's parser gave up on round 5's 20.7 KB line)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--gaussians", type=int, default=3_000_000, help="headline #Gaussians at 1920x1080 (BASELINE.json configs[4]: the 3 M stress)")
    ap.add_argument("--scene", choices=("cube", "shell"), default="cube", help="cube: scenes.synth (SURVEY 8d, the headline); shell: scenes.synth_shell (profiling runs)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--sh-degree", type=int, default=3)
    ap.add_argument("--exchange", choices=("gather", "sparse", "factors", "allreduce"), default="gather",
                    help="multi-GPU gradient exchange: 'gather' (default) = ONE all-gather of the rows each rank's own view touched (64 B per "
                         "row), added in rank order; 'factors' = all-reduce 11 + all-gather 3 floats/Gaussian; 'sparse' = the same for the "
                         "rows some rank touched only; 'allreduce' = all-reduce all 59 floats/Gaussian")
    ap.add_argument("--exp-mode", type=int, default=None, help="0 fixed-sequence (default), 1 ocml, 2 v_exp_f32")
    ap.add_argument("--sweep", type=str, default="100000,300000,1000000",
                    help="extra #Gaussians points of the metric's 'vs #Gaussians' (N=1 only; same protocol, table off); '' disables")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--extras", nargs="?", const=os.path.join(ROOT, "gpurun_out", "bench_report.json"), default=None, metavar="PATH",
                    help="also run the side legs of tools/bench_extras.py (per-stage tables at 1 M, shell scene, BASELINE cfg2 / cfg3 shapes, two views in "
                         "flight, eval FPS, training-like iterations, SURVEY 8f rows): written to PATH and printed as an earlier line")
    ap.add_argument("--ablate", type=int, default=0, help="kernel ablation experiments (not a valid bench)")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE", help="library option for A/B experiments (repeatable)")
    ap.add_argument("--ppl", type=int, default=0, help="force pixels per lane of both blend kernels (0 = auto)")
    ap.add_argument("--ppl-fwd", type=int, default=0)
    ap.add_argument("--ppl-bwd", type=int, default=0)
    ap.add_argument("--no-cull", action="store_true", help="disable wave-level strip culling (A/B experiments)")
    ap.add_argument("--no-lpt", action="store_true", help="disable heaviest-tile-first launch order (A/B experiments)")
    ap.add_argument("--binning", type=int, default=None, help="0 run-compressed binning (default), 1 instance-level two-pass sort")
    ap.add_argument("--cpu-budget-s", type=float, default=25.0)
    ap.add_argument("--force-collectives", action="store_true",
                    help="--gpus 1 only: run the N > 1 gradient exchange through a one-rank RCCL process group (backend nccl: ReduceOp.AVG, "
                         "the uint8 MAX and all_gather_into_tensor of the sparse exchange) instead of skipping it -- puts the collective "
                         "library under the bench on a 1-GPU box; a profiling run, not the headline (sweep / CPU legs off)")
    ap.add_argument("--poses", type=int, default=8,
                    help="camera poses each rank renders round-robin (the reference's batch loop renders different cameras one after "
                         "the other, train.py:198-226); 1 = the repeated-pose protocol of rounds 1-3")
    ap.add_argument("--preroll-ms", type=float, default=300.0,
                    help="untimed steps before the W warm-up steps until this much wall time has passed (reported as `preroll_steps`): the first "
                         "GPU process on a fresh box shows one 5-9 ms device hiccup some tens of ms into sustained load (clock / power "
                         "management settling); 0 disables")
    return ap.parse_args(argv)


class Workload:
    """One view of synth(P, seed) on this rank's GPU, ready to step."""

    def __init__(self, rast, scenes, P, W, H, deg, view_k, n_views, dev, kind="cube", poses=1, pose_stride=1):
        """poses > 1: step i renders camera (view_k + i * pose_stride) of the ring of n_views -- the reference's batch loop renders
        DIFFERENT cameras one after the other (train.py:198-226); poses = 1 repeats camera view_k."""
        self.rast, self.P, self.W, self.H = rast, P, W, H
        sc = scenes.synth(P, 0, sh_degree=deg) if kind == "cube" else scenes.synth_shell(P, 0, sh_degree=deg)
        self.sc = sc
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)  # noqa: E731
        bg = t(sc["bg"])

        def settings(k):
            cam = scenes.camera(k % max(n_views, 1), n_views, W, H)
            return cam, rast.GaussianRasterizationSettings(
                image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=bg,
                scale_modifier=1.0, viewmatrix=t(cam["viewmatrix"]), projmatrix=t(cam["projmatrix"]),
                sh_degree=deg, campos=t(cam["campos"]), prefiltered=False)

        cams = [settings(view_k + j * pose_stride) for j in range(max(poses, 1))]
        self.cam, self.rs = cams[0]
        self.rasters = [rast.GaussianRasterizer(rs) for _, rs in cams]
        self.step_no = 0
        self.leaves = {k: t(sc[k]).requires_grad_(True) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
        self.means2D = torch.zeros((P, 3), device=dev, requires_grad=True)
        self.params = list(self.leaves.values()) + [self.means2D]
        self.keep_grads = False      # True: a caller that reads the leaves' .grad after step() (tests): they are cleared at the START of the next step instead
        self.g = t(scenes.upstream_grad(H, W, 1))
        self.raster = self.rasters[0]
        self.dev = dev
        import view_parallel
        self.vp = view_parallel

    def step(self, bucket=None, world=1):
        L = self.leaves
        if bucket is not None or self.keep_grads:
            for p in self.params:
                p.grad = None
        if bucket is not None:
            bucket.zero_grad()          # start of a step: this backward writes into the arena (GradArena contract)
        raster = self.rasters[self.step_no % len(self.rasters)]
        self.step_no += 1
        color, radii, depth = raster(means3D=L["means3D"], means2D=self.means2D, opacities=L["opacities"],
                                     shs=L["shs"], scales=L["scales"], rotations=L["rotations"])
        color.backward(self.g)
        if bucket is None and not self.keep_grads:
            # optimizer.zero_grad(set_to_none=True) where the reference's loop has it: at the END of the iteration (train.py:221), behind the backward's
            # launches -- the host does it while the GPU works, not in front of the next forward's first launch
            for p in self.params:
                p.grad = None
        if bucket is not None and (world > 1 or self.vp.collectives_active()):      # (a one-rank group with forced collectives: tests/test_gpu_rccl.py)
            # the backward wrote the leaf gradients straight into the bucket (zero-copy GradArena)
            if getattr(bucket, "sh_factors", False):
                # all-reduce 11 + all-gather 3 floats/Gaussian -- sparse: of the rows some rank touched only
                self.exchanged = self.vp.exchange_gradients(bucket, L["means3D"].detach(), world, sparse=getattr(self, "sparse", False))
            else:
                self.vp.allreduce_mean_inplace(bucket.flat, world)                    # all-reduce 59 floats/Gaussian
                self.exchanged = {"allreduce": bucket.flat.numel() * 4, "allgather": 0, "rows": self.P}
        return radii

    def _forward_state(self, rs=None):
        _C = self.rast._C
        L = self.leaves
        e = torch.empty(0)
        rs = rs or self.rs
        R, color, radii, gb, bb, ib, depth = _C.rasterize_gaussians(
            rs.bg, L["means3D"].detach(), e, L["opacities"].detach(), L["scales"].detach(), L["rotations"].detach(),
            1.0, e, rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, self.H, self.W, L["shs"].detach(),
            rs.sh_degree, rs.campos, False)
        st = _C.debug_export(self.P, R, self.W, self.H, gb, bb, ib)
        nc = st["n_contrib"].to(torch.int64)
        gy, gx = (self.H + 15) // 16, (self.W + 15) // 16
        pad = torch.zeros((gy * 16, gx * 16), dtype=torch.int64, device=nc.device)
        pad[: self.H, : self.W] = nc
        tile_max = pad.view(gy, 16, gx, 16).amax(dim=(1, 3))
        rg = st["ranges"].to(torch.int64)
        return dict(R=int(R), R_eff=int(tile_max.sum().item()), P_vis=int((radii > 0).sum().item()),
                    pairs=int(nc.sum().item()), listed=int((rg[:, 1] - rg[:, 0]).sum().item()))

    def stats(self):
        """Measured R, R_eff, P_vis for the algorithmic-bytes formulas of SURVEY.md 8(d) (untimed forwards).
        R and R_eff are defined on the REFERENCE's tile lists (every tile of the 3-sigma square), so they are measured
        with tile_clip=0; the product default lists fewer instances (tile_clip=1): R_listed / R_eff_listed / Q."""
        _C = self.rast._C
        clip = _C.get_option("tile_clip")
        gy, gx = (self.H + 15) // 16, (self.W + 15) // 16
        per_pose = []
        for raster in self.rasters:         # every pose of the ring; the figures below are means over them
            rs = raster.raster_settings
            _C.set_option("tile_clip", 0)
            try:
                ref = self._forward_state(rs)
            finally:
                _C.set_option("tile_clip", clip)
            cur = self._forward_state(rs)
            per_pose.append(dict(R=ref["R"], R_eff=ref["R_eff"], P_vis=ref["P_vis"], pairs_fwd=ref["pairs"],
                                 R_listed=cur["listed"], R_eff_listed=cur["R_eff"], pairs_listed=cur["pairs"],
                                 Q=int(_C.get_option("last_runs")),
                                 # list cut (include/gsrast.h: options.no_list_cut): what this forward of the pose left out
                                 Q_early=_query(_C, "last_early_runs"), late=_query(_C, "last_late")))
        out = {k: (None if any(p[k] is None for p in per_pose) else int(round(sum(p[k] for p in per_pose) / len(per_pose)))) for k in per_pose[0]}
        out.update(T=gx * gy, N=self.W * self.H, poses=len(per_pose))
        return out


class Deformation:
    """A stand-in for what SaRO-GS's deformation field hands the rasterizer at timestamp t (scene/saro_gaussian.py:get_deformation, :782-847, with
    the shipped switches dx = drot = dopacity = True, arguments/__init__.py:68-72): per Gaussian a temporal position and a lifespan,
        opacity  = sigmoid(_opacity) * exp(-4 ((t - pos) / lifespan)^2)                                   (:791-792, :824-829)
        means3D  = _xyz + motion_residual(t),  rotations = normalize(_rotation + rot_residual[:, :4]),
        scales   = exp(_scaling + rot_residual[:, 4:])                                                   (:805-822)
    The residuals are smooth functions of (t - pos) with a fixed random direction per Gaussian: means move by up to 1.2 % of the scene's
    extent, scales by +-10 %, quaternions by ~3 degrees -- the size of a learned deformation, none of its cost (the reference's MLP heads
    are model code outside this path).  The tensors require a gradient, as the heads' outputs do: the backward writes their rows."""

    def __init__(self, P, dev, seed=5, motion=True):
        rng = np.random.default_rng(seed)
        t = lambda x: torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32, device=dev)  # noqa: E731
        self.tpos = t(rng.uniform(0.0, 1.0, size=(P, 1)))
        self.life = t(rng.uniform(0.2, 1.0, size=(P, 1)))
        self.ts = rng.uniform(0.0, 1.0, size=4096)
        self.motion = motion
        if motion:
            d = rng.normal(size=(P, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
            self.mdir = t(0.03 * d)
            self.rdir = t(np.concatenate([0.05 * rng.normal(size=(P, 4)), 0.1 * rng.uniform(-1.0, 1.0, size=(P, 3))], axis=1))

    def at(self, i):
        """(motion_residual, rot_residual, trbfoutput) of call i."""
        d = float(self.ts[i % len(self.ts)]) - self.tpos
        trbf = torch.exp(-4.0 * (d / self.life) ** 2)
        if not self.motion:
            return None, None, trbf
        s = torch.sin(6.283185307179586 * d)
        return (self.mdir * s).requires_grad_(True), (self.rdir * s).requires_grad_(True), trbf


def _query(_C, name):
    try:
        return int(_C.context_query(name))
    except ValueError:          # (a library without the list cut: A/B runs of tools/ab_variants.sh)
        return None


HOST_STEPS = {}     # per-step host enqueue times of the last timed() call: a stall of the host shows up here


def timed(workload, steps, warmup, bucket, world, vp, dev):
    for _ in range(warmup):
        workload.step(bucket, world)
    vp.barrier()
    torch.cuda.synchronize(dev)
    trace = os.environ.get("BENCH_STEP_TRACE")
    marks = []
    mallocs0 = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
    # Python's cyclic collector is paused over the timed window (a full collection walks every live object of the
    # process: milliseconds of host stall that have nothing to do with the step); nothing in a step relies on it
    import gc
    gc.collect()
    gc_was_enabled = gc.isenabled()
    gc.disable()
    t0 = time.perf_counter()
    for _ in range(steps):
        workload.step(bucket, world)
        marks.append(time.perf_counter())        # host-side enqueue time of each step (diagnostic only)
    t_loop = time.perf_counter()
    torch.cuda.synchronize(dev)
    t_sync = time.perf_counter()
    vp.barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    if gc_was_enabled:
        gc.enable()
    d = np.diff(np.array([t0] + marks)) * 1e3          # diagnostics, outside the timed window
    HOST_STEPS.clear()
    HOST_STEPS.update(median=round(float(np.median(d)), 4), max=round(float(d.max()), 4), argmax=int(d.argmax()),
                      drain_ms=round((t_sync - t_loop) * 1e3, 4), loop_ms=round((t_loop - t0) * 1e3, 4),
                      device_mallocs=int(torch.cuda.memory_stats(dev).get("num_device_alloc", 0) - mallocs0))
    if trace or d.max() > 20.0 * max(float(np.median(d)), 0.05):
        print("step host ms:", " ".join(f"{x:.2f}" for x in d), file=sys.stderr)
    return vp.max_over_ranks(dt, dev)



def timed_sync(workload, steps, bucket, world, vp, dev):
    """SURVEY.md 8(d) / BASELINE.md 2 as written (protocol adapted from the reference's test.py:155-168, renderer/__init__.py:149,202-203):
    every step is bracketed by device synchronisations and timed on the host's wall clock.  The K steps as a whole are bracketed by a
    barrier + synchronize on both sides (the contract's timed region).  Returns (median seconds per step -- the max over ranks of each
    rank's median --, mean seconds per step over the whole region, max over ranks)."""
    import gc
    vp.barrier()
    torch.cuda.synchronize(dev)
    gc.collect()
    gc_was_enabled = gc.isenabled()
    gc.disable()
    ts = []
    t0 = time.perf_counter()
    for _ in range(steps):
        s0 = time.perf_counter()
        workload.step(bucket, world)
        torch.cuda.synchronize(dev)
        ts.append(time.perf_counter() - s0)
    vp.barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    if gc_was_enabled:
        gc.enable()
    med = float(np.median(ts)) if ts else 0.0
    return vp.max_over_ranks(med, dev), vp.max_over_ranks(dt / max(steps, 1), dev), [round(x * 1e3, 4) for x in ts]


def _profile_json(name):
    path = os.path.join(ROOT, "profiles", name)
    try:
        return json.load(open(path))
    except Exception:
        return None


def roofline_of(st, launch_ms, P, kernel="blend_bwd_cull_t_kernel"):
    """The roofline object of a blend kernel, scalars only: the dominant one, blend_bwd_cull_t_kernel (`roofline`; reference backward.cu:399-557),
    or the forward blend_fwd_cull_kernel (`roofline_fwd`: north_star's "per-tile blend kernel", forward.cu:261-393).

      achieved / peak / frac   the contract's HBM pair: ALGORITHMIC bytes per launch (SURVEY.md 8d with this run's measured R_eff:
                               N*20 + R_eff*76 backward, R_eff*44 + N*24 forward) / this run's mean launch duration (HIP events on the launch
                               stream) / 8 TB/s
      traffic / traffic_ratio  HBM bytes per launch from the committed rocprofv3 PMC passes of the same workload (FETCH_SIZE x 2 + WRITE_SIZE,
                               gfx950 units per MI355X_MICROARCH.md; profiles/pmc_blend_*_3M.json -- not this run), over the algorithmic bytes
      binding = "valu issue"   what actually bounds the kernel: valu_issue_slot_frac = SQ_INSTS_VALU per launch (committed pass) x
                               the kernel's priced cycles per instruction / (1024 SIMDs x 2.4 GHz x this run's launch time), priced with the
                               MEASURED issue costs (v_fma_f32 2.4 cycles, tools/valu_calib.hip); ..._guide_2cyc_fma: the same with the guide's
                               2-cycle v_fma_f32 (MI355X_MICROARCH.md) -- the stricter ceiling."""
    fwd = kernel == "blend_fwd_cull_kernel"
    nbytes = st["N"] * 24 + st["R_eff"] * 44 if fwd else st["N"] * 20 + st["R_eff"] * 76
    achieved = nbytes / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
    out = {"kernel": kernel, "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None, "traffic_ratio": None,
           "algorithmic_bytes_per_launch": nbytes, "avg_launch_ms": round(launch_ms, 4), "binding": "valu issue",
           "valu_issue_slot_frac": None, "valu_issue_slot_frac_guide_2cyc_fma": None, "valu_wave_insts_per_launch": None,
           "counters_from": None}
    stem = "pmc_blend_fwd" if fwd else "pmc_blend_bwd"
    for name in (stem + "_3M.json", stem + ".json"):
        pj = _profile_json(name)
        if not (pj and pj.get("gaussians", 3_000_000 if "3M" in name else 1_000_000) == P):
            continue
        out["traffic"], out["counters_from"] = pj.get("hbm_bytes_per_launch"), "profiles/" + name + " (committed rocprofv3 pass, not this run)"
        if out["traffic"]:
            out["traffic_ratio"] = round(out["traffic"] / nbytes, 3)
        vi = pj.get("valu_wave_insts_per_launch")
        mix = None
        for r in ("r06", "r05", "r04", "r03", "r02"):
            mix = mix or _profile_json(r + "_valu_mix.json")
        if vi and launch_ms > 0 and mix and kernel in mix.get("kernels", {}):
            cyc = mix["kernels"][kernel]["avg_cycles_per_valu_inst"]
            rate = vi / (launch_ms * 1e-3)
            out["valu_wave_insts_per_launch"] = vi
            out["valu_issue_slot_frac"] = round(rate * cyc / (1024 * 2.4e9), 3)
            out["valu_issue_slot_frac_guide_2cyc_fma"] = round(rate * cyc * (2.0 / 2.4) / (1024 * 2.4e9), 3)
        break
    return out


def stage_table(_C, wl, st, P, deg, H):
    """Per-stage device times (every stage bracketed with HIP events; separate, untimed pass) next to this build's algorithmic
    bytes per launch."""
    dev = wl.dev
    _C.profile_reset()
    _C.set_option("profile", -1)
    for _ in range(10):
        wl.step(None, 1)
    torch.cuda.synchronize(dev)
    pk = _C.profile_read()
    _C.set_option("profile", 0)
    C = (deg + 1) ** 2
    Pv, R, Re, N, T = st["P_vis"], st["R"], st["R_eff"], st["N"], st["T"]
    Rl, Q = st["R_listed"], st.get("Q_early", st["Q"])      # (under the list cut: the early Gaussians' runs and instances)
    late_n = st.get("late") or 0
    Pe = max(Pv - late_n, 0)                                 # Gaussians whose colour is evaluated / whose gradient rows the per-Gaussian backward writes
    passes_t = 2 if T > 256 else 1
    run_binning = _C.get_option("binning") == 0 and T <= 65536 and (H + 15) // 16 <= 256
    bucket_sort = run_binning and _C.get_option("depth_sort") == 0 and P >= 32768
    alg = {
        "preprocess_fwd": P * (44 + 20) + Pv * (32 if late_n else 64),   # geometry half: in 44 B, out radii / tiles / rect / depth key 20 B + rec0, rec1 32 B (+ binrec 32 B when the list cut is not in force) per visible Gaussian
        "preprocess_color": (Pe if late_n else P) * (12 + 12 * C + 17 + (36 if deg > 0 else 0)) + (P if late_n else 0),   # colour half (side stream, beside the binning): means + SH in, rec2 + clamp flags + the 9 direction derivatives out (list cut: early Gaussians only, + a flag byte each)
        # bucket depth sort (default): scatter reads key + rect + tiles (16 B), writes a 16-B slab element; the sort kernel reads it
        # (twice, the second time from L2) and writes id + width scan (8 B) -- per visible Gaussian; radix passes: 20 B x 4
        "sort_depth": (P * 16 + Pv * 40) if bucket_sort else P * 20 * 4,
        "scan_tiles": (0 if bucket_sort else P * 12) if run_binning else P * 24,   # bucket sort: totals by the run emission's last workgroup
        # run-compressed: Q column runs of 10 B emitted, sorted by column (one pass), expanded once into Rl instances
        "emit_instances": (((Pe * 48 + P * 8) if late_n else P * 40) + Q * 10) if run_binning else (P * 24 + R * 6),
        "sort_tile": (Q * 22 + Q * 16 + Rl * 4) if run_binning else R * 14 * passes_t,
        "tile_ranges": T * 8 if run_binning else R * 2 + T * 8,
        "blend_fwd": Re * 44 + N * 24,
        "blend_bwd": N * 20 + Re * 76,
        # in: mean 12, radius 4, scale 12, rotation 16, gradient record 64, clamp flags 1, colour / direction derivatives 36;
        # out: dL/dmean2D 12, dL/dopacity 4, dL/dmean3D 12, dL/dsh 12 C, dL/dscale 12, dL/drot 16  (the SH block is not read any more)
        # (list cut: the grouped kernel reads and writes the rows of the Gaussians that are NOT late -- Pe of them -- plus a bit per
        # Gaussian; the late ones' zero rows are late_rows_zero's, written beside the blend backward)
        "preprocess_bwd": (Pe * (12 * C + 201) + P // 8) if late_n else P * (12 * C + 201),
        "late_rows_zero": late_n * (12 * C + 12 + 4 + 12 + 12 + 16) if late_n else 0,   # dL/dsh, dL/dmean2D, dL/dopacity, dL/dmean3D, dL/dscale, dL/drot rows of zeros
        "sh_dir_derivs": Pv * (12 * C + 12 + 36) + P * 4,       # side stream, beside the blend backward: SH + mean in, 36 B out
        "cut_redo": 0,                                          # list cut: the predicated second binning + blend (ten launches that return at once)
        "grec_zero_touched": Pe * 64 + P // 8,                  # behind the last blend: a bit per Gaussian in, one 64-byte record per consumed Gaussian out (at most the early ones)
    }
    per_kernel = {}
    for name, nbytes in alg.items():
        if name not in pk:
            continue
        ms = pk[name][0] / max(pk[name][1], 1)
        per_kernel[name] = {"ms": round(ms, 4), "algorithmic_MB": round(nbytes / 1e6, 1),
                            "GBps": round(nbytes / (ms * 1e-3) / 1e9, 1) if ms > 0 else None,
                            "hbm_frac": round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms > 0 else None}
    return per_kernel, pk


def step_bytes(st, per_kernel, P, deg, ms_per_step):
    """Algorithmic bytes of ONE step (forward + backward of one view): (a) what THIS build's kernels move by their own accounting (the
    stage table's algorithmic bytes: column runs, early sets, gradient records), (b) SURVEY.md 8(d)'s formula for the reference's
    data movement (one 64-bit key per instance, six 8-bit sort passes, nine zero-filled gradient tensors) with this run's measured
    R, R_eff, P_vis.  (b) / ms_per_step can exceed the HBM peak: this build does not move those bytes."""
    C = (deg + 1) ** 2
    R, Re, Pv, N, T = st["R"], st["R_eff"], st["P_vis"], st["N"], st["T"]
    bits = 32 + max(1, (T - 1).bit_length())
    passes = (bits + 7) // 8
    fwd = P * (44 + 12 * C) + Pv * 64 + P * 8 + R * 12 * (1 + 2 * passes) + R * 8 + T * 8 + Re * 44 + N * 24
    bwd = N * 20 + Re * 76 + Pv * (56 + 36) + Pv * (12 * C + 92) + Pv * (12 * C + 40) + P * 300
    mine = sum(v["algorithmic_MB"] for v in (per_kernel or {}).values()) * 1e6 if per_kernel else None
    out = {"survey_8d_formula_MB": round((fwd + bwd) / 1e6, 1), "survey_8d_formula_over_ms_per_step_GBps": round((fwd + bwd) / (ms_per_step * 1e-3) / 1e9, 1),
           "this_build_MB": round(mine / 1e6, 1) if mine else None,
           "this_build_over_ms_per_step_GBps": round(mine / (ms_per_step * 1e-3) / 1e9, 1) if mine else None,
           "this_build_frac_of_hbm_peak": round(mine / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if mine else None,
           "note": "this_build = sum of per_stage.*.algorithmic_MB (this build's own data movement: column runs instead of 64-bit instance keys, one "
                   "instance-level pass, early sets under the list cut, no gradient zero-fill); the survey formula prices the reference's data movement "
                   "and is not a bound on this build"}
    return out


def cpu_baseline(scenes, P, W, H, deg, budget_s):
    """The oracle (CPU port, fp32, OpenMP over Gaussians / tiles) on the host cores."""
    from oracle import oracle as orc
    orc.build()
    orc.set_exp_mode(0)
    cores = os.cpu_count() or 1

    def run(p, w, h, min_s=0.0, max_views=1):
        """mean seconds per view over enough views (of the same scene, other cameras) to fill ~min_s of CPU work"""
        sc = scenes.synth(p, 0, sh_degree=deg)
        g = scenes.upstream_grad(h, w, 1)
        total, n = 0.0, 0
        while n < max_views and (n == 0 or total < min_s):
            cam = scenes.camera(n, max_views, w, h)
            t0 = time.perf_counter()
            orc.render(sc, cam, g)
            total += time.perf_counter() - t0
            n += 1
        return total / n, n, total

    run(2000, 128, 96)                      # page in the library / spin up the OpenMP pool
    t_small, _, _ = run(10_000, 400, 400)   # BASELINE config 1
    # pixel-work scales with the image area; use config 1 to predict the full view
    predict = t_small * (W * H) / (400 * 400) * 1.5
    if predict <= budget_s:
        # a bounded sample of the same workload: whole views until ~12 s of CPU work (at most 8 views)
        t, n, total = run(P, W, H, min_s=min(12.0, budget_s), max_views=8)
        return dict(value=1.0 / t, unit="views/s", cores=cores, kind="port",
                    sample=f"{n} view(s) fwd+bwd of the same workload (P={P}, {W}x{H}, SH{deg}), {total:.1f} s of CPU work")
    return dict(value=1.0 / t_small, unit="views/s", cores=cores, kind="port",
                sample=f"1 view fwd+bwd of BASELINE config 1 (P=10000, 400x400, SH{deg}), {t_small:.2f} s; "
                       f"the full workload was predicted at {predict:.0f} s > budget")



def sweep_point(rast, scenes, vp, P, W, H, deg, dev, steps, warmup, poses, kind="cube"):
    """One more #Gaussians point with the headline's protocol: pose table OFF, per-call synchronised, median."""
    _C = rast._C
    wl = Workload(rast, scenes, P, W, H, deg, 0, max(poses, 1), dev, kind=kind, poses=poses)
    _C.set_option("no_order_hint", 1)
    try:
        for _ in range(max(warmup, 2 * poses)):
            wl.step(None, 1)
        med, _, _ = timed_sync(wl, steps, None, 1, vp, dev)
    finally:
        _C.set_option("no_order_hint", 0)
    del wl
    torch.cuda.empty_cache()
    return round(1.0 / med, 1)


def self_launch(a):
    """`python bench.py --gpus N` with N > 1 and no torchrun environment: start the N ranks ourselves -- the same command line under
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` (one process per GPU, rendezvous on 127.0.0.1) -- and pass
    rank 0's JSON line and the exit code through."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")))



def compact_line(result: dict) -> str:
    """The final stdout line: the result as compact JSON, at most LINE_LIMIT bytes.  Optional diagnostics are dropped, least important
    first, should a future field push it over (the contract's keys, roofline and cpu_baseline are never dropped)."""
    optional = ["per_step_ms", "stage_ms", "host", "sweep_1080p_cold", "pipelined", "value_no_list_cut", "roofline_fwd"]
    r = dict(result)
    line = json.dumps(r, separators=(",", ":"))
    while len(line) > LINE_LIMIT and optional:
        r.pop(optional.pop(0), None)
        line = json.dumps(r, separators=(",", ":"))
    if len(line) > LINE_LIMIT:
        raise RuntimeError(f"bench.py: the result line is {len(line)} bytes (> {LINE_LIMIT}) even without its optional fields")
    return line


def main():
    a = parse()
    if a.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(a)
    import view_parallel as vp
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != a.gpus:     # never print a line whose n_gpus differs from what was asked for
        print(f"bench.py: --gpus {a.gpus} but the launcher started {world_env} rank(s) (WORLD_SIZE); refusing to run", file=sys.stderr)
        raise SystemExit(2)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the rasterizer has no CPU fallback")
    # GSRAST_SINGLE_DEVICE=1: every rank on cuda:0 (only to exercise the N>1 path on a 1-GPU box, with gloo)
    single = os.environ.get("GSRAST_SINGLE_DEVICE") == "1"
    if a.gpus > 1 and not single and torch.cuda.device_count() < a.gpus:
        print(f"bench.py: --gpus {a.gpus} but only {torch.cuda.device_count()} device(s) are visible; refusing to run", file=sys.stderr)
        raise SystemExit(2)
    if a.force_collectives:
        if a.gpus != 1:
            raise SystemExit("bench.py: --force-collectives is for --gpus 1 (N > 1 runs the collectives anyway)")
        vp.force_collectives(True)
        a.sweep, a.no_cpu_baseline, a.extras = "", True, None
    rank, local, world = vp.init_from_env()
    assert world == a.gpus
    dev = torch.device("cuda", local if (world > 1 and not single) else 0)
    torch.cuda.set_device(dev)
    import diff_gaussian_rasterization_ch3 as rast
    import scenes
    _C = rast._C
    if a.exp_mode is not None:
        _C.set_option("exp_mode", a.exp_mode)
    exp_mode = _C.get_option("exp_mode")
    if a.ablate:
        _C.set_option("ablate", a.ablate)
    for kv in a.opt:
        name, _, val = kv.partition("=")
        _C.set_option(name, int(val))
    for flag, name in ((a.ppl, "pixels_per_lane"), (a.ppl_fwd, "fwd_pixels_per_lane"), (a.ppl_bwd, "bwd_pixels_per_lane")):
        if flag:
            _C.set_option(name, flag)
    if a.no_cull:
        _C.set_option("cull", 0)
    if a.no_lpt:
        _C.set_option("lpt", 0)
    if a.binning is not None:
        _C.set_option("binning", a.binning)

    P, W, H, deg = a.gaussians, a.width, a.height, a.sh_degree
    # every rank renders its own `poses` cameras of one ring round-robin: step i of rank r = camera r + i * world (mod poses * world)
    n_poses = max(a.poses, 1)
    wl = Workload(rast, scenes, P, W, H, deg, view_k=rank, n_views=max(world, 1) * n_poses, dev=dev, kind=a.scene, poses=n_poses, pose_stride=max(world, 1))
    bucket = None
    exchanging = world > 1 or a.force_collectives
    if exchanging:
        wl.force_exchange = a.force_collectives
        # one flat fp32 buffer holds every leaf gradient of the rasterizer (59 floats / Gaussian); the backward writes into it directly.
        # Default exchange (gather): ONE all-gather of the 64-byte rows each rank's own view touched, added in rank order; --exchange
        # allreduce: ONE in-place all-reduce of all 59 floats.  All forms give the batch-mean gradient of set_batch_gradient
        # (saro_gaussian.py:266-276).
        bucket = _C.GradArena(P, 16, dev, sh_factors=(a.exchange != "allreduce"), world=world)
        _C.set_grad_arena(bucket)
        wl.sparse = "gather" if a.exchange == "gather" else a.exchange == "sparse"
        if a.exchange == "factors":
            vp.overlap_factor_exchange(True)     # the all-gather starts between the two phases of the backward

    names = [_C.lib().gsrast_profile_kernel_name(k).decode() for k in range(_C.lib().gsrast_profile_kernel_count())]
    kid = {n: i for i, n in enumerate(names)}

    # ---- headline leg: pose table OFF (every forward a first visit), per-call synchronised ----
    _C.set_option("no_order_hint", 1)
    _C.profile_reset()
    # only the roofline kernel is bracketed with events inside the timed region (2 records/step, ~5 us of GPU idle each)
    _C.set_option("profile", 1 << kid["blend_bwd"])
    n_pre = 0
    if a.preroll_ms > 0:        # settle the device first (see --preroll-ms); reported as preroll_steps
        t_pre = time.perf_counter()
        while n_pre < 5000:
            for _ in range(20):
                wl.step(bucket, world)
            n_pre += 20
            torch.cuda.synchronize(dev)
            # the same elapsed time on every rank, so every rank runs the same number of rounds (and collectives)
            if vp.max_over_ranks(time.perf_counter() - t_pre, dev) * 1e3 >= a.preroll_ms:
                break
    for _ in range(a.warmup):
        wl.step(bucket, world)
    torch.cuda.synchronize(dev)
    _C.profile_reset()
    med, mean, per_step = timed_sync(wl, a.steps, bucket, world, vp, dev)
    prof = _C.profile_read()
    _C.set_option("profile", 0)
    late_cold = _query(_C, "last_late")
    d_pipe_cold = timed(wl, a.steps, 0, bucket, world, vp, dev)
    host_cold = dict(HOST_STEPS)

    st, per_kernel, pk = None, None, None
    if world == 1 and not exchanging:
        st = wl.stats()             # (measured under the headline's options: table off)
        per_kernel, pk = stage_table(_C, wl, st, P, deg, H)
    elif rank == 0:
        st = wl.stats()
    _C.set_option("no_order_hint", 0)

    # ---- the same protocol with the pose table ON: every pose seen before (an unchanged scene over a fixed rig) ----
    warm_warmup = max(a.warmup, 3 * n_poses)
    for _ in range(warm_warmup):
        wl.step(bucket, world)
    torch.cuda.synchronize(dev)
    med_w, mean_w, _ = timed_sync(wl, a.steps, bucket, world, vp, dev)
    late_warm = _query(_C, "last_late")
    d_pipe_warm = timed(wl, a.steps, 0, bucket, world, vp, dev)

    # ---- ... and with the LIST CUT switched off altogether (the floor for a scene no prediction can cut) ----
    med_nc = None
    if world == 1 and not exchanging:
        _C.set_option("no_list_cut", 1)
        try:
            for _ in range(a.warmup):
                wl.step(None, 1)
            med_nc, _, _ = timed_sync(wl, a.steps, None, 1, vp, dev)
        finally:
            _C.set_option("no_list_cut", 0)

    result = None
    scene_fn = "synth" if a.scene == "cube" else "synth_shell"
    ranks_info = None
    if exchanging and torch.distributed.is_initialized():
        mine = {"rank": rank, "device": int(dev.index or 0), "name": torch.cuda.get_device_name(dev)}
        ranks_info = [None] * torch.distributed.get_world_size()
        torch.distributed.all_gather_object(ranks_info, mine)
    if rank == 0:
        bwd_ms = prof["blend_bwd"][0] / max(prof["blend_bwd"][1], 1)
        fwd_ms = pk["blend_fwd"][0] / max(pk["blend_fwd"][1], 1) if per_kernel else 0.0   # (the event-bracketed all-stages pass behind the timed region)
        try:
            nccl_version = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception:       # noqa: BLE001
            nccl_version = None
        result = {
            "metric": "rendered views/s (fwd+bwd) at 1080p vs #Gaussians",
            "value": round(world / med, 3), "unit": "views/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(med * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "preroll_steps": n_pre,
            # the K-step region as a whole (barrier + synchronize on both sides, every step synchronised inside): wall time / K
            "mean_ms_per_step": round(mean * 1e3, 4),
            "value_warm": round(world / med_w, 3), "ms_per_step_warm": round(med_w * 1e3, 4), "warmup_warm": warm_warmup,
            "value_no_list_cut": round(1.0 / med_nc, 3) if med_nc else None,
            "pipelined": {"views_per_s_cold": round(world * a.steps / d_pipe_cold, 1), "views_per_s_warm": round(world * a.steps / d_pipe_warm, 1),
                          "note": "K steps back to back, one sync (rounds 1-5's headline): throughput, not the metric"},
            "config": {"workload": f"BASELINE configs[4] stress-1080p: {scene_fn}(P={P}, seed 0) SH{deg}, {W}x{H}, one view per GPU, fwd+bwd",
                       "protocol": "SURVEY 8d: 1 / median of per-call device-synchronised wall-clock fwd+bwd",
                       "pose_table": "off for value (every pose a first visit); on for value_warm",
                       "gaussians": P, "width": W, "height": H, "sh_degree": deg, "exp_mode": exp_mode, "poses_per_rank": n_poses,
                       "views_per_step": world, "R": st["R"], "R_listed": st["R_listed"], "Q": st["Q"], "R_eff": st["R_eff"], "visible": st["P_vis"],
                       "late_gaussians_cold": late_cold, "late_gaussians_warm": late_warm,
                       "word_fork": _C.get_option("word_fork"),
                       "exchange": a.exchange if exchanging else None},
            "roofline": roofline_of(st, bwd_ms, P),
            "roofline_fwd": roofline_of(st, fwd_ms, P, kernel="blend_fwd_cull_kernel") if fwd_ms > 0 else None,
            "stage_ms": {k: v["ms"] for k, v in per_kernel.items()} if per_kernel else None,
            "host": {"enqueue_ms_median": host_cold.get("median"), "device_mallocs_in_timed_steps": host_cold.get("device_mallocs")},
            "per_step_ms": {"min": min(per_step), "max": max(per_step)} if per_step else None,
        }
        if exchanging:
            result["config"].update({
                "exchange_backend": torch.distributed.get_backend() if torch.distributed.is_initialized() else None,
                "rccl_world": torch.distributed.get_world_size() if torch.distributed.is_initialized() else None,
                "rccl_version": nccl_version,
                "rank_devices": [r["device"] for r in ranks_info] if ranks_info else None,
                "exchange_bytes_per_rank_and_step": getattr(wl, "exchanged", None)})

    # ---- the metric's "vs #Gaussians": more points at 1080p, same protocol (single GPU only) ----
    if rank == 0 and world == 1 and a.sweep:
        del wl
        torch.cuda.empty_cache()
        sweep = {}
        for p in [int(x) for x in a.sweep.split(",") if x]:
            sweep[str(p)] = result["value"] if p == P else sweep_point(rast, scenes, vp, p, W, H, deg, dev, a.steps, a.warmup, n_poses)
        sweep[str(P)] = result["value"]
        result["sweep_1080p_cold"] = sweep
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        try:
            result["cpu_baseline"] = cpu_baseline(scenes, P, W, H, deg, a.cpu_budget_s)
        except Exception as e:  # the baseline is reported, never required for the GPU number
            result["cpu_baseline"] = {"value": None, "unit": "views/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e}"}
    if rank == 0 and world == 1 and a.extras:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_extras
        report = bench_extras.run(a, rast, scenes, vp, dev, dict(result, per_stage=per_kernel, stats=st))
        os.makedirs(os.path.dirname(os.path.abspath(a.extras)), exist_ok=True)
        with open(a.extras, "w") as f:
            json.dump(report, f, indent=1)
        print(json.dumps({"bench_report": a.extras, "report": report}), flush=True)       # an EARLIER line: never the last one
        result["extras"] = a.extras if not os.path.isabs(a.extras) else os.path.relpath(a.extras, ROOT)
    if rank == 0:
        print(compact_line(result), flush=True)
    if world > 1:
        vp.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
