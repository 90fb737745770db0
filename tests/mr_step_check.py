"""Helper of tests/test_gpu_multirank.py (run under torch.distributed.run, 2 ranks, gloo, both ranks on cuda:0):
view_parallel.distributed_step over the REAL chain -- raw parameters -> fused activation epilogue -> rasterizer -> fused
L1 + D-SSIM loss -- with 4 views on 2 ranks, against the reference's sequential batch loop (train.py:190-226,
saro_gaussian.py:226-294) run by every rank on its own: leaf gradients (batch mean) and the densification statistics."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "saro-gs_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import numpy as np  # noqa: E402
import torch  # noqa: E402
import scenes  # noqa: E402
import view_parallel as vp  # noqa: E402
import diff_gaussian_rasterization_ch3 as rast  # noqa: E402
import fused_epilogue  # noqa: E402
import fused_loss  # noqa: E402
from conftest import settings_from  # noqa: E402


def main():
    rank, local, world = vp.init_from_env()
    dev = torch.device("cuda:0")
    P, W, H, V = 12000, 256, 192, 4
    sc = scenes.synth(P, 151, scale_mul=1.2)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)  # noqa: E731
    cams = [scenes.camera(k, V, W, H) for k in range(V)]
    rss = [settings_from(rast, c, sc, dev) for c in cams]
    rng = np.random.default_rng(152)
    gts = [t(rng.uniform(0, 1, size=(3, H, W))) for _ in range(V)]

    def make_raw():
        return {k: v.requires_grad_(True) for k, v in dict(
            _xyz=t(sc["means3D"]), _rotation=t(sc["rotations"]), _scaling=torch.log(t(sc["scales"])),
            _opacity=torch.logit(t(sc["opacities"]).clamp(1e-4, 1 - 1e-4)), _features_dc=t(sc["shs"][:, :1]),
            _features_rest=t(sc["shs"][:, 1:])).items()}

    def render_loss(raw, k):
        motion, rot, scale, opa, shs = fused_epilogue.activate_gaussians(raw["_xyz"], raw["_rotation"], raw["_scaling"], raw["_opacity"],
                                                                          raw["_features_dc"], raw["_features_rest"])
        m2 = torch.zeros((P, 3), device=dev, requires_grad=True)
        color, radii, depth = rast.GaussianRasterizer(rss[k])(means3D=motion, means2D=m2, opacities=opa, shs=shs, scales=scale, rotations=rot)
        return {"loss": fused_loss.l1_dssim_loss(color, gts[k], 0.2), "viewspace_points": m2, "visibility_filter": radii > 0, "radii": radii}

    # the reference's loop, sequential, on this rank alone
    ref = make_raw()
    cache = {n: torch.zeros_like(p) for n, p in ref.items()}
    pg, vf, rd = [], [], []
    for k in range(V):
        out = render_loss(ref, k)
        out["loss"].backward()
        pg.append(torch.norm(out["viewspace_points"].grad[:, :2], dim=-1)); vf.append(out["visibility_filter"]); rd.append(out["radii"])
        for n, p in ref.items():
            cache[n] += p.grad.clone()
            p.grad = None
    want = {n: c / V for n, c in cache.items()}
    count = torch.stack(vf, 1).sum(1)
    wg = torch.stack(pg, 1).sum(1)
    wg[count > 0] = wg[count > 0] / count[count > 0]

    raw = make_raw()
    bucket = vp.StepBucket(raw)
    in_flight = int(os.environ.get("GSRAST_TEST_IN_FLIGHT", "1"))      # > 1: this rank's views alternate between that many streams
    stats = vp.distributed_step(bucket, list(range(V)), lambda k: render_loss(raw, k), views_in_flight=in_flight)
    if in_flight > 1:       # and again: the lanes' partial caches start from zero each step
        first = {n: raw[n].grad.clone() for n in want}
        stats = vp.distributed_step(bucket, list(range(V)), lambda k: render_loss(raw, k), views_in_flight=in_flight)
        for n in want:
            assert float((first[n] - raw[n].grad).abs().max()) <= 1e-6 + 1e-3 * float(first[n].abs().max()), n
    torch.cuda.synchronize()
    worst = 0.0
    for n in want:
        a, b = want[n], raw[n].grad
        worst = max(worst, float(((a - b).abs() / (1e-7 + 1e-4 * a.abs().max())).max()))       # float-atomic order + summation order
        assert float(a.abs().max()) > 0, n
    ok_stats = (torch.equal(stats["visibility_count"], count.float()) and torch.equal(stats["radii"], torch.stack(rd, 1).max(1)[0].float())
                and bool(((stats["viewspace_point_grad"][:, 0] - wg).abs() <= 1e-9 + 1e-4 * wg.abs()).all()))
    print(f"STEP_CHECK rank {rank} worst {worst:.3f} stats_ok {ok_stats}", flush=True)
    vp.barrier()
    if worst > 1.0 or not ok_stats:
        sys.exit(3)


if __name__ == "__main__":
    main()
