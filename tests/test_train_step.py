"""-m gpu: a static-stage training iteration assembled from the pieces of this repository only --
raw parameters -> fused activation epilogue -> rasterizer -> fused L1 + D-SSIM loss -> backward -> per-row-LR Adam
(the call sequence of /root/reference/train.py:190-250 with scene/saro_gaussian.py's static stage).
Sanity of the whole chain: fitting a perturbed scene to a target rendering must reduce the loss."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fused", [False, True], ids=["epilogue_then_rasterizer", "raw_entry_points"])
def test_static_stage_iterations_reduce_the_loss(fused, scenes, rast, gpu):
    """fused: the activations inside the rasterizer's per-Gaussian kernels (GaussianRasterizerRaw, round 3) instead of the standalone
    epilogue in front of the drop-in module."""
    from conftest import settings_from
    import fused_adam
    import fused_epilogue
    import fused_loss
    P, W, H = 4000, 160, 112
    sc = scenes.synth(P, 141, scale_mul=1.5)
    cam = scenes.camera(0, 1, W, H)
    rs = settings_from(rast, cam, sc, gpu)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=gpu)  # noqa: E731

    def raw_from(scene):
        return dict(xyz=t(scene["means3D"]), rotation=t(scene["rotations"]), scaling=torch.log(t(scene["scales"])),
                    opacity=torch.logit(t(scene["opacities"]).clamp(1e-4, 1 - 1e-4)), f_dc=t(scene["shs"][:, :1]),
                    f_rest=t(scene["shs"][:, 1:]))

    def render(raw):
        if fused:
            m2 = torch.zeros((P, 3), device=gpu, requires_grad=True)
            return rast.GaussianRasterizerRaw(rs)(raw["xyz"], m2, raw["rotation"], raw["scaling"], raw["opacity"], raw["f_dc"], raw["f_rest"])[0]
        motion, rot, scale, opa, shs = fused_epilogue.activate_gaussians(raw["xyz"], raw["rotation"], raw["scaling"], raw["opacity"],
                                                                          raw["f_dc"], raw["f_rest"])
        m2 = torch.zeros((P, 3), device=gpu, requires_grad=True)
        color, radii, depth = rast.GaussianRasterizer(rs)(means3D=motion, means2D=m2, opacities=opa, shs=shs, scales=scale, rotations=rot)
        return color

    with torch.no_grad():
        gt = render(raw_from(sc)).clone()
    rng = np.random.default_rng(142)
    pert = dict(sc)
    pert["shs"] = (sc["shs"] + 0.3 * rng.normal(size=sc["shs"].shape)).astype(np.float32)
    pert["opacities"] = np.clip(sc["opacities"] * rng.uniform(0.5, 1.0, size=sc["opacities"].shape), 1e-3, 0.999).astype(np.float32)
    raw = {k: v.requires_grad_(True) for k, v in raw_from(pert).items()}
    lrs = dict(xyz=1.6e-4, f_dc=2.5e-2, f_rest=2e-3, opacity=5e-2, scaling=5e-3, rotation=1e-3)
    inv = torch.ones(P, 1, device=gpu)                                   # static stage: inv_intergral = 1 (saro_gaussian.py:362-364)
    opt = fused_adam.GaussianAdam([{"params": [raw[k]], "lr": lrs[k] * inv if k != "f_rest" else lrs[k], "name": k} for k in raw], eps=1e-15)
    losses = []
    for it in range(40):
        loss = fused_loss.l1_dssim_loss(render(raw), gt, 0.2)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(np.isfinite(losses))
    assert losses[-1] < 0.6 * losses[0], (losses[0], losses[-1])
    assert min(losses[-5:]) <= min(losses[:5])
