"""The build's own counterpart of the rasterizer's CALLER: a restatement of how SaRO-GS drives
diff_gaussian_rasterization_ch3 -- /root/reference/renderer/__init__.py:35-138 (train_render) and :140-228 (test_render) -- written
for the tests; not shipped, no reference text.  What it keeps from the reference, because the drop-in has to survive exactly this:

  * `screenspace_points = zeros_like(xyz, requires_grad=True) + 0` is a NON-LEAF tensor with retain_grad(): its .grad[:, :2] norm
    drives densification (train.py:212; saro_gaussian.py:742);
  * settings are built per call from a camera object (tan(0.5 FoV), the camera's matrices moved with .cuda()), the module is
    constructed per call, forward is called by KEYWORD with every optional present, the absent ones as explicit None;
  * the returned dict: render, viewspace_points, visibility_filter = radii > 0, radii (+ depth / opacity in the eval flavour);
  * the segment pass of test_render: colors_precomp = lifespan.detach().expand(-1, 3) (a non-contiguous view), shs = None.

`pc` is any object with the GaussianModel accessors used here (get_xyz, get_opacity, get_scaling, get_rotation, get_features,
active_sh_degree, get_lifespan); TinyGaussians below computes them from raw leaves with the activations of
scene/saro_gaussian.py:39-47."""
import math

import torch
import torch.nn.functional as F

from diff_gaussian_rasterization_ch3 import GaussianRasterizationSettings, GaussianRasterizer


class TinyCamera:
    """The camera attributes the renderer reads (scene/cameras.py:84-101), from a scenes.camera() dict."""

    def __init__(self, cam: dict, timestamp: float = 0.0):
        self.image_height, self.image_width = cam["image_height"], cam["image_width"]
        self.FoVx, self.FoVy = 2.0 * math.atan(cam["tanfovx"]), 2.0 * math.atan(cam["tanfovy"])
        self.world_view_transform = torch.from_numpy(cam["viewmatrix"].copy())      # CPU tensors, moved per call like the reference's
        self.full_proj_transform = torch.from_numpy(cam["projmatrix"].copy())
        self.camera_center = torch.from_numpy(cam["campos"].copy())
        self.timestamp = timestamp


class TinyGaussians(torch.nn.Module):
    """Raw leaves + the accessors of GaussianModel (scene/saro_gaussian.py:39-47, get_* properties)."""

    def __init__(self, scene: dict, device, sh_degree: int):
        super().__init__()
        t = lambda a: torch.nn.Parameter(torch.as_tensor(a, dtype=torch.float32, device=device).contiguous())  # noqa: E731
        op = torch.as_tensor(scene["opacities"], dtype=torch.float64).clamp(1e-6, 1 - 1e-6)
        self._xyz = t(scene["means3D"])
        self._features_dc = t(scene["shs"][:, :1])
        self._features_rest = t(scene["shs"][:, 1:])
        self._scaling = t(torch.log(torch.as_tensor(scene["scales"], dtype=torch.float64)))
        self._rotation = t(torch.as_tensor(scene["rotations"]) * 1.7)
        self._opacity = t(torch.log(op / (1 - op)))
        self._lifespan = t(torch.linspace(0.05, 0.95, self._xyz.shape[0]).reshape(-1, 1))
        self.active_sh_degree = sh_degree

    get_xyz = property(lambda self: self._xyz)
    get_scaling = property(lambda self: torch.exp(self._scaling))
    get_rotation = property(lambda self: F.normalize(self._rotation))
    get_opacity = property(lambda self: torch.sigmoid(self._opacity))
    get_features = property(lambda self: torch.cat((self._features_dc, self._features_rest), dim=1))
    get_lifespan = property(lambda self: self._lifespan)


def _settings(viewpoint_camera, pc, bg_color, scaling_modifier):
    tanfovx = math.tan(viewpoint_camera.FoVx * 0.5)
    tanfovy = math.tan(viewpoint_camera.FoVy * 0.5)
    return GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width), tanfovx=tanfovx, tanfovy=tanfovy,
        bg=bg_color, scale_modifier=scaling_modifier, viewmatrix=viewpoint_camera.world_view_transform.cuda(),
        projmatrix=viewpoint_camera.full_proj_transform.cuda(), sh_degree=pc.active_sh_degree,
        campos=viewpoint_camera.camera_center.cuda(), prefiltered=False)


def train_render(viewpoint_camera, pc, bg_color, scaling_modifier=1.0):
    screenspace_points = torch.zeros_like(pc.get_xyz, dtype=pc.get_xyz.dtype, requires_grad=True, device="cuda") + 0
    try:
        screenspace_points.retain_grad()
    except Exception:       # noqa: BLE001  (the reference swallows this too)
        pass
    rasterizer = GaussianRasterizer(raster_settings=_settings(viewpoint_camera, pc, bg_color, scaling_modifier))
    rendered_image, radii, _ = rasterizer(
        means3D=pc.get_xyz, means2D=screenspace_points, shs=pc.get_features, colors_precomp=None, opacities=pc.get_opacity,
        scales=pc.get_scaling, rotations=pc.get_rotation, cov3D_precomp=None)
    return {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0, "radii": radii}


def test_render(viewpoint_camera, pc, bg_color, require_segment=False):
    screenspace_points = torch.zeros_like(pc.get_xyz, dtype=pc.get_xyz.dtype, requires_grad=True, device="cuda") + 0
    try:
        screenspace_points.retain_grad()
    except Exception:       # noqa: BLE001
        pass
    rasterizer = GaussianRasterizer(raster_settings=_settings(viewpoint_camera, pc, bg_color, 1.0))
    means3D, rotations, scales, opacity, shs = pc.get_xyz, pc.get_rotation, pc.get_scaling, pc.get_opacity, pc.get_features
    rendered_image, radii, depth = rasterizer(
        means3D=means3D, means2D=screenspace_points, shs=shs, colors_precomp=None, opacities=opacity, scales=scales, rotations=rotations,
        cov3D_precomp=None)
    res = {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0, "radii": radii,
           "opacity": opacity, "depth": depth}
    if require_segment:
        colors_precomp = pc.get_lifespan.detach().expand(-1, 3)
        rendered_image, radii, depth = rasterizer(
            means3D=means3D, means2D=screenspace_points, shs=None, colors_precomp=colors_precomp, opacities=opacity, scales=scales,
            rotations=rotations, cov3D_precomp=None)
        res["segment_render"] = rendered_image
    return res


test_render.__test__ = False        # a caller, not a pytest test
