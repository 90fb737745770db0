// gsrast_epilogue.h -- fused activation / deformation epilogue that produces the rasterizer's inputs
// (SURVEY.md 8f, rank 3).  Reference behaviour restated (paths relative to /root/reference/):
//   scene/saro_gaussian.py:807-847 (get_deformation tail) and :39-47 (activations):
//     motion  = _xyz + motion_residual                                   (args.dx)
//     rot     = F.normalize(_rotation + rot_residual[:, :4])             (args.drot; normalize: x / max(|x|, 1e-12))
//     scale   = exp(_scaling + rot_residual[:, 4:])
//     opacity = sigmoid(_opacity) * trbfoutput                           (args.dopacity)
//     shs     = cat(_features_dc, _features_rest, dim=1) + shs_residual  (args.dsh)
//   every residual / trbf may be absent (the static stage: plain activations, renderer/__init__.py:60-75).
// PyTorch runs this as ~12 elementwise kernels with intermediates; the [P,16,3] tensor alone is written by cat, then
// read + re-written by the add.  Here: ONE kernel for the 15 small floats per Gaussian and ONE for the SH rows
// (dc + rest + residual read once, shs written once), and one backward kernel for the small attributes -- the SH
// backward needs no kernel at all (d_dc / d_rest are slices of dL/dshs, d_residual is dL/dshs itself).
#pragma once
#include "gsrast_common.h"

namespace gsrast {

__global__ void __launch_bounds__(256)
epilogue_small_fwd_kernel(int P, const float* __restrict__ xyz, const float* __restrict__ motion_res,
                          const float* __restrict__ rotation, const float* __restrict__ rot_res /* [P][7] or null */,
                          const float* __restrict__ scaling, const float* __restrict__ opacity_logit,
                          const float* __restrict__ trbf, float* __restrict__ motion, float* __restrict__ rot,
                          float* __restrict__ scale, float* __restrict__ opacity)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
#pragma unroll
    for (int k = 0; k < 3; k++) motion[3 * (size_t)i + k] = xyz[3 * (size_t)i + k] + (motion_res ? motion_res[3 * (size_t)i + k] : 0.0f);
    float4 q = reinterpret_cast<const float4*>(rotation)[i];
    float s3[3] = { scaling[3 * (size_t)i], scaling[3 * (size_t)i + 1], scaling[3 * (size_t)i + 2] };
    if (rot_res) {
        const float* r = rot_res + 7 * (size_t)i;
        q.x += r[0]; q.y += r[1]; q.z += r[2]; q.w += r[3];
        s3[0] += r[4]; s3[1] += r[5]; s3[2] += r[6];
    }
    const float n = fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);     // F.normalize, eps 1e-12
    reinterpret_cast<float4*>(rot)[i] = make_float4(q.x / n, q.y / n, q.z / n, q.w / n);
#pragma unroll
    for (int k = 0; k < 3; k++) scale[3 * (size_t)i + k] = expf(s3[k]);
    const float sg = 1.0f / (1.0f + expf(-opacity_logit[i]));
    opacity[i] = trbf ? sg * trbf[i] : sg;
}

// shs[g][c] = (c < 3 ? dc[g][c] : rest[g][c - 3]) + res[g][c];  row = 3M floats, one float4 of output per lane
__global__ void __launch_bounds__(256)
epilogue_sh_fwd_kernel(size_t nq /* P * row / 4 */, int row, const float* __restrict__ dc, const float* __restrict__ rest,
                       const float* __restrict__ res, float* __restrict__ shs)
{
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const size_t e = q * 4;
    const size_t g = e / (size_t)row;
    const int c0 = (int)(e - g * (size_t)row);
    const int rrow = row - 3;
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int c = c0 + k;                        // row % 4 == 0: the four floats stay inside one row
        v[k] = c < 3 ? dc[g * 3 + c] : rest[g * (size_t)rrow + (c - 3)];
    }
    if (res) {
        const float4 r = reinterpret_cast<const float4*>(res)[q];
        v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
    }
    reinterpret_cast<float4*>(shs)[q] = make_float4(v[0], v[1], v[2], v[3]);
}
// general row length (not a multiple of 4 floats): one float per lane
__global__ void __launch_bounds__(256)
epilogue_sh_fwd_scalar_kernel(size_t n, int row, const float* __restrict__ dc, const float* __restrict__ rest,
                              const float* __restrict__ res, float* __restrict__ shs)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const size_t g = e / (size_t)row;
    const int c = (int)(e - g * (size_t)row);
    const float v = c < 3 ? dc[g * 3 + c] : rest[g * (size_t)(row - 3) + (c - 3)];
    shs[e] = res ? v + res[e] : v;
}

// Backward of the small attributes.  Upstream: d_rot [P][4], d_scale [P][3], d_opacity [P] (any may be null = zero).
//   d(rotation + res)  = (d_rot - rot * <rot, d_rot>) / max(|x|, eps)          (0 where |x| < eps, as autograd's clamp)
//   d(scaling + res)   = d_scale * scale
//   d(opacity logit)   = d_opacity * trbf * s (1 - s),   d(trbf) = d_opacity * s
// out: d_rotation [P][4], d_scaling [P][3], d_rot_res [P][7] (null when there is no residual), d_logit [P], d_trbf [P] (null ok)
__global__ void __launch_bounds__(256)
epilogue_small_bwd_kernel(int P, const float* __restrict__ rotation, const float* __restrict__ rot_res,
                          const float* __restrict__ scale /* forward output */, const float* __restrict__ opacity_logit,
                          const float* __restrict__ trbf, const float* __restrict__ d_rot, const float* __restrict__ d_scale,
                          const float* __restrict__ d_opacity, float* __restrict__ d_rotation, float* __restrict__ d_scaling,
                          float* __restrict__ d_rot_res, float* __restrict__ d_logit, float* __restrict__ d_trbf)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    float4 x = reinterpret_cast<const float4*>(rotation)[i];
    if (rot_res) { const float* r = rot_res + 7 * (size_t)i; x.x += r[0]; x.y += r[1]; x.z += r[2]; x.w += r[3]; }
    const float nn = sqrtf(x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w);
    float4 gq = make_float4(0.f, 0.f, 0.f, 0.f);
    if (d_rot) {
        const float4 g = reinterpret_cast<const float4*>(d_rot)[i];
        if (nn >= 1e-12f) {
            const float inv = 1.0f / nn;
            const float4 y = make_float4(x.x * inv, x.y * inv, x.z * inv, x.w * inv);
            const float dot = y.x * g.x + y.y * g.y + y.z * g.z + y.w * g.w;
            gq = make_float4((g.x - y.x * dot) * inv, (g.y - y.y * dot) * inv, (g.z - y.z * dot) * inv, (g.w - y.w * dot) * inv);
        } else {
            gq = make_float4(g.x * 1e12f, g.y * 1e12f, g.z * 1e12f, g.w * 1e12f);   // x / eps: the norm is clamped, not differentiated
        }
    }
    reinterpret_cast<float4*>(d_rotation)[i] = gq;
    float gs[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        gs[k] = d_scale ? d_scale[3 * (size_t)i + k] * scale[3 * (size_t)i + k] : 0.0f;
        d_scaling[3 * (size_t)i + k] = gs[k];
    }
    if (d_rot_res) {
        float* o = d_rot_res + 7 * (size_t)i;
        o[0] = gq.x; o[1] = gq.y; o[2] = gq.z; o[3] = gq.w; o[4] = gs[0]; o[5] = gs[1]; o[6] = gs[2];
    }
    const float s = 1.0f / (1.0f + expf(-opacity_logit[i]));
    const float go = d_opacity ? d_opacity[i] : 0.0f;
    const float tb = trbf ? trbf[i] : 1.0f;
    d_logit[i] = go * tb * s * (1.0f - s);
    if (d_trbf) d_trbf[i] = go * s;
}

} // namespace gsrast
