// gsrast_preprocess.h -- per-Gaussian kernels: forward projection / covariance / SH colour, and
// their backward.  One lane per Gaussian; fp32 arithmetic in a fixed evaluation order (the
// translation unit is built with -ffp-contract=off, FMAs appear only where written) so the
// integer outputs (radii, tile rectangles, depth key bits) match the CPU oracle bit for bit.
//
// Reference behaviour restated here (RST = reference cuda_rasterizer/):
//   forward : RST/forward.cu:155-256 preprocessCUDA, :74-113 computeCov2D, :118-152 computeCov3D,
//             :20-71 computeColorFromSH, RST/auxiliary.h:41-56 ndc2Pix/getRect, :139-164 in_frustum
//   backward: RST/backward.cu:144-274 computeCov2DCUDA, :346-396 preprocessCUDA,
//             :278-341 computeCov3D, :20-139 computeColorFromSH   (fused into ONE kernel here)
#pragma once
#include "gsrast_common.h"

namespace gsrast {

__device__ constexpr float kSH0 = 0.28209479177387814f;
__device__ constexpr float kSH1 = 0.4886025119029199f;
__device__ constexpr float kSH2[5] = { 1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                       -1.0925484305920792f, 0.5462742152960396f };
__device__ constexpr float kSH3[7] = { -0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                       0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                       -0.5900435899266435f };

struct Cam {            // per-call constants, passed by value in kernarg (SGPRs)
    float view[16];     // transposed storage: view[4*c + r] = V[r][c]
    float proj[16];
    float campos[3];
    float tanx, tany, fx, fy, scale_mod;
    int W, H, gx, gy;
};

// What the host passes: the three camera arrays stay in HBM (no D2H copy, no sync); every lane
// reads them through uniform (scalar) loads at kernel entry.
struct CamArgs {
    const float* view; const float* proj; const float* campos;
    float tanx, tany, fx, fy, scale_mod;
    int W, H, gx, gy;
};
__device__ __forceinline__ Cam load_cam(const CamArgs& a)
{
    Cam c;
#pragma unroll
    for (int k = 0; k < 16; k++) { c.view[k] = a.view[k]; c.proj[k] = a.proj[k]; }
    if (a.campos) { c.campos[0] = a.campos[0]; c.campos[1] = a.campos[1]; c.campos[2] = a.campos[2]; }
    else { c.campos[0] = c.campos[1] = c.campos[2] = 0.0f; }
    c.tanx = a.tanx; c.tany = a.tany; c.fx = a.fx; c.fy = a.fy; c.scale_mod = a.scale_mod;
    c.W = a.W; c.H = a.H; c.gx = a.gx; c.gy = a.gy;
    return c;
}

// ---- raw-parameter entry points (gsrast_forward_raw / gsrast_backward_raw) ------------------------------
// SaRO-GS hands the rasterizer ACTIVATED attributes: exp(_scaling), normalize(_rotation), sigmoid(_opacity) * trbf,
// cat(_features_dc, _features_rest) + residuals (scene/saro_gaussian.py:807-847, activations :39-47) -- a [P,16,3] tensor written
// by the model and re-read here, and the reverse in the backward.  With RAW = true the per-Gaussian kernels take the model's
// leaves (and the optional deformation residuals) themselves and apply the activations in registers; the backward applies the
// chain rule and writes the gradients of the leaves.  The expressions are epilogue_small_fwd / _bwd_kernel's (gsrast_epilogue.h),
// operation for operation, so the rasterizer sees bit-identical inputs either way.
struct RawArgs {
    const float* motion_res;     // [P][3] or null   motion = xyz + motion_res
    const float* rot_res;        // [P][7] or null   rot = normalize(rotation + [:, :4]), scale = exp(scaling + [:, 4:])
    const float* trbf;           // [P] or null      opacity = sigmoid(logit) * trbf
    const float* opacity_logit;  // [P]              (the backward needs it again; the forward gets it as `opacities`)
    const float* features_dc;    // [P][3]
    const float* features_rest;  // [P][(M-1)*3]     shs = cat(dc, rest) + shs_res
    const float* shs_res;        // [P][M*3] or null
};
struct RawGrads {
    float* d_rot_res;            // [P][7] or null
    float* d_trbf;               // [P] or null
    float* d_dc;                 // [P][3]           (may be null when d_shs_res is given: the caller then slices that)
    float* d_rest;               // [P][(M-1)*3]     ( " )
    float* d_shs_res;            // [P][M*3] or null
};
__device__ __forceinline__ void raw_mean(const RawArgs& r, int i, float p[3])
{
#pragma unroll
    for (int k = 0; k < 3; k++) p[k] = p[k] + (r.motion_res ? r.motion_res[3 * (size_t)i + k] : 0.0f);
}
// in: q = _rotation row, s = _scaling row.  out: q = normalised quaternion, s = exp(.), x = rotation + residual (what was normalised)
__device__ __forceinline__ void raw_rot_scale(const RawArgs& r, int i, float4& q, float s[3], float4& x)
{
    if (r.rot_res) {
        const float* rr = r.rot_res + 7 * (size_t)i;
        q.x += rr[0]; q.y += rr[1]; q.z += rr[2]; q.w += rr[3];
        s[0] += rr[4]; s[1] += rr[5]; s[2] += rr[6];
    }
    x = q;
    const float n = fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);     // F.normalize, eps 1e-12
    q = make_float4(q.x / n, q.y / n, q.z / n, q.w / n);
#pragma unroll
    for (int k = 0; k < 3; k++) s[k] = expf(s[k]);
}
__device__ __forceinline__ float raw_sigmoid(float logit) { return 1.0f / (1.0f + expf(-logit)); }

struct M3 { float m[3][3]; };  // m[c][r], column-major like the reference's glm::mat3

__device__ __forceinline__ void xform4x3(const float p[3], const float* M, float o[3])
{
#pragma unroll
    for (int r = 0; r < 3; r++) o[r] = M[r] * p[0] + M[4 + r] * p[1] + M[8 + r] * p[2] + M[12 + r];
}
__device__ __forceinline__ void xform4x4(const float p[3], const float* M, float o[4])
{
#pragma unroll
    for (int r = 0; r < 4; r++) o[r] = M[r] * p[0] + M[4 + r] * p[1] + M[8 + r] * p[2] + M[12 + r];
}
__device__ __forceinline__ float ndc2pix(float v, int S)
{ // evaluated in double like the reference (1.0 / 0.5 literals), then narrowed
    return (float)((((double)v + 1.0) * S - 1.0) * 0.5);
}
__device__ __forceinline__ void tile_rect(float px, float py, int rad, int gx, int gy, int rmin[2], int rmax[2])
{
    int a;
    a = (int)((px - rad) / TILE_X); a = a > 0 ? a : 0; rmin[0] = gx < a ? gx : a;
    a = (int)((py - rad) / TILE_Y); a = a > 0 ? a : 0; rmin[1] = gy < a ? gy : a;
    a = (int)((px + rad + TILE_X - 1) / TILE_X); a = a > 0 ? a : 0; rmax[0] = gx < a ? gx : a;
    a = (int)((py + rad + TILE_Y - 1) / TILE_Y); a = a > 0 ? a : 0; rmax[1] = gy < a ? gy : a;
}

__device__ __forceinline__ void quat_to_R(const float q[4], M3& R)
{
    const float r = q[0], x = q[1], y = q[2], z = q[3];
    R.m[0][0] = 1.0f - 2.0f * (y * y + z * z); R.m[0][1] = 2.0f * (x * y - r * z); R.m[0][2] = 2.0f * (x * z + r * y);
    R.m[1][0] = 2.0f * (x * y + r * z); R.m[1][1] = 1.0f - 2.0f * (x * x + z * z); R.m[1][2] = 2.0f * (y * z - r * x);
    R.m[2][0] = 2.0f * (x * z - r * y); R.m[2][1] = 2.0f * (y * z + r * x); R.m[2][2] = 1.0f - 2.0f * (x * x + y * y);
}
// Sigma = (S R)^T (S R), upper triangle; M = S*R returned for the backward.
__device__ __forceinline__ void cov3d_from_scale_rot(const float s_in[3], float mod, const float q[4], float c6[6], M3& M)
{
    M3 R; quat_to_R(q, R);
    const float s[3] = { mod * s_in[0], mod * s_in[1], mod * s_in[2] };
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int r = 0; r < 3; r++) M.m[c][r] = s[r] * R.m[c][r];
#define GS_SIG(c, r) (M.m[r][0] * M.m[c][0] + M.m[r][1] * M.m[c][1] + M.m[r][2] * M.m[c][2])
    c6[0] = GS_SIG(0, 0); c6[1] = GS_SIG(0, 1); c6[2] = GS_SIG(0, 2);
    c6[3] = GS_SIG(1, 1); c6[4] = GS_SIG(1, 2); c6[5] = GS_SIG(2, 2);
#undef GS_SIG
}

struct Cov2D { float t[3]; float txtz, tytz, limx, limy; M3 T, Wm, V; float a, b, c; };

__device__ __forceinline__ void cov2d_eval(const float mean[3], const Cam& cam, const float c6[6], Cov2D& o)
{
    float t[3];
    xform4x3(mean, cam.view, t);
    o.limx = 1.3f * cam.tanx; o.limy = 1.3f * cam.tany;
    o.txtz = t[0] / t[2]; o.tytz = t[1] / t[2];
    float cx = o.txtz < -o.limx ? -o.limx : o.txtz; cx = o.limx < cx ? o.limx : cx;
    float cy = o.tytz < -o.limy ? -o.limy : o.tytz; cy = o.limy < cy ? o.limy : cy;
    t[0] = cx * t[2]; t[1] = cy * t[2];
    o.t[0] = t[0]; o.t[1] = t[1]; o.t[2] = t[2];
    const float J00 = cam.fx / t[2], J02 = -(cam.fx * t[0]) / (t[2] * t[2]);
    const float J11 = cam.fy / t[2], J12 = -(cam.fy * t[1]) / (t[2] * t[2]);
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int r = 0; r < 3; r++) o.Wm.m[c][r] = cam.view[4 * r + c];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        o.T.m[0][r] = o.Wm.m[0][r] * J00 + o.Wm.m[2][r] * J02;
        o.T.m[1][r] = o.Wm.m[1][r] * J11 + o.Wm.m[2][r] * J12;
        o.T.m[2][r] = 0.0f;
    }
    o.V.m[0][0] = c6[0]; o.V.m[0][1] = c6[1]; o.V.m[0][2] = c6[2];
    o.V.m[1][0] = c6[1]; o.V.m[1][1] = c6[3]; o.V.m[1][2] = c6[4];
    o.V.m[2][0] = c6[2]; o.V.m[2][1] = c6[4]; o.V.m[2][2] = c6[5];
    float A[3][2];
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int r = 0; r < 2; r++)
            A[k][r] = o.T.m[r][0] * o.V.m[0][k] + o.T.m[r][1] * o.V.m[1][k] + o.T.m[r][2] * o.V.m[2][k];
    const float c00 = A[0][0] * o.T.m[0][0] + A[1][0] * o.T.m[0][1] + A[2][0] * o.T.m[0][2];
    const float c01 = A[0][1] * o.T.m[0][0] + A[1][1] * o.T.m[0][1] + A[2][1] * o.T.m[0][2];
    const float c11 = A[0][1] * o.T.m[1][0] + A[1][1] * o.T.m[1][1] + A[2][1] * o.T.m[1][2];
    o.a = c00 + 0.3f; o.b = c01; o.c = c11 + 0.3f;
}

// colour before clamping (SH + 0.5); sh points at this Gaussian's [M][3] block
__device__ __forceinline__ void sh_to_rgb(int deg, const float pos[3], const float campos[3], const float* __restrict__ sh, float out[3])
{
    const float d0 = pos[0] - campos[0], d1 = pos[1] - campos[1], d2 = pos[2] - campos[2];
    const float len = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
    const float x = d0 / len, y = d1 / len, z = d2 / len;
#pragma unroll
    for (int c = 0; c < 3; c++) {
#define SHC(k) sh[(k) * 3 + c]
        float res = kSH0 * SHC(0);
        if (deg > 0) {
            res = res - kSH1 * y * SHC(1) + kSH1 * z * SHC(2) - kSH1 * x * SHC(3);
            if (deg > 1) {
                const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                res = res + kSH2[0] * xy * SHC(4) + kSH2[1] * yz * SHC(5) +
                      kSH2[2] * (2.0f * zz - xx - yy) * SHC(6) + kSH2[3] * xz * SHC(7) +
                      kSH2[4] * (xx - yy) * SHC(8);
                if (deg > 2) {
                    res = res + kSH3[0] * y * (3.0f * xx - yy) * SHC(9) +
                          kSH3[1] * xy * z * SHC(10) +
                          kSH3[2] * y * (4.0f * zz - xx - yy) * SHC(11) +
                          kSH3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * SHC(12) +
                          kSH3[4] * x * (4.0f * zz - xx - yy) * SHC(13) +
                          kSH3[5] * z * (xx - yy) * SHC(14) +
                          kSH3[6] * x * (xx - 3.0f * yy) * SHC(15);
                }
            }
        }
#undef SHC
        out[c] = res + 0.5f;
    }
}


// d(colour)/d(view direction), 9 numbers {dx[c], dy[c], dz[c]} (backward.cu:78-127 dRGBdx / dRGBdy / dRGBdz): a function of the
// coefficients and the direction only -- of nothing the blend backward produces.  The forward's colour kernel evaluates it while the
// coefficient block is in LDS and stores the nine floats (36 B) (round 2, and still for a forward_only state: sh_dir_derivs_kernel on a
// side stream beside the blend backward); the per-Gaussian backward, which is on the critical path and bandwidth-bound, then reads
// those instead of the 12*M-byte coefficient block -- same expressions, same operands, same bits.
#define SHV(k, c) sh[(k) * 3 + (c)]
__device__ __forceinline__ void sh_dir_derivs(int deg, float x, float y, float z, const float* __restrict__ sh,
                                              float dx[3], float dy[3], float dz[3])
{
#pragma unroll
    for (int c = 0; c < 3; c++) { dx[c] = 0.0f; dy[c] = 0.0f; dz[c] = 0.0f; }
    if (deg > 0) {
#pragma unroll
        for (int c = 0; c < 3; c++) { dx[c] = -kSH1 * SHV(3, c); dy[c] = -kSH1 * SHV(1, c); dz[c] = kSH1 * SHV(2, c); }
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                dx[c] += kSH2[0] * y * SHV(4, c) + kSH2[2] * 2.0f * -x * SHV(6, c) + kSH2[3] * z * SHV(7, c) + kSH2[4] * 2.0f * x * SHV(8, c);
                dy[c] += kSH2[0] * x * SHV(4, c) + kSH2[1] * z * SHV(5, c) + kSH2[2] * 2.0f * -y * SHV(6, c) + kSH2[4] * 2.0f * -y * SHV(8, c);
                dz[c] += kSH2[1] * y * SHV(5, c) + kSH2[2] * 2.0f * 2.0f * z * SHV(6, c) + kSH2[3] * x * SHV(7, c);
            }
            if (deg > 2) {
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    dx[c] += (kSH3[0] * SHV(9, c) * 3.0f * 2.0f * xy + kSH3[1] * SHV(10, c) * yz +
                              kSH3[2] * SHV(11, c) * -2.0f * xy + kSH3[3] * SHV(12, c) * -3.0f * 2.0f * xz +
                              kSH3[4] * SHV(13, c) * (-3.0f * xx + 4.0f * zz - yy) +
                              kSH3[5] * SHV(14, c) * 2.0f * xz + kSH3[6] * SHV(15, c) * 3.0f * (xx - yy));
                    dy[c] += (kSH3[0] * SHV(9, c) * 3.0f * (xx - yy) + kSH3[1] * SHV(10, c) * xz +
                              kSH3[2] * SHV(11, c) * (-3.0f * yy + 4.0f * zz - xx) +
                              kSH3[3] * SHV(12, c) * -3.0f * 2.0f * yz + kSH3[4] * SHV(13, c) * -2.0f * xy +
                              kSH3[5] * SHV(14, c) * -2.0f * yz + kSH3[6] * SHV(15, c) * -3.0f * 2.0f * xy);
                    dz[c] += (kSH3[1] * SHV(10, c) * xy + kSH3[2] * SHV(11, c) * 4.0f * 2.0f * yz +
                              kSH3[3] * SHV(12, c) * 3.0f * (2.0f * zz - xx - yy) +
                              kSH3[4] * SHV(13, c) * 4.0f * 2.0f * xz + kSH3[5] * SHV(14, c) * (xx - yy));
                }
            }
        }
    }
}
#undef SHV

// -------------------------------------------------------------------------------------------
// SH coefficients are [P][M][3] AoS: one lane's block is M*12 bytes at a stride of M*12 bytes, the
// worst case for per-lane loads.  The workgroup therefore moves its PP_THREADS consecutive blocks
// with coalesced 16-byte loads into LDS (row stride M*3+1 floats: conflict-free for the per-lane
// reads that follow), and the backward writes dL/dsh back the same way.
constexpr int PP_THREADS = 128;
constexpr int PP_SH_MAX = 48;                   // (3+1)^2 coefficients x 3 channels
constexpr int PP_SH_STRIDE = PP_SH_MAX + 1;     // +1: conflict-free per-lane reads.  25 KB of LDS per workgroup = 6 workgroups per CU;
                                                // both kernels are latency-bound at that occupancy (4 per CU: +17 %, 3: +40 %)

#ifndef GSRAST_PB_GROUP
#define GSRAST_PB_GROUP 1024
#endif
constexpr int PB_GROUP = GSRAST_PB_GROUP;      // Gaussians per workgroup of the GROUPED per-Gaussian backward (preprocess_bwd_kernel): 512 ... 4096
// rows [off, off + len) of the LDS rows 0 .. cnt-1 to dst[ids[r]][0 .. len): the compacted Gaussians' rows, one by one (coalesced within a row)
__device__ __forceinline__ void scatter_sh_rows(float* __restrict__ dst, int rowlen, int off, int len, const uint32_t* ids, int cnt, const float* lds)
{
    if ((rowlen & 3) == 0 && (off & 3) == 0 && (len & 3) == 0 && ((uintptr_t)dst & 15) == 0) {
        const int r4 = len >> 2;
        for (int q = threadIdx.x; q < cnt * r4; q += PP_THREADS) {
            const int r = q / r4, c = (q - r * r4) * 4;
            const float* sp = lds + r * PP_SH_STRIDE + off + c;
            reinterpret_cast<float4*>(dst + (size_t)ids[r] * rowlen)[c >> 2] = make_float4(sp[0], sp[1], sp[2], sp[3]);
        }
    } else {
        for (int f = threadIdx.x; f < cnt * len; f += PP_THREADS) {
            const int r = f / len, c = f - r * len;
            dst[(size_t)ids[r] * rowlen + c] = lds[r * PP_SH_STRIDE + off + c];
        }
    }
}
__device__ __forceinline__ void stage_sh_out(float* __restrict__ dst_all, int P, int M, int base, const float* lds)
{
    const int ng = (P - base) < PP_THREADS ? (P - base) : PP_THREADS;
    const int row = M * 3;
    const int nfl = ng * row;
    float* dst = dst_all + (size_t)base * row;
    if ((row & 3) == 0 && ((uintptr_t)dst & 15) == 0) {
        float4* dst4 = reinterpret_cast<float4*>(dst);
        for (int q = threadIdx.x; q * 4 < nfl; q += PP_THREADS) {
            const int f = q * 4, g = f / row, c = f - g * row;
            const float* sp = lds + g * PP_SH_STRIDE + c;
            dst4[q] = make_float4(sp[0], sp[1], sp[2], sp[3]);
        }
    } else {
        for (int f = threadIdx.x; f < nfl; f += PP_THREADS) {
            const int g = f / row, c = f - g * row;
            dst[f] = lds[g * PP_SH_STRIDE + c];
        }
    }
}

// RAW: the rows of dL/dsh leave as the gradients of the model's two SH leaves, d_dc [P][3] and d_rest [P][row-3] (both blocks
// start 16-byte aligned: 128 rows), whole float4s coalesced, the last block's odd floats one by one.
__device__ __forceinline__ void stage_sh_out_split(float* __restrict__ d_dc, float* __restrict__ d_rest, int P, int row, int base, const float* lds)
{
    const int ng = (P - base) < PP_THREADS ? (P - base) : PP_THREADS;
    const int rrow = row - 3;
    if (rrow > 0) {
        const int nfr = ng * rrow, n4r = nfr >> 2;
        float* dst = d_rest + (size_t)base * rrow;
        float4* dst4 = reinterpret_cast<float4*>(dst);
        for (int q = threadIdx.x; q < n4r; q += PP_THREADS) {
            float e4[4];
#pragma unroll
            for (int e = 0; e < 4; e++) { const int f = q * 4 + e, g = f / rrow, c = f - g * rrow; e4[e] = lds[g * PP_SH_STRIDE + 3 + c]; }
            dst4[q] = make_float4(e4[0], e4[1], e4[2], e4[3]);
        }
        if ((int)threadIdx.x < (nfr & 3)) { const int f = n4r * 4 + threadIdx.x, g = f / rrow, c = f - g * rrow; dst[f] = lds[g * PP_SH_STRIDE + 3 + c]; }
    }
    {
        const int nfd = ng * 3, n4d = nfd >> 2;
        float* dst = d_dc + (size_t)base * 3;
        if ((int)threadIdx.x < n4d) {
            float e4[4];
#pragma unroll
            for (int e = 0; e < 4; e++) { const int f = threadIdx.x * 4 + e, g = f / 3, c = f - g * 3; e4[e] = lds[g * PP_SH_STRIDE + c]; }
            reinterpret_cast<float4*>(dst)[threadIdx.x] = make_float4(e4[0], e4[1], e4[2], e4[3]);
        }
        if ((int)threadIdx.x < (nfd & 3)) { const int f = n4d * 4 + threadIdx.x, g = f / 3, c = f - g * 3; dst[f] = lds[g * PP_SH_STRIDE + c]; }
    }
}

// -------------------------------------------------------------------------------------------
// K1 forward, colour half (forward.cu:20-71 computeColorFromSH, the colour lines of :237-246): SH -> RGB + clamp flags, or the
// caller's colors_precomp, into rec2.  It depends on nothing the geometry half produces -- so gsrast_forward runs it on a
// low-priority side stream, beside the geometry kernel, the depth sort and the binning (all latency-bound), and joins it in
// front of the blend, the first kernel to read a colour.  Evaluated for every Gaussian (a culled one is never read).
//
// Round 3: while a Gaussian's coefficient block sits in LDS the kernel also evaluates d(colour)/d(view direction)
// (backward.cu:78-127 dRGBdx / dy / dz, sh_dir_derivs above) and stores the nine floats (shdA / shdB / shdC, 36 B) for
// preprocess_bwd_kernel -- round 2 re-read the 12*M-byte blocks for that in the backward (sh_dir_derivs_kernel, as long as the
// blend backward itself at 3 M Gaussians).
// The block's rows enter LDS (row stride PP_SH_STRIDE floats) through coalesced 16-byte loads; loads are unconditional with
// clamped indices (a short last block re-reads its last 16 bytes) and all of a lane's loads are requested before the first LDS
// store.  ROW: floats per SH row known at compile time (48 for the reference's M = 16), 0 = runtime.
// RAW: the rows are cat(features_dc [P][3], features_rest [P][row-3]) + shs_res [P][row] (RawArgs), assembled in LDS: dc and rest are
// parked first, then the residual is added in place by the lane that loaded it (one add per element, like the model's `+`).
template <int ROW, bool RAW>
__device__ __forceinline__ void color_stage_rows(const float* __restrict__ shs, const RawArgs& raw, int P, int row_rt, int blk, float* lds)
{
    const int base = blk * PP_THREADS;
    const int ng = (P - base) < PP_THREADS ? (P - base) : PP_THREADS;
    const int row = ROW ? ROW : row_rt;
    const auto q_of = [&](int q, int n4, int) -> int { return q < n4 ? q : n4 - 1; };
    if (!RAW) {
        const int n4 = (ng * row) >> 2;
        const float4* src4 = reinterpret_cast<const float4*>(shs + (size_t)base * row);
        float4 v[PP_SH_MAX / 4];
#pragma unroll
        for (int u = 0; u < PP_SH_MAX / 4; u++) {
            const int q = threadIdx.x + u * PP_THREADS;
            // (non-temporal: 576 MB of SH rows at 3 M, each read once per forward -- they need not push the arrays that are read again out
            // of the Infinity Cache: first-seen-pose step +1.4 % views/s)
            typedef float v4f __attribute__((ext_vector_type(4)));
            const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(src4) + q_of(q, n4, row));
            v[u] = make_float4(t.x, t.y, t.z, t.w);
        }
#pragma unroll
        for (int u = 0; u < PP_SH_MAX / 4; u++) {
            const int q = threadIdx.x + u * PP_THREADS;
            if (q < n4) {
                const int f = q * 4, g = f / row, c = f - g * row;
                float* d = lds + g * PP_SH_STRIDE + c;
                d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
            }
        }
        return;
    }
    // RAW.  rest: ng * (row - 3) floats from a 16-byte aligned start (128 rows are a multiple of 16 bytes whatever the row length);
    // whole float4s with coalesced loads, the at most three floats behind them (last block only) one by one.
    const int rrow = row - 3;
    const int nfr = ng * rrow, n4r = nfr >> 2;
    const float* rsrc = raw.features_rest + (size_t)base * rrow;
    const float4* rsrc4 = reinterpret_cast<const float4*>(rsrc);
    const int nfd = ng * 3, n4d = nfd >> 2;
    const float* dsrc = raw.features_dc + (size_t)base * 3;
    const int n4s = (ng * row) >> 2;                               // residual: whole rows of a multiple of four floats
    const float4* ssrc4 = reinterpret_cast<const float4*>(raw.shs_res ? raw.shs_res + (size_t)base * row : raw.features_rest);
    float4 v[PP_SH_MAX / 4], vs[PP_SH_MAX / 4], vd;
#pragma unroll
    for (int u = 0; u < PP_SH_MAX / 4; u++) { const int q = threadIdx.x + u * PP_THREADS; v[u] = n4r > 0 ? rsrc4[q_of(q, n4r, rrow)] : make_float4(0.f, 0.f, 0.f, 0.f); }
    vd = n4d > 0 ? reinterpret_cast<const float4*>(dsrc)[q_of((int)threadIdx.x, n4d, 3)] : make_float4(0.f, 0.f, 0.f, 0.f);
    if (raw.shs_res) {
#pragma unroll
        for (int u = 0; u < PP_SH_MAX / 4; u++) { const int q = threadIdx.x + u * PP_THREADS; vs[u] = ssrc4[q_of(q, n4s, row)]; }
    }
    if (rrow > 0) {
#pragma unroll
        for (int u = 0; u < PP_SH_MAX / 4; u++) {
            const int q = threadIdx.x + u * PP_THREADS;
            if (q < n4r) {
                const float e4[4] = { v[u].x, v[u].y, v[u].z, v[u].w };
#pragma unroll
                for (int e = 0; e < 4; e++) { const int f = q * 4 + e, g = f / rrow, c = f - g * rrow; lds[g * PP_SH_STRIDE + 3 + c] = e4[e]; }
            }
        }
        if ((int)threadIdx.x < (nfr & 3)) { const int f = n4r * 4 + threadIdx.x, g = f / rrow, c = f - g * rrow; lds[g * PP_SH_STRIDE + 3 + c] = rsrc[f]; }
    }
    if ((int)threadIdx.x < n4d) {
        const float e4[4] = { vd.x, vd.y, vd.z, vd.w };
#pragma unroll
        for (int e = 0; e < 4; e++) { const int f = threadIdx.x * 4 + e, g = f / 3, c = f - g * 3; lds[g * PP_SH_STRIDE + c] = e4[e]; }
    }
    if ((int)threadIdx.x < (nfd & 3)) { const int f = n4d * 4 + threadIdx.x, g = f / 3, c = f - g * 3; lds[g * PP_SH_STRIDE + c] = dsrc[f]; }
    if (raw.shs_res) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PP_SH_MAX / 4; u++) {
            const int q = threadIdx.x + u * PP_THREADS;
            if (q < n4s) {
                const int f = q * 4, g = f / row, c = f - g * row;
                float* d = lds + g * PP_SH_STRIDE + c;
                d[0] = d[0] + vs[u].x; d[1] = d[1] + vs[u].y; d[2] = d[2] + vs[u].z; d[3] = d[3] + vs[u].w;
            }
        }
    }
}

// SH rows through LDS: M * 3 <= PP_SH_MAX floats per row, a multiple of four, 16-byte aligned bases (checked by the host).
// One workgroup per block of PP_THREADS Gaussians.  (Round 3 also measured PERSISTENT workgroups that request the next block's
// loads before they evaluate the current one -- 1536 / 1024 / 768 / 512 workgroups: the kernel alone stayed at 198-203 us for
// 3 M Gaussians, 80 us for 1 M, and beside the geometry kernel its 134 VGPRs x 3 waves per SIMD left that kernel no room
// (157 -> 264 us).  More bytes in flight do not help: ~37 MB are in flight either way.)
template <int ROW, bool RAW>
__global__ void __launch_bounds__(PP_THREADS)
preprocess_color_kernel(int P, int D, int M, const float* __restrict__ means3D, const float* __restrict__ shs, RawArgs raw,
                        const float* __restrict__ campos_dev,
                        float4* __restrict__ rec2, unsigned char* __restrict__ clamped,
                        float4* __restrict__ grec4 /* [P][4] or null: the backward's gradient records, zero-filled here when there is no side stream to do it */,
                        float4* __restrict__ shdA, float4* __restrict__ shdB, float* __restrict__ shdC /* null: no backward will follow (or D = 0) */,
                        const uint32_t* __restrict__ pred = nullptr /* list cut (gsrast_common.h): the predicated launch in front of the second blend */)
{
    __shared__ float sh_lds[PP_THREADS * PP_SH_STRIDE];
    if (pred && *pred == 0u) return;
    const int blk = blockIdx.x;
    const int i = blk * PP_THREADS + threadIdx.x;
    if (grec4) {   // this block's 128 records = 8 KB contiguous: four coalesced 16-byte stores per lane
        const size_t q0 = (size_t)blk * PP_THREADS * 4, q1 = (size_t)P * 4;
#pragma unroll
        for (int k = 0; k < 4; k++) { const size_t q = q0 + (size_t)k * PP_THREADS + threadIdx.x; if (q < q1) grec4[q] = make_float4(0.f, 0.f, 0.f, 0.f); }
    }
    const int ic = i < P ? i : P - 1;
    float p[3] = { means3D[3 * (size_t)ic], means3D[3 * (size_t)ic + 1], means3D[3 * (size_t)ic + 2] };       // requested before the staging barrier
    if (RAW) raw_mean(raw, ic, p);
    const float campos[3] = { campos_dev[0], campos_dev[1], campos_dev[2] };
    color_stage_rows<ROW, RAW>(shs, raw, P, M * 3, blk, sh_lds);
    __syncthreads();
    if (i < P) {
        const float* my_sh = sh_lds + threadIdx.x * PP_SH_STRIDE;
        float col[3];
        unsigned cl = 0;
        sh_to_rgb(D, p, campos, my_sh, col);
#pragma unroll
        for (int c = 0; c < 3; c++) { if (col[c] < 0.0f) { cl |= 1u << c; col[c] = 0.0f; } }
        if (shdA) {    // same expressions on the same operands as sh_dir_derivs_kernel: the same bits
            const float o0 = p[0] - campos[0], o1 = p[1] - campos[1], o2 = p[2] - campos[2];
            const float len = sqrtf(o0 * o0 + o1 * o1 + o2 * o2);
            const float x = o0 / len, y = o1 / len, z = o2 / len;
            float dx[3], dy[3], dz[3];
            sh_dir_derivs(D, x, y, z, my_sh, dx, dy, dz);
            shdA[i] = make_float4(dx[0], dx[1], dx[2], dy[0]);
            shdB[i] = make_float4(dy[1], dy[2], dz[0], dz[1]);
            shdC[i] = dz[2];
        }
        rec2[(size_t)REC_STRIDE * i] = make_float4(col[0], col[1], col[2], 0.0f);
        clamped[i] = (unsigned char)cl;
    }
}

// The same without LDS staging: colors_precomp, or SH rows the staged kernel does not take (more than 16 coefficients, rows
// that are not a multiple of 16 bytes, an unaligned base).  One lane per Gaussian, rows read in place.
__global__ void __launch_bounds__(256)
preprocess_color_direct_kernel(int P, int D, int M, const float* __restrict__ means3D, const float* __restrict__ shs,
                               const float* __restrict__ colors_precomp, const float* __restrict__ campos_dev,
                               float4* __restrict__ rec2, unsigned char* __restrict__ clamped, float4* __restrict__ grec4,
                               float4* __restrict__ shdA, float4* __restrict__ shdB, float* __restrict__ shdC,
                               const uint32_t* __restrict__ pred = nullptr /* list cut: the predicated launch in front of the second blend */)
{
    if (pred && *pred == 0u) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    if (grec4) {
#pragma unroll
        for (int k = 0; k < 4; k++) grec4[4 * (size_t)i + k] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float col[3];
    unsigned cl = 0;
    if (!colors_precomp) {
        const float p[3] = { means3D[3 * (size_t)i], means3D[3 * (size_t)i + 1], means3D[3 * (size_t)i + 2] };
        const float campos[3] = { campos_dev[0], campos_dev[1], campos_dev[2] };
        const float* my_sh = shs + (size_t)i * M * 3;
        sh_to_rgb(D, p, campos, my_sh, col);
#pragma unroll
        for (int c = 0; c < 3; c++) { if (col[c] < 0.0f) { cl |= 1u << c; col[c] = 0.0f; } }
        if (shdA) {
            const float o0 = p[0] - campos[0], o1 = p[1] - campos[1], o2 = p[2] - campos[2];
            const float len = sqrtf(o0 * o0 + o1 * o1 + o2 * o2);
            const float x = o0 / len, y = o1 / len, z = o2 / len;
            float dx[3], dy[3], dz[3];
            sh_dir_derivs(D, x, y, z, my_sh, dx, dy, dz);
            shdA[i] = make_float4(dx[0], dx[1], dx[2], dy[0]);
            shdB[i] = make_float4(dy[1], dy[2], dz[0], dz[1]);
            shdC[i] = dz[2];
        }
    } else {
#pragma unroll
        for (int c = 0; c < 3; c++) col[c] = colors_precomp[3 * (size_t)i + c];
    }
    rec2[(size_t)REC_STRIDE * i] = make_float4(col[0], col[1], col[2], 0.0f);
    clamped[i] = (unsigned char)cl;
}

// The colour half under the LIST CUT (gsrast_common.h): bit i of skip marks a Gaussian the bucket scatter found culled or LATE -- it is
// in no list, nobody reads its colour, and its gradient record stays zero, so the backward does not read its direction derivatives
// either: neither fetched nor evaluated (3 M cube: 87 % of them).  A workgroup takes 1024 consecutive Gaussians, compacts the ones
// to evaluate (ascending index order) and works through them 64 per wave: their rows are GATHERED into LDS with coalesced loads
// (consecutive lanes read consecutive floats of a row) and evaluated from there like preprocess_color_kernel does -- the same
// expressions on the same operands, the same bits.  Measured on the way here, 3 M cube, 0.3-0.4 M rows to evaluate: the staged kernel
// with skipped rows' loads redirected: 138 us (it still runs every instruction, and its 25 KB x 6 workgroups fill a CU's LDS: the run
// sort's scatter beside it 109 us instead of 15); one lane per Gaussian with masked lanes: 95 us (as many load instructions as before,
// few lanes each); the rows gathered in DEPTH order -- any order, in a random scene -- from the depth buckets: 140-620 us in three
// forms (TLB reach, not bytes: the gathers here walk the arrays in address order).
// (Measured and dropped: each compacted lane reading its own row into registers with 16-byte loads, no LDS -- the colour kernel 108 ->
// 122 us, the step -4 % at 3 M and 1 M: the rows of a wave are ~8 rows apart, every 64-byte sector is fetched four times.)
// RAW: the row is cat(features_dc, features_rest) + shs_res, assembled in LDS as preprocess_color_kernel<., true> does.
#ifndef GSRAST_PCC_WAVES
#define GSRAST_PCC_WAVES 2        // waves per workgroup (512 Gaussians each): 1 / 2 / 4 measured 747 / 761-767 / 765-769 views/s at 3 M, 1219 / 1220 / 1209 at 1 M
#endif
constexpr int PCC_WAVES = GSRAST_PCC_WAVES, PCC_IDS = 512 * PCC_WAVES;
__device__ __forceinline__ void pcc_wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
template <bool RAW>
__global__ void __launch_bounds__(64 * PCC_WAVES)
preprocess_color_compact_kernel(int P, int D, int M, const float* __restrict__ means3D, const float* __restrict__ shs,
                                const float* __restrict__ colors_precomp, RawArgs raw, const float* __restrict__ campos_dev,
                                float4* __restrict__ rec2, unsigned char* __restrict__ clamped,
                                float4* __restrict__ shdA, float4* __restrict__ shdB, float* __restrict__ shdC,
                                const unsigned char* __restrict__ skip /* bit i = skip Gaussian i; the words past P read as written by the scatter (all set) */,
                                const uint32_t* __restrict__ pred = nullptr /* or: a launch of the list cut's completion pass (returns unless *pred != 0) */)
{
    if (pred && *pred == 0u) return;
    constexpr int PER = PCC_IDS / (64 * PCC_WAVES);              // consecutive Gaussians per lane (8)
    static_assert(PER == 8, "one byte of flags per lane");
    __shared__ float s_rows[PCC_WAVES][64 * PP_SH_STRIDE];
    __shared__ uint32_t s_list[PCC_IDS];
    __shared__ uint32_t s_wtot[PCC_WAVES];
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t base = blockIdx.x * (uint32_t)PCC_IDS + threadIdx.x * (uint32_t)PER;
    // 1. compact the indices to evaluate, in ascending order
    uint32_t fl = 0xFFu;                                         // eight flag bits; past P: skipped
    if (base < (uint32_t)P) {
        fl = skip[base >> 3];
        for (int k = 0; k < PER; k++) if (base + k >= (uint32_t)P) fl |= 1u << k;
    }
    const uint32_t mine = (uint32_t)__builtin_popcount(~fl & 0xFFu);
    uint32_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d, 64); if (lane >= (unsigned)d) incl += o; }
    if (lane == 63u) s_wtot[wave] = incl;
    __syncthreads();
    uint32_t pos = incl - mine, total = 0;
#pragma unroll
    for (int w = 0; w < PCC_WAVES; w++) { if ((unsigned)w < wave) pos += s_wtot[w]; total += s_wtot[w]; }
#pragma unroll
    for (int k = 0; k < PER; k++) if (!((fl >> k) & 1u)) s_list[pos++] = base + k;
    __syncthreads();
    if (total == 0) return;
    // 2. evaluate them, 64 per wave
    const float campos[3] = { campos_dev ? campos_dev[0] : 0.f, campos_dev ? campos_dev[1] : 0.f, campos_dev ? campos_dev[2] : 0.f };
    const int row = M * 3;
    float* lds = s_rows[wave];
    for (uint32_t j0 = wave * 64u; j0 < total; j0 += 64u * PCC_WAVES) {
        const uint32_t cnt = (total - j0) < 64u ? (total - j0) : 64u;
        const uint32_t* ids = s_list + j0;
        const uint32_t i = lane < cnt ? ids[lane] : 0u;
        float col[3] = { 0.f, 0.f, 0.f };
        unsigned cl = 0;
        if (colors_precomp) {
            if (lane < cnt) {
#pragma unroll
                for (int c = 0; c < 3; c++) col[c] = colors_precomp[3 * (size_t)i + c];
            }
        } else {
            float p[3] = { 0.f, 0.f, 0.f };
            if (lane < cnt) { p[0] = means3D[3 * (size_t)i]; p[1] = means3D[3 * (size_t)i + 1]; p[2] = means3D[3 * (size_t)i + 2]; if (RAW) raw_mean(raw, (int)i, p); }
            // gather: element f of the cnt x len block -> row f / len, float f % len
            const auto gather = [&](const float* __restrict__ src, int len, int off, bool add) {
                const uint32_t tot = cnt * (uint32_t)len;
                for (uint32_t f0 = 0; f0 < tot; f0 += 64u * 8u) {
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const uint32_t f = f0 + u * 64u + lane;
                        const uint32_t g = f / (uint32_t)len, e = f - g * (uint32_t)len;
                        v[u] = f < tot ? src[(size_t)ids[g] * len + e] : 0.0f;
                    }
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const uint32_t f = f0 + u * 64u + lane;
                        const uint32_t g = f / (uint32_t)len, e = f - g * (uint32_t)len;
                        if (f < tot) { float* d = lds + g * PP_SH_STRIDE + off + e; *d = add ? *d + v[u] : v[u]; }
                    }
                }
            };
            const float* my_sh;
            if (row <= PP_SH_MAX) {
                if (RAW) {
                    gather(raw.features_dc, 3, 0, false);
                    if (row > 3) gather(raw.features_rest, row - 3, 3, false);
                    if (raw.shs_res) { pcc_wave_sync(); gather(raw.shs_res, row, 0, true); }
                } else gather(shs, row, 0, false);
                pcc_wave_sync();
                my_sh = lds + lane * PP_SH_STRIDE;
            } else my_sh = shs + (size_t)i * row;          // more coefficients than the staging holds (never RAW): rows in place
            if (lane < cnt) {
                sh_to_rgb(D, p, campos, my_sh, col);
#pragma unroll
                for (int c = 0; c < 3; c++) { if (col[c] < 0.0f) { cl |= 1u << c; col[c] = 0.0f; } }
                if (shdA) {
                    const float o0 = p[0] - campos[0], o1 = p[1] - campos[1], o2 = p[2] - campos[2];
                    const float len = sqrtf(o0 * o0 + o1 * o1 + o2 * o2);
                    const float x = o0 / len, y = o1 / len, z = o2 / len;
                    float dx[3], dy[3], dz[3];
                    sh_dir_derivs(D, x, y, z, my_sh, dx, dy, dz);
                    shdA[i] = make_float4(dx[0], dx[1], dx[2], dy[0]);
                    shdB[i] = make_float4(dy[1], dy[2], dz[0], dz[1]);
                    shdC[i] = dz[2];
                }
            }
            pcc_wave_sync();                               // the next round overwrites the rows
        }
        if (lane < cnt) {
            rec2[(size_t)REC_STRIDE * i] = make_float4(col[0], col[1], col[2], 0.0f);
            clamped[i] = (unsigned char)cl;
        }
    }
}

// K1 forward, geometry half.  Writes radii / tiles / rect for every Gaussian, the rest only for visible ones.
#ifndef GSRAST_PF_THREADS
#define GSRAST_PF_THREADS 256      // (1024-thread blocks shorten the depth-range reduction of the bucket scatter by 2 us, but beside the colour kernel they wait for whole-CU wave slots: 43 -> 107 us)
#endif
constexpr int PF_THREADS = GSRAST_PF_THREADS;
// RAW (gsrast_forward_raw): means3D / scales / rotations / opacities are the model's _xyz / _scaling / _rotation / _opacity leaves
// and `raw` carries the optional residuals; the activations happen right behind the loads (RawArgs above).
template <bool RAW>
__global__ void __launch_bounds__(PF_THREADS)
preprocess_fwd_kernel(int P, const float* __restrict__ means3D, const float* __restrict__ scales,
                      const float* __restrict__ rotations, const float* __restrict__ opacities, RawArgs raw,
                      const float* __restrict__ cov3D_precomp, CamArgs cam_args, int* __restrict__ radii,
                      float4* __restrict__ rec0, float4* __restrict__ rec1,
                      float* __restrict__ cov3D /* or null */,
                      uint32_t* __restrict__ tiles, uint2* __restrict__ rect, float4* __restrict__ binrec,
                      uint32_t* __restrict__ sort_key, uint32_t* __restrict__ sort_val /* or null */,
                      int clip_rect /* run-compressed binning with tile_clip: rect / binrec get the clipped rectangle */,
                      uint32_t* __restrict__ bucket_cnt /* [8 + 1][64] work-bucket counters of this call: zeroed here */,
                      uint32_t* __restrict__ zhist /* [ZH_COPIES][ZH_BINS] or null: sampled histogram of the visible depth keys, for the bucket depth sort
                                                      (gsrast_common.h: equalised buckets; zeroed by a memset in front of this kernel) */,
                      uint32_t zh_klo /* kmid */, int zh_shift, uint32_t zh_wave_mask /* wave w of the grid is sampled iff (hash(w) & mask) == 0 */,
                      uint32_t* __restrict__ zero_words, int n_zero_words /* that sort's counters: zeroed here, spread over the blocks */,
                      HintTable* __restrict__ hints /* or null: the context's launch-order hints of the forward blend (gsrast_common.h) */,
                      uint32_t* __restrict__ hint_sel /* [2]: this call's slot and whether it held this pose already */,
                      uint32_t* __restrict__ zcut_used /* or null; [ntiles_img]: this call's snapshot of the pose's cut depths (list cut, gsrast_common.h) */,
                      uint32_t ntiles_img, uint32_t* __restrict__ cut_scalars /* or null: GeomLayout::scalars, whose `undone` counter is zeroed here */,
                      unsigned long long* __restrict__ host_found = nullptr, uint32_t host_seq = 0 /* pinned host word: {this pose was in the table, the
                                                             call's sequence number} -- the host sizes the launches over the cut lists by it */,
                      int borrow = 0 /* r > 0: a pose the table does not know takes the estimates and cut depths of a near pose's slot (HintTable::cam), the cut depths widened over (2 r + 1)^2 tiles */,
                      float near_scale2 = 0.0f /* (camera-to-scene distance)^2 the near-pose tolerance is relative to; 0: the camera's distance from the origin */,
                      uint32_t* __restrict__ prefilter_violation = nullptr /* or a word that is set when a Gaussian is culled although the caller said `prefiltered` (auxiliary.h:156-160) */,
                      unsigned char* __restrict__ untouched = nullptr /* or GeomLayout::untouched: every byte set here, cleared by the forward blend */,
                      uint32_t* __restrict__ tau_hist = nullptr /* or ImgLayout::tau_hist [TAU_COPIES][ntiles_img][TAU_BINS] (zeroed by a memset): the predicted cut's opacity mass */,
                      TauBins tau_bins = TauBins{0u, 0.0f, 0.0f, 0u})
{
    if (blockIdx.x == 0) for (int i = threadIdx.x; i < (XCD_GROUPS + 1) * WORK_BUCKETS + GATE_WORDS; i += blockDim.x) bucket_cnt[i] = 0u;
    if (cut_scalars && blockIdx.x == 0 && threadIdx.x == 0) { cut_scalars[SC_UNDONE] = 0u; cut_scalars[SC_N_LATE] = 0u; cut_scalars[SC_GATE_COUNT] = 0u; cut_scalars[SC_TOUCH_VALID] = untouched ? 1u : 0u; cut_scalars[SC_GREC_SPARSE] = 0u; }      // (always: the backward reads them)
    // The camera pose's key: block 0 looks it up and claims its slot, or the least recently used one; the first blocks of the grid
    // look it up too and copy the slot's cut depths into this call's own image buffer -- the bucket scatter and the forward blend
    // must see the SAME values, whatever another forward of this context writes into the table meanwhile (a stale or torn snapshot
    // is only a poorer speculation: it is verified against itself).
    constexpr int SNAP_BLOCKS = 32;
    const bool snap = hints && zcut_used && blockIdx.x < (unsigned)SNAP_BLOCKS;
    if (hints && (blockIdx.x == 0 || snap)) {
        __shared__ int s_slot;
        __shared__ unsigned long long s_lru;
        if (threadIdx.x == 0) { s_slot = -1; s_lru = ~0ull; }
        __syncthreads();
        uint32_t h0 = 2166136261u, h1 = 0x9E3779B9u;                // (uniform: every lane hashes the same 32 words)
        for (int k = 0; k < 16; k++) {
            const uint32_t a = __float_as_uint(cam_args.view[k]), b = __float_as_uint(cam_args.proj[k]);
            h0 = (h0 ^ a) * 16777619u; h0 = (h0 ^ b) * 16777619u;
            h1 = (h1 + a) * 0x85EBCA6Bu; h1 ^= h1 >> 13; h1 = (h1 + b) * 0xC2B2AE35u; h1 ^= h1 >> 16;
        }
        h0 = (h0 ^ (uint32_t)cam_args.W) * 16777619u; h1 = (h1 + (uint32_t)cam_args.H) * 0x85EBCA6Bu;
        h0 |= 1u;                                                   // (0, 0) means "free"
        // this camera: position, viewing direction (third row of the world-to-view rotation; the matrices are stored transposed)
        const float cpx = cam_args.campos[0], cpy = cam_args.campos[1], cpz = cam_args.campos[2];
        const float fwx = cam_args.view[2], fwy = cam_args.view[6], fwz = cam_args.view[10];
        __shared__ unsigned long long s_near;
        if (threadIdx.x == 0) s_near = 0ull;
        __syncthreads();
        for (int k = threadIdx.x; k < HINT_SLOTS; k += blockDim.x) {        // one lane per slot
            const bool used = (hints->key[k][0] | hints->key[k][1]) != 0u;
            if (hints->key[k][0] == h0 && hints->key[k][1] == h1) s_slot = k;
            if (blockIdx.x == 0) atomicMin(&s_lru, ((unsigned long long)hints->stamp[k] << 32) | (unsigned long long)k);      // least recently used, lowest index first (block 0 alone reads and writes the stamps)
            if (used && borrow) {
                // a NEAR pose (a camera path's previous frame): within 12 % of the distance to the world origin and 12 degrees of the viewing
                // direction; the closest direction wins, the lower index on a tie.  (Lookup blocks that read a slot while block 0 rewrites
                // it may decide differently: a tile's snapshot then comes from another slot -- only a poorer speculation, verified like any.)
                const float* c = hints->cam[k];
                const float dx = c[0] - cpx, dy = c[1] - cpy, dz = c[2] - cpz;
                // (the tolerance is relative to the camera's distance from the SCENE -- the middle of the depth range this context has
                // learned from its forwards, near_scale2 --, not from the world origin, which a scene need not be centred on; a context's
                // first forwards have learned nothing yet and fall back to the distance from the origin)
                const float d2 = dx * dx + dy * dy + dz * dz, r2 = near_scale2 > 0.0f ? near_scale2 : fmaxf(cpx * cpx + cpy * cpy + cpz * cpz, 1e-12f);
                const float dot = c[3] * fwx + c[4] * fwy + c[5] * fwz;
                if (d2 <= 0.0144f * r2 && dot >= 0.978f)
                    atomicMax(&s_near, ((unsigned long long)__float_as_uint(dot) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)k));
            }
        }
        __syncthreads();
        const int near_slot = (s_slot < 0 && s_near != 0ull) ? (int)(0xFFFFFFFFu - (uint32_t)(s_near & 0xFFFFFFFFull)) : -1;
        if (threadIdx.x == 0 && blockIdx.x == 0) {
            int slot = s_slot;
            const uint32_t now = hints->clock + 1u;
            const uint32_t found = slot >= 0 ? 1u : (near_slot >= 0 ? 2u : 0u);
            // A pose the table does not hold takes the least recently used slot.  Its key and camera are NOT written here: this kernel's
            // other lookup blocks are reading the table right now (round 4 left that race in as benign); they travel in the call's own
            // scalars (HINT_PUB) and the forward blend -- a later kernel of the same stream -- publishes them with the slot's new contents.
            if (slot < 0) slot = (int)(uint32_t)(s_lru & 0xFFFFFFFFull);
            hints->stamp[slot] = now; hints->clock = now;
            uint32_t* pub = hint_sel + (HINT_PUB - HINT_SEL);
            pub[0] = found == 1u ? 0u : 1u; pub[1] = h0; pub[2] = h1;
            pub[3] = __float_as_uint(cpx); pub[4] = __float_as_uint(cpy); pub[5] = __float_as_uint(cpz);
            pub[6] = __float_as_uint(fwx); pub[7] = __float_as_uint(fwy); pub[8] = __float_as_uint(fwz);
            hint_sel[0] = (uint32_t)slot; hint_sel[1] = found; hint_sel[2] = found == 2u ? (uint32_t)near_slot : (uint32_t)slot;
            if (host_found) __hip_atomic_store(host_found, ((unsigned long long)host_seq << 32) | (unsigned long long)found, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (snap) {
            const int slot = s_slot >= 0 ? s_slot : near_slot;
            const uint32_t* zc = hint_zcut(hints, ntiles_img) + (size_t)(slot < 0 ? 0 : slot) * ntiles_img;
            const uint32_t nsb = gridDim.x < (unsigned)SNAP_BLOCKS ? gridDim.x : (unsigned)SNAP_BLOCKS;
            // (a BORROWED slot is another camera's: what a tile sees there, a tile a few columns or rows away sees here -- the cut depth is
            // the deepest of the (2 borrow + 1)^2 tiles around it, none if one of them has none)
            const int rad = s_slot >= 0 ? 0 : borrow, gxt = cam_args.gx, gyt = (int)ntiles_img / cam_args.gx;
            for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < ntiles_img; t += nsb * blockDim.x) {
                uint32_t z = slot < 0 ? ZCUT_NONE : zc[t];
                if (slot >= 0 && rad > 0) {
                    const int tx = (int)t % gxt, ty = (int)t / gxt;
                    for (int y = max(ty - rad, 0); y <= min(ty + rad, gyt - 1); y++)
                        for (int x = max(tx - rad, 0); x <= min(tx + rad, gxt - 1); x++) z = max(z, zc[y * gxt + x]);
                }
                zcut_used[t] = z;
            }
        }
    }
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n_zero_words; k += gridDim.x * blockDim.x) zero_words[k] = 0u;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (untouched && i < P) untouched[i] = 1;      // (the blend clears what it consumes)
    // Every per-Gaussian input is requested up front: loads issued where they are first used (inside the visibility / area
    // branches) put three more memory round trips into a latency-bound kernel.  Clamped index: lanes past P load a valid
    // element and never use it.
    const int ic = i < P ? i : P - 1;
    float p[3] = { means3D[3 * ic], means3D[3 * ic + 1], means3D[3 * ic + 2] };
    float s_in[3] = { 0.f, 0.f, 0.f };
    float4 q_in = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!cov3D_precomp) {                                   // uniform
        s_in[0] = scales[3 * ic]; s_in[1] = scales[3 * ic + 1]; s_in[2] = scales[3 * ic + 2];
        q_in = reinterpret_cast<const float4*>(rotations)[ic];
    }
    float op_in = opacities[ic];
    if (RAW) {
        raw_mean(raw, ic, p);
        float4 x_unused;
        raw_rot_scale(raw, ic, q_in, s_in, x_unused);
        const float sg = raw_sigmoid(op_in);
        op_in = raw.trbf ? sg * raw.trbf[ic] : sg;
    }
    const Cam cam = load_cam(cam_args);
    int rad_out = 0; uint32_t ntiles = 0; uint32_t key = 0xFFFFFFFFu; uint2 rc = make_uint2(0u, 0u);
    float tau_mass = 0.0f; int tau_tile = -1;          // predicted cut (gsrast_common.h): this Gaussian's opacity mass and the tile of its centre
    if (i < P) {

    float ph[4], pv[3];
    xform4x4(p, cam.proj, ph);
    const float pw = 1.0f / (ph[3] + 0.0000001f);
    const float pp0 = ph[0] * pw, pp1 = ph[1] * pw;
    xform4x3(p, cam.view, pv);
    if (pv[2] > 0.2f) {
        float c6[6];
        if (cov3D_precomp) {
#pragma unroll
            for (int k = 0; k < 6; k++) c6[k] = cov3D_precomp[6 * (size_t)i + k];
        } else {
            const float q[4] = { q_in.x, q_in.y, q_in.z, q_in.w };
            M3 Mm;
            cov3d_from_scale_rot(s_in, cam.scale_mod, q, c6, Mm);
            if (cov3D) {       // only for gsrast_debug_export (option "debug_state"): the backward recomputes it from scale / rotation
#pragma unroll
                for (int k = 0; k < 6; k++) cov3D[6 * (size_t)i + k] = c6[k];
            }
        }
        Cov2D cv;
        cov2d_eval(p, cam, c6, cv);
        const float det = cv.a * cv.c - cv.b * cv.b;
        if (det != 0.0f) {
            const float det_inv = 1.0f / det;
            const float con0 = cv.c * det_inv, con1 = -cv.b * det_inv, con2 = cv.a * det_inv;
            const float mid = 0.5f * (cv.a + cv.c);
            float disc = mid * mid - det; disc = disc < 0.1f ? 0.1f : disc;
            const float l1 = mid + sqrtf(disc), l2 = mid - sqrtf(disc);
            const float lmax = l1 < l2 ? l2 : l1;
            const float my_radius = ceilf(3.0f * sqrtf(lmax));
            const float px = ndc2pix(pp0, cam.W), py = ndc2pix(pp1, cam.H);
            const int rad = (int)my_radius;
            int rmin[2], rmax[2];
            tile_rect(px, py, rad, cam.gx, cam.gy, rmin, rmax);
            const int area = (rmax[0] - rmin[0]) * (rmax[1] - rmin[1]);
            if (area != 0) {
                const float op = op_in;
                // Conservative pre-test for the blend kernels: power < thr  ==>  op*exp(power) < 1/255
                // with a 2% margin, so skipping the exp for such pairs never changes a decision.
                // (clamped at -80 so that exp() is only ever evaluated on [-80, 0]: gs_exp<., BOUNDED>)
                const float thr = op > 0.0f ? fmaxf(logf(1.0f / (255.0f * op)) - 0.02f, -80.0f) : 1.0f;
                rec0[(size_t)REC_STRIDE * i] = make_float4(px, py, con0, con1);      // (non-temporal stores here, measured: the kernel +10 us, its readers -12: nothing)
                rec1[(size_t)REC_STRIDE * i] = make_float4(con2, op, pv[2], thr);
                rad_out = rad; ntiles = (uint32_t)area;
                key = __float_as_uint(pv[2]);
                if (tau_hist && px >= 0.0f && py >= 0.0f && px < (float)cam.W && py < (float)cam.H) {
                    tau_tile = ((int)py >> 4) * cam.gx + ((int)px >> 4);
                    // the Gaussian's OPTICAL-DEPTH mass: integral over the plane of -ln(1 - a(x)), a(x) = o exp(-q(x) / 2)  =  2 pi sqrt(det cov2D) Li2(o),
                    // Li2 the dilogarithm, here its series cut after six terms (a lower bound: the estimate stays conservative)
                    const float oc = fminf(op, 0.99f);
                    const float li2 = oc * (1.0f + oc * (0.25f + oc * (0.111111f + oc * (0.0625f + oc * (0.04f + oc * 0.0277778f)))));
                    // ... but never more than it can lay on ONE tile (its peak over all 256 pixels): a Gaussian wider than a tile hands the tile of
                    // its centre that much and its neighbours nothing, which only lowers their estimates
                    tau_mass = fminf(li2 * 6.28318531f * sqrtf(fmaxf(det, 0.0f)), -logf(1.0f - oc) * 256.0f);
                }
                if (clip_rect) {
                    // Bounding box of the ellipse the alpha >= 1/255 pixels lie in (same construction and margins as the
                    // per-column clipping in emit_column_runs_kernel, gsrast_binning.h): whole tile columns / rows of the
                    // 3-sigma square that it cannot reach produce no column runs at all.  tiles[] (-> num_rendered) keeps
                    // the reference's count.
                    const float DX = fmaxf(fabsf(px - 16.0f * (float)rmin[0]), fabsf(16.0f * (float)rmax[0] - px));
                    const float DY = fmaxf(fabsf(py - 16.0f * (float)rmin[1]), fabsf(16.0f * (float)rmax[1] - py));
                    const float Mb = (fabsf(con0) + fabsf(con1)) * DX * DX + (fabsf(con2) + fabsf(con1)) * DY * DY;
                    const double t = (double)(-thr + 1e-6f * Mb);
                    const double a = (double)con0, b = (double)con1, c = (double)con2;
                    const double dd = a * c - b * b;
                    if (!(t >= 0.0)) { rmax[0] = rmin[0]; rmax[1] = rmin[1]; }
                    else if (dd > 0.0 && a > 0.0 && c > 0.0) {
                        const double ex[2] = { sqrt(2.0 * t * c / dd), sqrt(2.0 * t * a / dd) };   // half extents in x, y
                        const double ctr[2] = { (double)px, (double)py };
                        const int lim[2] = { cam.W - 1, cam.H - 1 };
#pragma unroll
                        for (int ax = 0; ax < 2; ax++) {
                            if (!(ex[ax] < 1e30)) continue;
                            const double plo = ceil(ctr[ax] - ex[ax] - 1e-3), phi = floor(ctr[ax] + ex[ax] + 1e-3);
                            const int tlo = max(rmin[ax], (int)fmax(plo, 0.0) >> 4);
                            const int thi = min(rmax[ax] - 1, (int)fmin(phi, (double)lim[ax]) >> 4);
                            if (phi < 0.0 || plo > (double)lim[ax] || thi < tlo) { rmax[0] = rmin[0]; rmax[1] = rmin[1]; }
                            else { rmin[ax] = tlo; rmax[ax] = thi + 1; }
                        }
                    }
                }
                rc = make_uint2((uint32_t)rmin[0] | ((uint32_t)rmin[1] << 16), (uint32_t)rmax[0] | ((uint32_t)rmax[1] << 16));
                // everything the binning needs about this Gaussian in ONE 32-byte record (it is gathered in depth order)
                // (null under the list cut: the emission then gathers the few Gaussians it lists from rec0 / rec1 / rect, and 32 bytes per
                // Gaussian stay unwritten)
                if (binrec) {
                    binrec[2 * (size_t)i] = make_float4(px, py, con0, con1);
                    binrec[2 * (size_t)i + 1] = make_float4(con2, thr, __uint_as_float(rc.x), __uint_as_float(rc.y));
                }
            }
        }
    }
    else if (prefilter_violation) *prefilter_violation = 1u;      // (the reference prints and traps: auxiliary.h:156-160; here the host returns GSRAST_E_ARG)
    radii[i] = rad_out; tiles[i] = ntiles; rect[i] = rc;
    sort_key[i] = key;
    if (sort_val) sort_val[i] = (uint32_t)i;        // the radix depth sort's values; the bucket sort carries the index in its slab element
    } // i < P
    // equalised depth buckets: every sampled wave's visible Gaussians count into the depth histogram (one fire-and-forget atomic each,
    // ~65 k per launch, into the workgroup's XCD's own copy; the scatter turns the histogram into its bucket map -- any histogram gives a correct order)
    if (zhist && key != 0xFFFFFFFFu && ((((uint32_t)i >> 6) * 0x9E3779B1u >> 12) & zh_wave_mask) == 0u) {       // (a pseudo-random subset of the waves: every XCD's copy gets its share)
        uint32_t bin, pos; int wlog;
        zh_locate(key, zh_klo, zh_shift, bin, pos, wlog);
        atomicAdd(&zhist[(blockIdx.x & (unsigned)(ZH_COPIES - 1)) * (unsigned)ZH_BINS + bin], 1u);
    }
    // predicted cut: one fire-and-forget atomic per small visible Gaussian (the workgroup's XCD's own copy of the table)
    // (a pseudo-random subset of the waves, their mass scaled up: three million atomics cost the kernel 40 us at 3 M, a quarter of them 10)
    if (tau_tile >= 0 && ((((uint32_t)i >> 6) * 0x9E3779B1u >> 14) & tau_bins.wave_mask) == 0u) {
        const uint32_t q = (uint32_t)(tau_mass * (float)(tau_bins.wave_mask + 1u) + 0.5f);
        if (q) atomicAdd(&tau_hist[((size_t)(blockIdx.x & (unsigned)(TAU_COPIES - 1)) * ntiles_img + (uint32_t)tau_tile) * TAU_BINS + tau_bin_of(key, tau_bins)], q);
    }
}

// K0: reference rasterizer_impl.cu:54-66
__global__ void __launch_bounds__(256)
mark_visible_kernel(int P, const float* __restrict__ means3D, const float* __restrict__ view, unsigned char* __restrict__ present)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    struct { float view[16]; } cam;
#pragma unroll
    for (int k = 0; k < 16; k++) cam.view[k] = view[k];
    const float p[3] = { means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2] };
    float pv[3];
    xform4x3(p, cam.view, pv);
    present[i] = pv[2] > 0.2f ? 1 : 0;
}

// -------------------------------------------------------------------------------------------
// SH backward: writes dL_dsh rows [0,(deg+1)^2) and ADDS the view-direction term to dmean.
// FACTORS: dL/dsh is not written at all.  Row k of it is w_k(view direction) * g, with g = the clamp-masked colour
// gradient: a rank-1 outer product of 16 weights that any rank can recompute from the camera position and 3 numbers.
// The multi-GPU exchange moves g (returned in g_out) instead of the 48 products (gsrast_sh_grad_combine).
template <bool FACTORS = false>
__device__ __forceinline__ void sh_backward(int deg, const float pos[3], const float campos[3],
                                            const float dx[3], const float dy[3], const float dz[3] /* sh_dir_derivs */,
                                            unsigned cl, const float dcol[3],
                                            float dmean[3], float* __restrict__ dsh)
{
    const float o0 = pos[0] - campos[0], o1 = pos[1] - campos[1], o2 = pos[2] - campos[2];
    const float len = sqrtf(o0 * o0 + o1 * o1 + o2 * o2);
    const float x = o0 / len, y = o1 / len, z = o2 / len;
    float g[3];
#pragma unroll
    for (int c = 0; c < 3; c++) g[c] = dcol[c] * (((cl >> c) & 1u) ? 0.0f : 1.0f);
#define PUT(k, w) { if (!FACTORS) { const float w_ = (w); dsh[(k) * 3 + 0] = w_ * g[0]; dsh[(k) * 3 + 1] = w_ * g[1]; dsh[(k) * 3 + 2] = w_ * g[2]; } }
    PUT(0, kSH0);
    if (deg > 0) {
        PUT(1, -kSH1 * y); PUT(2, kSH1 * z); PUT(3, -kSH1 * x);
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            PUT(4, kSH2[0] * xy); PUT(5, kSH2[1] * yz); PUT(6, kSH2[2] * (2.0f * zz - xx - yy));
            PUT(7, kSH2[3] * xz); PUT(8, kSH2[4] * (xx - yy));
            if (deg > 2) {
                PUT(9, kSH3[0] * y * (3.0f * xx - yy)); PUT(10, kSH3[1] * xy * z);
                PUT(11, kSH3[2] * y * (4.0f * zz - xx - yy));
                PUT(12, kSH3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy));
                PUT(13, kSH3[4] * x * (4.0f * zz - xx - yy)); PUT(14, kSH3[5] * z * (xx - yy));
                PUT(15, kSH3[6] * x * (xx - 3.0f * yy));
            }
        }
    }
#undef PUT
    const float dd0 = dx[0] * g[0] + dx[1] * g[1] + dx[2] * g[2];
    const float dd1 = dy[0] * g[0] + dy[1] * g[1] + dy[2] * g[2];
    const float dd2 = dz[0] * g[0] + dz[1] * g[1] + dz[2] * g[2];
    const float sum2 = o0 * o0 + o1 * o1 + o2 * o2;
    const float inv32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
    dmean[0] += ((+sum2 - o0 * o0) * dd0 - o1 * o0 * dd1 - o2 * o0 * dd2) * inv32;
    dmean[1] += (-o0 * o1 * dd0 + (sum2 - o1 * o1) * dd1 - o2 * o1 * dd2) * inv32;
    dmean[2] += (-o0 * o2 * dd0 - o1 * o2 * dd1 + (sum2 - o2 * o2) * dd2) * inv32;
}

// Round 5: the gradient records (GeomLayout::grec, 64 B per Gaussian) of a forward that keeps untouched bits are NOT zero-filled whole any
// more (192 MB of stores per 3 M forward, from inside the forward blend: the step 1.19 -> 1.11 ms without them).  Only the Gaussians some pixel
// consumed can receive a gradient: grec_zero_touched_kernel, behind the forward's last blend, zeroes THEIR records (0.15 M of 3 M) and raises
// scalars[SC_GREC_SPARSE]; every reader of a record asks record_is_stale() first and takes a stale record for the zero it stands for.
__device__ __forceinline__ bool record_is_stale(const uint32_t* __restrict__ scalars, const unsigned char* __restrict__ untouched, int i)
{
    return scalars && untouched && scalars[SC_GREC_SPARSE] != 0u && untouched[i] != 0;
}
constexpr int GZ_PER = 16;           // Gaussians per lane: one 16-byte load of marks
__global__ void __launch_bounds__(256)
grec_zero_touched_kernel(int P, const unsigned char* __restrict__ untouched, float4* __restrict__ grec, uint32_t* __restrict__ scalars)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) scalars[SC_GREC_SPARSE] = 1u;
    // sixteen Gaussians per lane (one 16-byte load of marks; the array is padded past P), ~5 % of them consumed: 733 workgroups at 3 M instead of
    // 11.7 k that mostly find nothing to do.  The kernel is bound by the scattered 16-byte store transactions of the consumed ones' records
    // (3 M: 4.5 us without the stores, 6.7 with one per record, 10 with four)
    const size_t i0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * GZ_PER;
    if (i0 >= (size_t)P) return;
    const uint4 f = *reinterpret_cast<const uint4*>(untouched + i0);
    const uint32_t w[4] = { f.x, f.y, f.z, f.w };
#pragma unroll
    for (int q = 0; q < 4; q++) {
        if (w[q] == 0x01010101u) continue;            // (the usual case: four untouched Gaussians)
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const size_t i = i0 + 4 * q + e;
            if (((w[q] >> (8 * e)) & 0xFFu) == 0u && i < (size_t)P) {
                float4* r = grec + 4 * i;
                const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                r[0] = z; r[1] = z; r[2] = z;          // (the record's last quarter is padding: nine sums live in floats 0-8, no reader looks past float 11)
            }
        }
    }
}

// The factor of dL/dsh (see sh_backward<FACTORS>): g = the blend backward's colour gradient with the clamped channels masked,
// zero for a culled Gaussian -- final as soon as the blend backward has run.  Written by this small kernel so that a multi-GPU
// caller can start exchanging the factors WHILE the per-Gaussian backward (below) is still running (backward_phase).
__global__ void __launch_bounds__(256)
sh_factor_kernel(int P, const int* __restrict__ radii, const unsigned char* __restrict__ clamped, const float4* __restrict__ grec,
                 float* __restrict__ g_out /* [P][3] */, const uint32_t* __restrict__ scalars, const unsigned char* __restrict__ untouched)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const bool live = radii[i] > 0 && !record_is_stale(scalars, untouched, i);
    const float4 r1 = grec[4 * (size_t)i + 1], r2 = grec[4 * (size_t)i + 2];
    const unsigned cl = clamped[i];
    const float dcol[3] = { r1.z, r1.w, r2.x };
#pragma unroll
    for (int c = 0; c < 3; c++)      // the same product as sh_backward's (sign of zero, NaN propagation)
        g_out[3 * (size_t)i + c] = live ? dcol[c] * (((cl >> c) & 1u) ? 0.0f : 1.0f) : 0.0f;
}

// d(colour)/d(view direction) of every visible Gaussian (sh_dir_derivs above), for preprocess_bwd_kernel.  Launched by
// gsrast_backward on the context's side stream beside the blend backward: no LDS and few registers, so that its waves fit into the
// register space the blend kernel's five waves per SIMD leave free; a lane reads its 12*M-byte coefficient block with 16-byte loads.
__global__ void __launch_bounds__(64)
sh_dir_derivs_kernel(int P, int D, int M, const float* __restrict__ means3D, const float* __restrict__ shs,
                     const float* __restrict__ campos_dev, const int* __restrict__ radii,
                     float4* __restrict__ shdA, float4* __restrict__ shdB, float* __restrict__ shdC)
{
    for (int i = blockIdx.x * 64 + threadIdx.x; i < P; i += gridDim.x * 64) {
    if (radii[i] <= 0) continue;
    const float o0 = means3D[3 * i] - campos_dev[0], o1 = means3D[3 * i + 1] - campos_dev[1], o2 = means3D[3 * i + 2] - campos_dev[2];
    const float len = sqrtf(o0 * o0 + o1 * o1 + o2 * o2);
    const float x = o0 / len, y = o1 / len, z = o2 / len;
    const int ncoef = (D + 1) * (D + 1);
    float dx[3], dy[3], dz[3];
    const float* row = shs + (size_t)i * M * 3;
    if (((M * 3) & 3) == 0 && ((uintptr_t)shs & 15) == 0 && ncoef * 3 <= PP_SH_MAX) {
        float v[PP_SH_MAX];
        const float4* row4 = reinterpret_cast<const float4*>(row);
#pragma unroll
        for (int q = 0; q < PP_SH_MAX / 4; q++) {
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q * 4 < ncoef * 3) t = row4[q];
            v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
        }
        sh_dir_derivs(D, x, y, z, v, dx, dy, dz);
    } else {
        sh_dir_derivs(D, x, y, z, row, dx, dy, dz);
    }
    shdA[i] = make_float4(dx[0], dx[1], dx[2], dy[0]);
    shdB[i] = make_float4(dy[1], dy[2], dz[0], dz[1]);
    shdC[i] = dz[2];
    }
}

// K6 + K7 fused.  Every output row is written exactly once (zeros for culled Gaussians), so the
// caller does not have to zero-fill the five output arrays.  dL/dsh leaves through LDS (coalesced).
// RAW (gsrast_backward_raw): means3D / scales / rotations are the model's leaves as in preprocess_fwd_kernel<true>; the chain rule
// through the activations (epilogue_small_bwd_kernel's expressions) is applied before the stores: dL_dmeans3D = d_xyz (= d_motion_res),
// dL_dscale = d_scaling, dL_drot = d_rotation, dL_dopacity = d_opacity_logit, plus RawGrads (d_rot_res, d_trbf, the SH leaves).
// (Measured and dropped with the list cut: not even READING the gradient record of a Gaussian the forward's scatter marked late --
// 167 MB of the kernel's 1.0 GB at 3 M: 770 / 782 / 773 vs 783 / 773 / 768 views/s without, the kernel is bound by its 0.8 GB of stores.)
// SPARSE (round 3, the default): a Gaussian whose gradient record is all zero -- frustum-culled, or occluded: 83 % of the 3 M bench
// scene -- is not READ: every output is linear in the record's nine sums, so its rows are exactly zero and are written as such without
// its mean / scale / rotation / direction derivatives (80 of the 312 bytes the kernel moves per Gaussian; -28 us of 230 at 3 M, -19 us
// at 1 M).  Measured and dropped: not writing those rows either, the arrays zero-filled by a kernel on the side stream under the blend
// backward -- preprocess_bwd 229 -> 130 us at 3 M, but the fill's 700 MB slowed the VALU-bound blend backward by 45 us and the extra
// launches cost the small scenes 10-50 us: no better than this at 3 M, worse everywhere else.
template <bool RAW, bool SPARSE, bool GROUPED = false>
__global__ void __launch_bounds__(PP_THREADS)
preprocess_bwd_kernel(int P, int D, int M, const float* __restrict__ means3D, const int* __restrict__ radii, RawArgs raw, RawGrads rawg,
                      const float* __restrict__ shs /* only its presence matters: the coefficients are not read */,
                      const unsigned char* __restrict__ clamped,
                      const float4* __restrict__ shdA, const float4* __restrict__ shdB, const float* __restrict__ shdC /* the colour kernel's (or sh_dir_derivs_kernel's) output */,
                      const float* __restrict__ scales, const float* __restrict__ rotations,
                      const float* __restrict__ cov3D /* internal or precomp */, CamArgs cam_args,
                      const float4* __restrict__ grec /* [P][4]: the blend backward's gradient records (GeomLayout::grec) */,
                      float* __restrict__ dL_dmean2D /* [P][3] out */, float* __restrict__ dL_dconic /* [P][4] out, may be null */,
                      float* __restrict__ dL_dopacity /* [P] out */, float* __restrict__ dL_dcolor /* [P][3] out, may be null */,
                      float* __restrict__ dL_dmeans3D,
                      float* __restrict__ dL_dcov3D, float* __restrict__ dL_dsh, float* __restrict__ dL_dscale,
                      float* __restrict__ dL_drot,
                      int sh_factors /* the caller exchanges dL/dsh by its factor (sh_factor_kernel has written it): dL_dsh rows are not written */,
                      // GROUPED (with the forward's list cut, gsrast_common.h): a workgroup takes PB_GROUP consecutive Gaussians, 128 per round.
                      // Bit i of late_bits = Gaussian i was culled or LATE -- it is in no list, its gradient record is still the zero the
                      // forward left, all its output rows are zero: late_rows_zero_kernel writes them on the side stream WHILE the blend
                      // backward runs, and this kernel neither reads nor writes such a Gaussian: it COMPACTS the others (ascending index
                      // order) and works through them densely (3 M cube: 0.4 M of 3 M), their dL/dsh rows leaving one by one.  Both
                      // kernels trust the bits only if the forward's scalars say that the cut was in force (SC_N_LATE != 0) and held
                      // (SC_UNDONE == 0: no second binning over all Gaussians); otherwise the rounds are the group's eight blocks of 128
                      // Round 5: `untouched` (GeomLayout::untouched, kept by the forward blend whenever scalars[SC_TOUCH_VALID] says so) takes
                      // the bits' place -- "no pixel consumed this Gaussian", a superset of "culled or late" that needs no pose table
                      const unsigned long long* __restrict__ late_bits = nullptr, const uint32_t* __restrict__ cut_scalars = nullptr,
                      const unsigned char* __restrict__ untouched = nullptr)
{
    __shared__ float sh_lds[PP_THREADS * PP_SH_STRIDE];
    __shared__ uint32_t s_list[GROUPED ? PB_GROUP : 1];
    __shared__ uint32_t s_wtot[PP_THREADS / 64];
    const int ncoef = (D + 1) * (D + 1);
    const bool staged = shs && M * 3 <= PP_SH_MAX;
    float* my_lds = sh_lds + threadIdx.x * PP_SH_STRIDE;
    const bool touch_ok = GROUPED && untouched && cut_scalars[SC_TOUCH_VALID] != 0u;  // (uniform)
    const bool late_rows = GROUPED && (touch_ok || (late_bits && cut_scalars[SC_N_LATE] != 0u));      // (uniform; the completion pass has taken the Gaussians it listed after all out of the cut's bits)
    const int gbase = blockIdx.x * (GROUPED ? PB_GROUP : PP_THREADS);
    int nround = 1;
    uint32_t total = 0;
    if (GROUPED) {
        constexpr int PER = PB_GROUP / PP_THREADS;               // consecutive Gaussians per lane: one, two or four bytes of bits
        static_assert(PER == 4 || PER == 8 || PER == 16 || PER == 32, "a nibble, or whole bytes up to a word, of flags per lane");
        if (late_rows) {
            const uint32_t b0 = (uint32_t)gbase + threadIdx.x * (uint32_t)PER;
            uint32_t fl = 0xFFFFFFFFu;
            if (b0 < (uint32_t)P && touch_ok) {      // the blend's marks: a byte per Gaussian (the array is padded past P)
                fl = 0u;
                if (PER == 4) { const uint32_t w = *reinterpret_cast<const uint32_t*>(untouched + b0); for (int k = 0; k < 4; k++) fl |= ((w >> (8 * k)) & 0xFFu) ? 1u << k : 0u; }
                else for (int k = 0; k < PER; k++) fl |= untouched[b0 + k] ? 1u << k : 0u;
                for (int k = 0; k < PER; k++) if (b0 + k >= (uint32_t)P) fl |= 1u << k;
            } else if (b0 < (uint32_t)P) {
                const unsigned char* bytes = reinterpret_cast<const unsigned char*>(late_bits) + (b0 >> 3);
                fl = PER == 4 ? ((uint32_t)bytes[0] >> (b0 & 4u)) & 0xFu : PER == 8 ? (uint32_t)bytes[0] : PER == 16 ? (uint32_t)*reinterpret_cast<const unsigned short*>(bytes) : *reinterpret_cast<const uint32_t*>(bytes);
                for (int k = 0; k < PER; k++) if (b0 + k >= (uint32_t)P) fl |= 1u << k;      // (the array is padded: the word may reach past P)
            }
            if (PER < 32) fl |= ~0u << (PER & 31);
            const uint32_t mine = (uint32_t)__builtin_popcount(~fl);
            uint32_t incl = mine;
            const unsigned ln = threadIdx.x & 63u;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d, 64); if (ln >= (unsigned)d) incl += o; }
            if (ln == 63u) s_wtot[threadIdx.x >> 6] = incl;
            __syncthreads();
            uint32_t pos = incl - mine;
#pragma unroll
            for (int w = 0; w < PP_THREADS / 64; w++) { if ((unsigned)w < (threadIdx.x >> 6)) pos += s_wtot[w]; total += s_wtot[w]; }
#pragma unroll
            for (int k = 0; k < PER; k++) if (!((fl >> k) & 1u)) s_list[pos++] = b0 + k;
            __syncthreads();
            nround = (int)((total + PP_THREADS - 1) / PP_THREADS);
        } else {
            const int left = P - gbase;
            nround = left <= 0 ? 0 : (left >= PB_GROUP ? PB_GROUP / PP_THREADS : (left + PP_THREADS - 1) / PP_THREADS);
        }
    }
    for (int round = 0; round < nround; round++) {
    int i = gbase + round * PP_THREADS + threadIdx.x;
    if (GROUPED && late_rows) { const uint32_t j = (uint32_t)round * PP_THREADS + threadIdx.x; i = j < total ? (int)s_list[j] : P; }
    // every per-Gaussian input is requested up front, before the SH rows are staged (see preprocess_fwd_kernel): the loads used
    // to sit behind the staging barrier, the radius test and each other -- five memory round trips in a latency-bound kernel.
    // (SPARSE: the record and the radius first, the rest only for the Gaussians that need it.)
    const int ic = i < P ? i : P - 1;
    const int radius_in = radii[ic];
    // the Gaussian's gradient record: {dL/dmean2D.x, .y, dL/dconic a, b | c, dL/dopacity, dL/dr, dL/dg | dL/db, ...} -- three 16-byte
    // loads from one 64-byte line (zero for a Gaussian no tile listed)
    float4 gr0 = grec[4 * (size_t)ic], gr1 = grec[4 * (size_t)ic + 1], gr2 = grec[4 * (size_t)ic + 2];
    if (record_is_stale(cut_scalars, untouched, ic)) { gr0 = make_float4(0.f, 0.f, 0.f, 0.f); gr1 = gr0; gr2 = gr0; }      // (nobody zeroed it: no pixel consumed the Gaussian)
    bool touched = true;
    if (SPARSE) touched = gr0.x != 0.f || gr0.y != 0.f || gr0.z != 0.f || gr0.w != 0.f || gr1.x != 0.f || gr1.y != 0.f || gr1.z != 0.f || gr1.w != 0.f || gr2.x != 0.f;
    const bool live = i < P && radius_in > 0 && touched;
    float mean[3] = { 0.f, 0.f, 0.f };
    float s[3] = { 0.f, 0.f, 0.f };
    float4 q_in = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 raw_x = make_float4(0.f, 0.f, 0.f, 0.f);        // RAW: rotation + residual, before normalisation
    float raw_sg = 0.0f, raw_tb = 1.0f;                     // RAW: sigmoid(logit), trbf
    unsigned char clamped_in = 0;
    float4 sdA = make_float4(0.f, 0.f, 0.f, 0.f), sdB = sdA; float sdC = 0.0f;
    if (!SPARSE || live) {
        mean[0] = means3D[3 * ic]; mean[1] = means3D[3 * ic + 1]; mean[2] = means3D[3 * ic + 2];
        if (scales) {                                           // uniform
            s[0] = scales[3 * ic]; s[1] = scales[3 * ic + 1]; s[2] = scales[3 * ic + 2];
            q_in = reinterpret_cast<const float4*>(rotations)[ic];
        }
        if (RAW) {
            raw_mean(raw, ic, mean);
            raw_rot_scale(raw, ic, q_in, s, raw_x);
            raw_sg = raw_sigmoid(raw.opacity_logit[ic]);
            raw_tb = raw.trbf ? raw.trbf[ic] : 1.0f;
        }
        if (shs) {                                              // uniform
            clamped_in = clamped[ic];
            if (D > 0) { sdA = shdA[ic]; sdB = shdB[ic]; sdC = shdC[ic]; }
        }
        if (!SPARSE) {
            // the dense form reads every Gaussian -- but under the forward's list cut (gsrast_common.h) the colour kernel has left the
            // direction derivatives and clamp flags of a late Gaussian unwritten: its gradient record is zero, so they are taken as zero
            const bool zero_rec = gr0.x == 0.f && gr0.y == 0.f && gr0.z == 0.f && gr0.w == 0.f && gr1.x == 0.f && gr1.y == 0.f && gr1.z == 0.f && gr1.w == 0.f && gr2.x == 0.f;
            if (zero_rec) { sdA = make_float4(0.f, 0.f, 0.f, 0.f); sdB = sdA; sdC = 0.0f; clamped_in = 0; }
        }
    }
    const float4 dcon = make_float4(gr0.z, gr0.w, 0.0f, gr1.x);          // reference layout: .z is never written (backward.cu:549-551)
    const float g2x = gr0.x, g2y = gr0.y;
    const float dcol[3] = { gr1.z, gr1.w, gr2.x };
    const Cam cam = load_cam(cam_args);
    if (i < P) {    // the screen-space gradients leave in the reference's arrays (rasterize_points.cu:150-158), written once
        dL_dmean2D[3 * (size_t)i] = g2x; dL_dmean2D[3 * (size_t)i + 1] = g2y; dL_dmean2D[3 * (size_t)i + 2] = 0.0f;
        if (RAW) {      // d(opacity logit) = d_opacity * trbf * s (1 - s),  d(trbf) = d_opacity * s
            const float go = gr1.y;
            const bool have = !SPARSE || live;          // (an untouched Gaussian has go = 0: zeros, without its logit)
            dL_dopacity[i] = have ? go * raw_tb * raw_sg * (1.0f - raw_sg) : 0.0f;
            if (rawg.d_trbf) rawg.d_trbf[i] = have ? go * raw_sg : 0.0f;
        } else
        dL_dopacity[i] = gr1.y;
        if (dL_dcolor) { dL_dcolor[3 * (size_t)i] = dcol[0]; dL_dcolor[3 * (size_t)i + 1] = dcol[1]; dL_dcolor[3 * (size_t)i + 2] = dcol[2]; }
        if (dL_dconic) reinterpret_cast<float4*>(dL_dconic)[i] = dcon;
    }
    if (i < P && !live) {
#pragma unroll
        for (int k = 0; k < 3; k++) dL_dmeans3D[3 * (size_t)i + k] = 0.0f;
        if (dL_dcov3D) {
#pragma unroll
            for (int k = 0; k < 6; k++) dL_dcov3D[6 * (size_t)i + k] = 0.0f;
        }
        if (shs) {
            if (sh_factors) { }
            else if (staged) { for (int k = 0; k < M * 3; k++) my_lds[k] = 0.0f; }
            else { for (int k = 0; k < M * 3; k++) dL_dsh[(size_t)i * M * 3 + k] = 0.0f; }
        }
        if (scales) {
#pragma unroll
            for (int k = 0; k < 3; k++) dL_dscale[3 * (size_t)i + k] = 0.0f;
            reinterpret_cast<float4*>(dL_drot)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (RAW && rawg.d_rot_res) { float* o = rawg.d_rot_res + 7 * (size_t)i; for (int k = 0; k < 7; k++) o[k] = 0.0f; }
        }
    }
    if (live) {
    // 3D covariance: recomputed from scale / rotation when those are the inputs (bit-identical to what the forward
    // stored -- same function, same operands -- and 24 B / Gaussian less to read), else the caller's cov3D_precomp
    float c6[6];
    float q[4] = { q_in.x, q_in.y, q_in.z, q_in.w };
    M3 Mm;
    if (scales) {
        cov3d_from_scale_rot(s, cam.scale_mod, q, c6, Mm);
    } else {
#pragma unroll
        for (int k = 0; k < 6; k++) c6[k] = cov3D[6 * (size_t)i + k];
    }
    Cov2D cv;
    cov2d_eval(mean, cam, c6, cv);
    const float xgm = (cv.txtz < -cv.limx || cv.txtz > cv.limx) ? 0.0f : 1.0f;
    const float ygm = (cv.tytz < -cv.limy || cv.tytz > cv.limy) ? 0.0f : 1.0f;
    const float a = cv.a, b = cv.b, c = cv.c;
    const float dcx = dcon.x, dcy = dcon.y, dcz = dcon.w;
    const float denom = a * c - b * b;
    float dL_da = 0, dL_db = 0, dL_dc = 0;
    const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
    const M3& T = cv.T; const M3& V = cv.V; const M3& Wm = cv.Wm;
    float dcov[6];
    if (denom2inv != 0) {
        dL_da = denom2inv * (-c * c * dcx + 2.0f * b * c * dcy + (denom - a * c) * dcz);
        dL_dc = denom2inv * (-a * a * dcz + 2.0f * a * b * dcy + (denom - a * c) * dcx);
        dL_db = denom2inv * 2.0f * (b * c * dcx - (denom + 2.0f * b * b) * dcy + a * b * dcz);
        dcov[0] = (T.m[0][0] * T.m[0][0] * dL_da + T.m[0][0] * T.m[1][0] * dL_db + T.m[1][0] * T.m[1][0] * dL_dc);
        dcov[3] = (T.m[0][1] * T.m[0][1] * dL_da + T.m[0][1] * T.m[1][1] * dL_db + T.m[1][1] * T.m[1][1] * dL_dc);
        dcov[5] = (T.m[0][2] * T.m[0][2] * dL_da + T.m[0][2] * T.m[1][2] * dL_db + T.m[1][2] * T.m[1][2] * dL_dc);
        dcov[1] = 2.0f * T.m[0][0] * T.m[0][1] * dL_da + (T.m[0][0] * T.m[1][1] + T.m[0][1] * T.m[1][0]) * dL_db + 2.0f * T.m[1][0] * T.m[1][1] * dL_dc;
        dcov[2] = 2.0f * T.m[0][0] * T.m[0][2] * dL_da + (T.m[0][0] * T.m[1][2] + T.m[0][2] * T.m[1][0]) * dL_db + 2.0f * T.m[1][0] * T.m[1][2] * dL_dc;
        dcov[4] = 2.0f * T.m[0][2] * T.m[0][1] * dL_da + (T.m[0][1] * T.m[1][2] + T.m[0][2] * T.m[1][1]) * dL_db + 2.0f * T.m[1][1] * T.m[1][2] * dL_dc;
    } else {
#pragma unroll
        for (int k = 0; k < 6; k++) dcov[k] = 0.0f;
    }
    if (dL_dcov3D) {     // only a caller that passed cov3D_precomp wants it: with scales / rotations it is an intermediate
#pragma unroll
        for (int k = 0; k < 6; k++) dL_dcov3D[6 * (size_t)i + k] = dcov[k];
    }
    float dT[2][3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float r0 = T.m[0][0] * V.m[k][0] + T.m[0][1] * V.m[k][1] + T.m[0][2] * V.m[k][2];
        const float r1 = T.m[1][0] * V.m[k][0] + T.m[1][1] * V.m[k][1] + T.m[1][2] * V.m[k][2];
        dT[0][k] = 2.0f * r0 * dL_da + r1 * dL_db;
        dT[1][k] = 2.0f * r1 * dL_dc + r0 * dL_db;
    }
    const float dJ00 = Wm.m[0][0] * dT[0][0] + Wm.m[0][1] * dT[0][1] + Wm.m[0][2] * dT[0][2];
    const float dJ02 = Wm.m[2][0] * dT[0][0] + Wm.m[2][1] * dT[0][1] + Wm.m[2][2] * dT[0][2];
    const float dJ11 = Wm.m[1][0] * dT[1][0] + Wm.m[1][1] * dT[1][1] + Wm.m[1][2] * dT[1][2];
    const float dJ12 = Wm.m[2][0] * dT[1][0] + Wm.m[2][1] * dT[1][1] + Wm.m[2][2] * dT[1][2];
    const float tz = 1.0f / cv.t[2], tz2 = tz * tz, tz3 = tz2 * tz;
    const float dtx = xgm * -cam.fx * tz2 * dJ02;
    const float dty = ygm * -cam.fy * tz2 * dJ12;
    const float dtz = -cam.fx * tz2 * dJ00 - cam.fy * tz2 * dJ11 + (2.0f * cam.fx * cv.t[0]) * tz3 * dJ02 + (2.0f * cam.fy * cv.t[1]) * tz3 * dJ12;
    float dmean[3] = { cam.view[0] * dtx + cam.view[1] * dty + cam.view[2] * dtz,
                       cam.view[4] * dtx + cam.view[5] * dty + cam.view[6] * dtz,
                       cam.view[8] * dtx + cam.view[9] * dty + cam.view[10] * dtz };
    // mean gradient through the perspective projection of the 2D mean
    float mh[4];
    xform4x4(mean, cam.proj, mh);
    const float mw = 1.0f / (mh[3] + 0.0000001f);
    const float* pj = cam.proj;
    const float mul1 = (pj[0] * mean[0] + pj[4] * mean[1] + pj[8] * mean[2] + pj[12]) * mw * mw;
    const float mul2 = (pj[1] * mean[0] + pj[5] * mean[1] + pj[9] * mean[2] + pj[13]) * mw * mw;
    dmean[0] += (pj[0] * mw - pj[3] * mul1) * g2x + (pj[1] * mw - pj[3] * mul2) * g2y;
    dmean[1] += (pj[4] * mw - pj[7] * mul1) * g2x + (pj[5] * mw - pj[7] * mul2) * g2y;
    dmean[2] += (pj[8] * mw - pj[11] * mul1) * g2x + (pj[9] * mw - pj[11] * mul2) * g2y;
    if (shs) {
        const float ddx[3] = { sdA.x, sdA.y, sdA.z }, ddy[3] = { sdA.w, sdB.x, sdB.y }, ddz[3] = { sdB.z, sdB.w, sdC };
        if (sh_factors) {    // the factor itself left with sh_factor_kernel, right after the blend backward
            sh_backward<true>(D, mean, cam.campos, ddx, ddy, ddz, clamped_in, dcol, dmean, nullptr);
        } else if (staged) {
            sh_backward(D, mean, cam.campos, ddx, ddy, ddz, clamped_in, dcol, dmean, my_lds);
            for (int k = ncoef * 3; k < M * 3; k++) my_lds[k] = 0.0f;
        } else {
            float* dsh = dL_dsh + (size_t)i * M * 3;
            sh_backward(D, mean, cam.campos, ddx, ddy, ddz, clamped_in, dcol, dmean, dsh);
            for (int k = ncoef * 3; k < M * 3; k++) dsh[k] = 0.0f;
        }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) dL_dmeans3D[3 * (size_t)i + k] = dmean[k];
    if (scales) {
        M3 R;
        quat_to_R(q, R);
        const float sm[3] = { cam.scale_mod * s[0], cam.scale_mod * s[1], cam.scale_mod * s[2] };
        M3 dSig, M2, dM;
        dSig.m[0][0] = dcov[0]; dSig.m[0][1] = 0.5f * dcov[1]; dSig.m[0][2] = 0.5f * dcov[2];
        dSig.m[1][0] = 0.5f * dcov[1]; dSig.m[1][1] = dcov[3]; dSig.m[1][2] = 0.5f * dcov[4];
        dSig.m[2][0] = 0.5f * dcov[2]; dSig.m[2][1] = 0.5f * dcov[4]; dSig.m[2][2] = dcov[5];
#pragma unroll
        for (int cc = 0; cc < 3; cc++)
#pragma unroll
            for (int rr = 0; rr < 3; rr++) M2.m[cc][rr] = 2.0f * Mm.m[cc][rr];
#pragma unroll
        for (int cc = 0; cc < 3; cc++)
#pragma unroll
            for (int rr = 0; rr < 3; rr++)
                dM.m[cc][rr] = M2.m[0][rr] * dSig.m[cc][0] + M2.m[1][rr] * dSig.m[cc][1] + M2.m[2][rr] * dSig.m[cc][2];
        // Rt[k][j] = R[j][k], dMt[k][j] = dM[j][k]
        float ds[3];
#pragma unroll
        for (int k = 0; k < 3; k++) ds[k] = R.m[0][k] * dM.m[0][k] + R.m[1][k] * dM.m[1][k] + R.m[2][k] * dM.m[2][k];
        if (RAW) {      // d(scaling + residual) = d_scale * scale
#pragma unroll
            for (int k = 0; k < 3; k++) ds[k] = ds[k] * s[k];
        }
#pragma unroll
        for (int k = 0; k < 3; k++) dL_dscale[3 * (size_t)i + k] = ds[k];
#define D_(cc, rr) (dM.m[rr][cc] * sm[cc])
        const float r = q[0], x = q[1], y = q[2], z = q[3];
        float4 dq;
        dq.x = 2.0f * z * (D_(0, 1) - D_(1, 0)) + 2.0f * y * (D_(2, 0) - D_(0, 2)) + 2.0f * x * (D_(1, 2) - D_(2, 1));
        dq.y = 2.0f * y * (D_(1, 0) + D_(0, 1)) + 2.0f * z * (D_(2, 0) + D_(0, 2)) + 2.0f * r * (D_(1, 2) - D_(2, 1)) - 4.0f * x * (D_(2, 2) + D_(1, 1));
        dq.z = 2.0f * x * (D_(1, 0) + D_(0, 1)) + 2.0f * r * (D_(2, 0) - D_(0, 2)) + 2.0f * z * (D_(1, 2) + D_(2, 1)) - 4.0f * y * (D_(2, 2) + D_(0, 0));
        dq.w = 2.0f * r * (D_(0, 1) - D_(1, 0)) + 2.0f * x * (D_(2, 0) + D_(0, 2)) + 2.0f * y * (D_(1, 2) + D_(2, 1)) - 4.0f * z * (D_(1, 1) + D_(0, 0));
#undef D_
        if (RAW) {      // d(rotation + residual) = (g - y <y, g>) / |x|,  y = x / |x|  (|x| < eps: the clamp is not differentiated)
            const float4 g = dq;
            const float nn = sqrtf(raw_x.x * raw_x.x + raw_x.y * raw_x.y + raw_x.z * raw_x.z + raw_x.w * raw_x.w);
            if (nn >= 1e-12f) {
                const float inv = 1.0f / nn;
                const float4 yv = make_float4(raw_x.x * inv, raw_x.y * inv, raw_x.z * inv, raw_x.w * inv);
                const float dot = yv.x * g.x + yv.y * g.y + yv.z * g.z + yv.w * g.w;
                dq = make_float4((g.x - yv.x * dot) * inv, (g.y - yv.y * dot) * inv, (g.z - yv.z * dot) * inv, (g.w - yv.w * dot) * inv);
            } else {
                dq = make_float4(g.x * 1e12f, g.y * 1e12f, g.z * 1e12f, g.w * 1e12f);
            }
            if (rawg.d_rot_res) {
                float* o = rawg.d_rot_res + 7 * (size_t)i;
                o[0] = dq.x; o[1] = dq.y; o[2] = dq.z; o[3] = dq.w; o[4] = ds[0]; o[5] = ds[1]; o[6] = ds[2];
            }
        }
        reinterpret_cast<float4*>(dL_drot)[i] = dq;
    }
    } // live
    if (staged && !sh_factors) {
        __syncthreads();
        const int rbase = gbase + round * PP_THREADS;
        if (GROUPED && late_rows) {      // the compacted Gaussians' rows leave one by one
            const uint32_t* ids = s_list + round * PP_THREADS;
            const int cnt = (int)min((uint32_t)PP_THREADS, total - (uint32_t)round * PP_THREADS);
            if (RAW) {
                if (rawg.d_shs_res) scatter_sh_rows(rawg.d_shs_res, M * 3, 0, M * 3, ids, cnt, sh_lds);
                if (rawg.d_dc) { scatter_sh_rows(rawg.d_dc, 3, 0, 3, ids, cnt, sh_lds); if (M > 1) scatter_sh_rows(rawg.d_rest, M * 3 - 3, 3, M * 3 - 3, ids, cnt, sh_lds); }
            } else scatter_sh_rows(dL_dsh, M * 3, 0, M * 3, ids, cnt, sh_lds);
        } else if (RAW) {      // the rows leave as the gradient of shs_res (whole) and / or of the two SH leaves (split): whatever the caller gave
            if (rawg.d_shs_res) stage_sh_out(rawg.d_shs_res, P, M, rbase, sh_lds);
            if (rawg.d_dc) stage_sh_out_split(rawg.d_dc, rawg.d_rest, P, M * 3, rbase, sh_lds);
        } else stage_sh_out(dL_dsh, P, M, rbase, sh_lds);
        if (GROUPED) __syncthreads();           // the next round overwrites the rows
    }
    } // round
}

// The zero rows of the Gaussians the forward's list cut left out (gsrast_common.h; preprocess_bwd_kernel's late_bits): 87 % of the 0.8 GB
// the per-Gaussian backward writes at 3 M.  They depend on nothing the blend backward produces, so they are written on the side stream
// WHILE that VALU-bound kernel runs.  One wave per 64 consecutive Gaussians (one word of bits), every output array in turn, coalesced.
struct LateRowsArgs { float* ptr[12]; int rowlen[12]; int n; };
__global__ void __launch_bounds__(256)
late_rows_zero_kernel(int P, const unsigned long long* __restrict__ late_bits, const uint32_t* __restrict__ cut_scalars, LateRowsArgs a,
                      const unsigned char* __restrict__ untouched)
{
    const bool marks = untouched && cut_scalars[SC_TOUCH_VALID] != 0u;      // the same verdict as preprocess_bwd_kernel's: the blend's byte marks, or the cut's bits
    if (!marks && cut_scalars[SC_N_LATE] == 0u) return;
    const unsigned lane = threadIdx.x & 63u;
    const uint32_t nw = ((uint32_t)P + 63u) / 64u;
    for (uint32_t w = blockIdx.x * 4u + (threadIdx.x >> 6); w < nw; w += gridDim.x * 4u) {
        const uint32_t base = w * 64u, left = (uint32_t)P - base;
        unsigned long long m = marks ? __ballot(base + lane < (uint32_t)P && untouched[base + lane] != 0) : late_bits[w];
        if (left < 64u) m &= (1ull << left) - 1ull;              // (rows past P do not exist)
        if (m == 0ull) continue;
        for (int k = 0; k < a.n; k++) {
            const uint32_t rl = (uint32_t)a.rowlen[k];
            float* dst = a.ptr[k] + (size_t)base * rl;
            if ((rl & 3u) == 0u && ((uintptr_t)dst & 15) == 0) {
                const uint32_t r4 = rl >> 2;
                // NON-TEMPORAL stores: 0.8 GB of zeros per 3 M backward that nobody reads before the optimizer does -- written the ordinary way
                // they push the Gaussians' own arrays out of the 256 MB Infinity Cache right before the per-Gaussian backward and the next
                // forward read them (3 M: per-Gaussian backward 0.115 -> 0.093 ms, this kernel 0.27 -> 0.22, the step +4 % views/s)
                typedef float v4f __attribute__((ext_vector_type(4)));
                v4f* d4 = reinterpret_cast<v4f*>(dst);
                for (uint32_t q = lane; q < 64u * r4; q += 64u) if ((m >> (q / r4)) & 1ull) __builtin_nontemporal_store(v4f{0.f, 0.f, 0.f, 0.f}, d4 + q);
            } else {
                for (uint32_t f = lane; f < 64u * rl; f += 64u) if ((m >> (f / rl)) & 1ull) __builtin_nontemporal_store(0.0f, dst + f);
            }
        }
    }
}

// dL/dsh of a batch of N views from the N per-view factors (see sh_backward<FACTORS>):
//   dL_dsh[i][k][c] = scale * sum_r w_k(dir(means3D[i] - campos_r)) * g_r[i][c],   r in rank order.
// chunks: N records of `stride` floats: [3P floats g | 3 floats campos | padding].  With N = 1, scale = 1 the result is
// bit-identical to what preprocess_bwd_kernel writes itself (0 + w*g).
// rows / row_of (gsrast_sh_grad_combine_rows): the records hold `rows` factors, Gaussian i's is row row_of[i] (-1: nobody sent it, its
// gradient is zero); d_dc / d_rest: the result split into SaRO-GS's two SH leaves (and / or whole into dL_dsh).
// one view's term of dL/dsh for a Gaussian at `pos`: acc[k][c] += w_k(dir(pos - campos)) * g[c]
__device__ __forceinline__ void sh_factor_term(float (&acc)[PP_SH_MAX], const float (&pos)[3], int D, float cx, float cy, float cz, float g0, float g1, float g2)
{
    const float o0 = pos[0] - cx, o1 = pos[1] - cy, o2 = pos[2] - cz;
    const float len = sqrtf(o0 * o0 + o1 * o1 + o2 * o2);
    const float x = o0 / len, y = o1 / len, z = o2 / len;
    const float g[3] = { g0, g1, g2 };
#define ACC(k, w) { const float w_ = (w); acc[(k) * 3 + 0] += w_ * g[0]; acc[(k) * 3 + 1] += w_ * g[1]; acc[(k) * 3 + 2] += w_ * g[2]; }
    ACC(0, kSH0);
    if (D > 0) {
        ACC(1, -kSH1 * y); ACC(2, kSH1 * z); ACC(3, -kSH1 * x);
        if (D > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            ACC(4, kSH2[0] * xy); ACC(5, kSH2[1] * yz); ACC(6, kSH2[2] * (2.0f * zz - xx - yy));
            ACC(7, kSH2[3] * xz); ACC(8, kSH2[4] * (xx - yy));
            if (D > 2) {
                ACC(9, kSH3[0] * y * (3.0f * xx - yy)); ACC(10, kSH3[1] * xy * z);
                ACC(11, kSH3[2] * y * (4.0f * zz - xx - yy));
                ACC(12, kSH3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy));
                ACC(13, kSH3[4] * x * (4.0f * zz - xx - yy)); ACC(14, kSH3[5] * z * (xx - yy));
                ACC(15, kSH3[6] * x * (xx - 3.0f * yy));
            }
        }
    }
#undef ACC
}
// the sum over the N records for one Gaussian at `pos` whose factor is row `row` of each record
__device__ __forceinline__ void sh_factor_sum(float (&acc)[PP_SH_MAX], const float (&pos)[3], int D, int N, const float* __restrict__ chunks,
                                              size_t stride, int rows, int row)
{
    for (int r = 0; r < N; r++) {
        const float* ch = chunks + (size_t)r * stride;
        sh_factor_term(acc, pos, D, ch[3 * (size_t)rows], ch[3 * (size_t)rows + 1], ch[3 * (size_t)rows + 2],
                       ch[3 * (size_t)row], ch[3 * (size_t)row + 1], ch[3 * (size_t)row + 2]);
    }
}

__global__ void __launch_bounds__(PP_THREADS)
sh_grad_combine_kernel(int P, int D, int M, int N, const float* __restrict__ means3D, const float* __restrict__ chunks,
                       size_t stride, float scale, float* __restrict__ dL_dsh, int rows, const int* __restrict__ row_of,
                       float* __restrict__ d_dc, float* __restrict__ d_rest)
{
    __shared__ float sh_lds[PP_THREADS * PP_SH_STRIDE];
    const int i = blockIdx.x * PP_THREADS + threadIdx.x;
    const bool staged = M * 3 <= PP_SH_MAX;
    float acc[PP_SH_MAX];
#pragma unroll
    for (int k = 0; k < PP_SH_MAX; k++) acc[k] = 0.0f;
    const int row = i < P ? (row_of ? row_of[i] : i) : -1;
    if (row >= 0) {
        const float pos[3] = { means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2] };
        sh_factor_sum(acc, pos, D, N, chunks, stride, rows, row);
    }
    if (staged) {
        float* my_lds = sh_lds + threadIdx.x * PP_SH_STRIDE;
        if (i < P) {
#pragma unroll
            for (int k = 0; k < PP_SH_MAX; k++) if (k < M * 3) my_lds[k] = N == 1 && scale == 1.0f ? acc[k] : acc[k] * scale;
        }
        __syncthreads();
        if (dL_dsh) stage_sh_out(dL_dsh, P, M, blockIdx.x * PP_THREADS, sh_lds);
        if (d_dc) stage_sh_out_split(d_dc, d_rest, P, M * 3, blockIdx.x * PP_THREADS, sh_lds);
    } else if (i < P) {
        for (int k = 0; k < M * 3; k++) {
            const float v = k < PP_SH_MAX ? acc[k] * scale : 0.0f;
            if (dL_dsh) dL_dsh[(size_t)i * M * 3 + k] = v;
            if (d_dc) { if (k < 3) d_dc[(size_t)i * 3 + k] = v; else d_rest[(size_t)i * (M * 3 - 3) + (k - 3)] = v; }
        }
    }
}

// The same sum for the rows of a UNION only (gsrast_sh_grad_combine_union): record row j belongs to Gaussian idx[j] (ascending, distinct);
// rows of dL_dsh outside the union are NOT written -- the caller keeps them zero (view_parallel.py: the union of the previous step is
// cleared first).  3 M Gaussians, 150 k in the union: 29 MB written instead of 576 MB (0.275 -> ~0.03 ms on the exchange path).
// One thread per union row computes, the block then writes its 256 rows with 16-byte stores, consecutive lanes along a row.
__global__ void __launch_bounds__(PP_THREADS)
sh_grad_combine_union_kernel(int rows, const long long* __restrict__ idx, int D, int M, int N, const float* __restrict__ means3D,
                             const float* __restrict__ chunks, size_t stride, float scale, float* __restrict__ dL_dsh,
                             float* __restrict__ d_dc, float* __restrict__ d_rest)
{
    __shared__ float sh_lds[PP_THREADS * PP_SH_STRIDE];
    __shared__ long long s_idx[PP_THREADS];
    const int j = blockIdx.x * PP_THREADS + threadIdx.x;
    const int L = M * 3;                                   // (the host admits L <= PP_SH_MAX, L % 4 == 0 only)
    float acc[PP_SH_MAX];
#pragma unroll
    for (int k = 0; k < PP_SH_MAX; k++) acc[k] = 0.0f;
    long long i = -1;
    if (j < rows) {
        i = idx[j];
        const float pos[3] = { means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2] };
        sh_factor_sum(acc, pos, D, N, chunks, stride, rows, j);
    }
    s_idx[threadIdx.x] = i;
    float* my_lds = sh_lds + threadIdx.x * PP_SH_STRIDE;
#pragma unroll
    for (int k = 0; k < PP_SH_MAX; k++) if (k < L) my_lds[k] = N == 1 && scale == 1.0f ? acc[k] : acc[k] * scale;
    __syncthreads();
    const int q4 = L >> 2, n_here = min(PP_THREADS, rows - (int)blockIdx.x * PP_THREADS);
    for (int q = threadIdx.x; q < n_here * q4; q += PP_THREADS) {
        const int r = q / q4, part = q - r * q4;
        const float* src = sh_lds + r * PP_SH_STRIDE + part * 4;
        const float4 v = make_float4(src[0], src[1], src[2], src[3]);
        const size_t g = (size_t)s_idx[r];
        if (dL_dsh) *reinterpret_cast<float4*>(dL_dsh + g * L + part * 4) = v;
        if (d_dc) {
            // the split leaves: floats 0..2 of the row go to d_dc, 3..L-1 to d_rest -- a 16-byte piece straddles them only at part 0
            if (part == 0) { d_dc[g * 3] = v.x; d_dc[g * 3 + 1] = v.y; d_dc[g * 3 + 2] = v.z; d_rest[g * (L - 3)] = v.w; }
            else { float* d = d_rest + g * (L - 3) + (part * 4 - 3); d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w; }
        }
    }
}

// rows idx[0..n) of up to 8 row-major arrays side by side in one packed [n][sum of widths] array, and back (the sparse exchange's
// compaction: one launch each way instead of an index_select per array, a cat and an index_copy_ per array)
struct RowArrays { float* ptr[8]; int width[8]; int n; int total; };
template <bool PACK>
__global__ void __launch_bounds__(256)
rows_pack_kernel(long long n, const long long* __restrict__ idx, RowArrays a, float* __restrict__ packed)
{
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    if (q >= n * a.total) return;
    const long long j = q / a.total;
    int f = (int)(q - j * a.total), k = 0;
    while (f >= a.width[k]) { f -= a.width[k]; k++; }
    float* cell = a.ptr[k] + (size_t)idx[j] * a.width[k] + f;
    if (PACK) packed[q] = *cell; else *cell = packed[q];
}
} // namespace gsrast
