// gsrast_loss.h -- fused photometric loss right after the rasterizer:
//     loss = (1 - lambda) * mean|x - y| + lambda * (1 - mean(SSIM_map(x, y)))
// Reference behaviour restated (paths relative to /root/reference/):
//   utils/loss_utils.py:18-19   l1_loss  = mean |x - y|
//   utils/loss_utils.py:25-35   11-tap Gaussian window, sigma 1.5, normalised in fp32; 2-D window = outer product
//   utils/loss_utils.py:48-68   _ssim: five depthwise 11x11 convolutions with zero padding 5
//                               (mu1, mu2, E[x^2], E[y^2], E[xy]), C1 = 0.01^2, C2 = 0.03^2, mean over everything
//   helper_train.py:50-53       loss = (1 - lambda_dssim) * Ll1 + lambda_dssim * (1 - ssim)
// The reference runs 5 convolutions forward and autograd replays them backward (~10 full-image passes
// plus the elementwise chain).  Here: ONE forward kernel (both images staged once per 16x16 tile with a
// 5-pixel halo in LDS, separable window, SSIM + its three partial derivatives per pixel, per-block
// partial sums) and ONE backward kernel (three separable blurs of the derivative maps, combined with
// x, y and the L1 sign).  The separable evaluation differs from the reference's 2-D window by fp32
// rounding only (~1e-7 relative); tests bound it.
#pragma once
#include "gsrast_common.h"

namespace gsrast {

constexpr int LW = 11, LR = 5;               // window size / radius
constexpr int LT = 16;                       // output tile edge
constexpr int LH = LT + 2 * LR;              // staged edge (26)

struct LossWin { float w[LW]; };

// per-(channel, tile) partial sums: {sum |x-y|, sum ssim}
__global__ void __launch_bounds__(256)
loss_fwd_kernel(int C, int H, int W, const float* __restrict__ img, const float* __restrict__ gt, LossWin win,
                float* __restrict__ d_mu, float* __restrict__ d_e11, float* __restrict__ d_e12,
                float2* __restrict__ partial)
{
    __shared__ float sx[LH][LH + 1];
    __shared__ float sy[LH][LH + 1];
    __shared__ float hh[5][LH][LT + 1];      // horizontally blurred x, y, xx, yy, xy
    __shared__ float red[2][4];
    const int tiles_x = (W + LT - 1) / LT, tiles_y = (H + LT - 1) / LT;
    const int c = blockIdx.x / (tiles_x * tiles_y);
    const int tt = blockIdx.x % (tiles_x * tiles_y);
    const int tx = tt % tiles_x, ty = tt / tiles_x;
    const int x0 = tx * LT - LR, y0 = ty * LT - LR;
    const size_t plane = (size_t)H * W;
    const float* px = img + c * plane;
    const float* py = gt + c * plane;
    for (int k = threadIdx.x; k < LH * LH; k += 256) {
        const int r = k / LH, q = k - r * LH;
        const int gy = y0 + r, gx = x0 + q;
        const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;       // zero padding (F.conv2d padding=5)
        sx[r][q] = in ? px[(size_t)gy * W + gx] : 0.0f;
        sy[r][q] = in ? py[(size_t)gy * W + gx] : 0.0f;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < LH * LT; k += 256) {
        const int r = k / LT, q = k - r * LT;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;
#pragma unroll
        for (int j = 0; j < LW; j++) {
            const float xv = sx[r][q + j], yv = sy[r][q + j], wv = win.w[j];
            a0 = __builtin_fmaf(wv, xv, a0); a1 = __builtin_fmaf(wv, yv, a1);
            a2 = __builtin_fmaf(wv, xv * xv, a2); a3 = __builtin_fmaf(wv, yv * yv, a3); a4 = __builtin_fmaf(wv, xv * yv, a4);
        }
        hh[0][r][q] = a0; hh[1][r][q] = a1; hh[2][r][q] = a2; hh[3][r][q] = a3; hh[4][r][q] = a4;
    }
    __syncthreads();
    const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
    const int gx = tx * LT + lx, gy = ty * LT + ly;
    float l1 = 0.f, ss = 0.f;
    if (gx < W && gy < H) {
        float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
        for (int j = 0; j < LW; j++) {
            const float wv = win.w[j];
            mu1 = __builtin_fmaf(wv, hh[0][ly + j][lx], mu1); mu2 = __builtin_fmaf(wv, hh[1][ly + j][lx], mu2);
            e11 = __builtin_fmaf(wv, hh[2][ly + j][lx], e11); e22 = __builtin_fmaf(wv, hh[3][ly + j][lx], e22);
            e12 = __builtin_fmaf(wv, hh[4][ly + j][lx], e12);
        }
        const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
        const float mu1s = mu1 * mu1, mu2s = mu2 * mu2, mu12 = mu1 * mu2;
        const float s1 = e11 - mu1s, s2 = e22 - mu2s, s12 = e12 - mu12;
        const float A1 = 2.0f * mu12 + C1, A2 = 2.0f * s12 + C2, B1 = mu1s + mu2s + C1, B2 = s1 + s2 + C2;
        const float inv = 1.0f / (B1 * B2);
        const float S = A1 * A2 * inv;
        // partial derivatives of S w.r.t. the three blurred quantities that depend on x: mu1, E[x^2], E[xy]
        const size_t o = c * plane + (size_t)gy * W + gx;
        d_e11[o] = -S / B2;
        d_e12[o] = 2.0f * A1 * inv;
        d_mu[o] = 2.0f * mu2 * (A2 - A1) * inv - 2.0f * mu1 * S * (1.0f / B1 - 1.0f / B2);
        ss = S;
        l1 = fabsf(sx[ly + LR][lx + LR] - sy[ly + LR][lx + LR]);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { l1 += __shfl_xor(l1, d, 64); ss += __shfl_xor(ss, d, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = l1; red[1][threadIdx.x >> 6] = ss; }
    __syncthreads();
    if (threadIdx.x == 0)
        partial[blockIdx.x] = make_float2(red[0][0] + red[0][1] + red[0][2] + red[0][3], red[1][0] + red[1][1] + red[1][2] + red[1][3]);
}

// deterministic final reduction (one workgroup, fixed order) -> out = {loss, l1, ssim}
__global__ void __launch_bounds__(256)
loss_reduce_kernel(const float2* __restrict__ partial, int n, float inv_count, float lambda, float* __restrict__ out)
{
    __shared__ double r1[256], r2[256];
    double a = 0.0, b = 0.0;
    // eight independent loads per round, added in index order (the plain loop waited one memory round trip per element: 25 us for the
    // 24 480 partial sums of a 1080p image; the order of the additions -- and so the result -- is unchanged)
    for (int i0 = threadIdx.x; i0 < n; i0 += 256 * 8) {
        float2 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const int i = i0 + u * 256; v[u] = i < n ? partial[i] : make_float2(0.f, 0.f); }
#pragma unroll
        for (int u = 0; u < 8; u++) { const int i = i0 + u * 256; if (i < n) { a += (double)v[u].x; b += (double)v[u].y; } }
    }
    r1[threadIdx.x] = a; r2[threadIdx.x] = b;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { r1[threadIdx.x] += r1[threadIdx.x + s]; r2[threadIdx.x] += r2[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float l1 = (float)(r1[0] * (double)inv_count), ssim = (float)(r2[0] * (double)inv_count);
        out[0] = (1.0f - lambda) * l1 + lambda * (1.0f - ssim);
        out[1] = l1; out[2] = ssim;
    }
}

// dL/dx = up * [ (1-lambda)/N * sign(x-y) - lambda/N * ( blur(d_mu) + 2x blur(d_e11) + y blur(d_e12) ) ]
__global__ void __launch_bounds__(256)
loss_bwd_kernel(int C, int H, int W, const float* __restrict__ img, const float* __restrict__ gt, LossWin win,
                const float* __restrict__ d_mu, const float* __restrict__ d_e11, const float* __restrict__ d_e12,
                float lambda, float inv_count, const float* __restrict__ upstream, float* __restrict__ dL_dimg)
{
    __shared__ float sm[3][LH][LH + 1];
    __shared__ float hh[3][LH][LT + 1];
    const int tiles_x = (W + LT - 1) / LT, tiles_y = (H + LT - 1) / LT;
    const int c = blockIdx.x / (tiles_x * tiles_y);
    const int tt = blockIdx.x % (tiles_x * tiles_y);
    const int tx = tt % tiles_x, ty = tt / tiles_x;
    const int x0 = tx * LT - LR, y0 = ty * LT - LR;
    const size_t plane = (size_t)H * W;
    for (int k = threadIdx.x; k < LH * LH; k += 256) {
        const int r = k / LH, q = k - r * LH;
        const int gy = y0 + r, gx = x0 + q;
        const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;       // maps are zero outside the image
        const size_t o = c * plane + (size_t)gy * W + gx;
        sm[0][r][q] = in ? d_mu[o] : 0.0f; sm[1][r][q] = in ? d_e11[o] : 0.0f; sm[2][r][q] = in ? d_e12[o] : 0.0f;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < LH * LT; k += 256) {
        const int r = k / LT, q = k - r * LT;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int j = 0; j < LW; j++) {
            const float wv = win.w[j];
            a0 = __builtin_fmaf(wv, sm[0][r][q + j], a0); a1 = __builtin_fmaf(wv, sm[1][r][q + j], a1); a2 = __builtin_fmaf(wv, sm[2][r][q + j], a2);
        }
        hh[0][r][q] = a0; hh[1][r][q] = a1; hh[2][r][q] = a2;
    }
    __syncthreads();
    const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
    const int gx = tx * LT + lx, gy = ty * LT + ly;
    if (gx < W && gy < H) {
        float b0 = 0.f, b1 = 0.f, b2 = 0.f;
#pragma unroll
        for (int j = 0; j < LW; j++) {
            const float wv = win.w[j];
            b0 = __builtin_fmaf(wv, hh[0][ly + j][lx], b0); b1 = __builtin_fmaf(wv, hh[1][ly + j][lx], b1); b2 = __builtin_fmaf(wv, hh[2][ly + j][lx], b2);
        }
        const size_t o = c * plane + (size_t)gy * W + gx;
        const float x = img[o], y = gt[o];
        const float dssim = b0 + 2.0f * x * b1 + y * b2;
        const float d = x - y;
        const float sgn = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);
        const float up = upstream ? upstream[0] : 1.0f;
        dL_dimg[o] = up * inv_count * ((1.0f - lambda) * sgn - lambda * dssim);
    }
}

} // namespace gsrast
