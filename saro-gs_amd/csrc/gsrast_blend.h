// gsrast_blend.h -- per-tile alpha blending, forward (front-to-back) and backward (back-to-front).
//
// One 256-lane workgroup (4 wave64) per 16x16 tile -- the tile size is pinned by key parity with
// the reference (config.h:16-17).  Wave w owns pixel rows 4w..4w+3 of the tile, so a wave is a
// 16x4 pixel strip.  The tile's depth-sorted instance list is staged through LDS in batches of 256
// (48-byte records gathered with three 16-byte loads per lane); inside a batch every lane of a
// wave reads the SAME record (LDS broadcast) and evaluates it for its own pixel.  A wave leaves a
// batch as soon as all its lanes are saturated (exec mask empty), the workgroup leaves when every
// wave has -- no block-wide counting, only a barrier-and vote per batch.
//
// Reference behaviour restated: forward.cu:261-393 (renderCUDA fwd), backward.cu:399-557
// (renderCUDA bwd).  Order of tests per (pixel, instance) is parity-critical and kept:
//   contributor++ -> power > 0 skip -> alpha = min(0.99, o*exp(power)) -> alpha < 1/255 skip ->
//   test_T = T(1-alpha) < 1e-4 => done (without updating last_contributor) -> accumulate ->
//   median depth when T > 0.5 && test_T < 0.5.
#pragma once
#include "gsrast_common.h"

namespace gsrast {

// XCD-aware block -> tile map: consecutive workgroups are dealt round-robin to the 8 XCDs, so give
// each XCD one contiguous band of tiles (neighbouring tiles share Gaussians -> share that XCD's L2).
__device__ __forceinline__ uint32_t xcd_tile(uint32_t bid, uint32_t ntiles)
{
    const uint32_t per = (ntiles + 7) / 8;
    const uint32_t t = (bid & 7u) * per + (bid >> 3);
    return t;   // may be >= ntiles for the padded grid; caller checks
}

template <int EXPMODE>
__global__ void __launch_bounds__(256)
blend_fwd_kernel(const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, int W, int H,
                 int gx, uint32_t ntiles, const float4* __restrict__ rec0, const float4* __restrict__ rec1,
                 const float4* __restrict__ rec2, const float* __restrict__ bg,
                 float* __restrict__ out_color, float* __restrict__ out_depth, float* __restrict__ final_T,
                 uint32_t* __restrict__ n_contrib, uint32_t* __restrict__ tile_max)
{
    __shared__ float4 s0[256];
    __shared__ float4 s1[256];
    __shared__ float4 s2[256];
    __shared__ uint32_t s_max;

    const uint32_t tile = xcd_tile(blockIdx.x, ntiles);
    if (tile >= ntiles) return;
    const uint32_t tx = tile % (uint32_t)gx, ty = tile / (uint32_t)gx;
    const uint32_t t = threadIdx.x;
    const uint32_t px = tx * TILE_X + (t & 15u), py = ty * TILE_Y + (t >> 4);
    const bool inside = px < (uint32_t)W && py < (uint32_t)H;
    const float pxf = (float)px, pyf = (float)py;
    const uint2 range = ranges[tile];
    const uint32_t n = range.y - range.x;
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
    if (t == 0) s_max = 0;

    float T = 1.0f, C0 = 0.0f, C1 = 0.0f, C2 = 0.0f, Dm = 15.0f;
    uint32_t last = 0;
    bool done = !inside;

    for (uint32_t base = 0; base < n; base += 256) {
        if (__syncthreads_and(done)) break;
        const uint32_t i = base + t;
        if (i < n) {
            const uint32_t g = point_list[range.x + i];
            s0[t] = rec0[g]; s1[t] = rec1[g]; s2[t] = rec2[g];
        }
        __syncthreads();
        const uint32_t cnt = (n - base) < 256u ? (n - base) : 256u;
        for (uint32_t j = 0; !done && j < cnt; j++) {
            const float4 a = s0[j];
            const float4 c = s2[j];
            const float dx = a.x - pxf, dy = a.y - pyf;
            const float4 b = s1[j];
            const float power = gs_power(a.z, a.w, b.x, dx, dy);
            if (power > 0.0f || power < c.z) continue;   // c.z: conservative "alpha < 1/255" pre-test
            float alpha = b.y * gs_exp<EXPMODE>(power);
            alpha = alpha < 0.99f ? alpha : 0.99f;
            if (alpha < 1.0f / 255.0f) continue;
            const float test_T = T * (1.0f - alpha);
            if (test_T < 0.0001f) { done = true; continue; }
            const float w = alpha * T;
            C0 = __builtin_fmaf(b.z, w, C0);
            C1 = __builtin_fmaf(b.w, w, C1);
            C2 = __builtin_fmaf(c.x, w, C2);
            if (T > 0.5f && test_T < 0.5f) Dm = c.y;
            T = test_T;
            last = base + j + 1;
        }
    }
    if (inside) {
        const size_t pid = (size_t)W * py + px;
        const size_t plane = (size_t)W * H;
        final_T[pid] = T;
        n_contrib[pid] = last;
        out_color[pid] = __builtin_fmaf(T, bg0, C0);
        out_color[plane + pid] = __builtin_fmaf(T, bg1, C1);
        out_color[2 * plane + pid] = __builtin_fmaf(T, bg2, C2);
        out_depth[pid] = Dm;
    }
    // deepest list position consumed by any pixel of the tile (bounds the backward traversal)
    uint32_t m = last;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { uint32_t o = __shfl_xor(m, d, 64); m = o > m ? o : m; }
    __syncthreads();
    if (lane_id() == 0) atomicMax(&s_max, m);
    __syncthreads();
    if (t == 0) tile_max[tile] = s_max;
}

// Backward.  Per (pixel, instance) contribution -> 9 partial derivatives; each is summed over the
// 64 lanes of the wave with DPP row shifts / row broadcasts (no LDS, no shuffles) and committed
// with ONE hardware float atomic per wave per quantity -- instead of the reference's 9 atomics per
// (pixel, instance) pair (backward.cu:523-554).  A wave in which no lane contributes skips both
// the reduction and the atomics.
template <int EXPMODE>
__global__ void __launch_bounds__(256)
blend_bwd_kernel(const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, int W, int H,
                 int gx, uint32_t ntiles, const float4* __restrict__ rec0, const float4* __restrict__ rec1,
                 const float4* __restrict__ rec2, const float* __restrict__ bg,
                 const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
                 const uint32_t* __restrict__ tile_max, const float* __restrict__ dL_dpix,
                 float* __restrict__ dL_dmean2D /*[P][3]*/, float* __restrict__ dL_dconic /*[P][4]*/,
                 float* __restrict__ dL_dopacity /*[P]*/, float* __restrict__ dL_dcolors /*[P][3]*/)
{
#pragma clang fp contract(fast)
    __shared__ float4 s0[256];
    __shared__ float4 s1[256];
    __shared__ float s2[256];
    __shared__ uint32_t sid[256];

    const uint32_t tile = xcd_tile(blockIdx.x, ntiles);
    if (tile >= ntiles) return;
    const uint32_t tx = tile % (uint32_t)gx, ty = tile / (uint32_t)gx;
    const uint32_t t = threadIdx.x;
    const uint32_t px = tx * TILE_X + (t & 15u), py = ty * TILE_Y + (t >> 4);
    const bool inside = px < (uint32_t)W && py < (uint32_t)H;
    const float pxf = (float)px, pyf = (float)py;
    const uint2 range = ranges[tile];
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
    const uint32_t n_all = range.y - range.x;
    const uint32_t tm = tile_max[tile];
    const uint32_t n = tm < n_all ? tm : n_all;       // instances at list position >= n touch no pixel
    const size_t pid = (size_t)W * py + px;
    const size_t plane = (size_t)W * H;

    const float T_final = inside ? final_T[pid] : 0.0f;
    float T = T_final;
    const uint32_t last = inside ? n_contrib[pid] : 0u;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, lc0 = 0.f, lc1 = 0.f, lc2 = 0.f, last_alpha = 0.f;
    float dp0 = 0.f, dp1 = 0.f, dp2 = 0.f;
    if (inside) { dp0 = dL_dpix[pid]; dp1 = dL_dpix[plane + pid]; dp2 = dL_dpix[2 * plane + pid]; }
    const float bg_dot = bg0 * dp0 + bg1 * dp1 + bg2 * dp2;
    const float ddelx_dx = 0.5f * (float)W, ddely_dy = 0.5f * (float)H;
    const unsigned lane = lane_id();

    // position `pos` (0-based from the FRONT of the tile list) is visited from n-1 down to 0
    for (uint32_t base = 0; base < n; base += 256) {
        __syncthreads();
        const uint32_t i = base + t;
        if (i < n) {
            const uint32_t g = point_list[range.x + (n - 1 - i)];
            sid[t] = g;
            s0[t] = rec0[g]; s1[t] = rec1[g]; s2[t] = rec2[g].x;
        }
        __syncthreads();
        const uint32_t cnt = (n - base) < 256u ? (n - base) : 256u;
        for (uint32_t j = 0; j < cnt; j++) {
            const uint32_t pos = n - 1 - (base + j);
            // wave-uniform skip: nobody in this wave reaches this deep
            bool active = pos < last;
            if (!__any(active)) continue;
            const float4 a = s0[j];
            const float4 b = s1[j];
            const float dx = a.x - pxf, dy = a.y - pyf;
            const float power = gs_power(a.z, a.w, b.x, dx, dy);
            const float G = gs_exp<EXPMODE>(power);
            float alpha = b.y * G;
            alpha = alpha < 0.99f ? alpha : 0.99f;
            active = active && !(power > 0.0f) && !(alpha < 1.0f / 255.0f);
            if (!__any(active)) continue;

            float g_mx = 0.f, g_my = 0.f, g_ca = 0.f, g_cb = 0.f, g_cc = 0.f, g_op = 0.f, g_r = 0.f, g_g = 0.f, g_b = 0.f;
            if (active) {
                T = T * __builtin_amdgcn_rcpf(1.0f - alpha);
                const float dch = alpha * T;
                const float c0 = b.z, c1 = b.w, c2 = s2[j];
                acc0 = last_alpha * lc0 + (1.f - last_alpha) * acc0; lc0 = c0;
                acc1 = last_alpha * lc1 + (1.f - last_alpha) * acc1; lc1 = c1;
                acc2 = last_alpha * lc2 + (1.f - last_alpha) * acc2; lc2 = c2;
                float dL_dalpha = (c0 - acc0) * dp0 + (c1 - acc1) * dp1 + (c2 - acc2) * dp2;
                g_r = dch * dp0; g_g = dch * dp1; g_b = dch * dp2;
                dL_dalpha *= T;
                last_alpha = alpha;
                dL_dalpha += (-T_final * __builtin_amdgcn_rcpf(1.f - alpha)) * bg_dot;
                const float dL_dG = b.y * dL_dalpha;
                const float gdx = G * dx, gdy = G * dy;
                const float dG_ddelx = -gdx * a.z - gdy * a.w;
                const float dG_ddely = -gdy * b.x - gdx * a.w;
                g_mx = dL_dG * dG_ddelx * ddelx_dx;
                g_my = dL_dG * dG_ddely * ddely_dy;
                g_ca = -0.5f * gdx * dx * dL_dG;
                g_cb = -0.5f * gdx * dy * dL_dG;
                g_cc = -0.5f * gdy * dy * dL_dG;
                g_op = G * dL_dalpha;
            }
            g_mx = wave_sum_to_lane63(g_mx); g_my = wave_sum_to_lane63(g_my);
            g_ca = wave_sum_to_lane63(g_ca); g_cb = wave_sum_to_lane63(g_cb); g_cc = wave_sum_to_lane63(g_cc);
            g_op = wave_sum_to_lane63(g_op);
            g_r = wave_sum_to_lane63(g_r); g_g = wave_sum_to_lane63(g_g); g_b = wave_sum_to_lane63(g_b);
            if (lane == 63) {
                const uint32_t gid = sid[j];
                atomicAdd(&dL_dmean2D[3 * (size_t)gid], g_mx);
                atomicAdd(&dL_dmean2D[3 * (size_t)gid + 1], g_my);
                atomicAdd(&dL_dconic[4 * (size_t)gid], g_ca);
                atomicAdd(&dL_dconic[4 * (size_t)gid + 1], g_cb);
                atomicAdd(&dL_dconic[4 * (size_t)gid + 3], g_cc);
                atomicAdd(&dL_dopacity[gid], g_op);
                atomicAdd(&dL_dcolors[3 * (size_t)gid], g_r);
                atomicAdd(&dL_dcolors[3 * (size_t)gid + 1], g_g);
                atomicAdd(&dL_dcolors[3 * (size_t)gid + 2], g_b);
            }
        }
    }
}

} // namespace gsrast
