// gsrast_knn.h -- mean squared distance of every point to its 3 nearest neighbours (second item).
// Replaces `simple_knn._C.distCUDA2`, an un-vendored dependency of the reference (imported at
// /root/reference/scene/saro_gaussian.py:21, used at :187 to initialise the scales:
//     dist2 = clamp_min(distCUDA2(points), 1e-7);  scales = log(sqrt(dist2))).
// The package is not in /root/reference; its published algorithm (graphdeco-inria/simple-knn, the one 3DGS ships) is:
// Morton-order the points, take boxes of 1024 consecutive points with their bounding boxes, seed each point's three best
// squared distances from its neighbours in Morton order, then visit every box whose bounding box is closer than the
// current third-best distance.  That search is EXACT, so any exact 3-NN gives the same result up to fp32 rounding of
// (d1 + d2 + d3) / 3 -- the oracle in the tests is scipy's cKDTree in fp64.
#pragma once
#include "gsrast_common.h"

namespace gsrast {

constexpr int KNN_BOX = 1024;

__device__ __forceinline__ unsigned f2ord(float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float ord2f(unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u); }

// bbox[0..2] = min, [3..5] = max as order-preserving unsigned keys (initialised to ~0 / 0 by knn_init_kernel)
__global__ void knn_init_kernel(unsigned* bbox) { if (threadIdx.x < 3) bbox[threadIdx.x] = 0xFFFFFFFFu; else if (threadIdx.x < 6) bbox[threadIdx.x] = 0u; }

__global__ void __launch_bounds__(256)
knn_bbox_kernel(int P, const float* __restrict__ pts, unsigned* __restrict__ bbox)
{
    __shared__ unsigned smin[3][4], smax[3][4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    unsigned lo[3] = { 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu }, hi[3] = { 0u, 0u, 0u };
    if (i < P) {
#pragma unroll
        for (int k = 0; k < 3; k++) { const unsigned o = f2ord(pts[3 * (size_t)i + k]); lo[k] = o; hi[k] = o; }
    }
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const unsigned a = __shfl_xor(lo[k], d, 64), b = __shfl_xor(hi[k], d, 64);
            lo[k] = a < lo[k] ? a : lo[k]; hi[k] = b > hi[k] ? b : hi[k];
        }
    const unsigned lane = lane_id(), wave = threadIdx.x >> 6;
    if (lane == 0) { for (int k = 0; k < 3; k++) { smin[k][wave] = lo[k]; smax[k][wave] = hi[k]; } }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int k = threadIdx.x;
        unsigned a = smin[k][0], b = smax[k][0];
        for (int w = 1; w < 4; w++) { a = smin[k][w] < a ? smin[k][w] : a; b = smax[k][w] > b ? smax[k][w] : b; }
        atomicMin(&bbox[k], a); atomicMax(&bbox[3 + k], b);
    }
}

__device__ __forceinline__ unsigned spread10(unsigned x)
{   // 10 bits -> every third bit
    x = (x | (x << 16)) & 0x030000FFu;
    x = (x | (x << 8)) & 0x0300F00Fu;
    x = (x | (x << 4)) & 0x030C30C3u;
    x = (x | (x << 2)) & 0x09249249u;
    return x;
}
__global__ void __launch_bounds__(256)
knn_morton_kernel(int P, const float* __restrict__ pts, const unsigned* __restrict__ bbox, uint32_t* __restrict__ codes,
                  uint32_t* __restrict__ idx)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    unsigned c = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float lo = ord2f(bbox[k]), hi = ord2f(bbox[3 + k]);
        const float ext = hi - lo;
        float u = ext > 0.0f ? (pts[3 * (size_t)i + k] - lo) / ext : 0.0f;
        u = fminf(fmaxf(u * 1023.0f, 0.0f), 1023.0f);
        c |= spread10((unsigned)u) << (2 - k);
    }
    codes[i] = c; idx[i] = (uint32_t)i;
}

// bounding box of each run of KNN_BOX consecutive points in Morton order: {min xyz, max xyz}
__global__ void __launch_bounds__(256)
knn_boxes_kernel(int P, const float* __restrict__ pts, const uint32_t* __restrict__ order, float* __restrict__ boxes)
{
    __shared__ float smin[3][4], smax[3][4];
    const int b = blockIdx.x;
    float lo[3] = { 3.4e38f, 3.4e38f, 3.4e38f }, hi[3] = { -3.4e38f, -3.4e38f, -3.4e38f };
    for (int j = threadIdx.x; j < KNN_BOX; j += 256) {
        const int s = b * KNN_BOX + j;
        if (s < P) {
            const size_t g = order[s];
#pragma unroll
            for (int k = 0; k < 3; k++) { const float v = pts[3 * g + k]; lo[k] = fminf(lo[k], v); hi[k] = fmaxf(hi[k], v); }
        }
    }
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { lo[k] = fminf(lo[k], __shfl_xor(lo[k], d, 64)); hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], d, 64)); }
    const unsigned lane = lane_id(), wave = threadIdx.x >> 6;
    if (lane == 0) { for (int k = 0; k < 3; k++) { smin[k][wave] = lo[k]; smax[k][wave] = hi[k]; } }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int k = threadIdx.x;
        boxes[6 * (size_t)b + k] = fminf(fminf(smin[k][0], smin[k][1]), fminf(smin[k][2], smin[k][3]));
        boxes[6 * (size_t)b + 3 + k] = fmaxf(fmaxf(smax[k][0], smax[k][1]), fmaxf(smax[k][2], smax[k][3]));
    }
}

__device__ __forceinline__ void knn_insert(float d, float best[3])
{
    if (d < best[2]) {
        if (d < best[1]) { best[2] = best[1]; if (d < best[0]) { best[1] = best[0]; best[0] = d; } else best[1] = d; }
        else best[2] = d;
    }
}
__device__ __forceinline__ float knn_d2(const float p[3], const float q0, const float q1, const float q2)
{
    const float a = p[0] - q0, b = p[1] - q1, c = p[2] - q2;
    return a * a + b * b + c * c;
}

// one lane per point, in Morton order (neighbouring lanes visit the same boxes)
__global__ void __launch_bounds__(256)
knn_search_kernel(int P, const float* __restrict__ pts, const uint32_t* __restrict__ order, const float* __restrict__ boxes,
                  int nboxes, float* __restrict__ out)
{
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= P) return;
    const uint32_t self = order[s];
    const float p[3] = { pts[3 * (size_t)self], pts[3 * (size_t)self + 1], pts[3 * (size_t)self + 2] };
    float best[3] = { 3.402823466e38f, 3.402823466e38f, 3.402823466e38f };
    for (int j = s - 3; j <= s + 3; j++) {       // seed from the neighbours in Morton order
        if (j < 0 || j >= P || j == s) continue;
        const size_t g = order[j];
        knn_insert(knn_d2(p, pts[3 * g], pts[3 * g + 1], pts[3 * g + 2]), best);
    }
    // the seed only provides a rejection radius (the box scan below meets those neighbours again)
    const float reject = best[2];
    best[0] = best[1] = best[2] = 3.402823466e38f;
    for (int b = 0; b < nboxes; b++) {
        const float* bx = boxes + 6 * (size_t)b;
        float dd = 0.0f;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float lo = bx[k], hi = bx[3 + k];
            const float d = p[k] < lo ? lo - p[k] : (p[k] > hi ? p[k] - hi : 0.0f);
            dd += d * d;
        }
        if (dd > reject || dd > best[2]) continue;   // no point of this box can beat the current third-best
        const int j0 = b * KNN_BOX, j1 = (j0 + KNN_BOX) < P ? (j0 + KNN_BOX) : P;
        for (int j = j0; j < j1; j++) {
            if (j == s) continue;
            const size_t g = order[j];
            knn_insert(knn_d2(p, pts[3 * g], pts[3 * g + 1], pts[3 * g + 2]), best);
        }
    }
    out[self] = (best[0] + best[1] + best[2]) / 3.0f;
}

} // namespace gsrast
