// gsrast_hexplane.h -- mip-mapped feature-plane lookup of the scale-aware residual field, all planes and scales of the
// field in ONE forward launch, and its backward (first item).  Replaces the un-vendored dependency
// call `nvdiffrast.torch.texture(grid, coords, mip_level_bias=levels, boundary_mode="clamp", max_mip_level=7|0)` at
// /root/reference/scene/hexplane.py:49-56 and the plane loop around it (hexplane.py:95-139: six planes summed per scale,
// scales concatenated).  The published algorithm of that op is restated in oracle/texture_oracle.py (header there):
//   mip stack   level l+1 = 2x2 box average of level l (2x1 / 1x2 once an extent is 1), built on every call
//   level       flevel = clamp(bias, 0, n_levels), level0 = floor, second level + lerp only where flevel > 0
//   per level   u = uv.x * w - 0.5 clamped to [0, w-1]; i1 = i0 + 1 unless the clamp hit; bilinear; a + f * (b - a)
//   backward    texel gradients of both levels, pulled down the stack (transpose of the box average); uv and bias
//               gradients on request (the reference detaches all three inputs: saro_gaussian.py:780)
// Layout: a plane is channel-last [H][W][C] fp32 (what hexplane.py:35 produces by permute + contiguous; parameters kept in
// torch.channels_last need no copy).  C = 32 floats = one 128-byte texel: the forward gives a point C/4 lanes, each
// fetching a float4 of every texel, so a texel is one full cache line per lane group.  The work is a gather: 8 texels
// x 128 B per point and plane (6 kB per point for six planes) against 16 B of coordinates and 128 B of output, bound by
// the L2 / Infinity-Cache gather rate, not by HBM streaming.
// Backward to the texels: sorted runs accumulated in registers (default, see hex_grad_tex_sorted_kernel) or plain global
// float atomics, a lane per channel (option hexplane_scatter = 1; also the reference point of the measurement); then one
// pull-down launch per level.  (An LDS-resident pyramid with ds_add_f32 was built and measured slower than either.)
#pragma once
#include "gsrast_common.h"

namespace gsrast {

constexpr int HEX_MAX_PLANES = 24;     // 6 planes x 4 scales (arguments/__init__.py:89 multires [1, 2, 4, 8])
constexpr int HEX_MAX_LEVELS = 16;
struct HexPlane {
    const float* tex;      // level 0 values [H][W][C]
    float* mips;           // levels >= 1, packed one after the other (values)
    float* grad;           // backward: level-0 gradient [H][W][C]
    float* gmips;          // backward: gradient of levels >= 1, packed like mips
    int W, H;
    int cu, cv;            // columns of pts / levels holding this plane's u and v
    int n_levels;          // built levels above 0 (the clamp of flevel)
    int out_offset;        // first channel of the plane's block in a feature row
    int pad_[2];
};
struct HexArgs { int n_planes, C, N, D, F, pad_; HexPlane pl[HEX_MAX_PLANES]; };

__host__ __device__ inline int hex_extent(int e, int l) { const int s = e >> l; return s > 1 ? s : 1; }
// texel offset of level l (>= 1) inside the packed stack of levels >= 1
__host__ __device__ inline unsigned hex_level_offset(int W, int H, int l)
{
    unsigned off = 0;
    for (int k = 1; k < l; k++) off += (unsigned)hex_extent(W, k) * (unsigned)hex_extent(H, k);
    return off;
}

// ---- mip stack ----------------------------------------------------------------------------------------------
// One launch per level, all planes that have it (blockIdx.y = plane).  A lane = one float4 of one output texel.
__global__ void __launch_bounds__(256)
hex_mip_build_kernel(HexArgs a, int level)
{
    const HexPlane& P = a.pl[blockIdx.y];
    if (P.n_levels < level) return;
    const int q4 = a.C >> 2;
    const int w = hex_extent(P.W, level), h = hex_extent(P.H, level);
    const int pw = hex_extent(P.W, level - 1), ph = hex_extent(P.H, level - 1);
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (unsigned)(w * h * q4)) return;
    const int q = i % q4, t = i / q4, x = t % w, y = t / w;
    const float4* in = reinterpret_cast<const float4*>(level == 1 ? P.tex : P.mips + (size_t)hex_level_offset(P.W, P.H, level - 1) * a.C);
    float4* out = reinterpret_cast<float4*>(P.mips + (size_t)hex_level_offset(P.W, P.H, level) * a.C);
    float4 r;
    if (pw > 1 && ph > 1) {
        const float4 A = in[((size_t)(2 * y) * pw + 2 * x) * q4 + q], B = in[((size_t)(2 * y) * pw + 2 * x + 1) * q4 + q];
        const float4 Cc = in[((size_t)(2 * y + 1) * pw + 2 * x) * q4 + q], D = in[((size_t)(2 * y + 1) * pw + 2 * x + 1) * q4 + q];
        r = make_float4(0.25f * (A.x + B.x + Cc.x + D.x), 0.25f * (A.y + B.y + Cc.y + D.y), 0.25f * (A.z + B.z + Cc.z + D.z), 0.25f * (A.w + B.w + Cc.w + D.w));
    } else {
        const size_t i0 = pw > 1 ? (size_t)y * pw + 2 * x : (size_t)(2 * y) * pw + x;
        const size_t i1 = pw > 1 ? i0 + 1 : i0 + pw;
        const float4 A = in[i0 * q4 + q], B = in[i1 * q4 + q];
        r = make_float4(0.5f * (A.x + B.x), 0.5f * (A.y + B.y), 0.5f * (A.z + B.z), 0.5f * (A.w + B.w));
    }
    out[(size_t)t * q4 + q] = r;
}

// Transpose of the build, top level first: children += weight * parent.  Distinct parents own distinct children, so no atomics.
__global__ void __launch_bounds__(256)
hex_mip_pull_kernel(HexArgs a, int level)
{
    const HexPlane& P = a.pl[blockIdx.y];
    if (P.n_levels < level) return;
    const int q4 = a.C >> 2;
    const int w = hex_extent(P.W, level), h = hex_extent(P.H, level);
    const int pw = hex_extent(P.W, level - 1), ph = hex_extent(P.H, level - 1);
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (unsigned)(w * h * q4)) return;
    const int q = i % q4, t = i / q4, x = t % w, y = t / w;
    const float4 g = reinterpret_cast<const float4*>(P.gmips + (size_t)hex_level_offset(P.W, P.H, level) * a.C)[(size_t)t * q4 + q];
    float4* out = reinterpret_cast<float4*>(level == 1 ? P.grad : P.gmips + (size_t)hex_level_offset(P.W, P.H, level - 1) * a.C);
    const bool quad = pw > 1 && ph > 1;
    const float wgt = quad ? 0.25f : 0.5f;
    const float4 s = make_float4(wgt * g.x, wgt * g.y, wgt * g.z, wgt * g.w);
    size_t idx[4]; int n;
    if (quad) { idx[0] = (size_t)(2 * y) * pw + 2 * x; idx[1] = idx[0] + 1; idx[2] = idx[0] + pw; idx[3] = idx[2] + 1; n = 4; }
    else if (pw > 1) { idx[0] = (size_t)y * pw + 2 * x; idx[1] = idx[0] + 1; n = 2; }
    else { idx[0] = (size_t)(2 * y) * pw + x; idx[1] = idx[0] + pw; n = 2; }
    for (int k = 0; k < n; k++) {
        float4 o = out[idx[k] * q4 + q];
        o.x += s.x; o.y += s.y; o.z += s.z; o.w += s.w;
        out[idx[k] * q4 + q] = o;
    }
}

// ---- per-point addressing -----------------------------------------------------------------------------------
struct HexTap { unsigned i00, i10, i01, i11; float fu, fv; };     // texel indices inside the level, bilinear fractions
__device__ inline HexTap hex_tap(float u, float v, int w, int h)
{
    float x = u * (float)w - 0.5f, y = v * (float)h - 0.5f;
    x = fminf(fmaxf(x, 0.0f), (float)(w - 1));                      // NaN -> 0: indices stay inside the level
    y = fminf(fmaxf(y, 0.0f), (float)(h - 1));
    const bool cx = x == 0.0f || x == (float)(w - 1), cy = y == 0.0f || y == (float)(h - 1);
    const int ix = (int)floorf(x), iy = (int)floorf(y);
    const int jx = ix + (cx ? 0 : 1), jy = iy + (cy ? 0 : 1);
    HexTap t;
    t.i00 = (unsigned)(iy * w + ix); t.i10 = (unsigned)(iy * w + jx); t.i01 = (unsigned)(jy * w + ix); t.i11 = (unsigned)(jy * w + jx);
    t.fu = x - (float)ix; t.fv = y - (float)iy;
    return t;
}
struct HexLevel { int l0, l1; float f; bool two; };
__device__ inline HexLevel hex_level(float bias, int n_levels)
{
    HexLevel L;
    const float fl = fminf(fmaxf(bias, 0.0f), (float)n_levels);
    L.l0 = (int)floorf(fl);
    L.two = fl > 0.0f;
    L.l1 = L.two ? (L.l0 + 1 < n_levels ? L.l0 + 1 : n_levels) : 0;
    L.f = L.two ? fl - (float)L.l0 : 0.0f;
    return L;
}
// (explicit FMAs: the library is built with -ffp-contract=off for the rasterizer's bit-exactness; the forward is VALU-issue bound)
__device__ inline float4 hex_bilerp(const float4* __restrict__ lv, const HexTap& t, unsigned q4, unsigned q)
{
    const float4 a00 = lv[t.i00 * q4 + q], a10 = lv[t.i10 * q4 + q];        // 32-bit indices: a level has < 2^26 texels of <= 16 float4
    const float4 a01 = lv[t.i01 * q4 + q], a11 = lv[t.i11 * q4 + q];
    float4 r;
#define GS_BL(c) { const float top = __builtin_fmaf(t.fu, a10.c - a00.c, a00.c), bot = __builtin_fmaf(t.fu, a11.c - a01.c, a01.c); r.c = __builtin_fmaf(t.fv, bot - top, top); }
    GS_BL(x) GS_BL(y) GS_BL(z) GS_BL(w)
#undef GS_BL
    return r;
}

// ---- forward ------------------------------------------------------------------------------------------------
// C/4 lanes per point (a float4 of channels each); every plane of a feature block is summed in registers, one store.
__global__ void __launch_bounds__(256)
hex_sample_fwd_kernel(HexArgs a, const float* __restrict__ pts, const float* __restrict__ levels, float* __restrict__ features)
{
    __shared__ unsigned level_off[HEX_MAX_PLANES][HEX_MAX_LEVELS];      // float4 offset of every level inside the plane's stack
    const unsigned q4 = (unsigned)a.C >> 2;
    for (int i = threadIdx.x; i < a.n_planes * HEX_MAX_LEVELS; i += blockDim.x) {
        const int p = i / HEX_MAX_LEVELS, l = i % HEX_MAX_LEVELS;
        level_off[p][l] = l <= a.pl[p].n_levels ? hex_level_offset(a.pl[p].W, a.pl[p].H, l) * (unsigned)a.C : 0u;
    }
    __syncthreads();
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n = gid / q4;
    const unsigned q = (unsigned)(gid % q4);
    if (n >= (size_t)a.N) return;
    const float* pn = pts + n * a.D;
    const float* ln = levels + n * a.D;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
    for (int p = 0; p < a.n_planes; p++) {
        const HexPlane& P = a.pl[p];
        const float u = pn[P.cu], v = pn[P.cv];
        const HexLevel L = hex_level(fminf(ln[P.cu], ln[P.cv]), P.n_levels);
        const int w0 = hex_extent(P.W, L.l0), h0 = hex_extent(P.H, L.l0);
        const float4* lv0 = reinterpret_cast<const float4*>(L.l0 ? P.mips + level_off[p][L.l0] : P.tex);
        float4 r = hex_bilerp(lv0, hex_tap(u, v, w0, h0), q4, q);
        if (L.two) {
            const int w1 = hex_extent(P.W, L.l1), h1 = hex_extent(P.H, L.l1);
            const float4* lv1 = reinterpret_cast<const float4*>(L.l1 ? P.mips + level_off[p][L.l1] : P.tex);
            const float4 b = hex_bilerp(lv1, hex_tap(u, v, w1, h1), q4, q);
            r.x = __builtin_fmaf(L.f, b.x - r.x, r.x); r.y = __builtin_fmaf(L.f, b.y - r.y, r.y);
            r.z = __builtin_fmaf(L.f, b.z - r.z, r.z); r.w = __builtin_fmaf(L.f, b.w - r.w, r.w);
        }
        acc.x += r.x; acc.y += r.y; acc.z += r.z; acc.w += r.w;
        if (p + 1 == a.n_planes || a.pl[p + 1].out_offset != P.out_offset) {
            reinterpret_cast<float4*>(features + n * a.F + P.out_offset)[q] = acc;
            acc = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
}

// ---- backward to the texels, global atomics -------------------------------------------------------------------
// A lane per channel: a wave's atomic instruction covers 64 / C whole texels (C = 32: two 128-byte texels).
__device__ inline void hex_scatter_global(float* __restrict__ lv, const HexTap& t, int C, int c, float g)
{
    const float w00 = (1.0f - t.fu) * (1.0f - t.fv), w10 = t.fu * (1.0f - t.fv), w01 = (1.0f - t.fu) * t.fv, w11 = t.fu * t.fv;
    atomicAdd(lv + (size_t)t.i00 * C + c, g * w00);
    atomicAdd(lv + (size_t)t.i10 * C + c, g * w10);
    atomicAdd(lv + (size_t)t.i01 * C + c, g * w01);
    atomicAdd(lv + (size_t)t.i11 * C + c, g * w11);
}
__global__ void __launch_bounds__(256)
hex_grad_tex_global_kernel(HexArgs a, const float* __restrict__ pts, const float* __restrict__ levels, const float* __restrict__ dy)
{
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n = gid / a.C;
    const int c = (int)(gid % a.C);
    if (n >= (size_t)a.N) return;
    const float* pn = pts + n * a.D;
    const float* ln = levels + n * a.D;
#pragma unroll 1
    for (int p = 0; p < a.n_planes; p++) {
        const HexPlane& P = a.pl[p];
        const float g = dy[n * a.F + P.out_offset + c];
        if (g == 0.0f) continue;
        const float u = pn[P.cu], v = pn[P.cv];
        const HexLevel L = hex_level(fminf(ln[P.cu], ln[P.cv]), P.n_levels);
        float* lv0 = L.l0 ? P.gmips + (size_t)hex_level_offset(P.W, P.H, L.l0) * a.C : P.grad;
        hex_scatter_global(lv0, hex_tap(u, v, hex_extent(P.W, L.l0), hex_extent(P.H, L.l0)), a.C, c, g * (1.0f - L.f));
        if (L.two) {
            float* lv1 = L.l1 ? P.gmips + (size_t)hex_level_offset(P.W, P.H, L.l1) * a.C : P.grad;
            hex_scatter_global(lv1, hex_tap(u, v, hex_extent(P.W, L.l1), hex_extent(P.H, L.l1)), a.C, c, g * L.f);
        }
    }
}

// ---- backward to the texels, sorted runs (default) ------------------------------------------------------------------
// Float atomics are the whole cost of the scatter: measured on MI355X, 1 M points x 6 planes x 8 texels x 32 channels take
// 4.1 ms as global atomics (~365 G lane-atomics/s, whatever the contention) and 6.6 ms as LDS atomics (ds_add_f32 retires
// about one lane per 3.5 clocks).  So the atomics themselves have to go: every (plane, point) pair gets the key
//   plane | pyramid cell of its ORIGIN = (coarser level l1, floor texel (m, k) there)   [single-level points: (l0, texel)]
// and the pairs are radix-sorted by it.  All pairs of one key touch the same 2x2 texels of level l1 and the same 4x4 texels
// of level l1 - 1 (x0 = 2 x1 + 0.5, so floor(x0) - 2m is 0, 1 or 2).  A half-wave (lane = channel) walks a chunk of the sorted
// list, accumulates those 20 texels in registers and flushes them with one atomic each when the key changes: a 64x64 plane
// sees ~180 pairs per key, i.e. 20 atomics instead of 1440.  A lone pair flushes only the 8 texels it touched, so the
// scheme is never worse than direct atomics.
constexpr int HEX_RUN_CHUNK = 256;       // sorted pairs per half-wave (C = 32); 8 rounds of 32

// Everything the accumulation needs to know about a (plane, point) pair, computed where the point rows are read in order
// (the sorted walk would gather them at random) and fetched there as ONE 32-byte record.
struct __attribute__((aligned(32))) HexPair {
    float fuA, fvA, fuB, fvB, f;   // bilinear fractions at level lo - 1 (A) and lo (B); level lerp weight
    uint32_t code;                 // sx | sy << 2 | two << 4 | direct << 5: position of A's 2x2 inside the run's 4x4 window
    uint32_t gidx;                 // first float of the pair's dy block (host: N * F < 2^32)
    uint32_t n;                    // the point
};
__global__ void __launch_bounds__(256)
hex_keys_kernel(HexArgs a, int cell_bits, const float* __restrict__ pts, const float* __restrict__ levels,
                uint32_t* __restrict__ keys, uint32_t* __restrict__ vals, HexPair* __restrict__ pairs)
{
    const HexPlane& P = a.pl[blockIdx.y];
    const unsigned n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= (unsigned)a.N) return;
    const float* pn = pts + (size_t)n * a.D;
    const float* ln = levels + (size_t)n * a.D;
    const size_t e = (size_t)blockIdx.y * a.N + n;
    const float u = pn[P.cu], v = pn[P.cv];
    const HexLevel L = hex_level(fminf(ln[P.cu], ln[P.cv]), P.n_levels);
    const bool two = L.two && L.l1 != L.l0;
    const int lo = two ? L.l1 : L.l0;                       // the pair's origin level
    const int wb = hex_extent(P.W, lo), hb = hex_extent(P.H, lo);
    const HexTap tb = hex_tap(u, v, wb, hb);
    float fuA = 0.f, fvA = 0.f, f = 0.f;
    uint32_t code = 0;
    if (two) {
        const int wa = hex_extent(P.W, L.l0), ha = hex_extent(P.H, L.l0);
        const HexTap ta = hex_tap(u, v, wa, ha);
        fuA = ta.fu; fvA = ta.fv; f = L.f;
        const int sx = (int)(ta.i00 % (unsigned)wa) - 2 * (int)(tb.i00 % (unsigned)wb);
        const int sy = (int)(ta.i00 / (unsigned)wa) - 2 * (int)(tb.i00 / (unsigned)wb);
        code = 16u | (uint32_t)((sx & 3) | ((sy & 3) << 2));
        if (sx < 0 || sx > 2 || sy < 0 || sy > 2) code = 32u;       // outside the 4x4 window (never seen): direct atomics
    }
    const unsigned base = lo ? (unsigned)P.W * (unsigned)P.H + hex_level_offset(P.W, P.H, lo) : 0u;
    keys[e] = ((uint32_t)blockIdx.y << cell_bits) | (base + tb.i00);
    vals[e] = (uint32_t)e;
    reinterpret_cast<float4*>(pairs + e)[0] = make_float4(fuA, fvA, tb.fu, tb.fv);
    reinterpret_cast<float4*>(pairs + e)[1] = make_float4(f, __uint_as_float(code), __uint_as_float(n * (uint32_t)a.F + (uint32_t)P.out_offset), __uint_as_float(n));
}

struct HexRun {            // decoded key: where the 4x4 (level lo - 1) and 2x2 (level lo) footprints live
    float* pa; float* pb;  // level pointers (gradient stack), pa unused when lo == 0
    int wa, ha, wb, hb;    // extents of the two levels
    int m, k;              // origin texel at level lo
};

template <int C>
__global__ void __launch_bounds__(256)
hex_grad_tex_sorted_kernel(HexArgs a, int cell_bits, unsigned n_entries, const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                           const HexPair* __restrict__ pairs, const float* __restrict__ pts, const float* __restrict__ levels,
                           const float* __restrict__ dy, int ablate)
{
    constexpr int G = C;                                   // lanes per group: one per channel
    __shared__ HexPlane planes[HEX_MAX_PLANES];
    for (int i = threadIdx.x; i < a.n_planes * (int)(sizeof(HexPlane) / 4); i += blockDim.x)
        reinterpret_cast<uint32_t*>(planes)[i] = reinterpret_cast<const uint32_t*>(a.pl)[i];
    __syncthreads();
    const unsigned group = (blockIdx.x * blockDim.x + threadIdx.x) / G;
    const int c = threadIdx.x % G;
    const unsigned e0 = group * (unsigned)HEX_RUN_CHUNK;
    if (e0 >= n_entries) return;
    const unsigned e1 = e0 + HEX_RUN_CHUNK < n_entries ? e0 + HEX_RUN_CHUNK : n_entries;
    const uint32_t cell_mask = (1u << cell_bits) - 1u;

    float A[4][4], B[2][2];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int j = 0; j < 4; j++) A[r][j] = 0.0f;
    B[0][0] = B[0][1] = B[1][0] = B[1][1] = 0.0f;
    unsigned maskA = 0;
    uint32_t run_key = 0xFFFFFFFFu;
    HexRun R{};

    auto flush = [&]() {
        if (run_key == 0xFFFFFFFFu) return;
        if (ablate & 2) { maskA = 0; return; }
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (maskA & (1u << (r * 4 + j))) {
                    const int col = 2 * R.m + j, row = 2 * R.k + r;
                    if (col < R.wa && row < R.ha) atomicAdd(R.pa + ((size_t)row * R.wa + col) * C + c, A[r][j]);
                }
                A[r][j] = 0.0f;
            }
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int col = R.m + j, row = R.k + r;
                if (col < R.wb && row < R.hb) atomicAdd(R.pb + ((size_t)row * R.wb + col) * C + c, B[r][j]);
                B[r][j] = 0.0f;
            }
        maskA = 0;
    };

    for (unsigned eb = e0; eb < e1; eb += G) {
        // phase 1, lane = sorted pair: fetch the pair's record
        const unsigned e = eb + c;
        const bool have = e < e1;
        const uint32_t key = have ? keys[e] : 0xFFFFFFFFu;
        float fuA = 0.f, fvA = 0.f, fuB = 0.f, fvB = 0.f, f = 0.f;
        int code = 0;
        uint32_t gidx = 0, n = 0;
        if (have) {
            const float4* rec = reinterpret_cast<const float4*>(pairs + vals[e]);
            const float4 r0 = rec[0], r1 = rec[1];
            fuA = r0.x; fvA = r0.y; fuB = r0.z; fvB = r0.w; f = r1.x;
            code = (int)__float_as_uint(r1.y); gidx = __float_as_uint(r1.z); n = __float_as_uint(r1.w);
        }
        const int cnt = (int)((e1 - eb) < (unsigned)G ? (e1 - eb) : (unsigned)G);
        // phase 2, lane = channel: walk the pairs, four at a time so their dy rows are in flight together
        auto one = [&](int j, float g) {
            const uint32_t kj = (uint32_t)__shfl((int)key, j, G);
            const int cj = __shfl(code, j, G);
            const float fua = __shfl(fuA, j, G), fva = __shfl(fvA, j, G), fub = __shfl(fuB, j, G), fvb = __shfl(fvB, j, G), fj = __shfl(f, j, G);
            if (kj != run_key) {
                flush();
                run_key = kj;
                const HexPlane& P = planes[kj >> cell_bits];
                unsigned pc = kj & cell_mask;
                int lo = 0;
                const unsigned t0 = (unsigned)P.W * (unsigned)P.H;
                if (pc >= t0) {
                    pc -= t0; lo = 1;
                    for (;;) { const unsigned t = (unsigned)hex_extent(P.W, lo) * (unsigned)hex_extent(P.H, lo); if (pc < t) break; pc -= t; lo++; }
                }
                R.wb = hex_extent(P.W, lo); R.hb = hex_extent(P.H, lo);
                R.m = (int)(pc % (unsigned)R.wb); R.k = (int)(pc / (unsigned)R.wb);
                R.pb = lo ? P.gmips + (size_t)hex_level_offset(P.W, P.H, lo) * C : P.grad;
                R.wa = lo ? hex_extent(P.W, lo - 1) : 0; R.ha = lo ? hex_extent(P.H, lo - 1) : 0;
                R.pa = lo > 1 ? P.gmips + (size_t)hex_level_offset(P.W, P.H, lo - 1) * C : P.grad;
            }
            if (cj & 32) {                                   // safety net: this pair alone, straight to memory
                const uint32_t nj = (uint32_t)__shfl((int)n, j, G);
                const HexPlane& P = planes[kj >> cell_bits];
                const float u = pts[(size_t)nj * a.D + P.cu], v = pts[(size_t)nj * a.D + P.cv];
                const HexLevel L = hex_level(fminf(levels[(size_t)nj * a.D + P.cu], levels[(size_t)nj * a.D + P.cv]), P.n_levels);
                float* lv0 = L.l0 ? P.gmips + (size_t)hex_level_offset(P.W, P.H, L.l0) * C : P.grad;
                float* lv1 = L.l1 ? P.gmips + (size_t)hex_level_offset(P.W, P.H, L.l1) * C : P.grad;
                hex_scatter_global(lv0, hex_tap(u, v, hex_extent(P.W, L.l0), hex_extent(P.H, L.l0)), C, c, g * (1.0f - L.f));
                hex_scatter_global(lv1, hex_tap(u, v, hex_extent(P.W, L.l1), hex_extent(P.H, L.l1)), C, c, g * L.f);
                return;
            }
            // (explicit FMAs: the library is built with -ffp-contract=off for the rasterizer's bit-exactness)
            const bool two = (cj & 16) != 0;
            const float gb = two ? g * fj : g;
            const float ub = 1.0f - fub, vb = 1.0f - fvb;
            B[0][0] = __builtin_fmaf(gb, ub * vb, B[0][0]);  B[0][1] = __builtin_fmaf(gb, fub * vb, B[0][1]);
            B[1][0] = __builtin_fmaf(gb, ub * fvb, B[1][0]); B[1][1] = __builtin_fmaf(gb, fub * fvb, B[1][1]);
            if (two) {
                const float ga = g * (1.0f - fj);
                const float ua = 1.0f - fua, y0 = ga * (1.0f - fva), y1 = ga * fva;
                // the pair's 2x2 sits at (sx, sy) of the run's 4x4 window: one case per position keeps the window in registers
#define GS_ACC(SY, SX) case (SY) * 4 + (SX): \
                    A[SY][SX] = __builtin_fmaf(y0, ua, A[SY][SX]);         A[SY][SX + 1] = __builtin_fmaf(y0, fua, A[SY][SX + 1]); \
                    A[SY + 1][SX] = __builtin_fmaf(y1, ua, A[SY + 1][SX]); A[SY + 1][SX + 1] = __builtin_fmaf(y1, fua, A[SY + 1][SX + 1]); \
                    maskA |= (0x33u << (SX)) << (4 * (SY)); break;
                switch (cj & 15) {
                    GS_ACC(0, 0) GS_ACC(0, 1) GS_ACC(0, 2) GS_ACC(1, 0) GS_ACC(1, 1) GS_ACC(1, 2) GS_ACC(2, 0) GS_ACC(2, 1) GS_ACC(2, 2)
                    default: break;
                }
#undef GS_ACC
            }
        };
        for (int j0 = 0; j0 < cnt; j0 += 4) {
            float g4[4];
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const uint32_t gi = (uint32_t)__shfl((int)gidx, (j0 + t) & (G - 1), G);       // past cnt: some valid row, unused
                g4[t] = (ablate & 1) ? __uint_as_float(gi) : dy[(size_t)gi + c];
            }
#pragma unroll
            for (int t = 0; t < 4; t++) if (j0 + t < cnt) { if (ablate & 4) B[0][0] += g4[t]; else one(j0 + t, g4[t]); }
        }
    }
    flush();
}

// ---- backward to uv and bias (on request) ------------------------------------------------------------------------
// Same lane layout as the forward; the C/4 lanes of a point reduce their channel sums with DPP shuffles; lane 0 of the
// group accumulates over the planes and writes d_pts / d_levels rows (no atomics: the group owns the point).
__device__ inline void hex_slopes(const float4* __restrict__ lv, const HexTap& t, int q4, int q, int w, int h, const float4& g,
                                  float4& val, float& du, float& dv)
{
    const float4 a00 = lv[(size_t)t.i00 * q4 + q], a10 = lv[(size_t)t.i10 * q4 + q];
    const float4 a01 = lv[(size_t)t.i01 * q4 + q], a11 = lv[(size_t)t.i11 * q4 + q];
    du = 0.0f; dv = 0.0f;
#define GS_SL(c) { const float top = a00.c + t.fu * (a10.c - a00.c), bot = a01.c + t.fu * (a11.c - a01.c); val.c = top + t.fv * (bot - top); \
        du += g.c * (((a10.c - a00.c) * (1.0f - t.fv) + (a11.c - a01.c) * t.fv) * (float)w); \
        dv += g.c * (((a01.c - a00.c) * (1.0f - t.fu) + (a11.c - a10.c) * t.fu) * (float)h); }
    GS_SL(x) GS_SL(y) GS_SL(z) GS_SL(w)
#undef GS_SL
}
__global__ void __launch_bounds__(256)
hex_grad_uv_kernel(HexArgs a, const float* __restrict__ pts, const float* __restrict__ levels, const float* __restrict__ dy,
                   float* __restrict__ d_pts, float* __restrict__ d_levels)
{
    const int q4 = a.C >> 2;
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n_raw = gid / q4;
    const int q = (int)(gid % q4);
    const bool live = n_raw < (size_t)a.N;
    const size_t n = live ? n_raw : (size_t)a.N - 1;          // dead lanes shadow the last point: the shuffles need every lane
    const float* pn = pts + n * a.D;
    const float* ln = levels + n * a.D;
    float dp[8], dl[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { dp[k] = 0.0f; dl[k] = 0.0f; }
#pragma unroll 1
    for (int p = 0; p < a.n_planes; p++) {
        const HexPlane& P = a.pl[p];
        const float4 g = reinterpret_cast<const float4*>(dy + n * a.F + P.out_offset)[q];
        const float u = pn[P.cu], v = pn[P.cv];
        const float lu = ln[P.cu], lvv = ln[P.cv];
        const HexLevel L = hex_level(fminf(lu, lvv), P.n_levels);
        const int w0 = hex_extent(P.W, L.l0), h0 = hex_extent(P.H, L.l0);
        const float4* lv0 = reinterpret_cast<const float4*>(L.l0 ? P.mips + (size_t)hex_level_offset(P.W, P.H, L.l0) * a.C : P.tex);
        float4 va, vb = make_float4(0.f, 0.f, 0.f, 0.f);
        float du0, dv0, du1 = 0.0f, dv1 = 0.0f;
        hex_slopes(lv0, hex_tap(u, v, w0, h0), q4, q, w0, h0, g, va, du0, dv0);
        float db = 0.0f;
        if (L.two) {
            const int w1 = hex_extent(P.W, L.l1), h1 = hex_extent(P.H, L.l1);
            const float4* lv1 = reinterpret_cast<const float4*>(L.l1 ? P.mips + (size_t)hex_level_offset(P.W, P.H, L.l1) * a.C : P.tex);
            hex_slopes(lv1, hex_tap(u, v, w1, h1), q4, q, w1, h1, g, vb, du1, dv1);
            db = g.x * (vb.x - va.x) + g.y * (vb.y - va.y) + g.z * (vb.z - va.z) + g.w * (vb.w - va.w);
        }
        float du = (1.0f - L.f) * du0 + L.f * du1, dv = (1.0f - L.f) * dv0 + L.f * dv1;
        for (int m = 1; m < q4; m <<= 1) {                      // q4 is a power of two <= 16: the group sits inside a DPP row
            du += __shfl_xor(du, m); dv += __shfl_xor(dv, m); db += __shfl_xor(db, m);
        }
        // the bias is min(levels[cu], levels[cv]): its gradient goes to the smaller one (cu on a tie, as torch.min does)
        const int cb = lvv < lu ? P.cv : P.cu;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            dp[k] += (k == P.cu ? du : 0.0f) + (k == P.cv ? dv : 0.0f);
            dl[k] += k == cb ? db : 0.0f;
        }
    }
    if (live && q == 0) {
        for (int k = 0; k < a.D && k < 8; k++) {
            if (d_pts) d_pts[n * a.D + k] = dp[k];
            if (d_levels) d_levels[n * a.D + k] = dl[k];
        }
    }
}

} // namespace gsrast
