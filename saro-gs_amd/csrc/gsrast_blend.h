// gsrast_blend.h -- per-tile alpha blending, forward (front-to-back) and backward (back-to-front).
//
// One 256-lane workgroup (4 wave64) per 16x16 tile -- the tile size is pinned by key parity with
// the reference (config.h:16-17).  A wave owns 64 pixels of the tile: an 8x8 block in the default forward kernel,
// a 16x4 strip (rows 4w..4w+3) in the backward and in the un-culled variants.  The tile's depth-sorted instance list
// is staged through LDS in batches (48-byte records gathered with three 16-byte loads per lane: rec0 {x, y, conic.x, conic.y},
// rec1 {conic.z, opacity, depth, skip threshold}, rec2 {r, g, b, -}; rec2 comes from the colour kernel, which runs beside the
// depth sort on its own stream).  Default kernels
// (*_cull_kernel): per round of 64 staged instances every lane tests ONE instance against the wave's pixel block
// (strip_may_touch), the ballot is a 64-bit scalar mask, and only the survivors are evaluated -- each by all lanes, one
// LDS broadcast read of the record, each lane for its own pixel.  A wave leaves a batch as soon as all its lanes are
// saturated, the workgroup leaves when every wave has -- no block-wide counting, only a barrier-and vote per batch.
// Workgroups take tiles heaviest-first (tile_from_buckets).
//
// Reference behaviour restated: forward.cu:261-393 (renderCUDA fwd), backward.cu:399-557
// (renderCUDA bwd).  Order of tests per (pixel, instance) is parity-critical and kept:
//   contributor++ -> power > 0 skip -> alpha = min(0.99, o*exp(power)) -> alpha < 1/255 skip ->
//   test_T = T(1-alpha) < 1e-4 => done (without updating last_contributor) -> accumulate ->
//   median depth when T > 0.5 && test_T < 0.5.
#pragma once
#include "gsrast_common.h"

#ifndef GSRAST_CUT_MARGIN_X4
#define GSRAST_CUT_MARGIN_X4 6u      // the next cut depth of a tile = that of the list entry (this / 4) x as deep as the deepest one consumed, + 32: 1.5 x.
                                     // Alternating runs, views/s with 1.25 x / 1.5 x / 2 x: 3 M 825-832 / 795-824 / 771-809, 1 M 1215-1279 / 1267-1280 / 1229-1242 -- the cut is in
                                     // DEPTH (cloned or split Gaussians in front of it shift list positions, not the depth at which a tile saturates), 1.5 x leaves room for pruning
#endif
namespace gsrast {

#if defined(GSRAST_DEBUG_COUNTERS) || defined(GSRAST_DEBUG_TIMING)
__device__ unsigned long long g_dbg[16];
#endif
#if defined(GSRAST_DEBUG_COUNTERS) && !defined(GSRAST_DEBUG_TIMING)
#define GS_COUNT(i, v) do { const unsigned long long v_ = (unsigned long long)(v); if (lane_id() == 0) atomicAdd(&g_dbg[i], v_); } while (0)
#else
#define GS_COUNT(i, v) do { } while (0)
#endif

// XCD-aware block -> tile map: consecutive workgroups are dealt round-robin to the 8 XCDs, so give
// each XCD one contiguous band of tiles (neighbouring tiles share Gaussians -> share that XCD's L2).
__device__ __forceinline__ uint32_t xcd_tile(uint32_t bid, uint32_t ntiles)
{
    const uint32_t per = (ntiles + 7) / 8;
    const uint32_t t = (bid & 7u) * per + (bid >> 3);
    return t;   // may be >= ntiles for the padded grid; caller checks
}

// Launch order without a sorting kernel: workgroup b belongs to XCD group b mod 8 and takes entry b / 8 of the concatenation of
// that group's 64 work buckets (gsrast_common.h: XCD_GROUPS, ImgLayout::bucket_cnt [8][64] / bucket_list [8][64][Tg]).  Every
// thread of the workgroup calls this; the result is >= ntiles for a workgroup past the end of its group's lists.
__device__ __forceinline__ uint32_t tile_from_buckets(const uint32_t* __restrict__ cnt, const uint16_t* __restrict__ list,
                                                      uint32_t Tg, uint32_t b, uint32_t* s_tile)
{
    const uint32_t xg = b % (uint32_t)XCD_GROUPS, s = b / (uint32_t)XCD_GROUPS;
    if (threadIdx.x < (unsigned)WORK_BUCKETS) {
        const uint32_t c = cnt[xg * WORK_BUCKETS + threadIdx.x];
        uint32_t incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d, 64); if (threadIdx.x >= (unsigned)d) incl += o; }
        const uint32_t excl = incl - c;
        if (s >= excl && s < incl) *s_tile = list[((size_t)xg * WORK_BUCKETS + threadIdx.x) * Tg + (s - excl)];
        if (threadIdx.x == (unsigned)WORK_BUCKETS - 1u && s >= incl) *s_tile = 0xFFFFFFFFu;
    }
    __syncthreads();
    return *s_tile;
}
// ... and the grouped producer (unused by the kernels of this file: the forward lists are filled by tile_ranges_from_runs_kernel)
__device__ __forceinline__ void bucket_append(uint32_t* __restrict__ cnt, uint16_t* __restrict__ list, uint32_t Tg, uint32_t gx, uint32_t tile, uint32_t work)
{
    const uint32_t idx = ((tile / gx) % (uint32_t)XCD_GROUPS) * WORK_BUCKETS + work_bucket(work);
    list[(size_t)idx * Tg + atomicAdd(&cnt[idx], 1u)] = (uint16_t)tile;
}

// The BACKWARD order stays one global heaviest-first sequence (64 buckets, [64][T] lists): grouping it by XCD as well cut the
// backward's fetched bytes by 30 % but made it 15 % SLOWER -- neighbouring tiles then run at the same time and their gradient
// atomics to the Gaussians they share collide.
__device__ __forceinline__ uint32_t tile_from_buckets_global(const uint32_t* __restrict__ cnt, const uint16_t* __restrict__ list,
                                                             uint32_t T, uint32_t b, uint32_t* s_tile)
{
    if (threadIdx.x < (unsigned)WORK_BUCKETS) {
        const uint32_t c = cnt[threadIdx.x];
        uint32_t incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d, 64); if (threadIdx.x >= (unsigned)d) incl += o; }
        const uint32_t excl = incl - c;
        if (b >= excl && b < incl) *s_tile = list[(size_t)threadIdx.x * T + (b - excl)];
    }
    __syncthreads();
    return *s_tile;
}
__device__ __forceinline__ void bucket_append_global(uint32_t* __restrict__ cnt, uint16_t* __restrict__ list, uint32_t T, uint32_t tile, uint32_t work)
{
    const uint32_t b = work_bucket(work);
    list[(size_t)b * T + atomicAdd(&cnt[b], 1u)] = (uint16_t)tile;
}

// ---------------------------------------------------------------------------------------------
// Wave-level culling of a staged batch against the wave's pixel strip.
// Lane l takes instance (round*64 + l) of the batch and computes the EXACT minimum over the strip's
// rectangle of  q(d) = 0.5*(A dx^2 + C dy^2) + B dx dy  (convex: the conic is positive definite), i.e.
// the largest power = -q any pixel of the strip can see.  If even that is below the instance's skip
// threshold no pixel of the strip can pass the per-pixel tests, so the instance is dropped for this
// wave with ~0.5 instructions instead of a ~20-instruction evaluated-and-skipped iteration.  The
// ballot of survivors is a 64-bit mask in SGPRs which the blend loop walks with s_ff1.
// Safety: skip_threshold sits 0.02 below ln(1/(255*opacity)) (preprocess_fwd_kernel); culling at
// threshold + 0.01 leaves a 0.01 band (in units of power) for the rounding of this bound, three
// orders of magnitude more than its error -- culled instances can never contribute, so results are
// bit-identical with and without culling.
__device__ __forceinline__ bool strip_may_touch(const float4 a, const float cz, const float thr,
                                                float x0, float x1, float y0, float y1)
{
    // d = mean - pixel:  dx in [a.x - x1, a.x - x0], dy in [a.y - y1, a.y - y0]
    const float dx_lo = a.x - x1, dx_hi = a.x - x0, dy_lo = a.y - y1, dy_hi = a.y - y0;
    const float dxc = __builtin_amdgcn_fmed3f(0.0f, dx_lo, dx_hi);      // box point closest to the centre
    const float dyc = __builtin_amdgcn_fmed3f(0.0f, dy_lo, dy_hi);
    const float A = a.z, B = a.w, C = cz;
    // candidate 1: on the edge dx = dxc, best dy;  candidate 2: on the edge dy = dyc, best dx
    const float dy1 = __builtin_amdgcn_fmed3f(-B * dxc * __builtin_amdgcn_rcpf(fmaxf(C, 1e-20f)), dy_lo, dy_hi);
    const float dx2 = __builtin_amdgcn_fmed3f(-B * dyc * __builtin_amdgcn_rcpf(fmaxf(A, 1e-20f)), dx_lo, dx_hi);
    const float q1 = 0.5f * (A * dxc * dxc + C * dy1 * dy1) + B * dxc * dy1;
    const float q2 = 0.5f * (A * dx2 * dx2 + C * dyc * dyc) + B * dx2 * dyc;
    const float qmin = fminf(q1, q2);
    // The box minimum above is exact for a positive definite conic only.  The reference rejects det == 0 and nothing else
    // (forward.cu:220): fp32 cancellation on needle-shaped Gaussians can leave an indefinite conic, which the blend still
    // evaluates pixel by pixel -- such an instance is never culled here (it stays bit-identical, just not skipped).
    const bool pd = A > 0.0f && C > 0.0f && A * C > B * B;
    return (pd ? -qmin >= thr + 0.01f : true) && !(thr > 0.0f);     // thr > 0: opacity below 1/255, no pixel can pass
}

// Forward blend, 4 wave64 per tile, one pixel per lane, with wave-level culling (see above).
template <int EXPMODE>
__device__ __forceinline__ void
blend_fwd_cull_body(const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                      const uint32_t* __restrict__ order, int W, int H,
                      int gx, uint32_t ntiles, const float4* __restrict__ rec0, const float4* __restrict__ rec1,
                      const float4* __restrict__ rec2, const float* __restrict__ bg,
                      float* __restrict__ out_color, float* __restrict__ out_depth, float* __restrict__ final_T,
                      uint32_t* __restrict__ n_contrib, uint32_t* __restrict__ tile_max,
                      uint32_t* __restrict__ bucket_cnt /* fwd [8][64] | bwd [64], or null */, uint16_t* __restrict__ bucket_list /* fwd [8][64][Tg] | bwd [64][T] */,
                      int order_from_buckets,
                      float4* __restrict__ zero4 /* or null */, uint32_t n_zero4 /* the backward's gradient records (GeomLayout::grec), zero-filled here */,
                      HintTable* __restrict__ hints /* or null: the context's launch-order hints (gsrast_common.h); this view's slot receives every tile's consumed depth */,
                      const uint32_t* __restrict__ hint_sel /* [2]: slot, valid */,
                      // list cut (gsrast_common.h): zcut_used = this call's snapshot of the pose's per-tile cut depths when the lists were
                      // built from the EARLY Gaussians only (null: full lists); a tile's list then counts as ending where its depth exceeds
                      // the tile's cut depth -- up to there it is the full list -- and a tile with a cut whose pixels are not all
                      // saturated at that point raises cut_scalars[SC_UNDONE].  pred: the predicated second launch over the full lists
                      const uint32_t* __restrict__ zcut_used = nullptr, uint32_t* __restrict__ cut_scalars = nullptr,
                      const uint32_t* __restrict__ pred = nullptr,
                      // round 4: a tile whose cut list was too short is flagged (tile_flags[tile] = 1) for the COMPLETION pass, which
                      // lists the Gaussians that touch such tiles -- all of them, late or not -- and blends those tiles again from
                      // their full lists (the launch with `pred`); every other tile is final after this launch
                      unsigned char* __restrict__ tile_flags = nullptr,
                      uint32_t cut_margin_x4 = GSRAST_CUT_MARGIN_X4 /* round 5: the context's margin (6 = 1.5 x; it widens while completion passes are reported) */,
                      unsigned char* __restrict__ untouched = nullptr /* GeomLayout::untouched: byte i cleared = some pixel consumed Gaussian i */)
{
    constexpr uint32_t FB = 256;                  // instances staged per batch (64 / 128 / 256 measured equal)
    if (pred && *pred == 0u) return;
    __shared__ float4 s0[FB];
    __shared__ float4 s1[FB];
    __shared__ float4 s2[FB];
    __shared__ uint32_t s_max;
    __shared__ uint32_t s_tile;

    // The 64 B / Gaussian zero-fill of the gradient records, spread over the workgroups of this VALU-bound kernel: a few 16-byte stores
    // per lane that nobody waits for.  (Round 2: a memset on the side stream behind the colour kernel, joined behind this kernel --
    // one more cross-stream wait, 10-20 us of idle queue in front of the backward's first kernel.)  Every workgroup of the launch does
    // its slice, also the ones that find no tile below.
    if (zero4) {
        const uint32_t per = (n_zero4 + gridDim.x - 1) / gridDim.x;
        const uint32_t q0 = blockIdx.x * per, q1 = min(n_zero4, q0 + per);
        // (non-temporal: 192 MB of zeros of which the backward touches the few per cent of rows some tile blended -- kept out of the
        // Infinity Cache they leave it to the arrays that ARE read again: +1.3 % views/s at 3 M)
        { typedef float v4f __attribute__((ext_vector_type(4)));
          for (uint32_t q = q0 + threadIdx.x; q < q1; q += 256u) __builtin_nontemporal_store(v4f{0.f, 0.f, 0.f, 0.f}, reinterpret_cast<v4f*>(zero4) + q); }
    }
    if (!order_from_buckets && blockIdx.x >= ntiles) return;
    // heaviest tiles first, per XCD group
    const uint32_t Tg = xcd_group_tiles((uint32_t)gx, ntiles);
    const uint32_t tile = order_from_buckets ? tile_from_buckets(bucket_cnt, bucket_list, Tg, blockIdx.x, &s_tile)
                                             : (order ? order[blockIdx.x] : blockIdx.x);
    if (tile >= ntiles) return;                  // uniform: past the end of this group's lists
    const uint32_t tx = tile % (uint32_t)gx, ty = tile / (uint32_t)gx;
    const uint32_t t = threadIdx.x;
    const unsigned lane = lane_id(), wave = t >> 6;
    // this wave's block of the tile (pixel centres): 8 columns x 8 rows.  (A 16 x 4 strip has the same 64 pixels but a
    // longer outline: 6 % more (wave, instance) pairs survive the culling test on the bench scene.)
    const uint32_t bx = (wave & 1u) * 8u, by = (wave >> 1) * 8u;
    const uint32_t px = tx * TILE_X + bx + (lane & 7u), py = ty * TILE_Y + by + (lane >> 3);
    const float sx0 = (float)(tx * TILE_X + bx), sx1 = sx0 + 7.0f;
    const float sy0 = (float)(ty * TILE_Y + by), sy1 = sy0 + 7.0f;
    const bool inside = px < (uint32_t)W && py < (uint32_t)H;
    const float pxf = (float)px, pyf = (float)py;
    const uint2 range = ranges[tile];
    const uint32_t n = range.y - range.x;
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
    if (t == 0) s_max = 0;
    // (a call in which the scatter marked NO Gaussian late has full lists whatever the snapshot says: nothing may be cut short, and
    // the host enqueues no second pass for it -- cut_scalars[SC_N_LATE] is final, the scatter ran before the binning)
    const uint32_t zc = (zcut_used && cut_scalars[SC_N_LATE] != 0u) ? zcut_used[tile] : ZCUT_NONE;       // uniform
    // (what the hint writer at the kernel's end needs of the table, requested HERE: a load behind the last barrier is a memory round trip during
    // which the workgroup holds its slot for nothing)
    uint32_t h_slot = 0u, h_found = 0u, h_zold = ZCUT_NONE;
    if (hints && t == 0) {
        h_slot = hint_sel[0]; h_found = hint_sel[1];
        if (h_found == 1u) h_zold = hint_zcut(hints, ntiles)[(size_t)h_slot * ntiles + tile];
    }
    uint32_t n_safe = n;                          // entries of the list known to be ALL the tile's Gaussians up to their depth
    bool cut_here = false;

    float T = 1.0f, C0 = 0.0f, C1 = 0.0f, C2 = 0.0f, Dm = 15.0f;
    uint32_t last = 0;
    // A finished pixel (T would drop below 1e-4, forward.cu:377-381) moves to x = FAR: its power becomes ~ -1e30 x conic
    // (finite: conic entries are <= 1/0.3), so it fails the skip test like any far pixel and needs no flag in the loop.
    // The set of live lanes is a 64-bit scalar, updated from the ballot of the (rare) terminations only.
    constexpr float FAR = 1.0e15f;
    float pxa = inside ? pxf : FAR;
    uint64_t alive = __ballot(inside);
    uint64_t m_above = alive;                     // lanes whose T is still above 0.5 (median-depth test below)
#if defined(GSRAST_DEBUG_COUNTERS) && !defined(GSRAST_DEBUG_TIMING)
    uint32_t dbg_h[2] = { 0u, 0u }, dbg_q[4] = { 0u, 0u, 0u, 0u };
#endif

    // (Measured and dropped, round 3: software-pipelined staging as in the backward -- the next batch's records requested behind the
    // barrier that opens a batch.  -0.3 ... -1.2 % per step at 3 M / 1 M / shell: most tiles saturate inside their first batch and
    // then have fetched 256 records for nothing; eight workgroups per CU hide the staging latency of the few heavy ones.
    // Also measured and dropped: TWO PHASES -- every tile blends its first 256 entries, unfinished tiles park their per-pixel state
    // and a second, persistent launch continues them heaviest-remaining-first (bit-identical outputs): blend_fwd 0.33 -> 0.37 ms at
    // 3 M, 0.22 -> 0.26 at 1 M, 0.23 -> 0.36 on the shell scene.  The heavy tiles' serial chain of batches must run UNDER the bulk of
    // the light tiles, not behind it: what helps is starting them first, i.e. knowing them -- see the launch-order hint below.
    // And: TWO surviving instances per round (both alphas evaluated side by side for instruction-level parallelism in the heavy
    // tiles' serial chain, updates applied in list order: bit-identical) -- -1.5 ... -2.7 % per step at every size: the second exp is
    // wasted whenever one of the two fails its tests; s_setprio 3 / 1 for the first workgroups of the heaviest-first order: nothing;
    // DECOUPLED waves (every wave stages its own 64-instance batches with a register double buffer, no workgroup barrier in the
    // loop, bit-identical): +0.3 ... +0.8 % on the cube at 0.1 / 1 / 3 M, -6.5 % on the shell scene, whose long consumed lists are then
    // staged four times -- the barriers are not what the waves wait for.)
    // (GeomLayout::untouched) the Gaussian this lane staged in the previous batch, and that batch's first list position: a batch the tile
    // moves on from was consumed to its end -- its bits are cleared right here, from the register, with nothing to wait for; the last batch's
    // consumed prefix follows behind the loop
    uint32_t g_prev = 0xFFFFFFFFu, base_prev = 0u;
    for (uint32_t base = 0; base < n; base += FB) {
        if (__syncthreads_and(alive == 0ull)) break;
        if (untouched && g_prev != 0xFFFFFFFFu) untouched[g_prev] = 0;      // (a plain store: every writer writes the same zero)
        g_prev = 0xFFFFFFFFu; base_prev = base;
        const uint32_t i = base + t;
        bool safe = false;
        if (t < FB && i < n) {
            const uint32_t g = point_list[range.x + i];
            g_prev = g;
            const float4 b1 = rec1[(size_t)REC_STRIDE * g];
            s0[t] = rec0[(size_t)REC_STRIDE * g]; s1[t] = b1; s2[t] = rec2[(size_t)REC_STRIDE * g];
            safe = __float_as_uint(b1.z) <= zc;             // the list is in depth order: the safe entries are a prefix
        }
        uint32_t cnt = (n - base) < FB ? (n - base) : FB;
        if (zc != ZCUT_NONE) {
            const uint32_t ns = (uint32_t)__syncthreads_count(safe);
            if (ns < cnt) { cnt = ns; n_safe = base + ns; cut_here = true; }      // behind this entry Gaussians may be missing: the list ends here
        } else __syncthreads();
        if (alive != 0ull)                                  // (a saturated wave only helps staging)
#pragma unroll 1
        for (uint32_t r = 0; r < FB / 64u; r++) {
            const uint32_t slot = r * 64 + lane;
            bool touch = false;
            if (slot < cnt) {
                const float4 a = s0[slot];
                const float4 b = s1[slot];
                touch = strip_may_touch(a, b.x, b.w, sx0, sx1, sy0, sy1);
            }
            uint64_t mask = __ballot(touch);
            // (Round 5, measured and dropped: the NEXT survivor's three records requested before the current one is evaluated, a register double
            // buffer -- blend_fwd 0.275 -> 0.326 ms; the same in the backward blend's pair loop: 0.40 -> 0.449 ms.  The loop's scalar control flow then
            // depends on two mask updates per pair, and the wave does not wait for LDS here: other waves fill the slots.)
            while (mask) {
                const uint32_t j = r * 64 + (uint32_t)__builtin_ctzll(mask);
                mask &= mask - 1;
                const float4 a = s0[j];
                const float4 b = s1[j];
                const float4 c = s2[j];
                // three ds_read_b128 from one address register, up front: left alone the compiler sinks b.y, b.z and the colour
                // behind the tests as ds_read2_b32 + ds_read_b96 (8 LDS cycles where a b128 takes 4) and two more address moves
                asm volatile("" :: "v"(b.y), "v"(b.z), "v"(c.x), "v"(c.w));
                const float dx = a.x - pxa, dy = a.y - pyf;
                const float power = gs_power(a.z, a.w, b.x, dx, dy);
                // The reference's nested tests (forward.cu:368-381) as wave-uniform 64-bit masks: every test is ONE compare straight
                // into a scalar register pair, the masks are combined on the scalar unit and come back as a lane predicate for free
                // (inverse ballot).  No lane-private bool exists, so nothing has to be carried out of a divergent region (a
                // v_cndmask + v_cmp per flag) and the loop's own masks (`alive`, `mask`) stay scalar.  The arithmetic runs on all
                // lanes -- a wave instruction costs the same whatever the exec mask -- and lanes outside m_in compute garbage
                // that no mask lets through.
                const uint64_t m_in = __builtin_amdgcn_ballot_w64(power <= 0.0f) & __builtin_amdgcn_ballot_w64(power >= b.w);
                GS_COUNT(0, 1);
#if defined(GSRAST_DEBUG_COUNTERS) && !defined(GSRAST_DEBUG_TIMING)
                // (what packing two / four instances into one wave could save: iterations in which each half / quarter of the block has a lane in range)
                dbg_h[0] += (m_in & 0xFFFFFFFFull) ? 1u : 0u; dbg_h[1] += (m_in >> 32) ? 1u : 0u;
                dbg_q[0] += (m_in & 0xFFFFull) ? 1u : 0u; dbg_q[1] += ((m_in >> 16) & 0xFFFFull) ? 1u : 0u; dbg_q[2] += ((m_in >> 32) & 0xFFFFull) ? 1u : 0u; dbg_q[3] += (m_in >> 48) ? 1u : 0u;
#endif
                if (m_in == 0ull) continue;
                float alpha = b.y * gs_exp<EXPMODE, true>(power);          // b.w >= -80: the bounded exp is exact here
                alpha = alpha < 0.99f ? alpha : 0.99f;
                const float test_T = T * (1.0f - alpha);
                const uint64_t m_contrib = m_in & __builtin_amdgcn_ballot_w64(!(alpha < 1.0f / 255.0f));
                const uint64_t m_low = __builtin_amdgcn_ballot_w64(test_T < 0.0001f);
                const uint64_t m_upd = m_contrib & ~m_low, m_term = m_contrib & m_low;
                if (__builtin_amdgcn_inverse_ballot_w64(m_upd)) {
                    // forward.cu:361: C += feature * alpha * T, associated as in the source: (feature * alpha) * T
                    C0 = __builtin_fmaf(c.x * alpha, T, C0);
                    C1 = __builtin_fmaf(c.y * alpha, T, C1);
                    C2 = __builtin_fmaf(c.z * alpha, T, C2);
                    T = test_T;
                    last = base + j + 1;
                }
                // median depth (forward.cu:368-372: the Gaussian that takes T across 0.5).  T only falls, so the lanes still above
                // 0.5 are a shrinking scalar mask: once a pixel is at or below 0.5 the test costs it nothing (it was two compares
                // and a select per pair), and in an occluded scene that is most of the list.
                if (m_above & m_upd) {
                    const uint64_t m_here = m_above & m_upd;
                    const uint64_t m_cross = m_here & __builtin_amdgcn_ballot_w64(test_T < 0.5f);
                    Dm = __builtin_amdgcn_inverse_ballot_w64(m_cross) ? b.z : Dm;
                    m_above &= ~(m_here & __builtin_amdgcn_ballot_w64(!(test_T > 0.5f)));      // T == 0.5 exactly: never crosses (T > 0.5 fails from then on)
                }
#if defined(GSRAST_DEBUG_COUNTERS) && !defined(GSRAST_DEBUG_TIMING)
                if (lane == 0) { atomicAdd(&g_dbg[1], 1ull); atomicAdd(&g_dbg[2], (unsigned long long)__popcll(m_upd)); if (m_upd) atomicAdd(&g_dbg[3], 1ull); }
#endif
                if (m_term) {                                              // rare: some pixel is done
                    pxa = __builtin_amdgcn_inverse_ballot_w64(m_term) ? FAR : pxa;
                    alive &= ~m_term;
                    if (alive == 0ull) { mask = 0; r = FB / 64u; }        // wave saturated
                }
            }
        }
        if (cut_here) break;                                // uniform
    }
#if defined(GSRAST_DEBUG_COUNTERS) && !defined(GSRAST_DEBUG_TIMING)
    if (lane == 0) {
        atomicAdd(&g_dbg[10], (unsigned long long)max(dbg_h[0], dbg_h[1]));
        atomicAdd(&g_dbg[12], (unsigned long long)max(max(dbg_q[0], dbg_q[1]), max(dbg_q[2], dbg_q[3])));
    }
#endif
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
    uint32_t m = last;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { uint32_t o = __shfl_xor(m, d, 64); m = o > m ? o : m; }
    const bool all_done = __syncthreads_and(alive == 0ull) != 0;    // every pixel of the tile inside the image has saturated
    if (lane == 0) atomicMax(&s_max, m);
    __syncthreads();
    // Round 5: which Gaussians did anybody consume?  The backward walks this tile's list up to s_max and no further, so only those can
    // receive a gradient: their bits are cleared (fire-and-forget atomics of a VALU-bound kernel, from the registers the staging left the
    // ids in: a loop over the list behind the kernel's last barrier cost every workgroup a memory round trip, blend_fwd 0.246 -> 0.274 ms at
    // 3 M), every other Gaussian's output rows are zeros that late_rows_zero_kernel writes beside the blend backward and the per-Gaussian
    // backward skips -- for ANY forward, not only one that ran under a remembered cut.  (Batches in front of the last one are cleared whole, a
    // tile the completion pass blends again clears a few bits too many: harmless, a cleared bit only means "look at the record".)
    if (untouched && g_prev != 0xFFFFFFFFu && base_prev + t < s_max) untouched[g_prev] = 0;
    if (t == 0) {
        tile_max[tile] = s_max;
        // list cut: the speculation failed for this tile if it had a cut and some pixel would have looked further
        const bool again = zc != ZCUT_NONE && !all_done;
        if (tile_flags) tile_flags[tile] = again ? 1 : 0;
        if (again && cut_scalars) {      // (a RETURNING atomic, waited for: the gate below counts this workgroup out only after its verdict has arrived)
            const uint32_t before = atomicAdd(&cut_scalars[SC_UNDONE], 1u);
            asm volatile("" :: "v"(before));
        }
        if (hints) {
            const size_t hslot = (size_t)h_slot * ntiles + tile;
            hint_work(hints, ntiles)[hslot] = (uint16_t)(s_max < 65535u ? s_max : 65535u);
            // the tile's next cut depth: that of the entry 1.5 x (cut_margin_x4 / 4) as deep (+ 32) as the deepest one consumed; none for
            // a tile that did not saturate, or whose (full) list is shorter than that
            uint32_t znew = ZCUT_NONE;
            if (all_done) {
                const uint32_t p = cut_margin_x4 * s_max / 4u + 32u;
                if (p < n_safe) znew = __float_as_uint(rec1[(size_t)REC_STRIDE * point_list[range.x + p]].z);
                else if (zc != ZCUT_NONE && n_safe > 0u) {
                    // the cut list does not reach that deep: position -> depth extrapolated linearly from the list's first entry
                    const float z0 = rec1[(size_t)REC_STRIDE * point_list[range.x]].z, zcf = __uint_as_float(zc);
                    const float zn = z0 + (zcf - z0) * ((float)(p + 1u) / (float)n_safe);
                    znew = zn < 3.0e38f ? (zn > zcf ? __float_as_uint(zn) : zc) : ZCUT_NONE;
                } else if (zc != ZCUT_NONE) znew = zc;
            }
            uint32_t* const zslot = hint_zcut(hints, ntiles) + hslot;
            // Round 5: the cut REMEMBERS.  SaRO-GS renders a camera at a different timestamp every time (opacity = sigmoid * trbf(t), means
            // and scales moved by the deformation field: scene/saro_gaussian.py:791-829), so the depth at which a tile saturates varies
            // from visit to visit; a cut that follows the last visit alone is too short every other time, and one short tile costs the
            // call a completion pass.  A slot that already held THIS pose (hint_sel[1] == 1) keeps the deeper of its old cut and the new
            // one, the old one moved an eighth of the way towards the new per visit -- a running maximum over roughly the last eight
            // visits, which still follows a scene that gets more opaque for good.  A frozen scene's cut does not change.
            if (znew != ZCUT_NONE && h_found == 1u) {
                const uint32_t zold = h_zold;
                if (zold != ZCUT_NONE && zold > znew) {
                    const float fo = __uint_as_float(zold), fn = __uint_as_float(znew);
                    const float fm = fo - (fo - fn) * 0.125f;
                    if (fm < 3.0e38f && fm > fn) znew = __float_as_uint(fm);
                }
            }
            *zslot = znew;
            // the pose's key and camera are published HERE, not by preprocess_fwd's block 0 while that kernel's other blocks look the
            // table up (round 4's benign race): whoever blends tile 0 writes them (scalars[HINT_PUB ...] of this call's geometry buffer)
            if (tile == 0u && hint_sel[HINT_PUB - HINT_SEL] == 1u) {
                const uint32_t slot = hint_sel[0];
                hints->key[slot][0] = hint_sel[HINT_PUB - HINT_SEL + 1]; hints->key[slot][1] = hint_sel[HINT_PUB - HINT_SEL + 2];
#pragma unroll
                for (int q = 0; q < 6; q++) hints->cam[slot][q] = __uint_as_float(hint_sel[HINT_PUB - HINT_SEL + 3 + q]);
            }
        }
        // backward launch order: this tile's work there = the deepest list entry any of its pixels consumed (a tile that is blended
        // again enters the order then)
        if (bucket_cnt && !(again && tile_flags)) bucket_append_global(bucket_cnt + XCD_GROUPS * WORK_BUCKETS, bucket_list + (size_t)XCD_GROUPS * WORK_BUCKETS * xcd_group_tiles((uint32_t)gx, ntiles), ntiles, tile, s_max);
    }
}

// The kernel: the body above, and -- list cut, first pass -- the completion pass's GATE in the launch's last workgroup (gsrast_capi.hip,
// ChainGate): every workgroup counts itself out; the last one knows the launch's verdict (cut_scalars[SC_UNDONE]: tiles whose cut list was
// too short), copies it into the context's own word for the pass's predicated launches and, if it is "none", releases the caller's stream.
struct GateArgs { uint32_t* count /* zeroed by preprocess_fwd */; uint32_t* pred; uint32_t* done; uint32_t seq; };
template <int EXPMODE>
__global__ void __launch_bounds__(256)
blend_fwd_cull_kernel(const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                      const uint32_t* __restrict__ order, int W, int H,
                      int gx, uint32_t ntiles, const float4* __restrict__ rec0, const float4* __restrict__ rec1,
                      const float4* __restrict__ rec2, const float* __restrict__ bg,
                      float* __restrict__ out_color, float* __restrict__ out_depth, float* __restrict__ final_T,
                      uint32_t* __restrict__ n_contrib, uint32_t* __restrict__ tile_max,
                      uint32_t* __restrict__ bucket_cnt, uint16_t* __restrict__ bucket_list, int order_from_buckets,
                      float4* __restrict__ zero4, uint32_t n_zero4, HintTable* __restrict__ hints, const uint32_t* __restrict__ hint_sel,
                      const uint32_t* __restrict__ zcut_used, uint32_t* __restrict__ cut_scalars, const uint32_t* __restrict__ pred,
                      unsigned char* __restrict__ tile_flags, GateArgs gate, uint32_t cut_margin_x4, unsigned char* __restrict__ untouched)
{
    blend_fwd_cull_body<EXPMODE>(ranges, point_list, order, W, H, gx, ntiles, rec0, rec1, rec2, bg, out_color, out_depth, final_T, n_contrib, tile_max,
                                 bucket_cnt, bucket_list, order_from_buckets, zero4, n_zero4, hints, hint_sel, zcut_used, cut_scalars, pred, tile_flags, cut_margin_x4, untouched);
    // (no fence: a release fence here writes the L2 back once per workgroup -- measured: the launch 0.24 -> 0.62 ms.  None is needed: the
    // verdict travels in device-scope atomics, each workgroup's has returned before it counts itself out, and everything else the blend
    // wrote is ordered by the end of the kernel -- the wait behind it is a later command on the same stream)
    // Two levels of counters: 8160 returning atomics on ONE word serialise at the memory side (measured: the launch 0.24 -> 0.45 ms);
    // workgroup b counts into word b mod 64, the last of each residue class into the 65th.
    if (gate.count && threadIdx.x == 0) {          // (thread 0 is the one that raised SC_UNDONE for this workgroup's tile, if anybody did)
        const uint32_t cls = blockIdx.x & 63u, quota = (gridDim.x - cls + 63u) / 64u;        // workgroups b < gridDim.x with b mod 64 == cls
        if (atomicAdd(gate.count + cls, 1u) == quota - 1u && atomicAdd(gate.count + 64, 1u) == (gridDim.x < 64u ? gridDim.x : 64u) - 1u) {
            const uint32_t v = __hip_atomic_load(cut_scalars + SC_UNDONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *gate.pred = v;
            if (v == 0u) __hip_atomic_store(gate.done, gate.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // (relaxed: the waiter is a later command of the same stream)
        }
    }
}

// PPL = pixels per lane.  A tile is 256 pixels; a workgroup has NT = 256/PPL lanes (4/PPL wave64).
// Pixel k of lane t is tile-linear index k*NT + t, i.e. column t%16 for every k when NT is a
// multiple of 16, rows strided by NT/16: per-Gaussian terms that depend on the column only
// (dx, conic.x*dx*dx, conic.y*dx) are shared by the lane's PPL pixels, LDS broadcast reads, loop
// control, wave reductions and atomics are amortised over PPL pixels.  PPL=4 -> one wave per tile,
// no cross-wave synchronisation at all; PPL=1 -> the classic 4-wave workgroup (more waves in
// flight for small images).
template <int PPL> struct BlendCfg {
    static constexpr int NT = 256 / PPL;
    static constexpr int BATCH = (PPL == 4) ? 128 : NT;   // instances staged in LDS per round
    static constexpr int ROWSTEP = NT / 16;
};

template <int EXPMODE, int PPL>
__global__ void __launch_bounds__(256 / PPL)
blend_fwd_kernel(const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, int W, int H,
                 int gx, uint32_t ntiles, const float4* __restrict__ rec0, const float4* __restrict__ rec1,
                 const float4* __restrict__ rec2, const float* __restrict__ bg,
                 float* __restrict__ out_color, float* __restrict__ out_depth, float* __restrict__ final_T,
                 uint32_t* __restrict__ n_contrib, uint32_t* __restrict__ tile_max,
                 uint32_t* __restrict__ bucket_cnt /* fwd [8][64] | bwd [64], or null */, uint16_t* __restrict__ bucket_list /* fwd [8][64][Tg] | bwd [64][T] */)
{
    using Cfg = BlendCfg<PPL>;
    constexpr int NT = Cfg::NT, BATCH = Cfg::BATCH;
    __shared__ float4 s0[BATCH + 1];      // +1: the software-pipelined fetch reads one slot ahead
    __shared__ float4 s1[BATCH + 1];
    __shared__ float4 s2[BATCH + 1];
    __shared__ uint32_t s_max;

    const uint32_t tile = xcd_tile(blockIdx.x, ntiles);
    if (tile >= ntiles) return;
    const uint32_t tx = tile % (uint32_t)gx, ty = tile / (uint32_t)gx;
    const uint32_t t = threadIdx.x;
    const uint32_t px = tx * TILE_X + (t & 15u);
    const float pxf = (float)px;
    uint32_t py[PPL]; float pyf[PPL]; bool inside[PPL], done[PPL];
#pragma unroll
    for (int k = 0; k < PPL; k++) {
        py[k] = ty * TILE_Y + (t >> 4) + k * Cfg::ROWSTEP;
        pyf[k] = (float)py[k];
        inside[k] = px < (uint32_t)W && py[k] < (uint32_t)H;
        done[k] = !inside[k];
    }
    const uint2 range = ranges[tile];
    const uint32_t n = range.y - range.x;
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
    if (t == 0) s_max = 0;

    float T[PPL], C0[PPL], C1[PPL], C2[PPL], Dm[PPL];
    uint32_t last[PPL];
#pragma unroll
    for (int k = 0; k < PPL; k++) { T[k] = 1.0f; C0[k] = C1[k] = C2[k] = 0.0f; Dm[k] = 15.0f; last[k] = 0; }

    for (uint32_t base = 0; base < n; base += BATCH) {
        bool all_done = true;
#pragma unroll
        for (int k = 0; k < PPL; k++) all_done = all_done && done[k];
        if (__syncthreads_and(all_done)) break;
#pragma unroll
        for (int r = 0; r < BATCH / NT; r++) {
            const uint32_t slot = t + r * NT, i = base + slot;
            if (i < n) {
                const uint32_t g = point_list[range.x + i];
                s0[slot] = rec0[(size_t)REC_STRIDE * g]; s1[slot] = rec1[(size_t)REC_STRIDE * g]; s2[slot] = rec2[(size_t)REC_STRIDE * g];
            }
        }
        __syncthreads();
        const uint32_t cnt = (n - base) < (uint32_t)BATCH ? (n - base) : (uint32_t)BATCH;
        // Records are fetched one iteration ahead (register double buffer): the three wide LDS reads
        // of instance j+1 are in flight while instance j is evaluated, and the compiler cannot sink
        // them into the (rarely skipped) hit path.
        float4 na = s0[0], nb = s1[0], nc = s2[0];
        for (uint32_t j = 0; !all_done && j < cnt; j++) {
            const float4 a = na, b = nb, c = nc;
            na = s0[j + 1]; nb = s1[j + 1]; nc = s2[j + 1];
            const float dx = a.x - pxf;
            const float xx = (a.z * dx) * dx;      // conic.x * dx * dx   } shared by the lane's pixels
            const float xy = a.w * dx;             // conic.y * dx        }
            float power[PPL]; bool hit[PPL]; bool any_hit = false;
#pragma unroll
            for (int k = 0; k < PPL; k++) {
                const float dy = a.y - pyf[k];
                const float q = __builtin_fmaf(b.x * dy, dy, xx);
                power[k] = __builtin_fmaf(-0.5f, q, -(xy * dy));
                // b.w: conservative "alpha < 1/255" pre-test, see preprocess_fwd_kernel
                hit[k] = !done[k] && !(power[k] > 0.0f) && !(power[k] < b.w);
                any_hit = any_hit || hit[k];
            }
            if (!any_hit) continue;
            bool fin_any = false;
#pragma unroll
            for (int k = 0; k < PPL; k++) {
                if (hit[k]) {   // one exec-masked region per pixel slot, straight-line inside (selects, no branches)
                    float alpha = b.y * gs_exp<EXPMODE, true>(power[k]);
                    alpha = alpha < 0.99f ? alpha : 0.99f;
                    const bool ok = !(alpha < 1.0f / 255.0f);
                    const float test_T = T[k] * (1.0f - alpha);
                    const bool fin = ok && (test_T < 0.0001f);
                    const bool upd = ok && !fin;
                    const float Tm = upd ? T[k] : 0.0f;             // fma(x, 0, C) == C exactly; (feature * alpha) * T as in the source
                    C0[k] = __builtin_fmaf(c.x * alpha, Tm, C0[k]);
                    C1[k] = __builtin_fmaf(c.y * alpha, Tm, C1[k]);
                    C2[k] = __builtin_fmaf(c.z * alpha, Tm, C2[k]);
                    Dm[k] = (upd && T[k] > 0.5f && test_T < 0.5f) ? b.z : Dm[k];
                    T[k] = upd ? test_T : T[k];
                    last[k] = upd ? base + j + 1 : last[k];
                    done[k] = done[k] || fin;
                    fin_any = fin_any || fin;
                }
            }
            if (fin_any) {
                all_done = true;
#pragma unroll
                for (int k = 0; k < PPL; k++) all_done = all_done && done[k];
            }
        }
    }
    const size_t plane = (size_t)W * H;
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < PPL; k++) {
        if (inside[k]) {
            const size_t pid = (size_t)W * py[k] + px;
            final_T[pid] = T[k];
            n_contrib[pid] = last[k];
            out_color[pid] = __builtin_fmaf(T[k], bg0, C0[k]);
            out_color[plane + pid] = __builtin_fmaf(T[k], bg1, C1[k]);
            out_color[2 * plane + pid] = __builtin_fmaf(T[k], bg2, C2[k]);
            out_depth[pid] = Dm[k];
        }
        m = last[k] > m ? last[k] : m;
    }
    // deepest list position consumed by any pixel of the tile (bounds the backward traversal)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { uint32_t o = __shfl_xor(m, d, 64); m = o > m ? o : m; }
    __syncthreads();
    if (lane_id() == 0) atomicMax(&s_max, m);
    __syncthreads();
    if (t == 0) {
        tile_max[tile] = s_max;
        // backward launch order: this tile's work there = the deepest list entry any of its pixels consumed
        if (bucket_cnt) bucket_append_global(bucket_cnt + XCD_GROUPS * WORK_BUCKETS, bucket_list + (size_t)XCD_GROUPS * WORK_BUCKETS * xcd_group_tiles((uint32_t)gx, ntiles), ntiles, tile, s_max);
    }
}

// Backward.  Per (pixel, instance) contribution -> 9 partial derivatives.  A lane first adds up its
// own PPL pixels, the wave sums over its 64 lanes with DPP row shifts / row broadcasts (no LDS
// traffic, no shuffles), lane 63 adds the 9 totals into a per-batch LDS accumulator, and after the
// batch every lane commits ONE instance's 9 sums with hardware float atomics -- 9 atomics per
// (tile, instance) issued 64 lanes wide, instead of the reference's 9 atomics per (pixel, instance)
// pair (backward.cu:523-554).  A wave in which no pixel is touched skips everything.
template <int EXPMODE, int PPL, int ABLATE = 0>
__global__ void __launch_bounds__(256 / PPL)
blend_bwd_kernel(const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, int W, int H,
                 int gx, uint32_t ntiles, const float4* __restrict__ rec0, const float4* __restrict__ rec1,
                 const float4* __restrict__ rec2, const float* __restrict__ bg,
                 const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
                 const uint32_t* __restrict__ tile_max, const float* __restrict__ dL_dpix,
                 float* __restrict__ grec /*[P][GREC]: per-Gaussian gradient records, zero on entry*/)
{
#pragma clang fp contract(fast)
    using Cfg = BlendCfg<PPL>;
    constexpr int NT = Cfg::NT, BATCH = Cfg::BATCH;
    __shared__ float4 s0[BATCH];
    __shared__ float4 s1[BATCH];
    __shared__ float4 s2[BATCH];          // {r, g, b, -}
    __shared__ uint32_t sid[BATCH];
    __shared__ float acc[9][BATCH];

    const uint32_t tile = xcd_tile(blockIdx.x, ntiles);
    if (tile >= ntiles) return;
    const uint32_t tx = tile % (uint32_t)gx, ty = tile / (uint32_t)gx;
    const uint32_t t = threadIdx.x;
    const uint32_t px = tx * TILE_X + (t & 15u);
    const float pxf = (float)px;
    const uint2 range = ranges[tile];
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
    const uint32_t n_all = range.y - range.x;
    const uint32_t tm = tile_max[tile];
    const uint32_t n = tm < n_all ? tm : n_all;       // instances at list position >= n touch no pixel
    const size_t plane = (size_t)W * H;

    float pyf[PPL], T_final[PPL], T[PPL], last_alpha[PPL], bg_dot[PPL];
    float ac0[PPL], ac1[PPL], ac2[PPL], lc0[PPL], lc1[PPL], lc2[PPL], dp0[PPL], dp1[PPL], dp2[PPL];
    uint32_t last[PPL];
#pragma unroll
    for (int k = 0; k < PPL; k++) {
        const uint32_t py = ty * TILE_Y + (t >> 4) + k * Cfg::ROWSTEP;
        pyf[k] = (float)py;
        const bool inside = px < (uint32_t)W && py < (uint32_t)H;
        const size_t pid = (size_t)W * py + px;
        T_final[k] = inside ? final_T[pid] : 0.0f;
        T[k] = T_final[k];
        last[k] = inside ? n_contrib[pid] : 0u;
        dp0[k] = inside ? dL_dpix[pid] : 0.f; dp1[k] = inside ? dL_dpix[plane + pid] : 0.f; dp2[k] = inside ? dL_dpix[2 * plane + pid] : 0.f;
        bg_dot[k] = bg0 * dp0[k] + bg1 * dp1[k] + bg2 * dp2[k];
        ac0[k] = ac1[k] = ac2[k] = lc0[k] = lc1[k] = lc2[k] = last_alpha[k] = 0.f;
    }
    const float ddelx_dx = 0.5f * (float)W, ddely_dy = 0.5f * (float)H;
    const unsigned lane = lane_id();

    // list position `pos` (0-based from the FRONT of the tile list) is visited from n-1 down to 0
    for (uint32_t base = 0; base < n; base += BATCH) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < BATCH / NT; r++) {
            const uint32_t slot = t + r * NT, i = base + slot;
            if (i < n) {
                const uint32_t g = point_list[range.x + (n - 1 - i)];
                sid[slot] = g;
                s0[slot] = rec0[(size_t)REC_STRIDE * g]; s1[slot] = rec1[(size_t)REC_STRIDE * g]; s2[slot] = rec2[(size_t)REC_STRIDE * g];
            }
#pragma unroll
            for (int q = 0; q < 9; q++) acc[q][slot] = 0.0f;
        }
        __syncthreads();
        const uint32_t cnt = (n - base) < (uint32_t)BATCH ? (n - base) : (uint32_t)BATCH;
        for (uint32_t j = 0; j < cnt; j++) {
            const uint32_t pos = n - 1 - (base + j);
            bool reach = false;
#pragma unroll
            for (int k = 0; k < PPL; k++) reach = reach || (pos < last[k]);
            if (!__any(reach)) continue;               // nobody in this wave got this deep
            const float4 a = s0[j];
            const float4 b = s1[j];
            const float4 c = s2[j];
            const float dx = a.x - pxf;
            const float xx = (a.z * dx) * dx;
            const float xy = a.w * dx;
            float power[PPL], dyv[PPL]; bool hit[PPL]; bool any_hit = false;
#pragma unroll
            for (int k = 0; k < PPL; k++) {
                const float dy = a.y - pyf[k];
                dyv[k] = dy;
                const float q = __builtin_fmaf(b.x * dy, dy, xx);
                power[k] = __builtin_fmaf(-0.5f, q, -(xy * dy));
                hit[k] = pos < last[k] && !(power[k] > 0.0f) && !(power[k] < b.w);
                any_hit = any_hit || hit[k];
            }
            if (!__any(any_hit)) continue;

            float g_mx = 0.f, g_my = 0.f, g_ca = 0.f, g_cb = 0.f, g_cc = 0.f, g_op = 0.f, g_r = 0.f, g_g = 0.f, g_b = 0.f;
            bool contributed = false;
#pragma unroll
            for (int k = 0; k < PPL; k++) {
                if (!hit[k]) continue;
                const float G = gs_exp<EXPMODE, true>(power[k]);
                float alpha = b.y * G;
                alpha = alpha < 0.99f ? alpha : 0.99f;
                if (alpha < 1.0f / 255.0f) continue;
                contributed = true;
                const float dy = dyv[k];
                const float rcp1ma = __builtin_amdgcn_rcpf(1.0f - alpha);
                T[k] = T[k] * rcp1ma;
                const float dch = alpha * T[k];
                const float c0 = c.x, c1 = c.y, c2 = c.z;
                ac0[k] = last_alpha[k] * lc0[k] + (1.f - last_alpha[k]) * ac0[k]; lc0[k] = c0;
                ac1[k] = last_alpha[k] * lc1[k] + (1.f - last_alpha[k]) * ac1[k]; lc1[k] = c1;
                ac2[k] = last_alpha[k] * lc2[k] + (1.f - last_alpha[k]) * ac2[k]; lc2[k] = c2;
                float dL_dalpha = (c0 - ac0[k]) * dp0[k] + (c1 - ac1[k]) * dp1[k] + (c2 - ac2[k]) * dp2[k];
                g_r += dch * dp0[k]; g_g += dch * dp1[k]; g_b += dch * dp2[k];
                dL_dalpha *= T[k];
                last_alpha[k] = alpha;
                dL_dalpha += (-T_final[k] * rcp1ma) * bg_dot[k];
                const float dL_dG = b.y * dL_dalpha;
                const float gdx = G * dx, gdy = G * dy;
                const float dG_ddelx = -gdx * a.z - gdy * a.w;
                const float dG_ddely = -gdy * b.x - gdx * a.w;
                g_mx += dL_dG * dG_ddelx;
                g_my += dL_dG * dG_ddely;
                g_ca += gdx * dx * dL_dG;
                g_cb += gdx * dy * dL_dG;
                g_cc += gdy * dy * dL_dG;
                g_op += G * dL_dalpha;
            }
            if (!__any(contributed)) continue;
            if constexpr (ABLATE < 2) {
                g_mx = wave_sum_to_lane63(g_mx); g_my = wave_sum_to_lane63(g_my);
                g_ca = wave_sum_to_lane63(g_ca); g_cb = wave_sum_to_lane63(g_cb); g_cc = wave_sum_to_lane63(g_cc);
                g_op = wave_sum_to_lane63(g_op);
                g_r = wave_sum_to_lane63(g_r); g_g = wave_sum_to_lane63(g_g); g_b = wave_sum_to_lane63(g_b);
            }
            if constexpr (ABLATE >= 1) {
                asm volatile("" :: "v"(g_mx), "v"(g_my), "v"(g_ca), "v"(g_cb), "v"(g_cc), "v"(g_op), "v"(g_r), "v"(g_g), "v"(g_b));
            } else if (lane == 63) {
                if constexpr (NT == 64) {   // single wave: plain read-modify-write is race free
                    acc[0][j] += g_mx * ddelx_dx; acc[1][j] += g_my * ddely_dy;
                    acc[2][j] += -0.5f * g_ca; acc[3][j] += -0.5f * g_cb; acc[4][j] += -0.5f * g_cc;
                    acc[5][j] += g_op; acc[6][j] += g_r; acc[7][j] += g_g; acc[8][j] += g_b;
                } else {
                    atomicAdd(&acc[0][j], g_mx * ddelx_dx); atomicAdd(&acc[1][j], g_my * ddely_dy);
                    atomicAdd(&acc[2][j], -0.5f * g_ca); atomicAdd(&acc[3][j], -0.5f * g_cb); atomicAdd(&acc[4][j], -0.5f * g_cc);
                    atomicAdd(&acc[5][j], g_op); atomicAdd(&acc[6][j], g_r); atomicAdd(&acc[7][j], g_g); atomicAdd(&acc[8][j], g_b);
                }
            }
        }
        __syncthreads();
        // commit: one lane per staged instance, 9 float atomics each, all lanes wide
#pragma unroll
        for (int r = 0; r < BATCH / NT; r++) {
            const uint32_t slot = t + r * NT;
            if (slot < cnt) {
                const size_t gid = sid[slot];
                const float v0 = acc[0][slot], v1 = acc[1][slot], v2 = acc[2][slot], v3 = acc[3][slot], v4 = acc[4][slot];
                const float v5 = acc[5][slot], v6 = acc[6][slot], v7 = acc[7][slot], v8 = acc[8][slot];
                const bool any = (v0 != 0.f) | (v1 != 0.f) | (v2 != 0.f) | (v3 != 0.f) | (v4 != 0.f) | (v5 != 0.f) | (v6 != 0.f) | (v7 != 0.f) | (v8 != 0.f);
                if (any) {
                    float* rec = grec + gid * GREC;
                    atomicAdd(rec + 0, v0); atomicAdd(rec + 1, v1); atomicAdd(rec + 2, v2); atomicAdd(rec + 3, v3); atomicAdd(rec + 4, v4);
                    atomicAdd(rec + 5, v5); atomicAdd(rec + 6, v6); atomicAdd(rec + 7, v7); atomicAdd(rec + 8, v8);
                }
            }
        }
    }
}

// Backward blend with wave-level culling.  Same arithmetic and the same reduction / commit scheme as
// blend_bwd_kernel; the difference is WHICH (strip, instance) pairs are evaluated at all: per staged
// round of 64 instances every lane tests one instance against each of the wave's PPL pixel strips
// (exact box minimum of the quadratic, strip_may_touch) and against the deepest list position that
// strip still needs; the ballots are 64-bit SGPR masks, the blend loop walks their union with
// s_ff1 and enters pixel slot k only if bit j of mask k is set -- a scalar branch, no VALU work for
// untouched strips.
// Measured alternatives for the last steps of the cross-lane sums (all slower than the in-register butterfly below, 0.43 ms):
// ds_add_f32 LDS atomics after three butterfly steps, 8 lanes per address: 0.91 ms; after four steps, 4 lanes per address:
// 0.55 ms (same-address LDS atomics serialise); row totals combined through a 36-float LDS scratch (one ds_write_b32 + one
// ds_read_b128 instead of the two ds_bpermute exchanges and the ninth value's row_bcast steps): 0.44 ms (one more LDS round trip
// in the dependency chain costs more than the exchanges it removes).
template <int EXPMODE, int PPL>
__global__ void __launch_bounds__(256 / PPL)
blend_bwd_cull_kernel(const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                      const uint32_t* __restrict__ order, int W, int H,
                      int gx, uint32_t ntiles, const float4* __restrict__ rec0, const float4* __restrict__ rec1,
                      const float4* __restrict__ rec2, const float* __restrict__ bg,
                      const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
                      const uint32_t* __restrict__ tile_max, const float* __restrict__ dL_dpix,
                      float* __restrict__ grec /*[P][GREC]: per-Gaussian gradient records, zero on entry*/,
                      const uint32_t* __restrict__ bucket_cnt /* fwd [8][64] | bwd [64]: launch order from the bwd lists, or null */,
                      const uint16_t* __restrict__ bucket_list)
{
#pragma clang fp contract(fast)
    using Cfg = BlendCfg<PPL>;
    // 64 instances staged per batch: 3.5 % faster than 128 (less over-fetch past the deepest consumed entry)
    constexpr int NT = Cfg::NT, BATCH = 64, NW = NT / 64;
    __shared__ float4 s0[BATCH];
    __shared__ float4 s1[BATCH];
    __shared__ float4 s2[BATCH];          // {r, g, b, -}
    __shared__ uint32_t sid[BATCH];
    // one accumulator slice per wave (plain LDS read-add-write, no LDS atomics); a staged instance's nine sums are adjacent, like
    // the record they are committed to
    constexpr int AS = 12;
    constexpr int NS = NW;                          // accumulator slices
    __shared__ __attribute__((aligned(16))) float acc[NS][BATCH][AS];

    __shared__ uint32_t s_tile;
    if (blockIdx.x >= ntiles) return;
    // heaviest tiles first (one global sequence: see tile_from_buckets_global)
    const uint32_t tile = bucket_cnt ? tile_from_buckets_global(bucket_cnt + XCD_GROUPS * WORK_BUCKETS, bucket_list + (size_t)XCD_GROUPS * WORK_BUCKETS * xcd_group_tiles((uint32_t)gx, ntiles), ntiles, blockIdx.x, &s_tile)
                                     : (order ? order[blockIdx.x] : blockIdx.x);
    const uint32_t tx = tile % (uint32_t)gx, ty = tile / (uint32_t)gx;
    const uint32_t t = threadIdx.x;
    const unsigned lane = lane_id(), wave = t >> 6;
    // a wave owns PPL strips of 16 x 4 pixels, ROWSTEP rows apart (BLK: one 8 x 8 block, PPL == 1 only)
    constexpr bool BLK = false;     // 8 x 8 blocks (as in the forward) measured 3 % SLOWER here than 16 x 4 strips
    const uint32_t bx = BLK ? (wave & 1u) * 8u : 0u, by = BLK ? (wave >> 1) * 8u : wave * 4u;
    const uint32_t px = tx * TILE_X + (BLK ? bx + (lane & 7u) : (t & 15u));
    const float pxf = (float)px;
    constexpr float SW = BLK ? 7.0f : 15.0f, SH = BLK ? 7.0f : 3.0f;      // extent of a strip / block in pixel centres
    const uint2 range = ranges[tile];
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
    const uint32_t n_all = range.y - range.x;
    const uint32_t tm = tile_max[tile];
    const uint32_t n = tm < n_all ? tm : n_all;       // instances at list position >= n touch no pixel
    const size_t plane = (size_t)W * H;
    const float sx0 = (float)(tx * TILE_X + bx), sx1 = sx0 + SW;

    float pyf[PPL], tfbg[PPL], T[PPL], last_alpha[PPL], sy0[PPL];
    float ac0[PPL], ac1[PPL], ac2[PPL], lc0[PPL], lc1[PPL], lc2[PPL], dp0[PPL], dp1[PPL], dp2[PPL];
    uint32_t last[PPL], strip_last[PPL];
#pragma unroll
    for (int k = 0; k < PPL; k++) {
        const uint32_t py = ty * TILE_Y + (BLK ? by + (lane >> 3) : (t >> 4) + k * Cfg::ROWSTEP);
        pyf[k] = (float)py;
        sy0[k] = (float)(ty * TILE_Y + by + k * Cfg::ROWSTEP);
        const bool inside = px < (uint32_t)W && py < (uint32_t)H;
        const size_t pid = (size_t)W * py + px;
        const float Tf = inside ? final_T[pid] : 0.0f;
        T[k] = Tf;
        last[k] = inside ? n_contrib[pid] : 0u;
        dp0[k] = inside ? dL_dpix[pid] : 0.f; dp1[k] = inside ? dL_dpix[plane + pid] : 0.f; dp2[k] = inside ? dL_dpix[2 * plane + pid] : 0.f;
        tfbg[k] = -Tf * (bg0 * dp0[k] + bg1 * dp1[k] + bg2 * dp2[k]);
        ac0[k] = ac1[k] = ac2[k] = lc0[k] = lc1[k] = lc2[k] = last_alpha[k] = 0.f;
        uint32_t m = last[k];                            // deepest position the strip needs (wave-uniform)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { uint32_t o = __shfl_xor(m, d, 64); m = o > m ? o : m; }
        strip_last[k] = __builtin_amdgcn_readfirstlane(m);
    }
    const float ddelx_dx = 0.5f * (float)W, ddely_dy = 0.5f * (float)H;
    // factor applied by lane l (< 8) when it commits value l: {mean.x, mean.y, conic a, b, c, opacity, r, g}
    const unsigned kind = lane;
    const float commit_scale = kind == 0 ? -ddelx_dx : kind == 1 ? -ddely_dy : (kind >= 2 && kind <= 4) ? -0.5f : 1.0f;

    for (uint32_t base = 0; base < n; base += BATCH) {
        __syncthreads();
        for (uint32_t slot = t; slot < (uint32_t)BATCH; slot += NT) {
            const uint32_t i = base + slot;
            if (i < n) {
                const uint32_t g = point_list[range.x + (n - 1 - i)];
                sid[slot] = g;
                s0[slot] = rec0[(size_t)REC_STRIDE * g]; s1[slot] = rec1[(size_t)REC_STRIDE * g]; s2[slot] = rec2[(size_t)REC_STRIDE * g];
            }
        }
        for (uint32_t q4 = t; q4 < (uint32_t)(NS * BATCH * AS / 4); q4 += NT) reinterpret_cast<float4*>(&acc[0][0][0])[q4] = make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();
        const uint32_t cnt = (n - base) < (uint32_t)BATCH ? (n - base) : (uint32_t)BATCH;
#pragma unroll 1
        for (uint32_t r = 0; r < (uint32_t)BATCH / 64u; r++) {
            if (r * 64u >= cnt) break;
            const uint32_t slot = r * 64u + lane;
            const uint32_t spos = n - 1 - (base + slot);      // list position of this lane's instance
            uint64_t mk[PPL];
            uint64_t uni = 0;
            {
                const bool valid = slot < cnt;
                const float4 a = valid ? s0[slot] : make_float4(0.f, 0.f, 1.f, 0.f);
                const float czv = valid ? s1[slot].x : 1.f;
                const float thr = valid ? s1[slot].w : 1.f;
#pragma unroll
                for (int k = 0; k < PPL; k++) {
                    const bool touch = valid && spos < strip_last[k] &&
                                       strip_may_touch(a, czv, thr, sx0, sx1, sy0[k], sy0[k] + SH);
                    mk[k] = __ballot(touch);
                    uni |= mk[k];
                }
            }
            while (uni) {
                const uint32_t jb = (uint32_t)__builtin_ctzll(uni);
                uni &= uni - 1;
                const uint32_t j = r * 64u + jb;
                const uint32_t pos = n - 1 - (base + j);
                const float4 a = s0[j];
                const float4 b = s1[j];
                float4 c = s2[j];
                // keep the colour's LDS read up here with the others: sunk into the contributing branch (where it is
                // first used) its latency is exposed on every iteration (measured: +8 % kernel time)
                asm volatile("" : "+v"(c.x));
                const float dx = a.x - pxf;
                const float xx = (a.z * dx) * dx;
                const float xy = a.w * dx;
                // Per pixel slot, the divergent part produces only three numbers (zero when the pair does not
                // contribute): dch = alpha * T, dLa = dL/dalpha, Gk = G.  The nine per-instance partials are
                // products of those and are formed afterwards by ALL lanes, so no lane needs nine zero-initialised
                // registers per branch level and the exec mask is manipulated once per level instead of per output.
                float dch[PPL], dLa[PPL], Gk[PPL];
                bool contributed = false;
#pragma unroll
                for (int k = 0; k < PPL; k++) {
                    dch[k] = 0.f; dLa[k] = 0.f; Gk[k] = 0.f;
                    if (!((mk[k] >> jb) & 1ull)) continue;                 // scalar: strip k untouched
                    const float dy = a.y - pyf[k];
                    const float q = __builtin_fmaf(b.x * dy, dy, xx);
                    const float power = __builtin_fmaf(-0.5f, q, -(xy * dy));
                    if (pos < last[k] && power <= 0.0f && power >= b.w) {
                        const float G = gs_exp<EXPMODE, true>(power);
                        float alpha = b.y * G;
                        alpha = alpha < 0.99f ? alpha : 0.99f;
                        if (!(alpha < 1.0f / 255.0f)) {
                            contributed = true;
                            const float rcp1ma = __builtin_amdgcn_rcpf(1.0f - alpha);
                            T[k] = T[k] * rcp1ma;
                            // (folding the current pair into ac right after use -- ac += alpha (c - ac), no last_alpha /
                            // last_color state -- is 7 VALU shorter but measured 4 % SLOWER: it makes ac wait for this
                            // iteration's exp; here everything ac needs is known when the iteration starts)
                            const float c0 = c.x, c1 = c.y, c2 = c.z;
                            ac0[k] = last_alpha[k] * lc0[k] + (1.f - last_alpha[k]) * ac0[k]; lc0[k] = c0;
                            ac1[k] = last_alpha[k] * lc1[k] + (1.f - last_alpha[k]) * ac1[k]; lc1[k] = c1;
                            ac2[k] = last_alpha[k] * lc2[k] + (1.f - last_alpha[k]) * ac2[k]; lc2[k] = c2;
                            float dL_dalpha = (c0 - ac0[k]) * dp0[k] + (c1 - ac1[k]) * dp1[k] + (c2 - ac2[k]) * dp2[k];
                            last_alpha[k] = alpha;
                            dL_dalpha *= T[k];
                            dL_dalpha += tfbg[k] * rcp1ma;
                            dch[k] = alpha * T[k]; dLa[k] = dL_dalpha; Gk[k] = G;
                        }
                    }
                }
                GS_COUNT(4, 1); GS_COUNT(7, __popcll(__ballot(pos < last[0])));
                if (!__any(contributed)) continue;
                GS_COUNT(5, 1); GS_COUNT(6, __popcll(__ballot(contributed)));
                float g_mx = 0.f, g_my = 0.f, g_ca = 0.f, g_cb = 0.f, g_cc = 0.f, g_op = 0.f, g_r = 0.f, g_g = 0.f, g_b = 0.f;
#pragma unroll
                for (int k = 0; k < PPL; k++) {
                    if (PPL > 1 && !((mk[k] >> jb) & 1ull)) continue;      // scalar
                    const float dy = a.y - pyf[k];
                    const float dL_dG = b.y * dLa[k];
                    const float gx_ = Gk[k] * dx * dL_dG, gy_ = Gk[k] * dy * dL_dG;      // dL/dpower * d(-power)/d(...)
                    const float t_r = dch[k] * dp0[k], t_g = dch[k] * dp1[k], t_b = dch[k] * dp2[k];
                    const float t_mx = gx_ * a.z + gy_ * a.w;                            // sign folded into commit_scale
                    const float t_my = gy_ * b.x + gx_ * a.w;
                    const float t_ca = gx_ * dx, t_cb = gx_ * dy, t_cc = gy_ * dy, t_op = Gk[k] * dLa[k];
                    if (PPL == 1) {     // plain assignment: "0 + x" is not foldable under IEEE signed-zero rules
                        g_r = t_r; g_g = t_g; g_b = t_b; g_mx = t_mx; g_my = t_my; g_ca = t_ca; g_cb = t_cb; g_cc = t_cc; g_op = t_op;
                    } else {
                        g_r += t_r; g_g += t_g; g_b += t_b; g_mx += t_mx; g_my += t_my; g_ca += t_ca; g_cb += t_cb; g_cc += t_cc; g_op += t_op;
                    }
                }
                {
                    // eight of the nine sums through the transposing reduction (every lane l ends with the
                    // total of value l & 7), the ninth through the DPP chain to lane 63, read back as a scalar;
                    // lanes 0..8 then commit all nine into this wave's accumulator slice with ONE LDS read-add-write.
                    const float v8[8] = { g_mx, g_my, g_ca, g_cb, g_cc, g_op, g_r, g_g };
                    const float tot = wave_sum8_transposed(v8, lane);
                    const float tb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_sum_to_lane63(g_b)), 63));
                    if (lane < 9u) acc[wave][j][lane] += (lane < 8u ? tot : tb) * commit_scale;
                }
            }
        }
        __syncthreads();
        // commit: 16 adjacent lanes per staged instance, lane q < 9 adds sum q to float q of the Gaussian's 64-byte record -- the nine
        // atomics of a (tile, Gaussian) pair leave in ONE wave instruction and land in ONE cache line
        for (uint32_t e = t; e < cnt * 16u; e += NT) {
            const uint32_t slot = e >> 4, q = e & 15u;
            if (q < 9u) {
                float v = acc[0][slot][q];
#pragma unroll
                for (int w = 1; w < NS; w++) v += acc[w][slot][q];
                if (v != 0.f) atomicAdd(grec + (size_t)sid[slot] * GREC + q, v);
            }
        }
    }
}

// Backward blend, one pixel per lane, with the cross-lane sums TRANSPOSED out of the per-pair loop.
// In blend_bwd_cull_kernel every surviving (wave, instance) pair pays for nine 64-lane reductions (21 DPP steps, 7 selects, two
// ds_bpermute exchanges: measured 33 % of the kernel, the nine products another 15 %).  Here the divergent part of a pair only
// leaves two numbers per pixel -- u = G dL/dalpha and dch = alpha T -- in an LDS strip [instance][pixel]; after a group of
// eight staged instances the wave turns round: lane (k, jj) = (lane & 7, lane >> 3) walks pixels k, k + 8, ... of instance jj and
// accumulates that instance's nine sums in registers (14 full-rate FMA-class instructions per 8 pixels x 8 instances, no
// cross-lane traffic), and ONE three-step butterfly over the 8-lane groups finishes all eight instances at once.  Per staged
// instance and wave that is ~52 issue cycles instead of ~160 per surviving pair.  The geometric sums are taken about the pixel
// itself (dx, dy recomputed from the instance's mean: identical operands to the per-pair formulation).
// (Measured and dropped: compacting the surviving pairs into the eight rows -- a pair takes the next free row, its staged index in
// a packed scalar, the transposed phase runs when eight rows are full -- instead of fixed groups of eight consecutive staged
// instances: 332 -> 342 us at 1 M, 464 -> 455 at 3 M.  The groups are dense enough; the zeros are pixels inside a row.)
// (Also measured and dropped: v_exp_f32 for the backward's exp with the exact sequence only when some lane's alpha is within 1e-7 of
// 1/255 -- the decision stays the forward's, ~20 issue cycles per pair fewer on paper: 332 -> 332 us.  The kernel is not short of
// issue slots for that chain; the exp hides under the LDS round trips of the pair's records.)
// (Round 4, measured and dropped -- the kernel is bound by VALU throughput, not by waiting: DOUBLE-BUFFERED staging, the records of batch
// b + 1 written and the sums of batch b - 1 committed while batch b is processed, ONE workgroup barrier per batch instead of three: 0.46 ->
// 0.46 ms at 3 M, 0.326 -> 0.358 (batches of 64, 35 KB of LDS) / 0.338 (batches of 32) at 1 M, shell 0.513 -> 0.564 / 0.530; SIX workgroups
// per compute unit (27.1 KB of LDS, 76 VGPRs): 0.457 -> 0.478 at 3 M, 0.327 -> 0.340 at 1 M; s_setprio 3 for tiles of more than 256 / 768
// consumed entries: nothing; and the launch is NOT the heavy tiles' serial chain: consuming at most 256 entries per tile (16 % of the
// entries gone, wrong gradients) takes 0.455 -> 0.403 ms, in proportion.)
#ifdef GSRAST_BWD_WAVES      // (A/B: cap the registers for that many waves per SIMD)
#define GSRAST_BWD_OCC __attribute__((amdgpu_waves_per_eu(GSRAST_BWD_WAVES, GSRAST_BWD_WAVES)))
#else
#define GSRAST_BWD_OCC
#endif
template <int EXPMODE>
__global__ void __launch_bounds__(256) GSRAST_BWD_OCC
blend_bwd_cull_t_kernel(const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                        const uint32_t* __restrict__ order, int W, int H,
                        int gx, uint32_t ntiles, const float4* __restrict__ rec0, const float4* __restrict__ rec1,
                        const float4* __restrict__ rec2, const float* __restrict__ bg,
                        const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
                        const uint32_t* __restrict__ tile_max, const float* __restrict__ dL_dpix,
                        float* __restrict__ grec /*[P][GREC]: per-Gaussian gradient records, zero on entry*/,
                        const uint32_t* __restrict__ bucket_cnt, const uint16_t* __restrict__ bucket_list,
                        uint32_t* __restrict__ fork_word /* or null: "this kernel has started" for a stream that waits for it (gsrast_capi.hip: WORD FORKS) */, uint32_t fork_seq)
{
#pragma clang fp contract(fast)
    if (fork_word && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(fork_word, fork_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    constexpr int NT = 256, BATCH = 64, NW = 4;
    constexpr int GB = 8;            // instances per group
    // (Round 5, measured and dropped: the pair's derivative FRONT TO BACK --  dC/dalpha_i . g = T_i (c_i . g) - R_i / (1 - alpha_i), R_i = what
    // is left of colour_out . g behind contributor i: ONE scalar recurrence and the forward's own T update, 11 instructions where the
    // reference's back-to-front form (T / (1 - alpha), the three-channel accum_rec: backward.cu:503-516) has 19.  0.398 -> 0.390 ms at 3 M, and
    // the forward 18 us slower for the copy of its output colour the backward then needs; R is a difference from the pixel's TOTAL, whose
    // absolute error (the forward's accumulated rounding, ~ n eps |colour . g|) does not shrink with T_i as the back-to-front form's does:
    // tests/test_gpu_parity.py::test_golden_fixture failed the 1e-5 bar at 1.8e-5.)
    constexpr int PS = 72;           // float2 slots per instance row: 64 pixels + 8 of padding (row stride = 16 banks mod 64: the
                                     // transposed phase's ds_read_b64 of lanes (k, jj), jj = 0..3 within a 32-lane group, are conflict-free)
    constexpr int AS = 12;
    __shared__ float4 srec[3][BATCH];     // the staged instances' records: rec0 | rec1 | rec2 {r, g, b, -}
    float4* const s0 = srec[0];
    float4* const s1 = srec[1];
    float4* const s2 = srec[2];
    __shared__ uint32_t sid[BATCH];
    __shared__ __attribute__((aligned(16))) float acc[BATCH][AS];       // the batch's sums, shared by the four waves (LDS float adds to distinct addresses)
    // {u, dch} of the current group, [instance][pixel of the wave's strip]
    __shared__ float2 pbuf[NW][GB][PS];
    __shared__ uint32_t s_tile;
    if (blockIdx.x >= ntiles) return;
    const uint32_t tile = bucket_cnt ? tile_from_buckets_global(bucket_cnt + XCD_GROUPS * WORK_BUCKETS, bucket_list + (size_t)XCD_GROUPS * WORK_BUCKETS * xcd_group_tiles((uint32_t)gx, ntiles), ntiles, blockIdx.x, &s_tile)
                                     : (order ? order[blockIdx.x] : blockIdx.x);
    const uint32_t tx = tile % (uint32_t)gx, ty = tile / (uint32_t)gx;
    const uint32_t t = threadIdx.x;
#ifdef GSRAST_DEBUG_TIMING
    const long long dbg_t0 = wall_clock64();
#endif
    const unsigned lane = lane_id(), wave = t >> 6;
#ifndef GSRAST_BWD_BLOCK8
#define GSRAST_BWD_BLOCK8 1
#endif
    // the wave's pixels: an 8 x 8 block (lane = row * 8 + column), as in the forward -- the same 64 pixels as a 16 x 4 strip with a
    // shorter outline, so fewer (wave, instance) pairs survive the culling test; in the transposed phase lane (k, jj) then walks
    // column k of the block's eight rows (ONE dx per lane).  0: the 16 x 4 strip of round 2.  Alternating runs, views/s, strip ->
    // block: 3 M 745 / 775 -> 766 / 777, 1 M 1231 -> 1244, 1 M shell 928 -> 957, 0.3 M 1609 -> 1628.  (Round 1 had rejected 8 x 8
    // blocks for its backward, whose per-pair cross-lane reductions favoured the strip's row layout.)
    constexpr bool B8 = GSRAST_BWD_BLOCK8 != 0;
    const uint32_t px = B8 ? tx * TILE_X + (wave & 1u) * 8u + (lane & 7u) : tx * TILE_X + (t & 15u);
    const uint32_t py = B8 ? ty * TILE_Y + (wave >> 1) * 8u + (lane >> 3) : ty * TILE_Y + (t >> 4);
    const float pxf = (float)px, pyf = (float)py;
    const uint2 range = ranges[tile];
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
    const uint32_t n_all = range.y - range.x;
    const uint32_t tm = tile_max[tile];
    const uint32_t n = tm < n_all ? tm : n_all;       // instances at list position >= n touch no pixel
    const size_t plane = (size_t)W * H;
    const float sx0 = B8 ? (float)(tx * TILE_X + (wave & 1u) * 8u) : (float)(tx * TILE_X), sx1 = sx0 + (B8 ? 7.0f : 15.0f);
    const float sy0 = B8 ? (float)(ty * TILE_Y + (wave >> 1) * 8u) : (float)(ty * TILE_Y + wave * 4u);
    const float sy1 = sy0 + (B8 ? 7.0f : 3.0f);

    const bool inside = px < (uint32_t)W && py < (uint32_t)H;
    const size_t pid = (size_t)W * py + px;
    const float Tf = inside ? final_T[pid] : 0.0f;
    float T = Tf;
    const uint32_t last = inside ? n_contrib[pid] : 0u;
    const float dp0 = inside ? dL_dpix[pid] : 0.f, dp1 = inside ? dL_dpix[plane + pid] : 0.f, dp2 = inside ? dL_dpix[2 * plane + pid] : 0.f;
    const float tfbg = -Tf * (bg0 * dp0 + bg1 * dp1 + bg2 * dp2);
    float ac0 = 0.f, ac1 = 0.f, ac2 = 0.f;
    uint32_t strip_last;
    {
        uint32_t m = last;                            // deepest position the strip needs (wave-uniform)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { uint32_t o = __shfl_xor(m, d, 64); m = o > m ? o : m; }
        strip_last = __builtin_amdgcn_readfirstlane(m);
    }
    // dL/dpixel of the eight pixels lane (k, jj) walks in the transposed phase (column k of the block's rows): the same 24 numbers for
    // every group of the launch -- kept in registers (exchanged once through the strip buffer) instead of eight 16-byte LDS reads per phase
    // (round 4: two thirds of the phase's LDS traffic and 4 KB of LDS gone; 96 VGPRs = still five waves per SIMD; blend_bwd 0.452 -> 0.442 ms at
    // 3 M, 0.325 -> 0.322 at 1 M, shell 0.513 -> 0.502)
    float dpr[8][3];
    {
        float4* ex = reinterpret_cast<float4*>(&pbuf[wave][0][0]);
        ex[lane] = make_float4(dp0, dp1, dp2, 0.0f);
        __builtin_amdgcn_wave_barrier();
        const unsigned kk = lane & 7u;
#pragma unroll
        for (int i = 0; i < 8; i++) { const float4 v = ex[kk + 8 * i]; dpr[i][0] = v.x; dpr[i][1] = v.y; dpr[i][2] = v.z; }
        __builtin_amdgcn_wave_barrier();
    }
    // transposed phase: this lane's role
    const unsigned k = lane & 7u, jj = lane >> 3;
    const float commit_scale = k == 0 ? -0.5f * (float)W : k == 1 ? -0.5f * (float)H : (k >= 2 && k <= 4) ? -0.5f : 1.0f;
    const float pxk0 = sx0 + (float)k, pxk1 = pxk0 + 8.0f;       // columns of pixels k + 16 r and k + 8 + 16 r of the strip

    // Staging is software-pipelined: wave w < 3 fetches record w of the 64 instances of the NEXT batch (and the ids of the batch
    // after that) into registers right after the barrier that opens a batch, so the two dependent global latencies
    // (point_list -> record) pass while the batch is processed instead of with the whole workgroup waiting between two barriers.
    const float4* const recw = wave == 0u ? rec0 : (wave == 1u ? rec1 : rec2);
    const auto staged_id = [&](uint32_t b) -> uint32_t {
        const uint32_t i = b + lane;
        return (wave < 3u && i < n) ? point_list[range.x + (n - 1 - i)] : 0xFFFFFFFFu;
    };
    uint32_t id_cur = staged_id(0), id_next = staged_id(BATCH);
    float4 pre = id_cur != 0xFFFFFFFFu ? recw[(size_t)REC_STRIDE * id_cur] : make_float4(0.f, 0.f, 0.f, 0.f);

    for (uint32_t base = 0; base < n; base += BATCH) {
        __syncthreads();
        if (wave < 3u) {
            if (id_cur != 0xFFFFFFFFu) { srec[wave][lane] = pre; if (wave == 0u) sid[lane] = id_cur; }
        } else {
#pragma unroll
            for (int r = 0; r < BATCH * AS / 4 / 64; r++)
                reinterpret_cast<float4*>(&acc[0][0])[r * 64 + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
        id_cur = id_next;
        if (id_cur != 0xFFFFFFFFu) pre = recw[(size_t)REC_STRIDE * id_cur];
        id_next = staged_id(base + 2 * BATCH);
        const uint32_t cnt = (n - base) < (uint32_t)BATCH ? (n - base) : (uint32_t)BATCH;
        uint64_t mk;
        {
            const uint32_t spos = n - 1 - (base + lane);      // list position of this lane's instance
            const bool valid = lane < cnt;
            const float4 a = valid ? s0[lane] : make_float4(0.f, 0.f, 1.f, 0.f);
            const float czv = valid ? s1[lane].x : 1.f;
            const float thr = valid ? s1[lane].w : 1.f;
            mk = __ballot(valid && spos < strip_last && strip_may_touch(a, czv, thr, sx0, sx1, sy0, sy1));
        }
#pragma unroll 1
        for (uint32_t g = 0; g < (uint32_t)(BATCH / GB); g++) {
            uint32_t walk = (uint32_t)(mk >> (GB * g)) & ((1u << GB) - 1u);
            if (!walk) continue;                                            // scalar: no instance of the group can touch the strip
            uint32_t alive = 0;
            while (walk) {
                const uint32_t jb = (uint32_t)__builtin_ctz(walk);
                walk &= walk - 1;
                const uint32_t j = g * GB + jb;
                const uint32_t pos = n - 1 - (base + j);
                const float4 a = s0[j];
                const float4 b = s1[j];
                const float4 c = s2[j];
                asm volatile("" :: "v"(b.z), "v"(c.x), "v"(c.w));       // three ds_read_b128 up front (see blend_fwd_cull_kernel)
                const float dx = a.x - pxf, dy = a.y - pyf;
                const float q = __builtin_fmaf(b.x * dy, dy, (a.z * dx) * dx);
                const float power = __builtin_fmaf(-0.5f, q, -((a.w * dx) * dy));
                // The tests as wave-uniform masks (see blend_fwd_cull_kernel): one compare each, straight into a scalar register pair,
                // combined on the scalar unit; exp and alpha are evaluated on all lanes (same cost) and only masks decide.
                const uint64_t m_in = __builtin_amdgcn_ballot_w64(pos < last) & __builtin_amdgcn_ballot_w64(power <= 0.0f) &
                                      __builtin_amdgcn_ballot_w64(power >= b.w);
                GS_COUNT(4, 1); GS_COUNT(7, __popcll(__ballot(pos < last)));
                if (m_in == 0ull) continue;
                const float G = gs_exp<EXPMODE, true>(power);
                float alpha = b.y * G;
                alpha = alpha < 0.99f ? alpha : 0.99f;
                const uint64_t m_ok = m_in & __builtin_amdgcn_ballot_w64(!(alpha < 1.0f / 255.0f));
                if (m_ok == 0ull) continue;
                GS_COUNT(5, 1); GS_COUNT(6, __popcll(m_ok));
                float u = 0.f, dch = 0.f;
                if (__builtin_amdgcn_inverse_ballot_w64(m_ok)) {
                    const float om = 1.0f - alpha;
                    const float rcp1ma = __builtin_amdgcn_rcpf(om);
                    T = T * rcp1ma;
                    const float c0 = c.x, c1 = c.y, c2 = c.z;
                    float dL_dalpha = (c0 - ac0) * dp0 + (c1 - ac1) * dp1 + (c2 - ac2) * dp2;
                    // backward.cu:505-507's accum_rec = last_alpha * last_color + (1 - last_alpha) * accum_rec, evaluated HERE for the
                    // next contributor instead of there from a saved (last_alpha, last_color): same operands, same result, and
                    // four register moves per pair fewer (the first contributor sees 0 either way)
                    ac0 = alpha * c0 + om * ac0;
                    ac1 = alpha * c1 + om * ac1;
                    ac2 = alpha * c2 + om * ac2;
                    dL_dalpha *= T;
                    dL_dalpha += tfbg * rcp1ma;
                    dch = alpha * T; u = G * dL_dalpha;
                }
                alive |= 1u << jb;
                pbuf[wave][jb][lane] = make_float2(u, dch);
            }
            if (!alive) continue;
            GS_COUNT(8, 1); GS_COUNT(9, __popc(alive)); GS_COUNT(14, __popc(alive) == 1 ? 1 : 0); GS_COUNT(15, __popc(alive) == 2 ? 1 : 0);
            __builtin_amdgcn_wave_barrier();            // same wave: the LDS executes its accesses in order, no s_barrier needed
            // ---- transposed phase: lane (k, jj) sums pixels k, k + 8, ..., k + 56 of instance jj ----
            const uint32_t j = g * GB + jj;
            const float4 a = s0[j];
            const float4 b = s1[j];
            float v8[8], Cb = 0.f;
            const float2* urow = &pbuf[wave][jj][k];
            if constexpr (B8) {
                // Round 5: SEPARABLE moments.  In the 8 x 8 block lane (k, jj) walks ONE column (pixel k + 8 i = column k of row i): dx is the
                // same for its eight pixels and dy = (a.y - sy0) - i, so the five geometric sums are polynomials in (dx, by = a.y - sy0) of
                // THREE row moments of u -- M0 = sum u, M1 = sum u i, M2 = sum u i^2, whose weights are compile-time constants:
                //   sum u dx = dx M0, sum u dx^2 = dx^2 M0, sum u dy = by M0 - M1, sum u dx dy = dx (by M0 - M1), sum u dy^2 = by^2 M0 - 2 by M1 + M2
                // Six FMA-class instructions per pixel (three moments, three colour sums) instead of twelve; the polynomials cost nine once
                // per phase.  SQ_INSTS_VALU per launch at 3 M: 230 M -> see profiles/r05_*.  The rounding differs from summing u dx^2 term by
                // term but is of the same size (both are relative to |by|^2 sum |u|); gradients stay within the 1e-5 bar of the fp64 oracle.
                float M0 = 0.f, M1 = 0.f, M2 = 0.f, Cr = 0.f, Cg = 0.f;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const float2 ud = urow[i * 8];
                    asm volatile("" ::: "memory");                       // keeps the 8-byte reads apart: merged into ds_read2_b64 they cost 8 LDS cycles per pair, apart 2 each
                    const float uu = ud.x, dd = ud.y;
                    M0 += uu;
                    if (i == 1) { M1 += uu; M2 += uu; }
                    else if (i > 1) { M1 = __builtin_fmaf(uu, (float)i, M1); M2 = __builtin_fmaf(uu, (float)(i * i), M2); }
                    Cr = __builtin_fmaf(dd, dpr[i][0], Cr); Cg = __builtin_fmaf(dd, dpr[i][1], Cg); Cb = __builtin_fmaf(dd, dpr[i][2], Cb);
                }
                const float dx = a.x - pxk0, by = a.y - sy0;
                const float m0 = M0 * b.y, m1 = M1 * b.y, m2 = M2 * b.y;  // the opacity factor of dL/dG = opacity * dL/dalpha, once
                const float Sy = __builtin_fmaf(by, m0, -m1);
                const float Syy = __builtin_fmaf(by, Sy, -__builtin_fmaf(by, m1, -m2));
                const float Sx = dx * m0, Sxx = dx * Sx, Sxy = dx * Sy;
                v8[0] = Sx * a.z + Sy * a.w; v8[1] = Sy * b.x + Sx * a.w; v8[2] = Sxx; v8[3] = Sxy; v8[4] = Syy; v8[5] = M0; v8[6] = Cr; v8[7] = Cg;
            } else {
                const float dxa = a.x - pxk0, dxb = a.x - pxk1;
                float Sx = 0.f, Sy = 0.f, Sxx = 0.f, Sxy = 0.f, Syy = 0.f, Su = 0.f, Cr = 0.f, Cg = 0.f;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const float2 ud = urow[i * 8];
                    asm volatile("" ::: "memory");
                    const float uu = ud.x, dd = ud.y;
                    const float dx = (i & 1) ? dxb : dxa;
                    const float dy = a.y - (sy0 + (float)(i >> 1));
                    const float gxv = uu * dx, gyv = uu * dy;            // the opacity factor of dL/dG = opacity * dL/dalpha is applied once, below
                    Sx += gxv; Sy += gyv;
                    Sxx = __builtin_fmaf(gxv, dx, Sxx); Sxy = __builtin_fmaf(gxv, dy, Sxy); Syy = __builtin_fmaf(gyv, dy, Syy);
                    Su += uu;
                    Cr = __builtin_fmaf(dd, dpr[i][0], Cr); Cg = __builtin_fmaf(dd, dpr[i][1], Cg); Cb = __builtin_fmaf(dd, dpr[i][2], Cb);
                }
                Sx *= b.y; Sy *= b.y; Sxx *= b.y; Sxy *= b.y; Syy *= b.y;
                v8[0] = Sx * a.z + Sy * a.w; v8[1] = Sy * b.x + Sx * a.w; v8[2] = Sxx; v8[3] = Sxy; v8[4] = Syy; v8[5] = Su; v8[6] = Cr; v8[7] = Cg;
            }
            const float tot = group8_sum8_transposed(v8, lane);     // lane (k, jj): total of value k for instance jj
            const float tb = group8_sum(Cb);
            if ((alive >> jj) & 1u) {
                lds_add_f32(&acc[j][k], tot * commit_scale);
                if (k == 0u) lds_add_f32(&acc[j][8], tb);
            }
        }
        __syncthreads();
        // commit: 16 adjacent lanes per staged instance, lane q < 9 adds sum q to float q of the Gaussian's 64-byte record
        for (uint32_t e = t; e < cnt * 16u; e += NT) {
            const uint32_t slot = e >> 4, q = e & 15u;
            if (q < 9u) {
                const float v = acc[slot][q];
                if (v != 0.f) atomicAdd(grec + (size_t)sid[slot] * GREC + q, v);
            }
        }
    }
#ifdef GSRAST_DEBUG_TIMING
    if (t == 0) {       // the longest workgroup (10 ns ticks), its list length, the launch's first start and last end
        const long long t1 = wall_clock64();
        atomicMax(&g_dbg[10], (unsigned long long)(t1 - dbg_t0));
        atomicMin(&g_dbg[11], (unsigned long long)dbg_t0);
        atomicMax(&g_dbg[12], (unsigned long long)t1);
        if ((unsigned long long)(t1 - dbg_t0) >= g_dbg[10]) g_dbg[13] = n;
    }
#endif
}

} // namespace gsrast
