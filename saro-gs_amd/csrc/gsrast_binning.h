// gsrast_binning.h -- tile binning: prefix scan, stable LSD radix sort, instance emission and
// per-tile range detection.  Integer work only; every result is bit-exact by construction.
//
// What the reference does (rasterizer_impl.cu:277-319): inclusive scan of tiles_touched ->
// emit one 64-bit key (tile << 32 | depth bits) per (Gaussian, tile) instance -> ONE stable radix
// sort of all R instances over 32+log2(T) key bits -> boundary detection.
//
// What this build does instead (same final order):
//   1. stable sort of the P Gaussians by depth bits (4 x 8-bit passes over 8 B pairs);
//   2. default ("run-compressed", second half of this file): one 10-byte COLUMN RUN (x, y0, h) per tile column of a
//      Gaussian's rectangle, rows clipped to the alpha >= 1/255 ellipse; the runs are sorted by column (one pass over
//      Q ~ R/6 elements), then ONE instance-level pass by tile row expands them on the fly and writes every instance
//      once (4 bytes); the tile ranges come from the sorted runs and the scanned row histogram;
//      option binning=1 (and images with > 256 tile rows / > 65536 tiles): scan of tiles_touched in depth order, one
//      (tile id, Gaussian) pair per instance, stable sort of the R instances by tile id (2 passes at 1080p).
// A stable sort by tile of a depth-ordered (ties: Gaussian-index-ordered) sequence is exactly the
// stable sort by (tile, depth) of the index-ordered sequence the reference produces, so
// point_list and the tile ranges are identical to the reference's, bit for bit (with tile_clip=0;
// with row clipping the lists are ordered subsequences of the reference's and the outputs are bit-identical).
#pragma once
#include "gsrast_common.h"

namespace gsrast {
// Occupancy caps of the latency-bound binning kernels (A/B: -DGSRAST_<K>_WAVES=n compiles kernel K for n waves per SIMD; round 6)
#define GSRAST_OCC_ATTR(n) __attribute__((amdgpu_waves_per_eu(n, n)))
#ifdef GSRAST_RSCAT_WAVES
#define GSRAST_RSCAT_OCC GSRAST_OCC_ATTR(GSRAST_RSCAT_WAVES)
#else
#define GSRAST_RSCAT_OCC
#endif
#ifdef GSRAST_BSCAT_WAVES
#define GSRAST_BSCAT_OCC GSRAST_OCC_ATTR(GSRAST_BSCAT_WAVES)
#else
#define GSRAST_BSCAT_OCC
#endif
#ifdef GSRAST_BSORT_WAVES
#define GSRAST_BSORT_OCC GSRAST_OCC_ATTR(GSRAST_BSORT_WAVES)
#else
#define GSRAST_BSORT_OCC
#endif
#ifdef GSRAST_EMIT_WAVES
#define GSRAST_EMIT_OCC GSRAST_OCC_ATTR(GSRAST_EMIT_WAVES)
#else
#define GSRAST_EMIT_OCC
#endif
#ifdef GSRAST_ROWS_WAVES
#define GSRAST_ROWS_OCC GSRAST_OCC_ATTR(GSRAST_ROWS_WAVES)
#else
#define GSRAST_ROWS_OCC
#endif

// ------------------------------------------------------------------------------------------
// Scan (u32).  Three kernels per level: block sums -> scan of sums (recursive) -> apply.
// `idx` (optional) gathers the input: in[i] := src[idx[i]].
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if (lane_id() >= (unsigned)d) v += t;
    }
    return v;
}

// Block-wide exclusive scan of per-thread totals (256 threads). Returns exclusive prefix; *total = block sum.
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* total)
{
    __shared__ uint32_t wsum[4];
    const unsigned lane = lane_id(), wave = threadIdx.x >> 6;
    uint32_t inc = wave_incl_scan(v);
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) { uint32_t s = wsum[w]; if (w < (int)wave) base += s; tot += s; }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

// value i of the scanned sequence: src[idx[i]] (gather), src[i], or -- when `runs` is set -- the height
// field (.y >> 16) of the i-th column run
__device__ __forceinline__ uint32_t scan_load(const uint32_t* src, const uint32_t* idx, const uint2* runs, uint32_t i)
{
    if (runs) return runs[i].y >> 16;
    return idx ? src[idx[i]] : src[i];
}

__global__ void __launch_bounds__(256)
scan_block_sums_kernel(const uint32_t* __restrict__ src, const uint32_t* __restrict__ idx, const uint2* __restrict__ runs,
                       uint32_t n, uint32_t* __restrict__ sums,
                       const uint32_t* __restrict__ aux /* optional: a second array that is only summed */,
                       uint32_t* __restrict__ aux_sums)
{
    const uint32_t base = blockIdx.x * SC_CHUNK + threadIdx.x * 16;
    uint32_t s = 0, s2 = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        uint32_t i = base + k;
        if (i < n) { s += scan_load(src, idx, runs, i); if (aux) s2 += aux[i]; }
    }
    uint32_t tot;
    block_excl_scan(s, &tot);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
    if (aux) {      // a block's sum fits 32 bits (4096 entries of at most 2^16 tiles); the grand total is 64-bit
        block_excl_scan(s2, &tot);
        if (threadIdx.x == 0) aux_sums[blockIdx.x] = tot;
    }
}

// single block: in-place exclusive scan of m values (loops in chunks of 4096 with a carry)
__global__ void __launch_bounds__(256)
scan_single_block_kernel(uint32_t* __restrict__ data, uint32_t m, const uint32_t* __restrict__ aux_sums,
                         uint32_t* __restrict__ aux_total_lo, uint32_t* __restrict__ aux_total_hi)
{
    if (aux_sums) {
        __shared__ unsigned long long part[256];
        unsigned long long a = 0;
        for (uint32_t i = threadIdx.x; i < m; i += 256) a += aux_sums[i];
        part[threadIdx.x] = a;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) { if ((int)threadIdx.x < st) part[threadIdx.x] += part[threadIdx.x + st]; __syncthreads(); }
        if (threadIdx.x == 0) { *aux_total_lo = (uint32_t)part[0]; *aux_total_hi = (uint32_t)(part[0] >> 32); }
    }
    uint32_t carry = 0;
    for (uint32_t c0 = 0; c0 < m; c0 += SC_CHUNK) {
        const uint32_t base = c0 + threadIdx.x * 16;
        uint32_t v[16], s = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) { v[k] = (base + k < m) ? data[base + k] : 0u; s += v[k]; }
        uint32_t tot;
        uint32_t ex = block_excl_scan(s, &tot) + carry;
#pragma unroll
        for (int k = 0; k < 16; k++) { if (base + k < m) data[base + k] = ex; ex += v[k]; }
        carry += tot;
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256)
scan_apply_kernel(const uint32_t* __restrict__ src, const uint32_t* __restrict__ idx, const uint2* __restrict__ runs,
                  uint32_t n, const uint32_t* __restrict__ sums_scanned /* exclusive, may be null for 1 block */,
                  uint32_t* __restrict__ dst, int inclusive, uint32_t* __restrict__ total_out)
{
    const uint32_t base = blockIdx.x * SC_CHUNK + threadIdx.x * 16;
    uint32_t v[16], s = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        uint32_t i = base + k;
        v[k] = (i < n) ? scan_load(src, idx, runs, i) : 0u;
        s += v[k];
    }
    uint32_t tot;
    uint32_t ex = block_excl_scan(s, &tot) + (sums_scanned ? sums_scanned[blockIdx.x] : 0u);
#pragma unroll
    for (int k = 0; k < 16; k++) {
        uint32_t i = base + k;
        if (i < n) dst[i] = inclusive ? ex + v[k] : ex;
        ex += v[k];
    }
    if (total_out && blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) *total_out = ex;
}

// Element count of a launch that was sized for a CAPACITY before the host knew the real count: the grid and the
// histogram stride follow the capacity `n`, the kernel works on min(*n_dev, n) elements (blocks past the end find an
// empty chunk and contribute zeros).  n_dev == nullptr: `n` is exact.
__device__ __forceinline__ uint32_t dev_count(uint32_t n, const uint32_t* __restrict__ n_dev)
{
    if (!n_dev) return n;
    const uint32_t m = *n_dev;
    return m < n ? m : n;
}

// ------------------------------------------------------------------------------------------
// Stable LSD radix sort pass on (u32 key, u32 value) pairs, 8-bit digit.
// A block owns RS_CHUNK consecutive elements; wave w owns a contiguous quarter, visited in
// rounds of 64 (coalesced).  Ranks inside a round come from a ballot match, so equal digits keep
// their input order (stability) and LDS counters see no conflicts.
// Histogram: order does not matter here, so plain LDS atomics (per-wave private counters keep
// contention inside a wave; a uniform digit costs at most 64 LDS cycles per round, still far below
// the HBM time of the keys).
// Adaptive pass count of the 32-bit depth sort (SortAdapt, host side: radix_sort): the first pass also reduces the minimum and
// maximum of the visible keys (per block here, over the blocks in one extra workgroup of its row scan); keys that agree in
// their top byte -- view depths inside one [2^(2k-127), 2^(2k-125)) bracket, e.g. everything between 2 and 8 -- are fully sorted
// after THREE 8-bit passes, so the fourth pass's three kernels return at once.  So that a depth range which merely STRADDLES such
// a boundary (1.9 .. 6) qualifies too, passes two to four take their digits from key - base, base = minimum key with its low
// byte cleared (pass one has already used the raw low byte, which subtracting `base` does not change): the verdict is then
// "maximum - base fits 24 bits", i.e. any depth range narrower than 2^24 float steps.  The host cannot know that when it enqueues
// them, so every kernel reads the verdict from device memory (`sig` = significant key bits) and passes two and three pick their
// buffers accordingly (A -> B -> C -> A instead of A -> B -> A -> B -> A): the sorted sequence ends in A either way.
// RA_ASSUME: the host did not even enqueue the fourth pass, because the previous forward of the same context found a short span
// (consecutive views of one scene): every kernel then behaves as if the verdict were "short", and the first pass's extra
// workgroup raises sig[2] when that was wrong -- the host reads it back with the instance counts and repeats the sort with four
// passes (gsrast_forward: the same redo path as an undersized speculative binning buffer).
enum : int { RA_MINMAX = 1, RA_IN_ALT = 2, RA_SKIP = 4, RA_OUT_ALT = 8, RA_LAST_IF_SHORT = 16, RA_ASSUME = 32 };
__device__ __forceinline__ bool sort_is_short(const uint32_t* __restrict__ sig, int adapt = 0)      // sig[1] = base, sig[2] = wrong assumption
{
    return (adapt & RA_ASSUME) || (sig && sig[0] <= 24u);
}

template <typename KeyT, int ITEMS>
__global__ void __launch_bounds__(RS_THREADS)
radix_hist_kernel(const KeyT* __restrict__ keys, uint32_t n, const uint32_t* __restrict__ n_dev, int shift, uint32_t mask,
                  uint32_t* __restrict__ block_hist, uint32_t nblk,
                  const KeyT* __restrict__ keys_alt = nullptr, const uint32_t* __restrict__ sig = nullptr, int adapt = 0,
                  uint32_t* __restrict__ block_minmax = nullptr, const uint32_t* __restrict__ pred = nullptr /* predicated launch: returns unless *pred != 0 */)
{
    __shared__ uint32_t cnt[4][256];
    __shared__ uint32_t s_mm[2];
    if (pred && *pred == 0u) return;
    const bool is_short = sort_is_short(sig, adapt);
    if ((adapt & RA_SKIP) && is_short) return;
    if ((adapt & RA_IN_ALT) && is_short) keys = keys_alt;
    if ((adapt & RA_MINMAX) && threadIdx.x == 0) { s_mm[0] = 0xFFFFFFFFu; s_mm[1] = 0u; }
    const uint32_t base = sig ? sig[1] : 0u;
    n = dev_count(n, n_dev);
    const unsigned lane = lane_id(), wave = threadIdx.x >> 6;
    for (int k = threadIdx.x; k < 1024; k += RS_THREADS) (&cnt[0][0])[k] = 0;
    __syncthreads();
    const uint32_t wbase = blockIdx.x * (RS_THREADS * ITEMS) + wave * ((RS_THREADS * ITEMS) / 4);
    uint32_t key[ITEMS];
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const uint32_t i = wbase + r * 64 + lane;
        key[r] = i < n ? keys[i] : 0u;
    }
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const uint32_t i = wbase + r * 64 + lane;
        if (i < n) atomicAdd(&cnt[wave][((key[r] - base) >> shift) & mask], 1u);
    }
    if (adapt & RA_MINMAX) {     // key 0xFFFFFFFF marks a culled Gaussian (preprocess_fwd_kernel).  With four passes it sorts to the end; on the
                                 // three-pass (short) path its digits come from the low 24 bits of (key - base) and it may land ANYWHERE in
                                 // `order` -- harmless: no consumer relies on its place (emit_column_runs / emit_instances skip a Gaussian by
                                 // its width / tile count of 0, never by its position); it is only excluded from the minimum / maximum here
        uint32_t lo = 0xFFFFFFFFu, hi = 0u;
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            const uint32_t i = wbase + r * 64 + lane;
            if (i < n && (uint32_t)key[r] != 0xFFFFFFFFu) { lo = min(lo, (uint32_t)key[r]); hi = max(hi, (uint32_t)key[r]); }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { lo = min(lo, (uint32_t)__shfl_xor(lo, d, 64)); hi = max(hi, (uint32_t)__shfl_xor(hi, d, 64)); }
        if (lane == 0) { atomicMin(&s_mm[0], lo); atomicMax(&s_mm[1], hi); }
    }
    __syncthreads();
    const uint32_t t = threadIdx.x;
    if (t <= mask) block_hist[(size_t)t * nblk + blockIdx.x] = cnt[0][t] + cnt[1][t] + cnt[2][t] + cnt[3][t];   // rows past the digit range are never read
    if ((adapt & RA_MINMAX) && t < 2) block_minmax[2 * blockIdx.x + t] = s_mm[t];
}

// One workgroup per digit: in-place exclusive scan of that digit's row of block counts, row total
// out.  The scatter kernel adds the exclusive prefix over digit totals itself, so a radix pass is
// three launches (histogram, row scan, scatter) instead of five.
__global__ void __launch_bounds__(256)
radix_rowscan_kernel(uint32_t* __restrict__ block_hist, uint32_t nblk, uint32_t* __restrict__ digit_total,
                     uint32_t* __restrict__ sig = nullptr, int adapt = 0, const uint32_t* __restrict__ block_minmax = nullptr, uint32_t ndigits = 0,
                     const uint32_t* __restrict__ pred = nullptr)
{
    if (pred && *pred == 0u) return;
    if ((adapt & RA_SKIP) && sort_is_short(sig, adapt)) return;
    if ((adapt & RA_MINMAX) && blockIdx.x == ndigits) {      // the extra workgroup of the first pass: min / max over the blocks -> sig
        __shared__ uint32_t mm[2];
        if (threadIdx.x == 0) { mm[0] = 0xFFFFFFFFu; mm[1] = 0u; }
        __syncthreads();
        uint32_t lo = 0xFFFFFFFFu, hi = 0u;
        for (uint32_t b = threadIdx.x; b < nblk; b += 256) { lo = min(lo, block_minmax[2 * b]); hi = max(hi, block_minmax[2 * b + 1]); }
        atomicMin(&mm[0], lo); atomicMax(&mm[1], hi);
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t b = mm[0] > mm[1] ? 0u : (mm[0] & ~0xFFu);
            const uint32_t span = mm[0] > mm[1] ? 0u : mm[1] - b;
            sig[0] = span ? 32u - (uint32_t)__builtin_clz(span) : 0u;
            sig[1] = b;
            sig[2] = ((adapt & RA_ASSUME) && sig[0] > 24u) ? 1u : 0u;
        }
        return;
    }
    uint32_t* row = block_hist + (size_t)blockIdx.x * nblk;
    uint32_t carry = 0;
    // (Measured and dropped, round 3: requesting the next round's eight values before this round is scanned.  On gfx9 loads and stores
    // share vmcnt and the compiler must drain the counter when both kinds are outstanding, so every round still waits for the previous
    // round's stores: 26.0 -> 26.0 us for the ~20 000-entry rows of the tile-row pass at 3 M Gaussians.)
    for (uint32_t c0 = 0; c0 < nblk; c0 += 256 * 8) {
        const uint32_t base = c0 + threadIdx.x * 8;
        uint32_t v[8], sum = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) { v[k] = (base + k < nblk) ? row[base + k] : 0u; sum += v[k]; }
        uint32_t tot;
        uint32_t ex = block_excl_scan(sum, &tot) + carry;
#pragma unroll
        for (int k = 0; k < 8; k++) { if (base + k < nblk) row[base + k] = ex; ex += v[k]; }
        carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) digit_total[blockIdx.x] = carry;
}

template <typename KeyT, typename ValT, int ITEMS>
__global__ void __launch_bounds__(RS_THREADS) GSRAST_RSCAT_OCC
radix_scatter_kernel(const KeyT* __restrict__ keys_in, const ValT* __restrict__ vals_in,
                     KeyT* __restrict__ keys_out, ValT* __restrict__ vals_out, uint32_t n, const uint32_t* __restrict__ n_dev,
                     int shift, uint32_t mask,
                     const uint32_t* __restrict__ hist_scanned /* per-digit exclusive row scans */,
                     const uint32_t* __restrict__ digit_total, uint32_t nblk,
                     const uint2* __restrict__ gather_rect /* optional, last depth pass only */,
                     uint32_t* __restrict__ gather_tiles, uint32_t* __restrict__ gather_width,
                     const KeyT* __restrict__ keys_in_alt = nullptr, const ValT* __restrict__ vals_in_alt = nullptr,
                     KeyT* __restrict__ keys_out_alt = nullptr, ValT* __restrict__ vals_out_alt = nullptr,
                     const uint32_t* __restrict__ sig = nullptr, int adapt = 0, const uint32_t* __restrict__ pred = nullptr)
{
    if (pred && *pred == 0u) return;
    {
        const bool is_short = sort_is_short(sig, adapt);
        if ((adapt & RA_SKIP) && is_short) return;
        if ((adapt & RA_IN_ALT) && is_short) { keys_in = keys_in_alt; vals_in = vals_in_alt; }
        if ((adapt & RA_OUT_ALT) && is_short) { keys_out = keys_out_alt; vals_out = vals_out_alt; }
        if ((adapt & RA_LAST_IF_SHORT) && !is_short) gather_rect = nullptr;      // this pass gathers only when it is the last one
    }
    const uint32_t base = (sig && shift > 0) ? sig[1] : 0u;      // pass one runs before `base` exists and does not need it (low byte of base = 0)
    n = dev_count(n, n_dev);
    // Ranks -> block-local order in LDS -> coalesced write-out: after the exchange consecutive lanes
    // hold consecutive elements of the same digit, whose global destinations are consecutive too.
    __shared__ uint32_t cnt[4][256];
    __shared__ uint32_t dstart[256];       // first block-local slot of each digit
    __shared__ uint32_t gbase[256];        // global destination of that slot
    __shared__ KeyT xk[(RS_THREADS * ITEMS)];
    __shared__ ValT xv[(RS_THREADS * ITEMS)];
    const unsigned lane = lane_id(), wave = threadIdx.x >> 6;
    for (int k = threadIdx.x; k < 1024; k += RS_THREADS) (&cnt[0][0])[k] = 0;
    __syncthreads();
    volatile uint32_t* wc = cnt[wave];
    const uint32_t wbase = blockIdx.x * (RS_THREADS * ITEMS) + wave * ((RS_THREADS * ITEMS) / 4);
    const int nbits = 32 - __builtin_clz(mask);          // digit width of this pass (mask = 2^w - 1)
    uint32_t key[ITEMS], rk[ITEMS];
    ValT val[ITEMS];
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const uint32_t i = wbase + r * 64 + lane;
        const bool valid = i < n;
        key[r] = valid ? (uint32_t)keys_in[i] : 0xFFFFFFFFu;
        val[r] = valid ? vals_in[i] : ValT{};
    }
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const uint32_t i = wbase + r * 64 + lane;
        const bool valid = i < n;
        const uint32_t d = ((key[r] - base) >> shift) & mask;
        const uint64_t m = wave_match8(d, valid, nbits);
        const uint32_t prev = wc[d];
        const uint32_t below = (uint32_t)__popcll(m & lanemask_lt());
        if (valid && below == 0) wc[d] = prev + (uint32_t)__popcll(m);
        rk[r] = prev + below;
    }
    __syncthreads();
    {
        const uint32_t t = threadIdx.x;
        const uint32_t c0 = cnt[0][t], c1 = cnt[1][t], c2 = cnt[2][t], c3 = cnt[3][t];
        uint32_t tot, gtot;
        const uint32_t start = block_excl_scan(c0 + c1 + c2 + c3, &tot);
        const bool used = t <= mask;                                      // the row scan ran for mask + 1 digits only
        uint32_t before = 0, row_total = 0;
        if (digit_total) {
            if (used) { row_total = digit_total[t]; before = hist_scanned[(size_t)t * nblk + blockIdx.x]; }
        } else if (used) {
            // few blocks (nblk <= RS_SELF_SCAN_BLOCKS): the histogram rows are raw counts and every block sums its own
            // prefix -- one launch per pass less, which is what a small sort costs (launch latency, not bandwidth)
            const uint32_t* row = hist_scanned + (size_t)t * nblk;
            for (uint32_t b = 0; b < nblk; b++) { const uint32_t c = row[b]; row_total += c; before += b < blockIdx.x ? c : 0u; }
        }
        const uint32_t dbase = block_excl_scan(row_total, &gtot);         // elements with a smaller digit, globally
        dstart[t] = start;
        gbase[t] = dbase + before;
        cnt[0][t] = start; cnt[1][t] = start + c0; cnt[2][t] = start + c0 + c1; cnt[3][t] = start + c0 + c1 + c2;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const uint32_t i = wbase + r * 64 + lane;
        if (i < n) {
            const uint32_t d = ((key[r] - base) >> shift) & mask;
            const uint32_t slot = cnt[wave][d] + rk[r];
            xk[slot] = (KeyT)key[r]; xv[slot] = val[r];
        }
    }
    __syncthreads();
    const uint32_t b0 = blockIdx.x * (RS_THREADS * ITEMS);      // may lie past n in a capacity-sized launch
    const uint32_t nvalid = b0 >= n ? 0u : ((n - b0) < (uint32_t)(RS_THREADS * ITEMS) ? (n - b0) : (uint32_t)(RS_THREADS * ITEMS));
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const uint32_t slot = r * RS_THREADS + threadIdx.x;
        if (slot < nvalid) {
            const KeyT kk = xk[slot];
            const uint32_t k = (uint32_t)kk;
            const uint32_t d = ((k - base) >> shift) & mask;
            const uint32_t pos = gbase[d] + (slot - dstart[d]);
            keys_out[pos] = kk;
            const ValT vv = xv[slot];
            vals_out[pos] = vv;
            if constexpr (sizeof(ValT) == 4) {
                if (gather_rect) {   // last depth pass: tile count and rectangle width of each Gaussian, in depth order
                    const uint2 rc = gather_rect[vv];
                    const uint32_t w = (rc.y & 0xFFFFu) - (rc.x & 0xFFFFu), h = (rc.y >> 16) - (rc.x >> 16);
                    if (gather_tiles) gather_tiles[pos] = w * h;
                    gather_width[pos] = w;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Bucket depth sort: the default order-by-depth of the P Gaussians for the run-compressed binning (the LSD radix sort above is the
// fallback and serves the instance-level binning).
// An LSD radix pass costs three dependent launches, the scan that follows three more, and at 1e6 Gaussians all of them are launch /
// latency bound (sort 85-110 us + scan 23 us for 16 MB of keys).  Three launches do the same job:
//   depth_bucket_scatter_kernel  every visible Gaussian goes to bucket floor(NB * CDF(z)) -- monotone in z -- where CDF is the running
//       sum of the sampled depth histogram preprocess_fwd left (round 4, gsrast_common.h: buckets of equal POPULATION, whatever the
//       depth distribution; rounds 2-3 cut [zmin, zmax] into equal intervals, which overflowed on any peak);
//       NB ~ P / 256 buckets keep a few hundred elements each.  No scan: a bucket owns a fixed slab of (key, index)
//       slots, a workgroup ranks its 2048 elements per bucket with LDS atomics and reserves slab space with ONE returning global
//       atomic per non-empty bucket.  Counters and slabs are kept PER XCD (workgroup b runs on XCD b mod 8; each XCD has its own
//       L2): with one counter set all eight L2s fight over the same 128 cache lines and the atomics alone cost 15 us.
//   depth_bucket_sort_kernel     one workgroup per bucket: collects the bucket's eight sub-slabs, sorts them in LDS on the 64-bit
//       composite (depth bits << 32 | index) -- a total order, so the result does not depend on the arrival order and ties fall in
//       index order exactly as under the stable radix sort -- and writes, IN THE BUCKET'S OWN SLOT RANGE, the sorted ids and the
//       inclusive scan of their tile-rectangle widths, plus the bucket's totals.  Nothing here needs a prefix over buckets.
//   depth_bucket_scan_kernel     one workgroup: exclusive scan of the buckets' width totals (-> first column run of each bucket, Q),
//       sum of their tile counts (-> num_rendered), the overflow verdict.
// emit_column_runs_kernel then runs one workgroup per bucket.  Culled Gaussians (key ~0) are never touched.
// A scene whose depths pile up beyond the histogram's resolution (more Gaussians in one bucket than its slab holds: thousands at
// one depth) raises a flag the host reads back with the instance counts; the
// forward then repeats the sort with the radix passes and the context uses those for its next calls (gsrast_forward).
// PREDICTED CUT (gsrast_common.h): every tile's cut depth from this call's own opacity mass.  One workgroup = a 16 x 16 block of tiles of which
// the inner ones are written (the ring around them only feeds the 3 x 3 maxima), one lane per tile: TAU_COPIES x TAU_BINS / 4 = 32
// sixteen-byte loads, a running sum, the first bin whose far edge lies behind tau_req of mean alpha mass.
#ifndef GSRAST_TAU_BLK
#define GSRAST_TAU_BLK 16
#endif
constexpr int TAU_BLK = GSRAST_TAU_BLK;      // tiles per workgroup side (16: 45 workgroups at 1080p; 8 -- 240 one-wave workgroups -- measured equal, 1.160 ms per cold step either way)
constexpr int TAU_TILE = TAU_BLK - 2;
__global__ void __launch_bounds__(TAU_BLK * TAU_BLK)
tau_cut_kernel(const uint32_t* __restrict__ tau_hist /* [TAU_COPIES][ntiles][TAU_BINS] */, uint32_t ntiles, int gx, int gy,
               TauBins tau_bins, uint32_t tau_req_x256 /* tau_req in the table's unit: pixels^2 of alpha mass per tile = 256 x the mean */,
               const uint32_t* __restrict__ hint_sel /* or null (no pose table): [1] = 1: the pose has remembered cut depths */, int force /* 1: predicted cuts also for a pose the table knows */,
               uint32_t* __restrict__ zcut_used /* [ntiles] out */, int coarse_range /* 1: the depth histogram has no learned range (a context's first forward): no prediction */)
{
    __shared__ int s_bin[TAU_BLK][TAU_BLK];
    const bool known = hint_sel && hint_sel[1] != 0u;      // (1: the pose's own slot, 2: a near pose's, widened -- both tighter than a prediction)
    if (known && !force) return;                                   // (uniform) the remembered cut depths are already in zcut_used
    const int lx = (int)(threadIdx.x % (unsigned)TAU_BLK), ly = (int)(threadIdx.x / (unsigned)TAU_BLK);
    const int tx = (int)blockIdx.x * TAU_TILE - 1 + lx, ty = (int)blockIdx.y * TAU_TILE - 1 + ly;
    int cb = -1;                                                    // -1: not a tile of the image (ignored by its neighbours)
    if (tx >= 0 && ty >= 0 && tx < gx && ty < gy) {
        const uint32_t t = (uint32_t)ty * (uint32_t)gx + (uint32_t)tx;
        uint32_t h[TAU_BINS];
#pragma unroll
        for (int b = 0; b < TAU_BINS; b++) h[b] = 0u;
#pragma unroll
        for (int c = 0; c < TAU_COPIES; c++) {
            const uint4* src = reinterpret_cast<const uint4*>(tau_hist + ((size_t)c * ntiles + t) * TAU_BINS);
#pragma unroll
            for (int q = 0; q < TAU_BINS / 4; q++) { const uint4 v = src[q]; h[4 * q] += v.x; h[4 * q + 1] += v.y; h[4 * q + 2] += v.z; h[4 * q + 3] += v.w; }
        }
        cb = TAU_BINS;                                              // TAU_BINS: the tile does not saturate inside the table (no cut)
        uint32_t run = 0;
#pragma unroll
        for (int b = 0; b < TAU_BINS - 1; b++) {                    // (the far tail bin has no far edge)
            run += h[b];
            if (run >= tau_req_x256 && cb == TAU_BINS) cb = b;
        }
        if (coarse_range) cb = TAU_BINS;
    }
    s_bin[ly][lx] = cb;
    __syncthreads();
    if (cb < 0 || lx == 0 || ly == 0 || lx == TAU_BLK - 1 || ly == TAU_BLK - 1) return;
    int m = cb;
#pragma unroll
    for (int dy = -1; dy <= 1; dy++)
#pragma unroll
        for (int dx = -1; dx <= 1; dx++) { const int v = s_bin[ly + dy][lx + dx]; m = v > m ? v : m; }
    uint32_t z = ZCUT_NONE;
    if (m < TAU_BINS) z = tau_bin_far_edge((uint32_t)m, tau_bins);
    zcut_used[(uint32_t)ty * (uint32_t)gx + (uint32_t)tx] = z;
}

constexpr int BK_CAP = GSRAST_BK_CAP;    // slots per bucket = the largest bucket the LDS sort takes
constexpr int BK_XCD = 8;                // counter / slab sets
constexpr int BK_CAPX = BK_CAP / BK_XCD; // slots per (bucket, XCD) sub-slab
constexpr int BK_MAX_BUCKETS = 8192;
// Elements per lane of the scatter kernel.  The kernel is a chain of latencies (histogram -> LDS ranks -> returning global atomics ->
// slab stores), its workgroups all take about the same time, and a launch costs as many workgroup latencies as it has ROUNDS of
// resident workgroups: with 8 elements per lane (102 VGPRs, five workgroups per CU, 1280 resident) 3 M Gaussians are 1465
// workgroups -- two rounds, the second 14 % full: 117 us.  16 per lane (165 VGPRs, three per CU, 768 resident) makes them 733
// workgroups in ONE round: 92 us (measured at 3 M under pose cycling: sort_depth 0.150 -> 0.125 ms; 14 or 20 per lane, two rounds
// again: 0.158).  A round holds ~3.1 M elements whatever the choice (registers grow with the elements per lane), so the host
// picks 16 where that turns two rounds into one (depth_scatter_items) and 8 otherwise.
constexpr int BK_ITEMS = 8, BK_ITEMS_WIDE = 16;
constexpr uint32_t BK_ROUND = 1280u * 256u * BK_ITEMS, BK_ROUND_WIDE = 768u * 256u * BK_ITEMS_WIDE;      // elements of one round (256 CUs)
__host__ __device__ inline int depth_scatter_items(size_t n) { return n > BK_ROUND && n <= BK_ROUND_WIDE ? BK_ITEMS_WIDE : BK_ITEMS; }
#ifdef GSRAST_SCATTER_TIMING
__device__ unsigned long long g_scat[16];
#define SCAT_T(k) do { if (threadIdx.x == 0) { const long long now_ = wall_clock64(); atomicAdd(&g_scat[k], (unsigned long long)(now_ - scat_t)); scat_t = now_; } } while (0)
#else
#define SCAT_T(k) do { } while (0)
#endif
// TWO-LAUNCH form (round 6, COARSE = true + depth_bucket_refine_kernel): with ~P / 366 fine buckets nearly every element of a workgroup is alone in its bucket --
// one returning global atomic and one lone 16-byte store per element (3 M of each at 3 M Gaussians: 21 + 32 us of the kernel's 88, 135 MB written for 48 MB of
// elements: VERDICT r05 item 4a).  COARSE: the workgroup ranks and reserves per COARSE bucket (32 fine buckets: nb / 32 <= 256 of them), so a workgroup's elements of one
// coarse bucket are a run of ~16 consecutive slots (256 bytes, written within a microsecond from one compute unit) and the atomics are one per lane; the element carries
// its fine bucket's low five bits in bits 16-20 of the width word.  The refine kernel then runs one workgroup per (coarse bucket, XCD): it alone writes the 32 fine
// sub-slabs of its XCD, ranks in LDS, needs no global atomic and writes the fine counters once -- same slab, same counters as the one-launch form, so the sort and the
// emission behind it do not change.
constexpr int BK_CSHIFT = 5;                       // fine buckets per coarse bucket: 32
constexpr int BK_CCAP_MAX = (1 << BK_CSHIFT) * BK_CAPX;     // 4096 slots per (coarse bucket, XCD) at most: what 32 full fine sub-slabs hold (a coarse overflow is reported as a fine one: radix fallback)
// the coarse slab lives in the gradient records' memory (64 B per Gaussian, free until the forward's last blend): 4 P slots, a quarter full
__host__ __device__ inline uint32_t depth_coarse_cap(size_t P, uint32_t nb) { const size_t c = ((4 * P) / ((size_t)(nb >> BK_CSHIFT) * BK_XCD)) & ~(size_t)255; return (uint32_t)(c < (size_t)BK_CCAP_MAX ? c : (size_t)BK_CCAP_MAX); }
constexpr int BK_NBC_MAX = BK_MAX_BUCKETS >> BK_CSHIFT;      // 256
constexpr uint32_t BK_FINE_MASK = ((1u << BK_CSHIFT) - 1u) << 16;
template <int BK_ITEMS_T, bool COARSE = false>
__global__ void __launch_bounds__(256) GSRAST_BSCAT_OCC __attribute__((amdgpu_waves_per_eu(3)))      // (at least three waves per SIMD: 768 resident workgroups, ONE round at 3 M -- the coarse form came out at 177 VGPRs without it)
depth_bucket_scatter_kernel(const uint32_t* __restrict__ keys, const uint2* __restrict__ rect, const uint32_t* __restrict__ tiles,
                            uint32_t n, const uint32_t* __restrict__ zhist /* [ZH_COPIES][ZH_BINS]: sampled histogram of the visible depth keys (preprocess_fwd) */,
                            uint32_t zh_klo, int zh_shift /* its bins: 2^shift key steps each, from klo (gsrast_common.h) */,
                            uint32_t nb, uint32_t* __restrict__ gcount /* [8][nb]: per-XCD bucket counts (zeroed by preprocess_fwd) */,
                            uint4* __restrict__ slab /* [nb][8][BK_CAPX]: {depth key, id, rectangle width, tile count} -- what the sort kernel
                                                        needs of a Gaussian travels with it (gathering rect / tiles by id there cost 15 us) */,
                            uint32_t* __restrict__ bkey_out /* [nb + 1]: first key the bucket map sends to bucket b (approximately: the map's
                                                               inverse) -- the sort kernel spreads a bucket's elements over its sub-intervals by it */,
                            uint32_t* __restrict__ zbins_out /* scalars[SC_ZBINS]: first | last << 16 occupied bin (the host's next range hint) */,
                            // list cut (gsrast_common.h): this call's snapshot of the pose's per-tile cut depths, or null.  A Gaussian that
                            // lies behind the cut depth of EVERY tile of its rectangle is marked LATE (bit 31 of the width word): it keeps
                            // its place in the depth order but gets no column runs unless the forward has to fall back to the full lists
                            const uint32_t* __restrict__ zcut_used = nullptr, uint32_t ntiles_img = 0, uint32_t gx_tiles = 0,
                            uint32_t* __restrict__ n_late_out = nullptr,
                            unsigned long long* __restrict__ color_skip = nullptr /* [ceil(n / 64)] or null: bit i = Gaussian i is culled or
                                                                                  late: no list will hold it, the colour kernel need not evaluate it */,
                            uint32_t cshift = 1 /* the cells of the cut-depth table are (1 << cshift)^2 tiles: 2 x 2 up to 1080p-class images,
                                                   4 x 4 / 8 x 8 for larger ones (at most CUT_MAX_CELLS cells) */,
                            // LAYER mode (round 4): without remembered cut depths -- a pose the table does not know (layer_mode 1 and
                            // hint_sel[1] == 0), or no table at all (layer_mode 2) -- every tile's cut depth is ONE depth: the key below
                            // which the nearest `layer_frac` of the visible Gaussians lie (the bucket map's inverse).  The first pass
                            // then lists that layer only; the completion pass behind the blend lists the rest into the tiles that did
                            // not saturate inside it.  zcut_used (uninitialised or all "none") is filled with that key here
                            int layer_mode = 0, const uint32_t* __restrict__ hint_sel = nullptr, float layer_frac = 0.125f,
                            uint32_t* __restrict__ zcut_fill = nullptr, uint32_t ccap = 0 /* COARSE: slots per (coarse bucket, XCD) of the coarse slab (depth_coarse_cap) */)
{
    // per-bucket counters of this workgroup, two 16-bit counters per word (a workgroup has 2048 elements): 16 KB instead of 32 -- with
    // the 4 KB of the bucket map and the 6 KB of cut depths this latency-bound kernel keeps five workgroups per compute unit
    __shared__ uint32_t cnt[(COARSE ? BK_NBC_MAX : BK_MAX_BUCKETS) / 2];
    __shared__ uint32_t s_C[ZH_BINS + 1];          // running sum of (histogram + 1): strictly increasing
    __shared__ uint32_t s_fl[2];
    // the cut depths as maxima over cells of 2 x 2 tiles (4 x 4 / 8 x 8 for images of more than CUT_MAX_CELLS such cells), rounded UP to the 16 leading bits of the float (exponent + 7 mantissa bits:
    // within 0.8 % of the depth): a Gaussian is LATE when it lies behind every cell its rectangle touches -- typically four LDS reads,
    // no memory access in the loop; a larger cut only keeps more.  (The table must stay small: measured on the way: the tiles' own depths in
    // LDS, 16 KB: 77 -> 116 us at 3 M; 8 x 8 cells in LDS and the tiles' depths walked in global memory behind them: 152 us, a
    // dependent load per step of a divergent loop.)
    __shared__ uint16_t s_zc[CUT_MAX_CELLS];
    __shared__ uint32_t s_late;
    const unsigned lane = lane_id();
    const uint32_t xcd = blockIdx.x & (BK_XCD - 1);
#ifdef GSRAST_SCATTER_TIMING
    long long scat_t = wall_clock64();
    if (threadIdx.x == 0) { atomicAdd(&g_scat[15], 1ull); atomicMin(&g_scat[14], (unsigned long long)scat_t); }
#endif
    if (threadIdx.x == 0) { s_late = 0u; s_fl[0] = 0xFFFFu; s_fl[1] = 0u; }
    for (uint32_t k = threadIdx.x; k < (COARSE ? nb >> BK_CSHIFT : nb) / 2u; k += 256) cnt[k] = 0u;
    // (the keys are requested first: their round trip passes under the construction of the bucket map)
    const uint32_t base = blockIdx.x * (256 * BK_ITEMS_T);
    uint32_t key[BK_ITEMS_T];
#pragma unroll
    for (int r = 0; r < BK_ITEMS_T; r++) { const uint32_t i = base + r * 256 + threadIdx.x; key[r] = i < n ? keys[i] : 0xFFFFFFFFu; }
    static_assert(ZH_BINS == 4 * 256, "one uint4 of the histogram per lane");
    {   // the bucket map: running sum of the sampled histogram, one pseudo-count per bin (an unsampled bin keeps a positive width)
        uint4 hv = reinterpret_cast<const uint4*>(zhist)[threadIdx.x];
#pragma unroll
        for (int x = 1; x < ZH_COPIES; x++) {      // (one copy per XCD)
            const uint4 v = reinterpret_cast<const uint4*>(zhist)[x * (ZH_BINS / 4) + threadIdx.x];
            hv.x += v.x; hv.y += v.y; hv.z += v.z; hv.w += v.w;
        }
        // The histogram is a SAMPLE (a few tens of thousands of keys): a bin at the foot of the depth profile expects a handful of
        // samples and may hold a third of that, which sends three buckets' worth of Gaussians to one bucket (measured on the cube:
        // pose 29 of a 32-pose ring overflowed a sub-slab on every visit).  The middle bins are smoothed with the binomial weights
        // 1 4 6 4 1 (two passes of 1 2 1): the noise of a bin falls to a half, a straight ramp is unchanged; the wide tail bins keep
        // their own count.  Everything is in sixteenths of a sample from here on, plus a quarter sample per bin (an unsampled bin
        // keeps a positive width); the total stays far below 2^24, exact in the floats below.
        s_C[4 * threadIdx.x] = hv.x; s_C[4 * threadIdx.x + 1] = hv.y; s_C[4 * threadIdx.x + 2] = hv.z; s_C[4 * threadIdx.x + 3] = hv.w;
        __syncthreads();
        SCAT_T(0);      // histogram loaded
        uint32_t h0 = 16u * hv.x, h1 = 16u * hv.y, h2 = 16u * hv.z, h3 = 16u * hv.w;
        if (4u * threadIdx.x >= (uint32_t)ZH_TAIL && 4u * threadIdx.x < (uint32_t)(ZH_TAIL + ZH_MID)) {
            static_assert(ZH_TAIL % 4 == 0 && ZH_MID % 4 == 0, "a lane's four bins lie on one side of the tails' boundaries");
            uint32_t w[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int i = (int)(4u * threadIdx.x) - 2 + q;
                w[q] = s_C[i < ZH_TAIL ? ZH_TAIL : (i >= ZH_TAIL + ZH_MID ? ZH_TAIL + ZH_MID - 1 : i)];
            }
            h0 = w[0] + 4u * w[1] + 6u * w[2] + 4u * w[3] + w[4];
            h1 = w[1] + 4u * w[2] + 6u * w[3] + 4u * w[4] + w[5];
            h2 = w[2] + 4u * w[3] + 6u * w[4] + 4u * w[5] + w[6];
            h3 = w[3] + 4u * w[4] + 6u * w[5] + 4u * w[6] + w[7];
        }
        h0 += 4u; h1 += 4u; h2 += 4u; h3 += 4u;
        __syncthreads();                         // (every lane has read its neighbours' raw counts: the running sum may replace them)
        uint32_t tot;
        const uint32_t ex = block_excl_scan(h0 + h1 + h2 + h3, &tot);
        s_C[4 * threadIdx.x] = ex; s_C[4 * threadIdx.x + 1] = ex + h0; s_C[4 * threadIdx.x + 2] = ex + h0 + h1; s_C[4 * threadIdx.x + 3] = ex + h0 + h1 + h2;
        if (threadIdx.x == 255) s_C[ZH_BINS] = tot;
        if (blockIdx.x == 0) {          // (uniform) what the host derives its next range hint from
            const uint32_t any = hv.x | hv.y | hv.z | hv.w;
            if (any) {
                const uint32_t f = 4u * threadIdx.x + (hv.x ? 0u : hv.y ? 1u : hv.z ? 2u : 3u), l = 4u * threadIdx.x + (hv.w ? 3u : hv.z ? 2u : hv.y ? 1u : 0u);
                atomicMin(&s_fl[0], f); atomicMax(&s_fl[1], l);
            }
        }
    }
    const bool layer = zcut_used && (layer_mode == 2 || (layer_mode == 1 && hint_sel[1] == 0u));      // (uniform)
    const uint32_t csz = 1u << cshift;
    const uint32_t gy_tiles = zcut_used ? ntiles_img / gx_tiles : 0u, cgx = (gx_tiles + csz - 1u) >> cshift, cgy = (gy_tiles + csz - 1u) >> cshift;
    if (layer) { }
    else if (zcut_used && cshift != 1u) {        // larger cells (images beyond the 1080p class): plain loops
        for (uint32_t c = threadIdx.x; c < cgx * cgy; c += 256) {
            const uint32_t cx = c % cgx, cy = c / cgx;
            uint32_t m = 0;
            for (uint32_t y = cy << cshift; y < min((cy + 1u) << cshift, gy_tiles); y++)
                for (uint32_t x = cx << cshift; x < min((cx + 1u) << cshift, gx_tiles); x++) m = max(m, zcut_used[y * gx_tiles + x]);
            s_zc[c] = (uint16_t)(m >= 0xFFFF0000u ? 0xFFFFu : (m + 0xFFFFu) >> 16);
        }
    } else if (zcut_used) {
        for (uint32_t c0 = threadIdx.x; c0 < cgx * cgy; c0 += 256 * 4) {
            uint32_t v[4][4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t c = c0 + u * 256, cx = c % cgx, cy = c / cgx;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t x = cx * 2u + (k & 1), y = cy * 2u + (k >> 1);
                    v[u][k] = (c < cgx * cgy && x < gx_tiles && y < gy_tiles) ? zcut_used[y * gx_tiles + x] : 0u;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t c = c0 + u * 256;
                const uint32_t m = max(max(v[u][0], v[u][1]), max(v[u][2], v[u][3]));
                if (c < cgx * cgy) s_zc[c] = (uint16_t)(m >= 0xFFFF0000u ? 0xFFFFu : (m + 0xFFFFu) >> 16);
            }
        }
    }
    SCAT_T(1);          // scan done, cut cells requested
    __syncthreads();
    SCAT_T(2);          // cut cells in LDS
    const float scale = (float)nb / (float)s_C[ZH_BINS];
    if (blockIdx.x == 0 && threadIdx.x == 0) *zbins_out = s_fl[0] == 0xFFFFu ? 0xFFFFFFFFu : (s_fl[0] | (s_fl[1] << 16));
    uint32_t kB = ZCUT_NONE;
    if (layer) {      // the layer's far end: the key where the running sum reaches layer_frac of the samples (every lane computes the same)
        const float u = layer_frac * (float)s_C[ZH_BINS];
        uint32_t i = 0;
#pragma unroll
        for (uint32_t st = ZH_BINS / 2; st; st >>= 1) if ((float)s_C[i + st] <= u) i += st;
        const long long k0 = zh_bin_start(i, zh_klo, zh_shift), k1 = zh_bin_start(i + 1u, zh_klo, zh_shift);
        const float c0 = (float)s_C[i], c1 = (float)s_C[i + 1u];
        const long long k = k0 + (long long)(fminf(fmaxf((u - c0) / (c1 - c0), 0.0f), 1.0f) * (float)(k1 - k0));
        kB = k < 1 ? 1u : (k < (long long)ZH_KEY_TOP ? (uint32_t)k : ZH_KEY_TOP);
        for (uint32_t t = blockIdx.x * 256u + threadIdx.x; t < ntiles_img; t += gridDim.x * 256u) zcut_fill[t] = kB;
    }

    uint32_t dg[BK_ITEMS_T], lr[BK_ITEMS_T];
#pragma unroll
    for (int r = 0; r < BK_ITEMS_T; r++) {
        dg[r] = 0u; lr[r] = 0u;
        if (key[r] != 0xFFFFFFFFu) {
            // bucket = floor(nb * CDF(key)), the CDF linear inside a bin.  Monotone in the key: (bin, position) is, the conversion of the
            // position, its scaling by a power of two, fma, product with a positive constant and truncation are, and a bin's largest
            // value cannot exceed the next bin's start (c1 is representable: the rounded fma stays <= c1)
            uint32_t bin, pos; int wlog;
            zh_locate(key[r], zh_klo, zh_shift, bin, pos, wlog);
            const float fscale = __uint_as_float((uint32_t)(127 - wlog) << 23);      // 2^-wlog
            const uint32_t fr = pos;
            const uint32_t c0 = s_C[bin], c1 = s_C[bin + 1u];
            const float u = __builtin_fmaf((float)(c1 - c0), fminf((float)fr * fscale, 0.99999994f), (float)c0);
            const uint32_t d = (uint32_t)(u * scale);
            dg[r] = d < nb ? d : nb - 1u;
            const uint32_t cb = COARSE ? dg[r] >> BK_CSHIFT : dg[r];      // the bucket this workgroup ranks and reserves by
            const uint32_t sh = (cb & 1u) * 16u;
            lr[r] = (atomicAdd(&cnt[cb >> 1], 1u << sh) >> sh) & 0xFFFFu;
        }
    }
    // what travels with the element: coalesced, requested here so that the loads pass under the atomics' round trip below
    uint2 rc[BK_ITEMS_T]; uint32_t tl[BK_ITEMS_T];
#pragma unroll
    for (int r = 0; r < BK_ITEMS_T; r++) {      // (culled Gaussians have rect = tiles = 0)
        const uint32_t i = base + r * 256 + threadIdx.x;
        rc[r] = make_uint2(0u, 0u); tl[r] = 0u;
        if (i < n) { rc[r] = rect[i]; tl[r] = tiles[i]; }
    }
    SCAT_T(3);          // ranks + rect loads issued
    __syncthreads();
    SCAT_T(4);
    // one returning global atomic per non-empty bucket of this workgroup, sixteen in flight per lane (issued back to back: a loop
    // that stores each result before it asks for the next waits a full memory round trip per bucket)
    const uint32_t nres = COARSE ? nb >> BK_CSHIFT : nb;              // counters this workgroup reserves slots from
    uint32_t* gc = gcount + (size_t)xcd * (COARSE ? (uint32_t)BK_NBC_MAX : nb);
    for (uint32_t k0 = threadIdx.x; k0 < nres; k0 += 256 * 16) {      // (a lane owns counter k: a wave's atomics go to 64 consecutive words)
        uint32_t g[16];
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const uint32_t k = k0 + u * 256;
            const uint32_t c = k < nres ? (cnt[k >> 1] >> ((k & 1u) * 16u)) & 0xFFFFu : 0u;
            g[u] = 0xFFFFFFFFu;                                     // "no element of this workgroup in the bucket"
            if (c) g[u] = atomicAdd(&gc[k], c);
        }
#pragma unroll
        for (int u = 0; u < 16; u++) {
            // the sub-slab's first free slot, saturated (a position past the sub-slab's end is never stored); the even lane writes the
            // word for itself and its odd neighbour
            const uint32_t mine = g[u] < 0xFFFFu ? g[u] : 0xFFFFu;
            const uint32_t other = (uint32_t)__shfl_xor((int)mine, 1, 64);
            const uint32_t k = k0 + u * 256;
            if (!(threadIdx.x & 1u) && k < nres) cnt[k >> 1] = mine | (other << 16);
        }
    }
    SCAT_T(5);          // global atomics (thread 0's share)
    __syncthreads();
    SCAT_T(6);
    uint32_t nlate = 0;
#pragma unroll
    for (int r = 0; r < BK_ITEMS_T; r++) {
        bool skipped = true;                                                   // culled (or past the end)
        if (key[r] != 0xFFFFFFFFu) {
            uint32_t wword = (rc[r].y & 0xFFFFu) - (rc[r].x & 0xFFFFu);
            if (zcut_used) {
                const uint32_t x0 = rc[r].x & 0xFFFFu, y0 = rc[r].x >> 16, x1 = rc[r].y & 0xFFFFu, y1 = rc[r].y >> 16;
                const uint32_t kq = key[r] >> 16;
                bool late = wword != 0u && (x1 - x0) * (y1 - y0) <= 64u;       // (a large rectangle is not worth the walk: early)
                if (layer) late = late && key[r] > kB;
                else if (late) {
                    const uint32_t cx0 = x0 >> cshift, cx1 = (x1 - 1u) >> cshift, cy0 = y0 >> cshift, cy1 = (y1 - 1u) >> cshift;
                    uint32_t m = 0;
                    if (cx1 - cx0 <= 1u && cy1 - cy0 <= 1u)                    // the usual case: four independent reads, no loop
                        m = max(max((uint32_t)s_zc[cy0 * cgx + cx0], (uint32_t)s_zc[cy0 * cgx + cx1]), max((uint32_t)s_zc[cy1 * cgx + cx0], (uint32_t)s_zc[cy1 * cgx + cx1]));
                    else
                        for (uint32_t cy = cy0; cy <= cy1; cy++)
                            for (uint32_t cx = cx0; cx <= cx1; cx++) m = max(m, (uint32_t)s_zc[cy * cgx + cx]);
                    late = kq > m;                                             // (kq > the rounded-up cut  =>  key > the cut)
                }
                if (late) { wword |= LATE_BIT; nlate++; }
            }
            const uint32_t cb = COARSE ? dg[r] >> BK_CSHIFT : dg[r];
            const uint32_t pos = ((cnt[cb >> 1] >> ((cb & 1u) * 16u)) & 0xFFFFu) + lr[r];
            if (COARSE) {       // (slab = the coarse slab [nb / 32][8][ccap]; the fine bucket's low bits travel in the width word)
#ifdef GSRAST_SCAT_BATCH_STORES
                rc[r].x = wword | ((dg[r] & ((1u << BK_CSHIFT) - 1u)) << 16);      // (A/B: all stores behind the late tests, below)
                lr[r] = pos;
#else
                if (pos < ccap)
                    slab[((size_t)cb * BK_XCD + xcd) * ccap + pos] = make_uint4(key[r], base + r * 256 + threadIdx.x, wword | ((dg[r] & ((1u << BK_CSHIFT) - 1u)) << 16), tl[r]);
#endif
            } else if (pos < (uint32_t)BK_CAPX)
                slab[((size_t)dg[r] * BK_XCD + xcd) * BK_CAPX + pos] = make_uint4(key[r], base + r * 256 + threadIdx.x, wword, tl[r]);
            skipped = (wword & LATE_BIT) != 0u;
        }
        if (color_skip) {       // one 64-bit word per wave and round: the wave's 64 Gaussians are consecutive
            const unsigned long long m = __ballot(skipped);
            const uint32_t first = base + r * 256 + (threadIdx.x & ~63u);
            if (lane == 0 && first < n) color_skip[first >> 6] = m;
        }
    }
#ifdef GSRAST_SCAT_BATCH_STORES
    if (COARSE) {
        // A/B (round 6, measured and dropped): a workgroup's elements of one coarse bucket are a run of ~16 consecutive slots stored by 16 different lanes; issued
        // back to back here -- behind the late tests, not between them -- WRITE_SIZE 107 -> 96 MB, but the kernel 58 -> 70 us (the burst of stores waits where
        // the interleaved form overlaps them with the late tests' LDS reads)
#pragma unroll
        for (int r = 0; r < BK_ITEMS_T; r++)
            if (key[r] != 0xFFFFFFFFu && lr[r] < ccap)
                slab[((size_t)(dg[r] >> BK_CSHIFT) * BK_XCD + xcd) * ccap + lr[r]] = make_uint4(key[r], base + r * 256 + threadIdx.x, rc[r].x, tl[r]);
    }
#endif
    if (zcut_used) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) nlate += __shfl_xor(nlate, d, 64);
        if (lane == 0 && nlate) atomicAdd(&s_late, nlate);
        __syncthreads();
        if (threadIdx.x == 0 && s_late) atomicAdd(n_late_out, s_late);
    }
    SCAT_T(7);          // late test + slab stores
#ifdef GSRAST_SCATTER_TIMING
    if (threadIdx.x == 0) atomicMax(&g_scat[13], (unsigned long long)wall_clock64());
#endif
    {   // the map's inverse at the bucket boundaries, a few per workgroup: the key where the running sum reaches b * total / nb
        const uint32_t per = (nb + gridDim.x) / gridDim.x;      // ceil((nb + 1) / gridDim.x)
        const uint32_t b = blockIdx.x * per + threadIdx.x;
        if (threadIdx.x < per && b <= nb) {
            const float u = (float)b * ((float)s_C[ZH_BINS] / (float)nb);
            uint32_t i = 0;
#pragma unroll
            for (uint32_t st = ZH_BINS / 2; st; st >>= 1) if ((float)s_C[i + st] <= u) i += st;       // largest bin whose start is <= u
            const long long k0 = zh_bin_start(i, zh_klo, zh_shift), k1 = zh_bin_start(i + 1u, zh_klo, zh_shift);
            const float c0 = (float)s_C[i], c1 = (float)s_C[i + 1u];
            const float f = fminf(fmaxf((u - c0) / (c1 - c0), 0.0f), 1.0f);
            const long long k = k0 + (long long)(f * (float)(k1 - k0));
            bkey_out[b] = k < 0 ? 0u : (k < (long long)ZH_KEY_TOP ? (uint32_t)k : ZH_KEY_TOP);
        }
    }
}

// Second launch of the two-launch scatter: workgroup (c, x) = (coarse bucket, XCD) moves its coarse sub-slab's elements into the 32 fine sub-slabs of XCD x.
// It is the only writer of those sub-slabs and of their counters: ranks come from LDS, no global atomic; a coarse sub-slab that overflowed reports through the
// fine counter of its first bucket (> BK_CAPX: the sort kernel raises the overflow flag, the host falls back to the radix sort).
__global__ void __launch_bounds__(256)
depth_bucket_refine_kernel(const uint4* __restrict__ cslab /* [nb / 32][8][ccap] */, const uint32_t* __restrict__ gccount /* [8][BK_NBC_MAX] */, uint32_t nb,
                           uint32_t* __restrict__ gcount /* [8][nb] out */, uint4* __restrict__ slab /* [nb][8][BK_CAPX] out */, uint32_t ccap)
{
    __shared__ uint32_t cnt[1 << BK_CSHIFT];
    const uint32_t x = blockIdx.x & (BK_XCD - 1), c = blockIdx.x >> 3;      // (workgroup b runs on XCD b mod 8: the XCD whose workgroups wrote these elements)
    if (threadIdx.x < (1u << BK_CSHIFT)) cnt[threadIdx.x] = 0u;
    const uint32_t n_true = gccount[(size_t)x * BK_NBC_MAX + c];
    const uint32_t n = n_true < ccap ? n_true : ccap;
    const uint4* src = cslab + ((size_t)c * BK_XCD + x) * ccap;
    constexpr int RI = 8;            // elements per lane and pass: 2048 per pass -- one pass at the usual fill (~1500 at 3 M), two for a sub-slab near its capacity
    __syncthreads();                 // (the counters are zero)
    for (uint32_t e0 = 0; e0 < n; e0 += RI * 256) {
        uint4 el[RI];
#pragma unroll
        for (int r = 0; r < RI; r++) { const uint32_t e = e0 + r * 256 + threadIdx.x; if (e < n) el[r] = src[e]; }
#pragma unroll
        for (int r = 0; r < RI; r++) {
            const uint32_t e = e0 + r * 256 + threadIdx.x;
            if (e < n) {
                const uint32_t fb = (el[r].z >> 16) & ((1u << BK_CSHIFT) - 1u);
                const uint32_t pos = atomicAdd(&cnt[fb], 1u);
                if (pos < (uint32_t)BK_CAPX)
                    slab[((size_t)((c << BK_CSHIFT) + fb) * BK_XCD + x) * BK_CAPX + pos] = make_uint4(el[r].x, el[r].y, el[r].z & ~BK_FINE_MASK, el[r].w);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < (1u << BK_CSHIFT))
        gcount[(size_t)x * nb + (c << BK_CSHIFT) + threadIdx.x] = (n_true > ccap && threadIdx.x == 0u) ? 0xFFFFu : cnt[threadIdx.x];
}

// One WAVE per bucket, no workgroup barrier anywhere: everything a bucket needs is wave-synchronous (LDS operations of one wave
// execute in order; wave_sync() only keeps the compiler from moving them across each other).
// The sort inside a bucket is a second bucket pass in LDS, not a comparison network (a bitonic sort of 256-512 64-bit composites in
// LDS measured 10-15 us per bucket, dependent LDS round trips all the way): the bucket's depth interval is cut into BK_SUB
// sub-intervals (again monotone in z), an LDS counter per sub-interval hands out arrival ranks, a scan of the counters gives each
// sub-interval its place, and an element's final position is its sub-interval's start + the number of smaller (depth bits, index)
// composites among the one to three elements that share it.  Correct for any distribution (a sub-interval holding k elements costs
// k^2 comparisons -- scenes with massive depth ties overflow the slabs and take the radix path anyway).
__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
constexpr int BK_SUB = 256;
constexpr int BK_WAVES = 2;              // buckets per workgroup

__device__ __forceinline__ uint4 bucket_element(const uint4* __restrict__ slab, uint32_t b, const uint32_t (&start)[BK_XCD + 1], uint32_t e)
{   // element e of the bucket in arrival order = element (e - start[x]) of sub-slab x
    uint32_t x = 0, s0 = 0;
#pragma unroll
    for (int q = 1; q < BK_XCD; q++) { const bool ge = e >= start[q]; x += ge ? 1u : 0u; s0 = ge ? start[q] : s0; }
    return slab[((size_t)b * BK_XCD + x) * BK_CAPX + (e - s0)];
}
__global__ void __launch_bounds__(64 * BK_WAVES) GSRAST_BSORT_OCC
depth_bucket_sort_kernel(const uint4* __restrict__ slab, const uint32_t* __restrict__ gcount, uint32_t nb,
                         const uint32_t* __restrict__ bkey /* [nb + 1]: the buckets' key intervals (the scatter's bucket map, inverted) */,
                         uint32_t* __restrict__ border /* [nb][BK_CAP]: sorted Gaussian ids of each bucket */,
                         uint32_t* __restrict__ bwincl /* [nb][BK_CAP]: inclusive scan of their rectangle widths inside the bucket */,
                         uint4* __restrict__ binfo /* [nb]: {elements, column runs, tiles, overflow} */,
                         uint32_t* __restrict__ bwsum /* [nb]: column runs again, compact -- every emission workgroup sums the ones in front of it */,
                         // list cut (gsrast_common.h), EARLY-ONLY mode (binfo_all != null): the four arrays above receive the bucket's EARLY
                         // Gaussians only -- the late ones (bit 31 of the width word) are not sorted at all, they only count into
                         // binfo_all = {all elements, all column runs, all tiles, overflow}: what the host's counts and num_rendered are made
                         // of.  Should the cut lists turn out too short, this kernel runs again over everything (pred, plain mode)
                         uint4* __restrict__ binfo_all = nullptr,
                         const uint32_t* __restrict__ pred = nullptr /* the predicated launch among the ones behind the forward blend */,
                         // completion pass of the list cut: only the Gaussians whose bit is set are sorted (the CANDIDATES: their rectangle
                         // touches a tile whose cut list was too short); binfo = their {elements, column runs, tiles, overflow}
                         const unsigned long long* __restrict__ keep_bits = nullptr,
                         // round 5: the sampled depth histogram lives in the CONTEXT and is zeroed HERE for the context's next forward (every
                         // workgroup of the scatter in front of this kernel has read it): no memset launch in front of preprocess_fwd
                         uint32_t* __restrict__ zero_words = nullptr, uint32_t n_zero_words = 0,
                         // ... and so does the predicted cut's opacity-mass table when it is the context's (tau_cut_kernel, in front of this kernel, has read it)
                         uint4* __restrict__ zero_quads = nullptr, uint32_t n_zero_quads = 0)
{
    __shared__ unsigned long long s_grp[BK_WAVES][BK_CAP];      // composites grouped by sub-interval (arrival order inside)
    __shared__ uint16_t s_aux[BK_WAVES][BK_CAP];                // arrival ranks, later the widths in sorted order (their running sum goes through s_grp, free by then: 26 KB of LDS = six workgroups per CU)
    __shared__ uint16_t s_wid[BK_WAVES][BK_CAP];                // widths, grouped like s_grp
    __shared__ uint32_t s_cnt[BK_WAVES][BK_SUB + 1];
    if (pred && *pred == 0u) return;
    if (zero_words) for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n_zero_words; k += gridDim.x * blockDim.x) zero_words[k] = 0u;
    if (zero_quads) for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n_zero_quads; k += gridDim.x * blockDim.x) zero_quads[k] = make_uint4(0u, 0u, 0u, 0u);
    const unsigned lane = lane_id(), wave = threadIdx.x >> 6;
    const uint32_t b = blockIdx.x * BK_WAVES + wave;
    if (b >= nb) return;
    const bool early_only = binfo_all != nullptr;
    unsigned long long* grp = s_grp[wave];
    uint16_t* aux = s_aux[wave];
    uint16_t* wid = s_wid[wave];
    uint32_t* cnt = s_cnt[wave];
    // this bucket's key interval, cut into BK_SUB equal sub-intervals (an approximate interval is as good as the exact one: every
    // position is clamped into the bucket, only monotonicity matters)
    const uint32_t sub_lo = bkey[b], sub_hi = bkey[b + 1u];
    const float sub_scale = (float)BK_SUB / (float)(sub_hi > sub_lo ? sub_hi - sub_lo : 1u);
    auto sub_of = [&](uint32_t key) -> uint32_t {       // monotone in the key, clamped into [0, BK_SUB)
        if (key <= sub_lo) return 0u;
        const uint32_t d = (uint32_t)((float)(key - sub_lo) * sub_scale);
        return d < (uint32_t)BK_SUB ? d : (uint32_t)BK_SUB - 1u;
    };
    // the eight sub-slab counts: lanes 0-7 load, an 8-lane inclusive scan gives the sub-slabs' first positions in the bucket
    uint32_t c = lane < (unsigned)BK_XCD ? gcount[(size_t)lane * nb + b] : 0u;
    const uint32_t over = __ballot(c > (uint32_t)BK_CAPX) != 0ull ? 1u : 0u;
    c = c < (uint32_t)BK_CAPX ? c : (uint32_t)BK_CAPX;
    uint32_t inc = c;
#pragma unroll
    for (int d = 1; d < BK_XCD; d <<= 1) { const uint32_t t = __shfl_up(inc, d, 64); if (lane >= (unsigned)d) inc += t; }
    uint32_t start[BK_XCD + 1];
    start[0] = 0u;
#pragma unroll
    for (int x = 0; x < BK_XCD; x++) start[x + 1] = __builtin_amdgcn_readlane(inc, x);
    const uint32_t n_all = start[BK_XCD];
    for (uint32_t k = lane; k <= (uint32_t)BK_SUB; k += 64) cnt[k] = 0u;
    wave_sync();
    // 1. arrival rank inside the sub-interval; the tile counts (and, early-only, the late Gaussians' widths) are only summed
    uint32_t tsum = 0, wall = 0;
    for (uint32_t e0 = lane; e0 < n_all; e0 += 64 * 4) {
        uint4 kv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const uint32_t e = e0 + u * 64; kv[u] = e < n_all ? bucket_element(slab, b, start, e) : make_uint4(0u, 0u, 0u, 0u); }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t e = e0 + u * 64;
            if (e < n_all) {
                const bool kept = !(early_only && (kv[u].z & LATE_BIT)) && (!keep_bits || ((keep_bits[kv[u].y >> 6] >> (kv[u].y & 63u)) & 1ull));
                if (!keep_bits || kept) tsum += kv[u].w;
                wall += kv[u].z & ~LATE_BIT;
                if (kept) aux[e] = atomicAdd(&cnt[sub_of(kv[u].x)], 1u);
            }
        }
    }
    wave_sync();
    // 2. exclusive scan of the BK_SUB counters in place (lane owns four consecutive ones); cnt[BK_SUB] = elements to sort
    {
        uint32_t v[BK_SUB / 64], sum = 0;
#pragma unroll
        for (int q = 0; q < BK_SUB / 64; q++) { v[q] = cnt[lane * (BK_SUB / 64) + q]; sum += v[q]; }
        uint32_t run = wave_incl_scan(sum) - sum;
#pragma unroll
        for (int q = 0; q < BK_SUB / 64; q++) { cnt[lane * (BK_SUB / 64) + q] = run; run += v[q]; }
        if (lane == 63) cnt[BK_SUB] = run;
    }
    wave_sync();
    const uint32_t n = cnt[BK_SUB];             // (= n_all unless early-only)
    // 3. group by sub-interval (the slab is read again: L2)
    for (uint32_t e0 = lane; e0 < n_all; e0 += 64 * 4) {
        uint4 kv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const uint32_t e = e0 + u * 64; kv[u] = e < n_all ? bucket_element(slab, b, start, e) : make_uint4(0u, 0u, 0u, 0u); }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t e = e0 + u * 64;
            if (e < n_all && !(early_only && (kv[u].z & LATE_BIT)) && (!keep_bits || ((keep_bits[kv[u].y >> 6] >> (kv[u].y & 63u)) & 1ull))) {
                const uint32_t slot = cnt[sub_of(kv[u].x)] + aux[e];
                grp[slot] = ((unsigned long long)kv[u].x << 32) | kv[u].y;
                wid[slot] = (uint16_t)(kv[u].z & ~LATE_BIT);
            }
        }
    }
    wave_sync();
    // 4. final position = start of the group + smaller composites inside it; the id goes out, the width into LDS at that position
    for (uint32_t e = lane; e < n; e += 64) {
        const unsigned long long me = grp[e];
        const uint32_t d2 = sub_of((uint32_t)(me >> 32));
        const uint32_t s0 = cnt[d2], s1 = cnt[d2 + 1];
        uint32_t r = 0;
        for (uint32_t q = s0; q < s1; q++) r += grp[q] < me ? 1u : 0u;
        border[(size_t)b * BK_CAP + s0 + r] = (uint32_t)me;
        aux[s0 + r] = wid[e];                                   // (possibly clipped) rectangle width = column runs
    }
    wave_sync();
    // 5. inclusive scan of the widths in sorted order: lane t owns the E consecutive elements [t*E, t*E + E)
    const uint32_t E = (n + 63) / 64;            // <= BK_CAP / 64
    uint32_t wsum = 0;
    for (uint32_t e = 0; e < E; e++) { const uint32_t t = lane * E + e; if (t < n) wsum += aux[t]; }
    uint32_t run = wave_incl_scan(wsum) - wsum;
    const uint32_t wtot = __shfl(run + wsum, 63, 64);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { tsum += __shfl_xor(tsum, d, 64); wall += __shfl_xor(wall, d, 64); }
    uint32_t* scanbuf = reinterpret_cast<uint32_t*>(grp);        // (step 4 was the composites' last use)
    for (uint32_t e = 0; e < E; e++) { const uint32_t t = lane * E + e; if (t < n) { run += aux[t]; scanbuf[t] = run; } }
    wave_sync();
    for (uint32_t t = lane; t < n; t += 64) bwincl[(size_t)b * BK_CAP + t] = scanbuf[t];
    if (lane == 0) {
        if (early_only) { binfo[b] = make_uint4(n, wtot, n_all - n, over); binfo_all[b] = make_uint4(n_all, wall, tsum, over); }
        else binfo[b] = make_uint4(n, wtot, tsum, over);
        bwsum[b] = wtot;
    }
}

// Totals of the buckets: {num_rendered lo, Q, -, num_rendered hi} and the overflow verdict into `scalars` -- by the last workgroup of
// the bucketed run emission (below), or, when the host needs the counts BEFORE it can launch that (no capacity hint yet), by this
// one-workgroup kernel.
// host_out (optional): 16 words of pinned, device-mapped host memory.  The totals are ALSO stored there, followed by a system-scope
// fence and the call's sequence number in word 15: the host spins on that word instead of on a copy + event enqueued behind the
// kernel (a D2H copy between two kernels costs ~14 us of queue: the copy itself and the dependent-launch gaps around it).
__device__ __forceinline__ void depth_bucket_totals(const uint4* __restrict__ binfo, uint32_t nb, uint32_t* __restrict__ scalars,
                                                    uint32_t* host_out = nullptr, uint32_t host_seq = 0,
                                                    const uint4* __restrict__ binfo_e = nullptr /* list cut: the early Gaussians' totals */)
{
    __shared__ unsigned long long s_t[256];
    __shared__ uint32_t s_q[256], s_qe[256];
    uint32_t over = 0, q = 0, qe = 0;
    unsigned long long tsum = 0;
    for (uint32_t k0 = threadIdx.x; k0 < nb; k0 += 256 * 8) {       // coalesced, eight independent loads per lane and round
        uint4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const uint32_t k = k0 + u * 256; v[u] = k < nb ? binfo[k] : make_uint4(0u, 0u, 0u, 0u); }
#pragma unroll
        for (int u = 0; u < 8; u++) { q += v[u].y; tsum += v[u].z; over |= v[u].w; }
        if (binfo_e) {
#pragma unroll
            for (int u = 0; u < 8; u++) { const uint32_t k = k0 + u * 256; v[u] = k < nb ? binfo_e[k] : make_uint4(0u, 0u, 0u, 0u); }
#pragma unroll
            for (int u = 0; u < 8; u++) qe += v[u].y;
        }
    }
    s_t[threadIdx.x] = tsum; s_q[threadIdx.x] = q; s_qe[threadIdx.x] = qe;
    over = __syncthreads_or((int)over) ? 1u : 0u;
    for (int st = 128; st > 0; st >>= 1) { if ((int)threadIdx.x < st) { s_t[threadIdx.x] += s_t[threadIdx.x + st]; s_q[threadIdx.x] += s_q[threadIdx.x + st]; s_qe[threadIdx.x] += s_qe[threadIdx.x + st]; } __syncthreads(); }
    if (threadIdx.x == 0) {
        scalars[0] = (uint32_t)s_t[0]; scalars[1] = s_q[0]; scalars[3] = (uint32_t)(s_t[0] >> 32); scalars[11] = over;
        const uint32_t q_early = binfo_e ? s_qe[0] : s_q[0], n_late = binfo_e ? scalars[SC_N_LATE] : 0u;
        scalars[SC_Q_EARLY] = q_early;
        scalars[SC_EARLY_COUNTS] = (uint32_t)s_t[0]; scalars[SC_EARLY_COUNTS + 1] = q_early; scalars[SC_EARLY_COUNTS + 2] = 0u; scalars[SC_EARLY_COUNTS + 3] = (uint32_t)(s_t[0] >> 32);
        if (host_out) {
            // seven self-validating 64-bit words {value, sequence number}, each ONE relaxed system-scope atomic store: no fence.  (A
            // system-scope release in front of a flag word writes the L2's dirty lines back -- with the colour kernel dirtying lines
            // beside it this workgroup, and with it the whole kernel, lasted until that kernel was done: 35 -> 120 us at 3 M.)
            unsigned long long* h64 = reinterpret_cast<unsigned long long*>(host_out);
            const uint32_t vals[7] = { (uint32_t)s_t[0], s_q[0], over, (uint32_t)(s_t[0] >> 32), q_early, n_late, scalars[SC_ZBINS] };
#pragma unroll
            for (int k = 0; k < 7; k++)
                __hip_atomic_store(h64 + k, ((unsigned long long)host_seq << 32) | vals[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
__global__ void __launch_bounds__(256)
depth_bucket_scan_kernel(const uint4* __restrict__ binfo, uint32_t nb, uint32_t* __restrict__ scalars /* [0] R lo, [1] Q, [3] R hi, [11] overflow */)
{
    depth_bucket_totals(binfo, nb, scalars);
}

// Completion pass of the list cut (round 4; gsrast_common.h): which Gaussians touch a tile whose cut list was too short?  One lane per
// Gaussian in INDEX order (coalesced 8-byte reads of the tile rectangles), the flags of the tiles listed again in LDS as a bitmap.
// cand: bit i = Gaussian i is visible and its rectangle holds such a tile -- the completion pass sorts, lists and blends those only;
// skip: the forward's "culled or late" bits lose the candidates (a late candidate is listed after all: the backward must not take its
// rows for zero); skip2: bit i = the pass need NOT evaluate Gaussian i's colour (only the late candidates' colours are missing).
__global__ void __launch_bounds__(256)
cut_candidates_kernel(uint32_t P, const uint32_t* __restrict__ tiles, const uint2* __restrict__ rect, const unsigned char* __restrict__ need2,
                      uint32_t ntiles, uint32_t gx, unsigned long long* __restrict__ skip, unsigned long long* __restrict__ cand,
                      unsigned long long* __restrict__ skip2, const uint32_t* __restrict__ pred)
{
    __shared__ uint32_t s_need[(BUCKET_MAX_TILES + 32) / 32];
    if (pred && *pred == 0u) return;
    const uint32_t nw = (ntiles + 31u) / 32u;
    for (uint32_t w = threadIdx.x; w < nw; w += 256) {
        uint32_t m = 0;
        for (uint32_t k = 0; k < 32u; k++) { const uint32_t t = w * 32u + k; if (t < ntiles && need2[t]) m |= 1u << k; }
        s_need[w] = m;
    }
    __syncthreads();
    for (uint32_t i0 = blockIdx.x * 256u; i0 < ((P + 63u) & ~63u); i0 += gridDim.x * 256u) {
        const uint32_t i = i0 + threadIdx.x;
        bool c = false;
        if (i < P && tiles[i] != 0u) {
            const uint2 rc = rect[i];
            const uint32_t x0 = rc.x & 0xFFFFu, y0 = rc.x >> 16, x1 = rc.y & 0xFFFFu, y1 = rc.y >> 16;
            if ((x1 - x0) * (y1 - y0) > 1024u) c = true;      // (a huge rectangle is not worth the walk)
            else
                for (uint32_t y = y0; y < y1 && !c; y++)
                    for (uint32_t x = x0; x < x1; x++) { const uint32_t t = y * gx + x; if ((s_need[t >> 5] >> (t & 31u)) & 1u) { c = true; break; } }
        }
        const unsigned long long m = __ballot(c);
        if ((threadIdx.x & 63u) == 0u && i < ((P + 63u) & ~63u)) {
            const unsigned long long old = skip[i >> 6];
            cand[i >> 6] = m;
            skip2[i >> 6] = ~(old & m);
            skip[i >> 6] = old & ~m;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Instance emission (reference duplicateWithKeys, rasterizer_impl.cu:70-111), in depth order.
// order[j] = Gaussian at depth rank j; offsets = inclusive scan of tiles[order[.]].
// The 64 Gaussians of a wave own one contiguous output range; the wave walks that range 64 slots
// at a time and every lane finds the Gaussian its slot belongs to (6-step binary search over the
// wave's 64 segment starts in LDS), so the two output streams are written fully coalesced and a
// Gaussian covering thousands of tiles costs the same per instance as one covering two.
template <typename KeyT>
__global__ void __launch_bounds__(256)
emit_instances_kernel(int P, const uint32_t* __restrict__ order, const uint32_t* __restrict__ offsets,
                      const uint32_t* __restrict__ tiles, const uint2* __restrict__ rect, int gx,
                      KeyT* __restrict__ tile_keys, uint32_t* __restrict__ inst_vals)
{
    __shared__ uint32_t s_e[4][64], s_g[4][64], s_xy[4][64], s_w[4][64];
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned lane = lane_id(), wave = threadIdx.x >> 6;
    uint32_t g = 0, cnt = 0, incl, xy = 0, w = 1;
    if (j < P) {
        g = order[j];
        cnt = tiles[g];
        incl = offsets[j];
        if (cnt) {
            const uint2 rc = rect[g];
            xy = rc.x;                                  // x0 | y0 << 16
            w = (rc.y & 0xFFFFu) - (rc.x & 0xFFFFu);
        }
    } else {
        incl = P > 0 ? offsets[P - 1] : 0u;
    }
    const uint32_t e = incl - cnt;                      // exclusive start of this Gaussian's segment
    const uint32_t wstart = __shfl(e, 0, 64);
    const uint32_t wend = __shfl(incl, 63, 64);
    s_e[wave][lane] = e - wstart; s_g[wave][lane] = g; s_xy[wave][lane] = xy; s_w[wave][lane] = w;
    __syncthreads();
    const uint32_t C = wend - wstart;
    for (uint32_t o = lane; o < C; o += 64) {
        uint32_t sidx = 0;
#pragma unroll
        for (int step = 32; step >= 1; step >>= 1)
            if (s_e[wave][sidx + step] <= o) sidx += step;   // largest s with start[s] <= o (sidx + step <= 63)
        const uint32_t k = o - s_e[wave][sidx];
        const uint32_t ww = s_w[wave][sidx], pxy = s_xy[wave][sidx];
        uint32_t yy = (uint32_t)((float)k * __builtin_amdgcn_rcpf((float)ww));
        if (yy * ww > k) yy--;
        if ((yy + 1) * ww <= k) yy++;
        const uint32_t xx = k - yy * ww;
        tile_keys[wstart + o] = (KeyT)(((pxy >> 16) + yy) * (uint32_t)gx + (pxy & 0xFFFFu) + xx);
        inst_vals[wstart + o] = s_g[wave][sidx];
    }
}

// Tile ranges (reference identifyTileRanges, rasterizer_impl.cu:116-138) on 16- or 32-bit tile ids.
template <typename KeyT>
__global__ void __launch_bounds__(256)
tile_ranges_kernel(uint32_t R, const uint32_t* __restrict__ R_dev /* optional: the list may be shorter than the launch */,
                   const KeyT* __restrict__ tile_keys_sorted, uint2* __restrict__ ranges)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (R_dev) R = *R_dev;
    if (i >= R) return;
    const uint32_t cur = tile_keys_sorted[i];
    if (i == 0) ranges[cur].x = 0;
    else {
        const uint32_t prev = tile_keys_sorted[i - 1];
        if (cur != prev) { ranges[prev].y = i; ranges[cur].x = i; }
    }
    if (i == R - 1) ranges[cur].y = R;
}

// ==========================================================================================
// Run-compressed binning (default when the image has <= 256 tile rows).
//
// Sorting by tile id = sorting by (row y, column x).  LSD order: x first, then y.  The x pass does not
// need instances at all: a Gaussian's tile rectangle is w "column runs" (x, y0, h), so the first pass
// sorts Q' = sum(w) compact runs (about R/5 of them) instead of R = sum(w*h) instances.  The second pass
// sorts by y; its input is the virtual instance sequence obtained by expanding the x-sorted runs in
// order, which the scatter kernel generates on the fly, and its histogram is computed from the runs
// with a difference array.  Every instance is therefore written exactly ONCE (Gaussian id + tile id,
// 6 bytes) instead of being emitted and carried through two full radix passes (36 bytes of traffic).
// Stability of both passes keeps the depth order, so point_list and the tile ranges are again
// bit-identical to the reference's single 64-bit-key sort.
// Column runs of the depth-ordered Gaussians: run k of Gaussian g covers column x0+k, rows [y0, y0+h).
// key = x (16 bit), payload = {g, y0 | h << 16}.  Balanced and coalesced like emit_instances_kernel.
//
// Exact row clipping (CULL).  The reference bins a Gaussian into every tile of the bounding square of its
// 3-sigma circle (forward.cu:232-236, auxiliary.h:46-58); the blend then skips it at every pixel whose
// alpha is below 1/255 (forward.cu:372-374).  A tile in which EVERY pixel takes that skip changes nothing:
// not the colour, not T, not the contributor bookkeeping the backward uses (skipped entries are skipped
// again).  The pixels that do not skip lie inside the ellipse  q(d) = 0.5(a dx^2 + c dy^2) + b dx dy <= t,
// t = -skip_threshold; for a tile column (dx in [dx0, dx1]) the rows that can intersect it form ONE
// interval (a convex set cut by a strip), whose ends are either the ellipse's own top / bottom or its
// crossing with the strip's nearer edge (a root of the quadratic in dy at that dx; a negative
// discriminant means the ellipse does not reach the column).  The interval is evaluated in fp64 on the fp32 coefficients
// (so it is the exact set for the quadratic the blend evaluates, no cancellation even for needle-shaped
// Gaussians), t is raised by a bound on the blend kernel's own fp32 rounding of the power
// (1e-6 * (|a| + |b|) DX^2 + (|c| + |b|) DY^2, three roundings of terms that size), and the pixel
// interval is widened by 1e-3 pixel.  Runs whose interval is empty keep their slot with h = 0.
// Output bits are identical with and without clipping (tests/test_gpu_parity.py::test_tile_clipping_*);
// only the internal lists get shorter (on the bench scene 2x).
// sqrt of a double to ~1e-14 relative from the fp32 hardware sqrt / rcp and one Newton step in fp64
// (v_sqrt_f64 costs an order of magnitude more; 1e-14 * 4096 pixels is far inside the 1e-3 pixel slack)
__device__ __forceinline__ double sqrt_newton(double x)
{
    const float xf = (float)x;
    if (!(xf > 1e-30f) || !(xf < 1e30f)) return sqrt(x);
    const float s0f = __builtin_sqrtf(xf);
    const double s0 = (double)s0f;
    return __builtin_fma(__builtin_fma(-s0, s0, x), 0.5 * (double)__builtin_amdgcn_rcpf(s0f), s0);
}

__global__ void __launch_bounds__(256) GSRAST_EMIT_OCC
emit_column_runs_kernel(int P, const uint32_t* __restrict__ order, const uint32_t* __restrict__ woffsets /* incl. scan of widths */,
                        const float4* __restrict__ binrec, int W, int H, int cull, uint32_t capQ,
                        uint16_t* __restrict__ run_keys, uint2* __restrict__ run_vals,
                        // bucket depth sort (binfo != null): one workgroup per bucket; `order` / `woffsets` are the buckets' own slot
                        // ranges [nb][BK_CAP] (sorted ids, inclusive width scan inside the bucket), bbase the buckets' first runs
                        const uint4* __restrict__ binfo = nullptr, const uint32_t* __restrict__ bwsum = nullptr, uint32_t nbuckets = 0,
                        uint32_t* __restrict__ scalars = nullptr /* the last workgroup leaves the totals here (depth_bucket_totals) */,
                        uint32_t* __restrict__ host_out = nullptr, uint32_t host_seq = 0 /* ... and in pinned host memory (see depth_bucket_totals) */,
                        // list cut: when `woffsets / binfo / bwsum` are the EARLY set, binfo_all is the set over all Gaussians (the totals
                        // report both); pred: a launch of the predicated second binning (returns at once unless *pred != 0)
                        const uint4* __restrict__ binfo_all = nullptr, const uint32_t* __restrict__ pred = nullptr,
                        // ... whose workgroup `nbuckets` (one past the buckets) empties the work buckets the first pass filled and counts the event
                        uint32_t* __restrict__ redo_bucket_cnt = nullptr, int n_redo_cnt = 0, HintTable* __restrict__ redo_hints = nullptr,
                        unsigned long long* __restrict__ host_fallback = nullptr, uint32_t host_fb_seq = 0 /* ... and tells the host (pinned word: the
                                                              sequence number of the call and how many tiles were listed again): a context whose cuts keep failing pauses them */,
                        // COMPLETION pass (round 4): the buckets hold the CANDIDATES only; a column run is kept only if it crosses a tile that
                        // is listed again (need2[tile] != 0), and the bookkeeping workgroup also leaves the pass's totals in pass2_counts
                        const unsigned char* __restrict__ need2 = nullptr, int gx_tiles = 0, uint32_t* __restrict__ pass2_counts = nullptr,
                        // binrec == null (list cut: preprocess_fwd did not write the 32-byte binning records): the same numbers from the blend's
                        // records and the rectangles -- three gathers instead of two, for the few Gaussians that are listed
                        const float4* __restrict__ rec0 = nullptr, const float4* __restrict__ rec1 = nullptr, const uint2* __restrict__ rect = nullptr,
                        // word fork (gsrast_capi.hip: WORD FORKS): "this kernel has started", for the side stream's colour kernel
                        uint32_t* __restrict__ fork_word = nullptr, uint32_t fork_seq = 0)
{
    if (fork_word && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(fork_word, fork_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (pred && *pred == 0u) return;
    __shared__ uint32_t s_e[4][64], s_g[4][64], s_x0[4][64], s_yh[4][64];
    // per-Gaussian ellipse terms (fp64): det, 2tc, 1/c, dy_max, dx_top (b, the mean: exact in fp32);  mode 0 = keep the column, 1 = clip, 2 = empty
    // (round 6: 26.9 -> 21.8 KB of LDS -- b as a float, the bookkeeping workgroup's reduction arrays aliased onto these: the kernel is a chain
    // of memory latencies, what it needs is resident workgroups)
    __shared__ double s_det[4][64], s_t2c[4][64], s_invc[4][64], s_dymax[4][64], s_dxtop[4][64];
    __shared__ float s_b[4][64], s_mx[4][64], s_my[4][64];
    __shared__ uint32_t s_mode[4][64];
    if (redo_bucket_cnt && blockIdx.x == nbuckets) {
        for (int i = threadIdx.x; i < n_redo_cnt; i += blockDim.x) redo_bucket_cnt[i] = 0u;
        if (threadIdx.x == 0 && redo_hints) atomicAdd(&redo_hints->cut_fallbacks, 1u);
        uint32_t q2 = 1u;
        if (pass2_counts) {       // {tile counts lo, column runs, -, hi} of the candidates: what the sorts behind this emission are sized by
            unsigned long long* const s_t2 = reinterpret_cast<unsigned long long*>(&s_det[0][0]);      // [256] (this workgroup emits nothing)
            uint32_t* const s_q2 = &s_e[0][0];                                                         // [256]
            unsigned long long ts = 0; uint32_t q = 0;
            for (uint32_t k = threadIdx.x; k < nbuckets; k += 256) { const uint4 v = binfo[k]; q += v.y; ts += v.z; }
            s_t2[threadIdx.x] = ts; s_q2[threadIdx.x] = q;
            __syncthreads();
            for (int st = 128; st > 0; st >>= 1) { if ((int)threadIdx.x < st) { s_t2[threadIdx.x] += s_t2[threadIdx.x + st]; s_q2[threadIdx.x] += s_q2[threadIdx.x + st]; } __syncthreads(); }
            if (threadIdx.x == 0) { pass2_counts[0] = (uint32_t)s_t2[0]; pass2_counts[1] = s_q2[0]; pass2_counts[2] = 0u; pass2_counts[3] = (uint32_t)(s_t2[0] >> 32); q2 = s_q2[0] ? s_q2[0] : 1u; }
        }
        // (the host is told what the pass cost: the candidates' column runs -- a context whose passes keep costing too much pauses the cut)
        if (threadIdx.x == 0 && host_fallback) __hip_atomic_store(host_fallback, ((unsigned long long)host_fb_seq << 32) | (unsigned long long)q2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    const unsigned lane = lane_id(), wave = threadIdx.x >> 6;
    uint32_t nloc = 0, run0 = 0, nchunks = 1;
    uint32_t pre_g = 0, pre_w = 0, pre_wm = 0;
    if (binfo) {
        // first column run of this bucket = runs of the buckets in front of it: independent 16-byte loads of the compact totals
        __shared__ uint32_t s_run0[4];
        uint32_t part = 0;
        const uint4* w4 = reinterpret_cast<const uint4*>(bwsum);
        // workgroup 0 only sums the buckets' totals (-> scalars, and the host): it is dispatched first, so its stores to host memory
        // (slow: ~20 us) pass under the emission instead of behind it (as the last workgroup's job they made the kernel 22 us longer)
        if (scalars && blockIdx.x == 0) { depth_bucket_totals(binfo_all ? binfo_all : binfo, nbuckets, scalars, host_out, host_seq, binfo_all ? binfo : nullptr); return; }
        const uint32_t b = scalars ? blockIdx.x - 1u : blockIdx.x, n4 = (b + 3) / 4;
        order += (size_t)b * BK_CAP; woffsets += (size_t)b * BK_CAP;
        // the first chunk's ids and width scans are requested now (slots past the bucket's count hold stale values, never used):
        // their round trip passes under the prefix sum's instead of behind it
        pre_g = order[threadIdx.x]; pre_w = woffsets[threadIdx.x]; pre_wm = threadIdx.x ? woffsets[threadIdx.x - 1] : 0u;
        for (uint32_t q0 = threadIdx.x; q0 < n4; q0 += 256 * 8) {
            uint4 v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { const uint32_t q = q0 + u * 256; v[u] = q < n4 ? w4[q] : make_uint4(0u, 0u, 0u, 0u); }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const uint32_t k = 4 * (q0 + u * 256);
                part += (k < b ? v[u].x : 0u) + (k + 1 < b ? v[u].y : 0u) + (k + 2 < b ? v[u].z : 0u) + (k + 3 < b ? v[u].w : 0u);
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d, 64);
        if (lane == 0) s_run0[wave] = part;
        const uint4 bi = binfo[b];
        __syncthreads();
        run0 = s_run0[0] + s_run0[1] + s_run0[2] + s_run0[3];
        nloc = bi.x; nchunks = (nloc + 255u) / 256u;
        P = (int)nloc;
    }
    for (uint32_t chunk = 0; chunk < nchunks; chunk++) {
    if (chunk) __syncthreads();
    const int j = binfo ? (int)(chunk * 256u + threadIdx.x) : (int)(blockIdx.x * blockDim.x + threadIdx.x);
    uint32_t g = 0, w = 0, incl, x0 = 0, yh = 0;
    if (j < P) {
        const bool pre = binfo && chunk == 0;
        g = pre ? pre_g : order[j];
        incl = run0 + (pre ? pre_w : woffsets[j]);
        const uint32_t prev = j > 0 ? run0 + (pre ? pre_wm : woffsets[j - 1]) : run0;
        w = incl - prev;                         // culled Gaussians (width 0, wherever the depth sort left them) never touch binrec
        uint2 rc = make_uint2(0u, 0u);
        float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0;
        if (w != 0u) {
            if (binrec) {
                r0 = binrec[2 * (size_t)g]; r1 = binrec[2 * (size_t)g + 1];
                rc = make_uint2(__float_as_uint(r1.z), __float_as_uint(r1.w));
            } else {
                r0 = rec0[(size_t)REC_STRIDE * g]; const float4 t1 = rec1[(size_t)REC_STRIDE * g]; rc = rect[g];
                r1 = make_float4(t1.x, t1.w, 0.f, 0.f);             // (conic c, skip threshold)
            }
        }
        x0 = rc.x & 0xFFFFu;
        yh = (rc.x >> 16) | (((rc.y >> 16) - (rc.x >> 16)) << 16);
        if (cull && w != 0u) {
            const float cz = r1.x;
            const float thr = r1.y;
            const float mx = r0.x, my = r0.y, A = r0.z, B = r0.w;
            // largest |dx|, |dy| any pixel of the rectangle can see
            const float DX = fmaxf(fabsf(mx - 16.0f * (float)x0), fabsf(16.0f * (float)(rc.y & 0xFFFFu) - mx));
            const float DY = fmaxf(fabsf(my - 16.0f * (float)(rc.x >> 16)), fabsf(16.0f * (float)(rc.y >> 16) - my));
            const float M = (fabsf(A) + fabsf(B)) * DX * DX + (fabsf(cz) + fabsf(B)) * DY * DY;
            const double t = (double)(-thr + 1e-6f * M);       // thr > 0 (opacity < 1/255): t < 0, nothing survives
            const double a = (double)A, b = (double)B, c = (double)cz;
            const double det = a * c - b * b;
            uint32_t mode = 1;
            if (!(t >= 0.0)) mode = 2;                                              // also NaN
            else if (!(det > 0.0 && a > 0.0 && c > 0.0)) mode = 0;
            double dy_max = 0.0, dx_top = 0.0;
            if (mode == 1) {
                dy_max = sqrt(2.0 * t * a / det);
                dx_top = -b * dy_max / a;                       // where the ellipse reaches dy = +dy_max
                if (!(dy_max < 1e30)) mode = 0;
            }
            s_det[wave][lane] = det; s_t2c[wave][lane] = 2.0 * t * c; s_b[wave][lane] = B; s_invc[wave][lane] = 1.0 / c;
            s_dymax[wave][lane] = dy_max; s_dxtop[wave][lane] = dx_top;
            s_mx[wave][lane] = mx; s_my[wave][lane] = my; s_mode[wave][lane] = mode;
        }
    } else {
        incl = P > 0 ? run0 + woffsets[P - 1] : run0;
    }
    const uint32_t e = incl - w;
    const uint32_t wstart = __shfl(e, 0, 64);
    const uint32_t wend = __shfl(incl, 63, 64);
    s_e[wave][lane] = e - wstart; s_g[wave][lane] = g; s_x0[wave][lane] = x0; s_yh[wave][lane] = yh;
    __syncthreads();
    // capQ: the buffers may have been sized before the host knew Q (speculative launch); nothing is written past them
    const uint32_t C = wstart >= capQ ? 0u : ((wend < capQ ? wend : capQ) - wstart);
    for (uint32_t o = lane; o < C; o += 64) {
        uint32_t sidx = 0;
#pragma unroll
        for (int step = 32; step >= 1; step >>= 1)
            if (s_e[wave][sidx + step] <= o) sidx += step;
        const uint32_t k = o - s_e[wave][sidx];
        const uint32_t x = s_x0[wave][sidx] + k;
        uint32_t yhv = s_yh[wave][sidx];
        const uint32_t mode = cull ? s_mode[wave][sidx] : 0u;
        if (mode == 2u) yhv &= 0xFFFFu;
        else if (mode == 1u) {
            uint32_t y0 = yhv & 0xFFFFu, h = yhv >> 16;
            const double det = s_det[wave][sidx], t2c = s_t2c[wave][sidx], b = (double)s_b[wave][sidx], invc = s_invc[wave][sidx];
            const double dy_max = s_dymax[wave][sidx], dx_top = s_dxtop[wave][sidx];
            const double mx = (double)s_mx[wave][sidx], my = (double)s_my[wave][sidx];
            const int px1 = min((int)x * 16 + 15, W - 1);
            const double dx0 = mx - (double)px1, dx1 = mx - (double)((int)x * 16);   // dx = mean - pixel
            double hi = dy_max, lo = -dy_max;
            bool empty = false;
            {   const double dxc = fmin(fmax(dx_top, dx0), dx1);
                if (dxc != dx_top) { const double disc = __builtin_fma(-det * dxc, dxc, t2c); empty |= disc < 0.0; hi = (sqrt_newton(fmax(disc, 0.0)) - b * dxc) * invc; } }
            {   const double dxc = fmin(fmax(-dx_top, dx0), dx1);
                if (dxc != -dx_top) { const double disc = __builtin_fma(-det * dxc, dxc, t2c); empty |= disc < 0.0; lo = (-sqrt_newton(fmax(disc, 0.0)) - b * dxc) * invc; } }
            // pixel rows p with my - p in [lo, hi]
            const double plo = ceil(my - hi - 1e-3), phi = floor(my + 1e-3 - lo);
            const int tlo = max((int)y0, (int)fmax(plo, 0.0) >> 4);
            const int thi = min((int)(y0 + h) - 1, (int)fmin(phi, (double)(H - 1)) >> 4);
            if (empty || phi < 0.0 || plo > (double)(H - 1) || thi < tlo) h = 0;
            else { y0 = (uint32_t)tlo; h = (uint32_t)(thi - tlo + 1); }
            yhv = y0 | (h << 16);
        }
        if (need2 && (yhv >> 16) != 0u) {        // completion pass: a run that crosses no tile listed again lists nothing
            const uint32_t y0 = yhv & 0xFFFFu, h = yhv >> 16;
            bool any = false;
            for (uint32_t yy = y0; yy < y0 + h; yy++) any |= need2[yy * (uint32_t)gx_tiles + x] != 0;
            if (!any) yhv = y0;
        }
        run_keys[wstart + o] = (uint16_t)x;
        run_vals[wstart + o] = make_uint2(s_g[wave][sidx], yhv);
    }
    } // chunk
}

// Per-block histogram over tile rows of the instances of RUNS_PER_BLOCK consecutive (x-sorted) runs:
// +1 at y0, -1 at y0+h in an LDS difference array, prefix sum, one column of the digit-major table.
__global__ void __launch_bounds__(256)
run_hist_rows_kernel(const uint2* __restrict__ run_vals, uint32_t Q, const uint32_t* __restrict__ Q_dev,
                     uint32_t* __restrict__ block_hist, uint32_t nblk, uint32_t nrows /* tile rows: only these table rows exist */,
                     const uint32_t* __restrict__ pred = nullptr)
{
    __shared__ int diff[257];
    if (pred && *pred == 0u) return;
    Q = dev_count(Q, Q_dev);
    for (int k = threadIdx.x; k < 257; k += 256) diff[k] = 0;
    __syncthreads();
    const uint32_t r0 = blockIdx.x * RUNS_PER_BLOCK;
    for (uint32_t k = threadIdx.x; k < (uint32_t)RUNS_PER_BLOCK; k += 256) {
        const uint32_t r = r0 + k;
        if (r < Q) {
            const uint32_t yh = run_vals[r].y;
            const uint32_t y0 = yh & 0xFFFFu, h = yh >> 16;
            atomicAdd(&diff[y0], 1);
            atomicAdd(&diff[y0 + h], -1);
        }
    }
    __syncthreads();
    const uint32_t mine = (uint32_t)diff[threadIdx.x];
    uint32_t tot;
    const uint32_t incl = block_excl_scan(mine, &tot) + mine;      // wrap-around arithmetic of the signed deltas is exact
    if (threadIdx.x < nrows) block_hist[(size_t)threadIdx.x * nblk + blockIdx.x] = incl;
}

// Tile ranges without reading the sorted list back (and without a tile id stream next to it):
// tile (x, y) starts at  [instances in rows < y] + [instances in row y from runs of columns < x].
// Runs are sorted by column, so "runs of columns < x" is a prefix [0, F(x)) of the run array: whole
// blocks of it are in the scanned row histogram, the partial block is counted here.  One workgroup per
// tile column, one lane per tile row.  Empty tiles get {0, 0} as in the reference
// (rasterizer_impl.cu:311 + identifyTileRanges :116-138).
__device__ __forceinline__ uint32_t first_run_of_column(const uint16_t* __restrict__ run_keys, uint32_t Q, uint32_t x)
{
    uint32_t lo = 0, hi = Q;                                   // first index whose key >= x; 256-ary search, block-wide
    while (hi > lo) {
        const uint32_t step = (hi - lo + 255u) / 256u;
        const uint32_t idx = lo + threadIdx.x * step;
        const bool below = idx < hi && (uint32_t)run_keys[idx] < x;
        const uint32_t c = (uint32_t)__syncthreads_count(below);   // samples are monotone: the first c are below
        if (c == 0) { hi = lo; break; }
        const uint32_t nlo = lo + (c - 1u) * step + 1u;
        const uint32_t nhi = lo + c * step;
        lo = nlo; hi = nhi < hi ? nhi : hi;
    }
    return lo;
}
__device__ __forceinline__ uint32_t row_instances_before_run(const uint2* __restrict__ run_vals, uint32_t Q, uint32_t F,
                                                              const uint32_t* __restrict__ hist_scanned, uint32_t nblk,
                                                              uint32_t nrows, uint32_t row_total, int* diff /* LDS [257] */)
{
    if (F >= Q) return row_total;                              // uniform
    const uint32_t b0 = F / RUNS_PER_BLOCK, r0 = b0 * RUNS_PER_BLOCK;
    __syncthreads();
    for (int k = threadIdx.x; k < 257; k += 256) diff[k] = 0;
    __syncthreads();
    for (uint32_t r = r0 + threadIdx.x; r < F; r += 256) {
        const uint32_t yh = run_vals[r].y;
        atomicAdd(&diff[yh & 0xFFFFu], 1);
        atomicAdd(&diff[(yh & 0xFFFFu) + (yh >> 16)], -1);
    }
    __syncthreads();
    const uint32_t mine = (uint32_t)diff[threadIdx.x];
    uint32_t tot;
    const uint32_t incl = block_excl_scan(mine, &tot) + mine;
    return (threadIdx.x < nrows ? hist_scanned[(size_t)threadIdx.x * nblk + b0] : 0u) + incl;
}
__device__ __forceinline__ void
tile_ranges_from_runs_body(uint32_t column, const uint16_t* __restrict__ run_keys, const uint2* __restrict__ run_vals, uint32_t Q,
                             const uint32_t* __restrict__ counts_dev /* optional {R lo, Q, -, R hi}: speculative launch */,
                             uint32_t capR, int gx, int gy, const uint32_t* __restrict__ hist_scanned,
                             const uint32_t* __restrict__ digit_total, uint32_t nblk, uint2* __restrict__ ranges,
                             uint32_t* __restrict__ bucket_cnt /* forward launch order: [8][64] counts (zeroed), or null */,
                             uint16_t* __restrict__ bucket_list /* [8][64][Tg] */,
                             const HintTable* __restrict__ hints /* or null: what each tile of this camera pose consumed the last time (gsrast_common.h) */,
                             const uint32_t* __restrict__ hint_sel /* [2]: slot, valid */,
                             // completion pass of the list cut: only the tiles listed again (need2 != 0) get a range -- into the point list's
                             // second half (list_offset) -- and a place in the launch order; the others keep the first pass's
                             const unsigned char* __restrict__ need2 = nullptr, uint32_t list_offset = 0,
                             const uint32_t* __restrict__ zcut_pred = nullptr /* or this call's cut depths when they are PREDICTED ones (gsrast_common.h) */)
{
    __shared__ int diff[257];
    __shared__ uint32_t lcnt[XCD_GROUPS * WORK_BUCKETS], lbase[XCD_GROUPS * WORK_BUCKETS];
    const uint32_t x = column, y = threadIdx.x;
    bool overflow = false;
    if (counts_dev) {
        // The launch was sized for capacities (Q = capQ).  If the real counts do not fit, the lists are truncated:
        // publish empty ranges (the blend kernels then touch nothing) -- the host sees the same counts and redoes the
        // binning with exact sizes.
        overflow = counts_dev[1] > Q || counts_dev[3] != 0u || counts_dev[0] > capR;      // uniform
        if (!overflow) Q = counts_dev[1];
    }
    uint32_t work = 0;
    const bool mine = y < (uint32_t)gy && (!need2 || need2[y * (uint32_t)gx + x] != 0);
    if (overflow) {
        if (mine) ranges[y * (uint32_t)gx + x] = make_uint2(0u, 0u);
    } else {
    const uint32_t nrows = (uint32_t)gy;
    const uint32_t row_total = y < nrows ? digit_total[y] : 0u;
    uint32_t all;
    const uint32_t row_base = block_excl_scan(row_total, &all);
    const uint32_t F0 = first_run_of_column(run_keys, Q, x);
    const uint32_t F1 = first_run_of_column(run_keys, Q, x + 1u);
    const uint32_t before0 = row_instances_before_run(run_vals, Q, F0, hist_scanned, nblk, nrows, row_total, diff);
    const uint32_t before1 = row_instances_before_run(run_vals, Q, F1, hist_scanned, nblk, nrows, row_total, diff);
    if (mine)
        ranges[y * (uint32_t)gx + x] = before1 > before0 ? make_uint2(list_offset + row_base + before0, list_offset + row_base + before1) : make_uint2(0u, 0u);
    work = before1 > before0 ? before1 - before0 : 0u;
    // forward launch order: the prefix this tile consumed the last time this pose was rendered, if the context knows (never more than the list)
    if (hints && hint_sel[1] && work && y < (uint32_t)gy) {
        const uint32_t h = hint_work(hints, 0u)[(size_t)hint_sel[2] * ((uint32_t)gx * (uint32_t)gy) + y * (uint32_t)gx + x];      // ([2]: the pose's own slot, or the near pose's it borrows from)
        work = h < work ? (h ? h : 1u) : work;
    } else if (zcut_pred && work && y < (uint32_t)gy && zcut_pred[y * (uint32_t)gx + x] != ZCUT_NONE) {
        // Under PREDICTED cut depths nobody knows what the tile consumed last time, but a tile that got a cut is one whose pixels saturate
        // well in front of it (the prediction is conservative: the 3 M cube's interior tiles consume a seventh of their cut lists), while a
        // tile without one -- a silhouette -- may walk its whole list: two octaves down, so that the tiles without a cut of the same
        // length start first (they shared the half-octave work buckets with thousands of light tiles: blend_fwd 0.35 ms against 0.27 in
        // the exact order)
        work = work >> 2 ? work >> 2 : 1u;
    }
    }
    if (bucket_cnt) {
        // forward launch order: append this column's tiles to the work buckets of their XCD group (tile row mod 8); one global
        // atomic per non-empty (group, bucket) and column
        for (uint32_t i = y; i < (uint32_t)(XCD_GROUPS * WORK_BUCKETS); i += blockDim.x) lcnt[i] = 0;
        __syncthreads();
        const uint32_t idx = (y % (uint32_t)XCD_GROUPS) * WORK_BUCKETS + work_bucket(work);
        uint32_t slot = 0;
        if (mine) slot = atomicAdd(&lcnt[idx], 1u);
        __syncthreads();
        for (uint32_t i = y; i < (uint32_t)(XCD_GROUPS * WORK_BUCKETS); i += blockDim.x) lbase[i] = lcnt[i] ? atomicAdd(&bucket_cnt[i], lcnt[i]) : 0u;
        __syncthreads();
        const size_t Tg = (size_t)gx * (size_t)((gy + XCD_GROUPS - 1) / XCD_GROUPS);
        if (mine) bucket_list[(size_t)idx * Tg + lbase[idx] + slot] = (uint16_t)(y * (uint32_t)gx + x);
    }
}

// Second (final) pass: instances of the block's runs, generated in order, ranked by tile row with the
// same ballot-match / LDS-exchange scheme as radix_scatter_kernel, written once.
// The run an instance belongs to is found without searching: every run sets ONE bit (at its first
// instance) in a per-sub-batch LDS bitmap, and "number of run starts at or before slot i" is a word
// prefix count plus a popcount of the slot's word.
__device__ __forceinline__ void
run_scatter_rows_body(uint32_t block /* = blockIdx.x of a launch of its own */, const uint2* __restrict__ run_vals /* sorted by column */,
                        uint32_t Q, const uint32_t* __restrict__ Q_dev, uint32_t capR, int ybits, uint32_t nrows, const uint32_t* __restrict__ hist_scanned,
                        const uint32_t* __restrict__ digit_total, uint32_t nblk,
                        uint32_t* __restrict__ point_list,
                        uint32_t* __restrict__ total_out /* number of instances written (<= R with row clipping) */)
{
    __shared__ uint32_t s_start[RUNS_PER_BLOCK + 1]; // block-local first instance of every run (+ total at the end)
    __shared__ uint2 s_val[RUNS_PER_BLOCK];
    __shared__ uint32_t s_nruns;
    __shared__ uint32_t cnt[4][256];
    __shared__ uint32_t dstart[256], gbase[256], running[256];
    __shared__ unsigned long long bits[RUN_CHUNK / 64];
    __shared__ uint32_t wpre[RUN_CHUNK / 64];         // run starts before each bitmap word (relative to the sub-batch)
    __shared__ uint8_t xk[RUN_CHUNK];                 // tile row
    __shared__ uint32_t xv[RUN_CHUNK];
    const unsigned lane = lane_id(), wave = threadIdx.x >> 6;
    const uint32_t t = threadIdx.x;
    const uint32_t r0 = block * RUNS_PER_BLOCK;
    Q = dev_count(Q, Q_dev);
    if (r0 >= Q && block != 0) return;                    // block past the end of a capacity-sized launch (uniform)
    const uint32_t nslots = r0 >= Q ? 0u : ((Q - r0) < (uint32_t)RUNS_PER_BLOCK ? (Q - r0) : (uint32_t)RUNS_PER_BLOCK);
    uint32_t nruns;                                   // non-empty runs of this block (row clipping leaves h = 0 slots)
    {   // stage the non-empty runs, compacted; exclusive prefix of their heights = first instance of each run
        constexpr int RPT = RUNS_PER_BLOCK / RS_THREADS;     // consecutive runs per lane
        static_assert(RUNS_PER_BLOCK <= 1024, "packed scan: 20 bits of instances, 11 bits of runs");
        uint2 v[RPT];
        uint32_t packed = 0;                          // heights in the low 20 bits (<= 1024 * 256), non-empty count above
#pragma unroll
        for (int j = 0; j < RPT; j++) {
            const uint32_t k = RPT * t + j;
            v[j] = make_uint2(0u, 0u);
            if (k < nslots) v[j] = run_vals[r0 + k];
            const uint32_t h = v[j].y >> 16;
            packed += h + (h ? (1u << 20) : 0u);
        }
        uint32_t tot;
        uint32_t ex = block_excl_scan(packed, &tot);
#pragma unroll
        for (int j = 0; j < RPT; j++) {
            const uint32_t h = v[j].y >> 16;
            if (h) { const uint32_t k = ex >> 20; s_val[k] = v[j]; s_start[k] = ex & 0xFFFFFu; ex += h + (1u << 20); }
        }
        if (t == 0) { s_start[RUNS_PER_BLOCK] = tot & 0xFFFFFu; s_nruns = tot >> 20; }
        uint32_t gtot;
        const uint32_t dbase = block_excl_scan(t < nrows ? digit_total[t] : 0u, &gtot);   // instances in lower tile rows, globally
        if (block == 0 && t == 0) *total_out = gtot;
        running[t] = dbase + (t < nrows ? hist_scanned[(size_t)t * nblk + block] : 0u);
    }
    __syncthreads();
    const uint32_t ninst = s_start[RUNS_PER_BLOCK];
    nruns = s_nruns;
    volatile uint32_t* wc = cnt[wave];
    uint32_t first_run = 0;                           // runs starting before the current sub-batch (uniform)
    for (uint32_t sb = 0; sb < ninst; sb += RUN_CHUNK) {
        for (int k = t; k < 1024; k += RS_THREADS) (&cnt[0][0])[k] = 0;
        if (t < RUN_CHUNK / 64) bits[t] = 0ull;
        __syncthreads();
        const uint32_t nsub = (ninst - sb) < (uint32_t)RUN_CHUNK ? (ninst - sb) : (uint32_t)RUN_CHUNK;
        for (uint32_t k = t; k < nruns; k += RS_THREADS) {
            const uint32_t st = s_start[k];
            if (st >= sb && st < sb + RUN_CHUNK)
                atomicOr(&bits[(st - sb) >> 6], 1ull << ((st - sb) & 63u));
        }
        __syncthreads();
        if (t < 64) {   // wave 0: exclusive prefix of the words' popcounts
            const uint32_t pc = t < RUN_CHUNK / 64 ? (uint32_t)__popcll(bits[t]) : 0u;
            const uint32_t in = wave_incl_scan(pc);
            if (t < RUN_CHUNK / 64) wpre[t] = in - pc;
        }
        __syncthreads();
        // a short sub-batch is split evenly over the four waves: share = slots per wave (multiple of 64)
        const uint32_t share = ((nsub + 255u) >> 8) << 6;
        uint32_t key[RUN_ITEMS], val[RUN_ITEMS], rk[RUN_ITEMS];     // key = tile row
#pragma unroll
        for (int r = 0; r < RUN_ITEMS; r++) {
            if ((uint32_t)r * 64u >= share) break;
            const uint32_t ls = wave * share + r * 64 + lane;               // slot inside the sub-batch (= instance order)
            const bool valid = ls < nsub;
            uint32_t y = 0, g = 0;
            if (valid) {
                const uint32_t wd = ls >> 6;
                // runs that start at or before this slot, minus one = index of the run covering it
                const uint32_t upto = wpre[wd] + (uint32_t)__popcll(bits[wd] & (~0ull >> (63u - (ls & 63u))));
                const uint32_t k = first_run + upto - 1u;
                const uint2 v = s_val[k];
                y = (v.y & 0xFFFFu) + (sb + ls - s_start[k]);
                g = v.x;
            }
            key[r] = y; val[r] = g;
            const uint64_t m = wave_match8(y, valid, ybits);
            const uint32_t prev = wc[y];
            const uint32_t below = (uint32_t)__popcll(m & lanemask_lt());
            if (valid && below == 0) wc[y] = prev + (uint32_t)__popcll(m);
            rk[r] = prev + below;
        }
        first_run += wpre[RUN_CHUNK / 64 - 1] + (uint32_t)__popcll(bits[RUN_CHUNK / 64 - 1]);
        __syncthreads();
        {
            const uint32_t c0 = cnt[0][t], c1 = cnt[1][t], c2 = cnt[2][t], c3 = cnt[3][t];
            uint32_t tot;
            const uint32_t start = block_excl_scan(c0 + c1 + c2 + c3, &tot);
            dstart[t] = start;
            gbase[t] = running[t];
            running[t] += c0 + c1 + c2 + c3;
            cnt[0][t] = start; cnt[1][t] = start + c0; cnt[2][t] = start + c0 + c1; cnt[3][t] = start + c0 + c1 + c2;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < RUN_ITEMS; r++) {
            if ((uint32_t)r * 64u >= share) break;
            const uint32_t ls = wave * share + r * 64 + lane;
            if (ls < nsub) {
                const uint32_t slot = cnt[wave][key[r]] + rk[r];
                xk[slot] = (uint8_t)key[r]; xv[slot] = val[r];
            }
        }
        __syncthreads();
        // write-out: slot order = (row, instance order)
#pragma unroll
        for (int r = 0; r < RUN_ITEMS; r++) {
            const uint32_t slot = r * RS_THREADS + t;
            if ((uint32_t)r * RS_THREADS >= nsub) break;
            if (slot < nsub) {
                const uint32_t y = xk[slot];
                const uint32_t pos = gbase[y] + (slot - dstart[y]);
                if (pos < capR) point_list[pos] = xv[slot];        // capR: see tile_ranges_from_runs_kernel
            }
        }
        __syncthreads();
    }
}

// The row pass and the tile ranges in ONE launch: both only read the scanned row histogram and the column-sorted runs, neither reads what
// the other writes.  Workgroups [0, nblk) expand and rank the instances of their runs, workgroups [nblk, nblk + gx) compute the ranges
// of their tile column (one launch fewer per forward, and one fewer among the predicated launches of the list cut).
__global__ void __launch_bounds__(RS_THREADS) GSRAST_ROWS_OCC
rows_and_ranges_kernel(const uint16_t* __restrict__ run_keys, const uint2* __restrict__ run_vals, uint32_t Q, const uint32_t* __restrict__ counts_dev,
                       uint32_t capR, int ybits, int gx, int gy, const uint32_t* __restrict__ hist_scanned, const uint32_t* __restrict__ digit_total,
                       uint32_t nblk, uint32_t* __restrict__ point_list, uint32_t* __restrict__ total_out, uint2* __restrict__ ranges,
                       uint32_t* __restrict__ bucket_cnt, uint16_t* __restrict__ bucket_list, const HintTable* __restrict__ hints,
                       const uint32_t* __restrict__ hint_sel, const uint32_t* __restrict__ pred /* or null: predicated launch */,
                       const unsigned char* __restrict__ need2 = nullptr, uint32_t list_offset = 0 /* completion pass: see tile_ranges_from_runs_body (point_list is then the second half) */,
                       const uint32_t* __restrict__ zcut_pred = nullptr)
{
    static_assert(RS_THREADS == 256, "one lane per tile row in the ranges part");
    if (pred && *pred == 0u) return;
    if (blockIdx.x < nblk) run_scatter_rows_body(blockIdx.x, run_vals, Q, counts_dev ? counts_dev + 1 : nullptr, capR, ybits, (uint32_t)gy, hist_scanned, digit_total, nblk, point_list, total_out);
    else tile_ranges_from_runs_body(blockIdx.x - nblk, run_keys, run_vals, Q, counts_dev, capR, gx, gy, hist_scanned, digit_total, nblk, ranges, bucket_cnt, bucket_list, hints, hint_sel, need2, list_offset, zcut_pred);
}

// Launch order of the blend kernels: tiles sorted by descending work (bucketed counting sort,
// one workgroup).  work(tile) = ranges[tile].y - ranges[tile].x for the forward, tile_max[tile] for the
// backward.  A tile is processed by one wave(-group) from start to end, so without this the heaviest
// tiles, wherever they fall in the grid, set the kernel's tail.
constexpr int ORDER_BUCKETS = 64;
__global__ void __launch_bounds__(1024)
tile_order_kernel(uint32_t T, const uint2* __restrict__ ranges, const uint32_t* __restrict__ tile_max,
                  uint32_t* __restrict__ order)
{
    __shared__ uint32_t cnt[ORDER_BUCKETS];
    __shared__ uint32_t start[ORDER_BUCKETS];
    __shared__ uint32_t wmax;
    if (threadIdx.x < ORDER_BUCKETS) cnt[threadIdx.x] = 0;
    if (threadIdx.x == 0) wmax = 1;
    __syncthreads();
    uint32_t m = 0;
    for (uint32_t t = threadIdx.x; t < T; t += blockDim.x) {
        const uint32_t w = tile_max ? tile_max[t] : (ranges[t].y - ranges[t].x);
        m = w > m ? w : m;
    }
    atomicMax(&wmax, m);
    __syncthreads();
    const float scale = (float)(ORDER_BUCKETS - 1) / (float)wmax;
    for (uint32_t t = threadIdx.x; t < T; t += blockDim.x) {
        const uint32_t w = tile_max ? tile_max[t] : (ranges[t].y - ranges[t].x);
        const int b = ORDER_BUCKETS - 1 - (int)((float)w * scale);     // bucket 0 = heaviest
        atomicAdd(&cnt[b], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t a = 0; for (int b = 0; b < ORDER_BUCKETS; b++) { start[b] = a; a += cnt[b]; } }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < T; t += blockDim.x) {
        const uint32_t w = tile_max ? tile_max[t] : (ranges[t].y - ranges[t].x);
        const int b = ORDER_BUCKETS - 1 - (int)((float)w * scale);
        order[atomicAdd(&start[b], 1u)] = t;
    }
}

} // namespace gsrast
