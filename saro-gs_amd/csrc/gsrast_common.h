// gsrast_common.h -- shared constants, HBM state layout and wave64 helpers (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace gsrast {

constexpr int TILE_X = 16;          // reference config.h:16-17 -- pinned by key parity
constexpr int TILE_Y = 16;
constexpr int TILE_PIX = TILE_X * TILE_Y;
constexpr int WAVE = 64;            // CDNA4 wavefront

// ---------------------------------------------------------------------------------------------
// HBM layout of the three state buffers.  Every array starts on a 256-byte boundary.
// Geometry (per Gaussian), replaces GeometryState (reference rasterizer_impl.h:30-45):
//   (depth)              view-space z lives in rec1.z (also the low 32 key bits); there is no separate array
//   rec0     float4[P]   {mean2D.x, mean2D.y, conic.x, conic.y}     \  gathered by the blend
//   rec1     float4[P]   {conic.z, opacity, depth, skip_threshold}   > kernels, 48 B / instance
//   rec2     float4[P]   {r, g, b, 0}   (colour kernel, side stream) /
//   cov3D    f32[6P]     upper triangle (only when built from scale/rotation)
//   clamped  u8[P]       bit c set <=> colour channel c was clamped at 0
//   tiles    u32[P]      tiles touched
//   rect     uint2[P]    {min.x | min.y<<16, max.x | max.y<<16} tile rectangle
//   binrec   float4[2P]  {x, y, conic.x, conic.y} {conic.z, skip_threshold, rect.x bits, rect.y bits}: one 32-byte gather per
//                        Gaussian for the column-run emission (only written for visible Gaussians, only read for those)
//   keyA/B   u32[P] x2   depth-sort ping-pong keys        valA/B u32[P] x2   ping-pong values
//   offsets  u32[P]      inclusive scan of tiles[] in depth order
//   hist     u32[256*nblk(P)]   radix block histograms    scan_tmp u32[...]  scan partials
//   scalars  u32[64]     [0] = num_rendered ... [8] = significant bits of the depth keys (adaptive pass count of the depth sort)
//   keyC/valC u32[P] x2  third buffer pair of the depth sort, sort_minmax u32[2 nblk]: per-block key minimum / maximum
//   shdA/B   float4[P] x2, shdC f32[P]   d(colour)/d(view direction) {dx[3], dy[3], dz[3]}, 36 B: written by the forward's colour kernel
//                        while the SH block is in LDS (round 3; sh_dir_derivs_kernel in the backward for a forward_only state), so that the
//                        per-Gaussian backward reads 36 B instead of 12*M
//   zhist    u32[8][ZH_BINS]  sampled histogram (one copy per XCD) of the visible depth keys (preprocess_fwd; zeroed by a memset in front of it):
//                        the bucket depth sort (gsrast_binning.h, NB ~ P/256 buckets of CAP slots) cuts the depth axis into
//                        buckets of EQUAL POPULATION by it; bk_count u32[8][NB], bk_slab uint4[NB][8][CAP/8] (arrival order), bk_order / bk_wincl
//                        u32[NB][CAP] (sorted ids, inclusive width scan inside the bucket), bk_info uint4[NB], bk_base u32[NB] (compact column-run totals)
//   grec     f32[16P]    backward only: per-Gaussian gradient record {dL/dmean2D.x, .y, dL/dconic a, b, c, dL/dopacity, dL/dr, dg, db,
//                        7 unused}, one 64-byte line per Gaussian.  gsrast_backward zero-fills it, the blend backward adds the nine sums
//                        of a (tile, Gaussian) pair with nine adjacent lanes, the per-Gaussian backward reads it with three 16-byte loads
#ifndef GSRAST_REC_STRIDE
#define GSRAST_REC_STRIDE 1
#endif
constexpr int REC_STRIDE = GSRAST_REC_STRIDE;      // float4 steps between two Gaussians' rec0 (rec1, rec2): 1 = three arrays, 4 = interleaved 64-byte records
// (4 measured at 3 M: the blend forward stages one line per instance instead of three, 0.246 -> 0.238 ms, but the geometry kernel's 16-byte
// stores at a 64-byte stride cost it 0.068 -> 0.118 ms: 1 stays)
struct GeomLayout {
    size_t rec0, rec1, rec2, cov3D, clamped, tiles, rect, binrec, keyA, keyB, valA, valB, offsets, woffsets,
        hist, scan_tmp, scalars, grec, keyC, valC, sort_minmax, shdA, shdB, shdC, zhist, bk_key, bk_count, bk_ccount /* u32[8][256]: counters of the two-launch scatter's coarse level, right behind bk_count (zeroed with it) */, bk_slab, bk_order, bk_wincl, bk_info, bk_base,
        bk_order_e, bk_wincl_e, bk_info_e, bk_base_e /* the same four over the EARLY Gaussians only, compact (list cut, below) */,
        color_skip /* u64[ceil(P / 64)]: bit i = Gaussian i is culled or late (list cut): the colour kernel skips it */,
        cand_bits /* u64[ceil(P / 64)]: bit i = Gaussian i touches a tile the completion pass lists again */,
        skip2 /* u64[ceil(P / 64)]: bit i = the completion pass need NOT evaluate Gaussian i's colour (it is not a late candidate) */,
        untouched /* u8[P]: byte i != 0 = NO pixel consumed Gaussian i (bytes, not bits: the blend clears them with plain stores -- 2.4 M bit-clearing atomics per 3 M forward cost it 9 us) (round 5, stateless: set by preprocess_fwd, cleared by the forward
                     blend for the list prefix each tile consumed): the backward's rows of such a Gaussian are zero -- written beside the
                     blend backward, skipped by the per-Gaussian backward, whatever the pose table knows */, total;
};
constexpr int GREC = 16;            // floats per gradient record
constexpr size_t BUCKET_SORT_MIN_P = 32768;     // below this the depth sort is one or two self-scanned radix passes anyway
#ifndef GSRAST_BK_TARGET
#define GSRAST_BK_TARGET 256      // Gaussians per depth bucket, roughly
#endif
#ifndef GSRAST_BK_CAP
#define GSRAST_BK_CAP 1024        // slots per bucket
#endif
// Equalised depth buckets (round 4).  Buckets of equal DEPTH width overflow as soon as the depth distribution has a peak (the bench cube
// seen along a diagonal at 3 M: a triangular density, 2 x the mean in the middle -- 366 Gaussians per bucket on average, 128 slots per
// (bucket, XCD) sub-slab; a trained scene's walls and floors are worse): the forward then fell back to the radix passes, with an
// exponential back-off, and lost the list cut with them.  preprocess_fwd therefore also leaves a SAMPLED HISTOGRAM of the visible depth
// keys (ZH_BINS bins over a key range the context expects from its previous forwards, below -- positive floats order
// like their bits, so a bin is an interval of log-depth; ~130 k sampled Gaussians, one fire-and-forget atomic each), and the scatter
// maps a key through the histogram's running sum, linearly inside a bin: bucket = floor(nb * CDF(key)) -- monotone in the key whatever
// the histogram holds (a stale range or a torn count is only a poorer balance; an overflow still takes the radix path), equal
// population per bucket when it is right.
constexpr int ZH_BINS = 1024, ZH_COPIES = 8 /* one histogram per XCD (workgroup b runs on XCD b mod 8): same-address atomics from eight L2s serialise at the memory side -- 187 k of them on ~300 bins of ONE table cost the geometry kernel 65 us */;
// The table's bins: ZH_MID bins of 2^shift key steps from kmid -- the range the context expects --, and ZH_TAIL bins of sixteen times
// that width on either side, so that a view whose depth range lies a whole range away from the expected one still spreads over
// bins (a key beyond the tails is clamped into the end bin: many of those pile up in one bucket).
constexpr int ZH_TAIL = 64, ZH_MID = ZH_BINS - 2 * ZH_TAIL, ZH_TAIL_LOG = 4;
constexpr uint32_t ZH_KLO_DEFAULT = 0x3E400000u;   // just below the bits of 0.2f (a visible Gaussian has z > 0.2, auxiliary.h:154)
constexpr int ZH_SHIFT_DEFAULT = 21;              // 896 bins of 2^21 key steps (4 per octave) reach past every finite float: the first forward of a context
constexpr uint32_t ZH_KEY_TOP = 0x7F800000u;      // no finite positive float lies above
// bin of a key, its position inside the bin and log2 of the bin's width -- monotone in the key: (bin, pos) increases lexicographically
// (32-bit arithmetic throughout: keys and kmid are below 2^31, shift <= ZH_SHIFT_DEFAULT, so ZH_MID << shift < 2^31 and the tails'
// bin width 2^(shift + 4) <= 2^25)
__host__ __device__ inline void zh_locate(uint32_t key, uint32_t kmid, int shift, uint32_t& bin, uint32_t& pos, int& wlog)
{
    const int d = (int)key - (int)kmid;
    const int wt = shift + ZH_TAIL_LOG;
    if (d < 0) {
        wlog = wt;
        const uint32_t m = (uint32_t)(-d - 1), t = m >> wt, w1 = (1u << wt) - 1u;
        if (t >= (uint32_t)ZH_TAIL) { bin = 0u; pos = 0u; }
        else { bin = (uint32_t)(ZH_TAIL - 1) - t; pos = w1 - (m & w1); }
    } else if (((uint32_t)d >> shift) < (uint32_t)ZH_MID) {
        wlog = shift;
        bin = (uint32_t)ZH_TAIL + ((uint32_t)d >> shift); pos = (uint32_t)d & ((1u << shift) - 1u);
    } else {
        wlog = wt;
        const uint32_t e = (uint32_t)d - ((uint32_t)ZH_MID << shift), t = e >> wt, w1 = (1u << wt) - 1u;
        if (t >= (uint32_t)ZH_TAIL) { bin = (uint32_t)ZH_BINS - 1u; pos = w1; }
        else { bin = (uint32_t)(ZH_TAIL + ZH_MID) + t; pos = e & w1; }
    }
}
// first key of a bin (bin == ZH_BINS: one past the table), as a signed 64-bit number: the low tail may reach below zero
__host__ __device__ inline long long zh_bin_start(uint32_t bin, uint32_t kmid, int shift)
{
    if (bin < (uint32_t)ZH_TAIL) return (long long)kmid - ((long long)((uint32_t)ZH_TAIL - bin) << (shift + ZH_TAIL_LOG));
    if (bin < (uint32_t)(ZH_TAIL + ZH_MID)) return (long long)kmid + ((long long)(bin - (uint32_t)ZH_TAIL) << shift);
    return (long long)kmid + ((long long)ZH_MID << shift) + ((long long)(bin - (uint32_t)(ZH_TAIL + ZH_MID)) << (shift + ZH_TAIL_LOG));
}
static inline uint32_t depth_buckets_host(size_t P)     // buckets of the depth sort: ~P / 256, a power of two in [256, 8192] (gsrast_binning.h)
{
    uint32_t nb = 256;
    while ((size_t)nb * GSRAST_BK_TARGET < P && nb < 8192u) nb <<= 1;
    return nb;
}
// Binning (per instance), replaces BinningState (rasterizer_impl.h:56-65):
//   valA/B u32[C] x2  Gaussian id ping-pong (valA at offset 0 = the final point_list)
//   keyA/B u32[C] x2  tile id ping-pong (16-bit ids up to 65 536 tiles), hist u32[256*nblk(C)], scan_tmp
//   C = capacity >= R the buffer was requested for (1.25 x the previous call's R, or R itself)
struct BinLayout {
    size_t keyA, keyB, valA, valB, hist, scan_tmp, total;
};
// Image (per pixel / per tile), replaces ImageState (rasterizer_impl.h:47-54):
//   final_T f32[N], n_contrib u32[N], ranges uint2[T], tile_max u32[T] (deepest list position any
//   pixel of the tile consumed -- bounds the backward traversal)
//   order_fwd / order_bwd u32[T]: tiles sorted by descending work (longest-processing-time-first
//   launch order for the blend kernels; a tile is one indivisible unit of work per wave(-group))
//   bucket_cnt u32[2][64], bucket_list u16[2][64][T] (T <= 65535): the same launch order without a sorting kernel --
//   tiles are appended to one of 64 work buckets (half-octaves of the work estimate) by the kernel that produces the
//   estimate (tile ranges -> forward order, forward blend's deepest consumed entry -> backward order); block b of a blend
//   kernel finds its tile from the prefix sums of the 64 counts.  [0] = forward, [1] = backward.
struct ImgLayout {
    size_t final_T, n_contrib, ranges, tile_max, order_fwd, order_bwd, bucket_cnt, bucket_list, zcut_used /* u32[T]: this call's snapshot of the pose's cut depths (list cut, below) */,
        tile_flags /* u8[T]: 1 = the tile's cut list was too short: the completion pass lists and blends it again (list cut, below) */,
        tau_hist /* u32[TAU_COPIES][T][TAU_BINS]: opacity mass per tile and depth bin (predicted cut, below) */, total;
};
constexpr int WORK_BUCKETS = 64;
constexpr size_t BUCKET_MAX_TILES = 65535;     // tile ids are stored as u16
// The work buckets are kept per XCD group: workgroup b of a launch runs on XCD b mod 8 (round-robin dispatch), and each of the
// eight XCDs has its own L2.  A tile belongs to group (tile row mod 8), so the horizontal neighbours of a tile -- which share
// most of its Gaussians -- are blended through the same L2, heaviest first within the group.
constexpr int XCD_GROUPS = 8;
static inline size_t xcd_group_tiles_host(size_t gx, size_t gy) { return gx * ((gy + XCD_GROUPS - 1) / XCD_GROUPS); }

// Launch-order hints of the forward blend (round 3).  A tile's work in the forward is the list prefix its pixels CONSUME, unknown
// before it has been blended; its list LENGTH, the only estimate a stateless call has, is a poor one in an occluded scene (3 M cube:
// 2800 listed, 144 consumed on average -- but the tiles along the lower image edge consume all of their 1800 entries): heaviest-first
// by length started the truly heavy tiles late and the kernel ended in a long tail at half occupancy.  A gsrast_context therefore
// keeps, per DEVICE and per camera POSE (key = hash of the view and projection matrices and the image size), what every tile of that
// pose consumed the last time it was rendered -- u16 [HINT_SLOTS][T] in device memory, least-recently-used replacement -- and the next
// forward of the same pose orders its launch by it (3 M: blend_fwd 0.325 -> 0.241 ms; 1 M: 0.220 -> 0.184; a surface-like scene,
// where length is a good estimate already: unchanged).  Everything happens on the device (the matrices are device pointers): block 0
// of preprocess_fwd_kernel looks the pose up and claims a slot, tile_ranges_from_runs_kernel reads the slot's estimates, the blend
// writes the new ones.  Only the ORDER of a launch depends on it, never a result; a pose seen for the first time (or
// options.fwd_order_hint = 0) falls back to the list lengths.  SaRO-GS renders fixed camera rigs (Neural3D: ~20 cameras x 300 frames,
// D-NeRF: every pose again each epoch), and its evaluation of a Neural3D scene is ONE pose for 300 frames.
constexpr int HINT_SLOTS = 256;    // camera poses per context and device (D-NeRF rigs have 100-200 training views; 6 B per tile and pose: 12.5 MB at 1080p)
struct HintTable {
    uint32_t key[HINT_SLOTS][2];   // 0, 0 = free
    uint32_t stamp[HINT_SLOTS];    // value of `clock` when the slot was last used
    uint32_t clock, cut_fallbacks /* forwards whose cut lists turned out too short and were binned and blended again (diagnostic) */, pad[2];
    float cam[HINT_SLOTS][8];      // the slot's camera: position (3), viewing direction (3), -, - -- a pose the table does not know BORROWS the
                                   // estimates and cut depths of a slot whose camera is close (a camera path: the previous frame); the cut is verified either way
    // followed by uint16_t work[HINT_SLOTS][T], then (4-byte aligned) uint32_t zcut[HINT_SLOTS][T]
};
// A forward's own choice -- {slot, 1 = "the slot held estimates of this pose when the forward began" / 2 = "estimates borrowed from
// a near pose", the slot they are read from} -- lives in ITS geometry buffer (scalars[HINT_SEL], [HINT_SEL + 1], [HINT_SEL + 2]), so two forwards of one context in flight on two streams do not read each other's slot.
constexpr int HINT_SEL = 16;
// ... and what the forward blend PUBLISHES into a newly claimed slot once the call's lists exist (round 5: preprocess_fwd's block 0 used to
// write the key while the kernel's other lookup blocks read the table): scalars[HINT_PUB] = 1 "publish", [+1, +2] the pose's key,
// [+3 .. +8] its camera (position, viewing direction)
constexpr int HINT_PUB = 32;
constexpr int SC_GREC_SPARSE = 46;    // scalars: 1 = only the gradient records of the Gaussians whose untouched bit is CLEAR were zeroed (grec_zero_touched_kernel): every
                                      // other record holds whatever the buffer held -- its readers take it for zero (record_is_stale)
constexpr int SC_TOUCH_VALID = 45;    // scalars: 1 = this forward's blend kept GeomLayout::untouched (the backward trusts the bits)
constexpr int SC_PREFILTER = 44;      // scalars: set by preprocess_fwd when a Gaussian is culled although the caller passed prefiltered = 1
__host__ __device__ inline uint16_t* hint_work(HintTable* h, uint32_t) { return reinterpret_cast<uint16_t*>(h + 1); }
__host__ __device__ inline const uint16_t* hint_work(const HintTable* h, uint32_t) { return reinterpret_cast<const uint16_t*>(h + 1); }
static inline size_t hint_zcut_offset(size_t T) { return (sizeof(HintTable) + (size_t)HINT_SLOTS * T * 2 + 255) & ~(size_t)255; }
__host__ __device__ inline uint32_t* hint_zcut(HintTable* h, uint32_t T) { return reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(h) + ((sizeof(HintTable) + (size_t)HINT_SLOTS * T * 2 + 255) & ~(size_t)255)); }
__host__ __device__ inline const uint32_t* hint_zcut(const HintTable* h, uint32_t T) { return reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(h) + ((sizeof(HintTable) + (size_t)HINT_SLOTS * T * 2 + 255) & ~(size_t)255)); }
static inline size_t hint_table_bytes(size_t T) { return hint_zcut_offset(T) + (size_t)HINT_SLOTS * T * 4 + 256; }

// LIST CUT (round 3): the binning of an occluded scene works on instances nobody consumes (3 M cube: 23 M listed, 1.2 M consumed).  A
// pose that has been rendered before also remembers, per tile, a CUT DEPTH -- the depth of the list entry 1.5 x as deep (+ 32) as the
// deepest one any pixel of the tile consumed, or "none" (ZCUT_NONE) for a tile whose pixels did not all saturate.  The next forward of
// that pose snapshots the cut depths into its own image buffer (preprocess_fwd), and the bucket scatter marks a Gaussian LATE when it
// lies behind the cut depth of every tile of its rectangle.  Late Gaussians stay in the depth order but get no column runs: emission,
// run sort and row expansion only see the EARLY ones.  This is a speculation, and it is verified: the forward blend treats a tile's
// list as ending where its depth exceeds the tile's cut depth (up to there the cut list IS the full list), and a tile with a cut whose
// pixels are not all saturated at that point raises the call's `undone` counter.  Behind the blend the forward has enqueued the whole
// binning + blend once more over ALL Gaussians, every kernel predicated on that counter: nothing runs when the speculation held (the
// steady state of a repeated pose), everything is redone from the full lists when it did not.  Results never depend on the table.
constexpr uint32_t ZCUT_NONE = 0xFFFFFFFFu;
constexpr uint32_t CUT_MAX_CELLS = 3072;      // the bucket scatter keeps the cut depths in LDS as maxima over cells of 2 x 2 tiles -- 4 x 4 or 8 x 8 for images
                                              // with more than this many cells (4K: 4 x 4); images beyond 8 x 8 cells render without the list cut
static inline int cut_cell_shift(size_t gx, size_t gy)      // 1, 2, 3, or 0 = no list cut (also: the bucket sort packs widths into 15 bits)
{
    if (gx >= 32768) return 0;
    for (int cs = 1; cs <= 3; cs++) { const size_t c = (size_t)1 << cs; if (((gx + c - 1) >> cs) * ((gy + c - 1) >> cs) <= CUT_MAX_CELLS) return cs; }
    return 0;
}
// PREDICTED CUT (round 5): a cut depth for a pose the context has never rendered -- or whose remembered cut keeps failing because the scene
// is another one at every visit (SaRO-GS: opacity, means and scales are functions of the timestamp) -- from THIS call's own Gaussians.
// preprocess_fwd adds a visible Gaussian's OPTICAL-DEPTH mass -- the integral over the image plane of -ln(1 - alpha(x)) = 2 pi sqrt(det cov2D)
// Li2(opacity), in pixels^2, capped at what it can lay on one tile -- to the bin (tile of its centre, depth bin); mass / 256 estimates the
// tile's MEAN optical depth from below (series cut short, wide Gaussians capped, one wave in two sampled and doubled).  tau_cut_kernel walks
// every tile's bins front to back: the far edge of the bin where the running mean reaches tau_req (default 10; T < 1e-4 needs 9.2 at EVERY
// pixel -- the far edge and the 3 x 3 maximum are the slack) is the tile's predicted cut -- the deepest over the 3 x 3 tiles
// around it, none if one of those has none: silhouette tiles, whose pixels do not all saturate, get the full list straight away.  The
// prediction is verified like a remembered cut (the blend flags a tile whose cut list ended before it saturated, the completion pass lists
// it again), so it can only cost time; every reported pass raises tau_req.  Depth bins: TAU_BINS equal steps of the depth KEY (float bits:
// piecewise linear in log depth) over the key range [lo, hi] the context has learned from its forwards (the occupied range padded by an
// eighth, gsrast_capi.hip: learn_depth_range) -- the depth histogram's own bins are powers of two of key steps and cover up to twice
// that range, too coarse here: a tile's cut lands on a bin's FAR edge, so a bin's width is what the prediction gives away.  The last
// bin also takes everything behind hi and has no far edge (no cut there).
#ifndef GSRAST_TAU_BINS
#define GSRAST_TAU_BINS 32
#endif
constexpr int TAU_BINS = GSRAST_TAU_BINS, TAU_COPIES = 4;      // (copies: workgroup b adds into copy b mod 4)
struct TauBins { uint32_t lo; float scale /* bins per key step */; float inv_scale; uint32_t wave_mask /* wave w of preprocess_fwd adds its Gaussians iff (hash(w) & mask) == 0, each (mask + 1) times its mass */; };
__host__ __device__ inline uint32_t tau_bin_of(uint32_t key, const TauBins& tb)
{
    if (key <= tb.lo) return 0u;
    const float f = (float)(key - tb.lo) * tb.scale;
    return f >= (float)(TAU_BINS - 1) ? (uint32_t)(TAU_BINS - 1) : (uint32_t)f;
}
// a key that is certainly not in front of bin b's far edge, whatever the float rounding in tau_bin_of did (two key steps of slack per bin index)
__host__ __device__ inline uint32_t tau_bin_far_edge(uint32_t b, const TauBins& tb)
{
    const float e = (float)(b + 1u) * tb.inv_scale;
    const unsigned long long k = (unsigned long long)tb.lo + (unsigned long long)e + 2ull * (b + 2u);
    return k >= (unsigned long long)ZH_KEY_TOP ? ZCUT_NONE : (uint32_t)k;
}
constexpr uint32_t LATE_BIT = 0x80000000u;      // in the width word of a bucket-slab element
// words of GeomLayout::scalars used by the list cut
constexpr int SC_ZBINS = 7 /* first | last << 16 occupied bin of the sampled depth histogram (0xFFFFFFFF: no sample) */;
constexpr int GATE_WORDS = 80;      // 64 first-level counters of finished workgroups + 1 second-level (ImgLayout::bucket_cnt's tail)
constexpr int SC_GATE_COUNT = 28 /* workgroups of the cut forward's blend that have finished (its last one is the completion pass's gate) */;
constexpr int SC_PASS2 = 24 /* {instances (tile counts) lo, column runs, -, hi} of the completion pass's candidates */;
constexpr int SC_Q_EARLY = 4, SC_N_LATE = 5, SC_UNDONE = 6, SC_EARLY_COUNTS = 20 /* {R lo, Q early, -, R hi} */, SC_REDO_PRED = SC_UNDONE /* the predicate of the second binning + blend: some tile's cut list was too short */;

constexpr int RS_THREADS = 256;     // radix sort: 4 waves
constexpr uint32_t RS_SELF_SCAN_BLOCKS = 64;   // sorts of at most this many blocks skip the row-scan launch (radix_scatter_kernel)
constexpr int RS_ITEMS = 16;        // keys per lane (32 measured slower: profiles/)
constexpr int RS_CHUNK = RS_THREADS * RS_ITEMS;
constexpr int SC_CHUNK = 4096;      // scan: elements per block (256 threads x 16)

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
static inline size_t rs_blocks(size_t n) { return (n + RS_CHUNK - 1) / RS_CHUNK; }
#ifndef GSRAST_DEPTH_ITEMS
#define GSRAST_DEPTH_ITEMS 8      // elements per lane in the depth sort (P elements): measured 16 -> 112 us, 8 -> 95, 4 -> 99, 2 -> 122
#endif
#ifndef GSRAST_RUN_SORT_ITEMS
#define GSRAST_RUN_SORT_ITEMS 8   // ... in the sort of the column runs (Q elements): 4 / 8 / 16 measured equal on full lists (8.7 M runs); under the list cut (1.1 M runs, one round of workgroups either way) 8 makes the pass 5 us shorter than 16
#endif
static inline size_t rs_blocks_n(size_t n, int items) { return (n + (size_t)RS_THREADS * items - 1) / ((size_t)RS_THREADS * items); }
static inline size_t scan_tmp_elems(size_t n)
{ // partial sums for a multi-level scan of n elements
    size_t tot = 0;
    while (n > SC_CHUNK) { n = (n + SC_CHUNK - 1) / SC_CHUNK; tot += 2 * (align256(n * 4) / 4); }   // x2: room for the sums of an auxiliary array
    // the same scratch also receives the 256 digit totals of a radix pass (radix_rowscan_kernel): never fewer than 512 words.
    // (It used to be tot + 64: for n <= 4096 the digit totals ran 768 bytes past it -- over `scalars` and the end of the
    // buffer, usually into the allocator's padding, occasionally into an unmapped page.)
    return tot + 512;
}

static inline GeomLayout geom_layout(size_t P)
{
    GeomLayout L; size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o = align256(o + bytes); return r; };
    size_t Pp = P ? P : 1;
    if (REC_STRIDE == 4) { L.rec0 = take(Pp * 64); L.rec1 = L.rec0 + 16; L.rec2 = L.rec0 + 32; }      // one 64-byte record per Gaussian: {rec0, rec1, rec2, -}
    else { L.rec0 = take(Pp * 16); L.rec1 = take(Pp * 16); L.rec2 = take(Pp * 16); }
    L.cov3D = take(Pp * 24); L.clamped = take(Pp); L.tiles = take(Pp * 4); L.rect = take(Pp * 8); L.binrec = take(Pp * 32);
    L.keyA = take(Pp * 4); L.keyB = take(Pp * 4); L.valA = take(Pp * 4); L.valB = take(Pp * 4);
    L.offsets = take(Pp * 4); L.woffsets = take(Pp * 4);
    size_t hist_n = 256 * rs_blocks_n(Pp, GSRAST_DEPTH_ITEMS);
    L.hist = take(hist_n * 4);
    size_t st = scan_tmp_elems(hist_n) > scan_tmp_elems(Pp) ? scan_tmp_elems(hist_n) : scan_tmp_elems(Pp);
    L.scan_tmp = take(st * 4);
    L.scalars = take(256);
    L.grec = take(Pp * GREC * 4);
    L.keyC = take(Pp * 4); L.valC = take(Pp * 4);                               // third buffer pair of the adaptive depth sort
    L.sort_minmax = take(2 * rs_blocks_n(Pp, GSRAST_DEPTH_ITEMS) * 4);
    L.shdA = take(Pp * 16); L.shdB = take(Pp * 16); L.shdC = take(Pp * 4);
    // bucket depth sort (gsrast_binning.h): sampled depth histogram of preprocess_fwd and its running sum, bucket counters, slabs
    L.zhist = take(ZH_COPIES * ZH_BINS * 4);
    {
        const size_t nb = Pp >= BUCKET_SORT_MIN_P ? depth_buckets_host(Pp) : 0;
        L.bk_count = take(nb * 8 * 4);                       // [8 XCDs][nb]
        L.bk_ccount = take(8 * 256 * 4);                     // [8 XCDs][256 coarse buckets] (two-launch scatter, gsrast_binning.h); contiguous with bk_count: nb * 32 bytes is a multiple of 256
        L.bk_slab = take(nb * GSRAST_BK_CAP * 16);           // [nb][8][CAP / 8] {key, id, width, tiles}
        L.bk_order = take(nb * GSRAST_BK_CAP * 4); L.bk_wincl = take(nb * GSRAST_BK_CAP * 4);
        L.bk_info = take(nb * 16); L.bk_base = take(nb * 4); L.bk_key = take((nb + 1) * 4);
        L.bk_order_e = take(nb * GSRAST_BK_CAP * 4); L.bk_wincl_e = take(nb * GSRAST_BK_CAP * 4); L.bk_info_e = take(nb * 16); L.bk_base_e = take(nb * 4);
    }
    L.color_skip = take(((Pp + 63) / 64) * 8 + 256);
    L.cand_bits = take(((Pp + 63) / 64) * 8 + 256);
    L.skip2 = take(((Pp + 63) / 64) * 8 + 256);
    L.untouched = take(Pp + 256);            // (one BYTE per Gaussian: the blend marks with plain stores)
    L.total = o + 256;
    return L;
}
static inline BinLayout bin_layout(size_t R)
{
    BinLayout L; size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o = align256(o + bytes); return r; };
    size_t Rp = R ? R : 1;
    // valA first: the sorted Gaussian ids (point_list) always end at offset 0, so the backward needs
    // neither the instance count nor the capacity the buffer was sized for
    L.valA = take(Rp * 4); L.valB = take(Rp * 4); L.keyA = take(Rp * 4); L.keyB = take(Rp * 4);
    size_t hist_n = 256 * rs_blocks(Rp);
    L.hist = take(hist_n * 4);
    L.scan_tmp = take(scan_tmp_elems(hist_n) * 4);
    L.total = o + 256;
    return L;
}
// Run-compressed binning (gsrast_binning.h): capR = instance capacity, capQ = column-run capacity.
//   point_list u32[capR] (offset 0), run key / value ping-pong u16[capQ] x2 / uint2[capQ] x2,
//   histogram of the x pass (256 x blocks(capQ)), of the row pass (256 x capQ/RUNS_PER_BLOCK), scan partials
#ifndef GSRAST_RUNS_PER_BLOCK
#define GSRAST_RUNS_PER_BLOCK 512
#endif
#ifndef GSRAST_RUN_ITEMS
#define GSRAST_RUN_ITEMS 8
#endif
constexpr int RUNS_PER_BLOCK = GSRAST_RUNS_PER_BLOCK;  // runs per workgroup of the row pass
constexpr int RUN_ITEMS = GSRAST_RUN_ITEMS;            // instances per lane per sub-batch of the row pass
constexpr int RUN_CHUNK = RUN_ITEMS * 256;
struct RunBinLayout {
    size_t point_list, rkeyA, rkeyB, rvalA, rvalB, hist_x, hist_y, scan_tmp, total;
};
static inline RunBinLayout runbin_layout(size_t capR, size_t capQ)
{
    RunBinLayout L; size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o = align256(o + bytes); return r; };
    if (!capR) capR = 1;
    if (!capQ) capQ = 1;
    L.point_list = take(2 * capR * 4);      // (second half: the lists of the tiles the completion pass of the list cut lists again, below)
    L.rkeyA = take(capQ * 2); L.rkeyB = take(capQ * 2); L.rvalA = take(capQ * 8); L.rvalB = take(capQ * 8);
    L.hist_x = take(256 * rs_blocks_n(capQ, GSRAST_RUN_SORT_ITEMS) * 4);
    L.hist_y = take(256 * ((capQ + RUNS_PER_BLOCK - 1) / RUNS_PER_BLOCK) * 4);
    L.scan_tmp = take((scan_tmp_elems(capQ) + 512) * 4);
    L.total = o + 256;
    return L;
}

static inline ImgLayout img_layout(size_t W, size_t H)
{
    ImgLayout L; size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o = align256(o + bytes); return r; };
    size_t N = W * H ? W * H : 1;
    size_t T = ((W + TILE_X - 1) / TILE_X) * ((H + TILE_Y - 1) / TILE_Y);
    if (!T) T = 1;
    L.final_T = take(N * 4); L.n_contrib = take(N * 4); L.ranges = take(T * 8); L.tile_max = take(T * 4);
    L.order_fwd = take(T * 4); L.order_bwd = take(T * 4);
    const size_t Tg = xcd_group_tiles_host((W + TILE_X - 1) / TILE_X, (H + TILE_Y - 1) / TILE_Y);      // list capacity of one (group, bucket)
    L.bucket_cnt = take(((XCD_GROUPS + 1) * WORK_BUCKETS + GATE_WORDS) * 4);   // forward: per XCD group; backward: one global set; + the completion pass's gate counters (gsrast_blend.h, GateArgs), zeroed with them
    L.bucket_list = take(T <= BUCKET_MAX_TILES ? (XCD_GROUPS * WORK_BUCKETS * (Tg ? Tg : 1) + WORK_BUCKETS * T) * 2 : 0);
    L.zcut_used = take(T * 4);
    L.tile_flags = take(T);
    L.tau_hist = take((size_t)TAU_COPIES * T * TAU_BINS * 4);
    L.total = o + 256;
    return L;
}
// key bits / radix passes needed to sort tile ids < T
static inline int tile_bits(size_t T)
{
    int bits = 1;
    while (((size_t)1 << bits) < T) bits++;
    return bits;
}
static inline int tile_passes(size_t T)
{
    int p = (tile_bits(T) + 7) / 8;
    return p ? p : 1;
}

// ---------------------------------------------------------------------------------------------
// Device helpers
#if defined(__HIPCC__)

// work estimate -> bucket, 0 = heaviest: two buckets per octave, bucket 63 = no work at all
__device__ __forceinline__ uint32_t work_bucket(uint32_t w)
{
    if (w == 0u) return WORK_BUCKETS - 1;
    const int e = 31 - __builtin_clz(w);
    const uint32_t key = 2u * (uint32_t)e + (e >= 1 ? ((w >> (e - 1)) & 1u) : 0u);      // 0 .. 63
    return key >= (uint32_t)WORK_BUCKETS - 2u ? 0u : (uint32_t)WORK_BUCKETS - 2u - key;
}
__device__ __forceinline__ uint32_t xcd_group_tiles(uint32_t gx, uint32_t ntiles) { return gx * ((ntiles / gx + XCD_GROUPS - 1) / XCD_GROUPS); }
__device__ __forceinline__ unsigned lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// exp(): three interchangeable implementations (option "exp_mode").
//  0: fixed sequence of exactly rounded fp32 operations -- bit-identical to oracle det_expf().
//  1: OCML expf (<= 1 ulp).   2: v_exp_f32(x * log2e) (fast, ~1e-6 relative for |x| < 6).
// BOUNDED: the caller guarantees -80 <= x <= 0 (the blend kernels only evaluate exp for
// skip_threshold <= power <= 0 and preprocess clamps the threshold at -80), so the two range guards of
// the full function are no-ops and are left out -- same bits, four instructions less per call.
template <int MODE, bool BOUNDED = false>
__device__ __forceinline__ float gs_exp(float x)
{
    if constexpr (MODE == 0) {
        if constexpr (!BOUNDED) {
            if (x < -80.0f) return 0.0f;
            x = x > 80.0f ? 80.0f : x;
        }
        float n = __builtin_rintf(x * 1.44269504088896341f);
        float r = __builtin_fmaf(n, -0.693359375f, x);
        r = __builtin_fmaf(n, 2.12194440e-4f, r);
        float p = 1.9875691500e-4f;
        p = __builtin_fmaf(p, r, 1.3981999507e-3f);
        p = __builtin_fmaf(p, r, 8.3334519073e-3f);
        p = __builtin_fmaf(p, r, 4.1665795894e-2f);
        p = __builtin_fmaf(p, r, 1.6666665459e-1f);
        p = __builtin_fmaf(p, r, 5.0000001201e-1f);
        float y = __builtin_fmaf(p, r * r, r) + 1.0f;
        return __builtin_amdgcn_ldexpf(y, (int)n);
    } else if constexpr (MODE == 1) {
        return expf(x);
    } else {
        return __builtin_amdgcn_exp2f(x * 1.44269504088896341f);
    }
}

// power of the 2D Gaussian at offset d -- same fixed contraction as oracle power_f().
__device__ __forceinline__ float gs_power(float cx, float cy, float cz, float dx, float dy)
{
    float q = __builtin_fmaf(cz * dy, dy, (cx * dx) * dx);
    return __builtin_fmaf(-0.5f, q, -((cy * dx) * dy));
}

// DPP add: v + dpp_move(v); lanes without a source (or masked off) add 0.
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ float dpp_add(float v)
{
    int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, BANK_MASK, true);
    return v + __int_as_float(t);
}
// Sum over the 64 lanes of a wave; the total is valid in lane 63 only.  Six fused v_add_f32_dpp.
// The two broadcast steps run UNMASKED (row_mask 0xf): row r then adds row r-1's lane 15 (rows 1-3),
// and rows 2-3 add lane 31 = r0+r1, so lane 63 = (r2+r3) + (r0+r1); the other lanes hold partial
// sums nobody reads.  Unmasked, hipcc folds each step into one v_add_f32_dpp; with the textbook row
// masks it emits v_mov 0 + v_mov_dpp + v_add (3 instructions) per step.
__device__ __forceinline__ float wave_sum_to_lane63(float v)
{
    v = dpp_add<0x111>(v);              // row_shr:1
    v = dpp_add<0x112>(v);              // row_shr:2
    v = dpp_add<0x114>(v);              // row_shr:4
    v = dpp_add<0x118>(v);              // row_shr:8   -> lane 15 of each row holds the row sum
    v = dpp_add<0x142>(v);              // row_bcast:15 -> lane 31 = r0+r1, lane 63 = r2+r3
    v = dpp_add<0x143>(v);              // row_bcast:31 -> lane 63 = total
    return v;
}

// Transposing ("reduce-scatter") wave sum of EIGHT values at once.  Step by step each lane keeps half
// of its values and hands the other half to its partner, so the work halves every step:
//   xor 1 (quad_perm) 8 -> 4 values, xor 2 (quad_perm) 4 -> 2, rotate 4 / rotate 8 inside the 16-lane
//   row 2 -> 1, then the four rows are combined with two ds_bpermute exchanges (LDS crossbar, no VALU).
// Afterwards EVERY lane l holds the wave-wide total of v[l & 7].  28 VALU instructions instead of the
// 8 x 6 = 48 of eight independent DPP reductions.
template <int CTRL>
__device__ __forceinline__ float dpp_pair_sum(float v)     // v + v[partner], full rows, no bound control needed
{
    int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false);
    return v + __int_as_float(t);
}
__device__ __forceinline__ float wave_sum8_transposed(const float (&v)[8], unsigned lane)
{
    const bool b0 = lane & 1u, b1 = lane & 2u, b2 = lane & 4u;
    float w[4], x[2];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float a = dpp_pair_sum<0xB1>(v[2 * i]);          // quad_perm [1,0,3,2]: lanes l, l^1
        const float b = dpp_pair_sum<0xB1>(v[2 * i + 1]);
        w[i] = b0 ? b : a;
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const float a = dpp_pair_sum<0x4E>(w[2 * i]);          // quad_perm [2,3,0,1]: lanes l, l^2
        const float b = dpp_pair_sum<0x4E>(w[2 * i + 1]);
        x[i] = b1 ? b : a;
    }
    const float a = dpp_pair_sum<0x124>(x[0]);                 // row_ror:4 (same l & 3 => same value kind)
    const float b = dpp_pair_sum<0x124>(x[1]);
    float y = b2 ? b : a;
    y = dpp_pair_sum<0x128>(y);                                // row_ror:8 -> total of the 16-lane row
    y += __shfl_xor(y, 16, 64);                                // rows 0+1, 2+3
    y += __shfl_xor(y, 32, 64);                                // whole wave
    return y;
}

// The same butterfly over the eight lanes of an aligned 8-lane GROUP (three steps): afterwards lane l holds the total of
// value (l & 7) over its group.  Step three pairs lane l with l ^ 4: row_shl:4 serves the lanes with bit 2 clear, row_shr:4
// those with bit 2 set (a lane without a source adds 0 and is not the one selected).
__device__ __forceinline__ float group8_sum8_transposed(const float (&v)[8], unsigned lane)
{
    const bool b0 = lane & 1u, b1 = lane & 2u, b2 = lane & 4u;
    float w[4], x[2];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float a = dpp_pair_sum<0xB1>(v[2 * i]);
        const float b = dpp_pair_sum<0xB1>(v[2 * i + 1]);
        w[i] = b0 ? b : a;
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const float a = dpp_pair_sum<0x4E>(w[2 * i]);
        const float b = dpp_pair_sum<0x4E>(w[2 * i + 1]);
        x[i] = b1 ? b : a;
    }
    const float lo = dpp_add<0x104>(x[0]);                     // row_shl:4: + lane l + 4
    const float hi = dpp_add<0x114>(x[1]);                     // row_shr:4: + lane l - 4
    return b2 ? hi : lo;
}
// Plain sum over the aligned 8-lane group, valid in every lane of it.
__device__ __forceinline__ float group8_sum(float v)
{
    v = dpp_pair_sum<0xB1>(v);
    v = dpp_pair_sum<0x4E>(v);
    return dpp_pair_sum<0x141>(v);                             // row_half_mirror: the group's other quad (all its lanes hold the same sum)
}
__device__ __forceinline__ void lds_add_f32(float* p, float v)      // ds_add_f32, no return value, nothing to wait for
{
    (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Lanes of the wave whose digit equals mine (among `valid` lanes); only the low `nbits` (wave-uniform,
// <= 8) bits of the digit can differ, so only those are balloted.
__device__ __forceinline__ uint64_t wave_match8(uint32_t d, bool valid, int nbits = 8)
{
    uint64_t m = __ballot(valid);
    for (int b = 0; b < nbits; b++) {
        bool bit = (d >> b) & 1u;
        uint64_t bal = __ballot(bit);
        m &= bit ? bal : ~bal;
    }
    return m;
}
__device__ __forceinline__ uint64_t lanemask_lt() { return ((uint64_t)1 << lane_id()) - 1; }

#endif // __HIPCC__
} // namespace gsrast
