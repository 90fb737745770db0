/*
 * gsrast.h -- C ABI of the MI355X (gfx950) differentiable Gaussian rasterizer.
 *
 * This is the drop-in boundary for the ONE hot path of yjb6/SaRO-GS: the three static entry
 * points of the reference's native rasterizer,
 *     CudaRasterizer::Rasterizer::forward      (cuda_rasterizer/rasterizer.h:34-58,  impl rasterizer_impl.cu:198-339)
 *     CudaRasterizer::Rasterizer::backward     (cuda_rasterizer/rasterizer.h:60-89,  impl rasterizer_impl.cu:343-436)
 *     CudaRasterizer::Rasterizer::markVisible  (cuda_rasterizer/rasterizer.h:27-32,  impl rasterizer_impl.cu:141-153)
 * (paths relative to /root/reference/submodules/gaussian_rasterization_ch3/), which the reference
 * binds to Python through pybind11 in ext.cpp:15-19 / rasterize_points.cu:35-215.
 *
 * Same argument sets, same meaning, plus an explicit HIP stream.  Differences, all deliberate:
 *   - the three std::function<char*(size_t)> allocators become plain C callbacks + context;
 *   - all pointers are DEVICE pointers (HBM); a NULL pointer means "absent optional input"
 *     exactly as in the reference (shs / colors_precomp / scales / rotations / cov3D_precomp);
 *   - functions return a status (or num_rendered) instead of throwing; gsrast_last_error() has text;
 *   - the contents of the three state buffers are opaque and differ from the reference's chunks
 *     (typed SoA arrays); gsrast_debug_export() copies them out in the
 *     reference's array layout for parity tests;
 *   - `prefiltered` != 0 is the caller's promise that no Gaussian lies behind the near plane: the reference prints and TRAPS the kernel
 *     when one does (auxiliary.h:156-160), gsrast_forward returns GSRAST_E_ARG (after waiting for the device: the promise costs a
 *     synchronisation; SaRO-GS always passes False);
 *   - arithmetic, all within north_star's 1e-5 bar and checked against the fp64 oracle: exp() is a fixed sequence of exactly rounded fp32
 *     operations shared with the oracle (options.exp_mode 0; the reference calls exp(), forward.cu:349), so "bit-exact forward" means
 *     HIP == oracle; the blend backward rebuilds T with v_rcp_f32(1 - alpha) (1 ulp) where the reference divides (backward.cu:503), takes the
 *     geometric sums about the 8 x 8 pixel block's origin and shifts them to the Gaussian's mean afterwards (csrc/gsrast_blend.h: separable
 *     moments), and evaluates backward.cu:505-507's accum_rec for the NEXT contributor (same operands); gradients are accumulated with
 *     float atomics into one 64-byte record per Gaussian (the reference: nine atomics per pixel pair), so their last bits depend on the order.
 * No torch / STL types cross this boundary.
 */
#ifndef GSRAST_H_INCLUDED
#define GSRAST_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSRAST_ABI_VERSION 5   /* 5: gsrast_grad_rows_clear / gsrast_grad_rows_add take P (indices received from peers are bounds-checked).  4: gsrast_raw_grads.d_sh_factor (the struct grew), gsrast_sh_grad_combine_rows.  3: gsrast_options.no_list_cut (the struct grew).  2: gsrast_backward OVERWRITES every output array (version 1 accumulated into caller-zeroed arrays like the
                                    reference); options.forward_only; gsrast_forward_raw / gsrast_backward_raw */
#define GSRAST_TILE_X 16 /* reference config.h:16 */
#define GSRAST_TILE_Y 16 /* reference config.h:17 */

/* Allocation callback: must return a device pointer to at least `bytes` bytes, 256-byte aligned,
 * that stays valid until the matching backward call has completed (the reference keeps the
 * buffers alive through ctx.save_for_backward).  A callback may be invoked more than once per
 * forward call (the binning buffer is first requested from an estimate, then again if the estimate
 * was too small); the LAST pointer returned is the buffer in use, earlier ones may be released.
 * Replaces rasterize_points.cu:27-33. */
typedef void* (*gsrast_alloc_fn)(void* ctx, size_t bytes);

/* Forward pass.  Returns num_rendered (number of (Gaussian, tile) instances, >= 0) or a negative
 * GSRAST_E_* code.  out_color [3][H][W] planar, out_depth [1][H][W] (median depth, default 15.0),
 * radii [P] are fully written.  Blocks the host once on `stream` to learn num_rendered
 * (the reference does the same with a cudaMemcpy, rasterizer_impl.cu:282).
 * Replaces Rasterizer::forward, rasterizer.h:34-58. */
int gsrast_forward(gsrast_alloc_fn geometry_alloc, void* geometry_ctx,
                   gsrast_alloc_fn binning_alloc, void* binning_ctx,
                   gsrast_alloc_fn image_alloc, void* image_ctx,
                   int P, int D, int M,
                   const float* background,
                   int width, int height,
                   const float* means3D,
                   const float* shs,
                   const float* colors_precomp,
                   const float* opacities,
                   const float* scales,
                   float scale_modifier,
                   const float* rotations,
                   const float* cov3D_precomp,
                   const float* viewmatrix,
                   const float* projmatrix,
                   const float* cam_pos,
                   float tan_fovx, float tan_fovy,
                   int prefiltered,
                   float* out_color,
                   float* out_depth,
                   int* radii,
                   void* stream);

/* Backward pass.  Replaces Rasterizer::backward, rasterizer.h:60-89.
 * EVERY output array is fully overwritten (zeros for culled Gaussians), none has to be initialised -- the reference
 * accumulates into nine zero-filled arrays (300 B / Gaussian of memset, rasterize_points.cu:150-158); here the blend
 * backward accumulates into one 64-byte record per Gaussian inside geom_buffer (zero-filled by the forward, and again by
 * this call unless options->grads_zeroed says it is the first backward on that state) and the
 * per-Gaussian backward writes dL_dmean2D [P][3] (.z = 0), dL_dopacity [P], dL_dcolor [P][3], dL_dconic [P][4] (.z = 0,
 * as the reference never writes it) from it, next to dL_dmean3D [P][3], dL_dcov3D [P][6], dL_dsh [P][M][3],
 * dL_dscale [P][3], dL_drot [P][4].  May be NULL: dL_dconic (an intermediate), dL_dcolor unless colors_precomp is given,
 * dL_dcov3D when scales / rotations are given (the reference computes and returns all three regardless).
 * geom_buffer is written (the gradient records), binning_buffer / image_buffer are only read.
 * Returns 0 or GSRAST_E_*. */
int gsrast_backward(int P, int D, int M, int R,
                    const float* background,
                    int width, int height,
                    const float* means3D,
                    const float* shs,
                    const float* colors_precomp,
                    const float* scales,
                    float scale_modifier,
                    const float* rotations,
                    const float* cov3D_precomp,
                    const float* viewmatrix,
                    const float* projmatrix,
                    const float* campos,
                    float tan_fovx, float tan_fovy,
                    const int* radii,
                    char* geom_buffer,
                    char* binning_buffer,
                    char* image_buffer,
                    const float* dL_dpix,
                    float* dL_dmean2D,
                    float* dL_dconic,
                    float* dL_dopacity,
                    float* dL_dcolor,
                    float* dL_dmean3D,
                    float* dL_dcov3D,
                    float* dL_dsh,
                    float* dL_dscale,
                    float* dL_drot,
                    void* stream);

/* present[i] = view-space z of means3D[i] > 0.2.  Replaces Rasterizer::markVisible,
 * rasterizer.h:27-32 (kernel checkFrustum, rasterizer_impl.cu:54-66). */
int gsrast_mark_visible(int P, const float* means3D, const float* viewmatrix,
                        const float* projmatrix, unsigned char* present, void* stream);

/* Sizes the library will request through the callbacks (replaces required<T>(),
 * rasterizer_impl.h:67-73). */
size_t gsrast_geometry_bytes(int P);
size_t gsrast_binning_bytes(int num_rendered, int width, int height);
size_t gsrast_image_bytes(int width, int height);

/* A ready-made allocation callback over memory the caller ALREADY holds (round 6): pass gsrast_alloc_prealloc as the gsrast_alloc_fn and a
 * gsrast_prealloc* as its ctx.  Returns ptr when bytes <= capacity, NULL otherwise (the forward then fails with GSRAST_E_ALLOC); `requested`
 * records what was asked for.  The geometry and image buffers' sizes are known before the call (gsrast_geometry_bytes / gsrast_image_bytes), so a
 * host language whose callbacks are expensive (Python: ~5 us each, in front of the forward's first launch) need only keep its own callback for
 * the binning buffer, whose size the library decides. */
typedef struct gsrast_prealloc { void* ptr; size_t capacity; size_t requested; } gsrast_prealloc;
void* gsrast_alloc_prealloc(void* ctx /* gsrast_prealloc* */, size_t bytes);

/* Multi-GPU gradient exchange (no counterpart in the reference, which is single-GPU and sums the per-view gradients of a
 * batch in place, scene/saro_gaussian.py:226-247, :266-276).  Row k of a view's dL/dsh is w_k(view direction) * g, where
 * g[3] is that view's clamp-masked colour gradient of the Gaussian: instead of all-reducing 16 x 3 products per Gaussian,
 * ranks all-gather the 3 numbers and every rank recombines.  With gsrast_set_option("sh_grad_factors", 1),
 * gsrast_backward writes g into dL_dsh, which is then a [P][3] array.  gsrast_sh_grad_combine evaluates
 *     dL_dsh[i][k][c] = scale * sum_{r < N} w_k(normalize(means3D[i] - campos_r)) * g_r[i][c]
 * chunks = N records of chunk_stride floats: [3P floats g_r | 3 floats campos_r | padding].  All device pointers. */
int gsrast_sh_grad_combine(int P, int D, int M, int N, const float* means3D, const float* chunks, size_t chunk_stride,
                           float scale, float* dL_dsh /*[P][M][3]*/, void* stream);
/* Round 5: which Gaussians can have a non-zero gradient row in this view?  flags[i] = 1 if some pixel consumed Gaussian i in the forward
 * that filled geom_buffer (its blend keeps one bit per Gaussian for the backward, csrc/gsrast_common.h: GeomLayout::untouched), 0 if none
 * did -- every gradient row of such a Gaussian is exactly zero; all 1 if that forward kept no bits.  The sparse exchange takes its
 * "rows some rank touched" from here instead of scanning the five gradient arrays (56 B per Gaussian). */
int gsrast_touched_rows(int P, const char* geom_buffer, unsigned char* flags /*[P]*/, void* stream);
/* The same recombination for an exchange that moves only the rows some rank touched, and for the raw leaves (round 4):
 *   chunks = N records of chunk_stride floats: [3 * rows floats g_r | 3 floats campos_r | padding], rows <= P;
 *   row_of [P] or NULL: Gaussian i's factor is row row_of[i] of every record, -1 = no rank sent it (its dL/dsh is zero);
 *   NULL: rows == P, row i (then this is gsrast_sh_grad_combine);
 *   the result goes to dL_dsh [P][M][3] (rows [dc | rest]) and / or, split, to d_features_dc [P][1][3] + d_features_rest [P][M-1][3]
 *   (the gradients of SaRO-GS's two SH leaves, scene/saro_gaussian.py:836-845); every row of every array given is written. */
int gsrast_sh_grad_combine_rows(int P, int D, int M, int N, const float* means3D, const float* chunks, size_t chunk_stride, int rows,
                                const int* row_of, float scale, float* dL_dsh, float* d_features_dc, float* d_features_rest, void* stream);

/* Round 5: the recombination for the rows of the union ONLY.  idx [rows] (int64, ascending, distinct): record row j is Gaussian idx[j]'s
 * factor.  Rows of the output arrays outside the union are NOT written: the caller keeps them zero between steps (it clears the previous
 * step's union, a fraction of the array, instead of having all P rows rewritten: 3 M Gaussians, 150 k in the union, 29 MB instead of 576).
 * M * 3 must be a multiple of 4 and at most 48; dL_dsh 16-byte aligned. */
int gsrast_sh_grad_combine_union(int P, int D, int M, int N, const float* means3D, const float* chunks, size_t chunk_stride, int rows,
                                 const long long* idx, float scale, float* dL_dsh, float* d_features_dc, float* d_features_rest, void* stream);

/* Round 5: the compaction either side of the sparse exchange.  Rows idx[0..n) (int64, device) of n_arrays (<= 8) row-major float arrays
 * (arrays[k]: device pointer, widths[k] floats per row; `arrays` and `widths` themselves are HOST arrays) side by side into
 * packed [n][sum of widths], and back into those rows (other rows are not touched). */
int gsrast_rows_pack(long long n, const long long* idx, int n_arrays, const float* const* arrays, const int* widths, float* packed, void* stream);
int gsrast_rows_unpack(long long n, const long long* idx, int n_arrays, float* const* arrays, const int* widths, const float* packed, void* stream);

/* Round 5: the ALL-GATHER gradient exchange (csrc/gsrast_exchange.h; view_parallel.exchange_gradients(sparse="gather")).  Every rank
 * sends only the gradient rows its own view touched, 64-byte rows { Gaussian index | 11 dense floats: mean 3, opacity 1, scale 3,
 * rotation 4 | dL/dsh factor 3 | 0 }, in ONE all-gather of chunks { header row: count, campos x y z | cap rows }, and adds the chunks
 * into its arrays in rank order.  `dense`: a HOST array of four device pointers, [P][3], [P][1], [P][3], [P][4].
 *   pack : rows[0] word 0 (the count, zeroed by the caller) counts the touched rows, rows[1 + k] receive them (arrival order), at most cap.
 *   clear: zeroes, for every row the chunks name, the dense arrays' rows (dense != NULL) and / or the SH arrays' rows (any SH pointer given).
 *   add  : dense[idx] += scale * row, dL/dsh[idx] += scale * w(dir(means3D[idx] - campos)) (x) factor -- no atomics: indices within a chunk
 *          are distinct, chunks are added by consecutive launches.
 *   clear / add take P (round 6): the indices inside a chunk come from a peer; a row whose index is >= P is skipped, never written. */
int gsrast_grad_rows_pack(int P, const unsigned char* touched /*[P]*/, float* const* dense, const float* factor /*[P][3]*/, uint32_t* rows, uint32_t cap, void* stream);
int gsrast_grad_rows_clear(int P, const uint32_t* chunks, int n_chunks, size_t chunk_words, uint32_t cap, float* const* dense /* or NULL */, int M,
                           float* dL_dsh, float* d_features_dc, float* d_features_rest, void* stream);
int gsrast_grad_rows_add(int P, const uint32_t* chunk, uint32_t cap, float* const* dense, int D, int M, const float* means3D, float scale,
                         float* dL_dsh, float* d_features_dc, float* d_features_rest, void* stream);

/* Parity-test helper: copies internal state out in the reference's array layout
 * (GeometryState / BinningState / ImageState members, rasterizer_impl.h:30-65).  Any output
 * pointer may be NULL.  All pointers are device pointers.  keys_sorted is rebuilt as
 * (tile << 32) | depth_bits from the sorted instance list. */
/* NOTE for consumers of the state buffers: with options.tile_clip = 1 (the default) the binning lists a Gaussian only in the tiles
 * its alpha >= 1/255 ellipse reaches, so the instance list holds sum(ranges.y - ranges.x) <= num_rendered entries -- num_rendered (the
 * return value of gsrast_forward) keeps the REFERENCE's meaning (tiles of the 3-sigma squares, rasterizer_impl.cu:277-282) and is then
 * NOT the length of point_list / keys_sorted; size those arrays for num_rendered and read only the ranges.  With tile_clip = 0 the
 * lists are the reference's literal ones and the two numbers coincide.  n_contrib indexes the list in force. */
int gsrast_debug_export(int P, int R, int width, int height,
                        const char* geom_buffer, const char* binning_buffer, const char* image_buffer,
                        float* depths, float* means2D /*[P][2]*/, float* cov3D /*[P][6]*/,
                        float* conic_opacity /*[P][4]*/, float* rgb /*[P][3]*/,
                        unsigned char* clamped /*[P][3]*/, uint32_t* tiles_touched,
                        uint64_t* keys_sorted /*[R]*/, uint32_t* point_list /*[R]*/,
                        uint32_t* ranges /*[T][2]*/, float* final_T /*[H*W]*/,
                        uint32_t* n_contrib /*[H*W]*/, void* stream);

/* ---- Reentrancy: per-call options and contexts ---------------------------------------------------------------------
 * The reference's Rasterizer::{forward, backward} are stateless statics (rasterizer.h:24-83): any number of host threads
 * may call them on different streams / devices.  The same holds here:
 *   - everything that changes what a call computes or how it is scheduled is a field of gsrast_options, passed per call to
 *     gsrast_forward_ex / gsrast_backward_ex (NULL = a snapshot of the process defaults, taken once at entry);
 *   - the only state that outlives a forward call -- the capacity hints of the speculative launch and the counts of the last
 *     call -- lives in a gsrast_context the caller owns (NULL = a context private to the calling host thread);
 *   - gsrast_last_error() is per host thread.
 * gsrast_forward / gsrast_backward are exactly gsrast_forward_ex(NULL, NULL, ...) / gsrast_backward_ex(NULL, ...).
 * gsrast_set_option only edits the process DEFAULTS (plus the process-wide diagnostics "profile", "debug_sync"): callers that
 * want different behaviour on different threads pass a gsrast_options instead. */
typedef struct gsrast_options {
    int exp_mode;             /* 0 fixed-sequence exp (default), 1 libm-grade expf, 2 v_exp_f32 */
    int binning;              /* 0 run-compressed (default), 1 instance-level two-pass radix sort */
    int tile_clip;            /* 1 (default) list a Gaussian only in tiles its alpha >= 1/255 ellipse reaches; 0 literal lists */
    int cull;                 /* 1 (default) wave-level strip culling in the blend kernels */
    int lpt;                  /* 1 (default) heaviest-tile-first launch order of the blend kernels */
    int speculative;          /* 1 (default) enqueue binning + blend before num_rendered is read back */
    int fwd_pixels_per_lane;  /* 0 auto (default), 1 / 2 / 4 */
    int bwd_pixels_per_lane;  /* 0 auto (default), 1 / 2 / 4 */
    int sh_grad_factors;      /* backward: dL_dsh receives the [P][3] factor, see gsrast_sh_grad_combine */
    int side_stream;          /* 1 (default) forward: the colour kernel (SH -> RGB) runs on a stream of the context, forked off
                                 `stream` at entry and joined in front of the blend, beside the depth sort and the binning, and the
                                 zero-fill of the gradient records follows on it under the blend (joined before the call returns);
                                 backward: the SH view-direction derivatives are evaluated on the calling thread's context stream
                                 beside the blend backward, joined in front of the per-Gaussian backward.
                                 0 = everything on `stream` */
    int grads_zeroed;         /* backward: 1 = no backward has run on this forward's geometry buffer yet (the forward leaves the
                                 gradient records zero), so the 64 B / Gaussian zero-fill is skipped; 0 (default) = fill */
    int backward_phase;       /* backward: 0 (default) everything; 1 = the blend backward only (fills the gradient records and, with
                                 sh_grad_factors, writes the factors: all a multi-GPU caller needs to START its exchange); 2 = the
                                 per-Gaussian backward only (the rest of the outputs).  1 then 2 on the same arguments == 0 */
    int depth_sort;           /* forward: 0 (default) bucket depth sort -- two launches: Gaussians into ~P/256 depth buckets, one LDS sort per
                                 bucket (run-compressed binning, P >= 32768); a scene whose depths pile up in one bucket is detected on
                                 the device and re-sorted by the radix passes, which the context then uses for its next 16 calls
                                 ("bucket_skip"; doubling with every further overflow, up to 4096); 1 = always the LSD radix sort (3-4 passes of three launches).  Same order either way */
    int forward_only;         /* forward: 1 = no backward will follow on this call's state (evaluation / torch.no_grad()): the colour kernel
                                 does not store d(colour)/d(view direction) (36 B / Gaussian) for the backward.  A backward on such a
                                 state must be given forward_only = 1 as well; it then evaluates those derivatives itself
                                 (sh_dir_derivs_kernel, re-reading the SH blocks).  0 (default) = the forward prepares them */
    int no_order_hint;        /* forward: 1 = do not use / update the context's launch-order hints.  By default a context remembers, per device and
                                 camera pose (hash of the view and projection matrices and the image size; 256 poses, least recently used
                                 replaced; device memory, 6 B per tile and pose with the cut depths of no_list_cut below), how deep every tile's list was consumed the last time
                                 that pose was rendered, and starts the forward blend's heaviest tiles first by it -- the list length, the
                                 only estimate a first-seen pose has, is a poor one in occluded scenes.  Results never depend on it */
    int dense_backward;       /* backward: 1 = the per-Gaussian backward reads every Gaussian's inputs (round 2's form).  0 (default): a
                                 Gaussian whose gradient record is all zero (frustum-culled, or occluded: its gradient IS zero) gets its
                                 zero rows written without its inputs being read */
    int no_list_cut;          /* forward: 1 = always bin every Gaussian.  0 (default): LIST CUT -- with the launch-order hints a context also
                                 remembers, per pose and tile, a cut depth (that of the list entry 1.5 x as deep as the deepest one any pixel
                                 of the tile consumed the last time; none for a tile whose pixels did not all saturate).  The next forward of
                                 that pose gives column runs only to the Gaussians in front of the cut depth of some tile they cover -- in an
                                 occluded scene a few per cent of them -- and VERIFIES the speculation on the device: a tile's list counts
                                 as ending at its cut depth, and if some pixel of a cut tile is not saturated there, the whole binning and
                                 blend run again over all Gaussians (enqueued behind the blend in any case, every kernel predicated on the
                                 verdict).  Results never depend on it; needs tile_clip = 1 and the bucket depth sort.  It is
                                 applied where it pays: when the context's last forward had at least 1.5 M column runs, and not for the next 64
                                 forwards after one in which it removed fewer than that (gsrast_set_option("list_cut_always", 1) lifts both) */
} gsrast_options;
void gsrast_options_init(gsrast_options* options);   /* fills in the built-in defaults listed above */
/* A context may be used by one host thread at a time (it owns one side stream and one set of fork / join events per device, and -- per
 * device it has rendered on -- 6 bytes per tile and pose of device memory (12.5 MB at 1080p, 50 MB at 4K): the pose table of options.no_order_hint / no_list_cut); contexts are
 * independent of each other.  Destroy it only after the calls that used it have returned (gsrast_context_destroy frees the device
 * memory, which waits for the device). */
typedef struct gsrast_context gsrast_context;
gsrast_context* gsrast_context_create(void);
void gsrast_context_destroy(gsrast_context* ctx);
/* "last_late" (Gaussians the list cut left without column runs in the context's last forward call), "last_early_runs" (column runs of the others),
 * "cut_pause" (forwards the list cut still sits out: it saved too little, or its lists kept turning out too short), "cut_fallbacks" (forwards on the
 * current device whose cut lists turned out too short and were redone from the full lists; this query waits for the device),
 * "last_instances" (num_rendered), "last_runs" (column runs) of the context's last forward call, "redo_count"
 * (speculative launches / depth sorts that had to be repeated), "bucket_skip" (forwards that will still go straight to the radix
 * depth sort after a bucket overflow); ctx NULL = the calling thread's context. */
int gsrast_context_query(const gsrast_context* ctx, const char* name);
/* Round 5: also "completion_passes" (completion passes of the list cut the device has reported to this context), "cut_margin_x4" (the remembered
 * cut's margin in quarters: 6 = 1.5 x), "tau_req" / "tau_force" (the predicted cut's requirement / forwards that still use predicted cut depths for
 * every pose), "gate_inline_calls".
 *
 * The host-side decisions of a context (csrc/gsrast_policy.h: when the list cut is applied, paused and widened; how the speculative launch is
 * sized) can be driven WITHOUT a device -- a context that never renders touches no GPU.  One event per call, the return value is the
 * decision (or GSRAST_E_ARG for an unknown event):
 *   "begin"  (a = Gaussians P, b = column runs of the previous forward, c = 1: option list_cut_always) -> 1: the cut pays for this forward; 0: it sits out
 *   "counts" (a = all column runs Q, b = early column runs, c = bit 0: predicted cut depths available, bit 1: some Gaussian was late; P = the last "begin"'s) -> forwards of pause now pending
 *   "pass"   (a = column runs of the completion pass's candidates, b = column runs of the forward, c = 1: predicted cut depths available) -> the pass's points (0 ... 8)
 *   "clean"  (a cut forward behind which no pass was reported) -> the score
 *   "size"   (a = early-run hint, b = capacity in column runs, c = 1: an early set is expected) -> column runs the launches over the cut lists are sized for
 *   "grow"   (a = count) -> the capacity requested for it;   "follow" (a = hint, b = this forward's count, c = shift) -> the next hint
 *   "get"    (a = 0 pause, 1 score, 2 margin x 4, 3 tau_req, 4 tau_force, 5 length of the last fallback pause)
 *   "zrange" (a = first | last << 16 occupied bin of the depth histogram a forward filled with the context's current table) -> the next table's
 *            shift, | 256 if the current table was a learned one that held every key;   "zget" (a = 0 klo / 256, 1 shift, 2 khi / 256)
 *   "reset"  (a fresh policy: tests);   "tau_min" (a = the predicted cut's requirement and its floor: experiments)
 * ctx NULL = the calling thread's context. */
int gsrast_policy_event(gsrast_context* ctx, const char* what, int a, int b, int c);
int gsrast_forward_ex(gsrast_context* ctx, const gsrast_options* options,
                      gsrast_alloc_fn geometry_alloc, void* geometry_ctx,
                      gsrast_alloc_fn binning_alloc, void* binning_ctx,
                      gsrast_alloc_fn image_alloc, void* image_ctx,
                      int P, int D, int M, const float* background, int width, int height,
                      const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
                      const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                      const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                      float tan_fovx, float tan_fovy, int prefiltered,
                      float* out_color, float* out_depth, int* radii, void* stream);
int gsrast_backward_ex(const gsrast_options* options,
                       int P, int D, int M, int R, const float* background, int width, int height,
                       const float* means3D, const float* shs, const float* colors_precomp,
                       const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                       const float* viewmatrix, const float* projmatrix, const float* campos,
                       float tan_fovx, float tan_fovy, const int* radii,
                       char* geom_buffer, char* binning_buffer, char* image_buffer, const float* dL_dpix,
                       float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor,
                       float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, void* stream);

/* Process defaults of the options above (and process-wide diagnostics): "exp_mode" 0 = fixed-sequence exp (bit-reproducible vs the CPU oracle), 1 = libm-grade
 * expf, 2 = hardware v_exp_f32;  "profile" = bit mask of kernel ids (gsrast_profile_kernel_name) whose launches are bracketed
 * with HIP events on the launch stream, -1 = all, 0 = off;
 * "debug_sync" 0/1 = synchronise + check errors after every launch;
 * "binning" 0 = run-compressed binning (default), 1 = instance-level two-pass radix sort;
 * "tile_clip" 1 (default) = with "binning" 0, a Gaussian is listed only in the tiles its alpha >= 1/255 ellipse can
 * reach instead of every tile of its 3-sigma square (outputs bit-identical, the internal lists get shorter;
 * num_rendered keeps the reference's meaning), 0 = the reference's literal lists;
 * "cull" / "lpt" 0/1 = wave-level strip culling / heaviest-tile-first launch order in the blend kernels;
 * "pixels_per_lane" (+ "fwd_" / "bwd_" prefixed) 0 = auto, 1 / 2 / 4.  Returns 0 or GSRAST_E_ARG.
 * "speculative" 1 (default) = binning + forward blend are enqueued before the host has read num_rendered back, against
 * a capacity remembered from earlier calls (repeated with exact sizes if it did not fit), 0 = wait first;
 * "sh_grad_factors" see gsrast_sh_grad_combine;
 * "bwd_transposed" (process-wide A/B switch) 1 (default) = the one-pixel-per-lane backward blend sums across lanes once per
 * group of eight staged instances (blend_bwd_cull_t_kernel), 0 = nine wave reductions per surviving (wave, instance) pair.
 * "chain_gate" (process-wide A/B switch) 1 (default) = the list cut's completion pass (no_list_cut above) is enqueued on the context's
 * second stream and the caller's stream is released by the cut forward's blend itself (hipStreamWaitValue32 on a word of the
 * context's own), 0 = its predicated launches on the caller's stream (also chosen by itself when the process runs under a counter-collecting
 * profiler -- ROCPROF_COUNTERS / ROCPROF_COUNTER_GROUPS in the environment: such a profiler serialises kernels, a stream that waits for another deadlocks);  "layer_cut" 1 = a pose without remembered cut depths lists
 * the nearest eighth of the Gaussians first (measured slower: default 0);  "list_cut_always" 1 = the cut also where it does not pay;
 * "near_pose" r (default 0 = off; 3 in round 4): a camera pose the context's table does not know takes the launch order and the cut depths of a
 * NEAR pose's slot (a camera path's previous frame), the cut depths widened over (2 r + 1)^2 tiles -- verified like any cut.  Off since such a
 * pose gets predicted cut depths ("tau_cut"), which measured faster along a camera path.
 * Round 5 (process-wide A/B switches, default 1): "tau_cut" = cut depths PREDICTED from the call's own opacity mass for a pose without remembered
 * ones;  "touch_bits" = the forward blend keeps one "no pixel consumed it" bit per Gaussian for the backward;  "sparse_grec" = such a forward
 * zeroes only the consumed Gaussians' gradient records instead of all P (the backward takes every other record for zero);
 * "word_fork" = the side stream is forked by the next kernel's own start (a stored word, hipStreamWaitValue32) instead of an event where a kernel can do that;
 * "late_fill_min_p" (default 750000): scenes of at least that many Gaussians write the untouched Gaussians' zero rows beside the blend backward.
 * Read-only through gsrast_get_option: "last_instances" (num_rendered) and "last_runs" (column runs) of the
 * last forward call of the CALLING THREAD's context, "redo_count" (= gsrast_context_query(NULL, name)). */
int gsrast_set_option(const char* name, int value);
int gsrast_get_option(const char* name);

/* Per-kernel device timing gathered while "profile" is 1 (HIP events on the launch stream).
 * gsrast_profile_collect() synchronises outstanding events and folds them into the totals. */
int gsrast_profile_kernel_count(void);
const char* gsrast_profile_kernel_name(int kernel_id);
int gsrast_profile_collect(void);
int gsrast_profile_read(int kernel_id, double* total_ms, long long* launches);
void gsrast_profile_reset(void);

/* ---- "next" row of the scope table: the photometric loss right after the rasterizer ----
 * loss = (1 - lambda_dssim) * mean|img - gt| + lambda_dssim * (1 - mean(SSIM_map(img, gt)))
 * Replaces the pair l1_loss / ssim of the reference's utils/loss_utils.py:18-19, :38-68 as combined in
 * helper_train.py:50-53 (5 depthwise 11x11 convolutions forward, autograd through them backward) by one fused
 * forward and one fused backward kernel.  img, gt: [C][H][W] planar fp32 in HBM.
 * forward : out3 (device, 3 floats) = {loss, l1, ssim}; scratch keeps three derivative maps for the backward.
 * backward: dL_dimg [C][H][W] = dL_dloss * d loss / d img  (dL_dloss: device scalar, NULL means 1). */
size_t gsrast_loss_scratch_bytes(int C, int H, int W);
int gsrast_loss_forward(int C, int H, int W, const float* img, const float* gt, float lambda_dssim,
                        float* out3, char* scratch, void* stream);
int gsrast_loss_backward(int C, int H, int W, const float* img, const float* gt, float lambda_dssim,
                         const float* dL_dloss, const char* scratch, float* dL_dimg, void* stream);

/* ---- "next" row, rank 3: the activation / deformation epilogue that produces the rasterizer's inputs ----
 * Replaces the tail of get_deformation, scene/saro_gaussian.py:807-847 with the activations of :39-47:
 *   motion = xyz + motion_res;  rot = normalize(rotation + rot_res[:, :4]);  scale = exp(scaling + rot_res[:, 4:]);
 *   opacity = sigmoid(opacity_logit) * trbf;  shs = cat(features_dc [P][1][3], features_rest [P][M-1][3]) + shs_res [P][M][3]
 * motion_res, rot_res ([P][7]), trbf and shs_res may each be NULL (static stage: plain activations).
 * backward: upstream d_rot [P][4], d_scale [P][3], d_opacity [P] (NULL = zero) -> d_rotation [P][4], d_scaling [P][3],
 * d_rot_res [P][7] (NULL ok), d_opacity_logit [P], d_trbf [P] (NULL ok).  The other gradients need no kernel:
 * d_xyz = d_motion_res = d_motion; d_features_dc / d_features_rest are slices of d_shs, d_shs_res = d_shs. */
int gsrast_activate_forward(int P, int M, const float* xyz, const float* motion_res, const float* rotation,
                            const float* rot_res, const float* scaling, const float* opacity_logit, const float* trbf,
                            const float* features_dc, const float* features_rest, const float* shs_res,
                            float* motion, float* rot, float* scale, float* opacity, float* shs, void* stream);
int gsrast_activate_backward(int P, const float* rotation, const float* rot_res, const float* scale, const float* opacity_logit,
                             const float* trbf, const float* d_rot, const float* d_scale, const float* d_opacity,
                             float* d_rotation, float* d_scaling, float* d_rot_res, float* d_opacity_logit, float* d_trbf,
                             void* stream);

/* ---- rank 3 as SURVEY.md 8f wrote it: the epilogue FUSED INTO the per-Gaussian kernels (K1 / K7) ----
 * gsrast_forward_raw / gsrast_backward_raw are gsrast_forward_ex / gsrast_backward_ex taking the model's RAW leaves
 * (scene/saro_gaussian.py:306-319: _xyz, _rotation, _scaling, _opacity, _features_dc, _features_rest) and the optional deformation
 * residuals of get_deformation (:807-847) instead of activated attributes: the activations above run in registers inside
 * preprocess_fwd / preprocess_color, the chain rule inside preprocess_bwd, and the [P][M][3] coefficient tensor
 * (cat(dc, rest) + residual: 192 B / Gaussian written by the model and re-read by the rasterizer, and back) is never materialised.
 * The rasterizer's outputs and state are bit-identical to gsrast_activate_forward followed by gsrast_forward_ex.
 * The reference-shaped entry points are untouched; this pair is additional.  M in {4, 16}; rotation, features_dc, features_rest,
 * shs_res and the matching gradient arrays 16-byte aligned.  NULL = absent optional residual. */
typedef struct gsrast_raw_inputs {
    const float* xyz;             /* [P][3]  _xyz */
    const float* motion_res;      /* [P][3]  or NULL: means3D = xyz + motion_res */
    const float* rotation;        /* [P][4]  _rotation (not normalised) */
    const float* rot_res;         /* [P][7]  or NULL: [:, :4] added to rotation, [:, 4:] to scaling, before the activations */
    const float* scaling;         /* [P][3]  _scaling (log scale) */
    const float* opacity_logit;   /* [P]     _opacity */
    const float* trbf;            /* [P]     or NULL: opacity = sigmoid(logit) * trbf */
    const float* features_dc;     /* [P][1][3] */
    const float* features_rest;   /* [P][M-1][3] */
    const float* shs_res;         /* [P][M][3] or NULL */
} gsrast_raw_inputs;
typedef struct gsrast_raw_grads {   /* every array is fully overwritten */
    float* dL_dmean2D;            /* [P][3]  screen-space gradient (densification statistic), .z = 0 */
    float* d_xyz;                 /* [P][3]  = the gradient of motion_res too */
    float* d_rotation;            /* [P][4] */
    float* d_scaling;             /* [P][3] */
    float* d_rot_res;             /* [P][7]  or NULL ({d_rotation, d_scaling} side by side) */
    float* d_opacity_logit;       /* [P] */
    float* d_trbf;                /* [P]     or NULL */
    float* d_features_dc;         /* [P][1][3]    } may both be NULL when d_shs_res is given (its rows are [dc | rest]); given together with it, */
    float* d_features_rest;       /* [P][M-1][3]  } the rows are written twice -- cheaper than slicing them out afterwards */
    float* d_shs_res;             /* [P][M][3] or NULL; requires shs_res */
    float* d_sh_factor;           /* [P][3] or NULL (round 4, multi-GPU): the FACTOR of the SH leaves' gradient (gsrast_sh_grad_combine) instead of
                                     its 48 products per Gaussian -- d_features_dc / d_features_rest may then be NULL and are not written (the caller
                                     completes them with gsrast_sh_grad_combine_rows after the exchange); not together with shs_res / d_shs_res, whose
                                     gradient every rank needs whole for its own view */
} gsrast_raw_grads;
int gsrast_forward_raw(gsrast_context* ctx, const gsrast_options* options,
                       gsrast_alloc_fn geometry_alloc, void* geometry_ctx,
                       gsrast_alloc_fn binning_alloc, void* binning_ctx,
                       gsrast_alloc_fn image_alloc, void* image_ctx,
                       int P, int D, int M, const float* background, int width, int height,
                       const gsrast_raw_inputs* inputs, float scale_modifier,
                       const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                       float tan_fovx, float tan_fovy, float* out_color, float* out_depth, int* radii, void* stream);
int gsrast_backward_raw(const gsrast_options* options, int P, int D, int M, int R, const float* background, int width, int height,
                        const gsrast_raw_inputs* inputs, float scale_modifier,
                        const float* viewmatrix, const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy,
                        const int* radii, char* geom_buffer, char* binning_buffer, char* image_buffer,
                        const float* dL_dpix, const gsrast_raw_grads* grads, void* stream);

/* ---- "next" row, rank 4 (third item): Adam step of the per-Gaussian parameter groups with a PER-ROW learning rate ----
 * Replaces torch.optim.Adam(l, lr=0.0, eps=1e-15, fused=True) for the groups of scene/saro_gaussian.py:306-323 whose
 * 'lr' update_learning_rate (:345-398) sets to lr * inv_intergral, a [P,1] tensor.  One launch for up to 8 groups:
 *   m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= (lr_i / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
 * with lr_i = lr * (lr_rows ? lr_rows[row] : 1).  All tensors fp32, contiguous [rows][width], device pointers. */
typedef struct gsrast_adam_group {
    float* param; const float* grad; float* exp_avg; float* exp_avg_sq;
    const float* lr_rows;   /* [rows] or NULL */
    float lr;
    int rows, width;
} gsrast_adam_group;
int gsrast_adam_step(int n_groups, const gsrast_adam_group* groups /* host array */, double beta1, double beta2, double eps,
                     int step /* 1-based */, void* stream);   /* betas in fp64: (1 - 0.999f) would be off by 1.3e-5 relative */

/* ---- "next" row, rank 4 (second item): simple_knn._C.distCUDA2 ----
 * mean_dist2[i] = mean of the squared distances from point i to its 3 nearest neighbours (other indices; duplicates count).
 * Replaces the un-vendored dependency imported at scene/saro_gaussian.py:21 and used at :187 (scale initialisation).
 * points [P][3] fp32, mean_dist2 [P], scratch >= gsrast_knn_scratch_bytes(P) bytes; all device pointers.
 * With fewer than 4 points the missing neighbours count as FLT_MAX, as in the original. */
size_t gsrast_knn_scratch_bytes(int P);
int gsrast_knn3_mean_dist2(int P, const float* points, float* mean_dist2, char* scratch, void* stream);

/* ---- "next" row, rank 4 (first item): mip-mapped feature-plane lookup of the residual field ----
 * Replaces nvdiffrast.torch.texture(grid, coords, mip_level_bias=levels, boundary_mode="clamp", max_mip_level=7|0) at
 * scene/hexplane.py:49-56 together with the plane loop of interpolate_ms_features (scene/hexplane.py:95-139): every plane
 * of every scale in one forward launch.  Published algorithm of that op (linear-mipmap-linear, level from the bias only)
 * as restated in oracle/texture_oracle.py.
 *   plane      channel-last level 0 [H][W][C] fp32 (hexplane.py:35 layout); u = pts[n][cu], v = pts[n][cv] in [0,1]
 *              texture coordinates; bias = min(levels[n][cu], levels[n][cv]) (hexplane.py:46)
 *   features   [N][F]; plane p adds its C channels at features[n][out_offset .. out_offset + C); planes sharing an
 *              out_offset (the six planes of a scale) must be adjacent in the array and are summed in array order
 *   C          power of two in [4, 64]; D = floats per pts / levels row (4 in the reference)
 *   scratch    >= gsrast_hexplane_scratch_bytes() bytes (mip stacks: built by every forward call, as the op does; gradient
 *              stacks and the sorted (plane, point) pairs of the backward)
 * backward: grad_tex of every plane is overwritten with dL/dtex; d_pts / d_levels [N][D] optional (the reference detaches
 * both, scene/saro_gaussian.py:780); mips_built != 0 says scratch still holds the forward's stacks of these textures.
 * All pointers inside gsrast_plane and the tensor arguments are device pointers; `planes` itself is a host array. */
typedef struct {
    const float* tex;
    float* grad_tex;        /* backward only */
    int W, H;
    int cu, cv;
    int max_mip_level;      /* the op's max_mip_level: 0 = no mip stack */
    int out_offset;
} gsrast_plane;
size_t gsrast_hexplane_scratch_bytes(int n_planes, const gsrast_plane* planes, int C, int N);
int gsrast_hexplane_forward(int N, int D, int C, int F, int n_planes, const gsrast_plane* planes, const float* pts,
                            const float* levels, float* features, char* scratch, void* stream);
int gsrast_hexplane_backward(int N, int D, int C, int F, int n_planes, const gsrast_plane* planes, const float* pts,
                             const float* levels, const float* d_features, float* d_pts, float* d_levels, int mips_built,
                             char* scratch, void* stream);

const char* gsrast_last_error(void);
int gsrast_abi_version(void);

enum {
    GSRAST_OK = 0,
    GSRAST_E_ARG = -1,     /* bad argument combination / NULL required pointer */
    GSRAST_E_ALLOC = -2,   /* allocation callback returned NULL */
    GSRAST_E_DEVICE = -3,  /* HIP runtime error, see gsrast_last_error() */
    GSRAST_E_OVERFLOW = -4 /* more than 2^31-1 instances */
};

#ifdef __cplusplus
}
#endif
#endif /* GSRAST_H_INCLUDED */
