// gsrast_capi.hip -- host orchestration + the C ABI declared in include/gsrast.h.
// Built with: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -munsafe-fp-atomics (see build.py).
//
// Launch plan of the default path (every launch on the caller's stream except the colour kernel):
//   forward : preprocess_fwd (geometry; claims the pose's launch-order hint slot) -> depth_bucket_scatter -> depth_bucket_sort
//             -> [fork: preprocess_color on the context's side stream] -> emit_column_runs (workgroup 0: totals -> pinned host memory,
//             the host spins on them while the rest is enqueued speculatively) -> run sort by column (hist, rowscan, scatter)
//             -> run_hist_rows -> rowscan -> run_scatter_rows -> tile_ranges_from_runs -> [join] -> blend_fwd_cull (also zero-fills
//             the gradient records, writes the pose's hints and the backward's launch order)
//   backward: blend_bwd_cull_t -> preprocess_bwd (reference K6 + K7 fused; RAW: + the activations' chain rule)
// Fallbacks: radix depth sort (+ scan) after a bucket overflow or with options.depth_sort = 1; instance-level binning with
// options.binning = 1; repeated binning + blend with exact sizes when the speculative capacities did not fit.
#include "../../include/gsrast.h"
#include "gsrast_common.h"
#include "gsrast_policy.h"
#include "gsrast_preprocess.h"
#include "gsrast_binning.h"
#include "gsrast_blend.h"
#include "gsrast_loss.h"
#include "gsrast_epilogue.h"
#include "gsrast_adam.h"
#include "gsrast_knn.h"
#include "gsrast_hexplane.h"
#include "gsrast_exchange.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <thread>
#include <functional>
#include <map>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <dlfcn.h>
#ifndef GSRAST_LATE_FILL_WGS
#define GSRAST_LATE_FILL_WGS 256      // workgroups of late_rows_zero_kernel beside the blend backward: as few as finish under it -- 3 M views/s with 64 / 128 / 256 / 512 / 1024 / 4096: 676 (too slow: the per-Gaussian backward waits) / 803-806 / 790-813 / 806-808 / 801-804 / 797; 2 M: 883 / 935 / 953
#endif
#include <functional>
#include <mutex>
#include <string>
#include <vector>
#include <unordered_map>

using namespace gsrast;

namespace {

thread_local std::string g_err;
std::atomic<int> g_profile{0}, g_debug_sync{0}, g_ablate{0}, g_debug_state{0}, g_list_cut_always{0}, g_chain_gate{1} /* 1: the completion pass of the list cut runs on its own stream behind a gate (ChainGate); 0: inline, eleven predicated launches on the caller's stream */, g_touch_bits{1} /* 1: the forward blend keeps GeomLayout::untouched for the backward (A/B switch) */, g_sparse_grec{1} /* 1: a forward that keeps those bits zeroes only the consumed Gaussians' gradient records (A/B switch) */, g_late_fill_min_p{750000} /* scenes of at least this many Gaussians write their zero rows beside the blend backward */, g_near_pose{0} /* r > 0: a pose the table does not know borrows a near pose's launch order and cut depths (HintTable::cam), widened over (2 r + 1)^2 tiles.  Default 3 in round 4; 0 since round 5: such a pose gets PREDICTED cut depths, which serve a camera path better (3 M, 50 new poses 1.5 degrees apart, forward only: 0.786 ms per view against 0.887 with borrowing, which left 11 of the 50 frames without a cut) */, g_layer_cut{0} /* 1: a pose without remembered cut depths lists a depth LAYER first (measured slower, see DESIGN.md: off) */,
                 g_tau_sample{1} /* the predicted cut's opacity mass comes from one wave in 2^this of preprocess_fwd */, g_tau_cut{1} /* 1: a pose without (trustworthy) remembered cut depths gets PREDICTED ones from this call's own opacity mass (gsrast_common.h) */;      // process-wide diagnostics (not per-call behaviour)

// Per-call behaviour lives in a gsrast_options value: the *_ex entry points take one, the reference-shaped entry points
// snapshot the process defaults (gsrast_set_option) once at entry, so a call never sees a half-changed set and two host
// threads driving different streams / devices with different options cannot disturb each other.
struct DefaultOptions {
    std::atomic<int> exp_mode{0}, binning{0}, tile_clip{1}, cull{1}, lpt{1}, speculative{1}, fwd_ppl{0}, bwd_ppl{0}, sh_grad_factors{0}, side_stream{1}, depth_sort{0}, forward_only{0}, no_order_hint{0}, dense_backward{0}, no_list_cut{0};
} g_def;
gsrast_options snapshot_defaults()
{
    gsrast_options o{};
    o.exp_mode = g_def.exp_mode; o.binning = g_def.binning; o.tile_clip = g_def.tile_clip; o.cull = g_def.cull; o.lpt = g_def.lpt;
    o.speculative = g_def.speculative; o.fwd_pixels_per_lane = g_def.fwd_ppl; o.bwd_pixels_per_lane = g_def.bwd_ppl;
    o.sh_grad_factors = g_def.sh_grad_factors; o.side_stream = g_def.side_stream; o.depth_sort = g_def.depth_sort;
    o.forward_only = g_def.forward_only; o.no_order_hint = g_def.no_order_hint; o.dense_backward = g_def.dense_backward; o.no_list_cut = g_def.no_list_cut;
    return o;
}
bool options_valid(const gsrast_options& o)
{
    auto ppl_ok = [](int v) { return v == 0 || v == 1 || v == 2 || v == 4; };
    return o.exp_mode >= 0 && o.exp_mode <= 2 && (o.binning == 0 || o.binning == 1) && ppl_ok(o.fwd_pixels_per_lane) && ppl_ok(o.bwd_pixels_per_lane) &&
           o.backward_phase >= 0 && o.backward_phase <= 2 && (o.depth_sort == 0 || o.depth_sort == 1) && (o.dense_backward == 0 || o.dense_backward == 1);
}

int fail(int code, const char* what, hipError_t e = hipSuccess)
{
    char buf[512];
    if (e != hipSuccess) snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    else snprintf(buf, sizeof buf, "%s", what);
    g_err = buf;
    return code;
}

#define GS_HIP(call)                                                                  \
    do {                                                                              \
        hipError_t e_ = (call);                                                       \
        if (e_ != hipSuccess) return fail(GSRAST_E_DEVICE, #call, e_);                \
    } while (0)

// ---- per-kernel device timing (option "profile") -------------------------------------------
enum KernelId { K_PREPROCESS_FWD, K_SORT_DEPTH, K_SCAN_TILES, K_EMIT, K_SORT_TILE, K_RANGES, K_BLEND_FWD,
                K_BLEND_BWD, K_PREPROCESS_BWD, K_MARK_VISIBLE, K_LOSS_FWD, K_LOSS_BWD, K_COLOR, K_SH_DERIVS, K_CUT_REDO, K_LATE_ZERO, K_GREC_ZERO, K_COUNT };
const char* const kKernelNames[K_COUNT] = { "preprocess_fwd", "sort_depth", "scan_tiles", "emit_instances",
                                            "sort_tile", "tile_ranges", "blend_fwd", "blend_bwd",
                                            "preprocess_bwd", "mark_visible", "loss_fwd", "loss_bwd", "preprocess_color", "sh_dir_derivs",
                                            "cut_redo" /* list cut: the predicated second binning + blend behind the forward blend, as ONE stage */,
                                            "late_rows_zero" /* list cut: the late Gaussians' zero rows, on the side stream beside the blend backward */,
                                            "grec_zero_touched" /* the consumed Gaussians' gradient records zeroed behind the forward's last blend */ };
thread_local int t_prof_off = 0;      // > 0: the stages below are part of an enclosing one (cut_redo) and not recorded on their own
struct Pending { int id; hipEvent_t a, b; };
std::mutex g_prof_mu;
std::vector<Pending> g_pending;
double g_total_ms[K_COUNT];
long long g_launches[K_COUNT];

// roctx ranges (the stages as named ranges in a rocprofv3 --marker-trace): only with GSRAST_ROCTX set in the environment, the
// marker library looked up at run time (no link-time dependency).  A range brackets the host's ENQUEUE of a stage, as ranges do.
struct Roctx {
    int (*push)(const char*) = nullptr; int (*pop)() = nullptr;
    Roctx()
    {
        if (!getenv("GSRAST_ROCTX")) return;
        void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
        pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (!push || !pop) { push = nullptr; pop = nullptr; }
    }
};
const Roctx& roctx() { static const Roctx r; return r; }
struct RoctxRange {
    bool on;
    explicit RoctxRange(const char* name) : on(roctx().push != nullptr) { if (on) (void)roctx().push(name); }
    ~RoctxRange() { if (on) (void)roctx().pop(); }
};

struct ProfScope {
    int id; hipStream_t s; hipEvent_t a = nullptr, b = nullptr; bool on; RoctxRange range;
    ProfScope(int id_, hipStream_t s_) : id(id_), s(s_), on(((g_profile.load() >> id_) & 1) != 0 && (t_prof_off == 0 || id_ == K_CUT_REDO)), range(kKernelNames[id_])
    {
        if (on) { (void)hipEventCreate(&a); (void)hipEventCreate(&b); (void)hipEventRecord(a, s); }
    }
    ~ProfScope()
    {
        if (on) {
            (void)hipEventRecord(b, s);
            std::lock_guard<std::mutex> lk(g_prof_mu);
            g_pending.push_back({ id, a, b });
        }
    }
};

int post_launch(const char* what, hipStream_t s)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GSRAST_E_DEVICE, what, e);
    static const bool env_sync = getenv("GSRAST_DEBUG_SYNC") != nullptr;      // (diagnostics: localise a faulting launch)
    if (g_debug_sync.load() || env_sync) {
        if (env_sync) fprintf(stderr, "[gsrast] %s\n", what);
        e = hipStreamSynchronize(s);
        if (e != hipSuccess) return fail(GSRAST_E_DEVICE, what, e);
    }
    return GSRAST_OK;
}
#define GS_LAUNCHED(what)                                 \
    do {                                                  \
        int rc_ = post_launch(what, s);                   \
        if (rc_ != GSRAST_OK) return rc_;                 \
    } while (0)

// ---- scan / sort drivers --------------------------------------------------------------------
// dst[i] = scan of (idx ? src[idx[i]] : src[i]); two levels (single-block scan of block sums).
// `aux` (optional, n entries): only summed; its 64-bit total goes to aux_lo / aux_hi.
int scan_u32(const uint32_t* src, const uint32_t* idx, uint32_t n, uint32_t* dst, bool inclusive,
             uint32_t* tmp, uint32_t* total_out, hipStream_t s, const uint2* runs = nullptr,
             const uint32_t* aux = nullptr, uint32_t* aux_lo = nullptr, uint32_t* aux_hi = nullptr)
{
    if (n == 0) return GSRAST_OK;
    const uint32_t nb = (n + SC_CHUNK - 1) / SC_CHUNK;
    if (nb == 1 && !aux) {
        scan_apply_kernel<<<1, 256, 0, s>>>(src, idx, runs, n, nullptr, dst, inclusive ? 1 : 0, total_out);
        GS_LAUNCHED("scan_apply");
        return GSRAST_OK;
    }
    scan_block_sums_kernel<<<nb, 256, 0, s>>>(src, idx, runs, n, tmp, aux, tmp + nb);
    GS_LAUNCHED("scan_block_sums");
    scan_single_block_kernel<<<1, 256, 0, s>>>(tmp, nb, aux ? tmp + nb : nullptr, aux_lo, aux_hi);
    GS_LAUNCHED("scan_single_block");
    scan_apply_kernel<<<nb, 256, 0, s>>>(src, idx, runs, n, tmp, dst, inclusive ? 1 : 0, total_out);
    GS_LAUNCHED("scan_apply");
    return GSRAST_OK;
}

// Stable sort of n (key,value) pairs on the low `bits` key bits, in ceil(bits/8) passes of (nearly)
// equal digit width (13 bits -> 7 + 6: narrower digits mean fewer bins per block, i.e. longer
// contiguous runs in the scatter's write-out).  Result ends in (kA,vA) if the pass count is even,
// else in (kB,vB).
int radix_passes(int bits) { int p = (bits + 7) / 8; return p ? p : 1; }
// ITEMS = elements per lane: a workgroup sorts 256 * ITEMS consecutive elements.  The R-sized sort wants 16 (long
// contiguous runs in the write-out); the P- and Q-sized ones have too few elements to fill 256 CUs with 4096-element
// chunks (1 M Gaussians = 245 workgroups) and run faster with smaller ones.
// Device-side pass-count adaptation of the 32-bit depth sort (gsrast_binning.h: RA_*): a third buffer pair, the per-block
// key minima / maxima of pass 0, and the word that receives the number of significant key bits.
template <typename KeyT, typename ValT> struct SortAdapt { KeyT* kC; ValT* vC; uint32_t* block_minmax; uint32_t* sig; bool assume_short; };

template <typename KeyT, typename ValT = uint32_t, int ITEMS = RS_ITEMS>
int radix_sort(KeyT* kA, ValT* vA, KeyT* kB, ValT* vB, uint32_t n, int bits,
               uint32_t* hist, uint32_t* scan_tmp, hipStream_t s,
               const uint2* gather_rect = nullptr, uint32_t* gather_tiles = nullptr, uint32_t* gather_width = nullptr,
               const uint32_t* n_dev = nullptr /* n is a capacity, the real count is on the device (dev_count) */,
               const SortAdapt<KeyT, ValT>* ad = nullptr, const uint32_t* pred = nullptr /* predicated launch (list cut): only on the plain path below */)
{
    if (n == 0) return GSRAST_OK;
    const uint32_t nblk = (n + RS_THREADS * ITEMS - 1) / (RS_THREADS * ITEMS);
    const int passes = radix_passes(bits);
    if (ad && passes == 4 && bits == 32 && nblk > RS_SELF_SCAN_BLOCKS) {
        // A -> B -> (short ? C : A) -> (short ? A : B) -> [A]: see RA_* in gsrast_binning.h.  The result is in (kA, vA).
        const uint32_t mask = 255u;
        const int as = ad->assume_short ? RA_ASSUME : 0;
        for (int p = 0; p < (ad->assume_short ? 3 : 4); p++) {
            const int shift = 8 * p;
            const KeyT* kin = (p & 1) ? kB : kA; const ValT* vin = (p & 1) ? vB : vA;      // the long sort's ping-pong
            KeyT* kout = (p & 1) ? kA : kB; ValT* vout = (p & 1) ? vA : vB;
            const int hmode = (p == 0 ? RA_MINMAX : p == 2 ? RA_IN_ALT : p == 3 ? RA_SKIP : 0) | as;
            const int smode = (p == 1 ? RA_OUT_ALT : p == 2 ? (RA_IN_ALT | RA_OUT_ALT | RA_LAST_IF_SHORT) : p == 3 ? RA_SKIP : 0) | as;
            radix_hist_kernel<KeyT, ITEMS><<<nblk, RS_THREADS, 0, s>>>(kin, n, n_dev, shift, mask, hist, nblk, ad->kC, p ? ad->sig : nullptr, hmode, ad->block_minmax);
            GS_LAUNCHED("radix_hist");
            radix_rowscan_kernel<<<mask + 1 + (p == 0 ? 1 : 0), 256, 0, s>>>(hist, nblk, scan_tmp, ad->sig, (p == 0 ? RA_MINMAX : p == 3 ? RA_SKIP : 0) | as, ad->block_minmax, mask + 1);
            GS_LAUNCHED("radix_rowscan");
            // pass 1 writes C when short; pass 2 reads C and writes A when short (and gathers: it is then the last pass)
            radix_scatter_kernel<KeyT, ValT, ITEMS><<<nblk, RS_THREADS, 0, s>>>(kin, vin, kout, vout, n, n_dev, shift, mask, hist, scan_tmp, nblk,
                                                                         p >= 2 ? gather_rect : nullptr, gather_tiles, gather_width,
                                                                         ad->kC, ad->vC, p == 1 ? ad->kC : kA, p == 1 ? ad->vC : vA, ad->sig, smode);
            GS_LAUNCHED("radix_scatter");
        }
        return GSRAST_OK;
    }
    int shift = 0;
    for (int p = 0; p < passes; p++) {
        const int w = (bits - shift + (passes - p) - 1) / (passes - p);   // remaining bits spread evenly (7+6 == 6+7 measured)
        const uint32_t mask = (1u << w) - 1u;
        radix_hist_kernel<KeyT, ITEMS><<<nblk, RS_THREADS, 0, s>>>(kA, n, n_dev, shift, mask, hist, nblk, nullptr, nullptr, 0, nullptr, pred);
        GS_LAUNCHED("radix_hist");
        const bool self_scan = nblk <= RS_SELF_SCAN_BLOCKS;      // the scatter blocks sum the few block counts themselves
        if (!self_scan) {
            radix_rowscan_kernel<<<mask + 1, 256, 0, s>>>(hist, nblk, scan_tmp, nullptr, 0, nullptr, 0, pred);     // one workgroup per digit value in use
            GS_LAUNCHED("radix_rowscan");
        }
        const bool last = p == passes - 1;
        radix_scatter_kernel<KeyT, ValT, ITEMS><<<nblk, RS_THREADS, 0, s>>>(kA, vA, kB, vB, n, n_dev, shift, mask, hist,
                                                                     self_scan ? nullptr : scan_tmp, nblk,
                                                                     last ? gather_rect : nullptr, gather_tiles, gather_width,
                                                                     nullptr, nullptr, nullptr, nullptr, nullptr, 0, pred);
        GS_LAUNCHED("radix_scatter");
        std::swap(kA, kB); std::swap(vA, vB);
        shift += w;
    }
    return GSRAST_OK;
}

// Low-latency readback of one device word: async copy into pinned host memory, then spin on an event
// (hipStreamSynchronize may sleep; the GPU is idle while we wait, so every microsecond counts).
struct Readback {     // 64 pinned bytes + one event per (host thread, device); deliberately never freed: the destructor of a
    uint32_t* pinned = nullptr; hipEvent_t ev = nullptr;   // thread_local would call into the HIP runtime at thread / process exit,
    uint32_t* dev_alias = nullptr;                         // possibly after the runtime itself has been torn down
    uint32_t seq = 0;       // dev_alias: the same 64 bytes as the device sees them (a kernel stores the counts there itself: read_flag_*)
    uint32_t fb_seen = 0;   // list cut: sequence number of the last fallback this thread has taken note of (word RB_FALLBACK)
    bool prepared = false;  // read_flag_prepare has run for the forward in progress (the pose-found word is waited for before the counts)
};
// 64-bit words of the pinned buffer (128 bytes): 0-6 the counts of depth_bucket_totals, then
constexpr int RB_FOUND = 7;        // {1 = the pose had a slot in the context's table, sequence number}: preprocess_fwd, at its very start
constexpr int RB_FALLBACK = 8;     // {1, sequence number of the call}: the predicated second binning of a list-cut forward has run
constexpr size_t RB_BYTES = 128;
constexpr int kMaxDevices = 32;
thread_local Readback t_readback[kMaxDevices];     // one per (host thread, device): events belong to a device
// begin: enqueue the copy + event; finish: spin until it landed.  Work enqueued between the two runs on the GPU while
// the host waits (speculative launch in gsrast_forward).
int read_u32_begin(const uint32_t* dev, hipStream_t s, int nwords, Readback** handle)
{
    int device = 0;
    *handle = nullptr;
    GS_HIP(hipGetDevice(&device));
    if (device < 0 || device >= kMaxDevices) return GSRAST_OK;      // exotic topology: finish() does a blocking copy
    Readback& rb = t_readback[device];
    if (!rb.pinned) {
        GS_HIP(hipHostMalloc((void**)&rb.pinned, RB_BYTES, hipHostMallocPortable | hipHostMallocMapped));
        GS_HIP(hipEventCreateWithFlags(&rb.ev, hipEventDisableTiming));
        if (hipHostGetDevicePointer((void**)&rb.dev_alias, rb.pinned, 0) != hipSuccess) rb.dev_alias = nullptr;
        memset(rb.pinned, 0, RB_BYTES);
    }
    GS_HIP(hipMemcpyAsync(rb.pinned, dev, sizeof(uint32_t) * nwords, hipMemcpyDeviceToHost, s));
    GS_HIP(hipEventRecord(rb.ev, s));
    *handle = &rb;
    return GSRAST_OK;
}
int read_u32_finish(Readback* rb, const uint32_t* dev, hipStream_t s, uint32_t* out, int nwords)
{
    if (!rb) {
        GS_HIP(hipMemcpyAsync(out, dev, sizeof(uint32_t) * nwords, hipMemcpyDeviceToHost, s));
        GS_HIP(hipStreamSynchronize(s));
        return GSRAST_OK;
    }
    hipError_t e;
    while ((e = hipEventQuery(rb->ev)) == hipErrorNotReady) { }
    if (e != hipSuccess) return fail(GSRAST_E_DEVICE, "read_u32", e);
    for (int k = 0; k < nwords; k++) out[k] = rb->pinned[k];
    return GSRAST_OK;
}
// The read-back without a copy: the producing kernel stores the (12) words into the pinned buffer itself and then the sequence number
// into word 15 (depth_bucket_totals).  prepare: the buffer's device alias and this call's sequence number, or nullptr if mapped host
// memory is not to be had (the caller then uses the copy).  finish: spin on word 15; should it not arrive within two seconds the stream
// is drained and the counts are copied the slow way (a failed launch surfaces there).
Readback* read_flag_prepare(uint32_t** dev_alias, uint32_t* seq)
{
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess || device < 0 || device >= kMaxDevices) return nullptr;
    Readback& rb = t_readback[device];
    if (!rb.pinned) {
        if (hipHostMalloc((void**)&rb.pinned, RB_BYTES, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) { rb.pinned = nullptr; return nullptr; }
        if (hipEventCreateWithFlags(&rb.ev, hipEventDisableTiming) != hipSuccess) return nullptr;
        if (hipHostGetDevicePointer((void**)&rb.dev_alias, rb.pinned, 0) != hipSuccess) rb.dev_alias = nullptr;
        memset(rb.pinned, 0, RB_BYTES);
    }
    if (!rb.dev_alias) return nullptr;
    rb.seq = rb.seq + 1u ? rb.seq + 1u : 1u;
    for (int k = 0; k <= RB_FOUND; k++) reinterpret_cast<volatile unsigned long long*>(rb.pinned)[k] = 0ull;      // (no stale word may carry this number)
    std::atomic_thread_fence(std::memory_order_seq_cst);
    *dev_alias = rb.dev_alias; *seq = rb.seq;
    return &rb;
}
int read_flag_finish(Readback* rb, const uint32_t* dev, hipStream_t s, uint32_t* out, int nwords)
{
    // seven 64-bit words {value, sequence number} (depth_bucket_totals): each is valid as soon as its upper half carries this call's number
    volatile unsigned long long* p = reinterpret_cast<volatile unsigned long long*>(rb->pinned);
    const auto t0 = std::chrono::steady_clock::now();
    unsigned long long w[7];
    for (int k = 0; k < 7; k++) {
        for (uint64_t spins = 1; (uint32_t)((w[k] = p[k]) >> 32) != rb->seq; spins++) {
            if ((spins & 0xFFFFu) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
                GS_HIP(hipStreamSynchronize(s));
                if ((uint32_t)((w[k] = p[k]) >> 32) == rb->seq) break;
                GS_HIP(hipMemcpy(out, dev, sizeof(uint32_t) * nwords, hipMemcpyDeviceToHost));
                return GSRAST_OK;
            }
        }
    }
    for (int k = 0; k < nwords; k++) out[k] = 0u;
    out[0] = (uint32_t)w[0]; out[1] = (uint32_t)w[1]; out[3] = (uint32_t)w[3];       // {R low, Q, overflow verdict, R high}
    if (nwords > 11) { out[11] = (uint32_t)w[2]; out[SC_Q_EARLY] = (uint32_t)w[4]; out[SC_N_LATE] = (uint32_t)w[5]; out[SC_ZBINS] = (uint32_t)w[6]; }     // (list cut: early column runs, late Gaussians; the depth histogram's occupied bins)
    return GSRAST_OK;
}
// the pose-found word of preprocess_fwd (RB_FOUND): 1 / 0, or 0 ("size the launches as for an unknown pose": all column runs, the safe
// sizing) if it does not arrive within 200 ms
bool read_found(Readback* rb)
{
    volatile unsigned long long* p = reinterpret_cast<volatile unsigned long long*>(rb->pinned) + RB_FOUND;
    const auto t0 = std::chrono::steady_clock::now();
    unsigned long long w;
    for (uint64_t spins = 1; (uint32_t)((w = *p) >> 32) != rb->seq; spins++)
        if ((spins & 0xFFFFu) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) return false;
    return (uint32_t)w != 0u;
}
// has the device reported a completion pass of the list cut this thread has not taken note of yet (RB_FALLBACK)?  Returns the number
// of column runs of that pass's candidates (0: nothing new).
uint32_t take_fallback_event()
{
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess || device < 0 || device >= kMaxDevices) return 0u;
    Readback& rb = t_readback[device];
    if (!rb.pinned) return 0u;
    const unsigned long long w = reinterpret_cast<volatile unsigned long long*>(rb.pinned)[RB_FALLBACK];
    const uint32_t sq = (uint32_t)(w >> 32);
    if (sq == 0u || sq == rb.fb_seen) return 0u;
    rb.fb_seen = sq;
    return (uint32_t)w ? (uint32_t)w : 1u;
}
int read_u32(const uint32_t* dev, hipStream_t s, uint32_t* out, int nwords = 1)
{
    Readback* rb = nullptr;
    int rc = read_u32_begin(dev, s, nwords, &rb);
    if (rc != GSRAST_OK) return rc;
    return read_u32_finish(rb, dev, s, out, nwords);
}
// Instances of the previous forward call: the binning buffer is requested for 1.25x that many BEFORE
// the host waits for the real count, so the (Python) allocation callback runs while the GPU is still
// busy with preprocess / depth sort instead of in the idle gap after the readback.
// These hints (and the counts of the last call) belong to a gsrast_context: one per caller that renders a sequence of similar
// views.  The reference-shaped entry points use a context private to the calling host thread.
} // namespace
// WORD FORKS (round 5).  Forking work onto the side stream with an event costs the CALLER's stream ~7 us (tools/fork_probe.hip: A; record, other stream
// waits + kernel; B = +7.5 us against A; B -- the record is a barrier packet between A and B).  Where the caller's next kernel can say "I have
// started" itself -- thread 0 of its first workgroup stores a sequence number into a word of the context's -- the side stream waits for that word
// with hipStreamWaitValue32 instead (+0.0 us in the probe): everything in front of that kernel on the caller's stream has completed, which is all
// the event said.  The wait is enqueued AFTER the kernel that releases it (submission order, as for the completion pass's gate), one word and one
// counter per caller stream (two streams sharing a context must not release each other's waits).
struct ForkWord { uint32_t* word = nullptr; uint32_t seq = 0; };
struct SideStream { hipStream_t stream = nullptr; hipEvent_t fork = nullptr, join = nullptr, join2 = nullptr; bool can_wait = false; std::map<hipStream_t, ForkWord> words; };
// The list cut's completion pass OFF the critical path (round 4).  Its eleven predicated launches used to sit between the forward blend and
// whatever the caller enqueues next: 55-80 us of dependent-launch latency in the steady state, where every one of them returns at once.
// Now the blend's LAST workgroup (gsrast_blend.h, GateArgs) copies the blend's verdict into a word of the context's own (`pred`: the chain's predicate; the
// forward's buffers may be gone by the time the chain's no-ops run) and, if the verdict is "nothing to complete", releases the caller's
// stream at once (`done` = the call's sequence number, waited for with hipStreamWaitValue32); the chain runs on a second stream behind
// the blend and, if it had work to do, releases the caller's stream at its end.  Deadlock-free on in-order hardware queues however HIP
// maps streams onto them: everything the wait can be released by is SUBMITTED before the wait (gate kernel, event, chain, last the wait).
// The release / predicate words are a ring of GATE_RING slots indexed by the call's sequence number.  A slot may only be claimed again once
// every launch that can still read it has run: the host records an event behind each chain (`tail`) and claims sequence number n only
// if the chain of n - GATE_RING / 2 has completed (hipEventQuery, no wait) -- otherwise that call runs its completion pass inline on the
// caller's stream, the round-3 form, which needs no slot.  So at most GATE_RING / 2 chains are ever pending and no slot is rewritten
// under a chain (the gate stream has the lowest priority and nothing else bounded its backlog).
// A profiler that collects hardware counters SERIALISES the device's kernels: the caller's stream would sit in its wait while the chain that
// releases it cannot start (round 5: every rocprofv3 --pmc pass of bench.py hung until its timeout).  rocprofv3 announces counter collection
// in the environment of the process it launches; the completion pass then runs inline on the caller's stream, as with option chain_gate = 0.
static bool counter_collection_env()
{
    static const bool on = getenv("ROCPROF_COUNTERS") != nullptr || getenv("ROCPROF_COUNTER_GROUPS") != nullptr || getenv("GSRAST_SERIALIZED_KERNELS") != nullptr;
    return on;
}
constexpr uint32_t GATE_RING = 64;
struct ChainGate { hipStream_t stream = nullptr; hipEvent_t ev = nullptr; uint32_t* words = nullptr /* [64] done | [64] pred */; uint32_t seq = 0; bool failed = false;
                   hipEvent_t tail[GATE_RING] = {}; uint32_t tail_seq[GATE_RING] = {} /* sequence number whose chain the slot's event follows; 0 = none */; uint32_t inline_calls = 0 /* forwards that found the ring's older half still pending (diagnostic) */; };
__global__ void chain_done_kernel(const uint32_t* __restrict__ pred_copy, uint32_t* __restrict__ done, uint32_t seq)
{
    if (*pred_copy != 0u) __hip_atomic_store(done, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
struct gsrast_context {
    std::atomic<uint32_t> R_hint{0}, Q_hint{0}, last_R{0}, last_Q{0}, last_late{0}, last_Qe{0};
    std::atomic<uint32_t> Qe_hint{0};  // list cut: column runs of the early Gaussians in recent forwards (sizes the launches over the cut lists)
    std::atomic<uint32_t> Qe_hint_tau{0};  // ... of the forwards that ran under PREDICTED cut depths (they keep more: their own hint)
    CutPolicy pol;                     // list cut: when it is applied, paused, widened (gsrast_policy.h)
    // a HINT, nothing more: "the last forward whose view matrix lived at this device address found its pose in the table".  Callers keep
    // a camera's matrices in one tensor for its lifetime (scene/cameras.py:90-101), so a forward can tell BEFORE it launches anything
    // whether it will need predicted cut depths: a pose the table knows skips the opacity-mass histogram and its kernel.  A wrong guess
    // costs one forward its cut (an address reused for another camera: the snapshot finds no slot, nothing is predicted), never a result.
    std::unordered_map<const void*, uint8_t> pose_seen;
    // equalised depth buckets (gsrast_common.h): the key range the depth histogram's bins cover, learned from the previous forwards
    DepthRange zrange;                 // (gsrast_policy.h) klo / shift: the histogram's bins; khi: the range's un-rounded upper end (the predicted cut's bins span [klo, khi])
    std::atomic<uint32_t> last_prologue_ns{0};   // host time of the last forward from entry to the launch of its first kernel (diagnostic: tools/sync_probe.py)
    std::atomic<int> redo_count{0};   // forwards whose speculative launch did not fit and was repeated with exact sizes
    std::atomic<int> depth_short{0};  // the last forward's depth keys spanned < 2^24: the next one enqueues three sort passes, not four
    std::atomic<int> bucket_skip{0};  // > 0: a recent forward's bucket depth sort overflowed a bucket; that many forwards go straight to the radix sort
    std::atomic<int> bucket_backoff{0}, bucket_clean{0};   // length of the last such pause (doubles per overflow), bucket-sorted forwards without one since
    bool counted_streams = false;     // this context is one of g_stream_contexts (it owns a side stream)
    SideStream side[32];              // per device: the stream the colour kernel runs on beside the sort (created on first use)
    ChainGate gate[32];               // per device: the completion pass's own stream and release words (created on first use)
    uint32_t* tau_dev[32] = {}; size_t tau_dev_words[32] = {}; bool tau_dev_dirty[32] = {};      // per device: the predicted cut's opacity-mass table [TAU_COPIES][T][TAU_BINS] (ImgLayout::tau_hist's twin), kept ZERO between
                                                                                                   // forwards: filled by preprocess_fwd, read by tau_cut_kernel, zeroed again by the bucket sort behind it -- no memset launch per forward
    uint32_t* zhist_dev[32] = {};     // per device: the sampled depth histogram of the bucket depth sort [ZH_COPIES][ZH_BINS] -- filled by preprocess_fwd, read by the
                                      // scatter, zeroed again by the bucket sort behind it (no memset launch per forward).  Two forwards of one context in flight on two
                                      // streams mix their samples: the bucket map stays monotone whatever the histogram holds (gsrast_common.h), only the balance suffers
    struct Hints { HintTable* table = nullptr; uint32_t T = 0; uint64_t used = 0; } hints[32][4];   // (one table per image size in use, up to four: train / eval resolutions alternate)
    uint64_t hints_clock = 0;   // per device: launch-order hints of the forward blend (gsrast_common.h), device memory
    std::mutex mu;
};
namespace {
// The context's side stream on the current device (created on first use, lowest priority: its bandwidth-heavy kernels should fill
// the gaps the critical path leaves, not compete with it for compute units).  nullptr if it cannot be had.
// Word forks are used only while the library's calls do not OVERLAP in time and at most two contexts own streams (PyTorch's usual shape: the
// forward on the caller's thread, the backward on the autograd engine's thread, one after the other).  tests/test_gpu_gate.py's soak -- two threads
// submitting concurrently, completion passes for real -- hung in 3 of 28 runs with word forks forced on (tools/soak_stress.sh; both threads stuck in
// the forward's read-back, the device making no progress) and in 0 of 20 without them; the backward's late join alone: 0 of 14.  Every wait is
// still submitted behind its releaser, so this is not a dependency cycle; what grows with concurrent submitters is the number of queues sitting in
// a value wait at the same time (two gate waits + up to three side-stream waits there), and a queue that polls a word seems to keep its place
// on the command processor -- with enough of them the queue that holds a releaser is not scheduled.  One submitter at a time keeps it at two.
// Once two calls have been seen inside the library at the same time, or a third context has created its streams, the library forks with events for good.
static std::atomic<int> g_calls_inside{0}, g_stream_contexts{0};
static std::atomic<bool> g_concurrent_callers{false};
struct CallScope { CallScope() { if (g_calls_inside.fetch_add(1) > 0) g_concurrent_callers = true; } ~CallScope() { g_calls_inside.fetch_sub(1); } };
static bool single_host_thread()
{
    static const bool forced = getenv("GSRAST_FORCE_WORD_FORK") != nullptr;      // (experiments only: switches the rule off -- to reproduce the hang it avoids)
    return forced || (!g_concurrent_callers.load(std::memory_order_relaxed) && g_stream_contexts.load(std::memory_order_relaxed) <= 2);
}
SideStream* side_stream_of(gsrast_context* ctx)
{
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess || device < 0 || device >= 32) return nullptr;
    std::lock_guard<std::mutex> lk(ctx->mu);
    SideStream& x = ctx->side[device];
    if (!x.stream) {
        int prio_least = 0, prio_greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
        if (hipStreamCreateWithPriority(&x.stream, hipStreamNonBlocking, prio_least) != hipSuccess) { x.stream = nullptr; return nullptr; }
        if (hipEventCreateWithFlags(&x.fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&x.join, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&x.join2, hipEventDisableTiming) != hipSuccess) {
            if (x.fork) (void)hipEventDestroy(x.fork);          // all or nothing: the next call tries again
            if (x.join) (void)hipEventDestroy(x.join);
            if (x.join2) (void)hipEventDestroy(x.join2);
            (void)hipStreamDestroy(x.stream);
            x = SideStream{};
            return nullptr;
        }
        int can = 0;
        x.can_wait = hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, device) == hipSuccess && can != 0;
        if (!ctx->counted_streams) { ctx->counted_streams = true; g_stream_contexts++; }
    }
    return &x;
}
std::atomic<int> g_word_fork{getenv("GSRAST_WORD_FORK") ? 1 : 0};          // (GSRAST_WORD_FORK: experiments, tools/soak_stress.sh) 1: word forks where a kernel can signal its own start.  OPT-IN since round 6 (default 0: every fork is an event): the hang the
                                          // soak test showed with them (3 of 28 runs, two concurrent submitters) was never root-caused, and what they buy is 13 us of a
                                          // 1.1 ms step (1.2 %).  The reference's statics are callable from any number of threads (rasterizer.h:24-83); a drop-in may not
                                          // trade that for a percent.  With the option on, the rule of single_host_thread() still applies.
// TESTS ONLY (option "mutate"): a backward that is wrong on purpose, so that a test can show its tolerance would catch it (the
// full-size gradient bar let an all-zero dL/dsh pass).  bit 0: the blend backward of ONE tile -- the image's centre tile -- does not see the 64
// front-most entries of the tile's list (one staged batch dropped: the tile's range, its pixels' n_contrib and its tile_max are shifted by a
// small kernel in front of the blend backward; every other (pixel, Gaussian) pair gets exactly what it gets without the mutation).
// bit 1: the background term of dL/dalpha (backward.cu:531-534) is dropped (the blend backward is handed a zero background).
std::atomic<int> g_mutate{0};
std::atomic<int> g_binrec_cut{0} /* 1 (A/B): the 32-byte binning records are written under the list cut too */;
std::atomic<int> g_two_level{1} /* 1: the bucket scatter as two launches, coarse + refine (gsrast_binning.h; A/B switch) */, g_two_level_min_p{2500000} /* measured (kernel times, one box): 0.3 M 16.1 us in one launch against 15.8 + 5.7 in two, 1 M 41.0 against 36.6 + 9.1, 3 M 88.2 against 60.5 + 19.9: the second launch only pays where the scattered stores dominate */;
__global__ void mutate_drop_front_batch_kernel(uint2* ranges, uint32_t* n_contrib, uint32_t* tile_max, uint32_t tile, int W, int H, int gx)
{
    const uint32_t tx = tile % (uint32_t)gx, ty = tile / (uint32_t)gx;
    const uint32_t px = tx * TILE_X + (threadIdx.x & 15u), py = ty * TILE_Y + (threadIdx.x >> 4);
    const uint2 r = ranges[tile];
    const uint32_t drop = (r.y - r.x) < 64u ? (r.y - r.x) : 64u;
    __syncthreads();
    if (px < (uint32_t)W && py < (uint32_t)H) { const size_t pid = (size_t)W * py + px; const uint32_t c = n_contrib[pid]; n_contrib[pid] = c > drop ? c - drop : 0u; }
    if (threadIdx.x == 0) { ranges[tile] = make_uint2(r.x + drop, r.y); const uint32_t m = tile_max[tile]; tile_max[tile] = m > drop ? m - drop : 0u; }
}
// the caller stream's fork word and the next sequence number to signal, or {nullptr, 0}: fork with the event
static ForkWord fork_word_next(gsrast_context* ctx, SideStream* side, hipStream_t caller)
{
    if (!side || !side->can_wait || !g_word_fork.load() || counter_collection_env() || !single_host_thread()) return ForkWord{};
    std::lock_guard<std::mutex> lk(ctx->mu);
    ForkWord& w = side->words[caller];
    if (!w.word) {
        if (side->words.size() > 64) { side->words.erase(caller); return ForkWord{}; }      // (a caller that creates streams without end: events for it)
        uint32_t* p = nullptr;
        if (hipMalloc((void**)&p, 64) != hipSuccess) { side->words.erase(caller); return ForkWord{}; }
        if (hipMemset(p, 0, 64) != hipSuccess) { (void)hipFree(p); side->words.erase(caller); return ForkWord{}; }
        w.word = p; w.seq = 0;
    }
    if (w.seq >= 0xFFFFFFF0u) {      // (the comparison is >=: start over before the counter wraps -- nothing may still be waiting on the old values)
        if (hipStreamSynchronize(side->stream) != hipSuccess || hipStreamSynchronize(caller) != hipSuccess || hipMemset(w.word, 0, 64) != hipSuccess) return ForkWord{};
        w.seq = 0;
    }
    w.seq++;
    return w;
}
ChainGate* chain_gate_of(gsrast_context* ctx)
{
    int device = 0, can = 0;
    if (hipGetDevice(&device) != hipSuccess || device < 0 || device >= 32) return nullptr;
    std::lock_guard<std::mutex> lk(ctx->mu);
    ChainGate& g = ctx->gate[device];
    if (g.failed) return nullptr;
    if (!g.stream) {
        g.failed = true;                      // (until everything below has worked)
        if (hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, device) != hipSuccess || !can) return nullptr;
        int prio_least = 0, prio_greatest = 0;      // (lowest priority: the pass's no-ops run beside the backward and must not stand in its workgroups' way)
        (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
        if (hipMalloc((void**)&g.words, 128 * sizeof(uint32_t)) != hipSuccess) { g.words = nullptr; return nullptr; }
        if (hipMemset(g.words, 0, 128 * sizeof(uint32_t)) != hipSuccess ||
            hipStreamCreateWithPriority(&g.stream, hipStreamNonBlocking, prio_least) != hipSuccess) { (void)hipFree(g.words); g = ChainGate{}; g.failed = true; return nullptr; }
        if (hipEventCreateWithFlags(&g.ev, hipEventDisableTiming) != hipSuccess) { (void)hipStreamDestroy(g.stream); (void)hipFree(g.words); g = ChainGate{}; g.failed = true; return nullptr; }
        for (uint32_t k = 0; k < GATE_RING; k++)
            if (hipEventCreateWithFlags(&g.tail[k], hipEventDisableTiming) != hipSuccess) {
                for (uint32_t j = 0; j < k; j++) (void)hipEventDestroy(g.tail[j]);
                (void)hipEventDestroy(g.ev); (void)hipStreamDestroy(g.stream); (void)hipFree(g.words); g = ChainGate{}; g.failed = true; return nullptr;
            }
        g.failed = false;
    }
    return &g;
}
// The context's hint table on the current device for a T-tile image (allocated on first use, cleared when T changes); nullptr if
// it cannot be had -- the forward then orders its blend by list length, as a first-seen pose does.
HintTable* hints_of(gsrast_context* ctx, uint32_t T, hipStream_t s)
{
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess || device < 0 || device >= 32) return nullptr;
    std::lock_guard<std::mutex> lk(ctx->mu);
    // one table per tile count, up to four per device (train and eval resolutions, mixed camera sizes: each keeps its poses); a fifth
    // size takes over the least recently used table's place (rare: drain, then start over)
    gsrast_context::Hints* hp = nullptr;
    for (auto& c : ctx->hints[device]) if (c.table && c.T == T) hp = &c;
    if (!hp) {
        for (auto& c : ctx->hints[device]) if (!c.table) { hp = &c; break; }
        if (!hp) {
            hp = &ctx->hints[device][0];
            for (auto& c : ctx->hints[device]) if (c.used < hp->used) hp = &c;
            (void)hipStreamSynchronize(s); (void)hipFree(hp->table); hp->table = nullptr;
        }
    }
    auto& h = *hp;
    h.used = ++ctx->hints_clock;
    if (!h.table) {
        if (hipMalloc((void**)&h.table, hint_table_bytes(T)) != hipSuccess) { h.table = nullptr; return nullptr; }
        h.T = T;
        if (hipMemsetAsync(h.table, 0, hint_zcut_offset(T), s) != hipSuccess ||
            hipMemsetAsync(reinterpret_cast<char*>(h.table) + hint_zcut_offset(T), 0xFF, (size_t)HINT_SLOTS * T * 4, s) != hipSuccess) { (void)hipFree(h.table); h.table = nullptr; return nullptr; }
    }
    return h.table;
}
gsrast_context* thread_context()
{   // deliberately leaked at thread exit -- the host words AND the device pose tables it has grown (12.5 MB per image size at 1080p, include/gsrast.h):
    // see Readback above for why nothing here has a destructor; a host thread that renders and exits should use gsrast_context_create / _destroy
    thread_local gsrast_context* c = new gsrast_context();
    return c;
}

CamArgs make_cam(const float* view, const float* proj, const float* campos, float tanx, float tany,
                 float scale_mod, int W, int H)
{
    CamArgs c;
    c.view = view; c.proj = proj; c.campos = campos;
    c.tanx = tanx; c.tany = tany;
    c.fy = H / (2.0f * tany);   // reference rasterizer_impl.cu:222-223
    c.fx = W / (2.0f * tanx);
    c.scale_mod = scale_mod;
    c.W = W; c.H = H; c.gx = (W + TILE_X - 1) / TILE_X; c.gy = (H + TILE_Y - 1) / TILE_Y;
    return c;
}

template <typename T> T* at(char* base, size_t off) { return reinterpret_cast<T*>(base + off); }
template <typename T> const T* at(const char* base, size_t off) { return reinterpret_cast<const T*>(base + off); }

__global__ void __launch_bounds__(256)
export_geom_kernel(int P, const float4* __restrict__ rec0,
                   const float4* __restrict__ rec1, const float4* __restrict__ rec2,
                   const float* __restrict__ cov3D_in, const unsigned char* __restrict__ clamped_in,
                   const uint32_t* __restrict__ tiles_in, float* depths, float* means2D, float* cov3D,
                   float* conic_opacity, float* rgb, unsigned char* clamped, uint32_t* tiles)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const bool vis = tiles_in[i] != 0;
    const float4 a = rec0[(size_t)REC_STRIDE * i], b = rec1[(size_t)REC_STRIDE * i], c = rec2[(size_t)REC_STRIDE * i];
    if (depths) depths[i] = vis ? b.z : 0.0f;
    if (means2D) { means2D[2 * i] = vis ? a.x : 0.f; means2D[2 * i + 1] = vis ? a.y : 0.f; }
    if (cov3D) for (int k = 0; k < 6; k++) cov3D[6 * i + k] = cov3D_in[6 * i + k];
    if (conic_opacity) {
        conic_opacity[4 * i] = vis ? a.z : 0.f; conic_opacity[4 * i + 1] = vis ? a.w : 0.f;
        conic_opacity[4 * i + 2] = vis ? b.x : 0.f; conic_opacity[4 * i + 3] = vis ? b.y : 0.f;
    }
    if (rgb) { rgb[3 * i] = vis ? c.x : 0.f; rgb[3 * i + 1] = vis ? c.y : 0.f; rgb[3 * i + 2] = vis ? c.z : 0.f; }
    if (clamped) {
        const unsigned cl = vis ? clamped_in[i] : 0u;
        clamped[3 * i] = cl & 1u; clamped[3 * i + 1] = (cl >> 1) & 1u; clamped[3 * i + 2] = (cl >> 2) & 1u;
    }
    if (tiles) tiles[i] = tiles_in[i];
}

__global__ void __launch_bounds__(256) iota_kernel(uint32_t* __restrict__ v, uint32_t n)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) v[i] = i;
}

// one workgroup per tile: the 64-bit keys of the reference, rebuilt from the tile's range
__global__ void __launch_bounds__(256)
export_keys_kernel(const uint2* __restrict__ ranges, const uint32_t* __restrict__ vals_sorted,
                   const float4* __restrict__ rec1 /* .z = depth */, uint64_t* keys, uint32_t* point_list, uint32_t R /* entries of the two output arrays */)
{
    const uint2 r = ranges[blockIdx.x];
    // (a tile the list cut's completion pass listed again has its range in the point list's second half, past R: not exported -- the
    // entry-by-entry comparisons run with tile_clip = 0, i.e. without the cut)
    for (uint32_t i = r.x + threadIdx.x; i < r.y && i < R; i += 256) {
        const uint32_t g = vals_sorted[i];
        if (keys) keys[i] = ((uint64_t)blockIdx.x << 32) | (uint64_t)__float_as_uint(rec1[(size_t)REC_STRIDE * g].z);
        if (point_list) point_list[i] = g;
    }
}

std::atomic<int> g_hex_scatter{0};    // hexplane backward to the texels: 0 = sorted runs, 1 = direct global atomics
int pick_ppl(uint32_t ntiles, bool backward, const gsrast_options& o)
{
    const int forced = backward ? o.bwd_pixels_per_lane : o.fwd_pixels_per_lane;
    if (forced == 1 || forced == 2 || forced == 4) return forced;
    // Measured on MI355X (profiles/): the forward is fastest with one pixel per lane (finest cull /
    // early-exit granularity, most waves in flight).  With per-wave accumulator slices the backward is
    // within 2 % for 1, 2 and 4 pixels per lane at 1080p and above; fewer pixels per lane win for small
    // images (more waves) and small Gaussians (finer culling), more pixels per lane win at 4K.
    if (!backward) return 1;
    return ntiles >= 32768 ? 4 : (ntiles >= 8192 ? 2 : 1);
}

struct BlendArgs {
    uint32_t* bcnt = nullptr; uint16_t* blist = nullptr; int from_buckets = 0;   // launch order from the work buckets (ImgLayout)
    const uint2* ranges; const uint32_t* plist; const uint32_t* order; int W, H, gx; uint32_t T; const float4 *r0, *r1, *r2; const float* bg;
    float *oc, *od, *fT; uint32_t *nc, *tm;                       // forward outputs (fT / nc / tm: inputs of backward)
    const float* dpix; float* grec;                              // backward
    float4* zero4 = nullptr; uint32_t n_zero4 = 0;               // forward (culling kernel): the gradient records to zero-fill
    HintTable* hints = nullptr; const uint32_t* hint_sel = nullptr;   // forward: the context's launch-order hints, this call's slot
    const uint32_t* zcut_used = nullptr; uint32_t* cut_scalars = nullptr; const uint32_t* pred = nullptr;   // forward: list cut (gsrast_common.h)
    unsigned char* tile_flags = nullptr;                         // forward: tiles the completion pass lists and blends again
    GateArgs gate{};                                             // forward, list cut's first pass: the completion pass's gate (ChainGate)
    uint32_t cut_margin_x4 = 6;                                  // forward: the next cut depth's margin (gsrast_context::cut_margin)
    unsigned char* untouched = nullptr;                          // forward (culling kernel): GeomLayout::untouched
    uint32_t* fork_word = nullptr; uint32_t fork_seq = 0;        // backward (transposed kernel): the word fork's signal (SideStream)
};
template <int MODE, int PPL>
void launch_fwd(uint32_t grid, hipStream_t s, const BlendArgs& a)
{
    blend_fwd_kernel<MODE, PPL><<<grid, 256 / PPL, 0, s>>>(a.ranges, a.plist, a.W, a.H, a.gx, a.T, a.r0, a.r1, a.r2, a.bg, a.oc, a.od, a.fT, a.nc, a.tm, a.bcnt, a.blist);
}
template <int MODE, int PPL, int ABL>
void launch_bwd(uint32_t grid, hipStream_t s, const BlendArgs& a)
{
    blend_bwd_kernel<MODE, PPL, ABL><<<grid, 256 / PPL, 0, s>>>(a.ranges, a.plist, a.W, a.H, a.gx, a.T, a.r0, a.r1, a.r2, a.bg, a.fT, a.nc, a.tm, a.dpix, a.grec);
}
template <int MODE>
void dispatch_fwd(int ppl, uint32_t grid, hipStream_t s, const BlendArgs& a)
{
    if (ppl == 4) launch_fwd<MODE, 4>(grid, s, a); else if (ppl == 2) launch_fwd<MODE, 2>(grid, s, a); else launch_fwd<MODE, 1>(grid, s, a);
}
template <int MODE, int PPL>
void launch_bwd_cull(uint32_t grid, hipStream_t s, const BlendArgs& a)
{
    blend_bwd_cull_kernel<MODE, PPL><<<grid, 256 / PPL, 0, s>>>(a.ranges, a.plist, a.order, a.W, a.H, a.gx, a.T, a.r0, a.r1, a.r2, a.bg, a.fT, a.nc, a.tm, a.dpix, a.grec,
                                                                 a.from_buckets ? a.bcnt : nullptr, a.blist);
}
std::atomic<int> g_sort_hint{1};          // 1: enqueue three depth-sort passes when the context's last forward had short keys (A/B switch)
std::atomic<int> g_bwd_transposed{1};     // 1: blend_bwd_cull_t_kernel for one pixel per lane (default); 0: blend_bwd_cull_kernel<.., 1> (A/B switch)
template <int MODE>
void dispatch_bwd_cull(int ppl, uint32_t grid, hipStream_t s, const BlendArgs& a)
{
    if (ppl == 1 && g_bwd_transposed.load()) {
        blend_bwd_cull_t_kernel<MODE><<<grid, 256, 0, s>>>(a.ranges, a.plist, a.order, a.W, a.H, a.gx, a.T, a.r0, a.r1, a.r2, a.bg, a.fT, a.nc, a.tm, a.dpix, a.grec,
                                                           a.from_buckets ? a.bcnt : nullptr, a.blist, a.fork_word, a.fork_seq);
        return;
    }
    if (ppl == 4) launch_bwd_cull<MODE, 4>(grid, s, a); else if (ppl == 2) launch_bwd_cull<MODE, 2>(grid, s, a); else launch_bwd_cull<MODE, 1>(grid, s, a);
}
template <int MODE>
void launch_fwd_cull(uint32_t grid, hipStream_t s, const BlendArgs& a)
{
    blend_fwd_cull_kernel<MODE><<<grid, 256, 0, s>>>(a.ranges, a.plist, a.order, a.W, a.H, a.gx, a.T, a.r0, a.r1, a.r2, a.bg, a.oc, a.od, a.fT, a.nc, a.tm,
                                                     a.bcnt, a.blist, a.from_buckets, a.zero4, a.n_zero4, a.hints, a.hint_sel, a.zcut_used, a.cut_scalars, a.pred, a.tile_flags, a.gate, a.cut_margin_x4, a.untouched);
}
template <int MODE>
void dispatch_bwd(int ppl, uint32_t grid, hipStream_t s, const BlendArgs& a)
{
    if (ppl == 4) launch_bwd<MODE, 4, 0>(grid, s, a); else if (ppl == 2) launch_bwd<MODE, 2, 0>(grid, s, a); else launch_bwd<MODE, 1, 0>(grid, s, a);
}

} // namespace

extern "C" {

int gsrast_abi_version(void) { return GSRAST_ABI_VERSION; }
const char* gsrast_last_error(void) { return g_err.c_str(); }

void gsrast_options_init(gsrast_options* o)
{
    if (!o) return;
    memset(o, 0, sizeof *o);
    o->tile_clip = 1; o->cull = 1; o->lpt = 1; o->speculative = 1; o->side_stream = 1;
}
gsrast_context* gsrast_context_create(void) { return new (std::nothrow) gsrast_context(); }
void gsrast_context_destroy(gsrast_context* c)
{
    if (!c) return;
    for (auto& d : c->hints) for (auto& h : d) if (h.table) (void)hipFree(h.table);
    for (ChainGate& g : c->gate) {
        if (g.stream) { (void)hipStreamSynchronize(g.stream); (void)hipStreamDestroy(g.stream); }
        if (g.ev) (void)hipEventDestroy(g.ev);
        for (hipEvent_t e : g.tail) if (e) (void)hipEventDestroy(e);
        if (g.words) (void)hipFree(g.words);
    }
    for (uint32_t* z : c->zhist_dev) if (z) (void)hipFree(z);
    for (uint32_t* z : c->tau_dev) if (z) (void)hipFree(z);
    if (c->counted_streams) g_stream_contexts--;
    for (SideStream& x : c->side) {
        if (x.stream) { (void)hipStreamSynchronize(x.stream); (void)hipStreamDestroy(x.stream); }
        if (x.fork) (void)hipEventDestroy(x.fork);
        if (x.join) (void)hipEventDestroy(x.join);
        if (x.join2) (void)hipEventDestroy(x.join2);
        for (auto& kv : x.words) if (kv.second.word) (void)hipFree(kv.second.word);
    }
    delete c;
}
int gsrast_context_query(const gsrast_context* c, const char* name)
{
    if (!name) return GSRAST_E_ARG;
    if (!c) c = thread_context();
    if (!strcmp(name, "last_instances")) return (int)c->last_R.load();   // num_rendered / column runs of the context's last forward call
    if (!strcmp(name, "last_runs")) return (int)c->last_Q.load();
    if (!strcmp(name, "last_prologue_ns")) return (int)c->last_prologue_ns.load();
    if (!strcmp(name, "redo_count")) return c->redo_count.load();
    if (!strcmp(name, "bucket_skip")) return c->bucket_skip.load();
    if (!strcmp(name, "last_late")) return (int)c->last_late.load();
    if (!strcmp(name, "last_early_runs")) return (int)c->last_Qe.load();
    if (!strcmp(name, "completion_passes")) return (int)c->pol.passes_reported.load();
    if (!strcmp(name, "tau_req")) return c->pol.tau_req.load();
    if (!strcmp(name, "tau_force")) return c->pol.tau_force.load();
    if (!strcmp(name, "cut_margin_x4")) return c->pol.margin.load();      // the list cut's current margin, in quarters (6 = 1.5 x)
    if (!strcmp(name, "gate_inline_calls")) {     // cut forwards that ran their completion pass inline because the gate's ring was half full
        int device = 0; if (hipGetDevice(&device) != hipSuccess || device < 0 || device >= 32) return 0;
        return (int)c->gate[device].inline_calls;
    }
    if (!strcmp(name, "cut_pause")) return c->pol.pause.load();      // forwards the list cut still sits out (too little saved, or its lists kept failing)
    if (!strcmp(name, "cut_fallbacks")) {       // a device counter in the hint table of the current device (diagnostic: waits for the device)
        int device = 0;
        if (hipGetDevice(&device) != hipSuccess || device < 0 || device >= 32) return 0;
        if (hipDeviceSynchronize() != hipSuccess) return GSRAST_E_DEVICE;
        uint32_t sum = 0;                       // (over the device's tables: one per image size)
        for (const auto& h : c->hints[device]) {
            uint32_t v = 0;
            if (h.table && hipMemcpy(&v, &h.table->cut_fallbacks, sizeof v, hipMemcpyDeviceToHost) != hipSuccess) return GSRAST_E_DEVICE;
            sum += v;
        }
        return (int)sum;
    }
    return GSRAST_E_ARG;
}

int gsrast_policy_event(gsrast_context* c, const char* what, int a, int b, int c3)
{
    if (!what) return GSRAST_E_ARG;
    if (!c) c = thread_context();
    CutPolicy& pol = c->pol;
    if (!strcmp(what, "begin")) {
        pol.begin_forward((uint32_t)a);
        c->last_R = (uint32_t)a;       // (the P the "counts" / "pass" events below refer to)
        const bool pays = pol.pays((uint32_t)b, c3 != 0);
        if (!pays) pol.sits_out();
        return pays ? 1 : 0;
    }
    if (!strcmp(what, "counts")) { pol.forward_counts((c3 & 1) != 0, (c3 & 2) ? 1u : 0u, (uint32_t)a, (uint32_t)b, c->last_R.load(), false); return pol.pause.load(); }
    if (!strcmp(what, "pass")) return pol.completion_pass((uint32_t)a, (uint32_t)b, c->last_R.load(), false, c3 != 0);
    if (!strcmp(what, "clean")) { pol.clean_cut_forward(); return pol.fb_score.load(); }
    if (!strcmp(what, "size")) return (int)early_launch_runs((uint32_t)a, (uint32_t)b, c3 != 0);
    if (!strcmp(what, "grow")) return (int)grow_capacity((uint32_t)a);
    if (!strcmp(what, "follow")) return (int)follow_hint((uint32_t)a, (uint32_t)b, c3);
    if (!strcmp(what, "reset")) { c->pol.~CutPolicy(); new (&c->pol) CutPolicy(); return 0; }      // (tests: a fresh policy, whatever earlier calls left)
    if (!strcmp(what, "tau_min")) { pol.tau_min = std::max(1, a); pol.tau_req = std::max(1, a); return a; }      // (experiments: the predicted cut's requirement and its floor)
    if (!strcmp(what, "zrange")) {      // a = first | last << 16 occupied bin of a forward that used the context's current table -> the next table's shift (| 256 if that table had held every key)
        const uint32_t k = c->zrange.klo.load(); const int sh = c->zrange.shift.load();
        const bool held = c->zrange.learn((uint32_t)a, k, sh);
        return c->zrange.shift.load() | (held ? 256 : 0);
    }
    if (!strcmp(what, "zget")) return a == 0 ? (int)(c->zrange.klo.load() >> 8) : a == 1 ? c->zrange.shift.load() : (int)(c->zrange.khi.load() >> 8);      // (keys / 256: they do not fit an int)
    if (!strcmp(what, "get")) {
        switch (a) { case 0: return pol.pause.load(); case 1: return pol.fb_score.load(); case 2: return pol.margin.load(); case 3: return pol.tau_req.load();
                     case 4: return pol.tau_force.load(); case 5: return pol.fb_pause.load(); default: return GSRAST_E_ARG; }
    }
    return GSRAST_E_ARG;
}

int gsrast_set_option(const char* name, int value)
{
    if (!name) return GSRAST_E_ARG;
    if (!strcmp(name, "exp_mode")) { if (value < 0 || value > 2) return GSRAST_E_ARG; g_def.exp_mode = value; return 0; }
    if (!strcmp(name, "profile")) { g_profile = value; return 0; }  // bit k = time kernel id k; -1 = all
    if (!strcmp(name, "debug_sync")) { g_debug_sync = value ? 1 : 0; return 0; }
    if (!strcmp(name, "list_cut_always")) { g_list_cut_always = value ? 1 : 0; return 0; }   // the list cut also where it does not pay (tests)
    if (!strcmp(name, "chain_gate")) { g_chain_gate = value ? 1 : 0; return 0; }               // 0: the completion pass's launches on the caller's stream (round 3)
    if (!strcmp(name, "touch_bits")) { g_touch_bits = value ? 1 : 0; return 0; }               // 0: only the list cut's late bits serve the backward (round 4)
    if (!strcmp(name, "late_fill_min_p")) { g_late_fill_min_p = value < 0 ? 0 : value; return 0; }
    if (!strcmp(name, "sparse_grec")) { g_sparse_grec = value != 0; return 0; }
    if (!strcmp(name, "word_fork")) { g_word_fork = value != 0; return 0; }
    if (!strcmp(name, "near_pose")) { g_near_pose = value < 0 ? 0 : (value > 8 ? 8 : value); return 0; }                 // 0: only the pose's own slot (round 3)
    if (!strcmp(name, "tau_sample")) { g_tau_sample = value < 0 ? 0 : (value > 6 ? 6 : value); return 0; }
    if (!strcmp(name, "tau_cut")) { g_tau_cut = value ? 1 : 0; return 0; }                    // 0: only poses with remembered cut depths are cut (round 4's behaviour)
    if (!strcmp(name, "layer_cut")) { g_layer_cut = value ? 1 : 0; return 0; }                // 0: only poses with remembered cut depths are cut (round 3's behaviour)
    if (!strcmp(name, "debug_state")) { g_debug_state = value ? 1 : 0; return 0; }   // forwards also store what only gsrast_debug_export reads (cov3D)
    if (!strcmp(name, "ablate")) { g_ablate = value; return 0; }   // experiments only
    if (!strcmp(name, "binrec_cut")) { g_binrec_cut = value ? 1 : 0; return 0; }
    if (!strcmp(name, "two_level")) { g_two_level = value ? 1 : 0; return 0; }
    if (!strcmp(name, "two_level_min_p")) { g_two_level_min_p = value < 0 ? 0 : value; return 0; }
    if (!strcmp(name, "mutate")) { g_mutate = value; return 0; }   // tests only: a deliberately WRONG backward (see g_mutate) -- proves that a parity bar bites
    if (!strcmp(name, "bwd_transposed")) { g_bwd_transposed = value ? 1 : 0; return 0; }
    if (!strcmp(name, "sort_hint")) { g_sort_hint = value ? 1 : 0; return 0; }
    if (!strcmp(name, "cull")) { g_def.cull = value ? 1 : 0; return 0; }
    if (!strcmp(name, "binning")) { if (value != 0 && value != 1) return GSRAST_E_ARG; g_def.binning = value; return 0; }
    if (!strcmp(name, "tile_clip")) { g_def.tile_clip = value ? 1 : 0; return 0; }
    if (!strcmp(name, "sh_grad_factors")) { g_def.sh_grad_factors = value ? 1 : 0; return 0; }
    if (!strcmp(name, "speculative")) { g_def.speculative = value ? 1 : 0; return 0; }
    if (!strcmp(name, "side_stream")) { g_def.side_stream = value ? 1 : 0; return 0; }
    if (!strcmp(name, "depth_sort")) { if (value != 0 && value != 1) return GSRAST_E_ARG; g_def.depth_sort = value; return 0; }
    if (!strcmp(name, "forward_only")) { g_def.forward_only = value ? 1 : 0; return 0; }
    if (!strcmp(name, "no_order_hint")) { g_def.no_order_hint = value ? 1 : 0; return 0; }
    if (!strcmp(name, "dense_backward")) { g_def.dense_backward = value ? 1 : 0; return 0; }
    if (!strcmp(name, "no_list_cut")) { g_def.no_list_cut = value ? 1 : 0; return 0; }
    if (!strcmp(name, "lpt")) { g_def.lpt = value ? 1 : 0; return 0; }   // heaviest-tile-first launch order
    if (!strcmp(name, "hexplane_scatter")) { if (value != 0 && value != 1) return GSRAST_E_ARG; g_hex_scatter = value; return 0; }
    if (!strcmp(name, "pixels_per_lane") || !strcmp(name, "fwd_pixels_per_lane") || !strcmp(name, "bwd_pixels_per_lane")) {
        if (value != 0 && value != 1 && value != 2 && value != 4) return GSRAST_E_ARG;
        if (name[0] != 'b') g_def.fwd_ppl = value;
        if (name[0] != 'f') g_def.bwd_ppl = value;
        return 0;
    }
    return GSRAST_E_ARG;
}
int gsrast_get_option(const char* name)
{
    if (!name) return GSRAST_E_ARG;
    if (!strcmp(name, "exp_mode")) return g_def.exp_mode.load();
    if (!strcmp(name, "profile")) return g_profile.load();
    if (!strcmp(name, "bwd_transposed")) return g_bwd_transposed.load();
    if (!strcmp(name, "debug_sync")) return g_debug_sync.load();
    if (!strcmp(name, "list_cut_always")) return g_list_cut_always.load();
    if (!strcmp(name, "chain_gate")) return g_chain_gate.load();
    if (!strcmp(name, "touch_bits")) return g_touch_bits.load();
    if (!strcmp(name, "late_fill_min_p")) return g_late_fill_min_p.load();
    if (!strcmp(name, "sparse_grec")) return g_sparse_grec.load();
    if (!strcmp(name, "word_fork")) return g_word_fork.load();
    if (!strcmp(name, "mutate")) return g_mutate.load();
    if (!strcmp(name, "two_level")) return g_two_level.load();
    if (!strcmp(name, "two_level_min_p")) return g_two_level_min_p.load();
    if (!strcmp(name, "stream_contexts")) return g_stream_contexts.load();          // (diagnostics: the rule of the word forks)
    if (!strcmp(name, "concurrent_callers")) return g_concurrent_callers.load() ? 1 : 0;
    if (!strcmp(name, "near_pose")) return g_near_pose.load();
    if (!strcmp(name, "tau_sample")) return g_tau_sample.load();
    if (!strcmp(name, "tau_cut")) return g_tau_cut.load();
    if (!strcmp(name, "layer_cut")) return g_layer_cut.load();
    if (!strcmp(name, "debug_state")) return g_debug_state.load();
    if (!strcmp(name, "pixels_per_lane") || !strcmp(name, "fwd_pixels_per_lane")) return g_def.fwd_ppl.load();
    if (!strcmp(name, "bwd_pixels_per_lane")) return g_def.bwd_ppl.load();
    if (!strcmp(name, "cull")) return g_def.cull.load();
    if (!strcmp(name, "binning")) return g_def.binning.load();
    if (!strcmp(name, "tile_clip")) return g_def.tile_clip.load();
    if (!strcmp(name, "sh_grad_factors")) return g_def.sh_grad_factors.load();
    if (!strcmp(name, "last_instances") || !strcmp(name, "last_runs") || !strcmp(name, "redo_count") || !strcmp(name, "bucket_skip")) return gsrast_context_query(nullptr, name);
    if (!strcmp(name, "speculative")) return g_def.speculative.load();
    if (!strcmp(name, "side_stream")) return g_def.side_stream.load();
    if (!strcmp(name, "depth_sort")) return g_def.depth_sort.load();
    if (!strcmp(name, "forward_only")) return g_def.forward_only.load();
    if (!strcmp(name, "no_order_hint")) return g_def.no_order_hint.load();
    if (!strcmp(name, "dense_backward")) return g_def.dense_backward.load();
    if (!strcmp(name, "no_list_cut")) return g_def.no_list_cut.load();
    if (!strcmp(name, "lpt")) return g_def.lpt.load();
    if (!strcmp(name, "hexplane_scatter")) return g_hex_scatter.load();
    return GSRAST_E_ARG;
}

int gsrast_profile_kernel_count(void) { return K_COUNT; }
const char* gsrast_profile_kernel_name(int id) { return (id >= 0 && id < K_COUNT) ? kKernelNames[id] : ""; }
int gsrast_profile_collect(void)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& p : g_pending) {
        float ms = 0.f;
        if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            g_total_ms[p.id] += ms; g_launches[p.id] += 1;
        }
        (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b);
    }
    g_pending.clear();
    return 0;
}
int gsrast_profile_read(int id, double* total_ms, long long* launches)
{
    if (id < 0 || id >= K_COUNT) return GSRAST_E_ARG;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (total_ms) *total_ms = g_total_ms[id];
    if (launches) *launches = g_launches[id];
    return 0;
}
void gsrast_profile_reset(void)
{
    gsrast_profile_collect();
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (int i = 0; i < K_COUNT; i++) { g_total_ms[i] = 0; g_launches[i] = 0; }
}

size_t gsrast_geometry_bytes(int P) { return geom_layout((size_t)(P > 0 ? P : 0)).total; }
size_t gsrast_binning_bytes(int R, int, int) { return bin_layout((size_t)(R > 0 ? R : 0)).total; }
size_t gsrast_image_bytes(int W, int H) { return img_layout((size_t)(W > 0 ? W : 0), (size_t)(H > 0 ? H : 0)).total; }
void* gsrast_alloc_prealloc(void* ctx, size_t bytes)
{
    gsrast_prealloc* p = static_cast<gsrast_prealloc*>(ctx);
    if (!p) return nullptr;
    p->requested = bytes;
    return bytes <= p->capacity ? p->ptr : nullptr;
}

int gsrast_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                        unsigned char* present, void* stream)
{
    (void)projmatrix;
    hipStream_t s = (hipStream_t)stream;
    if (P < 0 || (P > 0 && (!means3D || !viewmatrix || !present))) return fail(GSRAST_E_ARG, "mark_visible: NULL argument");
    if (P == 0) return GSRAST_OK;
    ProfScope ps(K_MARK_VISIBLE, s);
    mark_visible_kernel<<<(P + 255) / 256, 256, 0, s>>>(P, means3D, viewmatrix, present);
    GS_LAUNCHED("mark_visible");
    return GSRAST_OK;
}

int gsrast_forward(gsrast_alloc_fn geometry_alloc, void* geometry_ctx, gsrast_alloc_fn binning_alloc,
                   void* binning_ctx, gsrast_alloc_fn image_alloc, void* image_ctx, int P, int D, int M,
                   const float* background, int width, int height, const float* means3D, const float* shs,
                   const float* colors_precomp, const float* opacities, const float* scales,
                   float scale_modifier, const float* rotations, const float* cov3D_precomp,
                   const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx,
                   float tan_fovy, int prefiltered, float* out_color, float* out_depth, int* radii, void* stream)
{
    return gsrast_forward_ex(nullptr, nullptr, geometry_alloc, geometry_ctx, binning_alloc, binning_ctx, image_alloc, image_ctx, P, D, M,
                             background, width, height, means3D, shs, colors_precomp, opacities, scales, scale_modifier, rotations,
                             cov3D_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, prefiltered, out_color, out_depth,
                             radii, stream);
}

static int prefilter_verdict(const uint32_t* word, hipStream_t s)
{
    uint32_t v = 0;
    GS_HIP(hipMemcpyAsync(&v, word, sizeof v, hipMemcpyDeviceToHost, s));
    GS_HIP(hipStreamSynchronize(s));
    return v ? fail(GSRAST_E_ARG, "forward: a Gaussian passed as prefiltered was culled by the near plane (the reference traps: auxiliary.h:156-160)") : GSRAST_OK;
}

// The forward behind gsrast_forward_ex (rawin == nullptr) and gsrast_forward_raw (rawin: means3D / opacities / scales / rotations
// are then the model's raw leaves, shs a non-null placeholder; the per-Gaussian kernels run as their RAW instantiations).
static int forward_impl(gsrast_context* ctx, const gsrast_options* options,
                      gsrast_alloc_fn geometry_alloc, void* geometry_ctx, gsrast_alloc_fn binning_alloc,
                      void* binning_ctx, gsrast_alloc_fn image_alloc, void* image_ctx, int P, int D, int M,
                      const float* background, int width, int height, const float* means3D, const float* shs,
                      const float* colors_precomp, const float* opacities, const float* scales,
                      float scale_modifier, const float* rotations, const float* cov3D_precomp,
                      const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx,
                      float tan_fovy, int prefiltered, float* out_color, float* out_depth, int* radii, void* stream,
                      const gsrast_raw_inputs* rawin)
{
    const auto t_entry = std::chrono::steady_clock::now();
    RoctxRange range_fwd(rawin ? "gsrast_forward_raw" : "gsrast_forward");
    CallScope call_scope;
    RawArgs raw{};
    if (rawin) {
        raw.motion_res = rawin->motion_res; raw.rot_res = rawin->rot_res; raw.trbf = rawin->trbf; raw.opacity_logit = rawin->opacity_logit;
        raw.features_dc = rawin->features_dc; raw.features_rest = rawin->features_rest; raw.shs_res = rawin->shs_res;
    }
    const gsrast_options o = options ? *options : snapshot_defaults();
    if (!options_valid(o)) return fail(GSRAST_E_ARG, "forward: bad option value");
    if (!ctx) ctx = thread_context();
    hipStream_t s = (hipStream_t)stream;
    const int W = width, H = height;
    if (P < 0 || W <= 0 || H <= 0 || D < 0 || D > 3) return fail(GSRAST_E_ARG, "forward: bad P / image size / SH degree");
    if (!geometry_alloc || !binning_alloc || !image_alloc) return fail(GSRAST_E_ARG, "forward: NULL allocator");
    if (!out_color || !out_depth || !background) return fail(GSRAST_E_ARG, "forward: NULL output / background");
    const size_t N = (size_t)W * H;
    if (P == 0) { // reference rasterize_points.cu:81 -- nothing is rendered, outputs stay zero
        GS_HIP(hipMemsetAsync(out_color, 0, 3 * N * sizeof(float), s));
        GS_HIP(hipMemsetAsync(out_depth, 0, N * sizeof(float), s));
        return 0;
    }
    if (!means3D || !opacities || !viewmatrix || !projmatrix || !radii) return fail(GSRAST_E_ARG, "forward: NULL required input");
    if (!shs && !colors_precomp) return fail(GSRAST_E_ARG, "forward: need shs or colors_precomp");
    if (!cov3D_precomp && (!scales || !rotations)) return fail(GSRAST_E_ARG, "forward: need scales+rotations or cov3D_precomp");
    if (shs && !colors_precomp && (!cam_pos || M < (D + 1) * (D + 1))) return fail(GSRAST_E_ARG, "forward: SH path needs campos and M >= (D+1)^2");

    const CamArgs cam = make_cam(viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, scale_modifier, W, H);
    const uint32_t T = (uint32_t)cam.gx * (uint32_t)cam.gy;

    const GeomLayout GL = geom_layout((size_t)P);
    char* geom = (char*)geometry_alloc(geometry_ctx, GL.total);
    if (!geom) return fail(GSRAST_E_ALLOC, "forward: geometry allocation failed");
    const ImgLayout IL = img_layout((size_t)W, (size_t)H);
    char* img = (char*)image_alloc(image_ctx, IL.total);
    if (!img) return fail(GSRAST_E_ALLOC, "forward: image allocation failed");

    float4* rec0 = at<float4>(geom, GL.rec0); float4* rec1 = at<float4>(geom, GL.rec1); float4* rec2 = at<float4>(geom, GL.rec2);
    uint32_t* tiles = at<uint32_t>(geom, GL.tiles);
    uint2* rect = at<uint2>(geom, GL.rect);
    uint32_t *kA = at<uint32_t>(geom, GL.keyA), *kB = at<uint32_t>(geom, GL.keyB);
    uint32_t *vA = at<uint32_t>(geom, GL.valA), *vB = at<uint32_t>(geom, GL.valB);
    uint32_t* offsets = at<uint32_t>(geom, GL.offsets);
    uint32_t* woffsets = at<uint32_t>(geom, GL.woffsets);
    uint32_t* hist = at<uint32_t>(geom, GL.hist);
    uint32_t* scan_tmp = at<uint32_t>(geom, GL.scan_tmp);
    uint32_t* scalars = at<uint32_t>(geom, GL.scalars);
    // `prefiltered` is the caller's promise that no Gaussian will be culled by the near plane; the reference prints and traps when one is
    // (auxiliary.h:156-160).  Here preprocess_fwd raises a word and this call returns GSRAST_E_ARG after it has waited for the device
    // (SaRO-GS always passes False, renderer/__init__.py:63: the promise costs a synchronisation, nobody makes it on a hot path).
    uint32_t* prefilter_word = prefiltered ? scalars + SC_PREFILTER : nullptr;
    if (prefilter_word) GS_HIP(hipMemsetAsync(prefilter_word, 0, sizeof(uint32_t), s));
    // near-pose borrowing (HintTable::cam): the position tolerance is relative to the camera's distance from the scene = the middle of the
    // depth range the context has learned (0: nothing learned yet, the kernel falls back to the distance from the origin)
    float near_scale2 = 0.0f;
    {   const uint32_t klo = ctx->zrange.klo.load(); const int sh = ctx->zrange.shift.load();
        if (!(klo == ZH_KLO_DEFAULT && sh == ZH_SHIFT_DEFAULT)) {
            const uint64_t kmid = (uint64_t)klo + (((uint64_t)ZH_MID << sh) >> 1);
            if (kmid < (uint64_t)ZH_KEY_TOP) { const uint32_t kb = (uint32_t)kmid; float z; memcpy(&z, &kb, sizeof z); if (z > 0.0f && z < 1e18f) near_scale2 = z * z; }
        } }
    // Run-compressed binning needs one 8-bit pass over tile rows and 16-bit tile ids.
    const bool runbin = o.binning == 0 && cam.gy <= 256 && T <= 65536u;
    const bool buckets_ok = T <= BUCKET_MAX_TILES;      // launch order of the blend kernels from work buckets (u16 tile ids)
    // Depth order of the Gaussians: bucket sort (two launches, gsrast_binning.h) unless the caller or the context's recent history
    // says radix sort
    const bool bucket_sort = runbin && o.depth_sort == 0 && (size_t)P >= BUCKET_SORT_MIN_P && ctx->bucket_skip.load() == 0;
    const uint32_t nbk = depth_buckets_host((size_t)P);
    // equalised depth buckets: the histogram's bins (range hint of this context) and which waves of preprocess_fwd are sampled (512-1024 of them)
    const uint32_t zh_klo = ctx->zrange.klo.load(); const int zh_shift = ctx->zrange.shift.load();
    uint32_t zh_wave_mask = 0u;
    while ((((size_t)P + 63) / 64) / ((size_t)zh_wave_mask + 1) > 1024) zh_wave_mask = 2u * zh_wave_mask + 1u;
    // Colour half of the per-Gaussian forward (SH -> RGB: most of its bytes) on the context's side stream, forked off the
    // caller's stream here and joined in front of the blend: it overlaps the geometry kernel, the depth sort and the binning.
    SideStream* side = nullptr;
    hipStream_t cs = s;
    if (o.side_stream) side = side_stream_of(ctx);
    // Where it forks matters little: its 200 MB of traffic stretches whatever latency-bound kernel runs beside it by about as much as
    // it hides (measured: beside the geometry kernel + depth sort +55 us, beside the run emission + run sort +45 / +65 us, beside the
    // LDS-bound run_scatter_rows it starves itself and delays the blend) -- round 2 forked it at entry, the best of those by ~20 us.
    // Round 3 (colour kernel now 162 us alone at 3 M, geometry kernel 68): it forks behind the depth sort -- see there.
    // The 64 B / Gaussian zero-fill of the backward's gradient records follows on the side stream, under the VALU-bound forward blend.
    // launch-order hints of the forward blend: per context, device and camera pose (gsrast_common.h); only with the work-bucket order
    HintTable* hints = (runbin && buckets_ok && o.cull != 0 && o.lpt != 0 && o.fwd_pixels_per_lane == 0 && !o.no_order_hint) ? hints_of(ctx, T, s) : nullptr;
    uint32_t* hint_sel = scalars + HINT_SEL;
    // List cut (gsrast_common.h): only with the hints, the bucket depth sort, clipped lists, and a capacity hint (the cut lists are
    // blended by the speculative launch; the verified fallback is enqueued behind it).  Whether THIS pose has cut depths is decided on the device.
    // It costs ~80 us per forward (the late test in the scatter, the compacting colour kernel, ten predicated launches behind the blend)
    // and saves ~50 us per million column runs it removes: it is used when the context's last forward had at least CUT_MIN_RUNS column
    // runs, and paused for CUT_PAUSE forwards whenever a cut forward removed fewer than that (a surface-like scene; measured
    // 1 M-Gaussian shell -7 %, 0.3 M cube -3 %, 0.1 M cube -8 % with the cut forced on; 1 M cube +8 %, 3 M cube +16 %).
    CutPolicy& pol = ctx->pol;          // (the decisions: gsrast_policy.h)
    if (pol.begin_forward((uint32_t)P)) {     // (what was learned belongs to the scene it was learned on: a context that moves on to a scene of another size starts afresh)
        // keys, stamps, clock: every slot free again, ordered in front of this call's lookup -- of EVERY image size's table of this device (round 6,
        // the other resolutions' tables kept the old scene's cut depths, running maxima that take eight visits to fade)
        int device = 0;
        std::lock_guard<std::mutex> lk(ctx->mu);
        if (hints && hipGetDevice(&device) == hipSuccess && device >= 0 && device < 32)
            for (auto& c : ctx->hints[device]) if (c.table) GS_HIP(hipMemsetAsync(c.table, 0, offsetof(HintTable, cam), s));
        ctx->pose_seen.clear(); ctx->Qe_hint = 0; ctx->Qe_hint_tau = 0;
    }
    const bool cut_pays = pol.pays(ctx->last_Q.load(), g_list_cut_always.load() != 0);
    const int cut_cs = cut_cell_shift((size_t)cam.gx, (size_t)cam.gy);
    // Round 4: the cut no longer needs the pose table.  With remembered cut depths (a pose the table knows) the first pass lists what lies
    // in front of them; WITHOUT -- a first-seen pose, the table switched off -- it lists the nearest eighth of the Gaussians (one cut
    // depth for every tile: LAYER mode, depth_bucket_scatter_kernel), and the completion pass behind the blend lists the rest into the
    // tiles that did not saturate inside that layer.  Either way the result is exact, and a pause (the passes did not pay) stops both.
    const bool cut_base = runbin && buckets_ok && o.cull != 0 && o.lpt != 0 && o.fwd_pixels_per_lane == 0 && bucket_sort && o.tile_clip != 0 && !o.no_list_cut &&
                          cut_cs != 0 && o.speculative != 0 && ctx->R_hint.load() != 0 && cut_pays;
    const bool tau_mode = g_tau_cut.load() != 0 && g_layer_cut.load() == 0;
    const bool cut = cut_base && (hints != nullptr || g_layer_cut.load() != 0 || tau_mode);
    // the 32-byte binning records (everything the run emission needs of a Gaussian in one line) are written where the emission gathers
    // EVERY visible Gaussian; under the list cut it gathers one in eight, from the blend's records, and preprocess_fwd writes 96 MB less at 3 M
    float4* binrec_p = (cut && g_binrec_cut.load() == 0) ? nullptr : at<float4>(geom, GL.binrec);
    const int layer_mode = !cut || g_layer_cut.load() == 0 ? 0 : (hints ? 1 : 2);
    if (!cut && !o.no_list_cut) pol.sits_out();
    const int tau_forced = cut && tau_mode && pol.forced_prediction() ? 1 : 0;
    bool pose_expected = false;         // (see gsrast_context::pose_seen)
    if (cut && tau_mode && hints && !tau_forced) { std::lock_guard<std::mutex> lk(ctx->mu); auto it = ctx->pose_seen.find(viewmatrix); pose_expected = it != ctx->pose_seen.end() && it->second != 0; }
    const bool tau_on = cut && tau_mode && !pose_expected;
    uint32_t* tau_hist = tau_on ? at<uint32_t>(img, IL.tau_hist) : nullptr;
    TauBins tau_bins{0u, 0.0f, 0.0f, (1u << g_tau_sample.load()) - 1u};      // (scale 0: the context has not learned a depth range yet -- no prediction in this call)
    {   const uint32_t klo = ctx->zrange.klo.load(), khi = ctx->zrange.khi.load();
        if (tau_on && khi > klo + (uint32_t)TAU_BINS) { tau_bins.lo = klo; tau_bins.scale = (float)TAU_BINS / (float)(khi - klo); tau_bins.inv_scale = (float)(khi - klo) / (float)TAU_BINS; } }
    uint32_t* zcut_used = cut ? at<uint32_t>(img, IL.zcut_used) : nullptr;
    // The backward's gradient records (64 B / Gaussian) are zero-filled by the forward: inside the default (culling) blend kernel; by a
    // memset behind the colour kernel (side stream) / by the colour kernel itself (no side stream) when another blend kernel runs.
    const bool zero_in_blend = o.cull != 0 && o.fwd_pixels_per_lane == 0 && (size_t)P * 4 <= 0xFFFFFFFFull;
    // ... and it keeps the "no pixel consumed this Gaussian" bits for the backward (GeomLayout::untouched), unless no backward will follow
    unsigned char* untouched = (o.cull != 0 && o.fwd_pixels_per_lane == 0 && !o.forward_only && g_touch_bits.load() != 0) ? at<unsigned char>(geom, GL.untouched) : nullptr;
    // ... and then only the records of the Gaussians somebody consumed are zeroed, by a kernel of their own behind the last blend (gsrast_preprocess.h:
    // grec_zero_touched_kernel) instead of all P records from inside the blend
    const bool zero_touched = untouched != nullptr && zero_in_blend && g_sparse_grec.load() != 0;
    auto finish_records = [&]() -> int {
        if (!zero_touched) return GSRAST_OK;
        ProfScope ps(K_GREC_ZERO, s);
        grec_zero_touched_kernel<<<(unsigned)(((size_t)P + 256 * GZ_PER - 1) / (256 * GZ_PER)), 256, 0, s>>>(P, untouched, at<float4>(geom, GL.grec), scalars);
        GS_LAUNCHED("grec_zero_touched");
        return GSRAST_OK;
    };
    bool color_launched = false;
    // Every exit after the fork must order the caller's stream behind the side stream: the colour kernel and the zero-fill write
    // into the geometry buffer, which the caller is free to release (on `s`) as soon as this function has returned -- an error
    // return (allocation failure, overflow, a failed launch) included: those drain the side stream.  The normal path enqueues the
    // two waits itself (in front of / behind the blend) and disarms the guard.
    struct SideJoinGuard {
        SideStream*& side; hipStream_t s; bool& launched; bool joined = false;
        ~SideJoinGuard() { if (launched && side && !joined) (void)hipStreamSynchronize(side->stream); }    // error path: cost is irrelevant
    } side_guard{ side, s, color_launched };
    // the colour kernel itself.  List cut: early_only = only the Gaussians the bucket scatter did not mark culled or late (the compacting
    // kernel); pred = the predicated launch over ALL Gaussians in front of the second blend
    auto color_kernels = [&](hipStream_t cs, bool early_only, const uint32_t* pred) -> int {
        {
            ProfScope ps(K_COLOR, cs);
            // d(colour)/d(view direction) for the backward (36 B / Gaussian), unless the caller said that no backward will follow
            const bool want_shd = shs && !colors_precomp && D > 0 && !o.forward_only;
            float4* sA = want_shd ? at<float4>(geom, GL.shdA) : nullptr;
            float4* sB = want_shd ? at<float4>(geom, GL.shdB) : nullptr;
            float* sC = want_shd ? at<float>(geom, GL.shdC) : nullptr;
            const float* sh_in = colors_precomp ? nullptr : shs;
            unsigned char* cl = at<unsigned char>(geom, GL.clamped);
            float4* gz = (side || zero_in_blend) ? nullptr : at<float4>(geom, GL.grec);
            const bool staged = sh_in && M * 3 <= PP_SH_MAX && ((M * 3) & 3) == 0 && ((uintptr_t)sh_in & 15) == 0;
            const int grid = (P + PP_THREADS - 1) / PP_THREADS;
            if (early_only) {       // list cut: Gaussians the bucket scatter found culled or late are skipped (gsrast_preprocess.h)
                // (with `pred`: the completion pass -- the colours of the late Gaussians that are listed after all, GeomLayout::skip2)
                const unsigned char* skip = at<unsigned char>(geom, pred ? GL.skip2 : GL.color_skip);
                const int cgrid = (P + PCC_IDS - 1) / PCC_IDS;
                if (rawin) preprocess_color_compact_kernel<true><<<cgrid, 64 * PCC_WAVES, 0, cs>>>(P, D, M, means3D, nullptr, nullptr, raw, cam_pos, rec2, cl, sA, sB, sC, skip, pred);
                else preprocess_color_compact_kernel<false><<<cgrid, 64 * PCC_WAVES, 0, cs>>>(P, D, M, means3D, sh_in, colors_precomp, raw, cam_pos, rec2, cl, sA, sB, sC, skip, pred);
            } else if (rawin) {        // (gsrast_forward_raw has checked M and the alignment of the three SH arrays)
                if (M * 3 == PP_SH_MAX) preprocess_color_kernel<PP_SH_MAX, true><<<grid, PP_THREADS, 0, cs>>>(P, D, M, means3D, nullptr, raw, cam_pos, rec2, cl, gz, sA, sB, sC, pred);
                else preprocess_color_kernel<0, true><<<grid, PP_THREADS, 0, cs>>>(P, D, M, means3D, nullptr, raw, cam_pos, rec2, cl, gz, sA, sB, sC, pred);
            } else if (staged && M * 3 == PP_SH_MAX)
                preprocess_color_kernel<PP_SH_MAX, false><<<grid, PP_THREADS, 0, cs>>>(P, D, M, means3D, sh_in, raw, cam_pos, rec2, cl, gz, sA, sB, sC, pred);
            else if (staged)
                preprocess_color_kernel<0, false><<<grid, PP_THREADS, 0, cs>>>(P, D, M, means3D, sh_in, raw, cam_pos, rec2, cl, gz, sA, sB, sC, pred);
            else
                preprocess_color_direct_kernel<<<(P + 255) / 256, 256, 0, cs>>>(P, D, M, means3D, sh_in, colors_precomp, cam_pos, rec2, cl, gz, sA, sB, sC, pred);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return fail(GSRAST_E_DEVICE, "preprocess_color", e);
        }
        return GSRAST_OK;
    };
    // List cut: a Gaussian the bucket scatter found late is in no list -- its colour is not evaluated (3 M cube: 87 % of them).
    // Not under the diagnostic option debug_state (gsrast_debug_export shows every Gaussian's colour).
    bool cut_colors = cut && zero_in_blend && !g_debug_state.load();       // (switched off below for a pose the table does not know: everything is early)
    // WORD FORK (see SideStream): with the run emission about to be launched, the colour kernels are not forked with an event here but
    // enqueued behind that launch, waiting for the word the emission kernel stores when it starts (color_fork: pending until then)
    ForkWord color_fork{};
    auto launch_color = [&](bool emit_follows = false) -> int {
        if (color_launched) return GSRAST_OK;
        color_launched = true;
        if (side && emit_follows) { color_fork = fork_word_next(ctx, side, s); if (color_fork.word) return GSRAST_OK; }
        if (side) {
            GS_HIP(hipEventRecord(side->fork, s));             // the inputs (and the buffers just handed out) are ordered on s
            GS_HIP(hipStreamWaitEvent(side->stream, side->fork, 0));
            cs = side->stream;
        }
        { int rc = color_kernels(cs, cut_colors, nullptr); if (rc != GSRAST_OK) return rc; }
        if (side) {
            GS_HIP(hipEventRecord(side->join, side->stream));
            if (!zero_in_blend) {
                GS_HIP(hipMemsetAsync(at<float>(geom, GL.grec), 0, (size_t)P * GREC * sizeof(float), side->stream));
                GS_HIP(hipEventRecord(side->join2, side->stream));
            }
        }
        return GSRAST_OK;
    };
    // the pending word fork's side-stream half: `signalled` = the kernel that stores the word has been launched on s; otherwise s stores it itself
    auto flush_color_fork = [&](bool signalled) -> int {
        if (!color_fork.word) return GSRAST_OK;
        const ForkWord fw = color_fork;
        color_fork = ForkWord{};
        if (!signalled) GS_HIP(hipStreamWriteValue32(s, fw.word, fw.seq, 0));
        GS_HIP(hipStreamWaitValue32(side->stream, fw.word, fw.seq, hipStreamWaitValueGte, 0xFFFFFFFFu));
        cs = side->stream;
        { int rc = color_kernels(cs, cut_colors, nullptr); if (rc != GSRAST_OK) return rc; }
        GS_HIP(hipEventRecord(side->join, side->stream));
        if (!zero_in_blend) {
            GS_HIP(hipMemsetAsync(at<float>(geom, GL.grec), 0, (size_t)P * GREC * sizeof(float), side->stream));
            GS_HIP(hipEventRecord(side->join2, side->stream));
        }
        return GSRAST_OK;
    };
    // list cut: the device says at once whether the pose has a slot in the table (RB_FOUND); the launches over the cut lists of a pose
    // that has none -- everything is early -- are sized for all column runs, not for the early runs of recent forwards (a launch
    // that turned out too small cost a first-seen pose a whole second forward: 1.35 instead of 1.05 ms forward-only at 3 M)
    int tau_ctx_device = -1;             // >= 0: tau_hist is the context's table on that device (gsrast_context::tau_dev)
    uint32_t* zhist_call = nullptr; bool zhist_ctx = false;      // the depth histogram this call fills and reads: the context's (zeroed by the bucket sort) or the call's own
    Readback* rb_pre = nullptr; uint32_t* pre_alias = nullptr; uint32_t pre_seq = 0;
    if (cut) rb_pre = read_flag_prepare(&pre_alias, &pre_seq);
    unsigned long long* host_found = rb_pre ? reinterpret_cast<unsigned long long*>(pre_alias) + RB_FOUND : nullptr;
    {
        ProfScope ps(K_PREPROCESS_FWD, s);
        const int pf_grid = (P + PF_THREADS - 1) / PF_THREADS;
        float* cov_dbg = g_debug_state.load() ? at<float>(geom, GL.cov3D) : nullptr;     // 24 B / Gaussian nobody but gsrast_debug_export reads
        const int clip = (runbin && o.tile_clip) ? 1 : 0;
        // equalised depth buckets (gsrast_common.h): the sampled depth histogram, zeroed in front of the kernel that fills it
        uint32_t* zr = nullptr;
        if (bucket_sort) {
            int device = 0;
            if (hipGetDevice(&device) == hipSuccess && device >= 0 && device < 32) {
                std::lock_guard<std::mutex> lk(ctx->mu);
                if (!ctx->zhist_dev[device]) {
                    uint32_t* z = nullptr;
                    if (hipMalloc((void**)&z, ZH_COPIES * ZH_BINS * sizeof(uint32_t)) == hipSuccess) {
                        if (hipMemset(z, 0, ZH_COPIES * ZH_BINS * sizeof(uint32_t)) == hipSuccess) ctx->zhist_dev[device] = z; else (void)hipFree(z);
                    }
                }
                zr = ctx->zhist_dev[device];
            }
            zhist_ctx = zr != nullptr;
            if (!zr) {      // (no context memory to be had: the call's own table and a memset, as before)
                zr = at<uint32_t>(geom, GL.zhist);
                GS_HIP(hipMemsetAsync(zr, 0, ZH_COPIES * ZH_BINS * sizeof(uint32_t), s));
            }
        }
        zhist_call = zr;
        if (tau_hist) {
            // the context's table where the bucket sort will zero it again behind tau_cut_kernel; else the call's own and a memset
            const size_t words = (size_t)TAU_COPIES * T * TAU_BINS;
            int device = 0;
            if (bucket_sort && zhist_ctx && !g_debug_state.load() /* (diagnostics read the table back from the image buffer: tools/tau_debug.py) */ && (words & 3) == 0 && words / 4 <= 0xFFFFFFFFull && hipGetDevice(&device) == hipSuccess && device >= 0 && device < 32) {
                std::lock_guard<std::mutex> lk(ctx->mu);
                if (ctx->tau_dev_words[device] < words) {
                    if (ctx->tau_dev[device]) { (void)hipFree(ctx->tau_dev[device]); ctx->tau_dev[device] = nullptr; ctx->tau_dev_words[device] = 0; }
                    uint32_t* z = nullptr;
                    if (hipMalloc((void**)&z, words * sizeof(uint32_t)) == hipSuccess) { ctx->tau_dev[device] = z; ctx->tau_dev_words[device] = words; ctx->tau_dev_dirty[device] = true; }
                }
                if (ctx->tau_dev[device]) {
                    tau_hist = ctx->tau_dev[device]; tau_ctx_device = device;
                    // (a fresh table, or a forward that filled it and never reached its bucket sort: zeroed here)
                    if (ctx->tau_dev_dirty[device]) GS_HIP(hipMemsetAsync(tau_hist, 0, ctx->tau_dev_words[device] * sizeof(uint32_t), s));
                    ctx->tau_dev_dirty[device] = true;
                }
            }
            if (tau_ctx_device < 0) GS_HIP(hipMemsetAsync(tau_hist, 0, words * sizeof(uint32_t), s));
        }
        const int nzero = bucket_sort ? (int)nbk * BK_XCD + BK_XCD * BK_NBC_MAX : 0;      // (fine counters + the two-launch scatter's coarse ones, contiguous)
        if (rawin)
            preprocess_fwd_kernel<true><<<pf_grid, PF_THREADS, 0, s>>>(
                P, means3D, scales, rotations, opacities, raw, cov3D_precomp, cam, radii, rec0, rec1, cov_dbg,
                tiles, rect, binrec_p, kA, bucket_sort ? nullptr : vA, clip, at<uint32_t>(img, IL.bucket_cnt), zr, zh_klo, zh_shift, zh_wave_mask, at<uint32_t>(geom, GL.bk_count), nzero, hints, hint_sel,
                zcut_used, T, scalars, host_found, pre_seq, g_near_pose.load(), near_scale2, prefilter_word, untouched, tau_hist, tau_bins);
        else
            preprocess_fwd_kernel<false><<<pf_grid, PF_THREADS, 0, s>>>(
                P, means3D, scales, rotations, opacities, raw, cov3D_precomp, cam, radii, rec0, rec1, cov_dbg,
                tiles, rect, binrec_p, kA, bucket_sort ? nullptr : vA, clip, at<uint32_t>(img, IL.bucket_cnt), zr, zh_klo, zh_shift, zh_wave_mask, at<uint32_t>(geom, GL.bk_count), nzero, hints, hint_sel,
                zcut_used, T, scalars, host_found, pre_seq, g_near_pose.load(), near_scale2, prefilter_word, untouched, tau_hist, tau_bins);
        GS_LAUNCHED("preprocess_fwd");
        ctx->last_prologue_ns = (uint32_t)std::min<long long>(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_entry).count(), 0xFFFFFFFFll);
        if (tau_hist) {       // the predicted cut depths of a pose without remembered ones (a no-op for a pose the table knows, unless forced)
            const dim3 tg((unsigned)((cam.gx + TAU_TILE - 1) / TAU_TILE), (unsigned)((cam.gy + TAU_TILE - 1) / TAU_TILE));
            tau_cut_kernel<<<tg, TAU_BLK * TAU_BLK, 0, s>>>(tau_hist, T, cam.gx, cam.gy, tau_bins, (uint32_t)pol.tau_req.load() * 256u, hints ? hint_sel : nullptr, tau_forced,
                                              zcut_used, tau_bins.scale > 0.0f ? 0 : 1);
            GS_LAUNCHED("tau_cut");
        }
    }
    const bool adaptive_sort = rs_blocks_n((size_t)P, GSRAST_DEPTH_ITEMS) > RS_SELF_SCAN_BLOCKS;      // see radix_sort
    const bool assume_short = adaptive_sort && g_sort_hint.load() != 0 && ctx->depth_short.load() != 0;
    const uint32_t* order = vA;     // the radix-sorted sequence ends in A under every pass count; the bucket sort leaves (kA, vA) alone
    bool bucketed = false;          // the order in force comes from the bucket sort (per-bucket slot ranges) rather than `order`
    bool totals_pending = false;    // ... and nobody has summed the buckets' totals (R, Q, overflow verdict) into `scalars` yet
    auto sort_and_scan = [&](bool assume, bool buckets) -> int {
        bucketed = buckets; totals_pending = false;
        if (buckets) {      // three launches instead of the radix passes and the scan (gsrast_binning.h)
            uint32_t* gcount = at<uint32_t>(geom, GL.bk_count);
            uint4* slab = at<uint4>(geom, GL.bk_slab);
            {   ProfScope ps(K_SORT_DEPTH, s);
                const int items = depth_scatter_items((size_t)P);       // (elements per lane: whatever makes the launch ONE round of workgroups)
                // two-launch form (gsrast_binning.h, round 6; from g_two_level_min_p Gaussians on): coarse scatter, then the refine kernel
                const bool two = g_two_level.load() != 0 && (size_t)P >= (size_t)g_two_level_min_p.load();
                auto scatter = two ? (items == BK_ITEMS_WIDE ? depth_bucket_scatter_kernel<BK_ITEMS_WIDE, true> : depth_bucket_scatter_kernel<BK_ITEMS, true>)
                                   : (items == BK_ITEMS_WIDE ? depth_bucket_scatter_kernel<BK_ITEMS_WIDE, false> : depth_bucket_scatter_kernel<BK_ITEMS, false>);
                // (the coarse slab: the gradient records' memory -- 64 B per Gaussian, nobody's until this forward's last blend; whoever zeroes or reads the records does so behind it)
                uint4* cslab = reinterpret_cast<uint4*>(at<float>(geom, GL.grec));
                const uint32_t ccap = depth_coarse_cap((size_t)P, nbk);
                uint32_t* gccount = at<uint32_t>(geom, GL.bk_ccount);
                scatter<<<(P + 256 * items - 1) / (256 * items), 256, 0, s>>>(kA, rect, tiles, (uint32_t)P, zhist_call, zh_klo, zh_shift, nbk, two ? gccount : gcount, two ? cslab : slab, at<uint32_t>(geom, GL.bk_key), scalars + SC_ZBINS,
                                                                               zcut_used, T, (uint32_t)cam.gx, scalars + SC_N_LATE,
                                                                               cut ? at<unsigned long long>(geom, GL.color_skip) : nullptr, (uint32_t)cut_cs,
                                                                               layer_mode, hint_sel, 0.125f, zcut_used, ccap);
                GS_LAUNCHED("depth_bucket_scatter");
                if (two) {
                    depth_bucket_refine_kernel<<<(nbk >> BK_CSHIFT) * BK_XCD, 256, 0, s>>>(cslab, gccount, nbk, gcount, slab, ccap);
                    GS_LAUNCHED("depth_bucket_refine");
                }
                // (List cut: the compacting colour kernel needs nothing but the scatter's late flags.  Forked HERE, beside the bucket sort and
                // the emission, instead of behind the depth sort: 3 M 767 / 764 vs 763 / 762 views/s, 1 M 1219 / 1222 vs 1221 / 1220 -- equal.)
                // list cut: only the bucket's EARLY Gaussians are sorted (into the early set); the late ones count into bk_info's totals
                uint4* const tau_zero = tau_ctx_device >= 0 ? reinterpret_cast<uint4*>(tau_hist) : nullptr;
                const uint32_t tau_zero_n = tau_ctx_device >= 0 ? (uint32_t)((size_t)TAU_COPIES * T * TAU_BINS / 4) : 0u;
                if (cut) depth_bucket_sort_kernel<<<(nbk + BK_WAVES - 1) / BK_WAVES, 64 * BK_WAVES, 0, s>>>(slab, gcount, nbk, at<uint32_t>(geom, GL.bk_key), at<uint32_t>(geom, GL.bk_order_e), at<uint32_t>(geom, GL.bk_wincl_e),
                                                                                                            at<uint4>(geom, GL.bk_info_e), at<uint32_t>(geom, GL.bk_base_e), at<uint4>(geom, GL.bk_info),
                                                                                                            nullptr, nullptr, zhist_ctx ? zhist_call : nullptr, ZH_COPIES * ZH_BINS, tau_zero, tau_zero_n);
                else depth_bucket_sort_kernel<<<(nbk + BK_WAVES - 1) / BK_WAVES, 64 * BK_WAVES, 0, s>>>(slab, gcount, nbk, at<uint32_t>(geom, GL.bk_key), at<uint32_t>(geom, GL.bk_order), at<uint32_t>(geom, GL.bk_wincl),
                                                                                                        at<uint4>(geom, GL.bk_info), at<uint32_t>(geom, GL.bk_base),
                                                                                                        nullptr, nullptr, nullptr, zhist_ctx ? zhist_call : nullptr, ZH_COPIES * ZH_BINS, tau_zero, tau_zero_n);
                GS_LAUNCHED("depth_bucket_sort");
                if (tau_zero) { std::lock_guard<std::mutex> lk(ctx->mu); ctx->tau_dev_dirty[tau_ctx_device] = false; } }
            totals_pending = true;      // by the run emission's last workgroup, or by launch_bucket_totals() if the host needs them first
            return GSRAST_OK;
        } else {
            order = vA;
            ProfScope ps(K_SORT_DEPTH, s);
            if (bucket_sort) {      // the geometry kernel left the values to the bucket sort (which overflowed): the identity, now
                iota_kernel<<<(P + 255) / 256, 256, 0, s>>>(vA, (uint32_t)P);
                GS_LAUNCHED("iota");
            }
            // the last pass also writes rectangle widths (and, for the instance-level binning, tile counts) in depth order
            const SortAdapt<uint32_t, uint32_t> ad{ at<uint32_t>(geom, GL.keyC), at<uint32_t>(geom, GL.valC), at<uint32_t>(geom, GL.sort_minmax), scalars + 8, assume };
            int rc = radix_sort<uint32_t, uint32_t, GSRAST_DEPTH_ITEMS>(kA, vA, kB, vB, (uint32_t)P, 32, hist, scan_tmp, s, rect, runbin ? nullptr : offsets, woffsets, nullptr, &ad);
            if (rc != GSRAST_OK) return rc;
        }
        ProfScope ps(K_SCAN_TILES, s);
        int rc;
        if (runbin) {   // only the widths are scanned (-> run offsets, Q); num_rendered is just the sum of the tile counts
            rc = scan_u32(woffsets, nullptr, (uint32_t)P, woffsets, true, scan_tmp, scalars + 1, s, nullptr, tiles, scalars, scalars + 3);
        } else {
            rc = scan_u32(offsets, nullptr, (uint32_t)P, offsets, true, scan_tmp, scalars, s);
            if (rc != GSRAST_OK) return rc;
            rc = scan_u32(woffsets, nullptr, (uint32_t)P, woffsets, true, scan_tmp, scalars + 1, s);   // column runs
        }
        return rc;
    };
    { int rc = sort_and_scan(assume_short, bucket_sort); if (rc != GSRAST_OK) return rc; }
    // The colour kernel forks HERE, behind the geometry kernel and the depth sort (round 3; rounds 1-2: at entry): the geometry kernel
    // is as bandwidth-bound as the colours are (running both at once is the sum of their times) and the bucket scatter's returning
    // atomics are what the colour traffic hurts most (70 -> 150 us at 3 M); the gather-bound run emission and the run sort that now
    // run beside it stretch less.  Two alternating runs each, views/s, entry -> here: 3 M 609 -> 619, 2 M 745 -> 753, 1 M 1059 -> 1082,
    // 0.3 M 1504 -> 1551, 0.1 M 1878 -> 1894, shell 1 M 884 -> 917, cfg2 2692 -> 2787.  Behind the geometry kernel only: 3 M -3.8 %
    // (the scatter beside the colours); behind the run emission: 3 M -4 % (the colours end after the binning, the blend waits).
    // (a pose without a slot in the table has no cut depths: every visible Gaussian is early and the plain colour kernel, not the
    // compacting one, evaluates them -- the device said so at the very start of preprocess_fwd, long before this point)
    const bool pose_known = !(cut && rb_pre && hints) || read_found(rb_pre);
    if (cut && tau_mode && hints && rb_pre) {
        std::lock_guard<std::mutex> lk(ctx->mu);
        if (ctx->pose_seen.size() > 4096) ctx->pose_seen.clear();
        ctx->pose_seen[viewmatrix] = pose_known ? 1 : 0;
    }
    if (!pose_known && layer_mode == 0 && !tau_on) cut_colors = false;
    // Everything that does not depend on num_rendered is enqueued / prepared before the host waits.
    uint2* ranges = at<uint2>(img, IL.ranges);
    // reference rasterizer_impl.cu:311 (the run-compressed path writes every tile's range itself, empty ones included)
    if (!runbin) GS_HIP(hipMemsetAsync(ranges, 0, (size_t)T * sizeof(uint2), s));
    const int tpasses = tile_passes(T);
    auto bin_bytes = [&](uint32_t capR, uint32_t capQ) {
        return runbin ? runbin_layout((size_t)capR, (size_t)capQ).total : bin_layout((size_t)capR).total;
    };
    auto grow = [](uint32_t v) { return grow_capacity(v); };
    uint32_t cap = 0, capQ = 0;
    char* bin = nullptr;
    Readback* rb_flag = nullptr;                 // the counts arrive by the emission kernel's own store into pinned memory (no copy enqueued)
    uint32_t* flag_alias = nullptr; uint32_t flag_seq = 0;
    static const bool trace = getenv("GSRAST_TRACE") != nullptr;
    auto t0 = std::chrono::steady_clock::now();
    if (const uint32_t hint = ctx->R_hint.load()) {
        cap = grow(hint); capQ = grow(ctx->Q_hint.load());
        bin = (char*)binning_alloc(binning_ctx, bin_bytes(cap, capQ));
        if (!bin) cap = capQ = 0;
    }
    auto t1 = std::chrono::steady_clock::now();
    // the colour kernels, beside the binning (with the speculative run emission next on s: forked by that kernel's own start, no event)
    { int rc = launch_color(runbin && bin != nullptr && o.speculative != 0 && bucketed && totals_pending); if (rc != GSRAST_OK) return rc; }

    // ---- the rest of the forward as two re-launchable pieces ----
    // run-compressed binning; nQ / capR are either exact counts (counts_dev == nullptr) or capacities with the real
    // counts read on the device (speculative launch: grids and histogram strides follow the capacities)
    // mode (list cut): 0 = all Gaussians (as ever); 1 = the EARLY Gaussians only (counts_dev = scalars + SC_EARLY_COUNTS); 2 = all
    // Gaussians again behind a blend over cut lists, every kernel predicated on scalars[SC_REDO_PRED] (counts_dev = scalars)
    const uint32_t* redo_pred = scalars + SC_REDO_PRED;      // the completion pass's predicate (the gate's copy of it when the pass runs on its own stream)
    auto launch_run_binning = [&](char* binb, uint32_t capR_, uint32_t capQ_, uint32_t nQ, const uint32_t* counts_dev, const std::function<int()>& after_emit = nullptr, int mode = 0) -> int {
        const uint32_t* pred = mode == 2 ? redo_pred : nullptr;
        const RunBinLayout RL = runbin_layout((size_t)capR_, (size_t)capQ_);
        uint16_t *rkA = at<uint16_t>(binb, RL.rkeyA), *rkB = at<uint16_t>(binb, RL.rkeyB);
        uint2 *rvA = at<uint2>(binb, RL.rvalA), *rvB = at<uint2>(binb, RL.rvalB);
        uint32_t* hist_x = at<uint32_t>(binb, RL.hist_x);
        uint32_t* hist_y = at<uint32_t>(binb, RL.hist_y);
        uint32_t* rscan = at<uint32_t>(binb, RL.scan_tmp);
        uint32_t* plist_w = at<uint32_t>(binb, RL.point_list);
        const uint32_t* Q_dev = counts_dev ? counts_dev + 1 : nullptr;
        const int xbits = tile_bits((size_t)cam.gx);
        {   ProfScope ps(K_EMIT, s);
            const ForkWord fw = mode != 2 ? color_fork : ForkWord{};      // (pending word fork of the colour kernels: this launch signals it)
            if (bucketed && mode == 1)
                emit_column_runs_kernel<<<nbk + 1, 256, 0, s>>>(P, at<uint32_t>(geom, GL.bk_order_e), at<uint32_t>(geom, GL.bk_wincl_e), binrec_p, W, H,
                                                            o.tile_clip, capQ_, rkA, rvA, at<uint4>(geom, GL.bk_info_e), at<uint32_t>(geom, GL.bk_base_e), nbk, scalars,
                                                            flag_alias, flag_seq, at<uint4>(geom, GL.bk_info), nullptr, nullptr, 0, nullptr, nullptr, 0, nullptr, 0, nullptr, rec0, rec1, rect,
                                                            fw.word, fw.seq);
            else if (bucketed && mode == 2) {
                // COMPLETION pass: only the CANDIDATES -- the Gaussians, early or late, whose rectangle touches a tile flagged by the
                // blend -- are sorted (the buckets' non-early arrays are free) and listed, and only into flagged tiles
                depth_bucket_sort_kernel<<<(nbk + BK_WAVES - 1) / BK_WAVES, 64 * BK_WAVES, 0, s>>>(at<uint4>(geom, GL.bk_slab), at<uint32_t>(geom, GL.bk_count), nbk, at<uint32_t>(geom, GL.bk_key), at<uint32_t>(geom, GL.bk_order),
                                                                                                   at<uint32_t>(geom, GL.bk_wincl), at<uint4>(geom, GL.bk_info), at<uint32_t>(geom, GL.bk_base), nullptr, pred,
                                                                                                   at<unsigned long long>(geom, GL.cand_bits));
                emit_column_runs_kernel<<<nbk + 1, 256, 0, s>>>(P, at<uint32_t>(geom, GL.bk_order), at<uint32_t>(geom, GL.bk_wincl), binrec_p, W, H,
                                                            o.tile_clip, capQ_, rkA, rvA, at<uint4>(geom, GL.bk_info), at<uint32_t>(geom, GL.bk_base), nbk, nullptr,
                                                            nullptr, 0, nullptr, pred, at<uint32_t>(img, IL.bucket_cnt), XCD_GROUPS * WORK_BUCKETS /* (the forward's order only: the backward's keeps the tiles that are final) */, hints,
                                                            rb_pre ? reinterpret_cast<unsigned long long*>(pre_alias) + RB_FALLBACK : nullptr, pre_seq,
                                                            at<unsigned char>(img, IL.tile_flags), cam.gx, scalars + SC_PASS2, rec0, rec1, rect); }
            else if (bucketed) {
                if (cut) {      // (a list-cut forward sorted the early Gaussians only, and now everything is listed after all: the buckets are sorted again, whole)
                    depth_bucket_sort_kernel<<<(nbk + BK_WAVES - 1) / BK_WAVES, 64 * BK_WAVES, 0, s>>>(at<uint4>(geom, GL.bk_slab), at<uint32_t>(geom, GL.bk_count), nbk, at<uint32_t>(geom, GL.bk_key), at<uint32_t>(geom, GL.bk_order),
                                                                                                       at<uint32_t>(geom, GL.bk_wincl), at<uint4>(geom, GL.bk_info), at<uint32_t>(geom, GL.bk_base));
                    GS_LAUNCHED("depth_bucket_sort");
                }
                emit_column_runs_kernel<<<nbk + 1, 256, 0, s>>>(P, at<uint32_t>(geom, GL.bk_order), at<uint32_t>(geom, GL.bk_wincl), binrec_p, W, H,
                                                            o.tile_clip, capQ_, rkA, rvA, at<uint4>(geom, GL.bk_info), at<uint32_t>(geom, GL.bk_base), nbk, scalars,
                                                            flag_alias, flag_seq, nullptr, nullptr, nullptr, 0, nullptr, nullptr, 0, nullptr, 0, nullptr, rec0, rec1, rect,
                                                            fw.word, fw.seq);
            } else
                emit_column_runs_kernel<<<(P + 255) / 256, 256, 0, s>>>(P, order, woffsets, binrec_p, W, H,
                                                                       o.tile_clip, capQ_, rkA, rvA, nullptr, nullptr, 0, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0, nullptr, nullptr, 0, nullptr, 0, nullptr, rec0, rec1, rect);
            GS_LAUNCHED("emit_column_runs");
            // (signalled only by the two bucketed launches above that were handed the word)
            if (fw.word) { int rc = flush_color_fork(bucketed); if (rc != GSRAST_OK) return rc; } }
        if (bucketed && mode != 2) totals_pending = false;
        if (after_emit) { int rc = after_emit(); if (rc != GSRAST_OK) return rc; }
        const uint32_t nblk = (nQ + RUNS_PER_BLOCK - 1) / RUNS_PER_BLOCK;
        {   ProfScope ps(K_SORT_TILE, s);
            int rc = radix_sort<uint16_t, uint2, GSRAST_RUN_SORT_ITEMS>(rkA, rvA, rkB, rvB, nQ, xbits, hist_x, rscan, s, nullptr, nullptr, nullptr, Q_dev, nullptr, pred);   // runs by column
            if (rc != GSRAST_OK) return rc;
            if (radix_passes(xbits) & 1) { std::swap(rkA, rkB); std::swap(rvA, rvB); }                   // sorted runs now in (rkA, rvA)
            run_hist_rows_kernel<<<nblk, 256, 0, s>>>(rvA, nQ, Q_dev, hist_y, nblk, (uint32_t)cam.gy, pred);
            GS_LAUNCHED("run_hist_rows");
            radix_rowscan_kernel<<<cam.gy, 256, 0, s>>>(hist_y, nblk, rscan, nullptr, 0, nullptr, 0, pred);     // one workgroup per tile row
            GS_LAUNCHED("radix_rowscan");
            // the row pass and the tile ranges in one launch (gsrast_binning.h)
            rows_and_ranges_kernel<<<nblk + (uint32_t)cam.gx, RS_THREADS, 0, s>>>(rkA, rvA, nQ, counts_dev, capR_, tile_bits((size_t)cam.gy), cam.gx, cam.gy, hist_y, rscan, nblk,
                                                                                 mode == 2 ? plist_w + capR_ : plist_w, mode == 2 ? scalars + SC_PASS2 + 2 : scalars + 2, ranges,
                                                                                 buckets_ok ? at<uint32_t>(img, IL.bucket_cnt) : nullptr, at<uint16_t>(img, IL.bucket_list), hints, hint_sel, pred,
                                                                                 mode == 2 ? at<unsigned char>(img, IL.tile_flags) : nullptr, mode == 2 ? capR_ : 0u,
                                                                                 (mode == 1 && tau_on) ? zcut_used : nullptr);
            GS_LAUNCHED("rows_and_ranges"); }
        return GSRAST_OK;
    };
    // the completion pass's gate (ChainGate): claimed when the cut forward's blend is launched -- its last workgroup is the gate
    ChainGate* gate = nullptr; uint32_t gseq = 0, *gdone = nullptr, *gpred = nullptr;
    auto launch_blend = [&](const uint32_t* plist, bool fwd_lists_built, int mode = 0 /* list cut: as launch_run_binning */) -> int {
        { int rc = launch_color(); if (rc != GSRAST_OK) return rc; }  // the other binning scheme / nothing to bin: not forked yet
        { int rc = flush_color_fork(false); if (rc != GSRAST_OK) return rc; }      // (a word fork no emission picked up: s releases it itself)
        if (side) { GS_HIP(hipStreamWaitEvent(s, side->join, 0)); if (zero_in_blend) side_guard.joined = true; }       // the colours (rec2) are the blend's input
        ProfScope ps(K_BLEND_FWD, s);
        uint32_t grid = ((T + 7) / 8) * 8;
        float* fT = at<float>(img, IL.final_T); uint32_t* nc = at<uint32_t>(img, IL.n_contrib);
        uint32_t* tm = at<uint32_t>(img, IL.tile_max);
        BlendArgs ba{};
        ba.ranges = ranges; ba.plist = plist; ba.W = W; ba.H = H; ba.gx = cam.gx; ba.T = T; ba.r0 = rec0; ba.r1 = rec1; ba.r2 = rec2;
        ba.bg = background; ba.oc = out_color; ba.od = out_depth; ba.fT = fT; ba.nc = nc; ba.tm = tm;
        const int ppl = pick_ppl(T, false, o);
        const bool cull = o.cull != 0 && o.fwd_pixels_per_lane == 0;   // a forced pixels-per-lane selects the un-culled template
        if (zero_in_blend && !zero_touched && !o.forward_only && mode != 2) { ba.zero4 = at<float4>(geom, GL.grec); ba.n_zero4 = (uint32_t)((size_t)P * 4); }
        if (fwd_lists_built) { ba.hints = hints; ba.hint_sel = hint_sel; }      // (the slot is only claimed on the work-bucket path)
        ba.cut_margin_x4 = (uint32_t)pol.margin.load();
        ba.untouched = untouched;
        if (mode == 1) { ba.zcut_used = zcut_used; ba.cut_scalars = scalars; ba.tile_flags = at<unsigned char>(img, IL.tile_flags); }
        if (mode == 1 && cull && g_chain_gate.load() != 0 && !counter_collection_env() && (gate = chain_gate_of(ctx)) != nullptr) {
            {   std::lock_guard<std::mutex> lk(ctx->mu);
                uint32_t n = gate->seq + 1u; if (n == 0u) n = 1u;
                // the slot of n was last used by n - GATE_RING; it is free once the chain of n - GATE_RING / 2 (enqueued later on the same
                // in-order stream) has run.  Not yet?  Then this call does without the gate (inline chain below).
                // (a call that claimed a number but never enqueued a chain recorded no event: the youngest recorded chain among
                // n - GATE_RING / 2 ... n - GATE_RING answers for all older ones)
                bool free_slot = true;
                for (uint32_t back = GATE_RING / 2u; back <= GATE_RING; back++) {
                    const uint32_t k = n - back;
                    if (k != 0u && gate->tail_seq[k % GATE_RING] == k) { free_slot = hipEventQuery(gate->tail[k % GATE_RING]) == hipSuccess; break; }
                }
                if (!free_slot) { gate->inline_calls++; gate = nullptr; }
                else { gate->seq = n; gseq = n; gate->tail_seq[n % GATE_RING] = 0u; }
            }
            if (gate) {
                gdone = gate->words + (gseq % GATE_RING); gpred = gate->words + GATE_RING + (gseq % GATE_RING);
                ba.gate = GateArgs{ at<uint32_t>(img, IL.bucket_cnt) + (XCD_GROUPS + 1) * WORK_BUCKETS, gpred, gdone, gseq };
            }
        }
        if (mode == 2) ba.pred = redo_pred;
        if (buckets_ok) { ba.bcnt = at<uint32_t>(img, IL.bucket_cnt); ba.blist = at<uint16_t>(img, IL.bucket_list); }   // backward order: always appended
        if (cull && o.lpt) {
            if (buckets_ok && fwd_lists_built) { ba.from_buckets = 1; grid = (uint32_t)(XCD_GROUPS * xcd_group_tiles_host((size_t)cam.gx, (size_t)cam.gy)); }   // the tile-range kernel already bucketed the tiles
            else {
                uint32_t* ord = at<uint32_t>(img, IL.order_fwd);
                tile_order_kernel<<<1, 1024, 0, s>>>(T, ranges, nullptr, ord);
                GS_LAUNCHED("tile_order");
                ba.order = ord;
            }
        }
        switch (o.exp_mode) {
        case 0: if (cull) launch_fwd_cull<0>(grid, s, ba); else dispatch_fwd<0>(ppl, grid, s, ba); break;
        case 1: if (cull) launch_fwd_cull<1>(grid, s, ba); else dispatch_fwd<1>(ppl, grid, s, ba); break;
        default: if (cull) launch_fwd_cull<2>(grid, s, ba); else dispatch_fwd<2>(ppl, grid, s, ba); break;
        }
        GS_LAUNCHED("blend_fwd");
        if (side && !zero_in_blend) { GS_HIP(hipStreamWaitEvent(s, side->join2, 0)); side_guard.joined = true; }      // the gradient records are zero before anything after this forward
        return GSRAST_OK;
    };

    // {instances R (low word), column runs Q, -, R (high word, run-compressed path), ..., [8] significant depth-key bits,
    //  [9] key base, [10] "three sort passes were assumed and were not enough"}
    uint32_t counts[12] = { 0 };
    uint32_t nQ1 = 0;
    Readback* rb = nullptr;
    const bool speculative = runbin && bin != nullptr && o.speculative != 0;
    auto begin_readback = [&]() -> int { return read_u32_begin(scalars, s, 12, &rb); };
    if (!(speculative && totals_pending)) {       // (with the bucket sort the totals come out of the speculative run emission: read back right behind it)
        if (totals_pending) {
            ProfScope ps(K_SCAN_TILES, s);
            depth_bucket_scan_kernel<<<1, 256, 0, s>>>(at<uint4>(geom, GL.bk_info), nbk, scalars);
            GS_LAUNCHED("depth_bucket_scan");
            totals_pending = false;
        }
        int rc = begin_readback(); if (rc != GSRAST_OK) return rc;
    }
    // Speculative launch: with a buffer sized from the previous call, binning and blend are enqueued BEFORE the host knows
    // R and Q (the kernels read the counts on the device), so the GPU never idles on the read-back.  If the counts turn
    // out not to fit, the device published empty ranges and the two pieces are simply launched again with exact sizes.
    if (speculative) {
        const bool late = totals_pending;
        if (late) { if (rb_pre) { rb_flag = rb_pre; flag_alias = pre_alias; flag_seq = pre_seq; } else rb_flag = read_flag_prepare(&flag_alias, &flag_seq); }
        // (list cut: the sorts over the cut lists are sized for the early runs of recent forwards, not for all runs)
        // (the early set differs from pose to pose -- 0.98 M / 1.29 M column runs at two neighbouring poses of the 3 M cube --, a launch
        // sized too small costs a whole second forward, one sized too large a few empty workgroups: half again as much as the largest of
        // the recent forwards)
        // (LAYER mode -- an unknown pose, or no table: the nearest eighth of the Gaussians owns about a fifth of the column runs)
        if (cut) {
            // (a forward under predicted cut depths keeps two or three times the early set of one under remembered ones: each kind is sized by its own history)
            const bool predicted_call = tau_on && (tau_forced || !pose_known || !hints);
            const uint32_t qe = predicted_call ? ctx->Qe_hint_tau.load() : ctx->Qe_hint.load();
            const bool layer_call = layer_mode == 2 || (layer_mode == 1 && !pose_known);
            const bool early_set_expected = pose_known || layer_mode != 0 || tau_on;
            if (layer_call) nQ1 = (qe && early_set_expected) ? (uint32_t)std::min<uint64_t>(capQ, std::max<uint64_t>((uint64_t)qe + qe / 2, (uint64_t)ctx->Q_hint.load() / 3) + 4096) : capQ;
            else nQ1 = early_launch_runs(qe, capQ, early_set_expected);
        }
        int rc = launch_run_binning(bin, cap, capQ, cut ? nQ1 : capQ, cut ? scalars + SC_EARLY_COUNTS : scalars, (late && !rb_flag) ? std::function<int()>(begin_readback) : std::function<int()>(), cut ? 1 : 0);
        flag_alias = nullptr;                    // (a repeated emission below reads its counts back the ordinary way)
        if (rc == GSRAST_OK) rc = launch_blend(at<uint32_t>(bin, 0), true, cut ? 1 : 0);
        if (rc != GSRAST_OK) return rc;
    }
    { int rc = rb_flag ? read_flag_finish(rb_flag, scalars, s, counts, 12) : read_u32_finish(rb, scalars, s, counts, 12); if (rc != GSRAST_OK) return rc; }
    // equalised depth buckets: the MIDDLE of the next forward's histogram (448-895 bins) covers this one's occupied key range, padded by
    // an eighth on either side (the tails beyond take a view a whole range away); it widens at once and narrows slowly (consecutive
    // forwards render different views).  Returns whether this call's table was a learned one that held every key.
    auto learn_depth_range = [&](uint32_t zb) -> bool { return ctx->zrange.learn(zb, zh_klo, zh_shift); };      // (gsrast_policy.h: DepthRange)
    bool sort_redone = false;
    if (!bucket_sort && o.depth_sort == 0) dec_to_zero(ctx->bucket_skip);
    if (bucket_sort) {
        if (counts[11] == 0 && ctx->bucket_clean.load() < 1024 && ++ctx->bucket_clean >= 64) ctx->bucket_backoff = 0;      // the scene changed: forget
        if (counts[11] != 0) {
            const int clean_before = ctx->bucket_clean.exchange(0);      // more Gaussians at (nearly) one depth than a bucket holds: sort
            ctx->redo_count++;      // again with the radix passes (everything enqueued so far used a wrong order, as below)
            // An ISOLATED overflow (32 clean forwards before it: one pose of a cycle whose depth profile has a hard edge inside a
            // histogram bin, which doubles a few buckets' share and now and then tips one over) costs this one repeat and nothing
            // else.  Overflows in quick succession start with the radix passes for a while -- exponential back-off: 16 radix forwards,
            // twice as many after each further one (a scene whose depths pile up for good pays the discarded speculative launch ever
            // more rarely), capped at 4096 --
            // ... unless the histogram behind the bucket map was not up to this view (a context's first forward: four bins per octave;
            // a depth range that has moved out of the table): the range is learned from this call, the next forward tries again at once
            if (learn_depth_range(counts[SC_ZBINS]) && (clean_before < 32 || ctx->bucket_backoff.load() > 0))
            { const int prev = ctx->bucket_backoff.load(); const int next = prev <= 0 ? 16 : (prev >= 2048 ? 4096 : prev * 2);
              ctx->bucket_backoff = next; ctx->bucket_skip = next; }
            ctx->depth_short = 0;   // the radix path's pass-count hint is stale (not refreshed on the bucket path): assume four passes
            int rc = sort_and_scan(false, false);
            if (rc == GSRAST_OK) rc = read_u32(scalars, s, counts, 12);
            if (rc != GSRAST_OK) return rc;
            sort_redone = true;
            if (adaptive_sort) ctx->depth_short = counts[8] <= 24u ? 1 : 0;
        }
    } else if (adaptive_sort) {
        if (assume_short && counts[10] != 0) {      // the scene's depth range widened: sort again with all four passes, then as after
            ctx->redo_count++;                      // an undersized speculative launch (everything enqueued so far used a wrong order)
            int rc = sort_and_scan(false, false);
            if (rc == GSRAST_OK) rc = read_u32(scalars, s, counts, 12);
            if (rc != GSRAST_OK) return rc;
            sort_redone = true;
        }
        ctx->depth_short = counts[8] <= 24u ? 1 : 0;
    }
    if (runbin && counts[3] != 0) return fail(GSRAST_E_OVERFLOW, "forward: more than 2^31-1 instances");
    auto t2 = std::chrono::steady_clock::now();
    if (trace) fprintf(stderr, "[gsrast] alloc(spec) %.1f us, readback wait %.1f us, cap %u R %u Q %u%s | cut %d (pays %d: last_Q %u pause %d) late %u Q_early %u nQ1 %u capQ %u bucket_sort %d (over %u) hints %d zh %08x >> %d bins %u-%u\n",
                       std::chrono::duration<double, std::micro>(t1 - t0).count(), std::chrono::duration<double, std::micro>(t2 - t1).count(), cap, counts[0], counts[1],
                       speculative ? " (speculative)" : "", (int)cut, (int)cut_pays, ctx->last_Q.load(), pol.pause.load(), counts[SC_N_LATE], counts[SC_Q_EARLY], nQ1, capQ,
                       (int)bucket_sort, counts[11], hints ? 1 : 0, zh_klo, zh_shift, counts[SC_ZBINS] & 0xFFFFu, counts[SC_ZBINS] >> 16);
    if (counts[0] > 0x7FFFFFFFu) return fail(GSRAST_E_OVERFLOW, "forward: more than 2^31-1 instances");
    const uint32_t R = counts[0], Q = counts[1];
    if (bucket_sort && !sort_redone) (void)learn_depth_range(counts[SC_ZBINS]);
    // capacity hints decay slowly: consecutive calls render different views, a buffer sized for the largest recent one
    // keeps the speculative launch valid
    ctx->R_hint = follow_hint(ctx->R_hint.load(), R, 4); ctx->Q_hint = follow_hint(ctx->Q_hint.load(), Q, 4);
    ctx->last_R = R; ctx->last_Q = Q;
    ctx->last_late = cut ? counts[SC_N_LATE] : 0u; ctx->last_Qe = cut ? counts[SC_Q_EARLY] : counts[1];
    // (a cut forward that removed too little to pay for itself; round 5: FOUR in a row before the context sits out)
    if (cut) pol.forward_counts(tau_on, counts[SC_N_LATE], counts[1], counts[SC_Q_EARLY], (uint32_t)P, g_list_cut_always.load() != 0);
    if (cut && speculative && !sort_redone) {
        std::atomic<uint32_t>& hint = (tau_on && (tau_forced || !pose_known || !hints)) ? ctx->Qe_hint_tau : ctx->Qe_hint;
        hint = follow_hint(hint.load(), counts[SC_Q_EARLY], 5); }
    // fallbacks of this thread's earlier cut forwards, as the device reported them (everything enqueued before this forward's counts has run)
    // (a completion pass over a few tiles is cheap and expected; its cost grows with the tiles it lists again: one over an eighth of
    // the image counts like rounds 3's whole second forward, smaller ones in proportion)
    if (const uint32_t q2 = take_fallback_event()) {       // (the column runs of the pass's candidates)
        const int before = pol.fb_score.load();
        const int pts = pol.completion_pass(q2, ctx->last_Q.load(), (uint32_t)P, g_list_cut_always.load() != 0, tau_mode);
        if (trace) fprintf(stderr, "[gsrast] completion pass reported: %u column runs of %u, %d points on a score of %d; margin %d / 4, tau_req %d, pause %d\n", q2, ctx->last_Q.load(), pts, before,
                           pol.margin.load(), pol.tau_req.load(), pol.pause.load());
    } else if (cut && counts[SC_N_LATE] != 0u) pol.clean_cut_forward();
    const bool early_fits = !cut || counts[SC_Q_EARLY] <= nQ1;
    if (speculative && !sort_redone && R <= cap && Q <= capQ && early_fits) {          // everything is already in flight
        if (cut && counts[SC_N_LATE] != 0u) {
            // List cut: the lists in flight hold the early Gaussians only.  The blend verifies them; behind it, the whole binning and
            // blend over ALL Gaussians, predicated on its verdict (gsrast_common.h).  Nothing of this runs in the steady state.
            // the pass on its own stream behind a gate (ChainGate above), unless that cannot be had
            const hipStream_t s_caller = s;
            if (gate) {      // (claimed with the blend's launch: its last workgroup has copied the verdict and, if there is nothing to do, released us)
                GS_HIP(hipEventRecord(gate->ev, s));
                GS_HIP(hipStreamWaitEvent(gate->stream, gate->ev, 0));
                redo_pred = gpred;
                s = gate->stream;                  // (the lambdas below launch on `s`)
            }
            struct Back { hipStream_t& s; hipStream_t v; const uint32_t*& p; const uint32_t* pv; ~Back() { s = v; p = pv; } } back{ s, s_caller, redo_pred, scalars + SC_REDO_PRED };
            ProfScope ps(K_CUT_REDO, s);           // (on the stream the pass runs on: its events bracket the chain, not an empty interval of the caller's stream)
            struct Off { Off() { t_prof_off++; } ~Off() { t_prof_off--; } } off;
            // (round 4: a COMPLETION pass, not a second forward: the Gaussians that touch a flagged tile -- the candidates --, their
            // missing colours, their column runs through flagged tiles, and the flagged tiles' blend from their full lists)
            cut_candidates_kernel<<<std::min((P + 255) / 256, 2048), 256, 0, s>>>((uint32_t)P, tiles, rect, at<unsigned char>(img, IL.tile_flags), T, (uint32_t)cam.gx,
                                                                                at<unsigned long long>(geom, GL.color_skip), at<unsigned long long>(geom, GL.cand_bits),
                                                                                at<unsigned long long>(geom, GL.skip2), redo_pred);
            GS_LAUNCHED("cut_candidates");
            int rc = cut_colors ? color_kernels(s, true, redo_pred) : GSRAST_OK;      // (the late candidates' colours)
            if (rc == GSRAST_OK) rc = launch_run_binning(bin, cap, capQ, capQ, scalars + SC_PASS2, nullptr, 2);
            if (rc == GSRAST_OK) rc = launch_blend(at<uint32_t>(bin, 0), true, 2);
            if (gate) {
                // (also after an error above: the caller's stream must not wait for a release nobody sends)
                chain_done_kernel<<<1, 1, 0, gate->stream>>>(redo_pred, gdone, gseq);
                const hipError_t e1 = hipGetLastError();
                {   std::lock_guard<std::mutex> lk(ctx->mu);      // the chain's tail: whoever wants this slot's successor asks this event
                    if (hipEventRecord(gate->tail[gseq % GATE_RING], gate->stream) == hipSuccess) gate->tail_seq[gseq % GATE_RING] = gseq; }
                const hipError_t e2 = hipStreamWaitValue32(s_caller, gdone, gseq, hipStreamWaitValueEq, 0xFFFFFFFFu);
                if (e1 != hipSuccess || e2 != hipSuccess) {     // no gate after all: an ordinary join behind the chain, and never again
                    { std::lock_guard<std::mutex> lk(ctx->mu); gate->failed = true; }
                    (void)hipEventRecord(gate->ev, gate->stream);
                    (void)hipStreamWaitEvent(s_caller, gate->ev, 0);
                }
            }
            if (rc != GSRAST_OK) return rc;
        }
        { int rc = finish_records(); if (rc != GSRAST_OK) return rc; }
        if (prefilter_word) { int rc = prefilter_verdict(prefilter_word, s); if (rc != GSRAST_OK) return rc; }
        return (int)R;
    }
    if (speculative && !sort_redone) ctx->redo_count++;
    if (cut) GS_HIP(hipMemsetAsync(scalars + SC_N_LATE, 0, sizeof(uint32_t), s));      // (the backward must not take a late Gaussian's rows for zero: below it is listed)
    if (cut_colors && color_launched) {     // everything from here on lists ALL Gaussians: the colours the list cut left out are evaluated now
        { int rc = flush_color_fork(false); if (rc != GSRAST_OK) return rc; }
        if (side) GS_HIP(hipStreamWaitEvent(s, side->join, 0));
        int rc = color_kernels(s, false, nullptr);
        if (rc != GSRAST_OK) return rc;
    }
    if (speculative)    // redo: the truncated pass already appended every tile to the work buckets once
        GS_HIP(hipMemsetAsync(at<uint32_t>(img, IL.bucket_cnt), 0, (XCD_GROUPS + 1) * WORK_BUCKETS * sizeof(uint32_t), s));
    if (!bin || R > cap || Q > capQ) {   // first call, or the scene grew by more than 25 %: ask again (the callback's last answer counts)
        cap = R; capQ = Q;
        bin = (char*)binning_alloc(binning_ctx, bin_bytes(cap, capQ));
        if (!bin) return fail(GSRAST_E_ALLOC, "forward: binning allocation failed");
    }
    // The layout inside the buffer follows its CAPACITY; the sorted Gaussian ids (point_list) always end in the
    // array at offset 0, whatever the capacity, the pass count and the binning scheme -- all the backward needs.
    const uint32_t* plist = at<uint32_t>(bin, 0);
    if (runbin && (R == 0 || Q == 0)) GS_HIP(hipMemsetAsync(ranges, 0, (size_t)T * sizeof(uint2), s));
    if (R > 0 && Q > 0 && runbin) {
        int rc = launch_run_binning(bin, cap, capQ, Q, nullptr);
        if (rc != GSRAST_OK) return rc;
    } else if (R > 0 && !runbin) {
        const BinLayout BL = bin_layout((size_t)cap);
        uint32_t *tkA = at<uint32_t>(bin, BL.keyA), *tkB = at<uint32_t>(bin, BL.keyB);
        uint32_t *tvA = at<uint32_t>(bin, BL.valA), *tvB = at<uint32_t>(bin, BL.valB);
        if (tpasses & 1) { std::swap(tkA, tkB); std::swap(tvA, tvB); }   // odd pass count: start in B, finish in A
        uint32_t* bhist = at<uint32_t>(bin, BL.hist);
        uint32_t* bscan = at<uint32_t>(bin, BL.scan_tmp);
        // Tile ids fit 16 bits up to 65 536 tiles (4096 x 4096 pixels): the R-sized key streams are then
        // half as wide (the key buffers are sized for 32-bit ids either way).
        const bool k16 = T <= 65536u;
        int rc = GSRAST_OK;
        if (k16) {
            uint16_t *hA = reinterpret_cast<uint16_t*>(tkA), *hB = reinterpret_cast<uint16_t*>(tkB);
            { ProfScope ps(K_EMIT, s);
              emit_instances_kernel<uint16_t><<<(P + 255) / 256, 256, 0, s>>>(P, order, offsets, tiles, rect, cam.gx, hA, tvA);
              GS_LAUNCHED("emit_instances"); }
            { ProfScope ps(K_SORT_TILE, s); rc = radix_sort<uint16_t>(hA, tvA, hB, tvB, R, tile_bits(T), bhist, bscan, s); }
            if (rc != GSRAST_OK) return rc;
            { ProfScope ps(K_RANGES, s);
              tile_ranges_kernel<uint16_t><<<(R + 255) / 256, 256, 0, s>>>(R, nullptr, reinterpret_cast<const uint16_t*>(at<uint32_t>(bin, BL.keyA)), ranges);
              GS_LAUNCHED("tile_ranges"); }
        } else {
            { ProfScope ps(K_EMIT, s);
              emit_instances_kernel<uint32_t><<<(P + 255) / 256, 256, 0, s>>>(P, order, offsets, tiles, rect, cam.gx, tkA, tvA);
              GS_LAUNCHED("emit_instances"); }
            { ProfScope ps(K_SORT_TILE, s); rc = radix_sort<uint32_t>(tkA, tvA, tkB, tvB, R, tile_bits(T), bhist, bscan, s); }
            if (rc != GSRAST_OK) return rc;
            { ProfScope ps(K_RANGES, s);
              tile_ranges_kernel<uint32_t><<<(R + 255) / 256, 256, 0, s>>>(R, nullptr, at<uint32_t>(bin, BL.keyA), ranges);
              GS_LAUNCHED("tile_ranges"); }
        }
    }
    { int rc = launch_blend(plist, runbin && R > 0 && Q > 0); if (rc != GSRAST_OK) return rc; }
    { int rc = finish_records(); if (rc != GSRAST_OK) return rc; }
    if (prefilter_word) { int rc = prefilter_verdict(prefilter_word, s); if (rc != GSRAST_OK) return rc; }
    return (int)R;
}

int gsrast_forward_ex(gsrast_context* ctx, const gsrast_options* options,
                      gsrast_alloc_fn geometry_alloc, void* geometry_ctx, gsrast_alloc_fn binning_alloc,
                      void* binning_ctx, gsrast_alloc_fn image_alloc, void* image_ctx, int P, int D, int M,
                      const float* background, int width, int height, const float* means3D, const float* shs,
                      const float* colors_precomp, const float* opacities, const float* scales,
                      float scale_modifier, const float* rotations, const float* cov3D_precomp,
                      const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx,
                      float tan_fovy, int prefiltered, float* out_color, float* out_depth, int* radii, void* stream)
{
    return forward_impl(ctx, options, geometry_alloc, geometry_ctx, binning_alloc, binning_ctx, image_alloc, image_ctx, P, D, M, background, width, height,
                        means3D, shs, colors_precomp, opacities, scales, scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix, cam_pos,
                        tan_fovx, tan_fovy, prefiltered, out_color, out_depth, radii, stream, nullptr);
}

static const char* raw_inputs_check(int P, int M, const gsrast_raw_inputs* in)
{
    if (!in) return "raw: NULL inputs";
    if (P == 0) return nullptr;
    if (!in->xyz || !in->rotation || !in->scaling || !in->opacity_logit || !in->features_dc || (M > 1 && !in->features_rest)) return "raw: NULL required input";
    if (M < 1 || M * 3 > PP_SH_MAX || ((M * 3) & 3)) return "raw: M must be 4 or 16 (SH rows of a multiple of 16 bytes, at most 16 coefficients)";
    if (((uintptr_t)in->rotation | (uintptr_t)in->features_dc | (uintptr_t)in->features_rest | (uintptr_t)in->shs_res) & 15) return "raw: rotation / features_dc / features_rest / shs_res must be 16-byte aligned";
    return nullptr;
}

int gsrast_forward_raw(gsrast_context* ctx, const gsrast_options* options,
                       gsrast_alloc_fn geometry_alloc, void* geometry_ctx, gsrast_alloc_fn binning_alloc, void* binning_ctx,
                       gsrast_alloc_fn image_alloc, void* image_ctx, int P, int D, int M, const float* background, int width, int height,
                       const gsrast_raw_inputs* in, float scale_modifier, const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                       float tan_fovx, float tan_fovy, float* out_color, float* out_depth, int* radii, void* stream)
{
    if (const char* e = raw_inputs_check(P, M, in)) return fail(GSRAST_E_ARG, e);
    return forward_impl(ctx, options, geometry_alloc, geometry_ctx, binning_alloc, binning_ctx, image_alloc, image_ctx, P, D, M, background, width, height,
                        in->xyz, in->features_dc /* "there are SH coefficients" */, nullptr, in->opacity_logit, in->scaling, scale_modifier, in->rotation, nullptr,
                        viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, 0, out_color, out_depth, radii, stream, in);
}

int gsrast_activate_forward(int P, int M, const float* xyz, const float* motion_res, const float* rotation,
                            const float* rot_res, const float* scaling, const float* opacity_logit, const float* trbf,
                            const float* features_dc, const float* features_rest, const float* shs_res,
                            float* motion, float* rot, float* scale, float* opacity, float* shs, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (P < 0 || M < 1) return fail(GSRAST_E_ARG, "activate_forward: bad sizes");
    if (P == 0) return GSRAST_OK;
    if (!xyz || !rotation || !scaling || !opacity_logit || !features_dc || (M > 1 && !features_rest) ||
        !motion || !rot || !scale || !opacity || !shs) return fail(GSRAST_E_ARG, "activate_forward: NULL required pointer");
    epilogue_small_fwd_kernel<<<(P + 255) / 256, 256, 0, s>>>(P, xyz, motion_res, rotation, rot_res, scaling, opacity_logit, trbf,
                                                             motion, rot, scale, opacity);
    GS_LAUNCHED("epilogue_small_fwd");
    const int row = 3 * M;
    const size_t n = (size_t)P * row;
    if ((row & 3) == 0 && (((uintptr_t)shs | (uintptr_t)shs_res) & 15) == 0) {
        const size_t nq = n / 4;
        epilogue_sh_fwd_kernel<<<(unsigned)((nq + 255) / 256), 256, 0, s>>>(nq, row, features_dc, features_rest, shs_res, shs);
    } else {
        epilogue_sh_fwd_scalar_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(n, row, features_dc, features_rest, shs_res, shs);
    }
    GS_LAUNCHED("epilogue_sh_fwd");
    return GSRAST_OK;
}

int gsrast_activate_backward(int P, const float* rotation, const float* rot_res, const float* scale, const float* opacity_logit,
                             const float* trbf, const float* d_rot, const float* d_scale, const float* d_opacity,
                             float* d_rotation, float* d_scaling, float* d_rot_res, float* d_opacity_logit, float* d_trbf,
                             void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (P < 0) return fail(GSRAST_E_ARG, "activate_backward: bad sizes");
    if (P == 0) return GSRAST_OK;
    if (!rotation || !scale || !opacity_logit || !d_rotation || !d_scaling || !d_opacity_logit)
        return fail(GSRAST_E_ARG, "activate_backward: NULL required pointer");
    epilogue_small_bwd_kernel<<<(P + 255) / 256, 256, 0, s>>>(P, rotation, rot_res, scale, opacity_logit, trbf, d_rot, d_scale, d_opacity,
                                                             d_rotation, d_scaling, d_rot_res, d_opacity_logit, d_trbf);
    GS_LAUNCHED("epilogue_small_bwd");
    return GSRAST_OK;
}

int gsrast_adam_step(int n_groups, const gsrast_adam_group* groups, double beta1, double beta2, double eps, int step, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (n_groups < 0 || n_groups > ADAM_MAX_GROUPS || (n_groups > 0 && !groups) || step < 1 || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0))
        return fail(GSRAST_E_ARG, "adam_step: bad arguments (at most 8 groups, step >= 1, betas in [0, 1))");
    AdamArgs a{};
    unsigned long long blocks = 0;
    for (int k = 0; k < n_groups; k++) {
        const gsrast_adam_group& g = groups[k];
        if (g.rows < 0 || g.width < 1) return fail(GSRAST_E_ARG, "adam_step: bad group shape");
        const unsigned long long n = (unsigned long long)g.rows * (unsigned long long)g.width;
        if (n && (!g.param || !g.grad || !g.exp_avg || !g.exp_avg_sq)) return fail(GSRAST_E_ARG, "adam_step: NULL tensor");
        AdamGroup& o = a.grp[a.n_groups];
        if (n == 0) continue;
        o.p = g.param; o.g = g.grad; o.m = g.exp_avg; o.v = g.exp_avg_sq; o.lr_rows = g.lr_rows; o.lr = g.lr; o.width = (unsigned)g.width;
        o.n = n; o.first_block = blocks; o.n_blocks = (n + ADAM_THREADS * ADAM_PER_THREAD - 1) / (ADAM_THREADS * ADAM_PER_THREAD);
        blocks += o.n_blocks;
        a.n_groups++;
    }
    if (blocks == 0) return GSRAST_OK;
    if (blocks > 0x7FFFFFFFull) return fail(GSRAST_E_OVERFLOW, "adam_step: too many elements for one launch");
    a.b1 = (float)beta1; a.b2 = (float)beta2; a.eps = (float)eps;
    a.omb1 = (float)(1.0 - beta1); a.omb2 = (float)(1.0 - beta2);      // as torch: the subtraction in fp64, one rounding
    a.inv_bc1 = (float)(1.0 / (1.0 - std::pow(beta1, (double)step)));
    a.inv_sqrt_bc2 = (float)(1.0 / std::sqrt(1.0 - std::pow(beta2, (double)step)));
    adam_step_kernel<<<(unsigned)blocks, ADAM_THREADS, 0, s>>>(a);
    GS_LAUNCHED("adam_step");
    return GSRAST_OK;
}

// scratch: bbox (8 words) | codes A/B | index A/B | radix histogram | scan scratch | boxes
namespace {
struct KnnLayout { size_t bbox, cA, cB, iA, iB, hist, scan, boxes, total; };
KnnLayout knn_layout(size_t P)
{
    KnnLayout L; size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o = align256(o + bytes); return r; };
    const size_t Pp = P ? P : 1;
    L.bbox = take(32); L.cA = take(Pp * 4); L.cB = take(Pp * 4); L.iA = take(Pp * 4); L.iB = take(Pp * 4);
    const size_t hist_n = 256 * rs_blocks_n(Pp, GSRAST_DEPTH_ITEMS);
    L.hist = take(hist_n * 4); L.scan = take(scan_tmp_elems(hist_n) * 4);
    L.boxes = take(((Pp + KNN_BOX - 1) / KNN_BOX) * 6 * 4);
    L.total = o + 256;
    return L;
}
}
size_t gsrast_knn_scratch_bytes(int P) { return knn_layout(P > 0 ? (size_t)P : 0).total; }

int gsrast_knn3_mean_dist2(int P, const float* points, float* mean_dist2, char* scratch, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (P < 0) return fail(GSRAST_E_ARG, "knn3: bad size");
    if (P == 0) return GSRAST_OK;
    if (!points || !mean_dist2 || !scratch) return fail(GSRAST_E_ARG, "knn3: NULL pointer");
    const KnnLayout L = knn_layout((size_t)P);
    unsigned* bbox = at<unsigned>(scratch, L.bbox);
    uint32_t *cA = at<uint32_t>(scratch, L.cA), *cB = at<uint32_t>(scratch, L.cB);
    uint32_t *iA = at<uint32_t>(scratch, L.iA), *iB = at<uint32_t>(scratch, L.iB);
    const int nb = (P + 255) / 256;
    knn_init_kernel<<<1, 64, 0, s>>>(bbox);
    GS_LAUNCHED("knn_init");
    knn_bbox_kernel<<<nb, 256, 0, s>>>(P, points, bbox);
    GS_LAUNCHED("knn_bbox");
    knn_morton_kernel<<<nb, 256, 0, s>>>(P, points, bbox, cA, iA);
    GS_LAUNCHED("knn_morton");
    int rc = radix_sort<uint32_t, uint32_t, GSRAST_DEPTH_ITEMS>(cA, iA, cB, iB, (uint32_t)P, 30, at<uint32_t>(scratch, L.hist), at<uint32_t>(scratch, L.scan), s);
    if (rc != GSRAST_OK) return rc;
    const uint32_t* order = (radix_passes(30) & 1) ? iB : iA;
    const int nboxes = (P + KNN_BOX - 1) / KNN_BOX;
    float* boxes = at<float>(scratch, L.boxes);
    knn_boxes_kernel<<<nboxes, 256, 0, s>>>(P, points, order, boxes);
    GS_LAUNCHED("knn_boxes");
    knn_search_kernel<<<nb, 256, 0, s>>>(P, points, order, boxes, nboxes, mean_dist2);
    GS_LAUNCHED("knn_search");
    return GSRAST_OK;
}

// ---- mip-mapped feature-plane lookup (gsrast_hexplane.h) ---------------------------------------------------------
// scratch: value stacks (levels >= 1) of every plane | gradient stacks of every plane, each 256-byte aligned
extern "C++" {
namespace {
struct HexLayout { size_t mips[HEX_MAX_PLANES], gmips[HEX_MAX_PLANES], gmips_begin, gmips_end, kA, kB, vA, vB, pairs, hist, scan, total; int levels[HEX_MAX_PLANES]; int cell_bits, key_bits; };
// number of levels above 0 the published op builds: halve while an extent is > 1 and the limit allows; -1 = odd extent
int hex_levels(int W, int H, int limit)
{
    int w = W, h = H, l = 0;
    while ((w > 1 || h > 1) && l < limit) {
        if ((w > 1 && (w & 1)) || (h > 1 && (h & 1))) return -1;
        w = w > 1 ? w >> 1 : 1; h = h > 1 ? h >> 1 : 1; l++;
    }
    return l;
}
const char* hex_check(int n_planes, const gsrast_plane* planes, int C, int D, int F)
{
    if (n_planes < 1 || n_planes > HEX_MAX_PLANES || !planes) return "hexplane: 1..24 planes";
    if (C < 4 || C > 64 || (C & (C - 1))) return "hexplane: channels must be a power of two in [4, 64]";
    if (D < 2 || D > 8 || F < C || (F & 3)) return "hexplane: 2..8 coordinates per point, feature rows a multiple of 4 floats";
    for (int p = 0; p < n_planes; p++) {
        const gsrast_plane& g = planes[p];
        if (g.W < 1 || g.H < 1 || (long long)g.W * g.H > (1ll << 26)) return "hexplane: bad plane extent";
        if (g.cu < 0 || g.cu >= D || g.cv < 0 || g.cv >= D) return "hexplane: coordinate column outside the point row";
        if (g.max_mip_level < 0) return "hexplane: negative max_mip_level";
        if (g.out_offset < 0 || (g.out_offset & 3) || g.out_offset + C > F) return "hexplane: feature block outside the row";
        if (hex_levels(g.W, g.H, g.max_mip_level) < 0) return "hexplane: a mip level has an odd extent > 1 (limit max_mip_level)";
        if (hex_levels(g.W, g.H, g.max_mip_level) >= HEX_MAX_LEVELS) return "hexplane: too many mip levels";
    }
    return nullptr;
}
HexLayout hex_layout(int n_planes, const gsrast_plane* planes, int C, size_t N)
{
    HexLayout L{}; size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o = align256(o + bytes); return r; };
    for (int pass = 0; pass < 2; pass++) {
        if (pass == 1) L.gmips_begin = o;
        for (int p = 0; p < n_planes; p++) {
            const int nl = hex_levels(planes[p].W, planes[p].H, planes[p].max_mip_level);
            L.levels[p] = nl;
            const size_t texels = hex_level_offset(planes[p].W, planes[p].H, nl + 1);
            (pass ? L.gmips[p] : L.mips[p]) = take(texels * (size_t)C * 4);
        }
    }
    L.gmips_end = o;
    // sorted-run backward: (key, point) pairs of every plane, double-buffered, + the radix sort's tables
    size_t widest = 1;
    for (int p = 0; p < n_planes; p++)
        widest = std::max(widest, (size_t)planes[p].W * planes[p].H + hex_level_offset(planes[p].W, planes[p].H, L.levels[p] + 1));
    L.cell_bits = 1; while (((size_t)1 << L.cell_bits) < widest) L.cell_bits++;
    int pb = 0; while ((1 << pb) < n_planes) pb++;
    L.key_bits = L.cell_bits + pb;
    const size_t E = std::max<size_t>((size_t)n_planes * N, 1);
    L.kA = take(E * 4); L.kB = take(E * 4); L.vA = take(E * 4); L.vB = take(E * 4); L.pairs = take(E * sizeof(HexPair));
    const size_t hist_n = 256 * rs_blocks_n(E, RS_ITEMS);
    L.hist = take(hist_n * 4); L.scan = take(scan_tmp_elems(hist_n) * 4);
    L.total = o + 256;
    return L;
}
void hex_fill(HexArgs& a, const HexLayout& L, int N, int D, int C, int F, int n_planes, const gsrast_plane* planes, char* scratch, bool backward)
{
    a.n_planes = n_planes; a.C = C; a.N = N; a.D = D; a.F = F;
    for (int p = 0; p < n_planes; p++) {
        HexPlane& P = a.pl[p];
        P.tex = planes[p].tex; P.grad = planes[p].grad_tex;
        P.mips = at<float>(scratch, L.mips[p]); P.gmips = at<float>(scratch, L.gmips[p]);
        P.W = planes[p].W; P.H = planes[p].H; P.cu = planes[p].cu; P.cv = planes[p].cv;
        P.n_levels = L.levels[p]; P.out_offset = planes[p].out_offset;
    }
}
int hex_build_mips(const HexArgs& a, hipStream_t s)
{
    int top = 0; unsigned long long widest[HEX_MAX_LEVELS + 1] = {0};
    for (int p = 0; p < a.n_planes; p++) {
        top = std::max(top, a.pl[p].n_levels);
        for (int l = 1; l <= a.pl[p].n_levels; l++)
            widest[l] = std::max(widest[l], (unsigned long long)hex_extent(a.pl[p].W, l) * hex_extent(a.pl[p].H, l) * (a.C >> 2));
    }
    for (int l = 1; l <= top; l++) {
        hex_mip_build_kernel<<<dim3((unsigned)((widest[l] + 255) / 256), (unsigned)a.n_planes), 256, 0, s>>>(a, l);
        GS_LAUNCHED("hex_mip_build");
    }
    return GSRAST_OK;
}
template <int C>
int hex_launch_sorted(const HexArgs& a, int cell_bits, unsigned E, const uint32_t* keys, const uint32_t* vals, const HexPair* pairs, const float* pts,
                      const float* levels, const float* dy, hipStream_t s)
{
    const unsigned long long groups = ((unsigned long long)E + HEX_RUN_CHUNK - 1) / HEX_RUN_CHUNK;
    hex_grad_tex_sorted_kernel<C><<<(unsigned)((groups * C + 255) / 256), 256, 0, s>>>(a, cell_bits, E, keys, vals, pairs, pts, levels, dy, g_ablate.load());
    GS_LAUNCHED("hex_grad_tex_sorted");
    return GSRAST_OK;
}
}
}   // extern "C++"

size_t gsrast_hexplane_scratch_bytes(int n_planes, const gsrast_plane* planes, int C, int N)
{
    if (N < 0 || hex_check(n_planes, planes, C, 8, 64)) return 0;
    return hex_layout(n_planes, planes, C, (size_t)N).total;
}

int gsrast_hexplane_forward(int N, int D, int C, int F, int n_planes, const gsrast_plane* planes, const float* pts, const float* levels,
                            float* features, char* scratch, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (N < 0) return fail(GSRAST_E_ARG, "hexplane_forward: bad size");
    if (const char* e = hex_check(n_planes, planes, C, D, F)) return fail(GSRAST_E_ARG, e);
    for (int p = 0; p < n_planes; p++) if (!planes[p].tex) return fail(GSRAST_E_ARG, "hexplane_forward: NULL plane");
    if (!scratch || (N > 0 && (!pts || !levels || !features))) return fail(GSRAST_E_ARG, "hexplane_forward: NULL pointer");
    const HexLayout L = hex_layout(n_planes, planes, C, (size_t)N);
    HexArgs a{};
    hex_fill(a, L, N, D, C, F, n_planes, planes, scratch, false);
    if (int rc = hex_build_mips(a, s)) return rc;
    if (N == 0) return GSRAST_OK;
    const unsigned long long lanes = (unsigned long long)N * (C >> 2);
    hex_sample_fwd_kernel<<<(unsigned)((lanes + 255) / 256), 256, 0, s>>>(a, pts, levels, features);
    GS_LAUNCHED("hex_sample_fwd");
    return GSRAST_OK;
}

int gsrast_hexplane_backward(int N, int D, int C, int F, int n_planes, const gsrast_plane* planes, const float* pts, const float* levels,
                             const float* d_features, float* d_pts, float* d_levels, int mips_built, char* scratch, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (N < 0) return fail(GSRAST_E_ARG, "hexplane_backward: bad size");
    if (const char* e = hex_check(n_planes, planes, C, D, F)) return fail(GSRAST_E_ARG, e);
    for (int p = 0; p < n_planes; p++) if (!planes[p].tex || !planes[p].grad_tex) return fail(GSRAST_E_ARG, "hexplane_backward: NULL plane or gradient");
    if (!scratch || (N > 0 && (!pts || !levels || !d_features))) return fail(GSRAST_E_ARG, "hexplane_backward: NULL pointer");
    const HexLayout L = hex_layout(n_planes, planes, C, (size_t)N);
    HexArgs a{};
    hex_fill(a, L, N, D, C, F, n_planes, planes, scratch, true);
    for (int p = 0; p < n_planes; p++) GS_HIP(hipMemsetAsync(planes[p].grad_tex, 0, (size_t)planes[p].W * planes[p].H * C * 4, s));
    if (N == 0) return GSRAST_OK;
    int top = 0;
    for (int p = 0; p < n_planes; p++) top = std::max(top, a.pl[p].n_levels);
    if (top) GS_HIP(hipMemsetAsync(scratch + L.gmips_begin, 0, L.gmips_end - L.gmips_begin, s));
    const unsigned long long E = (unsigned long long)n_planes * N;
    if (g_hex_scatter.load() == 0 && L.key_bits <= 32 && E < 0xFFFFFFFFull && (unsigned long long)N * F < 0xFFFFFFFFull) {
        uint32_t *kA = at<uint32_t>(scratch, L.kA), *kB = at<uint32_t>(scratch, L.kB), *vA = at<uint32_t>(scratch, L.vA), *vB = at<uint32_t>(scratch, L.vB);
        hex_keys_kernel<<<dim3((unsigned)((N + 255) / 256), (unsigned)n_planes), 256, 0, s>>>(a, L.cell_bits, pts, levels, kA, vA, at<HexPair>(scratch, L.pairs));
        GS_LAUNCHED("hex_keys");
        if (int rc = radix_sort<uint32_t, uint32_t, RS_ITEMS>(kA, vA, kB, vB, (uint32_t)E, L.key_bits, at<uint32_t>(scratch, L.hist), at<uint32_t>(scratch, L.scan), s)) return rc;
        const bool inB = radix_passes(L.key_bits) & 1;
        const uint32_t *ks = inB ? kB : kA, *vs = inB ? vB : vA;
        int rc;
        switch (C) {
            case 4: rc = hex_launch_sorted<4>(a, L.cell_bits, (unsigned)E, ks, vs, at<HexPair>(scratch, L.pairs), pts, levels, d_features, s); break;
            case 8: rc = hex_launch_sorted<8>(a, L.cell_bits, (unsigned)E, ks, vs, at<HexPair>(scratch, L.pairs), pts, levels, d_features, s); break;
            case 16: rc = hex_launch_sorted<16>(a, L.cell_bits, (unsigned)E, ks, vs, at<HexPair>(scratch, L.pairs), pts, levels, d_features, s); break;
            case 32: rc = hex_launch_sorted<32>(a, L.cell_bits, (unsigned)E, ks, vs, at<HexPair>(scratch, L.pairs), pts, levels, d_features, s); break;
            default: rc = hex_launch_sorted<64>(a, L.cell_bits, (unsigned)E, ks, vs, at<HexPair>(scratch, L.pairs), pts, levels, d_features, s); break;
        }
        if (rc) return rc;
    } else {
        const unsigned long long lanes = (unsigned long long)N * C;
        hex_grad_tex_global_kernel<<<(unsigned)((lanes + 255) / 256), 256, 0, s>>>(a, pts, levels, d_features);
        GS_LAUNCHED("hex_grad_tex_global");
    }
    for (int l = top; l >= 1; l--) {
        unsigned long long widest = 0;
        for (int p = 0; p < n_planes; p++) if (a.pl[p].n_levels >= l)
            widest = std::max(widest, (unsigned long long)hex_extent(a.pl[p].W, l) * hex_extent(a.pl[p].H, l) * (C >> 2));
        hex_mip_pull_kernel<<<dim3((unsigned)((widest + 255) / 256), (unsigned)n_planes), 256, 0, s>>>(a, l);
        GS_LAUNCHED("hex_mip_pull");
    }
    if (d_pts || d_levels) {
        if (!mips_built) if (int rc = hex_build_mips(a, s)) return rc;
        const unsigned long long lanes = (unsigned long long)N * (C >> 2);
        hex_grad_uv_kernel<<<(unsigned)((lanes + 255) / 256), 256, 0, s>>>(a, pts, levels, d_features, d_pts, d_levels);
        GS_LAUNCHED("hex_grad_uv");
    }
    return GSRAST_OK;
}

__global__ void __launch_bounds__(256)
touched_rows_kernel(int P, const unsigned char* __restrict__ untouched, const uint32_t* __restrict__ scalars, unsigned char* __restrict__ flags)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    flags[i] = scalars[SC_TOUCH_VALID] != 0u ? (unsigned char)(untouched[i] ? 0 : 1) : (unsigned char)1;
}
int gsrast_touched_rows(int P, const char* geom_buffer, unsigned char* flags, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (P < 0 || (P > 0 && (!geom_buffer || !flags))) return fail(GSRAST_E_ARG, "touched_rows: bad arguments");
    if (P == 0) return GSRAST_OK;
    const GeomLayout GL = geom_layout((size_t)P);
    touched_rows_kernel<<<(P + 255) / 256, 256, 0, s>>>(P, at<unsigned char>(geom_buffer, GL.untouched), at<uint32_t>(geom_buffer, GL.scalars), flags);
    GS_LAUNCHED("touched_rows");
    return GSRAST_OK;
}

static int rows_pack_impl(bool pack, long long n, const long long* idx, int n_arrays, float* const* arrays, const int* widths, float* packed, void* stream)
{
    if (n < 0 || n_arrays < 1 || n_arrays > 8 || !arrays || !widths) return fail(GSRAST_E_ARG, "rows_pack: bad arguments");
    if (n == 0) return GSRAST_OK;
    RowArrays a{}; a.n = n_arrays;
    for (int k = 0; k < n_arrays; k++) {
        if (!arrays[k] || widths[k] < 1) return fail(GSRAST_E_ARG, "rows_pack: NULL array or width < 1");
        a.ptr[k] = arrays[k]; a.width[k] = widths[k]; a.total += widths[k];
    }
    if (!idx || !packed || n * a.total > 0x7FFFFFFFll * 256) return fail(GSRAST_E_ARG, "rows_pack: NULL index / packed array, or too large");
    const unsigned blocks = (unsigned)((n * a.total + 255) / 256);
    hipStream_t s = (hipStream_t)stream;
    if (pack) rows_pack_kernel<true><<<blocks, 256, 0, s>>>(n, idx, a, packed);
    else rows_pack_kernel<false><<<blocks, 256, 0, s>>>(n, idx, a, packed);
    GS_LAUNCHED("rows_pack");
    return GSRAST_OK;
}
int gsrast_rows_pack(long long n, const long long* idx, int n_arrays, const float* const* arrays, const int* widths, float* packed, void* stream)
{ return rows_pack_impl(true, n, idx, n_arrays, const_cast<float* const*>(arrays), widths, packed, stream); }
int gsrast_rows_unpack(long long n, const long long* idx, int n_arrays, float* const* arrays, const int* widths, const float* packed, void* stream)
{ return rows_pack_impl(false, n, idx, n_arrays, arrays, widths, const_cast<float*>(packed), stream); }

// ---- the all-gather gradient exchange (gsrast_exchange.h) ----
static int grow_arrays(GradRowArrays& a, float* const* dense, int M, float* dL_dsh, float* d_dc, float* d_rest, const char* who, bool need_dense = true)
{
    for (int k = 0; k < 4; k++) {
        if (need_dense && (!dense || !dense[k])) return fail(GSRAST_E_ARG, who);
        a.dense[k] = dense ? dense[k] : nullptr;
    }
    a.sh = dL_dsh; a.dc = d_dc; a.rest = d_rest; a.M = M;
    if (dL_dsh || d_dc) {
        if (M < 1 || M * 3 > PP_SH_MAX || ((M * 3) & 3) || ((uintptr_t)dL_dsh & 15) || (d_dc && M > 1 && !d_rest)) return fail(GSRAST_E_ARG, who);
    }
    return GSRAST_OK;
}
int gsrast_grad_rows_pack(int P, const unsigned char* touched, float* const* dense, const float* factor, uint32_t* rows, uint32_t cap, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (P < 0 || (P > 0 && (!touched || !factor || !rows))) return fail(GSRAST_E_ARG, "grad_rows_pack: bad arguments");
    if (P == 0) return GSRAST_OK;
    GradRowArrays a{};
    if (int rc = grow_arrays(a, dense, 0, nullptr, nullptr, nullptr, "grad_rows_pack: four dense arrays are required")) return rc;
    grad_rows_pack_kernel<<<(P + GROW_PACK - 1) / GROW_PACK, 256, 0, s>>>(P, touched, a, factor, rows, cap);
    GS_LAUNCHED("grad_rows_pack");
    return GSRAST_OK;
}
int gsrast_grad_rows_clear(int P, const uint32_t* chunks, int n_chunks, size_t chunk_words, uint32_t cap, float* const* dense, int M, float* dL_dsh,
                           float* d_features_dc, float* d_features_rest, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (P < 0 || n_chunks < 0 || (n_chunks > 0 && cap > 0 && !chunks) || chunk_words < (size_t)(1 + (size_t)cap) * GROW_WORDS) return fail(GSRAST_E_ARG, "grad_rows_clear: bad arguments");
    if (n_chunks == 0 || cap == 0) return GSRAST_OK;
    const int what = (dense ? 1 : 0) | ((dL_dsh || d_features_dc) ? 2 : 0);
    if (!what) return GSRAST_OK;
    GradRowArrays a{};
    if (int rc = grow_arrays(a, dense, M, dL_dsh, d_features_dc, d_features_rest, "grad_rows_clear: bad arrays", dense != nullptr)) return rc;
    const size_t lanes = (size_t)n_chunks * cap * 16;
    if (lanes > 0x7FFFFFFFull * 256) return fail(GSRAST_E_ARG, "grad_rows_clear: too many rows");
    grad_rows_clear_kernel<<<(unsigned)((lanes + 255) / 256), 256, 0, s>>>(chunks, n_chunks, chunk_words, cap, a, what, (uint32_t)P);
    GS_LAUNCHED("grad_rows_clear");
    return GSRAST_OK;
}
int gsrast_grad_rows_add(int P, const uint32_t* chunk, uint32_t cap, float* const* dense, int D, int M, const float* means3D, float scale, float* dL_dsh,
                         float* d_features_dc, float* d_features_rest, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (cap == 0 || P == 0) return GSRAST_OK;
    if (P < 0 || !chunk || !means3D || D < 0 || D > 3 || ((dL_dsh || d_features_dc) && (D + 1) * (D + 1) > M)) return fail(GSRAST_E_ARG, "grad_rows_add: bad arguments");
    GradRowArrays a{};
    if (int rc = grow_arrays(a, dense, M, dL_dsh, d_features_dc, d_features_rest, "grad_rows_add: bad arrays")) return rc;
    grad_rows_add_kernel<<<(cap + PP_THREADS - 1) / PP_THREADS, PP_THREADS, 0, s>>>(chunk, cap, a, means3D, D, scale, (uint32_t)P);
    GS_LAUNCHED("grad_rows_add");
    return GSRAST_OK;
}

int gsrast_sh_grad_combine(int P, int D, int M, int N, const float* means3D, const float* chunks, size_t chunk_stride,
                           float scale, float* dL_dsh, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (P < 0 || N < 1 || M < 1 || D < 0 || D > 3 || (D + 1) * (D + 1) > M) return fail(GSRAST_E_ARG, "sh_grad_combine: bad sizes");
    if (P == 0) return GSRAST_OK;
    if (!means3D || !chunks || !dL_dsh || chunk_stride < (size_t)3 * P + 3) return fail(GSRAST_E_ARG, "sh_grad_combine: NULL or short buffer");
    sh_grad_combine_kernel<<<(P + PP_THREADS - 1) / PP_THREADS, PP_THREADS, 0, s>>>(P, D, M, N, means3D, chunks, chunk_stride, scale, dL_dsh, P, nullptr, nullptr, nullptr);
    GS_LAUNCHED("sh_grad_combine");
    return GSRAST_OK;
}

int gsrast_sh_grad_combine_rows(int P, int D, int M, int N, const float* means3D, const float* chunks, size_t chunk_stride, int rows,
                                const int* row_of, float scale, float* dL_dsh, float* d_features_dc, float* d_features_rest, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (P < 0 || N < 1 || M < 1 || D < 0 || D > 3 || (D + 1) * (D + 1) > M || rows < 0 || rows > P || (!row_of && rows != P))
        return fail(GSRAST_E_ARG, "sh_grad_combine_rows: bad sizes");
    if (P == 0) return GSRAST_OK;
    if (!means3D || !chunks || chunk_stride < (size_t)3 * rows + 3) return fail(GSRAST_E_ARG, "sh_grad_combine_rows: NULL or short buffer");
    if (!dL_dsh && !d_features_dc) return fail(GSRAST_E_ARG, "sh_grad_combine_rows: no output array");
    if (d_features_dc && M > 1 && !d_features_rest) return fail(GSRAST_E_ARG, "sh_grad_combine_rows: d_features_dc without d_features_rest");
    if (d_features_dc && (M * 3 > PP_SH_MAX || ((M * 3) & 3) || (((uintptr_t)d_features_dc | (uintptr_t)d_features_rest) & 15)))
        return fail(GSRAST_E_ARG, "sh_grad_combine_rows: the split outputs need M = 4 or 16 and 16-byte aligned arrays");
    sh_grad_combine_kernel<<<(P + PP_THREADS - 1) / PP_THREADS, PP_THREADS, 0, s>>>(P, D, M, N, means3D, chunks, chunk_stride, scale, dL_dsh, rows, row_of,
                                                                                   d_features_dc, d_features_rest);
    GS_LAUNCHED("sh_grad_combine_rows");
    return GSRAST_OK;
}

int gsrast_sh_grad_combine_union(int P, int D, int M, int N, const float* means3D, const float* chunks, size_t chunk_stride, int rows,
                                 const long long* idx, float scale, float* dL_dsh, float* d_features_dc, float* d_features_rest, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (P < 0 || N < 1 || M < 1 || D < 0 || D > 3 || (D + 1) * (D + 1) > M || rows < 0 || rows > P)
        return fail(GSRAST_E_ARG, "sh_grad_combine_union: bad sizes");
    if (P == 0 || rows == 0) return GSRAST_OK;
    if (!means3D || !chunks || !idx || chunk_stride < (size_t)3 * rows + 3) return fail(GSRAST_E_ARG, "sh_grad_combine_union: NULL or short buffer");
    if (!dL_dsh && !d_features_dc) return fail(GSRAST_E_ARG, "sh_grad_combine_union: no output array");
    if (d_features_dc && !d_features_rest) return fail(GSRAST_E_ARG, "sh_grad_combine_union: d_features_dc without d_features_rest");
    if (M * 3 > PP_SH_MAX || ((M * 3) & 3) || ((uintptr_t)dL_dsh & 15))
        return fail(GSRAST_E_ARG, "sh_grad_combine_union: needs M = 4, 8, 12 or 16 and a 16-byte aligned dL_dsh");
    sh_grad_combine_union_kernel<<<(rows + PP_THREADS - 1) / PP_THREADS, PP_THREADS, 0, s>>>(rows, idx, D, M, N, means3D, chunks, chunk_stride, scale,
                                                                                          dL_dsh, d_features_dc, d_features_rest);
    GS_LAUNCHED("sh_grad_combine_union");
    return GSRAST_OK;
}

int gsrast_backward(int P, int D, int M, int R, const float* background, int width, int height,
                    const float* means3D, const float* shs, const float* colors_precomp, const float* scales,
                    float scale_modifier, const float* rotations, const float* cov3D_precomp,
                    const float* viewmatrix, const float* projmatrix, const float* campos, float tan_fovx,
                    float tan_fovy, const int* radii, char* geom_buffer, char* binning_buffer, char* image_buffer,
                    const float* dL_dpix, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor,
                    float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, void* stream)
{
    return gsrast_backward_ex(nullptr, P, D, M, R, background, width, height, means3D, shs, colors_precomp, scales, scale_modifier, rotations,
                              cov3D_precomp, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, radii, geom_buffer, binning_buffer,
                              image_buffer, dL_dpix, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh,
                              dL_dscale, dL_drot, stream);
}

static int backward_impl(const gsrast_options* options, int P, int D, int M, int R, const float* background, int width, int height,
                       const float* means3D, const float* shs, const float* colors_precomp, const float* scales,
                       float scale_modifier, const float* rotations, const float* cov3D_precomp,
                       const float* viewmatrix, const float* projmatrix, const float* campos, float tan_fovx,
                       float tan_fovy, const int* radii, char* geom_buffer, char* binning_buffer, char* image_buffer,
                       const float* dL_dpix, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor,
                       float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, void* stream,
                       const gsrast_raw_inputs* rawin, const gsrast_raw_grads* rawout)
{
    RoctxRange range_bwd(rawin ? "gsrast_backward_raw" : "gsrast_backward");
    CallScope call_scope;
    RawArgs raw{}; RawGrads rawg{};
    if (rawin) {
        raw.motion_res = rawin->motion_res; raw.rot_res = rawin->rot_res; raw.trbf = rawin->trbf; raw.opacity_logit = rawin->opacity_logit;
        raw.features_dc = rawin->features_dc; raw.features_rest = rawin->features_rest; raw.shs_res = rawin->shs_res;
        rawg.d_rot_res = rawout->d_rot_res; rawg.d_trbf = rawout->d_trbf; rawg.d_dc = rawout->d_features_dc; rawg.d_rest = rawout->d_features_rest;
        rawg.d_shs_res = rawout->d_shs_res;
    }
    const gsrast_options o = options ? *options : snapshot_defaults();
    if (!options_valid(o)) return fail(GSRAST_E_ARG, "backward: bad option value");
    hipStream_t s = (hipStream_t)stream;
    const int W = width, H = height;
    if (P < 0 || R < 0 || W <= 0 || H <= 0) return fail(GSRAST_E_ARG, "backward: bad sizes");
    if (P == 0) return GSRAST_OK;
    if (!geom_buffer || !image_buffer || (R > 0 && !binning_buffer)) return fail(GSRAST_E_ARG, "backward: NULL state buffer");
    if (!means3D || !radii || !viewmatrix || !projmatrix || !dL_dpix || !background) return fail(GSRAST_E_ARG, "backward: NULL required input");
    if (!dL_dmean2D || !dL_dopacity || !dL_dmean3D) return fail(GSRAST_E_ARG, "backward: NULL gradient output");
    if (colors_precomp && !dL_dcolor) return fail(GSRAST_E_ARG, "backward: colors_precomp path needs dL_dcolor");
    if (dL_dconic && ((uintptr_t)dL_dconic & 15)) return fail(GSRAST_E_ARG, "backward: dL_dconic must be 16-byte aligned");
    if (rotations && (((uintptr_t)rotations | (uintptr_t)dL_drot) & 15)) return fail(GSRAST_E_ARG, "backward: rotations / dL_drot must be 16-byte aligned");
    if (cov3D_precomp && !dL_dcov3D) return fail(GSRAST_E_ARG, "backward: cov3D_precomp path needs dL_dcov3D");
    const bool use_sh = shs && !colors_precomp;
    const bool use_sr = !cov3D_precomp;
    if (use_sh && (!dL_dsh || !campos)) return fail(GSRAST_E_ARG, "backward: SH path needs dL_dsh and campos");
    if (use_sr && (!scales || !rotations || !dL_dscale || !dL_drot)) return fail(GSRAST_E_ARG, "backward: scale/rotation path needs their gradients");

    const CamArgs cam = make_cam(viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, scale_modifier, W, H);
    const uint32_t T = (uint32_t)cam.gx * (uint32_t)cam.gy;
    const GeomLayout GL = geom_layout((size_t)P);
    const ImgLayout IL = img_layout((size_t)W, (size_t)H);
    char* geom = geom_buffer; char* bin = binning_buffer; char* img = image_buffer;
    const uint32_t* plist = bin ? at<uint32_t>(bin, 0) : nullptr;     // BinLayout: point_list lives at offset 0
    const float4* rec0 = at<float4>(geom, GL.rec0); const float4* rec1 = at<float4>(geom, GL.rec1); const float4* rec2 = at<float4>(geom, GL.rec2);

    // per-Gaussian gradient records: the only memory the backward accumulates into (64 B / Gaussian, in the geometry buffer)
    // (the forward leaves them zero: a caller that knows this is the first backward on this state says so and saves the fill)
    float* grec = at<float>(geom, GL.grec);
    const bool do_blend = o.backward_phase != 2, do_geom = o.backward_phase != 1;
    // (a forward that was told no backward would follow has not zeroed the records)
    if (do_blend && (!o.grads_zeroed || o.forward_only)) GS_HIP(hipMemsetAsync(grec, 0, (size_t)P * GREC * sizeof(float), s));
    // What the per-Gaussian backward needs of the SH coefficients -- d(colour)/d(view direction), 36 B instead of 12*M -- depends on
    // nothing the blend backward produces: evaluated on the side stream of the calling thread's context WHILE the VALU-bound blend
    // backward runs, joined in front of preprocess_bwd (or at the end of phase 1 of a two-phase backward).
    // Round 3: the forward's colour kernel leaves those nine floats per Gaussian while it has the coefficient block in LDS
    // (preprocess_color_kernel), so this kernel only runs for a state whose forward was told that no backward would follow
    // (options.forward_only, passed to both calls by a caller that changed its mind).
    SideStream* side = nullptr;
    bool side_has_derivs = false;       // the side stream computes something the per-Gaussian backward reads (a forward_only state's direction derivatives)
    struct BwdSideGuard {       // an error exit below must not leave side-stream work running on the caller's arrays
        SideStream*& side; bool joined = false;
        ~BwdSideGuard() { if (side && !joined) (void)hipStreamSynchronize(side->stream); }
    } side_guard{ side };
    if (do_blend && use_sh && D > 0 && o.forward_only) {
        if (rawin) return fail(GSRAST_E_ARG, "backward_raw: the forward was run with forward_only");
        if (o.side_stream && R > 0 && !side) {
            side = side_stream_of(thread_context());
            if (side) { GS_HIP(hipEventRecord(side->fork, s)); GS_HIP(hipStreamWaitEvent(side->stream, side->fork, 0)); }
        }
        hipStream_t ds = side ? side->stream : s;
        // two waves per compute unit, grid-stride: enough loads in flight for ~1.5 TB/s, few enough not to push the blend kernel's
        // workgroups off the chip (an unthrottled launch slowed the blend backward by 20 %, this one by 2 %)
        const int grid = side ? std::min((P + 63) / 64, 512) : (P + 63) / 64;
        {
            ProfScope ps(K_SH_DERIVS, ds);
            sh_dir_derivs_kernel<<<grid, 64, 0, ds>>>(P, D, M, means3D, shs, campos, radii,
                                                      at<float4>(geom, GL.shdA), at<float4>(geom, GL.shdB), at<float>(geom, GL.shdC));
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return fail(GSRAST_E_DEVICE, "sh_dir_derivs", e);
        }
        if (side) { GS_HIP(hipEventRecord(side->join, side->stream)); side_has_derivs = true; }
    }
    // List cut (gsrast_common.h): the zero rows of the Gaussians the forward left out are written on the side stream, beside the blend
    // backward; preprocess_bwd then neither reads nor writes them.  Both kernels check on the device that the forward's cut was in force
    // and held.  Only where it pays for the two events (large scenes), with the sparse per-Gaussian backward, in a one-phase call.
    bool late_fill = false;
    ForkWord late_fork{};
    std::function<int()> launch_late_fill;
    if (do_blend && do_geom && R > 0 && !o.dense_backward && o.side_stream && (P >= g_late_fill_min_p.load() || g_list_cut_always.load() != 0)) {
        if (!side) {
            side = side_stream_of(thread_context());
            // word fork: the blend backward below signals its own start, the side stream waits for that -- when it is the transposed kernel
            if (side && o.cull != 0 && o.lpt && pick_ppl(T, true, o) == 1 && g_bwd_transposed.load() && g_ablate.load() == 0) late_fork = fork_word_next(thread_context(), side, s);
            if (side && !late_fork.word) { GS_HIP(hipEventRecord(side->fork, s)); GS_HIP(hipStreamWaitEvent(side->stream, side->fork, 0)); }
        }
        if (side) {
            LateRowsArgs la{}; int n = 0;
            auto add = [&](float* p, int rl) { if (p && rl > 0) { la.ptr[n] = p; la.rowlen[n] = rl; n++; } };
            add(dL_dmean2D, 3); add(dL_dopacity, 1); add(dL_dmean3D, 3); add(dL_dconic, 4); add(dL_dcolor, 3); add(dL_dcov3D, 6);
            if (rawin || use_sr) { add(dL_dscale, 3); add(dL_drot, 4); }
            if (rawin) { add(rawg.d_rot_res, 7); add(rawg.d_trbf, 1); add(rawg.d_shs_res, M * 3); add(rawg.d_dc, 3); add(rawg.d_rest, M * 3 - 3); }
            else if (use_sh && !o.sh_grad_factors) add(dL_dsh, M * 3);
            la.n = n;
            launch_late_fill = [=]() -> int {
                if (late_fork.word) GS_HIP(hipStreamWaitValue32(side->stream, late_fork.word, late_fork.seq, hipStreamWaitValueGte, 0xFFFFFFFFu));
                if (g_ablate.load() != 3)      // (3, experiments only: the step without the zero rows -- what a caller with persistent outputs could save)
                {   ProfScope ps(K_LATE_ZERO, side->stream);
                    late_rows_zero_kernel<<<GSRAST_LATE_FILL_WGS, 256, 0, side->stream>>>(P, at<unsigned long long>(geom, GL.color_skip), at<uint32_t>(geom, GL.scalars), la, at<unsigned char>(geom, GL.untouched)); }
                hipError_t e = hipGetLastError();
                if (e != hipSuccess) return fail(GSRAST_E_DEVICE, "late_rows_zero", e);
                GS_HIP(hipEventRecord(side->join, side->stream));
                return GSRAST_OK;
            };
            if (!late_fork.word) { int rc = launch_late_fill(); if (rc != GSRAST_OK) return rc; }       // (word fork: behind the blend backward's launch, below)
            late_fill = true;
        }
    }
    if (do_blend && R > 0) {
        ProfScope ps(K_BLEND_BWD, s);
        const uint32_t grid = ((T + 7) / 8) * 8;
        const uint2* ranges = at<uint2>(img, IL.ranges);
        const float* fT = at<float>(img, IL.final_T); const uint32_t* nc = at<uint32_t>(img, IL.n_contrib);
        const uint32_t* tm = at<uint32_t>(img, IL.tile_max);
        BlendArgs ba{};
        ba.ranges = ranges; ba.plist = plist; ba.W = W; ba.H = H; ba.gx = cam.gx; ba.T = T; ba.r0 = rec0; ba.r1 = rec1; ba.r2 = rec2;
        ba.bg = background; ba.fT = const_cast<float*>(fT); ba.nc = const_cast<uint32_t*>(nc); ba.tm = const_cast<uint32_t*>(tm);
        ba.dpix = dL_dpix; ba.grec = grec;
        ba.fork_word = late_fork.word; ba.fork_seq = late_fork.seq;
        if (const int mut = g_mutate.load()) {      // tests only: see g_mutate
            if (mut & 1) {
                mutate_drop_front_batch_kernel<<<1, 256, 0, s>>>(at<uint2>(img, IL.ranges), at<uint32_t>(img, IL.n_contrib), at<uint32_t>(img, IL.tile_max),
                                                                 (uint32_t)(cam.gy / 2) * (uint32_t)cam.gx + (uint32_t)(cam.gx / 2), W, H, cam.gx);
                GS_LAUNCHED("mutate_drop_front_batch");
            }
            if (mut & 2) {
                static float* zero_bg = nullptr;      // (never freed: a test-only path)
                if (!zero_bg) { GS_HIP(hipMalloc((void**)&zero_bg, 16)); GS_HIP(hipMemset(zero_bg, 0, 16)); }
                ba.bg = zero_bg;
            }
        }
        const int ppl = pick_ppl(T, true, o);
        const bool cull = o.cull != 0;
        if (cull && o.lpt) {
            if (T <= BUCKET_MAX_TILES) {      // the forward blend appended every tile to the backward work buckets
                ba.bcnt = const_cast<uint32_t*>(at<uint32_t>(img, IL.bucket_cnt)); ba.blist = const_cast<uint16_t*>(at<uint16_t>(img, IL.bucket_list));
                ba.from_buckets = 1;
            } else {
                uint32_t* ord = at<uint32_t>(img, IL.order_bwd);
                tile_order_kernel<<<1, 1024, 0, s>>>(T, ranges, tm, ord);
                GS_LAUNCHED("tile_order");
                ba.order = ord;
            }
        }
        if (g_ablate.load() == 1) launch_bwd<0, 4, 1>(grid, s, ba);
        else if (g_ablate.load() == 2) launch_bwd<0, 4, 2>(grid, s, ba);
        else
        switch (o.exp_mode) {
        case 0: if (cull) dispatch_bwd_cull<0>(ppl, grid, s, ba); else dispatch_bwd<0>(ppl, grid, s, ba); break;
        case 1: if (cull) dispatch_bwd_cull<1>(ppl, grid, s, ba); else dispatch_bwd<1>(ppl, grid, s, ba); break;
        default: if (cull) dispatch_bwd_cull<2>(ppl, grid, s, ba); else dispatch_bwd<2>(ppl, grid, s, ba); break;
        }
        GS_LAUNCHED("blend_bwd");
        // (belt and braces: should anything but the signalling kernel have been launched, the caller's stream releases the side stream itself)
        if (late_fork.word && !(g_ablate.load() == 0 && cull && ppl == 1 && g_bwd_transposed.load())) GS_HIP(hipStreamWriteValue32(s, late_fork.word, late_fork.seq, 0));
    }
    if (late_fork.word) { int rc = launch_late_fill(); if (rc != GSRAST_OK) return rc; }      // (its wait was released by the kernel just launched, or will be)
    if (do_blend && use_sh && o.sh_grad_factors) {      // dL_dsh is [P][3] in this mode: the factor, final after the blend backward
        sh_factor_kernel<<<(P + 255) / 256, 256, 0, s>>>(P, radii, at<unsigned char>(geom, GL.clamped), reinterpret_cast<const float4*>(grec), dL_dsh, at<uint32_t>(geom, GL.scalars), at<unsigned char>(geom, GL.untouched));
        GS_LAUNCHED("sh_factor");
    }
    // The zero rows and the per-Gaussian backward write DISJOINT rows (untouched / touched Gaussians, by the same bits): when they are all the side
    // stream carries, it is joined BEHIND the per-Gaussian backward -- that kernel neither waits for the last zero row nor pays the join's latency
    // in front of it (the table-off step showed it waiting 18 us for a fill that had started later than the blend backward)
    const bool join_late = side && late_fill && !side_has_derivs && do_geom && g_ablate.load() != 5 /* (5, experiments only: the join in front, as before) */;
    if (side && !join_late) { GS_HIP(hipStreamWaitEvent(s, side->join, 0)); side_guard.joined = true; }
    if (do_geom) {
        ProfScope ps(K_PREPROCESS_BWD, s);
        const float* cov = cov3D_precomp ? cov3D_precomp : at<float>(geom, GL.cov3D);
        const int pb_grid = (P + PP_THREADS - 1) / PP_THREADS;
        const float* sh_in = rawin ? shs : (use_sh ? shs : nullptr);
        const float* sc_in = rawin ? scales : (use_sr ? scales : nullptr);
        const float* ro_in = rawin ? rotations : (use_sr ? rotations : nullptr);
        const int factors = (use_sh && o.sh_grad_factors) ? 1 : 0;
#define GS_PB_ARGS P, D, M, means3D, radii, raw, rawg, sh_in, at<unsigned char>(geom, GL.clamped), at<float4>(geom, GL.shdA), at<float4>(geom, GL.shdB), \
                   at<float>(geom, GL.shdC), sc_in, ro_in, cov, cam, reinterpret_cast<const float4*>(grec), dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor,  \
                   dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot, factors, (late_fill ? at<unsigned long long>(geom, GL.color_skip) : nullptr), at<uint32_t>(geom, GL.scalars), at<unsigned char>(geom, GL.untouched)
        const bool skip = !o.dense_backward;        // Gaussians with an all-zero gradient record are not read
        if (late_fill) {       // (late_fill implies skip) grouped: 1024 Gaussians per workgroup, the ones late_rows_zero_kernel does not write compacted
            const int gg = (P + PB_GROUP - 1) / PB_GROUP;
            if (rawin) preprocess_bwd_kernel<true, true, true><<<gg, PP_THREADS, 0, s>>>(GS_PB_ARGS); else preprocess_bwd_kernel<false, true, true><<<gg, PP_THREADS, 0, s>>>(GS_PB_ARGS);
        } else
        if (rawin) { if (skip) preprocess_bwd_kernel<true, true><<<pb_grid, PP_THREADS, 0, s>>>(GS_PB_ARGS); else preprocess_bwd_kernel<true, false><<<pb_grid, PP_THREADS, 0, s>>>(GS_PB_ARGS); }
        else { if (skip) preprocess_bwd_kernel<false, true><<<pb_grid, PP_THREADS, 0, s>>>(GS_PB_ARGS); else preprocess_bwd_kernel<false, false><<<pb_grid, PP_THREADS, 0, s>>>(GS_PB_ARGS); }
#undef GS_PB_ARGS
        GS_LAUNCHED("preprocess_bwd");
    }
    if (join_late) { GS_HIP(hipStreamWaitEvent(s, side->join, 0)); side_guard.joined = true; }
    return GSRAST_OK;
}

int gsrast_backward_ex(const gsrast_options* options, int P, int D, int M, int R, const float* background, int width, int height,
                       const float* means3D, const float* shs, const float* colors_precomp, const float* scales,
                       float scale_modifier, const float* rotations, const float* cov3D_precomp,
                       const float* viewmatrix, const float* projmatrix, const float* campos, float tan_fovx,
                       float tan_fovy, const int* radii, char* geom_buffer, char* binning_buffer, char* image_buffer,
                       const float* dL_dpix, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor,
                       float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, void* stream)
{
    return backward_impl(options, P, D, M, R, background, width, height, means3D, shs, colors_precomp, scales, scale_modifier, rotations, cov3D_precomp,
                         viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, radii, geom_buffer, binning_buffer, image_buffer, dL_dpix, dL_dmean2D,
                         dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot, stream, nullptr, nullptr);
}

int gsrast_backward_raw(const gsrast_options* options, int P, int D, int M, int R, const float* background, int width, int height,
                        const gsrast_raw_inputs* in, float scale_modifier, const float* viewmatrix, const float* projmatrix, const float* campos,
                        float tan_fovx, float tan_fovy, const int* radii, char* geom_buffer, char* binning_buffer, char* image_buffer,
                        const float* dL_dpix, const gsrast_raw_grads* out, void* stream)
{
    if (const char* e = raw_inputs_check(P, M, in)) return fail(GSRAST_E_ARG, e);
    if (!out) return fail(GSRAST_E_ARG, "backward_raw: NULL gradient set");
    if (P == 0) return GSRAST_OK;
    if (!out->dL_dmean2D || !out->d_xyz || !out->d_rotation || !out->d_scaling || !out->d_opacity_logit) return fail(GSRAST_E_ARG, "backward_raw: NULL required gradient output");
    if ((in->rot_res != nullptr) != (out->d_rot_res != nullptr) && in->rot_res == nullptr) return fail(GSRAST_E_ARG, "backward_raw: d_rot_res without rot_res");
    if (out->d_shs_res && !in->shs_res) return fail(GSRAST_E_ARG, "backward_raw: d_shs_res without shs_res");
    const bool fac = out->d_sh_factor != nullptr;       // the SH leaves' gradient leaves as its [P][3] factor (multi-GPU exchange)
    if (fac && (in->shs_res || out->d_shs_res)) return fail(GSRAST_E_ARG, "backward_raw: d_sh_factor cannot be combined with shs_res / d_shs_res");
    if ((out->d_features_dc != nullptr) != (M > 1 ? out->d_features_rest != nullptr : out->d_features_dc != nullptr) || (!fac && !out->d_shs_res && !out->d_features_dc))
        return fail(GSRAST_E_ARG, "backward_raw: give d_features_dc + d_features_rest and / or (with shs_res) d_shs_res, whose rows hold both");
    if (((uintptr_t)out->d_rotation | (uintptr_t)out->d_features_dc | (uintptr_t)out->d_features_rest | (uintptr_t)out->d_shs_res) & 15)
        return fail(GSRAST_E_ARG, "backward_raw: d_rotation / d_features_dc / d_features_rest / d_shs_res must be 16-byte aligned");
    gsrast_options o = options ? *options : snapshot_defaults();
    o.sh_grad_factors = fac ? 1 : 0;
    gsrast_raw_grads og = *out;
    if (fac) { og.d_features_dc = nullptr; og.d_features_rest = nullptr; }     // (not written: the caller completes them after the exchange)
    out = &og;
    float* sh_marker = fac ? out->d_sh_factor : (out->d_shs_res ? out->d_shs_res : out->d_features_dc);
    return backward_impl(&o, P, D, M, R, background, width, height, in->xyz, in->features_dc, nullptr, in->scaling, scale_modifier, in->rotation, nullptr,
                         viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, radii, geom_buffer, binning_buffer, image_buffer, dL_dpix, out->dL_dmean2D,
                         nullptr, out->d_opacity_logit, nullptr, out->d_xyz, nullptr, sh_marker, out->d_scaling, out->d_rotation, stream, in, out);
}

int gsrast_debug_export(int P, int R, int width, int height, const char* geom_buffer, const char* binning_buffer,
                        const char* image_buffer, float* depths, float* means2D, float* cov3D, float* conic_opacity,
                        float* rgb, unsigned char* clamped, uint32_t* tiles_touched, uint64_t* keys_sorted,
                        uint32_t* point_list, uint32_t* ranges, float* final_T, uint32_t* n_contrib, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (P <= 0 || !geom_buffer) return fail(GSRAST_E_ARG, "debug_export: bad arguments");
    if (cov3D && !g_debug_state.load()) return fail(GSRAST_E_ARG, "debug_export: cov3D is only kept by forwards run with gsrast_set_option(\"debug_state\", 1)");
    const GeomLayout GL = geom_layout((size_t)P);
    const ImgLayout IL = img_layout((size_t)width, (size_t)height);
    const uint32_t T = (uint32_t)((width + TILE_X - 1) / TILE_X) * (uint32_t)((height + TILE_Y - 1) / TILE_Y);
    export_geom_kernel<<<(P + 255) / 256, 256, 0, s>>>(
        P, at<float4>(geom_buffer, GL.rec0), at<float4>(geom_buffer, GL.rec1),
        at<float4>(geom_buffer, GL.rec2), at<float>(geom_buffer, GL.cov3D), at<unsigned char>(geom_buffer, GL.clamped),
        at<uint32_t>(geom_buffer, GL.tiles), depths, means2D, cov3D, conic_opacity, rgb, clamped, tiles_touched);
    GS_LAUNCHED("export_geom");
    if (R > 0 && binning_buffer && image_buffer && (keys_sorted || point_list)) {
        export_keys_kernel<<<T, 256, 0, s>>>(at<uint2>(image_buffer, IL.ranges), at<uint32_t>(binning_buffer, 0),
                                             at<float4>(geom_buffer, GL.rec1), keys_sorted, point_list, (uint32_t)R);
        GS_LAUNCHED("export_keys");
    }
    if (image_buffer) {
        const size_t N = (size_t)width * height;
        if (ranges) GS_HIP(hipMemcpyAsync(ranges, image_buffer + IL.ranges, (size_t)T * 8, hipMemcpyDeviceToDevice, s));
        if (final_T) GS_HIP(hipMemcpyAsync(final_T, image_buffer + IL.final_T, N * 4, hipMemcpyDeviceToDevice, s));
        if (n_contrib) GS_HIP(hipMemcpyAsync(n_contrib, image_buffer + IL.n_contrib, N * 4, hipMemcpyDeviceToDevice, s));
    }
    return GSRAST_OK;
}

// ---- fused L1 + D-SSIM loss (gsrast_loss.h) ---------------------------------------------------
namespace {
LossWin make_window()
{   // utils/loss_utils.py:25-27: exp() in double, stored as fp32, divided (fp32) by the fp32 sum.  That sum is the correctly
    // rounded one in the reference (torch's CPU reduction; a sequential fp32 sum lands 1 ulp lower): summed in double here,
    // then rounded -- checked bit for bit against the reference's own window (tests/golden/loss_vectors.npz: ref_window_1d)
    LossWin w; double dsum = 0.0;
    for (int i = 0; i < LW; i++) { w.w[i] = (float)exp(-(double)((i - LW / 2) * (i - LW / 2)) / (2.0 * 1.5 * 1.5)); dsum += (double)w.w[i]; }
    const float sum = (float)dsum;
    for (int i = 0; i < LW; i++) w.w[i] = w.w[i] / sum;
    return w;
}
struct LossLayout { size_t d_mu, d_e11, d_e12, partial, total; int nblk; };
LossLayout loss_layout(int C, int H, int W)
{
    LossLayout L; size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o = align256(o + bytes); return r; };
    const size_t n = (size_t)C * H * W;
    L.nblk = C * ((W + LT - 1) / LT) * ((H + LT - 1) / LT);
    L.d_mu = take(n * 4); L.d_e11 = take(n * 4); L.d_e12 = take(n * 4); L.partial = take((size_t)L.nblk * 8);
    L.total = o + 256;
    return L;
}
} // namespace

size_t gsrast_loss_scratch_bytes(int C, int H, int W) { return (C > 0 && H > 0 && W > 0) ? loss_layout(C, H, W).total : 256; }

int gsrast_loss_forward(int C, int H, int W, const float* img, const float* gt, float lambda_dssim, float* out3,
                        char* scratch, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (C <= 0 || H <= 0 || W <= 0 || !img || !gt || !out3 || !scratch) return fail(GSRAST_E_ARG, "loss_forward: bad argument");
    const LossLayout L = loss_layout(C, H, W);
    ProfScope ps(K_LOSS_FWD, s);
    loss_fwd_kernel<<<L.nblk, 256, 0, s>>>(C, H, W, img, gt, make_window(), at<float>(scratch, L.d_mu), at<float>(scratch, L.d_e11),
                                           at<float>(scratch, L.d_e12), at<float2>(scratch, L.partial));
    GS_LAUNCHED("loss_fwd");
    loss_reduce_kernel<<<1, 256, 0, s>>>(at<float2>(scratch, L.partial), L.nblk, 1.0f / ((float)C * (float)H * (float)W), lambda_dssim, out3);
    GS_LAUNCHED("loss_reduce");
    return GSRAST_OK;
}

int gsrast_loss_backward(int C, int H, int W, const float* img, const float* gt, float lambda_dssim, const float* dL_dloss,
                         const char* scratch, float* dL_dimg, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (C <= 0 || H <= 0 || W <= 0 || !img || !gt || !dL_dimg || !scratch) return fail(GSRAST_E_ARG, "loss_backward: bad argument");
    const LossLayout L = loss_layout(C, H, W);
    ProfScope ps(K_LOSS_BWD, s);
    loss_bwd_kernel<<<L.nblk, 256, 0, s>>>(C, H, W, img, gt, make_window(), at<float>(scratch, L.d_mu), at<float>(scratch, L.d_e11),
                                           at<float>(scratch, L.d_e12), lambda_dssim, 1.0f / ((float)C * (float)H * (float)W), dL_dloss, dL_dimg);
    GS_LAUNCHED("loss_bwd");
    return GSRAST_OK;
}

} // extern "C"

#ifdef GSRAST_SCATTER_TIMING
extern "C" int gsrast_debug_scatter_timing(unsigned long long* out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(gsrast::g_scat), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; z[14] = ~0ull; if (hipMemcpyToSymbol(HIP_SYMBOL(gsrast::g_scat), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#endif
#if defined(GSRAST_DEBUG_COUNTERS) || defined(GSRAST_DEBUG_TIMING)
extern "C" int gsrast_debug_counters(unsigned long long* out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(gsrast::g_dbg), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; z[11] = ~0ull /* (a minimum) */; if (hipMemcpyToSymbol(HIP_SYMBOL(gsrast::g_dbg), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#endif
