// gsrast_policy.h -- the HOST-side decisions of the forward, free of any HIP call: when the list cut is applied, paused and widened, how
// the speculative launch is sized, how the depth histogram's range follows the scene.  gsrast_capi.hip enqueues; this file decides.
// Everything here runs on a CPU box: tests/test_policy.py drives it through gsrast_policy_event() (include/gsrast.h) on a context that
// never touches a device.  No result of a call depends on any of it -- only how much work the call enqueues.
#pragma once
#include <stdint.h>
#include <algorithm>
#include <atomic>
#include "gsrast_common.h"      // (the depth histogram's bin geometry: ZH_*, zh_bin_start)

namespace gsrast {

// counters of "that many forwards go without ..." shared by the lanes of a view-parallel caller: never below zero
inline void dec_to_zero(std::atomic<int>& a) { int v = a.load(); while (v > 0 && !a.compare_exchange_weak(v, v - 1)) { } }

// ---- the list cut's policy (gsrast_common.h: LIST CUT, PREDICTED CUT) ----------------------------------------------------------------
// The cut costs ~80 us per forward (the late test in the scatter, the compacting colour kernel, the predicated launches behind the blend)
// and saves ~50 us per million column runs it removes.
struct CutPolicy {
    static constexpr uint32_t MIN_RUNS = 1500000u;     // it is applied when the context's last forward had at least this many column runs
    static constexpr int PAUSE = 64;                   // forwards a context sits out after SMALL_STREAK cut forwards that removed fewer than MIN_RUNS
    static constexpr int SMALL_STREAK = 4;
    static constexpr int MARGIN_MIN = 6, MARGIN_MAX = 16;   // the next remembered cut sits (margin / 4) x as deep as the deepest entry consumed: 1.5 x ... 4 x
    static constexpr int TAU_MIN = 10, TAU_MAX = 96;        // mean optical depth a tile must have gathered in front of a PREDICTED cut (T < 1e-4 needs 9.2 at EVERY pixel; measured on
                                                            // the 3 M cube, 8 poses: no completion pass down to 8 -- the far bin edge and the 3 x 3 maximum are the slack --, passes at 4)
    static constexpr int TAU_FORCE = 512;                   // forwards that use predicted cut depths for every pose after a remembered one failed widely

    std::atomic<int> pause{0};             // > 0: that many forwards go without the cut ...
    std::atomic<uint32_t> pause_P{0};      // ... in a scene of this many Gaussians (another scene: the pause is void)
    // A cut list that turns out too short is completed by the pass behind the blend; the device reports each pass with its size.  Small
    // passes (up to an eighth of all column runs) are what the speculation is expected to cost; a quarter and more counts like a whole
    // second forward (8 points), in between in proportion.  16 points pause the cut -- 64 forwards, twice as long each further time (at
    // most 1024; after a pause the score restarts at 8: ONE more large pass pauses again), 64 cut forwards without a pass forget.
    std::atomic<int> fb_score{0}, fb_pause{0}, ok_streak{0}, small_streak{0};
    // Every reported pass widens the remembered cut's margin by half a step and raises the predicted cut's requirement by 8; 128 (64) cut
    // forwards without one take a quarter step (2) back.  A pass of 2 points or more while predicted cuts are available switches every
    // pose to them for TAU_FORCE forwards: the scene is another one at every visit, its remembered cuts are not to be trusted.
    std::atomic<int> margin{MARGIN_MIN}, margin_streak{0}, tau_req{TAU_MIN}, tau_force{0}, tau_streak{0}, tau_min{TAU_MIN};
    // ... and so does a RUN of small ones: every pass, whatever its size, is a chain of a dozen dependent launches the caller's stream waits
    // for (~0.1 ms); +8 per pass, -1 per clean cut forward, 24 = roughly one pass in six forwards for a while
    std::atomic<int> pass_rate{0};
    std::atomic<uint32_t> passes_reported{0};

    std::atomic<uint32_t> scene_P{0};      // the size of the scene the adaptive state above was learned on
    // start of a forward over P Gaussians: what the policy has learned belongs to the scene (size) it learned it on -- a pause, a widened
    // margin, a raised requirement, a switch to predicted cuts earned on one scene are void on another (a context that moves from a 3 M scene
    // to a 1 M one must not render the second with the first's scars: bench.py's sweep, a caller with several models)
    // Returns true when the scene is another one than the last forward's (the caller then also forgets the pose table's entries: cut
    // depths remembered for a pose of the OLD scene are running maxima and would take eight visits to fade).
    bool begin_forward(uint32_t P)
    {
        const uint32_t pp = pause_P.load();
        if (pause.load() > 0 && (pp > P ? pp - P : P - pp) > pp / 8) pause = 0;
        const uint32_t sp = scene_P.load();
        if (sp == 0u) { scene_P = P; return false; }
        if ((sp > P ? sp - P : P - sp) > sp / 8) {
            scene_P = P;
            fb_score = 0; fb_pause = 0; ok_streak = 0; small_streak = 0; margin = MARGIN_MIN; margin_streak = 0;
            tau_req = tau_min.load(); tau_force = 0; tau_streak = 0; pass_rate = 0;
            return true;
        }
        return false;
    }
    bool pays(uint32_t last_Q, bool always) const { return always || (last_Q >= MIN_RUNS && pause.load() == 0); }
    void sits_out() { dec_to_zero(pause); }             // a forward without the cut serves one forward of a pause
    bool forced_prediction() { const bool f = tau_force.load() > 0; dec_to_zero(tau_force); return f; }
    // the counts of a cut forward: Q column runs in all, Q_early listed
    void forward_counts(bool predicted_available, uint32_t n_late, uint32_t Q, uint32_t Q_early, uint32_t P, bool always)
    {
        if (always) return;
        if ((n_late != 0u || predicted_available) && Q - Q_early < MIN_RUNS) {
            if (++small_streak >= SMALL_STREAK) { small_streak = 0; pause = PAUSE; pause_P = P; }
        } else small_streak = 0;
    }
    static int pass_points(uint32_t q2, uint32_t qall)
    {
        qall = std::max(qall, 8u);
        const uint32_t lo = qall / 8u;
        return q2 <= lo ? 0 : (int)std::min<uint64_t>(8u, ((uint64_t)(q2 - lo) * 8u + lo - 1u) / lo);
    }
    // the device reported a completion pass over q2 column runs of candidates (of qall in the forward); returns its points
    int completion_pass(uint32_t q2, uint32_t qall, uint32_t P, bool always, bool predicted_available)
    {
        const int pts = pass_points(q2, qall);
        passes_reported++;
        if (pts == 0 && fb_score.load() > 0) fb_score--;
        if (pts >= 4) ok_streak = 0;
        { const int m = margin.load(); if (m < MARGIN_MAX) margin = std::min(MARGIN_MAX, m + 2); margin_streak = 0; }
        { const int r = tau_req.load(); if (r < TAU_MAX) tau_req = std::min(TAU_MAX, r + 8); tau_streak = 0; }
        if (predicted_available && (pts >= 2 || (pass_rate += 8) >= 24)) { tau_force = TAU_FORCE; pass_rate = 0; }
        if ((fb_score += pts) >= 16 && !always) {
            const int prev = fb_pause.load(), len = prev <= 0 ? 64 : (prev >= 512 ? 1024 : prev * 2);
            fb_pause = len; pause = len; pause_P = P; fb_score = 8;
        }
        return pts;
    }
    // a cut forward (late Gaussians > 0) behind which no pass was reported
    void clean_cut_forward()
    {
        if (fb_score.load() > 0) fb_score--;
        dec_to_zero(pass_rate);
        if (++ok_streak >= 64) fb_pause = 0;
        if (++margin_streak >= 128) { margin_streak = 0; const int m = margin.load(); if (m > MARGIN_MIN) margin = m - 1; }
        if (++tau_streak >= 64) { tau_streak = 0; const int r = tau_req.load(), lo = tau_min.load(); if (r > lo) tau_req = std::max(lo, r - 2); }
    }
};

// ---- the depth histogram's key range (gsrast_common.h: equalised depth buckets; the predicted cut's bins) ------------------------------
// The MIDDLE of the next forward's histogram covers this one's occupied key range padded by an eighth on either side (the tails beyond take
// a view a whole range away); the range widens at once and narrows by an eighth of the gap per forward (consecutive forwards render
// different views).  khi is the upper end BEFORE the bins' width is rounded up to a power of two: the predicted cut's 32 bins span [klo, khi].
struct DepthRange {
    std::atomic<uint32_t> klo{ZH_KLO_DEFAULT}; std::atomic<int> shift{ZH_SHIFT_DEFAULT}; std::atomic<uint32_t> khi{0};
    bool coarse() const { return shift.load() == ZH_SHIFT_DEFAULT && klo.load() == ZH_KLO_DEFAULT; }      // nothing learned yet: 4 bins per octave over every finite float
    // zb = first | last << 16 occupied bin of the histogram a forward filled with the table (call_klo, call_shift); 0xFFFFFFFF: no sample.
    // Returns whether that table was a learned one that held every key (an overflow of the depth buckets is then the scene's doing).
    bool learn(uint32_t zb, uint32_t call_klo, int call_shift)
    {
        if (zb == 0xFFFFFFFFu) return true;
        const uint32_t first = zb & 0xFFFFu, last = zb >> 16;
        const long long top = ZH_KEY_TOP, bot = ZH_KLO_DEFAULT;
        long long kmin = std::max(zh_bin_start(first, call_klo, call_shift), bot), kmax = std::min(zh_bin_start(last + 1u, call_klo, call_shift), top);
        if (kmax <= kmin) kmax = kmin + 1;
        const long long span = kmax - kmin;
        const bool was_coarse = call_shift == ZH_SHIFT_DEFAULT && call_klo == ZH_KLO_DEFAULT;
        const bool clipped = first == 0u || last >= (uint32_t)ZH_BINS - 1u;          // keys may lie beyond the table's tails
        const long long pad_lo = first == 0u ? 4 * span : span / 8 + 1, pad_hi = last >= (uint32_t)ZH_BINS - 1u ? 4 * span : span / 8 + 1;
        long long lo = std::max(kmin - pad_lo, bot), hi = std::min(kmax + pad_hi, top);
        if (!was_coarse) {      // (against the previous range -- its UN-rounded upper end if known: measured against the rounded one the range never narrowed below half the table)
            const uint32_t khi_prev = khi.load();
            const long long plo = call_klo, phi = khi_prev > call_klo ? (long long)khi_prev : (long long)call_klo + ((long long)ZH_MID << call_shift);
            lo = lo < plo ? lo : plo + (lo - plo) / 8;
            hi = hi > phi ? hi : phi - (phi - hi) / 8;
        }
        int sh = 0;
        while (((hi - lo) >> sh) >= (long long)ZH_MID) sh++;
        klo = (uint32_t)lo; shift = sh; khi = (uint32_t)hi;
        return !was_coarse && !clipped;
    }
};

// ---- capacities of the speculative launch --------------------------------------------------------------------------------------------
// The binning buffer is requested for 1.25 x (+ 4096) the hint BEFORE the host knows the counts; a hint follows the largest recent count
// and decays by 1 / 2^shift per forward (consecutive forwards render different views).
inline uint32_t grow_capacity(uint32_t v) { const uint64_t w = (uint64_t)v + v / 4 + 4096; return w > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)w; }
inline uint32_t follow_hint(uint32_t hint, uint32_t now, int shift) { const uint32_t d = hint - (hint >> shift); return now > d ? now : d; }
// column runs the launches over the CUT lists are sized for: half again as much as the largest early set of recent forwards of the same kind
// (remembered / predicted cut depths), all runs when nothing is known about this call's early set
inline uint32_t early_launch_runs(uint32_t qe_hint, uint32_t capQ, bool early_set_expected)
{
    if (!qe_hint || !early_set_expected) return capQ;
    return (uint32_t)std::min<uint64_t>(capQ, (uint64_t)qe_hint + qe_hint / 2 + 4096);
}

}  // namespace gsrast
