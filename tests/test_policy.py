"""CPU: the host-side decisions of a gsrast_context (saro-gs_amd/csrc/gsrast_policy.h) driven through gsrast_policy_event -- no device is
touched: when the list cut is applied, paused and put on probation, how its margin and the predicted cut's requirement follow the
completion passes the device reports, how the speculative launch is sized."""
import pytest


@pytest.fixture()
def ctx(rast):
    c = rast._C.Context()
    yield c
    c.close()


P, Q = 3_000_000, 8_700_000
PAUSE, SCORE, MARGIN, TAU_REQ, TAU_FORCE, FB_PAUSE = range(6)


def test_cut_pays_only_in_large_scenes_and_outside_a_pause(ctx):
    ev = ctx.policy_event
    assert ev("begin", P, 1_499_999, 0) == 0            # the previous forward had too few column runs: the cut does not pay
    assert ev("begin", P, 1_500_000, 0) == 1
    assert ev("begin", 10_000, 0, 1) == 1               # option list_cut_always (tests)
    assert ev("begin", P, Q, 0) == 1
    # four cut forwards in a row that removed fewer than 1.5 M column runs: the context sits out 64 forwards -- not after one, two or three
    for k in range(3):
        assert ev("counts", Q, Q - 1_000_000, 2) == 0, k
    assert ev("counts", Q, Q - 2_000_000, 2) == 0       # (one that paid: the streak starts again)
    for k in range(3):
        assert ev("counts", Q, Q - 1_000_000, 2) == 0
    assert ev("counts", Q, Q - 1_000_000, 2) == 64
    served = 0
    while ev("begin", P, Q, 0) == 0:                    # every forward without the cut serves one forward of the pause
        served += 1
    assert served == 64 and ev("get", PAUSE) == 0


def test_a_pause_belongs_to_the_scene_that_earned_it(ctx):
    ev = ctx.policy_event
    ev("begin", 1_000_000, Q, 0)
    for _ in range(4):
        ev("counts", Q, Q - 100, 2)
    assert ev("get", PAUSE) == 64
    assert ev("begin", 1_050_000, Q, 0) == 0            # about the same scene: the pause holds
    assert ev("begin", 3_000_000, Q, 0) == 1            # another scene: void


def test_what_was_learned_on_one_scene_is_void_on_another(ctx):
    ev = ctx.policy_event
    ev("begin", P, Q, 0)
    for _ in range(6):
        ev("pass", Q // 5, Q, 1)
    assert ev("get", MARGIN) == 16 and ev("get", TAU_REQ) > 40 and ev("get", TAU_FORCE) == 512
    ev("begin", P + P // 20, Q, 0)                      # the same scene after a densification step: kept
    assert ev("get", MARGIN) == 16
    ev("begin", 1_000_000, Q, 0)                        # another scene: a fresh policy
    assert ev("get", MARGIN) == 6 and ev("get", TAU_REQ) == 10 and ev("get", TAU_FORCE) == 0 and ev("get", SCORE) == 0


def test_completion_passes_are_scored_by_size(ctx):
    ev = ctx.policy_event
    ev("begin", P, Q, 0)
    assert ev("pass", Q // 8, Q, 0) == 0                # up to an eighth of all column runs: what the speculation is expected to cost
    assert ev("pass", Q // 4, Q, 0) == 8                # a quarter and more: like a whole second forward
    assert 3 <= ev("pass", 3 * Q // 16, Q, 0) <= 5      # in between in proportion
    assert ev("pass", Q, Q, 0) == 8


def test_two_large_passes_pause_the_cut_then_probation_doubles(ctx):
    ev = ctx.policy_event
    ev("begin", P, Q, 0)
    ev("pass", Q // 2, Q, 0)
    assert ev("get", PAUSE) == 0 and ev("get", SCORE) == 8
    ev("pass", Q // 2, Q, 0)
    assert ev("get", PAUSE) == 64 and ev("get", SCORE) == 8 and ev("get", FB_PAUSE) == 64      # on probation: the score restarts at half the bar
    while ev("begin", P, Q, 0) == 0:
        pass
    ev("pass", Q // 2, Q, 0)                            # ONE more large pass: paused again, twice as long
    assert ev("get", PAUSE) == 128
    for want in (256, 512, 1024, 1024):
        while ev("begin", P, Q, 0) == 0:
            pass
        ev("pass", Q // 2, Q, 0)
        assert ev("get", PAUSE) == want
    while ev("begin", P, Q, 0) == 0:
        pass
    for _ in range(64):                                 # 64 cut forwards without a pass: forgotten
        ev("clean")
    assert ev("get", FB_PAUSE) == 0 and ev("get", SCORE) == 0
    ev("pass", Q // 2, Q, 0)
    assert ev("get", PAUSE) == 0


def test_small_passes_do_not_pause(ctx):
    ev = ctx.policy_event
    ev("begin", P, Q, 0)
    for _ in range(200):
        assert ev("pass", Q // 50, Q, 0) == 0
    assert ev("get", PAUSE) == 0 and ev("get", SCORE) == 0


def test_margin_and_requirement_widen_with_passes_and_narrow_slowly(ctx):
    ev = ctx.policy_event
    ev("begin", P, Q, 0)
    assert ev("get", MARGIN) == 6 and ev("get", TAU_REQ) == 10
    ev("pass", Q // 50, Q, 1)
    assert ev("get", MARGIN) == 8 and ev("get", TAU_REQ) == 18
    for _ in range(20):
        ev("pass", Q // 50, Q, 1)
    assert ev("get", MARGIN) == 16 and ev("get", TAU_REQ) == 96          # 4 x the deepest consumed entry / 96 of mean alpha mass: the ends
    for _ in range(127):
        ev("clean")
    assert ev("get", MARGIN) == 16
    ev("clean")
    assert ev("get", MARGIN) == 15 and ev("get", TAU_REQ) == 92           # a quarter step per 128, 2 per 64 clean cut forwards
    for _ in range(128 * 12):
        ev("clean")
    assert ev("get", MARGIN) == 6 and ev("get", TAU_REQ) == 44


def test_a_wide_failure_switches_to_predicted_cut_depths(ctx):
    ev = ctx.policy_event
    ev("begin", P, Q, 0)
    ev("pass", Q // 2, Q, 0)                            # no predicted cut available (option tau_cut 0): nothing to switch to
    assert ev("get", TAU_FORCE) == 0
    ev("pass", Q // 50, Q, 1)                           # a small pass: the remembered cuts are fine
    assert ev("get", TAU_FORCE) == 0
    ev("pass", Q // 5, Q, 1)
    assert ev("get", TAU_FORCE) == 512


def test_frequent_small_passes_switch_to_predicted_cut_depths_too(ctx):
    ev = ctx.policy_event
    ev("begin", P, Q, 0)
    for _ in range(40):                                 # one small pass in twenty forwards: the price of the speculation
        ev("pass", Q // 50, Q, 1)
        for _ in range(19):
            ev("clean")
    assert ev("get", TAU_FORCE) == 0
    for k in range(12):                                 # one in five: every pass is a chain of launches the caller waits for
        ev("pass", Q // 50, Q, 1)
        for _ in range(4):
            ev("clean")
    assert ev("get", TAU_FORCE) == 512


def test_speculative_launch_sizes(ctx):
    ev = ctx.policy_event
    assert ev("grow", 1_000_000) == 1_254_096           # 1.25 x + 4096
    assert ev("grow", 2_000_000_000) == 0x7FFFFFFF
    assert ev("follow", 1600, 2000, 4) == 2000          # a hint follows a larger count at once ...
    assert ev("follow", 1600, 100, 4) == 1500           # ... and decays by a sixteenth per forward otherwise
    assert ev("size", 0, 5_000_000, 1) == 5_000_000     # nothing known about the early set: all column runs
    assert ev("size", 1_000_000, 5_000_000, 0) == 5_000_000
    assert ev("size", 1_000_000, 5_000_000, 1) == 1_504_096
    assert ev("size", 4_000_000, 5_000_000, 1) == 5_000_000
    with pytest.raises(ValueError):
        ev("no_such_event")


def test_depth_range_widens_at_once_and_narrows_slowly(ctx):
    """gsrast_policy.h: DepthRange -- the key range the depth histogram's middle bins (and the predicted cut's 32 bins) cover."""
    ev = ctx.policy_event
    ZH_TAIL, ZH_MID, ZH_BINS = 64, 896, 1024
    klo0, sh0 = ev("zget", 0), ev("zget", 1)
    assert sh0 == 21 and ev("zget", 2) == 0              # nothing learned: 4 bins per octave over every finite float, no upper end
    # a context's first forward (coarse table): the scene occupies bins 78-83 = depths 2.4 ... 6.7 -- not "held" (the table was not a learned one)
    r = ev("zrange", 78 | (83 << 16))
    assert not (r & 256)
    sh1, klo1, khi1 = ev("zget", 1), ev("zget", 0), ev("zget", 2)
    assert sh1 < sh0 and klo1 > klo0 and khi1 > klo1     # the table now resolves the scene: finer bins, a range around it
    span1 = khi1 - klo1
    # the same scene seen through the learned table occupies the middle (an eighth of padding on either side): held, and the range stays put
    first = ZH_TAIL + ZH_MID // 10
    last = ZH_TAIL + int(0.9 * ((span1 * 256) >> sh1))       # (zget returns keys / 256)
    r = ev("zrange", first | (last << 16))
    assert r & 256
    # a view whose keys reach the table's last bin: clipped (not held), the range widens AT ONCE, far beyond the old end
    r = ev("zrange", first | ((ZH_BINS - 1) << 16))
    assert not (r & 256) and ev("zget", 2) > khi1
    wide = ev("zget", 2) - ev("zget", 0)
    # ... and narrows by an eighth of the gap per forward once the views are narrow again
    prev = wide
    for _ in range(40):
        sh = ev("zget", 1)
        ev("zrange", (ZH_TAIL + 100) | ((ZH_TAIL + 200) << 16))
        now = ev("zget", 2) - ev("zget", 0)
        assert now <= prev
        prev = now
    assert prev < wide // 4
