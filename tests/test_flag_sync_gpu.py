"""The device-flag hand-overs between the two streams of a visual-inertial pass must never change a result: a wait that
runs into its bound is reported, the pass it belongs to is not judged, and the solve is resumed with event hand-overs -- same iterates,
bit for bit, as a run that used events from the start (verdict r3 weak #2, advice r3)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(tmp_path, name, **env):
    out = str(tmp_path / (name + ".npz"))
    e = dict(os.environ)
    for k in ("VICALIB_AMD_FLAG_SYNC", "VICALIB_AMD_SYNC_BOUND", "VICALIB_AMD_SYNC_BOUND_FROM_PASS", "VICALIB_AMD_STREAM2_PRIORITY", "GPU_MAX_HW_QUEUES", "VICALIB_AMD_BATCHED", "VICALIB_AMD_GRAPHS", "VICALIB_AMD_BACK_FUSED", "VICALIB_AMD_NO_MERGED_DECISION"):
        e.pop(k, None)
    e.update({k: str(v) for k, v in env.items()})
    r = subprocess.run([sys.executable, os.path.join(HERE, "sync_worker.py"), out], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return np.load(out), r.stderr


def _same(a, b):
    for k in ("trace", "K", "T", "frames", "biases", "toff"):
        assert np.array_equal(a[k], b[k]), "%s differs: max |d| = %g" % (k, np.max(np.abs(np.asarray(a[k]) - np.asarray(b[k]))))


@pytest.fixture(scope="module")
def events_run(tmp_path_factory):
    ref, _ = _run(tmp_path_factory.mktemp("sync"), "events", VICALIB_AMD_FLAG_SYNC=0)
    assert int(ref["timeouts"]) == 0 and len(ref["trace"]) > 20
    return ref


def test_flag_handovers_give_the_iterates_of_event_handovers(events_run, tmp_path):
    got, err = _run(tmp_path, "flags")
    assert int(got["timeouts"]) == 0, err
    _same(got, events_run)


@pytest.mark.parametrize("bound", [1, 10, 40])
def test_a_timed_out_flag_wait_is_loud_and_lossless(events_run, tmp_path, bound):
    """A bound of a few polls makes waits give up that would have been satisfied microseconds later -- on every kind of hand-over,
    depending on the bound: the device must withhold the decision, the host must say so and resume with events."""
    got, err = _run(tmp_path, "bound%d" % bound, VICALIB_AMD_SYNC_BOUND=bound)
    assert int(got["timeouts"]) >= 1, "bound %d never hit" % bound
    assert "ran into its bound" in err
    _same(got, events_run)


@pytest.mark.parametrize("from_pass,batched", [(n, b) for b in (0, 1) for n in range(3, 49, 5)])
def test_a_time_out_in_the_middle_of_a_solve_is_lossless(events_run, tmp_path, from_pass, batched):
    """The time-out hits a pass whose accepted state differs from its predecessor's (the waits of the first passes succeed: the tiny
    bound only applies from pass `from_pass` of the calibrator on).  What the resumed pass reads -- the IMU weights its predecessor
    left, in particular -- must not have been touched by the void pass or by the passes queued behind it (round 4: those passes'
    weight kernels copied / rewrote the buffer the resumed pass linearises with; a first-pass time-out never showed it, both buffers
    hold the same numbers there), and inside a streak of rejected steps the resumed pass must keep the linearisation in place (made with
    the weights of the pass that accepted the state) instead of linearising again with the current ones.  Passes 3, 8, .. 48 of the
    calibrator's ~55: every stage, first passes, accepted and rejected steps, passes queued past a stage's end."""
    env = dict(VICALIB_AMD_SYNC_BOUND=1, VICALIB_AMD_SYNC_BOUND_FROM_PASS=from_pass)
    if batched:
        env["VICALIB_AMD_BATCHED"] = 1
    got, err = _run(tmp_path, "mid%d_%d" % (from_pass, batched), **env)
    assert int(got["timeouts"]) >= 1 and "ran into its bound" in err, (int(got["timeouts"]), err[-300:])
    _same(got, events_run)


def test_timed_out_flag_wait_on_the_batched_schedule(events_run, tmp_path):
    got, err = _run(tmp_path, "batched", VICALIB_AMD_SYNC_BOUND=40, VICALIB_AMD_BATCHED=1)
    assert int(got["timeouts"]) >= 1 and "ran into its bound" in err
    _same(got, events_run)


def test_both_streams_on_one_hardware_queue(events_run, tmp_path):
    """The configuration the flags must not be used in -- one hardware queue for both streams, the second stream in the default
    priority class, flags forced on: whether or not a wait starves there, the results are those of the event hand-overs."""
    got, err = _run(tmp_path, "one_queue", GPU_MAX_HW_QUEUES=1, VICALIB_AMD_STREAM2_PRIORITY="default", VICALIB_AMD_FLAG_SYNC=1, VICALIB_AMD_SYNC_BOUND=20000)
    _same(got, events_run)
    print("one hardware queue: %d time-out(s)" % int(got["timeouts"]))


def test_graph_replay_of_the_visual_inertial_pass_gives_the_same_iterates(events_run, tmp_path):
    """VICALIB_AMD_GRAPHS=1 replays a captured pass: its kernel arguments are frozen, the pass number among them.  The fused
    back-substitution (k_chain_back_levels) orders producers and consumers by comparing per-frame ready words with that number, so from
    the second replay on its consumers would not wait (advice r4) -- a captured pass runs one launch per level instead.  Same iterates as
    the event run, bit for bit: every cost, accept / reject decision, radius, the final cameras, frames, biases, time offset.
    A captured pass cannot take the merged decision of the vision-only stages either (pass k judged at the head of pass k + 1's
    k_frame_schur, from k_trial's per-workgroup partials: alternating control records are kernel arguments) and decides in k_final, which
    sums the per-frame step terms in its own order: one gain ratio of the vision-only stage comes out one unit in the last place apart
    (round 6: tools/diag_graph_replay.sh) -- the whole trace is bit-identical to an event run that decides in k_final as well
    (VICALIB_AMD_NO_MERGED_DECISION=1), and within 1e-12 of the default event run."""
    got, err = _run(tmp_path, "graphs", VICALIB_AMD_GRAPHS=1)
    assert int(got["timeouts"]) == 0, err
    again, _ = _run(tmp_path, "graphs2", VICALIB_AMD_GRAPHS=1)
    _same(got, again)                                       # replay is deterministic
    assert got["trace"].shape == events_run["trace"].shape
    for col in (0, 1, 7, 8, 9):                             # iteration, cost, radius, accepted, stage
        assert np.array_equal(got["trace"][:, col], events_run["trace"][:, col]), col
    np.testing.assert_allclose(got["trace"], events_run["trace"], rtol=1e-12, atol=1e-13)
    unmerged, _ = _run(tmp_path, "events_unmerged", VICALIB_AMD_FLAG_SYNC=0, VICALIB_AMD_NO_MERGED_DECISION=1)
    _same(got, unmerged)                                    # same decision kernel: the whole trace, bit for bit
    for k in ("K", "T", "frames", "biases", "toff"):
        assert np.array_equal(got[k], events_run[k]), k
