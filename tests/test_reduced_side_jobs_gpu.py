"""What round 6 moved out of k_reduced (one workgroup the whole chip waits for), against k_reduced doing it itself:
* the trial cameras and the shared parameters' terms of the step scalars, formed by one extra workgroup of the back-substitution's launch
  (vc_reduced_tail.hpp; VICALIB_AMD_DEFER_TAIL=0: inside k_reduced) -- the same code on the same lanes: every number identical;
* the cameras' blocks, the IMU-parameter block and the chunk costs of the reduced system, formed ahead of k_reduced by side jobs of the
  chain's upper-level launches (vc_shared_blocks.hpp; VICALIB_AMD_HADD_EARLY=0: inside k_reduced) -- the same terms, the chunk records'
  sums and the additions into S in another order: costs, accept / reject sequence and final state at rounding level.
Frame counts: 130 (one level above the bottom one: the record rides in the top level's launch, the tail stays in k_reduced), 520 and 600
(two levels: both moves), mono and stereo."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(tmp_path, name, n_frames, models="kb4", **env):
    out = str(tmp_path / (name + ".npz"))
    e = dict(os.environ)
    for k in ("VICALIB_AMD_DEFER_TAIL", "VICALIB_AMD_HADD_EARLY"):
        e.pop(k, None)
    e["VICALIB_TEST_MODELS"] = models
    e.update({k: str(v) for k, v in env.items()})
    r = subprocess.run([sys.executable, os.path.join(HERE, "sync_worker.py"), out, str(n_frames)], env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return np.load(out)


@pytest.mark.parametrize("n_frames,models", [(520, "kb4"), (600, "fov,poly3")])
def test_tail_in_the_back_substitution_launch_is_bit_identical(tmp_path, n_frames, models):
    a = _run(tmp_path, "deferred", n_frames, models, VICALIB_AMD_DEFER_TAIL=1)
    b = _run(tmp_path, "inside", n_frames, models, VICALIB_AMD_DEFER_TAIL=0)
    assert int(a["timeouts"]) == 0 and int(b["timeouts"]) == 0
    assert len(a["trace"]) > 20
    for k in ("trace", "K", "T", "frames", "biases", "toff"):
        assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("n_frames,models", [(130, "kb4"), (520, "kb4"), (600, "fov,poly3")])
def test_shared_blocks_formed_ahead_of_the_reduced_solve(tmp_path, n_frames, models):
    a = _run(tmp_path, "ahead", n_frames, models, VICALIB_AMD_HADD_EARLY=1)
    b = _run(tmp_path, "inside", n_frames, models, VICALIB_AMD_HADD_EARLY=0)
    assert int(a["timeouts"]) == 0 and int(b["timeouts"]) == 0
    ta, tb = a["trace"], b["trace"]
    assert ta.shape == tb.shape and len(ta) > 20
    np.testing.assert_array_equal(ta[:, 8], tb[:, 8])                     # accept / reject
    np.testing.assert_allclose(ta[:, 1], tb[:, 1], rtol=1e-7)             # cost of every iteration
    np.testing.assert_allclose(a["K"], b["K"], rtol=1e-8)
    np.testing.assert_allclose(a["frames"], b["frames"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(a["biases"], b["biases"], rtol=1e-7, atol=1e-11)
    assert abs(float(a["toff"]) - float(b["toff"])) < 1e-11
