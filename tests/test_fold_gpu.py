"""The chain assembly folded into the bottom level of the elimination (k_chain_l0) against the two kernels apart
(VICALIB_AMD_FOLD_L0=0: k_chain_init + k_chain_fwd2): the same arithmetic per frame, the chunk sums of the camera / IMU-parameter blocks added
in a different order -- every iteration's cost, the accept / reject sequence and the final state at rounding level.  Frame counts: groups of
8 with a short last group of every length (57 .. 64 frames), one level and two levels below the top, and a count the fold does not serve
(7 frames: no level below the top -- both runs take the classic path and must agree bit for bit)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(tmp_path, name, n_frames, **env):
    out = str(tmp_path / (name + ".npz"))
    e = dict(os.environ)
    e.pop("VICALIB_AMD_FOLD_L0", None)
    e.update({k: str(v) for k, v in env.items()})
    r = subprocess.run([sys.executable, os.path.join(HERE, "sync_worker.py"), out, str(n_frames)], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return np.load(out)


@pytest.mark.parametrize("n_frames", [57, 58, 59, 60, 61, 62, 63, 64, 65, 130, 520])
def test_fold_matches_the_two_kernels(tmp_path, n_frames):
    a = _run(tmp_path, "fold", n_frames, VICALIB_AMD_FOLD_L0=1)
    b = _run(tmp_path, "apart", n_frames, VICALIB_AMD_FOLD_L0=0)
    assert int(a["timeouts"]) == 0 and int(b["timeouts"]) == 0
    ta, tb = a["trace"], b["trace"]
    assert ta.shape == tb.shape and len(ta) > 20
    np.testing.assert_array_equal(ta[:, 8], tb[:, 8])                     # accept / reject
    np.testing.assert_allclose(ta[:, 1], tb[:, 1], rtol=1e-7)             # cost of every iteration
    np.testing.assert_allclose(a["K"], b["K"], rtol=1e-8)
    np.testing.assert_allclose(a["frames"], b["frames"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(a["biases"], b["biases"], rtol=1e-7, atol=1e-11)
    assert abs(float(a["toff"]) - float(b["toff"])) < 1e-11


def test_fold_is_not_taken_where_it_does_not_apply(tmp_path):
    a = _run(tmp_path, "fold7", 7, VICALIB_AMD_FOLD_L0=1)
    b = _run(tmp_path, "apart7", 7, VICALIB_AMD_FOLD_L0=0)
    for k in ("trace", "K", "frames", "biases"):
        assert np.array_equal(a[k], b[k]), k
