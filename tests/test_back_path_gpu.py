"""The back-substitution as one launch (k_chain_back_path, DESIGN 4.2: every bottom group recomputes the levels above it, no hand-over
between workgroups) against the level-by-level form (VICALIB_AMD_BACK_PATH=0: k_chain_back + k_chain_back_levels): the same dependent
chain per group; z + Y delta_s is summed in another (fixed) order -- every iteration's cost, the accept / reject sequence and the final
state at rounding level.  Frame counts: short, full and empty last groups at the bottom level (57, 60, 64, 65), one, two and three levels
below the top (9, 60, 130, 600), counts where upper-level groups are short or empty (65, 513), and a count without a level
below the top (7: both runs take the classic kernel and must agree bit for bit).  Round 6: wide borders -- cfg4's rig (4 x poly3, D = 67: the
rows [Y] staged in two rounds of columns) and cfg5's (8 cameras, D = 115: four rounds); sharded passes with their pinned separator / ghost
frames are held to the single-process solve and to the oracle by tests/test_sharding.py (which also keeps one run on the level-by-level
form)."""
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
    e.pop("VICALIB_AMD_BACK_PATH", None)
    e.update({k: str(v) for k, v in env.items()})
    r = subprocess.run([sys.executable, os.path.join(HERE, "sync_worker.py"), out, str(n_frames)], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return np.load(out)


@pytest.mark.parametrize("n_frames,rig", [(n, "kb4") for n in (9, 17, 57, 60, 64, 65, 130, 513, 577, 600)] +
                         [(66, "poly3,poly3,poly3,poly3"), (130, "poly3,poly3,poly3,poly3"), (66, "fov,kb4,fov,kb4,fov,kb4,fov,kb4")])
def test_one_launch_back_substitution_matches_the_levels(tmp_path, n_frames, rig):
    a = _run(tmp_path, "path", n_frames, VICALIB_AMD_BACK_PATH=1, VICALIB_TEST_MODELS=rig)
    b = _run(tmp_path, "levels", n_frames, VICALIB_AMD_BACK_PATH=0, VICALIB_TEST_MODELS=rig)
    assert int(a["timeouts"]) == 0 and int(b["timeouts"]) == 0
    ta, tb = a["trace"], b["trace"]
    if n_frames < 20:
        # a handful of frames: a slowly converging, badly conditioned problem (400-500 iterations) on which rounding-level differences
        # grow by orders of magnitude over the run (measured: identical for the first 30-40 iterations, 1e-3 at the end) -- the first
        # 25 iterations are compared
        ta, tb = ta[:25], tb[:25]
        assert len(ta) == 25 and len(tb) == 25
        np.testing.assert_array_equal(ta[:, 8], tb[:, 8])
        np.testing.assert_allclose(ta[:, 1], tb[:, 1], rtol=1e-9)
        return
    assert ta.shape == tb.shape and len(ta) > 10
    np.testing.assert_array_equal(ta[:, 8], tb[:, 8])                     # accept / reject
    np.testing.assert_allclose(ta[:, 1], tb[:, 1], rtol=1e-7)             # cost of every iteration
    np.testing.assert_allclose(a["K"], b["K"], rtol=1e-8)
    np.testing.assert_allclose(a["frames"], b["frames"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(a["biases"], b["biases"], rtol=1e-7, atol=1e-11)
    assert abs(float(a["toff"]) - float(b["toff"])) < 1e-11


def test_one_launch_form_is_not_taken_without_a_level_below_the_top(tmp_path):
    a = _run(tmp_path, "path7", 7, VICALIB_AMD_BACK_PATH=1)
    b = _run(tmp_path, "levels7", 7, VICALIB_AMD_BACK_PATH=0)
    for k in ("trace", "K", "frames", "biases"):
        assert np.array_equal(a[k], b[k]), k
