"""The C++ problem generator (vicalib_amd/csrc/vc_synth.cpp) against the numpy one (vicalib_amd/synth.py): same
specification, same counter-based random stream.  Integer outputs are identical, BASELINE cfg1's detections are
bit-equal, everything else agrees to the last bits of the two math libraries.  CPU only."""
import numpy as np
import pytest

from vicalib_amd import synth


CASES = [
    synth.BASELINE_CONFIGS["cfg1"],
    synth.Config(models=("kb4", "fov", "poly2"), n_frames=40, imu=True, seed=7, first_frame=13),
    synth.Config(models=("poly3", "linear"), grid="large", n_frames=30, imu=True, seed=9),
    synth.Config(models=("fov", "kb4", "fov"), n_frames=12, imu=True, seed=5, extrinsics_prior=True),
]


@pytest.mark.parametrize("cfg", CASES, ids=["cfg1", "mixed_rig_imu_shard", "large_grid_imu", "extrinsics_prior"])
def test_native_generator_matches_numpy(cfg):
    a, b = synth.generate(cfg), synth.generate_native(cfg)
    fa, fb = synth.flatten(a), synth.flatten(b)
    for k in range(3):                                   # tile frames, cameras, offsets: identical
        np.testing.assert_array_equal(fa[k], fb[k])
    np.testing.assert_array_equal(np.concatenate([t[2] for t in a.tiles]), b.flat[3])     # the same dots are visible
    np.testing.assert_allclose(fa[4], fb[4], rtol=0, atol=1e-11)                              # pixels
    np.testing.assert_array_equal(fa[3], fb[3])                                               # target points
    for name in ("cam_T_ck_gt", "cam_T_ck_init", "frame_time", "frame_T_wk_gt", "frame_T_wk_init", "frame_v_gt", "imu_t", "imu_gyro", "imu_accel"):
        x, y = getattr(a, name), getattr(b, name)
        if x is None:
            assert y is None
        else:
            np.testing.assert_allclose(x, y, rtol=0, atol=1e-13, err_msg=name)
    for x, y in zip(a.cam_K_gt + a.cam_K_init, b.cam_K_gt + b.cam_K_init):
        np.testing.assert_array_equal(x, y)
    if a.imu_gt is not None:
        for k in a.imu_gt:
            np.testing.assert_array_equal(np.asarray(a.imu_gt[k]), np.asarray(b.imu_gt[k]))
    assert a.n_obs == b.n_obs and len(a.tiles) == len(b.tiles)


def test_cfg1_detections_bit_equal():
    cfg = synth.BASELINE_CONFIGS["cfg1"]
    a, b = synth.generate(cfg), synth.generate_native(cfg)
    np.testing.assert_array_equal(synth.flatten(a)[4], synth.flatten(b)[4])


def test_shards_concatenate_to_the_whole():
    """Every rank generates its own frame range: the union is the unsharded problem, bit for bit."""
    whole = synth.generate_native(synth.Config(models=("fov", "kb4"), n_frames=64, imu=True, seed=21))
    parts = [synth.generate_native(synth.Config(models=("fov", "kb4"), n_frames=16, imu=True, seed=21, first_frame=16 * r)) for r in range(4)]
    np.testing.assert_array_equal(np.concatenate([p.flat[4] for p in parts]), whole.flat[4])
    np.testing.assert_array_equal(np.concatenate([p.flat[3] for p in parts]), whole.flat[3])
    np.testing.assert_array_equal(np.concatenate([p.frame_T_wk_init for p in parts]), whole.frame_T_wk_init)
    t = np.concatenate([p.imu_t for p in parts]); g = np.concatenate([p.imu_gyro for p in parts])
    _, first = np.unique(t, return_index=True)            # the shards' IMU ranges overlap by the 0.1 s margins
    sel = np.isin(whole.imu_t, t[first])
    np.testing.assert_array_equal(g[first][np.isin(t[first], whole.imu_t)], whole.imu_gyro[sel])


def test_baseline_sizes_native():
    """BASELINE cfg4 / cfg5 at full size: seconds on the host, observation counts as DESIGN.md states them."""
    p4 = synth.generate_native(synth.BASELINE_CONFIGS["cfg4"])
    assert len(p4.frame_time) == 10000 and len(p4.flat[0]) == 40000 and 2.0e7 < p4.n_obs < 3.6e7 + 1
    assert p4.grid_points.shape == (900, 3) and p4.flat[3].max() == 899
