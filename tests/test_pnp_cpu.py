"""Pose initialisation of the front-end row (PnP; call site vicalib-task.cc:335-348).  Host code behind the
C ABI -- no GPU involved, so these run in the CPU suite."""
import numpy as np
import pytest

from vicalib_amd import synth
from vicalib_amd.lib import pnp_planar, pnp_planar_ransac


def _inv(T):
    from scipy.spatial.transform import Rotation as R
    r = R.from_quat(T[:4]).inv()
    return np.concatenate([r.as_quat(), -r.apply(T[4:])])


def _mul(A, B):
    from scipy.spatial.transform import Rotation as R
    ra, rb = R.from_quat(A[:4]), R.from_quat(B[:4])
    return np.concatenate([(ra * rb).as_quat(), ra.apply(B[4:]) + A[4:]])


@pytest.mark.parametrize("model", ["fov", "poly2", "poly3", "kb4", "linear", "rational6"])
def test_pnp_recovers_the_generating_pose(model):
    p = synth.generate(synth.Config(models=(model,), n_frames=12, seed=3, pixel_sigma=0.0))
    for (f, c, ids, pix) in p.tiles:
        T, rms = pnp_planar(model, p.cam_K_gt[c], p.grid_points[ids], pix)
        T_cw_gt = _mul(_inv(p.cam_T_ck_gt[c]) if False else p.cam_T_ck_gt[c], _inv(p.frame_T_wk_gt[f]))   # p_c = T_ck T_wk^-1 p_w
        assert rms < 1e-6
        sgn = np.sign(T[:4] @ T_cw_gt[:4])
        np.testing.assert_allclose(T[:4] * sgn, T_cw_gt[:4], atol=1e-7)
        np.testing.assert_allclose(T[4:], T_cw_gt[4:], atol=1e-7)


def test_pnp_with_the_engine_start_values_gives_a_usable_seed():
    """With the reference's start intrinsics (300, 300, w/2, h/2, ... vicalib-engine.cc:207-257) the seed pose is rough but
    on the right side of the target and finite for every view."""
    p = synth.generate(synth.Config(models=("poly3",), n_frames=10, seed=5))
    for (f, c, ids, pix) in p.tiles:
        T, rms = pnp_planar("poly3", p.cam_K_init[c], p.grid_points[ids], pix)
        assert np.all(np.isfinite(T)) and abs(np.linalg.norm(T[:4]) - 1) < 1e-12
        T_cw_gt = _mul(p.cam_T_ck_gt[c], _inv(p.frame_T_wk_gt[f]))
        assert T[6] > 0 and abs(T[:4] @ T_cw_gt[:4]) > 0.9      # within ~50 degrees, target in front


def test_pnp_rejects_degenerate_input():
    from vicalib_amd.lib import VicalibError
    pw = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0.0]]); uv = np.zeros((3, 2))
    with pytest.raises(VicalibError):
        pnp_planar("linear", [300, 300, 320, 240.0], pw, uv)                       # fewer than 4 corners
    pw = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0.5]]); uv = np.random.default_rng(0).normal(size=(4, 2))
    with pytest.raises(VicalibError):
        pnp_planar("linear", [300, 300, 320, 240.0], pw, uv)                       # not planar


@pytest.mark.parametrize("model", ["fov", "poly3", "kb4"])
def test_ransac_pose_survives_mismatched_dots(model):
    """20 % of the dots of every view carry the pixel of ANOTHER dot (a wrong grid association, what the robust branch of
    calibu::PosePnPRansac, vicalib-task.cc:323-325, is there for): the plain fit is dragged away, the consensus fit recovers
    the generating pose and flags exactly the corrupted correspondences."""
    p = synth.generate(synth.Config(models=(model,), n_frames=8, seed=11, pixel_sigma=0.05))
    rng = np.random.default_rng(4)
    worse = 0
    for (f, c, ids, pix) in p.tiles:
        n = len(ids)
        bad = rng.choice(n, size=max(1, n // 5), replace=False)
        pix2 = pix.copy()
        pix2[bad] = pix[(bad + 7 + rng.integers(0, n - 14, size=len(bad))) % n]      # another dot's detection
        changed = np.linalg.norm(pix2 - pix, axis=1) > 3.0
        T_cw_gt = _mul(p.cam_T_ck_gt[c], _inv(p.frame_T_wk_gt[f]))
        T0, rms0 = pnp_planar(model, p.cam_K_gt[c], p.grid_points[ids], pix2)
        T, rms, inl = pnp_planar_ransac(model, p.cam_K_gt[c], p.grid_points[ids], pix2, iterations=100, tol_px=1.0)
        sgn = np.sign(T[:4] @ T_cw_gt[:4])
        np.testing.assert_allclose(T[:4] * sgn, T_cw_gt[:4], atol=2e-3)
        np.testing.assert_allclose(T[4:], T_cw_gt[4:], atol=2e-3)
        assert rms < 0.2 and not np.any(inl & changed) and inl.sum() >= (~changed).sum() - 2
        worse += np.linalg.norm(T0[4:] - T_cw_gt[4:]) > 5 * np.linalg.norm(T[4:] - T_cw_gt[4:])
    assert worse >= len(p.tiles) // 2            # the non-robust fit is visibly off on most views


def test_ransac_with_zero_iterations_is_the_plain_fit():
    p = synth.generate(synth.Config(models=("poly3",), n_frames=3, seed=2))
    f, c, ids, pix = p.tiles[0]
    T0, r0 = pnp_planar("poly3", p.cam_K_gt[c], p.grid_points[ids], pix)
    T1, r1, inl = pnp_planar_ransac("poly3", p.cam_K_gt[c], p.grid_points[ids], pix, iterations=0, tol_px=0.0)
    np.testing.assert_array_equal(T0, T1)
    assert r0 == r1 and inl.all()
