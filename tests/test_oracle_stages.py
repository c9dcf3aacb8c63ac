"""The oracle's restatement of the calibrator's bookkeeping around the solver (SolveThread, vicalibrator.h:919-1040):
per-camera RMSE and mse, gravity initialisation, outlier removal, repeated solves after NO_CONVERGENCE -- each checked
against an independent numpy evaluation of what the reference text says, on the oracle's own residuals."""
import copy

import numpy as np
from scipy.spatial.transform import Rotation

import oracle_lib as ol
from vicalib_amd import synth


def _vision(models=("fov", "poly3"), n=14, seed=3):
    return synth.generate(synth.Config(models=models, n_frames=n, seed=seed))


def test_camera_rmse_and_mse_definitions():
    """camera_proj_rmse_ = sqrt(cost / n_blocks) with Ceres' cost = 1/2 sum r^2 and no loss (:958-971) -> the 1/2 stays
    inside the root and the division is by blocks, not scalars; mse_ = final_cost / num_residuals (scalars) (:975)."""
    p = _vision()
    o = ol.Oracle().load(p)
    o.set_options(calibrate_imu=False)
    o.solve()
    r, f, c = o.residuals()
    for cam in range(2):
        m = c == cam
        want = np.sqrt(0.5 * (r[m] ** 2).sum() / m.sum())
        assert abs(o.rmse()[cam] - want) <= 1e-13 * want
    tr = o.trace()
    assert abs(o.mse() - tr[-1, 1] / (2 * len(r))) <= 1e-12 * o.mse()
    # the robustified cost of the trace is what SoftLOneLoss(0.5) makes of those residuals (one copy of every block)
    s = (r * r).sum(axis=1)
    assert abs(tr[-1, 1] - 0.5 * (2 * 0.25 * (np.sqrt(1 + s / 0.25) - 1)).sum()) <= 1e-9 * tr[-1, 1]


def test_gravity_initialisation_formula():
    """:927-949: g_b = normalised accelerometer sample interpolated at the middle frame's time (no offset), g_w = R_wk g_b,
    p = asin(g_w.y), q = asin(-g_w.x / cos p)."""
    p = synth.generate(synth.Config(models=("kb4",), n_frames=21, imu=True, seed=9))
    o = ol.Oracle().load(p, init=False)
    o.L.vco_init_gravity(o.h)
    g = o.imu_state()[2]
    idx = len(p.frame_time) // 2
    t = p.frame_time[idx]
    k = int(np.searchsorted(p.imu_t, t, side="right")) - 1
    fr = (t - p.imu_t[k]) / (p.imu_t[k + 1] - p.imu_t[k])
    a = p.imu_accel[k] * (1 - fr) + p.imu_accel[k + 1] * fr
    gw = Rotation.from_quat(p.frame_T_wk_gt[idx][:4]).apply(a / np.linalg.norm(a))
    pp = np.arcsin(gw[1]); qq = np.arcsin(-gw[0] / np.cos(pp))
    np.testing.assert_allclose(g, [pp, qq], rtol=1e-12, atol=1e-14)
    # and it is a sensible estimate of the generator's gravity direction (sensor bias and motion limit it to ~0.1 rad)
    assert np.abs(g - p.imu_gt["g_dir"]).max() < 0.15


def _with_outliers(p, every=37, shift=25.0):
    q = copy.deepcopy(p)
    k = 0
    tiles = []
    for (f, c, ids, pix) in q.tiles:
        pix = pix.copy()
        for i in range(len(pix)):
            if k % every == 0:
                pix[i] += shift * np.array([1.0, -0.6])
            k += 1
        tiles.append((f, c, ids, pix))
    q.tiles = tiles
    return q


def test_outlier_removal_stage():
    """FLAGS_remove_outliers (:995-998, :1024-1027, RemoveOutliers :859-916): after the first converged solve the blocks
    whose unrobustified residual norm exceeds outlier_threshold x camera rmse leave the problem, then one more solve."""
    clean = _vision(models=("poly3",), n=20, seed=4)
    p = _with_outliers(clean)
    plain = ol.Oracle().load(p); plain.set_options(calibrate_imu=False); plain.solve()
    rem = ol.Oracle().load(p); rem.set_options(calibrate_imu=False, remove_outliers=True, outlier_threshold=2.0); rem.solve()
    ref = ol.Oracle().load(clean); ref.set_options(calibrate_imu=False); ref.solve()
    tp, tr = plain.trace(), rem.trace()
    assert set(tp[:, 9]) == {0.0} and set(tr[:, 9]) == {0.0, 1.0}         # one solve / two solves
    # with the gross errors in, the (unrobustified) rmse is dominated by them; with them removed it is at the noise floor
    assert plain.rmse()[0] > 1.0
    assert rem.rmse()[0] < 0.12
    # the latest copy's rmse is computed on the kept blocks only: every kept block is within threshold x (rmse at removal
    # time) -- bounded here by the first solve's rmse, which the plain run reproduces
    r, _, _ = rem.residuals()
    nrm = np.sqrt((r * r).sum(axis=1))
    kept = nrm <= 2.0 * plain.rmse()[0]
    injected = np.zeros(len(r), bool); injected[::37] = True
    assert np.all(nrm[injected] > 10.0)                                    # the shifted corners stay ~25 px off
    assert abs(rem.rmse()[0] - np.sqrt(0.5 * (r[kept] ** 2).sum() / kept.sum())) <= 0.02 * rem.rmse()[0]
    # and the calibration lands on the clean problem's answer
    np.testing.assert_allclose(rem.camera(0)[0][:4], ref.camera(0)[0][:4], rtol=2e-4)
    assert np.abs(plain.camera(0)[0][:4] / ref.camera(0)[0][:4] - 1).max() > np.abs(rem.camera(0)[0][:4] / ref.camera(0)[0][:4] - 1).max()


def test_no_convergence_solves_again_without_new_copies():
    """:954-1031: a solve that stops on max_num_iterations changes no flag and is simply run again on the same problem
    (SetupProblem is only re-entered after a `break`): many short solves end where one long solve ends."""
    p = _vision(models=("poly3",), n=12, seed=8)
    long = ol.Oracle().load(p); long.set_options(calibrate_imu=False, max_iters=200, function_tolerance=1e-13); long.solve()
    short = ol.Oracle().load(p); short.set_options(calibrate_imu=False, max_iters=4, function_tolerance=1e-13); short.solve()
    assert len(set(short.trace()[:, 9])) > 1 and len(set(long.trace()[:, 9])) == 1
    np.testing.assert_allclose(short.camera(0)[0], long.camera(0)[0], rtol=1e-5, atol=1e-7)
    # one copy of every block throughout: the final costs agree (a second copy would double it)
    assert abs(short.trace()[-1, 1] - long.trace()[-1, 1]) <= 1e-6 * long.trace()[-1, 1]
