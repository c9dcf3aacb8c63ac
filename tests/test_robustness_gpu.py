"""Error behaviour of the C ABI on the device: the reference aborts (CHECK / LOG(FATAL), vicalibrator.h:254, :377, :396,
:456) or misbehaves on degenerate input; the replacement returns status codes and always terminates."""
import ctypes as C
import time

import numpy as np
import pytest

import vicalib_amd.lib as lib
from vicalib_amd import synth
from vicalib_amd.lib import ViCalibrator, VicalibError

pytestmark = pytest.mark.gpu
I7 = np.array([0, 0, 0, 1, 0, 0, 0.0])


def test_empty_problem_terminates():
    cal = ViCalibrator(0)
    cal.Solve()                                   # nothing to do: SolveThread leaves at once (vicalibrator.h:921-923)
    assert cal.NumFrames() == 0 and cal.GetNumIterations() == 0
    cal.AddCamera("poly3", [300, 300, 320, 240, 0, 0, 0.0], I7, 640, 480)
    cal.AddFrame(I7, 0.0)
    cal.SetCalibrateImu(False)
    cal.Solve()                                   # frames but no observations
    assert not cal.IsRunning()


def test_single_frame_single_camera():
    p = synth.generate(synth.Config(models=("fov",), n_frames=1, seed=4))
    cal = ViCalibrator(0).load_problem(p); cal.SetCalibrateImu(False)
    cal.Solve()
    K, T = cal.GetCamera(0)
    assert np.all(np.isfinite(K)) and np.all(np.isfinite(cal.GetFrames()))


def test_camera_without_observations_and_bad_arguments():
    p = synth.generate(synth.Config(models=("poly2",), n_frames=10, seed=4))
    cal = ViCalibrator(0).load_problem(p)
    cal.AddCamera("kb4", [300, 300, 320, 240, 0, 0, 0, 0.0], I7, 640, 480)     # a second camera nobody observes
    cal.SetCalibrateImu(False)
    cal.Solve()
    assert np.all(np.isfinite(cal.GetCamera(0)[0]))
    np.testing.assert_array_equal(cal.GetCamera(1)[0], [300, 300, 320, 240, 0, 0, 0, 0.0])      # untouched
    L = lib.load()
    assert L.vc_add_observations(cal.h, 99, 0, 1, lib._d(np.zeros(3)), lib._d(np.zeros(2))) == -2      # VC_ERR_BAD_ARG
    assert L.vc_add_observations(cal.h, 0, 7, 1, lib._d(np.zeros(3)), lib._d(np.zeros(2))) == -2
    assert L.vc_set_frame_pose(cal.h, -1, lib._d(I7)) == -2
    assert L.vc_add_camera(cal.h, 11, lib._d(np.zeros(4)), 4, 640, 480, lib._d(I7)) == -2              # unknown model
    assert L.vc_add_camera(cal.h, 0, lib._d(np.zeros(4)), 4, 640, 480, lib._d(I7)) == -2               # fov needs 5 parameters
    n = C.c_int(0)
    assert L.vc_get_camera(cal.h, 5, lib._d(np.zeros(16)), C.byref(n), lib._d(np.zeros(7))) == -2


def test_more_cameras_than_supported_is_refused():
    cal = ViCalibrator(0)
    for _ in range(8):
        cal.AddCamera("linear", [300, 300, 320, 240.0], I7, 640, 480)
    with pytest.raises(VicalibError):
        cal.AddCamera("linear", [300, 300, 320, 240.0], I7, 640, 480)


def test_non_finite_detections_end_in_a_failure_not_a_hang():
    p = synth.generate(synth.Config(models=("poly3",), n_frames=8, seed=4))
    p.tiles[3][3][0, 0] = np.nan
    cal = ViCalibrator(0).load_problem(p); cal.SetCalibrateImu(False); cal.SetMaxIters(20)
    t0 = time.time()
    cal.Solve()
    assert time.time() - t0 < 30
    assert not cal.IsRunning()


def test_setters_are_refused_while_the_solver_runs_and_destroy_joins():
    p = synth.generate(synth.Config(models=("fov", "fov"), n_frames=300, seed=4))
    cal = ViCalibrator(0).load_problem(p); cal.SetCalibrateImu(False)
    L = lib.load()
    cal.Start()
    rc = L.vc_set_max_iters(cal.h, 5)
    running = cal.IsRunning()
    assert rc == -3 or not running                 # VC_ERR_RUNNING (reference: CHECK(!is_running_), vicalibrator.h:254)
    del cal                                        # vc_destroy while (possibly) running: stops and joins


def test_imu_time_order_and_missing_imu():
    cal = ViCalibrator(0)
    cal.AddImuMeasurements(np.zeros((2, 3)), np.zeros((2, 3)), [0.0, 0.1])
    with pytest.raises(VicalibError):
        cal.AddImuMeasurements(np.zeros((1, 3)), np.zeros((1, 3)), [0.05])         # not increasing (vicalibrator.h:373-378)
    # calibrate_imu requested but no IMU samples at all: every block is empty, the schedule still runs through
    p = synth.generate(synth.Config(models=("poly3",), n_frames=12, seed=4))
    c2 = ViCalibrator(0).load_problem(p); c2.SetMaxIters(30)
    c2.Solve()
    assert np.all(np.isfinite(c2.GetCamera(0)[0])) and c2.GetCameraProjRMSE()[0] < 0.2


def test_handle_reuse_clear_and_reconfigure():
    """One handle over several problems and settings: Clear() (vicalibrator.h:232-249), a different rig, fixed intrinsics,
    a second solve after changing the tolerance -- state from the previous use must not leak."""
    import oracle_lib as ol
    p1 = synth.generate(synth.Config(models=("fov", "fov"), n_frames=20, seed=31))
    p2 = synth.generate(synth.Config(models=("kb4",), n_frames=15, seed=32))
    cal = ViCalibrator(0).load_problem(p1); cal.SetCalibrateImu(False)
    cal.Solve()
    k1 = cal.GetCamera(0)[0].copy()
    cal.Clear()
    assert cal.NumFrames() == 0 and cal.NumCameras() == 0
    cal.load_problem(p2); cal.SetCalibrateImu(False)
    cal.Solve()
    orc = ol.Oracle().load(p2); orc.set_options(calibrate_imu=False); orc.solve()
    np.testing.assert_allclose(cal.GetCamera(0)[0], orc.camera(0)[0], rtol=1e-6)
    np.testing.assert_allclose(cal.trace()[:, 1], orc.trace()[:, 1], rtol=1e-6)
    # same handle, back to the first rig with the intrinsics held fixed at the earlier result
    cal.Clear()
    for c, m in enumerate(p1.cam_model):
        cal.AddCamera(m, k1 if c == 0 else p1.cam_K_gt[c], p1.cam_T_ck_init[c], 640, 480)
    for n in range(len(p1.frame_time)):
        cal.AddFrame(p1.frame_T_wk_init[n], p1.frame_time[n])
    for (f, c, ids, pix) in p1.tiles:
        cal.AddObservations(f, c, p1.grid_points[ids], pix)
    cal.SetCalibrateImu(False); cal.FixCameraIntrinsics(True)
    cal.Solve()
    np.testing.assert_array_equal(cal.GetCamera(0)[0], k1)                 # untouched
    assert cal.GetCameraProjRMSE()[0] < 0.15
    cal.SetFunctionTolerance(1e-12); cal.Resume(); cal.Solve()                           # tighter tolerance: a few more iterations, same optimum
    assert cal.GetCameraProjRMSE()[0] < 0.15
