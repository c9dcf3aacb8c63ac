"""The command-line tool (apps/vicalib.cpp): the reference tool's flags and outputs on top of the C ABI."""
import os
import re
import subprocess

import numpy as np
import pytest

from vicalib_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "vicalib_amd", "vicalib")


def _run(args, cwd=None):
    return subprocess.run([BIN] + args, cwd=cwd, capture_output=True, text=True, timeout=600)


def _read_xml(path):
    s = open(path).read()
    cams = []
    for block in s.split("<camera>")[1:]:
        typ = re.search(r'type="([^"]+)"', block).group(1)
        params = [float(x) for x in re.search(r"<params>\s*\[(.*?)\]", block, re.S).group(1).split(";")]
        T = [float(x) for x in re.split(r"[,;]", re.search(r"<T_wc>\s*\[(.*?)\]", block, re.S).group(1))]
        cams.append((typ, np.array(params), np.array(T).reshape(3, 4)))
    return cams


def test_help_and_flag_errors():
    r = _run(["--help"])
    assert r.returncode == 0
    for flag in ("-models", "-cam", "-imu", "-grid_preset", "-output", "-calibrate_imu", "-calibrate_intrinsics", "-find_time_offset",
                 "-has_initial_guess", "-model_files", "-max_iters", "-function_tolerance", "-gyro_sigma", "-accel_sigma",
                 "-remove_outliers", "-outlier_threshold", "-max_reprojection_error", "-num_vicalib_frames", "-frame_skip",
                 "-save_poses", "-print_poses", "-print_covariance", "-grid_height", "-grid_width", "-grid_spacing", "-grid_seed", "-gpus"):
        assert flag + " " in r.stdout, flag          # the CLI contract of SURVEY 8(b)
    assert _run([]).returncode == 1                   # "No camera URI given" (vicalib-engine.cc:445)
    r = _run(["-bogus", "1"]); assert r.returncode == 1 and "unknown command line flag" in r.stderr
    r = _run(["-max_iters", "abc"]); assert r.returncode == 1 and "illegal value" in r.stderr
    r = _run(["-cam", "detections:///does/not/exist.csv"]); assert r.returncode == 1 and "cannot open" in r.stderr


REFERENCE_FRONT_END_FLAGS = ["-device_serial", "abc123", "-scaled_ir_depth_cal", "-static_accel_threshold", "0.1", "-static_gyro_threshold=0.05",
                             "-static_threshold_preset", "1", "-use_static_threshold_preset", "-output_pattern_file", "pattern.svg",
                             "-grid_large_rad", "0.004", "-grid_small_rad", "0.003", "-noclip_good", "-max_imu_gyro_diff", "0.2",
                             "-max_imu_accel_diff=0.3"]


def test_every_reference_flag_is_accepted():
    """The flags of the reference's sensor / pattern front-end (vicalib-engine.cc:39, :65-77, :88-92, vicalib-task.cc:19, :45-48) parse:
    the command line gets as far as opening the detections (here: a file that does not exist), not 'unknown command line flag'."""
    r = _run(REFERENCE_FRONT_END_FLAGS + ["-cam", "detections:///does/not/exist.csv"])
    assert r.returncode == 1 and "cannot open" in r.stderr and "unknown command line flag" not in r.stderr
    h = _run(["--help"]).stdout
    for f in REFERENCE_FRONT_END_FLAGS:
        if f.startswith("-") and not f[1:2].isdigit():
            name = f.split("=")[0].replace("-no", "-", 1) if f.startswith("-noclip") else f.split("=")[0]
            assert name + " " in h, name


def test_no_gpu_is_a_loud_failure(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    p = synth.generate(synth.Config(models=("poly3",), n_frames=4, seed=2))
    files, _ = synth.write_dataset(p, str(tmp_path))
    r = _run(["-cam", "detections://" + files[0], "-models", "poly3", "-nocalibrate_imu", "-output", str(tmp_path / "cameras.xml")])
    assert r.returncode == 3 and "no HIP device" in r.stderr
    assert not (tmp_path / "cameras.xml").exists()


@pytest.mark.gpu
def test_cli_stereo_calibration_from_detection_files(tmp_path):
    p = synth.generate(synth.Config(models=("fov", "poly3"), n_frames=40, seed=12))
    files, _ = synth.write_dataset(p, str(tmp_path))
    out = tmp_path / "cameras.xml"
    r = _run(["-cam", "detections://" + ",".join(files), "-models", "fov,poly3", "-nocalibrate_imu", "-grid_preset", "small",
              "-output", str(out), "-save_poses", "-print_poses", "-print_covariance"], cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    # GetSolutionCovariance's log lines (vicalibrator.h:839-850): block names, then the matrix
    assert "Covariance calculated for blocks: c[0].q_ck:(4) c[0].p_ck:(3) c[0].params:(5) c[1].q_ck:(4) c[1].p_ck:(3) c[1].params:(7)" in r.stdout
    rows = r.stdout.split("Solution covariance:\n")[1].split("\n")[:26]
    cov = np.array([[float(x) for x in row.split()] for row in rows])
    assert cov.shape == (26, 26) and np.allclose(cov, cov.T, rtol=1e-5, atol=1e-300)
    assert np.all(cov[:7] == 0.0) and np.all(np.diag(cov)[7:] > 0)
    # residuals are in pixels (sigma 0.1 px): the focal length error is within a few predicted standard deviations
    sig_fu = p.cfg.pixel_sigma * np.sqrt(cov[7, 7])
    assert 0 < sig_fu < 1.0
    cams = _read_xml(str(out))
    assert [c[0] for c in cams] == ["calibu_fu_fv_u0_v0_w", "calibu_fu_fv_u0_v0_k1_k2_k3"]
    for c in range(2):
        np.testing.assert_allclose(cams[c][1][:4], p.cam_K_gt[c][:4], rtol=3e-3)
    # T_wc of camera 1 = T_ck^-1 (vision RDF): its translation is the stereo baseline
    from scipy.spatial.transform import Rotation as R
    T = p.cam_T_ck_gt[1]
    t_wc = -R.from_quat(T[:4]).inv().apply(T[4:])
    np.testing.assert_allclose(cams[1][2][:, 3], t_wc, atol=2e-3)
    assert (tmp_path / "poses.csv").exists() and (tmp_path / "poses.txt").exists()
    assert len(open(tmp_path / "poses.txt").read().strip().split("\n")) == 40
    assert "calibration succeeded" in r.stdout


@pytest.mark.gpu
def test_cli_visual_inertial_calibration(tmp_path):
    p = synth.generate(synth.Config(models=("kb4",), n_frames=80, imu=True, seed=5))
    files, imu_dir = synth.write_dataset(p, str(tmp_path))
    out = tmp_path / "cameras.xml"
    r = _run(["-cam", "detections://" + files[0], "-imu", "csv://" + imu_dir, "-models", "kb4", "-max_iters", "100", "-output", str(out)])
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"time offset: ([-0-9.e+]+) s", r.stdout)
    assert abs(float(m.group(1)) - p.imu_gt["time_offset"]) < 5e-4
    cams = _read_xml(str(out))
    assert cams[0][0] == "calibu_fu_fv_u0_v0_kb4"
    np.testing.assert_allclose(cams[0][1][:4], p.cam_K_gt[0][:4], rtol=3e-3)
    # WriteCameraModels with the IMU (vicalibrator.h:214-219): RDF = RdfRobotics and the pose written is
    # T_wc = T_ck^-1 * SE3(RdfRobotics^-1, 0)
    from scipy.spatial.transform import Rotation as R
    xml = open(out).read()
    assert "<right> [ 0; 1; 0 ] </right>" in xml and "<down> [ 0; 0; 1 ] </down>" in xml and "<forward> [ 1; 0; 0 ] </forward>" in xml
    T = p.cam_T_ck_gt[0]
    R_ck = R.from_quat(T[:4]).as_matrix()
    np.testing.assert_allclose(cams[0][2][:, :3], R_ck.T @ synth.RDF_ROBOTICS.T, atol=3e-3)
    np.testing.assert_allclose(cams[0][2][:, 3], -R_ck.T @ T[4:], atol=3e-3)
    # ... and exactly the solver's own T_ck, to the digits written: read the pose back and undo the RDF factor
    m = re.search(r"T_ck \[qx qy qz qw tx ty tz\]:((?: [-0-9.e+]+){7})", r.stdout)      # PrintResults, 10 digits
    Tck = np.array([float(x) for x in m.group(1).split()])
    Rs = R.from_quat(Tck[:4]).as_matrix()
    np.testing.assert_allclose(cams[0][2][:, :3] @ synth.RDF_ROBOTICS, Rs.T, atol=1e-8)
    np.testing.assert_allclose(cams[0][2][:, 3], -Rs.T @ Tck[4:], atol=1e-8)
    # round trip through -model_files (vicalib-engine.cc:186-199 reads the first camera of each rig file): with the intrinsics held
    # fixed the second run must write back the very same model, digit for digit
    out2 = tmp_path / "cameras2.xml"
    r2 = _run(["-cam", "detections://" + files[0], "-model_files", str(out), "-nocalibrate_intrinsics", "-nocalibrate_imu", "-output", str(out2)])
    assert r2.returncode == 0, r2.stdout + r2.stderr
    cams2 = _read_xml(str(out2))
    assert cams2[0][0] == cams[0][0]
    np.testing.assert_array_equal(cams2[0][1], cams[0][1])
    assert "<right> [ 1; 0; 0 ] </right>" in open(out2).read()          # no IMU: RdfVision (:221-225)


@pytest.mark.gpu
def test_cli_rational6_model(tmp_path):
    """-models rational6 (vicalib-engine.cc:233-240): start values (300, 300, w/2, h/2, 0 x 6), type string calibu_fu_fv_u0_v0_rational6."""
    p = synth.generate(synth.Config(models=("rational6",), n_frames=30, seed=7))
    files, _ = synth.write_dataset(p, str(tmp_path))
    out = tmp_path / "cameras.xml"
    r = _run(["-cam", "detections://" + files[0], "-models", "rational6", "-nocalibrate_imu", "-output", str(out)])
    assert r.returncode == 0, r.stdout + r.stderr
    cams = _read_xml(str(out))
    assert cams[0][0] == "calibu_fu_fv_u0_v0_rational6" and len(cams[0][1]) == 10
    np.testing.assert_allclose(cams[0][1][:4], p.cam_K_gt[0][:4], rtol=5e-3)
    assert "calibration succeeded" in r.stdout


@pytest.mark.gpu
def test_cli_initial_guess_reproduces_the_reference_exit_status(tmp_path):
    """-has_initial_guess: IsSuccessful also runs IMUCalibrationDiffer (vicalib-task.cc:807-829, :852-853) against the biases the
    calibrator was constructed with (zero, :129), and its comparisons are inverted in the reference ('<': a bias that moved by
    LESS than the limit counts as differing).  The default reproduces that exit status -- a good calibration is reported as FAILED
    whenever a bias stays below 0.1, i.e. always without an IMU -- and says so; -imu_diff_sense corrected judges it the way the
    message reads."""
    p = synth.generate(synth.Config(models=("kb4",), n_frames=80, imu=True, seed=5))
    files, imu_dir = synth.write_dataset(p, str(tmp_path))
    out = tmp_path / "cameras.xml"
    r = _run(["-cam", "detections://" + files[0], "-imu", "csv://" + imu_dir, "-models", "kb4", "-max_iters", "100", "-output", str(out)])
    assert r.returncode == 0, r.stdout + r.stderr
    # vision only from the calibrated model: RMSE and the camera comparisons pass, the bias comparison decides
    guess = ["-cam", "detections://" + files[0], "-model_files", str(out), "-has_initial_guess", "-nocalibrate_imu", "-output", str(tmp_path / "c2.xml")]
    r_ref = _run(guess + REFERENCE_FRONT_END_FLAGS[:2])
    assert r_ref.returncode == 2 and "calibration FAILED" in r_ref.stdout, r_ref.stdout + r_ref.stderr
    assert "IMU bias(es) for gyroscope differ" in r_ref.stderr and "reference comparison sense" in r_ref.stderr
    r_fix = _run(guess + ["-imu_diff_sense", "corrected"])
    assert r_fix.returncode == 0 and "calibration succeeded" in r_fix.stdout, r_fix.stdout + r_fix.stderr
    # with the IMU, Start(has_initial_guess) (vicalib-task.cc:226-234) goes straight to the stage with everything free, from the
    # model file's intrinsics and an identity T_ck (vicalib-engine.cc:190-193): far from a good start, as in the reference.  With
    # the earlier gates opened wide the run reaches the bias comparison in either sense.
    wide = ["-max_reprojection_error", "1e3", "-max_fx_diff", "1e3", "-max_fy_diff", "1e3", "-max_cx_diff", "1e3", "-max_cy_diff", "1e3",
            "-max_camera_trans_diff", "1e3", "-max_camera_angle_diff", "10"]
    vi = ["-cam", "detections://" + files[0], "-imu", "csv://" + imu_dir, "-model_files", str(out), "-has_initial_guess", "-max_iters", "30",
          "-output", str(tmp_path / "c3.xml")] + wide
    r_vi = _run(vi + ["-max_imu_gyro_diff", "1e9", "-max_imu_accel_diff", "1e9"])
    assert r_vi.returncode == 2 and "IMU bias(es) for gyroscope differ" in r_vi.stderr, r_vi.stdout + r_vi.stderr     # '<' 1e9: always
    r_vi2 = _run(vi + ["-imu_diff_sense", "corrected", "-max_imu_gyro_diff", "1e9", "-max_imu_accel_diff", "1e9"])
    assert r_vi2.returncode == 0, r_vi2.stdout + r_vi2.stderr


@pytest.mark.gpu
def test_cli_calibrates_from_images(tmp_path):
    """images -> dots -> grid -> PnP seed -> solve -> cameras.xml (SURVEY 8 f4 + f1; vicalib-task.cc:263-348 with PGM files for HAL): twelve
    rendered views of a 13 x 9 two-size dot target (pattern from -grid_seed), pinhole camera with fu = fv = 420."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import dot_images
    from vicalib_amd import lib
    pat = lib.target_make_pattern(9, 13, seed=71)
    rng = np.random.default_rng(4)
    for k in range(12):
        tilt = tuple(rng.uniform(-0.45, 0.45, size=2)) + (rng.uniform(-3.0, 3.0),)
        img, _ = dot_images.render(seed=k, tilt=tilt, dist=rng.uniform(0.36, 0.5), nx=13, ny=9, spacing=0.022, r_large=0.0069, r_small=0.0046,
                                   pattern=pat, missing=0.05, ss=4)
        with open(tmp_path / ("view_%03d.pgm" % k), "wb") as f:
            f.write(b"P5\n# rendered by tests/dot_images.py\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
            f.write(img.tobytes())
    out = tmp_path / "cameras.xml"
    r = _run(["-cam", "file://" + str(tmp_path / "view_*.pgm"), "-grid_height", "9", "-grid_width", "13", "-grid_spacing", "0.022", "-grid_seed", "71",
              "-models", "linear", "-nocalibrate_imu", "-output", str(out)])
    assert r.returncode == 0, r.stderr[-3000:]
    assert "12 images" in r.stderr
    typ, K, T = _read_xml(str(out))[0]
    np.testing.assert_allclose(K[:2], [420.0, 420.0], rtol=5e-3)
    np.testing.assert_allclose(K[2:4], [319.5, 239.5], atol=2.0)
    # a preset's pattern cannot be known without Calibu: refused with an explanation, not guessed
    r = _run(["-cam", "file://" + str(tmp_path / "view_*.pgm"), "-grid_preset", "small", "-models", "linear", "-nocalibrate_imu"])
    assert r.returncode == 1 and "grid_pattern_file" in r.stderr
    # ... and with the printed target's pattern supplied (-grid_pattern_file: rows of 0 / 1, 1 = large dot) a preset's target is usable with
    # images: the file's rows / columns replace the preset's, the preset keeps its spacing default unless -grid_spacing says otherwise
    # (vicalib-engine.cc:453-465 resolve the pattern inside Calibu; this is the escape hatch for a tree without it)
    pf = tmp_path / "pattern.txt"
    with open(pf, "w") as f:
        f.write("# 13 x 9 two-size dot target, the pattern of -grid_seed 71\n")
        for row in np.asarray(pat).reshape(9, 13):
            f.write(" ".join(str(int(x)) for x in row) + "\n")
    out2 = tmp_path / "cameras_pattern.xml"
    r = _run(["-cam", "file://" + str(tmp_path / "view_*.pgm"), "-grid_preset", "small", "-grid_pattern_file", str(pf), "-grid_spacing", "0.022",
              "-models", "linear", "-nocalibrate_imu", "-output", str(out2)])
    assert r.returncode == 0, r.stderr[-3000:]
    typ2, K2, T2 = _read_xml(str(out2))[0]
    np.testing.assert_allclose(K2, K, rtol=1e-9)          # the same target, the same calibration
