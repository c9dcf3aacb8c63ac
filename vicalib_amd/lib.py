"""ctypes binding of the C ABI (include/vicalib_amd.h) -- used by tests/ and bench.py.

`ViCalibrator` mirrors the public API of the reference class of the same name
(include/vicalib/vicalibrator.h:119-544): same method names and argument meaning, so the parity
tests read like code written against the reference.  There is no fallback: if the HIP library is
missing or no GPU is present, construction raises.
"""
from __future__ import annotations

import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VICALIB_AMD_LIB") or os.path.join(_HERE, "libvicalib_amd.so")
_lib = None

VC_OK = 0
ERRORS = {-1: "VC_ERR_NO_DEVICE", -2: "VC_ERR_BAD_ARG", -3: "VC_ERR_RUNNING", -4: "VC_ERR_TIME_ORDER",
          -5: "VC_ERR_TOO_MANY_POINTS", -6: "VC_ERR_NUMERIC", -7: "VC_ERR_UNSUPPORTED", -8: "VC_ERR_NO_CONVERGENCE"}
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int)

# every symbol include/vicalib_amd.h declares
SYMBOLS = [
    "vc_create", "vc_destroy", "vc_clear", "vc_add_camera", "vc_fix_camera_intrinsics", "vc_add_frame", "vc_set_frame_pose",
    "vc_add_observations", "vc_add_observation_tiles", "vc_add_imu", "vc_set_sigmas", "vc_set_biases", "vc_set_scale_factor", "vc_set_time_offset",
    "vc_set_function_tolerance", "vc_set_optimization_flags", "vc_set_max_iters", "vc_set_tolerances", "vc_set_gravity", "vc_set_frame_velocities", "vc_set_calibrate_imu", "vc_set_remove_outliers",
    "vc_solve", "vc_start", "vc_resume", "vc_set_stage_limit", "vc_sync_timeouts", "vc_set_kernel_timing", "vc_get_kernel_timing", "vc_is_running", "vc_stop", "vc_num_frames", "vc_num_cameras", "vc_get_camera", "vc_get_frame",
    "vc_get_biases", "vc_get_scale_factor", "vc_get_gravity", "vc_time_offset", "vc_mean_squared_error", "vc_get_camera_proj_rmse",
    "vc_get_num_iterations", "vc_num_imu_measurements", "vc_get_imu_measurements", "vc_get_integration_poses", "vc_print_results", "vc_write_camera_models", "vc_trace_len", "vc_get_trace", "vc_set_shard", "vc_get_stream", "vc_prepare",
    "vc_linearize", "vc_shared_dim", "vc_run_iterations", "vc_evaluate", "vc_time_kernels", "vc_time_stages", "vc_get_imu_blocks", "vc_get_debug_stamps", "vc_num_observations", "vc_num_tiles",
    "vc_init_frame_poses_pnp", "vc_pnp_planar", "vc_pnp_planar_ransac", "vc_set_pnp_ransac", "vc_rccl_unique_id", "vc_set_shard_rccl", "vc_shard_comm_create", "vc_set_shard_comm", "vc_shard_comm_destroy", "vc_allreduce_calls", "vc_shard_info", "vc_pass_paths", "vc_last_error", "vc_get_imu_weights",
    "vc_solution_covariance_dim", "vc_get_solution_covariance", "vc_get_solution_covariance_names",
    "vc_target_make_pattern", "vc_target_find",
    "vc_detector_create", "vc_detector_destroy", "vc_detector_set_params", "vc_detector_find", "vc_detector_find_conics",
]


class VicalibError(RuntimeError):
    pass


def target_make_pattern(rows, cols, seed=71):
    """vc_target_make_pattern: rows x cols array, 1 = large dot."""
    out = np.zeros((rows, cols), dtype=np.int32)
    _check(load().vc_target_make_pattern(int(rows), int(cols), C.c_uint(seed), out.ctypes.data_as(C.c_void_p)), "target_make_pattern")
    return out


def target_find(centres, conics, pattern):
    """vc_target_find: dot index (row * cols + col, or -1) of every conic; an all -1 result means no unambiguous placement."""
    centres = np.ascontiguousarray(centres, dtype=np.float64); conics = np.ascontiguousarray(conics, dtype=np.float64)
    pattern = np.ascontiguousarray(pattern, dtype=np.int32)
    n = len(centres)
    idx = np.full(n, -1, dtype=np.int32); m = C.c_int(0)
    _check(load().vc_target_find(_d(centres), _d(conics), n, pattern.ctypes.data_as(C.c_void_p), pattern.shape[0], pattern.shape[1],
                                 idx.ctypes.data_as(C.c_void_p), C.byref(m)), "target_find")
    return idx, m.value


def pnp_planar(model, params, p_w, p_c):
    """T_cw and RMS reprojection error (pixels) of one view of the planar grid (vc_pnp_planar; host code, needs no GPU)."""
    L = load()
    params = np.ascontiguousarray(params, dtype=np.float64)
    p_w = np.ascontiguousarray(p_w, dtype=np.float64); p_c = np.ascontiguousarray(p_c, dtype=np.float64)
    T = np.zeros(7); rms = C.c_double(0)
    from .synth import MODEL_IDS
    m = MODEL_IDS[model] if isinstance(model, str) else int(model)
    _check(L.vc_pnp_planar(m, _d(params), len(params), len(p_w), _d(p_w), _d(p_c), _d(T), C.byref(rms)), "pnp_planar")
    return T, rms.value


def pnp_planar_ransac(model, params, p_w, p_c, iterations=64, tol_px=2.0):
    """Robust pose of one view (vc_pnp_planar_ransac): T_cw, RMS over the inliers, inlier flags."""
    L = load()
    params = np.ascontiguousarray(params, dtype=np.float64)
    p_w = np.ascontiguousarray(p_w, dtype=np.float64); p_c = np.ascontiguousarray(p_c, dtype=np.float64)
    T = np.zeros(7); rms = C.c_double(0); n_in = C.c_int(0); flags = np.zeros(len(p_w), dtype=np.int8)
    from .synth import MODEL_IDS
    m = MODEL_IDS[model] if isinstance(model, str) else int(model)
    _check(L.vc_pnp_planar_ransac(m, _d(params), len(params), len(p_w), _d(p_w), _d(p_c), int(iterations), C.c_double(tol_px), _d(T),
                                  C.byref(rms), C.byref(n_in), flags.ctypes.data_as(C.c_void_p)), "pnp_planar_ransac")
    return T, rms.value, flags.astype(bool)


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VicalibError(f"{LIB_PATH} is missing: build it with vicalib_amd/csrc/build.sh (no CPU fallback exists)")
        L = C.CDLL(LIB_PATH)
        L.vc_time_offset.restype = C.c_double
        L.vc_mean_squared_error.restype = C.c_double
        L.vc_get_num_iterations.restype = C.c_uint
        L.vc_get_stream.restype = C.c_void_p
        L.vc_num_observations.restype = C.c_longlong
        L.vc_allreduce_calls.restype = C.c_longlong
        L.vc_last_error.restype = C.c_char_p
        for name in ("vc_destroy", "vc_detector_destroy", "vc_shard_comm_destroy"):
            getattr(L, name).restype = None
        _lib = L
    return _lib


def _check(rc, what):
    if rc < 0:
        raise VicalibError(f"{what} failed: {ERRORS.get(rc, rc)}")
    return rc


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(C.c_void_p)


class ShardComm:
    """One RCCL communicator for all calibrators of this process (vc_shard_comm_create): created once -- collective, the 128-byte id
    travels through torch.distributed --, lent to calibrators with ViCalibrator.set_shard_comm, destroyed with close() after them."""

    def __init__(self, device, rank, world, group=None):
        import torch.distributed as dist
        self.L = load()
        self.rank, self.world = int(rank), int(world)
        buf = C.create_string_buffer(128)
        if rank == 0:
            _check(self.L.vc_rccl_unique_id(buf), "rccl_unique_id")
        box = [bytes(buf.raw)]
        if world > 1:
            dist.broadcast_object_list(box, src=0, group=group)
        self.h = C.c_void_p()
        rc = self.L.vc_shard_comm_create(int(device), self.rank, self.world, C.create_string_buffer(box[0], 128), C.byref(self.h))
        if rc != 0:
            self.h = None
            raise VicalibError("shard_comm_create: status %d: %s" % (rc, (self.L.vc_last_error() or b"").decode(errors="replace")))

    def close(self):
        if getattr(self, "h", None):
            self.L.vc_shard_comm_destroy(self.h)
            self.h = None


class ViCalibrator:
    def __init__(self, device: int = 0):
        self.L = load()
        self.h = C.c_void_p()
        _check(self.L.vc_create(C.byref(self.h), int(device)), "vc_create")
        self._cb = None
        self.nk = []

    def close(self):
        if getattr(self, "h", None):
            self.L.vc_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    # ---- reference API ---------------------------------------------------------------------
    def AddCamera(self, model, params, T_ck, width=640, height=480):
        params = np.ascontiguousarray(params, dtype=np.float64)
        self.nk.append(len(params))
        if isinstance(model, str):               # the -models strings of vicalib-engine.cc:203-253
            from .synth import MODEL_IDS
            model = MODEL_IDS[model]
        return _check(self.L.vc_add_camera(self.h, int(model), _d(params), len(params), int(width), int(height), _d(T_ck)), "AddCamera")

    def Clear(self):
        _check(self.L.vc_clear(self.h), "Clear")
        self.nk = []

    def FixCameraIntrinsics(self, should_fix=True):
        _check(self.L.vc_fix_camera_intrinsics(self.h, int(should_fix)), "FixCameraIntrinsics")

    def AddFrame(self, T_wk, time):
        return _check(self.L.vc_add_frame(self.h, _d(T_wk), C.c_double(time)), "AddFrame")

    def SetFramePose(self, frame, T_wk):
        _check(self.L.vc_set_frame_pose(self.h, int(frame), _d(T_wk)), "SetFramePose")

    def InitFramePosesPnP(self):
        """calibu::PosePnPRansac + the pose write of vicalib-task.cc:335-348; returns the number of frames initialised."""
        n = C.c_int(0)
        _check(self.L.vc_init_frame_poses_pnp(self.h, C.byref(n)), "InitFramePosesPnP")
        return n.value

    def SetPnPRansac(self, iterations, tol_px): _check(self.L.vc_set_pnp_ransac(self.h, int(iterations), C.c_double(tol_px)), "SetPnPRansac")

    def AddObservations(self, frame, camera, p_w, p_c):
        p_w = np.ascontiguousarray(p_w, dtype=np.float64); p_c = np.ascontiguousarray(p_c, dtype=np.float64)
        _check(self.L.vc_add_observations(self.h, int(frame), int(camera), int(len(p_w)), _d(p_w), _d(p_c)), "AddObservation")

    def AddObservationTiles(self, tile_frame, tile_cam, tile_off, points, point_id, p_c):
        """AddObservation over many (frame, camera) groups in one call (vc_add_observation_tiles)."""
        tf = np.ascontiguousarray(tile_frame, dtype=np.int32); tc = np.ascontiguousarray(tile_cam, dtype=np.int32)
        off = np.ascontiguousarray(tile_off, dtype=np.int64); pid = np.ascontiguousarray(point_id, dtype=np.int32)
        pts = np.ascontiguousarray(points, dtype=np.float64); pc = np.ascontiguousarray(p_c, dtype=np.float64)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)      # noqa: E731
        _check(self.L.vc_add_observation_tiles(self.h, len(tf), vp(tf), vp(tc), vp(off), vp(pts), len(pts), vp(pid), vp(pc)), "AddObservationTiles")

    def AddImuMeasurements(self, gyro, accel, time):
        time = np.ascontiguousarray(time, dtype=np.float64)
        _check(self.L.vc_add_imu(self.h, int(len(time)), _d(gyro), _d(accel), _d(time)), "AddImuMeasurements")

    def SetSigmas(self, g, a): _check(self.L.vc_set_sigmas(self.h, C.c_double(g), C.c_double(a)), "SetSigmas")
    def SetBiases(self, b): _check(self.L.vc_set_biases(self.h, _d(b)), "SetBiases")
    def SetGravity(self, g): _check(self.L.vc_set_gravity(self.h, _d(np.asarray(g, dtype=np.float64))), "SetGravity")   # engine-level (the reference has no setter)
    def SetScaleFactor(self, s): _check(self.L.vc_set_scale_factor(self.h, _d(s)), "SetScaleFactor")
    def SetTimeOffset(self, o): _check(self.L.vc_set_time_offset(self.h, C.c_double(o)), "SetTimeOffset")
    def SetFunctionTolerance(self, t): _check(self.L.vc_set_function_tolerance(self.h, C.c_double(t)), "SetFunctionTolerance")

    def SetOptimizationFlags(self, bias_active, inertial_active, rotation_only, optimize_time_offset):
        _check(self.L.vc_set_optimization_flags(self.h, int(bias_active), int(inertial_active), int(rotation_only), int(optimize_time_offset)), "SetOptimizationFlags")

    def SetTolerances(self, gradient_tolerance=1e-10, parameter_tolerance=1e-8):
        _check(self.L.vc_set_tolerances(self.h, C.c_double(gradient_tolerance), C.c_double(parameter_tolerance)), "SetTolerances")

    def SetMaxIters(self, m): _check(self.L.vc_set_max_iters(self.h, int(m)), "max_iters")
    def SetCalibrateImu(self, c): _check(self.L.vc_set_calibrate_imu(self.h, int(c)), "calibrate_imu")
    def SetRemoveOutliers(self, r, th=2.0): _check(self.L.vc_set_remove_outliers(self.h, int(r), C.c_double(th)), "remove_outliers")

    def Solve(self): return _check(self.L.vc_solve(self.h), "Solve")
    def Start(self): _check(self.L.vc_start(self.h), "Start")
    def SetStageLimit(self, n): _check(self.L.vc_set_stage_limit(self.h, int(n)), "SetStageLimit")
    def Resume(self): _check(self.L.vc_resume(self.h), "Resume")
    def IsRunning(self): return bool(self.L.vc_is_running(self.h))
    def Stop(self): _check(self.L.vc_stop(self.h), "Stop")
    def NumFrames(self): return self.L.vc_num_frames(self.h)
    def NumCameras(self): return self.L.vc_num_cameras(self.h)

    def GetCamera(self, c):
        K = np.zeros(10); n = C.c_int(0); T = np.zeros(7)
        _check(self.L.vc_get_camera(self.h, int(c), _d(K), C.byref(n), _d(T)), "GetCamera")
        return K[:n.value].copy(), T

    def GetFrame(self, f):
        T = np.zeros(7); v = np.zeros(3); t = C.c_double(0)
        _check(self.L.vc_get_frame(self.h, int(f), _d(T), _d(v), C.byref(t)), "GetFrame")
        return T, v, t.value

    def GetVelocities(self):
        n = self.NumFrames()
        v = np.zeros((n, 3))
        for f in range(n):
            v[f] = self.GetFrame(f)[1]
        return v

    def GetFrames(self):
        n = self.NumFrames()
        T = np.zeros((n, 7))
        for f in range(n):
            T[f] = self.GetFrame(f)[0]
        return T

    def GetBiases(self):
        b = np.zeros(6); _check(self.L.vc_get_biases(self.h, _d(b)), "GetBiases"); return b

    def GetScaleFactor(self):
        s = np.zeros(6); _check(self.L.vc_get_scale_factor(self.h, _d(s)), "GetScaleFactor"); return s

    def GetGravity(self):
        g = np.zeros(2); _check(self.L.vc_get_gravity(self.h, _d(g)), "GetGravity"); return g

    def time_offset(self): return self.L.vc_time_offset(self.h)
    def MeanSquaredError(self): return self.L.vc_mean_squared_error(self.h)

    def GetCameraProjRMSE(self):
        r = np.zeros(max(self.NumCameras(), 1)); _check(self.L.vc_get_camera_proj_rmse(self.h, _d(r)), "GetCameraProjRMSE")
        return r[:self.NumCameras()]

    def GetNumIterations(self): return self.L.vc_get_num_iterations(self.h)

    def imu_buffer(self):
        n = _check(self.L.vc_num_imu_measurements(self.h), "imu_buffer")
        g = np.zeros((n, 3)); a = np.zeros((n, 3)); t = np.zeros(n)
        _check(self.L.vc_get_imu_measurements(self.h, _d(g), _d(a), _d(t), n), "imu_buffer")
        return g, a, t

    def GetIntegrationPoses(self, frame_id):
        n = _check(self.L.vc_get_integration_poses(self.h, int(frame_id), None, 0), "GetIntegrationPoses")     # the count first
        out = np.zeros((max(n, 1), 11))
        if n > 0:
            _check(self.L.vc_get_integration_poses(self.h, int(frame_id), _d(out), n), "GetIntegrationPoses")
        return out[:n]

    def PrintResults(self):
        # the length first (any number of cameras); a worker started with Start() may change the values -- and with them the
        # length of the %.10g text -- between the two calls: some slack, and another round if that was not enough
        for _ in range(8):
            n = _check(self.L.vc_print_results(self.h, None, 0), "PrintResults")
            buf = C.create_string_buffer(n + 65)
            if self.L.vc_print_results(self.h, buf, n + 65) >= 0:
                return buf.value.decode()
        raise VicalibError("PrintResults: the text kept growing")
    def WriteCameraModels(self, path): _check(self.L.vc_write_camera_models(self.h, path.encode()), "WriteCameraModels")

    # ---- engine-level ------------------------------------------------------------------------
    def load_problem(self, prob, init=True):
        for c, m in enumerate(prob.cam_model):
            self.AddCamera(m, prob.cam_K_init[c] if init else prob.cam_K_gt[c], prob.cam_T_ck_init[c] if init else prob.cam_T_ck_gt[c],
                           prob.cfg.width, prob.cfg.height)
        T = prob.frame_T_wk_init if init else prob.frame_T_wk_gt
        for n in range(len(prob.frame_time)):
            self.AddFrame(T[n], prob.frame_time[n])
        if getattr(prob, "flat", None) is not None:
            tf, tc, off, ids, pix = prob.flat
            self.AddObservationTiles(tf, tc, off, prob.grid_points, ids, pix)
        else:
            for (f, c, ids, pix) in prob.tiles:
                self.AddObservations(f, c, prob.grid_points[ids], pix)
        if prob.imu_t is not None:
            self.AddImuMeasurements(prob.imu_gyro, prob.imu_accel, prob.imu_t)
        return self

    def trace(self):
        n = _check(self.L.vc_trace_len(self.h), "trace_len")
        out = np.zeros((max(n, 1), 10))
        _check(self.L.vc_get_trace(self.h, _d(out), n), "get_trace")
        return out[:n]

    def set_shard(self, rank, world, fn=None):
        self._cb = ALLREDUCE_FN(fn) if fn is not None else None
        _check(self.L.vc_set_shard(self.h, int(rank), int(world), self._cb, None), "set_shard")

    def set_shard_rccl(self, rank, world, group=None):
        """Frame sharding over the library's own RCCL communicator; the 128-byte id travels through torch.distributed."""
        import torch.distributed as dist
        buf = C.create_string_buffer(128)
        if rank == 0:
            _check(self.L.vc_rccl_unique_id(buf), "rccl_unique_id")
        box = [bytes(buf.raw)]
        if world > 1:
            dist.broadcast_object_list(box, src=0, group=group)
        rc = self.L.vc_set_shard_rccl(self.h, int(rank), int(world), C.create_string_buffer(box[0], 128))
        if rc != 0:          # say which RCCL call failed and why (vc_last_error) before the caller falls back to another transport
            raise VicalibError("set_shard_rccl: status %d: %s" % (rc, (self.L.vc_last_error() or b"").decode(errors="replace")))

    def set_shard_comm(self, comm):
        """Frame sharding over a communicator shared with the process's other calibrators (ShardComm); not owned by this calibrator."""
        rc = self.L.vc_set_shard_comm(self.h, comm.h)
        if rc != 0:
            raise VicalibError("set_shard_comm: status %d: %s" % (rc, (self.L.vc_last_error() or b"").decode(errors="replace")))

    def allreduce_calls(self): return int(self.L.vc_allreduce_calls(self.h))

    def pass_paths(self):
        """Which forms of the visual-inertial pass the uploaded problem runs (vc_pass_paths)."""
        out = (C.c_int * 6)()
        _check(self.L.vc_pass_paths(self.h, out), "pass_paths")
        return dict(fold_l0=out[0], back_path=out[1], early_gram=out[2], top_gram_launch=out[3], tail_deferred=out[4], shared_blocks_ahead=out[5])

    def shard_info(self):
        """rank / world size the calibrator shards with and, for the library's own communicator, what RCCL reports (-1: none attached)."""
        v = [C.c_int(-1) for _ in range(4)]
        _check(self.L.vc_shard_info(self.h, *[C.byref(x) for x in v]), "shard_info")
        return dict(rank=v[0].value, world=v[1].value, rccl_ranks=v[2].value, rccl_rank=v[3].value)

    def stream(self): return self.L.vc_get_stream(self.h)
    def prepare(self): _check(self.L.vc_prepare(self.h), "prepare")
    def shared_dim(self): return _check(self.L.vc_shared_dim(self.h), "shared_dim")

    def linearize(self):
        self.prepare()
        n, D = self.NumFrames(), self.shared_dim()
        cost = C.c_double(0); H = np.zeros((n, 6, 6)); g = np.zeros((n, 6)); S = np.zeros((D, D)); gr = np.zeros(D); hd = np.zeros(D); gs = np.zeros(D)
        _check(self.L.vc_linearize(self.h, C.byref(cost), _d(H), _d(g), _d(S), _d(gr), _d(hd), _d(gs)), "linearize")
        return dict(cost=cost.value, Hpp=H, gp=g, S=S, g_red=gr, hss_diag=hd, g_s=gs)

    def evaluate(self):
        cost = C.c_double(0); sq = C.c_double(0)
        _check(self.L.vc_evaluate(self.h, C.byref(cost), C.byref(sq)), "evaluate")
        return cost.value, sq.value

    def run_iterations(self, iters):
        j = C.c_int(0); r = C.c_int(0)
        n = _check(self.L.vc_run_iterations(self.h, int(iters), C.byref(j), C.byref(r)), "run_iterations")
        return n, j.value, r.value

    def time_kernels(self, reps=20):
        a = C.c_double(0); b = C.c_double(0)
        _check(self.L.vc_time_kernels(self.h, int(reps), C.byref(a), C.byref(b)), "time_kernels")
        return a.value, b.value

    def time_stages(self, reps=50):
        out = np.zeros(6)
        _check(self.L.vc_time_stages(self.h, int(reps), _d(out)), "time_stages")
        return dict(zip(["jac", "frame_schur", "-", "reduced", "trial", "final"], (out * 1e3).tolist()))

    def set_kernel_timing(self, on=True): _check(self.L.vc_set_kernel_timing(self.h, int(on)), "set_kernel_timing")

    def sync_timeouts(self):
        """Device-flag hand-overs that ran into their bound so far (each one reported on stderr, the solve resumed with events)."""
        return int(self.L.vc_sync_timeouts(self.h))

    def kernel_timing(self):
        """{launch group: (launches, average ms)} of the solves run since set_kernel_timing(True)."""
        names = C.create_string_buffer(2048); tot = np.zeros(64); cnt = np.zeros(64, dtype=np.int64)
        n = _check(self.L.vc_get_kernel_timing(self.h, names, len(names), _d(tot), cnt.ctypes.data_as(C.c_void_p), 64), "kernel_timing")
        keys = names.value.decode().split(";") if n else []
        return {k: (int(cnt[i]), float(tot[i] / cnt[i])) for i, k in enumerate(keys)}

    def imu_weights(self):
        ns = max(self.NumFrames() - 1, 0)
        W = np.zeros((ns, 9, 9))
        _check(self.L.vc_get_imu_weights(self.h, _d(W)), "imu_weights")
        return W

    def GetSolutionCovariance(self):
        """(covariance n x n, block names) of q_ck / p_ck / params of every camera at the current state
        (GetSolutionCovariance, vicalibrator.h:802-857; names as in :563, :569, :596)."""
        n = _check(self.L.vc_solution_covariance_dim(self.h), "solution_covariance_dim")
        cov = np.zeros((n, n)); m = C.c_int(0)
        _check(self.L.vc_get_solution_covariance(self.h, _d(cov), n, C.byref(m)), "solution_covariance")
        buf = C.create_string_buffer(64 * max(1, self.NumCameras()))
        _check(self.L.vc_get_solution_covariance_names(self.h, buf, len(buf)), "solution_covariance_names")
        return cov, buf.value.decode().split()

    def imu_blocks(self):
        ns = max(self.NumFrames() - 1, 0)
        H = np.zeros((ns, 33, 33)); g = np.zeros((ns, 33)); c = np.zeros(ns)
        _check(self.L.vc_get_imu_blocks(self.h, _d(H), _d(g), _d(c)), "imu_blocks")
        return H, g, c

    def debug_stamps(self):
        out = np.zeros(32, dtype=np.int64)
        _check(self.L.vc_get_debug_stamps(self.h, out.ctypes.data_as(C.c_void_p)), "debug_stamps")
        return out

    def num_observations(self): return int(self.L.vc_num_observations(self.h))
    def num_tiles(self): return int(self.L.vc_num_tiles(self.h))


class ConicDetector:
    """The image front-end's first slice (include/vicalib_amd.h: vc_detector_*): what VicalibTask's image_processing_[i] /
    conic_finder_[i] pair does per image (vicalib-task.cc:264-270) -- adaptive threshold, dot components, one conic per dot --
    on the GPU.  find(image) -> centres [n, 2] (x, y)."""

    def __init__(self, width, height, device=0):
        self.L = load()
        self.h = C.c_void_p()
        _check(self.L.vc_detector_create(int(device), int(width), int(height), C.byref(self.h)), "detector_create")
        self.w, self.hh = int(width), int(height)

    def close(self):
        if getattr(self, "h", None):
            self.L.vc_detector_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def set_params(self, black_on_white=True, at_threshold=0.9, at_window_ratio=30.0, conic_min_area=4.0, conic_min_density=0.6, conic_min_aspect=0.2):
        _check(self.L.vc_detector_set_params(self.h, int(black_on_white), C.c_double(at_threshold), C.c_double(at_window_ratio), C.c_double(conic_min_area),
                                             C.c_double(conic_min_density), C.c_double(conic_min_aspect)), "detector_set_params")

    def find(self, image, max_conics=4096):
        image = np.ascontiguousarray(image, dtype=np.uint8)
        assert image.shape == (self.hh, self.w)
        out = np.empty((max_conics, 2)); n = C.c_int(0)
        _check(self.L.vc_detector_find(self.h, image.ctypes.data_as(C.c_void_p), int(image.strides[0]), _d(out), int(max_conics), C.byref(n)), "detector_find")
        return out[:min(n.value, max_conics)]

    def find_conics(self, image, max_conics=4096):
        """-> (centres [n, 2], conics [n, 3, 3] (x^T C x = 0 on the edge, unit Frobenius norm), boxes [n, 4] (x0, y0, x1, y1 inclusive))"""
        image = np.ascontiguousarray(image, dtype=np.uint8)
        assert image.shape == (self.hh, self.w)
        out = np.zeros((max_conics, 2)); con = np.zeros((max_conics, 9)); box = np.zeros((max_conics, 4), dtype=np.int32); n = C.c_int(0)
        _check(self.L.vc_detector_find_conics(self.h, image.ctypes.data_as(C.c_void_p), int(image.strides[0]), _d(out), _d(con),
                                              box.ctypes.data_as(C.c_void_p), int(max_conics), C.byref(n)), "detector_find_conics")
        k = min(n.value, max_conics)
        return out[:k], con[:k].reshape(-1, 3, 3), box[:k]
