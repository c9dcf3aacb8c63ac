"""Deterministic synthetic calibration problems.

Stands in for the reference's sensor front-end (vicalib-task.cc:247-372: conic
detection -> PnP -> AddFrame/AddObservation, and vicalib-engine.cc:557 IMU
handler): produces exactly the calls the reference makes on ViCalibrator --
cameras with the engine's start values (vicalib-engine.cc:203-263), frames with
an initial pose, per-(frame, camera) dot detections p_w = spacing * (i, j, 0)
(vicalib-task.cc:357-358) and 200 Hz IMU samples generated through the
reference's own measurement model (ceres-cost-functions.h:98-102).

Everything is a pure function of (config, seed, frame index), so every rank of a
multi-GPU run can build its own frame shard without communication.
Pure numpy; no dependency on the HIP library or on oracle/.
"""
from __future__ import annotations

import dataclasses
import numpy as np

MODEL_IDS = {"fov": 0, "poly2": 1, "poly3": 2, "poly": 2, "kb4": 3, "linear": 4, "rational6": 5, "rational": 5}
MODEL_NK = {0: 5, 1: 6, 2: 7, 3: 8, 4: 4, 5: 10}
GRAVITY = 9.8007  # types.h:40-42

GT_INTRINSICS = {
    0: [330.0, 330.0, 320.0, 240.0, 0.92],
    1: [400.0, 400.0, 320.0, 240.0, -0.28, 0.09],
    2: [400.0, 400.0, 320.0, 240.0, -0.28, 0.09, -0.012],
    3: [260.0, 260.0, 320.0, 240.0, -0.012, 0.004, -0.0015, 0.0002],
    4: [400.0, 400.0, 320.0, 240.0],
    5: [400.0, 400.0, 320.0, 240.0, 0.12, 0.05, 0.004, 0.40, -0.04, 0.002],
}
GRIDS = {"small": (19, 10, 0.254 / 18.0), "large": (25, 36, 0.03156)}
RDF_ROBOTICS = np.array([[0.0, 1.0, 0.0], [0.0, 0.0, 1.0], [1.0, 0.0, 0.0]])


# ----------------------------------------------------------------------------- rng
def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def hash_uniform(seed: int, *keys) -> np.ndarray:
    """Counter-based uniform(0,1): a pure function of (seed, keys...)."""
    with np.errstate(over="ignore"):
        h = np.uint64(seed & 0xFFFFFFFFFFFFFFFF)
        h = _splitmix64(np.asarray(h, dtype=np.uint64))
        for k in keys:
            k = np.asarray(k).astype(np.uint64)
            h = _splitmix64(h ^ (k * np.uint64(0xD6E8FEB86659FD93)))
    return ((h >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def hash_normal(seed: int, *keys) -> np.ndarray:
    u1 = hash_uniform(seed, *keys, 0x11)
    u2 = hash_uniform(seed, *keys, 0x22)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


# ----------------------------------------------------------------------------- lie helpers ([x,y,z,w] quaternions)
def quat_from_matrix(R: np.ndarray) -> np.ndarray:
    R = np.asarray(R, dtype=np.float64)
    out = np.empty(R.shape[:-2] + (4,))
    Rf = R.reshape(-1, 3, 3)
    of = out.reshape(-1, 4)
    for i, m in enumerate(Rf):
        tr = m[0, 0] + m[1, 1] + m[2, 2]
        if tr > 0:
            s = np.sqrt(tr + 1.0) * 2
            q = [(m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s, 0.25 * s]
        elif m[0, 0] > m[1, 1] and m[0, 0] > m[2, 2]:
            s = np.sqrt(1.0 + m[0, 0] - m[1, 1] - m[2, 2]) * 2
            q = [0.25 * s, (m[0, 1] + m[1, 0]) / s, (m[0, 2] + m[2, 0]) / s, (m[2, 1] - m[1, 2]) / s]
        elif m[1, 1] > m[2, 2]:
            s = np.sqrt(1.0 + m[1, 1] - m[0, 0] - m[2, 2]) * 2
            q = [(m[0, 1] + m[1, 0]) / s, 0.25 * s, (m[1, 2] + m[2, 1]) / s, (m[0, 2] - m[2, 0]) / s]
        else:
            s = np.sqrt(1.0 + m[2, 2] - m[0, 0] - m[1, 1]) * 2
            q = [(m[0, 2] + m[2, 0]) / s, (m[1, 2] + m[2, 1]) / s, 0.25 * s, (m[1, 0] - m[0, 1]) / s]
        q = np.array(q)
        of[i] = q / np.linalg.norm(q)
    return out


def quat_to_matrix(q: np.ndarray) -> np.ndarray:
    q = np.asarray(q, dtype=np.float64)
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    R = np.empty(q.shape[:-1] + (3, 3))
    R[..., 0, 0] = 1 - 2 * (y * y + z * z); R[..., 0, 1] = 2 * (x * y - w * z); R[..., 0, 2] = 2 * (x * z + w * y)
    R[..., 1, 0] = 2 * (x * y + w * z); R[..., 1, 1] = 1 - 2 * (x * x + z * z); R[..., 1, 2] = 2 * (y * z - w * x)
    R[..., 2, 0] = 2 * (x * z - w * y); R[..., 2, 1] = 2 * (y * z + w * x); R[..., 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def so3_exp_matrix(w: np.ndarray) -> np.ndarray:
    w = np.asarray(w, dtype=np.float64)
    th = np.linalg.norm(w, axis=-1)[..., None, None]
    K = np.zeros(w.shape[:-1] + (3, 3))
    K[..., 0, 1] = -w[..., 2]; K[..., 0, 2] = w[..., 1]
    K[..., 1, 0] = w[..., 2]; K[..., 1, 2] = -w[..., 0]
    K[..., 2, 0] = -w[..., 1]; K[..., 2, 1] = w[..., 0]
    th2 = th * th
    with np.errstate(divide="ignore", invalid="ignore"):
        a = np.where(th > 1e-8, np.sin(th) / th, 1.0 - th2 / 6.0)
        b = np.where(th > 1e-8, (1.0 - np.cos(th)) / th2, 0.5 - th2 / 24.0)
    return np.eye(3) + a * K + b * (K @ K)


def se3_from_Rt(R: np.ndarray, t: np.ndarray) -> np.ndarray:
    """[qx,qy,qz,qw,tx,ty,tz] -- Sophus SE3 data order."""
    return np.concatenate([quat_from_matrix(R), np.asarray(t, dtype=np.float64)], axis=-1)


# ----------------------------------------------------------------------------- camera models (generator side)
def project(model: int, K: np.ndarray, P: np.ndarray) -> np.ndarray:
    """Pixel of camera-frame points P (...,3). Same formulas as SURVEY 9.1."""
    X, Y, Z = P[..., 0], P[..., 1], P[..., 2]
    if model == 3:
        rxy = np.sqrt(X * X + Y * Y)
        th = np.arctan2(rxy, Z)
        th2 = th * th
        r = th * (1 + th2 * (K[4] + th2 * (K[5] + th2 * (K[6] + th2 * K[7]))))
        with np.errstate(divide="ignore", invalid="ignore"):
            c = np.where(rxy > 0, X / rxy, 1.0)
            s = np.where(rxy > 0, Y / rxy, 0.0)
        return np.stack([K[0] * r * c + K[2], K[1] * r * s + K[3]], axis=-1)
    x, y = X / Z, Y / Z
    r2 = x * x + y * y
    if model == 0:
        r = np.sqrt(r2)
        w = K[4]
        m = 2.0 * np.tan(w / 2.0)
        with np.errstate(divide="ignore", invalid="ignore"):
            fac = np.where(r2 < 1e-5, m / w, np.arctan(r * m) / (r * w))
    elif model == 1:
        fac = 1 + K[4] * r2 + K[5] * r2 * r2
    elif model == 2:
        fac = 1 + K[4] * r2 + K[5] * r2 * r2 + K[6] * r2 * r2 * r2
    elif model == 5:
        fac = (1 + K[4] * r2 + K[5] * r2 * r2 + K[6] * r2 * r2 * r2) / (1 + K[7] * r2 + K[8] * r2 * r2 + K[9] * r2 * r2 * r2)
    else:
        fac = np.ones_like(r2)
    return np.stack([K[0] * x * fac + K[2], K[1] * y * fac + K[3]], axis=-1)


# ----------------------------------------------------------------------------- problem description
@dataclasses.dataclass
class Config:
    models: tuple            # e.g. ("fov", "fov")
    grid: str = "small"
    n_frames: int = 50
    imu: bool = False
    seed: int = 1234
    width: int = 640
    height: int = 480
    frame_rate: float = 20.0
    imu_rate: float = 200.0
    pixel_sigma: float = 0.1
    pose_sigma_t: float = 0.01
    pose_sigma_r: float = np.deg2rad(1.5)
    first_frame: int = 0     # global index of the first frame generated (frame sharding)
    gt_intrinsics: tuple = None      # ((k...), ...) per camera: exact ground-truth intrinsics instead of the jittered defaults
    imu_truth: dict = None           # overrides of the IMU ground truth (bg, ba, sg, sa, g_dir, time_offset)
    imu_noise: bool = True           # white noise on the IMU samples at the reference's default sigmas
    extrinsics_prior: bool = False   # start T_ck of cameras 1.. from a rough prior (GT o exp(1 cm, 0.5 deg)) instead of the
                                     # engine's identity (vicalib-engine.cc:211) -- a rig description from `-model_files`


BASELINE_CONFIGS = {
    # BASELINE.json "configs", in order
    "cfg1": Config(models=("poly3",), grid="small", n_frames=50, imu=False),
    "cfg2": Config(models=("fov", "fov"), grid="small", n_frames=500, imu=False),
    "cfg3": Config(models=("kb4",), grid="small", n_frames=2000, imu=True),
    "cfg4": Config(models=("poly3",) * 4, grid="large", n_frames=10000, imu=True),
    # cfg5's rig spans 0.42 m next to a 0.25 m target: from the engine's identity extrinsics the vision-only stage needs > 200 LM
    # iterations (NO_CONVERGENCE, then the reference solves again); the 8-camera rig therefore starts from a rough prior, as a
    # rig description passed with -model_files would provide
    "cfg5": Config(models=("fov", "kb4") * 4, grid="small", n_frames=50000, imu=True, extrinsics_prior=True),
}


@dataclasses.dataclass
class Problem:
    cfg: Config
    grid_points: np.ndarray      # (M,3)
    cam_model: list              # model ids
    cam_K_gt: list               # GT intrinsics
    cam_K_init: list             # reference start values (vicalib-engine.cc:207-257)
    cam_T_ck_gt: np.ndarray      # (C,7)
    cam_T_ck_init: np.ndarray    # (C,7)
    frame_time: np.ndarray       # (N,)
    frame_T_wk_gt: np.ndarray    # (N,7)
    frame_T_wk_init: np.ndarray  # (N,7)
    frame_v_gt: np.ndarray       # (N,3)
    tiles: list                  # [(frame, cam, dot_ids (n,), pix (n,2))]
    imu_t: np.ndarray = None     # (S,)
    imu_gyro: np.ndarray = None  # (S,3)
    imu_accel: np.ndarray = None # (S,3)
    imu_gt: dict = None

    flat: tuple = None           # native generator: (tile_frame, tile_cam, tile_off int64 (T+1), ids int32 (n), pix (n,2))

    @property
    def n_obs(self) -> int:
        if self.flat is not None:
            return int(self.flat[2][-1])
        return int(sum(len(t[2]) for t in self.tiles))

    def __getattribute__(self, name):
        # `tiles` of a natively generated problem is built on first use (4e5 tuples at cfg5: most callers use `flat`)
        if name == "tiles":
            t = object.__getattribute__(self, "tiles")
            if t is None and object.__getattribute__(self, "flat") is not None:
                tf, tc, off, ids, pix = object.__getattribute__(self, "flat")
                t = [(int(tf[k]), int(tc[k]), ids[off[k]:off[k + 1]], pix[off[k]:off[k + 1]]) for k in range(len(tf))]
                object.__setattr__(self, "tiles", t)
            return t
        return object.__getattribute__(self, name)


def _trajectory(cfg: Config, t: np.ndarray, grid_w: float, grid_h: float):
    """Camera-0 pose in the world at times t: position (…,3), R_wc (…,3,3)."""
    cx, cy = 0.5 * grid_w, 0.5 * grid_h
    d = grid_w * (0.625 + 0.275 * np.sin(2 * np.pi * t / 6.3 + 1.0))      # in [0.35, 0.9] * grid_w
    x = cx + 0.30 * grid_w * np.sin(2 * np.pi * t / 3.1)
    y = cy + 0.30 * grid_h * np.sin(2 * np.pi * t / 4.7 + 0.5)
    p = np.stack([x, y, -d], axis=-1)
    # look roughly at a wandering point on the grid, then add roll
    tx = cx + 0.25 * grid_w * np.sin(2 * np.pi * t / 5.3 + 2.0)
    ty = cy + 0.25 * grid_h * np.sin(2 * np.pi * t / 3.7 + 0.3)
    target = np.stack([tx, ty, np.zeros_like(t)], axis=-1)
    z = target - p
    z = z / np.linalg.norm(z, axis=-1, keepdims=True)
    roll = np.deg2rad(25.0) * np.sin(2 * np.pi * t / 7.9 + 0.7)
    up = np.stack([np.sin(roll), np.cos(roll), np.zeros_like(t)], axis=-1)   # image "down" ~ +y world
    xax = np.cross(up, z)
    xax = xax / np.linalg.norm(xax, axis=-1, keepdims=True)
    yax = np.cross(z, xax)
    R = np.stack([xax, yax, z], axis=-1)   # columns = camera axes in the world
    return p, R


def generate(cfg: Config) -> Problem:
    gw, gh, sp = GRIDS[cfg.grid]
    ii, jj = np.meshgrid(np.arange(gw), np.arange(gh), indexing="ij")
    grid = np.stack([ii.ravel() * sp, jj.ravel() * sp, np.zeros(gw * gh)], axis=-1)
    grid_w, grid_h = (gw - 1) * sp, (gh - 1) * sp
    C = len(cfg.models)
    models = [MODEL_IDS[m] for m in cfg.models]
    seed = cfg.seed

    # cameras ---------------------------------------------------------------------------
    K_gt, K_init = [], []
    T_ck_gt = np.zeros((C, 7)); T_ck_init = np.zeros((C, 7))
    R_ck0 = RDF_ROBOTICS if cfg.imu else np.eye(3)
    for c, m in enumerate(models):
        base = np.array(GT_INTRINSICS[m])
        jit = 1.0 + 0.02 * (2.0 * hash_uniform(seed + 17, c, np.arange(len(base))) - 1.0)
        K_gt.append(np.array(cfg.gt_intrinsics[c], dtype=np.float64) if cfg.gt_intrinsics is not None else base * jit)
        k0 = [300.0, 300.0, cfg.width / 2.0, cfg.height / 2.0] + ([0.2] if m == 0 else [0.0] * (MODEL_NK[m] - 4))
        K_init.append(np.array(k0))
        # body -> camera c: camera c sits 0.06*c along camera-0 x (the rig's lateral axis), small rotation
        w = np.deg2rad(3.0) * (2.0 * hash_uniform(seed + 29, c, np.arange(3)) - 1.0) * (c > 0)
        R_c_c0 = so3_exp_matrix(w)                      # camera0 -> camera c rotation
        t_c_c0 = -R_c_c0 @ np.array([0.06 * c, 0.0, 0.0])
        R_ck = R_c_c0 @ R_ck0
        T_ck_gt[c] = se3_from_Rt(R_ck, t_c_c0)
        T_ck_init[c] = np.array([0, 0, 0, 1.0, 0, 0, 0])   # Sophus::SE3d(), vicalib-engine.cc:211
        if cfg.extrinsics_prior and c > 0:
            dw = np.deg2rad(0.5) * (2.0 * hash_uniform(seed + 31, c, np.arange(3)) - 1.0)
            dtr = 0.01 * (2.0 * hash_uniform(seed + 37, c, np.arange(3)) - 1.0)
            # expressed relative to camera 0, whose T_ck starts at identity: T_c_c0 with a small error
            T_ck_init[c] = se3_from_Rt(so3_exp_matrix(dw) @ R_c_c0, t_c_c0 + dtr)
    # frames ----------------------------------------------------------------------------
    f_idx = cfg.first_frame + np.arange(cfg.n_frames)
    ft = 1.0 + f_idx / cfg.frame_rate
    p_c0, R_wc0 = _trajectory(cfg, ft, grid_w, grid_h)
    R_wk = R_wc0 @ R_ck0                                 # body orientation
    T_wk_gt = se3_from_Rt(R_wk, p_c0)
    # velocity (world) by central difference of the analytic position
    h = 1e-5
    pp, _ = _trajectory(cfg, ft + h, grid_w, grid_h); pm, _ = _trajectory(cfg, ft - h, grid_w, grid_h)
    v_gt = (pp - pm) / (2 * h)
    # initial poses: the reference seeds T_wk = T_cw^-1 * T_ck with T_ck = identity (vicalib-task.cc:347-348),
    # i.e. the (PnP-estimated) camera-0 pose; PnP error is modelled as a right perturbation.
    dt = cfg.pose_sigma_t * hash_normal(seed + 41, f_idx[:, None], np.arange(3)[None, :])
    dr = cfg.pose_sigma_r * hash_normal(seed + 43, f_idx[:, None], np.arange(3)[None, :])
    R_init = R_wc0 @ so3_exp_matrix(dr)
    t_init = p_c0 + np.einsum("nij,nj->ni", R_wc0, dt)
    T_wk_init = se3_from_Rt(R_init, t_init)
    # detections ------------------------------------------------------------------------
    tiles = []
    M = grid.shape[0]
    for c, m in enumerate(models):
        R_ck = quat_to_matrix(T_ck_gt[c, :4]); t_ck = T_ck_gt[c, 4:]
        # p_k = R_wk^T (p_w - t_wk); p_c = R_ck p_k + t_ck
        pk = np.einsum("nji,nmj->nmi", R_wk, grid[None, :, :] - p_c0[:, None, :])
        pc = np.einsum("ij,nmj->nmi", R_ck, pk) + t_ck
        pix = project(m, K_gt[c], pc)
        ok = (pc[..., 2] > 1e-3) & (pix[..., 0] >= 5) & (pix[..., 0] <= cfg.width - 5) & (pix[..., 1] >= 5) & (pix[..., 1] <= cfg.height - 5)
        # reject points whose incidence angle is outside a sane field of view (model extrapolation)
        ang = np.arctan2(np.linalg.norm(pc[..., :2], axis=-1), pc[..., 2])
        ok &= ang < np.deg2rad(75.0 if m == 3 else 55.0)
        noise = cfg.pixel_sigma * hash_normal(seed + 5678, f_idx[:, None, None], c, np.arange(M)[None, :, None], np.arange(2)[None, None, :])
        pix = pix + noise
        for n in range(cfg.n_frames):
            ids = np.nonzero(ok[n])[0]
            if len(ids) >= 4:
                tiles.append((n, c, ids.astype(np.int32), pix[n, ids]))
    tiles.sort(key=lambda t: (t[0], t[1]))
    prob = Problem(cfg, grid, models, K_gt, K_init, T_ck_gt, T_ck_init, ft, T_wk_gt, T_wk_init, v_gt, tiles)
    if cfg.imu:
        _add_imu(prob, grid_w, grid_h, R_ck0)
    return prob


def gravity_vector(g_dir) -> np.ndarray:
    p, q = g_dir
    return -GRAVITY * np.array([np.cos(p) * np.sin(q), -np.sin(p), np.cos(p) * np.cos(q)])


def _add_imu(prob: Problem, grid_w: float, grid_h: float, R_ck0: np.ndarray) -> None:
    cfg = prob.cfg
    gt = dict(bg=np.array([0.002, -0.001, 0.0015]), ba=np.array([0.03, -0.02, 0.05]),
              sg=np.array([1.01, 0.99, 1.005]), sa=np.array([0.995, 1.01, 0.99]),
              g_dir=np.array([0.03, -0.02]), time_offset=0.003)
    if cfg.imu_truth:
        gt.update({k: (np.asarray(v, dtype=np.float64) if k != "time_offset" else float(v)) for k, v in cfg.imu_truth.items()})
    t0 = prob.frame_time[0] - 0.1
    t1 = prob.frame_time[-1] + 0.1
    k0 = int(np.floor(t0 * cfg.imu_rate)); k1 = int(np.ceil(t1 * cfg.imu_rate))
    k = np.arange(k0, k1 + 1)
    # the buffer compares (sample stamp + offset) with frame times (interpolation-buffer.h:122-125, :163-199):
    # a sample stamped t_imu was taken at image-clock time t_imu + offset
    t_imu = k / cfg.imu_rate
    tb = t_imu + gt["time_offset"]
    h = 1e-4
    p0, R0 = _trajectory(cfg, tb, grid_w, grid_h)
    pp, Rp = _trajectory(cfg, tb + h, grid_w, grid_h)
    pm, Rm = _trajectory(cfg, tb - h, grid_w, grid_h)
    a_w = (pp - 2 * p0 + pm) / (h * h)
    Rk0 = R0 @ R_ck0; Rkp = Rp @ R_ck0; Rkm = Rm @ R_ck0
    Wx = (Rkp - Rkm) / (2 * h) @ np.swapaxes(Rk0, -1, -2)          # [w_w]x = Rdot R^T
    w_w = np.stack([Wx[:, 2, 1] - Wx[:, 1, 2], Wx[:, 0, 2] - Wx[:, 2, 0], Wx[:, 1, 0] - Wx[:, 0, 1]], axis=-1) * 0.5
    g_w = gravity_vector(gt["g_dir"])
    # k_w = R (z_g * s_g + b_g) ; k_a = R (z_a * s_a + b_a) - g_w   (ceres-cost-functions.h:98-102)
    zg = (np.einsum("nji,nj->ni", Rk0, w_w) - gt["bg"]) / gt["sg"]
    za = (np.einsum("nji,nj->ni", Rk0, a_w + g_w) - gt["ba"]) / gt["sa"]
    if cfg.imu_noise:
        zg = zg + 5.3088444e-5 * hash_normal(cfg.seed + 9001, k[:, None], np.arange(3)[None, :])
        za = za + 0.001883649 * hash_normal(cfg.seed + 9002, k[:, None], np.arange(3)[None, :])
    prob.imu_t, prob.imu_gyro, prob.imu_accel, prob.imu_gt = t_imu, zg, za, gt


# ----------------------------------------------------------------------------- native generator (csrc/vc_synth.cpp)
_synth_lib = None


def _native_lib():
    global _synth_lib
    if _synth_lib is None:
        import ctypes as C
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libvicalib_synth.so")
        if not os.path.exists(path):
            raise RuntimeError(path + " is missing: build it with vicalib_amd/csrc/build.sh")
        L = C.CDLL(path)
        L.vcs_generate.restype = C.c_void_p; L.vcs_generate.argtypes = [C.c_void_p, C.c_int]
        L.vcs_free.restype = None; L.vcs_free.argtypes = [C.c_void_p]
        L.vcs_count.restype = C.c_longlong; L.vcs_count.argtypes = [C.c_void_p, C.c_int]
        L.vcs_array.restype = C.c_void_p; L.vcs_array.argtypes = [C.c_void_p, C.c_int]
        _synth_lib = L
    return _synth_lib


def vi_sim_config(n_frames=120, seed=1234):
    """The configuration of the reference's only end-to-end test, testing/vi_sim_test.cpp:18-23, :70-92: one `linear` camera, 800 x 600
    images, ground truth T_ck = [[0,1,0],[0,0,1],[1,0,0]] (zero translation), intrinsics (335.639853151, 335.639853151, 400, 300),
    no time offset, an ideal IMU (zero biases, unit scale factors), noise-free simulated detections.  Its image / IMU files are
    not in the reference tree and the `medium` grid preset's dimensions live in Calibu: the large preset's grid and this
    generator's trajectory stand in."""
    return Config(models=("linear",), grid="large", n_frames=n_frames, imu=True, seed=seed, width=800, height=600, pixel_sigma=0.0,
                  gt_intrinsics=((335.639853151, 335.639853151, 400.0, 300.0),), imu_noise=False,
                  imu_truth=dict(bg=(0, 0, 0), ba=(0, 0, 0), sg=(1, 1, 1), sa=(1, 1, 1), g_dir=(0.0, 0.0), time_offset=0.0))


def generate_native(cfg: Config, threads: int = 0) -> Problem:
    if cfg.gt_intrinsics is not None or cfg.imu_truth is not None or not cfg.imu_noise:
        raise ValueError("generate_native: gt_intrinsics / imu_truth / imu_noise are options of the numpy generator only")
    """The same problem as generate(cfg), produced by the C++ generator (all host cores): identical visible-dot sets,
    floating-point fields equal up to the last bits of the two math libraries.  BASELINE cfg4 / cfg5 take seconds."""
    import ctypes as C

    class _Cfg(C.Structure):
        _fields_ = [("n_cams", C.c_int), ("models", C.c_int * 8), ("grid", C.c_int), ("n_frames", C.c_int), ("imu", C.c_int),
                    ("seed", C.c_longlong), ("width", C.c_int), ("height", C.c_int), ("frame_rate", C.c_double),
                    ("imu_rate", C.c_double), ("pixel_sigma", C.c_double), ("pose_sigma_t", C.c_double),
                    ("pose_sigma_r", C.c_double), ("first_frame", C.c_longlong), ("threads", C.c_int), ("extrinsics_prior", C.c_int)]
    L = _native_lib()
    models = [MODEL_IDS[m] for m in cfg.models]
    c = _Cfg()
    c.n_cams = len(models)
    for i, m in enumerate(models):
        c.models[i] = m
    c.grid = 1 if cfg.grid == "large" else 0
    c.n_frames = cfg.n_frames; c.imu = int(cfg.imu); c.seed = cfg.seed; c.width = cfg.width; c.height = cfg.height
    c.frame_rate = cfg.frame_rate; c.imu_rate = cfg.imu_rate; c.pixel_sigma = cfg.pixel_sigma
    c.pose_sigma_t = cfg.pose_sigma_t; c.pose_sigma_r = float(cfg.pose_sigma_r); c.first_frame = cfg.first_frame; c.threads = threads; c.extrinsics_prior = int(cfg.extrinsics_prior)
    assert L.vcs_config_size() == C.sizeof(_Cfg)
    h = L.vcs_generate(C.byref(c), C.sizeof(_Cfg))
    if not h:
        raise RuntimeError("vcs_generate failed")
    try:
        def arr(what, n, dtype, shape=None):
            if n == 0:
                a = np.zeros(0, dtype=dtype)
            else:
                ptr = L.vcs_array(h, what)
                ct = {np.float64: C.c_double, np.int32: C.c_int, np.int64: C.c_longlong}[dtype]
                a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(n,)).copy()
            return a.reshape(shape) if shape else a
        M, T, n, S, N, Cn = (int(L.vcs_count(h, k)) for k in range(6))
        grid = arr(0, M * 3, np.float64, (M, 3))
        Kg = arr(1, Cn * 10, np.float64, (Cn, 10)); Ki = arr(2, Cn * 10, np.float64, (Cn, 10))
        K_gt = [Kg[i, :MODEL_NK[m]].copy() for i, m in enumerate(models)]
        K_init = [Ki[i, :MODEL_NK[m]].copy() for i, m in enumerate(models)]
        flat = (arr(9, T, np.int32), arr(10, T, np.int32), arr(11, T + 1, np.int64), arr(12, n, np.int32), arr(13, n * 2, np.float64, (n, 2)))
        prob = Problem(cfg, grid, models, K_gt, K_init, arr(3, Cn * 7, np.float64, (Cn, 7)), arr(4, Cn * 7, np.float64, (Cn, 7)),
                       arr(5, N, np.float64), arr(6, N * 7, np.float64, (N, 7)), arr(7, N * 7, np.float64, (N, 7)),
                       arr(8, N * 3, np.float64, (N, 3)), None, flat=flat)
        if cfg.imu and S:
            gt = arr(17, 15, np.float64)
            prob.imu_t = arr(14, S, np.float64); prob.imu_gyro = arr(15, S * 3, np.float64, (S, 3)); prob.imu_accel = arr(16, S * 3, np.float64, (S, 3))
            prob.imu_gt = dict(bg=gt[0:3], ba=gt[3:6], sg=gt[6:9], sa=gt[9:12], g_dir=gt[12:14], time_offset=float(gt[14]))
        return prob
    finally:
        L.vcs_free(h)


# ----------------------------------------------------------------------------- flat arrays for bulk ingest
def flatten(prob: Problem):
    """tile_frame, tile_cam, tile_off (T+1), p_w (n,3), p_c (n,2) in tile order."""
    if prob.flat is not None:
        tf, tc, off, ids, pix = prob.flat
        return tf, tc, off, np.ascontiguousarray(prob.grid_points[ids]), pix
    tf = np.array([t[0] for t in prob.tiles], dtype=np.int32)
    tc = np.array([t[1] for t in prob.tiles], dtype=np.int32)
    cnt = np.array([len(t[2]) for t in prob.tiles], dtype=np.int64)
    off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
    ids = np.concatenate([t[2] for t in prob.tiles]) if prob.tiles else np.zeros(0, np.int32)
    pw = prob.grid_points[ids]
    pc = np.concatenate([t[3] for t in prob.tiles]) if prob.tiles else np.zeros((0, 2))
    return tf, tc, off, np.ascontiguousarray(pw), np.ascontiguousarray(pc)


def write_dataset(prob, directory):
    """Write `prob` as the files the command-line tool reads: one detections CSV per camera
    (`frame,dot_id,u,v,X,Y,Z,time` -- the -output_conics line of vicalib-task.cc:313-317 plus a time column) and,
    with an IMU, HAL's csv:// layout (accel.txt, gyro.txt, timestamp.txt).  Returns (cam_files, imu_dir or None)."""
    import os
    os.makedirs(directory, exist_ok=True)
    n_cam = len(prob.cam_model)
    files = [os.path.join(directory, "cam%d.csv" % c) for c in range(n_cam)]
    handles = [open(f, "w") for f in files]
    for (f, c, ids, pix) in prob.tiles:
        P = prob.grid_points[ids]
        t = prob.frame_time[f]
        for k in range(len(ids)):
            handles[c].write("%d,%d,%.17g,%.17g,%.17g,%.17g,%.17g,%.17g\n" % (f, ids[k], pix[k, 0], pix[k, 1], P[k, 0], P[k, 1], P[k, 2], t))
    for h in handles:
        h.close()
    imu_dir = None
    if prob.imu_t is not None:
        imu_dir = os.path.join(directory, "imu")
        os.makedirs(imu_dir, exist_ok=True)
        np.savetxt(os.path.join(imu_dir, "accel.txt"), prob.imu_accel, fmt="%.17g", delimiter=",")
        np.savetxt(os.path.join(imu_dir, "gyro.txt"), prob.imu_gyro, fmt="%.17g", delimiter=",")
        np.savetxt(os.path.join(imu_dir, "timestamp.txt"), np.stack([prob.imu_t, prob.imu_t], axis=1), fmt="%.17g", delimiter=",")
    return files, imu_dir
