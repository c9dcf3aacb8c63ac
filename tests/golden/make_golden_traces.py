#!/usr/bin/env python
"""Per-iteration traces of the CPU oracle's LM loop on small seeded problems: cost, gradient max-norm,
accepted flag, radius, stage for every iteration, and the converged shared parameters.  The fixture pins (a) the oracle
against its own regressions (tests/test_oracle.py, no GPU) and (b) the GPU solver at iteration level
(tests/test_gpu_parity.py).  Run from the repo root:  python tests/golden/make_golden_traces.py
(adds missing cases; --all regenerates the committed ones too)."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_lib as ol  # noqa: E402
from vicalib_amd import synth  # noqa: E402

CASES = {
    # name: (generator config, oracle options)
    "cfg1_poly3_50": (dict(models=("poly3",), n_frames=50), dict(calibrate_imu=False)),
    "stereo_fov_kb4_30": (dict(models=("fov", "kb4"), n_frames=30, seed=7), dict(calibrate_imu=False)),
    "mono_kb4_imu_60": (dict(models=("kb4",), n_frames=60, imu=True, seed=5), dict(calibrate_imu=True, max_iters=100)),
    "mono_rational6_40": (dict(models=("rational6",), n_frames=40, seed=9), dict(calibrate_imu=False)),
    "rig4_mixed_imu_80": (dict(models=("fov", "poly3", "kb4", "rational6"), n_frames=80, imu=True, seed=13, extrinsics_prior=True),
                          dict(calibrate_imu=True, max_iters=100)),
    # BASELINE cfg3 at FULL size (the configuration every bench number is quoted on): mono kb4 + IMU, 2000 frames, 373 493
    # corners, complete A->D schedule.  The oracle needs a few minutes for it; the GPU test reads the fixture only.
    "cfg3_full": (dict(models=("kb4",), grid="small", n_frames=2000, imu=True), dict(calibrate_imu=True)),
    # BASELINE cfg4's rig and grid (4 x poly3 + IMU, 900-dot grid) at 400 frames, D = 67
    "cfg4_rig_400": (dict(models=("poly3",) * 4, grid="large", n_frames=400, imu=True, seed=21), dict(calibrate_imu=True)),
    # BASELINE cfg5's rig (8 cameras fov / kb4 alternating + IMU, small grid, extrinsics prior) at 160 frames, D = 115: three image columns
    # per lane's worth of border in the chain elimination, the blocked reduced solve
    "cfg5_rig_160": (dict(models=("fov", "kb4") * 4, grid="small", n_frames=160, imu=True, seed=33, extrinsics_prior=True), dict(calibrate_imu=True)),
}


def run(cfg_kw, opt):
    p = synth.generate(synth.Config(**cfg_kw))
    orc = ol.Oracle().load(p)
    orc.set_options(num_threads=8, **opt)
    orc.solve()
    tr = orc.trace()
    out = {"config": {k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg_kw.items()}, "options": opt,
           "trace_columns": ["iteration", "cost", "gradient_max_norm", "accepted", "radius", "stage"],
           "trace": tr[:, [0, 1, 3, 8, 7, 9]].tolist(),
           "cameras": [{"K": orc.camera(c)[0].tolist(), "T_ck": orc.camera(c)[1].tolist()} for c in range(len(p.cam_model))],
           "rmse": np.asarray(orc.rmse()).tolist()}
    if opt.get("calibrate_imu"):
        b, s, g, toff = orc.imu_state()
        out["imu"] = {"biases": np.asarray(b).tolist(), "scale": np.asarray(s).tolist(), "gravity": np.asarray(g).tolist(), "time_offset": float(toff)}
    return out


if __name__ == "__main__":
    path = os.path.join(HERE, "lm_traces.json")
    data = json.load(open(path)) if os.path.exists(path) and "--all" not in sys.argv else {}
    data.update({name: run(*spec) for name, spec in CASES.items() if name not in data})      # committed entries stay as they are
    with open(path, "w") as f:
        json.dump(data, f, indent=0)
    for k, v in data.items():
        print(k, len(v["trace"]), "trace rows, final cost", v["trace"][-1][1])
