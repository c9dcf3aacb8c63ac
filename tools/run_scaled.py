"""Scaled-down BASELINE cfg4 / cfg5 on one GPU: complete visual-inertial calibration, timing and ground-truth recovery.
usage: python tools/run_scaled.py cfg4|cfg5 [n_frames]"""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vicalib_amd import synth
from vicalib_amd.lib import ViCalibrator
which = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
cfg = synth.Config(models=("poly3",) * 4, grid="large", n_frames=n, imu=True) if which == "cfg4" else synth.Config(models=("fov", "kb4") * 4, grid="small", n_frames=n, imu=True)
t = time.time(); p = synth.generate(cfg); print(which, 'frames', n, 'gen %.1fs' % (time.time() - t), 'corners', p.n_obs, 'imu', len(p.imu_t), flush=True)
cal = ViCalibrator(0).load_problem(p)
t = time.time(); cal.Solve(); dt = time.time() - t
tr = cal.trace(); its = int((tr[:, 0] > 0).sum())
print('solve %.3fs, %d LM iterations over %d stages, D=%d -> %.2f ms/iter, %.3g corner-residuals/s' % (dt, its, int(tr[-1, 9]) + 1, cal.shared_dim(), 1e3 * dt / its, p.n_obs * its / dt))
gt = p.imu_gt
print('rmse', np.round(cal.GetCameraProjRMSE(), 4), 'toff %.6f (gt %.6f)' % (cal.time_offset(), gt['time_offset']))
print('bias err', np.abs(cal.GetBiases() - np.concatenate([gt['bg'], gt['ba']])).max(), 'scale err', np.abs(cal.GetScaleFactor() - np.concatenate([gt['sg'], gt['sa']])).max(), 'g err', np.abs(cal.GetGravity() - gt['g_dir']).max())
print('K err (rel, max over cameras)', max(np.abs(cal.GetCamera(c)[0][:4] / p.cam_K_gt[c][:4] - 1).max() for c in range(len(p.cam_model))))
