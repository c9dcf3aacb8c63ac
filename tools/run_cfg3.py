import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vicalib_amd import synth
from vicalib_amd.lib import ViCalibrator
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
t = time.time(); p = synth.generate(synth.Config(models=("kb4",), n_frames=n, imu=True)); print('gen %.1fs' % (time.time() - t), 'corners', p.n_obs, 'imu', len(p.imu_t))
cal = ViCalibrator(0).load_problem(p)
t = time.time(); cal.Solve(); dt = time.time() - t
tr = cal.trace()
its = int((tr[:, 0] > 0).sum())
print('solve %.3fs, %d LM iterations over %d stages -> %.1f it/s, %.2f ms/iter' % (dt, its, int(tr[-1, 9]) + 1, its / dt, 1e3 * dt / its))
for st in range(int(tr[-1, 9]) + 1):
    rows = tr[tr[:, 9] == st]
    print(' stage', st, 'iters', int(rows[-1, 0]), 'cost %.6g -> %.6g' % (rows[0, 1], rows[-1, 1]))
gt = p.imu_gt
print('rmse', cal.GetCameraProjRMSE(), 'toff', cal.time_offset(), 'gt', gt['time_offset'])
print('bias', cal.GetBiases(), 'gt', gt['bg'], gt['ba'])
print('scale', cal.GetScaleFactor()); print('g', cal.GetGravity(), 'gt', gt['g_dir'])
print('T_ck', cal.GetCamera(0)[1], 'gt', p.cam_T_ck_gt[0])
