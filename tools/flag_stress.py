"""Stress of the device-flag hand-overs: one small visual-inertial problem solved over and over with the flags; every solve's trace must be
bit-identical to the first solve with event hand-overs.  Run two instances at once to share the GPU between processes.
  python tools/flag_stress.py [n_frames] [repetitions]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vicalib_amd import synth
from vicalib_amd.lib import ViCalibrator
n = int(sys.argv[1]) if len(sys.argv) > 1 else 80
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
p = synth.generate(synth.Config(models=("kb4",), n_frames=n, imu=True, seed=5))
os.environ["VICALIB_AMD_FLAG_SYNC"] = "0"
ref = ViCalibrator(0).load_problem(p); ref.SetMaxIters(100); ref.Solve(); tr = ref.trace()
os.environ["VICALIB_AMD_FLAG_SYNC"] = "1"
bad = 0
for i in range(reps):
    cal = ViCalibrator(0).load_problem(p); cal.SetMaxIters(100); cal.Solve(); t = cal.trace()
    same = t.shape == tr.shape and np.array_equal(t[:, 1], tr[:, 1])
    if not same:
        bad += 1
        k = int(np.argmax(t[:, 1] != tr[:, 1])) if t.shape == tr.shape else -1
        print("solve", i, "differs from the event run at row", k, "timeouts", cal.sync_timeouts(), t[k:k + 3, 1] if k >= 0 else t.shape, tr[k:k + 3, 1] if k >= 0 else tr.shape, flush=True)
print("pid", os.getpid(), "frames", n, "solves", reps, "differing", bad, flush=True)
