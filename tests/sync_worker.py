"""Worker of tests/test_flag_sync_gpu.py: one complete visual-inertial calibration in a process of its own (the hand-over mode, the
bound of the flag waits and the hardware-queue count are read from the environment when the HIP runtime / the calibrator start).
Writes the LM trace, the final state and the number of flag time-outs to the .npz given as argument."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vicalib_amd import synth                     # noqa: E402
from vicalib_amd.lib import ViCalibrator          # noqa: E402

out = sys.argv[1]
n_frames = int(sys.argv[2]) if len(sys.argv) > 2 else 60
models = tuple(os.environ.get("VICALIB_TEST_MODELS", "kb4").split(","))      # (the rig: one model name per camera)
p = synth.generate(synth.Config(models=models, n_frames=n_frames, imu=True, seed=5, extrinsics_prior=len(models) > 1))
cal = ViCalibrator(0).load_problem(p)
cal.SetMaxIters(100)
cal.Solve()
K, T = cal.GetCamera(0)
np.savez(out, trace=cal.trace(), K=K, T=T, frames=cal.GetFrames(), biases=cal.GetBiases(), timeouts=cal.sync_timeouts(),
         toff=cal.time_offset())
