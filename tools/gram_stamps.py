"""Phase timeline of one workgroup of k_chain_gram (library built with -DVC_GRAM_STAMPS, VICALIB_AMD_LIB pointing at it):
tools/gram_stamps.py cfg5 6250.  100 MHz ticks -> microseconds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vicalib_amd import synth
from vicalib_amd.lib import ViCalibrator
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
base = synth.BASELINE_CONFIGS[name]
if len(sys.argv) > 2:
    base = synth.Config(models=base.models, grid=base.grid, n_frames=int(sys.argv[2]), imu=base.imu, extrinsics_prior=base.extrinsics_prior)
p = synth.generate_native(base)
cal = ViCalibrator(0).load_problem(p)
cal.SetStageLimit(3); cal.Solve(); cal.prepare()
cal.run_iterations(5)
st = cal.debug_stamps().astype(float)
t0 = st[0]; prev = t0
for i in range(1, 16):
    if st[i] >= t0 and st[i] > 0:
        print("  stamp %2d  %8.2f us  (+%.2f)" % (i, (st[i] - t0) / 100.0, (st[i] - prev) / 100.0)); prev = st[i]
