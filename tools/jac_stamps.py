"""Phase stamps of the first pass of tile 0 in k_reproj_jac's sweep (library built with -DVC_JAC_STAMPS, VICALIB_AMD_LIB pointing at it):
shader-clock cycles.  tools/jac_stamps.py cfg4 2500"""
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
if p.imu_t is not None:
    cal.SetStageLimit(3); cal.Solve()
else:
    cal.SetCalibrateImu(False)
cal.prepare()
cal.run_iterations(3)
st = cal.debug_stamps().astype(float)[8:14]
names = ["entry", "first corner loaded", "rows of the first pass in LDS", "first pass's MFMA steps done", "all passes done", "Gram record written"]
for i in range(1, 6):
    print("%-32s +%8.0f cycles" % (names[i], st[i] - st[i - 1]))
print("total %.0f cycles" % (st[5] - st[0]))
