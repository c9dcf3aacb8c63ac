"""Phase stamps of one group of k_chain_back per level (library built with -DVC_BACK_STAMPS, VICALIB_AMD_LIB pointing at it).
100 MHz ticks -> microseconds.  tools/back_stamps.py cfg5 6250"""
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
names = ["delta_s, separators requested", "t = z + Y delta_s + X_s delta_a", "dependent chain", "stores / trial states"]
for lvl in range(4):
    x = st[8 * lvl: 8 * lvl + 5]
    if x[0] <= 0: continue
    print("level %d: " % lvl + ", ".join("%s +%.2f" % (names[i], (x[i + 1] - x[i]) / 100.0) for i in range(4)) + "  | total %.2f us" % ((x[4] - x[0]) / 100.0))
