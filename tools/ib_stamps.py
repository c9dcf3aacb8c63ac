"""Phase timeline of one wavefront of k_imu_block on a BASELINE workload (library built with -DVC_IB_STAMPS, VICALIB_AMD_LIB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vicalib_amd import synth
from vicalib_amd.lib import ViCalibrator
p = synth.generate_native(synth.BASELINE_CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg3"])
cal = ViCalibrator(0).load_problem(p)
cal.SetStageLimit(3); cal.Solve(); cal.prepare()
cal.run_iterations(5)
st = cal.debug_stamps().astype(float)
names = ["entry", "control record", "sample range (lanes), pointers", "first interval's delta (loads + RK4 step)", "second interval's delta", "append", "scan (3 levels)", "record stored"]
prev = st[0]
for i, n in enumerate(names):
    print("%-60s %8.2f us  (+%.2f)" % (n, (st[i] - st[0]) / 100.0, (st[i] - prev) / 100.0)); prev = st[i]
