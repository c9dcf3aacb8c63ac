"""Phase timeline of the two wavefronts of group 0, level 1 of k_chain_fwd2 on BASELINE cfg3 (library built with -DVC_F2_STAMPS,
VICALIB_AMD_LIB pointing at it).  100 MHz ticks -> microseconds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vicalib_amd import synth
from vicalib_amd.lib import ViCalibrator
p = synth.generate_native(synth.BASELINE_CONFIGS["cfg3"])
cal = ViCalibrator(0).load_problem(p)
cal.SetStageLimit(3); cal.Solve(); cal.prepare()
cal.run_iterations(5)
st = cal.debug_stamps().astype(float)
t0 = st[0]
for w in (0, 1):
    print("wave", w)
    prev = t0
    for i in range(16):
        x = st[i + 16 * w]
        if x >= t0:
            print("  stamp %2d  %8.2f us  (+%.2f)" % (i, (x - t0) / 100.0, (x - prev) / 100.0)); prev = x
