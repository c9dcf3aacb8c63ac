"""Phase timeline of the first wavefront of k_chain_fwd (level 0) on BASELINE cfg3: build the library with
-DVC_CHAIN_STAMPS (bash vicalib_amd/csrc/build.sh -DVC_CHAIN_STAMPS after copying to a scratch name) and run with
VICALIB_AMD_LIB pointing at it.  100 MHz s_memrealtime ticks -> microseconds."""
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
names = {0: "entry", 1: "ctrl loaded", 30: "loop done"}
for i in range(7):
    names[2 + 4 * i] = "step %d: column loads + A in LDS" % i
    names[3 + 4 * i] = "step %d: factor" % i
    names[4 + 4 * i] = "step %d: solve + stores" % i
    names[5 + 4 * i] = "step %d: update" % i
prev = t0
for i in sorted(names):
    if st[i] >= t0:
        print("%-40s %8.2f us  (+%.2f)" % (names[i], (st[i] - t0) / 100.0, (st[i] - prev) / 100.0)); prev = st[i]
