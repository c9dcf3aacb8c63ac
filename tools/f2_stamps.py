"""Phase timeline of sweep 0 of group 0, level 1 of k_chain_fwd2 on BASELINE cfg3, and the phases inside its second elimination
(library built with -DVC_F2_STAMPS, VICALIB_AMD_LIB pointing at it).  100 MHz ticks -> microseconds."""
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
names = {1: "first frame's columns landed", 2: "e1 starts", 3: "e1 eliminated", 4: "e2 starts", 5: "e2 eliminated", 6: "e3 starts", 7: "e3 eliminated",
         8: "middle starts", 9: "middle eliminated", 12: "at the hand-over barrier", 13: "behind it", 14: "separator stored, end"}
prev = t0
for i in sorted(names):
    x = st[i]
    if x >= t0:
        print("  %-32s %8.2f us  (+%.2f)" % (names[i], (x - t0) / 100.0, (x - prev) / 100.0)); prev = x
ph = ["entry", "behind the first synchronisation (A complete)", "Cholesky", "own column solved", "hand-over stores issued", "behind the second synchronisation",
      "image stored, X_n's column taken", "rank-18 update"]
print("inside the second elimination:")
prev = st[16]
for i, n in enumerate(ph):
    x = st[16 + i]
    print("  %-48s %8.2f us  (+%.2f)" % (n, (x - st[16]) / 100.0, (x - prev) / 100.0)); prev = x
