"""Phase timeline of k_final in a steady visual-inertial pass of BASELINE cfg3 (library built with -DVC_FINAL_STAMPS, VICALIB_AMD_LIB pointing at it)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vicalib_amd import synth
from vicalib_amd.lib import ViCalibrator
p = synth.generate_native(synth.BASELINE_CONFIGS["cfg3"])
cal = ViCalibrator(0).load_problem(p)
cal.SetStageLimit(3); cal.Solve(); cal.prepare()
cal.run_iterations(5)
st = cal.debug_stamps().astype(float)
names = ["wavefront starts", "control record read", "count of the second stream's workgroups reached", "step scalars reduced", "before the decision", "decision taken, published",
         "second stream's flag seen", "own flag raised (fence)"]
order = [0, 1, 8, 2, 3, 4, 9, 10, 5, 6, 7]
names = {0: "wavefront starts", 1: "control record read", 8: "main stream's sums reduced", 2: "(old form: count reached)", 3: "count reached, second stream's cost summed", 4: "barrier behind it",
         9: "decision inputs landed", 10: "decision taken", 5: "record stored, progress published", 6: "second stream's flag seen", 7: "own flag raised (fence)"}
prev = st[0]
for i in order:
    if not (st[i] >= st[0]): continue
    print("  %-52s %8.2f us  (+%.2f)" % (names[i], (st[i] - st[0]) / 100.0, (st[i] - prev) / 100.0)); prev = st[i]
raise SystemExit
for i in range(8):
    print("  %-52s %8.2f us  (+%.2f)" % (names[i], (st[i] - st[0]) / 100.0, (st[i] - prev) / 100.0)); prev = st[i]
