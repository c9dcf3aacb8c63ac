"""Phase timeline of the first wavefront of k_imu_weights on BASELINE cfg3: build the library with -DVC_W_STAMPS into a scratch
name and run with VICALIB_AMD_LIB pointing at it (tools/weights_round.sh does both).  100 MHz s_memrealtime ticks -> microseconds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vicalib_amd import synth
from vicalib_amd.lib import ViCalibrator
p = synth.generate_native(synth.BASELINE_CONFIGS["cfg3"])
cal = ViCalibrator(0).load_problem(p)
cal.SetStageLimit(3); cal.Solve(); cal.prepare()
cal.run_iterations(5)
st = cal.debug_stamps().astype(float)
names = ["entry (control record read)", "parameters, sample range", "measurements of the interval", "1. interval delta (RK4 from identity)",
         "2. scan of the deltas, start quaternion", "3. maps F, Q", "4a. suffix scan of the maps", "4b. conjugation P Q P^T", "4c. butterfly sum",
         "end state, dLog_dSE3, J Sigma J^T row", "Cholesky (shuffles)", "inverse + store"]
prev = st[0]
for i, n in enumerate(names):
    print("%-44s %8.2f us  (+%.2f)" % (n, (st[i] - st[0]) / 100.0, (st[i] - prev) / 100.0)); prev = st[i]
