"""Phase timeline of one wavefront of k_chain_init (library built with -DVC_INIT_STAMPS, VICALIB_AMD_LIB pointing at it): the middle
workgroup's first wavefront.  usage: init_stamps.py [cfg3|cfg4|cfg5] [frames].  100 MHz ticks -> microseconds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vicalib_amd import synth
from vicalib_amd.lib import ViCalibrator
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
base = synth.BASELINE_CONFIGS[wl]
n = int(sys.argv[2]) if len(sys.argv) > 2 else base.n_frames
p = synth.generate_native(synth.Config(models=base.models, grid=base.grid, n_frames=n, imu=True, extrinsics_prior=base.extrinsics_prior))
cal = ViCalibrator(0).load_problem(p)
cal.SetStageLimit(3); cal.Solve(); cal.prepare()
cal.run_iterations(5)
st = cal.debug_stamps().astype(float)
names = ["entry", "ctrl + column table + barrier", "frame's requests landed (tile list, IMU records)", "Gram records in LDS + camera sums", "H_pp / g_p", "own block, damping",
         "image columns stored (issued)", "workgroup barrier", "chunk partials written (waited)"]
prev = st[0]
for i in range(9):
    if st[i] >= st[0]:
        print("  %-52s %8.2f us  (+%.2f)" % (names[i], (st[i] - st[0]) / 100.0, (st[i] - prev) / 100.0)); prev = st[i]
