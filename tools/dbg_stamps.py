# needs a profiling build: bash vicalib_amd/csrc/build.sh -DVC_REDUCED_STAMPS (and -DVC_JAC_STAMPS for the sweep stamps)
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vicalib_amd import synth
from vicalib_amd.lib import ViCalibrator
p = synth.generate(synth.BASELINE_CONFIGS['cfg2'])
cal = ViCalibrator(0).load_problem(p); cal.SetCalibrateImu(False); cal.prepare()
cal.time_stages(20)
st = cal.debug_stamps()
print([int(st[i+1]-st[i]) for i in range(6)], 'cycles between stamps 0..6 (100MHz const clock?)')
jac_ms, res_ms = cal.time_kernels(5)
st = cal.debug_stamps()
print('jac tile 0 (cycles): entry->body %d, prologue %d, first rows %d, first mfma block %d, remaining passes %d, G store %d' % tuple(int(st[8+i]-st[7+i]) for i in range(6)))
