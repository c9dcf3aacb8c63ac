import os, sys; sys.path.insert(0, '/root/repo')
from vicalib_amd import synth
from vicalib_amd.lib import ViCalibrator
p = synth.generate(synth.BASELINE_CONFIGS['cfg2'])
cal = ViCalibrator(0).load_problem(p); cal.SetCalibrateImu(False); cal.prepare()
print(cal.time_stages(20))
st = cal.debug_stamps()
print('phase', [int(st[i+1]-st[i]) for i in range(6)])
print('camera: P %d, T1 %d, Hcc+rmw %d, sync %d' % (st[16]-st[2], st[17]-st[16], st[18]-st[17], st[3]-st[18]))
print('tail: xterms %d, camcopy %d, camplus %d, imu %d, sums %d' % (st[20]-st[5], st[21]-st[20], st[22]-st[21], st[23]-st[22], st[6]-st[23]))
