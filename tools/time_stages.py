import sys; sys.path.insert(0,'.')
from vicalib_amd import synth
from vicalib_amd.lib import ViCalibrator
for name in ['cfg1','cfg2']:
    p = synth.generate(synth.BASELINE_CONFIGS[name])
    cal = ViCalibrator(0).load_problem(p); cal.SetCalibrateImu(False); cal.prepare()
    print(name, {k: round(v,2) for k,v in cal.time_stages(100).items()}, 'us')
