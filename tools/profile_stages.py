import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vicalib_amd import synth
from vicalib_amd.lib import ViCalibrator
name = sys.argv[1] if len(sys.argv) > 1 else 'cfg2x10'
cfgs = {'cfg2': synth.BASELINE_CONFIGS['cfg2'], 'cfg2x10': synth.Config(models=("fov","fov"), n_frames=5000),
        'poly3x4_large_1000': synth.Config(models=("poly3",)*4, grid="large", n_frames=1000)}
p = synth.generate(cfgs[name])
cal = ViCalibrator(0).load_problem(p); cal.SetCalibrateImu(False); cal.prepare()
print(name, cal.num_observations(), cal.time_stages(5))
