import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vicalib_amd import synth
from vicalib_amd.lib import ViCalibrator
cfgs = {
  'cfg2': synth.BASELINE_CONFIGS['cfg2'],
  'cfg2x10': synth.Config(models=("fov","fov"), n_frames=5000),
  'poly3x4_large_1000': synth.Config(models=("poly3",)*4, grid="large", n_frames=1000),
}
for name in sys.argv[1:] or list(cfgs):
    t=time.time(); p = synth.generate(cfgs[name]); tg=time.time()-t
    cal = ViCalibrator(0).load_problem(p); cal.SetCalibrateImu(False); cal.prepare()
    n = cal.num_observations()
    st = cal.time_stages(30)
    jac_ms, res_ms = cal.time_kernels(20)
    print(name, 'corners', n, 'tiles', cal.num_tiles(), 'gen %.1fs' % tg, {k: round(v,1) for k,v in st.items()}, 'us',
          '| jac %.1f us = %.1f Gcorner/s = %.1f TF/s (1050 flop/corner), %.0f GB/s @18B ; res %.1f us = %.0f GB/s' % (
           jac_ms*1e3, n/jac_ms/1e6, 1050*n/jac_ms/1e9, 18*n/jac_ms/1e6, res_ms*1e3, 18*n/res_ms/1e6))
