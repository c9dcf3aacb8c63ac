import os, sys, time
sys.path.insert(0, os.getcwd())
from vicalib_amd import synth
from vicalib_amd.lib import ViCalibrator
p = synth.generate_native(synth.BASELINE_CONFIGS["cfg3"])
for rep in range(2):
    cal = ViCalibrator(0).load_problem(p)
    t = time.time(); cal.Solve(); print("solve", time.time() - t, file=sys.stderr)
