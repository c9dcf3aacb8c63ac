"""Per-iteration relative cost changes of the bench loop's final-stage solves (what a convergence predictor would see)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vicalib_amd import synth
from vicalib_amd.lib import ViCalibrator
for name in sys.argv[1:] or ["cfg3"]:
    p = synth.generate_native(synth.BASELINE_CONFIGS[name])
    cal = ViCalibrator(0).load_problem(p)
    cal.Solve()
    tr = cal.trace()
    for st in sorted(set(tr[:, 9])):
        t = tr[tr[:, 9] == st]
        print(name, "stage", int(st), "iters", int(t[:, 0].max()), "rel change:", " ".join("%.0e%s" % (abs(r[2]) / max(r[1], 1e-300), "" if r[8] else "r") for r in t if r[0] > 0))
