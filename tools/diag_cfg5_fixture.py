"""GPU solve of the committed cfg5_rig_160 fixture (tests/golden/lm_traces.json) with the differences to the oracle's record printed per
quantity -- what the tolerances of test_solver_matches_committed_lm_traces[cfg5_rig_160] were set from.  Run from the repo root on a GPU box."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from vicalib_amd import synth
from vicalib_amd.lib import ViCalibrator
e = json.load(open("tests/golden/lm_traces.json"))["cfg5_rig_160"]
cfg = dict(e["config"]); cfg["models"] = tuple(cfg["models"])
p = synth.generate(synth.Config(**cfg))
cal = ViCalibrator(0).load_problem(p); cal.SetCalibrateImu(True); cal.Solve()
tr = cal.trace()[:, [0, 1, 3, 8, 7, 9]]; want = np.array(e["trace"])
print("shapes", tr.shape, want.shape)
k = min(len(tr), len(want))
rc = np.abs(tr[:k, 1] - want[:k, 1]) / want[:k, 1]; rr = np.abs(tr[:k, 4] - want[:k, 4]) / want[:k, 4]
print("cost max rel %.2e at %d; radius max rel %.2e at %d; rows above 1e-6: %s" % (rc.max(), rc.argmax(), rr.max(), rr.argmax(), np.nonzero(rr > 1e-6)[0].tolist()))
print("accept equal", np.array_equal(tr[:k, 3], want[:k, 3]), "stage equal", np.array_equal(tr[:k, 5], want[:k, 5]))
for c, cam in enumerate(e["cameras"]):
    K, T = cal.GetCamera(c)
    print(c, "K rel %.2e" % np.max(np.abs(K - np.array(cam["K"])) / np.abs(np.array(cam["K"]))), "T abs %.2e" % np.max(np.abs(T - np.array(cam["T_ck"]))))
print("rmse rel %.2e" % np.max(np.abs(cal.GetCameraProjRMSE() - np.array(e["rmse"])) / np.array(e["rmse"])))
print("biases abs %.2e" % np.max(np.abs(cal.GetBiases() - np.array(e["imu"]["biases"]))), "toff %.2e" % abs(cal.time_offset() - e["imu"]["time_offset"]))
for i in np.nonzero(rr > 1e-6)[0][:6]:
    print(" row", i, "stage", want[i, 5], "radius", tr[i, 4], want[i, 4], "cost", tr[i, 1], want[i, 1], "acc", tr[i, 3])
