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
# round 6: what is well determined -- the projection of every camera over the image (rays on a grid), in pixels per focal length
rays = np.stack(np.meshgrid(np.linspace(-0.6, 0.6, 9), np.linspace(-0.45, 0.45, 7), [1.0]), -1).reshape(-1, 3)
for c, cam in enumerate(e["cameras"]):
    K, T = cal.GetCamera(c); Ko = np.array(cam["K"]); To = np.array(cam["T_ck"])
    pg = synth.project(p.cam_model[c], K, rays); po = synth.project(p.cam_model[c], Ko, rays)
    print(c, "model", p.cam_model[c], "projection max |diff| / f = %.2e" % (np.abs(pg - po).max() / K[0]), "K rel per entry", np.array2string(np.abs(K - Ko) / np.abs(Ko), precision=1),
          "T_ck q abs %.1e t abs %.1e (|t| %.2f)" % (np.abs(T[:4] - To[:4]).max(), np.abs(T[4:] - To[4:]).max(), np.linalg.norm(To[4:])))
print("scale rel %.2e" % np.max(np.abs(cal.GetScaleFactor() - np.array(e["imu"]["scale"])) / np.abs(np.array(e["imu"]["scale"]))),
      "gravity abs %.2e" % np.max(np.abs(cal.GetGravity() - np.array(e["imu"]["gravity"]))), "biases", np.array2string(np.abs(cal.GetBiases() - np.array(e["imu"]["biases"])), precision=1), np.array2string(np.array(e["imu"]["biases"]), precision=3))
inv = np.abs(1.0 / tr[:k, 4] - 1.0 / want[:k, 4])
print("inverse radius: max abs diff %.2e; max (abs diff / (1e-12 + 1e-6 / radius)) = %.3f" % (inv.max(), (inv / (1e-12 + 1e-6 / want[:k, 4])).max()))
