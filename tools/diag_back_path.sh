# one-launch back-substitution (k_chain_back_path) against the level-by-level form on the same problem: traces and states
cd "$(dirname "$0")/.."
for n in ${@:-9 17 60 65 513 600}; do
  VICALIB_AMD_BACK_PATH=1 python tests/sync_worker.py /tmp/bp1_$n.npz $n 2>/dev/null
  VICALIB_AMD_BACK_PATH=0 python tests/sync_worker.py /tmp/bp0_$n.npz $n 2>/dev/null
  python - <<PY
import numpy as np
a, b = np.load("/tmp/bp1_$n.npz"), np.load("/tmp/bp0_$n.npz")
ta, tb = a["trace"], b["trace"]
print("frames $n: trace shapes", ta.shape, tb.shape, end=" ")
k = min(len(ta), len(tb))
rel = np.abs(ta[:k, 1] - tb[:k, 1]) / tb[:k, 1]
print("max rel cost diff %.2e (first row above 1e-9: %s)" % (rel.max(), str(int(np.argmax(rel > 1e-9))) if (rel > 1e-9).any() else "-"), "accept equal", np.array_equal(ta[:k, 8], tb[:k, 8]),
      "K rel %.2e" % np.max(np.abs(a["K"] - b["K"]) / np.abs(b["K"])), "frames abs %.2e" % np.max(np.abs(a["frames"] - b["frames"])), "timeouts", int(a["timeouts"]))
print("   first rows rel:", " ".join("%.1e" % x for x in rel[:8]))
PY
done
