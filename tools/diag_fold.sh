# fold (k_chain_l0) against the two kernels apart on the same problem: traces and states at rounding level (the chunk sums add in a different order)
cd "$(dirname "$0")/.."
for n in 60 100 333; do
  VICALIB_AMD_FOLD_L0=1 python tests/sync_worker.py /tmp/fold1_$n.npz $n 2>/dev/null
  VICALIB_AMD_FOLD_L0=0 python tests/sync_worker.py /tmp/fold0_$n.npz $n 2>/dev/null
  python - <<PY
import numpy as np
a, b = np.load("/tmp/fold1_$n.npz"), np.load("/tmp/fold0_$n.npz")
ta, tb = a["trace"], b["trace"]
print("frames $n: trace shapes", ta.shape, tb.shape, end=" ")
if ta.shape == tb.shape:
    print("max rel cost diff %.2e" % np.max(np.abs(ta[:, 1] - tb[:, 1]) / tb[:, 1]), "accept equal", np.array_equal(ta[:, 8], tb[:, 8]),
          "K rel %.2e" % np.max(np.abs(a["K"] - b["K"]) / np.abs(b["K"])), "frames abs %.2e" % np.max(np.abs(a["frames"] - b["frames"])), "timeouts", int(a["timeouts"]))
else:
    print()
PY
done
