cd $GRAFT_REPO_ROOT
VICALIB_AMD_FLAG_SYNC=0 python tests/sync_worker.py /tmp/ev.npz
VICALIB_AMD_GRAPHS=1 python tests/sync_worker.py /tmp/gr.npz
VICALIB_AMD_FLAG_SYNC=0 VICALIB_AMD_NO_MERGED_DECISION=1 python tests/sync_worker.py /tmp/ev_nm.npz
python - <<PY
import numpy as np
a=np.load("/tmp/ev.npz")["trace"]; b=np.load("/tmp/gr.npz")["trace"]; c=np.load("/tmp/ev_nm.npz")["trace"]
d=np.argwhere(a!=b)
print("events vs graphs: differing entries (row, col):", d.tolist())
for r,cc in d: print("  row", r, "col", cc, "stage", a[r,9], repr(a[r,cc]), repr(b[r,cc]))
print("events(merged off) vs graphs:", np.argwhere(c!=b).tolist())
PY
