cd $GRAFT_REPO_ROOT
VICALIB_AMD_FLAG_SYNC=0 python tests/sync_worker.py /tmp/ev.npz 2>/dev/null
for spec in "30 1" "30 0" "25 0" "22 0"; do set -- $spec
  if [ $2 = 1 ]; then export VICALIB_AMD_BATCHED=1; else unset VICALIB_AMD_BATCHED; fi
  VICALIB_AMD_SYNC_BOUND=1 VICALIB_AMD_SYNC_BOUND_FROM_PASS=$1 python tests/sync_worker.py /tmp/m.npz 2>/tmp/m.err
  python - <<PY
import numpy as np
a=np.load('/tmp/ev.npz'); b=np.load('/tmp/m.npz')
ta,tb=a['trace'],b['trace']
err=open('/tmp/m.err').read()
import re
m=re.findall(r'in LM pass (\d+)', err)
if ta.shape!=tb.shape: print("$1 $2: shapes", ta.shape, tb.shape, "timeout in pass", m)
else:
    d=np.nonzero(np.any(ta!=tb,axis=1))[0]
    print("$1 $2: timeouts", int(b['timeouts']), "reported LM pass", m, "rows", len(ta), "first differing row", d[:3] if len(d) else None)
    np.set_printoptions(linewidth=250, precision=17)
    for r in d[:2]: print(ta[r]); print(tb[r]); print(ta[r]-tb[r])
    if len(d): print("rows before:", ta[d[0]-2:d[0], [0,1,5,6,7,8,9]])
PY
done
