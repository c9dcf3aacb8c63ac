cd $GRAFT_REPO_ROOT
run() { name=$1; shift; env "$@" python tests/sync_worker.py /tmp/$name.npz 2>/dev/null; }
run ev VICALIB_AMD_FLAG_SYNC=0
run ev2 VICALIB_AMD_FLAG_SYNC=0
run evb VICALIB_AMD_FLAG_SYNC=0 VICALIB_AMD_BATCHED=1
run pl VICALIB_AMD_FLAG_SYNC=0 VICALIB_AMD_BACK_FUSED=0
run plb VICALIB_AMD_FLAG_SYNC=0 VICALIB_AMD_BACK_FUSED=0 VICALIB_AMD_BATCHED=1
run g1 VICALIB_AMD_GRAPHS=1
run g2 VICALIB_AMD_GRAPHS=1
python - <<'PY'
import numpy as np
L={k:np.load('/tmp/%s.npz'%k) for k in ('ev','ev2','evb','pl','plb','g1','g2')}
def cmp(a,b):
    ta,tb=L[a]['trace'],L[b]['trace']
    if ta.shape!=tb.shape: print(a,b,'shape',ta.shape,tb.shape); return
    d=np.argwhere(ta!=tb)
    print(a,'vs',b,': trace entries differing',len(d), 'first', [(int(r),int(c),ta[r,c],tb[r,c]-ta[r,c]) for r,c in d[:6]], 'K equal',np.array_equal(L[a]['K'],L[b]['K']),'frames equal',np.array_equal(L[a]['frames'],L[b]['frames']))
for a,b in (('ev','ev2'),('ev','evb'),('ev','pl'),('pl','plb'),('ev','g1'),('g1','g2'),('plb','g1')): cmp(a,b)
PY
