# round 4, call 35: level fingerprints of the coupled AMG at larger sizes, 1 against 4 ranks
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
cat > /tmp/show.py <<'PY'
import sys, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "tests")); sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from test_cpu_host import _spawn
import numpy as np
np.set_printoptions(linewidth=200, precision=15)
for kind in ("poisson_slab48", "poisson_slab96", "poisson_slab128"):
    for w in (1, 4):
        r = _spawn("amg", kind, world=w, timeout=800)
        for tag in ("ua", "sa"):
            lv = np.array([q["levels_" + tag] for q in r])
            print(kind, w, tag, "it", int(r[0]["res_" + tag][0]), "rows", lv[0, :, 0], "entries", lv[:, :, 1].sum(axis=0), "norm", lv[0, :, 2], flush=True)
PY
timeout 2400 python /tmp/show.py 2>&1 | grep "^poisson" 
