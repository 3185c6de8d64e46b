# round 4, call 32: coupled aggregation AMG on the GlobalMatrix -- P-way == 1-way tests, thin row blocks included
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04ze
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_distributed.py -m gpu -x -q -k "aggregation_amg" > $O/amg.log 2>&1
tail -60 $O/amg.log
cat > /tmp/show.py <<'PY'
import sys, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "tests")); sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from test_cpu_host import _spawn
for kind, worlds in (("gr3030x", (1, 2, 4)), ("thin", (1, 3, 8))):
    for env in (None, {"RAMD_GLOBAL_AMG": "decoupled"}):
        for w in worlds:
            r = _spawn("amg", kind, world=w, timeout=600, env=env)
            print(kind, env, w, "ua it", int(r[0]["res_ua"][0]), "sa it", int(r[0]["res_sa"][0]),
                  "levels ua", r[0]["levels_ua"][:, 0], "sa", r[0]["levels_sa"][:, 0], flush=True)
PY
timeout 900 python /tmp/show.py 2>&1 | grep -v "^$" | tail -20
