cd /root/repo; mkdir -p gpurun_out/shellv
python - <<'PY'
import sys; sys.path.insert(0,'.')
from rocalution_amd import generators as gen
import numpy as np
rp,ci,va = gen.shell_variant(549,'rcm')
np.savez('/tmp/rcm.npz', rp=rp, ci=ci, va=va)
PY
for d in 0 1 2 3; do
RAMD_BAND_DBG=$d python - <<'PY'
import sys, os, time; sys.path.insert(0,'.')
import numpy as np, rocalution_amd as ra
ra.init_rocalution()
z=np.load('/tmp/rcm.npz'); rp,ci,va=z['rp'],z['ci'],z['va']
n=len(rp)-1
A=ra.LocalMatrix(); A.SetDataPtrCSR(rp,ci,va); A.ILU0Factorize(); A.LUAnalyse()
b=ra.LocalVector(data=np.ones(n)); y=ra.LocalVector(); y.Allocate("",n)
for i in range(2): A.LUSolve(b,y)
ra.sync(); t0=time.time()
for i in range(5): A.LUSolve(b,y)
ra.sync(); print("dbg", os.environ["RAMD_BAND_DBG"], "LUSolve %.2f ms"%((time.time()-t0)/5*1e3), flush=True)
PY
done
