"""k_trsv_lat under GPU sharing: run several copies at once (python tools/lat_stress_stages.py nx ny nz reps tag l|u|lu &), each compares the
lattice solve with the level-scheduled kernel bit for bit and prints the bit patterns of what differs (profiles/r06_lattice_sharing_hazard.txt)"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
import numpy as np
import rocalution_amd as ra
ra.init_rocalution()
nx, ny, nz, reps, tag, stage = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], sys.argv[6]
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "tests"))
from test_gpu_lattice import lattice_csr
rp, ci, va = lattice_csr(nx, ny, nz)
n = len(rp) - 1
def mk():
    M = ra.LocalMatrix(); M.SetDataPtrCSR(rp, ci, va); M.ILU0Factorize(); return M
A, B = mk(), mk()
def ana(M):
    {"lu": M.LUAnalyse, "l": lambda: M.LAnalyse(True), "u": lambda: M.UAnalyse(False)}[stage]()
def sol(M, b, y):
    {"lu": M.LUSolve, "l": M.LSolve, "u": M.USolve}[stage](b, y)
os.environ["RAMD_TRSV_LAT"] = "0"; os.environ["RAMD_TRSV_CT"] = "0"
ana(B)
os.environ["RAMD_TRSV_LAT"] = "2"; del os.environ["RAMD_TRSV_CT"]
ana(A)
rng = np.random.default_rng(1)
y = ra.LocalVector(); y.Allocate("", n); z = ra.LocalVector(); z.Allocate("", n)
bad = 0
for rep in range(reps):
    bh = rng.uniform(-1, 1, n)
    b = ra.LocalVector(data=bh)
    sol(B, b, z); want = z.numpy()
    for k in range(4):
        sol(A, b, y)
        prev = got.copy() if "got" in dir() else None
        got = y.numpy()
        if not np.array_equal(got, want):
            bad += 1
            if bad <= 3:
                d = np.abs(got - want); i = int(np.argmax(d)); idx = np.flatnonzero(d > 0)
                big = np.flatnonzero(d > 0.1 * d.max())
                print(tag, stage, "rep", rep, k, "rows off:", len(idx), "max", float(d.max()), "at", (i % nx, (i // nx) % ny, i // (nx * ny)), flush=True)
                for j in idx[:6]:
                    print("    row", (int(j % nx), int((j // nx) % ny), int(j // (nx * ny))), "got", got[j:j+1].view(np.uint64)[0].item().to_bytes(8, "big").hex(),
                          "want", want[j:j+1].view(np.uint64)[0].item().to_bytes(8, "big").hex(), "prev", (prev[j:j+1].view(np.uint64)[0].item().to_bytes(8, "big").hex() if prev is not None else None),
                          "rhs", bh[j:j+1].view(np.uint64)[0].item().to_bytes(8, "big").hex(), flush=True)
print(tag, stage, (nx, ny, nz), "mismatching solves", bad, "of", reps * 4, flush=True)
