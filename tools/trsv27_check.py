"""The 27-point pencil solve against the general plans (RAMD_TRSV_BOX=0 / 1 in one process), L / U / LU, several solves, bit for bit:
python tools/trsv27_check.py N reps"""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import rocalution_amd as ra
ra.init_rocalution()
N = int(sys.argv[1]); reps = int(sys.argv[2])
nx = ny = nz = N
A = ra.LocalMatrix(); A.GenLaplace27(N, N, N)
A.ILU0Factorize()
n = N ** 3
rng = np.random.default_rng(5)
bh = rng.uniform(-1, 1, n)
b = ra.LocalVector(data=bh)
y = ra.LocalVector(); y.Allocate("", n)
for stage in ("l", "u", "lu"):
    os.environ["RAMD_TRSV_BOX"] = "0"
    if stage == "lu":
        A.LUAnalyse(); A.LUSolve(b, y)
    elif stage == "l":
        A.LAnalyse(True); A.LSolve(b, y)
    else:
        A.UAnalyse(False); A.USolve(b, y)
    want = y.numpy().copy()
    os.environ["RAMD_TRSV_BOX"] = "1"
    if stage == "lu":
        A.LUAnalyse()
    elif stage == "l":
        A.LAnalyse(True)
    else:
        A.UAnalyse(False)
    nbad = 0
    for r in range(reps):
        y.Zeros()
        {"lu": A.LUSolve, "l": A.LSolve, "u": A.USolve}[stage](b, y)
        got = y.numpy()
        bad = np.nonzero(got != want)[0]
        if len(bad):
            nbad += 1
            if nbad <= 3:
                i = bad[0] if stage != "u" else bad[-1]
                print(stage, "rep", r, "rows off", len(bad), "first", (int(i % nx), int((i // nx) % ny), int(i // (nx * ny))), "pencil J,K,j,k",
                      ((int((i // nx) % ny) + int(i // (nx * ny)) % 8) // 8, int(i // (nx * ny)) // 8, (int((i // nx) % ny) + int(i // (nx * ny)) % 8) % 8, int(i // (nx * ny)) % 8),
                      "got", got[i], "want", want[i], flush=True)
    print(stage, N, "mismatching solves", nbad, "of", reps, flush=True)
