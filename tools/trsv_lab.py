import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rocalution_amd as ra
ra.init_rocalution()
N = int(sys.argv[1])
A = ra.LocalMatrix(); A.GenPoisson7(N); n = A.GetM()
x = ra.LocalVector(); x.Allocate("", n); x.Ones()
y = ra.LocalVector(); y.Allocate("", n)
A.ILU0Factorize(); A.LUAnalyse(); ra.sync()
for _ in range(3): A.LUSolve(x, y)
ra.sync(); t0 = time.time()
for _ in range(10): A.LUSolve(x, y)
ra.sync(); dt = (time.time() - t0) / 10
print("N=%d lds=%s sleep=%s lusolve %.3f ms  (%.2f us/level)" % (N, os.environ.get("RAMD_TRSV_LDS"), os.environ.get("RAMD_TRSV_SLEEP"), dt * 1e3, dt * 1e6 / (2 * (3 * N - 2))))
