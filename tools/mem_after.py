"""free device memory once per second for a while (does memory of a process that just ended come back late?)"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rocalution_amd as ra
from rocalution_amd import capi
lib = capi.load()
ra.init_rocalution()
f, t = C.c_uint64(0), C.c_uint64(0)
t0 = time.time()
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    capi.check(lib.ramd_mem_info(C.byref(f), C.byref(t)))
    print("t=%5.1f s free %.2f GiB of %.2f" % (time.time() - t0, f.value / 2**30, t.value / 2**30), flush=True)
    time.sleep(1.0)
