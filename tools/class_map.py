"""Map the placement class (csrc/backend.hip) over the device memory: allocate blocks of BLOCK GiB one after the other until
the device is nearly full, print virtual address and class of each.   python tools/class_map.py [block_GiB] [keep_free_GiB]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rocalution_amd as ra
from rocalution_amd import capi
lib = capi.load()
ra.init_rocalution()
blk = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
keep = float(sys.argv[2]) if len(sys.argv) > 2 else 24.0
n = int(blk * (1 << 30) / 8)
lib.ramd_vec_data.restype = C.c_void_p
vecs, line = [], []
total = 0.0
while total + blk + keep < 288.0:
    v = ra.LocalVector()
    try:
        v.Allocate("b", n)
    except Exception as e:
        print("allocation failed at %.0f GiB: %r" % (total, e)); break
    cls = C.c_int(-9)
    capi.check(lib.ramd_vec_placement_class(v._h, C.byref(cls)))
    p = lib.ramd_vec_data(v._h)
    vecs.append(v); total += blk
    line.append(cls.value)
    print("block %3d at %#x (%.0f GiB allocated): class %d" % (len(vecs), p or 0, total, cls.value), flush=True)
print("classes in allocation order:", "".join(str(c) if c >= 0 else "-" for c in line))
