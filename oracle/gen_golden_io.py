#!/usr/bin/env python3
"""Generate tests/golden/io/* from the GENUINE rocALUTION library (TEST INFRASTRUCTURE).

oracle/_ref/ref_probe io: the installed rocALUTION host backend writes a small matrix as MatrixMarket
and as its binary CSR file, a vector as ASCII and binary, and reads three hand-made MatrixMarket
files (symmetric real, symmetric pattern, unsorted general with 1-based indices).  The files it wrote
and the arrays it read are committed as data; tests compare the own IO layer against them.
Run in the dev container:   python oracle/gen_golden_io.py
"""
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from rocalution_amd import generators as gen  # noqa: E402

PROBE = os.path.join(HERE, "_ref", "ref_probe")
OUT = os.path.join(ROOT, "tests", "golden", "io")

SYM = """%%MatrixMarket matrix coordinate real symmetric
% lower triangle, unsorted, 1-based
5 5 8
1 1 4.0
3 1 -1.5
2 2 3.25
5 2 0.125
3 3 1e-3
4 3 -7
4 4 2.5
5 5 6
"""
PAT = """%%MatrixMarket matrix coordinate pattern symmetric
4 4 5
1 1
2 1
3 3
4 2
4 4
"""
GEN = """%%MatrixMarket matrix coordinate real general
% rectangular, entries out of order
3 5 7
3 5 1.5
1 1 -2
2 4 3.5
1 3 0.25
3 1 9
2 2 -0.75
1 5 100
"""


def main():
    if not os.path.exists(PROBE):
        subprocess.check_call(["make", "-C", HERE, "ref"])
    rp, ci, va = gen.laplace2d(5)
    rng = np.random.default_rng(99)
    va = va + rng.uniform(-1e-3, 1e-3, len(va))  # values that need all printed digits
    x = rng.uniform(-4.0, 6.0, size=len(rp) - 1) * 10.0 ** rng.integers(-8, 8, len(rp) - 1)
    os.makedirs(OUT, exist_ok=True)
    with tempfile.TemporaryDirectory() as tin, tempfile.TemporaryDirectory() as tout:
        np.array([len(rp) - 1, len(va), 0, 0], dtype=np.int64).tofile(os.path.join(tin, "hdr.bin"))
        rp.astype(np.int32).tofile(os.path.join(tin, "rowptr.bin"))
        ci.astype(np.int32).tofile(os.path.join(tin, "col.bin"))
        va.astype(np.float64).tofile(os.path.join(tin, "val.bin"))
        x.tofile(os.path.join(tin, "x.bin"))
        for nm, txt in (("sym", SYM), ("pat", PAT), ("gen", GEN)):
            open(os.path.join(tin, nm + ".mtx"), "w").write(txt)
            open(os.path.join(OUT, "in_" + nm + ".mtx"), "w").write(txt)
        subprocess.check_call([PROBE, "io", tin, tout], stdout=subprocess.DEVNULL)
        for f in ("A.mtx", "A.csr", "x.dat", "x.bin"):
            shutil.copy(os.path.join(tout, f), os.path.join(OUT, "ref_" + f))
        d = dict(rowptr=rp.astype(np.int32), col=ci.astype(np.int32), val=va, x=x)
        for nm in ("sym", "pat", "gen"):
            d["read_%s_rowptr" % nm] = np.fromfile(os.path.join(tout, "read_%s_rowptr.bin" % nm), np.int32)
            d["read_%s_col" % nm] = np.fromfile(os.path.join(tout, "read_%s_col.bin" % nm), np.int32)
            d["read_%s_val" % nm] = np.fromfile(os.path.join(tout, "read_%s_val.bin" % nm), np.float64)
            d["read_%s_dims" % nm] = np.fromfile(os.path.join(tout, "read_%s_dims.bin" % nm), np.int64)
        np.savez_compressed(os.path.join(OUT, "io_expected.npz"), **d)
    for f in sorted(os.listdir(OUT)):
        print("%-20s %6d B" % (f, os.path.getsize(os.path.join(OUT, f))))


if __name__ == "__main__":
    main()
