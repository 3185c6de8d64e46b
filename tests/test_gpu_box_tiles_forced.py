"""The record-form box-tile triangular solve (k_trsv_rec, trisolve.hip) on the SMALL parity matrices.

By default a plan takes the box-tile form only for matrices of at least 4096 rows with chains of mean length >= 8, and
external values are de-duplicated only where references outnumber rows -- the goldens (30x30 grids, random patterns) and
most oracle comparisons would exercise only the level-scheduled fallback.  Here the same bit-exact tests run again in a
fresh process with the box-tile form forced for every matrix (RAMD_TRSV_CT_MINROWS / _MINLEN = 0), once with every external
value listed per reference and once de-duplicated (RAMD_TRSV_CT_DEDUP = 0 / 1): ILU(0)/IC factors + LUSolve / LLSolve /
LSolve / USolve vs the goldens (host_matrix_csr.cpp:1163-1221, :1344-1466), LUSolve on Poisson 24^3 / 40^3 vs the oracle,
preconditioner applications, the solver histories vs the goldens, and the af_shell10-class surrogate (8 lanes per row):
~300 tests per run.  The settings are read once per process,
hence the subprocess."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SELECT = ("(ilu or lusolve or trisolve or preconditioner_apply or sgs or solvers_vs_golden or rebuild_numeric or gmres30_ilu0) "
          "and not full_size and not cpp and not fresh_process")


@pytest.mark.gpu
@pytest.mark.parametrize("dedup", ["0", "1"])
def test_parity_suite_with_box_tiles_forced(dedup):
    env = dict(os.environ, RAMD_TRSV_CT_MINROWS="0", RAMD_TRSV_CT_MINLEN="0", RAMD_TRSV_CT_DEDUP=dedup,
               RAMD_TRSV_CT_VERBOSE="1")
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-s", "-m", "gpu", "-p", "no:cacheprovider",
           os.path.join(ROOT, "tests", "test_gpu_kernels.py"), os.path.join(ROOT, "tests", "test_gpu_solvers.py"),
           os.path.join(ROOT, "tests", "test_gpu_shell.py"), "-k", SELECT]
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    tail = p.stdout[-3000:]
    assert p.returncode == 0, tail
    assert " passed" in tail and "failed" not in tail, tail
    # the forced form was really used (the plan builder reports every box-tile plan it makes)
    assert "box-tile plan (lower)" in p.stdout and "box-tile plan (upper)" in p.stdout, tail
    if dedup == "1":
        assert "distinct values" in p.stdout, tail
