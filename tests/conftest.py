import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): built on demand, single-threaded for determinism."""
    from oracle import oracle as orc
    orc.build()
    orc.set_threads(1)
    return orc


# ---------------------------------------------------------------- forced-form suites in fresh processes, several at a time
# Many settings of the library are read once per process, so the tests that force a kernel form over the parity selection run
# `pytest -k <selection>` in a subprocess each: 21 of them, ~9-20 s apiece and mostly interpreter start-up and tiny kernels.  They
# are independent of each other and of the process that runs this suite, so a family of them is started together (at most
# RAMD_TEST_JOBS at a time, default 5) the first time one of its members is asked for, and every parametrised test only waits
# for its own member.  Nothing is skipped: each member still has to pass in full ("a suite that fits its budget").
import concurrent.futures as _cf
import subprocess as _sp

_POOL = None
_RUNS = {}


def forced_run(family, key, jobs):
    """jobs: {key: (cmd, env, timeout_s)} -- the whole family; returns (returncode, output) of `key`"""
    global _POOL
    if _POOL is None:
        _POOL = _cf.ThreadPoolExecutor(max_workers=max(1, int(os.environ.get("RAMD_TEST_JOBS", "5"))))

    def run(cmd, env, timeout):
        env = dict(env, RAMD_TEST_PRESTART="0")  # (a pass never starts families of its own)
        p = _sp.run(cmd, cwd=ROOT, env=env, stdout=_sp.PIPE, stderr=_sp.STDOUT, text=True, timeout=timeout)
        if p.returncode < 0:
            # the whole pass died by a signal (seen once in round 6: SIGABRT of one of five passes sharing a fresh GPU box, not
            # reproduced in 7 further runs of the family): keep its output for the post-mortem, run the pass again on its own
            try:
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                with open(os.path.join(ROOT, "gpurun_out", "forced_crash_%s_%s.log" % (family, key_of[id(cmd)])), "w") as f:
                    f.write("rc=%d\n" % p.returncode + p.stdout)
            except OSError:
                pass
            first = p
            p = _sp.run(cmd, cwd=ROOT, env=env, stdout=_sp.PIPE, stderr=_sp.STDOUT, text=True, timeout=timeout)
            p.stdout = ("[forced pass %s/%s died with signal %d; this is its second run]\n" % (family, key_of[id(cmd)], -first.returncode)
                        + p.stdout)
        logdir = os.environ.get("RAMD_TEST_LOGDIR")  # (the whole output of every forced pass, for a failure's beginning)
        if logdir:
            os.makedirs(logdir, exist_ok=True)
            with open(os.path.join(logdir, "forced_%s_%s.log" % (family, key_of[id(cmd)])), "w") as f:
                f.write("rc=%d\n" % p.returncode + p.stdout)
        return p.returncode, p.stdout

    key_of = {id(cmd): k for k, (cmd, env, timeout) in jobs.items()}
    if key is None or (family, key) not in _RUNS:
        for k, (cmd, env, timeout) in jobs.items():
            if (family, k) not in _RUNS:
                _RUNS[(family, k)] = _POOL.submit(run, cmd, env, timeout)
    return None if key is None else _RUNS[(family, key)].result()


_FORCED_WAITERS = ("test_parity_suite_with_box_tiles_forced", "with_the_lattice_form_forced", "test_parity_suite_with_the_sync_free",
                   "test_spmv_variants_forced_in_a_fresh_process")


def pytest_collection_modifyitems(config, items):
    """the tests that only WAIT for a forced-form pass go to the end of the run: by then the passes, started at collection time,
    have finished beside the tests of this process (order among themselves and among the others unchanged)"""
    if os.environ.get("RAMD_TEST_PRESTART", "1") == "0":
        return
    waiters = [it for it in items if any(w in it.nodeid for w in _FORCED_WAITERS)]
    if waiters and len(waiters) < len(items):
        rest     = [it for it in items if not any(w in it.nodeid for w in _FORCED_WAITERS)]
        items[:] = rest + waiters


def pytest_collection_finish(session):
    """The forced-form families start as soon as the collection knows they are wanted (a GPU run that selected their tests), not
    when the run reaches their files: they work beside the tests of this process instead of making it wait (round 6: the suite
    had grown to 650 s with the fixtures of the 27-point operator; no test was taken out)."""
    names = [it.nodeid for it in session.items]
    want_tri = any("test_parity_suite_with_box_tiles_forced" in n or "with_the_lattice_form_forced" in n or "syncfree" in n and "forced" in n
                   for n in names)
    want_spmv = any("test_spmv_variants_forced_in_a_fresh_process" in n for n in names)
    if not (want_tri or want_spmv) or os.environ.get("RAMD_TEST_PRESTART", "1") == "0":
        return
    try:
        if not os.path.exists("/dev/kfd"):  # (no GPU here: nothing to start -- and no 9-second import of torch to find out)
            return
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        if want_tri:
            from test_gpu_box_tiles_forced import _tri_family
            forced_run("tri", None, _tri_family())
        if want_spmv:
            from test_gpu_kernels import _spmv_family
            forced_run("spmv", None, _spmv_family())
    except Exception as e:  # (the tests themselves start their family when asked; a failed early start only costs time)
        print("forced-form families not started early: %r" % (e,))
