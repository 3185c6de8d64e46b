#!/usr/bin/env python3
"""bench.py -- benchmarks of the MI355X-native Krylov core.

    python bench.py --gpus N --steps K --warmup W                      (N > 1: spawns one rank per GPU itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Default workload (BASELINE.json `metric`): CG + Jacobi on the 3-D 7-point Poisson 512^3 operator in CSR, fp64,
rhs = A*1, x0 = 0 (the reference samples' convention, clients/samples/cg.cpp:77-82), synthetic operator generated
on the device.  A "step" is ONE Krylov iteration.  ONE Solve() runs W + K iterations (tolerances that cannot be met):
the W warm-up iterations are untimed, the clock starts when the solver has checked iteration W with the device drained
(ramd_solver_set_time_mark) and stops after barrier + device sync behind the Solve: exactly K iterations, all of their
work included, MAX over ranks.  (--warmup 0 times the whole Solve, i.e. also the initial residual and first direction:
at K = 20 that is 192 instead of 225 it/s.)
N > 1: the rows are split into z-slabs across ranks (strong scaling), halo exchange + scalar all-reduce over RCCL
(GlobalMatrix / GlobalVector path).

Other workloads through the same harness:
    --solver gmres --precond ilu0                   GMRES(30)+ILU(0) on the 512^3 operator (north_star's second target)
    --matrix shell --solver gmres --precond ilu0    BASELINE.json config 3 on the af_shell10-class surrogate
                                                    (generators.shell_surrogate; written as a MatrixMarket symmetric
                                                    file and read back through ReadFileMTX, as af_shell10.mtx would be)
    --solver bicgstab --precond mcsgs --format ell  config 4's solver; --solver mixed: config 5

Objects on the JSON line:
  roofline     -- the DOMINANT kernel of the run: the CSR SpMV for Jacobi-type runs, the sparse triangular solve
                  (`k_trsv`, two launches per ILU(0) apply) for --precond ilu0/ic/sgs.  achieved = algorithmic bytes per
                  launch / average launch duration from HIP events around every such launch inside a live solver run;
                  peak = 8 TB/s (MI355X HBM3E).  `kernels` lists the same figure for the other launch kinds of the iteration.
  cpu_baseline -- the same solver on the host cores: the genuine rocALUTION OpenMP backend through oracle/_ref/ref_probe
                  when the ROCm image ships librocalution ("reference"), else the C oracle ("port"); bounded sample,
                  stated in `sample`.
  N > 1 adds   -- per_rank roofline, halo_ms, halo_overlap_frac, allreduces_per_iter, rccl_nranks.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_COPY_GBPS = 6300.0  # ... and what a device-wide copy achieves there ("~6.3 TB/s"): the practical ceiling of a stream

PROF_SPMV, PROF_TRSV, PROF_HALO, PROF_HALO_WAIT, PROF_ALLREDUCE, PROF_VEC, PROF_PRECOND = range(7)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def spmv_bytes(n, nnz, vbytes=8):
    """clients/samples/benchmark.cpp:213-233: index arrays + values + x + y"""
    return 4 * (n + nnz) + vbytes * (2 * n + nnz)


def trsv_bytes(n, nnz, vbytes=8):
    """average algorithmic bytes of ONE triangle launch of an LU solve on a matrix with a full diagonal and a
    symmetric pattern: the same accounting as the SpMV applied to the strictly-lower part (unit diagonal, forward)
    and to the upper part incl. the diagonal (backward)"""
    nl = (nnz - n) // 2
    lo = 4 * (n + nl) + vbytes * (2 * n + nl)
    up = 4 * (n + nl + n) + vbytes * (2 * n + nl + n)
    return (lo + up) // 2


def mcsgs_bytes(n, nnz, vbytes=8):
    """algorithmic bytes of ONE multi-coloured SGS apply (preconditioner_multicolored_gs.cpp:127-215): every off-diagonal
    entry once (column index + value: the lower part in SolveL_, the upper part in SolveR_) and five vector streams
    (rhs in, the inverse diagonal twice, the diagonal, x out)"""
    return (4 + vbytes) * (nnz - n) + 5 * vbytes * n


MC_FORMS = {0: "k_mc_sweep: all colour sweeps of one apply",
            1: "k_mc_sweep: all colour sweeps of one apply, colour 0's forward sweep folded into its readers",
            2: "k_mc_rb: both colours of a red-black lattice operator in one pass"}


def mc_plan_info(precond=None):
    """what the multi-coloured preconditioner built last in this process looks like (ramd_mcsgs_info): the form of the SGS
    apply is the plan's own answer, not an inference from the command line.  MC-GS / MC-ILU applies always run the sweeps."""
    from rocalution_amd import capi
    st = (C.c_int64 * 8)()
    capi.check(capi.load().ramd_mcsgs_info(None, st))
    form = int(st[0]) if (precond or "mcsgs") == "mcsgs" else min(int(st[0]), 0)
    return dict(form=form, colours=int(st[1]), row_patterns=[int(st[2]), int(st[3])],
                lattice=[int(st[4]), int(st[5]), int(st[6])] if form == 2 else None)


def mc_form(precond=None):
    return MC_FORMS[mc_plan_info(precond)["form"]]


def mc_traffic_key(precond=None):
    return "mcsgs_512" if mc_plan_info(precond)["form"] == 2 else "mcsgs_512_sweeps"


def physical_cores():
    ncpu = os.cpu_count() or 1
    try:
        phys = int(subprocess.check_output("lscpu -p=CORE,SOCKET | grep -v '^#' | sort -u | wc -l",
                                           shell=True).decode().strip())
    except Exception:
        phys = ncpu
    return max(1, min(phys, ncpu))


SOLVER_LABEL = {"cg": "CG", "gmres": "GMRES(30)", "bicgstab": "BiCGStab", "mixed": "MixedPrecisionDC(fp64/fp32 CG)"}
PRECOND_LABEL = {"none": "none", "jacobi": "Jacobi", "ilu0": "ILU(0)", "mcsgs": "MC-SGS", "mcgs": "MC-GS",
                 "mcilu": "MC-ILU(0,1)", "ic": "IC", "sgs": "SGS", "uaamg": "UAAMG(PMIS)", "saamg": "SAAMG(PMIS)",
                 # the distributed driver only: AMG on the GlobalMatrix itself (coarse levels coupled across the ranks)
                 "global-uaamg": "UAAMG(PMIS) on the GlobalMatrix", "global-saamg": "SAAMG(PMIS) on the GlobalMatrix"}


def cpu_baseline(args, mtx_path=None):
    """the same solver / preconditioner on the host: reference OpenMP backend if available, else the C oracle (port)"""
    threads = physical_cores()
    iters = args.cpu_iters
    probe = os.path.join(ROOT, "oracle", "_ref", "ref_probe")
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), OMP_PROC_BIND="close", OMP_PLACES="cores")
    label = "%s+%s" % (SOLVER_LABEL.get(args.solver, args.solver), PRECOND_LABEL[args.precond])
    ref_solver = {"cg": "cg", "gmres": "gmres", "bicgstab": "bicgstab"}.get(args.solver)
    ref_precond = args.precond if args.precond in ("jacobi", "ilu0", "mcsgs", "none") else None
    if os.path.exists(probe) and os.path.exists("/opt/rocm/lib/librocalution.so") and ref_solver and ref_precond:
        try:
            src = mtx_path if mtx_path else (("lap27:%d" % args.cpu_grid) if args.matrix == "lap27" else str(args.cpu_grid))
            out = subprocess.check_output([probe, "bench", src, str(iters), str(threads), "0", ref_solver, ref_precond],
                                          env=env, stderr=subprocess.DEVNULL, timeout=1500).decode()
            rec = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
            if mtx_path:
                what = "the full %d-row %s (read by the reference's ReadFileMTX in %.1f s)" % (
                    rec["n"], "surrogate file" if args.matrix == "shell" else "file " + os.path.basename(mtx_path), rec["t_read_s"])
                scale = 1.0
            else:
                scale = (args.cpu_grid / float(args.grid)) ** 3
                what = "3-D %s %d^3 CSR fp64 (%.3g of the benchmark's rows)" % ("27-point Laplacian" if args.matrix == "lap27" else "Poisson",
                                                                                  args.cpu_grid, scale)
            r = dict(value=rec["iters_per_s"], unit="iters/s", cores=threads, kind="reference",
                     sample="%s, %s, %d iterations, rocALUTION (ROCm-installed) host/OpenMP backend, accelerator disabled"
                            % (label, what, rec["iters"]),
                     spmv_GBps=rec["spmv_GBps"], build_s=rec["t_build_s"])
            if scale != 1.0:
                r["equivalent_full_size_iters_per_s"] = rec["iters_per_s"] * scale
            return r
        except Exception as e:  # fall through to the port
            log("cpu_baseline: reference probe failed (%r), using the oracle port" % (e,))
    from oracle import oracle as orc
    from rocalution_amd import generators as gen
    orc.build()
    orc.set_threads(threads)
    if args.matrix == "file":
        import scipy.io
        import scipy.sparse as sp
        M = sp.csr_matrix(scipy.io.mmread(mtx_path))
        M.sort_indices()
        rp, ci, va = M.indptr.astype(np.int32), M.indices.astype(np.int32), M.data.astype(np.float64)
        what = "file " + os.path.basename(mtx_path)
    elif args.matrix == "shell":
        rp, ci, va = gen.shell_surrogate(min(args.shell_nx, 200))
        what = "shell surrogate %d^2 nodes" % min(args.shell_nx, 200)
    elif args.matrix == "lap27":
        Np = min(args.cpu_grid, 96)
        rp, ci, va = gen.laplace27(Np)
        what = "3-D 27-point Laplacian %d^3 CSR fp64" % Np
    else:
        Np = min(args.cpu_grid, 128)
        rp, ci, va = gen.poisson7(Np)
        what = "3-D Poisson %d^3 CSR fp64" % Np
    rhs = orc.csr_apply(rp, ci, va, np.ones(len(rp) - 1))
    osolver = {"cg": orc.CG, "gmres": orc.GMRES, "bicgstab": orc.BICGSTAB}.get(args.solver, orc.CG)
    opc = {"none": orc.PC_NONE, "jacobi": orc.PC_JACOBI, "ilu0": orc.PC_ILU0, "mcsgs": orc.PC_MCSGS}.get(args.precond, orc.PC_JACOBI)
    t0 = time.time()
    r = orc.solve(rp, ci, va, rhs, solver=osolver, precond=opc, abs_tol=0.0, rel_tol=0.0, div_tol=1e300,
                  max_iter=iters, history=False)
    dt = time.time() - t0
    return dict(value=r["iters"] / dt, unit="iters/s", cores=threads, kind="port",
                sample="%s, %s, %d iterations (incl. Build), C oracle (OpenMP)" % (label, what, r["iters"]))


def cpu_port_baseline(rp, ci, va, solver, precond, iters, what):
    """the C restatement of the reference's host path (oracle/, OpenMP where the reference has it) on arrays in memory: the CPU
    baseline of the extras whose operator exists only in memory (kind "port")"""
    from oracle import oracle as orc
    orc.build()
    threads = physical_cores()
    orc.set_threads(threads)
    rhs = orc.csr_apply(rp, ci, va, np.ones(len(rp) - 1))
    osolver = {"cg": orc.CG, "gmres": orc.GMRES, "bicgstab": orc.BICGSTAB}[solver]
    opc = {"none": orc.PC_NONE, "jacobi": orc.PC_JACOBI, "ilu0": orc.PC_ILU0, "mcsgs": orc.PC_MCSGS}[precond]
    t0 = time.time()
    r = orc.solve(rp, ci, va, rhs, solver=osolver, precond=opc, abs_tol=0.0, rel_tol=0.0, div_tol=1e300, max_iter=iters, history=False)
    dt = time.time() - t0
    return dict(value=round(r["iters"] / dt, 3), unit="iters/s", cores=threads, kind="port",
                sample="%s+%s, %s, %d iterations (incl. Build), C oracle (the reference's host loops restated; OpenMP where the "
                       "reference has it -- its triangular solves are sequential)" % (SOLVER_LABEL[solver], PRECOND_LABEL[precond], what, r["iters"]))


def reference_gpu(args):
    """optional vendor column: the reference's own rocSPARSE/rocBLAS HIP backend on this GPU (Poisson only)"""
    probe = os.path.join(ROOT, "oracle", "_ref", "ref_probe")
    if not (os.path.exists(probe) and os.path.exists("/opt/rocm/lib/librocalution_hip.so")):
        return None
    if args.matrix not in ("poisson", "lap27") or args.solver not in ("cg", "gmres", "bicgstab") or args.precond not in ("jacobi", "ilu0", "mcsgs"):
        return None
    def probe_run(grid, steps):
        out = subprocess.check_output([probe, "bench", ("lap27:%d" % grid) if args.matrix == "lap27" else str(grid), str(steps), "0", "1",
                                       args.solver, args.precond],
                                      stderr=subprocess.STDOUT, timeout=900).decode()
        rec = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
        return dict(iters_per_s=rec["iters_per_s"], spmv_GBps=rec["spmv_GBps"], t_spmv_ms=rec["t_spmv_s"] * 1e3,
                    what="rocALUTION (installed) HIP backend = rocSPARSE/rocBLAS wrapper, same workload")
    steps = min(args.steps, 60) if args.precond == "ilu0" else args.steps
    try:
        return probe_run(args.grid, steps)
    except subprocess.CalledProcessError as e:
        # (the installed library's ILU(0) on the HIP backend fails at 512^3 -- hip_matrix_csr.cpp:1352, its rocSPARSE csrilu0
        #  call: report that, and what it does at the largest size it runs)
        msg = [l for l in (e.output or b"").decode(errors="replace").splitlines() if l.strip()]
        log("reference_gpu: %s+%s at grid %d failed in the vendor library (%s)" % (args.solver, args.precond, args.grid, msg[-1] if msg else e))
        rg = dict(error="the vendor backend fails on this workload at grid %d: %s" % (args.grid, msg[-1][-160:] if msg else repr(e)))
        if args.matrix in ("poisson", "lap27") and args.grid > 256:
            try:
                small = probe_run(256, min(steps, 10))  # (its csrsv takes ~1.3 s per iteration there)
                small["grid"] = 256
                small["what"] += " at 256^3 (the largest grid of the series it completes)"
                rg["at_256"] = small
            except Exception as e2:
                log("reference_gpu: also at 256^3 (%r)" % (e2,))
        return rg
    except Exception as e:
        log("reference_gpu: failed (%r)" % (e,))
        return None


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_spawn(args):
    """`python bench.py --gpus N` without a launcher: one process per GPU (LOCAL_RANK = device), this process waits.
    Never falls back to fewer ranks: fewer than N visible devices is an error."""
    try:
        ndev = int(subprocess.check_output([sys.executable, "-c",
                                            "import sys; sys.path.insert(0, %r); import rocalution_amd as ra; "
                                            "print(ra.device_count())" % ROOT], stderr=subprocess.DEVNULL).decode().split()[-1])
    except Exception:
        ndev = 0
    if args.transport == "callback":
        if ndev < 1:
            log("bench.py: --transport callback: no device visible")
            sys.exit(2)
    elif ndev < args.gpus:
        log("bench.py: --gpus %d: need %d devices, %d visible -- not running on fewer ranks" % (args.gpus, args.gpus, ndev))
        sys.exit(2)
    port = free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in pending:  # one rank died: the others would hang in a collective
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    sys.exit(rc)


def prof_get(lib, capi, ch):
    cnt, avg, mn, mx = C.c_int(0), C.c_double(0), C.c_double(0), C.c_double(0)
    capi.check(lib.ramd_prof_result(ch, C.byref(cnt), C.byref(avg), C.byref(mn), C.byref(mx)))
    tot = C.c_int64(0)
    capi.check(lib.ramd_prof_count(ch, C.byref(tot)))
    return dict(launches=cnt.value, avg_ms=avg.value, min_ms=mn.value, max_ms=mx.value, count=tot.value)


def roof(name, bytes_alg, p, traffic=None):
    """frac = ALGORITHMIC bytes (the reference's accounting of the operation) / time / peak.  A kernel that rebuilds its column
    indices from a dictionary, or a triangular solve that needs none, moves fewer bytes than that: `moved_frac` is what the
    memory system really delivered (PMC traffic of the same launch, measured offline: profiles/) over the same time and peak,
    `achievable_frac` prices the algorithmic rate against the guide's copy figure instead of the 8 TB/s of the data sheet"""
    ach = bytes_alg / (p["avg_ms"] * 1e-3) / 1e9 if p["avg_ms"] > 0 else 0.0
    r = dict(bound="hbm", achieved=round(ach, 1), peak=HBM_PEAK_GBPS, unit="GB/s", frac=round(ach / HBM_PEAK_GBPS, 4),
             traffic=traffic, kernel=name, launches=p["launches"], avg_ms=round(p["avg_ms"], 5),
             min_ms=round(p["min_ms"], 5), max_ms=round(p["max_ms"], 5), algorithmic_bytes=int(bytes_alg),
             achievable_frac=round(ach / HBM_COPY_GBPS, 4))
    if traffic and p["avg_ms"] > 0:
        moved = traffic / (p["avg_ms"] * 1e-3) / 1e9
        r["moved_GBps"] = round(moved, 1)
        r["moved_frac"] = round(moved / HBM_PEAK_GBPS, 4)
        r["traffic_over_algorithmic"] = round(traffic / bytes_alg, 4)
    return r


TRI_FORMS = {0: "none", 1: "level-scheduled rows (k_trsv)", 2: "box tiles, record form (k_trsv_rec)",
             3: "box tiles with row groups (k_trsv_rec, grouped)", 4: "lattice pencils (k_trsv_lat)",
             6: "row groups handed from wave to wave (k_trsv_sf: a deep, narrow dependency graph of long rows; tiles = units of whole "
                "row groups of one group level, steps = row groups)",
             7: "sheared pencils of the 27-point stencil (k_trsv_box)"}


def tri_plan_stats(lib, capi):
    """what the last LUAnalyse / LAnalyse / UAnalyse of this process built (ramd_tri_plan_stats): so that a run on a matrix
    nobody here has seen (--mtx af_shell10.mtx) comes back with a diagnosis of its triangular solves, not just a rate"""
    out = {}
    for which, name in ((0, "lower"), (1, "upper")):
        st = (C.c_int64 * 16)()
        capi.check(lib.ramd_tri_plan_stats(which, st))
        if st[0] == 0:
            continue
        d = dict(form=TRI_FORMS.get(int(st[0]), str(st[0])), rows=int(st[1]), dependency_levels=int(st[2]),
                 tiles=int(st[3]), steps=int(st[4]), values_handed_between_tiles=int(st[5]), max_rows_per_tile=int(st[6]),
                 longest_row=int(st[7]), lanes_per_row=int(st[8]))
        if st[0] in (4, 7):
            d["lattice"] = [int(st[9]), int(st[10]), int(st[11])]
            d["plan_bytes"] = int(st[12])
        elif st[0] in (1, 6):
            d["why_not_box_tiles"] = {0: "", 1: "no chains of consecutively numbered dependent rows (mean chain length < 8): the tile coordinates of "
                                         "the box-tile form are built on such chains", 2: "no dependencies", 3: "triangular rows longer than 32 "
                                         "entries that do not form row groups", 4: "tile key range", 5: "entry index range",
                                      6: "the tiles cannot be made to fit the LDS", 7: "too few rows", 8: "switched off",
                                      9: "row groups with more than 24 entries outside the group (the sync-free grouped form takes these)"}.get(int(st[12]), str(st[12]))
        elif st[0] in (2, 3):
            d["box"] = [int(st[9]), int(st[10]), int(st[11])]
            d["chains"] = int(st[13])
            d["max_steps_per_tile"] = int(st[14])
            d["max_external_values_per_tile"] = int(st[15])
        out[name] = d
    return out


def traffic_for(key):
    """HBM bytes per launch from the PMC counters: measured offline with rocprofv3 in separate --pmc passes
    (tools/pmc_passes.sh) and committed under profiles/; bench.py does not run the profiler"""
    for f in ("r06_traffic.json", "r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
        p = os.path.join(ROOT, "profiles", f)
        if os.path.exists(p):
            d = json.load(open(p))
            if key in d:
                return d[key]
            if key == "spmv_csr_512" and "traffic_bytes" in d:
                return d["traffic_bytes"]
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--matrix", default="poisson", choices=["poisson", "lap27", "shell", "file"],
                    help="poisson: 3-D 7-point operator (device generator); lap27: the reference's own 3-D operator, the 27-point "
                         "Laplacian of clients/include/utility.hpp:110-177 (device generator, --grid N); shell: af_shell10-class surrogate "
                         "(BASELINE.json config 3; 1 GPU), read through ReadFileMTX; file: the MatrixMarket file given with --mtx")
    ap.add_argument("--mtx", default=None, metavar="PATH",
                    help="MatrixMarket file supplied on the box (e.g. SuiteSparse af_shell10.mtx for config 3): read with the "
                         "reference's ReadFileMTX semantics (host_io.cpp:135-276); implies --matrix file")
    ap.add_argument("--grid", type=int, default=512, help="Poisson grid edge N (operator is N^3 x N^3)")
    ap.add_argument("--shell-nx", type=int, default=549, help="shell surrogate: nx x nx mesh nodes, 5 unknowns each")
    ap.add_argument("--shell-variant", default="lex", choices=["lex", "rcm", "delaunay", "random", "morton"],
                    help="node numbering of the config-3-class operator (generators.shell_variant): lex = the surrogate; rcm = the "
                         "same mesh in reverse Cuthill-McKee order; delaunay = a jittered-point triangulation in RCM order; "
                         "random = a random node permutation")
    ap.add_argument("--format", default="csr", choices=["csr", "ell", "hyb"])
    ap.add_argument("--cpu-grid", type=int, default=None, help="Poisson grid of the CPU baseline sample (default: --grid)")
    ap.add_argument("--cpu-iters", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reference-gpu", action="store_true")
    ap.add_argument("--solver", default="cg", choices=["cg", "gmres", "bicgstab", "mixed"],
                    help="headline = cg; the others run the remaining BASELINE.json configs through the same harness")
    ap.add_argument("--precond", default="jacobi", choices=sorted(PRECOND_LABEL))
    ap.add_argument("--itsolve", type=int, default=0,
                    help="ILU / IC / SGS: iterative triangular solves (TriSolverAlg_Iterative) with this many sweeps")
    ap.add_argument("--force-global", action="store_true",
                    help="1 process: still go through the GlobalMatrix/RCCL code path (communicator of size 1, "
                         "collectives not skipped) - a check of the N>1 plumbing on a 1-GPU box")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "callback"],
                    help="data plane of the N > 1 run.  rccl (default): one GPU per rank, halo exchange and scalar all-reduce over "
                         "RCCL/xGMI.  callback: the ranks SHARE the visible device(s) (rank r on device r mod count) and talk "
                         "through the host-staged callback transport of the tests (gloo underneath) -- a REHEARSAL of the whole "
                         "N > 1 code path of this file on a 1-GPU box (RCCL refuses two ranks on one device); the line it prints "
                         "says so (\"transport\", \"rehearsal\": true) and is not a scaling measurement")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the GMRES(30)+ILU(0) and BiCGStab+MC-SGS legs (reported under `extras`)")
    args = ap.parse_args()
    if args.mtx:
        args.matrix = "file"
    if args.matrix == "file" and not (args.mtx and os.path.exists(args.mtx)):
        raise SystemExit("--matrix file needs --mtx PATH of an existing MatrixMarket file")
    if args.cpu_grid is None:
        args.cpu_grid = args.grid
    if args.cpu_iters is None:  # ~10-30 s of host work at the default sizes
        args.cpu_iters = 12 if (args.matrix == "poisson" and args.cpu_grid >= 384) else 30

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_spawn(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world != args.gpus:
        log("bench.py: --gpus %d but the launcher started %d rank(s): refusing to report a mislabelled run" % (args.gpus, world))
        sys.exit(2)
    if args.precond.startswith("global-"):
        args.force_global = True  # (the preconditioner of the GlobalMatrix path)
    if args.matrix in ("shell", "file") and (world > 1 or args.force_global):
        raise SystemExit("--matrix shell / --mtx is the 1-GPU workload of config 3 (LocalMatrix path)")

    if args.force_global:
        os.environ["RAMD_COMM_FORCE_COLLECTIVES"] = "1"
    import rocalution_amd as ra
    from rocalution_amd import capi
    lib = capi.load()
    rehearsal = args.transport == "callback" and world > 1
    if world > 1 and not rehearsal and ra.device_count() < world:
        log("bench.py: %d ranks need %d devices, %d visible" % (world, world, ra.device_count()))
        sys.exit(2)
    if rehearsal and ra.device_count() < 1:
        log("bench.py: --transport callback: no device visible")
        sys.exit(2)
    ra.init_rocalution(local_rank % ra.device_count() if rehearsal else local_rank)
    if rank == 0:
        log(ra.info_rocalution())

    dist = None
    comm = C.c_void_p()
    if world > 1 or args.force_global:
        # torch.distributed (gloo) is the control plane only: it ships the 128-byte RCCL id and the timing
        # reduction.  --force-global walks the very same path with one rank (plumbing check on a 1-GPU box).
        import torch
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        if rehearsal:
            from rocalution_amd import distributed as D_
            comm = D_.make_callback_comm(rank, world, dist)  # data plane: host-staged (several ranks per device)
        else:
            uid = C.create_string_buffer(128)
            if rank == 0:
                capi.check(lib.ramd_comm_unique_id(uid))
            t = torch.tensor(list(uid.raw), dtype=torch.uint8)
            dist.broadcast(t, src=0)
            uid = C.create_string_buffer(bytes(t.tolist()), 128)
            capi.check(lib.ramd_comm_init_rccl(rank, world, uid, C.byref(comm)))  # data plane: RCCL over xGMI

    if comm:
        C.CDLL(None).fflush(None)  # RCCL's version banner sits in C stdio: get it out before the JSON line

    def barrier():
        ra.sync()
        if dist is not None:
            dist.barrier()

    N = args.grid
    fmt = {"csr": ra.CSR, "ell": ra.ELL, "hyb": ra.HYB}[args.format]
    K, W = args.steps, args.warmup
    NEVER = (0.0, 0.0, 1e300)  # abs / rel / div tolerances that cannot trigger: exactly max_iter steps
    mixed = args.solver == "mixed"
    tri_pc = args.precond in ("ilu0", "ic", "sgs") and args.itsolve == 0
    label = SOLVER_LABEL[args.solver] + "+" + PRECOND_LABEL[args.precond]

    prof = None
    tri_plan = None
    kernels = {}
    extras = {}
    ingest = None
    mtx_path = None
    mtx_generated = False
    cols_read = None
    scaling_fields = {}
    if world == 1 and not args.force_global:
        from rocalution_amd import solvers as S
        A = ra.LocalMatrix()
        if args.matrix == "shell":
            from rocalution_amd import generators as gen
            t0 = time.perf_counter()
            rp_h, ci_h, va_h = gen.shell_variant(args.shell_nx, args.shell_variant)
            t_gen = time.perf_counter() - t0
            mtx_path = os.path.join(os.environ.get("TMPDIR", "/tmp"), "ramd_shell_%d_%s.mtx" % (args.shell_nx, args.shell_variant))
            mtx_generated = True
            t0 = time.perf_counter()
            stored = gen.write_mtx_symmetric(mtx_path, rp_h, ci_h, va_h)
            t_write = time.perf_counter() - t0
            n, nnz = len(rp_h) - 1, len(ci_h)
            del rp_h, ci_h, va_h
            t0 = time.perf_counter()
            A.ReadFileMTX(mtx_path)  # host_io.cpp:135-276 semantics: symmetric expansion + row sort, then upload
            ra.sync()
            ingest = dict(file_bytes=os.path.getsize(mtx_path), stored_entries=stored, read_s=round(time.perf_counter() - t0, 3),
                          generate_s=round(t_gen, 3), write_s=round(t_write, 3))
            assert (A.GetM(), A.GetNnz()) == (n, nnz)
            regen = lambda: A.ReadFileMTX(mtx_path)
            wl = ("af_shell10-class surrogate (SuiteSparse af_shell10 itself is not available offline): %d x %d mesh nodes x 5 "
                  "unknowns, node numbering '%s', n=%d, nnz=%d (%.2f per row; af_shell10: n=1508065, nnz=52259885), SPD, read from a "
                  "MatrixMarket symmetric file" % (args.shell_nx, args.shell_nx, args.shell_variant, n, nnz, nnz / n))
        elif args.matrix == "file":
            mtx_path = os.path.abspath(args.mtx)
            t0 = time.perf_counter()
            A.ReadFileMTX(mtx_path)
            ra.sync()
            n, nnz = A.GetM(), A.GetNnz()
            if A.GetN() != n:
                raise SystemExit("--mtx: the Krylov drivers need a square matrix (%d x %d)" % (n, A.GetN()))
            ingest = dict(file_bytes=os.path.getsize(mtx_path), read_s=round(time.perf_counter() - t0, 3))
            regen = lambda: A.ReadFileMTX(mtx_path)
            wl = "%s (n=%d, nnz=%d, %.2f per row), read through ReadFileMTX" % (os.path.basename(mtx_path), n, nnz, nnz / max(n, 1))
        elif args.matrix == "lap27":
            A.GenLaplace27(N)
            n, nnz = A.GetM(), A.GetNnz()
            assert n == N ** 3 and nnz == (3 * N - 2) ** 3
            regen = lambda: A.GenLaplace27(N)
            wl = ("3-D 27-point Laplacian %d^3 (the reference's gen_3d_laplacian, clients/include/utility.hpp:110-177; n=%d, nnz=%d)"
                  % (N, n, nnz))
        else:
            n, nnz = N ** 3, 7 * N ** 3 - 6 * N ** 2
            A.GenPoisson7(N)
            regen = lambda: A.GenPoisson7(N)
            wl = "3-D 7-point Poisson %d^3 (n=%d, nnz=%d)" % (N, n, nnz)
        ones = ra.LocalVector(); ones.Allocate("ones", n); ones.Ones()
        rhs = ra.LocalVector(); rhs.Allocate("rhs", n)
        x = ra.LocalVector(); x.Allocate("x", n)
        A.Apply(ones, rhs)

        CH = (PROF_SPMV, PROF_TRSV, PROF_VEC, PROF_PRECOND)
        SYSTEM = (A, rhs, x, regen, fmt)  # the operator of the headline run (the extras bring their own)

        def run(iters, solver_cls=S.CG, pc_cls=S.Jacobi, basis=None, warm=0, prof_iters=0, system=None):
            """one Solve of warm + iters iterations; warm > 0: the clock starts when iteration `warm` has been checked
            and the device drained (ramd_solver_set_time_mark), so the timed region is exactly `iters` iterations and
            the solver's preamble (initial residual, first direction) falls into the warm-up.
            prof_iters > 0: afterwards a SECOND Solve of that many iterations on the SAME built solver (same work vectors,
            same placement) with every SpMV / triangular-solve / fused-vector / preconditioner launch bracketed by HIP
            events on its stream -> the fifth value returned (dict channel -> statistics)"""
            A, rhs, x, regen, fmt = system if system is not None else SYSTEM
            if A.GetFormat() != ra.CSR:  # preconditioners are built from the CSR state
                regen()
            ls = solver_cls()
            ls.SetOperator(A)
            if pc_cls is not None:
                pc = pc_cls()
                if args.itsolve > 0 and pc_cls in (S.ILU, S.IC, S.SGS):  # TriSolverAlg_Iterative, fixed sweep count
                    d = S.SolverDescr(); d.SetTriSolverAlg(S.TriSolverAlg_Iterative)
                    d.SetIterativeSolverMaxIteration(args.itsolve); d.DisableIterativeSolverTolerance()
                    pc.SetSolverDescriptor(d)
                ls.SetPreconditioner(pc)
            if basis:
                ls.SetBasisSize(basis)
            ls.Init(NEVER[0], NEVER[1], NEVER[2], warm + iters)
            tb = time.perf_counter()
            ls.Build()
            ra.sync()
            tb = time.perf_counter() - tb
            if fmt != ra.CSR and A.GetFormat() == ra.CSR:
                A.ConvertTo(fmt)
            if warm > 0:
                ls.SetTimeMark(warm)
            x.Zeros()
            barrier()
            t0 = time.perf_counter()
            ls.Solve(rhs, x)
            barrier()
            dt = ls.GetSecondsSinceTimeMark() if warm > 0 else time.perf_counter() - t0
            assert dt > 0, "the marked iteration was not reached"
            it = ls.GetIterationCount() - warm
            res = ls.GetCurrentResidual()
            pr = None
            if prof_iters > 0:
                ls.Init(NEVER[0], NEVER[1], NEVER[2], prof_iters)
                ls.SetTimeMark(-1)
                x.Zeros()
                for ch in CH:
                    capi.check(lib.ramd_prof_enable(ch, 1))
                ls.Solve(rhs, x)
                ra.sync()
                pr = {ch: prof_get(lib, capi, ch) for ch in CH}
                for ch in CH:
                    capi.check(lib.ramd_prof_enable(ch, 0))
            ls.Clear()
            return dt, it, res, tb, pr

        HEAD = {"cg": S.CG, "gmres": S.GMRES, "bicgstab": S.BiCGStab}.get(args.solver, S.CG)
        HPC = {"none": None, "jacobi": S.Jacobi, "ilu0": S.ILU, "mcsgs": S.MultiColoredSGS,
               "mcgs": S.MultiColoredGS, "mcilu": S.MultiColoredILU, "ic": S.IC, "sgs": S.SGS, "uaamg": S.UAAMG, "saamg": S.SAAMG}[args.precond]
        basis = 30 if args.solver == "gmres" else None
        def run_mixed(iters, *_a, warm=0, prof_iters=0, pc_cls=None):
            """config 5 on one GPU: MixedPrecisionDC, fp64 outer defect correction around an fp32 CG
            (clients/samples/mixed-precision.cpp:85: inner tolerances 1e-5 / 1e-2 / 1e+20, 100000 iterations)"""
            if True:
                inner = S.CG(np.float32)
                if pc_cls is not None:
                    inner.SetPreconditioner(pc_cls())
                inner.Init(1e-5, 1e-2, 1e20, 100000)
                mp = S.MixedPrecisionDC(); mp.SetOperator(A); mp.Set(inner)
                mp.Init(NEVER[0], NEVER[1], NEVER[2], warm + iters)
                mp.Build()
                if warm > 0:
                    mp.SetTimeMark(warm)
                x.Zeros(); barrier()
                t0 = time.perf_counter(); mp.Solve(rhs, x); barrier()
                dt = mp.GetSecondsSinceTimeMark() if warm > 0 else time.perf_counter() - t0
                assert dt > 0, "the marked iteration was not reached"
                r = (dt, mp.GetIterationCount() - warm, mp.GetCurrentResidual(), 0.0)
                pr = None
                if prof_iters > 0:
                    mp.Init(NEVER[0], NEVER[1], NEVER[2], prof_iters)
                    mp.SetTimeMark(-1)
                    x.Zeros()
                    for ch in CH:
                        capi.check(lib.ramd_prof_enable(ch, 1))
                    mp.Solve(rhs, x)
                    ra.sync()
                    pr = {ch: prof_get(lib, capi, ch) for ch in CH}
                    for ch in CH:
                        capi.check(lib.ramd_prof_enable(ch, 0))
                mp.Clear()
                return r + (pr,)
        if mixed:
            def run(iters, *_a, warm=0, prof_iters=0):  # noqa: F811
                return run_mixed(iters, warm=warm, prof_iters=prof_iters, pc_cls=HPC)
        # W untimed warm-up iterations, then EXACTLY K timed ones, in one Solve (see run); --warmup 0 times the whole Solve
        # --- timed leg, then the roofline leg: the same built solver solves again with every SpMV / triangular-solve /
        # fused-vector launch bracketed by HIP events on the stream it runs on
        capi.check(lib.ramd_placement_seconds(None, 1))
        dt, it, res, tbuild, pr0 = run(K, HEAD, HPC, basis, warm=W, prof_iters=(min(K, 20) if mixed else min(K, 200)))
        placement_s = C.c_double(0.0)
        capi.check(lib.ramd_placement_seconds(C.byref(placement_s), 0))
        placement_s = placement_s.value
        assert it == K, (it, K)
        p_spmv, p_trsv, p_vec = pr0[PROF_SPMV], pr0[PROF_TRSV], pr0[PROF_VEC]
        vb = 4 if mixed else 8  # the launches of a mixed-precision run are (all but a handful) the fp32 inner SpMVs
        if args.format == "csr":
            b_spmv = spmv_bytes(n, nnz, vb)
            k_spmv = ("fp32 inner CSR SpMV + fused <p,q> (k_csr_pat2<float> for structured matrices, else k_csr_tr<float>; the few fp64 outer residual SpMVs are in the average)"
                      if mixed else "CSR SpMV (k_csr_pat2 for structured matrices -- columns from the row-pattern dictionary, two row blocks per workgroup: traffic below the algorithmic CSR bytes -- else k_csr_tr; rows of 16+ entries: k_csr_wr on stencils / long row patterns, k_csr_wp otherwise; with the fused dot where the solver uses it)")
        else:
            nnz_fmt = 7 * n if args.matrix == "poisson" else (27 * n if args.matrix == "lap27" else nnz)
            b_spmv = 4 * nnz_fmt + vb * (2 * n + nnz_fmt)
            k_spmv = "k_ell<%s>" % ("float" if mixed else "double")
        tkey = None
        if args.matrix == "poisson" and N == 512 and args.format in ("csr", "ell"):
            tkey = "spmv_%s_512%s" % (args.format, "_fp32" if mixed else "")
        if args.matrix == "shell" and args.format == "csr" and not mixed:
            tkey = "spmv_csr_shell"
        if args.matrix == "lap27" and N == 256 and args.format in ("csr", "ell", "hyb") and not mixed:
            tkey = "spmv_%s_lap27_256" % args.format
        r_spmv = roof(k_spmv, b_spmv, p_spmv, traffic_for(tkey) if tkey else None)
        if tri_pc and p_trsv["launches"] > 0:
            tk = "trsv_512" if (args.matrix == "poisson" and N == 512) else ("trsv_lap27_256" if (args.matrix == "lap27" and N == 256) else None)
            if args.matrix == "shell":  # (counter passes exist for the surrogate and for its RCM numbering)
                tk = "trsv_shell" if args.shell_variant == "lex" else "trsv_shell_" + args.shell_variant
            tri_plan = tri_plan_stats(lib, capi)
            prof = roof("sparse triangular solve, one launch per triangle: %s" % (tri_plan.get("lower", {}).get("form", "?")),
                        trsv_bytes(n, nnz, vb), p_trsv, traffic_for(tk) if tk else None)
            kernels["spmv"] = r_spmv
        else:
            prof = r_spmv
        if p_vec["launches"] > 0:
            # fused vector kernels: CG 40 n B per launch (k_cg_update / k_cg_direction: 3 reads + 2 writes),
            # GMRES k_mgs_step 32 n B (3 reads + 1 write)
            if args.solver == "gmres" and os.environ.get("RAMD_MGS_BLOCK", "1") != "0":
                # k_mgs_block, KB projections per pass (GMRES::doFusedMGS): Arnoldi step with m basis vectors =
                # first pass 1 + nc streams (w only read), later passes 2 + KB + nc, last pass 2 + nl (+ <w,w>);
                # the bracketed run did it_prof iterations of restart cycles of `basis`
                KB = lib.ramd_fused_mgs_block_max()
                it_prof, streams, launches = (min(K, 20) if mixed else min(K, 200)), 0, 0
                for j in range(it_prof):
                    m = j % basis + 1
                    nb = (m + KB - 1) // KB
                    for b in range(nb):
                        nc = min(KB, m - KB * b)
                        streams += (1 + nc) if b == 0 else (2 + KB + nc)
                    streams += 2 + (m - KB * (nb - 1))
                    launches += nb + 1
                launches = p_vec["launches"] or launches  # (the same count unless the iteration control stopped early)
                kernels["vector_updates"] = roof("k_mgs_block (modified Gram-Schmidt, %d projections per pass; average over the " % KB +
                                                 "passes of the restart cycles)", streams * n * vb // launches, p_vec)
            else:
                kernels["vector_updates"] = roof("k_mgs_step" if args.solver == "gmres" else "k_cg_update / k_cg_direction",
                                                 (32 if args.solver == "gmres" else 40) * n * vb // 8, p_vec)
        if args.precond in ("mcsgs", "mcgs", "mcilu") and pr0[PROF_PRECOND]["launches"] > 0:
            kernels["precond_apply"] = roof("multi-coloured %s apply (%s)" % (args.precond.upper()[2:], mc_form(args.precond)),
                                            mcsgs_bytes(n, nnz, vb), pr0[PROF_PRECOND],
                                            traffic_for(mc_traffic_key(args.precond)) if (args.matrix == "poisson" and N == 512 and not mixed)
                                            else (traffic_for("mcsgs_lap27_256") if (args.matrix == "lap27" and N == 256 and args.precond == "mcsgs" and not mixed) else None))
        st_pat = C.c_int(0)
        capi.check(lib.ramd_mat_pattern_info(A._h, C.byref(st_pat), None, None))
        if st_pat.value in (1, 2) and args.format in ("csr", "ell", "hyb"):
            # the SAME run with the columns read (the general path every unstructured matrix takes): its own timed
            # K iterations and its own HIP-event average; the row-pattern figure above stays the headline
            A.UseRowPatterns(False)
            d3, i3, _, _, pr3 = run(K, HEAD, HPC, basis, warm=W, prof_iters=(min(K, 20) if mixed else min(K, 200)))
            A.UseRowPatterns(True)
            ck = tkey + "_columns_read" if (tkey and tkey.endswith(("_512", "_512_fp32"))) else None
            cols_read = dict(iters_per_s=round(i3 / d3, 3), ms_per_step=round(d3 / i3 * 1e3, 5),
                             roofline=roof(k_spmv.split(" (")[0] + " with the stored columns read (ramd_mat_pattern_use(m, 0))",
                                           b_spmv, pr3[PROF_SPMV], traffic_for(ck) if ck else None))
        if not args.no_extras and args.solver == "cg" and args.precond == "jacobi" and args.matrix == "poisson":
            # the other two solver/preconditioner pairs of BASELINE.json on the same operator (same
            # "exactly K iterations" protocol; Build() reported separately, as in the reference samples)
            for name, sc, pc, bs, iters in (("gmres30_ilu0", S.GMRES, S.ILU, 30, min(K, 60)),
                                            ("bicgstab_mcsgs", S.BiCGStab, S.MultiColoredSGS, None, min(K, 60))):
                try:
                    d2, i2, r2, tb2, pe = run(iters, sc, pc, bs, warm=min(W, 10), prof_iters=iters)
                    extras[name] = dict(iters_per_s=round(i2 / d2, 2), iters=i2, ms_per_step=round(d2 / i2 * 1e3, 4),
                                        build_s=round(tb2, 3))
                    # first-class evidence: the dominant kernel of the leg with its own HIP-event average
                    big = (args.matrix == "poisson" and N == 512)
                    if name == "gmres30_ilu0" and pe[PROF_TRSV]["launches"] > 0:
                        tp = tri_plan_stats(lib, capi)
                        extras[name]["roofline"] = roof("sparse triangular solve, one launch per triangle: %s" % tp.get("lower", {}).get("form", "?"),
                                                        trsv_bytes(n, nnz, 8), pe[PROF_TRSV], traffic_for("trsv_512") if big else None)
                        extras[name]["tri_plan"] = tp
                    if name == "bicgstab_mcsgs" and pe[PROF_PRECOND]["launches"] > 0:
                        extras[name]["roofline"] = roof("multi-coloured SGS apply (%s)" % mc_form("mcsgs"),
                                                        mcsgs_bytes(n, nnz, 8), pe[PROF_PRECOND],
                                                        traffic_for(mc_traffic_key("mcsgs")) if big else None)
                    extras[name]["kernels"] = {"spmv": roof("CSR SpMV (k_csr_pat2 / k_csr_tr)", spmv_bytes(n, nnz, 8), pe[PROF_SPMV])}
                except Exception as e:
                    extras[name] = dict(error=repr(e))
            t_x = time.perf_counter()
            # BASELINE.json config 5 on this GPU: MixedPrecisionDC(fp64 outer / fp32 CG + Jacobi inner) on the same operator.
            # A "step" is one OUTER iteration (residual in fp64, an fp32 CG solve to 1e-2 relative, correction in fp64); the
            # dominant kernel is the fp32 inner product, once with the row patterns and once with the stored columns read
            try:
                mi, mw = min(K, 20), min(W, 3)
                d5, i5, r5, _, p5 = run_mixed(mi, warm=mw, prof_iters=min(K, 10), pc_cls=S.Jacobi)
                big = (args.matrix == "poisson" and N == 512)
                e5 = dict(outer_iters_per_s=round(i5 / d5, 3), iters=i5, ms_per_step=round(d5 / i5 * 1e3, 4), final_residual=r5,
                          inner_spmv_per_outer=round(p5[PROF_SPMV]["count"] / max(min(K, 10), 1), 1),
                          roofline=roof("fp32 inner CSR SpMV + fused <p,q> (k_csr_pat2<float>; the few fp64 outer residual products are in "
                                        "the average)", spmv_bytes(n, nnz, 4), p5[PROF_SPMV], traffic_for("spmv_csr_512_fp32") if big else None))
                A.UseRowPatterns(False)
                try:
                    d6, i6, _, _, p6 = run_mixed(mi, warm=mw, prof_iters=min(K, 5), pc_cls=S.Jacobi)  # (the same outer iterations: their inner counts differ)
                finally:
                    A.UseRowPatterns(True)
                e5["roofline_columns_read"] = roof("k_csr_tr<float> + fused <p,q>: the same inner product with the stored columns read "
                                                   "(ramd_mat_pattern_use(m, 0))", spmv_bytes(n, nnz, 4), p6[PROF_SPMV], None)
                e5["columns_read"] = dict(outer_iters_per_s=round(i6 / d6, 3), ms_per_step=round(d6 / i6 * 1e3, 4))
                extras["mixed_cg_jacobi"] = e5
            except Exception as e:
                extras["mixed_cg_jacobi"] = dict(error=repr(e))
            log("extras: mixed_cg_jacobi %.1f s" % (time.perf_counter() - t_x))
            t_x = time.perf_counter()
            # BASELINE.json config 3's class: GMRES(30)+ILU(0) on the af_shell10-class operator in reverse Cuthill-McKee order (the
            # numbering with the deep, narrow dependency graph; generators.shell_variant).  Generated in host memory and handed
            # over with SetDataPtrCSR -- no MatrixMarket round trip in the default run (bench.py --matrix shell has it).
            try:
                from rocalution_amd import generators as gen
                snx = args.shell_nx
                t0 = time.perf_counter()
                rp_h, ci_h, va_h = gen.shell_variant(snx, "rcm")
                t_gen = time.perf_counter() - t0
                n3, nnz3 = len(rp_h) - 1, len(ci_h)
                A3 = ra.LocalMatrix(); A3.SetDataPtrCSR(rp_h, ci_h, va_h)
                ones3 = ra.LocalVector(); ones3.Allocate("ones", n3); ones3.Ones()
                rhs3 = ra.LocalVector(); rhs3.Allocate("rhs", n3)
                x3 = ra.LocalVector(); x3.Allocate("x", n3)
                A3.Apply(ones3, rhs3)
                sys3 = (A3, rhs3, x3, lambda: A3.SetDataPtrCSR(rp_h, ci_h, va_h), ra.CSR)
                i3n = min(max(K, 60), 120)
                d3, i3, r3, tb3, p3 = run(i3n, S.GMRES, S.ILU, 30, warm=min(W, 10), prof_iters=min(i3n, 60), system=sys3)
                tp3 = tri_plan_stats(lib, capi)
                e3 = dict(iters_per_s=round(i3 / d3, 2), iters=i3, ms_per_step=round(d3 / i3 * 1e3, 4), build_s=round(tb3, 3),
                          final_residual=r3, generate_s=round(t_gen, 2),
                          workload="af_shell10-class operator (SuiteSparse af_shell10 itself is not available offline): %d x %d mesh "
                                   "nodes x 5 unknowns in reverse Cuthill-McKee order, n=%d, nnz=%d (%.2f per row; af_shell10: "
                                   "n=1508065, nnz=52259885), SPD, CSR fp64, rhs=A*1, x0=0" % (snx, snx, n3, nnz3, nnz3 / n3),
                          roofline=roof("sparse triangular solve, one launch per triangle: %s" % tp3.get("lower", {}).get("form", "?"),
                                        trsv_bytes(n3, nnz3, 8), p3[PROF_TRSV], traffic_for("trsv_shell_rcm")),
                          tri_plan=tp3,
                          kernels={"spmv": roof("CSR SpMV (k_csr_wp: rows of 16+ entries, wave-private passes, lane t sums row t)", spmv_bytes(n3, nnz3, 8),
                                                p3[PROF_SPMV], traffic_for("spmv_csr_shell"))})
                del A3, ones3, rhs3, x3, sys3
                if not args.no_cpu_baseline:
                    e3["cpu_baseline"] = cpu_port_baseline(rp_h, ci_h, va_h, "gmres", "ilu0", 30,
                                                           "the same %d-row operator, in memory" % n3)
                extras["gmres30_ilu0_shell_rcm"] = e3
            except Exception as e:
                extras["gmres30_ilu0_shell_rcm"] = dict(error=repr(e))
            log("extras: gmres30_ilu0_shell_rcm %.1f s" % (time.perf_counter() - t_x))
            t_x = time.perf_counter()
            # The reference's own 3-D operator (gen_3d_laplacian, clients/include/utility.hpp:110-177: the 27-point Laplacian its
            # samples and benchmarks generate) at 256^3, the three solver / preconditioner pairs of BASELINE.json, generated on the
            # device: driver-timed figures for the operator on which none of the 7-point lattice kernels applies (`bench.py
            # --matrix lap27 --grid 256 ...` gives the same lines one by one, with the CPU and vendor columns)
            try:
                N27 = 256
                n27 = N27 ** 3
                A27 = ra.LocalMatrix(); A27.GenLaplace27(N27)
                nnz27 = A27.GetNnz()
                ones27 = ra.LocalVector(); ones27.Allocate("ones", n27); ones27.Ones()
                rhs27 = ra.LocalVector(); rhs27.Allocate("rhs", n27)
                x27 = ra.LocalVector(); x27.Allocate("x", n27)
                A27.Apply(ones27, rhs27)
                sys27 = (A27, rhs27, x27, lambda: A27.GenLaplace27(N27), ra.CSR)
                e27 = dict(workload="3-D 27-point Laplacian %d^3 (n=%d, nnz=%d), CSR fp64, rhs=A*1, x0=0, generated on the device" % (N27, n27, nnz27))
                for name, sc, pc, bs, iters in (("cg_jacobi", S.CG, S.Jacobi, None, min(K, 100)), ("gmres30_ilu0", S.GMRES, S.ILU, 30, min(K, 60)),
                                                ("bicgstab_mcsgs", S.BiCGStab, S.MultiColoredSGS, None, min(K, 60))):
                    d7, i7, r7, tb7, p7 = run(iters, sc, pc, bs, warm=min(W, 10), prof_iters=iters, system=sys27)
                    leg = dict(iters_per_s=round(i7 / d7, 2), iters=i7, ms_per_step=round(d7 / i7 * 1e3, 4), build_s=round(tb7, 3),
                               kernels={"spmv": roof("CSR SpMV (k_csr_wr: rows of 16+ entries walked lane = row from wave-private LDS images; row patterns)",
                                                     spmv_bytes(n27, nnz27, 8), p7[PROF_SPMV], traffic_for("spmv_csr_lap27_256"))})
                    if name == "gmres30_ilu0" and p7[PROF_TRSV]["launches"] > 0:
                        tp7 = tri_plan_stats(lib, capi)
                        leg["roofline"] = roof("sparse triangular solve, one launch per triangle: %s" % tp7.get("lower", {}).get("form", "?"),
                                               trsv_bytes(n27, nnz27, 8), p7[PROF_TRSV], traffic_for("trsv_lap27_256"))
                        leg["tri_plan"] = tp7
                    if name == "bicgstab_mcsgs" and p7[PROF_PRECOND]["launches"] > 0:
                        leg["roofline"] = roof("multi-coloured SGS apply (%s)" % mc_form("mcsgs"), mcsgs_bytes(n27, nnz27, 8),
                                               p7[PROF_PRECOND], traffic_for("mcsgs_lap27_256"))
                    e27[name] = leg
                del A27, ones27, rhs27, x27, sys27
                extras["lap27_256"] = e27
            except Exception as e:
                extras["lap27_256"] = dict(error=repr(e))
            log("extras: lap27_256 %.1f s" % (time.perf_counter() - t_x))
    else:
        if args.matrix not in ("poisson", "lap27"):
            raise SystemExit("the distributed driver generates z-slabs of the Poisson operator or of the 27-point Laplacian")
        if args.matrix == "lap27":  # the operator the reference's MPI benchmark generates per rank (clients/include/common.hpp:926-1249)
            n, nnz = N ** 3, (3 * N - 2) ** 3
            wl = "3-D 27-point Laplacian %d^3 (n=%d, nnz=%d)" % (N, n, nnz)
        else:
            n, nnz = N ** 3, 7 * N ** 3 - 6 * N ** 2
            wl = "3-D 7-point Poisson %d^3 (n=%d, nnz=%d)" % (N, n, nnz)
        z0, z1 = (N * rank) // world, (N * (rank + 1)) // world
        g = C.c_void_p()
        SK = {"cg": capi.SOLVER_CG, "gmres": capi.SOLVER_GMRES, "bicgstab": capi.SOLVER_BICGSTAB}
        PK = {"none": capi.PC_NONE, "jacobi": capi.PC_JACOBI, "ilu0": capi.PC_ILU0, "mcsgs": capi.PC_MCSGS,
              "mcgs": capi.PC_MCGS, "mcilu": capi.PC_MCILU, "ic": capi.PC_IC, "sgs": capi.PC_SGS,
              "uaamg": capi.PC_UAAMG, "saamg": capi.PC_SAAMG,  # all but Jacobi: BlockJacobi over the ranks
              "global-uaamg": capi.PC_GLOBAL_UAAMG, "global-saamg": capi.PC_GLOBAL_SAAMG}
        if args.precond not in PK:
            raise SystemExit("--precond %s: not wired into the distributed driver" % args.precond)
        if mixed:  # config 5: fp64 defect correction around fp32 CG + Jacobi
            capi.check(lib.ramd_gsolver_create_mixed(comm, capi.SOLVER_CG, PK[args.precond], C.byref(g)))
            capi.check(lib.ramd_gsolver_init_inner(g, 1e-5, 1e-2, 1e20, 100000))
        else:
            capi.check(lib.ramd_gsolver_create(comm, SK[args.solver], PK[args.precond], C.byref(g)))
        capi.check((lib.ramd_gsolver_setup_laplace27 if args.matrix == "lap27" else lib.ramd_gsolver_setup_poisson)(g, N, z0, z1))

        def run(iters, warm=0):
            capi.check(lib.ramd_gsolver_init(g, NEVER[0], NEVER[1], NEVER[2], 0, warm + iters))
            capi.check(lib.ramd_gsolver_set_time_mark(g, warm if warm > 0 else -1))
            if fmt != ra.CSR:  # preconditioners are built from the CSR state (second run: convert back first)
                capi.check(lib.ramd_gsolver_convert(g, ra.CSR))
            barrier()
            tb0 = time.perf_counter()
            capi.check(lib.ramd_gsolver_build(g))
            barrier()
            tb = time.perf_counter() - tb0  # (Build() of solver + preconditioner on this rank, barrier to barrier)
            if fmt != ra.CSR:  # converted after Build(), as the reference tests do
                capi.check(lib.ramd_gsolver_convert(g, fmt))
            capi.check(lib.ramd_gsolver_prepare_ones(g))
            barrier()
            t0 = time.perf_counter()
            capi.check(lib.ramd_gsolver_solve_device(g))
            barrier()
            dt = time.perf_counter() - t0
            if warm > 0:  # the clock of every rank started when it had checked iteration `warm` (ranks are in step: two
                sec = C.c_double(0)  # all-reduces per iteration), device drained; max over ranks below
                capi.check(lib.ramd_gsolver_seconds_since_time_mark(g, C.byref(sec)))
                dt = sec.value
                assert dt > 0, "the marked iteration was not reached"
            itc, st, rs = C.c_int(0), C.c_int(0), C.c_double(0)
            capi.check(lib.ramd_gsolver_result(g, C.byref(itc), C.byref(st), C.byref(rs)))
            return dt, itc.value - warm, rs.value, tb

        # W untimed warm-up iterations, then EXACTLY K timed ones, in one Solve; --warmup 0 times the whole Solve
        capi.check(lib.ramd_placement_seconds(None, 1))
        dt, it, res, tbuild = run(K, warm=W)
        placement_s = C.c_double(0.0)
        capi.check(lib.ramd_placement_seconds(C.byref(placement_s), 0))
        placement_s = placement_s.value
        assert it == K, (it, K)
        import torch
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])
        # --- scaling leg (untimed): the same run with the interior SpMV, the halo exchange, the exposed part of
        # the halo wait and the all-reduces bracketed by HIP events on their streams
        chans = (PROF_SPMV, PROF_HALO, PROF_HALO_WAIT, PROF_ALLREDUCE, PROF_TRSV)
        for ch in chans:
            capi.check(lib.ramd_prof_enable(ch, 1))
        kp = min(K, 20) if mixed else min(K, 100)
        _, itp, _, _ = run(kp)
        pr = {ch: prof_get(lib, capi, ch) for ch in chans}
        for ch in chans:
            capi.check(lib.ramd_prof_enable(ch, 0))
        k = z1 - z0
        n_loc = k * N * N
        nnz_int = n_loc + 4 * (N - 1) * N * k + 2 * (k - 1) * N * N
        nnz_ghost = ((1 if z0 > 0 else 0) + (1 if z1 < N else 0)) * N * N
        if args.matrix == "lap27":  # (3N - 2)^2 entries per pair of planes one apart or equal
            nnz_int = (3 * N - 2) ** 2 * (3 * k - 2)
            nnz_ghost = ((1 if z0 > 0 else 0) + (1 if z1 < N else 0)) * (3 * N - 2) ** 2
        vb = 4 if mixed else 8
        if args.format == "csr":
            b_loc = spmv_bytes(n_loc, nnz_int, vb)
        else:
            w_fmt = 27 if args.matrix == "lap27" else 7
            b_loc = 4 * w_fmt * n_loc + vb * (2 * n_loc + w_fmt * n_loc)
        # interior launches carry the fused dot in most solvers; ghost COO ApplyAdd launches are in the same channel:
        # separate them by duration rank is fragile, so the channel average is reported together with the count per iteration
        mine = dict(rank=rank, rows=n_loc, interior_nnz=nnz_int, ghost_nnz=nnz_ghost, spmv_launches_per_iter=round(pr[PROF_SPMV]["count"] / max(itp, 1), 2),
                    spmv_avg_ms=round(pr[PROF_SPMV]["avg_ms"], 5), spmv_max_ms=round(pr[PROF_SPMV]["max_ms"], 5),
                    interior_GBps=round(b_loc / (pr[PROF_SPMV]["max_ms"] * 1e-3) / 1e9, 1) if pr[PROF_SPMV]["max_ms"] > 0 else 0.0,
                    halo_ms=round(pr[PROF_HALO]["avg_ms"], 5), halo_wait_ms=round(pr[PROF_HALO_WAIT]["avg_ms"], 5),
                    halo_exchanges=pr[PROF_HALO]["count"], allreduces=pr[PROF_ALLREDUCE]["count"],
                    allreduce_avg_ms=round(pr[PROF_ALLREDUCE]["avg_ms"], 5), iters=itp, algorithmic_bytes=b_loc)
        allr = [None] * world
        dist.all_gather_object(allr, mine)
        nr = C.c_int(0)
        capi.check(lib.ramd_comm_rccl_count(comm, C.byref(nr)))
        if rank == 0:
            # interior SpMV = the LONGEST launch kind of the channel (the ghost ApplyAdd is ~N^2 entries): use the max
            t_int = max(r["spmv_max_ms"] for r in allr)
            agg_bytes = sum(r["algorithmic_bytes"] for r in allr)
            ach = agg_bytes / (t_int * 1e-3) / 1e9 if t_int > 0 else 0.0
            prof = dict(bound="hbm", achieved=round(ach, 1), peak=HBM_PEAK_GBPS * world, unit="GB/s",
                        frac=round(ach / (HBM_PEAK_GBPS * world), 4), traffic=None,
                        kernel="interior SpMV of every rank (aggregate algorithmic bytes / slowest rank's longest launch)",
                        algorithmic_bytes=int(agg_bytes), per_rank=allr)
            halo = max(r["halo_ms"] for r in allr)
            wait = max(r["halo_wait_ms"] for r in allr)
            scaling_fields = dict(halo_ms=halo, halo_wait_ms=wait,
                                  halo_overlap_frac=round(max(0.0, 1.0 - wait / halo), 4) if halo > 0 else None,
                                  halo_time_frac=round(wait * allr[0]["halo_exchanges"] / max(itp, 1) / (dt / it * 1e3), 4),
                                  allreduces_per_iter=round(allr[0]["allreduces"] / max(itp, 1), 2),
                                  halo_exchanges_per_iter=round(allr[0]["halo_exchanges"] / max(itp, 1), 2),
                                  rccl_nranks=nr.value,
                                  transport="rccl" if not rehearsal else "callback (host-staged through gloo; the ranks share "
                                                                         "the device(s))")
            if rehearsal:
                scaling_fields["rehearsal"] = True  # the N > 1 code path on one box: NOT a scaling measurement

    if rank == 0:
        out = {
            "metric": "%s iterations/s, %s %s fp64" % (label, "3D 7-pt Poisson %d^3" % N if args.matrix == "poisson"
                                                       else "3D 27-pt Laplacian %d^3" % N if args.matrix == "lap27"
                                                       else ("af_shell10-class surrogate (n=%d)" % n if args.matrix == "shell"
                                                             else "%s (n=%d)" % (os.path.basename(args.mtx), n)), args.format.upper()),
            "value": round(it / dt, 3), "unit": "iters/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(dt / it * 1e3, 5), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64/f32" if mixed else "f64",
            "data": "synthetic",
            "config": {"workload": "%s+%s, %s %s fp64, rhs=A*1, x0=0, row-split over %d GPU(s)"
                                   % (args.solver, args.precond, wl, args.format.upper(), world),
                       "parallelism": "rows%d" % world, "fused": True},
            "final_residual": res, "build_s": round(tbuild, 4),
            # wall time of the placement measurements at the first Solve of the timed solver (before the timed window:
            # work vectors of the fused loops are placed by trial, once per Build; RAMD_PLACE_TRIES=0 switches it off)
            "placement_s": round(placement_s, 4),
        }
        if prof is not None:
            out["roofline"] = prof
        if cols_read is not None:
            out["roofline_columns_read"] = cols_read["roofline"]
            out["columns_read"] = {k: cols_read[k] for k in ("iters_per_s", "ms_per_step")}
        if kernels:
            out["kernels"] = kernels
            if "spmv" in kernels:
                out["spmv_GBps"] = kernels["spmv"]["achieved"]
        if prof is not None and "spmv_GBps" not in out and world == 1:
            out["spmv_GBps"] = prof["achieved"]
        out.update(scaling_fields)
        if tri_plan:
            out["tri_plan"] = tri_plan
        if ingest:
            out["ingest"] = ingest
        if extras:
            out["extras"] = extras
        if world == 1 and not args.no_cpu_baseline and not args.force_global:
            out["cpu_baseline"] = cpu_baseline(args, mtx_path)
        if world == 1 and not args.no_reference_gpu and not args.force_global:
            rg = reference_gpu(args)
            if rg is not None:
                out["reference_gpu"] = rg
            # ... and the same vendor column for the other two legs (rocSPARSE csrsv for ILU(0): hip_matrix_csr.cpp:1756-1821)
            for name, so, pc in (("gmres30_ilu0", "gmres", "ilu0"), ("bicgstab_mcsgs", "bicgstab", "mcsgs")):
                if name in extras and "iters" in extras[name]:
                    rg = reference_gpu(argparse.Namespace(**dict(vars(args), solver=so, precond=pc, steps=extras[name]["iters"])))
                    if rg is not None:
                        extras[name]["reference_gpu"] = rg
        C.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)
    if mtx_generated and mtx_path and os.path.exists(mtx_path):
        os.unlink(mtx_path)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
