#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native Krylov core.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json `metric`): CG + Jacobi on the 3-D 7-point Poisson 512^3 operator in CSR, fp64,
rhs = A*1, x0 = 0 (the reference samples' convention, clients/samples/cg.cpp:77-82), synthetic operator
generated on the device.  A "step" is ONE CG iteration; the timed region is Solve() running exactly K
iterations (tolerances that cannot be met), bracketed by barrier + device sync, MAX over ranks.
N > 1: the 512^3 rows are split into z-slabs across ranks (strong scaling), halo exchange + scalar
all-reduce over RCCL (GlobalMatrix / GlobalVector path).

Extra objects on the JSON line (N = 1):
  roofline     -- the CSR SpMV kernel: algorithmic bytes 4(n+nnz)+8(2n+nnz) (clients/samples/
                  benchmark.cpp:213-233) / its average duration measured with HIP events around every SpMV
                  launch inside a live CG run; peak = 8 TB/s (MI355X HBM3E).
  cpu_baseline -- the same solver on the host cores: the genuine rocALUTION OpenMP backend through
                  oracle/_ref/ref_probe when the ROCm image ships librocalution ("reference"), else the
                  C oracle ("port"); bounded sample (smaller grid), stated in `sample`.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def spmv_bytes(n, nnz, vbytes=8):
    return 4 * (n + nnz) + vbytes * (2 * n + nnz)


def cpu_baseline(args):
    """CG+Jacobi on the host: reference OpenMP backend if available, else the C oracle (port)."""
    ncpu = os.cpu_count() or 1
    try:
        phys = int(subprocess.check_output("lscpu -p=CORE,SOCKET | grep -v '^#' | sort -u | wc -l",
                                           shell=True).decode().strip())
    except Exception:
        phys = ncpu
    threads = max(1, min(phys, ncpu))
    Nc, iters = args.cpu_grid, args.cpu_iters
    probe = os.path.join(ROOT, "oracle", "_ref", "ref_probe")
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), OMP_PROC_BIND="close", OMP_PLACES="cores")
    if os.path.exists(probe) and os.path.exists("/opt/rocm/lib/librocalution.so"):
        try:
            out = subprocess.check_output([probe, "bench", str(Nc), str(iters), str(threads), "0", "cg", "jacobi"],
                                          env=env, stderr=subprocess.DEVNULL, timeout=600).decode()
            rec = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
            return dict(value=rec["iters_per_s"], unit="iters/s", cores=threads, kind="reference",
                        sample="CG+Jacobi, 3-D Poisson %d^3 CSR fp64 (%.3g of the 512^3 rows), %d iterations, "
                               "rocALUTION %s host/OpenMP backend (accelerator disabled)" %
                               (Nc, (Nc / 512.0) ** 3, rec["iters"], "installed"),
                        spmv_GBps=rec["spmv_GBps"], equivalent_512_iters_per_s=rec["iters_per_s"] * (Nc / 512.0) ** 3)
        except Exception as e:  # fall through to the port
            log("cpu_baseline: reference probe failed (%r), using the oracle port" % (e,))
    import numpy as np
    from oracle import oracle as orc
    from rocalution_amd import generators as gen
    orc.build()
    orc.set_threads(threads)
    Np = min(Nc, 128)
    rp, ci, va = gen.poisson7(Np)
    rhs = orc.csr_apply(rp, ci, va, np.ones(len(rp) - 1))
    t0 = time.time()
    r = orc.solve(rp, ci, va, rhs, solver=orc.CG, precond=orc.PC_JACOBI, abs_tol=0.0, rel_tol=0.0,
                  div_tol=1e300, max_iter=iters, history=False)
    dt = time.time() - t0
    return dict(value=r["iters"] / dt, unit="iters/s", cores=threads, kind="port",
                sample="CG+Jacobi, 3-D Poisson %d^3 CSR fp64, %d iterations, C oracle (OpenMP)" % (Np, r["iters"]),
                equivalent_512_iters_per_s=r["iters"] / dt * (Np / 512.0) ** 3)


def reference_gpu(args):
    """optional vendor column: the reference's own rocSPARSE/rocBLAS HIP backend on this GPU"""
    probe = os.path.join(ROOT, "oracle", "_ref", "ref_probe")
    if not (os.path.exists(probe) and os.path.exists("/opt/rocm/lib/librocalution_hip.so")):
        return None
    try:
        out = subprocess.check_output([probe, "bench", str(args.grid), str(args.steps), "0", "1", "cg",
                                       "jacobi"], stderr=subprocess.DEVNULL, timeout=900).decode()
        rec = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
        return dict(iters_per_s=rec["iters_per_s"], spmv_GBps=rec["spmv_GBps"], t_spmv_ms=rec["t_spmv_s"] * 1e3,
                    what="rocALUTION (installed) HIP backend = rocSPARSE/rocBLAS wrapper, same workload")
    except Exception as e:
        log("reference_gpu: failed (%r)" % (e,))
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--grid", type=int, default=512, help="Poisson grid edge N (operator is N^3 x N^3)")
    ap.add_argument("--format", default="csr", choices=["csr", "ell", "hyb", "dia"])
    ap.add_argument("--cpu-grid", type=int, default=256)
    ap.add_argument("--cpu-iters", type=int, default=40)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reference-gpu", action="store_true")
    ap.add_argument("--solver", default="cg", choices=["cg", "gmres", "bicgstab", "mixed"],
                    help="headline = cg; the others run the remaining BASELINE.json configs through the same harness")
    ap.add_argument("--precond", default="jacobi", choices=["none", "jacobi", "ilu0", "mcsgs", "mcgs", "mcilu", "ic", "sgs", "uaamg", "saamg"])
    ap.add_argument("--itsolve", type=int, default=0,
                    help="ILU / IC / SGS: iterative triangular solves (TriSolverAlg_Iterative) with this many sweeps")
    ap.add_argument("--force-global", action="store_true",
                    help="1 process: still go through the GlobalMatrix/RCCL code path (communicator of size 1, "
                         "collectives not skipped) - a check of the N>1 plumbing on a 1-GPU box")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the GMRES(30)+ILU(0) and BiCGStab+MC-SGS legs (reported under `extras`)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            log("bench.py: --gpus %d needs one process per GPU (torch.distributed.run); running 1" % args.gpus)
        args.gpus = world

    if args.force_global:
        os.environ["RAMD_COMM_FORCE_COLLECTIVES"] = "1"
    import rocalution_amd as ra
    from rocalution_amd import capi
    lib = capi.load()
    ra.init_rocalution(local_rank)
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
    n, nnz = N ** 3, 7 * N ** 3 - 6 * N ** 2
    fmt = {"csr": ra.CSR, "ell": ra.ELL, "hyb": ra.HYB, "dia": ra.DIA}[args.format]
    K, W = args.steps, args.warmup
    NEVER = (0.0, 0.0, 1e300)  # abs / rel / div tolerances that cannot trigger: exactly max_iter steps

    prof = None
    if world == 1 and not args.force_global:
        from rocalution_amd import solvers as S
        A = ra.LocalMatrix()
        A.GenPoisson7(N)
        ones = ra.LocalVector(); ones.Allocate("ones", n); ones.Ones()
        rhs = ra.LocalVector(); rhs.Allocate("rhs", n)
        x = ra.LocalVector(); x.Allocate("x", n)
        A.Apply(ones, rhs)

        def run(iters, solver_cls=S.CG, pc_cls=S.Jacobi, basis=None):
            if A.GetFormat() != ra.CSR:  # preconditioners are built from the CSR state
                A.GenPoisson7(N)
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
            ls.Init(NEVER[0], NEVER[1], NEVER[2], iters)
            tb = time.perf_counter()
            ls.Build()
            ra.sync()
            tb = time.perf_counter() - tb
            if fmt != ra.CSR and A.GetFormat() == ra.CSR:
                A.ConvertTo(fmt)
            x.Zeros()
            barrier()
            t0 = time.perf_counter()
            ls.Solve(rhs, x)
            barrier()
            dt = time.perf_counter() - t0
            it = ls.GetIterationCount()
            res = ls.GetCurrentResidual()
            ls.Clear()
            return dt, it, res, tb

        HEAD = {"cg": S.CG, "gmres": S.GMRES, "bicgstab": S.BiCGStab}.get(args.solver, S.CG)
        HPC = {"none": None, "jacobi": S.Jacobi, "ilu0": S.ILU, "mcsgs": S.MultiColoredSGS,
               "mcgs": S.MultiColoredGS, "mcilu": S.MultiColoredILU, "ic": S.IC, "sgs": S.SGS, "uaamg": S.UAAMG, "saamg": S.SAAMG}[args.precond]
        if args.solver == "mixed":
            def run(iters, *_a):  # noqa: F811  (config 5 on one GPU)
                inner = S.CG(np.float32)
                if HPC is not None:
                    inner.SetPreconditioner(HPC())
                inner.Init(1e-5, 1e-2, 1e20, 100000)
                mp = S.MixedPrecisionDC(); mp.SetOperator(A); mp.Set(inner)
                mp.Init(NEVER[0], NEVER[1], NEVER[2], iters)
                mp.Build(); x.Zeros(); barrier()
                t0 = time.perf_counter(); mp.Solve(rhs, x); barrier()
                dt = time.perf_counter() - t0
                r = (dt, mp.GetIterationCount(), mp.GetCurrentResidual(), 0.0)
                mp.Clear()
                return r
            run(W)
            dt, it, res, tbuild = run(K)
        else:
            run(W, HEAD, HPC, 30 if args.solver == "gmres" else None)  # warmup
            dt, it, res, tbuild = run(K, HEAD, HPC, 30 if args.solver == "gmres" else None)
        assert it == K, (it, K)
        # --- roofline leg: same run with every SpMV launch bracketed by HIP events
        capi.check(lib.ramd_prof_spmv_enable(1))
        run(min(K, 200), S.CG, S.Jacobi, None) if args.solver != "mixed" else run(min(K, 20))
        cnt, avg, mn, mx = C.c_int(0), C.c_double(0), C.c_double(0), C.c_double(0)
        capi.check(lib.ramd_prof_spmv_result(C.byref(cnt), C.byref(avg), C.byref(mn), C.byref(mx)))
        capi.check(lib.ramd_prof_spmv_enable(0))
        nnz_fmt = nnz if args.format == "csr" else 7 * n
        mixed = args.solver == "mixed"
        vb = 4 if mixed else 8  # the launches of a mixed-precision run are (all but a handful) the fp32 inner SpMVs
        bytes_alg = spmv_bytes(n, nnz, vb) if args.format == "csr" else 4 * nnz_fmt + vb * (2 * n + nnz_fmt)
        if args.format == "dia":  # values only (7 diagonals x n), x and y once
            bytes_alg = vb * (2 * n + nnz_fmt)
        ach = bytes_alg / (avg.value * 1e-3) / 1e9 if avg.value > 0 else 0.0
        traffic = None  # HBM bytes per launch from the PMC counters: measured offline with rocprofv3
        tfile = os.path.join(ROOT, "profiles", "r01_traffic.json")  # (separate --pmc passes), see that file
        if os.path.exists(tfile) and N == 512 and args.format == "csr" and not mixed:
            traffic = json.load(open(tfile)).get("traffic_bytes")
        kname = ("k_csr_tr<float,0,true> (fp32 inner CSR SpMV + fused <p,q>; the few fp64 outer residual SpMVs are "
                 "in the average)" if mixed else "k_csr_tr<double,0,true> (CSR SpMV + fused <p,q>)")
        prof = dict(bound="hbm", achieved=round(ach, 1), peak=HBM_PEAK_GBPS, unit="GB/s",
                    frac=round(ach / HBM_PEAK_GBPS, 4), traffic=traffic, kernel=kname
                    if args.format == "csr" else "k_%s<%s>" % ("dia" if args.format == "dia" else "ell",
                                                                "float" if mixed else "double"), launches=cnt.value,
                    avg_ms=round(avg.value, 5), min_ms=round(mn.value, 5), max_ms=round(mx.value, 5),
                    algorithmic_bytes=bytes_alg)
        extras = {}
        if not args.no_extras and args.solver == "cg" and args.precond == "jacobi":
            # the other two solver/preconditioner pairs of BASELINE.json on the same operator (same
            # "exactly K iterations" protocol; Build() reported separately, as in the reference samples)
            for name, sc, pc, basis, iters in (("gmres30_ilu0", S.GMRES, S.ILU, 30, min(K, 60)),
                                               ("bicgstab_mcsgs", S.BiCGStab, S.MultiColoredSGS, None, min(K, 60))):
                try:
                    d2, i2, r2, tb2 = run(iters, sc, pc, basis)
                    extras[name] = dict(iters_per_s=round(i2 / d2, 2), iters=i2, build_s=round(tb2, 3))
                except Exception as e:
                    extras[name] = dict(error=repr(e))
            # time to solution (relative residual 1e-8) of the same system with the aggregation AMG as CG's
            # preconditioner, next to plain CG+Jacobi run to the same tolerance: setup and solve reported apart
            for name, pc_cls in (("cg_jacobi_to_1e-8", S.Jacobi), ("cg_uaamg_to_1e-8", S.UAAMG)):
                try:
                    if A.GetFormat() != ra.CSR:
                        A.GenPoisson7(N)
                    ls = S.CG(); ls.SetOperator(A); ls.SetPreconditioner(pc_cls()); ls.Init(1e-15, 1e-8, 1e8, 5000)
                    tb = time.perf_counter(); ls.Build(); ra.sync(); tb = time.perf_counter() - tb
                    x.Zeros(); ra.sync()
                    t0 = time.perf_counter(); ls.Solve(rhs, x); ra.sync(); ts = time.perf_counter() - t0
                    extras[name] = dict(iters=ls.GetIterationCount(), status=ls.GetSolverStatus(),
                                        setup_s=round(tb, 3), solve_s=round(ts, 3))
                    ls.Clear()
                except Exception as e:
                    extras[name] = dict(error=repr(e))
    else:
        z0, z1 = (N * rank) // world, (N * (rank + 1)) // world
        g = C.c_void_p()
        SK = {"cg": capi.SOLVER_CG, "gmres": capi.SOLVER_GMRES, "bicgstab": capi.SOLVER_BICGSTAB}
        PK = {"none": capi.PC_NONE, "jacobi": capi.PC_JACOBI, "ilu0": capi.PC_ILU0, "mcsgs": capi.PC_MCSGS,
              "mcgs": capi.PC_MCGS, "mcilu": capi.PC_MCILU, "ic": capi.PC_IC, "sgs": capi.PC_SGS,
              "uaamg": capi.PC_UAAMG, "saamg": capi.PC_SAAMG}  # all but Jacobi: BlockJacobi over the ranks
        if args.precond not in PK:
            raise SystemExit("--precond %s: not wired into the distributed driver" % args.precond)
        if args.solver == "mixed":  # config 5: fp64 defect correction around fp32 CG + Jacobi
            capi.check(lib.ramd_gsolver_create_mixed(comm, capi.SOLVER_CG, PK[args.precond], C.byref(g)))
            capi.check(lib.ramd_gsolver_init_inner(g, 1e-5, 1e-2, 1e20, 100000))
        else:
            capi.check(lib.ramd_gsolver_create(comm, SK[args.solver], PK[args.precond], C.byref(g)))
        capi.check(lib.ramd_gsolver_setup_poisson(g, N, z0, z1))

        def run(iters):
            capi.check(lib.ramd_gsolver_init(g, NEVER[0], NEVER[1], NEVER[2], 0, iters))
            if fmt != ra.CSR:  # preconditioners are built from the CSR state (second run: convert back first)
                capi.check(lib.ramd_gsolver_convert(g, ra.CSR))
            capi.check(lib.ramd_gsolver_build(g))
            if fmt != ra.CSR:  # converted after Build(), as the reference tests do
                capi.check(lib.ramd_gsolver_convert(g, fmt))
            capi.check(lib.ramd_gsolver_prepare_ones(g))
            barrier()
            t0 = time.perf_counter()
            capi.check(lib.ramd_gsolver_solve_device(g))
            barrier()
            dt = time.perf_counter() - t0
            itc, st, rs = C.c_int(0), C.c_int(0), C.c_double(0)
            capi.check(lib.ramd_gsolver_result(g, C.byref(itc), C.byref(st), C.byref(rs)))
            return dt, itc.value, rs.value, 0.0

        run(W)
        dt, it, res, tbuild = run(K)
        assert it == K, (it, K)
        if dist is not None:
            import torch
            t = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t[0])
        extras = {}

    if rank == 0:
        out = {
            "metric": "%s iterations/s, 3D 7-pt Poisson %d^3 %s fp64" % (
                {"cg": "CG", "gmres": "GMRES(30)", "bicgstab": "BiCGStab", "mixed": "MixedPrecisionDC(fp64/fp32 CG)"}[args.solver]
                + "+" + {"none": "none", "jacobi": "Jacobi", "ilu0": "ILU(0)", "mcsgs": "MC-SGS", "mcgs": "MC-GS",
                         "mcilu": "MC-ILU(0,1)", "ic": "IC", "sgs": "SGS", "uaamg": "UAAMG(PMIS)", "saamg": "SAAMG(PMIS)"}[args.precond], N,
                args.format.upper()),
            "value": round(it / dt, 3), "unit": "iters/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(dt / it * 1e3, 5), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64/f32" if args.solver == "mixed" else "f64",
            "data": "synthetic",
            "config": {"workload": "%s+%s, 3-D 7-point Poisson %d^3 (n=%d, nnz=%d) %s fp64, rhs=A*1, x0=0, "
                                   "row-split over %d GPU(s)" % (args.solver, args.precond, N, n, nnz,
                                                                 args.format.upper(), world),
                       "parallelism": "rows%d" % world, "fused": True},
            "final_residual": res, "build_s": round(tbuild, 4),
        }
        if prof is not None:
            out["roofline"] = prof
            out["spmv_GBps"] = prof["achieved"]
        if extras:
            out["extras"] = extras
        if world == 1 and not args.no_cpu_baseline and not args.force_global:
            out["cpu_baseline"] = cpu_baseline(args)
        if world == 1 and not args.no_reference_gpu and not args.force_global:
            rg = reference_gpu(args)
            if rg is not None:
                out["reference_gpu"] = rg
        C.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
