// ref_probe.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Own driver (no reference source copied) that links the GENUINE rocALUTION library the ROCm
// image ships (/opt/rocm/lib/librocalution.so, public headers /opt/rocm/include/rocalution) and
// runs its host/OpenMP backend (accelerator disabled) -- or, for the optional vendor column, its
// rocSPARSE/rocBLAS HIP backend -- on inputs handed over as raw binary files.
//
//   ref_probe gen   <indir> <outdir>            golden fixtures (see oracle/gen_golden.py)
//   ref_probe bench <N | lap27:N | file.mtx> <iters> <threads> <accel 0|1> [solver=cg|gmres|bicgstab] [precond=...]
//                                               CG+Jacobi etc. on 3-D 7-pt Poisson N^3, prints JSON
//
// Built by oracle/Makefile into oracle/_ref/ (git-ignored). Used by oracle/gen_golden.py in the
// dev container and by bench.py's cpu_baseline leg ("kind":"reference") on the GPU box.
#include <rocalution/rocalution.hpp>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

using namespace rocalution;

typedef LocalMatrix<double> MatD;
typedef LocalVector<double> VecD;
typedef LocalMatrix<float>  MatF;
typedef LocalVector<float>  VecF;

static std::string g_out;

template <typename X>
static void dump(const std::string& name, const X* p, size_t n)
{
    std::ofstream f(g_out + "/" + name + ".bin", std::ios::binary);
    f.write(reinterpret_cast<const char*>(p), sizeof(X) * n);
}
static void dump_vec(const std::string& name, const VecD& v)
{
    std::vector<double> h(v.GetSize());
    v.CopyToHostData(h.data());
    dump(name, h.data(), h.size());
}
template <typename X>
static std::vector<X> slurp(const std::string& path)
{
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if(!f)
    {
        std::cerr << "cannot open " << path << std::endl;
        exit(2);
    }
    size_t         sz = f.tellg();
    std::vector<X> v(sz / sizeof(X));
    f.seekg(0);
    f.read(reinterpret_cast<char*>(v.data()), sz);
    return v;
}

struct Csr
{
    int64_t              n, nnz;
    std::vector<int32_t> rp, ci;
    std::vector<double>  va;
};

static void load_into(const Csr& A, MatD& m, const std::string& name)
{
    m.Clear();
    m.AllocateCSR(name, A.nnz, A.n, A.n);
    m.CopyFromCSR(A.rp.data(), A.ci.data(), A.va.data());
}

static void dump_csr(const std::string& name, const MatD& m)
{
    std::vector<int32_t> rp(m.GetM() + 1), ci(m.GetNnz());
    std::vector<double>  va(m.GetNnz());
    m.CopyToCSR(rp.data(), ci.data(), va.data());
    dump(name + "_rowptr", rp.data(), rp.size());
    dump(name + "_col", ci.data(), ci.size());
    dump(name + "_val", va.data(), va.size());
}

// one Build()+Solve(); writes <tag>_hist (text->bin), <tag>_x, <tag>_meta (iters,status,res)
template <class Solver>
static void run_solver(const std::string& tag, Solver& ls, VecD& rhs, VecD& x)
{
    ls.Verbose(0);
    ls.RecordResidualHistory();
    ls.Solve(rhs, &x);
    std::string hf = g_out + "/" + tag + "_hist.txt";
    if(ls.GetIterationCount() > 0) // WriteHistoryToFile asserts on a run without counted iterations
        ls.RecordHistory(hf);
    else
    {
        std::ofstream empty(hf.c_str());
    }
    double meta[3] = {(double)ls.GetIterationCount(), (double)ls.GetSolverStatus(),
                      ls.GetCurrentResidual()};
    dump(tag + "_meta", meta, 3);
    dump_vec(tag + "_x", x);
}

static int cmd_gen(const std::string& in, const std::string& out)
{
    g_out = out;
    disable_accelerator_rocalution(true);
    init_rocalution();
    set_omp_threads_rocalution(1); // fixtures are generated single-threaded (deterministic sums)

    std::vector<int64_t> hdr = slurp<int64_t>(in + "/hdr.bin");
    Csr                  A;
    A.n   = hdr[0];
    A.nnz = hdr[1];
    A.rp  = slurp<int32_t>(in + "/rowptr.bin");
    A.ci  = slurp<int32_t>(in + "/col.bin");
    A.va  = slurp<double>(in + "/val.bin");
    std::vector<double> xin = slurp<double>(in + "/x.bin");
    std::vector<double> yin = slurp<double>(in + "/y.bin");
    int                 do_solvers = (int)hdr[2];
    const bool          symmetric_spd = do_solvers != 0; // the solver cases are the SPD operators
    int                 basis      = (int)hdr[3];

    MatD mat;
    load_into(A, mat, "A");
    VecD x, y, ones, rhs, tmp;
    x.Allocate("x", A.n);
    y.Allocate("y", A.n);
    ones.Allocate("ones", A.n);
    rhs.Allocate("rhs", A.n);
    tmp.Allocate("tmp", A.n);
    x.CopyFromHostData(xin.data());
    ones.Ones();

    // --- SpMV in every format -------------------------------------------------------------
    mat.Apply(x, &y);
    dump_vec("spmv_csr", y);
    y.CopyFromHostData(yin.data());
    mat.ApplyAdd(x, -0.75, &y);
    dump_vec("spmv_csr_add", y);
    mat.Apply(ones, &rhs);
    dump_vec("rhs_ones", rhs);

    {
        MatD e;
        e.CloneFrom(mat);
        e.ConvertToELL();
        int fmt = e.GetFormat();
        dump("ell_format", &fmt, 1);
        if(fmt == ELL)
        {
            e.Apply(x, &y);
            dump_vec("spmv_ell", y);
            y.CopyFromHostData(yin.data());
            e.ApplyAdd(x, -0.75, &y);
            dump_vec("spmv_ell_add", y);
            int*    ec = NULL;
            double* ev = NULL;
            int     w  = 0;
            e.LeaveDataPtrELL(&ec, &ev, w);
            dump("ell_width", &w, 1);
            dump("ell_col", ec, (size_t)w * A.n);
            dump("ell_val", ev, (size_t)w * A.n);
            delete[] ec;
            delete[] ev;
        }
    }
    {
        MatD d;
        d.CloneFrom(mat);
        d.ConvertToDIA(); // refused (too many diagonals): falls back to CSR
        int fmt = d.GetFormat();
        dump("dia_format", &fmt, 1);
        if(fmt == DIA)
        {
            d.Apply(x, &y);
            dump_vec("spmv_dia", y);
            y.CopyFromHostData(yin.data());
            d.ApplyAdd(x, -0.75, &y);
            dump_vec("spmv_dia_add", y);
            MatD back;
            back.CloneFrom(d);
            back.ConvertToCSR(); // DIA -> CSR drops the padded zeros (and stored zeros)
            int*    rp = NULL;
            int*    ci = NULL;
            double* va = NULL;
            int64_t bn = back.GetNnz();
            back.LeaveDataPtrCSR(&rp, &ci, &va);
            dump("dia_back_rowptr", rp, (size_t)A.n + 1);
            dump("dia_back_col", ci, (size_t)bn);
            dump("dia_back_val", va, (size_t)bn);
            delete[] rp;
            delete[] ci;
            delete[] va;
            int*    off = NULL;
            double* dv  = NULL;
            int     nd  = 0;
            d.LeaveDataPtrDIA(&off, &dv, nd);
            dump("dia_offset", off, (size_t)nd);
            dump("dia_val", dv, (size_t)nd * A.n);
            delete[] off;
            delete[] dv;
        }
    }
    {
        MatD h;
        h.CloneFrom(mat);
        h.ConvertToHYB();
        h.Apply(x, &y);
        dump_vec("spmv_hyb", y);
        y.CopyFromHostData(yin.data());
        h.ApplyAdd(x, -0.75, &y);
        dump_vec("spmv_hyb_add", y);
    }
    {
        MatD c;
        c.CloneFrom(mat);
        c.ConvertToCOO();
        c.Apply(x, &y);
        dump_vec("spmv_coo", y);
        y.CopyFromHostData(yin.data());
        c.ApplyAdd(x, -0.75, &y);
        dump_vec("spmv_coo_add", y);
    }

    // --- CSR matrix algebra: Transpose, MatrixMult, MatrixAdd (subset and union patterns) ---------
    {
        auto dump_csr = [&](const std::string& tag, MatD& m) {
            int*    rp = NULL;
            int*    ci = NULL;
            double* va = NULL;
            int64_t nz = m.GetNnz();
            int     nr = (int)m.GetM();
            m.LeaveDataPtrCSR(&rp, &ci, &va);
            dump(tag + "_rowptr", rp, (size_t)nr + 1);
            dump(tag + "_col", ci, (size_t)nz);
            dump(tag + "_val", va, (size_t)nz);
            delete[] rp;
            delete[] ci;
            delete[] va;
        };
        MatD t;
        t.CloneFrom(mat);
        t.Transpose();
        MatD aa;
        aa.MatrixMult(mat, t); // A * A^T
        MatD sub;
        sub.CloneFrom(aa);
        sub.MatrixAdd(mat, 0.5, -2.0, false); // pattern(A) is a subset of pattern(A A^T) (full diagonal)
        MatD uni;
        uni.CloneFrom(mat);
        uni.MatrixAdd(aa, 1.5, 0.25, true); // union pattern
        dump_csr("alg_transpose", t);
        dump_csr("alg_matmult", aa);
        dump_csr("alg_add_subset", sub);
        dump_csr("alg_add_union", uni);
    }

    // --- BLAS-1 ----------------------------------------------------------------------------
    {
        VecD a, b;
        a.Allocate("a", A.n);
        b.Allocate("b", A.n);
        a.CopyFromHostData(xin.data());
        b.CopyFromHostData(yin.data());
        double sc[3] = {a.Dot(b), a.DotNonConj(b), a.Norm()};
        dump("blas_scalars", sc, 3);
        tmp.CopyFrom(a);
        tmp.AddScale(b, 0.375);
        dump_vec("blas_add_scale", tmp);
        tmp.CopyFrom(a);
        tmp.ScaleAdd(-1.25, b);
        dump_vec("blas_scale_add", tmp);
        tmp.CopyFrom(a);
        tmp.ScaleAdd2(0.3, b, -1.7, rhs, 0.11);
        dump_vec("blas_scale_add2", tmp);
        tmp.CopyFrom(a);
        tmp.Scale(1.0 / 3.0);
        dump_vec("blas_scale", tmp);
        tmp.PointWiseMult(a, b);
        dump_vec("blas_pointwise", tmp);
    }

    // --- diagonal, ILU(0), triangular solves -----------------------------------------------
    {
        VecD d;
        mat.ExtractInverseDiagonal(&d);
        dump_vec("inv_diag", d);
    }
    {
        MatD lu;
        lu.CloneFrom(mat);
        lu.ILU0Factorize();
        dump_csr("ilu0", lu);
        lu.LUAnalyse();
        lu.LUSolve(x, &y);
        dump_vec("lusolve", y);
        lu.LUAnalyseClear();
    }
    {
        MatD t;
        t.CloneFrom(mat);
        t.LAnalyse(false);
        t.LSolve(x, &y);
        dump_vec("lsolve_nonunit", y);
        t.LAnalyseClear();
        t.UAnalyse(false);
        t.USolve(x, &y);
        dump_vec("usolve_nonunit", y);
        t.UAnalyseClear();
    }

    // --- multi-colouring, permutation -------------------------------------------------------
    {
        int              nc    = 0;
        int*             sizes = NULL;
        LocalVector<int> perm;
        mat.MultiColoring(nc, &sizes, &perm);
        dump("mc_num_colors", &nc, 1);
        dump("mc_sizes", sizes, nc);
        std::vector<int> hp(A.n);
        perm.CopyToHostData(hp.data());
        dump("mc_perm", hp.data(), hp.size());
        MatD pm;
        pm.CloneFrom(mat);
        pm.Permute(perm);
        dump_csr("permuted", pm);
        tmp.CopyFromPermute(x, perm);
        dump_vec("vec_permute", tmp);
        tmp.CopyFromPermuteBackward(x, perm);
        dump_vec("vec_permute_backward", tmp);
        free_host(&sizes);
    }

    // --- preconditioner applies -------------------------------------------------------------
    {
        Jacobi<MatD, VecD, double> p;
        p.SetOperator(mat);
        p.Build();
        p.Solve(x, &y);
        dump_vec("pc_jacobi", y);
        p.Clear();
    }
    {
        ILU<MatD, VecD, double> p;
        p.SetOperator(mat);
        p.Build();
        p.Solve(x, &y);
        dump_vec("pc_ilu0", y);
        p.Clear();
    }
    // ILU(p): fill levels on the pattern of A^(p+1) (p = 1, 2), and ILU(0) on that whole pattern (level = false)
    {
        const int  ps[3]  = {1, 2, 1};
        const bool lv[3]  = {true, true, false};
        const char* nm[3] = {"ilu1", "ilu2", "ilu1n"};
        for(int k = 0; k < 3; ++k)
        {
            MatD lu;
            lu.CloneFrom(mat);
            lu.ILUpFactorize(ps[k], lv[k]);
            dump_csr(nm[k], lu);
        }
        ILU<MatD, VecD, double> p;
        p.Set(1);
        p.SetOperator(mat);
        p.Build();
        p.Solve(x, &y);
        dump_vec("pc_ilu1", y);
        p.Clear();
    }
    {
        MultiColoredSGS<MatD, VecD, double> p;
        p.SetOperator(mat);
        p.Build();
        p.Solve(x, &y);
        dump_vec("pc_mcsgs", y);
        p.Clear();
    }
    {
        MultiColoredGS<MatD, VecD, double> p;
        p.SetOperator(mat);
        p.Build();
        p.Solve(x, &y);
        dump_vec("pc_mcgs", y);
        p.Clear();
    }
    {
        GS<MatD, VecD, double> p;
        p.SetOperator(mat);
        p.Build();
        p.Solve(x, &y);
        dump_vec("pc_gs", y);
        p.Clear();
    }
    if(symmetric_spd) // incomplete Cholesky needs an SPD operator
    {
        IC<MatD, VecD, double> p;
        p.SetOperator(mat);
        p.Build();
        p.Solve(x, &y);
        dump_vec("pc_ic", y);
        p.Clear();
    }
    {
        SGS<MatD, VecD, double> p;
        p.SetOperator(mat);
        p.Build();
        p.Solve(x, &y);
        dump_vec("pc_sgs", y);
        p.Clear();
    }
    // --- TriSolverAlg_Iterative: Jacobi-sweep triangular solves; applied twice (the second apply starts from
    //     the first one's output and intermediate vector)
    {
        SolverDescr d;
        d.SetTriSolverAlg(TriSolverAlg_Iterative); // defaults: 30 sweeps, tolerance 1e-3 on
        ILU<MatD, VecD, double> p;
        p.SetSolverDescriptor(d);
        p.SetOperator(mat);
        p.Build();
        y.Zeros();
        p.Solve(x, &y);
        dump_vec("pc_itilu0", y);
        p.Solve(x, &y);
        dump_vec("pc_itilu0_2", y);
        p.Clear();
    }
    {
        SolverDescr d;
        d.SetTriSolverAlg(TriSolverAlg_Iterative);
        d.SetIterativeSolverMaxIteration(12);
        d.SetIterativeSolverTolerance(1e-2);
        SGS<MatD, VecD, double> p;
        p.SetSolverDescriptor(d);
        p.SetOperator(mat);
        p.Build();
        y.Zeros();
        p.Solve(x, &y);
        dump_vec("pc_itsgs", y);
        p.Solve(x, &y);
        dump_vec("pc_itsgs_2", y);
        p.Clear();
    }
    {
        SolverDescr d;
        d.SetTriSolverAlg(TriSolverAlg_Iterative);
        d.SetIterativeSolverMaxIteration(5);
        d.DisableIterativeSolverTolerance();
        GS<MatD, VecD, double> p;
        p.SetSolverDescriptor(d);
        p.SetOperator(mat);
        p.Build();
        y.Zeros();
        p.Solve(x, &y);
        dump_vec("pc_itgs", y);
        p.Clear();
    }
    if(symmetric_spd)
    {
        SolverDescr d;
        d.SetTriSolverAlg(TriSolverAlg_Iterative);
        d.SetIterativeSolverMaxIteration(8);
        d.DisableIterativeSolverTolerance();
        IC<MatD, VecD, double> p;
        p.SetSolverDescriptor(d);
        p.SetOperator(mat);
        p.Build();
        y.Zeros();
        p.Solve(x, &y);
        dump_vec("pc_itic", y);
        p.Solve(x, &y);
        dump_vec("pc_itic_2", y);
        p.Clear();
    }
    // --- approximate-inverse preconditioners built with matrix algebra (preconditioner_ai.cpp)
    {
        AIChebyshev<MatD, VecD, double> p;
        p.Set(3, 0.05, 16.0);
        p.SetOperator(mat);
        p.Build();
        p.Solve(x, &y);
        dump_vec("pc_aicheb", y);
        p.Clear();
    }
    if(symmetric_spd) // the dense sub-systems are factorised without pivoting: SPD operators
    {
        FSAI<MatD, VecD, double> p;
        p.SetOperator(mat);
        p.Build();
        p.Solve(x, &y);
        dump_vec("pc_fsai", y);
        p.Clear();
        MatD G;
        G.CloneFrom(mat);
        G.FSAI(1, NULL);
        dump_csr("fsai_G", G);
        MatD G2; // pattern of A^2
        G2.CloneFrom(mat);
        G2.FSAI(2, NULL);
        dump_csr("fsai2_G", G2);
        MatD pat, G3; // external pattern: the ILU(1) factor's
        pat.CloneFrom(mat);
        pat.ILUpFactorize(1, true);
        G3.CloneFrom(mat);
        G3.FSAI(1, &pat);
        dump_csr("fsai3_G", G3);
    }
    {
        SPAI<MatD, VecD, double> p;
        p.SetOperator(mat);
        p.Build();
        p.Solve(x, &y);
        dump_vec("pc_spai", y);
        p.Clear();
        MatD Ms;
        Ms.CloneFrom(mat);
        Ms.SPAI();
        dump_csr("spai_M", Ms);
    }
    {
        TNS<MatD, VecD, double> p; // implicit (default)
        p.SetOperator(mat);
        p.Build();
        p.Solve(x, &y);
        dump_vec("pc_tns", y);
        p.Clear();
    }
    {
        TNS<MatD, VecD, double> p;
        p.Set(false); // explicit matrix
        p.SetOperator(mat);
        p.Build();
        p.Solve(x, &y);
        dump_vec("pc_tns_expl", y);
        p.Clear();
    }
    {
        MultiColoredILU<MatD, VecD, double> p; // default ILU(0,1)
        p.SetOperator(mat);
        p.Build();
        p.Solve(x, &y);
        dump_vec("pc_mcilu", y);
        p.Clear();
    }

    // --- solvers (rhs = A*1, x0 = 0, default tolerances) -------------------------------------
    if(do_solvers)
    {
        VecD sol;
        sol.Allocate("sol", A.n);
        {
            CG<MatD, VecD, double> ls;
            ls.SetOperator(mat);
            ls.Build();
            sol.Zeros();
            run_solver("cg_none", ls, rhs, sol);
            ls.Clear();
        }
        {
            CG<MatD, VecD, double>     ls;
            Jacobi<MatD, VecD, double> p;
            ls.SetOperator(mat);
            ls.SetPreconditioner(p);
            ls.Build();
            sol.Zeros();
            run_solver("cg_jacobi", ls, rhs, sol);
            ls.Clear();
        }
        {
            // x0 = x (seeded random), tighter tolerance, like clients/include/testing_cg.hpp
            CG<MatD, VecD, double>     ls;
            Jacobi<MatD, VecD, double> p;
            ls.SetOperator(mat);
            ls.SetPreconditioner(p);
            ls.Init(1e-8, 0.0, 1e8, 10000);
            ls.Build();
            sol.CopyFrom(x);
            run_solver("cg_jacobi_x0", ls, rhs, sol);
            ls.Clear();
        }
        {
            GMRES<MatD, VecD, double> ls;
            ls.SetOperator(mat);
            ls.SetBasisSize(basis);
            ls.Build();
            sol.Zeros();
            run_solver("gmres_none", ls, rhs, sol);
            ls.Clear();
        }
        {
            GMRES<MatD, VecD, double> ls;
            ILU<MatD, VecD, double>   p;
            ls.SetOperator(mat);
            ls.SetPreconditioner(p);
            ls.SetBasisSize(basis);
            ls.Build();
            sol.Zeros();
            run_solver("gmres_ilu0", ls, rhs, sol);
            ls.Clear();
        }
        {
            GMRES<MatD, VecD, double> ls;
            ILU<MatD, VecD, double>   p;
            p.Set(1);
            ls.SetOperator(mat);
            ls.SetPreconditioner(p);
            ls.SetBasisSize(basis);
            ls.Build();
            sol.Zeros();
            run_solver("gmres_ilu1", ls, rhs, sol);
            ls.Clear();
        }
        {
            BiCGStab<MatD, VecD, double> ls;
            ls.SetOperator(mat);
            ls.Build();
            sol.Zeros();
            run_solver("bicgstab_none", ls, rhs, sol);
            ls.Clear();
        }
        {
            BiCGStab<MatD, VecD, double>        ls;
            MultiColoredSGS<MatD, VecD, double> p;
            ls.SetOperator(mat);
            ls.SetPreconditioner(p);
            ls.Build();
            sol.Zeros();
            run_solver("bicgstab_mcsgs", ls, rhs, sol);
            ls.Clear();
        }
        {
            BiCGStab<MatD, VecD, double>       ls;
            MultiColoredGS<MatD, VecD, double> p;
            ls.SetOperator(mat);
            ls.SetPreconditioner(p);
            ls.Build();
            sol.Zeros();
            run_solver("bicgstab_mcgs", ls, rhs, sol);
            ls.Clear();
        }
        {
            GMRES<MatD, VecD, double>           ls;
            MultiColoredILU<MatD, VecD, double> p;
            ls.SetOperator(mat);
            ls.SetPreconditioner(p);
            ls.SetBasisSize(basis);
            ls.Build();
            sol.Zeros();
            run_solver("gmres_mcilu", ls, rhs, sol);
            ls.Clear();
        }
        // --- further Krylov drivers, each bare and with a preconditioner
        {
            FCG<MatD, VecD, double> ls;
            ls.SetOperator(mat);
            ls.Build();
            sol.Zeros();
            run_solver("fcg_none", ls, rhs, sol);
            ls.Clear();
        }
        {
            FCG<MatD, VecD, double>    ls;
            Jacobi<MatD, VecD, double> p;
            ls.SetOperator(mat);
            ls.SetPreconditioner(p);
            ls.Build();
            sol.Zeros();
            run_solver("fcg_jacobi", ls, rhs, sol);
            ls.Clear();
        }
        {
            FCG<MatD, VecD, double>             ls;
            MultiColoredSGS<MatD, VecD, double> p;
            ls.SetOperator(mat);
            ls.SetPreconditioner(p);
            ls.Build();
            sol.Zeros();
            run_solver("fcg_mcsgs", ls, rhs, sol);
            ls.Clear();
        }
        {
            CR<MatD, VecD, double> ls;
            ls.SetOperator(mat);
            ls.Build();
            sol.Zeros();
            run_solver("cr_none", ls, rhs, sol);
            ls.Clear();
        }
        {
            CR<MatD, VecD, double>     ls;
            Jacobi<MatD, VecD, double> p;
            ls.SetOperator(mat);
            ls.SetPreconditioner(p);
            ls.Build();
            sol.Zeros();
            run_solver("cr_jacobi", ls, rhs, sol);
            ls.Clear();
        }
        {
            FGMRES<MatD, VecD, double> ls;
            ls.SetOperator(mat);
            ls.SetBasisSize(basis);
            ls.Build();
            sol.Zeros();
            run_solver("fgmres_none", ls, rhs, sol);
            ls.Clear();
        }
        {
            FGMRES<MatD, VecD, double> ls;
            ILU<MatD, VecD, double>    p;
            ls.SetOperator(mat);
            ls.SetPreconditioner(p);
            ls.SetBasisSize(basis);
            ls.Build();
            sol.Zeros();
            run_solver("fgmres_ilu0", ls, rhs, sol);
            ls.Clear();
        }
        {
            BiCGStabl<MatD, VecD, double> ls; // default l = 2
            ls.SetOperator(mat);
            ls.Build();
            sol.Zeros();
            run_solver("bicgstabl_none", ls, rhs, sol);
            ls.Clear();
        }
        {
            BiCGStabl<MatD, VecD, double> ls;
            Jacobi<MatD, VecD, double>    p;
            ls.SetOperator(mat);
            ls.SetPreconditioner(p);
            ls.SetOrder(3);
            ls.Build();
            sol.Zeros();
            run_solver("bicgstabl3_jacobi", ls, rhs, sol);
            ls.Clear();
        }
        {
            QMRCGStab<MatD, VecD, double> ls;
            ls.SetOperator(mat);
            ls.Build();
            sol.Zeros();
            run_solver("qmrcgstab_none", ls, rhs, sol);
            ls.Clear();
        }
        {
            QMRCGStab<MatD, VecD, double>       ls;
            MultiColoredSGS<MatD, VecD, double> p;
            ls.SetOperator(mat);
            ls.SetPreconditioner(p);
            ls.Build();
            sol.Zeros();
            run_solver("qmrcgstab_mcsgs", ls, rhs, sol);
            ls.Clear();
        }
        {
            CG<MatD, VecD, double>  ls;
            SGS<MatD, VecD, double> p;
            ls.SetOperator(mat);
            ls.SetPreconditioner(p);
            ls.Build();
            sol.Zeros();
            run_solver("cg_sgs", ls, rhs, sol);
            ls.Clear();
        }
        {
            BiCGStab<MatD, VecD, double> ls;
            GS<MatD, VecD, double>       p;
            ls.SetOperator(mat);
            ls.SetPreconditioner(p);
            ls.Build();
            sol.Zeros();
            run_solver("bicgstab_gs", ls, rhs, sol);
            ls.Clear();
        }
        {
            FixedPoint<MatD, VecD, double> ls; // damped Jacobi iteration, capped
            Jacobi<MatD, VecD, double>     p;
            ls.SetOperator(mat);
            ls.SetPreconditioner(p);
            ls.SetRelaxation(0.8);
            ls.InitMaxIter(40);
            ls.Build();
            sol.Zeros();
            run_solver("fixedpoint_jacobi", ls, rhs, sol);
            ls.Clear();
        }
        {
            FixedPoint<MatD, VecD, double>      ls; // as a smoother: 3 steps, no norms
            MultiColoredSGS<MatD, VecD, double> p;
            ls.SetOperator(mat);
            ls.SetPreconditioner(p);
            ls.FlagSmoother();
            ls.InitMaxIter(3);
            ls.Build();
            sol.Zeros();
            run_solver("fixedpoint_smoother_mcsgs", ls, rhs, sol);
            ls.Clear();
        }
        {
            Chebyshev<MatD, VecD, double> ls; // Gershgorin-style bounds of the test operators: (0.05, 16)
            ls.SetOperator(mat);
            ls.Set(0.05, 16.0);
            ls.InitMaxIter(60);
            ls.Build();
            sol.Zeros();
            run_solver("chebyshev_none", ls, rhs, sol);
            ls.Clear();
        }
        {
            Chebyshev<MatD, VecD, double> ls;
            Jacobi<MatD, VecD, double>    p;
            ls.SetOperator(mat);
            ls.SetPreconditioner(p);
            ls.Set(0.01, 2.0);
            ls.InitMaxIter(60);
            ls.Build();
            sol.Zeros();
            run_solver("chebyshev_jacobi", ls, rhs, sol);
            ls.Clear();
        }
        {
            CG<MatD, VecD, double> ls;
            IC<MatD, VecD, double> p;
            ls.SetOperator(mat);
            ls.SetPreconditioner(p);
            ls.Build();
            sol.Zeros();
            run_solver("cg_ic", ls, rhs, sol);
            ls.Clear();
        }
        {
            SolverDescr d;
            d.SetTriSolverAlg(TriSolverAlg_Iterative);
            d.SetIterativeSolverMaxIteration(20);
            d.SetIterativeSolverTolerance(1e-6);
            GMRES<MatD, VecD, double> ls;
            ILU<MatD, VecD, double>   p;
            p.SetSolverDescriptor(d);
            ls.SetOperator(mat);
            ls.SetPreconditioner(p);
            ls.SetBasisSize(basis);
            ls.InitMaxIter(300); // the sweeps need not converge on every case (gr_30_30 stagnates)
            ls.Build();
            sol.Zeros();
            run_solver("gmres_itilu0", ls, rhs, sol);
            ls.Clear();
        }
        {
            SolverDescr d;
            d.SetTriSolverAlg(TriSolverAlg_Iterative);
            d.SetIterativeSolverMaxIteration(10);
            d.DisableIterativeSolverTolerance();
            CG<MatD, VecD, double> ls;
            IC<MatD, VecD, double> p;
            p.SetSolverDescriptor(d);
            ls.SetOperator(mat);
            ls.SetPreconditioner(p);
            ls.InitMaxIter(300);
            ls.Build();
            sol.Zeros();
            run_solver("cg_itic", ls, rhs, sol);
            ls.Clear();
        }
        {
            // a different preconditioner on every application, round robin (flexible GMRES)
            FGMRES<MatD, VecD, double>                 ls;
            VariablePreconditioner<MatD, VecD, double> vp;
            Jacobi<MatD, VecD, double>                 v0;
            MultiColoredSGS<MatD, VecD, double>        v1;
            ILU<MatD, VecD, double>                    v2;
            Solver<MatD, VecD, double>*                list[3] = {&v0, &v1, &v2};
            vp.SetPreconditioner(3, list);
            ls.SetOperator(mat);
            ls.SetPreconditioner(vp);
            ls.SetBasisSize(basis);
            ls.Build();
            sol.Zeros();
            run_solver("fgmres_variable", ls, rhs, sol);
            ls.Clear();
        }
        for(int variant = 0; variant < 2; ++variant)
        {
            // block preconditioner: 3 row blocks (n/3, n/3, rest), ILU(0) per diagonal block; block forward substitution
            // (variant 0) or block-diagonal solve (variant 1)
            GMRES<MatD, VecD, double>               ls;
            BlockPreconditioner<MatD, VecD, double> bp;
            ILU<MatD, VecD, double>                 loc[3];
            Solver<MatD, VecD, double>*             list[3] = {&loc[0], &loc[1], &loc[2]};
            const int nn    = (int)A.n;
            const int sz[3] = {nn / 3, nn / 3, nn - 2 * (nn / 3)};
            bp.Set(3, sz, list);
            if(variant == 1)
                bp.SetDiagonalSolver();
            ls.SetOperator(mat);
            ls.SetPreconditioner(bp);
            ls.SetBasisSize(basis);
            ls.Build();
            sol.Zeros();
            run_solver(variant == 0 ? "gmres_block" : "gmres_blockdiag", ls, rhs, sol);
            ls.Clear();
        }
        for(int variant = 0; variant < 2; ++variant)
        {
            // (restricted) additive Schwarz: 3 blocks, overlap 4, ILU(0) on every block
            GMRES<MatD, VecD, double> ls;
            AS<MatD, VecD, double>    as;
            RAS<MatD, VecD, double>   ras;
            ILU<MatD, VecD, double>   loc[3];
            Solver<MatD, VecD, double>* list[3] = {&loc[0], &loc[1], &loc[2]};
            if(variant == 0)
                as.Set(3, 4, list);
            else
                ras.Set(3, 4, list);
            ls.SetOperator(mat);
            if(variant == 0)
                ls.SetPreconditioner(as);
            else
                ls.SetPreconditioner(ras);
            ls.SetBasisSize(basis);
            ls.Build();
            sol.Zeros();
            run_solver(variant == 0 ? "gmres_as" : "gmres_ras", ls, rhs, sol);
            ls.Clear();
        }
        {
            BiCGStab<MatD, VecD, double> ls;
            SPAI<MatD, VecD, double>     p;
            ls.SetOperator(mat);
            ls.SetPreconditioner(p);
            ls.Build();
            sol.Zeros();
            run_solver("bicgstab_spai", ls, rhs, sol);
            ls.Clear();
        }
        {
            CG<MatD, VecD, double>   ls;
            FSAI<MatD, VecD, double> p;
            ls.SetOperator(mat);
            ls.SetPreconditioner(p);
            ls.Build();
            sol.Zeros();
            run_solver("cg_fsai", ls, rhs, sol);
            ls.Clear();
        }
        {
            CG<MatD, VecD, double>  ls;
            TNS<MatD, VecD, double> p;
            ls.SetOperator(mat);
            ls.SetPreconditioner(p);
            ls.Build();
            sol.Zeros();
            run_solver("cg_tns", ls, rhs, sol);
            ls.Clear();
        }
        {
            CG<MatD, VecD, double>          ls;
            AIChebyshev<MatD, VecD, double> p;
            p.Set(3, 0.05, 16.0);
            ls.SetOperator(mat);
            ls.SetPreconditioner(p);
            ls.InitMaxIter(300);
            ls.Build();
            sol.Zeros();
            run_solver("cg_aicheb", ls, rhs, sol);
            ls.Clear();
        }
        {
            IDR<MatD, VecD, double> ls; // default s = 4; the seed must be fixed (default: time(NULL))
            ls.SetOperator(mat);
            ls.SetRandomSeed(12345ULL);
            ls.Build();
            sol.Zeros();
            run_solver("idr_none", ls, rhs, sol);
            ls.Clear();
        }
        {
            IDR<MatD, VecD, double>    ls;
            Jacobi<MatD, VecD, double> p;
            ls.SetOperator(mat);
            ls.SetPreconditioner(p);
            ls.SetShadowSpace(2);
            ls.SetRandomSeed(777ULL);
            ls.Build();
            sol.Zeros();
            run_solver("idr2_jacobi", ls, rhs, sol);
            ls.Clear();
        }
        {
            // operator converted AFTER Build, as the reference tests do (testing_cg.hpp:151-155)
            MatD e;
            e.CloneFrom(mat);
            BiCGStab<MatD, VecD, double>        ls;
            MultiColoredSGS<MatD, VecD, double> p;
            ls.SetOperator(e);
            ls.SetPreconditioner(p);
            ls.Build();
            e.ConvertToELL();
            sol.Zeros();
            run_solver("bicgstab_mcsgs_ell", ls, rhs, sol);
            ls.Clear();
        }
        {
            MatD e;
            e.CloneFrom(mat);
            CG<MatD, VecD, double>     ls;
            Jacobi<MatD, VecD, double> p;
            ls.SetOperator(e);
            ls.SetPreconditioner(p);
            ls.Build();
            e.ConvertToHYB();
            sol.Zeros();
            run_solver("cg_jacobi_hyb", ls, rhs, sol);
            ls.Clear();
        }
        {
            MatD e;
            e.CloneFrom(mat);
            CG<MatD, VecD, double>     ls;
            Jacobi<MatD, VecD, double> p;
            ls.SetOperator(e);
            ls.SetPreconditioner(p);
            ls.Build();
            e.ConvertToDIA();
            sol.Zeros();
            run_solver("cg_jacobi_dia", ls, rhs, sol);
            ls.Clear();
        }
        {
            // --- MultiGrid on a 3-level hierarchy built by pairing consecutive rows: P[i, i/2] = 1, R = P^T,
            //     Galerkin coarse operators R A P through MatrixMult.  FixedPoint(0.7)+Jacobi smoothers (2 pre, 1 post),
            //     CG on the coarsest level.  mg_v: V-cycle with scaling (the MultiGrid default) as a solver;
            //     mg_w: W-cycle, no scaling; cg_mgk: CG preconditioned by a K-cycle.
            auto pair_prolong = [](int nf, MatD& P) {
                int     nc = (nf + 1) / 2;
                int*    rp = new int[nf + 1];
                int*    ci = new int[nf];
                double* va = new double[nf];
                for(int i = 0; i < nf; ++i)
                {
                    rp[i] = i;
                    ci[i] = i / 2;
                    va[i] = 1.0;
                }
                rp[nf] = nf;
                P.SetDataPtrCSR(&rp, &ci, &va, "P", nf, nf, nc);
            };
            MatD P0, R0, A1, P1, R1, A2, tmp;
            int  n0 = (int)A.n, n1 = (n0 + 1) / 2;
            pair_prolong(n0, P0);
            P0.Transpose(&R0);
            tmp.MatrixMult(mat, P0);
            A1.MatrixMult(R0, tmp);
            pair_prolong(n1, P1);
            P1.Transpose(&R1);
            tmp.Clear();
            tmp.MatrixMult(A1, P1);
            A2.MatrixMult(R1, tmp);
            dump_csr("mg_A2", A2);
            MatD* ops[2] = {&A1, &A2};
            MatD* res[2] = {&R0, &R1};
            MatD* pro[2] = {&P0, &P1};
            for(int variant = 0; variant < 3; ++variant)
            {
                // heap objects that are never cleared or destroyed: the installed library's BaseMultiGrid::Finalize
                // walks an array only the AMG classes allocate and crashes for a user-built MultiGrid
                MultiGrid<MatD, VecD, double>&  mg     = *new MultiGrid<MatD, VecD, double>;
                FixedPoint<MatD, VecD, double>* fp     = new FixedPoint<MatD, VecD, double>[2];
                Jacobi<MatD, VecD, double>*     jac    = new Jacobi<MatD, VecD, double>[2];
                CG<MatD, VecD, double>&         coarse = *new CG<MatD, VecD, double>;
                IterativeLinearSolver<MatD, VecD, double>** sm = new IterativeLinearSolver<MatD, VecD, double>*[2];
                for(int l = 0; l < 2; ++l)
                {
                    fp[l].SetRelaxation(0.7);
                    fp[l].SetPreconditioner(jac[l]);
                    fp[l].Verbose(0);
                    sm[l] = &fp[l];
                }
                coarse.Verbose(0);
                mg.SetOperator(mat);
                mg.InitLevels(3);
                mg.SetOperatorHierarchy(ops);
                mg.SetRestrictOperator(res);
                mg.SetProlongOperator(pro);
                mg.SetSmoother(sm);
                mg.SetSmootherPreIter(2);
                mg.SetSmootherPostIter(1);
                mg.SetSolver(coarse);
                if(variant == 0)
                {
                    mg.InitMaxIter(40);
                    mg.Build();
                    sol.Zeros();
                    run_solver("mg_v", mg, rhs, sol);
                }
                else if(variant == 1)
                {
                    mg.SetScaling(false);
                    mg.SetCycle(Wcycle);
                    mg.InitMaxIter(40);
                    mg.Build();
                    sol.Zeros();
                    run_solver("mg_w", mg, rhs, sol);
                }
                else
                {
                    mg.SetCycle(Kcycle);
                    mg.Verbose(0);
                    CG<MatD, VecD, double>& ls = *new CG<MatD, VecD, double>;
                    ls.SetOperator(mat);
                    ls.SetPreconditioner(mg);
                    ls.InitMaxIter(60);
                    ls.Build();
                    sol.Zeros();
                    run_solver("cg_mgk", ls, rhs, sol);
                }
            }
        }
        {
            // --- unsmoothed-aggregation AMG, PMIS coarsening: the setup arrays of the first level and two runs
            LocalVector<bool>    conn;
            LocalVector<int64_t> agg, roots;
            mat.AMGPMISAggregate(0.01, &conn, &agg, &roots);
            MatD P;
            mat.AMGUnsmoothedAggregation(agg, roots, &P);
            std::vector<int32_t> ic((size_t)mat.GetNnz()), ia((size_t)A.n), ir((size_t)A.n);
            {
                std::vector<char> hb((size_t)mat.GetNnz()); // LocalVector<bool> -> bytes
                bool*             tmp = new bool[mat.GetNnz()];
                conn.CopyToHostData(tmp);
                for(size_t k = 0; k < ic.size(); ++k)
                    ic[k] = tmp[k] ? 1 : 0;
                delete[] tmp;
                std::vector<int64_t> h64((size_t)A.n);
                agg.CopyToHostData(h64.data());
                for(size_t k = 0; k < ia.size(); ++k)
                    ia[k] = (int32_t)h64[k];
                roots.CopyToHostData(h64.data());
                for(size_t k = 0; k < ir.size(); ++k)
                    ir[k] = (int32_t)h64[k];
            }
            dump("amg_conn", ic.data(), ic.size());
            dump("amg_agg", ia.data(), ia.size());
            dump("amg_roots", ir.data(), ir.size());
            dump_csr("amg_P", P);
            {
                // the default coarsening strategy: the sequential greedy sweep
                LocalVector<bool>    gconn;
                LocalVector<int64_t> gagg, groots;
                mat.AMGGreedyAggregate(0.01, &gconn, &gagg, &groots);
                std::vector<int64_t> h64((size_t)A.n);
                std::vector<int32_t> h32((size_t)A.n);
                gagg.CopyToHostData(h64.data());
                for(size_t k = 0; k < h32.size(); ++k)
                    h32[k] = (int32_t)h64[k];
                dump("amg_gagg", h32.data(), h32.size());
                groots.CopyToHostData(h64.data());
                for(size_t k = 0; k < h32.size(); ++k)
                    h32[k] = (int32_t)h64[k];
                dump("amg_groots", h32.data(), h32.size());
                for(int variant = 0; variant < 2; ++variant)
                {
                    CG<MatD, VecD, double> ls;
                    UAAMG<MatD, VecD, double>& ua = *new UAAMG<MatD, VecD, double>; // default strategy: Greedy
                    SAAMG<MatD, VecD, double>& sa = *new SAAMG<MatD, VecD, double>;
                    ua.SetCoarsestLevel(20);
                    sa.SetCoarsestLevel(20);
                    ua.Verbose(0);
                    sa.Verbose(0);
                    ls.SetOperator(mat);
                    if(variant == 0)
                        ls.SetPreconditioner(ua);
                    else
                        ls.SetPreconditioner(sa);
                    ls.InitMaxIter(100);
                    ls.Build();
                    double lv = (double)(variant == 0 ? ua.GetNumLevels() : sa.GetNumLevels());
                    dump(variant == 0 ? "uaamg_greedy_levels" : "saamg_greedy_levels", &lv, 1);
                    sol.Zeros();
                    run_solver(variant == 0 ? "cg_uaamg_greedy" : "cg_saamg_greedy", ls, rhs, sol);
                    ls.Clear();
                }
            }
            {
                MatD Ps;
                mat.AMGSmoothedAggregation(2.0 / 3.0, conn, agg, roots, &Ps, 0);
                dump_csr("amg_Ps", Ps);
                MatD Ps1;
                mat.AMGSmoothedAggregation(0.5, conn, agg, roots, &Ps1, 1); // SubtractWeakConnections
                dump_csr("amg_Ps1", Ps1);
            }
            {
                // classical (Ruge-Stueben) AMG: PMIS C/F splitting, direct interpolation
                LocalVector<int>  cf;
                LocalVector<bool> S;
                mat.RSPMISCoarsening(0.25f, &cf, &S);
                MatD Prs;
                mat.RSDirectInterpolation(cf, S, &Prs);
                std::vector<int32_t> hcf((size_t)A.n), hs((size_t)mat.GetNnz());
                cf.CopyToHostData(hcf.data());
                bool* tb = new bool[mat.GetNnz()];
                S.CopyToHostData(tb);
                for(size_t k = 0; k < hs.size(); ++k)
                    hs[k] = tb[k] ? 1 : 0;
                delete[] tb;
                dump("rs_cf", hcf.data(), hcf.size());
                dump("rs_S", hs.data(), hs.size());
                dump_csr("rs_P", Prs);
                for(int variant = 0; variant < 2; ++variant)
                {
                    RugeStuebenAMG<MatD, VecD, double>& amg = *new RugeStuebenAMG<MatD, VecD, double>;
                    amg.SetOperator(mat);
                    amg.SetCoarseningStrategy(PMIS);
                    amg.SetCoarsestLevel(20);
                    amg.Verbose(0);
                    if(variant == 0)
                    {
                        amg.InitMaxIter(60);
                        amg.Build();
                        double lv = (double)amg.GetNumLevels();
                        dump("rsamg_levels", &lv, 1);
                        sol.Zeros();
                        run_solver("rsamg_pmis", amg, rhs, sol);
                        amg.Clear();
                    }
                    else
                    {
                        CG<MatD, VecD, double> ls;
                        ls.SetOperator(mat);
                        ls.SetPreconditioner(amg);
                        ls.InitMaxIter(100);
                        ls.Build();
                        sol.Zeros();
                        run_solver("cg_rsamg", ls, rhs, sol);
                        ls.Clear();
                    }
                }
            }
            for(int variant = 0; variant < 2; ++variant)
            {
                SAAMG<MatD, VecD, double>& amg = *new SAAMG<MatD, VecD, double>;
                amg.SetOperator(mat);
                amg.SetCoarseningStrategy(PMIS);
                amg.SetCoarsestLevel(20);
                amg.Verbose(0);
                if(variant == 0)
                {
                    amg.InitMaxIter(60);
                    amg.Build();
                    double lv = (double)amg.GetNumLevels();
                    dump("saamg_levels", &lv, 1);
                    sol.Zeros();
                    run_solver("saamg_pmis", amg, rhs, sol);
                    amg.Clear();
                }
                else
                {
                    CG<MatD, VecD, double> ls;
                    ls.SetOperator(mat);
                    ls.SetPreconditioner(amg);
                    ls.InitMaxIter(100);
                    ls.Build();
                    sol.Zeros();
                    run_solver("cg_saamg", ls, rhs, sol);
                    ls.Clear();
                }
            }
            for(int variant = 0; variant < 2; ++variant)
            {
                UAAMG<MatD, VecD, double>& amg = *new UAAMG<MatD, VecD, double>;
                amg.SetOperator(mat);
                amg.SetCoarseningStrategy(PMIS);
                amg.SetCoarsestLevel(20);
                amg.Verbose(0);
                if(variant == 0)
                {
                    amg.InitMaxIter(60);
                    amg.Build();
                    double lv = (double)amg.GetNumLevels();
                    dump("uaamg_levels", &lv, 1);
                    sol.Zeros();
                    run_solver("uaamg_pmis", amg, rhs, sol);
                    amg.Clear();
                }
                else
                {
                    CG<MatD, VecD, double> ls;
                    ls.SetOperator(mat);
                    ls.SetPreconditioner(amg);
                    ls.InitMaxIter(100);
                    ls.Build();
                    sol.Zeros();
                    run_solver("cg_uaamg", ls, rhs, sol);
                    ls.Clear();
                }
            }
        }
        {
            // mixed precision: fp64 defect correction around fp32 CG+Jacobi with the sample's
            // inner tolerances (clients/samples/mixed-precision.cpp:85)
            MixedPrecisionDC<MatD, VecD, double, MatF, VecF, float> mp;
            CG<MatF, VecF, float>                                     cg;
            Jacobi<MatF, VecF, float>                                 p;
            cg.SetPreconditioner(p);
            cg.Init(1e-5, 1e-2, 1e+20, 100000);
            cg.Verbose(0);
            mp.SetOperator(mat);
            mp.Set(cg);
            mp.Build();
            sol.Zeros();
            run_solver("mixed_cg_jacobi", mp, rhs, sol);
            mp.Clear();
        }
    }

    stop_rocalution();
    return 0;
}

// 3-D 7-point Poisson, natural ordering, ascending columns
static void poisson7(int N, std::vector<int32_t>& rp, std::vector<int32_t>& ci,
                     std::vector<double>& va)
{
    int64_t n = (int64_t)N * N * N;
    rp.resize(n + 1);
    ci.clear();
    va.clear();
    ci.reserve(7 * n);
    va.reserve(7 * n);
    int64_t N2 = (int64_t)N * N;
    rp[0]      = 0;
    for(int z = 0; z < N; ++z)
        for(int y = 0; y < N; ++y)
            for(int x = 0; x < N; ++x)
            {
                int64_t r = ((int64_t)z * N + y) * N + x;
                if(z > 0) { ci.push_back((int32_t)(r - N2)); va.push_back(-1.0); }
                if(y > 0) { ci.push_back((int32_t)(r - N)); va.push_back(-1.0); }
                if(x > 0) { ci.push_back((int32_t)(r - 1)); va.push_back(-1.0); }
                ci.push_back((int32_t)r);
                va.push_back(6.0);
                if(x < N - 1) { ci.push_back((int32_t)(r + 1)); va.push_back(-1.0); }
                if(y < N - 1) { ci.push_back((int32_t)(r + N)); va.push_back(-1.0); }
                if(z < N - 1) { ci.push_back((int32_t)(r + N2)); va.push_back(-1.0); }
                rp[r + 1] = (int32_t)ci.size();
            }
}

// the reference's own 3-D operator: 27-point stencil, 26 on the diagonal, -1 elsewhere, ascending columns
// (behaviour of clients/include/utility.hpp:110-177 gen_3d_laplacian; rows filled in order here)
static void laplace27(int N, std::vector<int32_t>& rp, std::vector<int32_t>& ci, std::vector<double>& va)
{
    int64_t n = (int64_t)N * N * N;
    rp.resize(n + 1);
    ci.clear();
    va.clear();
    ci.reserve(27 * n);
    va.reserve(27 * n);
    rp[0] = 0;
    for(int z = 0; z < N; ++z)
        for(int y = 0; y < N; ++y)
            for(int x = 0; x < N; ++x)
            {
                int64_t r = ((int64_t)z * N + y) * N + x;
                for(int sz = -1; sz <= 1; ++sz)
                    for(int sy = -1; sy <= 1; ++sy)
                        for(int sx = -1; sx <= 1; ++sx)
                            if(z + sz >= 0 && z + sz < N && y + sy >= 0 && y + sy < N && x + sx >= 0 && x + sx < N)
                            {
                                int64_t c = r + ((int64_t)sz * N + sy) * N + sx;
                                ci.push_back((int32_t)c);
                                va.push_back(c == r ? 26.0 : -1.0);
                            }
                rp[r + 1] = (int32_t)ci.size();
            }
}

static int cmd_bench(int argc, char** argv)
{
    int         N       = atoi(argv[2]);
    int         iters   = atoi(argv[3]);
    int         threads = atoi(argv[4]);
    int         accel   = atoi(argv[5]);
    std::string solver  = argc > 6 ? argv[6] : "cg";
    std::string precond = argc > 7 ? argv[7] : "jacobi";

    if(!accel)
        disable_accelerator_rocalution(true);
    init_rocalution();
    if(threads > 0)
        set_omp_threads_rocalution(threads);

    // argv[2]: grid edge of the synthetic 7-point operator, or the path of a MatrixMarket file (config 3's input)
    const std::string src(argv[2]);
    const bool        from_file = src.size() > 4 && src.substr(src.size() - 4) == ".mtx";
    int64_t n = 0, nnz = 0;
    MatD    mat;
    double  t_read = 0.0;
    if(from_file)
    {
        double tr0 = rocalution_time();
        mat.ReadFileMTX(src);
        t_read = (rocalution_time() - tr0) / 1e6;
        n      = mat.GetM();
        nnz    = mat.GetNnz();
        N      = 0;
    }
    else
    {
        std::vector<int32_t> rp, ci;
        std::vector<double>  va;
        if(src.rfind("lap27:", 0) == 0) // "lap27:<N>": the 27-point operator instead of the 7-point one
        {
            N = atoi(src.c_str() + 6);
            laplace27(N, rp, ci, va);
        }
        else
            poisson7(N, rp, ci, va);
        n   = (int64_t)N * N * N;
        nnz = (int64_t)ci.size();
        mat.AllocateCSR("poisson7", nnz, n, n);
        mat.CopyFromCSR(rp.data(), ci.data(), va.data());
    }
    VecD x, rhs, e;
    if(accel)
    {
        mat.MoveToAccelerator();
        x.MoveToAccelerator();
        rhs.MoveToAccelerator();
        e.MoveToAccelerator();
    }
    x.Allocate("x", n);
    rhs.Allocate("rhs", n);
    e.Allocate("e", n);
    e.Ones();
    mat.Apply(e, &rhs);
    x.Zeros();

    // SpMV micro-benchmark (clients/samples/benchmark.cpp method: timed reps + sync)
    int    reps = accel ? 200 : 20;
    for(int i = 0; i < 3; ++i)
        mat.Apply(e, &x);
    _rocalution_sync();
    double t0 = rocalution_time();
    for(int i = 0; i < reps; ++i)
        mat.Apply(e, &x);
    _rocalution_sync();
    double t_spmv = (rocalution_time() - t0) / 1e6 / reps;
    x.Zeros();

    IterativeLinearSolver<MatD, VecD, double>* ls = NULL;
    CG<MatD, VecD, double>                     cg;
    GMRES<MatD, VecD, double>                  gm;
    BiCGStab<MatD, VecD, double>               bi;
    if(solver == "gmres")
    {
        gm.SetBasisSize(30);
        ls = &gm;
    }
    else if(solver == "bicgstab")
        ls = &bi;
    else
        ls = &cg;
    Jacobi<MatD, VecD, double>          pj;
    ILU<MatD, VecD, double>             pi;
    MultiColoredSGS<MatD, VecD, double> ps;
    ls->SetOperator(mat);
    if(precond == "jacobi")
        ls->SetPreconditioner(pj);
    else if(precond == "ilu0")
        ls->SetPreconditioner(pi);
    else if(precond == "mcsgs")
        ls->SetPreconditioner(ps);
    // exactly `iters` iterations: tolerances that can never be met
    ls->Init(0.0, 0.0, 1e300, iters);
    ls->Verbose(0);
    double tb0 = rocalution_time();
    ls->Build();
    _rocalution_sync();
    double t_build = (rocalution_time() - tb0) / 1e6;
    // untimed warm-up solve (first-use costs of the vendor libraries: kernel loading, lazy allocations) --
    // the same treatment bench.py gives the own backend (W warm-up steps)
    ls->InitMaxIter(accel ? 20 : (iters < 3 ? iters : 3));
    ls->Solve(rhs, &x);
    ls->InitMaxIter(iters);
    x.Zeros();
    _rocalution_sync();
    double ts0 = rocalution_time();
    ls->Solve(rhs, &x);
    _rocalution_sync();
    double t_solve = (rocalution_time() - ts0) / 1e6;
    int    it      = ls->GetIterationCount();
    double bytes   = 4.0 * (n + nnz) + 8.0 * (2.0 * n + nnz);
    printf("{\"ref_probe\":\"bench\",\"N\":%d,\"n\":%lld,\"nnz\":%lld,\"accel\":%d,\"threads\":%d,"
           "\"solver\":\"%s\",\"precond\":\"%s\",\"iters\":%d,\"t_build_s\":%.6f,\"t_solve_s\":%.6f,"
           "\"iters_per_s\":%.4f,\"t_spmv_s\":%.9f,\"spmv_GBps\":%.3f,\"final_res\":%.17g,\"t_read_s\":%.3f}\n",
           N, (long long)n, (long long)nnz, accel, threads, solver.c_str(), precond.c_str(), it,
           t_build, t_solve, it / t_solve, t_spmv, bytes / t_spmv / 1e9, ls->GetCurrentResidual(), t_read);
    ls->Clear();
    stop_rocalution();
    return 0;
}

// file-format fixtures: the genuine library writes a matrix / a vector in its four formats and
// reads MatrixMarket files back (symmetric / pattern storage) -- outputs pin the own IO layer
static int cmd_io(const std::string& in, const std::string& out)
{
    g_out = out;
    disable_accelerator_rocalution(true);
    init_rocalution();
    set_omp_threads_rocalution(1);
    std::vector<int64_t> hdr = slurp<int64_t>(in + "/hdr.bin");
    Csr                  A;
    A.n   = hdr[0];
    A.nnz = hdr[1];
    A.rp  = slurp<int32_t>(in + "/rowptr.bin");
    A.ci  = slurp<int32_t>(in + "/col.bin");
    A.va  = slurp<double>(in + "/val.bin");
    MatD m;
    load_into(A, m, "A");
    m.WriteFileMTX(out + "/A.mtx");
    m.WriteFileCSR(out + "/A.csr");
    std::vector<double> xv = slurp<double>(in + "/x.bin");
    VecD                x;
    x.Allocate("x", (int64_t)xv.size());
    x.CopyFromData(xv.data());
    x.WriteFileASCII(out + "/x.dat");
    x.WriteFileBinary(out + "/x.bin");
    const char* names[] = {"sym", "pat", "gen"};
    for(const char* nm : names)
    {
        std::string   f = in + "/" + nm + ".mtx";
        std::ifstream t(f.c_str());
        if(!t.good())
            continue;
        MatD r;
        r.ReadFileMTX(f);
        dump_csr(std::string("read_") + nm, r);
        int64_t dims[2] = {r.GetM(), r.GetN()};
        dump(std::string("read_") + nm + "_dims", dims, 2);
    }
    stop_rocalution();
    return 0;
}

int main(int argc, char** argv)
{
    if(argc >= 4 && std::string(argv[1]) == "io")
        return cmd_io(argv[2], argv[3]);
    if(argc >= 4 && std::string(argv[1]) == "gen")
        return cmd_gen(argv[2], argv[3]);
    if(argc >= 6 && std::string(argv[1]) == "bench")
        return cmd_bench(argc, argv);
    std::cerr << "usage: ref_probe gen <indir> <outdir> | bench <N> <iters> <threads> <accel>"
              << std::endl;
    return 1;
}
