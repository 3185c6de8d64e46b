// tests/drivers/multigrid_driver.cpp -- MultiGrid on a user-built hierarchy, written against include/rocalution the way a
// rocALUTION user writes it (the call sequence of the reference's MultiGrid / UAAMG setup code):
//   transfer operators P (pairs of consecutive rows -> one coarse row) and R = P^T, Galerkin coarse operators
//   R A P through LocalMatrix::MatrixMult, FixedPoint(0.7)+Jacobi smoothers (2 pre, 1 post), CG on the coarsest level.
// Usage: multigrid_driver <matrix.mtx> <variant>    variant: v (V-cycle + scaling, solver)
//                                                            w (W-cycle, no scaling, solver)
//                                                            k (CG preconditioned by a K-cycle)
//                                                            a (UAAMG, PMIS coarsening, as a solver)
//                                                            c (CG preconditioned by UAAMG)
//                                                            s (SAAMG, PMIS coarsening, as a solver)
//                                                            d (CG preconditioned by SAAMG)
//                                                            g / h (as c / d with the default Greedy coarsening)
// Prints one RESULT line and the residual history (HIST lines), which tests/test_gpu_solvers.py compares with the
// genuine library's run of the same setup (oracle/ref_probe).
#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>

#include <rocalution/rocalution.hpp>

using namespace rocalution;
typedef LocalMatrix<double> Mat;
typedef LocalVector<double> Vec;

static void pair_prolong(int nf, Mat& P)
{
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
    P.MoveToAccelerator();
}

int main(int argc, char* argv[])
{
    if(argc < 3)
    {
        std::cerr << argv[0] << " <matrix.mtx> <v|w|k>" << std::endl;
        return 1;
    }
    const std::string variant = argv[2];
    init_rocalution();
    Mat mat;
    Vec x, rhs, e;
    const std::string src = argv[1];
    if(src.compare(0, 8, "poisson:") == 0)
    {
        mat.MoveToAccelerator();
        mat.GeneratePoisson7(atoi(src.c_str() + 8)); // extension: 3-D 7-point operator built on the device
    }
    else
    {
        mat.ReadFileMTX(src);
        mat.MoveToAccelerator();
    }
    x.MoveToAccelerator();
    rhs.MoveToAccelerator();
    e.MoveToAccelerator();
    x.Allocate("x", mat.GetN());
    rhs.Allocate("rhs", mat.GetM());
    e.Allocate("e", mat.GetN());
    e.Ones();
    mat.Apply(e, &rhs);
    x.Zeros();

    if(variant == "a" || variant == "c" || variant == "s" || variant == "d" || variant == "g" || variant == "h")
    {
        // the reference's sample sequence for UAAMG / SAAMG (clients/samples/ua-amg.cpp, sa-amg.cpp), PMIS coarsening
        UAAMG<Mat, Vec, double>    ua;
        SAAMG<Mat, Vec, double>    sa;
        const bool                 smoothed = (variant == "s" || variant == "d" || variant == "h");
        const bool                 greedy   = (variant == "g" || variant == "h");
        BaseAMG<Mat, Vec, double>& amg      = smoothed ? static_cast<BaseAMG<Mat, Vec, double>&>(sa)
                                                       : static_cast<BaseAMG<Mat, Vec, double>&>(ua);
        CG<Mat, Vec, double>       cg;
        amg.SetOperator(mat);
        if(!greedy) // (Greedy is the default of both classes)
        {
            ua.SetCoarseningStrategy(PMIS);
            sa.SetCoarseningStrategy(PMIS);
        }
        amg.SetCoarsestLevel(20);
        amg.Verbose(0);
        IterativeLinearSolver<Mat, Vec, double>* s = &amg;
        if(variant == "a" || variant == "s")
            amg.InitMaxIter(60);
        else
        {
            cg.SetOperator(mat);
            cg.SetPreconditioner(amg);
            cg.InitMaxIter(100);
            s = &cg;
        }
        s->Verbose(0);
        s->RecordResidualHistory();
        if(argc > 3)
        {
            amg.SetCoarsestLevel(atoi(argv[3]));
            s->Init(1e-15, 1e-8, 1e8, 500);
        }
        double t0 = rocalution_time();
        s->Build();
        _rocalution_sync();
        double t1 = rocalution_time();
        const int levels = amg.GetNumLevels();
        s->Solve(rhs, &x);
        _rocalution_sync();
        double t2 = rocalution_time();
        std::cout << "TIMING build_s=" << (t1 - t0) / 1e6 << " solve_s=" << (t2 - t1) / 1e6 << " levels=" << levels
                  << std::endl;
        std::cout.precision(17);
        const std::vector<double> h = s->GetResidualHistory();
        for(size_t i = 0; i < h.size(); ++i)
            std::cout << "HIST " << h[i] << std::endl;
        e.ScaleAdd(-1.0, x);
        std::cout << "RESULT variant=" << variant << " coarse_n=" << levels << " coarse_nnz=0"
                  << " iters=" << s->GetIterationCount() << " status=" << s->GetSolverStatus()
                  << " residual=" << s->GetCurrentResidual() << " error=" << e.Norm() << std::endl;
        s->Clear();
        stop_rocalution();
        return 0;
    }

    Mat P0, R0, A1, P1, R1, A2, tmp;
    const int n0 = (int)mat.GetM(), n1 = (n0 + 1) / 2;
    pair_prolong(n0, P0);
    P0.Transpose(&R0);
    tmp.MatrixMult(mat, P0);
    A1.MatrixMult(R0, tmp);
    pair_prolong(n1, P1);
    P1.Transpose(&R1);
    tmp.Clear();
    tmp.MatrixMult(A1, P1);
    A2.MatrixMult(R1, tmp);
    Mat* ops[2] = {&A1, &A2};
    Mat* res[2] = {&R0, &R1};
    Mat* pro[2] = {&P0, &P1};

    MultiGrid<Mat, Vec, double>                mg;
    FixedPoint<Mat, Vec, double>               fp[2];
    Jacobi<Mat, Vec, double>                   jac[2];
    CG<Mat, Vec, double>                       coarse;
    IterativeLinearSolver<Mat, Vec, double>**  sm = new IterativeLinearSolver<Mat, Vec, double>*[2];
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

    CG<Mat, Vec, double>                     outer;
    IterativeLinearSolver<Mat, Vec, double>* ls = &mg;
    if(variant == "v")
        mg.InitMaxIter(40);
    else if(variant == "w")
    {
        mg.SetScaling(false);
        mg.SetCycle(Wcycle);
        mg.InitMaxIter(40);
    }
    else
    {
        mg.SetCycle(Kcycle);
        mg.Verbose(0);
        outer.SetOperator(mat);
        outer.SetPreconditioner(mg);
        outer.InitMaxIter(60);
        ls = &outer;
    }
    ls->Verbose(0);
    ls->RecordResidualHistory();
    ls->Build();
    ls->Solve(rhs, &x);
    _rocalution_sync();
    const int    iters  = ls->GetIterationCount();
    const int    status = ls->GetSolverStatus();
    const double resid  = ls->GetCurrentResidual();
    std::cout.precision(17);
    const std::vector<double> hist = ls->GetResidualHistory();
    for(size_t i = 0; i < hist.size(); ++i)
        std::cout << "HIST " << hist[i] << std::endl;
    e.ScaleAdd(-1.0, x);
    std::cout << "RESULT variant=" << variant << " coarse_n=" << A2.GetM() << " coarse_nnz=" << A2.GetNnz()
              << " iters=" << iters << " status=" << status << " residual=" << resid << " error=" << e.Norm()
              << std::endl;
    ls->Clear();
    delete[] sm;
    stop_rocalution();
    return 0;
}
