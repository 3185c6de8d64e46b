// krylov_driver -- command-line driver on the source-compatible C++ layer (include/rocalution): the same
// call sequence a rocALUTION sample uses (clients/samples/{cg,gmres,bicgstab,mixed-precision}.cpp), for any
// solver / preconditioner / format this backend provides.  Plain host C++: build with
//   g++ -O2 -Iinclude samples/krylov_driver.cpp -Lrocalution_amd -lrocalution_amd -Wl,-rpath,$PWD/rocalution_amd
//
//   krylov_driver <matrix.mtx | poisson:N> <solver> [precond] [format] [param]
//     solver : cg fcg cr gmres fgmres bicgstab bicgstabl qmrcgstab idr chebyshev fixedpoint mixed
//     precond: none jacobi gs sgs ilu ilu1 ilu2 ic fsai spai tns as ras block blockdiag variable mcgs mcsgs mcilu          (default jacobi; "mixed" accepts none|jacobi)
//     format : csr ell hyb      (the operator is converted AFTER Build(), as the reference's tests do)
//     param  : restart length (gmres/fgmres), l (bicgstabl), s (idr)
// Prints the reference's solver log and one machine-readable RESULT line.
#include <rocalution/rocalution.hpp>

#include <cstdlib>
#include <iostream>
#include <memory>
#include <string>

using namespace rocalution;
typedef LocalMatrix<double> Mat;
typedef LocalVector<double> Vec;
typedef Solver<Mat, Vec, double> AnySolver;

static int g_nrow = 0; // rows of the operator (block sizes of the block preconditioner)

static std::unique_ptr<AnySolver> make_precond(const std::string& p)
{
    if(p == "jacobi") return std::unique_ptr<AnySolver>(new Jacobi<Mat, Vec, double>);
    if(p == "gs") return std::unique_ptr<AnySolver>(new GS<Mat, Vec, double>);
    if(p == "sgs") return std::unique_ptr<AnySolver>(new SGS<Mat, Vec, double>);
    if(p == "ic") return std::unique_ptr<AnySolver>(new IC<Mat, Vec, double>);
    if(p == "ilu") return std::unique_ptr<AnySolver>(new ILU<Mat, Vec, double>);
    if(p == "ilu1" || p == "ilu2") // ILU(p) with fill levels
    {
        ILU<Mat, Vec, double>* q = new ILU<Mat, Vec, double>;
        q->Set(p == "ilu1" ? 1 : 2);
        return std::unique_ptr<AnySolver>(q);
    }
    if(p == "mcgs") return std::unique_ptr<AnySolver>(new MultiColoredGS<Mat, Vec, double>);
    if(p == "mcsgs") return std::unique_ptr<AnySolver>(new MultiColoredSGS<Mat, Vec, double>);
    if(p == "mcilu") return std::unique_ptr<AnySolver>(new MultiColoredILU<Mat, Vec, double>);
    if(p != "none")
    {
        std::cerr << "unknown preconditioner " << p << std::endl;
        exit(2);
    }
    return std::unique_ptr<AnySolver>();
}

int main(int argc, char* argv[])
{
    if(argc < 3)
    {
        std::cerr << argv[0] << " <matrix.mtx | poisson:N> <solver> [precond] [format] [param]" << std::endl;
        return 1;
    }
    const std::string src = argv[1], sname = argv[2];
    const std::string pname = argc > 3 ? argv[3] : "jacobi", fname = argc > 4 ? argv[4] : "csr";
    const int param = argc > 5 ? atoi(argv[5]) : 0;

    init_rocalution();
    info_rocalution();

    Mat mat;
    Vec x, rhs, e;
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
    mat.Apply(e, &rhs); // rhs = A * 1
    x.Zeros();

    std::unique_ptr<AnySolver>                             pc;
    std::unique_ptr<IterativeLinearSolver<Mat, Vec, double>> ls;
    // mixed precision keeps its fp32 inner solver alive next to the outer one
    CG<LocalMatrix<float>, LocalVector<float>, float>     inner;
    Jacobi<LocalMatrix<float>, LocalVector<float>, float> inner_pc;

    if(sname == "mixed")
    {
        auto* mp = new MixedPrecisionDC<Mat, Vec, double, LocalMatrix<float>, LocalVector<float>, float>;
        if(pname == "jacobi")
            inner.SetPreconditioner(inner_pc);
        inner.Init(1e-5, 1e-2, 1e+20, 100000);
        inner.Verbose(0);
        mp->Set(inner);
        ls.reset(mp);
    }
    else
    {
        if(sname == "cg") ls.reset(new CG<Mat, Vec, double>);
        else if(sname == "fcg") ls.reset(new FCG<Mat, Vec, double>);
        else if(sname == "cr") ls.reset(new CR<Mat, Vec, double>);
        else if(sname == "gmres") { auto* s = new GMRES<Mat, Vec, double>; if(param > 0) s->SetBasisSize(param); ls.reset(s); }
        else if(sname == "fgmres") { auto* s = new FGMRES<Mat, Vec, double>; if(param > 0) s->SetBasisSize(param); ls.reset(s); }
        else if(sname == "bicgstab") ls.reset(new BiCGStab<Mat, Vec, double>);
        else if(sname == "bicgstabl") { auto* s = new BiCGStabl<Mat, Vec, double>; if(param > 0) s->SetOrder(param); ls.reset(s); }
        else if(sname == "qmrcgstab") ls.reset(new QMRCGStab<Mat, Vec, double>);
        else if(sname == "idr") { auto* s = new IDR<Mat, Vec, double>; s->SetRandomSeed(12345ULL); if(param > 0) s->SetShadowSpace(param); ls.reset(s); }
        else if(sname == "fixedpoint") { auto* s = new FixedPoint<Mat, Vec, double>; s->SetRelaxation(0.8); ls.reset(s); }
        else
        {
            std::cerr << "unknown solver " << sname << std::endl;
            return 2;
        }
        g_nrow = (int)mat.GetM();
        pc     = make_precond(pname);
        if(pc)
            ls->SetPreconditioner(*pc);
    }
    ls->SetOperator(mat);
    ls->Build();
    if(fname == "ell")
        mat.ConvertToELL();
    else if(fname == "hyb")
        mat.ConvertToHYB();
    ls->Verbose(1);
    mat.Info();

    double tick = rocalution_time();
    ls->Solve(rhs, &x);
    _rocalution_sync();
    double tack = rocalution_time();
    const int    iters  = ls->GetIterationCount();
    const int    status = ls->GetSolverStatus();
    const double res    = ls->GetCurrentResidual();
    ls->Clear();

    e.ScaleAdd(-1.0, x);
    const double error = e.Norm();
    std::cout << "Solver execution:" << (tack - tick) / 1e6 << " sec" << std::endl;
    std::cout << "||e - x||_2 = " << error << std::endl;
    std::cout.precision(17);
    std::cout << "RESULT solver=" << sname << " precond=" << pname << " format=" << fname << " iters=" << iters
              << " status=" << status << " residual=" << res << " error=" << error
              << " seconds=" << (tack - tick) / 1e6 << std::endl;
    stop_rocalution();
    return 0;
}
