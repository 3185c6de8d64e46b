// distribute_matrix (include/rocalution/distribute.hpp; clients/include/common.hpp:56-431 of the reference) end to end on one
// rank: the call sequence of the reference's cg_mpi sample -- read/build the replicated matrix, distribute_matrix, move to
// the accelerator, CG + BlockJacobi(Jacobi-equivalent here: plain Jacobi) -- through a real RCCL communicator of size 1,
// against the same solve on the LocalMatrix.  Usage: distribute_driver [grid N]
#include <rocalution/rocalution.hpp>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace rocalution;

#define REQUIRE(cond)                                                              \
    do                                                                             \
    {                                                                              \
        if(!(cond))                                                                \
        {                                                                          \
            std::printf("distribute_driver: failed at line %d: %s\n", __LINE__, #cond); \
            return 1;                                                              \
        }                                                                          \
    } while(0)

static void poisson7(int N, std::vector<PtrType>& rp, std::vector<int>& col, std::vector<double>& val)
{
    rp.assign(1, 0);
    for(int k = 0; k < N; ++k)
        for(int j = 0; j < N; ++j)
            for(int i = 0; i < N; ++i)
            {
                const int r = (k * N + j) * N + i;
                if(k > 0) { col.push_back(r - N * N); val.push_back(-1); }
                if(j > 0) { col.push_back(r - N); val.push_back(-1); }
                if(i > 0) { col.push_back(r - 1); val.push_back(-1); }
                col.push_back(r); val.push_back(6);
                if(i < N - 1) { col.push_back(r + 1); val.push_back(-1); }
                if(j < N - 1) { col.push_back(r + N); val.push_back(-1); }
                if(k < N - 1) { col.push_back(r + N * N); val.push_back(-1); }
                rp.push_back((PtrType)col.size());
            }
}

int main(int argc, char** argv)
{
    const int N = argc > 1 ? std::atoi(argv[1]) : 16;
    init_rocalution();
    char id[128];
    REQUIRE(ramd_comm_unique_id(id) == RAMD_OK);
    ramd_comm_t comm = NULL;
    REQUIRE(ramd_comm_init_rccl(0, 1, id, &comm) == RAMD_OK);

    std::vector<PtrType> rp;
    std::vector<int>     col;
    std::vector<double>  val;
    poisson7(N, rp, col, val);
    const int64_t n = (int64_t)rp.size() - 1, nnz = (int64_t)col.size();

    // reference solve on the LocalMatrix
    LocalMatrix<double> lref;
    lref.AllocateCSR("A", nnz, n, n);
    lref.CopyFromCSR(rp.data(), col.data(), val.data());
    LocalVector<double> x, rhs, e;
    x.Allocate("x", n); rhs.Allocate("rhs", n); e.Allocate("e", n);
    lref.MoveToAccelerator(); x.MoveToAccelerator(); rhs.MoveToAccelerator(); e.MoveToAccelerator();
    e.Ones(); lref.Apply(e, &rhs); x.Zeros();
    CG<LocalMatrix<double>, LocalVector<double>, double>     ls;
    Jacobi<LocalMatrix<double>, LocalVector<double>, double> jac;
    ls.SetOperator(lref); ls.SetPreconditioner(jac); ls.Build(); ls.Verbose(0);
    ls.Solve(rhs, &x);
    const int    it_local  = ls.GetIterationCount();
    const double res_local = ls.GetCurrentResidual();
    ls.Clear();

    // the distributed path
    LocalMatrix<double> lmat;
    lmat.AllocateCSR("A", nnz, n, n);
    lmat.CopyFromCSR(rp.data(), col.data(), val.data());
    ParallelManager      pm;
    GlobalMatrix<double> gmat;
    distribute_matrix(comm, &lmat, &gmat, &pm);
    REQUIRE(lmat.GetNnz() == 0); // emptied, as the reference leaves it
    REQUIRE(pm.Status() && pm.GetNumProcs() == 1 && pm.GetLocalNrow() == n && pm.GetGlobalNrow() == n);
    REQUIRE(gmat.GetM() == n && gmat.GetLocalNnz() == nnz && gmat.GetGhostNnz() == 0);
    gmat.MoveToAccelerator();
    GlobalVector<double> gx(pm), grhs(pm), ge(pm);
    gx.Allocate("x", n); grhs.Allocate("rhs", n); ge.Allocate("e", n);
    gx.MoveToAccelerator(); grhs.MoveToAccelerator(); ge.MoveToAccelerator();
    ge.Ones(); gmat.Apply(ge, &grhs); gx.Zeros();
    CG<GlobalMatrix<double>, GlobalVector<double>, double>     gls;
    Jacobi<GlobalMatrix<double>, GlobalVector<double>, double> gjac;
    gls.SetOperator(gmat); gls.SetPreconditioner(gjac); gls.Build(); gls.Verbose(0);
    gls.Solve(grhs, &gx);
    REQUIRE(gls.GetIterationCount() == it_local);
    REQUIRE(std::fabs(gls.GetCurrentResidual() - res_local) <= 1e-12 * std::fabs(res_local));
    ge.ScaleAdd(-1.0, gx);
    REQUIRE(ge.Norm() < 1e-3);
    const int it_global = gls.GetIterationCount();
    gls.Clear();
    std::printf("distribute_driver ok: %d iterations (local %d), residual %.17g\n", it_global, it_local, res_local);
    REQUIRE(ramd_comm_destroy(comm) == RAMD_OK);
    stop_rocalution();
    return 0;
}
