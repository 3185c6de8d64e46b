// rocalution/solvers.hpp -- Solver / IterativeLinearSolver / IterationControl / Preconditioners /
// CG / GMRES / BiCGStab / MixedPrecisionDC with the reference's API surface and control flow:
//   src/solvers/solver.hpp:179-444, solver.cpp:443-512       Solver, IterativeLinearSolver
//   src/solvers/iter_ctrl.cpp:38-345                          IterationControl
//   src/solvers/preconditioners/preconditioner.cpp:66-166     Preconditioner, Jacobi
//   src/solvers/preconditioners/preconditioner.cpp:449-511    ILU
//   src/solvers/preconditioners/preconditioner_multicolored{,_gs}.cpp   MultiColored, MultiColoredSGS
//   src/solvers/krylov/cg.cpp:99-446, gmres.cpp:109-607, bicgstab.cpp:109-489
//   src/solvers/mixed_precision.cpp:159-437                   MixedPrecisionDC
// The numerical sequence of every solver is the reference's (same operations in the same order, same
// stopping rules).  On top of it, CG and GMRES have FUSED device paths (SetFused(true), default on)
// that run the identical arithmetic per element through the single-launch kernels of
// rocalution_amd.h ("fused hot-path ops"): 3 launches and ONE host read-back per CG iteration instead
// of 8 launches and 3 blocking reads, with the read-back hidden behind the next SpMV.
#pragma once

#include <ctime>

#include <cmath>
#include <limits>

#include "base.hpp"

namespace rocalution
{

// ============================================================================ IterationControl
class IterationControl
{
public:
    IterationControl()
    {
        this->Clear();
        this->rec_            = false;
        this->verb_           = 1;
        this->absolute_tol_   = 1e-15;
        this->relative_tol_   = 1e-6;
        this->divergence_tol_ = 1e+8;
        this->minimum_iter_   = 0;
        this->maximum_iter_   = 1000000;
        this->initial_residual_ = 0.0;
    }
    void Clear(void)
    {
        this->residual_history_.clear();
        this->iteration_     = 0;
        this->init_res_      = false;
        this->reached_       = 0;
        this->current_res_   = 0.0;
        this->current_index_ = -1;
    }
    void Init(double abs, double rel, double div, int max)
    {
        this->InitTolerance(abs, rel, div);
        this->InitMaximumIterations(max);
    }
    void Init(double abs, double rel, double div, int min, int max)
    {
        this->InitTolerance(abs, rel, div);
        this->InitMinimumIterations(min);
        this->InitMaximumIterations(max);
    }
    void InitTolerance(double abs, double rel, double div)
    {
        this->absolute_tol_   = abs;
        this->relative_tol_   = rel;
        this->divergence_tol_ = div;
    }
    void InitMinimumIterations(int min)
    {
        assert(min >= 0 && min <= this->maximum_iter_);
        this->minimum_iter_ = min;
    }
    void InitMaximumIterations(int max)
    {
        assert(max >= 0 && max >= this->minimum_iter_);
        this->maximum_iter_ = max;
    }
    int GetMinimumIterations(void) const
    {
        return this->minimum_iter_;
    }
    int GetMaximumIterations(void) const
    {
        return this->maximum_iter_;
    }
    int GetIterationCount(void) const
    {
        return this->iteration_;
    }
    double GetCurrentResidual(void) const
    {
        return this->current_res_;
    }
    int64_t GetAmaxResidualIndex(void) const
    {
        return this->current_index_;
    }
    int GetSolverStatus(void) const
    {
        return this->reached_;
    }
    const std::vector<double>& GetResidualHistory(void) const
    {
        return this->residual_history_;
    }
    void RecordHistory(void)
    {
        this->rec_ = true;
    }
    void Verbose(int verb)
    {
        this->verb_ = verb;
    }
    // iter_ctrl.cpp:89-121
    bool InitResidual(double res)
    {
        this->init_res_         = true;
        this->initial_residual_ = res; // current_res_ is NOT touched here (iter_ctrl.cpp:89-96)
        this->reached_          = 0;
        this->iteration_        = 0;
        if(this->verb_ > 0)
            LOG_INFO("IterationControl initial residual = " << res);
        if(this->rec_)
            this->residual_history_.push_back(res);
        if(this->bad_(res))
        {
            LOG_INFO("Residual = " << res << " !!!");
            return false;
        }
        if(std::abs(res) <= this->absolute_tol_)
        {
            this->reached_ = 1;
            return false;
        }
        return true;
    }
    // iter_ctrl.cpp:195-248
    bool CheckResidual(double res)
    {
        assert(this->init_res_ == true);
        this->iteration_++;
        this->current_res_ = res;
        if(this->verb_ > 1)
            LOG_INFO("IterationControl iter=" << this->iteration_ << "; residual=" << res);
        if(this->rec_)
            this->residual_history_.push_back(res);
        if(this->bad_(res))
        {
            LOG_INFO("Residual = " << res << " !!!");
            return true;
        }
        if(this->iteration_ >= this->minimum_iter_)
        {
            if(std::abs(res) <= this->absolute_tol_)
            {
                this->reached_ = 1;
                return true;
            }
            if(res / this->initial_residual_ <= this->relative_tol_)
            {
                this->reached_ = 2;
                return true;
            }
            if(this->iteration_ >= this->maximum_iter_)
            {
                this->reached_ = 4;
                return true;
            }
        }
        if(res / this->initial_residual_ >= this->divergence_tol_)
        {
            this->reached_ = 3;
            return true;
        }
        return false;
    }
    bool CheckResidual(double res, int64_t index)
    {
        this->current_index_ = index;
        return this->CheckResidual(res);
    }
    // iter_ctrl.cpp:295-306
    bool CheckMaximumIterNoCount(void)
    {
        assert(this->init_res_ == true);
        if(this->iteration_ + 1 >= this->maximum_iter_)
        {
            this->reached_ = 4;
            return true;
        }
        return false;
    }
    // iter_ctrl.cpp:256-289
    bool CheckResidualNoCount(double res)
    {
        assert(this->init_res_ == true);
        if(this->bad_(res))
        {
            LOG_INFO("Residual = " << res << " !!!");
            return true;
        }
        if(std::abs(res) <= this->absolute_tol_)
        {
            this->reached_ = 1;
            return true;
        }
        if(res / this->initial_residual_ <= this->relative_tol_)
        {
            this->reached_ = 2;
            return true;
        }
        if(res / this->initial_residual_ >= this->divergence_tol_)
        {
            this->reached_ = 3;
            return true;
        }
        if(this->iteration_ >= this->maximum_iter_)
        {
            this->reached_ = 4;
            return true;
        }
        return false;
    }
    // iter_ctrl.cpp:317-345: the first `iteration_` entries, scientific notation
    void WriteHistoryToFile(const std::string& filename) const
    {
        std::ofstream file(filename.c_str());
        if(!file.is_open())
        {
            LOG_INFO("Can not open file [write]:" << filename);
            FATAL_ERROR(__FILE__, __LINE__);
        }
        file.setf(std::ios::scientific);
        for(int n = 0; n < this->iteration_ && n < (int)this->residual_history_.size(); n++)
            file << this->residual_history_[n] << std::endl;
    }
    void PrintInit(void) const
    {
        LOG_INFO("IterationControl criteria: abs tol=" << this->absolute_tol_ << "; rel tol="
                                                       << this->relative_tol_ << "; div tol="
                                                       << this->divergence_tol_ << "; max iter="
                                                       << this->maximum_iter_);
    }
    void PrintStatus(void) const
    {
        static const char* why[] = {"NO CRITERIA", "ABSOLUTE criteria", "RELATIVE criteria",
                                    "DIVERGENCE criteria", "MAX ITER criteria"};
        LOG_INFO("IterationControl " << why[this->reached_] << " has been reached: res norm="
                                     << this->current_res_ << "; rel val="
                                     << this->current_res_ / this->initial_residual_
                                     << "; iter=" << this->iteration_);
    }

private:
    static bool bad_(double res)
    {
        return (std::abs(res) == std::numeric_limits<double>::infinity()) || (res != res);
    }
    std::vector<double> residual_history_;
    int                 iteration_;
    bool                init_res_, rec_;
    int                 verb_, reached_;
    double              initial_residual_, current_res_;
    int64_t             current_index_;
    double              absolute_tol_, relative_tol_, divergence_tol_;
    int                 minimum_iter_, maximum_iter_;
};

// ============================================================================ SolverDescr
// solver.hpp:33-148: which triangular-solve algorithm the preconditioners use, and the iterative one's limits
#define DISPATCH_OPERATOR_SOLVE_STRATEGY(descr_, op_, func_, ...)                                             \
    switch(descr_.GetTriSolverAlg())                                                                          \
    {                                                                                                         \
    case TriSolverAlg_Default:                                                                                \
        op_.func_(__VA_ARGS__);                                                                               \
        break;                                                                                                \
    case TriSolverAlg_Iterative:                                                                              \
        op_.It##func_(descr_.GetIterativeSolverMaxIteration(), descr_.GetIterativeSolverTolerance(),          \
                      descr_.GetIterativeSolverUseTolerance(), __VA_ARGS__);                                  \
        break;                                                                                                \
    }
#define DISPATCH_OPERATOR_ANALYSE_STRATEGY(descr_, op_, func_, ...) \
    switch(descr_.GetTriSolverAlg())                                \
    {                                                               \
    case TriSolverAlg_Default:                                      \
        op_.func_(__VA_ARGS__);                                     \
        break;                                                      \
    case TriSolverAlg_Iterative:                                    \
        op_.It##func_(__VA_ARGS__);                                 \
        break;                                                      \
    }

typedef enum _tri_solver_alg : unsigned int
{
    TriSolverAlg_Default   = 0, // level-scheduled direct solve
    TriSolverAlg_Iterative = 1 // Jacobi sweeps
} TriSolverAlg;

class SolverDescr
{
public:
    SolverDescr()
        : tri_solver_alg_(TriSolverAlg_Default)
        , itsolver_max_iter_(30)
        , itsolver_tol_(1e-3)
        , itsolver_use_tol_(true)
    {
    }
    virtual ~SolverDescr() {}
    void SetTriSolverAlg(TriSolverAlg alg)
    {
        this->tri_solver_alg_ = alg;
    }
    TriSolverAlg GetTriSolverAlg(void) const
    {
        return this->tri_solver_alg_;
    }
    void SetIterativeSolverMaxIteration(int max_iter)
    {
        this->itsolver_max_iter_ = max_iter;
    }
    int GetIterativeSolverMaxIteration(void) const
    {
        return this->itsolver_max_iter_;
    }
    void SetIterativeSolverTolerance(double tol)
    {
        this->itsolver_tol_ = tol;
    }
    double GetIterativeSolverTolerance(void) const
    {
        return this->itsolver_tol_;
    }
    void EnableIterativeSolverTolerance(void)
    {
        this->itsolver_use_tol_ = true;
    }
    void DisableIterativeSolverTolerance(void)
    {
        this->itsolver_use_tol_ = false;
    }
    bool GetIterativeSolverUseTolerance(void) const
    {
        return this->itsolver_use_tol_;
    }
    void Print(void) const
    {
        if(this->tri_solver_alg_ != TriSolverAlg_Iterative)
            return; // nothing is printed in the default direct case (solver.cpp:96-115)
        if(this->itsolver_use_tol_)
            LOG_INFO("TriSolverAlg = iterative (" << this->itsolver_max_iter_ << ", " << this->itsolver_tol_ << ")");
        else
            LOG_INFO("TriSolverAlg = iterative (" << this->itsolver_max_iter_ << ")");
    }

protected:
    TriSolverAlg tri_solver_alg_;
    int          itsolver_max_iter_;
    double       itsolver_tol_;
    bool         itsolver_use_tol_;
};

// ============================================================================ Solver
template <class OperatorType, class VectorType, typename ValueType>
class Solver
{
public:
    Solver()
        : op_(NULL)
        , precond_(NULL)
        , build_(false)
        , verb_(1)
        , is_precond_(false)
        , is_smoother_(false)
    {
    }
    virtual ~Solver() {}

    void SetOperator(const OperatorType& op)
    {
        assert(this->build_ == false);
        this->op_ = &op;
    }
    virtual void ResetOperator(const OperatorType& op)
    {
        this->op_ = &op;
    }
    virtual void Print(void) const = 0;
    virtual void Solve(const VectorType& rhs, VectorType* x) = 0;
    virtual void SolveZeroSol(const VectorType& rhs, VectorType* x)
    {
        x->Zeros();
        this->Solve(rhs, x);
    }
    virtual void Build(void)
    {
        if(this->build_)
            this->Clear();
        this->build_ = true;
    }
    virtual void Clear(void)
    {
        this->build_ = false;
    }
    virtual void MoveToHost(void) {}
    virtual void MoveToAccelerator(void) {}
    // Solver::ReBuildNumeric (solver.hpp:214-218): the operator kept its pattern but got new values
    // (UpdateValuesCSR): redo the numerical part.  Here: a full Clear() + Build() -- same result, and every
    // analysis of this backend runs on the device in milliseconds.
    virtual void ReBuildNumeric(void)
    {
        if(this->build_)
        {
            this->Clear();
            this->Build();
        }
    }
    virtual void Verbose(int verb = 1)
    {
        this->verb_ = verb;
    }
    void FlagPrecond(void)
    {
        this->is_precond_ = true;
    }
    void FlagSmoother(void) // solver.hpp:254-258
    {
        this->is_smoother_ = true;
    }
    // true: Solve() runs reductions of its own (nested Krylov solvers, multigrid cycles), i.e. it overwrites the
    // device scalar record -- an outer fused loop that keeps alpha/beta/rho there across the call must not be used
    virtual bool SolveUsesScalarRecord(void) const
    {
        return true;
    }
    // solver.cpp:293-301: the strategy cannot change once the solver is built
    virtual void SetSolverDescriptor(const SolverDescr& descr)
    {
        assert(this->build_ == false);
        this->solver_descr_ = descr;
    }

protected:
    SolverDescr                                  solver_descr_;
    const OperatorType*                          op_;
    Solver<OperatorType, VectorType, ValueType>* precond_;
    bool                                         build_;
    int                                          verb_;
    bool                                         is_precond_;
    bool                                         is_smoother_;
};

// ============================================================================ Preconditioner
template <class OperatorType, class VectorType, typename ValueType>
class Preconditioner : public Solver<OperatorType, VectorType, ValueType>
{
public:
    // preconditioner.cpp:66-72: no zero fill, plain Solve
    virtual void SolveZeroSol(const VectorType& rhs, VectorType* x)
    {
        this->Solve(rhs, x);
    }
    virtual bool SolveUsesScalarRecord(void) const // sweeps / triangular solves / SpMV only
    {
        return false;
    }
};

// ---- Jacobi: preconditioner.cpp:95-166
template <class OperatorType, class VectorType, typename ValueType>
class Jacobi : public Preconditioner<OperatorType, VectorType, ValueType>
{
public:
    virtual ~Jacobi()
    {
        this->Clear();
    }
    virtual void Print(void) const
    {
        LOG_INFO("Jacobi preconditioner");
    }
    virtual void Build(void)
    {
        if(this->build_)
            this->Clear();
        this->build_ = true;
        assert(this->op_ != NULL);
        this->inv_diag_entries_.CloneBackend(*this->op_);
        this->op_->ExtractInverseDiagonal(&this->inv_diag_entries_);
    }
    virtual void Clear(void)
    {
        this->inv_diag_entries_.Clear();
        this->build_ = false;
    }
    virtual void Solve(const VectorType& rhs, VectorType* x)
    {
        assert(this->build_ == true && x != NULL);
        if(this->inv_diag_entries_.GetSize() == 0) // empty inverse diagonal == identity
        {
            if(x != &rhs)
                x->CopyFrom(rhs);
            return;
        }
        if(x != &rhs)
            x->PointWiseMult(this->inv_diag_entries_, rhs);
        else
            x->PointWiseMult(this->inv_diag_entries_);
    }
    const VectorType& GetInverseDiagonal(void) const
    {
        return this->inv_diag_entries_;
    }

private:
    VectorType inv_diag_entries_;
};

// ---- ILU(p = 0): preconditioner.cpp:449-511
template <class OperatorType, class VectorType, typename ValueType>
class ILU : public Preconditioner<OperatorType, VectorType, ValueType>
{
public:
    ILU()
        : p_(0)
        , level_(true)
    {
    }
    virtual ~ILU()
    {
        this->Clear();
    }
    virtual void Print(void) const
    {
        LOG_INFO("ILU(" << this->p_ << ") preconditioner");
    }
    virtual void Set(int p, bool level = true)
    {
        assert(p >= 0 && this->build_ == false);
        this->p_     = p;
        this->level_ = level;
    }
    virtual void Build(void)
    {
        if(this->build_)
            this->Clear();
        this->build_ = true;
        assert(this->op_ != NULL);
        this->ILU_.CloneFrom(*this->op_);
        this->ILU_.ILUpFactorize(this->p_, this->level_);
        DISPATCH_OPERATOR_ANALYSE_STRATEGY(this->solver_descr_, this->ILU_, LUAnalyse);
    }
    virtual void Clear(void)
    {
        DISPATCH_OPERATOR_ANALYSE_STRATEGY(this->solver_descr_, this->ILU_, LUAnalyseClear);
        this->ILU_.Clear();
        this->build_ = false;
    }
    virtual void Solve(const VectorType& rhs, VectorType* x)
    {
        assert(this->build_ == true && x != NULL && x != &rhs);
        DISPATCH_OPERATOR_SOLVE_STRATEGY(this->solver_descr_, this->ILU_, LUSolve, rhs, x);
    }
    const OperatorType& GetFactors(void) const
    {
        return this->ILU_;
    }

private:
    OperatorType ILU_;
    int          p_;
    bool         level_;
};

// ---- IC (incomplete Cholesky, zero fill-in): preconditioner.cpp:826-925
template <class OperatorType, class VectorType, typename ValueType>
class IC : public Preconditioner<OperatorType, VectorType, ValueType>
{
public:
    IC() {}
    virtual ~IC()
    {
        this->Clear();
    }
    virtual void Print(void) const
    {
        LOG_INFO("IC preconditioner");
        if(this->build_)
            LOG_INFO("IC nnz = " << this->IC_.GetNnz());
    }
    virtual void Build(void)
    {
        if(this->build_)
            this->Clear();
        this->build_ = true;
        assert(this->op_ != NULL);
        this->IC_.CloneBackend(*this->op_);
        this->inv_diag_entries_.CloneBackend(*this->op_);
        this->op_->ExtractL(&this->IC_, true);
        this->IC_.ICFactorize(&this->inv_diag_entries_);
        DISPATCH_OPERATOR_ANALYSE_STRATEGY(this->solver_descr_, this->IC_, LLAnalyse);
    }
    virtual void Clear(void)
    {
        DISPATCH_OPERATOR_ANALYSE_STRATEGY(this->solver_descr_, this->IC_, LLAnalyseClear);
        this->inv_diag_entries_.Clear();
        this->IC_.Clear();
        this->build_ = false;
    }
    virtual void Solve(const VectorType& rhs, VectorType* x)
    {
        assert(this->build_ == true && x != NULL && x != &rhs);
        DISPATCH_OPERATOR_SOLVE_STRATEGY(this->solver_descr_, this->IC_, LLSolve, rhs, this->inv_diag_entries_, x);
    }
    const OperatorType& GetFactor(void) const
    {
        return this->IC_;
    }
    const VectorType& GetInverseDiagonal(void) const
    {
        return this->inv_diag_entries_;
    }

private:
    OperatorType IC_;
    VectorType   inv_diag_entries_;
};

// ---- GS / SGS: preconditioner.cpp:206-257 / :302-379 (sparse triangular solves on the matrix itself).
// SGS::Build fills diag_entries_ with the INVERSE diagonal (:318, ExtractInverseDiagonal) -- kept as is.
template <class OperatorType, class VectorType, typename ValueType>
class GS : public Preconditioner<OperatorType, VectorType, ValueType>
{
public:
    GS() {}
    virtual ~GS()
    {
        this->Clear();
    }
    virtual void Print(void) const
    {
        LOG_INFO("Gauss-Seidel (GS) preconditioner");
    }
    virtual void Build(void)
    {
        if(this->build_)
            this->Clear();
        this->build_ = true;
        assert(this->op_ != NULL);
        this->GS_.CloneFrom(*this->op_);
        DISPATCH_OPERATOR_ANALYSE_STRATEGY(this->solver_descr_, this->GS_, LAnalyse, false);
    }
    virtual void Clear(void)
    {
        DISPATCH_OPERATOR_ANALYSE_STRATEGY(this->solver_descr_, this->GS_, LAnalyseClear);
        this->GS_.Clear();
        this->build_ = false;
    }
    virtual void Solve(const VectorType& rhs, VectorType* x)
    {
        assert(this->build_ == true && x != NULL);
        DISPATCH_OPERATOR_SOLVE_STRATEGY(this->solver_descr_, this->GS_, LSolve, rhs, x);
    }

private:
    OperatorType GS_;
};

template <class OperatorType, class VectorType, typename ValueType>
class SGS : public Preconditioner<OperatorType, VectorType, ValueType>
{
public:
    SGS() {}
    virtual ~SGS()
    {
        this->Clear();
    }
    virtual void Print(void) const
    {
        LOG_INFO("Symmetric Gauss-Seidel (SGS) preconditioner");
    }
    virtual void Build(void)
    {
        if(this->build_)
            this->Clear();
        this->build_ = true;
        assert(this->op_ != NULL);
        this->SGS_.CloneFrom(*this->op_);
        DISPATCH_OPERATOR_ANALYSE_STRATEGY(this->solver_descr_, this->SGS_, LAnalyse, false);
        DISPATCH_OPERATOR_ANALYSE_STRATEGY(this->solver_descr_, this->SGS_, UAnalyse, false);
        this->diag_entries_.CloneBackend(*this->op_);
        this->diag_entries_.Allocate("diag", this->op_->GetM());
        this->SGS_.ExtractInverseDiagonal(&this->diag_entries_);
        this->v_.CloneBackend(*this->op_);
        this->v_.Allocate("v", this->op_->GetM());
    }
    virtual void Clear(void)
    {
        DISPATCH_OPERATOR_ANALYSE_STRATEGY(this->solver_descr_, this->SGS_, LAnalyseClear);
        DISPATCH_OPERATOR_ANALYSE_STRATEGY(this->solver_descr_, this->SGS_, UAnalyseClear);
        this->SGS_.Clear();
        this->diag_entries_.Clear();
        this->v_.Clear();
        this->build_ = false;
    }
    virtual void Solve(const VectorType& rhs, VectorType* x)
    {
        assert(this->build_ == true && x != NULL);
        DISPATCH_OPERATOR_SOLVE_STRATEGY(this->solver_descr_, this->SGS_, LSolve, rhs, &this->v_);
        this->v_.PointWiseMult(this->diag_entries_);
        DISPATCH_OPERATOR_SOLVE_STRATEGY(this->solver_descr_, this->SGS_, USolve, this->v_, x);
    }

private:
    OperatorType SGS_;
    VectorType   diag_entries_;
    VectorType   v_;
};

// ---- MultiColored framework + MC-SGS: preconditioner_multicolored.cpp:148-413, _gs.cpp:127-215
template <class OperatorType, class VectorType, typename ValueType>
class MultiColored : public Preconditioner<OperatorType, VectorType, ValueType>
{
public:
    MultiColored()
        : op_mat_format_(false)
        , precond_mat_format_(CSR)
        , format_block_dim_(1)
        , decomp_(true)
        , fused_sweeps_(true)
        , sweeps_(NULL)
        , preconditioner_(NULL)
        , num_blocks_(0)
        , block_sizes_(NULL)
    {
    }
    // extension: run the decomposed apply as 2*nb-1 fused colour sweeps (default) instead of the
    // reference's block-by-block sequence; both produce bit-identical results
    void SetFusedSweeps(bool on)
    {
        this->fused_sweeps_ = on;
    }
    virtual ~MultiColored()
    {
        this->Clear();
    }
    virtual void SetPrecondMatrixFormat(unsigned int mat_format, int blockdim = 1)
    {
        this->op_mat_format_      = true;
        this->precond_mat_format_ = mat_format;
        this->format_block_dim_   = blockdim;
    }
    virtual void SetDecomposition(bool decomp)
    {
        this->decomp_ = decomp;
    }
    int GetNumColors(void) const
    {
        return this->num_blocks_;
    }
    virtual void Build(void)
    {
        if(this->build_)
            this->Clear();
        assert(this->op_ != NULL);
        // Build_Analyser_: work on a clone of the operator
        this->preconditioner_ = new OperatorType;
        this->preconditioner_->CloneFrom(*this->op_);
        this->permutation_.CloneBackend(*this->op_);
        // Analyse_: greedy multi-colouring -> block sizes + permutation
        this->op_->MultiColoring(this->num_blocks_, &this->block_sizes_, &this->permutation_);
        // Permute_: P A P^T
        this->preconditioner_->Permute(this->permutation_);
        this->Factorize_();
        if(this->decomp_ && this->fused_sweeps_ && !this->op_mat_format_ && this->CanFuseSweeps_()
           && this->TryBuildSweeps_())
        {
            this->build_ = true;
            this->preconditioner_->Clear();
            return;
        }
        this->Decompose_();
        this->build_ = true;
        if(this->decomp_)
            this->preconditioner_->Clear();
        else
            this->PostAnalyse_();
    }
    virtual void Clear(void)
    {
        if(this->sweeps_ != NULL)
        {
            ramd_mcsgs_destroy(this->sweeps_);
            this->sweeps_ = NULL;
        }
        if(this->preconditioner_ != NULL)
        {
            this->preconditioner_->LAnalyseClear();
            this->preconditioner_->UAnalyseClear();
            this->preconditioner_->LUAnalyseClear();
            delete this->preconditioner_;
            this->preconditioner_ = NULL;
        }
        for(size_t i = 0; i < this->block_.size(); ++i)
            delete this->block_[i];
        for(size_t i = 0; i < this->x_block_.size(); ++i)
        {
            delete this->x_block_[i];
            delete this->diag_block_[i];
            delete this->diag_solver_[i];
        }
        this->block_.clear();
        this->x_block_.clear();
        this->diag_block_.clear();
        this->diag_solver_.clear();
        free_host(&this->block_sizes_);
        this->num_blocks_ = 0;
        this->diag_.Clear();
        this->x_.Clear();
        this->permutation_.Clear();
        this->build_ = false;
    }
    virtual void Solve(const VectorType& rhs, VectorType* x)
    {
        assert(this->build_ == true && x != NULL && x != &rhs);
        if(this->sweeps_ != NULL)
        {
            this->ApplySweeps_(rhs, x);
            return;
        }
        if(this->decomp_)
        {
            this->ExtractRHSinX_(rhs, x);
            this->SolveL_();
            this->SolveD_();
            this->SolveR_();
            this->InsertSolution_(x);
        }
        else
            this->Solve_(rhs, x);
    }

protected:
    virtual void Factorize_(void) {}
    virtual void PostAnalyse_(void) {}
    virtual bool CanFuseSweeps_(void) const
    {
        return false;
    }
    virtual int SweepKind_(void) const
    {
        return RAMD_MC_SGS;
    }
    template <class O = OperatorType>
    typename std::enable_if<std::is_same<O, LocalMatrix<ValueType>>::value, bool>::type TryBuildSweeps_(void)
    {
        if(!this->preconditioner_->is_accel_())
            return false;
        int s = ramd_mcsgs_build(this->preconditioner_->handle(), this->num_blocks_, this->block_sizes_,
                                 this->permutation_.handle(), &this->sweeps_);
        if(s == RAMD_ERR_UNSUPPORTED)
        {
            this->sweeps_ = NULL;
            return false;
        }
        RAMD_CHECK(s);
        return true;
    }
    template <class O = OperatorType>
    typename std::enable_if<!std::is_same<O, LocalMatrix<ValueType>>::value, bool>::type TryBuildSweeps_(void)
    {
        return false;
    }
    template <class V = VectorType>
    typename std::enable_if<std::is_same<V, LocalVector<ValueType>>::value, void>::type
        ApplySweeps_(const VectorType& rhs, VectorType* x)
    {
        RAMD_CHECK(ramd_mcsgs_apply_kind(this->sweeps_, this->SweepKind_(), rhs.handle(), x->handle()));
    }
    template <class V = VectorType>
    typename std::enable_if<!std::is_same<V, LocalVector<ValueType>>::value, void>::type
        ApplySweeps_(const VectorType&, VectorType*)
    {
    }
    virtual void SolveL_(void) = 0;
    virtual void SolveD_(void) = 0;
    virtual void SolveR_(void) = 0;
    virtual void Solve_(const VectorType& rhs, VectorType* x) = 0;

    OperatorType* blk_(int i, int j)
    {
        return this->block_[(size_t)i * this->num_blocks_ + j];
    }
    void Decompose_(void)
    {
        const int nb = this->num_blocks_;
        if(this->decomp_)
        {
            std::vector<int> offsets((size_t)nb + 1, 0);
            for(int i = 0; i < nb; ++i)
                offsets[i + 1] = offsets[i] + this->block_sizes_[i];
            this->block_.assign((size_t)nb * nb, NULL);
            std::vector<OperatorType**> rows((size_t)nb);
            for(int i = 0; i < nb; ++i)
            {
                for(int j = 0; j < nb; ++j)
                {
                    this->block_[(size_t)i * nb + j] = new OperatorType;
                    this->block_[(size_t)i * nb + j]->CloneBackend(*this->op_);
                }
                rows[i] = &this->block_[(size_t)i * nb];
            }
            this->preconditioner_->ExtractSubMatrices(nb, nb, offsets.data(), offsets.data(),
                                                      rows.data());
            this->x_block_.assign((size_t)nb, NULL);
            this->diag_block_.assign((size_t)nb, NULL);
            this->diag_solver_.assign((size_t)nb, NULL);
            for(int i = 0; i < nb; ++i)
            {
                this->diag_block_[i] = new VectorType;
                this->diag_block_[i]->CloneBackend(*this->op_);
                this->diag_block_[i]->Allocate("Diagonal preconditioners blocks", this->block_sizes_[i]);
                this->blk_(i, i)->ExtractDiagonal(this->diag_block_[i]);
                this->x_block_[i] = new VectorType;
                this->x_block_[i]->CloneBackend(*this->op_);
                this->x_block_[i]->Allocate("MultiColored Preconditioner x_block_",
                                            this->block_sizes_[i]);
                Jacobi<OperatorType, VectorType, ValueType>* jacobi
                    = new Jacobi<OperatorType, VectorType, ValueType>;
                jacobi->SetOperator(*this->blk_(i, i));
                jacobi->Build();
                this->diag_solver_[i] = jacobi;
                this->blk_(i, i)->Clear();
            }
            if(this->op_mat_format_)
                for(int i = 0; i < nb; ++i)
                    for(int j = 0; j < nb; ++j)
                        if(this->blk_(i, j)->GetNnz() > 0)
                            this->blk_(i, j)->ConvertTo(this->precond_mat_format_,
                                                        this->format_block_dim_);
        }
        else
        {
            this->diag_.CloneBackend(*this->op_);
            this->preconditioner_->ExtractDiagonal(&this->diag_);
        }
        this->x_.CloneBackend(*this->op_);
        this->x_.Allocate("Permuted solution vector", this->op_->GetM());
    }
    void ExtractRHSinX_(const VectorType& rhs, VectorType* x)
    {
        x->CopyFromPermute(rhs, this->permutation_);
        int64_t off = 0;
        for(int i = 0; i < this->num_blocks_; ++i)
        {
            this->x_block_[i]->CopyFrom(*x, off, 0, this->block_sizes_[i]);
            off += this->block_sizes_[i];
        }
    }
    void InsertSolution_(VectorType* x)
    {
        int64_t off = 0;
        for(int i = 0; i < this->num_blocks_; ++i)
        {
            this->x_.CopyFrom(*this->x_block_[i], 0, off, this->block_sizes_[i]);
            off += this->block_sizes_[i];
        }
        x->CopyFromPermuteBackward(this->x_, this->permutation_);
    }

    bool                          op_mat_format_;
    unsigned int                  precond_mat_format_;
    int                           format_block_dim_;
    bool                          decomp_;
    bool                          fused_sweeps_;
    ramd_mcsgs_t                  sweeps_;
    OperatorType*                 preconditioner_;
    std::vector<OperatorType*>    block_; // [i*nb+j]
    std::vector<VectorType*>      x_block_;
    std::vector<VectorType*>      diag_block_;
    std::vector<Solver<OperatorType, VectorType, ValueType>*> diag_solver_;
    VectorType                    x_;
    VectorType                    diag_;
    int                           num_blocks_;
    int*                          block_sizes_;
    LocalVector<int>              permutation_;
};

template <class OperatorType, class VectorType, typename ValueType>
class MultiColoredSGS : public MultiColored<OperatorType, VectorType, ValueType>
{
public:
    MultiColoredSGS()
        : omega_(static_cast<ValueType>(1))
    {
    }
    virtual ~MultiColoredSGS()
    {
        this->Clear();
    }
    virtual void Print(void) const
    {
        LOG_INFO("Multicolored Symmetric Gauss-Seidel (SGS) preconditioner");
        if(this->build_)
            LOG_INFO("number of colors = " << this->num_blocks_);
    }
    virtual void SetRelaxation(ValueType omega)
    {
        this->omega_ = omega;
    }

protected:
    virtual bool CanFuseSweeps_(void) const
    {
        return this->omega_ == static_cast<ValueType>(1); // the SSOR scalings are not fused
    }
    virtual void PostAnalyse_(void)
    {
        this->preconditioner_->LAnalyse(false);
        this->preconditioner_->UAnalyse(false);
    }
    void sweep_block_(int i, int j)
    {
        if(this->blk_(i, j)->GetNnz() > 0)
            this->blk_(i, j)->ApplyAdd(*this->x_block_[j], static_cast<ValueType>(-1),
                                       this->x_block_[i]);
    }
    virtual void SolveL_(void)
    {
        for(int i = 0; i < this->num_blocks_; ++i)
        {
            for(int j = 0; j < i; ++j)
                this->sweep_block_(i, j);
            this->diag_solver_[i]->Solve(*this->x_block_[i], this->x_block_[i]);
            if(this->omega_ != static_cast<ValueType>(1))
                this->x_block_[i]->Scale(static_cast<ValueType>(1) / this->omega_);
        }
    }
    virtual void SolveD_(void)
    {
        for(int i = 0; i < this->num_blocks_; ++i)
        {
            this->x_block_[i]->PointWiseMult(*this->diag_block_[i]);
            if(this->omega_ != static_cast<ValueType>(1))
                this->x_block_[i]->Scale(this->omega_ / (static_cast<ValueType>(2) - this->omega_));
        }
    }
    virtual void SolveR_(void)
    {
        for(int i = this->num_blocks_ - 1; i >= 0; --i)
        {
            for(int j = this->num_blocks_ - 1; j > i; --j) // descending j, as the reference
                this->sweep_block_(i, j);
            this->diag_solver_[i]->Solve(*this->x_block_[i], this->x_block_[i]);
            if(this->omega_ != static_cast<ValueType>(1))
                this->x_block_[i]->Scale(static_cast<ValueType>(1) / this->omega_);
        }
    }
    virtual void Solve_(const VectorType& rhs, VectorType* x)
    {
        this->x_.CopyFromPermute(rhs, this->permutation_);
        this->preconditioner_->LSolve(this->x_, x);
        x->PointWiseMult(this->diag_);
        this->preconditioner_->USolve(*x, &this->x_);
        x->CopyFromPermuteBackward(this->x_, this->permutation_);
    }
    ValueType omega_;
};

// preconditioner_multicolored_gs.cpp:218-288: class MultiColoredGS : public MultiColoredSGS --
// backward sweep only (SolveL_/SolveD_ empty); the non-decomposed form is "No implemented yet" there too
template <class OperatorType, class VectorType, typename ValueType>
class MultiColoredGS : public MultiColoredSGS<OperatorType, VectorType, ValueType>
{
public:
    MultiColoredGS() {}
    virtual ~MultiColoredGS()
    {
        this->Clear();
    }
    virtual void Print(void) const
    {
        LOG_INFO("Multicolored Gauss-Seidel (GS) preconditioner");
        if(this->build_)
            LOG_INFO("number of colors = " << this->num_blocks_);
    }

protected:
    virtual int SweepKind_(void) const
    {
        return RAMD_MC_GS;
    }
    virtual void PostAnalyse_(void)
    {
        this->preconditioner_->UAnalyse(false);
    }
    virtual void SolveL_(void) {}
    virtual void SolveD_(void) {}
    virtual void Solve_(const VectorType&, VectorType*)
    {
        LOG_INFO("No implemented yet");
        FATAL_ERROR(__FILE__, __LINE__);
    }
};

// preconditioner_multicolored_ilu.cpp: ILU(p,q) with the power(q)-pattern method.  Provided here:
// the default ILU(0,1) (colouring of A itself, ILU(0) of P A P^T).  p > 0 / q > 1 need
// SymbolicPower + ILUpFactorize, which this backend does not provide (fails loudly).
template <class OperatorType, class VectorType, typename ValueType>
class MultiColoredILU : public MultiColored<OperatorType, VectorType, ValueType>
{
public:
    MultiColoredILU()
        : q_(1)
        , p_(0)
        , level_(true)
        , nnz_(0)
    {
    }
    virtual ~MultiColoredILU()
    {
        this->Clear();
    }
    virtual void Print(void) const
    {
        LOG_INFO("Multicolored ILU preconditioner (power(q)-pattern method), ILU(" << this->p_ << ","
                                                                                   << this->q_ << ")");
        if(this->build_)
            LOG_INFO("number of colors = " << this->num_blocks_ << "; ILU nnz = " << this->nnz_);
    }
    virtual void Set(int p)
    {
        assert(this->build_ == false && p >= 0);
        this->p_ = p;
        this->q_ = p + 1;
    }
    virtual void Set(int p, int q, bool level = true)
    {
        assert(this->build_ == false && p >= 0 && q >= 1);
        this->p_     = p;
        this->q_     = q;
        this->level_ = level;
    }

protected:
    virtual bool CanFuseSweeps_(void) const
    {
        return true;
    }
    virtual int SweepKind_(void) const
    {
        return RAMD_MC_ILU;
    }
    virtual void Factorize_(void)
    {
        if(this->p_ != 0 || this->q_ != 1)
        {
            LOG_INFO("MultiColoredILU: only ILU(0,1) is provided by this backend (no SymbolicPower / "
                     "ILUpFactorize for p > 0)");
            FATAL_ERROR(__FILE__, __LINE__);
        }
        this->preconditioner_->ILU0Factorize(); // ILUpFactorize(0) (local_matrix.cpp:3920-3923)
        this->nnz_ = this->preconditioner_->GetNnz();
    }
    virtual void PostAnalyse_(void)
    {
        this->preconditioner_->LUAnalyse();
    }
    void sweep_block_(int i, int j)
    {
        if(this->blk_(i, j)->GetNnz() > 0)
            this->blk_(i, j)->ApplyAdd(*this->x_block_[j], static_cast<ValueType>(-1),
                                       this->x_block_[i]);
    }
    virtual void SolveL_(void)
    {
        for(int i = 0; i < this->num_blocks_; ++i)
            for(int j = 0; j < i; ++j)
                this->sweep_block_(i, j);
    }
    virtual void SolveD_(void) {}
    virtual void SolveR_(void)
    {
        for(int i = this->num_blocks_ - 1; i >= 0; --i)
        {
            for(int j = this->num_blocks_ - 1; j > i; --j)
                this->sweep_block_(i, j);
            this->diag_solver_[i]->Solve(*this->x_block_[i], this->x_block_[i]);
        }
    }
    virtual void Solve_(const VectorType& rhs, VectorType* x)
    {
        x->CopyFromPermute(rhs, this->permutation_);
        this->preconditioner_->LUSolve(*x, &this->x_);
        x->CopyFromPermuteBackward(this->x_, this->permutation_);
    }
    int     q_, p_;
    bool    level_;
    int64_t nnz_;
};

// ============================================================================ IterativeLinearSolver
template <class OperatorType, class VectorType, typename ValueType>
class IterativeLinearSolver : public Solver<OperatorType, VectorType, ValueType>
{
public:
    IterativeLinearSolver()
        : res_norm_type_(2)
        , index_(-1)
        , fused_(true)
    {
    }
    void Init(double abs_tol, double rel_tol, double div_tol, int max_iter)
    {
        this->iter_ctrl_.Init(abs_tol, rel_tol, div_tol, max_iter);
    }
    void Init(double abs_tol, double rel_tol, double div_tol, int min_iter, int max_iter)
    {
        this->iter_ctrl_.Init(abs_tol, rel_tol, div_tol, min_iter, max_iter);
    }
    void InitMinIter(int min_iter)
    {
        this->iter_ctrl_.InitMinimumIterations(min_iter);
    }
    void InitMaxIter(int max_iter)
    {
        this->iter_ctrl_.InitMaximumIterations(max_iter);
    }
    void InitTol(double abs, double rel, double div)
    {
        this->iter_ctrl_.InitTolerance(abs, rel, div);
    }
    virtual void ReBuildNumeric(void)
    {
        if(!this->build_)
            return;
        Solver<OperatorType, VectorType, ValueType>* pc = this->precond_;
        this->Clear(); // clears (and detaches) the preconditioner
        if(pc != NULL)
            this->precond_ = pc;
        this->Build();
    }
    void SetResidualNorm(int resnorm)
    {
        assert(resnorm == 1 || resnorm == 2 || resnorm == 3);
        this->res_norm_type_ = resnorm;
    }
    void RecordResidualHistory(void)
    {
        this->iter_ctrl_.RecordHistory();
    }
    void RecordHistory(const std::string& filename) const
    {
        this->iter_ctrl_.WriteHistoryToFile(filename);
    }
    const std::vector<double>& GetResidualHistory(void) const
    {
        return this->iter_ctrl_.GetResidualHistory();
    }
    virtual void Verbose(int verb = 1)
    {
        this->verb_ = verb;
        this->iter_ctrl_.Verbose(verb);
    }
    virtual int GetIterationCount(void)
    {
        return this->iter_ctrl_.GetIterationCount();
    }
    virtual double GetCurrentResidual(void)
    {
        return this->iter_ctrl_.GetCurrentResidual();
    }
    virtual int GetSolverStatus(void)
    {
        return this->iter_ctrl_.GetSolverStatus();
    }
    virtual int64_t GetAmaxResidualIndex(void)
    {
        return this->iter_ctrl_.GetAmaxResidualIndex();
    }
    virtual void SetPreconditioner(Solver<OperatorType, VectorType, ValueType>& precond)
    {
        assert(this != &precond);
        this->precond_ = &precond;
        this->precond_->FlagPrecond();
    }
    // extension: switch the fused device path of CG / GMRES on or off (default on). Both paths run
    // the same per-element arithmetic; only the summation order of the reductions differs.
    void SetFused(bool fused)
    {
        this->fused_ = fused;
    }
    // solver.cpp:470-500
    virtual void Solve(const VectorType& rhs, VectorType* x)
    {
        assert(x != NULL && x != &rhs && this->op_ != NULL && this->build_ == true);
        if(this->verb_ > 0)
        {
            this->PrintStart_();
            this->iter_ctrl_.PrintInit();
        }
        if(this->precond_ == NULL)
            this->SolveNonPrecond_(rhs, x);
        else
            this->SolvePrecond_(rhs, x);
        if(this->verb_ > 0)
        {
            this->iter_ctrl_.PrintStatus();
            this->PrintEnd_();
        }
    }

protected:
    virtual void PrintStart_(void) const = 0;
    virtual void PrintEnd_(void) const   = 0;
    virtual void SolveNonPrecond_(const VectorType& rhs, VectorType* x) = 0;
    virtual void SolvePrecond_(const VectorType& rhs, VectorType* x)    = 0;
    // solver.cpp:443-468
    ValueType Norm_(const VectorType& vec)
    {
        if(this->res_norm_type_ == 1)
            return vec.Asum();
        if(this->res_norm_type_ == 2)
            return vec.Norm();
        ValueType amax;
        this->index_ = vec.Amax(amax);
        return amax;
    }
    IterationControl iter_ctrl_;
    int              res_norm_type_;
    int64_t          index_;
    bool             fused_;
};

// The fused device loops are written against four small helpers so that the same loop serves
// Local objects (one GPU) and Global objects (one rank of a row-block decomposition; global.hpp adds
// the overloads: interior handle, halo-exchanging Apply + dot, and an RCCL all-reduce of the scalar
// record between the kernels -- stream-ordered, no host round trip).
template <class OperatorType, class VectorType, typename ValueType>
struct _fusable
{
    static constexpr bool value = false;
};
template <typename ValueType>
struct _fusable<LocalMatrix<ValueType>, LocalVector<ValueType>, ValueType>
{
    static constexpr bool value = true;
};
template <typename ValueType>
inline ramd_vec_t _fh(const LocalVector<ValueType>& v)
{
    return v.handle();
}
template <typename ValueType>
inline void _f_apply_dot(const LocalMatrix<ValueType>& A, const LocalVector<ValueType>& p,
                         LocalVector<ValueType>* q, int slot)
{
    RAMD_CHECK(ramd_fused_apply_dot(A.handle(), p.handle(), q->handle(), slot));
}
template <typename ValueType>
inline void _f_apply_dotv(const LocalMatrix<ValueType>& A, const LocalVector<ValueType>& x,
                          LocalVector<ValueType>* y, const LocalVector<ValueType>& w, int slot)
{
    RAMD_CHECK(ramd_fused_apply_dotv(A.handle(), x.handle(), y->handle(), w.handle(), slot)); // y = A x ; <w,y>
}
template <typename ValueType>
inline void _f_allreduce(const LocalMatrix<ValueType>&, int, int)
{
}

// ============================================================================ CG
template <class OperatorType, class VectorType, typename ValueType>
class CG : public IterativeLinearSolver<OperatorType, VectorType, ValueType>
{
public:
    virtual ~CG()
    {
        this->Clear();
    }
    virtual void Print(void) const
    {
        LOG_INFO("CG solver" << (this->precond_ ? ", with preconditioner" : " (non-precond)"));
    }
    // cg.cpp:99-137
    virtual void Build(void)
    {
        if(this->build_)
            this->Clear();
        assert(this->op_ != NULL && this->op_->GetM() == this->op_->GetN() && this->op_->GetM() > 0);
        this->build_ = true;
        if(this->precond_ != NULL)
        {
            this->precond_->SetOperator(*this->op_);
            this->precond_->Build();
            this->z_.CloneBackend(*this->op_);
            this->z_.Allocate("z", this->op_->GetM());
        }
        this->r_.CloneBackend(*this->op_);
        this->r_.Allocate("r", this->op_->GetM());
        this->p_.CloneBackend(*this->op_);
        this->p_.Allocate("p", this->op_->GetM());
        this->q_.CloneBackend(*this->op_);
        this->q_.Allocate("q", this->op_->GetM());
    }
    virtual void Clear(void)
    {
        if(this->build_)
        {
            if(this->precond_ != NULL)
            {
                this->precond_->Clear();
                this->precond_ = NULL;
            }
            this->r_.Clear();
            this->z_.Clear();
            this->p_.Clear();
            this->q_.Clear();
            this->iter_ctrl_.Clear();
            this->build_ = false;
        }
    }

protected:
    virtual void PrintStart_(void) const
    {
        LOG_INFO("CG " << (this->precond_ ? "" : "(non-precond) ") << "linear solver starts");
    }
    virtual void PrintEnd_(void) const
    {
        LOG_INFO("CG ends");
    }
    virtual void SolveNonPrecond_(const VectorType& rhs, VectorType* x)
    {
        this->Solve_(rhs, x, false);
    }
    virtual void SolvePrecond_(const VectorType& rhs, VectorType* x)
    {
        this->Solve_(rhs, x, true);
    }

private:
    // cg.cpp:291-362 / :366-446
    void Solve_(const VectorType& rhs, VectorType* x, bool precond)
    {
        const OperatorType* op = this->op_;
        VectorType *        r = &this->r_, *z = &this->z_, *p = &this->p_, *q = &this->q_;
        ValueType           alpha, beta, rho, rho_old;

        op->Apply(*x, r);
        r->ScaleAdd(static_cast<ValueType>(-1), rhs);
        ValueType res_norm = this->Norm_(*r);
        if(this->iter_ctrl_.InitResidual(std::abs(res_norm)) == false)
            return;
        if(precond)
        {
            this->precond_->SolveZeroSol(*r, z);
            p->CopyFrom(*z);
        }
        else
            p->CopyFrom(*r);

        if(this->fused_ && this->res_norm_type_ == 2 && this->FusedLoop_(rhs, x, precond))
            return;

        rho = precond ? r->DotNonConj(*z) : r->DotNonConj(*r);
        while(true)
        {
            op->Apply(*p, q);
            alpha = rho / p->DotNonConj(*q);
            x->AddScale(*p, alpha);
            r->AddScale(*q, -alpha);
            res_norm = this->Norm_(*r);
            if(this->iter_ctrl_.CheckResidual(std::abs(res_norm), this->index_))
                break;
            rho_old = rho;
            if(precond)
            {
                this->precond_->SolveZeroSol(*r, z);
                rho  = r->DotNonConj(*z);
                beta = rho / rho_old;
                p->ScaleAdd(beta, *z);
            }
            else
            {
                rho  = r->DotNonConj(*r);
                beta = rho / rho_old;
                p->ScaleAdd(beta, *r);
            }
        }
    }

    // Fused device loop (Local objects on the accelerator).  Per iteration:
    //   K1  q = A p, <p,q>                                   (ramd_fused_apply_dot)
    //   K2  r -= a q ; <r,r> ; [z = D^-1 r ; <r,z>]            (ramd_fused_cg_update)
    //   K3  x += a p ; p = (rho'/rho) p + z                    (ramd_fused_cg_direction)
    // The ||r|| read-back for iteration k overlaps K3(k) and K1(k+1), which are queued before the
    // host waits; the convergence decision is the reference's, made on the same ||r||.
    template <class O = OperatorType, class V = VectorType>
    typename std::enable_if<_fusable<O, V, ValueType>::value, bool>::type
        FusedLoop_(const VectorType& rhs, VectorType* x, bool precond)
    {
        (void)rhs;
        if(!this->op_->is_accel_() || !x->is_accel_())
            return false;
        const OperatorType& A = *this->op_;
        VectorType *r = &this->r_, *z = &this->z_, *p = &this->p_, *q = &this->q_;
        typedef Jacobi<OperatorType, VectorType, ValueType> JacobiType;
        JacobiType* jac = precond ? dynamic_cast<JacobiType*>(this->precond_) : NULL;
        ramd_vec_t  dinv = NULL;
        if(jac != NULL && jac->GetInverseDiagonal().GetSize() == r->GetSize())
            dinv = _fh(jac->GetInverseDiagonal());
        const bool  generic_pc = precond && dinv == NULL;
        if(generic_pc && this->precond_->SolveUsesScalarRecord())
            return false; // e.g. a multigrid cycle as preconditioner: the plain loop keeps its scalars on the host
        VectorType* zdir       = precond ? z : r;

        // scalar slots: <p,q> = 0, ||r||^2 = 2, rho alternates between 1 and 3 (always adjacent to
        // slot 2, so the two scalars of the update kernel are summed over ranks by ONE all-reduce)
        enum { S_PQ = 0, S_RR = 2 };
        int s_rho = 1, s_new = 3;
        const ramd_vec_t first[1] = {_fh(*r)};
        RAMD_CHECK(ramd_fused_multi_dot(first, 1, _fh(*zdir), s_rho)); // rho = <r, z> (or <r, r>)
        _f_allreduce(A, s_rho, 1);
        _f_apply_dot(A, *p, q, S_PQ);
        _f_allreduce(A, S_PQ, 1);
        int rec = 0;
        while(true)
        {
            RAMD_CHECK(ramd_fused_cg_update(_fh(*r), _fh(*q), dinv, dinv ? _fh(*z) : NULL, s_rho, S_PQ,
                                            S_RR, s_new));
            if(generic_pc)
            {
                this->precond_->SolveZeroSol(*r, z);
                RAMD_CHECK(ramd_fused_multi_dot(first, 1, _fh(*z), s_new));
            }
            _f_allreduce(A, s_new < S_RR ? s_new : S_RR, 2);
            RAMD_CHECK(ramd_scalars_fetch_async_begin(rec, S_RR, 1));
            RAMD_CHECK(ramd_fused_cg_direction(_fh(*x), _fh(*p), _fh(*zdir), s_rho, S_PQ, s_new));
            _f_apply_dot(A, *p, q, S_PQ);
            _f_allreduce(A, S_PQ, 1);
            double rr = 0.0;
            RAMD_CHECK(ramd_scalars_fetch_async_end(rec, &rr, 1));
            ValueType res_norm = (ValueType)std::sqrt(rr);
            if(this->iter_ctrl_.CheckResidual(std::abs(res_norm), this->index_))
                break;
            std::swap(s_rho, s_new);
            rec = (rec + 1) & 7;
        }
        return true;
    }
    template <class O = OperatorType, class V = VectorType>
    typename std::enable_if<!_fusable<O, V, ValueType>::value, bool>::type
        FusedLoop_(const VectorType&, VectorType*, bool)
    {
        return false;
    }

    VectorType r_, z_, p_, q_;
};

// ============================================================================ GMRES
template <class OperatorType, class VectorType, typename ValueType>
class GMRES : public IterativeLinearSolver<OperatorType, VectorType, ValueType>
{
public:
    GMRES()
        : flexible_(false)
        , size_basis_(30) // gmres.cpp:50
        , v_(NULL)
        , zb_(NULL)
    {
    }
    virtual ~GMRES()
    {
        this->Clear();
    }
    virtual void Print(void) const
    {
        LOG_INFO((this->flexible_ ? "FGMRES(" : "GMRES(")
                 << this->size_basis_ << ") solver"
                 << (this->precond_ ? ", with preconditioner" : " (non-precond)"));
    }
    virtual void SetBasisSize(int size_basis)
    {
        assert(size_basis > 0 && this->build_ == false);
        this->size_basis_ = size_basis;
    }
    // gmres.cpp:109-156
    virtual void Build(void)
    {
        if(this->build_)
            this->Clear();
        assert(this->op_ != NULL && this->op_->GetM() == this->op_->GetN() && this->op_->GetM() > 0);
        this->build_ = true;
        const int m  = this->size_basis_;
        this->c_.assign((size_t)m, ValueType(0));
        this->s_.assign((size_t)m, ValueType(0));
        this->r_.assign((size_t)m + 1, ValueType(0));
        this->H_.assign((size_t)(m + 1) * m, ValueType(0));
        this->v_ = new VectorType*[m + 1];
        for(int i = 0; i < m + 1; ++i)
        {
            this->v_[i] = new VectorType;
            this->v_[i]->CloneBackend(*this->op_);
            this->v_[i]->Allocate("v", this->op_->GetM());
        }
        if(this->precond_ != NULL)
        {
            if(this->flexible_) // fgmres.cpp:139-150: one z per basis vector
            {
                this->zb_ = new VectorType*[m + 1];
                for(int i = 0; i < m + 1; ++i)
                {
                    this->zb_[i] = new VectorType;
                    this->zb_[i]->CloneBackend(*this->op_);
                    this->zb_[i]->Allocate("z", this->op_->GetM());
                }
            }
            else
            {
                this->z_.CloneBackend(*this->op_);
                this->z_.Allocate("z", this->op_->GetM());
            }
            this->precond_->SetOperator(*this->op_);
            this->precond_->Build();
        }
    }
    virtual void Clear(void)
    {
        if(this->build_)
        {
            if(this->precond_ != NULL)
            {
                this->precond_->Clear();
                this->precond_ = NULL;
            }
            for(int i = 0; i < this->size_basis_ + 1; ++i)
            {
                delete this->v_[i];
                if(this->zb_ != NULL)
                    delete this->zb_[i];
            }
            delete[] this->v_;
            delete[] this->zb_;
            this->v_  = NULL;
            this->zb_ = NULL;
            this->z_.Clear();
            this->iter_ctrl_.Clear();
            this->build_ = false;
        }
    }

protected:
    virtual void PrintStart_(void) const
    {
        LOG_INFO((this->flexible_ ? "FGMRES(" : "GMRES(")
                 << this->size_basis_ << ") " << (this->precond_ ? "" : "(non-precond) ")
                 << "linear solver starts");
    }
    virtual void PrintEnd_(void) const
    {
        LOG_INFO((this->flexible_ ? "FGMRES(" : "GMRES(") << this->size_basis_ << ") ends");
    }
    virtual void SolveNonPrecond_(const VectorType& rhs, VectorType* x)
    {
        this->Solve_(rhs, x, false);
    }
    virtual void SolvePrecond_(const VectorType& rhs, VectorType* x)
    {
        this->Solve_(rhs, x, true);
    }

private:
    int hidx_(int i, int j) const // DENSE_IND, column-major (m+1) x m (matrix_formats_ind.hpp:30)
    {
        return i + j * (this->size_basis_ + 1);
    }
    // gmres.cpp:565-607
    static void GenerateGivensRotation_(ValueType dx, ValueType dy, ValueType& c, ValueType& s)
    {
        const ValueType zero = static_cast<ValueType>(0), one = static_cast<ValueType>(1);
        if(dy == zero)
        {
            c = one;
            s = zero;
        }
        else if(dx == zero)
        {
            c = zero;
            s = one;
        }
        else if(std::abs(dy) > std::abs(dx))
        {
            ValueType tmp = dx / dy;
            s             = one / std::sqrt(one + tmp * tmp);
            c             = tmp * s;
        }
        else
        {
            ValueType tmp = dy / dx;
            c             = one / std::sqrt(one + tmp * tmp);
            s             = tmp * c;
        }
    }
    static void ApplyGivensRotation_(ValueType c, ValueType s, ValueType& dx, ValueType& dy)
    {
        ValueType temp = dx;
        dx             = c * dx + s * dy;
        dy             = -s * temp + c * dy;
    }
    // residual -> v_0 (through z_ and M^-1 when preconditioned): gmres.cpp:444-454, :542-552
    void Residual_(const VectorType& rhs, VectorType* x, bool precond)
    {
        const ValueType one = static_cast<ValueType>(1);
        if(precond && !this->flexible_)
        {
            this->op_->Apply(*x, &this->z_);
            this->z_.ScaleAdd(-one, rhs);
            this->precond_->SolveZeroSol(this->z_, this->v_[0]);
        }
        else
        {
            this->op_->Apply(*x, this->v_[0]);
            this->v_[0]->ScaleAdd(-one, rhs);
        }
    }
    // one Arnoldi step: fills column i of H (rows 0..i+1) and normalises v_{i+1}
    void Arnoldi_(int i, bool precond)
    {
        VectorType**    v   = this->v_;
        ValueType*      H   = this->H_.data();
        const ValueType one = static_cast<ValueType>(1);
        if(precond && this->flexible_) // fgmres.cpp:462-466: M z_i = v_i ; v_i+1 = A z_i
        {
            this->precond_->SolveZeroSol(*v[i], this->zb_[i]);
            this->op_->Apply(*this->zb_[i], v[i + 1]);
        }
        else if(precond)
        {
            this->op_->Apply(*v[i], &this->z_);
            this->precond_->SolveZeroSol(this->z_, v[i + 1]);
        }
        else
            this->op_->Apply(*v[i], v[i + 1]);
        if(this->fused_ && this->res_norm_type_ == 2 && this->FusedMGS_(i))
            return;
        for(int k = 0; k <= i; ++k) // modified Gram-Schmidt
        {
            H[this->hidx_(k, i)] = v[k]->Dot(*v[i + 1]);
            v[i + 1]->AddScale(*v[k], -H[this->hidx_(k, i)]);
        }
        H[this->hidx_(i + 1, i)] = this->Norm_(*v[i + 1]);
        v[i + 1]->Scale(one / H[this->hidx_(i + 1, i)]);
    }
    // Fused MGS: every projection  w -= h_k v_k  is fused with the NEXT dot <v_{k+1}, w> (or with
    // <w,w> for the last one) and the normalisation reads ||w|| on the device: i+2 launches and
    // ONE host read per Arnoldi step instead of 2i+4 launches and i+2 blocking reads.  Same MGS
    // order and per-element arithmetic as the loop above.
    template <class O = OperatorType, class V = VectorType>
    typename std::enable_if<_fusable<O, V, ValueType>::value, bool>::type FusedMGS_(int i)
    {
        if(i + 3 > RAMD_NSCALARS - 2 || !this->v_[0]->is_accel_())
            return false;
        const OperatorType& A = *this->op_;
        VectorType**        v = this->v_;
        ValueType*          H = this->H_.data();
        ramd_vec_t          w = _fh(*v[i + 1]);
        const ramd_vec_t    v0[1] = {_fh(*v[0])};
        RAMD_CHECK(ramd_fused_multi_dot(v0, 1, w, 0)); // s[0] = <v_0, w>
        _f_allreduce(A, 0, 1);
        for(int k = 0; k <= i; ++k) // s[k+1] = <v_{k+1}, w - h_k v_k>  (k == i: <w,w>)
        {
            RAMD_CHECK(ramd_fused_mgs_step(w, _fh(*v[k]), k, (k < i) ? _fh(*v[k + 1]) : NULL, k + 1));
            _f_allreduce(A, k + 1, 1);
        }
        RAMD_CHECK(ramd_fused_normalize(w, i + 1, i + 2)); // s[i+2] = ||w|| ; w /= ||w||
        std::vector<double> h((size_t)i + 3);
        RAMD_CHECK(ramd_scalars_fetch(h.data(), 0, i + 3));
        for(int k = 0; k <= i; ++k)
            H[this->hidx_(k, i)] = (ValueType)h[k];
        H[this->hidx_(i + 1, i)] = (ValueType)h[i + 2];
        return true;
    }
    template <class O = OperatorType, class V = VectorType>
    typename std::enable_if<!_fusable<O, V, ValueType>::value, bool>::type FusedMGS_(int)
    {
        return false;
    }

    // gmres.cpp:274-413 / :416-562
    void Solve_(const VectorType& rhs, VectorType* x, bool precond)
    {
        VectorType**    v    = this->v_;
        ValueType *     c = this->c_.data(), *s = this->s_.data(), *r = this->r_.data();
        ValueType*      H    = this->H_.data();
        const ValueType one  = static_cast<ValueType>(1);
        const int       size = this->size_basis_;

        this->Residual_(rhs, x, precond);
        std::fill(this->r_.begin(), this->r_.end(), ValueType(0));
        r[0] = this->Norm_(*v[0]);
        if(this->iter_ctrl_.InitResidual(std::abs(r[0])) == false)
            return;
        while(true)
        {
            v[0]->Scale(one / r[0]);
            int i = 0;
            while(i < size)
            {
                this->Arnoldi_(i, precond);
                for(int k = 0; k < i; ++k)
                    ApplyGivensRotation_(c[k], s[k], H[this->hidx_(k, i)], H[this->hidx_(k + 1, i)]);
                GenerateGivensRotation_(H[this->hidx_(i, i)], H[this->hidx_(i + 1, i)], c[i], s[i]);
                ApplyGivensRotation_(c[i], s[i], H[this->hidx_(i, i)], H[this->hidx_(i + 1, i)]);
                ApplyGivensRotation_(c[i], s[i], r[i], r[i + 1]);
                if(this->iter_ctrl_.CheckResidual(std::abs(r[++i])))
                    break;
            }
            for(int j = i - 1; j >= 0; --j) // back substitution on the host
            {
                r[j] /= H[this->hidx_(j, j)];
                for(int k = 0; k < j; ++k)
                    r[k] -= H[this->hidx_(k, j)] * r[j];
            }
            VectorType** upd = (precond && this->flexible_) ? this->zb_ : v; // fgmres.cpp:527-532
            x->AddScale(*upd[0], r[0]);
            for(int j = 1; j < i; ++j)
                x->AddScale(*upd[j], r[j]);
            this->Residual_(rhs, x, precond);
            std::fill(this->r_.begin(), this->r_.end(), ValueType(0));
            r[0] = this->Norm_(*v[0]);
            if(this->iter_ctrl_.CheckResidualNoCount(std::abs(r[0])))
                break;
        }
    }

protected:
    bool flexible_; // FGMRES: right preconditioning with a stored z_i per basis vector

private:
    int                    size_basis_;
    VectorType**           v_;
    VectorType**           zb_;
    VectorType             z_;
    std::vector<ValueType> c_, s_, r_, H_;
};

// fgmres.cpp: flexible GMRES -- same Arnoldi/Givens machinery as GMRES (it is the same code in the
// reference), right-preconditioned: z_i = M^-1 v_i is kept for the solution update, the residual is
// the true one
template <class OperatorType, class VectorType, typename ValueType>
class FGMRES : public GMRES<OperatorType, VectorType, ValueType>
{
public:
    FGMRES()
    {
        this->flexible_ = true;
    }
    virtual ~FGMRES()
    {
        this->Clear();
    }
};

// ============================================================================ BiCGStab
template <class OperatorType, class VectorType, typename ValueType>
class BiCGStab : public IterativeLinearSolver<OperatorType, VectorType, ValueType>
{
public:
    virtual ~BiCGStab()
    {
        this->Clear();
    }
    virtual void Print(void) const
    {
        LOG_INFO("BiCGStab solver" << (this->precond_ ? ", with preconditioner" : " (non-precond)"));
    }
    // bicgstab.cpp:109-160
    virtual void Build(void)
    {
        if(this->build_)
            this->Clear();
        assert(this->op_ != NULL && this->op_->GetM() == this->op_->GetN() && this->op_->GetM() > 0);
        this->build_ = true;
        VectorType* all[] = {&this->r_, &this->r0_, &this->p_, &this->q_, &this->t_};
        for(VectorType* vec : all)
        {
            vec->CloneBackend(*this->op_);
            vec->Allocate("bicgstab", this->op_->GetM());
        }
        if(this->precond_ != NULL)
        {
            this->precond_->SetOperator(*this->op_);
            this->precond_->Build();
            this->v_.CloneBackend(*this->op_);
            this->v_.Allocate("v", this->op_->GetM());
            this->z_.CloneBackend(*this->op_);
            this->z_.Allocate("z", this->op_->GetM());
        }
    }
    virtual void Clear(void)
    {
        if(this->build_)
        {
            if(this->precond_ != NULL)
            {
                this->precond_->Clear();
                this->precond_ = NULL;
            }
            VectorType* all[] = {&this->r_, &this->r0_, &this->p_, &this->q_, &this->t_, &this->v_,
                                 &this->z_};
            for(VectorType* vec : all)
                vec->Clear();
            this->iter_ctrl_.Clear();
            this->build_ = false;
        }
    }

protected:
    virtual void PrintStart_(void) const
    {
        LOG_INFO("BiCGStab " << (this->precond_ ? "" : "(non-precond) ") << "linear solver starts");
    }
    virtual void PrintEnd_(void) const
    {
        LOG_INFO("BiCGStab ends");
    }
    virtual void SolveNonPrecond_(const VectorType& rhs, VectorType* x)
    {
        this->Solve_(rhs, x, false);
    }
    virtual void SolvePrecond_(const VectorType& rhs, VectorType* x)
    {
        this->Solve_(rhs, x, true);
    }

private:
    // Fused device loop: per iteration  K1 q = A dir + <r0,q> | K2 r -= alpha q | [v = M^-1 r] |
    // t = A sv, one pass for <t,r>,<t,t> | K3 x,r updates + <r,r>,<r0,r> | K4 p update | [z = M^-1 p].
    // alpha/omega/beta never leave the device; ONE host read per iteration (the four dots of K3's
    // record), overlapped with K4 and the next preconditioner apply.  Same per-element arithmetic and
    // the same breakdown / stopping decisions as the loop below.
    template <class O = OperatorType, class V = VectorType>
    typename std::enable_if<_fusable<O, V, ValueType>::value, bool>::type
        FusedLoop_(const VectorType& rhs, VectorType* x, bool precond)
    {
        if(!this->op_->is_accel_() || !x->is_accel_())
            return false;
        if(precond && this->precond_->SolveUsesScalarRecord())
            return false; // a preconditioner with reductions of its own would overwrite alpha / omega / rho on the device
        const OperatorType& A = *this->op_;
        VectorType *r = &this->r_, *r0 = &this->r0_, *p = &this->p_, *q = &this->q_, *t = &this->t_;
        VectorType *v = &this->v_, *z = &this->z_;
        const ValueType one = static_cast<ValueType>(1);
        // slots: <t,r> = 0, <t,t> = 1, <r0,q> = 2, ||r||^2 = 4, rho alternates between 3 and 5 (always next
        // to slot 4: the two sums of K3 cross the ranks in ONE all-reduce), breakdown flag = 6
        enum { S_TR = 0, S_R0Q = 2, S_RR = 4, S_FLAG = 6 };
        int s_rho = 3, s_new = 5;
        const ramd_vec_t rv[1] = {_fh(*r)};
        RAMD_CHECK(ramd_fused_multi_dot(rv, 1, _fh(*r), s_rho)); // rho = <r,r>
        _f_allreduce(A, s_rho, 1);
        int rec = 0;
        while(true)
        {
            const VectorType* dir = precond ? z : p;
            _f_apply_dotv(A, *dir, q, *r0, S_R0Q);
            _f_allreduce(A, S_R0Q, 1);
            RAMD_CHECK(ramd_fused_bicg_r_update(_fh(*r), _fh(*q), s_rho, S_R0Q));
            const VectorType* sv = r;
            if(precond)
            {
                this->precond_->SolveZeroSol(*r, v);
                sv = v;
            }
            A.Apply(*sv, t);
            const ramd_vec_t rt[2] = {_fh(*r), _fh(*t)};
            RAMD_CHECK(ramd_fused_multi_dot(rt, 2, _fh(*t), S_TR)); // <t,r>, <t,t> in one pass
            _f_allreduce(A, S_TR, 2);
            RAMD_CHECK(ramd_fused_bicg_xr_update(_fh(*x), precond ? _fh(*dir) : NULL, precond ? _fh(*sv) : NULL,
                                                 _fh(*r), _fh(*t), _fh(*r0), _fh(*p), s_rho, S_R0Q, S_TR, S_RR,
                                                 s_new, S_FLAG));
            _f_allreduce(A, s_new < S_RR ? s_new : S_RR, 2);
            RAMD_CHECK(ramd_scalars_fetch_async_begin(rec, 0, 7));
            // queued ahead of the host's decision; they only touch p / z, which a finished solve discards
            RAMD_CHECK(ramd_fused_bicg_direction(_fh(*p), _fh(*q), _fh(*r), s_rho, S_R0Q, S_TR, s_new));
            if(precond)
                this->precond_->SolveZeroSol(*p, z);
            double h[7];
            RAMD_CHECK(ramd_scalars_fetch_async_end(rec, h, 7));
            rec = (rec + 1) & 7;
            if(h[S_FLAG] != 0.0) // bicgstab.cpp:430-447 (x += alpha p was done by the kernel)
            {
                LOG_INFO("BiCGStab omega == 0 || Nan || Inf !!! Updated solution only in p-direction");
                A.Apply(*x, p);
                p->ScaleAdd(-one, rhs);
                ValueType res_norm = this->Norm_(*p);
                this->iter_ctrl_.CheckResidual(std::abs(res_norm), this->index_);
                break;
            }
            ValueType res_norm = (ValueType)std::sqrt(h[S_RR]);
            if(this->iter_ctrl_.CheckResidual(std::abs(res_norm), this->index_))
                break;
            if((ValueType)h[s_new] == static_cast<ValueType>(0))
            {
                LOG_INFO("BiCGStab rho == 0 !!!");
                break;
            }
            std::swap(s_rho, s_new);
        }
        return true;
    }
    template <class O = OperatorType, class V = VectorType>
    typename std::enable_if<!_fusable<O, V, ValueType>::value, bool>::type
        FusedLoop_(const VectorType&, VectorType*, bool)
    {
        return false;
    }

    // bicgstab.cpp:245-361 / :365-489 (right preconditioned: z = M^-1 p, v = M^-1 r)
    void Solve_(const VectorType& rhs, VectorType* x, bool precond)
    {
        const OperatorType* op = this->op_;
        VectorType *r = &this->r_, *r0 = &this->r0_, *p = &this->p_, *q = &this->q_, *t = &this->t_;
        VectorType *v = &this->v_, *z = &this->z_;
        ValueType   alpha, beta, omega, rho, rho_old;
        const ValueType one = static_cast<ValueType>(1);

        op->Apply(*x, r0);
        r0->ScaleAdd(-one, rhs);
        ValueType res_norm = this->Norm_(*r0);
        if(this->iter_ctrl_.InitResidual(std::abs(res_norm)) == false)
            return;
        r->CopyFrom(*r0);
        p->CopyFrom(*r);
        if(precond)
            this->precond_->SolveZeroSol(*r, z);
        if(this->fused_ && this->res_norm_type_ == 2 && this->FusedLoop_(rhs, x, precond))
            return;
        rho = r->Dot(*r);
        while(true)
        {
            const VectorType* dir = precond ? z : p;
            op->Apply(*dir, q);
            alpha = rho / r0->Dot(*q);
            r->AddScale(*q, -alpha);
            const VectorType* sv = r;
            if(precond)
            {
                this->precond_->SolveZeroSol(*r, v);
                sv = v;
            }
            op->Apply(*sv, t);
            omega = t->Dot(*r) / t->Dot(*t);
            if((std::abs(omega) == std::numeric_limits<ValueType>::infinity()) || (omega != omega)
               || (omega == static_cast<ValueType>(0)))
            {
                LOG_INFO("BiCGStab omega == 0 || Nan || Inf !!! Updated solution only in p-direction");
                x->AddScale(*p, alpha);
                op->Apply(*x, p);
                p->ScaleAdd(-one, rhs);
                res_norm = this->Norm_(*p);
                this->iter_ctrl_.CheckResidual(std::abs(res_norm), this->index_);
                break;
            }
            x->ScaleAdd2(one, *dir, alpha, *sv, omega);
            r->AddScale(*t, -omega);
            res_norm = this->Norm_(*r);
            if(this->iter_ctrl_.CheckResidual(std::abs(res_norm), this->index_))
                break;
            rho_old = rho;
            rho     = r0->Dot(*r);
            if(rho == static_cast<ValueType>(0))
            {
                LOG_INFO("BiCGStab rho == 0 !!!");
                break;
            }
            beta = (rho / rho_old) * (alpha / omega);
            p->ScaleAdd2(beta, *q, -beta * omega, *r, one);
            if(precond)
                this->precond_->SolveZeroSol(*p, z);
        }
    }
    VectorType r_, r0_, p_, q_, t_, v_, z_;
};

// ============================================================================ FCG
// src/solvers/krylov/fcg.cpp:232-318 / :321-420 (flexible CG).  InitResidual's verdict is not consulted.
template <class OperatorType, class VectorType, typename ValueType>
class FCG : public IterativeLinearSolver<OperatorType, VectorType, ValueType>
{
public:
    virtual ~FCG()
    {
        this->Clear();
    }
    virtual void Print(void) const
    {
        LOG_INFO((this->precond_ ? "Flexible PCG solver, with preconditioner" : "Flexible CG (non-precond) solver"));
    }
    virtual void Build(void)
    {
        if(this->build_)
            this->Clear();
        assert(this->op_ != NULL && this->op_->GetM() == this->op_->GetN() && this->op_->GetM() > 0);
        this->build_ = true;
        if(this->precond_ != NULL)
        {
            this->precond_->SetOperator(*this->op_);
            this->precond_->Build();
            this->z_.CloneBackend(*this->op_);
            this->z_.Allocate("z", this->op_->GetM());
        }
        VectorType* all[] = {&this->r_, &this->w_, &this->p_, &this->q_};
        for(VectorType* vec : all)
        {
            vec->CloneBackend(*this->op_);
            vec->Allocate("fcg", this->op_->GetM());
        }
    }
    virtual void Clear(void)
    {
        if(this->build_)
        {
            if(this->precond_ != NULL)
            {
                this->precond_->Clear();
                this->precond_ = NULL;
            }
            VectorType* all[] = {&this->r_, &this->w_, &this->p_, &this->q_, &this->z_};
            for(VectorType* vec : all)
                vec->Clear();
            this->iter_ctrl_.Clear();
            this->build_ = false;
        }
    }

protected:
    virtual void PrintStart_(void) const
    {
        LOG_INFO((this->precond_ ? "Flexible PCG solver starts, with preconditioner:" : "Flexible CG (non-precond) linear solver starts"));
    }
    virtual void PrintEnd_(void) const
    {
        LOG_INFO((this->precond_ ? "Flexible PCG ends" : "Flexible CG (non-precond) ends"));
    }
    virtual void SolveNonPrecond_(const VectorType& rhs, VectorType* x)
    {
        this->Solve_(rhs, x, false);
    }
    virtual void SolvePrecond_(const VectorType& rhs, VectorType* x)
    {
        this->Solve_(rhs, x, true);
    }

private:
    void Solve_(const VectorType& rhs, VectorType* x, bool precond)
    {
        const OperatorType* op = this->op_;
        VectorType *r = &this->r_, *w = &this->w_, *p = &this->p_, *q = &this->q_;
        VectorType* z = precond ? &this->z_ : r; // without a preconditioner z IS r
        ValueType   alpha, beta, rho, gamma, gamma_rho;
        op->Apply(*x, r);
        r->ScaleAdd(static_cast<ValueType>(-1), rhs);
        ValueType res = this->Norm_(*r);
        this->iter_ctrl_.InitResidual(std::abs(res));
        if(precond)
            this->precond_->SolveZeroSol(*r, z);
        op->Apply(*z, w);
        alpha = z->Dot(*r);
        beta  = z->Dot(*w);
        p->CopyFrom(*z);
        q->CopyFrom(*w);
        rho = beta;
        x->AddScale(*p, alpha / rho);
        r->AddScale(*q, -alpha / rho);
        res = this->Norm_(*r);
        while(!this->iter_ctrl_.CheckResidual(std::abs(res), this->index_))
        {
            if(precond)
                this->precond_->SolveZeroSol(*r, z);
            op->Apply(*z, w);
            beta      = z->Dot(*w);
            gamma     = z->Dot(*q);
            gamma_rho = -gamma / rho;
            p->ScaleAdd(gamma_rho, *z);
            q->ScaleAdd(gamma_rho, *w);
            rho   = beta + gamma * gamma_rho;
            alpha = z->Dot(*r) / rho;
            x->AddScale(*p, alpha);
            r->AddScale(*q, -alpha);
            res = this->Norm_(*r);
        }
    }
    VectorType r_, w_, z_, p_, q_;
};

// ============================================================================ CR
// src/solvers/krylov/cr.cpp:240-318 / :321-430 (conjugate residual; the preconditioned variant tests
// convergence on t, the unpreconditioned residual)
template <class OperatorType, class VectorType, typename ValueType>
class CR : public IterativeLinearSolver<OperatorType, VectorType, ValueType>
{
public:
    virtual ~CR()
    {
        this->Clear();
    }
    virtual void Print(void) const
    {
        LOG_INFO((this->precond_ ? "PCR solver, with preconditioner" : "CR (non-precond) solver"));
    }
    virtual void Build(void)
    {
        if(this->build_)
            this->Clear();
        assert(this->op_ != NULL && this->op_->GetM() == this->op_->GetN() && this->op_->GetM() > 0);
        this->build_ = true;
        if(this->precond_ != NULL)
        {
            this->precond_->SetOperator(*this->op_);
            this->precond_->Build();
            this->z_.CloneBackend(*this->op_);
            this->z_.Allocate("z", this->op_->GetM());
            this->t_.CloneBackend(*this->op_);
            this->t_.Allocate("t", this->op_->GetM());
        }
        VectorType* all[] = {&this->r_, &this->p_, &this->q_, &this->v_};
        for(VectorType* vec : all)
        {
            vec->CloneBackend(*this->op_);
            vec->Allocate("cr", this->op_->GetM());
        }
    }
    virtual void Clear(void)
    {
        if(this->build_)
        {
            if(this->precond_ != NULL)
            {
                this->precond_->Clear();
                this->precond_ = NULL;
            }
            VectorType* all[] = {&this->r_, &this->p_, &this->q_, &this->v_, &this->z_, &this->t_};
            for(VectorType* vec : all)
                vec->Clear();
            this->iter_ctrl_.Clear();
            this->build_ = false;
        }
    }

protected:
    virtual void PrintStart_(void) const
    {
        LOG_INFO((this->precond_ ? "PCR solver starts, with preconditioner:" : "CR (non-precond) linear solver starts"));
    }
    virtual void PrintEnd_(void) const
    {
        LOG_INFO((this->precond_ ? "PCR ends" : "CR (non-precond) ends"));
    }
    virtual void SolveNonPrecond_(const VectorType& rhs, VectorType* x)
    {
        const OperatorType* op = this->op_;
        VectorType *r = &this->r_, *p = &this->p_, *q = &this->q_, *v = &this->v_;
        ValueType   alpha, beta, rho, rho_old;
        op->Apply(*x, r);
        r->ScaleAdd(static_cast<ValueType>(-1), rhs);
        p->CopyFrom(*r);
        ValueType res_norm = this->Norm_(*r);
        if(this->iter_ctrl_.InitResidual(std::abs(res_norm)) == false)
            return;
        op->Apply(*r, v);
        rho = r->DotNonConj(*v);
        op->Apply(*p, q);
        alpha = rho / q->DotNonConj(*q);
        x->AddScale(*p, alpha);
        r->AddScale(*q, -alpha);
        res_norm = this->Norm_(*r);
        while(!this->iter_ctrl_.CheckResidual(std::abs(res_norm), this->index_))
        {
            rho_old = rho;
            op->Apply(*r, v);
            rho  = r->DotNonConj(*v);
            beta = rho / rho_old;
            p->ScaleAdd(beta, *r);
            q->ScaleAdd(beta, *v);
            alpha = rho / q->DotNonConj(*q);
            x->AddScale(*p, alpha);
            r->AddScale(*q, -alpha);
            res_norm = this->Norm_(*r);
        }
    }
    virtual void SolvePrecond_(const VectorType& rhs, VectorType* x)
    {
        const OperatorType* op = this->op_;
        VectorType *r = &this->r_, *z = &this->z_, *p = &this->p_, *q = &this->q_, *v = &this->v_, *t = &this->t_;
        ValueType   alpha, beta, rho, rho_old;
        op->Apply(*x, z);
        z->ScaleAdd(static_cast<ValueType>(-1), rhs);
        this->precond_->SolveZeroSol(*z, r);
        p->CopyFrom(*r);
        t->CopyFrom(*z);
        ValueType res_norm = this->Norm_(*t);
        if(this->iter_ctrl_.InitResidual(std::abs(res_norm)) == false)
            return;
        op->Apply(*r, v);
        rho = r->DotNonConj(*v);
        op->Apply(*p, q);
        this->precond_->SolveZeroSol(*q, z);
        alpha = rho / q->DotNonConj(*z);
        x->AddScale(*p, alpha);
        r->AddScale(*z, -alpha);
        t->AddScale(*q, -alpha);
        res_norm = this->Norm_(*t);
        while(!this->iter_ctrl_.CheckResidual(std::abs(res_norm), this->index_))
        {
            rho_old = rho;
            op->Apply(*r, v);
            rho  = r->DotNonConj(*v);
            beta = rho / rho_old;
            p->ScaleAdd(beta, *r);
            q->ScaleAdd(beta, *v);
            this->precond_->SolveZeroSol(*q, z);
            alpha = rho / q->DotNonConj(*z);
            x->AddScale(*p, alpha);
            r->AddScale(*z, -alpha);
            t->AddScale(*q, -alpha);
            res_norm = this->Norm_(*t);
        }
    }

private:
    VectorType r_, z_, p_, q_, v_, t_;
};

// ============================================================================ BiCGStab(l)
// src/solvers/krylov/bicgstabl.cpp:292-496 / :499-695.  l = 2 by default (SetOrder).  The
// preconditioned variant applies M^-1 after every operator product (left preconditioning) and tests
// the preconditioned residual.  One "iteration" is one outer sweep (l BiCG steps + the MR part).
template <class OperatorType, class VectorType, typename ValueType>
class BiCGStabl : public IterativeLinearSolver<OperatorType, VectorType, ValueType>
{
public:
    BiCGStabl()
        : l_(2)
        , r_(NULL)
        , u_(NULL)
    {
    }
    virtual ~BiCGStabl()
    {
        this->Clear();
    }
    virtual void Print(void) const
    {
        LOG_INFO("BiCGStab(" << this->l_ << ") solver" << (this->precond_ ? ", with preconditioner" : " (non-precond)"));
    }
    virtual void SetOrder(int l)
    {
        assert(l > 0 && this->build_ == false);
        this->l_ = l;
    }
    virtual void Build(void)
    {
        if(this->build_)
            this->Clear();
        assert(this->op_ != NULL && this->op_->GetM() == this->op_->GetN() && this->op_->GetM() > 0);
        this->build_ = true;
        if(this->precond_ != NULL)
        {
            this->precond_->SetOperator(*this->op_);
            this->precond_->Build();
            this->z_.CloneBackend(*this->op_);
            this->z_.Allocate("z", this->op_->GetM());
        }
        this->r0_.CloneBackend(*this->op_);
        this->r0_.Allocate("r0", this->op_->GetM());
        const int l = this->l_;
        this->r_    = new VectorType*[l + 1];
        this->u_    = new VectorType*[l + 1];
        for(int i = 0; i < l + 1; ++i)
        {
            this->r_[i] = new VectorType;
            this->r_[i]->CloneBackend(*this->op_);
            this->r_[i]->Allocate("r", this->op_->GetM());
            this->u_[i] = new VectorType;
            this->u_[i]->CloneBackend(*this->op_);
            this->u_[i]->Allocate("u", this->op_->GetM());
        }
        this->gamma0_.assign((size_t)l, ValueType(0));
        this->gamma1_.assign((size_t)l, ValueType(0));
        this->gamma2_.assign((size_t)l, ValueType(0));
        this->sigma_.assign((size_t)l, ValueType(0));
        this->tau_.assign((size_t)l * l, ValueType(0));
    }
    virtual void Clear(void)
    {
        if(this->build_)
        {
            this->r0_.Clear();
            for(int i = 0; i < this->l_ + 1; ++i)
            {
                delete this->r_[i];
                delete this->u_[i];
            }
            delete[] this->r_;
            delete[] this->u_;
            this->r_ = this->u_ = NULL;
            if(this->precond_ != NULL)
            {
                this->precond_->Clear();
                this->precond_ = NULL;
                this->z_.Clear();
            }
            this->iter_ctrl_.Clear();
            this->build_ = false;
        }
    }

protected:
    virtual void PrintStart_(void) const
    {
        LOG_INFO("BiCGStab(" << this->l_ << ") " << (this->precond_ ? "" : "(non-precond) ") << "linear solver starts");
    }
    virtual void PrintEnd_(void) const
    {
        LOG_INFO("BiCGStab(" << this->l_ << ") ends");
    }
    virtual void SolveNonPrecond_(const VectorType& rhs, VectorType* x)
    {
        this->Solve_(rhs, x, false);
    }
    virtual void SolvePrecond_(const VectorType& rhs, VectorType* x)
    {
        this->Solve_(rhs, x, true);
    }

private:
    // y = A in  (followed by M^-1 when preconditioned)
    void ApplyPrec_(const VectorType& in, VectorType* out, bool precond)
    {
        if(precond)
        {
            this->op_->Apply(in, &this->z_);
            this->precond_->SolveZeroSol(this->z_, out);
        }
        else
            this->op_->Apply(in, out);
    }
    void Solve_(const VectorType& rhs, VectorType* x, bool precond)
    {
        VectorType*  r0 = &this->r0_;
        VectorType** r  = this->r_;
        VectorType** u  = this->u_;
        const int    l  = this->l_;
        bool         converged = false;
        ValueType    alpha = static_cast<ValueType>(0), beta = static_cast<ValueType>(0);
        ValueType    omega = static_cast<ValueType>(1), rho_old = static_cast<ValueType>(-1), rho;
        ValueType *  gamma0 = this->gamma0_.data(), *gamma1 = this->gamma1_.data();
        ValueType *  gamma2 = this->gamma2_.data(), *sigma = this->sigma_.data(), *tau = this->tau_.data();
        const ValueType zero = static_cast<ValueType>(0);
        if(precond)
        {
            this->op_->Apply(*x, &this->z_);
            this->z_.ScaleAdd(static_cast<ValueType>(-1), rhs);
            this->precond_->SolveZeroSol(this->z_, r0);
        }
        else
        {
            this->op_->Apply(*x, r0);
            r0->ScaleAdd(static_cast<ValueType>(-1), rhs);
        }
        ValueType res = this->Norm_(*r0);
        this->iter_ctrl_.InitResidual(std::abs(res));
        r[0]->CopyFrom(*r0);
        u[0]->Zeros();
        while(true)
        {
            rho_old *= -omega;
            for(int j = 0; j < l; ++j) // BiCG part
            {
                rho = r0->Dot(*r[j]);
                if(rho == zero)
                {
                    LOG_INFO("BiCGStab(l) rho == 0 !!!");
                    converged = true;
                    break;
                }
                beta = alpha * rho / rho_old;
                for(int i = 0; i <= j; ++i)
                    u[i]->ScaleAdd(-beta, *r[i]);
                this->ApplyPrec_(*u[j], u[j + 1], precond);
                rho_old = r0->Dot(*u[j + 1]);
                if(rho_old == zero)
                {
                    LOG_INFO("BiCGStab(l) sigma == 0 !!!");
                    converged = true;
                    break;
                }
                alpha   = rho / rho_old;
                rho_old = rho;
                for(int i = 0; i <= j; ++i)
                    r[i]->AddScale(*u[i + 1], -alpha);
                this->ApplyPrec_(*r[j], r[j + 1], precond);
                x->AddScale(*u[0], alpha);
                res = this->Norm_(*r[0]);
                if(this->iter_ctrl_.CheckResidualNoCount(std::abs(res)))
                {
                    converged = true;
                    break;
                }
            }
            if(converged)
                break;
            for(int j = 0; j < l; ++j) // modified Gram-Schmidt (MR part)
            {
                for(int i = 0; i < j; ++i)
                {
                    tau[i * l + j] = r[j + 1]->Dot(*r[i + 1]) / sigma[i];
                    r[j + 1]->AddScale(*r[i + 1], -tau[i * l + j]);
                }
                sigma[j]  = r[j + 1]->Dot(*r[j + 1]);
                gamma1[j] = r[0]->Dot(*r[j + 1]) / sigma[j];
            }
            gamma0[l - 1] = gamma1[l - 1];
            omega         = gamma1[l - 1];
            for(int j = l - 2; j >= 0; --j)
            {
                gamma0[j] = gamma1[j];
                for(int i = j + 1; i < l; ++i)
                    gamma0[j] -= tau[j * l + i] * gamma0[i];
            }
            for(int j = 0; j < l - 1; ++j)
            {
                gamma2[j] = gamma0[j + 1];
                for(int i = j + 1; i < l - 1; ++i)
                    gamma2[j] += tau[j * l + i] * gamma0[i + 1];
            }
            x->AddScale(*r[0], gamma0[0]);
            r[0]->AddScale(*r[l], -gamma1[l - 1]);
            u[0]->AddScale(*u[l], -gamma0[l - 1]);
            for(int j = 1; j < l; ++j)
            {
                u[0]->AddScale(*u[j], -gamma0[j - 1]);
                x->AddScale(*r[j], gamma2[j - 1]);
                r[0]->AddScale(*r[j], -gamma1[j - 1]);
            }
            res = this->Norm_(*r[0]);
            if(this->iter_ctrl_.CheckResidual(std::abs(res), this->index_))
                break;
        }
    }
    int                    l_;
    VectorType             r0_, z_;
    VectorType**           r_;
    VectorType**           u_;
    std::vector<ValueType> gamma0_, gamma1_, gamma2_, sigma_, tau_;
};

// ============================================================================ QMRCGStab
// src/solvers/krylov/qmrcgstab.cpp:262-460 / :463-690.  The iteration control sees the bound
// sqrt(#iter+1)*|tau|; after the loop the TRUE residual is computed and checked once more (one more
// counted iteration).  p is only zero-filled by Build(), as in the reference.
template <class OperatorType, class VectorType, typename ValueType>
class QMRCGStab : public IterativeLinearSolver<OperatorType, VectorType, ValueType>
{
public:
    virtual ~QMRCGStab()
    {
        this->Clear();
    }
    virtual void Print(void) const
    {
        LOG_INFO("QMRCGStab solver" << (this->precond_ ? ", with preconditioner" : " (non-precond)"));
    }
    virtual void Build(void)
    {
        if(this->build_)
            this->Clear();
        assert(this->op_ != NULL && this->op_->GetM() == this->op_->GetN() && this->op_->GetM() > 0);
        this->build_ = true;
        if(this->precond_ != NULL)
        {
            this->precond_->SetOperator(*this->op_);
            this->precond_->Build();
            this->z_.CloneBackend(*this->op_);
            this->z_.Allocate("z", this->op_->GetM());
        }
        VectorType* all[] = {&this->r0_, &this->r_, &this->p_, &this->t_, &this->v_, &this->d_};
        for(VectorType* vec : all)
        {
            vec->CloneBackend(*this->op_);
            vec->Allocate("qmrcgstab", this->op_->GetM());
        }
    }
    virtual void Clear(void)
    {
        if(this->build_)
        {
            if(this->precond_ != NULL)
            {
                this->precond_->Clear();
                this->precond_ = NULL;
            }
            VectorType* all[] = {&this->r0_, &this->r_, &this->p_, &this->t_, &this->v_, &this->d_, &this->z_};
            for(VectorType* vec : all)
                vec->Clear();
            this->iter_ctrl_.Clear();
            this->build_ = false;
        }
    }

protected:
    virtual void PrintStart_(void) const
    {
        LOG_INFO("QMRCGStab " << (this->precond_ ? "" : "(non-precond) ") << "linear solver starts");
    }
    virtual void PrintEnd_(void) const
    {
        LOG_INFO("QMRCGStab ends");
    }
    virtual void SolveNonPrecond_(const VectorType& rhs, VectorType* x)
    {
        this->Solve_(rhs, x, false);
    }
    virtual void SolvePrecond_(const VectorType& rhs, VectorType* x)
    {
        this->Solve_(rhs, x, true);
    }

private:
    void Solve_(const VectorType& rhs, VectorType* x, bool precond)
    {
        const OperatorType* op = this->op_;
        VectorType *r0 = &this->r0_, *r = &this->r_, *p = &this->p_, *t = &this->t_, *v = &this->v_, *d = &this->d_;
        VectorType* z = &this->z_;
        VectorType* pz = precond ? z : p; // what A is applied to in the first half step
        VectorType* rz = precond ? z : r; // ... and in the second
        const ValueType one = static_cast<ValueType>(1), zero = static_cast<ValueType>(0);
        ValueType alpha, beta, omega, theta1, theta1sq, theta2, theta2sq, eta1, eta2, tau1, tau2, rho, rho_old, c;
        op->Apply(*x, r0);
        r0->ScaleAdd(-one, rhs);
        r->CopyFrom(*r0);
        tau2            = this->Norm_(*r0);
        double res_norm = std::abs(tau2);
        this->iter_ctrl_.InitResidual(res_norm);
        rho  = r0->Dot(*r);
        beta = rho;
        (void)beta;
        p->AddScale(*r, one);
        if(precond)
            this->precond_->SolveZeroSol(*p, z);
        op->Apply(*pz, v);
        rho_old = r0->Dot(*v);
        alpha   = rho / rho_old;
        r->AddScale(*v, -alpha);
        theta1   = this->Norm_(*r) / tau2;
        theta1sq = theta1 * theta1;
        c        = one / std::sqrt(one + theta1sq);
        tau1     = tau2 * theta1 * c;
        eta1     = c * c * alpha;
        d->CopyFrom(*pz);
        x->AddScale(*d, eta1);
        if(precond)
            this->precond_->SolveZeroSol(*r, z);
        op->Apply(*rz, t);
        omega = t->Dot(*r) / t->Dot(*t);
        d->ScaleAdd(theta1sq * eta1 / omega, *rz);
        r->AddScale(*t, -omega);
        theta2   = this->Norm_(*r) / tau1;
        theta2sq = theta2 * theta2;
        c        = one / std::sqrt(one + theta2sq);
        tau2     = tau1 * theta2 * c;
        eta2     = c * c * omega;
        x->AddScale(*d, eta2);
        res_norm = std::sqrt(static_cast<double>(this->iter_ctrl_.GetIterationCount() + 1)) * std::abs(tau2);
        while(!this->iter_ctrl_.CheckResidual(res_norm, this->index_))
        {
            rho_old = rho;
            rho     = r0->Dot(*r);
            beta    = (rho * alpha) / (rho_old * omega);
            p->AddScale(*v, -omega);
            p->Scale(beta);
            p->AddScale(*r, one);
            if(precond)
                this->precond_->SolveZeroSol(*p, z);
            op->Apply(*pz, v);
            rho_old = r0->Dot(*v);
            if(rho_old == zero)
            {
                LOG_INFO("QMRCGStab break rho_old == 0 !!!");
                break;
            }
            alpha = rho / rho_old;
            r->AddScale(*v, -alpha);
            theta1   = this->Norm_(*r) / tau2;
            theta1sq = theta1 * theta1;
            c        = one / std::sqrt(one + theta1sq);
            tau1     = tau2 * theta1 * c;
            eta1     = c * c * alpha;
            d->ScaleAdd(theta2sq * eta2 / alpha, *pz);
            x->AddScale(*d, eta1);
            if(precond)
                this->precond_->SolveZeroSol(*r, z);
            op->Apply(*rz, t);
            omega = t->Dot(*t);
            if(omega == zero)
            {
                LOG_INFO("QMRCGStab omega == 0 !!!");
                break;
            }
            omega = t->Dot(*r) / omega;
            d->ScaleAdd(theta1sq * eta1 / omega, *rz);
            r->AddScale(*t, -omega);
            theta2   = this->Norm_(*r) / tau1;
            theta2sq = theta2 * theta2;
            c        = one / std::sqrt(one + theta2sq);
            tau2     = tau1 * theta2 * c;
            eta2     = c * c * omega;
            x->AddScale(*d, eta2);
            res_norm = std::sqrt(static_cast<double>(this->iter_ctrl_.GetIterationCount() + 1)) * std::abs(tau2);
        }
        op->Apply(*x, r0);
        r0->ScaleAdd(-one, rhs);
        this->iter_ctrl_.CheckResidual(std::abs(this->Norm_(*r0)));
    }
    VectorType r0_, r_, p_, t_, v_, d_, z_;
};

// ============================================================================ IDR(s)
// src/solvers/krylov/idr.cpp: Build :127-185 (shadow space: s random-normal vectors, seed (i+1)*seed_,
// made orthonormal by modified Gram-Schmidt), SolveNonPrecond_ :335-520, SolvePrecond_ :523-730.
// The default seed is time(NULL) as in the reference: call SetRandomSeed for reproducible runs.
template <class OperatorType, class VectorType, typename ValueType>
class IDR : public IterativeLinearSolver<OperatorType, VectorType, ValueType>
{
public:
    IDR()
        : s_(4)
        , seed_((unsigned long long)time(NULL))
        , kappa_(static_cast<ValueType>(0.7f))
        , G_(NULL)
        , U_(NULL)
        , P_(NULL)
    {
    }
    virtual ~IDR()
    {
        this->Clear();
    }
    virtual void Print(void) const
    {
        LOG_INFO("IDR(" << this->s_ << ") solver" << (this->precond_ ? ", with preconditioner" : " (non-precond)"));
    }
    void SetShadowSpace(int s)
    {
        assert(this->build_ == false && s > 0);
        this->s_ = s;
    }
    void SetRandomSeed(unsigned long long seed)
    {
        assert(this->build_ == false && seed > 0ULL);
        this->seed_ = seed;
    }
    virtual void Build(void)
    {
        if(this->build_)
            this->Clear();
        assert(this->op_ != NULL && this->op_->GetM() == this->op_->GetN() && this->op_->GetM() > 0);
        assert((int64_t)this->s_ <= this->op_->GetM());
        const int s = this->s_;
        this->r_.CloneBackend(*this->op_);
        this->v_.CloneBackend(*this->op_);
        this->r_.Allocate("r", this->op_->GetM());
        this->v_.Allocate("v", this->op_->GetM());
        this->c_.assign((size_t)s, ValueType(0));
        this->f_.assign((size_t)s, ValueType(0));
        this->M_.assign((size_t)s * s, ValueType(0));
        this->G_ = new VectorType*[s];
        this->U_ = new VectorType*[s];
        this->P_ = new VectorType*[s];
        for(int i = 0; i < s; ++i)
        {
            this->G_[i] = new VectorType;
            this->U_[i] = new VectorType;
            this->P_[i] = new VectorType;
            this->G_[i]->CloneBackend(*this->op_);
            this->U_[i]->CloneBackend(*this->op_);
            this->P_[i]->CloneBackend(*this->op_);
            this->G_[i]->Allocate("g", this->op_->GetM());
            this->U_[i]->Allocate("u", this->op_->GetM());
            this->P_[i]->Allocate("P", this->op_->GetM());
            this->P_[i]->SetRandomNormal((unsigned long long)(i + 1) * this->seed_, 0.0, 1.0);
        }
        if(this->precond_ != NULL)
        {
            this->precond_->SetOperator(*this->op_);
            this->precond_->Build();
            this->t_.CloneBackend(*this->op_);
            this->t_.Allocate("t", this->op_->GetM());
        }
        for(int k = 0; k < s; ++k) // orthonormal basis of the shadow space (modified Gram-Schmidt)
        {
            this->P_[k]->Scale(static_cast<ValueType>(1) / this->P_[k]->Norm());
            ValueType invdotk = static_cast<ValueType>(1) / this->P_[k]->Dot(*this->P_[k]);
            for(int j = k + 1; j < s; ++j)
                this->P_[j]->AddScale(*this->P_[k], -this->P_[j]->Dot(*this->P_[k]) * invdotk);
        }
        this->build_ = true;
    }
    virtual void Clear(void)
    {
        if(this->build_)
        {
            this->r_.Clear();
            this->v_.Clear();
            this->t_.Clear();
            for(int i = 0; i < this->s_; ++i)
            {
                delete this->U_[i];
                delete this->G_[i];
                delete this->P_[i];
            }
            delete[] this->U_;
            delete[] this->G_;
            delete[] this->P_;
            this->U_ = this->G_ = this->P_ = NULL;
            if(this->precond_ != NULL)
            {
                this->precond_->Clear();
                this->precond_ = NULL;
            }
            this->iter_ctrl_.Clear();
            this->build_ = false;
        }
    }

protected:
    virtual void PrintStart_(void) const
    {
        LOG_INFO((this->precond_ ? "PIDR(" : "IDR(") << this->s_ << (this->precond_ ? ") solver starts, with preconditioner:" : ") (non-precond) linear solver starts"));
    }
    virtual void PrintEnd_(void) const
    {
        LOG_INFO((this->precond_ ? "PIDR(" : "IDR(") << this->s_ << (this->precond_ ? ") ends" : ") (non-precond) ends"));
    }
    virtual void SolveNonPrecond_(const VectorType& rhs, VectorType* x)
    {
        this->Solve_(rhs, x, false);
    }
    virtual void SolvePrecond_(const VectorType& rhs, VectorType* x)
    {
        this->Solve_(rhs, x, true);
    }

private:
    int mind_(int i, int j) const // DENSE_IND(i, j, s, s), column-major
    {
        return i + j * this->s_;
    }
    void Breakdown_(const char* what) const
    {
        LOG_INFO("IDR(s) break down ; " << what);
        FATAL_ERROR(__FILE__, __LINE__);
    }
    void CheckScalar_(ValueType val, const char* z, const char* nan, const char* inf) const
    {
        if(val == static_cast<ValueType>(0))
            this->Breakdown_(z);
        if(val != val)
            this->Breakdown_(nan);
        if(val == std::numeric_limits<ValueType>::infinity())
            this->Breakdown_(inf);
    }
    void Solve_(const VectorType& rhs, VectorType* x, bool precond)
    {
        const OperatorType* op = this->op_;
        VectorType *        r = &this->r_, *v = &this->v_, *t = &this->t_;
        VectorType **       G = this->G_, **U = this->U_, **P = this->P_;
        const int           s = this->s_;
        const ValueType     zero = static_cast<ValueType>(0), one = static_cast<ValueType>(1), kappa = this->kappa_;
        ValueType *         c = this->c_.data(), *f = this->f_.data(), *M = this->M_.data();
        ValueType           alpha, beta, rho, omega = one;
        op->Apply(*x, r);
        r->ScaleAdd(-one, rhs);
        ValueType res_norm = this->Norm_(*r);
        if(this->iter_ctrl_.InitResidual(std::abs(res_norm)) == false)
            return;
        for(int i = 0; i < s; ++i)
        {
            G[i]->Zeros();
            U[i]->Zeros();
            for(int j = 0; j < s; ++j)
                M[this->mind_(i, j)] = (i == j) ? one : zero;
        }
        while(true)
        {
            for(int i = 0; i < s; ++i) // f = P^T r
                f[i] = P[i]->Dot(*r);
            for(int k = 0; k < s; ++k) // loop over the shadow space
            {
                v->CopyFrom(*r);
                for(int i = k; i < s; ++i) // lower triangular system M c = f
                {
                    c[i] = f[i];
                    for(int j = k; j < i; ++j)
                        c[i] -= M[this->mind_(i, j)] * c[j];
                    c[i] /= M[this->mind_(i, i)];
                    v->AddScale(*G[i], -c[i]);
                }
                if(precond)
                {
                    this->precond_->SolveZeroSol(*v, t);
                    U[k]->ScaleAddScale(c[k], *t, omega);
                }
                else
                    U[k]->ScaleAddScale(c[k], *v, omega);
                for(int i = k + 1; i < s; ++i)
                    U[k]->AddScale(*U[i], c[i]);
                op->Apply(*U[k], G[k]);
                for(int i = 0; i < k; ++i) // make G_k orthogonal to P
                {
                    alpha = P[i]->Dot(*G[k]) / M[this->mind_(i, i)];
                    G[k]->AddScale(*G[i], -alpha);
                    U[k]->AddScale(*U[i], -alpha);
                }
                for(int i = k; i < s; ++i)
                    M[this->mind_(i, k)] = P[i]->Dot(*G[k]);
                this->CheckScalar_(M[this->mind_(k, k)], "M(k,k) == 0.0", "M(k,k) == NaN", "M(k,k) == inf");
                beta = f[k] / M[this->mind_(k, k)];
                r->AddScale(*G[k], -beta);
                x->AddScale(*U[k], beta);
                res_norm = this->Norm_(*r);
                if(this->iter_ctrl_.CheckResidualNoCount(std::abs(res_norm)))
                    break;
                for(int i = k + 1; i < s; ++i)
                    f[i] -= beta * M[this->mind_(i, k)];
            }
            if(this->iter_ctrl_.CheckResidual(std::abs(res_norm), this->index_))
                break;
            ValueType rt, nt; // dimension reduction step
            if(precond)
            {
                this->precond_->SolveZeroSol(*r, v);
                op->Apply(*v, t);
                rt = t->Dot(*r);
                nt = t->Norm();
            }
            else
            {
                op->Apply(*r, v);
                rt = v->Dot(*r);
                nt = v->Norm();
            }
            rt /= nt;
            rho   = std::abs(rt / res_norm);
            omega = rt / nt;
            if(rho < kappa)
                omega *= kappa / rho;
            this->CheckScalar_(omega, "w == 0.0", "w == NaN", "w == inf");
            if(precond)
            {
                r->AddScale(*t, -omega);
                x->AddScale(*v, omega);
            }
            else
            {
                x->AddScale(*r, omega);
                r->AddScale(*v, -omega);
            }
            res_norm = this->Norm_(*r);
        }
    }
    int                    s_;
    unsigned long long     seed_;
    ValueType              kappa_;
    VectorType             r_, v_, t_;
    VectorType**           G_;
    VectorType**           U_;
    VectorType**           P_;
    std::vector<ValueType> c_, f_, M_;
};

// ============================================================================ FixedPoint
// src/solvers/solver.cpp:517-775: x_{k+1} = x_k + omega M^-1 (b - A x_k); a preconditioner is mandatory.
// FlagSmoother(): exactly max_iter sweeps, no norms (the form multigrid uses).
template <class OperatorType, class VectorType, typename ValueType>
class FixedPoint : public IterativeLinearSolver<OperatorType, VectorType, ValueType>
{
public:
    FixedPoint()
        : omega_(static_cast<ValueType>(1))
    {
    }
    virtual ~FixedPoint()
    {
        this->Clear();
    }
    virtual void Print(void) const
    {
        LOG_INFO("Fixed Point Iteration solver, with preconditioner:");
        if(this->precond_)
            this->precond_->Print();
    }
    virtual void SetRelaxation(ValueType omega)
    {
        this->omega_ = omega;
    }
    virtual void Build(void)
    {
        if(this->build_)
            this->Clear();
        assert(this->op_ != NULL && this->precond_ != NULL);
        this->build_ = true;
        this->x_old_.CloneBackend(*this->op_);
        this->x_old_.Allocate("x_old", this->op_->GetM());
        this->x_res_.CloneBackend(*this->op_);
        this->x_res_.Allocate("x_res", this->op_->GetM());
        this->precond_->SetOperator(*this->op_);
        this->precond_->Build();
    }
    virtual void Clear(void)
    {
        if(this->build_)
        {
            if(this->precond_ != NULL)
            {
                this->precond_->Clear();
                this->precond_ = NULL;
            }
            this->x_old_.Clear();
            this->x_res_.Clear();
            this->iter_ctrl_.Clear();
            this->build_ = false;
        }
    }

protected:
    virtual void PrintStart_(void) const
    {
        LOG_INFO("Fixed Point Iteration solver starts");
    }
    virtual void PrintEnd_(void) const
    {
        LOG_INFO("Fixed Point Iteration solver ends");
    }
    virtual void SolveNonPrecond_(const VectorType&, VectorType*)
    {
        LOG_INFO("Preconditioner for the Fixed Point method is required");
        FATAL_ERROR(__FILE__, __LINE__);
    }
    virtual void SolvePrecond_(const VectorType& rhs, VectorType* x)
    {
        const ValueType one = static_cast<ValueType>(1);
        if(this->is_smoother_)
        {
            const int steps = this->iter_ctrl_.GetMaximumIterations();
            if(steps < 1)
                return;
            this->iter_ctrl_.InitResidual(1.0); // dummy: the smoother never looks at a residual
            if(this->fused_ && this->FusedJacobiSweeps_(rhs, x, steps))
                return;
            for(int iter = 0; iter < steps; ++iter)
            {
                this->op_->Apply(*x, &this->x_res_);
                this->x_res_.ScaleAdd(-one, rhs);
                this->precond_->SolveZeroSol(this->x_res_, &this->x_old_);
                x->AddScale(this->x_old_, this->omega_);
            }
            return;
        }
        if(this->iter_ctrl_.GetMaximumIterations() < 1)
            return;
        this->op_->Apply(*x, &this->x_res_);
        this->x_res_.ScaleAdd(-one, rhs);
        ValueType res = this->Norm_(this->x_res_);
        if(this->iter_ctrl_.InitResidual(std::abs(res)) == false)
            return;
        while(true)
        {
            this->precond_->SolveZeroSol(this->x_res_, &this->x_old_);
            x->AddScale(this->x_old_, this->omega_);
            if(this->iter_ctrl_.CheckMaximumIterNoCount()) // the last residual is never needed
                break;
            this->op_->Apply(*x, &this->x_res_);
            this->x_res_.ScaleAdd(-one, rhs);
            res = this->Norm_(this->x_res_);
            if(this->iter_ctrl_.CheckResidual(std::abs(res), this->index_))
                break;
        }
    }

private:
    // FixedPoint + Jacobi as a smoother on a Local CSR operator: every sweep is ONE pass (SpMV with the update as its
    // epilogue, ramd_fused_jacobi_sweep) plus the copy back, instead of SpMV + three vector kernels; same operations
    template <class O = OperatorType, class V = VectorType>
    typename std::enable_if<_fusable<O, V, ValueType>::value, bool>::type
        FusedJacobiSweeps_(const VectorType& rhs, VectorType* x, int steps)
    {
        typedef Jacobi<OperatorType, VectorType, ValueType> JacobiType;
        JacobiType* jac = dynamic_cast<JacobiType*>(this->precond_);
        if(jac == NULL || !this->op_->is_accel_() || !x->is_accel_() || this->op_->GetFormat() != CSR
           || jac->GetInverseDiagonal().GetSize() != x->GetSize())
            return false;
        for(int iter = 0; iter < steps; ++iter)
        {
            int s = ramd_fused_jacobi_sweep(this->op_->handle(), _fh(jac->GetInverseDiagonal()), _fh(rhs), _fh(*x),
                                            _fh(this->x_res_), (double)this->omega_);
            if(s == RAMD_ERR_UNSUPPORTED)
            {
                if(iter == 0)
                    return false;
                FATAL_ERROR(__FILE__, __LINE__);
            }
            RAMD_CHECK(s);
            x->CopyFrom(this->x_res_);
        }
        return true;
    }
    template <class O = OperatorType, class V = VectorType>
    typename std::enable_if<!_fusable<O, V, ValueType>::value, bool>::type
        FusedJacobiSweeps_(const VectorType&, VectorType*, int)
    {
        return false;
    }

    ValueType  omega_;
    VectorType x_old_, x_res_;
};

// ============================================================================ MixedPrecisionDC
// mixed_precision.cpp:159-236 (Build) and :372-437 (solve).  The reference keeps the fp64 defect
// correction on the HOST and ships r / d across PCIe every outer step; here both levels live on the
// accelerator (cast kernels instead of host loops), the arithmetic and control flow are unchanged.
template <class OperatorTypeH, class VectorTypeH, typename ValueTypeH, class OperatorTypeL,
          class VectorTypeL, typename ValueTypeL>
class MixedPrecisionDC : public IterativeLinearSolver<OperatorTypeH, VectorTypeH, ValueTypeH>
{
public:
    MixedPrecisionDC()
        : Solver_L_(NULL)
        , op_l_(NULL)
    {
    }
    virtual ~MixedPrecisionDC()
    {
        this->Clear();
    }
    virtual void Print(void) const
    {
        LOG_INFO("MixedPrecisionDC [" << 8 * sizeof(ValueTypeH) << "bit-" << 8 * sizeof(ValueTypeL)
                                      << "bit] solver");
    }
    void Set(Solver<OperatorTypeL, VectorTypeL, ValueTypeL>& Solver_L)
    {
        this->Solver_L_ = &Solver_L;
    }
    virtual void Build(void)
    {
        if(this->build_)
            this->Clear();
        assert(this->Solver_L_ != NULL && this->op_ != NULL);
        this->build_ = true;
        this->op_l_  = new OperatorTypeL;
        this->op_l_->template CastFrom<ValueTypeH>(*this->op_); // value-cast CSR copy (:201-229)
        this->r_h_.CloneBackend(*this->op_);
        this->r_h_.Allocate("r_h", this->op_->GetM());
        this->d_h_.CloneBackend(*this->op_);
        this->d_h_.Allocate("d_h", this->op_->GetM());
        this->r_l_.CloneBackend(*this->op_);
        this->r_l_.Allocate("r_l", this->op_->GetM());
        this->d_l_.CloneBackend(*this->op_);
        this->d_l_.Allocate("d_l", this->op_->GetM());
        this->Solver_L_->SetOperator(*this->op_l_);
        this->Solver_L_->Build();
    }
    virtual void Clear(void)
    {
        if(this->build_)
        {
            if(this->Solver_L_ != NULL)
            {
                this->Solver_L_->Clear();
                this->Solver_L_ = NULL;
            }
            delete this->op_l_;
            this->op_l_ = NULL;
            this->r_h_.Clear();
            this->d_h_.Clear();
            this->r_l_.Clear();
            this->d_l_.Clear();
            this->iter_ctrl_.Clear();
            this->build_ = false;
        }
    }

protected:
    virtual void PrintStart_(void) const
    {
        LOG_INFO("MixedPrecisionDC linear solver starts");
    }
    virtual void PrintEnd_(void) const
    {
        LOG_INFO("MixedPrecisionDC ends");
    }
    virtual void SolveNonPrecond_(const VectorTypeH& rhs, VectorTypeH* x)
    {
        const ValueTypeH one = static_cast<ValueTypeH>(1);
        this->op_->Apply(*x, &this->r_h_);
        this->r_h_.ScaleAdd(-one, rhs);
        ValueTypeH res = this->Norm_(this->r_h_);
        if(this->iter_ctrl_.InitResidual(res) == false)
            return;
        while(!this->iter_ctrl_.CheckResidual(res, this->index_))
        {
            this->r_l_.CopyFromDouble(this->r_h_);
            this->d_l_.Zeros();
            this->Solver_L_->Solve(this->r_l_, &this->d_l_);
            this->d_h_.CopyFromFloat(this->d_l_);
            x->AddScale(this->d_h_, one);
            this->op_->Apply(*x, &this->r_h_);
            this->r_h_.ScaleAdd(-one, rhs);
            res = this->Norm_(this->r_h_);
        }
    }
    virtual void SolvePrecond_(const VectorTypeH&, VectorTypeH*)
    {
        LOG_INFO("MixedPrecisionDC:: the preconditioner belongs to the inner solver");
        FATAL_ERROR(__FILE__, __LINE__);
    }

private:
    Solver<OperatorTypeL, VectorTypeL, ValueTypeL>* Solver_L_;
    OperatorTypeL*                                  op_l_;
    VectorTypeH                                     r_h_, d_h_;
    VectorTypeL                                     r_l_, d_l_;
};

// ============================================================================ multigrid
// BaseMultiGrid (src/solvers/multigrid/base_multigrid.cpp): V / W / K cycles over a user- or AMG-built hierarchy of
// operators, restriction / prolongation operators, per-level smoothers and a coarse solver; optional scaling of the
// coarse correction (:790-812, :873-905).  Host levels (SetHostLevels) do not exist here: every level lives on the GPU.
enum _cycle
{
    Vcycle = 0,
    Wcycle = 1,
    Kcycle = 2,
    Fcycle = 3
};

template <class OperatorType, class VectorType, typename ValueType>
class BaseMultiGrid : public IterativeLinearSolver<OperatorType, VectorType, ValueType>
{
public:
    BaseMultiGrid()
        : levels_(-1)
        , current_level_(0)
        , scaling_(false)
        , iter_pre_smooth_(1)
        , iter_post_smooth_(1)
        , cycle_(Vcycle)
        , kcycle_full_(true)
        , op_level_(NULL)
        , restrict_op_level_(NULL)
        , prolong_op_level_(NULL)
        , d_level_(NULL)
        , r_level_(NULL)
        , t_level_(NULL)
        , s_level_(NULL)
        , q_level_(NULL)
        , solver_coarse_(NULL)
        , smoother_level_(NULL)
        , res_norm_(static_cast<ValueType>(0))
    {
    }
    virtual ~BaseMultiGrid()
    {
        this->Clear();
    }
    virtual void InitLevels(int levels)
    {
        assert(this->build_ == false && levels > 0);
        this->levels_ = levels;
    }
    virtual void SetPreconditioner(Solver<OperatorType, VectorType, ValueType>&)
    {
        LOG_INFO("BaseMultiGrid::SetPreconditioner() Perhaps you want to set the smoothers on all levels? use "
                 "SetSmootherLevel() instead of SetPreconditioner!");
        FATAL_ERROR(__FILE__, __LINE__);
    }
    virtual void SetSmoother(IterativeLinearSolver<OperatorType, VectorType, ValueType>** smoother)
    {
        assert(smoother != NULL);
        this->smoother_level_ = smoother;
    }
    virtual void SetSmootherPreIter(int iter)
    {
        this->iter_pre_smooth_ = iter;
    }
    virtual void SetSmootherPostIter(int iter)
    {
        this->iter_post_smooth_ = iter;
    }
    virtual void SetSolver(Solver<OperatorType, VectorType, ValueType>& solver)
    {
        this->solver_coarse_ = &solver;
    }
    virtual void SetScaling(bool scaling)
    {
        if(this->build_ == false) // needs extra storage: before Build only (base_multigrid.cpp:144-158)
            this->scaling_ = scaling;
    }
    virtual void SetHostLevels(int)
    {
        LOG_INFO("BaseMultiGrid::SetHostLevels(): this backend keeps every level on the accelerator");
    }
    virtual void SetCycle(unsigned int cycle)
    {
        this->cycle_ = cycle;
    }
    virtual void SetKcycleFull(bool kcycle_full)
    {
        this->kcycle_full_ = kcycle_full;
    }
    virtual void Print(void) const
    {
        LOG_INFO("MultiGrid solver");
    }
    virtual void Build(void)
    {
        if(this->build_)
            this->Clear();
        for(int i = 0; i < this->levels_ - 1; ++i)
            assert(this->op_level_[i] != NULL && this->smoother_level_[i] != NULL && this->restrict_op_level_[i] != NULL
                   && this->prolong_op_level_[i] != NULL);
        assert(this->op_ != NULL && this->solver_coarse_ != NULL && this->levels_ > 0);
        this->Initialize();
        this->build_ = true;
    }
    virtual void Clear(void)
    {
        if(this->build_)
        {
            this->Finalize();
            this->levels_ = -1;
            this->build_  = false;
        }
    }

protected:
    // base_multigrid.cpp:219-311
    virtual void Initialize(void)
    {
        assert(this->build_ == false && this->smoother_level_ != NULL);
        this->smoother_level_[0]->SetOperator(*this->op_);
        this->smoother_level_[0]->Build();
        this->smoother_level_[0]->FlagSmoother();
        for(int i = 1; i < this->levels_ - 1; ++i)
        {
            this->smoother_level_[i]->SetOperator(*this->op_level_[i - 1]);
            this->smoother_level_[i]->Build();
            this->smoother_level_[i]->FlagSmoother();
        }
        this->solver_coarse_->SetOperator(*this->op_level_[this->levels_ - 2]);
        this->solver_coarse_->Build();
        this->d_level_ = new VectorType*[this->levels_];
        this->r_level_ = new VectorType*[this->levels_];
        this->t_level_ = new VectorType*[this->levels_];
        this->d_level_[0] = NULL;
        if(this->scaling_)
        {
            this->s_level_ = new VectorType*[this->levels_];
            for(int i = 0; i < this->levels_; ++i)
                this->s_level_[i] = this->new_vec_(i, "temporary");
        }
        if(this->cycle_ == Kcycle)
        {
            this->q_level_ = new VectorType*[this->levels_ > 2 ? this->levels_ - 2 : 1];
            for(int i = 0; i < this->levels_ - 2; ++i)
                this->q_level_[i] = this->new_vec_(i + 1, "q");
        }
        for(int i = 1; i < this->levels_; ++i)
        {
            this->d_level_[i] = this->new_vec_(i, "defect correction");
            this->r_level_[i] = this->new_vec_(i, "residual");
            this->t_level_[i] = this->new_vec_(i, "temporary");
        }
        this->r_level_[0] = this->new_vec_(0, "residual");
        this->t_level_[0] = this->new_vec_(0, "temporary");
    }
    // base_multigrid.cpp:360-425
    virtual void Finalize(void)
    {
        for(int i = 0; i < this->levels_; ++i)
        {
            if(i > 0 && this->d_level_)
                delete this->d_level_[i];
            if(this->r_level_)
                delete this->r_level_[i];
            if(this->t_level_)
                delete this->t_level_[i];
            if(this->s_level_)
                delete this->s_level_[i];
        }
        if(this->q_level_)
            for(int i = 0; i < this->levels_ - 2; ++i)
                delete this->q_level_[i];
        delete[] this->d_level_;
        delete[] this->r_level_;
        delete[] this->t_level_;
        delete[] this->s_level_;
        delete[] this->q_level_;
        this->d_level_ = this->r_level_ = this->t_level_ = this->s_level_ = this->q_level_ = NULL;
        for(int i = 0; i < this->levels_ - 1; ++i)
            this->smoother_level_[i]->Clear();
        this->solver_coarse_->Clear();
        this->iter_ctrl_.Clear();
    }

public:
    // base_multigrid.cpp:605-699
    virtual void Solve(const VectorType& rhs, VectorType* x)
    {
        assert(this->levels_ > 1 && x != NULL && x != &rhs && this->op_ != NULL && this->build_ == true);
        assert(this->precond_ == NULL && this->solver_coarse_ != NULL);
        if(this->verb_ > 0)
        {
            this->PrintStart_();
            this->iter_ctrl_.PrintInit();
        }
        if(this->is_precond_ == false)
        {
            this->op_->Apply(*x, this->r_level_[0]);
            this->r_level_[0]->ScaleAdd(static_cast<ValueType>(-1), rhs);
            this->res_norm_ = std::abs(this->Norm_(*this->r_level_[0]));
            if(this->iter_ctrl_.InitResidual(this->res_norm_) == false)
                return;
        }
        else
            this->iter_ctrl_.InitResidual(1.0);
        this->Vcycle_(rhs, x);
        if(this->is_precond_ == false)
            while(!this->iter_ctrl_.CheckResidual(this->res_norm_, this->index_))
                this->Vcycle_(rhs, x);
        if(this->verb_ > 0)
        {
            this->iter_ctrl_.PrintStatus();
            this->PrintEnd_();
        }
    }

protected:
    virtual void PrintStart_(void) const
    {
        assert(this->levels_ > 0);
        LOG_INFO("MultiGrid solver starts");
        LOG_INFO("MultiGrid Number of levels " << this->levels_);
    }
    virtual void PrintEnd_(void) const
    {
        LOG_INFO("MultiGrid ends");
    }
    virtual void SolveNonPrecond_(const VectorType&, VectorType*)
    {
        LOG_INFO("BaseMultiGrid:SolveNonPrecond_() this function is disabled");
        FATAL_ERROR(__FILE__, __LINE__);
    }
    virtual void SolvePrecond_(const VectorType&, VectorType*)
    {
        LOG_INFO("BaseMultiGrid:SolvePrecond_() this function is disabled");
        FATAL_ERROR(__FILE__, __LINE__);
    }
    virtual void Restrict_(const VectorType& fine, VectorType* coarse)
    {
        this->restrict_op_level_[this->current_level_]->Apply(fine, coarse);
    }
    virtual void Prolong_(const VectorType& coarse, VectorType* fine)
    {
        this->prolong_op_level_[this->current_level_]->Apply(coarse, fine);
    }
    // base_multigrid.cpp:720-916
    void Vcycle_(const VectorType& rhs, VectorType* x)
    {
        if(this->current_level_ == this->levels_ - 1)
        {
            this->solver_coarse_->SolveZeroSol(rhs, x);
            return;
        }
        IterativeLinearSolver<OperatorType, VectorType, ValueType>* smoother = this->smoother_level_[this->current_level_];
        const OperatorType* op = (this->current_level_ == 0) ? this->op_ : this->op_level_[this->current_level_ - 1];
        VectorType*         r  = this->r_level_[this->current_level_];
        VectorType*         rc = this->t_level_[this->current_level_ + 1];
        VectorType*         rf = this->t_level_[this->current_level_];
        VectorType*         xc = this->d_level_[this->current_level_ + 1];
        VectorType*         s  = (this->scaling_) ? this->s_level_[this->current_level_] : NULL;
        ValueType           factor, divisor;
        smoother->InitMaxIter(this->iter_pre_smooth_);
        if(this->is_precond_ || this->current_level_ != 0)
            smoother->SolveZeroSol(rhs, x);
        else
            smoother->Solve(rhs, x);
        if(this->scaling_ == true)
            if(this->current_level_ > 0 && this->current_level_ < this->levels_ - 2 && this->iter_pre_smooth_ > 0)
            {
                s->PointWiseMult(rhs, *x);
                factor = s->Reduce();
                op->Apply(*x, s);
                s->PointWiseMult(*x);
                divisor = s->Reduce();
                if(divisor == static_cast<ValueType>(0))
                    factor = static_cast<ValueType>(1);
                else
                    factor /= divisor;
                x->Scale(factor);
            }
        op->Apply(*x, r);
        r->ScaleAdd(static_cast<ValueType>(-1), rhs);
        if(this->scaling_ && this->current_level_ == 0)
            s->CopyFrom(*r);
        this->Restrict_(*r, rc);
        ++this->current_level_;
        switch(this->cycle_)
        {
        case Vcycle: this->Vcycle_(*rc, xc); break;
        case Wcycle: this->Wcycle_(*rc, xc); break;
        case Kcycle: this->Kcycle_(*rc, xc); break;
        case Fcycle: this->Fcycle_(*rc, xc); break;
        default: FATAL_ERROR(__FILE__, __LINE__); break;
        }
        --this->current_level_;
        this->Prolong_(*xc, r);
        if(this->scaling_ == true && this->current_level_ < this->levels_ - 2)
        {
            if(this->current_level_ == 0)
                s->PointWiseMult(*r);
            else
                s->PointWiseMult(*r, *rf);
            factor = s->Reduce();
            op->Apply(*r, s);
            s->PointWiseMult(*r);
            divisor = s->Reduce();
            if(divisor == static_cast<ValueType>(0))
                factor = static_cast<ValueType>(1);
            else
                factor /= divisor;
            x->AddScale(*r, factor);
        }
        else
            x->AddScale(*r, static_cast<ValueType>(1));
        smoother->InitMaxIter(this->iter_post_smooth_);
        smoother->Solve(rhs, x);
        if(this->current_level_ == 0 && this->is_precond_ == false)
        {
            op->Apply(*x, r);
            r->ScaleAdd(static_cast<ValueType>(-1), rhs);
            this->res_norm_ = std::abs(this->Norm_(*r));
        }
    }
    void Wcycle_(const VectorType& rhs, VectorType* x)
    {
        for(int i = 0; i < 2; ++i) // gamma = 2 hardcoded (base_multigrid.cpp:919-927)
            this->Vcycle_(rhs, x);
    }
    void Fcycle_(const VectorType&, VectorType*)
    {
        LOG_INFO("BaseMultiGrid:Fcycle_() not implemented yet"); // nor in the reference (:930-935)
        FATAL_ERROR(__FILE__, __LINE__);
    }
    // base_multigrid.cpp:938-1011: two steps of CG around the cycle on the coarse levels
    void Kcycle_(const VectorType& rhs, VectorType* x)
    {
        if(this->current_level_ != 1 && this->kcycle_full_ == false)
            this->Vcycle_(rhs, x);
        else if(this->current_level_ < this->levels_ - 1)
        {
            VectorType*         q  = this->q_level_[this->current_level_ - 1];
            VectorType*         r  = this->t_level_[this->current_level_];
            const OperatorType* op = this->op_level_[this->current_level_ - 1];
            ValueType           rho, rho_old, alpha;
            this->Vcycle_(rhs, x);
            if(r != &rhs)
                r->CopyFrom(rhs);
            rho = r->DotNonConj(*x);
            op->Apply(*x, q);
            alpha = rho / x->DotNonConj(*q);
            r->AddScale(*q, -alpha);
            this->Vcycle_(*r, q);
            rho_old = rho;
            rho     = r->DotNonConj(*q);
            r->CopyFrom(*x);
            r->ScaleAdd(rho / rho_old, *q);
            op->Apply(*r, q);
            x->Scale(alpha);
            alpha = rho / r->DotNonConj(*q);
            x->AddScale(*r, alpha);
        }
        else
            this->solver_coarse_->SolveZeroSol(rhs, x);
    }
    VectorType* new_vec_(int level, const char* name)
    {
        const OperatorType* op = (level == 0) ? this->op_ : this->op_level_[level - 1];
        VectorType*         v  = new VectorType;
        v->CloneBackend(*op);
        v->Allocate(name, op->GetM());
        return v;
    }

    int          levels_;
    int          current_level_;
    bool         scaling_;
    int          iter_pre_smooth_;
    int          iter_post_smooth_;
    unsigned int cycle_;
    bool         kcycle_full_;
    OperatorType** op_level_; // [levels-1]: operators of levels 1 .. levels-1 (level 0 is op_)
    OperatorType** restrict_op_level_;
    OperatorType** prolong_op_level_;
    VectorType**   d_level_;
    VectorType**   r_level_;
    VectorType**   t_level_;
    VectorType**   s_level_;
    VectorType**   q_level_;
    Solver<OperatorType, VectorType, ValueType>*                 solver_coarse_;
    IterativeLinearSolver<OperatorType, VectorType, ValueType>** smoother_level_;
    ValueType                                                    res_norm_;
};

// MultiGrid (src/solvers/multigrid/multigrid.cpp): the hierarchy is handed in by the user; scaling on by default
template <class OperatorType, class VectorType, typename ValueType>
class MultiGrid : public BaseMultiGrid<OperatorType, VectorType, ValueType>
{
public:
    MultiGrid()
    {
        this->scaling_ = true;
    }
    virtual ~MultiGrid()
    {
        this->Clear();
        delete[] this->restrict_op_level_;
        delete[] this->prolong_op_level_;
    }
    virtual void SetRestrictOperator(OperatorType** op)
    {
        assert(this->build_ == false && op != NULL && this->levels_ > 0);
        delete[] this->restrict_op_level_;
        this->restrict_op_level_ = new OperatorType*[this->levels_];
        for(int i = 0; i < this->levels_ - 1; ++i)
            this->restrict_op_level_[i] = op[i];
    }
    virtual void SetProlongOperator(OperatorType** op)
    {
        assert(this->build_ == false && op != NULL && this->levels_ > 0);
        delete[] this->prolong_op_level_;
        this->prolong_op_level_ = new OperatorType*[this->levels_];
        for(int i = 0; i < this->levels_ - 1; ++i)
            this->prolong_op_level_[i] = op[i];
    }
    virtual void SetOperatorHierarchy(OperatorType** op)
    {
        assert(this->build_ == false && op != NULL);
        this->op_level_ = op;
    }
};

// ============================================================================ AMG
// BaseAMG (src/solvers/multigrid/base_amg.cpp): builds the hierarchy level by level through Aggregate_ until the
// coarse operator has at most coarse_size_ rows; default smoothers FixedPoint(2/3) + Jacobi, default coarse solver
// CG(0, 1e-6, 1e8, 1000).
typedef enum _coarsening_strategy
{
    Greedy = 0,
    PMIS   = 1
} CoarseningStrategy;

template <class OperatorType, class VectorType, typename ValueType>
class BaseAMG : public BaseMultiGrid<OperatorType, VectorType, ValueType>
{
public:
    BaseAMG()
        : coarse_size_(300)
        , set_sm_(false)
        , set_s_(false)
        , hierarchy_(false)
        , op_format_(CSR)
        , sm_default_(NULL)
    {
    }
    virtual ~BaseAMG()
    {
        this->Clear();
    }
    virtual void SetCoarsestLevel(int coarse_size)
    {
        this->coarse_size_ = coarse_size;
    }
    virtual void SetManualSmoothers(bool sm_manual)
    {
        this->set_sm_ = sm_manual;
    }
    virtual void SetManualSolver(bool s_manual)
    {
        this->set_s_ = s_manual;
    }
    virtual void SetOperatorFormat(unsigned int op_format, int op_blockdim = 1)
    {
        (void)op_blockdim;
        this->op_format_ = op_format;
    }
    virtual int GetNumLevels(void)
    {
        return this->levels_;
    }
    // base_amg.cpp:119-170
    virtual void Build(void)
    {
        if(this->build_)
            this->Clear();
        this->BuildHierarchy();
        if(this->set_sm_ == false)
            this->BuildSmoothers();
        if(this->set_s_ == false)
        {
            CG<OperatorType, VectorType, ValueType>* cgs = new CG<OperatorType, VectorType, ValueType>;
            cgs->Init(0.0, 1e-6, 1e+8, 1000);
            cgs->Verbose(0);
            this->solver_coarse_ = cgs;
        }
        this->Initialize();
        if(this->op_format_ != CSR)
            for(int i = 0; i < this->levels_ - 1; ++i)
                this->op_level_[i]->ConvertTo(this->op_format_);
        this->build_ = true;
    }
    // base_amg.cpp:173-310
    virtual void BuildHierarchy(void)
    {
        if(this->hierarchy_)
            return;
        this->hierarchy_ = true;
        if(this->op_->GetM() <= static_cast<int64_t>(this->coarse_size_))
        {
            LOG_INFO("Problem size too small for AMG, use Krylov solver instead");
            FATAL_ERROR(__FILE__, __LINE__);
        }
        std::vector<OperatorType*> ops, res, pro;
        this->levels_ = 1;
        const OperatorType* prev = this->op_;
        while(true)
        {
            OperatorType* c = new OperatorType;
            OperatorType* r = new OperatorType;
            OperatorType* p = new OperatorType;
            c->CloneBackend(*this->op_);
            r->CloneBackend(*this->op_);
            p->CloneBackend(*this->op_);
            const bool ok = this->Aggregate_(*prev, p, r, c);
            if(!ok)
            {
                delete c;
                delete r;
                delete p;
                if(ops.empty())
                {
                    LOG_INFO("Could not build initial AMG level");
                    FATAL_ERROR(__FILE__, __LINE__);
                }
                break;
            }
            ops.push_back(c);
            res.push_back(r);
            pro.push_back(p);
            ++this->levels_;
            prev = c;
            if(!(c->GetM() > static_cast<int64_t>(this->coarse_size_)))
                break;
        }
        this->op_level_          = new OperatorType*[this->levels_ - 1];
        this->restrict_op_level_ = new OperatorType*[this->levels_ - 1];
        this->prolong_op_level_  = new OperatorType*[this->levels_ - 1];
        for(int i = 0; i < this->levels_ - 1; ++i)
        {
            this->op_level_[i]          = ops[i];
            this->restrict_op_level_[i] = res[i];
            this->prolong_op_level_[i]  = pro[i];
        }
    }
    // base_amg.cpp:313-338
    virtual void BuildSmoothers(void)
    {
        this->smoother_level_ = new IterativeLinearSolver<OperatorType, VectorType, ValueType>*[this->levels_ - 1];
        this->sm_default_     = new Solver<OperatorType, VectorType, ValueType>*[this->levels_ - 1];
        for(int i = 0; i < this->levels_ - 1; ++i)
        {
            FixedPoint<OperatorType, VectorType, ValueType>* sm  = new FixedPoint<OperatorType, VectorType, ValueType>;
            Jacobi<OperatorType, VectorType, ValueType>*     jac = new Jacobi<OperatorType, VectorType, ValueType>;
            sm->SetRelaxation(static_cast<ValueType>(2.f / 3.f));
            sm->SetPreconditioner(*jac);
            sm->Verbose(0);
            this->smoother_level_[i] = sm;
            this->sm_default_[i]     = jac;
        }
    }
    // base_amg.cpp:341-395
    virtual void Clear(void)
    {
        if(this->build_)
        {
            this->Finalize();
            for(int i = 0; i < this->levels_ - 1; ++i)
            {
                delete this->op_level_[i];
                delete this->restrict_op_level_[i];
                delete this->prolong_op_level_[i];
            }
            delete[] this->op_level_;
            delete[] this->restrict_op_level_;
            delete[] this->prolong_op_level_;
            this->op_level_ = this->restrict_op_level_ = this->prolong_op_level_ = NULL;
            if(this->set_sm_ == false)
            {
                for(int i = 0; i < this->levels_ - 1; ++i)
                {
                    delete this->smoother_level_[i];
                    delete this->sm_default_[i];
                }
                delete[] this->smoother_level_;
                delete[] this->sm_default_;
                this->smoother_level_ = NULL;
                this->sm_default_     = NULL;
            }
            if(this->set_s_ == false)
            {
                delete this->solver_coarse_;
                this->solver_coarse_ = NULL;
            }
            this->levels_    = -1;
            this->build_     = false;
            this->hierarchy_ = false;
        }
    }
    virtual void SetRestrictOperator(OperatorType**)
    {
        LOG_INFO("BaseAMG::SetRestrictOperator() Perhaps you want to use the MultiGrid class to set external "
                 "restriction operators");
        FATAL_ERROR(__FILE__, __LINE__);
    }
    virtual void SetProlongOperator(OperatorType**)
    {
        LOG_INFO("BaseAMG::SetProlongOperator() Perhaps you want to use the MultiGrid class to set external "
                 "prolongation operators");
        FATAL_ERROR(__FILE__, __LINE__);
    }
    virtual void SetOperatorHierarchy(OperatorType**)
    {
        LOG_INFO("BaseAMG::SetOperatorHierarchy() Perhaps you want to use the MultiGrid class to set external operators");
        FATAL_ERROR(__FILE__, __LINE__);
    }

protected:
    virtual bool Aggregate_(const OperatorType& op, OperatorType* pro, OperatorType* res, OperatorType* coarse) = 0;

    int          coarse_size_;
    bool         set_sm_;
    bool         set_s_;
    bool         hierarchy_;
    unsigned int op_format_;
    Solver<OperatorType, VectorType, ValueType>** sm_default_;
};

// UAAMG (src/solvers/multigrid/unsmoothed_amg.cpp): unsmoothed aggregation; both coarsening strategies run on the
// device (Greedy: the sequential sweep of the reference as a sync-free sweep with the same aggregates).
template <class OperatorType, class VectorType, typename ValueType>
class UAAMG : public BaseAMG<OperatorType, VectorType, ValueType>
{
public:
    UAAMG()
        : eps_(static_cast<ValueType>(0.01f))
        , over_interp_(static_cast<ValueType>(1.5f))
        , strat_(Greedy)
    {
    }
    virtual ~UAAMG()
    {
        this->Clear();
    }
    virtual void Print(void) const
    {
        LOG_INFO("UAAMG solver");
        LOG_INFO("UAAMG number of levels " << this->levels_);
        LOG_INFO("UAAMG using unsmoothed aggregation");
    }
    virtual void SetOverInterp(ValueType overInterp)
    {
        this->over_interp_ = overInterp;
    }
    virtual void SetCouplingStrength(ValueType eps)
    {
        this->eps_ = eps;
    }
    virtual void SetCoarseningStrategy(CoarseningStrategy strat)
    {
        this->strat_ = strat;
    }

protected:
    virtual void PrintStart_(void) const
    {
        LOG_INFO("UAAMG solver starts");
        LOG_INFO("UAAMG number of levels " << this->levels_);
    }
    virtual void PrintEnd_(void) const
    {
        LOG_INFO("UAAMG ends");
    }
    // unsmoothed_amg.cpp:204-263
    virtual bool Aggregate_(const OperatorType& op, OperatorType* pro, OperatorType* res, OperatorType* coarse)
    {
        assert(pro != NULL && res != NULL && coarse != NULL);
        LocalVector<int> connections, aggregates, aggregate_root_nodes;
        ValueType        eps = this->eps_;
        for(int i = 0; i < this->levels_ - 1; ++i)
            eps *= static_cast<ValueType>(0.5);
        if(this->strat_ == PMIS)
            op.AMGPMISAggregate(eps, &connections, &aggregates, &aggregate_root_nodes);
        else
            op.AMGGreedyAggregate(eps, &connections, &aggregates, &aggregate_root_nodes);
        op.AMGUnsmoothedAggregation(aggregates, aggregate_root_nodes, pro);
        connections.Clear();
        aggregates.Clear();
        aggregate_root_nodes.Clear();
        pro->Transpose(res);
        coarse->CloneBackend(op);
        coarse->TripleMatrixProduct(*res, op, *pro);
        if(this->over_interp_ > static_cast<ValueType>(1))
            coarse->Scale(static_cast<ValueType>(1) / this->over_interp_);
        return true;
    }

    ValueType          eps_;
    ValueType          over_interp_;
    CoarseningStrategy strat_;
};

typedef enum _lumping_strategy
{
    AddWeakConnections      = 0,
    SubtractWeakConnections = 1
} LumpingStrategy;

// SAAMG (src/solvers/multigrid/smoothed_amg.cpp): smoothed aggregation; aggregation on the device with PMIS
template <class OperatorType, class VectorType, typename ValueType>
class SAAMG : public BaseAMG<OperatorType, VectorType, ValueType>
{
public:
    SAAMG()
        : eps_(static_cast<ValueType>(0.01f))
        , relax_(static_cast<ValueType>(2.f / 3.f))
        , strat_(Greedy)
        , lumping_strat_(AddWeakConnections)
    {
    }
    virtual ~SAAMG()
    {
        this->Clear();
    }
    virtual void Print(void) const
    {
        LOG_INFO("SAAMG solver");
        LOG_INFO("SAAMG number of levels " << this->levels_);
        LOG_INFO(((this->strat_ == PMIS) ? "SAAMG using PMIS smoothed aggregation" : "SAAMG using greedy smoothed aggregation"));
    }
    virtual void SetCouplingStrength(ValueType eps)
    {
        this->eps_ = eps;
    }
    virtual void SetInterpRelax(ValueType relax)
    {
        this->relax_ = relax;
    }
    virtual void SetCoarseningStrategy(CoarseningStrategy strat)
    {
        this->strat_ = strat;
    }
    virtual void SetLumpingStrategy(LumpingStrategy lumping_strat)
    {
        this->lumping_strat_ = lumping_strat;
    }

protected:
    virtual void PrintStart_(void) const
    {
        LOG_INFO("SAAMG solver starts");
        LOG_INFO("SAAMG number of levels " << this->levels_);
    }
    virtual void PrintEnd_(void) const
    {
        LOG_INFO("SAAMG ends");
    }
    // smoothed_amg.cpp:244-316
    virtual bool Aggregate_(const OperatorType& op, OperatorType* pro, OperatorType* res, OperatorType* coarse)
    {
        assert(pro != NULL && res != NULL && coarse != NULL);
        LocalVector<int> connections, aggregates, aggregate_root_nodes;
        ValueType        eps = this->eps_;
        for(int i = 0; i < this->levels_ - 1; ++i)
            eps *= static_cast<ValueType>(0.5);
        if(this->strat_ == PMIS)
            op.AMGPMISAggregate(eps, &connections, &aggregates, &aggregate_root_nodes);
        else
            op.AMGGreedyAggregate(eps, &connections, &aggregates, &aggregate_root_nodes);
        op.AMGSmoothedAggregation(this->relax_, connections, aggregates, aggregate_root_nodes, pro,
                                  this->lumping_strat_ == AddWeakConnections ? 0 : 1);
        connections.Clear();
        aggregates.Clear();
        aggregate_root_nodes.Clear();
        if(pro->GetN() == 0) // R would have no rows: the level is reverted by the caller
            return false;
        pro->Transpose(res);
        coarse->CloneBackend(op);
        coarse->TripleMatrixProduct(*res, op, *pro);
        return true;
    }

    ValueType          eps_;
    ValueType          relax_;
    CoarseningStrategy strat_;
    LumpingStrategy    lumping_strat_;
};

} // namespace rocalution
