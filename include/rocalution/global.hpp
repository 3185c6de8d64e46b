// rocalution/global.hpp -- ParallelManager / GlobalVector / GlobalMatrix / BlockJacobi (subset):
// row-block domain decomposition, one process per GPU.
//   src/base/parallel_manager.hpp:60-148, parallel_manager.cpp:726-787   ParallelManager
//   src/base/global_vector.cpp:139-680                                   GlobalVector
//   src/base/global_matrix.cpp:913-1009                                  GlobalMatrix::ConvertTo/Apply
//   src/solvers/preconditioners/preconditioner_blockjacobi.cpp:80-141    BlockJacobi
// Same object model as the reference (interior matrix + ghost matrix indexed into a compact receive
// buffer + boundary index list), but the halo never touches the host: pack kernel -> RCCL
// send/recv on the ghost stream over xGMI, overlapped with the interior SpMV; scalars are summed by
// one RCCL all-reduce on the device (ramd_comm_* in rocalution_amd.h).  The communicator handle
// passed to SetMPICommunicator() is a ramd_comm_t (the reference passes an MPI_Comm*).
#pragma once

#include "solvers.hpp"

namespace rocalution
{

class ParallelManager
{
public:
    ParallelManager()
        : comm_(NULL)
        , rank_(0)
        , num_procs_(1)
        , global_nrow_(0)
        , global_ncol_(0)
        , local_nrow_(0)
        , local_ncol_(0)
    {
    }
    void SetMPICommunicator(const void* comm)
    {
        this->comm_ = (ramd_comm_t) const_cast<void*>(comm);
        ramd_comm_rank(this->comm_, &this->rank_);
        ramd_comm_size(this->comm_, &this->num_procs_);
    }
    void Clear(void)
    {
        this->boundary_index_.clear();
        this->recvs_.clear();
        this->sends_.clear();
        this->recv_offset_.clear();
        this->send_offset_.clear();
    }
    ramd_comm_t GetComm(void) const
    {
        return this->comm_;
    }
    int GetRank(void) const
    {
        return this->rank_;
    }
    int GetNumProcs(void) const
    {
        return this->num_procs_;
    }
    int64_t GetGlobalNrow(void) const
    {
        return this->global_nrow_;
    }
    int64_t GetGlobalNcol(void) const
    {
        return this->global_ncol_;
    }
    int64_t GetLocalNrow(void) const
    {
        return this->local_nrow_;
    }
    int64_t GetLocalNcol(void) const
    {
        return this->local_ncol_;
    }
    int GetNumReceivers(void) const
    {
        return this->recv_offset_.empty() ? 0 : (int)this->recv_offset_.back();
    }
    int GetNumSenders(void) const
    {
        return this->send_offset_.empty() ? 0 : (int)this->send_offset_.back();
    }
    void SetGlobalNrow(int64_t nrow)
    {
        this->global_nrow_ = nrow;
    }
    void SetGlobalNcol(int64_t ncol)
    {
        this->global_ncol_ = ncol;
    }
    void SetLocalNrow(int64_t nrow)
    {
        this->local_nrow_ = nrow;
    }
    void SetLocalNcol(int64_t ncol)
    {
        this->local_ncol_ = ncol;
    }
    // local row indices whose values are sent, concatenated per receiving neighbour
    void SetBoundaryIndex(int size, const int* index)
    {
        this->boundary_index_.assign(index, index + size);
    }
    const int* GetBoundaryIndex(void) const
    {
        return this->boundary_index_.data();
    }
    int GetBoundarySize(void) const
    {
        return (int)this->boundary_index_.size();
    }
    void SetReceivers(int nrecv, const int* recvs, const int* recv_offset)
    {
        this->recvs_.assign(recvs, recvs + nrecv);
        this->recv_offset_.assign(recv_offset, recv_offset + nrecv + 1);
    }
    void SetSenders(int nsend, const int* sends, const int* send_offset)
    {
        this->sends_.assign(sends, sends + nsend);
        this->send_offset_.assign(send_offset, send_offset + nsend + 1);
    }
    bool Status(void) const
    {
        // every neighbour is both sender and receiver in this implementation (symmetric patterns)
        return this->global_nrow_ > 0 && this->local_nrow_ >= 0 && this->recvs_ == this->sends_
               && (int)this->boundary_index_.size() == this->GetNumSenders();
    }
    // File IO of the communication pattern (parallel_manager.cpp:441-743): a head file naming one "<file>.rank.<r>" file
    // per rank, each a list of "#KEY" sections.  Files written here are read by the reference and vice versa; every rank
    // writes / reads its own file, rank 0 also writes the head file.
    void WriteFileASCII(const std::string& filename) const
    {
        RAMD_EXPECT(this->Status());
        if(this->rank_ == 0)
        {
            std::ofstream head(filename.c_str());
            if(!head.is_open())
                {
                say("cannot open ParallelManager file [write]: " + filename);
                RAMD_DIE();
            }
            for(int r = 0; r < this->num_procs_; ++r)
                head << filename << ".rank." << r << "\n";
        }
        const std::string name = filename + ".rank." + std::to_string(this->rank_);
        std::ofstream     out(name.c_str());
        if(!out.is_open())
            {
                say("cannot open ParallelManager file [write]: " + name);
                RAMD_DIE();
            }
        static const char* const bar = "%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%";
        const int nrecv = (int)this->recvs_.size(), nsend = (int)this->sends_.size();
        out << bar << "\n%% ROCALUTION MPI ParallelManager output %%\n" << bar << "\n";
        out << "#RANK\n" << this->rank_ << "\n" << bar << "\n";
        out << "#GLOBAL_NROW\n" << this->global_nrow_ << "\n" << bar << "\n";
        out << "#GLOBAL_NCOL\n" << this->global_ncol_ << "\n" << bar << "\n";
        out << "#LOCAL_NROW\n" << this->local_nrow_ << "\n" << bar << "\n";
        out << "#LOCAL_NCOL\n" << this->local_ncol_ << "\n" << bar << "\n";
        out << "#BOUNDARY_SIZE\n" << this->boundary_index_.size() << "\n" << bar << "\n";
        out << "#NUMBER_OF_RECEIVERS\n" << nrecv << "\n" << bar << "\n";
        out << "#NUMBER_OF_SENDERS\n" << nsend << "\n" << bar << "\n";
        out << "#RECEIVERS_RANK\n";
        for(int i = 0; i < nrecv; ++i)
            out << this->recvs_[(size_t)i] << "\n";
        out << bar << "\n#SENDERS_RANK\n";
        for(int i = 0; i < nsend; ++i)
            out << this->sends_[(size_t)i] << "\n";
        out << bar << "\n#RECEIVERS_INDEX_OFFSET\n";
        for(size_t i = 0; i < this->recv_offset_.size(); ++i)
            out << this->recv_offset_[i] << "\n";
        out << bar << "\n#SENDERS_INDEX_OFFSET\n";
        for(size_t i = 0; i < this->send_offset_.size(); ++i)
            out << this->send_offset_[i] << "\n";
        out << bar << "\n#BOUNDARY_INDEX\n";
        for(size_t i = 0; i < this->boundary_index_.size(); ++i)
            out << this->boundary_index_[i] << "\n";
        if(!out.good())
            {
                say("write error on ParallelManager file: " + name);
                RAMD_DIE();
            }
    }
    // needs the communicator (rank) first; the sub-files are looked up in the directory of the head file
    void ReadFileASCII(const std::string& filename)
    {
        RAMD_EXPECT(this->comm_ != NULL);
        std::ifstream head(filename.c_str());
        if(!head.is_open())
            {
                say("cannot open ParallelManager file [read]: " + filename);
                RAMD_DIE();
            }
        std::string name;
        for(int r = 0; r <= this->rank_; ++r)
            if(!std::getline(head, name))
                {
                say("ParallelManager head file names fewer ranks than this communicator has: " + filename);
                RAMD_DIE();
            }
        std::string trimmed;
        for(size_t i = 0; i < name.size(); ++i)
            if(!std::isspace((unsigned char)name[i]))
                trimmed += name[i];
        // (the reference prepends the head file's directory to the name as it stands in the head file)
        const size_t cut = filename.find_last_of("\\/");
        name             = (cut == std::string::npos ? std::string() : filename.substr(0, cut + 1)) + trimmed;
        std::ifstream in(name.c_str());
        if(!in.is_open())
        {
            in.open(trimmed.c_str()); // a head file that already carries the path (written with a path, read from elsewhere)
            if(!in.is_open())
                {
                say("cannot open ParallelManager file [read]: " + name);
                RAMD_DIE();
            }
        }
        this->Clear();
        int     rank = -1, nrecv = -1, nsend = -1;
        int64_t bsize = -1;
        std::string line;
        auto ints = [&](std::vector<int>& v, int64_t count) {
            v.resize((size_t)(count > 0 ? count : 0));
            for(size_t i = 0; i < v.size(); ++i)
                in >> v[i];
        };
        auto longs = [&](std::vector<int64_t>& v, int64_t count) {
            v.resize((size_t)(count > 0 ? count : 0));
            for(size_t i = 0; i < v.size(); ++i)
                in >> v[i];
        };
        while(std::getline(in, line))
        {
            if(line.find("#RANK") != std::string::npos)
                in >> rank;
            else if(line.find("#GLOBAL_SIZE") != std::string::npos)
            {
                in >> this->global_nrow_;
                this->global_ncol_ = this->global_nrow_;
            }
            else if(line.find("#GLOBAL_NROW") != std::string::npos)
                in >> this->global_nrow_;
            else if(line.find("#GLOBAL_NCOL") != std::string::npos)
                in >> this->global_ncol_;
            else if(line.find("#LOCAL_SIZE") != std::string::npos)
            {
                in >> this->local_nrow_;
                this->local_ncol_ = this->local_nrow_;
            }
            else if(line.find("#LOCAL_NROW") != std::string::npos)
                in >> this->local_nrow_;
            else if(line.find("#LOCAL_NCOL") != std::string::npos)
                in >> this->local_ncol_;
            else if(line.find("#BOUNDARY_SIZE") != std::string::npos)
                in >> bsize;
            else if(line.find("#NUMBER_OF_RECEIVERS") != std::string::npos)
                in >> nrecv;
            else if(line.find("#NUMBER_OF_SENDERS") != std::string::npos)
                in >> nsend;
            else if(line.find("#RECEIVERS_RANK") != std::string::npos)
                ints(this->recvs_, nrecv);
            else if(line.find("#SENDERS_RANK") != std::string::npos)
                ints(this->sends_, nsend);
            else if(line.find("#RECEIVERS_INDEX_OFFSET") != std::string::npos)
                longs(this->recv_offset_, (int64_t)nrecv + 1);
            else if(line.find("#SENDERS_INDEX_OFFSET") != std::string::npos)
                longs(this->send_offset_, (int64_t)nsend + 1);
            else if(line.find("#BOUNDARY_INDEX") != std::string::npos)
                ints(this->boundary_index_, bsize);
            if(in.fail() && !in.eof())
                {
                say("malformed ParallelManager file: " + name);
                RAMD_DIE();
            }
        }
        if(rank != this->rank_)
        {
            say("ParallelManager file " + name + " belongs to another rank");
            RAMD_DIE();
        }
        // a pattern written by the reference may name a rank as sender only or as receiver only (unsymmetric matrices): the
        // exchange here pairs every send with a receive, so the two lists become their union with empty pieces
        if(nrecv >= 0 && nsend >= 0 && this->recvs_ != this->sends_)
            this->doUnitePeers();
        if(nrecv < 0 || nsend < 0 || bsize < 0 || !this->Status())
            {
                say("incomplete ParallelManager file: " + name);
                RAMD_DIE();
            }
    }
    // receivers and senders -> one ascending peer list; a peer missing on one side gets an empty piece there.  The order of
    // the non-empty pieces (receive buffer = column numbering of the ghost matrix, boundary index) does not change, which
    // needs both lists ascending as the reference writes them.
    void doUnitePeers(void)
    {
        if(!std::is_sorted(this->recvs_.begin(), this->recvs_.end()) || !std::is_sorted(this->sends_.begin(), this->sends_.end())
           || this->recv_offset_.size() != this->recvs_.size() + 1 || this->send_offset_.size() != this->sends_.size() + 1)
            return; // (Status() reports it)
        std::vector<int> all(this->recvs_);
        all.insert(all.end(), this->sends_.begin(), this->sends_.end());
        std::sort(all.begin(), all.end());
        all.erase(std::unique(all.begin(), all.end()), all.end());
        auto spread = [&](const std::vector<int>& have, const std::vector<int64_t>& off) {
            std::vector<int64_t> out(1, 0);
            size_t               k = 0;
            for(int p : all)
            {
                int64_t len = 0;
                if(k < have.size() && have[k] == p)
                {
                    len = off[k + 1] - off[k];
                    ++k;
                }
                out.push_back(out.back() + len);
            }
            return out;
        };
        this->recv_offset_ = spread(this->recvs_, this->recv_offset_);
        this->send_offset_ = spread(this->sends_, this->send_offset_);
        this->recvs_       = all;
        this->sends_       = all;
    }
    const std::vector<int>& peers(void) const
    {
        return this->sends_;
    }
    const std::vector<int64_t>& send_offset(void) const
    {
        return this->send_offset_;
    }
    const std::vector<int64_t>& recv_offset(void) const
    {
        return this->recv_offset_;
    }

private:
    ramd_comm_t          comm_;
    int                  rank_, num_procs_;
    int64_t              global_nrow_, global_ncol_, local_nrow_, local_ncol_;
    std::vector<int>     boundary_index_;
    std::vector<int>     recvs_, sends_;
    std::vector<int64_t> recv_offset_, send_offset_;
};

template <typename ValueType>
class GlobalMatrix;

template <typename ValueType>
class GlobalVector
{
public:
    GlobalVector()
        : pm_(NULL)
    {
    }
    explicit GlobalVector(const ParallelManager& pm)
        : pm_(&pm)
    {
    }
    void SetParallelManager(const ParallelManager& pm)
    {
        this->pm_ = &pm;
    }
    const ParallelManager* pm(void) const
    {
        return this->pm_;
    }
    bool is_accel_(void) const
    {
        return this->m_owned.is_accel_();
    }
    void MoveToAccelerator(void)
    {
        this->m_owned.MoveToAccelerator();
    }
    void MoveToHost(void)
    {
        this->m_owned.MoveToHost();
    }
    template <class Obj>
    void CloneBackend(const Obj& src) // also adopts the parallel manager (base_rocalution.cpp:109)
    {
        this->pm_ = src.pm();
        if(src.is_accel_())
            this->MoveToAccelerator();
        else
            this->MoveToHost();
    }
    // global_vector.cpp:139-166: size is GLOBAL, the interior gets the local share
    void Allocate(std::string name, int64_t size)
    {
        int64_t local = size;
        if(this->pm_ != NULL)
        {
            assert(this->pm_->GetGlobalNrow() == size || this->pm_->GetGlobalNcol() == size);
            local = (this->pm_->GetGlobalNrow() == size) ? this->pm_->GetLocalNrow()
                                                         : this->pm_->GetLocalNcol();
        }
        this->m_owned.Allocate("Interior of " + name, local);
    }
    void Clear(void)
    {
        this->m_owned.Clear();
    }
    void PlaceApartFrom(const GlobalVector<ValueType>& other) // (placement hint of LocalVector, for the rank's share)
    {
        this->m_owned.PlaceApartFrom(other.m_owned);
    }
    void PlaceByTrial(const std::function<void()>& run, int tries, double stop_ratio = 0.0,
                      const GlobalVector<ValueType>* apart_from = nullptr)
    {
        // `run` launches kernels on this rank's vectors only -- never a collective (ramd_vec_place_by_trial): every rank
        // places its share by itself, with as many candidates as its own memory allows
        this->m_owned.PlaceByTrial(run, tries, stop_ratio, apart_from ? &apart_from->m_owned : nullptr);
    }
    int64_t GetSize(void) const
    {
        return this->pm_ ? this->pm_->GetGlobalNrow() : this->m_owned.GetSize();
    }
    int64_t GetLocalSize(void) const
    {
        return this->m_owned.GetSize();
    }
    LocalVector<ValueType>& GetInterior(void)
    {
        return this->m_owned;
    }
    const LocalVector<ValueType>& GetInterior(void) const
    {
        return this->m_owned;
    }
    void Info(void) const
    {
        LOG_INFO("GlobalVector size=" << this->GetSize() << "; local=" << this->GetLocalSize());
    }
    void Zeros(void)
    {
        this->m_owned.Zeros();
    }
    void Ones(void)
    {
        this->m_owned.Ones();
    }
    void SetValues(ValueType val)
    {
        this->m_owned.SetValues(val);
    }
    // global_vector.cpp:317-330: every rank fills its interior part from the same seed
    void SetRandomUniform(unsigned long long seed, ValueType a = static_cast<ValueType>(-1),
                          ValueType b = static_cast<ValueType>(1))
    {
        this->m_owned.SetRandomUniform(seed, a, b);
    }
    void SetRandomNormal(unsigned long long seed, ValueType mean = static_cast<ValueType>(0),
                         ValueType var = static_cast<ValueType>(1))
    {
        this->m_owned.SetRandomNormal(seed, mean, var);
    }
    void CopyFrom(const GlobalVector<ValueType>& src)
    {
        this->m_owned.CopyFrom(src.m_owned);
    }
    void AddScale(const GlobalVector<ValueType>& x, ValueType alpha)
    {
        this->m_owned.AddScale(x.m_owned, alpha);
    }
    void ScaleAdd(ValueType alpha, const GlobalVector<ValueType>& x)
    {
        this->m_owned.ScaleAdd(alpha, x.m_owned);
    }
    void ScaleAdd2(ValueType alpha, const GlobalVector<ValueType>& x, ValueType beta,
                   const GlobalVector<ValueType>& y, ValueType gamma)
    {
        this->m_owned.ScaleAdd2(alpha, x.m_owned, beta, y.m_owned, gamma);
    }
    void ScaleAddScale(ValueType alpha, const GlobalVector<ValueType>& x, ValueType beta)
    {
        this->m_owned.ScaleAddScale(alpha, x.m_owned, beta);
    }
    void Scale(ValueType alpha)
    {
        this->m_owned.Scale(alpha);
    }
    void PointWiseMult(const GlobalVector<ValueType>& x)
    {
        this->m_owned.PointWiseMult(x.m_owned);
    }
    void PointWiseMult(const GlobalVector<ValueType>& x, const GlobalVector<ValueType>& y)
    {
        this->m_owned.PointWiseMult(x.m_owned, y.m_owned);
    }
    void CopyFromFloat(const GlobalVector<float>& src)
    {
        this->m_owned.CopyFromFloat(src.GetInterior());
    }
    void CopyFromDouble(const GlobalVector<double>& src)
    {
        this->m_owned.CopyFromDouble(src.GetInterior());
    }
    // global_vector.cpp:547-588: local reduction, then sum over ranks; Norm = sqrt(allreduce(dot))
    ValueType Dot(const GlobalVector<ValueType>& x) const
    {
        return (ValueType)this->reduce_(x.m_owned.handle(), false);
    }
    ValueType DotNonConj(const GlobalVector<ValueType>& x) const
    {
        return this->Dot(x);
    }
    ValueType Norm(void) const
    {
        return (ValueType)std::sqrt(this->reduce_(this->m_owned.handle(), false));
    }
    ValueType Asum(void) const
    {
        double     local = (double)this->m_owned.Asum();
        return (ValueType)this->sum_ranks_(local);
    }
    int64_t Amax(ValueType& value) const
    {
        LOG_INFO("GlobalVector::Amax() is not provided by this backend");
        FATAL_ERROR(__FILE__, __LINE__);
        value = 0;
        return -1;
    }

private:
    double reduce_(ramd_vec_t other, bool) const
    {
        const int        slot  = RAMD_NSCALARS - 2;
        const ramd_vec_t vs[1] = {this->m_owned.handle()};
        RAMD_CHECK(ramd_fused_multi_dot(vs, 1, other, slot));
        if(this->pm_ != NULL && this->pm_->GetNumProcs() > 1)
            RAMD_CHECK(ramd_comm_allreduce_scalars(this->pm_->GetComm(), slot, 1));
        double r = 0.0;
        RAMD_CHECK(ramd_scalars_fetch(&r, slot, 1));
        return r;
    }
    double sum_ranks_(double local) const
    {
        if(this->pm_ == NULL || this->pm_->GetNumProcs() == 1)
            return local;
        const int slot = RAMD_NSCALARS - 2;
        RAMD_CHECK(ramd_scalars_set(slot, local));
        RAMD_CHECK(ramd_comm_allreduce_scalars(this->pm_->GetComm(), slot, 1));
        double r = 0.0;
        RAMD_CHECK(ramd_scalars_fetch(&r, slot, 1));
        return r;
    }
    const ParallelManager* pm_;
    LocalVector<ValueType> m_owned;
    friend class GlobalMatrix<ValueType>;
};

template <typename ValueType>
class GlobalMatrix
{
public:
    GlobalMatrix()
        : pm_(NULL)
    {
    }
    explicit GlobalMatrix(const ParallelManager& pm)
        : pm_(&pm)
    {
    }
    ~GlobalMatrix()
    {
        this->doDropHaloPlan();
    }
    void SetParallelManager(const ParallelManager& pm)
    {
        this->pm_ = &pm;
    }
    const ParallelManager* pm(void) const
    {
        return this->pm_;
    }
    bool is_accel_(void) const
    {
        return this->m_interior.is_accel_();
    }
    int64_t GetM(void) const
    {
        return this->pm_ ? this->pm_->GetGlobalNrow() : this->m_interior.GetM();
    }
    int64_t GetN(void) const
    {
        return this->pm_ ? this->pm_->GetGlobalNcol() : this->m_interior.GetN();
    }
    int64_t GetLocalM(void) const
    {
        return this->m_interior.GetM();
    }
    int64_t GetLocalN(void) const
    {
        return this->m_interior.GetN();
    }
    int64_t GetLocalNnz(void) const
    {
        return this->m_interior.GetNnz();
    }
    int64_t GetGhostNnz(void) const
    {
        return this->m_ghost.GetNnz();
    }
    LocalMatrix<ValueType>& GetInterior(void)
    {
        return this->m_interior;
    }
    const LocalMatrix<ValueType>& GetInterior(void) const
    {
        return this->m_interior;
    }
    LocalMatrix<ValueType>& GetGhost(void)
    {
        return this->m_ghost;
    }
    const LocalMatrix<ValueType>& GetGhost(void) const
    {
        return this->m_ghost;
    }
    bool IsTransferOperator(void) const // a coupled prolongation or a reverse-form restriction of the AMG setup
    {
        return this->m_coupled || this->m_reverse;
    }
    void Info(void) const
    {
        LOG_INFO("GlobalMatrix rows=" << this->GetM() << "; cols=" << this->GetN()
                                      << "; local nnz=" << this->GetLocalNnz()
                                      << "; ghost nnz=" << this->GetGhostNnz());
    }
    void SetLocalDataPtrCSR(PtrType** row_offset, int** col, ValueType** val, std::string name,
                            int64_t nnz)
    {
        assert(this->pm_ != NULL);
        this->m_amg.reset(); // (a new operator: the merged block of an earlier aggregation describes the old one)
        this->m_interior.SetDataPtrCSR(row_offset, col, val, "Interior of " + name, nnz,
                                             this->pm_->GetLocalNrow(), this->pm_->GetLocalNcol());
    }
    void SetGhostDataPtrCSR(PtrType** row_offset, int** col, ValueType** val, std::string name,
                            int64_t nnz)
    {
        assert(this->pm_ != NULL);
        this->m_amg.reset();
        this->m_ghost.SetDataPtrCSR(row_offset, col, val, "Ghost of " + name, nnz,
                                          this->pm_->GetLocalNrow(), this->pm_->GetNumReceivers());
    }
    void MoveToAccelerator(void)
    {
        this->m_interior.MoveToAccelerator();
        this->m_ghost.MoveToAccelerator();
        if(this->m_reverse)
        {
            // a restriction in the reverse form (doTransposeCoupled) has no halo plan of the forward kind -- announcing one
            // would be a collective the other ranks' objects do not enter -- and its scatter operator moves with it
            this->m_scatter.MoveToAccelerator();
            return;
        }
        this->doInitHalo();
    }
    // global_matrix.cpp:913-921: interior in the requested format, ghost part always COO
    void ConvertTo(unsigned int matrix_format, int blockdim = 1)
    {
        this->m_interior.ConvertTo(matrix_format, blockdim);
        if(this->m_ghost.GetNnz() > 0)
            this->m_ghost.ConvertTo(COO);
    }
    // extension: only the ghost part to (row-grouped) COO -- a CSR ghost part is walked over ALL local rows
    // although only the boundary rows have entries.  Results are bit-identical (same per-row order).
    void CompactGhost(void)
    {
        if(this->m_ghost.GetNnz() > 0 && this->m_ghost.GetFormat() == CSR)
            this->m_ghost.ConvertTo(COO);
    }
    void ConvertToCSR(void)
    {
        this->ConvertTo(CSR);
    }
    void ConvertToELL(void)
    {
        this->ConvertTo(ELL);
    }
    void ConvertToHYB(void)
    {
        this->ConvertTo(HYB);
    }
    void ExtractInverseDiagonal(GlobalVector<ValueType>* vec_inv_diag) const
    {
        this->m_interior.ExtractInverseDiagonal(&vec_inv_diag->m_owned);
    }
    // extension: value-cast copy (interior + ghost, same parallel manager) for MixedPrecisionDC on
    // Global objects -- the reference instantiates MixedPrecisionDC for LocalMatrix only
    // (src/solvers/mixed_precision.cpp:463-468); this is the row-block generalisation of :201-229
    template <typename OtherType>
    void CastFrom(const GlobalMatrix<OtherType>& src)
    {
        // (operators only: the coupled prolongation / reverse-form restriction of the AMG setup carry their own exchange state)
        RAMD_EXPECT(!src.IsTransferOperator());
        this->pm_ = src.pm();
        this->m_interior.template CastFrom<OtherType>(src.GetInterior());
        const unsigned int gfmt = src.GetGhost().GetFormat();
        if(gfmt == CSR)
            this->m_ghost.template CastFrom<OtherType>(src.GetGhost());
        else // the value cast is defined on CSR (mixed_precision.cpp:201); keep the source's ghost format
        {
            LocalMatrix<OtherType> tmp;
            tmp.CloneFrom(src.GetGhost());
            tmp.ConvertTo(CSR);
            this->m_ghost.template CastFrom<OtherType>(tmp);
            this->m_ghost.ConvertTo(gfmt);
        }
        this->doInitHalo();
    }
    // extension: per-rank slab of the synthetic 3-D Poisson operator, built on the device
    void GeneratePoisson7Slab(int N, int64_t row_begin, int64_t row_end)
    {
        this->m_interior.MoveToAccelerator();
        this->m_ghost.MoveToAccelerator();
        RAMD_CHECK(ramd_mat_gen_poisson7_slab(this->m_interior.handle(), this->m_ghost.handle(),
                                              N, row_begin, row_end));
        this->doInitHalo();
    }

    // ... and of the reference's own 27-point operator (clients/include/common.hpp:926-1249 generates it per rank)
    void GenerateLaplace27Slab(int nx, int ny, int nz, int z_begin, int z_end)
    {
        this->m_interior.MoveToAccelerator();
        this->m_ghost.MoveToAccelerator();
        RAMD_CHECK(ramd_mat_gen_laplace27_slab(this->m_interior.handle(), this->m_ghost.handle(), nx, ny, nz, z_begin, z_end));
        this->doInitHalo();
    }

    // global_matrix.cpp:924-1009, device-resident: pack | halo over xGMI || interior SpMV | ghost +=
    void Apply(const GlobalVector<ValueType>& in, GlobalVector<ValueType>* out) const
    {
        if(this->m_reverse)
        {
            this->doApplyReverse(in, out);
            return;
        }
        const bool comm = this->pm_ != NULL && (this->m_halo_plan > 0 || !this->pm_->peers().empty());
        if(comm)
        {
            in.m_owned.GetIndexValues(this->m_halo_rows, &this->m_send);
            RAMD_CHECK(ramd_comm_halo_begin_plan(this->pm_->GetComm(), this->m_halo_plan, this->m_send.handle(),
                                            this->m_recv.handle(), (int)this->pm_->peers().size(),
                                            this->pm_->peers().data(), this->pm_->send_offset().data(),
                                            this->pm_->recv_offset().data()));
        }
        this->m_interior.Apply(in.m_owned, &out->m_owned);
        if(comm)
        {
            RAMD_CHECK(ramd_comm_halo_end(this->pm_->GetComm()));
            if(this->m_ghost.GetNnz() > 0)
                this->m_ghost.ApplyAdd(this->m_recv, static_cast<ValueType>(1), &out->m_owned);
        }
    }

    // Apply + the rank-local part of <in, out> into a device scalar slot, without a second pass over
    // the vectors: the interior SpMV carries the dot, the ghost ApplyAdd corrects it on the rows it touches
    void ApplyDot(const GlobalVector<ValueType>& in, GlobalVector<ValueType>* out, int slot) const
    {
        this->ApplyDotV(in, in, out, slot);
    }
    // ... the same with the dot taken against another vector: slot = local part of <w, out>
    void ApplyDotV(const GlobalVector<ValueType>& in, const GlobalVector<ValueType>& w,
                   GlobalVector<ValueType>* out, int slot) const
    {
        RAMD_EXPECT(!this->m_reverse); // (a restriction is applied, never the operator of a fused product + dot)
        const bool comm = this->pm_ != NULL && (this->m_halo_plan > 0 || !this->pm_->peers().empty());
        if(comm)
        {
            in.m_owned.GetIndexValues(this->m_halo_rows, &this->m_send);
            RAMD_CHECK(ramd_comm_halo_begin_plan(this->pm_->GetComm(), this->m_halo_plan, this->m_send.handle(),
                                            this->m_recv.handle(), (int)this->pm_->peers().size(),
                                            this->pm_->peers().data(), this->pm_->send_offset().data(),
                                            this->pm_->recv_offset().data()));
        }
        if(&w == &in)
            RAMD_CHECK(ramd_fused_apply_dot(this->m_interior.handle(), in.m_owned.handle(),
                                            out->m_owned.handle(), slot));
        else
            RAMD_CHECK(ramd_fused_apply_dotv(this->m_interior.handle(), in.m_owned.handle(),
                                             out->m_owned.handle(), w.m_owned.handle(),
                                             slot));
        if(comm)
        {
            RAMD_CHECK(ramd_comm_halo_end(this->pm_->GetComm()));
            if(this->m_ghost.GetNnz() > 0)
                RAMD_CHECK(ramd_fused_apply_add_dot(this->m_ghost.handle(), this->m_recv.handle(), 1.0,
                                                out->m_owned.handle(),
                                                w.m_owned.handle(), slot));
        }
    }

    // ---- aggregation AMG on the row-block decomposition (global_matrix.cpp:1038-1880 Transpose / TripleMatrixProduct,
    // :2607-3558 AMG*Aggregate / AMG*Aggregation).  Two forms:
    //  * coupled (PMIS, the reference's only strategy on more than one rank): the aggregates cross the rank boundaries
    //    exactly as the reference's do -- the aggregation runs on the block [interior | ghost] with every per-node array
    //    extended over the ghost nodes (ramd_mat_amg_pmis_aggregate_global) and numbers the aggregates as ONE rank would,
    //    so the hierarchy (aggregates, P, the Galerkin operators) does not depend on the number of ranks.  A coarse row
    //    lives on the rank that owns the root node of its aggregate.  P is interior + ghost with a halo plan over the
    //    coarse vector; R = P^T is NOT assembled: it is applied as P_int^T x + the reverse exchange of P_ghost^T x (the
    //    ghost columns' partial sums travel to their owners and are added there).  The Galerkin product is formed row
    //    block by row block -- every rank multiplies out P^T (A P) over ITS fine rows (device SpGEMMs) and ships the
    //    coarse rows it does not own to their owners, who add them up.
    //  * decoupled (Greedy on more than one rank, an extension: the reference refuses, global_matrix.cpp:2607-2645; or
    //    RAMD_GLOBAL_AMG=decoupled): the aggregates of a rank stay inside its row block, the prolongation and restriction
    //    operators are block-diagonal and only the Galerkin product exchanges data.
    // On one rank both are the LocalMatrix algorithm, kernel for kernel.
    template <class Obj>
    void CloneBackend(const Obj& src)
    {
        if(src.is_accel_())
        {
            this->m_interior.MoveToAccelerator();
            this->m_ghost.MoveToAccelerator();
        }
    }
    void Scale(ValueType alpha)
    {
        this->m_interior.Scale(alpha);
        if(this->m_ghost.GetNnz() > 0)
            this->m_ghost.Scale(alpha);
    }
    void AMGPMISAggregate(ValueType eps, LocalVector<int>* connections, LocalVector<int>* aggregates,
                          LocalVector<int>* aggregate_root_nodes) const
    {
        this->m_amg.reset();
        if(!this->doCoupled())
        {
            this->m_interior.AMGPMISAggregate(eps, connections, aggregates, aggregate_root_nodes);
            return;
        }
        std::shared_ptr<AmgBlock> w(new AmgBlock);
        this->doRowBlock(&w->block);
        w->offsets   = doRankOffsets(this->pm_, this->m_interior.GetM());
        w->first_row = w->offsets[(size_t)this->pm_->GetRank()];
        connections->MoveToAccelerator();
        aggregates->MoveToAccelerator();
        aggregate_root_nodes->MoveToAccelerator();
        w->numbers.MoveToAccelerator();
        RAMD_CHECK(ramd_mat_amg_pmis_aggregate_global(
            w->block.handle(), (double)eps, this->pm_->GetComm(), this->m_halo_plan, (int)this->pm_->peers().size(),
            this->pm_->peers().data(), this->pm_->send_offset().data(), this->pm_->recv_offset().data(),
            this->m_halo_rows.handle(), w->first_row, w->numbers.handle(), connections->handle(), aggregates->handle(),
            aggregate_root_nodes->handle(), &w->agg_first, &w->agg_mine, &w->agg_total));
        this->m_amg = w;
    }
    void AMGGreedyAggregate(ValueType eps, LocalVector<int>* connections, LocalVector<int>* aggregates,
                            LocalVector<int>* aggregate_root_nodes) const
    {
        this->m_amg.reset(); // (rank-local aggregates: a coupled block of an earlier PMIS aggregation must not be taken up)
        this->m_interior.AMGGreedyAggregate(eps, connections, aggregates, aggregate_root_nodes);
    }
    void AMGUnsmoothedAggregation(const LocalVector<int>& aggregates, const LocalVector<int>& aggregate_root_nodes,
                                  GlobalMatrix<ValueType>* prolong) const
    {
        assert(prolong != NULL && prolong != this);
        if(this->m_amg)
        {
            // (the aggregates of the coupled form cover the rank's rows AND its ghost nodes: anything else was not made by the
            //  AMGPMISAggregate call this block belongs to)
            RAMD_EXPECT(aggregates.GetSize() == this->m_amg->block.GetN());
            this->doProlongCoupled(0, static_cast<ValueType>(0), 0, aggregates, aggregates, aggregate_root_nodes, prolong);
            return;
        }
        this->m_interior.AMGUnsmoothedAggregation(aggregates, aggregate_root_nodes, &prolong->m_interior);
        prolong->doBlockDiagonal(this->pm_, this->GetM(), -1);
    }
    void AMGSmoothedAggregation(ValueType relax, const LocalVector<int>& connections, const LocalVector<int>& aggregates,
                                const LocalVector<int>& aggregate_root_nodes, GlobalMatrix<ValueType>* prolong,
                                int lumping_strat = 0) const
    {
        assert(prolong != NULL && prolong != this);
        if(this->m_amg)
        {
            RAMD_EXPECT(aggregates.GetSize() == this->m_amg->block.GetN());
            this->doProlongCoupled(1, relax, lumping_strat, connections, aggregates, aggregate_root_nodes, prolong);
            return;
        }
        // The smoothing step I - w D^-1 A_F uses the interior block: couplings across ranks do not widen the rows of P.  They
        // are treated the way the filtered matrix A_F treats weak connections -- lumped onto the diagonal -- so that the
        // rows of A_F next to a rank boundary keep their row sum and P still reproduces what the tentative prolongation
        // reproduces there (without it CG + SA-AMG needed twice the iterations on two ranks).
        if(this->m_ghost.GetNnz() > 0)
        {
            const int64_t          n = this->m_interior.GetM();
            LocalVector<ValueType> ones, lump;
            ones.MoveToAccelerator();
            lump.MoveToAccelerator();
            ones.Allocate("ones", this->m_ghost.GetN());
            ones.Ones();
            lump.Allocate("ghost row sums", n);
            this->m_ghost.Apply(ones, &lump);
            std::vector<ValueType> h((size_t)n);
            lump.CopyToHostData(h.data());
            std::vector<PtrType> rp((size_t)n + 1);
            std::vector<int>     ci((size_t)n);
            for(int64_t i = 0; i <= n; ++i)
                rp[(size_t)i] = (PtrType)i;
            for(int64_t i = 0; i < n; ++i)
            {
                ci[(size_t)i] = (int)i;
                if(lumping_strat == 1) // (SubtractWeakConnections)
                    h[(size_t)i] = -h[(size_t)i];
            }
            LocalMatrix<ValueType> D, Al;
            D.MoveToAccelerator();
            D.AllocateCSR("lumped ghost couplings", n, n, n);
            D.CopyFromCSR(rp.data(), ci.data(), h.data());
            Al.CloneFrom(this->m_interior);
            Al.MatrixAdd(D, static_cast<ValueType>(1), static_cast<ValueType>(1), false);
            Al.AMGSmoothedAggregation(relax, connections, aggregates, aggregate_root_nodes, &prolong->m_interior, lumping_strat);
        }
        else
            this->m_interior.AMGSmoothedAggregation(relax, connections, aggregates, aggregate_root_nodes, &prolong->m_interior,
                                                    lumping_strat);
        prolong->doBlockDiagonal(this->pm_, this->GetM(), -1);
    }
    // of a block-diagonal operator (a prolongation / restriction of this class)
    void Transpose(GlobalMatrix<ValueType>* T) const
    {
        assert(T != NULL && T != this);
        if(this->m_coupled)
        {
            this->doTransposeCoupled(T);
            return;
        }
        RAMD_EXPECT(this->m_ghost.GetNnz() == 0);
        this->m_interior.Transpose(&T->m_interior);
        if(this->m_interior.GetNnz() == 0) // (nothing to transpose: an empty operator of the transposed shape)
            T->m_interior.AllocateCSR("transposed", 0, this->m_interior.GetN(), this->m_interior.GetM());
        T->doBlockDiagonal(this->pm_, this->GetN(), this->GetM());
    }
    // this = R A P for block-diagonal R and P: interior = R_i A_i P_i; ghost = R_i A_g P_g, where P_g holds the
    // prolongation rows of the ghost columns of A (the neighbours' boundary rows), received here
    void TripleMatrixProduct(const GlobalMatrix<ValueType>& R, const GlobalMatrix<ValueType>& A,
                             const GlobalMatrix<ValueType>& P)
    {
        assert(&R != this && &A != this && &P != this);
        if(P.m_coupled)
        {
            this->doGalerkinCoupled(A, P);
            return;
        }
        RAMD_EXPECT(R.m_ghost.GetNnz() == 0 && P.m_ghost.GetNnz() == 0);
        this->m_interior.CloneBackend(A.m_interior);
        this->m_ghost.CloneBackend(A.m_interior);
        this->m_interior.TripleMatrixProduct(R.m_interior, A.m_interior, P.m_interior);
        const ParallelManager* fpm = A.pm_;
        const int64_t          nc  = P.m_interior.GetN();
        RAMD_EXPECT(sizeof(ValueType) > 4 || nc < (1 << 24)); // (aggregate numbers travel as values of the matrix type)
        if(fpm == NULL)
        {
            this->pm_ = NULL;
            this->m_own_pm.reset();
            return;
        }
        std::shared_ptr<ParallelManager> cpm(new ParallelManager);
        cpm->SetMPICommunicator(fpm->GetComm());
        cpm->SetLocalNrow(nc);
        cpm->SetLocalNcol(nc);
        cpm->SetGlobalNrow(P.GetN());
        cpm->SetGlobalNcol(P.GetN());
        const std::vector<int>&     peers = fpm->peers();
        const std::vector<int64_t>& soff  = fpm->send_offset();
        const std::vector<int64_t>& roff  = fpm->recv_offset();
        const int                   np    = (int)peers.size();
        const int64_t               nsend = fpm->GetNumSenders(), nrecv = fpm->GetNumReceivers();
        std::vector<int>            c_boundary, c_soff(1, 0), c_roff(1, 0);
        // (kmax below is an all-reduce: EVERY rank of a multi-rank run enters this block, also one without neighbours --
        //  its kloc is 0 and its exchanges are empty; only the pairwise exchanges themselves may be skipped per rank)
        const bool                  talk = np > 0 || A.m_halo_plan > 0 || fpm->GetNumProcs() > 1;
        if(talk)
        {
            // prolongation rows of my boundary rows, on the host (k-th entry of every row per exchange)
            std::vector<PtrType>   prp((size_t)P.m_interior.GetM() + 1, 0);
            std::vector<int>       pci((size_t)P.m_interior.GetNnz());
            std::vector<ValueType> pva((size_t)P.m_interior.GetNnz());
            if(P.m_interior.GetNnz() > 0)
                P.m_interior.CopyToCSR(prp.data(), pci.data(), pva.data());
            const int* bidx = fpm->GetBoundaryIndex();
            int        kloc = 0;
            for(int64_t s2 = 0; s2 < nsend; ++s2)
                kloc = std::max(kloc, (int)(prp[(size_t)bidx[s2] + 1] - prp[(size_t)bidx[s2]]));
            const int kmax = A.doMaxRanks(kloc);
            std::vector<std::vector<int>>       gcol((size_t)kmax);
            std::vector<std::vector<ValueType>> gval((size_t)kmax);
            LocalVector<ValueType> sbuf, rbuf;
            sbuf.MoveToAccelerator();
            rbuf.MoveToAccelerator();
            sbuf.Allocate("prolongation rows out", nsend);
            rbuf.Allocate("prolongation rows in", nrecv);
            std::vector<ValueType> hs((size_t)nsend), hr((size_t)nrecv);
            for(int k = 0; k < kmax; ++k)
                for(int what = 0; what < 2; ++what) // column (as a number: exact below 2^53 / 2^24), then value
                {
                    for(int64_t s2 = 0; s2 < nsend; ++s2)
                    {
                        const int     b   = bidx[s2];
                        const PtrType at  = prp[(size_t)b] + k;
                        const bool    has = at < prp[(size_t)b + 1];
                        hs[(size_t)s2]    = what == 0 ? (has ? static_cast<ValueType>(pci[(size_t)at]) : static_cast<ValueType>(-1))
                                                      : (has ? pva[(size_t)at] : static_cast<ValueType>(0));
                    }
                    if(nsend > 0)
                        sbuf.CopyFromHostData(hs.data());
                    A.doExchange(sbuf, &rbuf);
                    if(nrecv > 0)
                        rbuf.CopyToHostData(hr.data());
                    if(what == 0)
                    {
                        gcol[(size_t)k].resize((size_t)nrecv);
                        for(int64_t g = 0; g < nrecv; ++g)
                            gcol[(size_t)k][(size_t)g] = (int)hr[(size_t)g];
                    }
                    else
                        gval[(size_t)k] = hr;
                }
            // coarse boundary towards every peer: the distinct aggregates under the boundary rows sent to it, ascending --
            // the receiver numbers its coarse ghost columns by the same rule from what it was sent
            std::vector<int> seen;
            for(int q = 0; q < np; ++q)
            {
                seen.clear();
                for(int64_t s2 = soff[(size_t)q]; s2 < soff[(size_t)q + 1]; ++s2)
                    for(PtrType at = prp[(size_t)bidx[s2]]; at < prp[(size_t)bidx[s2] + 1]; ++at)
                        seen.push_back(pci[(size_t)at]);
                std::sort(seen.begin(), seen.end());
                seen.erase(std::unique(seen.begin(), seen.end()), seen.end());
                c_boundary.insert(c_boundary.end(), seen.begin(), seen.end());
                c_soff.push_back((int)c_boundary.size());
            }
            // P_g: one row per ghost column of A, columns = coarse ghost columns
            std::vector<PtrType>   grp((size_t)nrecv + 1, 0);
            std::vector<int>       gci;
            std::vector<ValueType> gva;
            int                    ncg = 0;
            for(int q = 0; q < np; ++q)
            {
                seen.clear();
                for(int64_t g = roff[(size_t)q]; g < roff[(size_t)q + 1]; ++g)
                    for(int k = 0; k < kmax; ++k)
                        if(gcol[(size_t)k][(size_t)g] >= 0)
                            seen.push_back(gcol[(size_t)k][(size_t)g]);
                std::sort(seen.begin(), seen.end());
                seen.erase(std::unique(seen.begin(), seen.end()), seen.end());
                for(int64_t g = roff[(size_t)q]; g < roff[(size_t)q + 1]; ++g)
                {
                    for(int k = 0; k < kmax; ++k)
                        if(gcol[(size_t)k][(size_t)g] >= 0)
                        {
                            const int at = (int)(std::lower_bound(seen.begin(), seen.end(), gcol[(size_t)k][(size_t)g]) - seen.begin());
                            gci.push_back(ncg + at);
                            gva.push_back(gval[(size_t)k][(size_t)g]);
                        }
                    grp[(size_t)g + 1] = (PtrType)gci.size();
                }
                ncg += (int)seen.size();
                c_roff.push_back(ncg);
            }
            this->m_ghost.Clear();
            if(A.m_ghost.GetNnz() > 0 && !gci.empty() && nc > 0)
            {
                LocalMatrix<ValueType> Pg, Ag, AgPg;
                Pg.CloneBackend(A.m_interior);
                Pg.AllocateCSR("prolongation rows of the ghost columns", (int64_t)gci.size(), nrecv, ncg);
                Pg.CopyFromCSR(grp.data(), gci.data(), gva.data());
                Ag.CloneFrom(A.m_ghost);
                if(Ag.GetFormat() != CSR)
                    Ag.ConvertTo(CSR);
                AgPg.CloneBackend(A.m_interior);
                AgPg.MatrixMult(Ag, Pg);
                this->m_ghost.MatrixMult(R.m_interior, AgPg);
            }
            if(this->m_ghost.GetNnz() == 0)
                this->m_ghost.AllocateCSR("Ghost of the coarse operator", 0, nc, ncg);
        }
        else
            for(int q = 0; q < np; ++q)
            {
                c_soff.push_back(0);
                c_roff.push_back(0);
            }
        cpm->SetBoundaryIndex((int)c_boundary.size(), c_boundary.data());
        cpm->SetReceivers(np, peers.data(), c_roff.data());
        cpm->SetSenders(np, peers.data(), c_soff.data());
        this->m_own_pm = cpm;
        this->pm_      = cpm.get();
        this->doInitHalo();
    }

private:
    // ==== the coupled form ================================================================================
    struct AmgBlock // what AMGPMISAggregate leaves for the AMG*Aggregation call that follows it
    {
        LocalMatrix<ValueType> block;   // [interior | ghost]
        LocalVector<int>       numbers; // global number of every node of the extended block
        std::vector<int64_t>   offsets; // first row of every rank, and the global size
        int64_t                first_row = 0, agg_first = 0, agg_mine = 0, agg_total = 0;
    };
    bool doCoupled(void) const
    {
        static const bool off = [] {
            const char* e = std::getenv("RAMD_GLOBAL_AMG");
            return e != NULL && std::string(e) == "decoupled";
        }();
        return !off && this->pm_ != NULL && this->pm_->GetNumProcs() > 1;
    }
    // [interior | ghost] of this operator as one CSR block on the device
    void doRowBlock(LocalMatrix<ValueType>* block) const
    {
        LocalMatrix<ValueType>        ci, cg;
        const LocalMatrix<ValueType>* pi = &this->m_interior;
        const LocalMatrix<ValueType>* pg = &this->m_ghost;
        if(pi->GetFormat() != CSR)
        {
            ci.CloneFrom(*pi);
            ci.ConvertTo(CSR);
            pi = &ci;
        }
        if(pg->GetNnz() > 0 && pg->GetFormat() != CSR)
        {
            cg.CloneFrom(*pg);
            cg.ConvertTo(CSR);
            pg = &cg;
        }
        block->MoveToAccelerator();
        // (a ghost part without entries still has its columns: the halo pattern fixes their number)
        RAMD_CHECK(ramd_mat_merge_columns(pi->handle(), pg->GetNnz() > 0 ? pg->handle() : NULL,
                                          this->pm_->GetNumReceivers(), block->handle()));
    }
    // first entry of every rank's share of `local` items in rank order, and the total (one all-reduce)
    static std::vector<int64_t> doRankOffsets(const ParallelManager* pm, int64_t local)
    {
        const int            P = pm->GetNumProcs();
        std::vector<int64_t> all((size_t)P, 0);
        RAMD_CHECK(ramd_comm_allgather_i64(pm->GetComm(), &local, 1, all.data()));
        std::vector<int64_t> off((size_t)P + 1, 0);
        for(int k = 0; k < P; ++k)
            off[(size_t)k + 1] = off[(size_t)k] + all[(size_t)k];
        return off;
    }
    // one exchange of this operator's halo pattern with 8-byte entries whatever the value type (setup data: numbers)
    void doExchangeD(const std::vector<double>& send, std::vector<double>* recv) const
    {
        const int64_t ns = this->pm_->GetNumSenders(), nr = this->pm_->GetNumReceivers();
        RAMD_EXPECT((int64_t)send.size() == ns);
        LocalVector<double> s, r;
        s.MoveToAccelerator();
        r.MoveToAccelerator();
        s.Allocate("setup data out", ns);
        r.Allocate("setup data in", nr);
        if(ns > 0)
            s.CopyFromHostData(send.data());
        RAMD_CHECK(ramd_comm_halo_begin_plan(this->pm_->GetComm(), this->m_halo_plan, s.handle(), r.handle(),
                                             (int)this->pm_->peers().size(), this->pm_->peers().data(),
                                             this->pm_->send_offset().data(), this->pm_->recv_offset().data()));
        RAMD_CHECK(ramd_comm_halo_end(this->pm_->GetComm()));
        recv->assign((size_t)nr, 0.0);
        if(nr > 0)
            r.CopyToHostData(recv->data());
    }
    // messages of any length between any two ranks (out[q]: what goes to rank q; returns what every rank sent here).
    // Lengths first -- one all-gather of every rank's P lengths -- then ONE exchange in pairs.  A collective: every rank of the communicator calls it.
    static std::vector<std::vector<double>> doTalk(const ParallelManager* pm, const std::vector<std::vector<double>>& out)
    {
        const int P = pm->GetNumProcs(), r = pm->GetRank();
        RAMD_EXPECT((int)out.size() == P);
        std::vector<int64_t> len((size_t)P, 0), all((size_t)P * P, 0), nin((size_t)P, 0);
        for(int q = 0; q < P; ++q)
            len[(size_t)q] = (int64_t)out[(size_t)q].size();
        RAMD_CHECK(ramd_comm_allgather_i64(pm->GetComm(), len.data(), P, all.data()));
        for(int q = 0; q < P; ++q)
            nin[(size_t)q] = q == r ? 0 : all[(size_t)q * P + r];
        std::vector<int>     peers;
        std::vector<int64_t> so(1, 0), ro(1, 0);
        for(int q = 0; q < P; ++q)
            if(q != r && (!out[(size_t)q].empty() || nin[(size_t)q] > 0))
            {
                peers.push_back(q);
                so.push_back(so.back() + (int64_t)out[(size_t)q].size());
                ro.push_back(ro.back() + nin[(size_t)q]);
            }
        std::vector<std::vector<double>> in((size_t)P);
        if(peers.empty())
            return in;
        std::vector<double> hs((size_t)so.back()), hr((size_t)ro.back());
        for(size_t k = 0; k < peers.size(); ++k)
            std::copy(out[(size_t)peers[k]].begin(), out[(size_t)peers[k]].end(), hs.begin() + so[k]);
        LocalVector<double> s, rv;
        s.MoveToAccelerator();
        rv.MoveToAccelerator();
        s.Allocate("messages out", so.back());
        rv.Allocate("messages in", ro.back());
        if(!hs.empty())
            s.CopyFromHostData(hs.data());
        RAMD_CHECK(ramd_comm_halo_begin(pm->GetComm(), s.handle(), rv.handle(), (int)peers.size(), peers.data(), so.data(),
                                        ro.data()));
        RAMD_CHECK(ramd_comm_halo_end(pm->GetComm()));
        if(!hr.empty())
            rv.CopyToHostData(hr.data());
        for(size_t k = 0; k < peers.size(); ++k)
            in[(size_t)peers[k]].assign(hr.begin() + ro[k], hr.begin() + ro[k + 1]);
        return in;
    }
    static int doOwner(const std::vector<int64_t>& offsets, int64_t id)
    {
        return (int)(std::upper_bound(offsets.begin(), offsets.end(), id) - offsets.begin()) - 1;
    }
    // this = the rows (rp, gcol, val) of a distributed operator whose columns are GLOBAL numbers of a space distributed
    // by col_offsets: interior = the columns of this rank, ghost = the others in ascending order (= grouped by owner),
    // and the halo pattern that brings the ghost columns' vector entries here -- every rank tells the owners which of
    // their entries it needs (parallel_manager.cpp GenerateFromGhostColumnsWithParent_ and the boundary exchange after
    // it do this for the reference).  A collective.
    void doFromGlobalColumns(const ParallelManager* like, int64_t global_nrow, const std::vector<int64_t>& col_offsets,
                             const std::vector<PtrType>& rp, const std::vector<int64_t>& gcol,
                             const std::vector<ValueType>& val)
    {
        const int     r = like->GetRank();
        const int64_t n = (int64_t)rp.size() - 1, c0 = col_offsets[(size_t)r], nc = col_offsets[(size_t)r + 1] - c0;
        std::vector<int64_t> ghost;
        for(int64_t g : gcol)
            if(g < c0 || g >= c0 + nc)
                ghost.push_back(g);
        std::sort(ghost.begin(), ghost.end());
        ghost.erase(std::unique(ghost.begin(), ghost.end()), ghost.end());
        std::vector<PtrType>   irp((size_t)n + 1, 0), grp((size_t)n + 1, 0);
        std::vector<int>       ici, gci;
        std::vector<ValueType> iva, gva;
        std::vector<std::pair<int, ValueType>> row;
        for(int64_t i = 0; i < n; ++i)
        {
            for(int part = 0; part < 2; ++part)
            {
                row.clear();
                for(PtrType j = rp[(size_t)i]; j < rp[(size_t)i + 1]; ++j)
                {
                    const int64_t g   = gcol[(size_t)j];
                    const bool    own = g >= c0 && g < c0 + nc;
                    if(own == (part == 0))
                        row.emplace_back(own ? (int)(g - c0)
                                             : (int)(std::lower_bound(ghost.begin(), ghost.end(), g) - ghost.begin()),
                                         val[(size_t)j]);
                }
                std::stable_sort(row.begin(), row.end(),
                                 [](const std::pair<int, ValueType>& a, const std::pair<int, ValueType>& b) { return a.first < b.first; });
                for(const std::pair<int, ValueType>& e : row)
                {
                    (part == 0 ? ici : gci).push_back(e.first);
                    (part == 0 ? iva : gva).push_back(e.second);
                }
            }
            irp[(size_t)i + 1] = (PtrType)ici.size();
            grp[(size_t)i + 1] = (PtrType)gci.size();
        }
        this->m_interior.MoveToAccelerator();
        this->m_ghost.MoveToAccelerator();
        this->m_interior.AllocateCSR("Interior", (int64_t)ici.size(), n, nc);
        if(!ici.empty())
            this->m_interior.CopyFromCSR(irp.data(), ici.data(), iva.data());
        this->m_ghost.AllocateCSR("Ghost", (int64_t)gci.size(), n, (int64_t)ghost.size());
        if(!gci.empty())
            this->m_ghost.CopyFromCSR(grp.data(), gci.data(), gva.data());
        this->doAdoptGhosts(like, global_nrow, col_offsets, ghost);
    }
    // the halo pattern that brings the vector entries of the ghost columns `ghost` (global numbers, ascending) here: who
    // owns what I need, and the owners learn it from me.  interior / ghost are in place.  A collective.
    void doAdoptGhosts(const ParallelManager* like, int64_t global_nrow, const std::vector<int64_t>& col_offsets,
                       const std::vector<int64_t>& ghost)
    {
        const int     P = like->GetNumProcs(), r = like->GetRank();
        const int64_t n = this->m_interior.GetM(), c0 = col_offsets[(size_t)r], nc = col_offsets[(size_t)r + 1] - c0;
        std::vector<std::vector<double>> ask((size_t)P);
        for(int64_t g : ghost)
            ask[(size_t)doOwner(col_offsets, g)].push_back((double)g);
        const std::vector<std::vector<double>> asked = doTalk(like, ask);
        std::vector<int> peers, boundary, soff(1, 0), roff(1, 0);
        for(int q = 0; q < P; ++q)
            if(!ask[(size_t)q].empty() || !asked[(size_t)q].empty())
            {
                RAMD_EXPECT(q != r);
                peers.push_back(q);
                for(double g : asked[(size_t)q])
                {
                    const int64_t l = (int64_t)std::llround(g) - c0;
                    RAMD_EXPECT(l >= 0 && l < nc);
                    boundary.push_back((int)l);
                }
                soff.push_back((int)boundary.size());
                roff.push_back(roff.back() + (int)ask[(size_t)q].size());
            }
        std::shared_ptr<ParallelManager> pm(new ParallelManager);
        pm->SetMPICommunicator(like->GetComm());
        pm->SetLocalNrow(n);
        pm->SetLocalNcol(nc);
        pm->SetGlobalNrow(global_nrow);
        pm->SetGlobalNcol(col_offsets.back());
        pm->SetBoundaryIndex((int)boundary.size(), boundary.data());
        pm->SetReceivers((int)peers.size(), peers.data(), roff.data());
        pm->SetSenders((int)peers.size(), peers.data(), soff.data());
        this->m_own_pm      = pm;
        this->pm_           = pm.get();
        this->m_coupled     = true;
        this->m_reverse     = false;
        this->m_ghost_cols  = ghost;
        this->m_col_offsets = col_offsets;
        this->m_amg.reset();
        this->doInitHalo();
    }
    // rows of P for the rows of this rank, columns = global aggregate numbers
    void doProlongCoupled(int smoothed, ValueType relax, int lumping_strat, const LocalVector<int>& connections,
                          const LocalVector<int>& aggregates, const LocalVector<int>& aggregate_root_nodes,
                          GlobalMatrix<ValueType>* prolong) const
    {
        const AmgBlock&        w = *this->m_amg;
        LocalMatrix<ValueType> Pl;
        Pl.MoveToAccelerator();
        RAMD_CHECK(ramd_mat_amg_prolong_global(w.block.handle(), smoothed, (double)relax, lumping_strat, connections.handle(),
                                               aggregates.handle(), aggregate_root_nodes.handle(), w.agg_total, Pl.handle()));
        const int64_t          n = Pl.GetM(), nnz = Pl.GetNnz();
        std::vector<PtrType>   rp((size_t)n + 1, 0);
        std::vector<int>       ci((size_t)nnz);
        std::vector<ValueType> va((size_t)nnz);
        if(nnz > 0)
            Pl.CopyToCSR(rp.data(), ci.data(), va.data());
        std::vector<int64_t> gc(ci.begin(), ci.end());
        const std::vector<int64_t> coff = doRankOffsets(this->pm_, w.agg_mine);
        RAMD_EXPECT(coff[(size_t)this->pm_->GetRank()] == w.agg_first && coff.back() == w.agg_total);
        prolong->m_interior.CloneBackend(this->m_interior);
        prolong->doDropHaloPlan();
        prolong->doFromGlobalColumns(this->pm_, this->GetM(), coff, rp, gc, va);
        this->m_amg.reset();
    }
    // R = P^T in the reverse form: interior = P_int^T, ghost = P_ghost^T (one row per ghost column of P); Apply adds the
    // ghost rows' results into the rows of their owners -- the exchange of P run backwards
    void doTransposeCoupled(GlobalMatrix<ValueType>* T) const
    {
        const ParallelManager* pp = this->pm_;
        const int64_t          n = this->m_interior.GetM(), nc = this->m_interior.GetN(), ng = (int64_t)this->m_ghost_cols.size();
        T->doDropHaloPlan();
        T->m_interior.CloneBackend(this->m_interior);
        T->m_ghost.CloneBackend(this->m_interior);
        if(this->m_interior.GetNnz() > 0)
            this->m_interior.Transpose(&T->m_interior);
        else
            T->m_interior.AllocateCSR("transposed", 0, nc, n);
        if(this->m_ghost.GetNnz() > 0)
            this->m_ghost.Transpose(&T->m_ghost);
        else
            T->m_ghost.AllocateCSR("transposed ghost", 0, ng, n);
        // where the received partial sums go: row boundary[s] of the result += entry s (a 0/1 operator, so that equal
        // targets -- a coarse node needed by two ranks -- add up in a fixed order)
        const int64_t          ns = pp->GetNumSenders();
        std::vector<PtrType>   srp((size_t)nc + 1, 0);
        std::vector<int>       sci((size_t)ns);
        std::vector<ValueType> sva((size_t)ns, static_cast<ValueType>(1));
        const int*             b = pp->GetBoundaryIndex();
        for(int64_t s2 = 0; s2 < ns; ++s2)
            ++srp[(size_t)b[s2] + 1];
        for(int64_t i = 0; i < nc; ++i)
            srp[(size_t)i + 1] += srp[(size_t)i];
        std::vector<PtrType> at(srp.begin(), srp.end() - 1);
        for(int64_t s2 = 0; s2 < ns; ++s2)
            sci[(size_t)at[(size_t)b[s2]]++] = (int)s2;
        T->m_scatter.CloneBackend(this->m_interior);
        T->m_scatter.AllocateCSR("reverse halo", ns, nc, ns);
        if(ns > 0)
            T->m_scatter.CopyFromCSR(srp.data(), sci.data(), sva.data());
        // the pattern of P with the two directions swapped
        std::vector<int> peers(pp->peers()), soff, roff;
        for(int64_t o : pp->recv_offset())
            soff.push_back((int)o);
        for(int64_t o : pp->send_offset())
            roff.push_back((int)o);
        if(soff.empty())
        {
            soff.push_back(0);
            roff.push_back(0);
        }
        std::shared_ptr<ParallelManager> pm(new ParallelManager);
        pm->SetMPICommunicator(pp->GetComm());
        pm->SetLocalNrow(nc);
        pm->SetLocalNcol(n);
        pm->SetGlobalNrow(pp->GetGlobalNcol());
        pm->SetGlobalNcol(pp->GetGlobalNrow());
        pm->SetReceivers((int)peers.size(), peers.data(), roff.data());
        pm->SetSenders((int)peers.size(), peers.data(), soff.data());
        T->m_own_pm  = pm;
        T->pm_       = pm.get();
        T->m_coupled = true;
        T->m_reverse = true;
        T->m_ghost_cols.clear();
        T->m_col_offsets.clear();
        T->m_send.MoveToAccelerator();
        T->m_send.Allocate("reverse send buffer", ng);
        T->m_recv.MoveToAccelerator();
        T->m_recv.Allocate("reverse recv buffer", ns);
    }
    void doApplyReverse(const GlobalVector<ValueType>& in, GlobalVector<ValueType>* out) const
    {
        const bool comm = !this->pm_->peers().empty();
        if(comm)
        {
            if(this->m_ghost.GetNnz() > 0)
                this->m_ghost.Apply(in.m_owned, &this->m_send);
            else if(this->m_send.GetSize() > 0)
                this->m_send.Zeros();
            RAMD_CHECK(ramd_comm_halo_begin(this->pm_->GetComm(), this->m_send.handle(), this->m_recv.handle(),
                                            (int)this->pm_->peers().size(), this->pm_->peers().data(),
                                            this->pm_->send_offset().data(), this->pm_->recv_offset().data()));
        }
        this->m_interior.Apply(in.m_owned, &out->m_owned);
        if(comm)
        {
            RAMD_CHECK(ramd_comm_halo_end(this->pm_->GetComm()));
            if(this->m_scatter.GetNnz() > 0)
                this->m_scatter.ApplyAdd(this->m_recv, static_cast<ValueType>(1), &out->m_owned);
        }
    }
    // this = P^T A P, P coupled (see the head of this section)
    void doGalerkinCoupled(const GlobalMatrix<ValueType>& A, const GlobalMatrix<ValueType>& P)
    {
        const ParallelManager*      fpm  = A.pm_;
        const int                   Pn   = fpm->GetNumProcs(), r = fpm->GetRank();
        const std::vector<int64_t>& coff = P.m_col_offsets;
        const int64_t n = A.m_interior.GetM(), ngA = fpm->GetNumReceivers(), ns = fpm->GetNumSenders();
        const int64_t c0 = coff[(size_t)r], nc = coff[(size_t)r + 1] - c0;
        RAMD_EXPECT(P.m_interior.GetM() == n && P.m_interior.GetN() == nc);
        // rows of P on the host, global columns
        std::vector<PtrType>   prp((size_t)n + 1, 0);
        std::vector<int64_t>   pgc;
        std::vector<ValueType> pva;
        {
            std::vector<PtrType>   irp((size_t)n + 1, 0), grp((size_t)n + 1, 0);
            std::vector<int>       ici((size_t)P.m_interior.GetNnz()), gci((size_t)P.m_ghost.GetNnz());
            std::vector<ValueType> iva(ici.size()), gva(gci.size());
            if(!ici.empty())
                P.m_interior.CopyToCSR(irp.data(), ici.data(), iva.data());
            if(!gci.empty())
                P.m_ghost.CopyToCSR(grp.data(), gci.data(), gva.data());
            pgc.reserve(ici.size() + gci.size());
            pva.reserve(ici.size() + gci.size());
            for(int64_t i = 0; i < n; ++i)
            {
                for(PtrType j = irp[(size_t)i]; j < irp[(size_t)i + 1]; ++j)
                {
                    pgc.push_back(c0 + ici[(size_t)j]);
                    pva.push_back(iva[(size_t)j]);
                }
                for(PtrType j = grp[(size_t)i]; j < grp[(size_t)i + 1]; ++j)
                {
                    pgc.push_back(P.m_ghost_cols[(size_t)gci[(size_t)j]]);
                    pva.push_back(gva[(size_t)j]);
                }
                prp[(size_t)i + 1] = (PtrType)pgc.size();
            }
        }
        // ... and the rows of P of A's ghost nodes from their owners: the k-th entry of every boundary row per exchange
        const int* bidx = fpm->GetBoundaryIndex();
        int        kloc = 0;
        for(int64_t s2 = 0; s2 < ns; ++s2)
            kloc = std::max(kloc, (int)(prp[(size_t)bidx[s2] + 1] - prp[(size_t)bidx[s2]]));
        const int kmax = A.doMaxRanks(kloc);
        std::vector<std::vector<double>> gcol((size_t)kmax), gval((size_t)kmax);
        std::vector<double>              hs((size_t)ns);
        for(int k = 0; k < kmax; ++k)
            for(int what = 0; what < 2; ++what)
            {
                for(int64_t s2 = 0; s2 < ns; ++s2)
                {
                    const PtrType at  = prp[(size_t)bidx[s2]] + k;
                    const bool    has = at < prp[(size_t)bidx[s2] + 1];
                    hs[(size_t)s2]    = what == 0 ? (has ? (double)pgc[(size_t)at] : -1.0) : (has ? (double)pva[(size_t)at] : 0.0);
                }
                A.doExchangeD(hs, what == 0 ? &gcol[(size_t)k] : &gval[(size_t)k]);
            }
        // compact numbering of the coarse columns seen here: mine first, then the others ascending
        std::vector<int64_t> others;
        for(int64_t g : pgc)
            if(g < c0 || g >= c0 + nc)
                others.push_back(g);
        for(int k = 0; k < kmax; ++k)
            for(double g : gcol[(size_t)k])
                if(g >= 0.0 && ((int64_t)g < c0 || (int64_t)g >= c0 + nc))
                    others.push_back((int64_t)g);
        std::sort(others.begin(), others.end());
        others.erase(std::unique(others.begin(), others.end()), others.end());
        const int64_t ncc = nc + (int64_t)others.size();
        auto compact = [&](int64_t g) {
            return (g >= c0 && g < c0 + nc) ? (int)(g - c0)
                                            : (int)(nc + (std::lower_bound(others.begin(), others.end(), g) - others.begin()));
        };
        // P over the extended block (its own rows, then the ghost nodes' rows), rows sorted
        std::vector<PtrType>   xrp((size_t)(n + ngA) + 1, 0);
        std::vector<int>       xci;
        std::vector<ValueType> xva;
        std::vector<std::pair<int, ValueType>> row;
        auto flush = [&](int64_t i) {
            std::stable_sort(row.begin(), row.end(),
                             [](const std::pair<int, ValueType>& a, const std::pair<int, ValueType>& b) { return a.first < b.first; });
            for(const std::pair<int, ValueType>& e : row)
            {
                xci.push_back(e.first);
                xva.push_back(e.second);
            }
            xrp[(size_t)i + 1] = (PtrType)xci.size();
            row.clear();
        };
        for(int64_t i = 0; i < n; ++i)
        {
            for(PtrType j = prp[(size_t)i]; j < prp[(size_t)i + 1]; ++j)
                row.emplace_back(compact(pgc[(size_t)j]), pva[(size_t)j]);
            flush(i);
        }
        const PtrType nnz_own = xrp[(size_t)n];
        for(int64_t g = 0; g < ngA; ++g)
        {
            for(int k = 0; k < kmax; ++k)
                if(gcol[(size_t)k][(size_t)g] >= 0.0)
                    row.emplace_back(compact((int64_t)gcol[(size_t)k][(size_t)g]), static_cast<ValueType>(gval[(size_t)k][(size_t)g]));
            flush(n + g);
        }
        // the products on the device: C = P_own^T ([A_int | A_ghost] P_ext), rows and columns in the compact numbering
        LocalMatrix<ValueType> C;
        C.MoveToAccelerator();
        if(nnz_own > 0)
        {
            LocalMatrix<ValueType> Ablk, Px, Pown, Pt, AP;
            A.doRowBlock(&Ablk);
            const int64_t xrows = Ablk.GetN();
            RAMD_EXPECT(xrows == n + ngA);
            Px.MoveToAccelerator();
            Px.AllocateCSR("P over the extended block", (int64_t)xrp[(size_t)xrows], xrows, ncc);
            Px.CopyFromCSR(xrp.data(), xci.data(), xva.data());
            Pown.MoveToAccelerator();
            Pown.AllocateCSR("P of the own rows", (int64_t)nnz_own, n, ncc);
            Pown.CopyFromCSR(xrp.data(), xci.data(), xva.data());
            AP.MoveToAccelerator();
            AP.MatrixMult(Ablk, Px);
            Pt.MoveToAccelerator();
            Pown.Transpose(&Pt);
            C.MatrixMult(Pt, AP);
        }
        else
            C.AllocateCSR("empty product", 0, ncc, ncc);
        // coarse rows of other ranks go to their owners as (row, column, value) triplets -- values as they are
        std::vector<PtrType>   crp((size_t)ncc + 1, 0);
        std::vector<int>       cci((size_t)C.GetNnz());
        std::vector<ValueType> cva((size_t)C.GetNnz());
        if(C.GetNnz() > 0)
            C.CopyToCSR(crp.data(), cci.data(), cva.data());
        auto global_of = [&](int c) { return c < nc ? c0 + c : others[(size_t)(c - nc)]; };
        std::vector<std::vector<double>> out((size_t)Pn);
        for(int64_t I = nc; I < ncc; ++I)
        {
            const int64_t gI = others[(size_t)(I - nc)];
            const int     q  = doOwner(coff, gI);
            for(PtrType j = crp[(size_t)I]; j < crp[(size_t)I + 1]; ++j)
            {
                out[(size_t)q].push_back((double)gI);
                out[(size_t)q].push_back((double)global_of(cci[(size_t)j]));
                out[(size_t)q].push_back((double)cva[(size_t)j]);
            }
        }
        const std::vector<std::vector<double>> in = doTalk(fpm, out);
        // ghost columns of the coarse operator: the foreign columns of my rows, in what I computed and in what arrived
        std::vector<int64_t> ghost;
        for(PtrType j = 0; j < crp[(size_t)nc]; ++j)
            if(cci[(size_t)j] >= nc)
                ghost.push_back(others[(size_t)(cci[(size_t)j] - nc)]);
        for(int q = 0; q < Pn; ++q)
            for(size_t t = 0; t + 2 < in[(size_t)q].size(); t += 3)
            {
                const int64_t J = (int64_t)std::llround(in[(size_t)q][t + 1]);
                if(J < c0 || J >= c0 + nc)
                    ghost.push_back(J);
            }
        std::sort(ghost.begin(), ghost.end());
        ghost.erase(std::unique(ghost.begin(), ghost.end()), ghost.end());
        const int64_t ngh = (int64_t)ghost.size();
        auto final_of = [&](int64_t g) {
            return (g >= c0 && g < c0 + nc) ? (int)(g - c0)
                                            : (int)(nc + (std::lower_bound(ghost.begin(), ghost.end(), g) - ghost.begin()));
        };
        // my rows over [own columns | ghost columns]: the part computed here, then one operator per sending rank; the sums
        // are device MatrixAdds in that order (entries of one (row, column) are added own part first, then rank by rank)
        LocalMatrix<ValueType> M;
        M.MoveToAccelerator();
        {
            std::vector<PtrType>   mrp(crp.begin(), crp.begin() + nc + 1);
            std::vector<int>       mci((size_t)mrp[(size_t)nc]);
            std::vector<ValueType> mva(cva.begin(), cva.begin() + mrp[(size_t)nc]);
            for(size_t j = 0; j < mci.size(); ++j)
                mci[j] = final_of(global_of(cci[j])); // (ascending in a row: both numberings ascend with the global one)
            M.AllocateCSR("own part of the coarse rows", (int64_t)mci.size(), nc, nc + ngh);
            if(!mci.empty())
                M.CopyFromCSR(mrp.data(), mci.data(), mva.data());
        }
        for(int q = 0; q < Pn; ++q)
        {
            const std::vector<double>& msg = in[(size_t)q];
            if(msg.empty())
                continue;
            const size_t        nt = msg.size() / 3;
            std::vector<size_t> ord(nt);
            std::vector<int>    ti(nt), tj(nt);
            for(size_t t = 0; t < nt; ++t)
            {
                ord[t]          = t;
                const int64_t I = (int64_t)std::llround(msg[3 * t]) - c0;
                RAMD_EXPECT(I >= 0 && I < nc);
                ti[t] = (int)I;
                tj[t] = final_of((int64_t)std::llround(msg[3 * t + 1]));
            }
            std::sort(ord.begin(), ord.end(), [&](size_t a2, size_t b2) { return ti[a2] != ti[b2] ? ti[a2] < ti[b2] : tj[a2] < tj[b2]; });
            std::vector<PtrType>   rrp((size_t)nc + 1, 0);
            std::vector<int>       rci(nt);
            std::vector<ValueType> rva(nt);
            for(size_t k = 0; k < nt; ++k)
            {
                ++rrp[(size_t)ti[ord[k]] + 1];
                rci[k] = tj[ord[k]];
                rva[k] = static_cast<ValueType>(msg[3 * ord[k] + 2]);
            }
            for(int64_t I = 0; I < nc; ++I)
                rrp[(size_t)I + 1] += rrp[(size_t)I];
            LocalMatrix<ValueType> Rq;
            Rq.MoveToAccelerator();
            Rq.AllocateCSR("coarse rows computed elsewhere", (int64_t)nt, nc, nc + ngh);
            Rq.CopyFromCSR(rrp.data(), rci.data(), rva.data());
            M.MatrixAdd(Rq, static_cast<ValueType>(1), static_cast<ValueType>(1), true);
        }
        this->doDropHaloPlan();
        this->m_interior.CloneBackend(A.m_interior);
        this->m_ghost.CloneBackend(A.m_interior);
        if(nc > 0 && M.GetNnz() > 0)
            M.ExtractSubMatrix(0, 0, nc, nc, &this->m_interior);
        else
            this->m_interior.AllocateCSR("Interior", 0, nc, nc);
        if(nc > 0 && ngh > 0 && M.GetNnz() > 0)
            M.ExtractSubMatrix(0, nc, nc, ngh, &this->m_ghost);
        else
            this->m_ghost.AllocateCSR("Ghost", 0, nc, ngh);
        this->doAdoptGhosts(fpm, coff.back(), coff, ghost);
    }
    mutable std::shared_ptr<AmgBlock> m_amg;
    bool                   m_coupled = false; // built by the coupled setup: m_ghost_cols / m_col_offsets are valid
    bool                   m_reverse = false; // a restriction in the reverse form (doTransposeCoupled)
    std::vector<int64_t>   m_ghost_cols;      // global number of every ghost column
    std::vector<int64_t>   m_col_offsets;     // first column of every rank, and the global column count
    LocalMatrix<ValueType> m_scatter;         // reverse form: received partial sums -> rows

    // a block-diagonal operator between two row-block distributions: a ParallelManager of its own without neighbours
    // (ncol < 0: the global column count is the sum of the local ones)
    void doBlockDiagonal(const ParallelManager* like, int64_t global_nrow, int64_t global_ncol)
    {
        this->m_ghost.Clear();
        this->doDropHaloPlan();
        if(like == NULL)
        {
            this->pm_ = NULL;
            this->m_own_pm.reset();
            return;
        }
        std::shared_ptr<ParallelManager> pm(new ParallelManager);
        pm->SetMPICommunicator(like->GetComm());
        pm->SetLocalNrow(this->m_interior.GetM());
        pm->SetLocalNcol(this->m_interior.GetN());
        pm->SetGlobalNrow(global_nrow);
        if(global_ncol < 0)
            global_ncol = (int64_t)std::llround(doSumRanks(like, (double)this->m_interior.GetN()));
        pm->SetGlobalNcol(global_ncol);
        this->m_own_pm = pm;
        this->pm_      = pm.get();
    }
    static double doSumRanks(const ParallelManager* pm, double local)
    {
        if(pm == NULL || pm->GetNumProcs() == 1)
            return local;
        const int slot = RAMD_NSCALARS - 2;
        RAMD_CHECK(ramd_scalars_set(slot, local));
        RAMD_CHECK(ramd_comm_allreduce_scalars(pm->GetComm(), slot, 1));
        double r = 0.0;
        RAMD_CHECK(ramd_scalars_fetch(&r, slot, 1));
        return r;
    }
    // max over ranks of a small non-negative count (the all-reduce sums: one indicator slot per value)
    int doMaxRanks(int local) const
    {
        if(this->pm_ == NULL || this->pm_->GetNumProcs() == 1)
            return local;
        const int cap = 48, first = RAMD_NSCALARS - 8 - cap;
        RAMD_EXPECT(local <= cap);
        for(int k = 0; k < cap; ++k)
            RAMD_CHECK(ramd_scalars_set(first + k, k < local ? 1.0 : 0.0));
        RAMD_CHECK(ramd_comm_allreduce_scalars(this->pm_->GetComm(), first, cap));
        double v[48];
        RAMD_CHECK(ramd_scalars_fetch(v, first, cap));
        int r = 0;
        for(int k = 0; k < cap; ++k)
            if(v[k] > 0.5)
                r = k + 1;
        return r;
    }
    // one halo exchange of per-boundary-row values with the pattern of this matrix (device buffers)
    void doExchange(const LocalVector<ValueType>& send, LocalVector<ValueType>* recv) const
    {
        RAMD_CHECK(ramd_comm_halo_begin_plan(this->pm_->GetComm(), this->m_halo_plan, send.handle(), recv->handle(), (int)this->pm_->peers().size(),
                                        this->pm_->peers().data(), this->pm_->send_offset().data(),
                                        this->pm_->recv_offset().data()));
        RAMD_CHECK(ramd_comm_halo_end(this->pm_->GetComm()));
    }
    // global_matrix.cpp:4476-4513: halo index vector + device send/recv buffers
    void doInitHalo(void)
    {
        this->doDropHaloPlan();
        this->m_amg.reset();
        if(this->pm_ == NULL)
            return;
        {
            // every rank announces its plan: the ranks agree on the form of the exchange (pairs, or one all-gather when
            // some rank has many neighbours); in the all-gather form a rank without neighbours takes part as well
            int ag = 0;
            RAMD_CHECK(ramd_comm_halo_select(this->pm_->GetComm(), (int)this->pm_->peers().size(), this->pm_->peers().data(),
                                             this->pm_->send_offset().data(), this->pm_->recv_offset().data(), &ag));
            this->m_halo_plan = ag;
            this->m_halo_comm = ag > 0 ? this->pm_->GetComm() : NULL;
            if(ag > 0)
                RAMD_CHECK(ramd_comm_generation(this->m_halo_comm, &this->m_halo_gen));
        }
        if(this->pm_->peers().empty() && this->m_halo_plan == 0)
            return;
        const int nb = this->pm_->GetBoundarySize();
        this->m_halo_rows.MoveToAccelerator();
        this->m_halo_rows.Allocate("halo", nb);
        if(nb > 0)
            this->m_halo_rows.CopyFromHostData(this->pm_->GetBoundaryIndex());
        this->m_send.MoveToAccelerator();
        this->m_send.Allocate("send buffer", this->pm_->GetNumSenders());
        this->m_recv.MoveToAccelerator();
        this->m_recv.Allocate("recv buffer", this->pm_->GetNumReceivers());
    }
    const ParallelManager*         pm_;
    std::shared_ptr<ParallelManager> m_own_pm; // coarse levels and transfer operators own theirs
    LocalMatrix<ValueType>         m_interior;
    LocalMatrix<ValueType>         m_ghost;
    LocalVector<int>               m_halo_rows;
    mutable LocalVector<ValueType> m_send;
    mutable LocalVector<ValueType> m_recv;
    // form of the exchange the ranks agreed on (doInitHalo): 0 = send/recv pairs, k > 0 = the all-gather plan number k
    // handed out by ramd_comm_halo_select (owned by this matrix: released when the plan is replaced or the matrix dies)
    void doDropHaloPlan(void)
    {
        if(this->m_halo_plan > 0 && this->m_halo_comm != NULL)
            (void)ramd_comm_halo_release(this->m_halo_comm, this->m_halo_plan, this->m_halo_gen);
        this->m_halo_plan = 0;
        this->m_halo_comm = NULL;
    }
    int                            m_halo_plan = 0;
    ramd_comm_t                    m_halo_comm = NULL;
    long long                      m_halo_gen  = 0; // ramd_comm_generation of m_halo_comm when the plan was announced
};

// ---- fused-loop helpers for Global objects (see solvers.hpp: _fusable / _fh / _f_apply_dot / _f_allreduce)
template <typename ValueType>
struct _fusable<GlobalMatrix<ValueType>, GlobalVector<ValueType>, ValueType>
{
    static constexpr bool value = true;
};
template <typename ValueType>
inline ramd_vec_t _fh(const GlobalVector<ValueType>& v)
{
    return v.GetInterior().handle();
}
template <typename ValueType>
inline void _f_apply_dot(const GlobalMatrix<ValueType>& A, const GlobalVector<ValueType>& p,
                         GlobalVector<ValueType>* q, int slot)
{
    A.ApplyDot(p, q, slot); // pack | halo || interior SpMV + <p,q> | ghost += and dot correction
}
template <typename ValueType>
inline void _f_apply_dotv(const GlobalMatrix<ValueType>& A, const GlobalVector<ValueType>& x,
                          GlobalVector<ValueType>* y, const GlobalVector<ValueType>& w, int slot)
{
    A.ApplyDotV(x, w, y, slot);
}
template <typename ValueType>
inline void _f_allreduce(const GlobalMatrix<ValueType>& A, int first, int count)
{
    if(A.pm() != NULL) // (a communicator of size 1 returns at once unless RAMD_COMM_FORCE_COLLECTIVES is set)
        RAMD_CHECK(ramd_comm_allreduce_scalars(A.pm()->GetComm(), first, count));
}

// preconditioner_blockjacobi.cpp:80-141: the local preconditioner acts on the interior block only
template <class OperatorType, class VectorType, typename ValueType>
class BlockJacobi : public Preconditioner<OperatorType, VectorType, ValueType>
{
public:
    BlockJacobi()
        : m_local_precond(NULL)
    {
    }
    virtual ~BlockJacobi()
    {
        this->Clear();
    }
    virtual void Print(void) const
    {
        LOG_INFO("BlockJacobi preconditioner with local preconditioner:");
        if(this->m_local_precond)
            this->m_local_precond->Print();
    }
    void Set(Solver<LocalMatrix<ValueType>, LocalVector<ValueType>, ValueType>& precond)
    {
        this->m_local_precond = &precond;
    }
    virtual void Build(void)
    {
        if(this->m_build)
            this->Clear();
        assert(this->m_local_precond != NULL && this->m_op != NULL);
        this->m_build = true;
        this->m_local_precond->SetOperator(this->m_op->GetInterior());
        this->m_local_precond->Build();
    }
    virtual void Clear(void)
    {
        if(this->m_local_precond != NULL)
            this->m_local_precond->Clear();
        this->m_local_precond = NULL;
        this->m_build         = false;
    }
    virtual void Solve(const VectorType& rhs, VectorType* x)
    {
        this->m_local_precond->Solve(rhs.GetInterior(), &x->GetInterior());
    }
    // a local multigrid cycle / nested solver runs reductions of its own: the outer fused loop steps aside
    virtual bool SolveUsesScalarRecord(void) const
    {
        return this->m_local_precond != NULL && this->m_local_precond->SolveUsesScalarRecord();
    }

private:
    Solver<LocalMatrix<ValueType>, LocalVector<ValueType>, ValueType>* m_local_precond;
};

} // namespace rocalution
