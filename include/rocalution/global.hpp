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
        if(stop_ratio > 0.0) // (a `run` without exchanges: every rank places its share by itself)
        {
            this->m_owned.PlaceByTrial(run, tries, stop_ratio, apart_from ? &apart_from->m_owned : nullptr);
            return;
        }
        // every rank has to run `run` equally often (it exchanges halos): all of them take part, or none
        const double mine = (double)this->m_owned.GetSize() * sizeof(ValueType) >= (double)(64 << 20) ? 1.0 : 0.0;
        if(this->pm_ != NULL && this->pm_->GetNumProcs() > 1
           && this->sum_ranks_(mine) < (double)this->pm_->GetNumProcs() - 0.5)
            return;
        this->m_owned.PlaceByTrial(run, tries);
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
        this->m_interior.SetDataPtrCSR(row_offset, col, val, "Interior of " + name, nnz,
                                             this->pm_->GetLocalNrow(), this->pm_->GetLocalNcol());
    }
    void SetGhostDataPtrCSR(PtrType** row_offset, int** col, ValueType** val, std::string name,
                            int64_t nnz)
    {
        assert(this->pm_ != NULL);
        this->m_ghost.SetDataPtrCSR(row_offset, col, val, "Ghost of " + name, nnz,
                                          this->pm_->GetLocalNrow(), this->pm_->GetNumReceivers());
    }
    void MoveToAccelerator(void)
    {
        this->m_interior.MoveToAccelerator();
        this->m_ghost.MoveToAccelerator();
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

    // global_matrix.cpp:924-1009, device-resident: pack | halo over xGMI || interior SpMV | ghost +=
    void Apply(const GlobalVector<ValueType>& in, GlobalVector<ValueType>* out) const
    {
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
    // :2607-3558 AMG*Aggregate / AMG*Aggregation).  MI355X-first form: the aggregates of a rank stay inside its row block
    // ("decoupled" aggregation), so the prolongation and restriction operators are block-diagonal -- no ghost part, no
    // communication when a cycle applies them -- and only the Galerkin product exchanges data, once per level at Build:
    // the prolongation rows of the boundary rows travel to the neighbours through the same halo exchange the SpMV uses.
    // The coarse operator is again interior + ghost with a ParallelManager of its own (coarse boundary = the aggregates
    // the boundary rows belong to).  On one rank this IS the LocalMatrix algorithm, kernel for kernel; over P ranks it is
    // the P-way decoupled algorithm (the reference lets aggregates cross ranks: its hierarchy differs for P > 1).
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
        this->m_interior.AMGPMISAggregate(eps, connections, aggregates, aggregate_root_nodes);
    }
    void AMGGreedyAggregate(ValueType eps, LocalVector<int>* connections, LocalVector<int>* aggregates,
                            LocalVector<int>* aggregate_root_nodes) const
    {
        this->m_interior.AMGGreedyAggregate(eps, connections, aggregates, aggregate_root_nodes);
    }
    void AMGUnsmoothedAggregation(const LocalVector<int>& aggregates, const LocalVector<int>& aggregate_root_nodes,
                                  GlobalMatrix<ValueType>* prolong) const
    {
        assert(prolong != NULL && prolong != this);
        this->m_interior.AMGUnsmoothedAggregation(aggregates, aggregate_root_nodes, &prolong->m_interior);
        prolong->doBlockDiagonal(this->pm_, this->GetM(), -1);
    }
    void AMGSmoothedAggregation(ValueType relax, const LocalVector<int>& connections, const LocalVector<int>& aggregates,
                                const LocalVector<int>& aggregate_root_nodes, GlobalMatrix<ValueType>* prolong,
                                int lumping_strat = 0) const
    {
        assert(prolong != NULL && prolong != this);
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
            (void)ramd_comm_halo_release(this->m_halo_comm, this->m_halo_plan);
        this->m_halo_plan = 0;
        this->m_halo_comm = NULL;
    }
    int                            m_halo_plan = 0;
    ramd_comm_t                    m_halo_comm = NULL;
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
