// rocalution/distribute.hpp -- row-block distribution of a replicated matrix: what the reference's sample drivers do with
// clients/include/common.hpp:56-431 (distribute_matrix) before a distributed solve.
//
// Every rank holds the whole matrix (it read the same file) and keeps a contiguous block of rows -- the first
// (global_nrow % ranks) ranks one row more (common.hpp:93-112) --, split into the INTERIOR part (columns it owns, renumbered
// from 0) and the GHOST part (columns of other ranks, renumbered to positions in the receive buffer), plus the
// communication pattern for the ParallelManager.  Because the matrix is replicated, the pattern needs no communication:
//   what rank q must send to rank r  = the rows of q that r's rows refer to, ascending
//   what rank r receives from rank q = the same list (r's ghost columns owned by q, ascending)
// Both are computed here from the global pattern; for the symmetric patterns the reference assumes this is exactly the
// boundary list it builds from a rank's own rows, and it stays consistent for unsymmetric patterns, where that shortcut
// is not.  Neighbours are the union of senders and receivers (a one-directional pair gets an empty segment).
#pragma once

#include "global.hpp"

#include <algorithm>
#include <vector>

namespace rocalution
{

// first row of every rank's block, [ranks + 1]
inline std::vector<int64_t> row_block_offsets(int64_t global_nrow, int ranks)
{
    std::vector<int64_t> off((size_t)ranks + 1, 0);
    for(int r = 0; r < ranks; ++r)
        off[(size_t)r + 1] = off[(size_t)r] + global_nrow / ranks + (r < global_nrow % ranks ? 1 : 0);
    return off;
}

template <typename ValueType>
struct RankPiece
{
    int64_t                local_nrow = 0;
    std::vector<PtrType>   int_rp, gst_rp; // interior / ghost CSR of the rank's rows
    std::vector<int>       int_col, gst_col; // interior: local column; ghost: position in the receive buffer
    std::vector<ValueType> int_val, gst_val;
    std::vector<int>       peers; // neighbour ranks, ascending
    std::vector<int>       recv_offset, send_offset; // [peers + 1]
    std::vector<int64_t>   recv_global; // global column behind every receive-buffer position
    std::vector<int>       boundary; // local rows to send, concatenated per peer
};

// the piece of rank `rank` out of `ranks`; pure host arithmetic on the global CSR arrays (sorted or not)
template <typename ValueType>
RankPiece<ValueType> partition_csr(int rank, int ranks, int64_t global_nrow, const PtrType* rp, const int* col, const ValueType* val)
{
    RankPiece<ValueType>       P;
    const std::vector<int64_t> off = row_block_offsets(global_nrow, ranks);
    const int64_t              lo = off[(size_t)rank], hi = off[(size_t)rank + 1];
    auto owner = [&](int64_t c) { return (int)(std::upper_bound(off.begin(), off.end(), c) - off.begin()) - 1; };
    P.local_nrow = hi - lo;
    // receive side: my ghost columns, ascending = grouped by owner
    std::vector<int64_t> ghost;
    for(int64_t i = lo; i < hi; ++i)
        for(PtrType j = rp[i]; j < rp[i + 1]; ++j)
            if(col[j] < lo || col[j] >= hi)
                ghost.push_back(col[j]);
    std::sort(ghost.begin(), ghost.end());
    ghost.erase(std::unique(ghost.begin(), ghost.end()), ghost.end());
    P.recv_global = ghost;
    // send side: my rows other ranks refer to, per rank
    std::vector<std::vector<int>> need((size_t)ranks);
    for(int q = 0; q < ranks; ++q)
    {
        if(q == rank)
            continue;
        std::vector<int>& v = need[(size_t)q];
        for(int64_t i = off[(size_t)q]; i < off[(size_t)q + 1]; ++i)
            for(PtrType j = rp[i]; j < rp[i + 1]; ++j)
                if(col[j] >= lo && col[j] < hi)
                    v.push_back((int)(col[j] - lo));
        std::sort(v.begin(), v.end());
        v.erase(std::unique(v.begin(), v.end()), v.end());
    }
    std::vector<int> recv_count((size_t)ranks, 0);
    for(size_t g = 0; g < ghost.size(); ++g)
        ++recv_count[(size_t)owner(ghost[g])];
    for(int q = 0; q < ranks; ++q)
        if(q != rank && (recv_count[(size_t)q] > 0 || !need[(size_t)q].empty()))
            P.peers.push_back(q);
    P.recv_offset.assign(1, 0);
    P.send_offset.assign(1, 0);
    for(size_t k = 0; k < P.peers.size(); ++k)
    {
        const int q = P.peers[k];
        P.recv_offset.push_back(P.recv_offset.back() + recv_count[(size_t)q]);
        P.send_offset.push_back(P.send_offset.back() + (int)need[(size_t)q].size());
        P.boundary.insert(P.boundary.end(), need[(size_t)q].begin(), need[(size_t)q].end());
    }
    // the two matrices, entries of a row in their original order
    P.int_rp.assign((size_t)P.local_nrow + 1, 0);
    P.gst_rp.assign((size_t)P.local_nrow + 1, 0);
    for(int64_t i = lo; i < hi; ++i)
    {
        for(PtrType j = rp[i]; j < rp[i + 1]; ++j)
        {
            if(col[j] >= lo && col[j] < hi)
            {
                P.int_col.push_back((int)(col[j] - lo));
                P.int_val.push_back(val[j]);
            }
            else
            {
                P.gst_col.push_back((int)(std::lower_bound(ghost.begin(), ghost.end(), (int64_t)col[j]) - ghost.begin()));
                P.gst_val.push_back(val[j]);
            }
        }
        P.int_rp[(size_t)(i - lo) + 1] = (PtrType)P.int_col.size();
        P.gst_rp[(size_t)(i - lo) + 1] = (PtrType)P.gst_col.size();
    }
    return P;
}

// clients/include/common.hpp:56-431: lmat (the replicated matrix; emptied, as the reference leaves it) -> gmat + pm
template <typename ValueType>
void distribute_matrix(const void* comm, LocalMatrix<ValueType>* lmat, GlobalMatrix<ValueType>* gmat, ParallelManager* pm)
{
    RAMD_EXPECT(comm != NULL && lmat != NULL && gmat != NULL && pm != NULL);
    const int64_t gnrow = lmat->GetM(), gncol = lmat->GetN(), gnnz = lmat->GetNnz();
    RAMD_EXPECT(gnrow == gncol);
    std::vector<PtrType>   rp((size_t)gnrow + 1);
    std::vector<int>       col((size_t)gnnz);
    std::vector<ValueType> val((size_t)gnnz);
    lmat->CopyToCSR(rp.data(), col.data(), val.data());
    lmat->Clear();
    pm->SetMPICommunicator(comm);
    const RankPiece<ValueType> P = partition_csr<ValueType>(pm->GetRank(), pm->GetNumProcs(), gnrow, rp.data(), col.data(), val.data());
    pm->SetGlobalNrow(gnrow);
    pm->SetGlobalNcol(gncol);
    pm->SetLocalNrow(P.local_nrow);
    pm->SetLocalNcol(P.local_nrow);
    pm->SetBoundaryIndex((int)P.boundary.size(), P.boundary.data());
    pm->SetReceivers((int)P.peers.size(), P.peers.data(), P.recv_offset.data());
    pm->SetSenders((int)P.peers.size(), P.peers.data(), P.send_offset.data());
    gmat->SetParallelManager(*pm);
    auto hand_over = [](const auto& v, auto** out) { // (SetDataPtr takes ownership of arrays from allocate_host)
        allocate_host((int64_t)(v.empty() ? 1 : v.size()), out);
        std::copy(v.begin(), v.end(), *out);
    };
    PtrType*   irp = NULL;
    int*       ic  = NULL;
    ValueType* iv  = NULL;
    hand_over(P.int_rp, &irp);
    hand_over(P.int_col, &ic);
    hand_over(P.int_val, &iv);
    gmat->SetLocalDataPtrCSR(&irp, &ic, &iv, "mat", (int64_t)P.int_col.size());
    if(pm->GetNumProcs() > 1)
    {
        PtrType*   grp = NULL;
        int*       gc  = NULL;
        ValueType* gv  = NULL;
        hand_over(P.gst_rp, &grp);
        hand_over(P.gst_col, &gc);
        hand_over(P.gst_val, &gv);
        gmat->SetGhostDataPtrCSR(&grp, &gc, &gv, "mat", (int64_t)P.gst_col.size());
    }
}

} // namespace rocalution
