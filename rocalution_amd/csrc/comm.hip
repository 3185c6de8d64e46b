// comm.hip -- halo exchange and scalar all-reduce between the per-GPU processes of one node.
//
// Replaces the reference's MPI layer (src/utils/communicator.cpp) and the host-staged halo of
// GlobalMatrix::Apply (src/base/global_matrix.cpp:948-1008: pack -> D2H -> MPI_Isend/Irecv ->
// Waitall -> H2D).  MI355X-native form: the packed boundary values never leave the device; a grouped
// ncclSend/ncclRecv per neighbour runs on the GHOST stream over xGMI while the interior SpMV runs on
// the compute stream; events, not host syncs, order the two.  Scalars of a fused reduction are summed
// by ONE ncclAllReduce on the device scalar record.
#include "common.hpp"
#include "matrix_impl.hpp"

#include <rccl/rccl.h>

#include <vector>

using namespace ramd;

struct ramd_comm_s
{
    int  rank = 0, size = 1;
    bool use_rccl = false;
    ncclComm_t nccl = nullptr;
    ramd_exchange_cb  cb_exchange  = nullptr;
    ramd_allreduce_cb cb_allreduce = nullptr;
    void*             user         = nullptr;
    hipEvent_t ev_packed = nullptr, ev_halo = nullptr;
    // host staging (callback transport)
    void*  h_send = nullptr;
    void*  h_recv = nullptr;
    size_t h_send_bytes = 0, h_recv_bytes = 0;
};

#define RAMD_NCCL(expr)                                                                            \
    do                                                                                             \
    {                                                                                              \
        ncclResult_t r_ = (expr);                                                                  \
        if(r_ != ncclSuccess)                                                                      \
        {                                                                                          \
            ::ramd::set_error(__FILE__, __LINE__, std::string(#expr) + " -> " + ncclGetErrorString(r_)); \
            return RAMD_ERR_HIP;                                                                   \
        }                                                                                          \
    } while(0)

static int comm_common_init(ramd_comm_s* c)
{
    RAMD_HIP(hipEventCreateWithFlags(&c->ev_packed, hipEventDisableTiming));
    RAMD_HIP(hipEventCreateWithFlags(&c->ev_halo, hipEventDisableTiming));
    return RAMD_OK;
}

extern "C" {

int ramd_comm_unique_id(char id[128])
{
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId u;
    RAMD_NCCL(ncclGetUniqueId(&u));
    memcpy(id, &u, 128);
    return RAMD_OK;
}

int ramd_comm_init_rccl(int rank, int nranks, const char id[128], ramd_comm_t* out)
{
    RAMD_TRY(ensure_init());
    if(!out || rank < 0 || rank >= nranks)
        RAMD_FAIL(RAMD_ERR_ARG, "bad rank / size");
    ramd_comm_s* c = new ramd_comm_s;
    c->rank        = rank;
    c->size        = nranks;
    c->use_rccl    = true;
    ncclUniqueId u;
    memcpy(&u, id, 128);
    ncclResult_t r = ncclCommInitRank(&c->nccl, nranks, u, rank);
    if(r != ncclSuccess)
    {
        delete c;
        RAMD_FAIL(RAMD_ERR_HIP, std::string("ncclCommInitRank -> ") + ncclGetErrorString(r));
    }
    int s = comm_common_init(c);
    if(s != RAMD_OK)
    {
        delete c;
        return s;
    }
    *out = c;
    return RAMD_OK;
}

int ramd_comm_init_callback(int rank, int nranks, ramd_exchange_cb exchange, ramd_allreduce_cb allreduce,
                            void* user, ramd_comm_t* out)
{
    RAMD_TRY(ensure_init());
    if(!out || rank < 0 || rank >= nranks || !exchange || !allreduce)
        RAMD_FAIL(RAMD_ERR_ARG, "bad rank / size / callbacks");
    ramd_comm_s* c  = new ramd_comm_s;
    c->rank         = rank;
    c->size         = nranks;
    c->cb_exchange  = exchange;
    c->cb_allreduce = allreduce;
    c->user         = user;
    int s           = comm_common_init(c);
    if(s != RAMD_OK)
    {
        delete c;
        return s;
    }
    *out = c;
    return RAMD_OK;
}

int ramd_comm_destroy(ramd_comm_t c)
{
    if(!c)
        return RAMD_OK;
    if(c->nccl)
        (void)ncclCommDestroy(c->nccl);
    if(c->ev_packed)
        (void)hipEventDestroy(c->ev_packed);
    if(c->ev_halo)
        (void)hipEventDestroy(c->ev_halo);
    if(c->h_send)
        (void)hipHostFree(c->h_send);
    if(c->h_recv)
        (void)hipHostFree(c->h_recv);
    delete c;
    return RAMD_OK;
}

int ramd_comm_rank(ramd_comm_t c, int* rank)
{
    *rank = c ? c->rank : 0;
    return RAMD_OK;
}
int ramd_comm_size(ramd_comm_t c, int* size)
{
    *size = c ? c->size : 1;
    return RAMD_OK;
}

int ramd_comm_rccl_count(ramd_comm_t c, int* nranks)
{
    if(!nranks)
        RAMD_FAIL(RAMD_ERR_ARG, "null output");
    *nranks = 0;
    if(c && c->use_rccl)
        RAMD_NCCL(ncclCommCount(c->nccl, nranks));
    return RAMD_OK;
}

int ramd_comm_allreduce_scalars(ramd_comm_t c, int first, int count)
{
    // RAMD_COMM_FORCE_COLLECTIVES=1: do not skip the collective on a communicator of size 1 (plumbing check)
    static const bool force = [] { const char* e = getenv("RAMD_COMM_FORCE_COLLECTIVES"); return e && atoi(e) != 0; }();
    if(!c || (c->size == 1 && !force))
        return RAMD_OK;
    if(first < 0 || count < 1 || first + count > kScalarSlots)
        RAMD_FAIL(RAMD_ERR_ARG, "scalar range out of bounds");
    Backend& b = backend();
    if(c->use_rccl)
    {
        prof_begin(RAMD_PROF_ALLREDUCE, b.cur);
        RAMD_NCCL(ncclAllReduce(b.d_scalars + first, b.d_scalars + first, (size_t)count, ncclDouble,
                                ncclSum, c->nccl, b.cur));
        prof_end(RAMD_PROF_ALLREDUCE, b.cur);
        return RAMD_OK;
    }
    prof_count(RAMD_PROF_ALLREDUCE);
    double tmp[kScalarSlots];
    RAMD_HIP(hipMemcpyAsync(tmp, b.d_scalars + first, sizeof(double) * count, hipMemcpyDeviceToHost,
                            b.cur));
    RAMD_HIP(hipStreamSynchronize(b.cur));
    if(c->cb_allreduce(c->user, tmp, count) != 0)
        RAMD_FAIL(RAMD_ERR_STATE, "allreduce callback failed");
    RAMD_HIP(hipMemcpyAsync(b.d_scalars + first, tmp, sizeof(double) * count, hipMemcpyHostToDevice,
                            b.cur));
    RAMD_HIP(hipStreamSynchronize(b.cur));
    return RAMD_OK;
}

static int ensure_host(void** p, size_t* have, size_t need)
{
    if(*have >= need && *p)
        return RAMD_OK;
    if(*p)
        (void)hipHostFree(*p);
    *p = nullptr;
    RAMD_HIP(hipHostMalloc(p, need > 0 ? need : 16, hipHostMallocDefault));
    *have = need;
    return RAMD_OK;
}

int ramd_comm_halo_begin(ramd_comm_t c, ramd_vec_t send, ramd_vec_t recv, int npeers, const int* peers,
                         const int64_t* send_offset, const int64_t* recv_offset)
{
    if(!c || npeers <= 0)
        return RAMD_OK;
    if(!send || !recv || !peers || !send_offset || !recv_offset || send->dtype != recv->dtype)
        RAMD_FAIL(RAMD_ERR_ARG, "halo_begin: bad arguments");
    if(send_offset[npeers] > send->n || recv_offset[npeers] > recv->n)
        RAMD_FAIL(RAMD_ERR_ARG, "halo_begin: offsets exceed the buffer sizes");
    Backend&     b  = backend();
    const size_t es = (send->dtype == RAMD_F64) ? 8 : 4;
    // the exchange may start once the pack kernel (already queued on the current stream) is done
    RAMD_HIP(hipEventRecord(c->ev_packed, b.cur));
    RAMD_HIP(hipStreamWaitEvent(b.stream_ghost, c->ev_packed, 0));
    prof_begin(RAMD_PROF_HALO, b.stream_ghost);
    if(c->use_rccl)
    {
        const ncclDataType_t dt = (send->dtype == RAMD_F64) ? ncclDouble : ncclFloat;
        RAMD_NCCL(ncclGroupStart());
        for(int k = 0; k < npeers; ++k)
        {
            const int64_t ns = send_offset[k + 1] - send_offset[k];
            const int64_t nr = recv_offset[k + 1] - recv_offset[k];
            if(ns > 0)
                RAMD_NCCL(ncclSend((const char*)send->d + send_offset[k] * es, (size_t)ns, dt, peers[k],
                                   c->nccl, b.stream_ghost));
            if(nr > 0)
                RAMD_NCCL(ncclRecv((char*)recv->d + recv_offset[k] * es, (size_t)nr, dt, peers[k],
                                   c->nccl, b.stream_ghost));
        }
        RAMD_NCCL(ncclGroupEnd());
    }
    else
    {
        const size_t sb = (size_t)send_offset[npeers] * es, rb = (size_t)recv_offset[npeers] * es;
        RAMD_TRY(ensure_host(&c->h_send, &c->h_send_bytes, sb));
        RAMD_TRY(ensure_host(&c->h_recv, &c->h_recv_bytes, rb));
        if(sb > 0)
            RAMD_HIP(hipMemcpyAsync(c->h_send, send->d, sb, hipMemcpyDeviceToHost, b.stream_ghost));
        RAMD_HIP(hipStreamSynchronize(b.stream_ghost));
        std::vector<int64_t> so((size_t)npeers + 1), ro((size_t)npeers + 1);
        for(int k = 0; k <= npeers; ++k)
        {
            so[k] = send_offset[k] * (int64_t)es;
            ro[k] = recv_offset[k] * (int64_t)es;
        }
        if(c->cb_exchange(c->user, npeers, peers, c->h_send, so.data(), c->h_recv, ro.data()) != 0)
            RAMD_FAIL(RAMD_ERR_STATE, "halo exchange callback failed");
        if(rb > 0)
            RAMD_HIP(hipMemcpyAsync(recv->d, c->h_recv, rb, hipMemcpyHostToDevice, b.stream_ghost));
    }
    prof_end(RAMD_PROF_HALO, b.stream_ghost);
    RAMD_HIP(hipEventRecord(c->ev_halo, b.stream_ghost));
    return RAMD_OK;
}

int ramd_comm_halo_end(ramd_comm_t c)
{
    if(!c)
        return RAMD_OK;
    // the two events bracket only the wait: their distance is the part of the exchange the interior SpMV did not hide
    prof_begin(RAMD_PROF_HALO_WAIT, backend().cur);
    RAMD_HIP(hipStreamWaitEvent(backend().cur, c->ev_halo, 0));
    prof_end(RAMD_PROF_HALO_WAIT, backend().cur);
    return RAMD_OK;
}

} // extern "C"
