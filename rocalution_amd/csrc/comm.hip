// comm.hip -- halo exchange and scalar all-reduce between the per-GPU processes of one node.
//
// Replaces the reference's MPI layer (src/utils/communicator.cpp) and the host-staged halo of
// GlobalMatrix::Apply (src/base/global_matrix.cpp:948-1008: pack -> D2H -> MPI_Isend/Irecv ->
// Waitall -> H2D).  MI355X-native form: the packed boundary values never leave the device; a grouped
// ncclSend/ncclRecv per neighbour runs on the GHOST stream over xGMI while the interior SpMV runs on
// the compute stream; events, not host syncs, order the two.  Scalars of a fused reduction are summed
// by ONE ncclAllReduce on the device scalar record.
//
// Many-neighbour graphs (last row): a rank with more than kAgPeers peers exchanges its halo with ONE
// ncclAllGather of equally padded boundary buffers instead of a group of send/recv pairs -- on the point-to-point xGMI
// fabric a ring all-gather moves every block once over every link, while P-1 pairs per rank serialise on the rank's links.
// What a rank needs from the gathered buffer is copied out by an index list built once per exchange plan from a small
// table every rank contributes (its total and, per destination rank, where that rank's piece starts in its buffer).
// RAMD_COMM_HALO=allgather / sendrecv forces either form.
#include "common.hpp"
#include "matrix_impl.hpp"

#include <rccl/rccl.h>

#include <mutex>
#include <set>
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
    // all-gather form of the halo exchange: one plan per ramd_comm_halo_select call that chose it.  A plan is identified by
    // the sequence number of that (collective) call -- the same number on every rank -- never by a rank's own (peers,
    // offsets): the padded length M and the table behind d_idx are properties of ALL ranks' plans, and two matrices whose
    // plans coincide on one rank (e.g. a rank without neighbours on every AMG level) need not coincide on the others.
    struct AgPlan
    {
        int     id = 0; // sequence number of the halo_select call that made it (> 0)
        int     npeers = 0; // this rank's side of the plan, checked at every exchange
        int64_t nsend = 0;
        int64_t M = 0; // padded boundary length (elements): the maximum over the ranks
        int64_t nrecv = 0;
        int*    d_idx = nullptr; // [nrecv] position in the gathered buffer of every received value
        void*   d_send = nullptr; // [M] padded copy of the packed boundary
        void*   d_all  = nullptr; // [size * M] the gathered buffers
    };
    std::vector<AgPlan> ag;
    long long           generation = 0; // unique per communicator of this process (a later one may reuse this one's address)
    int                 select_calls = 0; // ramd_comm_halo_select calls so far (collective: equal on every rank)
};

template <typename T>
__global__ __launch_bounds__(256) void k_halo_pick(int64_t n, const int* __restrict__ idx, const T* __restrict__ all,
                                                   T* __restrict__ recv)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n)
        recv[i] = all[idx[i]];
}

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

// communicators that have not been destroyed yet (a halo plan's owner may ask to release it after the communicator went)
static std::set<const ramd_comm_s*>& live_comms()
{
    static std::set<const ramd_comm_s*> s;
    return s;
}
static std::mutex& live_comms_mutex()
{
    static std::mutex m;
    return m;
}

static int comm_common_init(ramd_comm_s* c)
{
    RAMD_HIP(hipEventCreateWithFlags(&c->ev_packed, hipEventDisableTiming));
    RAMD_HIP(hipEventCreateWithFlags(&c->ev_halo, hipEventDisableTiming));
    static long long next_generation = 0;
    std::lock_guard<std::mutex> lock(live_comms_mutex());
    c->generation = ++next_generation;
    live_comms().insert(c);
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
    {
        std::lock_guard<std::mutex> lock(live_comms_mutex());
        live_comms().erase(c);
    }
    for(auto& pl : c->ag)
    {
        dev_free(&pl.d_idx);
        if(pl.d_send)
            (void)cached_free(pl.d_send);
        if(pl.d_all)
            (void)cached_free(pl.d_all);
    }
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

static int comm_allgather_host(ramd_comm_s* c, const int64_t* mine, int count, int64_t* out);
// every rank contributes `count` 64-bit integers; out[q * count ..] = rank q's (host arrays).  The setup exchanges of the
// distributed AMG (row / aggregate offsets in rank order, message lengths between any two ranks) use it: any number of ranks,
// one collective.
int ramd_comm_allgather_i64(ramd_comm_t c, const int64_t* mine, int count, int64_t* out)
{
    if(!c || !mine || !out || count < 1)
        RAMD_FAIL(RAMD_ERR_ARG, "ramd_comm_allgather_i64: bad arguments");
    if(c->size == 1)
    {
        for(int k = 0; k < count; ++k)
            out[k] = mine[k];
        return RAMD_OK;
    }
    return comm_allgather_host(c, mine, count, out);
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

} // extern "C"

constexpr int kAgPeers = 4; // more peers than this: all-gather form

static int halo_forced_form()
{
    static const int mode = [] {
        const char* e = getenv("RAMD_COMM_HALO");
        if(e && std::string(e) == "allgather")
            return 1;
        if(e && std::string(e) == "sendrecv")
            return 0;
        return -1;
    }();
    return mode;
}
// all ranks contribute `count` int64 each; out[r * count ..] = rank r's contribution (host arrays)
static int comm_allgather_host(ramd_comm_s* c, const int64_t* mine, int count, int64_t* out)
{
    Backend& b = backend();
    if(c->use_rccl)
    {
        int64_t *d_in = nullptr, *d_out = nullptr;
        RAMD_TRY(dev_alloc(&d_in, count));
        int s = dev_alloc(&d_out, (int64_t)count * c->size);
        if(s != RAMD_OK)
        {
            dev_free(&d_in);
            return s;
        }
        hipError_t   e = hipMemcpyAsync(d_in, mine, sizeof(int64_t) * (size_t)count, hipMemcpyHostToDevice, b.stream_ghost);
        ncclResult_t r = ncclSuccess;
        if(e == hipSuccess)
            r = ncclAllGather(d_in, d_out, (size_t)count, ncclInt64, c->nccl, b.stream_ghost);
        if(e == hipSuccess && r == ncclSuccess)
            e = hipMemcpyAsync(out, d_out, sizeof(int64_t) * (size_t)count * c->size, hipMemcpyDeviceToHost, b.stream_ghost);
        if(e == hipSuccess && r == ncclSuccess)
            e = hipStreamSynchronize(b.stream_ghost);
        dev_free(&d_in);
        dev_free(&d_out);
        if(r != ncclSuccess)
            RAMD_FAIL(RAMD_ERR_HIP, std::string("ncclAllGather (halo plan) -> ") + ncclGetErrorString(r));
        RAMD_HIP(e);
        return RAMD_OK;
    }
    // callback transport: everybody sends its record to everybody else
    std::vector<int>     peers;
    std::vector<int64_t> so, ro;
    std::vector<int64_t> sbuf, rbuf((size_t)count * (c->size - 1));
    so.push_back(0);
    ro.push_back(0);
    for(int q = 0; q < c->size; ++q)
        if(q != c->rank)
        {
            peers.push_back(q);
            sbuf.insert(sbuf.end(), mine, mine + count);
            so.push_back(so.back() + (int64_t)sizeof(int64_t) * count);
            ro.push_back(ro.back() + (int64_t)sizeof(int64_t) * count);
        }
    if(c->cb_exchange(c->user, (int)peers.size(), peers.data(), sbuf.data(), so.data(), rbuf.data(), ro.data()) != 0)
        RAMD_FAIL(RAMD_ERR_STATE, "halo plan exchange callback failed");
    int k = 0;
    for(int q = 0; q < c->size; ++q)
    {
        const int64_t* src = (q == c->rank) ? mine : rbuf.data() + (size_t)count * (k++);
        memcpy(out + (size_t)count * q, src, sizeof(int64_t) * (size_t)count);
    }
    return RAMD_OK;
}

// builds the all-gather plan `id` (a collective: every rank is inside ramd_comm_halo_select for the same plan)
static int halo_ag_plan(ramd_comm_s* c, int id, int npeers, const int* peers, const int64_t* send_offset,
                        const int64_t* recv_offset)
{
    // every rank's table: [0] = its boundary length, [1 + 2q] / [2 + 2q] = start / length of the piece for rank q
    const int            P = c->size, cnt = 1 + 2 * P;
    std::vector<int64_t> mine((size_t)cnt, 0), all((size_t)cnt * P, 0);
    mine[0] = npeers > 0 ? send_offset[npeers] : 0;
    for(int k = 0; k < npeers; ++k)
    {
        if(peers[k] < 0 || peers[k] >= P)
            RAMD_FAIL(RAMD_ERR_ARG, "halo plan: peer rank out of range");
        mine[1 + 2 * peers[k]] = send_offset[k];
        mine[2 + 2 * peers[k]] = send_offset[k + 1] - send_offset[k];
    }
    RAMD_TRY(comm_allgather_host(c, mine.data(), cnt, all.data()));
    ramd_comm_s::AgPlan pl;
    pl.id     = id;
    pl.npeers = npeers;
    pl.nsend  = npeers > 0 ? send_offset[npeers] : 0;
    pl.nrecv  = npeers > 0 ? recv_offset[npeers] : 0;
    for(int q = 0; q < P; ++q)
        pl.M = std::max(pl.M, all[(size_t)cnt * q]);
    pl.M = (pl.M + 1) & ~(int64_t)1; // (16-byte multiples in fp64)
    if(pl.M * P >= (1ll << 31))
        RAMD_FAIL(RAMD_ERR_UNSUPPORTED, "halo all-gather: gathered buffer beyond 32-bit indices");
    std::vector<int> idx((size_t)pl.nrecv, 0);
    for(int k = 0; k < npeers; ++k)
    {
        const int      q   = peers[k];
        const int64_t  off = all[(size_t)cnt * q + 1 + 2 * c->rank], len = all[(size_t)cnt * q + 2 + 2 * c->rank];
        if(len != recv_offset[k + 1] - recv_offset[k])
            RAMD_FAIL(RAMD_ERR_STATE, "halo plan: a peer sends a different number of values than this rank expects");
        for(int64_t j = 0; j < len; ++j)
            idx[(size_t)(recv_offset[k] + j)] = (int)((int64_t)q * pl.M + off + j);
    }
    if(getenv("RAMD_COMM_DEBUG"))
    {
        fprintf(stderr, "[rank %d] halo all-gather plan %d: npeers=%d M=%lld nrecv=%lld table:", c->rank, id, npeers, (long long)pl.M,
                (long long)pl.nrecv);
        for(size_t i = 0; i < all.size(); ++i)
            fprintf(stderr, " %lld", (long long)all[i]);
        fprintf(stderr, " | idx:");
        for(size_t i = 0; i < idx.size() && i < 6; ++i)
            fprintf(stderr, " %d", idx[i]);
        fprintf(stderr, "\n");
    }
    RAMD_TRY(dev_alloc(&pl.d_idx, pl.nrecv > 0 ? pl.nrecv : 1));
    if(pl.nrecv > 0)
        RAMD_HIP(hipMemcpy(pl.d_idx, idx.data(), sizeof(int) * idx.size(), hipMemcpyHostToDevice));
    const size_t es = 8; // (buffers sized for fp64; fp32 exchanges of the same plan use their first half)
    RAMD_HIP(cached_malloc_bytes(&pl.d_send, (size_t)(pl.M > 0 ? pl.M : 2) * es));
    RAMD_HIP(cached_malloc_bytes(&pl.d_all, (size_t)(pl.M > 0 ? pl.M : 2) * es * P));
    // (on the ghost stream, where the exchanges fill and read it: a null-stream memset is not ordered against that stream
    //  and could land after the first packed boundary was copied in)
    RAMD_HIP(hipMemsetAsync(pl.d_send, 0, (size_t)(pl.M > 0 ? pl.M : 2) * es, backend().stream_ghost));
    RAMD_HIP(hipStreamSynchronize(backend().stream_ghost));
    c->ag.push_back(pl);
    return RAMD_OK;
}

extern "C" {

int ramd_comm_halo_select(ramd_comm_t c, int npeers, const int* peers, const int64_t* send_offset,
                          const int64_t* recv_offset, int* allgather)
{
    if(allgather)
        *allgather = 0;
    if(!c || c->size < 2)
        return RAMD_OK;
    if(npeers < 0 || (npeers > 0 && (!peers || !send_offset || !recv_offset)))
        RAMD_FAIL(RAMD_ERR_ARG, "halo_select: bad arguments");
    // every rank learns the largest peer count: the rule has to give the same answer everywhere
    std::vector<int64_t> all((size_t)c->size, 0);
    const int64_t        mine = npeers;
    RAMD_TRY(comm_allgather_host(c, &mine, 1, all.data()));
    int64_t most = 0;
    for(int64_t v : all)
        most = std::max(most, v);
    const int forced = halo_forced_form();
    const int ag     = forced >= 0 ? forced : (most > kAgPeers ? 1 : 0);
    // the plan's identity: the number of this collective call (every rank counts the same calls in the same order)
    const int id = ++c->select_calls;
    if(ag)
        RAMD_TRY(halo_ag_plan(c, id, npeers, peers, send_offset, recv_offset));
    if(allgather)
        *allgather = ag ? id : 0;
    return RAMD_OK;
}

int ramd_comm_generation(ramd_comm_t c, long long* generation)
{
    if(!c || !generation)
        RAMD_FAIL(RAMD_ERR_ARG, "ramd_comm_generation: bad arguments");
    *generation = c->generation;
    return RAMD_OK;
}

int ramd_comm_halo_release(ramd_comm_t c, int plan, long long generation)
{
    // (the owner of a plan may outlive the communicator: a destroyed one has already given everything back)
    // (the owner of a plan may outlive its communicator, and a later communicator may sit at the same address: the plan is
    //  released only in the communicator it was announced in)
    {
        std::lock_guard<std::mutex> lock(live_comms_mutex());
        if(!c || plan <= 0 || !live_comms().count(c) || c->generation != generation)
            return RAMD_OK;
    }
    for(size_t i = 0; i < c->ag.size(); ++i)
        if(c->ag[i].id == plan)
        {
            // (an exchange of this plan may still be queued on the ghost stream)
            (void)hipStreamSynchronize(backend().stream_ghost);
            dev_free(&c->ag[i].d_idx);
            if(c->ag[i].d_send)
                (void)cached_free(c->ag[i].d_send);
            if(c->ag[i].d_all)
                (void)cached_free(c->ag[i].d_all);
            c->ag.erase(c->ag.begin() + (long)i);
            break;
        }
    return RAMD_OK;
}

int ramd_comm_halo_begin(ramd_comm_t c, ramd_vec_t send, ramd_vec_t recv, int npeers, const int* peers,
                         const int64_t* send_offset, const int64_t* recv_offset)
{
    return ramd_comm_halo_begin_plan(c, 0, send, recv, npeers, peers, send_offset, recv_offset);
}

int ramd_comm_halo_begin_plan(ramd_comm_t c, int plan, ramd_vec_t send, ramd_vec_t recv, int npeers, const int* peers,
                              const int64_t* send_offset, const int64_t* recv_offset)
{
    if(!c)
        return RAMD_OK;
    const bool ag_form = c->size > 1 && npeers >= 0 && plan > 0;
    if(npeers <= 0 && !ag_form)
        return RAMD_OK;
    if(!send || !recv || (npeers > 0 && (!peers || !send_offset || !recv_offset)) || send->dtype != recv->dtype)
        RAMD_FAIL(RAMD_ERR_ARG, "halo_begin: bad arguments");
    if(npeers > 0 && (send_offset[npeers] > send->n || recv_offset[npeers] > recv->n))
        RAMD_FAIL(RAMD_ERR_ARG, "halo_begin: offsets exceed the buffer sizes");
    Backend&     b  = backend();
    const size_t es = (send->dtype == RAMD_F64) ? 8 : 4;
    // the exchange may start once the pack kernel (already queued on the current stream) is done
    RAMD_HIP(hipEventRecord(c->ev_packed, b.cur));
    RAMD_HIP(hipStreamWaitEvent(b.stream_ghost, c->ev_packed, 0));
    if(ag_form)
    {
        // (a collective: every rank of the communicator is here for this exchange, also one without neighbours -- the
        //  form was agreed on when the plan was announced, ramd_comm_halo_select)
        ramd_comm_s::AgPlan* pl = nullptr;
        for(auto& q : c->ag)
            if(q.id == plan)
                pl = &q;
        if(!pl)
            RAMD_FAIL(RAMD_ERR_STATE, "halo_begin: unknown all-gather plan (ramd_comm_halo_select hands out the number)");
        if(pl->npeers != npeers || pl->nsend != (npeers > 0 ? send_offset[npeers] : 0)
           || pl->nrecv != (npeers > 0 ? recv_offset[npeers] : 0))
            RAMD_FAIL(RAMD_ERR_ARG, "halo_begin: peers / offsets differ from the plan announced under this number");
        prof_begin(RAMD_PROF_HALO, b.stream_ghost);
        const size_t sb = npeers > 0 ? (size_t)send_offset[npeers] * es : 0;
        if(sb > 0)
            RAMD_HIP(hipMemcpyAsync(pl->d_send, send->d, sb, hipMemcpyDeviceToDevice, b.stream_ghost));
        if(c->use_rccl)
        {
            const ncclDataType_t dt = (send->dtype == RAMD_F64) ? ncclDouble : ncclFloat;
            RAMD_NCCL(ncclAllGather(pl->d_send, pl->d_all, (size_t)pl->M, dt, c->nccl, b.stream_ghost));
        }
        else
        {
            // callback transport: the padded buffer goes to every other rank (host staged)
            const int    P  = c->size;
            const size_t mb = (size_t)pl->M * es;
            RAMD_TRY(ensure_host(&c->h_send, &c->h_send_bytes, mb * (size_t)(P - 1)));
            RAMD_TRY(ensure_host(&c->h_recv, &c->h_recv_bytes, mb * (size_t)(P - 1)));
            std::vector<int>     pr;
            std::vector<int64_t> so(1, 0), ro(1, 0);
            for(int q = 0; q < P; ++q)
                if(q != c->rank)
                {
                    RAMD_HIP(hipMemcpyAsync((char*)c->h_send + mb * pr.size(), pl->d_send, mb, hipMemcpyDeviceToHost,
                                            b.stream_ghost));
                    pr.push_back(q);
                    so.push_back(so.back() + (int64_t)mb);
                    ro.push_back(ro.back() + (int64_t)mb);
                }
            RAMD_HIP(hipStreamSynchronize(b.stream_ghost));
            if(c->cb_exchange(c->user, (int)pr.size(), pr.data(), c->h_send, so.data(), c->h_recv, ro.data()) != 0)
                RAMD_FAIL(RAMD_ERR_STATE, "halo exchange callback failed");
            size_t k = 0;
            for(int q = 0; q < P; ++q)
                if(q != c->rank)
                    RAMD_HIP(hipMemcpyAsync((char*)pl->d_all + mb * (size_t)q, (char*)c->h_recv + mb * (k++), mb,
                                            hipMemcpyHostToDevice, b.stream_ghost));
        }
        if(pl->nrecv > 0)
        {
            const unsigned g = (unsigned)((pl->nrecv + 255) / 256);
            if(send->dtype == RAMD_F64)
                hipLaunchKernelGGL((k_halo_pick<double>), dim3(g), dim3(256), 0, b.stream_ghost, pl->nrecv, pl->d_idx,
                                   (const double*)pl->d_all, (double*)recv->d);
            else
                hipLaunchKernelGGL((k_halo_pick<float>), dim3(g), dim3(256), 0, b.stream_ghost, pl->nrecv, pl->d_idx,
                                   (const float*)pl->d_all, (float*)recv->d);
        }
        prof_end(RAMD_PROF_HALO, b.stream_ghost);
        RAMD_HIP(hipEventRecord(c->ev_halo, b.stream_ghost));
        return RAMD_OK;
    }
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
