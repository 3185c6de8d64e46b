// backend.hip -- lifecycle, streams, error state, scalar records.
// Replaces src/base/hip/backend_hip.cpp:50-487 of the reference (init/stop/info, the three
// streams and the compute_{default,interior,ghost} switches) -- no rocBLAS/rocSPARSE handles.
#include "common.hpp"

#include <map>
#include <mutex>
#include <unordered_map>

#include <mutex>

namespace ramd
{

static thread_local std::string g_err;
static Backend                  g_backend;

void set_error(const char* file, int line, const std::string& msg)
{
    const char* base = strrchr(file, '/');
    g_err = std::string(base ? base + 1 : file) + ":" + std::to_string(line) + ": " + msg;
}
const char* last_error()
{
    return g_err.c_str();
}
Backend& backend()
{
    return g_backend;
}

int ensure_init()
{
    if(g_backend.initialized)
        return RAMD_OK;
    return ramd_init(-1);
}

} // namespace ramd

using namespace ramd;

// ---------------------------------------------------------------- caching device allocator
// hipFree of multi-GB blocks is deferred by the runtime and paid by a LATER hipMalloc (measured: a solver
// Build() after the Clear() of a 512^3 preconditioner took +2.3..3.2 s at random).  Freed blocks are therefore
// kept and handed out again to requests of (nearly) the same size -- Build/Clear cycles repeat their sizes
// exactly.  RAMD_ALLOC_CACHE=0 disables the cache; it is emptied on out-of-memory and by ramd_stop().
namespace ramd
{
namespace
{
struct AllocCache
{
    std::mutex                       mu;
    std::multimap<size_t, void*>     free_blocks; // size -> block
    std::unordered_map<void*, size_t> live;       // block -> size
    size_t                           cached_bytes = 0;
    size_t                           cap          = 0; // bytes the cache may hold (set at first use)
    int                              enabled      = -1;
    bool on()
    {
        if(enabled < 0)
        {
            const char* e = getenv("RAMD_ALLOC_CACHE");
            enabled       = (e && atoi(e) == 0) ? 0 : 1;
        }
        return enabled == 1;
    }
    void drop_all()
    {
        for(auto& kv : free_blocks)
            (void)hipFree(kv.second);
        free_blocks.clear();
        cached_bytes = 0;
    }
};
AllocCache& cache()
{
    static AllocCache c;
    return c;
}
} // namespace

hipError_t cached_malloc_bytes(void** p, size_t bytes)
{
    AllocCache&                 c = cache();
    std::lock_guard<std::mutex> lk(c.mu);
    const size_t                need = (bytes + 255) & ~(size_t)255;
    if(c.on() && need >= (1u << 20)) // small blocks go straight to the runtime (its own pools are fine there)
    {
        auto it = c.free_blocks.lower_bound(need);
        if(it != c.free_blocks.end() && it->first <= need + need / 8)
        {
            *p = it->second;
            c.cached_bytes -= it->first;
            c.live[*p] = it->first;
            c.free_blocks.erase(it);
            return hipSuccess;
        }
    }
    hipError_t e = hipMalloc(p, need);
    if(e != hipSuccess && !c.free_blocks.empty())
    {
        (void)hipGetLastError();
        c.drop_all(); // out of memory: give the cached blocks back and try again
        e = hipMalloc(p, need);
    }
    if(e == hipSuccess)
        c.live[*p] = need;
    return e;
}

hipError_t cached_free(void* p)
{
    if(!p)
        return hipSuccess;
    AllocCache&                 c = cache();
    std::lock_guard<std::mutex> lk(c.mu);
    auto                        it = c.live.find(p);
    if(it == c.live.end())
        return hipFree(p); // not ours
    const size_t sz = it->second;
    c.live.erase(it);
    if(c.on() && sz >= (1u << 20))
    {
        if(c.cap == 0)
        {
            size_t f = 0, t = 0;
            c.cap    = (hipMemGetInfo(&f, &t) == hipSuccess) ? t / 5 * 2 : ((size_t)64 << 30); // 40 % of the device
        }
        if(c.cached_bytes + sz <= c.cap)
        {
            // kernels on any stream may still use the block: hipFree would have waited for the device, too
            (void)hipDeviceSynchronize();
            c.free_blocks.emplace(sz, p);
            c.cached_bytes += sz;
            return hipSuccess;
        }
    }
    return hipFree(p);
}

void cached_release_all(void)
{
    AllocCache&                 c = cache();
    std::lock_guard<std::mutex> lk(c.mu);
    c.drop_all();
}

} // namespace ramd

extern "C" {

const char* ramd_last_error(void)
{
    return last_error();
}
void ramd_set_last_error(const char* msg)
{
    set_error("host layer", 0, msg ? msg : "");
}

int ramd_device_count(int* count)
{
    int        c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if(e != hipSuccess)
        c = 0;
    if(count)
        *count = c;
    return RAMD_OK;
}

int ramd_init(int device)
{
    Backend& b = backend();
    if(b.initialized)
        return RAMD_OK;
    int        ndev = 0;
    hipError_t e    = hipGetDeviceCount(&ndev);
    if(e != hipSuccess || ndev <= 0)
        RAMD_FAIL(RAMD_ERR_NO_DEVICE,
                  "no HIP device available: this backend has no host compute path (the reference "
                  "would fall back to its OpenMP backend; use the reference/oracle for that)");
    if(device >= 0)
        RAMD_HIP(hipSetDevice(device % ndev));
    RAMD_HIP(hipGetDevice(&b.device));
    hipDeviceProp_t prop;
    RAMD_HIP(hipGetDeviceProperties(&prop, b.device));
    b.num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    snprintf(b.arch, sizeof(b.arch), "%s", prop.gcnArchName);
    if(char* colon = strchr(b.arch, ':'))
        *colon = 0;
    // default stream = NULL stream as in the reference (backend_hip.cpp:80); interior / ghost
    // are non-blocking side streams used by the GlobalMatrix choreography.
    b.stream_default = nullptr;
    RAMD_HIP(hipStreamCreateWithFlags(&b.stream_interior, hipStreamNonBlocking));
    RAMD_HIP(hipStreamCreateWithFlags(&b.stream_ghost, hipStreamNonBlocking));
    b.cur = b.stream_default;
    RAMD_HIP(hipMalloc((void**)&b.d_partials, sizeof(double) * kScalarSlots * kReduceBlocks));
    RAMD_HIP(hipMalloc((void**)&b.d_ticket, sizeof(unsigned int) * 64));
    RAMD_HIP(hipMemset(b.d_ticket, 0, sizeof(unsigned int) * 64));
    RAMD_HIP(hipMalloc((void**)&b.d_scalars, sizeof(double) * kScalarSlots));
    RAMD_HIP(hipMemset(b.d_scalars, 0, sizeof(double) * kScalarSlots));
    RAMD_HIP(hipHostMalloc((void**)&b.h_scalars, sizeof(double) * kScalarSlots * kScalarRecords,
                           hipHostMallocDefault));
    for(int i = 0; i < kScalarRecords; ++i)
        RAMD_HIP(hipEventCreateWithFlags(&b.ev_scalar[i], hipEventDisableTiming));
    b.initialized = true;
    return RAMD_OK;
}

int ramd_stop(void)
{
    Backend& b = backend();
    if(!b.initialized)
        return RAMD_OK;
    (void)hipDeviceSynchronize();
    cached_release_all();
    (void)hipFree(b.d_partials);
    (void)hipFree(b.d_ticket);
    (void)hipFree(b.d_scalars);
    (void)hipHostFree(b.h_scalars);
    for(int i = 0; i < kScalarRecords; ++i)
        (void)hipEventDestroy(b.ev_scalar[i]);
    (void)hipStreamDestroy(b.stream_interior);
    (void)hipStreamDestroy(b.stream_ghost);
    b = Backend();
    return RAMD_OK;
}

int ramd_is_initialized(void)
{
    return backend().initialized ? 1 : 0;
}

const char* ramd_get_arch(void)
{
    return backend().arch;
}

int ramd_info(char* buf, int buflen)
{
    Backend& b = backend();
    if(!b.initialized)
    {
        snprintf(buf, buflen, "rocalution_amd: backend not initialized");
        return RAMD_OK;
    }
    hipDeviceProp_t prop;
    RAMD_HIP(hipGetDeviceProperties(&prop, b.device));
    snprintf(buf, buflen,
             "rocalution_amd MI355X-native backend: device %d '%s' arch %s, %d CUs, %.1f GiB HBM, "
             "wavefront %d, hand-written HIP kernels (no rocSPARSE/rocBLAS)",
             b.device, prop.name, b.arch, b.num_cu, prop.totalGlobalMem / 1073741824.0,
             prop.warpSize);
    return RAMD_OK;
}

int ramd_sync(void)
{
    RAMD_HIP(hipDeviceSynchronize());
    return RAMD_OK;
}
int ramd_sync_default(void)
{
    RAMD_HIP(hipStreamSynchronize(backend().stream_default));
    return RAMD_OK;
}
int ramd_sync_interior(void)
{
    RAMD_HIP(hipStreamSynchronize(backend().stream_interior));
    return RAMD_OK;
}
int ramd_sync_ghost(void)
{
    RAMD_HIP(hipStreamSynchronize(backend().stream_ghost));
    return RAMD_OK;
}
int ramd_compute_default(void)
{
    backend().cur = backend().stream_default;
    return RAMD_OK;
}
int ramd_compute_interior(void)
{
    backend().cur = backend().stream_interior;
    return RAMD_OK;
}
int ramd_compute_ghost(void)
{
    backend().cur = backend().stream_ghost;
    return RAMD_OK;
}
void* ramd_current_stream(void)
{
    return (void*)backend().cur;
}

int ramd_alloc_pinned(void** ptr, int64_t bytes)
{
    RAMD_HIP(hipHostMalloc(ptr, (size_t)bytes, hipHostMallocDefault));
    return RAMD_OK;
}
int ramd_free_pinned(void* ptr)
{
    RAMD_HIP(hipHostFree(ptr));
    return RAMD_OK;
}

// ---------------------------------------------------------------- measurement hooks
static hipEvent_t g_t0 = nullptr, g_t1 = nullptr;

int ramd_mem_info(uint64_t* free_bytes, uint64_t* total_bytes)
{
    RAMD_TRY(ensure_init());
    size_t f = 0, t = 0;
    RAMD_HIP(hipMemGetInfo(&f, &t));
    if(free_bytes)
        *free_bytes = (uint64_t)f;
    if(total_bytes)
        *total_bytes = (uint64_t)t;
    return RAMD_OK;
}

int ramd_timer_start(void)
{
    RAMD_TRY(ensure_init());
    if(!g_t0)
    {
        RAMD_HIP(hipEventCreate(&g_t0));
        RAMD_HIP(hipEventCreate(&g_t1));
    }
    RAMD_HIP(hipEventRecord(g_t0, backend().cur));
    return RAMD_OK;
}
int ramd_timer_stop(double* elapsed_ms)
{
    if(!g_t0)
        RAMD_FAIL(RAMD_ERR_STATE, "timer_stop without timer_start");
    RAMD_HIP(hipEventRecord(g_t1, backend().cur));
    RAMD_HIP(hipEventSynchronize(g_t1));
    float ms = 0.f;
    RAMD_HIP(hipEventElapsedTime(&ms, g_t0, g_t1));
    if(elapsed_ms)
        *elapsed_ms = (double)ms;
    return RAMD_OK;
}

} // extern "C"

namespace ramd
{
// HIP-event brackets around the launches of one kind (bench.py roofline / scaling legs): a ring of event pairs per
// channel, recorded on the stream the launch goes to.  Off by default: no events, no cost.
constexpr int kProfRing = 8192;
struct ProfChan
{
    bool        on   = false;
    bool        init = false;
    int         n    = 0;
    hipEvent_t* ev   = nullptr; // 2 * kProfRing
};
static ProfChan g_prof[RAMD_PROF_NCHAN];
static int64_t  g_prof_count[RAMD_PROF_NCHAN] = {0};

void prof_begin(int ch, hipStream_t s)
{
    ProfChan& c = g_prof[ch];
    if(c.on && c.n < kProfRing)
        (void)hipEventRecord(c.ev[2 * c.n], s ? s : backend().cur);
}
void prof_end(int ch, hipStream_t s)
{
    ProfChan& c = g_prof[ch];
    ++g_prof_count[ch];
    if(c.on && c.n < kProfRing)
    {
        (void)hipEventRecord(c.ev[2 * c.n + 1], s ? s : backend().cur);
        ++c.n;
    }
}
void prof_count(int ch)
{
    ++g_prof_count[ch];
}
void prof_spmv_begin()
{
    prof_begin(RAMD_PROF_SPMV, nullptr);
}
void prof_spmv_end()
{
    prof_end(RAMD_PROF_SPMV, nullptr);
}
} // namespace ramd

extern "C" {

int ramd_prof_enable(int channel, int on)
{
    RAMD_TRY(ensure_init());
    if(channel < 0 || channel >= RAMD_PROF_NCHAN)
        RAMD_FAIL(RAMD_ERR_ARG, "profiling channel out of range");
    ProfChan& c = g_prof[channel];
    if(on && !c.init)
    {
        c.ev = new hipEvent_t[2 * kProfRing];
        for(int i = 0; i < 2 * kProfRing; ++i)
            RAMD_HIP(hipEventCreate(&c.ev[i]));
        c.init = true;
    }
    c.on = on != 0;
    if(on)
    {
        c.n                   = 0;
        g_prof_count[channel] = 0;
    }
    return RAMD_OK;
}
int ramd_prof_result(int channel, int* launches, double* avg_ms, double* min_ms, double* max_ms)
{
    if(channel < 0 || channel >= RAMD_PROF_NCHAN)
        RAMD_FAIL(RAMD_ERR_ARG, "profiling channel out of range");
    RAMD_HIP(hipDeviceSynchronize());
    ProfChan& c   = g_prof[channel];
    double    sum = 0.0, mn = 1e30, mx = 0.0;
    for(int i = 0; i < c.n; ++i)
    {
        float ms = 0.f;
        RAMD_HIP(hipEventElapsedTime(&ms, c.ev[2 * i], c.ev[2 * i + 1]));
        sum += ms;
        mn = ms < mn ? ms : mn;
        mx = ms > mx ? ms : mx;
    }
    if(launches)
        *launches = c.n;
    if(avg_ms)
        *avg_ms = c.n ? sum / c.n : 0.0;
    if(min_ms)
        *min_ms = c.n ? mn : 0.0;
    if(max_ms)
        *max_ms = mx;
    return RAMD_OK;
}
int ramd_prof_count(int channel, int64_t* count)
{
    if(channel < 0 || channel >= RAMD_PROF_NCHAN || !count)
        RAMD_FAIL(RAMD_ERR_ARG, "profiling channel out of range");
    *count = g_prof_count[channel];
    return RAMD_OK;
}
int ramd_prof_spmv_enable(int on)
{
    return ramd_prof_enable(RAMD_PROF_SPMV, on);
}
int ramd_prof_spmv_result(int* launches, double* avg_ms, double* min_ms, double* max_ms)
{
    return ramd_prof_result(RAMD_PROF_SPMV, launches, avg_ms, min_ms, max_ms);
}

// ---------------------------------------------------------------- scalar records
int ramd_scalars_set(int slot, double value)
{
    RAMD_TRY(ensure_init());
    if(slot < 0 || slot >= kScalarSlots)
        RAMD_FAIL(RAMD_ERR_ARG, "scalar slot out of range");
    Backend& b = backend();
    // pageable source is staged by the runtime before the call returns
    RAMD_HIP(hipMemcpyAsync(b.d_scalars + slot, &value, sizeof(double), hipMemcpyHostToDevice, b.cur));
    return RAMD_OK;
}

int ramd_scalars_fetch(double* host, int first, int count)
{
    RAMD_TRY(ensure_init());
    if(first < 0 || count < 0 || first + count > kScalarSlots)
        RAMD_FAIL(RAMD_ERR_ARG, "scalar range out of bounds");
    Backend& b = backend();
    RAMD_HIP(hipMemcpyAsync(b.h_scalars, b.d_scalars + first, sizeof(double) * count,
                            hipMemcpyDeviceToHost, b.cur));
    RAMD_HIP(hipStreamSynchronize(b.cur));
    memcpy(host, b.h_scalars, sizeof(double) * count);
    return RAMD_OK;
}

int ramd_scalars_fetch_async_begin(int record, int first, int count)
{
    RAMD_TRY(ensure_init());
    if(record < 0 || record >= kScalarRecords || first < 0 || count < 0
       || first + count > kScalarSlots)
        RAMD_FAIL(RAMD_ERR_ARG, "scalar record/range out of bounds");
    Backend& b = backend();
    RAMD_HIP(hipMemcpyAsync(b.h_scalars + (size_t)record * kScalarSlots, b.d_scalars + first,
                            sizeof(double) * count, hipMemcpyDeviceToHost, b.cur));
    RAMD_HIP(hipEventRecord(b.ev_scalar[record], b.cur));
    return RAMD_OK;
}

int ramd_scalars_fetch_async_end(int record, double* host, int count)
{
    Backend& b = backend();
    if(record < 0 || record >= kScalarRecords || count < 0 || count > kScalarSlots)
        RAMD_FAIL(RAMD_ERR_ARG, "scalar record out of bounds");
    RAMD_HIP(hipEventSynchronize(b.ev_scalar[record]));
    memcpy(host, b.h_scalars + (size_t)record * kScalarSlots, sizeof(double) * count);
    return RAMD_OK;
}

} // extern "C"
