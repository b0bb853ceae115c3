// hip_stub.cpp -- TEST INFRASTRUCTURE: the 31 HIP runtime entry points liblogreg_hip.so's HOST code uses (nm -u of the library's
// objects), implemented on the host heap, so that the host engine -- lr_api.hip, lr_engine.h, lr_model.h: handles, workspaces,
// per-stream side slots, fork / join events of two-part plans, argument packing, every error path -- runs under AddressSanitizer,
// UBSan and ThreadSanitizer in the GPU-less container (tests/test_engine_sanitizers.py; VERDICT r5 item 3, SURVEY section 5
// "sanitizers").  "Device" memory is malloc'ed (so an out-of-bounds copy, a use after free, a double free or a leak of a device
// buffer is an ASan / LSan finding), streams and events are heap objects, a kernel launch validates its configuration and its
// stream and does nothing else.  The stub also keeps books the harness reads through hipstub_*: launches per stream, waits on
// events that were never recorded, objects alive -- and can make the N-th hipMalloc fail (cleanup paths of lr_model_create).
// Nothing in the product links this file.
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <set>
#include <string>

namespace {

struct Stream { int id; int device; std::atomic<long> launches{0}; };
struct Event { int device; std::atomic<int> recorded{0}; Stream* on = nullptr; };

std::mutex g_mu;
std::set<void*> g_allocs;
std::set<Stream*> g_streams;
std::set<Event*> g_events;
std::map<const void*, std::string> g_kernels;  // host stub address -> kernel name
std::map<std::string, long> g_launch_by_kernel;
std::atomic<long> g_mallocs{0}, g_fail_at{-1}, g_launches{0}, g_null_launches{0}, g_bad_waits{0}, g_bad_launch{0}, g_next_stream{1};
std::atomic<long> g_work_on_device[8];  // allocations, stream / event creations and launches by the device that was CURRENT when they were made
std::atomic<long> g_wrong_device{0};    // work on a stream / an event / memory of ANOTHER device than the current one
std::map<void*, int> g_alloc_device;
int g_devices = 1;
thread_local hipError_t t_last = hipSuccess;
thread_local int t_device = 0;
thread_local struct { dim3 grid, block; size_t shmem; hipStream_t stream; } t_cfg;

hipError_t fail(hipError_t e) { t_last = e; return e; }
extern "C" void __sanitizer_print_stack_trace() __attribute__((weak));
void anomaly(std::atomic<long>& counter, const char* what) {  // counted, and located: the sanitizer runtimes print the stack
    ++counter;
    std::fprintf(stderr, "hip_stub: %s\n", what);
    if (__sanitizer_print_stack_trace) __sanitizer_print_stack_trace();
}
bool live_stream(hipStream_t s) {
    if (s == nullptr) return true;
    std::lock_guard<std::mutex> lk(g_mu);
    return g_streams.count(reinterpret_cast<Stream*>(s)) != 0;
}
bool live_event(hipEvent_t e) {
    std::lock_guard<std::mutex> lk(g_mu);
    return g_events.count(reinterpret_cast<Event*>(e)) != 0;
}

}  // namespace

// ---- the books (read by tests/host/engine_harness.cpp)
extern "C" {
long hipstub_launches() { return g_launches.load(); }
long hipstub_launches_on(void* stream) { return stream ? reinterpret_cast<Stream*>(stream)->launches.load() : g_null_launches.load(); }
long hipstub_launches_of(const char* name_part) {
    std::lock_guard<std::mutex> lk(g_mu);
    long n = 0;
    for (auto& kv : g_launch_by_kernel)
        if (kv.first.find(name_part) != std::string::npos) n += kv.second;
    return n;
}
long hipstub_bad_waits() { return g_bad_waits.load(); }      // hipStreamWaitEvent / hipEventSynchronize on an event never recorded
long hipstub_bad_launches() { return g_bad_launch.load(); }  // a launch with a zero / oversized configuration or a dead stream
long hipstub_live_allocs() { std::lock_guard<std::mutex> lk(g_mu); return (long)g_allocs.size(); }
long hipstub_live_streams() { std::lock_guard<std::mutex> lk(g_mu); return (long)g_streams.size(); }
long hipstub_live_events() { std::lock_guard<std::mutex> lk(g_mu); return (long)g_events.size(); }
long hipstub_mallocs() { return g_mallocs.load(); }
long hipstub_work_on_device(int d) { return d >= 0 && d < 8 ? g_work_on_device[d].load() : -1; }
long hipstub_wrong_device() { return g_wrong_device.load(); }  // a launch / record / copy that touched another device's stream, event or memory
void hipstub_fail_malloc_at(long nth) { g_fail_at = nth; }  // the nth hipMalloc FROM NOW (1 = the next) fails once; -1: never
void hipstub_set_devices(int n) { g_devices = n; }
}

// ---- registration of the code objects (the fat binary of every translation unit registers itself at load time)
extern "C" void** __hipRegisterFatBinary(const void*) {
    static void* handle = nullptr;
    return &handle;
}
extern "C" void __hipUnregisterFatBinary(void**) {}
extern "C" void __hipRegisterFunction(void**, const void* host_fn, char*, const char* name, unsigned, void*, void*, void*, void*, int*) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_kernels[host_fn] = name ? name : "?";
}
extern "C" void __hipRegisterVar(void**, void*, char*, char*, int, size_t, int, int) {}
extern "C" hipError_t __hipPushCallConfiguration(dim3 grid, dim3 block, size_t shmem, hipStream_t stream) {
    t_cfg.grid = grid; t_cfg.block = block; t_cfg.shmem = shmem; t_cfg.stream = stream;
    return hipSuccess;
}
extern "C" hipError_t __hipPopCallConfiguration(dim3* grid, dim3* block, size_t* shmem, hipStream_t* stream) {
    *grid = t_cfg.grid; *block = t_cfg.block; *shmem = t_cfg.shmem; *stream = t_cfg.stream;
    return hipSuccess;
}

extern "C" hipError_t hipLaunchKernel(const void* fn, dim3 grid, dim3 block, void** args, size_t shmem, hipStream_t stream) {
    const unsigned long long threads = (unsigned long long)block.x * block.y * block.z, blocks = (unsigned long long)grid.x * grid.y * grid.z;
    if (threads == 0 || threads > 1024 || blocks == 0 || shmem > 160 * 1024 || args == nullptr || !live_stream(stream)) {
        anomaly(g_bad_launch, "kernel launch with a zero / oversized configuration, no arguments or a destroyed stream");
        return fail(hipErrorInvalidConfiguration);
    }
    ++g_launches;
    ++g_work_on_device[t_device & 7];
    if (stream && reinterpret_cast<Stream*>(stream)->device != t_device) anomaly(g_wrong_device, "kernel launched on a stream of another device than the current one");
    if (stream) ++reinterpret_cast<Stream*>(stream)->launches; else ++g_null_launches;
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_kernels.find(fn);
    ++g_launch_by_kernel[it == g_kernels.end() ? "?" : it->second];
    return hipSuccess;
}

// ---- devices
extern "C" hipError_t hipGetDeviceCount(int* n) { *n = g_devices; return g_devices > 0 ? hipSuccess : fail(hipErrorNoDevice); }
extern "C" hipError_t hipSetDevice(int d) { if (d < 0 || d >= g_devices) return fail(hipErrorInvalidDevice); t_device = d; return hipSuccess; }
extern "C" hipError_t hipGetDevice(int* d) { *d = t_device; return hipSuccess; }
extern "C" hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_t* p, int d) {
    if (d < 0 || d >= g_devices) return fail(hipErrorInvalidDevice);
    std::memset(p, 0, sizeof(*p));
    std::snprintf(p->name, sizeof(p->name), "hip_stub gfx950");
    std::snprintf(p->gcnArchName, sizeof(p->gcnArchName), "gfx950:sramecc+:xnack-");
    p->multiProcessorCount = 256;
    p->totalGlobalMem = (size_t)288 << 30;
    p->sharedMemPerBlock = 64 * 1024;
    p->maxSharedMemoryPerMultiProcessor = 160 * 1024;
    p->maxThreadsPerBlock = 1024;
    p->warpSize = 64;
    p->pciBusID = 1 + d;
    for (int i = 0; i < 16; ++i) p->uuid.bytes[i] = (char)(0x10 + d);
    return hipSuccess;
}
extern "C" hipError_t hipDeviceGetPCIBusId(char* buf, int len, int d) {
    if (d < 0 || d >= g_devices) return fail(hipErrorInvalidDevice);
    std::snprintf(buf, (size_t)len, "0000:%02x:00.0", 1 + d);
    return hipSuccess;
}
extern "C" hipError_t hipDeviceSynchronize() { return hipSuccess; }
extern "C" hipError_t hipGetLastError() { const hipError_t e = t_last; t_last = hipSuccess; return e; }
extern "C" const char* hipGetErrorString(hipError_t e) {
    switch (e) {
    case hipSuccess: return "no error";
    case hipErrorOutOfMemory: return "out of memory (hip_stub)";
    case hipErrorInvalidDevice: return "invalid device ordinal (hip_stub)";
    case hipErrorInvalidConfiguration: return "invalid configuration argument (hip_stub)";
    case hipErrorInvalidHandle: return "invalid resource handle (hip_stub)";
    case hipErrorInvalidValue: return "invalid argument (hip_stub)";
    default: return "error (hip_stub)";
    }
}

// ---- memory
extern "C" hipError_t hipMalloc(void** p, size_t bytes) {
    const long k = ++g_mallocs;
    (void)k;
    long f = g_fail_at.load();
    if (f > 0 && g_fail_at.compare_exchange_strong(f, f - 1) && f == 1) {
        g_fail_at = -1;
        *p = nullptr;
        return fail(hipErrorOutOfMemory);
    }
    *p = std::malloc(bytes ? bytes : 1);
    if (!*p) return fail(hipErrorOutOfMemory);
    std::memset(*p, 0, bytes);  // (device memory of a fresh allocation is not zero on a GPU; zero keeps the no-op kernels' "results" defined for UBSan)
    ++g_work_on_device[t_device & 7];
    std::lock_guard<std::mutex> lk(g_mu);
    g_allocs.insert(*p);
    g_alloc_device[*p] = t_device;
    return hipSuccess;
}
extern "C" hipError_t hipFree(void* p) {
    if (!p) return hipSuccess;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (!g_allocs.erase(p)) return fail(hipErrorInvalidValue);  // not a device pointer, or freed twice
        g_alloc_device.erase(p);
    }
    std::free(p);
    return hipSuccess;
}
extern "C" hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind) { if (bytes) std::memcpy(dst, src, bytes); return hipSuccess; }
extern "C" hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind, hipStream_t s) {
    if (!live_stream(s)) return fail(hipErrorInvalidHandle);
    if (s && reinterpret_cast<Stream*>(s)->device != t_device) anomaly(g_wrong_device, "asynchronous copy on a stream of another device than the current one");
    if (bytes) std::memcpy(dst, src, bytes);
    return hipSuccess;
}
extern "C" hipError_t hipMemsetAsync(void* dst, int v, size_t bytes, hipStream_t s) {
    if (!live_stream(s)) return fail(hipErrorInvalidHandle);
    if (bytes) std::memset(dst, v, bytes);
    return hipSuccess;
}

// ---- streams and events
extern "C" hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) {
    auto* st = new Stream;
    st->id = (int)g_next_stream++;
    st->device = t_device;
    ++g_work_on_device[t_device & 7];
    std::lock_guard<std::mutex> lk(g_mu);
    g_streams.insert(st);
    *s = reinterpret_cast<hipStream_t>(st);
    return hipSuccess;
}
extern "C" hipError_t hipStreamDestroy(hipStream_t s) {
    auto* st = reinterpret_cast<Stream*>(s);
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (!g_streams.erase(st)) return fail(hipErrorInvalidHandle);
    }
    delete st;
    return hipSuccess;
}
extern "C" hipError_t hipStreamSynchronize(hipStream_t s) { return live_stream(s) ? hipSuccess : fail(hipErrorInvalidHandle); }
extern "C" hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) {
    auto* ev = new Event;
    ev->device = t_device;
    ++g_work_on_device[t_device & 7];
    std::lock_guard<std::mutex> lk(g_mu);
    g_events.insert(ev);
    *e = reinterpret_cast<hipEvent_t>(ev);
    return hipSuccess;
}
extern "C" hipError_t hipEventCreate(hipEvent_t* e) { return hipEventCreateWithFlags(e, 0); }
extern "C" hipError_t hipEventDestroy(hipEvent_t e) {
    auto* ev = reinterpret_cast<Event*>(e);
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (!g_events.erase(ev)) return fail(hipErrorInvalidHandle);
    }
    delete ev;
    return hipSuccess;
}
extern "C" hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
    if (!live_event(e) || !live_stream(s)) return fail(hipErrorInvalidHandle);
    auto* ev = reinterpret_cast<Event*>(e);
    if (ev->device != t_device || (s && reinterpret_cast<Stream*>(s)->device != t_device)) anomaly(g_wrong_device, "event recorded across devices");
    ev->on = reinterpret_cast<Stream*>(s);
    ++ev->recorded;
    return hipSuccess;
}
extern "C" hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) {
    if (!live_event(e) || !live_stream(s)) return fail(hipErrorInvalidHandle);
    if (reinterpret_cast<Event*>(e)->recorded.load() == 0) anomaly(g_bad_waits, "hipStreamWaitEvent on an event that was never recorded");  // (legal in HIP -- a no-op -- but never what the engine means)
    return hipSuccess;
}
extern "C" hipError_t hipEventSynchronize(hipEvent_t e) {
    if (!live_event(e)) return fail(hipErrorInvalidHandle);
    if (reinterpret_cast<Event*>(e)->recorded.load() == 0) anomaly(g_bad_waits, "hipEventSynchronize on an event that was never recorded");
    return hipSuccess;
}
extern "C" hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    if (!live_event(a) || !live_event(b)) return fail(hipErrorInvalidHandle);
    if (!reinterpret_cast<Event*>(a)->recorded.load() || !reinterpret_cast<Event*>(b)->recorded.load()) return fail(hipErrorInvalidHandle);
    *ms = 1.0f;
    return hipSuccess;
}

// ---- kernel attributes (lr_inst.h: static LDS of a kernel, the opt-in for dynamic LDS beyond 64 KB)
extern "C" hipError_t hipFuncGetAttributes(hipFuncAttributes* at, const void*) {
    std::memset(at, 0, sizeof(*at));
    at->maxThreadsPerBlock = 1024;
    return hipSuccess;
}
extern "C" hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int value) { return value <= 160 * 1024 ? hipSuccess : fail(hipErrorInvalidValue); }
