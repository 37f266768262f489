// glv_multi.cpp -- the multi-GPU driver of the batched path behind the C ABI (include/glv_spectrum.h, section 3).
//
// SURVEY.md 8e / BASELINE configs[3]: streams are independent, so device g of G owns the contiguous shard
// [g*B/G, (g+1)*B/G) -- PCM, state and spectra of those streams live only on that device -- and there is NO
// data-path collective.  One host thread per device launches that device's shard on its own HIP stream; the only
// communication is one ncclAllGather of a 32-byte stats record per rank and one ncclAllReduce(max) of the elapsed
// time, over RCCL (xGMI between the GPUs of a node).  This is the C twin of what bench.py does with one process per
// GPU and torch.distributed: the host side of the reference is C, so a C host gets the same thing without Python.
//
// librccl is resolved with dlopen at glv_multi_create: a single-GPU host that never calls glv_multi_* does not need
// the library to be installed, and libglvspectrum.so carries no link-time dependency on it.  The data path needs no
// collective at all, so a host without RCCL (or GLV_MULTI_RCCL=0) still runs: the stats table is then assembled on the
// host, where every shard's record already is (all shards are driven from this process).  Only in that mode may a device be
// listed more than once (several shards on one GPU: how the several-shard machinery is tested on a one-GPU box; RCCL itself
// refuses two ranks on one device).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/glv_spectrum.h"

extern "C" int glv_set_last_error(int code, const char* msg);   // glv_api.cpp: records the thread's error string

namespace {

int failm(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
int failm(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    return glv_set_last_error(code, buf);
}

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool load(std::string& why) {
        for (const char* name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) {
            handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (handle) break;
        }
        if (!handle) { const char* e = dlerror(); why = e ? e : "dlopen(librccl) failed"; return false; }   // dlerror() clears itself: read once
#define GLV_SYM(field, sym) field = reinterpret_cast<decltype(field)>(dlsym(handle, sym)); if (!field) { why = "librccl lacks " sym; return false; }
        GLV_SYM(CommInitAll, "ncclCommInitAll")
        GLV_SYM(CommDestroy, "ncclCommDestroy")
        GLV_SYM(AllGather, "ncclAllGather")
        GLV_SYM(AllReduce, "ncclAllReduce")
        GLV_SYM(GetErrorString, "ncclGetErrorString")
#undef GLV_SYM
        return true;
    }
};

// sense-reversing barrier for the worker threads (one per device; the timed region is bracketed by it on both sides)
struct SpinBarrier {
    std::atomic<int> count{0}, phase{0};
    int n = 1;
    void wait() {
        const int ph = phase.load(std::memory_order_acquire);
        if (count.fetch_add(1, std::memory_order_acq_rel) + 1 == n) {
            count.store(0, std::memory_order_relaxed);
            phase.store(ph + 1, std::memory_order_release);
        } else {
            while (phase.load(std::memory_order_acquire) == ph) std::this_thread::yield();
        }
    }
};

}  // namespace

struct glv_multi {
    glv_params p;
    uint64_t total_streams = 0;
    unsigned ops_mask = 0;
    std::vector<int> devices;
    std::vector<uint64_t> first;          // first stream of the shard
    std::vector<uint32_t> count;          // streams of the shard
    std::vector<glv_batch*> batch;
    std::vector<hipStream_t> stream;
    std::vector<ncclComm_t> comm;
    std::vector<glv_multi_stats*> d_stats;   // per device: [ndev] records (all-gather destination); [0] of an extra slot is the source
    std::vector<double*> d_secs;             // per device: 2 doubles (in, out) for the max all-reduce
    Rccl rccl;
    bool use_rccl = false;
};

extern "C" {

void glv_multi_shard_range(uint64_t total_streams, int rank, int world, uint64_t* first, uint64_t* count) {
    // contiguous, balanced: the first (total % world) ranks take one extra stream (== glava_amd/sharding.py shard_range)
    if (world < 1 || rank < 0 || rank >= world) { if (first) *first = 0; if (count) *count = 0; return; }
    const uint64_t base = total_streams / (uint64_t) world, extra = total_streams % (uint64_t) world;
    const uint64_t r = (uint64_t) rank;
    const uint64_t lo = r * base + (r < extra ? r : extra);
    if (first) *first = lo;
    if (count) *count = base + (r < extra ? 1 : 0);
}

int glv_multi_create(const glv_params* p, uint64_t total_streams, unsigned ops_mask, const int* devices, int ndev, glv_multi** out) {
    if (!out) return failm(GLV_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!p) return failm(GLV_ERR_INVALID, "params is NULL");
    if (ndev < 1 || ndev > 64) return failm(GLV_ERR_INVALID, "ndev=%d: must be in [1, 64]", ndev);
    if (total_streams < (uint64_t) ndev) return failm(GLV_ERR_INVALID, "total_streams=%llu: fewer streams than devices", (unsigned long long) total_streams);
    const int have = glv_device_count();
    if (have <= 0) return failm(GLV_ERR_NO_DEVICE, "no usable HIP device; this library has no CPU path");
    glv_multi* m = new (std::nothrow) glv_multi();
    if (!m) return failm(GLV_ERR_NOMEM, "out of host memory");
    m->p = *p; m->total_streams = total_streams; m->ops_mask = ops_mask;
    for (int i = 0; i < ndev; ++i) {
        const int d = devices ? devices[i] : i;
        if (d < 0 || d >= have) { delete m; return failm(GLV_ERR_INVALID, "device %d out of range [0, %d)", d, have); }
        m->devices.push_back(d);
    }
    const char* env_rccl = std::getenv("GLV_MULTI_RCCL");
    const bool want_rccl = !(env_rccl && env_rccl[0] == '0');
    for (int i = 0; i < ndev; ++i)
        for (int j = 0; j < i; ++j)
            if (m->devices[j] == m->devices[i] && want_rccl) {
                const int d = m->devices[i];
                delete m;
                return failm(GLV_ERR_INVALID, "device %d listed twice (one shard per device; several shards on one device only with GLV_MULTI_RCCL=0)", d);
            }
    m->first.resize(ndev); m->count.resize(ndev);
    m->batch.assign(ndev, nullptr); m->stream.assign(ndev, nullptr); m->comm.assign(ndev, nullptr);
    m->d_stats.assign(ndev, nullptr); m->d_secs.assign(ndev, nullptr);
    int rc = GLV_OK;
    for (int i = 0; i < ndev && rc == GLV_OK; ++i) {
        uint64_t lo, cnt;
        glv_multi_shard_range(total_streams, i, ndev, &lo, &cnt);
        if (cnt > (1ull << 30)) { rc = failm(GLV_ERR_INVALID, "shard of %llu streams exceeds 2^30", (unsigned long long) cnt); break; }
        m->first[i] = lo; m->count[i] = (uint32_t) cnt;
        rc = glv_batch_create(p, (uint32_t) cnt, ops_mask, m->devices[i], &m->batch[i]);
        if (rc != GLV_OK) break;
        if (hipSetDevice(m->devices[i]) != hipSuccess || hipStreamCreateWithFlags(&m->stream[i], hipStreamNonBlocking) != hipSuccess ||
            hipMalloc(&m->d_stats[i], sizeof(glv_multi_stats) * (size_t) (ndev + 1)) != hipSuccess ||
            hipMalloc(&m->d_secs[i], sizeof(double) * 2) != hipSuccess)
            rc = failm(GLV_ERR_HIP, "device %d: stream / stats buffer allocation failed", m->devices[i]);
    }
    if (rc == GLV_OK && want_rccl) {
        std::string why;
        if (m->rccl.load(why)) {
            ncclResult_t r = m->rccl.CommInitAll(m->comm.data(), ndev, m->devices.data());
            if (r != ncclSuccess) rc = failm(GLV_ERR_HIP, "ncclCommInitAll failed: %s", m->rccl.GetErrorString(r));
            else m->use_rccl = true;
        }
        // librccl not installed: not an error -- the stats are gathered on the host (glv_multi_uses_rccl tells which)
    }
    if (rc != GLV_OK) { std::string keep = glv_last_error(); glv_multi_destroy(m); return glv_set_last_error(rc, keep.c_str()); }
    *out = m;
    return GLV_OK;
}

int glv_multi_destroy(glv_multi* m) {
    if (!m) return GLV_OK;
    for (size_t i = 0; i < m->devices.size(); ++i) {
        (void) hipSetDevice(m->devices[i]);
        if (m->use_rccl && m->comm[i]) (void) m->rccl.CommDestroy(m->comm[i]);
        if (m->batch[i]) glv_batch_destroy(m->batch[i]);
        if (m->stream[i]) (void) hipStreamDestroy(m->stream[i]);
        if (m->d_stats[i]) (void) hipFree(m->d_stats[i]);
        if (m->d_secs[i]) (void) hipFree(m->d_secs[i]);
    }
    delete m;
    return GLV_OK;
}

int glv_multi_devices(const glv_multi* m) { return m ? (int) m->devices.size() : 0; }
int glv_multi_uses_rccl(const glv_multi* m) { return m && m->use_rccl ? 1 : 0; }

int glv_multi_shard(const glv_multi* m, int idx, int* device, uint64_t* first_stream, uint32_t* streams, glv_batch** batch) {
    if (!m) return failm(GLV_ERR_INVALID, "multi is NULL");
    if (idx < 0 || idx >= (int) m->devices.size()) return failm(GLV_ERR_INVALID, "shard %d out of range", idx);
    if (device) *device = m->devices[idx];
    if (first_stream) *first_stream = m->first[idx];
    if (streams) *streams = m->count[idx];
    if (batch) *batch = m->batch[idx];
    return GLV_OK;
}

int glv_multi_run_s16(glv_multi* m, const int16_t* const* d_pcm, float* const* d_out, unsigned ops, int warmup, int steps,
                      glv_multi_stats* stats, double* max_seconds) {
    if (!m) return failm(GLV_ERR_INVALID, "multi is NULL");
    if (!d_pcm || !d_out) return failm(GLV_ERR_INVALID, "NULL pointer table");
    if (steps < 1 || warmup < 0) return failm(GLV_ERR_INVALID, "steps must be >= 1, warmup >= 0");
    const int G = (int) m->devices.size();
    // a chain that ends in gravity may leave its spectra in the state (d_out[i] == NULL, glv_batch_process_s16); every other needs a buffer
    const bool out_optional = (ops & GLV_OP_GRAVITY) && !(ops & (GLV_OP_AVERAGE | GLV_OP_SMOOTH | GLV_OP_RAW | GLV_OP_BARS | GLV_OP_R16));
    for (int i = 0; i < G; ++i) {
        if (!d_pcm[i]) return failm(GLV_ERR_INVALID, "d_pcm[%d] is NULL", i);
        if (!d_out[i] && !out_optional) return failm(GLV_ERR_INVALID, "d_out[%d] is NULL", i);
    }
    int caller_device = -1;
    (void) hipGetDevice(&caller_device);                       // worker 0 runs on the caller's thread: put its device back afterwards
    SpinBarrier bar; bar.n = G;
    std::atomic<bool> cancel{false};
    std::atomic<int> started{0};
    std::vector<int> rcs(G, GLV_OK);
    std::vector<std::string> errs(G);
    std::vector<glv_multi_stats> gathered((size_t) G * G), mine_all(G);
    std::vector<double> maxs(G, 0.0);
    auto worker = [&](int i) {
        // parked until every worker thread exists: if one could not be created, nobody enters the barriers
        while (started.load(std::memory_order_acquire) < G - 1 && !cancel.load(std::memory_order_acquire)) std::this_thread::yield();
        if (cancel.load(std::memory_order_acquire)) return;
        int rc = GLV_OK;
        auto note = [&](int code, const char* what) { if (rc == GLV_OK) { rc = code; errs[i] = what; } };
        if (hipSetDevice(m->devices[i]) != hipSuccess) note(GLV_ERR_HIP, "hipSetDevice failed");
        hipStream_t st = m->stream[i];
        for (int k = 0; k < warmup && rc == GLV_OK; ++k)
            if (glv_batch_process_s16(m->batch[i], d_pcm[i], d_out[i], ops, st) != GLV_OK) note(GLV_ERR_HIP, glv_last_error());
        if (rc == GLV_OK && hipStreamSynchronize(st) != hipSuccess) note(GLV_ERR_HIP, "warm-up synchronize failed");
        // timed region: barrier + synchronize on both sides (the contract of bench.py, in C)
        bar.wait();
        const auto t0 = std::chrono::steady_clock::now();
        if (rc == GLV_OK) (void) glv_batch_timing_begin(m->batch[i]);
        for (int k = 0; k < steps && rc == GLV_OK; ++k)
            if (glv_batch_process_s16(m->batch[i], d_pcm[i], d_out[i], ops, st) != GLV_OK) note(GLV_ERR_HIP, glv_last_error());
        if (rc == GLV_OK && hipStreamSynchronize(st) != hipSuccess) note(GLV_ERR_HIP, "synchronize failed");
        bar.wait();
        const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        double kms = 0.0; uint64_t nl = 0;
        if (rc == GLV_OK) (void) glv_batch_timing_end(m->batch[i], &kms, &nl);
        glv_multi_stats& mine = mine_all[i];
        mine.frames = (uint64_t) m->count[i] * (uint64_t) steps;
        mine.seconds = secs;
        mine.bytes = glv_batch_algorithmic_bytes(m->batch[i], ops, 1) * (uint64_t) steps;
        mine.kernel_ms = kms;
        if (!m->use_rccl) {
            // no RCCL on this host: every shard's record is in this process -- wait for all of them, copy the table
            bar.wait();
            double mx = 0.0;
            for (int j = 0; j < G; ++j) { gathered[(size_t) i * G + j] = mine_all[j]; mx = mine_all[j].seconds > mx ? mine_all[j].seconds : mx; }
            maxs[i] = mx;
            rcs[i] = rc;
            return;
        }
        // the only communication of the whole path: 32 bytes per rank, gathered on every rank, + max of the elapsed time
        glv_multi_stats* src = m->d_stats[i] + G;
        bool ok = hipMemcpyAsync(src, &mine, sizeof(mine), hipMemcpyHostToDevice, st) == hipSuccess &&
                  hipMemcpyAsync(m->d_secs[i], &secs, sizeof(double), hipMemcpyHostToDevice, st) == hipSuccess;
        if (!ok) note(GLV_ERR_HIP, "stats upload failed");
        // every worker reaches the collectives (a rank that failed still takes part, or the others would hang)
        ncclResult_t r1 = m->rccl.AllGather(src, m->d_stats[i], sizeof(glv_multi_stats), ncclUint8, m->comm[i], st);
        ncclResult_t r2 = m->rccl.AllReduce(m->d_secs[i], m->d_secs[i] + 1, 1, ncclDouble, ncclMax, m->comm[i], st);
        if (r1 != ncclSuccess) note(GLV_ERR_HIP, m->rccl.GetErrorString(r1));
        if (r2 != ncclSuccess) note(GLV_ERR_HIP, m->rccl.GetErrorString(r2));
        if (hipMemcpyAsync(&gathered[(size_t) i * G], m->d_stats[i], sizeof(glv_multi_stats) * (size_t) G, hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipMemcpyAsync(&maxs[i], m->d_secs[i] + 1, sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess)
            note(GLV_ERR_HIP, "stats download failed");
        (void) hipStreamSynchronize(st);     // nothing may still read `secs` (this frame) once the worker returns
        rcs[i] = rc;
    };
    std::vector<std::thread> th;
    bool spawn_failed = false;
    for (int i = 1; i < G; ++i) {
        try { th.emplace_back(worker, i); started.fetch_add(1, std::memory_order_release); }
        catch (...) { spawn_failed = true; cancel.store(true, std::memory_order_release); break; }
    }
    if (!spawn_failed) worker(0);
    for (auto& t : th) t.join();
    if (caller_device >= 0) (void) hipSetDevice(caller_device);
    if (spawn_failed) return failm(GLV_ERR_NOMEM, "could not start one host thread per device (%d needed)", G - 1);
    for (int i = 0; i < G; ++i)
        if (rcs[i] != GLV_OK) return failm(rcs[i], "device %d: %s", m->devices[i], errs[i].c_str());
    // every rank must have received the same table: a mismatch means the collective did not do its job
    for (int i = 1; i < G; ++i)
        if (std::memcmp(&gathered[0], &gathered[(size_t) i * G], sizeof(glv_multi_stats) * (size_t) G) != 0 || maxs[i] != maxs[0])
            return failm(GLV_ERR_HIP, "rank %d gathered a different stats table than rank 0", i);
    if (stats) std::memcpy(stats, gathered.data(), sizeof(glv_multi_stats) * (size_t) G);
    if (max_seconds) *max_seconds = maxs[0];
    return GLV_OK;
}

}  // extern "C"
