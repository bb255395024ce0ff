// Host-pointer entry points of the C ABI - what the Julia `ccall` layer binds (include/rho2sdf_hip.h):
//   r2s_eval_distances / r2s_sign_detection / r2s_sdf   the three hot lines of rho2sdf() (RhoToSDF.jl:169-171)
//   r2s_rho2sdf                                          the whole of rho2sdf() (RhoToSDF.jl:116-242) minus file I/O,
//                                                        every stage chained in HBM
//   r2s_host_alloc / r2s_host_free                       pinned result arrays for the caller
//
// Everything that costs time but is not work is kept between calls in a per-device session (released by
// r2s_release_cache): the plan with its work arrays, the device copies of the mesh, the output volumes and two
// pinned staging buffers.  Results go to the caller by DMA: directly when the destination is pinned
// (r2s_host_alloc, or registered by the caller), otherwise through the staging buffers in chunks, the next
// chunk's DMA overlapping a multi-threaded copy of the previous one into the caller's pageable pages.
//
// n_gpus > 1 (r2s_params.n_gpus): single process, one host thread per device, interleaved 4-plane tile layers
// (r2s_params.zstride/zphase); every device sends its layers straight to their place in the caller's array over
// its own PCIe link, so the host result needs no device-side stitching.  r2s_rho2sdf moves the layers to contiguous
// Z-slabs (plane-wise peer copies over xGMI, hipMemcpyPeerAsync) and post-processes SLAB-DISTRIBUTED (components with an
// interface merge, RBF smoothing with halo exchange - r2s_post.hip); nothing is gathered on one device.
// r2s_release_cache destroys the sessions: it must not run concurrently with any other call (see the header).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cctype>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <pthread.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <thread>
#include <vector>

#include "r2s_common.hpp"
#include "r2s_internal.hpp"

namespace {

double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ---- a small persistent pool of host threads: staged chunks into pageable memory, sentinel fill and tile scatter of
// the sparse download ----
// the CPUs of every NUMA node ("0-63,128-191" in /sys/devices/system/node/nodeN/cpulist); empty when sysfs says nothing
static std::vector<std::vector<int>> numa_node_cpus()
{
    std::vector<std::vector<int>> nodes;
    for (int node = 0; node < 64; ++node) {
        char path[96];
        snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
        FILE* f = fopen(path, "r");
        if (!f) break;
        std::vector<int> cpus;
        char buf[4096] = {0};
        if (fgets(buf, sizeof buf, f)) {
            for (char* p = buf; *p && *p != '\n';) {
                char* e = nullptr;
                const long a = strtol(p, &e, 10);
                if (e == p) break;
                long b = a;
                p = e;
                if (*p == '-') { b = strtol(p + 1, &e, 10); p = e; }
                for (long c = a; c <= b && c < CPU_SETSIZE; ++c) cpus.push_back((int)c);
                if (*p == ',') ++p;
            }
        }
        fclose(f);
        if (!cpus.empty()) nodes.push_back(cpus);
    }
    return nodes;
}

class CopyPool {
  public:
    // spread: thread i stays on the CPUs of NUMA node i % nodes.  The threads stream through a gigabyte per call: they need
    // the memory channels of every socket, and the pages of a fresh result array belong to the node of the thread that
    // touches them first.  Left to the scheduler the split varies from process to process (tools/numa_where.py: a pageable
    // array with 1/3 of its pages on one node 11 ms per call, with all of them on one node 14.6-21 ms; all threads on the
    // node next to the device 18-20 ms).
    explicit CopyPool(int n, bool spread = false) : n_(n)
    {
        const std::vector<std::vector<int>> nodes = spread ? numa_node_cpus() : std::vector<std::vector<int>>();
        for (int i = 0; i < n_; ++i) {
            const std::vector<int> cpus = nodes.size() > 1 ? nodes[(size_t)i % nodes.size()] : std::vector<int>();
            th_.emplace_back([this, i, cpus] {
                if (!cpus.empty()) {
                    cpu_set_t set;
                    CPU_ZERO(&set);
                    for (int c : cpus) CPU_SET(c, &set);
                    (void)pthread_setaffinity_np(pthread_self(), sizeof set, &set);   // (refused: the thread runs where it may)
                }
                loop(i);
            });
        }
    }
    ~CopyPool()
    {
        {
            std::lock_guard<std::mutex> l(mu_);
            stop_ = true;
            ++gen_;
        }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    int size() const { return n_; }
    // every thread runs fn(id, n); returns at once - wait() before the next start()
    void start(std::function<void(int, int)> fn)
    {
        std::unique_lock<std::mutex> l(mu_);
        done_.wait(l, [this] { return pending_ == 0; });   // (a generation that was never waited for: finish it first)
        fn_ = std::move(fn);
        pending_ = n_;
        ++gen_;
        cv_.notify_all();
    }
    void wait()
    {
        std::unique_lock<std::mutex> l(mu_);
        done_.wait(l, [this] { return pending_ == 0; });
    }
    void copy(void* dst, const void* src, size_t bytes)
    {
        if (bytes < (1u << 20) || n_ <= 1) {
            memcpy(dst, src, bytes);
            return;
        }
        char* d = (char*)dst;
        const char* s = (const char*)src;
        start([d, s, bytes](int id, int n) {
            // 4 KiB-aligned slices: every destination page is first touched by exactly one thread
            const size_t per = ((bytes + n - 1) / n + 4095) & ~(size_t)4095;
            const size_t lo = std::min(bytes, per * (size_t)id), hi = std::min(bytes, per * (size_t)(id + 1));
            if (hi > lo) memcpy(d + lo, s + lo, hi - lo);
        });
        wait();
    }

  private:
    void loop(int id)
    {
        uint64_t seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> l(mu_);
            cv_.wait(l, [&] { return gen_ != seen; });
            seen = gen_;
            if (stop_) return;
            std::function<void(int, int)> fn = fn_;
            l.unlock();
            fn(id, n_);
            l.lock();
            if (--pending_ == 0) done_.notify_all();
        }
    }
    int n_;
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_, done_;
    uint64_t gen_ = 0;
    bool stop_ = false;
    std::function<void(int, int)> fn_;
    int pending_ = 0;
};

constexpr size_t STAGE_BYTES = 32u << 20;

struct HostSession {
    int device = 0;
    std::mutex mu;   // one host call at a time per device
    r2s_plan* plan = nullptr;
    DevBuf dX, dI, dR, dE;
    DevBuf pre_ws[7];   // node -> element lists and centroids of the nodal densities, element volumes (r2s_rho2sdf)
    DevBuf out[4];   // dist, sign, sdf, xp
    DevBuf fine, raw, slab;
    void* stage[2] = {nullptr, nullptr};
    hipStream_t cs = nullptr;   // copy stream
    hipEvent_t ev[2] = {nullptr, nullptr};
    std::vector<hipEvent_t> evf;   // one per forwarded chunk of the smoothed field (r2s_rho2sdf)
    void* rbf_ws = nullptr;        // buffers of the smoothing stage, kept between r2s_rho2sdf calls
    CopyPool* pool = nullptr;
    DevBuf pk;                     // packed tiles of the sparse download (device)
    void* pk_host = nullptr;       // ... and their pinned landing zone
    size_t pk_host_cap = 0;
    void* fine_host = nullptr;     // pinned landing zone of the smoothed field before its level shift (r2s_rho2sdf)
    size_t fine_host_cap = 0;

    int init(int dev)
    {
        device = dev;
        HIP_TRY(hipSetDevice(dev));
        int rc = r2s_plan_create(dev, &plan);
        if (rc) return rc;
        HIP_TRY(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
        {   // direct xGMI peer copies for the multi-device gather (no-op on a one-GPU box)
            int n = 0;
            if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
            for (int d = 0; d < n; ++d) {
                int can = 0;
                if (d != dev && hipDeviceCanAccessPeer(&can, dev, d) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(d, 0);
            }
            (void)hipGetLastError();
        }
        for (int i = 0; i < 2; ++i) {
            HIP_TRY(hipHostMalloc(&stage[i], STAGE_BYTES, hipHostMallocDefault));
            HIP_TRY(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
        }
        return 0;
    }
    void release()
    {
        (void)hipSetDevice(device);
        if (plan) r2s_plan_destroy(plan);
        plan = nullptr;
        DevBuf* all[] = {&dX, &dI, &dR, &dE, &out[0], &out[1], &out[2], &out[3], &fine, &raw, &slab, &pk,
                         &pre_ws[0], &pre_ws[1], &pre_ws[2], &pre_ws[3], &pre_ws[4], &pre_ws[5], &pre_ws[6]};
        for (DevBuf* b : all) b->release();
        if (pk_host) r2s_host_free(pk_host);
        pk_host = nullptr;
        if (fine_host) r2s_host_free(fine_host);
        fine_host = nullptr;
        fine_host_cap = 0;
        pk_host_cap = 0;
        for (int i = 0; i < 2; ++i) {
            if (stage[i]) (void)hipHostFree(stage[i]);
            if (ev[i]) (void)hipEventDestroy(ev[i]);
            stage[i] = nullptr;
            ev[i] = nullptr;
        }
        for (hipEvent_t e : evf) (void)hipEventDestroy(e);
        evf.clear();
        r2s_int::rbf_workspace_release(rbf_ws);
        rbf_ws = nullptr;
        if (cs) (void)hipStreamDestroy(cs);
        cs = nullptr;
        delete pool;
        pool = nullptr;
    }
};

std::mutex g_sessions_mu;
std::map<int, HostSession*> g_sessions;

// Test hook: R2S_MULTI_OVERSUBSCRIBE=1 lets n_gpus exceed the visible devices; logical device r then runs on
// physical device r % visible (its own session, plan and buffers).  It exists so that the single-process fan-out
// (threads, tile-layer segments, peer copies) can be exercised on a one-GPU box; never for performance.
bool oversubscribe()
{
    const char* e = getenv("R2S_MULTI_OVERSUBSCRIBE");   // read per call: tests switch it on for single cases
    return e && atoi(e);
}
int physical_device(int r)
{
    if (!oversubscribe()) return r;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return r;
    return r % n;
}

// `key`: logical device of a fan-out (= the physical one unless oversubscribed)
int get_session(int key, HostSession** out)
{
    const int device = physical_device(key);
    if (oversubscribe()) key += 1000;
    std::lock_guard<std::mutex> l(g_sessions_mu);
    auto it = g_sessions.find(key);
    if (it != g_sessions.end()) {
        *out = it->second;
        return 0;
    }
    HostSession* S = new HostSession();
    int rc = S->init(device);
    if (rc) {
        S->release();
        delete S;
        return rc;
    }
    g_sessions[key] = S;
    *out = S;
    return 0;
}

bool is_pinned(const void* p)
{
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof a);
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();   // ordinary pageable memory: not an error for us
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

// host phases of the calling thread's last r2s_sdf-type call (r2s_last_host_phases): upload, run (kernels, host waits),
// pack + copies issued, wait for the sentinel fill, wait for the pieces, scatter, whole call [ms], fill threads
thread_local double g_phases[8] = {0, 0, 0, 0, 0, 0, 0, 0};

struct Segment {   // one contiguous piece of a result: device source -> host destination
    char* dst;
    const char* src;
    size_t bytes;
};

// n doubles of one value with non-temporal stores: no read-for-ownership of a gigabyte that is about to be overwritten
// (half the memory traffic of plain stores, and the caches keep what the other threads are working on)
void fill_stream(double* p, int64_t n, double v)
{
    typedef double v2d __attribute__((ext_vector_type(2)));
    int64_t i = 0;
    while (i < n && ((uintptr_t)(p + i) & 15u)) p[i++] = v;
    const v2d vv = {v, v};
    for (; i + 2 <= n; i += 2) __builtin_nontemporal_store(vv, reinterpret_cast<v2d*>(p + i));
    for (; i < n; ++i) p[i] = v;
    std::atomic_thread_fence(std::memory_order_seq_cst);   // (sfence: the streamed lines are visible before the scatter)
}

// CPUs the process may actually use: a container's share is a cgroup quota (cpu.max: "1600000 100000" = 16 CPUs) on a
// host that shows all of its CPUs to hardware_concurrency(); threads beyond the quota only get the process throttled
static unsigned usable_cpus()
{
    unsigned n = std::max(1u, std::thread::hardware_concurrency());
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {   // cgroup v2
        char q[32] = {0};
        long long period = 0;
        if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0)
            n = std::min<unsigned>(n, (unsigned)std::max<long long>(1, atoll(q) / period));
        fclose(f);
    } else {   // cgroup v1
        long long quota = -1, period = 0;
        if (FILE* fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(fq, "%lld", &quota) != 1) quota = -1; fclose(fq); }
        if (FILE* fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(fp, "%lld", &period) != 1) period = 0; fclose(fp); }
        if (quota > 0 && period > 0) n = std::min<unsigned>(n, (unsigned)std::max<long long>(1, quota / period));
    }
    return n;
}

void ensure_pool(HostSession* S)
{
    if (S->pool) return;
    const unsigned hw = usable_cpus();
    static const int env = getenv("R2S_HOST_THREADS") ? atoi(getenv("R2S_HOST_THREADS")) : 0;
    // (default: 16, or the CPU quota of the container if that is smaller; a host without a quota: half of its CPUs at most)
    const unsigned all = std::max(1u, std::thread::hardware_concurrency());
    static const bool spread_env = !(getenv("R2S_HOST_SPREAD") && atoi(getenv("R2S_HOST_SPREAD")) == 0);
    S->pool = new CopyPool(env > 0 ? std::min(env, 64) : (int)std::min(16u, std::max(2u, hw < all ? hw : all / 2)), spread_env);
}

// device -> host.  Pinned destination: plain DMA.  Pageable: DMA into the two staging buffers, each staged chunk
// spread over the pool's threads while the next chunk is in flight.
int download(HostSession* S, const std::vector<Segment>& segs, bool pinned)
{
    if (pinned) {
        for (const Segment& g : segs)
            if (g.bytes) HIP_TRY(hipMemcpyAsync(g.dst, g.src, g.bytes, hipMemcpyDeviceToHost, S->cs));
        HIP_TRY(hipStreamSynchronize(S->cs));
        return 0;
    }
    ensure_pool(S);
    std::vector<Segment> chunks;
    for (const Segment& g : segs)
        for (size_t o = 0; o < g.bytes; o += STAGE_BYTES)
            chunks.push_back({g.dst + o, g.src + o, std::min(STAGE_BYTES, g.bytes - o)});
    const size_t n = chunks.size();
    if (!n) return 0;
    HIP_TRY(hipMemcpyAsync(S->stage[0], chunks[0].src, chunks[0].bytes, hipMemcpyDeviceToHost, S->cs));
    HIP_TRY(hipEventRecord(S->ev[0], S->cs));
    for (size_t c = 0; c < n; ++c) {
        if (c + 1 < n) {   // (its staging buffer was drained by the copy of chunk c-1, which has returned)
            HIP_TRY(hipMemcpyAsync(S->stage[(c + 1) & 1], chunks[c + 1].src, chunks[c + 1].bytes, hipMemcpyDeviceToHost, S->cs));
            HIP_TRY(hipEventRecord(S->ev[(c + 1) & 1], S->cs));
        }
        HIP_TRY(hipEventSynchronize(S->ev[c & 1]));
        S->pool->copy(chunks[c].dst, S->stage[c & 1], chunks[c].bytes);
    }
    return 0;
}

int upload(void* d, const void* h, size_t bytes)
{
    HIP_TRY(hipMemcpy(d, h, bytes, hipMemcpyHostToDevice));
    return 0;
}

// Z partition of one device among G (interleaved 4-plane tile layers, the plan API's zstride/zphase): the
// segments of its local output in the caller's (nz, ny, nx) array
void layer_segments(const r2s_grid* grid, int G, int r, const char* d_local, char* h_full, size_t elem_bytes,
                    std::vector<Segment>& segs, int64_t* local_planes)
{
    const int64_t nz = grid->N[2] + 1, plane = (grid->N[0] + 1) * (grid->N[1] + 1);
    if (G <= 1) {
        segs.push_back({h_full, d_local, (size_t)(nz * plane) * elem_bytes});
        if (local_planes) *local_planes = nz;
        return;
    }
    const int64_t layers = (nz + 3) / 4;
    int64_t li = 0;
    for (int64_t t = r; t < layers; t += G, ++li) {
        const int64_t k0 = 4 * t, k1 = std::min<int64_t>(nz, k0 + 4);
        segs.push_back({h_full + (size_t)(k0 * plane) * elem_bytes, d_local + (size_t)(4 * li * plane) * elem_bytes,
                        (size_t)((k1 - k0) * plane) * elem_bytes});
    }
    if (local_planes) *local_planes = 4 * li;
}

struct HostCall {
    const double* X; int64_t nnp; const int64_t* IEN; int64_t nel; const double* rho_n; double rho_t;
    const r2s_grid* grid; r2s_params prm; int mode;
    double *dist, *sign, *sdf, *xp;
};

// one device's share of a host-pointer call
int run_host_device(const HostCall& c, int device, int G, int r, r2s_stats* stats, double* ms3 /* upload, run, download */)
{
    HostSession* S = nullptr;
    int rc = get_session(device, &S);
    if (rc) return rc;
    std::lock_guard<std::mutex> lock(S->mu);
    HIP_TRY(hipSetDevice(S->device));
    const int nen = c.prm.elem_type == R2S_TET4 ? 4 : 8;
    const int64_t nz = c.grid->N[2] + 1, plane = (c.grid->N[0] + 1) * (c.grid->N[1] + 1);
    int64_t owned_planes = nz;
    if (G > 1) {
        const int64_t layers = (nz + 3) / 4;
        if (layers <= r) return 0;   // more devices than tile layers: nothing for this one
        owned_planes = 4 * ((layers - r + G - 1) / G);
    }
    const size_t nvox = (size_t)(owned_planes * plane);
    const double t0 = now_ms();
    ENSURE(S->dX, sizeof(double) * 3 * (size_t)c.nnp);
    ENSURE(S->dR, sizeof(double) * (size_t)c.nnp);
    ENSURE(S->dI, sizeof(int64_t) * nen * (size_t)c.nel);
    if ((rc = upload(S->dX.p, c.X, sizeof(double) * 3 * (size_t)c.nnp))) return rc;
    if ((rc = upload(S->dR.p, c.rho_n, sizeof(double) * (size_t)c.nnp))) return rc;
    if ((rc = upload(S->dI.p, c.IEN, sizeof(int64_t) * nen * (size_t)c.nel))) return rc;
    double* host[4] = {c.dist, c.sign, c.sdf, c.xp};
    const int bits[4] = {R2S_OUT_DIST, R2S_OUT_SIGN, R2S_OUT_SDF, R2S_OUT_XP};
    for (int i = 0; i < 4; ++i)
        if (c.mode & bits[i]) {
            if (S->out[i].ensure_exact(sizeof(double) * (i == 3 ? 3 : 1) * nvox))
                return fail(R2S_ERR_NOMEM, "hipMalloc of an output volume (%zu voxels) failed", nvox);
        }
    r2s_params prm = c.prm;
    prm.device = S->device;
    prm.zstride = G > 1 ? G : 0;
    prm.zphase = G > 1 ? r : 0;
    prm.n_gpus = 1;
    const double t1 = now_ms();
    // SPARSE DOWNLOAD of the fused field on one device: nine voxels in ten are the sentinel -1e10, and what the caller's
    // array costs is the PCIe transfer of all of them (1.07 GB ~ 19 ms for the north-star grid against 4 ms of kernels).
    // Host threads write the sentinel into the caller's array WHILE the device works (Z slabs of whole tile layers, one
    // per thread); only the tiles that can differ from it come down - band tiles as 64 doubles, sign-only tiles as one
    // 64-bit mask (the packing of the multi-GPU stitching) - and every thread scatters the tiles of its own slab.
    static const bool sparse_env = !(getenv("R2S_HOST_SPARSE") && atoi(getenv("R2S_HOST_SPARSE")) == 0);
    const bool sparse = sparse_env && G == 1 && c.mode == R2S_OUT_SDF && c.sdf && nvox >= ((size_t)1 << 22);
    const int nx = (int)c.grid->N[0] + 1, ny = (int)c.grid->N[1] + 1;
    const int64_t layers_z = (nz + 3) / 4;
    if (sparse) {
        ensure_pool(S);
        double* const out = c.sdf;
        const int64_t pl = plane;
        S->pool->start([out, pl, nz, layers_z](int id, int n) {
            const int64_t l0 = layers_z * id / n, l1 = layers_z * (id + 1) / n;
            const int64_t k0 = 4 * l0, k1 = std::min<int64_t>(nz, 4 * l1);
            fill_stream(out + k0 * pl, (k1 - k0) * pl, -1.0e10);
        });
    }
    r2s_stats st_local;
    r2s_stats* const stp = stats ? stats : &st_local;
    rc = r2s_plan_run_dev(S->plan, S->dX.as<double>(), c.nnp, S->dI.as<int64_t>(), c.nel, S->dR.as<double>(), c.rho_t, c.grid,
                          &prm, 0, nz, c.mode, S->out[0].as<double>(), S->out[1].as<double>(), S->out[2].as<double>(),
                          S->out[3].as<double>(), nullptr, stp);
    if (rc) {
        if (sparse) S->pool->wait();
        return rc;
    }
    const double t2 = now_ms();
    if (sparse) {
        const int64_t nf = stp->n_active_tiles, nm = stp->n_sign_only_tiles;
        // [payload nf x 64 doubles | masks nm x u64 | ids nf x u32 | mask ids nm x u32]
        const size_t off_masks = (size_t)nf * 512, off_ids = off_masks + (size_t)nm * 8, off_mids = off_ids + (((size_t)nf * 4 + 7) & ~(size_t)7);
        const size_t bytes = off_mids + (((size_t)nm * 4 + 7) & ~(size_t)7);
        auto bail = [&](int code) { S->pool->wait(); return code; };
        if (S->pk.ensure(std::max<size_t>(bytes, 8))) return bail(fail(R2S_ERR_NOMEM, "hipMalloc of the packed tiles failed"));
        if (S->pk_host_cap < bytes) {
            if (S->pk_host) r2s_host_free(S->pk_host);
            S->pk_host = nullptr;
            S->pk_host_cap = 0;
            const size_t want = bytes + bytes / 4 + 4096;
            if (!(S->pk_host = r2s_host_alloc(want))) {   // (pages interleaved over the NUMA nodes: the scatter threads are spread)
                (void)hipGetLastError();
                return bail(fail(R2S_ERR_NOMEM, "hipHostMalloc of the landing zone of the packed tiles failed"));
            }
            S->pk_host_cap = want;
        }
        char* const dpk = (char*)S->pk.p;
        int64_t nf2 = 0, nm2 = 0;
        rc = r2s_plan_pack_tiles2_dev(S->plan, S->out[2].as<double>(), (double*)dpk, (uint32_t*)(dpk + off_ids), nf,
                                      (uint64_t*)(dpk + off_masks), (uint32_t*)(dpk + off_mids), nm, &nf2, &nm2, S->cs);
        if (rc) return bail(rc);
        if (nf2 != nf || nm2 != nm) return bail(fail(R2S_ERR_HIP, "sparse download: tile counts changed between run and pack"));
        // Every thread scatters the tiles of ITS Z slab (the one it filled: its pages, its cache lines - a tile row is half
        // a cache line).  Two phases: the masks of the sign-only tiles and all ids come down first (6 MB) and are scattered
        // while the 97 MB of band tiles are on the bus.  (Band tiles in 2-8 pieces, scattered piece by piece behind the
        // transfer: slower - the tile list is roughly Z-ordered, so a piece keeps a fraction of the threads busy; shares
        // of the list instead of slabs: 9 instead of 4-5 ms.)
        constexpr int NPIECE = 2;
        static const bool mask_skip = !(getenv("R2S_HOST_MASKSKIP") && atoi(getenv("R2S_HOST_MASKSKIP")) == 0);
        // streaming stores of the scatter: into pinned arrays only (measured: 8.7-9.6 -> 8.0 ms there, 9.1 -> 9.7 ms into
        // pageable memory; R2S_HOST_NT=0 / 1 forces one or the other)
        static const int nt_force = getenv("R2S_HOST_NT") ? atoi(getenv("R2S_HOST_NT")) : -1;
        const bool nt_env = nt_force >= 0 ? nt_force != 0 : is_pinned(c.sdf);
        while (S->evf.size() < (size_t)NPIECE) {   // (an event joins the list only once it exists)
            hipEvent_t e = nullptr;
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return bail(fail(R2S_ERR_HIP, "hipEventCreate failed"));
            S->evf.push_back(e);
        }
        char* const hpk_w = (char*)S->pk_host;
        const int64_t f0[NPIECE + 1] = {0, 0, nf}, m0[NPIECE + 1] = {0, nm, nm};   // phase 0: masks, phase 1: band tiles
        if (bytes > off_masks && hipMemcpyAsync(hpk_w + off_masks, dpk + off_masks, bytes - off_masks, hipMemcpyDeviceToHost, S->cs) != hipSuccess)
            return bail(fail(R2S_ERR_HIP, "copy of the packed tiles failed"));
        if (hipEventRecord(S->evf[0], S->cs) != hipSuccess) return bail(fail(R2S_ERR_HIP, "hipEventRecord failed"));
        if (off_masks && hipMemcpyAsync(hpk_w, dpk, off_masks, hipMemcpyDeviceToHost, S->cs) != hipSuccess)
            return bail(fail(R2S_ERR_HIP, "copy of the packed tiles failed"));
        if (hipEventRecord(S->evf[1], S->cs) != hipSuccess) return bail(fail(R2S_ERR_HIP, "hipEventRecord failed"));
        const double t3 = now_ms();
        S->pool->wait();   // the sentinel is everywhere
        const double t4 = now_ms();
        const char* const hpk = (const char*)S->pk_host;
        double* const out = c.sdf;
        const int ntx = (nx + 3) / 4, nty = (ny + 3) / 4;
        const int64_t nzz = nz;
        double t_ev = 0.0, t_sc = 0.0;
        for (int q = 0; q < NPIECE; ++q) {
            const double ta = now_ms();
            if (hipEventSynchronize(S->evf[q]) != hipSuccess) return fail(R2S_ERR_HIP, "copy of the packed tiles failed");
            const double tb = now_ms();
            t_ev += tb - ta;
            const int64_t fa = f0[q], fb = f0[q + 1], ma = m0[q], mb2 = m0[q + 1];
            S->pool->start([=](int id, int n) {
                const double* payload = (const double*)hpk;
                const uint64_t* masks = (const uint64_t*)(hpk + off_masks);
                const uint32_t* ids = (const uint32_t*)(hpk + off_ids);
                const uint32_t* mids = (const uint32_t*)(hpk + off_mids);
                const int64_t l0 = layers_z * id / n, l1 = layers_z * (id + 1) / n;
                const uint32_t t_lo = (uint32_t)(l0 * nty * ntx), t_hi = (uint32_t)(l1 * nty * ntx);   // tile ids of the slab
                // Streaming stores (no read-for-ownership of lines the fill has just streamed out): a tile row is half a
                // cache line, so x-neighbouring tiles (consecutive in the list more often than not) are written together,
                // row by row - the two halves of a line leave the write-combining buffer as one line.
                typedef double v4d __attribute__((ext_vector_type(4)));
                const bool nt_ok = nt_env && (nx % 4 == 0) && (((uintptr_t)out) & 31u) == 0;   // every full tile row is 32-byte aligned
                for (int64_t w = fa; w < fb; ++w) {
                    const uint32_t t = ids[w];
                    if (t < t_lo || t >= t_hi) continue;
                    const int tx = (int)(t % (uint32_t)ntx), ty = (int)((t / (uint32_t)ntx) % (uint32_t)nty);
                    const int64_t tz = t / ((uint32_t)ntx * (uint32_t)nty);
                    const double* src = payload + w * 64;
                    const int i0 = 4 * tx, wx = std::min(4, nx - i0);
                    const bool pair = nt_ok && (tx & 1) == 0 && tx + 1 < ntx && w + 1 < fb && ids[w + 1] == t + 1u;
                    for (int z = 0; z < 4; ++z) {
                        const int64_t k = 4 * tz + z;
                        if (k >= nzz) break;
                        for (int y = 0; y < 4; ++y) {
                            const int j = 4 * ty + y;
                            if (j >= ny) break;
                            double* dst = out + (k * ny + j) * (int64_t)nx + i0;
                            const double* s4 = src + 16 * z + 4 * y;
                            if (pair) {
                                __builtin_nontemporal_store(*reinterpret_cast<const v4d*>(s4), reinterpret_cast<v4d*>(dst));
                                __builtin_nontemporal_store(*reinterpret_cast<const v4d*>(s4 + 64), reinterpret_cast<v4d*>(dst + 4));
                            } else if (nt_ok) {
                                __builtin_nontemporal_store(*reinterpret_cast<const v4d*>(s4), reinterpret_cast<v4d*>(dst));
                            } else {
                                for (int x = 0; x < wx; ++x) dst[x] = s4[x];
                            }
                        }
                    }
                    if (pair) ++w;
                }
                for (int64_t w = ma; w < mb2; ++w) {
                    const uint32_t t = mids[w];
                    if (t < t_lo || t >= t_hi) continue;
                    const uint64_t m = masks[w];
                    if (m == 0 && mask_skip) continue;   // every voxel -1e10: the sentinel is there already
                    const int tx = (int)(t % (uint32_t)ntx), ty = (int)((t / (uint32_t)ntx) % (uint32_t)nty);
                    const int64_t tz = t / ((uint32_t)ntx * (uint32_t)nty);
                    const int i0 = 4 * tx, wx = std::min(4, nx - i0);
                    for (int z = 0; z < 4; ++z) {
                        const int64_t k = 4 * tz + z;
                        if (k >= nzz) break;
                        for (int y = 0; y < 4; ++y) {
                            const int j = 4 * ty + y;
                            if (j >= ny) break;
                            const unsigned row = (unsigned)((m >> (16 * z + 4 * y)) & 15ull);
                            if (!row && mask_skip) continue;
                            double* dst = out + (k * ny + j) * (int64_t)nx + i0;
                            if (nt_ok) {
                                v4d v;
                                v.x = (row & 1u) ? 1.0e10 : -1.0e10; v.y = (row & 2u) ? 1.0e10 : -1.0e10;
                                v.z = (row & 4u) ? 1.0e10 : -1.0e10; v.w = (row & 8u) ? 1.0e10 : -1.0e10;
                                __builtin_nontemporal_store(v, reinterpret_cast<v4d*>(dst));
                            } else {
                                for (int x = 0; x < wx; ++x) dst[x] = ((row >> x) & 1u) ? 1.0e10 : -1.0e10;
                            }
                        }
                    }
                }
                std::atomic_thread_fence(std::memory_order_seq_cst);   // (sfence: the streamed rows are visible to the caller)
            });
            S->pool->wait();
            t_sc += now_ms() - tb;
        }
        static const bool timing_env = getenv("R2S_HOST_TIMING") && atoi(getenv("R2S_HOST_TIMING"));
        if (timing_env)
            fprintf(stderr, "[r2s host] upload %.2f run %.2f pack+copy (%.1f MB) %.2f wait for the fill %.2f scatter %.2f ms, %d threads\n",
                    t1 - t0, t2 - t1, bytes / 1e6, t3 - t2, t4 - t3, now_ms() - t4, S->pool->size()),
            fprintf(stderr, "[r2s host]   waiting for the pieces %.2f, scattering %.2f ms\n", t_ev, t_sc);
        {
            const double t5 = now_ms();
            const double ph[8] = {t1 - t0, t2 - t1, t3 - t2, t4 - t3, t_ev, t_sc, t5 - t0, (double)S->pool->size()};
            memcpy(g_phases, ph, sizeof ph);
        }
        if (ms3) { ms3[0] = t1 - t0; ms3[1] = t2 - t1; ms3[2] = now_ms() - t2; }
        return 0;
    }
    for (int i = 0; i < 4; ++i)
        if (c.mode & bits[i]) {
            std::vector<Segment> segs;
            const size_t eb = sizeof(double) * (i == 3 ? 3 : 1);
            layer_segments(c.grid, G, r, (const char*)S->out[i].p, (char*)host[i], eb, segs, nullptr);
            if ((rc = download(S, segs, is_pinned(host[i])))) return rc;
        }
    {
        const double t5 = now_ms();
        const double ph[8] = {t1 - t0, t2 - t1, 0.0, 0.0, 0.0, t5 - t2, t5 - t0, 0.0};   // (dense download: all of it under "scatter")
        memcpy(g_phases, ph, sizeof ph);
    }
    if (ms3) { ms3[0] = t1 - t0; ms3[1] = t2 - t1; ms3[2] = now_ms() - t2; }
    return 0;
}

int resolve_device(int device, int* out)
{
    int rc = check_device(device < 0 ? 0 : device);
    if (rc) return rc;
    if (device < 0) HIP_TRY(hipGetDevice(&device));
    *out = device;
    return 0;
}

int n_gpus_of(const r2s_params& p, int* G)
{
    int g = p.n_gpus > 1 ? p.n_gpus : 1;
    if (g > 1) {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
        if (g > n && !oversubscribe()) return fail(R2S_ERR_ARG, "n_gpus = %d but only %d HIP device(s) are visible", g, n);
        if (g > 64) return fail(R2S_ERR_ARG, "n_gpus = %d is not a node", g);
    }
    *G = g;
    return 0;
}

// runs fn(device index r of G) on G host threads (one per device); first error wins
int fan_out(int G, const std::function<int(int)>& fn)
{
    if (G <= 1) return fn(0);
    std::vector<int> rcs((size_t)G, 0);
    std::vector<std::string> errs((size_t)G);
    std::vector<std::thread> th;
    for (int r = 0; r < G; ++r)
        th.emplace_back([&, r] {
            rcs[(size_t)r] = fn(r);
            if (rcs[(size_t)r]) errs[(size_t)r] = g_err;   // the error text is thread-local
        });
    for (auto& t : th) t.join();
    for (int r = 0; r < G; ++r)
        if (rcs[(size_t)r]) return fail(rcs[(size_t)r], "device %d: %s", r, errs[(size_t)r].c_str());
    return 0;
}

int run_host(const double* X, int64_t nnp, const int64_t* IEN, int64_t nel, const double* rho_n, double rho_t,
             const r2s_grid* grid, const r2s_params* params, int mode, double* dist, double* sign, double* sdf,
             double* xp, r2s_stats* stats)
{
    if (!X || !IEN || !rho_n || !grid) return fail(R2S_ERR_ARG, "null argument");
    HostCall c{X, nnp, IEN, nel, rho_n, rho_t, grid, {}, mode, dist, sign, sdf, xp};
    if (params) c.prm = *params; else r2s_default_params(&c.prm);
    int G = 1, dev0 = 0, rc;
    if ((rc = resolve_device(c.prm.device, &dev0))) return rc;
    if ((rc = n_gpus_of(c.prm, &G))) return rc;
    if (G == 1) return run_host_device(c, dev0, 1, 0, stats, nullptr);
    // devices 0..G-1; the stats of device 0's share are reported
    return fan_out(G, [&](int r) { return run_host_device(c, r, G, r, r == 0 ? stats : nullptr, nullptr); });
}

}  // namespace

namespace r2s_int {
void release_host_sessions()
{
    std::lock_guard<std::mutex> l(g_sessions_mu);
    for (auto& kv : g_sessions) {
        std::lock_guard<std::mutex> l2(kv.second->mu);
        kv.second->release();
    }
    for (auto& kv : g_sessions) delete kv.second;
    g_sessions.clear();
}
}  // namespace r2s_int

extern "C" {

void r2s_last_host_phases(double out[8])
{
    if (out) memcpy(out, g_phases, sizeof g_phases);
}

// Pinned result arrays with their pages INTERLEAVED over the NUMA nodes: hipHostMalloc puts every page on the node next
// to the device (and ignores the caller's memory policy, also with hipHostMallocNumaUser), so the 16 host threads that
// fill and scatter into such an array - spread over the sockets - all write to one socket's memory: 13-16 ms per call
// against 10.4-12 into a pageable array whose pages they touch first, half on each node (tools/numa_where.py).  So: an
// anonymous mapping with the interleave policy (mbind), registered with the runtime.  Any step that fails: hipHostMalloc.
static std::mutex g_reg_mu;
static std::map<void*, size_t> g_reg;   // mappings of r2s_host_alloc that r2s_host_free has to unregister and unmap

static void* host_alloc_interleaved(size_t bytes)
{
#if defined(__linux__) && defined(__x86_64__)
    static const bool on = !(getenv("R2S_HOST_INTERLEAVE") && atoi(getenv("R2S_HOST_INTERLEAVE")) == 0);
    if (!on || bytes < ((size_t)1 << 20)) return nullptr;
    unsigned long mask = 0;
    if (FILE* f = fopen("/sys/devices/system/node/online", "r")) {   // "0-1", "0", "0-3", "0,2"
        char buf[256] = {0};
        if (fgets(buf, sizeof buf, f)) {
            for (char* c = buf; *c && *c != '\n';) {
                char* e = nullptr;
                const long a = strtol(c, &e, 10);
                if (e == c) break;
                long b = a;
                c = e;
                if (*c == '-') { b = strtol(c + 1, &e, 10); c = e; }
                for (long q = a; q <= b && q < 64; ++q) mask |= 1ul << q;
                if (*c == ',') ++c;
            }
        }
        fclose(f);
    }
    if (!mask || !(mask & (mask - 1))) return nullptr;   // one node: nothing to interleave
    const size_t size = (bytes + (((size_t)2 << 20) - 1)) & ~(((size_t)2 << 20) - 1);
    void* p = mmap(nullptr, size, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (p == MAP_FAILED) return nullptr;
    if (syscall(237 /* mbind */, p, size, 3 /* MPOL_INTERLEAVE */, &mask, 65ul, 0u) != 0 ||
        hipHostRegister(p, size, hipHostRegisterPortable) != hipSuccess) {
        (void)hipGetLastError();
        munmap(p, size);
        return nullptr;
    }
    std::lock_guard<std::mutex> lock(g_reg_mu);
    g_reg[p] = size;
    return p;
#else
    (void)bytes;
    return nullptr;
#endif
}

void* r2s_host_alloc(size_t bytes)
{
    void* p = nullptr;
    if (check_device(0)) return nullptr;
    if ((p = host_alloc_interleaved(bytes))) return p;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable) != hipSuccess) {
        (void)hipGetLastError();
        fail(R2S_ERR_NOMEM, "hipHostMalloc of %zu bytes failed", bytes);
        return nullptr;
    }
    return p;
}

void r2s_host_free(void* p)
{
    if (!p) return;
    size_t size = 0;
    {
        std::lock_guard<std::mutex> lock(g_reg_mu);
        auto it = g_reg.find(p);
        if (it != g_reg.end()) { size = it->second; g_reg.erase(it); }
    }
    if (size) {
        (void)hipHostUnregister(p);
        munmap(p, size);
    } else {
        (void)hipHostFree(p);
    }
}

int r2s_eval_distances(const double* X, int64_t nnp, const int64_t* IEN, int64_t nel, const double* rho_n,
                       double rho_t, const r2s_grid* grid, const r2s_params* params, double* dist_out,
                       double* xp_out, r2s_stats* stats)
{
    if (!dist_out) return fail(R2S_ERR_ARG, "dist_out is null");
    return run_host(X, nnp, IEN, nel, rho_n, rho_t, grid, params, R2S_OUT_DIST | (xp_out ? R2S_OUT_XP : 0),
                    dist_out, nullptr, nullptr, xp_out, stats);
}

int r2s_sign_detection(const double* X, int64_t nnp, const int64_t* IEN, int64_t nel, const double* rho_n,
                       double rho_t, const r2s_grid* grid, const r2s_params* params, double* signs_out,
                       r2s_stats* stats)
{
    if (!signs_out) return fail(R2S_ERR_ARG, "signs_out is null");
    return run_host(X, nnp, IEN, nel, rho_n, rho_t, grid, params, R2S_OUT_SIGN, nullptr, signs_out, nullptr,
                    nullptr, stats);
}

int r2s_sdf(const double* X, int64_t nnp, const int64_t* IEN, int64_t nel, const double* rho_n, double rho_t,
            const r2s_grid* grid, const r2s_params* params, double* sdf_out, r2s_stats* stats)
{
    if (!sdf_out) return fail(R2S_ERR_ARG, "sdf_out is null");
    return run_host(X, nnp, IEN, nel, rho_n, rho_t, grid, params, R2S_OUT_SDF, nullptr, nullptr, sdf_out,
                    nullptr, stats);
}

void r2s_default_options(r2s_options* o)
{
    memset(o, 0, sizeof *o);
    o->threshold_density = NAN;               // Rho2sdfOptions: nothing -> find_threshold_for_volume
    o->band_factor = 1.1;                     // sdfOnDensityField.jl:158
    o->artifact_min_component_ratio = 0.01;   // RhoToSDF.jl:33
    o->rbf_kernel_threshold = 1e-3;           // RBFs4Smoothing.jl:328
    o->elem_type = R2S_HEX8;
    o->rbf_interp = 1;
    o->rbf_smooth = 1;
    o->remove_artifacts = 1;
    o->device = -1;
    o->n_gpus = 1;
}

int r2s_rho2sdf(const double* X, int64_t nnp, const int64_t* IEN, int64_t nel, const double* rho_e,
                const r2s_options* options, const r2s_grid* grid, double* rho_n_out, double* sdf_raw_out,
                double* sdf_dists_out, float* fine_sdf_out, r2s_run_info* info)
{
    if (!X || !IEN || !rho_e || !grid || nnp <= 0 || nel <= 0) return fail(R2S_ERR_ARG, "r2s_rho2sdf: null / empty argument");
    r2s_options o;
    if (options) o = *options; else r2s_default_options(&o);
    if (o.elem_type != R2S_HEX8 && o.elem_type != R2S_TET4) return fail(R2S_ERR_UNSUPPORTED, "unknown element type %d", o.elem_type);
    if (!o.skip_rbf && !fine_sdf_out) return fail(R2S_ERR_ARG, "fine_sdf_out is null (set skip_rbf to stop after the raw SDF)");
    if (o.rbf_smooth < 1) o.rbf_smooth = 1;
    int dev0 = 0, rc, G = o.n_gpus > 1 ? o.n_gpus : 1;
    if ((rc = resolve_device(o.device, &dev0))) return rc;
    if (G > 1) {
        r2s_params t;
        r2s_default_params(&t);
        t.n_gpus = G;
        if ((rc = n_gpus_of(t, &G))) return rc;
        dev0 = 0;   // logical device 0 of the fan-out
    }
    HostSession* S = nullptr;
    if ((rc = get_session(dev0, &S))) return rc;
    std::unique_lock<std::mutex> lock(S->mu);
    dev0 = S->device;
    HIP_TRY(hipSetDevice(dev0));
    r2s_run_info ri;
    memset(&ri, 0, sizeof ri);
    const double t_start = now_ms();
    const int nen = o.elem_type == R2S_TET4 ? 4 : 8;
    const int64_t ngp = grid->ngp;
    // ---- upload: X, IEN, element densities (RhoToSDF.jl:128) ----
    ENSURE(S->dX, sizeof(double) * 3 * (size_t)nnp);
    ENSURE(S->dR, sizeof(double) * (size_t)nnp);
    ENSURE(S->dE, sizeof(double) * (size_t)nel);
    ENSURE(S->dI, sizeof(int64_t) * nen * (size_t)nel);
    if ((rc = upload(S->dX.p, X, sizeof(double) * 3 * (size_t)nnp))) return rc;
    if ((rc = upload(S->dI.p, IEN, sizeof(int64_t) * nen * (size_t)nel))) return rc;
    if ((rc = upload(S->dE.p, rho_e, sizeof(double) * (size_t)nel))) return rc;
    double t = now_ms();
    ri.ms_upload = t - t_start;
    // ---- pre-stage: mesh volume (:128), nodal densities (:148), threshold (:151-156) ----
    if ((rc = r2s_int::mesh_volume_dev(S->dX.as<double>(), S->dI.as<int64_t>(), nel, o.elem_type, S->dE.as<double>(),
                                       &ri.V_domain, &ri.V_frac, S->pre_ws + 5)))
        return rc;
    const double tp1 = now_ms();
    if ((rc = r2s_int::dense_in_nodes_dev(S->dX.as<double>(), nnp, S->dI.as<int64_t>(), nel, o.elem_type, S->dE.as<double>(),
                                          S->dR.as<double>(), S->pre_ws)))
        return rc;
    const double tp2 = now_ms();
    if (std::isnan(o.threshold_density)) {
        int it = 0;
        if ((rc = r2s_int::find_threshold_dev(S->dX.as<double>(), S->dI.as<int64_t>(), nel, o.elem_type, S->dR.as<double>(),
                                              ri.V_domain * ri.V_frac, 1e-4, 60, &ri.rho_t, &it)))
            return rc;
        ri.threshold_iters = it;
    } else {
        ri.rho_t = o.threshold_density;
    }
    if (rho_n_out) HIP_TRY(hipMemcpy(rho_n_out, S->dR.p, sizeof(double) * (size_t)nnp, hipMemcpyDeviceToHost));
    double t2 = now_ms();
    ri.ms_pre = t2 - t;
    {
        static const bool timing_env = getenv("R2S_HOST_TIMING") && atoi(getenv("R2S_HOST_TIMING"));
        if (timing_env)
            fprintf(stderr, "r2s pre: mesh volume %.2f ms, nodal densities %.2f, threshold %.2f (%d iterations)\n", tp1 - t, tp2 - tp1,
                    t2 - tp2, ri.threshold_iters);
    }
    // ---- raw SDF = dists .* signs (:169-171) ----
    if (G == 1 && S->out[2].ensure_exact(sizeof(double) * (size_t)ngp)) return fail(R2S_ERR_NOMEM, "hipMalloc of the SDF volume failed");
    r2s_params prm;
    r2s_default_params(&prm);
    prm.band_factor = o.band_factor;
    prm.elem_type = o.elem_type;
    prm.true_min = o.true_min;
    prm.sign_no_inner = o.sign_no_inner;
    prm.device = dev0;
    r2s_stats st;
    if (G == 1) {
        if ((rc = r2s_plan_run_dev(S->plan, S->dX.as<double>(), nnp, S->dI.as<int64_t>(), nel, S->dR.as<double>(), ri.rho_t, grid,
                                   &prm, 0, grid->N[2] + 1, R2S_OUT_SDF, nullptr, nullptr, S->out[2].as<double>(), nullptr,
                                   nullptr, &st)))
            return rc;
    } else {
        // ---- n_gpus > 1: nothing is ever gathered on one device --------------------------------------------------
        // raw SDF: every device computes its interleaved 4-plane tile layers (balanced: the band is not uniform in z)
        // from its own copy of (X, IEN, rho_n).  Post-processing works on contiguous Z-slabs with halos, so the layers
        // then travel to the slab(s) that hold their planes (peer copies over xGMI), the components are labelled per
        // slab and merged over the interface planes, the smoothing exchanges halos of its vectors, and every device
        // sends its planes of the two results straight to the caller's arrays.
        std::vector<double> h_rn((size_t)nnp);
        HIP_TRY(hipMemcpy(h_rn.data(), S->dR.p, sizeof(double) * (size_t)nnp, hipMemcpyDeviceToHost));
        const int64_t nz = grid->N[2] + 1, plane = (grid->N[0] + 1) * (grid->N[1] + 1), layers = (nz + 3) / 4;
        const int H = 4;   // planes a smoothing stencil can reach beyond a slab (build_stencil: offsets -3..4)
        std::vector<HostSession*> T((size_t)G, nullptr);
        std::vector<r2s_int::Slab> slabs((size_t)G);
        for (int r = 0; r < G; ++r)
            if ((rc = get_session(r, &T[(size_t)r]))) return rc;
        {
            const int64_t per = (nz + G - 1) / G;
            for (int r = 0; r < G; ++r) {
                r2s_int::Slab& sl = slabs[(size_t)r];
                sl.device = T[(size_t)r]->device;
                sl.stream = T[(size_t)r]->cs;
                sl.k0 = (int)std::min<int64_t>(nz, r * per);
                sl.k1 = (int)std::min<int64_t>(nz, (r + 1) * per);
                sl.h0 = std::max(0, sl.k0 - H);
                sl.h1 = (int)std::min<int64_t>(nz, sl.k1 + H);
                sl.d_sdf = nullptr;
            }
        }
        lock.unlock();   // device 0's own share takes the session lock again
        rc = fan_out(G, [&](int r) -> int {
            HostSession* Tr = T[(size_t)r];
            int rc2 = 0;
            std::lock_guard<std::mutex> l2(Tr->mu);
            HIP_TRY(hipSetDevice(Tr->device));
            const r2s_int::Slab& sl = slabs[(size_t)r];
            if (sl.k1 > sl.k0 && Tr->slab.ensure_exact(sizeof(double) * (size_t)((sl.h1 - sl.h0) * plane)))
                return fail(R2S_ERR_NOMEM, "hipMalloc of a slab failed");
            if (layers <= r) return 0;
            const int64_t owned = 4 * ((layers - r + G - 1) / G);
            if (r != 0) {
                ENSURE(Tr->dX, sizeof(double) * 3 * (size_t)nnp);
                ENSURE(Tr->dR, sizeof(double) * (size_t)nnp);
                ENSURE(Tr->dI, sizeof(int64_t) * nen * (size_t)nel);
                if ((rc2 = upload(Tr->dX.p, X, sizeof(double) * 3 * (size_t)nnp))) return rc2;
                if ((rc2 = upload(Tr->dI.p, IEN, sizeof(int64_t) * nen * (size_t)nel))) return rc2;
                if ((rc2 = upload(Tr->dR.p, h_rn.data(), sizeof(double) * (size_t)nnp))) return rc2;
            }
            if (Tr->raw.ensure_exact(sizeof(double) * (size_t)(owned * plane))) return fail(R2S_ERR_NOMEM, "hipMalloc failed");
            r2s_params p2 = prm;
            p2.device = Tr->device;
            p2.zstride = G;
            p2.zphase = r;
            r2s_stats st2;
            if ((rc2 = r2s_plan_run_dev(Tr->plan, Tr->dX.as<double>(), nnp, Tr->dI.as<int64_t>(), nel, Tr->dR.as<double>(), ri.rho_t,
                                        grid, &p2, 0, nz, R2S_OUT_SDF, nullptr, nullptr, Tr->raw.as<double>(), nullptr, nullptr,
                                        &st2)))
                return rc2;
            if (r == 0) st = st2;
            return 0;
        });
        lock.lock();
        if (rc) return rc;
        std::vector<std::unique_lock<std::mutex>> others;   // the post-processing uses every device's session (always locked in this order)
        for (int r = 1; r < G; ++r) others.emplace_back(T[(size_t)r]->mu);
        for (int r = 0; r < G; ++r) slabs[(size_t)r].d_sdf = T[(size_t)r]->slab.as<double>();
        // layers -> slabs: plane k of layer tl of device r goes to every slab that holds it (all-to-all over xGMI)
        for (int r = 0; r < G; ++r) {
            HostSession* Tr = T[(size_t)r];
            HIP_TRY(hipSetDevice(Tr->device));
            int64_t li = 0;
            for (int64_t tl = r; tl < layers; tl += G, ++li)
                for (int64_t k = 4 * tl; k < std::min<int64_t>(nz, 4 * tl + 4); ++k)
                    for (int q = 0; q < G; ++q) {
                        const r2s_int::Slab& sl = slabs[(size_t)q];
                        if (sl.k1 <= sl.k0 || k < sl.h0 || k >= sl.h1) continue;
                        HIP_TRY(hipMemcpyPeerAsync(sl.d_sdf + (k - sl.h0) * plane, sl.device,
                                                   Tr->raw.as<double>() + (4 * li + (k - 4 * tl)) * plane, Tr->device,
                                                   sizeof(double) * (size_t)plane, Tr->cs));
                    }
        }
        for (int r = 0; r < G; ++r) {
            HIP_TRY(hipSetDevice(T[(size_t)r]->device));
            HIP_TRY(hipStreamSynchronize(T[(size_t)r]->cs));
        }
        const double t3m = now_ms();
        ri.ms_sdf = t3m - t2;
        ri.ms_sdf_kernels = st.ms_prep + st.ms_bins + st.ms_main + st.ms_gather;
        // owned planes of a per-slab Float64 field -> their place in a host array
        auto slabs_to_host = [&](double* host) -> int {
            for (int r = 0; r < G; ++r) {
                const r2s_int::Slab& sl = slabs[(size_t)r];
                if (sl.k1 <= sl.k0) continue;
                HIP_TRY(hipSetDevice(sl.device));
                std::vector<Segment> segs{{(char*)(host + (int64_t)sl.k0 * plane), (const char*)(sl.d_sdf + (int64_t)(sl.k0 - sl.h0) * plane),
                                           sizeof(double) * (size_t)((sl.k1 - sl.k0) * plane)}};
                int rc2 = download(T[(size_t)r], segs, is_pinned(host));
                if (rc2) return rc2;
            }
            return 0;
        };
        if (sdf_raw_out && (rc = slabs_to_host(sdf_raw_out))) return rc;
        if (o.remove_artifacts) {
            if ((rc = r2s_int::remove_artifacts_slabs(slabs, grid, 0.0, o.artifact_min_component_ratio, &ri.n_flipped))) return rc;
            std::vector<void*> base((size_t)G);
            for (int r = 0; r < G; ++r) base[(size_t)r] = slabs[(size_t)r].d_sdf;
            if ((rc = r2s_int::exchange_halo_slabs(slabs, base, sizeof(double), plane, H))) return rc;   // flipped voxels reach the halos
        }
        const double t4m = now_ms();
        ri.ms_artifacts = t4m - t3m;
        if (sdf_dists_out && (rc = slabs_to_host(sdf_dists_out))) return rc;
        const double t5m = now_ms();
        if (!o.skip_rbf) {
            int its = 0;
            if ((rc = r2s_int::rbf_smooth_slabs(slabs, grid, o.rbf_interp, o.rbf_smooth, o.rbf_kernel_threshold, ri.V_frac * ri.V_domain,
                                                fine_sdf_out, &ri.level_shift, &its)))
                return rc;
            ri.cg_iters = its;
        }
        const double t6m = now_ms();
        ri.ms_rbf = t6m - t5m;
        ri.ms_download = t5m - t4m;
        ri.ms_total = t6m - t_start;
        HIP_TRY(hipSetDevice(dev0));
        if (info) *info = ri;
        return 0;
    }
    double t3 = now_ms();
    ri.ms_sdf = t3 - t2;
    ri.ms_sdf_kernels = st.ms_prep + st.ms_bins + st.ms_main + st.ms_gather;
    const bool pin_d = sdf_dists_out && is_pinned(sdf_dists_out);
    if (sdf_raw_out) {   // the field before artifact removal (export_analysis, RhoToSDF.jl:181-189)
        std::vector<Segment> segs{{(char*)sdf_raw_out, (const char*)S->out[2].p, sizeof(double) * (size_t)ngp}};
        if ((rc = download(S, segs, is_pinned(sdf_raw_out)))) return rc;
    }
    // ---- artifact removal (:174-208) ----
    if (o.remove_artifacts) {
        if ((rc = r2s_int::remove_artifacts_dev(S->out[2].as<double>(), grid, 0.0, o.artifact_min_component_ratio, nullptr,
                                                &ri.n_flipped)))
            return rc;
    }
    double t4 = now_ms();
    ri.ms_artifacts = t4 - t3;
    // The cleaned field travels to the host while the smoothing runs, and the chunks of the smoothed field follow as the
    // last kernel of the smoothing finishes them (r2s_int::rbf_smooth_dev, fine_chunk): one helper thread with the copy
    // stream and the staging buffers of the session (the smoothing uses the default stream).  Pinned destinations get
    // plain DMA, pageable ones the staged copy.
    bool dists_in_flight = false;
    std::thread dl_thread;
    int dl_rc = 0;
    std::string dl_err;
    struct Chunk { int64_t t0, t1; hipEvent_t ev; };
    std::mutex qmu;
    std::condition_variable qcv;
    std::vector<Chunk> queue;   // chunks of the smoothed field whose kernels have been launched
    std::vector<std::pair<int64_t, int64_t>> landed;   // early: chunks that have arrived in pinned memory (guarded by qmu)
    bool q_done = false;
    size_t nfine = 0;
    if (!o.skip_rbf) {
        nfine = 1;
        for (int i = 0; i < 3; ++i) nfine *= (size_t)(grid->N[i] * o.rbf_smooth + 1);
    }
    const bool pin_f = fine_sdf_out && is_pinned(fine_sdf_out);
    // The smoothed field is evaluated BEFORE the level bisection and travels while the level is found (it needs the weights
    // only); the host threads add the level shift afterwards.  Needs pinned memory for the field without its shift.
    static const bool early_env = !(getenv("R2S_FINE_EARLY") && atoi(getenv("R2S_FINE_EARLY")) == 0);
    bool early = early_env && !o.skip_rbf && fine_sdf_out && nfine >= ((size_t)1 << 22);
    if (early && !pin_f && S->fine_host_cap < sizeof(float) * nfine) {
        if (S->fine_host) r2s_host_free(S->fine_host);
        S->fine_host = nullptr;
        S->fine_host_cap = 0;
        if ((S->fine_host = r2s_host_alloc(sizeof(float) * nfine))) S->fine_host_cap = sizeof(float) * nfine;
        else { (void)hipGetLastError(); early = false; }
    }
    if (sdf_dists_out && pin_d) {
        HIP_TRY(hipMemcpyAsync(sdf_dists_out, S->out[2].p, sizeof(double) * (size_t)ngp, hipMemcpyDeviceToHost, S->cs));
        dists_in_flight = true;
    }
    if (!o.skip_rbf) {
        const bool dists_here = sdf_dists_out && !pin_d;
        dl_thread = std::thread([&, dists_here]() {
            if (hipSetDevice(dev0) != hipSuccess) { dl_rc = R2S_ERR_HIP; dl_err = "hipSetDevice failed"; return; }
            if (dists_here) {
                std::vector<Segment> segs{{(char*)sdf_dists_out, (const char*)S->out[2].p, sizeof(double) * (size_t)ngp}};
                dl_rc = download(S, segs, false);
                if (dl_rc) { dl_err = g_err; return; }
            }
            size_t next = 0;
            for (;;) {
                Chunk c;
                {
                    std::unique_lock<std::mutex> lk(qmu);
                    qcv.wait(lk, [&] { return next < queue.size() || q_done; });
                    if (next >= queue.size()) return;
                    c = queue[next++];
                }
                if (hipEventSynchronize(c.ev) != hipSuccess) { dl_rc = R2S_ERR_HIP; dl_err = "hipEventSynchronize failed"; return; }
                // (early: the chunk lands in pinned memory - the caller's array if it is pinned, else the landing zone - and gets its
                //  level shift from the host threads once the level is known)
                float* land = early ? (pin_f ? fine_sdf_out : (float*)S->fine_host) : fine_sdf_out;
                std::vector<Segment> segs{{(char*)(land + c.t0), (const char*)(S->fine.as<float>() + c.t0), sizeof(float) * (size_t)(c.t1 - c.t0)}};
                dl_rc = download(S, segs, early ? true : pin_f);
                if (dl_rc) { dl_err = g_err; return; }
                if (early) {
                    {
                        std::lock_guard<std::mutex> lk(qmu);
                        landed.push_back({c.t0, c.t1});
                    }
                    qcv.notify_all();
                }
            }
        });
        if (dists_here) dists_in_flight = true;
    }
    auto join_dl = [&]() -> int {
        {
            std::lock_guard<std::mutex> lk(qmu);
            q_done = true;
        }
        qcv.notify_all();
        if (dl_thread.joinable()) dl_thread.join();
        return dl_rc ? fail(dl_rc, "%s", dl_err.c_str()) : 0;
    };
    // ---- RBF smoothing (:222-224) ----
    if (!o.skip_rbf) {
        if (S->fine.ensure_exact(sizeof(float) * nfine)) { (void)join_dl(); return fail(R2S_ERR_NOMEM, "hipMalloc of the fine grid failed"); }
        int its = 0;
        if (S->fine.p == nullptr) { (void)join_dl(); return fail(R2S_ERR_NOMEM, "hipMalloc of the fine grid failed"); }
        if (!S->rbf_ws) S->rbf_ws = r2s_int::rbf_workspace_create();
        size_t n_ev = 0;
        const std::function<int(int64_t, int64_t)> forward = [&](int64_t t0, int64_t t1) -> int {
            if (n_ev >= S->evf.size()) {
                hipEvent_t e = nullptr;
                if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return fail(R2S_ERR_HIP, "hipEventCreate failed");
                S->evf.push_back(e);
            }
            hipEvent_t e = S->evf[n_ev++];
            if (hipEventRecord(e, nullptr) != hipSuccess) return fail(R2S_ERR_HIP, "hipEventRecord failed");   // (the smoothing's stream)
            {
                std::lock_guard<std::mutex> lk(qmu);
                queue.push_back({t0, t1, e});
            }
            qcv.notify_all();
            return 0;
        };
        rc = r2s_int::rbf_smooth_dev(S->out[2].as<double>(), grid, o.rbf_interp, o.rbf_smooth, o.rbf_kernel_threshold,
                                     ri.V_frac * ri.V_domain, S->fine.as<float>(), &ri.level_shift, &its, &forward, S->rbf_ws, early);
        if (rc) { (void)join_dl(); return rc; }
        ri.cg_iters = its;
    }
    double t5 = now_ms();
    ri.ms_rbf = t5 - t4;
    // ---- rest of the results to the caller ----
    if (early) {   // the level shift: fine = raw + th (Float32), by the host threads, chunk by chunk as the chunks arrive
        ensure_pool(S);
        const float th = ri.level_shift;
        const float* src = pin_f ? fine_sdf_out : (const float*)S->fine_host;
        float* dst = fine_sdf_out;
        {
            std::lock_guard<std::mutex> lk(qmu);
            q_done = true;   // (no further chunk will be queued: the smoothing has returned)
        }
        qcv.notify_all();
        size_t next = 0, total = 0;
        {
            std::lock_guard<std::mutex> lk(qmu);
            total = queue.size();
        }
        while (next < total) {
            std::pair<int64_t, int64_t> ch;
            {
                std::unique_lock<std::mutex> lk(qmu);
                // (polling wait: the download thread sets dl_rc and leaves without a signal when a copy fails)
                while (!qcv.wait_for(lk, std::chrono::milliseconds(20), [&] { return next < landed.size() || dl_rc != 0; })) {}
                if (dl_rc) break;
                ch = landed[next++];
            }
            const size_t c0 = (size_t)ch.first, cn = (size_t)(ch.second - ch.first);
            S->pool->start([=](int id, int n) {
                // (16-byte pieces where both sides are aligned; streaming stores into another array: no read-for-ownership)
                typedef float v4f __attribute__((ext_vector_type(4)));
                const float* s0 = src + c0;
                float* d0 = dst + c0;
                const size_t a = cn * (size_t)id / (size_t)n, b = cn * (size_t)(id + 1) / (size_t)n;
                size_t q = a;
                if (s0 != d0) {
                    while (q < b && (((uintptr_t)(d0 + q)) & 15u)) { d0[q] = s0[q] + th; ++q; }
                    if ((((uintptr_t)(s0 + q)) & 15u) == 0)
                        for (; q + 4 <= b; q += 4)
                            __builtin_nontemporal_store(*reinterpret_cast<const v4f*>(s0 + q) + th, reinterpret_cast<v4f*>(d0 + q));
                }
                for (; q < b; ++q) d0[q] = s0[q] + th;
                std::atomic_thread_fence(std::memory_order_seq_cst);
            });
            S->pool->wait();
        }
    }
    if ((rc = join_dl())) return rc;
    if (sdf_dists_out && !dists_in_flight) {
        std::vector<Segment> segs{{(char*)sdf_dists_out, (const char*)S->out[2].p, sizeof(double) * (size_t)ngp}};
        if ((rc = download(S, segs, false))) return rc;
    }
    if (dists_in_flight) HIP_TRY(hipStreamSynchronize(S->cs));
    const double t6 = now_ms();
    ri.ms_download = t6 - t5;
    ri.ms_total = t6 - t_start;
    if (info) *info = ri;
    return 0;
}

}  // extern "C"
