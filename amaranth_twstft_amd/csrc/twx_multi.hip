// twx_multi.hip — ONE host process, N devices: the multi-GPU driver behind the C ABI (twx_multi_* in include/twstft_hip.h).
//
// The reference's own shape for more than one correlation at a time is threads inside one process (one GoRanging worker per
// channel with a reader hand-off, processing/CPP/main.cpp:180-187,488-497) and side-by-side jobs (acquisition/goprocess.sh:9-11).
// Here: one correlator context and one persistent host thread per device; the windows of a capture are cut into contiguous
// blocks (one contiguous file extent per device, sizes differing by at most one — the rule of amaranth_twstft_amd/dist.py),
// every context runs twx_process_file / twx_process_windows[_dev] on its block, and the fixed-size result records are
// exchanged with ONE ncclAllGather over xGMI (RCCL, communicators from ncclCommInitAll — the single-process form) issued
// for all devices inside one ncclGroupStart/End.  A device list that names a device twice cannot form an RCCL communicator
// (one rank per device): the blocks are then concatenated on the host, which is also what lets a one-GPU box test the
// threading and the ordering.  RCCL is bound at run time (dlopen of librccl.so.1 on the first twx_multi_create that needs
// it): a 570-MB library that no single-GPU MEX call should have to map.  No process is ever re-executed; threads only.
//
// The exchange is 240 bytes per window: it must never be what loses a job.  When RCCL cannot be loaded, ncclCommInitAll fails or
// does not come back within TWX_RCCL_INIT_TIMEOUT_S (default 120 s), or a collective fails or does not complete within
// TWX_RCCL_GATHER_TIMEOUT_S (default 60 s), the driver says so in twx_multi_info (rccl_fallback, rccl_error) and carries on with
// the host-side concatenation — same records, same order.  Worker threads are bound to the CPUs of their device's NUMA node
// (twx_affinity.h; TWX_NO_PIN=1 turns that off).  TWX_MULTI_INJECT={init,init_hang,gather,gather_timeout} makes the corresponding
// RCCL step fail on purpose (tests/test_gpu_multi.py: a one-GPU box has no other way to reach these branches).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <string.h>
#include <sys/stat.h>
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>
#include "twx_internal.h"
#include "twx_workers.h"
#include "twx_affinity.h"

// The few RCCL declarations this file uses, restated from rccl/rccl.h (values of RCCL 2.x, unchanged since NCCL 2.0): the library
// is bound at run time, and a build box for single-GPU users need not carry the RCCL development headers.
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4,
               ncclInvalidUsage = 5, ncclRemoteError = 6, ncclInProgress = 7 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0 } ncclDataType_t;
}

namespace {

thread_local std::string g_multi_create_err;

double env_seconds(const char* name, double dflt) {
    const char* e = getenv(name);
    if (!e || !*e) return dflt;
    char* end = nullptr;
    const double v = strtod(e, &end);
    return (end != e && v > 0) ? v : dflt;
}
bool inject(const char* what) { const char* e = getenv("TWX_MULTI_INJECT"); return e && strcmp(e, what) == 0; }

// ---- RCCL, bound at run time --------------------------------------------------------------------------------------
struct Rccl {
    void* h = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    const char* (*GetLastError)(ncclComm_t) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    std::string err;
    bool load() {
        if (h) return true;
        const char* names[] = {getenv("TWX_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* nm : names) {
            if (!nm || !*nm) continue;
            h = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
            if (h) break;
            err = dlerror();
        }
        if (!h) { err = "cannot load RCCL (librccl.so.1): " + err; return false; }
        auto sym = [&](const char* n) { void* p = dlsym(h, n); if (!p) err = std::string("RCCL symbol missing: ") + n; return p; };
        CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        CommAbort = (decltype(CommAbort))dlsym(h, "ncclCommAbort");                 // optional: only the time-out path wants it
        AllGather = (decltype(AllGather))sym("ncclAllGather");
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        GetLastError = (decltype(GetLastError))sym("ncclGetLastError");
        GetVersion = (decltype(GetVersion))sym("ncclGetVersion");
        if (!CommInitAll || !CommDestroy || !AllGather || !GroupStart || !GroupEnd || !GetErrorString) { dlclose(h); h = nullptr; return false; }
        return true;
    }
};
Rccl& rccl() { static Rccl r; return r; }
std::mutex& rccl_mu() { static std::mutex m; return m; }
// An ncclCommInitAll that never returned leaves a thread of this process inside RCCL's bootstrap: a second one beside it would most
// likely hang the same way (another 120 s) and share state with the abandoned one.  From then on every twx_multi_create of the
// process skips RCCL at once, with the same reason (guarded by rccl_mu).
std::string& rccl_unusable() { static std::string why; return why; }

using twx::Worker;               // one persistent host thread per context (twx_workers.h: plain C++, run under -fsanitize=thread on the CPU)

}  // namespace

struct twx_multi {
    int n = 0;
    std::vector<int> devices;
    std::vector<twx_ctx*> ctx;
    std::vector<Worker*> workers;
    bool distinct = true, use_rccl = false;
    std::vector<ncclComm_t> comms;
    std::vector<hipStream_t> gstream;              // the gather's own stream on every device
    std::vector<void*> send_dev, recv_dev; size_t cap_records = 0;      // per device: one block / all blocks of the gather
    std::vector<std::vector<twx_result>> local;    // per context: the records of its block (host paths)
    std::vector<int> pinned;                       // per worker: 1 = bound to its device's NUMA CPUs
    std::string err;
    twx_multi_info info{};
    twx_info cinfo{};

    int fail(int code, const std::string& m) { err = m; return code; }
    ~twx_multi() {
        for (auto w : workers) if (w) { w->stop(); delete w; }
        if (use_rccl) for (int r = 0; r < (int)comms.size(); ++r) if (comms[r]) { (void)hipSetDevice(devices[r]); (void)rccl().CommDestroy(comms[r]); }
        for (int r = 0; r < n; ++r) {
            (void)hipSetDevice(devices[r]);
            if (r < (int)send_dev.size() && send_dev[r]) (void)hipFree(send_dev[r]);
            if (r < (int)recv_dev.size() && recv_dev[r]) (void)hipFree(recv_dev[r]);
            if (r < (int)gstream.size() && gstream[r]) (void)hipStreamDestroy(gstream[r]);
            if (r < (int)ctx.size() && ctx[r]) twx_destroy(ctx[r]);
        }
    }
    // every context runs f(rank) on its own thread; first failure wins (its context's message is kept).  The jobs hold pointers
    // into the caller's frame: whatever happens while they are being handed out (bad_alloc from a std::function copy), every job
    // already submitted is waited for before this frame is left.
    int run_all(const std::function<int(int)>& f) {
        int submitted = 0;
        try {
            for (int r = 0; r < n; ++r) { workers[r]->submit([&f, r]() { return f(r); }); ++submitted; }
        } catch (...) {
            for (int r = 0; r < submitted; ++r) (void)workers[r]->wait();
            throw;
        }
        int rc = TWX_OK;
        for (int r = 0; r < n; ++r) {
            const int rr = workers[r]->wait();
            if (rr && !rc) {
                rc = rr;
                const char* m = twx_last_error(ctx[r]);
                err = "context " + std::to_string(r) + " (device " + std::to_string(devices[r]) + "): " + (m && *m ? m : twx_strerror(rr));
            }
        }
        return rc;
    }
    static void shard(long long total, int r, int world, long long* start, long long* count) {      // dist.shard_windows
        const long long base = total / world, rem = total % world;
        *start = r * base + std::min<long long>(r, rem);
        *count = base + (r < rem ? 1 : 0);
    }
    int ensure_gather(size_t records) {
        if (records <= cap_records) return TWX_OK;
        const size_t cap = records + records / 4 + 16;
        for (int r = 0; r < n; ++r) {
            if (hipSetDevice(devices[r]) != hipSuccess) return fail(TWX_E_HIP, "hipSetDevice failed");
            if (send_dev[r]) { (void)hipFree(send_dev[r]); send_dev[r] = nullptr; }
            if (recv_dev[r]) { (void)hipFree(recv_dev[r]); recv_dev[r] = nullptr; }
            if (hipMalloc(&send_dev[r], cap * sizeof(twx_result)) != hipSuccess || hipMalloc(&recv_dev[r], cap * sizeof(twx_result) * n) != hipSuccess)
                return fail(TWX_E_NOMEM, "gather buffer allocation failed");
        }
        cap_records = cap;
        return TWX_OK;
    }
    // RCCL is given up for the rest of this object's life: the exchange continues as host-side concatenation (flagged, never silent)
    void give_up_rccl(int why, const std::string& text, bool abort_comms) {
        Rccl& R = rccl();
        int leaked = 0;
        for (int r = 0; r < (int)comms.size(); ++r) {
            if (!comms[r]) continue;
            (void)hipSetDevice(devices[r]);
            // a communicator with a collective still in flight is aborted, never destroyed (ncclCommDestroy would wait for it);
            // without ncclCommAbort it is left alone
            if (abort_comms) { if (R.CommAbort) (void)R.CommAbort(comms[r]); else ++leaked; }
            else (void)R.CommDestroy(comms[r]);
            comms[r] = nullptr;
        }
        comms.clear();
        use_rccl = false;
        info.rccl = 0; info.rccl_fallback = why;
        std::string t = text;
        if (leaked) t += " (" + std::to_string(leaked) + " communicators with a collective in flight left alone: this RCCL has no ncclCommAbort)";
        snprintf(info.rccl_error, sizeof(info.rccl_error), "%s", t.c_str());
    }
    // send_dev[r] holds `block` records on every device -> recv_dev[r] holds n*block on every device (one collective).
    // false: RCCL failed or timed out and has been given up (give_up_rccl); the caller continues on the host path.
    bool all_gather(size_t block) {
        Rccl& R = rccl();
        const auto t0 = std::chrono::steady_clock::now();
        auto text = [&](ncclResult_t r, const char* what, ncclComm_t c) {
            std::string m = std::string(what) + " failed: " + (R.GetErrorString ? R.GetErrorString(r) : "?");
            if (R.GetLastError && c) { const char* le = R.GetLastError(c); if (le && *le) m += std::string(" - ") + le; }
            return m;
        };
        if (inject("gather")) { give_up_rccl(2, "ncclAllGather failed: injected failure (TWX_MULTI_INJECT=gather)", false); return false; }
        const bool fake_hang = inject("gather_timeout");
        if (!fake_hang) {
            ncclResult_t g = R.GroupStart();
            if (g != ncclSuccess) { give_up_rccl(2, text(g, "ncclGroupStart", nullptr), false); return false; }
            for (int r = 0; r < n; ++r) {
                (void)hipSetDevice(devices[r]);
                ncclResult_t e = R.AllGather(send_dev[r], recv_dev[r], block * sizeof(twx_result), ncclChar, comms[r], gstream[r]);
                if (e != ncclSuccess) { (void)R.GroupEnd(); give_up_rccl(2, text(e, "ncclAllGather", comms[r]), true); return false; }
            }
            g = R.GroupEnd();
            if (g != ncclSuccess) { give_up_rccl(2, text(g, "ncclGroupEnd", comms[0]), true); return false; }
        }
        // completion with a deadline: a peer that never arrives must not hang the host program
        const double limit = fake_hang ? 0.2 : env_seconds("TWX_RCCL_GATHER_TIMEOUT_S", 60.0);
        for (int r = 0; r < n; ++r) {
            (void)hipSetDevice(devices[r]);
            for (;;) {
                const hipError_t q = fake_hang ? hipErrorNotReady : hipStreamQuery(gstream[r]);
                if (q == hipSuccess) break;
                if (q != hipErrorNotReady) { (void)hipGetLastError(); give_up_rccl(2, std::string("gather stream failed: ") + hipGetErrorString(q), true); return false; }
                (void)hipGetLastError();
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) {
                    char b[160];
                    snprintf(b, sizeof(b), "ncclAllGather did not complete on context %d within %.1f s%s", r, limit, fake_hang ? " (injected: TWX_MULTI_INJECT=gather_timeout)" : "");
                    give_up_rccl(2, b, true);
                    return false;
                }
                std::this_thread::sleep_for(std::chrono::microseconds(50));
            }
        }
        info.gather_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        info.records_gathered = (int64_t)(block * n); info.bytes_per_rank = (int64_t)(block * sizeof(twx_result));
        return true;
    }
    // send_dev[r] (block records each, complete) -> recv_dev[r] on every device, and `out` on the host: all n*block records in
    // context order, or — counts given — only the first counts[r]*per of every block, packed (a sharded recording: window order)
    int exchange_dev(size_t block, const std::vector<long long>* counts, twx_result* out, int per = 1) {
        auto pack = [&](const twx_result* all) {
            if (!out) return;
            if (!counts) { memcpy(out, all, block * n * sizeof(twx_result)); return; }
            size_t o = 0;
            for (int r = 0; r < n; ++r) { const size_t k = (size_t)(*counts)[r] * per; memcpy(out + o, all + (size_t)r * block, k * sizeof(twx_result)); o += k; }
        };
        if (use_rccl && all_gather(block)) {
            if (out) {
                (void)hipSetDevice(devices[0]);
                if (!counts) {
                    if (hipMemcpy(out, recv_dev[0], block * n * sizeof(twx_result), hipMemcpyDeviceToHost) != hipSuccess) return fail(TWX_E_HIP, "gathered records D2H failed");
                } else {
                    std::vector<twx_result> all(block * n);
                    if (hipMemcpy(all.data(), recv_dev[0], all.size() * sizeof(twx_result), hipMemcpyDeviceToHost) != hipSuccess) return fail(TWX_E_HIP, "gathered records D2H failed");
                    pack(all.data());
                }
            }
            return TWX_OK;
        }
        // host-side concatenation (repeated devices, TWX_MULTI_NO_RCCL, or RCCL given up — possibly just now, by all_gather);
        // every context's "gathered" buffer is filled from it so that both modes leave the same state
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<twx_result> all(block * n);
        for (int r = 0; r < n; ++r) {
            (void)hipSetDevice(devices[r]);
            if (hipMemcpy(all.data() + (size_t)r * block, send_dev[r], block * sizeof(twx_result), hipMemcpyDeviceToHost) != hipSuccess) return fail(TWX_E_HIP, "record D2H failed");
        }
        for (int r = 0; r < n; ++r) {
            (void)hipSetDevice(devices[r]);
            if (hipMemcpy(recv_dev[r], all.data(), all.size() * sizeof(twx_result), hipMemcpyHostToDevice) != hipSuccess) return fail(TWX_E_HIP, "record H2D failed");
        }
        pack(all.data());
        info.gather_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();     // the exchange as it ran: copies through the host
        info.records_gathered = (int64_t)all.size(); info.bytes_per_rank = (int64_t)(block * sizeof(twx_result));
        return TWX_OK;
    }
    // the blocks local[r][0 .. counts[r]*per) -> out (blocks in rank order, no padding); through the devices when RCCL is on
    int gather_host_blocks(const std::vector<long long>& counts, int per, twx_result* out) {
        long long mx = 0;
        for (long long c : counts) mx = std::max(mx, c);
        const size_t block = (size_t)mx * per;
        bool done = false;
        if (use_rccl && block > 0) {
            if (int rc = ensure_gather(block)) return rc;
            for (int r = 0; r < n; ++r) {
                (void)hipSetDevice(devices[r]);
                if (hipMemsetAsync(send_dev[r], 0, block * sizeof(twx_result), gstream[r]) != hipSuccess ||
                    hipMemcpyAsync(send_dev[r], local[r].data(), (size_t)counts[r] * per * sizeof(twx_result), hipMemcpyHostToDevice, gstream[r]) != hipSuccess)
                    return fail(TWX_E_HIP, "record upload failed");
            }
            if (all_gather(block)) {
                std::vector<twx_result> all(block * n);
                (void)hipSetDevice(devices[0]);
                if (hipMemcpy(all.data(), recv_dev[0], all.size() * sizeof(twx_result), hipMemcpyDeviceToHost) != hipSuccess) return fail(TWX_E_HIP, "gathered records D2H failed");
                size_t o = 0;
                for (int r = 0; r < n; ++r) { memcpy(out + o, all.data() + (size_t)r * block, (size_t)counts[r] * per * sizeof(twx_result)); o += (size_t)counts[r] * per; }
                done = true;
            }
        }
        if (!done) {                       // host-side concatenation (repeated devices, TWX_MULTI_NO_RCCL, or RCCL given up just now)
            size_t o = 0;
            for (int r = 0; r < n; ++r) { memcpy(out + o, local[r].data(), (size_t)counts[r] * per * sizeof(twx_result)); o += (size_t)counts[r] * per; }
            info.gather_ms = 0; info.records_gathered = (int64_t)o; info.bytes_per_rank = (int64_t)(block * sizeof(twx_result));
        }
        return TWX_OK;
    }
};

template <class F> static int multi_guard(twx_multi* m, F f) noexcept {
    // the launches of this library are checked with hipGetLastError(): an error another library left behind on this thread
    // (RCCL and PyTorch probe pointers and peers and do not clear what those probes set) must not be taken for ours
    (void)hipGetLastError();
    try { return f(); }
    catch (const std::bad_alloc&) { return m ? m->fail(TWX_E_NOMEM, "out of host memory") : TWX_E_NOMEM; }
    catch (const std::exception& e) { return m ? m->fail(TWX_E_STATE, std::string("internal error: ") + e.what()) : TWX_E_STATE; }
    catch (...) { return m ? m->fail(TWX_E_STATE, "internal error") : TWX_E_STATE; }
}

extern "C" {

const char* twx_multi_last_error(const twx_multi* m) { return m ? m->err.c_str() : g_multi_create_err.c_str(); }

static int multi_create_impl(const twx_config* cfg, const int32_t* devices, int32_t n, int32_t flags, twx_multi** out) {
    if (!cfg || !out || n < 1 || n > 64) { g_multi_create_err = "bad argument (1..64 contexts)"; return TWX_E_ARG; }
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { g_multi_create_err = "no HIP device available (the HIP path has no CPU fallback)"; return TWX_E_HIP; }
    std::unique_ptr<twx_multi> m(new twx_multi());
    m->n = n;
    for (int r = 0; r < n; ++r) {
        const int d = devices ? devices[r] : r % ndev;
        if (d < 0 || d >= ndev) { g_multi_create_err = "device " + std::to_string(d) + " of the list does not exist (" + std::to_string(ndev) + " visible)"; return TWX_E_ARG; }
        for (int q = 0; q < r; ++q) if (m->devices[q] == d) m->distinct = false;
        m->devices.push_back(d);
    }
    m->ctx.assign(n, nullptr); m->send_dev.assign(n, nullptr); m->recv_dev.assign(n, nullptr); m->gstream.assign(n, nullptr);
    m->local.resize(n);
    // one persistent thread per context, bound to the CPUs next to its device (pinned staging buffers are first touched, and the
    // capture is read, by this thread): /sys/bus/pci/devices/<bus id>/{numa_node,local_cpulist}
    const char* np_env = getenv("TWX_NO_PIN");
    const bool pin = !(np_env && atoi(np_env) != 0);
    m->pinned.assign(n, 0);               // written by the workers' prologues: lives in the object, which outlives its threads
    for (int r = 0; r < n; ++r) {
        const int d = m->devices[r];
        char bus[64] = {0};
        twx::DeviceAffinity aff;
        if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), d) == hipSuccess) aff = twx::affinity_of_pci(bus);
        else (void)hipGetLastError();
        if (r < 64) m->info.numa_node[r] = aff.numa_node;
        Worker* w = new Worker(); m->workers.push_back(w);
        int* flag = &m->pinned[r];
        std::vector<int> cpus = (pin && aff.numa_node >= 0) ? aff.cpus : std::vector<int>();
        w->start([d, cpus, flag]() { (void)hipSetDevice(d); *flag = twx::pin_current_thread(cpus) > 0 ? 1 : 0; }, TWX_E_NOMEM, TWX_E_STATE);
    }
    for (int r = n; r < 64; ++r) m->info.numa_node[r] = -1;
    // contexts are created side by side (tables, code spectrum, 3 pipeline slots each): one thread per device
    std::vector<std::string> cerr(n);
    int rc = TWX_OK;
    {
        twx_multi* mp = m.get();
        for (int r = 0; r < n; ++r) m->workers[r]->submit([mp, cfg, r, &cerr]() {
            twx_config c = *cfg;
            c.device = mp->devices[r];
            const int e = twx_create(&c, &mp->ctx[r]);
            if (e) cerr[r] = twx_last_error(nullptr);
            return e;
        });
        for (int r = 0; r < n; ++r) { const int e = m->workers[r]->wait(); if (e && !rc) { rc = e; g_multi_create_err = "context " + std::to_string(r) + ": " + cerr[r]; } }
    }
    if (rc) return rc;
    twx_get_info(m->ctx[0], &m->cinfo);
    for (int r = 0; r < n; ++r) {
        if (hipSetDevice(m->devices[r]) != hipSuccess || hipStreamCreateWithFlags(&m->gstream[r], hipStreamNonBlocking) != hipSuccess) {
            g_multi_create_err = "gather stream creation failed"; return TWX_E_HIP;
        }
    }
    // RCCL: one rank per device — only a list of distinct devices can form a communicator.  n = 1 needs no exchange at all;
    // TWX_MULTI_RCCL_ONE (flag or env TWX_MULTI_FORCE_RCCL=1) builds the world of one anyway, so that a one-GPU box runs the same calls.
    const char* fe = getenv("TWX_MULTI_FORCE_RCCL");
    const bool force_one = (flags & TWX_MULTI_RCCL_ONE) || (fe && atoi(fe) != 0);
    const bool no_rccl = (flags & TWX_MULTI_NO_RCCL) != 0;
    if (m->distinct && !no_rccl && (n > 1 || force_one)) {
        std::lock_guard<std::mutex> g(rccl_mu());
        Rccl& R = rccl();
        std::string why;
        if (!rccl_unusable().empty()) why = rccl_unusable() + " (an earlier twx_multi_create of this process; RCCL is not tried again)";
        else if (!R.load()) why = R.err;
        else if (inject("init")) why = "ncclCommInitAll failed: injected failure (TWX_MULTI_INJECT=init)";
        else {
            // ncclCommInitAll on its own thread with a deadline: a bootstrap that never returns (a peer's IPC handle that cannot be
            // opened, a dead link) must not hang the host program.  The thread is abandoned on a time-out — it holds only the
            // shared state below — and RCCL is not touched again by this object.
            struct Init { std::mutex mu; std::condition_variable cv; bool done = false; ncclResult_t rc = ncclSuccess; std::vector<ncclComm_t> comms; std::vector<int> devs; std::string last; };
            auto st = std::make_shared<Init>();
            st->comms.assign(n, nullptr); st->devs = m->devices;
            const bool hang = inject("init_hang");
            std::thread([st, &R, hang]() {
                ncclResult_t rc = ncclSuccess;
                if (hang) std::this_thread::sleep_for(std::chrono::seconds(3600));
                else rc = R.CommInitAll(st->comms.data(), (int)st->devs.size(), st->devs.data());
                std::string last;
                if (rc != ncclSuccess && R.GetLastError) { const char* le = R.GetLastError(nullptr); if (le && *le) last = le; }
                std::lock_guard<std::mutex> lk(st->mu);
                st->rc = rc; st->last = last; st->done = true;
                st->cv.notify_all();
            }).detach();
            const double limit = hang ? 0.3 : env_seconds("TWX_RCCL_INIT_TIMEOUT_S", 120.0);
            std::unique_lock<std::mutex> lk(st->mu);
            if (!st->cv.wait_for(lk, std::chrono::duration<double>(limit), [&]() { return st->done; })) {
                char b[160];
                snprintf(b, sizeof(b), "ncclCommInitAll did not return within %.1f s%s", limit, hang ? " (injected: TWX_MULTI_INJECT=init_hang)" : "");
                why = b;
                if (!hang) rccl_unusable() = why;           // (the injected hang is a sleeping thread, not RCCL state: later creates may try)
            } else if (st->rc != ncclSuccess) {
                why = std::string("ncclCommInitAll failed: ") + R.GetErrorString(st->rc);
                if (!st->last.empty()) why += " - " + st->last;
            } else {
                m->comms = st->comms;
                m->use_rccl = true;
                int v = 0;
                if (R.GetVersion && R.GetVersion(&v) == ncclSuccess) m->info.rccl_version = v;
            }
        }
        (void)hipGetLastError();
        if (!m->use_rccl) {                 // the job goes on without RCCL: host-side concatenation, flagged
            m->info.rccl_fallback = 1;
            snprintf(m->info.rccl_error, sizeof(m->info.rccl_error), "%s", why.c_str());
        }
    }
    m->info.threads_pinned = 0;
    for (int r = 0; r < n; ++r) { (void)m->workers[r]->wait(); m->info.threads_pinned += m->pinned[r]; }     // (the prologues ran before the creation jobs)
    m->info.n_contexts = n; m->info.rccl = m->use_rccl ? 1 : 0;
    {
        std::vector<int> u = m->devices; std::sort(u.begin(), u.end());
        m->info.n_devices_distinct = (int32_t)(std::unique(u.begin(), u.end()) - u.begin());
    }
    *out = m.release();
    return TWX_OK;
}
int twx_multi_create(const twx_config* cfg, const int32_t* devices, int32_t n_devices, int32_t flags, twx_multi** out) {
    try { return multi_create_impl(cfg, devices, n_devices, flags, out); }
    catch (const std::bad_alloc&) { if (out) *out = nullptr; g_multi_create_err = "out of host memory"; return TWX_E_NOMEM; }
    catch (...) { if (out) *out = nullptr; g_multi_create_err = "internal error"; return TWX_E_STATE; }
}
void twx_multi_destroy(twx_multi* m) { delete m; }
int twx_multi_get_info(const twx_multi* m, twx_multi_info* info) { if (!m || !info) return TWX_E_ARG; *info = m->info; return TWX_OK; }
twx_ctx* twx_multi_context(twx_multi* m, int32_t i) { return (m && i >= 0 && i < m->n) ? m->ctx[i] : nullptr; }

int twx_multi_process_file(twx_multi* m, const char* path, int32_t n_channels, int32_t channel, int64_t skip_samples, const twx_band* band,
                           double df_const, twx_result* out, int64_t max_windows, int64_t* n_done) {
    if (!m) return TWX_E_ARG;
    if (!path || !out || !n_done || max_windows < 0 || n_channels < 1 || channel < TWX_ALL_CHANNELS || channel >= n_channels || skip_samples < 0) return m->fail(TWX_E_ARG, "bad argument");
    *n_done = 0;
    return multi_guard(m, [&]() -> int {
        struct stat sb;
        if (stat(path, &sb) != 0) return m->fail(TWX_E_ARG, std::string("cannot open ") + path);
        const long long N = m->cinfo.n, per = channel < 0 ? n_channels : 1;
        const long long have = (long long)sb.st_size / (4ll * n_channels) - skip_samples;
        const long long total = std::max<long long>(0, std::min<long long>(max_windows, have / N));     // a short final window ends the loop
        std::vector<long long> start(m->n), count(m->n), done(m->n, 0);
        for (int r = 0; r < m->n; ++r) { twx_multi::shard(total, r, m->n, &start[r], &count[r]); m->local[r].resize((size_t)std::max<long long>(1, count[r] * per)); }
        int rc = m->run_all([&](int r) -> int {
            if (count[r] == 0) return TWX_OK;
            int64_t nd = 0;
            const int e = twx_process_file(m->ctx[r], path, n_channels, channel, skip_samples + start[r] * N, band, df_const, m->local[r].data(), count[r], &nd);
            done[r] = nd;
            return e;
        });
        if (rc) return rc;
        for (int r = 0; r < m->n; ++r)
            if (done[r] != count[r]) return m->fail(TWX_E_STATE, "context " + std::to_string(r) + " read fewer windows than the file held when the job was cut (capture truncated meanwhile?)");
        rc = m->gather_host_blocks(count, (int)per, out);
        if (rc == TWX_OK) *n_done = total;
        return rc;
    });
}

int twx_multi_process_windows(twx_multi* m, const int16_t* iq, int64_t n_windows, int32_t n_channels, int32_t channel, const twx_band* band,
                              const double* df, twx_result* out) {
    if (!m) return TWX_E_ARG;
    if (!iq || !out || n_windows < 0 || n_channels < 1 || channel < TWX_ALL_CHANNELS || channel >= n_channels || (!band && !df)) return m->fail(TWX_E_ARG, "bad argument");
    if (n_windows == 0) return TWX_OK;
    return multi_guard(m, [&]() -> int {
        const long long N = m->cinfo.n, per = channel < 0 ? n_channels : 1;
        std::vector<long long> start(m->n), count(m->n);
        for (int r = 0; r < m->n; ++r) { twx_multi::shard(n_windows, r, m->n, &start[r], &count[r]); m->local[r].resize((size_t)std::max<long long>(1, count[r] * per)); }
        int rc = m->run_all([&](int r) -> int {
            if (count[r] == 0) return TWX_OK;
            return twx_process_windows(m->ctx[r], iq + (size_t)start[r] * (size_t)N * n_channels * 2, count[r], n_channels, channel, band,
                                       df ? df + start[r] * per : nullptr, m->local[r].data());
        });
        if (rc) return rc;
        return m->gather_host_blocks(count, (int)per, out);
    });
}

// Device-resident form (what bench.py --single-process times): context r processes n_windows windows of ITS OWN recording
// iq_dev[r] (on its device); the records stay on the devices, are exchanged device to device (RCCL) or — repeated devices —
// fetched block by block, and out (host, n*n_windows*per records, rank order) may be NULL when only the devices' copies
// are wanted (twx_multi_fetch_gathered).
int twx_multi_process_windows_dev(twx_multi* m, const void* const* iq_dev, int64_t n_windows, int32_t n_channels, int32_t channel,
                                  const twx_band* band, const double* df, twx_result* out) {
    if (!m) return TWX_E_ARG;
    if (!iq_dev || n_windows < 1 || n_channels < 1 || channel < TWX_ALL_CHANNELS || channel >= n_channels || (!band && !df)) return m->fail(TWX_E_ARG, "bad argument");
    return multi_guard(m, [&]() -> int {
        const size_t per = channel < 0 ? n_channels : 1, block = (size_t)n_windows * per;
        if (int rc = m->ensure_gather(block)) return rc;
        int rc = m->run_all([&](int r) -> int {
            const int e = twx_process_windows_dev(m->ctx[r], iq_dev[r], n_windows, n_channels, channel, band, df, static_cast<twx_result*>(m->send_dev[r]));
            return e ? e : twx_synchronize(m->ctx[r]);                 // records complete before the collective reads them
        });
        if (rc) return rc;
        return m->exchange_dev(block, nullptr, out);
    });
}

// BASELINE.json configs[3] as written (godual_ranging.m:75-102: one recording, consecutive windows): n_windows_total windows of ONE
// recording, context r owning the contiguous block shard(total, r, n) whose first window sits at iq_block_dev[r] on its device.
// Blocks differ by at most one window: the gather runs on blocks padded to the longest, `out` is compact, in window order.
int twx_multi_block(const twx_multi* m, int64_t n_windows_total, int32_t i, int64_t* start, int64_t* count) {
    if (!m || i < 0 || i >= m->n || n_windows_total < 0) return TWX_E_ARG;
    long long s0 = 0, c0 = 0;
    twx_multi::shard(n_windows_total, i, m->n, &s0, &c0);
    if (start) *start = s0;
    if (count) *count = c0;
    return TWX_OK;
}
int twx_multi_process_recording_dev(twx_multi* m, const void* const* iq_block_dev, int64_t n_windows_total, int32_t n_channels, int32_t channel,
                                    const twx_band* band, const double* df, twx_result* out) {
    if (!m) return TWX_E_ARG;
    if (!iq_block_dev || n_windows_total < 1 || n_channels < 1 || channel < TWX_ALL_CHANNELS || channel >= n_channels || (!band && !df)) return m->fail(TWX_E_ARG, "bad argument");
    return multi_guard(m, [&]() -> int {
        const size_t per = channel < 0 ? n_channels : 1;
        std::vector<long long> start(m->n), count(m->n);
        long long mx = 0;
        for (int r = 0; r < m->n; ++r) { twx_multi::shard(n_windows_total, r, m->n, &start[r], &count[r]); mx = std::max(mx, count[r]); }
        const size_t block = (size_t)mx * per;
        if (int rc = m->ensure_gather(block)) return rc;
        int rc = m->run_all([&](int r) -> int {
            if (count[r] == 0) return TWX_OK;
            if (!iq_block_dev[r]) return TWX_E_ARG;
            const int e = twx_process_windows_dev(m->ctx[r], iq_block_dev[r], count[r], n_channels, channel, band, df ? df + start[r] * (long long)per : nullptr,
                                                  static_cast<twx_result*>(m->send_dev[r]));
            return e ? e : twx_synchronize(m->ctx[r]);                 // records complete before the collective reads them
        });
        if (rc) return rc;
        return m->exchange_dev(block, &count, out, (int)per);
    });
}
// The exchange alone, on whatever the last device-resident call left in the contexts' send buffers (compute removed): what one
// step of a sharded recording pays for its gather.  gather_ms of twx_multi_get_info holds its wall time.
int twx_multi_exchange_only(twx_multi* m, int64_t records_per_context) {
    if (!m) return TWX_E_ARG;
    if (records_per_context < 1 || (size_t)records_per_context > m->cap_records) return m->fail(TWX_E_ARG, "no device-resident call has left that many records in the send buffers");
    return multi_guard(m, [&]() -> int { return m->exchange_dev((size_t)records_per_context, nullptr, nullptr); });
}

// NUMA placement of a device, and binding the calling thread next to it (what the twx_multi workers do for themselves; the
// one-process-per-GPU ranks of dist.py / bench.py call it once after choosing their device)
int twx_device_affinity(int32_t device, int32_t* numa_node, char* cpulist, size_t cap) {
    (void)hipGetLastError();
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) != hipSuccess) { (void)hipGetLastError(); return TWX_E_ARG; }
    const twx::DeviceAffinity a = twx::affinity_of_pci(bus);
    if (numa_node) *numa_node = a.numa_node;
    if (cpulist && cap) snprintf(cpulist, cap, "%s", a.cpulist.c_str());
    return TWX_OK;
}
int twx_pin_thread_to_device(int32_t device, int32_t* numa_node, int32_t* n_cpus) {
    (void)hipGetLastError();
    if (numa_node) *numa_node = -1;
    if (n_cpus) *n_cpus = 0;
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) != hipSuccess) { (void)hipGetLastError(); return TWX_E_ARG; }
    const twx::DeviceAffinity a = twx::affinity_of_pci(bus);
    if (numa_node) *numa_node = a.numa_node;
    const char* np_env = getenv("TWX_NO_PIN");
    if ((np_env && atoi(np_env) != 0) || a.numa_node < 0) return TWX_OK;          // nothing to do is not an error
    const int k = twx::pin_current_thread(a.cpus);
    if (k < 0) return TWX_E_STATE;
    if (n_cpus) *n_cpus = k;
    return TWX_OK;
}

int twx_multi_fetch_gathered(twx_multi* m, int32_t i, twx_result* out, int64_t n_records) {
    if (!m) return TWX_E_ARG;
    if (!out || i < 0 || i >= m->n || n_records < 0 || (size_t)n_records > m->cap_records * (size_t)m->n) return m->fail(TWX_E_ARG, "bad argument");
    (void)hipSetDevice(m->devices[i]);
    return hipMemcpy(out, m->recv_dev[i], (size_t)n_records * sizeof(twx_result), hipMemcpyDeviceToHost) == hipSuccess ? TWX_OK : m->fail(TWX_E_HIP, "D2H failed");
}

}  // extern "C"
