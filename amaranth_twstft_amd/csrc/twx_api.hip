// twx_api.hip — C ABI (include/twstft_hip.h) of the TWSTFT correlator: context, plan choice,
// twiddle tables, the per-batch kernel sequence and the inspection entry points.
#include <hip/hip_runtime.h>
#include <shared_mutex>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <math.h>
#include <stdio.h>
#include <sys/types.h>
#include <string.h>
#include <stdlib.h>
#include <algorithm>
#include <atomic>
#include <deque>
#include <future>
#include <unistd.h>
#include <dlfcn.h>
#include <dirent.h>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "twx_kernels.h"
#include "twx_plans.h"
#include "twx_internal.h"
#include "twx_workers.h"

#ifndef TWX_PIN_MODE_DEFAULT
#define TWX_PIN_MODE_DEFAULT 0
#endif
namespace twx {

// ------------------------------------------------------------------------------------------
// registry
// ------------------------------------------------------------------------------------------
#define TWX_MAX_CHANNELS 4
static int wshift_of(int w) { int sft = 0; while ((1 << sft) < w) ++sft; return sft; }
LaunchEvents& launch_events() { static thread_local LaunchEvents le; return le; }
// Live contexts keep pointers to their plans while plug-ins loaded later register more: a deque never moves its
// elements on push_back (a vector would leave those pointers dangling), and the mutex orders registration (dlopen's
// static initialisers, possibly from another thread's twx_create) against lookups.
static thread_local std::string g_create_err;
struct ColEntry { ColOps o; bool plugin; };
struct RowEntry { RowOps o; bool plugin; };
static std::deque<ColEntry>& col_reg() { static std::deque<ColEntry> v; return v; }
static std::deque<RowEntry>& row_reg() { static std::deque<RowEntry> v; return v; }
static std::recursive_mutex& reg_mu() { static std::recursive_mutex m; return m; }
static bool& registering_plugin() { static bool b = false; return b; }      // true while load_plan_file() runs a plug-in's static initialisers (under reg_mu)
void register_col(const ColOps& o) { std::lock_guard<std::recursive_mutex> g(reg_mu()); col_reg().push_back(ColEntry{o, registering_plugin()}); }
void register_row(const RowOps& o) { std::lock_guard<std::recursive_mutex> g(reg_mu()); row_reg().push_back(RowEntry{o, registering_plugin()}); }
const ColOps* find_col(int L, int f64, int W) {            // W = 0: any tile width
    std::lock_guard<std::recursive_mutex> g(reg_mu());
    for (auto& e : col_reg()) if (e.o.L == L && e.o.f64 == f64 && (W == 0 || e.o.W == W)) return &e.o;
    return nullptr;
}
const RowOps* find_row(int L, int f64) {
    std::lock_guard<std::recursive_mutex> g(reg_mu());
    for (auto& e : row_reg()) if (e.o.L == L && e.o.f64 == f64) return &e.o;
    return nullptr;
}
// The pair (N1, N2, W) for a window of n samples is a function of n ALONE: pairs made of built-in plans win over anything a
// plug-in brings (so the fp32 rounding of a length the library was built for does not depend on which other lengths the
// process happened to load), and among plug-ins the ranking below does not depend on the load order.
static bool choose_split_from(long long n, int f64, bool builtin_only, const ColOps** col, const RowOps** row) {
    // Preference: N2 = 4000 where it divides n (its rows leave room for a third resident workgroup and the column
    // length N1 = n/4000 keeps the column workgroups full: measured 26.7 vs 22.8 Gsample/s at n = 2e5, 32.3 vs 29.2
    // at n = 1e6 against N2 = 8000, tools/prof_n.py), otherwise the longest row plan.
    const ColOps* bc = nullptr; const RowOps* br = nullptr;
    const char* force = getenv("TWX_N2");                 // experiments: force the row length
    const int forced = force ? atoi(force) : 0;
    for (auto& re : row_reg()) {
        const RowOps& r = re.o;
        if (builtin_only && re.plugin) continue;
        if (r.f64 != f64 || n % r.L || (r.L & 1)) continue;
        if (forced && r.L != forced) continue;
        const long long n1 = n / r.L;
        if (n1 > 100000) continue;
        const ColOps* c = nullptr;                        // widest column tile of that length whose width divides the row
        static const int force_w = [] { const char* e = getenv("TWX_COL_W"); return e ? atoi(e) : 0; }();      // experiments: force the tile width
        for (auto& ce : col_reg()) {
            const ColOps& o = ce.o;
            if (builtin_only && ce.plugin) continue;
            if (force_w && o.W != force_w) continue;
            if (o.L != (int)n1 || o.f64 != f64 || r.L % o.W != 0) continue;
            if (!c || o.W > c->W) c = &o;
        }
        if (!c) continue;
        // widest column tile first (HBM piece size), then the row: 4000, then the longest row that still leaves room for
        // two workgroups per CU (<= 8192), long rows last
        auto rank = [](int L) { return L == 4000 ? (1 << 30) : (L <= 8192 ? L : L - (1 << 20)); };
        if (!br || c->W > bc->W || (c->W == bc->W && rank(r.L) > rank(br->L))) { br = &r; bc = c; }
    }
    if (!br) return false;
    *col = bc; *row = br;
    return true;
}
// The LAST pass may use another tile width than the forward passes: Bz is not tile-blocked, so nothing but this kernel sees its W.
// Complex double: a tile of L x W x 16 bytes; two workgroups per CU (<= 80 KB each) beat one with a wider tile, and W = 8 still moves whole
// 128-byte pieces (k_col_inv 0.322 -> 0.233 ms at L = 625, profiles/r05_f64_colw.txt) — while the forward passes keep W = 16, because the
// row passes gather A in pieces of W elements and lose more with 128-byte pieces than the forward column passes gain.  fp32: the same plan.
static const ColOps* choose_col_inv(const ColOps* col, int n2, int f64) {
    if (!f64 || getenv("TWX_COL_W") || (long long)col->L * col->W * 16 <= 80 * 1024) return col;
    std::lock_guard<std::recursive_mutex> g(reg_mu());
    const ColOps* best = col;
    for (auto& ce : col_reg()) {
        const ColOps& o = ce.o;
        if (o.L != col->L || o.f64 != f64 || n2 % o.W != 0 || (long long)o.L * o.W * 16 > 80 * 1024) continue;
        if (best == col || o.W > best->W) best = &o;
    }
    return best;
}
bool choose_split(long long n, int f64, const ColOps** col, const RowOps** row) {
    std::lock_guard<std::recursive_mutex> g(reg_mu());
    return choose_split_from(n, f64, true, col, row) || choose_split_from(n, f64, false, col, row);
}


// ------------------------------------------------------------------------------------------
// plan plug-ins: shared objects built from twx_inst_col.hip / twx_inst_row.hip for one more transform length
// (amaranth_twstft_amd/plans.py); loading one runs its static registration (register_col / register_row above)
// ------------------------------------------------------------------------------------------
#ifndef TWX_SRC_HASH
#define TWX_SRC_HASH "dev"
#endif
static std::vector<std::string>& loaded_plugins() { static std::vector<std::string> v; return v; }
// A plug-in registers kernels whose argument structs come from the kernel sources it was compiled with: only files named
// *_<hash of THIS library's kernel sources>.so are accepted (amaranth_twstft_amd/plans.py names them so).  The whole of it
// runs under the registry mutex: two threads creating contexts for unknown lengths do not dlopen the same file twice.
static int load_plan_file(const std::string& path) {
    std::lock_guard<std::recursive_mutex> g(reg_mu());
    for (auto& p : loaded_plugins()) if (p == path) return 0;
    const std::string tail = std::string("_") + TWX_SRC_HASH + ".so";
    if (path.size() <= tail.size() || path.compare(path.size() - tail.size(), tail.size(), tail) != 0) {
        g_create_err = std::string("plan ") + path + " was not built from this library's kernel sources (expected *" + tail + ")";
        return -1;
    }
    registering_plugin() = true;
    void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_GLOBAL);
    registering_plugin() = false;
    if (!h) { g_create_err = std::string("cannot load plan ") + path + ": " + dlerror(); return -1; }
    loaded_plugins().push_back(path);
    return 0;
}
static std::string default_plan_dir() {
    if (const char* e = getenv("TWX_PLAN_DIR")) return e;
    Dl_info info;
    if (dladdr((const void*)&load_plan_file, &info) && info.dli_fname) {
        std::string p = info.dli_fname;
        const size_t k = p.rfind('/');
        return (k == std::string::npos ? std::string(".") : p.substr(0, k)) + "/plans";
    }
    return "plans";
}
// loads every plug-in of the plan directory built from THIS library's kernel sources (file name *_<hash>.so) that is
// not loaded yet; returns how many
static int scan_plan_dir() {
    std::lock_guard<std::recursive_mutex> g(reg_mu());
    const std::string dir = default_plan_dir();
    DIR* d = opendir(dir.c_str());
    if (!d) return 0;
    int n = 0;
    const std::string tail = std::string("_") + TWX_SRC_HASH + ".so";
    while (dirent* e = readdir(d)) {
        const std::string name = e->d_name;
        if (name.size() > tail.size() && name.compare(name.size() - tail.size(), tail.size(), tail) == 0) {
            const size_t before = loaded_plugins().size();
            if (load_plan_file(dir + "/" + name) == 0 && loaded_plugins().size() > before) ++n;
        }
    }
    closedir(d);
    return n;
}

#define HIPCHK(call)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) {                                                             \
            char buf_[256];                                                                 \
            snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return fail(TWX_E_HIP, buf_);                                                   \
        }                                                                                   \
    } while (0)

// ------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------
template <typename S, typename D>
__global__ void k_convert(const cpx<S>* __restrict__ in, cpx<D>* __restrict__ out, long long n, double scale) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = mk<D>((D)(in[i].x * scale), (D)(in[i].y * scale));
}

// spectrum layouts: the row pass leaves bin k = k1 + N1*k2 at [k1][k2]; hosts and the replica set-up of the DLL/PLL receiver
// (twx_rx.hip) speak natural order.  k_spec_to_natural: [k1][k2] (context precision) -> natural complex double;
// k_spec_from_natural: natural complex double x scale -> [k1][k2] (context precision).
template <typename T>
__global__ void k_spec_to_natural(const cpx<T>* __restrict__ in, cpx<double>* __restrict__ out, int n1, int n2) {
    const long long n = (long long)n1 * n2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long k1 = i / n2, k2 = i % n2;                    // coalesced reads, strided writes (set-up path)
        out[k1 + (long long)n1 * k2] = mk<double>((double)in[i].x, (double)in[i].y);
    }
}
template <typename T>
__global__ void k_spec_from_natural(const cpx<double>* __restrict__ in, cpx<T>* __restrict__ out, int n1, int n2, double scale) {
    const long long n = (long long)n1 * n2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long k1 = i / n2, k2 = i % n2;
        const cpx<double> v = in[k1 + (long long)n1 * k2];
        out[i] = mk<T>((T)(v.x * scale), (T)(v.y * scale));
    }
}

// every dec-th sample of a complex-float stream -> contiguous (downconv_acq reads smp[i*dec], rxcomplex.cpp:1039-1049)
__global__ void k_stride_copy(const cpx<float>* __restrict__ in, cpx<float>* __restrict__ out, long long n, int dec) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = in[i * dec];
}

// Sweep bookkeeping of the acquisition loop (rxcomplex.cpp:534-567) kept ON THE DEVICE, so the rounds of the step halving
// are enqueued back to back without a host round trip: scan the records of the round that just finished in trial order
// (`if (pk > ci[i].pk)`, :556-562: strictly larger wins, pk = |z| of the izamax element), then halve the step (:565-567)
// and write the next round's trial carriers fc-step, fc, fc+step... (`for (fcc = flow; fcc <= fhigh; fcc += fstep)`,
// :538: repeated addition, at most `cap` of them) into the batch's df vector.  One workgroup.
struct AcqState { double fc, pk, step; long long pt, n_trials; int cnt, pad; };
__global__ __launch_bounds__(256) void k_acq_update(const twx_result* __restrict__ rec, int n_host, AcqState* st, double* __restrict__ dfv, int cap, long long ptmod) {
    __shared__ double spk[256];
    __shared__ int sidx[256];
    const int t = threadIdx.x;
    const int n = n_host >= 0 ? n_host : st->cnt;
    // the sequential rule `pk > best` keeps the FIRST occurrence of the largest peak: a first-index arg-max, done in parallel
    double bp = -1.0; int bi = 0x7fffffff;
    for (int i = t; i < n; i += 256) {
        const double pk = sqrt(rec[i].xval[0] * rec[i].xval[0] + rec[i].xval[1] * rec[i].xval[1]);
        if (pk > bp) { bp = pk; bi = i; }
    }
    spk[t] = bp; sidx[t] = bi;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if (t < h && (spk[t + h] > spk[t] || (spk[t + h] == spk[t] && sidx[t + h] < sidx[t]))) { spk[t] = spk[t + h]; sidx[t] = sidx[t + h]; }
        __syncthreads();
    }
    if (t) return;
    AcqState s = *st;
    if (n > 0 && spk[0] > s.pk) {
        const int i = sidx[0];
        s.fc = rec[i].df; s.pk = spk[0]; s.pt = ptmod > 0 ? rec[i].indice0 % ptmod : rec[i].indice0;
    }
    s.n_trials += n;
    s.step = s.step / 2.0;
    const double frange = s.step, flow = s.fc - frange, fhigh = s.fc + frange;
    int c = 0;
    for (double fcc = flow; fcc <= fhigh && c < cap; fcc += s.step) dfv[c++] = fcc;
    s.cnt = c;
    for (int i = c; i < cap; ++i) dfv[i] = s.fc;          // unused slots of the batch: a valid carrier, records ignored
    *st = s;
}

// Replica variants on the finished code spectrum S = conj(FFT(2c-1)) (layout [k1][k2], DC at index 0), all exact in
// the spectrum because they only differ by a constant in the time domain:
//   0/1 levels:  conj(FFT(c)) = S/2 (+ N/2 at DC);  zero mean: DC = 0;  complex code ci + j cq: Si - j Sq.
template <typename T>
__global__ void k_code_variant(cpx<T>* __restrict__ si, const cpx<T>* __restrict__ sq, long long n, int unipolar, int zero_mean, double dc_add) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        double x = (double)si[i].x, y = (double)si[i].y, qx = 0, qy = 0;
        if (sq) { qx = (double)sq[i].x; qy = (double)sq[i].y; }
        if (unipolar) { x *= 0.5; y *= 0.5; qx *= 0.5; qy *= 0.5; if (i == 0) { x += dc_add; if (sq) qx += dc_add; } }
        // Si - j*Sq
        double rx = x + qy, ry = y - qx;
        if (zero_mean && i == 0) { rx = 0; ry = 0; }
        si[i] = mk<T>((T)rx, (T)ry);
    }
}

// chips of LFSR(bitlen,taps), seed 1, one byte per chip (amaranth_twstft/common.py:23-30,59-73).
// Every thread jumps to its segment by applying the 2^j-step transition matrices
// (precomputed on the host), then runs the bit-serial recurrence.
__global__ void k_lfsr(int bitlen, unsigned taps, long long n, long long seg, const unsigned* __restrict__ jump /*[40][32]*/,
                       unsigned char* __restrict__ out) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long start = t * seg;
    if (start >= n) return;
    unsigned state = 1u;
    for (int j = 0; j < 40; ++j) {
        if ((start >> j) & 1) {
            const unsigned* m = jump + j * 32;       // column i = image of basis state bit i
            unsigned ns = 0;
            for (int i = 0; i < bitlen; ++i) if ((state >> i) & 1u) ns ^= m[i];
            state = ns;
        }
    }
    const long long end = min(n, start + seg);
    for (long long i = start; i < end; ++i) {
        out[i] = (unsigned char)(state & 1u);
        const unsigned bit = __popc(state & taps) & 1u;
        state = (state >> 1) | (bit << (bitlen - 1));
    }
}

// integer synthetic capture generator — must stay bit-identical to amaranth_twstft_amd/synth.py
__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}
__device__ __forceinline__ void icossin_q30(unsigned ph, long long& cs, long long& sn) {
    const long long quad = ph >> 30;
    const long long x = ph & ((1u << 30) - 1);
    const long long y = (x * 1686629713ll) >> 30;
    const long long y2 = (y * y) >> 30;
    const long long one = 1ll << 30;
    long long t = one - y2 / 110;
    t = one - (((y2 / 72) * t) >> 30);
    t = one - (((y2 / 42) * t) >> 30);
    t = one - (((y2 / 20) * t) >> 30);
    t = one - (((y2 / 6) * t) >> 30);
    const long long s0 = (y * t) >> 30;
    long long u = one - y2 / 132;
    u = one - (((y2 / 90) * u) >> 30);
    u = one - (((y2 / 56) * u) >> 30);
    u = one - (((y2 / 30) * u) >> 30);
    u = one - (((y2 / 12) * u) >> 30);
    const long long c0 = one - (((y2 / 2) * u) >> 30);
    cs = quad == 0 ? c0 : quad == 1 ? -s0 : quad == 2 ? -c0 : s0;
    sn = quad == 0 ? s0 : quad == 1 ? c0 : quad == 2 ? -s0 : -c0;
}
struct SynthChan { long long delay_q8, fstep, phi0, amp, noise_gain, seed, stream, pad; };
struct SynthArgs { SynthChan ch[4]; };
__global__ void k_synth(short2* __restrict__ out, long long n, long long n0, const unsigned char* __restrict__ chips,
                        long long n_chips, int sps, int nch, SynthArgs a) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n * nch; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % nch);
        const long long idx = n0 + i / nch;
        const SynthChan p = a.ch[c];
        const long long period = n_chips * sps * 256;
        long long q = (idx * 256 - p.delay_q8) % period; if (q < 0) q += period;
        const long long code = 2 * (long long)chips[q / (sps * 256)] - 1;
        const unsigned ph = (unsigned)(((unsigned long long)idx * (unsigned long long)(p.fstep & 0xffffffffll) +
                                        (unsigned long long)(p.phi0 & 0xffffffffll)) & 0xffffffffull);
        long long cs, sn;
        icossin_q30(ph, cs, sn);
        long long si = (p.amp * code * cs) >> 30;
        long long sq = (p.amp * code * sn) >> 30;
        if (p.noise_gain) {
            const unsigned long long key = (unsigned long long)p.seed * 0xD6E8FEB86659FD93ull + (unsigned long long)p.stream * 0xA0761D6478BD642Full;
            const unsigned long long ctr = (unsigned long long)idx * 0x9E3779B97F4A7C15ull + key;
            const unsigned long long h[4] = {mix64(ctr), mix64(ctr ^ 0x5851F42D4C957F2Dull), mix64(ctr ^ 0x2545F4914F6CDD1Dull),
                                             mix64(ctr ^ 0x9FB21C651E98DF25ull)};
            long long ni = -4 * 65535, nq = -4 * 65535;
            for (int k = 0; k < 4; ++k) {
                ni += (long long)(h[k] & 0xffff) + (long long)((h[k] >> 16) & 0xffff);
                nq += (long long)((h[k] >> 32) & 0xffff) + (long long)(h[k] >> 48);
            }
            si += (ni * p.noise_gain) >> 20;
            sq += (nq * p.noise_gain) >> 20;
        }
        si = si < -32768 ? -32768 : si > 32767 ? 32767 : si;
        sq = sq < -32768 ? -32768 : sq > 32767 ? 32767 : sq;
        out[i] = make_short2((short)si, (short)sq);
    }
}

// ------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------
struct ProfRec { hipEvent_t a, b; int cls; long long units; };
static const char* kProfNames[] = {"k_sums", "k_col_fwd_square", "k_row_band", "k_df_tables", "k_col_fwd_mix",
                                   "k_row_mid", "k_col_inv", "k_peak", "k_row_caf", "k_col_inv_caf", "k_caf_reduce", "caf_forward"};
enum { PC_SUMS = 0, PC_COL_SQ, PC_ROW_BAND, PC_DFT, PC_COL_MIX, PC_ROW_MID, PC_COL_INV, PC_PEAK, PC_ROW_CAF, PC_COL_INV_CAF, PC_CAF_REDUCE,
       PC_CAF_FWD, PC_COUNT };

struct CtxBase {
    twx_config cfg{};
    std::string err;
    int dev = 0;
    hipStream_t stream = nullptr;
    int fir_mfma = -1;            // TWX_OPT_FIR_MFMA: -1 = follow TWX_FIR_MFMA of the environment, 0 = never, 1 = the matrix-core front end
    long long N = 0; int N1 = 0, N2 = 0, R = 1, B = 1;
    const ColOps* col = nullptr; const RowOps* row = nullptr;
    const ColOps* colinv = nullptr;          // the last pass's plan (choose_col_inv): col unless complex double needs a narrower tile
    std::vector<std::pair<void*, size_t>> allocs;
    long long dev_bytes = 0;
    // profiling
    bool profile = false;
    std::vector<ProfRec> prof_pending;
    std::vector<hipEvent_t> ev_pool;
    double prof_ms[PC_COUNT] = {0}; long long prof_n[PC_COUNT] = {0}; long long prof_units[PC_COUNT] = {0};

    virtual ~CtxBase() {
        for (auto& p : allocs) (void)hipFree(p.first);
        for (auto& r : prof_pending) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
        for (auto e : ev_pool) (void)hipEventDestroy(e);
        if (stream) { twx::fence_unregister(dev, stream); (void)hipStreamDestroy(stream); }
    }
    int fail(int code, const std::string& msg) { err = msg; return code; }
    // The pinned staging buffers of the ingest.  TWX_PIN_MODE: 0 hipHostMalloc default (coherent, fine-grained), 1 hipHostMallocNonCoherent
    // (host-cacheable), 2 ordinary pages (2-MB aligned, MADV_HUGEPAGE) pinned with hipHostRegister.  The kernel's copy_to_user of a pread
    // into the default kind runs at 1 - 2 GB/s per thread against 10 into ordinary memory (profiles/r06_io_rate.txt).
    static int pin_mode() { static const int m = [] { const char* e = getenv("TWX_PIN_MODE"); return e ? atoi(e) : TWX_PIN_MODE_DEFAULT; }(); return m; }
    std::vector<std::pair<void*, int>> pin_kinds;
    void* pin_alloc(size_t bytes) {
        void* p = nullptr;
        const int mode = pin_mode();
        if (mode == 2) {
            const size_t len = (bytes + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
            if (posix_memalign(&p, 2u << 20, len) != 0) return nullptr;
            (void)madvise(p, len, MADV_HUGEPAGE);
            memset(p, 0, len);                                   // touch: the pages exist before they are pinned
            if (hipHostRegister(p, len, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); free(p); return nullptr; }
        } else if (hipHostMalloc(&p, bytes, mode == 1 ? hipHostMallocNonCoherent : hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        pin_kinds.push_back({p, mode});
        return p;
    }
    void pin_free(void* p) {
        int mode = 0;
        for (auto it = pin_kinds.begin(); it != pin_kinds.end(); ++it) if (it->first == p) { mode = it->second; pin_kinds.erase(it); break; }
        if (mode == 2) { (void)hipHostUnregister(p); free(p); }
        else (void)hipHostFree(p);
    }
    template <typename U> int dalloc(U** p, size_t count) {
        void* q = nullptr;
        size_t bytes = std::max<size_t>(count * sizeof(U), 16);
        hipError_t e = hipMalloc(&q, bytes);
        if (e != hipSuccess) { char b[160]; snprintf(b, sizeof b, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e)); return fail(TWX_E_NOMEM, b); }
        allocs.push_back({q, bytes}); dev_bytes += (long long)bytes;
        *p = reinterpret_cast<U*>(q);
        return TWX_OK;
    }
    void dfree(void* p) {
        for (auto it = allocs.begin(); it != allocs.end(); ++it)
            if (it->first == p) { dev_bytes -= (long long)it->second; allocs.erase(it); break; }
        (void)hipFree(p);
    }
    // device temporaries of one call: returned to the allocator on every exit path
    struct Scratch {
        CtxBase* c; std::vector<void*> ptrs;
        explicit Scratch(CtxBase* c_) : c(c_) {}
        Scratch(const Scratch&) = delete;
        ~Scratch() { for (void* p : ptrs) c->dfree(p); }
        template <typename U> int get(U** p, size_t count) { int rc = c->dalloc(p, count); if (!rc) ptrs.push_back(*p); return rc; }
    };
    hipEvent_t get_event() {
        if (!ev_pool.empty()) { hipEvent_t e = ev_pool.back(); ev_pool.pop_back(); return e; }
        hipEvent_t e; (void)hipEventCreate(&e); return e;
    }
    struct ProfScope {
        CtxBase* c; ProfRec r; bool on;
        ProfScope(CtxBase* c_, int cls, long long units) : c(c_), on(c_->profile) {
            if (on) { r.cls = cls; r.units = units; r.a = c->get_event(); r.b = c->get_event(); launch_events() = LaunchEvents{r.a, r.b}; }
        }
        ~ProfScope() {
            if (!on) return;
            LaunchEvents& le = launch_events();
            if (le.start) { le = LaunchEvents{}; c->ev_pool.push_back(r.a); c->ev_pool.push_back(r.b); }   // nothing was launched
            else c->prof_pending.push_back(r);
        }
    };
    void prof_collect() {
        for (auto& r : prof_pending) {
            float ms = 0;
            (void)hipEventSynchronize(r.b);
            (void)hipEventElapsedTime(&ms, r.a, r.b);
            prof_ms[r.cls] += ms; prof_n[r.cls] += 1; prof_units[r.cls] += r.units;
            ev_pool.push_back(r.a); ev_pool.push_back(r.b);
        }
        prof_pending.clear();
    }
    void* aux_buf[AUX_SCRATCH_SLOTS] = {}; size_t aux_cap[AUX_SCRATCH_SLOTS] = {};   // ctx_scratch (twx_internal.h)
    std::vector<unsigned char> aux_shadow[AUX_SCRATCH_SLOTS];
    // Context-owned device buffer number `slot`, at least `bytes` long, kept across calls (slots 0-1: twx_aux.hip, 2-5: the CAF
    // surface): re-allocated only when it has to grow, after synchronising the context.  nullptr on failure (error text set).
    void* scratch_slot(int slot, size_t bytes) {
        if (slot < 0 || slot >= AUX_SCRATCH_SLOTS) { fail(TWX_E_ARG, "bad scratch slot"); return nullptr; }
        if (aux_cap[slot] >= bytes && aux_buf[slot]) return aux_buf[slot];
        aux_shadow[slot].clear();
        if (aux_buf[slot]) { (void)sync_all(); dfree(aux_buf[slot]); aux_buf[slot] = nullptr; aux_cap[slot] = 0; }
        char* p = nullptr;
        const size_t want = bytes + bytes / 8 + 256;          // a little head-room: sizes that creep up do not re-allocate every call
        if (dalloc(&p, want)) return nullptr;
        aux_buf[slot] = p; aux_cap[slot] = want;
        return p;
    }
    int remove_mean = 1;          // TWX_OPT_REMOVE_MEAN
    int dbg_fault = 0;            // TWX_OPT_DEBUG_FAULT
    int dbg_only = -1, dbg_repeat = 1;   // TWX_OPT_DEBUG_ONLY / _REPEAT: launch one kernel class of the chain, n times (power / clock probes)
    int reps(int cls) const { return dbg_only < 0 ? 1 : (dbg_only == cls ? dbg_repeat : 0); }
    int snr_valid = 1;            // 0 for replicas that are not a +-1 code
    // twx_set_resample: the velocity-compensated window (experiments/220706_TWSTFT/godual_ranging_OP_vitesse.m:4,40-43,68-71)
    double rs_v = 0.0, rs_t0 = 0.0; long long rs_dt = 0;       // vitesse (0 = off), and the carried t0 / dt BEFORE the next window
    virtual int set_resample(double vitesse, double t0, long long dt) = 0;
    // TWX_OPT_BRUIT_LEN / TWX_OPT_NOISE_SQUARE_LEN: the off-peak and squared-spectrum SNR estimators of process_OP.m (k_offpeak, k_sq_noise)
    int bruit_len = 0, sq_len = 0;
    twx_extra* extra_dev = nullptr; size_t extra_cap = 0; long long extra_n = 0;   // the estimators of the LAST call, record for record
    long long ex_off = -1;                   // record index of the batch being enqueued inside extra_dev (-1: this call has none)
    virtual int set_extra(int which, long long value) = 0;
    // TWX_OPT_SELFCHECK: Parseval's identity per row of the middle pass (k_rowd<MID, CHK> + k_chk_verdict)
    float selfcheck_tol = 0.f;    // 0 = off
    unsigned* chk_stat = nullptr; // [0] largest relative deviation seen (float bits), [1] rows flagged — since the last reset
    virtual int set_selfcheck(long long value) = 0;
    virtual int init() = 0;
    virtual int sync_all() = 0;
    virtual int pipeline_depth() const = 0;
    virtual int process(const void* iq_dev, long long nwin, int nch, int ch, const twx_band* band, const double* df,
                        twx_result* out_dev) = 0;
    virtual int fft_forward(const double* in, double* out) = 0;
    virtual int code_spectrum(double* out) = 0;
    virtual int xcorr_map(const int16_t* iq, int nch, int ch, double df, double* out) = 0;
    virtual int caf_bins(const int16_t* iq, int nch, int ch, long long k_lo, long long k_hi, double* pk, long long* lag) = 0;
    virtual int caf_bins_dev(const void* iq_dev, int nch, int ch, long long k_lo, long long k_hi, double* pk, long long* lag) = 0;
    virtual int caf_freqs(const int16_t* iq, int nch, int ch, const double* freqs, long long nf, twx_result* out) = 0;
    virtual int sqspec_bins(const void* iq_dev, long long L, int nch, int ch, const long long* bins, int nb, double* out) = 0;
    virtual int sqspec_band(const void* iq_dev, long long L, int nch, int ch, long long k_lo, long long nk, double* out) = 0;
    virtual int process_file(const char* path, int nch, int ch, long long skip, const twx_band* band, double df_const,
                             twx_result* out, long long max_windows, long long* n_done) = 0;
    virtual int process_host(const int16_t* iq, long long nwin, int nch, int ch, const twx_band* band, const double* df,
                             twx_result* out) = 0;
    virtual int process_complex(const double* re, const double* im, long long stride, long long nwin, const twx_band* band,
                                const double* df, twx_result* out) = 0;
    virtual int set_code_spectrum(const double* spec) = 0;
    virtual int set_code_spectrum_dev(const void* spec_dev) = 0;
    virtual int fft_forward_dev(const void* in_dev, void* out_dev) = 0;
    virtual int xcorr_map_dev(const void* iq_dev, int nch, int ch, double df, void* out_dev) = 0;
    virtual int caf_freqs_cdev(const void* d_dev, const double* freqs, long long nf, int flags, twx_result* out) = 0;
    virtual int acquire_cdev(const void* d_dev, double fc0, double frange, double fstep, long long ptmod, int flags, twx_acq_result* out) = 0;
};

template <typename T> static void host_twiddle(std::vector<cpx<T>>& v, long long count, long long num_mul, long long den, int sign) {
    v.resize(count);
    for (long long m = 0; m < count; ++m) {
        // angle = sign * 2*pi * (m*num_mul mod den)/den, evaluated in long double
        const long long r = (long long)(((__int128)m * num_mul) % den);
        const long double a = 2.0L * 3.14159265358979323846264338327950288L * (long double)r / (long double)den;
        v[m] = mk<T>((T)cosl(a), (T)(sign * sinl(a)));
    }
}


// StageTabs<P> entries for radices rad[0..S) (see twx_fft.h): tab[s][d][r][x] = exp(-2 pi i x r/den(s,d))
template <typename T> static void host_stage_tabs(std::vector<cpx<T>>& v, int S, const int* rad) {
    v.clear();
    const long double tp = 2.0L * 3.14159265358979323846264338327950288L;
    for (int s = 1; s < S; ++s)
        for (int d = 0; d < s; ++d) {
            long long den = rad[s];
            for (int i = d; i < s; ++i) den *= rad[i];
            for (int r = 0; r < rad[s]; ++r)
                for (int x = 0; x < rad[d]; ++x) {
                    const long double a = tp * (long double)(((long long)x * r) % den) / (long double)den;
                    v.push_back(mk<T>((T)cosl(a), (T)(-sinl(a))));
                }
        }
}

template <typename T> struct Ctx : CtxBase {
    using C = cpx<T>;
    int tshift = 11;
    double scale_pow2 = 1.0;
    // tables
    C *tw1 = nullptr, *tcw = nullptr, *stab_f = nullptr, *stab_i = nullptr, *ea = nullptr, *eb = nullptr, *ta = nullptr, *tb = nullptr, *ramp1 = nullptr;
    cpx<double>* tw1d = nullptr;
    C* cspec = nullptr;
    C *cspec_perm = nullptr, *dtabs = nullptr, *ea_d = nullptr, *eb_d = nullptr;   // DIF/DIT row pass (k_rowd)
    C *cspec_plain = nullptr, *cspec_perm_plain = nullptr;   // Hamming-window contexts: the UNWINDOWED code spectrum, for the wipe-off statistics
    C* wr_d = nullptr;            // exp(-2 pi i j/R), j < R: pruned last stage of k_rowd<BAND>
    C* wm_d = nullptr;            // exp(-2 pi i j/(R R)): k_rowd_bandsum (the pruned row pass without the row in LDS)
    C* vw_d = nullptr;            // [k1][2][R] exp(+2 pi i k1 a/N), exp(+2 pi i k1 R b/N): k_rowd<MID>'s folded output twiddle
    C* vc_d = nullptr;            // [k1][c] exp(+2 pi i k1 c M/N): stage C's per-row output twiddle of k_rowd<MID> (scalar loads)
    int use_rowd = 0;
    unsigned char* chips_dev = nullptr;
    // batch buffers
    WinSums* sums = nullptr; double* dfv = nullptr; long long* dfidx = nullptr;
    SumPart* sum_parts = nullptr;
    C *e1 = nullptr, *e2 = nullptr, *A = nullptr, *Bz = nullptr, *dc = nullptr;
    ArgPart<T>*part_band = nullptr, *part_peak = nullptr;
    twx_result* res_dev = nullptr;
    // Two pipeline slots (own stream + own batch buffers): consecutive batches alternate slots so that the
    // compute-heavy middle pass of one batch can share CUs with the memory-heavy column passes of the other.
    struct Slot {
        hipStream_t stream; WinSums* sums; double* dfv; long long* dfidx; C *e1, *e2, *A, *Bz, *dc;
        ArgPart<T>*part_band, *part_peak; twx_result* res_dev; double* fine_u; double* csum_part;
        short2* planar; WinSums* sums2;      // two-channel captures in all-channel mode: planar copies + both channels' sums (lazy)
        SumPart* sum_parts;                  // k_sums: [2][B][TWX_SUMS_MAXCHUNKS] per-workgroup partials
        double tables_df; int tables_nb;     // the NCO tables e1/e2 of windows [0, tables_nb) hold this carrier (tables_nb = 0: unknown)
        double* ex_part_b; double* ex_part_s; double* ex_sqmax;   // the SNR estimators' partial sums (allocated when an option is first set)
        float* chk_rows; int* chk_flag;      // TWX_OPT_SELFCHECK (allocated when the option is first set): [B][N1][TWX_CHK_SLOTS] sums, [B] status words
    };
    // twx_set_resample: per window of the CURRENT call the script's t0, edge word and dt (resample_begin), on the host and — uploaded
    // before the first batch of the call is enqueued — on the device; rs_off = first window of the batch being enqueued (-1: none)
    std::vector<double> rs_t0s, rs_t0_after; std::vector<int> rs_edges, rs_dts; std::vector<long long> rs_dt_after;
    double* rs_dev_t0 = nullptr; int* rs_dev_edge = nullptr; int* rs_dev_dt = nullptr; size_t rs_dev_cap = 0;
    long long rs_off = -1;
    Slot slots[4] = {}; int nslots = 1;
    bool sums_ready = false;                 // run_batch_in: `sums` already holds this batch's statistics (k_sums_deint2)
    hipEvent_t ev_fork = nullptr, ev_join[4] = {};   // ordering of slots 1.. against slot 0 = twx_stream() (process())
    struct Stage { void* host; short2* dev; size_t bytes; long long w0; int nb; };
    Stage stage[4] = {};     // pinned host + device staging of twx_process_file, kept across calls
    void use_slot(int k) {
        const Slot& q = slots[k];
        stream = q.stream; sums = q.sums; dfv = q.dfv; dfidx = q.dfidx; e1 = q.e1; e2 = q.e2; A = q.A; Bz = q.Bz; dc = q.dc;
        part_band = q.part_band; part_peak = q.part_peak; res_dev = q.res_dev; fine_u = q.fine_u; csum_part = q.csum_part;
        sum_parts = q.sum_parts;
    }
    // k_df_tables: slices per window, so that a launch of few windows still spreads its fp64 sincospi over ~64 workgroups
    int df_slices(int nb) const { return (int)std::max<long long>(1, std::min<long long>(std::min<long long>(16, (N1 + N2 + 511) / 512), (64 + nb - 1) / nb)); }
    // k_sums grid: enough workgroups for the chip whatever the batch (8 per CU over the launch; 64 per window starved a
    // one-window launch: 90 us for 20 MB, profiles/r03_aux_kernel_stats.md), at least 16 KB of samples each
    int sums_chunks(int nb) const {
        static const int per_cu = [] { const char* e = getenv("TWX_SUMS_WGS"); return e ? std::max(1, atoi(e)) : 8; }();
        const long long want = ((long long)per_cu * ncu + nb - 1) / nb, cap = std::max<long long>(1, N / 4096);
        return (int)std::max<long long>(1, std::min<long long>(std::min(want, cap), TWX_SUMS_MAXCHUNKS));
    }
    int ex_blocks() const { return (std::max(bruit_len, 1) + 255) / 256; }
    int set_extra(int which, long long value) override {
        if (value < 0 || value == 1 || value > (1 << 20)) return fail(TWX_E_ARG, "estimator length: 0 (off) or 2 .. 1 048 576 samples");
        if (which == 1 && value > 0 && N2 > 256 * TWX_SQN_EMAX) return fail(TWX_E_ARG, "TWX_OPT_NOISE_SQUARE_LEN: rows of more than 10 240 points are not supported");
        if (int rc = sync_all()) return rc;
        (which == 0 ? bruit_len : sq_len) = (int)value;
        HIPCHK(hipSetDevice(dev));
        for (int k = 0; k < nslots; ++k) {
            Slot& q = slots[k];
            if (q.ex_part_b) { dfree(q.ex_part_b); q.ex_part_b = nullptr; }
            if (bruit_len > 0) { if (int rc = dalloc(&q.ex_part_b, (size_t)B * ex_blocks() * 3)) return rc; }
            if (sq_len > 0 && !q.ex_part_s) { if (int rc = dalloc(&q.ex_part_s, (size_t)B * N1 * 2)) return rc; }
            if (!q.ex_sqmax) { if (int rc = dalloc(&q.ex_sqmax, (size_t)B)) return rc; }
        }
        return TWX_OK;
    }
    bool extras_on() const { return bruit_len > 0 || sq_len > 0; }
    // the records of a call start at index 0 of extra_dev: wait for the context's earlier work (it may still write there), grow if needed
    int extra_begin(long long nrec) {
        ex_off = -1; 
        if (!extras_on() || nrec <= 0) { extra_n = 0; return TWX_OK; }
        if (int rc = sync_all()) return rc;
        if ((size_t)nrec > extra_cap) {
            if (extra_dev) dfree(extra_dev);
            extra_dev = nullptr; extra_cap = 0;
            const size_t cap = (size_t)nrec + (size_t)nrec / 4 + 64;
            if (int rc = dalloc(&extra_dev, cap)) return rc;
            extra_cap = cap;
        }
        HIPCHK(hipMemset(extra_dev, 0xff, sizeof(twx_extra) * (size_t)nrec));        // (all-ones doubles are NaN: records a short call never reaches)
        HIPCHK(hipStreamSynchronize(nullptr));    // the context's streams are non-blocking: nothing orders them behind the null stream but the host
        extra_n = nrec;
        return TWX_OK;
    }
    int set_resample(double vitesse, double t0, long long dt) override {
        if (!(vitesse == vitesse) || !(t0 == t0) || fabs(vitesse) >= 0.5 || fabs(t0) >= 2.0) return fail(TWX_E_ARG, "twx_set_resample: vitesse inside (-0.5, 0.5), t0 inside (-2, 2)");
        if (vitesse != 0.0 && fabs((double)N * vitesse / (1.0 - vitesse)) >= 1.0)
            return fail(TWX_E_ARG, "twx_set_resample: |N * vitesse| must stay under one sample per window (the script's t0 / dt bookkeeping assumes it: godual_ranging_OP_vitesse.m:70-71)");
        if (int rc = sync_all()) return rc;
        rs_v = vitesse; rs_t0 = t0; rs_dt = dt;
        return TWX_OK;
    }
    // The script's loop bookkeeping for the next nwin windows (:40-43, :68-71), in fp64 with its own expressions, uploaded in one piece:
    // a call with the option on first waits for the context's earlier work (the arrays of the previous call may still be read), which
    // costs this variant the overlap between calls and nothing else.  resample_end(n) moves the carried state past n windows.
    int resample_begin(long long nwin) {
        if (rs_v == 0.0 || nwin <= 0) return TWX_OK;
        if (int rc = sync_all()) return rc;
        rs_t0s.resize((size_t)nwin); rs_edges.resize((size_t)nwin); rs_dts.resize((size_t)nwin); rs_t0_after.resize((size_t)nwin); rs_dt_after.resize((size_t)nwin);
        const double n1 = (double)(N - 1), den = 1.0 - rs_v;
        double t0c = rs_t0; long long dtc = rs_dt;
        for (long long w = 0; w < nwin; ++w) {
            const double t0 = t0c;
            auto xq = [&](double n) { return (n * 1.0) / den + t0; };            // [0:N-1]*1/(1-vitesse)+t0
            int edge = 0;
            if (xq(0.0) < 0.0) edge |= 1;                                         // interp1 gives NaN outside [0, N-1]
            if (xq(n1) > n1) edge |= 2;
            if (xq(1.0) < 0.0 || xq(n1 - 1.0) > n1) edge |= 4;                    // more than the two edge samples: the whole map is NaN
            rs_t0s[(size_t)w] = t0; rs_edges[(size_t)w] = edge; rs_dts[(size_t)w] = (int)dtc;
            t0c = t0c + (double)N * rs_v;                                          // t0=t0+length(y)*vitesse
            if (t0c >= 1.0) { t0c -= 1.0; dtc -= 1; }
            if (t0c <= -1.0) { t0c += 1.0; dtc += 1; }
            rs_t0_after[(size_t)w] = t0c; rs_dt_after[(size_t)w] = dtc;
        }
        if ((size_t)nwin > rs_dev_cap) {
            if (rs_dev_t0) dfree(rs_dev_t0);
            if (rs_dev_edge) dfree(rs_dev_edge);
            if (rs_dev_dt) dfree(rs_dev_dt);
            rs_dev_t0 = nullptr; rs_dev_edge = rs_dev_dt = nullptr; rs_dev_cap = 0;
            const size_t cap = (size_t)nwin + (size_t)nwin / 4 + 64;
            if (int rc = dalloc(&rs_dev_t0, cap)) return rc;
            if (int rc = dalloc(&rs_dev_edge, cap)) return rc;
            if (int rc = dalloc(&rs_dev_dt, cap)) return rc;
            rs_dev_cap = cap;
        }
        HIPCHK(hipMemcpy(rs_dev_t0, rs_t0s.data(), sizeof(double) * nwin, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(rs_dev_edge, rs_edges.data(), sizeof(int) * nwin, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(rs_dev_dt, rs_dts.data(), sizeof(int) * nwin, hipMemcpyHostToDevice));
        return TWX_OK;
    }
    void resample_end(long long ndone) {
        rs_off = -1;
        if (rs_v == 0.0 || ndone <= 0 || (size_t)ndone > rs_t0_after.size()) return;
        rs_t0 = rs_t0_after[(size_t)ndone - 1]; rs_dt = rs_dt_after[(size_t)ndone - 1];
    }
    // value: 0 off, 1 on at the default tolerance (1e-5 relative), > 1 the tolerance in units of 1e-9
    int set_selfcheck(long long value) override {
        if (value <= 0) { (void)sync_all(); selfcheck_tol = 0.f; return TWX_OK; }
        const int R0 = (use_rowd && row->S == 3) ? row->R[0] : 1;
        if (!use_rowd || R0 == 1) return fail(TWX_E_ARG, "TWX_OPT_SELFCHECK needs the three-stage DIF/DIT row pass (rows of 4000 / 8000 points and plug-ins of that shape)");
        if (int rc = sync_all()) return rc;
        HIPCHK(hipSetDevice(dev));
        for (int k = 0; k < nslots; ++k) {
            Slot& q = slots[k];
            if (!q.chk_rows) { if (int rc = dalloc(&q.chk_rows, (size_t)B * N1 * TWX_CHK_SLOTS)) return rc; }
            if (!q.chk_flag) { if (int rc = dalloc(&q.chk_flag, (size_t)B)) return rc; HIPCHK(hipMemset(q.chk_flag, 0, sizeof(int) * B)); }
        }
        if (!chk_stat) { if (int rc = dalloc(&chk_stat, (size_t)2)) return rc; HIPCHK(hipMemset(chk_stat, 0, 8)); }
        selfcheck_tol = value == 1 ? 1e-5f : (float)((double)value * 1e-9);
        return TWX_OK;
    }
    int pipeline_depth() const override { return nslots; }
    int sync_all() override {
        for (int k = 0; k < nslots; ++k) HIPCHK(hipStreamSynchronize(slots[k].stream));
        return TWX_OK;
    }
    unsigned long long* stamps_dev = nullptr;   // TWX_STAMPS diagnostic builds
    double* fine_u = nullptr; int fine_M = 0;   // TWX_FLAG_FINE_FREQ
    double* csum_part = nullptr;                // per-window partial sums of |d|^2 (complex-double input)
    int argmax_norm1 = 0;                       // cblas_izamax arg-max for the current call (twx_caf_freqs_cdev)
    int io_threads = 8;                         // TWX_IO_THREADS: concurrent preads per chunk in twx_process_file
    size_t io_sub = (size_t)8 << 20;            // TWX_IO_SUB_MB: a reader hands its piece to the device in sub-pieces of this size (0: whole)
    std::atomic<int> h2d_failed{0};             // set by a reader thread whose copy to the device was refused
    int ncu = 256;
    int ntiles = 0, ntiles_inv = 0;          // column tiles per row: forward passes / last pass
    // captured one-batch launches (process()): key = everything of the call that the kernels' arguments hold
    struct GraphEntry { const void* in; long long nwin; int nch, ch, rm; long long lo, hi; const void* out; hipGraphExec_t exec; unsigned long long used; };
    std::vector<GraphEntry> graphs;
    unsigned long long graph_clock = 0;
    int graph_state = -1;                    // -1: ask the environment; 0: direct launches; 1: graphs
    bool graphs_enabled() {
        if (graph_state < 0) { const char* e = getenv("TWX_GRAPH"); graph_state = (e && atoi(e) != 0) ? 1 : 0; }
        return graph_state == 1;
    }

    template <typename U> int upload(U** dst, const std::vector<U>& src) {
        int rc = dalloc(dst, src.size());
        if (rc) return rc;
        HIPCHK(hipMemcpy(*dst, src.data(), src.size() * sizeof(U), hipMemcpyHostToDevice));
        return TWX_OK;
    }

    int make_tables() {
        std::vector<C> h;
        host_twiddle<T>(h, N1, 1, N1, -1); if (int rc = upload(&tw1, h)) return rc;
        {   // tc[k1][c] = exp(-2 pi i k1 c/N)
            const int W = col->W;
            h.resize((size_t)N1 * W);
            const long double tp2 = 2.0L * 3.14159265358979323846264338327950288L;
            for (int k1 = 0; k1 < N1; ++k1)
                for (int c = 0; c < W; ++c) {
                    const long double a = tp2 * (long double)((long long)k1 * c) / (long double)N;
                    h[(size_t)k1 * W + c] = mk<T>((T)cosl(a), (T)(-sinl(a)));
                }
            if (int rc = upload(&tcw, h)) return rc;
        }
        {
            int rr[4] = {row->R[0], row->R[1], row->R[2], row->R[3]};
            host_stage_tabs<T>(h, row->S, rr); if (int rc = upload(&stab_f, h)) return rc;
            int ri[4] = {1, 1, 1, 1};
            for (int i = 0; i < row->S; ++i) ri[i] = row->R[row->S - 1 - i];
            host_stage_tabs<T>(h, row->S, ri); if (int rc = upload(&stab_i, h)) return rc;
        }
        tshift = 1; while ((2ll << (2 * tshift)) <= N) ++tshift;     // 2^tshift ≈ sqrt(N)
        host_twiddle<T>(h, (N >> tshift) + 1, 1ll << tshift, N, -1); if (int rc = upload(&ta, h)) return rc;
        host_twiddle<T>(h, 1ll << tshift, 1, N, -1); if (int rc = upload(&tb, h)) return rc;
        std::vector<cpx<double>> hd;
        host_twiddle<double>(hd, N1, 1, N1, -1); if (int rc = upload(&tw1d, hd)) return rc;
        // interpolation phase ramps exp(+2 pi i rho k/(R N)), k signed, split k = k1 + N1*k2:
        //   ramp1[rho][k1] = exp(+2 pi i rho k1/(R N)),  ramp2[rho][k2] = exp(+2 pi i rho k2s/(R N2))
        const int RL = row->R[row->S - 1], NSL = N2 / RL;
        std::vector<C> r1((size_t)R * N1), hea((size_t)R * NSL), heb((size_t)R * 2 * RL);
        const long double tp = 2.0L * 3.14159265358979323846264338327950288L;
        for (int rho = 0; rho < R; ++rho) {
            for (int k1 = 0; k1 < N1; ++k1) {
                long double a = tp * (long double)(((__int128)rho * k1) % ((__int128)R * N)) / ((long double)R * (long double)N);
                r1[(size_t)rho * N1 + k1] = mk<T>((T)cosl(a), (T)sinl(a));
            }
            // exp(+2 pi i rho k2s/(R N2)), k2 = j + NSL*r, k2s = k2 - N2*[2*k2 >= N2]
            for (int j = 0; j < NSL; ++j) {
                long double a = tp * (long double)((long long)rho * j) / ((long double)R * (long double)N2);
                hea[(size_t)rho * NSL + j] = mk<T>((T)cosl(a), (T)sinl(a));
            }
            for (int w = 0; w < 2; ++w)
                for (int r = 0; r < RL; ++r) {
                    long long num = ((long long)rho * ((long long)NSL * r - (long long)w * N2)) % ((long long)R * N2);
                    long double a = tp * (long double)num / ((long double)R * (long double)N2);
                    heb[((size_t)rho * 2 + w) * RL + r] = mk<T>((T)cosl(a), (T)sinl(a));
                }
        }
        if (int rc = upload(&ea, hea)) return rc;
        if (int rc = upload(&eb, heb)) return rc;
        {   // DIF/DIT row pass tables (RowD, twx_fft.h)
            const char* e = getenv("TWX_ROWD");
            // (TWX_ROWD=0, the Stockham row pass, is a diagnostic of fp32 contexts: in complex double its middle pass does not fit
            // the registers, so that form is not built — twx_inst_row.hip)
            use_rowd = (row->rowd != nullptr) && (!e || atoi(e) != 0 || sizeof(T) == 8);
        }
        if (use_rowd) {
            const int Rr = row->R[row->S - 1], R0 = row->S == 3 ? row->R[0] : 1, NU = R0 * Rr;
            std::vector<C> t((size_t)Rr * Rr + 2 * (size_t)R0 * Rr);
            auto Wf = [&](long long num, long long den) {
                const long double a = tp * (long double)(num % den) / (long double)den;
                return mk<T>((T)cosl(a), (T)(-sinl(a)));
            };
            for (int q = 0; q < Rr; ++q) for (int x = 0; x < Rr; ++x) t[(size_t)q * Rr + x] = Wf((long long)x * q, (long long)Rr * Rr);
            for (int q = 0; q < R0; ++q) for (int x = 0; x < Rr; ++x) {
                t[(size_t)Rr * Rr + (size_t)q * Rr + x] = Wf((long long)x * q, N2);
                t[(size_t)Rr * Rr + (size_t)R0 * Rr + (size_t)q * Rr + x] = Wf((long long)x * q, N2 / Rr);
            }
            if (int rc = upload(&dtabs, t)) return rc;
            std::vector<C> wr((size_t)Rr);
            for (int j = 0; j < Rr; ++j) wr[(size_t)j] = Wf(j, Rr);
            if (int rc = upload(&wr_d, wr)) return rc;
            {
                std::vector<C> wm((size_t)Rr * Rr);
                for (int j = 0; j < Rr * Rr; ++j) wm[(size_t)j] = Wf(j, (long long)Rr * Rr);
                if (int rc = upload(&wm_d, wm)) return rc;
            }
            {
                const long long Mblk = (long long)N2 / R0;
                std::vector<C> vc((size_t)N1 * R0);
                for (int k1 = 0; k1 < N1; ++k1) for (int c = 0; c < R0; ++c) {
                    const long long num = (long long)(((__int128)k1 * c * Mblk) % N);
                    const long double a = tp * (long double)num / (long double)N;
                    vc[(size_t)k1 * R0 + c] = mk<T>((T)cosl(a), (T)sinl(a));
                }
                if (int rc = upload(&vc_d, vc)) return rc;
                std::vector<C> vw((size_t)N1 * 2 * Rr);
                for (int k1 = 0; k1 < N1; ++k1) for (int j = 0; j < Rr; ++j) {
                    const long double aa = tp * (long double)(((long long)k1 * j) % N) / (long double)N;
                    const long double ab = tp * (long double)(((long long)k1 * Rr * j) % N) / (long double)N;
                    vw[((size_t)k1 * 2 + 0) * Rr + j] = mk<T>((T)cosl(aa), (T)sinl(aa));
                    vw[((size_t)k1 * 2 + 1) * Rr + j] = mk<T>((T)cosl(ab), (T)sinl(ab));
                }
                if (int rc = upload(&vw_d, vw)) return rc;
            }
            std::vector<C> a1((size_t)R * NU), b1((size_t)R * 2 * Rr);
            for (int rho = 0; rho < R; ++rho) {
                for (int q0 = 0; q0 < R0; ++q0) for (int q1 = 0; q1 < Rr; ++q1) {
                    const long double a = tp * (long double)((long long)rho * (q0 + R0 * q1)) / ((long double)R * (long double)N2);
                    a1[(size_t)rho * NU + q0 * Rr + q1] = mk<T>((T)cosl(a), (T)sinl(a));
                }
                for (int w = 0; w < 2; ++w) for (int q2 = 0; q2 < Rr; ++q2) {
                    long long num = ((long long)rho * ((long long)NU * q2 - (long long)w * N2)) % ((long long)R * N2);
                    const long double a = tp * (long double)num / ((long double)R * (long double)N2);
                    b1[((size_t)rho * 2 + w) * Rr + q2] = mk<T>((T)cosl(a), (T)sinl(a));
                }
            }
            if (int rc = upload(&ea_d, a1)) return rc;
            if (int rc = upload(&eb_d, b1)) return rc;
        }
        if (int rc = upload(&ramp1, r1)) return rc;
        return TWX_OK;
    }

    // conj(fft(code)) in [k1][k2] layout, computed in fp64 when the fp64 plans exist
    int make_code_spectrum() {
        if (int rc = dalloc(&cspec, (size_t)N)) return rc;
        if (cfg.window == TWX_WIN_HAMMING) {
            // The C++ twin, the only user of the window (processing/CPP/main.cpp:717-719), takes the peak from the correlation with
            // the WINDOWED spectrum (:288-301) but wipes the code off yint = ifft(zero-padded FFT(y)) — no fcode in it (:319-332).
            // The identity of DESIGN §SNR turns that into three samples of the correlation with the UNWINDOWED replica around the
            // peak, so such a context keeps both spectra and runs the row pass a second time for the statistics (run_batch_in).
            if (int rc = dalloc(&cspec_plain, (size_t)N)) return rc;
            cfg.window = TWX_WIN_NONE;
            const int rc = build_code_spectrum(cspec_plain);
            cfg.window = TWX_WIN_HAMMING;
            if (rc) return rc;
        }
        return build_code_spectrum(cspec);
    }
    int build_code_spectrum(C* cspec) {
        const ColOps* c64 = find_col(N1, 1, col->W); const RowOps* r64 = find_row(N2, 1);
        const bool use64 = !std::is_same<T, double>::value && c64 && r64 && c64->W == col->W;
        if (std::is_same<T, double>::value || !use64) return code_spectrum_T<T>(cspec, col, row, tw1, stab_f, ta, tb, tcw, scale_pow2 == 1.0 ? 0.0 : scale_pow2);
        // temporary fp64 tables and buffers
        cpx<double>*t1 = nullptr, *t2 = nullptr, *tad = nullptr, *tbd = nullptr, *spec = nullptr;
        int rr[4] = {r64->R[0], r64->R[1], r64->R[2], r64->R[3]};
        std::vector<cpx<double>> h;
        host_twiddle<double>(h, N1, 1, N1, -1); if (int rc = upload(&t1, h)) return rc;
        host_stage_tabs<double>(h, r64->S, rr); if (int rc = upload(&t2, h)) return rc;
        host_twiddle<double>(h, (N >> tshift) + 1, 1ll << tshift, N, -1); if (int rc = upload(&tad, h)) return rc;
        host_twiddle<double>(h, 1ll << tshift, 1, N, -1); if (int rc = upload(&tbd, h)) return rc;
        if (int rc = dalloc(&spec, (size_t)N)) return rc;
        cpx<double>* tcd = nullptr;
        {
            const int W = c64->W;
            h.resize((size_t)N1 * W);
            const long double tp2 = 2.0L * 3.14159265358979323846264338327950288L;
            for (int k1 = 0; k1 < N1; ++k1)
                for (int c = 0; c < W; ++c) {
                    const long double a = tp2 * (long double)((long long)k1 * c) / (long double)N;
                    h[(size_t)k1 * W + c] = mk<double>((double)cosl(a), (double)(-sinl(a)));
                }
            if (int rc = upload(&tcd, h)) return rc;
        }
        if (int rc = code_spectrum_T<double>(spec, c64, r64, t1, t2, tad, tbd, tcd)) return rc;
        TWX_LAUNCH((k_convert<double, T>), dim3(1024), dim3(256), stream, spec, cspec, N, scale_pow2);   // range scale folded in
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(stream));
        dfree(t1); dfree(t2); dfree(tad); dfree(tbd); dfree(tcd); dfree(spec);
        return TWX_OK;
    }
    template <typename U>
    int code_spectrum_T(cpx<U>* out, const ColOps* c, const RowOps* r, const cpx<U>* t1, const cpx<U>* t2,
                        const cpx<U>* tad, const cpx<U>* tbd, const cpx<U>* tcu, double store_scale = 0.0) {
        cpx<U>* tmp = nullptr;
        if (int rc = dalloc(&tmp, (size_t)N)) return rc;
        ColFwdArgs<U> ca{};
        ca.in_win_stride = 0; ca.sums = nullptr; ca.remove_mean = 0; ca.n = N; ca.n2 = N2; ca.ntiles = N2 / c->W; ca.nwin = 1;
        ca.e1 = nullptr; ca.e2 = nullptr; ca.tw1 = t1; ca.ta = tad; ca.tb = tbd; ca.tshift = tshift; ca.tc = tcu; ca.out = tmp;
        if (c->fwd(COL_PLAIN, IN_CHIPS, chips_dev, cfg.sps, &ca, (unsigned)ca.ntiles, stream)) return fail(TWX_E_HIP, "code col pass launch failed");
        RowArgs<U> ra{};
        ra.n = N; ra.n1 = N1; ra.nwin = 1; ra.A = tmp; ra.wshift = wshift_of(c->W); ra.stab_f = t2; ra.stab_i = t2; ra.spec_out = out;
        ra.conj_out = 1; ra.hamming = (cfg.window == TWX_WIN_HAMMING); ra.scale = (U)store_scale;
        if (r->run(ROW_STORE, &ra, (unsigned)N1, stream)) return fail(TWX_E_HIP, "code row pass launch failed");
        HIPCHK(hipStreamSynchronize(stream));
        dfree(tmp);
        return TWX_OK;
    }

    ~Ctx() override {
        for (auto& e : graphs) (void)hipGraphExecDestroy(e.exec);
        for (int k = 0; k < 4; ++k) { if (stage[k].host) pin_free(stage[k].host); if (stage[k].dev) (void)hipFree(stage[k].dev); }
        for (int k = 1; k < nslots; ++k) if (slots[k].stream) (void)hipStreamDestroy(slots[k].stream);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        for (auto e : ev_join) if (e) (void)hipEventDestroy(e);
        stream = slots[0].stream ? slots[0].stream : stream;   // base class destroys slot 0's stream
    }
    int init() override {
        memset(slots, 0, sizeof slots);
        profile = (cfg.flags & TWX_FLAG_PROFILE) != 0;
        HIPCHK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        twx::fence_register(dev, stream);          // (slots 1.. fork from and join this stream inside every call)
        R = cfg.nphase > 0 ? cfg.nphase : 2 * cfg.nint + 1;
        ntiles = N2 / col->W; ntiles_inv = N2 / colinv->W;
        if (cfg.max_batch > 0) B = cfg.max_batch;
        else {   // largest power of two whose A+Bz buffers stay under ~1.5 GiB; at most 16 windows per launch, up to 256
                 // for windows under 64 k samples where a launch would otherwise be shorter than its own overhead
                 // (tools/small_n.py: N = 20 000 runs 9.8 Gsample/s at 16, 28 at 64, 36 at 256)
            const long long fit = std::max<long long>(1, (1536ll << 20) / (N * (long long)sizeof(C) * (1 + R)));
            int cap = 16;
            if (N < 65536) { cap = 32; while (cap < 256 && (long long)cap * N < 5000000) cap *= 2; }   // ~5 M samples per launch
            B = 1; while (B * 2 <= fit && B * 2 <= cap) B *= 2;
        }
        // range safety: the unnormalised correlation peak reaches ~N^2*32768 — keep |z|^2 inside fp32
        int e = 0; while ((1ll << e) < N) ++e;
        scale_pow2 = std::is_same<T, float>::value ? ldexp(1.0, -e) : 1.0;
        // code
        if (int rc = dalloc(&chips_dev, (size_t)cfg.n_chips)) return rc;
        if (cfg.chips) {
            for (long long i = 0; i < cfg.n_chips; ++i)
                if (cfg.chips[i] > 1) return fail(TWX_E_ARG, "chips must be bytes 0/1");
            HIPCHK(hipMemcpy(chips_dev, cfg.chips, (size_t)cfg.n_chips, hipMemcpyHostToDevice));
        } else {
            if (int rc = lfsr_to_device(cfg.lfsr_bitlen, (unsigned)cfg.lfsr_taps, cfg.n_chips, chips_dev)) return rc;
        }
        if (int rc = make_tables()) return rc;
        if (int rc = make_code_spectrum()) return rc;
        {   // replica variants (header: chips_q, code_levels, TWX_FLAG_CODE_ZERO_MEAN)
            const int unipolar = cfg.code_levels == TWX_CODE_UNIPOLAR, zero_mean = (cfg.flags & TWX_FLAG_CODE_ZERO_MEAN) != 0;
            if (cfg.code_levels != TWX_CODE_BIPOLAR && cfg.code_levels != TWX_CODE_UNIPOLAR) return fail(TWX_E_ARG, "bad code_levels");
            snr_valid = !(unipolar || zero_mean || cfg.chips_q);
            if (!snr_valid) {
                if (cfg.window != TWX_WIN_NONE) return fail(TWX_E_ARG, "the Hamming-windowed replica is only defined for the +-1 code");
                if (cfg.chips_q && !cfg.chips) return fail(TWX_E_ARG, "chips_q needs chips");
                C* cq = nullptr;
                if (cfg.chips_q) {
                    for (long long i = 0; i < cfg.n_chips; ++i)
                        if (cfg.chips_q[i] > 1) return fail(TWX_E_ARG, "chips_q must be bytes 0/1");
                    unsigned char* ci_dev = chips_dev; unsigned char* cq_dev = nullptr;
                    if (int rc = dalloc(&cq_dev, (size_t)cfg.n_chips)) return rc;
                    HIPCHK(hipMemcpy(cq_dev, cfg.chips_q, (size_t)cfg.n_chips, hipMemcpyHostToDevice));
                    C* ci_spec = cspec; cspec = nullptr; chips_dev = cq_dev;
                    int rc = make_code_spectrum();            // allocates a fresh cspec for the quadrature chips
                    cq = cspec; cspec = ci_spec; chips_dev = ci_dev;
                    dfree(cq_dev);
                    if (rc) return rc;
                }
                // spectrum values carry scale_pow2; DC of an all-ones window is N (the conj is real)
                TWX_LAUNCH((k_code_variant<T>), dim3(1024), dim3(256), stream, cspec, (const C*)cq, (long long)N, unipolar, zero_mean,
                           0.5 * (double)N * scale_pow2);
                HIPCHK(hipGetLastError());
                HIPCHK(hipStreamSynchronize(stream));
                if (cq) dfree(cq);
            }
        }
        if (use_rowd) {
            if (int rc = dalloc(&cspec_perm, (size_t)N)) return rc;
            const int Rr = row->R[row->S - 1], R0 = row->S == 3 ? row->R[0] : 1;
            TWX_LAUNCH((k_cspec_perm<T>), dim3(N1), dim3(256), stream, cspec, cspec_perm, N1, N2, R0, Rr);
            HIPCHK(hipGetLastError());
            if (cspec_plain) {
                if (int rc = dalloc(&cspec_perm_plain, (size_t)N)) return rc;
                TWX_LAUNCH((k_cspec_perm<T>), dim3(N1), dim3(256), stream, cspec_plain, cspec_perm_plain, N1, N2, R0, Rr);
                HIPCHK(hipGetLastError());
            }
        }
        // batch buffers, one set per pipeline slot
        {
            const char* e = getenv("TWX_STREAMS");
            nslots = e ? std::max(1, std::min(4, atoi(e))) : 3;
            if (profile) nslots = 1;   // per-kernel HIP-event timing is only meaningful when kernels do not overlap
        }
        if (cfg.flags & TWX_FLAG_FINE_FREQ) {
            const long long third = (long long)floor(cfg.fs / 3.0);        // int(fs//3)
            if (N < third) return fail(TWX_E_ARG, "TWX_FLAG_FINE_FREQ needs a window of at least fs/3 samples (godual_ranging.py:26)");
            fine_M = (int)((third + 9) / 10);
        }
        for (int k = 0; k < nslots; ++k) {
            Slot& q = slots[k];
            memset(&q, 0, sizeof q);
            if (k == 0) { q.stream = stream; HIPCHK(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming)); }
            else {
                HIPCHK(hipStreamCreateWithFlags(&q.stream, hipStreamNonBlocking));
                HIPCHK(hipEventCreateWithFlags(&ev_join[k], hipEventDisableTiming));
            }
            if (int rc = dalloc(&q.sums, (size_t)B)) return rc;
            if (int rc = dalloc(&q.dfv, (size_t)B)) return rc;
            if (int rc = dalloc(&q.dfidx, (size_t)B)) return rc;
            if (int rc = dalloc(&q.e1, (size_t)B * N1)) return rc;
            if (int rc = dalloc(&q.e2, (size_t)B * N2)) return rc;
            if (int rc = dalloc(&q.A, (size_t)B * N)) return rc;
            if (int rc = dalloc(&q.Bz, (size_t)B * R * N)) return rc;
            if (int rc = dalloc(&q.dc, (size_t)B)) return rc;
            if (int rc = dalloc(&q.part_band, (size_t)B * N1)) return rc;
            if (int rc = dalloc(&q.part_peak, (size_t)B * R * ntiles_inv)) return rc;
            if (int rc = dalloc(&q.res_dev, (size_t)B * TWX_MAX_CHANNELS)) return rc;   // all-channel mode: B windows x channels
            if (fine_M) { if (int rc = dalloc(&q.fine_u, (size_t)B * fine_M)) return rc; }
            if (int rc = dalloc(&q.csum_part, (size_t)B * 64)) return rc;
            if (int rc = dalloc(&q.sum_parts, (size_t)2 * B * TWX_SUMS_MAXCHUNKS)) return rc;
        }
        use_slot(0);
        {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount;
            const char* io = getenv("TWX_IO_THREADS");
            if (io) io_threads = std::max(1, std::min(128, atoi(io)));
            else {
                // one pread out of the page cache runs at 1 - 2 GB/s, the link behind it at 56: as many readers as a quarter of the CPUs
                // this process may use, 8 ... 32 (profiles/r06_io_rate.txt)
                cpu_set_t cs; CPU_ZERO(&cs);
                const int ncpu = sched_getaffinity(0, sizeof cs, &cs) == 0 ? CPU_COUNT(&cs) : (int)std::thread::hardware_concurrency();
                io_threads = std::max(8, std::min(32, ncpu / 4));
            }
            if (const char* sb = getenv("TWX_IO_SUB_MB")) io_sub = (size_t)std::max(0, atoi(sb)) << 20;
        }
#ifdef TWX_STAMPS
        if (int rc = dalloc(&stamps_dev, (size_t)B * N1 * 8 * 32)) return rc;
        HIPCHK(hipMemset(stamps_dev, 0, (size_t)B * N1 * 8 * 32 * 8));
#endif
        HIPCHK(hipStreamSynchronize(stream));
        return TWX_OK;
    }

    int lfsr_to_device(int bitlen, unsigned taps, long long n, unsigned char* out) {
        if (bitlen < 2 || bitlen > 32 || taps == 0 || (bitlen < 32 && (taps >> bitlen))) return fail(TWX_E_ARG, "bad LFSR parameters");
        // transition matrices M^(2^j) as columns (image of each basis state)
        std::vector<unsigned> jump(40 * 32, 0);
        unsigned m[32];
        for (int i = 0; i < bitlen; ++i) {
            unsigned s = 1u << i;
            unsigned bit = __builtin_popcount(s & taps) & 1u;
            m[i] = (s >> 1) | (bit << (bitlen - 1));
        }
        for (int j = 0; j < 40; ++j) {
            for (int i = 0; i < bitlen; ++i) jump[j * 32 + i] = m[i];
            unsigned sq[32];
            for (int i = 0; i < bitlen; ++i) {   // sq = m∘m
                unsigned s = m[i], ns = 0;
                for (int b = 0; b < bitlen; ++b) if ((s >> b) & 1u) ns ^= m[b];
                sq[i] = ns;
            }
            memcpy(m, sq, sizeof m);
        }
        unsigned* jd = nullptr;
        if (int rc = dalloc(&jd, jump.size())) return rc;
        HIPCHK(hipMemcpy(jd, jump.data(), jump.size() * sizeof(unsigned), hipMemcpyHostToDevice));
        const long long seg = 256;
        const long long nthreads = (n + seg - 1) / seg;
        TWX_LAUNCH(k_lfsr, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), stream, bitlen, taps, n, seg, jd, out);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(stream));
        dfree(jd);
        return TWX_OK;
    }

    // one batch of nb windows starting at `in` (short2 units: window stride N*nch, channel offset applied)
    int run_batch(const short2* in, int nb, int nch, const twx_band* band, const double* df_host, twx_result* out_dev,
                  C* zout /*optional full map, nb must be 1*/, bool same_window = false, int res_stride = 1, double zscale = 0.0) {
        return run_batch_in(IN_I16, in, nullptr, nch, same_window ? 0 : (long long)N * nch, remove_mean, nb, band, df_host, out_dev, zout, res_stride, zscale);
    }
    // Both channels of nb windows of a two-channel capture ([I1 Q1 I2 Q2] frames, 16-byte aligned) on the CURRENT slot: one
    // pass over the frames takes both channels' statistics and writes planar copies (k_sums_deint2), then the chain runs
    // per channel on 4-byte samples.  df2: per-channel pointers or null.  Records go to out_dev[w*2 + c].
    static bool frames_ok_2ch(const void* frames) {
        static const bool off = getenv("TWX_NO_DEINT") != nullptr;      // experiments: the strided path
        return !off && (reinterpret_cast<unsigned long long>(frames) & 15ull) == 0;
    }
    int run_batch_2ch(int slot, const short2* frames, int nb, const twx_band* band, const double* const df2[2], twx_result* out_dev) {
        Slot& q = slots[slot];
        if (!q.planar) {
            if (int rc = dalloc(&q.planar, (size_t)2 * B * N)) return rc;
            if (int rc = dalloc(&q.sums2, (size_t)2 * B)) return rc;
        }
        {
            ProfScope ps(this, PC_SUMS, 2ll * nb * N);
            const int chunks = sums_chunks(nb);
            TWX_LAUNCH((k_sums_deint2<0>), dim3(chunks, nb), dim3(256), stream, reinterpret_cast<const int4*>(frames), (long long)N, (long long)N,
                       q.planar, q.planar + (size_t)B * N, q.sum_parts, q.sum_parts + (size_t)B * TWX_SUMS_MAXCHUNKS);
            HIPCHK(hipGetLastError());
            TWX_LAUNCH((k_sums_final<0>), dim3(nb, 2), dim3(256), stream, q.sum_parts, chunks, (long long)B * TWX_SUMS_MAXCHUNKS, q.sums2, q.sums2 + B);
            HIPCHK(hipGetLastError());
        }
        int rc = TWX_OK;
        WinSums* keep = sums;
        for (int c = 0; c < 2 && rc == TWX_OK; ++c) {
            sums = q.sums2 + (size_t)c * B; sums_ready = true;
            rc = run_batch(q.planar + (size_t)c * B * N, nb, 1, band, df2 ? df2[c] : nullptr, out_dev + c, nullptr, false, 2);
        }
        sums = keep; sums_ready = false;
        return rc;
    }
    // The (q1, q2) digit pairs of the row bins k2 = q0 + R0 q1 + R0 R q2 that a search band can touch (k = k1 + N1 k2, every k1
    // and q0): at most 8 pairs -> the pruned last stage of k_rowd<BAND>, otherwise 0 (full stage).  Rows with R0 > 1 only.
    void band_pairs(const twx_band* band, int* np, unsigned long long* pq1, unsigned long long* pq2) const {
        static const bool off = getenv("TWX_NO_PRUNE") != nullptr;
        *np = 0; *pq1 = 0; *pq2 = 0;
        if (off || !band || row->S != 3) return;
        const int R0 = row->R[0], Rr = row->R[2];
        const long long half = N / 2, cnt = band->k_hi - band->k_lo + 1;
        const long long kA = (band->k_lo + (N - half)) % N;          // shifted index i -> bin k = (i + N - half) mod N
        const long long nk2 = (kA % N1 + cnt + N1 - 1) / N1;          // row bins touched, contiguous modulo N2
        if (nk2 > 64 * (long long)R0) return;
        int n = 0; unsigned long long a1 = 0, a2 = 0;
        for (long long j = 0; j < nk2; ++j) {
            const long long k2 = (kA / N1 + j) % N2, m = k2 / R0;
            const unsigned q1 = (unsigned)(m % Rr), q2 = (unsigned)(m / Rr);
            bool seen = false;
            for (int p = 0; p < n; ++p) if (((a1 >> (8 * p)) & 0xff) == q1 && ((a2 >> (8 * p)) & 0xff) == q2) seen = true;
            if (seen) continue;
            if (n == 8) return;
            a1 |= (unsigned long long)q1 << (8 * n); a2 |= (unsigned long long)q2 << (8 * n); ++n;
        }
        if (R0 * n > row->NT) return;
        *np = n; *pq1 = a1; *pq2 = a2;
    }
    // intype IN_I16: p0 = short2 samples, aux = channels per sample, wstride in short2;  IN_C64S: p0/p1 = real/imaginary
    // doubles, aux = element stride, wstride in doubles (the mean-removed complex `d` of processing(d,k), godual_ranging.m:12)
    // zscale != 0 (with zout): MAP-ONLY call — the values of the map are written times zscale, no peak record is formed
    // (k_peak skipped) and, without mean removal, no window statistics are taken (k_sums skipped): the x2 interpolation of
    // short2double (rxcomplex.cpp:914-963) is such a call once per second and channel.
    int run_batch_in(int intype, const void* p0, const void* p1, int aux, long long wstride, int rm_mean, int nb, const twx_band* band,
                     const double* df_host, twx_result* out_dev, C* zout, int res_stride, double zscale = 0.0) {
        const bool map_only = zout && zscale != 0.0;
        const short2* in = reinterpret_cast<const short2*>(p0);
        const int nch = aux;
        SplitPtr sp{reinterpret_cast<const double*>(p0), reinterpret_cast<const double*>(p1)};
        const void* colin = intype == IN_C64S ? static_cast<const void*>(&sp) : p0;
        if (intype == IN_I16 && sums_ready) {
            // statistics taken by the de-interleaving pre-pass (run_batch_2ch)
        } else if (map_only && !rm_mean) {
            // nothing reads the window statistics
        } else if (intype == IN_I16) {
            ProfScope ps(this, PC_SUMS, (long long)nb * N);
            const int chunks = sums_chunks(nb);
            for (int it = 0, ne = reps(PC_SUMS); it < ne; ++it)
            TWX_LAUNCH((k_sums<0>), dim3(chunks, nb), dim3(256), stream, in, wstride, nch, N, sum_parts);
            HIPCHK(hipGetLastError());
            TWX_LAUNCH((k_sums_final<0>), dim3(nb, 1), dim3(256), stream, sum_parts, chunks, 0ll, sums, sums);
            HIPCHK(hipGetLastError());
        } else if (intype == IN_C32) {
            if (rm_mean) return fail(TWX_E_ARG, "complex input is taken as it is");
            HIPCHK(hipMemsetAsync(sums, 0, sizeof(WinSums) * nb, stream));        // no power statistics on this path
        } else {
            if (rm_mean) return fail(TWX_E_ARG, "complex input is taken as it is (the caller removed the mean, godual_ranging.m:80)");
            ProfScope ps(this, PC_SUMS, (long long)nb * N);
            TWX_LAUNCH((k_sums_c64<0>), dim3(64, nb), dim3(256), stream, InCplxSplit{sp.re, sp.im, aux}, wstride, (long long)N, csum_part);
            HIPCHK(hipGetLastError());
            TWX_LAUNCH((k_sums_c64_final<0>), dim3(nb), dim3(64), stream, csum_part, 64, sums);
            HIPCHK(hipGetLastError());
        }
        ColFwdArgs<T> ca{};
        ca.in_win_stride = wstride; ca.sums = sums; ca.remove_mean = rm_mean; ca.n = N; ca.n2 = N2; ca.ntiles = ntiles; ca.nwin = nb;
        ca.e1 = e1; ca.e2 = e2; ca.tw1 = tw1; ca.ta = ta; ca.tb = tb; ca.tshift = tshift; ca.tc = tcw; ca.out = A;
        RowArgs<T> ra{};
        ra.n = N; ra.n1 = N1; ra.nwin = nb; ra.A = A; ra.wshift = wshift_of(col->W); ra.stab_f = stab_f; ra.stab_i = stab_i; ra.ea = ea; ra.eb = eb;
        ra.part = part_band; ra.cspec = cspec; ra.ramp1 = ramp1; ra.nphase = R; ra.scale = (T)scale_pow2;
        ra.ta = ta; ra.tb = tb; ra.tshift = tshift; ra.Bz = Bz; ra.dc = dc;
        ra.stamps = stamps_dev;
        {
            // k_rowd<MID>: workgroups in the launch.  Two are resident per CU (LDS); five per CU = 3.9 rows each balance the tail
            // better than two per CU with 9 or 10 rows each (profiles/r03_rowd_resident.txt: 0.326 against 0.334 ms).
            static const int pf = [] { const char* e = getenv("TWX_ROW_PF"); return e ? atoi(e) : -1; }();      // experiments: 0 = one workgroup per row
            // a launch of no more rows than 1.5 x the resident slots (one window: 625 rows on 512 slots) runs as one resident workgroup
            // per slot — the first 113 take a second row while the others' CUs drain — instead of one workgroup per row in two uneven
            // waves: 28.3 -> 29.4 Gsample/s for the one-window chain (profiles/r04_b1_pf.txt)
            ra.pf_stride = pf >= 0 ? pf : ((long long)N1 * nb <= 3ll * ncu ? 2 * ncu : 5 * ncu);
        }
        if (band) {
            if (band->k_lo < 0 || band->k_hi >= N || band->k_lo > band->k_hi) return fail(TWX_E_ARG, "band outside 0..N-1");
            ra.band_lo = band->k_lo; ra.band_hi = band->k_hi;
            {
                ProfScope ps(this, PC_COL_SQ, (long long)nb * N);
                for (int it = 0, ne = reps(PC_COL_SQ); it < ne; ++it)
                if (col->fwd(COL_SQUARE, intype, colin, aux, &ca, (unsigned)(ntiles * nb), stream)) return fail(TWX_E_HIP, "k_col_fwd(square) launch failed");
            }
            {
                ProfScope ps(this, PC_ROW_BAND, (long long)nb * N);
                for (int it = 0, ne = reps(PC_ROW_BAND); it < ne; ++it)
                if (use_rowd) {
                    RowDArgs<T> rd{}; rd.r = ra; rd.dtabs = dtabs; rd.cspec_perm = cspec_perm; rd.ea_d = ea_d; rd.eb_d = eb_d;
                    rd.wr = wr_d; band_pairs(band, &rd.nprune, &rd.pr_q1, &rd.pr_q2);
                    // fp32, a band of at most 8 digit pairs: k_rowd_bandsum (the row never enters LDS).  TWX_BANDSUM=0: k_rowd<BAND> (A/B; read per
                    // call so that one process can compare the two forms)
                    const char* bs = getenv("TWX_BANDSUM");
                    rd.wm = (!bs || atoi(bs) != 0) ? wm_d : nullptr;
                    if (row->rowd(ROW_BAND, &rd, (unsigned)(N1 * nb), stream)) return fail(TWX_E_HIP, "k_rowd(band) launch failed");
                } else if (row->run(ROW_BAND, &ra, (unsigned)(N1 * nb), stream)) return fail(TWX_E_HIP, "k_row(band) launch failed");
            }
        } else if (df_host) {
            HIPCHK(hipMemcpyAsync(dfv, df_host, sizeof(double) * nb, hipMemcpyHostToDevice, stream));
        }                                            // else: the batch's df vector was written on the device (acquire_cdev)
        // A map-only call with the carrier the slot's tables already hold (the x2 interpolation of the receiver: df = 0, every
        // second, both channels) skips the table kernel: 8 625 fp64 sincospi to write the same ones again
        Slot* cur_slot = nullptr;
        for (int k = 0; k < nslots; ++k) if (slots[k].e1 == e1) cur_slot = &slots[k];
        const bool rs_on = rs_off >= 0 && intype == IN_I16 && rs_dev_t0 && !map_only && (size_t)(rs_off + nb) <= rs_t0s.size();
        if (rs_off >= 0 && !rs_on) return fail(TWX_E_STATE, "the velocity-compensated window applies to int16 captures through twx_process_windows[_dev] / twx_process_file");
        Slot* chk_slot = (selfcheck_tol > 0.f && use_rowd && cur_slot && cur_slot->chk_rows && !map_only) ? cur_slot : nullptr;
        const bool same_tables = map_only && df_host && cur_slot && cur_slot->tables_nb >= nb && cur_slot->tables_df == df_host[0] && !(cfg.flags & TWX_FLAG_FINE_FREQ);
        if (!same_tables) {
            ProfScope ps(this, PC_DFT, nb);
            for (int it = 0, ne = reps(PC_DFT); it < ne; ++it)
            TWX_LAUNCH((k_df_tables<T>), dim3(nb, df_slices(nb)), dim3(256), stream, band ? 1 : 0, part_band, N1, dfv, dfidx, cfg.fs,
                               (long long)N, N1, N2, e1, e2, (band && cur_slot) ? cur_slot->ex_sqmax : (double*)nullptr);
            HIPCHK(hipGetLastError());
            if (cur_slot) {
                bool uniform = df_host && !band;
                for (int i = 1; uniform && i < nb; ++i) uniform = df_host[i] == df_host[0];
                cur_slot->tables_nb = uniform ? nb : 0;
                cur_slot->tables_df = uniform ? df_host[0] : 0.0;
            }
        }
        // the SNR estimators of process_OP.m (optional): the bins behind the carrier peak, while A still holds the squared signal's column pass
        const bool ex_on = ex_off >= 0 && cur_slot && extra_dev && !map_only && (size_t)(ex_off + (long long)(nb - 1) * res_stride) < extra_cap;
        if (ex_off >= 0 && !ex_on) return fail(TWX_E_STATE, "the SNR estimators are not available on this entry point");
        ExtraArgs xa{};
        if (ex_on) {
            xa.n = N; xa.n1 = N1; xa.n2 = N2; xa.nphase = R; xa.bruit_len = bruit_len; xa.sq_len = sq_len;
            xa.part_b = cur_slot->ex_part_b; xa.nblk_b = ex_blocks(); xa.part_s = cur_slot->ex_part_s; xa.sqmax = band ? cur_slot->ex_sqmax : nullptr;
            xa.out = extra_dev + ex_off; xa.out_stride = res_stride;
            if (band && sq_len > 0) {
                TWX_LAUNCH((k_sq_noise<T>), dim3(N1, nb), dim3(256), stream, xa, (const C*)A, wshift_of(col->W), (const long long*)dfidx);
                HIPCHK(hipGetLastError());
            }
        }
        if (cfg.flags & TWX_FLAG_FINE_FREQ) {
            if (intype == IN_C32) return fail(TWX_E_ARG, "TWX_FLAG_FINE_FREQ is not available on complex-float input");
            if (intype == IN_I16) TWX_LAUNCH((k_fine_angle<InI16>), dim3(64, nb), dim3(256), stream, InI16{in, nch}, wstride, rm_mean, (long long)N, sums, dfv, cfg.fs, fine_M, fine_u);
            else TWX_LAUNCH((k_fine_angle<InCplxSplit>), dim3(64, nb), dim3(256), stream, InCplxSplit{sp.re, sp.im, aux}, wstride, rm_mean, (long long)N, sums, dfv, cfg.fs, fine_M, fine_u);
            HIPCHK(hipGetLastError());
            TWX_LAUNCH((k_fine_fit<0>), dim3(nb), dim3(1024), stream, fine_u, fine_M, cfg.fs, dfv);
            HIPCHK(hipGetLastError());
            TWX_LAUNCH((k_df_tables<T>), dim3(nb, df_slices(nb)), dim3(256), stream, 2, part_band, N1, dfv, dfidx, cfg.fs,
                               (long long)N, N1, N2, e1, e2, (double*)nullptr);       // rebuild the NCO tables for df + dfleftover
            HIPCHK(hipGetLastError());
        }
        {
            ProfScope ps(this, PC_COL_MIX, (long long)nb * N);
            ResamplePtr rp{};
            int mix_type = intype; const void* mix_in = colin;
            if (rs_on) {
                rp.iq = p0; rp.t0 = rs_dev_t0 + rs_off; rp.edge = rs_dev_edge + rs_off; rp.c = rs_v / (1.0 - rs_v);
                mix_type = IN_I16RS; mix_in = &rp;
            }
            for (int it = 0, ne = reps(PC_COL_MIX); it < ne; ++it)
            if (col->fwd(COL_MIX, mix_type, mix_in, aux, &ca, (unsigned)(ntiles * nb), stream)) return fail(TWX_E_HIP, "k_col_fwd(mix) launch failed");
        }
        {
            ProfScope ps(this, PC_ROW_MID, (long long)nb * N);
            for (int it = 0, ne = reps(PC_ROW_MID); it < ne; ++it)
            if (use_rowd) {
                RowDArgs<T> rd{}; rd.r = ra; rd.dtabs = dtabs; rd.cspec_perm = cspec_perm; rd.ea_d = ea_d; rd.eb_d = eb_d; rd.vc = vc_d; rd.vw = vw_d;
                rd.chk_rows = chk_slot ? chk_slot->chk_rows : nullptr; rd.chk_fault = dbg_fault;
                if (row->rowd(ROW_MID, &rd, (unsigned)(N1 * nb), stream)) return fail(TWX_E_HIP, "k_rowd(mid) launch failed");
            } else if (row->run(ROW_MID, &ra, (unsigned)(N1 * nb), stream)) return fail(TWX_E_HIP, "k_row(mid) launch failed");
        }
        if (chk_slot) {                                  // Parseval per row of the pass just enqueued -> the windows' status words
            TWX_LAUNCH((k_chk_verdict<0>), dim3(nb), dim3(256), stream, (const float*)chk_slot->chk_rows, N1, N2, R, selfcheck_tol, chk_slot->chk_flag, chk_stat);
            HIPCHK(hipGetLastError());
        }
        ColInvArgs<T> ia{};
        ia.n = N; ia.n2 = N2; ia.ntiles = ntiles_inv; ia.nphase = R; ia.nwin = nb; ia.Bz = Bz; ia.tw1 = tw1; ia.part = part_peak; ia.zout = zout; ia.norm1 = argmax_norm1; ia.zscale = (T)(map_only ? zscale : 1.0);
        {
            ProfScope ps(this, PC_COL_INV, (long long)nb * N);
            for (int it = 0, ne = reps(PC_COL_INV); it < ne; ++it)
            if (colinv->inv(&ia, (unsigned)(ntiles_inv * R * nb), stream)) return fail(TWX_E_HIP, "k_col_inv launch failed");
        }
        PeakArgs<T> pa{};
        pa.n = N; pa.n1 = N1; pa.n2 = N2; pa.nphase = R; pa.nparts = R * ntiles_inv; pa.part = part_peak; pa.Bz = Bz; pa.tw1d = tw1d;
        pa.sums = sums; pa.remove_mean = rm_mean; pa.dc = dc; pa.dfv = dfv; pa.dfidx = dfidx; pa.inv_scale = 1.0 / scale_pow2;
        pa.var_ddof = cfg.var_ddof; pa.snr_rot = cfg.snr_rot; pa.convention = cfg.convention; pa.snr_valid = snr_valid; pa.res = out_dev; pa.res_stride = res_stride;
        pa.chk_flag = chk_slot ? chk_slot->chk_flag : nullptr;
        pa.rs_dt = rs_on ? rs_dev_dt + rs_off : nullptr; pa.rs_edge = rs_on ? rs_dev_edge + rs_off : nullptr;
        if (!map_only) {
            ProfScope ps(this, PC_PEAK, nb);
            for (int it = 0, ne = reps(PC_PEAK); it < ne; ++it)
            TWX_LAUNCH((k_peak<T>), dim3(nb), dim3(1024), stream, pa);
            HIPCHK(hipGetLastError());
        }
        if (ex_on) {
            if (bruit_len > 0) {
                TWX_LAUNCH((k_offpeak<T>), dim3(xa.nblk_b, nb), dim3(256), stream, xa, (const C*)Bz, (const cpx<double>*)tw1d, (const twx_result*)out_dev, res_stride,
                           (int)cfg.convention, 1.0 / scale_pow2);
                HIPCHK(hipGetLastError());
            }
            TWX_LAUNCH((k_extra_final<0>), dim3(nb), dim3(256), stream, xa, (const twx_result*)out_dev, res_stride, (const long long*)dfidx);
            HIPCHK(hipGetLastError());
        }
        if (!map_only && cspec_plain && snr_valid) {
            // Hamming-window context (make_code_spectrum): the row pass once more on the same column-pass output with the unwindowed
            // spectrum, and the statistics from the twelve samples around the peak the first call found
            ra.cspec = cspec_plain;
            if (use_rowd) {
                RowDArgs<T> rd{}; rd.r = ra; rd.dtabs = dtabs; rd.cspec_perm = cspec_perm_plain; rd.ea_d = ea_d; rd.eb_d = eb_d; rd.vc = vc_d; rd.vw = vw_d;
                if (row->rowd(ROW_MID, &rd, (unsigned)(N1 * nb), stream)) return fail(TWX_E_HIP, "k_rowd(mid, unwindowed) launch failed");
            } else if (row->run(ROW_MID, &ra, (unsigned)(N1 * nb), stream)) return fail(TWX_E_HIP, "k_row(mid, unwindowed) launch failed");
            pa.snr_only = 1;
            TWX_LAUNCH((k_peak<T>), dim3(nb), dim3(1024), stream, pa);
            HIPCHK(hipGetLastError());
        }
        return TWX_OK;
    }

    // ch >= 0: one channel, records out_dev[w].  ch == TWX_ALL_CHANNELS: every channel of every window from the same
    // device copy, records out_dev[w*nch + c] and df[w*nch + c]
    int process(const void* iq_dev, long long nwin, int nch, int ch, const twx_band* band, const double* df,
                twx_result* out_dev) override {
        if (!band && !df) return fail(TWX_E_ARG, "either band or df must be given");
        const bool all = ch < 0;
        const int c_lo = all ? 0 : ch, c_hi = all ? nch : ch + 1, ostride = all ? nch : 1;
        std::vector<double> dfc;
        int k = 0;
        struct RsScope { long long& off; ~RsScope() { off = -1; } } rs_scope{rs_off};
        if (rs_v != 0.0) {
            if (all) return fail(TWX_E_ARG, "the velocity-compensated window (twx_set_resample) takes one channel at a time");
            if (int rc0 = resample_begin(nwin)) return rc0;
        }
        struct ExScope { long long& off; ~ExScope() { off = -1; } } ex_scope{ex_off};
        if (int rc0 = extra_begin(nwin * ostride)) return rc0;
        // One-batch calls with the carrier search on the device (the per-second callers: MEX form A, the receiver's loop) enqueue
        // nothing but kernels on slot 0's stream: the sequence is captured once per (buffers, band) and replayed as a hipGraph —
        // one submission instead of nine launches.  Anything the kernels' arguments hold is in the key or fixed for the context's
        // life (slot buffers, tables; twx_set_code_spectrum* rewrites the spectrum in place).
        if (graphs_enabled() && !profile && dbg_only < 0 && !stamps_dev && band && !df && !all && nwin <= B && rs_v == 0.0) {
            if (band->k_lo < 0 || band->k_hi >= N || band->k_lo > band->k_hi) return fail(TWX_E_ARG, "band outside 0..N-1");
            GraphEntry* g = nullptr;
            for (auto& e : graphs)
                if (e.in == iq_dev && e.nwin == nwin && e.nch == nch && e.ch == ch && e.rm == remove_mean && e.lo == band->k_lo && e.hi == band->k_hi &&
                    e.out == (const void*)out_dev) g = &e;
            if (!g) {
                use_slot(0);
                if (hipStreamBeginCapture(slots[0].stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                    const short2* base = reinterpret_cast<const short2*>(iq_dev) + ch;
                    const int rc = run_batch(base, (int)nwin, nch, band, nullptr, out_dev, nullptr, false, 1);
                    hipGraph_t gr = nullptr; hipGraphExec_t ex = nullptr;
                    const hipError_t e1 = hipStreamEndCapture(slots[0].stream, &gr);
                    if (rc == TWX_OK && e1 == hipSuccess && gr && hipGraphInstantiate(&ex, gr, nullptr, nullptr, 0) == hipSuccess) {
                        if (graphs.size() >= 8) {                                    // least recently used goes
                            size_t v = 0;
                            for (size_t i = 1; i < graphs.size(); ++i) if (graphs[i].used < graphs[v].used) v = i;
                            (void)hipGraphExecDestroy(graphs[v].exec); graphs.erase(graphs.begin() + (long)v);
                        }
                        graphs.push_back(GraphEntry{iq_dev, nwin, nch, ch, remove_mean, band->k_lo, band->k_hi, (const void*)out_dev, ex, 0});
                        g = &graphs.back();
                    } else {
                        (void)hipGetLastError();
                        graph_state = 0;                                             // this context launches directly from now on
                    }
                    if (gr) (void)hipGraphDestroy(gr);
                    if (rc != TWX_OK) return rc;
                } else { (void)hipGetLastError(); graph_state = 0; }
            }
            if (g) {
                g->used = ++graph_clock;
                HIPCHK(hipGraphLaunch(g->exec, slots[0].stream));
                return TWX_OK;
            }
        }
        // Everything is ordered against slot 0's stream (= twx_stream()): the other slots start after what the caller
        // enqueued there before this call (the producer of iq_dev), and slot 0 ends up waiting for all of them, so a
        // consumer of out_dev enqueued on twx_stream() after the call sees every record.
        const long long nbatches = ((nwin + B - 1) / B) * (c_hi - c_lo);
        const int nused = (int)std::min<long long>(nslots, nbatches);
        if (nused > 1) {
            HIPCHK(hipEventRecord(ev_fork, slots[0].stream));
            for (int j = 1; j < nused; ++j) HIPCHK(hipStreamWaitEvent(slots[j].stream, ev_fork, 0));
        }
        int rc = TWX_OK;
        const bool deint = all && nch == 2 && frames_ok_2ch(iq_dev) && !extras_on();
        std::vector<double> dfc1;
        for (long long w0 = 0; w0 < nwin && rc == TWX_OK; w0 += B) {
            const int nb = (int)std::min<long long>(B, nwin - w0);
            if (deint) {                                            // both channels of the batch on one slot, next batch on the next
                const double* d2[2] = {nullptr, nullptr};
                if (df) {
                    dfc.resize((size_t)nb); dfc1.resize((size_t)nb);
                    for (int i = 0; i < nb; ++i) { dfc[(size_t)i] = df[(w0 + i) * 2]; dfc1[(size_t)i] = df[(w0 + i) * 2 + 1]; }
                    d2[0] = dfc.data(); d2[1] = dfc1.data();
                }
                use_slot(k);
                rc = run_batch_2ch(k, reinterpret_cast<const short2*>(iq_dev) + w0 * N * 2, nb, band, df ? d2 : nullptr, out_dev + w0 * 2);
                k = (k + 1) % nslots;
                continue;
            }
            for (int c = c_lo; c < c_hi && rc == TWX_OK; ++c, k = (k + 1) % nslots) {
                const double* dfp = nullptr;
                if (df) {
                    if (all) { dfc.resize((size_t)nb); for (int i = 0; i < nb; ++i) dfc[(size_t)i] = df[(w0 + i) * nch + c]; dfp = dfc.data(); }
                    else dfp = df + w0;
                }
                use_slot(k);
                const short2* base = reinterpret_cast<const short2*>(iq_dev) + c;
                rs_off = rs_v != 0.0 ? w0 : -1;
                ex_off = extras_on() ? w0 * ostride + (all ? c : 0) : -1;
                rc = run_batch(base + w0 * N * nch, nb, nch, band, dfp, out_dev + w0 * ostride + (all ? c : 0), nullptr, false, ostride);
            }
        }
        use_slot(0);
        for (int j = 1; j < nused; ++j) {
            if (hipEventRecord(ev_join[j], slots[j].stream) != hipSuccess || hipStreamWaitEvent(slots[0].stream, ev_join[j], 0) != hipSuccess)
                if (rc == TWX_OK) rc = fail(TWX_E_HIP, "slot join failed");
        }
        if (rc == TWX_OK) resample_end(nwin);
        return rc;
    }

    int fft_forward(const double* in, double* out) override {
        cpx<double>* din = nullptr; C* tmp = nullptr; C* spec = nullptr;
        Scratch sc(this);
        if (int rc = sc.get(&din, (size_t)N)) return rc;
        if (int rc = sc.get(&tmp, (size_t)N)) return rc;
        if (int rc = sc.get(&spec, (size_t)N)) return rc;
        HIPCHK(hipMemcpy(din, in, (size_t)N * 16, hipMemcpyHostToDevice));
        ColFwdArgs<T> ca{};
        ca.n = N; ca.n2 = N2; ca.ntiles = ntiles; ca.nwin = 1; ca.tw1 = tw1; ca.ta = ta; ca.tb = tb; ca.tshift = tshift; ca.tc = tcw; ca.out = tmp;
        if (col->fwd(COL_PLAIN, IN_C64, din, 0, &ca, (unsigned)ntiles, stream)) return fail(TWX_E_HIP, "col pass launch failed");
        RowArgs<T> ra{};
        ra.n = N; ra.n1 = N1; ra.nwin = 1; ra.A = tmp; ra.wshift = wshift_of(col->W); ra.stab_f = stab_f; ra.stab_i = stab_i; ra.spec_out = spec;
        if (row->run(ROW_STORE, &ra, (unsigned)N1, stream)) return fail(TWX_E_HIP, "row pass launch failed");
        HIPCHK(hipStreamSynchronize(stream));
        std::vector<C> h((size_t)N);
        HIPCHK(hipMemcpy(h.data(), spec, (size_t)N * sizeof(C), hipMemcpyDeviceToHost));
        for (int k1 = 0; k1 < N1; ++k1)
            for (int k2 = 0; k2 < N2; ++k2) {
                const long long k = k1 + (long long)N1 * k2;
                out[2 * k] = (double)h[(size_t)k1 * N2 + k2].x; out[2 * k + 1] = (double)h[(size_t)k1 * N2 + k2].y;
            }
        return TWX_OK;
    }
    int code_spectrum(double* out) override {
        std::vector<C> h((size_t)N);
        HIPCHK(hipMemcpy(h.data(), cspec, (size_t)N * sizeof(C), hipMemcpyDeviceToHost));
        for (int k1 = 0; k1 < N1; ++k1)
            for (int k2 = 0; k2 < N2; ++k2) {
                const long long k = k1 + (long long)N1 * k2;
                out[2 * k] = (double)h[(size_t)k1 * N2 + k2].x / scale_pow2; out[2 * k + 1] = (double)h[(size_t)k1 * N2 + k2].y / scale_pow2;
            }
        return TWX_OK;
    }
    int xcorr_map(const int16_t* iq, int nch, int ch, double df, double* out) override {
        short2* din = nullptr; C* z = nullptr;
        Scratch sc(this);
        if (int rc = sc.get(&din, (size_t)N * nch)) return rc;
        if (int rc = sc.get(&z, (size_t)N * R)) return rc;
        HIPCHK(hipMemcpy(din, iq, (size_t)N * nch * 4, hipMemcpyHostToDevice));
        int rc = run_batch(din + ch, 1, nch, nullptr, &df, res_dev, z);
        if (rc) return rc;
        HIPCHK(hipStreamSynchronize(stream));
        std::vector<C> h((size_t)N * R);
        HIPCHK(hipMemcpy(h.data(), z, h.size() * sizeof(C), hipMemcpyDeviceToHost));
        const double nrm = 1.0 / scale_pow2 / ((double)N * R);
        for (size_t i = 0; i < h.size(); ++i) { out[2 * i] = (double)h[i].x * nrm; out[2 * i + 1] = (double)h[i].y * nrm; }
        return TWX_OK;
    }

    int caf_freqs(const int16_t* iq, int nch, int ch, const double* freqs, long long nf, twx_result* out) override {
        short2* din = nullptr;
        Scratch sc(this);
        if (int rc = sc.get(&din, (size_t)N * nch)) return rc;
        HIPCHK(hipMemcpy(din, iq, (size_t)N * nch * 4, hipMemcpyHostToDevice));
        for (long long f0 = 0; f0 < nf; f0 += B) {
            const int nb = (int)std::min<long long>(B, nf - f0);
            if (int rc = run_batch(din + ch, nb, nch, nullptr, freqs + f0, res_dev, nullptr, true)) return rc;
            HIPCHK(hipMemcpyAsync(out + f0, res_dev, sizeof(twx_result) * nb, hipMemcpyDeviceToHost, stream));
            HIPCHK(hipStreamSynchronize(stream));
        }
        return TWX_OK;
    }

    // d2 = fft(d.^2) of an L-sample chunk at a few bins (direct sums; L arbitrary)
    int sqspec_bins(const void* iq_dev, long long L, int nch, int ch, const long long* bins, int nb, double* out) override {
        if (L < 1 || nb < 1 || nb > 64) return fail(TWX_E_ARG, "sqspec_bins: need L >= 1 and 1..64 bins");
        long long hb[64];
        for (int i = 0; i < nb; ++i) { hb[i] = bins[i] % L; if (hb[i] < 0) hb[i] += L; }
        if (L >= (1ll << 32)) return fail(TWX_E_ARG, "sqspec_bins: chunk too long");
        // context-owned work area (slot 7): bins | sums | per-workgroup partials — no allocation per chunk of the tracked flow
        const unsigned grid = (unsigned)std::min<long long>(4ll * ncu, (L + 255) / 256);
        char* work = static_cast<char*>(scratch_slot(7, 512 + 1024 + (size_t)grid * nb * 16));
        if (!work) return TWX_E_NOMEM;
        long long* bd = reinterpret_cast<long long*>(work); double* acc = reinterpret_cast<double*>(work + 512); double* part = reinterpret_cast<double*>(work + 1536);
        HIPCHK(hipMemcpyAsync(bd, hb, sizeof(long long) * nb, hipMemcpyHostToDevice, stream));
        TWX_LAUNCH((k_sq_dft_bins<0>), dim3(grid), dim3(256), stream, reinterpret_cast<const short2*>(iq_dev) + ch, nch, L, bd, nb, part);
        HIPCHK(hipGetLastError());
        TWX_LAUNCH((k_sq_dft_final<0>), dim3(2 * nb), dim3(256), stream, part, (int)grid, 2 * nb, acc);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(out, acc, sizeof(double) * 2 * nb, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        return TWX_OK;
    }
    // |fft(d.^2)| of an L = M*N sample chunk over nk consecutive (signed) bins from k_lo: M decimated
    // N-point transforms with this context's plan, then the radix-M recombination of the band only
    int sqspec_band(const void* iq_dev, long long L, int nch, int ch, long long k_lo, long long nk, double* out) override {
        if (L < N || L % N) return fail(TWX_E_ARG, "sqspec_band: L must be a multiple of the context's window length");
        if (nk < 1 || nk > L) return fail(TWX_E_ARG, "sqspec_band: bad bin count");
        const long long M = L / N;
        if (M > 4096 || nch * M > 0x7fffffffll) return fail(TWX_E_ARG, "sqspec_band: chunk too long for this window length");
        // work buffers kept by the context (the `lo` flavour of the tracked flow calls this once per 2-s chunk: an 80-MB
        // allocation and release per call cost more than the transforms)
        C* spec = static_cast<C*>(scratch_slot(8, (size_t)(M * N) * sizeof(C)));
        double* mag = static_cast<double*>(scratch_slot(9, (size_t)nk * sizeof(double)));
        if (!spec || !mag) return TWX_E_NOMEM;
        if (int rc = sync_all()) return rc;
        use_slot(0);
        const short2* base = reinterpret_cast<const short2*>(iq_dev) + ch;
        for (long long r0 = 0; r0 < M; r0 += B) {
            const int nb = (int)std::min<long long>(B, M - r0);
            ColFwdArgs<T> ca{};
            ca.in_win_stride = nch; ca.sums = nullptr; ca.remove_mean = 0; ca.n = N; ca.n2 = N2; ca.ntiles = ntiles; ca.nwin = nb;
            ca.e1 = e1; ca.e2 = e2; ca.tw1 = tw1; ca.ta = ta; ca.tb = tb; ca.tshift = tshift; ca.tc = tcw; ca.out = A;
            if (col->fwd(COL_SQUARE, IN_I16, base + r0 * nch, (int)(nch * M), &ca, (unsigned)(ntiles * nb), stream)) return fail(TWX_E_HIP, "k_col_fwd(square) launch failed");
            RowArgs<T> ra{};
            ra.n = N; ra.n1 = N1; ra.nwin = nb; ra.A = A; ra.wshift = wshift_of(col->W); ra.stab_f = stab_f; ra.stab_i = stab_i; ra.spec_out = spec + r0 * N;
            if (row->run(ROW_STORE, &ra, (unsigned)(N1 * nb), stream)) return fail(TWX_E_HIP, "k_row(store) launch failed");
        }
        TWX_LAUNCH((k_sqspec_combine<T>), dim3((unsigned)((nk + 255) / 256)), dim3(256), stream, spec, (int)M, (long long)N, N1, N2, k_lo, nk, mag);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(out, mag, sizeof(double) * nk, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        return TWX_OK;
    }

    // ---- pipelined ingest shared by twx_process_file and twx_process_windows (host buffer) -----------------
    // read_at(dst, byte offset from the first window, length) -> bytes delivered (short only at the end of the
    // source).  Chunk c = B windows lives in pipeline slot c % nslots: a helper thread fills the slot's pinned buffer
    // (as io_threads concurrent pieces: one pread/memcpy runs at ~5 GB/s, PCIe takes ten times that), the H2D copy and
    // the kernels of the chunk go to the slot's stream, and the results of the chunk that used the slot before are
    // fetched when the slot comes round again.
    template <class ReadAt>
    int run_pipeline(ReadAt read_at, int nch, int ch, const twx_band* band, const double* df_arr, double df_const,
                     twx_result* out, long long max_windows, long long* n_done) {
        *n_done = 0;
        const size_t win_bytes = (size_t)N * nch * 4;
        Stage* st = stage;
        int rc = TWX_OK;
        const bool all = ch < 0;
        if (all && nch > TWX_MAX_CHANNELS) return fail(TWX_E_ARG, "too many channels for the all-channel mode");
        const int c_lo = all ? 0 : ch, c_hi = all ? nch : ch + 1, ostride = all ? nch : 1;
        struct RsScope { long long& off; ~RsScope() { off = -1; } } rs_scope{rs_off};
        if (rs_v != 0.0) {
            if (all) return fail(TWX_E_ARG, "the velocity-compensated window (twx_set_resample) takes one channel at a time");
            if (max_windows > (1ll << 22)) return fail(TWX_E_ARG, "twx_set_resample: give max_windows (at most 4 194 304 windows per call)");
            if (int rc0 = resample_begin(max_windows)) return rc0;
        }
        struct ExScope { long long& off; ~ExScope() { off = -1; } } ex_scope{ex_off};
        if (extras_on() && max_windows > (1ll << 22)) return fail(TWX_E_ARG, "SNR estimators: give max_windows (at most 4 194 304 windows per call)");
        if (int rc0 = extra_begin(max_windows * ostride)) return rc0;
        std::vector<double> dfs((size_t)B, df_const), dfc((size_t)B), dfc_b((size_t)B);
        for (int k = 0; k < nslots && rc == TWX_OK; ++k) {
            st[k].nb = 0;
            if (st[k].bytes >= win_bytes * B) continue;
            if (st[k].host) pin_free(st[k].host);
            if (st[k].dev) (void)hipFree(st[k].dev);
            st[k].host = nullptr; st[k].dev = nullptr; st[k].bytes = 0;
            if (!(st[k].host = pin_alloc(win_bytes * B))) rc = fail(TWX_E_NOMEM, "pinned staging allocation failed");
            else if (hipMalloc((void**)&st[k].dev, win_bytes * B) != hipSuccess) rc = fail(TWX_E_NOMEM, "device staging allocation failed");
            else st[k].bytes = win_bytes * B;
        }
        if (rc) return rc;
        auto drain = [&](int k) -> int {       // wait for slot k's batch and fetch its results
            if (st[k].nb == 0) return TWX_OK;
            if (hipStreamSynchronize(slots[k].stream) != hipSuccess) return fail(TWX_E_HIP, "stream synchronize failed");
            if (hipMemcpy(out + st[k].w0 * ostride, slots[k].res_dev, sizeof(twx_result) * st[k].nb * ostride, hipMemcpyDeviceToHost) != hipSuccess) return fail(TWX_E_HIP, "D2H copy failed");
            *n_done += st[k].nb;
            st[k].nb = 0;
            return TWX_OK;
        };
        auto start_read = [&](int k, long long chunk) {
            const long long first = chunk * B;
            const long long want = std::max<long long>(0, std::min<long long>(B, max_windows - first));
            char* dst = (char*)st[k].host;
            char* ddst = reinterpret_cast<char*>(st[k].dev);
            hipStream_t sk = slots[k].stream;
            const int nthr = io_threads, device = dev;
            const size_t sub = io_sub;
            std::atomic<int>* bad = &h2d_failed;
            return std::async(std::launch::async, [=]() -> long long {
                // every piece goes on to the device from the thread that read it (the link measured 56.6 GB/s for pinned copies, the
                // read-then-copy form of round 3 reached 35-41: profiles/r04_io_rate.txt); the chunk's kernels are enqueued on the
                // same stream after the last piece has been handed over (rd[k].get() in the loop below)
                auto to_device = [=](size_t lo, size_t got) {
                    (void)hipSetDevice(device);
                    if (hipMemcpyAsync(ddst + lo, dst + lo, got, hipMemcpyHostToDevice, sk) != hipSuccess) bad->store(1);
                };
                const size_t total = read_in_pieces(read_at, dst, (size_t)first * win_bytes, win_bytes * (size_t)want, nthr, to_device, sub);    // twx_workers.h
                return (long long)(total / win_bytes);              // whole windows only
            });
        };
        h2d_failed.store(0);
        std::future<long long> rd[4];
        struct Events {                          // destroyed on every exit path, exceptions included
            hipEvent_t e[4] = {}; bool used[4] = {};
            ~Events() { for (auto& x : e) if (x) { (void)hipEventSynchronize(x); (void)hipEventDestroy(x); } }
        } h2d;
        hipEvent_t* h2d_done = h2d.e;
        struct Readers {                         // never leave a helper thread writing into a pinned buffer behind
            std::future<long long>* r; int n;
            ~Readers() { for (int k = 0; k < n; ++k) if (r[k].valid()) { try { (void)r[k].get(); } catch (...) {} } }
        } readers{rd, nslots};
        for (int k = 0; k < nslots; ++k)
            if (hipEventCreateWithFlags(&h2d_done[k], hipEventDisableTiming) != hipSuccess) return fail(TWX_E_HIP, "hipEventCreate failed");
        long long next_chunk = 0;
        for (int k = 0; k < nslots; ++k) rd[k] = start_read(k, next_chunk++);
        long long w0 = 0;
        bool eof = false;
        int kprev = -1;
        for (int k = 0; rc == TWX_OK && !eof; k = (k + 1) % nslots) {
            const long long want = std::max<long long>(0, std::min<long long>(B, max_windows - w0));
            if (!rd[k].valid()) {                // single slot: nobody refilled this buffer while another slot ran
                if (h2d.used[k]) (void)hipEventSynchronize(h2d_done[k]);
                rd[k] = start_read(k, next_chunk++);
            }
            const int nb = (int)rd[k].get();
            if ((long long)nb < want || want == 0) eof = true;
            rc = drain(k);                       // results of the batch that used this slot nslots chunks ago
            if (rc) break;
            if (h2d_failed.load()) { rc = fail(TWX_E_HIP, "H2D copy failed"); break; }
            if (nb > 0) {
                use_slot(k);
                // the chunk is already on its way: the readers enqueued its pieces on this slot's stream
                (void)hipEventRecord(h2d_done[k], stream); h2d.used[k] = true;
                if (all && nch == 2 && frames_ok_2ch(st[k].dev) && !extras_on()) {         // one pass over the frames serves both channels
                    const double* d2[2] = {nullptr, nullptr};
                    if (!band) {
                        if (!df_arr) { d2[0] = d2[1] = dfs.data(); }
                        else {
                            dfc.resize((size_t)nb); dfc_b.resize((size_t)nb);
                            for (int i = 0; i < nb; ++i) { dfc[(size_t)i] = df_arr[(w0 + i) * 2]; dfc_b[(size_t)i] = df_arr[(w0 + i) * 2 + 1]; }
                            d2[0] = dfc.data(); d2[1] = dfc_b.data();
                        }
                    }
                    rc = run_batch_2ch(k, st[k].dev, nb, band, band ? nullptr : d2, slots[k].res_dev);
                } else
                for (int c = c_lo; c < c_hi && rc == TWX_OK; ++c) {        // same staged copy for every requested channel
                    const double* dfp = nullptr;
                    if (!band) {
                        if (!df_arr) dfp = dfs.data();
                        else if (!all) dfp = df_arr + w0;
                        else { for (int i = 0; i < nb; ++i) dfc[(size_t)i] = df_arr[(w0 + i) * nch + c]; dfp = dfc.data(); }
                    }
                    rs_off = rs_v != 0.0 ? w0 : -1;
                    ex_off = extras_on() ? w0 * ostride + (all ? c : 0) : -1;
                    rc = run_batch(st[k].dev + c, nb, nch, band, dfp, slots[k].res_dev + (all ? c : 0), nullptr, false, ostride);
                }
                st[k].w0 = w0; st[k].nb = nb;
                w0 += nb;
            }
            // refill the pinned buffer of the PREVIOUS slot: its H2D copy had a whole iteration to finish
            if (kprev >= 0 && kprev != k && !eof) { (void)hipEventSynchronize(h2d_done[kprev]); rd[kprev] = start_read(kprev, next_chunk++); }
            kprev = k;
        }
        for (int k = 0; k < nslots; ++k) if (rd[k].valid()) (void)rd[k].get();
        for (int k = 0; k < nslots; ++k) if (h2d.used[k]) (void)hipEventSynchronize(h2d_done[k]);
        for (int k = 0; k < nslots; ++k) { int r2 = drain(k); if (rc == TWX_OK) rc = r2; }
        use_slot(0);
        if (rc == TWX_OK) resample_end(w0);                       // the carried t0 / dt move past the windows that were there
        return rc;
    }

    int process_file(const char* path, int nch, int ch, long long skip, const twx_band* band, double df_const,
                     twx_result* out, long long max_windows, long long* n_done) override {
        *n_done = 0;
        struct File { FILE* f; ~File() { if (f) fclose(f); } } file{fopen(path, "rb")};
        FILE* f = file.f;
        if (!f) return fail(TWX_E_ARG, std::string("cannot open ") + path);
        const int fd = fileno(f);
        const off_t base_off = (off_t)skip * nch * 4;
        // TWX_FILE_MMAP=1 (A/B, profiles/r05_io_rate.txt): the capture mapped, the reader threads memcpy from the mapping instead
        // of pread — no copy_to_user, a minor fault per 4-KB page of a mapping this process touches for the first time
        // TWX_FILE_MMAP=2 (round 6): the same with MADV_POPULATE_READ on every piece before its memcpy — the page tables of the range filled in
        // one call instead of one minor fault per page
        static const int mmap_mode = [] { const char* e = getenv("TWX_FILE_MMAP"); return e ? atoi(e) : 0; }();
        const bool use_mmap = mmap_mode != 0;
        if (use_mmap) {
            struct stat sb;
            if (fstat(fd, &sb) == 0 && sb.st_size > base_off) {
                struct Mapping { void* p; size_t len; ~Mapping() { if (p != MAP_FAILED) munmap(p, len); } } mp{MAP_FAILED, (size_t)sb.st_size};
                mp.p = mmap(nullptr, mp.len, PROT_READ, MAP_SHARED, fd, 0);
                if (mp.p != MAP_FAILED) {
                    (void)madvise(mp.p, mp.len, MADV_SEQUENTIAL);
                    const char* src = static_cast<const char*>(mp.p) + base_off;
                    const size_t total = mp.len - (size_t)base_off;
                    const int mode = mmap_mode;
                    auto read_map = [src, total, mode](char* dst, size_t off, size_t len) -> size_t {
                        if (off >= total) return 0;
                        const size_t n = std::min(len, total - off);
#ifndef MADV_POPULATE_READ
#define MADV_POPULATE_READ 22
#endif
                        if (mode == 2) {
                            const size_t a0 = ((size_t)(src + off)) & ~(size_t)4095, a1 = ((size_t)(src + off + n) + 4095) & ~(size_t)4095;
                            (void)madvise((void*)a0, a1 - a0, MADV_POPULATE_READ);
                        }
                        memcpy(dst, src + off, n);
                        return n;
                    };
                    return run_pipeline(read_map, nch, ch, band, nullptr, df_const, out, max_windows, n_done);
                }
            }
        }
        // Round 6 (profiles/r06_io_rate.txt): a pread straight into a pinned slot — 160 MB of memory no cache holds — runs at 2 GB/s per thread
        // (the kernel's copy_to_user with ordinary stores: every destination line is first read), the same pread into a buffer that stays in
        // the reader's L2 at 10.  So a reader goes through a 1-MB bounce buffer of its own and moves each megabyte on with non-temporal
        // stores (no read of the destination): TWX_IO_BOUNCE_KB, 0 = the direct pread of rounds 3-5.
        static const size_t bounce_bytes = [] { const char* e = getenv("TWX_IO_BOUNCE_KB"); return (size_t)(e ? std::max(0, atoi(e)) : 1024) << 10; }();
        auto read_at = [fd, base_off](char* dst, size_t off, size_t len) -> size_t {
            size_t done = 0;
            if (bounce_bytes && (((size_t)dst) & 15) == 0) {
                // (the readers are short-lived threads: their buffers come from a process-wide free list, touched once)
                struct Pool {
                    std::mutex mu; std::vector<char*> free_list;
                    char* get() {
                        { std::lock_guard<std::mutex> g(mu); if (!free_list.empty()) { char* p = free_list.back(); free_list.pop_back(); return p; } }
                        void* p = nullptr;
                        if (posix_memalign(&p, 4096, bounce_bytes) != 0) return nullptr;
                        memset(p, 0, bounce_bytes);
                        return (char*)p;
                    }
                    void put(char* p) { std::lock_guard<std::mutex> g(mu); free_list.push_back(p); }
                };
                static Pool pool;
                struct Lease { Pool& pl; char* p; ~Lease() { if (p) pl.put(p); } } lease{pool, pool.get()};
                char* bb = lease.p;
                if (bb) {
                while (done < len) {
                    const size_t want = std::min(bounce_bytes, len - done);
                    size_t got = 0;
                    while (got < want) {
                        const ssize_t r = pread(fd, bb + got, want - got, base_off + (off_t)(off + done + got));
                        if (r <= 0) break;
                        got += (size_t)r;
                    }
                    typedef long long v2 __attribute__((vector_size(16), aligned(16)));
                    const size_t body = got & ~(size_t)15;
                    const v2* sp = reinterpret_cast<const v2*>(bb);
                    v2* dp = reinterpret_cast<v2*>(dst + done);
                    for (size_t i = 0; i < body / 16; ++i) __builtin_nontemporal_store(sp[i], dp + i);
                    if (got > body) memcpy(dst + done + body, bb + body, got - body);
                    done += got;
                    if (got < want) break;
                }
                __builtin_ia32_sfence();                          // the streamed lines are globally visible before the DMA is asked for
                return done;
                }
            }
            while (done < len) {
                const ssize_t r = pread(fd, dst + done, len - done, base_off + (off_t)(off + done));
                if (r <= 0) break;
                done += (size_t)r;
            }
            return done;
        };
        return run_pipeline(read_at, nch, ch, band, nullptr, df_const, out, max_windows, n_done);
    }

    int process_host(const int16_t* iq, long long nwin, int nch, int ch, const twx_band* band, const double* df,
                     twx_result* out) override {
        if (!band && !df) return fail(TWX_E_ARG, "either band or df must be given");
        const char* src = reinterpret_cast<const char*>(iq);
        const size_t total = (size_t)nwin * (size_t)N * nch * 4;
        auto read_at = [src, total](char* dst, size_t off, size_t len) -> size_t {
            if (off >= total) return 0;
            const size_t n = std::min(len, total - off);
            memcpy(dst, src + off, n);
            return n;
        };
        long long done = 0;
        const int rc = run_pipeline(read_at, nch, ch, band, df, 0.0, out, nwin, &done);
        if (rc == TWX_OK && done != nwin) return fail(TWX_E_HIP, "host pipeline processed fewer windows than requested");
        return rc;
    }

    // processing(d,k) / processing(d,df) on complex-double windows in HOST memory, nwin consecutive windows of N samples,
    // sample n of the capture at re[n*stride], im[n*stride].  No mean removal (the reference's callers do it).
    int process_complex(const double* re, const double* im, long long stride, long long nwin, const twx_band* band,
                        const double* df, twx_result* out) override {
        if (!band && !df) return fail(TWX_E_ARG, "either band or df must be given");
        if (stride < 1 || stride > 2) return fail(TWX_E_ARG, "stride must be 1 (separate arrays) or 2 (interleaved)");
        const bool inter = stride == 2 && im == re + 1;
        if (stride == 2 && !inter) return fail(TWX_E_ARG, "stride 2 means interleaved storage: im must be re + 1");
        if (int rc = sync_all()) return rc;
        use_slot(0);
        Scratch sc(this);
        double *dre = nullptr, *dim = nullptr;
        const size_t per = (size_t)N * (size_t)stride;
        if (int rc = sc.get(&dre, per * (size_t)B)) return rc;
        if (!inter) { if (int rc = sc.get(&dim, per * (size_t)B)) return rc; } else dim = dre + 1;
        for (long long w0 = 0; w0 < nwin; w0 += B) {
            const int nb = (int)std::min<long long>(B, nwin - w0);
            HIPCHK(hipMemcpyAsync(dre, re + (size_t)w0 * per, per * nb * sizeof(double), hipMemcpyHostToDevice, stream));
            if (!inter) HIPCHK(hipMemcpyAsync(dim, im + (size_t)w0 * per, per * nb * sizeof(double), hipMemcpyHostToDevice, stream));
            if (int rc = run_batch_in(IN_C64S, dre, dim, (int)stride, (long long)per, 0, nb, band, df ? df + w0 : nullptr, res_dev, nullptr, 1)) return rc;
            HIPCHK(hipMemcpyAsync(out + w0, res_dev, sizeof(twx_result) * nb, hipMemcpyDeviceToHost, stream));
            HIPCHK(hipStreamSynchronize(stream));
        }
        return TWX_OK;
    }

    // replica spectrum handed over by the caller (natural order) -> [k1][k2] layout in the context's precision, range
    // scale applied, DIF/DIT copy refreshed
    int set_code_spectrum(const double* spec) override {
        if (int rc = sync_all()) return rc;
        std::vector<C> h((size_t)N);
        for (int k1 = 0; k1 < N1; ++k1)
            for (int k2 = 0; k2 < N2; ++k2) {
                const size_t k = (size_t)k1 + (size_t)N1 * k2;
                h[(size_t)k1 * N2 + k2] = mk<T>((T)(spec[2 * k] * scale_pow2), (T)(spec[2 * k + 1] * scale_pow2));
            }
        HIPCHK(hipMemcpy(cspec, h.data(), (size_t)N * sizeof(C), hipMemcpyHostToDevice));
        if (use_rowd) {
            const int Rr = row->R[row->S - 1], R0 = row->S == 3 ? row->R[0] : 1;
            TWX_LAUNCH((k_cspec_perm<T>), dim3(N1), dim3(256), slots[0].stream, cspec, cspec_perm, N1, N2, R0, Rr);
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(slots[0].stream));
        }
        snr_valid = 0;
        return TWX_OK;
    }
    // the same from DEVICE memory (natural order, complex doubles): no host pass over the spectrum
    int set_code_spectrum_dev(const void* spec_dev) override {
        if (int rc = sync_all()) return rc;
        use_slot(0);
        TWX_LAUNCH((k_spec_from_natural<T>), dim3(2048), dim3(256), stream, reinterpret_cast<const cpx<double>*>(spec_dev), cspec, N1, N2, scale_pow2);
        HIPCHK(hipGetLastError());
        if (use_rowd) {
            const int Rr = row->R[row->S - 1], R0 = row->S == 3 ? row->R[0] : 1;
            TWX_LAUNCH((k_cspec_perm<T>), dim3(N1), dim3(256), stream, cspec, cspec_perm, N1, N2, R0, Rr);
            HIPCHK(hipGetLastError());
        }
        HIPCHK(hipStreamSynchronize(stream));
        snr_valid = 0;
        return TWX_OK;
    }
    // forward transform of N complex doubles, device to device, natural order out (asynchronous on the context's stream;
    // in_dev and out_dev may be the same buffer)
    int fft_forward_dev(const void* in_dev, void* out_dev) override {
        if (int rc = sync_all()) return rc;
        use_slot(0);
        C* tmp = A; C* spec = Bz;                                   // the batch buffers of slot 0 hold at least one window each
        ColFwdArgs<T> ca{};
        ca.n = N; ca.n2 = N2; ca.ntiles = ntiles; ca.nwin = 1; ca.tw1 = tw1; ca.ta = ta; ca.tb = tb; ca.tshift = tshift; ca.tc = tcw; ca.out = tmp;
        if (col->fwd(COL_PLAIN, IN_C64, in_dev, 0, &ca, (unsigned)ntiles, stream)) return fail(TWX_E_HIP, "col pass launch failed");
        RowArgs<T> ra{};
        ra.n = N; ra.n1 = N1; ra.nwin = 1; ra.A = tmp; ra.wshift = wshift_of(col->W); ra.stab_f = stab_f; ra.stab_i = stab_i; ra.spec_out = spec;
        if (row->run(ROW_STORE, &ra, (unsigned)N1, stream)) return fail(TWX_E_HIP, "row pass launch failed");
        TWX_LAUNCH((k_spec_to_natural<T>), dim3(2048), dim3(256), stream, (const C*)spec, reinterpret_cast<cpx<double>*>(out_dev), N1, N2);
        HIPCHK(hipGetLastError());
        return TWX_OK;
    }
    int xcorr_map_dev(const void* iq_dev, int nch, int ch, double df, void* out_dev) override {
        if (int rc = sync_all()) return rc;
        use_slot(0);
        C* z = reinterpret_cast<C*>(out_dev);
        // ifft normalisation and the range scale are undone by the last pass as it writes the map (no separate pass)
        return run_batch(reinterpret_cast<const short2*>(iq_dev) + ch, 1, nch, nullptr, &df, res_dev, z, false, 1, 1.0 / scale_pow2 / ((double)N * R));
    }
    int caf_freqs_cdev(const void* d_dev, const double* freqs, long long nf, int flags, twx_result* out) override {
        if (int rc = sync_all()) return rc;
        use_slot(0);
        Scratch sc(this);
        const void* win = nullptr;
        if (int rc = strided_window(sc, d_dev, flags, &win)) return rc;
        argmax_norm1 = (flags & TWX_ACQ_IZAMAX) ? 1 : 0;
        int rc = TWX_OK;
        for (long long f0 = 0; f0 < nf && rc == TWX_OK; f0 += B) {
            const int nb = (int)std::min<long long>(B, nf - f0);
            rc = run_batch_in(IN_C32, win, nullptr, 1, 0, 0, nb, nullptr, freqs + f0, res_dev, nullptr, 1);
            if (rc == TWX_OK && hipMemcpyAsync(out + f0, res_dev, sizeof(twx_result) * nb, hipMemcpyDeviceToHost, stream) != hipSuccess) rc = fail(TWX_E_HIP, "D2H copy failed");
            if (rc == TWX_OK && hipStreamSynchronize(stream) != hipSuccess) rc = fail(TWX_E_HIP, "stream synchronize failed");
        }
        argmax_norm1 = 0;
        return rc;
    }
    // complex-float stream with a decimation stride in flags bits 8..15 -> contiguous window (or d_dev itself)
    int strided_window(Scratch& sc, const void* d_dev, int flags, const void** win) {
        const int dec = std::max(1, (flags >> 8) & 0xff);
        *win = d_dev;
        if (dec == 1) return TWX_OK;
        cpx<float>* tmp = nullptr;
        if (int rc = sc.get(&tmp, (size_t)N)) return rc;
        TWX_LAUNCH(k_stride_copy, dim3(1024), dim3(256), stream, reinterpret_cast<const cpx<float>*>(d_dev), tmp, (long long)N, dec);
        HIPCHK(hipGetLastError());
        *win = tmp;
        return TWX_OK;
    }
    // The whole sweep of rxcomplex.cpp:534-567 in one call: coarse sweep fc0 +- frange in fstep, then step halving with
    // range = step until the step drops under 1 Hz.  Every round's records stay on the device, the bookkeeping between
    // rounds is k_acq_update, and the host synchronises once, at the end.
    int acquire_cdev(const void* d_dev, double fc0, double frange, double fstep, long long ptmod, int flags, twx_acq_result* out) override {
        if (!(fstep > 0) || !(frange >= 0)) return fail(TWX_E_ARG, "acquire: need fstep > 0 and frange >= 0");
        if (int rc = sync_all()) return rc;
        use_slot(0);
        Scratch sc(this);
        const void* win = nullptr;
        if (int rc = strided_window(sc, d_dev, flags, &win)) return rc;
        std::vector<double> trial;
        for (double fcc = fc0 - frange; fcc <= fc0 + frange; fcc += fstep) {        // :536-538, the same repeated addition
            trial.push_back(fcc);
            if (trial.size() > (1u << 20)) return fail(TWX_E_ARG, "acquire: more than 2^20 trial carriers in the coarse sweep");
        }
        int later = 0;
        for (double st = fstep / 2.0; !(st < 1.0); st /= 2.0) ++later;              // rounds after the coarse one (:565-567)
        // every later round evaluates fc-step, fc, fc+step (+1 slot for rounding slack) in ONE launch: a context whose batch
        // is shorter would silently drop the fc+step trial and sweep differently from rxcomplex.cpp:534-567
        if (B < 3) return fail(TWX_E_ARG, "acquire: the context's max_batch must be at least 3 (three trial carriers per refinement round)");
        const int cap = std::min(B, 4);
        twx_result* rec = nullptr; AcqState* st_dev = nullptr;
        if (int rc = sc.get(&rec, trial.size() + (size_t)later * cap + 1)) return rc;
        if (int rc = sc.get(&st_dev, 1)) return rc;
        AcqState h{}; h.fc = fc0; h.pk = 0.0; h.step = fstep; h.pt = 0; h.n_trials = 0; h.cnt = 0;
        HIPCHK(hipMemcpyAsync(st_dev, &h, sizeof h, hipMemcpyHostToDevice, stream));
        argmax_norm1 = (flags & TWX_ACQ_IZAMAX) ? 1 : 0;
        int rc = TWX_OK;
        const long long n1 = (long long)trial.size();
        for (long long f0 = 0; f0 < n1 && rc == TWX_OK; f0 += B) {
            const int nb = (int)std::min<long long>(B, n1 - f0);
            rc = run_batch_in(IN_C32, win, nullptr, 1, 0, 0, nb, nullptr, trial.data() + f0, rec + f0, nullptr, 1);
        }
        twx_result* cur = rec + n1;
        int n_prev = (int)n1; const twx_result* prev = rec;
        for (int r = 0; r <= later && rc == TWX_OK; ++r) {
            TWX_LAUNCH(k_acq_update, dim3(1), dim3(256), stream, prev, n_prev, st_dev, dfv, cap, ptmod);
            if (hipGetLastError() != hipSuccess) { rc = fail(TWX_E_HIP, "k_acq_update launch failed"); break; }
            if (r == later) break;                                                   // the last update only folds the last round in
            rc = run_batch_in(IN_C32, win, nullptr, 1, 0, 0, cap, nullptr, nullptr, cur, nullptr, 1);
            prev = cur; cur += cap; n_prev = -1;
        }
        argmax_norm1 = 0;
        if (rc) return rc;
        HIPCHK(hipMemcpyAsync(&h, st_dev, sizeof h, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        out->fc = h.fc; out->pk = h.pk; out->pt = h.pt; out->n_trials = h.n_trials;
        return TWX_OK;
    }

    int caf_bins(const int16_t* iq, int nch, int ch, long long k_lo, long long k_hi, double* pk, long long* lag) override {
        if (k_hi < k_lo) return fail(TWX_E_ARG, "k_hi < k_lo");
        short2* din = nullptr;
        Scratch sc(this);
        if (int rc = sc.get(&din, (size_t)N * nch)) return rc;
        HIPCHK(hipMemcpy(din, iq, (size_t)N * nch * 4, hipMemcpyHostToDevice));
        return caf_bins_dev(din, nch, ch, k_lo, k_hi, pk, lag);
    }
    // the window already in device memory (outputs to the host: 16 bytes per bin)
    int caf_bins_dev(const void* iq_dev, int nch, int ch, long long k_lo, long long k_hi, double* pk, long long* lag) override {
        if (k_hi < k_lo) return fail(TWX_E_ARG, "k_hi < k_lo");
        if (int rc = sync_all()) return rc;
        use_slot(0);
        C* Ysp = nullptr; double* pk_d = nullptr; long long* lag_d = nullptr;
        const int nbmax = B * R;
        const long long nbins = k_hi - k_lo + 1;
        Scratch sc(this);
        // the surface's work buffers stay with the context (a 2.5-GB bin buffer allocated and freed per call cost a third of
        // the call, profiles/r03_caf_sweep.txt)
        if (!(Ysp = static_cast<C*>(scratch_slot(2, (size_t)N * sizeof(C))))) return TWX_E_NOMEM;
        if (int rc = sc.get(&pk_d, (size_t)nbins)) return rc;       // every bin's record stays on the device until the end:
        if (int rc = sc.get(&lag_d, (size_t)nbins)) return rc;      // one D2H copy and one synchronisation per call
        const short2* in = reinterpret_cast<const short2*>(iq_dev) + ch;
        TWX_LAUNCH((k_sums<0>), dim3((unsigned)sums_chunks(1), 1), dim3(256), stream, in, 0ll, nch, N, sum_parts);
        TWX_LAUNCH((k_sums_final<0>), dim3(1, 1), dim3(256), stream, sum_parts, sums_chunks(1), 0ll, sums, sums);
        HIPCHK(hipGetLastError());
        ColFwdArgs<T> ca{};
        ca.in_win_stride = 0; ca.sums = sums; ca.remove_mean = 1; ca.n = N; ca.n2 = N2; ca.ntiles = ntiles; ca.nwin = 1;
        ca.tw1 = tw1; ca.ta = ta; ca.tb = tb; ca.tshift = tshift; ca.tc = tcw; ca.out = A;
        if (col->fwd(COL_PLAIN, IN_I16, in, nch, &ca, (unsigned)ntiles, stream)) return fail(TWX_E_HIP, "k_col_fwd(plain) launch failed");
        RowArgs<T> ra{};
        ra.n = N; ra.n1 = N1; ra.nwin = 1; ra.A = A; ra.wshift = wshift_of(col->W); ra.stab_f = stab_f; ra.stab_i = stab_i; ra.spec_out = Ysp;
        if (row->run(ROW_STORE, &ra, (unsigned)N1, stream)) return fail(TWX_E_HIP, "k_row(store) launch failed");
        // DIF/DIT form of the per-bin row pass (k_rowd_caf) where the row plan has one: Y in block-thread order, several
        // bins per workgroup, and a bin buffer of its own (up to 64 bins or 2.5 GB per launch instead of the B*R batch windows)
        // bins per launch / per workgroup: tools/caf_rate.py sweeps (profiles/r03_caf_sweep.txt)
        static const int caf_bpl = [] { const char* e = getenv("TWX_CAF_BPL"); return e ? std::max(1, atoi(e)) : 64; }();
        static const long long caf_maxmb = [] { const char* e = getenv("TWX_CAF_MAXMB"); return e ? std::max(64ll, atoll(e)) : 2560ll; }();
        const bool caf_stockham = getenv("TWX_CAF_STOCKHAM") != nullptr;            // tests/experiments: force the Stockham form (read per call)
        const bool dform = cspec_perm && row->rowd && row->S == 3 && !caf_stockham;
        C* Yperm = nullptr; C* Bzc = Bz; ArgPart<T>* partc = part_peak;
        int nbpl = nbmax, bpw = 1;
        // Two bin buffers on two streams: launch g+1's row pass (VALU / LDS bound) runs beside launch g's column pass (read
        // bound), as the pipeline slots of the main chain do (TWX_CAF_SERIAL=1: one stream, profiles/r04_caf_overlap.txt)
        static const bool caf_serial = [] { const char* e = getenv("TWX_CAF_SERIAL"); return e && atoi(e) != 0; }();
        int nstr = 1;
        if (dform) {
            if (!(Yperm = static_cast<C*>(scratch_slot(3, (size_t)N * sizeof(C))))) return TWX_E_NOMEM;
            TWX_LAUNCH((k_cspec_perm<T>), dim3(N1), dim3(256), stream, Ysp, Yperm, N1, N2, row->R[0], row->R[2]);
            HIPCHK(hipGetLastError());
            const long long want = std::min<long long>(std::min<long long>(caf_bpl, nbins), std::max<long long>(1, (caf_maxmb << 20) / (N * (long long)sizeof(C))));
            nstr = (nslots >= 2 && !caf_serial && nbins > want) ? 2 : 1;
            if (want * nstr > nbmax) {
                if (!(Bzc = static_cast<C*>(scratch_slot(4, (size_t)N * want * nstr * sizeof(C))))) return TWX_E_NOMEM;
                if (!(partc = static_cast<ArgPart<T>*>(scratch_slot(5, (size_t)ntiles_inv * want * nstr * sizeof(ArgPart<T>))))) return TWX_E_NOMEM;
            }
            nbpl = (int)want;                        // <= nbmax: the batch buffers of the chain serve as the bin buffer
            static const int bpw_env = [] { const char* e = getenv("TWX_CAF_BPW"); return e ? std::max(1, atoi(e)) : 32; }();
            bpw = std::min(bpw_env, nbpl);
        }
        // non-temporal bin-buffer stores only when the launch's bins cannot stay cached (TWX_CAF_NT=0/1 forces)
        static const int nt_env = [] { const char* e = getenv("TWX_CAF_NT"); return e ? atoi(e) : -1; }();
        const int nt = nt_env >= 0 ? nt_env : ((long long)nbpl * N * (long long)sizeof(C) > (192ll << 20) ? 1 : 0);
        if (nstr > 1) {                                          // the second stream starts after Y / Yperm are complete
            HIPCHK(hipEventRecord(ev_fork, slots[0].stream));
            HIPCHK(hipStreamWaitEvent(slots[1].stream, ev_fork, 0));
        }
        C* const Bzc0 = Bzc; ArgPart<T>* const partc0 = partc;
        long long grp = 0;
        for (long long k0 = k_lo; k0 <= k_hi; k0 += nbpl, ++grp) {
            const int nb = (int)std::min<long long>(nbpl, k_hi - k0 + 1);
            const int sidx = (int)(grp % nstr);
            hipStream_t stream = slots[sidx].stream;              // shadows the context's current stream inside the loop
            Bzc = Bzc0 + (size_t)sidx * nbpl * N; partc = partc0 + (size_t)sidx * nbpl * ntiles_inv;
            CafArgs<T> fa{};
            fa.n = N; fa.n1 = N1; fa.nbins = nb; fa.kappa0 = k0; fa.Y = Ysp; fa.cspec = cspec; fa.stab_i = stab_i;
            fa.ta = ta; fa.tb = tb; fa.tshift = tshift; fa.scale = (T)scale_pow2; fa.Bz = Bzc;
            fa.Yperm = Yperm; fa.cspec_perm = cspec_perm; fa.dtabs = dtabs; fa.vc = vc_d; fa.bpw = bpw; fa.nt = nt;
            static const int caf_rot = [] { const char* e = getenv("TWX_CAF_ROTATE"); return e ? atoi(e) : 1; }();      // 0: plain bin order (A/B, profiles/r04_caf_rotate.txt)
            fa.rotate = caf_rot;
            static const int caf_pad = [] { const char* e = getenv("TWX_CAF_PAD"); return e ? std::max(0, atoi(e)) : 0; }();
            fa.lds_pad = (nstr > 1 && !profile) ? caf_pad : 0;
            const unsigned grid = dform ? (unsigned)(N1 * ((nb + bpw - 1) / bpw)) : (unsigned)(N1 * nb);
            {
                ProfScope ps(this, PC_ROW_CAF, (long long)nb * N);
                if (row->caf(&fa, grid, stream)) return fail(TWX_E_HIP, "k_row_caf launch failed");
            }
            ColInvArgs<T> ia{};
            ia.n = N; ia.n2 = N2; ia.ntiles = ntiles_inv; ia.nphase = 1; ia.nwin = nb; ia.Bz = Bzc; ia.tw1 = tw1; ia.part = partc; ia.zout = nullptr;
            {
                ProfScope ps(this, PC_COL_INV_CAF, (long long)nb * N);
                if (colinv->inv(&ia, (unsigned)(ntiles_inv * nb), stream)) return fail(TWX_E_HIP, "k_col_inv launch failed");
            }
            {
                ProfScope ps(this, PC_CAF_REDUCE, nb);
                TWX_LAUNCH((k_caf_reduce<T>), dim3(nb), dim3(256), stream, partc, ntiles_inv, 1.0 / scale_pow2 / (double)N, pk_d + (k0 - k_lo), lag_d + (k0 - k_lo));
                HIPCHK(hipGetLastError());
            }
        }
        if (nstr > 1) {
            HIPCHK(hipEventRecord(ev_join[1], slots[1].stream));
            HIPCHK(hipStreamWaitEvent(slots[0].stream, ev_join[1], 0));
        }
        HIPCHK(hipMemcpyAsync(pk, pk_d, sizeof(double) * (size_t)nbins, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipMemcpyAsync(lag, lag_d, sizeof(long long) * (size_t)nbins, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        return TWX_OK;
    }
};

}  // namespace twx

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
using namespace twx;
struct twx_ctx { CtxBase* impl; };

namespace twx {
hipStream_t ctx_stream(twx_ctx* ctx) { return ctx->impl->stream; }
int ctx_fail(twx_ctx* ctx, int code, const char* msg) { return ctx->impl->fail(code, msg); }
int ctx_set_device(twx_ctx* ctx) { return hipSetDevice(ctx->impl->dev) == hipSuccess ? TWX_OK : ctx->impl->fail(TWX_E_HIP, "hipSetDevice failed"); }
std::vector<unsigned char>& ctx_scratch_shadow(twx_ctx* ctx, int slot) { return ctx->impl->aux_shadow[slot < 0 || slot >= AUX_SCRATCH_SLOTS ? 0 : slot]; }
void* ctx_scratch(twx_ctx* ctx, int slot, size_t bytes) { return ctx->impl->scratch_slot(slot, bytes); }
int ctx_device(twx_ctx* ctx) { return ctx->impl->dev; }
int ctx_fir_mfma_option(twx_ctx* ctx) { return ctx->impl->fir_mfma; }

// ---- device-wide fence around the matrix-core FIR (twx_internal.h) ----
namespace {
struct DeviceFence {
    std::shared_mutex mu;                                  // shared: any enqueue; exclusive: a matrix-core FIR launch
    std::mutex reg_mu;
    std::vector<std::pair<hipStream_t, hipEvent_t>> streams;   // every stream the library created on this device + its marker event
    hipEvent_t fir_ev = nullptr;                           // recorded behind the last matrix-core FIR (written under the exclusive lock)
    std::atomic<int> fir_seen{0};
};
DeviceFence& fence_of(int dev) { static DeviceFence f[64]; return f[dev & 63]; }
std::atomic<long long> g_mfma_launches{0};
}  // namespace
void fence_register(int dev, hipStream_t s) {
    DeviceFence& f = fence_of(dev);
    std::lock_guard<std::mutex> g(f.reg_mu);
    for (auto& p : f.streams) if (p.first == s) return;
    f.streams.push_back({s, nullptr});                     // the marker event is created by the first exclusive section that needs it
}
static thread_local int t_fence_depth[64];
void fence_unregister(int dev, hipStream_t s) {
    DeviceFence& f = fence_of(dev);
    // no launch in progress while a stream leaves; a thread that destroys a context from inside one of its own enqueue sections gives its
    // share up for the moment (it would wait for itself otherwise)
    const bool mine = t_fence_depth[dev & 63] > 0;
    if (mine) f.mu.unlock_shared();
    struct Back { DeviceFence& f; bool on; ~Back() { if (on) f.mu.lock_shared(); } } back{f, mine};
    std::unique_lock<std::shared_mutex> x(f.mu);
    std::lock_guard<std::mutex> g(f.reg_mu);
    for (size_t i = 0; i < f.streams.size(); ++i)
        if (f.streams[i].first == s) { if (f.streams[i].second) (void)hipEventDestroy(f.streams[i].second); f.streams.erase(f.streams.begin() + (long)i); break; }
}
// entry points call each other (the tracked flow and the receiver run on the correlator's own entries): only the outermost section of a
// thread takes the lock — a reader that re-enters behind a waiting writer would wait for itself (t_fence_depth)
FenceShared::FenceShared(int dev_, hipStream_t s) : dev(dev_) {
    DeviceFence& f = fence_of(dev);
    if (t_fence_depth[dev & 63]++ == 0) f.mu.lock_shared();
    if (f.fir_seen.load(std::memory_order_acquire) && f.fir_ev) (void)hipStreamWaitEvent(s, f.fir_ev, 0);
}
FenceShared::~FenceShared() { if (--t_fence_depth[dev & 63] == 0) fence_of(dev).mu.unlock_shared(); }
// TWX_FIR_MFMA_UNFENCED=1: DIAGNOSTIC ONLY — the launch is not ordered against anything (tools/chain_corunner.py needs the two
// kernels resident together to show that TWX_OPT_SELFCHECK catches what then goes wrong)
static bool fence_off() { const char* e = getenv("TWX_FIR_MFMA_UNFENCED"); return e && atoi(e) != 0; }
FenceExclusive::FenceExclusive(int dev_, hipStream_t s_) : dev(dev_), s(s_) {
    DeviceFence& f = fence_of(dev);
    // (a thread inside a shared section of this device asks for the matrix-core FIR: its own section ends here, and resumes afterwards)
    if (t_fence_depth[dev & 63] > 0) f.mu.unlock_shared();
    f.mu.lock();
    std::lock_guard<std::mutex> g(f.reg_mu);
    if (fence_off()) return;
    for (auto& p : f.streams) {
        if (p.first == s) continue;
        if (!p.second && hipEventCreateWithFlags(&p.second, hipEventDisableTiming) != hipSuccess) { p.second = nullptr; (void)hipGetLastError(); continue; }
        if (hipEventRecord(p.second, p.first) == hipSuccess && hipStreamWaitEvent(s, p.second, 0) == hipSuccess) ++waited;
        else (void)hipGetLastError();
    }
}
FenceExclusive::~FenceExclusive() {
    DeviceFence& f = fence_of(dev);
    if (!fence_off() && !f.fir_ev && hipEventCreateWithFlags(&f.fir_ev, hipEventDisableTiming) != hipSuccess) { f.fir_ev = nullptr; (void)hipGetLastError(); }
    if (!fence_off() && f.fir_ev) { (void)hipEventRecord(f.fir_ev, s); f.fir_seen.store(1, std::memory_order_release); }
    g_mfma_launches.fetch_add(1);
    f.mu.unlock();
    if (t_fence_depth[dev & 63] > 0) f.mu.lock_shared();
}
long long fence_mfma_launches() { return g_mfma_launches.load(); }
}  // namespace twx

// No exception may cross the C boundary (std::async, std::vector and std::string can throw).
template <class F> static int guarded(CtxBase* c, F f) noexcept {
    // the launches of this library are checked with hipGetLastError(): an error another library left behind on this thread
    // (RCCL and PyTorch probe pointers and peers and do not clear what those probes set) must not be taken for ours
    (void)hipGetLastError();
    try {
        if (c && c->stream) { twx::FenceShared fence(c->dev, c->stream); return f(); }      // never beside a matrix-core FIR (twx_internal.h)
        return f();
    }
    catch (const std::bad_alloc&) { return c ? c->fail(TWX_E_NOMEM, "out of host memory") : TWX_E_NOMEM; }
    catch (const std::exception& e) { return c ? c->fail(TWX_E_STATE, std::string("internal error: ") + e.what()) : TWX_E_STATE; }
    catch (...) { return c ? c->fail(TWX_E_STATE, "internal error") : TWX_E_STATE; }
}

extern "C" {

int twx_abi_version(void) { return TWX_ABI_VERSION; }

const char* twx_strerror(int status) {
    switch (status) {
        case TWX_OK: return "ok";
        case TWX_E_ARG: return "invalid argument";
        case TWX_E_SIZE: return "unsupported window length";
        case TWX_E_HIP: return "HIP runtime error";
        case TWX_E_NOMEM: return "out of device memory";
        case TWX_E_STATE: return "invalid state";
        default: return "unknown status";
    }
}
const char* twx_last_error(const twx_ctx* ctx) { return ctx ? ctx->impl->err.c_str() : g_create_err.c_str(); }

static int create_impl(const twx_config* cfg, twx_ctx** out) {
    if (!cfg || !out) { g_create_err = "null argument"; return TWX_E_ARG; }
    *out = nullptr;
    if (!(cfg->fs > 0) || cfg->sps < 1 || cfg->nint < 0 || cfg->nint > 2 || cfg->n_chips < 1) { g_create_err = "bad fs/sps/nint/n_chips"; return TWX_E_ARG; }
    if (cfg->nphase < 0 || cfg->nphase > TWX_MAX_PHASE) { g_create_err = "nphase must be 0 (= 2*nint+1) or 1..5"; return TWX_E_ARG; }
    if (cfg->precision != TWX_F32 && cfg->precision != TWX_F64) { g_create_err = "bad precision"; return TWX_E_ARG; }
    if (cfg->convention != TWX_CONV_GODUAL && cfg->convention != TWX_CONV_CLAUDIO) { g_create_err = "bad convention"; return TWX_E_ARG; }
    if (cfg->var_ddof < 0 || cfg->var_ddof > 1) { g_create_err = "var_ddof must be 0 or 1"; return TWX_E_ARG; }
    const long long N = cfg->n_chips * cfg->sps;
    if (N % 2) { g_create_err = "window length must be even"; return TWX_E_SIZE; }
    if ((long long)N * (cfg->nphase > 0 ? cfg->nphase : 2 * cfg->nint + 1) >= 0xffffffffll) { g_create_err = "window too long for 32-bit lag indices"; return TWX_E_SIZE; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { g_create_err = "no HIP device available (the HIP path has no CPU fallback)"; return TWX_E_HIP; }
    if (cfg->device >= 0) { if (hipSetDevice(cfg->device) != hipSuccess) { g_create_err = "hipSetDevice failed"; return TWX_E_HIP; } }
    (void)hipGetLastError();      // a stale error of another library on this thread is not ours (see guarded())
    const ColOps* col; const RowOps* row;
    const int f64 = cfg->precision == TWX_F64;
    if (!choose_split(N, f64, &col, &row)) {
        // plan plug-ins built earlier for other lengths (python -m amaranth_twstft_amd.plans N) live beside the library
        if (scan_plan_dir() == 0 || !choose_split(N, f64, &col, &row)) {
            char b[256]; snprintf(b, sizeof b, "no plan pair N1*N2 = %lld is built in or found in %s: build one with `python -m amaranth_twstft_amd.plans %lld` (DESIGN.md §plans)", N, default_plan_dir().c_str(), N);
            g_create_err = b; return TWX_E_SIZE;
        }
    }
    if ((col->W & (col->W - 1)) != 0) { g_create_err = "column tile width must be a power of two (tile-blocked A layout)"; return TWX_E_SIZE; }
    CtxBase* c = f64 ? static_cast<CtxBase*>(new Ctx<double>()) : static_cast<CtxBase*>(new Ctx<float>());
    c->cfg = *cfg; c->N = N; c->N1 = col->L; c->N2 = row->L; c->col = col; c->row = row; c->colinv = choose_col_inv(col, row->L, f64);
    (void)hipGetDevice(&c->dev);
    int rc = guarded(c, [&]() { return c->init(); });
    if (rc) { g_create_err = c->err; delete c; return rc; }
    c->cfg.chips = nullptr;
    // TWX_SELFCHECK=<value of TWX_OPT_SELFCHECK>: the option's default for every context of the process whose row pass has the form
    // (contexts of other shapes stay as they are; twx_set_option overrides either way)
    if (const char* e = getenv("TWX_SELFCHECK")) { if (atoll(e) > 0) { (void)c->set_selfcheck(atoll(e)); c->err.clear(); } }
    twx_ctx* h = new (std::nothrow) twx_ctx{c};
    if (!h) { delete c; g_create_err = "out of host memory"; return TWX_E_NOMEM; }
    *out = h;
    return TWX_OK;
}
// strings, the plan-directory scan and the context allocation can throw: nothing may cross the C boundary
int twx_create(const twx_config* cfg, twx_ctx** out) {
    try { return create_impl(cfg, out); }
    catch (const std::bad_alloc&) { if (out) *out = nullptr; return TWX_E_NOMEM; }
    catch (...) { if (out) *out = nullptr; return TWX_E_STATE; }
}

void twx_destroy(twx_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->impl->dev);
    (void)ctx->impl->sync_all();
    delete ctx->impl;
    delete ctx;
}

const char* twx_plan_source_hash(void) { return TWX_SRC_HASH; }
int twx_load_plan(const char* path) {
    if (!path) return TWX_E_ARG;
    try { return load_plan_file(path) ? TWX_E_ARG : TWX_OK; } catch (...) { return TWX_E_STATE; }
}
int twx_plan_available(int64_t n, int32_t precision) {
    const ColOps* c; const RowOps* r;
    try {
        if (choose_split(n, precision == TWX_F64, &c, &r)) return 1;
        scan_plan_dir();
        return choose_split(n, precision == TWX_F64, &c, &r) ? 1 : 0;
    } catch (...) { return 0; }
}
int twx_plan_lengths(int32_t kind, int32_t precision, int32_t* lengths, int32_t* widths, int32_t max_entries) {
    int n = 0;
    std::lock_guard<std::recursive_mutex> g(reg_mu());
    if (kind == 0) { for (auto& e : col_reg()) if (e.o.f64 == (precision == TWX_F64)) { if (n < max_entries) { if (lengths) lengths[n] = e.o.L; if (widths) widths[n] = e.o.W; } ++n; } }
    else { for (auto& e : row_reg()) if (e.o.f64 == (precision == TWX_F64)) { if (n < max_entries) { if (lengths) lengths[n] = e.o.L; if (widths) widths[n] = 0; } ++n; } }
    return n;
}

int twx_get_info(const twx_ctx* ctx, twx_info* info) {
    if (!ctx || !info) return TWX_E_ARG;
    const CtxBase* c = ctx->impl;
    info->n = c->N; info->n1 = c->N1; info->n2 = c->N2; info->nphase = c->R; info->batch = c->B;
    info->precision = c->cfg.precision; info->col_w = c->col->W; info->device_bytes = c->dev_bytes;
    return TWX_OK;
}

int twx_process_windows_dev(twx_ctx* ctx, const void* iq_dev, int64_t n_windows, int32_t n_channels, int32_t channel,
                            const twx_band* band, const double* df, twx_result* out_dev) {
    if (!ctx) return TWX_E_ARG;
    CtxBase* c = ctx->impl;
    if (!iq_dev || !out_dev || n_windows < 0 || n_channels < 1 || channel < TWX_ALL_CHANNELS || channel >= n_channels || (channel < 0 && n_channels > TWX_MAX_CHANNELS)) return c->fail(TWX_E_ARG, "bad argument");
    (void)hipSetDevice(c->dev);
    return guarded(c, [&]() { return c->process(iq_dev, n_windows, n_channels, channel, band, df, out_dev); });
}

int twx_synchronize(twx_ctx* ctx) {
    if (!ctx) return TWX_E_ARG;
    return ctx->impl->sync_all();
}
int twx_set_option(twx_ctx* ctx, int32_t option, int64_t value) {
    if (!ctx) return TWX_E_ARG;
    if (option == TWX_OPT_REMOVE_MEAN) { (void)ctx->impl->sync_all(); ctx->impl->remove_mean = value ? 1 : 0; return TWX_OK; }
    if (option == TWX_OPT_DEBUG_ONLY) { if (value > PC_PEAK) return ctx->impl->fail(TWX_E_ARG, "no such kernel class"); ctx->impl->dbg_only = value < 0 ? -1 : (int)value; return TWX_OK; }
    if (option == TWX_OPT_SELFCHECK) return ctx->impl->set_selfcheck(value);
    if (option == TWX_OPT_BRUIT_LEN) return ctx->impl->set_extra(0, value);
    if (option == TWX_OPT_NOISE_SQUARE_LEN) return ctx->impl->set_extra(1, value);
    if (option == TWX_OPT_DEBUG_FAULT) { ctx->impl->dbg_fault = (int)value; return TWX_OK; }
    if (option == TWX_OPT_FIR_MFMA) { ctx->impl->fir_mfma = value < 0 ? -1 : (value ? 1 : 0); return TWX_OK; }
    if (option == TWX_OPT_DEBUG_REPEAT) { ctx->impl->dbg_repeat = (int)std::max<long long>(1, std::min<long long>(value, 1000000)); return TWX_OK; }
    return ctx->impl->fail(TWX_E_ARG, "unknown option");
}
void* twx_stream(twx_ctx* ctx) { return ctx ? (void*)ctx->impl->stream : nullptr; }
int twx_fetch_extra(twx_ctx* ctx, twx_extra* out_host, int64_t n_records) {
    if (!ctx) return TWX_E_ARG;
    CtxBase* c = ctx->impl;
    if (!out_host || n_records < 0) return c->fail(TWX_E_ARG, "bad argument");
    if (n_records > c->extra_n) return c->fail(TWX_E_STATE, "the last call left fewer records (are TWX_OPT_BRUIT_LEN / TWX_OPT_NOISE_SQUARE_LEN on?)");
    (void)hipSetDevice(c->dev);
    if (int rc = c->sync_all()) return rc;
    if (n_records && hipMemcpy(out_host, c->extra_dev, sizeof(twx_extra) * (size_t)n_records, hipMemcpyDeviceToHost) != hipSuccess) return c->fail(TWX_E_HIP, "D2H failed");
    return TWX_OK;
}
int twx_set_resample(twx_ctx* ctx, double vitesse, double t0, int64_t dt) {
    if (!ctx) return TWX_E_ARG;
    CtxBase* c = ctx->impl;
    return guarded(c, [&]() { return c->set_resample(vitesse, t0, dt); });
}
int twx_get_resample(twx_ctx* ctx, double* vitesse, double* t0, int64_t* dt) {
    if (!ctx) return TWX_E_ARG;
    if (vitesse) *vitesse = ctx->impl->rs_v;
    if (t0) *t0 = ctx->impl->rs_t0;
    if (dt) *dt = ctx->impl->rs_dt;
    return TWX_OK;
}
int twx_selfcheck_stats(twx_ctx* ctx, double* max_rel_dev, int64_t* rows_flagged, int32_t reset) {
    if (!ctx) return TWX_E_ARG;
    CtxBase* c = ctx->impl;
    if (!c->chk_stat) return c->fail(TWX_E_STATE, "TWX_OPT_SELFCHECK was never switched on in this context");
    (void)hipSetDevice(c->dev);
    if (int rc = c->sync_all()) return rc;
    unsigned h[2] = {0, 0};
    if (hipMemcpy(h, c->chk_stat, 8, hipMemcpyDeviceToHost) != hipSuccess) return c->fail(TWX_E_HIP, "D2H failed");
    float f; memcpy(&f, &h[0], 4);
    if (max_rel_dev) *max_rel_dev = (double)f;
    if (rows_flagged) *rows_flagged = (int64_t)h[1];
    if (reset && hipMemset(c->chk_stat, 0, 8) != hipSuccess) return c->fail(TWX_E_HIP, "memset failed");
    return TWX_OK;
}

int twx_process_windows(twx_ctx* ctx, const int16_t* iq, int64_t n_windows, int32_t n_channels, int32_t channel,
                        const twx_band* band, const double* df, twx_result* out) {
    if (!ctx) return TWX_E_ARG;
    CtxBase* c = ctx->impl;
    if (!iq || !out || n_windows < 0 || n_channels < 1 || channel < TWX_ALL_CHANNELS || channel >= n_channels || (channel < 0 && n_channels > TWX_MAX_CHANNELS)) return c->fail(TWX_E_ARG, "bad argument");
    if (n_windows == 0) return TWX_OK;
    (void)hipSetDevice(c->dev);
    // pinned double-buffered staging: the host-side copy, the H2D transfer and the kernels of consecutive
    // chunks overlap (same pipeline as twx_process_file)
    return guarded(c, [&]() { return c->process_host(iq, n_windows, n_channels, channel, band, df, out); });
}

int twx_process_complex(twx_ctx* ctx, const double* d_re, const double* d_im, int64_t stride, int64_t n_windows,
                        const twx_band* band, const double* df, twx_result* out) {
    if (!ctx) return TWX_E_ARG;
    CtxBase* c = ctx->impl;
    if (!d_re || !d_im || !out || n_windows < 0) return c->fail(TWX_E_ARG, "bad argument");
    if (n_windows == 0) return TWX_OK;
    (void)hipSetDevice(c->dev);
    return guarded(c, [&]() { return c->process_complex(d_re, d_im, stride, n_windows, band, df, out); });
}

int twx_set_code_spectrum(twx_ctx* ctx, const double* spec) {
    if (!ctx || !spec) return TWX_E_ARG;
    (void)hipSetDevice(ctx->impl->dev);
    return guarded(ctx->impl, [&]() { return ctx->impl->set_code_spectrum(spec); });
}
int twx_set_code_spectrum_dev(twx_ctx* ctx, const void* spec_dev) {
    if (!ctx || !spec_dev) return TWX_E_ARG;
    (void)hipSetDevice(ctx->impl->dev);
    return guarded(ctx->impl, [&]() { return ctx->impl->set_code_spectrum_dev(spec_dev); });
}
int twx_fft_forward_dev(twx_ctx* ctx, const void* in_dev, void* out_dev) {
    if (!ctx || !in_dev || !out_dev) return TWX_E_ARG;
    (void)hipSetDevice(ctx->impl->dev);
    return guarded(ctx->impl, [&]() { return ctx->impl->fft_forward_dev(in_dev, out_dev); });
}
int twx_xcorr_map_dev(twx_ctx* ctx, const void* iq_dev, int32_t n_channels, int32_t channel, double df, void* out_dev) {
    if (!ctx || !iq_dev || !out_dev || n_channels < 1 || channel < 0 || channel >= n_channels) return TWX_E_ARG;
    (void)hipSetDevice(ctx->impl->dev);
    return guarded(ctx->impl, [&]() { return ctx->impl->xcorr_map_dev(iq_dev, n_channels, channel, df, out_dev); });
}
int twx_caf_freqs_cdev(twx_ctx* ctx, const void* d_dev, const double* freqs, int64_t n_freqs, int32_t flags, twx_result* out) {
    if (!ctx || !d_dev || !freqs || !out || n_freqs < 0) return TWX_E_ARG;
    (void)hipSetDevice(ctx->impl->dev);
    return guarded(ctx->impl, [&]() { return ctx->impl->caf_freqs_cdev(d_dev, freqs, n_freqs, flags, out); });
}

int twx_acquire_cdev(twx_ctx* ctx, const void* d_dev, double fc_init, double frange, double fstep, int64_t pt_modulus, int32_t flags,
                     twx_acq_result* out) {
    if (!ctx || !d_dev || !out || pt_modulus < 0) return TWX_E_ARG;
    (void)hipSetDevice(ctx->impl->dev);
    return guarded(ctx->impl, [&]() { return ctx->impl->acquire_cdev(d_dev, fc_init, frange, fstep, pt_modulus, flags, out); });
}

int twx_fft_forward(twx_ctx* ctx, const double* in, double* out) {
    if (!ctx || !in || !out) return TWX_E_ARG;
    (void)hipSetDevice(ctx->impl->dev);
    return guarded(ctx->impl, [&]() { return ctx->impl->fft_forward(in, out); });
}
int twx_get_code_spectrum(twx_ctx* ctx, double* out) {
    if (!ctx || !out) return TWX_E_ARG;
    (void)hipSetDevice(ctx->impl->dev);
    return guarded(ctx->impl, [&]() { return ctx->impl->code_spectrum(out); });
}
int twx_xcorr_map(twx_ctx* ctx, const int16_t* iq, int32_t n_channels, int32_t channel, double df, double* out) {
    if (!ctx || !iq || !out || n_channels < 1 || channel < 0 || channel >= n_channels) return TWX_E_ARG;
    (void)hipSetDevice(ctx->impl->dev);
    return guarded(ctx->impl, [&]() { return ctx->impl->xcorr_map(iq, n_channels, channel, df, out); });
}

int twx_debug_stamps(twx_ctx* ctx, unsigned long long* out, long long count) {
    if (!ctx || !out) return TWX_E_ARG;
    Ctx<float>* c = dynamic_cast<Ctx<float>*>(ctx->impl);
    if (!c || !c->stamps_dev) return TWX_E_STATE;
    (void)hipStreamSynchronize(c->stream);
    return hipMemcpy(out, c->stamps_dev, (size_t)count * 8, hipMemcpyDeviceToHost) == hipSuccess ? TWX_OK : TWX_E_HIP;
}

int twx_process_file(twx_ctx* ctx, const char* path, int32_t n_channels, int32_t channel, int64_t skip_samples, const twx_band* band,
                     double df_const, twx_result* out, int64_t max_windows, int64_t* n_done) {
    if (!ctx) return TWX_E_ARG;
    CtxBase* c = ctx->impl;
    if (!path || !out || !n_done || max_windows < 0 || n_channels < 1 || channel < TWX_ALL_CHANNELS || channel >= n_channels || (channel < 0 && n_channels > TWX_MAX_CHANNELS) || skip_samples < 0) return c->fail(TWX_E_ARG, "bad argument");
    (void)hipSetDevice(c->dev);
    long long nd = 0;
    int rc = guarded(c, [&]() { return c->process_file(path, n_channels, channel, skip_samples, band, df_const, out, max_windows, &nd); });
    *n_done = nd;
    return rc;
}

int twx_caf_bins(twx_ctx* ctx, const int16_t* iq, int32_t n_channels, int32_t channel, int64_t k_lo, int64_t k_hi, double* pk, int64_t* lag) {
    if (!ctx || !iq || !pk || !lag || n_channels < 1 || channel < 0 || channel >= n_channels) return TWX_E_ARG;
    (void)hipSetDevice(ctx->impl->dev);
    return guarded(ctx->impl, [&]() { return ctx->impl->caf_bins(iq, n_channels, channel, k_lo, k_hi, pk, (long long*)lag); });
}
int twx_caf_bins_dev(twx_ctx* ctx, const void* iq_dev, int32_t n_channels, int32_t channel, int64_t k_lo, int64_t k_hi, double* pk, int64_t* lag) {
    if (!ctx || !iq_dev || !pk || !lag || n_channels < 1 || channel < 0 || channel >= n_channels) return TWX_E_ARG;
    (void)hipSetDevice(ctx->impl->dev);
    return guarded(ctx->impl, [&]() { return ctx->impl->caf_bins_dev(iq_dev, n_channels, channel, k_lo, k_hi, pk, (long long*)lag); });
}
int twx_caf_freqs(twx_ctx* ctx, const int16_t* iq, int32_t n_channels, int32_t channel, const double* freqs, int64_t n_freqs, twx_result* out) {
    if (!ctx || !iq || !freqs || !out || n_freqs < 0 || n_channels < 1 || channel < 0 || channel >= n_channels) return TWX_E_ARG;
    (void)hipSetDevice(ctx->impl->dev);
    return guarded(ctx->impl, [&]() { return ctx->impl->caf_freqs(iq, n_channels, channel, freqs, n_freqs, out); });
}

int twx_sqspec_bins_dev(twx_ctx* ctx, const void* iq_dev, int64_t n_samples, int32_t n_channels, int32_t channel,
                        const int64_t* bins, int32_t n_bins, double* out_re_im) {
    if (!ctx || !iq_dev || !bins || !out_re_im || n_channels < 1 || channel < 0 || channel >= n_channels) return TWX_E_ARG;
    (void)hipSetDevice(ctx->impl->dev);
    return guarded(ctx->impl, [&]() { return ctx->impl->sqspec_bins(iq_dev, n_samples, n_channels, channel, (const long long*)bins, n_bins, out_re_im); });
}
int twx_sqspec_band_dev(twx_ctx* ctx, const void* iq_dev, int64_t n_samples, int32_t n_channels, int32_t channel,
                        int64_t k_lo, int64_t n_bins, double* out_mag) {
    if (!ctx || !iq_dev || !out_mag || n_channels < 1 || channel < 0 || channel >= n_channels) return TWX_E_ARG;
    (void)hipSetDevice(ctx->impl->dev);
    return guarded(ctx->impl, [&]() { return ctx->impl->sqspec_band(iq_dev, n_samples, n_channels, channel, k_lo, n_bins, out_mag); });
}

int twx_profile_reset(twx_ctx* ctx) {
    if (!ctx) return TWX_E_ARG;
    CtxBase* c = ctx->impl;
    (void)c->sync_all();
    c->prof_collect();
    for (int i = 0; i < PC_COUNT; ++i) { c->prof_ms[i] = 0; c->prof_n[i] = 0; c->prof_units[i] = 0; }
    return TWX_OK;
}
int twx_profile_get(twx_ctx* ctx, twx_prof_entry* entries, int32_t max_entries, int32_t* n_entries) {
    if (!ctx || !entries || !n_entries) return TWX_E_ARG;
    CtxBase* c = ctx->impl;
    (void)c->sync_all();
    c->prof_collect();
    int n = 0;
    for (int i = 0; i < PC_COUNT && n < max_entries; ++i) {
        if (!c->prof_n[i]) continue;
        memset(&entries[n], 0, sizeof(twx_prof_entry));
        snprintf(entries[n].name, sizeof entries[n].name, "%s", kProfNames[i]);
        entries[n].ms_total = c->prof_ms[i]; entries[n].launches = c->prof_n[i]; entries[n].units = c->prof_units[i];
        ++n;
    }
    *n_entries = n;
    return TWX_OK;
}

int twx_lfsr_chips(int32_t bitlen, int32_t taps, int64_t n, uint8_t* out_host) {
    if (!out_host || n < 1) return TWX_E_ARG;
    Ctx<float> tmp;   // borrows the allocator / error plumbing only
    if (hipStreamCreateWithFlags(&tmp.stream, hipStreamNonBlocking) != hipSuccess) { g_create_err = "no HIP device"; return TWX_E_HIP; }
    unsigned char* d = nullptr;
    int rc = tmp.dalloc(&d, (size_t)n);
    if (!rc) rc = tmp.lfsr_to_device(bitlen, (unsigned)taps, n, d);
    if (!rc && hipMemcpy(out_host, d, (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) rc = TWX_E_HIP;
    if (rc) g_create_err = tmp.err;
    return rc;
}

int twx_synth_capture_dev(void* out_dev, int64_t n, int64_t n0, const uint8_t* chips_dev, int64_t n_chips, int32_t sps,
                          int32_t n_channels, const int64_t* params_host, void* stream) {
    if (!out_dev || !chips_dev || !params_host || n_channels < 1 || n_channels > 4 || n < 0) return TWX_E_ARG;
    SynthArgs a{};
    for (int c = 0; c < n_channels; ++c) {
        const int64_t* p = params_host + 8 * c;
        a.ch[c] = SynthChan{p[0], p[1], p[2], p[3], p[4], p[5], p[6], 0};
    }
    const long long total = n * n_channels;
    const unsigned blocks = (unsigned)std::min<long long>(8192, (total + 255) / 256);
    if (!blocks) return TWX_OK;
    hipLaunchKernelGGL(k_synth, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (short2*)out_dev, (long long)n, (long long)n0,
                       chips_dev, (long long)n_chips, sps, n_channels, a);
    return hipGetLastError() == hipSuccess ? TWX_OK : TWX_E_HIP;
}

void* twx_ctx_alloc(twx_ctx* ctx, size_t bytes) {
    if (!ctx) return nullptr;
    (void)hipSetDevice(ctx->impl->dev);
    char* p = nullptr;
    try { return ctx->impl->dalloc(&p, bytes) == TWX_OK ? p : nullptr; } catch (...) { return nullptr; }
}
void twx_ctx_free(twx_ctx* ctx, void* p) {
    if (!ctx || !p) return;
    (void)hipSetDevice(ctx->impl->dev);
    (void)ctx->impl->sync_all();
    ctx->impl->dfree(p);
}
void* twx_dev_alloc(size_t bytes) { void* p = nullptr; return hipMalloc(&p, bytes) == hipSuccess ? p : nullptr; }
void twx_dev_free(void* p) { (void)hipFree(p); }
int twx_memcpy_h2d(void* d, const void* s, size_t b) { return hipMemcpy(d, s, b, hipMemcpyHostToDevice) == hipSuccess ? TWX_OK : TWX_E_HIP; }
int twx_memcpy_d2h(void* d, const void* s, size_t b) { return hipMemcpy(d, s, b, hipMemcpyDeviceToHost) == hipSuccess ? TWX_OK : TWX_E_HIP; }

}  // extern "C"
