// twx_tracked.hip — the tracked multi-code ranging flow behind the C ABI (twx_tracked_* in include/twstft_hip.h):
// the device Backend of twx_tracked_core.h.  A twx_tracked owns a correlator context (claudio convention, Octave
// variances), one device sample buffer [dold | chunk], pinned staging for the capture chunks (the next chunk is read
// while the current one is measured) and the record buffers; every sample operation is a call into the library's own
// entry points (twx_process_windows_dev, twx_sqspec_bins_dev, twx_sqspec_band_dev, twx_xcorr_map_dev) plus one small
// kernel for search_df's per-candidate statistic.
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <future>
#include <new>
#include <string>
#include <vector>
#include "twx_internal.h"
#include "twx_tracked_core.h"

namespace {

// search_df's test of one candidate carrier (claudio_aligned_code_ranging_separate.m:38-43) on the interpolated map the
// context wrote (nphase*N values normalised like ifft; every r-th one times r is the N-point ifft of :38):
//   [prnsig,b]=max(prnmap); prnmap(b-5:b+5)=0; snr=prnsig^2/var(prnmap)          (var: N-1 normalisation)
// One workgroup; fp64 accumulation; first index wins ties.  out = {snr, prnsig, b}
template <typename T>
__global__ __launch_bounds__(1024) void k_map_snr(const T* __restrict__ z, long long n, int r, double* __restrict__ out) {
    __shared__ double sv[1024];
    __shared__ long long si[1024];
    const int t = threadIdx.x;
    auto mag = [&](long long i) {
        const double x = (double)z[2 * i * r], y = (double)z[2 * i * r + 1];
        return hypot(x, y) * (double)r;
    };
    double bv = -1.0; long long bi = 0;
    for (long long i = t; i < n; i += 1024) { const double m = mag(i); if (m > bv) { bv = m; bi = i; } }
    sv[t] = bv; si[t] = bi;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (t < s) {
            const double ov = sv[t + s]; const long long oi = si[t + s];
            if (ov > sv[t] || (ov == sv[t] && oi < si[t])) { sv[t] = ov; si[t] = oi; }
        }
        __syncthreads();
    }
    const double prnsig = sv[0]; const long long b = si[0];
    __syncthreads();
    const long long z_lo = b - 5 < 0 ? 0 : b - 5, z_hi = b + 5;              // zeroed span, clipped like a slice
    double s1 = 0;
    for (long long i = t; i < n; i += 1024) if (i < z_lo || i > z_hi) s1 += mag(i);
    sv[t] = s1;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) { if (t < s) sv[t] += sv[t + s]; __syncthreads(); }
    const double mean = sv[0] / (double)n;
    __syncthreads();
    double s2 = 0;
    for (long long i = t; i < n; i += 1024) { const double d = ((i < z_lo || i > z_hi) ? mag(i) : 0.0) - mean; s2 += d * d; }
    sv[t] = s2;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) { if (t < s) sv[t] += sv[t + s]; __syncthreads(); }
    if (t == 0) { out[0] = prnsig * prnsig / (sv[0] / (double)(n - 1)); out[1] = prnsig; out[2] = (double)b; }
}

thread_local std::string g_trk_create_err;

}  // namespace

struct twx_tracked : twx_trk::Backend {
    twx_tracked_config cfg{};
    twx_trk::Params P;
    twx_ctx* ctx = nullptr;
    std::string err;
    int dev = 0;
    hipStream_t st = nullptr;                 // = twx_stream(ctx)
    // device: sample buffer, tail scratch, records, search_df map + statistic
    short2* buf = nullptr; size_t buf_samples = 0;
    short2* tail = nullptr;
    twx_result* rec_dev = nullptr; size_t rec_cap = 0;
    void* map_dev = nullptr; double* stat_dev = nullptr;
    // pinned staging of capture chunks: slot `cur` holds the chunk being uploaded, the other one the read-ahead
    void* pin[2] = {nullptr, nullptr};
    // the read-ahead goes all the way to the device: the helper thread that fills pin[s] also copies it to stage_dev[s] on a
    // stream of its own, so the PCIe transfer of chunk c+1 runs beside the measurements of chunk c; load_chunk then places it
    // behind the carried tail with a device-to-device copy (where the tail ends is only known once chunk c is through)
    short2* stage_dev[2] = {nullptr, nullptr};
    hipStream_t cst = nullptr; hipEvent_t h2d_ev[2] = {nullptr, nullptr};
    // wall time the control flow spent inside each backend call of the last run (twx_tracked_timing)
    double t_stage[TWX_TRK_NSTAGES] = {0}; long long n_stage[TWX_TRK_NSTAGES] = {0};
    struct Tick {
        twx_tracked* t; int k; std::chrono::steady_clock::time_point t0;
        Tick(twx_tracked* t_, int k_) : t(t_), k(k_), t0(std::chrono::steady_clock::now()) {}
        ~Tick() { t->t_stage[k] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); ++t->n_stage[k]; }
    };
    std::vector<twx_result> rec_host;
    // capture source of the current run
    int fd = -1; const int16_t* host_src = nullptr; long long src_i16 = 0;
    std::future<long long> ahead; long long ahead_pos = -1; int ahead_slot = 0; int cur = 0;
    bool search_mode = false;
    twx_trk::Output out;

    int fail(int code, const std::string& m) { err = m; return code; }
    int lib(int rc) { if (rc) err = twx_last_error(ctx); return rc; }

    ~twx_tracked() override {
        drop_ahead();
        if (ctx) { (void)hipSetDevice(dev); (void)twx_synchronize(ctx); }
        for (void* p : {(void*)buf, (void*)tail, (void*)rec_dev, map_dev, (void*)stat_dev, (void*)stage_dev[0], (void*)stage_dev[1]}) if (p) (void)hipFree(p);
        for (void* p : pin) if (p) (void)hipHostFree(p);
        for (auto e : h2d_ev) if (e) (void)hipEventDestroy(e);
        if (cst) (void)hipStreamDestroy(cst);
        if (ctx) twx_destroy(ctx);
    }
    void drop_ahead() { if (ahead.valid()) { try { (void)ahead.get(); } catch (...) {} } ahead_pos = -1; }

    int init() {
        twx_config c;
        memset(&c, 0, sizeof c);
        c.fs = cfg.fs; c.sps = cfg.sps; c.nint = cfg.nint; c.chips = cfg.chips; c.n_chips = cfg.n_chips;
        c.lfsr_bitlen = cfg.lfsr_bitlen; c.lfsr_taps = cfg.lfsr_taps;
        c.convention = TWX_CONV_CLAUDIO;        // fcode.*conj(ffty) (:59)
        c.var_ddof = 1;                         // Octave var
        c.snr_rot = -1;                         // codetmp(indice-1:end) (:91)
        c.precision = cfg.precision; c.device = cfg.device; c.max_batch = cfg.max_batch;
        if (c.max_batch <= 0) {
            // The flow measures the codes of ONE chunk and then looks at the records (does the window have to move?): nothing of the
            // next chunk is in flight meanwhile, so the chunk should be one launch sequence, not four of 16 windows — batch = the codes
            // of a chunk (the scripts' 2-s chunks of 40-ms codes: 50), at most 64 (A + Bz: 64 x 200 000 x 32 B = 0.4 GB per slot)
            static const int forced = [] { const char* e = getenv("TWX_TRK_BATCH"); return e ? atoi(e) : 0; }();
            const long long n_code = (long long)cfg.n_chips * std::max(cfg.sps, 1);
            const long long per_chunk = n_code > 0 ? cfg.chunk_samples / n_code : 0;
            c.max_batch = forced > 0 ? forced : (int)std::min<long long>(64, std::max<long long>(16, per_chunk));
        }
        if (int rc = twx_create(&c, &ctx)) { err = twx_last_error(nullptr); return rc; }
        cfg.chips = nullptr;
        (void)hipGetDevice(&dev);
        st = (hipStream_t)twx_stream(ctx);
        twx_info info;
        twx_get_info(ctx, &info);
        P.n = info.n; P.L = cfg.chunk_samples; P.r = info.nphase; P.fs = cfg.fs;
        P.band_lo = cfg.band_lo_hz; P.band_hi = cfg.band_hi_hz; P.carrier = cfg.carrier; P.indice_floor = cfg.indice_floor;
        P.df_threshold = cfg.df_threshold;
        if (P.L < P.n || P.L % P.n) return fail(TWX_E_ARG, "chunk_samples must be a whole number of code periods");
        buf_samples = (size_t)(P.L + P.n + 64);
        rec_cap = (size_t)(P.L / P.n + 2);
        const size_t esz = cfg.precision == TWX_F64 ? 16 : 8;
        if (hipMalloc((void**)&buf, buf_samples * 4) != hipSuccess || hipMalloc((void**)&tail, (size_t)(P.n + 64) * 4) != hipSuccess ||
            hipMalloc((void**)&rec_dev, rec_cap * sizeof(twx_result)) != hipSuccess ||
            hipMalloc(&map_dev, (size_t)P.n * (size_t)P.r * esz) != hipSuccess || hipMalloc((void**)&stat_dev, 4 * sizeof(double)) != hipSuccess)
            return fail(TWX_E_NOMEM, "device allocation failed");
        for (auto& p : pin) if (hipHostMalloc(&p, (size_t)P.L * 4, hipHostMallocDefault) != hipSuccess) return fail(TWX_E_NOMEM, "pinned staging allocation failed");
        for (auto& p : stage_dev) if (hipMalloc((void**)&p, (size_t)P.L * 4) != hipSuccess) return fail(TWX_E_NOMEM, "device staging allocation failed");
        if (hipStreamCreateWithFlags(&cst, hipStreamNonBlocking) != hipSuccess) return fail(TWX_E_HIP, "copy stream creation failed");
        for (auto& e : h2d_ev) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return fail(TWX_E_HIP, "event creation failed");
        rec_host.resize(rec_cap);
        return TWX_OK;
    }

    // ---- capture source ----------------------------------------------------------------------------------
    // int16 positions [pos, pos + 2L) into dst; returns the int16 count delivered.  The chunk (40 MB for the scripts' 2-s
    // chunks) is fetched as IO_PIECES concurrent pieces: one pread / memcpy runs at ~5 GB/s, the PCIe copy behind it at ten
    // times that (same finding as twx_process_file's ingest)
    // (TWX_IO_THREADS pieces, default 8: with 4 the 180-s record ran at the speed of the page-cache copy, 25 GB/s — profiles/r04_tracked_rate.txt)
    enum { IO_PIECES_MAX = 32 };
    int io_pieces = [] { const char* e = getenv("TWX_IO_THREADS"); return e ? std::max(1, std::min((int)IO_PIECES_MAX, atoi(e))) : 8; }();
    size_t read_bytes(size_t off, char* dst, size_t len) const {           // [off, off+len) of the capture, in bytes
        if (host_src) {
            const size_t total = (size_t)src_i16 * 2;
            if (off >= total) return 0;
            const size_t n = std::min(len, total - off);
            memcpy(dst, reinterpret_cast<const char*>(host_src) + off, n);
            return n;
        }
        size_t done = 0;
        while (done < len) {
            const ssize_t g = pread(fd, dst + done, len - done, (off_t)(off + done));
            if (g <= 0) break;
            done += (size_t)g;
        }
        return done;
    }
    // stage >= 0: every piece that arrives whole goes on to stage_dev[stage] at once, on the copy stream, from the thread that read
    // it — the PCIe transfer of the first pieces runs beside the reads of the later ones; *staged_ok = every piece went out
    long long read_at(long long pos, void* dst, int stage = -1, bool* staged_ok = nullptr) const {
        const size_t need = (size_t)P.L * 4, off0 = (size_t)pos * 2;
        const int NP = io_pieces;
        const size_t piece = ((need + NP - 1) / NP + 4095) & ~(size_t)4095;
        std::future<size_t> parts[IO_PIECES_MAX];
        std::atomic<int> copy_failed{0};
        auto one = [this, off0, dst, stage, &copy_failed](size_t lo, size_t hi) -> size_t {
            const size_t got = read_bytes(off0 + lo, (char*)dst + lo, hi - lo);
            if (stage >= 0 && got == hi - lo && got) {
                (void)hipSetDevice(dev);
                if (hipMemcpyAsync(reinterpret_cast<char*>(stage_dev[stage]) + lo, (const char*)dst + lo, got, hipMemcpyHostToDevice, cst) != hipSuccess) copy_failed = 1;
            }
            return got;
        };
        for (int i = 1; i < NP; ++i) {
            const size_t lo = std::min(need, piece * i), hi = std::min(need, piece * (i + 1));
            if (hi > lo) parts[i] = std::async(std::launch::async, one, lo, hi);
        }
        size_t total = one(0, std::min(need, piece));
        bool contiguous = total == std::min(need, piece);
        for (int i = 1; i < NP; ++i) {
            if (!parts[i].valid()) continue;
            const size_t lo = std::min(need, piece * i), hi = std::min(need, piece * (i + 1));
            const size_t got = parts[i].get();
            if (contiguous) { total += got; contiguous = got == hi - lo; }
        }
        if (staged_ok) *staged_ok = copy_failed.load() == 0;
        return (long long)(total / 2);
    }

    int load_chunk(long long pos, long long carry, int* full) override {
        Tick tk(this, TWX_TRK_T_LOAD);
        if ((size_t)(carry + P.L) > buf_samples) return fail(TWX_E_STATE, "carry exceeds the sample buffer");
        long long got;
        bool staged = false;
        if (ahead.valid() && ahead_pos == pos) { got = ahead.get(); cur = ahead_slot; ahead_pos = -1; staged = got == 2 * P.L; }
        else { drop_ahead(); cur ^= 1; got = read_at(pos, pin[cur]); }
        if (got < 0) return fail(TWX_E_HIP, "read-ahead: copy of the chunk to the device failed");
        *full = got == 2 * P.L;
        if (!*full) return TWX_OK;
        if (staged) {
            // the helper thread has already sent the chunk to stage_dev[cur]: the context's stream waits for that copy, then moves
            // the chunk behind the tail (40 MB device to device: microseconds); nothing here waits on the host
            if (hipStreamWaitEvent(st, h2d_ev[cur], 0) != hipSuccess ||
                hipMemcpyAsync(buf + carry, stage_dev[cur], (size_t)P.L * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return fail(TWX_E_HIP, "staged chunk copy failed");
        } else {
            if (hipMemcpyAsync(buf + carry, pin[cur], (size_t)P.L * 4, hipMemcpyHostToDevice, st) != hipSuccess) return fail(TWX_E_HIP, "H2D copy failed");
            if (hipStreamSynchronize(st) != hipSuccess) return fail(TWX_E_HIP, "stream synchronize failed");
        }
        // read ahead: the chunk that follows, into the other pinned buffer and on to the device, while this one is measured.
        // pin[cur ^ 1] / stage_dev[cur ^ 1] held the chunk before this one: its measurements ended with a host synchronisation.
        ahead_slot = cur ^ 1; ahead_pos = pos + 2 * P.L;
        const int as = ahead_slot; const long long np = ahead_pos;
        ahead = std::async(std::launch::async, [this, np, as]() {
            bool ok = true;
            const long long g = read_at(np, pin[as], as, &ok);
            if (g == 2 * P.L) {
                (void)hipSetDevice(dev);
                if (!ok || hipEventRecord(h2d_ev[as], cst) != hipSuccess) return (long long)-1;    // load_chunk reports it
            }
            return g;
        });
        return TWX_OK;
    }

    int measure(long long start, int count, double df, twx_trk::Meas* m) override {
        Tick tk(this, TWX_TRK_T_MEASURE);
        if (start < 0 || (size_t)(start + (long long)count * P.n) > buf_samples || (size_t)count > rec_cap) return fail(TWX_E_STATE, "measure outside the sample buffer");
        std::vector<double> dfv((size_t)count, df);
        if (int rc = lib(twx_process_windows_dev(ctx, buf + start, count, 1, 0, nullptr, dfv.data(), rec_dev))) return rc;
        if (hipMemcpyAsync(rec_host.data(), rec_dev, sizeof(twx_result) * (size_t)count, hipMemcpyDeviceToHost, st) != hipSuccess) return fail(TWX_E_HIP, "D2H copy failed");
        if (int rc = lib(twx_synchronize(ctx))) return rc;
        for (int j = 0; j < count; ++j) {
            const twx_result& g = rec_host[(size_t)j];
            m[j] = twx_trk::Meas{(long long)g.indice0, g.correction, g.xval[0], g.xval[1], g.SNRr, g.SNRi, g.puissance, g.puissancecode, g.puissancenoise};
        }
        return TWX_OK;
    }
    int sq_bins(long long ns, const long long* bins, int nb, double* o) override {
        Tick tk(this, TWX_TRK_T_SQBINS);
        return lib(twx_sqspec_bins_dev(ctx, buf, ns, 1, 0, (const int64_t*)bins, nb, o));
    }
    int sq_band(long long off, long long k_lo, long long nk, double* mag) override {
        Tick tk(this, TWX_TRK_T_SQBAND);
        return lib(twx_sqspec_band_dev(ctx, buf + off, P.L, 1, 0, k_lo, nk, mag));
    }
    int candidate_snr(long long off, double dftmp, double* snr) override {
        Tick tk(this, TWX_TRK_T_CANDIDATE);
        // y=d(1:length(fcode)).*lo on the RAW chunk (:36): no mean removal for this call
        if (!search_mode) { if (int rc = lib(twx_set_option(ctx, TWX_OPT_REMOVE_MEAN, 0))) return rc; search_mode = true; }
        if (int rc = lib(twx_xcorr_map_dev(ctx, buf + off, 1, 0, dftmp, map_dev))) return rc;
        if (cfg.precision == TWX_F64) hipLaunchKernelGGL((k_map_snr<double>), dim3(1), dim3(1024), 0, st, (const double*)map_dev, (long long)P.n, P.r, stat_dev);
        else hipLaunchKernelGGL((k_map_snr<float>), dim3(1), dim3(1024), 0, st, (const float*)map_dev, (long long)P.n, P.r, stat_dev);
        if (hipGetLastError() != hipSuccess) return fail(TWX_E_HIP, "k_map_snr launch failed");
        double h[3];
        if (hipMemcpyAsync(h, stat_dev, sizeof h, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return fail(TWX_E_HIP, "D2H copy failed");
        *snr = h[0];
        return TWX_OK;
    }
    int search_done() override { return end_search(); }
    int end_search() {
        if (!search_mode) return TWX_OK;
        search_mode = false;
        return lib(twx_set_option(ctx, TWX_OPT_REMOVE_MEAN, 1));
    }
    int slide_tail(long long from, long long count) override {
        Tick tk(this, TWX_TRK_T_SLIDE);
        if (count < 0 || count > P.n + 64 || (size_t)(from + count) > buf_samples) return fail(TWX_E_STATE, "tail longer than a code period");
        if (!count) return TWX_OK;
        if (hipMemcpyAsync(tail, buf + from, (size_t)count * 4, hipMemcpyDeviceToDevice, st) != hipSuccess ||
            hipMemcpyAsync(buf, tail, (size_t)count * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return fail(TWX_E_HIP, "D2D copy failed");
        return TWX_OK;
    }

    int run(long long skip, long long kbon_hint, twx_tracked_summary* s) {
        (void)hipSetDevice(dev);
        cur = 0; ahead_pos = -1;
        for (int k = 0; k < TWX_TRK_NSTAGES; ++k) { t_stage[k] = 0; n_stage[k] = 0; }
        const auto t_run = std::chrono::steady_clock::now();
        int rc = twx_trk::run(P, *this, skip < 0 ? cfg.skip_samples : skip, kbon_hint, out);
        t_stage[TWX_TRK_T_TOTAL] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_run).count(); n_stage[TWX_TRK_T_TOTAL] = 1;
        drop_ahead();
        const int rc2 = end_search();
        if (rc == -1 && err.empty()) rc = fail(TWX_E_ARG, "the search band holds no bin of the chunk axis");
        if (!rc) rc = rc2;
        if (s) {
            s->n_codes = (int64_t)out.codes.size(); s->n_chunks = (int64_t)out.df.size(); s->n_moved = (int64_t)out.moved.size();
            s->kbon = out.kbon; s->batches = out.batches; s->puissancecode = out.pcode; s->puissancenoise = out.pnoise;
        }
        return rc;
    }
};

template <class F> static int trk_guard(twx_tracked* t, F f) noexcept {
    // the launches of this library are checked with hipGetLastError(): an error another library left behind on this thread
    // (RCCL and PyTorch probe pointers and peers and do not clear what those probes set) must not be taken for ours
    (void)hipGetLastError();
    try {
        if (t && t->ctx) { twx::FenceShared fence(twx::ctx_device(t->ctx), twx::ctx_stream(t->ctx)); return f(); }     // never beside a matrix-core FIR (twx_internal.h)
        return f();
    }
    catch (const std::bad_alloc&) { if (t) t->err = "out of host memory"; return TWX_E_NOMEM; }
    catch (const std::exception& e) { if (t) t->err = std::string("internal error: ") + e.what(); return TWX_E_STATE; }
    catch (...) { if (t) t->err = "internal error"; return TWX_E_STATE; }
}

extern "C" {

int twx_tracked_defaults(int32_t mode, int32_t OP, double fs, twx_tracked_config* cfg) {
    if (!cfg || !(fs > 0) || mode < TWX_TRK_RANGING || mode > TWX_TRK_LO) return TWX_E_ARG;
    cfg->fs = fs; cfg->sps = 2; cfg->nint = 1;
    cfg->chunk_samples = (int64_t)llround(fs * 2);          // ls=2 (:16)
    cfg->df_threshold = 20.0;                               // :20
    cfg->carrier = TWX_CARRIER_SEARCH_DF; cfg->indice_floor = 0;
    cfg->skip_samples = (int64_t)llround(30 * fs);          // fseek(f,30*fs*2*2) (:128)
    switch (mode) {
        case TWX_TRK_RANGING: cfg->band_lo_hz = -8000; cfg->band_hi_hz = 8000; break;                       // :135
        case TWX_TRK_RE:                                                                                     // :137-141
            if (OP == 1) { cfg->band_lo_hz = -108000; cfg->band_hi_hz = -92000; } else { cfg->band_lo_hz = 92000; cfg->band_hi_hz = 108000; }
            break;
        default:                                                                                             // lo :105-106,126-134
            cfg->band_lo_hz = -20000; cfg->band_hi_hz = 20000; cfg->carrier = TWX_CARRIER_CHUNK_BAND; cfg->indice_floor = 1;
            cfg->skip_samples = 0;
            break;
    }
    return TWX_OK;
}

int twx_tracked_create(const twx_tracked_config* cfg, twx_tracked** out) {
    if (!cfg || !out) { g_trk_create_err = "null argument"; return TWX_E_ARG; }
    *out = nullptr;
    if (!(cfg->fs > 0) || cfg->chunk_samples < 1 || !(cfg->band_lo_hz < cfg->band_hi_hz) || cfg->skip_samples < 0 ||
        (cfg->carrier != TWX_CARRIER_SEARCH_DF && cfg->carrier != TWX_CARRIER_CHUNK_BAND)) { g_trk_create_err = "bad fs / chunk_samples / band / carrier / skip"; return TWX_E_ARG; }
    twx_tracked* t = new (std::nothrow) twx_tracked();
    if (!t) { g_trk_create_err = "out of host memory"; return TWX_E_NOMEM; }
    t->cfg = *cfg;
    const int rc = trk_guard(t, [&]() { return t->init(); });
    if (rc) { g_trk_create_err = t->err; delete t; return rc; }
    *out = t;
    return TWX_OK;
}
void twx_tracked_destroy(twx_tracked* t) { delete t; }
const char* twx_tracked_last_error(const twx_tracked* t) { return t ? t->err.c_str() : g_trk_create_err.c_str(); }
twx_ctx* twx_tracked_context(twx_tracked* t) { return t ? t->ctx : nullptr; }

int twx_tracked_file(twx_tracked* t, const char* path, int64_t skip_samples, int64_t kbon_hint, twx_tracked_summary* summary) {
    if (!t) return TWX_E_ARG;
    if (!path) return t->fail(TWX_E_ARG, "null path");
    // the descriptor is closed on EVERY exit path, an exception out of run() (bad_alloc from the record vectors or from
    // std::async) included: a long-running MEX / Octave host must not leak one per failed capture
    struct Fd {
        twx_tracked* t; int fd;
        ~Fd() { t->drop_ahead(); if (fd >= 0) close(fd); t->fd = -1; }
    } guard{t, open(path, O_RDONLY)};
    if (guard.fd < 0) return t->fail(TWX_E_ARG, std::string("cannot open ") + path);
    return trk_guard(t, [&]() {
        t->fd = guard.fd; t->host_src = nullptr; t->src_i16 = 0;
        return t->run(skip_samples, kbon_hint, summary);
    });
}
int twx_tracked_host(twx_tracked* t, const int16_t* iq, int64_t n_samples, int64_t skip_samples, int64_t kbon_hint, twx_tracked_summary* summary) {
    if (!t) return TWX_E_ARG;
    if (!iq || n_samples < 0) return t->fail(TWX_E_ARG, "bad capture buffer");
    return trk_guard(t, [&]() {
        t->fd = -1; t->host_src = iq; t->src_i16 = 2 * n_samples;
        const int rc = t->run(skip_samples, kbon_hint, summary);
        t->drop_ahead();
        t->host_src = nullptr;
        return rc;
    });
}
int twx_tracked_timing(const twx_tracked* t, double* seconds, int64_t* calls) {
    if (!t || !seconds) return TWX_E_ARG;
    for (int k = 0; k < TWX_TRK_NSTAGES; ++k) { seconds[k] = t->t_stage[k]; if (calls) calls[k] = t->n_stage[k]; }
    return TWX_OK;
}
int twx_tracked_fetch(twx_tracked* t, twx_tracked_code* codes, double* df, int64_t* moved, double* movedval) {
    if (!t) return TWX_E_ARG;
    const twx_trk::Output& o = t->out;
    if (codes) for (size_t i = 0; i < o.codes.size(); ++i) {
        const twx_trk::Code& c = o.codes[i];
        codes[i] = twx_tracked_code{{c.xre, c.xim}, c.indice1, c.correction1, c.snr_r, c.snr_i, c.puissance1};
    }
    if (df) for (size_t i = 0; i < o.df.size(); ++i) df[i] = o.df[i];
    if (moved) for (size_t i = 0; i < o.moved.size(); ++i) moved[i] = o.moved[i];
    if (movedval) for (size_t i = 0; i < o.movedval.size(); ++i) movedval[i] = o.movedval[i];
    return TWX_OK;
}
int twx_tracked_search_df(twx_tracked* t, const int16_t* iq, int64_t n_samples, int64_t* kbon) {
    if (!t) return TWX_E_ARG;
    if (!iq || !kbon || n_samples < t->P.L) return t->fail(TWX_E_ARG, "search_df needs one whole chunk");
    return trk_guard(t, [&]() {
        (void)hipSetDevice(t->dev);
        t->fd = -1; t->host_src = iq; t->src_i16 = 2 * n_samples; t->cur = 0; t->ahead_pos = -1;
        int full = 0;
        int rc = t->load_chunk(0, 0, &full);
        t->drop_ahead();
        t->host_src = nullptr;
        if (rc) return rc;
        const twx_trk::FreqAxis freq(t->P.fs, t->P.L);
        long long k0 = -1, nk = 0, kb = -1;
        twx_trk::band_indices(freq, t->P.band_lo, t->P.band_hi, &k0, &nk);
        rc = twx_trk::search_df(t->P, *t, freq, k0, nk, &kb);
        const int rc2 = t->end_search();
        *kbon = kb;
        return rc ? rc : rc2;
    });
}

}  // extern "C"
