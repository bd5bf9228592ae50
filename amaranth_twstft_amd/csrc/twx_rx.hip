// twx_rx.hip — the DLL/PLL receiver of experiments/231001_DLL_PLL/rxcomplex.cpp behind the C ABI (twx_rx_* in
// include/twstft_hip.h): parameter file in, .dat rows out.  With cfg.ninterp = 1 it is the other program of that directory,
// rx.cpp: real samples, no interpolation, successive interference cancellation for the 'S' rows (see "real-sample program"
// below); line numbers without a file name are rxcomplex.cpp's.
//
// What runs where.  Everything that touches samples is a kernel of this library: the x2 FFT-domain interpolation of both
// physical channels (short2double :914-963 = the fused chain with two output phases and a weight vector as "code spectrum"),
// the received power (:481-489), the acquisition sweep (:521-567 = twx_acquire_cdev on the channel's own context, whose
// replica spectrum is the zero-padded sampled code, :416-437), the tracking correlations (:593-605 = the direct sliding
// dot product on the interpolated complex stream).  The replica set-up (PRN_sampling :965-978, memcpy_acq :980-987, lowpass
// :1020-1037, psbb :431-432, the replica FFT :434-437, cross_spectrum's mask and 1/n^2 :1001-1018) is a handful of small
// kernels below around the contexts' device-to-device transform (twx_fft_forward_dev, fp64).  The host keeps what the program
// keeps in channel_info: flags, carrier, code phase, the two weighted fits (twx_track_update), the text of the rows.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <map>
#include <memory>
#include <new>
#include <random>
#include <string>
#include <vector>
#include "twx_internal.h"

namespace {

thread_local std::string g_rx_create_err;
typedef double2 cd;

// PRN_sampling :965-978: idx = floor(fmod((i/fs - delay*1e-9)*rc, clen)), wrapped; value code[idx] = 1 - 2*byte (SDRcode :879)
__global__ void k_rx_prn_sampling(long long nobs, const unsigned char* __restrict__ code, double rc, double fs, int clen, double delay_ns, cd* __restrict__ prn) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nobs; i += (long long)gridDim.x * blockDim.x) {
        int idx = (int)floor(fmod(((double)i / fs - delay_ns * 1.0e-9) * rc, (double)clen));
        if (idx < 0) idx += clen; else if (idx >= clen) idx -= clen;
        prn[i] = make_double2((double)(1 - 2 * (int)code[idx]), 0.0);
    }
}
// memcpy_acq :980-987 into a zeroed nfft buffer: wav_acq[i].re = wav[i*dec].re for i < m, everything else 0
__global__ void k_rx_memcpy_acq(long long nfft, long long m, int dec, const cd* __restrict__ wav, cd* __restrict__ acq) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nfft; i += (long long)gridDim.x * blockDim.x)
        acq[i] = make_double2(i < m ? wav[i * dec].x : 0.0, 0.0);
}
// the pass-band test shared by lowpass :1025-1026 and cross_spectrum :1007-1008 on the signed bin index
__device__ __forceinline__ bool rx_in_band(long long i, long long n, double df, double fmax, double fmin) {
    const long long idx = (i >= n / 2) ? i - n : i;
    return (double)idx * df < fmax && (double)idx * df > fmin && idx != 0;
}
// lowpass :1020-1037 followed by the conjugation that turns the next FORWARD transform into FFTW's unnormalised backward one
__global__ void k_rx_lowpass_conj(long long n, cd* __restrict__ s, double df, double fmax, double fmin) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const cd v = s[i];
        s[i] = rx_in_band(i, n, df, fmax, fmin) ? make_double2(v.x / (double)n, -v.y / (double)n) : make_double2(0.0, 0.0);
    }
}
// filt = conj(t): psbb partial sums of |filt|^2 (dznrm2^2 :431) and the tracking replica = real part (PRN_mapping takes .real(), :997)
__global__ __launch_bounds__(256) void k_rx_replica_out(long long n, const cd* __restrict__ t, float* __restrict__ replica, double* __restrict__ partial) {
    double acc = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const cd v = t[i];
        replica[i] = (float)v.x;
        acc += v.x * v.x + v.y * v.y;
    }
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_down(acc, d, 64);
    __shared__ double sh[4];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
// operand of the acquisition context: cross_spectrum's obs*conj(prn)/n^2 inside the pass-band (:1001-1018) with downconv_acq's
// sqrt(2) (:1046-1047, linear) folded in; one 1/n is the ifft normalisation the context applies to its maps
__global__ void k_rx_acq_spec(long long n, cd* __restrict__ s, double df, double fmax, double fmin) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const cd v = s[i];
        const double g = 1.4142135624 / (double)n;
        s[i] = rx_in_band(i, n, df, fmax, fmin) ? make_double2(v.x * g, -v.y * g) : make_double2(0.0, 0.0);
    }
}
// short2double's spectrum handling as a weight vector (:931-938,957-959): the lower half of the nobs/Ninterp-point spectrum
// stays as it is, the upper half moves to the top divided by nobs/Ninterp; input scale 1/32768 (:922-928); the final /nobs is
// the context's own ifft normalisation
__global__ void k_rx_interp_weights(long long half, cd* __restrict__ w) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < half; i += (long long)gridDim.x * blockDim.x)
        w[i] = make_double2((i < half / 2 ? 1.0 : 1.0 / (double)half) / 32768.0, 0.0);
}
// received power :481-489: zdotc over every dec-th sample; per-workgroup partials, added on the host in the fixed order
__global__ __launch_bounds__(256) void k_rx_power(long long n, int dec, const float2* __restrict__ smp, double* __restrict__ partial) {
    double acc = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float2 v = smp[i * dec];
        acc += (double)v.x * (double)v.x + (double)v.y * (double)v.y;
    }
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_down(acc, d, 64);
    __shared__ double sh[4];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// ---- real-sample program (rx.cpp) ----
// short2double rx.cpp:892-900: the I sample of physical channel A (frame offset 0) / B (offset 2), /32768; kept as a complex
// float with a zero imaginary part so that the acquisition and tracking kernels of the complex program run on it unchanged
// (downconv_acq rx.cpp:976-986 and downconv_trk rx.cpp:988-998 are the complex ones with imag(smp) = 0)
__global__ void k_rx_real_in(long long n, const short2* __restrict__ frames, int chan, float2* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = make_float2((float)frames[2 * i + chan].x * (1.0f / 32768.0f), 0.0f);
}
// MAI_up rx.cpp:1011-1020: the signal of one tracked channel rebuilt from its per-period records, added to acc[].x
__global__ void k_rx_mai_up(long long n, long long ld, long long pt, const double* __restrict__ amp, const float* __restrict__ wav,
                            const int* __restrict__ pidx, double ff, const double* __restrict__ pmod, int bps, float2* __restrict__ acc) {
    for (long long i = pt + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long p = (i - pt) / ld;
        if (p >= bps) continue;                                  // the program's arrays end there (pt >= 0: never taken)
        const long long k = (i - pidx[p] - pt + ld) % ld;
        double t = ff * (double)i + pmod[p];
        t -= floor(t);
        acc[i].x += (float)(0.5 * amp[p] * (double)wav[k] * cospi(2.0 * t));
    }
}
// MAI_out rx.cpp:1022-1027: out = in - out
__global__ void k_rx_mai_out(long long n, const float2* __restrict__ in, float2* __restrict__ acc) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        acc[i] = make_float2(in[i].x - acc[i].x, 0.0f);
}

double v2todBm(double v2) { return v2 > 0.0 ? 10.0 * log10(v2 * 1000.0 / 25.0) : 0.0; }      // :1236-1240

struct Channel {
    twx_rx_row row{};
    bool is_chA = true, is_sic = false;
    int cid = 0, rc = 0, clen = 0, nlag = 0, bps = 0;
    long long nobs = 0, nfft = 0;
    double duration = 0, fc_init = 0, fltmax = 0, fltmin = 0, range = 0, step = 0, snr_min = 0, psbb = 0;
    bool is_trk = false, is_first = false;
    twx_track_state st{};
    double gd = 0, dg = 0, sdgd = 0, pk = 0, px = 0;
    int cnt = 0;
    twx_ctx* acq = nullptr;         // the acquisition context of this row's transform length — SHARED by all rows of that length (twx_rx::acq_ctx)
    void* acq_spec = nullptr;       // this row's acquisition operand (natural order, complex double), loaded into the shared context before a sweep
    float* replica_dev = nullptr;
    std::string dat_name;
    // the records MAI_up reads (rx.cpp:664-666,757), host and device: amp[bps], phase[bps] (double), pk_idx[bps] (int)
    std::vector<double> mai_amp, mai_phase;
    std::vector<int32_t> mai_pk;
    void* mai_dev = nullptr;
    int shown() const { return is_sic ? cid + 50 : cid; }      // the PRN the program prints (rx.cpp:708,745)
};

constexpr int RX_PARTS = 1024;

}  // namespace

struct twx_rx {
    twx_rx_config cfg{};
    std::string code_dir, out_dir, err;
    int dev = 0;
    double fs = 0;                 // si.fs = sps / dec (:236)
    long long n_in = 0, sps = 0;
    twx_ctx* interp = nullptr;
    short2* iq_dev = nullptr;      // one second of [IA QA IB QB] frames
    float2* smp[2] = {nullptr, nullptr};
    float2* mai_free = nullptr;    // dev_smp_MAI_free (rx.cpp:251), allocated when the list has an 'S' row
    hipStream_t own_stream = nullptr;
    bool real = false, any_sic = false;
    double* part_dev = nullptr;
    std::vector<Channel> ch;
    std::mt19937_64 rng;
    bool need[2] = {false, false};
    double last_pwr[2] = {0, 0};
    // One fp32 acquisition context per distinct transform length (a context of 2^20 points with its three pipeline slots is
    // about 3 GB: one per ROW would not fit the program's 120 rows, rxcomplex.cpp:34).  Rows of one length differ only in the
    // operand conj(FFT(replica)), which each row keeps (16 bytes a point) and loads before its sweep when it is not the one in place.
    std::map<long long, twx_ctx*> acq_ctx;
    std::map<twx_ctx*, const Channel*> acq_loaded;
    int load_acq_operand(const Channel& c) {
        auto it = acq_loaded.find(c.acq);
        if (it != acq_loaded.end() && it->second == &c) return TWX_OK;
        if (int rc = lib(c.acq, twx_set_code_spectrum_dev(c.acq, c.acq_spec))) return rc;
        acq_loaded[c.acq] = &c;
        return TWX_OK;
    }

    int fail(int code, const std::string& m) { err = m; return code; }
    int lib(twx_ctx* c, int rc) { if (rc) { const char* m = twx_last_error(c); err = m && *m ? m : twx_strerror(rc); } return rc; }
    ~twx_rx() {
        (void)hipSetDevice(dev);
        for (auto& c : ch) { if (c.acq_spec) (void)hipFree(c.acq_spec); if (c.replica_dev) (void)hipFree(c.replica_dev); if (c.mai_dev) (void)hipFree(c.mai_dev); }
        for (auto& kv : acq_ctx) twx_destroy(kv.second);
        if (interp) twx_destroy(interp);
        for (void* p : {(void*)iq_dev, (void*)smp[0], (void*)smp[1], (void*)part_dev, (void*)mai_free}) if (p) (void)hipFree(p);
        if (own_stream) { twx::fence_unregister(dev, own_stream); (void)hipStreamDestroy(own_stream); }
    }
    hipStream_t stream() const { return interp ? (hipStream_t)twx_stream(interp) : own_stream; }
    void log_line(const char* text) const {
        if (out_dir.empty()) return;
        FILE* f = fopen((out_dir + (real ? "/rxreal.log" : "/rxcomplex.log")).c_str(), "a");     // rx.cpp:440 / rxcomplex.cpp:439
        if (f) { fputs(text, f); fclose(f); }
    }
    static twx_config plain_cfg(long long n, double fs, int precision, int nphase, int max_batch, int device) {
        twx_config c;
        memset(&c, 0, sizeof c);
        c.fs = fs; c.sps = 1; c.nint = 0; c.chips = nullptr; c.n_chips = n; c.lfsr_bitlen = 20; c.lfsr_taps = 9;     // any code: its spectrum is replaced
        c.precision = precision; c.nphase = nphase; c.max_batch = max_batch; c.device = device; c.snr_rot = -1;
        return c;
    }
    // fp64 work contexts of the set-up, one per transform length, dropped when the set-up is done
    struct Work {
        std::map<long long, twx_ctx*> m;
        ~Work() { for (auto& kv : m) twx_destroy(kv.second); }
    };
    int work_ctx(Work& w, long long n, twx_ctx** out) {
        auto it = w.m.find(n);
        if (it != w.m.end()) { *out = it->second; return TWX_OK; }
        const twx_config c = plain_cfg(n, 1.0, TWX_F64, 1, 1, dev);
        twx_ctx* x = nullptr;
        if (int rc = twx_create(&c, &x)) { err = std::string("fp64 work context of ") + std::to_string(n) + " points: " + twx_last_error(nullptr); return rc; }
        w.m[n] = x; *out = x;
        return TWX_OK;
    }

    int setup_channel(Channel& c, Work& work, const std::vector<uint8_t>& code) {
        const unsigned g = 1024, b = 256;
        unsigned char* code_dev = nullptr; cd *wav_t = nullptr, *wav_acq = nullptr; double* part = nullptr;
        struct Free { std::vector<void*> p; ~Free() { for (void* q : p) if (q) (void)hipFree(q); } } fr;
        auto alloc = [&](void** p, size_t bytes) { if (hipMalloc(p, bytes) != hipSuccess) return false; fr.p.push_back(*p); return true; };
        if (!alloc((void**)&code_dev, code.size()) || !alloc((void**)&wav_t, (size_t)c.nobs * 16) || !alloc((void**)&part, RX_PARTS * 8))
            return fail(TWX_E_NOMEM, "device allocation failed (channel set-up)");
        if (hipMalloc((void**)&wav_acq, (size_t)c.nfft * 16) != hipSuccess) return fail(TWX_E_NOMEM, "device allocation failed (acquisition operand)");
        c.acq_spec = wav_acq;                                                            // kept: released with the receiver
        if (hipMalloc((void**)&c.replica_dev, (size_t)c.nobs * 4) != hipSuccess) return fail(TWX_E_NOMEM, "device allocation failed (replica)");
        if (hipMemcpy(code_dev, code.data(), code.size(), hipMemcpyHostToDevice) != hipSuccess) return fail(TWX_E_HIP, "code upload failed");
        twx_ctx *cn = nullptr, *cf = nullptr;
        if (int rc = work_ctx(work, c.nobs, &cn)) return rc;
        if (int rc = work_ctx(work, c.nfft, &cf)) return rc;
        hipStream_t sn = (hipStream_t)twx_stream(cn), sf = (hipStream_t)twx_stream(cf);
        // sampled code waveform (:416) and its zero-padded, decimated copy for the acquisition (:418-420), on cn's stream
        hipLaunchKernelGGL(k_rx_prn_sampling, dim3(g), dim3(b), 0, sn, c.nobs, code_dev, (double)c.rc, fs, c.clen, 0.0, wav_t);
        hipLaunchKernelGGL(k_rx_memcpy_acq, dim3(g), dim3(b), 0, sn, c.nfft, c.nobs / cfg.dec_a, cfg.dec_a, wav_t, wav_acq);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(sn) != hipSuccess) return fail(TWX_E_HIP, "replica sampling failed");
        // filtered waveform (:422-428): FFT, brick wall / nobs, unnormalised backward transform = conj(FFT(conj(.)))
        if (int rc = lib(cn, twx_fft_forward_dev(cn, wav_t, wav_t))) return rc;
        hipLaunchKernelGGL(k_rx_lowpass_conj, dim3(g), dim3(b), 0, sn, c.nobs, wav_t, fs / (double)c.nobs, c.fltmax, c.fltmin);
        if (int rc = lib(cn, twx_fft_forward_dev(cn, wav_t, wav_t))) return rc;
        hipLaunchKernelGGL(k_rx_replica_out, dim3(RX_PARTS), dim3(b), 0, sn, c.nobs, wav_t, c.replica_dev, part);       // |conj(t)| = |t|, real part unchanged
        std::vector<double> hp(RX_PARTS);
        if (hipGetLastError() != hipSuccess || hipMemcpyAsync(hp.data(), part, RX_PARTS * 8, hipMemcpyDeviceToHost, sn) != hipSuccess ||
            hipStreamSynchronize(sn) != hipSuccess) return fail(TWX_E_HIP, "replica filtering failed");
        double nrm2 = 0;
        for (double v : hp) nrm2 += v;
        c.psbb = nrm2 / (double)c.nobs;                                                  // :431-432
        // acquisition operand (:434-437 + cross_spectrum): FFT of the zero-padded copy, conj, mask, sqrt(2)/n
        if (int rc = lib(cf, twx_fft_forward_dev(cf, wav_acq, wav_acq))) return rc;
        hipLaunchKernelGGL(k_rx_acq_spec, dim3(g), dim3(b), 0, sf, c.nfft, wav_acq, (fs / (double)cfg.dec_a) / (double)c.nfft, c.fltmax, c.fltmin);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(sf) != hipSuccess) return fail(TWX_E_HIP, "acquisition operand failed");
        auto it = acq_ctx.find(c.nfft);
        if (it == acq_ctx.end()) {
            const twx_config ac = plain_cfg(c.nfft, fs / (double)cfg.dec_a, TWX_F32, 0, 64, dev);
            twx_ctx* x = nullptr;
            if (int rc = twx_create(&ac, &x)) { err = std::string("acquisition context of ") + std::to_string(c.nfft) + " points: " + twx_last_error(nullptr); return rc; }
            it = acq_ctx.emplace(c.nfft, x).first;
            if (int rc = lib(x, twx_set_option(x, TWX_OPT_REMOVE_MEAN, 0))) return rc;
        }
        c.acq = it->second;
        return TWX_OK;
    }

    int init(const twx_rx_row* rows, int n_rows) {
        if (!(cfg.fs_in >= 1000.0) || (cfg.ninterp != 1 && cfg.ninterp != 2) || (cfg.dec_a != 1 && cfg.dec_a != 2))
            return fail(TWX_E_ARG, "need fs_in >= 1000, ninterp = 2 (rxcomplex.cpp:29) or 1 (rx.cpp), dec_a 1 or 2");
        real = cfg.ninterp == 1;
        if (n_rows < 1 || n_rows > 120) return fail(TWX_E_ARG, "1..120 channels (nch_max, rxcomplex.cpp:34)");
        if (cfg.device >= 0 && hipSetDevice(cfg.device) != hipSuccess) return fail(TWX_E_HIP, "hipSetDevice failed");
        if (hipGetDevice(&dev) != hipSuccess) return fail(TWX_E_HIP, "no HIP device available (the HIP path has no CPU fallback)");
        n_in = (long long)llround(cfg.fs_in);
        sps = n_in * cfg.ninterp;                                                     // :33
        fs = (double)sps;                                                              // si.fs :236 (dec = 1)
        rng.seed(cfg.seed);
        Work work;
        for (int i = 0; i < n_rows; ++i) {
            const twx_rx_row& r = rows[i];
            Channel c;
            c.row = r; c.row.code = nullptr;
            if (r.ch != 'A' && r.ch != 'B') return fail(TWX_E_ARG, "row " + std::to_string(i) + ": channel must be A or B");
            if (r.mode == 'S' && !real) return fail(TWX_E_ARG, "row " + std::to_string(i) + ": SIC rows need ninterp = 1 (rx.cpp); the code is commented out in rxcomplex.cpp:508-519");
            if (r.mode != 'N' && r.mode != 'S') return fail(TWX_E_ARG, "row " + std::to_string(i) + ": mode must be N or S");
            c.is_sic = r.mode == 'S';
            any_sic = any_sic || c.is_sic;
            // the acceptance test of :288
            if (!(r.pn >= 0 && r.pn <= 131 && r.kcps == 2500 && r.fc_init >= -200000. && r.fc_init < 200000. && r.frange >= 0.0 && r.frange < 200000. &&
                  r.frange > r.fstep && r.snr_min_db > -100.0)) return fail(TWX_E_ARG, "row " + std::to_string(i) + ": values outside the ranges of rxcomplex.cpp:288");
            c.is_chA = r.ch == 'A';
            c.rc = r.kcps * 1000;
            if (r.pn < 100) { c.clen = 10000; c.duration = 0.004; c.nlag = 14; }                  // :299-304
            else { c.clen = 100000; c.duration = 0.04; c.nlag = 28; }                              // :305-311
            c.bps = c.rc / c.clen;                                                                 // :363
            c.nobs = sps / c.bps;                                                                  // :364
            if (c.bps < 2 || c.nobs < 64) return fail(TWX_E_ARG, "row " + std::to_string(i) + ": sample rate too low for this code");
            c.cid = r.pn; c.fc_init = r.fc_init; c.fltmax = (double)c.rc; c.fltmin = -c.fltmax;   // :365-368
            c.nfft = 1;
            while (true) { c.nfft *= 2; if (c.nfft > c.nobs * 2 / cfg.dec_a) break; }              // :369-374
            c.range = 1.0; while (c.range < r.frange) c.range *= 2.0;                              // :375-376
            c.step = 1.0; while (c.step < r.fstep) c.step *= 2.0;                                  // :377-378
            c.snr_min = pow(10.0, r.snr_min_db / 10.0);                                            // :379
            if ((sps - c.nfft * cfg.dec_a) / c.nobs < 1) return fail(TWX_E_ARG, "row " + std::to_string(i) + ": the one-second buffer is shorter than the acquisition window");
            char nm[64];
            snprintf(nm, sizeof nm, "ch%s.pn%02d.%dkcps.dat", c.is_chA ? "A" : "B", c.shown(), c.rc / 1000);   // :720, rx.cpp:708
            c.dat_name = nm;
            std::vector<uint8_t> code;
            if (r.code) {
                if (r.code_len < c.clen) return fail(TWX_E_ARG, "row " + std::to_string(i) + ": code shorter than the code length");
                code.assign(r.code, r.code + c.clen);
            } else {
                if (r.pn < 100) return fail(TWX_E_ARG, "row " + std::to_string(i) + ": PRN < 100 has no code source in the program (SDRcode only reads <pn-100>.bin, rxcomplex.cpp:873); pass the chips with the row");
                const std::string path = code_dir + "/" + std::to_string(r.pn - 100) + ".bin";     // :875
                FILE* f = fopen(path.c_str(), "rb");
                if (!f) return fail(TWX_E_ARG, "Code filename error " + path);
                code.resize((size_t)c.clen);
                const size_t got = fread(code.data(), 1, code.size(), f);
                fclose(f);
                if (got != code.size()) return fail(TWX_E_ARG, "code file " + path + " is shorter than " + std::to_string(c.clen) + " chips");
            }
            for (uint8_t v : code) if (v > 1) return fail(TWX_E_ARG, "row " + std::to_string(i) + ": code bytes must be 0/1");
            need[c.is_chA ? 0 : 1] = true;
            ch.push_back(c);
            if (int rc = setup_channel(ch.back(), work, code)) return rc;
            Channel& k = ch.back();
            memset(&k.st, 0, sizeof k.st);
            k.st.fs = fs; k.st.duration = k.duration; k.st.psbb = k.psbb;
            char line[256];
            if (real) {
                k.mai_amp.assign((size_t)k.bps, 0.0); k.mai_phase.assign((size_t)k.bps, 0.0); k.mai_pk.assign((size_t)k.bps, 0);
                if (hipMalloc(&k.mai_dev, (size_t)k.bps * 20) != hipSuccess) return fail(TWX_E_NOMEM, "device allocation failed (interference records)");
            }
            snprintf(line, sizeof line, "set param   : Ch. %s, PRN#%2d, %8.0lf %4d %5.0lf %5.0lf %5.0lf %3.0lf\n", k.is_chA ? "A" : "B", k.shown(), k.fc_init,
                     k.rc / 1000, k.fltmax * 1.0e-3, k.range, k.step, k.snr_min);                  // :441
            log_line(line);
        }
        if (real) {                                                                   // rx.cpp: no interpolation, the samples as they come
            if (hipStreamCreateWithFlags(&own_stream, hipStreamNonBlocking) != hipSuccess) return fail(TWX_E_HIP, "stream creation failed");
            twx::fence_register(dev, own_stream);
            if (any_sic && hipMalloc((void**)&mai_free, (size_t)sps * 8) != hipSuccess) return fail(TWX_E_NOMEM, "device allocation failed (interference-free stream)");
        } else {
        // interpolator: the fused chain with two output phases and short2double's weights as replica spectrum
        const twx_config ic = plain_cfg(n_in, 1.0, TWX_F32, cfg.ninterp, 1, dev);
        if (int rc = twx_create(&ic, &interp)) { err = std::string("interpolation context of ") + std::to_string(n_in) + " points: " + twx_last_error(nullptr); return rc; }
        {
            cd* w = nullptr;
            if (hipMalloc((void**)&w, (size_t)n_in * 16) != hipSuccess) return fail(TWX_E_NOMEM, "device allocation failed (weights)");
            hipLaunchKernelGGL(k_rx_interp_weights, dim3(1024), dim3(256), 0, (hipStream_t)twx_stream(interp), n_in, w);
            int rc = hipGetLastError() == hipSuccess ? TWX_OK : TWX_E_HIP;
            if (!rc) rc = twx_set_code_spectrum_dev(interp, w);
            (void)hipFree(w);
            if (rc) return lib(interp, rc);
        }
        if (int rc = lib(interp, twx_set_option(interp, TWX_OPT_REMOVE_MEAN, 0))) return rc;
        }
        if (hipMalloc((void**)&iq_dev, (size_t)n_in * 8) != hipSuccess || hipMalloc((void**)&part_dev, RX_PARTS * 8) != hipSuccess) return fail(TWX_E_NOMEM, "device allocation failed");
        for (int p = 0; p < 2; ++p)
            if (need[p] && hipMalloc((void**)&smp[p], (size_t)sps * 8) != hipSuccess) return fail(TWX_E_NOMEM, "device allocation failed (stream)");
        return TWX_OK;
    }

    int power(const float2* buf, double* out) {
        hipStream_t s = stream();
        hipLaunchKernelGGL(k_rx_power, dim3(RX_PARTS), dim3(256), 0, s, sps / cfg.dec_a, cfg.dec_a, buf, part_dev);
        std::vector<double> hp(RX_PARTS);
        if (hipGetLastError() != hipSuccess || hipMemcpyAsync(hp.data(), part_dev, RX_PARTS * 8, hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess) return fail(TWX_E_HIP, "power reduction failed");
        double acc = 0;
        for (double v : hp) acc += v;
        *out = acc / (fs / (double)cfg.dec_a);                                                      // :483,487
        return TWX_OK;
    }

    // the loop body :468-832 on the second that sits in `src` (device)
    int second(const void* src, twx_rx_report* rep) {
        double pwr[2] = {0, 0};
        for (int p = 0; p < 2; ++p) {
            if (!need[p]) continue;
            if (real) hipLaunchKernelGGL(k_rx_real_in, dim3(2048), dim3(256), 0, stream(), sps, static_cast<const short2*>(src), p, smp[p]);   // rx.cpp:478
            else if (int rc = lib(interp, twx_xcorr_map_dev(interp, src, 2, p, 0.0, smp[p]))) return rc;  // short2double :477
            if (int rc = power(smp[p], &pwr[p])) return rc;                                                // synchronises the stream: smp[p] is complete
        }
        last_pwr[0] = pwr[0]; last_pwr[1] = pwr[1];
        for (size_t i = 0; i < ch.size(); ++i) {
            Channel& c = ch[i];
            twx_rx_report& r = rep[i];
            memset(&r, 0, sizeof r);
            const int p = c.is_chA ? 0 : 1;
            c.px = pwr[p];                                                                                  // :493-503
            const float2* obs = smp[p];
            if (c.is_sic) {                                                                                 // rx.cpp:505-518
                hipStream_t s = stream();
                if (hipMemsetAsync(mai_free, 0, (size_t)sps * 8, s) != hipSuccess) return fail(TWX_E_HIP, "memset failed");
                for (size_t k = 0; k < i; ++k) {
                    const Channel& o = ch[k];
                    if (o.is_chA != c.is_chA || o.cid == c.cid || o.is_sic || !o.is_trk || o.is_first) continue;      // rx.cpp:511
                    const double* amp = static_cast<const double*>(o.mai_dev);
                    hipLaunchKernelGGL(k_rx_mai_up, dim3(2048), dim3(256), 0, s, sps, o.nobs, std::max<long long>(o.st.pt_prev, 0), amp, o.replica_dev,
                                       reinterpret_cast<const int*>(amp + 2 * o.bps), (o.st.fc + o.st.df) / fs, amp + o.bps, o.bps, mai_free);
                }
                hipLaunchKernelGGL(k_rx_mai_out, dim3(2048), dim3(256), 0, s, sps, smp[p], mai_free);
                if (hipGetLastError() != hipSuccess) return fail(TWX_E_HIP, "interference cancellation launch failed");
                if (int rc = power(mai_free, &c.px)) return rc;                                            // rx.cpp:515-516; synchronises
                obs = mai_free;
            }
            char line[320];
            const char* chs = c.is_chA ? "A" : "B";
            if (!c.is_trk) {                                                                                // acquisition :521-586
                const long long nblk = (sps - c.nfft * cfg.dec_a) / c.nobs;
                const long long blk = cfg.acq_block >= 0 ? std::min<long long>(cfg.acq_block, nblk - 1) : (long long)(rng() % (unsigned long long)nblk);
                const long long idx = blk * c.nobs;                                                         // :529
                twx_acq_result a{};
                const int flags = TWX_ACQ_IZAMAX | (cfg.dec_a > 1 ? TWX_ACQ_DEC(cfg.dec_a) : 0);
                if (int rc = load_acq_operand(c)) return rc;
                if (int rc = lib(c.acq, twx_acquire_cdev(c.acq, obs + idx, c.fc_init, c.range, c.step, c.nobs / cfg.dec_a, flags, &a))) return rc;
                c.st.fc = a.fc; c.st.pt = a.pt;
                c.pk = 8.0 * a.pk * a.pk / c.psbb;                                                          // :570
                r.acq_idx = idx; r.n_trials = a.n_trials;
                if ((1.0 + c.snr_min) * c.pk > c.snr_min * c.px) {                                          // :573
                    c.st.pt = c.st.pt * cfg.dec_a;                                                          // :575
                    c.gd = (double)c.st.pt * 1.0e+9 / fs;
                    c.is_trk = true; c.is_first = true;
                    snprintf(line, sizeof line, "acquisition : Ch. %s, PRN#%2d, %3d %8.0lf %7.0lf %6d %8.3lf %8.3lf\n", chs, c.shown(), (int)(idx / 2 / c.nobs),
                             c.st.fc, c.gd, (int)c.st.pt, v2todBm(c.pk), v2todBm(c.px));                    // :582
                    log_line(line);
                    r.status = TWX_RX_ACQUIRED;
                } else r.status = TWX_RX_NO_SIGNAL;
            } else {                                                                                        // tracking :589-790
                twx_track_result t{};
                const bool in_buf = c.st.pt >= 0 && c.st.pt + c.nobs * (c.bps - 1) <= sps;                   // the program would read past its buffer
                if (in_buf) {
                    const twx_track_mai rec{c.mai_pk.data(), c.mai_amp.data(), c.mai_phase.data()};
                    if (int rc = lib(c.acq, twx_track_epoch_cdev_mai(c.acq, obs, sps, c.nobs, c.bps, c.nlag, c.replica_dev, 1.4142135624, &c.st, &t,
                                                                     real && any_sic && !c.is_sic ? &rec : nullptr))) return rc;
                }
                c.cnt = t.cnt;
                if (t.updated && real && any_sic && !c.is_sic) {                                            // rx.cpp:664-666,757: the records MAI_up reads
                    char* d = static_cast<char*>(c.mai_dev);
                    const size_t nb = (size_t)c.bps;
                    if (hipMemcpy(d, c.mai_amp.data(), nb * 8, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d + nb * 8, c.mai_phase.data(), nb * 8, hipMemcpyHostToDevice) != hipSuccess ||
                        hipMemcpy(d + nb * 16, c.mai_pk.data(), nb * 4, hipMemcpyHostToDevice) != hipSuccess) return fail(TWX_E_HIP, "H2D copy failed (interference records)");
                }
                if (t.updated) {
                    c.gd = t.gd; c.dg = t.dg; c.sdgd = t.sdgd; c.pk = t.pk;
                    if (!c.is_first) {                                                                      // :718-754
                        snprintf(r.dat_row, sizeof r.dat_row, "%14.6lf %11.8lf %3d %5.3lf %14.6lf %11.6lf %8.4lf %7.3lf %7.3lf\n", c.st.fc + c.st.df, c.st.phi,
                                 t.cnt, 0.0, c.gd, c.dg, c.sdgd, v2todBm(c.pk), v2todBm(c.px - c.pk));
                        if (!out_dir.empty()) {
                            FILE* f = fopen((out_dir + "/" + c.dat_name).c_str(), "a");
                            if (!f) return fail(TWX_E_ARG, "cannot append to " + out_dir + "/" + c.dat_name);
                            fputs(r.dat_row, f); fclose(f);
                        }
                        r.status = TWX_RX_TRACKED;
                    } else {
                        snprintf(line, sizeof line, "code lock   : Ch. %s, PRN#%2d, count = %d / %d\n", chs, c.shown(), t.cnt, c.bps);     // :761
                        log_line(line);
                        c.is_first = false;                                                                 // :766
                        r.status = TWX_RX_CODE_LOCK;
                    }
                } else {
                    snprintf(line, sizeof line, "%s : Ch. %s, PRN#%2d, count = %d / %d\n", c.is_first ? "acq failed " : "lock lost  ", chs, c.shown(), t.cnt, c.bps);   // :779,787
                    log_line(line);
                    r.status = c.is_first ? TWX_RX_ACQ_FAILED : TWX_RX_LOCK_LOST;
                    c.is_trk = false; c.st.last_phi = 0.0;                                                  // :792-793
                }
            }
            r.cnt = c.cnt; r.fc = c.st.fc; r.df = c.st.df; r.phi = c.st.phi; r.gd = c.gd; r.dg = c.dg; r.sdgd = c.sdgd;
            r.pk = c.pk; r.px = c.px; r.pt = c.st.pt;
        }
        return TWX_OK;
    }
};

template <class F> static int rx_guard(twx_rx* rx, F f) noexcept {
    // the launches of this library are checked with hipGetLastError(): an error another library left behind on this thread
    // (RCCL and PyTorch probe pointers and peers and do not clear what those probes set) must not be taken for ours
    (void)hipGetLastError();
    try {
        if (rx && rx->stream()) { twx::FenceShared fence(rx->dev, rx->stream()); return f(); }      // never beside a matrix-core FIR (twx_internal.h)
        return f();
    }
    catch (const std::bad_alloc&) { return rx ? rx->fail(TWX_E_NOMEM, "out of host memory") : TWX_E_NOMEM; }
    catch (const std::exception& e) { return rx ? rx->fail(TWX_E_STATE, std::string("internal error: ") + e.what()) : TWX_E_STATE; }
    catch (...) { return rx ? rx->fail(TWX_E_STATE, "internal error") : TWX_E_STATE; }
}

extern "C" {

const char* twx_rx_last_error(const twx_rx* rx) { return rx ? rx->err.c_str() : g_rx_create_err.c_str(); }

int twx_rx_parse_param(const char* path, twx_rx_row* rows, int32_t max_rows) {
    if (!path || !rows || max_rows < 1) return TWX_E_ARG;
    FILE* f = fopen(path, "r");
    if (!f) { g_rx_create_err = std::string("no such parameter file : ") + path; return TWX_E_ARG; }
    char str[200], str2[200];
    int n = 0;
    while (n < max_rows && fgets(str, sizeof str, f)) {
        if (str[0] == '#') continue;                                                      // :267
        strcpy(str2, str);
        int k = 0;
        for (char* p = strtok(str2, " ;\r\n"); p; p = strtok(nullptr, " ;\r\n")) ++k;    // :270-273
        if (!((str[0] == 'A' || str[0] == 'B') && (str[2] == 'N' || str[2] == 'S') && k == 9)) continue;   // :274
        char chs[64], ids[64];
        twx_rx_row r;
        memset(&r, 0, sizeof r);
        if (sscanf(str, "%63s %63s %d %lf %d %lf %lf %lf %lf", chs, ids, &r.pn, &r.fc_init, &r.kcps, &r.fltkhz, &r.frange, &r.fstep, &r.snr_min_db) != 9) continue;   // :280
        r.ch = str[0]; r.mode = str[2];
        if (!(r.pn >= 0 && r.pn <= 131 && r.kcps == 2500 && r.fc_init >= -200000. && r.fc_init < 200000. && r.frange >= 0.0 && r.frange < 200000. &&
              r.frange > r.fstep && r.snr_min_db > -100.0)) continue;                     // :288: rows that fail are skipped
        rows[n++] = r;
    }
    fclose(f);
    return n;
}

int twx_rx_create(const twx_rx_config* cfg, const twx_rx_row* rows, int32_t n_rows, twx_rx** out) {
    if (!cfg || !rows || !out) { g_rx_create_err = "null argument"; return TWX_E_ARG; }
    *out = nullptr;
    twx_rx* rx = new (std::nothrow) twx_rx();
    if (!rx) { g_rx_create_err = "out of host memory"; return TWX_E_NOMEM; }
    const int rc = rx_guard(rx, [&]() {
        rx->cfg = *cfg;
        rx->code_dir = cfg->code_dir && *cfg->code_dir ? cfg->code_dir : ".";
        rx->out_dir = cfg->out_dir ? cfg->out_dir : "";
        rx->cfg.code_dir = rx->cfg.out_dir = nullptr;
        return rx->init(rows, n_rows);
    });
    if (rc) { g_rx_create_err = rx->err; delete rx; return rc; }
    *out = rx;
    return TWX_OK;
}
void twx_rx_destroy(twx_rx* rx) { delete rx; }

int twx_rx_channel(const twx_rx* rx, int32_t i, twx_rx_channel_info* info) {
    if (!rx || !info || i < 0 || i >= (int)rx->ch.size()) return TWX_E_ARG;
    const Channel& c = rx->ch[(size_t)i];
    memset(info, 0, sizeof *info);
    info->pn = c.cid; info->is_chA = c.is_chA; info->clen = c.clen; info->nlag = c.nlag; info->bps = c.bps; info->is_sic = c.is_sic; info->nobs = c.nobs; info->nfft = c.nfft;
    info->duration = c.duration; info->range = c.range; info->step = c.step; info->snr_min = c.snr_min; info->psbb = c.psbb;
    snprintf(info->dat_name, sizeof info->dat_name, "%s", c.dat_name.c_str());
    return TWX_OK;
}
int twx_rx_powers(const twx_rx* rx, double pwr_v2[2]) {
    if (!rx || !pwr_v2) return TWX_E_ARG;
    pwr_v2[0] = rx->last_pwr[0]; pwr_v2[1] = rx->last_pwr[1];
    return TWX_OK;
}
int twx_rx_console_line(const twx_rx* rx, int32_t i, const twx_rx_report* r, char* buf, int32_t cap) {
    if (!rx || !r || !buf || cap < 2 || i < 0 || i >= (int)rx->ch.size()) return TWX_E_ARG;
    const Channel& c = rx->ch[(size_t)i];
    const char* chs = c.is_chA ? "A" : "B";
    const double mcps = (double)c.rc * 1.0e-6;
    int n;
    if (r->status == TWX_RX_NO_SIGNAL || r->status == TWX_RX_ACQ_FAILED || r->status == TWX_RX_LOCK_LOST) {                 // is_trk == 0 (:808-817)
        if (r->px <= r->pk) n = snprintf(buf, (size_t)cap, "%s: #%02d %4.1lf Mcps SNR       Low      , no signal\n", chs, c.shown(), mcps);
        else n = snprintf(buf, (size_t)cap, "%s: %lf %lf %s: #%02d %4.1lf Mcps SNR %6.2lf < %6.2lf, no signal\n", chs, r->pk, r->px, chs, c.shown(), mcps,
                          10.0 * log10(r->pk / (r->px - r->pk)), 10.0 * log10(c.snr_min));
    } else if (r->status == TWX_RX_ACQUIRED) {                                                                             // is_first == 1 (:820-824)
        n = snprintf(buf, (size_t)cap, "%s: %lf %lf %s: #%02d %4.1lf Mcps SNR %6.2lf > %6.2lf, analyzing\n", chs, r->pk, r->px, chs, c.shown(), mcps,
                     10.0 * log10(r->pk / (r->px - r->pk)), 10.0 * log10(c.snr_min));
    } else {                                                                                                               // :826-830 (ib = 0)
        n = snprintf(buf, (size_t)cap, "%s: #%02d %4.1lf Mcps %12.3lf Hz %13.3lf (%6.3lf) ns SNR %6.2lf dB\n", chs, c.shown(), mcps, r->fc + r->df, r->gd,
                     r->sdgd / sqrt((double)r->cnt), v2todBm(r->pk) - v2todBm(r->px - r->pk));
    }
    return n < 0 ? TWX_E_STATE : std::min(n, cap - 1);
}
const void* twx_rx_stream_dev(const twx_rx* rx, int32_t p) { return !rx ? nullptr : (p == 0 || p == 1) ? (const void*)rx->smp[p] : p == 2 ? (const void*)rx->mai_free : nullptr; }

int twx_rx_second_dev(twx_rx* rx, const void* iq_dev, twx_rx_report* reports) {
    if (!rx) return TWX_E_ARG;
    if (!iq_dev || !reports) return rx->fail(TWX_E_ARG, "bad argument");
    (void)hipSetDevice(rx->dev);
    return rx_guard(rx, [&]() { return rx->second(iq_dev, reports); });
}
int twx_rx_second(twx_rx* rx, const int16_t* iq, twx_rx_report* reports) {
    if (!rx) return TWX_E_ARG;
    if (!iq || !reports) return rx->fail(TWX_E_ARG, "bad argument");
    (void)hipSetDevice(rx->dev);
    return rx_guard(rx, [&]() {
        if (hipMemcpy(rx->iq_dev, iq, (size_t)rx->n_in * 8, hipMemcpyHostToDevice) != hipSuccess) return rx->fail(TWX_E_HIP, "H2D copy failed");
        return rx->second(rx->iq_dev, reports);
    });
}
int twx_rx_file(twx_rx* rx, const char* path, int64_t max_seconds, twx_rx_report* reports, int64_t report_seconds, int64_t* n_seconds) {
    if (!rx) return TWX_E_ARG;
    if (!path || !n_seconds || max_seconds < 0 || (reports && report_seconds < 0)) return rx->fail(TWX_E_ARG, "bad argument");
    *n_seconds = 0;
    (void)hipSetDevice(rx->dev);
    return rx_guard(rx, [&]() {
        struct File { FILE* f; ~File() { if (f) fclose(f); } } file{fopen(path, "rb")};
        if (!file.f) return rx->fail(TWX_E_ARG, std::string("Data filename error ") + path);                // :255
        std::vector<int16_t> buf((size_t)rx->n_in * 4);
        std::vector<twx_rx_report> tmp(rx->ch.size());
        // like the program (`do { fread ... } while (datares == sps*4/Ninterp)`, :468,832) but a short final read is not processed
        for (int64_t s = 0; s < max_seconds; ++s) {
            if (fread(buf.data(), 2, buf.size(), file.f) != buf.size()) break;
            twx_rx_report* dst = (reports && s < report_seconds) ? reports + s * (int64_t)rx->ch.size() : tmp.data();
            if (hipMemcpy(rx->iq_dev, buf.data(), buf.size() * 2, hipMemcpyHostToDevice) != hipSuccess) return rx->fail(TWX_E_HIP, "H2D copy failed");
            if (int rc = rx->second(rx->iq_dev, dst)) return rc;
            *n_seconds = s + 1;
        }
        return (int)TWX_OK;
    });
}

}  // extern "C"
