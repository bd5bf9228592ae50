// twx_filedf.hip — what the C++ program of the reference does around its per-window loop (processing/CPP/main.cpp), behind the C ABI:
//
//   twx_file_df      GoRanging::df (:363-450): ONE carrier estimate per capture FILE and channel — every N-th [I1 Q1 I2 Q2] frame of
//                    the whole file, mixed by foffset, minus the mean of the raw samples, squared, DFT of that ARBITRARY length,
//                    halves swapped, arg-max of |.| (channel 1 inside +-2*8 kHz of the decimated axis, channel 2 anywhere),
//                    freq(pos)/2 + foffset.
//   twx_write_cmat   GoRanging::save (:521-656): the `<capture>C.mat` container (MAT v5, uncompressed, n x 1 columns).
//
// The DFT of length L (any L: file_samples / N) is Bluestein's  X[k] = conj(w[k]) sum_n (v[n] conj(w[n])) w[k-n],  w[n] = exp(i pi n^2/L),
// evaluated ON THE DEVICE with the library's own fp64 two-pass transform of ONE fixed built-in length M (5 000 000 = 625 x 8000),
// whatever L is: the linear convolution is cut into blocks of B = M/2 inputs and B outputs; for output block j
//     Y_j = sum_i FFT(a_i) . FFT(h_{j-i}),   a_i = block i of v conj(w) (zero-padded),  h_d[m] = w[|d B - (B-1) + m|],
// summed in the frequency domain, one inverse transform per output block: nb + (2 nb - 1) + nb transforms of M points and nb^2
// multiply-adds — a 330-s capture (L = 6.6e7, nb = 27) takes about a hundred 5e6-point fp64 transforms, no plan plug-in, no length limit
// (cpp_twin.file_level_df needed one transform of >= 2L-1 points and refused captures over ~250 s).
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <complex>
#include <map>
#include <new>
#include <string>
#include <vector>
#include "twx_internal.h"

namespace {

typedef double2 cd;
thread_local std::string g_filedf_err;

__device__ __forceinline__ cd cmul(cd a, cd b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
// w[n] = exp(i pi n^2 / L): the angle reduced exactly in integers (n^2 mod 2L), then sincospi
__device__ __forceinline__ cd chirp(long long n, long long L) {
    if (n < 0) n = -n;
    const unsigned long long r = ((unsigned long long)n * (unsigned long long)n) % (unsigned long long)(2 * L);     // n < 2^31
    double s, c;
    sincospi((double)r / (double)L, &s, &c);
    return make_double2(c, s);
}

// v[n] = ((I + jQ) * exp(tlo foffset t_n) - mean)^2 * conj(w[n]) for n in [n0, n0 + B), zero elsewhere up to M  (:386-421)
__global__ void k_fdf_fill_a(const short* __restrict__ rec /*[L][4]*/, const double* __restrict__ t, int ch, long long L, long long n0, long long B,
                             long long M, double w_re, double w_im /*tlo*foffset*/, double mean_re, double mean_im, cd* __restrict__ a) {
    for (long long m = blockIdx.x * (long long)blockDim.x + threadIdx.x; m < M; m += (long long)gridDim.x * blockDim.x) {
        cd o = make_double2(0, 0);
        const long long n = n0 + m;
        if (m < B && n < L) {
            const double re = (double)rec[n * 4 + 2 * ch], im = (double)rec[n * 4 + 2 * ch + 1];
            // exp((w_re + j w_im) t): w_re = 0 for the program's tlo = -j 2 pi (kept general: std::exp of a complex)
            const double tt = t[n];
            double s, c;
            sincos(w_im * tt, &s, &c);
            const double e = w_re == 0.0 ? 1.0 : exp(w_re * tt);
            cd x = cmul(make_double2(re, im), make_double2(e * c, e * s));
            x.x -= mean_re; x.y -= mean_im;
            const cd v = cmul(x, x);
            const cd w = chirp(n, L);
            o = cmul(v, make_double2(w.x, -w.y));
        }
        a[m] = o;
    }
}
// h_d[m] = w[|d0 + m|] for m < 2B - 1 (the lags an input block and an output block can be apart), zero up to M
__global__ void k_fdf_fill_h(long long L, long long d0, long long B, long long M, cd* __restrict__ h) {
    for (long long m = blockIdx.x * (long long)blockDim.x + threadIdx.x; m < M; m += (long long)gridDim.x * blockDim.x) {
        cd o = make_double2(0, 0);
        const long long d = d0 + m;
        if (m < 2 * B - 1 && d > -L && d < L) o = chirp(d, L);
        h[m] = o;
    }
}
__global__ void k_fdf_mac(const cd* __restrict__ a, const cd* __restrict__ h, cd* __restrict__ y, long long M, int first) {
    for (long long m = blockIdx.x * (long long)blockDim.x + threadIdx.x; m < M; m += (long long)gridDim.x * blockDim.x) {
        const cd p = cmul(a[m], h[m]);
        if (first) y[m] = make_double2(p.x, -p.y);                 // conj: the inverse transform goes through the forward one
        else { cd q = y[m]; q.x += p.x; q.y -= p.y; y[m] = q; }
    }
}
// X[k0 + r] = conj(w[k]) * conj(c[r + B - 1]) / M  ->  |X|^2 into the swapped-halves position (:423-424), squared magnitude only
__global__ void k_fdf_out(const cd* __restrict__ c, long long L, long long k0, long long B, long long M, double* __restrict__ mag /*[L]*/) {
    const long long h = L / 2;
    for (long long r = blockIdx.x * (long long)blockDim.x + threadIdx.x; r < B; r += (long long)gridDim.x * blockDim.x) {
        const long long k = k0 + r;
        if (k >= L) return;
        const cd v = c[r + B - 1];
        const cd w = chirp(k, L);
        const cd x = cmul(make_double2(v.x / (double)M, -v.y / (double)M), make_double2(w.x, -w.y));
        // out[0..h) = fft[h..2h), out[h..2h) = fft[0..h); an odd L leaves out[L-1] = 0 and drops fft[L-1]
        long long pos = -1;
        if (k < h) pos = h + k; else if (k < 2 * h) pos = k - h;
        if (pos >= 0) mag[pos] = hypot(x.x, x.y);                   // std::abs of the complex value (:757-765)
    }
}
// first index of the largest value in [lo, hi)  (std::max_element, :757-765): two steps, ties to the lower index
struct Arg { double v; long long i; };
__global__ __launch_bounds__(256) void k_fdf_argmax(const double* __restrict__ mag, long long lo, long long hi, Arg* __restrict__ part) {
    __shared__ Arg sh[256];
    Arg b{-1.0, (long long)0x7fffffffffffffffll};
    const long long per = (hi - lo + gridDim.x - 1) / gridDim.x;
    const long long s = lo + (long long)blockIdx.x * per, e = min(hi, s + per);
    for (long long i = s + threadIdx.x; i < e; i += 256) { const double v = mag[i]; if (v > b.v || (v == b.v && i < b.i)) { b.v = v; b.i = i; } }
    sh[threadIdx.x] = b;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if (threadIdx.x < d) { const Arg o = sh[threadIdx.x + d]; Arg& m = sh[threadIdx.x]; if (o.v > m.v || (o.v == m.v && o.i < m.i)) m = o; }
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}

struct Dev {
    std::vector<void*> p;
    ~Dev() { for (void* q : p) if (q) (void)hipFree(q); }
    template <typename T> T* get(size_t count) { void* q = nullptr; if (hipMalloc(&q, std::max<size_t>(count * sizeof(T), 16)) != hipSuccess) return nullptr; p.push_back(q); return (T*)q; }
};
struct Map {
    void* p = MAP_FAILED; size_t len = 0; int fd = -1;
    ~Map() { if (p != MAP_FAILED) munmap(p, len); if (fd >= 0) close(fd); }
};

int fail(int code, const std::string& m) { g_filedf_err = m; return code; }

int file_df_impl(const char* path, double fs, int n_dec, int remote, double foffset, int device, long long block_override, double* df1, double* df2) {
    if (!path || !df1 || !(fs > 0) || n_dec < 1) return fail(TWX_E_ARG, "twx_file_df: bad argument");
    if (df2) *df2 = nan("");
    *df1 = nan("");
    Map mp;
    mp.fd = open(path, O_RDONLY);
    if (mp.fd < 0) return fail(TWX_E_ARG, std::string("cannot open ") + path);
    struct stat sb;
    if (fstat(mp.fd, &sb) != 0) return fail(TWX_E_ARG, std::string("cannot stat ") + path);
    const long long L = (long long)sb.st_size / (8ll * n_dec);                      // file_size :375: records of N frames
    if (L < 4) return fail(TWX_E_ARG, "capture shorter than four records of N samples");
    if (L >= (1ll << 31)) return fail(TWX_E_SIZE, "series too long");
    mp.len = (size_t)sb.st_size;
    mp.p = mmap(nullptr, mp.len, PROT_READ, MAP_PRIVATE, mp.fd, 0);
    if (mp.p == MAP_FAILED) return fail(TWX_E_ARG, std::string("cannot map ") + path);
    // every N-th frame (fread 4 shorts, fseek 4(N-1) :379-382) and the accumulated time base t += N/fs (:392)
    std::vector<short> rec((size_t)L * 4);
    std::vector<double> tt((size_t)L);
    long long sum[4] = {0, 0, 0, 0};
    {
        const short* src = static_cast<const short*>(mp.p);
        double t = 0.0;
        const double step = (double)n_dec / fs;
        for (long long i = 0; i < L; ++i) {
            const short* f = src + i * 4ll * n_dec;
            for (int c = 0; c < 4; ++c) { rec[(size_t)i * 4 + c] = f[c]; sum[c] += f[c]; }
            tt[(size_t)i] = t;
            t += step;
        }
    }
    if (device >= 0 && hipSetDevice(device) != hipSuccess) return fail(TWX_E_HIP, "hipSetDevice failed");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(TWX_E_HIP, "no HIP device available (the HIP path has no CPU fallback)");
    // the transform: one built-in fp64 length
    const long long M = 5000000;
    long long B = M / 2;
    if (block_override >= 2 && block_override <= M / 2) B = block_override;        // tests: many blocks on a short capture
    const long long nb = (L + B - 1) / B;
    twx_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.fs = 1.0; cfg.sps = 1; cfg.nint = 0; cfg.n_chips = M; cfg.lfsr_bitlen = 20; cfg.lfsr_taps = 9; cfg.precision = TWX_F64; cfg.nphase = 1;
    cfg.max_batch = 1; cfg.device = device; cfg.snr_rot = -1;
    twx_ctx* ctx = nullptr;
    if (int rc = twx_create(&cfg, &ctx)) return fail(rc, std::string("fp64 transform context: ") + twx_last_error(nullptr));
    struct CtxGuard { twx_ctx* c; ~CtxGuard() { if (c) twx_destroy(c); } } guard{ctx};
    hipStream_t st = (hipStream_t)twx_stream(ctx);
    Dev dv;
    short* rec_d = dv.get<short>((size_t)L * 4);
    double* t_d = dv.get<double>((size_t)L);
    double* mag = dv.get<double>((size_t)L);
    cd* tmp = dv.get<cd>((size_t)M);
    cd* yacc = dv.get<cd>((size_t)M);
    Arg* part = dv.get<Arg>(1024);
    std::vector<cd*> A((size_t)nb, nullptr);
    std::map<long long, cd*> H;
    if (!rec_d || !t_d || !mag || !tmp || !yacc || !part) return fail(TWX_E_NOMEM, "device allocation failed");
    for (long long i = 0; i < nb; ++i) if (!(A[(size_t)i] = dv.get<cd>((size_t)M))) return fail(TWX_E_NOMEM, "device allocation failed (input block spectra)");
    for (long long d = -(nb - 1); d <= nb - 1; ++d) if (!(H[d] = dv.get<cd>((size_t)M))) return fail(TWX_E_NOMEM, "device allocation failed (chirp spectra)");
    if (hipMemcpyAsync(rec_d, rec.data(), rec.size() * sizeof(short), hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync(t_d, tt.data(), tt.size() * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) return fail(TWX_E_HIP, "upload failed");
    auto lib = [&](int rc) { if (rc) g_filedf_err = twx_last_error(ctx); return rc; };
    const unsigned G = 2048, T = 256;
    // chirp spectra, one per block distance
    for (auto& kv : H) {
        hipLaunchKernelGGL(k_fdf_fill_h, dim3(G), dim3(T), 0, st, L, kv.first * B - (B - 1), B, M, tmp);
        if (hipGetLastError() != hipSuccess) return fail(TWX_E_HIP, "launch failed");
        if (int rc = lib(twx_fft_forward_dev(ctx, tmp, kv.second))) return rc;
    }
    // freq axis and the band of channel 1 (:399-406): last index satisfying each test
    const double f0 = -fs / 2 / n_dec, f1 = fs / 2 / n_dec, delta = (f1 - f0) / (double)(L - 1);
    auto freq = [&](long long i) { return i == L - 1 ? f1 : f0 + delta * (double)i; };          // linspace :734-757
    long long kmax = 0, kmin = 0;
    const double frange = 8000.0;
    for (long long i = 0; i < L; ++i) { if (freq(i) < 2 * frange) kmax = i; if (freq(i) <= -2 * frange) kmin = i; }
    const float foffset_f = (float)foffset;                                           // _foffset is a float member (:59)
    const double two = (double)2.0f;
    const double tw_im = -two * M_PI * foffset;                                        // tlo = -j 2 pi (:28), tlo * foffset (:372)
    for (int ch = 0; ch < (remote ? 1 : 2); ++ch) {
        const double mre = (double)sum[2 * ch] / (double)L, mim = (double)sum[2 * ch + 1] / (double)L;       // mean of the RAW samples (:384,416)
        for (long long i = 0; i < nb; ++i) {
            hipLaunchKernelGGL(k_fdf_fill_a, dim3(G), dim3(T), 0, st, rec_d, t_d, ch, L, i * B, B, M, 0.0, tw_im, mre, mim, tmp);
            if (hipGetLastError() != hipSuccess) return fail(TWX_E_HIP, "launch failed");
            if (int rc = lib(twx_fft_forward_dev(ctx, tmp, A[(size_t)i]))) return rc;
        }
        if (hipMemsetAsync(mag, 0, (size_t)L * sizeof(double), st) != hipSuccess) return fail(TWX_E_HIP, "memset failed");
        for (long long j = 0; j < nb; ++j) {
            for (long long i = 0; i < nb; ++i)
                hipLaunchKernelGGL(k_fdf_mac, dim3(G), dim3(T), 0, st, A[(size_t)i], H[j - i], yacc, M, i == 0 ? 1 : 0);
            if (hipGetLastError() != hipSuccess) return fail(TWX_E_HIP, "launch failed");
            if (int rc = lib(twx_fft_forward_dev(ctx, yacc, tmp))) return rc;          // tmp = FFT(conj(Y)) = conj(M * c)
            hipLaunchKernelGGL(k_fdf_out, dim3(G), dim3(T), 0, st, tmp, L, j * B, B, M, mag);
            if (hipGetLastError() != hipSuccess) return fail(TWX_E_HIP, "launch failed");
        }
        const long long lo = ch == 0 ? kmin : 0, hi = ch == 0 ? kmax : L;               // [kmin, kmax) for channel 1 (:427-430), everything for 2 (:443)
        if (hi <= lo) return fail(TWX_E_ARG, "the +-16 kHz band of the decimated axis is empty (capture too short)");
        hipLaunchKernelGGL(k_fdf_argmax, dim3(1024), dim3(256), 0, st, mag, lo, hi, part);
        std::vector<Arg> hp(1024);
        if (hipGetLastError() != hipSuccess || hipMemcpyAsync(hp.data(), part, hp.size() * sizeof(Arg), hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) return fail(TWX_E_HIP, "arg-max failed");
        Arg b{-1.0, (long long)0x7fffffffffffffffll};
        for (const Arg& a : hp) if (a.v > b.v || (a.v == b.v && a.i < b.i)) b = a;
        const double df = freq(b.i) / 2.0 + (double)foffset_f;
        if (ch == 0) *df1 = df; else if (df2) *df2 = df;
    }
    return TWX_OK;
}

// ---- MAT v5 (uncompressed): header + one miMATRIX per variable, n x 1 double or complex double -------------------------------
void put32(std::vector<unsigned char>& b, uint32_t v) { for (int i = 0; i < 4; ++i) b.push_back((unsigned char)(v >> (8 * i))); }
void pad8(std::vector<unsigned char>& b) { while (b.size() % 8) b.push_back(0); }
void put_doubles(std::vector<unsigned char>& b, const double* v, size_t n, size_t stride) {
    put32(b, 9 /*miDOUBLE*/); put32(b, (uint32_t)(n * 8));
    for (size_t i = 0; i < n; ++i) { unsigned char t[8]; memcpy(t, v + i * stride, 8); b.insert(b.end(), t, t + 8); }
    pad8(b);
}
void put_matrix(std::vector<unsigned char>& out, const char* name, const double* re, const double* im, size_t n, size_t stride) {
    std::vector<unsigned char> b;
    put32(b, 6 /*miUINT32*/); put32(b, 8); put32(b, 6 /*mxDOUBLE_CLASS*/ | (im ? 0x0800u : 0u)); put32(b, 0);       // array flags
    put32(b, 5 /*miINT32*/); put32(b, 8); put32(b, (uint32_t)n); put32(b, 1);                                      // n x 1
    const size_t nl = strlen(name);
    if (nl <= 4) { put32(b, (uint32_t)(1 /*miINT8*/ | (nl << 16))); unsigned char t[4] = {0, 0, 0, 0}; memcpy(t, name, nl); b.insert(b.end(), t, t + 4); }
    else { put32(b, 1); put32(b, (uint32_t)nl); b.insert(b.end(), name, name + nl); pad8(b); }
    put_doubles(b, re, n, stride);
    if (im) put_doubles(b, im, n, stride);
    put32(out, 14 /*miMATRIX*/); put32(out, (uint32_t)b.size());
    out.insert(out.end(), b.begin(), b.end());
}

}  // namespace

extern "C" {

const char* twx_file_df_last_error(void) { return g_filedf_err.c_str(); }

int twx_file_df(const char* path, double fs, int32_t n_dec, int32_t remote, double foffset, int32_t device, double* df1, double* df2) {
    (void)hipGetLastError();
    try {
        const char* e = getenv("TWX_FILEDF_BLOCK");                                   // tests: force several blocks on a short capture
        return file_df_impl(path, fs, n_dec, remote, foffset, device, e ? atoll(e) : 0, df1, df2);
    } catch (const std::bad_alloc&) { return fail(TWX_E_NOMEM, "out of host memory"); }
    catch (...) { return fail(TWX_E_STATE, "internal error"); }
}

/* GoRanging::save (:521-656): variables correction<c> (= indice0 + correction, :310), SNR<c> (10 log10(SNRr + SNRi), :355), df<c>,
 * puissance<c>, puissance<c>code (dB, :343), xval<c>, xval<c>m1, xval<c>p1 for channel c = 1 and — when res2 is given — 2.
 * Same variable set, order and values as results_io.save_cpp_mat. */
int twx_write_cmat(const char* path, const twx_result* res1, const twx_result* res2, int64_t n_windows) {
    if (!path || !res1 || n_windows < 0) return fail(TWX_E_ARG, "twx_write_cmat: bad argument");
    try {
        std::vector<unsigned char> out(128, ' ');
        const char* head = "MATLAB 5.0 MAT-file, written by libtwstft_hip (GoRanging::save layout)";
        memcpy(out.data(), head, strlen(head));
        memset(out.data() + 116, 0, 8);
        out[124] = 0x00; out[125] = 0x01; out[126] = 'I'; out[127] = 'M';             // version 0x0100, little endian
        const size_t n = (size_t)n_windows;
        std::vector<double> a(n), b(n);
        for (int c = 0; c < 2; ++c) {
            const twx_result* r = c ? res2 : res1;
            if (!r) continue;
            const std::string s = c ? "2" : "1";
            for (size_t i = 0; i < n; ++i) a[i] = (double)r[i].indice0 + r[i].correction;
            put_matrix(out, ("correction" + s).c_str(), a.data(), nullptr, n, 1);
            for (size_t i = 0; i < n; ++i) a[i] = 10 * log10(r[i].SNRr + r[i].SNRi);
            put_matrix(out, ("SNR" + s).c_str(), a.data(), nullptr, n, 1);
            for (size_t i = 0; i < n; ++i) a[i] = r[i].df;
            put_matrix(out, ("df" + s).c_str(), a.data(), nullptr, n, 1);
            for (size_t i = 0; i < n; ++i) a[i] = r[i].puissance;
            put_matrix(out, ("puissance" + s).c_str(), a.data(), nullptr, n, 1);
            for (size_t i = 0; i < n; ++i) a[i] = 10 * log10(r[i].puissancecode);
            put_matrix(out, ("puissance" + s + "code").c_str(), a.data(), nullptr, n, 1);
            const char* suffix[3] = {"", "m1", "p1"};
            for (int k = 0; k < 3; ++k) {
                for (size_t i = 0; i < n; ++i) {
                    const double* z = k == 0 ? r[i].xval : k == 1 ? r[i].xvalm1 : r[i].xvalp1;
                    a[i] = z[0]; b[i] = z[1];
                }
                put_matrix(out, ("xval" + s + suffix[k]).c_str(), a.data(), b.data(), n, 1);
            }
        }
        FILE* f = fopen(path, "wb");
        if (!f) return fail(TWX_E_ARG, std::string("cannot create ") + path);
        const bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
        if (fclose(f) != 0 || !ok) return fail(TWX_E_STATE, std::string("short write to ") + path);
        return TWX_OK;
    } catch (const std::bad_alloc&) { return fail(TWX_E_NOMEM, "out of host memory"); }
    catch (...) { return fail(TWX_E_STATE, "internal error"); }
}

}  // extern "C"
