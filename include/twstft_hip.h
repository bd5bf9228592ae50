/* twstft_hip.h — C ABI of libtwstft_hip.so, the MI355X (gfx950) implementation of the TWSTFT
 * correlation post-processing hot path of oscimp/amaranth_twstft.
 *
 * The reference has no FFI seam for this path: the kernel is an inline Octave function using
 * globals (processing/Octave/godual_ranging.m:3,12-49), restated in numpy
 * (experiments/221219_twoway/processing/godual_ranging.py:18-65) and C++
 * (processing/CPP/main.cpp:224-361 GoRanging::_process_method).  Each entry point below names
 * the reference interface it replaces.  Plain pointers and sizes only; the library never throws
 * across this boundary: every function returns 0 (TWX_OK) or a negative twx_status, and
 * twx_last_error() gives the text.  A context is NOT thread-safe: one context per
 * (host thread, GPU).  Host-side bindings: INTEGRATION.md (MEX / Octave / ctypes).
 */
#ifndef TWSTFT_HIP_H
#define TWSTFT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: twx_config.reserved became the live field nphase (the struct must be zero-initialised), the tracked-ranging,
 *    acquisition and *_dev entry points were added.  twx_abi_version() of an older library answers 1.
 * 3: twx_multi_* (several GPUs from one host process) and twx_rx_* (the DLL/PLL receiver) added; nothing changed.
 * 5: twx_multi_info grew (rccl_fallback, threads_pinned, numa_node, rccl_error: the RCCL exchange falls back to host-side
 *    concatenation instead of failing the job); twx_device_affinity / twx_pin_thread_to_device, twx_file_df / twx_write_cmat added.
 * 6: twx_multi_block / twx_multi_process_recording_dev / twx_multi_exchange_only (BASELINE.json configs[3] as written: ONE recording
 *    sharded over the contexts); the options TWX_OPT_FIR_MFMA / TWX_OPT_SELFCHECK and the twx_result.status bits; the velocity-compensated
 *    window (twx_set_resample; twx_result.reserved became the live field `dt`); the off-peak / squared-spectrum SNR estimators
 *    (twx_extra, twx_fetch_extra).  No struct changed its size. */
#define TWX_ABI_VERSION 6

typedef struct twx_ctx twx_ctx;

typedef enum twx_status {
    TWX_OK = 0,
    TWX_E_ARG = -1,        /* bad argument */
    TWX_E_SIZE = -2,       /* window length not factorable by the built plans */
    TWX_E_HIP = -3,        /* HIP runtime error (no GPU, launch failure, …) */
    TWX_E_NOMEM = -4,
    TWX_E_STATE = -5
} twx_status;

/* xcorr conventions (SURVEY.md §7 "convention zoo") */
enum { TWX_CONV_GODUAL = 0,   /* fft(y).*conj(fft(code))   godual_ranging.m:26,66            */
       TWX_CONV_CLAUDIO = 1 };/* fft(code).*conj(fft(y))   claudio_aligned_code_ranging_separate.m:59,124 */
enum { TWX_WIN_NONE = 0, TWX_WIN_HAMMING = 1 };   /* Hamming on fcode: processing/CPP/main.cpp:717-719 */
/* With the Hamming window the peak (indice, xval*, correction) comes from the correlation with the WINDOWED spectrum and the wipe-off
 * statistics (SNRr, SNRi, puissancecode, puissancenoise) from the unwindowed yint, as in the C++ program (main.cpp:288-301 against
 * :319-332): the context keeps both spectra and runs its row pass twice per batch. */
enum { TWX_F32 = 0, TWX_F64 = 1 };
enum { TWX_FLAG_PROFILE = 1,                      /* time every kernel launch with HIP events */
       TWX_FLAG_FINE_FREQ = 2,                    /* add the phase-drift fine carrier step of
                                                     experiments/221219_twoway/processing/godual_ranging.py:26-30
                                                     (needs N >= fs/3; off = processing/Octave/godual_ranging.m) */
       TWX_FLAG_CODE_ZERO_MEAN = 4 };             /* replica = code - mean(code): experiments/220616_Besancon/godual.m:7,
                                                     experiments/220822_qpsk_vs_bpsk/goqpsk.m:13 */
/* Replica chip levels: 2*c-1 (godual_ranging.m:65) or the raw 0/1 bytes (220616_Besancon/godual.m:5-7, goqpsk.m:5-12). */
enum { TWX_CODE_BIPOLAR = 0, TWX_CODE_UNIPOLAR = 1 };

/* Replaces the script constants / globals `fs Nint code fcode` (godual_ranging.m:3-5,62-66),
 * GoRanging's constructor arguments (processing/CPP/main.cpp:93-189). */
typedef struct twx_config {
    double fs;               /* sample rate, Hz (5e6) */
    int32_t sps;             /* samples per chip (2: `repelems` ×2, godual_ranging.m:64) */
    int32_t nint;            /* Nint; interpolation factor is 2*nint+1 (godual_ranging.m:5,27) */
    const uint8_t* chips;    /* host pointer, n_chips bytes 0/1 (code file contents), or NULL … */
    int64_t n_chips;
    int32_t lfsr_bitlen;     /* … to generate LFSR(bitlen,taps) from seed 1 on the device      */
    int32_t lfsr_taps;       /*   (amaranth_twstft/common.py:23-30,59-73)                       */
    int32_t convention;      /* TWX_CONV_* */
    int32_t window;          /* TWX_WIN_*  */
    int32_t precision;       /* TWX_F32 / TWX_F64 */
    int32_t var_ddof;        /* 0: numpy np.var; 1: Octave var (godual_ranging.m:44-48) */
    int32_t snr_rot;         /* rotate offset of the wipe-off relative to the 0-based peak: -1
                                (godual_ranging.m:43, godual_ranging.py(221219):59, main.cpp:332) */
    int32_t device;          /* HIP device ordinal, -1 = current */
    int32_t max_batch;       /* channel-windows per launch (0 = default) */
    int32_t flags;           /* TWX_FLAG_* */
    const uint8_t* chips_q;  /* NULL, or n_chips quadrature chips: complex replica chips + j*chips_q
                                (QPSK code of experiments/220822_qpsk_vs_bpsk/goqpsk.m:10-12).  With a complex,
                                unipolar or zero-mean replica the wipe-off statistics SNRr, SNRi, puissancecode and
                                puissancenoise are not defined (they rely on |code| = 1) and are returned as NaN. */
    int32_t code_levels;     /* TWX_CODE_* */
    int32_t nphase;          /* 0: 2*nint+1 output phases (godual_ranging.m:27); 1..5: that many, e.g. 2 for the x2 FFT-domain
                                interpolation of experiments/231001_DLL_PLL/rxcomplex.cpp:914-963 */
} twx_config;

/* Carrier search band `k` of processing(d,k) (godual_ranging.m:83-89): inclusive range of
 * 0-based indices into the fftshifted spectrum of d.^2 (Octave's k(1)-1 .. k(end)-1). */
typedef struct twx_band { int64_t k_lo, k_hi; } twx_band;

/* One channel-window result = the outputs of processing(d,k) (godual_ranging.m:12) plus the
 * complex peak samples the later scripts save (`xval*`, claudio…separate.m:207). */
typedef struct twx_result {
    int64_t indice0;         /* 0-based arg-max in the (2*nint+1)*N grid (Octave indice = +1) */
    double correction;       /* parabolic vertex offset, godual_ranging.m:33 */
    double xval[2], xvalm1[2], xvalp1[2];   /* prnmap(indice), (indice-1), (indice+1): re, im */
    double zwin[7][2];       /* prnmap(indice-3 … indice+3), circular */
    double df;               /* carrier offset used (Hz) */
    int64_t df_index;        /* 0-based fftshifted arg-max index of |fft(d.^2)|, -1 if df was supplied */
    double SNRr, SNRi, puissance, puissancecode, puissancenoise;   /* godual_ranging.m:44-48 */
    int32_t status;          /* 0, or TWX_STATUS_* bits */
    int32_t dt;              /* velocity-compensated window (twx_set_resample): the carried whole-sample offset the script adds to this
                                window's indice (`indice1(p)=indice1(p)+dt`, godual_ranging_OP_vitesse.m:68); 0 otherwise.  indice0 is
                                the arg-max itself. */
} twx_result;
/* twx_result.status bits.  TWX_STATUS_SELFCHECK: TWX_OPT_SELFCHECK was on and a row of this window's middle pass broke Parseval's
 * identity — the record is not to be trusted (re-run the window). */
enum { TWX_STATUS_SELFCHECK = 1,
       /* twx_set_resample: more than the first / last sample of the resampled window fell outside the window — Octave's interp1 leaves NaN
        * there, the map is NaN, max() answers index 1: the record holds indice0 = 0 and NaN values, as the script's workspace would. */
       TWX_STATUS_RESAMPLE_NAN = 2 };

typedef struct twx_info {
    int64_t n;               /* complex samples per channel-window = n_chips*sps */
    int32_t n1, n2;          /* N = n1*n2 (column pass × row pass) */
    int32_t nphase;          /* 2*nint+1 */
    int32_t batch;           /* channel-windows per launch */
    int32_t precision;
    int32_t col_w;           /* adjacent columns per column-pass workgroup */
    int64_t device_bytes;    /* device memory held by the context */
} twx_info;

const char* twx_strerror(int status);
const char* twx_last_error(const twx_ctx* ctx);      /* ctx may be NULL: last create() error */
int twx_abi_version(void);

/* Transform lengths.  A window of N = n_chips*sps samples is transformed as N1 x N2 (column pass x row pass); the
 * library is built with the pairs the reference's code lengths need and takes further lengths
 * N = 2^a 3^b 5^c 7^d from plan plug-ins: shared objects compiled from the same kernel sources for one more length
 * (`python -m amaranth_twstft_amd.plans N`, needs hipcc), loaded explicitly with twx_load_plan() or found by
 * twx_create() in the directory `plans/` beside the library (TWX_PLAN_DIR overrides).  The reference reads any code
 * file (godual_ranging.m:62-66); twx_create() answers TWX_E_SIZE only when no plan pair exists for the length.
 * twx_plan_lengths: kind 0 = column plans (lengths[i], tile widths[i]), 1 = row plans; returns the count. */
const char* twx_plan_source_hash(void);   /* plug-ins named *_<this>.so match the library's kernel sources */
int twx_load_plan(const char* path);
int twx_plan_available(int64_t n, int32_t precision);
int twx_plan_lengths(int32_t kind, int32_t precision, int32_t* lengths, int32_t* widths, int32_t max_entries);

/* Build plans/twiddles, upload or generate the code and compute conj(fft(code)) once
 * (godual_ranging.m:62-66; GoRanging::fill_fcode main.cpp:658-732). */
int twx_create(const twx_config* cfg, twx_ctx** out);
void twx_destroy(twx_ctx* ctx);
int twx_get_info(const twx_ctx* ctx, twx_info* info);

/* processing(d,k) over n_windows consecutive windows of a raw capture held in HOST memory.
 * iq: little-endian int16, n_channels interleaved IQ pairs per sample ([I Q] or [I1 Q1 I2 Q2],
 * godual_ranging.m:76-79), window w occupying samples [w*N, (w+1)*N).  `channel` selects the
 * pair.  band != NULL: estimate df per window in that band (godual_ranging.m:14-15);
 * band == NULL: df[w] is used (processing(d,df), claudio…separate.m:49).  The window mean is
 * removed (godual_ranging.m:80).  out: n_windows results (host). */
int twx_process_windows(twx_ctx* ctx, const int16_t* iq, int64_t n_windows, int32_t n_channels,
                        int32_t channel, const twx_band* band, const double* df, twx_result* out);

/* The reference's own signature: processing(d,k) (godual_ranging.m:12) and processing(d,df)
 * (claudio_aligned_code_ranging_separate.m:49) on the complex DOUBLE column `d` the scripts build — n_windows
 * consecutive windows of N samples in HOST memory, already mean-removed by the caller (godual_ranging.m:80,94; the
 * library does not touch the mean here).  Sample n is d_re[n*stride] + j*d_im[n*stride]: stride 1 = separate real and
 * imaginary arrays (mxGetPr/mxGetPi, Octave), stride 2 with d_im == d_re+1 = interleaved (mxGetComplexDoubles, numpy
 * complex128).  band != NULL: coarse df per window; else df[w].  The context's convention selects the
 * godual (fft(y).*conj(fft(code))) or claudio (fft(code).*conj(fft(y))) result. */
int twx_process_complex(twx_ctx* ctx, const double* d_re, const double* d_im, int64_t stride, int64_t n_windows,
                        const twx_band* band, const double* df, twx_result* out);

/* channel = TWX_ALL_CHANNELS in twx_process_windows, twx_process_windows_dev and twx_process_file processes every
 * channel of every window from ONE copy of the capture (godual_ranging.m:91,95 calls processing() on both channels of
 * each window): results out[w*n_channels + c], and df — where given — df[w*n_channels + c].  n_channels <= 4. */
#define TWX_ALL_CHANNELS (-1)

/* Same with the capture already resident in DEVICE memory (iq_dev) and results written to
 * DEVICE memory (out_dev, n_windows records).  Asynchronous on the context's stream:
 * call twx_synchronize() before reading.  df (host pointer) is copied at enqueue time. */
int twx_process_windows_dev(twx_ctx* ctx, const void* iq_dev, int64_t n_windows, int32_t n_channels,
                            int32_t channel, const twx_band* band, const double* df, twx_result* out_dev);
int twx_synchronize(twx_ctx* ctx);

/* The velocity-compensated window of experiments/220706_TWSTFT/godual_ranging_OP_vitesse.m (:4 `vitesse=-3.25e-9`): after the NCO mix the
 * window is resampled linearly on a stretched time axis, `yi=interp1([0:N-1],y,[0:N-1]*1/(1-vitesse)+t0)` (:40), with the offset carried
 * from window to window, `t0=t0+length(y)*vitesse` (:41), the edge rule `if isnan(yi(end)) yi(end)=yi(end-1)`, `if isnan(yi(1))
 * yi(1)=yi(2)` (:42-43), and `t0` wrapped into (-1, 1) with the whole-sample count `dt` following it (:70-71).  twx_set_resample switches
 * it on for every later twx_process_windows[_dev] / twx_process_file call of the context on ONE channel (vitesse = 0: off) and sets the
 * carried state (the script starts at t0 = 0, dt = 0); windows are taken in call order, the state advances by one window each — also
 * across calls — and twx_get_resample returns it as it stands BEFORE the next window.  Each record carries its window's `dt`
 * (twx_result.dt); the script's `indice1(p)` is indice0 + 1 + dt.  The interpolation runs inside the column pass that loads the samples
 * (two neighbouring int16 frames per output, gathered from the same cache lines): no extra pass over the window.  |N * vitesse| < 1.
 * The script correlates with the zero-mean 0/1 replica, no interpolation: twx_config {nint = 0, code_levels = TWX_CODE_UNIPOLAR,
 * flags = TWX_FLAG_CODE_ZERO_MEAN}; its band is (96 200, 106 200) Hz of the linspace axis (:32). */
int twx_set_resample(twx_ctx* ctx, double vitesse, double t0, int64_t dt);
int twx_get_resample(twx_ctx* ctx, double* vitesse, double* t0, int64_t* dt);

/* Run-time options.  TWX_OPT_REMOVE_MEAN (default 1): subtract the window's complex mean before the
 * NCO (d=d-mean(d), godual_ranging.m:80,94, done by the caller of processing() in the reference);
 * 0 leaves the samples as they are (search_df mixes the raw chunk,
 * acquisition/claudio_aligned_code_ranging_separate.m:34). */
enum { TWX_OPT_REMOVE_MEAN = 1,
       /* Diagnostics (tools/kernel_alone.py): launch only kernel class `value` of the chain (order of twx_profile_get's
        * names: 0 k_sums, 1 k_col_fwd_square, 2 k_row_band, 3 k_df_tables, 4 k_col_fwd_mix, 5 k_row_mid, 6 k_col_inv,
        * 7 k_peak; -1 = the whole chain again), TWX_OPT_DEBUG_REPEAT times per batch, on whatever the batch buffers hold
        * from the last complete call.  Results of such a call are meaningless. */
       TWX_OPT_DEBUG_ONLY = 100, TWX_OPT_DEBUG_REPEAT = 101,
       /* Diagnostic of TWX_OPT_SELFCHECK (ignored while that option is off): damage ONE value of one row of the middle pass of every batch —
        * value = 2 * (row + 1) + which, row = k1 * batch_windows + window-in-batch; which 0: between the stages of the forward row
        * transform, 1: between the stages of the last phase's inverse transform; 0 = none.  The window must come back flagged. */
       TWX_OPT_DEBUG_FAULT = 102,
       /* The matrix-core form of the FIR front end (k_fir_mfma) for twx_fir_decimate_dev on THIS context: 1 = use it, 0 = never,
        * -1 (default) = follow the environment variable TWX_FIR_MFMA.  That kernel is 20 % faster than the vector form and makes
        * packed-fp32 arithmetic of waves resident beside it go wrong (profiles/r05_fir_mfma.txt), so the library never lets it share the
        * device with other work of the process: every such launch is ordered behind everything this library has enqueued on the device
        * so far, on any context or stream, and everything enqueued later waits for it (csrc/twx_internal.h).  Work of OTHER processes on
        * the same GPU is out of the library's reach: leave the option off unless the process owns the GPU. */
       TWX_OPT_FIR_MFMA = 2,
       /* Run-time self-check of the fused middle pass (FFT pass 2, x conj(FFT(code)), the R inverse row transforms = what
        * godual_ranging.m:25-28 asks of that pass): per row, sum |input row|^2 * N2 against sum |spectrum row|^2 and sum |product row|^2 * N2
        * against sum |output row|^2 of every phase (Parseval; every twiddle and ramp has modulus one), from values the pass holds in
        * registers.  A row outside the tolerance sets TWX_STATUS_SELFCHECK in its window's record.  value: 0 = off (default), 1 = on at
        * 1e-5 relative, > 1 = the tolerance in units of 1e-9.  The detector for silent faults of the fault class of
        * profiles/r05_fir_mfma.txt (whole rows of this pass wrong while another kernel is resident beside it).  Rows of 4000 / 8000 points
        * (three-stage plans); TWX_E_ARG elsewhere.  The environment variable TWX_SELFCHECK=<value> is the default of every context created
        * afterwards whose pass has the form.  twx_selfcheck_stats: the largest relative deviation seen and the number of rows
        * flagged since the last reset (synchronises the context). */
       TWX_OPT_SELFCHECK = 3 };
int twx_selfcheck_stats(twx_ctx* ctx, double* max_rel_dev, int64_t* rows_flagged, int32_t reset);

/* The other SNR estimators the reference compares (experiments/220830_OP/process_OP.m:94-97,119-121,138; the three-way comparison of
 * experiments/221127_SNR/simu_snr.m and its README) as optional outputs beside the wipe-off SNR of twx_result:
 *   bruit          var(prnmap(indice+20:indice+20+L-1)), the off-peak variance of the correlation map (Octave var: N-1, complex
 *                  deviations by modulus), L = TWX_OPT_BRUIT_LEN (the script: 1001 for bruit1, 10001 for bruit2); NaN where the range
 *                  leaves the map (the guard `(indice1(p)+1020)<length(prnmap01)` of :119)
 *   valmax_square  max(d22(freqindex)), d22 = fftshift(abs(fft(d1.^2))): the carrier peak of the squared signal (:95)
 *   noise_square   var(d22(tmpdf+20:tmpdf+20+L-1)), L = TWX_OPT_NOISE_SQUARE_LEN (the script: 10001, :97); NaN where the range leaves
 *                  the spectrum.  Both NaN when the carrier was supplied (no squared spectrum is formed).
 * Either option > 0 switches the outputs on for the context's later twx_process_windows[_dev] / twx_process_file
 * calls (0, the default: off; the calls then cost nothing extra); twx_fetch_extra copies the estimators of the LAST call, record for
 * record (extra[i] belongs to out[i]), to the host — it synchronises the context.  With the options on a call first waits for the
 * context's earlier work.  Rows up to 10 240 points (TWX_E_ARG beyond). */
typedef struct twx_extra { double bruit, valmax_square, noise_square, reserved; } twx_extra;
enum { TWX_OPT_BRUIT_LEN = 4, TWX_OPT_NOISE_SQUARE_LEN = 5 };
int twx_fetch_extra(twx_ctx* ctx, twx_extra* out_host, int64_t n_records);
int twx_set_option(twx_ctx* ctx, int32_t option, int64_t value);
void* twx_stream(twx_ctx* ctx);                      /* hipStream_t of the context */

/* File-in / results-out: the window loop of godual_ranging.m:70-103 over a capture FILE (raw int16,
 * n_channels interleaved IQ pairs per sample; skip_samples complex samples are skipped first, cf. the
 * 30-s fseek of claudio_aligned_code_ranging_separate.m:128).  Reads through pinned host buffers, one per
 * pipeline slot, so file I/O, PCIe copies and kernels overlap (the reader/worker overlap of
 * processing/CPP/main.cpp:488-497).  band != NULL: per-window coarse df; else df_const is used for every
 * window (GoRanging's per-file foffset).  A short final window ends the loop (godual_ranging.m:81,102).
 * out: capacity max_windows; *n_done = windows processed. */
int twx_process_file(twx_ctx* ctx, const char* path, int32_t n_channels, int32_t channel, int64_t skip_samples,
                     const twx_band* band, double df_const, twx_result* out, int64_t max_windows, int64_t* n_done);

/* Test / inspection entry points -------------------------------------------------------- */
/* Forward FFT of n complex doubles (host, interleaved re/im) through the context's two-pass
 * transform; out in natural order.  (Parity check of the transform vs np.fft.fft.) */
int twx_fft_forward(twx_ctx* ctx, const double* in, double* out);
/* conj(fft(code)) (× window) as held by the context, natural order, interleaved re/im doubles. */
int twx_get_code_spectrum(twx_ctx* ctx, double* out);
/* Full interpolated correlation map prnmap (godual_ranging.m:28) of ONE window: (2*nint+1)*N
 * complex doubles (host).  Slow path for tests.  Always the godual form ifft(pad(fft(y).*fcode)), also in a TWX_CONV_CLAUDIO
 * context, whose own map is the mirrored conjugate prnmap_c[m] = conj(prnmap[(M - m) mod M]) (the records carry that index
 * and those samples; magnitudes, maxima and variances of the two maps are the same). */
int twx_xcorr_map(twx_ctx* ctx, const int16_t* iq, int32_t n_channels, int32_t channel, double df,
                  double* out);

/* Replica given by its SPECTRUM: replaces what the context multiplies FFT(y) with (conj(fft(code)) by default) by
 * spec[k], k = 0..N-1 in natural FFT order, complex doubles (re, im) in HOST memory.  This is how the acquisition stage
 * of experiments/231001_DLL_PLL/rxcomplex.cpp gets its operand: conj(FFT(zero-padded sampled code)) (:416-437) times the
 * pass-band mask and 1/n^2 of cross_spectrum (:1001-1018) — and how the x2 interpolation of short2double (:914-963)
 * becomes a 2-phase "correlation" with a weight vector.  The wipe-off statistics are undefined afterwards (NaN). */
int twx_set_code_spectrum(twx_ctx* ctx, const double* spec);
/* The same with spec in DEVICE memory (N complex doubles, natural order), and the context's forward transform device to
 * device: N complex doubles in, N complex doubles out in natural order (in_dev == out_dev allowed; asynchronous on the
 * context's stream).  The replica set-up of the DLL/PLL receiver (twx_rx_*, rxcomplex.cpp:414-437) is made of these. */
int twx_set_code_spectrum_dev(twx_ctx* ctx, const void* spec_dev);
int twx_fft_forward_dev(twx_ctx* ctx, const void* in_dev, void* out_dev);
/* prnmap of ONE window of a DEVICE-resident int16 capture into DEVICE memory: nphase*N complex floats (re, im),
 * normalised like ifft (1/(nphase*N)), natural order.  No mean removal when TWX_OPT_REMOVE_MEAN is 0. */
int twx_xcorr_map_dev(twx_ctx* ctx, const void* iq_dev, int32_t n_channels, int32_t channel, double df, void* out_dev);
/* processing(d,df) for n_freqs trial offsets on ONE window of complex FLOAT samples resident in DEVICE memory
 * (d_dev: N x (re, im) float32; no mean removal): the per-bin body of the acquisition sweep rxcomplex.cpp:534-563 —
 * downconv_acq, FFT, cross_spectrum (through the context's replica spectrum), IFFT, arg-max.  flags & 1: the arg-max is
 * cblas_izamax's, i.e. of |re|+|im| (rxcomplex.cpp:553), not of the modulus.  out: n_freqs results (host). */
#define TWX_ACQ_IZAMAX 1
/* flags | TWX_ACQ_DEC(d): the window is every d-th sample of the stream at d_dev (downconv_acq's smp[i*dec] with dec = dec_a,
 * rxcomplex.cpp:543,1039-1049; the B210 build runs dec_a = 2, :228-230); d_dev then holds at least N*d samples. */
#define TWX_ACQ_DEC(d) (((d) & 0xff) << 8)
int twx_caf_freqs_cdev(twx_ctx* ctx, const void* d_dev, const double* freqs, int64_t n_freqs, int32_t flags, twx_result* out);
/* The acquisition sweep itself, rxcomplex.cpp:534-567, as ONE call: trial carriers fc_init-frange .. fc_init+frange in
 * fstep (`for (fcc = flow; fcc <= fhigh; fcc += fstep)`), keep the strictly highest peak (:556-562), then halve the step
 * with range = step until it drops under 1 Hz (:565-567).  The bookkeeping between rounds runs on the device; the host
 * synchronises once.  out: fc (Hz), pk = |z| at the arg-max, pt = its 0-based lag modulo pt_modulus (`% (nobs/dec_a)`,
 * :561; 0 = no modulus), n_trials = carriers evaluated.  flags: TWX_ACQ_IZAMAX, TWX_ACQ_DEC(d). */
typedef struct twx_acq_result { double fc, pk; int64_t pt, n_trials; } twx_acq_result;
int twx_acquire_cdev(twx_ctx* ctx, const void* d_dev, double fc_init, double frange, double fstep, int64_t pt_modulus, int32_t flags,
                     twx_acq_result* out);

/* Delay x Doppler cross-ambiguity of ONE window (host int16 capture, as twx_process_windows) ------
 * Replaces the acquisition sweep of experiments/231001_DLL_PLL/rxcomplex.cpp:534-563 (per trial
 * offset: down-convert → FFT → cross_spectrum → IFFT → izamax).
 * twx_caf_bins: integer-bin Doppler grid f = kappa*fs/N, kappa = k_lo..k_hi inclusive (fs/N = 1 Hz
 *   for the 1-s window); one forward FFT, then per bin a circular spectrum shift, the code-spectrum
 *   product and an inverse FFT (no interpolation).  pk[i] = max |xcorr| (same normalisation as
 *   prnmap, 1/N), lag[i] = its 0-based lag, i = kappa-k_lo.
 * twx_caf_freqs: arbitrary trial offsets (Hz): full processing(d,df) of the same window per offset
 *   (context's nint applies); out = n_freqs results. */
int twx_caf_bins(twx_ctx* ctx, const int16_t* iq, int32_t n_channels, int32_t channel, int64_t k_lo, int64_t k_hi,
                 double* pk, int64_t* lag);
int twx_caf_freqs(twx_ctx* ctx, const int16_t* iq, int32_t n_channels, int32_t channel, const double* freqs,
                  int64_t n_freqs, twx_result* out);
/* twx_caf_bins with the window already in DEVICE memory (iq_dev); pk / lag are host arrays (16 bytes per bin). */
int twx_caf_bins_dev(twx_ctx* ctx, const void* iq_dev, int32_t n_channels, int32_t channel, int64_t k_lo, int64_t k_hi,
                     double* pk, int64_t* lag);

/* Long squared spectra for carrier acquisition ----------------------------------------------------
 * Replaces d2=fftshift(abs(fft(d.^2))) over a whole ls-second chunk in
 * acquisition/claudio_aligned_code_ranging_separate.m:30 (search_df) and :162-163 (per-chunk carrier
 * update on kbon-3..kbon+3).  iq_dev is a DEVICE pointer to n_samples interleaved int16 IQ samples
 * (no mean removal, as in the script); bins are signed DFT bin numbers of the n_samples-point
 * transform (shifted index i of the script = bin + floor(n_samples/2)); outputs are HOST arrays.
 * twx_sqspec_bins_dev: complex fft(d.^2)[bin] for up to 64 bins by direct summation (any n_samples).
 * twx_sqspec_band_dev: abs(fft(d.^2)) for n_bins consecutive bins starting at k_lo; n_samples must
 *   be a multiple M of the context's window length N (M decimated N-point transforms + recombination). */
int twx_sqspec_bins_dev(twx_ctx* ctx, const void* iq_dev, int64_t n_samples, int32_t n_channels, int32_t channel,
                        const int64_t* bins, int32_t n_bins, double* out_re_im);
int twx_sqspec_band_dev(twx_ctx* ctx, const void* iq_dev, int64_t n_samples, int32_t n_channels, int32_t channel,
                        int64_t k_lo, int64_t n_bins, double* out_mag);

/* Direct sliding dot-product correlator for short codes (tracking stage) -----------------------
 * Replaces downconv_trk + cblas_dgemm(W^T X)/nobs + the PRN_mapping replica matrix of
 * experiments/231001_DLL_PLL/rxcomplex.cpp:593-605,989-999,1051-1061.  For code period p < ncodes
 * and lag index li (lag = li - nlag, nlag <= 31):
 *   out[p][li] = (scale/nobs) * sum_i x[pt+p*nobs+i] * exp(-2 pi j (ff*(p*nobs+i)+phi)) * replica[(i-lag) mod nobs]
 * iq: host int16 capture of n_samples samples; ff in cycles/sample, phi in cycles; out: ncodes*(2*nlag+1)
 * complex doubles (re, im).  cor = re^2+im^2 and phi = atan2(im,re)/2pi (get_cor_and_phi, :1063) are host one-liners. */
int twx_sliding_dot(const int16_t* iq, int64_t n_samples, int32_t n_channels, int32_t channel, int64_t pt, int64_t nobs,
                    int32_t ncodes, int32_t nlag, const float* replica, double ff, double phi, double scale, double* out);

/* One tracking epoch of experiments/231001_DLL_PLL/rxcomplex.cpp:593-745 ------------------------------------------
 * twx_track_update: the arithmetic on the (bps-1) x (2*nlag+1) power / phase matrices of get_cor_and_phi (:1063-1072,
 *   row-major, HOST arrays): per code period cblas_idamax and the high-resolution-correlator delay (:630-661), the
 *   3-sigma filter on median / inter-quartile range (:689-700), the BPSK half-cycle phase unwrap against last_phi
 *   (:703-716), gsl_fit_wlinear of phase -> carrier update and of delay -> code-phase update (:728-745).  `state` holds
 *   the channel_info fields the epoch reads and writes; out->updated = 0 when no more than half of the periods had a
 *   usable peak (:667; state untouched).  No GPU needed.
 * twx_track_epoch_dev: the whole epoch on a DEVICE-resident int16 capture: phi = fmod(pt*fc/fs, 1) (:594), downconv_trk +
 *   cblas_dgemm against the lagged replicas (:599-605) as twx_sliding_dot_dev with pt = state->pt, ff = fc/fs, then
 *   get_cor_and_phi and twx_track_update.  One host synchronisation per epoch. */
typedef struct twx_track_state {
    double fs, duration, psbb;      /* sample rate, code period in s (ci.duration), reference power (ci.psbb) */
    double fc, df, phi, last_phi;   /* carrier (Hz, integer part), its fractional part, phase (cycles), unwrap reference */
    int64_t pt;                     /* code phase in samples */
    double fc_prev; int64_t pt_prev;
} twx_track_state;
typedef struct twx_track_result {
    double freq, phi, gd, dg, sdgd, pk;   /* fc+df (Hz), phase (cycles), delay (ns), its slope (ns/epoch), sqrt(chisq/cnt), mean signal power */
    int32_t cnt, updated;
} twx_track_result;
int twx_track_update(const double* cor, const double* phi, int32_t bps, int32_t nlag, twx_track_state* state, twx_track_result* out);
int twx_track_epoch_dev(twx_ctx* ctx, const void* iq_dev, int64_t n_samples, int32_t n_channels, int32_t channel, int64_t nobs,
                        int32_t bps, int32_t nlag, const float* replica_dev, double scale, twx_track_state* state,
                        twx_track_result* out);

/* FIR low-pass + decimation front end (BASELINE.json configs[4], 70 Msps → 5 Msps): y[m] = sum_j taps[j]*x[m*dec+j],
 * "valid" part only: *n_out = (n_in-ntaps)/dec+1.  ntaps <= 1024, dec <= 16.  out_i16 (interleaved IQ, rounded
 * half-to-even, saturated) and/or out_f32 (interleaved re,im) may be NULL.  No twin in the reference. */
int twx_fir_decimate(const int16_t* iq, int64_t n_in, int32_t n_channels, int32_t channel, const float* taps, int32_t ntaps,
                     int32_t dec, int16_t* out_i16, float* out_f32, int64_t* n_out);

/* The same two kernels on DEVICE-resident data, asynchronous on the context's stream (twx_stream(ctx)), work buffers
 * owned by the context (no allocation per call once they have their size).  iq_dev, replica_dev, out*_dev are device
 * pointers; taps is a HOST array (copied at enqueue time); *n_out is written before the call returns.  The int16
 * output of twx_fir_decimate_dev is a 1-channel [I Q] capture that twx_process_windows_dev consumes directly, so the
 * 70 Msps -> 5 Msps -> correlator chain of BASELINE.json configs[4] never leaves the device.  out_i16_dev must be
 * 16-byte aligned, out_f32_dev 32-byte aligned (any hipMalloc pointer is). */
int twx_sliding_dot_dev(twx_ctx* ctx, const void* iq_dev, int64_t n_samples, int32_t n_channels, int32_t channel, int64_t pt,
                        int64_t nobs, int32_t ncodes, int32_t nlag, const float* replica_dev, double ff, double phi, double scale,
                        double* out_dev);
int twx_fir_decimate_dev(twx_ctx* ctx, const void* iq_dev, int64_t n_in, int32_t n_channels, int32_t channel, const float* taps,
                         int32_t ntaps, int32_t dec, void* out_i16_dev, void* out_f32_dev, int64_t* n_out);

/* twx_sliding_dot_dev / twx_track_epoch_dev on complex FLOAT samples resident in DEVICE memory (smp_dev: n_samples x (re, im)
 * float32) — the x2-interpolated stream ci.dev_smp that rxcomplex.cpp tracks on (:477,602). */
int twx_sliding_dot_cdev(twx_ctx* ctx, const void* smp_dev, int64_t n_samples, int64_t pt, int64_t nobs, int32_t ncodes, int32_t nlag,
                         const float* replica_dev, double ff, double phi, double scale, double* out_dev);
int twx_track_epoch_cdev(twx_ctx* ctx, const void* smp_dev, int64_t n_samples, int64_t nobs, int32_t bps, int32_t nlag,
                         const float* replica_dev, double scale, twx_track_state* state, twx_track_result* out);

/* The per-period records the real-sample program keeps for its successive interference cancellation
 * (experiments/231001_DLL_PLL/rx.cpp:664-666,752-757: dev_pk_idx, dev_res_amp, dev_raw_phi; consumed by MAI_up :1011-1020).
 * Arrays of bps entries each, owned by the caller; written only when the epoch updates the state (out->updated = 1):
 *   pk_idx[p]  lag of the period's peak, -nlag..nlag (:638); 0 where the period had no usable peak
 *   amp[p]     sqrt(2 cor)/psbb (:640)
 *   phase[p]   carrier phase at the peak BEFORE the BPSK adjustment (raw_phi :664) minus
 *              (fc + df - fc_prev) * ((p+1)*nobs + pt_prev) / fs with the UPDATED fc, df (:754); entry bps-1 is 0.
 * The _mai variants are twx_track_update / twx_track_epoch_cdev with these arrays filled in (mai may be NULL). */
typedef struct twx_track_mai { int32_t* pk_idx; double* amp; double* phase; } twx_track_mai;
int twx_track_update_mai(const double* cor, const double* phi, int32_t bps, int32_t nlag, int64_t nobs, twx_track_state* state,
                         twx_track_result* out, const twx_track_mai* mai);
int twx_track_epoch_cdev_mai(twx_ctx* ctx, const void* smp_dev, int64_t n_samples, int64_t nobs, int32_t bps, int32_t nlag,
                             const float* replica_dev, double scale, twx_track_state* state, twx_track_result* out,
                             const twx_track_mai* mai);

/* The DLL/PLL receiver as a program: sdr.param in, ch?.pn??.????kcps.dat rows out -------------------------------------
 * cfg.ninterp selects which of the two programs of experiments/231001_DLL_PLL runs:
 *   2  rxcomplex.cpp — complex samples, x2 interpolation, 'N' rows only (described below);
 *   1  rx.cpp — the REAL-sample program: no interpolation (short2double :892-900 keeps the I sample of each physical channel,
 *      sps = fs_in), the same set-up / acquisition / tracking on those samples (downconv_acq :976-986, downconv_trk :988-998),
 *      log lines in rxreal.log, and successive interference cancellation for 'S' rows (:505-518): before such a row is
 *      searched or tracked, the signals of the 'N' rows EARLIER in the list that sit on the same physical channel with another
 *      code, are tracking and past their code-lock second, are rebuilt from this second's tracking records (MAI_up
 *      :1011-1020) and subtracted (MAI_out :1022-1027); the row's received power is that of the cleaned stream (:515-516), its
 *      PRN is printed + 50 (:708,745).
 * Replaces experiments/231001_DLL_PLL/rxcomplex.cpp as a whole (everything between reading the parameter file and the
 * rows it appends): the parameter parser and per-channel set-up (:263-460 — PRN_sampling :965-978, memcpy_acq :980-987,
 * lowpass :1020-1037, the replica FFT :434-437, psbb :431-432, all on the device), and the per-second loop (:463-835):
 * short2double x2 interpolation of both physical channels (:914-963), received power (:481-489), per channel either the
 * acquisition sweep with its SNR gate (:521-586) or one tracking epoch (:589-790) with the is_trk / is_first hand-over,
 * the .dat rows (:724-754) and the rxcomplex.log lines (:439-443,580-584,757-781).
 * Differences, all stated: the program draws the acquisition offset with rand() seeded by time(NULL) (:240,529) — here a
 * generator seeded by cfg.seed (or a fixed block), reported per second; 'S' (SIC) rows are refused when ninterp = 2: their code is commented
 * out in rxcomplex.cpp:508-519,539-541,596-598; codes come from <code_dir>/<pn-100>.bin as SDRcode does (:866-884) or from
 * the row's own pointer.  A twx_rx is not thread-safe. */
typedef struct twx_rx twx_rx;
typedef struct twx_rx_row {        /* one row of sdr.param: chA_or_B Sic_or_Normal PRN_no. center_freq chip_rate LPF_cutoff range step least_SNR */
    char ch, mode;                 /* 'A' | 'B',  'N' | 'S' (SIC: ninterp = 1 only) */
    int16_t reserved;
    int32_t pn;                    /* 100.. : 100 000-chip SDR codes, 40 ms (:305-311); < 100: 10 000 chips, 4 ms (:299-304) */
    double fc_init;                /* Hz */
    int32_t kcps;                  /* 2500 */
    int32_t reserved2;
    double fltkhz, frange, fstep, snr_min_db;
    const uint8_t* code;           /* NULL: read <code_dir>/<pn-100>.bin (pn >= 100); else code_len bytes 0/1 */
    int64_t code_len;
} twx_rx_row;
typedef struct twx_rx_config {
    double fs_in;                  /* sample rate of the capture, 5e6 (sps / Ninterp, :33) */
    int32_t ninterp;               /* 2: rxcomplex.cpp (:29);  1: rx.cpp, real samples, SIC rows allowed */
    int32_t dec_a;                 /* 1: X310 build, 2: N210/B210 build (:226-231) */
    const char* code_dir;          /* where 0.bin, 1.bin live (NULL: current directory) */
    const char* out_dir;           /* where the .dat files and rxcomplex.log are appended (NULL: no files, reports only) */
    uint64_t seed;                 /* generator of the acquisition offset */
    int32_t acq_block;             /* >= 0: the offset is always this many code periods (idx = acq_block * nobs) */
    int32_t device;                /* -1 = current */
} twx_rx_config;
enum { TWX_RX_NO_SIGNAL = 0,       /* searched, gate not passed (:573) */
       TWX_RX_ACQUIRED = 1,        /* gate passed this second: "analyzing" (:574-585) */
       TWX_RX_CODE_LOCK = 2,       /* first tracking epoch after acquisition: state updated, no row yet (:757-767) */
       TWX_RX_TRACKED = 3,         /* tracking epoch with a .dat row (:724-754) */
       TWX_RX_ACQ_FAILED = 4,      /* first epoch had too few usable periods (:777-781) */
       TWX_RX_LOCK_LOST = 5 };     /* a later epoch had too few usable periods (:783-789) */
typedef struct twx_rx_report {     /* one channel, one second */
    int32_t status, cnt;           /* TWX_RX_*; usable code periods of the epoch */
    double fc, df, phi;            /* carrier (Hz, integer part), its fraction, phase (cycles) */
    double gd, dg, sdgd;           /* code phase (ns), its rate (ns/s), scatter (ns) */
    double pk, px;                 /* signal power, received power (V^2) */
    int64_t pt;                    /* code phase in samples of the interpolated stream */
    int64_t acq_idx, n_trials;     /* acquisition: sample offset used, trial carriers evaluated (else 0) */
    char dat_row[128];             /* TWX_RX_TRACKED: the row appended to ch<A|B>.pn<id>.<kcps>kcps.dat, newline included */
} twx_rx_report;
typedef struct twx_rx_channel_info {
    int32_t pn, is_chA, clen, nlag, bps, is_sic;
    int64_t nobs, nfft;
    double duration, range, step, snr_min, psbb;
    char dat_name[64];
} twx_rx_channel_info;
/* Parses a parameter file with the program's own rules (:263-296: '#' comments, 9 tokens, the value ranges of :288;
 * rows that fail them are skipped as the program skips them); returns the number of rows stored (<= max_rows) or < 0. */
int twx_rx_parse_param(const char* path, twx_rx_row* rows, int32_t max_rows);
int twx_rx_create(const twx_rx_config* cfg, const twx_rx_row* rows, int32_t n_rows, twx_rx** out);
void twx_rx_destroy(twx_rx* rx);
const char* twx_rx_last_error(const twx_rx* rx);               /* rx may be NULL: last create error */
int twx_rx_channel(const twx_rx* rx, int32_t i, twx_rx_channel_info* info);
/* One pass of the loop body :468-832 on one second of capture: fs_in frames [IA QA IB QB] of int16 in HOST (iq) or DEVICE
 * (iq_dev) memory; reports[n_rows]. */
int twx_rx_second(twx_rx* rx, const int16_t* iq, twx_rx_report* reports);
int twx_rx_second_dev(twx_rx* rx, const void* iq_dev, twx_rx_report* reports);
/* The program's main loop over a capture file (`./rxcomplex data.bin sdr.param`): whole seconds until the file ends or
 * max_seconds; reports (may be NULL) receives n_rows records per second, capacity report_seconds seconds. */
int twx_rx_file(twx_rx* rx, const char* path, int64_t max_seconds, twx_rx_report* reports, int64_t report_seconds, int64_t* n_seconds);
/* What the program prints on stdout after a second (:804-831): twx_rx_powers gives the two received powers of the "PWR A / PWR B"
 * line in V^2 (a physical channel no row listens to is not converted: 0), twx_rx_console_line the line of channel i — "no signal"
 * (SNR Low, or SNR x < y), "analyzing", or carrier / code phase / SNR — from the report of that second, newline included;
 * returns the length or < 0.  apps/rxcomplex_hip.cpp is the program built on these: `./rxcomplex_hip [data.bin [sdr.param]]`. */
int twx_rx_powers(const twx_rx* rx, double pwr_v2[2]);
int twx_rx_console_line(const twx_rx* rx, int32_t i, const twx_rx_report* report, char* buf, int32_t cap);
/* Device pointer to the interpolated stream of physical channel 0 (A) / 1 (B) of the last second (fs_in*ninterp complex floats;
 * ninterp = 1: the real samples with a zero imaginary part); physical_channel 2: the interference-free stream of the LAST 'S' row
 * processed (NULL when there is none). */
const void* twx_rx_stream_dev(const twx_rx* rx, int32_t physical_channel);

/* Tracked multi-code ranging: a whole capture in, per-code delay records out ---------------------------------
 * Replaces the capture loops of the three production scripts, everything between fopen and `save`:
 *   acquisition/claudio_aligned_code_ranging_separate.m:143-205 — search_df (:27-47) on the first chunk(s), per chunk the
 *     carrier from the 7 bins around kbon of fft([dold;d].^2) (:166-169), the 40-ms code loop with processing(dpart,df),
 *     the MOVED re-alignment (:170-193) and the `dold` carry (:196-200);
 *   acquisition/claudio_aligned_code_re_separate.m — the same text with the remote band (:137-141);
 *   acquisition/claudio_aligned_code_lo_separate.m:117-164 — no search: carrier = arg-max of the fresh chunk's
 *     fftshift(abs(fft(d.^2))) over the whole band k (:126,129), processing(dpart,k,df), floor() of the lag (:134).
 * The capture is a single-channel int16 [I Q] file (:148-151) or host buffer; chunks stay in device memory and the
 * codes of a chunk are measured in one batched launch, cut and re-measured where the script re-aligns.
 * A twx_tracked owns its correlator context (convention claudio, Octave variances); not thread-safe. */
typedef struct twx_tracked twx_tracked;
enum { TWX_TRK_RANGING = 0, TWX_TRK_RE = 1, TWX_TRK_LO = 2 };
enum { TWX_CARRIER_SEARCH_DF = 0,        /* search_df once, then kbon-3..kbon+3 per chunk (ranging, re) */
       TWX_CARRIER_CHUNK_BAND = 1 };     /* arg-max over the band on every fresh chunk (lo) */
typedef struct twx_tracked_config {
    double fs;                /* 5e6 */
    int32_t sps, nint;        /* 2, 1 */
    const uint8_t* chips;     /* code file bytes 0/1, or NULL with the LFSR fields below */
    int64_t n_chips;
    int32_t lfsr_bitlen, lfsr_taps;
    int64_t chunk_samples;    /* fs*ls complex samples per fread (:16,148); a whole number of code periods */
    double band_lo_hz, band_hi_hz;   /* k=find((freq>lo)&(freq<hi)) on freq=linspace(-fs/2,fs/2-1,chunk_samples) (:132-141) */
    int32_t carrier;          /* TWX_CARRIER_* */
    int32_t indice_floor;     /* 1: indice1=floor(indice/(2*Nint+1)) (lo :134); 0: plain division (:174) */
    double df_threshold;      /* 20 (:20) */
    int64_t skip_samples;     /* the script's fseek(f,30*fs*2*2) (:128) in complex samples; what twx_tracked_file uses when
                                 its own skip argument is negative */
    int32_t precision;        /* TWX_F32 / TWX_F64 */
    int32_t device;           /* -1 = current */
    int32_t max_batch, reserved;
} twx_tracked_config;
/* Fills every mode-dependent field (band, carrier, indice_floor, skip_samples, chunk_samples = 2*fs, df_threshold) and the
 * script constants sps = 2, nint = 1 for TWX_TRK_*; OP is the station flag of the scripts (band sign of the remote
 * channel, :137-141).  chips / n_chips / lfsr_* / precision / device are left as they are. */
int twx_tracked_defaults(int32_t mode, int32_t OP, double fs, twx_tracked_config* cfg);
/* one code period: xval1(p) indice1(p) correction1(p) SNR1r(p) SNR1i(p) puissance1(p) (:173-174,185) */
typedef struct twx_tracked_code { double xval[2]; double indice1, correction1, SNR1r, SNR1i, puissance1; } twx_tracked_code;
typedef struct twx_tracked_summary {
    int64_t n_codes, n_chunks, n_moved;
    int64_t kbon;             /* 0-based index of the accepted carrier bin on the shifted chunk axis, -1: none (script: 0) */
    int64_t batches;          /* batched correlation calls issued (diagnostic) */
    double puissancecode, puissancenoise;   /* the script's workspace keeps the LAST code's values (:173) */
} twx_tracked_summary;
int twx_tracked_create(const twx_tracked_config* cfg, twx_tracked** out);
void twx_tracked_destroy(twx_tracked* trk);
const char* twx_tracked_last_error(const twx_tracked* trk);     /* trk may be NULL: last create error */
twx_ctx* twx_tracked_context(twx_tracked* trk);                /* the correlator context it owns (inspection, profiling) */
/* Run over a capture file / a host buffer of n_samples complex samples.  skip_samples < 0: the configured skip.
 * kbon_hint >= 0: carrier bin known beforehand (no search_df).  The records stay in the object until the next run. */
int twx_tracked_file(twx_tracked* trk, const char* path, int64_t skip_samples, int64_t kbon_hint, twx_tracked_summary* summary);
int twx_tracked_host(twx_tracked* trk, const int16_t* iq, int64_t n_samples, int64_t skip_samples, int64_t kbon_hint,
                     twx_tracked_summary* summary);
/* Copies the records of the last run: codes[n_codes], df[n_chunks] (Hz, one per chunk), moved[n_moved] (the 1-based p
 * of every re-alignment) and movedval[n_moved]; any pointer may be NULL. */
int twx_tracked_fetch(twx_tracked* trk, twx_tracked_code* codes, double* df, int64_t* moved, double* movedval);
/* Where the last run's wall time went: seconds[TWX_TRK_NSTAGES] spent inside each device operation of the control flow (the
 * chunk wait = file read + PCIe copy not hidden behind the previous chunk's measurements; carrier bins; band spectrum; the
 * batched code measurements incl. their result copy; search_df's candidates; the tail carry) and the whole run; calls may be NULL. */
enum { TWX_TRK_T_LOAD = 0, TWX_TRK_T_SQBINS, TWX_TRK_T_SQBAND, TWX_TRK_T_MEASURE, TWX_TRK_T_CANDIDATE, TWX_TRK_T_SLIDE, TWX_TRK_T_TOTAL, TWX_TRK_NSTAGES };
int twx_tracked_timing(const twx_tracked* trk, double* seconds, int64_t* calls);
/* search_df alone (:27-47) on the first chunk_samples of a host buffer: *kbon as in the summary. */
int twx_tracked_search_df(twx_tracked* trk, const int16_t* iq, int64_t n_samples, int64_t* kbon);

/* Several GPUs from ONE host process ----------------------------------------------------------------------------
 * A MATLAB / Octave / C host is one process; the reference's own shape for concurrent correlations is threads inside the
 * program (one GoRanging worker per channel with a reader hand-off, processing/CPP/main.cpp:180-187,488-497) and jobs side
 * by side (acquisition/goprocess.sh:9-11).  A twx_multi owns one correlator context and one persistent host thread per
 * entry of `devices` (NULL: 0, 1, ... modulo the visible devices).  The windows of a capture are cut into contiguous blocks,
 * sizes differing by at most one (one contiguous file extent per device — independent windows, godual_ranging.m:75-102);
 * every context works through its block with twx_process_file / twx_process_windows[_dev], and the fixed-size records are
 * exchanged with ONE ncclAllGather over xGMI (RCCL communicators from ncclCommInitAll, every device's call inside one
 * ncclGroupStart/End).  RCCL needs one rank per device: when the list names a device twice the blocks are concatenated on
 * the host instead (a one-GPU box then exercises the threading and the ordering with e.g. {0,0,0,0}).  RCCL is bound at run
 * time (librccl.so.1, TWX_RCCL_LIB overrides), so single-GPU hosts never map it.  Results are those of one context, record
 * for record.  Not thread-safe: one caller at a time per twx_multi.
 * The exchange never loses the job: RCCL that cannot be loaded, an ncclCommInitAll that fails or does not return within
 * TWX_RCCL_INIT_TIMEOUT_S (120), a collective that fails or does not complete within TWX_RCCL_GATHER_TIMEOUT_S (60) turn the
 * object to host-side concatenation for good, reported in twx_multi_info.rccl_fallback / rccl_error. */
typedef struct twx_multi twx_multi;
enum { TWX_MULTI_NO_RCCL = 1,      /* host-side concatenation even for distinct devices */
       TWX_MULTI_RCCL_ONE = 2 };   /* a list of ONE device still builds its RCCL world of one (same calls as N > 1) */
typedef struct twx_multi_info {
    int32_t n_contexts, n_devices_distinct;
    int32_t rccl;                  /* 1: the record exchange is ncclAllGather; 0: host-side concatenation */
    int32_t rccl_version;          /* ncclGetVersion, 0 when RCCL is not in use */
    int64_t records_gathered;      /* last call: records in the gathered buffer (blocks padded to the longest) */
    int64_t bytes_per_rank;        /* last call: bytes each context contributed */
    double gather_ms;              /* last call: wall time of the collective incl. its synchronisation */
    /* ABI 5 */
    int32_t rccl_fallback;         /* 0: none.  1: RCCL could not be loaded / ncclCommInitAll failed or timed out at creation; 2: a collective
                                    * failed or timed out later.  Either way the records are concatenated on the host since (rccl = 0) */
    int32_t threads_pinned;        /* worker threads bound to the CPUs of their device's NUMA node */
    int32_t numa_node[64];         /* per context: NUMA node of its device (-1: the platform does not say) */
    char rccl_error[256];          /* the text behind rccl_fallback ("" when 0) */
} twx_multi_info;
int twx_multi_create(const twx_config* cfg, const int32_t* devices, int32_t n_devices, int32_t flags, twx_multi** out);
void twx_multi_destroy(twx_multi* m);
const char* twx_multi_last_error(const twx_multi* m);          /* m may be NULL: last create error */
int twx_multi_get_info(const twx_multi* m, twx_multi_info* info);
twx_ctx* twx_multi_context(twx_multi* m, int32_t i);           /* context i (inspection, twx_get_info, options) */
/* twx_process_file over all devices: same arguments, same records in `out`, *n_done = windows processed. */
int twx_multi_process_file(twx_multi* m, const char* path, int32_t n_channels, int32_t channel, int64_t skip_samples,
                           const twx_band* band, double df_const, twx_result* out, int64_t max_windows, int64_t* n_done);
/* twx_process_windows over all devices (capture in HOST memory, e.g. the MEX argument). */
int twx_multi_process_windows(twx_multi* m, const int16_t* iq, int64_t n_windows, int32_t n_channels, int32_t channel,
                              const twx_band* band, const double* df, twx_result* out);
/* Device-resident form: context i processes n_windows windows of ITS OWN recording iq_dev[i] (memory of its device) —
 * BASELINE.json configs[3]'s weak-scaling step.  The records are exchanged device to device; out (host, n_contexts *
 * n_windows [* n_channels] records in context order) may be NULL.  df as in twx_process_windows_dev, shared by all. */
int twx_multi_process_windows_dev(twx_multi* m, const void* const* iq_dev, int64_t n_windows, int32_t n_channels,
                                  int32_t channel, const twx_band* band, const double* df, twx_result* out);
/* BASELINE.json configs[3] as written — ONE recording of n_windows_total consecutive windows (godual_ranging.m:75-102) sharded over the
 * contexts: context i owns the contiguous block twx_multi_block() names (sizes differ by at most one; the rule of
 * twx_multi_process_file) and iq_block_dev[i] points at the FIRST window of that block in the memory of its device.  One exchange per
 * call, on blocks padded to the longest; out (host, may be NULL): n_windows_total [* n_channels] records in WINDOW order; df (when
 * band == NULL): one entry per window [and channel] of the whole recording.  Every context's device copy of the padded gather stays
 * available through twx_multi_fetch_gathered (context r's block starts at record r * ceil(n_windows_total / n_contexts) [* n_channels]). */
int twx_multi_block(const twx_multi* m, int64_t n_windows_total, int32_t i, int64_t* start, int64_t* count);
int twx_multi_process_recording_dev(twx_multi* m, const void* const* iq_block_dev, int64_t n_windows_total, int32_t n_channels,
                                    int32_t channel, const twx_band* band, const double* df, twx_result* out);
/* The exchange ALONE (the compute removed): gathers records_per_context records per context as the last device-resident call left them
 * in the send buffers; its wall time is twx_multi_info.gather_ms.  What one step of a sharded recording pays for the gather. */
int twx_multi_exchange_only(twx_multi* m, int64_t records_per_context);
/* Context i's copy of the gathered records of the last *_dev call (what the collective delivered to that device). */
int twx_multi_fetch_gathered(twx_multi* m, int32_t i, twx_result* out, int64_t n_records);
/* NUMA placement of a device (/sys/bus/pci/devices/<bus id>/{numa_node,local_cpulist}): *numa_node = -1 where the platform does
 * not say; cpulist (may be NULL) receives the kernel's list text.  twx_pin_thread_to_device binds the CALLING thread to those CPUs
 * (intersected with the CPUs it may already use; TWX_NO_PIN=1 or an unknown node: nothing is changed, still TWX_OK) — what the
 * twx_multi workers do for themselves, exported for hosts that run one process or thread per GPU of their own. */
int twx_device_affinity(int32_t device, int32_t* numa_node, char* cpulist, size_t cap);
int twx_pin_thread_to_device(int32_t device, int32_t* numa_node, int32_t* n_cpus);

/* The C++ program of the reference as a library (processing/CPP/main.cpp) --------------------------------------------
 * twx_file_df = GoRanging::df (:363-450): ONE carrier estimate per capture FILE and channel of a 2-channel int16 capture
 * [I1 Q1 I2 Q2]: every n_dec-th frame of the whole file (the program: n_dec = 25, :776), mixed by foffset, minus the mean of the raw
 * samples, squared, DFT of that arbitrary length, halves swapped, arg-max of |.| — channel 1 inside +-2*8 kHz of the decimated axis,
 * channel 2 (remote = 0 only; *df2 = NaN otherwise, df2 may be NULL) over the whole spectrum — and freq(pos)/2 + foffset.  The
 * transform runs on the device (Bluestein in blocks on the library's fp64 FFT of 5e6 points): any file length, no plan plug-in.
 * The two numbers are what the program passes on as the carrier of every window (twx_process_file's df_const).
 * twx_write_cmat = GoRanging::save (:521-656): the `<capture>C.mat` container (MAT v5, uncompressed, n x 1 columns) from the
 * records of channel 1 and — when not NULL — channel 2: correction<c> = indice0 + correction (:310), SNR<c> in dB (:355), df<c>,
 * puissance<c>, puissance<c>code in dB (:343), complex xval<c>, xval<c>m1, xval<c>p1.  Errors: twx_file_df_last_error(). */
int twx_file_df(const char* path, double fs, int32_t n_dec, int32_t remote, double foffset, int32_t device, double* df1, double* df2);
int twx_write_cmat(const char* path, const twx_result* res1, const twx_result* res2, int64_t n_windows);
const char* twx_file_df_last_error(void);

/* Profiling (TWX_FLAG_PROFILE): per kernel class, HIP-event time on the context's stream. */
#define TWX_PROF_MAX 16
typedef struct twx_prof_entry { char name[32]; double ms_total; int64_t launches; int64_t units; } twx_prof_entry;
int twx_profile_reset(twx_ctx* ctx);
int twx_profile_get(twx_ctx* ctx, twx_prof_entry* entries, int32_t max_entries, int32_t* n_entries);
/* Diagnostic builds only (-DTWX_STAMPS, tools/stamps.py): s_memtime stamps written by one wave per workgroup of
 * the Stockham row kernel; TWX_E_STATE in a normal build. */
int twx_debug_stamps(twx_ctx* ctx, unsigned long long* out, long long count);

/* Device-side helpers (replica generation, synthetic captures) ---------------------------- */
/* LFSR chips into host memory, generated on the device (common.py:59-73 semantics). */
int twx_lfsr_chips(int32_t bitlen, int32_t taps, int64_t n, uint8_t* out_host);
/* Integer synthetic capture generator (amaranth_twstft_amd/synth.py) writing int16 into DEVICE
 * memory: n samples × n_channels, channel c described by params[c*8 .. c*8+7] =
 * {delay_q8, fstep, phi0, amp, noise_gain, seed, stream, 0}; sample indices start at n0. */
int twx_synth_capture_dev(void* out_dev, int64_t n, int64_t n0, const uint8_t* chips_dev, int64_t n_chips,
                          int32_t sps, int32_t n_channels, const int64_t* params_host, void* stream);
/* Device memory OWNED BY THE CONTEXT (on its device): released by twx_ctx_free or, at the latest, by twx_destroy — a host that
 * bails out on an error leaks nothing.  twx_dev_alloc / twx_dev_free are the context-free forms (plain hipMalloc / hipFree). */
void* twx_ctx_alloc(twx_ctx* ctx, size_t bytes);
void twx_ctx_free(twx_ctx* ctx, void* p);
void* twx_dev_alloc(size_t bytes);
void twx_dev_free(void* p);
int twx_memcpy_h2d(void* dst_dev, const void* src_host, size_t bytes);
int twx_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* TWSTFT_HIP_H */
