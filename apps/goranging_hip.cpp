// goranging_hip.cpp — `GoRanging` (processing/CPP/main.cpp) as a command-line drop-in over the C ABI of libtwstft_hip.so:
//
//     goranging_hip data.bin code.bin [remote=0] [foffset=0.]
//
// same arguments (:773-784), same output file name rule (`<capture>C.mat`, `remote` prefixed when remote = 1, :786-798), the
// program's stdout lines (usage line, code / map lengths :115, `df1=` / ` df2=` :431-446, one row per window with the delay, the
// carrier, the window power and the SNR of each channel :315,353,498, `No more data` :468, `temps:` :507) and the variable set of
// GoRanging::save (:541-647).  The arithmetic — file-level carrier estimate, Hamming-windowed code spectrum (:717-719), the
// per-window correlation with x3 interpolation, peak, parabola and wipe-off SNR — runs on the GPU (twx_file_df, twx_process_file).
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <regex>
#include <string>
#include <vector>
#include "twstft_hip.h"

int main(int argc, char** argv) {
    const double fs = 5e6;
    const int N = 25;                        // 5 MS/s decimated by 25 = 200 kS/s (:776)
    const int Nint = 1;
    int remote = 0;
    double foffset = 0.;
    printf("%s data.bin code.bin [remote=0] [foffset=0.]\n", argv[0]);
    if (argc < 3) return EXIT_FAILURE;       // (the program dereferences argv[1] / argv[2] regardless)
    if (argc >= 4) remote = atoi(argv[3]);
    if (argc >= 5) foffset = atof(argv[4]);
    const std::string filename = argv[1];
    std::string matname, fullpath(filename);
    const size_t index = fullpath.find_last_of('/') + 1;
    if (std::string::npos != index) { matname = fullpath.substr(0, index); fullpath.erase(0, index); }
    if (remote == 1) matname += "remote";
    std::regex e("([^ ]*)(.+bin)");
    matname += std::regex_replace(fullpath, e, "$1C.mat");

    // fill_fcode (:658-732): chips {0,1} as bytes
    FILE* fc = fopen(argv[2], "rb");
    if (!fc) { printf("fcode read: FAIL\n"); return EXIT_FAILURE; }
    std::vector<uint8_t> chips;
    {
        uint8_t buf[65536];
        size_t got;
        while ((got = fread(buf, 1, sizeof buf, fc)) > 0) chips.insert(chips.end(), buf, buf + got);
        fclose(fc);
    }
    if (chips.empty()) { printf("fcode read: FAIL\n"); return EXIT_FAILURE; }
    for (uint8_t& c : chips) c = c ? 1 : 0;
    const long long n = 2ll * (long long)chips.size();
    printf("file size : %ld %ld\n", (long)(2 * chips.size()), (long)n);
    printf("%ld %ld\n", (long)n, (long)(n * (2 * Nint + 1)));

    twx_config cfg{};
    cfg.fs = fs; cfg.sps = 2; cfg.nint = Nint; cfg.chips = chips.data(); cfg.n_chips = (int64_t)chips.size();
    cfg.window = TWX_WIN_HAMMING; cfg.convention = TWX_CONV_GODUAL; cfg.precision = TWX_F32; cfg.var_ddof = 0; cfg.snr_rot = -1; cfg.device = -1;
    twx_ctx* ctx = nullptr;
    if (int rc = twx_create(&cfg, &ctx)) { printf("init error: %s (%s)\n", twx_last_error(nullptr), twx_strerror(rc)); return EXIT_FAILURE; }

    double df1 = 0, df2 = 0;
    if (int rc = twx_file_df(filename.c_str(), fs, N, remote, foffset, -1, &df1, &df2)) {
        printf("df: %s (%s)\n", twx_file_df_last_error(), twx_strerror(rc));
        twx_destroy(ctx);
        return EXIT_FAILURE;
    }
    printf("df1=%.3f\n", df1);
    if (remote == 0) printf(" df2=%.3f\n", df2); else printf("\n");

    FILE* fd = fopen(filename.c_str(), "rb");
    if (!fd) { printf("cannot open %s\n", filename.c_str()); twx_destroy(ctx); return EXIT_FAILURE; }
    fseek(fd, 0, SEEK_END);
    const long long nwin = (long long)ftell(fd) / (8ll * n);
    fclose(fd);
    std::vector<twx_result> r1((size_t)std::max<long long>(nwin, 1)), r2((size_t)std::max<long long>(nwin, 1));
    const auto t_start = std::chrono::high_resolution_clock::now();
    int64_t n1 = 0, n2 = 0;
    int rc = twx_process_file(ctx, filename.c_str(), 2, 0, 0, nullptr, df1, r1.data(), nwin, &n1);
    if (!rc && remote == 0) rc = twx_process_file(ctx, filename.c_str(), 2, 1, 0, nullptr, df2, r2.data(), nwin, &n2);
    if (rc) { printf("processing: %s (%s)\n", twx_last_error(ctx), twx_strerror(rc)); twx_destroy(ctx); return EXIT_FAILURE; }
    const auto t_end = std::chrono::high_resolution_clock::now();
    for (long long p = 0; p < n1; ++p) {
        const twx_result& a = r1[(size_t)p];
        printf("%d/%d %0.12lf\t%.3f\t%.1lf\t%0.1lf\t", (int)p, 0, ((double)a.indice0 + a.correction) / fs / (2 * Nint + 1.), a.df, a.puissance,
               10 * log10(a.SNRr + a.SNRi));
        if (remote == 0 && p < n2) {
            const twx_result& b = r2[(size_t)p];
            printf("%d/%d %0.12lf\t%.3f\t%.1lf\t%0.1lf\t", (int)p, 1, ((double)b.indice0 + b.correction) / fs / (2 * Nint + 1.), b.df, b.puissance,
                   10 * log10(b.SNRr + b.SNRi));
        }
        printf("\n");
    }
    printf("No more data\n");
    printf("temps: %lf\n", std::chrono::duration<double, std::milli>(t_end - t_start).count());
    twx_destroy(ctx);
    if (n1 == 0) { printf("Nothing to save\n"); return EXIT_SUCCESS; }
    if (twx_write_cmat(matname.c_str(), r1.data(), remote == 0 ? r2.data() : nullptr, n1)) { printf("mat file creation: FAIL\n"); return EXIT_FAILURE; }
    return EXIT_SUCCESS;
}
