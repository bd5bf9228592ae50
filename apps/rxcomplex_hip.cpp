// rxcomplex_hip / rx_hip — the two DLL/PLL receiver programs of experiments/231001_DLL_PLL as command-line drop-ins on the GPU.
//
//   ./rxcomplex_hip [data.bin [sdr.param]]      (rxcomplex.cpp:175-215: the same arguments and defaults)
//   ./rx_hip        [data.bin [sdr.param]]      (rx.cpp, the real-sample program with the SIC rows; built with -DTWX_RX_REAL)
//
// Everything the programs compute is behind the C ABI (twx_rx_* in include/twstft_hip.h); this file is their main(): argument
// handling and error texts (:175-180,205-215,255), the capture loop (:463-835) as twx_rx_second per whole second read, and the
// console lines (:804-831) from twx_rx_console_line.  Codes <pn-100>.bin are read from the current directory and the .dat files /
// rx*.log are appended there, as the programs do (TWX_RX_CODES / TWX_RX_OUT move them).  The programs seed rand() with time(NULL)
// (:240); so does this (TWX_RX_SEED fixes it).  The N210 / B210 build (dec_a = 2, :226-231) is TWX_RX_DEC_A=2.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <vector>
#include "twstft_hip.h"

static double v2todBm(double v2) { return v2 > 0.0 ? 10.0 * log10(v2 * 1000.0 / 25.0) : 0.0; }     // :1236-1240

int main(int argc, char* argv[]) {
    if (argc != 1 && argc != 2 && argc != 3) {                                          // :175-180
        printf("usage:\n");
        printf("%s out_path param_file\n", argv[0]);
        return 1;
    }
    const char* datafile = argc > 1 ? argv[1] : "./data.bin";                         // :205-209
    const char* paramfile = argc > 2 ? argv[2] : "sdr.param";                         // :185,211-212
    printf("%s\n", datafile);
    FILE* fparam = fopen(paramfile, "r");
    if (!fparam) { printf("no such parameter file : %s\n", paramfile); return 1; }    // :213-217
    fclose(fparam);
    FILE* fd = fopen(datafile, "rb");
    if (!fd) { printf("Data filename error\n"); return 1; }                           // :254-255
    std::vector<twx_rx_row> rows(120);                                                 // nch_max :34
    const int n_rows = twx_rx_parse_param(paramfile, rows.data(), (int32_t)rows.size());
    if (n_rows < 1) { printf("no usable row in %s\n", paramfile); fclose(fd); return 1; }
    twx_rx_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.fs_in = 5e6;                                                                   // const int sps :33
#ifdef TWX_RX_REAL
    cfg.ninterp = 1;
#else
    cfg.ninterp = 2;                                                                   // #define Ninterp 2 :29
#endif
    const char* e;
    cfg.dec_a = (e = getenv("TWX_RX_DEC_A")) ? atoi(e) : 1;
    cfg.code_dir = (e = getenv("TWX_RX_CODES")) ? e : ".";
    cfg.out_dir = (e = getenv("TWX_RX_OUT")) ? e : ".";
    cfg.seed = (e = getenv("TWX_RX_SEED")) ? strtoull(e, nullptr, 10) : (uint64_t)time(nullptr);      // srand(time(NULL)) :240
    cfg.acq_block = -1;
    cfg.device = -1;
    twx_rx* rx = nullptr;
    if (twx_rx_create(&cfg, rows.data(), n_rows, &rx)) { printf("%s\n", twx_rx_last_error(nullptr)); fclose(fd); return 1; }
    const size_t n_in = (size_t)cfg.fs_in;
    std::vector<int16_t> buf(n_in * 4);
    std::vector<twx_rx_report> rep((size_t)n_rows);
    char line[512];
    int rc = 0;
    // do { fread ... } while (datares == sps*4/Ninterp) (:468,832); a short final read is not processed
    while (fread(buf.data(), 2, buf.size(), fd) == buf.size()) {
        if (twx_rx_second(rx, buf.data(), rep.data())) { printf("%s\n", twx_rx_last_error(rx)); rc = 1; break; }
        double pwr[2] = {0, 0};
        twx_rx_powers(rx, pwr);
        printf("\nPWR A: %6.2lf dBm , PWR B: %6.2lf dBm\n\n", v2todBm(pwr[0]), v2todBm(pwr[1]));    // :804
        for (int i = 0; i < n_rows; ++i)
            if (twx_rx_console_line(rx, i, &rep[(size_t)i], line, (int32_t)sizeof line) > 0) fputs(line, stdout);
        fflush(stdout);
    }
    twx_rx_destroy(rx);
    fclose(fd);
    return rc;
}
