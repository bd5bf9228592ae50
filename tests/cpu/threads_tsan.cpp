// threads_tsan.cpp — the host threads of the library (amaranth_twstft_amd/csrc/twx_workers.h) under -fsanitize=thread, no GPU:
//   * the per-device worker pool of twx_multi: N persistent workers, run_all() rounds as twx_multi::run_all drives them
//     (submit to every worker, wait for every worker, first failure wins), jobs writing disjoint blocks of one result vector;
//   * the ingest of twx_process_file / twx_process_windows: chunks fetched as concurrent pieces (read_in_pieces) into the
//     buffers of a 3-slot rotation while the "device side" consumes the chunk that used the slot before — the hand-over
//     order of Ctx::run_pipeline (a buffer is refilled only after its consumer is done; a short source ends the run).
//   g++ -std=c++17 -O1 -g -fsanitize=thread -Iamaranth_twstft_amd/csrc tests/cpu/threads_tsan.cpp -o threads_tsan -lpthread
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <atomic>
#include <memory>
#include <numeric>
#include "twx_workers.h"

static int fails = 0;
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "CHECK failed: %s (line %d)\n", #c, __LINE__); ++fails; } } while (0)

static void worker_rounds() {
    const int n = 8, rounds = 300;
    std::vector<std::unique_ptr<twx::Worker>> w;
    std::vector<int> started(n, 0);
    for (int r = 0; r < n; ++r) { w.emplace_back(new twx::Worker()); int* s = &started[r]; w.back()->start([s]() { *s = 1; }, -4, -5); }
    std::vector<long long> out(1000, 0);
    for (int it = 0; it < rounds; ++it) {
        const long long total = 1 + (it * 37) % 1000;
        auto f = [&](int r) -> int {
            const long long base = total / n, rem = total % n, start = r * base + std::min<long long>(r, rem), cnt = base + (r < rem ? 1 : 0);
            for (long long i = start; i < start + cnt; ++i) out[(size_t)i] = (long long)it * 1000003 + i;
            if (it == 17 && r == 3) return -3;                       // one failing context
            if (it == 18 && r == 5) throw std::bad_alloc();          // an exception inside a job stays inside the worker
            return 0;
        };
        for (int r = 0; r < n; ++r) w[r]->submit([&f, r]() { return f(r); });
        int rc = 0;
        for (int r = 0; r < n; ++r) { const int rr = w[r]->wait(); if (rr && !rc) rc = rr; }
        CHECK(rc == (it == 17 ? -3 : it == 18 ? -4 : 0));
        for (long long i = 0; i < total; ++i) CHECK(out[(size_t)i] == (long long)it * 1000003 + i);
    }
    for (auto& x : w) x->stop();
    for (int r = 0; r < n; ++r) CHECK(started[r] == 1);
}

static void ingest_rotation(size_t total_bytes, size_t chunk_bytes, int nslots, int nthreads) {
    std::vector<unsigned char> src(total_bytes);
    for (size_t i = 0; i < total_bytes; ++i) src[i] = (unsigned char)((i * 2654435761u) >> 13);
    auto read_at = [&src](char* dst, size_t off, size_t len) -> size_t {
        if (off >= src.size()) return 0;
        const size_t k = std::min(len, src.size() - off);
        memcpy(dst, src.data() + off, k);
        return k;
    };
    std::vector<std::vector<char>> slot((size_t)nslots, std::vector<char>(chunk_bytes));
    std::vector<std::future<size_t>> rd((size_t)nslots);
    std::vector<std::future<unsigned long long>> consumer((size_t)nslots);      // stands for the slot's H2D copy + kernels
    std::vector<unsigned long long> hooked((size_t)nslots, 0);
    unsigned long long hook_sum = 0;
    size_t next_chunk = 0;
    auto start_read = [&](int k) {
        const size_t off = next_chunk++ * chunk_bytes;
        char* dst = slot[(size_t)k].data();
        // the piece hook stands for the per-piece copy to the device: it reads what the piece's thread has just written
        unsigned long long* hk = &hooked[(size_t)k];
        rd[(size_t)k] = std::async(std::launch::async, [=, &read_at]() {
            std::atomic<unsigned long long> seen{0};
            auto hook = [dst, &seen](size_t lo, size_t got) { unsigned long long a = 0; for (size_t i = 0; i < got; ++i) a += (unsigned char)dst[lo + i]; seen += a; };
            const size_t g = twx::read_in_pieces(read_at, dst, off, chunk_bytes, nthreads, hook);
            *hk = seen.load();
            return g;
        });
    };
    for (int k = 0; k < nslots; ++k) start_read(k);
    unsigned long long sum = 0, want = 0;
    for (unsigned char c : src) want += c;
    size_t consumed = 0;
    bool eof = false;
    int kprev = -1;
    for (int k = 0; !eof; k = (k + 1) % nslots) {
        if (!rd[(size_t)k].valid()) { if (consumer[(size_t)k].valid()) sum += consumer[(size_t)k].get(); start_read(k); }
        const size_t got = rd[(size_t)k].get();
        if (got < chunk_bytes) eof = true;
        hook_sum += hooked[(size_t)k];
        if (consumer[(size_t)k].valid()) sum += consumer[(size_t)k].get();       // drain: the batch that used this slot before
        if (got) {
            const char* p = slot[(size_t)k].data();
            consumer[(size_t)k] = std::async(std::launch::async, [p, got]() { unsigned long long s = 0; for (size_t i = 0; i < got; ++i) s += (unsigned char)p[i]; return s; });
            consumed += got;
        }
        // refill the PREVIOUS slot: its consumer had a whole iteration to finish (run_pipeline waits for the slot's H2D event)
        if (kprev >= 0 && kprev != k && !eof) { if (consumer[(size_t)kprev].valid()) sum += consumer[(size_t)kprev].get(); start_read(kprev); }
        kprev = k;
    }
    for (int k = 0; k < nslots; ++k) { if (rd[(size_t)k].valid()) (void)rd[(size_t)k].get(); if (consumer[(size_t)k].valid()) sum += consumer[(size_t)k].get(); }
    CHECK(consumed == total_bytes);
    CHECK(sum == want);
    CHECK(hook_sum == want);
}

int main() {
    worker_rounds();
    ingest_rotation((size_t)37 << 20, (size_t)9 << 20, 3, 4);        // ragged last chunk, several pieces per chunk
    ingest_rotation((size_t)27 << 20, (size_t)9 << 20, 3, 4);        // source ends on a chunk boundary (one empty read)
    ingest_rotation((size_t)5 << 20, (size_t)9 << 20, 1, 4);         // single slot, source shorter than one chunk
    ingest_rotation((size_t)40 << 20, (size_t)8 << 20, 2, 1);        // one piece per chunk
    if (fails) { fprintf(stderr, "%d check(s) failed\n", fails); return 1; }
    printf("threads ok\n");
    return 0;
}
