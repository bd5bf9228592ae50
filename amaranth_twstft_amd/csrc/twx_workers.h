// twx_workers.h — host threads of the library, plain C++ (no HIP in this file, so the CPU test suite can run it under
// -fsanitize=thread: tests/cpu/threads_tsan.cpp):
//   Worker            one persistent thread with a one-deep job slot (twx_multi: one per device context)
//   read_in_pieces    one chunk of a capture fetched as several concurrent pieces (the ingest of twx_process_file /
//                     twx_process_windows: one pread / memcpy runs at ~5 GB/s, the PCIe copy behind it at ten times that)
#pragma once
#include <stddef.h>
#include <algorithm>
#include <condition_variable>
#include <functional>
#include <future>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

namespace twx {

struct Worker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<int()> job;
    bool has_job = false, done = true, quit = false;
    int rc = 0;
    // `prologue` runs once on the new thread before the first job (hipSetDevice of the worker's device)
    void start(std::function<void()> prologue, int fail_nomem, int fail_other) {
        th = std::thread([this, prologue, fail_nomem, fail_other]() {
            if (prologue) prologue();
            std::unique_lock<std::mutex> lk(mu);
            for (;;) {
                cv.wait(lk, [this]() { return has_job || quit; });
                if (quit) return;
                std::function<int()> f = std::move(job);
                has_job = false;
                lk.unlock();
                int r;
                try { r = f(); } catch (const std::bad_alloc&) { r = fail_nomem; } catch (...) { r = fail_other; }
                lk.lock();
                rc = r; done = true;
                cv.notify_all();
            }
        });
    }
    void submit(std::function<int()> f) {
        std::lock_guard<std::mutex> lk(mu);
        job = std::move(f); has_job = true; done = false;
        cv.notify_all();
    }
    int wait() {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [this]() { return done; });
        return rc;
    }
    void stop() {
        if (!th.joinable()) return;
        { std::lock_guard<std::mutex> lk(mu); quit = true; cv.notify_all(); }
        th.join();
    }
    ~Worker() { stop(); }
};

// A process-wide pool of reader threads (round 6): the pieces of a chunk used to be std::async threads of their own — 32 thread starts per
// 160-MB chunk, 768 for a 4-GB capture.  The pool grows to the largest reader count ever asked for and lives as long as the process (its
// threads are detached: nothing of the library's shutdown waits for a thread that sits in a condition variable).
class IoPool {
    struct State {
        std::mutex mu;
        std::condition_variable cv;
        std::vector<std::function<void()>> queue;
        int threads = 0, idle = 0;
    };
    std::shared_ptr<State> st = std::make_shared<State>();      // owned by the workers as well: they may outlive the static object
    static void worker(std::shared_ptr<State> s) {
        std::unique_lock<std::mutex> lk(s->mu);
        for (;;) {
            ++s->idle;
            s->cv.wait(lk, [&]() { return !s->queue.empty(); });
            --s->idle;
            std::function<void()> f = std::move(s->queue.back());
            s->queue.pop_back();
            lk.unlock();
            f();
            lk.lock();
        }
    }
public:
    static IoPool& instance() { static IoPool* p = new IoPool(); return *p; }      // never destroyed
    // runs f on a pool thread; at most `want` threads exist because of this call (more may exist already)
    template <class F> std::future<size_t> submit(F f, int want) {
        auto task = std::make_shared<std::packaged_task<size_t()>>(std::move(f));
        std::future<size_t> fut = task->get_future();
        {
            std::lock_guard<std::mutex> lk(st->mu);
            st->queue.push_back([task]() { (*task)(); });
            if (st->idle < (int)st->queue.size() && st->threads < std::max(1, want)) {
                ++st->threads;
                std::thread(worker, st).detach();
            }
        }
        st->cv.notify_one();
        return fut;
    }
};

// [off0, off0 + need) of a source into dst as up to `nthreads` concurrent pieces (4096-byte aligned cuts, pieces of at least
// 4 MB).  read_at(dst, offset, len) -> bytes delivered (short only at the end of the source).  Returns the bytes delivered
// CONTIGUOUSLY from off0 (a short piece ends the count: what follows it is not part of the capture).
// on_piece(lo, got) runs in the thread that read a piece, as soon as its `got` bytes (at dst + lo) are there: the ingest hands every
// piece on to the device at once, so the PCIe copy of a chunk runs beside the reads of its later pieces.
struct NoPieceHook { void operator()(size_t, size_t) const {} };
template <class ReadAt, class OnPiece = NoPieceHook>
size_t read_in_pieces(ReadAt read_at, char* dst, size_t off0, size_t need, int nthreads, OnPiece on_piece = OnPiece(), size_t sub = 0) {
    const int P = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(1, nthreads), need >> 21));      // pieces of at least 2 MB
    const size_t piece = ((need + P - 1) / P + 4095) & ~(size_t)4095;
    // a piece is fetched in sub-pieces of `sub` bytes (0: in one go), each handed on as soon as it is there: the copy to the device of a
    // piece's first megabytes runs beside the read of its last ones instead of behind it
    auto fetch = [=](size_t lo, size_t hi) -> size_t {
        const size_t step = sub ? std::max<size_t>(sub, 1 << 20) & ~(size_t)4095 : hi - lo;
        size_t got_all = 0;
        for (size_t a = lo; a < hi; a += step) {
            const size_t len = std::min(step, hi - a);
            const size_t g = read_at(dst + a, off0 + a, len);
            if (g) on_piece(a, g);
            got_all += g;
            if (g < len) break;                                 // the source ends here
        }
        return got_all;
    };
    std::vector<std::future<size_t>> parts;
    for (int i = 1; i < P; ++i) {
        const size_t lo = std::min(need, piece * i), hi = std::min(need, piece * (i + 1));
        parts.push_back(IoPool::instance().submit([=]() { return lo < hi ? fetch(lo, hi) : (size_t)0; }, 3 * nthreads));      // (three chunks are in flight at a time)
    }
    const size_t first_len = std::min(need, piece);
    size_t total = fetch(0, first_len);
    bool contiguous = total == first_len;
    for (int i = 1; i < P; ++i) {
        const size_t lo = std::min(need, piece * i), hi = std::min(need, piece * (i + 1));
        const size_t got = parts[i - 1].get();
        if (contiguous) { total += got; contiguous = got == hi - lo; }
    }
    return total;
}

}  // namespace twx
