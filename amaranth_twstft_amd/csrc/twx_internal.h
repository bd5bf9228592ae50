// twx_internal.h — what the translation units of libtwstft_hip.so share about a context besides the public C ABI
// (hidden visibility: none of this is exported).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <vector>
#include "../../include/twstft_hip.h"

#define TWX_HIDDEN __attribute__((visibility("hidden")))

namespace twx {
enum { AUX_SCRATCH_SLOTS = 12 };      // 0-1, 6: twx_aux.hip; 2-5: the CAF surface; 7-9: the long squared spectra
TWX_HIDDEN hipStream_t ctx_stream(twx_ctx* ctx);                        // = twx_stream(ctx)
// Context-owned device buffer number `slot`, at least `bytes` long: kept across calls, re-allocated only when it has
// to grow (after synchronising the context), released by twx_destroy.  nullptr on failure (error text set).
TWX_HIDDEN void* ctx_scratch(twx_ctx* ctx, int slot, size_t bytes);
TWX_HIDDEN int ctx_fail(twx_ctx* ctx, int code, const char* msg);       // sets twx_last_error, returns code
TWX_HIDDEN int ctx_set_device(twx_ctx* ctx);
// host-side shadow of a scratch slot's contents (lets a caller skip an upload when the bytes have not changed)
TWX_HIDDEN std::vector<unsigned char>& ctx_scratch_shadow(twx_ctx* ctx, int slot);

// ---- The matrix-core FIR never shares the device with other work of this process (twx_api.hip) -------------------------------------
// k_fir_mfma makes packed-fp32 results of waves resident beside it go wrong (rows of a correlation's k_rowd on another stream:
// profiles/r05_fir_mfma.txt; cause unknown, below the ISA as far as anyone could tell).  The library therefore keeps the two apart BY
// CONSTRUCTION instead of by documentation: a registry of every stream the library has created per device and a reader/writer fence —
//   * every entry point that enqueues work holds a FenceShared for the duration of its enqueue: its stream first waits for the last
//     matrix-core FIR launched on the device (one hipStreamWaitEvent, only if such a launch ever happened);
//   * a matrix-core FIR launch holds a FenceExclusive: its stream waits for everything enqueued so far on every other registered
//     stream of the device, then the kernel is launched and the event the others wait for is recorded behind it.
// No other process's work is covered (nothing in a library can be): TWX_OPT_FIR_MFMA stays opt-in.
TWX_HIDDEN void fence_register(int dev, hipStream_t s);
TWX_HIDDEN void fence_unregister(int dev, hipStream_t s);
struct TWX_HIDDEN FenceShared {
    int dev;
    FenceShared(int dev_, hipStream_t s);
    ~FenceShared();
    FenceShared(const FenceShared&) = delete;
};
struct TWX_HIDDEN FenceExclusive {
    int dev; hipStream_t s;
    FenceExclusive(int dev_, hipStream_t s_);      // s may be the null stream (context-free host form)
    ~FenceExclusive();                             // records the event behind the launch
    FenceExclusive(const FenceExclusive&) = delete;
    int waited = 0;                                // streams this launch was ordered behind (diagnostic)
};
TWX_HIDDEN long long fence_mfma_launches();        // matrix-core FIR launches of this process so far
TWX_HIDDEN int ctx_device(twx_ctx* ctx);
TWX_HIDDEN int ctx_fir_mfma_option(twx_ctx* ctx);  // TWX_OPT_FIR_MFMA of the context: -1 follow the environment, 0 never, 1 use it
}  // namespace twx
