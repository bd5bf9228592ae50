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
}  // namespace twx
