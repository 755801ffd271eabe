// What the translation units of libppgfront share about a context beyond the C ABI (not exported).
#pragma once
#include "../../include/ppgfront.h"

#define PPG_HIDDEN __attribute__((visibility("hidden")))

// records the message in the context (or, for ctx == NULL, in the calling thread's ppg_last_error(NULL) slot) and returns `code`
PPG_HIDDEN int ppg_set_error(ppg_ctx* ctx, int code, const char* fmt, ...) __attribute__((format(printf, 3, 4)));
PPG_HIDDEN int ppg_ctx_device(const ppg_ctx* ctx);
PPG_HIDDEN int ppg_ctx_max_num(const ppg_ctx* ctx);
PPG_HIDDEN const ppg_config* ppg_ctx_config(const ppg_ctx* ctx);
PPG_HIDDEN void* ppg_ctx_match_stream(ppg_ctx* ctx);   // hipStream_t of the context's resident matcher
PPG_HIDDEN int ppg_ctx_raise_match_stream(ppg_ctx* ctx);   // replaces that stream by one of the highest priority
// device address and row count of the descriptors of frame `frame` of the batch that the last ppg_extract* / ppg_wait_batch
// on this context handed out (the slot ring keeps them untouched until the next of those calls)
PPG_HIDDEN int ppg_ctx_last_desc(ppg_ctx* ctx, int frame, const float** d_desc, int* n);
// all-pairs claim match of `batch` resident problems on the context's match stream (ppg_match.cpp); counts are device ints
extern "C" PPG_HIDDEN int match_resident(ppg_ctx* c, const float* dA, const int* nA, const float* dB, const int* nB, int batch,
                                         float nnratio, int32_t* d_match, int* d_mcounts);
