// What the translation units of libppgfront share about a context beyond the C ABI (not exported).
#pragma once
#include "../../include/ppgfront.h"

#define PPG_HIDDEN __attribute__((visibility("hidden")))

// records the message in the context (or, for ctx == NULL, in the calling thread's ppg_last_error(NULL) slot) and returns `code`
PPG_HIDDEN int ppg_set_error(ppg_ctx* ctx, int code, const char* fmt, ...) __attribute__((format(printf, 3, 4)));
PPG_HIDDEN int ppg_ctx_device(const ppg_ctx* ctx);
PPG_HIDDEN int ppg_ctx_max_num(const ppg_ctx* ctx);
