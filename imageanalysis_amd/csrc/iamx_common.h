// Internal helpers shared by the libiamx translation units (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/iamx.h"

namespace iamx {

// per-thread last-error text, returned by iamx_last_error()
char *err_buf();
int fail(int code, const char *fmt, ...);

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// Check the launch that was just enqueued.
int check_launch(const char *what);

}  // namespace iamx

#define IAMX_REQUIRE(cond, what)                                               \
    do {                                                                       \
        if (!(cond)) return iamx::fail(IAMX_EINVAL, "%s: %s", __func__, what); \
    } while (0)
