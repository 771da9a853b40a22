// pg_internal.h -- shared by host.cpp and kernels.hip (not part of the ABI)
#ifndef PG_INTERNAL_H
#define PG_INTERNAL_H
#include "pangaea_feat.h"

// records the message for pg_last_error() on this thread and returns `code`
int pg_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

#endif
