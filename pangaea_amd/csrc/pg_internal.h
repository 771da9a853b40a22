// pg_internal.h -- shared by host.cpp and kernels.hip (not part of the ABI)
#ifndef PG_INTERNAL_H
#define PG_INTERNAL_H
#include "pangaea_feat.h"

// records the message for pg_last_error() on this thread and returns `code`
int pg_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));


// the row shuffle of kernels.hip (S2 + S3), for the pipelines of other translation units: the (row, bin) words of bucket b
// lie at words_e[in_begin[b] .. emit_end[b]) inside `workspace` (layout below; emit_end at emit_off, uint64 per bucket)
struct pg_shuffle_layout {
    int vbits;
    size_t emit_off, words_e_off, words_a_off, total;      // words_a: the shuffle's second word buffer, free until it runs
};
int pg_internal_shuffle_layout(int64_t cap, int64_t n_rows, int vsize, pg_shuffle_layout *out);
int pg_internal_shuffle_rows(const unsigned long long *in_begin, int nb, int64_t cap, const pg_rows *rows, int vsize, int32_t *abd_out,
                             void *workspace, int64_t workspace_bytes, void *stream);

#endif
