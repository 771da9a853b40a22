// pg_internal.h -- shared by host.cpp and kernels.hip (not part of the ABI)
#ifndef PG_INTERNAL_H
#define PG_INTERNAL_H
#include "pangaea_feat.h"

// records the message for pg_last_error() on this thread and returns `code`
int pg_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

// the threaded interleaved ingest with a sink for its pieces (host.cpp; the sink of ingest_dev.hip copies them to the GPU): a
// worker thread calls copy() with the packed characters of a finished piece (n_words words of codes and of validity bits, packed
// from bit 0) and the word offset the piece was given in the staging arrays; copy() returns 0 or a pg_status whose message it
// has recorded.  Pieces are at most capacity_words words together.
struct pg_piece_sink {
    void *ctx;
    int64_t capacity_words;
    // (lowq: the low-quality plane of a piece of -1 / -2 input, NULL for interleaved input)
    int (*copy)(void *ctx, int worker, int64_t dst_word, const uint64_t *codes, const uint32_t *valid, const uint32_t *lowq, int64_t n_words);
};
int pg_internal_ingest_to_sink(const char *path, int part, int n_parts, const int64_t *newlines_before, const pg_piece_sink *sink, pg_reads **out);
int pg_internal_ingest_pair_to_sink(const char *r1, const char *r2, const pg_piece_sink *sink, pg_reads **out);
bool pg_internal_reads_staged_lowq(const pg_reads *r);
// the pieces of a pg_reads that came through a sink: piece p lies at word soff[p] of the staging arrays and belongs at characters
// [cstart[p], cstart[p + 1]) of the stream; returns their number, -1 for a pg_reads with host arrays
int64_t pg_internal_reads_pieces(const pg_reads *r, const int64_t **soff, const int64_t **cstart);

// the row shuffle of kernels.hip (S2 + S3), for the pipelines of other translation units: the (row, bin) words of bucket b
// lie at words_e[in_begin[b] .. emit_end[b]) inside `workspace` (layout below; emit_end at emit_off, uint64 per bucket)
struct pg_shuffle_layout {
    int vbits;
    size_t emit_off, words_e_off, words_a_off, total;      // words_a: the shuffle's second word buffer, free until it runs
};
// one_pass_bits: up to 2^one_pass_bits row groups the first scatter is the only one (PG_SHUFFLE_ONE_PASS_BITS = what the scatter kernels of
// kernels.hip manage; a lookup pass that scatters by itself may manage more and must then name the same number in every call)
#define PG_SHUFFLE_ONE_PASS_BITS 10
// no_input: the caller keeps its provisional data elsewhere (the merged lookups): with one scatter pass the first word buffer is
// left out of the layout (every call on one workspace must say the same)
int pg_internal_shuffle_layout(int64_t cap, int64_t n_rows, int vsize, pg_shuffle_layout *out, int one_pass_bits = PG_SHUFFLE_ONE_PASS_BITS, int no_input = 0);
// a lookup pass that scatters its (row, bin) words by row group itself (mini.hip): prepare fills `ctx` and clears the cursors;
// the pass puts a word whose first digit is d = (word >> dshift) & (2^gb1 - 1) at words_out[goff[d << gb2] + (atomicAdd on
// gcur1[d])]; finish runs what is left (second pass for more than 2^one_pass_bits row groups, row histograms).
// `narrow` (one pass only, i.e. at most 2^one_pass_bits row groups): a group region already says which 64 rows a word belongs to, so the
// pass stores 2-byte words -- (row & 63) << vbits | bin, 15 bits at most -- at the same element offsets of words_out taken as
// uint16_t: half the bytes written by the pass and read by the row histograms.  finish must be told (same flag).
struct pg_shuffle_ctx {
    const unsigned long long *goff;
    unsigned long long *gcur1;
    uint32_t *words_in, *words_out;
    int vbits, gb1, gb2, dshift;
    int narrow;
    unsigned long long words_cap;                 // elements (4-byte words) of words_in / words_out
};
// word forms a lookup pass may leave for finish: plain 4-byte (row << vbits | bin), narrow 2-byte (above), or COUNTED 4-byte:
// (n - 1) << PG_SHUFFLE_COUNT_SHIFT | row << vbits | bin stands for n equal words (needs row bits + vbits <= the shift)
#define PG_SHUFFLE_WORDS_PLAIN 0
#define PG_SHUFFLE_WORDS_NARROW 1
#define PG_SHUFFLE_WORDS_COUNTED 2
#define PG_SHUFFLE_COUNT_SHIFT 28
int pg_internal_shuffle_is_narrow(int64_t cap, int64_t n_rows, int vsize, int one_pass_bits = PG_SHUFFLE_ONE_PASS_BITS);       // what prepare will put into ctx->narrow
int pg_internal_shuffle_prepare(int64_t cap, const pg_rows *rows, int vsize, void *workspace, int64_t workspace_bytes, void *stream,
                                pg_shuffle_ctx *ctx, int one_pass_bits = PG_SHUFFLE_ONE_PASS_BITS, int no_input = 0, int ctx_only = 0);
int pg_internal_shuffle_finish(int64_t cap, const pg_rows *rows, int vsize, int32_t *abd_out, void *workspace, int64_t workspace_bytes, void *stream,
                               int word_form, int one_pass_bits = PG_SHUFFLE_ONE_PASS_BITS, int no_input = 0);
int pg_internal_shuffle_rows(const unsigned long long *in_begin, int nb, int64_t cap, const pg_rows *rows, int vsize, int32_t *abd_out,
                             void *workspace, int64_t workspace_bytes, void *stream);

#endif
