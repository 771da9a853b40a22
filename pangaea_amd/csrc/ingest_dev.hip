// ingest_dev.hip -- the threaded interleaved ingest with its last phase on the GPU (VERDICT r2 item 7:
// "H2D of finished blocks under the parse").
//
// The host ingest (host.cpp: ingest_interleaved_range) ends by shifting the threads' local streams into place in one host array
// (every piece starts at an arbitrary bit offset of the stream), and only then can the copy to the GPU begin.  Here the byte
// range is cut into pieces of PG_INGEST_PIECE bytes (16 MiB) that the parser threads take from a queue; a thread that has
// packed a piece copies it -- still packed from bit 0 -- into device staging arrays while the other threads go on parsing (so
// the PCIe copy hides under the parse), and the shift into place is one streaming kernel over the staged words
// (ingest_place_kernel: output word g finds the piece(s) its 32 characters come from by binary search over the pieces' first
// characters and funnel-shifts them out of at most two staging words per piece).  No host copy of the stream exists.
// Replaces, for uncompressed interleaved input: count_tnf.cpp:234-283 (the producer loop) + the device copy.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>
#include <mutex>

#include "pangaea_feat.h"
#include "pg_internal.h"

namespace {

constexpr int PLACE_BLOCK = 256;
constexpr int MAX_WORKERS = 64;

int64_t piece_bytes()
{
    if (const char *e = getenv("PG_INGEST_PIECE")) { long v = atol(e); if (v >= 64) return v; }
    return (int64_t)16 << 20;
}
// an upper bound of the number of pieces of any byte range of a file of this size, and the words of their tables
int64_t max_pieces(int64_t file_bytes) { return file_bytes / piece_bytes() + 2; }
int64_t table_words(int64_t file_bytes) { return 2 * (max_pieces(file_bytes) + 1) + 8; }

struct DeviceSink {
    uint64_t *codes;
    uint32_t *valid;
    uint32_t *lowq = nullptr;                    // (-1 / -2 input: the plane of bases with a quality below '?')
    int device;
    std::mutex making;
    hipStream_t streams[MAX_WORKERS] = {};       // one per parser thread, created when the thread has its first piece
};

int sink_copy(void *ctx, int worker, int64_t dst_word, const uint64_t *codes, const uint32_t *valid, const uint32_t *lowq, int64_t n_words)
{
    DeviceSink *s = (DeviceSink *)ctx;
    // (worker threads are fresh threads: their current device is 0 until they say otherwise)
    hipError_t e = hipSetDevice(s->device);
    hipStream_t &slot = s->streams[worker % MAX_WORKERS];
    if (e == hipSuccess) {
        std::lock_guard<std::mutex> lock(s->making);
        if (!slot) e = hipStreamCreateWithFlags(&slot, hipStreamNonBlocking);
    }
    hipStream_t st = slot;
    if (e == hipSuccess) e = hipMemcpyAsync(s->codes + dst_word, codes, (size_t)n_words * sizeof(uint64_t), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(s->valid + dst_word, valid, (size_t)n_words * sizeof(uint32_t), hipMemcpyHostToDevice, st);
    if (e == hipSuccess && lowq && s->lowq) e = hipMemcpyAsync(s->lowq + dst_word, lowq, (size_t)n_words * sizeof(uint32_t), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);             // the local stream is reused for the thread's next piece
    if (e != hipSuccess) return pg_fail(PG_EHIP, "pg_ingest_fastq_device: copy of a piece failed: %s", hipGetErrorString(e));
    return PG_OK;
}

// output word g = characters [32 g, 32 g + 32) of the stream; piece p holds characters [cstart[p], cstart[p + 1]) packed from bit 0
// of staging word soff[p] (bits beyond a piece's last character are zero)
// (sq / lowq: a second 1-bit plane placed like the validity plane -- the low-quality plane of -1 / -2 input; NULL: none)
__global__ __launch_bounds__(PLACE_BLOCK) void ingest_place_kernel(const uint64_t *__restrict__ sc, const uint32_t *__restrict__ sv,
                                                                   const uint32_t *__restrict__ sq,
                                                                   const int64_t *__restrict__ soff, const int64_t *__restrict__ cstart, int n_pieces,
                                                                   uint64_t *__restrict__ codes, uint32_t *__restrict__ valid, uint32_t *__restrict__ lowq,
                                                                   int64_t n_words)
{
    const int64_t total = cstart[n_pieces];
    for (int64_t g = (int64_t)blockIdx.x * PLACE_BLOCK + threadIdx.x; g < n_words; g += (int64_t)gridDim.x * PLACE_BLOCK) {
        const int64_t c0 = g * 32;
        uint64_t c = 0;
        uint32_t v = 0, q = 0;
        if (c0 < total) {
            int lo = 0, hi = n_pieces - 1;                  // the first piece that ends behind c0
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (cstart[mid + 1] > c0) hi = mid; else lo = mid + 1; }
            for (int p = lo; p < n_pieces && cstart[p] < c0 + 32; ++p) {
                const int64_t first = cstart[p], end = cstart[p + 1];
                const int64_t a = first > c0 ? first : c0, b = end < c0 + 32 ? end : c0 + 32;
                if (b <= a) continue;                       // an empty piece
                const int64_t l0 = a - first;
                const int cnt = (int)(b - a), sh = (int)(l0 & 31), d = (int)(a - c0);
                const int64_t i = soff[p] + (l0 >> 5), last = soff[p] + ((end - first - 1) >> 5);
                uint64_t wc = sc[i] >> (2 * sh);
                uint32_t wv = sv[i] >> sh, wq = sq ? sq[i] >> sh : 0u;
                if (sh && i < last) { wc |= sc[i + 1] << (64 - 2 * sh); wv |= sv[i + 1] << (32 - sh); if (sq) wq |= sq[i + 1] << (32 - sh); }
                if (cnt < 32) { wc &= (1ull << (2 * cnt)) - 1; wv &= (1u << cnt) - 1; wq &= (1u << cnt) - 1; }
                c |= wc << (2 * d);
                v |= wv << d;
                q |= wq << d;
            }
        }
        codes[g] = c;
        valid[g] = v;
        if (lowq) lowq[g] = q;
    }
}

}  // namespace

extern "C" int64_t pg_ingest_staging_words(int64_t file_bytes)
{
    if (file_bytes < 0) return pg_fail(PG_EINVAL, "pg_ingest_staging_words: negative size");
    // a character costs at least a byte of the file; every piece rounds up to a whole word; the pieces' tables ride at the end
    return file_bytes / 32 + max_pieces(file_bytes) + 64 + table_words(file_bytes);
}

extern "C" int pg_ingest_fastq_device(const char *path, int part, int n_parts, const int64_t *newlines_before, int64_t file_bytes,
                                      uint64_t *staging_codes, uint32_t *staging_valid, int64_t staging_words, pg_reads **out)
{
    if (!path || !out || !staging_codes || !staging_valid || n_parts < 1 || part < 0 || part >= n_parts || (n_parts > 1 && !newlines_before))
        return pg_fail(PG_EINVAL, "pg_ingest_fastq_device: bad argument");
    *out = nullptr;
    if (staging_words < pg_ingest_staging_words(file_bytes))
        return pg_fail(PG_EINVAL, "pg_ingest_fastq_device: staging arrays of %lld words, pg_ingest_staging_words(%lld) = %lld", (long long)staging_words,
                       (long long)file_bytes, (long long)pg_ingest_staging_words(file_bytes));
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, staging_codes) != hipSuccess || at.type != hipMemoryTypeDevice)
        return pg_fail(PG_EINVAL, "pg_ingest_fastq_device: the staging arrays must be device memory");
    DeviceSink s;
    s.codes = staging_codes; s.valid = staging_valid; s.device = at.device;
    int before = 0;
    (void)hipGetDevice(&before);
    (void)hipSetDevice(s.device);
    pg_piece_sink sink{&s, staging_words - table_words(file_bytes), sink_copy};
    const int rc = pg_internal_ingest_to_sink(path, part, n_parts, newlines_before, &sink, out);
    for (hipStream_t st : s.streams) if (st) (void)hipStreamDestroy(st);
    (void)hipSetDevice(before);
    return rc;
}

namespace {
int place_impl(const pg_reads *r, uint64_t *staging_codes, const uint32_t *staging_valid, const uint32_t *staging_lowq, int64_t staging_words,
               uint64_t *codes, uint32_t *valid, uint32_t *lowq, int64_t n_words, void *stream)
{
    const int64_t *soff = nullptr, *cstart = nullptr;
    const int64_t P = pg_internal_reads_pieces(r, &soff, &cstart);
    if (P < 0) return pg_fail(PG_EINVAL, "pg_ingest_place: not a handle of pg_ingest_fastq_device");
    if (n_words != pg_reads_n_words(r)) return pg_fail(PG_EINVAL, "pg_ingest_place: %lld output words, the stream has %lld", (long long)n_words, (long long)pg_reads_n_words(r));
    if (n_words == 0) return PG_OK;
    if (!staging_codes || !staging_valid || !codes || !valid) return pg_fail(PG_EINVAL, "pg_ingest_place: null array");
    hipStream_t s = (hipStream_t)stream;
    if (P == 0) {           // no piece, no character: padding only
        if (hipMemsetAsync(codes, 0, (size_t)n_words * 8, s) != hipSuccess || hipMemsetAsync(valid, 0, (size_t)n_words * 4, s) != hipSuccess ||
            (lowq && hipMemsetAsync(lowq, 0, (size_t)n_words * 4, s) != hipSuccess))
            return pg_fail(PG_EHIP, "pg_ingest_place: memset failed");
        return PG_OK;
    }
    // the tables ride behind the pieces in the staging array of the codes: [soff: P][cstart: P + 1]
    const int64_t tw = 2 * P + 1;
    if (staging_words < tw) return pg_fail(PG_EINVAL, "pg_ingest_place: staging arrays too small");
    int64_t *tab = (int64_t *)(staging_codes + (staging_words - tw));
    if (hipMemcpyAsync(tab, soff, (size_t)P * 8, hipMemcpyHostToDevice, s) != hipSuccess ||
        hipMemcpyAsync(tab + P, cstart, (size_t)(P + 1) * 8, hipMemcpyHostToDevice, s) != hipSuccess)
        return pg_fail(PG_EHIP, "pg_ingest_place: copy of the pieces' tables failed");
    // (the tables are the handle's own pageable vectors; the stream has nothing else to do at this point, so waiting for the two
    // copies costs nothing and the handle may be freed as soon as this call returns)
    if (hipStreamSynchronize(s) != hipSuccess) return pg_fail(PG_EHIP, "pg_ingest_place: copy of the pieces' tables failed");
    const int64_t blocks = std::min<int64_t>((n_words + PLACE_BLOCK - 1) / PLACE_BLOCK, 256 * 32);
    ingest_place_kernel<<<(int)blocks, PLACE_BLOCK, 0, s>>>(staging_codes, staging_valid, lowq ? staging_lowq : (const uint32_t *)nullptr, tab, tab + P, (int)P,
                                                            codes, valid, lowq, n_words);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return pg_fail(PG_EHIP, "pg_ingest_place: %s", hipGetErrorString(e));
    return PG_OK;
}

// pieces of -1 / -2 input: two local streams (kept pairs, skipped pairs' reads) per piece of R1 and per tail
int64_t max_pieces_pair(int64_t r1_bytes) { return 2 * (r1_bytes / piece_bytes() + 3); }
int64_t table_words_pair(int64_t r1_bytes) { return 2 * (max_pieces_pair(r1_bytes) + 1) + 8; }
}  // namespace

extern "C" int pg_ingest_place(const pg_reads *r, uint64_t *staging_codes, const uint32_t *staging_valid, int64_t staging_words,
                               uint64_t *codes, uint32_t *valid, int64_t n_words, void *stream)
{
    return place_impl(r, staging_codes, staging_valid, nullptr, staging_words, codes, valid, nullptr, n_words, stream);
}

extern "C" int64_t pg_ingest_pair_staging_words(int64_t r1_bytes, int64_t r2_bytes)
{
    if (r1_bytes < 0 || r2_bytes < 0) return pg_fail(PG_EINVAL, "pg_ingest_pair_staging_words: negative size");
    return (r1_bytes + r2_bytes) / 32 + max_pieces_pair(r1_bytes) + 64 + table_words_pair(r1_bytes);
}

extern "C" int pg_ingest_fastq_pair_device(const char *r1, const char *r2, int64_t r1_bytes, int64_t r2_bytes, uint64_t *staging_codes,
                                           uint32_t *staging_valid, uint32_t *staging_lowq, int64_t staging_words, pg_reads **out)
{
    if (!r1 || !r2 || !out || !staging_codes || !staging_valid || !staging_lowq) return pg_fail(PG_EINVAL, "pg_ingest_fastq_pair_device: bad argument");
    *out = nullptr;
    if (staging_words < pg_ingest_pair_staging_words(r1_bytes, r2_bytes))
        return pg_fail(PG_EINVAL, "pg_ingest_fastq_pair_device: staging arrays of %lld words, pg_ingest_pair_staging_words = %lld", (long long)staging_words,
                       (long long)pg_ingest_pair_staging_words(r1_bytes, r2_bytes));
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, staging_codes) != hipSuccess || at.type != hipMemoryTypeDevice)
        return pg_fail(PG_EINVAL, "pg_ingest_fastq_pair_device: the staging arrays must be device memory");
    DeviceSink s;
    s.codes = staging_codes; s.valid = staging_valid; s.lowq = staging_lowq; s.device = at.device;
    int before = 0;
    (void)hipGetDevice(&before);
    (void)hipSetDevice(s.device);
    pg_piece_sink sink{&s, staging_words - table_words_pair(r1_bytes), sink_copy};
    const int rc = pg_internal_ingest_pair_to_sink(r1, r2, &sink, out);
    for (hipStream_t st : s.streams) if (st) (void)hipStreamDestroy(st);
    (void)hipSetDevice(before);
    return rc;
}

extern "C" int pg_reads_staged_lowq(const pg_reads *r) { return pg_internal_reads_staged_lowq(r) ? 1 : 0; }

extern "C" int pg_ingest_place_pair(const pg_reads *r, uint64_t *staging_codes, const uint32_t *staging_valid, const uint32_t *staging_lowq,
                                    int64_t staging_words, uint64_t *codes, uint32_t *valid, uint32_t *lowq, int64_t n_words, void *stream)
{
    if (pg_internal_reads_staged_lowq(r) && (!lowq || !staging_lowq)) return pg_fail(PG_EINVAL, "pg_ingest_place_pair: the handle has a low-quality plane (pg_reads_staged_lowq)");
    return place_impl(r, staging_codes, staging_valid, staging_lowq, staging_words, codes, valid, pg_internal_reads_staged_lowq(r) ? lowq : nullptr, n_words, stream);
}
