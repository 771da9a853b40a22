// host.cpp -- host half of libpangaea_feat.so: FASTQ ingest into the packed read stream, barcode-run
// bookkeeping, segment planning, TNF column tables and the %g CSV cache writer.
//
// Reference behaviour reproduced here (file:line under /root/reference/src/cpptools):
//   header grammar / mode latch         count_tnf.cpp:23-52
//   interleaved producer loop           count_tnf.cpp:238-289  (line % 8 in {1,2,6})
//   paired producer loop                count_tnf.cpp:174-231  (line % 4 in {1,2}; mismatching pairs skipped)
//   row filter                          count_tnf.cpp:81
//   CSV rows                            count_tnf.cpp:293-303
// Design differs on purpose: the file is decoded through a large zlib buffer and scanned in place with
// memchr; characters go straight into 2-bit/1-bit words; a run is only a pair of stream offsets.
#include <stdarg.h>
#include <stdio.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <unistd.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#include <immintrin.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "pg_internal.h"

namespace {
thread_local char g_err[512] = "";
}

int pg_fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char *pg_last_error(void) { return g_err; }
extern "C" int pg_abi_version(void) { return PG_ABI_VERSION; }

// ------------------------------------------------------------------------------------ stream writer

namespace {

struct StreamWriter {
    std::vector<uint64_t> codes;
    std::vector<uint32_t> valid;
    std::vector<uint32_t> lower;   // bit set for a LOWER-case a c g t: a base for jellyfish, a reset for the reference's own counters
    bool any_lower = false;
    // bit set for a base (either case) whose quality character is below '?': jellyfish is run with --min-qual-char=? on
    // paired files (feature.py:76-83) and turns such bases into N, the reference's own counters never look at qualities
    std::vector<uint32_t> lowq;
    bool any_lowq = false;
    int64_t n = 0;          // characters written
    uint64_t cw = 0;        // word under construction
    uint32_t vw = 0, lw = 0, qw = 0;

    inline void put(unsigned char c)
    {
        // A C G T -> valid with code (c>>1)&3 ; everything else (N, lower case, IUPAC, '\r', separators) invalid.
        // a c g t keep their code ((c>>1)&3 is the same in both cases) and are marked in the `lower` plane.
        const bool ok = (c == 'A') | (c == 'C') | (c == 'G') | (c == 'T');
        const bool lo = (c == 'a') | (c == 'c') | (c == 'g') | (c == 't');
        const int sh = (int)(n & 31);
        if (ok | lo) cw |= (uint64_t)((c >> 1) & 3) << (2 * sh);
        if (ok) vw |= 1u << sh;
        if (lo) { lw |= 1u << sh; any_lower = true; }
        if (sh == 31) { codes.push_back(cw); valid.push_back(vw); lower.push_back(lw); lowq.push_back(qw); cw = 0; vw = 0; lw = 0; qw = 0; }
        ++n;
    }
    void put_span(const char *s, size_t len)
    {
        for (size_t i = 0; i < len; ++i) put((unsigned char)s[i]);
    }
    // a read with its quality line (paired files): bases below '?' are marked (a missing quality character marks nothing)
    void put_span_q(const char *s, size_t len, const char *q, size_t qlen)
    {
        for (size_t i = 0; i < len; ++i) {
            const unsigned char c = (unsigned char)s[i];
            const bool base = (c == 'A') | (c == 'C') | (c == 'G') | (c == 'T') | (c == 'a') | (c == 'c') | (c == 'g') | (c == 't');
            if (base && i < qlen && (unsigned char)q[i] < (unsigned char)'?') { qw |= 1u << (int)(n & 31); any_lowq = true; }
            put(c);
        }
    }
    void finish()
    {
        if (n & 31) { codes.push_back(cw); valid.push_back(vw); lower.push_back(lw); lowq.push_back(qw); cw = 0; vw = 0; lw = 0; qw = 0; }
        size_t words = codes.size();
        size_t padded = (words + PG_WORD_ALIGN - 1) / PG_WORD_ALIGN * PG_WORD_ALIGN;
        if (padded == 0) padded = PG_WORD_ALIGN;
        codes.resize(padded, 0);
        valid.resize(padded, 0);
        if (any_lower) lower.resize(padded, 0); else std::vector<uint32_t>().swap(lower);
        if (any_lowq) lowq.resize(padded, 0); else std::vector<uint32_t>().swap(lowq);
    }
};

// a whole file in memory (malloc'd: no zero fill of hundreds of MB before they are overwritten)
struct FileBuf {
    char *p = nullptr;
    size_t n = 0;
    FileBuf() = default;
    FileBuf(const FileBuf &) = delete;
    FileBuf &operator=(const FileBuf &) = delete;
    ~FileBuf() { free(p); }
    const char *data() const { return p ? p : ""; }
    size_t size() const { return n; }
    size_t cap_ = 0;
    // 2 MiB-aligned and advised for transparent huge pages: hundreds of MB are then faulted in and torn down as a few
    // hundred huge pages instead of ~10^5 small ones (first touch during the read, munmap at the end)
    bool reserve_exact(size_t cap)
    {
        if (cap <= cap_) return true;
        const size_t align = (size_t)2 << 20;
        const size_t bytes = (cap + align - 1) / align * align;
        void *q = nullptr;
        if (posix_memalign(&q, align, bytes) != 0) return false;
#ifdef MADV_HUGEPAGE
        madvise(q, bytes, MADV_HUGEPAGE);
#endif
        if (p && n) memcpy(q, p, n);
        free(p);
        p = (char *)q;
        cap_ = bytes;
        return true;
    }
};

int ingest_threads();
template <typename F> void run_threads(int T, F &&f);

// whole file into memory.  Plain files are read straight into a buffer of the file's size; gzip files (magic 1f 8b) go
// through zlib -- so both kinds are accepted transparently, as gzstream's gzopen does for the reference.
int slurp(const char *path, FileBuf &out)
{
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return pg_fail(PG_EIO, "cannot open %s", path);
    unsigned char magic[2] = {0, 0};
    const ssize_t got_magic = pread(fd, magic, 2, 0);
    const bool gz = got_magic == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
    struct stat st;
    const bool regular = fstat(fd, &st) == 0 && S_ISREG(st.st_mode);
    if (!gz && regular) {
        const size_t size = (size_t)st.st_size;
        if (!out.reserve_exact(size)) { close(fd); return pg_fail(PG_ENOMEM, "out of memory reading %s", path); }
        // page-cache (or NVMe) reads scale with threads: every thread preads its own slice
        const int T = size >= ((size_t)8 << 20) ? ingest_threads() : 1;
        std::vector<char> bad(T, 0);
        run_threads(T, [&](int t) {
            size_t a = size * (size_t)t / T, b = size * (size_t)(t + 1) / T;
            while (a < b) {
                const ssize_t got = pread(fd, out.p + a, b - a, (off_t)a);
                if (got <= 0) { bad[t] = 1; return; }
                a += (size_t)got;
            }
        });
        close(fd);
        for (char x : bad) if (x) return pg_fail(PG_EIO, "read error in %s", path);
        out.n = size;
        return PG_OK;
    }
    close(fd);
    // gzip (or a pipe): zlib reads both, as the reference's gzstream does
    gzFile f = gzopen(path, "rb");
    if (!f) return pg_fail(PG_EIO, "cannot open %s", path);
    gzbuffer(f, 1 << 22);
    size_t cap = (size_t)1 << 24;
    if (!out.reserve_exact(cap)) { gzclose(f); return pg_fail(PG_ENOMEM, "out of memory reading %s", path); }
    size_t n = 0;
    for (;;) {
        if (cap - n < ((size_t)1 << 22)) {
            cap *= 2;
            out.n = n;
            if (!out.reserve_exact(cap)) { gzclose(f); return pg_fail(PG_ENOMEM, "out of memory reading %s", path); }
        }
        int got = gzread(f, out.p + n, (unsigned)std::min<size_t>(cap - n, (size_t)1 << 30));
        if (got < 0) { gzclose(f); return pg_fail(PG_EIO, "read error in %s", path); }
        if (got == 0) break;
        n += (size_t)got;
    }
    gzclose(f);
    out.n = n;
    return PG_OK;
}

// getline-style cursor over an in-memory file
struct Lines {
    const char *p, *end;
    explicit Lines(const FileBuf &s) : p(s.data()), end(s.data() + s.size()) {}
    Lines(const char *b, size_t n) : p(b), end(b + n) {}
    bool next(const char *&b, size_t &len)
    {
        if (p >= end) return false;
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        b = p;
        len = nl ? (size_t)(nl - p) : (size_t)(end - p);
        p = nl ? nl + 1 : end;
        return true;
    }
};

enum { MODE_UNSET = 0, MODE_10X = 1, MODE_STLFR = 2 };
constexpr size_t NPOS = (size_t)-1;

size_t find_chr(const char *s, size_t n, char c, size_t from)
{
    if (from >= n) return NPOS;
    const void *q = memchr(s + from, c, n - from);
    return q ? (size_t)((const char *)q - s) : NPOS;
}

size_t find_bxz(const char *s, size_t n)
{
    for (size_t i = 0; i + 4 <= n; ++i)
        if (s[i] == 'B' && s[i + 1] == 'X' && s[i + 2] == ':' && s[i + 3] == 'Z') return i;
    return NPOS;
}

struct Span { size_t b = 0, n = 0; };

// clipped sub-string (pos, count); false where std::string::substr would throw
bool clip(size_t size, size_t pos, size_t count, Span &out)
{
    if (pos > size) return false;
    out.b = pos;
    out.n = count < size - pos ? count : size - pos;
    return true;
}

// header line -> spans of read name and barcode.  The first header that contains "BX:Z" (else '#')
// fixes the grammar for the rest of the input.
bool header_fields(const char *s, size_t n, int &mode, Span &name, Span &bc)
{
    if (mode == MODE_UNSET) {
        if (find_bxz(s, n) != NPOS) mode = MODE_10X;
        else if (find_chr(s, n, '#', 0) != NPOS) mode = MODE_STLFR;
    }
    name = Span(); bc = Span();
    if (mode == MODE_STLFR) {
        const size_t p1 = find_chr(s, n, '#', 0);
        const size_t p2 = find_chr(s, n, '/', p1 + 1);       // p1 may be NPOS: unsigned wrap-around as in the reference
        if (!clip(n, 0, p1, name)) return false;
        if (!clip(n, p1 + 1, p2 - p1 - 1, bc)) return false;
        if (bc.n == 5 && memcmp(s + bc.b, "0_0_0", 5) == 0) bc.n = 0;
    } else {
        size_t e = NPOS;
        for (size_t i = 0; i < n; ++i)
            if (s[i] == ' ' || s[i] == '\t' || s[i] == '\r' || s[i] == '\n') { e = i; break; }
        clip(n, 0, e, name);
        const size_t p1 = find_bxz(s, n);
        if (p1 != NPOS) {
            const size_t p2 = find_chr(s, n, '-', p1 + 5);
            if (!clip(n, p1 + 5, p2 - p1 - 5, bc)) return false;
        }
    }
    return true;
}

}  // namespace

struct pg_reads {
    StreamWriter st;                   // the serial loops build the stream here ...
    void *huge_c = nullptr, *huge_v = nullptr;   // ... the threaded one in uninitialised huge-page arrays
    uint64_t *codes_w = nullptr;
    uint32_t *valid_w = nullptr;
    uint32_t *lower_w = nullptr;                 // NULL unless the input has lower-case bases
    uint32_t *lowq_w = nullptr;                  // NULL unless paired input has bases of quality below '?'
    std::vector<uint32_t> lower_plane;           // (the threaded paths keep the rare plane here)
    std::vector<uint32_t> lowq_plane;            // (... and the threaded paired path its low-quality plane)
    int64_t n_words = 0, n_chars = 0;
    std::vector<int64_t> run_off;      // [n_runs + 1]
    std::vector<std::string> run_name;
    int64_t n_pairs = 0, n_unpaired = 0;
    int mode = MODE_UNSET;
    // pg_ingest_fastq_device: no host arrays; piece p's packed characters lie at word piece_soff[p] of the staging arrays and
    // belong at characters [piece_cstart[p], piece_cstart[p + 1]) of the stream
    std::vector<int64_t> piece_soff, piece_cstart;
    bool staged_lowq = false;                    // (... of paired input: the staging arrays hold a low-quality plane to place as well)

    pg_reads() = default;
    pg_reads(const pg_reads &) = delete;
    pg_reads &operator=(const pg_reads &) = delete;
    ~pg_reads() { free(huge_c); free(huge_v); }
    void seal_serial()
    {
        st.finish();
        codes_w = st.codes.data(); valid_w = st.valid.data();
        lower_w = st.any_lower ? st.lower.data() : nullptr;
        lowq_w = st.any_lowq ? st.lowq.data() : nullptr;
        n_words = (int64_t)st.codes.size(); n_chars = st.n;
    }
    static size_t padded_words(int64_t total_chars)
    {
        size_t words = (size_t)((total_chars + 31) / 32);
        size_t padded = (words + PG_WORD_ALIGN - 1) / PG_WORD_ALIGN * PG_WORD_ALIGN;
        return padded ? padded : PG_WORD_ALIGN;
    }
    void set_stream_size(int64_t total_chars) { n_words = (int64_t)padded_words(total_chars); n_chars = total_chars; }
    bool alloc_stream(int64_t total_chars)
    {
        size_t padded = padded_words(total_chars);
        auto grab = [](size_t bytes) -> void * {
            const size_t align = bytes >= ((size_t)4 << 20) ? (size_t)2 << 20 : 64;
            void *q = nullptr;
            if (posix_memalign(&q, align, (bytes + align - 1) / align * align) != 0) return nullptr;
#ifdef MADV_HUGEPAGE
            if (align > 64) madvise(q, (bytes + align - 1) / align * align, MADV_HUGEPAGE);
#endif
            return q;
        };
        huge_c = grab(padded * sizeof(uint64_t));
        huge_v = grab(padded * sizeof(uint32_t));
        if (!huge_c || !huge_v) return false;
        codes_w = (uint64_t *)huge_c; valid_w = (uint32_t *)huge_v;
        n_words = (int64_t)padded; n_chars = total_chars;
        return true;
    }
};

// ------------------------------------------------------------------------------------ parallel interleaved ingest
//
// Same result as the serial loop below.  The file is never held in memory: T threads stream their byte ranges through
// small block buffers (which stay in cache), twice:
//   A. newline counts per range -> the line number at every range start, hence each thread's first 8-line unit
//   B. per unit: header -> barcode, both sequences packed (2-bit codes + validity) into a thread-local stream, barcode
//      change points noted
// then   C. the local streams are shifted into place in the (uninitialised, huge-page) output arrays -- atomic OR only
//           on the words two threads share --
//        D. and the change points are stitched in thread order (the append-then-compare rule).
// The grammar latch (first header with "BX:Z", else '#') is found first, from the start of the file, so that every
// header is parsed exactly as the sequential latch would.
namespace {

int g_ingest_threads = 0;      // 0 = hardware concurrency (capped)

int ingest_threads()
{
    if (g_ingest_threads > 0) return g_ingest_threads;
    if (const char *e = getenv("PG_INGEST_THREADS")) { int v = atoi(e); if (v > 0) return v; }
    unsigned hc = std::thread::hardware_concurrency();
    return (int)std::min<unsigned>(hc ? hc : 1, 32);
}

template <typename F> void run_threads(int T, F &&f)
{
    std::vector<std::thread> th;
    th.reserve(T);
    for (int t = 1; t < T; ++t) th.emplace_back([&f, t] { f(t); });
    f(0);
    for (auto &x : th) x.join();
}

struct PhaseTimer {
    bool on = getenv("PG_INGEST_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void lap(const char *what)
    {
        if (!on) return;
        auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[pg_ingest] %-10s %7.1f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

struct Latch { uint64_t unit; int mode; };     // first unit whose header fixes the grammar (UINT64_MAX: none)

// what a byte range of a file inherits from the bytes before it (pg_ingest_fastq_shard); a whole file uses the defaults
struct ShardCtx {
    uint64_t unit_base = 0;       // index in the file of the range's first unit
    std::string last;             // barcode of the pair before the range (the comparison state at entry)
    bool trailing = true;         // the range ends at the end of the file: enqueue the trailing accumulator
};

uint64_t count_newlines_avx2(const char *p, size_t n);
bool simd_ok();

struct UnitLines { const char *p[8]; size_t n[8]; int count; size_t off; };   // up to 8 lines; off = file offset of the unit

// sequential reader of file bytes [begin, end) through one block buffer; hands out lines and 8-line units
class UnitReader {
public:
    // (the buffer has SLACK bytes behind its logical capacity: 32-byte vector loads may start at any byte of a line)
    static constexpr size_t SLACK = 64;
    UnitReader(int fd, size_t begin, size_t end, size_t block) : fd_(fd), base_(begin), end_(end), cap_(block), buf_(block + SLACK) {}
    bool io_error() const { return io_error_; }
    size_t offset() const { return base_ + cur_; }
    // the rest of the current line (up to and including its '\n'); false if the range ends first
    bool skip_line()
    {
        for (;;) {
            if (cur_ < have_) {
                const char *q = (const char *)memchr(buf_.data() + cur_, '\n', have_ - cur_);
                if (q) { cur_ = (size_t)(q - buf_.data()) + 1; return true; }
                cur_ = have_;
            }
            if (eof_ || !more()) return false;
        }
    }
    uint64_t count_newlines()
    {
        uint64_t c = 0;
        for (;;) {
            const char *p = buf_.data() + cur_, *e = buf_.data() + have_;
            if (simd_) c += count_newlines_avx2(p, (size_t)(e - p));
            else while (p < e) { const char *q = (const char *)memchr(p, '\n', (size_t)(e - p)); if (!q) break; ++c; p = q + 1; }
            cur_ = have_;
            if (eof_ || !more()) return c;
        }
    }
    // the next `want` (<= 8) lines as one unit: an interleaved pair (8) or a record of one file (4)
    bool next(UnitLines &u, int want = 8)
    {
        for (;;) {
            size_t q = cur_;
            int k = 0;
            if (simd_) q = scan_lines_avx2(u, k, want);
            else
            while (k < want && q < have_) {
                const char *b = buf_.data() + q;
                const char *nl = (const char *)memchr(b, '\n', have_ - q);
                if (!nl) break;
                u.p[k] = b; u.n[k] = (size_t)(nl - b); ++k;
                q = (size_t)(nl - buf_.data()) + 1;
            }
            if (k < want) {
                // (more() moves the unread tail to the front of the buffer even when it finds nothing further to read: the
                // positions collected so far are stale either way, so the unit is scanned again)
                if (!eof_) { more(); if (io_error_) return false; continue; }
                if (q < have_) { u.p[k] = buf_.data() + q; u.n[k] = have_ - q; ++k; q = have_; }   // unterminated last line
                if (k == 0) return false;
            }
            u.count = k; u.off = base_ + cur_;
            cur_ = q;
            return true;
        }
    }
private:
    // up to `want` lines from cur_ on, their newlines found 32 bytes at a time (one pass over the unit instead of a memchr call
    // per line); returns where the next line starts
    __attribute__((target("avx2"))) size_t scan_lines_avx2(UnitLines &u, int &k, int want) const
    {
        const char *buf = buf_.data();
        const __m256i nl = _mm256_set1_epi8('\n');
        size_t start = cur_;
        for (size_t pos = cur_; pos < have_ && k < want; pos += 32) {
            uint32_t m = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i *)(buf + pos)), nl));
            if (have_ - pos < 32) m &= (1u << (have_ - pos)) - 1u;          // (bytes behind the data: the slack, stale)
            while (m && k < want) {
                const size_t at = pos + (size_t)__builtin_ctz(m);
                u.p[k] = buf + start; u.n[k] = at - start; ++k;
                start = at + 1;
                m &= m - 1;
            }
        }
        return start;
    }
    // keep the unread tail, read on; false at the end of the range (or on error)
    bool more()
    {
        if (cur_ > 0) { memmove(buf_.data(), buf_.data() + cur_, have_ - cur_); base_ += cur_; have_ -= cur_; cur_ = 0; }
        if (have_ == cap_) { cap_ *= 2; buf_.resize(cap_ + SLACK); }
        const size_t at = base_ + have_;
        const size_t want = std::min(cap_ - have_, end_ > at ? end_ - at : 0);
        if (want == 0) { eof_ = true; return false; }
        const ssize_t got = pread(fd_, buf_.data() + have_, want, (off_t)at);
        if (got <= 0) { eof_ = true; io_error_ = got < 0 || at < end_; return false; }
        have_ += (size_t)got;
        return true;
    }
    int fd_;
    size_t base_, end_;                 // file offset of buf_[0]; end of the range
    size_t cap_;                        // bytes of buf_ that hold file data at most (SLACK more are allocated)
    std::vector<char> buf_;
    size_t have_ = 0, cur_ = 0;
    bool eof_ = false, io_error_ = false;
    const bool simd_ = simd_ok();
};

// block size of the readers (PG_INGEST_BLOCK overrides: tests use tiny blocks to exercise refills and growth)
size_t reader_block(size_t dflt)
{
    if (const char *e = getenv("PG_INGEST_BLOCK")) { long v = atol(e); if (v >= 16) return (size_t)v; }
    return dflt;
}

inline int mode_of(const Latch &L, uint64_t unit) { return unit < L.unit ? (int)MODE_UNSET : L.mode; }

// first R1 header from the start of the file that fixes the grammar
Latch find_latch(int fd, size_t size)
{
    UnitReader rd(fd, 0, size, reader_block((size_t)1 << 16));
    UnitLines u;
    for (uint64_t i = 0; rd.next(u); ++i) {
        if (find_bxz(u.p[0], u.n[0]) != NPOS) return Latch{i, MODE_10X};
        if (find_chr(u.p[0], u.n[0], '#', 0) != NPOS) return Latch{i, MODE_STLFR};
    }
    return Latch{UINT64_MAX, MODE_UNSET};
}

// ---- character packing, 8 or 32 at a time.  A C G T -> valid, code (c >> 1) & 3; anything else -> invalid, code 0.

// 8 characters (little-endian in x) -> 16 code bits + 8 validity bits, plain 64-bit arithmetic
inline void pack8(uint64_t x, uint64_t &cb, uint32_t &vb)
{
    const uint64_t L7 = 0x7F7F7F7F7F7F7F7FULL;
    auto is_zero = [&](uint64_t t) { return ~(((t & L7) + L7) | t | L7); };        // 0x80 in exactly the zero bytes
    const uint64_t m = is_zero(x ^ 0x4141414141414141ULL) | is_zero(x ^ 0x4343434343434343ULL) |
                       is_zero(x ^ 0x4747474747474747ULL) | is_zero(x ^ 0x5454545454545454ULL);
    const uint64_t ones = m >> 7;                                                    // 0x01 per valid byte
    vb = (uint32_t)((ones * 0x0102040810204080ULL) >> 56);                           // bit j = byte j
    uint64_t y = (x >> 1) & 0x0303030303030303ULL & (ones * 0xFF);
    y = (y | (y >> 6)) & 0x000F000F000F000FULL;
    y = (y | (y >> 12)) & 0x000000FF000000FFULL;
    cb = (y | (y >> 24)) & 0xFFFF;
}

// 32 characters -> 64 code bits + 32 validity bits
__attribute__((target("avx2,bmi2"))) inline void pack32_avx2(const char *s, uint64_t &cb, uint32_t &vb)
{
    const __m256i x = _mm256_loadu_si256((const __m256i *)s);
    const __m256i ok = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(x, _mm256_set1_epi8('A')), _mm256_cmpeq_epi8(x, _mm256_set1_epi8('C'))),
                                       _mm256_or_si256(_mm256_cmpeq_epi8(x, _mm256_set1_epi8('G')), _mm256_cmpeq_epi8(x, _mm256_set1_epi8('T'))));
    vb = (uint32_t)_mm256_movemask_epi8(ok);
    const __m256i xm = _mm256_and_si256(x, ok);
    const uint64_t M = 0x0606060606060606ULL;
    cb = _pext_u64((uint64_t)_mm256_extract_epi64(xm, 0), M) | _pext_u64((uint64_t)_mm256_extract_epi64(xm, 1), M) << 16 |
         _pext_u64((uint64_t)_mm256_extract_epi64(xm, 2), M) << 32 | _pext_u64((uint64_t)_mm256_extract_epi64(xm, 3), M) << 48;
}

__attribute__((target("avx2"))) uint64_t count_newlines_avx2(const char *p, size_t n)
{
    uint64_t c = 0;
    const __m256i nl = _mm256_set1_epi8('\n');
    size_t i = 0;
    for (; i + 32 <= n; i += 32)
        c += (uint64_t)__builtin_popcount((unsigned)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i *)(p + i)), nl)));
    for (; i < n; ++i) c += p[i] == '\n';
    return c;
}

bool simd_ok()
{
    static const bool ok = [] { __builtin_cpu_init(); return __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2") && !getenv("PG_INGEST_NO_SIMD"); }();
    return ok;
}

// lower-case a c g t of a block of `count` <= 32 characters that is not all upper-case bases (the rare path): their mask,
// and their codes OR-ed into cb (a base keeps its code in both cases: (c >> 1) & 3)
inline uint32_t lower_bases(const char *s, int count, uint64_t &cb)
{
    uint32_t lb = 0;
    for (int i = 0; i < count; ++i) {
        const unsigned char c = (unsigned char)s[i];
        if ((c == 'a') | (c == 'c') | (c == 'g') | (c == 't')) { lb |= 1u << i; cb |= (uint64_t)((c >> 1) & 3) << (2 * i); }
    }
    return lb;
}

// 32 quality characters -> bit j set iff (unsigned char)q[j] < '?'
__attribute__((target("avx2"))) inline uint32_t lowq32_avx2(const char *q)
{
    const __m256i x = _mm256_loadu_si256((const __m256i *)q);
    const uint32_t below = (uint32_t)_mm256_movemask_epi8(_mm256_cmpgt_epi8(_mm256_set1_epi8('?'), x));      // signed compare ...
    return below & ~(uint32_t)_mm256_movemask_epi8(x);                                                        // ... bytes >= 128 are not below
}

struct LowerMask { int64_t pos; uint32_t mask; };     // lower-case bases at characters pos .. pos + 31 of the thread's stream

struct LocalStream {          // one thread's characters, packed from bit 0
    std::vector<uint64_t> codes;
    std::vector<uint32_t> valid;
    std::vector<LowerMask> lower;         // sparse: sequencer reads are upper case
    std::vector<uint32_t> lowq;           // paired files only (with_q): bases of quality below '?' -- dense, those are common
    bool with_q = false, any_q = false;
    bool slack32 = false;                 // every line handed to put_line may be read 32 bytes past its end (UnitReader::SLACK)
    int64_t n = 0;
    uint64_t cw = 0; uint32_t vw = 0, qw = 0;
    const bool simd = simd_ok();
    // `count` (1..32) characters; bits above them are zero
    inline void put_bits(uint64_t cb, uint32_t vb, int count, uint32_t qb = 0)
    {
        const int sh = (int)(n & 31);
        cw |= cb << (2 * sh);
        vw |= vb << sh;
        qw |= qb << sh;
        n += count;
        if (sh + count >= 32) {
            codes.push_back(cw); valid.push_back(vw);
            if (with_q) lowq.push_back(qw);
            if (sh) { cw = cb >> (2 * (32 - sh)); vw = vb >> (32 - sh); qw = qb >> (32 - sh); } else { cw = 0; vw = 0; qw = 0; }
        }
    }
    // positions j < count of a block that starts at character i of its line whose quality character is below '?'
    inline uint32_t lowq_bits(const char *q, size_t qlen, size_t i, int count) const
    {
        if (count == 32 && simd && i + 32 <= qlen) return lowq32_avx2(q + i);
        uint32_t b = 0;
        for (int j = 0; j < count && i + (size_t)j < qlen; ++j) b |= (uint32_t)((unsigned char)q[i + (size_t)j] < (unsigned char)'?') << j;
        return b;
    }
    // a sequence line with its quality line (StreamWriter::put_span_q + the separator): a BASE (either case) of quality below
    // '?' is marked in the lowq plane; a missing quality character marks nothing
    void put_line_q(const char *s, size_t len, const char *q, size_t qlen)
    {
        size_t i = 0;
        uint64_t cb; uint32_t vb;
        auto block = [&](int count, int chars) {       // `chars` characters of the line (+ the separator when count > chars)
            uint32_t lb = 0;
            if (vb != (chars == 32 ? 0xffffffffu : (1u << chars) - 1u) && chars) {
                lb = lower_bases(s + i, chars, cb);
                if (lb) lower.push_back(LowerMask{n, lb});
            }
            const uint32_t qb = lowq_bits(q, qlen, i, chars) & (vb | lb);
            any_q |= qb != 0;
            put_bits(cb, vb, count, qb);
        };
        if (simd) for (; i + 32 <= len; i += 32) { pack32_avx2(s + i, cb, vb); block(32, 32); }
        for (; i + 8 <= len; i += 8) { uint64_t x; memcpy(&x, s + i, 8); pack8(x, cb, vb); block(8, 8); }
        uint64_t x = 0;
        memcpy(&x, s + i, len - i);                  // < 8 characters, zero padded (zero bytes are invalid), then the separator
        pack8(x, cb, vb);
        block((int)(len - i) + 1, (int)(len - i));
    }
    // one sequence line followed by the separator the reference appends ('N': invalid)
    void put_line(const char *s, size_t len)
    {
        size_t i = 0;
        uint64_t cb; uint32_t vb;
        if (simd) for (; i + 32 <= len; i += 32) {
            pack32_avx2(s + i, cb, vb);
            if (vb != 0xffffffffu) note_lower(s + i, 32, cb);
            put_bits(cb, vb, 32);
        }
        if (simd && slack32) {
            // the last < 32 characters and the separator in one step: the 32-byte load runs past the line (into the next one, or
            // into the slack behind the reader's buffer) and what it finds there is masked off
            const int rem = (int)(len - i);
            pack32_avx2(s + i, cb, vb);
            const uint32_t keep = (1u << rem) - 1u;
            vb &= keep;
            cb &= (1ull << (2 * rem)) - 1ull;
            if (vb != keep) note_lower(s + i, rem, cb);
            put_bits(cb, vb, rem + 1);
            return;
        }
        for (; i + 8 <= len; i += 8) {
            uint64_t x; memcpy(&x, s + i, 8); pack8(x, cb, vb);
            if (vb != 0xffu) note_lower(s + i, 8, cb);
            put_bits(cb, vb, 8);
        }
        uint64_t x = 0;
        memcpy(&x, s + i, len - i);                  // < 8 characters, zero padded (zero bytes are invalid), then the separator
        pack8(x, cb, vb);
        if (len - i) note_lower(s + i, (int)(len - i), cb);
        put_bits(cb, vb, (int)(len - i) + 1);
    }
    inline void note_lower(const char *s, int count, uint64_t &cb)
    {
        const uint32_t lb = lower_bases(s, count, cb);
        if (lb) lower.push_back(LowerMask{n, lb});
    }
    void finish() { if (n & 31) { codes.push_back(cw); valid.push_back(vw); if (with_q) lowq.push_back(qw); cw = 0; vw = 0; qw = 0; } }
    void reset() { codes.clear(); valid.clear(); lower.clear(); lowq.clear(); any_q = false; n = 0; cw = 0; vw = 0; qw = 0; }     // (capacity stays)
};

struct Change { int64_t end_pos; std::string prev; };       // a run ends at end_pos (thread-local); it carries `prev`

struct ThreadOut {                // what a piece (a byte range of the file) leaves behind besides its packed characters
    int64_t n = 0;                // its characters
    int64_t soff = 0;             // (sink form) word offset of its local stream in the staging arrays
    std::vector<LowerMask> lower;
    std::vector<Change> changes;
    bool any = false;             // saw a complete pair
    std::string first, last;      // barcodes of the first and the last complete pair
    int64_t first_end = 0;        // local position after the first complete pair
    int64_t pairs = 0;
    uint64_t bad_unit = UINT64_MAX;
    bool io_error = false;
};

// bytes [A, B) of the open file: A is the start of a unit, B the start of a unit or the end of the file.
// Without a sink the range is cut into T pieces, one per thread, whose local streams are shifted into place in host arrays
// (phase C).  With a sink (pg_ingest_fastq_device) it is cut into pieces of PG_INGEST_PIECE bytes that the threads take from a
// queue; a thread hands the local stream of a finished piece to the sink (a copy to the GPU) while the others go on parsing,
// phase C is left to the device (pg_ingest_place) and R keeps the pieces' offsets instead of the arrays.
int ingest_interleaved_range(int fd, size_t A, size_t B, const char *path, pg_reads *R, int T, const Latch &L, const ShardCtx &ctx,
                             const pg_piece_sink *sink = nullptr)
{
    PhaseTimer tm;
    const size_t n = B - A;
    const size_t block = reader_block((size_t)1 << 20);
    int P = T;
    if (sink) {
        size_t piece = (size_t)16 << 20;
        if (const char *e = getenv("PG_INGEST_PIECE")) { long v = atol(e); if (v >= 64) piece = (size_t)v; }
        P = (int)std::max<size_t>(1, std::min<size_t>((n + piece - 1) / piece, (size_t)1 << 20));
    }
    std::vector<size_t> rb(P + 1);
    for (int t = 0; t <= P; ++t) rb[t] = A + (size_t)((unsigned __int128)n * (unsigned)t / (unsigned)P);
    std::atomic<int> next_piece{0};
    // every thread its own piece (P == T), or pieces from the queue
    auto for_pieces = [&](auto &&body) {
        next_piece = 0;
        run_threads(T, [&](int worker) {
            if (!sink) { body(worker, worker); return; }
            for (int p; (p = next_piece.fetch_add(1, std::memory_order_relaxed)) < P;) body(p, worker);
        });
    };
    // ---- A. newlines per piece; does the piece start at the start of a line?
    std::vector<uint64_t> nl(P, 0);
    std::vector<char> at_line_start(P, 1), io_bad(P, 0);
    for_pieces([&](int t, int) {
        if (t > 0 && rb[t] > A) { char c = 0; if (pread(fd, &c, 1, (off_t)(rb[t] - 1)) != 1) io_bad[t] = 1; at_line_start[t] = c == '\n'; }
        UnitReader rd(fd, rb[t], rb[t + 1], block);
        nl[t] = rd.count_newlines();
        if (rd.io_error()) io_bad[t] = 1;
    });
    for (char x : io_bad) if (x) return pg_fail(PG_EIO, "read error in %s", path);
    std::vector<uint64_t> nl_before(P + 1, 0);
    for (int t = 0; t < P; ++t) nl_before[t + 1] = nl_before[t] + nl[t];
    tm.lap("lines");
    // ---- B. parse + pack into local streams
    std::vector<ThreadOut> out(P);
    std::vector<LocalStream> local(sink ? T : P);            // one per piece, or (sink) one per thread, reused
    std::atomic<int64_t> staged{0};
    std::atomic<int> sink_rc{0};
    for_pieces([&](int t, int worker) {
        ThreadOut &o = out[t];
        LocalStream &st = local[sink ? worker : t];
        if (rb[t] >= rb[t + 1]) return;
        st.reset();
        st.slack32 = true;
        UnitReader rd(fd, rb[t], B, block);
        uint64_t line = nl_before[t];                      // index (from A) of the first line that starts in this range
        if (!at_line_start[t]) { if (!rd.skip_line()) { o.io_error = rd.io_error(); return; } ++line; }
        for (uint64_t skip = (8 - line % 8) % 8; skip; --skip, ++line)
            if (!rd.skip_line()) { o.io_error = rd.io_error(); return; }
        uint64_t unit = ctx.unit_base + line / 8;
        st.codes.reserve((rb[t + 1] - rb[t]) / 64 + 64);
        st.valid.reserve((rb[t + 1] - rb[t]) / 64 + 64);
        UnitLines u;
        while (rd.offset() < rb[t + 1] && rd.next(u)) {
            int mode = mode_of(L, unit);
            Span nm, bc;
            if (!header_fields(u.p[0], u.n[0], mode, nm, bc)) { o.bad_unit = unit; return; }
            if (u.count >= 2) st.put_line(u.p[1], u.n[1]);
            if (u.count >= 6) {
                st.put_line(u.p[5], u.n[5]);
                ++o.pairs;
                const char *b = u.p[0] + bc.b;
                if (!o.any) { o.any = true; o.first.assign(b, bc.n); o.first_end = st.n; o.last = o.first; }
                else if (bc.n != o.last.size() || (bc.n && memcmp(b, o.last.data(), bc.n) != 0)) {
                    o.changes.push_back(Change{st.n, o.last});
                    o.last.assign(b, bc.n);
                }
            }
            ++unit;
        }
        o.io_error = rd.io_error();
        st.finish();
        o.n = st.n;
        o.lower = std::move(st.lower);
        st.lower.clear();
        if (sink && !o.io_error) {
            const int64_t nw = (int64_t)st.codes.size();
            o.soff = staged.fetch_add(nw, std::memory_order_relaxed);
            if (nw == 0) return;
            int rc = o.soff + nw <= sink->capacity_words ? sink->copy(sink->ctx, worker, o.soff, st.codes.data(), st.valid.data(), nullptr, nw) : PG_EINVAL;
            if (rc) { int zero = 0; sink_rc.compare_exchange_strong(zero, rc); }
        }
    });
    uint64_t first_bad = UINT64_MAX;
    for (int t = 0; t < P; ++t) {
        if (out[t].io_error) return pg_fail(PG_EIO, "read error in %s", path);
        first_bad = std::min(first_bad, out[t].bad_unit);
    }
    if (first_bad != UINT64_MAX)
        return pg_fail(PG_EFORMAT, "%s line %llu: header ends inside its BX:Z tag (the reference aborts here)", path, (unsigned long long)(first_bad * 8 + 1));
    if (sink_rc.load() == PG_EINVAL && staged.load() > sink->capacity_words)
        return pg_fail(PG_EINVAL, "staging arrays of %lld words are too small for %s (pg_ingest_staging_words)", (long long)sink->capacity_words, path);
    if (sink_rc.load()) return sink_rc.load();                 // (the sink recorded its message)
    tm.lap(sink ? "parse+copy" : "parse+pack");
    // ---- C. place the local streams
    std::vector<int64_t> cstart(P + 1, 0);
    for (int t = 0; t < P; ++t) cstart[t + 1] = cstart[t] + out[t].n;
    const int64_t total = cstart[P];
    if (sink) {                                       // the device does it: pg_ingest_place
        R->set_stream_size(total);
        R->piece_soff.resize(P);
        for (int t = 0; t < P; ++t) R->piece_soff[t] = out[t].soff;
        R->piece_cstart = cstart;
    } else {
        if (!R->alloc_stream(total)) return pg_fail(PG_ENOMEM, "out of memory while ingesting %s", path);
        uint64_t *gc = R->codes_w; uint32_t *gv = R->valid_w;
        for (int64_t w = (total + 31) >> 5; w < R->n_words; ++w) { gc[w] = 0; gv[w] = 0; }
        for (int t = 0; t < P; ++t)
            if (out[t].n) {
                const int64_t a = cstart[t] >> 5, b = (cstart[t + 1] - 1) >> 5;
                gc[a] = 0; gv[a] = 0; gc[b] = 0; gv[b] = 0;
            }
        run_threads(T, [&](int t) {
            const LocalStream &ls = local[t];
            if (out[t].n == 0) return;
            const int64_t w0 = cstart[t] >> 5, w1 = (cstart[t + 1] - 1) >> 5;
            const int sh = (int)(cstart[t] & 31);
            const int64_t nw = (int64_t)ls.codes.size();
            for (int64_t g = w0; g <= w1; ++g) {
                const int64_t i = g - w0;
                uint64_t c = i < nw ? ls.codes[i] << (2 * sh) : 0;
                uint32_t v = i < nw ? ls.valid[i] << sh : 0;
                if (sh && i > 0) { c |= ls.codes[i - 1] >> (64 - 2 * sh); v |= ls.valid[i - 1] >> (32 - sh); }
                if (g == w0 || g == w1) { __atomic_fetch_or(&gc[g], c, __ATOMIC_RELAXED); __atomic_fetch_or(&gv[g], v, __ATOMIC_RELAXED); }
                else { gc[g] = c; gv[g] = v; }
            }
        });
    }
    bool any_lower = false;
    for (int t = 0; t < P; ++t) any_lower |= !out[t].lower.empty();
    if (any_lower) {                                  // rare: a dense plane, bits set from the pieces' sparse lists
        R->lower_plane.assign((size_t)R->n_words, 0u);
        for (int t = 0; t < P; ++t)
            for (const LowerMask &m : out[t].lower) {
                const int64_t pos = cstart[t] + m.pos;
                const int sh = (int)(pos & 31);
                R->lower_plane[(size_t)(pos >> 5)] |= m.mask << sh;
                if (sh && (m.mask >> (32 - sh))) R->lower_plane[(size_t)(pos >> 5) + 1] |= m.mask >> (32 - sh);
            }
        R->lower_w = R->lower_plane.data();
    }
    tm.lap("place");
    // ---- D. runs: the first complete pair of a piece is compared with the last one of the pieces before it
    R->mode = L.mode;
    R->run_off.push_back(0);
    const std::string *last = &ctx.last;
    for (int t = 0; t < P; ++t) {
        R->n_pairs += out[t].pairs;
        if (!out[t].any) continue;
        if (out[t].first != *last) { R->run_off.push_back(cstart[t] + out[t].first_end); R->run_name.push_back(*last); }
        for (const Change &c : out[t].changes) { R->run_off.push_back(cstart[t] + c.end_pos); R->run_name.push_back(c.prev); }
        last = &out[t].last;
    }
    if (ctx.trailing) { R->run_off.push_back(total); R->run_name.push_back(*last); }
    tm.lap("runs");
    return PG_OK;
}

// a regular, uncompressed file?  (fd stays open on success)
int open_plain(const char *path, int &fd, size_t &size, bool &plain)
{
    plain = false;
    fd = open(path, O_RDONLY);
    if (fd < 0) return pg_fail(PG_EIO, "cannot open %s", path);
    struct stat st;
    if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode)) {
        size = (size_t)st.st_size;
        unsigned char magic[2] = {0, 0};
        plain = !(size >= 2 && pread(fd, magic, 2, 0) == 2 && magic[0] == 0x1f && magic[1] == 0x8b);
    }
    return PG_OK;
}

// how many bytes of inflated text one gzip input may park in memory (an anonymous memfd is shmem: it has no quota of its own for
// write() to fail against -- past the machine's or the cgroup's limit the kernel kills the process).  Half of what is
// available right now, shared by the files inflated side by side; PG_INFLATE_MAX_BYTES overrides (0 = never inflate to memory).
static uint64_t read_u64_file(const char *path)
{
    FILE *f = fopen(path, "r");
    if (!f) return UINT64_MAX;
    char buf[64] = {0};
    const bool ok = fgets(buf, sizeof buf, f) != nullptr;
    fclose(f);
    if (!ok || buf[0] < '0' || buf[0] > '9') return UINT64_MAX;          // ("max": no limit)
    return strtoull(buf, nullptr, 10);
}
uint64_t inflate_budget(int n_files)
{
    if (const char *e = getenv("PG_INFLATE_MAX_BYTES")) return strtoull(e, nullptr, 10);
    uint64_t avail = UINT64_MAX;
    if (FILE *f = fopen("/proc/meminfo", "r")) {
        char line[256];
        while (fgets(line, sizeof line, f)) {
            unsigned long long kb;
            if (sscanf(line, "MemAvailable: %llu kB", &kb) == 1) { avail = (uint64_t)kb << 10; break; }
        }
        fclose(f);
    }
    // the cgroup's headroom (v2, then v1), where there is a limit
    uint64_t lim = read_u64_file("/sys/fs/cgroup/memory.max"), cur = read_u64_file("/sys/fs/cgroup/memory.current");
    if (lim == UINT64_MAX) { lim = read_u64_file("/sys/fs/cgroup/memory/memory.limit_in_bytes"); cur = read_u64_file("/sys/fs/cgroup/memory/memory.usage_in_bytes"); }
    if (lim != UINT64_MAX && cur != UINT64_MAX && lim < ((uint64_t)1 << 60)) avail = std::min(avail, lim > cur ? lim - cur : 0);
    if (avail == UINT64_MAX) return (uint64_t)1 << 30;                    // nothing known: a cautious gigabyte
    return avail / 2 / (uint64_t)(n_files > 0 ? n_files : 1);
}

// a gzip file for the threaded readers: inflated (one stream = one thread, as pigz -dc in feature.py:76-91) into an anonymous
// in-memory file, which pread serves like any other -- the parse that follows is then threaded instead of serial.  fd < 0
// on return (with PG_OK) means "not worth it / not possible here": the caller takes the serial path, which streams the file
// through zlib as the reference's pigz pipe does and holds nothing but the packed stream.  `budget`: bytes of text this file
// may park in memory (inflate_budget); a file whose compressed size alone says it will not fit (FASTQ deflates to a quarter
// or less) is not started, one that grows past the budget is dropped.
int inflate_to_memfd(const char *path, int &fd, size_t &size, uint64_t budget)
{
    fd = -1; size = 0;
    struct stat st;
    if (stat(path, &st) == 0 && (uint64_t)st.st_size * 2 > budget) return PG_OK;
    gzFile f = gzopen(path, "rb");
    if (!f) return pg_fail(PG_EIO, "cannot open %s", path);
    gzbuffer(f, 1 << 22);
    int m = memfd_create("pg_inflate", MFD_CLOEXEC);
    if (m < 0) { gzclose(f); return PG_OK; }
    std::vector<char> buf((size_t)1 << 22);
    for (;;) {
        const int got = gzread(f, buf.data(), (unsigned)buf.size());
        if (got < 0) { gzclose(f); close(m); return pg_fail(PG_EIO, "read error in %s", path); }
        if (got == 0) break;
        if ((uint64_t)size + (uint64_t)got > budget) { gzclose(f); close(m); size = 0; return PG_OK; }    // too big to park: serial path
        size_t done = 0;
        while (done < (size_t)got) {
            const ssize_t w = write(m, buf.data() + done, (size_t)got - done);
            if (w <= 0) { gzclose(f); close(m); size = 0; return PG_OK; }           // (no memory for it: serial path, which reports its own errors)
            done += (size_t)w;
        }
        size += (size_t)got;
    }
    gzclose(f);
    fd = m;
    return PG_OK;
}

// ------------------------------------------------------------------------------------ parallel paired ingest (-1 / -2)
//
// Same result as the serial paired loop of pg_ingest_fastq (count_tnf.cpp:174-231: lines of the two files in lockstep, a pair
// whose names or barcodes differ is skipped but its reads still feed the global table, jellyfish's --min-qual-char=? marks).
// Records are cut by R1 byte ranges: newline counts of both files give every thread the record index of its first R1 record,
// and the finer counts of R2 where the same record starts there.  Threads cover the records that are complete in BOTH files;
// whatever lies behind them (a record cut short, an R2 that is shorter or longer) is a tail the serial rules are applied to.
struct PairSink {                 // what a run of (R1 record, R2 record) pairs leaves behind
    LocalStream main, orph;       // kept pairs; reads of skipped pairs (they follow the last run)
    std::vector<Change> changes;
    bool any = false;
    std::string first, last;
    int64_t first_end = 0, pairs = 0, unpaired = 0;
    int64_t soff_main = 0, soff_orph = 0;     // (sink form) word offsets of the two local streams in the staging arrays
    PairSink() { main.with_q = true; orph.with_q = true; }
    void keep(const char *s1, size_t l1, const char *q1, size_t q1n, const char *s2, size_t l2, const char *q2, size_t q2n,
              const char *bc, size_t bcn)
    {
        main.put_line_q(s1, l1, q1, q1n);
        main.put_line_q(s2, l2, q2, q2n);
        ++pairs;
        if (!any) { any = true; first.assign(bc, bcn); first_end = main.n; last = first; }
        else if (bcn != last.size() || (bcn && memcmp(bc, last.data(), bcn) != 0)) {
            changes.push_back(Change{main.n, last});
            last.assign(bc, bcn);
        }
    }
    void skip(const char *s1, size_t l1, const char *q1, size_t q1n, const char *s2, size_t l2, const char *q2, size_t q2n)
    {
        orph.put_line_q(s1, l1, q1, q1n);
        orph.put_line_q(s2, l2, q2, q2n);
    }
};

struct PairLatch { uint64_t header; int mode; };     // first header (2 * record + file) that fixes the grammar (UINT64_MAX: none)
inline int mode_of(const PairLatch &L, uint64_t header) { return header < L.header ? (int)MODE_UNSET : L.mode; }

inline bool same_span(const char *a, const Span &x, const char *b, const Span &y)
{
    return x.n == y.n && (x.n == 0 || memcmp(a + x.b, b + y.b, x.n) == 0);
}

// lines of a file = newlines + an unterminated last line
int count_lines_blocks(int fd, size_t size, int n_blocks, int T, std::vector<uint64_t> &nl_before, uint64_t &n_lines, const char *path)
{
    std::vector<uint64_t> nl(n_blocks, 0);
    std::vector<char> bad(T, 0);
    const size_t block = reader_block((size_t)1 << 20);
    run_threads(T, [&](int t) {
        for (int j = t; j < n_blocks; j += T) {
            const size_t a = (size_t)((unsigned __int128)size * (unsigned)j / (unsigned)n_blocks);
            const size_t b = (size_t)((unsigned __int128)size * (unsigned)(j + 1) / (unsigned)n_blocks);
            UnitReader rd(fd, a, b, block);
            nl[j] = rd.count_newlines();
            if (rd.io_error()) bad[t] = 1;
        }
    });
    for (char x : bad) if (x) return pg_fail(PG_EIO, "read error in %s", path);
    nl_before.assign(n_blocks + 1, 0);
    for (int j = 0; j < n_blocks; ++j) nl_before[j + 1] = nl_before[j] + nl[j];
    char c = '\n';
    if (size > 0 && pread(fd, &c, 1, (off_t)(size - 1)) != 1) return pg_fail(PG_EIO, "read error in %s", path);
    n_lines = nl_before[n_blocks] + (size > 0 && c != '\n' ? 1 : 0);
    return PG_OK;
}

// byte offset of line `line` (0-based) of a file whose newlines before every block start are known; size if there is no such line
int offset_of_line(int fd, size_t size, int n_blocks, const std::vector<uint64_t> &nl_before, uint64_t line, size_t &off, const char *path)
{
    if (line == 0) { off = 0; return PG_OK; }
    if (line > nl_before[n_blocks]) { off = size; return PG_OK; }
    int j = 0;
    while (nl_before[j + 1] < line) ++j;                    // the block that holds the line-th newline
    const size_t a = (size_t)((unsigned __int128)size * (unsigned)j / (unsigned)n_blocks);
    UnitReader rd(fd, a, size, reader_block((size_t)1 << 16));
    for (uint64_t k = nl_before[j]; k < line; ++k)
        if (!rd.skip_line()) { if (rd.io_error()) return pg_fail(PG_EIO, "read error in %s", path); break; }
    off = rd.offset();
    return PG_OK;
}

// the serial rules on what the threads leave: R1 from a1, R2 from a2 (both at the start of record `rec0`)
int paired_tail(int fd1, size_t a1, size_t n1, int fd2, size_t a2, size_t n2, uint64_t rec0, const PairLatch &L, int &mode, PairSink &o,
                const char *r1)
{
    std::vector<char> t1(n1 - a1), t2(n2 - a2);
    if ((!t1.empty() && pread(fd1, t1.data(), t1.size(), (off_t)a1) != (ssize_t)t1.size()) ||
        (!t2.empty() && pread(fd2, t2.data(), t2.size(), (off_t)a2) != (ssize_t)t2.size()))
        return pg_fail(PG_EIO, "read error in %s", r1);
    Lines L1(t1.data(), t1.size()), L2(t2.data(), t2.size());
    if (L.header != UINT64_MAX && 2 * rec0 >= L.header) mode = L.mode;     // (else the tail latches by itself, as the serial loop does)
    uint64_t line_no = 0;
    const char *b, *c; size_t len, clen;
    Span n1s, b1s, n2s, b2s;
    const char *h1 = "", *h2 = "";
    const char *s1 = "", *s2 = ""; size_t l1 = 0, l2 = 0;
    bool have_pair = false, keep_pair = false;
    auto flush_pair = [&](const char *q1, size_t q1n, const char *q2, size_t q2n) {
        if (!have_pair) return;
        have_pair = false;
        if (keep_pair) o.keep(s1, l1, q1, q1n, s2, l2, q2, q2n, h1 + b1s.b, b1s.n);
        else o.skip(s1, l1, q1, q1n, s2, l2, q2, q2n);
    };
    while (L1.next(b, len)) {
        if (!L2.next(c, clen)) { c = ""; clen = 0; }
        switch (++line_no % 4) {
        case 1:
            if (!header_fields(b, len, mode, n1s, b1s) || !header_fields(c, clen, mode, n2s, b2s))
                return pg_fail(PG_EFORMAT, "%s line %llu: header ends inside its BX:Z tag (the reference aborts here)", r1,
                               (unsigned long long)(4 * rec0 + line_no));
            h1 = b; h2 = c;
            break;
        case 2:
            flush_pair("", 0, "", 0);
            keep_pair = same_span(h1, n1s, h2, n2s) && same_span(h1, b1s, h2, b2s);
            if (!keep_pair) o.unpaired++;
            s1 = b; l1 = len; s2 = c; l2 = clen;
            have_pair = true;
            break;
        case 0:
            flush_pair(b, len, c, clen);
            break;
        default: break;
        }
    }
    flush_pair("", 0, "", 0);
    uint64_t ln2 = line_no;
    const char *os = nullptr; size_t on = 0;
    while (L2.next(c, clen)) {                                  // R2 records beyond the end of R1: input of the global counter only
        ++ln2;
        if (ln2 % 4 == 2) { if (os) o.orph.put_line_q(os, on, "", 0); os = c; on = clen; }
        else if (ln2 % 4 == 0 && os) { o.orph.put_line_q(os, on, c, clen); os = nullptr; }
    }
    if (os) o.orph.put_line_q(os, on, "", 0);
    return PG_OK;
}

// With a sink (pg_ingest_fastq_pair_device) R1 is cut into pieces of PG_INGEST_PIECE bytes that the threads take from a queue; the
// local streams of a finished piece go to the sink (a copy to the GPU) while the others parse on, and the placement is left to
// the device (pg_ingest_place_pair); without one there is a piece per thread, placed in host arrays.
int ingest_paired_threaded(int fd1, size_t n1, int fd2, size_t n2, const char *r1, const char *r2, pg_reads *R, int T, const pg_piece_sink *sink = nullptr)
{
    int P = T;                                                  // pieces of R1
    if (sink) {
        size_t piece = (size_t)16 << 20;
        if (const char *e = getenv("PG_INGEST_PIECE")) { long v = atol(e); if (v >= 64) piece = (size_t)v; }
        P = (int)std::max<size_t>(1, std::min<size_t>((n1 + piece - 1) / piece, (size_t)1 << 20));
    }
    PhaseTimer tm;
    const size_t block = reader_block((size_t)1 << 20);
    // ---- A. lines of both files; R1 in P ranges (the pieces), R2 in finer blocks (to find a record's start)
    const int B2 = 16 * std::max(T, std::min(P, 4096));
    std::vector<uint64_t> nl1, nl2;
    uint64_t lines1 = 0, lines2 = 0;
    int rc;
    if ((rc = count_lines_blocks(fd1, n1, P, T, nl1, lines1, r1))) return rc;
    if ((rc = count_lines_blocks(fd2, n2, B2, T, nl2, lines2, r2))) return rc;
    const uint64_t full = std::min(lines1 / 4, lines2 / 4);     // records that are complete in both files
    tm.lap("lines");
    // ---- the grammar latch: first header, R1 before R2 record by record, that carries a tag
    PairLatch L{UINT64_MAX, MODE_UNSET};
    {
        UnitReader a(fd1, 0, n1, reader_block((size_t)1 << 16)), b(fd2, 0, n2, reader_block((size_t)1 << 16));
        UnitLines u, v;
        for (uint64_t r = 0; r < full && a.next(u, 4) && b.next(v, 4); ++r) {
            if (find_bxz(u.p[0], u.n[0]) != NPOS) { L = PairLatch{2 * r, MODE_10X}; break; }
            if (find_chr(u.p[0], u.n[0], '#', 0) != NPOS) { L = PairLatch{2 * r, MODE_STLFR}; break; }
            if (find_bxz(v.p[0], v.n[0]) != NPOS) { L = PairLatch{2 * r + 1, MODE_10X}; break; }
            if (find_chr(v.p[0], v.n[0], '#', 0) != NPOS) { L = PairLatch{2 * r + 1, MODE_STLFR}; break; }
        }
        if (a.io_error() || b.io_error()) return pg_fail(PG_EIO, "read error in %s", r1);
    }
    tm.lap("latch");
    // ---- B. every thread: first record of its R1 range, the same record in R2, then pairs up to the next thread's first record
    std::vector<uint64_t> rec0(P + 1, full);
    std::vector<char> at_line_start(P, 1), bad(P, 0);
    for (int t = 0; t < P; ++t) {
        const size_t a = (size_t)((unsigned __int128)n1 * (unsigned)t / (unsigned)P);
        if (t > 0 && a > 0) { char c = 0; if (pread(fd1, &c, 1, (off_t)(a - 1)) != 1) return pg_fail(PG_EIO, "read error in %s", r1); at_line_start[t] = c == '\n'; }
        const uint64_t line = nl1[t] + (at_line_start[t] ? 0 : 1);       // first line that starts in the range
        rec0[t] = std::min<uint64_t>((line + 3) / 4, full);
    }
    std::vector<PairSink> out(P + 1);                           // [P] = the tail
    std::vector<uint64_t> bad_rec(P, UINT64_MAX);
    std::atomic<int> next_piece{0};
    std::atomic<int64_t> staged{0};
    std::atomic<int> sink_rc{0};
    // the two local streams of a finished piece -> the sink; their vectors are given back (n, any_q and the sparse lists stay)
    auto to_sink = [&](PairSink &o, int worker) {
        int64_t *soff[2] = {&o.soff_main, &o.soff_orph};
        LocalStream *ls[2] = {&o.main, &o.orph};
        for (int i = 0; i < 2; ++i) {
            ls[i]->finish();
            const int64_t nw = (int64_t)ls[i]->codes.size();
            *soff[i] = staged.fetch_add(nw, std::memory_order_relaxed);
            if (nw) {
                const int rc_ = *soff[i] + nw <= sink->capacity_words
                              ? sink->copy(sink->ctx, worker, *soff[i], ls[i]->codes.data(), ls[i]->valid.data(), ls[i]->lowq.data(), nw) : PG_EINVAL;
                if (rc_) { int zero = 0; sink_rc.compare_exchange_strong(zero, rc_); }
            }
            std::vector<uint64_t>().swap(ls[i]->codes); std::vector<uint32_t>().swap(ls[i]->valid); std::vector<uint32_t>().swap(ls[i]->lowq);
        }
    };
    auto one_piece = [&](int t, int worker) {
        PairSink &o = out[t];
        const uint64_t ra = rec0[t], rb = rec0[t + 1];
        if (ra >= rb) return;
        const size_t a = (size_t)((unsigned __int128)n1 * (unsigned)t / (unsigned)P);
        UnitReader rd1(fd1, a, n1, block);
        uint64_t line = nl1[t];
        if (!at_line_start[t]) { if (!rd1.skip_line()) { bad[t] = 2; return; } ++line; }
        for (; line < 4 * ra; ++line) if (!rd1.skip_line()) { bad[t] = 3; return; }
        size_t off2 = 0;
        if (offset_of_line(fd2, n2, B2, nl2, 4 * ra, off2, r2)) { bad[t] = 4; return; }
        UnitReader rd2(fd2, off2, n2, block);
        o.main.codes.reserve((size_t)(rb - ra) * 10 + 64);
        o.main.valid.reserve((size_t)(rb - ra) * 10 + 64);
        o.main.lowq.reserve((size_t)(rb - ra) * 10 + 64);
        UnitLines u, v;
        for (uint64_t r = ra; r < rb; ++r) {
            if (!rd1.next(u, 4) || !rd2.next(v, 4) || u.count < 4 || v.count < 4) { bad[t] = 5; return; }
            int m1 = mode_of(L, 2 * r), m2 = mode_of(L, 2 * r + 1);
            Span a1, c1, a2, c2;
            if (!header_fields(u.p[0], u.n[0], m1, a1, c1) || !header_fields(v.p[0], v.n[0], m2, a2, c2)) { bad_rec[t] = r; return; }
            if (same_span(u.p[0], a1, v.p[0], a2) && same_span(u.p[0], c1, v.p[0], c2))
                o.keep(u.p[1], u.n[1], u.p[3], u.n[3], v.p[1], v.n[1], v.p[3], v.n[3], u.p[0] + c1.b, c1.n);
            else {
                o.unpaired++;
                o.skip(u.p[1], u.n[1], u.p[3], u.n[3], v.p[1], v.n[1], v.p[3], v.n[3]);
            }
        }
        if (rd1.io_error() || rd2.io_error()) { bad[t] = 6; return; }
        if (sink) to_sink(o, worker);
    };
    run_threads(T, [&](int worker) {
        if (!sink) { one_piece(worker, worker); return; }
        for (int p; (p = next_piece.fetch_add(1, std::memory_order_relaxed)) < P;) one_piece(p, worker);
    });
    uint64_t first_bad = UINT64_MAX;
    for (int t = 0; t < P; ++t) {
        if (bad[t]) return pg_fail(PG_EIO, "read error in %s / %s (stage %d, thread %d)", r1, r2, (int)bad[t], t);
        first_bad = std::min(first_bad, bad_rec[t]);
    }
    if (first_bad != UINT64_MAX)
        return pg_fail(PG_EFORMAT, "%s line %llu: header ends inside its BX:Z tag (the reference aborts here)", r1, (unsigned long long)(first_bad * 4 + 1));
    tm.lap("parse+pack");
    // ---- the tail, by the serial rules
    int mode = MODE_UNSET;
    {
        size_t a1 = 0, a2 = 0;
        if ((rc = offset_of_line(fd1, n1, P, nl1, 4 * full, a1, r1)) || (rc = offset_of_line(fd2, n2, B2, nl2, 4 * full, a2, r2))) return rc;
        if ((rc = paired_tail(fd1, a1, n1, fd2, a2, n2, full, L, mode, out[P], r1))) return rc;
        if (L.header != UINT64_MAX) mode = L.mode;
        if (sink) to_sink(out[P], 0);
    }
    if (sink) {
        if (sink_rc.load() == PG_EINVAL && staged.load() > sink->capacity_words)
            return pg_fail(PG_EINVAL, "staging arrays of %lld words are too small for %s / %s (pg_ingest_pair_staging_words)", (long long)sink->capacity_words, r1, r2);
        if (sink_rc.load()) return sink_rc.load();               // (the sink recorded its message)
    }
    // ---- C. place: kept pairs of the threads and of the tail, then the reads of the skipped pairs in the same order
    std::vector<LocalStream *> piece;
    for (int t = 0; t <= P; ++t) { if (!sink) out[t].main.finish(); piece.push_back(&out[t].main); }
    for (int t = 0; t <= P; ++t) { if (!sink) out[t].orph.finish(); piece.push_back(&out[t].orph); }
    const int NP = (int)piece.size();
    std::vector<int64_t> cstart(NP + 1, 0);
    for (int i = 0; i < NP; ++i) cstart[i + 1] = cstart[i] + piece[i]->n;
    const int64_t total = cstart[NP];
    bool any_q = false, any_lower = false;
    for (int i = 0; i < NP; ++i) { any_q |= piece[i]->any_q; any_lower |= !piece[i]->lower.empty(); }
    if (sink) {                                       // the device does it: pg_ingest_place_pair
        R->set_stream_size(total);
        R->piece_soff.resize(NP);
        for (int t = 0; t <= P; ++t) { R->piece_soff[t] = out[t].soff_main; R->piece_soff[P + 1 + t] = out[t].soff_orph; }
        R->piece_cstart = cstart;
        R->staged_lowq = any_q;
    } else {
    if (!R->alloc_stream(total)) return pg_fail(PG_ENOMEM, "out of memory while ingesting %s", r1);
    uint64_t *gc = R->codes_w; uint32_t *gv = R->valid_w;
    if (any_q) R->lowq_plane.assign((size_t)R->n_words, 0u);
    uint32_t *gq = any_q ? R->lowq_plane.data() : nullptr;
    for (int64_t w = (total + 31) >> 5; w < R->n_words; ++w) { gc[w] = 0; gv[w] = 0; }
    for (int i = 0; i < NP; ++i)
        if (piece[i]->n) {
            const int64_t a = cstart[i] >> 5, b = (cstart[i + 1] - 1) >> 5;
            gc[a] = 0; gv[a] = 0; gc[b] = 0; gv[b] = 0;
        }
    run_threads(T, [&](int t) {
        for (int i = t; i < NP; i += T) {
            const LocalStream &ls = *piece[i];
            if (ls.n == 0) continue;
            const int64_t w0 = cstart[i] >> 5, w1 = (cstart[i + 1] - 1) >> 5;
            const int sh = (int)(cstart[i] & 31);
            const int64_t nw = (int64_t)ls.codes.size();
            for (int64_t g = w0; g <= w1; ++g) {
                const int64_t k = g - w0;
                uint64_t c = k < nw ? ls.codes[k] << (2 * sh) : 0;
                uint32_t v = k < nw ? ls.valid[k] << sh : 0;
                uint32_t q = gq && k < nw ? ls.lowq[k] << sh : 0;
                if (sh && k > 0) { c |= ls.codes[k - 1] >> (64 - 2 * sh); v |= ls.valid[k - 1] >> (32 - sh); if (gq) q |= ls.lowq[k - 1] >> (32 - sh); }
                if (g == w0 || g == w1) {
                    __atomic_fetch_or(&gc[g], c, __ATOMIC_RELAXED); __atomic_fetch_or(&gv[g], v, __ATOMIC_RELAXED);
                    if (gq) __atomic_fetch_or(&gq[g], q, __ATOMIC_RELAXED);
                } else { gc[g] = c; gv[g] = v; if (gq) gq[g] = q; }
            }
        }
    });
    if (any_q) R->lowq_w = R->lowq_plane.data();
    }
    if (any_lower) {
        R->lower_plane.assign((size_t)R->n_words, 0u);
        for (int i = 0; i < NP; ++i)
            for (const LowerMask &m : piece[i]->lower) {
                const int64_t pos = cstart[i] + m.pos;
                const int sh = (int)(pos & 31);
                R->lower_plane[(size_t)(pos >> 5)] |= m.mask << sh;
                if (sh && (m.mask >> (32 - sh))) R->lower_plane[(size_t)(pos >> 5) + 1] |= m.mask >> (32 - sh);
            }
        R->lower_w = R->lower_plane.data();
    }
    tm.lap("place");
    // ---- D. runs (as in the interleaved form); the skipped pairs' reads lie behind the last run
    R->mode = mode;
    R->run_off.push_back(0);
    std::string none;
    const std::string *last = &none;
    for (int t = 0; t <= P; ++t) {
        R->n_pairs += out[t].pairs;
        R->n_unpaired += out[t].unpaired;
        if (!out[t].any) continue;
        if (out[t].first != *last) { R->run_off.push_back(cstart[t] + out[t].first_end); R->run_name.push_back(*last); }
        for (const Change &c : out[t].changes) { R->run_off.push_back(cstart[t] + c.end_pos); R->run_name.push_back(c.prev); }
        last = &out[t].last;
    }
    R->run_off.push_back(cstart[P + 1]);
    R->run_name.push_back(*last);
    tm.lap("runs");
    return PG_OK;
}

}  // namespace

extern "C" void pg_set_ingest_threads(int n) { g_ingest_threads = n > 0 ? n : 0; }

extern "C" int pg_ingest_fastq(const char *r1, const char *r2, pg_reads **out)
{
    if (!r1 || !out) return pg_fail(PG_EINVAL, "pg_ingest_fastq: null argument");
    *out = nullptr;
    int rc;
    const int T = ingest_threads();
    if (!r2 && T > 1) {           // an uncompressed interleaved file of some size: threaded, streamed from the file
        int fd; size_t size = 0; bool plain;
        if ((rc = open_plain(r1, fd, size, plain))) return rc;
        if (!plain && size >= ((size_t)1 << 14) * (size_t)T) {        // gzip of some size: inflate once, then parse by threads
            close(fd);
            if ((rc = inflate_to_memfd(r1, fd, size, inflate_budget(1)))) return rc;
            if (fd < 0) { if ((rc = open_plain(r1, fd, size, plain))) return rc; plain = false; }
            else plain = true;
        }
        if (plain && size >= ((size_t)1 << 16) * (size_t)T) {
            pg_reads *R = new (std::nothrow) pg_reads();
            if (!R) { close(fd); return pg_fail(PG_ENOMEM, "out of memory"); }
            try {
                PhaseTimer tm;
                const Latch L = find_latch(fd, size);
                tm.lap("latch");
                rc = ingest_interleaved_range(fd, 0, size, r1, R, T, L, ShardCtx());
            } catch (const std::bad_alloc &) {
                rc = pg_fail(PG_ENOMEM, "out of memory while ingesting %s", r1);
            }
            close(fd);
            if (rc) { delete R; return rc; }
            *out = R;
            return PG_OK;
        }
        close(fd);
    }
    if (r2 && T > 1) {            // two uncompressed files of some size: threaded as well
        int fd1, fd2; size_t n1 = 0, n2 = 0; bool p1, p2;
        if ((rc = open_plain(r1, fd1, n1, p1))) return rc;
        if ((rc = open_plain(r2, fd2, n2, p2))) { close(fd1); return rc; }
        if ((!p1 || !p2) && n1 >= ((size_t)1 << 14) * (size_t)T) {     // gzip: the two streams are inflated side by side
            int g1 = -1, g2 = -1; size_t m1 = 0, m2 = 0; int rc1 = PG_OK, rc2 = PG_OK;
            const uint64_t budget = inflate_budget((p1 ? 0 : 1) + (p2 ? 0 : 1));
            std::thread other([&] { if (!p2) rc2 = inflate_to_memfd(r2, g2, m2, budget); });
            if (!p1) rc1 = inflate_to_memfd(r1, g1, m1, budget);
            other.join();
            if (rc1 || rc2) { close(fd1); close(fd2); if (g1 >= 0) close(g1); if (g2 >= 0) close(g2); return rc1 ? rc1 : rc2; }
            if ((p1 || g1 >= 0) && (p2 || g2 >= 0)) {
                if (!p1) { close(fd1); fd1 = g1; n1 = m1; p1 = true; }
                if (!p2) { close(fd2); fd2 = g2; n2 = m2; p2 = true; }
            } else { if (g1 >= 0) close(g1); if (g2 >= 0) close(g2); }
        }
        if (p1 && p2 && n1 >= ((size_t)1 << 16) * (size_t)T) {
            pg_reads *R = new (std::nothrow) pg_reads();
            if (!R) { close(fd1); close(fd2); return pg_fail(PG_ENOMEM, "out of memory"); }
            try {
                rc = ingest_paired_threaded(fd1, n1, fd2, n2, r1, r2, R, T);
            } catch (const std::bad_alloc &) {
                rc = pg_fail(PG_ENOMEM, "out of memory while ingesting %s", r1);
            }
            close(fd1); close(fd2);
            if (rc) { delete R; return rc; }
            *out = R;
            return PG_OK;
        }
        close(fd1); close(fd2);
    }
    FileBuf f1, f2;
    rc = slurp(r1, f1);
    if (rc) return rc;
    if (r2 && (rc = slurp(r2, f2))) return rc;

    pg_reads *R = new (std::nothrow) pg_reads();
    if (!R) return pg_fail(PG_ENOMEM, "out of memory");
    try {
        R->st.codes.reserve(f1.size() / 64 + f2.size() / 64 + PG_WORD_ALIGN);
        R->st.valid.reserve(f1.size() / 64 + f2.size() / 64 + PG_WORD_ALIGN);
        R->run_off.push_back(0);
        std::string last, cur_bc;
        const char *b; size_t len;
        Span nm, bc;

        if (!r2) {
            Lines L(f1);
            uint64_t line_no = 0;
            while (L.next(b, len)) {
                switch (++line_no % 8) {
                case 1:
                    if (!header_fields(b, len, R->mode, nm, bc)) {
                        delete R;
                        return pg_fail(PG_EFORMAT, "%s line %llu: header ends inside its BX:Z tag (the reference aborts here)", r1, (unsigned long long)line_no);
                    }
                    cur_bc.assign(b + bc.b, bc.n);
                    break;
                case 2:
                    R->st.put_span(b, len); R->st.put('N');
                    break;
                case 6:
                    R->st.put_span(b, len); R->st.put('N');
                    R->n_pairs++;
                    if (cur_bc != last) {        // the pair just appended closes the run of the PREVIOUS barcode
                        R->run_off.push_back(R->st.n);
                        R->run_name.push_back(last);
                        last = cur_bc;
                    }
                    break;
                default: break;
                }
            }
            R->run_off.push_back(R->st.n);       // trailing accumulator
            R->run_name.push_back(last);
        } else {
            Lines L1(f1), L2(f2);
            uint64_t line_no = 0;
            std::string n1, n2, b1, b2, pair_bc;                    // pair_bc: the barcode of the pair that is staged (b1 moves on with the next header)
            struct ReadQ { const char *s; size_t n; const char *q; size_t qn; };
            std::vector<ReadQ> orphans;                             // reads of skipped pairs (still counted globally)
            const char *c; size_t clen;
            const char *s1 = "", *s2 = ""; size_t l1 = 0, l2 = 0;    // the pair's sequence lines, written when its quality lines are known
            bool have_pair = false, keep_pair = false;
            auto flush_pair = [&](const char *q1, size_t q1n, const char *q2, size_t q2n) {
                if (!have_pair) return;
                have_pair = false;
                if (!keep_pair) {
                    orphans.push_back(ReadQ{s1, l1, q1, q1n});
                    orphans.push_back(ReadQ{s2, l2, q2, q2n});
                    return;
                }
                R->st.put_span_q(s1, l1, q1, q1n); R->st.put('N');
                R->st.put_span_q(s2, l2, q2, q2n); R->st.put('N');
                R->n_pairs++;
                if (pair_bc != last) {
                    R->run_off.push_back(R->st.n);
                    R->run_name.push_back(last);
                    last = pair_bc;
                }
            };
            while (L1.next(b, len)) {
                if (!L2.next(c, clen)) { c = ""; clen = 0; }        // a short R2 reads as empty lines
                switch (++line_no % 4) {
                case 1: {
                    Span s1n, s1b, s2n, s2b;
                    if (!header_fields(b, len, R->mode, s1n, s1b) || !header_fields(c, clen, R->mode, s2n, s2b)) {
                        delete R;
                        return pg_fail(PG_EFORMAT, "%s line %llu: header ends inside its BX:Z tag (the reference aborts here)", r1, (unsigned long long)line_no);
                    }
                    n1.assign(b + s1n.b, s1n.n); b1.assign(b + s1b.b, s1b.n);
                    n2.assign(c + s2n.b, s2n.n); b2.assign(c + s2b.b, s2b.n);
                    break;
                }
                case 2:
                    flush_pair("", 0, "", 0);                       // (a record cut short of its quality line)
                    keep_pair = !(n1 != n2 || b1 != b2);
                    if (!keep_pair) R->n_unpaired++;
                    s1 = b; l1 = len; s2 = c; l2 = clen;
                    pair_bc = b1;
                    have_pair = true;
                    break;
                case 0:
                    flush_pair(b, len, c, clen);
                    break;
                default: break;
                }
            }
            flush_pair("", 0, "", 0);
            // R2 records beyond the end of R1 are still input of the global counter
            uint64_t ln2 = line_no;
            const char *os = nullptr; size_t on = 0;
            while (L2.next(c, clen)) {
                ++ln2;
                if (ln2 % 4 == 2) { if (os) orphans.push_back(ReadQ{os, on, "", 0}); os = c; on = clen; }
                else if (ln2 % 4 == 0 && os) { orphans.push_back(ReadQ{os, on, c, clen}); os = nullptr; }
            }
            if (os) orphans.push_back(ReadQ{os, on, "", 0});
            R->run_off.push_back(R->st.n);
            R->run_name.push_back(last);
            for (auto &o : orphans) { R->st.put_span_q(o.s, o.n, o.q, o.qn); R->st.put('N'); }
        }
        R->seal_serial();
    } catch (const std::bad_alloc &) {
        delete R;
        return pg_fail(PG_ENOMEM, "out of memory while ingesting %s", r1);
    }
    *out = R;
    return PG_OK;
}

// ------------------------------------------------------------------------------------ sharded ingest (one rank per GPU)
//
// Rank `part` of `n_parts` parses only its own byte range of an uncompressed interleaved file.  Byte boundary i is
// size * i / n_parts; the ranks count the newlines of their ranges (pg_fastq_count_newlines), exchange the counts, and
// every rank then knows the line number at both of its boundaries -- so record alignment (line number mod 8) is exact,
// not guessed from '@' characters.  A boundary is moved forward to the end of the run in progress: with c = the first
// pair that begins at or after the boundary byte, the cut falls behind the first later pair whose barcode differs from
// its predecessor's (that pair closes the predecessor's run, the append-then-compare rule), and the run that follows is
// named after it.  Both neighbours derive the same cut from the same bytes, so every pair belongs to exactly one shard
// and the shards' runs, concatenated in rank order, are the runs of the whole file.

namespace {

inline size_t shard_byte(size_t size, int i, int n_parts) { return (size_t)((unsigned __int128)size * (unsigned)i / (unsigned)n_parts); }

struct Cut { size_t pos = 0; bool found = false; std::string last; uint64_t unit = 0; };

// the cut that belongs to boundary byte b (0 < b < size); nl_before = newlines in [0, b)
int find_cut(int fd, size_t size, size_t b, uint64_t nl_before, const Latch &L, const char *path, Cut &out)
{
    out = Cut(); out.pos = size;
    UnitReader rd(fd, b, size, reader_block((size_t)1 << 16));
    uint64_t line = nl_before;
    char before = '\n';
    if (b > 0 && pread(fd, &before, 1, (off_t)(b - 1)) != 1) return pg_fail(PG_EIO, "read error in %s", path);
    if (before != '\n') { if (!rd.skip_line()) return rd.io_error() ? pg_fail(PG_EIO, "read error in %s", path) : (int)PG_OK; ++line; }
    for (uint64_t skip = (8 - line % 8) % 8; skip; --skip, ++line)
        if (!rd.skip_line()) return rd.io_error() ? pg_fail(PG_EIO, "read error in %s", path) : (int)PG_OK;
    uint64_t u = line / 8;                            // the reader stands at the start of unit u: the context pair
    UnitLines ul;
    std::string prev;
    bool have_prev = false;
    for (; rd.next(ul); ++u) {
        int mode = mode_of(L, u);
        Span nm, bc;
        if (!header_fields(ul.p[0], ul.n[0], mode, nm, bc))
            return pg_fail(PG_EFORMAT, "%s line %llu: header ends inside its BX:Z tag (the reference aborts here)", path, (unsigned long long)(u * 8 + 1));
        if (ul.count < 6) break;                      // no further complete pair: the run lasts to the end of the file
        const char *c = ul.p[0] + bc.b;
        if (have_prev && (bc.n != prev.size() || (bc.n && memcmp(c, prev.data(), bc.n) != 0))) {
            out.pos = rd.offset(); out.found = true; out.last.assign(c, bc.n); out.unit = u + 1;
            return PG_OK;
        }
        if (!have_prev) { prev.assign(c, bc.n); have_prev = true; }
    }
    if (rd.io_error()) return pg_fail(PG_EIO, "read error in %s", path);
    return PG_OK;
}

int open_for_shards(const char *path, int &fd, size_t &size)
{
    bool plain;
    int rc = open_plain(path, fd, size, plain);
    if (rc) return rc;
    if (!plain) { close(fd); fd = -1; return pg_fail(PG_EFORMAT, "%s: sharded ingest needs an uncompressed regular file", path); }
    return PG_OK;
}

}  // namespace

extern "C" int pg_fastq_count_newlines(const char *path, int part, int n_parts, int64_t *n_newlines)
{
    if (!path || !n_newlines || n_parts < 1 || part < 0 || part >= n_parts) return pg_fail(PG_EINVAL, "pg_fastq_count_newlines: bad argument");
    int fd; size_t size = 0;
    int rc = open_for_shards(path, fd, size);
    if (rc) return rc;
    PhaseTimer tm;
    const size_t a = shard_byte(size, part, n_parts), b = shard_byte(size, part + 1, n_parts);
    const int T = (b - a) >= ((size_t)8 << 20) ? ingest_threads() : 1;
    std::vector<uint64_t> c(T, 0);
    std::vector<char> bad(T, 0);
    run_threads(T, [&](int t) {
        UnitReader rd(fd, a + (b - a) * (size_t)t / T, a + (b - a) * (size_t)(t + 1) / T, reader_block((size_t)1 << 20));
        c[t] = rd.count_newlines();
        bad[t] = rd.io_error();
    });
    close(fd);
    for (char x : bad) if (x) return pg_fail(PG_EIO, "read error in %s", path);
    uint64_t total = 0;
    for (uint64_t v : c) total += v;
    *n_newlines = (int64_t)total;
    tm.lap("newlines");
    return PG_OK;
}

namespace {
int ingest_shard_impl(const char *path, int part, int n_parts, const int64_t *newlines_before, const pg_piece_sink *sink, pg_reads **out)
{
    *out = nullptr;
    int fd; size_t size = 0;
    int rc = open_for_shards(path, fd, size);
    if (rc) return rc;
    pg_reads *R = nullptr;
    try {
        PhaseTimer tm;
        const Latch L = find_latch(fd, size);
        Cut lo, hi;
        lo.pos = 0; lo.found = true;
        hi.pos = size; hi.found = false;
        if (part > 0) rc = find_cut(fd, size, shard_byte(size, part, n_parts), (uint64_t)newlines_before[part], L, path, lo);
        if (!rc && part + 1 < n_parts) rc = find_cut(fd, size, shard_byte(size, part + 1, n_parts), (uint64_t)newlines_before[part + 1], L, path, hi);
        tm.lap("cuts");
        if (!rc) R = new pg_reads();
        if (!rc && !lo.found) {                       // the run in progress at this boundary lasts to the end of the file
            R->mode = L.mode;
            R->run_off.push_back(0);
            R->seal_serial();
            if (sink) R->piece_cstart.assign(1, 0);
        } else if (!rc) {
            ShardCtx ctx;
            ctx.unit_base = lo.unit;
            ctx.last = lo.last;
            ctx.trailing = !hi.found;
            const size_t a = lo.pos, b = std::max(lo.pos, hi.pos);
            const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)ingest_threads(), (b - a) >> 16));
            rc = ingest_interleaved_range(fd, a, b, path, R, T, L, ctx, sink);
        }
    } catch (const std::bad_alloc &) {
        rc = pg_fail(PG_ENOMEM, "out of memory while ingesting %s", path);
    }
    close(fd);
    if (rc) { delete R; return rc; }
    *out = R;
    return PG_OK;
}
}  // namespace

extern "C" int pg_ingest_fastq_shard(const char *path, int part, int n_parts, const int64_t *newlines_before, pg_reads **out)
{
    if (!path || !out || !newlines_before || n_parts < 1 || part < 0 || part >= n_parts) return pg_fail(PG_EINVAL, "pg_ingest_fastq_shard: bad argument");
    return ingest_shard_impl(path, part, n_parts, newlines_before, nullptr, out);
}

// the same ingests with the pieces handed to a sink (ingest_dev.hip: copies to the GPU); *out stays NULL (and the status PG_OK)
// when the input is not an uncompressed interleaved file: the caller then takes pg_ingest_fastq and copies the arrays itself
int pg_internal_ingest_to_sink(const char *path, int part, int n_parts, const int64_t *newlines_before, const pg_piece_sink *sink, pg_reads **out)
{
    *out = nullptr;
    int fd; size_t size = 0; bool plain;
    int rc = open_plain(path, fd, size, plain);
    if (rc) return rc;
    if (!plain) { close(fd); return PG_OK; }
    if (n_parts > 1) { close(fd); return ingest_shard_impl(path, part, n_parts, newlines_before, sink, out); }
    pg_reads *R = new (std::nothrow) pg_reads();
    if (!R) { close(fd); return pg_fail(PG_ENOMEM, "out of memory"); }
    try {
        PhaseTimer tm;
        const Latch L = find_latch(fd, size);
        tm.lap("latch");
        const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)ingest_threads(), size >> 16));
        rc = ingest_interleaved_range(fd, 0, size, path, R, T, L, ShardCtx(), sink);
    } catch (const std::bad_alloc &) {
        rc = pg_fail(PG_ENOMEM, "out of memory while ingesting %s", path);
    }
    close(fd);
    if (rc) { delete R; return rc; }
    *out = R;
    return PG_OK;
}

extern "C" int pg_inflate_to_memfd(const char *path, int *fd_out, int64_t *bytes_out)
{
    if (!path || !fd_out || !bytes_out) return pg_fail(PG_EINVAL, "pg_inflate_to_memfd: null argument");
    *fd_out = -1; *bytes_out = 0;
    int fd; size_t size = 0; bool plain;
    int rc = open_plain(path, fd, size, plain);
    if (rc) return rc;
    close(fd);
    if (plain) return PG_OK;
    int m = -1; size_t text = 0;
    if ((rc = inflate_to_memfd(path, m, text, inflate_budget(1)))) return rc;
    if (m < 0) return PG_OK;
    *fd_out = m;
    *bytes_out = (int64_t)text;
    return PG_OK;
}

// -1 / -2 input with the pieces handed to a sink; *out stays NULL (status PG_OK) when the two are not uncompressed files of some size
int pg_internal_ingest_pair_to_sink(const char *r1, const char *r2, const pg_piece_sink *sink, pg_reads **out)
{
    *out = nullptr;
    int fd1, fd2; size_t n1 = 0, n2 = 0; bool p1, p2;
    int rc = open_plain(r1, fd1, n1, p1);
    if (rc) return rc;
    if ((rc = open_plain(r2, fd2, n2, p2))) { close(fd1); return rc; }
    if (!p1 || !p2 || n1 < 64) { close(fd1); close(fd2); return PG_OK; }
    pg_reads *R = new (std::nothrow) pg_reads();
    if (!R) { close(fd1); close(fd2); return pg_fail(PG_ENOMEM, "out of memory"); }
    try {
        const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)ingest_threads(), n1 >> 10));
        rc = ingest_paired_threaded(fd1, n1, fd2, n2, r1, r2, R, T, sink);
    } catch (const std::bad_alloc &) {
        rc = pg_fail(PG_ENOMEM, "out of memory while ingesting %s", r1);
    }
    close(fd1); close(fd2);
    if (rc) { delete R; return rc; }
    *out = R;
    return PG_OK;
}
bool pg_internal_reads_staged_lowq(const pg_reads *r) { return r && r->staged_lowq; }

int64_t pg_internal_reads_pieces(const pg_reads *r, const int64_t **soff, const int64_t **cstart)
{
    if (!r || r->piece_cstart.empty()) return -1;
    *soff = r->piece_soff.data();
    *cstart = r->piece_cstart.data();
    return (int64_t)r->piece_cstart.size() - 1;
}

extern "C" void pg_reads_free(pg_reads *r) { delete r; }
extern "C" int64_t pg_reads_n_chars(const pg_reads *r) { return r->n_chars; }
extern "C" int64_t pg_reads_n_words(const pg_reads *r) { return r->n_words; }
extern "C" int64_t pg_reads_n_pairs(const pg_reads *r) { return r->n_pairs; }
extern "C" int64_t pg_reads_n_unpaired(const pg_reads *r) { return r->n_unpaired; }
extern "C" int64_t pg_reads_n_runs(const pg_reads *r) { return (int64_t)r->run_name.size(); }
extern "C" const uint64_t *pg_reads_codes(const pg_reads *r) { return r->codes_w; }
extern "C" const uint32_t *pg_reads_valid(const pg_reads *r) { return r->valid_w; }
extern "C" const uint32_t *pg_reads_lower(const pg_reads *r) { return r->lower_w; }
extern "C" const uint32_t *pg_reads_lowq(const pg_reads *r) { return r->lowq_w; }
extern "C" const int64_t *pg_reads_run_off(const pg_reads *r) { return r->run_off.data(); }
extern "C" const char *pg_reads_run_name(const pg_reads *r, int64_t i)
{
    if (i < 0 || i >= (int64_t)r->run_name.size()) return "";
    return r->run_name[(size_t)i].c_str();
}
// all run names in one call (51 k ctypes calls for the names of a 10 M-pair file took 25 ms): the names, each followed by a
// NUL, into out[0 .. cap); returns the bytes that takes (call with cap = 0 to size the buffer)
extern "C" int64_t pg_reads_run_names(const pg_reads *r, char *out, int64_t cap)
{
    int64_t need = 0;
    for (const std::string &n : r->run_name) need += (int64_t)n.size() + 1;
    if (out && cap >= need) {
        char *q = out;
        for (const std::string &n : r->run_name) { memcpy(q, n.data(), n.size()); q += n.size(); *q++ = 0; }
    }
    return need;
}
extern "C" const char *pg_reads_mode(const pg_reads *r)
{
    return r->mode == MODE_10X ? "10x" : r->mode == MODE_STLFR ? "stLFR" : "";
}

extern "C" int64_t pg_reads_rows(const pg_reads *r, int min_len, int64_t *row_run)
{
    int64_t n = 0;
    for (size_t i = 0; i < r->run_name.size(); ++i) {
        const int64_t len = r->run_off[i + 1] - r->run_off[i];
        if (r->run_name[i].empty() || len <= (int64_t)min_len) continue;
        if (row_run) row_run[n] = (int64_t)i;
        ++n;
    }
    return n;
}

extern "C" int64_t pg_words_for(int64_t n_chars)
{
    if (n_chars < 0) return pg_fail(PG_EINVAL, "negative length");
    int64_t words = (n_chars + 31) / 32;
    int64_t padded = (words + PG_WORD_ALIGN - 1) / PG_WORD_ALIGN * PG_WORD_ALIGN;
    return padded ? padded : PG_WORD_ALIGN;
}

extern "C" int pg_pack_ascii_lower(const char *text, int64_t n_chars, uint64_t *codes, uint32_t *valid, uint32_t *lower)
{
    if (n_chars < 0 || (n_chars > 0 && !text) || !codes || !valid) return pg_fail(PG_EINVAL, "pg_pack_ascii: bad arguments");
    const int64_t words = pg_words_for(n_chars);
    memset(codes, 0, (size_t)words * sizeof(uint64_t));
    memset(valid, 0, (size_t)words * sizeof(uint32_t));
    if (lower) memset(lower, 0, (size_t)words * sizeof(uint32_t));
    int any = 0;
    for (int64_t i = 0; i < n_chars; ++i) {
        const unsigned char c = (unsigned char)text[i];
        const bool up = c == 'A' || c == 'C' || c == 'G' || c == 'T';
        const bool lo = c == 'a' || c == 'c' || c == 'g' || c == 't';
        if (up || lo) codes[i >> 5] |= (uint64_t)((c >> 1) & 3) << (2 * (i & 31));
        if (up) valid[i >> 5] |= 1u << (i & 31);
        if (lo && lower) { lower[i >> 5] |= 1u << (i & 31); any = 1; }
    }
    return any;             // 1: the text has lower-case bases (only reported when `lower` is given)
}

extern "C" int pg_pack_ascii(const char *text, int64_t n_chars, uint64_t *codes, uint32_t *valid)
{
    const int rc = pg_pack_ascii_lower(text, n_chars, codes, valid, nullptr);
    return rc < 0 ? rc : PG_OK;
}

extern "C" int64_t pg_plan_segments(const int64_t *row_start, const int64_t *row_end, int64_t n_rows, int64_t seg_chars,
                                    int32_t *seg_row, int64_t *seg_start, int64_t *seg_end)
{
    if (n_rows < 0 || (n_rows > 0 && (!row_start || !row_end))) return pg_fail(PG_EINVAL, "pg_plan_segments: bad rows");
    if (seg_chars < 32 || seg_chars % 32) return pg_fail(PG_EINVAL, "pg_plan_segments: seg_chars must be a positive multiple of 32");
    if (n_rows > 0x7fffffffLL) return pg_fail(PG_EINVAL, "pg_plan_segments: too many rows");
    int64_t n = 0;
    for (int64_t r = 0; r < n_rows; ++r) {
        if (row_end[r] < row_start[r]) return pg_fail(PG_EINVAL, "pg_plan_segments: row %lld has end < start", (long long)r);
        for (int64_t s = row_start[r]; s < row_end[r]; s += seg_chars) {
            if (seg_row) {
                seg_row[n] = (int32_t)r;
                seg_start[n] = s;
                seg_end[n] = s + seg_chars < row_end[r] ? s + seg_chars : row_end[r];
            }
            ++n;
        }
    }
    return n;
}

// ------------------------------------------------------------------------------------ TNF columns

namespace {
uint32_t revcomp_code(uint32_t x, int k)
{
    uint32_t r = 0;
    for (int i = 0; i < k; ++i) { r = (r << 2) | ((x & 3) ^ 2); x >>= 2; }
    return r;
}
}  // namespace

extern "C" int pg_tnf_ncols(int k)
{
    if (k < 1 || k > PG_TNF_MAX_K) return pg_fail(PG_EINVAL, "tnf k must be in [1,%d] (got %d)", PG_TNF_MAX_K, k);
    int n = 0;
    for (uint32_t c = 0; c < (1u << (2 * k)); ++c) n += c <= revcomp_code(c, k);
    return n;
}

extern "C" int pg_tnf_colmap(int k, uint16_t *colmap, uint32_t *col_code)
{
    const int ncols = pg_tnf_ncols(k);
    if (ncols < 0) return ncols;
    if (!colmap) return pg_fail(PG_EINVAL, "pg_tnf_colmap: colmap is null");
    const uint32_t n = 1u << (2 * k);
    std::vector<int> col_of(n, -1);
    int col = 0;
    for (uint32_t c = 0; c < n; ++c)
        if (c <= revcomp_code(c, k)) {          // ascending canonical code == std::map iteration order
            if (col_code) col_code[col] = c;
            col_of[c] = col++;
        }
    for (uint32_t c = 0; c < n; ++c) {
        const uint32_t r = revcomp_code(c, k);
        colmap[c] = (uint16_t)col_of[c < r ? c : r];
    }
    return ncols;
}

// ------------------------------------------------------------------------------------ CSV cache

namespace {

// ",<value>" as `ostream << double` prints it: %g with 6 significant digits.  Non-negative integers below 10^6 print as
// plain decimals under %g, which is the whole matrix in practice -- only larger counts take the snprintf path.
inline void append_number(std::string &out, int32_t v)
{
    out.push_back(',');
    if (v >= 0 && v < 1000000) {
        char tmp[8];
        int n = 0;
        do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
        while (n) out.push_back(tmp[--n]);
    } else {
        char num[48];
        const int m = snprintf(num, sizeof num, "%g", (double)v);
        out.append(num, (size_t)m);
    }
}

// one complete gzip member holding `text`
bool gzip_member(const std::string &text, std::string &out)
{
    z_stream z;
    memset(&z, 0, sizeof z);
    if (deflateInit2(&z, 1, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
    out.resize(deflateBound(&z, (uLong)text.size()) + 64);
    z.next_in = (Bytef *)text.data();
    z.avail_in = (uInt)text.size();
    z.next_out = (Bytef *)&out[0];
    z.avail_out = (uInt)out.size();
    const int rc = deflate(&z, Z_FINISH);
    const size_t produced = out.size() - z.avail_out;
    deflateEnd(&z);
    if (rc != Z_STREAM_END) return false;
    out.resize(produced);
    return true;
}

}  // namespace

extern "C" int pg_write_csv_gz(const char *path, const char *names, const int32_t *mat, int64_t n_rows, int64_t n_cols)
{
    if (!path || n_rows < 0 || n_cols < 0 || (n_rows > 0 && (!names || (n_cols > 0 && !mat))))
        return pg_fail(PG_EINVAL, "pg_write_csv_gz: bad arguments");
    // row chunks are formatted and compressed independently (each becomes one gzip member; a gzip file may hold any
    // number of members and reads back as their concatenation), then written in order
    std::vector<const char *> name_at((size_t)n_rows);
    {
        const char *nm = names;
        for (int64_t i = 0; i < n_rows; ++i) { name_at[(size_t)i] = nm; nm += strlen(nm) + 1; }
    }
    const int64_t rows_per_chunk = std::max<int64_t>(1, (int64_t)(((size_t)1 << 22) / (size_t)(8 * (n_cols + 4))));   // ~4 MB of text
    const int64_t n_chunks = n_rows ? (n_rows + rows_per_chunk - 1) / rows_per_chunk : 1;
    std::vector<std::string> packed((size_t)n_chunks);
    std::vector<char> bad((size_t)n_chunks, 0);
    const int T = (int)std::min<int64_t>(ingest_threads(), n_chunks);
    std::atomic<int64_t> next{0};
    run_threads(T, [&](int) {
        std::string text;
        for (;;) {
            const int64_t c = next.fetch_add(1);
            if (c >= n_chunks) break;
            text.clear();
            const int64_t r0 = c * rows_per_chunk, r1 = std::min(n_rows, r0 + rows_per_chunk);
            for (int64_t i = r0; i < r1; ++i) {
                text.append(name_at[(size_t)i]);
                for (int64_t j = 0; j < n_cols; ++j) append_number(text, mat[i * n_cols + j]);
                text.push_back('\n');
            }
            if (!gzip_member(text, packed[(size_t)c])) bad[(size_t)c] = 1;
        }
    });
    for (char x : bad) if (x) return pg_fail(PG_ENOMEM, "pg_write_csv_gz: compression failed");
    FILE *f = fopen(path, "wb");
    if (!f) return pg_fail(PG_EIO, "cannot create %s", path);
    for (const std::string &m : packed)
        if (fwrite(m.data(), 1, m.size(), f) != m.size()) { fclose(f); return pg_fail(PG_EIO, "write error on %s", path); }
    if (fclose(f) != 0) return pg_fail(PG_EIO, "close error on %s", path);
    return PG_OK;
}

// ------------------------------------------------------------------------------------ bin writer
//
// clusters.tsv -> <prefix>_bin<label>.fq / .barcode, as the reference's extract_reads.cpp:57-190 writes them:
//   * every tsv line "<label>\t<bc>,<bc>,..." opens both files of its label ("-1" lines are skipped entirely) and
//     maps its barcodes to it (a barcode listed twice belongs to the later line);
//   * interleaved input: a pair is kept when the barcode of mate 1's header is mapped; mate 1's header is rewritten to
//     "<name>\tBX:Z:<barcode>-1", the other seven lines are copied as they are (extract_reads.cpp:98-125);
//   * paired input: additionally both headers must agree in name and barcode, both are rewritten, and the pair goes to
//     the .fq as mate 1 record + mate 2 record (extract_reads.cpp:141-176);
//   * the .barcode file receives the barcode once per kept pair.

namespace {

struct BinFiles {
    FILE *fq = nullptr, *bc = nullptr;
    std::string fq_buf, bc_buf;
    void flush()
    {
        if (fq && !fq_buf.empty()) { fwrite(fq_buf.data(), 1, fq_buf.size(), fq); fq_buf.clear(); }
        if (bc && !bc_buf.empty()) { fwrite(bc_buf.data(), 1, bc_buf.size(), bc); bc_buf.clear(); }
    }
};

}  // namespace

namespace {

// Threaded form of the interleaved branch below, byte for byte the same files: T threads stream their byte ranges of the
// (uncompressed) input twice -- first to size every thread's share of every cluster file, then to write the records with
// pwrite at their exact offsets -- so every file keeps the input order although no thread ever waits for another.
int extract_interleaved_parallel(int fd, size_t size, const char *path, const std::unordered_map<std::string, uint32_t> &cluster_of,
                                 const std::vector<std::string> &stems, int T, int64_t *pairs_written)
{
    const size_t n_bins = stems.size();
    const Latch L = find_latch(fd, size);
    const size_t block = reader_block((size_t)1 << 20);
    std::vector<size_t> rb(T + 1);
    for (int t = 0; t <= T; ++t) rb[t] = (size_t)((unsigned __int128)size * (unsigned)t / (unsigned)T);
    std::vector<uint64_t> nl(T, 0);
    std::vector<char> at_line_start(T, 1), io_bad(T, 0);
    run_threads(T, [&](int t) {
        if (t > 0 && rb[t] > 0) { char c = 0; if (pread(fd, &c, 1, (off_t)(rb[t] - 1)) != 1) io_bad[t] = 1; at_line_start[t] = c == '\n'; }
        UnitReader rd(fd, rb[t], rb[t + 1], block);
        nl[t] = rd.count_newlines();
        if (rd.io_error()) io_bad[t] = 1;
    });
    for (char x : io_bad) if (x) return pg_fail(PG_EIO, "read error in %s", path);
    std::vector<uint64_t> nl_before(T + 1, 0);
    for (int t = 0; t < T; ++t) nl_before[t + 1] = nl_before[t] + nl[t];
    // one unit -> (cluster, rewritten header, barcode); false = not kept
    struct Out { std::vector<uint64_t> fq, bc; uint64_t bad_unit = UINT64_MAX; bool io_error = false; int64_t pairs = 0; };
    std::vector<Out> out(T);
    std::vector<int> fq_fd(n_bins, -1), bc_fd(n_bins, -1);
    auto walk = [&](int t, auto &&visit) {
        Out &o = out[t];
        if (rb[t] >= rb[t + 1]) return;
        UnitReader rd(fd, rb[t], size, block);
        uint64_t line = nl_before[t];
        if (!at_line_start[t]) { if (!rd.skip_line()) { o.io_error = rd.io_error(); return; } ++line; }
        for (uint64_t skip = (8 - line % 8) % 8; skip; --skip, ++line)
            if (!rd.skip_line()) { o.io_error = rd.io_error(); return; }
        uint64_t unit = line / 8;
        UnitLines u;
        std::string barcode;
        while (rd.offset() < rb[t + 1] && rd.next(u)) {
            int mode = mode_of(L, unit);
            Span nm, bc;
            if (!header_fields(u.p[0], u.n[0], mode, nm, bc)) { o.bad_unit = unit; return; }
            if (u.count == 8) {                                    // the sequential loop emits on the 8th line only
                barcode.assign(u.p[0] + bc.b, bc.n);
                auto it = cluster_of.find(barcode);
                if (it != cluster_of.end()) visit(it->second, u, nm, barcode);
            }
            ++unit;
        }
        o.io_error = rd.io_error();
    };
    // pass 1: bytes per (thread, cluster)
    run_threads(T, [&](int t) {
        out[t].fq.assign(n_bins, 0); out[t].bc.assign(n_bins, 0);
        walk(t, [&](uint32_t id, const UnitLines &u, const Span &nm, const std::string &barcode) {
            uint64_t bytes = nm.n + 6 + barcode.size() + 3;         // name \t BX:Z: barcode -1 \n
            for (int k = 1; k < 8; ++k) bytes += u.n[k] + 1;
            out[t].fq[id] += bytes;
            out[t].bc[id] += barcode.size() + 1;
            ++out[t].pairs;
        });
    });
    uint64_t first_bad = UINT64_MAX;
    int64_t written = 0;
    for (int t = 0; t < T; ++t) {
        if (out[t].io_error) return pg_fail(PG_EIO, "read error in %s", path);
        first_bad = std::min(first_bad, out[t].bad_unit);
        written += out[t].pairs;
    }
    if (first_bad != UINT64_MAX)
        return pg_fail(PG_EFORMAT, "%s line %llu: header ends inside its BX:Z tag", path, (unsigned long long)(first_bad * 8 + 1));
    // offsets of every thread's share, files sized up front
    std::vector<std::vector<uint64_t>> fq_at(T, std::vector<uint64_t>(n_bins)), bc_at(T, std::vector<uint64_t>(n_bins));
    int rc = PG_OK;
    for (size_t b = 0; b < n_bins && !rc; ++b) {
        uint64_t f = 0, c = 0;
        for (int t = 0; t < T; ++t) { fq_at[t][b] = f; f += out[t].fq[b]; bc_at[t][b] = c; c += out[t].bc[b]; }
        fq_fd[b] = open((stems[b] + ".fq").c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
        bc_fd[b] = open((stems[b] + ".barcode").c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
        if (fq_fd[b] < 0 || bc_fd[b] < 0 || ftruncate(fq_fd[b], (off_t)f) != 0 || ftruncate(bc_fd[b], (off_t)c) != 0)
            rc = pg_fail(PG_EIO, "cannot create %s.{fq,barcode}", stems[b].c_str());
    }
    // pass 2: the records, through per-cluster buffers, at their offsets
    std::vector<char> wr_bad(T, 0);
    if (!rc) run_threads(T, [&](int t) {
        std::vector<std::string> fq_buf(n_bins), bc_buf(n_bins);
        auto flush = [&](size_t b, bool all) {
            if (!fq_buf[b].empty() && (all || fq_buf[b].size() > ((size_t)1 << 18))) {
                if (pwrite(fq_fd[b], fq_buf[b].data(), fq_buf[b].size(), (off_t)fq_at[t][b]) != (ssize_t)fq_buf[b].size()) wr_bad[t] = 1;
                fq_at[t][b] += fq_buf[b].size(); fq_buf[b].clear();
            }
            if (!bc_buf[b].empty() && (all || bc_buf[b].size() > ((size_t)1 << 16))) {
                if (pwrite(bc_fd[b], bc_buf[b].data(), bc_buf[b].size(), (off_t)bc_at[t][b]) != (ssize_t)bc_buf[b].size()) wr_bad[t] = 1;
                bc_at[t][b] += bc_buf[b].size(); bc_buf[b].clear();
            }
        };
        walk(t, [&](uint32_t id, const UnitLines &u, const Span &nm, const std::string &barcode) {
            std::string &f = fq_buf[id];
            f.append(u.p[0] + nm.b, nm.n).append("\tBX:Z:").append(barcode).append("-1\n");
            for (int k = 1; k < 8; ++k) { f.append(u.p[k], u.n[k]); f.push_back('\n'); }
            bc_buf[id].append(barcode).push_back('\n');
            flush(id, false);
        });
        for (size_t b = 0; b < n_bins; ++b) flush(b, true);
    });
    for (size_t b = 0; b < n_bins; ++b) { if (fq_fd[b] >= 0) close(fq_fd[b]); if (bc_fd[b] >= 0) close(bc_fd[b]); }
    if (rc) return rc;
    for (char x : wr_bad) if (x) return pg_fail(PG_EIO, "write error in %s_bin*", path);
    if (pairs_written) *pairs_written = written;
    return PG_OK;
}

// The -1 / -2 form, byte for byte the files of the serial branch below (extract_reads.cpp:57-190, paired input): records cut by R1
// byte ranges and located in R2 as in ingest_paired_threaded, two passes over the records that are complete in both files
// (sizes per thread and cluster, then pwrite at exact offsets), the serial rules on whatever lies behind them, appended.
int extract_paired_parallel(int fd1, size_t n1, int fd2, size_t n2, const char *r1, const char *r2,
                            const std::unordered_map<std::string, uint32_t> &cluster_of, const std::vector<std::string> &stems, int T,
                            int64_t *pairs_written)
{
    const size_t n_bins = stems.size();
    const size_t block = reader_block((size_t)1 << 20);
    const int B2 = 16 * T;
    std::vector<uint64_t> nl1, nl2;
    uint64_t lines1 = 0, lines2 = 0;
    int rc;
    if ((rc = count_lines_blocks(fd1, n1, T, T, nl1, lines1, r1))) return rc;
    if ((rc = count_lines_blocks(fd2, n2, B2, T, nl2, lines2, r2))) return rc;
    const uint64_t full = std::min(lines1 / 4, lines2 / 4);
    PairLatch L{UINT64_MAX, MODE_UNSET};
    {
        UnitReader a(fd1, 0, n1, reader_block((size_t)1 << 16)), b(fd2, 0, n2, reader_block((size_t)1 << 16));
        UnitLines u, v;
        for (uint64_t r = 0; r < full && a.next(u, 4) && b.next(v, 4); ++r) {
            if (find_bxz(u.p[0], u.n[0]) != NPOS) { L = PairLatch{2 * r, MODE_10X}; break; }
            if (find_chr(u.p[0], u.n[0], '#', 0) != NPOS) { L = PairLatch{2 * r, MODE_STLFR}; break; }
            if (find_bxz(v.p[0], v.n[0]) != NPOS) { L = PairLatch{2 * r + 1, MODE_10X}; break; }
            if (find_chr(v.p[0], v.n[0], '#', 0) != NPOS) { L = PairLatch{2 * r + 1, MODE_STLFR}; break; }
        }
        if (a.io_error() || b.io_error()) return pg_fail(PG_EIO, "read error in %s", r1);
    }
    std::vector<uint64_t> rec0(T + 1, full);
    std::vector<char> at_line_start(T, 1);
    for (int t = 0; t < T; ++t) {
        const size_t a = (size_t)((unsigned __int128)n1 * (unsigned)t / (unsigned)T);
        if (t > 0 && a > 0) { char c = 0; if (pread(fd1, &c, 1, (off_t)(a - 1)) != 1) return pg_fail(PG_EIO, "read error in %s", r1); at_line_start[t] = c == '\n'; }
        const uint64_t line = nl1[t] + (at_line_start[t] ? 0 : 1);
        rec0[t] = std::min<uint64_t>((line + 3) / 4, full);
    }
    struct Out { std::vector<uint64_t> fq, bc; uint64_t bad_rec = UINT64_MAX; bool io_error = false; int64_t pairs = 0; };
    std::vector<Out> out(T);
    auto walk = [&](int t, auto &&visit) {
        Out &o = out[t];
        const uint64_t ra = rec0[t], rb = rec0[t + 1];
        if (ra >= rb) return;
        const size_t a = (size_t)((unsigned __int128)n1 * (unsigned)t / (unsigned)T);
        UnitReader rd1(fd1, a, n1, block);
        uint64_t line = nl1[t];
        if (!at_line_start[t]) { if (!rd1.skip_line()) { o.io_error = true; return; } ++line; }
        for (; line < 4 * ra; ++line) if (!rd1.skip_line()) { o.io_error = true; return; }
        size_t off2 = 0;
        if (offset_of_line(fd2, n2, B2, nl2, 4 * ra, off2, r2)) { o.io_error = true; return; }
        UnitReader rd2(fd2, off2, n2, block);
        UnitLines u, v;
        std::string barcode;
        for (uint64_t r = ra; r < rb; ++r) {
            if (!rd1.next(u, 4) || !rd2.next(v, 4) || u.count < 4 || v.count < 4) { o.io_error = true; return; }
            int m1 = mode_of(L, 2 * r), m2 = mode_of(L, 2 * r + 1);
            Span a1, c1, a2, c2;
            if (!header_fields(u.p[0], u.n[0], m1, a1, c1) || !header_fields(v.p[0], v.n[0], m2, a2, c2)) { o.bad_rec = r; return; }
            barcode.assign(u.p[0] + c1.b, c1.n);
            auto it = cluster_of.find(barcode);
            if (it != cluster_of.end() && same_span(u.p[0], a1, v.p[0], a2) && same_span(u.p[0], c1, v.p[0], c2))
                visit(it->second, u, v, a1, a2, barcode);
        }
        if (rd1.io_error() || rd2.io_error()) o.io_error = true;
    };
    run_threads(T, [&](int t) {
        out[t].fq.assign(n_bins, 0); out[t].bc.assign(n_bins, 0);
        walk(t, [&](uint32_t id, const UnitLines &u, const UnitLines &v, const Span &a1, const Span &a2, const std::string &barcode) {
            uint64_t bytes = a1.n + a2.n + 2 * (6 + barcode.size() + 3);
            for (int k = 1; k < 4; ++k) bytes += u.n[k] + 1 + v.n[k] + 1;
            out[t].fq[id] += bytes;
            out[t].bc[id] += barcode.size() + 1;
            ++out[t].pairs;
        });
    });
    uint64_t first_bad = UINT64_MAX;
    int64_t written = 0;
    for (int t = 0; t < T; ++t) {
        if (out[t].io_error) return pg_fail(PG_EIO, "read error in %s / %s", r1, r2);
        first_bad = std::min(first_bad, out[t].bad_rec);
        written += out[t].pairs;
    }
    if (first_bad != UINT64_MAX)
        return pg_fail(PG_EFORMAT, "%s line %llu: header ends inside its BX:Z tag", r1, (unsigned long long)(first_bad * 4 + 1));
    std::vector<std::vector<uint64_t>> fq_at(T, std::vector<uint64_t>(n_bins)), bc_at(T, std::vector<uint64_t>(n_bins));
    std::vector<int> fq_fd(n_bins, -1), bc_fd(n_bins, -1);
    std::vector<uint64_t> fq_end(n_bins, 0), bc_end(n_bins, 0);
    for (size_t b = 0; b < n_bins && !rc; ++b) {
        uint64_t f = 0, c = 0;
        for (int t = 0; t < T; ++t) { fq_at[t][b] = f; f += out[t].fq[b]; bc_at[t][b] = c; c += out[t].bc[b]; }
        fq_end[b] = f; bc_end[b] = c;
        fq_fd[b] = open((stems[b] + ".fq").c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
        bc_fd[b] = open((stems[b] + ".barcode").c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
        if (fq_fd[b] < 0 || bc_fd[b] < 0 || ftruncate(fq_fd[b], (off_t)f) != 0 || ftruncate(bc_fd[b], (off_t)c) != 0)
            rc = pg_fail(PG_EIO, "cannot create %s.{fq,barcode}", stems[b].c_str());
    }
    std::vector<char> wr_bad(T + 1, 0);
    if (!rc) run_threads(T, [&](int t) {
        std::vector<std::string> fq_buf(n_bins), bc_buf(n_bins);
        auto flush = [&](size_t b, bool all) {
            if (!fq_buf[b].empty() && (all || fq_buf[b].size() > ((size_t)1 << 18))) {
                if (pwrite(fq_fd[b], fq_buf[b].data(), fq_buf[b].size(), (off_t)fq_at[t][b]) != (ssize_t)fq_buf[b].size()) wr_bad[t] = 1;
                fq_at[t][b] += fq_buf[b].size(); fq_buf[b].clear();
            }
            if (!bc_buf[b].empty() && (all || bc_buf[b].size() > ((size_t)1 << 16))) {
                if (pwrite(bc_fd[b], bc_buf[b].data(), bc_buf[b].size(), (off_t)bc_at[t][b]) != (ssize_t)bc_buf[b].size()) wr_bad[t] = 1;
                bc_at[t][b] += bc_buf[b].size(); bc_buf[b].clear();
            }
        };
        walk(t, [&](uint32_t id, const UnitLines &u, const UnitLines &v, const Span &a1, const Span &a2, const std::string &barcode) {
            std::string &f = fq_buf[id];
            f.append(u.p[0] + a1.b, a1.n).append("\tBX:Z:").append(barcode).append("-1\n");
            for (int k = 1; k < 4; ++k) { f.append(u.p[k], u.n[k]); f.push_back('\n'); }
            f.append(v.p[0] + a2.b, a2.n).append("\tBX:Z:").append(barcode).append("-1\n");
            for (int k = 1; k < 4; ++k) { f.append(v.p[k], v.n[k]); f.push_back('\n'); }
            bc_buf[id].append(barcode).push_back('\n');
            flush(id, false);
        });
        for (size_t b = 0; b < n_bins; ++b) flush(b, true);
    });
    // the tail (a record cut short, an R2 of another length): the serial rules, appended behind the threads' shares
    if (!rc) {
        size_t a1 = 0, a2 = 0;
        if ((rc = offset_of_line(fd1, n1, T, nl1, 4 * full, a1, r1)) == 0 && (rc = offset_of_line(fd2, n2, B2, nl2, 4 * full, a2, r2)) == 0) {
            std::vector<char> t1(n1 - a1), t2(n2 - a2);
            if ((!t1.empty() && pread(fd1, t1.data(), t1.size(), (off_t)a1) != (ssize_t)t1.size()) ||
                (!t2.empty() && pread(fd2, t2.data(), t2.size(), (off_t)a2) != (ssize_t)t2.size()))
                rc = pg_fail(PG_EIO, "read error in %s", r1);
            int mode = L.header != UINT64_MAX && 2 * full >= L.header ? L.mode : (int)MODE_UNSET;
            Lines L1(t1.data(), t1.size()), L2(t2.data(), t2.size());
            uint64_t line_no = 0;
            bool keep = false;
            uint32_t id = 0;
            std::string rec1, rec2, barcode;
            const char *b, *c; size_t len, clen;
            while (!rc && L1.next(b, len)) {
                if (!L2.next(c, clen)) { c = ""; clen = 0; }
                const int ph = (int)(++line_no % 4);
                if (ph == 1) {
                    Span s1, b1, s2, b2;
                    if (!header_fields(b, len, mode, s1, b1) || !header_fields(c, clen, mode, s2, b2)) {
                        rc = pg_fail(PG_EFORMAT, "%s line %llu: header ends inside its BX:Z tag", r1, (unsigned long long)(4 * full + line_no));
                        break;
                    }
                    barcode.assign(b + b1.b, b1.n);
                    auto it = cluster_of.find(barcode);
                    keep = it != cluster_of.end() && same_span(b, s1, c, s2) && same_span(b, b1, c, b2);
                    rec1.clear(); rec2.clear();
                    if (keep) {
                        id = it->second;
                        rec1.append(b + s1.b, s1.n).append("\tBX:Z:").append(barcode).append("-1\n");
                        rec2.append(c + s2.b, s2.n).append("\tBX:Z:").append(barcode).append("-1\n");
                    }
                } else if (keep) {
                    rec1.append(b, len).push_back('\n');
                    rec2.append(c, clen).push_back('\n');
                    if (ph == 0) {
                        const std::string both = rec1 + rec2, bl = barcode + "\n";
                        if (pwrite(fq_fd[id], both.data(), both.size(), (off_t)fq_end[id]) != (ssize_t)both.size() ||
                            pwrite(bc_fd[id], bl.data(), bl.size(), (off_t)bc_end[id]) != (ssize_t)bl.size()) wr_bad[T] = 1;
                        fq_end[id] += both.size(); bc_end[id] += bl.size();
                        ++written;
                        rec1.clear(); rec2.clear();
                    }
                }
            }
        }
    }
    for (size_t b = 0; b < n_bins; ++b) { if (fq_fd[b] >= 0) close(fq_fd[b]); if (bc_fd[b] >= 0) close(bc_fd[b]); }
    if (rc) return rc;
    for (char x : wr_bad) if (x) return pg_fail(PG_EIO, "write error in %s_bin*", r1);
    if (pairs_written) *pairs_written = written;
    return PG_OK;
}

}  // namespace

extern "C" int pg_extract_reads(const char *r1, const char *r2, const char *clusters_tsv, const char *out_prefix, int64_t *pairs_written)
{
    if (!r1 || !clusters_tsv || !out_prefix) return pg_fail(PG_EINVAL, "pg_extract_reads: null argument");
    std::string tsv;
    {
        FILE *f = fopen(clusters_tsv, "rb");
        if (!f) return pg_fail(PG_EIO, "cannot open %s", clusters_tsv);
        char buf[1 << 16];
        size_t got;
        while ((got = fread(buf, 1, sizeof buf, f)) > 0) tsv.append(buf, got);
        fclose(f);
    }
    std::unordered_map<std::string, uint32_t> cluster_of;
    std::vector<BinFiles> bins;
    std::vector<std::string> stems;
    auto close_all = [&]() {
        for (auto &b : bins) { b.flush(); if (b.fq) fclose(b.fq); if (b.bc) fclose(b.bc); }
    };
    {
        Lines L(tsv.data(), tsv.size());
        const char *b; size_t len;
        while (L.next(b, len)) {
            size_t pos = find_chr(b, len, '\t', 0);
            const std::string label(b, pos == NPOS ? len : pos);
            if (label == "-1") continue;
            BinFiles files;
            const std::string stem = std::string(out_prefix) + "_bin" + label;
            files.bc = fopen((stem + ".barcode").c_str(), "wb");
            files.fq = fopen((stem + ".fq").c_str(), "wb");
            if (!files.bc || !files.fq) {
                if (files.bc) fclose(files.bc);
                if (files.fq) fclose(files.fq);
                close_all();
                return pg_fail(PG_EIO, "cannot create %s.{fq,barcode}", stem.c_str());
            }
            bins.push_back(std::move(files));
            stems.push_back(stem);
            const uint32_t id = (uint32_t)bins.size() - 1;
            // the reference walks `pos` with wrap-around when the line has no TAB (then the whole line is a barcode list)
            std::string bc;
            while (pos != len) {
                while (++pos < len && b[pos] != ',') bc.push_back(b[pos]);
                cluster_of[bc] = id;
                bc.clear();
            }
        }
    }
    const int T = ingest_threads();
    if (!r2 && T > 1) {               // an uncompressed interleaved file of some size: the threaded form (same bytes)
        int fd; size_t size = 0; bool plain;
        int rc0 = open_plain(r1, fd, size, plain);
        if (rc0) { close_all(); return rc0; }
        if (plain && size >= ((size_t)1 << 16) * (size_t)T) {
            close_all();                                             // the files exist (empty); the threads reopen them
            try { rc0 = extract_interleaved_parallel(fd, size, r1, cluster_of, stems, T, pairs_written); }
            catch (const std::bad_alloc &) { rc0 = pg_fail(PG_ENOMEM, "out of memory while writing the bins of %s", r1); }
            close(fd);
            return rc0;
        }
        close(fd);
    }
    if (r2 && T > 1) {                // two uncompressed files of some size: threaded as well
        int fd1, fd2; size_t n1 = 0, n2 = 0; bool p1, p2;
        int rc0 = open_plain(r1, fd1, n1, p1);
        if (rc0) { close_all(); return rc0; }
        if ((rc0 = open_plain(r2, fd2, n2, p2))) { close(fd1); close_all(); return rc0; }
        if (p1 && p2 && n1 >= ((size_t)1 << 16) * (size_t)T) {
            close_all();
            try { rc0 = extract_paired_parallel(fd1, n1, fd2, n2, r1, r2, cluster_of, stems, T, pairs_written); }
            catch (const std::bad_alloc &) { rc0 = pg_fail(PG_ENOMEM, "out of memory while writing the bins of %s", r1); }
            close(fd1); close(fd2);
            return rc0;
        }
        close(fd1); close(fd2);
    }
    FileBuf f1, f2;
    int rc = slurp(r1, f1);
    if (!rc && r2) rc = slurp(r2, f2);
    if (rc) { close_all(); return rc; }
    int64_t written = 0;
    int mode = MODE_UNSET;
    const char *b; size_t len;
    auto emit = [&](uint32_t id, const std::string &barcode, const std::string &rec1, const std::string &rec2) {
        BinFiles &o = bins[id];
        o.bc_buf.append(barcode).push_back('\n');
        o.fq_buf.append(rec1).append(rec2);
        if (o.fq_buf.size() > (1u << 20)) o.flush();
        ++written;
    };
    if (!r2) {
        Lines L(f1);
        uint64_t line_no = 0;
        bool keep = false;
        uint32_t id = 0;
        std::string rec, barcode;
        const std::string none;
        while (L.next(b, len)) {
            const int ph = (int)(++line_no % 8);
            if (ph == 1) {
                Span nm, bc;
                if (!header_fields(b, len, mode, nm, bc)) { close_all(); return pg_fail(PG_EFORMAT, "%s line %llu: header ends inside its BX:Z tag", r1, (unsigned long long)line_no); }
                barcode.assign(b + bc.b, bc.n);
                auto it = cluster_of.find(barcode);
                keep = it != cluster_of.end();
                if (keep) {
                    id = it->second;
                    rec.append(b + nm.b, nm.n).append("\tBX:Z:").append(barcode).append("-1\n");
                }
            } else if (keep) {
                rec.append(b, len).push_back('\n');
                if (ph == 0) { emit(id, barcode, rec, none); rec.clear(); }
            }
        }
    } else {
        Lines L1(f1), L2(f2);
        uint64_t line_no = 0;
        bool keep = false;
        uint32_t id = 0;
        std::string rec1, rec2, barcode;
        const char *c; size_t clen;
        while (L1.next(b, len)) {
            if (!L2.next(c, clen)) { c = ""; clen = 0; }
            const int ph = (int)(++line_no % 4);
            if (ph == 1) {
                Span n1, b1, n2, b2;
                if (!header_fields(b, len, mode, n1, b1) || !header_fields(c, clen, mode, n2, b2)) {
                    close_all();
                    return pg_fail(PG_EFORMAT, "%s line %llu: header ends inside its BX:Z tag", r1, (unsigned long long)line_no);
                }
                barcode.assign(b + b1.b, b1.n);
                auto it = cluster_of.find(barcode);
                keep = it != cluster_of.end() && n1.n == n2.n && memcmp(b + n1.b, c + n2.b, n1.n) == 0 &&
                       b1.n == b2.n && memcmp(b + b1.b, c + b2.b, b1.n) == 0;
                if (keep) {
                    id = it->second;
                    rec1.append(b + n1.b, n1.n).append("\tBX:Z:").append(barcode).append("-1\n");
                    rec2.append(c + n2.b, n2.n).append("\tBX:Z:").append(barcode).append("-1\n");
                }
            } else if (keep) {
                rec1.append(b, len).push_back('\n');
                rec2.append(c, clen).push_back('\n');
                if (ph == 0) { emit(id, barcode, rec1, rec2); rec1.clear(); rec2.clear(); }
            }
        }
    }
    close_all();
    if (pairs_written) *pairs_written = written;
    return PG_OK;
}
