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
    int64_t n = 0;          // characters written
    uint64_t cw = 0;        // word under construction
    uint32_t vw = 0;

    inline void put(unsigned char c)
    {
        // A C G T -> valid with code (c>>1)&3 ; everything else (N, lower case, IUPAC, '\r', separators) invalid
        const bool ok = (c == 'A') | (c == 'C') | (c == 'G') | (c == 'T');
        const int sh = (int)(n & 31);
        if (ok) {
            cw |= (uint64_t)((c >> 1) & 3) << (2 * sh);
            vw |= 1u << sh;
        }
        if (sh == 31) { codes.push_back(cw); valid.push_back(vw); cw = 0; vw = 0; }
        ++n;
    }
    void put_span(const char *s, size_t len)
    {
        for (size_t i = 0; i < len; ++i) put((unsigned char)s[i]);
    }
    void finish()
    {
        if (n & 31) { codes.push_back(cw); valid.push_back(vw); cw = 0; vw = 0; }
        size_t words = codes.size();
        size_t padded = (words + PG_WORD_ALIGN - 1) / PG_WORD_ALIGN * PG_WORD_ALIGN;
        if (padded == 0) padded = PG_WORD_ALIGN;
        codes.resize(padded, 0);
        valid.resize(padded, 0);
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
    StreamWriter st;
    std::vector<int64_t> run_off;      // [n_runs + 1]
    std::vector<std::string> run_name;
    int64_t n_pairs = 0, n_unpaired = 0;
    int mode = MODE_UNSET;
};

// ------------------------------------------------------------------------------------ parallel interleaved ingest
//
// Same result as the serial loop below, in phases that each split the in-memory file over T threads:
//   1. line index   : newline counts per byte chunk -> byte offset of any line number
//   2. latch scan   : first header that fixes the grammar ("BX:Z" -> 10x, else '#' -> stLFR); headers before it are
//                     parsed with the grammar still undecided, exactly as the sequential latch would
//   3. parse        : per 8-line unit: barcode span, sequence spans, character count
//   4. prefix       : character offset of every thread's first unit
//   5. pack         : 2-bit codes + validity written at their exact bit offsets (atomic OR only on the words two
//                     threads share)
//   6. runs         : barcode change points stitched across threads, in order (the append-then-compare rule)
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

struct Unit {                 // one 8-line record group of the interleaved file
    const char *bc; uint32_t bc_n;
    const char *s1; uint32_t s1_n;
    const char *s2; uint32_t s2_n;
    uint8_t have;             // bit0: line 2 present, bit1: line 6 present (pair complete), bit2: bad header
};

struct Packer {               // writes characters at absolute positions into shared word arrays
    uint64_t *codes; uint32_t *valid;
    int64_t pos, first_word, last_word;
    uint64_t cw = 0; uint32_t vw = 0;
    void flush()
    {
        const int64_t w = (pos - 1) >> 5;
        if (w == first_word || w == last_word) {
            __atomic_fetch_or(&codes[w], cw, __ATOMIC_RELAXED);
            __atomic_fetch_or(&valid[w], vw, __ATOMIC_RELAXED);
        } else {
            codes[w] = cw; valid[w] = vw;
        }
        cw = 0; vw = 0;
    }
    inline void put(unsigned char c)
    {
        const bool ok = (c == 'A') | (c == 'C') | (c == 'G') | (c == 'T');
        const int sh = (int)(pos & 31);
        if (ok) { cw |= (uint64_t)((c >> 1) & 3) << (2 * sh); vw |= 1u << sh; }
        ++pos;
        if (sh == 31) flush();
    }
    void finish() { if (pos & 31) flush(); }
};

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

int ingest_interleaved_parallel(const FileBuf &f, const char *path, pg_reads *R, int T)
{
    PhaseTimer tm;
    const char *base = f.data();
    const size_t n = f.size();
    // ---- 1. line index
    std::vector<size_t> cb(T + 1);
    for (int t = 0; t <= T; ++t) cb[t] = n * (size_t)t / T;
    std::vector<uint64_t> nl(T, 0);
    run_threads(T, [&](int t) {
        uint64_t c = 0;
        const char *p = base + cb[t], *e = base + cb[t + 1];
        while (p < e) { const char *q = (const char *)memchr(p, '\n', (size_t)(e - p)); if (!q) break; ++c; p = q + 1; }
        nl[t] = c;
    });
    std::vector<uint64_t> nl_before(T + 1, 0);
    for (int t = 0; t < T; ++t) nl_before[t + 1] = nl_before[t] + nl[t];
    const uint64_t n_lines = nl_before[T] + ((n && base[n - 1] != '\n') ? 1 : 0);
    const uint64_t n_units = (n_lines + 7) / 8;
    if (n_units == 0) { R->run_off.push_back(0); R->run_name.emplace_back(); R->st.finish(); return PG_OK; }
    // byte offset of the first character of line `ln` (0-based)
    auto line_start = [&](uint64_t ln) -> size_t {
        if (ln == 0) return 0;
        // the ln-th newline (1-based) ends line ln-1; find the chunk holding it
        int t = (int)(std::upper_bound(nl_before.begin(), nl_before.end(), ln - 1) - nl_before.begin()) - 1;
        uint64_t need = ln - nl_before[t];
        const char *p = base + cb[t], *e = base + cb[t + 1];
        while (need) { const char *q = (const char *)memchr(p, '\n', (size_t)(e - p)); p = q + 1; --need; }
        return (size_t)(p - base);
    };
    // units per thread
    std::vector<uint64_t> ub(T + 1);
    for (int t = 0; t <= T; ++t) ub[t] = n_units * (uint64_t)t / T;
    std::vector<size_t> ustart(T + 1);
    run_threads(T, [&](int t) { ustart[t] = ub[t] < n_units ? line_start(ub[t] * 8) : n; });
    ustart[T] = n;
    tm.lap("lines");
    // ---- 2. latch scan
    struct Latch { uint64_t unit; int mode; };
    std::vector<Latch> latch(T, Latch{UINT64_MAX, MODE_UNSET});
    run_threads(T, [&](int t) {
        const char *p = base + ustart[t], *e = base + n;
        for (uint64_t u = ub[t]; u < ub[t + 1] && p < e; ++u) {
            const char *q = (const char *)memchr(p, '\n', (size_t)(e - p));
            const size_t len = q ? (size_t)(q - p) : (size_t)(e - p);
            int m = MODE_UNSET;
            if (find_bxz(p, len) != NPOS) m = MODE_10X;
            else if (find_chr(p, len, '#', 0) != NPOS) m = MODE_STLFR;
            if (m != MODE_UNSET) { latch[t] = Latch{u, m}; return; }
            // skip the other 7 lines of the unit
            const char *r = q ? q + 1 : e;
            for (int k = 0; k < 7 && r < e; ++k) { const char *z = (const char *)memchr(r, '\n', (size_t)(e - r)); r = z ? z + 1 : e; }
            p = r;
        }
    });
    Latch L{UINT64_MAX, MODE_UNSET};
    for (int t = 0; t < T; ++t) if (latch[t].unit < L.unit) L = latch[t];
    tm.lap("latch");
    // ---- 3. parse
    std::vector<std::vector<Unit>> units(T);
    std::vector<int64_t> chars(T, 0), pairs(T, 0);
    std::vector<uint64_t> bad(T, UINT64_MAX);
    run_threads(T, [&](int t) {
        auto &U = units[t];
        U.reserve((size_t)(ub[t + 1] - ub[t]));
        const char *p = base + ustart[t], *e = base + n;
        int64_t c = 0, np = 0;
        for (uint64_t u = ub[t]; u < ub[t + 1]; ++u) {
            Unit x{nullptr, 0, nullptr, 0, nullptr, 0, 0};
            for (int k = 1; k <= 8 && p < e; ++k) {
                const char *q = (const char *)memchr(p, '\n', (size_t)(e - p));
                const size_t len = q ? (size_t)(q - p) : (size_t)(e - p);
                if (k == 1) {
                    int mode = u < L.unit ? MODE_UNSET : L.mode;      // the latching header decides for itself
                    Span nm, bc;
                    if (!header_fields(p, len, mode, nm, bc)) { x.have |= 4; if (bad[t] == UINT64_MAX) bad[t] = u; }
                    x.bc = p + bc.b; x.bc_n = (uint32_t)bc.n;
                } else if (k == 2) {
                    x.s1 = p; x.s1_n = (uint32_t)len; x.have |= 1; c += (int64_t)len + 1;
                } else if (k == 6) {
                    x.s2 = p; x.s2_n = (uint32_t)len; x.have |= 2; c += (int64_t)len + 1; ++np;
                }
                p = q ? q + 1 : e;
            }
            U.push_back(x);
        }
        chars[t] = c; pairs[t] = np;
    });
    uint64_t first_bad = UINT64_MAX;
    for (int t = 0; t < T; ++t) first_bad = std::min(first_bad, bad[t]);
    if (first_bad != UINT64_MAX)
        return pg_fail(PG_EFORMAT, "%s line %llu: header ends inside its BX:Z tag (the reference aborts here)", path, (unsigned long long)(first_bad * 8 + 1));
    tm.lap("parse");
    // ---- 4. prefix
    std::vector<int64_t> cstart(T + 1, 0);
    for (int t = 0; t < T; ++t) cstart[t + 1] = cstart[t] + chars[t];
    const int64_t total = cstart[T];
    {
        size_t words = (size_t)((total + 31) / 32);
        size_t padded = (words + PG_WORD_ALIGN - 1) / PG_WORD_ALIGN * PG_WORD_ALIGN;
        if (padded == 0) padded = PG_WORD_ALIGN;
        R->st.codes.assign(padded, 0);
        R->st.valid.assign(padded, 0);
        R->st.n = total;
    }
    tm.lap("alloc");
    // ---- 5. pack
    run_threads(T, [&](int t) {
        if (chars[t] == 0) return;
        Packer P{R->st.codes.data(), R->st.valid.data(), cstart[t], cstart[t] >> 5, (cstart[t + 1] - 1) >> 5};
        for (const Unit &x : units[t]) {
            if (x.have & 1) { for (uint32_t i = 0; i < x.s1_n; ++i) P.put((unsigned char)x.s1[i]); P.put('N'); }
            if (x.have & 2) { for (uint32_t i = 0; i < x.s2_n; ++i) P.put((unsigned char)x.s2[i]); P.put('N'); }
        }
        P.finish();
    });
    tm.lap("pack");
    // ---- 6. runs: every thread finds the barcode changes inside its units (the pair that differs from its predecessor
    // closes the predecessor's run); the first complete pair of a thread is compared with the last one of the threads
    // before it while stitching, in order
    R->mode = L.mode;
    struct Change { int64_t end_pos; const char *prev; uint32_t prev_n; };
    std::vector<std::vector<Change>> changes(T);
    struct Edge { const char *first = nullptr; uint32_t first_n = 0; int64_t first_end = 0; const char *last = nullptr; uint32_t last_n = 0; bool any = false; };
    std::vector<Edge> edge(T);
    run_threads(T, [&](int t) {
        int64_t pos = cstart[t];
        const char *last = nullptr; uint32_t last_n = 0;
        Edge e;
        for (const Unit &x : units[t]) {
            if (x.have & 1) pos += (int64_t)x.s1_n + 1;
            if (x.have & 2) {
                pos += (int64_t)x.s2_n + 1;
                if (!e.any) { e.any = true; e.first = x.bc; e.first_n = x.bc_n; e.first_end = pos; }
                else if (x.bc_n != last_n || (last_n && memcmp(x.bc, last, last_n) != 0)) changes[t].push_back(Change{pos, last, last_n});
                last = x.bc; last_n = x.bc_n;
            }
        }
        e.last = last; e.last_n = last_n;
        edge[t] = e;
    });
    R->run_off.push_back(0);
    const char *last = ""; uint32_t last_n = 0;
    for (int t = 0; t < T; ++t) {
        R->n_pairs += pairs[t];
        if (!edge[t].any) continue;
        if (edge[t].first_n != last_n || (last_n && memcmp(edge[t].first, last, last_n) != 0)) {
            R->run_off.push_back(edge[t].first_end);
            R->run_name.emplace_back(last, last_n);
        }
        for (const Change &c : changes[t]) { R->run_off.push_back(c.end_pos); R->run_name.emplace_back(c.prev, c.prev_n); }
        last = edge[t].last; last_n = edge[t].last_n;
    }
    R->run_off.push_back(total);
    R->run_name.emplace_back(last, last_n);
    tm.lap("runs");
    { std::vector<std::vector<Unit>>().swap(units); }
    tm.lap("free units");
    return PG_OK;
}

}  // namespace

extern "C" void pg_set_ingest_threads(int n) { g_ingest_threads = n > 0 ? n : 0; }

extern "C" int pg_ingest_fastq(const char *r1, const char *r2, pg_reads **out)
{
    if (!r1 || !out) return pg_fail(PG_EINVAL, "pg_ingest_fastq: null argument");
    *out = nullptr;
    FileBuf f1, f2;
    PhaseTimer tm_read;
    int rc = slurp(r1, f1);
    if (rc) return rc;
    if (r2 && (rc = slurp(r2, f2))) return rc;
    tm_read.lap("read");

    pg_reads *R = new (std::nothrow) pg_reads();
    if (!R) return pg_fail(PG_ENOMEM, "out of memory");
    try {
        R->st.codes.reserve(f1.size() / 64 + f2.size() / 64 + PG_WORD_ALIGN);
        R->st.valid.reserve(f1.size() / 64 + f2.size() / 64 + PG_WORD_ALIGN);
        R->run_off.push_back(0);
        std::string last, cur_bc;
        const char *b; size_t len;
        Span nm, bc;

        const int T = ingest_threads();
        if (!r2 && T > 1 && f1.size() >= ((size_t)1 << 16) * (size_t)T) {
            R->run_off.clear();
            rc = ingest_interleaved_parallel(f1, r1, R, T);
            if (rc) { delete R; return rc; }
            PhaseTimer tm_free;
            free(f1.p); f1.p = nullptr; f1.n = 0; f1.cap_ = 0;
            tm_free.lap("free file");
            *out = R;
            return PG_OK;
        }
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
            std::string n1, n2, b1, b2;
            std::vector<std::pair<const char *, size_t>> orphans;   // reads of skipped pairs (still counted globally)
            const char *c; size_t clen;
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
                    if (n1 != n2 || b1 != b2) {
                        R->n_unpaired++;
                        orphans.emplace_back(b, len);
                        orphans.emplace_back(c, clen);
                    } else {
                        R->st.put_span(b, len); R->st.put('N');
                        R->st.put_span(c, clen); R->st.put('N');
                        R->n_pairs++;
                        if (b1 != last) {
                            R->run_off.push_back(R->st.n);
                            R->run_name.push_back(last);
                            last = b1;
                        }
                    }
                    break;
                default: break;
                }
            }
            // R2 records beyond the end of R1 are still input of the global counter
            uint64_t l2 = line_no;
            while (L2.next(c, clen))
                if (++l2 % 4 == 2) orphans.emplace_back(c, clen);
            R->run_off.push_back(R->st.n);
            R->run_name.push_back(last);
            for (auto &o : orphans) { R->st.put_span(o.first, o.second); R->st.put('N'); }
        }
        R->st.finish();
    } catch (const std::bad_alloc &) {
        delete R;
        return pg_fail(PG_ENOMEM, "out of memory while ingesting %s", r1);
    }
    *out = R;
    return PG_OK;
}

extern "C" void pg_reads_free(pg_reads *r) { delete r; }
extern "C" int64_t pg_reads_n_chars(const pg_reads *r) { return r->st.n; }
extern "C" int64_t pg_reads_n_words(const pg_reads *r) { return (int64_t)r->st.codes.size(); }
extern "C" int64_t pg_reads_n_pairs(const pg_reads *r) { return r->n_pairs; }
extern "C" int64_t pg_reads_n_unpaired(const pg_reads *r) { return r->n_unpaired; }
extern "C" int64_t pg_reads_n_runs(const pg_reads *r) { return (int64_t)r->run_name.size(); }
extern "C" const uint64_t *pg_reads_codes(const pg_reads *r) { return r->st.codes.data(); }
extern "C" const uint32_t *pg_reads_valid(const pg_reads *r) { return r->st.valid.data(); }
extern "C" const int64_t *pg_reads_run_off(const pg_reads *r) { return r->run_off.data(); }
extern "C" const char *pg_reads_run_name(const pg_reads *r, int64_t i)
{
    if (i < 0 || i >= (int64_t)r->run_name.size()) return "";
    return r->run_name[(size_t)i].c_str();
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

extern "C" int pg_pack_ascii(const char *text, int64_t n_chars, uint64_t *codes, uint32_t *valid)
{
    if (n_chars < 0 || (n_chars > 0 && !text) || !codes || !valid) return pg_fail(PG_EINVAL, "pg_pack_ascii: bad arguments");
    const int64_t words = pg_words_for(n_chars);
    memset(codes, 0, (size_t)words * sizeof(uint64_t));
    memset(valid, 0, (size_t)words * sizeof(uint32_t));
    for (int64_t i = 0; i < n_chars; ++i) {
        const unsigned char c = (unsigned char)text[i];
        if (c == 'A' || c == 'C' || c == 'G' || c == 'T') {
            codes[i >> 5] |= (uint64_t)((c >> 1) & 3) << (2 * (i & 31));
            valid[i >> 5] |= 1u << (i & 31);
        }
    }
    return PG_OK;
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
