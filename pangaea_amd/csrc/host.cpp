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
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <string>
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

// whole-file decode; zlib reads gzip and plain files alike
int slurp(const char *path, std::string &out)
{
    gzFile f = gzopen(path, "rb");
    if (!f) return pg_fail(PG_EIO, "cannot open %s", path);
    gzbuffer(f, 1 << 22);
    std::vector<char> buf(1 << 22);
    for (;;) {
        int got = gzread(f, buf.data(), (unsigned)buf.size());
        if (got < 0) { gzclose(f); return pg_fail(PG_EIO, "read error in %s", path); }
        if (got == 0) break;
        out.append(buf.data(), (size_t)got);
    }
    gzclose(f);
    return PG_OK;
}

// getline-style cursor over an in-memory file
struct Lines {
    const char *p, *end;
    explicit Lines(const std::string &s) : p(s.data()), end(s.data() + s.size()) {}
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

extern "C" int pg_ingest_fastq(const char *r1, const char *r2, pg_reads **out)
{
    if (!r1 || !out) return pg_fail(PG_EINVAL, "pg_ingest_fastq: null argument");
    *out = nullptr;
    std::string f1, f2;
    int rc = slurp(r1, f1);
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

extern "C" int pg_write_csv_gz(const char *path, const char *names, const int32_t *mat, int64_t n_rows, int64_t n_cols)
{
    if (!path || n_rows < 0 || n_cols < 0 || (n_rows > 0 && (!names || (n_cols > 0 && !mat))))
        return pg_fail(PG_EINVAL, "pg_write_csv_gz: bad arguments");
    gzFile f = gzopen(path, "wb1");
    if (!f) return pg_fail(PG_EIO, "cannot create %s", path);
    gzbuffer(f, 1 << 20);
    std::string line;
    char num[48];
    const char *nm = names;
    for (int64_t i = 0; i < n_rows; ++i) {
        line.assign(nm);
        nm += line.size() + 1;
        for (int64_t j = 0; j < n_cols; ++j) {
            // the reference streams each count through ostream<<double: %g with 6 significant digits
            int m = snprintf(num, sizeof num, ",%g", (double)mat[i * n_cols + j]);
            line.append(num, (size_t)m);
        }
        line.push_back('\n');
        if (gzwrite(f, line.data(), (unsigned)line.size()) != (int)line.size()) {
            gzclose(f);
            return pg_fail(PG_EIO, "write error on %s", path);
        }
    }
    if (gzclose(f) != Z_OK) return pg_fail(PG_EIO, "close error on %s", path);
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
        Lines L(tsv);
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
    std::string f1, f2;
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
