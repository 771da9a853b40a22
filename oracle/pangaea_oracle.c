/*
 * pangaea_oracle.c -- CPU restatement of Pangaea's barcode-aware k-mer feature path.
 *
 * ======================================================================================
 *  TEST INFRASTRUCTURE ONLY.  This file is the *checker*: only tests/, the smoke() entry
 *  and the cpu_baseline leg of bench.py may load liboracle.so.  Nothing under
 *  pangaea_amd/ links, imports or executes it; the product path is the HIP library and
 *  fails loudly when that library is missing.
 *
 *  Parity status: PINNED.  tests/test_oracle_golden.py checks this restatement against
 *  the CSV fixtures under tests/golden, which are outputs of the reference's own count_tnf / count_kmer
 *  binaries (compiled by `make -C oracle ref` from /root/reference/src/cpptools, see
 *  tests/golden/make_goldens.py), and -- when oracle/_ref exists -- against those
 *  binaries run live on randomised inputs.
 * ======================================================================================
 *
 * What is restated (reference file:line, all under /root/reference/src/cpptools):
 *   header -> (name, barcode)          count_tnf.cpp:23-52   == count_kmer.cpp:24-53
 *   run assembly, interleaved          count_tnf.cpp:238-289 == count_kmer.cpp:239-281
 *   run assembly, paired               count_tnf.cpp:174-231 == count_kmer.cpp:186-232
 *   reverse complement of a 2k-bit code count_tnf.cpp:10-20  == count_kmer.cpp:11-21
 *   TNF row of one run                 count_tnf.cpp:78-113  (+ column set :138-163)
 *   abundance row of one run           count_kmer.cpp:55-108
 *   jellyfish-dump loader              count_kmer.cpp:139-170
 *   CSV row emit (ostream<<double)     count_tnf.cpp:293-303 == count_kmer.cpp:283-292
 * plus an exact canonical k-mer counter standing in for `jellyfish count -C -m k` +
 * `jellyfish dump -c -t` (src/feature.py:94,103; jellyfish itself is an un-vendored
 * bioconda dependency, environment.yaml:14 -- its result is the mathematically exact
 * multiplicity of every canonical k-mer over every read, which is what is computed here).
 *
 * The code below is written from the behaviour of those functions, not transliterated:
 * one in-memory line table instead of stream getline, flat arrays instead of
 * std::map/unordered_map, explicit size_t arithmetic where the reference relies on
 * std::string::npos wrap-around.
 */
#define _GNU_SOURCE
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_NPOS ((size_t)-1)

/* ------------------------------------------------------------------ small utilities */

typedef struct {
    char *p;
    size_t n, cap;
} buf_t;

static int buf_reserve(buf_t *b, size_t extra)
{
    if (b->n + extra <= b->cap) return 0;
    size_t nc = b->cap ? b->cap : 4096;
    while (nc < b->n + extra) nc *= 2;
    char *q = (char *)realloc(b->p, nc);
    if (!q) return -1;
    b->p = q;
    b->cap = nc;
    return 0;
}

static int buf_add(buf_t *b, const void *src, size_t n)
{
    if (buf_reserve(b, n + 1)) return -1;
    memcpy(b->p + b->n, src, n);
    b->n += n;
    return 0;
}

static int buf_addc(buf_t *b, char c) { return buf_add(b, &c, 1); }

typedef struct {
    int64_t *p;
    size_t n, cap;
} ivec_t;

static int ivec_push(ivec_t *v, int64_t x)
{
    if (v->n == v->cap) {
        size_t nc = v->cap ? v->cap * 2 : 1024;
        int64_t *q = (int64_t *)realloc(v->p, nc * sizeof(int64_t));
        if (!q) return -1;
        v->p = q;
        v->cap = nc;
    }
    v->p[v->n++] = x;
    return 0;
}

/* whole file (gzip or plain -- zlib's gzread is transparent, as gzstream.C:61 gzopen) */
static int slurp(const char *path, buf_t *out)
{
    gzFile f = gzopen(path, "rb");
    if (!f) return -1;
    gzbuffer(f, 1 << 20);
    for (;;) {
        if (buf_reserve(out, (1 << 20) + 1)) { gzclose(f); return -1; }
        int got = gzread(f, out->p + out->n, 1 << 20);
        if (got < 0) { gzclose(f); return -1; }
        if (got == 0) break;
        out->n += (size_t)got;
    }
    gzclose(f);
    return 0;
}

/* line table with std::getline semantics: '\n' stripped, a final piece without '\n' is a
 * line iff it is non-empty ... except that getline also yields an empty last line when the
 * stream is still good, which only happens for "\n\n"-style content and is covered by the
 * split itself. */
typedef struct {
    const char *base;
    ivec_t start, len;
} lines_t;

static int split_lines(const buf_t *b, lines_t *L)
{
    memset(L, 0, sizeof *L);
    L->base = b->p;
    size_t i = 0;
    while (i < b->n) {
        const char *nl = (const char *)memchr(b->p + i, '\n', b->n - i);
        size_t e = nl ? (size_t)(nl - b->p) : b->n;
        if (ivec_push(&L->start, (int64_t)i) || ivec_push(&L->len, (int64_t)(e - i))) return -1;
        i = e + 1;
    }
    return 0;
}

static void lines_free(lines_t *L)
{
    free(L->start.p);
    free(L->len.p);
}

/* ------------------------------------------------------------------ header grammar */

enum { MODE_UNSET = 0, MODE_10X = 1, MODE_STLFR = 2 };

static size_t find_str(const char *s, size_t n, const char *pat, size_t from)
{
    size_t m = strlen(pat);
    if (from > n || m > n) return ORC_NPOS;
    for (size_t i = from; i + m <= n; ++i)
        if (memcmp(s + i, pat, m) == 0) return i;
    return ORC_NPOS;
}

static size_t find_chr(const char *s, size_t n, char c, size_t from)
{
    if (from >= n) return ORC_NPOS;
    const char *q = (const char *)memchr(s + from, c, n - from);
    return q ? (size_t)(q - s) : ORC_NPOS;
}

/* std::string::substr(pos, count): pos > size throws; count is clipped. */
static int substr_span(size_t n, size_t pos, size_t count, size_t *b, size_t *len)
{
    if (pos > n) return -1;
    size_t avail = n - pos;
    *b = pos;
    *len = count < avail ? count : avail;
    return 0;
}

/* (name, barcode) of a header line; the mode latches on the first header that decides it.
 * Returns 0, or -1 where the reference would throw std::out_of_range (header ending in a
 * bare "BX:Z"). */
static int header_fields(const char *s, size_t n, int *mode,
                         size_t *name_b, size_t *name_n, size_t *bc_b, size_t *bc_n)
{
    if (*mode == MODE_UNSET) {
        if (find_str(s, n, "BX:Z", 0) != ORC_NPOS) *mode = MODE_10X;
        else if (find_chr(s, n, '#', 0) != ORC_NPOS) *mode = MODE_STLFR;
    }
    *name_b = *name_n = *bc_b = *bc_n = 0;
    if (*mode == MODE_STLFR) {
        size_t p1 = find_chr(s, n, '#', 0);            /* may be npos: arithmetic wraps */
        size_t p2 = find_chr(s, n, '/', p1 + 1);
        if (substr_span(n, 0, p1, name_b, name_n)) return -1;
        if (substr_span(n, p1 + 1, p2 - p1 - 1, bc_b, bc_n)) return -1;
        if (*bc_n == 5 && memcmp(s + *bc_b, "0_0_0", 5) == 0) *bc_n = 0;
    } else {
        size_t e = ORC_NPOS;
        for (size_t i = 0; i < n; ++i)
            if (s[i] == ' ' || s[i] == '\r' || s[i] == '\t' || s[i] == '\n') { e = i; break; }
        substr_span(n, 0, e, name_b, name_n);
        size_t p1 = find_str(s, n, "BX:Z", 0);
        if (p1 != ORC_NPOS) {
            size_t p2 = find_chr(s, n, '-', p1 + 5);
            if (substr_span(n, p1 + 5, p2 - p1 - 5, bc_b, bc_n)) return -1;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------ run assembly */

typedef struct orc_reads {
    int64_t n_runs;      /* every enqueued run, in order, dropped ones included        */
    buf_t seq;           /* concatenated run strings (reads each followed by 'N')      */
    ivec_t seq_off;      /* [n_runs+1]                                                  */
    buf_t names;         /* barcode of each run, NUL-terminated, concatenated          */
    ivec_t name_off;     /* [n_runs+1]                                                  */
    buf_t all_seq;       /* every read of the file(s) + 'N': the jellyfish input       */
    int64_t n_pairs, n_unpaired;
    int mode;
} orc_reads;

static int push_run(orc_reads *R, const buf_t *cur, const char *bc, size_t bc_n)
{
    if (buf_add(&R->seq, cur->p ? cur->p : "", cur->n)) return -1;
    if (ivec_push(&R->seq_off, (int64_t)R->seq.n)) return -1;
    if (buf_add(&R->names, bc, bc_n) || buf_addc(&R->names, '\0')) return -1;
    if (ivec_push(&R->name_off, (int64_t)R->names.n)) return -1;
    R->n_runs++;
    return 0;
}

void orc_reads_free(orc_reads *R)
{
    if (!R) return;
    free(R->seq.p); free(R->seq_off.p); free(R->names.p); free(R->name_off.p);
    free(R->all_seq.p);
    free(R);
}

/* Parse one interleaved FASTQ (r2 == NULL) or an R1/R2 pair into runs.
 * The current pair is appended BEFORE the barcode comparison, so a run holds
 * pairs 2..n of its own barcode plus the first pair of the next barcode; the first pair
 * of the file lands in the leading ""-run. */
orc_reads *orc_parse_fastq(const char *r1, const char *r2)
{
    orc_reads *R = (orc_reads *)calloc(1, sizeof *R);
    buf_t f1 = {0}, f2 = {0}, cur = {0}, last = {0}, bc = {0};
    lines_t L1, L2;
    memset(&L1, 0, sizeof L1); memset(&L2, 0, sizeof L2);
    int ok = -1;
    if (!R) return NULL;
    if (ivec_push(&R->seq_off, 0) || ivec_push(&R->name_off, 0)) goto done;
    if (slurp(r1, &f1) || split_lines(&f1, &L1)) goto done;

    if (!r2) {
        for (size_t i = 0; i < L1.start.n; ++i) {
            const char *s = f1.p + L1.start.p[i];
            size_t n = (size_t)L1.len.p[i];
            switch ((i + 1) % 8) {
            case 1: {
                size_t nb, nn, bb, bn;
                if (header_fields(s, n, &R->mode, &nb, &nn, &bb, &bn)) goto done;
                bc.n = 0;
                if (buf_add(&bc, s + bb, bn)) goto done;
                break;
            }
            case 2:
                if (buf_add(&cur, s, n) || buf_addc(&cur, 'N')) goto done;
                if (buf_add(&R->all_seq, s, n) || buf_addc(&R->all_seq, 'N')) goto done;
                break;
            case 6:
                if (buf_add(&cur, s, n) || buf_addc(&cur, 'N')) goto done;
                if (buf_add(&R->all_seq, s, n) || buf_addc(&R->all_seq, 'N')) goto done;
                R->n_pairs++;
                if (bc.n != last.n || (bc.n && memcmp(bc.p, last.p, bc.n))) {
                    if (push_run(R, &cur, last.p ? last.p : "", last.n)) goto done;
                    last.n = 0;
                    if (buf_add(&last, bc.p ? bc.p : "", bc.n)) goto done;
                    cur.n = 0;
                }
                break;
            default: break;
            }
        }
    } else {
        if (slurp(r2, &f2) || split_lines(&f2, &L2)) goto done;
        buf_t n1 = {0}, n2 = {0}, b1 = {0}, b2 = {0};
        int bad = 0;
        /* jellyfish sees every sequence line of both files -- run with --min-qual-char=? in this branch (feature.py:76-83):
         * a base whose quality character is below '?' reads as N */
        for (int which = 0; which < 2; ++which) {
            const buf_t *f = which ? &f2 : &f1;
            const lines_t *L = which ? &L2 : &L1;
            for (size_t i = 1; i < L->start.n; i += 4) {
                const char *sq = f->p + L->start.p[i];
                const size_t sn = (size_t)L->len.p[i];
                const char *q = i + 2 < L->start.n ? f->p + L->start.p[i + 2] : "";
                const size_t qn = i + 2 < L->start.n ? (size_t)L->len.p[i + 2] : 0;
                for (size_t j = 0; j < sn; ++j)
                    if (buf_addc(&R->all_seq, j < qn && (unsigned char)q[j] < (unsigned char)'?' ? 'N' : sq[j])) bad = 1;
                if (buf_addc(&R->all_seq, 'N')) bad = 1;
            }
        }
        for (size_t i = 0; i < L1.start.n && !bad; ++i) {
            const char *s1 = f1.p + L1.start.p[i];
            size_t l1 = (size_t)L1.len.p[i];
            /* a short R2 yields empty lines (failed getline leaves "" behind) */
            const char *s2 = i < L2.start.n ? f2.p + L2.start.p[i] : "";
            size_t l2 = i < L2.start.n ? (size_t)L2.len.p[i] : 0;
            switch ((i + 1) % 4) {
            case 1: {
                size_t nb, nn, bb, bn;
                if (header_fields(s1, l1, &R->mode, &nb, &nn, &bb, &bn)) { bad = 1; break; }
                n1.n = b1.n = 0;
                if (buf_add(&n1, s1 + nb, nn) || buf_add(&b1, s1 + bb, bn)) bad = 1;
                if (header_fields(s2, l2, &R->mode, &nb, &nn, &bb, &bn)) { bad = 1; break; }
                n2.n = b2.n = 0;
                if (buf_add(&n2, s2 + nb, nn) || buf_add(&b2, s2 + bb, bn)) bad = 1;
                break;
            }
            case 2:
                if (n1.n != n2.n || memcmp(n1.p, n2.p, n1.n) ||
                    b1.n != b2.n || memcmp(b1.p, b2.p, b1.n)) {
                    R->n_unpaired++;
                } else {
                    if (buf_add(&cur, s1, l1) || buf_addc(&cur, 'N') ||
                        buf_add(&cur, s2, l2) || buf_addc(&cur, 'N')) { bad = 1; break; }
                    R->n_pairs++;
                    if (b1.n != last.n || (b1.n && memcmp(b1.p, last.p, b1.n))) {
                        if (push_run(R, &cur, last.p ? last.p : "", last.n)) { bad = 1; break; }
                        last.n = 0;
                        if (buf_add(&last, b1.p, b1.n)) bad = 1;
                        cur.n = 0;
                    }
                }
                break;
            default: break;
            }
        }
        free(n1.p); free(n2.p); free(b1.p); free(b2.p);
        if (bad) goto done;
    }
    /* trailing accumulator is always enqueued */
    if (push_run(R, &cur, last.p ? last.p : "", last.n)) goto done;
    ok = 0;
done:
    free(f1.p); free(f2.p); free(cur.p); free(last.p); free(bc.p);
    lines_free(&L1); lines_free(&L2);
    if (ok) { orc_reads_free(R); return NULL; }
    return R;
}

int64_t orc_reads_n_runs(const orc_reads *R) { return R->n_runs; }
int64_t orc_reads_n_pairs(const orc_reads *R) { return R->n_pairs; }
int64_t orc_reads_n_unpaired(const orc_reads *R) { return R->n_unpaired; }
int orc_reads_mode(const orc_reads *R) { return R->mode; }
const char *orc_reads_seq(const orc_reads *R) { return R->seq.p ? R->seq.p : ""; }
const int64_t *orc_reads_seq_off(const orc_reads *R) { return R->seq_off.p; }
const char *orc_reads_name(const orc_reads *R, int64_t i) { return R->names.p + R->name_off.p[i]; }
const char *orc_reads_all_seq(const orc_reads *R) { return R->all_seq.p ? R->all_seq.p : ""; }
int64_t orc_reads_all_len(const orc_reads *R) { return (int64_t)R->all_seq.n; }

/* a run yields an output row iff its barcode is non-empty and its string is longer than
 * min_len (the string includes one 'N' per read) */
int orc_run_survives(const orc_reads *R, int64_t i, int min_len)
{
    int64_t len = R->seq_off.p[i + 1] - R->seq_off.p[i];
    return orc_reads_name(R, i)[0] != '\0' && !(len <= (int64_t)min_len);
}

/* ------------------------------------------------------------------ k-mer codes */

static inline int is_base(unsigned char c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; }
static inline uint64_t base_code(unsigned char c) { return (uint64_t)((c >> 1) & 3); } /* A0 C1 T2 G3 */

/* reverse complement of a 2k-bit code: reverse the 2-bit digits, complement = digit ^ 2 */
uint64_t orc_revcomp(uint64_t x, int k)
{
    uint64_t r = 0;
    for (int i = 0; i < k; ++i) {
        r = (r << 2) | ((x & 3) ^ 2);
        x >>= 2;
    }
    return r;
}

static inline uint64_t kmask(int k) { return k >= 32 ? ~0ULL : ((1ULL << (2 * k)) - 1); }

/* incremental forward / reverse-complement pair */
typedef struct {
    uint64_t fw, rc, mask;
    int k, run, shift;
} roll_t;

static inline void roll_init(roll_t *r, int k)
{
    r->fw = r->rc = 0; r->run = 0; r->k = k; r->mask = kmask(k); r->shift = 2 * (k - 1);
}

/* feed one character; returns 1 when a full k-mer ends here (canon in *out) */
static inline int roll_feed(roll_t *r, unsigned char c, uint64_t *out)
{
    if (!is_base(c)) { r->fw = r->rc = 0; r->run = 0; return 0; }
    uint64_t d = base_code(c);
    r->fw = ((r->fw << 2) & r->mask) | d;
    r->rc = (r->rc >> 2) | ((d ^ 2) << r->shift);
    if (r->run < r->k) r->run++;
    if (r->run < r->k) return 0;
    *out = r->fw < r->rc ? r->fw : r->rc;
    return 1;
}

/* ------------------------------------------------------------------ TNF */

#define ORC_TNF_MAXK 12

int orc_tnf_ncols(int k)
{
    if (k < 1 || k > ORC_TNF_MAXK) return -1;
    int n = 0;
    for (uint64_t c = 0; c < (1ULL << (2 * k)); ++c)
        if (c <= orc_revcomp(c, k)) ++n;
    return n;
}

/* ascending canonical codes == iteration order of the reference's std::map */
int orc_tnf_columns(int k, uint32_t *codes)
{
    if (k < 1 || k > ORC_TNF_MAXK) return -1;
    int n = 0;
    for (uint64_t c = 0; c < (1ULL << (2 * k)); ++c)
        if (c <= orc_revcomp(c, k)) codes[n++] = (uint32_t)c;
    return n;
}

static int32_t *tnf_colmap(int k)
{
    size_t n = (size_t)1 << (2 * k);
    int32_t *m = (int32_t *)malloc(n * sizeof(int32_t));
    if (!m) return NULL;
    int col = 0;
    for (uint64_t c = 0; c < n; ++c) m[c] = (c <= orc_revcomp(c, k)) ? col++ : -1;
    return m;
}

static void tnf_row_with_map(const char *seq, int64_t n, int k, const int32_t *colmap, int64_t *out, int ncols)
{
    memset(out, 0, (size_t)ncols * sizeof(int64_t));
    roll_t r;
    roll_init(&r, k);
    uint64_t canon;
    for (int64_t i = 0; i < n; ++i)
        if (roll_feed(&r, (unsigned char)seq[i], &canon)) out[colmap[canon]]++;
}

int orc_tnf_row(const char *seq, int64_t n, int k, int64_t *out)
{
    int ncols = orc_tnf_ncols(k);
    if (ncols < 0) return -1;
    int32_t *m = tnf_colmap(k);
    if (!m) return -1;
    tnf_row_with_map(seq, n, k, m, out, ncols);
    free(m);
    return ncols;
}

/* ------------------------------------------------------------------ exact global counter */

typedef struct {
    uint64_t *keys;   /* canon + 1, 0 = empty */
    uint64_t *vals;
    uint64_t cap, n;  /* cap is a power of two */
} sub_t;

typedef struct orc_table {
    int k, nsub;
    sub_t *sub;
} orc_table;

static inline uint64_t mix64(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}

static int sub_init(sub_t *s, uint64_t cap)
{
    s->keys = (uint64_t *)calloc(cap, sizeof(uint64_t));
    s->vals = (uint64_t *)calloc(cap, sizeof(uint64_t));
    s->cap = cap; s->n = 0;
    return (s->keys && s->vals) ? 0 : -1;
}

static uint64_t *sub_slot(sub_t *s, uint64_t key1, uint64_t h, int create);

static int sub_grow(sub_t *s)
{
    sub_t t;
    if (sub_init(&t, s->cap * 2)) return -1;
    for (uint64_t i = 0; i < s->cap; ++i)
        if (s->keys[i]) {
            uint64_t *v = sub_slot(&t, s->keys[i], mix64(s->keys[i] - 1), 1);
            *v = s->vals[i];
        }
    free(s->keys); free(s->vals);
    *s = t;
    return 0;
}

static uint64_t *sub_slot(sub_t *s, uint64_t key1, uint64_t h, int create)
{
    uint64_t m = s->cap - 1, i = (h >> 8) & m;
    for (;;) {
        if (s->keys[i] == key1) return &s->vals[i];
        if (!s->keys[i]) {
            if (!create) return NULL;
            s->keys[i] = key1; s->n++;
            return &s->vals[i];
        }
        i = (i + 1) & m;
    }
}

orc_table *orc_table_new(int k, int nsub)
{
    if (k < 1 || k > 31) return NULL;
    if (nsub < 1) nsub = 1;
    if (nsub > 255) nsub = 255;
    orc_table *T = (orc_table *)calloc(1, sizeof *T);
    if (!T) return NULL;
    T->k = k; T->nsub = nsub;
    T->sub = (sub_t *)calloc((size_t)nsub, sizeof(sub_t));
    for (int i = 0; i < nsub; ++i)
        if (sub_init(&T->sub[i], 1 << 12)) return NULL;
    return T;
}

void orc_table_free(orc_table *T)
{
    if (!T) return;
    for (int i = 0; i < T->nsub; ++i) { free(T->sub[i].keys); free(T->sub[i].vals); }
    free(T->sub); free(T);
}

static inline int sub_of(const orc_table *T, uint64_t h) { return (int)((h & 0xff) % (uint64_t)T->nsub); }

static int table_add(orc_table *T, int s, uint64_t canon, uint64_t h, uint64_t by, int assign)
{
    sub_t *S = &T->sub[s];
    if ((S->n + 1) * 10 > S->cap * 6 && sub_grow(S)) return -1;
    uint64_t *v = sub_slot(S, canon + 1, h, 1);
    if (assign) *v = by; else *v += by;
    return 0;
}

/* count every k-mer of `seq` (reads separated by non-base characters).  Every read is
 * scanned independently of barcodes.  lowercase_is_base=1 mimics jellyfish, which also
 * accepts acgt; Pangaea's own counters never do (count_kmer.cpp:73-78). */
int orc_table_count_seq(orc_table *T, const char *seq, int64_t n, int lowercase_is_base)
{
    int fail = 0;
#pragma omp parallel num_threads(T->nsub) reduction(| : fail)
    {
#ifdef _OPENMP
        int me = omp_get_thread_num(), nt = omp_get_num_threads();
#else
        int me = 0, nt = 1;
#endif
        /* every thread scans the whole text and keeps only the k-mers it owns */
        for (int s = me; s < T->nsub; s += nt) {
            roll_t r;
            roll_init(&r, T->k);
            uint64_t canon;
            for (int64_t i = 0; i < n; ++i) {
                unsigned char c = (unsigned char)seq[i];
                if (lowercase_is_base && c >= 'a' && c <= 'z') c = (unsigned char)(c - 32);
                if (roll_feed(&r, c, &canon)) {
                    uint64_t h = mix64(canon);
                    if (sub_of(T, h) == s && table_add(T, s, canon, h, 1, 0)) fail = 1;
                }
            }
        }
    }
    return fail ? -1 : 0;
}

/* the same scan, but only k-mers that are ALREADY keys of the table are counted (nothing is created): with the keys of a few
 * rows set to 0 beforehand this gives their exact global multiplicities over a text far too large to count whole --
 * how the full-size tests check abundance rows against the reference's definition (count_kmer.cpp:86-96) */
int orc_table_count_known(orc_table *T, const char *seq, int64_t n, int lowercase_is_base)
{
#pragma omp parallel num_threads(T->nsub)
    {
#ifdef _OPENMP
        int me = omp_get_thread_num(), nt = omp_get_num_threads();
#else
        int me = 0, nt = 1;
#endif
        for (int s = me; s < T->nsub; s += nt) {
            roll_t r;
            roll_init(&r, T->k);
            uint64_t canon;
            for (int64_t i = 0; i < n; ++i) {
                unsigned char c = (unsigned char)seq[i];
                if (lowercase_is_base && c >= 'a' && c <= 'z') c = (unsigned char)(c - 32);
                if (roll_feed(&r, c, &canon)) {
                    uint64_t h = mix64(canon);
                    if (sub_of(T, h) == s) {
                        uint64_t *v = sub_slot(&T->sub[s], canon + 1, h, 0);
                        if (v) ++*v;
                    }
                }
            }
        }
    }
    return 0;
}

int64_t orc_table_size(const orc_table *T)
{
    int64_t n = 0;
    for (int i = 0; i < T->nsub; ++i) n += (int64_t)T->sub[i].n;
    return n;
}

/* returns the count, 0 when the k-mer is absent (*found tells which) */
uint64_t orc_table_get(const orc_table *T, uint64_t canon, int *found)
{
    uint64_t h = mix64(canon);
    sub_t *S = &T->sub[sub_of(T, h)];
    uint64_t *v = sub_slot(S, canon + 1, h, 0);
    if (found) *found = v != NULL;
    return v ? *v : 0;
}

void orc_table_export(const orc_table *T, uint64_t *keys, uint64_t *vals)
{
    int64_t j = 0;
    for (int s = 0; s < T->nsub; ++s)
        for (uint64_t i = 0; i < T->sub[s].cap; ++i)
            if (T->sub[s].keys[i]) { keys[j] = T->sub[s].keys[i] - 1; vals[j] = T->sub[s].vals[i]; ++j; }
}

int orc_table_set(orc_table *T, uint64_t canon, uint64_t count)
{
    uint64_t h = mix64(canon);
    return table_add(T, sub_of(T, h), canon, h, count, 1);
}

/* `jellyfish dump -c -t` text: "<KMER>\t<COUNT>\n", one line per distinct canonical k-mer.
 * The k-mer is spelled from the A0 C1 T2 G3 code; which strand is printed is immaterial
 * because the loader re-canonicalises (count_kmer.cpp:166). */
int orc_table_dump(const orc_table *T, const char *path)
{
    static const char L[4] = {'A', 'C', 'T', 'G'};
    FILE *f = fopen(path, "w");
    if (!f) return -1;
    char km[40];
    for (int s = 0; s < T->nsub; ++s)
        for (uint64_t i = 0; i < T->sub[s].cap; ++i)
            if (T->sub[s].keys[i]) {
                uint64_t c = T->sub[s].keys[i] - 1;
                for (int j = 0; j < T->k; ++j) km[j] = L[(c >> (2 * (T->k - 1 - j))) & 3];
                km[T->k] = 0;
                fprintf(f, "%s\t%llu\n", km, (unsigned long long)T->sub[s].vals[i]);
            }
    return fclose(f) ? -1 : 0;
}

/* the reference's dump loader: roll over the text before the first TAB and assign the
 * count to every k-mer completed inside it (later lines overwrite earlier ones) */
orc_table *orc_table_load_dump(const char *path, int k, int nsub)
{
    orc_table *T = orc_table_new(k, nsub);
    FILE *f = fopen(path, "r");
    if (!T || !f) { if (f) fclose(f); orc_table_free(T); return NULL; }
    char *line = NULL;
    size_t cap = 0;
    ssize_t got;
    while ((got = getline(&line, &cap, f)) >= 0) {
        size_t n = (size_t)got;
        if (n && line[n - 1] == '\n') --n;
        size_t tab = find_chr(line, n, '\t', 0);
        size_t kn = tab == ORC_NPOS ? n : tab;
        /* std::stol(line.substr(pos + 1)): with no TAB, pos+1 wraps to 0 */
        const char *num = tab == ORC_NPOS ? line : line + tab + 1;
        char tmp[64];
        size_t nn = (size_t)((line + n) - num);
        if (nn >= sizeof tmp) nn = sizeof tmp - 1;
        memcpy(tmp, num, nn); tmp[nn] = 0;
        uint64_t cnt = (uint64_t)strtol(tmp, NULL, 10);
        roll_t r;
        roll_init(&r, k);
        uint64_t canon;
        for (size_t i = 0; i < kn; ++i)
            if (roll_feed(&r, (unsigned char)line[i], &canon)) orc_table_set(T, canon, cnt);
    }
    free(line);
    fclose(f);
    return T;
}

/* ------------------------------------------------------------------ abundance row */

/* hist[count / window]++ for every k-mer occurrence whose canonical form is in the table
 * and whose bin is < vsize (count / window is an unsigned division, then narrowed to int) */
int orc_abd_row(const char *seq, int64_t n, int k, const orc_table *T, int window, int vsize, int64_t *out)
{
    if (!T || T->k != k || window <= 0 || vsize <= 0) return -1;
    memset(out, 0, (size_t)vsize * sizeof(int64_t));
    roll_t r;
    roll_init(&r, k);
    uint64_t canon;
    for (int64_t i = 0; i < n; ++i)
        if (roll_feed(&r, (unsigned char)seq[i], &canon)) {
            int found;
            uint64_t c = orc_table_get(T, canon, &found);
            if (!found) continue;
            int pos = (int)(c / (uint64_t)window);
            if (pos < vsize) out[pos]++;
        }
    return 0;
}

/* ------------------------------------------------------------------ whole-file features */

/* rows for every surviving run, in run order.  tnf_out [n_rows, ncols(k_tnf)] and/or
 * abd_out [n_rows, vsize] may be NULL.  row_run[n_rows] receives the run index of each
 * row.  Returns n_rows (call with all outputs NULL to size them). */
int64_t orc_features(const orc_reads *R, int min_len, int k_tnf, int64_t *tnf_out,
                     int k_abd, const orc_table *T, int window, int vsize, int64_t *abd_out,
                     int64_t *row_run, int threads)
{
    int64_t n_rows = 0;
    int64_t *rows = (int64_t *)malloc((size_t)(R->n_runs + 1) * sizeof(int64_t));
    if (!rows) return -1;
    for (int64_t i = 0; i < R->n_runs; ++i)
        if (orc_run_survives(R, i, min_len)) rows[n_rows++] = i;
    if (row_run) memcpy(row_run, rows, (size_t)n_rows * sizeof(int64_t));
    int ncols = tnf_out ? orc_tnf_ncols(k_tnf) : 0;
    int32_t *cm = tnf_out ? tnf_colmap(k_tnf) : NULL;
    int bad = (tnf_out && (!cm || ncols < 0)) ? 1 : 0;
    if (threads < 1) threads = 1;
    if (!bad) {
#pragma omp parallel for schedule(dynamic, 8) num_threads(threads)
        for (int64_t j = 0; j < n_rows; ++j) {
            const char *s = R->seq.p + R->seq_off.p[rows[j]];
            int64_t n = R->seq_off.p[rows[j] + 1] - R->seq_off.p[rows[j]];
            if (tnf_out) tnf_row_with_map(s, n, k_tnf, cm, tnf_out + j * ncols, ncols);
            if (abd_out) orc_abd_row(s, n, k_abd, T, window, vsize, abd_out + j * (int64_t)vsize);
        }
    }
    free(cm);
    free(rows);
    return bad ? -1 : n_rows;
}

/* ------------------------------------------------------------------ CSV (gz) */

/* one row per line: name,v1,...,vD ; numbers as `ostream << double` prints them (%g, six
 * significant digits).  `names` holds n_rows NUL-terminated strings back to back. */
int orc_write_csv_gz(const char *path, const char *names, const int64_t *mat, int64_t n_rows, int64_t n_cols)
{
    gzFile f = gzopen(path, "wb");
    if (!f) return -1;
    const char *nm = names;
    char num[64];
    for (int64_t i = 0; i < n_rows; ++i) {
        gzputs(f, nm);
        nm += strlen(nm) + 1;
        for (int64_t j = 0; j < n_cols; ++j) {
            int m = snprintf(num, sizeof num, ",%g", (double)mat[i * n_cols + j]);
            gzwrite(f, num, (unsigned)m);
        }
        gzputc(f, '\n');
    }
    return gzclose(f) == Z_OK ? 0 : -1;
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
