/*
 * pangaea_feat.h -- C ABI of libpangaea_feat.so, the MI355X (gfx950) implementation of Pangaea's
 * barcode-aware k-mer feature path.
 *
 * The reference has no in-process FFI on this path: src/feature.py shells out to two binaries
 * (count_tnf, count_kmer) and to jellyfish, and reads their CSV files back.  This header is the
 * boundary a maintainer binds instead (ctypes stub in INTEGRATION.md).  Every entry point names the
 * reference interface it replaces.  Conventions:
 *   - plain pointers and sizes only; no C++ or torch types;
 *   - every function returns PG_OK (0) or a negative pg_status; pg_last_error() gives the message of the
 *     last failure on the calling thread; nothing here calls exit();
 *   - "device" pointers are HIP device pointers owned by the caller (e.g. torch tensors' data_ptr());
 *     `stream` is a hipStream_t passed as void* (NULL = the null stream).  Device functions only enqueue
 *     work on `stream`; they never allocate, free or synchronise;
 *   - host handles (pg_reads) are owned by the library until the matching free.
 *
 * Read-stream layout (host and device): the text the reference accumulates per barcode run --
 * read1 + 'N' + read2 + 'N' per pair (count_tnf.cpp:248,251) -- is kept for the WHOLE file as one
 * character stream in file order.  Character j lives in
 *     codes[j / 32] bits [2*(j%32), 2*(j%32)+2)   A=0 C=1 T=2 G=3   (count_tnf.cpp:99: (c>>1)&3)
 *     valid[j / 32] bit  (j % 32)                  1 iff the character is one of 'A','C','G','T'
 * Separators, N, lower-case and IUPAC characters have valid=0 (and code 0).  Arrays are padded with
 * invalid characters to a whole number of PG_WORD_ALIGN words.  A barcode run is a contiguous
 * character range of the stream, so rows never need per-read offsets.
 */
#ifndef PANGAEA_FEAT_H
#define PANGAEA_FEAT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PG_ABI_VERSION 9   /* 9: pg_ingest_fastq_pair_device / pg_ingest_place_pair (-1 / -2 input with the device copy inside);
                              8: pg_inflate_to_memfd (gzip input for the ingest with the device copy inside);
                              7: pg_build_flags (what kind of build a loaded library is), pg_mini_gather_entries names its buffer's size,
                                 pg_mini_shuffle_bytes_merged, a stream counted in pieces (pg_mini_count_piece / pg_mini_lookup_*);
                              6: pg_ingest_fastq_device / pg_ingest_place (ingest with the device copy inside);
                              5: pg_mini_count takes the merged lookups' word buffer (pg_mini_merge_words), the multi-rank half entries;
                              4: pg_mini_records_bytes takes the table, status bits;
                              3: PG_TABLE_MINI and the super-k-mer entry points; 2: packed hash slots keyed by pg_key42(code) */
#define PG_CHARS_PER_WORD 32
#define PG_WORD_ALIGN 256 /* stream arrays are padded to a multiple of this many words */

typedef enum {
    PG_OK = 0,
    PG_EINVAL = -1,      /* bad argument (unsupported k, null pointer, ...) */
    PG_EIO = -2,         /* cannot open / read / write a file */
    PG_EFORMAT = -3,     /* input the reference itself would abort on (header ending in a bare "BX:Z") */
    PG_ENOMEM = -4,
    PG_EHIP = -5,        /* a HIP runtime call or kernel launch failed */
    PG_ETABLEFULL = -6,  /* hash table too small: enlarge log2_slots and count again */
    PG_ENODEVICE = -7    /* no gfx950 device / code object not loadable */
} pg_status;

int pg_abi_version(void);
/* What kind of build this library is -- 0 for the product (libpangaea_feat.so as `make` builds it).  A loader refuses anything
 * else unless it asked for it: a checked library is slower, a variant may compute WRONG results (timing experiments).
 *   PG_BUILD_CHECKED  the super-k-mer kernels check every global store against its buffer's capacity (PG_STATUS_BOUNDS)
 *   PG_BUILD_STAMPS   phase timers inside the bucket kernels (-DPG_MINI_STAMPS)
 *   PG_BUILD_VARIANT  built with extra compiler flags (`make variant NAME=... KFLAGS=...`): not the product, whatever it does */
#define PG_BUILD_CHECKED 1u
#define PG_BUILD_STAMPS 2u
#define PG_BUILD_VARIANT 4u
uint32_t pg_build_flags(void);
const char *pg_last_error(void);
/* number of visible HIP devices, or a negative pg_status */
int pg_device_count(void);

/* ----------------------------------------------------------------------------------------------
 * Host ingest: FASTQ -> read stream + barcode runs.
 * Replaces the single-threaded producer loops of count_tnf.cpp:174-289 and count_kmer.cpp:186-281
 * together with getBarcode (count_tnf.cpp:23-52).  r2 == NULL selects the interleaved form (-i),
 * otherwise r1/r2 are the -1/-2 files.  gzip or plain, as gzstream's gzopen.
 * Runs are assembled exactly as the reference does (the pair is appended before the barcode
 * comparison), so run i = [run_off[i], run_off[i+1]) in characters.  Reads that jellyfish counts but
 * that belong to no run (paired mode, name/barcode mismatch, count_tnf.cpp:183) are placed after the
 * last run, in [run_off[n_runs], n_chars).
 * ---------------------------------------------------------------------------------------------- */
typedef struct pg_reads pg_reads;

int pg_ingest_fastq(const char *r1_or_interleaved, const char *r2_or_null, pg_reads **out);
/* Threads the interleaved parser may use (0 = all hardware threads, at most 32; PG_INGEST_THREADS overrides the
 * default).  The result does not depend on it. */
void pg_set_ingest_threads(int n);
/* Sharded ingest, one shard per rank/GPU (contiguous ranges of barcode runs): shard `part` of `n_parts`
 * parses only its byte range [size*part/n_parts, size*(part+1)/n_parts) of an UNCOMPRESSED interleaved file, moved
 * forward at both ends to the end of the barcode run in progress there, so that the shards' runs concatenated in
 * rank order are exactly the runs pg_ingest_fastq gives for the whole file (the producer loop of
 * count_tnf.cpp:238-289 is inherently serial; this is its data-parallel form).  Record alignment comes from line
 * numbers, not from guessing at '@': first every rank calls pg_fastq_count_newlines for its own range, the counts
 * are exchanged, and newlines_before[i] (i = 0..n_parts) = number of '\n' in bytes [0, size*i/n_parts).  gzip input
 * fails with PG_EFORMAT (callers fall back to pg_ingest_fastq on every rank). */
int pg_fastq_count_newlines(const char *path, int part, int n_parts, int64_t *n_newlines);
int pg_ingest_fastq_shard(const char *path, int part, int n_parts, const int64_t *newlines_before, pg_reads **out);
/* The same two ingests with the copy to the GPU inside (ingest_dev.hip; replaces count_tnf.cpp:234-283 + the device copy for
 * uncompressed interleaved input): the byte range is cut into pieces (PG_INGEST_PIECE bytes, default 16 MiB) that the parser
 * threads take from a queue; a thread copies the packed characters of a piece it has finished into the device STAGING arrays
 * while the other threads go on parsing, so the PCIe copy hides under the parse, and the last host phase -- shifting every
 * piece to its bit offset in the stream -- becomes one kernel: pg_ingest_place writes codes[n_words] / valid[n_words]
 * (n_words = pg_reads_n_words) from the staging arrays.  No host copy of the stream exists: pg_reads_codes / pg_reads_valid
 * of such a handle are NULL; runs, names, counts and the (host, rare) lower-case plane are as for pg_ingest_fastq.
 * part / n_parts / newlines_before as for pg_ingest_fastq_shard (0 / 1 / NULL: the whole file).  staging_codes
 * (uint64) and staging_valid (uint32) are DEVICE arrays of staging_words >= pg_ingest_staging_words(file size) elements, free
 * again after pg_ingest_place.  *out stays NULL with status PG_OK when the input is not an uncompressed file: the caller then
 * uses pg_ingest_fastq and copies the arrays itself.  The stream is identical to pg_ingest_fastq's for every piece size. */
int64_t pg_ingest_staging_words(int64_t file_bytes);
/* gzip input for the device ingest (feature.py:76-91 reads *.gz through `pigz -dc`: one inflate stream): the file is inflated
 * into an anonymous in-memory file, as pg_ingest_fastq does for its threaded parse; *fd_out is then a descriptor that
 * "/proc/self/fd/<fd>" names for pg_ingest_fastq_device (*bytes_out = the size of the text, for pg_ingest_staging_words) and
 * that the caller closes.  *fd_out = -1 with status PG_OK: not a gzip file, or its text does not fit the memory this process
 * may park it in (PG_INFLATE_MAX_BYTES; half of what is available by default) -- the caller takes pg_ingest_fastq, which
 * then streams it. */
int pg_inflate_to_memfd(const char *path, int *fd_out, int64_t *bytes_out);
int pg_ingest_fastq_device(const char *path, int part, int n_parts, const int64_t *newlines_before, int64_t file_bytes,
                           uint64_t *staging_codes, uint32_t *staging_valid, int64_t staging_words, pg_reads **out);
int pg_ingest_place(const pg_reads *r, uint64_t *staging_codes, const uint32_t *staging_valid, int64_t staging_words,
                    uint64_t *codes, uint32_t *valid, int64_t n_words, void *stream);
/* The same for -1 / -2 input (count_tnf.cpp:174-231, the paired loop: a pair whose names or barcodes differ is skipped, its reads
 * follow the last run; feature.py:76-83: bases of a quality below '?' are not bases to jellyfish -- the low-quality plane):
 * R1 is cut into pieces of PG_INGEST_PIECE bytes, a thread that has paired and packed the records of a piece (and found the
 * same records in R2) copies its local streams -- codes, validity, low quality -- into the three device STAGING arrays while
 * the others go on, and pg_ingest_place_pair shifts the pieces into place: codes, valid and -- if pg_reads_staged_lowq --
 * lowq[n_words] (pg_reads_lowq of such a handle is NULL: the plane exists on the device only).  staging_words >=
 * pg_ingest_pair_staging_words(bytes of R1, bytes of R2).  *out stays NULL with status PG_OK when the two are not
 * uncompressed files (gzip: pg_inflate_to_memfd first): the caller then uses pg_ingest_fastq.  Same stream, runs and counters
 * as pg_ingest_fastq(r1, r2) for every piece size and thread count. */
int64_t pg_ingest_pair_staging_words(int64_t r1_bytes, int64_t r2_bytes);
int pg_ingest_fastq_pair_device(const char *r1, const char *r2, int64_t r1_bytes, int64_t r2_bytes, uint64_t *staging_codes,
                                uint32_t *staging_valid, uint32_t *staging_lowq, int64_t staging_words, pg_reads **out);
int pg_reads_staged_lowq(const pg_reads *r);
int pg_ingest_place_pair(const pg_reads *r, uint64_t *staging_codes, const uint32_t *staging_valid, const uint32_t *staging_lowq,
                         int64_t staging_words, uint64_t *codes, uint32_t *valid, uint32_t *lowq, int64_t n_words, void *stream);
void pg_reads_free(pg_reads *r);
int64_t pg_reads_n_chars(const pg_reads *r);
int64_t pg_reads_n_words(const pg_reads *r); /* padded word count of codes[] and valid[] */
int64_t pg_reads_n_pairs(const pg_reads *r);
int64_t pg_reads_n_unpaired(const pg_reads *r);
int64_t pg_reads_n_runs(const pg_reads *r);
const uint64_t *pg_reads_codes(const pg_reads *r);
const uint32_t *pg_reads_valid(const pg_reads *r);
/* NULL, or the plane of LOWER-case a c g t (same layout as valid) when the input has any: jellyfish counts them as bases
 * (feature.py:94), the reference's own counters reset on them (count_tnf.cpp:91-96) -- so `valid` excludes them, their
 * codes are in `codes` all the same, and a caller that wants jellyfish's table counts with valid | lower (and gives the
 * strict plane in pg_rows.strict_valid). */
const uint32_t *pg_reads_lower(const pg_reads *r);
/* NULL, or the plane of bases (either case) whose quality character is below '?' -- paired (-1/-2) input only: there the
 * reference runs jellyfish with --min-qual-char=? (feature.py:76-83), which counts such bases as N, while count_tnf /
 * count_kmer never read the quality lines.  The multiplicity table is therefore counted with (valid | lower) & ~lowq and
 * the rows with `valid`: a row's k-mer may be missing from the table (count_kmer.cpp:87 skips it), so these inputs take
 * the lookup form of pg_features. */
const uint32_t *pg_reads_lowq(const pg_reads *r);
const int64_t *pg_reads_run_off(const pg_reads *r); /* [n_runs + 1] */
const char *pg_reads_run_name(const pg_reads *r, int64_t i);
/* every run name followed by a NUL into out[0 .. cap); returns the bytes needed (cap = 0, out = NULL: size only) */
int64_t pg_reads_run_names(const pg_reads *r, char *out, int64_t cap);
/* "" (undecided), "10x" or "stLFR": the header mode the file latched (count_tnf.cpp:27-32) */
const char *pg_reads_mode(const pg_reads *r);
/* Surviving rows: runs with a non-empty barcode and more than min_len characters
 * (count_tnf.cpp:81 == count_kmer.cpp:62).  row_run may be NULL to obtain only the count. */
int64_t pg_reads_rows(const pg_reads *r, int min_len, int64_t *row_run);

/* ASCII run text -> stream words (the packing used by pg_ingest_fastq, exposed for callers that
 * already hold reads in memory).  codes/valid must hold pg_words_for(n) words. */
int64_t pg_words_for(int64_t n_chars);
int pg_pack_ascii(const char *text, int64_t n_chars, uint64_t *codes, uint32_t *valid);
/* the same with the lower-case plane (may be NULL); returns 1 if the text has lower-case bases, 0 if not */
int pg_pack_ascii_lower(const char *text, int64_t n_chars, uint64_t *codes, uint32_t *valid, uint32_t *lower);

/* Split rows into work segments of at most seg_chars characters (a multiple of 32).  Call with
 * seg_row == NULL to obtain the number of segments.  Host arrays. */
int64_t pg_plan_segments(const int64_t *row_start, const int64_t *row_end, int64_t n_rows, int64_t seg_chars,
                         int32_t *seg_row, int64_t *seg_start, int64_t *seg_end);

/* ----------------------------------------------------------------------------------------------
 * TNF columns.  Column c of a TNF row counts the canonical k-mer with the c-th smallest code
 * (std::map iteration order, count_tnf.cpp:108-109).  colmap[code] (4^k entries) gives the column of
 * min(code, revcomp(code)).
 * ---------------------------------------------------------------------------------------------- */
#define PG_TNF_MAX_K 6
int pg_tnf_ncols(int k);
int pg_tnf_colmap(int k, uint16_t *colmap /* [4^k] host */, uint32_t *col_code /* [ncols] host, may be NULL */);

/* ----------------------------------------------------------------------------------------------
 * Global canonical k-mer multiplicities over every read (device).
 * Replaces `jellyfish count -C -m k` + `jellyfish dump` (src/feature.py:94,103) and the dump reload
 * of count_kmer.cpp:139-170.  Two table forms:
 *   PG_TABLE_DENSE  k <= 16: uint32_t counts[4^k], indexed by the canonical code.
 *   PG_TABLE_HASH   k <= 21: uint64_t slots[2^log2_slots], open addressing, linear probing;
 *                   slot = (key << 22) | count, 0 = empty, where key = pg_key42(canonical code) is a
 *                   BIJECTION of the 42-bit codes onto themselves that mixes like a hash (below): a k-mer is
 *                   hashed once and its key then serves as bucket id (top bits), slot index (the bits
 *                   under them) and identity.  The count field stops growing at PG_HASH_COUNT_SAT (2^21),
 *                   far above any vector_size*window, so every histogram bin is exact.  A key's home slot
 *                   is its top log2_slots bits.  With log2_bucket_slots = b > 0 the table
 *                   is split into buckets of 2^b consecutive slots and probing wraps inside the
 *                   bucket (b = 0: one bucket = the whole table); pg_kmer_count_bucketed needs
 *                   b <= PG_BUCKET_MAX_LOG2_SLOTS so that one bucket fits in LDS.
 *   PG_TABLE_WIDE   k <= 31 (the reference's whole range, count_kmer.cpp:11-21): uint64_t keys[2^log2_slots]
 *                   holding code + 1 (0 = empty), immediately followed by uint32_t counts[2^log2_slots]
 *                   (12 bytes per slot); same placement and probing, unbucketed, built and read by the direct
 *                   kernels (pg_kmer_count / pg_features) only; counts are plain uint32 (no saturation).
 * Tables must be zero-filled by the caller before the first count; counting accumulates, so a stream
 * may be counted in pieces (and tables of several GPUs can be summed).
 * ---------------------------------------------------------------------------------------------- */
enum { PG_TABLE_DENSE = 1, PG_TABLE_HASH = 2, PG_TABLE_WIDE = 3, PG_TABLE_MINI = 4, PG_TABLE_MINI_WIDE = 5 };
/*   PG_TABLE_MINI   PG_MINI_MIN_K <= k <= 21: uint64_t slots[2^log2_slots], slot = (canonical code << 22) | count, in
 *                   buckets of 2^log2_bucket_slots slots (<= PG_BUCKET_MAX_LOG2_SLOTS, at most 2^16 buckets); a k-mer's
 *                   bucket is a hash of its MINIMIZER (the smallest hashed canonical 13-mer inside it; 11-mer for k < 16), its home slot
 *                   inside the bucket a hash of its code; linear probing wraps inside the bucket.  Consecutive k-mers
 *                   of a read share their minimizer, so the partition passes move super-k-mers (12 bytes for a run of
 *                   k-mers) instead of 8 bytes per occurrence: built by pg_mini_plan + pg_mini_count, read by
 *                   pg_features (lookups) and pg_kmer_merge (entries of a dump).
 *   PG_TABLE_MINI_WIDE  22 <= k <= 31, the same pipeline with the slot layout of PG_TABLE_WIDE: uint64_t keys[2^log2_slots]
 *                   (canonical code + 1, 0 = empty) followed by uint32_t counts[2^log2_slots] (plain 32-bit counts, no
 *                   saturation value); buckets of <= 2^PG_MINI_WIDE_MAX_LOG2_BUCKET_SLOTS slots (12 bytes per slot in
 *                   LDS).  The minimizer is taken over the CENTRAL 8 or 9 13-mers of the k-mer (a window that the
 *                   reverse complement maps onto itself), so the first pass keeps at most 9 values per lane for every k.
 *                   Entries of a dump are merged with pg_kmer_merge_wide. */
#define PG_MINI_M 13          /* minimizer length for k >= 16 */
#define PG_MINI_M_SMALL 11    /* ... and for PG_MINI_MIN_K <= k <= 15 */
#define PG_MINI_MIN_K 13
#define PG_MINI_MAX_LOG2_BUCKETS 16
#define PG_MINI_MAX_ROWS ((1 << 20) - 2)
#define PG_MINI_WIDE_MAX_LOG2_BUCKET_SLOTS 13
#define PG_DENSE_MAX_K 16
#define PG_HASH_MAX_K 21
#define PG_WIDE_MAX_K 31
/* pg_key42: x ^= x >> 21; x = x * M1 mod 2^42; x ^= x >> 21; x = x * M2 mod 2^42; x ^= x >> 21  (every step invertible:
 * the xorshifts are involutions on 42 bits, the multipliers are odd) */
#define PG_KEY42_M1 0x3d7ed558ccdULL
#define PG_KEY42_M2 0x1fe1a85ec53ULL
#define PG_HASH_COUNT_BITS 22
#define PG_HASH_COUNT_SAT (1u << 21)
#define PG_BUCKET_MAX_LOG2_SLOTS 14   /* 2^14 slots x 8 B = 128 KiB of the CU's 160 KiB LDS */
#define PG_BUCKET_MAX_LOG2_BUCKETS 17  /* two scatter passes of <= 9 bits; the histogram takes 2^15 bins per launch */

typedef struct {
    int32_t kind;       /* PG_TABLE_DENSE, _HASH, _WIDE or _MINI */
    int32_t k;
    int32_t log2_slots;        /* hash only */
    int32_t log2_bucket_slots; /* hash only; 0 = unbucketed */
    void *data;         /* device: uint32_t[4^k], uint64_t[2^log2_slots], or the wide form's keys followed by counts */
} pg_table;

/* Count the k-mers ending in words [word_begin, word_end) of the stream into `t`.
 * status (device uint32_t[2], zeroed by the caller): [0] is a set of PG_STATUS_* bits, non-zero after synchronising when the
 * count did not complete (PG_STATUS_TABLE_FULL: the PG_ETABLEFULL condition), [1] unused.  May be NULL for dense tables. */
#define PG_STATUS_TABLE_FULL 1u      /* a bucket (or the table) has no free slot left: enlarge log2_slots and count again */
#define PG_STATUS_OVERFLOW_LIST 2u   /* the exchange's list of count remainders >= 0xffff ran over its capacity */
#define PG_STATUS_PLAN_MISMATCH 4u   /* pg_mini_count: the plan describes more records than the record workspace holds (a plan
                                        of another stream): nothing was counted */
#define PG_STATUS_BOUNDS 8u          /* checked builds only (libpangaea_feat_checked.so): a kernel was about to store outside
                                        the buffer it was given; the store was dropped */
int pg_kmer_count(const uint64_t *codes, const uint32_t *valid, int64_t word_begin, int64_t word_end,
                  const pg_table *t, uint32_t *status, void *stream);

/* HyperLogLog sketch of the distinct canonical k-mers of a word range, for sizing a table before counting:
 * registers = device uint32_t[PG_HLL_REGISTERS], zeroed by the caller (several ranges / GPUs combine by element-wise
 * max).  Estimate = alpha * m^2 / sum_j 2^-registers[j] with the usual small-range correction (pangaea_amd/kmer.py). */
#define PG_HLL_REGISTERS 4096
int pg_kmer_distinct_sketch(const uint64_t *codes, const uint32_t *valid, int64_t word_begin, int64_t word_end, int k,
                            uint32_t *registers, void *stream);

/* Rows = barcode runs that yield an output row: sorted, disjoint character ranges of the stream (device arrays). */
typedef struct {
    const int64_t *row_start; /* device [n_rows] */
    const int64_t *row_end;   /* device [n_rows] */
    int64_t n_rows;           /* < 2^22 - 1 per launch */
    /* NULL, or the STRICT validity plane (upper-case ACGT only) when the `valid` plane given to the counting call also
     * accepts lower-case bases -- jellyfish counts them (feature.py:94), the reference's own row counters reset on them
     * (count_kmer.cpp:73-78): k-mers that are only valid under the lenient rule enter the table but no row.  Device
     * uint32_t[n_words], same layout as `valid`. */
    const uint32_t *strict_valid;
} pg_rows;

/* The same result as pg_kmer_count for a bucketed hash table, without random HBM traffic: the k-mer
 * occurrences of the word range are hash-partitioned into the table's buckets by two streaming scatter
 * passes, each bucket is counted inside LDS by one workgroup and written back as its slice of the table.
 *   accumulate = 0: the table is taken to be empty and every slice is overwritten (no clearing needed);
 *   accumulate = 1: slices are loaded first, so counts add to what the table holds.
 * rows (may be NULL): when given, every partition record also carries the index of the row its k-mer ends in,
 * and the bucket-ordered records stay in the workspace for pg_abundance_from_records.
 * workspace: device scratch of pg_kmer_count_workspace_bytes(word_end - word_begin, t) bytes, 256-B aligned
 * (two record buffers of 8 B per character of the range + counters).  status as for pg_kmer_count:
 * [0] != 0 after synchronising means a bucket overflowed (PG_ETABLEFULL condition). */
int64_t pg_kmer_count_workspace_bytes(int64_t n_words, const pg_table *t);
int pg_kmer_count_bucketed(const uint64_t *codes, const uint32_t *valid, int64_t word_begin, int64_t word_end,
                           const pg_table *t, int accumulate, const pg_rows *rows, void *workspace, int64_t workspace_bytes,
                           uint32_t *status, void *stream);

/* One GPU, a fresh table, abundance parameters known: pg_kmer_count_bucketed with the lookup pass of
 * pg_abundance_from_records fused into the counting kernel -- while a bucket's counts are still in LDS its records are
 * looked up and their (row, bin) words left in `shuffle_workspace` (pg_abundance_workspace_bytes); the abundance rows
 * are then finished by pg_abundance_from_emitted (same arguments as pg_abundance_from_records).  Needs rows and at least
 * 2^11 buckets. */
int pg_kmer_count_bucketed_emit(const uint64_t *codes, const uint32_t *valid, int64_t word_begin, int64_t word_end,
                                const pg_table *t, const pg_rows *rows, void *workspace, int64_t workspace_bytes,
                                int window, int vsize, void *shuffle_workspace, int64_t shuffle_workspace_bytes,
                                uint32_t *status, void *stream);

/* Add n (key,count) pairs -- `pairs[i] = (pg_key42(code) << 22) | count`, the slot format -- into a hash table:
 * the merge step after tables of other GPUs have been gathered. */
int pg_kmer_merge(const uint64_t *pairs, int64_t n, const pg_table *t, uint32_t *status, void *stream);

/* (code, count) pairs into a wide table (separate arrays). */
int pg_kmer_merge_wide(const uint64_t *codes, const uint32_t *counts, int64_t n, const pg_table *t, uint32_t *status, void *stream);

/* The same merge for a bucketed table whose foreign entries arrive in bucket order (a table's slots are laid out
 * bucket by bucket, so the occupied slots of another rank's table, in slot order, already are): one workgroup per
 * bucket merges inside LDS, no global atomics.  `pairs` = n_parts vectors concatenated; seg[p * (n_buckets + 1) + b]
 * .. seg[p * (n_buckets + 1) + b + 1] delimit part p's entries of bucket b (absolute offsets into pairs, device). */
int pg_kmer_merge_bucketed(const uint64_t *pairs, const int64_t *seg, int n_parts, const pg_table *t,
                           uint32_t *status, void *stream);

/* The exchange step of the multi-GPU path in three launches around one all-gather (pangaea_amd/dist.py):
 *   pg_table_bucket_fill      fill[b] = occupied slots of bucket b (device, [n_buckets])
 *   pg_table_compact          occupied slots of bucket b -> out[seg[b] .. seg[b+1]), seg = exclusive scan of fill
 *                             (device, [n_buckets + 1]); order inside a bucket is unspecified
 *   pg_kmer_rebuild_bucketed  as pg_kmer_merge_bucketed, but the table is REBUILT from the parts alone (they include
 *                             this rank's own compacted table): the old slots are neither read nor need be initialised */
int pg_table_bucket_fill(const pg_table *t, int64_t *fill, void *stream);
/* Counting for the exchange without materialising this rank's table: `t` has the geometry of the union over all ranks
 * (the records are partitioned the way the final lookups need them), 2^group_log2 adjacent buckets are counted together
 * in one LDS table (a rank's share of the keys is that much sparser), and only the occupied entries and fill[b] are
 * produced -- t->data is NOT written.  The entries stay in the workspace until pg_deferred_gather copies bucket b's to
 * out[seg[b] ..] (seg = exclusive scan of fill).  Needs more than 256 buckets; the row-tagged records for
 * pg_abundance_from_records are left behind exactly as by pg_kmer_count_bucketed. */
#define PG_DEFERRED_MAX_GROUP_LOG2 3
int pg_kmer_count_deferred(const uint64_t *codes, const uint32_t *valid, int64_t word_begin, int64_t word_end,
                           const pg_table *t, int group_log2, const pg_rows *rows, void *workspace, int64_t workspace_bytes,
                           int64_t *fill, uint32_t *status, void *stream);
int pg_deferred_gather(const pg_table *t, const void *count_workspace, int64_t count_workspace_bytes, int64_t n_words_counted,
                       const int64_t *fill, const int64_t *seg, uint64_t *out, void *stream);
/* The 6-byte exchange format (tables of >= 2^11 buckets): inside its bucket an entry is a 4-byte tag (the key's bits
 * below the bucket id) and a 2-byte count, sent in two planes; a count >= 0xffff sends 0xffff and its remainder as a whole
 * 8-byte entry in `overflow` (merged afterwards with pg_kmer_merge; overflow_count is a device counter, status bit 1 is set
 * when overflow_cap is too small).  tag_elem[b] / cnt_elem[b] = index of bucket b's first tag / count in the uint32 /
 * uint16 view of `out`.  pg_kmer_rebuild_planes_range rebuilds buckets [bucket_begin, bucket_end) from the gathered
 * planes: part p's tags at buf + p * part_stride_bytes, its counts 4 * cap bytes later, seg[p][j] = index of the first
 * entry of bucket bucket_begin + j (bucket_end - bucket_begin + 1 entries per part). */
int pg_deferred_gather_planes(const pg_table *t, const void *count_workspace, int64_t count_workspace_bytes, int64_t n_words_counted,
                              const int64_t *fill, const int64_t *tag_elem, const int64_t *cnt_elem, void *out,
                              uint64_t *overflow, uint64_t *overflow_count, int64_t overflow_cap, uint32_t *status, void *stream);
int pg_kmer_rebuild_planes_range(const void *buf, int64_t part_stride_bytes, int64_t cap, const int64_t *seg, int n_parts,
                                 const pg_table *t, int64_t bucket_begin, int64_t bucket_end, uint32_t *status, void *stream);
/* For the owner-partitioned exchange (>= 4 ranks): a rank that has merged its own bucket range sends that range on.
 * fill[j] = occupied slots of bucket bucket_begin + j; the compaction writes the range in the 6-byte format, tag_elem /
 * cnt_elem indexed by j. */
int pg_table_bucket_fill_range(const pg_table *t, int64_t bucket_begin, int64_t bucket_end, int64_t *fill, void *stream);
int pg_table_compact_planes_range(const pg_table *t, int64_t bucket_begin, int64_t bucket_end, const int64_t *tag_elem,
                                  const int64_t *cnt_elem, void *out, uint64_t *overflow, uint64_t *overflow_count,
                                  int64_t overflow_cap, uint32_t *status, void *stream);
int pg_table_compact(const pg_table *t, const int64_t *seg, uint64_t *out, void *stream);
int pg_kmer_rebuild_bucketed(const uint64_t *pairs, const int64_t *seg, int n_parts, const pg_table *t,
                             uint32_t *status, void *stream);
/* the same for buckets [bucket_begin, bucket_end) only -- seg then has (bucket_end - bucket_begin + 1) entries per part,
 * entry j for bucket bucket_begin + j -- so that the rebuild of one range can run while the all-gather of the next is
 * still in flight */
int pg_kmer_rebuild_bucketed_range(const uint64_t *pairs, const int64_t *seg, int n_parts, const pg_table *t,
                                   int64_t bucket_begin, int64_t bucket_end, uint32_t *status, void *stream);

/* ----------------------------------------------------------------------------------------------
 * Per-run feature rows (device).  One launch fills both matrices.
 *   tnf_out [n_rows, ncols(k_tnf)] int32: canonical k_tnf-mer counts        (count_tnf.cpp:78-113)
 *   abd_out [n_rows, vsize]        int32: hist[count(kmer)/window]++ where the bin is < vsize
 *                                          (count_kmer.cpp:55-108)
 * Either output may be NULL (then its parameters are ignored).  Both must be zero-filled by the
 * caller; segments of one row add into the same row.  seg_* are the device copies of
 * pg_plan_segments' arrays; colmap is the device copy of pg_tnf_colmap's table.
 * ---------------------------------------------------------------------------------------------- */
int pg_features(const uint64_t *codes, const uint32_t *valid, int64_t n_words,
                const int32_t *seg_row, const int64_t *seg_start, const int64_t *seg_end, int64_t n_segs,
                int k_tnf, const uint16_t *colmap, int32_t *tnf_out,
                const pg_table *t, int window, int vsize, int32_t *abd_out, void *stream);

/* The abundance matrix of pg_features WITHOUT random table reads (count_kmer.cpp:55-108 again): the records left in
 * `count_workspace` by pg_kmer_count_bucketed(..., rows, ...) are looked up bucket by bucket in LDS copies of the
 * table's slices (so `t` may meanwhile have been merged with other ranks' tables), turned into (row, bin) words,
 * scattered back by groups of 64 rows and histogrammed in LDS.  abd_out [n_rows, vsize] is overwritten (no zero
 * fill needed).  Requirements: the same `t` geometry, `rows` and word count as the counting call, one counting
 * call since the table was last reset, vsize <= PG_SHUFFLE_MAX_VSIZE.  workspace: pg_abundance_workspace_bytes. */
#define PG_SHUFFLE_MAX_VSIZE 512
int64_t pg_abundance_workspace_bytes(int64_t n_words_counted, int64_t n_rows, int vsize, const pg_table *t);
int pg_abundance_from_records(const pg_table *t, const pg_rows *rows, int window, int vsize, int32_t *abd_out,
                              const void *count_workspace, int64_t count_workspace_bytes, int64_t n_words_counted,
                              void *workspace, int64_t workspace_bytes, void *stream);
int pg_abundance_from_emitted(const pg_table *t, const pg_rows *rows, int window, int vsize, int32_t *abd_out,
                              const void *count_workspace, int64_t count_workspace_bytes, int64_t n_words_counted,
                              void *workspace, int64_t workspace_bytes, void *stream);

/* ----------------------------------------------------------------------------------------------
 * The super-k-mer form of the table build + abundance lookups (one GPU, PG_TABLE_MINI; same results as
 * pg_kmer_count_bucketed_emit + pg_abundance_from_emitted, i.e. jellyfish count -C (feature.py:94) and
 * count_kmer.cpp:55-108).
 *   pg_mini_plan   one pass over the word range: how many super-k-mer records every CHUNK of the stream sends to each
 *                  first-pass region and how many records every bucket receives -> exact write offsets for both scatter
 *                  passes (no global cursor atomics in the first one).  Depends on the stream, `rows` and the table
 *                  geometry only: a caller that counts the same range again (another pass, a benchmark step) keeps the
 *                  plan.  Leaves the number of records in the first 8 bytes of `plan_ws` (device).
 *   pg_mini_count  stream -> records by region -> records by bucket -> every bucket counted inside LDS by one
 *                  workgroup (slots of a FRESH table are overwritten, no clearing) and, when window > 0, looked up again
 *                  while its counts are still in LDS: the (row, bin) words of count_kmer.cpp:86-96 are left in
 *                  `shuffle_ws` for pg_mini_abundance_from_emitted.  `rows` may be NULL and window = vsize = 0 (table only).
 * plan_ws: pg_mini_plan_bytes; rec_ws: pg_mini_records_bytes(n_records, t) (two buffers of 12 B per record);
 * shuffle_ws: pg_mini_shuffle_bytes.  All 256-byte aligned
 * device memory.  A plan that names more records than rec_ws holds sets PG_STATUS_PLAN_MISMATCH and counts nothing.
 * ---------------------------------------------------------------------------------------------- */
int64_t pg_mini_plan_bytes(int64_t n_words, const pg_table *t);
int pg_mini_plan(const uint64_t *codes, const uint32_t *valid, int64_t word_begin, int64_t word_end, const pg_table *t,
                 const pg_rows *rows, void *plan_ws, int64_t plan_ws_bytes, void *stream);
int64_t pg_mini_records_bytes(int64_t n_records, const pg_table *t);
int64_t pg_mini_shuffle_bytes(int64_t n_words, int64_t n_rows, int vsize);
/* the same for a caller that passes merge_ws to pg_mini_count: where the merged lookups apply (the library's own rule: the slot
 * form, fewer than 2^20 rows, row and bin in 28 bits) and the row groups take one scatter pass, the row shuffle's buffer of
 * provisional words is not part of the layout -- half the bytes (12 GB less per 10 M read pairs) */
int64_t pg_mini_shuffle_bytes_merged(int64_t n_words, int64_t n_rows, int vsize, const pg_table *t);
/* merge_ws (may be NULL: the word-wise lookups): pg_mini_merge_words() 4-byte words of device memory for the MERGED form of the
 * lookups (the default; PG_MINI_MERGE=0 in the environment or a NULL buffer: word-wise) -- the k-mers of a record that share row
 * and bin travel as one word with a count; the provisional data are the k-mers' 2-byte slot numbers in fixed places per record,
 * sized from the plan's record counts (the plan workspace's first 8-byte word: records; third: records of more than 4 k-mers). */
int64_t pg_mini_merge_words(int64_t n_words, int64_t n_records, int64_t n_long_records, const pg_table *t);

/* ---- a stream counted in PIECES (one GPU, packed mini tables, the merged lookups): the scratch of pg_mini_count grows with the
 * stream (about 3.4 KB per 150 bp read pair); a stream too large for one piece is counted word range by word range.  The reference
 * streams one barcode group at a time in O(group) memory (count_tnf.cpp:257-271) and needs the finished jellyfish table before the
 * first lookup (count_kmer.cpp:139-170); here every piece is planned, partitioned and counted INTO the table's buckets (the bucket's
 * slice is loaded into LDS, counted on, written back: slots keep their places), leaving only its 2-byte provisional slots and the
 * meta words of its records behind; when the table is final the pieces' slots are looked up in it:
 *   pg_mini_plan(codes, valid, w0, w1, ...)      the piece's plan (its own plan_ws, kept until the lookups)
 *   pg_mini_count_piece(... first ...)          plan_ws, rec_ws, merge_ws of THIS piece; first != 0 for the first piece of a fresh table
 *   (keep: plan_ws, merge_ws and the second meta plane of rec_ws -- records x 4 bytes -- of every piece)
 *   pg_mini_lookup_begin(t, rows, n_words_total, vsize, shuffle_ws, ...)   once; shuffle_ws: pg_mini_shuffle_bytes_merged(n_words_total, ...)
 *   pg_mini_lookup_piece(...)                   per piece, any order
 *   pg_mini_abundance_from_emitted(t, rows, vsize, abd, <a plan_ws of pg_mini_plan_bytes(n_words_total)>, ..., n_words_total, shuffle_ws, ...)
 * rows: the rows of the WHOLE stream (their positions are absolute). */
int pg_mini_count_piece(const uint64_t *codes, const uint32_t *valid, int64_t word_begin, int64_t word_end, const pg_table *t,
                        const pg_rows *rows, void *plan_ws, int64_t plan_ws_bytes, void *rec_ws, int64_t rec_ws_bytes,
                        int window, int vsize, void *merge_ws, int64_t merge_ws_words, int first, uint32_t *status, void *stream);
int pg_mini_lookup_begin(const pg_table *t, const pg_rows *rows, int64_t n_words_total, int vsize, void *shuffle_ws, int64_t shuffle_ws_bytes, void *stream);
int pg_mini_lookup_piece(const pg_table *t, const pg_rows *rows, const void *plan_ws, int64_t plan_ws_bytes, int64_t n_words_piece,
                         const uint32_t *meta, int64_t n_words_total, int window, int vsize, void *shuffle_ws, int64_t shuffle_ws_bytes,
                         const void *merge_ws, uint32_t *status, void *stream);
int pg_mini_count(const uint64_t *codes, const uint32_t *valid, int64_t word_begin, int64_t word_end, const pg_table *t,
                  const pg_rows *rows, void *plan_ws, int64_t plan_ws_bytes, void *rec_ws, int64_t rec_ws_bytes,
                  int window, int vsize, void *shuffle_ws, int64_t shuffle_ws_bytes, void *merge_ws, int64_t merge_ws_words,
                  uint32_t *status, void *stream);
/* `stream` waits for the first scatter pass of the calling thread's latest pg_mini_count (an event recorded there): the next
 * batch's pg_mini_plan, enqueued on that stream afterwards, runs beside the memory-bound second pass. */
int pg_mini_wait_first_pass(void *stream);
int pg_mini_abundance_from_emitted(const pg_table *t, const pg_rows *rows, int vsize, int32_t *abd_out,
                                   const void *plan_ws, int64_t plan_ws_bytes, int64_t n_words_counted,
                                   void *shuffle_ws, int64_t shuffle_ws_bytes, void *stream);

/* ---- the super-k-mer form on N > 1 ranks (the reference runs ONE jellyfish over all reads: feature.py:94; here every rank
 * counts its own reads and the counts meet at bucket owners).  A rank's counts are not final until the other ranks' occurrences
 * of the same k-mers are in, so pg_mini_count is cut in two:
 *   pg_mini_count_half    as pg_mini_count up to the bucket workgroups, which leave -- instead of the slice and the lookups --
 *                         their occupied slots (canonical code << 22 | count) as compacted ENTRIES in slot order, the number of
 *                         them in fill[bucket] (device int64), the occupancy bitmaps and the provisional (row, slot) words.
 *                         `local` has the UNION's bucket count and, per bucket, slots for this rank's own k-mers; its `data` is
 *                         never written.
 *   pg_mini_gather_entries  bucket b's entries -> out[dst_elem[b] ..): the send buffer (out_elems entries), bucket ranges in owner order
 *   (all-to-all: 8 bytes per entry)
 *   pg_mini_merge_bins    owner side, buckets [bucket_begin, bucket_end) of the union table `t`: part p's entries of owned bucket i
 *                         lie at recv[p * part_stride + seg[p * (n_owned + 1) + i] .. seg[p * (n_owned + 1) + i + 1]); the parts are
 *                         summed inside LDS (counts saturate at PG_HASH_COUNT_SAT, as one rank's would), the merged slices are
 *                         written to `t`, and bins_out (same layout, uint16) receives for EVERY entry the bin of its k-mer in the
 *                         merged table: count / window + 1, or 0xffff beyond the vector (count_kmer.cpp:86-96)
 *   (all-to-all back: 2 bytes per entry)
 *   pg_mini_lookup_half   bins_in[bin_elem[b] ..) = the bins of bucket b's entries, in the order they were sent -> the lookups of
 *                         the provisional words and the row-group scatter, exactly as pg_mini_count ends;
 *                         pg_mini_abundance_from_emitted(local, ...) then writes the rows.
 * half_ws: pg_mini_half_bytes(local), 256-byte aligned device memory; merge_ws as for pg_mini_count (the same buffer in both calls);
 * rec_ws: the count half's record workspace, untouched in between (the merged lookups read the records' lengths and rows again). */
int64_t pg_mini_half_bytes(const pg_table *local);
int pg_mini_count_half(const uint64_t *codes, const uint32_t *valid, int64_t word_begin, int64_t word_end, const pg_table *local,
                       const pg_rows *rows, void *plan_ws, int64_t plan_ws_bytes, void *rec_ws, int64_t rec_ws_bytes,
                       int window, int vsize, void *shuffle_ws, int64_t shuffle_ws_bytes, void *merge_ws, int64_t merge_ws_words,
                       void *half_ws, int64_t half_ws_bytes, int64_t *fill, uint32_t *status, void *stream);
int pg_mini_gather_entries(const pg_table *local, const void *half_ws, int64_t half_ws_bytes, const int64_t *fill,
                           const int64_t *dst_elem, uint64_t *out, int64_t out_elems, uint32_t *status, void *stream);
int pg_mini_merge_bins(const uint64_t *recv, int64_t part_stride, const int64_t *seg, int n_parts, const pg_table *t,
                       int64_t bucket_begin, int64_t bucket_end, int window, int vsize, uint16_t *bins_out, uint32_t *status, void *stream);
int pg_mini_lookup_half(const pg_table *local, const pg_rows *rows, const void *plan_ws, int64_t plan_ws_bytes, const void *rec_ws, int64_t rec_ws_bytes,
                        int64_t n_words_counted,
                        int vsize, void *shuffle_ws, int64_t shuffle_ws_bytes, const void *merge_ws, int64_t merge_ws_words,
                        const void *half_ws, int64_t half_ws_bytes,
                        const uint16_t *bins_in, const int64_t *bin_elem, uint32_t *status, void *stream);

/* ----------------------------------------------------------------------------------------------
 * Row normalisation of a count matrix (a9: Data.__init__, src/data.py:16-21 -- sklearn normalize(norm="l1") in float64, the
 * sampling weight = (row maximum of the normalised abundance)^2, matrices narrowed to float32).  One pass over device int32
 * [n_rows, n_cols]: out[r, c] = (float)((double)m[r, c] / (double)sum_c |m[r, c]|), an all-zero row stays zero (sklearn puts 1
 * for a zero norm); weight[r] (may be NULL) = ((double)max_c m[r, c] / sum)^2 -- division by a positive number is monotone, so
 * this is the square of the row maximum of the quotients.  Integer sums and IEEE double division: bit-identical to the CPU.
 * ---------------------------------------------------------------------------------------------- */
int pg_normalize_rows(const int32_t *m, int64_t n_rows, int n_cols, float *out, double *weight, void *stream);

/* ----------------------------------------------------------------------------------------------
 * Cache files.  Rows as the reference binaries print them: "<name>,v1,...,vD\n", numbers through
 * ostream<<double (%g, six significant digits; count_tnf.cpp:293-303), gzip container.
 * names: n_rows NUL-terminated strings back to back.  mat: host int32 [n_rows, n_cols].
 * ---------------------------------------------------------------------------------------------- */
int pg_write_csv_gz(const char *path, const char *names, const int32_t *mat, int64_t n_rows, int64_t n_cols);

/* ----------------------------------------------------------------------------------------------
 * Bin writer (host): clusters.tsv -> <out_prefix>_bin<label>.fq and .barcode, the on-disk bin layout the
 * unchanged reassembly stage reads.  Replaces the reference's extract_reads tool (extract_reads.cpp:57-190):
 * same tsv grammar ("-1" labels skipped), same kept pairs, same rewritten headers
 * ("<name>\tBX:Z:<barcode>-1").  r2 == NULL: interleaved input.  pairs_written may be NULL.
 * ---------------------------------------------------------------------------------------------- */
int pg_extract_reads(const char *r1_or_interleaved, const char *r2_or_null, const char *clusters_tsv,
                     const char *out_prefix, int64_t *pairs_written);

#ifdef __cplusplus
}
#endif
#endif /* PANGAEA_FEAT_H */
