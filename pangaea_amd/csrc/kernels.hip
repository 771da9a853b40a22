// kernels.hip -- gfx950 kernels of the barcode k-mer feature path + their C-ABI launchers.
//
// Replaces (reference file:line, /root/reference/src):
//   K2  pg_kmer_count*            jellyfish count -C (feature.py:94) + dump reload (cpptools/count_kmer.cpp:139-170)
//   K1  pg_features / tnf         cpptools/count_tnf.cpp:78-113  (per-run canonical k_tnf-mer counts)
//   K3  pg_features / abd         cpptools/count_kmer.cpp:55-108 (per-run histogram of global multiplicities), by table lookups
//       pg_abundance_from_records the same rows without random HBM reads, from the partitioned occurrence records
//
// Work decomposition (wave64):
//   * the read stream is 2-bit codes + 1-bit validity, 32 characters per word (include/pangaea_feat.h);
//     one lane owns one word per step, so a wave reads 512 B of codes + 256 B of validity, coalesced;
//   * a lane rolls the forward and reverse-complement codes over its 32 characters after pre-rolling
//     the k-1 characters before its word (taken from the previous word), so no cross-lane traffic;
//   * which positions end a valid k-mer comes from one bit-parallel pass over the 64-bit validity
//     window (runs of >= k ones), not from a per-character run counter;
//   * integer counting only: LDS histograms / LDS hash tables, global atomics only where unavoidable.
//     No MFMA -- there is no contraction on this path.
//
// Hash table building, two ways:
//   direct    (kmer_count_kernel)      one random 64-B line + one memory-side atomic per k-mer occurrence;
//   bucketed  (pg_kmer_count_bucketed) occurrences are hash-partitioned by two streaming scatter passes (LDS-staged so
//             HBM sees whole runs), every bucket is then counted inside LDS by one workgroup and its LDS image is
//             written back as that bucket's slice of the table.  HBM only sees streams; counting atomics are LDS atomics.
// Abundance rows, two ways:
//   lookups   (features_kernel)        one random 64-B line per k-mer occurrence;
//   shuffle   (pg_abundance_from_records) the partition records carry their row id; every bucket's slice is loaded into
//             LDS, each record's count becomes a (row, bin) word, the words are scattered back by row group (two passes)
//             and histogrammed in LDS.  Again only streams.
#include "pg_device.hpp"

namespace {
enum { TK_NONE = 0, TK_DENSE = 1, TK_HASH = 2, TK_WIDE = 3, TK_MINI = 4, TK_MINIW = 5 };

// Packed hash tables (k <= 21) do not store the canonical code but key42(code): a BIJECTION of the 42-bit codes onto
// themselves with the avalanche of a hash (two xorshift-multiply rounds modulo 2^42; every step is invertible, the
// inverse is in pangaea_amd/kmer.py).  A k-mer is hashed ONCE, where it leaves the read stream; from then on every
// consumer takes its digit, bucket and slot straight from the bits of the key -- bucket = top bits, slot = the bits
// below -- and the key alone identifies the k-mer, in partition records, in table slots and between ranks.
constexpr int KEY_BITS = 42;
constexpr uint64_t KEY_MASK = (1ull << KEY_BITS) - 1;
static_assert(2 * PG_HASH_MAX_K == KEY_BITS, "the packed table keys cover exactly the codes of the largest k");
__device__ __forceinline__ uint64_t key42(uint64_t x)
{
    x ^= x >> 21; x = (x * PG_KEY42_M1) & KEY_MASK;
    x ^= x >> 21; x = (x * PG_KEY42_M2) & KEY_MASK;
    x ^= x >> 21;
    return x;
}

// hash table as the kernels see it: 2^log2_slots slots in buckets of 2^log2_bucket slots; a key's home slot is its top
// log2_slots bits (packed form) / the top bits of mix64(code) (wide form), probing is linear and wraps inside the
// bucket.  Slots are never freed, so a lookup may stop at the first empty slot.
struct HashView {
    uint64_t *slots;
    int log2_slots;
    int log2_bucket;
    __device__ __forceinline__ uint64_t home_wide(uint64_t code) const { return mix64(code) >> (64 - log2_slots); }
    __device__ __forceinline__ uint64_t home_key(uint64_t key) const { return key >> (KEY_BITS - log2_slots); }
    // PG_TABLE_MINI: bucket from the k-mer's minimizer, slot inside the bucket from a hash of the code (the slots hold codes)
    __device__ __forceinline__ uint64_t home_mini(uint64_t code, int k) const
    {
        const uint32_t b = mini_bucket(mini_minimizer_of(code, k), log2_slots - log2_bucket);
        return ((uint64_t)b << log2_bucket) | (mini_slot_hash<true>(code) & ((1u << log2_bucket) - 1u));
    }
    __device__ __forceinline__ uint64_t next(uint64_t s) const
    {
        const uint64_t bm = (1ull << log2_bucket) - 1;
        return (s & ~bm) | ((s + 1) & bm);
    }
    __device__ __forceinline__ uint32_t limit() const { return log2_bucket < 14 ? (1u << log2_bucket) : MAX_PROBE; }
};

template <typename KT> __device__ __forceinline__ KT low_mask(int k)
{
    return (2 * k >= (int)(8 * sizeof(KT))) ? (KT)~(KT)0 : (KT)(((KT)1 << (2 * k)) - 1);
}

// rolling state of one lane: forward / reverse-complement code of the last k characters
template <typename KT> struct Roller {
    KT fw, rc, kmask;
    int rc_shift;
    __device__ __forceinline__ void init(int k)
    {
        fw = rc = 0;
        kmask = low_mask<KT>(k);
        rc_shift = 2 * (k - 1);
    }
    __device__ __forceinline__ void push(uint32_t c)
    {
        fw = (KT)(fw << 2) | (KT)c;
        rc = (KT)(rc >> 2) | (KT)((KT)(c ^ 2u) << rc_shift);
    }
    __device__ __forceinline__ KT canon() const
    {
        const KT f = fw & kmask;
        return f < rc ? f : rc;
    }
};

// -------------------------------------------------------------------------------- table access

__device__ __forceinline__ void dense_add(uint32_t *table, uint32_t code) { atomicAdd(&table[code], 1u); }

// slot = (key << 22) | count, key = key42(code) ; 0 = empty.  Keys never change once written, so a stale (cached)
// read can only show "empty", and the compare-and-swap then returns the real occupant.
__device__ __forceinline__ void hash_add_from(const HashView &t, uint64_t s, uint64_t cur, uint64_t code, uint32_t *status)
{
    const uint32_t limit = t.limit();
    for (uint32_t i = 0; i < limit; ++i) {
        if (cur == 0) {
            cur = atomicCAS((unsigned long long *)&t.slots[s], 0ull, (unsigned long long)((code << HASH_CBITS) | 1ull));
            if (cur == 0) return;
        }
        if ((cur >> HASH_CBITS) == code) {
            // stop growing at SAT; overshoot is bounded by the threads in flight (< 2^20 < 2^22 - SAT)
            if ((uint32_t)(cur & HASH_CMASK) < HASH_SAT) atomicAdd((unsigned long long *)&t.slots[s], 1ull);
            return;
        }
        s = t.next(s);
        cur = t.slots[s];
    }
    atomicOr(status, 1u);
}

__device__ __forceinline__ uint32_t hash_probe(const HashView &t, uint64_t s, uint64_t cur, uint64_t code, bool *found)
{
    const uint32_t limit = t.limit();
    for (uint32_t i = 0; i < limit; ++i) {
        if (cur == 0) break;
        if ((cur >> HASH_CBITS) == code) { *found = true; return (uint32_t)(cur & HASH_CMASK); }
        s = t.next(s);
        cur = t.slots[s];
    }
    *found = false;
    return 0;
}

// Wide form (k <= 31): keys[2^s] hold code + 1 (0 = empty), counts[2^s] (uint32) follow the key array.  Same placement
// and probing as the packed form; counts wrap modulo 2^32 like any uint32 counter.
__device__ __forceinline__ uint32_t *wide_counts(const HashView &t) { return reinterpret_cast<uint32_t *>(t.slots + (1ull << t.log2_slots)); }

// (mini_k != 0: a PG_TABLE_MINI_WIDE table -- home slot from the minimizer bucket, same keys/counts planes)
__device__ __forceinline__ void wide_add(const HashView &t, uint64_t code, uint32_t add, uint32_t *status, int mini_k = 0)
{
    const uint32_t limit = t.limit();
    const uint64_t key1 = code + 1;
    uint64_t s = mini_k ? t.home_mini(code, mini_k) : t.home_wide(code);
    for (uint32_t i = 0; i < limit; ++i) {
        uint64_t cur = t.slots[s];
        if (cur == 0) {
            cur = atomicCAS((unsigned long long *)&t.slots[s], 0ull, (unsigned long long)key1);
            if (cur == 0) cur = key1;
        }
        if (cur == key1) { atomicAdd(&wide_counts(t)[s], add); return; }
        s = t.next(s);
    }
    atomicOr(status, 1u);
}

__device__ __forceinline__ uint32_t wide_probe(const HashView &t, uint64_t s, uint64_t cur, uint64_t code, bool *found)
{
    const uint32_t limit = t.limit();
    const uint64_t key1 = code + 1;
    for (uint32_t i = 0; i < limit; ++i) {
        if (cur == 0) break;
        if (cur == key1) { *found = true; return wide_counts(t)[s]; }
        s = t.next(s);
        cur = t.slots[s];
    }
    *found = false;
    return 0;
}

__global__ __launch_bounds__(BLOCK) void wide_merge_kernel(const uint64_t *__restrict__ codes, const uint32_t *__restrict__ counts, int64_t n,
                                                           HashView t, uint32_t *status, int mini_k)
{
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK)
        if (counts[i]) wide_add(t, codes[i], counts[i], status, mini_k);
}

// -------------------------------------------------------------------------------- K2 direct: global counts

template <typename KT, int TK>
__global__ __launch_bounds__(BLOCK) void kmer_count_kernel(const uint64_t *__restrict__ codes, const uint32_t *__restrict__ valid,
                                                           int64_t word_begin, int64_t word_end, int k, uint32_t *dense,
                                                           HashView t, uint32_t *status)
{
    for (int64_t w = word_begin + (int64_t)blockIdx.x * BLOCK + threadIdx.x; w < word_end; w += (int64_t)gridDim.x * BLOCK) {
        const Word x = load_word(codes, valid, w, k);
        if (x.ok == 0) continue;
        Roller<KT> r;
        r.init(k);
        for (int i = 33 - k; i < 32; ++i) r.push((uint32_t)(x.pw >> (2 * i)) & 3u);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            KT canon[8];
            uint64_t cur[8];
            uint64_t hh[8];
            uint64_t kk[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = b * 8 + u;
                r.push((uint32_t)(x.cw >> (2 * j)) & 3u);
                canon[u] = r.canon();
                if ((x.ok >> j) & 1) {
                    if (TK == TK_DENSE) {
                        dense_add(dense, (uint32_t)canon[u]);
                    } else if (TK == TK_WIDE) {
                        wide_add(t, (uint64_t)canon[u], 1u, status);
                    } else {                       // issue the first probe of the whole batch before resolving any
                        kk[u] = key42((uint64_t)canon[u]);
                        hh[u] = t.home_key(kk[u]);
                        cur[u] = t.slots[hh[u]];
                    }
                }
            }
            if (TK == TK_HASH) {
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if ((x.ok >> (b * 8 + u)) & 1) hash_add_from(t, hh[u], cur[u], kk[u], status);
            }
        }
    }
}

// merge (key,count) pairs of another table (slot format); counts saturate at SAT exactly (CAS loop; not a hot path)
__global__ __launch_bounds__(BLOCK) void kmer_merge_kernel(const uint64_t *__restrict__ pairs, int64_t n, HashView t, uint32_t *status, int mini_k)
{
    const uint32_t limit = t.limit();
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
        const uint64_t p = pairs[i];
        if (p == 0) continue;
        const uint64_t code = p >> HASH_CBITS;
        uint32_t add = (uint32_t)(p & HASH_CMASK);
        if (add > HASH_SAT) add = HASH_SAT;
        uint64_t s = mini_k ? t.home_mini(code, mini_k) : t.home_key(code);
        bool done = false;
        for (uint32_t tries = 0; tries < limit && !done; ++tries) {
            uint64_t cur = __hip_atomic_load(&t.slots[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (;;) {
                if (cur != 0 && (cur >> HASH_CBITS) != code) break;          // occupied by another key
                uint32_t have = (uint32_t)(cur & HASH_CMASK);
                uint32_t sum = have + add > HASH_SAT ? HASH_SAT : have + add;
                uint64_t want = (code << HASH_CBITS) | sum;
                uint64_t old = atomicCAS((unsigned long long *)&t.slots[s], (unsigned long long)cur, (unsigned long long)want);
                if (old == cur) { done = true; break; }
                cur = old;
            }
            s = t.next(s);
        }
        if (!done) atomicOr(status, 1u);
    }
}

// HyperLogLog sketch of the canonical k-mers of a word range: registers[j] = max over k-mers with hash prefix j of
// (1 + leading zeros of the remaining hash bits).  4096 registers: ~1.6 % standard error -- enough to size a table.
constexpr int HLL_BITS = 12;
__global__ __launch_bounds__(BIG_BLOCK) void distinct_sketch_kernel(const uint64_t *__restrict__ codes, const uint32_t *__restrict__ valid,
                                                                    int64_t word_begin, int64_t word_end, int k, uint32_t *__restrict__ registers)
{
    __shared__ uint32_t reg[1 << HLL_BITS];
    for (int i = threadIdx.x; i < (1 << HLL_BITS); i += BIG_BLOCK) reg[i] = 0;
    __syncthreads();
    for (int64_t w = word_begin + (int64_t)blockIdx.x * BIG_BLOCK + threadIdx.x; w < word_end; w += (int64_t)gridDim.x * BIG_BLOCK) {
        const Word x = load_word(codes, valid, w, k);
        if (x.ok == 0) continue;
        Roller<uint64_t> r;
        r.init(k);
        for (int i = 33 - k; i < 32; ++i) r.push((uint32_t)(x.pw >> (2 * i)) & 3u);
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            r.push((uint32_t)(x.cw >> (2 * j)) & 3u);
            if ((x.ok >> j) & 1) {
                // an independent hash (the table's own hash decides bucket membership; reusing it would not bias the
                // sketch, but keeping them apart costs nothing)
                const uint64_t h = mix64(r.canon() ^ 0x9E3779B97F4A7C15ull);
                const uint32_t idx = (uint32_t)(h >> (64 - HLL_BITS));
                const uint64_t rest = (h << HLL_BITS) | (1ull << (HLL_BITS - 1));      // sentinel bit bounds the rank
                atomicMax(&reg[idx], (uint32_t)__clzll((long long)rest) + 1u);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < (1 << HLL_BITS); i += BIG_BLOCK)
        if (reg[i]) atomicMax(&registers[i], reg[i]);
}

// -------------------------------------------------------------------------------- partition machinery
//
// occurrence record = canonical code (low 42 bits) | row id << 42 (22 bits, ROW_NONE = not inside any row)

constexpr int RPL64 = 32, RPL32 = 32; // record scatter passes: 8-byte / 4-byte records a lane keeps in registers
constexpr int MAX_FAN_BITS = 9;       // <= 512-way scatter per pass
constexpr int WIDE_FAN_BITS = 10;     // the one-pass row shuffle: 1024 row groups (65536 rows), 64 words per lane
constexpr int RPL32_WIDE = 64;
constexpr int REC_KEY_BITS = 42;
constexpr uint64_t REC_KEY_MASK = (1ull << REC_KEY_BITS) - 1;
constexpr uint32_t ROW_NONE = (1u << (64 - REC_KEY_BITS)) - 1;
constexpr int GROUP_ROWS_LOG2 = 6;    // rows per LDS row-histogram group (64 x 512 bins x 4 B = 128 KiB at most)

constexpr int TILE_WORDS = 256;       // words per stream tile of the first scatter pass (= A1_TILE_WORDS)
// first row whose end lies beyond the first character of each stream tile (rows sorted, disjoint)
__global__ __launch_bounds__(BLOCK) void tile_rows_kernel(const int64_t *__restrict__ row_end, int64_t n_rows, int64_t word_begin,
                                                          int64_t n_tiles, int32_t *__restrict__ tile_row)
{
    const int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= n_tiles) return;
    const int64_t c0 = (word_begin + t * TILE_WORDS) * 32;
    int64_t lo = 0, hi = n_rows;                                    // first r with row_end[r] > c0
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (row_end[mid] > c0) hi = mid; else lo = mid + 1;
    }
    tile_row[t] = (int32_t)lo;
}

// A0: histogram of final bucket ids (top `bits` bits of the hash) over every valid k-mer of the word range; one launch
// covers the bins [bin_base, bin_base + n_bins) (n_bins <= 2^15: 128 KiB of LDS counters).  With chunk_hist given it
// also records, for every CHUNK_WORDS-word chunk of the range, how many k-mers fall into each first-pass digit
// (top bits1 bits): chunk_hist[d * n_chunks + chunk].  A per-digit scan of that table gives every chunk its exact write
// offsets, so the first scatter pass needs no global cursor atomics.
constexpr int CHUNK_WORDS = 4096;     // = 32 stream tiles of the first scatter pass
__global__ __launch_bounds__(BIG_BLOCK) void bucket_hist_kernel(const uint64_t *__restrict__ codes, const uint32_t *__restrict__ valid,
                                                                int64_t word_begin, int64_t word_end, int k, int bits,
                                                                uint32_t bin_base, int n_bins, unsigned long long *__restrict__ hist,
                                                                int bits1, unsigned long long *__restrict__ chunk_hist, int64_t n_chunks,
                                                                int64_t chunk_stride)
{
    // chunks are placed in the order slot = chunk * chunk_stride mod n_chunks (stride coprime to n_chunks): regions then
    // hold the stream in scrambled chunk order, so later passes do not see row-sorted input
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t *coarse = lds + n_bins;                                // [256]
    for (int i = threadIdx.x; i < n_bins; i += BIG_BLOCK) lds[i] = 0;
    const int sh = KEY_BITS - bits, sh1 = KEY_BITS - bits1;
    for (int64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        if (threadIdx.x < 256) coarse[threadIdx.x] = 0;
        __syncthreads();
        const int64_t w0 = word_begin + chunk * CHUNK_WORDS;
        const int64_t w1 = w0 + CHUNK_WORDS < word_end ? w0 + CHUNK_WORDS : word_end;
        for (int64_t w = w0 + threadIdx.x; w < w1; w += BIG_BLOCK) {
            const Word x = load_word(codes, valid, w, k);
            if (x.ok == 0) continue;
            Roller<uint64_t> r;
            r.init(k);
            for (int i = 33 - k; i < 32; ++i) r.push((uint32_t)(x.pw >> (2 * i)) & 3u);
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                r.push((uint32_t)(x.cw >> (2 * j)) & 3u);
                if ((x.ok >> j) & 1) {
                    const uint64_t h = key42(r.canon());
                    const uint32_t bin = (uint32_t)(h >> sh) - bin_base;
                    if (bin < (uint32_t)n_bins) atomicAdd(&lds[bin], 1u);
                    if (chunk_hist) atomicAdd(&coarse[h >> sh1], 1u);
                }
            }
        }
        __syncthreads();
        if (chunk_hist && threadIdx.x < (1u << bits1))
            chunk_hist[(int64_t)threadIdx.x * n_chunks + (int64_t)(((__int128)chunk * chunk_stride) % n_chunks)] = coarse[threadIdx.x];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_bins; i += BIG_BLOCK)
        if (lds[i]) atomicAdd(&hist[bin_base + i], (unsigned long long)lds[i]);
}

// LDS bookkeeping shared by the scatter passes (<= 2^FAN_BITS digits)
template <int FAN_BITS> struct ScatterLds {
    uint32_t start[(1 << FAN_BITS) + 1];
    unsigned long long gbase[1 << FAN_BITS];                        // its first half doubles as the digit counters until the scan
    uint32_t wave_tot[WAVES];
    __device__ __forceinline__ uint32_t *cnt() { return reinterpret_cast<uint32_t *>(gbase); }
};

// exclusive scan of L.cnt[0..n_dig) into L.start[0..n_dig] (PER = 2^FAN_BITS / BLOCK consecutive entries per lane); all lanes call
template <int FAN_BITS> __device__ __forceinline__ void scatter_scan(ScatterLds<FAN_BITS> &L, int n_dig)
{
    constexpr int PER = (1 << FAN_BITS) / BLOCK > 0 ? (1 << FAN_BITS) / BLOCK : 1;
    uint32_t e[PER];
    uint32_t v = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int i = PER * (int)threadIdx.x + j;
        e[j] = i < n_dig ? L.cnt()[i] : 0;
        v += e[j];
    }
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d);
        if ((int)(threadIdx.x & 63) >= d) incl += o;
    }
    if ((threadIdx.x & 63) == 63) L.wave_tot[threadIdx.x >> 6] = incl;
    lds_sync();
    uint32_t before = 0;
    for (int wv = 0; wv < (int)(threadIdx.x >> 6); ++wv) before += L.wave_tot[wv];
    uint32_t run = before + incl - v;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int i = PER * (int)threadIdx.x + j;
        if (i < n_dig) L.start[i] = run;
        run += e[j];
    }
    if (threadIdx.x == BLOCK - 1) L.start[n_dig] = before + incl;
    lds_sync();
}

// A1: stream -> 2^bits1 regions (bits1 <= 8).  A lane owns HALF a word (16 characters) and keeps its <= 16 records in
// registers; one returning LDS atomic per record yields its rank inside its digit, the tile (128 words, <= 4096
// records) is then laid out digit-sorted in LDS and every digit's run is appended to its region with one global cursor
// add, so HBM receives contiguous runs.  36 KiB of LDS per workgroup: four workgroups (16 waves) per CU.
// With rows given, every record also carries the index of the row its k-mer ends in (ROW_NONE outside all rows).
constexpr int A1_CHARS = 32;            // characters per lane: 16 (two lanes share a word) or 32
constexpr int A1_LANES_PER_WORD = 32 / A1_CHARS;
constexpr int A1_TILE_WORDS = BLOCK * A1_CHARS / 32;
static_assert(A1_TILE_WORDS == TILE_WORDS, "tile_rows_kernel and scatter_stream_kernel must agree on the tile size");
constexpr int A1_TILE = BLOCK * A1_CHARS;
struct StreamLds {
    uint32_t cnt[256];
    uint32_t start[257];
    unsigned long long gbase[256];
    unsigned long long cur[256];                                    // running write offsets of this chunk, per digit
    uint32_t wave_tot[WAVES];
};
static_assert(CHUNK_WORDS % A1_TILE_WORDS == 0, "a chunk is a whole number of stream tiles");
__global__ __launch_bounds__(BLOCK) void scatter_stream_kernel(const uint64_t *__restrict__ codes, const uint32_t *__restrict__ valid,
                                                               int64_t word_begin, int64_t word_end, int k, int bits1,
                                                               const int64_t *__restrict__ row_start, const int64_t *__restrict__ row_end,
                                                               int64_t n_rows, const uint32_t *__restrict__ strict,
                                                               const int32_t *__restrict__ tile_row,
                                                               uint64_t *__restrict__ rec_out,
                                                               const unsigned long long *__restrict__ chunk_off, int64_t n_chunks,
                                                               int64_t chunk_stride)
{
    // one workgroup per chunk: chunk_off[d * n_chunks + chunk] is where this chunk's records of digit d start
    __shared__ uint64_t buf[A1_TILE];
    __shared__ StreamLds L;
    const int n_dig = 1 << bits1;
    const int dsh = KEY_BITS - bits1;
    const int64_t n_tiles = (word_end - word_begin + A1_TILE_WORDS - 1) / A1_TILE_WORDS;
    const int half = threadIdx.x % A1_LANES_PER_WORD;               // which A1_CHARS characters of the word
    const int64_t chunk = blockIdx.x;
    constexpr int TILES_PER_CHUNK = CHUNK_WORDS / A1_TILE_WORDS;
    const int64_t slot = (int64_t)(((__int128)chunk * chunk_stride) % n_chunks);
    if ((int)threadIdx.x < n_dig) L.cur[threadIdx.x] = chunk_off[(int64_t)threadIdx.x * n_chunks + slot];
    for (int64_t tile = chunk * TILES_PER_CHUNK; tile < (chunk + 1) * TILES_PER_CHUNK && tile < n_tiles; ++tile) {
        L.cnt[threadIdx.x] = 0;
        lds_sync();
        uint64_t rec[A1_CHARS];
        uint32_t dr[A1_CHARS];                                      // digit << 16 | rank inside the digit
        uint32_t ok = 0;
        const int64_t w = word_begin + tile * A1_TILE_WORDS + (threadIdx.x / A1_LANES_PER_WORD);
        if (w < word_end) {
            const Word x = load_word(codes, valid, w, k);
            ok = (uint32_t)(((uint64_t)x.ok >> (A1_CHARS * half)) & ((1ull << A1_CHARS) - 1ull));
            // `valid` may count lower-case bases as bases (jellyfish's rule for the table); the rows follow the reference's own
            // counters, which do not: with a second, strict validity plane a k-mer that is only valid under the lenient rule is
            // counted but belongs to no row
            uint32_t ok_row = ok;
            if (strict && ok) {
                const uint32_t sv = strict[w], sp = w > 0 ? strict[w - 1] : 0u;
                ok_row &= (uint32_t)(((runs_of(((uint64_t)sv << 32) | sp, k) >> 32) >> (A1_CHARS * half)) & ((1ull << A1_CHARS) - 1ull));
            }
            if (ok) {
                // row bookkeeping: r = first row that can still contain a position >= the current one
                int64_t r = n_rows, rs = INT64_MAX, re = INT64_MAX;
                const int64_t pos0 = (w << 5) + A1_CHARS * half;
                if (row_start) {
                    r = tile_row[tile];
                    while (r < n_rows && row_end[r] <= pos0) ++r;
                    if (r < n_rows) { rs = row_start[r]; re = row_end[r]; }
                }
                Roller<uint64_t> rl;
                rl.init(k);
                // pre-roll the k-1 characters before the lane's first one: word positions c in [first-(k-1), first)
                for (int c = A1_CHARS * half - (k - 1); c < A1_CHARS * half; ++c)
                    rl.push((uint32_t)((c < 0 ? x.pw >> (2 * (c + 32)) : x.cw >> (2 * c)) & 3u));
                const uint64_t mine = x.cw >> (2 * A1_CHARS * half);
#pragma unroll
                for (int j = 0; j < A1_CHARS; ++j) {
                    rl.push((uint32_t)(mine >> (2 * j)) & 3u);
                    if ((ok >> j) & 1) {
                        const int64_t pos = pos0 + j;
                        while (pos >= re) {                         // rows are at least one character long: terminates
                            ++r;
                            rs = r < n_rows ? row_start[r] : INT64_MAX;
                            re = r < n_rows ? row_end[r] : INT64_MAX;
                        }
                        const uint64_t row = (pos >= rs && ((ok_row >> j) & 1)) ? (uint64_t)r : (uint64_t)ROW_NONE;
                        const uint64_t key = key42(rl.canon());
                        const uint32_t d = (uint32_t)(key >> dsh);
                        rec[j] = key | (row << REC_KEY_BITS);
                        dr[j] = (d << 16) | atomicAdd(&L.cnt[d], 1u);
                    }
                }
            }
        }
        lds_sync();
        {   // exclusive scan of cnt[256] -> start[257]: one entry per lane
            const uint32_t v = L.cnt[threadIdx.x];
            uint32_t incl = v;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = __shfl_up(incl, d);
                if ((int)(threadIdx.x & 63) >= d) incl += o;
            }
            if ((threadIdx.x & 63) == 63) L.wave_tot[threadIdx.x >> 6] = incl;
            lds_sync();
            uint32_t before = 0;
            for (int wv = 0; wv < (int)(threadIdx.x >> 6); ++wv) before += L.wave_tot[wv];
            L.start[threadIdx.x] = before + incl - v;
            if (threadIdx.x == BLOCK - 1) L.start[BLOCK] = before + incl;
            // gbase[d] = (where the digit's run goes) - (where it starts in the tile): the copy-out adds the tile position
            if ((int)threadIdx.x < n_dig) { L.gbase[threadIdx.x] = L.cur[threadIdx.x] - (before + incl - v); L.cur[threadIdx.x] += v; }
            lds_sync();
        }
#pragma unroll
        for (int j = 0; j < A1_CHARS; ++j)
            if ((ok >> j) & 1) buf[L.start[dr[j] >> 16] + (dr[j] & 0xffffu)] = rec[j];
        lds_sync();
        // copy out: LDS position i holds a record of digit d at rank i - start[d]; the digit is recomputed from the record
        // (pure ALU), so the loop is a flat, pipelinable sweep and consecutive lanes write consecutive addresses per run
        const uint32_t total = L.start[BLOCK];
        for (uint32_t i = threadIdx.x; i < total; i += BLOCK) {
            const uint64_t r = buf[i];
            const uint32_t d = (uint32_t)((r & REC_KEY_MASK) >> dsh);
            rec_out[L.gbase[d] + i] = r;
        }
        lds_sync();
    }
}

// Generic record scatter pass (A2, S2a, S2b).  Region = blockIdx.x / tiles_x; its input records are
//   [in_begin[region << in_shift], in_cnt ? begin + in_cnt[region] : in_end[region << in_shift]).
// Digit: DIG_HASH -> (key >> dshift) & mask of a (key, row) record;  DIG_ROW -> (word >> dshift) & mask of a 32-bit (row, bin) word.
// Destination of digit d: out[obase[(base + d) << oshift] + cursor[base + d] ...], base = flat ? 0 : region << dbits.
enum { DIG_HASH = 0, DIG_ROW = 1 };
template <typename REC, int DIG, int REC_PER_LANE, int FAN_BITS>
__global__ __launch_bounds__(BLOCK) void scatter_records_kernel(const REC *__restrict__ rec_in,
                                                                const unsigned long long *__restrict__ in_begin,
                                                                const unsigned long long *__restrict__ in_end, int in_shift,
                                                                const unsigned long long *__restrict__ in_cnt, int tiles_x,
                                                                REC *__restrict__ rec_out, const unsigned long long *__restrict__ obase,
                                                                unsigned long long *__restrict__ cursor, int oshift, int flat,
                                                                int dbits, int dshift)
{
    constexpr int TILE1 = BLOCK * REC_PER_LANE;
    __shared__ REC buf[TILE1];
    __shared__ ScatterLds<FAN_BITS> L;
    const int n_dig = 1 << dbits;
    const uint32_t dmask = (uint32_t)n_dig - 1u;
    // (all workgroups of a region on ONE of the eight XCDs, which take workgroups round-robin by index: the runs its tiles append
    // to a destination are completed in that XCD's L2 -- see mini_scatter2_kernel.  Flat destinations are shared by all regions.)
    const unsigned n_regions = gridDim.x / (unsigned)tiles_x;
    const bool by_xcd = !flat && n_regions % 8u == 0u;
    const int64_t region = by_xcd ? (int64_t)(blockIdx.x % 8u) + 8 * (int64_t)((blockIdx.x / 8u) / (unsigned)tiles_x) : (int64_t)(blockIdx.x / tiles_x);
    const int64_t base_index = flat ? 0 : (region << dbits);
    const int64_t r0 = (int64_t)in_begin[region << in_shift];
    const int64_t r1 = in_cnt ? r0 + (int64_t)in_cnt[region] : (int64_t)in_end[region << in_shift];
    const int64_t n_tiles = (r1 - r0 + TILE1 - 1) / TILE1;
    auto digit_of = [&](REC r) -> uint32_t {
        if (DIG == DIG_HASH) return (uint32_t)(((uint64_t)r & REC_KEY_MASK) >> dshift) & dmask;
        return ((uint32_t)r >> dshift) & dmask;
    };
    for (int64_t tile = by_xcd ? (blockIdx.x / 8u) % (unsigned)tiles_x : blockIdx.x % tiles_x; tile < n_tiles; tile += tiles_x) {
        for (int i = threadIdx.x; i < n_dig; i += BLOCK) L.cnt()[i] = 0;
        lds_sync();
        const int64_t t0 = r0 + tile * TILE1;
        REC rec[REC_PER_LANE];
        uint32_t dr[REC_PER_LANE];
#pragma unroll
        for (int j = 0; j < REC_PER_LANE; ++j) {                    // all loads of the lane in flight together
            const int64_t i = t0 + j * BLOCK + threadIdx.x;
            rec[j] = i < r1 ? rec_in[i] : (REC)0;
        }
#pragma unroll
        for (int j = 0; j < REC_PER_LANE; ++j) {
            const int64_t i = t0 + j * BLOCK + threadIdx.x;
            // (an all-ones word is a hole: a lookup pass that keeps its words in place marks dropped ones that way)
            if (i < r1 && !(DIG == DIG_ROW && (uint32_t)rec[j] == 0xffffffffu)) {
                const uint32_t d = digit_of(rec[j]);
                dr[j] = (d << 16) | atomicAdd(&L.cnt()[d], 1u);
            }
        }
        lds_sync();
        scatter_scan(L, n_dig);
        // the returning cursor adds (one per digit and tile) are issued first and consumed after the placement: their round
        // trip to L2 overlaps the LDS stores instead of standing in front of them
        constexpr int DPL = ((1 << FAN_BITS) + BLOCK - 1) / BLOCK;  // digits per lane
        unsigned long long gpos[DPL];
#pragma unroll
        for (int q = 0; q < DPL; ++q) {                             // cnt[] is gone (it shared gbase's storage): start[] has it
            const int d = (int)threadIdx.x + q * BLOCK;
            gpos[q] = 0;
            if (d < n_dig) {
                const uint32_t c = L.start[d + 1] - L.start[d];
                if (c) gpos[q] = obase[(base_index + d) << oshift] + atomicAdd(&cursor[base_index + d], (unsigned long long)c) - L.start[d];
            }
        }
#pragma unroll
        for (int j = 0; j < REC_PER_LANE; ++j) {
            const int64_t i = t0 + j * BLOCK + threadIdx.x;
            if (i < r1 && !(DIG == DIG_ROW && (uint32_t)rec[j] == 0xffffffffu)) buf[L.start[dr[j] >> 16] + (dr[j] & 0xffffu)] = rec[j];
        }
#pragma unroll
        for (int q = 0; q < DPL; ++q) {
            const int d = (int)threadIdx.x + q * BLOCK;
            if (d < n_dig) L.gbase[d] = gpos[q];
        }
        lds_sync();
        const uint32_t total = L.start[n_dig];
        for (uint32_t i = threadIdx.x; i < total; i += BLOCK) {    // flat sweep: digit recomputed from the record
            const REC r = buf[i];
            const uint32_t d = digit_of(r);
            rec_out[L.gbase[d] + i] = r;                            // gbase[d] = destination of the run - its start in the tile
        }
        lds_sync();
    }
}

// insert one record into the LDS table, starting from an already fetched first slot; false = table full
__device__ __forceinline__ bool lds_insert(unsigned long long *tab, uint32_t smask, uint32_t limit,
                                           uint64_t code, bool live, uint32_t s, unsigned long long cur)
{
    if (!live) return true;                                         // padding lane of the last batch
    for (uint32_t i = 0; i < limit; ++i) {
        if (cur == 0) {
            cur = atomicCAS(&tab[s], 0ull, (unsigned long long)((code << HASH_CBITS) | 1ull));
            if (cur == 0) return true;
        }
        if ((cur >> HASH_CBITS) == code) {
            if ((uint32_t)(cur & HASH_CMASK) < HASH_SAT) atomicAdd(&tab[s], 1ull);
            return true;
        }
        s = (s + 1) & smask;
        cur = tab[s];
    }
    return false;
}

// B: one workgroup per final bucket.  The bucket's slice of the table (2^log2_bucket slots) lives in LDS while the
// bucket's records stream through (8 loads per lane in flight, first LDS probes of a batch issued back to back);
// the LDS image is then written back as the slice.
constexpr int CNT_BATCH = 8;               // the resolve macros below are written for exactly 8

// ---- one batch of a bucket's records.  The records [r0, r1) of a bucket are read as aligned 16-byte pairs (pair q = records
// 2q and 2q + 1 of the buffer), CNT_BATCH / 2 pairs per lane and batch, all loads in flight together.
struct PairRange {
    const ulonglong2 *rec2;
    int64_t q0, q1, r0, r1;
    __device__ __forceinline__ PairRange(const uint64_t *rec, int64_t r0_, int64_t r1_)
        : rec2(reinterpret_cast<const ulonglong2 *>(rec)), q0(r0_ >> 1), q1((r1_ + 1) >> 1), r0(r0_), r1(r1_) {}
    static constexpr int64_t stride = (int64_t)BIG_BLOCK * (CNT_BATCH / 2);
    // pairs of the batch at `base`; lanes beyond the range get `pad` in both halves
    __device__ __forceinline__ void load(int64_t base, ulonglong2 (&v)[CNT_BATCH / 2], unsigned long long pad) const
    {
#pragma unroll
        for (int j = 0; j < CNT_BATCH / 2; ++j) {
            const int64_t q = base + (int64_t)j * BIG_BLOCK + threadIdx.x;
            v[j] = q < q1 ? rec2[q] : make_ulonglong2(pad, pad);
        }
    }
    // the NEXT batch's loads, issued behind the point where `dep` (something computed from the current batch) is known: without
    // the dependency the compiler hoists these conditional loads in front of the s_waitcnt vmcnt(0) for the current batch, which
    // then waits for them too -- the prefetch is gone, and a kernel that runs one workgroup per CU has nothing else to do meanwhile
    __device__ __forceinline__ void load_next(int64_t base, ulonglong2 (&v)[CNT_BATCH / 2], unsigned long long pad, uint32_t dep) const
    {
        int64_t b = base + stride;
        asm volatile("" : "+v"(b) : "v"(dep));
        load(b, v, pad);
    }
    // which of the batch's records belong to the bucket (the first / last pair may straddle its ends)
    __device__ __forceinline__ void live(int64_t base, bool (&l)[CNT_BATCH]) const
    {
#pragma unroll
        for (int j = 0; j < CNT_BATCH / 2; ++j) {
            const int64_t q = base + (int64_t)j * BIG_BLOCK + threadIdx.x;
            l[2 * j] = q < q1 && 2 * q >= r0;
            l[2 * j + 1] = q < q1 && 2 * q + 1 < r1;
        }
    }
};

__global__ __launch_bounds__(BIG_BLOCK) void bucket_count_kernel(const uint64_t *__restrict__ rec, const unsigned long long *__restrict__ off,
                                                                 HashView t, int accumulate, uint32_t *status)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long tab[];
    const uint32_t n_slots = 1u << t.log2_bucket;
    const uint32_t smask = n_slots - 1;
    const uint32_t limit = t.limit();
    const int hsh = KEY_BITS - t.log2_slots;
    uint64_t *slice = t.slots + ((uint64_t)blockIdx.x << t.log2_bucket);
    const int64_t r0 = (int64_t)off[blockIdx.x], r1 = (int64_t)off[blockIdx.x + 1];
    if (r0 == r1 && accumulate) return;                         // nothing to add, slice stays as it is
    for (uint32_t i = threadIdx.x; i < n_slots; i += BIG_BLOCK) tab[i] = accumulate ? slice[i] : 0ull;
    __syncthreads();
    bool full = false;
    const PairRange pr(rec, r0, r1);
    ulonglong2 nxt[CNT_BATCH / 2];
    pr.load(pr.q0, nxt, 0ull);
    for (int64_t base = pr.q0; base < pr.q1; base += PairRange::stride) {
        uint64_t rr[CNT_BATCH];
        bool live[CNT_BATCH];
        pr.live(base, live);
#pragma unroll
        for (int j = 0; j < CNT_BATCH / 2; ++j) {
            rr[2 * j] = live[2 * j] ? nxt[j].x & REC_KEY_MASK : 0ull;
            rr[2 * j + 1] = live[2 * j + 1] ? nxt[j].y & REC_KEY_MASK : 0ull;
        }
        pr.load(base + PairRange::stride, nxt, 0ull);               // software pipeline: the next batch's loads fly during the inserts
        uint32_t ss[CNT_BATCH];
        unsigned long long first[CNT_BATCH];
#pragma unroll
        for (int j = 0; j < CNT_BATCH; ++j) {
            ss[j] = (uint32_t)(rr[j] >> hsh) & smask;
            first[j] = live[j] ? tab[ss[j]] : 0ull;
        }
        // resolved one by one, written out by hand: an unrolled loop around the probe loop is not unrolled by hipcc and
        // would push the batch arrays into scratch
#define PG_RESOLVE(J) full |= !lds_insert(tab, smask, limit, rr[J], live[J], ss[J], first[J]);
        PG_RESOLVE(0) PG_RESOLVE(1) PG_RESOLVE(2) PG_RESOLVE(3) PG_RESOLVE(4) PG_RESOLVE(5) PG_RESOLVE(6) PG_RESOLVE(7)
#undef PG_RESOLVE
    }
    if (full) atomicOr(status, 1u);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_slots; i += BIG_BLOCK) slice[i] = tab[i];
}

// split-table view of a batch: tag (the key's bits below the bucket / group id, | 0x80000000), home slot, row id
__device__ __forceinline__ void split32(const ulonglong2 (&v)[CNT_BATCH / 2], uint32_t tag_mask, int hsh, uint32_t smask,
                                        uint32_t (&tg)[CNT_BATCH], uint32_t (&ss)[CNT_BATCH], uint32_t (&row)[CNT_BATCH])
{
#pragma unroll
    for (int j = 0; j < CNT_BATCH / 2; ++j) {
        const uint64_t k0 = v[j].x & REC_KEY_MASK, k1 = v[j].y & REC_KEY_MASK;
        tg[2 * j] = ((uint32_t)k0 & tag_mask) | 0x80000000u;
        tg[2 * j + 1] = ((uint32_t)k1 & tag_mask) | 0x80000000u;
        ss[2 * j] = (uint32_t)(k0 >> hsh) & smask;
        ss[2 * j + 1] = (uint32_t)(k1 >> hsh) & smask;
        row[2 * j] = (uint32_t)(v[j].x >> REC_KEY_BITS);
        row[2 * j + 1] = (uint32_t)(v[j].y >> REC_KEY_BITS);
    }
}

// the two 4-byte planes of a slice <-> its packed 8-byte slots (counts are clamped to the saturation value on the way out)
__device__ __forceinline__ void planes_from_slice(uint32_t *tags, uint32_t *cnts, uint32_t n_slots, uint32_t tag_mask, const uint64_t *slice)
{
    for (uint32_t i = threadIdx.x; i < n_slots; i += BIG_BLOCK) {
        const uint64_t v = slice ? slice[i] : 0ull;
        tags[i] = v ? ((uint32_t)(v >> HASH_CBITS) & tag_mask) | 0x80000000u : 0u;
        cnts[i] = (uint32_t)(v & HASH_CMASK);
    }
}
__device__ __forceinline__ uint64_t packed_slot(uint64_t high, uint32_t tag, uint32_t tag_mask, uint32_t count)
{
    return ((high | (tag & tag_mask)) << HASH_CBITS) | (count < HASH_SAT ? count : HASH_SAT);
}
__device__ __forceinline__ void slice_from_planes(const uint32_t *tags, const uint32_t *cnts, uint32_t n_slots, uint32_t tag_mask, uint64_t high,
                                                  uint64_t *slice)
{
    for (uint32_t i = threadIdx.x; i < n_slots; i += BIG_BLOCK) slice[i] = tags[i] ? packed_slot(high, tags[i], tag_mask, cnts[i]) : 0ull;
}

// B with a split LDS table (buckets >= 2^11, i.e. the key's bits below the bucket id fit 31 bits): tags[] = those bits
// | 0x80000000, counts[] = plain uint32 -- the same 128 KiB, but every LDS operation is 4 bytes wide and the increment is a
// 32-bit add with no saturation test (a bucket has < 2^32 records; counts are clamped to SAT when the slots are packed).
__device__ __forceinline__ bool lds_insert32(uint32_t *tags, uint32_t *cnts, uint32_t smask, uint32_t limit, uint32_t tag, bool live,
                                             uint32_t s, uint32_t cur)
{
    if (!live) return true;
    for (uint32_t i = 0; i < limit; ++i) {
        if (cur == 0) {
            cur = atomicCAS(&tags[s], 0u, tag);
            if (cur == 0) cur = tag;
        }
        if (cur == tag) { atomicAdd(&cnts[s], 1u); return true; }
        s = (s + 1) & smask;
        cur = tags[s];
    }
    return false;
}

__global__ __launch_bounds__(BIG_BLOCK) void bucket_count32_kernel(const uint64_t *__restrict__ rec, const unsigned long long *__restrict__ off,
                                                                   HashView t, int accumulate, uint32_t *status)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long tab[];
    const uint32_t n_slots = 1u << t.log2_bucket;
    uint32_t *tags = reinterpret_cast<uint32_t *>(tab), *cnts = tags + n_slots;
    const uint32_t smask = n_slots - 1;
    const uint32_t limit = t.limit();
    const int hsh = KEY_BITS - t.log2_slots;
    const int tag_bits = KEY_BITS - (t.log2_slots - t.log2_bucket);  // <= 31
    const uint32_t tag_mask = (1u << tag_bits) - 1u;
    uint64_t *slice = t.slots + ((uint64_t)blockIdx.x << t.log2_bucket);
    const int64_t r0 = (int64_t)off[blockIdx.x], r1 = (int64_t)off[blockIdx.x + 1];
    if (r0 == r1 && accumulate) return;
    planes_from_slice(tags, cnts, n_slots, tag_mask, accumulate ? slice : nullptr);
    __syncthreads();
    bool full = false;
    const PairRange pr(rec, r0, r1);
    ulonglong2 nxt[CNT_BATCH / 2];
    pr.load(pr.q0, nxt, 0ull);
    for (int64_t base = pr.q0; base < pr.q1; base += PairRange::stride) {
        uint32_t tg[CNT_BATCH], ss[CNT_BATCH], row[CNT_BATCH], first[CNT_BATCH];
        bool live[CNT_BATCH];
        pr.live(base, live);
        split32(nxt, tag_mask, hsh, smask, tg, ss, row);
        pr.load_next(base, nxt, 0ull, tg[0]);                       // software pipeline: the next batch's loads fly during the inserts
#pragma unroll
        for (int j = 0; j < CNT_BATCH; ++j) first[j] = live[j] ? tags[ss[j]] : 0u;
#define PG_RESOLVE(J) full |= !lds_insert32(tags, cnts, smask, limit, tg[J], live[J], ss[J], first[J]);
        PG_RESOLVE(0) PG_RESOLVE(1) PG_RESOLVE(2) PG_RESOLVE(3) PG_RESOLVE(4) PG_RESOLVE(5) PG_RESOLVE(6) PG_RESOLVE(7)
#undef PG_RESOLVE
    }
    if (full) atomicOr(status, 1u);
    __syncthreads();
    slice_from_planes(tags, cnts, n_slots, tag_mask, (uint64_t)blockIdx.x << tag_bits, slice);     // the bucket id is the key's top bits
}

// B for the multi-GPU path: the rank's table is never materialised.  The table descriptor has the geometry of the UNION
// over all ranks (so that the records are partitioned the way the final lookups need them), but one rank's share of the
// keys is 2^g times sparser: a workgroup therefore counts 2^g adjacent final buckets together in one LDS table of the
// usual size (hash bits after the group's), and instead of a 2^log2_bucket-slot image per final bucket it writes only
// the occupied entries, split by final bucket, to scratch[off[b] ..] (a bucket has no more distinct keys than records)
// and their number to fill[b] -- exactly what the exchange sends.  The LDS rebuild after the all-gather writes the table.
__global__ __launch_bounds__(BIG_BLOCK) void bucket_count_compact_kernel(const uint64_t *__restrict__ rec, const unsigned long long *__restrict__ off,
                                                                         HashView t, int g, uint64_t *__restrict__ scratch,
                                                                         long long *__restrict__ fill, uint32_t *status)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long tab[];
    __shared__ uint32_t qcnt[1 << PG_DEFERRED_MAX_GROUP_LOG2];
    const uint32_t n_slots = 1u << t.log2_bucket;
    const uint32_t smask = n_slots - 1;
    const uint32_t limit = t.limit();
    const int hsh = KEY_BITS - (t.log2_slots - g);                  // slot inside the group's table
    const int bsh = KEY_BITS - (t.log2_slots - t.log2_bucket);      // final bucket id
    const uint32_t qmask = (1u << g) - 1u;
    const int64_t b0 = (int64_t)blockIdx.x << g;
    const int64_t r0 = (int64_t)off[b0], r1 = (int64_t)off[b0 + (1 << g)];
    if (threadIdx.x <= qmask) qcnt[threadIdx.x] = 0;
    if (r0 == r1) { if (threadIdx.x <= qmask) fill[b0 + threadIdx.x] = 0; return; }
    for (uint32_t i = threadIdx.x; i < n_slots; i += BIG_BLOCK) tab[i] = 0ull;
    __syncthreads();
    bool full = false;
    const PairRange pr(rec, r0, r1);
    for (int64_t base = pr.q0; base < pr.q1; base += PairRange::stride) {
        ulonglong2 v[CNT_BATCH / 2];
        uint64_t rr[CNT_BATCH];
        bool live[CNT_BATCH];
        pr.load(base, v, 0ull);
        pr.live(base, live);
#pragma unroll
        for (int j = 0; j < CNT_BATCH / 2; ++j) {
            rr[2 * j] = live[2 * j] ? v[j].x & REC_KEY_MASK : 0ull;
            rr[2 * j + 1] = live[2 * j + 1] ? v[j].y & REC_KEY_MASK : 0ull;
        }
        uint32_t ss[CNT_BATCH];
        unsigned long long first[CNT_BATCH];
#pragma unroll
        for (int j = 0; j < CNT_BATCH; ++j) {
            ss[j] = (uint32_t)(rr[j] >> hsh) & smask;
            first[j] = live[j] ? tab[ss[j]] : 0ull;
        }
#define PG_RESOLVE(J) full |= !lds_insert(tab, smask, limit, rr[J], live[J], ss[J], first[J]);
        PG_RESOLVE(0) PG_RESOLVE(1) PG_RESOLVE(2) PG_RESOLVE(3) PG_RESOLVE(4) PG_RESOLVE(5) PG_RESOLVE(6) PG_RESOLVE(7)
#undef PG_RESOLVE
    }
    if (full) atomicOr(status, 1u);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_slots; i += BIG_BLOCK) {
        const unsigned long long v = tab[i];
        if (v) {
            const uint32_t q = (uint32_t)((v >> HASH_CBITS) >> bsh) & qmask;
            scratch[off[b0 + q] + atomicAdd(&qcnt[q], 1u)] = v;
        }
    }
    __syncthreads();
    if (threadIdx.x <= qmask) fill[b0 + threadIdx.x] = (long long)qcnt[threadIdx.x];
}

// the same with the split 32-bit LDS table (the key's bits below the GROUP id fit 31 bits)
__global__ __launch_bounds__(BIG_BLOCK) void bucket_count_compact32_kernel(const uint64_t *__restrict__ rec, const unsigned long long *__restrict__ off,
                                                                           HashView t, int g, uint64_t *__restrict__ scratch,
                                                                           long long *__restrict__ fill, uint32_t *status)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long tab[];
    __shared__ uint32_t qcnt[1 << PG_DEFERRED_MAX_GROUP_LOG2];
    const uint32_t n_slots = 1u << t.log2_bucket;
    uint32_t *tags = reinterpret_cast<uint32_t *>(tab), *cnts = tags + n_slots;
    const uint32_t smask = n_slots - 1;
    const uint32_t limit = t.limit();
    const int hsh = KEY_BITS - (t.log2_slots - g);                  // slot inside the group's table
    const int tag_bits = KEY_BITS - (t.log2_slots - t.log2_bucket - g);   // key bits below the group id, <= 31
    const uint32_t tag_mask = (1u << tag_bits) - 1u;
    const uint32_t qmask = (1u << g) - 1u;
    const int64_t b0 = (int64_t)blockIdx.x << g;
    const int64_t r0 = (int64_t)off[b0], r1 = (int64_t)off[b0 + (1 << g)];
    if (threadIdx.x <= qmask) qcnt[threadIdx.x] = 0;
    if (r0 == r1) { if (threadIdx.x <= qmask) fill[b0 + threadIdx.x] = 0; return; }
    planes_from_slice(tags, cnts, n_slots, tag_mask, nullptr);
    __syncthreads();
    bool full = false;
    const PairRange pr(rec, r0, r1);
    ulonglong2 v[CNT_BATCH / 2];
    pr.load(pr.q0, v, 0ull);
    for (int64_t base = pr.q0; base < pr.q1; base += PairRange::stride) {
        uint32_t tg[CNT_BATCH], ss[CNT_BATCH], row[CNT_BATCH], first[CNT_BATCH];
        bool live[CNT_BATCH];
        pr.live(base, live);
        split32(v, tag_mask, hsh, smask, tg, ss, row);
        pr.load_next(base, v, 0ull, tg[0]);                         // the next batch's loads fly during the inserts
#pragma unroll
        for (int j = 0; j < CNT_BATCH; ++j) first[j] = live[j] ? tags[ss[j]] : 0u;
#define PG_RESOLVE(J) full |= !lds_insert32(tags, cnts, smask, limit, tg[J], live[J], ss[J], first[J]);
        PG_RESOLVE(0) PG_RESOLVE(1) PG_RESOLVE(2) PG_RESOLVE(3) PG_RESOLVE(4) PG_RESOLVE(5) PG_RESOLVE(6) PG_RESOLVE(7)
#undef PG_RESOLVE
    }
    if (full) atomicOr(status, 1u);
    __syncthreads();
    const uint64_t high = (uint64_t)blockIdx.x << tag_bits;         // the group id is the key's top bits
    for (uint32_t i = threadIdx.x; i < n_slots; i += BIG_BLOCK) {
        const uint32_t tg = tags[i];
        if (tg) {
            const uint32_t q = ((tg & tag_mask) >> (tag_bits - g)) & qmask;          // the final bucket inside the group
            scratch[off[b0 + q] + atomicAdd(&qcnt[q], 1u)] = packed_slot(high, tg, tag_mask, cnts[i]);
        }
    }
    __syncthreads();
    if (threadIdx.x <= qmask) fill[b0 + threadIdx.x] = (long long)qcnt[threadIdx.x];
}

// the deferred entries of bucket b, scratch[off[b] .. off[b] + fill[b]), to out[seg[b] ..]: one wavefront per bucket
__global__ __launch_bounds__(BLOCK) void deferred_gather_kernel(const uint64_t *__restrict__ scratch, const unsigned long long *__restrict__ off,
                                                                const long long *__restrict__ fill, const long long *__restrict__ seg,
                                                                int64_t n_buckets, uint64_t *__restrict__ out)
{
    const int64_t b = (int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
    if (b >= n_buckets) return;
    const uint64_t *src = scratch + off[b];
    uint64_t *dst = out + seg[b];
    const long long n = fill[b];
    for (long long i = threadIdx.x & 63; i < n; i += 64) dst[i] = src[i];
}

constexpr int MERGE_BATCH = 4;             // entries per lane in flight in the merge kernels

// The exchange format: 6 bytes per entry instead of 8.  Inside bucket b a key is its bits below the bucket id (the tag, <= 31
// bits for >= 2^11 buckets) and a count is almost always < 2^16, so an entry travels as a 4-byte tag and a 2-byte count in
// two planes; a count >= 0xffff sends 0xffff there and the remainder as a full 8-byte entry in a (tiny) overflow list that
// is merged afterwards -- sums are exact.  tag_elem[b] / cnt_elem[b]: where bucket b's first tag / count goes, in elements
// of the uint32 / uint16 view of the send buffer.
__global__ __launch_bounds__(BLOCK) void deferred_gather_planes_kernel(const uint64_t *__restrict__ scratch, const unsigned long long *__restrict__ off,
                                                                       const long long *__restrict__ fill, const long long *__restrict__ tag_elem,
                                                                       const long long *__restrict__ cnt_elem, int64_t n_buckets, uint32_t tag_mask,
                                                                       uint32_t *__restrict__ out32, uint16_t *__restrict__ out16,
                                                                       uint64_t *__restrict__ ovf, unsigned long long *__restrict__ ovf_count,
                                                                       unsigned long long ovf_cap, uint32_t *status)
{
    const int64_t b = (int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
    if (b >= n_buckets) return;
    const uint64_t *src = scratch + off[b];
    uint32_t *t32 = out32 + tag_elem[b];
    uint16_t *c16 = out16 + cnt_elem[b];
    const long long n = fill[b];
    for (long long i = threadIdx.x & 63; i < n; i += 64) {
        const uint64_t e = src[i];
        const uint32_t c = (uint32_t)(e & HASH_CMASK);
        t32[i] = (uint32_t)(e >> HASH_CBITS) & tag_mask;
        c16[i] = (uint16_t)(c < 0xffffu ? c : 0xffffu);
        if (c >= 0xffffu && c > 0xffffu) {                          // the remainder travels as a whole entry
            const unsigned long long at = atomicAdd(ovf_count, 1ull);
            if (at < ovf_cap) ovf[at] = (e & ~HASH_CMASK) | (uint64_t)(c - 0xffffu);
            else atomicOr(status, 2u);
        }
    }
}

// rebuild of buckets [bucket_base, bucket_base + gridDim.x) from the gathered planes of n_parts ranks: part p's tags are at
// buf + p * part_stride, its counts `cap` tags later; seg[p][j] = index of the first entry of bucket bucket_base + j
__global__ __launch_bounds__(BIG_BLOCK) void bucket_merge_planes_kernel(const uint8_t *__restrict__ buf, int64_t part_stride, int64_t cap,
                                                                        const long long *__restrict__ seg, int n_parts, HashView t,
                                                                        uint32_t *status, int64_t bucket_base, int64_t n_seg)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long tab[];
    const uint32_t n_slots = 1u << t.log2_bucket;
    uint32_t *tags = reinterpret_cast<uint32_t *>(tab), *cnts = tags + n_slots;
    const uint32_t smask = n_slots - 1;
    const uint32_t limit = t.limit();
    const int hsh = KEY_BITS - t.log2_slots;
    const int tag_bits = KEY_BITS - (t.log2_slots - t.log2_bucket);
    const uint32_t tag_mask = (1u << tag_bits) - 1u;
    const int64_t bucket = bucket_base + blockIdx.x;
    uint64_t *slice = t.slots + ((uint64_t)bucket << t.log2_bucket);
    planes_from_slice(tags, cnts, n_slots, tag_mask, nullptr);
    __syncthreads();
    bool full = false;
    for (int p = 0; p < n_parts; ++p) {
        const int64_t a = seg[p * (n_seg + 1) + blockIdx.x], b = seg[p * (n_seg + 1) + blockIdx.x + 1];
        const uint32_t *pt = reinterpret_cast<const uint32_t *>(buf + p * part_stride);
        const uint16_t *pc = reinterpret_cast<const uint16_t *>(buf + p * part_stride + 4 * cap);
        for (int64_t base = a; base < b; base += (int64_t)BIG_BLOCK * MERGE_BATCH) {
            uint32_t tg[MERGE_BATCH], add[MERGE_BATCH], ss[MERGE_BATCH], first[MERGE_BATCH];
            bool live[MERGE_BATCH];
#pragma unroll
            for (int j = 0; j < MERGE_BATCH; ++j) {
                const int64_t i = base + (int64_t)j * BIG_BLOCK + threadIdx.x;
                live[j] = i < b;
                tg[j] = live[j] ? pt[i] : 0u;
                add[j] = live[j] ? (uint32_t)pc[i] : 0u;
            }
#pragma unroll
            for (int j = 0; j < MERGE_BATCH; ++j) {
                ss[j] = (tg[j] >> hsh) & smask;                     // the slot index is the top bits of the tag
                tg[j] = (tg[j] & tag_mask) | 0x80000000u;
                first[j] = live[j] ? tags[ss[j]] : 0u;
            }
#define PG_MERGE(J)                                                                                                         \
            if (live[J]) {                                                                                                  \
                uint32_t sl = ss[J], cur = first[J];                                                                        \
                bool done = false;                                                                                          \
                for (uint32_t tries = 0; tries < limit && !done; ++tries) {                                                 \
                    if (cur == 0) { cur = atomicCAS(&tags[sl], 0u, tg[J]); if (cur == 0) cur = tg[J]; }                     \
                    if (cur == tg[J]) { atomicAdd(&cnts[sl], add[J]); done = true; }                                        \
                    else { sl = (sl + 1) & smask; cur = tags[sl]; }                                                         \
                }                                                                                                           \
                full |= !done;                                                                                              \
            }
            PG_MERGE(0) PG_MERGE(1) PG_MERGE(2) PG_MERGE(3)
#undef PG_MERGE
        }
    }
    if (full) atomicOr(status, 1u);
    __syncthreads();
    slice_from_planes(tags, cnts, n_slots, tag_mask, (uint64_t)bucket << tag_bits, slice);
}

// Bucket-wise merge of other tables into this one: the compacted tables of the other ranks arrive bucket by bucket
// (a table's slots are laid out by bucket, so compaction keeps bucket order).  One workgroup per bucket loads its slice
// into LDS, adds every foreign entry of that bucket (counts saturate exactly) and writes the slice back.
//   pairs  : the foreign tables' occupied slots, concatenated part after part
//   seg    : [n_parts][n_buckets + 1] offsets into `pairs` (absolute)
// rebuild != 0: the slice is built from the parts alone (they include this rank's own compacted table), so the old
// slice is neither read nor assumed to be initialised.
__device__ __forceinline__ bool lds_merge(unsigned long long *tab, uint32_t smask, uint32_t limit, uint64_t code, uint32_t add,
                                          uint32_t s, unsigned long long cur)
{
    for (uint32_t tries = 0; tries < limit; ++tries) {
        for (;;) {
            if (cur != 0 && (cur >> HASH_CBITS) != code) break;
            const uint32_t have = (uint32_t)(cur & HASH_CMASK);
            const uint32_t sum = have + add > HASH_SAT ? HASH_SAT : have + add;
            const unsigned long long old = atomicCAS(&tab[s], cur, (unsigned long long)((code << HASH_CBITS) | sum));
            if (old == cur) return true;
            cur = old;
        }
        s = (s + 1) & smask;
        cur = tab[s];
    }
    return false;
}

// The launch covers buckets [bucket_base, bucket_base + gridDim.x); seg has n_seg + 1 entries per part, entry j for
// bucket bucket_base + j (a whole-table launch: base 0, n_seg = number of buckets).
__global__ __launch_bounds__(BIG_BLOCK) void bucket_merge_kernel(const uint64_t *__restrict__ pairs, const long long *__restrict__ seg,
                                                                 int n_parts, HashView t, uint32_t *status, int rebuild,
                                                                 int64_t bucket_base, int64_t n_seg)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long tab[];
    const uint32_t n_slots = 1u << t.log2_bucket;
    const uint32_t smask = n_slots - 1;
    const uint32_t limit = t.limit();
    const int hsh = KEY_BITS - t.log2_slots;
    const int64_t n_buckets = n_seg;                                // row stride of seg is n_seg + 1
    uint64_t *slice = t.slots + ((uint64_t)(bucket_base + blockIdx.x) << t.log2_bucket);
    int64_t total = 0;
    for (int p = 0; p < n_parts; ++p) total += seg[p * (n_buckets + 1) + blockIdx.x + 1] - seg[p * (n_buckets + 1) + blockIdx.x];
    if (total == 0 && !rebuild) return;
    for (uint32_t i = threadIdx.x; i < n_slots; i += BIG_BLOCK) tab[i] = rebuild ? 0ull : slice[i];
    __syncthreads();
    bool full = false;
    for (int p = 0; p < n_parts; ++p) {
        const int64_t a = seg[p * (n_buckets + 1) + blockIdx.x], b = seg[p * (n_buckets + 1) + blockIdx.x + 1];
        for (int64_t base = a; base < b; base += (int64_t)BIG_BLOCK * MERGE_BATCH) {
            uint64_t e[MERGE_BATCH];
#pragma unroll
            for (int j = 0; j < MERGE_BATCH; ++j) {                  // a part's whole segment of this bucket is usually one batch
                const int64_t i = base + (int64_t)j * BIG_BLOCK + threadIdx.x;
                e[j] = i < b ? pairs[i] : 0ull;
            }
            uint32_t ss[MERGE_BATCH];
            unsigned long long first[MERGE_BATCH];
#pragma unroll
            for (int j = 0; j < MERGE_BATCH; ++j) {
                ss[j] = (uint32_t)((e[j] >> HASH_CBITS) >> hsh) & smask;
                first[j] = e[j] ? tab[ss[j]] : 0ull;
            }
#define PG_MERGE(J)                                                                                                         \
            if (e[J]) {                                                                                                     \
                uint32_t add = (uint32_t)(e[J] & HASH_CMASK);                                                               \
                if (add > HASH_SAT) add = HASH_SAT;                                                                         \
                full |= !lds_merge(tab, smask, limit, e[J] >> HASH_CBITS, add, ss[J], first[J]);                            \
            }
            PG_MERGE(0) PG_MERGE(1) PG_MERGE(2) PG_MERGE(3)
#undef PG_MERGE
        }
    }
    if (full) atomicOr(status, 1u);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_slots; i += BIG_BLOCK) slice[i] = tab[i];
}

// the merge / rebuild with the split 32-bit LDS table: an entry adds its count with ONE 32-bit LDS add (no CAS loop; the sum
// of all parts' counts stays far below 2^32 and is clamped to SAT when the slots are packed -- the exact saturating sum)
__global__ __launch_bounds__(BIG_BLOCK) void bucket_merge32_kernel(const uint64_t *__restrict__ pairs, const long long *__restrict__ seg,
                                                                   int n_parts, HashView t, uint32_t *status, int rebuild,
                                                                   int64_t bucket_base, int64_t n_seg)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long tab[];
    const uint32_t n_slots = 1u << t.log2_bucket;
    uint32_t *tags = reinterpret_cast<uint32_t *>(tab), *cnts = tags + n_slots;
    const uint32_t smask = n_slots - 1;
    const uint32_t limit = t.limit();
    const int hsh = KEY_BITS - t.log2_slots;
    const int tag_bits = KEY_BITS - (t.log2_slots - t.log2_bucket);
    const uint32_t tag_mask = (1u << tag_bits) - 1u;
    const int64_t bucket = bucket_base + blockIdx.x;
    uint64_t *slice = t.slots + ((uint64_t)bucket << t.log2_bucket);
    int64_t total = 0;
    for (int p = 0; p < n_parts; ++p) total += seg[p * (n_seg + 1) + blockIdx.x + 1] - seg[p * (n_seg + 1) + blockIdx.x];
    if (total == 0 && !rebuild) return;
    planes_from_slice(tags, cnts, n_slots, tag_mask, rebuild ? nullptr : slice);
    __syncthreads();
    bool full = false;
    for (int p = 0; p < n_parts; ++p) {
        const int64_t a = seg[p * (n_seg + 1) + blockIdx.x], b = seg[p * (n_seg + 1) + blockIdx.x + 1];
        for (int64_t base = a; base < b; base += (int64_t)BIG_BLOCK * MERGE_BATCH) {
            uint64_t e[MERGE_BATCH];
#pragma unroll
            for (int j = 0; j < MERGE_BATCH; ++j) {
                const int64_t i = base + (int64_t)j * BIG_BLOCK + threadIdx.x;
                e[j] = i < b ? pairs[i] : 0ull;
            }
            uint32_t ss[MERGE_BATCH], first[MERGE_BATCH];
#pragma unroll
            for (int j = 0; j < MERGE_BATCH; ++j) {
                ss[j] = (uint32_t)((e[j] >> HASH_CBITS) >> hsh) & smask;
                first[j] = e[j] ? tags[ss[j]] : 0u;
            }
#define PG_MERGE(J)                                                                                                         \
            if (e[J]) {                                                                                                     \
                const uint32_t tg = ((uint32_t)(e[J] >> HASH_CBITS) & tag_mask) | 0x80000000u;                              \
                uint32_t add = (uint32_t)(e[J] & HASH_CMASK);                                                               \
                if (add > HASH_SAT) add = HASH_SAT;                                                                         \
                uint32_t sl = ss[J], cur = first[J];                                                                        \
                bool done = false;                                                                                          \
                for (uint32_t tries = 0; tries < limit && !done; ++tries) {                                                 \
                    if (cur == 0) { cur = atomicCAS(&tags[sl], 0u, tg); if (cur == 0) cur = tg; }                           \
                    if (cur == tg) { atomicAdd(&cnts[sl], add); done = true; }                                              \
                    else { sl = (sl + 1) & smask; cur = tags[sl]; }                                                         \
                }                                                                                                           \
                full |= !done;                                                                                              \
            }
            PG_MERGE(0) PG_MERGE(1) PG_MERGE(2) PG_MERGE(3)
#undef PG_MERGE
        }
    }
    if (full) atomicOr(status, 1u);
    __syncthreads();
    slice_from_planes(tags, cnts, n_slots, tag_mask, (uint64_t)bucket << tag_bits, slice);
}

// occupied slots of every bucket: the segment lengths of a bucket-ordered compaction (one workgroup per bucket)
__global__ __launch_bounds__(BLOCK) void bucket_fill_kernel(const uint64_t *__restrict__ slots, int log2_bucket, long long *__restrict__ fill)
{
    __shared__ uint32_t part[WAVES];
    const ulonglong2 *s2 = reinterpret_cast<const ulonglong2 *>(slots + ((uint64_t)blockIdx.x << log2_bucket));
    const uint32_t n2 = 1u << (log2_bucket - 1);
    uint32_t c = 0;
    for (uint32_t i = threadIdx.x; i < n2; i += BLOCK) {
        const ulonglong2 v = s2[i];
        c += (v.x != 0) + (v.y != 0);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (int w = 0; w < WAVES; ++w) tot += part[w];
        fill[blockIdx.x] = (long long)tot;
    }
}

// bucket-ordered compaction: the occupied slots of bucket b go to out[seg[b] .. seg[b + 1]) (seg = exclusive scan of the
// fills; order inside a bucket is free, the merge inserts them one by one)
__global__ __launch_bounds__(BLOCK) void table_compact_kernel(const uint64_t *__restrict__ slots, int log2_bucket,
                                                              const long long *__restrict__ seg, uint64_t *__restrict__ out)
{
    __shared__ uint32_t cursor;
    if (threadIdx.x == 0) cursor = 0;
    __syncthreads();
    const uint64_t *slice = slots + ((uint64_t)blockIdx.x << log2_bucket);
    uint64_t *dst = out + seg[blockIdx.x];
    const uint32_t n = 1u << log2_bucket;
    const uint32_t lane = threadIdx.x & 63;
    for (uint32_t base = 0; base < n; base += BLOCK) {
        const uint32_t i = base + threadIdx.x;
        const uint64_t v = i < n ? slice[i] : 0ull;
        const unsigned long long m = __ballot(v != 0);
        if (m) {
            const int leader = __ffsll((long long)m) - 1;
            uint32_t at = 0;
            if ((int)lane == leader) at = atomicAdd(&cursor, (uint32_t)__popcll(m));
            at = __shfl(at, leader);
            if (v != 0) dst[at + __popcll(m & ((1ull << lane) - 1ull))] = v;
        }
    }
}

// the occupied slots of the launch's buckets (slots points at the first of them) in the 6-byte exchange format: bucket j's
// tags from element tag_elem[j] of the uint32 view of the buffer, its counts from element cnt_elem[j] of the uint16 view;
// remainders of counts >= 0xffff go to the overflow list as whole entries
__global__ __launch_bounds__(BLOCK) void table_compact_planes_kernel(const uint64_t *__restrict__ slots, int log2_bucket,
                                                                     const long long *__restrict__ tag_elem, const long long *__restrict__ cnt_elem,
                                                                     uint32_t tag_mask, uint32_t *__restrict__ out32, uint16_t *__restrict__ out16,
                                                                     uint64_t *__restrict__ ovf, unsigned long long *__restrict__ ovf_count,
                                                                     unsigned long long ovf_cap, uint32_t *status)
{
    __shared__ uint32_t cursor;
    if (threadIdx.x == 0) cursor = 0;
    __syncthreads();
    const uint64_t *slice = slots + ((uint64_t)blockIdx.x << log2_bucket);
    uint32_t *t32 = out32 + tag_elem[blockIdx.x];
    uint16_t *c16 = out16 + cnt_elem[blockIdx.x];
    const uint32_t n = 1u << log2_bucket;
    const uint32_t lane = threadIdx.x & 63;
    for (uint32_t base = 0; base < n; base += BLOCK) {
        const uint32_t i = base + threadIdx.x;
        const uint64_t v = i < n ? slice[i] : 0ull;
        const unsigned long long m = __ballot(v != 0);
        if (m) {
            const int leader = __ffsll((long long)m) - 1;
            uint32_t at = 0;
            if ((int)lane == leader) at = atomicAdd(&cursor, (uint32_t)__popcll(m));
            at = __shfl(at, leader);
            if (v != 0) {
                const uint32_t j = at + __popcll(m & ((1ull << lane) - 1ull));
                const uint32_t c = (uint32_t)(v & HASH_CMASK);
                t32[j] = (uint32_t)(v >> HASH_CBITS) & tag_mask;
                c16[j] = (uint16_t)(c < 0xffffu ? c : 0xffffu);
                if (c > 0xffffu) {
                    const unsigned long long o = atomicAdd(ovf_count, 1ull);
                    if (o < ovf_cap) ovf[o] = (v & ~HASH_CMASK) | (uint64_t)(c - 0xffffu);
                    else atomicOr(status, 2u);
                }
            }
        }
    }
}

// -------------------------------------------------------------------------------- K3 by shuffle

// count of one record's code in the LDS copy of its bucket slice, starting from an already fetched first slot
__device__ __forceinline__ uint32_t lds_lookup(const unsigned long long *tab, uint32_t smask, uint32_t limit, uint64_t code,
                                               uint32_t s, unsigned long long cur, bool *found)
{
    for (uint32_t i = 0; i < limit; ++i) {
        if (cur == 0) break;
        if ((cur >> HASH_CBITS) == code) { *found = true; return (uint32_t)(cur & HASH_CMASK); }
        s = (s + 1) & smask;
        cur = tab[s];
    }
    *found = false;
    return 0;
}

// S1: one workgroup per bucket: the (final, possibly merged) slice is loaded into LDS and every record of the bucket
// that lies inside a row becomes the 32-bit word (row << vbits | count / window) if that bin is < vsize.  Words are
// packed at the front of the bucket's record range (order irrelevant); emit_end[b] = one past the last word.
__global__ __launch_bounds__(BIG_BLOCK) void bucket_lookup_kernel(const uint64_t *__restrict__ rec, const unsigned long long *__restrict__ off,
                                                                  HashView t, uint32_t window, uint32_t vsize, int vbits,
                                                                  uint32_t *__restrict__ words, unsigned long long *__restrict__ emit_end)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long tab[];
    __shared__ uint32_t emitted;
    const uint32_t n_slots = 1u << t.log2_bucket;
    const uint32_t smask = n_slots - 1;
    const uint32_t limit = t.limit();
    const int hsh = KEY_BITS - t.log2_slots;
    const uint64_t *slice = t.slots + ((uint64_t)blockIdx.x << t.log2_bucket);
    const int64_t r0 = (int64_t)off[blockIdx.x], r1 = (int64_t)off[blockIdx.x + 1];
    if (threadIdx.x == 0) emitted = 0;
    if (r0 == r1) { if (threadIdx.x == 0) emit_end[blockIdx.x] = (unsigned long long)r0; return; }
    for (uint32_t i = threadIdx.x; i < n_slots; i += BIG_BLOCK) tab[i] = slice[i];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    const PairRange pr(rec, r0, r1);
    for (int64_t base = pr.q0; base < pr.q1; base += PairRange::stride) {
        ulonglong2 v[CNT_BATCH / 2];
        uint64_t rr[CNT_BATCH];
        bool live[CNT_BATCH];
        pr.load(base, v, ~0ull);                                    // a padding lane carries ROW_NONE
        pr.live(base, live);
#pragma unroll
        for (int j = 0; j < CNT_BATCH / 2; ++j) {
            rr[2 * j] = live[2 * j] ? v[j].x : ~0ull;
            rr[2 * j + 1] = live[2 * j + 1] ? v[j].y : ~0ull;
        }
        uint32_t ss[CNT_BATCH];
        unsigned long long first[CNT_BATCH];
#pragma unroll
        for (int j = 0; j < CNT_BATCH; ++j) {
            live[j] = live[j] && (uint32_t)(rr[j] >> REC_KEY_BITS) != ROW_NONE;
            ss[j] = (uint32_t)((rr[j] & REC_KEY_MASK) >> hsh) & smask;
            first[j] = live[j] ? tab[ss[j]] : 0ull;
        }
#define PG_EMIT(J)                                                                                                          \
        {                                                                                                                   \
            bool found = false;                                                                                             \
            uint32_t cnt = live[J] ? lds_lookup(tab, smask, limit, rr[J] & REC_KEY_MASK, ss[J], first[J], &found) : 0u;     \
            const uint32_t bin = cnt / window;                                                                              \
            const bool put = found && bin < vsize;                                                                          \
            const unsigned long long m = __ballot(put);                                                                     \
            if (m) {                                                                                                        \
                const int leader = __ffsll((long long)m) - 1;                                                               \
                uint32_t at = 0;                                                                                            \
                if ((int)lane == leader) at = atomicAdd(&emitted, (uint32_t)__popcll(m));                                   \
                at = __shfl(at, leader);                                                                                    \
                if (put) words[r0 + at + __popcll(m & ((1ull << lane) - 1ull))] = ((uint32_t)(rr[J] >> REC_KEY_BITS) << vbits) | bin; \
            }                                                                                                               \
        }
        PG_EMIT(0) PG_EMIT(1) PG_EMIT(2) PG_EMIT(3) PG_EMIT(4) PG_EMIT(5) PG_EMIT(6) PG_EMIT(7)
#undef PG_EMIT
    }
    __syncthreads();
    if (threadIdx.x == 0) emit_end[blockIdx.x] = (unsigned long long)r0 + emitted;
}

// S1 with a split LDS table (buckets >= 2^11): tags[] (the key's bits below the bucket id | 0x80000000) and, instead of the
// counts, bins[] = count / window computed ONCE per slot while the slice is loaded (0xffff: bin >= vsize, never emitted).
// A lookup is a 4-byte read and, on a hit, a 2-byte read; the division leaves the per-record path.
__global__ __launch_bounds__(BIG_BLOCK) void bucket_lookup32_kernel(const uint64_t *__restrict__ rec, const unsigned long long *__restrict__ off,
                                                                    HashView t, uint32_t window, uint32_t vsize, int vbits,
                                                                    uint32_t *__restrict__ words, unsigned long long *__restrict__ emit_end)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long tab[];
    __shared__ uint32_t emitted;
    const uint32_t n_slots = 1u << t.log2_bucket;
    uint32_t *tags = reinterpret_cast<uint32_t *>(tab);
    uint16_t *bins = reinterpret_cast<uint16_t *>(tags + n_slots);
    const uint32_t smask = n_slots - 1;
    const uint32_t limit = t.limit();
    const int hsh = KEY_BITS - t.log2_slots;
    const int tag_bits = KEY_BITS - (t.log2_slots - t.log2_bucket);
    const uint32_t tag_mask = (1u << tag_bits) - 1u;
    const uint64_t *slice = t.slots + ((uint64_t)blockIdx.x << t.log2_bucket);
    const int64_t r0 = (int64_t)off[blockIdx.x], r1 = (int64_t)off[blockIdx.x + 1];
    if (threadIdx.x == 0) emitted = 0;
    if (r0 == r1) { if (threadIdx.x == 0) emit_end[blockIdx.x] = (unsigned long long)r0; return; }
    for (uint32_t i = threadIdx.x; i < n_slots; i += BIG_BLOCK) {
        const uint64_t v = slice[i];
        const uint32_t bin = (uint32_t)(v & HASH_CMASK) / window;
        tags[i] = v ? ((uint32_t)(v >> HASH_CBITS) & tag_mask) | 0x80000000u : 0u;
        bins[i] = (uint16_t)(bin < vsize ? bin : 0xffffu);             // vsize <= PG_SHUFFLE_MAX_VSIZE = 512
    }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    const PairRange pr(rec, r0, r1);
    ulonglong2 w[CNT_BATCH / 2];
    pr.load(pr.q0, w, ~0ull);                                       // a padding lane carries ROW_NONE
    for (int64_t base = pr.q0; base < pr.q1; base += PairRange::stride) {
        uint32_t tg[CNT_BATCH], ss[CNT_BATCH], first[CNT_BATCH], row[CNT_BATCH];
        bool live[CNT_BATCH];
        pr.live(base, live);
        split32(w, tag_mask, hsh, smask, tg, ss, row);
        pr.load_next(base, w, ~0ull, tg[0]);
#pragma unroll
        for (int j = 0; j < CNT_BATCH; ++j) {
            live[j] = live[j] && row[j] != ROW_NONE;
            first[j] = live[j] ? tags[ss[j]] : 0u;
        }
#define PG_EMIT(J)                                                                                                          \
        {                                                                                                                   \
            uint32_t bin = 0xffffu;                                                                                         \
            if (live[J]) {                                                                                                  \
                uint32_t sl = ss[J], cur = first[J];                                                                        \
                for (uint32_t i = 0; i < limit && cur != 0; ++i) {                                                          \
                    if (cur == tg[J]) { bin = bins[sl]; break; }                                                            \
                    sl = (sl + 1) & smask;                                                                                  \
                    cur = tags[sl];                                                                                         \
                }                                                                                                           \
            }                                                                                                               \
            const bool put = bin != 0xffffu;                                                                                \
            const unsigned long long m = __ballot(put);                                                                     \
            if (m) {                                                                                                        \
                const int leader = __ffsll((long long)m) - 1;                                                               \
                uint32_t at = 0;                                                                                            \
                if ((int)lane == leader) at = atomicAdd(&emitted, (uint32_t)__popcll(m));                                   \
                at = __shfl(at, leader);                                                                                    \
                if (put) words[r0 + at + __popcll(m & ((1ull << lane) - 1ull))] = (row[J] << vbits) | bin;                  \
            }                                                                                                               \
        }
        PG_EMIT(0) PG_EMIT(1) PG_EMIT(2) PG_EMIT(3) PG_EMIT(4) PG_EMIT(5) PG_EMIT(6) PG_EMIT(7)
#undef PG_EMIT
    }
    __syncthreads();
    if (threadIdx.x == 0) emit_end[blockIdx.x] = (unsigned long long)r0 + emitted;
}

// B and S1 in one kernel (one GPU, a fresh table): while the bucket's counts are still in LDS -- right after the last insert --
// the bucket's records are streamed a second time (they were read a moment ago: L2 / MALL hits) and looked up, so the slice
// is never re-loaded or re-split, and the packed slice image is written out while the lookups run.
__global__ __launch_bounds__(BIG_BLOCK) void bucket_count_emit32_kernel(const uint64_t *__restrict__ rec, const unsigned long long *__restrict__ off,
                                                                        HashView t, uint32_t window, uint32_t vsize, int vbits,
                                                                        uint32_t *__restrict__ words, unsigned long long *__restrict__ emit_end,
                                                                        uint32_t *status)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long tab[];
    __shared__ uint32_t emitted;
    const uint32_t n_slots = 1u << t.log2_bucket;
    uint32_t *tags = reinterpret_cast<uint32_t *>(tab), *cnts = tags + n_slots;
    const uint32_t smask = n_slots - 1;
    const uint32_t limit = t.limit();
    const int hsh = KEY_BITS - t.log2_slots;
    const int tag_bits = KEY_BITS - (t.log2_slots - t.log2_bucket);
    const uint32_t tag_mask = (1u << tag_bits) - 1u;
    uint64_t *slice = t.slots + ((uint64_t)blockIdx.x << t.log2_bucket);
    const int64_t r0 = (int64_t)off[blockIdx.x], r1 = (int64_t)off[blockIdx.x + 1];
    if (threadIdx.x == 0) emitted = 0;
    planes_from_slice(tags, cnts, n_slots, tag_mask, nullptr);
    __syncthreads();
    bool full = false;
    const PairRange pr(rec, r0, r1);
    ulonglong2 v[CNT_BATCH / 2];
    pr.load(pr.q0, v, 0ull);
    for (int64_t base = pr.q0; base < pr.q1; base += PairRange::stride) {       // ---- count
        uint32_t tg[CNT_BATCH], ss[CNT_BATCH], row[CNT_BATCH], first[CNT_BATCH];
        bool live[CNT_BATCH];
        pr.live(base, live);
        split32(v, tag_mask, hsh, smask, tg, ss, row);
        pr.load_next(base, v, 0ull, tg[0]);                         // the next batch's loads fly during the inserts
#pragma unroll
        for (int j = 0; j < CNT_BATCH; ++j) first[j] = live[j] ? tags[ss[j]] : 0u;
#define PG_RESOLVE(J) full |= !lds_insert32(tags, cnts, smask, limit, tg[J], live[J], ss[J], first[J]);
        PG_RESOLVE(0) PG_RESOLVE(1) PG_RESOLVE(2) PG_RESOLVE(3) PG_RESOLVE(4) PG_RESOLVE(5) PG_RESOLVE(6) PG_RESOLVE(7)
#undef PG_RESOLVE
    }
    if (full) atomicOr(status, 1u);
    __syncthreads();
    slice_from_planes(tags, cnts, n_slots, tag_mask, (uint64_t)blockIdx.x << tag_bits, slice);      // ---- the slice image (stores overlap the lookups)
    const uint32_t lane = threadIdx.x & 63;
    pr.load(pr.q0, v, ~0ull);                                       // a padding lane carries ROW_NONE
    for (int64_t base = pr.q0; base < pr.q1; base += PairRange::stride) {       // ---- lookups of the same records
        uint32_t tg[CNT_BATCH], ss[CNT_BATCH], first[CNT_BATCH], row[CNT_BATCH];
        bool live[CNT_BATCH];
        pr.live(base, live);
        split32(v, tag_mask, hsh, smask, tg, ss, row);
        pr.load_next(base, v, ~0ull, tg[0]);
#pragma unroll
        for (int j = 0; j < CNT_BATCH; ++j) {
            live[j] = live[j] && row[j] != ROW_NONE;
            first[j] = live[j] ? tags[ss[j]] : 0u;
        }
#define PG_EMIT(J)                                                                                                          \
        {                                                                                                                   \
            uint32_t bin = vsize;                                                                                           \
            if (live[J]) {                                                                                                  \
                uint32_t sl = ss[J], cur = first[J];                                                                        \
                for (uint32_t i = 0; i < limit && cur != 0; ++i) {                                                          \
                    if (cur == tg[J]) { bin = cnts[sl] / window; break; }                                                   \
                    sl = (sl + 1) & smask;                                                                                  \
                    cur = tags[sl];                                                                                         \
                }                                                                                                           \
            }                                                                                                               \
            const bool put = bin < vsize;                                                                                   \
            const unsigned long long m = __ballot(put);                                                                     \
            if (m) {                                                                                                        \
                const int leader = __ffsll((long long)m) - 1;                                                               \
                uint32_t at = 0;                                                                                            \
                if ((int)lane == leader) at = atomicAdd(&emitted, (uint32_t)__popcll(m));                                   \
                at = __shfl(at, leader);                                                                                    \
                if (put) words[r0 + at + __popcll(m & ((1ull << lane) - 1ull))] = (row[J] << vbits) | bin;                  \
            }                                                                                                               \
        }
        PG_EMIT(0) PG_EMIT(1) PG_EMIT(2) PG_EMIT(3) PG_EMIT(4) PG_EMIT(5) PG_EMIT(6) PG_EMIT(7)
#undef PG_EMIT
    }
    __syncthreads();
    if (threadIdx.x == 0) emit_end[blockIdx.x] = (unsigned long long)r0 + emitted;
}

// capacity of every row group (64 rows): the rows' character counts bound their k-mer counts
__global__ __launch_bounds__(BLOCK) void group_caps_kernel(const int64_t *__restrict__ row_start, const int64_t *__restrict__ row_end,
                                                           int64_t n_rows, int64_t n_groups_padded, unsigned long long *__restrict__ caps)
{
    const int64_t g = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (g >= n_groups_padded) return;
    unsigned long long c = 0;
    for (int64_t r = g << GROUP_ROWS_LOG2; r < ((g + 1) << GROUP_ROWS_LOG2) && r < n_rows; ++r) c += (unsigned long long)(row_end[r] - row_start[r]);
    caps[g] = c;
}

// S3: one workgroup per row group: LDS histogram [64 rows][vsize] of the group's (row, bin) words, written out as the
// rows of the abundance matrix (plain stores: every row belongs to exactly one group)
// WORD = uint16_t: the narrow words of a one-pass shuffle (row inside the group << vbits | bin; 0xffff = none)
// COUNTED (4-byte words): (n - 1) << PG_SHUFFLE_COUNT_SHIFT | row << vbits | bin stands for n equal words -- a lookup pass that
// merges the runs of a record itself (mini.hip) has left hardly any equal neighbours: no run detection here, the count is added
template <typename WORD, bool COUNTED = false>
__global__ __launch_bounds__(BIG_BLOCK) void row_hist_kernel(const WORD *__restrict__ words, const unsigned long long *__restrict__ goff,
                                                             const unsigned long long *__restrict__ gcnt, int vbits, uint32_t vsize,
                                                             int64_t n_rows, int32_t *__restrict__ abd_out, int64_t g0, int split)
{
    // split > 1: `split` workgroups share a group -- each takes a slice of its words and ADDS its partial histogram to rows that
    // were cleared beforehand.  For the groups of a last, mostly empty round of workgroups (one per CU: the histogram fills LDS):
    // 782 groups on 256 CUs are four rounds of which the last holds 14 groups.
    extern __shared__ __attribute__((aligned(16))) uint32_t hist[];
    const int64_t g = g0 + (int64_t)(blockIdx.x / (unsigned)split);
    const int part = (int)(blockIdx.x % (unsigned)split);
    const int64_t row0 = g << GROUP_ROWS_LOG2;
    const int64_t rows_here = n_rows - row0 < (1 << GROUP_ROWS_LOG2) ? n_rows - row0 : (1 << GROUP_ROWS_LOG2);
    // rows lie `stride` words apart in LDS, an ODD number: with the matrix's own stride (V = 400 = 16 mod 32) the same bin of all
    // 64 rows falls on two of the 32 banks, and most k-mers of a sample share a handful of bins -- the adds were bank-conflict bound
    const uint32_t stride = vsize | 1u;
    const uint32_t n_bins = (uint32_t)rows_here * stride;
    for (uint32_t i = threadIdx.x; i < n_bins; i += BIG_BLOCK) hist[i] = 0;
    __syncthreads();
    int64_t a = (int64_t)goff[g], b = a + (int64_t)gcnt[g];
    if (split > 1) {
        const int64_t len = b - a, lo = a + len * part / split, hi = a + len * (part + 1) / split;
        a = lo;
        b = hi;
    }
    const uint32_t bmask = (1u << vbits) - 1u;
    constexpr uint32_t NONE = (uint32_t)(WORD)~(WORD)0;
    // (one workgroup per CU -- the histogram fills LDS -- so nothing else hides the load latency: the next batch's 8 loads per
    // lane are issued before this batch's adds)
    auto fetch = [&](int64_t base, uint32_t (&e)[8]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t i = base + (int64_t)j * BIG_BLOCK + threadIdx.x;
            if (sizeof(WORD) == 2) {
                // (through the dword that holds it: two lanes share a load, and the compiler does not serialise dword loads the
                // way it does 2-byte ones -- it put s_waitcnt vmcnt(0) behind every global_load_ushort)
                const uint32_t pair = i < b ? reinterpret_cast<const uint32_t *>(words)[i >> 1] : 0xffffffffu;
                e[j] = (i & 1) ? pair >> 16 : pair & 0xffffu;
            } else {
                e[j] = i < b ? (uint32_t)words[i] : NONE;
            }
        }
    };
    uint32_t e[8], en[8];
    fetch(a, e);
    for (int64_t base = a; base < b; base += (int64_t)BIG_BLOCK * 8) {
        fetch(base + (int64_t)BIG_BLOCK * 8, en);
        // equal words in neighbouring lanes are added once: consecutive k-mers of a read share their row and, mostly, their bin,
        // and a lookup pass that works on super-k-mers leaves them next to each other -- as single adds they would queue up on
        // one LDS address
        const uint32_t lane = threadIdx.x & 63;
        if (COUNTED) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (e[j] != NONE)
                    atomicAdd(&hist[((e[j] >> vbits) & ((1u << GROUP_ROWS_LOG2) - 1u)) * stride + (e[j] & bmask)], (e[j] >> PG_SHUFFLE_COUNT_SHIFT) + 1u);
        } else
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t prev = __shfl_up(e[j], 1);
            const bool head = lane == 0 || e[j] != prev;
            const unsigned long long hm = __ballot(head);
            const unsigned long long above = lane == 63 ? 0ull : hm >> (lane + 1);
            const uint32_t run = above ? (uint32_t)__ffsll((long long)above) : 64u - lane;
            if (head && e[j] != NONE) atomicAdd(&hist[((e[j] >> vbits) & ((1u << GROUP_ROWS_LOG2) - 1u)) * stride + (e[j] & bmask)], run);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = en[j];
    }
    __syncthreads();
    int32_t *dst = abd_out + row0 * (int64_t)vsize;
    const uint32_t n_out = (uint32_t)rows_here * vsize;
    if (split > 1) {
        for (uint32_t i = threadIdx.x; i < n_out; i += BIG_BLOCK) {
            const uint32_t v = hist[(i / vsize) * stride + i % vsize];
            if (v) atomicAdd(&dst[i], (int32_t)v);
        }
        return;
    }
    for (uint32_t i = threadIdx.x; i < n_out; i += BIG_BLOCK) dst[i] = (int32_t)hist[(i / vsize) * stride + i % vsize];
}

// -------------------------------------------------------------------------------- K1 + K3 by lookups: per-run rows

// LDS: abd_copies x [vsize] abundance bins, then tnf_copies x [4^k_tnf] raw k_tnf-mer bins
// (one copy per wavefront while that fits comfortably, else one shared copy).
template <typename KT, int TK>
__global__ __launch_bounds__(BLOCK) void features_kernel(const uint64_t *__restrict__ codes, const uint32_t *__restrict__ valid,
                                                         int64_t n_words, const int32_t *__restrict__ seg_row,
                                                         const int64_t *__restrict__ seg_start, const int64_t *__restrict__ seg_end,
                                                         int k_tnf, const uint16_t *__restrict__ colmap, int tnf_cols, int tnf_copies, int abd_copies,
                                                         int32_t *__restrict__ tnf_out, int k, const uint32_t *__restrict__ dense, HashView t,
                                                         uint32_t window, int vsize, int32_t *__restrict__ abd_out)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const bool do_tnf = tnf_out != nullptr;
    const bool do_abd = TK != TK_NONE && abd_out != nullptr;
    const int n_raw = do_tnf ? 1 << (2 * k_tnf) : 0;
    const int abd_bins = do_abd ? vsize : 0;
    uint32_t *abd_lds = lds;
    uint32_t *tnf_lds = lds + abd_copies * abd_bins;
    const int lds_words = abd_copies * abd_bins + tnf_copies * n_raw;
    for (int i = threadIdx.x; i < lds_words; i += BLOCK) lds[i] = 0;
    __syncthreads();

    const int wave = threadIdx.x >> 6;
    uint32_t *my_abd = abd_lds + (abd_copies > 1 ? wave * abd_bins : 0);
    uint32_t *my_tnf = tnf_lds + (tnf_copies > 1 ? wave * n_raw : 0);

    const int64_t s0 = seg_start[blockIdx.x], s1 = seg_end[blockIdx.x];
    const int kk = do_abd ? k : k_tnf;                                       // what the roller rolls
    const int kroll = do_abd ? (k > k_tnf || !do_tnf ? k : k_tnf) : k_tnf;   // characters to pre-roll + 1
    const uint32_t tmask = do_tnf ? (uint32_t)n_raw - 1u : 0u;

    for (int64_t w = (s0 >> 5) + threadIdx.x; w <= ((s1 - 1) >> 5) && w < n_words; w += BLOCK) {
        const uint64_t cw = codes[w];
        const uint32_t vw = valid[w];
        const uint64_t pw = w > 0 ? codes[w - 1] : 0;
        const uint32_t pv = w > 0 ? valid[w - 1] : 0;
        // characters of this word that belong to the segment
        const int64_t base = w << 5;
        const int lo = s0 > base ? (int)(s0 - base) : 0;
        const int hi = s1 < base + 32 ? (int)(s1 - base) : 32;
        uint32_t in_seg = (hi >= 32 ? 0xffffffffu : ((1u << hi) - 1u)) & ~((1u << lo) - 1u);
        const uint64_t m = ((uint64_t)vw << 32) | pv;
        const uint32_t ok_t = do_tnf ? (uint32_t)(runs_of(m, k_tnf) >> 32) & in_seg : 0u;
        const uint32_t ok_a = do_abd ? (uint32_t)(runs_of(m, k) >> 32) & in_seg : 0u;
        if ((ok_t | ok_a) == 0) continue;

        Roller<KT> r;
        r.init(kk);
        for (int i = 33 - kroll; i < 32; ++i) r.push((uint32_t)(pw >> (2 * i)) & 3u);
        // batches of 8 characters: roll, issue the table reads of the batch, then bin them
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            KT canon[8];
            uint64_t cur[8];
            uint64_t hh[8];
            uint64_t key8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = b * 8 + u;
                r.push((uint32_t)(cw >> (2 * j)) & 3u);
                if (do_tnf && ((ok_t >> j) & 1)) atomicAdd(&my_tnf[(uint32_t)r.fw & tmask], 1u);
                if (TK != TK_NONE) {
                    canon[u] = r.canon();
                    cur[u] = 0;
                    if ((ok_a >> j) & 1) {
                        if (TK == TK_DENSE) {
                            cur[u] = dense[(uint32_t)canon[u]];
                        } else {
                            key8[u] = TK == TK_HASH ? key42((uint64_t)canon[u]) : (uint64_t)canon[u];
                            hh[u] = TK == TK_WIDE ? t.home_wide(key8[u]) : (TK == TK_MINI || TK == TK_MINIW) ? t.home_mini(key8[u], k) : t.home_key(key8[u]);
                            // each table line is used once per launch: keep it out of the way of the stream (measured -2 %)
                            cur[u] = __builtin_nontemporal_load(&t.slots[hh[u]]);
                        }
                    }
                }
            }
            if (TK != TK_NONE) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int j = b * 8 + u;
                    if ((ok_a >> j) & 1) {
                        uint32_t cnt;
                        bool found = true;
                        if (TK == TK_DENSE) {
                            cnt = (uint32_t)cur[u];
                            found = cnt != 0;       // absent from the table <=> never counted
                        } else if (TK == TK_WIDE || TK == TK_MINIW) {
                            cnt = wide_probe(t, hh[u], cur[u], key8[u], &found);
                        } else {
                            cnt = hash_probe(t, hh[u], cur[u], key8[u], &found);
                        }
                        if (found) {
                            uint32_t bin = cnt / window;
                            if (bin < (uint32_t)vsize) atomicAdd(&my_abd[bin], 1u);
                        }
                    }
                }
            }
        }
    }
    __syncthreads();

    const int64_t row = seg_row[blockIdx.x];
    if (do_abd) {
        for (int i = threadIdx.x; i < abd_bins; i += BLOCK) {
            uint32_t s = 0;
            for (int c = 0; c < abd_copies; ++c) s += abd_lds[c * abd_bins + i];
            if (s) atomicAdd(&abd_out[row * vsize + i], (int32_t)s);
        }
    }
    if (do_tnf) {
        for (int i = threadIdx.x; i < n_raw; i += BLOCK) {
            uint32_t s = 0;
            for (int c = 0; c < tnf_copies; ++c) s += tnf_lds[c * n_raw + i];
            if (s) atomicAdd(&tnf_out[row * tnf_cols + colmap[i]], (int32_t)s);
        }
    }
}

// -------------------------------------------------------------------------------- launch helpers
int check_table(const pg_table *t)
{
    if (!t || !t->data) return pg_fail(PG_EINVAL, "table descriptor is null");
    if (t->kind == PG_TABLE_DENSE) {
        if (t->k < 1 || t->k > PG_DENSE_MAX_K) return pg_fail(PG_EINVAL, "dense table needs 1 <= k <= %d (got %d)", PG_DENSE_MAX_K, t->k);
    } else if (t->kind == PG_TABLE_HASH) {
        if (t->k < 1 || t->k > PG_HASH_MAX_K) return pg_fail(PG_EINVAL, "hash table needs 1 <= k <= %d (got %d)", PG_HASH_MAX_K, t->k);
        if (t->log2_slots < 10 || t->log2_slots > 40) return pg_fail(PG_EINVAL, "log2_slots %d out of range [10,40]", t->log2_slots);
        if (t->log2_bucket_slots != 0 && (t->log2_bucket_slots < 4 || t->log2_bucket_slots > t->log2_slots))
            return pg_fail(PG_EINVAL, "log2_bucket_slots %d out of range [4,%d]", t->log2_bucket_slots, t->log2_slots);
    } else if (t->kind == PG_TABLE_MINI) {
        if (t->k < PG_MINI_MIN_K || t->k > PG_HASH_MAX_K) return pg_fail(PG_EINVAL, "mini table needs %d <= k <= %d (got %d)", PG_MINI_MIN_K, PG_HASH_MAX_K, t->k);
        if (t->log2_bucket_slots < 4 || t->log2_bucket_slots > PG_BUCKET_MAX_LOG2_SLOTS || t->log2_slots < t->log2_bucket_slots ||
            t->log2_slots - t->log2_bucket_slots > PG_MINI_MAX_LOG2_BUCKETS)
            return pg_fail(PG_EINVAL, "mini table geometry 2^%d slots in buckets of 2^%d", t->log2_slots, t->log2_bucket_slots);
    } else if (t->kind == PG_TABLE_MINI_WIDE) {
        if (t->k <= PG_HASH_MAX_K || t->k > PG_WIDE_MAX_K) return pg_fail(PG_EINVAL, "wide mini table needs %d < k <= %d (got %d)", PG_HASH_MAX_K, PG_WIDE_MAX_K, t->k);
        if (t->log2_bucket_slots < 4 || t->log2_bucket_slots > PG_MINI_WIDE_MAX_LOG2_BUCKET_SLOTS || t->log2_slots < t->log2_bucket_slots ||
            t->log2_slots - t->log2_bucket_slots > PG_MINI_MAX_LOG2_BUCKETS)
            return pg_fail(PG_EINVAL, "wide mini table geometry 2^%d slots in buckets of 2^%d", t->log2_slots, t->log2_bucket_slots);
    } else if (t->kind == PG_TABLE_WIDE) {
        if (t->k < 1 || t->k > PG_WIDE_MAX_K) return pg_fail(PG_EINVAL, "wide table needs 1 <= k <= %d (got %d)", PG_WIDE_MAX_K, t->k);
        if (t->log2_slots < 10 || t->log2_slots > 40) return pg_fail(PG_EINVAL, "log2_slots %d out of range [10,40]", t->log2_slots);
        if (t->log2_bucket_slots != 0) return pg_fail(PG_EINVAL, "wide tables are not bucketed");
    } else {
        return pg_fail(PG_EINVAL, "unknown table kind %d", t->kind);
    }
    return PG_OK;
}

HashView view_of(const pg_table *t)
{
    HashView v;
    v.slots = (uint64_t *)t->data;
    v.log2_slots = t->log2_slots;
    v.log2_bucket = t->log2_bucket_slots ? t->log2_bucket_slots : t->log2_slots;
    return v;
}

// workspace carving of the bucketed counter
struct BucketPlan {
    int bits, bits1, bits2;           // bucket id bits, split over the two scatter passes
    int64_t cap, n_tiles, n_chunks;   // record capacity of each record buffer; stream tiles and chunks of the range
    size_t hist_off, off_off, cur1_off, cur2_off, tile_off, chunk_off, bufa_off, bufb_off, total;
    size_t final_off() const { return bits2 ? bufb_off : bufa_off; }
};

int plan_buckets(const pg_table *t, int64_t n_words, BucketPlan *p)
{
    if (t->kind != PG_TABLE_HASH || t->log2_bucket_slots == 0)
        return pg_fail(PG_EINVAL, "bucketed counting needs a hash table with log2_bucket_slots set");
    if (t->log2_bucket_slots > PG_BUCKET_MAX_LOG2_SLOTS)
        return pg_fail(PG_EINVAL, "log2_bucket_slots %d exceeds the LDS-resident maximum %d", t->log2_bucket_slots, PG_BUCKET_MAX_LOG2_SLOTS);
    p->bits = t->log2_slots - t->log2_bucket_slots;
    if (p->bits < 1 || p->bits > PG_BUCKET_MAX_LOG2_BUCKETS)
        return pg_fail(PG_EINVAL, "bucketed counting needs 1 <= log2_slots - log2_bucket_slots <= %d (got %d)", PG_BUCKET_MAX_LOG2_BUCKETS, p->bits);
    p->bits1 = p->bits < 8 ? p->bits : 8;
    p->bits2 = p->bits - p->bits1;
    if (p->bits2 > MAX_FAN_BITS) return pg_fail(PG_EINVAL, "too many buckets for two scatter passes");
    p->cap = n_words * 32;
    p->n_tiles = (n_words + TILE_WORDS - 1) / TILE_WORDS;
    p->n_chunks = (n_words + CHUNK_WORDS - 1) / CHUNK_WORDS;
    const size_t nb = (size_t)1 << p->bits;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += (bytes + 255) / 256 * 256; return at; };
    p->hist_off = take(nb * 8);
    p->off_off = take((nb + 1) * 8);
    p->cur1_off = take(((size_t)1 << p->bits1) * 8);
    p->cur2_off = take(nb * 8);
    p->tile_off = take((size_t)(p->n_tiles + 1) * 4);
    p->chunk_off = take(((size_t)p->n_chunks << p->bits1) * 8);
    p->bufa_off = take((size_t)p->cap * 8);
    p->bufb_off = p->bits2 ? take((size_t)p->cap * 8) : p->bufa_off;
    p->total = o;
    return PG_OK;
}

int check_rows(const pg_rows *rows, const char *who)
{
    if (!rows) return PG_OK;
    if (rows->n_rows < 0 || rows->n_rows >= (int64_t)ROW_NONE)
        return pg_fail(PG_EINVAL, "%s: %lld rows (at most %u per launch)", who, (long long)rows->n_rows, ROW_NONE - 1);
    if (rows->n_rows > 0 && (!rows->row_start || !rows->row_end)) return pg_fail(PG_EINVAL, "%s: null row arrays", who);
    return PG_OK;
}

// carving of the shuffle workspace
struct ShufflePlan {
    int vbits, gbits, gb1, gb2;
    int64_t n_groups, n_groups_padded;
    size_t emit_off, caps_off, goff_off, gcur1_off, gcur2_off, dhist_off, words_e_off, words_a_off, words_b_off, total;
};

// no_input (callers that keep their provisional data elsewhere -- the merged lookups of mini.hip): when one scatter pass suffices the
// first word buffer is not needed at all (12 GB per 10 M read pairs that were allocated and never touched)
int plan_shuffle(int64_t cap, int64_t n_rows, int vsize, int64_t n_buckets, ShufflePlan *p, int one_pass_bits = WIDE_FAN_BITS, int no_input = 0)
{
    if (vsize < 1 || vsize > PG_SHUFFLE_MAX_VSIZE)
        return pg_fail(PG_EINVAL, "the shuffle path needs 1 <= vector size <= %d (got %d)", PG_SHUFFLE_MAX_VSIZE, vsize);
    p->vbits = 1;
    while ((1 << p->vbits) < vsize) ++p->vbits;
    p->n_groups = (n_rows + (1 << GROUP_ROWS_LOG2) - 1) >> GROUP_ROWS_LOG2;
    if (p->n_groups < 1) p->n_groups = 1;
    p->gbits = 0;
    while (((int64_t)1 << p->gbits) < p->n_groups) ++p->gbits;
    p->gb1 = p->gbits < 8 ? p->gbits : 8;
    // one pass up to 2^one_pass_bits row groups: 2^10 (65536 rows) for the scatter kernels of this file; a lookup pass that
    // scatters by itself may manage more (mini.hip: 2^11)
    if (p->gbits <= one_pass_bits && !getenv("PG_S2_TWO_PASS")) p->gb1 = p->gbits;
    p->gb2 = p->gbits - p->gb1;
    if (p->gb2 > MAX_FAN_BITS) return pg_fail(PG_EINVAL, "too many rows for two scatter passes");
    p->n_groups_padded = (int64_t)1 << p->gbits;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += (bytes + 255) / 256 * 256; return at; };
    p->emit_off = take(((size_t)1 << PG_BUCKET_MAX_LOG2_BUCKETS) * 8);
    p->caps_off = take((size_t)p->n_groups_padded * 8);
    p->goff_off = take((size_t)(p->n_groups_padded + 1) * 8);
    p->gcur1_off = take(((size_t)1 << p->gb1) * 8);
    p->gcur2_off = take((size_t)p->n_groups_padded * 8);
    p->dhist_off = o;
    (void)n_buckets;
    p->words_e_off = take(no_input && p->gb2 == 0 ? 0 : (size_t)cap * 4);   // emitted by the lookup pass, bucket order
    p->words_a_off = take((size_t)cap * 4);                              // after the first row pass
    p->words_b_off = p->gb2 ? p->words_e_off : p->words_a_off;           // the second row pass reuses the first buffer
    p->total = o;
    return PG_OK;
}

}  // namespace

extern "C" int pg_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return pg_fail(PG_ENODEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}

extern "C" int pg_kmer_count(const uint64_t *codes, const uint32_t *valid, int64_t word_begin, int64_t word_end,
                             const pg_table *t, uint32_t *status, void *stream)
{
    if (!codes || !valid) return pg_fail(PG_EINVAL, "pg_kmer_count: null stream arrays");
    if (word_begin < 0 || word_end < word_begin) return pg_fail(PG_EINVAL, "pg_kmer_count: bad word range [%lld,%lld)", (long long)word_begin, (long long)word_end);
    int rc = check_table(t);
    if (rc) return rc;
    if (t->kind == PG_TABLE_MINI || t->kind == PG_TABLE_MINI_WIDE) return pg_fail(PG_EINVAL, "pg_kmer_count: mini tables are built by pg_mini_plan + pg_mini_count");
    if (t->kind != PG_TABLE_DENSE && !status) return pg_fail(PG_EINVAL, "pg_kmer_count: hash tables need a status word");
    if (word_end == word_begin) return PG_OK;
    hipStream_t s = (hipStream_t)stream;
    int grid = grid_for(word_end - word_begin);
    if (t->kind == PG_TABLE_DENSE) {
        HashView none{nullptr, 0, 0};
        hipLaunchKernelGGL((kmer_count_kernel<uint32_t, TK_DENSE>), dim3(grid), dim3(BLOCK), 0, s, codes, valid, word_begin, word_end,
                           t->k, (uint32_t *)t->data, none, status);
    } else if (t->kind == PG_TABLE_WIDE) {
        hipLaunchKernelGGL((kmer_count_kernel<uint64_t, TK_WIDE>), dim3(grid), dim3(BLOCK), 0, s, codes, valid, word_begin, word_end,
                           t->k, (uint32_t *)nullptr, view_of(t), status);
    } else {
        hipLaunchKernelGGL((kmer_count_kernel<uint64_t, TK_HASH>), dim3(grid), dim3(BLOCK), 0, s, codes, valid, word_begin, word_end,
                           t->k, (uint32_t *)nullptr, view_of(t), status);
    }
    return check_launch("pg_kmer_count");
}

extern "C" int pg_kmer_distinct_sketch(const uint64_t *codes, const uint32_t *valid, int64_t word_begin, int64_t word_end, int k,
                                       uint32_t *registers, void *stream)
{
    if (!codes || !valid || !registers) return pg_fail(PG_EINVAL, "pg_kmer_distinct_sketch: null argument");
    if (k < 1 || k > PG_WIDE_MAX_K) return pg_fail(PG_EINVAL, "pg_kmer_distinct_sketch: k %d out of range", k);
    if (word_begin < 0 || word_end < word_begin) return pg_fail(PG_EINVAL, "pg_kmer_distinct_sketch: bad word range");
    if (word_end == word_begin) return PG_OK;
    int grid = (int)((word_end - word_begin + BIG_BLOCK - 1) / BIG_BLOCK);
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(distinct_sketch_kernel, dim3(grid), dim3(BIG_BLOCK), 0, (hipStream_t)stream, codes, valid, word_begin, word_end, k, registers);
    return check_launch("pg_kmer_distinct_sketch");
}

extern "C" int64_t pg_kmer_count_workspace_bytes(int64_t n_words, const pg_table *t)
{
    if (n_words < 0) return pg_fail(PG_EINVAL, "negative word count");
    int rc = check_table(t);
    if (rc) return rc;
    BucketPlan p;
    rc = plan_buckets(t, n_words, &p);
    if (rc) return rc;
    return (int64_t)p.total;
}

namespace {
struct EmitArgs { int window, vsize; void *shuffle_ws; int64_t shuffle_ws_bytes; };     // fused count + lookup (one GPU)
int count_bucketed_impl(const uint64_t *codes, const uint32_t *valid, int64_t word_begin, int64_t word_end,
                        const pg_table *t, int accumulate, const pg_rows *rows, void *workspace, int64_t workspace_bytes,
                        uint32_t *status, void *stream, int deferred_group, int64_t *fill, const EmitArgs *emit = nullptr);
}

extern "C" int pg_kmer_count_bucketed(const uint64_t *codes, const uint32_t *valid, int64_t word_begin, int64_t word_end,
                                      const pg_table *t, int accumulate, const pg_rows *rows, void *workspace, int64_t workspace_bytes,
                                      uint32_t *status, void *stream)
{
    return count_bucketed_impl(codes, valid, word_begin, word_end, t, accumulate, rows, workspace, workspace_bytes, status, stream, -1, nullptr);
}

extern "C" int pg_kmer_count_deferred(const uint64_t *codes, const uint32_t *valid, int64_t word_begin, int64_t word_end,
                                      const pg_table *t, int group_log2, const pg_rows *rows, void *workspace, int64_t workspace_bytes,
                                      int64_t *fill, uint32_t *status, void *stream)
{
    if (group_log2 < 0 || group_log2 > PG_DEFERRED_MAX_GROUP_LOG2) return pg_fail(PG_EINVAL, "pg_kmer_count_deferred: group_log2 %d not in [0,%d]", group_log2, PG_DEFERRED_MAX_GROUP_LOG2);
    if (!fill) return pg_fail(PG_EINVAL, "pg_kmer_count_deferred: null fill array");
    return count_bucketed_impl(codes, valid, word_begin, word_end, t, 0, rows, workspace, workspace_bytes, status, stream, group_log2, fill);
}

extern "C" int pg_deferred_gather(const pg_table *t, const void *count_workspace, int64_t count_workspace_bytes, int64_t n_words_counted,
                                  const int64_t *fill, const int64_t *seg, uint64_t *out, void *stream)
{
    int rc = check_table(t);
    if (rc) return rc;
    if (!count_workspace || !fill || !seg || !out) return pg_fail(PG_EINVAL, "pg_deferred_gather: null argument");
    BucketPlan p;
    if ((rc = plan_buckets(t, n_words_counted, &p))) return rc;
    if (!p.bits2) return pg_fail(PG_EINVAL, "pg_deferred_gather: the deferred form needs more than 256 buckets");
    if ((int64_t)p.total > count_workspace_bytes) return pg_fail(PG_EINVAL, "pg_deferred_gather: count workspace does not match n_words_counted");
    const char *ws = (const char *)count_workspace;
    const int64_t nb = (int64_t)1 << p.bits;
    hipLaunchKernelGGL(deferred_gather_kernel, dim3((unsigned)((nb + WAVES - 1) / WAVES)), dim3(BLOCK), 0, (hipStream_t)stream,
                       (const uint64_t *)(ws + p.bufa_off), (const unsigned long long *)(ws + p.off_off), (const long long *)fill,
                       (const long long *)seg, nb, out);
    return check_launch("pg_deferred_gather");
}

extern "C" int pg_deferred_gather_planes(const pg_table *t, const void *count_workspace, int64_t count_workspace_bytes, int64_t n_words_counted,
                                         const int64_t *fill, const int64_t *tag_elem, const int64_t *cnt_elem, void *out,
                                         uint64_t *overflow, uint64_t *overflow_count, int64_t overflow_cap, uint32_t *status, void *stream)
{
    int rc = check_table(t);
    if (rc) return rc;
    if (!count_workspace || !fill || !tag_elem || !cnt_elem || !out || !overflow || !overflow_count || !status)
        return pg_fail(PG_EINVAL, "pg_deferred_gather_planes: null argument");
    BucketPlan p;
    if ((rc = plan_buckets(t, n_words_counted, &p))) return rc;
    if (!p.bits2) return pg_fail(PG_EINVAL, "pg_deferred_gather_planes: the deferred form needs more than 256 buckets");
    if (KEY_BITS - p.bits > 31) return pg_fail(PG_EINVAL, "pg_deferred_gather_planes: needs at least 2^11 buckets (tags of at most 31 bits)");
    if ((int64_t)p.total > count_workspace_bytes) return pg_fail(PG_EINVAL, "pg_deferred_gather_planes: count workspace does not match n_words_counted");
    const char *ws = (const char *)count_workspace;
    const int64_t nb = (int64_t)1 << p.bits;
    hipLaunchKernelGGL(deferred_gather_planes_kernel, dim3((unsigned)((nb + WAVES - 1) / WAVES)), dim3(BLOCK), 0, (hipStream_t)stream,
                       (const uint64_t *)(ws + p.bufa_off), (const unsigned long long *)(ws + p.off_off), (const long long *)fill,
                       (const long long *)tag_elem, (const long long *)cnt_elem, nb, (uint32_t)((1u << (KEY_BITS - p.bits)) - 1u),
                       (uint32_t *)out, (uint16_t *)out, overflow, (unsigned long long *)overflow_count, (unsigned long long)overflow_cap, status);
    return check_launch("pg_deferred_gather_planes");
}

extern "C" int pg_kmer_rebuild_planes_range(const void *buf, int64_t part_stride_bytes, int64_t cap, const int64_t *seg, int n_parts,
                                            const pg_table *t, int64_t bucket_begin, int64_t bucket_end, uint32_t *status, void *stream)
{
    int rc = check_table(t);
    if (rc) return rc;
    if (t->kind != PG_TABLE_HASH || t->log2_bucket_slots == 0 || t->log2_bucket_slots > PG_BUCKET_MAX_LOG2_SLOTS)
        return pg_fail(PG_EINVAL, "pg_kmer_rebuild_planes_range: needs a bucketed hash table with LDS-sized buckets");
    const int bits = t->log2_slots - t->log2_bucket_slots;
    if (KEY_BITS - bits > 31) return pg_fail(PG_EINVAL, "pg_kmer_rebuild_planes_range: needs at least 2^11 buckets (tags of at most 31 bits)");
    if (n_parts < 1 || !buf || !seg || !status || cap < 0 || part_stride_bytes < 6 * cap) return pg_fail(PG_EINVAL, "pg_kmer_rebuild_planes_range: bad arguments");
    if (bucket_begin < 0 || bucket_end < bucket_begin || bucket_end > ((int64_t)1 << bits))
        return pg_fail(PG_EINVAL, "pg_kmer_rebuild_planes_range: bucket range [%lld,%lld) outside the table", (long long)bucket_begin, (long long)bucket_end);
    if (bucket_end == bucket_begin) return PG_OK;
    const size_t lds = (size_t)8 << t->log2_bucket_slots;
    if ((rc = raise_lds_limit((const void *)bucket_merge_planes_kernel, lds, "pg_kmer_rebuild_planes_range"))) return rc;
    hipLaunchKernelGGL(bucket_merge_planes_kernel, dim3((unsigned)(bucket_end - bucket_begin)), dim3(BIG_BLOCK), lds, (hipStream_t)stream,
                       (const uint8_t *)buf, part_stride_bytes, cap, (const long long *)seg, n_parts, view_of(t), status, bucket_begin,
                       bucket_end - bucket_begin);
    return check_launch("pg_kmer_rebuild_planes_range");
}

extern "C" int pg_kmer_count_bucketed_emit(const uint64_t *codes, const uint32_t *valid, int64_t word_begin, int64_t word_end,
                                           const pg_table *t, const pg_rows *rows, void *workspace, int64_t workspace_bytes,
                                           int window, int vsize, void *shuffle_workspace, int64_t shuffle_workspace_bytes,
                                           uint32_t *status, void *stream)
{
    if (!rows || rows->n_rows < 1) return pg_fail(PG_EINVAL, "pg_kmer_count_bucketed_emit: needs rows");
    if (!shuffle_workspace) return pg_fail(PG_EINVAL, "pg_kmer_count_bucketed_emit: null shuffle workspace");
    if (window < 1 || (int64_t)window * vsize > (int64_t)PG_HASH_COUNT_SAT)
        return pg_fail(PG_EINVAL, "pg_kmer_count_bucketed_emit: window %d x vector size %d outside the exact range of the hash table", window, vsize);
    const EmitArgs e{window, vsize, shuffle_workspace, shuffle_workspace_bytes};
    return count_bucketed_impl(codes, valid, word_begin, word_end, t, 0, rows, workspace, workspace_bytes, status, stream, -1, nullptr, &e);
}

namespace {
int count_bucketed_impl(const uint64_t *codes, const uint32_t *valid, int64_t word_begin, int64_t word_end,
                        const pg_table *t, int accumulate, const pg_rows *rows, void *workspace, int64_t workspace_bytes,
                        uint32_t *status, void *stream, int deferred_group, int64_t *fill, const EmitArgs *emit)
{
    if (!codes || !valid || !workspace || !status) return pg_fail(PG_EINVAL, "pg_kmer_count_bucketed: null argument");
    if (word_begin < 0 || word_end < word_begin) return pg_fail(PG_EINVAL, "pg_kmer_count_bucketed: bad word range");
    int rc = check_table(t);
    if (rc) return rc;
    if ((rc = check_rows(rows, "pg_kmer_count_bucketed"))) return rc;
    BucketPlan p;
    rc = plan_buckets(t, word_end - word_begin, &p);
    if (rc) return rc;
    if ((int64_t)p.total > workspace_bytes)
        return pg_fail(PG_EINVAL, "pg_kmer_count_bucketed: workspace of %lld bytes, %lld needed", (long long)workspace_bytes, (long long)p.total);
    if ((reinterpret_cast<uintptr_t>(workspace) & 255) != 0) return pg_fail(PG_EINVAL, "pg_kmer_count_bucketed: workspace must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    if (deferred_group >= 0) {
        if (!p.bits2) return pg_fail(PG_EINVAL, "pg_kmer_count_deferred: the deferred form needs more than 256 buckets");
        if (deferred_group > p.bits2) return pg_fail(PG_EINVAL, "pg_kmer_count_deferred: group_log2 %d exceeds the second-pass bits %d", deferred_group, p.bits2);
        if (word_end == word_begin) {
            if (hipMemsetAsync(fill, 0, sizeof(int64_t) << p.bits, s) != hipSuccess) return pg_fail(PG_EHIP, "pg_kmer_count_deferred: memset failed");
            return PG_OK;
        }
    }
    if (word_end == word_begin) return PG_OK;
    char *ws = (char *)workspace;
    auto *hist = (unsigned long long *)(ws + p.hist_off);
    auto *off = (unsigned long long *)(ws + p.off_off);
    auto *cur2 = (unsigned long long *)(ws + p.cur2_off);
    auto *tile_row = (int32_t *)(ws + p.tile_off);
    auto *chunk_tab = (unsigned long long *)(ws + p.chunk_off);
    auto *bufa = (uint64_t *)(ws + p.bufa_off);
    auto *bufb = (uint64_t *)(ws + p.bufb_off);
    const int nb = 1 << p.bits;
    // counters are contiguous at the front of the workspace: one clear
    if (hipMemsetAsync(ws, 0, p.tile_off, s) != hipSuccess) return pg_fail(PG_EHIP, "pg_kmer_count_bucketed: memset failed");
    const size_t slice_lds = (size_t)8 << t->log2_bucket_slots;
    const int hist_bins = nb < (1 << 15) ? nb : (1 << 15);
    const size_t hist_lds = (size_t)hist_bins * 4 + 1024;
    if ((rc = raise_lds_limit((const void *)bucket_hist_kernel, hist_lds, "pg_kmer_count_bucketed"))) return rc;
    if ((rc = raise_lds_limit((const void *)bucket_count_kernel, slice_lds, "pg_kmer_count_bucketed"))) return rc;

    const bool with_rows = rows && rows->n_rows > 0;
    if (with_rows)
        hipLaunchKernelGGL(tile_rows_kernel, dim3((unsigned)((p.n_tiles + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s, rows->row_end, rows->n_rows,
                           word_begin, p.n_tiles, tile_row);
    // golden-ratio stride, made coprime to the chunk count: a bijection on chunk indices
    int64_t chunk_stride = (int64_t)((double)p.n_chunks * 0.6180339887) | 1;
    {
        auto gcd = [](int64_t a, int64_t b) { while (b) { int64_t t = a % b; a = b; b = t; } return a; };
        while (gcd(chunk_stride, p.n_chunks) != 1) chunk_stride += 2;
    }
    // A0 histogram of final bucket ids + per-chunk first-digit counts, then offsets of buckets and of chunks
    {
        const int grid = (int)(p.n_chunks < 512 ? p.n_chunks : 512);
        for (int base = 0; base < nb; base += hist_bins)
            hipLaunchKernelGGL(bucket_hist_kernel, dim3(grid), dim3(BIG_BLOCK), hist_lds, s, codes, valid, word_begin, word_end,
                               t->k, p.bits, (uint32_t)base, hist_bins, hist, p.bits1, base == 0 ? chunk_tab : (unsigned long long *)nullptr, p.n_chunks,
                               chunk_stride);
        hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(BIG_BLOCK), 0, s, (const unsigned long long *)hist, (int64_t)nb, off);
        hipLaunchKernelGGL(digit_scan_kernel, dim3(1u << p.bits1), dim3(BIG_BLOCK), 0, s, chunk_tab, p.n_chunks, (const unsigned long long *)off,
                           p.bits2, (unsigned long long *)nullptr);
    }
    // A1: stream -> 2^bits1 regions (region d1 = final buckets [d1 << bits2, (d1+1) << bits2)); one workgroup per chunk
    hipLaunchKernelGGL(scatter_stream_kernel, dim3((unsigned)p.n_chunks), dim3(BLOCK), 0, s, codes, valid, word_begin, word_end, t->k, p.bits1,
                       with_rows ? rows->row_start : (const int64_t *)nullptr, with_rows ? rows->row_end : (const int64_t *)nullptr,
                       with_rows ? rows->n_rows : (int64_t)0, with_rows ? rows->strict_valid : (const uint32_t *)nullptr,
                       (const int32_t *)tile_row, bufa, (const unsigned long long *)chunk_tab, p.n_chunks,
                       chunk_stride);
    // A2: every region -> its 2^bits2 final buckets
    if (p.bits2) {
        const int tiles_x = 96;
        hipLaunchKernelGGL((scatter_records_kernel<uint64_t, DIG_HASH, RPL64, MAX_FAN_BITS>), dim3((unsigned)(tiles_x << p.bits1)), dim3(BLOCK), 0, s,
                           (const uint64_t *)bufa, (const unsigned long long *)off, (const unsigned long long *)(off + ((size_t)1 << p.bits2)),
                           p.bits2, (const unsigned long long *)nullptr, tiles_x, bufb, (const unsigned long long *)off, cur2, 0, 0,
                           p.bits2, KEY_BITS - p.bits);
    }
    if (deferred_group >= 0) {
        // B, deferred: groups of final buckets counted in LDS, occupied entries + fills only (bufa is free after A2)
        const bool split32 = KEY_BITS - (p.bits - deferred_group) <= 31 && !getenv("PG_B64");
        const void *fn = split32 ? (const void *)bucket_count_compact32_kernel : (const void *)bucket_count_compact_kernel;
        if ((rc = raise_lds_limit(fn, slice_lds, "pg_kmer_count_deferred"))) return rc;
        hipLaunchKernelGGL(split32 ? bucket_count_compact32_kernel : bucket_count_compact_kernel, dim3((unsigned)(nb >> deferred_group)),
                           dim3(BIG_BLOCK), slice_lds, s, (const uint64_t *)bufb,
                           (const unsigned long long *)off, view_of(t), deferred_group, bufa, (long long *)fill, status);
        return check_launch("pg_kmer_count_deferred");
    }
    if (emit) {
        // B + S1 fused: counts, slice image and the (row, bin) words of the abundance rows in one pass over the buckets
        if (KEY_BITS - p.bits > 31) return pg_fail(PG_EINVAL, "pg_kmer_count_bucketed_emit: needs at least 2^11 buckets");
        ShufflePlan sp;
        if ((rc = plan_shuffle(p.cap, rows->n_rows, emit->vsize, (int64_t)nb, &sp))) return rc;
        if ((int64_t)sp.total > emit->shuffle_ws_bytes || (reinterpret_cast<uintptr_t>(emit->shuffle_ws) & 255) != 0)
            return pg_fail(PG_EINVAL, "pg_kmer_count_bucketed_emit: shuffle workspace of %lld bytes (256-byte aligned), %lld needed",
                           (long long)emit->shuffle_ws_bytes, (long long)sp.total);
        char *sws = (char *)emit->shuffle_ws;
        if ((rc = raise_lds_limit((const void *)bucket_count_emit32_kernel, slice_lds, "pg_kmer_count_bucketed_emit"))) return rc;
        hipLaunchKernelGGL(bucket_count_emit32_kernel, dim3(nb), dim3(BIG_BLOCK), slice_lds, s, (const uint64_t *)(p.bits2 ? bufb : bufa),
                           (const unsigned long long *)off, view_of(t), (uint32_t)emit->window, (uint32_t)emit->vsize, sp.vbits,
                           (uint32_t *)(sws + sp.words_e_off), (unsigned long long *)(sws + sp.emit_off), status);
        return check_launch("pg_kmer_count_bucketed_emit");
    }
    // B: count every bucket inside LDS and write its slice of the table
    if (KEY_BITS - p.bits <= 31 && !getenv("PG_B64")) {
        if ((rc = raise_lds_limit((const void *)bucket_count32_kernel, slice_lds, "pg_kmer_count_bucketed"))) return rc;
        hipLaunchKernelGGL(bucket_count32_kernel, dim3(nb), dim3(BIG_BLOCK), slice_lds, s, (const uint64_t *)(p.bits2 ? bufb : bufa),
                           (const unsigned long long *)off, view_of(t), accumulate ? 1 : 0, status);
    } else
    hipLaunchKernelGGL(bucket_count_kernel, dim3(nb), dim3(BIG_BLOCK), slice_lds, s, (const uint64_t *)(p.bits2 ? bufb : bufa),
                       (const unsigned long long *)off, view_of(t), accumulate ? 1 : 0, status);
    return check_launch("pg_kmer_count_bucketed");
}
}  // namespace

extern "C" int pg_kmer_merge(const uint64_t *pairs, int64_t n, const pg_table *t, uint32_t *status, void *stream)
{
    int rc = check_table(t);
    if (rc) return rc;
    if (t->kind != PG_TABLE_HASH && t->kind != PG_TABLE_MINI)
        return pg_fail(PG_EINVAL, "pg_kmer_merge: hash and mini tables only (dense tables are summed with an all-reduce)");
    if (n < 0 || (n > 0 && !pairs) || !status) return pg_fail(PG_EINVAL, "pg_kmer_merge: bad arguments");
    if (n == 0) return PG_OK;
    hipLaunchKernelGGL(kmer_merge_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, (hipStream_t)stream, pairs, n, view_of(t), status,
                       t->kind == PG_TABLE_MINI ? t->k : 0);
    return check_launch("pg_kmer_merge");
}

extern "C" int pg_kmer_merge_wide(const uint64_t *codes, const uint32_t *counts, int64_t n, const pg_table *t, uint32_t *status, void *stream)
{
    int rc = check_table(t);
    if (rc) return rc;
    if (t->kind != PG_TABLE_WIDE && t->kind != PG_TABLE_MINI_WIDE) return pg_fail(PG_EINVAL, "pg_kmer_merge_wide: wide and wide mini tables only");
    if (n < 0 || (n > 0 && (!codes || !counts)) || !status) return pg_fail(PG_EINVAL, "pg_kmer_merge_wide: bad arguments");
    if (n == 0) return PG_OK;
    hipLaunchKernelGGL(wide_merge_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, (hipStream_t)stream, codes, counts, n, view_of(t), status,
                       t->kind == PG_TABLE_MINI_WIDE ? t->k : 0);
    return check_launch("pg_kmer_merge_wide");
}

extern "C" int pg_kmer_merge_bucketed(const uint64_t *pairs, const int64_t *seg, int n_parts, const pg_table *t, uint32_t *status, void *stream)
{
    int rc = check_table(t);
    if (rc) return rc;
    if (t->kind != PG_TABLE_HASH || t->log2_bucket_slots == 0 || t->log2_bucket_slots > PG_BUCKET_MAX_LOG2_SLOTS)
        return pg_fail(PG_EINVAL, "pg_kmer_merge_bucketed: needs a bucketed hash table with LDS-sized buckets");
    if (n_parts < 0 || (n_parts > 0 && (!pairs || !seg)) || !status) return pg_fail(PG_EINVAL, "pg_kmer_merge_bucketed: bad arguments");
    if (n_parts == 0) return PG_OK;
    const int bits = t->log2_slots - t->log2_bucket_slots;
    if (bits < 0 || bits > 30) return pg_fail(PG_EINVAL, "pg_kmer_merge_bucketed: bad bucket geometry");
    const size_t lds = (size_t)8 << t->log2_bucket_slots;
    const bool split32 = KEY_BITS - bits <= 31 && !getenv("PG_B64");
    if ((rc = raise_lds_limit(split32 ? (const void *)bucket_merge32_kernel : (const void *)bucket_merge_kernel, lds, "pg_kmer_merge_bucketed"))) return rc;
    hipLaunchKernelGGL(split32 ? bucket_merge32_kernel : bucket_merge_kernel, dim3(1u << bits), dim3(BIG_BLOCK), lds, (hipStream_t)stream, pairs, (const long long *)seg,
                       n_parts, view_of(t), status, 0, (int64_t)0, (int64_t)1 << bits);
    return check_launch("pg_kmer_merge_bucketed");
}

extern "C" int pg_kmer_rebuild_bucketed_range(const uint64_t *pairs, const int64_t *seg, int n_parts, const pg_table *t,
                                              int64_t bucket_begin, int64_t bucket_end, uint32_t *status, void *stream)
{
    int rc = check_table(t);
    if (rc) return rc;
    if (t->kind != PG_TABLE_HASH || t->log2_bucket_slots == 0 || t->log2_bucket_slots > PG_BUCKET_MAX_LOG2_SLOTS)
        return pg_fail(PG_EINVAL, "pg_kmer_rebuild_bucketed: needs a bucketed hash table with LDS-sized buckets");
    if (n_parts < 1 || !pairs || !seg || !status) return pg_fail(PG_EINVAL, "pg_kmer_rebuild_bucketed: bad arguments");
    const int bits = t->log2_slots - t->log2_bucket_slots;
    if (bits < 0 || bits > 30) return pg_fail(PG_EINVAL, "pg_kmer_rebuild_bucketed: bad bucket geometry");
    if (bucket_begin < 0 || bucket_end < bucket_begin || bucket_end > ((int64_t)1 << bits))
        return pg_fail(PG_EINVAL, "pg_kmer_rebuild_bucketed: bucket range [%lld,%lld) outside the table", (long long)bucket_begin, (long long)bucket_end);
    if (bucket_end == bucket_begin) return PG_OK;
    const size_t lds = (size_t)8 << t->log2_bucket_slots;
    const bool split32 = KEY_BITS - bits <= 31 && !getenv("PG_B64");
    if ((rc = raise_lds_limit(split32 ? (const void *)bucket_merge32_kernel : (const void *)bucket_merge_kernel, lds, "pg_kmer_rebuild_bucketed"))) return rc;
    hipLaunchKernelGGL(split32 ? bucket_merge32_kernel : bucket_merge_kernel, dim3((unsigned)(bucket_end - bucket_begin)), dim3(BIG_BLOCK), lds,
                       (hipStream_t)stream, pairs,
                       (const long long *)seg, n_parts, view_of(t), status, 1, bucket_begin, bucket_end - bucket_begin);
    return check_launch("pg_kmer_rebuild_bucketed");
}

extern "C" int pg_kmer_rebuild_bucketed(const uint64_t *pairs, const int64_t *seg, int n_parts, const pg_table *t, uint32_t *status, void *stream)
{
    int rc = check_table(t);
    if (rc) return rc;
    return pg_kmer_rebuild_bucketed_range(pairs, seg, n_parts, t, 0, (int64_t)1 << (t->log2_slots - t->log2_bucket_slots), status, stream);
}

static int check_bucketed(const pg_table *t, const char *who)
{
    int rc = check_table(t);
    if (rc) return rc;
    if (t->kind != PG_TABLE_HASH || t->log2_bucket_slots < 1) return pg_fail(PG_EINVAL, "%s: needs a bucketed hash table", who);
    if (t->log2_slots - t->log2_bucket_slots > 30) return pg_fail(PG_EINVAL, "%s: too many buckets for one launch", who);
    return PG_OK;
}

extern "C" int pg_table_bucket_fill(const pg_table *t, int64_t *fill, void *stream)
{
    int rc = check_bucketed(t, "pg_table_bucket_fill");
    if (rc) return rc;
    if (!fill) return pg_fail(PG_EINVAL, "pg_table_bucket_fill: null output");
    hipLaunchKernelGGL(bucket_fill_kernel, dim3(1u << (t->log2_slots - t->log2_bucket_slots)), dim3(BLOCK), 0, (hipStream_t)stream,
                       (const uint64_t *)t->data, t->log2_bucket_slots, (long long *)fill);
    return check_launch("pg_table_bucket_fill");
}

extern "C" int pg_table_bucket_fill_range(const pg_table *t, int64_t bucket_begin, int64_t bucket_end, int64_t *fill, void *stream)
{
    int rc = check_bucketed(t, "pg_table_bucket_fill_range");
    if (rc) return rc;
    const int64_t nb = (int64_t)1 << (t->log2_slots - t->log2_bucket_slots);
    if (!fill || bucket_begin < 0 || bucket_end < bucket_begin || bucket_end > nb) return pg_fail(PG_EINVAL, "pg_table_bucket_fill_range: bad arguments");
    if (bucket_end == bucket_begin) return PG_OK;
    hipLaunchKernelGGL(bucket_fill_kernel, dim3((unsigned)(bucket_end - bucket_begin)), dim3(BLOCK), 0, (hipStream_t)stream,
                       (const uint64_t *)t->data + ((uint64_t)bucket_begin << t->log2_bucket_slots), t->log2_bucket_slots, (long long *)fill);
    return check_launch("pg_table_bucket_fill_range");
}

extern "C" int pg_table_compact_planes_range(const pg_table *t, int64_t bucket_begin, int64_t bucket_end, const int64_t *tag_elem,
                                             const int64_t *cnt_elem, void *out, uint64_t *overflow, uint64_t *overflow_count,
                                             int64_t overflow_cap, uint32_t *status, void *stream)
{
    int rc = check_bucketed(t, "pg_table_compact_planes_range");
    if (rc) return rc;
    const int bits = t->log2_slots - t->log2_bucket_slots;
    if (KEY_BITS - bits > 31) return pg_fail(PG_EINVAL, "pg_table_compact_planes_range: needs at least 2^11 buckets (tags of at most 31 bits)");
    if (!tag_elem || !cnt_elem || !out || !overflow || !overflow_count || !status || bucket_begin < 0 || bucket_end < bucket_begin ||
        bucket_end > ((int64_t)1 << bits))
        return pg_fail(PG_EINVAL, "pg_table_compact_planes_range: bad arguments");
    if (bucket_end == bucket_begin) return PG_OK;
    hipLaunchKernelGGL(table_compact_planes_kernel, dim3((unsigned)(bucket_end - bucket_begin)), dim3(BLOCK), 0, (hipStream_t)stream,
                       (const uint64_t *)t->data + ((uint64_t)bucket_begin << t->log2_bucket_slots), t->log2_bucket_slots,
                       (const long long *)tag_elem, (const long long *)cnt_elem, (uint32_t)((1u << (KEY_BITS - bits)) - 1u), (uint32_t *)out,
                       (uint16_t *)out, overflow, (unsigned long long *)overflow_count, (unsigned long long)overflow_cap, status);
    return check_launch("pg_table_compact_planes_range");
}

extern "C" int pg_table_compact(const pg_table *t, const int64_t *seg, uint64_t *out, void *stream)
{
    int rc = check_bucketed(t, "pg_table_compact");
    if (rc) return rc;
    if (!seg || !out) return pg_fail(PG_EINVAL, "pg_table_compact: null argument");
    hipLaunchKernelGGL(table_compact_kernel, dim3(1u << (t->log2_slots - t->log2_bucket_slots)), dim3(BLOCK), 0, (hipStream_t)stream,
                       (const uint64_t *)t->data, t->log2_bucket_slots, (const long long *)seg, out);
    return check_launch("pg_table_compact");
}

extern "C" int64_t pg_abundance_workspace_bytes(int64_t n_words_counted, int64_t n_rows, int vsize, const pg_table *t)
{
    if (n_words_counted < 0 || n_rows < 0) return pg_fail(PG_EINVAL, "negative size");
    int rc = check_table(t);
    if (rc) return rc;
    BucketPlan p;
    if ((rc = plan_buckets(t, n_words_counted, &p))) return rc;
    ShufflePlan sp;
    if ((rc = plan_shuffle(p.cap, n_rows, vsize, (int64_t)1 << p.bits, &sp))) return rc;
    return (int64_t)sp.total;
}

static int abundance_impl(const pg_table *t, const pg_rows *rows, int window, int vsize, int32_t *abd_out,
                          const void *count_workspace, int64_t count_workspace_bytes, int64_t n_words_counted,
                          void *workspace, int64_t workspace_bytes, void *stream, bool emitted);
static int shuffle_tail(const ShufflePlan &sp, const unsigned long long *in_begin, int nb, const pg_rows *rows, int vsize,
                        int32_t *abd_out, char *ws, hipStream_t s);

// for the other translation units (pg_internal.h): where the lookup pass of another pipeline leaves its words, and the
// row shuffle that finishes them
int pg_internal_shuffle_layout(int64_t cap, int64_t n_rows, int vsize, pg_shuffle_layout *out, int one_pass_bits, int no_input)
{
    ShufflePlan sp;
    int rc = plan_shuffle(cap, n_rows, vsize, 0, &sp, one_pass_bits, no_input);
    if (rc) return rc;
    out->vbits = sp.vbits;
    out->emit_off = sp.emit_off;
    out->words_e_off = sp.words_e_off;
    out->words_a_off = sp.words_a_off;
    out->total = sp.total;
    return PG_OK;
}

int pg_internal_shuffle_rows(const unsigned long long *in_begin, int nb, int64_t cap, const pg_rows *rows, int vsize, int32_t *abd_out,
                             void *workspace, int64_t workspace_bytes, void *stream)
{
    ShufflePlan sp;
    int rc = plan_shuffle(cap, rows->n_rows, vsize, 0, &sp);
    if (rc) return rc;
    if ((int64_t)sp.total > workspace_bytes) return pg_fail(PG_EINVAL, "row shuffle: workspace of %lld bytes, %lld needed", (long long)workspace_bytes, (long long)sp.total);
    if (rows->n_rows == 0) return PG_OK;
    return shuffle_tail(sp, in_begin, nb, rows, vsize, abd_out, (char *)workspace, (hipStream_t)stream);
}

extern "C" int pg_abundance_from_records(const pg_table *t, const pg_rows *rows, int window, int vsize, int32_t *abd_out,
                                         const void *count_workspace, int64_t count_workspace_bytes, int64_t n_words_counted,
                                         void *workspace, int64_t workspace_bytes, void *stream)
{
    return abundance_impl(t, rows, window, vsize, abd_out, count_workspace, count_workspace_bytes, n_words_counted, workspace, workspace_bytes,
                          stream, false);
}

extern "C" int pg_abundance_from_emitted(const pg_table *t, const pg_rows *rows, int window, int vsize, int32_t *abd_out,
                                         const void *count_workspace, int64_t count_workspace_bytes, int64_t n_words_counted,
                                         void *workspace, int64_t workspace_bytes, void *stream)
{
    return abundance_impl(t, rows, window, vsize, abd_out, count_workspace, count_workspace_bytes, n_words_counted, workspace, workspace_bytes,
                          stream, true);
}

static int abundance_impl(const pg_table *t, const pg_rows *rows, int window, int vsize, int32_t *abd_out,
                          const void *count_workspace, int64_t count_workspace_bytes, int64_t n_words_counted,
                          void *workspace, int64_t workspace_bytes, void *stream, bool emitted)
{
    int rc = check_table(t);
    if (rc) return rc;
    if (!rows || !abd_out || !count_workspace || !workspace) return pg_fail(PG_EINVAL, "pg_abundance_from_records: null argument");
    if ((rc = check_rows(rows, "pg_abundance_from_records"))) return rc;
    if (window < 1) return pg_fail(PG_EINVAL, "pg_abundance_from_records: window %d", window);
    if ((int64_t)window * vsize > (int64_t)PG_HASH_COUNT_SAT)
        return pg_fail(PG_EINVAL, "pg_abundance_from_records: window*vector_size %lld exceeds the exact range of the hash table (%u)",
                       (long long)window * vsize, PG_HASH_COUNT_SAT);
    BucketPlan p;
    if ((rc = plan_buckets(t, n_words_counted, &p))) return rc;
    if ((int64_t)p.total > count_workspace_bytes) return pg_fail(PG_EINVAL, "pg_abundance_from_records: count workspace does not match n_words_counted");
    ShufflePlan sp;
    if ((rc = plan_shuffle(p.cap, rows->n_rows, vsize, (int64_t)1 << p.bits, &sp))) return rc;
    if ((int64_t)sp.total > workspace_bytes)
        return pg_fail(PG_EINVAL, "pg_abundance_from_records: workspace of %lld bytes, %lld needed", (long long)workspace_bytes, (long long)sp.total);
    if ((reinterpret_cast<uintptr_t>(workspace) & 255) != 0) return pg_fail(PG_EINVAL, "pg_abundance_from_records: workspace must be 256-byte aligned");
    if (rows->n_rows == 0) return PG_OK;
    hipStream_t s = (hipStream_t)stream;
    const char *cws = (const char *)count_workspace;
    const auto *off = (const unsigned long long *)(cws + p.off_off);
    const auto *recs = (const uint64_t *)(cws + p.final_off());
    char *ws = (char *)workspace;
    auto *emit_end = (unsigned long long *)(ws + sp.emit_off);
    auto *words_e = (uint32_t *)(ws + sp.words_e_off);
    const int nb = 1 << p.bits;
    const size_t slice_lds = (size_t)8 << t->log2_bucket_slots;
    if ((rc = raise_lds_limit((const void *)bucket_lookup_kernel, slice_lds, "pg_abundance_from_records"))) return rc;
    // (emitted: the lookup pass already ran inside pg_kmer_count_bucketed_emit and left emit_end + the words behind)
    if (!emitted && hipMemsetAsync(ws, 0, sp.caps_off, s) != hipSuccess)
        return pg_fail(PG_EHIP, "pg_abundance_from_records: memset failed");
    // S1: counts out of the LDS copies of the slices -> (row, bin) words, packed per bucket
    if (emitted) {
    } else if (KEY_BITS - p.bits <= 31 && !getenv("PG_B64")) {
        if ((rc = raise_lds_limit((const void *)bucket_lookup32_kernel, slice_lds, "pg_abundance_from_records"))) return rc;
        hipLaunchKernelGGL(bucket_lookup32_kernel, dim3(nb), dim3(BIG_BLOCK), slice_lds, s, recs, off, view_of(t), (uint32_t)window, (uint32_t)vsize,
                           sp.vbits, words_e, emit_end);
    } else
    hipLaunchKernelGGL(bucket_lookup_kernel, dim3(nb), dim3(BIG_BLOCK), slice_lds, s, recs, off, view_of(t), (uint32_t)window, (uint32_t)vsize, sp.vbits,
                       words_e, emit_end);
    return shuffle_tail(sp, off, nb, rows, vsize, abd_out, ws, s);
}

// S2 + S3: the (row, bin) words of every bucket b, words_e[in_begin[b] .. emit_end[b]), -> rows of the abundance matrix.
// Three parts so that a lookup pass which scatters its words by row group itself (mini.hip) can take the first pass's place:
//   prepare   row-group capacities -> offsets of the group regions, cursors cleared
//   pass one  scatter by the first gb1 bits of the row group
//   finish    the second pass (more than 2^10 groups only) and the LDS row histograms
static int shuffle_prepare(const ShufflePlan &sp, const pg_rows *rows, char *ws, hipStream_t s)
{
    auto *caps = (unsigned long long *)(ws + sp.caps_off);
    auto *goff = (unsigned long long *)(ws + sp.goff_off);
    if (hipMemsetAsync(ws + sp.caps_off, 0, sp.dhist_off - sp.caps_off, s) != hipSuccess)
        return pg_fail(PG_EHIP, "pg_abundance_from_records: memset failed");
    hipLaunchKernelGGL(group_caps_kernel, dim3((unsigned)((sp.n_groups_padded + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s, rows->row_start, rows->row_end,
                       rows->n_rows, sp.n_groups_padded, caps);
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(BIG_BLOCK), 0, s, (const unsigned long long *)caps, sp.n_groups_padded, goff);
    return check_launch("pg_abundance_from_records");
}

static int shuffle_finish(const ShufflePlan &sp, const pg_rows *rows, int vsize, int32_t *abd_out, char *ws, hipStream_t s, int word_form = PG_SHUFFLE_WORDS_PLAIN)
{
    const bool narrow = word_form == PG_SHUFFLE_WORDS_NARROW, counted = word_form == PG_SHUFFLE_WORDS_COUNTED;
    if (counted && sp.vbits + GROUP_ROWS_LOG2 + sp.gbits > PG_SHUFFLE_COUNT_SHIFT) return pg_fail(PG_EINVAL, "row shuffle: counted words need row bits + bin bits <= %d", PG_SHUFFLE_COUNT_SHIFT);
    int rc;
    auto *goff = (unsigned long long *)(ws + sp.goff_off);
    auto *gcur1 = (unsigned long long *)(ws + sp.gcur1_off);
    auto *gcur2 = (unsigned long long *)(ws + sp.gcur2_off);
    auto *words_a = (uint32_t *)(ws + sp.words_a_off);
    auto *words_b = (uint32_t *)(ws + sp.words_b_off);
    const size_t hist_lds = ((size_t)4 << GROUP_ROWS_LOG2) * (size_t)(vsize | 1);
    if ((rc = raise_lds_limit((const void *)row_hist_kernel<uint32_t>, hist_lds, "pg_abundance_from_records"))) return rc;
    if ((rc = raise_lds_limit((const void *)row_hist_kernel<uint16_t>, hist_lds, "pg_abundance_from_records"))) return rc;
    if ((rc = raise_lds_limit((const void *)row_hist_kernel<uint32_t, true>, hist_lds, "pg_abundance_from_records"))) return rc;
    if (narrow && sp.gb2) return pg_fail(PG_EINVAL, "row shuffle: narrow words need a one-pass shuffle");
    const int gshift = sp.vbits + GROUP_ROWS_LOG2;
    const unsigned long long *gcnt = gcur1;
    const uint32_t *final_words = words_a;
    if (sp.gb2) {
        const int tiles_x = 64;
        hipLaunchKernelGGL((scatter_records_kernel<uint32_t, DIG_ROW, RPL32, MAX_FAN_BITS>), dim3((unsigned)(tiles_x << sp.gb1)), dim3(BLOCK), 0, s,
                           (const uint32_t *)words_a, (const unsigned long long *)goff, (const unsigned long long *)nullptr, sp.gb2,
                           (const unsigned long long *)gcur1, tiles_x, words_b, (const unsigned long long *)goff, gcur2, 0, 0, sp.gb2, gshift);
        gcnt = gcur2;
        final_words = words_b;
    }
    // S3: LDS row histograms -> rows of the matrix.  One workgroup per group and CU: the groups of a last round that would leave most
    // CUs idle (at most a quarter of them busy) are shared out instead, `split` workgroups each (PG_ROW_HIST_SPLIT="groups,split"
    // forces a split, "0" none)
    static int n_cus = 0;
    if (!n_cus) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cus <= 0) n_cus = 256;
    }
    int64_t n_tail = sp.n_groups > n_cus ? sp.n_groups % n_cus : 0;
    int split = n_tail > 0 && n_tail <= n_cus / 4 ? (int)std::min<int64_t>(32, n_cus / n_tail) : 1;
    if (const char *force = getenv("PG_ROW_HIST_SPLIT")) {
        long long fg = 0, fs = 1;
        if (sscanf(force, "%lld,%lld", &fg, &fs) == 2 && fg > 0 && fs > 1) { n_tail = std::min<int64_t>(fg, sp.n_groups); split = (int)std::min<long long>(fs, 64); }
        else { n_tail = 0; split = 1; }
    }
    if (split <= 1) n_tail = 0;
    const int64_t n_main = sp.n_groups - n_tail;
    if (n_tail) {
        const int64_t row0 = n_main << GROUP_ROWS_LOG2;
        if (row0 < rows->n_rows && hipMemsetAsync(abd_out + row0 * (int64_t)vsize, 0, (size_t)(rows->n_rows - row0) * vsize * sizeof(int32_t), s) != hipSuccess)
            return pg_fail(PG_EHIP, "pg_abundance_from_records: memset failed");
    }
#define PG_ROW_HIST(WORD_, COUNTED_, WORDS_)                                                                                 \
    do {                                                                                                                    \
        if (n_main) hipLaunchKernelGGL((row_hist_kernel<WORD_, COUNTED_>), dim3((unsigned)n_main), dim3(BIG_BLOCK), hist_lds, s, WORDS_, (const unsigned long long *)goff, \
                                       gcnt, sp.vbits, (uint32_t)vsize, rows->n_rows, abd_out, (int64_t)0, 1);               \
        if (n_tail) hipLaunchKernelGGL((row_hist_kernel<WORD_, COUNTED_>), dim3((unsigned)(n_tail * split)), dim3(BIG_BLOCK), hist_lds, s, WORDS_,           \
                                       (const unsigned long long *)goff, gcnt, sp.vbits, (uint32_t)vsize, rows->n_rows, abd_out, n_main, split); \
    } while (0)
    if (narrow) PG_ROW_HIST(uint16_t, false, (const uint16_t *)final_words);
    else if (counted) PG_ROW_HIST(uint32_t, true, final_words);
    else PG_ROW_HIST(uint32_t, false, final_words);
#undef PG_ROW_HIST
    return check_launch("pg_abundance_from_records");
}

static int shuffle_tail(const ShufflePlan &sp, const unsigned long long *in_begin, int nb, const pg_rows *rows, int vsize,
                        int32_t *abd_out, char *ws, hipStream_t s)
{
    int rc = shuffle_prepare(sp, rows, ws, s);
    if (rc) return rc;
    auto *emit_end = (unsigned long long *)(ws + sp.emit_off);
    auto *goff = (unsigned long long *)(ws + sp.goff_off);
    auto *gcur1 = (unsigned long long *)(ws + sp.gcur1_off);
    auto *words_e = (uint32_t *)(ws + sp.words_e_off);
    auto *words_a = (uint32_t *)(ws + sp.words_a_off);
    const unsigned long long *off = in_begin;
    const int gshift = sp.vbits + GROUP_ROWS_LOG2;
    // S2a: scatter the words by the first gb1 bits of their row group (destinations from global cursors, one add per digit
    // and tile)
    {
        // ONE workgroup per bucket walks the bucket's tiles (about five wide ones): measured 10.5 ms against 12.2 / 13.8 / 14.9 ms with
        // 2 / 4 / 8 workgroups per bucket -- every workgroup first waits for the two dependent loads of its region bounds
        const int tiles_x = 1;
        if (sp.gb1 > MAX_FAN_BITS)         // all row groups in ONE pass: bigger tiles keep the runs per group at 64 B
            hipLaunchKernelGGL((scatter_records_kernel<uint32_t, DIG_ROW, RPL32_WIDE, WIDE_FAN_BITS>), dim3((unsigned)(tiles_x * nb)), dim3(BLOCK), 0, s,
                               (const uint32_t *)words_e, off, (const unsigned long long *)emit_end, 0, (const unsigned long long *)nullptr, tiles_x,
                               words_a, (const unsigned long long *)goff, gcur1, sp.gb2, 1, sp.gb1, gshift + sp.gb2);
        else
        hipLaunchKernelGGL((scatter_records_kernel<uint32_t, DIG_ROW, RPL32, MAX_FAN_BITS>), dim3((unsigned)(tiles_x * nb)), dim3(BLOCK), 0, s,
                           (const uint32_t *)words_e, off, (const unsigned long long *)emit_end, 0, (const unsigned long long *)nullptr, tiles_x,
                           words_a, (const unsigned long long *)goff, gcur1, sp.gb2, 1, sp.gb1, gshift + sp.gb2);
    }
    return shuffle_finish(sp, rows, vsize, abd_out, ws, s);
}

// the two outer parts for a lookup pass that does the first scatter itself: `ctx` tells it where the group regions are
int pg_internal_shuffle_is_narrow(int64_t cap, int64_t n_rows, int vsize, int one_pass_bits)
{
    ShufflePlan sp;
    if (plan_shuffle(cap, n_rows, vsize, 0, &sp, one_pass_bits)) return 0;
    return sp.gb2 == 0 && sp.vbits + GROUP_ROWS_LOG2 <= 15 && !getenv("PG_WIDE_WORDS");
}

int pg_internal_shuffle_prepare(int64_t cap, const pg_rows *rows, int vsize, void *workspace, int64_t workspace_bytes, void *stream,
                                pg_shuffle_ctx *ctx, int one_pass_bits, int no_input, int ctx_only)
{
    ShufflePlan sp;
    int rc = plan_shuffle(cap, rows->n_rows, vsize, 0, &sp, one_pass_bits, no_input);
    if (rc) return rc;
    if ((int64_t)sp.total > workspace_bytes) return pg_fail(PG_EINVAL, "row shuffle: workspace of %lld bytes, %lld needed", (long long)workspace_bytes, (long long)sp.total);
    char *ws = (char *)workspace;
    ctx->goff = (const unsigned long long *)(ws + sp.goff_off);
    ctx->gcur1 = (unsigned long long *)(ws + sp.gcur1_off);
    ctx->words_in = (uint32_t *)(ws + sp.words_e_off);
    ctx->words_out = (uint32_t *)(ws + sp.words_a_off);
    ctx->vbits = sp.vbits;
    ctx->gb1 = sp.gb1;
    ctx->gb2 = sp.gb2;
    ctx->dshift = sp.vbits + GROUP_ROWS_LOG2 + sp.gb2;
    ctx->narrow = pg_internal_shuffle_is_narrow(cap, rows->n_rows, vsize, one_pass_bits);
    ctx->words_cap = (unsigned long long)cap;
    if (rows->n_rows == 0 || ctx_only) return PG_OK;             // (ctx_only: offsets and cursors are in use already -- a further launch into them)
    return shuffle_prepare(sp, rows, ws, (hipStream_t)stream);
}

int pg_internal_shuffle_finish(int64_t cap, const pg_rows *rows, int vsize, int32_t *abd_out, void *workspace, int64_t workspace_bytes, void *stream,
                               int word_form, int one_pass_bits, int no_input)
{
    ShufflePlan sp;
    int rc = plan_shuffle(cap, rows->n_rows, vsize, 0, &sp, one_pass_bits, no_input);
    if (rc) return rc;
    if ((int64_t)sp.total > workspace_bytes) return pg_fail(PG_EINVAL, "row shuffle: workspace of %lld bytes, %lld needed", (long long)workspace_bytes, (long long)sp.total);
    if (rows->n_rows == 0) return PG_OK;
    return shuffle_finish(sp, rows, vsize, abd_out, (char *)workspace, (hipStream_t)stream, word_form);
}

// a9: L1 row normalisation in float64, narrowed to float32, + the sampling weight; one wavefront per row (src/data.py:16-21)
namespace {
__global__ __launch_bounds__(BLOCK) void normalize_rows_kernel(const int32_t *__restrict__ m, int64_t n_rows, int n_cols,
                                                               float *__restrict__ out, double *__restrict__ weight)
{
    const int64_t r = (int64_t)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    if (r >= n_rows) return;
    const uint32_t lane = threadIdx.x & 63;
    const int32_t *row = m + r * n_cols;
    long long sum = 0;
    int32_t top = INT32_MIN;
    for (int c = lane; c < n_cols; c += 64) {
        const int32_t x = row[c];
        sum += x < 0 ? -(long long)x : (long long)x;
        top = x > top ? x : top;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        sum += __shfl_xor(sum, d);
        const int32_t o = __shfl_xor(top, d);
        top = o > top ? o : top;
    }
    const double norm = sum == 0 ? 1.0 : (double)sum;                // (sklearn: a zero norm counts as 1)
    float *dst = out + r * n_cols;
    for (int c = lane; c < n_cols; c += 64) dst[c] = (float)((double)row[c] / norm);
    if (weight && lane == 0) {
        const double q = n_cols > 0 ? (double)top / norm : 0.0;
        weight[r] = q * q;
    }
}
}  // namespace

extern "C" int pg_normalize_rows(const int32_t *m, int64_t n_rows, int n_cols, float *out, double *weight, void *stream)
{
    if (n_rows < 0 || n_cols < 0) return pg_fail(PG_EINVAL, "pg_normalize_rows: negative size");
    if (n_rows == 0 || n_cols == 0) return PG_OK;
    if (!m || !out) return pg_fail(PG_EINVAL, "pg_normalize_rows: null argument");
    const int64_t per = BLOCK / 64;
    const int64_t grid = (n_rows + per - 1) / per;
    if (grid > 0x7fffffffLL) return pg_fail(PG_EINVAL, "pg_normalize_rows: too many rows for one launch");
    hipLaunchKernelGGL(normalize_rows_kernel, dim3((unsigned)grid), dim3(BLOCK), 0, (hipStream_t)stream, m, n_rows, n_cols, out, weight);
    return check_launch("pg_normalize_rows");
}

extern "C" int pg_features(const uint64_t *codes, const uint32_t *valid, int64_t n_words,
                           const int32_t *seg_row, const int64_t *seg_start, const int64_t *seg_end, int64_t n_segs,
                           int k_tnf, const uint16_t *colmap, int32_t *tnf_out,
                           const pg_table *t, int window, int vsize, int32_t *abd_out, void *stream)
{
    if (!codes || !valid || n_words < 0) return pg_fail(PG_EINVAL, "pg_features: null stream arrays");
    if (n_segs < 0 || (n_segs > 0 && (!seg_row || !seg_start || !seg_end))) return pg_fail(PG_EINVAL, "pg_features: bad segment arrays");
    if (n_segs > 0x7fffffffLL) return pg_fail(PG_EINVAL, "pg_features: too many segments for one launch");
    const bool do_tnf = tnf_out != nullptr;
    const bool do_abd = abd_out != nullptr;
    if (!do_tnf && !do_abd) return pg_fail(PG_EINVAL, "pg_features: no output requested");
    int tnf_cols = 0, tnf_copies = 0, n_raw = 0;
    if (do_tnf) {
        if (k_tnf < 1 || k_tnf > PG_TNF_MAX_K) return pg_fail(PG_EINVAL, "pg_features: tnf k must be in [1,%d] (got %d)", PG_TNF_MAX_K, k_tnf);
        if (!colmap) return pg_fail(PG_EINVAL, "pg_features: colmap is null");
        tnf_cols = pg_tnf_ncols(k_tnf);
        n_raw = 1 << (2 * k_tnf);
        tnf_copies = k_tnf <= 4 ? WAVES : 1;
    }
    int kind = TK_NONE, k = 0;
    HashView view{nullptr, 0, 0};
    const uint32_t *dense = nullptr;
    if (do_abd) {
        int rc = check_table(t);
        if (rc) return rc;
        if (window < 1 || vsize < 1 || vsize > 8192) return pg_fail(PG_EINVAL, "pg_features: window %d / vector size %d out of range", window, vsize);
        if ((t->kind == PG_TABLE_HASH || t->kind == PG_TABLE_MINI) && (int64_t)window * vsize > (int64_t)PG_HASH_COUNT_SAT)
            return pg_fail(PG_EINVAL, "pg_features: window*vector_size %lld exceeds the exact range of the hash table (%u)",
                           (long long)window * vsize, PG_HASH_COUNT_SAT);
        kind = t->kind == PG_TABLE_DENSE ? TK_DENSE : t->kind == PG_TABLE_WIDE ? TK_WIDE : t->kind == PG_TABLE_MINI ? TK_MINI
             : t->kind == PG_TABLE_MINI_WIDE ? TK_MINIW : TK_HASH;
        k = t->k;
        if (kind == TK_DENSE) dense = (const uint32_t *)t->data;
        else view = view_of(t);
    }
    if (n_segs == 0) return PG_OK;
    const int abd_copies = do_abd ? (vsize <= 1024 ? WAVES : 1) : 0;
    const size_t lds_bytes = sizeof(uint32_t) * ((size_t)abd_copies * (do_abd ? vsize : 0) + (size_t)tnf_copies * n_raw);
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)n_segs), block(BLOCK);
#define PG_LAUNCH(KT, TK)                                                                                                   \
    hipLaunchKernelGGL((features_kernel<KT, TK>), grid, block, lds_bytes, s, codes, valid, n_words, seg_row, seg_start, seg_end, \
                       k_tnf, colmap, tnf_cols, tnf_copies, abd_copies, tnf_out, k, dense, view, (uint32_t)window, vsize, abd_out)
    if (kind == TK_NONE) PG_LAUNCH(uint32_t, TK_NONE);
    else if (kind == TK_DENSE) PG_LAUNCH(uint32_t, TK_DENSE);
    else if (kind == TK_WIDE) PG_LAUNCH(uint64_t, TK_WIDE);
    else if (kind == TK_MINI) PG_LAUNCH(uint64_t, TK_MINI);
    else if (kind == TK_MINIW) PG_LAUNCH(uint64_t, TK_MINIW);
    else PG_LAUNCH(uint64_t, TK_HASH);
#undef PG_LAUNCH
    return check_launch("pg_features");
}
