// kernels.hip -- gfx950 kernels of the barcode k-mer feature path + their C-ABI launchers.
//
// Replaces (reference file:line, /root/reference/src):
//   K2  pg_kmer_count*  jellyfish count -C (feature.py:94) + dump reload (cpptools/count_kmer.cpp:139-170)
//   K1  pg_features/tnf cpptools/count_tnf.cpp:78-113  (per-run canonical k_tnf-mer counts)
//   K3  pg_features/abd cpptools/count_kmer.cpp:55-108 (per-run histogram of global multiplicities)
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
// Two ways to build the hash table:
//   direct    (kmer_count_kernel)   one random 64-B line + one memory-side atomic per k-mer occurrence;
//   bucketed  (pg_kmer_count_bucketed) occurrences are hash-partitioned with two streaming scatter passes
//             (<= 256-way then <= 512-way, LDS-staged so HBM sees whole runs), every final bucket is then
//             counted inside LDS by one workgroup and its LDS image is written back as that bucket's slice
//             of the table.  HBM only sees streaming traffic; all counting atomics are LDS atomics.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pangaea_feat.h"
#include "pg_internal.h"

namespace {

constexpr int BLOCK = 256;
constexpr int WAVES = BLOCK / 64;
constexpr uint32_t HASH_CBITS = PG_HASH_COUNT_BITS;
constexpr uint64_t HASH_CMASK = (1ull << HASH_CBITS) - 1;
constexpr uint32_t HASH_SAT = PG_HASH_COUNT_SAT;
constexpr uint32_t MAX_PROBE = 1u << 14;
constexpr uint32_t PRIMARY_PROBES = 32;   // slots tried from the minimizer home before the k-mer's own hash takes over
constexpr int MINI_W = 7;                 // m-mers per k-mer in the minimizer window: w = min(7, k), m = k - w + 1
#ifndef PG_PLACEMENT_MINIMIZER
#define PG_PLACEMENT_MINIMIZER 0
#endif
constexpr bool MINIMIZER = PG_PLACEMENT_MINIMIZER != 0;   // 0: home = uniform hash of the k-mer (no locality)

enum { TK_NONE = 0, TK_DENSE = 1, TK_HASH = 2 };

__device__ __forceinline__ uint64_t mix64(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return x;
}

__device__ __forceinline__ uint32_t mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x85ebca6bu;
    x ^= x >> 13; x *= 0xc2b2ae35u;
    x ^= x >> 16;
    return x;
}

// Home slot (log2_slots bits) of a k-mer.  MINIMIZER placement: the 64-B line (8 slots) is chosen by the k-mer's
// minimizer, the slot inside the line by the k-mer's own code, so the ~w consecutive k-mers of a read that share a
// minimizer sit in ONE line and mostly in different slots of it.
__device__ __forceinline__ uint64_t home_slot(uint32_t min_order, uint64_t code, int log2_slots)
{
    if (MINIMIZER)
        return ((((uint64_t)min_order * 0x9E3779B97F4A7C15ull) >> (64 - log2_slots)) & ~7ull) | ((code * 0x9E3779B97F4A7C15ull) >> 61);
    return mix64(code) >> (64 - log2_slots);
}

// Hash table as the kernels see it: 2^log2_slots slots in buckets of 2^log2_bucket slots.
// PLACEMENT (shared by every kernel, the LDS tables included):
//   home   = home_slot(min over the k-mer's w canonical m-mers of mix32(m-mer), code): line by minimizer, slot in
//            the line by the code.  Consecutive k-mers of a read mostly share their minimizer, hence their line:
//            a lookup stream walks the table supermer by supermer instead of touching one random line per k-mer.
//   probe  = PRIMARY_PROBES consecutive slots from home (wrapping inside the bucket), then consecutive slots from
//            (mix64(code) mod bucket) in the same bucket: crowded minimizers spill to uniformly hashed places.
//   Slots are never freed, so a lookup may stop at the first empty slot of the sequence.
struct HashView {
    uint64_t *slots;
    int log2_slots;
    int log2_bucket;
    __device__ __forceinline__ uint64_t bmask() const { return (1ull << log2_bucket) - 1; }
    __device__ __forceinline__ uint32_t primary() const { return log2_bucket < 5 ? (1u << log2_bucket) : PRIMARY_PROBES; }
    __device__ __forceinline__ uint32_t limit() const { return primary() + (log2_bucket < 14 ? (1u << log2_bucket) : MAX_PROBE); }
    // slot after `s`, `i` = number of slots already tried (i >= 1)
    __device__ __forceinline__ uint64_t next(uint64_t s, uint32_t i, uint64_t code) const
    {
        const uint64_t bm = bmask();
        const uint64_t in = i == primary() ? (mix64(code) & bm) : ((s + 1) & bm);
        return (s & ~bm) | in;
    }
};

// bit p of the result is set iff bits p-k+1..p of m are all set (1 <= k <= 32): which positions of the
// 64-character window [previous word | this word] end a run of >= k valid characters.
__device__ __forceinline__ uint64_t runs_of(uint64_t m, int k)
{
    uint64_t r = m;
    int len = 1;
    while (2 * len <= k) { r &= r << len; len *= 2; }
    if (len < k) r &= r << (k - len);
    return r;
}

template <typename KT> __device__ __forceinline__ KT low_mask(int k)
{
    return (2 * k >= (int)(8 * sizeof(KT))) ? (KT)~(KT)0 : (KT)(((KT)1 << (2 * k)) - 1);
}

// Rolling state of one lane: forward / reverse-complement code of the last k characters and, when MINI, the
// order values of the last w canonical m-mers (a shift register: no dynamic register indexing).
template <typename KT, bool MINI> struct Roller {
    KT fw, rc, kmask;
    uint32_t mmask, o0, o1, o2, o3, o4, o5, o6;
    int rc_shift, rcm_shift, w;

    __device__ __forceinline__ void init(int k)
    {
        fw = rc = 0;
        kmask = low_mask<KT>(k);
        rc_shift = 2 * (k - 1);
        w = k < MINI_W ? k : MINI_W;
        const int m = k - w + 1;
        mmask = m >= 16 ? 0xffffffffu : ((1u << (2 * m)) - 1u);
        rcm_shift = 2 * (k - m);
        o0 = o1 = o2 = o3 = o4 = o5 = o6 = 0xffffffffu;
    }
    __device__ __forceinline__ void push(uint32_t c)
    {
        fw = (KT)(fw << 2) | (KT)c;
        rc = (KT)(rc >> 2) | (KT)((KT)(c ^ 2u) << rc_shift);
        if (MINI) {
            const uint32_t a = (uint32_t)fw & mmask;
            const uint32_t b = (uint32_t)(rc >> rcm_shift);
            o6 = o5; o5 = o4; o4 = o3; o3 = o2; o2 = o1; o1 = o0;
            o0 = mix32(a < b ? a : b);
        }
    }
    __device__ __forceinline__ KT canon() const
    {
        const KT f = fw & kmask;
        return f < rc ? f : rc;
    }
    __device__ __forceinline__ uint64_t slot(int log2_slots) const { return home_slot(min_order(), (uint64_t)canon(), log2_slots); }
    __device__ __forceinline__ uint32_t min_order() const
    {
        if (!MINI) return 0;
        uint32_t mn = o0;
        if (w > 1) mn = min(mn, o1);
        if (w > 2) mn = min(mn, o2);
        if (w > 3) mn = min(mn, o3);
        if (w > 4) mn = min(mn, o4);
        if (w > 5) mn = min(mn, o5);
        if (w > 6) mn = min(mn, o6);
        return mn;
    }
};

// the same home slot from a canonical code alone (merge path; not hot)
__device__ uint64_t slot_of_code(uint64_t code, int k, int log2_slots)
{
    if (!MINIMIZER) return home_slot(0, code, log2_slots);
    const int w = k < MINI_W ? k : MINI_W;
    const int m = k - w + 1;
    const uint32_t mmask = m >= 16 ? 0xffffffffu : ((1u << (2 * m)) - 1u);
    uint64_t rck = 0, x = code;
    for (int i = 0; i < k; ++i) { rck = (rck << 2) | ((x & 3) ^ 2); x >>= 2; }
    uint32_t mn = 0xffffffffu;
    for (int i = 0; i < w; ++i) {
        const uint32_t a = (uint32_t)(code >> (2 * (w - 1 - i))) & mmask;
        const uint32_t b = (uint32_t)(rck >> (2 * i)) & mmask;
        mn = min(mn, mix32(a < b ? a : b));
    }
    return home_slot(mn, code, log2_slots);
}

// -------------------------------------------------------------------------------- table access

__device__ __forceinline__ void dense_add(uint32_t *table, uint32_t code) { atomicAdd(&table[code], 1u); }

// slot = (code << 22) | count ; 0 = empty.  Keys never change once written, so a stale (cached)
// read can only show "empty", and the compare-and-swap then returns the real occupant.
__device__ __forceinline__ void hash_add_from(const HashView &t, uint64_t s, uint64_t cur, uint64_t code, uint32_t *status)
{
    const uint32_t limit = t.limit();
    for (uint32_t i = 1; i <= limit; ++i) {
        if (cur == 0) {
            cur = atomicCAS((unsigned long long *)&t.slots[s], 0ull, (unsigned long long)((code << HASH_CBITS) | 1ull));
            if (cur == 0) return;
        }
        if ((cur >> HASH_CBITS) == code) {
            // stop growing at SAT; overshoot is bounded by the threads in flight (< 2^20 < 2^22 - SAT)
            if ((uint32_t)(cur & HASH_CMASK) < HASH_SAT) atomicAdd((unsigned long long *)&t.slots[s], 1ull);
            return;
        }
        s = t.next(s, i, code);
        cur = t.slots[s];
    }
    atomicOr(status, 1u);
}

__device__ __forceinline__ uint32_t hash_probe(const HashView &t, uint64_t s, uint64_t cur, uint64_t code, bool *found)
{
    const uint32_t limit = t.limit();
    for (uint32_t i = 1; i <= limit; ++i) {
        if (cur == 0) break;
        if ((cur >> HASH_CBITS) == code) { *found = true; return (uint32_t)(cur & HASH_CMASK); }
        s = t.next(s, i, code);
        cur = t.slots[s];
    }
    *found = false;
    return 0;
}

// -------------------------------------------------------------------------------- K2 direct: global counts

template <typename KT, int TK>
__global__ __launch_bounds__(BLOCK) void kmer_count_kernel(const uint64_t *__restrict__ codes, const uint32_t *__restrict__ valid,
                                                           int64_t word_begin, int64_t word_end, int k, uint32_t *dense,
                                                           HashView t, uint32_t *status)
{
    for (int64_t w = word_begin + (int64_t)blockIdx.x * BLOCK + threadIdx.x; w < word_end; w += (int64_t)gridDim.x * BLOCK) {
        const uint64_t cw = codes[w];
        const uint32_t vw = valid[w];
        const uint64_t pw = w > 0 ? codes[w - 1] : 0;
        const uint32_t pv = w > 0 ? valid[w - 1] : 0;
        const uint32_t ok = (uint32_t)(runs_of(((uint64_t)vw << 32) | pv, k) >> 32);
        if (ok == 0) continue;
        Roller<KT, TK == TK_HASH && MINIMIZER> r;
        r.init(k);
        for (int i = 33 - k; i < 32; ++i) r.push((uint32_t)(pw >> (2 * i)) & 3u);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            KT canon[8];
            uint64_t cur[8];
            uint64_t hh[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = b * 8 + u;
                r.push((uint32_t)(cw >> (2 * j)) & 3u);
                canon[u] = r.canon();
                if ((ok >> j) & 1) {
                    if (TK == TK_DENSE) {
                        dense_add(dense, (uint32_t)canon[u]);
                    } else {                       // issue the first probe of the whole batch before resolving any
                        hh[u] = r.slot(t.log2_slots);
                        cur[u] = t.slots[hh[u]];
                    }
                }
            }
            if (TK == TK_HASH) {
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if ((ok >> (b * 8 + u)) & 1) hash_add_from(t, hh[u], cur[u], (uint64_t)canon[u], status);
            }
        }
    }
}

// merge (code,count) pairs of another table; counts saturate at SAT exactly (CAS loop; not a hot path)
__global__ __launch_bounds__(BLOCK) void kmer_merge_kernel(const uint64_t *__restrict__ pairs, int64_t n, int k, HashView t, uint32_t *status)
{
    const uint32_t limit = t.limit();
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
        const uint64_t p = pairs[i];
        if (p == 0) continue;
        const uint64_t code = p >> HASH_CBITS;
        uint32_t add = (uint32_t)(p & HASH_CMASK);
        if (add > HASH_SAT) add = HASH_SAT;
        uint64_t s = slot_of_code(code, k, t.log2_slots);
        bool done = false;
        for (uint32_t tries = 1; tries <= limit && !done; ++tries) {
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
            s = t.next(s, tries, code);
        }
        if (!done) atomicOr(status, 1u);
    }
}

// -------------------------------------------------------------------------------- K2 bucketed: partition + LDS counting
//
// record = canonical code (42 bits) | hfield << 42, hfield = the placement-hash bits that are still needed after the
// first scatter pass: the (log2_slots - bits1) bits below the level-1 digit, i.e. [level-2 digit | slot in bucket].
// The scatter-2 and bucket-count kernels therefore never hash: they read their digit / slot out of the record.

constexpr int HIST_BLOCK = 1024;
constexpr int TILE0 = 8192;           // records per stream tile: 256 lanes x 32 characters
constexpr int REC_PER_LANE = 16;      // scatter pass 2: records a lane keeps in registers
constexpr int TILE1 = BLOCK * REC_PER_LANE;
constexpr int MAX_FAN_BITS = 9;       // <= 512-way scatter per pass
constexpr int REC_KEY_BITS = 42;
constexpr uint64_t REC_KEY_MASK = (1ull << REC_KEY_BITS) - 1;

// A0: histogram of final bucket ids (top `bits` bits of the placement hash) over every valid k-mer of the word range
__global__ __launch_bounds__(HIST_BLOCK) void bucket_hist_kernel(const uint64_t *__restrict__ codes, const uint32_t *__restrict__ valid,
                                                                 int64_t word_begin, int64_t word_end, int k, int log2_slots, int bits,
                                                                 uint32_t bin_base, int n_bins, unsigned long long *__restrict__ hist)
{
    // one launch covers the bins [bin_base, bin_base + n_bins) (n_bins <= 2^15: 128 KiB of LDS counters); tables
    // with more buckets take several launches over the stream
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    for (int i = threadIdx.x; i < n_bins; i += HIST_BLOCK) lds[i] = 0;
    __syncthreads();
    for (int64_t w = word_begin + (int64_t)blockIdx.x * HIST_BLOCK + threadIdx.x; w < word_end; w += (int64_t)gridDim.x * HIST_BLOCK) {
        const uint64_t cw = codes[w];
        const uint32_t vw = valid[w];
        const uint64_t pw = w > 0 ? codes[w - 1] : 0;
        const uint32_t pv = w > 0 ? valid[w - 1] : 0;
        const uint32_t ok = (uint32_t)(runs_of(((uint64_t)vw << 32) | pv, k) >> 32);
        if (ok == 0) continue;
        Roller<uint64_t, MINIMIZER> r;
        r.init(k);
        for (int i = 33 - k; i < 32; ++i) r.push((uint32_t)(pw >> (2 * i)) & 3u);
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            r.push((uint32_t)(cw >> (2 * j)) & 3u);
            if ((ok >> j) & 1) {
                const uint32_t bin = (uint32_t)(r.slot(log2_slots) >> (log2_slots - bits)) - bin_base;
                if (bin < (uint32_t)n_bins) atomicAdd(&lds[bin], 1u);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_bins; i += HIST_BLOCK)
        if (lds[i]) atomicAdd(&hist[bin_base + i], (unsigned long long)lds[i]);
}

// exclusive prefix sum of hist[n] -> off[n+1] (one workgroup; n <= 2^17)
__global__ __launch_bounds__(HIST_BLOCK) void bucket_scan_kernel(const unsigned long long *__restrict__ hist, int n, unsigned long long *__restrict__ off)
{
    __shared__ unsigned long long part[HIST_BLOCK];
    const int per = (n + HIST_BLOCK - 1) / HIST_BLOCK;
    const int a = threadIdx.x * per, b = min(n, a + per);
    unsigned long long s = 0;
    for (int i = a; i < b; ++i) s += hist[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long run = 0;
        for (int i = 0; i < HIST_BLOCK; ++i) { unsigned long long v = part[i]; part[i] = run; run += v; }
        off[n] = run;
    }
    __syncthreads();
    unsigned long long run = part[threadIdx.x];
    for (int i = a; i < b; ++i) { off[i] = run; run += hist[i]; }
}

// LDS bookkeeping shared by both scatter passes
struct ScatterLds {
    uint32_t cnt[1 << MAX_FAN_BITS];
    uint32_t start[(1 << MAX_FAN_BITS) + 1];
    unsigned long long gbase[1 << MAX_FAN_BITS];
    uint32_t wave_tot[WAVES];
};

// exclusive scan of L.cnt[0..n_dig) into L.start[0..n_dig] (n_dig <= 512: two entries per lane); all lanes call
__device__ __forceinline__ void scatter_scan(ScatterLds &L, int n_dig)
{
    const int i0 = 2 * threadIdx.x, i1 = i0 + 1;
    const uint32_t a = i0 < n_dig ? L.cnt[i0] : 0, b = i1 < n_dig ? L.cnt[i1] : 0;
    const uint32_t v = a + b;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d);
        if ((int)(threadIdx.x & 63) >= d) incl += o;
    }
    if ((threadIdx.x & 63) == 63) L.wave_tot[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t before = 0;
    for (int wv = 0; wv < (int)(threadIdx.x >> 6); ++wv) before += L.wave_tot[wv];
    const uint32_t excl = before + incl - v;
    if (i0 < n_dig) L.start[i0] = excl;
    if (i1 < n_dig) L.start[i1] = excl + a;
    if (threadIdx.x == BLOCK - 1) L.start[n_dig] = before + incl;
    __syncthreads();
}

// A1: stream -> 2^bits1 regions.  A lane keeps the <= 32 records of its word in registers; one returning LDS atomic
// per record yields its rank inside its digit, the tile is then laid out digit-sorted in LDS and every digit's run
// is appended to its region with one global cursor add, so HBM receives contiguous runs.
__global__ __launch_bounds__(BLOCK) void scatter_stream_kernel(const uint64_t *__restrict__ codes, const uint32_t *__restrict__ valid,
                                                               int64_t word_begin, int64_t word_end, int k, int log2_slots, int bits1,
                                                               uint64_t *__restrict__ rec_out, const unsigned long long *__restrict__ off,
                                                               unsigned long long *__restrict__ cursor, int off_shift)
{
    __shared__ uint64_t buf[TILE0];
    __shared__ ScatterLds L;
    const int n_dig = 1 << bits1;
    const int low_bits = log2_slots - bits1;                        // hfield width
    const uint64_t low_mask64 = (1ull << low_bits) - 1;
    const int64_t n_tiles = (word_end - word_begin + BLOCK - 1) / BLOCK;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        for (int i = threadIdx.x; i < n_dig; i += BLOCK) L.cnt[i] = 0;
        __syncthreads();
        uint64_t rec[32];
        uint32_t dr[32];                                            // digit << 16 | rank inside the digit
        uint32_t ok = 0;
        const int64_t w = word_begin + tile * BLOCK + threadIdx.x;
        if (w < word_end) {
            const uint64_t cw = codes[w];
            const uint32_t vw = valid[w];
            const uint64_t pw = w > 0 ? codes[w - 1] : 0;
            const uint32_t pv = w > 0 ? valid[w - 1] : 0;
            ok = (uint32_t)(runs_of(((uint64_t)vw << 32) | pv, k) >> 32);
            if (ok) {
                Roller<uint64_t, MINIMIZER> r;
                r.init(k);
                for (int i = 33 - k; i < 32; ++i) r.push((uint32_t)(pw >> (2 * i)) & 3u);
#pragma unroll
                for (int j = 0; j < 32; ++j) {
                    r.push((uint32_t)(cw >> (2 * j)) & 3u);
                    if ((ok >> j) & 1) {
                        const uint64_t g = r.slot(log2_slots);
                        const uint32_t d = (uint32_t)(g >> low_bits);
                        rec[j] = r.canon() | ((g & low_mask64) << REC_KEY_BITS);
                        dr[j] = (d << 16) | atomicAdd(&L.cnt[d], 1u);
                    }
                }
            }
        }
        __syncthreads();
        scatter_scan(L, n_dig);
        for (int d = threadIdx.x; d < n_dig; d += BLOCK)
            if (L.cnt[d]) L.gbase[d] = off[(int64_t)d << off_shift] + atomicAdd(&cursor[d], (unsigned long long)L.cnt[d]);
#pragma unroll
        for (int j = 0; j < 32; ++j)
            if ((ok >> j) & 1) buf[L.start[dr[j] >> 16] + (dr[j] & 0xffffu)] = rec[j];
        __syncthreads();
        // copy out run by run (a wave per digit): lanes write consecutive records of one digit to consecutive addresses
        for (int d = threadIdx.x >> 6; d < n_dig; d += WAVES) {
            const uint32_t a0 = L.start[d], c = L.start[d + 1] - a0;
            const unsigned long long g0 = L.gbase[d];
            for (uint32_t i = threadIdx.x & 63; i < c; i += 64) rec_out[g0 + i] = buf[a0 + i];
        }
        __syncthreads();
    }
}

// A2: region blockIdx.y of pass 1 -> its 2^bits2 final buckets.  Digit = top bits2 of the record's hfield.
__global__ __launch_bounds__(BLOCK) void scatter_records_kernel(const uint64_t *__restrict__ rec_in, uint64_t *__restrict__ rec_out,
                                                                const unsigned long long *__restrict__ off,
                                                                unsigned long long *__restrict__ cursor, int bits2, int log2_bucket)
{
    __shared__ uint64_t buf[TILE1];
    __shared__ ScatterLds L;
    const int n_dig = 1 << bits2;
    const int region = blockIdx.y;
    const int64_t base_index = (int64_t)region << bits2;
    const int64_t r0 = (int64_t)off[base_index], r1 = (int64_t)off[base_index + n_dig];
    const int64_t n_tiles = (r1 - r0 + TILE1 - 1) / TILE1;
    const int dshift = REC_KEY_BITS + log2_bucket;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        for (int i = threadIdx.x; i < n_dig; i += BLOCK) L.cnt[i] = 0;
        __syncthreads();
        const int64_t t0 = r0 + tile * TILE1;
        uint64_t rec[REC_PER_LANE];
        uint32_t dr[REC_PER_LANE];
#pragma unroll
        for (int j = 0; j < REC_PER_LANE; ++j) {                    // all loads of the lane in flight together
            const int64_t i = t0 + j * BLOCK + threadIdx.x;
            rec[j] = i < r1 ? rec_in[i] : ~0ull;
        }
#pragma unroll
        for (int j = 0; j < REC_PER_LANE; ++j) {
            const int64_t i = t0 + j * BLOCK + threadIdx.x;
            if (i < r1) {
                const uint32_t d = (uint32_t)(rec[j] >> dshift);
                dr[j] = (d << 16) | atomicAdd(&L.cnt[d], 1u);
            }
        }
        __syncthreads();
        scatter_scan(L, n_dig);
        for (int d = threadIdx.x; d < n_dig; d += BLOCK)
            if (L.cnt[d]) L.gbase[d] = off[base_index + d] + atomicAdd(&cursor[base_index + d], (unsigned long long)L.cnt[d]);
#pragma unroll
        for (int j = 0; j < REC_PER_LANE; ++j) {
            const int64_t i = t0 + j * BLOCK + threadIdx.x;
            if (i < r1) buf[L.start[dr[j] >> 16] + (dr[j] & 0xffffu)] = rec[j];
        }
        __syncthreads();
        const uint32_t total = L.start[n_dig];
        for (uint32_t i = threadIdx.x; i < total; i += BLOCK) {
            const uint64_t r = buf[i];
            const uint32_t d = (uint32_t)(r >> dshift);
            rec_out[L.gbase[d] + (i - L.start[d])] = r;
        }
        __syncthreads();
    }
}

// insert one record into the LDS table, starting from an already fetched first slot; false = table full
__device__ __forceinline__ bool lds_insert(unsigned long long *tab, uint32_t smask, uint32_t primary, uint32_t limit,
                                           uint64_t rec, uint32_t s, unsigned long long cur)
{
    if (rec == ~0ull) return true;                                  // padding lane of the last batch
    const uint64_t code = rec & REC_KEY_MASK;
    for (uint32_t i = 1; i <= limit; ++i) {
        if (cur == 0) {
            cur = atomicCAS(&tab[s], 0ull, (unsigned long long)((code << HASH_CBITS) | 1ull));
            if (cur == 0) return true;
        }
        if ((cur >> HASH_CBITS) == code) {
            if ((uint32_t)(cur & HASH_CMASK) < HASH_SAT) atomicAdd(&tab[s], 1ull);
            return true;
        }
        s = i == primary ? (uint32_t)(mix64(code) & smask) : ((s + 1) & smask);
        cur = tab[s];
    }
    return false;
}

// B: one workgroup per final bucket.  The bucket's slice of the table (2^log2_bucket slots) lives in LDS while the
// bucket's records stream through (8 loads per lane in flight); the LDS image is then written back as the slice.
constexpr int CNT_BATCH = 8;               // PG_RESOLVE below is written for exactly 8
__global__ __launch_bounds__(HIST_BLOCK) void bucket_count_kernel(const uint64_t *__restrict__ rec, const unsigned long long *__restrict__ off,
                                                                  HashView t, int accumulate, uint32_t *status)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long tab[];
    const uint32_t n_slots = 1u << t.log2_bucket;
    const uint32_t smask = n_slots - 1;
    const uint32_t primary = t.primary(), limit = t.limit();
    uint64_t *slice = t.slots + ((uint64_t)blockIdx.x << t.log2_bucket);
    const int64_t r0 = (int64_t)off[blockIdx.x], r1 = (int64_t)off[blockIdx.x + 1];
    if (r0 == r1 && accumulate) return;                         // nothing to add, slice stays as it is
    for (uint32_t i = threadIdx.x; i < n_slots; i += HIST_BLOCK) tab[i] = accumulate ? slice[i] : 0ull;
    __syncthreads();
    bool full = false;
    for (int64_t base = r0; base < r1; base += (int64_t)HIST_BLOCK * CNT_BATCH) {
        uint64_t rr[CNT_BATCH];
#pragma unroll
        for (int j = 0; j < CNT_BATCH; ++j) {
            const int64_t i = base + (int64_t)j * HIST_BLOCK + threadIdx.x;
            rr[j] = i < r1 ? rec[i] : ~0ull;
        }
        uint32_t ss[CNT_BATCH];
        unsigned long long first[CNT_BATCH];
#pragma unroll
        for (int j = 0; j < CNT_BATCH; ++j) {                       // first probes of the batch issued back to back
            ss[j] = (uint32_t)(rr[j] >> REC_KEY_BITS) & smask;
            first[j] = rr[j] == ~0ull ? 0ull : tab[ss[j]];
        }
        // resolved one by one, written out by hand: an unrolled loop around the probe loop is not unrolled by hipcc and
        // would push the batch arrays into scratch
#define PG_RESOLVE(J) full |= !lds_insert(tab, smask, primary, limit, rr[J], ss[J], first[J]);
        PG_RESOLVE(0) PG_RESOLVE(1) PG_RESOLVE(2) PG_RESOLVE(3) PG_RESOLVE(4) PG_RESOLVE(5) PG_RESOLVE(6) PG_RESOLVE(7)
#undef PG_RESOLVE
    }
    if (full) atomicOr(status, 1u);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_slots; i += HIST_BLOCK) slice[i] = tab[i];
}

// Bucket-wise merge of other tables into this one: the compacted tables of the other ranks arrive bucket by bucket
// (a table's slots are laid out by bucket, so compaction keeps bucket order).  One workgroup per bucket loads its slice
// into LDS, adds every foreign entry of that bucket (counts saturate exactly) and writes the slice back.
//   pairs  : the foreign tables' occupied slots, concatenated part after part
//   seg    : [n_parts][n_buckets + 1] offsets into `pairs` (absolute)
__global__ __launch_bounds__(HIST_BLOCK) void bucket_merge_kernel(const uint64_t *__restrict__ pairs, const long long *__restrict__ seg,
                                                                  int n_parts, int k, HashView t, uint32_t *status)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long tab[];
    const uint32_t n_slots = 1u << t.log2_bucket;
    const uint32_t smask = n_slots - 1;
    const uint32_t primary = t.primary(), limit = t.limit();
    const int64_t n_buckets = (int64_t)1 << (t.log2_slots - t.log2_bucket);
    uint64_t *slice = t.slots + ((uint64_t)blockIdx.x << t.log2_bucket);
    int64_t total = 0;
    for (int p = 0; p < n_parts; ++p) total += seg[p * (n_buckets + 1) + blockIdx.x + 1] - seg[p * (n_buckets + 1) + blockIdx.x];
    if (total == 0) return;
    for (uint32_t i = threadIdx.x; i < n_slots; i += HIST_BLOCK) tab[i] = slice[i];
    __syncthreads();
    bool full = false;
    for (int p = 0; p < n_parts; ++p) {
        const int64_t a = seg[p * (n_buckets + 1) + blockIdx.x], b = seg[p * (n_buckets + 1) + blockIdx.x + 1];
        for (int64_t i = a + threadIdx.x; i < b; i += HIST_BLOCK) {
            const uint64_t e = pairs[i];
            const uint64_t code = e >> HASH_CBITS;
            uint32_t add = (uint32_t)(e & HASH_CMASK);
            if (add > HASH_SAT) add = HASH_SAT;
            uint32_t s = (uint32_t)slot_of_code(code, k, t.log2_slots) & smask;
            bool done = false;
            for (uint32_t tries = 1; tries <= limit && !done; ++tries) {
                unsigned long long cur = tab[s];
                for (;;) {
                    if (cur != 0 && (cur >> HASH_CBITS) != code) break;
                    const uint32_t have = (uint32_t)(cur & HASH_CMASK);
                    const uint32_t sum = have + add > HASH_SAT ? HASH_SAT : have + add;
                    const unsigned long long want = (code << HASH_CBITS) | sum;
                    const unsigned long long old = atomicCAS(&tab[s], cur, want);
                    if (old == cur) { done = true; break; }
                    cur = old;
                }
                s = tries == primary ? (uint32_t)(mix64(code) & smask) : ((s + 1) & smask);
            }
            full |= !done;
        }
    }
    if (full) atomicOr(status, 1u);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_slots; i += HIST_BLOCK) slice[i] = tab[i];
}

// -------------------------------------------------------------------------------- K1 + K3: per-run rows

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

        Roller<KT, TK == TK_HASH && MINIMIZER> r;
        r.init(kk);
        for (int i = 33 - kroll; i < 32; ++i) r.push((uint32_t)(pw >> (2 * i)) & 3u);
        // batches of 8 characters: roll, issue the table reads of the batch, then bin them
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            KT canon[8];
            uint64_t cur[8];
            uint64_t hh[8];
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
                            hh[u] = r.slot(t.log2_slots);
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
                        } else {
                            cnt = hash_probe(t, hh[u], cur[u], (uint64_t)canon[u], &found);
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

int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return pg_fail(PG_EHIP, "%s: %s", what, hipGetErrorString(e));
    return PG_OK;
}

int grid_for(int64_t items, int block = BLOCK)
{
    int64_t blocks = (items + block - 1) / block;
    const int64_t cap = 256 * 16;     // 256 CUs x 16 resident workgroups' worth, grid-stride beyond
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

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
    int64_t cap;                      // record capacity of each of the two record buffers
    size_t hist_off, off_off, cur1_off, cur2_off, bufa_off, bufb_off, total;
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
    // first-level fan-out: 8 bits, 9 when the record's 22 spare bits could not hold the rest of the slot index
    p->bits1 = p->bits < 8 ? p->bits : (t->log2_slots - 8 > 64 - REC_KEY_BITS ? 9 : 8);
    if (p->bits1 > p->bits) p->bits1 = p->bits;
    p->bits2 = p->bits - p->bits1;
    if (p->bits2 > MAX_FAN_BITS) return pg_fail(PG_EINVAL, "too many buckets for two scatter passes");
    if (t->log2_slots - p->bits1 > 64 - REC_KEY_BITS)
        return pg_fail(PG_EINVAL, "bucketed counting needs log2_slots <= %d (got %d)", 64 - REC_KEY_BITS + p->bits1, t->log2_slots);
    p->cap = n_words * 32;
    const size_t nb = (size_t)1 << p->bits;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += (bytes + 255) / 256 * 256; return at; };
    p->hist_off = take(nb * 8);
    p->off_off = take((nb + 1) * 8);
    p->cur1_off = take(((size_t)1 << p->bits1) * 8);
    p->cur2_off = take(nb * 8);
    p->bufa_off = take((size_t)p->cap * 8);
    p->bufb_off = p->bits2 ? take((size_t)p->cap * 8) : p->bufa_off;
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
    if (t->kind == PG_TABLE_HASH && !status) return pg_fail(PG_EINVAL, "pg_kmer_count: hash tables need a status word");
    if (word_end == word_begin) return PG_OK;
    hipStream_t s = (hipStream_t)stream;
    int grid = grid_for(word_end - word_begin);
    if (t->kind == PG_TABLE_DENSE) {
        HashView none{nullptr, 0, 0};
        hipLaunchKernelGGL((kmer_count_kernel<uint32_t, TK_DENSE>), dim3(grid), dim3(BLOCK), 0, s, codes, valid, word_begin, word_end,
                           t->k, (uint32_t *)t->data, none, status);
    } else {
        hipLaunchKernelGGL((kmer_count_kernel<uint64_t, TK_HASH>), dim3(grid), dim3(BLOCK), 0, s, codes, valid, word_begin, word_end,
                           t->k, (uint32_t *)nullptr, view_of(t), status);
    }
    return check_launch("pg_kmer_count");
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

extern "C" int pg_kmer_count_bucketed(const uint64_t *codes, const uint32_t *valid, int64_t word_begin, int64_t word_end,
                                      const pg_table *t, int accumulate, void *workspace, int64_t workspace_bytes,
                                      uint32_t *status, void *stream)
{
    if (!codes || !valid || !workspace || !status) return pg_fail(PG_EINVAL, "pg_kmer_count_bucketed: null argument");
    if (word_begin < 0 || word_end < word_begin) return pg_fail(PG_EINVAL, "pg_kmer_count_bucketed: bad word range");
    int rc = check_table(t);
    if (rc) return rc;
    BucketPlan p;
    rc = plan_buckets(t, word_end - word_begin, &p);
    if (rc) return rc;
    if ((int64_t)p.total > workspace_bytes)
        return pg_fail(PG_EINVAL, "pg_kmer_count_bucketed: workspace of %lld bytes, %lld needed", (long long)workspace_bytes, (long long)p.total);
    if ((reinterpret_cast<uintptr_t>(workspace) & 255) != 0) return pg_fail(PG_EINVAL, "pg_kmer_count_bucketed: workspace must be 256-byte aligned");
    if (word_end == word_begin) return PG_OK;
    hipStream_t s = (hipStream_t)stream;
    char *ws = (char *)workspace;
    auto *hist = (unsigned long long *)(ws + p.hist_off);
    auto *off = (unsigned long long *)(ws + p.off_off);
    auto *cur1 = (unsigned long long *)(ws + p.cur1_off);
    auto *cur2 = (unsigned long long *)(ws + p.cur2_off);
    auto *bufa = (uint64_t *)(ws + p.bufa_off);
    auto *bufb = (uint64_t *)(ws + p.bufb_off);
    const int nb = 1 << p.bits;
    // counters are contiguous at the front of the workspace: one clear
    if (hipMemsetAsync(ws, 0, p.bufa_off, s) != hipSuccess) return pg_fail(PG_EHIP, "pg_kmer_count_bucketed: memset failed");
    const int64_t n_words = word_end - word_begin;

    // both LDS-heavy kernels may need more than the default 64 KiB of dynamic LDS
    if (nb > (1 << 14) || ((size_t)8 << t->log2_bucket_slots) > 64 * 1024) {
        if (hipFuncSetAttribute((const void *)bucket_hist_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void *)bucket_count_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess)
            return pg_fail(PG_EHIP, "pg_kmer_count_bucketed: cannot raise the dynamic LDS limit");
    }
    // A0 histogram of final bucket ids, then offsets
    {
        int grid = (int)((n_words + HIST_BLOCK - 1) / HIST_BLOCK);
        if (grid > 512) grid = 512;
        const int per = nb < (1 << 15) ? nb : (1 << 15);
        for (int base = 0; base < nb; base += per)
            hipLaunchKernelGGL(bucket_hist_kernel, dim3(grid), dim3(HIST_BLOCK), (size_t)per * 4, s, codes, valid, word_begin, word_end,
                               t->k, t->log2_slots, p.bits, (uint32_t)base, per, hist);
        hipLaunchKernelGGL(bucket_scan_kernel, dim3(1), dim3(HIST_BLOCK), 0, s, hist, nb, off);
    }
    // A1: stream -> 2^bits1 regions (region d1 = final buckets [d1 << bits2, (d1+1) << bits2))
    {
        int64_t tiles = (n_words + BLOCK - 1) / BLOCK;
        int grid = (int)(tiles > 8192 ? 8192 : tiles);
        hipLaunchKernelGGL(scatter_stream_kernel, dim3(grid), dim3(BLOCK), 0, s, codes, valid, word_begin, word_end, t->k,
                           t->log2_slots, p.bits1, bufa, off, cur1, p.bits2);
    }
    // A2: every region -> its 2^bits2 final buckets
    if (p.bits2)
        hipLaunchKernelGGL(scatter_records_kernel, dim3(96, 1 << p.bits1), dim3(BLOCK), 0, s, bufa, bufb, off, cur2, p.bits2,
                           t->log2_bucket_slots);
    // B: count every bucket inside LDS and write its slice of the table
    {
        const size_t lds = (size_t)8 << t->log2_bucket_slots;
        hipLaunchKernelGGL(bucket_count_kernel, dim3(nb), dim3(HIST_BLOCK), lds, s, p.bits2 ? bufb : bufa, off, view_of(t), accumulate ? 1 : 0, status);
    }
    return check_launch("pg_kmer_count_bucketed");
}

extern "C" int pg_kmer_merge(const uint64_t *pairs, int64_t n, const pg_table *t, uint32_t *status, void *stream)
{
    int rc = check_table(t);
    if (rc) return rc;
    if (t->kind != PG_TABLE_HASH) return pg_fail(PG_EINVAL, "pg_kmer_merge: hash tables only (dense tables are summed with an all-reduce)");
    if (n < 0 || (n > 0 && !pairs) || !status) return pg_fail(PG_EINVAL, "pg_kmer_merge: bad arguments");
    if (n == 0) return PG_OK;
    hipLaunchKernelGGL(kmer_merge_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, (hipStream_t)stream, pairs, n, t->k, view_of(t), status);
    return check_launch("pg_kmer_merge");
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
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void *)bucket_merge_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess)
        return pg_fail(PG_EHIP, "pg_kmer_merge_bucketed: cannot raise the dynamic LDS limit");
    hipLaunchKernelGGL(bucket_merge_kernel, dim3(1u << bits), dim3(HIST_BLOCK), lds, (hipStream_t)stream, pairs, (const long long *)seg,
                       n_parts, t->k, view_of(t), status);
    return check_launch("pg_kmer_merge_bucketed");
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
        if (t->kind == PG_TABLE_HASH && (int64_t)window * vsize > (int64_t)PG_HASH_COUNT_SAT)
            return pg_fail(PG_EINVAL, "pg_features: window*vector_size %lld exceeds the exact range of the hash table (%u)",
                           (long long)window * vsize, PG_HASH_COUNT_SAT);
        kind = t->kind == PG_TABLE_DENSE ? TK_DENSE : TK_HASH;
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
    else PG_LAUNCH(uint64_t, TK_HASH);
#undef PG_LAUNCH
    return check_launch("pg_features");
}
