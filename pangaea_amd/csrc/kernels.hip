// kernels.hip -- gfx950 kernels of the barcode k-mer feature path + their C-ABI launchers.
//
// Replaces (reference file:line, /root/reference/src):
//   K2  pg_kmer_count   jellyfish count -C (feature.py:94) + dump reload (cpptools/count_kmer.cpp:139-170)
//   K1  pg_features/tnf cpptools/count_tnf.cpp:78-113  (per-run canonical k_tnf-mer counts)
//   K3  pg_features/abd cpptools/count_kmer.cpp:55-108 (per-run histogram of global multiplicities)
//
// Work decomposition (wave64, 256-thread workgroups):
//   * the read stream is 2-bit codes + 1-bit validity, 32 characters per word (include/pangaea_feat.h);
//     one lane owns one word per step, so a wave reads 512 B of codes + 256 B of validity, coalesced;
//   * a lane rolls the forward and reverse-complement codes over its 32 characters after pre-rolling
//     the k-1 characters before its word (taken from the previous word), so no cross-lane traffic;
//   * which positions end a valid k-mer comes from one bit-parallel pass over the 64-bit validity
//     window (runs of >= k ones), not from a per-character run counter;
//   * integer counting only: LDS histograms per wavefront, global atomics into the count tables.
//     No MFMA -- there is no contraction on this path.
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

enum { TK_NONE = 0, TK_DENSE = 1, TK_HASH = 2 };

__device__ __forceinline__ uint64_t mix64(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return x;
}

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

// -------------------------------------------------------------------------------- table access

__device__ __forceinline__ void dense_add(uint32_t *table, uint32_t code) { atomicAdd(&table[code], 1u); }

// slot = (code << 22) | count ; 0 = empty.  Keys never change once written, so a stale (cached)
// read can only show "empty", and the compare-and-swap then returns the real occupant.
__device__ __forceinline__ void hash_add_from(uint64_t *slots, uint64_t mask, uint64_t h, uint64_t cur, uint64_t code, uint32_t *status)
{
    const uint32_t limit = mask < MAX_PROBE ? (uint32_t)mask + 1u : MAX_PROBE;
    for (uint32_t probe = 0; probe < limit; ++probe) {
        if (cur == 0) {
            cur = atomicCAS((unsigned long long *)&slots[h], 0ull, (unsigned long long)((code << HASH_CBITS) | 1ull));
            if (cur == 0) return;
        }
        if ((cur >> HASH_CBITS) == code) {
            // stop growing at SAT; overshoot is bounded by the threads in flight (< 2^20 < 2^22 - SAT)
            if ((uint32_t)(cur & HASH_CMASK) < HASH_SAT) atomicAdd((unsigned long long *)&slots[h], 1ull);
            return;
        }
        h = (h + 1) & mask;
        cur = slots[h];
    }
    atomicOr(status, 1u);
}

__device__ __forceinline__ uint32_t hash_probe(const uint64_t *slots, uint64_t mask, uint64_t h, uint64_t cur, uint64_t code, bool *found)
{
    for (uint32_t probe = 0; probe < MAX_PROBE; ++probe) {
        if (cur == 0) break;
        if ((cur >> HASH_CBITS) == code) { *found = true; return (uint32_t)(cur & HASH_CMASK); }
        h = (h + 1) & mask;
        cur = slots[h];
    }
    *found = false;
    return 0;
}

// -------------------------------------------------------------------------------- K2: global counts

template <typename KT, int TK>
__global__ __launch_bounds__(BLOCK) void kmer_count_kernel(const uint64_t *__restrict__ codes, const uint32_t *__restrict__ valid,
                                                           int64_t word_begin, int64_t word_end, int k, void *table,
                                                           int log2_slots, uint32_t *status)
{
    const KT kmask = low_mask<KT>(k);
    const int rc_shift = 2 * (k - 1);
    const uint64_t smask = TK == TK_HASH ? (1ull << log2_slots) - 1 : 0;
    for (int64_t w = word_begin + (int64_t)blockIdx.x * BLOCK + threadIdx.x; w < word_end; w += (int64_t)gridDim.x * BLOCK) {
        const uint64_t cw = codes[w];
        const uint32_t vw = valid[w];
        const uint64_t pw = w > 0 ? codes[w - 1] : 0;
        const uint32_t pv = w > 0 ? valid[w - 1] : 0;
        const uint32_t ok = (uint32_t)(runs_of(((uint64_t)vw << 32) | pv, k) >> 32);
        if (ok == 0) continue;
        KT fw = 0, rc = 0;
        for (int i = 33 - k; i < 32; ++i) {                // pre-roll the k-1 characters before the word
            KT c = (KT)((pw >> (2 * i)) & 3);
            fw = (KT)(fw << 2) | c;
            rc = (KT)(rc >> 2) | (KT)((c ^ 2) << rc_shift);
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            KT canon[8];
            uint64_t cur[8];
            uint64_t hh[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = b * 8 + u;
                KT c = (KT)((cw >> (2 * j)) & 3);
                fw = (KT)(fw << 2) | c;
                rc = (KT)(rc >> 2) | (KT)((c ^ 2) << rc_shift);
                KT f = fw & kmask;
                canon[u] = f < rc ? f : rc;
                if ((ok >> j) & 1) {
                    if (TK == TK_DENSE) {
                        dense_add((uint32_t *)table, (uint32_t)canon[u]);
                    } else {                       // issue the first probe of the whole batch before resolving any
                        hh[u] = mix64((uint64_t)canon[u]) >> (64 - log2_slots);
                        cur[u] = ((const uint64_t *)table)[hh[u]];
                    }
                }
            }
            if (TK == TK_HASH) {
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if ((ok >> (b * 8 + u)) & 1) hash_add_from((uint64_t *)table, smask, hh[u], cur[u], (uint64_t)canon[u], status);
            }
        }
    }
}

// merge (code,count) pairs of another table; counts saturate at SAT exactly (CAS loop; not a hot path)
__global__ __launch_bounds__(BLOCK) void kmer_merge_kernel(const uint64_t *__restrict__ pairs, int64_t n, uint64_t *slots,
                                                           int log2_slots, uint32_t *status)
{
    const uint64_t mask = (1ull << log2_slots) - 1;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
        const uint64_t p = pairs[i];
        if (p == 0) continue;
        const uint64_t code = p >> HASH_CBITS;
        uint32_t add = (uint32_t)(p & HASH_CMASK);
        if (add > HASH_SAT) add = HASH_SAT;
        uint64_t h = mix64(code) >> (64 - log2_slots);
        bool done = false;
        for (uint32_t probe = 0; probe < MAX_PROBE && !done; ++probe) {
            uint64_t cur = __hip_atomic_load(&slots[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (;;) {
                if (cur != 0 && (cur >> HASH_CBITS) != code) break;          // occupied by another key
                uint32_t have = (uint32_t)(cur & HASH_CMASK);
                uint32_t sum = have + add > HASH_SAT ? HASH_SAT : have + add;
                uint64_t want = (code << HASH_CBITS) | sum;
                uint64_t old = atomicCAS((unsigned long long *)&slots[h], (unsigned long long)cur, (unsigned long long)want);
                if (old == cur) { done = true; break; }
                cur = old;
            }
            h = (h + 1) & mask;
        }
        if (!done) atomicOr(status, 1u);
    }
}

// -------------------------------------------------------------------------------- K1 + K3: per-run rows

// LDS: abd_copies x [vsize] abundance bins, then tnf_copies x [4^k_tnf] raw k_tnf-mer bins
// (one copy per wavefront while that fits comfortably, else one shared copy).
template <typename KT, int TK>
__global__ __launch_bounds__(BLOCK) void features_kernel(const uint64_t *__restrict__ codes, const uint32_t *__restrict__ valid,
                                                         int64_t n_words, const int32_t *__restrict__ seg_row,
                                                         const int64_t *__restrict__ seg_start, const int64_t *__restrict__ seg_end,
                                                         int k_tnf, const uint16_t *__restrict__ colmap, int tnf_cols, int tnf_copies, int abd_copies,
                                                         int32_t *__restrict__ tnf_out, int k, const void *__restrict__ table,
                                                         int log2_slots, uint32_t window, int vsize, int32_t *__restrict__ abd_out)
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
    const int kroll = do_abd ? (k > k_tnf || !do_tnf ? k : k_tnf) : k_tnf;   // characters to pre-roll + 1
    const KT kmask = low_mask<KT>(do_abd ? k : 1);
    const int rc_shift = do_abd ? 2 * (k - 1) : 0;
    const uint32_t tmask = do_tnf ? (uint32_t)n_raw - 1u : 0u;
    const uint64_t smask = TK == TK_HASH ? (1ull << log2_slots) - 1 : 0;

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

        KT fw = 0, rc = 0;
        for (int i = 33 - kroll; i < 32; ++i) {
            KT c = (KT)((pw >> (2 * i)) & 3);
            fw = (KT)(fw << 2) | c;
            rc = (KT)(rc >> 2) | (KT)((c ^ 2) << rc_shift);
        }
        // batches of 8 characters: roll, issue the table reads of the batch, then bin them
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            KT canon[8];
            uint64_t cur[8];
            uint64_t hh[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = b * 8 + u;
                KT c = (KT)((cw >> (2 * j)) & 3);
                fw = (KT)(fw << 2) | c;
                if (do_abd) rc = (KT)(rc >> 2) | (KT)((c ^ 2) << rc_shift);
                if (do_tnf && ((ok_t >> j) & 1)) atomicAdd(&my_tnf[(uint32_t)fw & tmask], 1u);
                if (TK != TK_NONE) {
                    KT f = fw & kmask;
                    canon[u] = f < rc ? f : rc;
                    cur[u] = 0;
                    if ((ok_a >> j) & 1) {
                        if (TK == TK_DENSE) {
                            cur[u] = ((const uint32_t *)table)[(uint32_t)canon[u]];
                        } else {
                            hh[u] = mix64((uint64_t)canon[u]) >> (64 - log2_slots);
                            cur[u] = ((const uint64_t *)table)[hh[u]];
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
                            cnt = hash_probe((const uint64_t *)table, smask, hh[u], cur[u], (uint64_t)canon[u], &found);
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

int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return pg_fail(PG_EHIP, "%s: %s", what, hipGetErrorString(e));
    return PG_OK;
}

int grid_for(int64_t items)
{
    int64_t blocks = (items + BLOCK - 1) / BLOCK;
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
    } else {
        return pg_fail(PG_EINVAL, "unknown table kind %d", t->kind);
    }
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
    if (t->kind == PG_TABLE_DENSE)
        hipLaunchKernelGGL((kmer_count_kernel<uint32_t, TK_DENSE>), dim3(grid), dim3(BLOCK), 0, s, codes, valid, word_begin, word_end,
                           t->k, t->data, 0, status);
    else
        hipLaunchKernelGGL((kmer_count_kernel<uint64_t, TK_HASH>), dim3(grid), dim3(BLOCK), 0, s, codes, valid, word_begin, word_end,
                           t->k, t->data, t->log2_slots, status);
    return check_launch("pg_kmer_count");
}

extern "C" int pg_kmer_merge(const uint64_t *pairs, int64_t n, const pg_table *t, uint32_t *status, void *stream)
{
    int rc = check_table(t);
    if (rc) return rc;
    if (t->kind != PG_TABLE_HASH) return pg_fail(PG_EINVAL, "pg_kmer_merge: hash tables only (dense tables are summed with an all-reduce)");
    if (n < 0 || (n > 0 && !pairs) || !status) return pg_fail(PG_EINVAL, "pg_kmer_merge: bad arguments");
    if (n == 0) return PG_OK;
    hipLaunchKernelGGL(kmer_merge_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, (hipStream_t)stream, pairs, n, (uint64_t *)t->data,
                       t->log2_slots, status);
    return check_launch("pg_kmer_merge");
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
    int kind = TK_NONE, k = 0, log2_slots = 0;
    const void *data = nullptr;
    if (do_abd) {
        int rc = check_table(t);
        if (rc) return rc;
        if (window < 1 || vsize < 1 || vsize > 8192) return pg_fail(PG_EINVAL, "pg_features: window %d / vector size %d out of range", window, vsize);
        if (t->kind == PG_TABLE_HASH && (int64_t)window * vsize > (int64_t)PG_HASH_COUNT_SAT)
            return pg_fail(PG_EINVAL, "pg_features: window*vector_size %lld exceeds the exact range of the hash table (%u)",
                           (long long)window * vsize, PG_HASH_COUNT_SAT);
        kind = t->kind == PG_TABLE_DENSE ? TK_DENSE : TK_HASH;
        k = t->k; log2_slots = t->log2_slots; data = t->data;
    }
    if (n_segs == 0) return PG_OK;
    const int abd_copies = do_abd ? (vsize <= 1024 ? WAVES : 1) : 0;
    const size_t lds_bytes = sizeof(uint32_t) * ((size_t)abd_copies * (do_abd ? vsize : 0) + (size_t)tnf_copies * n_raw);
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)n_segs), block(BLOCK);
#define PG_LAUNCH(KT, TK)                                                                                                   \
    hipLaunchKernelGGL((features_kernel<KT, TK>), grid, block, lds_bytes, s, codes, valid, n_words, seg_row, seg_start, seg_end, \
                       k_tnf, colmap, tnf_cols, tnf_copies, abd_copies, tnf_out, k, data, log2_slots, (uint32_t)window, vsize, abd_out)
    if (kind == TK_NONE) PG_LAUNCH(uint32_t, TK_NONE);
    else if (kind == TK_DENSE) PG_LAUNCH(uint32_t, TK_DENSE);
    else PG_LAUNCH(uint64_t, TK_HASH);
#undef PG_LAUNCH
    return check_launch("pg_features");
}
