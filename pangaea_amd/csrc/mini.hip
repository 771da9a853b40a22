// mini.hip -- the super-k-mer form of K2 (global multiplicities) with K3's lookups inside, for PG_TABLE_MINI tables.
//
// Replaces (reference file:line, /root/reference/src): jellyfish count -C (feature.py:94) + the dump reload of
// cpptools/count_kmer.cpp:139-170, and the per-occurrence table lookups of cpptools/count_kmer.cpp:86-96.
//
// Why: the key-partitioned pipeline of kernels.hip moves one 8-byte record per k-mer OCCURRENCE through two scatter
// passes and two LDS-table passes (20.8 GB per trip at 10 M read pairs).  Here a k-mer's bucket is a hash of its MINIMIZER
// (pg_device.hpp), consecutive k-mers of a read share it, and what travels is the SUPER-k-mer: one 12-byte record
//     bases : uint64  the 32 characters ending at the record's last character, stream order (oldest in the low bits)
//     meta  : uint32  row << 12 | (n - 1) << 8 | d2      row = index of the row the k-mers lie in (MINI_ROW_NONE: none),
//                                                        n = k-mers in the record (they end at the last n characters),
//                                                        d2 = the bucket's second-pass digit
// for a run of n <= 16 k-mers (about 4.3 on 150 bp reads, k = 21), i.e. under 3 bytes per occurrence.  The bucket
// workgroups re-derive the k-mers: forward codes are shifts of the character-reversed word, reverse-complement codes are
// shifts of the complemented word -- one reversal per record, two shifts and a compare per k-mer.
//
// Launches of one count (10 M pairs: 94 M words, ~600 M records):
//   mini_plan_kernel      (pg_mini_plan; depends on the stream, the rows and the geometry only) records per bucket and,
//                         per 4096-word chunk, records per first-pass region  -> scans -> exact offsets
//   mini_scatter_kernel   A1': one 512-thread workgroup per chunk, 512 words per round; a lane segments its word, ranks its
//                         records with one LDS atomic each, the round is laid out region-sorted in LDS and copied out
//   mini_scatter2_kernel  A2': every region -> its buckets (LDS multisplit, one global cursor add per digit and tile)
//   mini_count_kernel     B' + S1': one workgroup per bucket, 8-byte LDS slots (canonical code << 22 | count); the packed
//                         slice is written once; then the counts become bins in place and the records are read again
//                         (L2 / MALL) and looked up: (row, bin) words for the row shuffle of kernels.hip
#include <type_traits>

#include "pg_device.hpp"

namespace {

constexpr int S1_BLOCK = 512;                          // threads of a first-pass workgroup (256 VGPRs per lane: no spills)
constexpr int ROUND_WORDS = S1_BLOCK;                  // words per round of the first pass: one per lane
constexpr int MINI_CHUNK_WORDS = 4096;                 // words per chunk (one workgroup of the first pass)
constexpr int ROUNDS_PER_CHUNK = MINI_CHUNK_WORDS / ROUND_WORDS;
constexpr int STAGE_CAP = 8 * S1_BLOCK;                // records of a (sub-)round staged in LDS: 8 positions x 512 lanes always fit
constexpr int META_D2_BITS = 8, META_LEN_BITS = 4, META_ROW_SHIFT = META_D2_BITS + META_LEN_BITS;
constexpr uint32_t MINI_ROW_NONE = (1u << (32 - META_ROW_SHIFT)) - 1u;
constexpr int MINI_MAX_LEN = 1 << META_LEN_BITS;       // k-mers per record
constexpr int MINI_BITS1 = 8;                          // first-pass digits (regions); a table of 2^16 buckets: 2^8 regions x 2^8 buckets each
constexpr int MINI_MAX_BITS1 = 8;
#ifndef PG_SHORT_MAX
#define PG_SHORT_MAX 4
#endif
constexpr int PG_SHORT_MAX_ = PG_SHORT_MAX;                       // a record with at most this many k-mers is "short" (see mini_count_kernel)
static_assert(PG_MINI_MAX_ROWS == (int)MINI_ROW_NONE - 1, "row field of the record");
static_assert(PG_MINI_MAX_LOG2_BUCKETS == MINI_MAX_BITS1 + META_D2_BITS, "bucket id = region digit + second-pass digit");

struct MiniView {
    uint64_t *slots;
    int log2_slots, log2_bucket, k;
    __device__ __forceinline__ int bits() const { return log2_slots - log2_bucket; }
};

// first row whose end lies beyond the first character of each round (rows sorted, disjoint)
__global__ __launch_bounds__(BLOCK) void round_rows_kernel(const int64_t *__restrict__ row_end, int64_t n_rows, int64_t word_begin,
                                                           int64_t n_rounds, int32_t *__restrict__ round_row)
{
    const int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= n_rounds) return;
    const int64_t c0 = (word_begin + t * ROUND_WORDS) * 32;
    int64_t lo = 0, hi = n_rows;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (row_end[mid] > c0) hi = mid; else lo = mid + 1;
    }
    round_row[t] = (int32_t)lo;
}

// ---- segmentation of ONE word into super-k-mer records.  A record is a maximal run of consecutive valid k-mers of the word
// (records never cross words: the lane that owns the word sees the k - 1 characters before it, like every stream reader
// here) that share their bucket and their row, capped at `cap` k-mers.  emit(e, n, bucket, row) is called once per record, e =
// position of its last character in the word; e is a compile-time constant at every call site (the position loop is fully
// unrolled), so callers may index register arrays with it.  The plan kernel and the scatter kernel both call this: they
// see the same records by construction.
struct RowBits {
    // where the rows (sorted, disjoint character ranges) cut the lane's 32 characters: bit p of `starts` = a row starts at
    // character p (p > 0), bit p of `ends` = character p is the first one behind a row; r0 = the row that holds character 0
    // (inside0) or the next row to start.  No per-position loop is needed afterwards.
    uint32_t starts, ends, r0;
    bool inside0;
    __device__ __forceinline__ void init(const int64_t *__restrict__ row_start, const int64_t *__restrict__ row_end, int64_t n_rows, int64_t first, int64_t pos0)
    {
        starts = ends = 0; r0 = 0; inside0 = false;
        if (!row_start) return;
        int64_t r = first;
        while (r < n_rows && row_end[r] <= pos0) ++r;           // first row that ends behind character 0
        r0 = (uint32_t)r;
        inside0 = r < n_rows && row_start[r] <= pos0;
        for (; r < n_rows; ++r) {
            const int64_t a = row_start[r] - pos0, b = row_end[r] - pos0;
            if (a >= 32) break;
            if (a > 0) starts |= 1u << a;
            if (b < 32) ends |= 1u << b;
        }
    }
    __device__ __forceinline__ uint32_t cuts() const { return starts | ends; }
    // row of character p, MINI_ROW_NONE outside every row
    __device__ __forceinline__ uint32_t at(int p) const
    {
        const uint32_t upto = p == 31 ? 0xffffffffu : (2u << p) - 1u;
        const int s = __popc(starts & upto), e = __popc(ends & upto);
        const int inside = (inside0 ? 1 : 0) + s - e;
        return inside > 0 ? r0 + (uint32_t)s - (inside0 ? 0u : 1u) : MINI_ROW_NONE;
    }
};

// The segmentation in two steps that the plan kernel and the scatter kernel share (they see the same records by construction):
//   mini_minimizers   per_pos(p, mv) for p = 0 .. 31 (p a compile-time constant at every call): mv = the minimizer value of the
//                     k-mer that ends at character p of the word;
//   mini_record_ends  from the lane's validity masks and the mask of positions whose minimizer value equals their predecessor's:
//                     the positions where a record ENDS.  Pure bit arithmetic on 32-bit masks: until round 4 the record logic
//                     ran position by position (a counter, four comparisons, a predicated block per position that the whole
//                     wavefront issued although a word has six or seven records).
// A record's bucket is mini_bucket(mv of its last k-mer) -- all its k-mers share the value --, its length and row follow from where
// it ends (the callers keep the bucket of EVERY position in an LDS column of the lane and walk over the set bits of the end mask).
// DELAY (k > 21): the minimizer of a k-mer is taken over its central W M-mers only (pg_device.hpp: mini_window), i.e. the
// window that ends `off` characters before the k-mer does -- the window minimum passes through a delay line of `off` <= 5 steps.
constexpr int MINI_MAX_OFF = 5;
template <int W, bool DELAY, int M, class PerPos>
__device__ __forceinline__ void mini_minimizers(const Word &x, int off, PerPos &&per_pos)
{
    constexpr uint32_t MMASK = (1u << (2 * M)) - 1u;
    uint32_t win[W];                                            // win[0] = newest hashed canonical M-mer
    uint32_t dl[DELAY ? MINI_MAX_OFF : 1];                      // dl[i] = window minimum i + 1 characters ago
#pragma unroll
    for (int i = 0; i < W; ++i) win[i] = 0xffffffffu;
#pragma unroll
    for (int i = 0; i < (DELAY ? MINI_MAX_OFF : 1); ++i) dl[i] = 0xffffffffu;
    // The M-mer that ends at character q of the 64 characters (pw, cw) is cut straight out of them -- its reverse complement out of
    // the words as they are (oldest character lowest, complemented: ^ 2 per character), the M-mer itself out of the character-
    // reversed words (newest character lowest) --, at offsets that are constants once the loops below are unrolled: one funnel
    // shift + one mask each, no rolling state, no pre-roll of the characters in front of the first wanted M-mer.  (The values
    // are those of the rolling forms fwm = ((fwm << 2) | ch) & MMASK, rcm = (rcm >> 2) | ((ch ^ 2) << 2 (M - 1)).)
    const uint64_t rlo = rev2_64(x.cw), rhi = rev2_64(x.pw);
    auto cut = [](uint64_t lo, uint64_t hi, int b) -> uint32_t {                  // 32 bits from bit b of (hi:lo)
        return b == 0 ? (uint32_t)lo : b < 64 ? (uint32_t)((lo >> b) | (hi << (64 - b))) : (uint32_t)(hi >> (b - 64));
    };
    constexpr uint32_t COMPL = 0xAAAAAAAAu & MMASK;
    constexpr int PRE = W - 1 + (DELAY ? MINI_MAX_OFF : 0);     // M-mers in front of the word that are wanted in the window
    static_assert(32 - PRE >= M - 1, "the first wanted M-mer lies inside the previous word");
    auto step = [&](int q) -> uint32_t {                         // the minimizer value of the k-mer that ends at character q
        const uint32_t fwm = cut(rlo, rhi, 2 * (63 - q)) & MMASK;
        const uint32_t rcm = (cut(x.pw, x.cw, 2 * (q - (M - 1))) & MMASK) ^ COMPL;
#pragma unroll
        for (int i = W - 1; i > 0; --i) win[i] = win[i - 1];
        win[0] = mhash(fwm < rcm ? fwm : rcm);
        uint32_t mv = win[0];
#pragma unroll
        for (int i = 1; i < W; ++i) mv = win[i] < mv ? win[i] : mv;
        if (!DELAY) return mv;
        uint32_t use = mv;                                      // (off is wave-uniform: scalar selects)
#pragma unroll
        for (int i = 0; i < MINI_MAX_OFF; ++i) use = off == i + 1 ? dl[i] : use;
#pragma unroll
        for (int i = MINI_MAX_OFF - 1; i > 0; --i) dl[i] = dl[i - 1];
        dl[0] = mv;
        return use;
    };
#pragma unroll
    for (int c = 32 - PRE; c < 32; ++c) step(c);
#pragma unroll
    for (int p = 0; p < 32; ++p) per_pos(p, step(32 + p));
}

// bit p of the result: bits p - n + 1 .. p of m are all set (1 <= n <= 32)
__device__ __forceinline__ uint32_t runs32(uint32_t m, int n)
{
    uint32_t r = m;
    int len = 1;
    while (2 * len <= n) { r &= r << len; len *= 2; }
    if (len < n) r &= r << (n - len);
    return r;
}

// Where the records of a word end.  ok / ok_row: positions that end a valid k-mer (for the table / for the rows), cuts: positions
// where the row changes, eq: positions whose minimizer value equals that of the position before, cap: k-mers per record at most.
// A k-mer CONTINUES the open record if it and its predecessor are valid and of the same kind (row-counting or not), no row boundary
// lies at its last character and the minimizer value is the same (records never cross words: position 0 continues nothing) -- and
// if the record has room: of `cap` continuations in a row the last one starts a new record instead (lowest first, as a counter
// running over the positions would have it; rare -- a minimizer covers at most W = cap k-mers unless the same hashed M-mer recurs).
__device__ __forceinline__ uint32_t mini_record_ends(uint32_t ok, uint32_t ok_row, uint32_t cuts, uint32_t eq, int cap)
{
    uint32_t cont = ok & (ok << 1) & ~(ok_row ^ (ok_row << 1)) & ~cuts & eq;
    if (cap <= 1) cont = 0;
    else {
        uint32_t over = runs32(cont, cap);
        while (over) {                                          // (cont has a run of `cap` bits: the record would hold cap + 1 k-mers)
            cont &= ~(over & (0u - over));
            over = runs32(cont, cap);
        }
    }
    return ok & ~(cont >> 1);                                   // a record ends where the next position does not continue it
}

// what a lane needs to segment word w: the word, its valid k-mer ends under the counting rule and under the rows' strict rule
struct LaneWord {
    Word x;
    uint32_t ok, ok_row;
};
__device__ __forceinline__ LaneWord load_lane_word(const uint64_t *__restrict__ codes, const uint32_t *__restrict__ valid,
                                                   const uint32_t *__restrict__ strict, int64_t w, int k)
{
    LaneWord lw;
    lw.x = load_word(codes, valid, w, k);
    lw.ok = lw.x.ok;
    lw.ok_row = lw.ok;
    if (strict && lw.ok) {
        const uint32_t sv = strict[w], sp = w > 0 ? strict[w - 1] : 0u;
        lw.ok_row &= (uint32_t)(runs_of(((uint64_t)sv << 32) | sp, k) >> 32);
    }
    return lw;
}

// ---- plan: per chunk, records per first-pass region (chunk_hist[d * n_chunks + slot(chunk)]).  Nothing else: the per-bucket
// counts are taken from the records themselves once the first pass has written them (mini_bucket_hist_kernel).  With 2 KB of
// LDS, 256 threads and ~48 registers this kernel fits beside the workgroups of the other kernels on a CU -- beside the count
// kernel, which fills LDS but leaves half of the VALU issue slots and a third of the registers idle: KmerTable.prefetch_plan
// runs the next batch's plan on a side stream UNDER this batch's second scatter pass and count instead of in front of them
template <int W, bool DELAY, int M>
__global__ __launch_bounds__(BLOCK) void mini_plan_kernel(const uint64_t *__restrict__ codes, const uint32_t *__restrict__ valid,
                                                          int64_t word_begin, int64_t word_end, int k, int woff, int bits, int bits2, int cap,
                                                          const int64_t *__restrict__ row_start, const int64_t *__restrict__ row_end, int64_t n_rows,
                                                          const uint32_t *__restrict__ strict, const int32_t *__restrict__ round_row,
                                                          unsigned long long *__restrict__ chunk_hist, int64_t n_chunks, int64_t chunk_stride,
                                                          unsigned long long *__restrict__ class_totals)
{
    __shared__ uint32_t coarse[1 << MINI_MAX_BITS1];
    __shared__ uint32_t n_long_here;                             // records of more than SHORT_MAX k-mers in this chunk
    __shared__ uint16_t col[32 * BLOCK];                         // the bucket of every position of the lane's word (a column per lane)
    const int n_dig = 1 << (bits - bits2);
    for (int64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        for (int i = threadIdx.x; i < (1 << MINI_MAX_BITS1); i += BLOCK) coarse[i] = 0;
        if (threadIdx.x == 0) n_long_here = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < MINI_CHUNK_WORDS; i += BLOCK) {
            const int64_t wi = chunk * MINI_CHUNK_WORDS + i;        // word index inside the range
            const int64_t w = word_begin + wi;
            if (w >= word_end) continue;
            const LaneWord lw = load_lane_word(codes, valid, strict, w, k);
            if (lw.ok == 0) continue;
            RowBits rb;
            rb.init(row_start, row_end, n_rows, row_start ? round_row[wi / ROUND_WORDS] : 0, w << 5);
            uint32_t eq = 0, prev = 0;
            mini_minimizers<W, DELAY, M>(lw.x, woff, [&](int p, uint32_t mv) {
                eq |= mv == prev ? 1u << p : 0u;
                prev = mv;
                col[p * BLOCK + threadIdx.x] = (uint16_t)mini_bucket(mv, bits);
            });
            const uint32_t has = mini_record_ends(lw.ok, lw.ok_row, rb.cuts(), eq, cap);
            // (the column is the lane's own: no barrier between its writes and these reads)
            for (uint32_t m = has; m; m &= m - 1u)
                atomicAdd(&coarse[(uint32_t)col[(uint32_t)__builtin_ctz(m) * BLOCK + threadIdx.x] >> bits2], 1u);
            // records of more than SHORT_MAX k-mers: ends with SHORT_MAX positions in front of them that neither end a record nor
            // fail to end a k-mer
            const uint32_t nonk = has | ~lw.ok;
            uint32_t lg = has & ~((1u << PG_SHORT_MAX_) - 1u);
#pragma unroll
            for (int q = 1; q <= PG_SHORT_MAX_; ++q) lg &= ~(nonk << q);
            if (lg) atomicAdd(&n_long_here, (uint32_t)__popc(lg));
        }
        __syncthreads();
        if (threadIdx.x == 0 && n_long_here) atomicAdd(class_totals, (unsigned long long)n_long_here);     // header[2]: long records of the stream
        for (int d = threadIdx.x; d < n_dig; d += BLOCK)
            chunk_hist[(int64_t)d * n_chunks + (int64_t)(((__int128)chunk * chunk_stride) % n_chunks)] = coarse[d];
        __syncthreads();
    }
}

// row sums of table[d][0..n): one workgroup per digit
__global__ __launch_bounds__(BIG_BLOCK) void digit_totals_kernel(const unsigned long long *__restrict__ table, int64_t n,
                                                                 unsigned long long *__restrict__ totals)
{
    __shared__ unsigned long long part[BIG_BLOCK / 64];
    const unsigned long long *row = table + (int64_t)blockIdx.x * n;
    unsigned long long s = 0;
    for (int64_t i = threadIdx.x; i < n; i += BIG_BLOCK) s += row[i];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long run = 0;
        for (int i = 0; i < BIG_BLOCK / 64; ++i) run += part[i];
        totals[blockIdx.x] = run;
    }
}

// records per bucket, from the records the first pass has written: region = blockIdx.x / tiles_x holds [region_off[region],
// region_off[region + 1]); its workgroups count the second-pass digits of their share of the meta plane and add them to hist
__global__ __launch_bounds__(BLOCK) void mini_bucket_hist_kernel(const uint32_t *__restrict__ meta, const unsigned long long *__restrict__ region_off,
                                                                 int bits2, int tiles_x, unsigned long long *__restrict__ hist,
                                                                 const unsigned long long *__restrict__ header, unsigned long long rec_cap, const uint32_t *status)
{
    __shared__ uint32_t cnt[1 << META_D2_BITS];
    if (header[0] > rec_cap || (*status & PG_STATUS_PLAN_MISMATCH)) return;      // (a plan of another stream: see mini_count_kernel)
    const int n_dig = 1 << bits2;
    const uint32_t dmask = (uint32_t)n_dig - 1u;
    const int64_t region = blockIdx.x / tiles_x;
    const int64_t r0 = (int64_t)region_off[region], r1 = (int64_t)region_off[region + 1];
    if ((int)threadIdx.x < n_dig) cnt[threadIdx.x] = 0;
    __syncthreads();
    for (int64_t i0 = r0 + (int64_t)(blockIdx.x % tiles_x) * (8 * BLOCK); i0 < r1; i0 += (int64_t)tiles_x * (8 * BLOCK)) {
        uint32_t m[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t i = i0 + j * BLOCK + threadIdx.x;
            m[j] = i < r1 ? meta[i] : 0xffffffffu;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (i0 + j * BLOCK + (int64_t)threadIdx.x < r1) atomicAdd(&cnt[m[j] & dmask], 1u);
    }
    __syncthreads();
    if ((int)threadIdx.x < n_dig && cnt[threadIdx.x]) atomicAdd(&hist[(region << bits2) + threadIdx.x], (unsigned long long)cnt[threadIdx.x]);
}

// total records = off[nb] -> header[0]
__global__ void mini_total_kernel(const unsigned long long *__restrict__ off, int nb, unsigned long long *__restrict__ header)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) header[0] = off[nb];
}

// exclusive scan of cnt[0..N) -> start[0..N] by the first N / 64 wavefronts (N = 128 or 256); every lane of the workgroup calls
template <int N, bool LDS_ONLY = false>
__device__ __forceinline__ void scan_digits(const uint32_t *cnt, uint32_t *start, uint32_t *wave_tot, uint32_t *start_copy = nullptr)
{
    auto sync = [] { if (LDS_ONLY) lds_sync(); else __syncthreads(); };
    uint32_t v = 0, incl = 0;
    if (threadIdx.x < N) {
        v = cnt[threadIdx.x];
        incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(incl, d);
            if ((int)(threadIdx.x & 63) >= d) incl += o;
        }
        if ((threadIdx.x & 63) == 63) wave_tot[threadIdx.x >> 6] = incl;
    }
    sync();
    if (threadIdx.x < N) {
        uint32_t before = 0;
        for (int wv = 0; wv < (int)(threadIdx.x >> 6); ++wv) before += wave_tot[wv];
        start[threadIdx.x] = before + incl - v;
        if (start_copy) start_copy[threadIdx.x] = before + incl - v;
        if (threadIdx.x == N - 1) start[N] = before + incl;
    }
    sync();
}

// the same for N digits on BLK <= N threads: thread t owns the digits [t * N / BLK, (t + 1) * N / BLK)
template <int N, int BLK, bool LDS_ONLY = false>
__device__ __forceinline__ void scan_digits_blk(const uint32_t *cnt, uint32_t *start, uint32_t *wave_tot)
{
    constexpr int PER = N / BLK;
    static_assert(PER >= 1 && PER * BLK == N, "whole digits per thread");
    auto sync = [] { if (LDS_ONLY) lds_sync(); else __syncthreads(); };
    uint32_t v[PER], sum = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) { v[q] = cnt[threadIdx.x * PER + q]; sum += v[q]; }
    uint32_t incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d);
        if ((int)(threadIdx.x & 63) >= d) incl += o;
    }
    if ((threadIdx.x & 63) == 63) wave_tot[threadIdx.x >> 6] = incl;
    sync();
    uint32_t base = incl - sum;
    for (int wv = 0; wv < (int)(threadIdx.x >> 6); ++wv) base += wave_tot[wv];
#pragma unroll
    for (int q = 0; q < PER; ++q) { start[threadIdx.x * PER + q] = base; base += v[q]; }
    if (threadIdx.x == BLK - 1) start[N] = base;
    sync();
}

// ---- A1': stream -> regions
// What a round stages in LDS is not the records but a 4-byte REFERENCE per record, region-sorted -- which lane's word it was cut
// from, where it ends, its bucket -- next to the lanes' words and masks; the record itself (32 characters, length, row) is put
// together by the copy-out thread that stores it.  The lane that segmented the word would have to do that in 32 predicated
// blocks, one per position a record may end at, every one of them issued for the whole wavefront although a word has six or
// seven records: the assembly cost 32 x ~30 vector instructions per word, in the copy-out it costs ~40 per RECORD at full lane
// occupancy (the first pass: 6.9 -> see DESIGN.md), and the stage shrinks from 13 to 4 bytes per record.
constexpr int REF_E_SHIFT = 16, REF_LANE_SHIFT = 21;               // ref = lane << 21 | e << 16 | bucket
static_assert(PG_MINI_MAX_LOG2_BUCKETS <= REF_E_SHIFT && S1_BLOCK <= (1 << (32 - REF_LANE_SHIFT)), "fields of a stage reference");
template <int N1> struct Scatter1Lds {                              // N1 regions: 256 (until round 4 also 512, for tables of 2^16 buckets: see Scatter2Lds)
    uint32_t ref[STAGE_CAP];
    uint16_t col[32 * S1_BLOCK];                                    // the bucket of every position of the lanes' words (a column per lane)
    uint64_t cw[S1_BLOCK], pw[S1_BLOCK];                            // the lanes' words and the words before them
    uint32_t nonk[S1_BLOCK];                                        // positions that end a record or end no k-mer at all
    uint32_t ok_row[S1_BLOCK];                                      // k-mer ends that count for rows
    uint32_t starts[S1_BLOCK], ends[S1_BLOCK], r0[S1_BLOCK];        // RowBits of the lane (r0: bit 31 = inside0)
    uint32_t cnt[N1];
    uint32_t start[N1 + 1];
    uint32_t fill[N1];                                              // start[], counted up as the references are placed
    // (512 regions: the array would push the structure 40 bytes over half of the CU's 160 KiB -- one workgroup per CU instead of
    // two; there the copy-out derives the value from cur[] and start[])
    unsigned long long gbase[N1 > 256 ? 1 : N1];
    unsigned long long cur[N1];                                     // running write offsets of this chunk, per region
    uint32_t wave_tot[N1 / 64];
};

static_assert(2 * sizeof(Scatter1Lds<256>) <= 160 * 1024, "two first-pass workgroups share a CU's LDS");

template <int W, bool DELAY, int M, int N1>
__global__ __launch_bounds__(S1_BLOCK, 4) void mini_scatter_kernel(const uint64_t *__restrict__ codes, const uint32_t *__restrict__ valid,
                                                                 int64_t word_begin, int64_t word_end, int k, int woff, int bits, int bits2, int cap,
                                                                 const int64_t *__restrict__ row_start, const int64_t *__restrict__ row_end, int64_t n_rows,
                                                                 const uint32_t *__restrict__ strict, const int32_t *__restrict__ round_row,
                                                                 uint64_t *__restrict__ out_bases, uint32_t *__restrict__ out_meta,
                                                                 const unsigned long long *__restrict__ chunk_off, int64_t n_chunks, int64_t chunk_stride,
                                                                 const unsigned long long *__restrict__ header, const unsigned long long *__restrict__ region_off,
                                                                 unsigned long long rec_cap, uint32_t *status)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    static_assert(N1 <= S1_BLOCK, "one lane per region");
    if (header[0] > rec_cap) {                                      // (a plan of another stream: see mini_count_kernel)
        if (threadIdx.x == 0 && blockIdx.x == 0) atomicOr(status, PG_STATUS_PLAN_MISMATCH);
        return;
    }
    Scatter1Lds<N1> &L = *reinterpret_cast<Scatter1Lds<N1> *>(lds_raw);
    const int n_dig = 1 << (bits - bits2);
    const uint32_t d2mask = (1u << bits2) - 1u;
    const int64_t chunk = blockIdx.x;
    const int64_t slot = (int64_t)(((__int128)chunk * chunk_stride) % n_chunks);
    if (threadIdx.x < N1) L.cur[threadIdx.x] = (int)threadIdx.x < n_dig ? chunk_off[(int64_t)threadIdx.x * n_chunks + slot] : 0ull;
    for (int rd = 0; rd < ROUNDS_PER_CHUNK; ++rd) {
        const int64_t round = chunk * ROUNDS_PER_CHUNK + rd;
        const int64_t w = word_begin + round * ROUND_WORDS + threadIdx.x;
        if (word_begin + round * ROUND_WORDS >= word_end) break;    // (uniform)
        if (threadIdx.x < N1) L.cnt[threadIdx.x] = 0;
        lds_sync();                                                 // (also: the previous round's copy-out has read the lanes' words)
        // The lane's word -> the bucket of every position (an LDS column of the lane's own) and the mask of the positions where a
        // record ends; then the records are COUNTED per region, one LDS add without return per set bit (nothing waits for it).
        uint32_t has = 0;
        LaneWord lw;
        lw.ok = 0; lw.ok_row = 0; lw.x.cw = lw.x.pw = 0;
        if (w < word_end) lw = load_lane_word(codes, valid, strict, w, k);
        RowBits rb;
        rb.starts = rb.ends = rb.r0 = 0; rb.inside0 = false;
        uint16_t *const mycol = L.col + threadIdx.x;
        if (lw.ok) {
            rb.init(row_start, row_end, n_rows, row_start ? round_row[round] : 0, w << 5);
            uint32_t eq = 0, prev = 0;
            mini_minimizers<W, DELAY, M>(lw.x, woff, [&](int p, uint32_t mv) {
                eq |= mv == prev ? 1u << p : 0u;
                prev = mv;
                mycol[p * S1_BLOCK] = (uint16_t)mini_bucket(mv, bits);
            });
            has = mini_record_ends(lw.ok, lw.ok_row, rb.cuts(), eq, cap);
            for (uint32_t m = has; m; m &= m - 1u)
                __hip_atomic_fetch_add(&L.cnt[(uint32_t)mycol[(uint32_t)__builtin_ctz(m) * S1_BLOCK] >> bits2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        // what the copy-out needs of this lane's word: a record's length and row follow from where it ends -- it starts behind the
        // previous record's end or behind the last position that ends no valid k-mer, whichever is later; its k-mers share one
        // row (that of its last character)
        L.cw[threadIdx.x] = lw.x.cw;
        L.pw[threadIdx.x] = lw.x.pw;
        L.nonk[threadIdx.x] = has | ~lw.ok;
        L.ok_row[threadIdx.x] = lw.ok_row;
        L.starts[threadIdx.x] = rb.starts;
        L.ends[threadIdx.x] = rb.ends;
        L.r0[threadIdx.x] = rb.r0 | (rb.inside0 ? 0x80000000u : 0u);
        lds_sync();
        scan_digits<N1, true>(L.cnt, L.start, L.wave_tot, L.fill);
        const uint32_t total = L.start[N1];
        // a round has at most 32 x 512 records; 8 positions x 512 lanes always fit the stage.  Nearly every round fits whole.
        const int n_win = total <= (uint32_t)STAGE_CAP ? 1 : 4;
        for (int win = 0; win < n_win; ++win) {
            const uint32_t wmask = n_win == 1 ? 0xffffffffu : (0xffu << (8 * win));
            if (n_win > 1) {                                        // count again, this window's records only
                lds_sync();
                if (threadIdx.x < N1) L.cnt[threadIdx.x] = 0;
                lds_sync();
                for (uint32_t m = has & wmask; m; m &= m - 1u)
                    __hip_atomic_fetch_add(&L.cnt[(uint32_t)mycol[(uint32_t)__builtin_ctz(m) * S1_BLOCK] >> bits2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                lds_sync();
                scan_digits<N1, true>(L.cnt, L.start, L.wave_tot, L.fill);
            }
            // gbase[d] = (where the region's run goes) - (where it starts in the stage): the copy-out adds the stage position
            if (threadIdx.x < N1) {
                const uint32_t c = L.start[threadIdx.x + 1] - L.start[threadIdx.x];
                if constexpr (N1 <= 256) L.gbase[threadIdx.x] = L.cur[threadIdx.x] - L.start[threadIdx.x];
                L.cur[threadIdx.x] += c;                            // (= gbase + start[d + 1])
            }
            {
                // the lane's records, two per turn: their buckets out of the column, their places in the stage from the regions'
                // fill counters (a returning add each, both in flight together), the references written
                const uint32_t mine = (uint32_t)threadIdx.x << REF_LANE_SHIFT;
                uint32_t m = has & wmask;
                while (m) {
                    const uint32_t e0 = (uint32_t)__builtin_ctz(m);
                    m &= m - 1u;
                    const bool two = m != 0u;
                    const uint32_t e1 = two ? (uint32_t)__builtin_ctz(m) : e0;
                    m &= m - 1u;                                    // (0 stays 0)
                    const uint32_t b0 = mycol[e0 * S1_BLOCK], b1 = mycol[e1 * S1_BLOCK];
                    const uint32_t p0 = __hip_atomic_fetch_add(&L.fill[b0 >> bits2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const uint32_t p1 = __hip_atomic_fetch_add(&L.fill[b1 >> bits2], two ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    L.ref[p0] = mine | (e0 << REF_E_SHIFT) | b0;
                    if (two) L.ref[p1] = mine | (e1 << REF_E_SHIFT) | b1;
                }
            }
            lds_sync();
            const uint32_t tot = L.start[N1];
            // copy-out: thread i puts record i of the region-sorted round together and stores it, two records at a time (their LDS
            // reads in flight together)
            for (uint32_t i0 = 0; i0 < tot; i0 += 2 * S1_BLOCK) {
                uint32_t rf[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const uint32_t i = i0 + u * S1_BLOCK + threadIdx.x;
                    rf[u] = L.ref[i < tot ? i : 0u];                 // (tot > 0 here: ref[0] is a record)
                }
                uint64_t cwv[2], pwv[2];
                uint32_t nonk[2], okr[2], st[2], en[2], r0v[2];
                unsigned long long gb[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const uint32_t ln = rf[u] >> REF_LANE_SHIFT;
                    cwv[u] = L.cw[ln]; pwv[u] = L.pw[ln];
                    nonk[u] = L.nonk[ln]; okr[u] = L.ok_row[ln];
                    st[u] = L.starts[ln]; en[u] = L.ends[ln]; r0v[u] = L.r0[ln];
                    const uint32_t d = (rf[u] & 0xffffu) >> bits2;
                    if constexpr (N1 <= 256) gb[u] = L.gbase[d];
                    else gb[u] = L.cur[d] - L.start[d + 1];
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const uint32_t i = i0 + u * S1_BLOCK + threadIdx.x;
                    const uint32_t e = (rf[u] >> REF_E_SHIFT) & 31u, b = rf[u] & 0xffffu;
                    const uint32_t below = nonk[u] & ((1u << e) - 1u);                       // ends / non-k-mers before e
                    const uint32_t n = below ? e - (31u - (uint32_t)__clz((int)below)) : e + 1u;
                    uint32_t row = MINI_ROW_NONE;
                    if ((okr[u] >> e) & 1u) {                                                // (RowBits::at)
                        const uint32_t upto = e == 31u ? 0xffffffffu : (2u << e) - 1u;
                        const int sc = __popc(st[u] & upto), ec = __popc(en[u] & upto);
                        const int in0 = (int)(r0v[u] >> 31);
                        if (in0 + sc - ec > 0) row = (r0v[u] & 0x7fffffffu) + (uint32_t)sc - (in0 ? 0u : 1u);
                    }
                    // the 32 characters ending at position e of the word
                    const uint64_t rec = e == 31u ? cwv[u] : (cwv[u] << (2u * (31u - e))) | (pwv[u] >> (2u * (e + 1u)));
                    const unsigned long long g = gb[u] + i;
                    // (a record beyond the buffers is dropped: cannot happen with the plan of THIS stream -- the offsets are exact --,
                    // and the plan of another one is reported below)
                    if (i < tot && g < rec_cap) {
                        out_bases[g] = rec;
                        out_meta[g] = (row << META_ROW_SHIFT) | ((n - 1u) << META_D2_BITS) | (b & d2mask);
                    }
                }
            }
        }
    }
    // every region's run of this chunk must end where the plan put the next chunk's: a plan that was computed for other reads
    // (KmerTable ties its cached plans to the stream; this is the backstop) shows here, and the count is void
    if ((int)threadIdx.x < n_dig) {
        const unsigned long long planned = slot + 1 < n_chunks ? chunk_off[(int64_t)threadIdx.x * n_chunks + slot + 1] : region_off[threadIdx.x + 1];
        if (L.cur[threadIdx.x] != planned) atomicOr(status, PG_STATUS_PLAN_MISMATCH);
    }
}

// ---- A2': region -> buckets.  tiles_x workgroups per region (all of them on one XCD, see below), its records are
// [off[region << bits2], off[(region + 1) << bits2]);
// digit = the low bits2 bits of meta (+ 128 for a LONG record: more than short_max k-mers).  Inside its bucket's range
// [off[b], off[b + 1]) the short records fill from the front (cursor[b]) and the long ones from the back (cursor_l[b]): the
// bucket workgroups treat a record in as many unrolled steps as its class can have k-mers, so sorting the two classes apart
// saves them the steps that short records would leave idle.  cursor[b] is the number of short records afterwards.
constexpr int S2_RPL = 16;
constexpr int S2_TILE = BLOCK * S2_RPL;
// DIG digits: 2 classes x 128 buckets of a region (tables of up to 2^15 buckets), or x 256 (2^16 buckets: the geometry of several
// ranks -- until round 4 those tables had 512 first-pass regions instead, whose runs of six records per round cost that pass 1.6 ms)
template <int DIG> struct Scatter2Lds {
    uint64_t bases[S2_TILE];
    uint32_t meta[S2_TILE];
    uint32_t cnt[DIG];
    uint32_t start[DIG + 1];
    unsigned long long gbase[DIG];
    uint32_t wave_tot[WAVES];
};
template <int DIG>
__global__ __launch_bounds__(BLOCK) void mini_scatter2_kernel(const uint64_t *__restrict__ in_bases, const uint32_t *__restrict__ in_meta,
                                                              const unsigned long long *__restrict__ off, int bits2, int tiles_x, int short_max,
                                                              uint64_t *__restrict__ out_bases, uint32_t *__restrict__ out_meta,
                                                              unsigned long long *__restrict__ cursor, unsigned long long *__restrict__ cursor_l,
                                                              unsigned long long *__restrict__ kwords,
                                                              const unsigned long long *__restrict__ header, unsigned long long rec_cap, uint32_t *status)
{
    if (header[0] > rec_cap || (*status & PG_STATUS_PLAN_MISMATCH)) return;      // (a plan of another stream: see mini_count_kernel)
    // kwords[b] += the k-mers of bucket b that lie inside a row (= the words its workgroup will emit): they ride in the high
    // half of the tile's rank counters -- a tile has 4096 records of at most 9 k-mers, both halves stay below 2^16
    static_assert(S2_TILE * MINI_MAX_WINDOW < 65536, "two 16-bit halves per rank counter");
    __shared__ Scatter2Lds<DIG> L;
    constexpr int DPT = DIG / BLOCK;                            // digits per lane: threadIdx.x * DPT + q
    constexpr uint32_t CLS = DIG / 2;                           // the class bit of a digit
    const int n_dig = 1 << bits2;
    const uint32_t dmask = (uint32_t)n_dig - 1u;
    // Workgroups go to the eight XCDs round-robin by their index, and each XCD has its own L2.  All workgroups of a region are put
    // on ONE XCD (region = index mod 8 + ...): the runs that the tiles of a region append to a bucket lie next to each other, so
    // the partial lines at their ends are completed in that L2 instead of being written back twice, half-filled, by two of them
    // (second pass 4.38 -> 2.97 ms on one box; all its stores left out: 1.60 ms, stored in tile order instead: 2.60 ms).
    const bool by_xcd = gridDim.x / (unsigned)tiles_x >= 8u;        // (a power of two of regions)
    const int64_t region = by_xcd ? (int64_t)(blockIdx.x % 8u) + 8 * (int64_t)((blockIdx.x / 8u) / (unsigned)tiles_x) : (int64_t)(blockIdx.x / tiles_x);
    const int64_t b0 = region << bits2;
    const int64_t r0 = (int64_t)off[b0], r1 = (int64_t)off[b0 + n_dig];
    const int64_t n_tiles = (r1 - r0 + S2_TILE - 1) / S2_TILE;
    auto digit_of = [&](uint32_t m) -> uint32_t {
        const int n = (int)((m >> META_D2_BITS) & (MINI_MAX_LEN - 1)) + 1;
        return (m & dmask) | (n > short_max ? CLS : 0u);
    };
    for (int64_t tile = by_xcd ? (blockIdx.x / 8u) % (unsigned)tiles_x : blockIdx.x % tiles_x; tile < n_tiles; tile += tiles_x) {
#pragma unroll
        for (int q = 0; q < DPT; ++q) L.cnt[threadIdx.x * DPT + q] = 0;
        lds_sync();
        const int64_t t0 = r0 + tile * S2_TILE;
        uint64_t rb[S2_RPL];
        uint32_t rm[S2_RPL], dr[S2_RPL];
#pragma unroll
        for (int j = 0; j < S2_RPL; ++j) {                          // all loads of the lane in flight together
            const int64_t i = t0 + j * BLOCK + threadIdx.x;
            rb[j] = i < r1 ? in_bases[i] : 0ull;
            rm[j] = i < r1 ? in_meta[i] : 0u;
        }
#pragma unroll
        for (int j = 0; j < S2_RPL; ++j) {
            const int64_t i = t0 + j * BLOCK + threadIdx.x;
            if (i < r1) {
                const uint32_t d = digit_of(rm[j]);
                // (kwords == NULL: the merged lookups do not need the tally)
                const uint32_t kw = kwords && (rm[j] >> META_ROW_SHIFT) != MINI_ROW_NONE ? ((rm[j] >> META_D2_BITS) & (MINI_MAX_LEN - 1)) + 1u : 0u;
                dr[j] = (d << 16) | (atomicAdd(&L.cnt[d], 1u | (kw << 16)) & 0xffffu);
            }
        }
        lds_sync();
        uint32_t kw_mine[DPT];                                      // (this lane's digits; the scan reads the same entries next)
#pragma unroll
        for (int q = 0; q < DPT; ++q) { kw_mine[q] = L.cnt[threadIdx.x * DPT + q] >> 16; L.cnt[threadIdx.x * DPT + q] &= 0xffffu; }
        if constexpr (DPT == 1) scan_digits<DIG, true>(L.cnt, L.start, L.wave_tot);
        else { lds_sync(); scan_digits_blk<DIG, BLOCK, true>(L.cnt, L.start, L.wave_tot); }
        // the returning cursor adds (one per digit and tile) are issued first and consumed after the placement
        unsigned long long gpos[DPT];
#pragma unroll
        for (int q = 0; q < DPT; ++q) {
            const uint32_t d = threadIdx.x * DPT + q;
            const uint32_t c = L.start[d + 1] - L.start[d];
            gpos[q] = 0;
            if (kw_mine[q]) atomicAdd(&kwords[b0 + (d & (CLS - 1u))], (unsigned long long)kw_mine[q]);
            if (c) {
                const int64_t b = b0 + (d & (CLS - 1u));
                gpos[q] = d < CLS ? off[b] + atomicAdd(&cursor[b], (unsigned long long)c) - L.start[d]
                                  : off[b + 1] - (atomicAdd(&cursor_l[b], (unsigned long long)c) + c) - L.start[d];
            }
        }
#pragma unroll
        for (int j = 0; j < S2_RPL; ++j) {
            const int64_t i = t0 + j * BLOCK + threadIdx.x;
            if (i < r1) {
                const uint32_t at = L.start[dr[j] >> 16] + (dr[j] & 0xffffu);
                L.bases[at] = rb[j];
                L.meta[at] = rm[j];
            }
        }
#pragma unroll
        for (int q = 0; q < DPT; ++q) L.gbase[threadIdx.x * DPT + q] = gpos[q];
        lds_sync();
        const uint32_t total = L.start[DIG];
        for (uint32_t i = threadIdx.x; i < total; i += BLOCK) {     // flat sweep: the digit is in the record
            const uint32_t m = L.meta[i];
            const unsigned long long g = L.gbase[digit_of(m)] + i;
            gstore(out_bases, g, rec_cap, L.bases[i], status);
            gstore(out_meta, g, rec_cap, m, status);
        }
        lds_sync();
    }
}

// ---- B' + S1': one workgroup per bucket
//
// Count pass.  A lane owns one record per batch and treats its (up to CAP) k-mers in straight-line, fully unrolled code: all
// codes, all home slots and all first LDS probes of the record are issued before any is resolved (CAP independent LDS reads in
// flight per lane; the slots beyond the record's length are predicated off).  What the first probe does not settle -- a k-mer
// seen for the first time, a collision -- goes onto a small per-wavefront ring in LDS and is worked off 64 at a time by the
// general probing loop, so that loop always runs with every lane busy.
// Lookups.  An occurrence that lies in a row needs the FINAL count of its k-mer, which exists only when the bucket has been
// counted.  SLOTS form (rows < 2^(32 - log2 bucket slots)): the count pass leaves a provisional word (row, slot index) per
// occurrence -- it knows the slot when the insert is done -- and once the counts have become bins in place, the provisional
// words are streamed back (they are fresh in L2 / MALL) and turned into (row, bin) words by ONE LDS read each: no second
// derivation of the k-mers, no second probing.  General form: the records are read and probed a second time.
constexpr uint32_t RING_NOPLACE = 0xffff0000u;              // (MERGE: a ring entry of a k-mer outside every row; places stay below it)
constexpr int RING = 128;                                    // entries per wavefront ring: < 64 waiting + <= 64 pushed per step
constexpr int COUNT_WAVES = BIG_BLOCK / 64;
constexpr uint32_t BIN_NONE = (uint32_t)HASH_CMASK;          // a slot's count field after the counts have become bins: bin + 1, or this

// a value that is the same in every lane (read from LDS, say), moved to scalar registers: addresses built from it are a scalar
// base + a 32-bit lane offset, comparisons with it are scalar operands -- as vector registers such bases were spilled and
// reloaded in front of every load of an unrolled group, each reload with a wait for ALL memory operations in flight
__device__ __forceinline__ uint32_t uniform32(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
template <class T> __device__ __forceinline__ T *uniform_ptr(T *p)
{
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    return reinterpret_cast<T *>(((uint64_t)uniform32((uint32_t)(v >> 32)) << 32) | uniform32((uint32_t)v));
}
__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ uint32_t lanes_below(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// the general insert; returns the slot the k-mer lives in, or 0xffffffff when the bucket is full (inactive lanes: 0).
// Written as ONE wave-uniform loop with two short predicated regions per round (claim an empty slot / add to a match): the
// structurizer's rendering of the obvious per-lane loop with early returns spends several times as many scalar
// instructions on mask bookkeeping, and those -- not the LDS -- were the cost of this path.
// WIDE (k > 21): tab[] holds code + 1 (0 = empty) and the counts live in their own 32-bit plane cnts[].
template <bool WIDE>
__device__ __forceinline__ uint32_t mini_insert_slow(unsigned long long *tab, uint32_t *cnts, uint32_t smask, uint32_t limit, uint64_t code, bool act, uint32_t skip = 0u)
{
    // (skip = 1: the first probe has seen the home slot taken by ANOTHER key -- nothing ever leaves a slot, so the search starts
    // behind it: one dependent LDS round trip less for most of what comes off the ring)
    uint32_t s = (mini_slot_hash<WIDE>(code) + skip) & smask;
    uint32_t res = act ? 0xffffffffu : 0u;
    bool todo = act;
    const unsigned long long fresh = WIDE ? (unsigned long long)(code + 1ull) : (unsigned long long)((code << HASH_CBITS) | 1ull);
    // one slot of the chain: claim it if it is empty, count on it if it holds the key; true when the lane is settled
    auto visit = [&](uint32_t at, unsigned long long cur) {
        if (todo && cur == 0) cur = atomicCAS(&tab[at], 0ull, fresh);
        const bool claimed = todo && cur == 0;
        if (WIDE) {
            const bool match = todo && cur == fresh;
            if (claimed || match) { atomicAdd(&cnts[at], 1u); res = at; todo = false; }
        } else {
            const bool match = todo && cur != 0 && (cur >> HASH_CBITS) == code;
            // stop growing at SAT; the overshoot is bounded by the lanes in flight and clamped when the slice is packed
            if (match && (uint32_t)(cur & HASH_CMASK) < HASH_SAT) atomicAdd(reinterpret_cast<uint32_t *>(&tab[at]), 1u);   // (the count is in the low dword)
            if (claimed || match) { res = at; todo = false; }
        }
    };
    // TWO slots per turn, both reads in flight together: a turn is a dependent LDS round trip (two where a slot is claimed), and
    // the round lasts as long as its longest chain.  The second slot is only visited by lanes whose first one holds another key
    // (what was read from it may be stale by then: a claim goes through compare-and-swap, which answers with what is there).
    for (uint32_t i = 0; i < limit; i += 2) {
        if (!__any(todo)) break;
        const uint32_t s1 = (s + 1) & smask;
        const unsigned long long cur0 = tab[s], cur1 = tab[s1];  // (settled lanes read their last slots again: harmless)
        visit(s, cur0);
        visit(s1, cur1);
        s = todo ? (s + 2) & smask : s;
    }
    return res;
}

// the general lookup: count field of `code` (BIN_NONE if absent)
template <bool WIDE>
__device__ __forceinline__ uint32_t mini_lookup_slow(const unsigned long long *tab, const uint32_t *cnts, uint32_t smask, uint32_t limit, uint64_t code, bool act)
{
    uint32_t s = mini_slot_hash<WIDE>(code) & smask;
    uint32_t res = BIN_NONE;
    bool todo = act;
    for (uint32_t q = 0; q < limit; ++q) {
        if (!__any(todo)) break;
        const unsigned long long cur = tab[s];
        const bool match = todo && (WIDE ? cur == code + 1ull : cur != 0 && (cur >> HASH_CBITS) == code);
        if (match) res = WIDE ? cnts[s] : (uint32_t)(cur & HASH_CMASK);
        if (match || cur == 0) todo = false;
        s = todo ? (s + 1) & smask : s;
    }
    return res;
}

#ifndef PG_SHORT_MAX
#define PG_SHORT_MAX 4
#endif
// where the SLOTS form scatters its (row, bin) words: straight into the row shuffle's group regions (pg_shuffle_ctx)
struct ShufArgs {
    const unsigned long long *goff;
    unsigned long long *gcur1;
    uint32_t *words_out;
    int gb1, gb2, dshift;
    int narrow;                                                  // one-pass shuffle: 2-byte words (row inside its group, bin)
    unsigned long long words_cap;                                // 4-byte elements of words_out (checked builds)
};
// LDS of that scatter (bytes from the start of the dynamic area; the table has become 2-byte bins by then)
template <int BLK, int DIG> struct LookupLds {                 // BLK threads: 1024, or 512 (buckets of at most 2^13 slots: two workgroups per CU);
    static constexpr uint32_t TILE = 16 * BLK;                  // DIG row-group digits: 1024, or 2048 (65 537 .. 131 072 rows in one pass).  Words per tile: 16 per lane
    static constexpr uint32_t BUF = BLK == 1024 ? 32 * 1024 : 16 * 1024;        // (the 2-byte bins of the table's slots lie in front)
    static constexpr uint32_t CNT = BUF + 4 * TILE, START = CNT + 4 * DIG, GBASE = START + 4 * (DIG + 8), WAVE = GBASE + 8 * DIG, END = WAVE + 64;
};
constexpr int SHORT_MAX = PG_SHORT_MAX;                                 // a record with at most this many k-mers is "short"

// ---- the word-wise lookups of the slot form: provisional (row, slot) words -> (row, bin) words, 16 per lane and tile, scattered
// by the first digit of their row group into the row shuffle's regions (an LDS multisplit per tile, one global cursor add per
// digit and tile).  Called by a whole workgroup of BLK threads once the bucket's 2-byte bins lie at the start of the dynamic LDS:
// by the counting kernel while the counts are still there, or by mini_lookup_half_kernel.
struct WordCtx {
    unsigned char *lds;
    uint32_t smask, np;
    int lb, vbits;
    ShufArgs sh;
    uint32_t *status;
    const uint32_t *prov_b;
    unsigned long long *dbg;
};
template <int BLK, int DIG>
__device__ __forceinline__ void wordwise_lookup(const WordCtx &c)
{
    using FL = LookupLds<BLK, DIG>;
    constexpr int DPT = DIG / BLK;
    const uint16_t *bins16 = reinterpret_cast<const uint16_t *>(c.lds);
    uint32_t *buf = reinterpret_cast<uint32_t *>(c.lds + FL::BUF), *cnt = reinterpret_cast<uint32_t *>(c.lds + FL::CNT);
    uint32_t *start = reinterpret_cast<uint32_t *>(c.lds + FL::START), *wave_tot = reinterpret_cast<uint32_t *>(c.lds + FL::WAVE);
    unsigned long long *gbase = reinterpret_cast<unsigned long long *>(c.lds + FL::GBASE);
    const ShufArgs &sh = c.sh;
    const uint32_t smask = c.smask, np = c.np;
    const int lb = c.lb, vbits = c.vbits;
    uint32_t *const status = c.status;
    const uint32_t dmask = (1u << sh.gb1) - 1u;
#ifdef PG_MINI_STAMPS
    unsigned long long *dbg = c.dbg;
    unsigned long long tl = __builtin_amdgcn_s_memtime();
#define PG_LAP(K) do { if (threadIdx.x == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); atomicAdd(&dbg[K], now_ - tl); tl = now_; } } while (0)
#else
#define PG_LAP(K) do { } while (0)
#endif
    // Software pipeline.  Every barrier in the loop is an LDS-only one (lds_sync), and there is ONE wait for global memory per
    // tile, in front of the copy-out: by then the next tile's words (requested at the top of the tile), the cursor adds
    // (requested as soon as the ranks, and with them the digits' counts, exist: in front of the scan) and the stores of the
    // previous tile's copy-out have had the whole tile to come back.  (vmcnt counts loads and stores in one queue: a wait
    // for the next words at the top of a tile would wait for the copy-out stores issued just before it.)
    uint32_t w[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const uint32_t i = j * BLK + threadIdx.x;
        w[j] = i < np ? c.prov_b[i] : 0xffffffffu;
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);                      // vmcnt(0): the first tile's words (no wait for w inside the loop)
    for (uint32_t t0 = 0; t0 < np; t0 += FL::TILE) {
#pragma unroll
        for (int q = 0; q < DPT; ++q) cnt[threadIdx.x * DPT + q] = 0;
        uint32_t wn[16];                                     // the next tile's words: in flight until the wait in front of the copy-out
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t i = t0 + FL::TILE + j * BLK + threadIdx.x;
            wn[j] = i < np ? c.prov_b[i] : 0xffffffffu;
        }
        lds_sync();                                          // (also: every wavefront is done with the previous tile's buffer)
        PG_LAP(16);
        // Every step below is written for all 16 words of the lane at once and without branches around the LDS operations -- 16
        // reads in flight and one wait, then 16 returning adds in flight and one wait (a word that is not placed adds 0 to some
        // counter): compiled from a per-word `if`, every word waited twice for a full LDS round trip
        uint32_t dr[16];
        uint32_t live = 0;
        {
            uint32_t b1[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) b1[j] = bins16[w[j] & smask];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const uint32_t i = t0 + j * BLK + threadIdx.x;
                const bool ok = i < np && (uint32_t)(b1[j] - 1u) < 0xfffeu;        // a bin: not 0 (slot never filled), not 0xffff (out of range)
                w[j] = ((w[j] >> lb) << vbits) | (b1[j] - 1u);
                const uint32_t d = (w[j] >> sh.dshift) & dmask;
                dr[j] = (d << 16) | __hip_atomic_fetch_add(&cnt[d], ok ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                live |= ok ? 1u << j : 0u;
            }
        }
        lds_sync();
        PG_LAP(17);
        // this lane's digits: their words of the tile go to a range of the digit's region claimed with one global add
        unsigned long long g_region[DPT], g_claimed[DPT];
#pragma unroll
        for (int q = 0; q < DPT; ++q) {
            const uint32_t d = threadIdx.x * DPT + q, c_mine = cnt[d];
            g_region[q] = g_claimed[q] = 0;
            if (c_mine) {
                g_region[q] = sh.goff[(uint64_t)d << sh.gb2];
                g_claimed[q] = atomicAdd(&sh.gcur1[d], (unsigned long long)c_mine);
            }
        }
        scan_digits_blk<DIG, BLK, true>(cnt, start, wave_tot);
        PG_LAP(18);
        {
            uint32_t at[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) at[j] = start[dr[j] >> 16] + (dr[j] & 0xffffu);
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if ((live >> j) & 1u) buf[at[j]] = w[j];
        }
#pragma unroll
        for (int q = 0; q < DPT; ++q)                         // (the wait for global memory; unused for an empty digit)
            gbase[threadIdx.x * DPT + q] = g_region[q] + g_claimed[q] - start[threadIdx.x * DPT + q];
        lds_sync();
        PG_LAP(19);
        const uint32_t total = start[DIG];
        // copy-out, four words of the lane at a time (their LDS reads in flight together)
        for (uint32_t i0 = 0; i0 < total; i0 += 4 * BLK) {
            uint32_t r[4];
            unsigned long long g[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) r[u] = buf[i0 + u * BLK + threadIdx.x];            // (< FL::TILE: total <= FL::TILE = 16 blocks)
#pragma unroll
            for (int u = 0; u < 4; ++u) g[u] = gbase[(r[u] >> sh.dshift) & dmask] + i0 + u * BLK + threadIdx.x;
            if (sh.narrow) {                                 // (the group region implies the rows' upper bits)
                uint16_t *out16 = reinterpret_cast<uint16_t *>(sh.words_out);
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (i0 + u * BLK + threadIdx.x < total) gstore(out16, g[u], 2ull * sh.words_cap, r[u] & 0x7fffu, status);
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (i0 + u * BLK + threadIdx.x < total) gstore(sh.words_out, g[u], sh.words_cap, r[u], status);
            }
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) w[j] = wn[j];
        PG_LAP(20);
    }
}

// MERGE (SLOTS form only): the k-mers of a record share their row and, being neighbours in a read, mostly their bin.  The lookup
// phase then works record-wise: a lane takes ONE record of a batch, fetches the provisional words of its k-mers, turns them into
// bins and lets every run of equal neighbouring (row, bin) words travel on as ONE word with a count on top,
//     word = (run length - 1) << MERGE_CSHIFT | row << vbits | bin,
// so that ranks, region starts, staging and copy-out are paid per record and per run instead of per occurrence (measured on the
// bench workload: 0.28 words left per provisional word).
// The provisional data of this form lie in FIXED places, so that the lookup phase can fetch them without reading anything else
// first, and they are 2 bytes per k-mer: the bucket's range is
//     [ batches of short records: CXS rows of 64 halfwords each | batches of long records: CAP rows each ]
// and halfword (b x CX + j) x 64 + l of a class holds the SLOT of k-mer j of the record that lane l treated in batch b -- written
// exactly once: by the count loop when the first probe settled the k-mer, else by the general insert when it has found or made
// the slot (the ring carries the halfword's position next to the code).  Which halfwords mean anything follows from the record
// itself (its length and its row, re-read from the records' meta plane: batch b, lane l = record 64 b + l of its class), so
// nothing has to be cleared or marked, every occurrence of a record is in its lane's hands in the lookup phase (runs are never
// broken by k-mers that took the slow path) and the buffer's size follows from the classes' record counts alone, which the plan
// tallies (header[2]): 12 bytes per short record, 22 per long one, against 16 / 36 + a word per slow-path k-mer in round 3's first
// layout (4-byte (row, slot) words with 0xffffffff in the places of absent k-mers, the singles appended behind the batches).
constexpr int MERGE_CSHIFT = PG_SHUFFLE_COUNT_SHIFT;              // needs row bits + vbits <= 28 (mini_merge_form)
constexpr uint32_t WORD_NONE = 0xffffffffu;                       // (never a (row, bin) word: rows stay below 2^20 in this form)
struct MergeArgs {
    uint32_t *prov;                                              // the provisional slots of the MERGE form (halfwords, addressed as dwords)
    unsigned long long cap;                                      // its capacity, in dwords (the buckets claim their ranges on the word cursor)
};
// ---- the record-wise lookups of the MERGE form: provisional slots -> (run, row, bin) words, scattered by the first digit of
// their row group into the row shuffle's regions.  Called by a whole workgroup of BLK threads once the bucket's 2-byte bins lie
// at the start of the dynamic LDS (MergeLds layout).
struct MergeCtx {
    unsigned char *lds;                                          // dynamic LDS (bins16 first)
    uint32_t smask;
    int vbits;
    ShufArgs sh;
    uint32_t *status;
    const uint32_t *prov_b;                                      // the bucket's range (dwords of two slots)
    const uint32_t *meta_s, *meta_l;                             // meta words of the bucket's short / long records
    uint32_t n_s, n_l;                                           // ... and how many there are
    unsigned long long *dbg;                                     // PG_MINI_STAMPS builds: cycle sums of the phases (diagnostic)
};

// LDS of the MERGE form's lookup phase: the 2-byte bins in front, then the stage the wavefronts append their run words to --
// WPL words per lane, sorted IN PLACE: every lane keeps its WPL words and their places in registers across the scan (the first
// layouts of round 3 sorted into a second buffer of the same size instead, i.e. half the stage and twice the sort rounds per
// bucket: the kernel of those days had no registers to spare)
#ifndef PG_MERGE_WPL
#define PG_MERGE_WPL 24
#endif
template <int BLK, int DIG> struct MergeLds {
    static constexpr int WPL = (BLK == 1024 && DIG == 1024) ? PG_MERGE_WPL : (PG_MERGE_WPL < 20 ? PG_MERGE_WPL : 20);   // (144 KiB with 1024 threads, 72 KiB with 512: two workgroups per CU)
    static constexpr uint32_t TILE = WPL * BLK;                          // (>= 64 x 9 words per wavefront: what the wavefronts carry over from a full stage fits an empty one)
    static constexpr uint32_t BUF = BLK == 1024 ? 32 * 1024 : 16 * 1024;        // (the 2-byte bins of the table's slots lie in front)
    static constexpr uint32_t CNT = BUF + 4 * TILE, START = CNT + 4 * DIG, GBASE = START + 4 * (DIG + 8), WAVE = GBASE + 8 * DIG, END = WAVE + 64;
};
// a step of the lookup phase: two batches of short records or one batch of long ones per wavefront (a table whose records are
// not sorted into classes -- caps of at most PG_SHORT_MAX k-mers -- has short records only); KS k-mers per lane, NL loads per
// lane (the slot rows + the records' meta words)
template <int CAP> struct MergeStep {
    static constexpr bool TWO = CAP > PG_SHORT_MAX_;
    static constexpr int CXS = TWO ? PG_SHORT_MAX_ : CAP;
    static constexpr int KS = CAP > 2 * CXS ? CAP : 2 * CXS;
    static constexpr int NL = 2 + 2 * CXS;
    static constexpr int SLACK = 32 * (KS + 2);                                 // dwords a step may read behind the bucket's range (whole rows)
    static_assert(!TWO || (CAP <= 2 * CXS + 1 && NL == 10), "the long batch's rows fit the short step's loads");
};

template <int CAP, int BLK, int DIG>
__device__ __forceinline__ void merged_lookup(const MergeCtx &c)
{
    // Two stages, repeated until the bucket's records are used up (once or twice per bucket).
    // A (no barrier inside): every wavefront walks over its share of the work in STEPS -- two batches of short records (a lane
    //   takes one record of each) or one batch of long records --: fetch the slot rows and the records' meta words (one scalar base
    //   + constant offsets each, nothing to read first), turn the slots into bins, merge equal neighbouring (row, bin) words of a
    //   record into one word with a count, append the step's words to a stage of FL::TILE words in LDS (one returning add per
    //   step claims the positions).  The loads of the NEXT step are in flight meanwhile, and the wavefronts are at different
    //   points of the loop at any time.
    // B (when the stage is full or the work is done): the staged words are sorted by the first digit of their row group into a
    //   second buffer (WPL per lane: rank, scan, place) and copied out to the group regions, one global cursor add per digit.
    using FL = MergeLds<BLK, DIG>;
    using ST = MergeStep<CAP>;
    constexpr int DPT = DIG / BLK, WAVES_B = BLK / 64, WPL = FL::WPL;
    constexpr bool TWO = ST::TWO;
    constexpr int CXS = ST::CXS, KS = ST::KS, NL = ST::NL;
    constexpr uint32_t NONE = WORD_NONE;
    __shared__ uint32_t staged, valid_end, busy;
    const uint16_t *bins16 = reinterpret_cast<const uint16_t *>(c.lds);
    uint32_t *buf = reinterpret_cast<uint32_t *>(c.lds + FL::BUF), *cnt = reinterpret_cast<uint32_t *>(c.lds + FL::CNT);
    uint32_t *start = reinterpret_cast<uint32_t *>(c.lds + FL::START), *wave_tot = reinterpret_cast<uint32_t *>(c.lds + FL::WAVE);
    unsigned long long *gbase = reinterpret_cast<unsigned long long *>(c.lds + FL::GBASE);
    const ShufArgs &sh = c.sh;
    const int vbits = c.vbits;
    uint32_t *const status = c.status;
    const uint32_t *const prov_b = c.prov_b;
    // (one class only: its records -- handed over as the "long" ones, which is what the count loop's second range is -- are
    // treated two batches per step, as short records are)
    const uint32_t *const meta_s = TWO ? c.meta_s : c.meta_l, *const meta_l = c.meta_l;
    const uint32_t lane = lane_id(), wave = uniform32(threadIdx.x >> 6);
    const uint32_t half_lane = lane >> 1, parity_shift = (lane & 1u) * 16u;      // a row of 64 halfwords is read as 32 dwords: two lanes share one
    const uint32_t dmask = (1u << sh.gb1) - 1u;
    uint32_t *const wout = sh.words_out;
#ifdef PG_MINI_STAMPS
    unsigned long long tl_ = __builtin_amdgcn_s_memtime();
#define PG_MLAP(K) do { if (threadIdx.x == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); atomicAdd(&c.dbg[48 + K], now_ - tl_); tl_ = now_; } } while (0)
#else
#define PG_MLAP(K) do { } while (0)
#endif
    const uint32_t n_s = uniform32(TWO ? c.n_s : c.n_l), n_l = TWO ? uniform32(c.n_l) : 0u;
    const uint32_t n_sb = (n_s + 63u) >> 6, n_lb = (n_l + 63u) >> 6;
    const uint32_t long_base = n_sb * 32u * CXS;                 // (dwords: a row of 64 slots is 32 of them)
    const uint32_t n_ss = (n_sb + 1u) >> 1;                      // steps of short batches, then one per long batch
    const uint32_t n_steps = n_ss + n_lb;
    if (n_steps == 0u) return;                                   // (uniform)
    // a step: first dword of its slot rows (A0), is it a step of short records (SHORT), first record of its (first) batch and the
    // records of its class (REC, NCLS)
    struct StepGeo { uint32_t a0, is_short, rec, ncls; };
    auto geometry = [&](uint32_t st) -> StepGeo {
        StepGeo g;
        if (!TWO || st < n_ss) { g.a0 = st * (64u * CXS); g.is_short = 1u; g.rec = st * 128u; g.ncls = n_s; }
        else { g.a0 = long_base + (st - n_ss) * (32u * CAP); g.is_short = 0u; g.rec = (st - n_ss) * 64u; g.ncls = n_l; }
        return g;
    };
    // the loads of a step: scalar bases and constant offsets (whole rows; what lies beyond the step's own rows is read and
    // ignored: the buffer has slack for it; a record index beyond its class is clamped).  Nothing is computed from the values here.
    //   w[0] = meta word of the lane's (first) record, w[1 + i] = slot row i, w[9] (two classes) = meta word of the second short
    //   record, or slot row 8 of a long one
    auto fetch_rows = [&](const StepGeo &g, uint32_t (&w)[NL]) {
        const uint32_t *rows = prov_b + g.a0;
        const uint32_t *meta_c = g.is_short ? meta_s : meta_l;
        const uint32_t last = g.ncls - 1u;
        const uint32_t ia = g.rec + lane < last ? g.rec + lane : last;
        w[0] = meta_c[ia];
#pragma unroll
        for (int i = 0; i < 2 * CXS; ++i) w[1 + i] = rows[32 * i + half_lane];
        {
            const uint32_t ib = g.rec + 64u + lane < last ? g.rec + 64u + lane : last;
            const uint32_t *p9 = g.is_short ? meta_c : rows + 32 * (2 * CXS);
            w[NL - 1] = p9[g.is_short ? ib : half_lane];
        }
    };
    // meta words + slots -> bins -> (row, bin) words; equal neighbours inside a record become ONE word with the run length on top
    auto process = [&](const StepGeo &g, const uint32_t (&r)[NL], uint32_t (&w)[KS]) {
        const uint32_t ma = r[0];
        const uint32_t row_a = ma >> META_ROW_SHIFT, n_a = ((ma >> META_D2_BITS) & (uint32_t)(MINI_MAX_LEN - 1)) + 1u;
        const bool ok_a = g.rec + lane < g.ncls && row_a != MINI_ROW_NONE;
        // the record that k-mer slots CXS .. KS - 1 belong to: the second short record, or the long one again
        uint32_t row_x = row_a, n_x = n_a;
        bool ok_x = ok_a;
        {
            const uint32_t mb = r[NL - 1];
            const uint32_t row_b = mb >> META_ROW_SHIFT, n_b = ((mb >> META_D2_BITS) & (uint32_t)(MINI_MAX_LEN - 1)) + 1u;
            const bool ok_b = g.rec + 64u + lane < g.ncls && row_b != MINI_ROW_NONE;
            row_x = g.is_short ? row_b : row_a;
            n_x = g.is_short ? n_b + (uint32_t)CXS : n_a;        // (a short record has at most CXS k-mers: slot 2 CXS is never its own)
            ok_x = g.is_short ? ok_b : ok_a;
        }
        uint32_t b1[KS];
#pragma unroll
        for (int i = 0; i < KS; ++i) {
            // (no mask with the bucket's slot count: a halfword that means nothing may hold anything, and whatever it selects in
            // the 128 KiB behind the bins is read and ignored)
            const uint32_t pair = r[i < 2 * CXS ? 1 + i : NL - 1];
            b1[i] = bins16[__builtin_amdgcn_ubfe(pair, parity_shift, 16u)];
        }
#pragma unroll
        for (int i = 0; i < KS; ++i) {
            const bool first = i < CXS;
            const bool ok = (first ? ok_a && (uint32_t)i < n_a : ok_x && (uint32_t)i < n_x) && b1[i] != 0xffffu;
            w[i] = ok ? ((first ? row_a : row_x) << vbits) | b1[i] : NONE;
        }
        // (from the last slot down: `run` = how many slots behind this one repeat it.  Two absent neighbours also count as "the same",
        // which never reaches a head: the run is reset where an absent slot follows a present one)
        uint32_t run = 0;
#pragma unroll
        for (int i = KS - 1; i >= 0; --i) {
            const uint32_t cur = w[i];
            const bool rec_start = i == 0 || (i == CXS && g.is_short);
            const bool same = !rec_start && cur == w[i > 0 ? i - 1 : 0];
            w[i] = same ? NONE : (run << MERGE_CSHIFT) | cur;     // (an absent slot stays absent: NONE is all ones)
            run = same ? run + 1u : 0u;
        }
    };
    // this thread's digits of a tile: claim their ranges in the group regions (one global add per digit and tile), scan
    auto claim_and_scan = [&]() {
        unsigned long long g_region[DPT], g_claimed[DPT];
#pragma unroll
        for (int q = 0; q < DPT; ++q) {
            const uint32_t d = threadIdx.x * DPT + q, c_mine = cnt[d];
            g_region[q] = g_claimed[q] = 0;
            if (c_mine) {
                g_region[q] = sh.goff[(uint64_t)d << sh.gb2];
                g_claimed[q] = atomicAdd(&sh.gcur1[d], (unsigned long long)c_mine);
            }
        }
        scan_digits_blk<DIG, BLK, true>(cnt, start, wave_tot);
#pragma unroll
        for (int q = 0; q < DPT; ++q)                             // (the wait for global memory; unused for an empty digit)
            gbase[threadIdx.x * DPT + q] = g_region[q] + g_claimed[q] - start[threadIdx.x * DPT + q];
    };
    if (threadIdx.x == 0) { staged = 0; valid_end = NONE; busy = 0; }
    uint32_t step = wave;                                        // this wavefront's next step
    bool pending = false;                                        // pw[] holds words that did not fit the stage yet (wave-uniform)
    uint32_t pw[KS];                                             // the words of the step in hand
    // the words of pw[] -> the stage; false (and nothing written) when they do not fit this round any more
    auto append = [&]() -> bool {
        // (the ballots are taken once and the lanes picked from the scalar masks: a predicate that crosses the branches below
        // would be re-materialised per use as v_cndmask + v_cmp)
        unsigned long long hm[KS];
        uint32_t n = 0;
#pragma unroll
        for (int j = 0; j < KS; ++j) { hm[j] = __builtin_amdgcn_ballot_w64(pw[j] != NONE); n += (uint32_t)__popcll(hm[j]); }
        if (n == 0) return true;                                 // (uniform)
        uint32_t at = 0;
        if (lane == 0) at = atomicAdd(&staged, n);
        at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
        if (at + n > FL::TILE) {                                 // the stage is full: this step's words wait for the next round
            if (lane == 0) atomicMin(&valid_end, at);
            return false;
        }
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            if (hm[j] == 0ull) continue;                         // (uniform)
            if (__builtin_amdgcn_inverse_ballot_w64(hm[j])) buf[at + lanes_below(hm[j])] = pw[j];
            at += (uint32_t)__popcll(hm[j]);
        }
        return true;
    };
    // one step: its loads are in `mine`; the next step's go out into `next` first, then the step is worked on and appended.
    // 0: done, 1: no step left, 2: the stage is full (the step's words stay in pw[])
    auto one_step = [&](uint32_t (&mine)[NL], uint32_t (&next)[NL]) -> int {
        if (step >= n_steps) return 1;                           // (uniform)
        const StepGeo g = geometry(step);
        step += WAVES_B;
        // (scheduling barriers: the compiler must neither sink these requests below the work on the step in hand nor pull
        // that work -- and with it the wait for what was just requested -- up in front of them)
        __builtin_amdgcn_sched_barrier(0);
        fetch_rows(geometry(step < n_steps ? step : 0u), next);  // the next step's loads: in flight while this one is worked on
        __builtin_amdgcn_sched_barrier(0);
        process(g, mine, pw);
        return append() ? 0 : 2;
    };
    lds_sync();
    for (;;) {
        // ---- A.  (What a full stage left in pw[] goes first: the stage is empty now.  The loads of `step` are requested again
        // behind every stage B, which had the registers; the two buffers swap roles from step to step: no copies)
        if (pending) pending = !append();                        // (a step's words always fit an empty stage: 64 KS words per wavefront)
        if (!pending) {
            uint32_t ca[NL], cb[NL];
            fetch_rows(geometry(step < n_steps ? step : 0u), ca);
            for (;;) {
                int r = one_step(ca, cb);
                if (r == 0) r = one_step(cb, ca);
                if (r) { pending = r == 2; break; }
            }
        }
        if (lane == 0 && (pending || step < n_steps)) atomicOr(&busy, 1u);
#pragma unroll
        for (int q = 0; q < DPT; ++q) cnt[threadIdx.x * DPT + q] = 0;     // (nobody reads the counters between the placement of one round and here)
        lds_sync();
        PG_MLAP(0);                                              // (stage A)
        const uint32_t n_valid = staged < valid_end ? staged : valid_end;
        const bool more = busy != 0u;
        // ---- B: the staged words [0, n_valid), WPL per lane (as many of them as the round has: uniform tests): rank, scan, place
        // -- in place: every word is in its lane's registers before the first barrier --, copy out
        {
            uint32_t w[WPL], at[WPL];
#pragma unroll
            for (int j = 0; j < WPL; ++j) w[j] = (uint32_t)(j * BLK) < n_valid ? buf[j * BLK + threadIdx.x] : 0u;
#pragma unroll
            for (int j = 0; j < WPL; ++j) {
                const uint32_t i = j * BLK + threadIdx.x;
                at[j] = 0;
                if ((uint32_t)(j * BLK) < n_valid)                  // (uniform)
                    at[j] = __hip_atomic_fetch_add(&cnt[(w[j] >> sh.dshift) & dmask], i < n_valid ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            lds_sync();
            if (threadIdx.x == 0) { staged = 0; valid_end = NONE; busy = 0; }     // (everybody has read them: a barrier lies in between)
            claim_and_scan();
#pragma unroll
            for (int j = 0; j < WPL; ++j)
                if ((uint32_t)(j * BLK) < n_valid) at[j] += start[(w[j] >> sh.dshift) & dmask];
#pragma unroll
            for (int j = 0; j < WPL; ++j)
                if ((uint32_t)(j * BLK) + threadIdx.x < n_valid) buf[at[j]] = w[j];
        }
        lds_sync();
        PG_MLAP(1);                                              // (ranks, cursor adds, scan, placement)
        // (not unrolled: the three iterations a full stage takes would each keep their own four 64-bit store bases -- wout + position
        // -- alive across the whole round loop: 24 vector registers, some of them spilled and reloaded in front of the stores with
        // a wait for every store in flight)
#pragma unroll 1
        for (uint32_t i0 = 0; i0 < n_valid; i0 += 4 * BLK) {
            uint32_t r[4];
            unsigned long long g[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t i = i0 + u * BLK + threadIdx.x;
                r[u] = buf[i < FL::TILE ? i : 0u];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) g[u] = gbase[(r[u] >> sh.dshift) & dmask] + i0 + u * BLK + threadIdx.x;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i0 + u * BLK + threadIdx.x < n_valid) gstore(wout, g[u], sh.words_cap, r[u], status);
        }
#ifdef PG_MINI_STAMPS
        if (threadIdx.x == 0) { atomicAdd(&c.dbg[48 + 6], 1ull); atomicAdd(&c.dbg[48 + 7], (unsigned long long)n_valid); }
#endif
        PG_MLAP(2);                                              // (copy-out)
        if (!more) break;                                        // (uniform)
        lds_sync();                                              // (the stage is free again: the next round's wavefronts may append)
    }
}

// HALF (N > 1 ranks; SLOTS form, packed slots): the COUNT half of the kernel.  The counts of a rank's own reads are not final --
// the other ranks hold occurrences of the same k-mers --, so instead of the slice and the lookups the bucket leaves
//   * its occupied slots (code << 22 | count) in slot order, compacted, in its slab of `ent` (the entries that travel to the
//     bucket's owner: pg_mini_gather_entries), their number in `fill`, and WHICH slots they are as a bitmap in `occ`;
//   * what the lookup half needs to find the bucket's provisional words (merged form: slots) again: where they start (`wbeg`)
//     (`ring_cnt`: the singles of the merged form's first layout; always 0 since the general insert writes into fixed places).
// mini_lookup_half_kernel finishes the bucket once the owner has answered with the bins of exactly these entries, in order.
// PIECES (one GPU, a stream counted in several word ranges because its scratch would not fit in one piece: pg_mini_count_piece):
// the same count half, but on the TABLE's own buckets -- the bucket's slice is loaded into LDS first (accum = 2; the first piece
// starts from an empty table: accum = 1), counted on, and written back; the slots keep their places from piece to piece (nothing
// ever leaves a slot), so the provisional slots of every piece stay valid until mini_lookup_slice_merge_kernel has looked them up
// in the final table.
struct HalfArgs {
    unsigned long long *ent;                                     // bucket b: entries from b << log2 bucket slots
    unsigned long long *occ;                                     // bucket b: max(1, bucket slots / 64) words from b * that
    long long *fill;                                             // [buckets]
    uint32_t *ring_cnt;                                          // [buckets]
    int accum;                                                   // 0: N-rank count half; 1, 2: a piece (see above)
};
template <int CAP, bool SLOTS, bool WIDE, int BLK, int DIG, bool MERGE = false, bool HALF = false>
__global__ __launch_bounds__(BLK, 4) void mini_count_kernel(const uint64_t *__restrict__ bases, const uint32_t *__restrict__ meta,
                                                               const unsigned long long *__restrict__ off,
                                                               const unsigned long long *__restrict__ n_short,
                                                               const unsigned long long *__restrict__ kwords, MiniView t,
                                                               uint32_t window, uint32_t vsize, int vbits, uint32_t *__restrict__ words,
                                                               uint32_t *__restrict__ prov, unsigned long long *__restrict__ word_cursor,
                                                               unsigned long long *__restrict__ wbeg, unsigned long long *__restrict__ emit_end,
                                                               ShufArgs sh, MergeArgs mg, unsigned long long rec_cap, uint32_t *status, HalfArgs hv = HalfArgs{nullptr, nullptr, nullptr, nullptr, 0})
{
    static_assert(!MERGE || SLOTS, "merged words are a form of the slot lookups");
    static_assert(!HALF || (SLOTS && !WIDE), "the count half is a form of the slot lookups on packed slots");
#ifdef PG_MINI_GAPS                                              // (variant build for tools/wg_gaps.py: when a workgroup began and ended, and where)
    const unsigned long long gap_t0 = __builtin_amdgcn_s_memtime();
#endif
    // (the plan this launch was given describes another stream: its record count does not fit the record buffers -- nothing is
    // touched, bit 2 of the status word says so; KmerTable ties its cached plans to the stream, this is the backstop)
    if (word_cursor[-1] > rec_cap || (*status & PG_STATUS_PLAN_MISMATCH)) {     // (... or the first pass found the plan to be another stream's)
        if (threadIdx.x == 0 && blockIdx.x == 0) atomicOr(status, PG_STATUS_PLAN_MISMATCH);
        return;
    }
    extern __shared__ __attribute__((aligned(16))) unsigned long long tab[];
    __shared__ uint32_t emitted, emitted_ring;
    __shared__ unsigned long long n_lookups, wbase;
    __shared__ unsigned long long wave_words[(BLK / 64)];
    const uint32_t n_slots = 1u << t.log2_bucket;
    const uint32_t smask = n_slots - 1;
    const uint32_t limit = n_slots < MAX_PROBE ? n_slots : MAX_PROBE;
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    uint32_t *const cnts = reinterpret_cast<uint32_t *>(tab + n_slots);             // WIDE: the count plane behind the key plane
    const uint32_t tab_units = WIDE ? n_slots + n_slots / 2 : n_slots;              // 8-byte units of the table
    unsigned long long *ring = tab + tab_units + wave * RING;                        // [RING] codes of this wavefront
    uint32_t *ring_row = reinterpret_cast<uint32_t *>(tab + tab_units + (BLK / 64) * RING) + wave * RING;
    const int k = t.k, lb = t.log2_bucket;
    const uint64_t kmask = (1ull << (2 * k)) - 1ull;
    const int rc_sh0 = 2 * (32 - k);
    uint64_t *slice = t.slots + ((uint64_t)blockIdx.x << t.log2_bucket);
    uint32_t *slice_counts = reinterpret_cast<uint32_t *>(t.slots + (1ull << t.log2_slots)) + ((uint64_t)blockIdx.x << t.log2_bucket);   // WIDE
    const int64_t r0 = (int64_t)off[blockIdx.x], r1 = (int64_t)off[blockIdx.x + 1];
    // [r0, rs) short records (at most SHORT_MAX k-mers; the second scatter pass put them first), [rs, r1) the others
    const int64_t rs = n_short && CAP > SHORT_MAX ? r0 + (int64_t)n_short[blockIdx.x] : r0;
    const bool emit_slots = SLOTS && window != 0;
    // every lane's first record of both classes, requested before anything else: with one workgroup per CU nothing hides a bucket's
    // start (bounds -> table cleared -> range claimed -> barrier -> first records -> first probes), so the records' round trip
    // runs under the clearing of the table and the claim
    const int64_t i_s = r0 + (int64_t)wave * 64 + lane, i_l = rs + (int64_t)wave * 64 + lane;
    const uint64_t first_s = i_s < rs ? bases[i_s] : 0ull, first_l = i_l < r1 ? bases[i_l] : 0ull;
    const uint32_t first_ms = i_s < rs ? meta[i_s] : 0xffffffffu, first_ml = i_l < r1 ? meta[i_l] : 0xffffffffu;
#ifdef PG_MINI_STAMPS
    unsigned long long *dbg = word_cursor + 7;                   // header[8..]: phase cycle sums (diagnostic build only)
    const unsigned long long st0 = __builtin_amdgcn_s_memtime();
#define PG_STAMP(K) do { __syncthreads(); if (threadIdx.x == 0) atomicAdd(&dbg[K], (unsigned long long)(__builtin_amdgcn_s_memtime() - st0)); } while (0)
    // per-wave laps inside the count loop: dbg[32 + K] += time since the previous lap (everything in flight is waited for first)
    unsigned long long wl = st0, wacc[5] = {0, 0, 0, 0, 0};
#if PG_MINI_STAMPS + 0 >= 2              // (-DPG_MINI_STAMPS=2: laps inside the count loop too -- every lap waits for all memory operations, which slows the loop down several times)
#define PG_WLAP(K) do { __builtin_amdgcn_s_waitcnt(0); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
                        wacc[K] += now_ - wl; wl = now_; } while (0)
#else
#define PG_WLAP(K) do { (void)wl; (void)wacc; } while (0)
#endif
#else
#define PG_STAMP(K) do { } while (0)
#define PG_WLAP(K) do { } while (0)
#endif
    if (HALF && hv.accum == 2) {                                 // (a later piece: the bucket as the pieces before left it)
        for (uint32_t i = threadIdx.x; i < tab_units; i += BLK) tab[i] = i < n_slots ? (unsigned long long)slice[i] : 0ull;
    } else {
        for (uint32_t i = threadIdx.x; i < tab_units; i += BLK) tab[i] = 0ull;
    }
    if (threadIdx.x == 0) { emitted = 0; emitted_ring = 0; n_lookups = 0; }
    if (emit_slots) {
        // every occurrence that lies in a row leaves exactly one word: the bucket's word range is claimed before the first word is
        // written (one global add).  The total comes with the records (the second scatter pass tallies it per bucket); without that
        // pass -- at most 256 buckets -- a pre-pass over the meta plane counts it.
        // (MERGE: nothing but fixed places, 2 bytes per k-mer of every batch of 64 records -- see MergeArgs; claimed in dwords)
        const unsigned long long fixed_words = MERGE ? 32ull * (unsigned long long)((rs - r0 + 63) >> 6) * (CAP > SHORT_MAX ? SHORT_MAX : CAP)
                                                       + 32ull * (unsigned long long)((r1 - rs + 63) >> 6) * CAP : 0ull;
        if (MERGE) {
            if (threadIdx.x == 0) wbase = atomicAdd(word_cursor, fixed_words);
        } else if (kwords) {
            if (threadIdx.x == 0) {
                n_lookups = kwords[blockIdx.x];
                wbase = atomicAdd(word_cursor, n_lookups);
            }
        } else {
            unsigned long long mine = 0;
            for (int64_t i0 = r0 + threadIdx.x; i0 < r1; i0 += BLK) {
                const uint32_t m = meta[i0];
                if ((m >> META_ROW_SHIFT) != MINI_ROW_NONE) mine += ((m >> META_D2_BITS) & (MINI_MAX_LEN - 1)) + 1;
            }
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) mine += __shfl_down(mine, d);
            if (lane == 0) wave_words[wave] = mine;
            __syncthreads();
            if (threadIdx.x == 0) {
                unsigned long long run = 0;
                for (int w = 0; w < (BLK / 64); ++w) run += wave_words[w];
                n_lookups = run;
                wbase = atomicAdd(word_cursor, run);
            }
        }
    }
    __syncthreads();
    if (MERGE && emit_slots) {
        // (the word buffer was sized from the plan's record counts; a range that does not fit it -- a plan of other reads -- is
        // not written: bit 2 of the status word, as for the record buffers)
        const unsigned long long need = 32ull * (unsigned long long)((rs - r0 + 63) >> 6) * (CAP > SHORT_MAX ? SHORT_MAX : CAP)
                                        + 32ull * (unsigned long long)((r1 - rs + 63) >> 6) * CAP + (unsigned long long)MergeStep<CAP>::SLACK;
        if (wbase + need > mg.cap || 2ull * need >= (unsigned long long)RING_NOPLACE) {      // (uniform; places are 32-bit halfword numbers below RING_NOPLACE)
            if (threadIdx.x == 0) atomicOr(status, PG_STATUS_PLAN_MISMATCH);
            return;
        }
    }
    unsigned long long wb = emit_slots ? wbase : 0ull;
    // (read from LDS, hence a vector register to the compiler; made a scalar so that the words' addresses are a scalar base + a
    // 32-bit lane offset instead of a 64-bit sum per lane and load)
    wb = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(wb >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)wb);
    uint32_t *const prov_b = (MERGE ? mg.prov : prov) + wb;      // the bucket's words (32-bit positions from here on)
    // words are placed by claiming positions on the bucket's LDS counter `emitted`: ONE returning add per batch of 64 records
    // (all its first-probe hits) and one per general insert round, by lane 0
    auto claim = [&](uint32_t n_words) -> uint32_t {
        uint32_t at = 0;
        if (lane == 0) at = atomicAdd(&emitted, n_words);
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
    };
    // MERGE: the bucket's range = batches of short records | batches of long records, a halfword per k-mer (see MergeArgs)
    constexpr uint32_t CXS_ = CAP > SHORT_MAX ? SHORT_MAX : CAP;
    const uint32_t n_sb = (uint32_t)((rs - r0 + 63) >> 6), n_lb = (uint32_t)((r1 - rs + 63) >> 6);
    const uint32_t long_base = n_sb * 64u * CXS_;                // (halfwords)
    const uint32_t np_half = long_base + n_lb * 64u * CAP;       // (... of the bucket, for the checked build)
    uint16_t *const prov_h = reinterpret_cast<uint16_t *>(prov_b);
    const uint32_t np_all = (uint32_t)n_lookups;                 // (emit_slots: the bucket's words, known before the first is written)
    PG_STAMP(0);
    bool full = false;
    unsigned long long mine = 0;                                 // k-mers of this lane's records that lie inside a row
    uint32_t head = 0, tail = 0;                                 // ring positions (wave-uniform)
    // one pending occurrence per lane off the ring: the general insert, and its word.
    // Packed slots (k <= 21): ONE turn per round.  A ring entry is code | next slot to visit << 42 | turns taken << 56; the round
    // reads that slot and the one behind it (both reads in flight), claims or counts, and a lane that is not settled by them puts
    // its entry BACK on the ring, two slots further on -- a round used to last as long as the longest probe chain among its 64
    // lanes (three to four dependent turns of three LDS round trips each, every other lane waiting), now a lane costs as many
    // turns as its own chain has (1.15 on average at load 0.35).  After 255 turns (510 slots) the bucket counts as full.
    constexpr uint64_t RING_CODE = (1ull << 42) - 1ull;
    constexpr int RING_SLOT_SHIFT = 42, RING_TURN_SHIFT = 56;
    static_assert(PG_BUCKET_MAX_LOG2_SLOTS <= RING_TURN_SHIFT - RING_SLOT_SHIFT, "a slot index fits its field of the ring entry");
    auto slow_round = [&](bool act) {
        const uint32_t at = (head + lane) & (RING - 1);
        const uint64_t c_ = act ? ring[at] : 0ull;
        constexpr uint32_t RING_NONE = MERGE ? 0xffffffffu : MINI_ROW_NONE;      // (MERGE: ring_row holds places, not rows; no place: anything from RING_NOPLACE on)
        const uint32_t rw = act && emit_slots ? ring_row[at] : RING_NONE;
        uint32_t sl;
        bool settled = act;                                      // (lanes whose word may be written this round)
        if constexpr (WIDE) {
            const uint64_t c = c_ & ~(1ull << 63);               // (bit 63: the home slot was taken by another key when the first probe looked)
            sl = mini_insert_slow<WIDE>(tab, cnts, smask, limit, c, act, (uint32_t)(c_ >> 63));
        } else {
            const uint64_t code = c_ & RING_CODE;
            const uint32_t s0 = (uint32_t)(c_ >> RING_SLOT_SHIFT) & smask, s1 = (s0 + 1u) & smask;
            const uint32_t turns = (uint32_t)(c_ >> RING_TURN_SHIFT);
            const unsigned long long fresh = (unsigned long long)((code << HASH_CBITS) | 1ull);
            const unsigned long long cur0 = tab[s0], cur1 = tab[s1];
            bool todo = act;
            sl = act ? 0xffffffffu : 0u;
            // one slot of the chain: claim it if it is empty, count on it if it holds the key (what was read may be stale by now:
            // a claim goes through compare-and-swap, which answers with what is there)
            auto visit = [&](uint32_t slot, unsigned long long cur) {
                if (todo && cur == 0) cur = atomicCAS(&tab[slot], 0ull, fresh);
                const bool claimed = todo && cur == 0;
                const bool match = todo && cur != 0 && (cur >> HASH_CBITS) == code;
                // stop growing at SAT; the overshoot is bounded by the lanes in flight and clamped when the slice is packed
                if (match && (uint32_t)(cur & HASH_CMASK) < HASH_SAT) atomicAdd(reinterpret_cast<uint32_t *>(&tab[slot]), 1u);   // (the count is in the low dword)
                if (claimed || match) { sl = slot; todo = false; }
            };
            visit(s0, cur0);
            visit(s1, cur1);
            // (a bucket with fewer slots than a lane may visit: it has seen them all)
            const bool give_up = todo && (turns == 255u || 2u * (turns + 1u) >= limit);
            const bool again = todo && !give_up;
            settled = act && !again;
            const unsigned long long am = __ballot(again);
            if (am) {                                            // (uniform; the round's own 64 places are free: they were read above)
                if (again) {
                    const uint32_t to = (tail + lanes_below(am)) & (RING - 1);
                    ring[to] = code | ((uint64_t)((s0 + 2u) & smask) << RING_SLOT_SHIFT) | ((uint64_t)(turns + 1u) << RING_TURN_SHIFT);
                    if (emit_slots) ring_row[to] = rw;
                }
                tail += (uint32_t)__popcll(am);
            }
        }
        full |= settled && sl == 0xffffffffu;
        if (emit_slots) {
            // (a full bucket still gets its word -- the slot of a k-mer that is not there reads as "no bin")
            const bool put = settled && (MERGE ? rw < RING_NOPLACE : rw != RING_NONE);
            const unsigned long long qm = __ballot(put);
            if (MERGE) {
                // (the ring carried the k-mer's own place among the bucket's halfwords instead of its row)
                if (put) gstore(prov_h, (uint64_t)rw, (uint64_t)np_half, (uint16_t)(sl == 0xffffffffu ? 0xffffu : sl & smask), status);
            } else
            if (qm) {                                            // (uniform)
                {
                    const uint32_t at = claim((uint32_t)__popcll(qm));
                    if (put) gstore(prov_b, (uint64_t)at + lanes_below(qm), (uint64_t)np_all, (rw << lb) | (sl & smask), status);
                }
            }
        }
    };
    // ---- count: wavefront w takes the batches [ra + 64 (w + 16 t), + 64) of a class; CX = k-mers per record at most there
    auto count_range = [&](auto cx, int64_t ra, int64_t rb, uint64_t R, uint32_t m) {       // (R, m: the lane's first record of the range, loaded at the top of the kernel)
        constexpr int CX = decltype(cx)::value;
        int64_t i = ra + (int64_t)wave * 64 + lane;
        for (int64_t i0 = ra + (int64_t)wave * 64; i0 < rb; i0 += BLK) {
            PG_WLAP(0);                                          // (loop top: the record has arrived, last batch's stores are out)
            const bool live = i0 + lane < rb;
            const int n = live ? (int)((m >> META_D2_BITS) & (MINI_MAX_LEN - 1)) + 1 : 0;
            const uint32_t row = m >> META_ROW_SHIFT;
            const bool in_row = live && row != MINI_ROW_NONE;
            // (MERGE) the lane's first halfword in the bucket's range: k-mer j of its record has place0 + 64 j
            const uint32_t place0 = (CX < CAP ? 0u : long_base) + (uint32_t)((i0 - ra) >> 6) * (64u * CX) + lane;
            // (what the ring carries for k-mer j is ring_place0 + 64 j: its place, or -- outside every row -- something at or beyond RING_NOPLACE)
            const uint32_t ring_place0 = row != MINI_ROW_NONE ? place0 : RING_NOPLACE;
            if (in_row) mine += (unsigned long long)n;
            const uint64_t FW = rev2_64(R), RC = R ^ 0xAAAAAAAAAAAAAAAAull;
            i = i0 + BLK + lane;                           // the next batch's loads fly during this one
            {
                // ... provided they are issued BEHIND the wait for this batch's record: the compiler hoists them to the top of the
                // loop, in front of that wait, and -- the loads being conditional -- the wait is an s_waitcnt vmcnt(0) that then
                // covers them too: every batch stood still for a full memory round trip (one workgroup per CU: nothing else to
                // run).  Making their address depend on this batch's record pins them behind the wait.
                const uint32_t arrived = (uint32_t)FW;
                asm volatile("" : "+v"(i) : "v"(arrived));
            }
            R = i < rb ? bases[i] : 0ull;
            m = i < rb ? meta[i] : 0xffffffffu;
            // Straight-line code from here to the ballots: the reads and the adds go out for every lane (a lane without a j-th k-mer
            // reads some slot and adds 0 to it), so that no predicate crosses a branch -- a predicate that does comes back as
            // v_cndmask + v_cmp per use, and the branches around each predicated add cost more than the add
            uint64_t code[CX];
            uint32_t sl[CX];
            unsigned long long cur[CX];
#pragma unroll
            for (int j = 0; j < CX; ++j) {                       // every first probe of the record in flight
                // k-mer j and its reverse complement, both pushed up to bit 63 (what lies below them is the same junk for the
                // comparison's purposes: the 2k bits on top decide, a palindrome gives the same code either way)
                const uint64_t fw = FW << ((rc_sh0 - 2 * j) & 63), rc = RC << (2 * j);        // (j beyond the cap: no such k-mer, any value will do)
                code[j] = (fw < rc ? fw : rc) >> rc_sh0;
                sl[j] = mini_slot_hash<WIDE>(code[j]) & smask;
                cur[j] = tab[sl[j]];
            }
            // (ballots of single comparisons, combined as scalars: the ballot of a compound predicate is compiled as
            // v_cndmask + v_cmp on top of the scalar logic)
            const unsigned long long in_row_m = __builtin_amdgcn_ballot_w64(row != MINI_ROW_NONE);
            PG_WLAP(1);                                          // (codes, hashes, first probes back)
            unsigned long long pm[CX], qm[CX];                   // lanes whose j-th k-mer was settled by the first probe inside a row / is still pending
#pragma unroll
            for (int j = 0; j < CX; ++j) {
                const unsigned long long act = __builtin_amdgcn_ballot_w64(j < n);
                unsigned long long hit;
                if (WIDE) {
                    hit = act & __builtin_amdgcn_ballot_w64(cur[j] == code[j] + 1ull);
                    __hip_atomic_fetch_add(&cnts[sl[j]], __builtin_amdgcn_inverse_ballot_w64(hit) ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                } else {
                    hit = act & __builtin_amdgcn_ballot_w64(cur[j] != 0) & __builtin_amdgcn_ballot_w64((cur[j] >> HASH_CBITS) == code[j]);
                    // (the count sits in the low 22 bits of the slot's low dword and stops far below 2^22: a 32-bit LDS add is enough)
                    const unsigned long long room = __builtin_amdgcn_ballot_w64(((uint32_t)cur[j] & HASH_SAT) == 0);
                    __hip_atomic_fetch_add(reinterpret_cast<uint32_t *>(&tab[sl[j]]), __builtin_amdgcn_inverse_ballot_w64(hit & room) ? 1u : 0u,
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                pm[j] = hit & in_row_m;
                qm[j] = act & ~hit;
            }
            uint32_t at = 0;
            if (emit_slots) {
                // positions for the provisional words of the hits: the claim is issued here and used behind the ring pushes
                uint32_t total = 0;
#pragma unroll
                for (int j = 0; j < CX; ++j) total += (uint32_t)__popcll(pm[j]);
                if (total && !MERGE) at = claim(total);          // (uniform; the MERGE form's words have fixed places)
            }
            PG_WLAP(2);                                          // (hits added, positions claimed)
#pragma unroll
            for (int j = 0; j < CX; ++j) {
                const unsigned long long mask = qm[j];
                if (mask) {                                      // (uniform)
                    if (__builtin_amdgcn_inverse_ballot_w64(mask)) {
                        const uint32_t at = (tail + lanes_below(mask)) & (RING - 1);
                        // (a slot that is not empty and not a hit holds another key, and nothing ever leaves a slot: the search
                        // starts behind it)
                        if constexpr (WIDE) ring[at] = code[j] | (cur[j] != 0ull ? 1ull << 63 : 0ull);
                        else ring[at] = code[j] | ((uint64_t)((sl[j] + (cur[j] != 0ull ? 1u : 0u)) & smask) << RING_SLOT_SHIFT);
                        // (MERGE: the general insert writes the slot to the k-mer's own place: the ring carries that place)
                        if (emit_slots) ring_row[at] = MERGE ? ring_place0 + 64u * j : row;
                    }
                    tail += (uint32_t)__popcll(mask);
                    while (tail - head >= 64) {                  // (a round may put unsettled lanes back: at most as many as it took off)
#ifdef PG_MINI_STAMPS
                        const unsigned long long ts = __builtin_amdgcn_s_memtime();
#endif
                        slow_round(true);
                        head += 64;
#ifdef PG_MINI_STAMPS
                        if (lane == 0) atomicAdd(&dbg[6], (unsigned long long)(__builtin_amdgcn_s_memtime() - ts));     // per-wave time in the general insert
#endif
                    }
                }
            }
            PG_WLAP(3);                                          // (ring pushes, general inserts)
            if (emit_slots) {
                if (MERGE) {
                    // fixed places: halfword row j of the batch = the slots of the j-th k-mers of its 64 records; a k-mer that the
                    // first probe did not settle gets its slot from the general insert, one outside every row gets none
#pragma unroll
                    for (int j = 0; j < CX; ++j)
                        if (__builtin_amdgcn_inverse_ballot_w64(pm[j]))
                            gstore(prov_h, (uint64_t)place0 + 64u * j, (uint64_t)np_half, (uint16_t)sl[j], status);
                } else
                // the words, slot by slot (neighbours in the buffer come from different records: the row histograms behind the
                // shuffle do not like runs of equal words).  (Deferring these stores to the next iteration's top, behind its
                // vmcnt wait, changed nothing: 17.33 ms either way.)
                {
#pragma unroll
                for (int j = 0; j < CX; ++j) {
                    if (__builtin_amdgcn_inverse_ballot_w64(pm[j])) gstore(prov_b, (uint64_t)at + lanes_below(pm[j]), (uint64_t)np_all, (row << lb) | sl[j], status);
                    at += (uint32_t)__popcll(pm[j]);
                }
                }
            }
            PG_WLAP(4);                                          // (the words' stores, waited for)
        }
    };
    count_range(std::integral_constant<int, (CAP > SHORT_MAX ? SHORT_MAX : CAP)>{}, r0, rs, first_s, first_ms);
    count_range(std::integral_constant<int, CAP>{}, rs, r1, first_l, first_ml);
    while (tail != head) {                                       // what is left on the ring
        const uint32_t left = tail - head < 64u ? tail - head : 64u;
        slow_round(lane < left);
        head += left;
    }
#ifdef PG_MINI_STAMPS
    if (lane == 0) {
        atomicAdd(&dbg[5], (unsigned long long)(__builtin_amdgcn_s_memtime() - st0));     // per-wave end of the count loop
        for (int q = 0; q < 5; ++q) atomicAdd(&dbg[32 + q], wacc[q]);
    }
#endif
    PG_STAMP(1);
    if (full) atomicOr(status, 1u);
    if (window && !emit_slots) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) mine += __shfl_down(mine, d);
        if (lane == 0 && mine) atomicAdd(&n_lookups, mine);
    }
    __syncthreads();
    if (HALF && hv.accum) {
        // a piece: the slice back (counts clamped: the adds stop at SAT, the overshoot of concurrent ones is cut here)
        const uint64_t slice0 = (uint64_t)blockIdx.x << t.log2_bucket;
        for (uint32_t i = threadIdx.x; i < n_slots; i += BLK) {
            const unsigned long long v = tab[i];
            uint32_t c = (uint32_t)(v & HASH_CMASK);
            if (c > HASH_SAT) c = HASH_SAT;
            gstore(t.slots, slice0 + i, 1ull << t.log2_slots, v ? (v & ~(unsigned long long)HASH_CMASK) | c : 0ull, status);
        }
        if (threadIdx.x == 0) wbeg[blockIdx.x] = wbase;
        return;
    }
    if (HALF) {
        // the bucket's occupied slots, in slot order: a bitmap of the occupancy (64 consecutive slots = the lanes of one wavefront:
        // a ballot), ranks from the popcounts of its words, the entries written to consecutive places (coalesced).  The LDS table
        // stays as it is until every slot has been read (the bitmap and its scan live behind it, where the rings were).
        unsigned long long *occ_l = tab + tab_units;             // [n_slots / 64] occupancy words, then as many exclusive ranks (uint32)
        const uint32_t n_occ = n_slots >= 64u ? n_slots >> 6 : 1u;
        uint32_t *rank_l = reinterpret_cast<uint32_t *>(occ_l + n_occ);
        __shared__ uint32_t n_entries;
        unsigned long long v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const uint32_t i = q * BLK + threadIdx.x;
            unsigned long long x = i < n_slots ? tab[i] : 0ull;
            if (x) {                                             // (the count stops at SAT; the overshoot of concurrent adds is clamped here)
                uint32_t c = (uint32_t)(x & HASH_CMASK);
                if (c > HASH_SAT) c = HASH_SAT;
                x = (x & ~(unsigned long long)HASH_CMASK) | c;
            }
            v[q] = x;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(x != 0ull);
            if (lane == 0 && q * BLK + wave * 64u < (n_slots >= 64u ? n_slots : 64u)) occ_l[(q * BLK + wave * 64u) >> 6] = m;
        }
        __syncthreads();
        if (threadIdx.x < 64) {                                  // exclusive ranks of the occupancy words (at most 256 of them), by one wavefront
            uint32_t run = 0;
            for (uint32_t w0 = 0; w0 < n_occ; w0 += 64) {
                const uint32_t wi = w0 + threadIdx.x;
                const uint32_t c = wi < n_occ ? (uint32_t)__popcll(occ_l[wi]) : 0u;
                uint32_t incl = c;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t o = __shfl_up(incl, d);
                    if ((int)threadIdx.x >= d) incl += o;
                }
                if (wi < n_occ) rank_l[wi] = run + incl - c;
                run += __shfl(incl, 63);
            }
            if (threadIdx.x == 0) n_entries = run;
        }
        __syncthreads();
        // (capacities for the checked build: the slabs of all buckets of this launch)
        const uint64_t ent0 = (uint64_t)blockIdx.x << t.log2_bucket, ent_cap = (uint64_t)gridDim.x << t.log2_bucket;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const uint32_t i = q * BLK + threadIdx.x;
            if (i < n_slots && v[q]) {
                const unsigned long long m = occ_l[i >> 6];
                gstore(hv.ent, ent0 + rank_l[i >> 6] + (uint32_t)__popcll(m & ((1ull << (i & 63u)) - 1ull)), ent_cap, v[q], status);
            }
        }
        for (uint32_t wi = threadIdx.x; wi < n_occ; wi += BLK) gstore(hv.occ, (uint64_t)blockIdx.x * n_occ + wi, (uint64_t)gridDim.x * n_occ, occ_l[wi], status);
        if (threadIdx.x == 0) {
            hv.fill[blockIdx.x] = (long long)n_entries;
            hv.ring_cnt[blockIdx.x] = emitted_ring;
            wbeg[blockIdx.x] = wbase;
        }
        return;
    }
    if (emit_slots) {
        // the packed slice (an empty table needs no clearing: every slot is written), and the table shrinks to 2-byte bins:
        // 0 = slot never filled, 0xffff = bin out of range, else bin + 1.  (Every lane first reads all its slots -- the bins
        // land on top of the first quarter of the table.)
        uint16_t mybin[16];
        const float rcp_window = 1.0f / (float)window;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const uint32_t i = q * BLK + threadIdx.x;
            mybin[q] = 0;
            if (i < n_slots) {
                const unsigned long long v = tab[i];
                uint32_t c;
                if (WIDE) {
                    c = cnts[i];
                    slice[i] = v;
                    slice_counts[i] = c;
                } else {
                    c = (uint32_t)(v & HASH_CMASK);
                    if (c > HASH_SAT) c = HASH_SAT;
                    slice[i] = v ? (v & ~(unsigned long long)HASH_CMASK) | c : 0ull;
                }
                const uint32_t bin = div_uniform(c, window, rcp_window);
                // (the merged lookups read the bin itself, 0xffff = none; the word-wise ones bin + 1, 0 = slot never filled)
                mybin[q] = MERGE ? (uint16_t)(v && bin < vsize ? bin : 0xffffu) : (uint16_t)(v ? (bin < vsize ? bin + 1u : 0xffffu) : 0u);
            }
        }
        __syncthreads();
        uint16_t *bins16 = reinterpret_cast<uint16_t *>(tab);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const uint32_t i = q * BLK + threadIdx.x;
            if (i < n_slots) bins16[i] = mybin[q];
        }
        PG_STAMP(2);
        // ---- provisional (row, slot) words -> (row, bin) words, scattered at once by the first digit of their row group into
        // the row shuffle's regions (an LDS multisplit per tile of 16 Ki words, one global cursor add per digit and tile): the
        // final words never make a trip of their own through HBM
        unsigned char *lds = reinterpret_cast<unsigned char *>(tab);
        const uint32_t np = (uint32_t)n_lookups;
        if (MERGE) {
            MergeCtx mc;
            mc.lds = lds; mc.smask = smask; mc.vbits = vbits; mc.sh = sh; mc.status = status;
            mc.prov_b = prov_b; mc.meta_s = meta + r0; mc.meta_l = meta + rs; mc.n_s = (uint32_t)(rs - r0); mc.n_l = (uint32_t)(r1 - rs);
#ifdef PG_MINI_STAMPS
            mc.dbg = dbg;
#endif
            merged_lookup<CAP, BLK, DIG>(mc);
            PG_STAMP(3);
#ifdef PG_MINI_GAPS
            __syncthreads();
            if (threadIdx.x == 0 && n_short) {             // (n_short: there was a second pass, `bases` is buffer B)
                unsigned long long *gap = (unsigned long long *)(bases - rec_cap) + 4ull * blockIdx.x;     // (the first pass's buffer: dead by now)
                gap[0] = gap_t0;
                gap[1] = __builtin_amdgcn_s_memtime();
                gap[2] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 32);    // HW_ID, XCC_ID
                gap[3] = (unsigned long long)(r1 - r0);
            }
#endif
            return;
        }
        {
            WordCtx wc;
            wc.lds = lds; wc.smask = smask; wc.np = np; wc.lb = lb; wc.vbits = vbits; wc.sh = sh; wc.status = status; wc.prov_b = prov_b;
#ifdef PG_MINI_STAMPS
            wc.dbg = dbg;
#endif
            wordwise_lookup<BLK, DIG>(wc);
        }
        PG_STAMP(3);
        return;
    }
    // the packed slice (an empty table needs no clearing: every slot is written); then counts -> bins, in place
    const float rcp_window_g = window ? 1.0f / (float)window : 0.0f;
    for (uint32_t i = threadIdx.x; i < n_slots; i += BLK) {
        const unsigned long long v = tab[i];
        if (WIDE) {
            const uint32_t c = cnts[i];
            slice[i] = v;
            slice_counts[i] = c;
            if (window && v) {
                const uint32_t bin = div_uniform(c, window, rcp_window_g);
                cnts[i] = bin < vsize ? bin + 1u : BIN_NONE;
            }
            continue;
        }
        uint32_t c = (uint32_t)(v & HASH_CMASK);
        if (c > HASH_SAT) c = HASH_SAT;
        slice[i] = v ? (v & ~(unsigned long long)HASH_CMASK) | c : 0ull;
        if (window && v) {
            const uint32_t bin = div_uniform(c, window, rcp_window_g);
            tab[i] = (v & ~(unsigned long long)HASH_CMASK) | (bin < vsize ? bin + 1u : BIN_NONE);
        }
    }
    PG_STAMP(2);
    if (!window) return;
    // this bucket's words go to a range of the word buffer claimed with one global add: an upper bound, every k-mer inside a row
    if (threadIdx.x == 0) {
        wbase = atomicAdd(word_cursor, n_lookups);
        wbeg[blockIdx.x] = wbase;
    }
    __syncthreads();
    wb = wbase;
    // ---- general form: the records are read and probed a second time
    // (row, bin) words of 64 lanes -> the bucket's word range: one LDS add per call, the stores are contiguous
    auto emit = [&](bool put, uint32_t row, uint32_t bin1) {
        const unsigned long long pm = __ballot(put);
        if (pm) {
            uint32_t at = 0;
            if (lane == 0) at = atomicAdd(&emitted, (uint32_t)__popcll(pm));
            at = __shfl(at, 0);
            if (put) words[wb + at + lanes_below(pm)] = (row << vbits) | (bin1 - 1u);
        }
    };
    {
        head = tail = 0;
        int64_t i = r0 + (int64_t)wave * 64 + lane;
        uint64_t R = i < r1 ? bases[i] : 0ull;
        uint32_t m = i < r1 ? meta[i] : 0xffffffffu;
        for (int64_t i0 = r0 + (int64_t)wave * 64; i0 < r1; i0 += BLK) {
            const bool live = i0 + lane < r1;
            const uint32_t row = m >> META_ROW_SHIFT;
            const int n = live && row != MINI_ROW_NONE ? (int)((m >> META_D2_BITS) & (MINI_MAX_LEN - 1)) + 1 : 0;
            const uint64_t FW = rev2_64(R), RC = R ^ 0xAAAAAAAAAAAAAAAAull;
            i = i0 + BLK + lane;
            R = i < r1 ? bases[i] : 0ull;
            m = i < r1 ? meta[i] : 0xffffffffu;
            uint64_t code[CAP];
            unsigned long long cur[CAP];
            uint32_t bin1[CAP];                                  // count field of a k-mer found at its home slot: bin + 1, or BIN_NONE
#pragma unroll
            for (int j = 0; j < CAP; ++j) {
                const uint64_t fw = (FW >> (2 * j)) & kmask, rc = (RC >> (rc_sh0 - 2 * j)) & kmask;
                code[j] = fw < rc ? fw : rc;
                const uint32_t sl = mini_slot_hash<WIDE>(code[j]) & smask;
                cur[j] = j < n ? tab[sl] : 0ull;
                bin1[j] = WIDE ? (j < n ? cnts[sl] : 0u) : 0u;
            }
            // settled by the first probe: a hit (emit, unless its bin is out of range) or an empty slot (cannot happen for a
            // counted k-mer; nothing to emit).  Slot by slot, ONE add on the bucket's counter per batch.
            unsigned long long pm[CAP];
            uint32_t total = 0;
#pragma unroll
            for (int j = 0; j < CAP; ++j) {
                const bool hit = j < n && (WIDE ? cur[j] == code[j] + 1ull : cur[j] != 0 && (cur[j] >> HASH_CBITS) == code[j]);
                if (!WIDE) bin1[j] = (uint32_t)(cur[j] & HASH_CMASK);
                pm[j] = __ballot(hit && bin1[j] != BIN_NONE);
                total += (uint32_t)__popcll(pm[j]);
            }
#ifdef PG_MINI_STAMPS
            {   // how many words would be left if equal neighbouring words of a record travelled as one (dbg[10] words, dbg[11] runs)
                uint32_t nw = 0, nr = 0;
#pragma unroll
                for (int j = 0; j < CAP; ++j) {
                    const bool wj = (pm[j] >> lane) & 1ull;
                    const bool same = j > 0 && ((pm[j > 0 ? j - 1 : 0] >> lane) & 1ull) && bin1[j] == bin1[j > 0 ? j - 1 : 0];
                    nw += wj;
                    nr += wj && !same;
                }
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) { nw += __shfl_down(nw, d); nr += __shfl_down(nr, d); }
                if (lane == 0) { atomicAdd(&dbg[10], (unsigned long long)nw); atomicAdd(&dbg[11], (unsigned long long)nr); }
            }
#endif
            if (total) {                                         // (uniform)
                uint32_t at = 0;
                if (lane == 0) at = atomicAdd(&emitted, total);
                at = __shfl(at, 0);
#pragma unroll
                for (int j = 0; j < CAP; ++j) {
                    if ((pm[j] >> lane) & 1ull) words[wb + at + lanes_below(pm[j])] = (row << vbits) | (bin1[j] - 1u);
                    at += (uint32_t)__popcll(pm[j]);
                }
            }
#pragma unroll
            for (int j = 0; j < CAP; ++j) {                      // collisions: onto the ring
                const bool pend = j < n && cur[j] != 0 && (WIDE ? cur[j] != code[j] + 1ull : (cur[j] >> HASH_CBITS) != code[j]);
                const unsigned long long mask = __ballot(pend);
                if (mask) {
                    if (pend) {
                        const uint32_t at = (tail + lanes_below(mask)) & (RING - 1);
                        ring[at] = code[j];
                        ring_row[at] = row;
                    }
                    tail += (uint32_t)__popcll(mask);
                    if (tail - head >= 64) {
                        const uint32_t at = (head + lane) & (RING - 1);
                        const uint64_t c = ring[at];
                        const uint32_t rw = ring_row[at];
                        head += 64;
                        const uint32_t b1 = mini_lookup_slow<WIDE>(tab, cnts, smask, limit, c, true);
                        emit(b1 != BIN_NONE, rw, b1);
                    }
                }
            }
        }
        if (tail != head) {
            const bool act = lane < tail - head;
            const uint32_t at = (head + lane) & (RING - 1);
            const uint64_t c = act ? ring[at] : 0ull;
            const uint32_t rw = act ? ring_row[at] : 0u;
            const uint32_t b1 = mini_lookup_slow<WIDE>(tab, cnts, smask, limit, c, act);
            emit(b1 != BIN_NONE, rw, b1);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) emit_end[blockIdx.x] = wb + emitted;
}

// ---- N > 1 ranks: the other kernels of the super-k-mer form (pangaea_amd/dist.py: features_sharded_mini)
//
// rank:   plan -> A1' -> A2' -> mini_count_kernel<..., HALF>     entries + occupancy per bucket, provisional words
//         mini_gather_entries_kernel                              entries, bucket-ordered, into the send buffer (owner ranges)
//         ---- all-to-all: every bucket range to its owner (8 bytes per entry)
// owner:  mini_merge_bins_kernel                                  per owned bucket: the N parts summed inside LDS -> its slice of the
//                                                                 global table, and for EVERY entry received its bin (2 bytes), in order
//         ---- all-to-all back (2 bytes per entry)
// rank:   mini_lookup_half_kernel                                 bins + occupancy -> the bucket's 2-byte bins in LDS -> lookups of the
//                                                                 provisional words (as the one-GPU kernel does), row-group scatter
__global__ __launch_bounds__(BLOCK) void mini_gather_entries_kernel(const unsigned long long *__restrict__ ent, int log2_bucket,
                                                                    const long long *__restrict__ fill, const long long *__restrict__ dst_elem,
                                                                    unsigned long long *__restrict__ out, unsigned long long out_cap, uint32_t *status)
{
    const unsigned long long *src = ent + ((uint64_t)blockIdx.x << log2_bucket);
    const long long n = fill[blockIdx.x], d0 = dst_elem[blockIdx.x];
    for (long long i = threadIdx.x; i < n; i += BLOCK) gstore(out, (uint64_t)(d0 + i), out_cap, src[i], status);
}

// one workgroup per OWNED bucket: part p's entries of owned bucket i lie at recv[p * part_stride + seg[p * (n_owned + 1) + i] ..
// seg[... + i + 1]) (slot format: canonical code << 22 | count, counts <= SAT); bins_out has the same layout in uint16:
// bin + 1 of the entry's k-mer in the MERGED table, 0xffff when the bin lies beyond the vector
__global__ __launch_bounds__(BIG_BLOCK) void mini_merge_bins_kernel(const unsigned long long *__restrict__ recv, long long part_stride,
                                                                    const long long *__restrict__ seg, int n_parts, int n_owned,
                                                                    MiniView t, long long bucket0, uint32_t window, uint32_t vsize,
                                                                    uint16_t *__restrict__ bins_out, uint32_t *status)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long tab[];
    const uint32_t n_slots = 1u << t.log2_bucket, smask = n_slots - 1u;
    const uint32_t limit = n_slots < MAX_PROBE ? n_slots : MAX_PROBE;
    for (uint32_t i = threadIdx.x; i < n_slots; i += BIG_BLOCK) tab[i] = 0ull;
    __syncthreads();
    bool full = false;
    // Insert: four entries of the lane in flight (their loads, then their first probes, issued together); the slot an entry ends up
    // in is left where its bin will go -- the second sweep then needs neither the entry nor its hash nor a probe, only the slot's
    // final count (until round 4 it read the 8-byte entries again and probed again: 3.16 ms for the 8-rank geometry)
    constexpr int U = 4;
    const uint64_t out_cap = (uint64_t)n_parts * (uint64_t)part_stride;
    for (int p = 0; p < n_parts; ++p) {
        const long long a = seg[(long long)p * (n_owned + 1) + blockIdx.x], b = seg[(long long)p * (n_owned + 1) + blockIdx.x + 1];
        for (long long e0 = a + threadIdx.x; e0 < b; e0 += (long long)U * BIG_BLOCK) {
            unsigned long long x[U], cur[U];
            uint32_t s[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long long e = e0 + (long long)u * BIG_BLOCK;
                x[u] = e < b ? recv[(long long)p * part_stride + e] : 0ull;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                s[u] = mini_slot_hash<false>(x[u] >> HASH_CBITS) & smask;
                cur[u] = tab[s[u]];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long long e = e0 + (long long)u * BIG_BLOCK;
                if (e >= b) continue;
                const uint64_t code = x[u] >> HASH_CBITS;
                const uint32_t c = (uint32_t)(x[u] & HASH_CMASK);
                uint32_t at = s[u];
                unsigned long long cv = cur[u];
                bool done = false;
                for (uint32_t i = 0; i < limit && !done; ++i) {
                    if (i) cv = tab[at];
                    if (cv == 0ull) {
                        cv = atomicCAS(&tab[at], 0ull, x[u]);
                        if (cv == 0ull) { done = true; break; }
                    }
                    if ((cv >> HASH_CBITS) == code) {            // add, saturating: parts of up to SAT each must not carry into the code
                        for (;;) {
                            const uint32_t sum = (uint32_t)(cv & HASH_CMASK) + c;
                            const unsigned long long nv = (cv & ~(unsigned long long)HASH_CMASK) | (sum > HASH_SAT ? HASH_SAT : sum);
                            const unsigned long long old = atomicCAS(&tab[at], cv, nv);
                            if (old == cv) break;
                            cv = old;
                        }
                        done = true;
                        break;
                    }
                    at = (at + 1) & smask;
                }
                full |= !done;
                // (a k-mer that found no place -- the table is full and reported so -- has no slot and no bin)
                gstore(bins_out, (uint64_t)((long long)p * part_stride + e), out_cap, (uint16_t)(done ? at : 0xffffu), status);
            }
        }
    }
    if (full) atomicOr(status, PG_STATUS_TABLE_FULL);
    __syncthreads();
    const uint64_t slice0 = (uint64_t)(bucket0 + blockIdx.x) << t.log2_bucket;
    for (uint32_t i = threadIdx.x; i < n_slots; i += BIG_BLOCK) gstore(t.slots, slice0 + i, 1ull << t.log2_slots, tab[i], status);
    // slots -> bins (every lane reads back what it wrote itself: the same entries in the same order)
    const float rcp_window = 1.0f / (float)window;
    for (int p = 0; p < n_parts; ++p) {
        const long long a = seg[(long long)p * (n_owned + 1) + blockIdx.x], b = seg[(long long)p * (n_owned + 1) + blockIdx.x + 1];
        for (long long e0 = a + threadIdx.x; e0 < b; e0 += (long long)U * BIG_BLOCK) {
            uint32_t sl[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long long e = e0 + (long long)u * BIG_BLOCK;
                sl[u] = e < b ? (uint32_t)bins_out[(long long)p * part_stride + e] : 0xffffu;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long long e = e0 + (long long)u * BIG_BLOCK;
                if (e >= b) continue;
                uint32_t out = 0xffffu;
                if (sl[u] != 0xffffu) {
                    const uint32_t bin = div_uniform((uint32_t)(tab[sl[u] & smask] & HASH_CMASK), window, rcp_window);
                    out = bin < vsize ? bin + 1u : 0xffffu;
                }
                gstore(bins_out, (uint64_t)((long long)p * part_stride + e), out_cap, (uint16_t)out, status);
            }
        }
    }
}

// (the word-wise lookups; the merged form -- merged_lookup with the hit planes of a MERGE count half -- plugs in here once it
// is the faster one)
template <int BLK, int DIG>
__global__ __launch_bounds__(BLK, 4) void mini_lookup_half_kernel(const unsigned long long *__restrict__ kwords, const unsigned long long *__restrict__ wbeg,
                                                                const unsigned long long *__restrict__ occ,
                                                                const uint16_t *__restrict__ bins_in, const long long *__restrict__ bin_elem,
                                                                int log2_bucket, int vbits, const uint32_t *__restrict__ prov, ShufArgs sh, uint32_t *status)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    if (*status & PG_STATUS_OVERFLOW_LIST) return;               // (the exchange did not take place: dist.MiniSharded.exchange; no bins to read)
    const uint32_t n_slots = 1u << log2_bucket, smask = n_slots - 1u;
    const uint32_t n_occ = n_slots >= 64u ? n_slots >> 6 : 1u;
    const uint32_t np_all = (uint32_t)kwords[blockIdx.x];
    if (np_all == 0u) return;                                    // (uniform: no k-mer of this bucket lies inside a row)
    // the bucket's 2-byte bins: slot i holds the entry of rank (occupied slots below i), whose bin came back in that order.
    // Occupancy words and their exclusive ranks go to LDS first (behind the bins and the lookup's areas: the tail of the tile
    // buffer, which the lookups only use later), two barriers in all
    uint16_t *bins16 = reinterpret_cast<uint16_t *>(lds);
    const uint16_t *bi = bins_in + bin_elem[blockIdx.x];
    unsigned long long *occ_l = reinterpret_cast<unsigned long long *>(lds + LookupLds<BLK, DIG>::BUF);          // [n_occ] words, then [n_occ] ranks
    uint32_t *rank_l = reinterpret_cast<uint32_t *>(occ_l + n_occ);
    for (uint32_t wi = threadIdx.x; wi < n_occ; wi += BLK) occ_l[wi] = occ[(uint64_t)blockIdx.x * n_occ + wi];
    __syncthreads();
    if (threadIdx.x < 64) {
        uint32_t run = 0;
        for (uint32_t w0 = 0; w0 < n_occ; w0 += 64) {
            const uint32_t wi = w0 + threadIdx.x;
            const uint32_t c = wi < n_occ ? (uint32_t)__popcll(occ_l[wi]) : 0u;
            uint32_t incl = c;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = __shfl_up(incl, d);
                if ((int)threadIdx.x >= d) incl += o;
            }
            if (wi < n_occ) rank_l[wi] = run + incl - c;
            run += __shfl(incl, 63);
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_slots; i += BLK) {
        const unsigned long long m = occ_l[i >> 6];
        bins16[i] = (m >> (i & 63u)) & 1ull ? bi[rank_l[i >> 6] + (uint32_t)__popcll(m & ((1ull << (i & 63u)) - 1ull))] : (uint16_t)0;
    }
    __syncthreads();
    WordCtx wc;
    wc.lds = lds; wc.smask = smask; wc.np = np_all; wc.lb = log2_bucket; wc.vbits = vbits; wc.sh = sh; wc.status = status;
    wc.prov_b = prov + wbeg[blockIdx.x];
    wc.dbg = nullptr;
    wordwise_lookup<BLK, DIG>(wc);
}

// the merged form of the lookup half: the count half left its provisional words in fixed slots (MergeArgs)
template <int CAP, int BLK, int DIG>
__global__ __launch_bounds__(BLK, 4) void mini_lookup_half_merge_kernel(const unsigned long long *__restrict__ off, const unsigned long long *__restrict__ n_short,
                                                                      const unsigned long long *__restrict__ wbeg, const uint32_t *__restrict__ ring_cnt,
                                                                      const unsigned long long *__restrict__ occ,
                                                                      const uint16_t *__restrict__ bins_in, const long long *__restrict__ bin_elem,
                                                                      int log2_bucket, int vbits, const uint32_t *__restrict__ mprov,
                                                                      const uint32_t *__restrict__ meta, ShufArgs sh, uint32_t *status)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    if (*status & PG_STATUS_OVERFLOW_LIST) return;               // (the exchange did not take place: dist.MiniSharded.exchange; no bins to read)
    const uint32_t n_slots = 1u << log2_bucket, smask = n_slots - 1u;
    const uint32_t n_occ = n_slots >= 64u ? n_slots >> 6 : 1u;
    const int64_t r0 = (int64_t)off[blockIdx.x], r1 = (int64_t)off[blockIdx.x + 1];
    const int64_t rs = n_short && CAP > SHORT_MAX ? r0 + (int64_t)n_short[blockIdx.x] : r0;
    uint16_t *bins16 = reinterpret_cast<uint16_t *>(lds);
    const uint16_t *bi = bins_in + bin_elem[blockIdx.x];
    unsigned long long *occ_l = reinterpret_cast<unsigned long long *>(lds + MergeLds<BLK, DIG>::BUF);          // [n_occ] words, then [n_occ] ranks
    uint32_t *rank_l = reinterpret_cast<uint32_t *>(occ_l + n_occ);
    for (uint32_t wi = threadIdx.x; wi < n_occ; wi += BLK) occ_l[wi] = occ[(uint64_t)blockIdx.x * n_occ + wi];
    __syncthreads();
    if (threadIdx.x < 64) {
        uint32_t run = 0;
        for (uint32_t w0 = 0; w0 < n_occ; w0 += 64) {
            const uint32_t wi = w0 + threadIdx.x;
            const uint32_t c = wi < n_occ ? (uint32_t)__popcll(occ_l[wi]) : 0u;
            uint32_t incl = c;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = __shfl_up(incl, d);
                if ((int)threadIdx.x >= d) incl += o;
            }
            if (wi < n_occ) rank_l[wi] = run + incl - c;
            run += __shfl(incl, 63);
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_slots; i += BLK) {
        const unsigned long long m = occ_l[i >> 6];
        // (what travels is bin + 1, 0xffff = beyond the vector; merged_lookup reads the bin itself, 0xffff = none)
        const uint16_t x = (m >> (i & 63u)) & 1ull ? bi[rank_l[i >> 6] + (uint32_t)__popcll(m & ((1ull << (i & 63u)) - 1ull))] : (uint16_t)0;
        bins16[i] = (uint16_t)(x != 0 && x != 0xffffu ? x - 1u : 0xffffu);
    }
    __syncthreads();
    MergeCtx mc;
    mc.lds = lds; mc.smask = smask; mc.vbits = vbits; mc.sh = sh; mc.status = status;
    mc.prov_b = mprov + wbeg[blockIdx.x];
    mc.meta_s = meta + r0; mc.meta_l = meta + rs; mc.n_s = (uint32_t)(rs - r0); mc.n_l = (uint32_t)(r1 - rs);
    mc.dbg = nullptr;
    merged_lookup<CAP, BLK, DIG>(mc);
}

// PIECES: the lookups of ONE piece's provisional slots in the final table -- the bucket's slice becomes its 2-byte bins in LDS
// (as the fused kernel makes them), then merged_lookup on the piece's slots and the kept meta words of its records
template <int CAP, int BLK, int DIG>
__global__ __launch_bounds__(BLK, 4) void mini_lookup_slice_merge_kernel(const unsigned long long *__restrict__ off, const unsigned long long *__restrict__ n_short,
                                                                       const unsigned long long *__restrict__ wbeg, MiniView t, uint32_t window, uint32_t vsize,
                                                                       int vbits, const uint32_t *__restrict__ mprov, const uint32_t *__restrict__ meta,
                                                                       ShufArgs sh, uint32_t *status)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const uint32_t n_slots = 1u << t.log2_bucket, smask = n_slots - 1u;
    const int64_t r0 = (int64_t)off[blockIdx.x], r1 = (int64_t)off[blockIdx.x + 1];
    if (r1 == r0) return;                                        // (uniform: the piece has no record of this bucket)
    const int64_t rs = n_short && CAP > SHORT_MAX ? r0 + (int64_t)n_short[blockIdx.x] : r0;
    uint16_t *bins16 = reinterpret_cast<uint16_t *>(lds);
    const uint64_t *slice = t.slots + ((uint64_t)blockIdx.x << t.log2_bucket);
    const float rcp_window = 1.0f / (float)window;
    for (uint32_t i = threadIdx.x; i < n_slots; i += BLK) {
        const unsigned long long v = slice[i];
        const uint32_t bin = div_uniform((uint32_t)(v & HASH_CMASK), window, rcp_window);
        bins16[i] = (uint16_t)(v && bin < vsize ? bin : 0xffffu);        // (merged_lookup reads the bin itself, 0xffff = none)
    }
    __syncthreads();
    MergeCtx mc;
    mc.lds = lds; mc.smask = smask; mc.vbits = vbits; mc.sh = sh; mc.status = status;
    mc.prov_b = mprov + wbeg[blockIdx.x];
    mc.meta_s = meta + r0; mc.meta_l = meta + rs; mc.n_s = (uint32_t)(rs - r0); mc.n_l = (uint32_t)(r1 - rs);
    mc.dbg = nullptr;
    merged_lookup<CAP, BLK, DIG>(mc);
}

// ---- workspace of the plan: header | region_tot | region_off | off | hist | cur2 | cur2l | kwords | wbeg | round_row | chunk table
struct MiniPlan {
    int bits, bits1, bits2;
    int64_t n_rounds, n_chunks, chunk_stride;
    size_t header_off, rtot_off, roff_off, hist_off, off_off, cur2_off, cur2l_off, kw_off, wbeg_off, round_off, chunk_off, total;
};

// k-mers per record at most: what fits the 32 characters of a record and the 4-bit length field -- and the window: a
// minimizer covers at most W = k - M + 1 consecutive k-mers, so longer runs only exist where the same hashed M-mer recurs
// (homopolymers, short tandem repeats).  The bucket workgroups treat every record in `cap` unrolled, predicated steps: cutting
// those rare runs at W costs a few records and saves a quarter of the steps (k = 21: 12 -> 9; mini_count 19.1 -> 18.5 ms).
// PG_MINI_CAP overrides (tuning).
static_assert(MINI_MAX_WINDOW <= 9, "count kernels are instantiated for up to 9 k-mers per record");
int mini_cap(int k)
{
    int wc, woff;
    mini_window(k, &wc, &woff);
    int cap = 33 - k < MINI_MAX_LEN ? 33 - k : MINI_MAX_LEN;
    if (wc < cap) cap = wc;
    static const int forced = getenv("PG_MINI_CAP") ? atoi(getenv("PG_MINI_CAP")) : 0;
    if (forced >= 1 && forced < cap) cap = forced;
    return cap;
}

int check_mini(const pg_table *t, const char *who)
{
    if (!t || !t->data) return pg_fail(PG_EINVAL, "%s: table descriptor is null", who);
    if (t->kind != PG_TABLE_MINI && t->kind != PG_TABLE_MINI_WIDE) return pg_fail(PG_EINVAL, "%s: needs a PG_TABLE_MINI or PG_TABLE_MINI_WIDE table", who);
    const bool wide = t->kind == PG_TABLE_MINI_WIDE;
    const int k_lo = wide ? PG_HASH_MAX_K + 1 : PG_MINI_MIN_K, k_hi = wide ? PG_WIDE_MAX_K : PG_HASH_MAX_K;
    if (t->k < k_lo || t->k > k_hi) return pg_fail(PG_EINVAL, "%s: %s tables need %d <= k <= %d (got %d)", who, wide ? "wide mini" : "mini", k_lo, k_hi, t->k);
    const int lb_hi = wide ? PG_MINI_WIDE_MAX_LOG2_BUCKET_SLOTS : PG_BUCKET_MAX_LOG2_SLOTS;
    if (t->log2_bucket_slots < 4 || t->log2_bucket_slots > lb_hi)
        return pg_fail(PG_EINVAL, "%s: log2_bucket_slots %d out of range [4,%d]", who, t->log2_bucket_slots, lb_hi);
    const int bits = t->log2_slots - t->log2_bucket_slots;
    if (bits < 0 || bits > PG_MINI_MAX_LOG2_BUCKETS) return pg_fail(PG_EINVAL, "%s: 2^%d buckets (at most 2^%d)", who, bits, PG_MINI_MAX_LOG2_BUCKETS);
    return PG_OK;
}

int plan_mini(const pg_table *t, int64_t n_words, MiniPlan *p)
{
    p->bits = t->log2_slots - t->log2_bucket_slots;
    p->bits1 = p->bits < MINI_BITS1 ? p->bits : MINI_BITS1;
    if (p->bits - p->bits1 > META_D2_BITS) p->bits1 = p->bits - META_D2_BITS;       // (cannot happen: at most 2^16 buckets)
    p->bits2 = p->bits - p->bits1;
    p->n_rounds = (n_words + ROUND_WORDS - 1) / ROUND_WORDS;
    p->n_chunks = (n_words + MINI_CHUNK_WORDS - 1) / MINI_CHUNK_WORDS;
    if (p->n_chunks < 1) p->n_chunks = 1;
    // golden-ratio stride, made coprime to the chunk count: regions hold the stream in scrambled chunk order, so the bucket
    // workgroups (and the row shuffle behind them) do not all see the same rows at the same time
    p->chunk_stride = (int64_t)((double)p->n_chunks * 0.6180339887) | 1;
    auto gcd = [](int64_t a, int64_t b) { while (b) { int64_t r = a % b; a = b; b = r; } return a; };
    while (gcd(p->chunk_stride, p->n_chunks) != 1) p->chunk_stride += 2;
    const size_t nb = (size_t)1 << p->bits;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += (bytes + 255) / 256 * 256; return at; };
    p->header_off = take(512);
    p->rtot_off = take(((size_t)1 << MINI_MAX_BITS1) * 8);          // plan: records per region
    p->roff_off = take((((size_t)1 << MINI_MAX_BITS1) + 1) * 8);    // plan: where every region starts
    p->off_off = take((nb + 1) * 8);                                // count: where every bucket starts
    p->hist_off = take(nb * 8);                                     // count: records per bucket (cleared with the cursors behind it)
    p->cur2_off = take(nb * 8);
    p->cur2l_off = take(nb * 8);
    p->kw_off = take(nb * 8);
    p->wbeg_off = take(nb * 8);
    p->round_off = take((size_t)(p->n_rounds + 1) * 4);
    p->chunk_off = take(((size_t)p->n_chunks << p->bits1) * 8);
    p->total = o;
    return PG_OK;
}

MiniView mini_view(const pg_table *t)
{
    MiniView v;
    v.slots = (uint64_t *)t->data;
    v.log2_slots = t->log2_slots;
    v.log2_bucket = t->log2_bucket_slots;
    v.k = t->k;
    return v;
}

// the SLOTS form of the lookups needs a row and a slot index in one 32-bit word (PG_MINI_PROBE_TWICE forces the general form)
// row groups that the count kernel's own scatter reaches in one pass: 2^11 (131 072 rows; 2048 digits in its lookup tiles)
constexpr int MINI_ONE_PASS_BITS = 11;
bool mini_slots_form(const pg_table *t, const pg_rows *rows)
{
    return rows && rows->n_rows > 0 && rows->n_rows < ((int64_t)1 << (32 - t->log2_bucket_slots)) - 1 && !getenv("PG_MINI_PROBE_TWICE");
}

// merged lookups (mini_count_kernel<..., MERGE>): the slot form, with room for the run length on top of row and bin.  The default
// wherever it applies and the caller gave pg_mini_count the word buffer it needs (PG_MINI_MERGE=0: the word-wise form): on
// config 2 the step is 0.3-2 ms shorter (box to box), the row histograms see 0.28 of the words
bool mini_merge_form(const pg_table *t, const pg_rows *rows, int vsize)
{
    const char *want = getenv("PG_MINI_MERGE");
    if ((want && atoi(want) == 0) || getenv("PG_MINI_NO_MERGE")) return false;
    if (!mini_slots_form(t, rows) || vsize < 1 || rows->n_rows >= ((int64_t)1 << 20)) return false;
    int vbits = 1, gbits = 0;
    while ((1 << vbits) < vsize) ++vbits;
    const int64_t n_groups = (rows->n_rows + 63) >> 6;
    while (((int64_t)1 << gbits) < n_groups) ++gbits;
    return vbits + 6 + gbits <= MERGE_CSHIFT;
}

// the record workspace: [bases A | bases B | meta A | meta B] of `cap` records each
struct MiniRecLayout { size_t cap, total; };
MiniRecLayout mini_rec_layout(size_t cap, size_t)
{
    return MiniRecLayout{cap, 24 * cap};
}

// the largest record capacity (a multiple of 256) whose layout fits a workspace of that many bytes
size_t mini_rec_cap(int64_t rec_ws_bytes, size_t nb)
{
    size_t cap = 0;
    if (rec_ws_bytes > 0) {
        cap = (size_t)rec_ws_bytes / 24 / 256 * 256;
        while (cap >= 256 && mini_rec_layout(cap, nb).total > (size_t)rec_ws_bytes) cap -= 256;
    }
    return cap;
}

int check_mini_rows(const pg_rows *rows, const char *who)
{
    if (!rows) return PG_OK;
    if (rows->n_rows < 0 || rows->n_rows > PG_MINI_MAX_ROWS) return pg_fail(PG_EINVAL, "%s: %lld rows (at most %d per launch)", who, (long long)rows->n_rows, PG_MINI_MAX_ROWS);
    if (rows->n_rows > 0 && (!rows->row_start || !rows->row_end)) return pg_fail(PG_EINVAL, "%s: null row arrays", who);
    return PG_OK;
}

}  // namespace

// (compiled into the translation unit whose kernels the flags change: a library linked from a checked / stamped / variant
// mini.o says so whatever its file name is)
extern "C" uint32_t pg_build_flags(void)
{
    uint32_t f = 0;
#ifdef PG_CHECKED
    f |= PG_BUILD_CHECKED;
#endif
#ifdef PG_MINI_STAMPS
    f |= PG_BUILD_STAMPS;
#endif
#ifdef PG_VARIANT
    f |= PG_BUILD_VARIANT;
#endif
    return f;
}

extern "C" int64_t pg_mini_plan_bytes(int64_t n_words, const pg_table *t)
{
    if (n_words < 0) return pg_fail(PG_EINVAL, "negative word count");
    int rc = check_mini(t, "pg_mini_plan_bytes");
    if (rc) return rc;
    MiniPlan p;
    plan_mini(t, n_words, &p);
    return (int64_t)p.total;
}

extern "C" int64_t pg_mini_records_bytes(int64_t n_records, const pg_table *t)
{
    if (n_records < 0) return pg_fail(PG_EINVAL, "negative record count");
    int rc = check_mini(t, "pg_mini_records_bytes");
    if (rc) return rc;
    const size_t n = ((size_t)n_records + 255) / 256 * 256 + 256;
    return (int64_t)mini_rec_layout(n, (size_t)1 << (t->log2_slots - t->log2_bucket_slots)).total;
}

extern "C" int64_t pg_mini_shuffle_bytes(int64_t n_words, int64_t n_rows, int vsize)
{
    if (n_words < 0 || n_rows < 0) return pg_fail(PG_EINVAL, "negative size");
    // (enough for either form of the row shuffle: the count kernel scattering into up to 2^11 row groups itself, or the general form)
    pg_shuffle_layout sl, sl1;
    int rc = pg_internal_shuffle_layout(n_words * 32, n_rows, vsize, &sl);
    if (rc) return rc;
    if ((rc = pg_internal_shuffle_layout(n_words * 32, n_rows, vsize, &sl1, MINI_ONE_PASS_BITS))) return rc;
    return (int64_t)(sl.total > sl1.total ? sl.total : sl1.total);
}

// the same for a caller that brings the merged lookups' slot buffer (pg_mini_merge_words): where that form applies and the row
// groups take one scatter pass, the provisional-word buffer of the row shuffle is not part of the layout (half the bytes)
extern "C" int64_t pg_mini_shuffle_bytes_merged(int64_t n_words, int64_t n_rows, int vsize, const pg_table *t)
{
    int rc = check_mini(t, "pg_mini_shuffle_bytes_merged");
    if (rc) return rc;
    if (n_words < 0 || n_rows < 0) return pg_fail(PG_EINVAL, "negative size");
    pg_rows r{nullptr, nullptr, n_rows, nullptr};
    if (!mini_merge_form(t, &r, vsize)) return pg_mini_shuffle_bytes(n_words, n_rows, vsize);
    pg_shuffle_layout sl;
    if ((rc = pg_internal_shuffle_layout(n_words * 32, n_rows, vsize, &sl, MINI_ONE_PASS_BITS, 1))) return rc;
    return (int64_t)sl.total;
}

// the first-pass kernels are instantiated per window length (registers of the rolling minimum); k > 21 runs the delayed
// central window of 8 or 9 M-mers (mini_window)
#define PG_MINI_DISPATCH_W(K_, CALL)                                                                                        \
    {                                                                                                                       \
        int wc_, woff;                                                                                                      \
        mini_window(K_, &wc_, &woff);                                                                                       \
        if ((K_) > PG_HASH_MAX_K) {                                                                                         \
            constexpr bool DELAY = true;                                                                                    \
            constexpr int M = MINI_M;                                                                                       \
            if (wc_ == 8) { constexpr int W = 8; CALL; } else { constexpr int W = 9; CALL; }                                \
        } else if (mini_m(K_) == MINI_M_SMALL) {                                                                            \
            constexpr bool DELAY = false;                                                                                   \
            constexpr int M = MINI_M_SMALL;                                                                                 \
            switch (wc_) {                                                                                                  \
            case 3: { constexpr int W = 3; CALL; } break;                                                                   \
            case 4: { constexpr int W = 4; CALL; } break;                                                                   \
            default: { constexpr int W = 5; CALL; } break;                                                                  \
            }                                                                                                               \
        } else {                                                                                                            \
            constexpr bool DELAY = false;                                                                                   \
            constexpr int M = MINI_M;                                                                                       \
            switch (wc_) {                                                                                                  \
            case 4: { constexpr int W = 4; CALL; } break;                                                                   \
            case 5: { constexpr int W = 5; CALL; } break;                                                                   \
            case 6: { constexpr int W = 6; CALL; } break;                                                                   \
            case 7: { constexpr int W = 7; CALL; } break;                                                                   \
            case 8: { constexpr int W = 8; CALL; } break;                                                                   \
            default: { constexpr int W = 9; CALL; } break;                                                                  \
            }                                                                                                               \
        }                                                                                                                   \
    }

extern "C" int pg_mini_plan(const uint64_t *codes, const uint32_t *valid, int64_t word_begin, int64_t word_end, const pg_table *t,
                            const pg_rows *rows, void *plan_ws, int64_t plan_ws_bytes, void *stream)
{
    if (!codes || !valid || !plan_ws) return pg_fail(PG_EINVAL, "pg_mini_plan: null argument");
    if (word_begin < 0 || word_end < word_begin) return pg_fail(PG_EINVAL, "pg_mini_plan: bad word range");
    int rc = check_mini(t, "pg_mini_plan");
    if (rc) return rc;
    if ((rc = check_mini_rows(rows, "pg_mini_plan"))) return rc;
    MiniPlan p;
    plan_mini(t, word_end - word_begin, &p);
    if ((int64_t)p.total > plan_ws_bytes) return pg_fail(PG_EINVAL, "pg_mini_plan: workspace of %lld bytes, %lld needed", (long long)plan_ws_bytes, (long long)p.total);
    if ((reinterpret_cast<uintptr_t>(plan_ws) & 255) != 0) return pg_fail(PG_EINVAL, "pg_mini_plan: workspace must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    char *ws = (char *)plan_ws;
    auto *header = (unsigned long long *)(ws + p.header_off);
    auto *region_tot = (unsigned long long *)(ws + p.rtot_off);
    auto *region_off = (unsigned long long *)(ws + p.roff_off);
    auto *round_row = (int32_t *)(ws + p.round_off);
    auto *chunk_tab = (unsigned long long *)(ws + p.chunk_off);
    const int n_regions = 1 << p.bits1;
    if (hipMemsetAsync(ws, 0, p.round_off, s) != hipSuccess) return pg_fail(PG_EHIP, "pg_mini_plan: memset failed");
    if (hipMemsetAsync(chunk_tab, 0, ((size_t)p.n_chunks << p.bits1) * 8, s) != hipSuccess) return pg_fail(PG_EHIP, "pg_mini_plan: memset failed");
    const bool with_rows = rows && rows->n_rows > 0;
    if (word_end > word_begin) {
        if (with_rows)
            hipLaunchKernelGGL(round_rows_kernel, dim3((unsigned)((p.n_rounds + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s, rows->row_end, rows->n_rows,
                               word_begin, p.n_rounds, round_row);
        const int grid = (int)(p.n_chunks < 4096 ? p.n_chunks : 4096);
        PG_MINI_DISPATCH_W(t->k,
            hipLaunchKernelGGL((mini_plan_kernel<W, DELAY, M>), dim3(grid), dim3(BLOCK), 0, s, codes, valid, word_begin, word_end, t->k, woff, p.bits, p.bits2, mini_cap(t->k),
                               with_rows ? rows->row_start : (const int64_t *)nullptr, with_rows ? rows->row_end : (const int64_t *)nullptr,
                               with_rows ? rows->n_rows : (int64_t)0, with_rows ? rows->strict_valid : (const uint32_t *)nullptr,
                               (const int32_t *)round_row, chunk_tab, p.n_chunks, p.chunk_stride, header + 2))
    }
    // records per region -> where the regions start -> exact offset of every (chunk, region) run; total -> header[0]
    hipLaunchKernelGGL(digit_totals_kernel, dim3((unsigned)n_regions), dim3(BIG_BLOCK), 0, s, (const unsigned long long *)chunk_tab, p.n_chunks, region_tot);
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(BIG_BLOCK), 0, s, (const unsigned long long *)region_tot, (int64_t)n_regions, region_off);
    hipLaunchKernelGGL(digit_scan_kernel, dim3((unsigned)n_regions), dim3(BIG_BLOCK), 0, s, chunk_tab, p.n_chunks, (const unsigned long long *)region_off,
                       0, (unsigned long long *)nullptr);
    hipLaunchKernelGGL(mini_total_kernel, dim3(1), dim3(64), 0, s, (const unsigned long long *)region_off, n_regions, header);
    return check_launch("pg_mini_plan");
}

static thread_local hipEvent_t first_pass_event = nullptr;         // recorded by every pg_mini_count behind its first scatter pass

// makes `stream` wait for the first scatter pass of this thread's latest pg_mini_count: work enqueued on it afterwards (the next
// batch's plan) overlaps the memory-bound second pass instead of the VALU-bound first one
extern "C" int pg_mini_wait_first_pass(void *stream)
{
    if (!first_pass_event) return pg_fail(PG_EINVAL, "pg_mini_wait_first_pass: no count yet");
    if (hipStreamWaitEvent((hipStream_t)stream, first_pass_event, 0) != hipSuccess) return pg_fail(PG_EHIP, "pg_mini_wait_first_pass: wait failed");
    return PG_OK;
}

// `half` (N > 1 ranks): the count half only -- no slice, no lookups; the bucket workgroups leave entries, occupancy and what
// the lookup half needs (HalfArgs).  `t` is then the rank's LOCAL geometry (the union's buckets, slots for the rank's own
// k-mers); its slots are never written.
static int mini_count_impl(const uint64_t *codes, const uint32_t *valid, int64_t word_begin, int64_t word_end, const pg_table *t,
                           const pg_rows *rows, void *plan_ws, int64_t plan_ws_bytes, void *rec_ws, int64_t rec_ws_bytes,
                           int window, int vsize, void *shuffle_ws, int64_t shuffle_ws_bytes, uint32_t *status, void *stream, const HalfArgs *half,
                           void *merge_ws = nullptr, int64_t merge_ws_words = 0)
{
    if (!codes || !valid || !plan_ws || !rec_ws || !status) return pg_fail(PG_EINVAL, "pg_mini_count: null argument");
    if (word_begin < 0 || word_end < word_begin) return pg_fail(PG_EINVAL, "pg_mini_count: bad word range");
    int rc = check_mini(t, "pg_mini_count");
    if (rc) return rc;
    if ((rc = check_mini_rows(rows, "pg_mini_count"))) return rc;
    const bool with_rows = rows && rows->n_rows > 0;
    if (window < 0 || vsize < 0 || (window > 0) != (vsize > 0)) return pg_fail(PG_EINVAL, "pg_mini_count: window %d / vector size %d", window, vsize);
    const bool piece = half && half->accum;                      // (pg_mini_count_piece: no lookups in this launch, no row shuffle)
    if (window > 0) {
        if (!with_rows) return pg_fail(PG_EINVAL, "pg_mini_count: the lookup pass needs rows");
        if (!shuffle_ws && !piece) return pg_fail(PG_EINVAL, "pg_mini_count: null shuffle workspace");
        if (vsize > PG_SHUFFLE_MAX_VSIZE || (t->kind == PG_TABLE_MINI && (int64_t)window * vsize > (int64_t)PG_HASH_COUNT_SAT))
            return pg_fail(PG_EINVAL, "pg_mini_count: window %d x vector size %d outside the exact range of the table", window, vsize);
    }
    MiniPlan p;
    plan_mini(t, word_end - word_begin, &p);
    if ((int64_t)p.total > plan_ws_bytes) return pg_fail(PG_EINVAL, "pg_mini_count: plan workspace of %lld bytes, %lld needed", (long long)plan_ws_bytes, (long long)p.total);
    if ((reinterpret_cast<uintptr_t>(plan_ws) & 255) != 0 || (reinterpret_cast<uintptr_t>(rec_ws) & 255) != 0)
        return pg_fail(PG_EINVAL, "pg_mini_count: workspaces must be 256-byte aligned");
    // the largest record capacity (a multiple of 256) whose layout fits the workspace
    const size_t cap = mini_rec_cap(rec_ws_bytes, (size_t)1 << p.bits);
    if (cap < 256) return pg_fail(PG_EINVAL, "pg_mini_count: record workspace of %lld bytes (pg_mini_records_bytes)", (long long)rec_ws_bytes);
    hipStream_t s = (hipStream_t)stream;
    char *ws = (char *)plan_ws;
    auto *header = (unsigned long long *)(ws + p.header_off);
    auto *off = (unsigned long long *)(ws + p.off_off);
    auto *cur2 = (unsigned long long *)(ws + p.cur2_off);
    auto *cur2l = (unsigned long long *)(ws + p.cur2l_off);
    auto *kwords = (unsigned long long *)(ws + p.kw_off);            // (hist | cur2 | cur2l | kwords are cleared together below)
    auto *hist = (unsigned long long *)(ws + p.hist_off);
    auto *region_off = (unsigned long long *)(ws + p.roff_off);
    auto *wbeg = (unsigned long long *)(ws + p.wbeg_off);
    auto *round_row = (int32_t *)(ws + p.round_off);
    auto *chunk_tab = (unsigned long long *)(ws + p.chunk_off);
    const int nb = 1 << p.bits;
    // record buffers: [bases A | bases B | meta A | meta B], cap records each
    auto *bases_a = (uint64_t *)rec_ws;
    auto *bases_b = bases_a + cap;
    auto *meta_a = (uint32_t *)(bases_b + cap);
    auto *meta_b = meta_a + cap;
    if (hipMemsetAsync(hist, 0, p.wbeg_off - p.hist_off, s) != hipSuccess || hipMemsetAsync(header + 1, 0, 8, s) != hipSuccess)
        return pg_fail(PG_EHIP, "pg_mini_count: memset failed");
    pg_shuffle_layout sl{0, 0, 0, 0, 0};
    if (window > 0 && !piece) {
        if ((rc = pg_internal_shuffle_layout((word_end - word_begin) * 32, rows->n_rows, vsize, &sl, mini_slots_form(t, rows) ? MINI_ONE_PASS_BITS : PG_SHUFFLE_ONE_PASS_BITS,
                                             merge_ws && merge_ws_words > 0 && mini_merge_form(t, rows, vsize) ? 1 : 0)))
            return rc;
        if ((int64_t)sl.total > shuffle_ws_bytes || (reinterpret_cast<uintptr_t>(shuffle_ws) & 255) != 0)
            return pg_fail(PG_EINVAL, "pg_mini_count: shuffle workspace of %lld bytes (256-byte aligned), %lld needed", (long long)shuffle_ws_bytes, (long long)sl.total);
    }
    const bool wide = t->kind == PG_TABLE_MINI_WIDE;              // 8-byte keys + 4-byte counts per slot
    const size_t table_lds = (size_t)(wide ? 12 : 8) << t->log2_bucket_slots;
    const size_t slice_lds = table_lds + (size_t)COUNT_WAVES * RING * 12;      // table + the wavefronts' rings (1024 threads)
    if (word_end > word_begin) {
#define PG_MINI_LAUNCH_SCATTER(N1_)                                                                                          \
        PG_MINI_DISPATCH_W(t->k,                                                                                            \
            const size_t lds1 = sizeof(Scatter1Lds<N1_>);                                                                   \
            if ((rc = raise_lds_limit((const void *)(mini_scatter_kernel<W, DELAY, M, N1_>), lds1, "pg_mini_count"))) return rc; \
            hipLaunchKernelGGL((mini_scatter_kernel<W, DELAY, M, N1_>), dim3((unsigned)p.n_chunks), dim3(S1_BLOCK), lds1, s, codes, valid, word_begin, word_end, t->k, woff, p.bits, \
                               p.bits2, mini_cap(t->k), with_rows ? rows->row_start : (const int64_t *)nullptr, with_rows ? rows->row_end : (const int64_t *)nullptr, \
                               with_rows ? rows->n_rows : (int64_t)0, with_rows ? rows->strict_valid : (const uint32_t *)nullptr, \
                               (const int32_t *)round_row, bases_a, meta_a, (const unsigned long long *)chunk_tab, p.n_chunks, p.chunk_stride, \
                               (const unsigned long long *)header, (const unsigned long long *)region_off, (unsigned long long)cap, status))
        PG_MINI_LAUNCH_SCATTER(256)
#undef PG_MINI_LAUNCH_SCATTER
    }
    // (for pg_mini_wait_first_pass: what is enqueued on another stream behind this event runs beside the second pass and the count)
    if (!first_pass_event && hipEventCreateWithFlags(&first_pass_event, hipEventDisableTiming) != hipSuccess)
        return pg_fail(PG_EHIP, "pg_mini_count: event creation failed");
    if (hipEventRecord(first_pass_event, s) != hipSuccess) return pg_fail(PG_EHIP, "pg_mini_count: event record failed");
    if (p.bits2) {
        // records per bucket from the regions' meta plane -> where every bucket starts (an empty range: all zero)
        const int tiles_h = 32;
        if (word_end > word_begin)
            hipLaunchKernelGGL(mini_bucket_hist_kernel, dim3((unsigned)(tiles_h << p.bits1)), dim3(BLOCK), 0, s, (const uint32_t *)meta_a,
                               (const unsigned long long *)region_off, p.bits2, tiles_h, hist, (const unsigned long long *)header, (unsigned long long)cap, (const uint32_t *)status);
        hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(BIG_BLOCK), 0, s, (const unsigned long long *)hist, (int64_t)nb, off);
    } else {
        off = region_off;                                   // buckets = regions
    }
    // (the merged lookups size their buffer from record counts: the second pass need not tally the k-mers in rows per bucket)
    const bool merge_a2 = window > 0 && merge_ws && merge_ws_words > 0 && mini_merge_form(t, rows, vsize);
    if (word_end > word_begin) {
        if (p.bits2) {
            // about one workgroup per tile of a region (from the record capacity: the count itself is on the device): with the regions'
            // workgroups on one XCD each, 96 workgroups per region walking a dozen tiles each took 3.32 ms where 576 to 1152 take 3.05
            const int tiles_x = (int)std::min<size_t>(2048, std::max<size_t>(8, cap / ((size_t)S2_TILE << p.bits1) + 1));
            if (p.bits2 > 7)
                hipLaunchKernelGGL(mini_scatter2_kernel<512>, dim3((unsigned)(tiles_x << p.bits1)), dim3(BLOCK), 0, s, (const uint64_t *)bases_a, (const uint32_t *)meta_a,
                                   (const unsigned long long *)off, p.bits2, tiles_x, mini_cap(t->k) > SHORT_MAX ? SHORT_MAX : 0, bases_b, meta_b, cur2, cur2l,
                                   merge_a2 ? (unsigned long long *)nullptr : kwords,
                                   (const unsigned long long *)header, (unsigned long long)cap, status);
            else
                hipLaunchKernelGGL(mini_scatter2_kernel<256>, dim3((unsigned)(tiles_x << p.bits1)), dim3(BLOCK), 0, s, (const uint64_t *)bases_a, (const uint32_t *)meta_a,
                                   (const unsigned long long *)off, p.bits2, tiles_x, mini_cap(t->k) > SHORT_MAX ? SHORT_MAX : 0, bases_b, meta_b, cur2, cur2l,
                                   merge_a2 ? (unsigned long long *)nullptr : kwords,
                                   (const unsigned long long *)header, (unsigned long long)cap, status);
        }
    }
    const bool slots_form = window > 0 && mini_slots_form(t, rows);
    // (the classes are sorted apart by the second scatter pass: without it -- at most 256 buckets -- every record counts as long)
    const unsigned long long *n_short = p.bits2 && mini_cap(t->k) > SHORT_MAX ? (const unsigned long long *)cur2 : (const unsigned long long *)nullptr;
    uint32_t *words_e = window && !piece ? (uint32_t *)((char *)shuffle_ws + sl.words_e_off) : (uint32_t *)nullptr;
    uint32_t *words_a = window && !piece ? (uint32_t *)((char *)shuffle_ws + sl.words_a_off) : (uint32_t *)nullptr;
    ShufArgs sh{nullptr, nullptr, nullptr, 0, 0, 0, 0, 0ull};
    const MergeArgs mg{(uint32_t *)merge_ws, (unsigned long long)(merge_ws ? merge_ws_words : 0)};
    const bool merge = window > 0 && merge_ws && merge_ws_words > 0 && mini_merge_form(t, rows, vsize);
    if (merge_ws && (reinterpret_cast<uintptr_t>(merge_ws) & 255) != 0) return pg_fail(PG_EINVAL, "pg_mini_count: workspaces must be 256-byte aligned");
    if (half && (wide || !(window > 0 && mini_slots_form(t, rows))))
        return pg_fail(PG_EINVAL, "pg_mini_count_half: needs packed slots (k <= %d), rows and fewer than 2^(32 - log2 bucket slots) of them", PG_HASH_MAX_K);
    if (half && !p.bits2) return pg_fail(PG_EINVAL, "pg_mini_count_half: needs more than 256 buckets");
    if (piece && !merge) return pg_fail(PG_EINVAL, "pg_mini_count_piece: needs the merged lookups (fewer than 2^20 rows, their slot buffer given)");
    const HalfArgs hv = half ? *half : HalfArgs{nullptr, nullptr, nullptr, nullptr, 0};
    size_t count_lds = slice_lds;
    // Buckets of at most 2^13 8-byte slots (64 KiB): 512-thread workgroups, TWO per CU -- one can be in its count loop (VALU, waits)
    // while the other is in its lookup phase (LDS throughput).  PG_COUNT_BLOCK=1024: the one-workgroup form for such tables too.
    const char *blk_env = getenv("PG_COUNT_BLOCK");
    const bool many_groups = window > 0 && rows && rows->n_rows > ((int64_t)1 << (PG_SHUFFLE_ONE_PASS_BITS + 6));     // more than 2^10 row groups: 2048 digits
    // (the count half does no lookups: the row-group digits do not matter to its geometry)
    const bool half_block = !wide && window > 0 && mini_slots_form(t, rows) && t->log2_bucket_slots <= 13 && (!many_groups || half) && !(blk_env && atoi(blk_env) == 1024);
    if (half_block) count_lds = table_lds + (size_t)(512 / 64) * RING * 12;
    if (slots_form && !piece) {
        // the count kernel scatters its words into the row shuffle's group regions itself: offsets and cursors must be ready
        pg_shuffle_ctx ctx;
        if ((rc = pg_internal_shuffle_prepare((word_end - word_begin) * 32, rows, vsize, shuffle_ws, shuffle_ws_bytes, stream, &ctx, MINI_ONE_PASS_BITS, merge ? 1 : 0))) return rc;
        sh = ShufArgs{ctx.goff, ctx.gcur1, ctx.words_out, ctx.gb1, ctx.gb2, ctx.dshift, merge ? 0 : ctx.narrow, ctx.words_cap};
        words_a = ctx.words_in;                          // the provisional words wait in the shuffle's input buffer
        const size_t lookup_lds = merge ? (half_block ? MergeLds<512, 1024>::END : sh.gb1 > 10 ? MergeLds<BIG_BLOCK, 2048>::END : MergeLds<BIG_BLOCK, 1024>::END)
                                        : half_block ? LookupLds<512, 1024>::END : sh.gb1 > 10 ? LookupLds<BIG_BLOCK, 2048>::END : LookupLds<BIG_BLOCK, 1024>::END;
        if (sh.gb1 > MINI_ONE_PASS_BITS) return pg_fail(PG_EINVAL, "pg_mini_count: %d first-pass digits of the row shuffle", sh.gb1);
        if (count_lds < lookup_lds && !half) count_lds = lookup_lds;
    }
    unsigned long long *emit_end = window && !piece ? (unsigned long long *)((char *)shuffle_ws + sl.emit_off) : (unsigned long long *)nullptr;
#define PG_MINI_LAUNCH_COUNT__(CAP_, SLOTS_, WIDE_, BLK_, DIG_, MERGE_, HALF_, LDS_)                                         \
    do {                                                                                                                    \
        if ((rc = raise_lds_limit((const void *)(mini_count_kernel<CAP_, SLOTS_, WIDE_, BLK_, DIG_, MERGE_, HALF_>), LDS_, "pg_mini_count"))) return rc; \
        hipLaunchKernelGGL((mini_count_kernel<CAP_, SLOTS_, WIDE_, BLK_, DIG_, MERGE_, HALF_>), dim3(nb), dim3(BLK_), LDS_, s, \
                           (const uint64_t *)(p.bits2 ? bases_b : bases_a), (const uint32_t *)(p.bits2 ? meta_b : meta_a),  \
                           (const unsigned long long *)off, n_short, p.bits2 ? (const unsigned long long *)kwords : (const unsigned long long *)nullptr, \
                           mini_view(t), (uint32_t)window, (uint32_t)vsize,                                                 \
                           sl.vbits, words_e, words_a, header + 1, wbeg, emit_end, sh, mg, (unsigned long long)cap, status, hv); \
    } while (0)
#define PG_MINI_LAUNCH_COUNT_(CAP_, SLOTS_, WIDE_, BLK_, DIG_, MERGE_, LDS_) PG_MINI_LAUNCH_COUNT__(CAP_, SLOTS_, WIDE_, BLK_, DIG_, MERGE_, false, LDS_)
#define PG_MINI_LAUNCH_SLOTS_(CAP_, WIDE_, BLK_, DIG_)                                                                      \
    do {                                                                                                                    \
        if (merge) PG_MINI_LAUNCH_COUNT_(CAP_, true, WIDE_, BLK_, DIG_, true, count_lds);                                    \
        else PG_MINI_LAUNCH_COUNT_(CAP_, true, WIDE_, BLK_, DIG_, false, count_lds);                                         \
    } while (0)
#define PG_MINI_LAUNCH_SLOTS(CAP_, WIDE_)                                                                                   \
    do {                                                                                                                    \
        if (sh.gb1 > 10) PG_MINI_LAUNCH_SLOTS_(CAP_, WIDE_, BIG_BLOCK, 2048);                                                \
        else PG_MINI_LAUNCH_SLOTS_(CAP_, WIDE_, BIG_BLOCK, 1024);                                                            \
    } while (0)
#define PG_MINI_LAUNCH_COUNT(CAP_)                                                                                          \
    do {                                                                                                                    \
        if (half) { if (half_block) { if (merge) PG_MINI_LAUNCH_COUNT__(CAP_, true, false, 512, 1024, true, true, count_lds);  \
                                      else PG_MINI_LAUNCH_COUNT__(CAP_, true, false, 512, 1024, false, true, count_lds); }   \
                    else { if (merge) PG_MINI_LAUNCH_COUNT__(CAP_, true, false, BIG_BLOCK, 1024, true, true, count_lds);     \
                           else PG_MINI_LAUNCH_COUNT__(CAP_, true, false, BIG_BLOCK, 1024, false, true, count_lds); } }      \
        else if (wide) { if (slots_form) PG_MINI_LAUNCH_SLOTS(CAP_, true); else PG_MINI_LAUNCH_COUNT_(CAP_, false, true, BIG_BLOCK, 1024, false, slice_lds); } \
        else if (half_block) PG_MINI_LAUNCH_SLOTS_(CAP_, false, 512, 1024);                                                  \
        else { if (slots_form) PG_MINI_LAUNCH_SLOTS(CAP_, false); else PG_MINI_LAUNCH_COUNT_(CAP_, false, false, BIG_BLOCK, 1024, false, slice_lds); } \
    } while (0)
    switch (mini_cap(t->k)) {                                          // k-mers per record at most (as the first pass cuts them)
    case 1: case 2: case 3: case 4: PG_MINI_LAUNCH_COUNT(4); break;
    case 5: case 6: PG_MINI_LAUNCH_COUNT(6); break;
    case 7: case 8: PG_MINI_LAUNCH_COUNT(8); break;
    default: PG_MINI_LAUNCH_COUNT(9); break;        // (cap <= W <= 9: see mini_cap)
    }
#undef PG_MINI_LAUNCH_COUNT
#undef PG_MINI_LAUNCH_SLOTS
#undef PG_MINI_LAUNCH_SLOTS_
#undef PG_MINI_LAUNCH_COUNT_
#undef PG_MINI_LAUNCH_COUNT__
    return check_launch("pg_mini_count");
}

extern "C" int pg_mini_count(const uint64_t *codes, const uint32_t *valid, int64_t word_begin, int64_t word_end, const pg_table *t,
                             const pg_rows *rows, void *plan_ws, int64_t plan_ws_bytes, void *rec_ws, int64_t rec_ws_bytes,
                             int window, int vsize, void *shuffle_ws, int64_t shuffle_ws_bytes, void *merge_ws, int64_t merge_ws_words,
                             uint32_t *status, void *stream)
{
    return mini_count_impl(codes, valid, word_begin, word_end, t, rows, plan_ws, plan_ws_bytes, rec_ws, rec_ws_bytes, window, vsize, shuffle_ws,
                           shuffle_ws_bytes, status, stream, nullptr, merge_ws, merge_ws_words);
}

// ---- a stream counted in pieces (one GPU; see HalfArgs / mini_lookup_slice_merge_kernel)
extern "C" int pg_mini_count_piece(const uint64_t *codes, const uint32_t *valid, int64_t word_begin, int64_t word_end, const pg_table *t,
                                   const pg_rows *rows, void *plan_ws, int64_t plan_ws_bytes, void *rec_ws, int64_t rec_ws_bytes,
                                   int window, int vsize, void *merge_ws, int64_t merge_ws_words, int first, uint32_t *status, void *stream)
{
    int rc = check_mini(t, "pg_mini_count_piece");
    if (rc) return rc;
    if (t->kind != PG_TABLE_MINI) return pg_fail(PG_EINVAL, "pg_mini_count_piece: packed mini tables (13 <= k <= %d)", PG_HASH_MAX_K);
    if (window < 1 || vsize < 1 || !merge_ws) return pg_fail(PG_EINVAL, "pg_mini_count_piece: needs the abundance parameters and the slot buffer");
    const HalfArgs hv{nullptr, nullptr, nullptr, nullptr, first ? 1 : 2};
    return mini_count_impl(codes, valid, word_begin, word_end, t, rows, plan_ws, plan_ws_bytes, rec_ws, rec_ws_bytes, window, vsize, nullptr, 0,
                           status, stream, &hv, merge_ws, merge_ws_words);
}

// offsets and cursors of the row shuffle for the words of ALL pieces (n_words_total: the words of the whole stream); once, before the
// first pg_mini_lookup_piece
extern "C" int pg_mini_lookup_begin(const pg_table *t, const pg_rows *rows, int64_t n_words_total, int vsize, void *shuffle_ws, int64_t shuffle_ws_bytes, void *stream)
{
    int rc = check_mini(t, "pg_mini_lookup_begin");
    if (rc) return rc;
    if (!rows || !shuffle_ws || (rc = check_mini_rows(rows, "pg_mini_lookup_begin"))) return rc ? rc : pg_fail(PG_EINVAL, "pg_mini_lookup_begin: null argument");
    if (!mini_merge_form(t, rows, vsize)) return pg_fail(PG_EINVAL, "pg_mini_lookup_begin: the merged lookups do not apply to these rows");
    pg_shuffle_ctx ctx;
    return pg_internal_shuffle_prepare(n_words_total * 32, rows, vsize, shuffle_ws, shuffle_ws_bytes, stream, &ctx, MINI_ONE_PASS_BITS, 1);
}

// the lookups of one piece: plan_ws = the piece's plan workspace as pg_mini_count_piece left it (n_words_piece words), meta = the
// kept meta words of its bucket-ordered records (the second meta plane of its record workspace, n_records of them), merge_ws its
// slot buffer; the words go to the row shuffle's regions behind those of the pieces before.  pg_mini_abundance_from_emitted (with
// n_words_total) then writes the rows.
extern "C" int pg_mini_lookup_piece(const pg_table *t, const pg_rows *rows, const void *plan_ws, int64_t plan_ws_bytes, int64_t n_words_piece,
                                    const uint32_t *meta, int64_t n_words_total, int window, int vsize, void *shuffle_ws, int64_t shuffle_ws_bytes,
                                    const void *merge_ws, uint32_t *status, void *stream)
{
    int rc = check_mini(t, "pg_mini_lookup_piece");
    if (rc) return rc;
    if (t->kind != PG_TABLE_MINI || !rows || !plan_ws || !meta || !shuffle_ws || !merge_ws || !status || window < 1 || vsize < 1)
        return pg_fail(PG_EINVAL, "pg_mini_lookup_piece: bad argument");
    if ((rc = check_mini_rows(rows, "pg_mini_lookup_piece"))) return rc;
    if (!mini_merge_form(t, rows, vsize)) return pg_fail(PG_EINVAL, "pg_mini_lookup_piece: the merged lookups do not apply to these rows");
    MiniPlan p;
    plan_mini(t, n_words_piece, &p);
    if ((int64_t)p.total > plan_ws_bytes) return pg_fail(PG_EINVAL, "pg_mini_lookup_piece: plan workspace does not match n_words_piece");
    if (!p.bits2) return pg_fail(PG_EINVAL, "pg_mini_lookup_piece: needs more than 256 buckets");
    pg_shuffle_ctx ctx;
    if ((rc = pg_internal_shuffle_prepare(n_words_total * 32, rows, vsize, shuffle_ws, shuffle_ws_bytes, stream, &ctx, MINI_ONE_PASS_BITS, 1, 1))) return rc;
    if (ctx.gb1 > MINI_ONE_PASS_BITS) return pg_fail(PG_EINVAL, "pg_mini_lookup_piece: %d first-pass digits of the row shuffle", ctx.gb1);
    const ShufArgs sh{ctx.goff, ctx.gcur1, ctx.words_out, ctx.gb1, ctx.gb2, ctx.dshift, 0, ctx.words_cap};
    const char *ws = (const char *)plan_ws;
    const auto *off = (const unsigned long long *)(ws + p.off_off);
    const auto *cur2 = (const unsigned long long *)(ws + p.cur2_off);
    const auto *wbeg = (const unsigned long long *)(ws + p.wbeg_off);
    const int cap_k = mini_cap(t->k);
    const unsigned long long *n_short = cap_k > SHORT_MAX ? cur2 : (const unsigned long long *)nullptr;
    const unsigned nb = 1u << p.bits;
    hipStream_t s = (hipStream_t)stream;
#define PG_LOOKUP_SLICE_M(CAP_, DIG_)                                                                                       \
    do {                                                                                                                    \
        const size_t lds_ = MergeLds<BIG_BLOCK, DIG_>::END;                                                                 \
        if ((rc = raise_lds_limit((const void *)(mini_lookup_slice_merge_kernel<CAP_, BIG_BLOCK, DIG_>), lds_, "pg_mini_lookup_piece"))) return rc; \
        hipLaunchKernelGGL((mini_lookup_slice_merge_kernel<CAP_, BIG_BLOCK, DIG_>), dim3(nb), dim3(BIG_BLOCK), lds_, s, off, n_short, wbeg, mini_view(t), \
                           (uint32_t)window, (uint32_t)vsize, ctx.vbits, (const uint32_t *)merge_ws, meta, sh, status);     \
    } while (0)
#define PG_LOOKUP_SLICE_MC(CAP_) do { if (ctx.gb1 > 10) PG_LOOKUP_SLICE_M(CAP_, 2048); else PG_LOOKUP_SLICE_M(CAP_, 1024); } while (0)
    switch (cap_k) {
    case 1: case 2: case 3: case 4: PG_LOOKUP_SLICE_MC(4); break;
    case 5: case 6: PG_LOOKUP_SLICE_MC(6); break;
    case 7: case 8: PG_LOOKUP_SLICE_MC(8); break;
    default: PG_LOOKUP_SLICE_MC(9); break;
    }
#undef PG_LOOKUP_SLICE_MC
#undef PG_LOOKUP_SLICE_M
    return check_launch("pg_mini_lookup_piece");
}

// dwords of the merged form's provisional buffer (pg_mini_count's merge_ws): a halfword per k-mer slot of every batch of 64
// records of either class, a step's slack.  n_records / n_long_records: the first and third 8-byte word of the plan workspace.
extern "C" int64_t pg_mini_merge_words(int64_t n_words, int64_t n_records, int64_t n_long_records, const pg_table *t)
{
    int rc = check_mini(t, "pg_mini_merge_words");
    if (rc) return rc;
    if (n_words < 0 || n_records < 0 || n_long_records < 0 || n_long_records > n_records) return pg_fail(PG_EINVAL, "pg_mini_merge_words: bad counts");
    const int64_t nb = (int64_t)1 << (t->log2_slots - t->log2_bucket_slots);
    const int cap = mini_cap(t->k);
    const int cap_t = cap <= 4 ? 4 : cap <= 6 ? 6 : cap <= 8 ? 8 : 9;           // (the kernels' instantiations)
    MiniPlan p;
    plan_mini(t, n_words, &p);
    const bool two = cap_t > SHORT_MAX && p.bits2 > 0;               // (without a second scatter pass the classes are not sorted apart: all "long")
    const int64_t n_long = two ? n_long_records : n_records, n_short = two ? n_records - n_long_records : 0;
    // (dwords: a batch of 64 records has a row of 64 halfwords = 32 dwords per k-mer of its class's cap; every bucket rounds both
    // classes up to whole batches; a step's slack behind the last bucket)
    const int64_t fixed = 32 * (int64_t)(two ? SHORT_MAX : cap_t) * (n_short / 64 + nb) + 32 * (int64_t)cap_t * (n_long / 64 + nb);
    return (fixed + 32 * 16 + 255) / 256 * 256;
}

// ---- N > 1 ranks (see the kernels above): workspace of the count half: entry slabs | occupancy bitmaps | ring counts
namespace {
struct MiniHalfLayout { size_t ent_off, occ_off, ring_off, total; };
MiniHalfLayout mini_half_layout(const pg_table *t)
{
    const size_t nb = (size_t)1 << (t->log2_slots - t->log2_bucket_slots), n_slots = (size_t)1 << t->log2_bucket_slots;
    const size_t n_occ = n_slots >= 64 ? n_slots / 64 : 1;
    MiniHalfLayout l;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += (bytes + 255) / 256 * 256; return at; };
    l.ent_off = take(nb * n_slots * 8);
    l.occ_off = take(nb * n_occ * 8);
    l.ring_off = take(nb * 4);
    l.total = o;
    return l;
}
}  // namespace

extern "C" int64_t pg_mini_half_bytes(const pg_table *local)
{
    int rc = check_mini(local, "pg_mini_half_bytes");
    if (rc) return rc;
    return (int64_t)mini_half_layout(local).total;
}

extern "C" int pg_mini_count_half(const uint64_t *codes, const uint32_t *valid, int64_t word_begin, int64_t word_end, const pg_table *local,
                                  const pg_rows *rows, void *plan_ws, int64_t plan_ws_bytes, void *rec_ws, int64_t rec_ws_bytes,
                                  int window, int vsize, void *shuffle_ws, int64_t shuffle_ws_bytes, void *merge_ws, int64_t merge_ws_words,
                                  void *half_ws, int64_t half_ws_bytes, int64_t *fill, uint32_t *status, void *stream)
{
    int rc = check_mini(local, "pg_mini_count_half");
    if (rc) return rc;
    if (!half_ws || !fill) return pg_fail(PG_EINVAL, "pg_mini_count_half: null argument");
    const MiniHalfLayout hl = mini_half_layout(local);
    if ((int64_t)hl.total > half_ws_bytes || (reinterpret_cast<uintptr_t>(half_ws) & 255) != 0)
        return pg_fail(PG_EINVAL, "pg_mini_count_half: workspace of %lld bytes (256-byte aligned), %lld needed", (long long)half_ws_bytes, (long long)hl.total);
    char *hw = (char *)half_ws;
    const HalfArgs hv{(unsigned long long *)(hw + hl.ent_off), (unsigned long long *)(hw + hl.occ_off), (long long *)fill, (uint32_t *)(hw + hl.ring_off), 0};
    return mini_count_impl(codes, valid, word_begin, word_end, local, rows, plan_ws, plan_ws_bytes, rec_ws, rec_ws_bytes, window, vsize, shuffle_ws,
                           shuffle_ws_bytes, status, stream, &hv, merge_ws, merge_ws_words);
}

extern "C" int pg_mini_gather_entries(const pg_table *local, const void *half_ws, int64_t half_ws_bytes, const int64_t *fill,
                                      const int64_t *dst_elem, uint64_t *out, int64_t out_elems, uint32_t *status, void *stream)
{
    int rc = check_mini(local, "pg_mini_gather_entries");
    if (rc) return rc;
    if (!half_ws || !fill || !dst_elem || !out || !status || out_elems < 0) return pg_fail(PG_EINVAL, "pg_mini_gather_entries: null argument");
    const MiniHalfLayout hl = mini_half_layout(local);
    if ((int64_t)hl.total > half_ws_bytes) return pg_fail(PG_EINVAL, "pg_mini_gather_entries: workspace does not match the table");
    const unsigned nb = 1u << (local->log2_slots - local->log2_bucket_slots);
    hipLaunchKernelGGL(mini_gather_entries_kernel, dim3(nb), dim3(BLOCK), 0, (hipStream_t)stream,
                       (const unsigned long long *)((const char *)half_ws + hl.ent_off), local->log2_bucket_slots, (const long long *)fill,
                       (const long long *)dst_elem, (unsigned long long *)out, (unsigned long long)out_elems, status);
    return check_launch("pg_mini_gather_entries");
}

extern "C" int pg_mini_merge_bins(const uint64_t *recv, int64_t part_stride, const int64_t *seg, int n_parts, const pg_table *t,
                                  int64_t bucket_begin, int64_t bucket_end, int window, int vsize, uint16_t *bins_out, uint32_t *status, void *stream)
{
    int rc = check_mini(t, "pg_mini_merge_bins");
    if (rc) return rc;
    if (t->kind != PG_TABLE_MINI) return pg_fail(PG_EINVAL, "pg_mini_merge_bins: packed mini tables only");
    const int64_t nb = (int64_t)1 << (t->log2_slots - t->log2_bucket_slots);
    if (!recv || !seg || !bins_out || !status || n_parts < 1 || bucket_begin < 0 || bucket_end < bucket_begin || bucket_end > nb || window < 1 || vsize < 1)
        return pg_fail(PG_EINVAL, "pg_mini_merge_bins: bad arguments");
    if (bucket_end == bucket_begin) return PG_OK;
    const size_t lds = (size_t)8 << t->log2_bucket_slots;
    if ((rc = raise_lds_limit((const void *)mini_merge_bins_kernel, lds, "pg_mini_merge_bins"))) return rc;
    hipLaunchKernelGGL(mini_merge_bins_kernel, dim3((unsigned)(bucket_end - bucket_begin)), dim3(BIG_BLOCK), lds, (hipStream_t)stream,
                       (const unsigned long long *)recv, (long long)part_stride, (const long long *)seg, n_parts, (int)(bucket_end - bucket_begin),
                       mini_view(t), (long long)bucket_begin, (uint32_t)window, (uint32_t)vsize, bins_out, status);
    return check_launch("pg_mini_merge_bins");
}

extern "C" int pg_mini_lookup_half(const pg_table *local, const pg_rows *rows, const void *plan_ws, int64_t plan_ws_bytes, const void *rec_ws, int64_t rec_ws_bytes,
                                   int64_t n_words_counted,
                                   int vsize, void *shuffle_ws, int64_t shuffle_ws_bytes, const void *merge_ws, int64_t merge_ws_words,
                                   const void *half_ws, int64_t half_ws_bytes,
                                   const uint16_t *bins_in, const int64_t *bin_elem, uint32_t *status, void *stream)
{
    int rc = check_mini(local, "pg_mini_lookup_half");
    if (rc) return rc;
    if (!rows || !plan_ws || !shuffle_ws || !half_ws || !bins_in || !bin_elem || !status) return pg_fail(PG_EINVAL, "pg_mini_lookup_half: null argument");
    if ((rc = check_mini_rows(rows, "pg_mini_lookup_half"))) return rc;
    if (local->kind != PG_TABLE_MINI || !mini_slots_form(local, rows)) return pg_fail(PG_EINVAL, "pg_mini_lookup_half: not the slot form");
    MiniPlan p;
    plan_mini(local, n_words_counted, &p);
    if ((int64_t)p.total > plan_ws_bytes) return pg_fail(PG_EINVAL, "pg_mini_lookup_half: plan workspace does not match n_words_counted");
    const MiniHalfLayout hl = mini_half_layout(local);
    if ((int64_t)hl.total > half_ws_bytes) return pg_fail(PG_EINVAL, "pg_mini_lookup_half: workspace does not match the table");
    const char *ws = (const char *)plan_ws;
    const auto *kwords = (const unsigned long long *)(ws + p.kw_off);
    const auto *wbeg = (const unsigned long long *)(ws + p.wbeg_off);
    if (!p.bits2) return pg_fail(PG_EINVAL, "pg_mini_lookup_half: needs more than 256 buckets");
    // the row shuffle's regions: offsets and cursors as the count half's launch prepared them (same layout call)
    pg_shuffle_ctx ctx;
    const bool merge = merge_ws && merge_ws_words > 0 && mini_merge_form(local, rows, vsize);         // (as the count half decided)
    if ((rc = pg_internal_shuffle_prepare(n_words_counted * 32, rows, vsize, shuffle_ws, shuffle_ws_bytes, stream, &ctx, MINI_ONE_PASS_BITS, merge ? 1 : 0))) return rc;
    if (ctx.gb1 > MINI_ONE_PASS_BITS) return pg_fail(PG_EINVAL, "pg_mini_lookup_half: %d first-pass digits of the row shuffle", ctx.gb1);
    const ShufArgs sh{ctx.goff, ctx.gcur1, ctx.words_out, ctx.gb1, ctx.gb2, ctx.dshift, merge ? 0 : ctx.narrow, ctx.words_cap};
    const unsigned nb = 1u << p.bits;
    const auto *occ = (const unsigned long long *)((const char *)half_ws + hl.occ_off);
    if (merge) {
        // (the merged form reads the records' meta words again: lengths and rows say which provisional slots mean anything)
        if (!rec_ws) return pg_fail(PG_EINVAL, "pg_mini_lookup_half: the merged form needs the count half's record workspace");
        const size_t rcap = mini_rec_cap(rec_ws_bytes, (size_t)1 << p.bits);
        if (rcap < 256) return pg_fail(PG_EINVAL, "pg_mini_lookup_half: record workspace of %lld bytes (pg_mini_records_bytes)", (long long)rec_ws_bytes);
        const uint32_t *meta_b = (const uint32_t *)((const uint64_t *)rec_ws + 2 * rcap) + rcap;     // [bases A | bases B | meta A | meta B]
        const auto *off = (const unsigned long long *)(ws + p.off_off);
        const auto *cur2 = (const unsigned long long *)(ws + p.cur2_off);
        const auto *ring = (const uint32_t *)((const char *)half_ws + hl.ring_off);
        const int cap_k = mini_cap(local->k);
        const unsigned long long *n_short = cap_k > SHORT_MAX ? cur2 : (const unsigned long long *)nullptr;
        hipStream_t s2 = (hipStream_t)stream;
#define PG_LOOKUP_HALF_M(CAP_, DIG_)                                                                                        \
        do {                                                                                                                \
            const size_t lds_ = MergeLds<BIG_BLOCK, DIG_>::END;                                                             \
            if ((rc = raise_lds_limit((const void *)(mini_lookup_half_merge_kernel<CAP_, BIG_BLOCK, DIG_>), lds_, "pg_mini_lookup_half"))) return rc; \
            hipLaunchKernelGGL((mini_lookup_half_merge_kernel<CAP_, BIG_BLOCK, DIG_>), dim3(nb), dim3(BIG_BLOCK), lds_, s2, off, n_short, wbeg, ring, occ, \
                               bins_in, (const long long *)bin_elem, local->log2_bucket_slots, ctx.vbits, (const uint32_t *)merge_ws, meta_b, sh, status); \
        } while (0)
        // buckets of at most 2^13 slots (what a rank's own reads need at 4+ ranks): 512-thread workgroups, two per CU -- this kernel
        // starts cold (bins, meta words and provisional slots all come from HBM, where the one-GPU kernel's lookup phase finds
        // them in L2), and a second workgroup on the CU hides what one alone waits for: 7.8 -> 5.9 ms, rehearsed 8-rank step
        // 34.3 -> 32.3 ms on one box (PG_LOOKUP_HALF_1024=1: the one-workgroup form)
#define PG_LOOKUP_HALF_M5(CAP_)                                                                                             \
        do {                                                                                                                \
            const size_t lds_ = MergeLds<512, 1024>::END;                                                                   \
            if ((rc = raise_lds_limit((const void *)(mini_lookup_half_merge_kernel<CAP_, 512, 1024>), lds_, "pg_mini_lookup_half"))) return rc; \
            hipLaunchKernelGGL((mini_lookup_half_merge_kernel<CAP_, 512, 1024>), dim3(nb), dim3(512), lds_, s2, off, n_short, wbeg, ring, occ, \
                               bins_in, (const long long *)bin_elem, local->log2_bucket_slots, ctx.vbits, (const uint32_t *)merge_ws, meta_b, sh, status); \
        } while (0)
#define PG_LOOKUP_HALF_MC(CAP_) do { if (ctx.gb1 > 10) PG_LOOKUP_HALF_M(CAP_, 2048); else if (local->log2_bucket_slots <= 13 && !getenv("PG_LOOKUP_HALF_1024")) PG_LOOKUP_HALF_M5(CAP_); else PG_LOOKUP_HALF_M(CAP_, 1024); } while (0)
        switch (cap_k) {
        case 1: case 2: case 3: case 4: PG_LOOKUP_HALF_MC(4); break;
        case 5: case 6: PG_LOOKUP_HALF_MC(6); break;
        case 7: case 8: PG_LOOKUP_HALF_MC(8); break;
        default: PG_LOOKUP_HALF_MC(9); break;
        }
#undef PG_LOOKUP_HALF_MC
#undef PG_LOOKUP_HALF_M5
#undef PG_LOOKUP_HALF_M
        return check_launch("pg_mini_lookup_half");
    }
    // 1024-thread workgroups whatever the bucket size: tiles of 16 Ki words keep the runs per row group at 32 bytes -- with
    // 512 threads (8 Ki-word tiles, two workgroups per CU) this kernel took 14.1 ms where the one-GPU kernel's lookup phase
    // takes 8.5 (PG_LOOKUP_HALF_512=1: that form, for comparison)
    const bool half_block = local->log2_bucket_slots <= 13 && ctx.gb1 <= 10 && getenv("PG_LOOKUP_HALF_512");
    hipStream_t s = (hipStream_t)stream;
#define PG_LOOKUP_HALF(BLK_, DIG_)                                                                                          \
    do {                                                                                                                    \
        const size_t lds_ = LookupLds<BLK_, DIG_>::END;                                                                     \
        if ((rc = raise_lds_limit((const void *)(mini_lookup_half_kernel<BLK_, DIG_>), lds_, "pg_mini_lookup_half"))) return rc; \
        hipLaunchKernelGGL((mini_lookup_half_kernel<BLK_, DIG_>), dim3(nb), dim3(BLK_), lds_, s, kwords, wbeg, occ, bins_in, (const long long *)bin_elem, \
                           local->log2_bucket_slots, ctx.vbits, (const uint32_t *)ctx.words_in, sh, status);                \
    } while (0)
    if (half_block) PG_LOOKUP_HALF(512, 1024);
    else if (ctx.gb1 > 10) PG_LOOKUP_HALF(BIG_BLOCK, 2048);
    else PG_LOOKUP_HALF(BIG_BLOCK, 1024);
#undef PG_LOOKUP_HALF
    return check_launch("pg_mini_lookup_half");
}

extern "C" int pg_mini_abundance_from_emitted(const pg_table *t, const pg_rows *rows, int vsize, int32_t *abd_out,
                                              const void *plan_ws, int64_t plan_ws_bytes, int64_t n_words_counted,
                                              void *shuffle_ws, int64_t shuffle_ws_bytes, void *stream)
{
    int rc = check_mini(t, "pg_mini_abundance_from_emitted");
    if (rc) return rc;
    if (!rows || !abd_out || !plan_ws || !shuffle_ws) return pg_fail(PG_EINVAL, "pg_mini_abundance_from_emitted: null argument");
    if ((rc = check_mini_rows(rows, "pg_mini_abundance_from_emitted"))) return rc;
    MiniPlan p;
    plan_mini(t, n_words_counted, &p);
    if ((int64_t)p.total > plan_ws_bytes) return pg_fail(PG_EINVAL, "pg_mini_abundance_from_emitted: plan workspace does not match n_words_counted");
    if ((reinterpret_cast<uintptr_t>(shuffle_ws) & 255) != 0) return pg_fail(PG_EINVAL, "pg_mini_abundance_from_emitted: workspace must be 256-byte aligned");
    if (mini_slots_form(t, rows))        // the count kernel has scattered the words by row group already
        return pg_internal_shuffle_finish(n_words_counted * 32, rows, vsize, abd_out, shuffle_ws, shuffle_ws_bytes, stream,
                                          mini_merge_form(t, rows, vsize) ? PG_SHUFFLE_WORDS_COUNTED
                                          : pg_internal_shuffle_is_narrow(n_words_counted * 32, rows->n_rows, vsize, MINI_ONE_PASS_BITS) ? PG_SHUFFLE_WORDS_NARROW
                                          : PG_SHUFFLE_WORDS_PLAIN, MINI_ONE_PASS_BITS, mini_merge_form(t, rows, vsize) ? 1 : 0);
    const auto *wbeg = (const unsigned long long *)((const char *)plan_ws + p.wbeg_off);
    return pg_internal_shuffle_rows(wbeg, 1 << p.bits, n_words_counted * 32, rows, vsize, abd_out, shuffle_ws, shuffle_ws_bytes, stream);
}
