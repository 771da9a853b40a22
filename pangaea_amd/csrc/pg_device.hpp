// pg_device.hpp -- device helpers and small kernels shared by the translation units of libpangaea_feat.so
// (kernels.hip: the key-partitioned pipeline; mini.hip: the super-k-mer pipeline).  Everything lives in an anonymous
// namespace: each translation unit gets its own copy.
#ifndef PG_DEVICE_HPP
#define PG_DEVICE_HPP
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pangaea_feat.h"
#include "pg_internal.h"

namespace {


constexpr int BLOCK = 256;
constexpr int WAVES = BLOCK / 64;
constexpr int BIG_BLOCK = 1024;           // one workgroup per CU kernels (LDS-resident tables / histograms)
constexpr uint32_t HASH_CBITS = PG_HASH_COUNT_BITS;
constexpr uint64_t HASH_CMASK = (1ull << HASH_CBITS) - 1;
constexpr uint32_t HASH_SAT = PG_HASH_COUNT_SAT;
constexpr uint32_t MAX_PROBE = 1u << 14;

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

// word w of the stream, the word before it, and the mask of its positions that end a valid k-mer
struct Word {
    uint64_t cw, pw;
    uint32_t ok;
};
__device__ __forceinline__ Word load_word(const uint64_t *__restrict__ codes, const uint32_t *__restrict__ valid, int64_t w, int k)
{
    Word x;
    x.cw = codes[w];
    const uint32_t vw = valid[w];
    x.pw = w > 0 ? codes[w - 1] : 0;
    const uint32_t pv = w > 0 ? valid[w - 1] : 0;
    x.ok = (uint32_t)(runs_of(((uint64_t)vw << 32) | pv, k) >> 32);
    return x;
}

// Workgroup barrier for phases that communicate through LDS only.  __syncthreads() is a workgroup-scope fence + barrier and
// the fence waits for every outstanding GLOBAL load, store and atomic of the wavefront (s_waitcnt vmcnt(0)): in the LDS-staged
// scatter loops that exposes the full store latency of every copy-out at the next barrier (measured in mini_count's scatter
// phase: 8.6 of 26 k cycles per tile).  This one waits for the wavefront's LDS operations only; global stores, prefetch loads
// and cursor atomics stay in flight across it.  Not for data that another wavefront reads from GLOBAL memory.
// (builtins, not inline asm: the barrier intrinsic is convergent, so the compiler will not duplicate or sink it into divergent code;
// 0xc07f = lgkmcnt(0) with the vmcnt and expcnt fields at their maxima, i.e. not waited for; the signal fences keep the compiler
// from moving memory operations across)
__device__ __forceinline__ void lds_sync()
{
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
}

// A global store that names the capacity (in elements) of the buffer it writes into.  Ordinary builds: a plain store.  Checked
// builds (make checked -> libpangaea_feat_checked.so, PANGAEA_LIB=checked loads it; the GPU suite's super-k-mer cases run
// through it once): an index outside the buffer is NOT written and sets PG_STATUS_BOUNDS -- a kernel whose indexing is wrong
// fails a test instead of faulting the GPU (a memory fault of one process can reset every GPU of the host).
template <class T, class V> __device__ __forceinline__ void gstore(T *p, uint64_t idx, uint64_t cap, V v, uint32_t *status)
{
#ifdef PG_CHECKED
    if (idx >= cap) { atomicOr(status, PG_STATUS_BOUNDS); return; }
#endif
    p[idx] = (T)v;
}

// c / d for a divisor that is the same for every thread of a launch (rcp = 1.0f / (float)d): a float product and one correction
// each way -- about ten instructions where the generic 32-bit division takes forty; sixteen of those per thread and bucket made
// the bins of a bucket's slots cost more than writing its slice.  Exact: below 2^24 the product is off by less than 1 (for d >= 2
// its error is below 2 / d, for d = 1 there is none); beyond (32-bit counts of wide tables) the division itself.
__device__ __forceinline__ uint32_t div_uniform(uint32_t c, uint32_t d, float rcp)
{
    if (c >= (1u << 24)) return c / d;
    uint32_t q = (uint32_t)((float)c * rcp);
    int32_t r = (int32_t)(c - q * d);
    if (r < 0) { --q; r += (int32_t)d; }
    if ((uint32_t)r >= d) ++q;
    return q;
}

// per-digit exclusive scan of table[d][0..n) in place, plus base[d << base_shift]; totals[d] (may be NULL) = row sum.
// One workgroup per digit.
__global__ __launch_bounds__(BIG_BLOCK) void digit_scan_kernel(unsigned long long *__restrict__ table, int64_t n,
                                                               const unsigned long long *__restrict__ base, int base_shift,
                                                               unsigned long long *__restrict__ totals)
{
    __shared__ unsigned long long part[BIG_BLOCK];
    unsigned long long *row = table + (int64_t)blockIdx.x * n;
    const int64_t per = (n + BIG_BLOCK - 1) / BIG_BLOCK;
    const int64_t a = threadIdx.x * per, b = a + per < n ? a + per : n;
    unsigned long long s = 0;
    for (int64_t i = a; i < b; ++i) s += row[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long run = 0;
        for (int i = 0; i < BIG_BLOCK; ++i) { unsigned long long v = part[i]; part[i] = run; run += v; }
        if (totals) totals[blockIdx.x] = run;
    }
    __syncthreads();
    unsigned long long run = part[threadIdx.x] + base[(int64_t)blockIdx.x << base_shift];
    for (int64_t i = a; i < b; ++i) { const unsigned long long v = row[i]; row[i] = run; run += v; }
}

// exclusive prefix sum of hist[n] -> off[n+1] (one workgroup)
__global__ __launch_bounds__(BIG_BLOCK) void scan_kernel(const unsigned long long *__restrict__ hist, int64_t n, unsigned long long *__restrict__ off)
{
    __shared__ unsigned long long part[BIG_BLOCK];
    const int64_t per = (n + BIG_BLOCK - 1) / BIG_BLOCK;
    const int64_t a = threadIdx.x * per, b = a + per < n ? a + per : n;
    unsigned long long s = 0;
    for (int64_t i = a; i < b; ++i) s += hist[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long run = 0;
        for (int i = 0; i < BIG_BLOCK; ++i) { unsigned long long v = part[i]; part[i] = run; run += v; }
        off[n] = run;
    }
    __syncthreads();
    unsigned long long run = part[threadIdx.x];
    for (int64_t i = a; i < b; ++i) { off[i] = run; run += hist[i]; }
}


// -------------------------------------------------------------------------------- minimizer buckets (PG_TABLE_MINI)
//
// A MINI table places a canonical k-mer by the MINIMIZER of its M-mers (M = 13: odd, so no M-mer is its own reverse
// complement): bucket = a hash of the smallest mhash(canonical M-mer) over the k - M + 1 M-mers of the k-mer.  Consecutive
// k-mers of a read mostly share their minimizer, so the occurrences that go to one bucket travel as SUPER-k-mers: one
// 12-byte record (32 bases + row + length) for a run of up to 16 k-mers instead of 8 bytes per occurrence.  Inside the
// bucket the home slot is slot_hash(code); slots hold (canonical code << 22) | count, 0 = empty.
// M depends on k alone: 13 from k = 16 on, 11 for 13 <= k <= 15 (Pangaea's default k = 15 then has windows of five 11-mers;
// with 13-mers it would have three and hardly any k-mer would share its minimizer with a neighbour)
constexpr int MINI_M = PG_MINI_M, MINI_M_SMALL = PG_MINI_M_SMALL;
__host__ __device__ __forceinline__ int mini_m(int k) { return k >= MINI_M + 3 ? MINI_M : MINI_M_SMALL; }
static_assert(PG_MINI_MIN_K == MINI_M_SMALL + 2, "windows of at least three M-mers");

// The order in which M-mers compete for "minimizer": a multiplicative hash of the low 24 bits of the canonical M-mer (its
// oldest character does not take part: M-mers that differ only there tie, and a tie is harmless -- the bucket is a function of
// the minimum VALUE, which both strands compute from the same multiset).  The xor keeps poly-A (code 0) from being the
// smallest value of all.  Two full-rate instructions (v_xor, v_mul_u32_u24): this runs once per character of the stream in
// kernels that are bound by VALU issue; the three-xorshift mixer it replaces took seven.
__device__ __forceinline__ uint32_t mhash(uint32_t x)
{
    return __umul24(x ^ 0x5E3779u, 0xC2B2AFu);
}
// bucket of a minimizer value.  The value is a minimum -- its HIGH bits are biased towards 0 --, but its low 24 bits are a
// bijection of the minimizer M-mer's own low 24 bits (mhash: a product with an odd constant), as uniform as the M-mers are: one
// more 24-bit product mixes them upwards and the top bits of its low dword are the bucket.  (v_mul_u32_u24 takes the low 24 bits of
// its operands by itself and issues at full rate; the 32-bit product this replaces, v_mul_lo_u32, at a quarter of it -- and since
// round 4 the bucket of EVERY position of the stream is computed, not one per record.)
__device__ __forceinline__ uint32_t mini_bucket(uint32_t minv, int bits)
{
    return bits ? (uint32_t)__umul24(minv, 0x9E3779u) >> (32 - bits) : 0u;       // (HIP declares __umul24 as returning int)
}
__device__ __forceinline__ uint32_t swap_pairs32(uint32_t x) { return ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1); }
// order of the 32 two-bit characters of a word reversed
__device__ __forceinline__ uint64_t rev2_64(uint64_t x)
{
    const uint32_t lo = swap_pairs32(__brev((uint32_t)x)), hi = swap_pairs32(__brev((uint32_t)(x >> 32)));
    return ((uint64_t)lo << 32) | hi;
}
// (24-bit multiplies -- v_mul_u32_u24 issues at full rate, a 32-bit v_mul_lo_u32 at a quarter of it, and this runs once per
// k-mer occurrence in kernels that are bound by VALU issue; the fold brings the well-mixed high product bits, which depend
// on all bits of the code, down to the slot index).  WIDE: codes of k > 21 (up to 62 bits) add a third term, which is 0 for
// the 42-bit codes -- the two forms agree wherever both apply.
template <bool WIDE = false> __device__ __forceinline__ uint32_t mini_slot_hash(uint64_t code)
{
    const uint32_t a = (uint32_t)code & 0xffffffu, b = (uint32_t)(code >> 21);
    uint32_t x = __umul24(a, 0x9E3779u) ^ __umul24(b, 0xC2B2AFu);
    if (WIDE) x ^= __umul24((uint32_t)(code >> 45), 0x85EBCBu);
    x ^= x >> 15;
    return x;
}
// The M-mers of a k-mer that compete for its minimizer: all k - M + 1 of them up to 9, beyond that (k > 21) the CENTRAL 8 or 9
// (same parity as k - M + 1, so that `off` M-mers are left out on either side).  The reverse complement maps M-mer t to
// k - M - t, i.e. the central window onto itself: both strands pick the same M-mer, as they must.  A window of at most 9
// keeps the rolling minimum of the first scatter pass in 9 registers for every k.
constexpr int MINI_MAX_WINDOW = 9;
__host__ __device__ __forceinline__ void mini_window(int k, int *wc, int *off)
{
    const int w = k - mini_m(k) + 1;
    *wc = w <= MINI_MAX_WINDOW ? w : MINI_MAX_WINDOW - 1 + (w & 1);
    *off = (w - *wc) / 2;
}
// minimizer value of one k-mer given as a (forward or canonical) code, newest character in the low bits
__device__ __forceinline__ uint32_t mini_minimizer_of(uint64_t code, int k)
{
    int wc, off;
    mini_window(k, &wc, &off);
    const int m = mini_m(k);
    const uint32_t mmask = (1u << (2 * m)) - 1u;
    uint32_t best = 0xffffffffu;
    for (int t = off; t < off + wc; ++t) {
        const uint32_t fw = (uint32_t)(code >> (2 * t)) & mmask;
        const uint32_t rc = (swap_pairs32(__brev(fw)) >> (32 - 2 * m)) ^ (0xAAAAAAAAu & mmask);
        const uint32_t h = mhash(fw < rc ? fw : rc);
        best = h < best ? h : best;
    }
    return best;
}

int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return pg_fail(PG_EHIP, "%s: %s", what, hipGetErrorString(e));
    return PG_OK;
}

__attribute__((unused)) int grid_for(int64_t items, int block = BLOCK)
{
    int64_t blocks = (items + block - 1) / block;
    const int64_t cap = 256 * 16;     // 256 CUs x 16 resident workgroups' worth, grid-stride beyond
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

int raise_lds_limit(const void *kernel, size_t bytes, const char *who)
{
    if (bytes <= 64 * 1024) return PG_OK;
    // (the CU has 160 KiB; kernels here keep a few hundred bytes of static LDS besides)
    const int want = bytes <= 144 * 1024 ? 144 * 1024 : (int)bytes;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, want) != hipSuccess)
        return pg_fail(PG_EHIP, "%s: cannot raise the dynamic LDS limit", who);
    return PG_OK;
}

}  // namespace
#endif
