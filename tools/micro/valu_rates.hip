#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int OP> __global__ void k(uint64_t *out, uint32_t s, int n)
{
    uint64_t a[8]; uint32_t b[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0x9E3779B97F4A7C15ull + i; b[i] = threadIdx.x * 2654435761u + i; }
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) b[i] = b[i] + s;                                  // v_add_u32
            if (OP == 1) a[i] = (a[i] << s) ^ (uint64_t)it;                // v_lshlrev_b64 (+ xor x2)
            if (OP == 2) b[i] = b[i] * 0xC2B2AE3Du + s;                    // v_mul_lo_u32
            if (OP == 3) b[i] = __umul24(b[i], 0xC2B2AFu) + s;             // v_mul_u32_u24
            if (OP == 4) b[i] += (a[i] < (uint64_t)b[(i + 1) & 7] * 77ull) ? 1u : 2u;   // v_cmp_lt_u64 (+ mul etc.)
            if (OP == 5) { b[i] = __builtin_amdgcn_alignbit(b[i], b[(i + 1) & 7], s); }   // v_alignbit
            if (OP == 6) a[i] = a[i] ^ (a[i] >> s);                        // v_lshrrev_b64 + 2 xor
            if (OP == 7) b[i] = b[i] ^ (b[i] >> s);                        // v_lshrrev_b32 + xor
        }
    }
    uint64_t r = 0;
    for (int i = 0; i < 8; ++i) r += a[i] + b[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int OP> void run(const char *name, uint64_t *d)
{
    const int n = 4096, blocks = 256 * 8, threads = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, threads>>>(d, 3, 16);
    hipEventRecord(e0);
    k<OP><<<blocks, threads>>>(d, 3, n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)blocks * threads / 64, ops = waves * n * 8;
    printf("%-28s %8.3f ms  %.2f ns per wave-op per SIMD (1024 SIMDs)\n", name, ms, ms * 1e6 / (ops / 1024));
}
int main()
{
    uint64_t *d; hipMalloc(&d, 256 * 8 * 256 * 8);
    run<0>("v_add_u32", d); run<7>("v_lshrrev_b32 + v_xor", d); run<5>("v_alignbit_b32", d); run<3>("v_mul_u32_u24 + add", d); run<2>("v_mul_lo_u32 + add", d);
    run<1>("v_lshlrev_b64 + 2 xor", d); run<6>("v_lshrrev_b64 + 2 xor", d); run<4>("v_cmp_lt_u64 + mul64 + ...", d);
    return 0;
}
