// Dependent-chain latency of v_mad_u64_u32 on gfx950: NCH independent accumulator chains per wave,
// each mad depends on the previous one of its chain (as in a product-scanning column).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while(0)

template <int NCH, int KIND>
__global__ void __launch_bounds__(256) k(uint64_t *out, uint32_t seed, int iters)
{
    uint64_t acc[NCH];
    uint32_t a[8], b = seed | 1;
    for (int i = 0; i < NCH; i++) acc[i] = seed + i + threadIdx.x;
    for (int i = 0; i < 8; i++) a[i] = seed * (i + 3) + threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 64 / NCH; r++) {
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                if (KIND == 0)
                    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[c]) : "v"(a[(r + c) & 7]), "v"(b) : "vcc");
                else if (KIND == 1)   // mad followed by dependent 64-bit shift (column end)
                    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_lshrrev_b64 %0, 1, %0" : "+v"(acc[c]) : "v"(a[(r + c) & 7]), "v"(b) : "vcc");
                else                  // v_mul_lo_u32 dependent on acc low, then mad using it (q computation pattern)
                {
                    uint32_t lo = (uint32_t)acc[c], q;
                    asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(q) : "v"(lo), "v"(b));
                    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[c]) : "v"(q), "v"(b) : "vcc");
                }
            }
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < NCH; i++) s += acc[i];
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NCH, int KIND>
void run(const char *name, uint64_t *d, int ncu)
{
    const int iters = 20000;
    for (int wps = 1; wps <= 4; wps++) {
        int blocks = ncu * wps;
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        k<NCH, KIND><<<blocks, 256>>>(d, 7, 10);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        k<NCH, KIND><<<blocks, 256>>>(d, 7, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        double n = (double)iters * (64 / NCH) * NCH;   // chain steps per wave
        printf("%-34s chains=%d wps=%d  %8.3f ms  %6.2f cyc@2.4GHz per step per wave  -> %6.2f per step per SIMD\n", name, NCH, wps, ms,
               ms * 1e-3 * 2.4e9 / n, ms * 1e-3 * 2.4e9 / n / wps);
    }
}

int main()
{
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    int ncu = p.multiProcessorCount;
    uint64_t *d; CHECK(hipMalloc(&d, (size_t)ncu * 4 * 256 * 8));
    run<1, 0>("mad dependent", d, ncu);
    run<2, 0>("mad dependent", d, ncu);
    run<4, 0>("mad dependent", d, ncu);
    run<8, 0>("mad dependent", d, ncu);
    run<1, 1>("mad+lshr64 dependent (2 instr)", d, ncu);
    run<2, 1>("mad+lshr64 dependent (2 instr)", d, ncu);
    run<1, 2>("mul_lo->mad dependent (2 instr)", d, ncu);
    run<2, 2>("mul_lo->mad dependent (2 instr)", d, ncu);
    return 0;
}
