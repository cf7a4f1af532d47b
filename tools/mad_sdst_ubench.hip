// v_mad_u64_u32 issues every 4.7 cycles with two or more wavefronts per SIMD where v_mul_lo_u32 or v_fma_f64 issue
// every 4.2 (profiles/r01_valu_ubench_gfx950.txt).  Is the carry-out (sdst) the reason?  Same chain of multiply-adds
// with the carry-out always in vcc, alternating between two SGPR pairs, rotating over four; with an SGPR multiplier;
// the signed form; and a v_mul_lo_u32 / v_mul_hi_u32 pair for scale.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while(0)

template <int KIND>
__global__ void __launch_bounds__(256) k(uint64_t *out, uint32_t seed, int iters)
{
    uint64_t acc[4];
    uint32_t a[8], b = seed | 1;
    const uint32_t sb = __builtin_amdgcn_readfirstlane(seed * 77u + 5u);
    for (int i = 0; i < 4; i++) acc[i] = seed + i + threadIdx.x;
    for (int i = 0; i < 8; i++) a[i] = seed * (i + 3) + threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            if (KIND == 0) {
                asm volatile("v_mad_u64_u32 %0, vcc, %4, %8, %0\n\tv_mad_u64_u32 %1, vcc, %5, %8, %1\n\t"
                             "v_mad_u64_u32 %2, vcc, %6, %8, %2\n\tv_mad_u64_u32 %3, vcc, %7, %8, %3"
                             : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])
                             : "v"(a[r & 7]), "v"(a[(r + 1) & 7]), "v"(a[(r + 2) & 7]), "v"(a[(r + 3) & 7]), "v"(b) : "vcc");
            } else if (KIND == 1) {
                asm volatile("v_mad_u64_u32 %0, vcc, %4, %8, %0\n\tv_mad_u64_u32 %1, s[20:21], %5, %8, %1\n\t"
                             "v_mad_u64_u32 %2, vcc, %6, %8, %2\n\tv_mad_u64_u32 %3, s[20:21], %7, %8, %3"
                             : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])
                             : "v"(a[r & 7]), "v"(a[(r + 1) & 7]), "v"(a[(r + 2) & 7]), "v"(a[(r + 3) & 7]), "v"(b) : "vcc", "s20", "s21");
            } else if (KIND == 2) {
                asm volatile("v_mad_u64_u32 %0, vcc, %4, %8, %0\n\tv_mad_u64_u32 %1, s[20:21], %5, %8, %1\n\t"
                             "v_mad_u64_u32 %2, s[22:23], %6, %8, %2\n\tv_mad_u64_u32 %3, s[24:25], %7, %8, %3"
                             : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])
                             : "v"(a[r & 7]), "v"(a[(r + 1) & 7]), "v"(a[(r + 2) & 7]), "v"(a[(r + 3) & 7]), "v"(b)
                             : "vcc", "s20", "s21", "s22", "s23", "s24", "s25");
            } else if (KIND == 3) {      // SGPR multiplier (the q*N products of the one-lane kernels)
                asm volatile("v_mad_u64_u32 %0, vcc, %4, %8, %0\n\tv_mad_u64_u32 %1, vcc, %5, %8, %1\n\t"
                             "v_mad_u64_u32 %2, vcc, %6, %8, %2\n\tv_mad_u64_u32 %3, vcc, %7, %8, %3"
                             : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])
                             : "v"(a[r & 7]), "v"(a[(r + 1) & 7]), "v"(a[(r + 2) & 7]), "v"(a[(r + 3) & 7]), "s"(sb) : "vcc");
            } else if (KIND == 4) {      // signed form
                asm volatile("v_mad_i64_i32 %0, vcc, %4, %8, %0\n\tv_mad_i64_i32 %1, vcc, %5, %8, %1\n\t"
                             "v_mad_i64_i32 %2, vcc, %6, %8, %2\n\tv_mad_i64_i32 %3, vcc, %7, %8, %3"
                             : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])
                             : "v"(a[r & 7]), "v"(a[(r + 1) & 7]), "v"(a[(r + 2) & 7]), "v"(a[(r + 3) & 7]), "v"(b) : "vcc");
            } else if (KIND == 5) {      // one accumulator, dependent chain, carry-out alternating
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %5, %0\n\tv_mad_u64_u32 %0, s[20:21], %2, %5, %0\n\t"
                             "v_mad_u64_u32 %0, vcc, %3, %5, %0\n\tv_mad_u64_u32 %0, s[20:21], %4, %5, %0"
                             : "+v"(acc[0])
                             : "v"(a[r & 7]), "v"(a[(r + 1) & 7]), "v"(a[(r + 2) & 7]), "v"(a[(r + 3) & 7]), "v"(b) : "vcc", "s20", "s21");
            } else {                     // four v_mul_lo_u32 for scale
                uint32_t t0 = (uint32_t)acc[0], t1 = (uint32_t)acc[1], t2 = (uint32_t)acc[2], t3 = (uint32_t)acc[3];
                asm volatile("v_mul_lo_u32 %0, %0, %4\n\tv_mul_lo_u32 %1, %1, %4\n\tv_mul_lo_u32 %2, %2, %4\n\tv_mul_lo_u32 %3, %3, %4"
                             : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3) : "v"(b));
                acc[0] = t0; acc[1] = t1; acc[2] = t2; acc[3] = t3;
            }
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < 4; i++) s += acc[i];
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
void run(const char *name, uint64_t *d, int ncu)
{
    const int iters = 20000;
    for (int wps = 1; wps <= 4; wps *= 2) {
        int blocks = ncu * wps;
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        k<KIND><<<blocks, 256>>>(d, 7, 10);
        CHECK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipEventRecord(e0));
            k<KIND><<<blocks, 256>>>(d, 7, iters);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        double n = (double)iters * 64;      // instructions per wave
        printf("%-46s wps=%d  %8.3f ms  %5.2f cycles@2.4GHz per instruction per SIMD\n", name, wps, best, best * 1e-3 * 2.4e9 / n / wps);
    }
}

int main()
{
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    int ncu = p.multiProcessorCount;
    uint64_t *d; CHECK(hipMalloc(&d, (size_t)ncu * 4 * 256 * 8));
    run<0>("mad_u64_u32, carry-out vcc", d, ncu);
    run<1>("mad_u64_u32, carry-out vcc / s[20:21]", d, ncu);
    run<2>("mad_u64_u32, carry-out over 4 SGPR pairs", d, ncu);
    run<3>("mad_u64_u32, SGPR multiplier, carry-out vcc", d, ncu);
    run<4>("mad_i64_i32, carry-out vcc", d, ncu);
    run<5>("mad_u64_u32 dependent, carry-out alternating", d, ncu);
    run<6>("v_mul_lo_u32", d, ncu);
    return 0;
}
