// VALU issue-rate microbenchmark for gfx950 (MI355X).
// Measures cycles per wave-instruction for the candidate instructions of the
// Montgomery multiply inner loop, at 1/2/4 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_ubench.hip -o tools/valu_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>
#include <algorithm>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while(0)

#define ITER 4000
// 16 independent chains, each op repeated in a block of 16 -> 16 instr per iteration of inner asm

#define R16a(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
#define R16(X) R16a(X) R16a(X) R16a(X) R16a(X) R16a(X) R16a(X) R16a(X) R16a(X) R16a(X) R16a(X) R16a(X) R16a(X) R16a(X) R16a(X) R16a(X) R16a(X)
#define REP 16
#define R8a(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define R8(X) R8a(X) R8a(X) R8a(X) R8a(X) R8a(X) R8a(X) R8a(X) R8a(X) R8a(X) R8a(X) R8a(X) R8a(X) R8a(X) R8a(X) R8a(X) R8a(X)

template <int OP>
__global__ void __launch_bounds__(512) k_bench(uint64_t *out, uint64_t seed, int iters)
{
    double d[16]; uint64_t q[16]; uint32_t u[16]; float f[16];
    uint32_t lane = threadIdx.x;
    for (int i = 0; i < 16; i++) {
        q[i] = seed * (i + 3) + lane * 0x9E3779B97F4A7C15ull;
        u[i] = (uint32_t)(q[i] >> 7) | 1;
        d[i] = (double)(q[i] & 0xFFFFFFFFFFFFFull);
        f[i] = (float)(u[i] & 0xFFFF);
    }
    double dc = (double)(seed & 0xFFFFF) + 3.0;
    uint32_t uc = (uint32_t)seed | 1;
    uint64_t qc = seed | 5;
    float fc = 1.0001f;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if constexpr (OP == 0) {        // v_fma_f64
#define X(i) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(dc));
            R16(X)
#undef X
        } else if constexpr (OP == 1) { // v_add_f64
#define X(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dc));
            R16(X)
#undef X
        } else if constexpr (OP == 2) { // v_mul_f64
#define X(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dc));
            R16(X)
#undef X
        } else if constexpr (OP == 3) { // v_mad_u64_u32
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(u[i]), "v"(uc) : "vcc");
            R16(X)
#undef X
        } else if constexpr (OP == 4) { // v_mul_lo_u32
#define X(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(uc));
            R16(X)
#undef X
        } else if constexpr (OP == 5) { // v_mul_hi_u32
#define X(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u[i]) : "v"(uc));
            R16(X)
#undef X
        } else if constexpr (OP == 6) { // v_mad_u32_u24
#define X(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(u[i]) : "v"(uc));
            R16(X)
#undef X
        } else if constexpr (OP == 7) { // v_mul_hi_u32_u24
#define X(i) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(u[i]) : "v"(uc));
            R16(X)
#undef X
        } else if constexpr (OP == 8) { // v_lshl_add_u64
#define X(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q[i]) : "v"(qc));
            R16(X)
#undef X
        } else if constexpr (OP == 9) { // add_co + addc pair (64-bit add as 2 instr)
#define X(i) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(u[i]), "+v"(u[(i+8)&15]) : "v"(uc), "v"(uc) : "vcc");
            R8(X)
#undef X
        } else if constexpr (OP == 10) { // v_add_u32
#define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(uc));
            R16(X)
#undef X
        } else if constexpr (OP == 11) { // v_add3_u32
#define X(i) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(u[i]) : "v"(uc));
            R16(X)
#undef X
        } else if constexpr (OP == 12) { // v_lshrrev_b64
#define X(i) asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(q[i]));
            R16(X)
#undef X
        } else if constexpr (OP == 13) { // v_alignbit_b32
#define X(i) asm volatile("v_alignbit_b32 %0, %0, %1, 3" : "+v"(u[i]) : "v"(uc));
            R16(X)
#undef X
        } else if constexpr (OP == 14) { // v_and_b32
#define X(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(uc));
            R16(X)
#undef X
        } else if constexpr (OP == 15) { // v_fma_f32
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[i]) : "v"(fc));
            R16(X)
#undef X
        } else if constexpr (OP == 16) { // v_pk_fma_f32
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(d[i]) : "v"(dc));
            R16(X)
#undef X
        } else if constexpr (OP == 17) { // v_dot2_u32_u16  (gfx950: may not exist -> use v_dot4_u32_u8)
#define X(i) asm volatile("v_dot4_u32_u8 %0, %0, %1, %0" : "+v"(u[i]) : "v"(uc));
            R16(X)
#undef X
        } else if constexpr (OP == 18) { // v_cvt_f64_u32
#define X(i) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d[i]) : "v"(u[i]));
            R16(X)
#undef X
        } else if constexpr (OP == 19) { // mix: fma_f64 interleaved with v_add_u32 (same wave)
#define X(i) asm volatile("v_fma_f64 %0, %0, %2, %0\n\tv_add_u32 %1, %1, %3" : "+v"(d[i]), "+v"(u[i]) : "v"(dc), "v"(uc));
            R16(X)
#undef X
        } else if constexpr (OP == 20) { // mix: fma_f64 interleaved with v_lshl_add_u64
#define X(i) asm volatile("v_fma_f64 %0, %0, %2, %0\n\tv_lshl_add_u64 %1, %1, 0, %3" : "+v"(d[i]), "+v"(q[i]) : "v"(dc), "v"(qc));
            R16(X)
#undef X
        } else if constexpr (OP == 21) { // mix: fma_f64 + mad_u64_u32
#define X(i) asm volatile("v_fma_f64 %0, %0, %2, %0\n\tv_mad_u64_u32 %1, vcc, %3, %4, %1" : "+v"(d[i]), "+v"(q[i]) : "v"(dc), "v"(u[i]), "v"(uc) : "vcc");
            R16(X)
#undef X
        } else if constexpr (OP == 22) { // v_mul_u32_u24
#define X(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(u[i]) : "v"(uc));
            R16(X)
#undef X
        } else if constexpr (OP == 23) { // v_fma_f64 with SGPR operand
            double sc = __builtin_bit_cast(double, __builtin_amdgcn_readfirstlane((int)(seed)) | 0x4330000000000000ull);
#define X(i) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "s"(sc));
            R16(X)
#undef X
        } else if constexpr (OP == 24) { // v_mad_u64_u32 with mixed wave roles: even waves fp64 fma, odd waves mad (co-issue test)
            if (threadIdx.x >= 256) {
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(u[i]), "v"(uc) : "vcc");
                R16(X)
#undef X
            } else {
#define X(i) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(dc));
                R16(X)
#undef X
            }
        } else if constexpr (OP == 25) { // co-issue test: even waves fp64 fma, odd waves v_add_u32
            if (threadIdx.x >= 256) {
#define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(uc));
                R16(X)
#undef X
            } else {
#define X(i) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(dc));
                R16(X)
#undef X
            }
        } else if constexpr (OP == 26) { // v_cvt_u32_f64
#define X(i) asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(u[i]) : "v"(d[i]));
            R16(X)
#undef X
        } else if constexpr (OP == 27) { // v_mov_b32
#define X(i) asm volatile("v_mov_b32 %0, %1" : "=v"(u[i]) : "v"(u[(i+1)&15]));
            R16(X)
#undef X
        } else if constexpr (OP == 28) { // v_mov_b64 (gfx940+)
#define X(i) asm volatile("v_mov_b64 %0, %1" : "=v"(q[i]) : "v"(q[(i+1)&15]));
            R16(X)
#undef X
        } else if constexpr (OP == 29) { // v_add_co_u32 with sgpr carry + v_addc chain (carry chain, dependent)
#define X(i) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(u[i]) : "v"(uc) : "vcc");
            R16(X)
#undef X
        } else if constexpr (OP == 30) { // v_mul_lo_u32 + v_mul_hi_u32 pair
#define X(i) asm volatile("v_mul_lo_u32 %0, %0, %2\n\tv_mul_hi_u32 %1, %1, %2" : "+v"(u[i]), "+v"(u[(i+8)&15]) : "v"(uc));
            R8(X)
#undef X
        } else if constexpr (OP == 31) { // v_pk_add_u16 as cheap filler reference / v_xor
#define X(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[i]) : "v"(uc));
            R16(X)
#undef X
        }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    uint64_t acc = 0;
    for (int i = 0; i < 16; i++) acc += q[i] + u[i] + (uint64_t)d[i] + (uint64_t)f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) out[gridDim.x * blockDim.x + blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

struct Op { int id; const char *name; int per_iter; };

template <int OP>
void run(const char *name, int per_iter, uint64_t *dout, int ncu)
{
    for (int wps = 1; wps <= 4; wps *= 2) {       // waves per SIMD
        int threads = 256;                          // 4 waves per block -> 1 per SIMD
        int blocks = ncu * wps;                     // wps blocks per CU
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        k_bench<OP><<<blocks, threads>>>(dout, 12345, 10);   // warm
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        k_bench<OP><<<blocks, threads>>>(dout, 12345, ITER);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        int nw = blocks * threads / 64;
        std::vector<uint64_t> cy(nw); CHECK(hipMemcpy(cy.data(), dout + (size_t)blocks * threads, 8 * nw, hipMemcpyDeviceToHost));
        std::sort(cy.begin(), cy.end());
        uint64_t cyc = cy[nw / 2]; uint64_t cmin = cy[0], cmax = cy[nw - 1];
        double ninstr = (double)ITER * per_iter * REP;            // per wave
        // s_memtime ticks at 100MHz-ish constant clock? report both
        double wall_cyc_per_instr_per_simd = (ms * 1e-3 * 2.4e9) / (ninstr * wps);
        printf("%-28s wps=%d  time=%8.3f ms  memtime/instr med=%6.3f min=%6.3f max=%6.3f  wallcyc@2.4G/instr/SIMD=%6.2f  Ginstr/s(chip)=%8.1f\n",
               name, wps, ms, (double)cyc / ninstr, (double)cmin / ninstr, (double)cmax / ninstr, wall_cyc_per_instr_per_simd,
               ninstr * blocks * 4 / (ms * 1e-3) / 1e9);
    }
}

template <int OP>
void run_co(const char *name, uint64_t *dout, int ncu)
{
    int threads = 512, blocks = ncu;              // waves 0-3: fp64 fma, waves 4-7: other op; 2 waves per SIMD
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k_bench<OP><<<blocks, threads>>>(dout, 12345, 10);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    k_bench<OP><<<blocks, threads>>>(dout, 12345, ITER);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double ninstr = (double)ITER * 16 * REP;
    printf("%-28s 512thr/block (2 waves/SIMD, split roles) time=%8.3f ms  wall_cycles@2.4GHz per (fma,other) pair per SIMD=%6.2f\n",
           name, ms, (ms * 1e-3 * 2.4e9) / ninstr);
}

int main()
{
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    int ncu = p.multiProcessorCount;
    printf("device %s  CUs=%d clock=%d kHz\n", p.name, ncu, p.clockRate);
    uint64_t *dout; CHECK(hipMalloc(&dout, (size_t)(ncu * 4 * 512 + ncu * 64) * 8));
    run<0>("v_fma_f64", 16, dout, ncu);
    run<23>("v_fma_f64(sgpr src)", 16, dout, ncu);
    run<1>("v_add_f64", 16, dout, ncu);
    run<2>("v_mul_f64", 16, dout, ncu);
    run<3>("v_mad_u64_u32", 16, dout, ncu);
    run<4>("v_mul_lo_u32", 16, dout, ncu);
    run<5>("v_mul_hi_u32", 16, dout, ncu);
    run<30>("mul_lo+mul_hi pair (2 instr)", 16, dout, ncu);
    run<6>("v_mad_u32_u24", 16, dout, ncu);
    run<22>("v_mul_u32_u24", 16, dout, ncu);
    run<7>("v_mul_hi_u32_u24", 16, dout, ncu);
    run<8>("v_lshl_add_u64", 16, dout, ncu);
    run<9>("add_co+addc pair (2 instr)", 16, dout, ncu);
    run<29>("v_addc_co_u32 chain", 16, dout, ncu);
    run<10>("v_add_u32", 16, dout, ncu);
    run<11>("v_add3_u32", 16, dout, ncu);
    run<12>("v_lshrrev_b64", 16, dout, ncu);
    run<13>("v_alignbit_b32", 16, dout, ncu);
    run<14>("v_and_b32", 16, dout, ncu);
    run<31>("v_xor_b32", 16, dout, ncu);
    run<27>("v_mov_b32", 16, dout, ncu);
    run<28>("v_mov_b64", 16, dout, ncu);
    run<15>("v_fma_f32", 16, dout, ncu);
    run<16>("v_pk_fma_f32", 16, dout, ncu);
    run<17>("v_dot4_u32_u8", 16, dout, ncu);
    run<18>("v_cvt_f64_u32", 16, dout, ncu);
    run<26>("v_cvt_u32_f64", 16, dout, ncu);
    run<19>("fma_f64+add_u32 (2 instr)", 32, dout, ncu);
    run<20>("fma_f64+lshl_add_u64 (2)", 32, dout, ncu);
    run<21>("fma_f64+mad_u64_u32 (2)", 32, dout, ncu);
    run_co<24>("coissue fma64|mad64 waves", dout, ncu);
    run_co<25>("coissue fma64|add_u32 waves", dout, ncu);
    return 0;
}
