// row_mul_check.hip — stand-alone check and timing of the 16-lanes-per-residue multiply (csrc/gecm_row.hpp).
//   row_mul_check <in.bin> <out.bin> [iters]
// in.bin : u32 nq, u32 rho1 (1: modulus = -1 mod 2^28), u32 rho, u32 count, then 16*nq limbs of the modulus,
//          then count x (a limbs, b limbs) as int32, 16*nq each.
// out.bin: count x 16*nq int32 result limbs of a*b/R' ; then the chained-multiply timing is printed.
#include "../avx-ecm_amd/csrc/gecm_row.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int NQ, bool RHO1>
__global__ void __launch_bounds__(64, 2) k_check(const int32_t *in, int32_t *out, const uint32_t *mod, uint32_t rho, uint32_t count)
{
    const uint32_t g = blockIdx.x * 4u + (threadIdx.x >> 4), l = threadIdx.x & 15u;
    if (g >= count) return;          // count is a multiple of 4: whole rows leave together
    RowMod<NQ> m;
    FeR<NQ> a, b, r;
    for (int t = 0; t < NQ; t++) {
        m.n[t] = mod[NQ * l + t];
        a.v[t] = in[(size_t)g * 32 * NQ + NQ * l + t];
        b.v[t] = in[(size_t)g * 32 * NQ + 16 * NQ + NQ * l + t];
    }
    m.rho = rho;
    fer_mul<NQ, 16 * NQ, RHO1, false>(r, a, b, m);
    for (int t = 0; t < NQ; t++) out[(size_t)g * 16 * NQ + NQ * l + t] = r.v[t];
}

template <int NQ, bool RHO1>
__global__ void __launch_bounds__(64, 2) k_chain(const int32_t *in, int32_t *out, const uint32_t *mod, uint32_t rho, uint32_t iters)
{
    const uint32_t l = threadIdx.x & 15u;
    RowMod<NQ> m;
    FeR<NQ> a, b;
    for (int t = 0; t < NQ; t++) {
        m.n[t] = mod[NQ * l + t];
        a.v[t] = in[NQ * l + t];
        b.v[t] = in[16 * NQ + NQ * l + t];
    }
    m.rho = rho;
    for (uint32_t i = 0; i < iters; i++) {
        fer_mul<NQ, 16 * NQ, RHO1, false>(a, a, b, m);
        fer_mul<NQ, 16 * NQ, RHO1, false>(b, b, a, m);
    }
    for (int t = 0; t < NQ; t++) out[(size_t)(blockIdx.x * 64 + threadIdx.x) * NQ + t] = a.v[t] + b.v[t];
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NQ, bool RHO1>
static int run(const std::vector<uint32_t> &w, const char *outp, uint32_t iters)
{
    const uint32_t rho = w[2], count = w[3];
    const uint32_t *mod = &w[4];
    const int32_t *ab = (const int32_t *)&w[4 + 16 * NQ];
    uint32_t *dmod;
    int32_t *din, *dout;
    CK(hipMalloc(&dmod, 16 * NQ * 4));
    CK(hipMalloc(&din, (size_t)count * 32 * NQ * 4));
    CK(hipMalloc(&dout, (size_t)2048 * 64 * NQ * 4 + (size_t)count * 16 * NQ * 4));
    CK(hipMemcpy(dmod, mod, 16 * NQ * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(din, ab, (size_t)count * 32 * NQ * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL((k_check<NQ, RHO1>), dim3((count + 3) / 4), dim3(64), 0, 0, din, dout, dmod, rho, count);
    CK(hipDeviceSynchronize());
    std::vector<int32_t> out((size_t)count * 16 * NQ);
    CK(hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost));
    FILE *f = fopen(outp, "wb");
    fwrite(out.data(), 4, out.size(), f);
    fclose(f);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (unsigned blocks : {512u, 1024u, 2048u, 4096u}) {
        hipLaunchKernelGGL((k_chain<NQ, RHO1>), dim3(blocks), dim3(64), 0, 0, din, dout, dmod, rho, 10u);
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_chain<NQ, RHO1>), dim3(blocks), dim3(64), 0, 0, din, dout, dmod, rho, iters);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double muls_per_wave = 2.0 * iters;
        printf("nq=%d rho1=%d wavefronts=%u: %.3f ms, %.1f cycles@2.4GHz per wave-multiply, %.3f G residue-multiplies/s\n", NQ,
               (int)RHO1, blocks, ms, ms * 1e-3 * 2.4e9 / muls_per_wave, blocks * 4.0 * muls_per_wave / (ms * 1e-3) / 1e9);
    }
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<uint32_t> w(n / 4);
    if (fread(w.data(), 4, w.size(), f) != w.size()) return 2;
    fclose(f);
    const uint32_t iters = argc > 3 ? (uint32_t)atoi(argv[3]) : 20000u;
    const uint32_t nq = w[0], rho1 = w[1];
    if (nq == 1) return rho1 ? run<1, true>(w, argv[2], iters) : run<1, false>(w, argv[2], iters);
    if (nq == 2) return rho1 ? run<2, true>(w, argv[2], iters) : run<2, false>(w, argv[2], iters);
    if (nq == 3) return rho1 ? run<3, true>(w, argv[2], iters) : run<3, false>(w, argv[2], iters);
    return 2;
}
