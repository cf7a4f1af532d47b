// row_chain_ubench.hip — latency of dependent chains mixing v_mad_u64_u32 with cross-lane operations (DPP row
// broadcast / shift, ds_swizzle, v_readlane), the building blocks of csrc/gecm_row.hpp's row step.
// cycles per chain step at 1 and 2 wavefronts per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define BODY8(S) S S S S S S S S
// fixed registers: v[40:41] = accumulator t, v42 = x, v43 = y, v44 = q, v[46:47] = (L, 0)
#define CLOB "v40", "v41", "v42", "v43", "v44", "v46", "v47", "vcc", "s20"
template <int KIND>
__global__ void __launch_bounds__(64, 2) k(uint32_t *out, uint32_t iters, uint32_t seed)
{
    uint32_t r;
    asm volatile("v_mov_b32 v40, %0\n\tv_mov_b32 v41, 0\n\tv_mov_b32 v42, %1\n\tv_mov_b32 v43, 0x0fffffff\n\tv_mov_b32 v44, 0\n\t"
                 "v_mov_b32 v46, 0\n\tv_mov_b32 v47, 0" : : "v"(seed + threadIdx.x), "v"(seed * 3 + threadIdx.x) : CLOB);
    for (uint32_t i = 0; i < iters; i++) {
        if (KIND == 0) {          // mad only
            BODY8(asm volatile("v_mad_u64_u32 v[40:41], vcc, v42, v43, v[40:41]" : : : CLOB);)
        } else if (KIND == 1) {   // mad -> dpp bcast of low word -> mad using it
            BODY8(asm volatile("v_mad_u64_u32 v[40:41], vcc, v42, v43, v[40:41]\n\ts_nop 1\n\tv_mov_b32_dpp v42, v40 row_newbcast:0 row_mask:0xf bank_mask:0xf" : : : CLOB);)
        } else if (KIND == 2) {   // mad -> plain v_mov of low word -> mad
            BODY8(asm volatile("v_mad_u64_u32 v[40:41], vcc, v42, v43, v[40:41]\n\tv_mov_b32 v42, v40" : : : CLOB);)
        } else if (KIND == 3) {   // mad -> ds_swizzle bcast -> mad
            BODY8(asm volatile("v_mad_u64_u32 v[40:41], vcc, v42, v43, v[40:41]\n\tds_swizzle_b32 v42, v40 offset:swizzle(BROADCAST,16,0)\n\ts_waitcnt lgkmcnt(0)" : : : CLOB);)
        } else if (KIND == 4) {   // mad -> dpp row_shl:1 -> mad
            BODY8(asm volatile("v_mad_u64_u32 v[40:41], vcc, v42, v43, v[40:41]\n\ts_nop 1\n\tv_mov_b32_dpp v42, v40 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : : : CLOB);)
        } else if (KIND == 5) {   // mad -> v_readfirstlane -> mad with sgpr
            BODY8(asm volatile("v_mad_u64_u32 v[40:41], vcc, v42, v43, v[40:41]\n\tv_readfirstlane_b32 s20, v40\n\ts_nop 3\n\tv_mad_u64_u32 v[40:41], vcc, s20, v43, v[40:41]" : : : CLOB);)
        } else if (KIND == 6) {   // the row step of gecm_row.hpp: mad, dppQ, mad, dppL, mad16
            BODY8(asm volatile("v_mad_i64_i32 v[40:41], vcc, v42, v43, v[40:41]\n\ts_nop 1\n\tv_mov_b32_dpp v44, v40 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                               "v_mad_u64_u32 v[40:41], vcc, v44, v43, v[40:41]\n\ts_nop 1\n\tv_mov_b32_dpp v46, v40 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                               "v_mad_i64_i32 v[40:41], vcc, v41, 16, v[46:47]" : : : CLOB);)
        } else if (KIND == 7) {   // dpp -> dpp dependent
            BODY8(asm volatile("s_nop 1\n\tv_mov_b32_dpp v42, v42 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : : : CLOB);)
        } else if (KIND == 8) {   // v_add dependent (baseline)
            BODY8(asm volatile("v_add_u32 v42, v42, v43" : : : CLOB);)
        } else if (KIND == 9) {   // mad -> v_add_u32_dpp (consumer with dpp modifier) -> mad
            BODY8(asm volatile("v_mad_u64_u32 v[40:41], vcc, v42, v43, v[40:41]\n\ts_nop 1\n\tv_add_u32_dpp v42, v40, v43 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : : : CLOB);)
        } else if (KIND == 10) {  // mad -> permlane16_swap -> mad
            BODY8(asm volatile("v_mad_u64_u32 v[40:41], vcc, v42, v43, v[40:41]\n\tv_mov_b32 v42, v40\n\tv_permlane16_swap_b32 v42, v40" : : : CLOB);)
        } else if (KIND == 11) {  // row step without the wait states (is the hazard real? timing only)
            BODY8(asm volatile("v_mad_i64_i32 v[40:41], vcc, v42, v43, v[40:41]\n\tv_mov_b32_dpp v44, v40 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                               "v_mad_u64_u32 v[40:41], vcc, v44, v43, v[40:41]\n\tv_mov_b32_dpp v46, v40 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                               "v_mad_i64_i32 v[40:41], vcc, v41, 16, v[46:47]" : : : CLOB);)
        } else if (KIND == 14) {  // independent DPP moves (throughput)
            BODY8(asm volatile("v_mov_b32_dpp v44, v42 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp v46, v43 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : : : CLOB);)
        } else if (KIND == 15) {  // independent plain moves (throughput)
            BODY8(asm volatile("v_mov_b32 v44, v42\n\tv_mov_b32 v46, v43" : : : CLOB);)
        } else if (KIND == 16) {  // independent ds_swizzle (throughput)
            BODY8(asm volatile("ds_swizzle_b32 v44, v42 offset:swizzle(BROADCAST,16,3)\n\tds_swizzle_b32 v46, v43 offset:swizzle(SWAP,16)\n\ts_waitcnt lgkmcnt(0)" : : : CLOB);)
        } else if (KIND == 17) {  // mad + independent dpp (mix, no dependence)
            BODY8(asm volatile("v_mad_u64_u32 v[40:41], vcc, v42, v43, v[40:41]\n\tv_mov_b32_dpp v44, v43 row_newbcast:3 row_mask:0xf bank_mask:0xf" : : : CLOB);)
        } else if (KIND == 18) {  // mad + independent v_mov
            BODY8(asm volatile("v_mad_u64_u32 v[40:41], vcc, v42, v43, v[40:41]\n\tv_mov_b32 v44, v43" : : : CLOB);)
        } else if (KIND == 19) {  // VALU throughput of the row's mix: 3 mads + 3 dpp, independent
            BODY8(asm volatile("v_mad_u64_u32 v[40:41], vcc, v42, v43, v[40:41]\n\tv_mov_b32_dpp v44, v43 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                               "v_mad_u64_u32 v[48:49], vcc, v42, v43, v[48:49]\n\tv_mov_b32_dpp v46, v43 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                               "v_mad_u64_u32 v[50:51], vcc, v42, v43, v[50:51]\n\tv_mov_b32_dpp v47, v43 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : : : CLOB, "v48", "v49", "v50", "v51");)
        } else if (KIND == 20) {  // the same with one dpp replaced by a ds_swizzle (waited for one body later)
            BODY8(asm volatile("s_waitcnt lgkmcnt(0)\n\tv_mad_u64_u32 v[40:41], vcc, v42, v43, v[40:41]\n\tds_swizzle_b32 v44, v43 offset:swizzle(BROADCAST,16,3)\n\t"
                               "v_mad_u64_u32 v[48:49], vcc, v42, v43, v[48:49]\n\tv_mov_b32_dpp v46, v43 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                               "v_mad_u64_u32 v[50:51], vcc, v42, v43, v[50:51]\n\tv_mov_b32_dpp v47, v43 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : : : CLOB, "v48", "v49", "v50", "v51");)
        } else if (KIND == 21) {  // 3 mads + 2 dpp only
            BODY8(asm volatile("v_mad_u64_u32 v[40:41], vcc, v42, v43, v[40:41]\n\t"
                               "v_mad_u64_u32 v[48:49], vcc, v42, v43, v[48:49]\n\tv_mov_b32_dpp v46, v43 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                               "v_mad_u64_u32 v[50:51], vcc, v42, v43, v[50:51]\n\tv_mov_b32_dpp v47, v43 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : : : CLOB, "v48", "v49", "v50", "v51");)
        } else if (KIND == 12) {  // mul_lo chain (single pass multiplier?)
            BODY8(asm volatile("v_mul_lo_u32 v42, v42, v43" : : : CLOB);)
        } else if (KIND == 13) {  // mad_u32_u24 chain
            BODY8(asm volatile("v_mad_u32_u24 v42, v42, v43, v42" : : : CLOB);)
        }
    }
    asm volatile("v_add_u32 %0, v40, v42\n\tv_add_u32 %0, %0, v44" : "=v"(r) : : CLOB);
    out[blockIdx.x * 64 + threadIdx.x] = r;
}

template <int KIND>
static int run(const char *name, int steps_per_body)
{
    uint32_t *d;
    CK(hipMalloc(&d, 4096 * 64 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const uint32_t iters = 20000;
    for (unsigned blocks : {1024u, 2048u, 4096u}) {
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d, 10u, 1u);
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d, iters, 1u);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-44s waves/SIMD=%u  %7.2f cycles@2.4GHz per body step per wave\n", name, blocks / 1024, ms * 1e-3 * 2.4e9 / (iters * 8.0));
    }
    (void)steps_per_body;
    CK(hipFree(d));
    return 0;
}

int main()
{
    run<0>("mad", 1);
    run<8>("v_add_u32", 1);
    run<7>("dpp row_shl (nop 1 + dpp)", 1);
    run<2>("mad + v_mov", 2);
    run<1>("mad + nop1 + dpp newbcast", 2);
    run<4>("mad + nop1 + dpp row_shl", 2);
    run<9>("mad + nop1 + v_add_u32_dpp", 2);
    run<3>("mad + ds_swizzle + waitcnt", 2);
    run<5>("mad + readfirstlane + nop3 + mad(sgpr)", 2);
    run<10>("mad + mov + permlane16_swap", 3);
    run<6>("row step (3 mad, 2 dpp)", 5);
    run<11>("row step without s_nop (timing only)", 5);
    run<19>("3 mads + 3 dpp, independent", 6);
    run<20>("3 mads + 2 dpp + 1 ds_swizzle, independent", 6);
    run<21>("3 mads + 2 dpp, independent", 5);
    run<14>("2 independent dpp moves", 2);
    run<15>("2 independent plain moves", 2);
    run<16>("2 independent ds_swizzle + waitcnt", 2);
    run<17>("mad + independent dpp move", 2);
    run<18>("mad + independent plain move", 2);
    run<12>("v_mul_lo_u32", 1);
    run<13>("v_mad_u32_u24", 1);
    return 0;
}
