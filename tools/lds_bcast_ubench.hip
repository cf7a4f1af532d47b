// lds_bcast_ubench.hip — would the row multiply of csrc/gecm_row.hpp gain from taking the limbs of its broadcast
// operand out of LDS (4 x ds_read_b128, every lane of a DPP row reading the same 16 bytes, requested one multiply
// ahead) instead of 16 x v_mov_b32_dpp row_newbcast?  Bodies of 16 "rows":
//   KIND 0  the row as it is:       mad(a_i*b) | dpp a_(i+1) | dpp q | mad(q*n) | dpp shl | mad(x16)
//   KIND 1  a from registers:       mad | dpp q | mad | dpp shl | mad           (lower bound: the broadcast is free)
//   KIND 2  KIND 1 + per body: 1 ds_write_b32 of the "result", 4 ds_read_b128 for the NEXT body's a (waited for at the
//           top of the next body): what a prefetching kernel would execute
//   KIND 3  KIND 2 with 16 ds_swizzle broadcasts instead of the 4 reads (the ALDS variant, but requested a body ahead)
// cycles per body per wavefront at 1, 2 and 4 wavefronts per SIMD (workgroups of one wavefront, 2 KB of LDS each).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// registers: v[40:41] T, v42 b, v43 n, v44 q, v[46:47] (L,0), v48 a (dpp variant), v[60:75] a-limbs from LDS (current),
// v[80:95] a-limbs being fetched for the next body, v50 = LDS byte address of this lane's row base, v51 = own slot address
#define CLOB "v40", "v41", "v42", "v43", "v44", "v46", "v47", "v48", "v50", "v51", "vcc", \
    "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", \
    "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95"

#define ROW_DPP(i)                                                                                         \
    "v_mad_i64_i32 v[40:41], vcc, v48, v42, v[40:41]\n\t"                                                  \
    "v_mov_b32_dpp v48, v42 row_newbcast:" #i " row_mask:0xf bank_mask:0xf\n\t"                            \
    "s_nop 0\n\t"                                                                                          \
    "v_mov_b32_dpp v44, v40 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"                                 \
    "v_mad_u64_u32 v[40:41], vcc, v44, v43, v[40:41]\n\t"                                                  \
    "s_nop 1\n\t"                                                                                          \
    "v_mov_b32_dpp v46, v40 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                         \
    "v_mad_i64_i32 v[40:41], vcc, v41, 16, v[46:47]\n\t"
#define ROW_REG(r)                                                                                         \
    "v_mad_i64_i32 v[40:41], vcc, v" #r ", v42, v[40:41]\n\t"                                              \
    "s_nop 1\n\t"                                                                                          \
    "v_mov_b32_dpp v44, v40 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"                                 \
    "v_mad_u64_u32 v[40:41], vcc, v44, v43, v[40:41]\n\t"                                                  \
    "s_nop 1\n\t"                                                                                          \
    "v_mov_b32_dpp v46, v40 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                         \
    "v_mad_i64_i32 v[40:41], vcc, v41, 16, v[46:47]\n\t"

template <int KIND>
__global__ void __launch_bounds__(64, 4) k(uint32_t *out, uint32_t iters, uint32_t seed)
{
    __shared__ uint32_t lds[4 * 16 * 8];          // 4 rows x 16 limbs x 8 slots
    for (int i = threadIdx.x; i < 4 * 16 * 8; i += 64) lds[i] = seed + i;
    __syncthreads();
    uint32_t r;
    const uint32_t rowbase = (uint32_t)(uintptr_t)lds + (threadIdx.x >> 4) * 64u;     // this DPP row's 16 limbs (64 B)
    const uint32_t own = rowbase + (threadIdx.x & 15u) * 4u;
    asm volatile("v_mov_b32 v40, %0\n\tv_mov_b32 v41, 0\n\tv_mov_b32 v42, %1\n\tv_mov_b32 v43, 0x0fffffff\n\tv_mov_b32 v44, 0\n\t"
                 "v_mov_b32 v46, 0\n\tv_mov_b32 v47, 0\n\tv_mov_b32 v48, %1\n\tv_mov_b32 v50, %2\n\tv_mov_b32 v51, %3"
                 : : "v"(seed + threadIdx.x), "v"(seed * 3 + threadIdx.x), "v"(rowbase), "v"(own) : CLOB);
    if (KIND >= 2)
        asm volatile("ds_read_b128 v[80:83], v50\n\tds_read_b128 v[84:87], v50 offset:16\n\tds_read_b128 v[88:91], v50 offset:32\n\t"
                     "ds_read_b128 v[92:95], v50 offset:48" : : : CLOB);
    for (uint32_t i = 0; i < iters; i++) {
        if (KIND == 0) {
            asm volatile(ROW_DPP(1) ROW_DPP(2) ROW_DPP(3) ROW_DPP(4) ROW_DPP(5) ROW_DPP(6) ROW_DPP(7) ROW_DPP(8)
                         ROW_DPP(9) ROW_DPP(10) ROW_DPP(11) ROW_DPP(12) ROW_DPP(13) ROW_DPP(14) ROW_DPP(15) ROW_DPP(0) : : : CLOB);
        } else {
            if (KIND >= 2) {
                // the limbs requested during the previous body arrive; move them to the working set
                asm volatile("s_waitcnt lgkmcnt(0)\n\t"
                             "v_mov_b32 v60, v80\n\tv_mov_b32 v61, v81\n\tv_mov_b32 v62, v82\n\tv_mov_b32 v63, v83\n\t"
                             "v_mov_b32 v64, v84\n\tv_mov_b32 v65, v85\n\tv_mov_b32 v66, v86\n\tv_mov_b32 v67, v87\n\t"
                             "v_mov_b32 v68, v88\n\tv_mov_b32 v69, v89\n\tv_mov_b32 v70, v90\n\tv_mov_b32 v71, v91\n\t"
                             "v_mov_b32 v72, v92\n\tv_mov_b32 v73, v93\n\tv_mov_b32 v74, v94\n\tv_mov_b32 v75, v95" : : : CLOB);
                if (KIND == 2)
                    asm volatile("ds_write_b32 v51, v40\n\t"
                                 "ds_read_b128 v[80:83], v50\n\tds_read_b128 v[84:87], v50 offset:16\n\t"
                                 "ds_read_b128 v[88:91], v50 offset:32\n\tds_read_b128 v[92:95], v50 offset:48" : : : CLOB);
                else
                    asm volatile("ds_swizzle_b32 v80, v40 offset:swizzle(BROADCAST,16,0)\n\tds_swizzle_b32 v81, v40 offset:swizzle(BROADCAST,16,1)\n\t"
                                 "ds_swizzle_b32 v82, v40 offset:swizzle(BROADCAST,16,2)\n\tds_swizzle_b32 v83, v40 offset:swizzle(BROADCAST,16,3)\n\t"
                                 "ds_swizzle_b32 v84, v40 offset:swizzle(BROADCAST,16,4)\n\tds_swizzle_b32 v85, v40 offset:swizzle(BROADCAST,16,5)\n\t"
                                 "ds_swizzle_b32 v86, v40 offset:swizzle(BROADCAST,16,6)\n\tds_swizzle_b32 v87, v40 offset:swizzle(BROADCAST,16,7)\n\t"
                                 "ds_swizzle_b32 v88, v40 offset:swizzle(BROADCAST,16,8)\n\tds_swizzle_b32 v89, v40 offset:swizzle(BROADCAST,16,9)\n\t"
                                 "ds_swizzle_b32 v90, v40 offset:swizzle(BROADCAST,16,10)\n\tds_swizzle_b32 v91, v40 offset:swizzle(BROADCAST,16,11)\n\t"
                                 "ds_swizzle_b32 v92, v40 offset:swizzle(BROADCAST,16,12)\n\tds_swizzle_b32 v93, v40 offset:swizzle(BROADCAST,16,13)\n\t"
                                 "ds_swizzle_b32 v94, v40 offset:swizzle(BROADCAST,16,14)\n\tds_swizzle_b32 v95, v40 offset:swizzle(BROADCAST,16,15)" : : : CLOB);
            }
            asm volatile(ROW_REG(60) ROW_REG(61) ROW_REG(62) ROW_REG(63) ROW_REG(64) ROW_REG(65) ROW_REG(66) ROW_REG(67)
                         ROW_REG(68) ROW_REG(69) ROW_REG(70) ROW_REG(71) ROW_REG(72) ROW_REG(73) ROW_REG(74) ROW_REG(75) : : : CLOB);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\tv_add_u32 %0, v40, v42\n\tv_add_u32 %0, %0, v44\n\tv_add_u32 %0, %0, v80" : "=v"(r) : : CLOB);
    out[blockIdx.x * 64 + threadIdx.x] = r;
}

template <int KIND>
static int run(const char *name)
{
    uint32_t *d;
    CK(hipMalloc(&d, 4096 * 64 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const uint32_t iters = 4000;
    for (unsigned blocks : {1024u, 2048u, 4096u}) {
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d, 10u, 1u);
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d, iters, 1u);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-58s waves/SIMD=%u  %8.1f cycles@2.4GHz per 16-row body per wave\n", name, blocks / 1024, ms * 1e-3 * 2.4e9 / iters);
    }
    CK(hipFree(d));
    return 0;
}

int main()
{
    run<0>("16 rows, a_i by DPP (the kernel's row)");
    run<1>("16 rows, a_i in registers (no broadcast at all)");
    run<2>("16 rows, a_i by 4 ds_read_b128 requested a body ahead");
    run<3>("16 rows, a_i by 16 ds_swizzle requested a body ahead");
    return 0;
}
