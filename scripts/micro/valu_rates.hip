// Issue cost of the VALU instructions the traversal and shading kernels are made of, 8 waves per SIMD of independent chains
// (inline assembly: the compiler neither packs nor rewrites).  Build: hipcc -O3 --offload-arch=gfx950 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters){
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const float m = 1.0000001f, c = 1e-7f;
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pm = {m, m}, pc = {c, c};
    uint32_t u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, um = 747796405u;
    for(int i = 0; i < iters; ++i){
        if(KIND == 0) asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(c));
        if(KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm), "v"(pc));
        if(KIND == 2) asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm));
        if(KIND == 3) asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4\n"
                                   : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(um));
        if(KIND == 4) asm volatile("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4\n"
                                   : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(um));
        if(KIND == 5) asm volatile("v_cvt_f32_u32_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_u32_sdwa %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                                   "v_cvt_f32_u32_sdwa %2, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_u32_sdwa %3, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                                   : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
        if(KIND == 6) asm volatile("v_max3_f32 %0, %0, %4, %5\n v_max3_f32 %1, %1, %4, %5\n v_max3_f32 %2, %2, %4, %5\n v_max3_f32 %3, %3, %4, %5\n"
                                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(c));
        if(KIND == 7) asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        if(KIND == 8) asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        if(KIND == 9) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1\n v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1\n"
                                   : "+v"(p0), "+v"(p1) : "v"(u0), "v"(um) : "vcc");
        if(KIND == 10) asm volatile("v_bfi_b32 %0, %4, %0, %1\n v_bfi_b32 %1, %4, %1, %2\n v_bfi_b32 %2, %4, %2, %3\n v_bfi_b32 %3, %4, %3, %0\n"
                                   : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(um));
        if(KIND == 11) asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc\n"
                                   : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) :: "vcc");
        if(KIND == 13) asm volatile("v_cndmask_b32_e64 %0, %0, %4, s[10:11]\n v_cndmask_b32_e64 %1, %1, %4, s[10:11]\n v_cndmask_b32_e64 %2, %2, %4, s[10:11]\n v_cndmask_b32_e64 %3, %3, %4, s[10:11]\n"
                                   : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(um) : "s10", "s11");
        if(KIND == 14) asm volatile("v_cndmask_b32_e32 %0, %0, %4, vcc\n v_cndmask_b32_e32 %1, %1, %4, vcc\n v_cndmask_b32_e32 %2, %2, %4, vcc\n v_cndmask_b32_e32 %3, %3, %4, vcc\n"
                                   : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(um) : "vcc");
        if(KIND == 15) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %4\n v_cndmask_b32_e32 %1, %1, %5, vcc\n v_cmp_lt_f32_e32 vcc, %2, %4\n v_cndmask_b32_e32 %3, %3, %5, vcc\n"
                                   : "+v"(a0), "+v"(u1), "+v"(a2), "+v"(u3) : "v"(m), "v"(um) : "vcc");
        if(KIND == 16) asm volatile("v_min_f32 %0, %0, %4\n v_max_f32 %1, %1, %4\n v_min_f32 %2, %2, %5\n v_max_f32 %3, %3, %5\n"
                                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(c));
        if(KIND == 17) asm volatile("v_add_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_add_f32 %2, %2, %5\n v_mul_f32 %3, %3, %4\n"
                                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(c));
        if(KIND == 18) asm volatile("v_lshl_add_u32 %0, %0, 2, %4\n v_and_or_b32 %1, %1, %4, %4\n v_add3_u32 %2, %2, %4, %4\n v_lshrrev_b32 %3, 3, %3\n"
                                   : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(um));
        if(KIND == 19) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %4\n v_cndmask_b32_e32 %1, %1, %5, vcc\n v_cndmask_b32_e32 %2, %2, %5, vcc\n v_cndmask_b32_e32 %3, %3, %5, vcc\n"
                                   : "+v"(a0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(m), "v"(um) : "vcc");
        if(KIND == 20) asm volatile("v_cmp_lt_f32_e64 s[10:11], %0, %4\n v_cndmask_b32_e64 %1, %1, %5, s[10:11]\n v_cndmask_b32_e64 %2, %2, %5, s[10:11]\n v_cndmask_b32_e64 %3, %3, %5, s[10:11]\n"
                                   : "+v"(a0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(m), "v"(um) : "s10", "s11");
        if(KIND == 21) asm volatile("s_mov_b64 vcc, exec\n v_cndmask_b32_e32 %0, %0, %4, vcc\n v_cndmask_b32_e32 %1, %1, %4, vcc\n v_cndmask_b32_e32 %2, %2, %4, vcc\n v_cndmask_b32_e32 %3, %3, %4, vcc\n"
                                   : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(um) : "vcc");
        if(KIND == 12) asm volatile("v_div_scale_f32 %0, vcc, %0, %4, %0\n v_div_scale_f32 %1, vcc, %1, %4, %1\n v_div_fmas_f32 %2, %2, %4, %5\n v_div_fixup_f32 %3, %3, %4, %5\n"
                                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(c) : "vcc");
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + p0.x + p0.y + p1.x + p1.y + p2.x + p3.y + (float) (u0 + u1 + u2 + u3);
}
template <int KIND> void run(const char *name, float *d){
    const int blocks = 256 * 8, iters = 1 << 15;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<blocks, 256>>>(d, 256); hipDeviceSynchronize();
    hipEventRecord(e0); k<KIND><<<blocks, 256>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double wave_insts_per_simd = (double) blocks * 4 / 1024.0 * iters * 4.0;
    printf("%-28s %8.3f ms -> %.2f cycles per wave-instruction at 2.4 GHz\n", name, ms, ms * 1e-3 * 2.4e9 / wave_insts_per_simd);
}
int main(){
    float *d; hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    run<0>("v_fma_f32", d); run<1>("v_pk_fma_f32 (2 per lane)", d); run<2>("v_pk_mul_f32", d); run<3>("v_mul_lo_u32", d); run<4>("v_mul_hi_u32", d);
    run<5>("v_cvt_f32_u32 sdwa", d); run<6>("v_max3_f32", d); run<7>("v_rcp_f32", d); run<8>("v_sqrt_f32", d); run<9>("v_mad_u64_u32", d);
    run<10>("v_bfi_b32", d); run<11>("v_cndmask_b32 (chain, vcc)", d); run<12>("div_scale/fmas/fixup mix", d);
    run<13>("v_cndmask_b32_e64 sgpr mask", d); run<14>("v_cndmask_b32_e32 vcc", d); run<15>("v_cmp + v_cndmask pairs", d); run<16>("v_min/max_f32", d); run<17>("v_add/mul_f32", d);
    run<18>("lshl_add/and_or/add3/lshr", d); run<19>("1 v_cmp + 3 v_cndmask (vcc)", d); run<20>("1 v_cmp + 3 v_cndmask (sgpr)", d); run<21>("s_mov vcc + 4 v_cndmask", d);
    return 0;
}
