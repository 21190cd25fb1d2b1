// Does a wave64 VALU instruction cost 4 cycles whatever EXEC holds, or are quarter-waves with no active lane skipped?
// Build: hipcc -O3 --offload-arch=gfx950 -o valu_exec_mask scripts/micro/valu_exec_mask.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#ifndef OP
#define OP "v_fma_f32"
#endif
__global__ __launch_bounds__(256) void k(float *out, int iters, uint64_t mask_lo_hi){
    const uint32_t lane = threadIdx.x & 63u;
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const float m = 1.0000001f, c = 1e-7f;
    if((mask_lo_hi >> lane) & 1ull){
        for(int i = 0; i < iters; ++i){
            // inline assembly so that the compiler neither packs pairs into v_pk_fma_f32 nor rewrites the loop
            asm volatile(OP " %0, %0, %8, %9\n " OP " %1, %1, %8, %9\n " OP " %2, %2, %8, %9\n " OP " %3, %3, %8, %9\n"
                         OP " %4, %4, %8, %9\n " OP " %5, %5, %8, %9\n " OP " %6, %6, %8, %9\n " OP " %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
int main(){
    const int blocks = 256 * 8, iters = 1 << 16;
    float *d; hipMalloc(&d, blocks * 256 * sizeof(float));
    struct { const char *name; uint64_t mask; } cases[] = {
        {"all 64 lanes", ~0ull}, {"lanes 0-15", 0xFFFFull}, {"lanes 0-11", 0xFFFull}, {"lanes 0-10", 0x7FFull}, {"lanes 0-9", 0x3FFull},
        {"lanes 0-8", 0x1FFull}, {"lanes 0-7", 0xFFull}, {"lane 0 only", 1ull},
        {"9 lanes spread (0,7,..,56)", 0x0101010101010101ull | (1ull << 63)}, {"8 lanes spread", 0x0101010101010101ull},
        {"lanes 0-7 and 32-39", 0x000000FF000000FFull}, {"lanes 0-3 and 32-35 and 16-19", 0x0000000F000F000Full}};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for(auto &cs : cases){
        k<<<blocks, 256>>>(d, 1024, cs.mask);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<<<blocks, 256>>>(d, iters, cs.mask);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // per SIMD: blocks*4 waves / 1024 SIMDs waves, each iters*8 FMAs
        double wave_insts_per_simd = (double) blocks * 4 / 1024.0 * iters * 8.0;
        printf("%-45s %8.3f ms  -> %.2f cycles per wave-instruction at 2.4 GHz\n", cs.name, ms, ms * 1e-3 * 2.4e9 / wave_insts_per_simd);
    }
    return 0;
}
