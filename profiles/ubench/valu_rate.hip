// Micro-benchmark: integer VALU issue rate on gfx950 with W waves per SIMD (answers: how many cycles does a
// wave64 integer VALU instruction cost a SIMD?).  hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int KIND>
__global__ void k(uint32_t* out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    uint64_t b0 = a0, b1 = a1, b2 = a2, b3 = a3;
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) {   // 32-bit logic/add mix, 8 independent chains, 16 VALU ops per iteration
            a0 = (a0 ^ seed) + 1; a1 = (a1 ^ seed) + 2; a2 = (a2 ^ seed) + 3; a3 = (a3 ^ seed) + 4;
            a4 = (a4 ^ seed) + 5; a5 = (a5 ^ seed) + 6; a6 = (a6 ^ seed) + 7; a7 = (a7 ^ seed) + 8;
        } else if (KIND == 1) {  // v_mul_lo_u32
            a0 = a0 * seed + 1; a1 = a1 * seed + 2; a2 = a2 * seed + 3; a3 = a3 * seed + 4;
            a4 = a4 * seed + 5; a5 = a5 * seed + 6; a6 = a6 * seed + 7; a7 = a7 * seed + 8;
        } else if (KIND == 2) {  // mul24
            a0 = __umul24(a0, seed) + 1; a1 = __umul24(a1, seed) + 2; a2 = __umul24(a2, seed) + 3; a3 = __umul24(a3, seed) + 4;
            a4 = __umul24(a4, seed) + 5; a5 = __umul24(a5, seed) + 6; a6 = __umul24(a6, seed) + 7; a7 = __umul24(a7, seed) + 8;
        } else if (KIND == 3) {  // 64-bit shift + compare/select
            b0 = (b0 << 2) ^ (b0 > b1 ? b0 : b1); b1 = (b1 >> 3) ^ (b1 > b2 ? b1 : b2);
            b2 = (b2 << 5) ^ (b2 > b3 ? b2 : b3); b3 = (b3 >> 7) ^ (b3 > b0 ? b3 : b0);
        } else {                 // alignbit
            a0 = __builtin_amdgcn_alignbit(a0, a1, a2); a1 = __builtin_amdgcn_alignbit(a1, a2, a3); a2 = __builtin_amdgcn_alignbit(a2, a3, a4);
            a3 = __builtin_amdgcn_alignbit(a3, a4, a5); a4 = __builtin_amdgcn_alignbit(a4, a5, a6); a5 = __builtin_amdgcn_alignbit(a5, a6, a7);
            a6 = __builtin_amdgcn_alignbit(a6, a7, a0); a7 = __builtin_amdgcn_alignbit(a7, a0, a1);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (uint32_t)(b0 ^ b1 ^ b2 ^ b3);
}
template <int KIND> void run(const char* name, int opsPerIter, int wavesPerSimd) {
    const int cu = 256, threads = 256 * wavesPerSimd;   // 4 SIMDs x W waves x 64 lanes per block... one block per CU
    const int blockThreads = threads > 1024 ? 1024 : threads;
    const int blocksPerCU = threads / blockThreads;
    uint32_t* d; hipMalloc(&d, (size_t)cu * threads * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 200000;
    k<KIND><<<cu * blocksPerCU, blockThreads>>>(d, 1000, 12345u);
    hipDeviceSynchronize();
    hipEventRecord(e0); k<KIND><<<cu * blocksPerCU, blockThreads>>>(d, iters, 12345u); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waveInstr = (double)iters * opsPerIter * wavesPerSimd;            // per SIMD
    printf("%-28s W=%d  %.3f ms  -> %.2f ns per wave-instr per SIMD (%.2f cycles @2.4GHz)\n", name, wavesPerSimd, ms, ms * 1e6 / waveInstr, ms * 1e6 / waveInstr * 2.4);
    hipFree(d);
}
int main() {
    for (int w : {1, 2, 4, 8}) {
        run<0>("xor+add (32-bit)", 16, w); run<1>("v_mul_lo_u32 + add", 16, w); run<2>("mul24 + add", 16, w);
        run<3>("64-bit shl/cmp/sel/xor", 4 * 5, w); run<4>("v_alignbit_b32", 8, w);
    }
    return 0;
}
