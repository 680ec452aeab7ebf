// Micro-benchmark: issue cost of individual gfx950 VALU instructions (inline asm, 8 independent dependency chains,
// W waves per SIMD).  Prints cycles per wave-instruction per SIMD at 2.4 GHz.
// hipcc --offload-arch=gfx950 -O3 valu_ops.hip -o valu_ops
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define CHAIN8(STR) \
    asm volatile(STR : "+v"(a0), "+v"(b0) : "v"(c) ); asm volatile(STR : "+v"(a1), "+v"(b1) : "v"(c) ); \
    asm volatile(STR : "+v"(a2), "+v"(b2) : "v"(c) ); asm volatile(STR : "+v"(a3), "+v"(b3) : "v"(c) ); \
    asm volatile(STR : "+v"(a4), "+v"(b4) : "v"(c) ); asm volatile(STR : "+v"(a5), "+v"(b5) : "v"(c) ); \
    asm volatile(STR : "+v"(a6), "+v"(b6) : "v"(c) ); asm volatile(STR : "+v"(a7), "+v"(b7) : "v"(c) );

template <int KIND>
__global__ void k(uint32_t* out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    uint32_t b0 = a0 ^ 1, b1 = a1 ^ 2, b2 = a2 ^ 3, b3 = a3 ^ 4, b4 = a4 ^ 5, b5 = a5 ^ 6, b6 = a6 ^ 7, b7 = a7 ^ 8;
    uint32_t c = seed * 77 + threadIdx.x;
    uint64_t q0 = a0, q1 = a1, q2 = a2, q3 = a3;
#pragma unroll 4
    for (int i = 0; i < iters; i++) {
        if (KIND == 0)  { CHAIN8("v_xor_b32 %0, %0, %2") }
        if (KIND == 1)  { CHAIN8("v_mul_lo_u32 %0, %0, %2") }
        if (KIND == 2)  { CHAIN8("v_mul_u32_u24 %0, %0, %2") }
        if (KIND == 3)  { CHAIN8("v_mad_u32_u24 %0, %0, %2, %1") }
        if (KIND == 4)  { CHAIN8("v_add3_u32 %0, %0, %2, %1") }
        if (KIND == 5)  { CHAIN8("v_alignbit_b32 %0, %0, %1, %2") }
        if (KIND == 6)  { CHAIN8("v_bfe_u32 %0, %0, 3, 15") }
        if (KIND == 7)  { CHAIN8("v_and_or_b32 %0, %0, %2, %1") }
        if (KIND == 8)  { CHAIN8("v_lshl_add_u32 %0, %0, 2, %1") }
        if (KIND == 9)  { CHAIN8("v_cndmask_b32 %0, %0, %1, vcc") }
        if (KIND == 10) { CHAIN8("v_cmp_eq_u32_sdwa vcc, %0, %2 src0_sel:WORD_1 src1_sel:DWORD") }
        if (KIND == 11) { CHAIN8("v_cmp_gt_u32 vcc, %0, %2") }
        if (KIND == 12) { CHAIN8("v_lshrrev_b32 %0, %2, %0") }
        if (KIND == 13) { CHAIN8("v_bitop3_b32 %0, %0, %2, %1 bitop3:0x6c") }
        if (KIND == 14) { CHAIN8("v_perm_b32 %0, %0, %1, %2") }
        if (KIND == 15) { CHAIN8("v_pk_min_u16 %0, %0, %2") }
        if (KIND == 16) { CHAIN8("v_mul_hi_u32 %0, %0, %2") }
        if (KIND == 17) { CHAIN8("v_min_u32 %0, %0, %2") }
        if (KIND == 18) { CHAIN8("v_mul_u32_u24_sdwa %0, %0, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD") }
        if (KIND == 30) { CHAIN8("v_add_u32 %0, %0, %2") }
        if (KIND == 31) { CHAIN8("v_and_b32 %0, %0, %2") }
        if (KIND == 32) { CHAIN8("v_or_b32 %0, %0, %2") }
        if (KIND == 33) { CHAIN8("v_lshlrev_b32 %0, 3, %0") }
        if (KIND == 34) { CHAIN8("v_sub_u32 %0, %0, %2") }
        if (KIND == 35) { CHAIN8("v_cndmask_b32 %0, %0, %1, s[10:11]") }
        if (KIND == 36) { CHAIN8("v_mov_b32 %0, %1") }
        if (KIND == 37) { CHAIN8("v_max_u32 %0, %0, %2") }
        if (KIND == 38) { CHAIN8("v_cmp_ne_u32 s[10:11], %0, %2") }
        if (KIND == 39) { CHAIN8("v_ffbl_b32 %0, %0") }
        if (KIND == 40) { CHAIN8("v_not_b32 %0, %0") }
        if (KIND == 41) { CHAIN8("v_xad_u32 %0, %0, %2, %1") }
        if (KIND == 42) { CHAIN8("v_or3_b32 %0, %0, %2, %1") }
        if (KIND == 43) { CHAIN8("v_and_b32 %0, s12, %0") }
        if (KIND == 44) { CHAIN8("v_lshrrev_b32 %0, 3, %0") }
        if (KIND == 45) { CHAIN8("v_bfi_b32 %0, %0, %2, %1") }
        if (KIND == 46) { CHAIN8("v_xor_b32 %0, 0x12345678, %0") }
        if (KIND == 47) { CHAIN8("v_add_co_u32 %0, vcc, %0, %2") }
        if (KIND == 48) { CHAIN8("v_bitop3_b32 %0, %0, s12, %1 bitop3:0xec") }
        if (KIND == 49) { CHAIN8("v_mov_b64 %0, %0") }
        if (KIND == 50) { CHAIN8("v_bcnt_u32_b32 %0, %0, %2") }
        if (KIND == 51) { CHAIN8("v_mbcnt_lo_u32_b32 %0, %0, %2") }
        if (KIND == 52) { CHAIN8("v_lshl_or_b32 %0, %0, 2, %1") }
        if (KIND == 53) { CHAIN8("v_ashrrev_i32 %0, 4, %0") }
        if (KIND == 54) { CHAIN8("v_subrev_u32 %0, s12, %0") }
        if (KIND == 60) { CHAIN8("v_fma_f32 %0, %0, %2, %1") }
        if (KIND == 61) { CHAIN8("v_add_f32 %0, %0, %2") }
        if (KIND == 62) { CHAIN8("v_mul_f32 %0, %0, %2") }
        if (KIND == 63) { CHAIN8("v_and_b32 %0, 0x0f0f0f0f, %0") }
        if (KIND == 64) { CHAIN8("v_cvt_f32_u32 %0, %0") }
        if (KIND == 70) { CHAIN8("v_and_b32 %0, %2, %0") }
        if (KIND == 71) { CHAIN8("v_and_b32 %0, s12, %0") }
        if (KIND == 72) { CHAIN8("v_and_b32 %0, 0x0f0f0f0f, %0") }
        if (KIND == 73) { CHAIN8("v_and_b32 %0, 15, %0") }
        if (KIND == 74) { CHAIN8("v_lshrrev_b32 %0, %2, %0") }
        if (KIND == 75) { CHAIN8("v_lshrrev_b32 %0, s12, %0") }
        if (KIND == 76) { CHAIN8("v_lshrrev_b32 %0, 3, %0") }
        if (KIND == 77) { CHAIN8("v_mul_lo_u32 %0, %0, %2") }
        if (KIND == 78) { CHAIN8("v_mul_lo_u32 %0, %0, s12") }
        if (KIND == 79) { CHAIN8("v_add_u32 %0, %2, %0") }
        if (KIND == 80) { CHAIN8("v_add_u32 %0, s12, %0") }
        if (KIND == 81) { CHAIN8("v_add_u32 %0, 17, %0") }
        if (KIND == 82) { CHAIN8("v_alignbit_b32 %0, %0, %1, s12") }
        if (KIND == 83) { CHAIN8("v_alignbit_b32 %0, %0, %1, 7") }
        if (KIND == 84) { CHAIN8("v_cndmask_b32 %0, %0, %1, s[10:11]") }
        if (KIND == 85) { CHAIN8("v_cmp_gt_u32 s[10:11], %0, %2") }
        if (KIND == 86) { CHAIN8("v_xor_b32 %0, %2, %0") }
        if (KIND == 87) { CHAIN8("v_xor_b32 %0, s12, %0") }
        if (KIND == 20) {  // 64-bit: compare
            asm volatile("v_cmp_gt_u64 vcc, %0, %1" :: "v"(q0), "v"(q1)); asm volatile("v_cmp_gt_u64 vcc, %0, %1" :: "v"(q1), "v"(q2));
            asm volatile("v_cmp_gt_u64 vcc, %0, %1" :: "v"(q2), "v"(q3)); asm volatile("v_cmp_gt_u64 vcc, %0, %1" :: "v"(q3), "v"(q0));
            asm volatile("v_cmp_gt_u64 vcc, %0, %1" :: "v"(q0), "v"(q2)); asm volatile("v_cmp_gt_u64 vcc, %0, %1" :: "v"(q1), "v"(q3));
            asm volatile("v_cmp_gt_u64 vcc, %0, %1" :: "v"(q2), "v"(q0)); asm volatile("v_cmp_gt_u64 vcc, %0, %1" :: "v"(q3), "v"(q1));
        }
        if (KIND == 21) {  // 64-bit shift
            asm volatile("v_lshrrev_b64 %0, 2, %0" : "+v"(q0)); asm volatile("v_lshrrev_b64 %0, 2, %0" : "+v"(q1));
            asm volatile("v_lshrrev_b64 %0, 2, %0" : "+v"(q2)); asm volatile("v_lshrrev_b64 %0, 2, %0" : "+v"(q3));
            asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(q0)); asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(q1));
            asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(q2)); asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(q3));
        }
        if (KIND == 22) {  // 64-bit mad
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q0) : "v"(a0), "v"(c) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q1) : "v"(a1), "v"(c) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q2) : "v"(a2), "v"(c) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q3) : "v"(a3), "v"(c) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q0) : "v"(a4), "v"(c) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q1) : "v"(a5), "v"(c) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q2) : "v"(a6), "v"(c) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q3) : "v"(a7), "v"(c) : "vcc");
        }
        if (KIND == 23) {  // 64-bit add
            asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q0) : "v"(q1)); asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q1) : "v"(q2));
            asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q2) : "v"(q3)); asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q3) : "v"(q0));
            asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q0) : "v"(q2)); asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q1) : "v"(q3));
            asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q2) : "v"(q0)); asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q3) : "v"(q1));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ b0 ^ b1 ^ b2 ^ b3 ^ b4 ^ b5 ^ b6 ^ b7 ^ (uint32_t)(q0 ^ q1 ^ q2 ^ q3);
}
template <int KIND> void run(const char* name, int wavesPerSimd) {
    const int cu = 256, threads = 256 * wavesPerSimd;
    uint32_t* d; hipMalloc(&d, (size_t)cu * threads * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 100000;
    k<KIND><<<cu, threads>>>(d, 1000, 12345u);
    hipDeviceSynchronize();
    hipEventRecord(e0); k<KIND><<<cu, threads>>>(d, iters, 12345u); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waveInstr = (double)iters * 8 * wavesPerSimd;
    printf("%-22s W=%d  %.2f cycles/wave-instr/SIMD\n", name, wavesPerSimd, ms * 1e6 / waveInstr * 2.4);
    hipFree(d);
}
int main() {
    for (int w : {4}) {
        run<0>("v_xor_b32", w); run<1>("v_mul_lo_u32", w); run<2>("v_mul_u32_u24", w); run<3>("v_mad_u32_u24", w); run<4>("v_add3_u32", w);
        run<5>("v_alignbit_b32", w); run<6>("v_bfe_u32", w); run<7>("v_and_or_b32", w); run<8>("v_lshl_add_u32", w); run<9>("v_cndmask_b32", w);
        run<10>("v_cmp_eq_u32_sdwa", w); run<11>("v_cmp_gt_u32", w); run<12>("v_lshrrev_b32", w); run<13>("v_bitop3_b32", w); run<14>("v_perm_b32", w);
        run<15>("v_pk_min_u16", w); run<16>("v_mul_hi_u32", w); run<17>("v_min_u32", w); run<18>("v_mul_u32_u24_sdwa", w);
        run<30>("v_add_u32", w); run<31>("v_and_b32", w); run<32>("v_or_b32", w); run<33>("v_lshlrev_b32", w); run<34>("v_sub_u32", w);
        run<35>("v_cndmask_b32 sgpr", w); run<36>("v_mov_b32", w); run<37>("v_max_u32", w); run<38>("v_cmp_ne_u32 ->sgpr", w); run<39>("v_ffbl_b32", w);
        run<40>("v_not_b32", w); run<41>("v_xad_u32", w); run<42>("v_or3_b32", w); run<43>("v_and_b32 sgpr", w); run<44>("v_lshrrev_b32 imm", w);
        run<45>("v_bfi_b32", w); run<46>("v_xor_b32 literal", w); run<47>("v_add_co_u32", w); run<48>("v_bitop3 sgpr", w); run<50>("v_bcnt_u32_b32", w);
        run<51>("v_mbcnt_lo", w); run<52>("v_lshl_or_b32", w); run<53>("v_ashrrev_i32", w); run<54>("v_subrev_u32 sgpr", w);
        run<60>("v_fma_f32", w); run<61>("v_add_f32", w); run<62>("v_mul_f32", w); run<63>("v_and_b32 literal", w); run<64>("v_cvt_f32_u32", w);
        printf("-- operand source --\n");
        run<70>("and vgpr", w); run<71>("and sgpr", w); run<72>("and literal", w); run<73>("and inline", w);
        run<74>("lshr vgpr", w); run<75>("lshr sgpr", w); run<76>("lshr imm", w);
        run<77>("mul_lo vgpr", w); run<78>("mul_lo sgpr", w);
        run<79>("add vgpr", w); run<80>("add sgpr", w); run<81>("add inline", w);
        run<82>("alignbit sgpr", w); run<83>("alignbit imm", w); run<84>("cndmask sgprmask", w); run<85>("cmp->sgpr", w);
        run<86>("xor vgpr", w); run<87>("xor sgpr", w);
        run<20>("v_cmp_gt_u64", w); run<21>("v_lshrrev_b64", w); run<22>("v_mad_u64_u32", w); run<23>("v_lshl_add_u64", w);
    }
    return 0;
}
