// Does gfx950 read an LDS dword at ANY byte address?  (round 5: a presence filter addressed by byte -- one shift instead of shift + mask per probe --
// needs it.)  hipcc --offload-arch=gfx950 -O3 profiles/microbench/lds_unaligned.hip -o /tmp/lds_unaligned && /tmp/lds_unaligned
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(uint32_t* out) {
    __shared__ uint32_t s[64];
    s[threadIdx.x] = 0x03020100u + 0x04040404u * threadIdx.x;     // byte i of the array holds i
    __syncthreads();
    const uint32_t addr = (uint32_t)(size_t)(__attribute__((address_space(3))) uint32_t*)s + threadIdx.x;   // byte address base + lane
    uint32_t v;
    asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
    out[threadIdx.x] = v;
}
int main() {
    uint32_t* d; hipMalloc(&d, 64 * 4);
    k<<<1, 64>>>(d);
    uint32_t h[64]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int ok = 1;
    for (int i = 0; i < 64; i++) { const uint32_t want = (uint32_t)i | ((uint32_t)(i + 1) << 8) | ((uint32_t)(i + 2) << 16) | ((uint32_t)(i + 3) << 24); if (h[i] != want) ok = 0; }
    printf("{\"lds_unaligned_dword_reads\": %s, \"lane1\": \"0x%08x\", \"lane2\": \"0x%08x\", \"lane3\": \"0x%08x\"}\n", ok ? "true" : "false", h[1], h[2], h[3]);
    return 0;
}
