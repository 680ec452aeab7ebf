// Does anything make the fabric fetch LESS than 128 bytes for a 16-byte gather?  (every TCC_EA0_RDREQ of the big layout's scan is a
// 128-byte request for a 64-byte line.)  Random GRAN-aligned 16-byte gathers out of one large allocation, per load flavour (cache-policy bits of
// global_load_dwordx4: none, nt, sc0, sc1, sc0 sc1, sc0 sc1 nt) and per allocation kind (hipMalloc; hipExtMallocWithFlags uncached / fine-grained):
// rate in G gathers/s; the request-size counters come from rand_gran_pmc.sh.  A measurement program for profiles/, not part of the library.
//
//   rand_gran <span GiB> <gathers M> <alloc: 0 plain | 1 uncached | 2 finegrained> <flavour 0..5> [gran bytes = 64] [waves per SIMD = 4] [pairs = 0]
//   pairs = 1: every lane asks for BOTH 64-byte halves of a random 128-byte line with two consecutive loads (what a two-half line costs when the
//   sibling half is wanted too); the rate counts lines.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31);
}
template <int F>
__device__ __forceinline__ void ld(u32x4& t, const uint8_t* p) {
    if constexpr (F == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(t) : "v"(p) : "memory");
    else if constexpr (F == 1) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(t) : "v"(p) : "memory");
    else if constexpr (F == 2) asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(t) : "v"(p) : "memory");
    else if constexpr (F == 3) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(t) : "v"(p) : "memory");
    else if constexpr (F == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(t) : "v"(p) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "=v"(t) : "v"(p) : "memory");
}

template <int F, int U>
__global__ void __launch_bounds__(256) gather_kernel(const uint8_t* __restrict__ base, const uint64_t units, const uint32_t gran, const uint64_t perLane, const uint64_t seed,
                                                     unsigned long long* __restrict__ sink, const int pairs) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t s = mix64(seed + gid * 0x9E3779B97F4A7C15ULL);
    uint32_t acc = 0; const uint8_t* last = base;
    for (uint64_t it = 0; it < perLane; it += U) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint8_t* p;
            if (pairs && (u & 1)) p = reinterpret_cast<const uint8_t*>(reinterpret_cast<uint64_t>(last) ^ 64ULL);      // the sibling half of the line just asked for
            else {
                s = s * 6364136223846793005ULL + 1442695040888963407ULL;
                p = base + __umul64hi((s >> 11) << 11, units) * (uint64_t)gran + ((gid & 3) << 4);
            }
            last = p;
            ld<F>(v[u], p);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < U; u++) { asm volatile("" : "+v"(v[u])); acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w; }
    }
    if (acc == 0x12345677u) atomicAdd(sink, 1ULL);
}

template <int F>
static void launch(int blocks, const uint8_t* base, uint64_t units, uint32_t gran, uint64_t perLane, uint64_t seed, unsigned long long* sink, int pairs) {
    gather_kernel<F, 4><<<blocks, 256>>>(base, units, gran, perLane, seed, sink, pairs);
}

int main(int argc, char** argv) {
    if (argc < 5) { fprintf(stderr, "usage: rand_gran <span GiB> <gathers M> <alloc 0|1|2> <flavour 0..5> [gran=64] [waves/SIMD=4] [pairs=0]\n"); return 2; }
    const double spanGiB = atof(argv[1]);
    const uint64_t gathers = (uint64_t)(atof(argv[2]) * 1e6);
    const int alloc = atoi(argv[3]), fl = atoi(argv[4]);
    const uint32_t gran = argc > 5 ? (uint32_t)atoi(argv[5]) : 64u;
    const int wps = argc > 6 ? atoi(argv[6]) : 4, pairs = argc > 7 ? atoi(argv[7]) : 0;
    const uint64_t span = (uint64_t)(spanGiB * 1024.0 * 1024.0 * 1024.0) & ~(uint64_t)127;
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    uint8_t* base = nullptr;
    if (alloc == 0) CHECK(hipMalloc(&base, span));
    else if (alloc == 1) CHECK(hipExtMallocWithFlags((void**)&base, span, hipDeviceMallocUncached));
    else CHECK(hipExtMallocWithFlags((void**)&base, span, hipDeviceMallocFinegrained));
    unsigned long long* sink = nullptr; CHECK(hipMalloc(&sink, 8)); CHECK(hipMemset(sink, 0, 8));
    CHECK(hipMemset(base, 0x5a, span)); CHECK(hipDeviceSynchronize());
    const int blocks = prop.multiProcessorCount * wps;
    const uint64_t lanes = (uint64_t)blocks * 256;
    uint64_t perLane = (gathers + lanes - 1) / lanes; perLane = ((perLane + 3) / 4) * 4;
    const uint64_t units = pairs ? span / 128 : span / gran;
    const uint32_t g = pairs ? 128u : gran;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    double best = 1e30;
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(e0, 0));
        switch (fl) {
        case 0: launch<0>(blocks, base, units, g, perLane, 77 + rep, sink, pairs); break;
        case 1: launch<1>(blocks, base, units, g, perLane, 77 + rep, sink, pairs); break;
        case 2: launch<2>(blocks, base, units, g, perLane, 77 + rep, sink, pairs); break;
        case 3: launch<3>(blocks, base, units, g, perLane, 77 + rep, sink, pairs); break;
        case 4: launch<4>(blocks, base, units, g, perLane, 77 + rep, sink, pairs); break;
        default: launch<5>(blocks, base, units, g, perLane, 77 + rep, sink, pairs); break;
        }
        CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    const uint64_t done = perLane * lanes / (pairs ? 2 : 1);
    static const char* FN[] = {"plain", "nt", "sc0", "sc1", "sc0 sc1", "sc0 sc1 nt"}; static const char* AN[] = {"hipMalloc", "uncached", "finegrained"};
    printf("{\"span_GiB\": %.1f, \"alloc\": \"%s\", \"flavour\": \"%s\", \"gran\": %u, \"pairs\": %d, \"waves_per_simd\": %d, \"gathers\": %llu, \"ms\": %.3f, \"G_per_s\": %.2f}\n",
           spanGiB, AN[alloc], FN[fl], g, pairs, wps, (unsigned long long)done, best, done / best / 1e6);
    return 0;
}
