// What the memory system gives a gather kernel: random 128-byte lines out of one large allocation, 16 bytes per lane (the big layout's tag-pair
// gather), at the occupancy and memory-level parallelism of the scan kernels -- over the whole span, and confined to slices of it the way a
// launch partitioned by line high bits would be ("partition each launch's probes by the line index's high bits so that
// concurrently resident waves gather inside one TLB-reachable slice").  The gap between the two columns is the most such a partition could return
// on the gathers, before the cost of binning the probes.  A measurement program for profiles/, not part of the library.
//
//   rand_lines <span GiB> <gathers (millions)> <slice MiB | 0 = whole span> [waves per SIMD = 4] [loads in flight per lane = 4] [bytes per lane = 16]
//              [dup = 1] [group = 1]
//   dup:   that many CONSECUTIVE load instructions of a lane ask for the same line (its next 16-byte words) -- the big layout's stream scan asks for a
//          minimizer run's line from up to four instructions, one per position of the lane; "gathers" and the rates count distinct (lane, line) pairs
//   group: that many adjacent lanes share a line within one instruction (positions of one run in neighbouring lanes)
//
// One launch per slice (the slices in a shuffled order, the launch's gathers uniform inside its slice); the time is all launches together, HIP events.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31);
}

template <int U, int BYTES>
__global__ void __launch_bounds__(256) gather_kernel(const uint8_t* __restrict__ base, const uint64_t sliceLines, const uint64_t perLane, const uint64_t seed,
                                                     unsigned long long* __restrict__ sink, const int dup, const int group) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t s = mix64(seed + (gid / (uint64_t)group) * 0x9E3779B97F4A7C15ULL);
    uint64_t line = 0; int left = 0;
    uint32_t acc = 0;
    for (uint64_t it = 0; it < perLane; it += U) {
        uint32_t v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (left == 0) {
                s = s * 6364136223846793005ULL + 1442695040888963407ULL;
                line = __umul64hi((s >> 11) << 11, sliceLines);              // uniform in [0, sliceLines)
                left = dup;
            }
            left--;
            const uint8_t* p = base + line * 128 + (((gid + left) & 7) << 4);
            if (BYTES == 16) { const uint4 t = *reinterpret_cast<const uint4*>(p); v[u] = t.x ^ t.y ^ t.z ^ t.w; }
            else if (BYTES == 8) { const uint2 t = *reinterpret_cast<const uint2*>(p); v[u] = t.x ^ t.y; }
            else v[u] = *reinterpret_cast<const uint32_t*>(p);
        }
#pragma unroll
        for (int u = 0; u < U; u++) acc ^= v[u];
    }
    if (acc == 0x12345677u) atomicAdd(sink, 1ULL);                            // keeps the loads
}

template <int BYTES>
static void launch(int U, int blocks, const uint8_t* base, uint64_t sliceLines, uint64_t perLane, uint64_t seed, unsigned long long* sink, int dup, int group) {
    switch (U) {
    case 1: gather_kernel<1, BYTES><<<blocks, 256>>>(base, sliceLines, perLane, seed, sink, dup, group); break;
    case 2: gather_kernel<2, BYTES><<<blocks, 256>>>(base, sliceLines, perLane, seed, sink, dup, group); break;
    case 4: gather_kernel<4, BYTES><<<blocks, 256>>>(base, sliceLines, perLane, seed, sink, dup, group); break;
    case 8: gather_kernel<8, BYTES><<<blocks, 256>>>(base, sliceLines, perLane, seed, sink, dup, group); break;
    default: gather_kernel<16, BYTES><<<blocks, 256>>>(base, sliceLines, perLane, seed, sink, dup, group); break;
    }
}

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: rand_lines <span GiB> <gathers M> <slice MiB|0> [waves/SIMD] [loads in flight] [bytes/lane]\n"); return 2; }
    const double spanGiB = atof(argv[1]);
    const uint64_t gathers = (uint64_t)(atof(argv[2]) * 1e6);
    const double sliceMiB = atof(argv[3]);
    const int wps = argc > 4 ? atoi(argv[4]) : 4, U = argc > 5 ? atoi(argv[5]) : 4, bytes = argc > 6 ? atoi(argv[6]) : 16,
              dup = argc > 7 ? atoi(argv[7]) : 1, group = argc > 8 ? atoi(argv[8]) : 1;
    const uint64_t span = (uint64_t)(spanGiB * 1024.0 * 1024.0 * 1024.0) & ~(uint64_t)127;
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    uint8_t* base = nullptr; CHECK(hipMalloc(&base, span));
    unsigned long long* sink = nullptr; CHECK(hipMalloc(&sink, 8)); CHECK(hipMemset(sink, 0, 8));
    // touch the span once (a fill kernel of the runtime): page tables and HBM pages exist before the timed gathers
    CHECK(hipMemset(base, 0x5a, span)); CHECK(hipDeviceSynchronize());
    const uint64_t sliceBytes = sliceMiB > 0 ? ((uint64_t)(sliceMiB * 1024.0 * 1024.0) & ~(uint64_t)127) : span;
    const uint64_t nSlices = (span + sliceBytes - 1) / sliceBytes;
    const int blocks = cus * wps;                                              // 256 lanes = 4 waves = one per SIMD; wps blocks per CU
    const uint64_t lanes = (uint64_t)blocks * 256;
    uint64_t perLane = (gathers * dup / nSlices + lanes - 1) / lanes; perLane = ((perLane + U - 1) / U) * U; if (perLane == 0) perLane = U;
    std::vector<uint64_t> order(nSlices); for (uint64_t i = 0; i < nSlices; i++) order[i] = i;
    uint64_t rs = 12345; for (uint64_t i = nSlices; i > 1; i--) { rs = rs * 6364136223846793005ULL + 1; std::swap(order[i - 1], order[(rs >> 33) % i]); }
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    double best = 1e30; uint64_t done = 0;
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(e0, 0)); done = 0;
        for (uint64_t q = 0; q < nSlices; q++) {
            const uint64_t off = order[q] * sliceBytes, len = (off + sliceBytes <= span) ? sliceBytes : span - off;
            if (bytes == 16) launch<16>(U, blocks, base + off, len / 128, perLane, 77 + rep * 1000003 + q, sink, dup, group);
            else if (bytes == 8) launch<8>(U, blocks, base + off, len / 128, perLane, 77 + rep * 1000003 + q, sink, dup, group);
            else launch<4>(U, blocks, base + off, len / 128, perLane, 77 + rep * 1000003 + q, sink, dup, group);
            done += perLane * lanes / dup;
        }
        CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("{\"span_GiB\": %.1f, \"slice_MiB\": %.1f, \"slices\": %llu, \"waves_per_simd\": %d, \"loads_in_flight_per_lane\": %d, \"bytes_per_lane\": %d, "
           "\"dup\": %d, \"group\": %d, \"gathers\": %llu, \"ms\": %.3f, \"G_lines_per_s\": %.2f, \"TB_per_s_of_128B_lines\": %.3f, \"launches\": %llu}\n",
           spanGiB, sliceBytes / 1048576.0, (unsigned long long)nSlices, wps, U, bytes, dup, group, (unsigned long long)done, best, done / best / 1e6, done * 128.0 / best / 1e9,
           (unsigned long long)nSlices);
    return 0;
}
