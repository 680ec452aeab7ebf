// bbduk_stream.hip -- bbduk_stream_kernel: the length-agnostic first-hit scan of the specialised family (ktrim=r; kfilter with
// maxbadkmers = 0), i.e. wave_body with the stream scan of bbduk_stream_scan.inc in place of the pair scan (SHAPE = 3).  DESIGN 4.1.
#include "bbduk_device.inc"
#include "bbduk_kernels.h"

template <int MODE, bool SHORT, bool FORBIDN, int FMT, bool SEED = false>
__global__ __launch_bounds__(BLOCK_THREADS)
void bbduk_stream_kernel(const KParams P, const uint8_t* __restrict__ bases, const int64_t* __restrict__ offsets,
                         const int64_t n, const int64_t totalBases, const int paired,
                         int32_t* __restrict__ outA, int32_t* __restrict__ outId, uint8_t* __restrict__ outFlags,
                         int64_t* __restrict__ counters, const int* __restrict__ slowFlag) {
    static_assert(MODE == BBDUK_MODE_KTRIM_R || MODE == BBDUK_MODE_KFILTER, "first-hit scans only");
    wave_body<MODE, SHORT, FORBIDN, false, FMT, SEED, 3>(P, bases, offsets, n, totalBases, paired, outA, outId, outFlags, counters, slowFlag);
}

// the GENERAL family (restrictleft/right, skipr1/2, qskip, speed, rcomp=f, k < 16, mink with a middle mask): one instantiation per mode,
// input format decided per launch
template <int MODE>
__global__ __launch_bounds__(BLOCK_THREADS)
void bbduk_stream_general_kernel(const KParams P, const uint8_t* __restrict__ bases, const int64_t* __restrict__ offsets,
                                 const int64_t n, const int64_t totalBases, const int paired,
                                 int32_t* __restrict__ outA, int32_t* __restrict__ outId, uint8_t* __restrict__ outFlags,
                                 int64_t* __restrict__ counters, const int* __restrict__ slowFlag) {
    wave_body<MODE, true, true, true, 2, false, 3>(P, bases, offsets, n, totalBases, paired, outA, outId, outFlags, counters, slowFlag);
}

// ktrim=rl without forbidNs: both passes off one stream scan (wave_body: TIPS && STREAM)
template <int FMT>
__global__ __launch_bounds__(BLOCK_THREADS)
void bbduk_stream_tips_kernel(const KParams P, const uint8_t* __restrict__ bases, const int64_t* __restrict__ offsets,
                              const int64_t n, const int64_t totalBases, const int paired,
                              int32_t* __restrict__ outA, int32_t* __restrict__ outId, uint8_t* __restrict__ outFlags,
                              int64_t* __restrict__ counters, const int* __restrict__ slowFlag) {
    wave_body<BBDUK_MODE_KTRIM_TIPS, true, false, false, FMT, false, 3>(P, bases, offsets, n, totalBases, paired, outA, outId, outFlags, counters, slowFlag);
}
batch_kernel_t bbduk_pick_stream_tips(bool packed) { return packed ? bbduk_stream_tips_kernel<1> : bbduk_stream_tips_kernel<0>; }

template <int MODE, bool SHORT>
static batch_kernel_t pick(bool forbidN, bool packed) {
    if (packed) return forbidN ? bbduk_stream_kernel<MODE, SHORT, true, 1> : bbduk_stream_kernel<MODE, SHORT, false, 1>;
    return forbidN ? bbduk_stream_kernel<MODE, SHORT, true, 0> : bbduk_stream_kernel<MODE, SHORT, false, 0>;
}
batch_kernel_t bbduk_pick_stream(int mode, bool useShort, bool forbidN, bool packed, bool general) {
    if (general) return mode == BBDUK_MODE_KFILTER ? bbduk_stream_general_kernel<BBDUK_MODE_KFILTER> : bbduk_stream_general_kernel<BBDUK_MODE_KTRIM_R>;
    if (mode == BBDUK_MODE_KFILTER) return pick<BBDUK_MODE_KFILTER, false>(forbidN, packed);
    return useShort ? pick<BBDUK_MODE_KTRIM_R, true>(forbidN, packed) : pick<BBDUK_MODE_KTRIM_R, false>(forbidN, packed);
}
// the seed layout (bbduk_seed.inc): large hdist=1 kfilter maps
batch_kernel_t bbduk_pick_stream_seed(bool forbidN, bool packed) {
    if (packed) return forbidN ? bbduk_stream_kernel<BBDUK_MODE_KFILTER, false, true, 1, true> : bbduk_stream_kernel<BBDUK_MODE_KFILTER, false, false, 1, true>;
    return forbidN ? bbduk_stream_kernel<BBDUK_MODE_KFILTER, false, true, 0, true> : bbduk_stream_kernel<BBDUK_MODE_KFILTER, false, false, 0, true>;
}
