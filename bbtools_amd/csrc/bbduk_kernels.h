// bbduk_kernels.h -- run-time -> template dispatch of the batch kernels.  The kernel families are instantiated in translation units of
// their own (bbduk_k_*.hip) so that they compile in parallel; the host code (bbduk_hip.hip) only sees these pick functions.
#ifndef BBDUK_KERNELS_H
#define BBDUK_KERNELS_H
#include "bbduk_internal.h"

typedef void (*batch_kernel_t)(const KParams, const uint8_t*, const int64_t*, const int64_t, const int64_t, const int,
                               int32_t*, int32_t*, uint8_t*, int64_t*, const int*);
// wave / wavePacked: bbduk_wave_kernel for the ASCII / the packed boundary format; tile: bbduk_batch_kernel (units beyond a wave's planes).
// The first-hit scans of ktrim=r / kfilter run in bbduk_stream_kernel instead (bbduk_pick_stream) wherever it takes the handle.
struct KernelPair { batch_kernel_t wave, wavePacked, tile; };

KernelPair bbduk_pick_ktrim_r(bool general, bool useShort, bool forbidN);       // bbduk_k_ktrimr.hip
KernelPair bbduk_pick_ktrim_l(bool general, bool useShort, bool forbidN);       // bbduk_k_ktriml.hip
KernelPair bbduk_pick_kfilter(bool general, bool forbidN);                      // bbduk_k_kfilter.hip
KernelPair bbduk_pick_kfilter_big(bool forbidN);                                // bbduk_k_kfilter.hip (HBM-resident layout)
batch_kernel_t bbduk_pick_ktrim_r_big_tile();                                   // bbduk_k_ktrimr.hip: units beyond a wave's planes against a big-layout ktrim=r map
// the other modes of bbduk_wave_kernel: BBDUK_MODE_FBM / _KBIG (bbduk_k_modes_a.hip), _KSPLIT / _KTRIM_TIPS / _KMASK (bbduk_k_modes_b.hip)
batch_kernel_t bbduk_pick_mode_wave(int mode, bool general, bool packed, bool forbidN);
// the stream kernels (bbduk_stream.hip): mode = BBDUK_MODE_KTRIM_R | BBDUK_MODE_KFILTER
batch_kernel_t bbduk_pick_stream(int mode, bool useShort, bool forbidN, bool packed, bool general);
// the every-hit scans on the stream (bbduk_stream_every*.hip): mode = BBDUK_MODE_KTRIM_L | _KFILTER (with a threshold) | _KSPLIT | _KMASK
batch_kernel_t bbduk_pick_stream_every(int mode, bool useShort, bool forbidN, bool general);
batch_kernel_t bbduk_pick_stream_tips(bool packed);                            // ktrim=rl, no forbidNs, specialised family
batch_kernel_t bbduk_pick_stream_seed(bool forbidN, bool packed);              // the stream scan over a seed-layout map (bbduk_seed.inc)
batch_kernel_t bbduk_pick_ktrim_l_big_tile();
batch_kernel_t bbduk_pick_bigs_every(int mode, bool useShort);                                       // bbduk_bigs_every.hip: ... and its every-hit form (ktrim=l, kfilter with maxbadkmers > 0)
batch_kernel_t bbduk_pick_bigs(int mode, bool useShort, bool packed);
batch_kernel_t bbduk_pick_bigs_kbig(bool general);                                                // bbduk_bigs_kbig.hip: k > 31 over a big-layout map
batch_kernel_t bbduk_pick_bigs_general(int mode);                                                  // bbduk_bigs_general*.hip: the GENERAL family (restrict*, skipr*, qskip, speed, rcomp=f) over a big-layout map, every mode (mode = BBDUK_MODE_* | _FBM)
// bbduk_big_tiles.hip: the tiled / long-read fallbacks of the secondary operators with a big-layout map's exact lookups (maps without a twin)
typedef void (*kmask_tile_t)(const KParams, const uint8_t*, const int64_t*, const int64_t, const int64_t, const int, int32_t*, int32_t*, uint8_t*, uint32_t*, int64_t*, int*);
typedef void (*kmask_long_t)(const KParams, const uint8_t*, const int64_t*, const int64_t, const int64_t, const int, int32_t*, int32_t*, uint32_t*, int64_t*, const int*);
typedef void (*tips_tile_t)(const KParams, const uint8_t*, const int64_t*, const int64_t, const int64_t, const int, int32_t*, int32_t*, int32_t*, uint8_t*, int64_t*, const int*);
typedef void (*kscan_tile_t)(const KParams, const uint8_t*, const int64_t*, const int64_t, const int64_t, const int, int32_t*, int32_t*, uint8_t*, int32_t*, int32_t*, int64_t*, const int*);
kmask_tile_t bbduk_pick_kmask_big_tile();
kmask_long_t bbduk_pick_kmask_big_long();
tips_tile_t bbduk_pick_tips_big_tile();
tips_tile_t bbduk_pick_tips_big_long();
kscan_tile_t bbduk_pick_kscan_big_tile(int red);       // red = RED_SPLIT | RED_BEST (k > 31 keeps the cache-resident layout)
kscan_tile_t bbduk_pick_kscan_big_long(int red);                                 // bbduk_bigs.hip: the stream scan over a big-layout map with the 32-bit line function

#ifdef BBDUK_DEVICE_INC            /* translation units that hold kernel templates */
template <int MODE, bool SHORT, bool FORBIDN, bool GENERAL>
static KernelPair kpair() {
    // the specialised wave kernels exist once per input format; the general one and the tile fallback decide per launch
    KernelPair kp = GENERAL ? KernelPair{bbduk_wave_kernel<MODE, SHORT, FORBIDN, GENERAL, 2>, bbduk_wave_kernel<MODE, SHORT, FORBIDN, GENERAL, 2>,
                                         bbduk_batch_kernel<MODE, SHORT, FORBIDN, GENERAL>}
                            : KernelPair{bbduk_wave_kernel<MODE, SHORT, FORBIDN, GENERAL, 0>, bbduk_wave_kernel<MODE, SHORT, FORBIDN, GENERAL, 1>,
                                         bbduk_batch_kernel<MODE, SHORT, FORBIDN, GENERAL>};
    return kp;
}
template <int MODE>
static KernelPair pick_kernel_mode(bool general, bool useShort, bool forbidN) {
    if (general) return kpair<MODE, true, true, true>();
    if (MODE == BBDUK_MODE_KFILTER) return forbidN ? kpair<MODE, false, true, false>() : kpair<MODE, false, false, false>();
    if (useShort) return forbidN ? kpair<MODE, true, true, false>() : kpair<MODE, true, false, false>();
    return forbidN ? kpair<MODE, false, true, false>() : kpair<MODE, false, false, false>();
}
template <int MODE, bool SHORT>
static batch_kernel_t pick_mode_wave(bool general, bool packed, bool forbidN) {
    if (general) return bbduk_wave_kernel<MODE, SHORT, true, true, 2>;
    if (packed) return forbidN ? bbduk_wave_kernel<MODE, SHORT, true, false, 1> : bbduk_wave_kernel<MODE, SHORT, false, false, 1>;
    return forbidN ? bbduk_wave_kernel<MODE, SHORT, true, false, 0> : bbduk_wave_kernel<MODE, SHORT, false, false, 0>;
}
#endif
#endif
