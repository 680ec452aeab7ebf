// bbduk_k_ktrimr.hip -- the ktrim=r instantiations of bbduk_wave_kernel / bbduk_batch_kernel (one translation
// unit per kernel family: they compile in parallel).
#include "bbduk_device.inc"
#include "bbduk_kernels.h"
KernelPair bbduk_pick_ktrim_r(bool general, bool useShort, bool forbidN) { return pick_kernel_mode<BBDUK_MODE_KTRIM_R>(general, useShort, forbidN); }
// ktrim=r against a big-layout map: the wave kernel is bbduk_bigs_kernel; units beyond a wave's planes take the tiled kernel's exact lookups
batch_kernel_t bbduk_pick_ktrim_r_big_tile() { return bbduk_batch_kernel<BBDUK_MODE_KTRIM_R, true, true, true, true>; }
