// bbduk_k_ktriml.hip -- the ktrim=l instantiations of bbduk_wave_kernel / bbduk_batch_kernel.
#include "bbduk_device.inc"
#include "bbduk_kernels.h"
KernelPair bbduk_pick_ktrim_l(bool general, bool useShort, bool forbidN) { return pick_kernel_mode<BBDUK_MODE_KTRIM_L>(general, useShort, forbidN); }
// ktrim=l against a big-layout map: the wave kernel is bbduk_bigs_every_kernel; units beyond a wave's planes take the tiled kernel's exact lookups
batch_kernel_t bbduk_pick_ktrim_l_big_tile() { return bbduk_batch_kernel<BBDUK_MODE_KTRIM_L, true, true, true, true>; }
