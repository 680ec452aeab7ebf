// bbduk_k_ktrimr.hip -- the ktrim=r instantiations of bbduk_wave_kernel / bbduk_batch_kernel (one translation
// unit per kernel family: they compile in parallel).
#include "bbduk_device.inc"
#include "bbduk_kernels.h"
KernelPair bbduk_pick_ktrim_r(bool general, bool useShort, bool forbidN) { return pick_kernel_mode<BBDUK_MODE_KTRIM_R>(general, useShort, forbidN); }
