// bbduk_k_ktriml.hip -- the ktrim=l instantiations of bbduk_wave_kernel / bbduk_batch_kernel.
#include "bbduk_device.inc"
#include "bbduk_kernels.h"
KernelPair bbduk_pick_ktrim_l(bool general, bool useShort, bool forbidN) { return pick_kernel_mode<BBDUK_MODE_KTRIM_L>(general, useShort, forbidN); }
