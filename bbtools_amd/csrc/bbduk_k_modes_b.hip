// bbduk_k_modes_b.hip -- bbduk_wave_kernel<KSPLIT>, <KTRIM_TIPS> (ktrim=rl) and <KMASK> (ktrim=n).
#include "bbduk_device.inc"
#include "bbduk_kernels.h"
batch_kernel_t bbduk_pick_mode_wave_a(int mode, bool general, bool packed, bool forbidN);
batch_kernel_t bbduk_pick_mode_wave(int mode, bool general, bool packed, bool forbidN) {
    if (mode == BBDUK_MODE_FBM || mode == BBDUK_MODE_KBIG) return bbduk_pick_mode_wave_a(mode, general, packed, forbidN);
    if (mode == BBDUK_MODE_KSPLIT)                                  // (ksplit has no packed operator on the wave kernel)
        return general ? bbduk_wave_kernel<BBDUK_MODE_KSPLIT, true, true, true, 2>
                       : (forbidN ? bbduk_wave_kernel<BBDUK_MODE_KSPLIT, true, true, false, 0> : bbduk_wave_kernel<BBDUK_MODE_KSPLIT, true, false, false, 0>);
    if (mode == BBDUK_MODE_KTRIM_TIPS) return pick_mode_wave<BBDUK_MODE_KTRIM_TIPS, true>(general, packed, forbidN);
    return pick_mode_wave<BBDUK_MODE_KMASK, true>(general, packed, forbidN);
}
