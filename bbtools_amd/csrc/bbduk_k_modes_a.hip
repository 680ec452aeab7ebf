// bbduk_k_modes_a.hip -- bbduk_wave_kernel<FBM> (findbestmatch / rename) and <KBIG> (k > 31).
#include "bbduk_device.inc"
#include "bbduk_kernels.h"
batch_kernel_t bbduk_pick_mode_wave_a(int mode, bool general, bool packed, bool forbidN) {
    if (mode == BBDUK_MODE_FBM) return pick_mode_wave<BBDUK_MODE_FBM, false>(general, packed, forbidN);
    return pick_mode_wave<BBDUK_MODE_KBIG, false>(general, packed, forbidN);
}
