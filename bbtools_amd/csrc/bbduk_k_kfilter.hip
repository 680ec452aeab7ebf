// bbduk_k_kfilter.hip -- the kfilter instantiations of bbduk_wave_kernel / bbduk_batch_kernel, including the
// BIG ones of the HBM-resident layout (BASELINE configs[3]).
#include "bbduk_device.inc"
#include "bbduk_kernels.h"
KernelPair bbduk_pick_kfilter(bool general, bool forbidN) { return pick_kernel_mode<BBDUK_MODE_KFILTER>(general, false, forbidN); }
KernelPair bbduk_pick_kfilter_big(bool forbidN) {
    // HBM-resident layout: chosen at build time only for the plain kfilter configurations (big_layout_eligible: BASELINE
    // configs[3]), whose first-hit scan has the minimizer-sharing candidate form; the exact scans (maxbadkmers > 0, impostors) and
    // the tile / long-read fallbacks are the BIG instantiations of the same functions
    const batch_kernel_t tile = bbduk_batch_kernel<BBDUK_MODE_KFILTER, true, true, true, true>;
    if (forbidN) return KernelPair{bbduk_wave_kernel<BBDUK_MODE_KFILTER, false, true, false, 0, true>, bbduk_wave_kernel<BBDUK_MODE_KFILTER, false, true, false, 1, true>, tile};
    return KernelPair{bbduk_wave_kernel<BBDUK_MODE_KFILTER, false, false, false, 0, true>, bbduk_wave_kernel<BBDUK_MODE_KFILTER, false, false, false, 1, true>, tile};
}
