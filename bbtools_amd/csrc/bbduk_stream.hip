// bbduk_stream.hip -- the length-agnostic first-hit scan (bbduk_stream_kernel), DESIGN 4.1.
#include "bbduk_device.inc"
#include "bbduk_kernels.h"
batch_kernel_t bbduk_pick_stream(int mode, bool useShort, bool forbidN, bool packed) { return nullptr; }
