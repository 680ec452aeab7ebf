// bbduk_big_tiles.hip -- the tiled and the long-read fallbacks of ktrim=n, ktrim=rl, ksplit, findbestmatch and k > 31 (units beyond a wave's planes)
// instantiated with the exact lookups of a big-layout map (BIGT = 1: lookup4<.., BIG> / lookup<.., BIG> -> big_find: a full-length key in its
// minimizer line, a short one in the secondary map).  They serve the maps that have no cache-resident twin (beyond 2^25 keys); with a twin the plain
// instantiations of bbduk_hip.hip run over it, which is the faster of the two (DESIGN "The twin").  A translation unit of its own for the parallel build.
#include "bbduk_device.inc"
#include "bbduk_kernels.h"

kmask_tile_t bbduk_pick_kmask_big_tile() { return bbduk_kmask_kernel<1>; }
kmask_long_t bbduk_pick_kmask_big_long() { return bbduk_kmask_long_kernel<1>; }
tips_tile_t bbduk_pick_tips_big_tile() { return bbduk_ktrimtips_kernel<1>; }
tips_tile_t bbduk_pick_tips_big_long() { return bbduk_long_tips_kernel<1>; }
kscan_tile_t bbduk_pick_kscan_big_tile(int red) { return red == RED_SPLIT ? bbduk_kscan_kernel<RED_SPLIT, 1> : (red == RED_BEST ? bbduk_kscan_kernel<RED_BEST, 1> : bbduk_kscan_kernel<RED_BIG, 1>); }
kscan_tile_t bbduk_pick_kscan_big_long(int red) { return red == RED_SPLIT ? bbduk_kscan_long_kernel<RED_SPLIT, 1> : (red == RED_BEST ? bbduk_kscan_long_kernel<RED_BEST, 1> : bbduk_kscan_long_kernel<RED_BIG, 1>); }
