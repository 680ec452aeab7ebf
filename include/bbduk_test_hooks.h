/*
 * bbduk_test_hooks.h -- controls for tests and experiments; NOT part of the drop-in boundary (include/bbduk_gpu.h) and never
 * called by a product caller (bbduk_cli, the JNI shim, bench.py's timed path).  They replace the environment variables an
 * earlier build read on its hot host path: the library itself reads no environment variable.
 */
#ifndef BBDUK_TEST_HOOKS_H
#define BBDUK_TEST_HOOKS_H
#include "bbduk_gpu.h"
#ifdef __cplusplus
extern "C" {
#endif
#define BBDUK_HOOK_FORCE_TILE   1   /* value != 0: every batch of this handle (any operator) takes the tile-synchronous fallback kernels */
#define BBDUK_HOOK_BUCKET_BITS  2   /* before finalize / device build: log2 of the map's bucket count (4..32; 0 = sized by key count) */
#define BBDUK_HOOK_LDS_BITS     3   /* before finalize: log2 bits of the LDS presence filter (0 = no filter, -1 = sized by key count) */
#define BBDUK_HOOK_TIMING_MASK  4   /* -DBBDUK_TIMING_SWITCHES builds only: bit n deletes stage n of the scan (results become wrong) */
#define BBDUK_HOOK_PAIR_SCAN    6   /* value != 0: the first-hit scans run bbduk_wave_kernel's pair scan instead of bbduk_stream_kernel: A/B runs,
                                       and the tests that keep the pair scan's candidate form covered (tables beyond 2^28 buckets use it) */
#define BBDUK_HOOK_SEED_LAYOUT  7   /* before a device build: value != 0 asks for the seed layout (parents under their halves) at any size, where it is
                                       served (kfilter hdist=1, see bbduk_seed.inc) */
#define BBDUK_HOOK_BIG_LAYOUT   5   /* before finalize / device build: value > 0 forces the HBM-resident map layout at any size, value < 0 keeps it
                                       from being chosen (A/B runs of the cache-resident layout at sizes that would take it) */
#define BBDUK_HOOK_BIG_LOAD     8   /* before finalize / device build: keys per 100 slots the big layout's lines are sized for (0 = the default); value 2 of
                                       BBDUK_HOOK_BIG_LAYOUT forces round 2's 52-bit line function and its pair scan, value 3 the wide values that maps beyond 2^31 keys
                                       take (gap_v52, same scan as the 32-bit values) */
int  bbduk_test_hook(bbduk_handle* h, int32_t which, int64_t value);
/* the same controls on the map behind a Seal handle (include/seal_gpu.h), before seal_finalize */
struct seal_handle;
int  seal_test_hook(struct seal_handle* h, int32_t which, int64_t value);
/* big layout: keys that found their pair of words AND the sibling pair of the line's other half full and live in the secondary map (0 for the cache-resident layout) */
int64_t bbduk_table_spilled(const bbduk_handle* h);
/* big layout: out33[c] = number of 64-byte HALF lines (32 slots; a line is 128 bytes) that hold c keys */
int  bbduk_table_line_histogram(bbduk_handle* h, int64_t* out33);
/* which layout the finalized map took: 0 cache-resident, 1 big (minimizer or plain lines), 2 seed (parents under their halves); + 4: a cache-resident twin beside it; + 8: the query-side expansion (qhdist = 1) is tabulated (the map the kernels look up is the expansion); -1: not finalized */
int  bbduk_table_layout(const bbduk_handle* h);
#ifdef __cplusplus
}
#endif
#endif
