// bbduk_hip.hip -- MI355X (gfx950) kernels + the C ABI of include/bbduk_gpu.h.
//
// Hot path: per-read 2-bit k-mer encode with reverse-complement
// canonicalisation, optional query-side Hamming expansion, lookup into a device image of the reference
// k-mer map, first-hit / hit-count reduction with wave ballots, trim / filter decision, pair logic and
// counters.  Integer work only (no MFMA).  With a cache-resident map the kernel is bound by VALU issue
// (~130 VALU wave-instructions per read, every one ~4.8 SIMD cycles on gfx950), so the code
// below is written to keep instructions -- vector AND scalar -- out of the per-position path.
//
// Design (not a translation of the Java loops):
//   * reads are contiguous in the concatenated `bases` buffer, so a wave stages its reads with 16-byte
//     coalesced loads and converts them on the fly to three bit-planes in its own slice of LDS (2-bit
//     forward codes in *reversed* base order, 2-bit complement codes, 1-bit undefined mask);
//   * the scan is position-parallel: a lane serves two adjacent k-mer end positions and cuts both k-mers
//     and their reverse complements out of the planes with two funnel shifts per plane (closed form,
//     SURVEY A.12) instead of rolling them along the read;
//   * lookups cascade LDS presence bit -> 8-byte fingerprint gather -> key record, and the hot block is
//     predicate-free: compares write wave masks, the scalar unit combines them, one branch per 256 positions;
//   * everything per read (bookkeeping, trim/filter decision, outputs, counters) is data-parallel, one lane
//     per read of a 62-read mini-tile.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <string>
#include <vector>
#include <algorithm>
#include <mutex>
#include <type_traits>
#include "../../include/bbduk_gpu.h"
#include "../../include/bbduk_test_hooks.h"
#include "bbduk_internal.h"
#include "synth.h"

static_assert(sizeof(bbduk_params) == 136, "bbduk_params layout is part of the ABI");
static_assert(sizeof(bbduk_synth_params) == 80, "bbduk_synth_params layout is part of the ABI");

#define BBDUK_MAIN_TU
#include <map>
#include "bbduk_device.inc"
#include "bbduk_kernels.h"

// Pre-pass: does every unit (mate pair, or single read) fit a wave's planes?  One thread per unit.
// wmax / hmax: longest unit the first / the second kernel of the operator accepts (flag bit 0 / bit 1 otherwise).
__global__ void bbduk_span_kernel(const int64_t* __restrict__ offsets, const int64_t n, const int paired, int* __restrict__ slowFlag,
                                  const int64_t wmax = WUNIT_MAX, const int64_t hmax = CAP_BASES - 64) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int step = paired ? 2 : 1;
    const int64_t units = n / step;
    bool bad = false, huge = false;
    for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < units; u += stride) {
        const int64_t len = offsets[u * step + step] - offsets[u * step];
        bad |= len > wmax;
        huge |= len > hmax;                                       // not even the tile kernel's planes hold this unit: bbduk_long_kernel
    }
    if (__ballot(bad) && (threadIdx.x & 63) == 0) atomicOr(slowFlag, 1);
    if (__ballot(huge) && (threadIdx.x & 63) == 0) atomicOr(slowFlag, 2);
}

// runtime -> template dispatch: the kernel families live in translation units of their own (bbduk_kernels.h)
// the specialised kernels assume k >= 16 (BBDuk's usual 23-31) and what BBDukParser guarantees (mink turns
// maskMiddle off, :295-301); anything else takes the general kernel
// The largest dynamic LDS size a kernel may be launched with is a per-function (and device) attribute.  Handles differ in what they need
// (the filter's size), and several host threads launch at once (one per handle: bbduk_cli devices=, the JVM's worker threads): the
// attribute is only ever RAISED, under one process-wide lock, so a launch never meets a smaller limit than it asked for -- and the
// per-launch calls of rounds 1-2 (three per batch) are gone.
static hipError_t ensure_dyn_lds(const void* fn, size_t bytes) {
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, size_t> have;
    if (bytes == 0) return hipSuccess;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lg(mu);
    size_t& cur = have[std::make_pair(dev, fn)];
    if (bytes <= cur) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) cur = bytes;
    return e;
}
static bool kparams_general(const KParams& K) {
    return K.qhdist > 0 || K.qhdist2 > 0 || K.restrictLeft > 0 || K.restrictRight > 0 || K.skipR1 || K.skipR2 || !K.rcomp ||
                         (K.useShort && K.middleMask != ~0ULL) || K.k < 16 || K.qskip > 1 || K.speed > 0 || K.mkf != 0.f || K.mcf > 0.f;
}
static KernelPair pick_kernel(const KParams& K) {
    const bool general = kparams_general(K);
    if (K.big || K.seed) return bbduk_pick_kfilter_big(K.forbidNs != 0);        // HBM-resident layout (BASELINE configs[3]): see bbduk_k_kfilter.hip
    if (K.mode == BBDUK_MODE_KFILTER) return bbduk_pick_kfilter(general, K.forbidNs != 0);
    if (K.mode == BBDUK_MODE_KTRIM_L) return bbduk_pick_ktrim_l(general, K.useShort != 0, K.forbidNs != 0);
    return bbduk_pick_ktrim_r(general, K.useShort != 0, K.forbidNs != 0);
}

// ASCII bases -> the packed boundary format (one thread per 16-base word)
__global__ void bbduk_pack_kernel(const uint8_t* __restrict__ bases, const int64_t total, uint32_t* __restrict__ codes, uint16_t* __restrict__ undef16) {
    const int64_t words = (total + 15) >> 4;
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < words; w += (int64_t)gridDim.x * blockDim.x) {
        uint32_t r, comp, valid;
        encode_chunk(bases, 16 * w, total, r, comp, valid);
        const uint32_t x = __brev(r);                                 // undo the plane's symbol reversal: codes in base order
        codes[w] = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
        undef16[w] = (uint16_t)(~valid & 0xFFFFu);
    }
}

__global__ void bbduk_lookup_kernel(const KParams P, const int64_t* keys, int64_t n, int32_t* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const uint64_t key = (uint64_t)keys[i], v = strip_len(key);
        out[i] = (keys[i] < 0) ? -1 : P.qx ? qx_find_key(P, key) : (P.seed ? seed_find_key(P, key) : (P.big ? big_find(P, key, mix_a(v), mix_b(v)) : table_get(P, key)));
    }
}

__global__ void bbduk_synth_kernel(const bb_synth_dev sp, const int64_t firstPair, const int64_t nPairs,
                                   uint8_t* __restrict__ bases, int64_t* __restrict__ offsets) {
    const int64_t per = 2LL * sp.read_len;
    const int64_t total = nPairs * per;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    // grid-stride: one base per thread-iteration (a launch of > 2^31 threads is not portable)
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total || g <= 2 * nPairs; g += stride) {
        if (g <= 2 * nPairs) offsets[g] = g * sp.read_len;
        if (g >= total) continue;
        const int64_t p = g / per;
        const int32_t rem = (int32_t)(g - p * per);
        const int32_t mate = rem >= sp.read_len ? 1 : 0;
        const int32_t j = rem - mate * sp.read_len;
        const bb_pair_hdr h = bb_synth_pair_header(sp, (uint64_t)(firstPair + p));
        bases[g] = bb_synth_read_base(sp, (uint64_t)(firstPair + p), h, mate, j);
    }
}

// --------------------------------------------------------------------------------------------------
// host side of the C ABI


extern "C" int bbduk_abi_version(void) { return BBDUK_ABI_VERSION; }
extern "C" const char* bbduk_last_error(const bbduk_handle* h) {
    if (!h) return "null handle";
    static thread_local std::string copy;                         // (the caller's own copy: another thread may fail meanwhile)
    { std::lock_guard<std::mutex> lg(const_cast<bbduk_handle*>(h)->errMu); copy = h->err; }
    return copy.c_str();
}

extern "C" int bbduk_create(const bbduk_params* p, bbduk_handle** out) {
    if (!p || !out) return BBDUK_ERR_ARG;
    *out = nullptr;
    if (p->abi_version != BBDUK_ABI_VERSION) return BBDUK_ERR_ARG;
    if (p->k < 1 || p->k > 31) return BBDUK_ERR_ARG;
    if (p->mode < BBDUK_MODE_KFILTER || p->mode > BBDUK_MODE_KSPLIT) return BBDUK_ERR_ARG;
    if (p->qhdist < 0 || p->qhdist > 3 || p->qhdist2 < 0 || p->qhdist2 > 3) return BBDUK_ERR_ARG;
    if (p->numScaffolds < 1 || p->maxBadKmers < 0) return BBDUK_ERR_ARG;
    const bool useShort = p->mink > 0 && p->mink < p->k;
    if (useShort && p->mode == BBDUK_MODE_KFILTER) return BBDUK_ERR_ARG;      // BBDukParser.java:301
    if (useShort && p->middleMask != -1) return BBDUK_ERR_ARG;                // BBDukProcessorS.java:2035 assert
    if (p->minlen != p->k - 1) return BBDUK_ERR_ARG;
    if (p->speed < 0 || p->speed > 16 || p->qSkip < 0) return BBDUK_ERR_ARG;    // BBDukParser.java:568
    if (p->kmaskFullyCovered && p->mode != BBDUK_MODE_KMASK) return BBDUK_ERR_ARG;
    if (p->reserved0 != 0) return BBDUK_ERR_ARG;
    const bool big = p->kbig > p->k;
    if (big) {                                                                  // BBDukParser.java:164, 207-243, 299
        if (p->k != 31 || p->kbig > BBDUK_MAX_READ_LEN) return BBDUK_ERR_ARG;
        if (p->mode != BBDUK_MODE_KFILTER || p->speed > 0 || p->qSkip > 1) return BBDUK_ERR_ARG;    // the parser reduces kbig to k there
        if (p->middleMask != -1 || p->minlen2 != p->k) return BBDUK_ERR_ARG;    // maskMiddle is disabled before minlen2 is derived
        if (p->findBestMatch) return BBDUK_ERR_ARG;                             // mcf: countCoveredBases never looks at kbig (:1038-1049, 1602-1651)
    }
    if (p->findBestMatch) {
        if (p->mode != BBDUK_MODE_KFILTER || p->minCoveredFraction > 0.f) return BBDUK_ERR_ARG;
        // with found <= maxBadKmers the reference leaves findBestMatch's per-thread countArray dirty (:1694 is skipped), so
        // its answers depend on which reads the thread saw before: only the history-free case is served
        if (p->maxBadKmers != 0 || p->minKmerFraction != 0.f) return BBDUK_ERR_ARG;
    }
    if (p->mode == BBDUK_MODE_KSPLIT && p->trimPad > 0) return BBDUK_ERR_ARG;   // rightmost may pass the read end: Read.subRead throws there
    if (!(p->minKmerFraction >= 0.f && p->minKmerFraction <= 1.f) || !(p->minCoveredFraction >= 0.f && p->minCoveredFraction <= 1.f)) return BBDUK_ERR_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return BBDUK_ERR_DEVICE;
    if (p->device < 0 || p->device >= ndev) return BBDUK_ERR_ARG;
    bbduk_handle* h = new (std::nothrow) bbduk_handle();
    if (!h) return BBDUK_ERR_NOMEM;
    h->p = *p;
    if (hipSetDevice(p->device) != hipSuccess) { delete h; return BBDUK_ERR_DEVICE; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, p->device) == hipSuccess) h->numCU = prop.multiProcessorCount;
    if (hipStreamCreate(&h->stream) != hipSuccess) { delete h; return BBDUK_ERR_DEVICE; }
    const size_t nc = (size_t)(BBDUK_NCOUNTERS + 2 * p->numScaffolds);
    if (hipMalloc(&h->d_counters, nc * sizeof(int64_t)) != hipSuccess ||
        hipMemset(h->d_counters, 0, nc * sizeof(int64_t)) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess || hipMalloc(&h->d_slowFlag, 4 * bbduk_handle::EV_RING * sizeof(int)) != hipSuccess) { hipStreamDestroy(h->stream); delete h; return BBDUK_ERR_DEVICE; }
    *out = h;
    return BBDUK_OK;
}

static void build_release(bbduk_handle* h);
static int qx_rewrite(bbduk_handle* h);

// Test-only controls (include/bbduk_test_hooks.h): explicit calls on a handle instead of environment variables.
extern "C" int bbduk_test_hook(bbduk_handle* h, int32_t which, int64_t value) {
    if (!h) return BBDUK_ERR_ARG;
    std::lock_guard<std::mutex> g(h->mu);
    switch (which) {
    case BBDUK_HOOK_FORCE_TILE:  h->hookForceTile = value != 0; return BBDUK_OK;
    case BBDUK_HOOK_PAIR_SCAN:   h->hookPairScan = value != 0; return BBDUK_OK;
    case BBDUK_HOOK_BUCKET_BITS: if (h->finalized) return fail(h, BBDUK_ERR_STATE, "hook after finalize"); h->hookBucketBits = (int)value; return BBDUK_OK;
    case BBDUK_HOOK_LDS_BITS:    if (h->finalized) return fail(h, BBDUK_ERR_STATE, "hook after finalize"); h->hookLdsBits = (int)value; return BBDUK_OK;
    case BBDUK_HOOK_SEED_LAYOUT: if (h->finalized) return fail(h, BBDUK_ERR_STATE, "hook after finalize"); h->hookSeedLayout = value != 0; return BBDUK_OK;
    case BBDUK_HOOK_BIG_LOAD:    if (h->finalized) return fail(h, BBDUK_ERR_STATE, "hook after finalize"); h->hookBigLoad = (int)value; return BBDUK_OK;
    case BBDUK_HOOK_BIG_LAYOUT:  if (h->finalized) return fail(h, BBDUK_ERR_STATE, "hook after finalize"); h->hookBigLayout = value > 0; h->hookNoBigLayout = value < 0; h->hookBig52 = value == 2; h->hookBigWide = value == 3; return BBDUK_OK;
    case BBDUK_HOOK_TIMING_MASK:
#ifdef BBDUK_TIMING_SWITCHES
        h->hookDbg = (int)value; return BBDUK_OK;
#else
        return fail(h, BBDUK_ERR_ARG, "this build has no timing switches (-DBBDUK_TIMING_SWITCHES)");
#endif
    default: return fail(h, BBDUK_ERR_ARG, "unknown hook");
    }
}

extern "C" int bbduk_destroy(bbduk_handle* h) {
    if (!h) return BBDUK_ERR_ARG;
    bbduk_comm_destroy(h);
    hipSetDevice(h->p.device);
    build_release(h);
    hipFree(h->d_bigTags); hipFree(h->d_bigKeys); hipFree(h->d_bigIds);
    for (auto& q : h->slot) { hipFree(q.d_bases); hipFree(q.d_undef); hipFree(q.d_off); hipFree(q.d_a); hipFree(q.d_id); hipFree(q.d_fl); hipFree(q.d_status); if (q.stream) hipStreamDestroy(q.stream); if (q.copyStream) hipStreamDestroy(q.copyStream); for (auto& e : q.evPiece) if (e) hipEventDestroy(e); }
    hipFree(h->d_tags); hipFree(h->d_bkv);
    hipFree(h->d_ldsImage); hipFree(h->d_slowFlag); hipFree(h->d_tagsAlt); hipFree(h->d_bkvAlt); hipFree(h->d_ldsAlt); hipFree(h->d_tagsQx); hipFree(h->d_bkvQx);
    for (int q = 0; q < bbduk_handle::EV_RING; q++) { if (h->ev0[q]) hipEventDestroy(h->ev0[q]); if (h->ev1[q]) hipEventDestroy(h->ev1[q]); if (h->evDone[q]) hipEventDestroy(h->evDone[q]); }
    hipFree(h->d_counters);
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
    return BBDUK_OK;
}

extern "C" int bbduk_upload_pairs(bbduk_handle* h, const int64_t* keys, const int32_t* values, int64_t n) {
    if (!h || n < 0 || (n > 0 && (!keys || !values))) return fail(h, BBDUK_ERR_ARG, "upload_pairs: bad argument");
    std::lock_guard<std::mutex> g(h->mu);
    if (h->finalized) return fail(h, BBDUK_ERR_STATE, "table already finalized");
    // ids index the counter vector on the device (scaffoldReadCounts[id], scaffoldBaseCounts[id]): one outside
    // 1..numScaffolds-1 would write outside the caller's vector, one <= 0 would turn hits into misses
    for (int64_t i = 0; i < n; i++) {
        if (keys[i] < 0) return fail(h, BBDUK_ERR_ARG, "upload_pairs: negative key");
        if (values[i] < 1 || values[i] >= h->p.numScaffolds) return fail(h, BBDUK_ERR_ARG, "upload_pairs: scaffold id outside 1..numScaffolds-1");
    }
    for (int64_t i = 0; i < n; i++) { h->hkeys.push_back(keys[i]); h->hvals.push_back(values[i]); }
    return BBDUK_OK;
}

extern "C" int bbduk_upload_table_way(bbduk_handle* h, int32_t way, int32_t prime, const int64_t* keys, const int32_t* values,
                                      int64_t ncells, const int64_t* vkeys, const int32_t* vvals, int64_t nvictims) {
    (void)way; (void)prime;      // the device re-hashes into its own layout; geometry of the Java image is not needed
    if (!h || ncells < 0 || nvictims < 0 || (ncells > 0 && (!keys || !values)) || (nvictims > 0 && (!vkeys || !vvals)))
        return fail(h, BBDUK_ERR_ARG, "upload_table_way: bad argument");
    std::lock_guard<std::mutex> g(h->mu);
    if (h->finalized) return fail(h, BBDUK_ERR_STATE, "table already finalized");
    for (int64_t i = 0; i < ncells; i++) if (keys[i] >= 0 && (values[i] < 1 || values[i] >= h->p.numScaffolds))
        return fail(h, BBDUK_ERR_ARG, "upload_table_way: scaffold id outside 1..numScaffolds-1 (stale table, or numScaffolds too small?)");
    for (int64_t i = 0; i < nvictims; i++) if (vkeys[i] >= 0 && (vvals[i] < 1 || vvals[i] >= h->p.numScaffolds))
        return fail(h, BBDUK_ERR_ARG, "upload_table_way: victim scaffold id outside 1..numScaffolds-1");
    for (int64_t i = 0; i < ncells; i++) if (keys[i] >= 0) {     // NOT_PRESENT = -1 (kmer/AbstractKmerTable.java:807)
        h->hkeys.push_back(keys[i]); h->hvals.push_back(values[i]);
    }
    for (int64_t i = 0; i < nvictims; i++) if (vkeys[i] >= 0) { h->hkeys.push_back(vkeys[i]); h->hvals.push_back(vvals[i]); }
    return BBDUK_OK;
}

// ---- table construction, all of it on the device ---------------------------------------------------------------------------
// Two sources -- (key, id) pairs a host built (bbduk_upload_pairs / bbduk_upload_table_way + bbduk_finalize_table: the JVM's own
// tables) or the reference sequences themselves (bbduk_build_table_device, or the streaming bbduk_build_begin / _add_device / _end)
// -- and two layouts, chosen by the expected key count: the cache-resident one (scratch set, then placement into buckets sized for
// the distinct count, LDS filter) and the big one (in place).  BuildState lives from begin to end.
struct BuildState {
    bool seed = false;                   // the seed layout (bbduk_seed.inc): parents under their halves, inserted in place
    bool twin = false;                   // a streamed build of a big / seed map of up to 2^25 keys: every piece also goes into the scratch set, which becomes the cache-resident twin (round 5)
    bool big = false; int hdist = 0, hdist2 = 0, edist = 0, edist2 = 0;      // edist / edist2 (<= 1): bbduk_build_table_device_edits
    uint64_t* d_sk = nullptr; int32_t* d_si = nullptr; uint64_t cslots = 0;     // scratch set of the two-pass build
    unsigned long long* d_cnt = nullptr;                                        // [0] distinct, [1] overflow flag
    uint8_t* d_stage = nullptr; size_t stageCap = 0;                            // upload staging of bbduk_build_table_device / finalize
    int64_t* d_roff = nullptr; int32_t* d_rid = nullptr; uint8_t* d_rfl = nullptr; size_t pieceCap = 0;
};
static void build_release(bbduk_handle* h) {
    BuildState* st = h->build;
    if (!st) return;
    hipFree(st->d_sk); hipFree(st->d_si); hipFree(st->d_cnt); hipFree(st->d_stage); hipFree(st->d_roff); hipFree(st->d_rid); hipFree(st->d_rfl);
    delete st; h->build = nullptr;
}
static void table_release(bbduk_handle* h) {       // a failed build leaves no half-made map behind
    hipFree(h->d_tags); hipFree(h->d_bkv); hipFree(h->d_ldsImage); hipFree(h->d_bigTags); hipFree(h->d_bigKeys); hipFree(h->d_bigIds);
    h->d_tags = nullptr; h->d_bkv = nullptr; h->d_ldsImage = nullptr; h->d_bigTags = nullptr; h->d_bigKeys = nullptr; h->d_bigIds = nullptr;
    h->big = false; h->seed = false; h->nbuckets = 0; h->bigLines = 0; h->ldsBits = 0; h->nkeys = 0; h->nkeysRef = 0;
    hipFree(h->d_tagsAlt); hipFree(h->d_bkvAlt); hipFree(h->d_ldsAlt);
    h->d_tagsAlt = nullptr; h->d_bkvAlt = nullptr; h->d_ldsAlt = nullptr; h->hasAlt = false; h->nbucketsAlt = 0;
    hipFree(h->d_tagsQx); hipFree(h->d_bkvQx); h->d_tagsQx = nullptr; h->d_bkvQx = nullptr; h->qx = false; h->nbucketsQx = 0; h->nkeysQx = 0;
}
// gapped-minimizer geometry of the big layout for this k and middle mask (see "big layout"); false: k too small for it
static bool big_geometry(bbduk_handle* h, const double maxKeys = 0.0) {
    const int k = h->p.k;
    const uint64_t full = (2 * k > 63) ? ~0ULL : ~(~0ULL << (2 * k));
    const uint64_t masked = ~(uint64_t)h->p.middleMask & full;           // 2 bits per masked base
    int H;
    if (!masked) H = k / 2;
    else {
        const int loBit = __builtin_ctzll(masked), hiBit = 63 - __builtin_clzll(masked);
        const int firstMasked = k - 1 - hiBit / 2, lastMasked = k - 1 - loBit / 2;       // base indices, 0 = the k-mer's first base
        H = std::min(firstMasked, k - 1 - lastMasked);
    }
    if (H < 4) return false;
    h->gH = H; h->gD = k - H; h->gm = std::min(10, H - 1);
    // The 32-bit variant of the line function (gap_v32, bbduk_bigs.inc's scan) serves every size the tag word index allows (2^29 lines: 10^10 keys at 0.6
    // per slot); the 52-bit one remains behind BBDUK_HOOK_BIG_LAYOUT = 2 (tests and A/B runs).
    h->gV32 = (maxKeys > 0.0 && !h->hookBig52) ? ((maxKeys > 2147483648.0 || h->hookBigWide) ? 2 : 1) : 0;      // (2: the wide values, gap_v52)
    if (h->gV32) {
        // m follows the reference's size: a minimizer has to be rare in the REFERENCE (4m bits well above log2 of its positions), and
        // every base m gives up widens the window W = H - m + 1 over which consecutive k-mers share a line: 4.6 M keys at k = 31: m = 7, W = 9,
        // 5 keys per run (24 lines per 150-base read) where m = 10 gives 3.5 (34 lines); 10^9 keys keep m = 9-10
        // (measured, 4.6 M keys, k = 31: m = 6 224 Gbases/s -- 5.3 % of the keys spill --, 7 262, 8 255, 9 240, 10 219: ~6 bits of margin over the key count)
        // Re-swept behind the position-ordered gathers (profiles/r04_m_sweep_gather_order.txt).  Two things pull m down -- every base it gives up widens
        // the window, and short halves have little window to give (k = 21, H = 10, 4.6 M keys: m = 6 264, 7 232, 8 191 Gbases/s; k = 17, 3e7 keys: m = 6 140,
        // 7 116) -- and one pushes it up: a minimizer of 2m bases has to be rare among the reference's positions, or lines overflow (k = 21, 3e7 keys:
        // m = 6 185 with 8 % of the keys spilled, m = 7 194; 10^8 keys with m = 6: the build gives up on minimizer lines, 78 Gbases/s).  So: 4m >= log2(keys)
        // + c with c = 1 for short halves and 3 from H = 13 on (k = 27: 3e7 keys m = 7 265, 8 259; k = 31: 4.6 M keys m = 6 288, 7 308, 8 311), at most
        // H - 2 (W >= 3), and never less than log2(keys) - 1 bits.
        const double L2 = std::log2(std::max(maxKeys, 1024.0));
        // Round 6 (128-byte lines: a crowded line's full pairs overflow into its other half before anything is spilled): the 10^10-key map takes m = 9, W = 7 --
        // 31 lines per 150-base read instead of 35, 1.7 % of the keys spilled at 0.5 per slot, 208 against 195 Gbases/s (profiles/r06_c4_sweep.jsonl); m = 8
        // overflows its secondary map there.  10^10 / 4^18 = 0.15 reference positions per minimizer value is what the lines take: c = 2.75 from H = 13 on.
        const double c = H >= 13 ? 2.75 : (double)std::max(1, H - 10);
        const int need = (int)std::ceil((L2 + c) / 4.0), least = (int)std::ceil((L2 - 1.0) / 4.0);
        h->gm = std::max(std::min(6, H - 1), std::min(std::min(10, H - 1), std::max(least, std::min(need, std::max(6, H - 2)))));
    }
    if (h->hookLdsBits >= 4 && h->hookLdsBits <= H - 1) h->gm = h->hookLdsBits;      // (experiments: BBDUK_HOOK_LDS_BITS = m; such maps have no LDS filter)
    h->gW = h->bigPlain ? 0 : H - h->gm + 1;
    if (h->gV32 == 2 && h->gW > 8) h->gV32 = 1;                   // (the wide scan is instantiated for W <= 8: m = 10 gives W <= 7)
    if (h->gV32 == 2 && h->sealTable) h->gV32 = 1;                // (a Seal map stays below 2^28 lines: its kernel knows the four class bits of the 32-bit form only)
    return true;
}
static BigGeom host_geom(const bbduk_handle* h) { BigGeom G; G.k = h->p.k; G.m = h->gm; G.W = h->gW; G.H = h->gH; G.D = h->gD; G.nlines = h->bigLines; G.middleMask = (uint64_t)h->p.middleMask; G.v32 = h->gV32; G.lb = h->gLb; G.sib = h->gSib; return G; }
static Sink make_sink(const bbduk_handle* h, const BuildState* st) {
    Sink S; memset(&S, 0, sizeof S);
    S.big = st->big ? 1 : 0; S.skeys = st->d_sk; S.sids = st->d_si; S.cmask = st->cslots ? st->cslots - 1 : 0;
    S.tags = h->d_bigTags; S.keys = h->d_bigKeys; S.ids = h->d_bigIds; S.idBytes = h->bigIdBytes; S.G = host_geom(h);
    S.tags2 = h->d_tags; S.bkv2 = h->d_bkv; S.bucketBits2 = h->bucketBits; S.bucketMask2 = (uint32_t)(h->nbuckets - 1);
    S.distinct = st->d_cnt;
    return S;
}
// The configurations the big-layout kernels run: plain kfilter (what pick_kernel calls "not general"), no k>31 runs, no findBestMatch.
// Every other configuration keeps the cache-resident layout at any size it can index (2^29 buckets, ~10^9 keys), as before.
// qhdist / qhdist2 > 0: every query k-mer is looked up with its whole Hamming neighbourhood (1 + 3k lookups per position at distance 1).  Only the
// tiled kernels carry that code; the wave kernels stand back (their flag starts at 1).
static bool query_expansion(const bbduk_params& p) { return p.qhdist > 0 || (p.qhdist2 > 0 && p.mink > 0 && p.mink < p.k); }
// Round 6: a handle whose expansion is tabulated (qx_rewrite: the map the kernels look up is keyed by the forward k-mer, no expansion left to do) runs the
// wave kernels like any rcomp=f handle -- the stream scans in their every-hit form (exact hit plane by qx_lookup, bbduk_stream_scan.inc), the pair scans
// with exact lookups.  Not with restrictright: a span that starts inside the read cuts the windows in front of it, and a cut window's rkmer is not its
// kmer's reverse complement -- the tiled kernels keep those handles.
static bool qx_fast(const bbduk_handle* h) { return h->qx && h->p.restrictRight <= 0; }
static bool expands_on_tiles(const bbduk_handle* h) { return query_expansion(h->p) && !qx_fast(h); }
static bool params_general(const bbduk_params& p) {               // the same predicate as pick_kernel's, on the boundary struct
    const bool useShort = p.mink > 0 && p.mink < p.k;
    return p.qhdist > 0 || p.qhdist2 > 0 || p.restrictLeft > 0 || p.restrictRight > 0 || p.skipR1 || p.skipR2 || !p.rcomp ||
           (useShort && p.middleMask != -1) || p.k < 16 || p.qSkip > 1 || p.speed > 0 || p.minKmerFraction != 0.f || p.minCoveredFraction > 0.f;
}
// Round 4: ktrim=r too (its first-hit scan is bbduk_bigs_kernel's, its short k-mers live in the secondary map) -- such maps take the 32-bit line
// function, i.e. up to 2^31 keys; a ktrim map beyond that is refused at build time (big_geometry_ok_for).
// ... and ktrim=l, like kfilter with maxbadkmers > 0 an every-hit scan: bbduk_bigs_every_kernel (the scan's candidates all verified, the mode's facts read
// out of the exact hit plane).
// Round 5: every operator family -- ksplit, ktrim=n, ktrim=rl, findbestmatch and kfilter with mkf / mcf read their facts out of the same exact hit plane
// (bbduk_bigs_every_kernel, wave_body BIGS && EVERY), and so does k > 31 (bbduk_bigs_kbig.hip: the run state machine replayed over the reads that have
// hits).  Still cache-resident at any size: query expansion (qhdist: the tiled kernels) and k < 16 (at most 4^15 keys: the cache-resident layout indexes them all).  The other flags of the GENERAL family --
// restrictleft / restrictright, skipr1 / skipr2, qskip, speed, rcomp=f -- only decide which positions are looked up, cut a window or gate a key: the
// bbduk_bigs_general kernels serve them (kparams_general_flags).
static bool big_layout_eligible(const bbduk_params& p) {
    const bool useShort = p.mink > 0 && p.mink < p.k;
    return p.mode >= BBDUK_MODE_KFILTER && p.mode <= BBDUK_MODE_KSPLIT && p.qhdist == 0 && p.qhdist2 == 0 && p.k >= 16 && !(useShort && p.middleMask != -1);
}
static bool kparams_general_flags(const KParams& K) {            // kparams_general without the thresholds mkf / mcf (they only read the hit plane)
    return K.restrictLeft > 0 || K.restrictRight > 0 || K.skipR1 || K.skipR2 || !K.rcomp || K.qskip > 1 || K.speed > 0;
}
// seed layout (bbduk_seed.inc): the two halves beside the (at most one) masked middle base, each <= 16 bases
static bool seed_geometry(bbduk_handle* h, const double maxKeys = 0.0) {
    const int k = h->p.k;
    if (k < 16 || k > 31) return false;
    const uint64_t full = ~(~0ULL << (2 * k));
    const uint64_t masked = ~(uint64_t)h->p.middleMask & full;
    if (!masked) { h->seedHl = k / 2; h->seedHr = k - k / 2; }
    else {
        if (__builtin_popcountll(masked) != 2) return false;      // more than one masked base
        const int base = k - 1 - __builtin_ctzll(masked) / 2;     // index of the masked base, 0 = the k-mer's first
        h->seedHl = base; h->seedHr = k - 1 - base;
        // seed_check looks the FORWARD window up and relies on rc(mutant) carrying its masked base where the stored rc window does: true only
        // when the mask sits on the mirror-symmetric centre (odd k).  Even k with an explicit midmasklen=1 would lose windows whose canonical
        // strand is the other one and that differ at the masked base plus one more (36/1868 hit windows at k=30) -> plain big layout.
        if (h->seedHl != h->seedHr) return false;
    }
    // m-mers of the halves' minimizers: two candidates per half (W = 2).  Measured on the 4.6 Mbase genome (profiles/bench_hdist_big.py,
    // 2^25 buckets): m = 10 5 Gbases/s, 11 15, 12 34, 13 46, 14 50; plain buckets 26 -- short m-mers have few values and pile their records
    // on few lines (chains of overflowed buckets, walked with dependent gathers), long ones still put a lane's two positions on one line
    h->seedM = std::min(h->seedHl, h->seedHr) - 1;
    // Round 4 (the scan no longer hides it: profiles/r04_seed_m_sweep.txt, 4.6 Mbase genome, halves of 15): m = 14 74 Gbases/s, 13 98.6, 12 118, 11 43
    // -- a shorter m-mer widens the window over which neighbouring halves share a line (W = H - m + 1: 2, 3, 4), until the m-mers are too few for
    // the reference's H-mers (2 x windows of them: 4^11 = 4.2 M against 9.2 M) and pile their records on few lines.  So: 4^m ~ 2 x the H-mers.
    if (maxKeys > 0.0) {
        const double windows = std::max(maxKeys / (1.0 + 3.0 * k), 1024.0);
        const int mm = (int)std::lround((std::log2(2.0 * windows) + 1.0) / 2.0);
        h->seedM = std::max(8, std::min(std::min(h->seedHl, h->seedHr) - 1, mm));
    }
    if (h->hookLdsBits == 0) h->seedM = 0;                        // (experiments: BBDUK_HOOK_LDS_BITS = 0 -> plain buckets, no minimizer lines; 6.. = m)
    else if (h->hookLdsBits >= 6 && h->hookLdsBits <= std::min(h->seedHl, h->seedHr) - 1) h->seedM = h->hookLdsBits;
    return h->seedHl >= 7 && h->seedHr >= 7 && h->seedHl <= 16 && h->seedHr <= 16;
}
// Which layout a map takes, by its (announced) key count.  Measured at the end of round 3 (profiles/r03_layout_mid*.jsonl, r03_l2_boundary*.jsonl;
// kfilter k=31, 2x150 bp, 20 M reads): the cache-resident layout lives off its fingerprint array staying in an XCD's 4 MB of L2 -- 4 MB: 480-500
// Gbases/s, 8 MB: 200-310, 16 MB: 105-135, >= 64 MB: 62 -- while the big layout's minimizer lines hold 170 Gbases/s from 5e5 to 3e7 keys (hdist=0;
// 139 at 1e9, 88 at 1e10) and its plain lines (reference-side hdist>0) 130 / 107 / 87 / 72 at 1.3 / 2.5 / 4.9 / 9.7 M keys against 129 / 81 / 66 / 63.
// So eligible maps (big_layout_eligible) beyond 2^20 keys take the big layout (was: 2^25); the seed layout keeps its threshold (below it the plain
// lines are faster: 58-64 against 60-130).
#define BIG_LAYOUT_MIN_KEYS (1LL << 20)
#define SEED_LAYOUT_MIN_KEYS (1LL << 25)
#define SEED_JOINT_MIN_KEYS (1LL << 22)
// Rounds 3-4a kept the early threshold for k >= 25 only (the pair scan's line arithmetic paid less below: k = 21 64 against the cache-resident map's 76
// Gbases/s at 4.6 M keys).  With bbduk_bigs_kernel it holds for every k the layout serves (profiles/r04_small_k.jsonl, big / cache-resident Gbases/s at
// 1.15 M, 4.6 M, 30 M keys: k = 17 238/204, 144/78, 116/63; k = 19 299/203, 189/78, 120/63; k = 21 320/202, 231/77, 158/62; k = 23 324/200, 268/76, 194/62).
static inline long long big_min_keys(const bbduk_params& p) { (void)p; return BIG_LAYOUT_MIN_KEYS; }
#define BIG_PLAIN_MIN_KEYS (1LL << 21)             // plain lines (hdist > 0 on the reference side) take over later than minimizer lines: build_both

// expected number of keys (an upper bound is fine) -> layout, allocations
static int build_begin_impl(bbduk_handle* h, double maxKeys, int hdist, int hdist2, const bool streamed = false) {
    if (h->finalized) return fail(h, BBDUK_ERR_STATE, "table already finalized");
    if (h->build) return fail(h, BBDUK_ERR_STATE, "a table build is already in progress");
    HIP_TRY(h, hipSetDevice(h->p.device));
    BuildState* st = new (std::nothrow) BuildState();
    if (!st) return BBDUK_ERR_NOMEM;
    h->build = st; st->hdist = hdist; st->hdist2 = hdist2;
    // reference-side Hamming neighbourhoods put ~2/3 of a k-mer's 1+3k variants on one minimizer: such maps take plain lines
    if (hdist > 0) h->bigPlain = true;
    // large hdist=1 maps of the plain first-hit kfilter: the seed layout stores the parents, not their 1+3k neighbours (bbduk_seed.inc)
    // (round 4: with halves of equal length -- one gather per read position, bbduk_seed.inc -- the seed layout beats the big layout's plain lines from
    // 2^22 keys on: a 52 kbase reference, 4.7 M keys, 115 against 87 Gbases/s; 19 M keys 159 / 75; 2.45 M keys 87 / 102 -- profiles/r04_seed_vs_plain.jsonl;
    // up to 2^25 keys build_both keeps the cache-resident twin for the units beyond a wave's planes)
    st->seed = (maxKeys > (double)SEED_JOINT_MIN_KEYS || h->hookSeedLayout) && !h->hookBigLayout && !h->hookNoBigLayout && hdist == 1 && big_layout_eligible(h->p) &&
               h->p.mode == BBDUK_MODE_KFILTER && !h->p.findBestMatch && !(h->p.kbig > h->p.k) && !params_general(h->p) && !h->sealTable && seed_geometry(h, maxKeys);      // (its stream scan is the plain kfilter's)      // (maxbadkmers > 0 and forbidn too, round 4: the walk counts, seed_window resets)
    if (st->seed && h->seedHl != h->seedHr && !(maxKeys > (double)SEED_LAYOUT_MIN_KEYS || h->hookSeedLayout)) st->seed = false;      // (the two-gather form keeps 2^25)
    long long bigMin = hdist > 0 ? std::max<long long>(big_min_keys(h->p), BIG_PLAIN_MIN_KEYS) : big_min_keys(h->p);      // (plain lines take over later: build_both)
    // (a streamed build -- bbduk_build_begin / _add_device / _end -- sees the reference once: it builds the cache-resident twin build_both gives the other
    // builders ALONGSIDE, every piece into both sinks while it is in HBM: st->twin below.  Until round 5 it kept the 2^25-key threshold instead.)
    // (a Seal map -- seal_gpu.h: its own kernel looks the map up, bbduk_seal.inc -- takes the big layout by its size alone, round 4)
    st->big = !st->seed && !h->hookNoBigLayout && (maxKeys > (double)bigMin || h->hookBigLayout) && (big_layout_eligible(h->p) || (h->sealTable && h->p.k >= 16)) && big_geometry(h, maxKeys);
    // ktrim=r has the stream scan of the 32-bit line function only (no pair-scan form): beyond 2^31 keys, or behind the 52-bit hook, it keeps the
    // cache-resident layout (which refuses what it cannot index)
    if (st->big && h->p.mode != BBDUK_MODE_KFILTER && !h->gV32) st->big = false;
    // (the 52-bit line function exists behind its hook only; its pair scan serves the plain kfilter: the families round 5 made eligible keep the cache-resident map there)
    if (st->big && !h->gV32 && !h->sealTable && (params_general(h->p) || h->p.findBestMatch || h->p.kbig > h->p.k)) st->big = false;
    auto bail = [&](int code, const char* msg) { build_release(h); table_release(h); return fail(h, code, msg); };
    if (hipMalloc(&st->d_cnt, 64) != hipSuccess || hipMemsetAsync(st->d_cnt, 0, 64, h->stream) != hipSuccess) return bail(BBDUK_ERR_NOMEM, "hipMalloc");      // [0..3]: Sink::distinct, [4]: the twin's distinct keys
    // the twin of a streamed build: the same sizes build_both keeps one for (wants_twin), never beside a forced layout
    st->twin = streamed && (st->big || st->seed) && maxKeys <= (double)SEED_LAYOUT_MIN_KEYS && !h->hookBigLayout && !h->hookSeedLayout && !h->sealTable;
    if (st->twin) {
        uint64_t cslots = 1024; while ((double)cslots < 2.0 * maxKeys + 16.0) cslots <<= 1;
        if (hipMalloc(&st->d_sk, cslots * 8) != hipSuccess || hipMalloc(&st->d_si, cslots * 4) != hipSuccess) return bail(BBDUK_ERR_NOMEM, "hipMalloc (scratch set of the twin)");
        st->cslots = cslots;
        hipMemsetAsync(st->d_sk, 0xFF, cslots * 8, h->stream);
        hipMemsetAsync(st->d_si, 0x7F, cslots * 4, h->stream);
    }
    if (st->seed) {
        // four records per reference window (two orientations x two halves); maxKeys counted 1 + 3k keys per window
        const double records = 4.0 * maxKeys / (1.0 + 3.0 * h->p.k) + 64.0;
        int sbits = 10;
        while (sbits < 28 && (double)(1ULL << sbits) < records / 0.6) sbits++;      // <= 0.6 records per 4-way bucket: the minimizer lines load unevenly
        if (h->hookBucketBits >= 4 && h->hookBucketBits <= 28) sbits = h->hookBucketBits;
        const uint64_t snb = 1ULL << sbits;
        if ((double)(4 * snb) < records * 1.05) return bail(BBDUK_ERR_ARG, "too many reference windows for the seed layout");
        if (hipMalloc(&h->d_tags, (snb + 1) * 8) != hipSuccess || hipMalloc(&h->d_bkv, 4 * snb * sizeof(uint4)) != hipSuccess) return bail(BBDUK_ERR_NOMEM, "hipMalloc (map)");
        h->seed = true; h->nbuckets = snb; h->bucketBits = sbits;
        hipMemsetAsync(h->d_tags, 0, (snb + 1) * 8, h->stream);
        hipMemsetAsync(h->d_bkv, 0xFF, 4 * snb * sizeof(uint4), h->stream);
    } else if (st->big) {
        // Lines at 0.3-0.6 keys per slot (the lines' loads vary with the minimizers).  12 or 14 bytes per slot: 10^10 keys = 200-240 GB.
        // Round 6: the 32-bit / wide line functions' maps (gV32) have 128-byte lines of 64 slots (big_words32: a full pair overflows into the line's other
        // half before anything is spilled); round 2's 52-bit form behind its hook keeps 64-byte lines of 32 slots.
        const int idBytes = (h->p.numScaffolds <= 65535 && !h->sealTable) ? 2 : 4;      // (a Seal record is a scaffold or SEAL_MULTI | offset into the id lists: 32 bits)
        h->gLb = (h->gV32 && !h->sealTable) ? 4 : 3;              // (Seal's reads hit at almost every position: 64-byte lines keep a run's keys and ids within 256 bytes, big_words32)
        h->gSib = (h->gV32 && !h->sealTable) ? 1 : 0;
        const int lineSlots = 4 << h->gLb;
        size_t freeB = 0, totalB = 0;
        if (hipMemGetInfo(&freeB, &totalB) != hipSuccess) return bail(BBDUK_ERR_DEVICE, "hipMemGetInfo");
        // the secondary map holds what the lines spill: 64-byte lines ~6 % of the keys at 0.6 keys per slot, 128-byte lines 1-3 % (profiles/sim_lines.c; both
        // pairs of a key full); its buckets are sized for twice that at 2.5 keys per bucket (it keeps working, with longer chains, until it is full)
        // (+ the short k-mers of mink, which live there alone -- big_insert: h->expectShort, where the caller can tell)
        int sbits = 10;
        // (a map that fills HBM -- 10^10 keys -- sizes it for 3 %: 1.7-2.4 % spill at 0.5-0.55 keys per slot, profiles/r06_c4_sweep.jsonl, and the ten gigabytes go to the lines)
        while (sbits < 29 && (double)(1ULL << sbits) < (h->gV32 ? (maxKeys > 4e9 ? 0.03 : 0.06) : 0.12) * maxKeys / 2.5 + h->expectShort / 2.0) sbits++;
        const uint64_t snb = 1ULL << sbits;
        const double perLine = (double)lineSlots * (2.0 + 8.0 + idBytes), spillBytes = (double)snb * (8.0 + 64.0);
        uint64_t nlines = 0;
        // The stream scan (bbduk_bigs.inc) does not look beyond a window's primary pair while it streams: a window whose pair carries its class bit
        // becomes a candidate and is looked up afterwards, so full pairs should be rare -- 0.30 keys per slot where the room is there (maps of up to
        // 2^31 keys = 80 GB); the 10^10-key map takes what HBM leaves it (0.5-0.6).
        const double loads32[5] = {0.30, 0.45, 0.5, 0.55, 0.6}, loads52[5] = {0.6, 0.7, 0.8, 0.8, 0.8};
        for (int li = 0; li < 5; li++) {
            const double load = (h->hookBigLoad > 0 && li == 0) ? 0.01 * h->hookBigLoad : (h->gV32 ? loads32[li] : loads52[li]);
            nlines = std::max<uint64_t>(64, (uint64_t)(maxKeys / ((double)lineSlots * load)) + 1);
            // (beyond 0.45 the map is filling the device: leave the batches 20 GB -- 100 M reads of 2x150 with their offsets and results are 17)
            if (nlines < (1ULL << 29) && (double)nlines * perLine + spillBytes + (li >= 2 && li < 4 ? 20e9 : 3e9) < (double)freeB) break;      // (2^29 lines: the 32-bit pair index of the scan's gathers)
            nlines = 0;
        }
        if (!nlines) return bail(BBDUK_ERR_NOMEM, "the map does not fit this device's memory (or the 32-bit pair index: 2^29 lines)");
        if (h->sealTable && nlines >= (1ULL << 28)) return bail(BBDUK_ERR_NOMEM, "a Seal map of more than 2^28 lines (its kernel indexes tag words with 32 bits)");
        if (hipMalloc(&h->d_bigTags, nlines * 2 * (size_t)lineSlots) != hipSuccess || hipMalloc(&h->d_bigKeys, nlines * 8 * (size_t)lineSlots) != hipSuccess ||
            hipMalloc(&h->d_bigIds, nlines * lineSlots * (size_t)idBytes) != hipSuccess ||
            hipMalloc(&h->d_tags, (snb + 1) * 8) != hipSuccess || hipMalloc(&h->d_bkv, 4 * snb * sizeof(uint4)) != hipSuccess) return bail(BBDUK_ERR_NOMEM, "hipMalloc (map)");
        h->big = true; h->bigLines = (uint32_t)nlines; h->bigIdBytes = idBytes; h->nbuckets = snb; h->bucketBits = sbits;
        hipMemsetAsync(h->d_bigTags, 0, nlines * 2 * (size_t)lineSlots, h->stream);
        hipMemsetAsync(h->d_bigKeys, 0xFF, nlines * 8 * (size_t)lineSlots, h->stream);
        hipMemsetAsync(h->d_bigIds, 0xFF, nlines * lineSlots * (size_t)idBytes, h->stream);
        hipMemsetAsync(h->d_tags, 0, (snb + 1) * 8, h->stream);      // (+ the dummy word behind the last bucket: see StreamProbe)
        hipMemsetAsync(h->d_bkv, 0xFF, 4 * snb * sizeof(uint4), h->stream);
    } else {
        uint64_t cslots = 1024; while ((double)cslots < 2.0 * maxKeys + 16.0) cslots <<= 1;      // a power of two, load <= 0.5
        if (cslots > (1ULL << 34)) return bail(BBDUK_ERR_NOMEM, "key set too large for the scratch set");
        if (hipMalloc(&st->d_sk, cslots * 8) != hipSuccess || hipMalloc(&st->d_si, cslots * 4) != hipSuccess) return bail(BBDUK_ERR_NOMEM, "hipMalloc (scratch set)");
        st->cslots = cslots;
        hipMemsetAsync(st->d_sk, 0xFF, cslots * 8, h->stream);
        hipMemsetAsync(st->d_si, 0x7F, cslots * 4, h->stream);                   // 0x7F7F7F7F: larger than any id
    }
    return BBDUK_OK;
}

// pieces of reference sequence already in HBM -> the sink
static int build_add_pieces(bbduk_handle* h, const uint8_t* d_refs, const int64_t* roff, const int32_t* rid, const uint8_t* rfl, int32_t npieces) {
    BuildState* st = h->build;
    const int64_t total = roff[npieces];
    if (npieces == 0 || total == 0) return BBDUK_OK;
    if ((size_t)npieces > st->pieceCap) {
        hipFree(st->d_roff); hipFree(st->d_rid); hipFree(st->d_rfl); st->d_roff = nullptr; st->d_rid = nullptr; st->d_rfl = nullptr; st->pieceCap = 0;
        const size_t cap = (size_t)npieces + 1024;
        if (hipMalloc(&st->d_roff, (cap + 1) * 8) != hipSuccess || hipMalloc(&st->d_rid, cap * 4) != hipSuccess || hipMalloc(&st->d_rfl, cap) != hipSuccess)
            return fail(h, BBDUK_ERR_NOMEM, "hipMalloc (piece table)");
        st->pieceCap = cap;
    }
    HIP_TRY(h, hipMemcpyAsync(st->d_roff, roff, (size_t)(npieces + 1) * 8, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(st->d_rid, rid, (size_t)npieces * 4, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(st->d_rfl, rfl, (size_t)npieces, hipMemcpyHostToDevice, h->stream));
    BuildParams B;
    B.k = h->p.k; B.mink = h->p.mink; B.useShort = (h->p.mink > 0 && h->p.mink < h->p.k) ? 1 : 0; B.hdist = st->hdist; B.hdist2 = st->hdist2;
    B.rcomp = h->p.rcomp; B.middleMask = (uint64_t)h->p.middleMask; B.totalBases = total; B.nrefs = npieces; B.edist = st->edist; B.edist2 = st->edist2;
    if (st->seed) {
        const int grid = (int)std::min<int64_t>((total + 255) / 256, (int64_t)h->numCU * 32);
        bbduk_build_seed_kernel<<<dim3(std::max(grid, 1)), dim3(256), 0, h->stream>>>(B, d_refs, st->d_roff, st->d_rid, h->seedHl, h->seedHr, h->seedM, make_sink(h, st));
        HIP_TRY(h, hipGetLastError());
        if (!st->twin) { HIP_TRY(h, hipStreamSynchronize(h->stream)); return BBDUK_OK; }
    }
    const bool edits = (st->hdist > 0 && st->edist > 0) || (B.useShort && st->hdist2 > 0 && st->edist2 > 0);
    const int V1 = edits ? 8 * B.k - 4 : ((st->hdist > 0 || (B.useShort && st->hdist2 > 0)) ? 1 + 3 * B.k : 1);      // first-level choices per position (emit_variants / emit_edits1)
    const int64_t work = total * (int64_t)V1;
    const int grid = (int)std::min<int64_t>((work + 255) / 256, (int64_t)h->numCU * 32);
    if (!st->seed) bbduk_build_enum_kernel<<<dim3(std::max(grid, 1)), dim3(256), 0, h->stream>>>(B, d_refs, st->d_roff, st->d_rid, st->d_rfl, V1, make_sink(h, st));
    if (st->twin) {                                              // the same pieces into the scratch set of the cache-resident twin
        Sink T = make_sink(h, st); T.big = 0; T.distinct = st->d_cnt + 4;
        bbduk_build_enum_kernel<<<dim3(std::max(grid, 1)), dim3(256), 0, h->stream>>>(B, d_refs, st->d_roff, st->d_rid, st->d_rfl, V1, T);
    }
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipStreamSynchronize(h->stream));                 // the caller may reuse d_refs and the host arrays
    return BBDUK_OK;
}

static int build_end_impl(bbduk_handle* h, const bool canStartOver = false) {      // canStartOver: the caller still holds the keys and repeats the build with plain lines on BBDUK_ERR_NOMEM
    BuildState* st = h->build;
    auto bail = [&](int code, const char* msg) { build_release(h); table_release(h); return fail(h, code, msg); };
    unsigned long long cnt[6] = {0, 0, 0, 0, 0, 0};                  // [0] distinct keys, [1] overflow, [2] spilled, [4] the twin's distinct keys, [5] its overflow
    if (hipMemcpyAsync(cnt, st->d_cnt, 48, hipMemcpyDeviceToHost, h->stream) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess)
        return bail(BBDUK_ERR_DEVICE, "table build failed on the device");
    const unsigned long long distinct = cnt[0];
    // the scratch set -> a cache-resident map: buckets sized for the distinct count, LDS filter
    // (fatal = false: the twin's placement -- a failure frees what it allocated and leaves the map that exists standing)
    auto place_scratch = [&](const unsigned long long nkeysD, const bool hooks, uint64_t*& tags, uint4*& bkv, uint32_t*& lds, uint64_t& nbOut, int& bbitsOut, int& lbOut, const bool fatal = true) -> int {
        auto give_up = [&](int code, const char* msg) -> int {
            if (fatal) return bail(code, msg);
            hipFree(tags); hipFree(bkv); hipFree(lds); tags = nullptr; bkv = nullptr; lds = nullptr;
            return code;
        };
        // 4-way buckets of 15-bit fingerprints, >= 1 bucket per key (load 0.5-1 keys/bucket: ~0.1-0.4 % of buckets
        // overflow and carry the continuation flag, so almost every lookup ends in its home bucket).
        int bbits = 10;
        while (bbits < 32 && (1ULL << bbits) < nkeysD) bbits++;
        // From 2^20 buckets on the fingerprint array would leave the L2 (8 MB and more): half as many buckets -- 1-2 keys per bucket, ~5 % of
        // them overflowing at 2 -- is faster at every size measured (5.6e5 keys: 451 against 309 Gbases/s, 1.04e6: 233 / 203, 1.27e6: 180 / 129,
        // 2.5e6: 97 / 82, 1e7: 65 / 63; profiles/r03_l2_boundary*.jsonl).  Up to 2^19 buckets the array fits and the sparser map wins
        // (5.2e5 keys at 0.98 / 1.97 keys per bucket: 478 / 331).  (This also keeps the slot index within 31 bits for up to 2^30 keys.)
        if (bbits >= 20) bbits--;
        if (hooks && h->hookBucketBits >= 4 && h->hookBucketBits <= 32) bbits = h->hookBucketBits;      // bbduk_test_hook
        const uint64_t nb = 1ULL << bbits;
        if (4 * nb < nkeysD + nb / 8 || 4 * nb > (1ULL << 31)) return give_up(BBDUK_ERR_ARG, "too many keys for the bucket index");
        // Presence filter in front of the map.  Most query k-mers are absent, so one bit per hash slot held in LDS
        // (<=128 KiB per workgroup) answers most of them without leaving the CU.  Size follows the key count.
        auto ceil_log2 = [](uint64_t x) { int b = 0; while ((1ULL << b) < x) b++; return b; };
        int lb = 0;
        if (nkeysD > 0 && nkeysD <= (1ULL << 22)) lb = std::min(MAX_LDS_BITS, std::max(10, ceil_log2(32ULL * nkeysD)));
        if (hooks && h->hookLdsBits >= 0) lb = h->hookLdsBits == 0 ? 0 : std::min(MAX_LDS_BITS, std::max(10, h->hookLdsBits));   // bbduk_test_hook
        if (hipMalloc(&tags, (nb + 1) * sizeof(uint64_t)) != hipSuccess || hipMalloc(&bkv, 4 * nb * sizeof(uint4)) != hipSuccess ||
            (lb && hipMalloc(&lds, ((size_t)1 << (lb - 5)) * 4) != hipSuccess)) return give_up(BBDUK_ERR_NOMEM, "hipMalloc (map)");
        hipMemsetAsync(tags, 0, (nb + 1) * sizeof(uint64_t), h->stream);      // (+ the dummy word behind the last bucket: always zero, see StreamProbe)
        hipMemsetAsync(bkv, 0xFF, 4 * nb * sizeof(uint4), h->stream);
        if (lb) hipMemsetAsync(lds, 0, ((size_t)1 << (lb - 5)) * 4, h->stream);
        const int grid = (int)std::min<uint64_t>((st->cslots + 255) / 256, (uint64_t)h->numCU * 32);
        bbduk_build_place_kernel<<<dim3(grid), dim3(256), 0, h->stream>>>(st->d_sk, st->d_si, st->cslots, tags, bkv, bbits, (uint32_t)(nb - 1), lds, lb);
        if (hipStreamSynchronize(h->stream) != hipSuccess) return give_up(BBDUK_ERR_DEVICE, "device build (placement) failed");
        nbOut = nb; bbitsOut = bbits; lbOut = lb;
        return BBDUK_OK;
    };
    if (st->seed) {                                               // nkeys = RECORDS (four per distinct reference window), not the reference's key count
        if (cnt[1]) return bail(BBDUK_ERR_NOMEM, "the map overflowed: more reference windows than announced to bbduk_build_begin");
        h->nkeys = (int64_t)distinct; h->ldsBits = 0;
    } else if (st->big) {
        if (cnt[1]) return bail(BBDUK_ERR_NOMEM, "the map overflowed: more keys than announced to bbduk_build_begin");
        // Minimizer lines that spill far beyond the ~5.5 % of a random reference hold keys that crowd on few minimizers (an uploaded map of
        // Hamming neighbourhoods: 27 % spilled at 1.8 M keys, and the scan ran at 9.8 Gbases/s through the secondary map): the callers start
        // over with plain lines, as they do when the secondary map overflows
        if (canStartOver && !h->bigPlain && cnt[2] > distinct / 8) return bail(BBDUK_ERR_NOMEM, "the minimizer lines are overloaded");
        h->nkeys = (int64_t)distinct; h->ldsBits = 0; h->nspilled = (int64_t)cnt[2];
    } else {
        if (cnt[1]) return bail(BBDUK_ERR_NOMEM, "the scratch set overflowed: more keys than announced to bbduk_build_begin");
        uint64_t nb = 0; int bbits = 0, lb = 0;
        const int rc = place_scratch(distinct, true, h->d_tags, h->d_bkv, h->d_ldsImage, nb, bbits, lb);
        if (rc != BBDUK_OK) return rc;
        h->nbuckets = nb; h->bucketBits = bbits; h->nkeys = (int64_t)distinct; h->ldsBits = lb;
    }
    if (st->twin) {                                               // the scratch set of a streamed big / seed build -> the cache-resident twin (build_both's, for the other builders)
        // (a twin whose scratch set overflowed -- the caller announced fewer keys than it sent -- or whose placement fails is dropped; the big / seed map is complete
        // and serves every batch, units beyond a wave's planes through the big layout's tiled kernels)
        const int rc = cnt[5] ? BBDUK_ERR_NOMEM : place_scratch(cnt[4], false, h->d_tagsAlt, h->d_bkvAlt, h->d_ldsAlt, h->nbucketsAlt, h->bucketBitsAlt, h->ldsBitsAlt, false);
        if (rc == BBDUK_OK) h->hasAlt = true;
        if (h->seed) h->nkeysRef = (int64_t)cnt[4];               // (bbduk_table_size answers in the reference's key count)
    }
    build_release(h);
    h->finalized = true;
    return BBDUK_OK;
}

extern "C" int bbduk_build_begin(bbduk_handle* h, int64_t max_keys, int32_t hdist, int32_t hdist2) {
    if (!h) return BBDUK_ERR_ARG;
    if (max_keys < 0 || hdist < 0 || hdist > 3 || hdist2 < 0 || hdist2 > 3) return fail(h, BBDUK_ERR_ARG, "build_begin: bad argument (the device build serves hdist <= 3)");
    std::lock_guard<std::mutex> g(h->mu);
    if (!h->hkeys.empty()) return fail(h, BBDUK_ERR_STATE, "pairs were uploaded already: finalize them instead");
    // the short k-mers of mink live in the secondary map of a big-layout map alone (big_insert), which is sized from this estimate: a streamed build
    // does not say how many scaffolds will come, so it is the bound numScaffolds gives -- two ends per scaffold, every length, every variant
    h->expectShort = 0.0;
    if (h->p.mink > 0 && h->p.mink < h->p.k) {
        auto variants = [](int len, int d) { const double t = 3.0 * len; double v = 1.0; if (d >= 1) v += t; if (d >= 2) v += t * (t - 3.0) / 2.0; if (d >= 3) v += t * (t - 3.0) * (t - 6.0) / 6.0; return v; };
        for (int L = h->p.mink; L < h->p.k; L++) h->expectShort += 2.0 * (double)(h->p.numScaffolds - 1) * variants(L, hdist2);
        h->expectShort = std::min(h->expectShort, 4.0 * (double)max_keys + 1e6);      // (never more than the announcement could hold)
    }
    return build_begin_impl(h, (double)max_keys, hdist, hdist2, true);
}
extern "C" int bbduk_build_add_device(bbduk_handle* h, const uint8_t* d_refs, const int64_t* ref_offsets, int32_t n_refs, int32_t first_id) {
    if (!h) return BBDUK_ERR_ARG;
    std::lock_guard<std::mutex> g(h->mu);
    if (!h->build) return fail(h, BBDUK_ERR_STATE, "build_add_device without bbduk_build_begin");
    if (n_refs < 0 || !ref_offsets || ref_offsets[0] != 0 || (n_refs > 0 && ref_offsets[n_refs] > 0 && !d_refs)) return fail(h, BBDUK_ERR_ARG, "build_add_device: bad argument");
    if (first_id < 1 || (int64_t)first_id + n_refs > (int64_t)h->p.numScaffolds) return fail(h, BBDUK_ERR_ARG, "build_add_device: scaffold ids must stay within 1..numScaffolds-1");
    for (int32_t i = 0; i < n_refs; i++) if (ref_offsets[i + 1] < ref_offsets[i]) return fail(h, BBDUK_ERR_ARG, "build_add_device: offsets must ascend");
    HIP_TRY(h, hipSetDevice(h->p.device));
    std::vector<int32_t> rid((size_t)n_refs); std::vector<uint8_t> rfl((size_t)n_refs, (uint8_t)3);
    for (int32_t i = 0; i < n_refs; i++) rid[i] = first_id + i;
    return build_add_pieces(h, d_refs, ref_offsets, rid.data(), rfl.data(), n_refs);
}
extern "C" int bbduk_build_end(bbduk_handle* h) {
    if (!h) return BBDUK_ERR_ARG;
    std::lock_guard<std::mutex> g(h->mu);
    if (!h->build) return fail(h, BBDUK_ERR_STATE, "build_end without bbduk_build_begin");
    HIP_TRY(h, hipSetDevice(h->p.device));
    const int rc = build_end_impl(h);
    return rc == BBDUK_OK ? qx_rewrite(h) : rc;
}

// The pairs a host staged with bbduk_upload_pairs / bbduk_upload_table_way go to the device in chunks and are placed there (one
// thread per pair; 10^8 keys took 17.8 s in a serial host loop, they take about a second this way).
static int finalize_once(bbduk_handle* h) {
    const int64_t n = (int64_t)h->hkeys.size();
    int rc = BBDUK_OK;
  for (int attempt = 0; attempt < 2; attempt++) {                  // second attempt: plain lines, if the minimizer lines spilled too much
    rc = build_begin_impl(h, (double)n, 0, 0);
    if (rc != BBDUK_OK) return rc;
    BuildState* st = h->build;
    auto bail = [&](int code, const char* msg) { build_release(h); table_release(h); return fail(h, code, msg); };
    const int64_t CH = 32LL << 20;                                  // pairs per upload: 384 MB of staging
    if (n > 0) {
        const int64_t cap = std::min<int64_t>(n, CH);
        if (hipMalloc(&st->d_stage, (size_t)cap * 12) != hipSuccess) return bail(BBDUK_ERR_NOMEM, "hipMalloc (upload staging)");
        int64_t* dk = reinterpret_cast<int64_t*>(st->d_stage); int32_t* dv = reinterpret_cast<int32_t*>(st->d_stage + (size_t)cap * 8);
        const Sink S = make_sink(h, st);
        for (int64_t q = 0; q < n; q += CH) {
            const int64_t m = std::min<int64_t>(CH, n - q);
            if (hipMemcpyAsync(dk, h->hkeys.data() + q, (size_t)m * 8, hipMemcpyHostToDevice, h->stream) != hipSuccess ||
                hipMemcpyAsync(dv, h->hvals.data() + q, (size_t)m * 4, hipMemcpyHostToDevice, h->stream) != hipSuccess) return bail(BBDUK_ERR_DEVICE, "uploading the pairs");
            const int grid = (int)std::min<int64_t>((m + 255) / 256, (int64_t)h->numCU * 32);
            bbduk_insert_pairs_kernel<<<dim3(std::max(grid, 1)), dim3(256), 0, h->stream>>>(dk, dv, m, S);
            if (hipStreamSynchronize(h->stream) != hipSuccess) return bail(BBDUK_ERR_DEVICE, "placing the pairs");
        }
    }
    rc = build_end_impl(h, true);
    if (rc == BBDUK_ERR_NOMEM && !h->bigPlain && !h->finalized) { h->bigPlain = true; continue; }
    break;
  }
    return rc;
}
// Would a map of this (announced) size take the big layout -- and is it small enough to keep a cache-resident twin beside it?  (Measured,
// profiles/r03_layout_mid_long.jsonl: against a 4.6 M-key map, 2x150 and 2x1000 reads run 2.3x / 2.1x faster on the big layout's stream scan, but
// units beyond a wave's planes -- the tile and long-read kernels, whose big-layout instantiations look every key up on its own -- 2x3000: 33 against
// 70 Gbases/s, 2x20000: 6.8 against 28.  The twin costs 65 bytes per key up to 2^25 keys, at most 2 GB.)  A forced layout (test hooks) stays single.
static bool wants_twin(bbduk_handle* h, const double maxKeys, const int hdist) {
    if (h->hookBigLayout || h->hookNoBigLayout || h->hookSeedLayout || h->sealTable) return false;
    if (!(maxKeys > (double)big_min_keys(h->p)) || maxKeys > (double)SEED_LAYOUT_MIN_KEYS) return false;
    (void)hdist;
    return big_layout_eligible(h->p) && big_geometry(h);
}
template <class Once>
static int build_both(bbduk_handle* h, const double maxKeys, const int hdist, Once once) {
    if (!wants_twin(h, maxKeys, hdist)) return once();
    h->hookNoBigLayout = true;
    int rc = once();
    h->hookNoBigLayout = false;
    if (rc != BBDUK_OK) return rc;
    if (h->big || h->seed) return BBDUK_OK;                          // (cannot happen: the hook keeps both away)
    // Plain lines (reference-side Hamming neighbourhoods) pay less than minimizer lines: up to 2^21 keys the cache-resident map alone is the faster
    // one (1.27 M keys: 180 against 130 Gbases/s; 2.45 M: 98 / 108; 4.9 M: 72 / 87 -- profiles/r03_layout_mid_hdist1*.jsonl)
    if (hdist > 0 && h->nkeys <= BIG_PLAIN_MIN_KEYS) return BBDUK_OK;
    const int64_t nkeys1 = h->nkeys;
    // the map just built becomes the twin; it is kept OUT of the handle while the second build runs (a build that fails -- and one that
    // starts over with plain lines -- releases whatever map the handle holds)
    uint64_t* tags = h->d_tags; uint4* bkv = h->d_bkv; uint32_t* lds = h->d_ldsImage;
    const uint64_t nb = h->nbuckets; const int bbits = h->bucketBits, lbits = h->ldsBits;
    h->d_tags = nullptr; h->d_bkv = nullptr; h->d_ldsImage = nullptr; h->nbuckets = 0; h->bucketBits = 0; h->ldsBits = 0; h->nkeys = 0; h->finalized = false;
    rc = once();
    if (rc == BBDUK_ERR_NOMEM && !h->finalized) {                    // the second layout found no room (HBM is short): the map already built serves the handle
        table_release(h);
        h->d_tags = tags; h->d_bkv = bkv; h->d_ldsImage = lds; h->nbuckets = nb; h->bucketBits = bbits; h->ldsBits = lbits; h->nkeys = nkeys1; h->finalized = true;
        h->bigPlain = false;
        return BBDUK_OK;
    }
    if (rc != BBDUK_OK || !(h->big || h->seed)) {                    // failed, or not big after all: one cache-resident map is enough
        hipFree(tags); hipFree(bkv); hipFree(lds);
        return rc;
    }
    if (h->seed) h->nkeysRef = nkeys1;                               // (the twin holds the reference's keys: bbduk_table_size answers in its semantics)
    if (h->bigPlain && nkeys1 <= BIG_PLAIN_MIN_KEYS) {               // uploaded pairs that turned out to be such a map (the minimizer lines overflowed): back to the first map
        table_release(h);
        h->d_tags = tags; h->d_bkv = bkv; h->d_ldsImage = lds; h->nbuckets = nb; h->bucketBits = bbits; h->ldsBits = lbits; h->nkeys = nkeys1; h->finalized = true;
        return BBDUK_OK;
    }
    h->d_tagsAlt = tags; h->d_bkvAlt = bkv; h->d_ldsAlt = lds; h->nbucketsAlt = nb; h->bucketBitsAlt = bbits; h->ldsBitsAlt = lbits; h->hasAlt = true;
    return BBDUK_OK;
}
// An uploaded map of reference-side Hamming-1 neighbourhoods (a JVM-built hdist=1 table) -> the seed layout: see bbduk_collapse_parents_kernel.  Called with
// the plain map of the pairs built and the pairs still on the host.  Returns BBDUK_OK whether or not the map was collapsed (a map that is no union of
// full 1-neighbourhoods, or too small to gain, keeps the layout it has); an error only for device failures.
static KParams make_kparams(const bbduk_handle* h);
// Query-side Hamming expansion (qhdist = 1; BBDukIndexMod.java:462-481) tabulated at the end of the table build (round 5).  getValue's answer for
// a window depends on kmer alone but for its first, direct lookup (see qx_lookup), and only the forward k-mers within one substitution of a stored key (in
// either orientation) can get one: they are enumerated, evaluated by the reference's own loop against the map just built, and placed as a second map keyed
// by the forward k-mer -- which the kernels then look up with rcomp = 0 and no expansion at all: one lookup per window where the reference spends 1 + 3k
// (round 4: two data-parallel stages, 64 neighbour evaluations and ~12 gathers per window, 6.8 Gbases/s).  Served: qhdist <= 1 and qhdist2 <= 1, rcomp=t,
// no middle mask (mink, or mm=f: a masked base would make the answer depend on more than the key), speed = 0, a cache-resident map, an expansion of at
// most 2^28 keys; anything else keeps the expansion inside the tiled kernels.
static int qx_rewrite(bbduk_handle* h) {
    const bbduk_params& p = h->p;
    const bool useShort = p.mink > 0 && p.mink < p.k;
    const int qh2 = useShort ? p.qhdist2 : 0;
    if (!h->finalized || h->qx || h->big || h->seed || h->sealTable || h->nkeys < 1) return BBDUK_OK;
    if (!(p.qhdist == 1 || qh2 == 1) || p.qhdist > 1 || qh2 > 1 || !p.rcomp || p.speed > 0 || p.kbig > p.k || p.k > 31) return BBDUK_OK;      // (round 6: with a middle mask too -- keyed by the masked forward k-mer, KParams::qx)
    const double ub = (double)h->nkeys * 2.0 * (1.0 + 3.0 * p.k);
    if (ub > (double)(1ULL << 27)) return BBDUK_OK;                  // (the scratch set is 2-4x that many slots of 16 bytes for the duration of this call: at most 8 GiB -- ADVICE r5)
    HIP_TRY(h, hipSetDevice(p.device));
    uint64_t cslots = 1024; while ((double)cslots < 2.0 * ub + 16.0) cslots <<= 1;
    uint64_t* d_sk = nullptr; int32_t* d_si = nullptr; int32_t* d_sn = nullptr; unsigned long long* d_cnt = nullptr;
    uint64_t* nTags = nullptr; uint4* nBkv = nullptr; uint32_t* nLds = nullptr;
    auto release = [&]() { hipFree(d_sk); hipFree(d_si); hipFree(d_sn); hipFree(d_cnt); hipFree(nTags); hipFree(nBkv); hipFree(nLds); };
    if (hipMalloc(&d_sk, cslots * 8) != hipSuccess || hipMalloc(&d_si, cslots * 4) != hipSuccess || hipMalloc(&d_sn, cslots * 4) != hipSuccess || hipMalloc(&d_cnt, 64) != hipSuccess) { release(); return BBDUK_OK; }
    hipMemsetAsync(d_sk, 0xFF, cslots * 8, h->stream); hipMemsetAsync(d_si, 0x7F, cslots * 4, h->stream); hipMemsetAsync(d_sn, 0, cslots * 4, h->stream); hipMemsetAsync(d_cnt, 0, 64, h->stream);
    KParams K = make_kparams(h);                                    // the reference's map, rcomp = 1, qhdist as given
    K.qhdist2 = qh2;
    Sink S; memset(&S, 0, sizeof S); S.skeys = d_sk; S.sids = d_si; S.cmask = cslots - 1; S.distinct = d_cnt;
    const uint64_t nslots = 4ULL * h->nbuckets;
    const uint64_t work = nslots * 2ULL * (uint64_t)(1 + 3 * p.k);
    bbduk_qx_enum_kernel<<<dim3((unsigned)std::min<uint64_t>((work + 255) / 256, (uint64_t)h->numCU * 64)), dim3(256), 0, h->stream>>>(K, nslots, S);
    bbduk_qx_eval_kernel<<<dim3((unsigned)std::min<uint64_t>((cslots + 255) / 256, (uint64_t)h->numCU * 64)), dim3(256), 0, h->stream>>>(K, d_sk, d_si, d_sn, cslots);
    unsigned long long distinct = 0;
    if (hipMemcpyAsync(&distinct, d_cnt, 8, hipMemcpyDeviceToHost, h->stream) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) { release(); return fail(h, BBDUK_ERR_DEVICE, "tabulating the query expansion"); }
    int bbits = 10;
    while (bbits < 32 && (1ULL << bbits) < distinct) bbits++;
    if (bbits >= 20) bbits--;
    const uint64_t nb = 1ULL << bbits;
    if (4 * nb < distinct + nb / 8 || 4 * nb > (1ULL << 31)) { release(); return BBDUK_OK; }
    auto ceil_log2 = [](uint64_t x) { int b = 0; while ((1ULL << b) < x) b++; return b; };
    int lb = 0;
    if (distinct > 0 && distinct <= (1ULL << 22)) lb = std::min(MAX_LDS_BITS, std::max(10, ceil_log2(32ULL * distinct)));
    if (hipMalloc(&nTags, (nb + 1) * 8) != hipSuccess || hipMalloc(&nBkv, 4 * nb * sizeof(uint4)) != hipSuccess || (lb && hipMalloc(&nLds, ((size_t)1 << (lb - 5)) * 4) != hipSuccess)) { release(); return BBDUK_OK; }
    hipMemsetAsync(nTags, 0, (nb + 1) * 8, h->stream); hipMemsetAsync(nBkv, 0xFF, 4 * nb * sizeof(uint4), h->stream);
    if (lb) hipMemsetAsync(nLds, 0, ((size_t)1 << (lb - 5)) * 4, h->stream);
    bbduk_build_place_kernel<<<dim3((unsigned)std::min<uint64_t>((cslots + 255) / 256, (uint64_t)h->numCU * 32)), dim3(256), 0, h->stream>>>(d_sk, d_si, cslots, nTags, nBkv, bbits, (uint32_t)(nb - 1), nLds, lb, d_sn);
    if (hipStreamSynchronize(h->stream) != hipSuccess) { release(); return fail(h, BBDUK_ERR_DEVICE, "tabulating the query expansion (placement)"); }
    // the expansion becomes the map the kernels look up; the reference's map moves beside it
    h->d_tagsQx = h->d_tags; h->d_bkvQx = h->d_bkv; h->nbucketsQx = h->nbuckets; h->bucketBitsQx = h->bucketBits; h->nkeysQx = h->nkeys;
    hipFree(h->d_ldsImage);
    h->d_tags = nTags; h->d_bkv = nBkv; h->d_ldsImage = nLds; h->nbuckets = nb; h->bucketBits = bbits; h->ldsBits = lb; h->nkeys = (int64_t)distinct; h->qx = true;
    nTags = nullptr; nBkv = nullptr; nLds = nullptr;
    release();
    return BBDUK_OK;
}
static int try_collapse_to_seed(bbduk_handle* h) {
    const int64_t n = (int64_t)h->hkeys.size();
    if (h->seed || h->sealTable || h->hookBigLayout || h->hookNoBigLayout) return BBDUK_OK;
    if (!(h->p.mode == BBDUK_MODE_KFILTER && !h->p.findBestMatch && !(h->p.kbig > h->p.k) && !params_general(h->p) && big_layout_eligible(h->p))) return BBDUK_OK;      // (the seed layout's scan is the plain kfilter's)
    if (!(n > SEED_JOINT_MIN_KEYS || h->hookSeedLayout) || h->expectShort > 0.0) return BBDUK_OK;
    if (!seed_geometry(h, (double)n)) return BBDUK_OK;
    if (h->seedHl != h->seedHr && !(n > SEED_LAYOUT_MIN_KEYS || h->hookSeedLayout)) return BBDUK_OK;
    if (h->nkeys != n) return BBDUK_OK;                              // (duplicates among the pairs: leave it)
    HIP_TRY(h, hipSetDevice(h->p.device));
    const int64_t CH = 32LL << 20;
    const uint64_t cap = (uint64_t)n / 16 + 4096;                    // a Hamming-1 map has n / (1 + 3k) parents; more than n / 16: it is something else
    uint8_t* d_stage = nullptr; uint64_t* d_W = nullptr; int32_t* d_pid = nullptr; unsigned long long* d_cnt = nullptr;
    uint64_t* sTags = nullptr; uint4* sBkv = nullptr;
    auto release = [&]() { hipFree(d_stage); hipFree(d_W); hipFree(d_pid); hipFree(d_cnt); hipFree(sTags); hipFree(sBkv); };
    const int64_t chunk = std::min<int64_t>(n, CH);
    if (hipMalloc(&d_stage, (size_t)chunk * 12) != hipSuccess || hipMalloc(&d_W, cap * 8) != hipSuccess || hipMalloc(&d_pid, cap * 4) != hipSuccess ||
        hipMalloc(&d_cnt, 64) != hipSuccess) { release(); return BBDUK_OK; }      // (no room for the attempt: the plain map stands)
    int64_t* dk = reinterpret_cast<int64_t*>(d_stage); int32_t* dv = reinterpret_cast<int32_t*>(d_stage + (size_t)chunk * 8);
    const KParams Kold = make_kparams(h);
    auto for_chunks = [&](auto launch) -> bool {
        for (int64_t q = 0; q < n; q += CH) {
            const int64_t m = std::min<int64_t>(CH, n - q);
            if (hipMemcpyAsync(dk, h->hkeys.data() + q, (size_t)m * 8, hipMemcpyHostToDevice, h->stream) != hipSuccess ||
                hipMemcpyAsync(dv, h->hvals.data() + q, (size_t)m * 4, hipMemcpyHostToDevice, h->stream) != hipSuccess) return false;
            launch(m, (int)std::min<int64_t>((m + 255) / 256, (int64_t)h->numCU * 32));
            if (hipStreamSynchronize(h->stream) != hipSuccess) return false;
        }
        return true;
    };
    unsigned long long cnt[4] = {0, 0, 0, 0};
    for (int allMiddles = 0; allMiddles < 2; allMiddles++) {         // second attempt: every middle base whose neighbourhood is complete (a strand that turns with it)
        hipMemsetAsync(d_cnt, 0, 64, h->stream);
        if (!for_chunks([&](int64_t m, int grid) { bbduk_collapse_parents_kernel<<<dim3(std::max(grid, 1)), dim3(256), 0, h->stream>>>(Kold, dk, dv, m, allMiddles, d_W, d_pid, d_cnt, cap); })) { release(); return fail(h, BBDUK_ERR_DEVICE, "collapsing the uploaded map (parents)"); }
        if (hipMemcpy(cnt, d_cnt, 32, hipMemcpyDeviceToHost) != hipSuccess) { release(); return fail(h, BBDUK_ERR_DEVICE, "collapsing the uploaded map"); }
        const unsigned long long np = cnt[0];
        if (cnt[2] != 0 || np == 0 || np > cap) { release(); return BBDUK_OK; }
        // the seed map: four records per parent, <= 0.6 records per 4-way bucket (build_begin_impl's sizing)
        const double records = 4.0 * (double)np + 64.0;
        int sbits = 10;
        while (sbits < 28 && (double)(1ULL << sbits) < records / 0.6) sbits++;
        if (h->hookBucketBits >= 4 && h->hookBucketBits <= 28) sbits = h->hookBucketBits;
        const uint64_t snb = 1ULL << sbits;
        if ((double)(4 * snb) < records * 1.05) { release(); return BBDUK_OK; }
        hipFree(sTags); hipFree(sBkv); sTags = nullptr; sBkv = nullptr;
        if (hipMalloc(&sTags, (snb + 1) * 8) != hipSuccess || hipMalloc(&sBkv, 4 * snb * sizeof(uint4)) != hipSuccess) { release(); return BBDUK_OK; }
        hipMemsetAsync(sTags, 0, (snb + 1) * 8, h->stream);
        hipMemsetAsync(sBkv, 0xFF, 4 * snb * sizeof(uint4), h->stream);
        Sink S; memset(&S, 0, sizeof S);
        S.tags2 = sTags; S.bkv2 = sBkv; S.bucketBits2 = sbits; S.bucketMask2 = (uint32_t)(snb - 1); S.distinct = d_cnt + 4;      // [4] records, [5] overflow
        bbduk_collapse_insert_kernel<<<dim3((unsigned)std::min<uint64_t>((np + 255) / 256, (uint64_t)h->numCU * 32)), dim3(256), 0, h->stream>>>(h->p.k, d_W, d_pid, (int64_t)np, h->seedHl, h->seedHr, h->seedM, S);
        unsigned long long rec[2] = {0, 0};
        if (hipMemcpyAsync(rec, d_cnt + 4, 16, hipMemcpyDeviceToHost, h->stream) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) { release(); return fail(h, BBDUK_ERR_DEVICE, "collapsing the uploaded map (insert)"); }
        if (rec[1]) { release(); return BBDUK_OK; }
        KParams Kseed = Kold;
        Kseed.big = 0; Kseed.bigTags = nullptr; Kseed.bigKeys = nullptr; Kseed.bigIds = nullptr; Kseed.bigLines = 0;
        Kseed.seed = 1; Kseed.seedHl = h->seedHl; Kseed.seedHr = h->seedHr; Kseed.seedM = h->seedM;
        Kseed.tags = sTags; Kseed.bkv = sBkv; Kseed.bucketMask = (uint32_t)(snb - 1); Kseed.bucketBits = sbits; Kseed.ldsImage = nullptr; Kseed.ldsBits = 0;
        if (!for_chunks([&](int64_t m, int grid) { bbduk_collapse_check_kernel<<<dim3(std::max(grid, 1)), dim3(256), 0, h->stream>>>(Kseed, dk, dv, m, d_cnt); })) { release(); return fail(h, BBDUK_ERR_DEVICE, "collapsing the uploaded map (check)"); }
        if (hipMemcpy(cnt, d_cnt, 32, hipMemcpyDeviceToHost) != hipSuccess) { release(); return fail(h, BBDUK_ERR_DEVICE, "collapsing the uploaded map"); }
        if (cnt[3] != 0) continue;                                   // some key is not answered (with its id) by the parents found
        // every uploaded key is answered with its id: the seed map replaces the plain one (the cache-resident twin, where there is one, stays)
        hipFree(h->d_bigTags); hipFree(h->d_bigKeys); hipFree(h->d_bigIds); h->d_bigTags = nullptr; h->d_bigKeys = nullptr; h->d_bigIds = nullptr;
        if (!h->big && !h->hasAlt && n <= SEED_LAYOUT_MIN_KEYS) {    // a cache-resident map alone: it becomes the twin
            h->d_tagsAlt = h->d_tags; h->d_bkvAlt = h->d_bkv; h->d_ldsAlt = h->d_ldsImage; h->nbucketsAlt = h->nbuckets; h->bucketBitsAlt = h->bucketBits; h->ldsBitsAlt = h->ldsBits; h->hasAlt = true;
        } else { hipFree(h->d_tags); hipFree(h->d_bkv); hipFree(h->d_ldsImage); }
        h->d_tags = sTags; h->d_bkv = sBkv; h->d_ldsImage = nullptr; h->ldsBits = 0; sTags = nullptr; sBkv = nullptr;
        h->big = false; h->bigLines = 0; h->bigPlain = false; h->seed = true; h->nbuckets = snb; h->bucketBits = sbits;
        h->nkeys = (int64_t)rec[0]; h->nkeysRef = n; h->nspilled = 0;
        release();
        return BBDUK_OK;
    }
    release();
    return BBDUK_OK;
}

extern "C" int bbduk_finalize_table(bbduk_handle* h) {
    if (!h) return BBDUK_ERR_ARG;
    std::lock_guard<std::mutex> g(h->mu);
    if (h->finalized) return fail(h, BBDUK_ERR_STATE, "table already finalized");
    if (h->build) return fail(h, BBDUK_ERR_STATE, "a device-side build is in progress: end it with bbduk_build_end");
    h->expectShort = 0.0;                                            // uploaded pairs: the keys of other lengths are the short k-mers
    { const int sh = 2 * h->p.k; for (const int64_t key : h->hkeys) if (((uint64_t)key >> sh) != 1ULL) h->expectShort += 1.0; }
    const int rc = build_both(h, (double)h->hkeys.size(), 0, [&]() { return finalize_once(h); });
    int rc2 = BBDUK_OK;
    if (rc == BBDUK_OK) rc2 = try_collapse_to_seed(h);                // (a JVM-built hdist=1 table -> the seed layout, where the pairs are exactly that)
    if (rc == BBDUK_OK && rc2 == BBDUK_OK) rc2 = qx_rewrite(h);       // (qhdist = 1: the expansion tabulated)
    if (rc == BBDUK_OK) { h->hkeys.clear(); h->hkeys.shrink_to_fit(); h->hvals.clear(); h->hvals.shrink_to_fit(); }
    return rc != BBDUK_OK ? rc : rc2;
}

// bbduk_build_table_device: the reference sequences are HOST memory here; they go to the device in chunks of whole scaffolds
// (a scaffold longer than a chunk as pieces that overlap by k-1 bases) through bbduk_build_begin / build_add_pieces / bbduk_build_end.
static int build_device_once(bbduk_handle* h, const uint8_t* refs, const int64_t* ref_offsets, const int32_t n_refs, const int32_t hdist, const int32_t hdist2, const double ub,
                             const int32_t edist, const int32_t edist2);
extern "C" int bbduk_build_table_device(bbduk_handle* h, const uint8_t* refs, const int64_t* ref_offsets, int32_t n_refs,
                                        int32_t hdist, int32_t hdist2) {
    return bbduk_build_table_device_edits(h, refs, ref_offsets, n_refs, hdist, hdist2, 0, 0);
}
extern "C" int bbduk_build_table_device_edits(bbduk_handle* h, const uint8_t* refs, const int64_t* ref_offsets, int32_t n_refs,
                                              int32_t hdist, int32_t hdist2, int32_t edist, int32_t edist2) {
    if (!h) return BBDUK_ERR_ARG;
    if (edist < 0 || edist > 1 || edist2 < 0 || edist2 > 1) return fail(h, BBDUK_ERR_ARG, "the device build serves edist <= 1 (as bbduk_host_parse does)");
    if ((edist > 0 && hdist < edist) || (edist2 > 0 && hdist2 < edist2)) return fail(h, BBDUK_ERR_ARG, "hdist must be max(edist, hdist) as BBDukParser.java:146 leaves it");
    std::lock_guard<std::mutex> g(h->mu);
    if (h->finalized) return fail(h, BBDUK_ERR_STATE, "table already finalized");
    if (!h->hkeys.empty()) return fail(h, BBDUK_ERR_STATE, "pairs were uploaded already: finalize them instead");
    if (n_refs < 0 || !ref_offsets || hdist < 0 || hdist > 3 || hdist2 < 0 || hdist2 > 3) return fail(h, BBDUK_ERR_ARG, "bad argument (device build serves hdist <= 3)");
    const int64_t total = n_refs > 0 ? ref_offsets[n_refs] : 0;
    if (n_refs > 0 && (ref_offsets[0] != 0 || total < 0 || (total > 0 && !refs))) return fail(h, BBDUK_ERR_ARG, "bad offsets");
    for (int32_t i = 0; i < n_refs; i++) if (ref_offsets[i + 1] < ref_offsets[i]) return fail(h, BBDUK_ERR_ARG, "bad offsets");
    if (n_refs + 1 != h->p.numScaffolds) return fail(h, BBDUK_ERR_ARG, "numScaffolds given to bbduk_create must be n_refs + 1");
    const int k = h->p.k; const bool useShort = h->p.mink > 0 && h->p.mink < k;
    // upper bound on the keys: every position times the variants within hdist substitutions, plus the short k-mers of mink
    auto variants = [](int len, int d, int ed) {
        if (d > 0 && ed > 0) return 8.0 * len - 4.0;               // emit_edits1: the k-mer, 3 len substitutions, len-1 deletions, 4 (len-1) insertions
        const double t = 3.0 * len; double v = 1.0; if (d >= 1) v += t; if (d >= 2) v += t * (t - 3.0) / 2.0; if (d >= 3) v += t * (t - 3.0) * (t - 6.0) / 6.0; return v; };
    double ub = (double)total * variants(k, hdist, edist);
    h->expectShort = 0.0;
    if (useShort) for (int L = h->p.mink; L < k; L++) h->expectShort += 2.0 * (double)n_refs * variants(L, hdist2, edist2);
    ub += h->expectShort;
    const int rc = build_both(h, ub, hdist, [&]() { return build_device_once(h, refs, ref_offsets, n_refs, hdist, hdist2, ub, edist, edist2); });
    return rc == BBDUK_OK ? qx_rewrite(h) : rc;
}
static int build_device_once(bbduk_handle* h, const uint8_t* refs, const int64_t* ref_offsets, const int32_t n_refs, const int32_t hdist, const int32_t hdist2, const double ub,
                             const int32_t edist, const int32_t edist2) {
    const int k = h->p.k;
    const bool edits = edist > 0 || edist2 > 0;
    const int64_t total = n_refs > 0 ? ref_offsets[n_refs] : 0;
    int rc = BBDUK_OK;
  for (int attempt = 0; attempt < 2; attempt++) {                  // second attempt: plain lines, if the minimizer lines spilled too much
    rc = build_begin_impl(h, ub, hdist, hdist2);
    if (rc != BBDUK_OK) return rc;
    BuildState* st = h->build;
    st->edist = edist; st->edist2 = edist2;
    auto bail = [&](int code, const char* msg) { build_release(h); table_release(h); return fail(h, code, msg); };
    const int64_t CH = 256LL << 20;                                 // bases per upload
    st->stageCap = (size_t)std::min<int64_t>(std::max<int64_t>(total, 16), CH) + 64;
    if (hipMalloc(&st->d_stage, st->stageCap) != hipSuccess) return bail(BBDUK_ERR_NOMEM, "hipMalloc (reference staging)");
    std::vector<int64_t> srcOff, roff(1, 0); std::vector<int32_t> rid; std::vector<uint8_t> rfl;
    auto flush = [&]() -> int {
        if (rid.empty()) return BBDUK_OK;
        for (size_t q = 0; q < rid.size(); q++) {
            const int64_t len = roff[q + 1] - roff[q];
            if (len > 0 && hipMemcpyAsync(st->d_stage + roff[q], refs + srcOff[q], (size_t)len, hipMemcpyHostToDevice, h->stream) != hipSuccess) return BBDUK_ERR_DEVICE;
        }
        const int r = build_add_pieces(h, st->d_stage, roff.data(), rid.data(), rfl.data(), (int32_t)rid.size());
        srcOff.clear(); roff.assign(1, 0); rid.clear(); rfl.clear();
        return r;
    };
    for (int32_t sidx = 0; sidx < n_refs; sidx++) {
        const int64_t s0 = ref_offsets[sidx], n = ref_offsets[sidx + 1] - s0;
        int64_t pos = 0;
        do {                                                        // pieces of at most CH bases, consecutive ones share k-1 bases
            const int64_t len = std::min<int64_t>(n - pos, CH);
            if (roff.back() + len > CH && (rc = flush()) != BBDUK_OK) return bail(rc, "device build (upload)");
            srcOff.push_back(s0 + pos); roff.push_back(roff.back() + len); rid.push_back(sidx + 1);
            rfl.push_back((uint8_t)((pos == 0 ? 1 : 0) | (pos + len == n ? 2 : 0)));
            if (pos + len == n) break;
            pos += len - (edits ? k : k - 1);                       // (edits: one base more, so that the next piece sees the base behind this piece's last window)
        } while (true);
    }
    if ((rc = flush()) != BBDUK_OK) return bail(rc, "device build (enumeration)");
    rc = build_end_impl(h, true);
    if (rc == BBDUK_ERR_NOMEM && !h->bigPlain && !h->finalized) { h->bigPlain = true; continue; }
    break;
  }
    return rc;
}

// diagnostics of the big layout (include/bbduk_test_hooks.h): how many lines hold 0..32 keys
// (32 slots each: the 64-byte lines of round 2's form, or the HALVES of the 128-byte lines)
__global__ void bbduk_line_hist_kernel(const uint64_t* __restrict__ keys, const uint64_t nlines, unsigned long long* __restrict__ hist) {
    for (uint64_t l = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; l < nlines; l += (uint64_t)gridDim.x * blockDim.x) {
        int c = 0;
        for (int q = 0; q < 32; q++) c += keys[32ULL * l + q] != EMPTY_KEY;
        atomicAdd(&hist[c], 1ULL);
    }
}
extern "C" int bbduk_table_line_histogram(bbduk_handle* h, int64_t* out33) {
    if (!h || !out33) return BBDUK_ERR_ARG;
    if (!h->finalized || !h->big) return fail(h, BBDUK_ERR_STATE, "no big-layout map");
    std::lock_guard<std::mutex> g(h->mu);
    HIP_TRY(h, hipSetDevice(h->p.device));
    unsigned long long* d = nullptr;
    HIP_TRY(h, hipMalloc(&d, 33 * 8));
    hipMemsetAsync(d, 0, 33 * 8, h->stream);
    bbduk_line_hist_kernel<<<dim3(h->numCU * 16), dim3(256), 0, h->stream>>>(h->d_bigKeys, (uint64_t)h->bigLines << (h->gLb - 3), d);
    hipMemcpyAsync(out33, d, 33 * 8, hipMemcpyDeviceToHost, h->stream);
    const hipError_t e = hipStreamSynchronize(h->stream);
    hipFree(d);
    return e == hipSuccess ? BBDUK_OK : BBDUK_ERR_DEVICE;
}
extern "C" int bbduk_table_layout(const bbduk_handle* h) { return (h && h->finalized) ? ((h->seed ? 2 : (h->big ? 1 : 0)) + (h->hasAlt ? 4 : 0) + (h->qx ? 8 : 0)) : -1; }
extern "C" int64_t bbduk_table_spilled(const bbduk_handle* h) { return (h && h->finalized && h->big) ? h->nspilled : 0; }
extern "C" int64_t bbduk_table_size(const bbduk_handle* h) { return (h && h->finalized) ? (h->qx ? h->nkeysQx : ((h->seed && h->nkeysRef > 0) ? h->nkeysRef : h->nkeys)) : -1; }
extern "C" int64_t bbduk_table_bytes(const bbduk_handle* h) {
    if (!h || !h->finalized) return -1;
    if (h->big) return ((int64_t)h->bigLines << (h->gLb - 3)) * (64 + 256 + 32 * h->bigIdBytes) + (int64_t)(h->nbuckets * (8 + 4 * 16)) +
                       (h->hasAlt ? (int64_t)(h->nbucketsAlt * (8 + 4 * 16)) + (h->ldsBitsAlt ? (1LL << (h->ldsBitsAlt - 3)) : 0) : 0);
    return (int64_t)(h->nbuckets * (8 + 4 * 16)) + (h->ldsBits ? (1LL << (h->ldsBits - 3)) : 0) +
           (h->hasAlt ? (int64_t)(h->nbucketsAlt * (8 + 4 * 16)) + (h->ldsBitsAlt ? (1LL << (h->ldsBitsAlt - 3)) : 0) : 0) +      // (a seed-layout map's twin)
           (h->qx ? (int64_t)(h->nbucketsQx * (8 + 4 * 16)) : 0);
}

static KParams make_kparams(const bbduk_handle* h) {
    const bbduk_params& p = h->p;
    KParams K;
    memset(&K, 0, sizeof K);
    K.mode = p.mode; K.k = p.k; K.mink = p.mink; K.rcomp = p.rcomp; K.forbidNs = p.forbidNs;
    K.minlen = p.minlen; K.minlen2 = p.minlen2; K.qhdist = p.qhdist; K.qhdist2 = p.qhdist2;
    K.maxBadKmers = p.maxBadKmers; K.minReadLength = p.minReadLength; K.minLenFraction = p.minLenFraction;
    K.rieb = p.removePairsIfEitherBad; K.trimPad = p.trimPad; K.ktrimExclusive = p.ktrimExclusive;
    K.restrictLeft = p.restrictLeft; K.restrictRight = p.restrictRight; K.skipR1 = p.skipR1; K.skipR2 = p.skipR2;
    K.tpe = (p.trimPairsEvenly && (p.mode == BBDUK_MODE_KTRIM_R || p.mode == BBDUK_MODE_KTRIM_TIPS)) ? 1 : 0; K.qskip = p.qSkip; K.speed = p.speed;
    K.mkf = p.mode == BBDUK_MODE_KFILTER ? p.minKmerFraction : 0.f; K.mcf = p.mode == BBDUK_MODE_KFILTER ? p.minCoveredFraction : 0.f;
    K.kbig = p.kbig > p.k ? p.kbig : p.k; K.fbm = p.findBestMatch ? 1 : 0;
    K.mfc = (p.kmaskFullyCovered && p.mode == BBDUK_MODE_KMASK) ? 1 : 0;
    K.tf = p.trimFailuresTo1bp ? 1 : 0;
    if (K.tf) K.rieb = 0;                                          // BBDukParser.java:109
    K.numScaffolds = p.numScaffolds;
    K.useShort = (p.mink > 0 && p.mink < p.k) ? 1 : 0;
    K.mask = (2 * p.k > 63) ? ~0ULL : ~(~0ULL << (2 * p.k));
    K.kmask = 1ULL << (2 * p.k);
    K.middleMask = (uint64_t)p.middleMask;
    K.tags = h->d_tags; K.bkv = h->d_bkv; K.bucketMask = (uint32_t)(h->nbuckets - 1); K.bucketBits = h->bucketBits;
    K.storedKmers = h->nkeys; K.undef = nullptr;
    K.big = h->big ? 1 : 0; K.bigTags = h->d_bigTags; K.bigKeys = h->d_bigKeys; K.bigIds = h->d_bigIds; K.bigIdBytes = h->bigIdBytes; K.bigLines = h->bigLines;
    K.gm = h->gm; K.gW = h->gW; K.gH = h->gH; K.gD = h->gD; K.gV32 = h->gV32; K.gLb = h->gLb; K.gSib = h->gSib;
    K.seed = h->seed ? 1 : 0; K.seedHl = h->seedHl; K.seedHr = h->seedHr; K.seedM = h->seedM;
    K.matchN = nullptr; K.matchIds = nullptr; K.matchCnt = nullptr; K.matchCap = 0;
    K.dbg = h->hookDbg;
    K.ldsImage = h->d_ldsImage; K.ldsBits = h->ldsBits;
    if (h->qx) {                                                  // the tabulated query expansion: forward keys, no expansion left to do (qx_rewrite)
        K.qx = 1; K.qxTags = h->d_tagsQx; K.qxBkv = h->d_bkvQx; K.qxBucketMask = (uint32_t)(h->nbucketsQx - 1); K.qxBucketBits = h->bucketBitsQx;
        K.rcomp = 0; K.qhdist = 0; K.qhdist2 = 0;
        K.qxQh = p.qhdist; K.qxQh2 = (p.mink > 0 && p.mink < p.k) ? p.qhdist2 : 0;
    }
    return K;
}

// the same parameters over the cache-resident twin of a big-layout map (bbduk_handle::hasAlt)
static KParams alt_kparams(const bbduk_handle* h, KParams K) {
    K.big = 0; K.bigTags = nullptr; K.bigKeys = nullptr; K.bigIds = nullptr; K.bigLines = 0; K.seed = 0;
    K.tags = h->d_tagsAlt; K.bkv = h->d_bkvAlt; K.bucketMask = (uint32_t)(h->nbucketsAlt - 1); K.bucketBits = h->bucketBitsAlt;
    K.ldsImage = h->d_ldsAlt; K.ldsBits = h->ldsBitsAlt;
    return K;
}

struct MatchOut { int32_t* n; int32_t* ids; int32_t* counts; int32_t cap; };     // device buffers of bbduk_kfilter_batch_matches*

// Does the stream scan with the exact hit plane (bbduk_stream_every_kernel: ktrim=l, ksplit, ktrim=n, kfilter with a threshold) take this
// handle's batches?  Cache-resident map whose bucket offsets fit 32 bits, no query expansion (the tiled kernels expand), no qskip (its grid
// is per read; the pair scan keeps it), not behind BBDUK_HOOK_PAIR_SCAN.
static bool stream_every_ok(const bbduk_handle* h, const KParams& K) {
    return !K.big && !K.seed && K.bucketBits <= 28 && K.qhdist == 0 && K.qhdist2 == 0 && K.qskip < 2 && !h->hookPairScan;
}

// kbig / findBestMatch (through the kfilter operators) and ksplit: bbduk_kscan_kernel

// The pre-pass flag block (and the timing events) of ring entry evi come round again after EV_RING launches -- possibly on another stream
// while the launch that had them before is still queued (the *_device operators take the caller's stream).  The new launch waits for the
// last kernel of that earlier one (evDone, recorded at the end of every launch function) before it clears the block.  
static hipError_t ring_acquire(bbduk_handle* h, const int evi, hipStream_t st) {
    if (!h->evDone[evi]) return hipEventCreateWithFlags(&h->evDone[evi], hipEventDisableTiming);
    return hipStreamWaitEvent(st, h->evDone[evi], 0);
}
static int launch_kscan(bbduk_handle* h, const uint8_t* d_bases, const int64_t* d_offsets, int64_t n, int64_t total_bases, int32_t paired,
                        int32_t* d_a, int32_t* d_id, uint8_t* d_fl, int32_t* d_left, int32_t* d_right, int64_t* d_counters, hipStream_t st,
                        const uint32_t* d_undef, bool packed, const MatchOut* mo = nullptr, int64_t* d_status = nullptr) {
    KParams K = make_kparams(h);
    K.undef = packed ? d_undef : nullptr;
    K.status = reinterpret_cast<unsigned long long*>(d_status);
    if (mo) { K.matchN = mo->n; K.matchIds = mo->ids; K.matchCnt = mo->counts; K.matchCap = mo->cap; }
    const int red = h->p.mode == BBDUK_MODE_KSPLIT ? RED_SPLIT : (K.fbm ? RED_BEST : RED_BIG);
    typedef void (*kscan_t)(const KParams, const uint8_t*, const int64_t*, const int64_t, const int64_t, const int,
                            int32_t*, int32_t*, uint8_t*, int32_t*, int32_t*, int64_t*);
    typedef void (*kscan_full_t)(const KParams, const uint8_t*, const int64_t*, const int64_t, const int64_t, const int,
                                 int32_t*, int32_t*, uint8_t*, int32_t*, int32_t*, int64_t*, const int*);
    // A big-layout map (round 5: ksplit and findbestmatch take it like the other operators): bbduk_bigs_every_kernel scans, the fallbacks for units beyond
    // a wave's planes run over the cache-resident twin where the map has one (build_both), else their big-layout instantiations (bbduk_big_tiles.hip)
    const bool bigs = K.big != 0;
    if (bigs && (!K.gV32 || h->hookPairScan)) return fail(h, BBDUK_ERR_STATE, "big-layout map with a scan its kernels do not serve");
    if (K.seed) return fail(h, BBDUK_ERR_STATE, "seed-layout map with a scan its kernel does not serve");
    const bool twin = bigs && h->hasAlt;
    const kscan_full_t fn = (bigs && !twin) ? bbduk_pick_kscan_big_tile(red) : (red == RED_SPLIT ? bbduk_kscan_kernel<RED_SPLIT> : (red == RED_BEST ? bbduk_kscan_kernel<RED_BEST> : bbduk_kscan_kernel<RED_BIG>));
    const kscan_full_t lfn = (bigs && !twin) ? bbduk_pick_kscan_big_long(red) : (red == RED_SPLIT ? bbduk_kscan_long_kernel<RED_SPLIT> : (red == RED_BEST ? bbduk_kscan_long_kernel<RED_BEST> : bbduk_kscan_long_kernel<RED_BIG>));
    const size_t dynLds = h->ldsBits ? ((size_t)1 << (h->ldsBits - 3)) : 0;
    const size_t dynLds2 = twin ? (h->ldsBitsAlt ? ((size_t)1 << (h->ldsBitsAlt - 3)) : 0) : dynLds;      // the fallbacks' filter: the twin's
    HIP_TRY(h, ensure_dyn_lds(reinterpret_cast<const void*>(fn), dynLds2));
    const int64_t ntiles = (n + TILE_READS - 1) / TILE_READS;
    const int grid = (int)std::min<int64_t>(ntiles, (int64_t)h->numCU);
    std::lock_guard<std::mutex> lg(h->launchMu);
    const int evi = (int)(h->evCount % bbduk_handle::EV_RING);
    if (!h->ev0[evi]) { HIP_TRY(h, hipEventCreate(&h->ev0[evi])); HIP_TRY(h, hipEventCreate(&h->ev1[evi])); }
    HIP_TRY(h, hipEventRecord(h->ev0[evi], st));
    int* const d_flag = h->d_slowFlag + 4 * evi;      // [0] pre-pass bits (1: a unit beyond a wave's planes, 2: beyond the tile kernel's)
    HIP_TRY(h, ring_acquire(h, evi, st));
    HIP_TRY(h, hipMemsetAsync(d_flag, 0, 4 * sizeof(int), st));
    if (expands_on_tiles(h) || h->hookForceTile) { const int one = 1; HIP_TRY(h, hipMemcpyAsync(d_flag, &one, sizeof(int), hipMemcpyHostToDevice, st)); }   // the tiled kernels expand (BBDUK_HOOK_FORCE_TILE: tests)
    {   // pre-pass per READ: one beyond the tiled kernel's planes sends the batch to bbduk_kscan_long_kernel; ksplit: one beyond a
        // wave's planes (bit 0) sends it to the tiled kernel, else bbduk_wave_kernel<KSPLIT> takes it
        const int sgrid = (int)std::min<int64_t>((n + 255) / 256, (int64_t)h->numCU * 8);
        const bool splitFour = red == RED_SPLIT && !packed && stream_every_ok(h, K) && (K.forbidNs || kparams_general(K));   // four planes per wave: shorter ones (not over a big-layout map: stream_every_ok)
        bbduk_span_kernel<<<dim3(std::max(sgrid, 1)), dim3(256), 0, st>>>(d_offsets, n, 0, d_flag, red == RED_SPLIT && !packed ? (int64_t)(splitFour ? WUNIT_MAX_KM : WUNIT_MAX) : (int64_t)(KM_CAP_BASES - 32), (int64_t)(KM_CAP_BASES - 32));
        if (red != RED_SPLIT) {                                     // findBestMatch, k > 31: a unit (pair) beyond a wave's planes (bit 0) -> the tiled kernel
            const int64_t units = paired ? n / 2 : n;
            const int ugrid = (int)std::min<int64_t>((units + 255) / 256, (int64_t)h->numCU * 8);
            const bool bestFour = red == RED_BEST && stream_every_ok(h, K) && (K.forbidNs || kparams_general(K));     // four planes per wave: shorter ones
            bbduk_span_kernel<<<dim3(std::max(ugrid, 1)), dim3(256), 0, st>>>(d_offsets, n, (int)paired, d_flag, (int64_t)(bestFour ? WUNIT_MAX_KM : WUNIT_MAX), (int64_t)0x7FFFFFFFFFFFLL);
        }
    }
    if (red != RED_SPLIT) {                                         // the main kernel's shape; the pair scan keeps an id list (main_scan_pair_best) or the run state
        K.waveFirst = 1;                                            // (main_scan_pair_kbig) per read
        const bool general = params_general(h->p);
        const bool every = red == RED_BEST && stream_every_ok(h, K);     // findbestmatch: the stream scan, ids gathered per read (wave_body: FBM)
        const batch_kernel_t wk = bigs ? (red == RED_BIG ? bbduk_pick_bigs_kbig(kparams_general_flags(K)) : kparams_general_flags(K) ? bbduk_pick_bigs_general(BBDUK_MODE_FBM) : bbduk_pick_bigs_every(BBDUK_MODE_FBM, false)) : every ? bbduk_pick_stream_every(BBDUK_MODE_FBM, false, K.forbidNs != 0, general)
                                        : bbduk_pick_mode_wave(red == RED_BEST ? BBDUK_MODE_FBM : BBDUK_MODE_KBIG, general, packed, K.forbidNs != 0);
        const size_t waveLds = bigs ? dynLds + WAVE_LDS_BYTES_BIGS : dynLds + ((every && (K.forbidNs || general)) ? WAVE_LDS_BYTES_KM : WAVE_LDS_BYTES);
        HIP_TRY(h, ensure_dyn_lds(reinterpret_cast<const void*>(wk), waveLds));
        const int64_t nmt = (n + MT_READS - 1) / MT_READS;
        const int wgrid = (int)std::min<int64_t>((nmt + NWAVES - 1) / NWAVES, (int64_t)h->numCU);
        wk<<<dim3(std::max(wgrid, 1)), dim3(BLOCK_THREADS), waveLds, st>>>(K, d_bases, d_offsets, n, total_bases, (int)paired, d_a, d_id, d_fl, d_counters, d_flag);
    }
    if (red == RED_SPLIT && !packed) {                              // the main kernel's shape: wave-autonomous mini-tiles, one lane per read in the finish
        K.waveFirst = 1; K.outLeft = d_left; K.outRight = d_right;
        const bool general = params_general(h->p);
        const bool every = stream_every_ok(h, K);
        const batch_kernel_t wk = bigs ? (kparams_general_flags(K) ? bbduk_pick_bigs_general(BBDUK_MODE_KSPLIT) : bbduk_pick_bigs_every(BBDUK_MODE_KSPLIT, true)) : every ? bbduk_pick_stream_every(BBDUK_MODE_KSPLIT, true, K.forbidNs != 0, general) : bbduk_pick_mode_wave(BBDUK_MODE_KSPLIT, general, false, K.forbidNs != 0);
        const size_t waveLds = bigs ? dynLds + WAVE_LDS_BYTES_BIGS : dynLds + ((every && (K.forbidNs || general)) ? WAVE_LDS_BYTES_KM : WAVE_LDS_BYTES);
        HIP_TRY(h, ensure_dyn_lds(reinterpret_cast<const void*>(wk), waveLds));
        const int64_t nmt = (n + MT_READS - 1) / MT_READS;
        const int wgrid = (int)std::min<int64_t>((nmt + NWAVES - 1) / NWAVES, (int64_t)h->numCU);
        wk<<<dim3(std::max(wgrid, 1)), dim3(BLOCK_THREADS), waveLds, st>>>(K, d_bases, d_offsets, n, total_bases, 0, d_a, d_id, d_fl, d_counters, d_flag);
    }
    if (red == RED_SPLIT && packed && bigs) return fail(h, BBDUK_ERR_STATE, "big-layout map with a scan its kernels do not serve");
    const KParams K2 = twin ? alt_kparams(h, K) : K;                // (after K's outputs are set)
    fn<<<dim3(grid), dim3(BLOCK_THREADS), dynLds2, st>>>(K2, d_bases, d_offsets, n, total_bases, (int)paired, d_a, d_id, d_fl, d_left, d_right, d_counters, d_flag);
    HIP_TRY(h, hipEventRecord(h->ev1[evi], st));
    h->evCount++;
    HIP_TRY(h, ensure_dyn_lds(reinterpret_cast<const void*>(lfn), dynLds2));
    const int64_t lunits = (paired && red != RED_SPLIT) ? n / 2 : n;
    const int lgrid = (int)std::min<int64_t>((lunits + NWAVES - 1) / NWAVES, (int64_t)h->numCU);
    lfn<<<dim3(std::max(lgrid, 1)), dim3(BLOCK_THREADS), dynLds2, st>>>(K2, d_bases, d_offsets, n, total_bases, (int)paired, d_a, d_id, d_fl, d_left, d_right, d_counters, d_flag);
    HIP_TRY(h, hipEventRecord(h->evDone[evi], st));
    HIP_TRY(h, hipGetLastError());
    return BBDUK_OK;
}

static int launch_batch(bbduk_handle* h, int wantKfilter, const uint8_t* d_bases, const int64_t* d_offsets, int64_t n,
                        int64_t total_bases, int32_t paired, int32_t* d_a, int32_t* d_id, uint8_t* d_fl,
                        int64_t* d_counters, hipStream_t st, const uint32_t* d_undef = nullptr, bool packed = false, const MatchOut* mo = nullptr,
                        int64_t* d_status = nullptr) {
    if (!h) return BBDUK_ERR_ARG;
    if (!h->finalized) return fail(h, BBDUK_ERR_STATE, "table not finalized");
    if (mo && !h->p.findBestMatch) return fail(h, BBDUK_ERR_STATE, "match lists need findBestMatch (rename / fbm) in the parameters given to bbduk_create");
    if (mo && n > 0 && (!mo->n || mo->cap < 1 || mo->cap > KS_MAX_IDS || !mo->ids || !mo->counts)) return fail(h, BBDUK_ERR_ARG, "match lists: null buffer or max_ids outside 1..64");
    if (h->p.mode == BBDUK_MODE_KMASK || h->p.mode == BBDUK_MODE_KTRIM_TIPS || h->p.mode == BBDUK_MODE_KSPLIT || (h->p.mode == BBDUK_MODE_KFILTER) != (wantKfilter != 0)) return fail(h, BBDUK_ERR_STATE, "operator does not match the mode given to bbduk_create");
    if (n < 0 || total_bases < 0 || (paired && (n & 1))) return fail(h, BBDUK_ERR_ARG, "bad batch shape");
    if (n == 0) return BBDUK_OK;
    if (!d_bases && total_bases > 0) return fail(h, BBDUK_ERR_ARG, "null bases");
    if (!d_offsets || !d_a || !d_id || !d_fl || !d_counters) return fail(h, BBDUK_ERR_ARG, "null buffer");
    if (!packed && ((uintptr_t)d_bases & 15) != 0) return fail(h, BBDUK_ERR_ARG, "d_bases must be 16-byte aligned");
    if (packed && (!d_undef || ((uintptr_t)d_bases & 3) != 0)) return fail(h, BBDUK_ERR_ARG, "packed input needs both planes, 4-byte aligned");
    if ((h->p.kbig > h->p.k && !(h->p.minCoveredFraction > 0.f)) || h->p.findBestMatch)   // countSetKmersBig / findBestMatch behind the kfilter operators;
                                                                                // with mcf the reference runs countCoveredBases on the 31-mers instead (:1038)
        return launch_kscan(h, d_bases, d_offsets, n, total_bases, paired, d_a, d_id, d_fl, nullptr, nullptr, d_counters, st, d_undef, packed, mo, d_status);
    KParams K = make_kparams(h);
    K.undef = packed ? d_undef : nullptr;
    K.status = reinterpret_cast<unsigned long long*>(d_status);
    const int64_t ntiles = (n + TILE_READS - 1) / TILE_READS;
    const size_t dynLds = h->ldsBits ? ((size_t)1 << (h->ldsBits - 3)) : 0;    // tile kernel: the filter only
    const size_t waveLds = dynLds + WAVE_LDS_BYTES;                             // wave kernel: filter + its per-wave state
    KernelPair kp = pick_kernel(K);
    if (packed) kp.wave = kp.wavePacked;
    const bool firstHit = K.mode == BBDUK_MODE_KTRIM_R || (K.mode == BBDUK_MODE_KFILTER && K.maxBadKmers == 0 && K.mkf == 0.f && K.mcf == 0.f);
    // the first-hit scans run as bbduk_stream_kernel (one body for every read length and both parameter families, bbduk_stream_scan.inc);
    // bbduk_wave_kernel's pair scan keeps the every-hit scans, ktrim=l, the big layout, tables beyond 2^28 buckets and a handle with
    // BBDUK_HOOK_PAIR_SCAN set
    size_t waveLdsUse = waveLds;
    int64_t wunitMax = WUNIT_MAX;
    // (query expansion -- qhdist -- runs on the tiled kernels: the pre-pass flag starts at 1 for such handles)
    if (firstHit && !K.big && !K.seed && K.bucketBits <= 28 && K.qhdist == 0 && K.qhdist2 == 0 && !h->hookPairScan && !h->qx) {
        const bool general = kparams_general(K);
        kp.wave = bbduk_pick_stream(K.mode, K.useShort != 0, K.forbidNs != 0, packed, general);
        if (K.forbidNs || general) { waveLdsUse = dynLds + WAVE_LDS_BYTES_KM; wunitMax = WUNIT_MAX_KM; }      // four planes per wave (wave_body: FOURP)
    }
    // (the tabulated query expansion: its first-hit operators take the every-hit form too -- the candidates that see an undefined base need the lanes = candidates verification)
    const bool everyStream = (!firstHit || qx_fast(h)) && (K.mode == BBDUK_MODE_KTRIM_L || K.mode == BBDUK_MODE_KFILTER || (K.mode == BBDUK_MODE_KTRIM_R && qx_fast(h))) && stream_every_ok(h, K);
    if (everyStream) {                                            // ktrim=l, kfilter with a threshold: the stream scan + the exact hit plane
        const bool general = kparams_general(K);
        kp.wave = bbduk_pick_stream_every(K.mode, K.useShort != 0, K.forbidNs != 0, general);
        if (K.forbidNs || general) { waveLdsUse = dynLds + WAVE_LDS_BYTES_KM; wunitMax = WUNIT_MAX_KM; }
    }
    if (K.seed) {                                                 // seed layout: its own stream scan (such a map exists for the first-hit kfilter only)
        if (!(K.mode == BBDUK_MODE_KFILTER && K.mkf == 0.f && K.mcf == 0.f)) return fail(h, BBDUK_ERR_STATE, "seed-layout map with a scan its kernel does not serve");
        kp.wave = bbduk_pick_stream_seed(K.forbidNs != 0, packed); waveLdsUse = dynLds + WAVE_LDS_BYTES_BIGS;      // (candidate planes + the verification's list)
    }
    const bool genFlags = kparams_general_flags(K);
    if (K.big && K.gV32 && firstHit && !genFlags && (!h->hookPairScan || K.mode != BBDUK_MODE_KFILTER)) {      // big layout, 32-bit line function: its own stream scan (bbduk_bigs.inc)
        kp.wave = bbduk_pick_bigs(K.mode, K.useShort != 0, packed); waveLdsUse = dynLds + WAVE_LDS_BYTES_BIGS;
        if (K.mode == BBDUK_MODE_KTRIM_R) kp.tile = bbduk_pick_ktrim_r_big_tile();
    }
    bool bigsChosen = K.big && K.gV32 && firstHit && !genFlags && (!h->hookPairScan || K.mode != BBDUK_MODE_KFILTER);
    if (K.big && K.gV32 && !h->hookPairScan && (!firstHit || genFlags)) {      // ... and its every-hit form: ktrim=l, kfilter with maxbadkmers > 0, mkf, mcf; every scan of the GENERAL family
        bigsChosen = true;
        kp.wave = genFlags ? bbduk_pick_bigs_general(K.mode) : bbduk_pick_bigs_every(K.mode, K.useShort != 0); waveLdsUse = dynLds + WAVE_LDS_BYTES_BIGS;
        if (K.mode == BBDUK_MODE_KTRIM_L) kp.tile = bbduk_pick_ktrim_l_big_tile();
        if (K.mode == BBDUK_MODE_KTRIM_R) kp.tile = bbduk_pick_ktrim_r_big_tile();
    }
    // (pick_kernel hands every big-layout map the kfilter pair scan -- round 2's kernel, which serves plain kfilter only: any other configuration must have
    // been given one of the stream scans above, or it would run a MODE = KFILTER kernel over a ktrim handle: ADVICE r4)
    if (K.big && !bigsChosen && (K.mode != BBDUK_MODE_KFILTER || K.mkf != 0.f || K.mcf > 0.f || genFlags))
        return fail(h, BBDUK_ERR_STATE, "big-layout map with a scan its kernels do not serve (BBDUK_HOOK_PAIR_SCAN serves plain kfilter only)");
    // a big-layout map with a cache-resident twin (build_both): the wave kernel scans the big layout, the fallbacks for units beyond a wave's
    // planes (tile kernel, long-read kernel) run their cache-resident instantiations over the twin
    // every unit to the kernel that holds it (not where the pre-pass flag is set by hand: query expansion and BBDUK_HOOK_FORCE_TILE send whole batches to the tiled kernel)
    K.route = (expands_on_tiles(h) || h->hookForceTile) ? 0 : 1; K.wunitMax = (int32_t)wunitMax;
    const bool twin = (K.big || K.seed) && h->hasAlt;
    const KParams K2 = twin ? alt_kparams(h, K) : K;
    const size_t dynLds2 = twin ? (K2.ldsBits ? ((size_t)1 << (K2.ldsBits - 3)) : 0) : dynLds;
    if (twin) kp.tile = pick_kernel(K2).tile;
    HIP_TRY(h, ensure_dyn_lds(reinterpret_cast<const void*>(kp.wave), waveLdsUse));
    HIP_TRY(h, ensure_dyn_lds(reinterpret_cast<const void*>(kp.tile), dynLds2));
    // pre-pass: if some pair is longer than a wave's planes the tile kernel takes the whole batch, else the wave kernel
    std::lock_guard<std::mutex> lg(h->launchMu);
    const int evi = (int)(h->evCount % bbduk_handle::EV_RING);
    int* const d_flag = h->d_slowFlag + 4 * evi;      // [0] pre-pass bits (1: a unit beyond a wave's planes, 2: beyond the tile kernel's)
    HIP_TRY(h, ring_acquire(h, evi, st));
    HIP_TRY(h, hipMemsetAsync(d_flag, 0, 4 * sizeof(int), st));
    if (expands_on_tiles(h) || h->hookForceTile) { const int one = 1; HIP_TRY(h, hipMemcpyAsync(d_flag, &one, sizeof(int), hipMemcpyHostToDevice, st)); }   // the tiled kernels expand (BBDUK_HOOK_FORCE_TILE: tests)
    {
        const int64_t units = paired ? n / 2 : n;
        const int sgrid = (int)std::min<int64_t>((units + 255) / 256, (int64_t)h->numCU * 8);
        bbduk_span_kernel<<<dim3(std::max(sgrid, 1)), dim3(256), 0, st>>>(d_offsets, n, (int)paired, d_flag, wunitMax, (int64_t)(CAP_BASES - 64));
    }
#ifndef WAVE_WGS_PER_CU
#define WAVE_WGS_PER_CU 1
#endif
    const int perCU = WAVE_WGS_PER_CU;                                            // 1024-thread workgroups; VGPR budget admits one per CU
    const int64_t nmt = (n + MT_READS - 1) / MT_READS;
    const int wgrid = (int)std::min<int64_t>((nmt + NWAVES - 1) / NWAVES, (int64_t)h->numCU * perCU);
    const int tgrid = (int)std::min<int64_t>(ntiles, (int64_t)h->numCU * perCU);
    if (!h->ev0[evi]) { HIP_TRY(h, hipEventCreate(&h->ev0[evi])); HIP_TRY(h, hipEventCreate(&h->ev1[evi])); }
    HIP_TRY(h, hipEventRecord(h->ev0[evi], st));
    kp.wave<<<dim3(wgrid), dim3(BLOCK_THREADS), waveLdsUse, st>>>(K, d_bases, d_offsets, n, total_bases, (int)paired, d_a, d_id, d_fl, d_counters, d_flag);
    kp.tile<<<dim3(tgrid), dim3(BLOCK_THREADS), dynLds2, st>>>(K2, d_bases, d_offsets, n, total_bases, (int)paired, d_a, d_id, d_fl, d_counters, d_flag);
    HIP_TRY(h, hipEventRecord(h->ev1[evi], st));                    // around both: whichever of the two took the batch (the other returns at once)
    h->evCount++;
    {   // reads beyond BBDUK_MAX_READ_LEN: chunked scan, one wave per unit (returns at once unless the pre-pass asked for it)
        const batch_kernel_t lk = K2.mode == BBDUK_MODE_KFILTER ? ((K2.big || K2.seed) ? bbduk_long_kernel<BBDUK_MODE_KFILTER, true> : bbduk_long_kernel<BBDUK_MODE_KFILTER>) :
                                  (K2.mode == BBDUK_MODE_KTRIM_L ? (K2.big ? bbduk_long_kernel<BBDUK_MODE_KTRIM_L, true> : bbduk_long_kernel<BBDUK_MODE_KTRIM_L>) : (K2.big ? bbduk_long_kernel<BBDUK_MODE_KTRIM_R, true> : bbduk_long_kernel<BBDUK_MODE_KTRIM_R>));
        HIP_TRY(h, ensure_dyn_lds(reinterpret_cast<const void*>(lk), dynLds2));
        const int64_t units = paired ? n / 2 : n;
        const int lgrid = (int)std::min<int64_t>((units + NWAVES - 1) / NWAVES, (int64_t)h->numCU);
        lk<<<dim3(std::max(lgrid, 1)), dim3(BLOCK_THREADS), dynLds2, st>>>(K2, d_bases, d_offsets, n, total_bases, (int)paired, d_a, d_id, d_fl, d_counters, d_flag);
    }
    HIP_TRY(h, hipEventRecord(h->evDone[evi], st));
    HIP_TRY(h, hipGetLastError());
    return BBDUK_OK;
}

extern "C" int bbduk_ktrim_batch_device(bbduk_handle* h, const uint8_t* d_bases, const int64_t* d_offsets, int64_t n,
                                        int64_t total_bases, int32_t paired, int32_t* d_out_trimmed, int32_t* d_out_id0,
                                        uint8_t* d_out_flags, int64_t* d_counters, void* stream) {
    return launch_batch(h, 0, d_bases, d_offsets, n, total_bases, paired, d_out_trimmed, d_out_id0, d_out_flags, d_counters, (hipStream_t)stream);
}
extern "C" int bbduk_kfilter_batch_device(bbduk_handle* h, const uint8_t* d_bases, const int64_t* d_offsets, int64_t n,
                                          int64_t total_bases, int32_t paired, int32_t* d_out_found, int32_t* d_out_id,
                                          uint8_t* d_out_flags, int64_t* d_counters, void* stream) {
    return launch_batch(h, 1, d_bases, d_offsets, n, total_bases, paired, d_out_found, d_out_id, d_out_flags, d_counters, (hipStream_t)stream);
}

extern "C" int bbduk_kfilter_batch_matches_device(bbduk_handle* h, const uint8_t* d_bases, const int64_t* d_offsets, int64_t n,
                                                  int64_t total_bases, int32_t paired, int32_t* d_out_found, int32_t* d_out_id,
                                                  uint8_t* d_out_flags, int32_t max_ids, int32_t* d_out_nids, int32_t* d_out_match_ids,
                                                  int32_t* d_out_match_counts, int64_t* d_counters, void* stream) {
    const MatchOut mo{d_out_nids, d_out_match_ids, d_out_match_counts, max_ids};
    return launch_batch(h, 1, d_bases, d_offsets, n, total_bases, paired, d_out_found, d_out_id, d_out_flags, d_counters, (hipStream_t)stream, nullptr, false, &mo);
}

// ---- packed boundary format (2-bit codes + undefined bits)
extern "C" int bbduk_pack_bases_device(const uint8_t* d_bases, int64_t total_bases, uint32_t* d_codes, uint32_t* d_undef, int32_t device, void* stream) {
    if (total_bases < 0 || (total_bases > 0 && (!d_bases || !d_codes || !d_undef))) return BBDUK_ERR_ARG;
    if (total_bases == 0) return BBDUK_OK;
    if (hipSetDevice(device) != hipSuccess) return BBDUK_ERR_DEVICE;
    const int64_t words = (total_bases + 15) >> 4;
    const int grid = (int)std::min<int64_t>((words + 255) / 256, 1 << 20);
    bbduk_pack_kernel<<<dim3(grid), dim3(256), 0, (hipStream_t)stream>>>(d_bases, total_bases, d_codes, reinterpret_cast<uint16_t*>(d_undef));
    return hipGetLastError() == hipSuccess ? BBDUK_OK : BBDUK_ERR_DEVICE;
}
extern "C" int bbduk_pack_bases_host(const uint8_t* bases, int64_t total_bases, uint32_t* codes, uint32_t* undef) {
    if (total_bases < 0 || (total_bases > 0 && (!bases || !codes || !undef))) return BBDUK_ERR_ARG;
    const int64_t cw = (total_bases + 15) >> 4, uw = (total_bases + 31) >> 5;
    for (int64_t w = 0; w < cw; w++) codes[w] = 0;
    for (int64_t w = 0; w < uw; w++) undef[w] = 0;
    for (int64_t b = 0; b < total_bases; b++) {
        const uint8_t l = bases[b] | 0x20;
        const int c = l == 'a' ? 0 : l == 'c' ? 1 : l == 'g' ? 2 : (l == 't' || l == 'u') ? 3 : -1;      // dna/AminoAcid.java:1284-1298
        if (c < 0) undef[b >> 5] |= 1u << (b & 31); else codes[b >> 4] |= (uint32_t)c << (2 * (b & 15));
    }
    for (int64_t b = total_bases; b < 32 * uw; b++) undef[b >> 5] |= 1u << (b & 31);                     // the tail of the last word is undefined
    return BBDUK_OK;
}
extern "C" int bbduk_ktrim_batch_packed_device(bbduk_handle* h, const uint32_t* d_codes, const uint32_t* d_undef, const int64_t* d_offsets, int64_t n,
                                               int64_t total_bases, int32_t paired, int32_t* d_out_trimmed, int32_t* d_out_id0,
                                               uint8_t* d_out_flags, int64_t* d_counters, void* stream) {
    return launch_batch(h, 0, reinterpret_cast<const uint8_t*>(d_codes), d_offsets, n, total_bases, paired, d_out_trimmed, d_out_id0, d_out_flags, d_counters, (hipStream_t)stream, d_undef, true);
}
extern "C" int bbduk_kfilter_batch_packed_device(bbduk_handle* h, const uint32_t* d_codes, const uint32_t* d_undef, const int64_t* d_offsets, int64_t n,
                                                 int64_t total_bases, int32_t paired, int32_t* d_out_found, int32_t* d_out_id,
                                                 uint8_t* d_out_flags, int64_t* d_counters, void* stream) {
    return launch_batch(h, 1, reinterpret_cast<const uint8_t*>(d_codes), d_offsets, n, total_bases, paired, d_out_found, d_out_id, d_out_flags, d_counters, (hipStream_t)stream, d_undef, true);
}

// The host-buffer operators take their offsets from an untrusted caller (the JNI shim): they must ascend from 0 and no read may
// exceed an int, or the kernels would stage from negative / oversized lengths.  O(n), next to a PCIe copy of the same array.
// (The *_device operators trust their caller: the offsets are in HBM.)
static bool offsets_ok(const int64_t* offsets, int64_t n) {
    if (offsets[0] != 0) return false;
    for (int64_t i = 0; i < n; i++) { const int64_t d = offsets[i + 1] - offsets[i]; if (d < 0 || d > 0x7FFFFFFFLL) return false; }
    return true;
}

static int host_batch(bbduk_handle* h, int wantKfilter, const uint8_t* bases, const int64_t* offsets, int64_t n, int32_t paired,
                      int32_t* out_a, int32_t* out_id, uint8_t* out_fl, const uint32_t* undef = nullptr, bool packed = false,
                      const MatchOut* hostMatches = nullptr) {
    if (!h) return BBDUK_ERR_ARG;
    if (n < 0 || !offsets || (n > 0 && (!out_a || !out_id || !out_fl))) return fail(h, BBDUK_ERR_ARG, "bad argument");
    if (hostMatches && n > 0 && (!hostMatches->n || !hostMatches->ids || !hostMatches->counts || hostMatches->cap < 1 || hostMatches->cap > KS_MAX_IDS))
        return fail(h, BBDUK_ERR_ARG, "match lists: null buffer or max_ids outside 1..64");
    if (n == 0) return BBDUK_OK;
    const int64_t total = offsets[n];
    if (!offsets_ok(offsets, n) || (total > 0 && !bases)) return fail(h, BBDUK_ERR_ARG, "bad offsets (must ascend from 0, reads <= INT_MAX bases)");
    if (packed && total > 0 && !undef) return fail(h, BBDUK_ERR_ARG, "packed input needs the undefined-bit plane");
    const size_t baseBytes = packed ? 4 * (size_t)((total + 15) >> 4) : (size_t)total;      // what crosses PCIe for the bases
    const size_t undefBytes = packed ? 4 * (size_t)((total + 31) >> 5) : 0;
    HIP_TRY(h, hipSetDevice(h->p.device));
    // take one of the handle's staging slots (a third submitter waits for the first free one)
    bbduk_handle::Slot* S = nullptr;
    {
        std::unique_lock<std::mutex> lk(h->slotMu);
        h->slotCv.wait(lk, [&] { for (auto& q : h->slot) if (!q.busy) return true; return false; });
        for (auto& q : h->slot) if (!q.busy) { S = &q; break; }
        S->busy = true;
    }
    struct Release { bbduk_handle* h; bbduk_handle::Slot* S; ~Release() { { std::lock_guard<std::mutex> lk(h->slotMu); S->busy = false; } h->slotCv.notify_one(); } } rel{h, S};
    if (!S->stream) HIP_TRY(h, hipStreamCreateWithFlags(&S->stream, hipStreamNonBlocking));
    if (!S->d_status) { HIP_TRY(h, hipMalloc(&S->d_status, sizeof(int64_t))); HIP_TRY(h, hipMemset(S->d_status, 0, sizeof(int64_t))); HIP_TRY(h, hipStreamSynchronize(nullptr)); }
    // a capacity is recorded only once its buffers exist: a failed hipMalloc leaves the handle usable for a smaller batch
    if (undefBytes + 8 > S->cap_undef) {
        hipFree(S->d_undef); S->d_undef = nullptr; S->cap_undef = 0;
        const size_t cap = undefBytes + 8 + undefBytes / 4;
        HIP_TRY(h, hipMalloc(&S->d_undef, cap));
        S->cap_undef = cap;
    }
    if ((size_t)total + 16 > S->cap_bases) {
        hipFree(S->d_bases); S->d_bases = nullptr; S->cap_bases = 0;
        const size_t cap = (size_t)total + 16 + (size_t)total / 4;
        HIP_TRY(h, hipMalloc(&S->d_bases, cap));
        S->cap_bases = cap;
    }
    if ((size_t)n + 1 > S->cap_reads) {
        hipFree(S->d_off); hipFree(S->d_a); hipFree(S->d_id); hipFree(S->d_fl);
        S->d_off = nullptr; S->d_a = nullptr; S->d_id = nullptr; S->d_fl = nullptr; S->cap_reads = 0;
        const size_t cap = (size_t)n + 1 + (size_t)n / 4;
        HIP_TRY(h, hipMalloc(&S->d_off, cap * sizeof(int64_t)));
        HIP_TRY(h, hipMalloc(&S->d_a, cap * sizeof(int32_t)));
        HIP_TRY(h, hipMalloc(&S->d_id, cap * sizeof(int32_t)));
        HIP_TRY(h, hipMalloc(&S->d_fl, cap));
        S->cap_reads = cap;
    }
    const hipStream_t st = S->stream;
    // Round 6: a large call goes through in PIECES of whole pairs -- piece c+1's upload (the slot's copy stream) runs under piece c's kernel and result
    // download (its compute stream; the two copy directions have engines of their own), so that one submitting thread already keeps the link busy:
    // 20 M packed reads per call 57.8 -> 79 Gbases/s unpinned-to-pinned, and from there with the pieces (bench.py end_to_end).  The kernels index reads by their
    // absolute offsets, so a piece is just a range of reads; a plane word two pieces share is uploaded by both with the same content.
    // (only from page-locked buffers -- bbduk_pinned_malloc, BBDukGpu.allocPinned: a copy out of pageable memory is staged by the runtime on the calling thread,
    // and six of them per call cost the pageable path 15-25 %, measured)
    bool pinnedIn = false;
    if (!hostMatches && n >= (1 << 20)) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, bases) == hipSuccess) pinnedIn = at.type == hipMemoryTypeHost;
        else (void)hipGetLastError();                             // (an unregistered host pointer: not an error of ours)
    }
    const int npieces = pinnedIn ? 6 : 1;
    if (npieces > 1) {
        if (!S->copyStream) HIP_TRY(h, hipStreamCreateWithFlags(&S->copyStream, hipStreamNonBlocking));
        for (int c = 0; c < npieces; c++) if (!S->evPiece[c]) HIP_TRY(h, hipEventCreateWithFlags(&S->evPiece[c], hipEventDisableTiming));
        const hipStream_t cs = S->copyStream;
        int64_t r0 = 0;
        for (int c = 0; c < npieces; c++) {
            int64_t r1 = (c + 1 == npieces) ? n : ((n * (int64_t)(c + 1) / npieces) & ~1LL);
            const int64_t b0 = offsets[r0], b1 = offsets[r1];
            if (b1 > b0) {
                if (packed) {
                    const int64_t w0 = b0 >> 4, w1 = (b1 + 15) >> 4, u0 = b0 >> 5, u1 = (b1 + 31) >> 5;
                    HIP_TRY(h, hipMemcpyAsync(S->d_bases + 4 * w0, bases + 4 * w0, (size_t)(4 * (w1 - w0)), hipMemcpyHostToDevice, cs));
                    HIP_TRY(h, hipMemcpyAsync(S->d_undef + 4 * u0, reinterpret_cast<const uint8_t*>(undef) + 4 * u0, (size_t)(4 * (u1 - u0)), hipMemcpyHostToDevice, cs));
                } else HIP_TRY(h, hipMemcpyAsync(S->d_bases + b0, bases + b0, (size_t)(b1 - b0), hipMemcpyHostToDevice, cs));
            }
            HIP_TRY(h, hipMemcpyAsync(S->d_off + r0, offsets + r0, (size_t)(r1 - r0 + 1) * sizeof(int64_t), hipMemcpyHostToDevice, cs));
            HIP_TRY(h, hipEventRecord(S->evPiece[c], cs));
            HIP_TRY(h, hipStreamWaitEvent(st, S->evPiece[c], 0));
            if (r1 > r0) {
                const int rc = launch_batch(h, wantKfilter, S->d_bases, S->d_off + r0, r1 - r0, total, paired, S->d_a + r0, S->d_id + r0, S->d_fl + r0, h->d_counters, st,
                                            reinterpret_cast<const uint32_t*>(S->d_undef), packed, nullptr, S->d_status);
                if (rc != BBDUK_OK) { hipStreamSynchronize(cs); hipStreamSynchronize(st); return rc; }
                HIP_TRY(h, hipMemcpyAsync(out_a + r0, S->d_a + r0, (size_t)(r1 - r0) * sizeof(int32_t), hipMemcpyDeviceToHost, st));
                HIP_TRY(h, hipMemcpyAsync(out_id + r0, S->d_id + r0, (size_t)(r1 - r0) * sizeof(int32_t), hipMemcpyDeviceToHost, st));
                HIP_TRY(h, hipMemcpyAsync(out_fl + r0, S->d_fl + r0, (size_t)(r1 - r0), hipMemcpyDeviceToHost, st));
            }
            r0 = r1;
        }
        int64_t status = 0;
        HIP_TRY(h, hipMemcpyAsync(&status, S->d_status, sizeof status, hipMemcpyDeviceToHost, st));
        HIP_TRY(h, hipStreamSynchronize(st));
        if (status != 0) {
            int64_t z = 0;
            hipMemcpyAsync(S->d_status, &z, sizeof z, hipMemcpyHostToDevice, st);
            hipStreamSynchronize(st);
            return fail(h, -(int)status, "device reported an error (a read longer than BBDUK_MAX_READ_LEN)");
        }
        return BBDUK_OK;
    }
    if (total > 0) HIP_TRY(h, hipMemcpyAsync(S->d_bases, bases, baseBytes, hipMemcpyHostToDevice, st));
    if (undefBytes) HIP_TRY(h, hipMemcpyAsync(S->d_undef, undef, undefBytes, hipMemcpyHostToDevice, st));
    HIP_TRY(h, hipMemcpyAsync(S->d_off, offsets, (size_t)(n + 1) * sizeof(int64_t), hipMemcpyHostToDevice, st));
    MatchOut dm{nullptr, nullptr, nullptr, 0};                  // match lists: device buffers of this call only
    auto release = [&]() { hipFree(dm.n); hipFree(dm.ids); hipFree(dm.counts); };
    if (hostMatches) {
        dm.cap = hostMatches->cap;
        const size_t lw = (size_t)n * (size_t)dm.cap * sizeof(int32_t);
        if (hipMalloc(&dm.n, (size_t)n * sizeof(int32_t)) != hipSuccess || hipMalloc(&dm.ids, lw) != hipSuccess || hipMalloc(&dm.counts, lw) != hipSuccess ||
            hipMemsetAsync(dm.ids, 0, lw, st) != hipSuccess || hipMemsetAsync(dm.counts, 0, lw, st) != hipSuccess) { release(); return fail(h, BBDUK_ERR_NOMEM, "hipMalloc (match lists)"); }
    }
    const int rc = launch_batch(h, wantKfilter, S->d_bases, S->d_off, n, total, paired, S->d_a, S->d_id, S->d_fl, h->d_counters, st,
                                reinterpret_cast<const uint32_t*>(S->d_undef), packed, hostMatches ? &dm : nullptr, S->d_status);
    if (rc != BBDUK_OK) { release(); return rc; }
    if (hostMatches) {
        const size_t lw = (size_t)n * (size_t)dm.cap * sizeof(int32_t);
        const bool okc = hipMemcpyAsync(hostMatches->n, dm.n, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, st) == hipSuccess &&
                         hipMemcpyAsync(hostMatches->ids, dm.ids, lw, hipMemcpyDeviceToHost, st) == hipSuccess &&
                         hipMemcpyAsync(hostMatches->counts, dm.counts, lw, hipMemcpyDeviceToHost, st) == hipSuccess &&
                         hipStreamSynchronize(st) == hipSuccess;
        release();
        if (!okc) return fail(h, BBDUK_ERR_DEVICE, "copying the match lists back");
    }
    HIP_TRY(h, hipMemcpyAsync(out_a, S->d_a, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipMemcpyAsync(out_id, S->d_id, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipMemcpyAsync(out_fl, S->d_fl, (size_t)n, hipMemcpyDeviceToHost, st));
    int64_t status = 0;
    HIP_TRY(h, hipMemcpyAsync(&status, S->d_status, sizeof status, hipMemcpyDeviceToHost, st));      // this slot's own word: the other
    HIP_TRY(h, hipStreamSynchronize(st));                                                             // submitter's errors are its own
    if (status != 0) {
        int64_t z = 0;
        hipMemcpyAsync(S->d_status, &z, sizeof z, hipMemcpyHostToDevice, st);
        hipStreamSynchronize(st);
        return fail(h, -(int)status, "device reported an error (a read longer than BBDUK_MAX_READ_LEN)");
    }
    return BBDUK_OK;
}

extern "C" int bbduk_ktrim_batch(bbduk_handle* h, const uint8_t* bases, const int64_t* offsets, int64_t n, int32_t paired,
                                 int32_t* out_trimmed, int32_t* out_id0, uint8_t* out_flags) {
    return host_batch(h, 0, bases, offsets, n, paired, out_trimmed, out_id0, out_flags);
}
extern "C" int bbduk_kfilter_batch(bbduk_handle* h, const uint8_t* bases, const int64_t* offsets, int64_t n, int32_t paired,
                                   int32_t* out_found, int32_t* out_id, uint8_t* out_flags) {
    return host_batch(h, 1, bases, offsets, n, paired, out_found, out_id, out_flags);
}

extern "C" int bbduk_kfilter_batch_matches(bbduk_handle* h, const uint8_t* bases, const int64_t* offsets, int64_t n, int32_t paired,
                                           int32_t* out_found, int32_t* out_id, uint8_t* out_flags, int32_t max_ids, int32_t* out_nids,
                                           int32_t* out_match_ids, int32_t* out_match_counts) {
    const MatchOut mo{out_nids, out_match_ids, out_match_counts, max_ids};
    return host_batch(h, 1, bases, offsets, n, paired, out_found, out_id, out_flags, nullptr, false, &mo);
}

// ---- ktrim=rl
static int launch_tips(bbduk_handle* h, const uint8_t* d_bases, const int64_t* d_offsets, int64_t n, int64_t total_bases, int32_t paired,
                       int32_t* d_r, int32_t* d_l, int32_t* d_id, uint8_t* d_fl, int64_t* d_counters, hipStream_t st,
                       const uint32_t* d_undef = nullptr, bool packed = false) {
    if (!h) return BBDUK_ERR_ARG;
    if (!h->finalized) return fail(h, BBDUK_ERR_STATE, "table not finalized");
    if (h->p.mode != BBDUK_MODE_KTRIM_TIPS) return fail(h, BBDUK_ERR_STATE, "operator does not match the mode given to bbduk_create");
    if (n < 0 || total_bases < 0 || (paired && (n & 1))) return fail(h, BBDUK_ERR_ARG, "bad batch shape");
    if (n == 0) return BBDUK_OK;
    if (!d_bases && total_bases > 0) return fail(h, BBDUK_ERR_ARG, "null bases");
    if (!d_offsets || !d_r || !d_l || !d_id || !d_fl || !d_counters) return fail(h, BBDUK_ERR_ARG, "null buffer");
    if (!packed && ((uintptr_t)d_bases & 15) != 0) return fail(h, BBDUK_ERR_ARG, "d_bases must be 16-byte aligned");
    if (packed && (!d_undef || ((uintptr_t)d_bases & 3) != 0)) return fail(h, BBDUK_ERR_ARG, "packed input needs both planes, 4-byte aligned");
    KParams K = make_kparams(h);
    K.undef = packed ? d_undef : nullptr;
    const size_t dynLds = h->ldsBits ? ((size_t)1 << (h->ldsBits - 3)) : 0;
    // a big-layout map (round 5): bbduk_bigs_every_kernel<KTRIM_TIPS> scans; units beyond a wave's planes: the twin, or the big-layout tile kernels
    const bool bigs = K.big != 0;
    if (bigs && (!K.gV32 || h->hookPairScan)) return fail(h, BBDUK_ERR_STATE, "big-layout map with a scan its kernels do not serve");
    if (K.seed) return fail(h, BBDUK_ERR_STATE, "seed-layout map with a scan its kernel does not serve");
    const bool twin = bigs && h->hasAlt;
    const size_t dynLds2 = twin ? (h->ldsBitsAlt ? ((size_t)1 << (h->ldsBitsAlt - 3)) : 0) : dynLds;
    const tips_tile_t tfn = (bigs && !twin) ? bbduk_pick_tips_big_tile() : bbduk_ktrimtips_kernel<>;
    const tips_tile_t tlfn = (bigs && !twin) ? bbduk_pick_tips_big_long() : bbduk_long_tips_kernel<>;
    HIP_TRY(h, ensure_dyn_lds(reinterpret_cast<const void*>(tfn), dynLds2));
    const int64_t ntiles = (n + TILE_READS - 1) / TILE_READS;
    const int grid = (int)std::min<int64_t>(ntiles, (int64_t)h->numCU);
    std::lock_guard<std::mutex> lg(h->launchMu);
    const int evi = (int)(h->evCount % bbduk_handle::EV_RING);
    int* const d_flag = h->d_slowFlag + 4 * evi;      // [0] pre-pass bits (1: a unit beyond a wave's planes, 2: beyond the tile kernel's)
    HIP_TRY(h, ring_acquire(h, evi, st));
    HIP_TRY(h, hipMemsetAsync(d_flag, 0, 4 * sizeof(int), st));
    if (expands_on_tiles(h) || h->hookForceTile) { const int one = 1; HIP_TRY(h, hipMemcpyAsync(d_flag, &one, sizeof(int), hipMemcpyHostToDevice, st)); }   // the tiled kernels expand (BBDUK_HOOK_FORCE_TILE: tests)
    {   // pre-pass: a unit (pair) beyond a wave's planes (bit 0) sends the batch to the tiled kernel, a READ beyond the tiled kernel's
        // planes (bit 1) to bbduk_long_tips_kernel; else bbduk_wave_kernel<KTRIM_TIPS> takes it
        const int64_t units = paired ? n / 2 : n;
        const int ugrid = (int)std::min<int64_t>((units + 255) / 256, (int64_t)h->numCU * 8);
        bbduk_span_kernel<<<dim3(std::max(ugrid, 1)), dim3(256), 0, st>>>(d_offsets, n, (int)paired, d_flag, (int64_t)WUNIT_MAX, (int64_t)0x7FFFFFFFFFFFLL);
        const int sgrid = (int)std::min<int64_t>((n + 255) / 256, (int64_t)h->numCU * 8);
        bbduk_span_kernel<<<dim3(std::max(sgrid, 1)), dim3(256), 0, st>>>(d_offsets, n, 0, d_flag, (int64_t)0x7FFFFFFFFFFFLL, (int64_t)(KM_CAP_BASES - 32));
    }
    if (!h->ev0[evi]) { HIP_TRY(h, hipEventCreate(&h->ev0[evi])); HIP_TRY(h, hipEventCreate(&h->ev1[evi])); }
    HIP_TRY(h, hipEventRecord(h->ev0[evi], st));
    {   // the main kernel's shape: wave-autonomous mini-tiles, candidate scan for the right pass, one lane per read in the finish
        K.waveFirst = 1; K.outLeft = d_l;
        const bool general = params_general(h->p);
        const batch_kernel_t wk = bigs ? (kparams_general_flags(K) ? bbduk_pick_bigs_general(BBDUK_MODE_KTRIM_TIPS) : bbduk_pick_bigs_every(BBDUK_MODE_KTRIM_TIPS, true)) : (stream_every_ok(h, K) && !K.forbidNs && !general) ? bbduk_pick_stream_tips(packed)
                                                                                      : bbduk_pick_mode_wave(BBDUK_MODE_KTRIM_TIPS, general, packed, K.forbidNs != 0);
        const size_t waveLds = dynLds + (bigs ? WAVE_LDS_BYTES_BIGS : WAVE_LDS_BYTES);
        HIP_TRY(h, ensure_dyn_lds(reinterpret_cast<const void*>(wk), waveLds));
        const int64_t nmt = (n + MT_READS - 1) / MT_READS;
        const int wgrid = (int)std::min<int64_t>((nmt + NWAVES - 1) / NWAVES, (int64_t)h->numCU);
        wk<<<dim3(std::max(wgrid, 1)), dim3(BLOCK_THREADS), waveLds, st>>>(K, d_bases, d_offsets, n, total_bases, (int)paired, d_r, d_id, d_fl, d_counters, d_flag);
    }
    const KParams K2 = twin ? alt_kparams(h, K) : K;
    tfn<<<dim3(grid), dim3(BLOCK_THREADS), dynLds2, st>>>(K2, d_bases, d_offsets, n, total_bases, (int)paired, d_r, d_l, d_id, d_fl, d_counters, d_flag);
    HIP_TRY(h, hipEventRecord(h->ev1[evi], st));
    h->evCount++;
    HIP_TRY(h, ensure_dyn_lds(reinterpret_cast<const void*>(tlfn), dynLds2));
    const int64_t units = paired ? n / 2 : n;
    const int lgrid = (int)std::min<int64_t>((units + NWAVES - 1) / NWAVES, (int64_t)h->numCU);
    tlfn<<<dim3(std::max(lgrid, 1)), dim3(BLOCK_THREADS), dynLds2, st>>>(K2, d_bases, d_offsets, n, total_bases, (int)paired, d_r, d_l, d_id, d_fl, d_counters, d_flag);
    HIP_TRY(h, hipEventRecord(h->evDone[evi], st));
    HIP_TRY(h, hipGetLastError());
    return BBDUK_OK;
}
extern "C" int bbduk_ktrimtips_batch_device(bbduk_handle* h, const uint8_t* d_bases, const int64_t* d_offsets, int64_t n,
                                            int64_t total_bases, int32_t paired, int32_t* d_out_right, int32_t* d_out_left,
                                            int32_t* d_out_id0, uint8_t* d_out_flags, int64_t* d_counters, void* stream) {
    return launch_tips(h, d_bases, d_offsets, n, total_bases, paired, d_out_right, d_out_left, d_out_id0, d_out_flags, d_counters, (hipStream_t)stream);
}
extern "C" int bbduk_ktrimtips_batch_packed_device(bbduk_handle* h, const uint32_t* d_codes, const uint32_t* d_undef, const int64_t* d_offsets, int64_t n,
                                                   int64_t total_bases, int32_t paired, int32_t* d_out_right, int32_t* d_out_left,
                                                   int32_t* d_out_id0, uint8_t* d_out_flags, int64_t* d_counters, void* stream) {
    return launch_tips(h, reinterpret_cast<const uint8_t*>(d_codes), d_offsets, n, total_bases, paired, d_out_right, d_out_left, d_out_id0, d_out_flags, d_counters,
                       (hipStream_t)stream, d_undef, true);
}
extern "C" int bbduk_ktrimtips_batch(bbduk_handle* h, const uint8_t* bases, const int64_t* offsets, int64_t n, int32_t paired,
                                     int32_t* out_right, int32_t* out_left, int32_t* out_id0, uint8_t* out_flags) {
    if (!h) return BBDUK_ERR_ARG;
    if (n < 0 || !offsets || (n > 0 && (!out_right || !out_left || !out_id0 || !out_flags))) return fail(h, BBDUK_ERR_ARG, "bad argument");
    if (n == 0) return BBDUK_OK;
    const int64_t total = offsets[n];
    if (!offsets_ok(offsets, n) || (total > 0 && !bases)) return fail(h, BBDUK_ERR_ARG, "bad offsets (must ascend from 0, reads <= INT_MAX bases)");
    std::lock_guard<std::mutex> g(h->mu);
    HIP_TRY(h, hipSetDevice(h->p.device));
    uint8_t* db = nullptr; int64_t* doff = nullptr; int32_t* dr = nullptr; int32_t* dl = nullptr; int32_t* did = nullptr; uint8_t* dfl = nullptr;
    auto release = [&]() { hipFree(db); hipFree(doff); hipFree(dr); hipFree(dl); hipFree(did); hipFree(dfl); };
    if (hipMalloc(&db, (size_t)total + 16) != hipSuccess || hipMalloc(&doff, (size_t)(n + 1) * 8) != hipSuccess ||
        hipMalloc(&dr, (size_t)n * 4) != hipSuccess || hipMalloc(&dl, (size_t)n * 4) != hipSuccess ||
        hipMalloc(&did, (size_t)n * 4) != hipSuccess || hipMalloc(&dfl, (size_t)n) != hipSuccess) { release(); return fail(h, BBDUK_ERR_NOMEM, "hipMalloc"); }
    hipError_t e = hipSuccess;
    if (total > 0) e = hipMemcpyAsync(db, bases, (size_t)total, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(doff, offsets, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, h->stream);
    if (e != hipSuccess) { release(); h->err = hipGetErrorString(e); return BBDUK_ERR_DEVICE; }
    const int rc = launch_tips(h, db, doff, n, total, paired, dr, dl, did, dfl, h->d_counters, h->stream);
    if (rc != BBDUK_OK) { release(); return rc; }
    hipMemcpyAsync(out_right, dr, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream);
    hipMemcpyAsync(out_left, dl, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream);
    hipMemcpyAsync(out_id0, did, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream);
    hipMemcpyAsync(out_flags, dfl, (size_t)n, hipMemcpyDeviceToHost, h->stream);
    e = hipStreamSynchronize(h->stream);
    release();
    if (e != hipSuccess) { h->err = hipGetErrorString(e); return BBDUK_ERR_DEVICE; }
    int64_t status = 0;
    HIP_TRY(h, hipMemcpy(&status, h->d_counters + BBDUK_CTR_STATUS, sizeof status, hipMemcpyDeviceToHost));
    if (status != 0) {
        int64_t z = 0;
        hipMemcpy(h->d_counters + BBDUK_CTR_STATUS, &z, sizeof z, hipMemcpyHostToDevice);
        return fail(h, -(int)status, "device reported an error (a read longer than BBDUK_MAX_READ_LEN)");
    }
    return BBDUK_OK;
}

// ---- ktrim=n
static int launch_kmask(bbduk_handle* h, const uint8_t* d_bases, const int64_t* d_offsets, int64_t n, int64_t total_bases, int32_t paired,
                        int32_t* d_a, int32_t* d_id, uint8_t* d_fl, uint32_t* d_mask, int64_t* d_counters, hipStream_t st,
                        const uint32_t* d_undef = nullptr, bool packed = false) {
    if (!h) return BBDUK_ERR_ARG;
    if (!h->finalized) return fail(h, BBDUK_ERR_STATE, "table not finalized");
    if (h->p.mode != BBDUK_MODE_KMASK) return fail(h, BBDUK_ERR_STATE, "operator does not match the mode given to bbduk_create");
    if (n < 0 || total_bases < 0 || (paired && (n & 1))) return fail(h, BBDUK_ERR_ARG, "bad batch shape");
    if (n == 0) return BBDUK_OK;
    if (!d_bases && total_bases > 0) return fail(h, BBDUK_ERR_ARG, "null bases");
    if (!d_offsets || !d_a || !d_id || !d_fl || !d_mask || !d_counters) return fail(h, BBDUK_ERR_ARG, "null buffer");
    if (!packed && ((uintptr_t)d_bases & 15) != 0) return fail(h, BBDUK_ERR_ARG, "d_bases must be 16-byte aligned");
    if (packed && (!d_undef || ((uintptr_t)d_bases & 3) != 0)) return fail(h, BBDUK_ERR_ARG, "packed input needs both planes, 4-byte aligned");
    KParams K = make_kparams(h);
    K.undef = packed ? d_undef : nullptr;
    const size_t dynLds = h->ldsBits ? ((size_t)1 << (h->ldsBits - 3)) : 0;
    // a big-layout map (round 5): bbduk_bigs_every_kernel<KMASK> scans, its exact hit plane is the mask stage's; units beyond a wave's planes: the twin, or
    // the big-layout tile kernels
    const bool bigs = K.big != 0;
    if (bigs && (!K.gV32 || h->hookPairScan)) return fail(h, BBDUK_ERR_STATE, "big-layout map with a scan its kernels do not serve");
    if (K.seed) return fail(h, BBDUK_ERR_STATE, "seed-layout map with a scan its kernel does not serve");
    const bool twin = bigs && h->hasAlt;
    const size_t dynLds2 = twin ? (h->ldsBitsAlt ? ((size_t)1 << (h->ldsBitsAlt - 3)) : 0) : dynLds;
    const kmask_tile_t mfn = (bigs && !twin) ? bbduk_pick_kmask_big_tile() : bbduk_kmask_kernel<>;
    const kmask_long_t mlfn = (bigs && !twin) ? bbduk_pick_kmask_big_long() : bbduk_kmask_long_kernel<>;
    HIP_TRY(h, ensure_dyn_lds(reinterpret_cast<const void*>(mfn), dynLds2));
    HIP_TRY(h, hipMemsetAsync(d_mask, 0, ((size_t)(total_bases + 31) / 32 + 2) * sizeof(uint32_t), st));
    const int64_t ntiles = (n + TILE_READS - 1) / TILE_READS;
    const int grid = (int)std::min<int64_t>(ntiles, (int64_t)h->numCU);
    std::lock_guard<std::mutex> lg(h->launchMu);
    const int evi = (int)(h->evCount % bbduk_handle::EV_RING);
    int* const d_flag = h->d_slowFlag + 4 * evi;      // [0] pre-pass bits (1: a unit beyond a wave's planes, 2: beyond the tile kernel's)
    HIP_TRY(h, ring_acquire(h, evi, st));
    HIP_TRY(h, hipMemsetAsync(d_flag, 0, 4 * sizeof(int), st));
    if (expands_on_tiles(h) || h->hookForceTile) { const int one = 1; HIP_TRY(h, hipMemcpyAsync(d_flag, &one, sizeof(int), hipMemcpyHostToDevice, st)); }   // the tiled kernels expand (BBDUK_HOOK_FORCE_TILE: tests)
    if (!h->ev0[evi]) { HIP_TRY(h, hipEventCreate(&h->ev0[evi])); HIP_TRY(h, hipEventCreate(&h->ev1[evi])); }
    {   // pre-pass: a unit (pair) beyond a wave's planes sends the batch to the tiled kernel (which in turn leaves the reads beyond ITS planes
        // to bbduk_kmask_long_kernel); else bbduk_wave_kernel<KMASK> takes it
        const int64_t units = paired ? n / 2 : n;
        const int ugrid = (int)std::min<int64_t>((units + 255) / 256, (int64_t)h->numCU * 8);
        bbduk_span_kernel<<<dim3(std::max(ugrid, 1)), dim3(256), 0, st>>>(d_offsets, n, (int)paired, d_flag, (int64_t)(bigs ? WUNIT_MAX : WUNIT_MAX_KM), (int64_t)0x7FFFFFFFFFFFLL);
    }
    HIP_TRY(h, hipEventRecord(h->ev0[evi], st));
    {   // the main kernel's shape: wave-autonomous mini-tiles, a fourth plane for the hit positions, one lane per read in the finish
        K.waveFirst = 1; K.outMask = d_mask;
        const bool general = params_general(h->p);
        const batch_kernel_t wk = bigs ? (kparams_general_flags(K) ? bbduk_pick_bigs_general(BBDUK_MODE_KMASK) : bbduk_pick_bigs_every(BBDUK_MODE_KMASK, true)) : stream_every_ok(h, K) ? bbduk_pick_stream_every(BBDUK_MODE_KMASK, true, K.forbidNs != 0, general)
                                                        : bbduk_pick_mode_wave(BBDUK_MODE_KMASK, general, packed, K.forbidNs != 0);
        const size_t waveLds = dynLds + (bigs ? WAVE_LDS_BYTES_BIGS : WAVE_LDS_BYTES_KM);
        HIP_TRY(h, ensure_dyn_lds(reinterpret_cast<const void*>(wk), waveLds));
        const int64_t nmt = (n + MT_READS - 1) / MT_READS;
        const int wgrid = (int)std::min<int64_t>((nmt + NWAVES - 1) / NWAVES, (int64_t)h->numCU);
        wk<<<dim3(std::max(wgrid, 1)), dim3(BLOCK_THREADS), waveLds, st>>>(K, d_bases, d_offsets, n, total_bases, (int)paired, d_a, d_id, d_fl, d_counters, d_flag);
    }
    const KParams K2 = twin ? alt_kparams(h, K) : K;
    mfn<<<dim3(grid), dim3(BLOCK_THREADS), dynLds2, st>>>(K2, d_bases, d_offsets, n, total_bases, (int)paired, d_a, d_id, d_fl, d_mask, d_counters, d_flag);
    HIP_TRY(h, hipEventRecord(h->ev1[evi], st));
    h->evCount++;
    HIP_TRY(h, ensure_dyn_lds(reinterpret_cast<const void*>(mlfn), dynLds2));
    const int lgrid = (int)std::min<int64_t>((n + NWAVES - 1) / NWAVES, (int64_t)h->numCU);     // sequences beyond the tiled kernel's planes
    mlfn<<<dim3(std::max(lgrid, 1)), dim3(BLOCK_THREADS), dynLds2, st>>>(K2, d_bases, d_offsets, n, total_bases, (int)paired, d_a, d_id, d_mask, d_counters, d_flag);
    HIP_TRY(h, hipEventRecord(h->evDone[evi], st));
    HIP_TRY(h, hipGetLastError());
    return BBDUK_OK;
}
// ---- ksplit (unpaired reads)
static int check_ksplit(bbduk_handle* h, const uint8_t* d_bases, const int64_t* d_offsets, int64_t n, int64_t total_bases,
                        const void* a, const void* b, const void* c, const void* d, const void* e, const void* f) {
    if (!h) return BBDUK_ERR_ARG;
    if (!h->finalized) return fail(h, BBDUK_ERR_STATE, "table not finalized");
    if (h->p.mode != BBDUK_MODE_KSPLIT) return fail(h, BBDUK_ERR_STATE, "operator does not match the mode given to bbduk_create");
    if (n < 0 || total_bases < 0) return fail(h, BBDUK_ERR_ARG, "bad batch shape");
    if (n == 0) return BBDUK_OK;
    if (!d_bases && total_bases > 0) return fail(h, BBDUK_ERR_ARG, "null bases");
    if (!d_offsets || !a || !b || !c || !d || !e || !f) return fail(h, BBDUK_ERR_ARG, "null buffer");
    if (((uintptr_t)d_bases & 15) != 0) return fail(h, BBDUK_ERR_ARG, "d_bases must be 16-byte aligned");
    return 1;
}
extern "C" int bbduk_ksplit_batch_device(bbduk_handle* h, const uint8_t* d_bases, const int64_t* d_offsets, int64_t n, int64_t total_bases,
                                         int32_t* d_out_trimmed, int32_t* d_out_leftmost, int32_t* d_out_rightmost, int32_t* d_out_id0,
                                         uint8_t* d_out_flags, int64_t* d_counters, void* stream) {
    const int rc = check_ksplit(h, d_bases, d_offsets, n, total_bases, d_out_trimmed, d_out_leftmost, d_out_rightmost, d_out_id0, d_out_flags, d_counters);
    if (rc != 1) return rc;
    return launch_kscan(h, d_bases, d_offsets, n, total_bases, 0, d_out_trimmed, d_out_id0, d_out_flags, d_out_leftmost, d_out_rightmost, d_counters, (hipStream_t)stream, nullptr, false);
}
extern "C" int bbduk_ksplit_batch(bbduk_handle* h, const uint8_t* bases, const int64_t* offsets, int64_t n,
                                  int32_t* out_trimmed, int32_t* out_leftmost, int32_t* out_rightmost, int32_t* out_id0, uint8_t* out_flags) {
    if (!h) return BBDUK_ERR_ARG;
    if (n < 0 || !offsets || (n > 0 && (!out_trimmed || !out_leftmost || !out_rightmost || !out_id0 || !out_flags))) return fail(h, BBDUK_ERR_ARG, "bad argument");
    if (n == 0) return BBDUK_OK;
    const int64_t total = offsets[n];
    if (!offsets_ok(offsets, n) || (total > 0 && !bases)) return fail(h, BBDUK_ERR_ARG, "bad offsets (must ascend from 0, reads <= INT_MAX bases)");
    std::lock_guard<std::mutex> g(h->mu);
    HIP_TRY(h, hipSetDevice(h->p.device));
    uint8_t* db = nullptr; int64_t* doff = nullptr; int32_t* dx = nullptr; int32_t* dl = nullptr; int32_t* dr = nullptr; int32_t* did = nullptr; uint8_t* dfl = nullptr;
    auto release = [&]() { hipFree(db); hipFree(doff); hipFree(dx); hipFree(dl); hipFree(dr); hipFree(did); hipFree(dfl); };
    if (hipMalloc(&db, (size_t)total + 16) != hipSuccess || hipMalloc(&doff, (size_t)(n + 1) * 8) != hipSuccess ||
        hipMalloc(&dx, (size_t)n * 4) != hipSuccess || hipMalloc(&dl, (size_t)n * 4) != hipSuccess || hipMalloc(&dr, (size_t)n * 4) != hipSuccess ||
        hipMalloc(&did, (size_t)n * 4) != hipSuccess || hipMalloc(&dfl, (size_t)n) != hipSuccess) { release(); return fail(h, BBDUK_ERR_NOMEM, "hipMalloc"); }
    hipError_t e = hipSuccess;
    if (total > 0) e = hipMemcpyAsync(db, bases, (size_t)total, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(doff, offsets, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, h->stream);
    if (e != hipSuccess) { release(); h->err = hipGetErrorString(e); return BBDUK_ERR_DEVICE; }
    const int rc = bbduk_ksplit_batch_device(h, db, doff, n, total, dx, dl, dr, did, dfl, h->d_counters, h->stream);
    if (rc != BBDUK_OK) { release(); return rc; }
    hipMemcpyAsync(out_trimmed, dx, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream);
    hipMemcpyAsync(out_leftmost, dl, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream);
    hipMemcpyAsync(out_rightmost, dr, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream);
    hipMemcpyAsync(out_id0, did, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream);
    hipMemcpyAsync(out_flags, dfl, (size_t)n, hipMemcpyDeviceToHost, h->stream);
    e = hipStreamSynchronize(h->stream);
    release();
    if (e != hipSuccess) { h->err = hipGetErrorString(e); return BBDUK_ERR_DEVICE; }
    int64_t status = 0;
    HIP_TRY(h, hipMemcpy(&status, h->d_counters + BBDUK_CTR_STATUS, sizeof status, hipMemcpyDeviceToHost));
    if (status != 0) {
        int64_t z = 0;
        hipMemcpy(h->d_counters + BBDUK_CTR_STATUS, &z, sizeof z, hipMemcpyHostToDevice);
        return fail(h, -(int)status, "a read exceeds BBDUK_MAX_READ_LEN");
    }
    return BBDUK_OK;
}

extern "C" int bbduk_kmask_batch_device(bbduk_handle* h, const uint8_t* d_bases, const int64_t* d_offsets, int64_t n,
                                        int64_t total_bases, int32_t paired, int32_t* d_out_masked, int32_t* d_out_id0,
                                        uint8_t* d_out_flags, uint32_t* d_out_mask, int64_t* d_counters, void* stream) {
    return launch_kmask(h, d_bases, d_offsets, n, total_bases, paired, d_out_masked, d_out_id0, d_out_flags, d_out_mask, d_counters, (hipStream_t)stream);
}
extern "C" int bbduk_kmask_batch_packed_device(bbduk_handle* h, const uint32_t* d_codes, const uint32_t* d_undef, const int64_t* d_offsets, int64_t n,
                                               int64_t total_bases, int32_t paired, int32_t* d_out_masked, int32_t* d_out_id0,
                                               uint8_t* d_out_flags, uint32_t* d_out_mask, int64_t* d_counters, void* stream) {
    return launch_kmask(h, reinterpret_cast<const uint8_t*>(d_codes), d_offsets, n, total_bases, paired, d_out_masked, d_out_id0, d_out_flags, d_out_mask, d_counters,
                        (hipStream_t)stream, d_undef, true);
}
extern "C" int bbduk_kmask_batch(bbduk_handle* h, const uint8_t* bases, const int64_t* offsets, int64_t n, int32_t paired,
                                 int32_t* out_masked, int32_t* out_id0, uint8_t* out_flags, uint32_t* out_mask) {
    if (!h) return BBDUK_ERR_ARG;
    if (n < 0 || !offsets || (n > 0 && (!out_masked || !out_id0 || !out_flags || !out_mask))) return fail(h, BBDUK_ERR_ARG, "bad argument");
    if (n == 0) return BBDUK_OK;
    const int64_t total = offsets[n];
    if (!offsets_ok(offsets, n) || (total > 0 && !bases)) return fail(h, BBDUK_ERR_ARG, "bad offsets (must ascend from 0, reads <= INT_MAX bases)");
    std::lock_guard<std::mutex> g(h->mu);
    HIP_TRY(h, hipSetDevice(h->p.device));
    uint8_t* db = nullptr; int64_t* doff = nullptr; int32_t* da = nullptr; int32_t* did = nullptr; uint8_t* dfl = nullptr; uint32_t* dm = nullptr;
    const size_t mwords = (size_t)(total + 31) / 32 + 2;
    auto release = [&]() { hipFree(db); hipFree(doff); hipFree(da); hipFree(did); hipFree(dfl); hipFree(dm); };
    if (hipMalloc(&db, (size_t)total + 16) != hipSuccess || hipMalloc(&doff, (size_t)(n + 1) * 8) != hipSuccess ||
        hipMalloc(&da, (size_t)n * 4) != hipSuccess || hipMalloc(&did, (size_t)n * 4) != hipSuccess ||
        hipMalloc(&dfl, (size_t)n) != hipSuccess || hipMalloc(&dm, mwords * 4) != hipSuccess) { release(); return fail(h, BBDUK_ERR_NOMEM, "hipMalloc"); }
    hipError_t e = hipSuccess;
    if (total > 0) e = hipMemcpyAsync(db, bases, (size_t)total, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(doff, offsets, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, h->stream);
    if (e != hipSuccess) { release(); h->err = hipGetErrorString(e); return BBDUK_ERR_DEVICE; }
    const int rc = launch_kmask(h, db, doff, n, total, paired, da, did, dfl, dm, h->d_counters, h->stream);
    if (rc != BBDUK_OK) { release(); return rc; }
    hipMemcpyAsync(out_masked, da, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream);
    hipMemcpyAsync(out_id0, did, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream);
    hipMemcpyAsync(out_flags, dfl, (size_t)n, hipMemcpyDeviceToHost, h->stream);
    hipMemcpyAsync(out_mask, dm, ((size_t)(total + 31) / 32) * 4, hipMemcpyDeviceToHost, h->stream);
    e = hipStreamSynchronize(h->stream);
    release();
    if (e != hipSuccess) { h->err = hipGetErrorString(e); return BBDUK_ERR_DEVICE; }
    int64_t status = 0;
    HIP_TRY(h, hipMemcpy(&status, h->d_counters + BBDUK_CTR_STATUS, sizeof status, hipMemcpyDeviceToHost));
    if (status != 0) {
        int64_t z = 0;
        hipMemcpy(h->d_counters + BBDUK_CTR_STATUS, &z, sizeof z, hipMemcpyHostToDevice);
        return fail(h, -(int)status, "device reported an error (a read longer than BBDUK_MAX_READ_LEN)");
    }
    return BBDUK_OK;
}

extern "C" int bbduk_ktrim_batch_packed(bbduk_handle* h, const uint32_t* codes, const uint32_t* undef, const int64_t* offsets, int64_t n, int32_t paired,
                                        int32_t* out_trimmed, int32_t* out_id0, uint8_t* out_flags) {
    return host_batch(h, 0, reinterpret_cast<const uint8_t*>(codes), offsets, n, paired, out_trimmed, out_id0, out_flags, undef, true);
}
extern "C" int bbduk_kfilter_batch_packed(bbduk_handle* h, const uint32_t* codes, const uint32_t* undef, const int64_t* offsets, int64_t n, int32_t paired,
                                          int32_t* out_found, int32_t* out_id, uint8_t* out_flags) {
    return host_batch(h, 1, reinterpret_cast<const uint8_t*>(codes), offsets, n, paired, out_found, out_id, out_flags, undef, true);
}

extern "C" int bbduk_table_lookup(bbduk_handle* h, const int64_t* keys, int64_t n, int32_t* out_ids) {
    if (!h || n < 0 || (n > 0 && (!keys || !out_ids))) return fail(h, BBDUK_ERR_ARG, "bad argument");
    if (!h->finalized) return fail(h, BBDUK_ERR_STATE, "table not finalized");
    if (n == 0) return BBDUK_OK;
    std::lock_guard<std::mutex> g(h->mu);
    HIP_TRY(h, hipSetDevice(h->p.device));
    int64_t* dk = nullptr; int32_t* dv = nullptr;
    HIP_TRY(h, hipMalloc(&dk, (size_t)n * sizeof(int64_t)));
    if (hipMalloc(&dv, (size_t)n * sizeof(int32_t)) != hipSuccess) { hipFree(dk); return fail(h, BBDUK_ERR_NOMEM, "hipMalloc"); }
    hipMemcpy(dk, keys, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice);
    const KParams K = make_kparams(h);
    hipLaunchKernelGGL(bbduk_lookup_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, K, dk, n, dv);
    hipError_t e = hipStreamSynchronize(h->stream);
    if (e == hipSuccess) e = hipMemcpy(out_ids, dv, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost);
    hipFree(dk); hipFree(dv);
    if (e != hipSuccess) { h->err = hipGetErrorString(e); return BBDUK_ERR_DEVICE; }
    return BBDUK_OK;
}

extern "C" int bbduk_kernel_time_ms(bbduk_handle* h, int32_t last_k, float* avg_ms) {
    if (!h || !avg_ms || last_k < 1) return BBDUK_ERR_ARG;
    const int64_t have = std::min<int64_t>(h->evCount, bbduk_handle::EV_RING);
    const int64_t k = std::min<int64_t>(last_k, have);
    if (k < 1) return fail(h, BBDUK_ERR_STATE, "no launch recorded");
    double sum = 0;
    for (int64_t q = 0; q < k; q++) {
        const int evi = (int)((h->evCount - 1 - q) % bbduk_handle::EV_RING);
        HIP_TRY(h, hipEventSynchronize(h->ev1[evi]));
        float ms = 0; HIP_TRY(h, hipEventElapsedTime(&ms, h->ev0[evi], h->ev1[evi]));
        sum += ms;
    }
    *avg_ms = (float)(sum / (double)k);
    return BBDUK_OK;
}

extern "C" int bbduk_counters_len(const bbduk_handle* h) { return h ? BBDUK_NCOUNTERS + 2 * h->p.numScaffolds : BBDUK_ERR_ARG; }
extern "C" int bbduk_get_counters(bbduk_handle* h, int64_t* out, int32_t n) {
    if (!h || !out || n != bbduk_counters_len(h)) return fail(h, BBDUK_ERR_ARG, "bad argument");
    std::lock_guard<std::mutex> g(h->mu);
    HIP_TRY(h, hipSetDevice(h->p.device));
    HIP_TRY(h, hipMemcpy(out, h->d_counters, (size_t)n * sizeof(int64_t), hipMemcpyDeviceToHost));
    return BBDUK_OK;
}
extern "C" int bbduk_reset_counters(bbduk_handle* h) {
    if (!h) return BBDUK_ERR_ARG;
    std::lock_guard<std::mutex> g(h->mu);
    HIP_TRY(h, hipSetDevice(h->p.device));
    HIP_TRY(h, hipMemset(h->d_counters, 0, (size_t)bbduk_counters_len(h) * sizeof(int64_t)));
    // hipMemset on device memory returns before the fill has run, and the null stream does not order against the operators' non-blocking streams:
    // a batch submitted right behind this call could add its counters and THEN be zeroed (round 5: seen as lost counts under a loaded GPU once a batch
    // with long units ran three kernels instead of one).  The counters are zero when this returns.
    HIP_TRY(h, hipStreamSynchronize(nullptr));
    return BBDUK_OK;
}

// ---- synthetic generator
static bool synth_ok(const bbduk_synth_params* sp) {
    return sp && sp->read_len > 0 && sp->ins_min > 0 && sp->ins_max >= sp->ins_min && sp->adapter1_len >= 0 && sp->adapter2_len >= 0 &&
           (sp->adapter1_len == 0 || sp->adapter1) && (sp->adapter2_len == 0 || sp->adapter2) && (sp->contam_len == 0 || sp->contam);
}
static bb_synth_dev to_dev(const bbduk_synth_params* sp) {
    bb_synth_dev d;
    d.seed = sp->seed; d.read_len = sp->read_len; d.ins_min = sp->ins_min; d.ins_max = sp->ins_max;
    d.adapter1_len = sp->adapter1_len; d.adapter2_len = sp->adapter2_len;
    d.sub_rate_q32 = sp->sub_rate_q32; d.n_rate_q32 = sp->n_rate_q32; d.contam_frac_q32 = sp->contam_frac_q32;
    d.contam_len = sp->contam_len; d.adapter1 = sp->adapter1; d.adapter2 = sp->adapter2; d.contam = sp->contam;
    return d;
}

extern "C" int bbduk_synth_generate_host(const bbduk_synth_params* sp, int64_t first_pair, int64_t n_pairs, uint8_t* bases, int64_t* offsets) {
    if (!synth_ok(sp) || n_pairs < 0 || !offsets || (n_pairs > 0 && !bases)) return BBDUK_ERR_ARG;
    const bb_synth_dev d = to_dev(sp);
    for (int64_t p = 0; p < n_pairs; p++) {
        const bb_pair_hdr h = bb_synth_pair_header(d, (uint64_t)(first_pair + p));
        for (int mate = 0; mate < 2; mate++) {
            uint8_t* dst = bases + (2 * p + mate) * (int64_t)d.read_len;
            for (int j = 0; j < d.read_len; j++) dst[j] = bb_synth_read_base(d, (uint64_t)(first_pair + p), h, mate, j);
        }
    }
    for (int64_t r = 0; r <= 2 * n_pairs; r++) offsets[r] = r * (int64_t)d.read_len;
    return BBDUK_OK;
}

extern "C" int bbduk_synth_pair_inserts(const bbduk_synth_params* sp, int64_t first_pair, int64_t n_pairs, int32_t* out_insert) {
    if (!synth_ok(sp) || n_pairs < 0 || (n_pairs > 0 && !out_insert)) return BBDUK_ERR_ARG;
    const bb_synth_dev d = to_dev(sp);
    for (int64_t p = 0; p < n_pairs; p++) out_insert[p] = bb_synth_pair_header(d, (uint64_t)(first_pair + p)).ins;
    return BBDUK_OK;
}

extern "C" int bbduk_synth_generate_device(const bbduk_synth_params* sp, int64_t first_pair, int64_t n_pairs,
                                           uint8_t* d_bases, int64_t* d_offsets, int32_t device, void* stream) {
    if (!synth_ok(sp) || n_pairs < 0 || !d_offsets || (n_pairs > 0 && !d_bases)) return BBDUK_ERR_ARG;
    if (hipSetDevice(device) != hipSuccess) return BBDUK_ERR_DEVICE;
    bb_synth_dev d = to_dev(sp);
    uint8_t *da1 = nullptr, *da2 = nullptr, *dc = nullptr;
    hipStream_t st = (hipStream_t)stream;
    auto up = [&](const uint8_t* src, int64_t len, uint8_t** dst) -> bool {
        if (len <= 0) { *dst = nullptr; return true; }
        if (hipMalloc(dst, (size_t)len) != hipSuccess) return false;
        return hipMemcpy(*dst, src, (size_t)len, hipMemcpyHostToDevice) == hipSuccess;
    };
    bool ok = up(sp->adapter1, sp->adapter1_len, &da1) && up(sp->adapter2, sp->adapter2_len, &da2) && up(sp->contam, sp->contam_len, &dc);
    int rc = BBDUK_OK;
    if (ok) {
        d.adapter1 = da1; d.adapter2 = da2; d.contam = dc;
        const int64_t total = std::max<int64_t>(n_pairs * 2LL * d.read_len, 2 * n_pairs + 1);
        const int64_t blocks = std::min<int64_t>((total + 255) / 256, 1 << 20);
        hipLaunchKernelGGL(bbduk_synth_kernel, dim3((unsigned)blocks), dim3(256), 0, st, d, first_pair, n_pairs, d_bases, d_offsets);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) rc = BBDUK_ERR_DEVICE;
    } else rc = BBDUK_ERR_DEVICE;
    hipFree(da1); hipFree(da2); hipFree(dc);
    return rc;
}

// ---- jgi/Seal.java on the same core (include/seal_gpu.h)
#include "bbduk_seal.inc"
