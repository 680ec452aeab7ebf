// bbduk_internal.h -- shared by the translation units of libbbduk_hip.so (not part of the ABI).
#ifndef BBDUK_INTERNAL_H
#define BBDUK_INTERNAL_H
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include <mutex>
#include <condition_variable>
#include <memory>
#include "../../include/bbduk_gpu.h"

struct KParams {
    int32_t mode, k, mink, rcomp, forbidNs, minlen, minlen2, qhdist, qhdist2, maxBadKmers, minReadLength;
    float   minLenFraction;
    int32_t rieb, trimPad, ktrimExclusive, restrictLeft, restrictRight, skipR1, skipR2, numScaffolds, useShort;
    int32_t tpe, qskip, speed;   // trimpairsevenly; qskip (<2 = off) and speed (0 = off) run in the general kernels only
    float   mkf, mcf;            // kfilter: minkmerfraction / mincoveredfraction (general kernels only)
    int32_t mfc;                 // kmaskfullycovered (ktrim=n): mask only bases all of whose covering k-mers match
    int32_t tf;                  // trimfailuresto1bp (rieb is then off): see tf1bp()
    int32_t route, wunitMax;     // route != 0 (ktrim=r|l, kfilter; round 5): every UNIT goes to the kernel that holds it -- the wave kernel takes the units of up to
                                 // wunitMax bases, the tiled kernel the ones up to CAP_BASES - 64, the chunked long-unit kernel the rest -- instead of the whole batch
                                 // following its longest unit (one 60 kb read among 8 M short ones: 522 -> 32 Gbases/s)
    int32_t kbig, fbm;           // kfilter variants of bbduk_kscan_kernel: k>31 emulation (kbig>k), findBestMatch
    int32_t* matchN; int32_t* matchIds; int32_t* matchCnt; int32_t matchCap;    // findBestMatch's idList / countList per read (rename, :2508-2522), or null
    int32_t* outLeft; int32_t* outRight;   // ksplit: the span it computes per read (leftmost, rightmost), or -1, -1
    uint32_t* outMask;          // ktrim=n on the wave kernel: the per-base mask, bit offsets[read]+b
    int32_t waveFirst;          // the tiled secondary kernel runs only if the span pre-pass found a read beyond a wave's planes
    const uint32_t* undef;      // packed input (bbduk_*_batch_packed): 1 bit per base, set = undefined; `bases` then points at
                                // 2-bit codes, 16 bases per 32-bit word (A0 C1 G2 T/U3, undefined 0).  nullptr = ASCII bases
    uint64_t mask, kmask, middleMask;
    // The map, device layout: 4-way buckets.  tags[b] packs four 15-bit fingerprints in 16-bit lanes (0 = free
    // way) plus the bucket's continuation flag in bit 63; the full key and its id live together in bkv[4*b+way]
    // (16 bytes: one fetch verifies the key and yields the id) and are touched only when a fingerprint matches.
    // A key sits in the first bucket >= its home bucket that had a free way (bucket-granular linear probing);
    // every full bucket it passed gets the continuation flag, so a query stops at the first unflagged bucket.
    // One 8-byte gather answers almost every absent k-mer.
    const uint64_t* tags;
    const uint4*    bkv;        // {key lo, key hi, id, 0}
    uint32_t bucketMask;
    int32_t  bucketBits;
    int64_t  storedKmers;
    // The HBM-resident layout for maps far beyond the caches (BASELINE configs[3]; `big` != 0): the same 4-way tag words, eight of
    // them (32 slots) per 64-byte line; a full-length key's LINE comes from the minimum over its gapped (gm+gm)-mers of a strand-
    // symmetric hash -- consecutive k-mers of a read share it, so they share one HBM sector -- its WORD in the line from mix_b, its
    // fingerprint from mix_a; probing walks the line's words cyclically, then the next line.  Keys and ids live in two slot-parallel
    // arrays (8 + 2|4 bytes) touched only on a fingerprint match.  See "big layout" in bbduk_hip.hip.
    int32_t  big;
    const uint64_t* bigTags;    // [16 * bigLines] tag words (128-byte lines, round 6; round 2's form behind the 52-bit hook: 8); `tags` / `bkv` above then hold the secondary map of the spilled keys
    const uint64_t* bigKeys;    // [64 * bigLines], EMPTY_KEY = free
    const void*     bigIds;     // uint16 or uint32 per slot
    int32_t  bigIdBytes;
    uint32_t bigLines;
    int32_t  seed, seedHl, seedHr, seedM;   // the seed layout (bbduk_seed.inc): tags / bkv hold PARENTS under their left (seedHl bases) and right (seedHr) halves
    int32_t  gSib;              // a full pair overflows into the sibling pair of the line's other half before anything is spilled (0: Seal's maps, round 2's form)
    int32_t  gLb;               // log2 of the tag words per line: 4 = 128-byte lines (round 6), 3 = 64-byte lines (Seal's maps; round 2's form)
    int32_t  gm, gW, gH, gD;    // gapped minimizer: m bases from each half, W candidates, half length H, right half starts at D = k-H
    // Query-side Hamming expansion precomputed (round 5; bbduk_hip.hip: qx_rewrite).  qx != 0: tags / bkv above hold the EXPANSION -- every forward k-mer X for
    // which getValue(X, rc X, qhdist) finds something, keyed as it stands (so the lookups run with rcomp = 0), bkv.z = getValue's answer, bkv.w = the answer
    // of its neighbour loop alone -- and the map the reference holds sits here, for the windows whose rolling rkmer is not kmer's reverse complement.
    // Round 6: with a middle mask the expansion is keyed by the MASKED forward k-mer (what the kernels look up anyway); an entry whose answer would depend on
    // the masked base(s) -- the four fillings disagree: needs a k-mer whose halves mirror each other -- holds z = w = -2 and is evaluated exactly (qx_exact).
    int32_t  qx; const uint64_t* qxTags; const uint4* qxBkv; uint32_t qxBucketMask; int32_t qxBucketBits;
    int32_t  qxQh, qxQh2;       // the handle's own qhdist / qhdist2 (KParams::qhdist is 0 on such a handle: nothing left to expand)
    int32_t  gV32;              // the line function's variant: 1 = 32-bit values (gap_v32 / gap_line32: maps of up to 2^31 keys), 2 = wide values (gap_v52: beyond), 0 = round 2's 52-bit minima; bbduk_bigs.inc scans 1 and 2
    // presence filter in front of the map: one bit per hash slot, copied into LDS by every workgroup
    const uint32_t* ldsImage;   // HBM copy of the LDS bitmap (2^ldsBits bits); 0 bits = absent
    int32_t  ldsBits;
    unsigned long long* status; // where a kernel reports a device-side error code; nullptr = counters[BBDUK_CTR_STATUS] (the *_device operators).
                                // The host-buffer operators give every staging slot a word of its own: two submitting threads share the counters.
    int32_t  dbg;               // timing mask (bbduk_test_hook): timing experiments only (results become wrong); needs -DBBDUK_TIMING_SWITCHES
};

struct bbduk_comm;                   // bbduk_comm.cpp: the RCCL communicator(s) a handle belongs to

struct bbduk_handle {
    bbduk_params p;
    std::string err; std::mutex errMu;                           // (several submitters may fail at once: writes and bbduk_last_error copy under errMu)
    std::mutex mu;
    bool finalized = false;
    std::vector<int64_t> hkeys;          // staged (key,value) pairs before finalize
    std::vector<int32_t> hvals;
    int64_t nkeys = 0;
    int64_t nkeysRef = 0;                // seed layout with a twin: the reference's distinct key count (nkeys counts records there)
    uint64_t* d_tags = nullptr; uint4* d_bkv = nullptr; uint64_t nbuckets = 0; int bucketBits = 0;
    // A big-layout map of 2^20..2^25 keys keeps a cache-resident map of the same keys beside it (bbduk_hip.hip: build_both): batches with units
    // beyond a wave's planes run the tile / long-read kernels, whose big-layout instantiations look every key up on its own
    uint64_t* d_tagsAlt = nullptr; uint4* d_bkvAlt = nullptr; uint32_t* d_ldsAlt = nullptr; uint64_t nbucketsAlt = 0; int bucketBitsAlt = 0, ldsBitsAlt = 0; bool hasAlt = false;
    // big layout (HBM-resident maps): 8 * bigLines tag words, slot-parallel keys / ids; d_tags / d_bkv = the secondary map of the spilled keys
    bool big = false; uint64_t* d_bigTags = nullptr; uint64_t* d_bigKeys = nullptr; void* d_bigIds = nullptr; int bigIdBytes = 0; uint32_t bigLines = 0;
    int gm = 0, gW = 0, gH = 0, gD = 0; int64_t nspilled = 0; int gV32 = 0; int gLb = 4; int gSib = 1;
    // qhdist = 1 handles (round 5): the expansion is the map the kernels look up, the reference's own map is kept beside it (KParams::qx)
    bool qx = false; uint64_t* d_tagsQx = nullptr; uint4* d_bkvQx = nullptr; uint64_t nbucketsQx = 0; int bucketBitsQx = 0; int64_t nkeysQx = 0;
    bool seed = false; int seedHl = 0, seedHr = 0, seedM = 0;   // seed layout: parents only, under their halves (large hdist=1 maps built on the device)
    double expectShort = 0.0;            // short k-mers of mink the next build will see (they live in the secondary map of a big-layout map)
    bool bigPlain = false;               // lines by a plain key hash instead of the gapped minimizer (gW = 0)
    bool sealTable = false;              // the map of a seal_handle: record ids may be SEAL_MULTI | offset; always the cache-resident layout
    // streaming device-side build (bbduk_build_begin / _add_device / _end)
    struct BuildState* build = nullptr;
    uint32_t* d_ldsImage = nullptr; int ldsBits = 0;
    int* d_slowFlag = nullptr;
    static const int EV_RING = 64;                                // HIP events around the dominant kernel of the last launches
    hipEvent_t ev0[EV_RING] = {}, ev1[EV_RING] = {}; int64_t evCount = 0;
    hipEvent_t evDone[EV_RING] = {};                              // behind the LAST kernel of the launch that holds ring entry q (its flag block: ring_acquire)
    std::mutex launchMu;                                          // device-buffer operators may be issued from several host threads /
                                                                  // streams at once: slot choice and enqueue of one launch are atomic,
                                                                  // and every launch in flight has its own pre-pass flag (d_slowFlag[slot])
    // host-operator staging: two slots, each with its own device buffers and stream, so that two submitting host threads overlap --
    // the H2D copy of one call runs under the kernel and the D2H copy of the other (the copy engines and the CUs are different units)
    struct Slot {
        uint8_t* d_bases = nullptr; size_t cap_bases = 0;
        uint8_t* d_undef = nullptr; size_t cap_undef = 0;     // packed boundary: one undefined-base bit per base
        int64_t* d_off = nullptr;   size_t cap_reads = 0;
        int32_t* d_a = nullptr; int32_t* d_id = nullptr; uint8_t* d_fl = nullptr;
        int64_t* d_status = nullptr;                             // this slot's device-side error word (see KParams::status)
        hipStream_t stream = nullptr; bool busy = false;
        hipStream_t copyStream = nullptr; hipEvent_t evPiece[8] = {};  // large calls: the next piece's upload runs under this piece's kernel (host_batch)
    };
    static const int NSLOTS = 2;
    Slot slot[NSLOTS];
    std::mutex slotMu; std::condition_variable slotCv;
    int64_t* d_counters = nullptr;
    hipStream_t stream = nullptr;                                 // table builds, counters, the one-off operators
    int numCU = 256;
    bbduk_comm* comm = nullptr;          // set by bbduk_comm_create / bbduk_comm_create_local
    // include/bbduk_test_hooks.h (tests and experiments only)
    bool hookForceTile = false, hookBigLayout = false, hookNoBigLayout = false, hookPairScan = false, hookSeedLayout = false, hookBig52 = false, hookBigWide = false; int hookBucketBits = 0, hookLdsBits = -1, hookDbg = 0, hookBigLoad = 0;
};

#define HIP_TRY(h, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    { std::lock_guard<std::mutex> lg_(( h)->errMu); (h)->err = std::string(#call) + ": " + hipGetErrorString(e_); } return BBDUK_ERR_DEVICE; } } while (0)

int bbduk_comm_allreduce_i64(bbduk_handle* h, int64_t* d_buf, int64_t n, void* stream);   // bbduk_comm.hip

static inline int fail(bbduk_handle* h, int code, const char* msg) { if (h) { std::lock_guard<std::mutex> lg(h->errMu); h->err = msg; } return code; }

#endif
