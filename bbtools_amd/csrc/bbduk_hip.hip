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

#ifndef BLOCK_THREADS
#define BLOCK_THREADS   1024
#endif
#define NWAVES          (BLOCK_THREADS / 64)
#ifdef BBDUK_AB_NO_TF
#define PTF 0
#else
#define PTF P.tf
#endif
#define TILE_READS      256                    // reads per tile (even: whole pairs)
#define CAP_BASES       40960                  // LDS plane capacity in bases (>= 2*BBDUK_MAX_READ_LEN + 32: a pair fits)
#define CAP_CHUNKS      (CAP_BASES / 16)
#define PLANE_PAD       12                     // words of slack on both sides of the 2-bit planes
#define EMPTY_KEY       0xFFFFFFFFFFFFFFFFULL  // keys are < 2^63
#ifndef MAX_LDS_BITS
#define MAX_LDS_BITS    20                     // 128 KiB presence filter per workgroup
#endif
#define BIGLOC          999999999

// Deletion experiments (profiles/ab.sh): a build with -DBBDUK_TIMING_SWITCHES honours bbduk_test_hook(BBDUK_HOOK_TIMING_MASK, n) and skips one
// or more stages of the scan (results become wrong).  Production builds compile the switches out.
#ifdef BBDUK_TIMING_SWITCHES
#define TSW(P, n) ((((P).dbg) >> (n)) & 1)      /* the timing mask: bit n deletes stage n */
#else
#define TSW(P, n) false
#endif


// --------------------------------------------------------------------------------------------------
// device helpers

// reverseComplementBinaryFast(long,int) (dna/AminoAcid.java:585-601): complement, reverse 2-bit groups, right-align
__device__ __forceinline__ uint64_t dev_rcomp(uint64_t kmer, int len) {
    uint64_t x = __brevll(~kmer);
    x = ((x >> 1) & 0x5555555555555555ULL) | ((x & 0x5555555555555555ULL) << 1);
    return x >> (64 - 2 * len);
}

// Two independent 32-bit multiplicative mixes of a key (< 2^63): lo*A0 + hi*A1 with odd constants.  On gfx950
// v_mul_lo_u32 issues at (almost) the rate of a simple integer op (profiles/ubench/valu_rate.hip: 2.6 vs 2.4
// cycles per wave-instruction), so two multiplies and an add beat the three 24-bit pieces used before.  The top
// bits of each sum are well mixed.  mix_a feeds the LDS presence filter (word index from its top bits, bit index
// from its low 5 bits) and the 15-bit fingerprint (bits 16-30); mix_b feeds the bucket index (top bits).
#define HA0 0x9E3779B1u
#define HA1 0x85EBCA6Bu
#define HB0 0xC2B2AE35u
#define HB1 0x27D4EB2Fu
// Both are taken of a key's VALUE, i.e. the key without its length bit (the scans have the value first and would pay
// two more instructions per position to hash the finished key); strip_len recovers the value from a stored key.
__host__ __device__ __forceinline__ uint32_t mix_a(uint64_t value) { return (uint32_t)value * HA0 + (uint32_t)(value >> 32) * HA1; }
__host__ __device__ __forceinline__ uint32_t mix_b(uint64_t value) { return (uint32_t)value * HB0 + (uint32_t)(value >> 32) * HB1; }
__host__ __device__ __forceinline__ uint64_t strip_len(uint64_t key) {      // key = value | 1<<2*len, value < 1<<2*len
#if defined(__HIP_DEVICE_COMPILE__)
    return key ? key ^ (1ULL << (63 - __clzll((long long)key))) : 0ULL;
#else
    return key ? key ^ (1ULL << (63 - __builtin_clzll(key))) : 0ULL;
#endif
}
__host__ __device__ __forceinline__ uint32_t bucket_of(uint32_t mb, int bucketBits) { return mb >> (32 - bucketBits); }
// 15-bit fingerprint (0 is a legal value: a free way's lane also reads 0, so a query whose fingerprint is 0 sees
// free ways as candidates and the key check rejects them).  Bit 63 of a bucket's tag word is its continuation flag:
// some key found this bucket full and was placed further along, so an unmatched lookup goes on to the next bucket.
__host__ __device__ __forceinline__ uint32_t tag_of(uint32_t ma) { return (ma >> 16) & 0x7FFFu; }
// LDS presence filter of 2^ldsBits bits: byte address of the word, and the bit inside it (low 5 bits of ma; shifts
// use only those bits of their count)
__host__ __device__ __forceinline__ uint32_t filt_byte(uint32_t ma, int ldsBits) { return (ma >> (35 - ldsBits)) & ~3u; }
__device__ __forceinline__ uint32_t filt_test(const uint32_t* s_filt, uint32_t ma, int ldsBits) {
    const uint32_t w = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(s_filt) + filt_byte(ma, ldsBits));
    return (w >> (ma & 31u)) & 1u;
}
#define TAG_CONT  0x8000000000000000ULL
#define TAG_FPS15 0x7FFF7FFF7FFF7FFFULL           /* big layout: the four fingerprints without the lanes' top bits (spill bits) */
#define TAG_TOPS  0x8000800080008000ULL
#define TAG_FPS   0x7FFFFFFFFFFFFFFFULL

// 0x8000 in every 16-bit lane of v that is zero (may also flag lanes above a true zero lane: callers verify)
__device__ __forceinline__ uint64_t zero16(uint64_t v) { return (v - 0x0001000100010001ULL) & ~v & 0x8000800080008000ULL; }

// ---- big layout (HBM-resident maps, KParams::big) -------------------------------------------------------------------
// Where a key lives must be a function of the key alone (the build sees only keys), yet consecutive k-mers of a read
// should land in the same 64-byte HBM sector.  A k-mer key is max(kmer, rkmer) with its middle base(s) masked
// (BBDukIndexMod.java:532-544), so its two clean halves of H = (k - midMaskLen)/2 bases are all it has in common with its
// neighbours.  Candidates: the W = H-m+1 gapped (m+m)-mers  G(p) = key[p, p+m) ++ key[p+D, p+D+m),  D = k-H  (one m-mer in
// each half, both clear of the masked middle).  The reverse complement of the key holds rc(G(p)) at position H-m-p, so
//     h(p) = gap_f(G(p)) + gap_f(rc(G(p)))
// is the same multiset whichever strand became the key, and  hmin = min_p h(p)  picks the key's line.  Windows i and i+1 of a
// read share W-1 of their candidates: a read of 120 31-mers touches ~31 lines instead of 120 (m = 9, W = 7).
// Per-orientation mix of a gapped-mer given as its two m-mers (m <= 12: they fit 24 bits), then an avalanche: the ORDER of
// these values decides which candidate wins, so they should look random.
__host__ __device__ __forceinline__ uint32_t gap_f(uint32_t l, uint32_t r) {
    uint32_t x = l * 0x9E3779B1u + r * 0x85EBCA6Bu;
    x ^= x >> 15; x *= 0x2C1B3C6Du;                               // the product's high bits carry the order
    return x;
}
// The two orientations' values combine symmetrically into 52 bits, their sum below and the top of their product above (sum and
// xor would be linearly related bit by bit: measured, the line hash then behaved like a 30-bit one and whole runs shared lines at
// 10^10 keys).  The candidates are ordered by these values.
__host__ __device__ __forceinline__ uint64_t gap_pair(const uint32_t fa, const uint32_t fb) {
    return ((uint64_t)((fa * fb) >> 12) << 32) | (uint32_t)(fa + fb);
}
// minima are biased towards 0: scramble before the multiply-shift that maps onto [0, nlines)
__host__ __device__ __forceinline__ uint32_t gap_line(const uint64_t hmin, const uint32_t nlines) {
    uint32_t y = (uint32_t)hmin * 0x297A2D39u; y ^= y >> 15;
    y = y * 0xC2B2AE35u + (uint32_t)(hmin >> 32) * 0x9E3779B1u; y ^= y >> 13; y *= 0x85EBCA6Bu;
    return (uint32_t)(((uint64_t)y * (uint64_t)nlines) >> 32);
}
__host__ __device__ __forceinline__ uint64_t rcomp_hd(uint64_t kmer, int len) {      // dev_rcomp, host and device
    uint64_t x = ~kmer;
    x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
    x = ((x >> 8) & 0x00FF00FF00FF00FFULL) | ((x & 0x00FF00FF00FF00FFULL) << 8);
    x = ((x >> 16) & 0x0000FFFF0000FFFFULL) | ((x & 0x0000FFFF0000FFFFULL) << 16);
    x = (x >> 32) | (x << 32);
    return x >> (64 - 2 * len);
}
struct BigGeom { int32_t k, m, W, H, D; uint32_t nlines; uint64_t middleMask; };
// hmin of a full-length key value (the slow, exact form: every kernel's generic path, the build, the host).  A = the value,
// B = its reverse complement with the same middle mask: {A, B} = {kmer & mm, rkmer & mm} whichever one the key is.
__host__ __device__ __forceinline__ uint64_t gap_hmin_value(const BigGeom& G, const uint64_t A) {
    const uint64_t B = rcomp_hd(A, G.k) & G.middleMask;
    const uint32_t mk = (1u << (2 * G.m)) - 1u;
    uint64_t best = ~0ULL;
    for (int p = 0; p < G.W; p++) {
        const int q = G.H - G.m - p;
        const uint32_t la = (uint32_t)(A >> (2 * (G.k - G.m - p))) & mk, ra = (uint32_t)(A >> (2 * (G.k - G.D - G.m - p))) & mk;
        const uint32_t lb = (uint32_t)(B >> (2 * (G.k - G.m - q))) & mk, rb = (uint32_t)(B >> (2 * (G.k - G.D - G.m - q))) & mk;
        const uint64_t h = gap_pair(gap_f(la, ra), gap_f(lb, rb));
        best = h < best ? h : best;
    }
    return best;
}
// line of any key (with its length bit): full-length keys by their gapped minimizer, the short k-mers of mink (other lengths)
// by a plain hash
__host__ __device__ __forceinline__ uint32_t big_line_of_key(const BigGeom& G, const uint64_t key, const uint32_t ma) {
    const uint64_t v = strip_len(key);
    // G.W == 0: lines by a plain hash of the key (maps whose keys crowd on few minimizers: reference-side Hamming neighbourhoods)
    const uint64_t h = ((key >> (2 * G.k)) == 1ULL && G.W > 0) ? gap_hmin_value(G, v) : (uint64_t)(ma ^ 0x5BD1E995u);
    return gap_line(h, G.nlines);
}
// A key has two words in its line: the primary (top 3 bits of mix_b) and an alternate (the next 3 bits, made distinct).  It lives
// in the first free way of the primary, else of the alternate; if both are full it is SPILLED into the secondary map (the
// cache-resident layout's buckets: KParams::tags / bkv) and the primary word gets one of its four spill bits.  So a lookup is: both
// words (one 64-byte sector), and only if the primary carries the key's spill bit one more gather.  No probe chains: an overloaded line costs its
// absent keys nothing more.  (~6 % of the keys spill at 0.6 keys per slot, ~9 % of the words carry the flag.)
// The lanes' top bits of a tag word form a 4-bit filter over the keys spilled from it (the word being their primary): a spilled key
// sets bit 16*j+15, j = two hash bits of its own; a lookup goes to the secondary map only if ITS bit is set.
__host__ __device__ __forceinline__ int spill_bit(const uint32_t ma) { return 16 * (int)((ma >> 13) & 3u) + 15; }
__host__ __device__ __forceinline__ void big_words(const uint32_t line, const uint32_t mb, uint32_t& w1, uint32_t& w2) {
    const uint32_t a = mb >> 29; uint32_t b = (mb >> 26) & 7u;
    b = (b == a) ? (b ^ 1u) : b;
    w1 = 8u * line + a; w2 = 8u * line + b;
}
__device__ __forceinline__ BigGeom big_geom(const KParams& P) { BigGeom G; G.k = P.k; G.m = P.gm; G.W = P.gW; G.H = P.gH; G.D = P.gD; G.nlines = P.bigLines; G.middleMask = P.middleMask; return G; }
__device__ __forceinline__ int big_id_at(const KParams& P, const uint64_t slot) {
    return P.bigIdBytes == 2 ? (int)reinterpret_cast<const uint16_t*>(P.bigIds)[slot] : (int)reinterpret_cast<const uint32_t*>(P.bigIds)[slot];
}
__device__ __forceinline__ int table_find_t(const KParams& P, uint64_t key, uint32_t ma, uint32_t mb, uint64_t t0);
// exact lookup given the key's two words and their tags
__device__ __forceinline__ int big_find_in(const KParams& P, const uint64_t key, const uint32_t ma, const uint32_t mb,
                                           const uint32_t w1, const uint32_t w2, const uint64_t t1, const uint64_t t2) {
    const uint64_t pat = (uint64_t)tag_of(ma) * 0x0001000100010001ULL;
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const uint64_t t = q ? t2 : t1; const uint32_t word = q ? w2 : w1;
        uint64_t cand = zero16((t & TAG_FPS15) ^ pat);
        while (cand) {
            const int way = (__ffsll((unsigned long long)cand) - 1) >> 4;
            if (P.bigKeys[4ULL * word + way] == key) return big_id_at(P, 4ULL * word + way);
            cand &= cand - 1;
        }
    }
    if (!((t1 >> spill_bit(ma)) & 1ULL)) return -1;                // no key of this key's kind was ever spilled from the primary word
    return table_find_t(P, key, ma, mb, P.tags[bucket_of(mb, P.bucketBits)]);
}
// The exact per-key lookup.  Only the kernels that serve big-layout maps contain it (bbduk_wave_kernel<.., BIG>, the BIG instantiations
// of its tile / long-read fallbacks, the point-lookup test hook): threaded through every general kernel's probe sites it cost those
// kernels 100-500 spilled VGPRs and half their speed, so the big layout is chosen only for the configurations these kernels run.
__device__ __forceinline__ int big_find(const KParams& P, const uint64_t key, const uint32_t ma, const uint32_t mb) {
    uint32_t w1, w2;
    big_words(big_line_of_key(big_geom(P), key, ma), mb, w1, w2);
    return big_find_in(P, key, ma, mb, w1, w2, P.bigTags[w1], P.bigTags[w2]);
}

// map lookup with the mixes already computed: id (>0) of the key, or -1.  t0 = tags of the home bucket.
__device__ __forceinline__ int table_find_t(const KParams& P, uint64_t key, uint32_t ma, uint32_t mb, uint64_t t0) {
    uint32_t b = bucket_of(mb, P.bucketBits);
    const uint64_t pat = (uint64_t)tag_of(ma) * 0x0001000100010001ULL;
    uint64_t t = t0;
    for (;;) {
        uint64_t cand = zero16((t & TAG_FPS) ^ pat);
        while (cand) {
            const int way = (__ffsll((unsigned long long)cand) - 1) >> 4;
            const uint4 kv = P.bkv[4ULL * b + way];
            if ((((uint64_t)kv.y << 32) | kv.x) == key) return (int)kv.z;
            cand &= cand - 1;
        }
        if (!(t & TAG_CONT)) return -1;           // nothing ever overflowed from here: the key cannot be further along
        b = (b + 1) & P.bucketMask;
        t = P.tags[b];
    }
}
__device__ __forceinline__ int table_find_m(const KParams& P, uint64_t key, uint32_t ma, uint32_t mb) {
    return table_find_t(P, key, ma, mb, P.tags[bucket_of(mb, P.bucketBits)]);
}
__device__ __forceinline__ int table_get(const KParams& P, uint64_t key) { const uint64_t v = strip_len(key); return table_find_m(P, key, mix_a(v), mix_b(v)); }
// A lookup result ("ref") is -1 = absent or the id (>0) itself.
__device__ __forceinline__ int ref_to_id(const KParams& P, int ref) { return ref; }

// passesSpeed (bbduk/BBDukIndexMod.java:562): with this index the gate sits on the query side only, so it has to
// be applied here (keys that fail it ARE in the map).  General kernels only.
__device__ __forceinline__ bool passes_speed(const KParams& P, uint64_t key) { return P.speed < 1 || (int)(key % 17ULL) >= P.speed; }

// key -> ref through the cascade: LDS presence bit -> bucket fingerprints -> key.  `ok` = lane has a real query.
__device__ __forceinline__ int probe_ref(const KParams& P, const uint32_t* s_filt, uint64_t value, uint64_t lengthMask, bool ok) {
    const uint32_t ma = mix_a(value);
    bool p = ok;
    if (P.speed > 0) p = p && passes_speed(P, value | lengthMask);
    if (P.ldsBits) p = p & (bool)filt_test(s_filt, ma, P.ldsBits);
    int ref = -1;
    if (p) ref = table_find_m(P, value | lengthMask, ma, mix_b(value));
    return ref;
}

// getValueInner (bbduk/BBDukIndexMod.java:492-520): canonicalise, mask middle, add length bit
template <bool GENERAL>
__device__ __forceinline__ uint64_t make_value(const KParams& P, uint64_t kmer, uint64_t rkmer) {
    // values < 2^62: unsigned max == Java's signed Tools.max
    const uint64_t mx = (!GENERAL || P.rcomp) ? (kmer > rkmer ? kmer : rkmer) : kmer;
    return mx & P.middleMask;
}
template <bool GENERAL>
__device__ __forceinline__ uint64_t make_key(const KParams& P, uint64_t kmer, uint64_t rkmer, uint64_t lengthMask) {
    return make_value<GENERAL>(P, kmer, rkmer) | lengthMask;
}

// getValue (bbduk/BBDukIndexMod.java:462-481): query-side Hamming expansion, same (j,i) order, first id>=1 wins
template <int D>
__device__ int get_value(const KParams& P, uint64_t kmer, uint64_t rkmer, uint64_t lengthMask, int len, int qh) {
    const uint64_t key0 = make_key<true>(P, kmer, rkmer, lengthMask);
    int id = passes_speed(P, key0) ? table_get(P, key0) : -1;
    if constexpr (D > 0) {
        if (id < 1 && qh > 0) {
            for (int j = 0; j < 4 && id < 1; j++) {
                for (int i = 0; i < len && id < 1; i++) {
                    const uint64_t temp = (kmer & ~(3ULL << (2 * i))) | ((uint64_t)j << (2 * i));
                    if (temp != kmer) id = get_value<D - 1>(P, temp, dev_rcomp(temp, len), lengthMask, len, qh - 1);
                }
            }
        }
    }
    return id;
}
// qhdist 3 has its own (deeper, fatter) body so that the usual qhdist <= 2 callers keep their register budget
__device__ __noinline__ int get_value_expand3(const KParams& P, uint64_t kmer, uint64_t rkmer, uint64_t lengthMask, int len, int qh) {
    return get_value<3>(P, kmer, rkmer, lengthMask, len, qh);
}
__device__ __noinline__ int get_value_expand(const KParams& P, uint64_t kmer, uint64_t rkmer, uint64_t lengthMask, int len, int qh) {
    if (qh > 2) return get_value_expand3(P, kmer, rkmer, lengthMask, len, qh);       // qhdist <= 3 (bbduk_create refuses more)
    return get_value<2>(P, kmer, rkmer, lengthMask, len, qh);
}
// index.getValue(kmer, rkmer, lengthMask, qPos, len, qHDist) as a ref: filtered fast path when there is no query expansion
// QH = false (bbduk_wave_kernel): no query expansion in this instantiation -- batches with qhdist / qhdist2 > 0 go to the tiled kernels, so the
// wave kernels carry neither the expansion's calls nor the register spills around them
template <bool GENERAL, bool QH = true>
__device__ __forceinline__ int lookup(const KParams& P, const uint32_t* s_filt, uint64_t kmer, uint64_t rkmer,
                                      uint64_t lengthMask, int len, int qh, bool ok) {
    if constexpr (GENERAL && QH) {
        if (qh > 0) {
            const int id = ok ? get_value_expand(P, kmer, rkmer, lengthMask, len, qh) : -1;
            return id > 0 ? id : -1;
        }
    }
    return probe_ref(P, s_filt, make_value<GENERAL>(P, kmer, rkmer), lengthMask, ok);
}

// symbols [idx, idx+32) of a little-endian 2-bit stream, as a 64-bit value (caller masks)
__device__ __forceinline__ uint64_t extract2raw(const uint32_t* plane, int idx) {
    const int bit = idx * 2, w = bit >> 5;
    const uint32_t w0 = plane[w], w1 = plane[w + 1], w2 = plane[w + 2];
    const uint32_t lo = __builtin_amdgcn_alignbit(w1, w0, bit);      // the shift uses bit[4:0] only
    const uint32_t hi = __builtin_amdgcn_alignbit(w2, w1, bit);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t extract2(const uint32_t* plane, int idx, int nb) {   // nb in 1..31
    return extract2raw(plane, idx) & ((1ULL << (2 * nb)) - 1ULL);
}
// nb (1..31) bits starting at bit index `idx` of a little-endian 1-bit stream
__device__ __forceinline__ uint32_t extract1(const uint32_t* plane, int idx, int nb) {
    const int w = idx >> 5;
    const uint32_t v = __builtin_amdgcn_alignbit(plane[w + 1], plane[w], idx);
    return v & ((1u << nb) - 1u);
}

// 4 ASCII bases -> 4x2-bit forward codes (base 0 in bits 0-1), 4x2-bit complement codes, 4 valid bits.
// AminoAcid.baseToNumber0 / baseToComplementNumber0 / baseToNumber>=0 (dna/AminoAcid.java:1284-1311):
// A/a C/c G/g T/t U/u are defined, every other byte is undefined and encodes as 0 in both tables.
// The results stay one byte per base here (x, c: 2-bit codes; y: 0x01 per defined base); encode_chunk packs four of
// them at a time.  Validity: the 2-bit code picks the lower-case letter it stands for out of "acgt" (v_perm_b32) and
// the byte must equal it -- or be 'u'.
__device__ __forceinline__ void encode4(uint32_t w, uint32_t& x, uint32_t& c, uint32_t& y) {
    const uint32_t lower = w | 0x20202020u;
    const uint32_t raw = (w >> 1) & 0x03030303u;                  // A:0 C:1 G:3 T/U:2
    const uint32_t expect = __builtin_amdgcn_perm(0u, 0x67746361u, raw);   // byte = "actg"[raw]
    auto zb = [](uint32_t t) { return ~(((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t) & 0x80808080u; };   // 0x80 per zero byte (exact)
    const uint32_t v = zb(lower ^ expect) | zb(lower ^ 0x75757575u);
    y = v >> 7;                                                   // 0x01 per defined base
    const uint32_t vm = y * 3u;                                   // 0x03 per defined base
    x = (raw ^ ((raw >> 1) & 0x01010101u)) & vm;                  // A:0 C:1 G:2 T/U:3, undefined:0
    c = (~x) & vm;                                                // 3-x, undefined:0
}

// 16 consecutive bases starting at byte a -> 32-bit reversed forward codes, 32-bit complement codes, 16 valid bits
__device__ __forceinline__ void encode_chunk(const uint8_t* __restrict__ bases, const int64_t a, const int64_t totalBases,
                                             uint32_t& fwdRev, uint32_t& comp, uint32_t& valid, uint32_t* raw = nullptr) {
    uint32_t w[4];
    if (a + 16 <= totalBases) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(bases + a));   // streamed once: keep it out of L2's way
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
    } else {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t x = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int64_t p = a + 4 * q + b;
                if (p < totalBases) x |= (uint32_t)bases[p] << (8 * b);
            }
            w[q] = x;
        }
    }
    if (raw) { raw[0] = w[0]; raw[1] = w[1]; raw[2] = w[2]; raw[3] = w[3]; }   // the 16 symbols themselves (Seal: which of them is the letter N)
    // Packing: a multiply moves the four 2-bit fields of a word (bits 8j) next to each other into the top byte
    // (field j lands at 24+2j; all partial products fall on distinct bits, so nothing carries), and byte permutes
    // collect the four top bytes.  The 1-bit validity fields pack the same way with a 7-bit stride.
    uint32_t px[4], pc[4], py[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        uint32_t x, c, y;
        encode4(w[q], x, c, y);
        px[q] = x * 0x01041040u; pc[q] = c * 0x01041040u; py[q] = y * 0x01020408u;
    }
    auto top4 = [](const uint32_t* p) {                           // byte q of the result = top byte of p[q]
        const uint32_t lo = __builtin_amdgcn_perm(p[1], p[0], 0x0c0c0703u), hi = __builtin_amdgcn_perm(p[3], p[2], 0x07030c0cu);
        return lo | hi;
    };
    const uint32_t code = top4(px);
    comp = top4(pc);
    valid = ((py[0] >> 24) & 0xFu) | ((py[1] >> 20) & 0xF0u) | ((py[2] >> 16) & 0xF00u) | ((py[3] >> 12) & 0xF000u);
    uint32_t r = __brev(code);                                    // reverse the order of the 16 symbols
    fwdRev = ((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1);
}

// shared/TrimRead.java:304-345 trimByAmount on lengths
// One 16-base chunk of the batch -> the three plane words, from either boundary format.  Packed input (0.375
// B/base instead of 1) needs no character work at all: reverse the code word for the forward plane, complement it for
// the other, and the undefined bits are already there.
__device__ __forceinline__ uint32_t spread2(uint32_t v16) {        // bit j -> bits 2j and 2j+1
    uint32_t x = v16;
    x = (x | (x << 8)) & 0x00FF00FFu; x = (x | (x << 4)) & 0x0F0F0F0Fu; x = (x | (x << 2)) & 0x33333333u; x = (x | (x << 1)) & 0x55555555u;
    return x * 3u;
}
// FMT: 0 = ASCII, 1 = packed (both fixed at compile time: the specialised wave kernels), 2 = decided per launch
template <int FMT = 2>
__device__ __forceinline__ void stage_chunk(const KParams& P, const uint8_t* __restrict__ bases, const int64_t a, const int64_t totalBases,
                                            uint32_t& fwdRev, uint32_t& comp, uint32_t& valid) {
    if (FMT == 0 || (FMT == 2 && P.undef == nullptr)) { encode_chunk(bases, a, totalBases, fwdRev, comp, valid); return; }
    const int64_t w = a >> 4;
    const int64_t left = totalBases - a;                             // bases of this chunk inside the batch
    uint32_t code = 0, und = 0xFFFFu;
    if (left > 0) {
        code = reinterpret_cast<const uint32_t*>(bases)[w];
        und = (P.undef[w >> 1] >> (16 * (int)(w & 1))) & 0xFFFFu;
        if (left < 16) und |= 0xFFFFu << (int)left;
    }
    valid = ~und & 0xFFFFu;
    const uint32_t vm = spread2(valid);
    code &= vm;
    comp = ~code & vm;
    const uint32_t r = __brev(code);
    fwdRev = ((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1);
}

__device__ __forceinline__ int trim_by_amount(int len, int left, int right, int minRes, int& newLen) {
    left = max(left, 0); right = max(right, 0);
    if (len < 1) { newLen = len; return 0; }
    minRes = min(len, max(minRes, 0));
    if (left + right + minRes > len) { right = max(1, len - minRes); left = 0; }
    newLen = len - (left + right);
    return left + right;
}
// setDiscarded / isDiscarded with trimfailuresto1bp (BBDukProcessorS.java:1464-1482): a read that was to be discarded is cut to one base
// (if it is longer), and "discarded" then means "exactly one base long" -- also for a read that is one base long for any other reason
__device__ __forceinline__ void tf1bp(const KParams& P, bool& d, int& len) { if (PTF) { if (d && len > 1) len = 1; d = (len == 1); } }
__device__ __forceinline__ int imid(int lo, int x, int hi) { return x < lo ? lo : (x > hi ? hi : x); }
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }   // value known wave-uniform -> SGPR

// fwdBits / cmpBits = 8 x the LDS byte address of the plane's symbol 0: "bit address" arithmetic lets a lane get from
// a symbol index to (word address, shift) in three instructions (cut64_lds).
struct Planes { const uint32_t* fwd; const uint32_t* cmp; const uint32_t* nm; const uint32_t* filt; int T; uint32_t fwdBits, cmpBits; };
typedef __attribute__((address_space(3))) const uint32_t lds_cu32;
__device__ __forceinline__ uint32_t lds_word_at(uint32_t byteAddr) { return *reinterpret_cast<lds_cu32*>(byteAddr); }
__device__ __forceinline__ uint32_t lds_bits_of(const uint32_t* p) { return 8u * (uint32_t)(size_t)(lds_cu32*)p; }
// 32 symbols of a 2-bit stream from LDS bit address tbits (shifts use tbits[4:0] only)
__device__ __forceinline__ uint64_t cut64_lds(const uint32_t tbits) {
    const uint32_t a = (tbits >> 3) & ~3u;
    const uint32_t w0 = lds_word_at(a), w1 = lds_word_at(a + 4u), w2 = lds_word_at(a + 8u);
    return ((uint64_t)__builtin_amdgcn_alignbit(w2, w1, tbits) << 32) | __builtin_amdgcn_alignbit(w1, w0, tbits);
}

// Wave-uniform state of one read's scan (all fields live in SGPRs).  Only raw scan facts are kept here; the
// trim / id arithmetic happens afterwards in finish_read(), which the wave kernel runs data-parallel over the
// reads of a mini-tile (one lane per read) and the tile kernel runs per read.
struct ReadScan {
    int base0, L, start, stop;          // base0 = index of the read's first base in the planes
    bool scan;                          // false: the reference returns 0 before scanning (too short, skipR1/2, no k-mers)
    int found, iFirst, iLast;           // main scan: hits seen, first / last hit position (iFirst==0 marks the kfilter exit)
    int ref;                            // ref (see ref_to_id) of the hit whose id the reference reports
    int shortFl, shortLl;               // short k-mer scan: first / last hit lane (length index); shortFl<0: none
    int candSlot; uint32_t candKeyLo, candKeyHi;   // candidate mode: first unverified fingerprint match (slot, key)
    uint32_t candWord;                  // big layout: the candidate's tag word (candSlot is then its way 0..3)
    int hasN;                           // undefined base inside [start,stop)?  1/0, or -1 = not known yet (forbidNs only)
    int maxBad;                         // kfilter: this read's maxBadKmers (mkf) or minCoveredBases (mcf)
};

template <int MODE, bool SHORT, bool GENERAL>
__device__ __forceinline__ bool scan_due(const KParams& P, int L, int pairnum, bool present) {
    const int k = P.k;
    bool s = present && P.storedKmers > 0;
    if (MODE == 7 /* BBDUK_MODE_KBIG */) s = s && (L >= P.kbig);                          // :1727-1728
    else if (MODE == BBDUK_MODE_KFILTER || MODE == BBDUK_MODE_KSPLIT || MODE == BBDUK_MODE_KMASK || MODE == 6 /* BBDUK_MODE_FBM */) s = s && (L >= k);   // BBDukProcessorS.java:1535; ksplit :2333, 2338; kmask :2151
    else s = s && (L >= max(1, (SHORT && P.useShort) ? min(k, P.mink) : k));               // :1995
    if constexpr (GENERAL) {
        if ((P.skipR1 && pairnum == 0) || (P.skipR2 && pairnum == 1)) s = false;           // :1536, :1996
    }
    return s;
}
template <bool GENERAL> __device__ __forceinline__ int span_start(const KParams& P, int L) {      // :1808-1809, :1542-1543
    if constexpr (GENERAL) return (P.restrictRight < 1 ? 0 : max(0, L - P.restrictRight));
    return 0;
}
template <bool GENERAL> __device__ __forceinline__ int span_stop(const KParams& P, int L) {
    if constexpr (GENERAL) return (P.restrictLeft < 1 ? L : min(L, P.restrictLeft));
    return L;
}
template <int MODE, bool SHORT, bool GENERAL>
__device__ __forceinline__ void read_init(const KParams& P, ReadScan& R, int base0, int L, int pairnum, bool present) {
    R.base0 = base0; R.L = L; R.hasN = -1; R.maxBad = P.maxBadKmers;
    R.found = 0; R.iFirst = BIGLOC; R.iLast = -1; R.ref = -1; R.shortFl = -1; R.shortLl = -1;
    R.start = span_start<GENERAL>(P, L); R.stop = span_stop<GENERAL>(P, L);
    R.scan = scan_due<MODE, SHORT, GENERAL>(P, L, pairnum, present);
}

// From the raw scan facts of one read to the operator's outputs (works on wave-uniform or per-lane values alike).
// ktrim: bbduk/BBDukProcessorS.java:2031-2032, 2108-2139 + shared/TrimRead.java:273-345; kfilter: :1575-1591.
// a = ktrim x | countSetKmers return; ref = ref of the credited scaffold or -1; hit = a scaffold counter is due.
template <int MODE>
__device__ __forceinline__ void finish_read(const KParams& P, const int L, const int start, const int stop, const int found,
                                            const int iFirst, const int iLast, const int shortFl, const int shortLl, const int refIn,
                                            int& a, int& newLen, int& ref, bool& hit) {
    a = 0; newLen = L; ref = -1; hit = false;
    if (MODE == BBDUK_MODE_KFILTER) {
        a = found;
        if (iFirst == 0) { ref = refIn; hit = true; }                   // early exit taken
        return;
    }
    const int k = P.k;
    int minLoc = BIGLOC, minLocEx = BIGLOC, maxLoc = -1, maxLocEx = -1;
    if (found > 0 && shortFl < 0) { minLoc = iFirst - k + 1; maxLoc = iLast; minLocEx = minLoc + k; maxLocEx = maxLoc - k; }
    if (shortFl >= 0) {                                                  // short k-mer hits (only when the main scan found none)
        if (MODE == BBDUK_MODE_KTRIM_L) {
            minLoc = 0; minLocEx = start + (P.mink + shortFl); maxLoc = start + (P.mink + shortLl) - 1; maxLocEx = 0;
        } else {
            minLoc = stop - (P.mink + shortLl); minLocEx = L; maxLoc = L - 1; maxLocEx = stop - (P.mink + shortFl) - 1;
        }
    }
    if (found == 0) return;                                              // :2108
    hit = true; ref = refIn;
    if (P.trimPad != 0) {                                                // :2121-2126
        maxLoc = imid(0, maxLoc + P.trimPad, L);
        minLoc = imid(0, minLoc - P.trimPad, L);
        maxLocEx = imid(0, maxLocEx + P.trimPad, L);
        minLocEx = imid(0, minLocEx - P.trimPad, L);
    }
    if (MODE == BBDUK_MODE_KTRIM_L) {     // trimToPosition(r, leftLoc, len-1, 1)
        const int leftLoc = P.ktrimExclusive ? maxLocEx + 1 : maxLoc + 1;
        a = trim_by_amount(L, leftLoc, 0, 1, newLen);
    } else {                              // trimToPosition(r, 0, rightLoc, 1)
        const int rightLoc = P.ktrimExclusive ? minLocEx - 1 : minLoc - 1;
        a = trim_by_amount(L, 0, L - rightLoc - 1, 1, newLen);
    }
}

// ---- main scan -------------------------------------------------------------------------------------
// One wave scans the two reads of a unit together.  Each loop iteration covers 128 k-mer end positions of
// each read: four "slots" (read A/B x positions +0/+64), one lane per position in each slot (closed form,
// SURVEY A.12).  The four slots are computed in straight-line code so that their LDS reads and their bucket
// gathers are in flight together (the scan is latency-bound otherwise: LDS -> filter -> L2 gather per pass).
// bbduk/BBDukProcessorS.java:2009-2029 (ktrim) == :1547-1591 (countSetKmers).

struct ReadWin { int first, stop, start, base0; bool on, full, hasN; };   // wave-uniform per read

template <bool FORBIDN, bool GENERAL, bool BIG = false, bool SPAN = false>
__device__ __forceinline__ void win_init(const KParams& P, const Planes& Q, const ReadScan& R, ReadWin& W, const int lane) {
    W.start = R.start; W.stop = R.stop; W.base0 = R.base0;
    W.first = max(R.start, P.k - 1);                             // i>=minlen (minlen=k-1)
    W.on = R.scan && W.first < W.stop;
    W.hasN = false;                                              // undefined base inside [start,stop)? (forbidNs; big layout: always)
    if (((FORBIDN && P.forbidNs) || BIG) && W.on && R.hasN >= 0) W.hasN = R.hasN != 0;
    else if (((FORBIDN && P.forbidNs) || BIG) && W.on) {
        const int b0 = W.base0 + W.start, b1 = W.base0 + W.stop;
        uint32_t acc = 0;
        for (int w = (b0 >> 5) + lane; w <= ((b1 - 1) >> 5); w += 64) {
            uint32_t v = Q.nm[w];
            const int lo = w << 5;
            if (lo < b0) v &= ~0u << (b0 - lo);
            if (lo + 32 > b1) v &= ~0u >> (lo + 32 - b1);
            acc |= v;
        }
        W.hasN = __ballot(acc != 0) != 0;
    }
    // full: the fast window path serves the read -- every window holds k bases and every position is looked up; with SPAN it also cuts the
    // windows in front of `start` and thins the positions by qskip itself
    W.full = (!GENERAL || SPAN || (W.start == 0 && P.qskip < 2));
}

// kmer / rkmer of the windows ending at the ADJACENT positions i and i+1 of read W (lane-varying i); ok=false: no
// lookup due.  Plain reads (no reset, no cut): one (k+1)-symbol cut per plane serves both positions (k+1 <= 32 symbols
// fit the 64-bit cut); the planes are padded, so no clamping -- out-of-read lanes are simply not ok.
// SPAN (specialised kernels of ktrim=rl): the scan span may start inside the read.  The reference starts its rolling k-mer at `start`, so
// the first k-1 windows of the span hold fewer than k bases -- bases in front of `start` are cut out of both k-mers and such a window is
// looked up only from minlen2 bases on (:2010-2019 with the loop's own start).
template <bool FORBIDN, bool GENERAL, bool SPAN = false>
__device__ __forceinline__ void windows2(const KParams& P, const Planes& Q, const ReadWin& W, const int i, const bool on,
                                         uint64_t* kmer, uint64_t* rk, bool* ok, uint64_t* rkRaw = nullptr) {
    const int k = P.k;
    if (W.full) {
        const uint64_t wf = cut64_lds(Q.fwdBits + 2u * (uint32_t)(Q.T - 1 - W.base0 - (i + 1)));   // base i+1 in bits 0-1, base i-k+1 on top
        const uint64_t wc = cut64_lds(Q.cmpBits + 2u * (uint32_t)(W.base0 - k + 1 + i));           // base i-k+1 in bits 0-1, base i+1 on top
        if constexpr (!GENERAL) {                                // specialised kernels run with k >= 16: the mask's low word is all ones
            const uint64_t mh = P.mask | 0xFFFFFFFFULL;
            kmer[1] = wf & mh; kmer[0] = (wf >> 2) & mh;
            rk[0] = wc & mh;   rk[1] = (wc >> 2) & mh;
        } else {
            kmer[1] = wf & P.mask; kmer[0] = (wf >> 2) & P.mask;
            rk[0] = wc & P.mask;   rk[1] = (wc >> 2) & P.mask;
        }
        ok[0] = on & (i < W.stop);
        ok[1] = on & (i + 1 < W.stop);
        if constexpr (GENERAL && SPAN) {
            if (P.qskip > 1) { ok[0] = ok[0] && (i % P.qskip) == 0; ok[1] = ok[1] && ((i + 1) % P.qskip) == 0; }     // BBDukIndexMod.java:494
        }
        if (rkRaw) { rkRaw[0] = rk[0]; rkRaw[1] = rk[1]; }        // before any reset (big layout: the neighbours' minimizer hashes)
        int cut[2] = {0, 0};                                      // SPAN: bases of the window that lie in front of the span
        if constexpr (SPAN) {
            if (W.start > 0 && __ballot(i - k + 1 < W.start) != 0ULL) {
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int c = min(max(W.start - (i + h - k + 1), 0), k);     // window [i+h-k+1, i+h], span from W.start
                    cut[h] = c;
                    kmer[h] &= (1ULL << (2 * (k - c))) - 1ULL;                    // the k-c bases inside the span are the low ones
                    rk[h] &= ~((1ULL << (2 * c)) - 1ULL);                         // complement of base i+h-k+1+t sits at bits 2t
                    ok[h] = ok[h] && (k - c) >= P.minlen2;
                }
            }
        }
        if (FORBIDN && P.forbidNs && W.hasN) {                   // the read holds an undefined base somewhere: patch the few windows that see it
            const int nidx = min(W.base0 - k + 1 + i, Q.T);      // bit t <=> base i-k+1+t undefined
            const uint32_t nw = __builtin_amdgcn_alignbit(Q.nm[(nidx >> 5) + 1], Q.nm[nidx >> 5], nidx);
            if (__ballot(nw != 0u) != 0ULL) {
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    uint32_t nh = (nw >> h) & ((1u << k) - 1u);
                    if constexpr (SPAN) nh &= ~0u << cut[h];     // an undefined base in front of the span does not count
                    if (nh) {
                        const int msb = 31 - __clz(nh);
                        rk[h] &= ~0ULL << (2 * (msb + 1));       // rkmer was reset there; kmer keeps its history
                        ok[h] = ok[h] && (k - 1 - msb) >= P.minlen2;   // len = bases after the last undefined one
                    }
                }
            }
        }
        return;
    }
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int ih = i + h;
        const bool act = on && ih < W.stop;
        const int ic = min(max(ih, W.first), W.stop - 1);        // inactive lanes read in-bounds
        const int lo = max(W.start, ic - k + 1);
        const int nb = ic - lo + 1;                              // bases in the window (== k unless cut by start)
        uint64_t km = extract2(Q.fwd, Q.T - 1 - W.base0 - ic, nb);
        uint64_t rr = extract2(Q.cmp, W.base0 + lo, nb);
        int len = ic - W.start + 1;
        if (FORBIDN && W.hasN) {
            const uint32_t nwin = extract1(Q.nm, W.base0 + lo, nb);          // bit t <=> base lo+t undefined
            if (nwin) {
                const int msb = 31 - __clz(nwin);
                len = nb - 1 - msb;                              // bases after the last undefined one
                rr &= ~0ULL << (2 * (msb + 1));                  // rkmer was reset there; kmer keeps its history
            }
        }
        rr <<= 2 * (k - nb);                                     // base j sits at 2*(k-1-(i-j))
        kmer[h] = km; rk[h] = rr;
        ok[h] = act && len >= P.minlen2;
        if constexpr (GENERAL) { if (P.qskip > 1) ok[h] = ok[h] && (ih % P.qskip) == 0; }            // BBDukIndexMod.java:494
    }
}

// four independent key -> ref lookups with their memory operations overlapped
template <bool GENERAL, bool BIG = false>
__device__ __forceinline__ void lookup4(const KParams& P, const uint32_t* s_filt, const uint64_t* kmer, const uint64_t* rk,
                                        const bool* ok, int* ref) {
    if constexpr (GENERAL) {
        if (P.qhdist > 0) {
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const int id = ok[s] ? get_value_expand(P, kmer[s], rk[s], P.kmask, P.k, P.qhdist) : -1;
                ref[s] = id > 0 ? id : -1;
            }
            return;
        }
    }
    uint64_t key[4], t[4]; uint32_t ma[4], mb[4]; bool p[4];
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const uint64_t value = make_value<GENERAL>(P, kmer[s], rk[s]);
        key[s] = value | P.kmask;
        ma[s] = mix_a(value);
        mb[s] = mix_b(value);
        p[s] = ok[s];
        if constexpr (GENERAL) { if (P.speed > 0) p[s] = p[s] && passes_speed(P, key[s]); }
    }
    if (TSW(P, 2)) {                                             // experiment: keys and hashes only
#pragma unroll
        for (int s = 0; s < 4; s++) ref[s] = (p[s] && ma[s] == 0x12345u && mb[s] == 0x54321u) ? 0 : -1;
        return;
    }
    if constexpr (BIG) {                                          // (no other kernel ever meets a big map: see big_find)
        if (P.big) {                                              // HBM-resident layout: the exact generic lookup (the first-hit scans of the
            for (int s = 0; s < 4; s++) ref[s] = p[s] ? big_find(P, key[s], ma[s], mb[s]) : -1;     // plain configurations have cand_probe4_big)
            return;
        }
    }
    if (P.ldsBits) {                                              // four presence bits, read together
        uint32_t w[4];
#pragma unroll
        for (int s = 0; s < 4; s++) w[s] = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(s_filt) + filt_byte(ma[s], P.ldsBits));
#pragma unroll
        for (int s = 0; s < 4; s++) p[s] = p[s] & (bool)((w[s] >> (ma[s] & 31u)) & 1u);
    }
    if (TSW(P, 1)) {                                             // experiment: filter but no gathers
#pragma unroll
        for (int s = 0; s < 4; s++) ref[s] = (p[s] && ma[s] == 0x12345u) ? 0 : -1;
        return;
    }
#pragma unroll
    for (int s = 0; s < 4; s++) t[s] = p[s] ? P.tags[bucket_of(mb[s], P.bucketBits)] : 0ULL;  // four gathers in flight
    if (TSW(P, 6)) {                                             // experiment: gathers issued, matches ignored
#pragma unroll
        for (int s = 0; s < 4; s++) ref[s] = (t[s] == 0x123456789ULL) ? 0 : -1;
        return;
    }
#pragma unroll
    for (int s = 0; s < 4; s++) {
        // rare: a fingerprint matched, or the home bucket overflowed -> check the key / walk the bucket chain
        const uint64_t cand = zero16((t[s] & TAG_FPS) ^ ((uint64_t)tag_of(ma[s]) * 0x0001000100010001ULL));
        ref[s] = -1;
        if (p[s] && (cand != 0ULL || (t[s] & TAG_CONT))) ref[s] = table_find_t(P, key[s], ma[s], mb[s], t[s]);
    }
}

// Candidate form of lookup4 for the first-hit-only scans, in two parts.  cand_probe4 is the straight-line part every
// slot runs: key value, both mixes, LDS presence bit, fingerprint gather, four 16-bit compares; it returns the union
// of the four ballots "this lane needs a closer look" (a fingerprint matched or the home bucket overflowed).  Keys of
// matching fingerprints are NOT fetched.  There is no `ok` input: every lane is looked up (the bucket index is always
// in range) and the caller rejects tail lanes by position, which keeps per-lane predicates out of the hot code.
// cand_resolve4 runs only when that union is non-zero: ref = slot (4*bucket+way) of the first fingerprint match,
// -3-id for a hit already verified (overflowed home bucket without a match: the chain is walked at once), -1 if
// certainly absent.  C.key = the lanes' key VALUES (no length bit).  NOMM: the middle mask is known to be off.
struct Cand4 { uint64_t key[4], t[4]; uint32_t ma[4], mb[4], pv[4]; uint64_t hm[4]; };   // hm: per-slot wave masks of flagged lanes

// FILT0: the filter starts at LDS address 0 (wave kernel), so a word's LDS address is its byte offset
template <bool GENERAL, bool NOMM, bool FILT0>
__device__ __forceinline__ uint64_t cand_probe4(const KParams& P, const uint32_t* s_filt, const uint64_t* kmer, const uint64_t* rk, Cand4& C) {
    const uint32_t mmLo = (uint32_t)P.middleMask, mmHi = (uint32_t)(P.middleMask >> 32);
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const uint64_t mx = (!GENERAL || P.rcomp) ? (kmer[s] > rk[s] ? kmer[s] : rk[s]) : kmer[s];
        const uint32_t klo = NOMM ? (uint32_t)mx : ((uint32_t)mx & mmLo), khi = NOMM ? (uint32_t)(mx >> 32) : ((uint32_t)(mx >> 32) & mmHi);
        C.key[s] = ((uint64_t)khi << 32) | klo;
        C.ma[s] = klo * HA0 + khi * HA1;
        C.mb[s] = klo * HB0 + khi * HB1;
        C.pv[s] = 1u;
    }
    bool sp[4] = {true, true, true, true};
    if constexpr (GENERAL) {
        if (P.speed > 0) {
#pragma unroll
            for (int s = 0; s < 4; s++) sp[s] = passes_speed(P, C.key[s] | P.kmask);
        }
    }
    if (TSW(P, 2)) {                                              // experiment: keys and hashes only
        uint64_t a = 0;
#pragma unroll
        for (int s = 0; s < 4; s++) { C.pv[s] = 0; C.t[s] = 0; C.hm[s] = __ballot(C.ma[s] == 0x12345u && C.mb[s] == 0x54321u); a |= C.hm[s]; }
        return a;
    }
    if (P.ldsBits) {                                              // four presence bits, read together
        uint32_t w[4];
#pragma unroll
        for (int s = 0; s < 4; s++) w[s] = FILT0 ? lds_word_at(filt_byte(C.ma[s], P.ldsBits))
                                                 : *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(s_filt) + filt_byte(C.ma[s], P.ldsBits));
#pragma unroll
        for (int s = 0; s < 4; s++) C.pv[s] = __builtin_amdgcn_ubfe(w[s], C.ma[s], 1u);   // the offset operand uses ma[4:0] only
    }
    if constexpr (GENERAL) {
#pragma unroll
        for (int s = 0; s < 4; s++) C.pv[s] = sp[s] ? C.pv[s] : 0u;
    }
    if (TSW(P, 1)) {                                              // experiment: filter but no gathers
        uint64_t a = 0;
#pragma unroll
        for (int s = 0; s < 4; s++) { C.hm[s] = __ballot(C.pv[s] && C.ma[s] == 0x12345u); a |= C.hm[s]; C.pv[s] = 0; C.t[s] = 0; }
        return a;
    }
#pragma unroll
    for (int s = 0; s < 4; s++) C.t[s] = P.tags[C.pv[s] ? bucket_of(C.mb[s], P.bucketBits) : 0u];   // four gathers in flight; lanes the
                                                                                              // filter rejected all read tags[0] (one line)
    if (TSW(P, 6)) {                                              // experiment: gathers issued, matches ignored
        uint64_t a = 0;
#pragma unroll
        for (int s = 0; s < 4; s++) { C.hm[s] = __ballot(C.t[s] == 0x123456789ULL); a |= C.hm[s]; C.pv[s] = 0; }
        return a;
    }
    uint64_t any = 0;
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const uint32_t fp = tag_of(C.ma[s]), tlo = (uint32_t)C.t[s], thi = (uint32_t)(C.t[s] >> 32);
        // four 16-bit compares (v_cmp_eq_u32_sdwa) and the overflow flag, combined as wave masks on the scalar unit.
        // Lane 3 of an overflowed bucket carries the flag in its top bit and never compares equal: such buckets take
        // the chain walk, which masks the flag.
        const uint64_t b = __ballot((tlo & 0xFFFFu) == fp) | __ballot((tlo >> 16) == fp) | __ballot((thi & 0xFFFFu) == fp) |
                           __ballot((thi >> 16) == fp) | __ballot((int32_t)thi < 0);
        C.hm[s] = b & __ballot(C.pv[s] != 0u);
        any |= C.hm[s];
    }
    return any;
}

__device__ __forceinline__ void cand_resolve4(const KParams& P, const Cand4& C, int* ref) {
#pragma unroll
    for (int s = 0; s < 4; s++) {
        ref[s] = -1;
        if (C.hm[s] == 0ULL) continue;                            // nothing flagged in this slot (wave-uniform)
        const uint32_t fp = tag_of(C.ma[s]);
        const uint64_t cand = zero16((C.t[s] & TAG_FPS) ^ ((uint64_t)fp * 0x0001000100010001ULL));
        if (C.pv[s] && (cand != 0ULL || (C.t[s] & TAG_CONT))) {
            if (!(C.t[s] & TAG_CONT)) ref[s] = (int)(4u * bucket_of(C.mb[s], P.bucketBits)) + ((__ffsll((unsigned long long)cand) - 1) >> 4);
            else {
                const int id = table_find_t(P, C.key[s] | P.kmask, C.ma[s], C.mb[s], C.t[s]);
                ref[s] = id > 0 ? -3 - id : -1;                    // <= -4: a verified id, nothing left to check
            }
        }
    }
}

// lookup4 in the candidate probe's style, for the wave kernel's scans that need EVERY hit (ktrim=l, ktrim=n, ksplit, the left pass of
// ktrim=rl): the straight-line probe, one scalar test that ends most blocks, and only flagged lanes fetch keys.  Exact like lookup4.
template <bool GENERAL>
__device__ __forceinline__ void lookup4_probe(const KParams& P, const uint32_t* s_filt, const uint64_t* kmer, const uint64_t* rk, const bool* ok, int* ref) {
    Cand4 C;                                                      // (no query expansion here: such batches run on the tiled kernels)
    const uint64_t any = cand_probe4<GENERAL, false, true>(P, s_filt, kmer, rk, C);
#pragma unroll
    for (int s = 0; s < 4; s++) ref[s] = -1;
    if (any == 0ULL) return;
#pragma unroll
    for (int s = 0; s < 4; s++) {
        if (C.hm[s] == 0ULL) continue;                            // nothing flagged in this slot (wave-uniform)
        const uint64_t cand = zero16((C.t[s] & TAG_FPS) ^ ((uint64_t)tag_of(C.ma[s]) * 0x0001000100010001ULL));
        if (ok[s] && C.pv[s] && (cand != 0ULL || (C.t[s] & TAG_CONT))) ref[s] = table_find_t(P, C.key[s] | P.kmask, C.ma[s], C.mb[s], C.t[s]);
    }
}

// ---- big layout, fast form of the candidate probe (specialised kernels, plain k >= 16 configurations) -----------------
// The 52-bit candidate values travel as the mantissas of doubles in [1, 2): positive normal doubles order like their bit patterns, so
// one v_min_f64 (full rate on CDNA) is the 52-bit minimum.  next_lane: the value of lane+1 (wave_shl:1, DPP, gfx9: no LDS crossbar);
// lane 63 gets +inf-like `fill`.
typedef double gapv;
__device__ __forceinline__ gapv gap_pack(const uint64_t h52) { return __longlong_as_double((long long)(h52 | 0x3FF0000000000000ULL)); }
__device__ __forceinline__ uint64_t gap_unpack(const gapv v) { return (uint64_t)__double_as_longlong(v) & 0x000FFFFFFFFFFFFFULL; }
__device__ __forceinline__ gapv gmin(const gapv a, const gapv b) { gapv d; asm("v_min_f64 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ gapv next_lane(const gapv v) {
    const uint64_t u = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)(uint32_t)u, 0x130, 0xF, 0xF, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp((int)0x3FFFFFFFu, (int)(uint32_t)(u >> 32), 0x130, 0xF, 0xF, false);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
// Sliding minimum over W consecutive POSITIONS of per-position values held two per lane (e = position 2l, o = 2l+1):
// me / mo = min over positions [2l, 2l+W) / [2l+1, 2l+1+W).  Lanes near the top of the wave see the fill value beyond it.
__device__ __forceinline__ void window_min(const int W, const gapv e, const gapv o, gapv& me, gapv& mo) {
    if (W == 6 || W == 7) {                                       // k = 31: H = 15, m = 10 / 9
        const gapv s1e = gmin(e, o), s1o = gmin(o, next_lane(e));                     // 2 positions: [2l,2l+2), [2l+1,2l+3)
        const gapv s2e = gmin(s1e, next_lane(s1e)), s2o = gmin(s1o, next_lane(s1o));  // 4 positions
        if (W == 6) {
            me = gmin(s2e, next_lane(next_lane(s1e)));            // [2l,2l+4) + [2l+4,2l+6)
            mo = gmin(s2o, next_lane(next_lane(s1o)));            // [2l+1,2l+5) + [2l+5,2l+7)
        } else {
            me = gmin(s2e, next_lane(s2o));                       // [2l,2l+4) + [2l+3,2l+7)
            mo = gmin(s2o, next_lane(next_lane(s2e)));            // [2l+1,2l+5) + [2l+4,2l+8)
        }
        return;
    }
    gapv ce = e, co = o; me = e; mo = o;
    for (int d = 1; d < W; d++) {                                 // shift by one position: (e, o) <- (o, next lane's e)
        const gapv ne = co, no = next_lane(ce);
        ce = ne; co = no;
        me = gmin(me, ce); mo = gmin(mo, co);
    }
}
// h(0) of the window whose forward / reverse-complement k-mers are kmer / rk: its left-most gapped-mer, read from kmer, and the
// reverse complement of that gapped-mer, which is the right-most candidate of rk (see "big layout" above)
__device__ __forceinline__ uint64_t gap_h0(const KParams& P, const uint64_t kmer, const uint64_t rk) {
    const uint32_t mk = (1u << (2 * P.gm)) - 1u;
    if (P.k == 31 && P.gH == 15) {                                // halves = bases 0-14 (high word) and 16-30 (low word): 32-bit cuts
        const uint32_t sh = 2u * (uint32_t)(15 - P.gm), hi = (uint32_t)(kmer >> 32), lo = (uint32_t)kmer;
        return gap_pair(gap_f(hi >> sh, (lo >> sh) & mk), gap_f((uint32_t)(rk >> 32) & mk, (uint32_t)rk & mk));
    }
    const uint32_t la = (uint32_t)(kmer >> (2 * (P.k - P.gm))) & mk, ra = (uint32_t)(kmer >> (2 * (P.k - P.gD - P.gm))) & mk;
    const uint32_t lb = (uint32_t)(rk >> (2 * (P.k - P.gH))) & mk, rb = (uint32_t)rk & mk;
    return gap_pair(gap_f(la, ra), gap_f(lb, rb));
}
// Candidate probe on the big layout.  Slots 0/1 = read A's positions 2l / 2l+1, slots 2/3 = read B's.  The line of a window comes
// from the minimum over W consecutive per-position values, shared across lanes (window_min) instead of recomputed per key; that
// is exact only for windows whose kmer / rkmer are true reverse complements, so windows that see an undefined base (nf) are
// flagged and looked up by the generic exact path in cand_resolve4_big.  Windows at the top of the block (position >= 128 - W
// within it) lack their successors: the caller advances by BIG_STEP positions and ignores them.
#define BIG_STEP 120
// (the tag words themselves are not kept: the rare resolve step loads them again, which keeps 24 VGPRs out of the common path's live ranges)
struct Cand4Big { uint64_t key[4]; uint32_t ma[4], mb[4], w1[4], w2[4]; uint64_t hm[4]; bool nf[4], sp[4]; };
template <bool NOMM>
__device__ __forceinline__ uint64_t cand_probe4_big(const KParams& P, const uint64_t* kmer, const uint64_t* rk, const uint64_t* rkRaw, const bool* nf, Cand4Big& C) {
    const uint32_t mmLo = (uint32_t)P.middleMask, mmHi = (uint32_t)(P.middleMask >> 32);
    uint64_t hw[4];
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const uint64_t mx = kmer[s] > rk[s] ? kmer[s] : rk[s];
        const uint32_t klo = NOMM ? (uint32_t)mx : ((uint32_t)mx & mmLo), khi = NOMM ? (uint32_t)(mx >> 32) : ((uint32_t)(mx >> 32) & mmHi);
        C.key[s] = ((uint64_t)khi << 32) | klo;
        C.ma[s] = klo * HA0 + khi * HA1;
        C.mb[s] = klo * HB0 + khi * HB1;
        C.nf[s] = nf[s];
    }
    if (P.gW > 0) {
        gapv h0[4], hm_[4];
#pragma unroll
        for (int s = 0; s < 4; s++) h0[s] = gap_pack(gap_h0(P, kmer[s], rkRaw[s]));     // the raw complement cut: a reset further right does not touch these bases
        window_min(P.gW, h0[0], h0[1], hm_[0], hm_[1]);
        window_min(P.gW, h0[2], h0[3], hm_[2], hm_[3]);
#pragma unroll
        for (int s = 0; s < 4; s++) hw[s] = gap_unpack(hm_[s]);
    } else {                                                      // plain lines: a function of the key itself
#pragma unroll
        for (int s = 0; s < 4; s++) hw[s] = (uint64_t)(C.ma[s] ^ 0x5BD1E995u);
    }
    uint64_t t1[4], t2[4], ts[4];
#pragma unroll
    for (int s = 0; s < 4; s++) {
        big_words(gap_line(hw[s], P.bigLines), C.mb[s], C.w1[s], C.w2[s]);
        t1[s] = P.bigTags[C.w1[s]];                                // eight gathers in flight; a lane's two words and its neighbours' share a sector
        t2[s] = P.bigTags[C.w2[s]];
    }
    // the key's spill bit in its primary word: only those lanes (~5 %) look into the secondary map, the others re-read its bucket 0
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const uint32_t half = (C.ma[s] & 0x4000u) ? (uint32_t)(t1[s] >> 32) : (uint32_t)t1[s];    // spill_bit(ma) = 16*((ma>>13)&3)+15
        C.sp[s] = ((half >> ((C.ma[s] & 0x2000u) ? 31 : 15)) & 1u) != 0u;
        ts[s] = P.tags[C.sp[s] ? bucket_of(C.mb[s], P.bucketBits) : 0u];
    }
    uint64_t any = 0;
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const uint32_t fp = tag_of(C.ma[s]);
        const uint32_t alo = (uint32_t)t1[s] & 0x7FFF7FFFu, ahi = (uint32_t)(t1[s] >> 32) & 0x7FFF7FFFu;      // without the spill bits
        const uint32_t blo = (uint32_t)t2[s] & 0x7FFF7FFFu, bhi = (uint32_t)(t2[s] >> 32) & 0x7FFF7FFFu;
        const uint32_t slo = (uint32_t)ts[s], shi = (uint32_t)(ts[s] >> 32);
        // twelve 16-bit compares; in the secondary map's bucket lane 3 carries the continuation flag and never compares equal then
        const uint64_t prim = __ballot((alo & 0xFFFFu) == fp) | __ballot((alo >> 16) == fp) | __ballot((ahi & 0xFFFFu) == fp) | __ballot((ahi >> 16) == fp) |
                              __ballot((blo & 0xFFFFu) == fp) | __ballot((blo >> 16) == fp) | __ballot((bhi & 0xFFFFu) == fp) | __ballot((bhi >> 16) == fp);
        const uint64_t sec = (__ballot((slo & 0xFFFFu) == fp) | __ballot((slo >> 16) == fp) | __ballot((shi & 0xFFFFu) == fp) | __ballot((shi >> 16) == fp) |
                              __ballot((int32_t)shi < 0)) & __ballot(C.sp[s]);
        C.hm[s] = prim | sec | __ballot(nf[s]);
        any |= C.hm[s];
    }
    return any;
}
// ref: way (0..3) of the first fingerprint match, in word C.w1 (ref < 4) or C.w2 (ref - 4), unverified; -3-id for a hit verified
// here (a spilled key found in the secondary map, or the exact lookup of a window with an undefined base); -1 = certainly absent
__device__ __forceinline__ void cand_resolve4_big(const KParams& P, const Cand4Big& C, int* ref) {
#pragma unroll
    for (int s = 0; s < 4; s++) {
        ref[s] = -1;
        if (C.hm[s] == 0ULL) continue;                            // nothing flagged in this slot (wave-uniform)
        if (C.nf[s]) {                                            // kmer / rkmer are not each other's reverse complement: the key's own line
            const int id = big_find(P, C.key[s] | P.kmask, C.ma[s], C.mb[s]);
            ref[s] = id > 0 ? -3 - id : -1;
            continue;
        }
        const uint64_t pat = (uint64_t)tag_of(C.ma[s]) * 0x0001000100010001ULL;
        const uint64_t t1 = P.bigTags[C.w1[s]], t2 = P.bigTags[C.w2[s]];       // (again: cache hits)
        const uint64_t c1 = zero16((t1 & TAG_FPS15) ^ pat), c2 = zero16((t2 & TAG_FPS15) ^ pat);
        if (c1) ref[s] = (__ffsll((unsigned long long)c1) - 1) >> 4;
        else if (c2) ref[s] = 4 + ((__ffsll((unsigned long long)c2) - 1) >> 4);
        else if (C.sp[s]) {                                       // a key of this kind was spilled from the primary word: the secondary map answers
            const uint64_t ts = P.tags[bucket_of(C.mb[s], P.bucketBits)];
            const uint64_t c3 = zero16((ts & TAG_FPS) ^ pat);
            if (c3 != 0ULL || (ts & TAG_CONT)) {
                const int id = table_find_t(P, C.key[s] | P.kmask, C.ma[s], C.mb[s], ts);
                ref[s] = id > 0 ? -3 - id : -1;
            }
        }
    }
}

// bit 2j of the result = bit j of e, bit 2j+1 = bit j of o (wave-uniform scalar work)
__device__ __forceinline__ uint64_t interleave32(uint32_t e, uint32_t o) {
    auto spread = [](uint64_t x) {
        x = (x | (x << 16)) & 0x0000FFFF0000FFFFULL; x = (x | (x << 8)) & 0x00FF00FF00FF00FFULL;
        x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0FULL;  x = (x | (x << 2)) & 0x3333333333333333ULL;
        x = (x | (x << 1)) & 0x5555555555555555ULL;  return x;
    };
    return spread(e) | (spread(o) << 1);
}

// fold 64 consecutive positions' hit mask (bit p <=> position ibase+p, held by lane laneBase+(p>>1), parity p&1) into
// the read's scan state; returns true when the scan of this read is over
template <int MODE>
__device__ __forceinline__ bool fold_hits(const KParams& P, ReadScan& R, int& found, const uint64_t m, const int refE, const int refO,
                                          const int ibase, const int laneBase) {
    if (!m) return false;
    auto ref_at = [&](int p) { const int l = laneBase + (p >> 1); return (p & 1) ? __builtin_amdgcn_readlane(refO, l) : __builtin_amdgcn_readlane(refE, l); };
    if (MODE != BBDUK_MODE_KFILTER) {
        const int fl = __ffsll((unsigned long long)m) - 1, ll = 63 - __clzll((long long)m);
        if (found == 0) { R.iFirst = ibase + fl; R.ref = ref_at(fl); }
        R.iLast = ibase + ll;
        found += __popcll(m);
        return MODE == BBDUK_MODE_KTRIM_R;                       // only minLoc/id0 of the first hit are used
    } else {
        if (P.mcf > 0.f) {                                       // countCoveredBases (:1631-1648): hits in position order
            uint64_t mm = m;
            while (mm) {
                const int p = __ffsll((unsigned long long)mm) - 1, i = ibase + p;
                found += min(P.k, i - R.iLast);
                R.iLast = i;
                if (found >= R.maxBad) { R.ref = ref_at(p); R.iFirst = 0; return true; }
                mm &= mm - 1;
            }
            return false;
        }
        const int c = __popcll(m);
        if (found + c > R.maxBad) {                              // the (maxBadKmers+1)-th hit is in this block
            uint64_t mm = m;
            for (int q = found; q < R.maxBad; q++) mm &= mm - 1;
            const int fl = __ffsll((unsigned long long)mm) - 1;
            R.ref = ref_at(fl);
            found = R.maxBad + 1;
            R.iFirst = 0;                                        // marks the early exit
            return true;
        }
        found += c;
        return false;
    }
}

// Short k-mer lookup of the specialised wave kernel (no middle mask, no query expansion; filter at LDS address 0), in the
// style of cand_probe4: straight-line probe for every lane, one scalar test, and only flagged lanes check keys.
// Returns the id (>0) or -1.
template <bool GENERAL = false>
__device__ __forceinline__ int short_probe(const KParams& P, const uint64_t kmer, const uint64_t rk, const uint64_t lengthMask, bool act) {
    uint64_t mx = (!GENERAL || P.rcomp) ? (kmer > rk ? kmer : rk) : kmer;
    if constexpr (GENERAL) {                                      // (the specialised kernels run without middle mask and speed)
        mx &= P.middleMask;
        if (P.speed > 0) act = act && passes_speed(P, mx | lengthMask);
    }
    const uint32_t ma = mix_a(mx), mb = mix_b(mx);
    uint32_t pv = P.ldsBits ? __builtin_amdgcn_ubfe(lds_word_at(filt_byte(ma, P.ldsBits)), ma, 1u) : 1u;
    pv = act ? pv : 0u;
    const uint64_t t = P.tags[pv ? bucket_of(mb, P.bucketBits) : 0u];
    const uint32_t fp = tag_of(ma), tlo = (uint32_t)t, thi = (uint32_t)(t >> 32);
    const uint64_t b = __ballot((tlo & 0xFFFFu) == fp) | __ballot((tlo >> 16) == fp) | __ballot((thi & 0xFFFFu) == fp) |
                       __ballot((thi >> 16) == fp) | __ballot((int32_t)thi < 0);
    int sref = -1;
    if ((b & __ballot(pv != 0u)) != 0ULL) {                       // rare
        const uint64_t cand = zero16((t & TAG_FPS) ^ ((uint64_t)fp * 0x0001000100010001ULL));
        if (pv && (cand != 0ULL || (t & TAG_CONT))) sref = table_find_t(P, mx | lengthMask, ma, mb, t);
    }
    return sref;
}

// Two short k-mer probes with their filter reads and fingerprint gathers in flight together (the short scans are latency-bound: one
// dependent L2 access per pass otherwise).
template <bool GENERAL = false>
__device__ __forceinline__ void short_probe2(const KParams& P, const uint64_t* kmer, const uint64_t* rk, const uint64_t* lengthMask, const bool* act0, int* sref) {
    uint64_t mx[2], t[2]; uint32_t ma[2], mb[2], pv[2]; bool act[2] = {act0[0], act0[1]};
#pragma unroll
    for (int q = 0; q < 2; q++) {
        mx[q] = (!GENERAL || P.rcomp) ? (kmer[q] > rk[q] ? kmer[q] : rk[q]) : kmer[q];
        if constexpr (GENERAL) {
            mx[q] &= P.middleMask;
            if (P.speed > 0) act[q] = act[q] && passes_speed(P, mx[q] | lengthMask[q]);
        }
        ma[q] = mix_a(mx[q]); mb[q] = mix_b(mx[q]);
        pv[q] = P.ldsBits ? __builtin_amdgcn_ubfe(lds_word_at(filt_byte(ma[q], P.ldsBits)), ma[q], 1u) : 1u;
    }
#pragma unroll
    for (int q = 0; q < 2; q++) { pv[q] = act[q] ? pv[q] : 0u; t[q] = P.tags[pv[q] ? bucket_of(mb[q], P.bucketBits) : 0u]; }
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const uint32_t fp = tag_of(ma[q]), tlo = (uint32_t)t[q], thi = (uint32_t)(t[q] >> 32);
        const uint64_t b = __ballot((tlo & 0xFFFFu) == fp) | __ballot((tlo >> 16) == fp) | __ballot((thi & 0xFFFFu) == fp) |
                           __ballot((thi >> 16) == fp) | __ballot((int32_t)thi < 0);
        sref[q] = -1;
        if ((b & __ballot(pv[q] != 0u)) != 0ULL) {                // rare
            const uint64_t cand = zero16((t[q] & TAG_FPS) ^ ((uint64_t)fp * 0x0001000100010001ULL));
            if (pv[q] && (cand != 0ULL || (t[q] & TAG_CONT))) sref[q] = table_find_t(P, mx[q] | lengthMask[q], ma[q], mb[q], t[q]);
        }
    }
}

// kfilter thresholds that depend on the read (general kernels).  numValidKmers (stream/Read.java:1673-1683), wave-cooperative:
// one lane per k-mer end position, valid = no undefined base in the window.
__device__ __forceinline__ int valid_kmers_wave(const Planes& Q, const int base0, const int L, const int k, const int lane) {
    int cnt = 0;
    for (int i0 = k - 1; i0 < L; i0 += 64) {
        const int i = i0 + lane;
        const bool v = (i < L) && extract1(Q.nm, base0 + min(i, L - 1) - k + 1, k) == 0u;
        cnt += __popcll(__ballot(v));
    }
    return cnt;
}
// The same count for any k (keff = kbig may exceed a 32-bit window): 64 positions per step, the undefined bases of a
// step split it into defined segments; a segment that takes the run of defined bases from r0 to r0+seg adds the
// positions whose run length reaches k.
__device__ __forceinline__ int valid_kmers_any_k(const Planes& Q, const int base0, const int L, const int k, const int lane) {
    int cnt = 0, run = 0;
    for (int i0 = 0; i0 < L; i0 += 64) {
        const int b = base0 + min(i0 + lane, L - 1);
        const uint64_t U = __ballot(((Q.nm[b >> 5] >> (b & 31)) & 1u) != 0u);
        const int nv = min(64, L - i0);
        int pos = 0;
        while (pos < nv) {
            const uint64_t rest = U >> pos;
            const int nextU = rest ? min(nv, pos + __ffsll((unsigned long long)rest) - 1) : nv;
            const int seg = nextU - pos;
            cnt += max(0, run + seg - max(run, k - 1));
            run += seg;
            if (nextU < nv) { run = 0; pos = nextU + 1; } else pos = nv;
        }
    }
    return cnt;
}
// maxBadKmersR (bbduk/BBDukProcessorS.java:1055-1062) or minCoveredBases (:1040,1045) of one read
__device__ __forceinline__ int kfilter_threshold(const KParams& P, const Planes& Q, const int base0, const int L, const int lane) {
    if (P.mcf > 0.f) return (int)ceilf(P.mcf * (float)L);
    if (P.mkf != 0.f) {
        const int vk = (L >= P.k) ? valid_kmers_wave(Q, base0, L, P.k, lane) : 0;
        return max(P.maxBadKmers, (int)((float)(vk - 1) * P.mkf));
    }
    return P.maxBadKmers;
}

// firstA >= 0 (bbduk_long_kernel): read A is scanned in chunks; this call resumes at position firstA with A.found hits so far.
// MASK (ktrim=n): every position that matches -- with kmaskfullycovered every position of the span that does NOT -- sets its bit of
// `hitPlane` (plane coordinates, like the undefined-plane), see bbduk_kmask_kernel.
// FAST (wave kernel only: the filter sits at LDS address 0): lookup4_probe instead of lookup4.
template <int MODE, bool FORBIDN, bool GENERAL, bool BIG = false, bool SPAN = false, bool MASK = false, bool FAST = false>
__device__ __forceinline__ void main_scan_pair(const KParams& P, const Planes& Q, ReadScan& A, ReadScan& B, const int lane, const int firstA = -1,
                                               uint32_t* hitPlane = nullptr) {
    ReadWin WA, WB;
    win_init<FORBIDN, GENERAL, false, SPAN>(P, Q, A, WA, lane);
    win_init<FORBIDN, GENERAL, false, SPAN>(P, Q, B, WB, lane);
    if (firstA >= 0) { WA.first = max(WA.first, firstA); WA.on = A.scan && WA.first < WA.stop; }
    int ibA = WA.first, ibB = WB.first, foundA = (firstA >= 0) ? A.found : 0, foundB = 0;
    bool onA = WA.on, onB = WB.on;
    while (onA || onB) {
        uint64_t kmer[4], rk[4]; bool ok[4]; int id[4];
        windows2<FORBIDN, GENERAL, SPAN>(P, Q, WA, ibA + 2 * lane, onA, kmer, rk, ok);
        windows2<FORBIDN, GENERAL, SPAN>(P, Q, WB, ibB + 2 * lane, onB, kmer + 2, rk + 2, ok + 2);
        if constexpr (FAST && !BIG) lookup4_probe<GENERAL>(P, Q.filt, kmer, rk, ok, id);
        else lookup4<GENERAL, BIG>(P, Q.filt, kmer, rk, ok, id);
        const uint64_t m0 = __ballot(id[0] != -1), m1 = __ballot(id[1] != -1), m2 = __ballot(id[2] != -1), m3 = __ballot(id[3] != -1);
        if constexpr (MASK) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const ReadWin& W = q < 2 ? WA : WB;
                const int i = (q < 2 ? ibA : ibB) + 2 * lane + (q & 1);
                const bool mark = P.mfc ? ((q < 2 ? onA : onB) && i < W.stop && id[q] == -1) : (id[q] != -1);
                if (mark) atomicOr(&hitPlane[(W.base0 + i) >> 5], 1u << ((W.base0 + i) & 31));
            }
        }
        if (onA) {
            bool ex = false;
            if (m0 | m1) {
                ex = fold_hits<MODE>(P, A, foundA, interleave32((uint32_t)m0, (uint32_t)m1), id[0], id[1], ibA, 0);
                if (!ex) ex = fold_hits<MODE>(P, A, foundA, interleave32((uint32_t)(m0 >> 32), (uint32_t)(m1 >> 32)), id[0], id[1], ibA + 64, 32);
            }
            ibA += 128;
            onA = !ex && ibA < WA.stop;
        }
        if (onB) {
            bool ex = false;
            if (m2 | m3) {
                ex = fold_hits<MODE>(P, B, foundB, interleave32((uint32_t)m2, (uint32_t)m3), id[2], id[3], ibB, 0);
                if (!ex) ex = fold_hits<MODE>(P, B, foundB, interleave32((uint32_t)(m2 >> 32), (uint32_t)(m3 >> 32)), id[2], id[3], ibB + 64, 32);
            }
            ibB += 128;
            onB = !ex && ibB < WB.stop;
        }
    }
    A.found = foundA; B.found = foundB;
}

// The left pass of ktrim=rl scans about half a read (:1821-1824: [0, mid+k-1)), i.e. at most 64 positions of a 150-base read where the pair
// scan above gives every read 128: FOUR reads share a block here.  R[0] and R[1] sit in lanes 0-31 of the slot pairs (0,1) and (2,3), R[2] and
// R[3] in lanes 32-63; the windows are cut with lane-varying read coordinates and each read folds its half of the ballots.  Spans that start
// at 0 and are looked up at every position only (W.full).
template <bool FORBIDN, bool GENERAL>
__device__ __forceinline__ void left_scan_quad(const KParams& P, const Planes& Q, ReadScan* R, const int lane) {
    const bool hi = lane >= 32; const int l5 = lane & 31;
    ReadWin W[2];
#pragma unroll
    for (int sp = 0; sp < 2; sp++) {
        W[sp].base0 = hi ? R[2 + sp].base0 : R[sp].base0; W[sp].stop = hi ? R[2 + sp].stop : R[sp].stop;
        W[sp].start = 0; W[sp].first = P.k - 1; W[sp].on = true; W[sp].full = true;
        W[sp].hasN = (FORBIDN && P.forbidNs) && (R[sp].hasN != 0 || R[2 + sp].hasN != 0);
    }
    int found[4] = {0, 0, 0, 0};
    for (int ib = P.k - 1; ; ib += 64) {
        bool on[4];
#pragma unroll
        for (int q = 0; q < 4; q++) on[q] = R[q].scan && ib < R[q].stop;
        if (!(on[0] || on[1] || on[2] || on[3])) break;
        uint64_t kmer[4], rk[4]; bool ok[4]; int id[4];
        windows2<FORBIDN, GENERAL, false>(P, Q, W[0], ib + 2 * l5, hi ? on[2] : on[0], kmer, rk, ok);
        windows2<FORBIDN, GENERAL, false>(P, Q, W[1], ib + 2 * l5, hi ? on[3] : on[1], kmer + 2, rk + 2, ok + 2);
        lookup4_probe<GENERAL>(P, Q.filt, kmer, rk, ok, id);
        const uint64_t m0 = __ballot(id[0] != -1), m1 = __ballot(id[1] != -1), m2 = __ballot(id[2] != -1), m3 = __ballot(id[3] != -1);
        if (!(m0 | m1 | m2 | m3)) continue;
        fold_hits<BBDUK_MODE_KTRIM_L>(P, R[0], found[0], interleave32((uint32_t)m0, (uint32_t)m1), id[0], id[1], ib, 0);
        fold_hits<BBDUK_MODE_KTRIM_L>(P, R[2], found[2], interleave32((uint32_t)(m0 >> 32), (uint32_t)(m1 >> 32)), id[0], id[1], ib, 32);
        fold_hits<BBDUK_MODE_KTRIM_L>(P, R[1], found[1], interleave32((uint32_t)m2, (uint32_t)m3), id[2], id[3], ib, 0);
        fold_hits<BBDUK_MODE_KTRIM_L>(P, R[3], found[3], interleave32((uint32_t)(m2 >> 32), (uint32_t)(m3 >> 32)), id[2], id[3], ib, 32);
    }
#pragma unroll
    for (int q = 0; q < 4; q++) R[q].found = found[q];
}

// Candidate form of the pair scan (ktrim=r, kfilter with maxbadkmers=0: only the first hit of a read matters): stop
// at the first fingerprint match of each read WITHOUT fetching its key; the caller verifies the candidates of a whole
// sub-tile in one overlapped batch (one lane per read) and falls back to main_scan_pair for the rare impostor.
template <bool FORBIDN, bool GENERAL, bool NOMM, bool BIG = false, bool SPAN = false>
__device__ __forceinline__ void main_scan_pair_cand(const KParams& P, const Planes& Q, ReadScan& A, ReadScan& B, const int lane, const int firstA = -1) {
    ReadWin WA, WB;
    win_init<FORBIDN, GENERAL, BIG, SPAN>(P, Q, A, WA, lane);
    win_init<FORBIDN, GENERAL, BIG, SPAN>(P, Q, B, WB, lane);
    if (firstA >= 0) { WA.first = max(WA.first, firstA); WA.on = A.scan && WA.first < WA.stop; }     // resume behind an impostor
    int ibA = WA.first, ibB = WB.first;
    bool onA = WA.on, onB = WB.on;
    A.candSlot = -1; B.candSlot = -1;
    constexpr int STEP = BIG ? BIG_STEP : 128;                   // big layout: the top lanes' windows lack their successors (cand_probe4_big)
    // Lanes past a read's end look up whatever lies behind it in the planes.  When every window of the read is plain
    // nothing masks them: positions grow with the lane, so a first candidate at a position >= stop means the read has
    // none.  Reads with cut or reset windows (restrictRight, an undefined base) mask their ballots with `ok` instead.
    const bool thin = GENERAL && SPAN && P.qskip > 1;             // qskip: `ok` thins the positions
    const bool plainA = WA.full && !thin && !((FORBIDN || BIG) && WA.hasN) && !(SPAN && WA.start > 0), plainB = WB.full && !thin && !((FORBIDN || BIG) && WB.hasN) && !(SPAN && WB.start > 0);
    while (onA || onB) {
        uint64_t kmer[4], rk[4], rkRaw[4]; bool ok[4]; int ref[4]; Cand4 C;
        windows2<FORBIDN, GENERAL, SPAN>(P, Q, WA, ibA + 2 * lane, onA, kmer, rk, ok, BIG ? rkRaw : nullptr);
        windows2<FORBIDN, GENERAL, SPAN>(P, Q, WB, ibB + 2 * lane, onB, kmer + 2, rk + 2, ok + 2, BIG ? rkRaw + 2 : nullptr);
#ifdef BBDUK_TIMING_SWITCHES
        if (TSW(P, 8)) {                                             // experiment: 16 extra dependent-free VALU ops per block
            uint32_t z0 = (uint32_t)kmer[0], z1 = (uint32_t)kmer[1], z2 = (uint32_t)kmer[2], z3 = (uint32_t)kmer[3];
#pragma unroll
            for (int q = 0; q < 4; q++) { asm volatile("v_xor_b32 %0, %0, %1" : "+v"(z0) : "v"(z1)); asm volatile("v_xor_b32 %0, %0, %1" : "+v"(z1) : "v"(z2));
                                          asm volatile("v_xor_b32 %0, %0, %1" : "+v"(z2) : "v"(z3)); asm volatile("v_xor_b32 %0, %0, %1" : "+v"(z3) : "v"(z0)); }
            if (z0 == 0x12345u && z1 == z2 && z3 == 7u) kmer[0] ^= 1;
        }
        if (TSW(P, 9)) {                                             // experiment: 16 extra SALU ops per block
            uint32_t u0 = (uint32_t)__builtin_amdgcn_readfirstlane(ibA), u1 = (uint32_t)__builtin_amdgcn_readfirstlane(ibB);
#pragma unroll
            for (int q = 0; q < 8; q++) { asm volatile("s_xor_b32 %0, %0, %1" : "+s"(u0) : "s"(u1)); asm volatile("s_add_u32 %0, %0, %1" : "+s"(u1) : "s"(u0) : "scc"); }
            if (u0 == 0x12345u && u1 == 99u) kmer[0] ^= 1;
        }
#endif
        Cand4Big CB; uint64_t anyFlag; const uint64_t* key;
        if constexpr (BIG) {
            bool nf[4] = {false, false, false, false};
            if ((WA.hasN | WB.hasN) && P.gW > 0) {                   // which windows see an undefined base (bit t of nw <=> base i-k+1+t)
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    const ReadWin& W = r ? WB : WA;
                    if (!W.hasN) continue;
                    const int nidx = min(W.base0 - P.k + 1 + (r ? ibB : ibA) + 2 * lane, Q.T);
                    const uint32_t nw = __builtin_amdgcn_alignbit(Q.nm[(nidx >> 5) + 1], Q.nm[nidx >> 5], nidx);
                    const uint32_t km = (P.k >= 32) ? ~0u : ((1u << P.k) - 1u);
                    nf[2 * r] = ok[2 * r] && (nw & km) != 0u; nf[2 * r + 1] = ok[2 * r + 1] && ((nw >> 1) & km) != 0u;
                }
            }
            anyFlag = cand_probe4_big<NOMM>(P, kmer, rk, rkRaw, nf, CB);
            key = CB.key;
        } else {
            anyFlag = cand_probe4<GENERAL, NOMM, true>(P, Q.filt, kmer, rk, C);
            key = C.key;
        }
        if (anyFlag == 0ULL) {                                       // the common block: nothing to look at
            if (onA) { ibA += STEP; onA = ibA < WA.stop; }
            if (onB) { ibB += STEP; onB = ibB < WB.stop; }
            continue;
        }
        if constexpr (BIG) cand_resolve4_big(P, CB, ref); else cand_resolve4(P, C, ref);
#pragma unroll
        for (int r = 0; r < 2; r++) {
            ReadScan& R = r ? B : A;
            bool& on = r ? onB : onA;
            int& ib = r ? ibB : ibA;
            if (!on) continue;
            uint64_t me = __ballot(ref[2 * r] != -1), mo = __ballot(ref[2 * r + 1] != -1);
            if (!(r ? plainB : plainA)) { me &= __ballot(ok[2 * r]); mo &= __ballot(ok[2 * r + 1]); }
            const int stopR = r ? WB.stop : WA.stop;
            const int le = me ? __ffsll((unsigned long long)me) - 1 : 64, lo = mo ? __ffsll((unsigned long long)mo) - 1 : 64;
            const int h = (2 * lo + 1 < 2 * le) ? 1 : 0;
            const int l = h ? lo : le;
            if ((me | mo) && ib + 2 * l + h < stopR && (!BIG || 2 * l + h < BIG_STEP)) {   // first candidate in position order: lane l, parity h
                const int rs = h ? ref[2 * r + 1] : ref[2 * r];
                const uint64_t ks = h ? key[2 * r + 1] : key[2 * r];
                R.candSlot = __builtin_amdgcn_readlane(rs, l);
                if constexpr (BIG) {                             // way 0..3 of the primary word, 4..7 = of the alternate
                    const int sl = 2 * r + h;
                    const uint32_t ws = (rs >= 4) ? (h ? CB.w2[2 * r + 1] : CB.w2[2 * r]) : (h ? CB.w1[2 * r + 1] : CB.w1[2 * r]);
                    (void)sl;
                    R.candWord = (uint32_t)__builtin_amdgcn_readlane((int)ws, l);
                    if (R.candSlot >= 4) R.candSlot -= 4;
                }
                R.candKeyLo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)ks, l);
                R.candKeyHi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(ks >> 32), l);
                R.iFirst = ib + 2 * l + h;
                on = false;
            } else {
                ib += STEP;
                on = ib < stopR;
            }
        }
    }
}

// The candidate scan for reads whose spans are short: up to FOUR reads share one block.  The right pass of ktrim=rl looks at 76 positions of
// a 150-base read (38 lanes) where the pair scan gives every read 128, so three reads fit the 128 lane-slots of a block (two slot pairs of 64
// lanes): read r owns the lane-slots [T[r], T[r+1]) in the order (slot pair, lane), two adjacent positions per lane-slot from first[r] on, and may
// straddle the two slot pairs.  The windows are cut with lane-varying read coordinates; every read then takes its first candidate out of its
// part of the ballots.  One block serves all positions (the caller groups the reads so that they fit); the spans' windows are all "full"
// (every position looked up, SPAN cuts in front of `start`).
// (T[0] < 0 and T[4] > 128 are allowed -- a read that starts in an earlier block or goes on in the next: the lane-slot packing tried for
// 2x151 reads, DESIGN 4.1 "Read length"; a packed block costs 1.43x a pair block, so only ktrim=rl's right pass uses this scan.)
template <bool FORBIDN, bool GENERAL, bool NOMM = false, bool SPAN = true>
__device__ __forceinline__ void packed_scan_cand(const KParams& P, const Planes& Q, const int ra, const int nr, const int* T, const bool anyN,
                                                 const int vBase0, const int vStart, const int vStop, const int vFirstLook, const int lane,
                                                 int& vCSlot, int& vFirst, uint32_t& vCKeyLo, uint32_t& vCKeyHi) {
    // (the reads' coordinates come from their lanes by shuffle: keeping them wave-uniform for four reads costs more SGPRs than the kernel has)
    ReadWin W[2]; int pos[2]; bool val[2];
#pragma unroll
    for (int sp = 0; sp < 2; sp++) {
        const int t = 64 * sp + lane;
        const int r = (t >= T[1] ? 1 : 0) + (t >= T[2] ? 1 : 0) + (t >= T[3] ? 1 : 0);
        val[sp] = t < T[4];
        const int src = min(ra + r, 63);
        W[sp].base0 = __shfl(vBase0, src); W[sp].start = __shfl(vStart, src); W[sp].stop = __shfl(vStop, src);
        const int f = __shfl(vFirstLook, src);
        const int tr = r == 0 ? T[0] : (r == 1 ? T[1] : (r == 2 ? T[2] : T[3]));
        pos[sp] = f + 2 * (t - tr);
        W[sp].first = f; W[sp].on = true; W[sp].full = true; W[sp].hasN = anyN;
    }
    uint64_t kmer[4], rk[4]; bool ok[4]; int ref[4]; Cand4 C;
    windows2<FORBIDN, GENERAL, SPAN>(P, Q, W[0], pos[0], val[0], kmer, rk, ok);
    windows2<FORBIDN, GENERAL, SPAN>(P, Q, W[1], pos[1], val[1], kmer + 2, rk + 2, ok + 2);
    const uint64_t anyFlag = cand_probe4<GENERAL, NOMM, true>(P, Q.filt, kmer, rk, C);
    if (anyFlag == 0ULL) return;                                  // the common block: nothing to look at
    cand_resolve4(P, C, ref);
    const uint64_t m[4] = {__ballot(ref[0] != -1 && ok[0]), __ballot(ref[1] != -1 && ok[1]), __ballot(ref[2] != -1 && ok[2]), __ballot(ref[3] != -1 && ok[3])};
    if (!(m[0] | m[1] | m[2] | m[3])) return;
#pragma unroll
    for (int r = 0; r < 4; r++) {                                 // (compile-time indices: the arrays stay in registers)
        if (r >= nr) continue;
        bool found = false;
#pragma unroll
        for (int sp = 0; sp < 2; sp++) {
            const int lo = max(T[r], 64 * sp) - 64 * sp, hi = min(T[r + 1], 64 * sp + 64) - 64 * sp;
            if (found || lo >= hi) continue;
            const uint64_t mask = (hi >= 64 ? ~0ULL : ((1ULL << hi) - 1ULL)) & ~((1ULL << lo) - 1ULL);
            const uint64_t me = m[2 * sp] & mask, mo = m[2 * sp + 1] & mask;
            if (!(me | mo)) continue;
            const int le = me ? __ffsll((unsigned long long)me) - 1 : 64, lq = mo ? __ffsll((unsigned long long)mo) - 1 : 64;
            const int h = (2 * lq + 1 < 2 * le) ? 1 : 0;
            const int l = h ? lq : le;
            const int rs = h ? ref[2 * sp + 1] : ref[2 * sp];
            const uint64_t ks = h ? C.key[2 * sp + 1] : C.key[2 * sp];
            const int cs = __builtin_amdgcn_readlane(rs, l);
            const uint32_t klo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)ks, l), khi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(ks >> 32), l);
            const int ifirst = __builtin_amdgcn_readlane(vFirstLook, min(ra + r, 63)) + 2 * (64 * sp + l - T[r]) + h;
            vCSlot = (lane == ra + r) ? cs : vCSlot; vFirst = (lane == ra + r) ? ifirst : vFirst;
            vCKeyLo = (lane == ra + r) ? klo : vCKeyLo; vCKeyHi = (lane == ra + r) ? khi : vCKeyHi;
            found = true;
        }
    }
}

// The candidate scan for short reads (wave_body<.., SHAPE>): THREE reads share a block when none of them has more than TRI_MAX k-mer
// end positions (a 100-base read with k=23 has 78 and leaves 39 % of the pair scan's lanes idle).  R[0] and R[1] take lanes 0-41 of the two
// slot pairs (84 positions each), R[2] takes lanes 42-63 of both (44 + 44 positions); the read coordinates stay wave-uniform per lane
// group (two selects per field, no shuffles: that is what made lane-slot packing too dear).  One block serves all three reads.
#define TRI_LANES 42
#define TRI_MAX   (2 * TRI_LANES)
template <bool FORBIDN, bool GENERAL, bool NOMM, bool SPAN>
__device__ __forceinline__ void tri_scan_cand(const KParams& P, const Planes& Q, ReadScan* R, const int lane) {
    const bool hiL = lane >= TRI_LANES;
    const int lt = hiL ? lane - TRI_LANES : lane;
    const int f0 = max(R[0].start, P.k - 1), f1 = max(R[1].start, P.k - 1), f2 = max(R[2].start, P.k - 1);
    ReadWin W[2]; int pos[2]; bool on[2];
    W[0].base0 = hiL ? R[2].base0 : R[0].base0; W[0].stop = hiL ? R[2].stop : R[0].stop; W[0].start = hiL ? R[2].start : R[0].start;
    W[1].base0 = hiL ? R[2].base0 : R[1].base0; W[1].stop = hiL ? R[2].stop : R[1].stop; W[1].start = hiL ? R[2].start : R[1].start;
    pos[0] = (hiL ? f2 : f0) + 2 * lt; pos[1] = (hiL ? f2 + 2 * (64 - TRI_LANES) : f1) + 2 * lt;
    on[0] = hiL ? R[2].scan : R[0].scan; on[1] = hiL ? R[2].scan : R[1].scan;
#pragma unroll
    for (int sp = 0; sp < 2; sp++) {
        W[sp].first = P.k - 1; W[sp].on = true; W[sp].full = true;
        W[sp].hasN = (FORBIDN && P.forbidNs) && (R[0].hasN != 0 || R[1].hasN != 0 || R[2].hasN != 0);
    }
    R[0].candSlot = -1; R[1].candSlot = -1; R[2].candSlot = -1;
    uint64_t kmer[4], rk[4]; bool ok[4]; int ref[4]; Cand4 C;
    windows2<FORBIDN, GENERAL, SPAN>(P, Q, W[0], pos[0], on[0], kmer, rk, ok);
    windows2<FORBIDN, GENERAL, SPAN>(P, Q, W[1], pos[1], on[1], kmer + 2, rk + 2, ok + 2);
    const uint64_t anyFlag = cand_probe4<GENERAL, NOMM, true>(P, Q.filt, kmer, rk, C);
    if (anyFlag == 0ULL) return;                                  // the common block: nothing to look at
    cand_resolve4(P, C, ref);
    const uint64_t m0e = __ballot(ref[0] != -1 && ok[0]), m0o = __ballot(ref[1] != -1 && ok[1]);
    const uint64_t m1e = __ballot(ref[2] != -1 && ok[2]), m1o = __ballot(ref[3] != -1 && ok[3]);
    if (!(m0e | m0o | m1e | m1o)) return;
    const uint64_t LOW = (1ULL << TRI_LANES) - 1ULL;
    // first candidate in position order among the lanes [lb, lb+..) of one slot pair: lane l, parity h
    auto take = [&](ReadScan& T, const uint64_t me, const uint64_t mo, const int refE, const int refO, const uint64_t keyE, const uint64_t keyO,
                    const int lb, const int posBase) {
        const int le = me ? __ffsll((unsigned long long)me) - 1 : 64, lo = mo ? __ffsll((unsigned long long)mo) - 1 : 64;
        const int h = (2 * lo + 1 < 2 * le) ? 1 : 0;
        const int l = h ? lo : le;
        const int rs = h ? refO : refE;
        const uint64_t ks = h ? keyO : keyE;
        T.candSlot = __builtin_amdgcn_readlane(rs, l);
        T.candKeyLo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)ks, l);
        T.candKeyHi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(ks >> 32), l);
        T.iFirst = posBase + 2 * (l - lb) + h;
    };
    if ((m0e | m0o) & LOW) take(R[0], m0e & LOW, m0o & LOW, ref[0], ref[1], C.key[0], C.key[1], 0, f0);
    if ((m1e | m1o) & LOW) take(R[1], m1e & LOW, m1o & LOW, ref[2], ref[3], C.key[2], C.key[3], 0, f1);
    if ((m0e | m0o) & ~LOW) take(R[2], m0e & ~LOW, m0o & ~LOW, ref[0], ref[1], C.key[0], C.key[1], TRI_LANES, f2);
    else if ((m1e | m1o) & ~LOW) take(R[2], m1e & ~LOW, m1o & ~LOW, ref[2], ref[3], C.key[2], C.key[3], TRI_LANES, f2 + 2 * (64 - TRI_LANES));
}

// maximum of v over the lanes of a wave, returned wave-uniform
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
    return __builtin_amdgcn_readfirstlane(v);
}

// The tail pass of the candidate scans (ktrim=r, kfilter): reads whose span overshoots the pair scan's last full block by a few positions
// (vTail of them, 1..32) have those positions looked up here, many reads per block: LPR lanes per read (4, 8 or 16, by the longest tail of
// the sub-tile), two adjacent positions per lane, two groups of 64 lane-slots per block.  Only reads without a candidate so far take part
// (their first candidate is all that matters); `sel` is the wave's scratch list in LDS.
template <bool FORBIDN, bool GENERAL, bool NOMM, bool SPAN>
__device__ __forceinline__ void tail_scan_cand(const KParams& P, const Planes& Q, uint8_t* sel, const bool anyN, const int vBase0, const int vStart, const int vStop,
                                               const int vTail, const int lane, int& vCSlot, int& vFirst, uint32_t& vCKeyLo, uint32_t& vCKeyHi) {
    const bool need = vTail > 0 && vCSlot == -1;
    const uint64_t needM = __ballot(need);
    if (!needM) return;
    const int rank = __popcll(needM & ((1ULL << lane) - 1ULL));
    if (need) sel[rank] = (uint8_t)lane;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int nneed = __popcll(needM);
    const int tmax = wave_max_i(need ? vTail : 0);
    const int sh = tmax <= 8 ? 2 : (tmax <= 16 ? 3 : 4);        // log2(lanes per read)
    const int rpg = 64 >> sh;                                     // reads per group of 64 lane-slots
    const int u = lane & ((1 << sh) - 1);
    for (int pb = 0; pb < nneed; pb += 2 * rpg) {
        ReadWin W[2]; int pos[2]; bool val[2];
#pragma unroll
        for (int sp = 0; sp < 2; sp++) {
            const int idx = pb + sp * rpg + (lane >> sh);
            val[sp] = idx < nneed;
            const int src = sel[min(idx, nneed - 1)];
            W[sp].base0 = __shfl(vBase0, src); W[sp].start = __shfl(vStart, src); W[sp].stop = __shfl(vStop, src);
            pos[sp] = W[sp].stop - __shfl(vTail, src) + 2 * u;
            W[sp].first = P.k - 1; W[sp].on = true; W[sp].full = true; W[sp].hasN = anyN;
        }
        uint64_t kmer[4], rk[4]; bool ok[4]; int ref[4]; Cand4 C;
        windows2<FORBIDN, GENERAL, SPAN>(P, Q, W[0], pos[0], val[0], kmer, rk, ok);
        windows2<FORBIDN, GENERAL, SPAN>(P, Q, W[1], pos[1], val[1], kmer + 2, rk + 2, ok + 2);
        const uint64_t anyFlag = cand_probe4<GENERAL, NOMM, true>(P, Q.filt, kmer, rk, C);
        if (anyFlag == 0ULL) continue;
        cand_resolve4(P, C, ref);
#pragma unroll
        for (int sp = 0; sp < 2; sp++) {
            const uint64_t me = __ballot(ref[2 * sp] != -1 && ok[2 * sp]), mo = __ballot(ref[2 * sp + 1] != -1 && ok[2 * sp + 1]);
            uint64_t mm = me | mo;
            while (mm) {                                          // rare: a read with a candidate in its tail
                const int g = (__ffsll((unsigned long long)mm) - 1) >> sh;
                const uint64_t gm = ((sh == 4 ? 0xFFFFULL : (sh == 3 ? 0xFFULL : 0xFULL)) << (g << sh));
                const uint64_t ge = me & gm, go = mo & gm;
                const int le = ge ? __ffsll((unsigned long long)ge) - 1 : 64, lo = go ? __ffsll((unsigned long long)go) - 1 : 64;
                const int h = (2 * lo + 1 < 2 * le) ? 1 : 0;
                const int l = h ? lo : le;
                const int rs = h ? ref[2 * sp + 1] : ref[2 * sp];
                const uint64_t ks = h ? C.key[2 * sp + 1] : C.key[2 * sp];
                const int cs = __builtin_amdgcn_readlane(rs, l);
                const uint32_t klo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)ks, l), khi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(ks >> 32), l);
                const int ifirst = __builtin_amdgcn_readlane(pos[sp], l) + h;
                const int jr = sel[pb + sp * rpg + g];
                vCSlot = (lane == jr) ? cs : vCSlot; vFirst = (lane == jr) ? ifirst : vFirst;
                vCKeyLo = (lane == jr) ? klo : vCKeyLo; vCKeyHi = (lane == jr) ? khi : vCKeyHi;
                mm &= ~gm;
            }
        }
    }
}

// The tail pass of the every-hit scans (ktrim=l, ksplit): the same lane layout as tail_scan_cand, exact lookups, and every read with a hit in
// its tail folds it into its scan facts (fold_hits: count, first / last position, id of the first hit) -- the tails lie behind everything
// the pair scan saw, so position order is kept.
// MASK (ktrim=n): the tails' positions also set their bits of the hit plane, as main_scan_pair<.., MASK> does for the positions it sees.
template <int MODE, bool FORBIDN, bool GENERAL, bool MASK = false>
__device__ __forceinline__ void tail_scan_hits(const KParams& P, const Planes& Q, uint8_t* sel, const bool anyN, const int vBase0, const int vStart, const int vStop,
                                               const int vTail, const int lane, int& vFound, int& vFirst, int& vLast, int& vRef, uint32_t* hitPlane = nullptr,
                                               const int s0 = 0, const int e0 = 0) {
    const bool need = vTail > 0;
    const uint64_t needM = __ballot(need);
    if (!needM) return;
    // the reads that take part: a compact list in `sel` -- except for ktrim=n, whose list would lie on the undefined-plane the windows
    // still read: there the reads [s0, e0) of the sub-tile are walked as they stand (sixteen 151-base reads: one or two blocks either way)
    int nneed;
    if constexpr (!MASK) {
        const int rank = __popcll(needM & ((1ULL << lane) - 1ULL));
        if (need) sel[rank] = (uint8_t)lane;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        nneed = __popcll(needM);
    } else nneed = e0 - s0;
    const int tmax = wave_max_i(need ? vTail : 0);
    const int sh = tmax <= 8 ? 2 : (tmax <= 16 ? 3 : 4);        // log2(lanes per read)
    const int rpg = 64 >> sh;
    const int u = lane & ((1 << sh) - 1);
    for (int pb = 0; pb < nneed; pb += 2 * rpg) {
        ReadWin W[2]; int pos[2]; bool val[2];
#pragma unroll
        for (int sp = 0; sp < 2; sp++) {
            const int idx = pb + sp * rpg + (lane >> sh);
            int src;
            if constexpr (!MASK) src = sel[min(idx, nneed - 1)]; else src = min(s0 + idx, 63);
            const int tl = __shfl(vTail, src);
            val[sp] = idx < nneed && tl > 0;
            W[sp].base0 = __shfl(vBase0, src); W[sp].start = __shfl(vStart, src); W[sp].stop = __shfl(vStop, src);
            pos[sp] = W[sp].stop - tl + 2 * u;
            W[sp].first = P.k - 1; W[sp].on = true; W[sp].full = true; W[sp].hasN = anyN;
        }
        if constexpr (MASK) { if (__ballot(val[0] || val[1]) == 0ULL) continue; }
        uint64_t kmer[4], rk[4]; bool ok[4]; int id[4];
        windows2<FORBIDN, GENERAL, GENERAL>(P, Q, W[0], pos[0], val[0], kmer, rk, ok);
        windows2<FORBIDN, GENERAL, GENERAL>(P, Q, W[1], pos[1], val[1], kmer + 2, rk + 2, ok + 2);
        lookup4_probe<GENERAL>(P, Q.filt, kmer, rk, ok, id);
        if constexpr (MASK) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int sp = q >> 1, i = pos[sp] + (q & 1);
                const bool mark = P.mfc ? (val[sp] && i < W[sp].stop && id[q] == -1) : (id[q] != -1);
                if (mark) atomicOr(&hitPlane[(W[sp].base0 + i) >> 5], 1u << ((W[sp].base0 + i) & 31));
            }
        }
#pragma unroll
        for (int sp = 0; sp < 2; sp++) {
            const uint64_t me = __ballot(id[2 * sp] != -1), mo = __ballot(id[2 * sp + 1] != -1);
            uint64_t mm = me | mo;
            while (mm) {                                          // rare: a read with hits in its tail
                const int g = (__ffsll((unsigned long long)mm) - 1) >> sh;
                const int l0 = g << sh;
                const uint32_t lm = sh == 4 ? 0xFFFFu : (sh == 3 ? 0xFFu : 0xFu);
                const uint64_t m = interleave32((uint32_t)(me >> l0) & lm, (uint32_t)(mo >> l0) & lm);
                const int jr = MASK ? (s0 + pb + sp * rpg + g) : __builtin_amdgcn_readfirstlane((int)sel[pb + sp * rpg + g]);
                ReadScan R;
                int found = __builtin_amdgcn_readlane(vFound, jr);
                R.iFirst = __builtin_amdgcn_readlane(vFirst, jr); R.iLast = __builtin_amdgcn_readlane(vLast, jr); R.ref = __builtin_amdgcn_readlane(vRef, jr);
                fold_hits<MODE>(P, R, found, m, id[2 * sp], id[2 * sp + 1], __builtin_amdgcn_readlane(pos[sp], l0), l0);
                vFound = (lane == jr) ? found : vFound; vFirst = (lane == jr) ? R.iFirst : vFirst;
                vLast = (lane == jr) ? R.iLast : vLast; vRef = (lane == jr) ? R.ref : vRef;
                mm &= ~((uint64_t)lm << l0);
            }
        }
    }
}

// Short k-mer scans of two reads in one pass: lanes 0-31 serve read A, lanes 32-63 read B, one lane per
// length mink..  (bbduk/BBDukProcessorS.java:2034-2103).  Only reads whose main scan found nothing take part.
template <int MODE, bool GENERAL>
__device__ __forceinline__ void short_scan_pair(const KParams& P, const Planes& Q, ReadScan& A, ReadScan& B, const int lane) {
    const bool needA = A.scan && A.found == 0, needB = B.scan && B.found == 0;
    if (!needA && !needB) return;
    const int k = P.k;
    const bool hiHalf = lane >= 32;
    const bool need = hiHalf ? needB : needA;
    const int base0 = hiHalf ? B.base0 : A.base0;
    const int start = hiHalf ? B.start : A.start;
    const int stop  = hiHalf ? B.stop : A.stop;
    const int Ls = P.mink + (lane & 31);
    int id = -1;
    if (MODE == BBDUK_MODE_KTRIM_L) {
        const int Lmax = min(k, stop) - start;                   // lengths 1..Lmax, i = start+Ls-1
        bool act = need && Ls <= Lmax;
        if constexpr (GENERAL) { if (P.qskip > 1) act = act && ((start + Ls - 1) % P.qskip) == 0; }
        const int Lc = act ? Ls : 1;
        uint64_t kmer = 0, rk = 0;
        if (act) {
            kmer = extract2(Q.fwd, Q.T - 1 - (base0 + start + Lc - 1), Lc) & P.mask;
            rk   = extract2(Q.cmp, base0 + start, Lc);
        }
        id = lookup<GENERAL>(P, Q.filt, kmer, rk, 1ULL << (2 * Lc), Lc, P.qhdist2, act);
    } else {
        const int Lmax = (stop >= k ? k - 1 : stop);             // lengths 1..Lmax, i = stop-Ls
        bool act = need && Ls <= Lmax;
        if constexpr (GENERAL) { if (P.qskip > 1) act = act && ((stop - Ls) % P.qskip) == 0; }
        const int Lc = act ? Ls : 1;
        uint64_t kmer = 0, rk = 0;
        if (act) {
            kmer = extract2(Q.fwd, Q.T - 1 - (base0 + stop - 1), Lc);            // base stop-1 in bits 0-1
            rk   = extract2(Q.cmp, base0 + stop - Lc, Lc) & P.mask;              // base i in bits 0-1
        }
        id = lookup<GENERAL>(P, Q.filt, kmer, rk, 1ULL << (2 * Lc), Lc, P.qhdist2, act);
    }
    const uint64_t m = __ballot(id != -1);
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const uint32_t mh = (uint32_t)(m >> (32 * h));
        if (!mh) continue;
        ReadScan& R = h ? B : A;
        const int fl = __ffs(mh) - 1, ll = 31 - __clz(mh);
        R.ref = __builtin_amdgcn_readlane(id, 32 * h + fl);       // first hit in scan order = shortest length
        R.found = __popc(mh);
        R.shortFl = fl; R.shortLl = ll;
    }
}

// Per-wave accumulator for scaffoldReadCounts / scaffoldBaseCounts (BBDukProcessorS.java:2111-2119, 1577-1583).
// Hits cluster on very few scaffold ids (a library has one adapter per mate), so bumping the global counters
// once per read serialises the whole grid on two or three addresses (measured: 55 of 68 ms).  A wave keeps a
// 4-entry cache of (id, reads, bases) in wave-uniform registers and only an evicted entry costs two atomics.
struct ScafAcc { int i0, i1, i2, i3; int r0, r1, r2, r3; long long b0, b1, b2, b3; };
__device__ __forceinline__ void scaf_init(ScafAcc& S) {
    S.i0 = S.i1 = S.i2 = S.i3 = -1; S.r0 = S.r1 = S.r2 = S.r3 = 0; S.b0 = S.b1 = S.b2 = S.b3 = 0;
}
__device__ __forceinline__ void scaf_emit(const KParams& P, int id, int reads, long long bases, const int lane, int64_t* __restrict__ counters) {
    if (id > 0 && lane == 0) {
        atomicAdd((unsigned long long*)&counters[BBDUK_NCOUNTERS + id], (unsigned long long)reads);
        atomicAdd((unsigned long long*)&counters[BBDUK_NCOUNTERS + P.numScaffolds + id], (unsigned long long)bases);
    }
}
__device__ __forceinline__ void scaf_flush(const KParams& P, ScafAcc& S, const int lane, int64_t* __restrict__ counters) {
    scaf_emit(P, S.i0, S.r0, S.b0, lane, counters); scaf_emit(P, S.i1, S.r1, S.b1, lane, counters);
    scaf_emit(P, S.i2, S.r2, S.b2, lane, counters); scaf_emit(P, S.i3, S.r3, S.b3, lane, counters);
    scaf_init(S);
}
__device__ __forceinline__ void scaf_add_n(const KParams& P, ScafAcc& S, int id, int nreads, int L, const int lane, int64_t* __restrict__ counters) {
    // move-to-front (true LRU): the one or two hot ids stay in the cache, a rare id only ever evicts another rare id
    if (id == S.i0) { S.r0 += nreads; S.b0 += L; return; }
    int ri, rr; long long rb;                                     // the entry that goes to the front
    if (id == S.i1) { ri = S.i1; rr = S.r1 + nreads; rb = S.b1 + L; }
    else if (id == S.i2) { ri = S.i2; rr = S.r2 + nreads; rb = S.b2 + L; S.i2 = S.i1; S.r2 = S.r1; S.b2 = S.b1; }
    else if (id == S.i3) { ri = S.i3; rr = S.r3 + nreads; rb = S.b3 + L; S.i3 = S.i2; S.r3 = S.r2; S.b3 = S.b2; S.i2 = S.i1; S.r2 = S.r1; S.b2 = S.b1; }
    else {                                                       // miss: evict the least recently used entry
        scaf_emit(P, S.i3, S.r3, S.b3, lane, counters);
        ri = id; rr = nreads; rb = L;
        S.i3 = S.i2; S.r3 = S.r2; S.b3 = S.b2; S.i2 = S.i1; S.r2 = S.r1; S.b2 = S.b1;
    }
    S.i1 = S.i0; S.r1 = S.r0; S.b1 = S.b0;
    S.i0 = ri; S.r0 = rr; S.b0 = rb;
}
__device__ __forceinline__ void scaf_add(const KParams& P, ScafAcc& S, int id, int L, const int lane, int64_t* __restrict__ counters) {
    scaf_add_n(P, S, id, 1, L, lane, counters);
}

// Scalar (per-read) finish used by the tile kernel: outputs + scaffold counters.
struct ReadOut { int L, a, id, newLen, thr; };   // thr: the read's kfilter threshold (maxBadKmers | minCoveredBases)
template <int MODE>
__device__ __forceinline__ void read_finish(const KParams& P, const ReadScan& R, ReadOut& O, const int lane, ScafAcc& S, int64_t* __restrict__ counters) {
    O.L = R.L; O.a = 0; O.id = -1; O.newLen = R.L; O.thr = R.maxBad;
    if (!R.scan) return;
    int ref; bool hit;
    finish_read<MODE>(P, R.L, R.start, R.stop, R.found, R.iFirst, R.iLast, R.shortFl, R.shortLl, R.ref, O.a, O.newLen, ref, hit);
    if (hit) { O.id = ref_to_id(P, ref); scaf_add(P, S, O.id, R.L, lane, counters); }
}

// One logical record (a pair, or a single read): discard / remove decision and the additive counters.
// bbduk/BBDukProcessorS.java:807-818, 948-1093, 1431-1443, 1464-1493.  acc[0]=readsKTrimmed acc[1]=basesKTrimmed
// acc[2]=readsOutm acc[3]=basesOutm; the other counters follow from these and readsIn/basesIn (see kernel end).
template <int MODE>
__device__ __forceinline__ void record_stage(const KParams& P, ReadOut& X, ReadOut* Y, int* acc, uint8_t& f1, uint8_t& f2) {
    const bool two = (Y != nullptr);
    const int l1 = X.L, l2 = two ? Y->L : 0;
    int n1 = X.newLen, n2 = two ? Y->newLen : 0;
    const int pairCount = two ? 2 : 1;
    const float g1 = (float)l1 * P.minLenFraction, g2 = (float)l2 * P.minLenFraction;
    const int minlen1 = (int)(g1 > (float)P.minReadLength ? g1 : (float)P.minReadLength);
    const int minlen2 = (int)(g2 > (float)P.minReadLength ? g2 : (float)P.minReadLength);
    bool d1 = false, d2 = false, remove = false;
    if (P.storedKmers > 0) {
        if (MODE != BBDUK_MODE_KFILTER) {
            const int x2 = two ? Y->a : 0;
            int xsum = X.a + x2, rkt = (X.a > 0) + (x2 > 0);
            d1 = n1 < minlen1;
            d2 = two && (n2 < minlen2);
            if ((P.rieb && (d1 || d2)) || (d1 && (!two || d2))) { xsum += n1 + n2; rkt = pairCount; remove = true; }
            else if (MODE == BBDUK_MODE_KTRIM_R && P.tpe && xsum > 0 && two && n1 != n2) {   // trimpairsevenly (:1021-1031)
                int x;
                if (n1 > n2) { x = trim_by_amount(n1, 0, n1 - n2, 1, n1); X.a += x; X.newLen = n1; }
                else { x = trim_by_amount(n2, 0, n2 - n1, 1, n2); Y->a += x; Y->newLen = n2; }
                if (rkt < 2) rkt++;
                xsum += x;
            }
            acc[0] += rkt; acc[1] += xsum;
        } else {
            d1 = (P.mcf > 0.f) ? (X.a >= X.thr) : (X.a > X.thr);            // :1042,1047 | :1069-1070
            d2 = two && ((P.mcf > 0.f) ? (Y->a >= Y->thr) : (Y->a > Y->thr));
            if ((P.rieb && (d1 || d2)) || (d1 && (!two || d2))) remove = true;
        }
    }
    if (remove) { acc[2] += pairCount; acc[3] += n1 + n2; }
    f1 = (uint8_t)((d1 ? BBDUK_FLAG_DISCARDED : 0) | (remove ? BBDUK_FLAG_REMOVED : 0));
    f2 = (uint8_t)((d2 ? BBDUK_FLAG_DISCARDED : 0) | (remove ? BBDUK_FLAG_REMOVED : 0));
}

// counters (BBDukProcessorS.add, :300-342).  With rkt/xsum/outm and in-totals known, the rest follows:
// readsOutu = readsIn - readsOutm; ktrim: basesOutu = basesIn - basesKTrimmed (every base either survives in a
// kept pair or is counted as k-trimmed); kfilter: basesOutu = basesIn - basesOutm, filtered == removed.
// trimfailuresto1bp (tf): nothing is evicted; the accumulators then hold rkt / xs = readsKTrimmed / basesKTrimmed or, for kfilter,
// readsKFiltered / basesKFiltered, and bm = the bases that remain (rm = 0).
template <int MODE>
__device__ __forceinline__ void publish_counters(const unsigned long long* s_acc, int64_t* __restrict__ counters, const bool tf = false) {
    const unsigned long long rkt = s_acc[0], xs = s_acc[1], rm = s_acc[2], bm = s_acc[3], rin = s_acc[4], bin = s_acc[5];
    auto add = [&](int slot, unsigned long long v) { if (v) atomicAdd((unsigned long long*)&counters[slot], v); };
    add(BBDUK_READS_IN, rin); add(BBDUK_BASES_IN, bin);
    if (tf) {
        add(BBDUK_READS_OUTU, rin); add(BBDUK_BASES_OUTU, bm);
        if (MODE == BBDUK_MODE_KFILTER) { add(BBDUK_READS_KFILTERED, rkt); add(BBDUK_BASES_KFILTERED, xs); }
        else { add(BBDUK_READS_KTRIMMED, rkt); add(BBDUK_BASES_KTRIMMED, xs); }
        return;
    }
    add(BBDUK_READS_OUTM, rm); add(BBDUK_BASES_OUTM, bm);
    add(BBDUK_READS_OUTU, rin - rm);
    if (MODE == BBDUK_MODE_KSPLIT) {                              // :999-1013, 1431-1443: the split pieces leave through outm
        add(BBDUK_READS_KTRIMMED, rkt); add(BBDUK_BASES_KTRIMMED, xs);
        add(BBDUK_BASES_OUTU, bin - xs - bm);
    } else if (MODE == BBDUK_MODE_KMASK) {                        // masking keeps every read's length
        add(BBDUK_READS_KTRIMMED, rkt); add(BBDUK_BASES_KTRIMMED, xs);
        add(BBDUK_BASES_OUTU, bin - bm);
    } else if (MODE != BBDUK_MODE_KFILTER) {
        add(BBDUK_READS_KTRIMMED, rkt); add(BBDUK_BASES_KTRIMMED, xs);
        add(BBDUK_BASES_OUTU, bin - xs);
    } else {
        add(BBDUK_READS_KFILTERED, rm); add(BBDUK_BASES_KFILTERED, bm);
        add(BBDUK_BASES_OUTU, bin - bm);
    }
}

// --------------------------------------------------------------------------------------------------
// The batch kernel: persistent workgroups (one per CU when the LDS filter is large) walk tiles of reads.
// Template flags strip what a configuration cannot need: SHORT (mink), FORBIDN (undefined-base resets),
// GENERAL (qhdist, restrictleft/right, skipr1/2, rcomp=f).
template <int MODE, bool SHORT, bool FORBIDN, bool GENERAL, bool BIG = false>
__global__ __launch_bounds__(BLOCK_THREADS)
void bbduk_batch_kernel(const KParams P, const uint8_t* __restrict__ bases, const int64_t* __restrict__ offsets,
                        const int64_t n, const int64_t totalBases, const int paired,
                        int32_t* __restrict__ outA, int32_t* __restrict__ outId, uint8_t* __restrict__ outFlags,
                        int64_t* __restrict__ counters, const int* __restrict__ slowFlag) {
    if (PTF) return;                                             // trimfailuresto1bp: bbduk_wave_kernel reports BBDUK_ERR_UNSUPPORTED for such batches
    if (*slowFlag != 1) return;                                   // 0: every unit fits a wave's planes, the wave kernel ran; 2/3: a unit
                                                                  // exceeds this kernel's planes too, bbduk_long_kernel takes the batch
    __shared__ uint32_t s_fwd[PLANE_PAD + CAP_CHUNKS + PLANE_PAD];   // padded both ends: the plain path reads past a read's end unclamped
    __shared__ uint32_t s_cmp[PLANE_PAD + CAP_CHUNKS + PLANE_PAD];
    __shared__ uint32_t s_nm[CAP_CHUNKS / 2 + 4];
    __shared__ int64_t  s_off[TILE_READS + 1];
    __shared__ int32_t  s_a[TILE_READS];
    __shared__ int32_t  s_id[TILE_READS];
    __shared__ uint8_t  s_fl[TILE_READS];
    __shared__ unsigned long long s_acc[6];                       // rkt, basesKTrimmed, readsOutm, basesOutm, readsIn, basesIn
    extern __shared__ uint32_t s_filt[];                          // 2^ldsBits bits, copied once per workgroup

    const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
    const int64_t ntiles = (n + TILE_READS - 1) / TILE_READS;
    ScafAcc scaf; scaf_init(scaf);

    if (tid < 6) s_acc[tid] = 0;
    if (P.ldsBits) {
        const int words = 1 << (P.ldsBits - 5);
        for (int w = tid; w < words; w += BLOCK_THREADS) s_filt[w] = P.ldsImage[w];
    }

    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t r0 = tile * TILE_READS;
        const int cnt = (int)min((int64_t)TILE_READS, n - r0);
        __syncthreads();                                            // previous tile's LDS fully consumed (and filter landed)
        if (tid <= cnt) s_off[tid] = offsets[r0 + tid];
        __syncthreads();
        if (tid == 0) { s_acc[4] += (unsigned long long)cnt; s_acc[5] += (unsigned long long)(s_off[cnt] - s_off[0]); }
        int acc[4] = {0, 0, 0, 0};

        int s = 0;
        while (s < cnt) {
            // how many consecutive reads fit the planes?
            const int64_t off_s = s_off[s];
            const int cand = s + 1 + tid;
            const int okc = (cand <= cnt) && (s_off[min(cand, cnt)] - off_s <= (int64_t)(CAP_BASES - 32));
            int fit = uni(__syncthreads_count(okc));
            if (paired) fit &= ~1;
            if (fit == 0) {                                         // read (or pair) too long for the LDS tile
                if (tid == 0) atomicMax((unsigned long long*)&counters[BBDUK_CTR_STATUS], (unsigned long long)(-BBDUK_ERR_READ_TOO_LONG));
                const int skip = min(paired ? 2 : 1, cnt - s);
                if (tid < skip) { s_a[s + tid] = 0; s_id[s + tid] = -1; s_fl[s + tid] = 0; }
                s += skip;
                continue;
            }
            const int e = s + fit;
            const int64_t B0 = off_s, B1 = s_off[e];
            const int64_t A0 = B0 & ~15LL;
            const int nchunks = (int)((B1 - A0 + 15) >> 4);
            // ---- stage: 16 bases per thread-iteration -> three bit-planes
            for (int c = tid; c < nchunks; c += BLOCK_THREADS) {
                uint32_t r, comp, valid;
                stage_chunk(P, bases, A0 + 16LL * c, totalBases, r, comp, valid);
                s_fwd[PLANE_PAD + nchunks - 1 - c] = r;
                s_cmp[PLANE_PAD + c] = comp;
                reinterpret_cast<uint16_t*>(s_nm)[c] = (uint16_t)(~valid & 0xFFFFu);
            }
            if (tid == 0 && (nchunks & 1)) reinterpret_cast<uint16_t*>(s_nm)[nchunks] = 0;
            __syncthreads();

            // ---- scan: one wave per unit of two consecutive reads (a pair when paired)
            Planes Q; Q.fwd = s_fwd + PLANE_PAD; Q.cmp = s_cmp + PLANE_PAD; Q.nm = s_nm; Q.filt = s_filt; Q.T = nchunks * 16;
            Q.fwdBits = lds_bits_of(Q.fwd); Q.cmpBits = lds_bits_of(Q.cmp);
            const int nunits = (e - s + 1) >> 1;
            const int a0lo = (int)(A0 - s_off[0]);                  // tile-relative origin of the planes (fits int)
            for (int u = wave; u < nunits; u += NWAVES) {
                const int ra = s + 2 * u;
                const bool hasB = (ra + 1) < e;
                const int o0 = uni((int)(s_off[ra] - s_off[0]));
                const int o1 = uni((int)(s_off[ra + 1] - s_off[0]));
                const int o2 = hasB ? uni((int)(s_off[ra + 2] - s_off[0])) : o1;
                ReadScan A, Bz;
                read_init<MODE, SHORT, GENERAL>(P, A, o0 - a0lo, o1 - o0, 0, true);
                read_init<MODE, SHORT, GENERAL>(P, Bz, o1 - a0lo, o2 - o1, paired ? 1 : 0, hasB);
                if constexpr (GENERAL && MODE == BBDUK_MODE_KFILTER) {
                    A.maxBad = kfilter_threshold(P, Q, A.base0, A.L, lane);
                    Bz.maxBad = kfilter_threshold(P, Q, Bz.base0, Bz.L, lane);
                }
                main_scan_pair<MODE, FORBIDN, GENERAL, BIG>(P, Q, A, Bz, lane);
                if constexpr (MODE != BBDUK_MODE_KFILTER && SHORT) {
                    if (P.useShort) short_scan_pair<MODE, GENERAL>(P, Q, A, Bz, lane);
                }
                ReadOut OA, OB;
                read_finish<MODE>(P, A, OA, lane, scaf, counters);
                read_finish<MODE>(P, Bz, OB, lane, scaf, counters);
                uint8_t f1 = 0, f2 = 0, f3 = 0, f4 = 0;
                if (paired) record_stage<MODE>(P, OA, &OB, acc, f1, f2);
                else {
                    record_stage<MODE>(P, OA, nullptr, acc, f1, f3);
                    if (hasB) record_stage<MODE>(P, OB, nullptr, acc, f2, f4);
                }
                if (lane == 0) {
                    s_a[ra] = OA.a; s_id[ra] = OA.id; s_fl[ra] = f1;
                    if (hasB) { s_a[ra + 1] = OB.a; s_id[ra + 1] = OB.id; s_fl[ra + 1] = f2; }
                }
            }
            __syncthreads();
            s = e;
        }
        // ---- coalesced write-back of the tile's results; per-wave partial sums -> LDS
        if (tid < cnt) {
            outA[r0 + tid] = s_a[tid];
            outId[r0 + tid] = s_id[tid];
            outFlags[r0 + tid] = s_fl[tid];
        }
        if (lane == 0) {
#pragma unroll
            for (int q = 0; q < 4; q++) if (acc[q]) atomicAdd(&s_acc[q], (unsigned long long)acc[q]);
        }
    }
    scaf_flush(P, scaf, lane, counters);
    __syncthreads();
    if (tid == 0) publish_counters<MODE>(s_acc, counters);
}

// --------------------------------------------------------------------------------------------------
// Device-side table build: reference sequences -> key -> id map, in HBM, without the host.
// What BBDukLoader.addToMap (bbduk/BBDukLoader.java:416-494) and BBDukIndexMod.addToMap / mutate (:289-445) compute: every
// k-mer of every scaffold whose k bases are all defined, all sequences within `hdist` substitutions of it (ref-side
// Hamming expansion), for mink the prefixes of a scaffold's first k-mer and the suffixes of its last one (lengths
// k-1..mink, `hdist2`), canonicalised, middle-masked, length-tagged; a key keeps the id of the FIRST scaffold that
// produced it (HashArray.setIfNotPresent) -- ids ascend in file order, so that is the minimum id.
//   pass 1  bbduk_build_enum_kernel: one thread per (reference position, first substitution); keys go into an
//           open-addressed scratch set with atomicCAS, ids with atomicMin; distinct keys are counted.
//   pass 2  bbduk_build_place_kernel: one thread per scratch slot; the keys are placed into the final 4-way
//           fingerprint buckets (sized for the distinct count), the LDS presence filter is set with atomicOr.
struct BuildParams { int32_t k, mink, useShort, hdist, hdist2, rcomp; uint64_t middleMask; int64_t totalBases; int32_t nrefs; };

__device__ __forceinline__ int ref_code(uint8_t b) {              // dna/AminoAcid.java:1284-1298 baseToNumber (-1 undefined)
    const uint8_t l = b | 0x20;
    return l == 'a' ? 0 : l == 'c' ? 1 : l == 'g' ? 2 : (l == 't' || l == 'u') ? 3 : -1;
}
__device__ __forceinline__ uint64_t hash64(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }

// Where the build kernels put a (key, id): the open-addressed scratch set of the two-pass build (cache-resident layout), or the
// final big-layout table itself, in place (an HBM-resident map leaves no room for a second copy).  Both keep the SMALLEST id of
// a key: ids ascend in file order, so that is the first scaffold that holds the k-mer (HashArray.setIfNotPresent).
struct Sink {
    int32_t big;
    uint64_t* skeys; int32_t* sids; uint64_t cmask;              // scratch set
    uint64_t* tags; uint64_t* keys; void* ids; int32_t idBytes; BigGeom G;     // big layout: the lines
    uint64_t* tags2; uint4* bkv2; int32_t bucketBits2; uint32_t bucketMask2;    // big layout: the secondary map of the spilled keys
    unsigned long long* distinct;                                // [0] distinct keys, [1] != 0: a key found no slot (map full), [2] spilled keys
};
__device__ __forceinline__ void scratch_insert(const Sink& S, const uint64_t key, const int id) {
    uint64_t hslot = hash64(key) & S.cmask;
    for (;;) {
        const unsigned long long prev = atomicCAS((unsigned long long*)&S.skeys[hslot], (unsigned long long)EMPTY_KEY, (unsigned long long)key);
        if (prev == EMPTY_KEY || prev == key) {
            atomicMin(&S.sids[hslot], id);
            if (prev == EMPTY_KEY) atomicAdd(S.distinct, 1ULL);
            return;
        }
        hslot = (hslot + 1) & S.cmask;
    }
}
__device__ __forceinline__ void big_id_min(const Sink& S, const uint64_t slot, const int id) {
    if (S.idBytes == 4) {
        uint32_t* p = reinterpret_cast<uint32_t*>(S.ids) + slot;
        if (*reinterpret_cast<volatile uint32_t*>(p) > (uint32_t)id) atomicMin(p, (uint32_t)id);
        return;
    }
    uint32_t* wp = reinterpret_cast<uint32_t*>(S.ids) + (slot >> 1);            // two 16-bit ids per word
    const int sh = (int)(slot & 1ULL) * 16;
    for (;;) {
        const uint32_t old = *reinterpret_cast<volatile uint32_t*>(wp);
        if (((old >> sh) & 0xFFFFu) <= (uint32_t)id) return;
        const uint32_t nw = (old & ~(0xFFFFu << sh)) | ((uint32_t)id << sh);
        if (atomicCAS(wp, old, nw) == old) return;
    }
}
// Insert into the secondary map (the cache-resident layout's buckets, filled in place here): first free way at or after the home
// bucket, continuation flags on the full buckets passed; an existing copy of the key keeps the smaller id.
#define SPILL_MAX_BUCKETS 4096
__device__ __forceinline__ void spill_insert(const Sink& S, const uint64_t key, const uint32_t ma, const uint32_t mb, const int id) {
    uint32_t b = bucket_of(mb, S.bucketBits2);
    for (int i = 0; i < SPILL_MAX_BUCKETS; i++) {
        for (int w = 0; w < 4; w++) {
            unsigned long long* slot = reinterpret_cast<unsigned long long*>(&S.bkv2[4ULL * b + w]);       // {key lo, key hi}
            unsigned long long prev = *reinterpret_cast<volatile unsigned long long*>(slot);
            if (prev == EMPTY_KEY) prev = atomicCAS(slot, (unsigned long long)EMPTY_KEY, (unsigned long long)key);
            if (prev == EMPTY_KEY || prev == key) {
                atomicMin(&S.bkv2[4ULL * b + w].z, (uint32_t)id);     // .z starts at 0xFFFFFFFF
                if (prev == EMPTY_KEY) {
                    atomicOr((unsigned long long*)&S.tags2[b], (unsigned long long)tag_of(ma) << (16 * w));
                    atomicAdd(S.distinct, 1ULL); atomicAdd(S.distinct + 2, 1ULL);
                }
                return;
            }
        }
        atomicOr((unsigned long long*)&S.tags2[b], (unsigned long long)TAG_CONT);
        b = (b + 1) & S.bucketMask2;
    }
    atomicOr(S.distinct + 1, 1ULL);                                // the secondary map is full
}
__device__ __forceinline__ void big_insert(const Sink& S, const uint64_t key, const int id) {
    const uint64_t v = strip_len(key);
    const uint32_t ma = mix_a(v), mb = mix_b(v);
    uint32_t w1, w2;
    big_words(big_line_of_key(S.G, key, ma), mb, w1, w2);
    const unsigned long long fp = (unsigned long long)tag_of(ma);
#pragma unroll
    for (int q = 0; q < 2; q++) {                                 // the same order for every inserter of this key: no duplicates
        const uint32_t word = q ? w2 : w1;
        for (int way = 0; way < 4; way++) {
            const uint64_t slot = 4ULL * word + way;
            unsigned long long prev = *reinterpret_cast<volatile unsigned long long*>(&S.keys[slot]);
            if (prev == EMPTY_KEY) prev = atomicCAS((unsigned long long*)&S.keys[slot], (unsigned long long)EMPTY_KEY, (unsigned long long)key);
            if (prev == EMPTY_KEY) {                              // claimed: fingerprint, id, count
                atomicOr((unsigned long long*)&S.tags[word], fp << (16 * way));
                big_id_min(S, slot, id);
                atomicAdd(S.distinct, 1ULL);
                return;
            }
            if (prev == key) { big_id_min(S, slot, id); return; }
        }
    }
    atomicOr((unsigned long long*)&S.tags[w1], 1ULL << spill_bit(ma));              // both words full: the key is spilled
    spill_insert(S, key, ma, mb, id);
}
__device__ __forceinline__ void sink_insert(const Sink& S, const uint64_t key, const int id) {
    if (S.big) big_insert(S, key, id); else scratch_insert(S, key, id);
}
// one sequence v of `len` bases -> its key (toValue, BBDukIndexMod.java:532-544)
__device__ __forceinline__ uint64_t build_key(const BuildParams& B, const uint64_t v, const int len) {
    const uint64_t r = dev_rcomp(v, len);
    const uint64_t mx = B.rcomp ? (v > r ? v : r) : v;
    return (mx & B.middleMask) | (1ULL << (2 * len));
}
// v and everything within `dist` (0..3) substitutions of it, starting from first-level choice `v1` (0 = v itself,
// 1+3*i+j = base i replaced by its j-th alternative); the caller spreads v1 over threads
__device__ __forceinline__ void emit_variants(const BuildParams& B, const uint64_t v, const int len, const int dist, const int v1, const int id, const Sink& S) {
    uint64_t t1 = v;
    if (v1 > 0) {
        if (dist < 1) return;
        const int i = (v1 - 1) / 3, j = (v1 - 1) % 3;
        if (i >= len) return;
        const uint64_t cur = (v >> (2 * i)) & 3ULL;
        t1 = (v & ~(3ULL << (2 * i))) | (((cur + 1 + j) & 3ULL) << (2 * i));
    }
    sink_insert(S, build_key(B, t1, len), id);
    if (dist >= 2 && v1 > 0) {                                    // second substitution at a lower position (each pair once)
        const int i1 = (v1 - 1) / 3;
        for (int i = 0; i < i1; i++) {
            const uint64_t cur = (t1 >> (2 * i)) & 3ULL;
            for (int j = 0; j < 3; j++) {
                const uint64_t t2 = (t1 & ~(3ULL << (2 * i))) | (((cur + 1 + j) & 3ULL) << (2 * i));
                sink_insert(S, build_key(B, t2, len), id);
                if (dist >= 3) {                                  // third substitution, lower still
                    for (int i3 = 0; i3 < i; i3++) {
                        const uint64_t cur3 = (t2 >> (2 * i3)) & 3ULL;
                        for (int j3 = 0; j3 < 3; j3++)
                            sink_insert(S, build_key(B, (t2 & ~(3ULL << (2 * i3))) | (((cur3 + 1 + j3) & 3ULL) << (2 * i3)), len), id);
                    }
                }
            }
        }
    }
}

// refs = the pieces' bases concatenated, roff[nrefs+1] their offsets, rid[nrefs] the scaffold id of each piece, rfl[nrefs] bit 0 /
// bit 1 = the piece holds its scaffold's first / last base (a scaffold longer than one upload chunk arrives as overlapping pieces;
// the short k-mers of mink belong to the scaffold's first and last k-mer only).  V1 = first-level choices per position: 1 + 3k with
// a Hamming distance, 1 without (so a plain 10 Gbase reference is not paid for 94 times).
__global__ void bbduk_build_enum_kernel(const BuildParams B, const uint8_t* __restrict__ refs, const int64_t* __restrict__ roff,
                                        const int32_t* __restrict__ rid, const uint8_t* __restrict__ rfl, const int V1, const Sink S) {
    const int k = B.k;
    const int64_t work = B.totalBases * (int64_t)V1;
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < work; w += (int64_t)gridDim.x * blockDim.x) {
        const int64_t g = w / V1; const int v1 = (int)(w - g * V1);
        int lo = 0, hi = B.nrefs;                                 // piece of base g: last s with roff[s] <= g
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (roff[mid] <= g) lo = mid; else hi = mid; }
        const int64_t s0 = roff[lo], n = roff[lo + 1] - s0, i = g - s0;
        if (n < k || i < k - 1) continue;
        uint64_t fwd = 0; bool ok = true;
        for (int q = k - 1; q >= 0; q--) {                        // window [i-k+1, i]: every base has to be defined (BBDukLoader.java:441-452)
            const int c = ref_code(refs[g - q]);
            ok = ok && c >= 0;
            fwd = (fwd << 2) | (uint64_t)(c < 0 ? 0 : c);
        }
        if (!ok) continue;
        const int id = rid[lo];                                   // scaffoldNames[0] is reserved (bbduk/BBDukIndex.java:105-107): ids start at 1
        emit_variants(B, fwd, k, B.hdist, v1, id, S);
        const bool first = (i == k - 1) && (rfl[lo] & 1), last = (i == n - 1) && (rfl[lo] & 2);
        if (B.useShort && (first || last)) {
            for (int L = k - 1; L >= B.mink; L--) {
                if (first) emit_variants(B, fwd >> (2 * (k - L)), L, B.hdist2, v1, id, S);        // addToMapRightShift
                if (last) emit_variants(B, fwd & ((1ULL << (2 * L)) - 1ULL), L, B.hdist2, v1, id, S);   // addToMapLeftShift
            }
        }
    }
}

// (key, id) pairs a host built (bbduk_upload_pairs / bbduk_upload_table_way) -> the same sinks
__global__ void bbduk_insert_pairs_kernel(const int64_t* __restrict__ keys, const int32_t* __restrict__ vals, const int64_t n, const Sink S) {
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x)
        sink_insert(S, (uint64_t)keys[q], vals[q]);
}

__global__ void bbduk_build_place_kernel(const uint64_t* __restrict__ skeys, const int32_t* __restrict__ sids, const uint64_t cslots,
                                         uint64_t* __restrict__ tags, uint4* __restrict__ bkv, const int bucketBits, const uint32_t bucketMask,
                                         uint32_t* __restrict__ ldsImage, const int ldsBits) {
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < cslots; q += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t key = skeys[q];
        if (key == EMPTY_KEY) continue;
        const uint64_t v = strip_len(key);
        const uint32_t ma = mix_a(v), mb = mix_b(v);
        uint32_t b = bucket_of(mb, bucketBits);
        for (bool placed = false; !placed;) {
            for (int w = 0; w < 4 && !placed; w++) {
                unsigned long long* slot = reinterpret_cast<unsigned long long*>(&bkv[4ULL * b + w]);    // {key lo, key hi} are its first 8 bytes
                if (atomicCAS(slot, (unsigned long long)EMPTY_KEY, (unsigned long long)key) == EMPTY_KEY) {
                    bkv[4ULL * b + w].z = (uint32_t)sids[q]; bkv[4ULL * b + w].w = 0u;
                    atomicOr((unsigned long long*)&tags[b], (unsigned long long)tag_of(ma) << (16 * w));
                    placed = true;
                }
            }
            if (!placed) { atomicOr((unsigned long long*)&tags[b], (unsigned long long)TAG_CONT); b = (b + 1) & bucketMask; }
        }
        if (ldsBits) atomicOr(&ldsImage[filt_byte(ma, ldsBits) >> 2], 1u << (ma & 31u));
    }
}

// --------------------------------------------------------------------------------------------------
// ktrim=n (kmask): bbduk/BBDukProcessorS.java:2149-2323 with kmaskFullyCovered=false.  A secondary operator, written
// for clarity rather than speed on the run-time-general code paths (GENERAL scans, every flag honoured): tiles staged
// like bbduk_batch_kernel, one wave per READ (mates only meet in the record stage), a fourth LDS bit-plane that
// collects the k-mer END positions that hit.  A base b is masked iff some hit ends in [b-trimPad, b+k-1-trimPad]
// (bs.set(max(0,i-minus), i+plus), :2190), or a short k-mer on either side covers it (:2236, :2279; the short scans run
// always here, not only when the main scan found nothing).  out = number of masked bases (BitSet.cardinality(), which
// also counts the bits a positive trimPad pushes past the read end), id0, flags, and the per-base mask.
#define KM_CAP_BASES  32768                    // >= BBDUK_MAX_READ_LEN + 32: any single read fits
#define KM_CAP_CHUNKS (KM_CAP_BASES / 16)
__global__ __launch_bounds__(BLOCK_THREADS)
void bbduk_kmask_kernel(const KParams P, const uint8_t* __restrict__ bases, const int64_t* __restrict__ offsets,
                        const int64_t n, const int64_t totalBases, const int paired,
                        int32_t* __restrict__ outA, int32_t* __restrict__ outId, uint8_t* __restrict__ outFlags,
                        uint32_t* __restrict__ outMask, int64_t* __restrict__ counters, int* __restrict__ longFlag) {
    __shared__ uint32_t s_fwd[PLANE_PAD + KM_CAP_CHUNKS + PLANE_PAD];
    __shared__ uint32_t s_cmp[PLANE_PAD + KM_CAP_CHUNKS + PLANE_PAD];
    __shared__ uint32_t s_nm[KM_CAP_CHUNKS / 2 + 4];
    __shared__ uint32_t s_hit[KM_CAP_CHUNKS / 2 + 4];             // bit p <=> a k-mer ending at plane position p matched
    __shared__ int64_t  s_off[TILE_READS + 1];
    __shared__ int32_t  s_a[TILE_READS];
    __shared__ int32_t  s_id[TILE_READS];
    __shared__ unsigned long long s_acc[6];                       // rkt, basesKTrimmed, readsOutm, basesOutm, readsIn, basesIn
    extern __shared__ uint32_t s_filt[];

    if (PTF) return;                                             // trimfailuresto1bp: bbduk_wave_kernel reports BBDUK_ERR_UNSUPPORTED for such batches
    if (P.waveFirst && *longFlag == 0) return;                    // every unit fits a wave's planes: bbduk_wave_kernel<KMASK> did the batch
    const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
    const int64_t ntiles = (n + TILE_READS - 1) / TILE_READS;
    const int k = P.k, tp = P.trimPad;
    ScafAcc scaf; scaf_init(scaf);
    if (tid < 6) s_acc[tid] = 0;
    if (P.ldsBits) {
        const int words = 1 << (P.ldsBits - 5);
        for (int w = tid; w < words; w += BLOCK_THREADS) s_filt[w] = P.ldsImage[w];
    }
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t r0 = tile * TILE_READS;
        const int cnt = (int)min((int64_t)TILE_READS, n - r0);
        __syncthreads();
        if (tid <= cnt) s_off[tid] = offsets[r0 + tid];
        __syncthreads();
        if (tid == 0) { s_acc[4] += (unsigned long long)cnt; s_acc[5] += (unsigned long long)(s_off[cnt] - s_off[0]); }
        int s = 0;
        while (s < cnt) {
            const int64_t off_s = s_off[s];
            const int cand = s + 1 + tid;
            const int okc = (cand <= cnt) && (s_off[min(cand, cnt)] - off_s <= (int64_t)(KM_CAP_BASES - 32));
            const int fit = uni(__syncthreads_count(okc));
            if (fit == 0) {                                         // a read beyond these planes: bbduk_kmask_long_kernel masks it (its
                if (tid == 0) { *longFlag = 1; s_a[s] = -1; s_id[s] = -1; }    // flags depend on lengths only and are written below)
                s += 1;
                continue;
            }
            const int e = s + fit;
            const int64_t B0 = off_s, B1 = s_off[e];
            const int64_t A0 = B0 & ~15LL;
            const int nchunks = (int)((B1 - A0 + 15) >> 4);
            for (int c = tid; c < nchunks; c += BLOCK_THREADS) {
                uint32_t r, comp, valid;
                stage_chunk(P, bases, A0 + 16LL * c, totalBases, r, comp, valid);
                s_fwd[PLANE_PAD + nchunks - 1 - c] = r;
                s_cmp[PLANE_PAD + c] = comp;
                reinterpret_cast<uint16_t*>(s_nm)[c] = (uint16_t)(~valid & 0xFFFFu);
            }
            if (tid == 0 && (nchunks & 1)) reinterpret_cast<uint16_t*>(s_nm)[nchunks] = 0;
            for (int w = tid; w < (nchunks + 1) / 2 + 2; w += BLOCK_THREADS) s_hit[w] = 0;
            __syncthreads();

            Planes Q; Q.fwd = s_fwd + PLANE_PAD; Q.cmp = s_cmp + PLANE_PAD; Q.nm = s_nm; Q.filt = s_filt; Q.T = nchunks * 16;
            Q.fwdBits = lds_bits_of(Q.fwd); Q.cmpBits = lds_bits_of(Q.cmp);
            for (int rd = s + wave; rd < e; rd += NWAVES) {        // one wave per read
                const int L = uni((int)(s_off[rd + 1] - s_off[rd]));
                const int base0 = uni((int)(s_off[rd] - A0));
                const int pairnum = paired ? (rd & 1) : 0;
                ReadScan R;
                R.base0 = base0; R.L = L; R.hasN = -1; R.maxBad = 0;
                R.start = span_start<true>(P, L); R.stop = span_stop<true>(P, L);
                R.scan = P.storedKmers > 0 && L >= k && !((P.skipR1 && pairnum == 0) || (P.skipR2 && pairnum == 1));   // :2151-2154
                int found = 0, id0 = -1;
                int leftEnd = 0, rightStart = L;                    // bases [0,leftEnd) and [rightStart,L) are masked by short k-mers
                if (R.scan) {
                    ReadWin W;
                    win_init<true, true>(P, Q, R, W, lane);
                    for (int ib = W.first; W.on && ib < W.stop; ib += 256) {       // 4 positions per lane and iteration
                        uint64_t kmer[4], rk[4]; bool ok[4]; int ref[4];
                        windows2<true, true>(P, Q, W, ib + 2 * lane, true, kmer, rk, ok);
                        if (ib + 128 < W.stop) windows2<true, true>(P, Q, W, ib + 128 + 2 * lane, true, kmer + 2, rk + 2, ok + 2);
                        else { kmer[2] = kmer[3] = 0; rk[2] = rk[3] = 0; ok[2] = ok[3] = false; }
                        lookup4<true>(P, Q.filt, kmer, rk, ok, ref);
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const int i = ib + 128 * (q >> 1) + 2 * lane + (q & 1);
                            // the plane marks hits -- or, with kmaskfullycovered, the positions that do NOT match (looked up or
                            // not, :2193-2195): every one of those clears its window of the initially full bit set
                            if (P.mfc ? (i < W.stop && ref[q] == -1) : (ref[q] != -1)) atomicOr(&s_hit[(base0 + i) >> 5], 1u << ((base0 + i) & 31));
                        }
#pragma unroll
                        for (int hb = 0; hb < 2; hb++) {            // id0 = id of the first hit in position order
                            const uint64_t me = __ballot(ref[2 * hb] != -1), mo = __ballot(ref[2 * hb + 1] != -1);
                            if (id0 < 0 && (me | mo)) {
                                const int le = me ? __ffsll((unsigned long long)me) - 1 : 64, lo = mo ? __ffsll((unsigned long long)mo) - 1 : 64;
                                id0 = (2 * lo + 1 < 2 * le) ? __builtin_amdgcn_readlane(ref[2 * hb + 1], lo) : __builtin_amdgcn_readlane(ref[2 * hb], le);
                            }
                            found += __popcll(me) + __popcll(mo);
                        }
                    }
                    if (P.useShort) {                               // lanes 0-31: left side, lanes 32-63: right side; length mink + (lane&31)
                        const bool right = lane >= 32;
                        const int Ls = P.mink + (lane & 31);
                        bool act; int i;
                        uint64_t km = 0, rr = 0;
                        if (!right) {
                            const int Lmax = min(k, R.stop) - R.start;            // i = start+Ls-1 < min(k, stop)
                            act = Ls <= Lmax; i = R.start + Ls - 1;
                            const int Lc = act ? Ls : 1;
                            if (act) { km = extract2(Q.fwd, Q.T - 1 - (base0 + R.start + Lc - 1), Lc) & P.mask; rr = extract2(Q.cmp, base0 + R.start, Lc); }
                        } else {
                            const int Lmax = (R.stop >= k ? k - 1 : R.stop);      // i = stop-Ls > max(-1, stop-k)
                            act = Ls <= Lmax; i = R.stop - Ls;
                            const int Lc = act ? Ls : 1;
                            if (act) { km = extract2(Q.fwd, Q.T - 1 - (base0 + R.stop - 1), Lc); rr = extract2(Q.cmp, base0 + R.stop - Lc, Lc) & P.mask; }
                        }
                        if (P.qskip > 1) act = act && (i % P.qskip) == 0;
                        const int Lc = act ? Ls : 1;
                        const int sref = lookup<true>(P, Q.filt, km, rr, 1ULL << (2 * Lc), Lc, P.qhdist2, act);
                        const uint64_t hm = __ballot(sref != -1);
                        const uint32_t mL = (uint32_t)hm, mR = (uint32_t)(hm >> 32);
                        if (id0 < 0 && mL) id0 = __builtin_amdgcn_readlane(sref, __ffs(mL) - 1);          // left hits first, shortest first
                        if (id0 < 0 && mR) id0 = __builtin_amdgcn_readlane(sref, 32 + __ffs(mR) - 1);
                        found += __popc(mL) + __popc(mR);
                        if (!P.mfc) {
                            if (mL) { const int iMax = R.start + (P.mink + (31 - __clz(mL))) - 1; leftEnd = max(0, min(L, iMax + tp + 1)); }     // :2236
                            if (mR) { const int iMin = R.stop - (P.mink + (31 - __clz(mR))); rightStart = min(L, max(0, iMin - tp)); }           // :2279
                        } else {
                            // fully covered: a length that does not match clears its end (:2243-2245, 2286-2288); the length mink-1 is
                            // examined (len2>=minminlen) but never looked up, so it always clears.  leftEnd / rightStart become the
                            // borders of the CLEARED prefix / suffix: the longest non-matching length decides.
                            const int LmaxL = min(k, R.stop) - R.start, LmaxR = (R.stop >= k ? k - 1 : R.stop);
                            const uint32_t actL = LmaxL >= P.mink ? (LmaxL - P.mink >= 31 ? ~0u : ((2u << (LmaxL - P.mink)) - 1u)) : 0u;
                            const uint32_t actR = LmaxR >= P.mink ? (LmaxR - P.mink >= 31 ? ~0u : ((2u << (LmaxR - P.mink)) - 1u)) : 0u;
                            const uint32_t missL = actL & ~mL, missR = actR & ~mR;
                            const int lenL = missL ? P.mink + (31 - __clz(missL)) : ((P.mink - 1 >= 1 && LmaxL >= P.mink - 1) ? P.mink - 1 : 0);
                            const int lenR = missR ? P.mink + (31 - __clz(missR)) : ((P.mink - 1 >= 1 && LmaxR >= P.mink - 1) ? P.mink - 1 : 0);
                            if (lenL > 0) leftEnd = max(0, min(L, R.start + lenL - 1 + tp + 1));
                            if (lenR > 0) rightStart = min(L, max(0, R.stop - lenR - tp));
                        }
                    }
                }
                int card = 0;
                if (found > 0) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    scaf_add(P, scaf, id0, L, lane, counters);
                    const int64_t g0 = s_off[rd];                   // bit offset of the read in the output mask
                    const int nb = P.mfc ? L : L + max(tp, 0) + 1;   // BitSet size: bits >= L count but are not written
                    for (int b0 = 0; b0 < nb; b0 += 64) {
                        const int b = b0 + lane;
                        bool cov = false;
                        if (b < nb) {
                            const int lo = max(0, b - tp), hi = min(L - 1, b + k - 1 - tp);
                            if (lo <= hi) cov = extract1(s_hit, base0 + lo, hi - lo + 1) != 0u;      // hi-lo+1 <= k <= 31
                            if (!P.mfc) { if (b < L) cov = cov || b < leftEnd || b >= rightStart; }
                            else cov = !cov && b >= leftEnd && b < rightStart;                        // nothing cleared this base
                        }
                        const uint64_t cm = __ballot(cov);
                        card += __popcll(cm);
                        const uint64_t wm = cm & ((L - b0 >= 64) ? ~0ULL : ((L - b0 <= 0) ? 0ULL : ((1ULL << (L - b0)) - 1ULL)));   // bases only
                        if (wm && lane < 3) {                       // up to three 32-bit words of the global mask
                            const int64_t g = g0 + b0; const int sh = (int)(g & 31);
                            const uint64_t plo = wm << sh, phi = sh ? (wm >> (64 - sh)) : 0ULL;
                            const uint32_t piece = lane == 0 ? (uint32_t)plo : (lane == 1 ? (uint32_t)(plo >> 32) : (uint32_t)phi);
                            if (piece) atomicOr(&outMask[(g >> 5) + lane], piece);
                        }
                    }
                }
                if (lane == 0) { s_a[rd] = card; s_id[rd] = found > 0 ? id0 : -1; }
            }
            __syncthreads();
            s = e;
        }
        // ---- record stage (:984-998, 1009-1016, 1028-1029, 1431-1443): one thread per read, mates look at each other
        if (tid < cnt) {
            const int L1 = (int)(s_off[tid + 1] - s_off[tid]);
            const float g1 = (float)L1 * P.minLenFraction;
            const bool d = P.storedKmers > 0 && L1 < (int)(g1 > (float)P.minReadLength ? g1 : (float)P.minReadLength);
            bool remove = d;
            if (paired) {
                const int m = tid ^ 1;
                const int L2 = (int)(s_off[m + 1] - s_off[m]);
                const float g2 = (float)L2 * P.minLenFraction;
                const bool dm = P.storedKmers > 0 && L2 < (int)(g2 > (float)P.minReadLength ? g2 : (float)P.minReadLength);
                remove = (P.rieb && (d || dm)) || (d && dm);
            }
            const int a = s_a[tid];
            if (a >= 0) { outA[r0 + tid] = a; outId[r0 + tid] = s_id[tid]; }       // a < 0: left to bbduk_kmask_long_kernel
            outFlags[r0 + tid] = (uint8_t)((d ? BBDUK_FLAG_DISCARDED : 0) | (remove ? BBDUK_FLAG_REMOVED : 0));
            if (a > 0) { atomicAdd(&s_acc[0], 1ULL); atomicAdd(&s_acc[1], (unsigned long long)a); }     // rktsum / xsum: unchanged by removal (ktrimN)
            if (remove) { atomicAdd(&s_acc[2], 1ULL); atomicAdd(&s_acc[3], (unsigned long long)L1); }
        }
    }
    scaf_flush(P, scaf, lane, counters);
    __syncthreads();
    if (tid == 0) {
        const unsigned long long rkt = s_acc[0], xs = s_acc[1], rm = s_acc[2], bm = s_acc[3], rin = s_acc[4], bin = s_acc[5];
        auto add = [&](int slot, unsigned long long v) { if (v) atomicAdd((unsigned long long*)&counters[slot], v); };
        add(BBDUK_READS_IN, rin); add(BBDUK_BASES_IN, bin);
        add(BBDUK_READS_KTRIMMED, rkt); add(BBDUK_BASES_KTRIMMED, xs);
        add(BBDUK_READS_OUTM, rm); add(BBDUK_BASES_OUTM, bm);
        add(BBDUK_READS_OUTU, rin - rm); add(BBDUK_BASES_OUTU, bin - bm);      // masking keeps every read's length
    }
}

// --------------------------------------------------------------------------------------------------
// The remaining reductions over the same main scan, one kernel, one wave per read, staged like
// bbduk_kmask_kernel and like it written for clarity on the run-time-general scan functions:
//   RED_BIG   countSetKmersBig (bbduk/BBDukProcessorS.java:1726-1804): k > 31 emulated by runs of consecutive matching
//             31-mers.  The run state machine is order dependent (positions that are not looked up neither extend nor
//             close a run), so blocks that hold a hit or meet an open run are replayed position by position in scalar code,
//             reading each lane's result with v_readlane; all other blocks cost nothing beyond the lookups.
//   RED_BEST  findBestMatch (:1659-1719): every hit counts for its scaffold, the read goes to the scaffold with the most
//             hits, the earliest-seen one among equals.  Lane j keeps the j-th distinct id of the read and its count.
//   RED_SPLIT ksplit (:2332-2506): first and last hit of the main scan, or the short k-mers of the right end, or (if still
//             nothing) of the left end, give (leftmost, rightmost); the caller trims or splits the read.
#define RED_BIG   0
#define RED_BEST  1
#define RED_SPLIT 2
#define KS_MAX_IDS 64
// Scan state of bbduk_kscan_kernel's reductions, carried across the blocks (and, for long reads, the chunks) of one read.
struct KScanState {
    int found, rid;                       // what the reference's method returns / credits
    int firstI, lastI, id0;               // RED_SPLIT: first and last hit of the main scan, id of the first
    int bkStart, bkStop, lastId; bool done;   // RED_BIG run state
    int myId, myCnt, nids;                // RED_BEST: lane j owns the j-th distinct id
};
__device__ __forceinline__ void kscan_init(KScanState& S) {
    S.found = 0; S.rid = -1; S.firstI = -1; S.lastI = -1; S.id0 = -1; S.bkStart = -1; S.bkStop = -1; S.lastId = -1; S.done = false;
    S.myId = 0; S.myCnt = 0; S.nids = 0;
}
// findBestMatch's counting (:1672-1690) over 128 positions (he / ho: the lanes whose even / odd position hit, refE / refO their ids):
// hits in position order, every remaining hit with the same id counted at once; lane j owns the j-th distinct id
__device__ __forceinline__ void best_fold(KScanState& S, const uint64_t he, const uint64_t ho, const int refE, const int refO, const int lane,
                                          int64_t* __restrict__ counters) {
    uint64_t re = he, ro = ho;
    while (re | ro) {
        const int le = re ? __ffsll((unsigned long long)re) - 1 : 64, lo = ro ? __ffsll((unsigned long long)ro) - 1 : 64;
        const bool odd = 2 * lo + 1 < 2 * le;
        const int id = odd ? __builtin_amdgcn_readlane(refO, lo) : __builtin_amdgcn_readlane(refE, le);
        const uint64_t se = re & __ballot(refE == id), so = ro & __ballot(refO == id);
        const int c = __popcll(se) + __popcll(so);
        const uint64_t have = __ballot(lane < S.nids && S.myId == id);
        if (have) { if (lane == __ffsll((unsigned long long)have) - 1) S.myCnt += c; }
        else if (S.nids < KS_MAX_IDS) { if (lane == S.nids) { S.myId = id; S.myCnt = c; } S.nids++; }
        else if (lane == 0) atomicMax((unsigned long long*)&counters[BBDUK_CTR_STATUS], (unsigned long long)(-BBDUK_ERR_ID_OVERFLOW));
        S.found += c;
        re &= ~se; ro &= ~so;
    }
}
// countSetKmersBig's run state machine (:1749-1779) over 128 positions, replayed position by position in scalar code: ke / ko = the lanes whose
// even / odd position was looked up (the others are transparent), refE / refO their results, i0 = the position of lane 0's even slot
__device__ __forceinline__ void big_fold(KScanState& S, const uint64_t ke, const uint64_t ko, const int refE, const int refO, const int i0, const int sub, const int thr) {
    for (int j = 0; j < 64 && !S.done; j++) {
#pragma unroll
        for (int par = 0; par < 2; par++) {
            if (S.done || !(((par ? ko : ke) >> j) & 1ULL)) continue;    // not looked up: transparent
            const int i = i0 + 2 * j + par;
            const int id = __builtin_amdgcn_readlane(par ? refO : refE, j);
            if (id > 0) { S.lastId = id; if (S.bkStart == -1) S.bkStart = i; S.bkStop = i; }
            else if (S.bkStart > -1) {
                const int dif = S.bkStop - S.bkStart - sub;
                S.bkStop = S.bkStart = -1;
                if (dif > 0) {
                    const int old = S.found;
                    S.found += dif;
                    if (S.found > thr && old <= thr) { S.rid = S.lastId; S.done = true; }     // :1763-1773 early exit
                }
            }
        }
    }
}
// the positions [W.first, W.stop) of one read (or of one chunk of it), 256 per step
template <int RED>
__device__ __forceinline__ void kscan_window(const KParams& P, const Planes& Q, const ReadWin& W, KScanState& S, const int thr, const int lane,
                                             int64_t* __restrict__ counters) {
    const int sub = P.kbig - P.k - 1;
    for (int ib = W.first; W.on && !S.done && ib < W.stop; ib += 256) {
        uint64_t kmer[4], rk[4]; bool ok[4]; int ref[4];
        windows2<true, true>(P, Q, W, ib + 2 * lane, true, kmer, rk, ok);
        if (ib + 128 < W.stop) windows2<true, true>(P, Q, W, ib + 128 + 2 * lane, true, kmer + 2, rk + 2, ok + 2);
        else { kmer[2] = kmer[3] = 0; rk[2] = rk[3] = 0; ok[2] = ok[3] = false; }
        lookup4<true>(P, Q.filt, kmer, rk, ok, ref);
#pragma unroll
        for (int hb = 0; hb < 2; hb++) {
            const uint64_t he = __ballot(ref[2 * hb] != -1), ho = __ballot(ref[2 * hb + 1] != -1);
            const int i0 = ib + 128 * hb;            // position of (lane j, parity p) = i0 + 2j + p
            if (RED == RED_SPLIT) {
                if (he | ho) {
                    const int le = he ? __ffsll((unsigned long long)he) - 1 : 64, lo = ho ? __ffsll((unsigned long long)ho) - 1 : 64;
                    if (S.firstI < 0) {
                        const bool odd = 2 * lo + 1 < 2 * le;
                        S.firstI = i0 + (odd ? 2 * lo + 1 : 2 * le);
                        S.id0 = odd ? __builtin_amdgcn_readlane(ref[2 * hb + 1], lo) : __builtin_amdgcn_readlane(ref[2 * hb], le);
                    }
                    const int me = he ? 63 - __clzll((unsigned long long)he) : -1, mo = ho ? 63 - __clzll((unsigned long long)ho) : -1;
                    S.lastI = i0 + max(2 * me, 2 * mo + 1);
                    S.found += __popcll(he) + __popcll(ho);
                }
            } else if (RED == RED_BEST) {
                best_fold(S, he, ho, ref[2 * hb], ref[2 * hb + 1], lane, counters);
            } else {
                if (!(he | ho) && S.bkStart < 0) continue;          // nothing to open, nothing to close
                big_fold(S, __ballot(ok[2 * hb]), __ballot(ok[2 * hb + 1]), ref[2 * hb], ref[2 * hb + 1], i0, sub, thr);
            }
        }
    }
}
// after the last position: a run that reaches the end of the read (RED_BIG), the best scaffold (RED_BEST)
template <int RED>
__device__ __forceinline__ void kscan_finish(KScanState& S, const int thr, const int sub, const int lane) {
    if (RED == RED_BIG && !S.done && S.bkStart > -1) {                 // :1783-1800
        const int dif = S.bkStop - S.bkStart - sub;
        if (dif > 0) { const int old = S.found; S.found += dif; if (S.found > thr && old <= thr) S.rid = S.lastId; }
    }
    if (RED == RED_BEST && S.found > thr) {                            // condenseLoose + first maximum (:1694-1701)
        int mx = (lane < S.nids) ? S.myCnt : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o));
        const uint64_t best = __ballot(lane < S.nids && S.myCnt == mx);
        S.rid = __builtin_amdgcn_readlane(S.myId, __ffsll((unsigned long long)best) - 1);
    }
}

// rename's input (:1702, 2508-2522): the distinct scaffolds a matched read hit, in first-hit order, with their hit counts.
// matchN = idList.size (0 for a read that did not match); at most matchCap entries are written.
__device__ __forceinline__ void kscan_write_matches(const KParams& P, const KScanState& S, const int thr, const int64_t read, const int lane) {
    if (!P.matchN) return;
    const bool m = S.found > thr;
    if (lane == 0) P.matchN[read] = m ? S.nids : 0;
    if (m && lane < S.nids && lane < P.matchCap) {
        P.matchIds[read * P.matchCap + lane] = S.myId;
        P.matchCnt[read * P.matchCap + lane] = S.myCnt;
    }
}

template <int RED>
__global__ __launch_bounds__(BLOCK_THREADS)
void bbduk_kscan_kernel(const KParams P, const uint8_t* __restrict__ bases, const int64_t* __restrict__ offsets,
                        const int64_t n, const int64_t totalBases, const int paired,
                        int32_t* __restrict__ outA, int32_t* __restrict__ outId, uint8_t* __restrict__ outFlags,
                        int32_t* __restrict__ outLeft, int32_t* __restrict__ outRight, int64_t* __restrict__ counters, const int* __restrict__ longFlag) {
    if (PTF) return;                                             // trimfailuresto1bp: bbduk_wave_kernel reports BBDUK_ERR_UNSUPPORTED for such batches
    if (*longFlag & 2) return;                                    // a read beyond these planes: bbduk_kscan_long_kernel takes the batch
    if (P.waveFirst && *longFlag == 0) return;                    // every read fits a wave's planes: bbduk_wave_kernel<KSPLIT> did the batch
    __shared__ uint32_t s_fwd[PLANE_PAD + KM_CAP_CHUNKS + PLANE_PAD];
    __shared__ uint32_t s_cmp[PLANE_PAD + KM_CAP_CHUNKS + PLANE_PAD];
    __shared__ uint32_t s_nm[KM_CAP_CHUNKS / 2 + 4];
    __shared__ int64_t  s_off[TILE_READS + 1];
    __shared__ int32_t  s_a[TILE_READS];
    __shared__ int32_t  s_id[TILE_READS];
    __shared__ int32_t  s_thr[TILE_READS];                        // kfilter: maxBadKmersR of the read; ksplit: its new pair length
    __shared__ uint8_t  s_split[TILE_READS];
    __shared__ unsigned long long s_acc[6];                       // rkt, xsum, readsOutm, basesOutm, readsIn, basesIn
    extern __shared__ uint32_t s_filt[];

    const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
    const int64_t ntiles = (n + TILE_READS - 1) / TILE_READS;
    const int k = P.k, tp = P.trimPad;
    ScafAcc scaf; scaf_init(scaf);
    if (tid < 6) s_acc[tid] = 0;
    if (P.ldsBits) {
        const int words = 1 << (P.ldsBits - 5);
        for (int w = tid; w < words; w += BLOCK_THREADS) s_filt[w] = P.ldsImage[w];
    }
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t r0 = tile * TILE_READS;
        const int cnt = (int)min((int64_t)TILE_READS, n - r0);
        __syncthreads();
        if (tid <= cnt) s_off[tid] = offsets[r0 + tid];
        __syncthreads();
        if (tid == 0) { s_acc[4] += (unsigned long long)cnt; s_acc[5] += (unsigned long long)(s_off[cnt] - s_off[0]); }
        int s = 0;
        while (s < cnt) {
            const int64_t off_s = s_off[s];
            const int cand = s + 1 + tid;
            const int okc = (cand <= cnt) && (s_off[min(cand, cnt)] - off_s <= (int64_t)(KM_CAP_BASES - 32));
            const int fit = uni(__syncthreads_count(okc));
            if (fit == 0) {                                         // a read longer than BBDUK_MAX_READ_LEN
                if (tid == 0) {
                    atomicMax((unsigned long long*)&counters[BBDUK_CTR_STATUS], (unsigned long long)(-BBDUK_ERR_READ_TOO_LONG));
                    s_a[s] = 0; s_id[s] = -1; s_thr[s] = RED == RED_SPLIT ? (int)(s_off[s + 1] - s_off[s]) : P.maxBadKmers; s_split[s] = 0;
                    if (RED == RED_SPLIT) { outLeft[r0 + s] = -1; outRight[r0 + s] = -1; }
                }
                s += 1;
                continue;
            }
            const int e = s + fit;
            const int64_t B0 = off_s, B1 = s_off[e];
            const int64_t A0 = B0 & ~15LL;
            const int nchunks = (int)((B1 - A0 + 15) >> 4);
            for (int c = tid; c < nchunks; c += BLOCK_THREADS) {
                uint32_t r, comp, valid;
                stage_chunk(P, bases, A0 + 16LL * c, totalBases, r, comp, valid);
                s_fwd[PLANE_PAD + nchunks - 1 - c] = r;
                s_cmp[PLANE_PAD + c] = comp;
                reinterpret_cast<uint16_t*>(s_nm)[c] = (uint16_t)(~valid & 0xFFFFu);
            }
            if (tid == 0 && (nchunks & 1)) reinterpret_cast<uint16_t*>(s_nm)[nchunks] = 0;
            __syncthreads();

            Planes Q; Q.fwd = s_fwd + PLANE_PAD; Q.cmp = s_cmp + PLANE_PAD; Q.nm = s_nm; Q.filt = s_filt; Q.T = nchunks * 16;
            Q.fwdBits = lds_bits_of(Q.fwd); Q.cmpBits = lds_bits_of(Q.cmp);
            for (int rd = s + wave; rd < e; rd += NWAVES) {        // one wave per read
                const int L = uni((int)(s_off[rd + 1] - s_off[rd]));
                const int base0 = uni((int)(s_off[rd] - A0));
                const int pairnum = paired ? (rd & 1) : 0;
                ReadScan R;
                R.base0 = base0; R.L = L; R.hasN = -1; R.maxBad = 0;
                R.start = span_start<true>(P, L); R.stop = span_stop<true>(P, L);
                const bool skipped = (P.skipR1 && pairnum == 0) || (P.skipR2 && pairnum == 1);
                if (RED == RED_BIG)       R.scan = P.storedKmers > 0 && L >= P.kbig && !skipped;                       // :1727-1728
                else if (RED == RED_BEST) R.scan = P.storedKmers > 0 && L >= k && !skipped;                            // :1661-1662
                else                      R.scan = P.storedKmers > 0 && L >= k;                                        // :2333, 2338 (unpaired)
                int thr = P.maxBadKmers;                            // :1056-1062 with keff = max(k, kbig)
                if (RED != RED_SPLIT && P.mkf != 0.f) {
                    const int keff = max(k, P.kbig);
                    const int vk = (L >= keff) ? valid_kmers_any_k(Q, base0, L, keff, lane) : 0;
                    thr = max(P.maxBadKmers, (int)((float)(vk - 1) * P.mkf));
                }
                KScanState S; kscan_init(S);
                if (R.scan) {
                    ReadWin W;
                    win_init<true, true>(P, Q, R, W, lane);
                    kscan_window<RED>(P, Q, W, S, thr, lane, counters);
                    kscan_finish<RED>(S, thr, P.kbig - k - 1, lane);
                }
                if (RED == RED_BEST) kscan_write_matches(P, S, thr, r0 + rd, lane);
                const int rid = S.rid, firstI = S.firstI, lastI = S.lastI;
                int found = S.found, id0 = S.id0;
                if (RED != RED_SPLIT) {
                    if (rid > 0) scaf_add(P, scaf, rid, L, lane, counters);
                    if (lane == 0) { s_a[rd] = found; s_id[rd] = rid; s_thr[rd] = thr; s_split[rd] = 0; }
                    continue;
                }
                // ---- ksplit: span of the main hits, else the short k-mers (right end first, :2388-2474)
                int leftmost = 0x7FFFFFFF, rightmost = -1;
                if (found > 0) { leftmost = max(0, firstI - (k - 1 - tp)); rightmost = lastI + tp; }
                if (R.scan && P.useShort && id0 == -1) {
                    const bool right = lane >= 32;                  // lanes 0-31: left end, lanes 32-63: right end; length mink + (lane&31)
                    const int Ls = P.mink + (lane & 31);
                    bool act; int i;
                    uint64_t km = 0, rr = 0;
                    if (!right) {
                        const int Lmax = min(k, R.stop) - R.start;
                        act = Ls <= Lmax; i = R.start + Ls - 1;
                        const int Lc = act ? Ls : 1;
                        if (act) { km = extract2(Q.fwd, Q.T - 1 - (base0 + R.start + Lc - 1), Lc) & P.mask; rr = extract2(Q.cmp, base0 + R.start, Lc); }
                    } else {
                        const int Lmax = (R.stop >= k ? k - 1 : R.stop);
                        act = Ls <= Lmax; i = R.stop - Ls;
                        const int Lc = act ? Ls : 1;
                        if (act) { km = extract2(Q.fwd, Q.T - 1 - (base0 + R.stop - 1), Lc); rr = extract2(Q.cmp, base0 + R.stop - Lc, Lc) & P.mask; }
                    }
                    if (P.qskip > 1) act = act && (i % P.qskip) == 0;
                    const int Lc = act ? Ls : 1;
                    const int sref = lookup<true>(P, Q.filt, km, rr, 1ULL << (2 * Lc), Lc, P.qhdist2, act);
                    const uint64_t hm = __ballot(sref != -1);
                    const uint32_t mL = (uint32_t)hm, mR = (uint32_t)(hm >> 32);
                    if (mR) {                                       // :2417-2427: every hit counts, the longest one reaches furthest left
                        id0 = __builtin_amdgcn_readlane(sref, 32 + __ffs(mR) - 1);        // first in loop order = shortest
                        const int iMin = R.stop - (P.mink + (31 - __clz(mR)));
                        leftmost = min(leftmost, max(0, iMin - tp)); rightmost = L - 1;
                        found += __popc(mR);
                    } else if (mL) {                                // :2434: only if the right end gave nothing
                        id0 = __builtin_amdgcn_readlane(sref, __ffs(mL) - 1);
                        const int iMax = R.start + (P.mink + (31 - __clz(mL))) - 1;
                        leftmost = 0; rightmost = max(rightmost, iMax + tp);
                        found += __popc(mL);
                    }
                }
                int trimmed = 0, npl = L, split = 0;
                if (found > 0) {
                    scaf_add(P, scaf, id0, L, lane, counters);
                    int n1 = L;
                    if (leftmost == 0) { trim_by_amount(L, rightmost + 1, 0, 1, n1); npl = n1; }                        // :2485-2487
                    else if (rightmost == L - 1) { trim_by_amount(L, 0, L - leftmost, 1, n1); npl = n1; }               // :2488-2490
                    else {                                                                                                // :2491-2498
                        const int n2 = (L - 1) - (rightmost + 1);   // subRead(rightmost+1, length-1): the copy excludes index length-1
                        trim_by_amount(L, 0, L - leftmost, 1, n1);
                        npl = n1 + n2; split = 1;
                    }
                    trimmed = L - npl;
                }
                if (lane == 0) {
                    s_a[rd] = trimmed; s_id[rd] = found > 0 ? id0 : -1; s_thr[rd] = npl; s_split[rd] = (uint8_t)split;
                    outLeft[r0 + rd] = found > 0 ? leftmost : -1; outRight[r0 + rd] = found > 0 ? rightmost : -1;
                }
            }
            __syncthreads();
            s = e;
        }
        // ---- record stage: one thread per read, mates look at each other
        if (tid < cnt) {
            const int L1 = (int)(s_off[tid + 1] - s_off[tid]);
            const int a = s_a[tid];
            outA[r0 + tid] = a; outId[r0 + tid] = s_id[tid];
            if (RED == RED_SPLIT) {                                 // :999-1013, 1028-1029, 1431-1443
                const bool remove = s_split[tid] != 0;             // remove=(r1.mate!=null): the two pieces go to outm together
                outFlags[r0 + tid] = (uint8_t)(remove ? BBDUK_FLAG_REMOVED : 0);
                if (a > 0) { atomicAdd(&s_acc[0], 1ULL); atomicAdd(&s_acc[1], (unsigned long long)a); }
                if (remove) { atomicAdd(&s_acc[2], 1ULL); atomicAdd(&s_acc[3], (unsigned long long)s_thr[tid]); }
            } else {                                                // :1064-1089
                const bool d = P.storedKmers > 0 && (RED == RED_BEST ? s_id[tid] > 0 : a > s_thr[tid]);
                bool remove = d;
                if (paired) {
                    const int m = tid ^ 1;
                    const bool dm = P.storedKmers > 0 && (RED == RED_BEST ? s_id[m] > 0 : s_a[m] > s_thr[m]);
                    remove = (P.rieb && (d || dm)) || (d && dm);
                }
                outFlags[r0 + tid] = (uint8_t)((d ? BBDUK_FLAG_DISCARDED : 0) | (remove ? BBDUK_FLAG_REMOVED : 0));
                if (remove) { atomicAdd(&s_acc[2], 1ULL); atomicAdd(&s_acc[3], (unsigned long long)L1); }
            }
        }
    }
    scaf_flush(P, scaf, lane, counters);
    __syncthreads();
    if (tid == 0) {
        if (RED == RED_SPLIT) {
            const unsigned long long rkt = s_acc[0], xs = s_acc[1], rm = s_acc[2], bm = s_acc[3], rin = s_acc[4], bin = s_acc[5];
            auto add = [&](int slot, unsigned long long v) { if (v) atomicAdd((unsigned long long*)&counters[slot], v); };
            add(BBDUK_READS_IN, rin); add(BBDUK_BASES_IN, bin);
            add(BBDUK_READS_KTRIMMED, rkt); add(BBDUK_BASES_KTRIMMED, xs);
            add(BBDUK_READS_OUTM, rm); add(BBDUK_BASES_OUTM, bm);                // pairCount stays 1 for a split read (:1437)
            add(BBDUK_READS_OUTU, rin - rm); add(BBDUK_BASES_OUTU, bin - xs - bm);
        } else publish_counters<BBDUK_MODE_KFILTER>(s_acc, counters);
    }
}

// --------------------------------------------------------------------------------------------------
// ktrim=rl / ktrimtips (bbduk/BBDukProcessorS.java:1813-1985): a right pass over [start, len) and then a left pass over
// [0, stop) of the read as the right pass left it.  Like bbduk_kmask_kernel a secondary operator on the run-time-general
// scan functions: one wave per read, the ktrim=r scan + finish with the right-hand span, then the ktrim=l scan + finish
// on the shortened read (the planes still hold it: a right trim keeps a prefix).  Outputs the two amounts separately.
__global__ __launch_bounds__(BLOCK_THREADS)
void bbduk_ktrimtips_kernel(const KParams P, const uint8_t* __restrict__ bases, const int64_t* __restrict__ offsets,
                            const int64_t n, const int64_t totalBases, const int paired,
                            int32_t* __restrict__ outRight, int32_t* __restrict__ outLeft, int32_t* __restrict__ outId,
                            uint8_t* __restrict__ outFlags, int64_t* __restrict__ counters, const int* __restrict__ longFlag) {
    if (PTF) return;                                             // trimfailuresto1bp: bbduk_wave_kernel reports BBDUK_ERR_UNSUPPORTED for such batches
    if (*longFlag & 2) return;                                    // a read beyond these planes: bbduk_long_tips_kernel takes the batch
    if (P.waveFirst && *longFlag == 0) return;                    // every unit fits a wave's planes: bbduk_wave_kernel<KTRIM_TIPS> did the batch
    __shared__ uint32_t s_fwd[PLANE_PAD + KM_CAP_CHUNKS + PLANE_PAD];
    __shared__ uint32_t s_cmp[PLANE_PAD + KM_CAP_CHUNKS + PLANE_PAD];
    __shared__ uint32_t s_nm[KM_CAP_CHUNKS / 2 + 4];
    __shared__ int64_t  s_off[TILE_READS + 1];
    __shared__ int32_t  s_xr[TILE_READS];
    __shared__ int32_t  s_xl[TILE_READS];
    __shared__ int32_t  s_len[TILE_READS];
    __shared__ int32_t  s_id[TILE_READS];
    __shared__ unsigned long long s_acc[6];                       // rkt, basesKTrimmed, readsOutm, basesOutm, readsIn, basesIn
    extern __shared__ uint32_t s_filt[];

    const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
    const int64_t ntiles = (n + TILE_READS - 1) / TILE_READS;
    const int k = P.k;
    ScafAcc scaf; scaf_init(scaf);
    if (tid < 6) s_acc[tid] = 0;
    if (P.ldsBits) {
        const int words = 1 << (P.ldsBits - 5);
        for (int w = tid; w < words; w += BLOCK_THREADS) s_filt[w] = P.ldsImage[w];
    }
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t r0 = tile * TILE_READS;
        const int cnt = (int)min((int64_t)TILE_READS, n - r0);
        __syncthreads();
        if (tid <= cnt) s_off[tid] = offsets[r0 + tid];
        __syncthreads();
        if (tid == 0) { s_acc[4] += (unsigned long long)cnt; s_acc[5] += (unsigned long long)(s_off[cnt] - s_off[0]); }
        int s = 0;
        while (s < cnt) {
            const int64_t off_s = s_off[s];
            const int cand = s + 1 + tid;
            const int okc = (cand <= cnt) && (s_off[min(cand, cnt)] - off_s <= (int64_t)(KM_CAP_BASES - 32));
            const int fit = uni(__syncthreads_count(okc));
            if (fit == 0) {
                if (tid == 0) { atomicMax((unsigned long long*)&counters[BBDUK_CTR_STATUS], (unsigned long long)(-BBDUK_ERR_READ_TOO_LONG));
                                s_xr[s] = 0; s_xl[s] = 0; s_id[s] = -1; s_len[s] = (int)(s_off[s + 1] - s_off[s]); }
                s += 1;
                continue;
            }
            const int e = s + fit;
            const int64_t B0 = off_s, B1 = s_off[e];
            const int64_t A0 = B0 & ~15LL;
            const int nchunks = (int)((B1 - A0 + 15) >> 4);
            for (int c = tid; c < nchunks; c += BLOCK_THREADS) {
                uint32_t r, comp, valid;
                stage_chunk(P, bases, A0 + 16LL * c, totalBases, r, comp, valid);
                s_fwd[PLANE_PAD + nchunks - 1 - c] = r;
                s_cmp[PLANE_PAD + c] = comp;
                reinterpret_cast<uint16_t*>(s_nm)[c] = (uint16_t)(~valid & 0xFFFFu);
            }
            if (tid == 0 && (nchunks & 1)) reinterpret_cast<uint16_t*>(s_nm)[nchunks] = 0;
            __syncthreads();

            Planes Q; Q.fwd = s_fwd + PLANE_PAD; Q.cmp = s_cmp + PLANE_PAD; Q.nm = s_nm; Q.filt = s_filt; Q.T = nchunks * 16;
            Q.fwdBits = lds_bits_of(Q.fwd); Q.cmpBits = lds_bits_of(Q.cmp);
            for (int ra = s + 2 * wave; ra < e; ra += 2 * NWAVES) {   // one wave per TWO consecutive reads: all four scan slots busy
                const bool hasB = (ra + 1) < e;
                const int rb = hasB ? ra + 1 : ra;
                int L[2], base0[2], pn[2], mid[2], cur[2], xr[2] = {0, 0}, xl[2] = {0, 0}, idr[2] = {-1, -1}, idl[2] = {-1, -1};
                L[0] = uni((int)(s_off[ra + 1] - s_off[ra])); L[1] = hasB ? uni((int)(s_off[rb + 1] - s_off[rb])) : 0;
                base0[0] = uni((int)(s_off[ra] - A0)); base0[1] = uni((int)(s_off[rb] - A0));
                pn[0] = paired ? (ra & 1) : 0; pn[1] = paired ? (rb & 1) : 0;
#pragma unroll
                for (int q = 0; q < 2; q++) { mid[q] = L[q] / 2 - (k - 1) / 2; cur[q] = L[q]; }      // :1815
                ReadScan R[2];
                auto reset = [&](ReadScan& X, int b0, int len, int start, int stop, bool scan) {
                    X.base0 = b0; X.L = len; X.hasN = -1; X.maxBad = 0; X.start = start; X.stop = stop; X.scan = scan;
                    X.found = 0; X.iFirst = BIGLOC; X.iLast = -1; X.ref = -1; X.shortFl = -1; X.shortLl = -1;
                };
                {   // right tips (:1817-1820): ktrimTip(r, start, len, right)
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        const int start = max(0, P.restrictRight < 1 ? mid[q] : L[q] - P.restrictRight);
                        reset(R[q], base0[q], cur[q], start, cur[q], (q == 0 || hasB) && scan_due<BBDUK_MODE_KTRIM_R, true, true>(P, cur[q], pn[q], true));
                    }
                    main_scan_pair<BBDUK_MODE_KTRIM_R, true, true>(P, Q, R[0], R[1], lane);
                    if (P.useShort) short_scan_pair<BBDUK_MODE_KTRIM_R, true>(P, Q, R[0], R[1], lane);
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        int a, newLen, ref; bool hit;
                        finish_read<BBDUK_MODE_KTRIM_R>(P, cur[q], R[q].start, R[q].stop, R[q].found, R[q].iFirst, R[q].iLast, R[q].shortFl, R[q].shortLl, R[q].ref, a, newLen, ref, hit);
                        if (R[q].scan) { if (hit) { idr[q] = ref; scaf_add(P, scaf, idr[q], cur[q], lane, counters); } xr[q] = a; cur[q] = newLen; }
                    }
                }
                {   // left tips (:1821-1824) on the reads as they are now
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        const int stop = min(cur[q], P.restrictLeft < 1 ? mid[q] + k - 1 : P.restrictLeft);
                        reset(R[q], base0[q], cur[q], 0, stop, (q == 0 || hasB) && scan_due<BBDUK_MODE_KTRIM_L, true, true>(P, cur[q], pn[q], true));
                    }
                    main_scan_pair<BBDUK_MODE_KTRIM_L, true, true>(P, Q, R[0], R[1], lane);
                    if (P.useShort) short_scan_pair<BBDUK_MODE_KTRIM_L, true>(P, Q, R[0], R[1], lane);
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        int a, newLen, ref; bool hit;
                        finish_read<BBDUK_MODE_KTRIM_L>(P, cur[q], R[q].start, R[q].stop, R[q].found, R[q].iFirst, R[q].iLast, R[q].shortFl, R[q].shortLl, R[q].ref, a, newLen, ref, hit);
                        if (R[q].scan) { if (hit) { idl[q] = ref; scaf_add(P, scaf, idl[q], cur[q], lane, counters); } xl[q] = a; cur[q] = newLen; }
                    }
                }
                if (lane == 0) {
                    s_xr[ra] = xr[0]; s_xl[ra] = xl[0]; s_len[ra] = cur[0]; s_id[ra] = idr[0] >= 0 ? idr[0] : idl[0];
                    if (hasB) { s_xr[rb] = xr[1]; s_xl[rb] = xl[1]; s_len[rb] = cur[1]; s_id[rb] = idr[1] >= 0 ? idr[1] : idl[1]; }
                }
            }
            __syncthreads();
            s = e;
        }
        // ---- record stage (:954-967, 1009-1033, 1431-1443): the even thread of a pair (every thread when unpaired) decides
        if (tid < cnt && (!paired || !(tid & 1))) {
            const bool two = paired != 0;
            const int l1 = (int)(s_off[tid + 1] - s_off[tid]), l2 = two ? (int)(s_off[tid + 2] - s_off[tid + 1]) : 0;
            int n1 = s_len[tid], n2 = two ? s_len[tid + 1] : 0;
            int xr1 = s_xr[tid], xr2 = two ? s_xr[tid + 1] : 0;
            const int xl1 = s_xl[tid], xl2 = two ? s_xl[tid + 1] : 0;
            const float g1 = (float)l1 * P.minLenFraction, g2 = (float)l2 * P.minLenFraction;
            const int minlen1 = (int)(g1 > (float)P.minReadLength ? g1 : (float)P.minReadLength);
            const int minlen2 = (int)(g2 > (float)P.minReadLength ? g2 : (float)P.minReadLength);
            bool d1 = false, d2 = false, remove = false;
            if (P.storedKmers > 0) {
                int xsum = xr1 + xl1 + xr2 + xl2, rkt = ((xr1 + xl1) > 0) + ((xr2 + xl2) > 0);
                d1 = n1 < minlen1; d2 = two && (n2 < minlen2);
                if ((P.rieb && (d1 || d2)) || (d1 && (!two || d2))) { xsum += n1 + n2; rkt = two ? 2 : 1; remove = true; }
                else if (P.tpe && xsum > 0 && two && n1 != n2) {      // trimpairsevenly: ktrimRight is set in this mode (:1021-1031)
                    int x;
                    if (n1 > n2) { x = trim_by_amount(n1, 0, n1 - n2, 1, n1); xr1 += x; }
                    else { x = trim_by_amount(n2, 0, n2 - n1, 1, n2); xr2 += x; }
                    if (rkt < 2) rkt++;
                    xsum += x;
                }
                atomicAdd(&s_acc[0], (unsigned long long)rkt); atomicAdd(&s_acc[1], (unsigned long long)xsum);
            }
            if (remove) { atomicAdd(&s_acc[2], two ? 2ULL : 1ULL); atomicAdd(&s_acc[3], (unsigned long long)(n1 + n2)); }
            const uint8_t f1 = (uint8_t)((d1 ? BBDUK_FLAG_DISCARDED : 0) | (remove ? BBDUK_FLAG_REMOVED : 0));
            const uint8_t f2 = (uint8_t)((d2 ? BBDUK_FLAG_DISCARDED : 0) | (remove ? BBDUK_FLAG_REMOVED : 0));
            outRight[r0 + tid] = xr1; outLeft[r0 + tid] = xl1; outId[r0 + tid] = s_id[tid]; outFlags[r0 + tid] = f1;
            if (two) { outRight[r0 + tid + 1] = xr2; outLeft[r0 + tid + 1] = xl2; outId[r0 + tid + 1] = s_id[tid + 1]; outFlags[r0 + tid + 1] = f2; }
        }
    }
    scaf_flush(P, scaf, lane, counters);
    __syncthreads();
    if (tid == 0) publish_counters<BBDUK_MODE_KTRIM_R>(s_acc, counters);
}

// findBestMatch on the wave kernel's pair scan: the two reads of a block keep their own id lists (lane j = the j-th distinct id of that
// read); readA / readB = their batch indices for the match lists of rename.  A.found / A.ref = hits counted / the scaffold returned.
#define BBDUK_MODE_FBM 6                            // internal: kfilter with findBestMatch, as a mode of bbduk_wave_kernel
template <bool FORBIDN, bool GENERAL>
__device__ __forceinline__ void main_scan_pair_best(const KParams& P, const Planes& Q, ReadScan& A, ReadScan& B, const int lane,
                                                    int64_t* __restrict__ counters, const int64_t readA, const int64_t readB, const bool hasB) {
    ReadWin WA, WB;
    win_init<FORBIDN, GENERAL, false, GENERAL>(P, Q, A, WA, lane);
    win_init<FORBIDN, GENERAL, false, GENERAL>(P, Q, B, WB, lane);
    int ibA = WA.first, ibB = WB.first;
    bool onA = WA.on, onB = WB.on;
    KScanState SA, SB; kscan_init(SA); kscan_init(SB);
    while (onA || onB) {
        uint64_t kmer[4], rk[4]; bool ok[4]; int id[4];
        windows2<FORBIDN, GENERAL, GENERAL>(P, Q, WA, ibA + 2 * lane, onA, kmer, rk, ok);
        windows2<FORBIDN, GENERAL, GENERAL>(P, Q, WB, ibB + 2 * lane, onB, kmer + 2, rk + 2, ok + 2);
        lookup4_probe<GENERAL>(P, Q.filt, kmer, rk, ok, id);
        const uint64_t m0 = __ballot(id[0] != -1), m1 = __ballot(id[1] != -1), m2 = __ballot(id[2] != -1), m3 = __ballot(id[3] != -1);
        if (onA) { if (m0 | m1) best_fold(SA, m0, m1, id[0], id[1], lane, counters); ibA += 128; onA = ibA < WA.stop; }
        if (onB) { if (m2 | m3) best_fold(SB, m2, m3, id[2], id[3], lane, counters); ibB += 128; onB = ibB < WB.stop; }
    }
    kscan_finish<RED_BEST>(SA, A.maxBad, 0, lane); kscan_finish<RED_BEST>(SB, B.maxBad, 0, lane);
    kscan_write_matches(P, SA, A.maxBad, readA, lane);
    if (hasB) kscan_write_matches(P, SB, B.maxBad, readB, lane);
    A.found = SA.found; A.ref = SA.rid; B.found = SB.found; B.ref = SB.rid;
}

// countSetKmersBig (k > 31: runs of consecutive matching 31-mers) on the wave kernel's pair scan; A.found / A.ref = the count / the scaffold
// returned, A.maxBad = the read's threshold
#define BBDUK_MODE_KBIG 7                           // internal: kfilter with kbig > k, as a mode of bbduk_wave_kernel
template <bool FORBIDN, bool GENERAL>
__device__ __forceinline__ void main_scan_pair_kbig(const KParams& P, const Planes& Q, ReadScan& A, ReadScan& B, const int lane) {
    ReadWin WA, WB;
    win_init<FORBIDN, GENERAL, false, GENERAL>(P, Q, A, WA, lane);
    win_init<FORBIDN, GENERAL, false, GENERAL>(P, Q, B, WB, lane);
    int ibA = WA.first, ibB = WB.first;
    bool onA = WA.on, onB = WB.on;
    const int sub = P.kbig - P.k - 1;
    KScanState SA, SB; kscan_init(SA); kscan_init(SB);
    while (onA || onB) {
        uint64_t kmer[4], rk[4]; bool ok[4]; int id[4];
        windows2<FORBIDN, GENERAL, GENERAL>(P, Q, WA, ibA + 2 * lane, onA, kmer, rk, ok);
        windows2<FORBIDN, GENERAL, GENERAL>(P, Q, WB, ibB + 2 * lane, onB, kmer + 2, rk + 2, ok + 2);
        lookup4_probe<GENERAL>(P, Q.filt, kmer, rk, ok, id);
        const uint64_t m0 = __ballot(id[0] != -1), m1 = __ballot(id[1] != -1), m2 = __ballot(id[2] != -1), m3 = __ballot(id[3] != -1);
        if (onA) {
            if ((m0 | m1) || SA.bkStart >= 0) big_fold(SA, __ballot(ok[0]), __ballot(ok[1]), id[0], id[1], ibA, sub, A.maxBad);
            ibA += 128; onA = !SA.done && ibA < WA.stop;
        }
        if (onB) {
            if ((m2 | m3) || SB.bkStart >= 0) big_fold(SB, __ballot(ok[2]), __ballot(ok[3]), id[2], id[3], ibB, sub, B.maxBad);
            ibB += 128; onB = !SB.done && ibB < WB.stop;
        }
    }
    kscan_finish<RED_BIG>(SA, A.maxBad, sub, lane); kscan_finish<RED_BIG>(SB, B.maxBad, sub, lane);
    A.found = SA.found; A.ref = SA.rid; B.found = SB.found; B.ref = SB.rid;
}

// --------------------------------------------------------------------------------------------------
// Wave-autonomous batch kernel (the fast path): every wave owns a mini-tile of MT_READS consecutive reads,
// stages it into its private slice of LDS and scans it, with no workgroup barrier after the one that lands
// the presence filter.  Waves of a CU therefore sit in different phases (HBM load, LDS extraction, L2 gather),
// which is what hides the latencies; with the tile-synchronous kernel above 52 % of all wave cycles were
// waits, much of it at barriers behind the slowest wave.  Requires every unit (pair) to fit WCAP_BASES;
// a pre-pass (bbduk_span_kernel) raises *slowFlag otherwise and the tile kernel takes the batch instead.
#define MT_READS     62                            // reads per wave mini-tile (even: whole pairs)
#ifndef WCAP_BASES
#define WCAP_BASES   2560                          // per-wave plane capacity in bases
#endif
#define WCAP_CHUNKS  (WCAP_BASES / 16)
#define WPLANE_WORDS (PLANE_PAD + WCAP_CHUNKS + PLANE_PAD)
#define WNM_WORDS    (WCAP_CHUNKS / 2 + 4)
#define SEL_BYTES    ((MT_READS + 3) & ~3)
#define WAVE_LDS_BYTES ((2 * NWAVES * WPLANE_WORDS + NWAVES * WNM_WORDS) * 4 + 6 * 8 + NWAVES * SEL_BYTES)   // behind the filter
#define WUNIT_MAX    (WCAP_BASES - 48)             // longest unit (pair) the wave kernel accepts
#define TAIL_MAX     32                            // positions a read may leave to the tail pass (wave_body<.., SHAPE>)
#define TRI_SHARE    2                             // ... or one read in TRI_SHARE is short enough for three to share a block (tri_scan_cand);
                                                   // a triple needs three short reads in a row: below a half the slower body 2 is not paid back
#define TAIL_SHARE   8                             // the tail-pass body takes a batch in which at least one read in TAIL_SHARE has a tail
// ktrim=n keeps a fourth per-wave plane (the k-mer end positions that hit).  To fit behind a 128 KiB filter its planes are a little shorter
// (sixteen 150-base reads still fit) and its short-scan lists live in the undefined-plane, which is dead once the main scan of the
// sub-tile is over.
#define WCAP_BASES_KM 2432
#define WPLANE_WORDS_KM (PLANE_PAD + WCAP_BASES_KM / 16 + PLANE_PAD)
#define WNM_WORDS_KM (WCAP_BASES_KM / 32 + 4)
#define WHIT_WORDS   (WCAP_BASES_KM / 32 + 2)
#define WAVE_LDS_BYTES_KM ((2 * NWAVES * WPLANE_WORDS_KM + NWAVES * WNM_WORDS_KM + NWAVES * WHIT_WORDS) * 4 + 6 * 8)
#define WUNIT_MAX_KM (WCAP_BASES_KM - 48)
static_assert((128 << 10) + WAVE_LDS_BYTES_KM <= (160 << 10), "ktrim=n wave kernel: LDS budget");
static_assert(WNM_WORDS_KM * 4 >= SEL_BYTES, "short-scan list fits the undefined-plane");

// sum of v over the lanes of a wave, returned wave-uniform
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return __builtin_amdgcn_readfirstlane(v);
}

#ifndef WAVE_KERNEL_ATTR
#define WAVE_KERNEL_ATTR
#endif
// Which kernel takes a batch of the first-hit scans: what the pre-pass counted (launch_batch), wave-uniform.  0 = bbduk_wave_kernel,
// 1 / 2 = bbduk_wave_shape_kernel with its tail-pass body / its three-reads-per-block body.
__device__ __forceinline__ int batch_shape(const int* __restrict__ slowFlag, const int64_t n) {
    const int tails = slowFlag[1], shorts = slowFlag[2];
    if (tails > 0 && (int64_t)tails * TAIL_SHARE >= n) return 1;
    if (shorts > 0 && (int64_t)shorts * TRI_SHARE >= n) return 2;
    return 0;
}

// SHAPE picks what the first-hit scans (ktrim=r, kfilter with maxbadkmers=0) do about reads the pair scan fits badly: 0 = nothing (the
// 2x150 shape: 128 positions per read and block), 1 = the tail pass, 2 = three short reads per block.  bbduk_wave_kernel is body 0,
// bbduk_wave_shape_kernel holds bodies 1 and 2 (see there).
template <int MODE, bool SHORT, bool FORBIDN, bool GENERAL, int FMT, bool BIG, int SHAPE>
__device__ __forceinline__
void wave_body(const KParams& P, const uint8_t* __restrict__ bases, const int64_t* __restrict__ offsets,
               const int64_t n, const int64_t totalBases, const int paired,
               int32_t* __restrict__ outA, int32_t* __restrict__ outId, uint8_t* __restrict__ outFlags,
               int64_t* __restrict__ counters, const int* __restrict__ slowFlag) {
    // One dynamic LDS block, the presence filter FIRST: its words are then addressed by the hash bits alone (LDS
    // address 0 + offset), which saves an add per lookup.  Behind it: per-wave planes, counters, short-scan lists.
    extern __shared__ uint32_t s_dyn[];
    uint32_t* const s_filt = s_dyn;
    constexpr bool KBIG = MODE == BBDUK_MODE_KBIG;                // kfilter with k > 31: runs of matching 31-mers (main_scan_pair_kbig)
    constexpr bool FBM = MODE == BBDUK_MODE_FBM;                  // kfilter with findBestMatch: per-read id lists in the pair scan (main_scan_pair_best)
    constexpr bool KMASK = MODE == BBDUK_MODE_KMASK;              // ktrim=n: every hit of the main scan, both ends' short k-mers, a mask per base
    constexpr bool TAILSCAN = SHAPE == 1, TRISCAN = SHAPE == 2;   // see "tails" below and tri_scan_cand
    constexpr bool TAILHITS = SHAPE == 0 && (MODE == BBDUK_MODE_KTRIM_L || MODE == BBDUK_MODE_KSPLIT || KMASK);   // the every-hit scans' tail pass (tail_scan_hits), always on
    static_assert(SHAPE == 0 || ((MODE == BBDUK_MODE_KTRIM_R || MODE == BBDUK_MODE_KFILTER) && !BIG), "shapes: first-hit scans of the cache-resident layout only");
    constexpr int CAPB = KMASK ? WCAP_BASES_KM : WCAP_BASES;      // per-wave plane capacity in bases
    constexpr int PLW = KMASK ? WPLANE_WORDS_KM : WPLANE_WORDS, NMW = KMASK ? WNM_WORDS_KM : WNM_WORDS, HW = KMASK ? WHIT_WORDS : 0;
    uint32_t* const s_wfAll = s_dyn + (P.ldsBits ? (1 << (P.ldsBits - 5)) : 0);
    uint32_t* const s_wcAll = s_wfAll + NWAVES * PLW;
    uint32_t* const s_wnAll = s_wcAll + NWAVES * PLW;
    uint32_t* const s_whAll = s_wnAll + NWAVES * NMW;             // ktrim=n: bit p <=> a k-mer ending at plane position p matched
    unsigned long long* const s_acc = reinterpret_cast<unsigned long long*>(s_whAll + NWAVES * HW);   // rkt, basesKTrimmed, readsOutm, basesOutm, readsIn, basesIn
    uint8_t* const s_selAll = reinterpret_cast<uint8_t*>(s_acc + 6);   // short-scan: compacted list of participating reads (lane ids)

    const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
    if (tid < 6) s_acc[tid] = 0;
    if (P.ldsBits) {
        const int words = 1 << (P.ldsBits - 5);
        for (int w = tid; w < words; w += BLOCK_THREADS) s_filt[w] = P.ldsImage[w];
    }
    __syncthreads();
    // slowFlag[0] != 0: a unit does not fit a wave's planes, the tile kernel's job; slowFlag[3] != 0 (bbduk_shape_kernel's verdict on the
    // pre-pass's counts): bbduk_wave_shape_kernel's batch.  One combined test, as before the shapes existed.
    const int sf0 = slowFlag[0];
    const int sf3 = (SHAPE == 0 && (MODE == BBDUK_MODE_KTRIM_R || MODE == BBDUK_MODE_KFILTER) && !BIG) ? slowFlag[3] : 0;
    if ((sf0 | sf3) != 0) {
        if (sf0 != 0 && PTF && tid == 0 && blockIdx.x == 0) atomicMax((unsigned long long*)&counters[BBDUK_CTR_STATUS], (unsigned long long)(-BBDUK_ERR_UNSUPPORTED));
        return;                                                   // (trimfailuresto1bp is served by this kernel only: the others stand back)
    }
    // short-scan geometry: `lens` candidate lengths per read, rpp reads per 64-lane pass
    // first-hit-only operators verify candidates in batches (see main_scan_pair_cand)
    // (a big-layout map has its fast candidate form in the BIG instantiations only; elsewhere it takes the exact scans)
    const bool candMode = (!GENERAL || P.qhdist == 0) && (BIG || !P.big) &&
                          (MODE == BBDUK_MODE_KTRIM_R || (MODE == BBDUK_MODE_KFILTER && P.maxBadKmers == 0 && P.mkf == 0.f && P.mcf == 0.f));
    constexpr bool TIPS = MODE == BBDUK_MODE_KTRIM_TIPS;          // ktrim=rl: a right pass over [mid, L), then a left pass over [0, mid+k-1) of what is left (:1813-1826)
    const int lens = max(1, (MODE == BBDUK_MODE_KTRIM_L || MODE == BBDUK_MODE_KSPLIT || TIPS || KMASK) ? (P.k - P.mink + 1) : (P.k - P.mink));
    const int rpp = max(1, 64 / lens);
    const int sslot = lane / lens, st = lane - sslot * lens;

    uint32_t* const wf = s_wfAll + wave * PLW; uint32_t* const wc = s_wcAll + wave * PLW;
    uint32_t* const wn = s_wnAll + wave * NMW;
    uint32_t* const wh = s_whAll + wave * HW;
    uint8_t* const sel = KMASK ? reinterpret_cast<uint8_t*>(wn) : s_selAll + wave * SEL_BYTES;   // (ktrim=n: see WAVE_LDS_BYTES_KM)
    // scaffold-counter cache: lane w (< SCAF_LANES) owns one (id, reads, bases) entry in registers; a hit is one ballot
    // plus a predicated add, a miss evicts round-robin with two atomics.  The adapter library has ~6 frequent ids:
    // with the 4-entry scalar cache 4-15 % of the hit reads still caused evictions onto a dozen hot addresses.
    constexpr int SCAF_LANES = 16;
    int scId = -1, scReads = 0, scNext = 0; long long scBases = 0;
    // per-lane partial sums (lane j accumulates what read j of every mini-tile contributes), reduced once at the end
    unsigned long long vRkt = 0, vXs = 0, vRm = 0, vBm = 0;
    unsigned long long sIn = 0, sBin = 0;                         // wave-uniform: reads / bases seen
    const int64_t nmt = (n + MT_READS - 1) / MT_READS;
    const int64_t gw = (int64_t)blockIdx.x * NWAVES + wave, nw = (int64_t)gridDim.x * NWAVES;

    int64_t myoffNext = (gw < nmt) ? offsets[gw * MT_READS + min((int64_t)lane, min((int64_t)MT_READS, n - gw * MT_READS))] : 0;
    for (int64_t mt = gw; mt < nmt; mt += nw) {
        const int64_t r0 = mt * MT_READS;
        const int cnt = (int)min((int64_t)MT_READS, n - r0);
        const int64_t myoff = myoffNext;
        {   // prefetch the next mini-tile's offsets: their HBM latency hides behind this mini-tile's work
            const int64_t mtn = mt + nw;
            if (mtn < nmt) myoffNext = offsets[mtn * MT_READS + min((int64_t)lane, min((int64_t)MT_READS, n - mtn * MT_READS))];
        }
        const int64_t O0 = ((int64_t)__builtin_amdgcn_readfirstlane((int)(myoff >> 32)) << 32) |
                           (uint32_t)__builtin_amdgcn_readfirstlane((int)myoff);
        const int rel = (int)(myoff - O0);                         // lane j (<= cnt): start of read j relative to the mini-tile
        const int relEnd = __builtin_amdgcn_readlane(rel, cnt);
        sIn += (unsigned long long)cnt; sBin += (unsigned long long)relEnd;
        // per-lane view of "my read" (lane j < cnt)
        int vL = __shfl_down(rel, 1) - rel;                        // length of read j (ktrim=rl: as the right pass left it, in the left pass)
        const int vL0 = vL;
        const bool mine = lane < cnt;
        const int vPairnum = paired ? (lane & 1) : 0;
        bool vScan; int vStart, vStop;
        const int vMid = vL0 / 2 - (P.k - 1) / 2;                   // :1815
        if constexpr (TIPS) {
            vScan = scan_due<BBDUK_MODE_KTRIM_R, SHORT, GENERAL>(P, vL, vPairnum, mine);
            vStart = max(0, (!GENERAL || P.restrictRight < 1) ? vMid : vL - P.restrictRight); vStop = vL;       // :1817-1820
        } else {
            vScan = scan_due<MODE, SHORT, GENERAL>(P, vL, vPairnum, mine);
            vStart = span_start<GENERAL>(P, vL); vStop = span_stop<GENERAL>(P, vL);
        }
        uint64_t scanMask = __ballot(vScan);
        int tXr = 0, tIdr = -1, tHitLen = 0;                        // ktrim=rl: what the right pass of my read gave (amount, scaffold, length it was credited with)
        // raw scan facts of my read, filled in by v_writelane as the pairs are scanned
        int vFound = 0, vFirst = BIGLOC, vLast = -1, vRef = -1, vSFl = -1, vSLl = -1;
        int vSide = 0;                                            // where vFound comes from: 0 main scan, 1 / 2 short k-mers of the right / left end
        int vCSlot = -1; uint32_t vCKeyLo = 0, vCKeyHi = 0;       // candidate mode: my read's first unverified match
        uint32_t vCWord = 0;                                      // big layout: its tag word (vCSlot = way)
        // ktrim=n: the lengths of my read's left / right end that matched (bit t <=> length mink+t) with the id of the shortest,
        // then hits in all, id of the first, masked bases
        uint32_t kSegL = 0, kSegR = 0; int kIdL = -1, kIdR = -1, kFound = 0, kId0 = -1, kCard = 0;
        int vThr = P.maxBadKmers;                                 // kfilter: my read's threshold (mkf: filled in when its pair is scanned)
        if constexpr (GENERAL && MODE == BBDUK_MODE_KFILTER) { if (P.mcf > 0.f) vThr = (int)ceilf(P.mcf * (float)vL); }

        int s = 0;
        while (s < cnt) {
            const int rel_s = __builtin_amdgcn_readlane(rel, s);
            const uint64_t okm = __ballot(lane > s && lane <= cnt && (rel - rel_s) <= (CAPB - 32));
            int fit = __popcll(okm);
            if (paired) fit &= ~1;
            if (fit == 0) {                                         // cannot happen when the span pre-pass ran
                if (lane == 0) atomicMax((unsigned long long*)&counters[BBDUK_CTR_STATUS], (unsigned long long)(-BBDUK_ERR_READ_TOO_LONG));
                s += min(paired ? 2 : 1, cnt - s);
                continue;
            }
            const int e = s + fit;
            const int rel_e = __builtin_amdgcn_readlane(rel, e);
            const int64_t B0 = O0 + rel_s;
            const int64_t A0 = B0 & ~15LL;
            const int lead = (int)(B0 - A0);                        // bases in front of read s inside the first chunk
            const int nchunks = (lead + (rel_e - rel_s) + 15) >> 4;
            // ---- stage this wave's reads: 16 bases per lane-iteration -> the wave's private bit-planes
            for (int c = lane; c < nchunks && !TSW(P, 5); c += 64) {
                uint32_t r, comp, valid;
                stage_chunk<FMT>(P, bases, A0 + 16LL * c, totalBases, r, comp, valid);
                wf[PLANE_PAD + nchunks - 1 - c] = r;
                wc[PLANE_PAD + c] = comp;
                reinterpret_cast<uint16_t*>(wn)[c] = (uint16_t)(~valid & 0xFFFFu);
            }
            if (lane == 0 && (nchunks & 1)) reinterpret_cast<uint16_t*>(wn)[nchunks] = 0;
            if constexpr (KMASK) { for (int w = lane; w < HW; w += 64) wh[w] = 0; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // LDS is in-order per wave; keep the compiler honest
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

            Planes Q; Q.fwd = wf + PLANE_PAD; Q.cmp = wc + PLANE_PAD; Q.nm = wn; Q.filt = s_filt; Q.T = nchunks * 16;
            Q.fwdBits = lds_bits_of(Q.fwd); Q.cmpBits = lds_bits_of(Q.cmp);
            const int origin = lead - rel_s;                        // plane index of a base = origin + (its offset in the mini-tile)
          for (int pass = 0; pass < (TIPS ? 2 : 1); pass++) {
            if constexpr (TIPS) {
                if (pass == 1) {                                    // the right pass is over for the reads of [s,e): its outcome, then the left pass's span
                    int a0, n0, ref0; bool hit0;
                    finish_read<BBDUK_MODE_KTRIM_R>(P, vL, vStart, vStop, vFound, vFirst, vLast, vSFl, vSLl, vRef, a0, n0, ref0, hit0);
                    if (lane >= s && lane < e) {
                        const bool sc = mine && vScan;
                        tXr = sc ? a0 : 0; tIdr = (sc && hit0) ? ref0 : -1; tHitLen = vL;
                        vL = sc ? n0 : vL;
                        vStart = 0; vStop = min(vL, (!GENERAL || P.restrictLeft < 1) ? vMid + P.k - 1 : P.restrictLeft);        // :1821-1824
                        vScan = scan_due<BBDUK_MODE_KTRIM_L, SHORT, GENERAL>(P, vL, vPairnum, mine);
                        vFound = 0; vFirst = BIGLOC; vLast = -1; vRef = -1; vSFl = -1; vSLl = -1; vSide = 0; vCSlot = -1;
                    }
                    scanMask = __ballot(vScan);
                }
            }
            const bool candP = TIPS ? (pass == 0 && (!GENERAL || P.qhdist == 0)) : candMode;
            uint64_t nMask = 0;                                     // reads of [s,e) with an undefined base inside their span
            if ((FORBIDN && P.forbidNs) || BIG) {                   // lane j looks at read j's words of the undefined-plane
                uint32_t acc = 0;
                if (lane >= s && lane < e && vScan) {
                    const int b0 = origin + rel + vStart, b1 = origin + rel + vStop;
                    for (int w = b0 >> 5; w <= ((b1 - 1) >> 5); w++) {
                        uint32_t v = wn[w];
                        const int lo = w << 5;
                        if (lo < b0) v &= ~0u << (b0 - lo);
                        if (lo + 32 > b1) v &= ~0u >> (lo + 32 - b1);
                        acc |= v;
                    }
                }
                nMask = __ballot(acc != 0u);
            }
            bool quadDone = false;
            if constexpr (TIPS) {
                if (pass == 0 && candP && (!GENERAL || P.qskip < 2) && !TSW(P, 3)) {
                    // the right pass: reads packed by the lanes their spans need, up to four per block (packed_scan_cand)
                    const int firstLook = max(P.k - 1, vStart > 0 ? vStart + P.minlen2 - 1 : 0);
                    const int vCnt = (vScan && lane >= s && lane < e && vStop > firstLook) ? ((vStop - firstLook + 1) >> 1) : 0;
                    if (__ballot(vCnt > 64) == 0ULL) {
                        quadDone = true;
                        const int vBase0p = origin + rel;
                        const bool anyN = (FORBIDN && P.forbidNs) && nMask != 0ULL;
                        int ra = s;
                        while (ra < e) {
                            int T4[5]; int nr = 0, lanes = 0;
                            T4[0] = 0;
#pragma unroll
                            for (int q = 0; q < 4; q++) {
                                const int c = (ra + q < e) ? __builtin_amdgcn_readlane(vCnt, min(ra + q, 63)) : 0;
                                const bool take = nr == q && (ra + q) < e && lanes + c <= 128;
                                if (take) { lanes += c; nr++; }
                                T4[q + 1] = lanes;
                            }
                            if (lanes > 0) packed_scan_cand<FORBIDN, GENERAL>(P, Q, ra, nr, T4, anyN, vBase0p, vStart, vStop, firstLook, lane, vCSlot, vFirst, vCKeyLo, vCKeyHi);
                            ra += nr;
                        }
                    }
                }
                if (pass == 1 && (!GENERAL || P.qskip < 2)) {       // the left pass: four reads per block (left_scan_quad)
                    quadDone = true;
                    for (int ra = s; ra < e; ra += 4) {
                        ReadScan R4[4];
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const int rd = min(ra + q, 63); const bool has = (ra + q) < e;
                            R4[q].hasN = (int)((nMask >> rd) & 1); R4[q].maxBad = P.maxBadKmers;
                            R4[q].base0 = origin + __builtin_amdgcn_readlane(rel, rd);
                            R4[q].L = has ? __builtin_amdgcn_readlane(vL, rd) : 0;
                            R4[q].scan = has && ((scanMask >> rd) & 1);
                            R4[q].start = 0; R4[q].stop = has ? __builtin_amdgcn_readlane(vStop, rd) : 0;
                            R4[q].found = 0; R4[q].iFirst = BIGLOC; R4[q].iLast = -1; R4[q].ref = -1; R4[q].shortFl = -1; R4[q].shortLl = -1;
                        }
                        left_scan_quad<FORBIDN, GENERAL>(P, Q, R4, lane);
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            if (R4[q].found) {
                                vFound = (lane == ra + q) ? R4[q].found : vFound; vFirst = (lane == ra + q) ? R4[q].iFirst : vFirst;
                                vLast = (lane == ra + q) ? R4[q].iLast : vLast;   vRef = (lane == ra + q) ? R4[q].ref : vRef;
                            }
                        }
                    }
                }
            }
            // ---- tails.  The pair scan gives a read 128 positions per block: a 150-base read with k=23 fills its half exactly, a 151-base read
            // would need a second block for ONE position.  A read that overshoots its last full block by at most TAIL_MAX positions leaves
            // them to the tail pass below, which serves the tails of up to 32 reads in one block (lane-varying read coordinates).
            int vTail = 0;
            uint64_t triMask = 0;                                   // short reads: three of them share a block (tri_scan_cand)
            if constexpr (TAILSCAN || TRISCAN || TAILHITS) {
                if ((TAILHITS || candP) && (!GENERAL || P.qskip < 2)) {
                    const int npos = (vScan && lane >= s && lane < e) ? max(0, vStop - max(vStart, P.k - 1)) : 0;
                    const int rem = npos & 127;
                    if constexpr (TAILSCAN || TAILHITS) vTail = (npos > 128 && rem >= 1 && rem <= TAIL_MAX) ? rem : 0;
                    if constexpr (TRISCAN) triMask = __ballot(lane >= s && lane < e && npos <= TRI_MAX);
                }
            }
            const uint64_t tailMask = (TAILSCAN || TAILHITS) ? __ballot(vTail > 0) : 0ULL;
            int step = 2;
            for (int ra = s; ra < e && !quadDone; ra += step) {
                if constexpr (TRISCAN) {
                    step = 2;
                    if (ra + 2 < e && ((triMask >> ra) & 7ULL) == 7ULL) {
                        ReadScan T3[3];
#pragma unroll
                        for (int q = 0; q < 3; q++) {
                            T3[q].base0 = origin + __builtin_amdgcn_readlane(rel, ra + q); T3[q].L = __builtin_amdgcn_readlane(vL, ra + q);
                            T3[q].scan = (scanMask >> (ra + q)) & 1; T3[q].hasN = (int)((nMask >> (ra + q)) & 1);
                            T3[q].start = span_start<GENERAL>(P, T3[q].L); T3[q].stop = span_stop<GENERAL>(P, T3[q].L);
                        }
                        tri_scan_cand<FORBIDN, GENERAL, SHORT && !GENERAL, GENERAL>(P, Q, T3, lane);
#pragma unroll
                        for (int q = 0; q < 3; q++) {
                            if (T3[q].candSlot != -1) {
                                vCSlot = (lane == ra + q) ? T3[q].candSlot : vCSlot; vFirst = (lane == ra + q) ? T3[q].iFirst : vFirst;
                                vCKeyLo = (lane == ra + q) ? T3[q].candKeyLo : vCKeyLo; vCKeyHi = (lane == ra + q) ? T3[q].candKeyHi : vCKeyHi;
                            }
                        }
                        step = 3;
                        continue;
                    }
                }
                const bool hasB = (ra + 1) < e;
                ReadScan A, Bz;
                A.hasN = (int)((nMask >> ra) & 1); Bz.hasN = (int)((nMask >> (ra + 1)) & 1);
                A.maxBad = P.maxBadKmers; Bz.maxBad = P.maxBadKmers;
                A.base0 = origin + __builtin_amdgcn_readlane(rel, ra);
                A.L = __builtin_amdgcn_readlane(vL, ra);
                A.scan = (scanMask >> ra) & 1;
                Bz.base0 = origin + __builtin_amdgcn_readlane(rel, ra + 1);
                Bz.L = hasB ? __builtin_amdgcn_readlane(vL, ra + 1) : 0;
                Bz.scan = hasB && ((scanMask >> (ra + 1)) & 1);
                A.start = span_start<GENERAL>(P, A.L); A.stop = span_stop<GENERAL>(P, A.L);
                Bz.start = span_start<GENERAL>(P, Bz.L); Bz.stop = span_stop<GENERAL>(P, Bz.L);
                if constexpr (TIPS) {
                    A.start = __builtin_amdgcn_readlane(vStart, ra); A.stop = __builtin_amdgcn_readlane(vStop, ra);
                    Bz.start = hasB ? __builtin_amdgcn_readlane(vStart, ra + 1) : 0; Bz.stop = hasB ? __builtin_amdgcn_readlane(vStop, ra + 1) : 0;
                }
                if constexpr (TAILSCAN || TAILHITS) {
                    if ((tailMask >> ra) & 3ULL) { A.stop -= __builtin_amdgcn_readlane(vTail, ra); if (hasB) Bz.stop -= __builtin_amdgcn_readlane(vTail, ra + 1); }
                }
                A.found = 0; A.iFirst = BIGLOC; A.iLast = -1; A.ref = -1; A.shortFl = -1; A.shortLl = -1;
                Bz.found = 0; Bz.iFirst = BIGLOC; Bz.iLast = -1; Bz.ref = -1; Bz.shortFl = -1; Bz.shortLl = -1;
                if (candP) {
                    if (!TSW(P, 3)) main_scan_pair_cand<FORBIDN, GENERAL, SHORT && !GENERAL && !TIPS, BIG, TIPS || GENERAL>(P, Q, A, Bz, lane); else { A.candSlot = -1; Bz.candSlot = -1; }
                    if (A.candSlot != -1) {
                        vCSlot = (lane == ra) ? A.candSlot : vCSlot; vFirst = (lane == ra) ? A.iFirst : vFirst;
                        vCKeyLo = (lane == ra) ? A.candKeyLo : vCKeyLo; vCKeyHi = (lane == ra) ? A.candKeyHi : vCKeyHi;
                        if constexpr (BIG) vCWord = (lane == ra) ? A.candWord : vCWord;
                    }
                    if (Bz.candSlot != -1) {
                        vCSlot = (lane == ra + 1) ? Bz.candSlot : vCSlot; vFirst = (lane == ra + 1) ? Bz.iFirst : vFirst;
                        vCKeyLo = (lane == ra + 1) ? Bz.candKeyLo : vCKeyLo; vCKeyHi = (lane == ra + 1) ? Bz.candKeyHi : vCKeyHi;
                        if constexpr (BIG) vCWord = (lane == ra + 1) ? Bz.candWord : vCWord;
                    }
                    continue;
                }
                if constexpr (GENERAL && MODE == BBDUK_MODE_KFILTER) {
                    if (P.mkf != 0.f || P.mcf > 0.f) {
                        A.maxBad = kfilter_threshold(P, Q, A.base0, A.L, lane);
                        Bz.maxBad = hasB ? kfilter_threshold(P, Q, Bz.base0, Bz.L, lane) : P.maxBadKmers;
                        vThr = (lane == ra) ? A.maxBad : ((lane == ra + 1) ? Bz.maxBad : vThr);
                    }
                }
                if constexpr (TIPS) {
                    if (pass == 0) main_scan_pair<BBDUK_MODE_KTRIM_R, FORBIDN, GENERAL, false, true, false, true>(P, Q, A, Bz, lane);
                    else main_scan_pair<BBDUK_MODE_KTRIM_L, FORBIDN, GENERAL, false, true, false, true>(P, Q, A, Bz, lane);
                } else if constexpr (KBIG) {
                    if constexpr (GENERAL) {
                        if (P.mkf != 0.f) {                         // :1056-1062 with keff = kbig: numValidKmers over windows longer than a plane word
                            const int va = (A.L >= P.kbig) ? valid_kmers_any_k(Q, A.base0, A.L, P.kbig, lane) : 0;
                            const int vb = (hasB && Bz.L >= P.kbig) ? valid_kmers_any_k(Q, Bz.base0, Bz.L, P.kbig, lane) : 0;
                            A.maxBad = max(P.maxBadKmers, (int)((float)(va - 1) * P.mkf)); Bz.maxBad = max(P.maxBadKmers, (int)((float)(vb - 1) * P.mkf));
                            vThr = (lane == ra) ? A.maxBad : ((lane == ra + 1) ? Bz.maxBad : vThr);
                        }
                    }
                    main_scan_pair_kbig<FORBIDN, GENERAL>(P, Q, A, Bz, lane);
                } else if constexpr (FBM) main_scan_pair_best<FORBIDN, GENERAL>(P, Q, A, Bz, lane, counters, r0 + ra, r0 + ra + 1, hasB);
                else if constexpr (KMASK) main_scan_pair<BBDUK_MODE_KTRIM_L, FORBIDN, GENERAL, false, GENERAL, true, true>(P, Q, A, Bz, lane, -1, wh);   // hits counted, first id kept, none ends the scan
                else main_scan_pair<MODE, FORBIDN, GENERAL, BIG, GENERAL, false, true>(P, Q, A, Bz, lane);
                if (A.found) {                                      // hand the facts to lane ra (most reads have none)
                    vFound = (lane == ra) ? A.found : vFound; vFirst = (lane == ra) ? A.iFirst : vFirst;
                    vLast = (lane == ra) ? A.iLast : vLast;   vRef = (lane == ra) ? A.ref : vRef;
                    vSFl = (lane == ra) ? A.shortFl : vSFl;   vSLl = (lane == ra) ? A.shortLl : vSLl;
                }
                if (Bz.found) {
                    vFound = (lane == ra + 1) ? Bz.found : vFound; vFirst = (lane == ra + 1) ? Bz.iFirst : vFirst;
                    vLast = (lane == ra + 1) ? Bz.iLast : vLast;   vRef = (lane == ra + 1) ? Bz.ref : vRef;
                    vSFl = (lane == ra + 1) ? Bz.shortFl : vSFl;   vSLl = (lane == ra + 1) ? Bz.shortLl : vSLl;
                }
            }
            if constexpr (TAILHITS) {
                if (tailMask) tail_scan_hits<KMASK ? BBDUK_MODE_KTRIM_L : MODE, FORBIDN, GENERAL, KMASK>(P, Q, sel, (FORBIDN && P.forbidNs) && (nMask & tailMask) != 0ULL, origin + rel, vStart, vStop,
                                                                                                      vTail, lane, vFound, vFirst, vLast, vRef, wh, s, e);
            }
            if constexpr (TAILSCAN) {
                if (tailMask) tail_scan_cand<FORBIDN, GENERAL, SHORT && !GENERAL, GENERAL>(P, Q, sel, (FORBIDN && P.forbidNs) && (nMask & tailMask) != 0ULL, origin + rel, vStart, vStop,
                                                                                               vTail, lane, vCSlot, vFirst, vCKeyLo, vCKeyHi);
            }
            if (candP) {
                // ---- verify the sub-tile's candidates together: lane j fetches key+id of read j's candidate
                const bool inSub = lane >= s && lane < e;
                bool fb = false;
                if (inSub && vCSlot <= -4) {                        // verified during the scan (chain walk)
                    vRef = -3 - vCSlot; vFound = (MODE == BBDUK_MODE_KFILTER) ? P.maxBadKmers + 1 : 1;
                    if (MODE == BBDUK_MODE_KFILTER) vFirst = 0;
                }
                if (inSub && vCSlot >= 0) {
                    bool same; int idv;
                    if constexpr (BIG) {                            // slot-parallel arrays: the key, and the id only if it is the key
                        const uint64_t slot = 4ULL * vCWord + (uint32_t)vCSlot;
                        same = P.bigKeys[slot] == ((((uint64_t)vCKeyHi << 32) | vCKeyLo) | P.kmask);
                        idv = same ? big_id_at(P, slot) : -1;
                    } else {
                        const uint4 kv = P.bkv[vCSlot];
                        same = kv.x == (vCKeyLo | (uint32_t)P.kmask) && kv.y == (vCKeyHi | (uint32_t)(P.kmask >> 32));
                        idv = (int)kv.z;
                    }
                    if (same) {
                        vRef = idv; vFound = (MODE == BBDUK_MODE_KFILTER) ? P.maxBadKmers + 1 : 1;
                        if (MODE == BBDUK_MODE_KFILTER) vFirst = 0;    // marks the early exit
                    } else fb = true;
                }
                uint64_t fbm = __ballot(fb);
                if constexpr (BIG) {
                    // an impostor fingerprint (twelve 15-bit fingerprints are compared per window: ~4 % of the pairs meet one): the
                    // read's candidate scan resumes right behind it, and its next candidate is verified at once
                    while (fbm) {
                        const int j = __ffsll((unsigned long long)fbm) - 1;
                        ReadScan A, Bz;
                        A.hasN = (int)((nMask >> j) & 1); A.maxBad = P.maxBadKmers;
                        A.base0 = origin + __builtin_amdgcn_readlane(rel, j); A.L = __builtin_amdgcn_readlane(vL, j); A.scan = true;
                        A.start = span_start<GENERAL>(P, A.L); A.stop = span_stop<GENERAL>(P, A.L);
                        A.found = 0; A.iFirst = BIGLOC; A.iLast = -1; A.ref = -1; A.shortFl = -1; A.shortLl = -1;
                        Bz = A; Bz.scan = false; Bz.L = 0; Bz.start = 0; Bz.stop = 0; Bz.hasN = 0;
                        main_scan_pair_cand<FORBIDN, GENERAL, SHORT && !GENERAL, true>(P, Q, A, Bz, lane, __builtin_amdgcn_readlane(vFirst, j) + 1);
                        bool done = true, hitNow = false; int nref = -1;
                        if (A.candSlot <= -4) { nref = -3 - A.candSlot; hitNow = true; }
                        else if (A.candSlot >= 0) {
                            const uint64_t slot = 4ULL * A.candWord + (uint32_t)A.candSlot;          // wave-uniform: every lane reads the same slot
                            if (P.bigKeys[slot] == ((((uint64_t)A.candKeyHi << 32) | A.candKeyLo) | P.kmask)) { nref = big_id_at(P, slot); hitNow = true; }
                            else done = false;                                                       // another impostor: go on behind it
                        }
                        if (hitNow) {
                            vRef = (lane == j) ? nref : vRef; vFound = (lane == j) ? ((MODE == BBDUK_MODE_KFILTER) ? P.maxBadKmers + 1 : 1) : vFound;
                            vFirst = (lane == j) ? ((MODE == BBDUK_MODE_KFILTER) ? 0 : A.iFirst) : vFirst;
                        } else vFirst = (lane == j) ? (done ? BIGLOC : A.iFirst) : vFirst;
                        if (done) fbm &= ~(1ULL << j);
                    }
                }
                while (fbm) {                                       // rare: an impostor fingerprint; rescan that pair exactly
                    const int j = __ffsll((unsigned long long)fbm) - 1;
                    const int ra = s + ((j - s) & ~1);
                    const bool hasB = (ra + 1) < e;
                    ReadScan A, Bz;
                    A.hasN = -1; Bz.hasN = -1; A.maxBad = P.maxBadKmers; Bz.maxBad = P.maxBadKmers;
                    A.base0 = origin + __builtin_amdgcn_readlane(rel, ra); A.L = __builtin_amdgcn_readlane(vL, ra); A.scan = (scanMask >> ra) & 1;
                    Bz.base0 = origin + __builtin_amdgcn_readlane(rel, ra + 1); Bz.L = hasB ? __builtin_amdgcn_readlane(vL, ra + 1) : 0;
                    Bz.scan = hasB && ((scanMask >> (ra + 1)) & 1);
                    A.start = span_start<GENERAL>(P, A.L); A.stop = span_stop<GENERAL>(P, A.L);
                    Bz.start = span_start<GENERAL>(P, Bz.L); Bz.stop = span_stop<GENERAL>(P, Bz.L);
                    if constexpr (TIPS) {
                        A.start = __builtin_amdgcn_readlane(vStart, ra); A.stop = __builtin_amdgcn_readlane(vStop, ra);
                        Bz.start = hasB ? __builtin_amdgcn_readlane(vStart, ra + 1) : 0; Bz.stop = hasB ? __builtin_amdgcn_readlane(vStop, ra + 1) : 0;
                    }
                    A.found = 0; A.iFirst = BIGLOC; A.iLast = -1; A.ref = -1; A.shortFl = -1; A.shortLl = -1;
                    Bz.found = 0; Bz.iFirst = BIGLOC; Bz.iLast = -1; Bz.ref = -1; Bz.shortFl = -1; Bz.shortLl = -1;
                    if constexpr (TIPS) main_scan_pair<BBDUK_MODE_KTRIM_R, FORBIDN, GENERAL, false, true, false, true>(P, Q, A, Bz, lane);
                    else main_scan_pair<MODE, FORBIDN, GENERAL, BIG, GENERAL, false, GENERAL>(P, Q, A, Bz, lane);   // (the probe form only where lookup4 would bring the query expansion along)
                    vFound = (lane == ra) ? A.found : vFound; vFirst = (lane == ra) ? A.iFirst : vFirst; vRef = (lane == ra) ? A.ref : vRef;
                    if (hasB) { vFound = (lane == ra + 1) ? Bz.found : vFound; vFirst = (lane == ra + 1) ? Bz.iFirst : vFirst; vRef = (lane == ra + 1) ? Bz.ref : vRef; }
                    fbm &= ~(3ULL << ra);
                }
            }
            // ---- short k-mers (:2034-2103) for the reads of [s,e) whose main scan found nothing, several reads per
            // pass: worker lane w looks up length mink+st of the read in slot sslot (lens lengths per read).  One pass serves one END
            // of the reads (LEFT: ktrim=l, :2037-2069; right: ktrim=r, :2072-2102); ksplit takes the right end first and the left end for
            // the reads that still have nothing (:2388-2474).
            if constexpr (MODE != BBDUK_MODE_KFILTER && SHORT) {
                auto short_pass = [&](auto leftTag, const bool need, const int side) {
                    constexpr bool LEFT = decltype(leftTag)::value;
                    uint64_t needM = __ballot(need);
                    if (!needM) return;
                    const int rank = __popcll(needM & ((1ULL << lane) - 1ULL));
                    if (need) sel[rank] = (uint8_t)lane;   // compact list of the reads that take part
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    const int nneed = __popcll(needM);
                    const int vBase0 = origin + rel;
                    // what the hits of one pass (reads sel[pb ..]) mean for their reads
                    auto take_hits = [&](uint64_t hm, const int sref, const int pb) {
                        while (hm) {                                // rare: some read has a short hit
                            const int l0 = __ffsll((unsigned long long)hm) - 1;
                            const int slot = l0 / lens;
                            const uint64_t seg = (hm >> (slot * lens)) & ((1ULL << lens) - 1ULL);
                            const int fl = __ffsll((unsigned long long)seg) - 1, ll = 63 - __clzll((long long)seg);
                            const int jr = sel[pb + slot];
                            const int r1 = __builtin_amdgcn_readlane(sref, slot * lens + fl);   // first hit in scan order = shortest
                            const int jru = __builtin_amdgcn_readfirstlane(jr);
                            if constexpr (KMASK) {
                                if (lane == jru) { if (side == 2) { kSegL = (uint32_t)seg; kIdL = r1; } else { kSegR = (uint32_t)seg; kIdR = r1; } }
                            } else if (lane == jru) { vRef = r1; vFound = __popcll(seg); vSFl = fl; vSLl = ll; vSide = side; }
                            hm &= ~(((1ULL << lens) - 1ULL) << (slot * lens));
                        }
                    };
                    {
                        // Two passes' probes in flight together.  Unpredicated cuts: inactive lanes cut a 1-base window.  (The general kernels add
                        // the span start, qskip, rcomp=f, speed; query expansion never reaches this kernel.)
                        const int Ls = P.mink + st;
                        // (ktrim=r keeps one probe per step: its scan is the headline's, where the second probe's registers cost more than the
                        // overlap returns -- 457 vs 450 Gbases/s)
                        constexpr int SPW = (MODE == BBDUK_MODE_KTRIM_R) ? 1 : 2;
                        for (int pb = 0; pb < nneed; pb += SPW * rpp) {
                            uint64_t kmer[2], rk[2], lm[2]; bool act[2]; int sref[2];
                            const bool two = SPW == 2 && pb + rpp < nneed;      // (wave-uniform) an odd pass at the end goes alone
#pragma unroll
                            for (int u = 0; u < SPW; u++) {
                                if (u == 1 && !two) break;
                                const int q = pb + u * rpp + sslot;
                                const bool have = sslot < rpp && q < nneed;
                                const int j = have ? sel[q] : 0;   // the read this worker lane serves
                                const int pk = __shfl(vBase0 | (vStop << 16), j);       // one shuffle carries base0 (< 2^16) and stop
                                const int jb = pk & 0xFFFF, jstop = (int)((unsigned)pk >> 16);
                                int jstart = 0;
                                if constexpr (GENERAL) jstart = __shfl(vStart, j);
                                if (LEFT) {
                                    const int Lmax = min(P.k, jstop) - jstart;         // lengths 1..Lmax, i = start+Ls-1
                                    act[u] = have && Ls <= Lmax;
                                    if constexpr (GENERAL) { if (P.qskip > 1) act[u] = act[u] && ((jstart + Ls - 1) % P.qskip) == 0; }
                                    const int Lc = act[u] ? Ls : 1;
                                    lm[u] = 1ULL << (2 * Lc);
                                    kmer[u] = cut64_lds(Q.fwdBits + 2u * (uint32_t)(Q.T - 1 - (jb + jstart + Lc - 1))) & (lm[u] - 1ULL);
                                    rk[u]   = cut64_lds(Q.cmpBits + 2u * (uint32_t)(jb + jstart)) & (lm[u] - 1ULL);
                                } else {
                                    const int Lmax = (jstop >= P.k ? P.k - 1 : jstop);  // lengths 1..Lmax, i = stop-Ls
                                    act[u] = have && Ls <= Lmax;
                                    if constexpr (GENERAL) { if (P.qskip > 1) act[u] = act[u] && ((jstop - Ls) % P.qskip) == 0; }
                                    const int Lc = act[u] ? Ls : 1;
                                    lm[u] = 1ULL << (2 * Lc);
                                    kmer[u] = cut64_lds(Q.fwdBits + 2u * (uint32_t)(Q.T - 1 - (jb + jstop - 1))) & (lm[u] - 1ULL);
                                    rk[u]   = cut64_lds(Q.cmpBits + 2u * (uint32_t)(jb + max(jstop - Lc, 0))) & (lm[u] - 1ULL);
                                }
                            }
                            if constexpr (SPW == 2) {
                                if (two) short_probe2<GENERAL>(P, kmer, rk, lm, act, sref);
                                else { sref[0] = short_probe<GENERAL>(P, kmer[0], rk[0], lm[0], act[0]); sref[1] = -1; }
                                take_hits(__ballot(sref[0] != -1), sref[0], pb);
                                if (two) take_hits(__ballot(sref[1] != -1), sref[1], pb + rpp);
                            } else {
                                sref[0] = short_probe<GENERAL>(P, kmer[0], rk[0], lm[0], act[0]);
                                take_hits(__ballot(sref[0] != -1), sref[0], pb);
                            }
                        }
                    }
                };
                if (P.useShort && !TSW(P, 4)) {
                    const bool need = mine && vScan && vFound == 0 && lane >= s && lane < e;
                    if constexpr (MODE == BBDUK_MODE_KSPLIT) {
                        short_pass(std::false_type{}, need, 1);
                        short_pass(std::true_type{}, mine && vScan && vFound == 0 && lane >= s && lane < e, 2);
                    } else if constexpr (KMASK) {                   // both ends, whatever the main scan found (:2200-2290)
                        const bool both = mine && vScan && lane >= s && lane < e;
                        short_pass(std::true_type{}, both, 2);
                        short_pass(std::false_type{}, both, 1);
                    } else if constexpr (TIPS) { if (pass == 0) short_pass(std::false_type{}, need, 1); else short_pass(std::true_type{}, need, 2); }
                    else if constexpr (MODE == BBDUK_MODE_KTRIM_L) short_pass(std::true_type{}, need, 2);
                    else short_pass(std::false_type{}, need, 1);
                }
            }
            if constexpr (KMASK) {
                // ---- the mask of the sub-tile's reads that met a hit, while their plane of hit positions is still here (:2190, 2236, 2279;
                // see bbduk_kmask_kernel): the wave takes them one at a time, 64 bases per step
                const int k = P.k, tp = P.trimPad;
                const bool inSub = mine && vScan && lane >= s && lane < e;
                int leftEnd = 0, rightStart = vL;                   // bases [0,leftEnd) and [rightStart,L) are masked by short k-mers
                if (inSub) {
                    kFound = vFound + __popc(kSegL) + __popc(kSegR);
                    kId0 = vFound > 0 ? vRef : (kSegL ? kIdL : kIdR);                       // main scan, then left hits, shortest first
                    if (!P.mfc) {
                        if (kSegL) { const int iMax = vStart + (P.mink + (31 - __clz(kSegL))) - 1; leftEnd = max(0, min(vL, iMax + tp + 1)); }
                        if (kSegR) { const int iMin = vStop - (P.mink + (31 - __clz(kSegR))); rightStart = min(vL, max(0, iMin - tp)); }
                    } else if (P.useShort) {
                        const int LmaxL = min(k, vStop) - vStart, LmaxR = (vStop >= k ? k - 1 : vStop);
                        const uint32_t actL = LmaxL >= P.mink ? (LmaxL - P.mink >= 31 ? ~0u : ((2u << (LmaxL - P.mink)) - 1u)) : 0u;
                        const uint32_t actR = LmaxR >= P.mink ? (LmaxR - P.mink >= 31 ? ~0u : ((2u << (LmaxR - P.mink)) - 1u)) : 0u;
                        const uint32_t missL = actL & ~kSegL, missR = actR & ~kSegR;
                        const int lenL = missL ? P.mink + (31 - __clz(missL)) : ((P.mink - 1 >= 1 && LmaxL >= P.mink - 1) ? P.mink - 1 : 0);
                        const int lenR = missR ? P.mink + (31 - __clz(missR)) : ((P.mink - 1 >= 1 && LmaxR >= P.mink - 1) ? P.mink - 1 : 0);
                        if (lenL > 0) leftEnd = max(0, min(vL, vStart + lenL - 1 + tp + 1));
                        if (lenR > 0) rightStart = min(vL, max(0, vStop - lenR - tp));
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                // one lane per read, 64 bases per step: H = the read's hit bits (zero outside [0, L)), the mask word = H dilated over the k
                // positions [b - tp, b + k - 1 - tp] (log-step OR of shifted copies), then the short k-mers' prefix / suffix
                const bool doCov = inSub && kFound > 0;
                const int L = vL, base0 = origin + rel;
                const int nb = P.mfc ? L : L + max(tp, 0) + 1;      // BitSet size: bits >= L count but are not written
                int myWords = doCov ? (nb + 63) >> 6 : 0;
                int maxWords = myWords;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) maxWords = max(maxWords, __shfl_xor(maxWords, o));
                maxWords = __builtin_amdgcn_readfirstlane(maxWords);
                auto prefix = [](const int n) -> uint64_t { return n <= 0 ? 0ULL : (n >= 64 ? ~0ULL : ((1ULL << n) - 1ULL)); };
                auto hits64 = [&](const int p0) -> uint64_t {       // hit bits of read positions [p0, p0 + 64)
                    if (p0 >= L || p0 + 64 <= 0) return 0ULL;
                    const int q = max(p0, 0), pos = base0 + q;
                    const uint32_t w0 = wh[pos >> 5], w1 = wh[(pos >> 5) + 1], w2 = wh[(pos >> 5) + 2];
                    uint64_t v = ((uint64_t)__builtin_amdgcn_alignbit(w2, w1, pos & 31) << 32) | __builtin_amdgcn_alignbit(w1, w0, pos & 31);
                    v &= prefix(L - q);
                    return p0 < 0 ? (v << (-p0)) : v;
                };
                int card = 0;
                const int64_t g0 = O0 + rel;                        // bit offset of my read in the output mask
                for (int w = 0; w < maxWords; w++) {
                    if (w < myWords) {
                        const int b0 = 64 * w;
                        uint64_t lo = hits64(b0 - tp), hi = hits64(b0 - tp + 64);
                        for (int cover = 1; cover < k;) {           // after the step the word ORs `cover` consecutive positions
                            const int c = min(cover, k - cover);
                            lo |= (lo >> c) | (hi << (64 - c)); hi |= hi >> c;
                            cover += c;
                        }
                        uint64_t cov = lo & prefix(nb - b0);
                        if (!P.mfc) cov |= (prefix(leftEnd - b0) | ~prefix(rightStart - b0)) & prefix(L - b0);
                        else cov = ~lo & ~prefix(leftEnd - b0) & prefix(rightStart - b0) & prefix(nb - b0);   // nothing cleared this base
                        card += __popcll(cov);
                        const uint64_t wm = cov & prefix(L - b0);   // bases only
                        if (wm) {                                   // up to three 32-bit words of the global mask
                            const int64_t g = g0 + b0; const int sh = (int)(g & 31);
                            const uint64_t plo = wm << sh; const uint32_t phi = sh ? (uint32_t)(wm >> (64 - sh)) : 0u;
                            uint32_t* const dst = P.outMask + (g >> 5);
                            if ((uint32_t)plo) atomicOr(dst, (uint32_t)plo);
                            if ((uint32_t)(plo >> 32)) atomicOr(dst + 1, (uint32_t)(plo >> 32));
                            if (phi) atomicOr(dst + 2, phi);
                        }
                    }
                }
                if (doCov) kCard = card;
            }
          }   // pass
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // scans done before the planes are overwritten
            __builtin_amdgcn_wave_barrier();
            s = e;
        }

        if (TSW(P, 7)) continue;                                 // experiment: no decide stage, no outputs
        // ---- finish, data-parallel: lane j turns read j's scan facts into outputs, pair logic via the neighbour lane
        int a, newLen, ref; bool hit;
        bool d = false, remove = false;
        int id = -1;
        if constexpr (MODE == BBDUK_MODE_KSPLIT) {
            // ksplit (:2332-2506, unpaired): the span of the main hits, else of the right end's short k-mers, else of the left end's
            // (vSide); then :2485-2498 -- trim at an end, or cut the span out and send the two pieces to outm as a pair
            const int tp = P.trimPad, k = P.k, L = vL;
            const bool any = mine && vScan && vFound > 0;
            int leftmost = 0x7FFFFFFF, rightmost = -1;
            if (vSide == 0) { leftmost = max(0, vFirst - (k - 1 - tp)); rightmost = vLast + tp; }
            else if (vSide == 1) { leftmost = max(0, (vStop - (P.mink + vSLl)) - tp); rightmost = L - 1; }
            else { leftmost = 0; rightmost = max(-1, (vStart + (P.mink + vSLl) - 1) + tp); }     // (:2434: max with the -1 it starts from; a negative trimpad)
            int npl = L, split = 0;
            if (any) {
                int n1 = L;
                if (leftmost == 0) { trim_by_amount(L, rightmost + 1, 0, 1, n1); npl = n1; }                        // :2485-2487
                else if (rightmost == L - 1) { trim_by_amount(L, 0, L - leftmost, 1, n1); npl = n1; }               // :2488-2490
                else {                                                                                                // :2491-2498
                    const int n2 = (L - 1) - (rightmost + 1);   // subRead(rightmost+1, length-1): the copy excludes index length-1
                    trim_by_amount(L, 0, L - leftmost, 1, n1);
                    npl = n1 + n2; split = 1;
                }
            }
            a = any ? L - npl : 0; newLen = npl; hit = any; ref = vRef;
            if (hit) id = ref_to_id(P, ref);
            remove = split != 0 && !PTF;                           // remove=(r1.mate!=null): the two pieces go to outm together (trimfailuresto1bp: they stay, :1431)
            if (mine) {
                outA[r0 + lane] = a; outId[r0 + lane] = id;
                outFlags[r0 + lane] = (uint8_t)(remove ? BBDUK_FLAG_REMOVED : 0);
                P.outLeft[r0 + lane] = any ? leftmost : -1; P.outRight[r0 + lane] = any ? rightmost : -1;
                vRkt += a > 0 ? 1u : 0u; vXs += (unsigned)a;
                if (remove) { vRm += 1; vBm += (unsigned)npl; }
                if (PTF) vBm += (unsigned)npl;
            }
        } else if constexpr (FBM || KBIG) {
            // findBestMatch (:1064-1089): discard iff a scaffold was returned; countSetKmersBig: iff the count passes the read's threshold
            // (the scaffold it returns is credited then); the counters are kfilter's
            hit = mine && vScan && vRef > 0; ref = vRef; a = (mine && vScan) ? vFound : 0; newLen = vL;
            if (hit) id = ref_to_id(P, ref);
            if (P.matchN && mine && !vScan) P.matchN[r0 + lane] = 0;
            int fLen = vL;
            if (P.storedKmers > 0) {
                d = KBIG ? (a > vThr) : hit; tf1bp(P, d, fLen);
                if (paired) { const bool dm = __shfl_xor((int)d, 1) != 0; remove = (P.rieb && (d || dm)) || (d && dm); }
                else remove = d;
            }
            if (mine) {
                outA[r0 + lane] = a; outId[r0 + lane] = (mine && vScan) ? vRef : -1;
                outFlags[r0 + lane] = (uint8_t)((d ? BBDUK_FLAG_DISCARDED : 0) | ((remove && !PTF) ? BBDUK_FLAG_REMOVED : 0));
                if (PTF) { if (remove) { vRkt += 1; vXs += (unsigned)vL; } vBm += (unsigned)fLen; }
                else if (remove) { vRm += 1; vBm += (unsigned)vL; }
            }
        } else if constexpr (KMASK) {
            // ktrim=n (:984-998, 1009-1016, 1028-1029, 1431-1443): lengths stay, so the verdicts depend on them alone; the counters take the
            // masked bases whether or not the pair is removed
            hit = mine && vScan && kFound > 0; ref = kId0; a = hit ? kCard : 0; newLen = vL;
            if (hit) id = ref_to_id(P, ref);
            const float g = (float)vL * P.minLenFraction;
            const int minlenR = (int)(g > (float)P.minReadLength ? g : (float)P.minReadLength);
            int fLen = vL;
            if (P.storedKmers > 0) {
                d = vL < minlenR; tf1bp(P, d, fLen);
                if (paired) { const bool dm = __shfl_xor((int)d, 1) != 0; remove = (P.rieb && (d || dm)) || (d && dm); }
                else remove = d;
            }
            if (mine) {
                outA[r0 + lane] = a; outId[r0 + lane] = id;
                outFlags[r0 + lane] = (uint8_t)((d ? BBDUK_FLAG_DISCARDED : 0) | ((remove && !PTF) ? BBDUK_FLAG_REMOVED : 0));
                vRkt += a > 0 ? 1u : 0u; vXs += (unsigned)a;
                if (PTF) vBm += (unsigned)fLen;
                else if (remove) { vRm += 1; vBm += (unsigned)vL; }
            }
        } else if constexpr (TIPS) {
            // ktrim=rl (:954-967, 1009-1033): the left pass's outcome on the read as the right pass left it, then the pair rules on the final
            // lengths (minlen from the ORIGINAL lengths, :812-813); outA = the right amount, P.outLeft = the left amount
            int aL, nL; bool hitL;
            finish_read<BBDUK_MODE_KTRIM_L>(P, vL, vStart, vStop, vFound, vFirst, vLast, vSFl, vSLl, vRef, aL, nL, ref, hitL);
            const bool scL = mine && vScan;
            hit = hitL && scL;                                      // (the left pass's credit; the right pass's is tIdr)
            int xr = tXr; const int xl = scL ? aL : 0;
            int n1 = scL ? nL : vL;
            id = tIdr >= 0 ? tIdr : (hit ? ref : -1);
            const float g = (float)vL0 * P.minLenFraction;
            const int minlenR = (int)(g > (float)P.minReadLength ? g : (float)P.minReadLength);
            const int nPre = n1;                                    // rlen: what the two passes left (:960, 966)
            if (P.storedKmers > 0) {
                d = n1 < minlenR; tf1bp(P, d, n1);
                if (paired) { const bool dm = __shfl_xor((int)d, 1) != 0; remove = (P.rieb && (d || dm)) || (d && dm); }
                else remove = d;
            }
            bool evened = false;
            if (P.tpe && paired && P.storedKmers > 0) {             // trimpairsevenly: ktrimRight is set in this mode (:1021-1031)
                const int xm = __shfl_xor(xr + xl, 1), nm = __shfl_xor(n1, 1);
                evened = mine && !remove && (xr + xl + xm) > 0 && n1 != nm;
                if (evened && n1 > nm) xr += trim_by_amount(n1, 0, n1 - nm, 1, n1);
            }
            a = xr; newLen = n1;
            if (mine) {
                outA[r0 + lane] = xr; P.outLeft[r0 + lane] = xl; outId[r0 + lane] = id;
                outFlags[r0 + lane] = (uint8_t)((d ? BBDUK_FLAG_DISCARDED : 0) | ((remove && !PTF) ? BBDUK_FLAG_REMOVED : 0));
                if (P.storedKmers > 0) {
                    vRkt += (remove || evened) ? 1u : ((xr + xl) > 0 ? 1u : 0u);
                    vXs += (unsigned)(xr + xl) + (remove ? (unsigned)nPre : 0u);
                }
                if (PTF) vBm += (unsigned)n1;
                else if (remove) { vRm += 1; vBm += (unsigned)n1; }
            }
        } else {
        finish_read<MODE>(P, vL, vStart, vStop, vFound, vFirst, vLast, vSFl, vSLl, vRef, a, newLen, ref, hit);
        hit = hit && mine && vScan;
        if (!(mine && vScan)) { a = 0; newLen = vL; }
        if (hit) id = ref_to_id(P, ref);
        const float g = (float)vL * P.minLenFraction;              // BBDukProcessorS.java:812-813
        const int minlenR = (int)(g > (float)P.minReadLength ? g : (float)P.minReadLength);
        // (two copies of the verdict block, chosen by one wave-uniform branch: with trimfailuresto1bp folded into a single copy the headline
        // kernel lost 1.2 % to the extra live values, measured)
        auto verdicts = [&](auto tfTag) {
            constexpr bool TF = decltype(tfTag)::value;
            const int nPre = newLen;                               // rlen: the length the k-trim left (:974, 980)
            if (P.storedKmers > 0) {
                d = (MODE != BBDUK_MODE_KFILTER) ? (newLen < minlenR) : ((GENERAL && P.mcf > 0.f) ? (a >= vThr) : (a > vThr));
                if constexpr (TF) { if (d && newLen > 1) newLen = 1; d = (newLen == 1); }       // setDiscarded / isDiscarded (:1464-1482)
                if (paired) {
                    const bool dm = __shfl_xor((int)d, 1) != 0;     // my mate's verdict
                    remove = (P.rieb && (d || dm)) || (d && dm);    // shouldRemove (:1489-1492)
                } else remove = d;
            }
            bool evened = false;
            if (MODE == BBDUK_MODE_KTRIM_R && P.tpe && paired && P.storedKmers > 0) {   // trimpairsevenly (:1021-1031)
                const int am = __shfl_xor(a, 1), nm = __shfl_xor(newLen, 1);
                evened = mine && !remove && (a + am) > 0 && newLen != nm;               // the same verdict in both mates' lanes
                if (evened && newLen > nm) a += trim_by_amount(newLen, 0, newLen - nm, 1, newLen);
            }
            if (mine) {
                outA[r0 + lane] = a; outId[r0 + lane] = id;
                outFlags[r0 + lane] = (uint8_t)((d ? BBDUK_FLAG_DISCARDED : 0) | ((remove && !TF) ? BBDUK_FLAG_REMOVED : 0));
                if (MODE != BBDUK_MODE_KFILTER) {                   // :1011-1029, per read: the pair's sums are the mates' sums
                    vRkt += (remove || evened) ? 1u : (a > 0 ? 1u : 0u);   // evened pairs count both mates (rktsum -> 2)
                    vXs += (unsigned)a + (remove ? (unsigned)(TF ? nPre : newLen) : 0u);
                } else if (TF && remove) { vRkt += 1; vXs += (unsigned)vL; }        // readsKFiltered / basesKFiltered (:1079-1088)
                if (TF) vBm += (unsigned)newLen;
                else if (remove) { vRm += 1; vBm += (unsigned)newLen; }
            }
        };
        if (PTF) verdicts(std::true_type{}); else verdicts(std::false_type{});
        }
        // scaffold counters (:2111-2119, :1577-1583): group the hit lanes by id, one cache update per distinct id
        auto credit = [&](const bool hit_, const int id_, const int len_) {
            uint64_t hm = __ballot(hit_);
            while (hm) {
                const int l0 = __ffsll((unsigned long long)hm) - 1;
                const int sid = __builtin_amdgcn_readlane(id_, l0);
                const bool same = hit_ && id_ == sid;
                const uint64_t sm = __ballot(same);
                const int nrd = __popcll(sm), nbs = wave_sum(same ? len_ : 0);
                const uint64_t mt_ = __ballot(lane < SCAF_LANES && scId == sid);
                if (mt_) { if (lane < SCAF_LANES && scId == sid) { scReads += nrd; scBases += nbs; } }
                else {
                    if (lane == scNext) {
                        if (scId > 0) {
                            atomicAdd((unsigned long long*)&counters[BBDUK_NCOUNTERS + scId], (unsigned long long)scReads);
                            atomicAdd((unsigned long long*)&counters[BBDUK_NCOUNTERS + P.numScaffolds + scId], (unsigned long long)scBases);
                        }
                        scId = sid; scReads = nrd; scBases = nbs;
                    }
                    scNext = (scNext + 1) & (SCAF_LANES - 1);
                }
                hm &= ~sm;
            }
        };
        if constexpr (TIPS) { credit(tIdr >= 0, tIdr, tHitLen); credit(hit, ref, vL); }     // each pass credits its own scaffold (:1817-1824)
        else credit(hit, id, vL);
    }
    if (lane < SCAF_LANES && scId > 0) {
        atomicAdd((unsigned long long*)&counters[BBDUK_NCOUNTERS + scId], (unsigned long long)scReads);
        atomicAdd((unsigned long long*)&counters[BBDUK_NCOUNTERS + P.numScaffolds + scId], (unsigned long long)scBases);
    }
    {   // wave reduction of the per-lane sums (64-bit, via two 32-bit halves is unnecessary: use shuffles on long long)
        unsigned long long t4[4] = {vRkt, vXs, vRm, vBm};
#pragma unroll
        for (int q = 0; q < 4; q++) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) t4[q] += __shfl_xor(t4[q], o);
        }
        if (lane == 0) {
#pragma unroll
            for (int q = 0; q < 4; q++) if (t4[q]) atomicAdd(&s_acc[q], t4[q]);
            if (sIn) atomicAdd(&s_acc[4], sIn);
            if (sBin) atomicAdd(&s_acc[5], sBin);
        }
    }
    __syncthreads();
    if (tid == 0) publish_counters<MODE == BBDUK_MODE_KTRIM_TIPS ? BBDUK_MODE_KTRIM_R : ((MODE == BBDUK_MODE_FBM || MODE == BBDUK_MODE_KBIG) ? BBDUK_MODE_KFILTER : MODE)>(s_acc, counters, PTF != 0);
}

template <int MODE, bool SHORT, bool FORBIDN, bool GENERAL, int FMT, bool BIG = false>
__global__ __launch_bounds__(BLOCK_THREADS) WAVE_KERNEL_ATTR
void bbduk_wave_kernel(const KParams P, const uint8_t* __restrict__ bases, const int64_t* __restrict__ offsets,
                       const int64_t n, const int64_t totalBases, const int paired,
                       int32_t* __restrict__ outA, int32_t* __restrict__ outId, uint8_t* __restrict__ outFlags,
                       int64_t* __restrict__ counters, const int* __restrict__ slowFlag) {
    wave_body<MODE, SHORT, FORBIDN, GENERAL, FMT, BIG, 0>(P, bases, offsets, n, totalBases, paired, outA, outId, outFlags, counters, slowFlag);
}

// The bodies for badly fitting read lengths, in a kernel of their own: inside bbduk_wave_kernel -- as extra code in its body or as further
// bodies behind a wave-uniform branch -- they cost the 2x150 hot loop 4-4.5 % through register allocation alone.  Both kernels are launched;
// this one returns before it touches anything unless the batch is its own (bbduk_wave_kernel stands back then: SHAPE 0's test in wave_body).
template <int MODE, bool SHORT, bool FORBIDN, bool GENERAL, int FMT>
__global__ __launch_bounds__(BLOCK_THREADS) WAVE_KERNEL_ATTR
void bbduk_wave_shape_kernel(const KParams P, const uint8_t* __restrict__ bases, const int64_t* __restrict__ offsets,
                             const int64_t n, const int64_t totalBases, const int paired,
                             int32_t* __restrict__ outA, int32_t* __restrict__ outId, uint8_t* __restrict__ outFlags,
                             int64_t* __restrict__ counters, const int* __restrict__ slowFlag) {
    static_assert(MODE == BBDUK_MODE_KTRIM_R || MODE == BBDUK_MODE_KFILTER, "first-hit scans only");
    if (slowFlag[0] != 0) return;
    const int shape = slowFlag[3];
    if (shape == 1) wave_body<MODE, SHORT, FORBIDN, GENERAL, FMT, false, 1>(P, bases, offsets, n, totalBases, paired, outA, outId, outFlags, counters, slowFlag);
    else if (shape == 2) wave_body<MODE, SHORT, FORBIDN, GENERAL, FMT, false, 2>(P, bases, offsets, n, totalBases, paired, outA, outId, outFlags, counters, slowFlag);
}

// --------------------------------------------------------------------------------------------------
// Reads longer than the LDS tiles (BBDUK_MAX_READ_LEN): one wave per unit again, but a read streams through the wave's
// private planes in chunks of LCHUNK k-mer end positions; each chunk is staged with the k-1 bases in front of it, the scan
// state (hits so far, first / last hit, credited id) is carried across chunks in the ReadScan, and the scan stops early
// where the reference's loop would not need to go on (ktrim=r after its first hit, kfilter at its exit).  The short k-mers
// of an end are looked up on a small chunk staged for that end.  Run-time-general scan functions; any read length that
// fits an int.  Takes the whole batch when the pre-pass finds such a read (long-read data sets consist of them).
#define LCHUNK (WCAP_BASES - 128)
// The chunk loop of a long read for one scan span [R.start, R.stop) (bbduk_long_kernel, bbduk_long_tips_kernel).
// stage(off, lo, hi) puts bases [lo, hi) of the read at `off` into the wave's planes and returns the plane index of base 0.
template <int MODE, bool BIG = false, class Stage>
__device__ __forceinline__ void long_scan(const KParams& P, const Planes& Q, Stage& stage, ReadScan& R, const int64_t off, const int lane) {
    if (!R.scan) return;
    const int k = P.k;
    ReadScan none; read_init<MODE, true, true>(P, none, 0, 0, 0, false);
    const int start = R.start, stop = R.stop;
    bool staged = false; int lastLo = 0;
    for (int ci = max(start, k - 1); ci < stop; ci += LCHUNK) {
        const int ce = min(stop, ci + LCHUNK);
        const int lo = max(start, ci - (k - 1));
        R.base0 = stage(off, lo, ce); staged = true; lastLo = lo;
        R.start = lo; R.stop = ce; R.hasN = -1;
        main_scan_pair<MODE, true, true, BIG>(P, Q, R, none, lane, ci);
        R.start = start; R.stop = stop;
        if (MODE == BBDUK_MODE_KTRIM_R && R.found > 0) break;                  // only the first hit matters (:2019-2030)
        if (MODE == BBDUK_MODE_KFILTER && R.iFirst == 0) break;                // countSetKmers / countCoveredBases returned
    }
    if (MODE != BBDUK_MODE_KFILTER && P.useShort && R.found == 0) {            // :2034-2103: the end's short k-mers
        // the end scans take the bases next to stop (right, regardless of start: :2072-2076) / between start and min(k, stop) (left)
        if (MODE == BBDUK_MODE_KTRIM_L) { if (start < min(k, stop)) R.base0 = stage(off, start, min(k, stop)); }
        else if (stop > 0 && (!staged || lastLo > max(0, stop - k))) R.base0 = stage(off, max(0, stop - k), stop);
        short_scan_pair<MODE, true>(P, Q, R, none, lane);
    }
}

template <int MODE, bool BIG = false>
__global__ __launch_bounds__(BLOCK_THREADS)
void bbduk_long_kernel(const KParams P, const uint8_t* __restrict__ bases, const int64_t* __restrict__ offsets,
                       const int64_t n, const int64_t totalBases, const int paired,
                       int32_t* __restrict__ outA, int32_t* __restrict__ outId, uint8_t* __restrict__ outFlags,
                       int64_t* __restrict__ counters, const int* __restrict__ slowFlag) {
    if (PTF) return;                                             // trimfailuresto1bp: bbduk_wave_kernel reports BBDUK_ERR_UNSUPPORTED for such batches
    if ((*slowFlag & 2) == 0) return;
    __shared__ uint32_t s_wf[NWAVES * WPLANE_WORDS];
    __shared__ uint32_t s_wc[NWAVES * WPLANE_WORDS];
    __shared__ uint32_t s_wn[NWAVES * WNM_WORDS];
    __shared__ unsigned long long s_acc[6];                       // rkt, basesKTrimmed, readsOutm, basesOutm, readsIn, basesIn
    extern __shared__ uint32_t s_filt[];
    const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
    const int k = P.k;
    ScafAcc scaf; scaf_init(scaf);
    if (tid < 6) s_acc[tid] = 0;
    if (P.ldsBits) {
        const int words = 1 << (P.ldsBits - 5);
        for (int w = tid; w < words; w += BLOCK_THREADS) s_filt[w] = P.ldsImage[w];
    }
    __syncthreads();
    uint32_t* const wf = s_wf + wave * WPLANE_WORDS; uint32_t* const wc = s_wc + wave * WPLANE_WORDS; uint32_t* const wn = s_wn + wave * WNM_WORDS;
    Planes Q; Q.fwd = wf + PLANE_PAD; Q.cmp = wc + PLANE_PAD; Q.nm = wn; Q.filt = s_filt; Q.T = 0;
    Q.fwdBits = lds_bits_of(Q.fwd); Q.cmpBits = lds_bits_of(Q.cmp);
    // stage bases [lo, hi) of the read at `off` into this wave's planes; returns the plane index of the read's base 0
    auto stage = [&](const int64_t off, const int lo, const int hi) -> int {
        const int64_t B0 = off + lo;
        const int64_t A0 = B0 & ~15LL;
        const int nchunks = (int)((off + hi - A0 + 15) >> 4);
        for (int c = lane; c < nchunks; c += 64) {
            uint32_t r, comp, valid;
            stage_chunk(P, bases, A0 + 16LL * c, totalBases, r, comp, valid);
            wf[PLANE_PAD + nchunks - 1 - c] = r;
            wc[PLANE_PAD + c] = comp;
            reinterpret_cast<uint16_t*>(wn)[c] = (uint16_t)(~valid & 0xFFFFu);
        }
        if (lane == 0) { reinterpret_cast<uint16_t*>(wn)[nchunks] = 0; reinterpret_cast<uint16_t*>(wn)[nchunks + 1] = 0; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        Q.T = nchunks * 16;
        return (int)(off - A0);
    };
    auto scan_read = [&](ReadScan& R, const int64_t off, const int pairnum, const bool present) {
        const int L = R.L;
        if (MODE == BBDUK_MODE_KFILTER && P.mcf > 0.f) R.maxBad = (int)ceilf(P.mcf * (float)L);    // :1040 (also for reads that are not scanned)
        if (MODE == BBDUK_MODE_KFILTER && P.mkf != 0.f && L < k) R.maxBad = max(P.maxBadKmers, (int)((float)(0 - 1) * P.mkf));
        if (!R.scan) return;
        if (MODE == BBDUK_MODE_KFILTER) {                           // thresholds that depend on the whole read
            if (P.mcf > 0.f) {}
            else if (P.mkf != 0.f) {                                // numValidKmers over the chunks, the run of defined bases carried along
                int cnt = 0, run = 0;
                for (int c0 = 0; c0 < L; c0 += LCHUNK) {
                    const int c1 = min(L, c0 + LCHUNK);
                    const int b0 = stage(off, c0, c1);
                    for (int i0 = c0; i0 < c1; i0 += 64) {
                        const int b = b0 + min(i0 + lane, c1 - 1);
                        const uint64_t U = __ballot(((Q.nm[b >> 5] >> (b & 31)) & 1u) != 0u);
                        const int nv = min(64, c1 - i0);
                        int pos = 0;
                        while (pos < nv) {
                            const uint64_t rest = U >> pos;
                            const int nextU = rest ? min(nv, pos + __ffsll((unsigned long long)rest) - 1) : nv;
                            const int seg = nextU - pos;
                            cnt += max(0, run + seg - max(run, k - 1));
                            run += seg;
                            if (nextU < nv) { run = 0; pos = nextU + 1; } else pos = nv;
                        }
                    }
                }
                R.maxBad = max(P.maxBadKmers, (int)((float)((L >= k ? cnt : 0) - 1) * P.mkf));
            }
        }
        long_scan<MODE, BIG>(P, Q, stage, R, off, lane);
    };
    const int step = paired ? 2 : 1;
    const int64_t units = (n + step - 1) / step;
    unsigned long long rIn = 0, bIn = 0;
    int acc[4] = {0, 0, 0, 0};
    for (int64_t u = (int64_t)blockIdx.x * NWAVES + wave; u < units; u += (int64_t)gridDim.x * NWAVES) {
        const int64_t ra = u * step;
        const bool hasB = paired && (ra + 1) < n;
        const int64_t o0 = offsets[ra], o1 = offsets[ra + 1], o2 = hasB ? offsets[ra + 2] : o1;
        ReadScan A, Bz;
        read_init<MODE, true, true>(P, A, 0, (int)(o1 - o0), 0, true);
        read_init<MODE, true, true>(P, Bz, 0, (int)(o2 - o1), 1, hasB);
        rIn += hasB ? 2 : 1; bIn += (unsigned long long)(o2 - o0);
        scan_read(A, o0, 0, true);
        if (hasB) scan_read(Bz, o1, 1, true);
        ReadOut OA, OB;
        read_finish<MODE>(P, A, OA, lane, scaf, counters);
        read_finish<MODE>(P, Bz, OB, lane, scaf, counters);
        uint8_t f1 = 0, f2 = 0;
        record_stage<MODE>(P, OA, hasB ? &OB : nullptr, acc, f1, f2);
        if (lane == 0) {
            outA[ra] = OA.a; outId[ra] = OA.id; outFlags[ra] = f1;
            if (hasB) { outA[ra + 1] = OB.a; outId[ra + 1] = OB.id; outFlags[ra + 1] = f2; }
#pragma unroll
            for (int q = 0; q < 4; q++) { if (acc[q]) atomicAdd(&s_acc[q], (unsigned long long)acc[q]); }     // per unit: these sums outgrow an int
        }
#pragma unroll
        for (int q = 0; q < 4; q++) acc[q] = 0;
    }
    scaf_flush(P, scaf, lane, counters);
    if (lane == 0) { atomicAdd(&s_acc[4], rIn); atomicAdd(&s_acc[5], bIn); }
    __syncthreads();
    if (tid == 0) publish_counters<MODE>(s_acc, counters);
}

// ktrim=n for sequences beyond bbduk_kmask_kernel's planes (contigs, long reads): one wave per such read, chunked like
// bbduk_long_kernel.  Every hit ORs its k (+trimPad) bases straight into the output mask (hits are rare; the in-LDS
// coverage pass of the tiled kernel needs the whole read), the short k-mers of both ends add their end ranges, and
// BitSet.cardinality() is read back from the mask words of this read plus the bits a positive trimPad pushes past the
// end.  Pair flags depend on lengths only and were written by bbduk_kmask_kernel; this kernel adds the read's counters.
__global__ __launch_bounds__(BLOCK_THREADS)
void bbduk_kmask_long_kernel(const KParams P, const uint8_t* __restrict__ bases, const int64_t* __restrict__ offsets,
                             const int64_t n, const int64_t totalBases, const int paired,
                             int32_t* __restrict__ outA, int32_t* __restrict__ outId, uint32_t* __restrict__ outMask,
                             int64_t* __restrict__ counters, const int* __restrict__ longFlag) {
    if (PTF) return;                                             // trimfailuresto1bp: bbduk_wave_kernel reports BBDUK_ERR_UNSUPPORTED for such batches
    if (*longFlag == 0) return;
    __shared__ uint32_t s_wf[NWAVES * WPLANE_WORDS];
    __shared__ uint32_t s_wc[NWAVES * WPLANE_WORDS];
    __shared__ uint32_t s_wn[NWAVES * WNM_WORDS];
    extern __shared__ uint32_t s_filt[];
    const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
    const int k = P.k, tp = P.trimPad;
    ScafAcc scaf; scaf_init(scaf);
    if (P.ldsBits) {
        const int words = 1 << (P.ldsBits - 5);
        for (int w = tid; w < words; w += BLOCK_THREADS) s_filt[w] = P.ldsImage[w];
    }
    __syncthreads();
    uint32_t* const wf = s_wf + wave * WPLANE_WORDS; uint32_t* const wc = s_wc + wave * WPLANE_WORDS; uint32_t* const wn = s_wn + wave * WNM_WORDS;
    Planes Q; Q.fwd = wf + PLANE_PAD; Q.cmp = wc + PLANE_PAD; Q.nm = wn; Q.filt = s_filt; Q.T = 0;
    Q.fwdBits = lds_bits_of(Q.fwd); Q.cmpBits = lds_bits_of(Q.cmp);
    auto stage = [&](const int64_t off, const int lo, const int hi) -> int {
        const int64_t A0 = (off + lo) & ~15LL;
        const int nchunks = (int)((off + hi - A0 + 15) >> 4);
        for (int c = lane; c < nchunks; c += 64) {
            uint32_t r, comp, valid;
            stage_chunk(P, bases, A0 + 16LL * c, totalBases, r, comp, valid);
            wf[PLANE_PAD + nchunks - 1 - c] = r;
            wc[PLANE_PAD + c] = comp;
            reinterpret_cast<uint16_t*>(wn)[c] = (uint16_t)(~valid & 0xFFFFu);
        }
        if (lane == 0) { reinterpret_cast<uint16_t*>(wn)[nchunks] = 0; reinterpret_cast<uint16_t*>(wn)[nchunks + 1] = 0; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        Q.T = nchunks * 16;
        return (int)(off - A0);
    };
    // set mask bits [g0+lo, g0+hi) (read-relative lo/hi clipped to [0, L)); wave-cooperative, 32 bits per lane
    auto set_range = [&](const int64_t g0, int lo, int hi, const int L) {
        lo = max(lo, 0); hi = min(hi, L);
        if (lo >= hi) return;
        const int64_t a = g0 + lo, b = g0 + hi;
        for (int64_t w = (a >> 5) + lane; w <= ((b - 1) >> 5); w += 64) {
            uint32_t m = ~0u;
            if ((w << 5) < a) m &= ~0u << (int)(a - (w << 5));
            if ((w << 5) + 32 > b) m &= ~0u >> (int)((w << 5) + 32 - b);
            atomicOr(&outMask[w], m);
        }
    };
    unsigned long long rkt = 0, xs = 0;
    for (int64_t rd = (int64_t)blockIdx.x * NWAVES + wave; rd < n; rd += (int64_t)gridDim.x * NWAVES) {
        const int64_t off = offsets[rd];
        const int64_t L64 = offsets[rd + 1] - off;
        if (L64 <= (int64_t)(KM_CAP_BASES - 32)) continue;          // bbduk_kmask_kernel did this one
        const int L = (int)L64;
        const int pairnum = paired ? (int)(rd & 1) : 0;
        ReadScan R;
        R.L = L; R.hasN = -1; R.maxBad = 0; R.base0 = 0;
        R.start = span_start<true>(P, L); R.stop = span_stop<true>(P, L);
        R.scan = P.storedKmers > 0 && L >= k && !((P.skipR1 && pairnum == 0) || (P.skipR2 && pairnum == 1));   // :2151-2154
        int found = 0, id0 = -1, iLast = -1;
        if (R.scan) {
            const int start = R.start, stop = R.stop;
            for (int ci = max(start, k - 1); ci < stop; ci += LCHUNK) {
                const int ce = min(stop, ci + LCHUNK);
                const int lo = max(start, ci - (k - 1));
                ReadScan C = R; C.base0 = stage(off, lo, ce); C.start = lo; C.stop = ce; C.hasN = -1;
                ReadWin W;
                win_init<true, true>(P, Q, C, W, lane);
                W.first = max(W.first, ci); W.on = W.first < W.stop;
                for (int ib = W.first; W.on && ib < W.stop; ib += 256) {
                    uint64_t kmer[4], rk[4]; bool ok[4]; int ref[4];
                    windows2<true, true>(P, Q, W, ib + 2 * lane, true, kmer, rk, ok);
                    if (ib + 128 < W.stop) windows2<true, true>(P, Q, W, ib + 128 + 2 * lane, true, kmer + 2, rk + 2, ok + 2);
                    else { kmer[2] = kmer[3] = 0; rk[2] = rk[3] = 0; ok[2] = ok[3] = false; }
                    lookup4<true>(P, Q.filt, kmer, rk, ok, ref);
#pragma unroll
                    for (int hb = 0; hb < 2; hb++) {
                        const uint64_t me = __ballot(ref[2 * hb] != -1), mo = __ballot(ref[2 * hb + 1] != -1);
                        if (P.mfc && !(me | mo)) {                  // fully covered, no match in these 128 positions: they clear one span
                            const int iA = ib + 128 * hb, iB = min(iA + 128, W.stop);
                            if (iA < iB) set_range(off, iA - (k - 1 - tp), iB - 1 + tp + 1, L);
                        }
                        if (!(me | mo)) continue;
                        const int i0 = ib + 128 * hb;
                        if (id0 < 0) {
                            const int le = me ? __ffsll((unsigned long long)me) - 1 : 64, lo2 = mo ? __ffsll((unsigned long long)mo) - 1 : 64;
                            id0 = (2 * lo2 + 1 < 2 * le) ? __builtin_amdgcn_readlane(ref[2 * hb + 1], lo2) : __builtin_amdgcn_readlane(ref[2 * hb], le);
                        }
                        const int he = me ? 63 - __clzll((unsigned long long)me) : -1, ho = mo ? 63 - __clzll((unsigned long long)mo) : -1;
                        iLast = max(iLast, i0 + max(2 * he, 2 * ho + 1));
                        found += __popcll(me) + __popcll(mo);
#pragma unroll
                        for (int par = 0; par < 2; par++) {            // bs.set(max(0,i-minus), i+plus) for this lane's hit (:2190);
                            const int ip = i0 + 2 * lane + par;        // fully covered: the lanes that do NOT match mark what they clear (:2194)
                            if (P.mfc ? (ip < W.stop && ref[2 * hb + par] == -1) : (ref[2 * hb + par] != -1)) {
                                const int i = ip;
                                const int b0 = max(0, i - (k - 1 - tp)), b1 = min(L, i + tp + 1);
                                if (b0 < b1) {
                                    const int64_t a = off + b0, b = off + b1;   // at most k+|tp| <= 64 bits: up to three words
                                    for (int64_t w = a >> 5; w <= ((b - 1) >> 5); w++) {
                                        uint32_t m = ~0u;
                                        if ((w << 5) < a) m &= ~0u << (int)(a - (w << 5));
                                        if ((w << 5) + 32 > b) m &= ~0u >> (int)((w << 5) + 32 - b);
                                        atomicOr(&outMask[w], m);
                                    }
                                }
                            }
                        }
                    }
                }
            }
            if (P.useShort) {                                       // both ends, always (:2199-2283); see bbduk_kmask_kernel
                int leftEnd = 0, rightStart = L;
                for (int side = 0; side < 2; side++) {
                    const bool right = side == 1;
                    const int Ls = P.mink + lane;                   // one length per lane (k - mink < 64)
                    bool act; int i; uint64_t km = 0, rr = 0;
                    const int b0 = right ? (R.stop > 0 ? stage(off, max(0, R.stop - k), R.stop) : 0)
                                         : (R.start < min(k, R.stop) ? stage(off, R.start, min(k, R.stop)) : 0);   // nothing to stage: no lane is active
                    if (!right) {
                        const int Lmax = min(k, R.stop) - R.start;
                        act = Ls <= Lmax; i = R.start + Ls - 1;
                        const int Lc = act ? Ls : 1;
                        if (act) { km = extract2(Q.fwd, Q.T - 1 - (b0 + R.start + Lc - 1), Lc) & P.mask; rr = extract2(Q.cmp, b0 + R.start, Lc); }
                    } else {
                        const int Lmax = (R.stop >= k ? k - 1 : R.stop);
                        act = Ls <= Lmax; i = R.stop - Ls;
                        const int Lc = act ? Ls : 1;
                        if (act) { km = extract2(Q.fwd, Q.T - 1 - (b0 + R.stop - 1), Lc); rr = extract2(Q.cmp, b0 + R.stop - Lc, Lc) & P.mask; }
                    }
                    if (P.qskip > 1) act = act && (i % P.qskip) == 0;
                    const int Lc = act ? Ls : 1;
                    const int sref = lookup<true>(P, Q.filt, km, rr, 1ULL << (2 * Lc), Lc, P.qhdist2, act);
                    const uint64_t hm = __ballot(sref != -1);
                    if (hm) {
                        if (id0 < 0) id0 = __builtin_amdgcn_readlane(sref, __ffsll((unsigned long long)hm) - 1);     // left side first, shortest first
                        found += __popcll(hm);
                    }
                    if (!P.mfc) {
                        if (hm) {
                            const int longest = P.mink + (63 - __clzll((unsigned long long)hm));
                            if (!right) leftEnd = max(0, min(L, R.start + longest - 1 + tp + 1));                       // :2236
                            else rightStart = min(L, max(0, R.stop - longest - tp));                                    // :2279
                        }
                    } else {                                        // the longest length that does not match clears its end (see bbduk_kmask_kernel)
                        const int Lmax = right ? (R.stop >= k ? k - 1 : R.stop) : (min(k, R.stop) - R.start);
                        const uint64_t actM = Lmax >= P.mink ? (Lmax - P.mink >= 63 ? ~0ULL : ((2ULL << (Lmax - P.mink)) - 1ULL)) : 0ULL;
                        const uint64_t miss = actM & ~hm;
                        const int lenM = miss ? P.mink + (63 - __clzll((unsigned long long)miss)) : ((P.mink - 1 >= 1 && Lmax >= P.mink - 1) ? P.mink - 1 : 0);
                        if (lenM > 0) { if (!right) leftEnd = max(0, min(L, R.start + lenM - 1 + tp + 1)); else rightStart = min(L, max(0, R.stop - lenM - tp)); }
                    }
                }
                if (found > 0 || P.mfc) { set_range(off, 0, leftEnd, L); set_range(off, rightStart, L, L); }
            }
        }
        int card = 0;
        if (P.mfc && R.scan) {                                      // what was marked is what is CLEARED: flip the read's bits, or drop them
            __threadfence();
            int c = 0;
            const int64_t a = off, b = off + L;
            for (int64_t w = (a >> 5) + lane; w <= ((b - 1) >> 5); w += 64) {
                uint32_t m = ~0u;
                if ((w << 5) < a) m &= ~0u << (int)(a - (w << 5));
                if ((w << 5) + 32 > b) m &= ~0u >> (int)((w << 5) + 32 - b);
                if (found > 0) c += __popc((atomicXor(&outMask[w], m) ^ m) & m); else atomicAnd(&outMask[w], ~m);
            }
            if (found > 0) { scaf_add(P, scaf, id0, L, lane, counters); card = wave_sum(c); }
        } else if (found > 0) {
            scaf_add(P, scaf, id0, L, lane, counters);
            __threadfence();
            int c = 0;
            const int64_t a = off, b = off + L;
            for (int64_t w = (a >> 5) + lane; w <= ((b - 1) >> 5); w += 64) {
                uint32_t v = atomicOr(&outMask[w], 0u);             // read at the coherence point: this wave's own atomics are in
                if ((w << 5) < a) v &= ~0u << (int)(a - (w << 5));
                if ((w << 5) + 32 > b) v &= ~0u >> (int)((w << 5) + 32 - b);
                c += __popc(v);
            }
            card = wave_sum(c) + max(0, min(iLast + tp + 1, L + max(tp, 0) + 1) - L);   // bits a positive trimPad pushes past the end
        }
        if (lane == 0) { outA[rd] = card; outId[rd] = found > 0 ? id0 : -1; }
        if (card > 0) { rkt += 1; xs += (unsigned long long)card; }
    }
    scaf_flush(P, scaf, lane, counters);
    if (lane == 0) {
        if (rkt) atomicAdd((unsigned long long*)&counters[BBDUK_READS_KTRIMMED], rkt);
        if (xs) atomicAdd((unsigned long long*)&counters[BBDUK_BASES_KTRIMMED], xs);
    }
}

// k>31, findBestMatch and ksplit for reads beyond bbduk_kscan_kernel's planes (ksplit's natural input are long reads with
// an adapter somewhere inside): one wave per unit, the reductions' state carried across chunks by kscan_window.
template <int RED>
__global__ __launch_bounds__(BLOCK_THREADS)
void bbduk_kscan_long_kernel(const KParams P, const uint8_t* __restrict__ bases, const int64_t* __restrict__ offsets,
                             const int64_t n, const int64_t totalBases, const int paired,
                             int32_t* __restrict__ outA, int32_t* __restrict__ outId, uint8_t* __restrict__ outFlags,
                             int32_t* __restrict__ outLeft, int32_t* __restrict__ outRight, int64_t* __restrict__ counters, const int* __restrict__ longFlag) {
    if (PTF) return;                                             // trimfailuresto1bp: bbduk_wave_kernel reports BBDUK_ERR_UNSUPPORTED for such batches
    if ((*longFlag & 2) == 0) return;
    __shared__ uint32_t s_wf[NWAVES * WPLANE_WORDS];
    __shared__ uint32_t s_wc[NWAVES * WPLANE_WORDS];
    __shared__ uint32_t s_wn[NWAVES * WNM_WORDS];
    __shared__ unsigned long long s_acc[6];
    extern __shared__ uint32_t s_filt[];
    const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
    const int k = P.k, tp = P.trimPad;
    ScafAcc scaf; scaf_init(scaf);
    if (tid < 6) s_acc[tid] = 0;
    if (P.ldsBits) {
        const int words = 1 << (P.ldsBits - 5);
        for (int w = tid; w < words; w += BLOCK_THREADS) s_filt[w] = P.ldsImage[w];
    }
    __syncthreads();
    uint32_t* const wf = s_wf + wave * WPLANE_WORDS; uint32_t* const wc = s_wc + wave * WPLANE_WORDS; uint32_t* const wn = s_wn + wave * WNM_WORDS;
    Planes Q; Q.fwd = wf + PLANE_PAD; Q.cmp = wc + PLANE_PAD; Q.nm = wn; Q.filt = s_filt; Q.T = 0;
    Q.fwdBits = lds_bits_of(Q.fwd); Q.cmpBits = lds_bits_of(Q.cmp);
    auto stage = [&](const int64_t off, const int lo, const int hi) -> int {
        const int64_t A0 = (off + lo) & ~15LL;
        const int nchunks = (int)((off + hi - A0 + 15) >> 4);
        for (int c = lane; c < nchunks; c += 64) {
            uint32_t r, comp, valid;
            stage_chunk(P, bases, A0 + 16LL * c, totalBases, r, comp, valid);
            wf[PLANE_PAD + nchunks - 1 - c] = r;
            wc[PLANE_PAD + c] = comp;
            reinterpret_cast<uint16_t*>(wn)[c] = (uint16_t)(~valid & 0xFFFFu);
        }
        if (lane == 0) { reinterpret_cast<uint16_t*>(wn)[nchunks] = 0; reinterpret_cast<uint16_t*>(wn)[nchunks + 1] = 0; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        Q.T = nchunks * 16;
        return (int)(off - A0);
    };
    // one read: found / credited id (kfilter variants) or trimmed / id0 / leftmost / rightmost / new pair length / split (ksplit)
    auto one = [&](const int64_t read, const int64_t off, const int L, const int pairnum, int& found, int& rid, int& thr, int& lm, int& rm, int& npl, int& split) {
        ReadScan R;
        R.base0 = 0; R.L = L; R.hasN = -1; R.maxBad = 0;
        R.start = span_start<true>(P, L); R.stop = span_stop<true>(P, L);
        const bool skipped = (P.skipR1 && pairnum == 0) || (P.skipR2 && pairnum == 1);
        if (RED == RED_BIG)       R.scan = P.storedKmers > 0 && L >= P.kbig && !skipped;
        else if (RED == RED_BEST) R.scan = P.storedKmers > 0 && L >= k && !skipped;
        else                      R.scan = P.storedKmers > 0 && L >= k;
        thr = P.maxBadKmers; lm = -1; rm = -1; npl = L; split = 0;
        if (RED != RED_SPLIT && P.mkf != 0.f) {                     // numValidKmers(keff) over the chunks
            const int keff = max(k, P.kbig);
            int cnt = 0, run = 0;
            for (int c0 = 0; c0 < L && L >= keff; c0 += LCHUNK) {
                const int c1 = min(L, c0 + LCHUNK);
                const int b0 = stage(off, c0, c1);
                for (int i0 = c0; i0 < c1; i0 += 64) {
                    const int b = b0 + min(i0 + lane, c1 - 1);
                    const uint64_t U = __ballot(((Q.nm[b >> 5] >> (b & 31)) & 1u) != 0u);
                    const int nv = min(64, c1 - i0);
                    int pos = 0;
                    while (pos < nv) {
                        const uint64_t rest = U >> pos;
                        const int nextU = rest ? min(nv, pos + __ffsll((unsigned long long)rest) - 1) : nv;
                        const int seg = nextU - pos;
                        cnt += max(0, run + seg - max(run, keff - 1));
                        run += seg;
                        if (nextU < nv) { run = 0; pos = nextU + 1; } else pos = nv;
                    }
                }
            }
            thr = max(P.maxBadKmers, (int)((float)(cnt - 1) * P.mkf));
        }
        KScanState S; kscan_init(S);
        if (R.scan) {
            const int start = R.start, stop = R.stop;
            for (int ci = max(start, k - 1); ci < stop && !S.done; ci += LCHUNK) {
                const int ce = min(stop, ci + LCHUNK);
                const int lo = max(start, ci - (k - 1));
                ReadScan C = R; C.base0 = stage(off, lo, ce); C.start = lo; C.stop = ce; C.hasN = -1;
                ReadWin W;
                win_init<true, true>(P, Q, C, W, lane);
                W.first = max(W.first, ci); W.on = W.first < W.stop;
                kscan_window<RED>(P, Q, W, S, thr, lane, counters);
            }
            kscan_finish<RED>(S, thr, P.kbig - k - 1, lane);
        }
        if (RED == RED_BEST) kscan_write_matches(P, S, thr, read, lane);
        found = S.found; rid = S.rid;
        if (RED != RED_SPLIT) { if (rid > 0) scaf_add(P, scaf, rid, L, lane, counters); return; }
        // ---- ksplit: span of the main hits, else the short k-mers of the right end, else of the left end (:2388-2474)
        int id0 = S.id0, leftmost = 0x7FFFFFFF, rightmost = -1;
        if (found > 0) { leftmost = max(0, S.firstI - (k - 1 - tp)); rightmost = S.lastI + tp; }
        if (R.scan && P.useShort && id0 == -1) {
            for (int side = 1; side >= 0 && id0 == -1; side--) {    // right first
                const bool right = side == 1;
                const int Ls = P.mink + lane;
                bool act; int i; uint64_t km = 0, rr = 0;
                const int b0 = right ? (R.stop > 0 ? stage(off, max(0, R.stop - k), R.stop) : 0)
                                         : (R.start < min(k, R.stop) ? stage(off, R.start, min(k, R.stop)) : 0);   // nothing to stage: no lane is active
                if (!right) {
                    const int Lmax = min(k, R.stop) - R.start;
                    act = Ls <= Lmax; i = R.start + Ls - 1;
                    const int Lc = act ? Ls : 1;
                    if (act) { km = extract2(Q.fwd, Q.T - 1 - (b0 + R.start + Lc - 1), Lc) & P.mask; rr = extract2(Q.cmp, b0 + R.start, Lc); }
                } else {
                    const int Lmax = (R.stop >= k ? k - 1 : R.stop);
                    act = Ls <= Lmax; i = R.stop - Ls;
                    const int Lc = act ? Ls : 1;
                    if (act) { km = extract2(Q.fwd, Q.T - 1 - (b0 + R.stop - 1), Lc); rr = extract2(Q.cmp, b0 + R.stop - Lc, Lc) & P.mask; }
                }
                if (P.qskip > 1) act = act && (i % P.qskip) == 0;
                const int Lc = act ? Ls : 1;
                const int sref = lookup<true>(P, Q.filt, km, rr, 1ULL << (2 * Lc), Lc, P.qhdist2, act);
                const uint64_t hm = __ballot(sref != -1);
                if (!hm) continue;
                id0 = __builtin_amdgcn_readlane(sref, __ffsll((unsigned long long)hm) - 1);      // first in loop order = shortest
                const int longest = P.mink + (63 - __clzll((unsigned long long)hm));
                if (right) { leftmost = min(leftmost, max(0, R.stop - longest - tp)); rightmost = L - 1; }
                else { leftmost = 0; rightmost = max(rightmost, R.start + longest - 1 + tp); }
                found += __popcll(hm);
            }
        }
        int trimmed = 0;
        if (found > 0) {
            scaf_add(P, scaf, id0, L, lane, counters);
            int n1 = L;
            if (leftmost == 0) { trim_by_amount(L, rightmost + 1, 0, 1, n1); npl = n1; }
            else if (rightmost == L - 1) { trim_by_amount(L, 0, L - leftmost, 1, n1); npl = n1; }
            else { const int n2 = (L - 1) - (rightmost + 1); trim_by_amount(L, 0, L - leftmost, 1, n1); npl = n1 + n2; split = 1; }
            trimmed = L - npl;
            lm = leftmost; rm = rightmost;
        }
        rid = found > 0 ? id0 : -1;
        found = trimmed;                                            // out_trimmed
    };
    const int step = (paired && RED != RED_SPLIT) ? 2 : 1;
    const int64_t units = (n + step - 1) / step;
    for (int64_t u = (int64_t)blockIdx.x * NWAVES + wave; u < units; u += (int64_t)gridDim.x * NWAVES) {
        const int64_t ra = u * step;
        const bool two = step == 2 && (ra + 1) < n;
        const int64_t o0 = offsets[ra], o1 = offsets[ra + 1], o2 = two ? offsets[ra + 2] : o1;
        const int l1 = (int)(o1 - o0), l2 = (int)(o2 - o1);
        int f1v, id1, thr1, lm1, rm1, npl1, sp1, f2v = 0, id2 = -1, thr2 = 0, lm2, rm2, npl2, sp2;
        one(ra, o0, l1, 0, f1v, id1, thr1, lm1, rm1, npl1, sp1);
        if (two) one(ra + 1, o1, l2, 1, f2v, id2, thr2, lm2, rm2, npl2, sp2);
        if (lane == 0) {
            atomicAdd(&s_acc[4], two ? 2ULL : 1ULL); atomicAdd(&s_acc[5], (unsigned long long)(o2 - o0));
            outA[ra] = f1v; outId[ra] = id1;
            if (two) { outA[ra + 1] = f2v; outId[ra + 1] = id2; }
            if (RED == RED_SPLIT) {
                outLeft[ra] = lm1; outRight[ra] = rm1;
                outFlags[ra] = (uint8_t)(sp1 ? BBDUK_FLAG_REMOVED : 0);
                if (f1v > 0) { atomicAdd(&s_acc[0], 1ULL); atomicAdd(&s_acc[1], (unsigned long long)f1v); }
                if (sp1) { atomicAdd(&s_acc[2], 1ULL); atomicAdd(&s_acc[3], (unsigned long long)npl1); }
            } else {
                const bool d1 = P.storedKmers > 0 && (RED == RED_BEST ? id1 > 0 : f1v > thr1);
                const bool d2 = two && P.storedKmers > 0 && (RED == RED_BEST ? id2 > 0 : f2v > thr2);
                const bool remove = two ? ((P.rieb && (d1 || d2)) || (d1 && d2)) : d1;
                outFlags[ra] = (uint8_t)((d1 ? BBDUK_FLAG_DISCARDED : 0) | (remove ? BBDUK_FLAG_REMOVED : 0));
                if (two) outFlags[ra + 1] = (uint8_t)((d2 ? BBDUK_FLAG_DISCARDED : 0) | (remove ? BBDUK_FLAG_REMOVED : 0));
                if (remove) { atomicAdd(&s_acc[2], two ? 2ULL : 1ULL); atomicAdd(&s_acc[3], (unsigned long long)(o2 - o0)); }
            }
        }
    }
    scaf_flush(P, scaf, lane, counters);
    __syncthreads();
    if (tid == 0) {
        if (RED == RED_SPLIT) {
            const unsigned long long rkt = s_acc[0], xs = s_acc[1], rm = s_acc[2], bm = s_acc[3], rin = s_acc[4], bin = s_acc[5];
            auto add = [&](int slot, unsigned long long v) { if (v) atomicAdd((unsigned long long*)&counters[slot], v); };
            add(BBDUK_READS_IN, rin); add(BBDUK_BASES_IN, bin);
            add(BBDUK_READS_KTRIMMED, rkt); add(BBDUK_BASES_KTRIMMED, xs);
            add(BBDUK_READS_OUTM, rm); add(BBDUK_BASES_OUTM, bm);
            add(BBDUK_READS_OUTU, rin - rm); add(BBDUK_BASES_OUTU, bin - xs - bm);
        } else publish_counters<BBDUK_MODE_KFILTER>(s_acc, counters);
    }
}

// ktrim=rl / ktrimtips for reads beyond bbduk_ktrimtips_kernel's planes (long-read adapter trimming): one wave per unit,
// the right pass and then the left pass of every read through long_scan, pair logic as in the tiled kernel's record stage.
__global__ __launch_bounds__(BLOCK_THREADS)
void bbduk_long_tips_kernel(const KParams P, const uint8_t* __restrict__ bases, const int64_t* __restrict__ offsets,
                            const int64_t n, const int64_t totalBases, const int paired,
                            int32_t* __restrict__ outRight, int32_t* __restrict__ outLeft, int32_t* __restrict__ outId,
                            uint8_t* __restrict__ outFlags, int64_t* __restrict__ counters, const int* __restrict__ longFlag) {
    if (PTF) return;                                             // trimfailuresto1bp: bbduk_wave_kernel reports BBDUK_ERR_UNSUPPORTED for such batches
    if ((*longFlag & 2) == 0) return;
    __shared__ uint32_t s_wf[NWAVES * WPLANE_WORDS];
    __shared__ uint32_t s_wc[NWAVES * WPLANE_WORDS];
    __shared__ uint32_t s_wn[NWAVES * WNM_WORDS];
    __shared__ unsigned long long s_acc[6];
    extern __shared__ uint32_t s_filt[];
    const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
    const int k = P.k;
    ScafAcc scaf; scaf_init(scaf);
    if (tid < 6) s_acc[tid] = 0;
    if (P.ldsBits) {
        const int words = 1 << (P.ldsBits - 5);
        for (int w = tid; w < words; w += BLOCK_THREADS) s_filt[w] = P.ldsImage[w];
    }
    __syncthreads();
    uint32_t* const wf = s_wf + wave * WPLANE_WORDS; uint32_t* const wc = s_wc + wave * WPLANE_WORDS; uint32_t* const wn = s_wn + wave * WNM_WORDS;
    Planes Q; Q.fwd = wf + PLANE_PAD; Q.cmp = wc + PLANE_PAD; Q.nm = wn; Q.filt = s_filt; Q.T = 0;
    Q.fwdBits = lds_bits_of(Q.fwd); Q.cmpBits = lds_bits_of(Q.cmp);
    auto stage = [&](const int64_t off, const int lo, const int hi) -> int {
        const int64_t A0 = (off + lo) & ~15LL;
        const int nchunks = (int)((off + hi - A0 + 15) >> 4);
        for (int c = lane; c < nchunks; c += 64) {
            uint32_t r, comp, valid;
            stage_chunk(P, bases, A0 + 16LL * c, totalBases, r, comp, valid);
            wf[PLANE_PAD + nchunks - 1 - c] = r;
            wc[PLANE_PAD + c] = comp;
            reinterpret_cast<uint16_t*>(wn)[c] = (uint16_t)(~valid & 0xFFFFu);
        }
        if (lane == 0) { reinterpret_cast<uint16_t*>(wn)[nchunks] = 0; reinterpret_cast<uint16_t*>(wn)[nchunks + 1] = 0; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        Q.T = nchunks * 16;
        return (int)(off - A0);
    };
    auto reset = [&](ReadScan& R, int len, int start, int stop, bool scan) {
        R.base0 = 0; R.L = len; R.hasN = -1; R.maxBad = 0; R.start = start; R.stop = stop; R.scan = scan;
        R.found = 0; R.iFirst = BIGLOC; R.iLast = -1; R.ref = -1; R.shortFl = -1; R.shortLl = -1;
    };
    // ktrimTips of one read (:1813-1828): returns the two amounts, the new length and the credited id
    auto tips = [&](const int64_t off, const int L, const int pairnum, int& xr, int& xl, int& cur, int& id) {
        const int mid = L / 2 - (k - 1) / 2;
        int idr = -1, idl = -1;
        xr = 0; xl = 0; cur = L;
        ReadScan A;
        {
            const int start = max(0, P.restrictRight < 1 ? mid : L - P.restrictRight);
            reset(A, cur, start, cur, scan_due<BBDUK_MODE_KTRIM_R, true, true>(P, cur, pairnum, true));
            long_scan<BBDUK_MODE_KTRIM_R>(P, Q, stage, A, off, lane);
            int a, newLen, ref; bool hit;
            finish_read<BBDUK_MODE_KTRIM_R>(P, cur, A.start, A.stop, A.found, A.iFirst, A.iLast, A.shortFl, A.shortLl, A.ref, a, newLen, ref, hit);
            if (A.scan) { if (hit) { idr = ref; scaf_add(P, scaf, idr, cur, lane, counters); } xr = a; cur = newLen; }
        }
        {
            const int stop = min(cur, P.restrictLeft < 1 ? mid + k - 1 : P.restrictLeft);
            reset(A, cur, 0, stop, scan_due<BBDUK_MODE_KTRIM_L, true, true>(P, cur, pairnum, true));
            long_scan<BBDUK_MODE_KTRIM_L>(P, Q, stage, A, off, lane);
            int a, newLen, ref; bool hit;
            finish_read<BBDUK_MODE_KTRIM_L>(P, cur, A.start, A.stop, A.found, A.iFirst, A.iLast, A.shortFl, A.shortLl, A.ref, a, newLen, ref, hit);
            if (A.scan) { if (hit) { idl = ref; scaf_add(P, scaf, idl, cur, lane, counters); } xl = a; cur = newLen; }
        }
        id = idr >= 0 ? idr : idl;
    };
    const int step = paired ? 2 : 1;
    const int64_t units = (n + step - 1) / step;
    for (int64_t u = (int64_t)blockIdx.x * NWAVES + wave; u < units; u += (int64_t)gridDim.x * NWAVES) {
        const int64_t ra = u * step;
        const bool two = paired && (ra + 1) < n;
        const int64_t o0 = offsets[ra], o1 = offsets[ra + 1], o2 = two ? offsets[ra + 2] : o1;
        const int l1 = (int)(o1 - o0), l2 = (int)(o2 - o1);
        int xr1, xl1, n1, id1, xr2 = 0, xl2 = 0, n2 = 0, id2 = -1;
        tips(o0, l1, 0, xr1, xl1, n1, id1);
        if (two) tips(o1, l2, 1, xr2, xl2, n2, id2);
        // record stage (:954-967, 1009-1033, 1431-1443), wave-uniform
        const float g1 = (float)l1 * P.minLenFraction, g2 = (float)l2 * P.minLenFraction;
        const int minlen1 = (int)(g1 > (float)P.minReadLength ? g1 : (float)P.minReadLength);
        const int minlen2 = (int)(g2 > (float)P.minReadLength ? g2 : (float)P.minReadLength);
        bool d1 = false, d2 = false, remove = false;
        long long xsum = 0; int rkt = 0;
        if (P.storedKmers > 0) {
            xsum = (long long)xr1 + xl1 + xr2 + xl2; rkt = ((xr1 + xl1) > 0) + ((xr2 + xl2) > 0);
            d1 = n1 < minlen1; d2 = two && (n2 < minlen2);
            if ((P.rieb && (d1 || d2)) || (d1 && (!two || d2))) { xsum += (long long)n1 + n2; rkt = two ? 2 : 1; remove = true; }
            else if (P.tpe && xsum > 0 && two && n1 != n2) {
                int x;
                if (n1 > n2) { x = trim_by_amount(n1, 0, n1 - n2, 1, n1); xr1 += x; }
                else { x = trim_by_amount(n2, 0, n2 - n1, 1, n2); xr2 += x; }
                if (rkt < 2) rkt++;
                xsum += x;
            }
        }
        if (lane == 0) {
            atomicAdd(&s_acc[0], (unsigned long long)rkt); atomicAdd(&s_acc[1], (unsigned long long)xsum);
            if (remove) { atomicAdd(&s_acc[2], two ? 2ULL : 1ULL); atomicAdd(&s_acc[3], (unsigned long long)((long long)n1 + n2)); }
            atomicAdd(&s_acc[4], two ? 2ULL : 1ULL); atomicAdd(&s_acc[5], (unsigned long long)(o2 - o0));
            const uint8_t f1 = (uint8_t)((d1 ? BBDUK_FLAG_DISCARDED : 0) | (remove ? BBDUK_FLAG_REMOVED : 0));
            const uint8_t f2 = (uint8_t)((d2 ? BBDUK_FLAG_DISCARDED : 0) | (remove ? BBDUK_FLAG_REMOVED : 0));
            outRight[ra] = xr1; outLeft[ra] = xl1; outId[ra] = id1; outFlags[ra] = f1;
            if (two) { outRight[ra + 1] = xr2; outLeft[ra + 1] = xl2; outId[ra + 1] = id2; outFlags[ra + 1] = f2; }
        }
    }
    scaf_flush(P, scaf, lane, counters);
    __syncthreads();
    if (tid == 0) publish_counters<BBDUK_MODE_KTRIM_R>(s_acc, counters);
}

// Pre-pass: does every unit (mate pair, or single read) fit a wave's planes?  One thread per unit.
// wmax / hmax: longest unit the first / the second kernel of the operator accepts (flag bit 0 / bit 1 otherwise).
// tailK >= 0 (launch_batch, first-hit scans): slowFlag[1] counts the reads whose k-mer end positions (length - tailK of them) overshoot the pair
// scan's 128-position blocks by 1..TAIL_MAX: enough of them and bbduk_wave_kernel runs its tail-pass body (wave_body<.., 1>); slowFlag[2] counts the reads with 1..TRI_MAX positions (wave_body<.., 2>).
__global__ void bbduk_span_kernel(const int64_t* __restrict__ offsets, const int64_t n, const int paired, int* __restrict__ slowFlag,
                                  const int64_t wmax = WUNIT_MAX, const int64_t hmax = CAP_BASES - 64, const int tailK = -1) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int step = paired ? 2 : 1;
    const int64_t units = n / step;
    bool bad = false, huge = false;
    int tails = 0, shorts = 0;
    for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < units; u += stride) {
        const int64_t o0 = offsets[u * step], o2 = offsets[u * step + step];
        const int64_t len = o2 - o0;
        bad |= len > wmax;
        huge |= len > hmax;                                       // not even the tile kernel's planes hold this unit: bbduk_long_kernel
        if (tailK >= 0) {
            const int64_t o1 = paired ? offsets[u * step + 1] : o2;
            const int64_t pa = (o1 - o0) - tailK, pb = (o2 - o1) - tailK;
            tails += (pa > 128 && (pa & 127) >= 1 && (pa & 127) <= TAIL_MAX) ? 1 : 0;
            tails += (paired && pb > 128 && (pb & 127) >= 1 && (pb & 127) <= TAIL_MAX) ? 1 : 0;
            shorts += (pa >= 1 && pa <= TRI_MAX) ? 1 : 0;
            shorts += (paired && pb >= 1 && pb <= TRI_MAX) ? 1 : 0;
        }
    }
    if (__ballot(bad) && (threadIdx.x & 63) == 0) atomicOr(slowFlag, 1);
    if (__ballot(huge) && (threadIdx.x & 63) == 0) atomicOr(slowFlag, 2);
    if (tailK >= 0) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { tails += __shfl_xor(tails, o); shorts += __shfl_xor(shorts, o); }
        if (tails > 0 && (threadIdx.x & 63) == 0) atomicAdd(slowFlag + 1, tails);
        if (shorts > 0 && (threadIdx.x & 63) == 0) atomicAdd(slowFlag + 2, shorts);
    }
}
// one thread: the pre-pass's counts -> which kernel takes the batch (slowFlag[3]: 0 bbduk_wave_kernel, 1 / 2 bbduk_wave_shape_kernel's bodies)
__global__ void bbduk_shape_kernel(int* __restrict__ slowFlag, const int64_t n) { slowFlag[3] = batch_shape(slowFlag, n); }

// runtime -> template dispatch
typedef void (*batch_kernel_t)(const KParams, const uint8_t*, const int64_t*, const int64_t, const int64_t, const int,
                               int32_t*, int32_t*, uint8_t*, int64_t*, const int*);
struct KernelPair { batch_kernel_t wave, wavePacked, tile, shape = nullptr, shapePacked = nullptr; };
template <int MODE, bool SHORT, bool FORBIDN, bool GENERAL>
static KernelPair kpair() {
    // the specialised wave kernels exist once per input format; the general one and the tile fallback decide per launch
    KernelPair kp = GENERAL ? KernelPair{bbduk_wave_kernel<MODE, SHORT, FORBIDN, GENERAL, 2>, bbduk_wave_kernel<MODE, SHORT, FORBIDN, GENERAL, 2>,
                                         bbduk_batch_kernel<MODE, SHORT, FORBIDN, GENERAL>}
                            : KernelPair{bbduk_wave_kernel<MODE, SHORT, FORBIDN, GENERAL, 0>, bbduk_wave_kernel<MODE, SHORT, FORBIDN, GENERAL, 1>,
                                         bbduk_batch_kernel<MODE, SHORT, FORBIDN, GENERAL>};
    if constexpr (MODE == BBDUK_MODE_KTRIM_R || MODE == BBDUK_MODE_KFILTER) {       // the first-hit scans have a kernel for badly fitting read lengths each
        kp.shape = GENERAL ? bbduk_wave_shape_kernel<MODE, SHORT, FORBIDN, GENERAL, 2> : bbduk_wave_shape_kernel<MODE, SHORT, FORBIDN, GENERAL, 0>;
        kp.shapePacked = GENERAL ? bbduk_wave_shape_kernel<MODE, SHORT, FORBIDN, GENERAL, 2> : bbduk_wave_shape_kernel<MODE, SHORT, FORBIDN, GENERAL, 1>;
    }
    return kp;
}
template <int MODE>
static KernelPair pick_kernel_mode(bool general, bool useShort, bool forbidN) {
    if (general) return kpair<MODE, true, true, true>();
    if (MODE == BBDUK_MODE_KFILTER) return forbidN ? kpair<MODE, false, true, false>() : kpair<MODE, false, false, false>();
    if (useShort) return forbidN ? kpair<MODE, true, true, false>() : kpair<MODE, true, false, false>();
    return forbidN ? kpair<MODE, false, true, false>() : kpair<MODE, false, false, false>();
}
static KernelPair pick_kernel(const KParams& K) {
    // the specialised kernels assume k >= 16 (BBDuk's usual 23-31) and what BBDukParser guarantees (mink turns
    // maskMiddle off, :295-301); anything else takes the general kernel
    const bool general = K.qhdist > 0 || K.qhdist2 > 0 || K.restrictLeft > 0 || K.restrictRight > 0 || K.skipR1 || K.skipR2 || !K.rcomp ||
                         (K.useShort && K.middleMask != ~0ULL) || K.k < 16 || K.qskip > 1 || K.speed > 0 || K.mkf != 0.f || K.mcf > 0.f;
    if (K.big) {
        // HBM-resident layout: chosen at build time only for the plain kfilter configurations (big_layout_eligible: BASELINE
        // configs[3]), whose first-hit scan has the minimizer-sharing candidate form; the exact scans (maxbadkmers > 0, impostors) and
        // the tile / long-read fallbacks are the BIG instantiations of the same functions
        const batch_kernel_t tile = bbduk_batch_kernel<BBDUK_MODE_KFILTER, true, true, true, true>;
        if (K.forbidNs) return KernelPair{bbduk_wave_kernel<BBDUK_MODE_KFILTER, false, true, false, 0, true>, bbduk_wave_kernel<BBDUK_MODE_KFILTER, false, true, false, 1, true>, tile};
        return KernelPair{bbduk_wave_kernel<BBDUK_MODE_KFILTER, false, false, false, 0, true>, bbduk_wave_kernel<BBDUK_MODE_KFILTER, false, false, false, 1, true>, tile};
    }
    if (K.mode == BBDUK_MODE_KFILTER) return pick_kernel_mode<BBDUK_MODE_KFILTER>(general, false, K.forbidNs != 0);
    if (K.mode == BBDUK_MODE_KTRIM_L) return pick_kernel_mode<BBDUK_MODE_KTRIM_L>(general, K.useShort != 0, K.forbidNs != 0);
    return pick_kernel_mode<BBDUK_MODE_KTRIM_R>(general, K.useShort != 0, K.forbidNs != 0);
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
        out[i] = (keys[i] < 0) ? -1 : (P.big ? big_find(P, key, mix_a(v), mix_b(v)) : table_get(P, key));
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
extern "C" const char* bbduk_last_error(const bbduk_handle* h) { return h ? h->err.c_str() : "null handle"; }

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
        hipMemset(h->d_counters, 0, nc * sizeof(int64_t)) != hipSuccess || hipMalloc(&h->d_slowFlag, 4 * bbduk_handle::EV_RING * sizeof(int)) != hipSuccess) { hipStreamDestroy(h->stream); delete h; return BBDUK_ERR_DEVICE; }
    *out = h;
    return BBDUK_OK;
}

static void build_release(bbduk_handle* h);

// Test-only controls (include/bbduk_test_hooks.h): explicit calls on a handle instead of environment variables.
extern "C" int bbduk_test_hook(bbduk_handle* h, int32_t which, int64_t value) {
    if (!h) return BBDUK_ERR_ARG;
    std::lock_guard<std::mutex> g(h->mu);
    switch (which) {
    case BBDUK_HOOK_FORCE_TILE:  h->hookForceTile = value != 0; return BBDUK_OK;
    case BBDUK_HOOK_BUCKET_BITS: if (h->finalized) return fail(h, BBDUK_ERR_STATE, "hook after finalize"); h->hookBucketBits = (int)value; return BBDUK_OK;
    case BBDUK_HOOK_LDS_BITS:    if (h->finalized) return fail(h, BBDUK_ERR_STATE, "hook after finalize"); h->hookLdsBits = (int)value; return BBDUK_OK;
    case BBDUK_HOOK_BIG_LAYOUT:  if (h->finalized) return fail(h, BBDUK_ERR_STATE, "hook after finalize"); h->hookBigLayout = value != 0; return BBDUK_OK;
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
    for (auto& q : h->slot) { hipFree(q.d_bases); hipFree(q.d_undef); hipFree(q.d_off); hipFree(q.d_a); hipFree(q.d_id); hipFree(q.d_fl); if (q.stream) hipStreamDestroy(q.stream); }
    hipFree(h->d_tags); hipFree(h->d_bkv);
    hipFree(h->d_ldsImage); hipFree(h->d_slowFlag);
    for (int q = 0; q < bbduk_handle::EV_RING; q++) { if (h->ev0[q]) hipEventDestroy(h->ev0[q]); if (h->ev1[q]) hipEventDestroy(h->ev1[q]); }
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
    bool big = false; int hdist = 0, hdist2 = 0;
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
    h->big = false; h->nbuckets = 0; h->bigLines = 0; h->ldsBits = 0; h->nkeys = 0;
}
// gapped-minimizer geometry of the big layout for this k and middle mask (see "big layout"); false: k too small for it
static bool big_geometry(bbduk_handle* h) {
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
    h->gH = H; h->gD = k - H; h->gm = std::min(10, H - 1); h->gW = h->bigPlain ? 0 : H - h->gm + 1;
    return true;
}
static BigGeom host_geom(const bbduk_handle* h) { BigGeom G; G.k = h->p.k; G.m = h->gm; G.W = h->gW; G.H = h->gH; G.D = h->gD; G.nlines = h->bigLines; G.middleMask = (uint64_t)h->p.middleMask; return G; }
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
static bool params_general(const bbduk_params& p) {               // the same predicate as pick_kernel's, on the boundary struct
    const bool useShort = p.mink > 0 && p.mink < p.k;
    return p.qhdist > 0 || p.qhdist2 > 0 || p.restrictLeft > 0 || p.restrictRight > 0 || p.skipR1 || p.skipR2 || !p.rcomp ||
           (useShort && p.middleMask != -1) || p.k < 16 || p.qSkip > 1 || p.speed > 0 || p.minKmerFraction != 0.f || p.minCoveredFraction > 0.f;
}
static bool big_layout_eligible(const bbduk_params& p) {
    return p.mode == BBDUK_MODE_KFILTER && !params_general(p) && !(p.kbig > p.k) && !p.findBestMatch;
}
#define BIG_LAYOUT_MIN_KEYS (1LL << 25)            // beyond ~3e7 keys the fingerprints alone outgrow L2 + Infinity Cache

// expected number of keys (an upper bound is fine) -> layout, allocations
static int build_begin_impl(bbduk_handle* h, double maxKeys, int hdist, int hdist2) {
    if (h->finalized) return fail(h, BBDUK_ERR_STATE, "table already finalized");
    if (h->build) return fail(h, BBDUK_ERR_STATE, "a table build is already in progress");
    HIP_TRY(h, hipSetDevice(h->p.device));
    BuildState* st = new (std::nothrow) BuildState();
    if (!st) return BBDUK_ERR_NOMEM;
    h->build = st; st->hdist = hdist; st->hdist2 = hdist2;
    // reference-side Hamming neighbourhoods put ~2/3 of a k-mer's 1+3k variants on one minimizer: such maps take plain lines
    if (hdist > 0) h->bigPlain = true;
    st->big = (maxKeys > (double)BIG_LAYOUT_MIN_KEYS || h->hookBigLayout) && big_layout_eligible(h->p) && !h->sealTable && big_geometry(h);
    auto bail = [&](int code, const char* msg) { build_release(h); table_release(h); return fail(h, code, msg); };
    if (hipMalloc(&st->d_cnt, 32) != hipSuccess || hipMemsetAsync(st->d_cnt, 0, 32, h->stream) != hipSuccess) return bail(BBDUK_ERR_NOMEM, "hipMalloc");
    if (st->big) {
        // 32-slot lines at ~0.6 keys per slot (the lines' loads vary with the minimizers: 10 % of them overflow into the next line
        // there, 3 % of the keys); a tighter fit is tried when HBM is short.  12 or 14 bytes per slot: 10^10 keys = 200-233 GB.
        const int idBytes = h->p.numScaffolds <= 65535 ? 2 : 4;
        size_t freeB = 0, totalB = 0;
        if (hipMemGetInfo(&freeB, &totalB) != hipSuccess) return bail(BBDUK_ERR_DEVICE, "hipMemGetInfo");
        // the secondary map holds what the lines spill (both words of a key full): ~6 % of the keys at 0.6 keys per slot; its
        // buckets are sized for twice that at 2.5 keys per bucket (it keeps working, with longer chains, until it is full)
        int sbits = 10;
        while (sbits < 29 && (double)(1ULL << sbits) < 0.12 * maxKeys / 2.5) sbits++;
        const uint64_t snb = 1ULL << sbits;
        const double perLine = 64.0 + 256.0 + 32.0 * idBytes, spillBytes = (double)snb * (8.0 + 64.0);
        uint64_t nlines = 0;
        for (const double load : {0.6, 0.7, 0.8}) {
            nlines = std::max<uint64_t>(64, (uint64_t)(maxKeys / (32.0 * load)) + 1);
            if ((double)nlines * perLine + spillBytes + 3e9 < (double)freeB) break;
            nlines = 0;
        }
        if (!nlines) return bail(BBDUK_ERR_NOMEM, "the map does not fit this device's memory");
        if (nlines >= (1ULL << 29)) return bail(BBDUK_ERR_ARG, "too many keys for the 32-bit tag word index");
        if (hipMalloc(&h->d_bigTags, nlines * 64) != hipSuccess || hipMalloc(&h->d_bigKeys, nlines * 256) != hipSuccess ||
            hipMalloc(&h->d_bigIds, nlines * 32 * (size_t)idBytes) != hipSuccess ||
            hipMalloc(&h->d_tags, snb * 8) != hipSuccess || hipMalloc(&h->d_bkv, 4 * snb * sizeof(uint4)) != hipSuccess) return bail(BBDUK_ERR_NOMEM, "hipMalloc (map)");
        h->big = true; h->bigLines = (uint32_t)nlines; h->bigIdBytes = idBytes; h->nbuckets = snb; h->bucketBits = sbits;
        hipMemsetAsync(h->d_bigTags, 0, nlines * 64, h->stream);
        hipMemsetAsync(h->d_bigKeys, 0xFF, nlines * 256, h->stream);
        hipMemsetAsync(h->d_bigIds, 0xFF, nlines * 32 * (size_t)idBytes, h->stream);
        hipMemsetAsync(h->d_tags, 0, snb * 8, h->stream);
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
    B.rcomp = h->p.rcomp; B.middleMask = (uint64_t)h->p.middleMask; B.totalBases = total; B.nrefs = npieces;
    const int V1 = (st->hdist > 0 || (B.useShort && st->hdist2 > 0)) ? 1 + 3 * B.k : 1;
    const int64_t work = total * (int64_t)V1;
    const int grid = (int)std::min<int64_t>((work + 255) / 256, (int64_t)h->numCU * 32);
    bbduk_build_enum_kernel<<<dim3(std::max(grid, 1)), dim3(256), 0, h->stream>>>(B, d_refs, st->d_roff, st->d_rid, st->d_rfl, V1, make_sink(h, st));
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipStreamSynchronize(h->stream));                 // the caller may reuse d_refs and the host arrays
    return BBDUK_OK;
}

static int build_end_impl(bbduk_handle* h) {
    BuildState* st = h->build;
    auto bail = [&](int code, const char* msg) { build_release(h); table_release(h); return fail(h, code, msg); };
    unsigned long long cnt[3] = {0, 0, 0};
    if (hipMemcpyAsync(cnt, st->d_cnt, 24, hipMemcpyDeviceToHost, h->stream) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess)
        return bail(BBDUK_ERR_DEVICE, "table build failed on the device");
    const unsigned long long distinct = cnt[0];
    if (st->big) {
        if (cnt[1]) return bail(BBDUK_ERR_NOMEM, "the map overflowed: more keys than announced to bbduk_build_begin");
        h->nkeys = (int64_t)distinct; h->ldsBits = 0; h->nspilled = (int64_t)cnt[2];
    } else {
        // 4-way buckets of 15-bit fingerprints, >= 1 bucket per key (load 0.5-1 keys/bucket: ~0.1-0.4 % of buckets
        // overflow and carry the continuation flag, so almost every lookup ends in its home bucket).
        int bbits = 10;
        while (bbits < 32 && (1ULL << bbits) < distinct) bbits++;
        if (bbits == 30 && distinct <= (1ULL << 30)) bbits = 29;     // the slot index is 31 bits: the largest maps of this layout run at up to 2 keys per bucket
        if (h->hookBucketBits >= 4 && h->hookBucketBits <= 32) bbits = h->hookBucketBits;      // bbduk_test_hook
        const uint64_t nb = 1ULL << bbits;
        if (4 * nb < distinct + nb / 8 || 4 * nb > (1ULL << 31)) return bail(BBDUK_ERR_ARG, "too many keys for the bucket index");
        // Presence filter in front of the map.  Most query k-mers are absent, so one bit per hash slot held in LDS
        // (<=128 KiB per workgroup) answers most of them without leaving the CU.  Size follows the key count.
        auto ceil_log2 = [](uint64_t x) { int b = 0; while ((1ULL << b) < x) b++; return b; };
        int lb = 0;
        if (distinct > 0 && distinct <= (1ULL << 22)) lb = std::min(MAX_LDS_BITS, std::max(10, ceil_log2(32ULL * distinct)));
        if (h->hookLdsBits >= 0) lb = h->hookLdsBits == 0 ? 0 : std::min(MAX_LDS_BITS, std::max(10, h->hookLdsBits));   // bbduk_test_hook
        if (hipMalloc(&h->d_tags, nb * sizeof(uint64_t)) != hipSuccess || hipMalloc(&h->d_bkv, 4 * nb * sizeof(uint4)) != hipSuccess ||
            (lb && hipMalloc(&h->d_ldsImage, ((size_t)1 << (lb - 5)) * 4) != hipSuccess)) return bail(BBDUK_ERR_NOMEM, "hipMalloc (map)");
        hipMemsetAsync(h->d_tags, 0, nb * sizeof(uint64_t), h->stream);
        hipMemsetAsync(h->d_bkv, 0xFF, 4 * nb * sizeof(uint4), h->stream);
        if (lb) hipMemsetAsync(h->d_ldsImage, 0, ((size_t)1 << (lb - 5)) * 4, h->stream);
        const int grid = (int)std::min<uint64_t>((st->cslots + 255) / 256, (uint64_t)h->numCU * 32);
        bbduk_build_place_kernel<<<dim3(grid), dim3(256), 0, h->stream>>>(st->d_sk, st->d_si, st->cslots, h->d_tags, h->d_bkv, bbits, (uint32_t)(nb - 1), h->d_ldsImage, lb);
        if (hipStreamSynchronize(h->stream) != hipSuccess) return bail(BBDUK_ERR_DEVICE, "device build (placement) failed");
        h->nbuckets = nb; h->bucketBits = bbits; h->nkeys = (int64_t)distinct; h->ldsBits = lb;
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
    return build_begin_impl(h, (double)max_keys, hdist, hdist2);
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
    return build_end_impl(h);
}

// The pairs a host staged with bbduk_upload_pairs / bbduk_upload_table_way go to the device in chunks and are placed there (one
// thread per pair; 10^8 keys took 17.8 s in a serial host loop, they take about a second this way).
extern "C" int bbduk_finalize_table(bbduk_handle* h) {
    if (!h) return BBDUK_ERR_ARG;
    std::lock_guard<std::mutex> g(h->mu);
    if (h->finalized) return fail(h, BBDUK_ERR_STATE, "table already finalized");
    if (h->build) return fail(h, BBDUK_ERR_STATE, "a device-side build is in progress: end it with bbduk_build_end");
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
    rc = build_end_impl(h);
    if (rc == BBDUK_ERR_NOMEM && !h->bigPlain && !h->finalized) { h->bigPlain = true; continue; }
    break;
  }
    if (rc == BBDUK_OK) { h->hkeys.clear(); h->hkeys.shrink_to_fit(); h->hvals.clear(); h->hvals.shrink_to_fit(); }
    return rc;
}

// bbduk_build_table_device: the reference sequences are HOST memory here; they go to the device in chunks of whole scaffolds
// (a scaffold longer than a chunk as pieces that overlap by k-1 bases) through bbduk_build_begin / build_add_pieces / bbduk_build_end.
extern "C" int bbduk_build_table_device(bbduk_handle* h, const uint8_t* refs, const int64_t* ref_offsets, int32_t n_refs,
                                        int32_t hdist, int32_t hdist2) {
    if (!h) return BBDUK_ERR_ARG;
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
    auto variants = [](int len, int d) { const double t = 3.0 * len; double v = 1.0; if (d >= 1) v += t; if (d >= 2) v += t * (t - 3.0) / 2.0; if (d >= 3) v += t * (t - 3.0) * (t - 6.0) / 6.0; return v; };
    double ub = (double)total * variants(k, hdist);
    if (useShort) for (int L = h->p.mink; L < k; L++) ub += 2.0 * (double)n_refs * variants(L, hdist2);
    int rc = BBDUK_OK;
  for (int attempt = 0; attempt < 2; attempt++) {                  // second attempt: plain lines, if the minimizer lines spilled too much
    rc = build_begin_impl(h, ub, hdist, hdist2);
    if (rc != BBDUK_OK) return rc;
    BuildState* st = h->build;
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
            pos += len - (k - 1);
        } while (true);
    }
    if ((rc = flush()) != BBDUK_OK) return bail(rc, "device build (enumeration)");
    rc = build_end_impl(h);
    if (rc == BBDUK_ERR_NOMEM && !h->bigPlain && !h->finalized) { h->bigPlain = true; continue; }
    break;
  }
    return rc;
}

// diagnostics of the big layout (include/bbduk_test_hooks.h): how many lines hold 0..32 keys
__global__ void bbduk_line_hist_kernel(const uint64_t* __restrict__ keys, const uint32_t nlines, unsigned long long* __restrict__ hist) {
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
    bbduk_line_hist_kernel<<<dim3(h->numCU * 16), dim3(256), 0, h->stream>>>(h->d_bigKeys, h->bigLines, d);
    hipMemcpyAsync(out33, d, 33 * 8, hipMemcpyDeviceToHost, h->stream);
    const hipError_t e = hipStreamSynchronize(h->stream);
    hipFree(d);
    return e == hipSuccess ? BBDUK_OK : BBDUK_ERR_DEVICE;
}
extern "C" int64_t bbduk_table_spilled(const bbduk_handle* h) { return (h && h->finalized && h->big) ? h->nspilled : 0; }
extern "C" int64_t bbduk_table_size(const bbduk_handle* h) { return (h && h->finalized) ? h->nkeys : -1; }
extern "C" int64_t bbduk_table_bytes(const bbduk_handle* h) {
    if (!h || !h->finalized) return -1;
    if (h->big) return (int64_t)h->bigLines * (64 + 256 + 32 * h->bigIdBytes) + (int64_t)(h->nbuckets * (8 + 4 * 16));
    return (int64_t)(h->nbuckets * (8 + 4 * 16)) + (h->ldsBits ? (1LL << (h->ldsBits - 3)) : 0);
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
    K.gm = h->gm; K.gW = h->gW; K.gH = h->gH; K.gD = h->gD;
    K.matchN = nullptr; K.matchIds = nullptr; K.matchCnt = nullptr; K.matchCap = 0;
    K.dbg = h->hookDbg;
    K.ldsImage = h->d_ldsImage; K.ldsBits = h->ldsBits;
    return K;
}

struct MatchOut { int32_t* n; int32_t* ids; int32_t* counts; int32_t cap; };     // device buffers of bbduk_kfilter_batch_matches*

// kbig / findBestMatch (through the kfilter operators) and ksplit: bbduk_kscan_kernel
static int launch_kscan(bbduk_handle* h, const uint8_t* d_bases, const int64_t* d_offsets, int64_t n, int64_t total_bases, int32_t paired,
                        int32_t* d_a, int32_t* d_id, uint8_t* d_fl, int32_t* d_left, int32_t* d_right, int64_t* d_counters, hipStream_t st,
                        const uint32_t* d_undef, bool packed, const MatchOut* mo = nullptr) {
    KParams K = make_kparams(h);
    K.undef = packed ? d_undef : nullptr;
    if (mo) { K.matchN = mo->n; K.matchIds = mo->ids; K.matchCnt = mo->counts; K.matchCap = mo->cap; }
    const int red = h->p.mode == BBDUK_MODE_KSPLIT ? RED_SPLIT : (K.fbm ? RED_BEST : RED_BIG);
    typedef void (*kscan_t)(const KParams, const uint8_t*, const int64_t*, const int64_t, const int64_t, const int,
                            int32_t*, int32_t*, uint8_t*, int32_t*, int32_t*, int64_t*);
    typedef void (*kscan_full_t)(const KParams, const uint8_t*, const int64_t*, const int64_t, const int64_t, const int,
                                 int32_t*, int32_t*, uint8_t*, int32_t*, int32_t*, int64_t*, const int*);
    const kscan_full_t fn = red == RED_SPLIT ? bbduk_kscan_kernel<RED_SPLIT> : (red == RED_BEST ? bbduk_kscan_kernel<RED_BEST> : bbduk_kscan_kernel<RED_BIG>);
    const kscan_full_t lfn = red == RED_SPLIT ? bbduk_kscan_long_kernel<RED_SPLIT> : (red == RED_BEST ? bbduk_kscan_long_kernel<RED_BEST> : bbduk_kscan_long_kernel<RED_BIG>);
    const size_t dynLds = h->ldsBits ? ((size_t)1 << (h->ldsBits - 3)) : 0;
    HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dynLds));
    const int64_t ntiles = (n + TILE_READS - 1) / TILE_READS;
    const int grid = (int)std::min<int64_t>(ntiles, (int64_t)h->numCU);
    std::lock_guard<std::mutex> lg(h->launchMu);
    const int evi = (int)(h->evCount % bbduk_handle::EV_RING);
    if (!h->ev0[evi]) { HIP_TRY(h, hipEventCreate(&h->ev0[evi])); HIP_TRY(h, hipEventCreate(&h->ev1[evi])); }
    HIP_TRY(h, hipEventRecord(h->ev0[evi], st));
    int* const d_flag = h->d_slowFlag + 4 * evi;      // [0] pre-pass bits, [1] reads with a tail, [2] short reads (launch_batch)
    HIP_TRY(h, hipMemsetAsync(d_flag, 0, 4 * sizeof(int), st));
    if (query_expansion(h->p)) { const int one = 1; HIP_TRY(h, hipMemcpyAsync(d_flag, &one, sizeof(int), hipMemcpyHostToDevice, st)); }   // the tiled kernels expand
    {   // pre-pass per READ: one beyond the tiled kernel's planes sends the batch to bbduk_kscan_long_kernel; ksplit: one beyond a
        // wave's planes (bit 0) sends it to the tiled kernel, else bbduk_wave_kernel<KSPLIT> takes it
        const int sgrid = (int)std::min<int64_t>((n + 255) / 256, (int64_t)h->numCU * 8);
        bbduk_span_kernel<<<dim3(std::max(sgrid, 1)), dim3(256), 0, st>>>(d_offsets, n, 0, d_flag, red == RED_SPLIT && !packed ? (int64_t)WUNIT_MAX : (int64_t)(KM_CAP_BASES - 32), (int64_t)(KM_CAP_BASES - 32));
        if (red != RED_SPLIT) {                                     // findBestMatch, k > 31: a unit (pair) beyond a wave's planes (bit 0) -> the tiled kernel
            const int64_t units = paired ? n / 2 : n;
            const int ugrid = (int)std::min<int64_t>((units + 255) / 256, (int64_t)h->numCU * 8);
            bbduk_span_kernel<<<dim3(std::max(ugrid, 1)), dim3(256), 0, st>>>(d_offsets, n, (int)paired, d_flag, (int64_t)WUNIT_MAX, (int64_t)0x7FFFFFFFFFFFLL);
        }
    }
    if (red != RED_SPLIT) {                                         // the main kernel's shape; the pair scan keeps an id list (main_scan_pair_best) or the run state
        K.waveFirst = 1;                                            // (main_scan_pair_kbig) per read
        const bool general = params_general(h->p);
        const batch_kernel_t wkBest = general ? bbduk_wave_kernel<BBDUK_MODE_FBM, false, true, true, 2>
                                : (packed ? (K.forbidNs ? bbduk_wave_kernel<BBDUK_MODE_FBM, false, true, false, 1> : bbduk_wave_kernel<BBDUK_MODE_FBM, false, false, false, 1>)
                                          : (K.forbidNs ? bbduk_wave_kernel<BBDUK_MODE_FBM, false, true, false, 0> : bbduk_wave_kernel<BBDUK_MODE_FBM, false, false, false, 0>));
        const batch_kernel_t wkBig = general ? bbduk_wave_kernel<BBDUK_MODE_KBIG, false, true, true, 2>
                                : (packed ? (K.forbidNs ? bbduk_wave_kernel<BBDUK_MODE_KBIG, false, true, false, 1> : bbduk_wave_kernel<BBDUK_MODE_KBIG, false, false, false, 1>)
                                          : (K.forbidNs ? bbduk_wave_kernel<BBDUK_MODE_KBIG, false, true, false, 0> : bbduk_wave_kernel<BBDUK_MODE_KBIG, false, false, false, 0>));
        const batch_kernel_t wk = red == RED_BEST ? wkBest : wkBig;
        const size_t waveLds = dynLds + WAVE_LDS_BYTES;
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(wk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)waveLds));
        const int64_t nmt = (n + MT_READS - 1) / MT_READS;
        const int wgrid = (int)std::min<int64_t>((nmt + NWAVES - 1) / NWAVES, (int64_t)h->numCU);
        wk<<<dim3(std::max(wgrid, 1)), dim3(BLOCK_THREADS), waveLds, st>>>(K, d_bases, d_offsets, n, total_bases, (int)paired, d_a, d_id, d_fl, d_counters, d_flag);
    }
    if (red == RED_SPLIT && !packed) {                              // the main kernel's shape: wave-autonomous mini-tiles, one lane per read in the finish
        K.waveFirst = 1; K.outLeft = d_left; K.outRight = d_right;
        const bool general = params_general(h->p);
        const batch_kernel_t wk = general ? bbduk_wave_kernel<BBDUK_MODE_KSPLIT, true, true, true, 2>
                                : (K.forbidNs ? bbduk_wave_kernel<BBDUK_MODE_KSPLIT, true, true, false, 0> : bbduk_wave_kernel<BBDUK_MODE_KSPLIT, true, false, false, 0>);
        const size_t waveLds = dynLds + WAVE_LDS_BYTES;
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(wk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)waveLds));
        const int64_t nmt = (n + MT_READS - 1) / MT_READS;
        const int wgrid = (int)std::min<int64_t>((nmt + NWAVES - 1) / NWAVES, (int64_t)h->numCU);
        wk<<<dim3(std::max(wgrid, 1)), dim3(BLOCK_THREADS), waveLds, st>>>(K, d_bases, d_offsets, n, total_bases, 0, d_a, d_id, d_fl, d_counters, d_flag);
    }
    fn<<<dim3(grid), dim3(BLOCK_THREADS), dynLds, st>>>(K, d_bases, d_offsets, n, total_bases, (int)paired, d_a, d_id, d_fl, d_left, d_right, d_counters, d_flag);
    HIP_TRY(h, hipEventRecord(h->ev1[evi], st));
    h->evCount++;
    HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(lfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dynLds));
    const int64_t lunits = (paired && red != RED_SPLIT) ? n / 2 : n;
    const int lgrid = (int)std::min<int64_t>((lunits + NWAVES - 1) / NWAVES, (int64_t)h->numCU);
    lfn<<<dim3(std::max(lgrid, 1)), dim3(BLOCK_THREADS), dynLds, st>>>(K, d_bases, d_offsets, n, total_bases, (int)paired, d_a, d_id, d_fl, d_left, d_right, d_counters, d_flag);
    HIP_TRY(h, hipGetLastError());
    return BBDUK_OK;
}

static int launch_batch(bbduk_handle* h, int wantKfilter, const uint8_t* d_bases, const int64_t* d_offsets, int64_t n,
                        int64_t total_bases, int32_t paired, int32_t* d_a, int32_t* d_id, uint8_t* d_fl,
                        int64_t* d_counters, hipStream_t st, const uint32_t* d_undef = nullptr, bool packed = false, const MatchOut* mo = nullptr) {
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
        return launch_kscan(h, d_bases, d_offsets, n, total_bases, paired, d_a, d_id, d_fl, nullptr, nullptr, d_counters, st, d_undef, packed, mo);
    KParams K = make_kparams(h);
    K.undef = packed ? d_undef : nullptr;
    const int64_t ntiles = (n + TILE_READS - 1) / TILE_READS;
    const size_t dynLds = h->ldsBits ? ((size_t)1 << (h->ldsBits - 3)) : 0;    // tile kernel: the filter only
    const size_t waveLds = dynLds + WAVE_LDS_BYTES;                             // wave kernel: filter + its per-wave state
    KernelPair kp = pick_kernel(K);
    if (packed) { kp.wave = kp.wavePacked; kp.shape = kp.shapePacked; }
    // the tail pass and the three-read blocks belong to the candidate form of the scans (wave_body's candMode): elsewhere the pre-pass counts
    // nothing and bbduk_wave_shape_kernel is not launched
    const bool tailForm = kp.shape && !K.big && K.qhdist == 0 && K.qskip < 2 &&
                          (K.mode == BBDUK_MODE_KTRIM_R || (K.mode == BBDUK_MODE_KFILTER && K.maxBadKmers == 0 && K.mkf == 0.f && K.mcf == 0.f));
    HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(kp.wave), hipFuncAttributeMaxDynamicSharedMemorySize, (int)waveLds));
    HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(kp.tile), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dynLds));
    if (tailForm) HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(kp.shape), hipFuncAttributeMaxDynamicSharedMemorySize, (int)waveLds));
    // pre-pass: if some pair is longer than a wave's planes the tile kernel takes the whole batch, else the wave kernel
    std::lock_guard<std::mutex> lg(h->launchMu);
    const int evi = (int)(h->evCount % bbduk_handle::EV_RING);
    int* const d_flag = h->d_slowFlag + 4 * evi;      // [0] pre-pass bits, [1] reads with a tail, [2] short reads (launch_batch)
    HIP_TRY(h, hipMemsetAsync(d_flag, 0, 4 * sizeof(int), st));
    if (query_expansion(h->p)) { const int one = 1; HIP_TRY(h, hipMemcpyAsync(d_flag, &one, sizeof(int), hipMemcpyHostToDevice, st)); }   // the tiled kernels expand
    if (h->hookForceTile) { const int one = 1; HIP_TRY(h, hipMemcpyAsync(d_flag, &one, sizeof(int), hipMemcpyHostToDevice, st)); }
    {
        const int64_t units = paired ? n / 2 : n;
        const int sgrid = (int)std::min<int64_t>((units + 255) / 256, (int64_t)h->numCU * 8);
        bbduk_span_kernel<<<dim3(std::max(sgrid, 1)), dim3(256), 0, st>>>(d_offsets, n, (int)paired, d_flag, (int64_t)WUNIT_MAX, (int64_t)(CAP_BASES - 64), tailForm ? K.k - 1 : -1);
        if (tailForm) bbduk_shape_kernel<<<dim3(1), dim3(1), 0, st>>>(d_flag, n);
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
    kp.wave<<<dim3(wgrid), dim3(BLOCK_THREADS), waveLds, st>>>(K, d_bases, d_offsets, n, total_bases, (int)paired, d_a, d_id, d_fl, d_counters, d_flag);
    if (tailForm) kp.shape<<<dim3(wgrid), dim3(BLOCK_THREADS), waveLds, st>>>(K, d_bases, d_offsets, n, total_bases, (int)paired, d_a, d_id, d_fl, d_counters, d_flag);
    kp.tile<<<dim3(tgrid), dim3(BLOCK_THREADS), dynLds, st>>>(K, d_bases, d_offsets, n, total_bases, (int)paired, d_a, d_id, d_fl, d_counters, d_flag);
    HIP_TRY(h, hipEventRecord(h->ev1[evi], st));                    // around both: whichever of the two took the batch (the other returns at once)
    h->evCount++;
    {   // reads beyond BBDUK_MAX_READ_LEN: chunked scan, one wave per unit (returns at once unless the pre-pass asked for it)
        const batch_kernel_t lk = K.mode == BBDUK_MODE_KFILTER ? (K.big ? bbduk_long_kernel<BBDUK_MODE_KFILTER, true> : bbduk_long_kernel<BBDUK_MODE_KFILTER>) :
                                  (K.mode == BBDUK_MODE_KTRIM_L ? bbduk_long_kernel<BBDUK_MODE_KTRIM_L> : bbduk_long_kernel<BBDUK_MODE_KTRIM_R>);
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(lk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dynLds));
        const int64_t units = paired ? n / 2 : n;
        const int lgrid = (int)std::min<int64_t>((units + NWAVES - 1) / NWAVES, (int64_t)h->numCU);
        lk<<<dim3(std::max(lgrid, 1)), dim3(BLOCK_THREADS), dynLds, st>>>(K, d_bases, d_offsets, n, total_bases, (int)paired, d_a, d_id, d_fl, d_counters, d_flag);
    }
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
                                reinterpret_cast<const uint32_t*>(S->d_undef), packed, hostMatches ? &dm : nullptr);
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
    HIP_TRY(h, hipMemcpyAsync(&status, h->d_counters + BBDUK_CTR_STATUS, sizeof status, hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipStreamSynchronize(st));
    if (status != 0) {
        int64_t z = 0;
        hipMemcpyAsync(h->d_counters + BBDUK_CTR_STATUS, &z, sizeof z, hipMemcpyHostToDevice, st);
        hipStreamSynchronize(st);
        return fail(h, -(int)status, "device reported an error (a read longer than BBDUK_MAX_READ_LEN, or trimfailuresto1bp on a unit beyond the main kernel's planes)");
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
    HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(bbduk_ktrimtips_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dynLds));
    const int64_t ntiles = (n + TILE_READS - 1) / TILE_READS;
    const int grid = (int)std::min<int64_t>(ntiles, (int64_t)h->numCU);
    std::lock_guard<std::mutex> lg(h->launchMu);
    const int evi = (int)(h->evCount % bbduk_handle::EV_RING);
    int* const d_flag = h->d_slowFlag + 4 * evi;      // [0] pre-pass bits, [1] reads with a tail, [2] short reads (launch_batch)
    HIP_TRY(h, hipMemsetAsync(d_flag, 0, 4 * sizeof(int), st));
    if (query_expansion(h->p)) { const int one = 1; HIP_TRY(h, hipMemcpyAsync(d_flag, &one, sizeof(int), hipMemcpyHostToDevice, st)); }   // the tiled kernels expand
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
        const batch_kernel_t wk = general ? bbduk_wave_kernel<BBDUK_MODE_KTRIM_TIPS, true, true, true, 2>
                                : (packed ? (K.forbidNs ? bbduk_wave_kernel<BBDUK_MODE_KTRIM_TIPS, true, true, false, 1> : bbduk_wave_kernel<BBDUK_MODE_KTRIM_TIPS, true, false, false, 1>)
                                          : (K.forbidNs ? bbduk_wave_kernel<BBDUK_MODE_KTRIM_TIPS, true, true, false, 0> : bbduk_wave_kernel<BBDUK_MODE_KTRIM_TIPS, true, false, false, 0>));
        const size_t waveLds = dynLds + WAVE_LDS_BYTES;
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(wk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)waveLds));
        const int64_t nmt = (n + MT_READS - 1) / MT_READS;
        const int wgrid = (int)std::min<int64_t>((nmt + NWAVES - 1) / NWAVES, (int64_t)h->numCU);
        wk<<<dim3(std::max(wgrid, 1)), dim3(BLOCK_THREADS), waveLds, st>>>(K, d_bases, d_offsets, n, total_bases, (int)paired, d_r, d_id, d_fl, d_counters, d_flag);
    }
    bbduk_ktrimtips_kernel<<<dim3(grid), dim3(BLOCK_THREADS), dynLds, st>>>(K, d_bases, d_offsets, n, total_bases, (int)paired, d_r, d_l, d_id, d_fl, d_counters, d_flag);
    HIP_TRY(h, hipEventRecord(h->ev1[evi], st));
    h->evCount++;
    HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(bbduk_long_tips_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dynLds));
    const int64_t units = paired ? n / 2 : n;
    const int lgrid = (int)std::min<int64_t>((units + NWAVES - 1) / NWAVES, (int64_t)h->numCU);
    bbduk_long_tips_kernel<<<dim3(std::max(lgrid, 1)), dim3(BLOCK_THREADS), dynLds, st>>>(K, d_bases, d_offsets, n, total_bases, (int)paired, d_r, d_l, d_id, d_fl, d_counters, d_flag);
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
        return fail(h, -(int)status, "device reported an error (a read longer than BBDUK_MAX_READ_LEN, or trimfailuresto1bp on a unit beyond the main kernel's planes)");
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
    HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(bbduk_kmask_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dynLds));
    HIP_TRY(h, hipMemsetAsync(d_mask, 0, ((size_t)(total_bases + 31) / 32 + 2) * sizeof(uint32_t), st));
    const int64_t ntiles = (n + TILE_READS - 1) / TILE_READS;
    const int grid = (int)std::min<int64_t>(ntiles, (int64_t)h->numCU);
    std::lock_guard<std::mutex> lg(h->launchMu);
    const int evi = (int)(h->evCount % bbduk_handle::EV_RING);
    int* const d_flag = h->d_slowFlag + 4 * evi;      // [0] pre-pass bits, [1] reads with a tail, [2] short reads (launch_batch)
    HIP_TRY(h, hipMemsetAsync(d_flag, 0, 4 * sizeof(int), st));
    if (query_expansion(h->p)) { const int one = 1; HIP_TRY(h, hipMemcpyAsync(d_flag, &one, sizeof(int), hipMemcpyHostToDevice, st)); }   // the tiled kernels expand
    if (!h->ev0[evi]) { HIP_TRY(h, hipEventCreate(&h->ev0[evi])); HIP_TRY(h, hipEventCreate(&h->ev1[evi])); }
    {   // pre-pass: a unit (pair) beyond a wave's planes sends the batch to the tiled kernel (which in turn leaves the reads beyond ITS planes
        // to bbduk_kmask_long_kernel); else bbduk_wave_kernel<KMASK> takes it
        const int64_t units = paired ? n / 2 : n;
        const int ugrid = (int)std::min<int64_t>((units + 255) / 256, (int64_t)h->numCU * 8);
        bbduk_span_kernel<<<dim3(std::max(ugrid, 1)), dim3(256), 0, st>>>(d_offsets, n, (int)paired, d_flag, (int64_t)WUNIT_MAX_KM, (int64_t)0x7FFFFFFFFFFFLL);
    }
    HIP_TRY(h, hipEventRecord(h->ev0[evi], st));
    {   // the main kernel's shape: wave-autonomous mini-tiles, a fourth plane for the hit positions, one lane per read in the finish
        K.waveFirst = 1; K.outMask = d_mask;
        const bool general = params_general(h->p);
        const batch_kernel_t wk = general ? bbduk_wave_kernel<BBDUK_MODE_KMASK, true, true, true, 2>
                                : (packed ? (K.forbidNs ? bbduk_wave_kernel<BBDUK_MODE_KMASK, true, true, false, 1> : bbduk_wave_kernel<BBDUK_MODE_KMASK, true, false, false, 1>)
                                          : (K.forbidNs ? bbduk_wave_kernel<BBDUK_MODE_KMASK, true, true, false, 0> : bbduk_wave_kernel<BBDUK_MODE_KMASK, true, false, false, 0>));
        const size_t waveLds = dynLds + WAVE_LDS_BYTES_KM;
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(wk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)waveLds));
        const int64_t nmt = (n + MT_READS - 1) / MT_READS;
        const int wgrid = (int)std::min<int64_t>((nmt + NWAVES - 1) / NWAVES, (int64_t)h->numCU);
        wk<<<dim3(std::max(wgrid, 1)), dim3(BLOCK_THREADS), waveLds, st>>>(K, d_bases, d_offsets, n, total_bases, (int)paired, d_a, d_id, d_fl, d_counters, d_flag);
    }
    bbduk_kmask_kernel<<<dim3(grid), dim3(BLOCK_THREADS), dynLds, st>>>(K, d_bases, d_offsets, n, total_bases, (int)paired, d_a, d_id, d_fl, d_mask, d_counters, d_flag);
    HIP_TRY(h, hipEventRecord(h->ev1[evi], st));
    h->evCount++;
    HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(bbduk_kmask_long_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dynLds));
    const int lgrid = (int)std::min<int64_t>((n + NWAVES - 1) / NWAVES, (int64_t)h->numCU);     // sequences beyond the tiled kernel's planes
    bbduk_kmask_long_kernel<<<dim3(std::max(lgrid, 1)), dim3(BLOCK_THREADS), dynLds, st>>>(K, d_bases, d_offsets, n, total_bases, (int)paired, d_a, d_id, d_mask, d_counters, d_flag);
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
        return fail(h, -(int)status, "device reported an error (a read longer than BBDUK_MAX_READ_LEN, or trimfailuresto1bp on a unit beyond the main kernel's planes)");
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
