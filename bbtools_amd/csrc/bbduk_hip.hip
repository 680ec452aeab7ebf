// bbduk_hip.hip -- MI355X (gfx950) kernels + the C ABI of include/bbduk_gpu.h.
//
// Hot path: per-read 2-bit k-mer encode with reverse-complement
// canonicalisation, optional query-side Hamming expansion, open-addressed lookup into an HBM-resident
// image of the reference k-mer map, first-hit / hit-count reduction with wave ballots, trim / filter
// decision, pair logic and counters.  Integer work only (no MFMA): the bounds are the HBM read stream of
// the bases and the gather rate into the table.
//
// Design (not a translation of the Java loops):
//   * a tile of reads is contiguous in the concatenated `bases` buffer, so it is staged with 16-byte
//     coalesced loads and converted on the fly to three bit-planes in LDS (2-bit forward codes in
//     *reversed* base order, 2-bit complement codes, 1-bit undefined mask);
//   * the scan is position-parallel: one wave64 lane per k-mer end position; a lane cuts its k-mer and
//     its reverse complement out of the planes with two funnel shifts each (closed form SURVEY A.12)
//     instead of rolling them along the read;
//   * hits are reduced with __ballot / ffs / popcount; everything per read is wave-uniform scalar work;
//   * results are staged in LDS and written back coalesced; counters are reduced per block.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <string>
#include <vector>
#include <algorithm>
#include <mutex>
#include "../../include/bbduk_gpu.h"
#include "synth.h"

static_assert(sizeof(bbduk_params) == 128, "bbduk_params layout is part of the ABI");
static_assert(sizeof(bbduk_synth_params) == 80, "bbduk_synth_params layout is part of the ABI");

#define BLOCK_THREADS   256
#define NWAVES          (BLOCK_THREADS / 64)
#define TILE_READS      128                    // reads per tile (even: whole pairs)
#define CAP_BASES       33792                  // LDS plane capacity in bases (>= 2*BBDUK_MAX_READ_LEN + 32: a pair fits)
#define CAP_CHUNKS      (CAP_BASES / 16)
#define EMPTY_KEY       0xFFFFFFFFFFFFFFFFULL  // keys are < 2^63
#define HASH_MULT       0x9E3779B97F4A7C15ULL
#define BIGLOC          999999999

struct KParams {
    int32_t mode, k, mink, rcomp, forbidNs, minlen, minlen2, qhdist, qhdist2, maxBadKmers, minReadLength;
    float   minLenFraction;
    int32_t rieb, trimPad, ktrimExclusive, restrictLeft, restrictRight, skipR1, skipR2, numScaffolds, useShort;
    uint64_t mask, kmask, middleMask;
    const uint64_t* tkeys;      // open-addressed, power-of-two capacity, EMPTY_KEY = free
    const int32_t*  tvals;
    uint64_t capMask;
    int32_t  hashShift;
    int64_t  storedKmers;
};

// --------------------------------------------------------------------------------------------------
// device helpers

// reverseComplementBinaryFast(long,int) (dna/AminoAcid.java:585-601): complement, reverse 2-bit groups, right-align
__device__ __forceinline__ uint64_t dev_rcomp(uint64_t kmer, int len) {
    uint64_t x = __brevll(~kmer);
    x = ((x >> 1) & 0x5555555555555555ULL) | ((x & 0x5555555555555555ULL) << 1);
    return x >> (64 - 2 * len);
}

__device__ __forceinline__ int table_get(const KParams& P, uint64_t key) {
    uint64_t s = (key * HASH_MULT) >> P.hashShift;
    for (;;) {
        const uint64_t kk = P.tkeys[s];
        if (kk == key) return P.tvals[s];
        if (kk == EMPTY_KEY) return -1;
        s = (s + 1) & P.capMask;
    }
}

// getValueInner (bbduk/BBDukIndexMod.java:492-520): canonicalise, mask middle, add length bit, probe
__device__ __forceinline__ int get_value_inner(const KParams& P, uint64_t kmer, uint64_t rkmer, uint64_t lengthMask) {
    const uint64_t mx = P.rcomp ? (kmer > rkmer ? kmer : rkmer) : kmer;   // values < 2^62: unsigned max == Java signed max
    return table_get(P, (mx & P.middleMask) | lengthMask);
}

// getValue (bbduk/BBDukIndexMod.java:462-481): query-side Hamming expansion, same (j,i) order, first id>=1 wins
template <int D>
__device__ int get_value(const KParams& P, uint64_t kmer, uint64_t rkmer, uint64_t lengthMask, int len, int qh) {
    int id = get_value_inner(P, kmer, rkmer, lengthMask);
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
__device__ __forceinline__ int lookup(const KParams& P, uint64_t kmer, uint64_t rkmer, uint64_t lengthMask, int len, int qh) {
    if (qh <= 0) return get_value_inner(P, kmer, rkmer, lengthMask);
    return get_value<2>(P, kmer, rkmer, lengthMask, len, qh);
}

// nb (1..31) 2-bit symbols starting at symbol index `idx` of a little-endian 2-bit stream
__device__ __forceinline__ uint64_t extract2(const uint32_t* plane, int idx, int nb) {
    const int bit = idx * 2, w = bit >> 5, sh = bit & 31;
    const uint32_t w0 = plane[w], w1 = plane[w + 1], w2 = plane[w + 2];
    const uint32_t lo = __builtin_amdgcn_alignbit(w1, w0, sh);
    const uint32_t hi = __builtin_amdgcn_alignbit(w2, w1, sh);
    const uint64_t v = ((uint64_t)hi << 32) | lo;
    return v & ((1ULL << (2 * nb)) - 1ULL);
}
// nb (1..31) bits starting at bit index `idx` of a little-endian 1-bit stream
__device__ __forceinline__ uint32_t extract1(const uint32_t* plane, int idx, int nb) {
    const int w = idx >> 5, sh = idx & 31;
    const uint32_t v = __builtin_amdgcn_alignbit(plane[w + 1], plane[w], sh);
    return v & ((1u << nb) - 1u);
}

// 4 ASCII bases -> 4x2-bit forward codes (base 0 in bits 0-1), 4x2-bit complement codes, 4 valid bits.
// AminoAcid.baseToNumber0 / baseToComplementNumber0 / baseToNumber>=0 (dna/AminoAcid.java:1284-1311):
// A/a C/c G/g T/t U/u are defined, every other byte is undefined and encodes as 0 in both tables.
__device__ __forceinline__ void encode4(uint32_t w, uint32_t& code8, uint32_t& comp8, uint32_t& valid4) {
    const uint32_t lower = w | 0x20202020u;
    auto eqb = [](uint32_t v, uint32_t pat) {     // 0x80 in every byte of v equal to the pattern byte (exact, no carries)
        const uint32_t t = v ^ pat;
        return ~(((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t) & 0x80808080u;
    };
    const uint32_t v = eqb(lower, 0x61616161u) | eqb(lower, 0x63636363u) | eqb(lower, 0x67676767u) |
                       eqb(lower, 0x74747474u) | eqb(lower, 0x75757575u);
    const uint32_t y = v >> 7;                    // 0x01 per valid byte
    const uint32_t vm = y * 3u;                   // 0x03 per valid byte
    uint32_t x = (w >> 1) & 0x03030303u;          // A:0 C:1 G:3 T/U:2
    x = (x ^ ((x >> 1) & 0x01010101u)) & vm;      // A:0 C:1 G:2 T/U:3, undefined:0
    const uint32_t c = (~x) & vm;                 // 3-x, undefined:0
    auto pack = [](uint32_t z) { uint32_t t = (z | (z >> 6)) & 0x000F000Fu; return (t | (t >> 12)) & 0xFFu; };
    code8 = pack(x);
    comp8 = pack(c);
    valid4 = (y | (y >> 7) | (y >> 14) | (y >> 21)) & 0xFu;
}

struct ReadResult { int a; int id; int newLen; };

// shared/TrimRead.java:304-345 trimByAmount on lengths
__device__ __forceinline__ int trim_by_amount(int len, int left, int right, int minRes, int& newLen) {
    left = max(left, 0); right = max(right, 0);
    if (len < 1) { newLen = len; return 0; }
    minRes = min(len, max(minRes, 0));
    if (left + right + minRes > len) { right = max(1, len - minRes); left = 0; }
    newLen = len - (left + right);
    return left + right;
}
__device__ __forceinline__ int imid(int lo, int x, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

// One read, one wave.  base0 = index of the read's first base in the tile's planes; T = 16*nchunks.
// ktrim: bbduk/BBDukProcessorS.java:1806-1811,1993-2140.  kfilter: :1534-1593.  All outputs wave-uniform.
__device__ ReadResult scan_read(const KParams& P, const uint32_t* s_fwd, const uint32_t* s_cmp, const uint32_t* s_nm,
                                const int base0, const int L, const int pairnum, const int T, const int lane,
                                int64_t* __restrict__ counters) {
    ReadResult R; R.a = 0; R.id = -1; R.newLen = L;
    const int k = P.k;
    const bool kfilter = (P.mode == BBDUK_MODE_KFILTER);
    if (P.storedKmers < 1) return R;
    if (kfilter) { if (L < k) return R; }
    else { if (L < max(1, P.useShort ? min(k, P.mink) : k)) return R; }
    if ((P.skipR1 && pairnum == 0) || (P.skipR2 && pairnum == 1)) return R;
    const int start = (P.restrictRight < 1 ? 0 : max(0, L - P.restrictRight));
    const int stop  = (P.restrictLeft  < 1 ? L : min(L, P.restrictLeft));

    // does [start,stop) hold an undefined base?  (only matters when forbidNs)
    bool hasN = false;
    if (P.forbidNs) {
        const int b0 = base0 + start, b1 = base0 + stop;            // bit range in the N plane
        uint32_t acc = 0;
        for (int w = (b0 >> 5) + lane; w <= ((b1 - 1) >> 5) && b1 > b0; w += 64) {
            uint32_t v = s_nm[w];
            const int lo = w << 5;
            if (lo < b0) v &= ~0u << (b0 - lo);
            if (lo + 32 > b1) v &= ~0u >> (lo + 32 - b1);
            acc |= v;
        }
        hasN = __ballot(acc != 0) != 0;
    }

    int found = 0, iFirst = BIGLOC, iLast = -1, id0 = -1;
    bool kfDone = false;
    const int first = max(start, k - 1);                            // i>=minlen (minlen=k-1)
    for (int ib = first; ib < stop; ib += 64) {
        const int i = ib + lane;
        const bool act = i < stop;
        const int ic = act ? i : stop - 1;                          // clamp so inactive lanes read in-bounds
        const int lo = max(start, ic - k + 1);
        const int nb = ic - lo + 1;                                 // bases in the window (== k unless cut by start)
        uint64_t kmer = extract2(s_fwd, T - 1 - (base0 + ic), nb);  // base ic in bits 0-1, base lo on top
        uint64_t rk   = extract2(s_cmp, base0 + lo, nb);            // base lo in bits 0-1
        int len = ic - start + 1;                                   // no reset seen
        if (hasN) {
            const uint32_t nwin = extract1(s_nm, base0 + lo, nb);   // bit t <=> base lo+t undefined
            if (nwin) {
                const int msb = 31 - __clz(nwin);
                len = nb - 1 - msb;                                 // bases after the last undefined one
                rk &= ~0ULL << (2 * (msb + 1));                     // rkmer was reset there; kmer keeps its history
            }
        }
        rk <<= 2 * (k - nb);                                        // base j sits at 2*(k-1-(i-j))
        const bool ok = act && len >= P.minlen2;
        int id = -1;
        if (ok) id = lookup(P, kmer, rk, P.kmask, k, P.qhdist);
        const uint64_t m = __ballot(id > 0);
        if (m) {
            if (!kfilter) {
                const int fl = __ffsll((unsigned long long)m) - 1, ll = 63 - __clzll((long long)m);
                if (found == 0) { iFirst = ib + fl; id0 = __builtin_amdgcn_readlane(id, fl); }
                iLast = ib + ll;
                found += __popcll(m);
            } else {
                const int c = __popcll(m);
                if (found + c > P.maxBadKmers) {                    // the (maxBadKmers+1)-th hit is in this pass
                    uint64_t mm = m;
                    for (int q = found; q < P.maxBadKmers; q++) mm &= mm - 1;
                    const int fl = __ffsll((unsigned long long)mm) - 1;
                    id0 = __builtin_amdgcn_readlane(id, fl);
                    found = P.maxBadKmers + 1;
                    kfDone = true;
                } else found += c;
            }
        }
        if (kfDone) break;
    }

    if (kfilter) {
        R.a = found;
        if (kfDone) {
            R.id = id0;
            if (lane == 0) {
                atomicAdd((unsigned long long*)&counters[BBDUK_NCOUNTERS + id0], 1ULL);
                atomicAdd((unsigned long long*)&counters[BBDUK_NCOUNTERS + P.numScaffolds + id0], (unsigned long long)L);
            }
        }
        return R;
    }

    const bool ktrimLeft = (P.mode == BBDUK_MODE_KTRIM_L);
    int minLoc = BIGLOC, minLocEx = BIGLOC, maxLoc = -1, maxLocEx = -1;
    if (found > 0) { minLoc = iFirst - k + 1; maxLoc = iLast; minLocEx = minLoc + k; maxLocEx = maxLoc - k; }

    if (P.useShort && found == 0) {                                 // :2034-2103, one lane per short length
        if (ktrimLeft) {
            const int Lmax = min(k, stop) - start;                  // lengths 1..Lmax, i = start+Ls-1
            const int Ls = P.mink + lane;
            const bool act = Ls <= Lmax;
            const int Lc = act ? Ls : max(1, min(Lmax, 1));
            int id = -1;
            if (act && Lmax >= 1) {
                const int i = start + Lc - 1;
                const uint64_t kmer = extract2(s_fwd, T - 1 - (base0 + i), Lc);
                const uint64_t rk   = extract2(s_cmp, base0 + start, Lc);
                id = lookup(P, kmer, rk, 1ULL << (2 * Lc), Lc, P.qhdist2);
            }
            const uint64_t m = __ballot(id > 0);
            if (m) {
                const int fl = __ffsll((unsigned long long)m) - 1, ll = 63 - __clzll((long long)m);
                id0 = __builtin_amdgcn_readlane(id, fl);             // first hit in scan order = shortest
                found = __popcll(m);
                minLoc = 0;
                minLocEx = start + (P.mink + fl) - 1 + 1;            // min over hits of i+1
                maxLoc = start + (P.mink + ll) - 1;                  // max over hits of i
                maxLocEx = 0;                                        // max(-1, 0)
            }
        } else {
            const int Lmax = (stop >= k ? k - 1 : stop);             // lengths 1..Lmax, i = stop-Ls
            const int Ls = P.mink + lane;
            const bool act = Ls <= Lmax;
            int id = -1;
            if (act) {
                const uint64_t kmer = extract2(s_fwd, T - 1 - (base0 + stop - 1), Ls);   // base stop-1 in bits 0-1
                const uint64_t rk   = extract2(s_cmp, base0 + stop - Ls, Ls) & P.mask;   // base i in bits 0-1
                id = lookup(P, kmer, rk, 1ULL << (2 * Ls), Ls, P.qhdist2);
            }
            const uint64_t m = __ballot(id > 0);
            if (m) {
                const int fl = __ffsll((unsigned long long)m) - 1, ll = 63 - __clzll((long long)m);
                id0 = __builtin_amdgcn_readlane(id, fl);             // first hit in scan order = shortest
                found = __popcll(m);
                minLoc = stop - (P.mink + ll);                       // last hit overwrites: longest match
                minLocEx = L;                                        // min(BIG, bases.length)
                maxLoc = L - 1;
                maxLocEx = stop - (P.mink + fl) - 1;                 // max over hits of i-1
            }
        }
    }
    if (found == 0) return R;
    if (lane == 0) {                                                // :2111-2119
        atomicAdd((unsigned long long*)&counters[BBDUK_NCOUNTERS + id0], 1ULL);
        atomicAdd((unsigned long long*)&counters[BBDUK_NCOUNTERS + P.numScaffolds + id0], (unsigned long long)L);
    }
    if (P.trimPad != 0) {                                           // :2121-2126
        maxLoc = imid(0, maxLoc + P.trimPad, L);
        minLoc = imid(0, minLoc - P.trimPad, L);
        maxLocEx = imid(0, maxLocEx + P.trimPad, L);
        minLocEx = imid(0, minLocEx - P.trimPad, L);
    }
    R.id = id0;
    if (ktrimLeft) {   // trimToPosition(r, leftLoc, len-1, 1)  (shared/TrimRead.java:273-276)
        const int leftLoc = P.ktrimExclusive ? maxLocEx + 1 : maxLoc + 1;
        R.a = trim_by_amount(L, leftLoc, L - (L - 1) - 1, 1, R.newLen);
    } else {           // trimToPosition(r, 0, rightLoc, 1)
        const int rightLoc = P.ktrimExclusive ? minLocEx - 1 : minLoc - 1;
        R.a = trim_by_amount(L, 0, L - rightLoc - 1, 1, R.newLen);
    }
    return R;
}

// --------------------------------------------------------------------------------------------------
// The batch kernel: persistent blocks walk tiles of TILE_READS reads.
__global__ __launch_bounds__(BLOCK_THREADS)
void bbduk_batch_kernel(const KParams P, const uint8_t* __restrict__ bases, const int64_t* __restrict__ offsets,
                        const int64_t n, const int64_t totalBases, const int paired,
                        int32_t* __restrict__ outA, int32_t* __restrict__ outId, uint8_t* __restrict__ outFlags,
                        int64_t* __restrict__ counters) {
    __shared__ uint32_t s_fwd[CAP_CHUNKS + 4];
    __shared__ uint32_t s_cmp[CAP_CHUNKS + 4];
    __shared__ uint32_t s_nm[CAP_CHUNKS / 2 + 4];
    __shared__ int64_t  s_off[TILE_READS + 1];
    __shared__ int32_t  s_a[TILE_READS];
    __shared__ int32_t  s_id[TILE_READS];
    __shared__ uint8_t  s_fl[TILE_READS];
    __shared__ long long s_ctr[NWAVES][10];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t ntiles = (n + TILE_READS - 1) / TILE_READS;
    const int step = paired ? 2 : 1;
    long long c_[10];
#pragma unroll
    for (int q = 0; q < 10; q++) c_[q] = 0;

    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t r0 = tile * TILE_READS;
        const int cnt = (int)min((int64_t)TILE_READS, n - r0);
        __syncthreads();                                            // previous tile's LDS fully consumed
        if (tid <= cnt) s_off[tid] = offsets[r0 + tid];
        __syncthreads();

        int s = 0;
        while (s < cnt) {
            // how many consecutive reads fit the planes?
            const int64_t off_s = s_off[s];
            const int cand = s + 1 + tid;
            const int okc = (cand <= cnt) && (s_off[min(cand, cnt)] - off_s <= (int64_t)(CAP_BASES - 32));
            int fit = __syncthreads_count(okc);
            if (paired) fit &= ~1;
            if (fit == 0) {                                         // read (or pair) too long for the LDS tile
                if (tid == 0) atomicMax((unsigned long long*)&counters[BBDUK_CTR_STATUS], (unsigned long long)(-BBDUK_ERR_READ_TOO_LONG));
                const int skip = min(step, cnt - s);
                if (tid < skip) {
                    s_a[s + tid] = 0; s_id[s + tid] = -1; s_fl[s + tid] = 0;
                }
                s += skip;
                continue;
            }
            const int e = s + fit;
            const int64_t B0 = off_s, B1 = s_off[e];
            const int64_t A0 = B0 & ~15LL;
            const int nchunks = (int)((B1 - A0 + 15) >> 4);
            const int T = nchunks * 16;
            // ---- stage: 16 bases per thread-iteration -> three bit-planes
            for (int c = tid; c < nchunks; c += BLOCK_THREADS) {
                const int64_t a = A0 + 16LL * c;
                uint32_t w[4];
                if (a + 16 <= totalBases) {
                    const uint4 v = *reinterpret_cast<const uint4*>(bases + a);
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
                uint32_t code = 0, comp = 0, valid = 0;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    uint32_t c8, m8, v4;
                    encode4(w[q], c8, m8, v4);
                    code |= c8 << (8 * q); comp |= m8 << (8 * q); valid |= v4 << (4 * q);
                }
                uint32_t r = __brev(code);                          // reverse the order of the 16 symbols
                r = ((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1);
                s_fwd[nchunks - 1 - c] = r;
                s_cmp[c] = comp;
                reinterpret_cast<uint16_t*>(s_nm)[c] = (uint16_t)(~valid & 0xFFFFu);
            }
            if (tid == 0 && (nchunks & 1)) reinterpret_cast<uint16_t*>(s_nm)[nchunks] = 0;
            __syncthreads();

            // ---- scan: one wave per read / pair
            const int nunits = (e - s) / step;
            for (int u = wave; u < nunits; u += NWAVES) {
                const int ra = s + u * step;
                const int L1 = (int)(s_off[ra + 1] - s_off[ra]);
                const ReadResult A = scan_read(P, s_fwd, s_cmp, s_nm, (int)(s_off[ra] - A0), L1, 0, T, lane, counters);
                ReadResult Bz; Bz.a = 0; Bz.id = -1; Bz.newLen = 0;
                int L2 = 0;
                if (paired) {
                    L2 = (int)(s_off[ra + 2] - s_off[ra + 1]);
                    Bz = scan_read(P, s_fwd, s_cmp, s_nm, (int)(s_off[ra + 1] - A0), L2, 1, T, lane, counters);
                }
                // ---- pair stage (bbduk/BBDukProcessorS.java:807-818, 948-1093, 1431-1443), wave-uniform
                const int pairCount = paired ? 2 : 1;
                const float f1 = (float)L1 * P.minLenFraction, f2 = (float)L2 * P.minLenFraction;
                const int minlen1 = (int)(f1 > (float)P.minReadLength ? f1 : (float)P.minReadLength);
                const int minlen2 = (int)(f2 > (float)P.minReadLength ? f2 : (float)P.minReadLength);
                bool d1 = false, d2 = false, remove = false;
                c_[BBDUK_READS_IN] += pairCount; c_[BBDUK_BASES_IN] += L1 + L2;
                if (P.storedKmers > 0) {
                    if (P.mode != BBDUK_MODE_KFILTER) {
                        int xsum = A.a + Bz.a, rkt = (A.a > 0) + (Bz.a > 0);
                        d1 = A.newLen < minlen1;
                        d2 = paired && (Bz.newLen < minlen2);
                        if ((P.rieb && (d1 || d2)) || (d1 && (!paired || d2))) { xsum += A.newLen + Bz.newLen; rkt = pairCount; remove = true; }
                        c_[BBDUK_BASES_KTRIMMED] += xsum; c_[BBDUK_READS_KTRIMMED] += rkt;
                    } else {
                        d1 = A.a > P.maxBadKmers;
                        d2 = paired && (Bz.a > P.maxBadKmers);
                        if ((P.rieb && (d1 || d2)) || (d1 && (!paired || d2))) {
                            remove = true;
                            c_[BBDUK_READS_KFILTERED] += pairCount; c_[BBDUK_BASES_KFILTERED] += L1 + L2;
                        }
                    }
                }
                if (remove) { c_[BBDUK_READS_OUTM] += pairCount; c_[BBDUK_BASES_OUTM] += A.newLen + Bz.newLen; }
                else        { c_[BBDUK_READS_OUTU] += pairCount; c_[BBDUK_BASES_OUTU] += A.newLen + Bz.newLen; }
                if (lane == 0) {
                    s_a[ra] = A.a; s_id[ra] = A.id;
                    s_fl[ra] = (uint8_t)((d1 ? BBDUK_FLAG_DISCARDED : 0) | (remove ? BBDUK_FLAG_REMOVED : 0));
                    if (paired) {
                        s_a[ra + 1] = Bz.a; s_id[ra + 1] = Bz.id;
                        s_fl[ra + 1] = (uint8_t)((d2 ? BBDUK_FLAG_DISCARDED : 0) | (remove ? BBDUK_FLAG_REMOVED : 0));
                    }
                }
            }
            __syncthreads();
            s = e;
        }
        // ---- coalesced write-back of the tile's results
        if (tid < cnt) {
            outA[r0 + tid] = s_a[tid];
            outId[r0 + tid] = s_id[tid];
            outFlags[r0 + tid] = s_fl[tid];
        }
    }
    // ---- counters: per-wave registers -> LDS -> one atomic per slot per block
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < 10; q++) s_ctr[wave][q] = c_[q];
    }
    __syncthreads();
    if (tid < 10) {
        long long v = 0;
        for (int w = 0; w < NWAVES; w++) v += s_ctr[w][tid];
        if (v) atomicAdd((unsigned long long*)&counters[tid], (unsigned long long)v);
    }
}

__global__ void bbduk_lookup_kernel(const KParams P, const int64_t* keys, int64_t n, int32_t* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (keys[i] < 0) ? -1 : table_get(P, (uint64_t)keys[i]);
}

__global__ void bbduk_synth_kernel(const bb_synth_dev sp, const int64_t firstPair, const int64_t nPairs,
                                   uint8_t* __restrict__ bases, int64_t* __restrict__ offsets) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // one thread per base
    const int64_t per = 2LL * sp.read_len;
    const int64_t total = nPairs * per;
    if (g <= 2 * nPairs) offsets[g] = g * sp.read_len;
    if (g >= total) return;
    const int64_t p = g / per;
    const int32_t rem = (int32_t)(g - p * per);
    const int32_t mate = rem >= sp.read_len ? 1 : 0;
    const int32_t j = rem - mate * sp.read_len;
    const bb_pair_hdr h = bb_synth_pair_header(sp, (uint64_t)(firstPair + p));
    bases[g] = bb_synth_read_base(sp, (uint64_t)(firstPair + p), h, mate, j);
}

// --------------------------------------------------------------------------------------------------
// host side of the C ABI

struct bbduk_handle {
    bbduk_params p;
    std::string err;
    std::mutex mu;
    bool finalized = false;
    std::vector<int64_t> hkeys;          // staged (key,value) pairs before finalize
    std::vector<int32_t> hvals;
    int64_t nkeys = 0;
    uint64_t* d_tkeys = nullptr; int32_t* d_tvals = nullptr; uint64_t cap = 0; int hashShift = 0;
    // host-operator staging
    uint8_t* d_bases = nullptr; size_t cap_bases = 0;
    int64_t* d_off = nullptr;   size_t cap_reads = 0;
    int32_t* d_a = nullptr; int32_t* d_id = nullptr; uint8_t* d_fl = nullptr;
    int64_t* d_counters = nullptr;
    hipStream_t stream = nullptr;
    int numCU = 256;
};

#define HIP_TRY(h, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    (h)->err = std::string(#call) + ": " + hipGetErrorString(e_); return BBDUK_ERR_DEVICE; } } while (0)

static int fail(bbduk_handle* h, int code, const char* msg) { if (h) h->err = msg; return code; }

extern "C" int bbduk_abi_version(void) { return BBDUK_ABI_VERSION; }
extern "C" const char* bbduk_last_error(const bbduk_handle* h) { return h ? h->err.c_str() : "null handle"; }

extern "C" int bbduk_create(const bbduk_params* p, bbduk_handle** out) {
    if (!p || !out) return BBDUK_ERR_ARG;
    *out = nullptr;
    if (p->abi_version != BBDUK_ABI_VERSION) return BBDUK_ERR_ARG;
    if (p->k < 1 || p->k > 31) return BBDUK_ERR_ARG;
    if (p->mode != BBDUK_MODE_KFILTER && p->mode != BBDUK_MODE_KTRIM_R && p->mode != BBDUK_MODE_KTRIM_L) return BBDUK_ERR_ARG;
    if (p->qhdist < 0 || p->qhdist > 2 || p->qhdist2 < 0 || p->qhdist2 > 2) return BBDUK_ERR_ARG;
    if (p->numScaffolds < 1 || p->maxBadKmers < 0) return BBDUK_ERR_ARG;
    const bool useShort = p->mink > 0 && p->mink < p->k;
    if (useShort && p->mode == BBDUK_MODE_KFILTER) return BBDUK_ERR_ARG;      // BBDukParser.java:301
    if (useShort && p->middleMask != -1) return BBDUK_ERR_ARG;                // BBDukProcessorS.java:2035 assert
    if (p->minlen != p->k - 1) return BBDUK_ERR_ARG;
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
        hipMemset(h->d_counters, 0, nc * sizeof(int64_t)) != hipSuccess) { hipStreamDestroy(h->stream); delete h; return BBDUK_ERR_DEVICE; }
    *out = h;
    return BBDUK_OK;
}

extern "C" int bbduk_destroy(bbduk_handle* h) {
    if (!h) return BBDUK_ERR_ARG;
    hipSetDevice(h->p.device);
    hipFree(h->d_tkeys); hipFree(h->d_tvals); hipFree(h->d_bases); hipFree(h->d_off);
    hipFree(h->d_a); hipFree(h->d_id); hipFree(h->d_fl); hipFree(h->d_counters);
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
    return BBDUK_OK;
}

extern "C" int bbduk_upload_pairs(bbduk_handle* h, const int64_t* keys, const int32_t* values, int64_t n) {
    if (!h || n < 0 || (n > 0 && (!keys || !values))) return fail(h, BBDUK_ERR_ARG, "upload_pairs: bad argument");
    std::lock_guard<std::mutex> g(h->mu);
    if (h->finalized) return fail(h, BBDUK_ERR_STATE, "table already finalized");
    for (int64_t i = 0; i < n; i++) {
        if (keys[i] < 0) return fail(h, BBDUK_ERR_ARG, "upload_pairs: negative key");
        h->hkeys.push_back(keys[i]); h->hvals.push_back(values[i]);
    }
    return BBDUK_OK;
}

extern "C" int bbduk_upload_table_way(bbduk_handle* h, int32_t way, int32_t prime, const int64_t* keys, const int32_t* values,
                                      int64_t ncells, const int64_t* vkeys, const int32_t* vvals, int64_t nvictims) {
    (void)way; (void)prime;      // the device re-hashes into its own layout; geometry of the Java image is not needed
    if (!h || ncells < 0 || nvictims < 0 || (ncells > 0 && (!keys || !values)) || (nvictims > 0 && (!vkeys || !vvals)))
        return fail(h, BBDUK_ERR_ARG, "upload_table_way: bad argument");
    std::lock_guard<std::mutex> g(h->mu);
    if (h->finalized) return fail(h, BBDUK_ERR_STATE, "table already finalized");
    for (int64_t i = 0; i < ncells; i++) if (keys[i] >= 0) {     // NOT_PRESENT = -1 (kmer/AbstractKmerTable.java:807)
        h->hkeys.push_back(keys[i]); h->hvals.push_back(values[i]);
    }
    for (int64_t i = 0; i < nvictims; i++) if (vkeys[i] >= 0) { h->hkeys.push_back(vkeys[i]); h->hvals.push_back(vvals[i]); }
    return BBDUK_OK;
}

extern "C" int bbduk_finalize_table(bbduk_handle* h) {
    if (!h) return BBDUK_ERR_ARG;
    std::lock_guard<std::mutex> g(h->mu);
    if (h->finalized) return fail(h, BBDUK_ERR_STATE, "table already finalized");
    HIP_TRY(h, hipSetDevice(h->p.device));
    const size_t n = h->hkeys.size();
    uint64_t cap = 1024;
    while (cap < 2 * (uint64_t)n + 2) cap <<= 1;                  // load factor <= 0.5
    int bits = 0; while ((1ULL << bits) < cap) bits++;
    std::vector<uint64_t> tk(cap, EMPTY_KEY);
    std::vector<int32_t> tv(cap, 0);
    const int shift = 64 - bits;
    int64_t distinct = 0;
    for (size_t i = 0; i < n; i++) {                              // first writer wins (HashArray.setIfNotPresent)
        const uint64_t key = (uint64_t)h->hkeys[i];
        uint64_t s = (key * HASH_MULT) >> shift;
        for (;;) {
            if (tk[s] == key) break;
            if (tk[s] == EMPTY_KEY) { tk[s] = key; tv[s] = h->hvals[i]; distinct++; break; }
            s = (s + 1) & (cap - 1);
        }
    }
    HIP_TRY(h, hipMalloc(&h->d_tkeys, cap * sizeof(uint64_t)));
    HIP_TRY(h, hipMalloc(&h->d_tvals, cap * sizeof(int32_t)));
    HIP_TRY(h, hipMemcpy(h->d_tkeys, tk.data(), cap * sizeof(uint64_t), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->d_tvals, tv.data(), cap * sizeof(int32_t), hipMemcpyHostToDevice));
    h->cap = cap; h->hashShift = shift; h->nkeys = distinct;
    h->hkeys.clear(); h->hkeys.shrink_to_fit(); h->hvals.clear(); h->hvals.shrink_to_fit();
    h->finalized = true;
    return BBDUK_OK;
}

extern "C" int64_t bbduk_table_size(const bbduk_handle* h) { return (h && h->finalized) ? h->nkeys : -1; }
extern "C" int64_t bbduk_table_bytes(const bbduk_handle* h) { return (h && h->finalized) ? (int64_t)(h->cap * 12) : -1; }

static KParams make_kparams(const bbduk_handle* h) {
    const bbduk_params& p = h->p;
    KParams K;
    memset(&K, 0, sizeof K);
    K.mode = p.mode; K.k = p.k; K.mink = p.mink; K.rcomp = p.rcomp; K.forbidNs = p.forbidNs;
    K.minlen = p.minlen; K.minlen2 = p.minlen2; K.qhdist = p.qhdist; K.qhdist2 = p.qhdist2;
    K.maxBadKmers = p.maxBadKmers; K.minReadLength = p.minReadLength; K.minLenFraction = p.minLenFraction;
    K.rieb = p.removePairsIfEitherBad; K.trimPad = p.trimPad; K.ktrimExclusive = p.ktrimExclusive;
    K.restrictLeft = p.restrictLeft; K.restrictRight = p.restrictRight; K.skipR1 = p.skipR1; K.skipR2 = p.skipR2;
    K.numScaffolds = p.numScaffolds;
    K.useShort = (p.mink > 0 && p.mink < p.k) ? 1 : 0;
    K.mask = (2 * p.k > 63) ? ~0ULL : ~(~0ULL << (2 * p.k));
    K.kmask = 1ULL << (2 * p.k);
    K.middleMask = (uint64_t)p.middleMask;
    K.tkeys = h->d_tkeys; K.tvals = h->d_tvals; K.capMask = h->cap - 1; K.hashShift = h->hashShift;
    K.storedKmers = h->nkeys;
    return K;
}

static int launch_batch(bbduk_handle* h, int wantKfilter, const uint8_t* d_bases, const int64_t* d_offsets, int64_t n,
                        int64_t total_bases, int32_t paired, int32_t* d_a, int32_t* d_id, uint8_t* d_fl,
                        int64_t* d_counters, hipStream_t st) {
    if (!h) return BBDUK_ERR_ARG;
    if (!h->finalized) return fail(h, BBDUK_ERR_STATE, "table not finalized");
    if ((h->p.mode == BBDUK_MODE_KFILTER) != (wantKfilter != 0)) return fail(h, BBDUK_ERR_STATE, "operator does not match the mode given to bbduk_create");
    if (n < 0 || total_bases < 0 || (paired && (n & 1))) return fail(h, BBDUK_ERR_ARG, "bad batch shape");
    if (n == 0) return BBDUK_OK;
    if (!d_bases && total_bases > 0) return fail(h, BBDUK_ERR_ARG, "null bases");
    if (!d_offsets || !d_a || !d_id || !d_fl || !d_counters) return fail(h, BBDUK_ERR_ARG, "null buffer");
    if (((uintptr_t)d_bases & 15) != 0) return fail(h, BBDUK_ERR_ARG, "d_bases must be 16-byte aligned");
    const KParams K = make_kparams(h);
    const int64_t ntiles = (n + TILE_READS - 1) / TILE_READS;
    const int grid = (int)std::min<int64_t>(ntiles, (int64_t)h->numCU * 8);
    hipLaunchKernelGGL(bbduk_batch_kernel, dim3(grid), dim3(BLOCK_THREADS), 0, st, K, d_bases, d_offsets, n, total_bases,
                       (int)paired, d_a, d_id, d_fl, d_counters);
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

static int host_batch(bbduk_handle* h, int wantKfilter, const uint8_t* bases, const int64_t* offsets, int64_t n, int32_t paired,
                      int32_t* out_a, int32_t* out_id, uint8_t* out_fl) {
    if (!h) return BBDUK_ERR_ARG;
    if (n < 0 || !offsets || (n > 0 && (!out_a || !out_id || !out_fl))) return fail(h, BBDUK_ERR_ARG, "bad argument");
    if (n == 0) return BBDUK_OK;
    const int64_t total = offsets[n];
    if (offsets[0] != 0 || total < 0 || (total > 0 && !bases)) return fail(h, BBDUK_ERR_ARG, "bad offsets");
    std::lock_guard<std::mutex> g(h->mu);       // one staging area per handle: concurrent submitters serialise here
    HIP_TRY(h, hipSetDevice(h->p.device));
    if ((size_t)total + 16 > h->cap_bases) {
        hipFree(h->d_bases); h->d_bases = nullptr;
        h->cap_bases = (size_t)total + 16 + (size_t)total / 4;
        HIP_TRY(h, hipMalloc(&h->d_bases, h->cap_bases));
    }
    if ((size_t)n + 1 > h->cap_reads) {
        hipFree(h->d_off); hipFree(h->d_a); hipFree(h->d_id); hipFree(h->d_fl);
        h->d_off = nullptr; h->d_a = nullptr; h->d_id = nullptr; h->d_fl = nullptr;
        h->cap_reads = (size_t)n + 1 + (size_t)n / 4;
        HIP_TRY(h, hipMalloc(&h->d_off, h->cap_reads * sizeof(int64_t)));
        HIP_TRY(h, hipMalloc(&h->d_a, h->cap_reads * sizeof(int32_t)));
        HIP_TRY(h, hipMalloc(&h->d_id, h->cap_reads * sizeof(int32_t)));
        HIP_TRY(h, hipMalloc(&h->d_fl, h->cap_reads));
    }
    if (total > 0) HIP_TRY(h, hipMemcpyAsync(h->d_bases, bases, (size_t)total, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->d_off, offsets, (size_t)(n + 1) * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
    const int rc = launch_batch(h, wantKfilter, h->d_bases, h->d_off, n, total, paired, h->d_a, h->d_id, h->d_fl, h->d_counters, h->stream);
    if (rc != BBDUK_OK) return rc;
    HIP_TRY(h, hipMemcpyAsync(out_a, h->d_a, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(out_id, h->d_id, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(out_fl, h->d_fl, (size_t)n, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    int64_t status = 0;
    HIP_TRY(h, hipMemcpy(&status, h->d_counters + BBDUK_CTR_STATUS, sizeof status, hipMemcpyDeviceToHost));
    if (status != 0) {
        int64_t z = 0;
        hipMemcpy(h->d_counters + BBDUK_CTR_STATUS, &z, sizeof z, hipMemcpyHostToDevice);
        return fail(h, -(int)status, "device reported an error (read longer than BBDUK_MAX_READ_LEN?)");
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
        const int64_t blocks = (total + 255) / 256;
        hipLaunchKernelGGL(bbduk_synth_kernel, dim3((unsigned)blocks), dim3(256), 0, st, d, first_pair, n_pairs, d_bases, d_offsets);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) rc = BBDUK_ERR_DEVICE;
    } else rc = BBDUK_ERR_DEVICE;
    hipFree(da1); hipFree(da2); hipFree(dc);
    return rc;
}
