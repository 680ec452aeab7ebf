// bbduk_ingest.hip -- device-side FASTQ ingest for the BBDuk k-mer path, gfx950.
//
// What it restates: the record splitting of the reference's FASTQ reader.  stream/FASTQ.java:778-853 (toReadList) takes
// the lines fileIO/ByteFile.nextLine() returns -- a line ends at '\n', one preceding '\r' is dropped -- and turns every
// four of them into one read: quad[0] = '@' header, quad[1] = bases, quad[2] = '+' line, quad[3] = qualities
// (quadToRead_slow asserts the '@' and the '+', :1047-1049).  With two input files read i of file 1 and read i of file 2
// are mates (stream/ConcurrentGenericReadInputStream pairs them); interleaved files simply alternate.
//
// What it produces, all in HBM and without the host touching a base: the byte offset of every line (so that a writer
// can cut trimmed records out of the same text), the reads' base offsets, and the reads in the packed boundary format
// of include/bbduk_gpu.h (2-bit codes + undefined bits), ready for bbduk_*_batch_packed_device.
//
// Kernels (all HBM-streaming; the text is read three times, 16 B per lane and load):
//   fq_count_kernel     newlines per 16 KB block (SWAR byte compare + popcount)
//   scan_sums_kernel    exclusive scan of the block sums (one workgroup)
//   fq_lines_kernel     line start offsets: block-local rank + block prefix
//   fq_records_kernel   per read: checks '@' / '+', bases == qualities in length, length of the bases line
//   block_sum/scan_final  lengths -> base offsets (int64)
//   fq_pack_kernel      one thread per 16 output bases: locate the read (binary search narrowed per block), gather bytes,
//                       encode (dna/AminoAcid.java:1284-1298), store one code word + 16 undefined bits
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include "../../include/bbduk_gpu.h"

namespace {

constexpr int FQ_THREADS = 256;
constexpr int FQ_BYTES_PER_THREAD = 64;
constexpr int FQ_BLOCK_BYTES = FQ_THREADS * FQ_BYTES_PER_THREAD;      // 16 KB of text per workgroup

// exact per-byte equality mask (0x80 in every byte of x that equals c)
__device__ __forceinline__ uint32_t eq_bytes(uint32_t x, uint32_t c4) {
    const uint32_t t = x ^ c4;
    return ~(((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t | 0x7F7F7F7Fu);
}

// 16-byte values as two 64-bit halves; byte j of the value is byte j of the text
struct U128 { uint64_t lo, hi; };
__device__ __forceinline__ U128 shl_bytes(const U128 v, const int sft) {          // s in [0, 16]
    if (sft == 0) return v;
    if (sft >= 16) return U128{0, 0};
    if (sft < 8) return U128{v.lo << (8 * sft), (v.hi << (8 * sft)) | (v.lo >> (64 - 8 * sft))};
    return U128{0, v.lo << (8 * (sft - 8))};
}
__device__ __forceinline__ U128 low_bytes(const U128 v, const int c) {            // keep the c lowest bytes, c in [0, 16]
    if (c >= 16) return v;
    if (c > 8) return U128{v.lo, v.hi & (~0ULL >> (8 * (16 - c)))};
    if (c == 8) return U128{v.lo, 0};
    if (c == 0) return U128{0, 0};
    return U128{v.lo & (~0ULL >> (8 * (8 - c))), 0};
}
// 16 bytes at t+src; bytes at or past `lim` (the end of the indexed text) read as 0 without being touched
__device__ __forceinline__ U128 load16(const uint8_t* __restrict__ t, const int64_t src, const int64_t lim) {
    U128 v;
    if (src + 16 <= lim) { __builtin_memcpy(&v, t + src, 16); return v; }
    v.lo = v.hi = 0;
    for (int j = 0; j < 16 && src + j < lim; j++) { if (j < 8) v.lo |= (uint64_t)t[src + j] << (8 * j); else v.hi |= (uint64_t)t[src + j] << (8 * (j - 8)); }
    return v;
}
// 16 ASCII bases -> 32 code bits + 16 undefined bits (dna/AminoAcid.java:1284-1298), SWAR
__device__ __forceinline__ void encode16(const U128 v, uint32_t& code, uint32_t& und) {
    const uint32_t x[4] = {(uint32_t)v.lo, (uint32_t)(v.lo >> 32), (uint32_t)v.hi, (uint32_t)(v.hi >> 32)};
    code = 0; und = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t y = x[q] | 0x20202020u;
        const uint32_t ok = (eq_bytes(y, 0x61616161u) | eq_bytes(y, 0x63636363u) | eq_bytes(y, 0x67676767u) | eq_bytes(y, 0x74747474u) | eq_bytes(y, 0x75757575u)) >> 7;
        uint32_t z = (y >> 1) & 0x03030303u;          // a 0, c 1, g 3, t/u 2 ...
        z ^= (z >> 1) & 0x01010101u;                  // ... -> a 0, c 1, g 2, t/u 3
        z &= ok * 3u;
        code |= ((z | (z >> 6) | (z >> 12) | (z >> 18)) & 0xFFu) << (8 * q);
        und |= ((~(ok | (ok >> 7) | (ok >> 14) | (ok >> 21))) & 0xFu) << (4 * q);
    }
}

// 64 bytes of text starting at byte `a`, zero past nbytes
__device__ __forceinline__ void load64(const uint8_t* __restrict__ text, const int64_t a, const int64_t nbytes, uint32_t* w) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int64_t p = a + 16 * q;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (p + 16 <= nbytes) __builtin_memcpy(&v, text + p, 16);   // the text may start at any byte (a chunk's carried-over tail)
        else if (p < nbytes) {
            uint32_t t[4] = {0, 0, 0, 0};
            for (int b = 0; b < 16 && p + b < nbytes; b++) t[b >> 2] |= (uint32_t)text[p + b] << (8 * (b & 3));
            v = make_uint4(t[0], t[1], t[2], t[3]);
        }
        w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
    }
}

template <class T>
__device__ __forceinline__ T block_exclusive_scan(T v, T* s_tmp, T& total) {      // FQ_THREADS threads
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    T x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const T y = __shfl_up(x, o); if (lane >= o) x += y; }
    if (lane == 63) s_tmp[wave] = x;
    __syncthreads();
    T base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < FQ_THREADS / 64; w++) { const T c = s_tmp[w]; if (w < wave) base += c; tot += c; }
    __syncthreads();
    total = tot;
    return base + x - v;
}

__global__ __launch_bounds__(FQ_THREADS)
void fq_count_kernel(const uint8_t* __restrict__ text, const int64_t nbytes, int64_t* __restrict__ sums) {
    __shared__ int s_tmp[FQ_THREADS / 64];
    const int64_t a = (int64_t)blockIdx.x * FQ_BLOCK_BYTES + (int64_t)threadIdx.x * FQ_BYTES_PER_THREAD;
    uint32_t w[16];
    int c = 0;
    if (a < nbytes) {
        load64(text, a, nbytes, w);
#pragma unroll
        for (int q = 0; q < 16; q++) c += __popc(eq_bytes(w[q], 0x0A0A0A0Au));
    }
    int total;
    block_exclusive_scan(c, s_tmp, total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// in-place exclusive scan of sums[0..nb), total in sums[nb]; one workgroup of 1024 threads
__global__ __launch_bounds__(1024)
void scan_sums_kernel(int64_t* __restrict__ sums, const int64_t nb) {
    __shared__ int64_t s_w[16];
    __shared__ int64_t s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int64_t i0 = 0; i0 < nb; i0 += 1024) {
        const int64_t i = i0 + tid;
        const int64_t v = i < nb ? sums[i] : 0;
        int64_t x = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int64_t y = __shfl_up(x, o); if (lane >= o) x += y; }
        if (lane == 63) s_w[wave] = x;
        __syncthreads();
        int64_t base = s_carry, tot = 0;
        for (int w = 0; w < 16; w++) { const int64_t c = s_w[w]; if (w < wave) base += c; tot += c; }
        if (i < nb) sums[i] = base + x - v;
        __syncthreads();
        if (tid == 0) s_carry += tot;
        __syncthreads();
    }
    if (tid == 0) sums[nb] = s_carry;
}

// lines[j] = byte offset of line j (lines[0] = 0, lines[j] = position after the j-th newline); entries beyond cap are dropped
__global__ __launch_bounds__(FQ_THREADS)
void fq_lines_kernel(const uint8_t* __restrict__ text, const int64_t nbytes, const int64_t* __restrict__ sums,
                     int64_t* __restrict__ lines, const int64_t cap) {
    __shared__ int s_tmp[FQ_THREADS / 64];
    const int64_t a = (int64_t)blockIdx.x * FQ_BLOCK_BYTES + (int64_t)threadIdx.x * FQ_BYTES_PER_THREAD;
    uint32_t w[16];
    int c = 0;
    if (a < nbytes) {
        load64(text, a, nbytes, w);
#pragma unroll
        for (int q = 0; q < 16; q++) c += __popc(eq_bytes(w[q], 0x0A0A0A0Au));
    }
    int total;
    const int local = block_exclusive_scan(c, s_tmp, total);
    if (blockIdx.x == 0 && threadIdx.x == 0) lines[0] = 0;
    if (c == 0) return;
    int64_t j = sums[blockIdx.x] + local + 1;
#pragma unroll
    for (int q = 0; q < 16; q++) {
        uint32_t m = eq_bytes(w[q], 0x0A0A0A0Au);
        while (m) {
            const int b = (__ffs(m) - 1) >> 3;
            m &= m - 1;
            if (j <= cap) lines[j] = a + 4 * q + b + 1;
            j++;
        }
    }
}

// length of line [s, e) where e is the offset of the next line (i.e. one past the '\n'): drop the '\n' and one '\r'
__device__ __forceinline__ int line_len(const uint8_t* __restrict__ text, const int64_t s, const int64_t e) {
    int64_t len = e - s - 1;
    if (len > 0 && text[e - 2] == '\r') len--;
    return (int)len;
}

// read i -> (stream i % ns, record i / ns).  lens[i] = bases of read i; firstBad = smallest malformed read index.
__global__ void fq_records_kernel(const uint8_t* __restrict__ t1, const int64_t* __restrict__ l1,
                                  const uint8_t* __restrict__ t2, const int64_t* __restrict__ l2, const int ns,
                                  const int64_t n, int32_t* __restrict__ lens, unsigned long long* __restrict__ firstBad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool second = (ns == 2) && (i & 1);
    const uint8_t* t = second ? t2 : t1;
    const int64_t* l = second ? l2 : l1;
    const int64_t r = (ns == 2) ? (i >> 1) : i;
    const int64_t h = l[4 * r], s = l[4 * r + 1], p = l[4 * r + 2], q = l[4 * r + 3], e = l[4 * r + 4];
    const int ls = line_len(t, s, p), lq = line_len(t, q, e);
    const bool ok = line_len(t, h, s) > 0 && t[h] == '@' && line_len(t, p, q) > 0 && t[p] == '+' && ls == lq;   // FASTQ.java:1047-1049
    lens[i] = ls;
    if (!ok) atomicMin(firstBad, (unsigned long long)i);
}

constexpr int SC_PER_THREAD = 8;
constexpr int SC_BLOCK = FQ_THREADS * SC_PER_THREAD;
__global__ __launch_bounds__(FQ_THREADS)
void block_sum_kernel(const int32_t* __restrict__ in, const int64_t n, int64_t* __restrict__ sums) {
    __shared__ int64_t s_tmp[FQ_THREADS / 64];
    const int64_t a = (int64_t)blockIdx.x * SC_BLOCK + (int64_t)threadIdx.x * SC_PER_THREAD;
    int64_t c = 0;                                  // 64-bit: a block of long reads (or of their output records) can pass 2^31
    for (int q = 0; q < SC_PER_THREAD; q++) if (a + q < n) c += in[a + q];
    int64_t total;
    block_exclusive_scan(c, s_tmp, total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}
__global__ __launch_bounds__(FQ_THREADS)
void scan_final_kernel(const int32_t* __restrict__ in, const int64_t n, const int64_t* __restrict__ sums, int64_t* __restrict__ out) {
    __shared__ int64_t s_tmp[FQ_THREADS / 64];
    const int64_t a = (int64_t)blockIdx.x * SC_BLOCK + (int64_t)threadIdx.x * SC_PER_THREAD;
    int v[SC_PER_THREAD]; int64_t c = 0;
    for (int q = 0; q < SC_PER_THREAD; q++) { v[q] = (a + q < n) ? in[a + q] : 0; c += v[q]; }
    int64_t total;
    int64_t run = sums[blockIdx.x] + block_exclusive_scan(c, s_tmp, total);
    for (int q = 0; q < SC_PER_THREAD; q++) {
        if (a + q <= n) out[a + q] = run;            // out[n] = total
        run += v[q];
    }
}

// 16 output bases per thread.  offsets[0..n] ascending; read r owns bases [offsets[r], offsets[r+1]).
__device__ __forceinline__ int64_t read_of_base(const int64_t* __restrict__ offsets, int64_t lo, int64_t hi, const int64_t b) {
    while (lo < hi) {                                 // largest r in [lo, hi] with offsets[r] <= b
        const int64_t mid = (lo + hi + 1) >> 1;
        if (offsets[mid] <= b) lo = mid; else hi = mid - 1;
    }
    return lo;
}
// Which item (read / output record) owns position k*4096, for every k: one thread per item scatters its index to the
// workgroup slots its span covers.  The consumers then start from two table reads instead of a search over all items
// (a binary search over 10^8 offsets is ~27 dependent loads; one per workgroup at its start paced these kernels).
constexpr int64_t FQ_UNIT = 16 * FQ_THREADS;                     // output positions per workgroup
__global__ void block_first_kernel(const int64_t* __restrict__ offsets, const int64_t n, int64_t* __restrict__ first) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int64_t lo = offsets[r], hi = offsets[r + 1];
    for (int64_t k = (lo + FQ_UNIT - 1) / FQ_UNIT; k * FQ_UNIT < hi; k++) first[k] = r;
}
// The offsets of the items a workgroup can touch, staged in LDS: s_off[j] = offsets[r0 + j], j = 0 .. r1-r0+1.
// Returns false (nothing staged) when they do not fit; the caller then searches in HBM.
constexpr int FQ_STAGE = 512;
__device__ __forceinline__ bool stage_offsets(const int64_t* __restrict__ offsets, const int64_t* __restrict__ first, const int64_t nblk,
                                              const int64_t n, int64_t& r0, int64_t& r1, int64_t* s_off) {
    r0 = first[blockIdx.x];
    r1 = ((int64_t)blockIdx.x + 1 < nblk) ? first[blockIdx.x + 1] : n - 1;
    const bool fits = (r1 - r0 + 2) <= FQ_STAGE;
    if (fits) for (int j = threadIdx.x; j < (int)(r1 - r0 + 2); j += FQ_THREADS) s_off[j] = offsets[r0 + j];
    __syncthreads();
    return fits;
}
__device__ __forceinline__ int lds_item_of(const int64_t* s_off, int hi, const int64_t b) {     // largest j in [0, hi] with s_off[j] <= b
    int lo = 0;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_off[mid] <= b) lo = mid; else hi = mid - 1; }
    return lo;
}
__global__ __launch_bounds__(FQ_THREADS)
void fq_pack_kernel(const uint8_t* __restrict__ t1, const int64_t* __restrict__ l1, const uint8_t* __restrict__ t2, const int64_t* __restrict__ l2,
                    const int ns, const int64_t* __restrict__ offsets, const int64_t n, const int64_t total,
                    const int64_t* __restrict__ first, const int64_t nblk, const int64_t* __restrict__ plim1, const int64_t* __restrict__ plim2,
                    uint32_t* __restrict__ codes, uint16_t* __restrict__ undef16) {
    __shared__ int64_t s_off[FQ_STAGE];
    const int64_t lim1 = *plim1 - 1, lim2 = plim2 ? *plim2 - 1 : 0;   // lines[4*records] - 1: the last indexed byte is a (possibly virtual) newline, never data
    const int64_t w0 = (int64_t)blockIdx.x * FQ_THREADS;
    const int64_t words = (total + 15) >> 4;
    int64_t r0, r1;
    const bool staged = stage_offsets(offsets, first, nblk, n, r0, r1, s_off);    // the reads this workgroup's 4096 bases can belong to
    const int64_t w = w0 + threadIdx.x;
    if (w >= words) return;
    int64_t b = 16 * w;
    int64_t r, rEnd, rBeg;
    if (staged) { const int j = lds_item_of(s_off, (int)(r1 - r0), b); r = r0 + j; rBeg = s_off[j]; rEnd = s_off[j + 1]; }
    else { r = read_of_base(offsets, r0, r1, b); rBeg = offsets[r]; rEnd = offsets[r + 1]; }
    auto src_of = [&](int64_t rd, const uint8_t*& t) -> int64_t {
        const bool second = (ns == 2) && (rd & 1);
        t = second ? t2 : t1;
        return (second ? l2 : l1)[4 * ((ns == 2) ? (rd >> 1) : rd) + 1];
    };
    const uint8_t* t; int64_t src = src_of(r, t) + (b - rBeg);
    // assemble the 16 bases run by run (a run = the part of one read inside this word; nearly always one or two runs),
    // one unaligned 16-byte load per run, then one SWAR encode
    U128 acc{0, 0};
    int filled = 0;
    const int want = (int)((total - b) < 16 ? (total - b) : 16);
    while (filled < want) {
        while (b >= rEnd) { r++; rEnd = offsets[r + 1]; src = src_of(r, t); }     // empty reads are stepped over
        const int c = (int)((rEnd - b) < (int64_t)(want - filled) ? (rEnd - b) : (int64_t)(want - filled));
        const U128 v = low_bytes(load16(t, src, (t == t1) ? lim1 : lim2), c);
        const U128 sh = shl_bytes(v, filled);
        acc.lo |= sh.lo; acc.hi |= sh.hi;
        filled += c; b += c; src += c;
    }
    uint32_t code, und;
    encode16(acc, code, und);
    if (want < 16) und |= 0xFFFFu << want;                       // past the end of the batch
    codes[w] = code;
    undef16[w] = (uint16_t)und;
}

// ---- writer: trimmed records back to FASTQ text (stream/FASTQ.java:474-490 toFASTQ: '@' id, bases, a bare '+', qualities)
// Output layout of one selected read: header[hl] '\n' bases[nl] '\n' '+' '\n' qualities[nl] '\n'  = hl + 2*nl + 5 bytes.
struct OutRec { int64_t h, s, q, mbit; int32_t hl, nl; };         // source offsets (trim applied), bit index of the first written base in
                                                                  // the batch's base mask (ktrim=n), lengths; 40 bytes
// 8 mask bits -> 8 bytes of 0xFF / 0x00
__device__ __forceinline__ uint64_t spread8(const uint32_t m8) {
    uint64_t t = ((uint64_t)m8 * 0x0101010101010101ULL) & 0x8040201008040201ULL;
    t = ((t + 0x7F7F7F7F7F7F7F7FULL) | t) & 0x8080808080808080ULL;
    return (t >> 7) * 0xFFULL;
}
__device__ __forceinline__ uint32_t mask16(const uint32_t* __restrict__ mask, const int64_t bit) {
    const int64_t w = bit >> 5; const int sh = (int)(bit & 31);
    const uint64_t two = ((uint64_t)mask[w + 1] << 32) | mask[w];
    return (uint32_t)(two >> sh) & 0xFFFFu;
}
// the masked bytes of v become `sym` (sym >= 0) or, for A-Z, lower case (sym < 0): BBDukProcessorS.java:2309-2320
__device__ __forceinline__ U128 apply_mask(U128 v, const uint32_t m16, const int sym) {
    const uint64_t bm[2] = {spread8(m16 & 0xFFu), spread8(m16 >> 8)};
    uint64_t x[2] = {v.lo, v.hi};
#pragma unroll
    for (int q = 0; q < 2; q++) {
        if (sym >= 0) x[q] = (x[q] & ~bm[q]) | (((uint64_t)(uint32_t)sym * 0x0101010101010101ULL) & bm[q]);
        else {
            const uint64_t lo7 = x[q] & 0x7F7F7F7F7F7F7F7FULL;
            const uint64_t upper = (lo7 + 0x3F3F3F3F3F3F3F3FULL) & ~(lo7 + 0x2525252525252525ULL) & ~x[q] & 0x8080808080808080ULL;   // 'A'..'Z'
            x[q] |= (upper >> 2) & bm[q];
        }
    }
    return U128{x[0], x[1]};
}
__global__ void fq_out_sizes_kernel(const uint8_t* __restrict__ t1, const int64_t* __restrict__ l1, const uint8_t* __restrict__ t2, const int64_t* __restrict__ l2,
                                    const int ns, const int64_t n, const int32_t* __restrict__ left, const int32_t* __restrict__ right,
                                    const uint8_t* __restrict__ flags, const int wantRemoved, const int64_t* __restrict__ baseOff,
                                    int32_t* __restrict__ sizes, OutRec* __restrict__ recs) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool second = (ns == 2) && (i & 1);
    const uint8_t* t = second ? t2 : t1;
    const int64_t* l = second ? l2 : l1;
    const int64_t r = (ns == 2) ? (i >> 1) : i;
    const bool sel = (((flags ? flags[i] : 0) & BBDUK_FLAG_REMOVED) != 0) == (wantRemoved != 0);
    int sz = 0;
    OutRec R; R.h = R.s = R.q = 0; R.mbit = 0; R.hl = R.nl = 0;
    if (sel) {
        const int64_t h = l[4 * r], sq = l[4 * r + 1];
        const int L = line_len(t, sq, l[4 * r + 2]);
        const int a = left ? max(left[i], 0) : 0, b = right ? max(right[i], 0) : 0;
        R.hl = line_len(t, h, sq);
        R.nl = max(L - a - b, 0);
        R.h = h; R.s = sq + a; R.q = l[4 * r + 3] + a;
        R.mbit = baseOff ? baseOff[i] + a : 0;
        sz = R.hl + 2 * R.nl + 5;
    }
    sizes[i] = sz;
    recs[i] = R;
}
// one thread per 16 output bytes: locate the record, then either one unaligned 16-byte load (the chunk lies inside a header,
// a bases or a qualities run) or sixteen single bytes (it crosses a seam); one aligned 16-byte store either way
__global__ __launch_bounds__(FQ_THREADS)
void fq_write_kernel(const uint8_t* __restrict__ t1, const uint8_t* __restrict__ t2, const int ns, const int64_t n,
                     const OutRec* __restrict__ recs, const int64_t* __restrict__ outOff, const int64_t total,
                     const int64_t* __restrict__ first, const int64_t nblk, const int64_t* __restrict__ plim1, const int64_t* __restrict__ plim2,
                     const uint32_t* __restrict__ mask, const int sym, uint8_t* __restrict__ out) {
    __shared__ int64_t s_off[FQ_STAGE];
    const int64_t lim1 = *plim1 - 1, lim2 = plim2 ? *plim2 - 1 : 0;
    const int64_t c0 = (int64_t)blockIdx.x * FQ_THREADS;
    int64_t r0, r1;
    const bool staged = stage_offsets(outOff, first, nblk, n, r0, r1, s_off);
    const int64_t p = 16 * (c0 + threadIdx.x);
    if (p >= total) return;
    int64_t i; int rel;
    if (staged) { const int j = lds_item_of(s_off, (int)(r1 - r0), p); i = r0 + j; rel = (int)(p - s_off[j]); }
    else { i = read_of_base(outOff, r0, r1, p); rel = (int)(p - outOff[i]); }
    OutRec R = recs[i];
    const uint8_t* t = ((ns == 2) && (i & 1)) ? t2 : t1;
    // assemble the 16 output bytes run by run: header, '\n', bases, "\n+\n", qualities, '\n', next record ...
    // (a chunk inside one run -- the common case -- is a single unaligned 16-byte load)
    U128 acc{0, 0};
    int filled = 0;
    const int want = (int)((total - p) < 16 ? (total - p) : 16);
    int size = R.hl + 2 * R.nl + 5;
    while (filled < want) {
        if (rel >= size) {                            // next selected record (unselected ones own no output byte)
            rel -= size;
            do { i++; R = recs[i]; } while (outOff[i + 1] == outOff[i]);
            size = R.hl + 2 * R.nl + 5;
            t = ((ns == 2) && (i & 1)) ? t2 : t1;
        }
        const int room = want - filled;
        int c; U128 v;
        const int64_t lim = (t == t1) ? lim1 : lim2;
        if (rel < R.hl) { c = min(R.hl - rel, room); v = load16(t, R.h + rel, lim); }
        else if (rel == R.hl) { c = 1; v = U128{0x0AULL, 0}; }
        else if (rel < R.hl + 1 + R.nl) {
            const int x = rel - R.hl - 1; c = min(R.nl - x, room); v = load16(t, R.s + x, lim);
            if (mask) { const uint32_t m16 = mask16(mask, R.mbit + x); if (m16) v = apply_mask(v, m16, sym); }
        }
        else if (rel < R.hl + 4 + R.nl) { const int x = rel - (R.hl + 1 + R.nl); c = min(3 - x, room); v = U128{0x0A2B0AULL >> (8 * x), 0}; }
        else if (rel < R.hl + 4 + 2 * R.nl) {
            const int x = rel - (R.hl + 4 + R.nl); c = min(R.nl - x, room); v = load16(t, R.q + x, lim);
            if (mask && sym == 'N') { const uint32_t m16 = mask16(mask, R.mbit + x); if (m16) v = apply_mask(v, m16, '!'); }    // quals[i]=0 (:2318)
        }
        else { c = 1; v = U128{0x0AULL, 0}; }
        const U128 sh = shl_bytes(low_bytes(v, c), filled);
        acc.lo |= sh.lo; acc.hi |= sh.hi;
        filled += c; rel += c;
    }
    if (want == 16) { uint4 o = make_uint4((uint32_t)acc.lo, (uint32_t)(acc.lo >> 32), (uint32_t)acc.hi, (uint32_t)(acc.hi >> 32)); *reinterpret_cast<uint4*>(out + p) = o; }
    else for (int j = 0; j < want; j++) out[p + j] = (uint8_t)((j < 8 ? acc.lo >> (8 * j) : acc.hi >> (8 * (j - 8))) & 0xFF);
}

// Scratch for the ingest / write calls: per host thread and device, grown on demand and kept.  hipMalloc / hipFree
// synchronise the whole device, which would serialise callers that pipeline chunks from several host threads (one stream
// each: the H2D of one chunk then overlaps the kernels and the D2H of another).
struct ScratchPool {
    static constexpr int SLOTS = 8;
    void* p[SLOTS] = {}; size_t cap[SLOTS] = {}; int device = -1;
    void* get(int slot, size_t bytes, int dev) {
        if (dev != device) { drop(); device = dev; }
        if (cap[slot] < bytes) {
            if (p[slot]) hipFree(p[slot]);
            p[slot] = nullptr; cap[slot] = 0;
            const size_t want = bytes + bytes / 8 + 256;
            if (hipMalloc(&p[slot], want) != hipSuccess) return nullptr;
            cap[slot] = want;
        }
        return p[slot];
    }
    void drop() { for (int i = 0; i < SLOTS; i++) { if (p[i]) hipFree(p[i]); p[i] = nullptr; cap[i] = 0; } }
    ~ScratchPool() { /* the runtime may already be gone at thread exit: leave the buffers to process teardown */ }
};
thread_local ScratchPool g_scratch;

}  // namespace

extern "C" int bbduk_fastq_ingest_device(const uint8_t* d_text1, int64_t nbytes1, const uint8_t* d_text2, int64_t nbytes2, int32_t is_final,
                                         int64_t max_reads, int64_t max_bases, int64_t* d_lines1, int64_t* d_lines2,
                                         int64_t* d_offsets, uint32_t* d_codes, uint32_t* d_undef,
                                         int32_t device, void* stream, bbduk_fastq_result* out) {
    if (!out) return BBDUK_ERR_ARG;
    out->n_reads = 0; out->total_bases = 0; out->consumed1 = 0; out->consumed2 = 0; out->first_bad_read = -1;
    const int ns = d_text2 ? 2 : 1;
    if ((!d_text1 && nbytes1 > 0) || (d_text2 == nullptr && nbytes2 > 0) || nbytes1 < 0 || nbytes2 < 0 || max_reads < 0 || max_bases < 0 || !d_lines1 || (ns == 2 && !d_lines2) || !d_offsets || !d_codes || !d_undef) return BBDUK_ERR_ARG;
    if (((uintptr_t)d_codes & 15) || ((uintptr_t)d_undef & 3)) return BBDUK_ERR_ARG;
    if (ns == 2 && (max_reads & 1)) max_reads--;
    if (hipSetDevice(device) != hipSuccess) return BBDUK_ERR_DEVICE;
    hipStream_t st = (hipStream_t)stream;
    const int64_t recCap = max_reads / ns;                       // records per text
    const uint8_t* texts[2] = {d_text1, d_text2}; const int64_t nbytes[2] = {nbytes1, nbytes2}; int64_t* lines[2] = {d_lines1, d_lines2};
    int64_t nrec[2] = {0, 0};
    auto release = [&]() {};                                      // scratch stays with the calling thread (g_scratch)
    const int64_t nbMax = (std::max(nbytes1, nbytes2) + FQ_BLOCK_BYTES - 1) / FQ_BLOCK_BYTES;
    const int64_t sbMax = (max_reads + SC_BLOCK - 1) / SC_BLOCK;
    int64_t* d_sums = (int64_t*)g_scratch.get(0, (size_t)(std::max(nbMax, sbMax) + 2) * 8, device);
    int32_t* d_lens = (int32_t*)g_scratch.get(1, (size_t)(max_reads + 1) * 4, device);
    unsigned long long* d_bad = (unsigned long long*)g_scratch.get(2, 8, device);
    if (!d_sums || !d_lens || !d_bad) return BBDUK_ERR_NOMEM;
    for (int s = 0; s < ns; s++) {
        const int64_t nb = (nbytes[s] + FQ_BLOCK_BYTES - 1) / FQ_BLOCK_BYTES;
        int64_t nl = 0;
        if (nb > 0) {
            fq_count_kernel<<<dim3((unsigned)nb), dim3(FQ_THREADS), 0, st>>>(texts[s], nbytes[s], d_sums);
            scan_sums_kernel<<<dim3(1), dim3(1024), 0, st>>>(d_sums, nb);
            fq_lines_kernel<<<dim3((unsigned)nb), dim3(FQ_THREADS), 0, st>>>(texts[s], nbytes[s], d_sums, lines[s], 4 * recCap);
            uint8_t last = '\n';
            if (hipMemcpyAsync(&nl, d_sums + nb, 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
                hipMemcpyAsync(&last, texts[s] + nbytes[s] - 1, 1, hipMemcpyDeviceToHost, st) != hipSuccess ||
                hipStreamSynchronize(st) != hipSuccess) { release(); return BBDUK_ERR_DEVICE; }
            if (is_final && last != '\n') {                       // ByteFile.nextLine returns an unterminated last line too
                nl++;
                const int64_t virt = nbytes[s] + 1;               // as if a '\n' sat at nbytes
                if (nl <= 4 * recCap && hipMemcpyAsync(lines[s] + nl, &virt, 8, hipMemcpyHostToDevice, st) != hipSuccess) { release(); return BBDUK_ERR_DEVICE; }
                if (hipStreamSynchronize(st) != hipSuccess) { release(); return BBDUK_ERR_DEVICE; }
            }
        } else {
            const int64_t z = 0;
            if (hipMemcpyAsync(lines[s], &z, 8, hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { release(); return BBDUK_ERR_DEVICE; }
        }
        nrec[s] = std::min(nl / 4, recCap);
    }
    const int64_t rec = ns == 2 ? std::min(nrec[0], nrec[1]) : nrec[0];
    const int64_t n = rec * ns;
    auto consumed = [&](int s, int64_t* dst) -> int {
        if (rec == 0) { *dst = 0; return BBDUK_OK; }
        int64_t v = 0;
        if (hipMemcpyAsync(&v, lines[s] + 4 * rec, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return BBDUK_ERR_DEVICE;
        *dst = std::min(v, nbytes[s]);                           // the virtual newline of an unterminated last line
        return BBDUK_OK;
    };
    if (consumed(0, &out->consumed1) != BBDUK_OK || (ns == 2 && consumed(1, &out->consumed2) != BBDUK_OK)) { release(); return BBDUK_ERR_DEVICE; }
    out->n_reads = n;
    if (n == 0) { const int64_t z = 0; hipMemcpyAsync(d_offsets, &z, 8, hipMemcpyHostToDevice, st); hipStreamSynchronize(st); release(); return BBDUK_OK; }
    const unsigned long long none = ~0ULL;
    hipMemcpyAsync(d_bad, &none, 8, hipMemcpyHostToDevice, st);
    fq_records_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(d_text1, d_lines1, d_text2, d_lines2, ns, n, d_lens, d_bad);
    const int64_t sb = (n + 1 + SC_BLOCK - 1) / SC_BLOCK;       // out[n] is written by the thread that owns index n
    block_sum_kernel<<<dim3((unsigned)sb), dim3(FQ_THREADS), 0, st>>>(d_lens, n, d_sums);
    scan_sums_kernel<<<dim3(1), dim3(1024), 0, st>>>(d_sums, sb);
    scan_final_kernel<<<dim3((unsigned)sb), dim3(FQ_THREADS), 0, st>>>(d_lens, n, d_sums, d_offsets);
    unsigned long long bad = none; int64_t total = 0;
    if (hipMemcpyAsync(&bad, d_bad, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipMemcpyAsync(&total, d_offsets + n, 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) { release(); return BBDUK_ERR_DEVICE; }
    if (bad != none) { out->first_bad_read = (int64_t)bad; release(); return BBDUK_ERR_FORMAT; }
    out->total_bases = total;
    if (total > max_bases) { release(); return BBDUK_ERR_ARG; }
    if (total > 0) {
        const int64_t words = (total + 15) >> 4;
        const int64_t nblk = (words + FQ_THREADS - 1) / FQ_THREADS;
        int64_t* d_first = (int64_t*)g_scratch.get(3, (size_t)(nblk + 1) * 8, device);
        if (!d_first) return BBDUK_ERR_NOMEM;
        block_first_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(d_offsets, n, d_first);
        fq_pack_kernel<<<dim3((unsigned)nblk), dim3(FQ_THREADS), 0, st>>>(
            d_text1, d_lines1, d_text2, d_lines2, ns, d_offsets, n, total, d_first, nblk, d_lines1 + 4 * rec, ns == 2 ? d_lines2 + 4 * rec : nullptr,
            d_codes, reinterpret_cast<uint16_t*>(d_undef));
        if ((words & 1) != 0) {                                   // the upper half of the last undefined word: past the end
            const uint16_t ones = 0xFFFFu;
            hipMemcpyAsync(reinterpret_cast<uint16_t*>(d_undef) + words, &ones, 2, hipMemcpyHostToDevice, st);
        }
    }
    const hipError_t e = hipStreamSynchronize(st);
    release();
    if (e != hipSuccess || hipGetLastError() != hipSuccess) return BBDUK_ERR_DEVICE;
    return BBDUK_OK;
}

// Writes the selected reads (want_removed == 0: those without BBDUK_FLAG_REMOVED; != 0: those with it), trimmed by
// d_left[i] / d_right[i] bases (either may be NULL = 0), as FASTQ text into d_out, in input order (mates stay adjacent).
// With d_mask (one bit per base of the batch, as bbduk_kmask_batch* return it) and d_base_offsets the masked bases are written as
// `symbol` (their qualities as '!' when the symbol is 'N') or, symbol < 0, in lower case: BBDukProcessorS.java:2309-2320.
extern "C" int bbduk_fastq_write_masked_device(const uint8_t* d_text1, const int64_t* d_lines1, const uint8_t* d_text2, const int64_t* d_lines2,
                                               int64_t n, const int32_t* d_left, const int32_t* d_right, const uint8_t* d_flags, int32_t want_removed,
                                               const int64_t* d_base_offsets, const uint32_t* d_mask, int32_t symbol,
                                               uint8_t* d_out, int64_t cap_out, int32_t device, void* stream, int64_t* out_bytes) {
    if (!out_bytes) return BBDUK_ERR_ARG;
    if ((d_mask != nullptr) != (d_base_offsets != nullptr) || symbol > 255) return BBDUK_ERR_ARG;
    *out_bytes = 0;
    const int ns = d_text2 ? 2 : 1;
    if (n < 0 || cap_out < 0 || (n > 0 && (!d_text1 || !d_lines1 || (ns == 2 && !d_lines2) || !d_out))) return BBDUK_ERR_ARG;
    if (n == 0) return BBDUK_OK;
    if (hipSetDevice(device) != hipSuccess) return BBDUK_ERR_DEVICE;
    hipStream_t st = (hipStream_t)stream;
    if ((uintptr_t)d_out & 15) return BBDUK_ERR_ARG;
    auto release = [&]() {};
    const int64_t sb = (n + 1 + SC_BLOCK - 1) / SC_BLOCK;
    int32_t* d_sizes = (int32_t*)g_scratch.get(4, (size_t)(n + 1) * 4, device);
    int64_t* d_sums = (int64_t*)g_scratch.get(0, (size_t)(sb + 2) * 8, device);
    int64_t* d_off = (int64_t*)g_scratch.get(5, (size_t)(n + 1) * 8, device);
    OutRec* d_recs = (OutRec*)g_scratch.get(6, (size_t)n * sizeof(OutRec), device);
    if (!d_sizes || !d_sums || !d_off || !d_recs) return BBDUK_ERR_NOMEM;
    fq_out_sizes_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(d_text1, d_lines1, d_text2, d_lines2, ns, n, d_left, d_right, d_flags, want_removed, d_base_offsets, d_sizes, d_recs);
    block_sum_kernel<<<dim3((unsigned)sb), dim3(FQ_THREADS), 0, st>>>(d_sizes, n, d_sums);
    scan_sums_kernel<<<dim3(1), dim3(1024), 0, st>>>(d_sums, sb);
    scan_final_kernel<<<dim3((unsigned)sb), dim3(FQ_THREADS), 0, st>>>(d_sizes, n, d_sums, d_off);
    int64_t total = 0;
    if (hipMemcpyAsync(&total, d_off + n, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { release(); return BBDUK_ERR_DEVICE; }
    *out_bytes = total;
    if (total > cap_out) { release(); return BBDUK_ERR_ARG; }
    if (total > 0) {
        const int64_t chunks = (total + 15) >> 4;
        const int64_t nblk = (chunks + FQ_THREADS - 1) / FQ_THREADS;
        int64_t* d_first = (int64_t*)g_scratch.get(3, (size_t)(nblk + 1) * 8, device);
        if (!d_first) return BBDUK_ERR_NOMEM;
        block_first_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(d_off, n, d_first);
        const int64_t rec = n / ns;
        fq_write_kernel<<<dim3((unsigned)nblk), dim3(FQ_THREADS), 0, st>>>(d_text1, d_text2, ns, n, d_recs, d_off, total, d_first, nblk,
                                                                           d_lines1 + 4 * rec, ns == 2 ? d_lines2 + 4 * rec : nullptr, d_mask, (int)symbol, d_out);
    }
    const hipError_t e = hipStreamSynchronize(st);
    release();
    if (e != hipSuccess || hipGetLastError() != hipSuccess) return BBDUK_ERR_DEVICE;
    return BBDUK_OK;
}

extern "C" int bbduk_fastq_write_device(const uint8_t* d_text1, const int64_t* d_lines1, const uint8_t* d_text2, const int64_t* d_lines2,
                                        int64_t n, const int32_t* d_left, const int32_t* d_right, const uint8_t* d_flags, int32_t want_removed,
                                        uint8_t* d_out, int64_t cap_out, int32_t device, void* stream, int64_t* out_bytes) {
    return bbduk_fastq_write_masked_device(d_text1, d_lines1, d_text2, d_lines2, n, d_left, d_right, d_flags, want_removed, nullptr, nullptr, 0,
                                           d_out, cap_out, device, stream, out_bytes);
}

// ---- memory helpers for callers without HIP bindings
extern "C" int bbduk_device_malloc(int32_t device, int64_t bytes, void** out) {
    if (!out || bytes < 0) return BBDUK_ERR_ARG;
    *out = nullptr;
    if (hipSetDevice(device) != hipSuccess) return BBDUK_ERR_DEVICE;
    return hipMalloc(out, (size_t)std::max<int64_t>(bytes, 16)) == hipSuccess ? BBDUK_OK : BBDUK_ERR_NOMEM;
}
extern "C" int bbduk_device_free(int32_t device, void* p) {
    if (hipSetDevice(device) != hipSuccess) return BBDUK_ERR_DEVICE;
    return hipFree(p) == hipSuccess ? BBDUK_OK : BBDUK_ERR_DEVICE;
}
extern "C" int bbduk_pinned_malloc(int64_t bytes, void** out) {
    if (!out || bytes < 0) return BBDUK_ERR_ARG;
    *out = nullptr;
    return hipHostMalloc(out, (size_t)std::max<int64_t>(bytes, 16), hipHostMallocDefault) == hipSuccess ? BBDUK_OK : BBDUK_ERR_NOMEM;
}
extern "C" int bbduk_pinned_free(void* p) { return hipHostFree(p) == hipSuccess ? BBDUK_OK : BBDUK_ERR_DEVICE; }
extern "C" int bbduk_copy_to_device(int32_t device, void* d_dst, const void* src, int64_t bytes, void* stream) {
    if (bytes < 0 || (bytes > 0 && (!d_dst || !src))) return BBDUK_ERR_ARG;
    if (bytes == 0) return BBDUK_OK;
    if (hipSetDevice(device) != hipSuccess) return BBDUK_ERR_DEVICE;
    if (hipMemcpyAsync(d_dst, src, (size_t)bytes, hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess) return BBDUK_ERR_DEVICE;
    return hipStreamSynchronize((hipStream_t)stream) == hipSuccess ? BBDUK_OK : BBDUK_ERR_DEVICE;
}
extern "C" int bbduk_copy_from_device(int32_t device, void* dst, const void* d_src, int64_t bytes, void* stream) {
    if (bytes < 0 || (bytes > 0 && (!dst || !d_src))) return BBDUK_ERR_ARG;
    if (bytes == 0) return BBDUK_OK;
    if (hipSetDevice(device) != hipSuccess) return BBDUK_ERR_DEVICE;
    if (hipMemcpyAsync(dst, d_src, (size_t)bytes, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return BBDUK_ERR_DEVICE;
    return hipStreamSynchronize((hipStream_t)stream) == hipSuccess ? BBDUK_OK : BBDUK_ERR_DEVICE;
}
// The asynchronous forms (round 5, the pipelined deviceingest=t path of bbduk_cli): a stream of the caller's own that does not synchronise with
// the default stream, copies that return at once, and the wait.
extern "C" int bbduk_stream_create(int32_t device, void** out) {
    if (!out) return BBDUK_ERR_ARG;
    *out = nullptr;
    if (hipSetDevice(device) != hipSuccess) return BBDUK_ERR_DEVICE;
    hipStream_t st;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return BBDUK_ERR_DEVICE;
    *out = (void*)st;
    return BBDUK_OK;
}
extern "C" int bbduk_stream_destroy(int32_t device, void* stream) {
    if (!stream) return BBDUK_OK;
    if (hipSetDevice(device) != hipSuccess) return BBDUK_ERR_DEVICE;
    return hipStreamDestroy((hipStream_t)stream) == hipSuccess ? BBDUK_OK : BBDUK_ERR_DEVICE;
}
extern "C" int bbduk_stream_synchronize(int32_t device, void* stream) {
    if (hipSetDevice(device) != hipSuccess) return BBDUK_ERR_DEVICE;
    return hipStreamSynchronize((hipStream_t)stream) == hipSuccess ? BBDUK_OK : BBDUK_ERR_DEVICE;
}
extern "C" int bbduk_copy_async(int32_t device, void* dst, const void* src, int64_t bytes, int32_t kind, void* stream) {
    if (bytes < 0 || (bytes > 0 && (!dst || !src)) || kind < 0 || kind > 2) return BBDUK_ERR_ARG;
    if (bytes == 0) return BBDUK_OK;
    if (hipSetDevice(device) != hipSuccess) return BBDUK_ERR_DEVICE;
    const hipMemcpyKind k = kind == 0 ? hipMemcpyHostToDevice : kind == 1 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    return hipMemcpyAsync(dst, src, (size_t)bytes, k, (hipStream_t)stream) == hipSuccess ? BBDUK_OK : BBDUK_ERR_DEVICE;
}
extern "C" int bbduk_device_memset(int32_t device, void* d_dst, int32_t value, int64_t bytes, void* stream) {
    if (bytes < 0 || (bytes > 0 && !d_dst)) return BBDUK_ERR_ARG;
    if (bytes == 0) return BBDUK_OK;
    if (hipSetDevice(device) != hipSuccess) return BBDUK_ERR_DEVICE;
    if (hipMemsetAsync(d_dst, value, (size_t)bytes, (hipStream_t)stream) != hipSuccess) return BBDUK_ERR_DEVICE;
    return hipStreamSynchronize((hipStream_t)stream) == hipSuccess ? BBDUK_OK : BBDUK_ERR_DEVICE;
}
