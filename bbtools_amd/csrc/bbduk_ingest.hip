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

// 64 bytes of text starting at byte `a` (a multiple of 16; the buffer is 16-byte aligned), zero past nbytes
__device__ __forceinline__ void load64(const uint8_t* __restrict__ text, const int64_t a, const int64_t nbytes, uint32_t* w) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int64_t p = a + 16 * q;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (p + 16 <= nbytes) v = *reinterpret_cast<const uint4*>(text + p);
        else if (p < nbytes) {
            uint32_t t[4] = {0, 0, 0, 0};
            for (int b = 0; b < 16 && p + b < nbytes; b++) t[b >> 2] |= (uint32_t)text[p + b] << (8 * (b & 3));
            v = make_uint4(t[0], t[1], t[2], t[3]);
        }
        w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
    }
}

__device__ __forceinline__ int block_exclusive_scan(int v, int* s_tmp, int& total) {      // FQ_THREADS threads
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o); if (lane >= o) x += y; }
    if (lane == 63) s_tmp[wave] = x;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < FQ_THREADS / 64; w++) { const int c = s_tmp[w]; if (w < wave) base += c; tot += c; }
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
    __shared__ int s_tmp[FQ_THREADS / 64];
    const int64_t a = (int64_t)blockIdx.x * SC_BLOCK + (int64_t)threadIdx.x * SC_PER_THREAD;
    int c = 0;
    for (int q = 0; q < SC_PER_THREAD; q++) if (a + q < n) c += in[a + q];
    int total;
    block_exclusive_scan(c, s_tmp, total);          // reads are < 2^31 bases per 2048 reads: BBDUK_MAX_READ_LEN * 2048 fits
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}
__global__ __launch_bounds__(FQ_THREADS)
void scan_final_kernel(const int32_t* __restrict__ in, const int64_t n, const int64_t* __restrict__ sums, int64_t* __restrict__ out) {
    __shared__ int s_tmp[FQ_THREADS / 64];
    const int64_t a = (int64_t)blockIdx.x * SC_BLOCK + (int64_t)threadIdx.x * SC_PER_THREAD;
    int v[SC_PER_THREAD]; int c = 0;
    for (int q = 0; q < SC_PER_THREAD; q++) { v[q] = (a + q < n) ? in[a + q] : 0; c += v[q]; }
    int total;
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
__global__ __launch_bounds__(FQ_THREADS)
void fq_pack_kernel(const uint8_t* __restrict__ t1, const int64_t* __restrict__ l1, const uint8_t* __restrict__ t2, const int64_t* __restrict__ l2,
                    const int ns, const int64_t* __restrict__ offsets, const int64_t n, const int64_t total,
                    uint32_t* __restrict__ codes, uint16_t* __restrict__ undef16) {
    __shared__ int64_t s_range[2];
    const int64_t w0 = (int64_t)blockIdx.x * FQ_THREADS;
    const int64_t words = (total + 15) >> 4;
    if (threadIdx.x == 0) {                           // the reads this workgroup's 4096 bases can belong to
        s_range[0] = read_of_base(offsets, 0, n - 1, (16 * w0 < total - 1 ? 16 * w0 : total - 1));
        s_range[1] = read_of_base(offsets, 0, n - 1, (16 * (w0 + FQ_THREADS) - 1 < total - 1 ? 16 * (w0 + FQ_THREADS) - 1 : total - 1));
    }
    __syncthreads();
    const int64_t w = w0 + threadIdx.x;
    if (w >= words) return;
    int64_t b = 16 * w;
    int64_t r = read_of_base(offsets, s_range[0], s_range[1], b);
    int64_t rEnd = offsets[r + 1];
    auto src_of = [&](int64_t rd, const uint8_t*& t) -> int64_t {
        const bool second = (ns == 2) && (rd & 1);
        t = second ? t2 : t1;
        return (second ? l2 : l1)[4 * ((ns == 2) ? (rd >> 1) : rd) + 1];
    };
    const uint8_t* t; int64_t src = src_of(r, t) + (b - offsets[r]);
    uint32_t code = 0, und = 0;
    for (int j = 0; j < 16; j++, b++) {
        if (b >= total) { und |= 0xFFFFu << j; break; }
        while (b >= rEnd) { r++; rEnd = offsets[r + 1]; src = src_of(r, t); }     // empty reads are stepped over
        const uint32_t ch = t[src++] | 0x20u;
        const int c = ch == 'a' ? 0 : ch == 'c' ? 1 : ch == 'g' ? 2 : (ch == 't' || ch == 'u') ? 3 : -1;   // AminoAcid.java:1284-1298
        if (c < 0) und |= 1u << j; else code |= (uint32_t)c << (2 * j);
    }
    codes[w] = code;
    undef16[w] = (uint16_t)und;
}

// ---- writer: trimmed records back to FASTQ text (stream/FASTQ.java:474-490 toFASTQ: '@' id, bases, a bare '+', qualities)
// sizes[i] = bytes read i occupies in the output (0 if it is not selected)
__global__ void fq_out_sizes_kernel(const uint8_t* __restrict__ t1, const int64_t* __restrict__ l1, const uint8_t* __restrict__ t2, const int64_t* __restrict__ l2,
                                    const int ns, const int64_t n, const int32_t* __restrict__ left, const int32_t* __restrict__ right,
                                    const uint8_t* __restrict__ flags, const int wantRemoved, int32_t* __restrict__ sizes) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool second = (ns == 2) && (i & 1);
    const uint8_t* t = second ? t2 : t1;
    const int64_t* l = second ? l2 : l1;
    const int64_t r = (ns == 2) ? (i >> 1) : i;
    const bool sel = (((flags ? flags[i] : 0) & BBDUK_FLAG_REMOVED) != 0) == (wantRemoved != 0);
    int sz = 0;
    if (sel) {
        const int hl = line_len(t, l[4 * r], l[4 * r + 1]);
        const int L = line_len(t, l[4 * r + 1], l[4 * r + 2]);
        const int a = left ? max(left[i], 0) : 0, b = right ? max(right[i], 0) : 0;
        const int nl = max(L - a - b, 0);
        sz = hl + 1 + nl + 3 + nl + 1;
    }
    sizes[i] = sz;
}
// one wave per selected read: header line, bases[left, L-right), "+", qualities[left, L-right)
__global__ __launch_bounds__(256)
void fq_write_kernel(const uint8_t* __restrict__ t1, const int64_t* __restrict__ l1, const uint8_t* __restrict__ t2, const int64_t* __restrict__ l2,
                     const int ns, const int64_t n, const int32_t* __restrict__ left, const int32_t* __restrict__ right,
                     const int64_t* __restrict__ outOff, uint8_t* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t wavesPerGrid = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t i = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); i < n; i += wavesPerGrid) {
        const int64_t o0 = outOff[i], o1 = outOff[i + 1];
        if (o1 == o0) continue;
        const bool second = (ns == 2) && (i & 1);
        const uint8_t* t = second ? t2 : t1;
        const int64_t* l = second ? l2 : l1;
        const int64_t r = (ns == 2) ? (i >> 1) : i;
        const int64_t h = l[4 * r], s = l[4 * r + 1], q = l[4 * r + 3];
        const int hl = line_len(t, h, s);
        const int L = line_len(t, s, l[4 * r + 2]);
        const int a = left ? max(left[i], 0) : 0, b = right ? max(right[i], 0) : 0;
        const int nl = max(L - a - b, 0);
        uint8_t* dst = out + o0;
        for (int j = lane; j < hl; j += 64) dst[j] = t[h + j];
        for (int j = lane; j < nl; j += 64) { dst[hl + 1 + j] = t[s + a + j]; dst[hl + 1 + nl + 3 + j] = t[q + a + j]; }
        if (lane == 0) { dst[hl] = '\n'; dst[hl + 1 + nl] = '\n'; dst[hl + 1 + nl + 1] = '+'; dst[hl + 1 + nl + 2] = '\n'; dst[hl + 1 + nl + 3 + nl] = '\n'; }
    }
}

}  // namespace

extern "C" int bbduk_fastq_ingest_device(const uint8_t* d_text1, int64_t nbytes1, const uint8_t* d_text2, int64_t nbytes2, int32_t is_final,
                                         int64_t max_reads, int64_t max_bases, int64_t* d_lines1, int64_t* d_lines2,
                                         int64_t* d_offsets, uint32_t* d_codes, uint32_t* d_undef,
                                         int32_t device, void* stream, bbduk_fastq_result* out) {
    if (!out) return BBDUK_ERR_ARG;
    out->n_reads = 0; out->total_bases = 0; out->consumed1 = 0; out->consumed2 = 0; out->first_bad_read = -1;
    const int ns = d_text2 ? 2 : 1;
    if ((!d_text1 && nbytes1 > 0) || (d_text2 == nullptr && nbytes2 > 0) || nbytes1 < 0 || nbytes2 < 0 || max_reads < 0 || max_bases < 0 || !d_lines1 || (ns == 2 && !d_lines2) || !d_offsets || !d_codes || !d_undef) return BBDUK_ERR_ARG;
    if (((uintptr_t)d_text1 & 15) || ((uintptr_t)d_text2 & 15) || ((uintptr_t)d_codes & 15) || ((uintptr_t)d_undef & 3)) return BBDUK_ERR_ARG;
    if (ns == 2 && (max_reads & 1)) max_reads--;
    if (hipSetDevice(device) != hipSuccess) return BBDUK_ERR_DEVICE;
    hipStream_t st = (hipStream_t)stream;
    const int64_t recCap = max_reads / ns;                       // records per text
    const uint8_t* texts[2] = {d_text1, d_text2}; const int64_t nbytes[2] = {nbytes1, nbytes2}; int64_t* lines[2] = {d_lines1, d_lines2};
    int64_t nrec[2] = {0, 0};
    int64_t* d_sums = nullptr; int32_t* d_lens = nullptr; unsigned long long* d_bad = nullptr;
    auto release = [&]() { hipFree(d_sums); hipFree(d_lens); hipFree(d_bad); };
    const int64_t nbMax = (std::max(nbytes1, nbytes2) + FQ_BLOCK_BYTES - 1) / FQ_BLOCK_BYTES;
    const int64_t sbMax = (max_reads + SC_BLOCK - 1) / SC_BLOCK;
    if (hipMalloc(&d_sums, (size_t)(std::max(nbMax, sbMax) + 2) * 8) != hipSuccess || hipMalloc(&d_lens, (size_t)(max_reads + 1) * 4) != hipSuccess ||
        hipMalloc(&d_bad, 8) != hipSuccess) { release(); return BBDUK_ERR_NOMEM; }
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
        fq_pack_kernel<<<dim3((unsigned)((words + FQ_THREADS - 1) / FQ_THREADS)), dim3(FQ_THREADS), 0, st>>>(
            d_text1, d_lines1, d_text2, d_lines2, ns, d_offsets, n, total, d_codes, reinterpret_cast<uint16_t*>(d_undef));
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
extern "C" int bbduk_fastq_write_device(const uint8_t* d_text1, const int64_t* d_lines1, const uint8_t* d_text2, const int64_t* d_lines2,
                                        int64_t n, const int32_t* d_left, const int32_t* d_right, const uint8_t* d_flags, int32_t want_removed,
                                        uint8_t* d_out, int64_t cap_out, int32_t device, void* stream, int64_t* out_bytes) {
    if (!out_bytes) return BBDUK_ERR_ARG;
    *out_bytes = 0;
    const int ns = d_text2 ? 2 : 1;
    if (n < 0 || cap_out < 0 || (n > 0 && (!d_text1 || !d_lines1 || (ns == 2 && !d_lines2) || !d_out))) return BBDUK_ERR_ARG;
    if (n == 0) return BBDUK_OK;
    if (hipSetDevice(device) != hipSuccess) return BBDUK_ERR_DEVICE;
    hipStream_t st = (hipStream_t)stream;
    int32_t* d_sizes = nullptr; int64_t* d_sums = nullptr; int64_t* d_off = nullptr;
    auto release = [&]() { hipFree(d_sizes); hipFree(d_sums); hipFree(d_off); };
    const int64_t sb = (n + 1 + SC_BLOCK - 1) / SC_BLOCK;
    if (hipMalloc(&d_sizes, (size_t)(n + 1) * 4) != hipSuccess || hipMalloc(&d_sums, (size_t)(sb + 2) * 8) != hipSuccess ||
        hipMalloc(&d_off, (size_t)(n + 1) * 8) != hipSuccess) { release(); return BBDUK_ERR_NOMEM; }
    fq_out_sizes_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(d_text1, d_lines1, d_text2, d_lines2, ns, n, d_left, d_right, d_flags, want_removed, d_sizes);
    block_sum_kernel<<<dim3((unsigned)sb), dim3(FQ_THREADS), 0, st>>>(d_sizes, n, d_sums);
    scan_sums_kernel<<<dim3(1), dim3(1024), 0, st>>>(d_sums, sb);
    scan_final_kernel<<<dim3((unsigned)sb), dim3(FQ_THREADS), 0, st>>>(d_sizes, n, d_sums, d_off);
    int64_t total = 0;
    if (hipMemcpyAsync(&total, d_off + n, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { release(); return BBDUK_ERR_DEVICE; }
    *out_bytes = total;
    if (total > cap_out) { release(); return BBDUK_ERR_ARG; }
    if (total > 0) {
        const int64_t blocks = std::min<int64_t>((n + 3) / 4, 1 << 20);
        fq_write_kernel<<<dim3((unsigned)blocks), dim3(256), 0, st>>>(d_text1, d_lines1, d_text2, d_lines2, ns, n, d_left, d_right, d_off, d_out);
    }
    const hipError_t e = hipStreamSynchronize(st);
    release();
    if (e != hipSuccess || hipGetLastError() != hipSuccess) return BBDUK_ERR_DEVICE;
    return BBDUK_OK;
}
