// synth.h -- deterministic synthetic 2x150bp read-pair model, shared bit-for-bit by the
// device generator (bbduk_synth_generate_device) and the host generator (bbduk_synth_generate_host).
//
// Counter-based: every base of every read is a pure function of (seed, pair index, stream, position), so
// any slice of the 100M / 1B-read workloads can be regenerated anywhere (GPU shard, CPU sample) without
// materialising the rest.  Model, following the reference's own recipe for adapter-trimming truth sets
// (docs/guides/AddAdaptersGuide.txt:14-27; jgi/AddAdapters.java): a fragment of `ins` ~ U[ins_min,ins_max]
// genome bases; r1 reads it forward, r2 reads its reverse complement; when ins < read_len the 3' tail is
// adapter read-through followed by random bases; substitutions inside adapter/contaminant bases; a small
// N rate everywhere; optionally a fraction of pairs is drawn from a contaminant sequence (phiX, kfilter).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define BB_HD __host__ __device__ __forceinline__
#else
#define BB_HD inline
#endif

struct bb_synth_dev {              // device/host-neutral copy of bbduk_synth_params (pointers valid where used)
    uint64_t seed;
    int32_t  read_len, ins_min, ins_max, adapter1_len, adapter2_len;
    uint32_t sub_rate_q32, n_rate_q32, contam_frac_q32;
    int64_t  contam_len;
    const uint8_t* adapter1;
    const uint8_t* adapter2;
    const uint8_t* contam;
};

BB_HD uint64_t bb_mix64(uint64_t z) {          // splitmix64 finaliser
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
BB_HD uint64_t bb_rand(uint64_t seed, uint64_t pair, uint32_t stream, uint32_t j) {
    return bb_mix64(bb_mix64(seed ^ (pair * 0xD1342543DE82EF95ULL)) + (((uint64_t)stream << 32) | j));
}
BB_HD uint8_t bb_code_to_base(uint32_t c) { return (uint8_t)("ACGT"[c & 3]); }
BB_HD int bb_base_to_code(uint8_t b) {         // -1 if not ACGT (upper case only; generator inputs are upper case)
    return b == 'A' ? 0 : b == 'C' ? 1 : b == 'G' ? 2 : b == 'T' ? 3 : -1;
}
BB_HD uint8_t bb_complement(uint8_t b) {
    int c = bb_base_to_code(b);
    return c < 0 ? b : bb_code_to_base(3 - (uint32_t)c);
}
// substitute with probability sub_rate: always to a different base
BB_HD uint8_t bb_maybe_sub(uint8_t b, uint64_t ev, uint32_t sub_rate_q32) {
    int c = bb_base_to_code(b);
    if (c >= 0 && (uint32_t)ev < sub_rate_q32) return bb_code_to_base((uint32_t)c + 1u + (uint32_t)((ev >> 32) % 3u));
    return b;
}

struct bb_pair_hdr { int32_t ins; int32_t is_contam; int32_t strand; int64_t pos; };

BB_HD bb_pair_hdr bb_synth_pair_header(const bb_synth_dev& sp, uint64_t pair) {
    bb_pair_hdr h;
    const uint64_t h0 = bb_rand(sp.seed, pair, 0, 0);
    const uint32_t span = (uint32_t)(sp.ins_max - sp.ins_min + 1);
    h.ins = sp.ins_min + (int32_t)((uint32_t)(h0 & 0xFFFFFFFFu) % span);
    h.is_contam = (sp.contam_len > 0 && (uint32_t)(h0 >> 32) < sp.contam_frac_q32) ? 1 : 0;
    h.strand = 0; h.pos = 0;
    if (h.is_contam) {
        if (sp.contam_len < h.ins) h.ins = (int32_t)sp.contam_len;
        const uint64_t h1 = bb_rand(sp.seed, pair, 0, 1);
        h.pos = (int64_t)((h1 >> 1) % (uint64_t)(sp.contam_len - h.ins + 1));
        h.strand = (int32_t)(h1 & 1);
    }
    return h;
}
// fragment base j (0 <= j < ins), in fragment orientation
BB_HD uint8_t bb_synth_frag_base(const bb_synth_dev& sp, uint64_t pair, const bb_pair_hdr& h, int32_t j) {
    const uint64_t ev = bb_rand(sp.seed, pair, 1, (uint32_t)j);
    if (h.is_contam) {
        uint8_t b = h.strand ? bb_complement(sp.contam[h.pos + h.ins - 1 - j]) : sp.contam[h.pos + j];
        return bb_maybe_sub(b, ev, sp.sub_rate_q32);
    }
    return bb_code_to_base((uint32_t)(ev >> 62));
}
// base j of mate `mate` (0 = r1, 1 = r2)
BB_HD uint8_t bb_synth_read_base(const bb_synth_dev& sp, uint64_t pair, const bb_pair_hdr& h, int32_t mate, int32_t j) {
    const uint64_t evn = bb_rand(sp.seed, pair, 4u + (uint32_t)mate, (uint32_t)j);
    if ((uint32_t)evn < sp.n_rate_q32) return (uint8_t)'N';
    if (j < h.ins) {
        return mate == 0 ? bb_synth_frag_base(sp, pair, h, j)
                         : bb_complement(bb_synth_frag_base(sp, pair, h, h.ins - 1 - j));
    }
    const int32_t a = j - h.ins;
    const uint64_t ev = bb_rand(sp.seed, pair, 2u + (uint32_t)mate, (uint32_t)j);
    const uint8_t* ad = mate == 0 ? sp.adapter1 : sp.adapter2;
    const int32_t alen = mate == 0 ? sp.adapter1_len : sp.adapter2_len;
    if (a < alen) return bb_maybe_sub(ad[a], ev, sp.sub_rate_q32);
    return bb_code_to_base((uint32_t)(ev >> 62));
}
