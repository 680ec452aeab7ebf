// Line-layout simulator for the big layout (round 6): spilled keys, line requests per 150-base read and spill-bit candidates per
// read for   (a) 64-byte lines, a key in its pair of tag words (8 ways) or spilled -- the round-4/5 layout;
//            (b) 128-byte lines of two 64-byte halves: primary pair in one half, overflow pair in the sibling half, then spilled;
// under      (R) the plain sliding maximum over the W = H-m+1 gapped (m+m)-mers of a key (random minimizer, density 2/(W+1)), or
//            (M) mod-sampling: the position x of the smallest symmetric hash over the H-t+1 gapped (t+t)-mers picks candidate x mod W (needs W | H-t+1;
//                a window whose minimum is tied falls back to a plain key hash) -- density (floor((H-t)/W)+2)/(H-t+2).
// The crowding of a 10^10-key map (the same gapped-mer met at several reference positions) is kept at a small key count by folding the gapped-mers'
// identities into a space of  keys / (1e10 / 4^(2m))  values.   A measurement program for profiles/, not part of the library.
//   gcc -O2 -o /tmp/sim_lines profiles/sim_lines.c && /tmp/sim_lines <keys M> <m> <scheme R|M> <t> <lineBytes 64|128> <load> [emulate keys = 1e10] [spill classes = 4]
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static uint64_t rs = 88172645463325252ULL;
static inline uint64_t rnd(void) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs; }
static inline uint64_t mix(uint64_t z) { z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31); }

static int K = 31, H = 15, D = 16, M, W, T, scheme;
static uint64_t space;            // identities of the gapped-mers fold into this many values (0 = no folding)

// symmetric value of the gapped (n+n)-mer at offset p of the base array b (b[0..K)), n = M or T
static inline uint64_t gval(const uint8_t* b, int p, int n) {
    uint64_t f = 0, r = 0;
    for (int i = 0; i < n; i++) f = (f << 2) | b[p + i];
    for (int i = 0; i < n; i++) f = (f << 2) | b[p + D + i];
    // reverse complement of the gapped-mer: the right part reversed-complemented becomes the left part
    for (int i = n - 1; i >= 0; i--) r = (r << 2) | (3 - b[p + D + i]);
    for (int i = n - 1; i >= 0; i--) r = (r << 2) | (3 - b[p + i]);
    uint64_t id = f < r ? f : r;
    if (space && n == M) id = mix(id) % space;
    return mix(id * 2 + 1);
}
// line of the key whose bases are b[0..K) (middle base ignored); *tie = the mod-sampling minimum was tied
static uint64_t line_value(const uint8_t* b, int* tie) {
    *tie = 0;
    if (scheme == 'R') {
        uint64_t best = 0;
        for (int p = 0; p < W; p++) { uint64_t v = gval(b, p, M); if (v > best) best = v; }
        return best;
    }
    const int np = H - T + 1;
    uint64_t mn = ~0ULL; int x = 0, cnt = 0;
    for (int p = 0; p < np; p++) { uint64_t v = gval(b, p, T) >> 40; if (v < mn) { mn = v; x = p; cnt = 1; } else if (v == mn) cnt++; }      // 24-bit values: ties happen
    if (cnt > 1) { *tie = 1; uint64_t f = 0, r = 0; for (int i = 0; i < K; i++) { if (i == H) continue; f = f * 4 + b[i]; r = r * 4 + (3 - b[K - 1 - i]); } return mix(f > r ? f : r); }
    return gval(b, x % W, M);
}

int main(int argc, char** argv) {
    if (argc < 7) { fprintf(stderr, "usage: sim_lines <keys M> <m> <R|M> <t> <lineBytes> <load> [emulate]\n"); return 2; }
    const uint64_t N = (uint64_t)(atof(argv[1]) * 1e6);
    M = atoi(argv[2]); scheme = argv[3][0]; T = atoi(argv[4]);
    const int lineBytes = atoi(argv[5]); const double load = atof(argv[6]);
    const double emu = argc > 7 ? atof(argv[7]) : 1e10;
    const int NC = argc > 8 ? atoi(argv[8]) : 4;
    const int NST = argc > 9 ? atoi(argv[9]) : 2;                 // pairs a key may sit in (128-byte lines): its primary pair, pair ^ 4, pair ^ 2, pair ^ 6                  // spill classes per pair (4: the even word's lane tops; 8: both words')
    W = H - M + 1;
    if (scheme == 'M' && (H - T + 1) % W) { fprintf(stderr, "W must divide H-t+1\n"); return 2; }
    double ratio = emu; for (int i = 0; i < 2 * M; i++) ratio /= 4.0;
    space = emu > 0 ? (uint64_t)((double)N / ratio) : 0;
    const int pairsPerLine = lineBytes / 16, slots = lineBytes / 2;
    const uint64_t nlines = (uint64_t)((double)N / (slots * load));
    uint8_t* ref = malloc(N + K);
    for (uint64_t i = 0; i < N + K; i++) ref[i] = rnd() & 3;
    uint8_t* fill = calloc(nlines * pairsPerLine, 1);      // keys in each pair (<= 8)
    uint8_t* sbits = calloc(nlines * pairsPerLine, 1);     // 4 class bits: some key of this class left the pair
    uint64_t spilled = 0, toSibling = 0, ties = 0;
    uint32_t* lineLoad = calloc(nlines, 4);
    for (uint64_t i = 0; i < N; i++) {
        int tie; const uint64_t v = line_value(ref + i, &tie); ties += tie;
        uint64_t kh = 0; for (int j = 0; j < K; j++) if (j != H) kh = kh * 4 + ref[i + j];
        kh = mix(kh);
        const uint64_t line = (uint64_t)(((__uint128_t)mix(v) * nlines) >> 64);
        lineLoad[line]++;
        const int pr = kh % pairsPerLine, cls = (kh >> 8) % NC;
        static const int X[4] = {0, 4, 2, 6};
        int placed = 0;
        for (int st = 0; st < (lineBytes == 128 ? NST : 1); st++) {
            uint64_t a = line * pairsPerLine + (pr ^ X[st]);
            if (fill[a] < 8) { fill[a]++; if (st) toSibling++; placed = 1; break; }
            sbits[a] |= 1 << cls;
        }
        if (placed) continue;
        spilled++;
    }
    // absent reads
    const int R = 200000, L = 150;
    uint64_t lines = 0, cand1 = 0, cand2 = 0, wins = 0, extra = 0;
    uint8_t rd[160];
    for (int r = 0; r < R; r++) {
        for (int i = 0; i < L; i++) rd[i] = rnd() & 3;
        uint64_t prev = ~0ULL;
        for (int i = 0; i + K <= L; i++) {
            int tie; const uint64_t v = line_value(rd + i, &tie);
            uint64_t kh = 0; for (int j = 0; j < K; j++) if (j != H) kh = kh * 4 + rd[i + j];
            kh = mix(kh);
            const uint64_t line = (uint64_t)(((__uint128_t)mix(v) * nlines) >> 64);
            if (line != prev) { lines++; prev = line; }
            const int pr = kh % pairsPerLine, cls = (kh >> 8) % NC;
            wins++;
            {
                static const int X[4] = {0, 4, 2, 6};
                int st = 0;
                while (st < (lineBytes == 128 ? NST : 1) && (sbits[line * pairsPerLine + (pr ^ X[st])] & (1 << cls))) st++;
                if (st >= 1) cand1++;                             // looks beyond its primary pair
                extra += st > 1 ? st - 1 : 0;                     // pairs looked at beyond the sibling
                if (st == (lineBytes == 128 ? NST : 1)) cand2++;  // ... and ends in the secondary map
            }
        }
    }
    uint64_t over = 0; for (uint64_t l = 0; l < nlines; l++) if (lineLoad[l] > (uint32_t)slots) over += lineLoad[l] - slots;
    printf("{\"keys\": %llu, \"m\": %d, \"W\": %d, \"scheme\": \"%c\", \"t\": %d, \"line_bytes\": %d, \"load\": %.2f, \"emulates_keys\": %.0e, \"spilled_pct\": %.3f, "
           "\"to_sibling_pct\": %.3f, \"line_capacity_overflow_pct\": %.3f, \"tied_keys_pct\": %.3f, \"lines_per_read\": %.2f, \"spill_bit_windows_per_read\": %.3f, "
           "\"secondary_lookups_per_read\": %.3f, \"classes\": %d, \"stages\": %d, \"pairs_beyond_sibling_per_read\": %.3f}\n",
           (unsigned long long)N, M, W, scheme, T, lineBytes, load, emu, 100.0 * spilled / N, 100.0 * toSibling / N, 100.0 * over / N, 100.0 * ties / N,
           (double)lines / R, (double)cand1 / R, (double)cand2 / R, NC, NST, (double)extra / R);
    return 0;
}
