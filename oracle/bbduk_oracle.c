/*
 * bbduk_oracle.c -- CPU ORACLE (test infrastructure only; see bbduk_oracle.h).  PARITY UNPINNED.
 *
 * Loop-for-loop restatement, in plain C, of the reference's BBDuk k-mer matching path
 * (BBTools v40.02).  Every function cites the reference file:line it follows, relative to
 * /root/reference/current/ .  Nothing here is optimised: the sequential rolling loops, the
 * recursive mutate(), the 7-way key%7 HashArray1D with `extra`=60 no-wrap probing and a
 * victim overflow are kept as the reference has them so that the restatement can be read
 * side by side with the Java.
 */
#include "bbduk_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <pthread.h>

/* ------------------------------------------------------------------------------------------ */
/* dna/AminoAcid.java:1284-1311 -- baseToNumber (-1 undefined), baseToNumber0, baseToComplementNumber0 */
static int8_t T_num[128], T_num0[128], T_cnum0[128];
static int tables_ready = 0;
static void init_tables(void) {
    if (tables_ready) return;
    memset(T_num, -1, sizeof T_num);
    memset(T_num0, 0, sizeof T_num0);
    memset(T_cnum0, 0, sizeof T_cnum0);
    const char* acgt = "ACGT";
    for (int i = 0; i < 4; i++) {                       /* :1288-1295 */
        int x = acgt[i], x2 = x + 32;
        T_num[x] = T_num[x2] = (int8_t)i;
        T_num0[x] = T_num0[x2] = (int8_t)i;
    }
    T_num['U'] = T_num['u'] = 3;                         /* :1296-1297 */
    T_num0['U'] = T_num0['u'] = 3;
    T_cnum0['A'] = T_cnum0['a'] = 3;                     /* :1306-1311 */
    T_cnum0['C'] = T_cnum0['c'] = 2;
    T_cnum0['G'] = T_cnum0['g'] = 1;
    T_cnum0['T'] = T_cnum0['t'] = 0;
    T_cnum0['U'] = T_cnum0['u'] = 0;
    tables_ready = 1;
}
/* Java indexes these tables with a signed byte; a byte >=128 would throw.  Reads reach this path
 * after Read.validate, so it cannot happen there; here such bytes are treated as undefined (0). */
int bbo_base_to_number(int b)             { init_tables(); return (b < 0 || b > 127) ? -1 : T_num[b]; }
int bbo_base_to_number0(int b)            { init_tables(); return (b < 0 || b > 127) ? 0 : T_num0[b]; }
int bbo_base_to_complement_number0(int b) { init_tables(); return (b < 0 || b > 127) ? 0 : T_cnum0[b]; }

/* bbduk/BBDukProcessorS.java:2858-2860 / BBDukLoader.java:538-540: symbol>=0 && symbolToNumber[symbol]>=0 */
static inline int is_fully_defined(uint8_t b) { return b < 128 && T_num[b] >= 0; }
static inline int64_t num0(uint8_t b)  { return b < 128 ? T_num0[b] : 0; }
static inline int64_t cnum0(uint8_t b) { return b < 128 ? T_cnum0[b] : 0; }

/* dna/AminoAcid.java:585-601 reverseComplementBinaryFast(long,int).  Java >>> is a logical shift. */
int64_t bbo_rcomp(int64_t kmer, int k) {
    uint64_t x = ~(uint64_t)kmer;
    x = ((x & 0x3333333333333333ULL) << 2)  | ((x & 0xCCCCCCCCCCCCCCCCULL) >> 2);
    x = ((x & 0x0F0F0F0F0F0F0F0FULL) << 4)  | ((x & 0xF0F0F0F0F0F0F0F0ULL) >> 4);
    x = ((x & 0x00FF00FF00FF00FFULL) << 8)  | ((x & 0xFF00FF00FF00FF00ULL) >> 8);
    x = ((x & 0x0000FFFF0000FFFFULL) << 16) | ((x & 0xFFFF0000FFFF0000ULL) >> 16);
    x = (x << 32) | (x >> 32);
    x = x >> (2 * (32 - k));      /* k in 1..31 here; Java shift distance is mod 64, same for these k */
    return (int64_t)x;
}

/* ------------------------------------------------------------------------------------------ */
/* kmer/HashArray.java + kmer/HashArray1D.java restated.  One instance per way.
 * NOT_PRESENT=-1, HASH_COLLISION=-2 (kmer/AbstractKmerTable.java:807).                         */
#define NOT_PRESENT   (-1)
#define HASH_COLLISION (-2)
#define HA_EXTRA 60                         /* kmer/HashArray.java:687 */
/* Test knob (bbo_test_set_probe_window): the probe window actually walked, <= HA_EXTRA.  The reference's is 60, which real reference
 * sets never fill, so its victim path (HashForest) would go untested; with a window of 2 the SAME code sends a few per cent of the
 * keys there.  The arrays keep their prime + 60 cells either way. */
static int g_probe_window = HA_EXTRA;
void bbo_test_set_probe_window(int n) { g_probe_window = n < 1 ? 1 : (n > HA_EXTRA ? HA_EXTRA : n); }
#define HA_MAX_LOAD 0.88f                   /* :695 */
#define HA_MIN_LOAD 0.58f                   /* :693 */
#define HA_RESIZE_MULT 2.0f                 /* :691 */
#define HA_INITIAL_SIZE 128000              /* kmer/ScheduleMaker.java:104,233 initialSizeDefault */

typedef struct {
    int      prime;
    int64_t  size, sizeLimit;
    int64_t* array;                         /* long[prime+extra], NOT_PRESENT = empty */
    int32_t* values;                        /* int [prime+extra]                       */
    /* victims: the reference uses a HashForest (bucket array of BSTs, kmer/HashForest.java).  It is a
     * key->value map reached only after `extra` consecutive occupied cells; a flat list with linear
     * search is decision-equivalent and is what victims().toList() hands across the boundary. */
    int64_t  vsize, vcap;
    int64_t* vkeys;
    int32_t* vvals;
} hash_array;

static int is_prime(int64_t x) {
    if (x < 2) return 0;
    if (x % 2 == 0) return x == 2;
    for (int64_t d = 3; d * d <= x; d += 2) if (x % d == 0) return 0;
    return 1;
}
static int64_t prime_at_least(int64_t x) { while (!is_prime(x)) x++; return x; }  /* shared/Primes.java:100 */

static void ha_alloc(hash_array* h, int prime) {
    h->prime = prime;
    h->sizeLimit = (int64_t)(HA_MAX_LOAD * prime);                  /* HashArray.java:62 */
    h->array  = (int64_t*)malloc(sizeof(int64_t) * ((size_t)prime + HA_EXTRA));
    h->values = (int32_t*)calloc((size_t)prime + HA_EXTRA, sizeof(int32_t));
    for (int64_t i = 0; i < (int64_t)prime + HA_EXTRA; i++) h->array[i] = NOT_PRESENT;   /* :65 */
    h->size = 0;
}
static void ha_init(hash_array* h) {
    memset(h, 0, sizeof *h);
    ha_alloc(h, (int)prime_at_least(HA_INITIAL_SIZE));
}
static void ha_free(hash_array* h) { free(h->array); free(h->values); free(h->vkeys); free(h->vvals); }

/* kmer/HashArray.java:153-156 kmerToCell; coreMask=-1 for BBDuk.  Java % on a non-negative long. */
static inline int ha_kmer_to_cell(const hash_array* h, int64_t kmer) { return (int)(kmer % h->prime); }

/* kmer/HashArray.java:434-447 findKmer */
static int ha_find_kmer(const hash_array* h, int64_t kmer) {
    int cell = ha_kmer_to_cell(h, kmer);
    for (const int max = cell + g_probe_window; cell < max; cell++) {
        const int64_t n = h->array[cell];
        if (n == kmer) return cell;
        else if (n == NOT_PRESENT) return NOT_PRESENT;
    }
    return HASH_COLLISION;
}
static int victims_get(const hash_array* h, int64_t kmer) {           /* kmer/HashForest.java:229-233 */
    for (int64_t i = 0; i < h->vsize; i++) if (h->vkeys[i] == kmer) return h->vvals[i];
    return NOT_PRESENT;
}
static int victims_set_if_not_present(hash_array* h, int64_t kmer, int value) {
    for (int64_t i = 0; i < h->vsize; i++) if (h->vkeys[i] == kmer) return 0;
    if (h->vsize == h->vcap) {
        h->vcap = h->vcap ? h->vcap * 2 : 16;
        h->vkeys = (int64_t*)realloc(h->vkeys, sizeof(int64_t) * (size_t)h->vcap);
        h->vvals = (int32_t*)realloc(h->vvals, sizeof(int32_t) * (size_t)h->vcap);
    }
    h->vkeys[h->vsize] = kmer; h->vvals[h->vsize] = value; h->vsize++;
    return 1;
}
/* kmer/HashArray.java:242-247 getValue + kmer/HashArray1D.java:117-119 readCellValue */
static int ha_get_value(const hash_array* h, int64_t kmer) {
    int cell = ha_find_kmer(h, kmer);
    if (cell == NOT_PRESENT) return NOT_PRESENT;
    if (cell == HASH_COLLISION) return victims_get(h, kmer);
    return h->values[cell];
}
static int ha_set_if_not_present(hash_array* h, int64_t kmer, int value);

/* kmer/HashArray1D.java:260-339 resize(), schedule==null branch ("old method"); re-inserts every
 * (key,value).  The reference's default BBDuk schedule depends on the JVM heap size
 * (kmer/ScheduleMaker.java:60-163) and is not reproducible here; table geometry never affects a
 * lookup result (it is a map), so only the map semantics are pinned. */
static void ha_resize(hash_array* h) {
    const int64_t totalSize = h->size + h->vsize;
    const int64_t maxAllowed = (int64_t)(totalSize * (1 / HA_MIN_LOAD));
    const int64_t minAllowed = (int64_t)(totalSize * (1 / HA_MAX_LOAD));
    if (maxAllowed < h->prime) { h->sizeLimit = (int64_t)(HA_MAX_LOAD * h->prime); return; }
    int64_t x = 10 + (int64_t)(h->prime * HA_RESIZE_MULT);
    if (x < minAllowed) x = minAllowed;
    if (x > maxAllowed) x = maxAllowed;
    int prime2 = (int)prime_at_least(x);
    if (prime2 <= h->prime) { h->sizeLimit = (int64_t)(HA_MAX_LOAD * h->prime); return; }
    hash_array old = *h;
    h->vkeys = NULL; h->vvals = NULL; h->vsize = h->vcap = 0;
    ha_alloc(h, prime2);
    for (int64_t i = 0; i < (int64_t)old.prime + HA_EXTRA; i++)
        if (old.array[i] > NOT_PRESENT) ha_set_if_not_present(h, old.array[i], old.values[i]);
    for (int64_t i = 0; i < old.vsize; i++) ha_set_if_not_present(h, old.vkeys[i], old.vvals[i]);
    free(old.array); free(old.values); free(old.vkeys); free(old.vvals);
}
/* kmer/HashArray.java:221-239 setIfNotPresent: first writer wins */
static int ha_set_if_not_present(hash_array* h, int64_t kmer, int value) {
    int cell = ha_kmer_to_cell(h, kmer);
    for (const int max = cell + g_probe_window; cell < max; cell++) {
        int64_t n = h->array[cell];
        if (n == kmer) return 0;
        else if (n == NOT_PRESENT) {
            h->array[cell] = kmer;
            h->values[cell] = value;                                   /* HashArray1D.insertValue */
            h->size++;
            if (h->size + h->vsize > h->sizeLimit) ha_resize(h);
            return 1;
        }
    }
    int x = victims_set_if_not_present(h, kmer, value);
    if (h->size + h->vsize > h->sizeLimit) ha_resize(h);
    return x;
}

/* ------------------------------------------------------------------------------------------ */
#define WAYS 7                               /* jgi/BBDuk.java:5425; BBDukIndexMod key%WAYS */
#define SYMBOLS 4
#define SYMBOL_ARRAY_LEN 32                  /* (64+2-1)/2, BBDukParser.java:251 */

struct bbo_ctx {
    bbo_args a;
    /* derived -- BBDukParser.java:130-312 */
    int k, mink, hammingDistance, hammingDistance2, editDistance, editDistance2, qHammingDistance, qHammingDistance2;
    int forbidNs, rcomp, maskMiddle, midMaskLen, useShortKmers, kfilter, kbig, keff;
    int minlen, minminlen, minlen2, shift, shift2, minSkip, maxSkip;
    int64_t mask, kmask, middleMask, symbolMask;
    int64_t clearMasks[SYMBOL_ARRAY_LEN], leftMasks[SYMBOL_ARRAY_LEN], rightMasks[SYMBOL_ARRAY_LEN], lengthMasks[SYMBOL_ARRAY_LEN];
    int64_t setMasks[SYMBOLS][SYMBOL_ARRAY_LEN];
    int removePairsIfEitherBad, trimFailuresTo1bp;
    /* index */
    hash_array keySets[WAYS];
    int numScaffolds;                        /* scaffoldNames.size(); [0] reserved (BBDukIndex.java:105-107) */
    int64_t storedKmers;
    /* counters (thread 0 / merged) */
    int64_t counters[BBO_NCOUNTERS];
    int64_t* scafReads; int64_t* scafBases; int scafCap;
};

void bbo_default_args(bbo_args* a) {
    memset(a, 0, sizeof *a);
    a->k = 27;                /* BBDukParser.java:163 */
    a->mink = -1;             /* :1230 */
    a->hdist = 0; a->hdist2 = -1; a->edist = 0; a->edist2 = -1; a->qhdist = 0; a->qhdist2 = -1;
    a->maskMiddle = 1;        /* :1091 */
    a->midMaskLen = 0;
    a->rcomp = 1;             /* :1208 */
    a->forbidN = 0;
    a->maxBadKmers0 = 0;      /* :1232 */
    a->minReadLength = 10;    /* :437 */
    a->minLenFraction = 0.f;  /* :439 */
    a->minSkip = 1; a->maxSkip = 1;
    a->trimPairsEvenly = 0; a->qSkip = 1; a->speed = 0;
    a->minKmerFraction = 0.f; a->minCoveredFraction = 0.f; a->ktrimN = 0;
    a->kbig = -1; a->findBestMatch = 0; a->ksplit = 0; a->kmaskFullyCovered = 0; a->trimFailuresTo1bp = 0;
}

static int imax(int a, int b) { return a > b ? a : b; }
static int imin(int a, int b) { return a < b ? a : b; }
static int imid(int lo, int x, int hi) { return x < lo ? lo : (x > hi ? hi : x); }   /* Tools.mid */

bbo_ctx* bbo_create(const bbo_args* a) {
    init_tables();
    if (a->k < 1 || a->k > 31) return NULL;                 /* maxSupportedK=31; a longer k arrives as kbig (:162-165) */
    if (a->edist > 0 || a->edist2 > 0) { /* allowed: mutate() below restates Del/Ins too */ }
    bbo_ctx* c = (bbo_ctx*)calloc(1, sizeof *c);
    c->a = *a;
    /* BBDukParser.java:130-132 */
    int hd = a->hdist, hd2 = a->hdist2, ed = a->edist, ed2 = a->edist2, qhd = a->qhdist, qhd2 = a->qhdist2;
    if (hd2 == -1) hd2 = hd;
    if (qhd2 == -1) qhd2 = qhd;
    if (ed2 == -1) ed2 = ed;
    /* :146-150 */
    hd = imax(ed, hd);
    hd2 = imax(ed2, hd2);
    c->minSkip = imax(1, imin(a->minSkip, a->maxSkip));
    c->maxSkip = imax(c->minSkip, a->maxSkip);
    c->forbidNs = (a->forbidN || hd < 1);
    c->hammingDistance = hd; c->hammingDistance2 = hd2; c->editDistance = ed; c->editDistance2 = ed2;
    c->qHammingDistance = qhd; c->qHammingDistance2 = qhd2;
    c->rcomp = a->rcomp;                                    /* :158-159 (amino=false) */
    c->k = a->k;
    /* :230-245 */
    c->maskMiddle = a->maskMiddle;
    c->kbig = (a->kbig > c->k ? a->kbig : c->k);
    if (c->kbig > c->k && (a->ktrimRight || a->ktrimLeft || a->ktrimN || a->ksplit || a->speed > 0 || a->qSkip > 1)) { free(c); return NULL; }
                                                            /* :207-223: the parser reduces kbig to k there (the caller passes that) */
    c->keff = imax(c->k, c->kbig);                          /* :231 */
    if (c->kbig > c->k) c->maskMiddle = 0;                  /* :237-243 */
    if (a->findBestMatch && c->kbig > c->k) { free(c); return NULL; }   /* :299 assert */
    if (c->maskMiddle) c->midMaskLen = (a->midMaskLen > 0 ? a->midMaskLen : 2 - (c->k & 1));
    else c->midMaskLen = 0;
    c->mink = imin(a->mink, c->k);
    /* :247-281 */
    c->symbolMask = 3;
    for (int i = 0; i < SYMBOL_ARRAY_LEN; i++) {
        c->clearMasks[i]  = (int64_t)~((uint64_t)c->symbolMask << (2 * i));
        c->leftMasks[i]   = (int64_t)(~0ULL << (2 * i));
        c->rightMasks[i]  = (int64_t)~(~0ULL << (2 * i));
        c->lengthMasks[i] = (int64_t)(1ULL << (2 * i));
        for (int64_t j = 0; j < SYMBOLS; j++) c->setMasks[j][i] = (int64_t)((uint64_t)j << (2 * i));
    }
    c->minlen = c->k - 1;
    c->minminlen = c->mink - 1;
    c->minlen2 = (c->maskMiddle ? (c->k - c->midMaskLen) / 2 : c->k);   /* computed BEFORE mink disables maskMiddle */
    c->shift = 2 * c->k;
    c->shift2 = c->shift - 2;
    c->mask = (c->shift > 63 ? -1LL : (int64_t)~(~0ULL << c->shift));
    c->kmask = c->lengthMasks[c->k];
    /* :289-296 */
    if (c->mink > 0 && c->mink < c->k) c->useShortKmers = 1;
    if (c->useShortKmers && c->maskMiddle) { c->maskMiddle = 0; c->midMaskLen = 0; }
    /* :298 */
    c->kfilter = !(a->ktrimRight || a->ktrimLeft || a->ktrimN || a->ksplit);
    if (c->useShortKmers && !(a->ktrimRight || a->ktrimLeft || a->ktrimN || a->ksplit)) { free(c); return NULL; }   /* :301 assert */
    /* :303-312 */
    if (c->maskMiddle) {
        if (!(c->k > c->midMaskLen + 1)) { free(c); return NULL; }
        int bits = c->midMaskLen * 2;
        int shift = ((c->k - c->midMaskLen) / 2) * 2;
        c->middleMask = (int64_t)~((~(~0ULL << bits)) << shift);
    } else c->middleMask = -1LL;
    /* :105-109 */
    c->trimFailuresTo1bp = a->trimFailuresTo1bp ? 1 : 0;
    c->removePairsIfEitherBad = (!a->requireBothBad) && (!c->trimFailuresTo1bp);
    for (int w = 0; w < WAYS; w++) ha_init(&c->keySets[w]);
    c->numScaffolds = 1;                                    /* scaffoldNames.add("") : first id is 1 */
    c->scafCap = 64;
    c->scafReads = (int64_t*)calloc((size_t)c->scafCap, sizeof(int64_t));
    c->scafBases = (int64_t*)calloc((size_t)c->scafCap, sizeof(int64_t));
    return c;
}
void bbo_destroy(bbo_ctx* c) {
    if (!c) return;
    for (int w = 0; w < WAYS; w++) ha_free(&c->keySets[w]);
    free(c->scafReads); free(c->scafBases); free(c);
}
int64_t bbo_constant(const bbo_ctx* c, const char* n) {
#define K(s, v) if (!strcmp(n, s)) return (int64_t)(v)
    K("k", c->k); K("mink", c->mink); K("hdist", c->hammingDistance); K("hdist2", c->hammingDistance2);
    K("forbidNs", c->forbidNs); K("minlen", c->minlen); K("minlen2", c->minlen2); K("shift2", c->shift2);
    K("mask", c->mask); K("kmask", c->kmask); K("middleMask", c->middleMask); K("useShortKmers", c->useShortKmers);
    K("maskMiddle", c->maskMiddle); K("midMaskLen", c->midMaskLen); K("kfilter", c->kfilter);
    K("qhdist", c->qHammingDistance); K("qhdist2", c->qHammingDistance2);
#undef K
    return INT64_MIN;
}

/* ------------------------------------------------------------------------------------------ */
/* bbduk/BBDukIndexMod.java:532-544 toValue */
static inline int64_t to_value(const bbo_ctx* c, int64_t kmer, int64_t rkmer, int64_t lengthMask) {
    const int64_t value = (c->rcomp ? (kmer > rkmer ? kmer : rkmer) : kmer);     /* Tools.max, signed */
    return (value & c->middleMask) | lengthMask;
}

/* bbduk/BBDukIndexMod.java:383-445 mutate */
static int64_t mutate(bbo_ctx* c, const int64_t kmer, const int64_t rkmer, const int len, const int id,
                      const int dist, const int64_t extraBase, const int tnum) {
    hash_array* map = &c->keySets[tnum];
    int64_t added = 0;
    const int64_t key = to_value(c, kmer, rkmer, c->lengthMasks[len]);
    if (key % WAYS == tnum) added += ha_set_if_not_present(map, key, id);
    if (dist > 0) {
        const int dist2 = dist - 1;
        /* Sub */
        for (int j = 0; j < SYMBOLS; j++) {
            for (int i = 0; i < len; i++) {
                const int64_t temp = (kmer & c->clearMasks[i]) | c->setMasks[j][i];
                if (temp != kmer) {
                    int64_t rtemp = bbo_rcomp(temp, len);
                    added += mutate(c, temp, rtemp, len, id, dist2, extraBase, tnum);
                }
            }
        }
        if (c->editDistance > 0) {
            /* Del (:415-425) */
            if (extraBase >= 0 && extraBase <= 3) {
                for (int i = 1; i < len; i++) {
                    const int64_t temp = (kmer & c->leftMasks[i]) | ((int64_t)((uint64_t)kmer << 2) & c->rightMasks[i]) | extraBase;
                    if (temp != kmer) {
                        int64_t rtemp = bbo_rcomp(temp, len);
                        added += mutate(c, temp, rtemp, len, id, dist2, -1, tnum);
                    }
                }
            }
            /* Ins (:427-439); Java's >> on a non-negative long == logical */
            const int64_t eb2 = kmer & c->symbolMask;
            for (int i = 1; i < len; i++) {
                const int64_t temp0 = (kmer & c->leftMasks[i]) | ((kmer & c->rightMasks[i]) >> 2);
                for (int j = 0; j < SYMBOLS; j++) {
                    const int64_t temp = temp0 | c->setMasks[j][i - 1];
                    if (temp != kmer) {
                        int64_t rtemp = bbo_rcomp(temp, len);
                        added += mutate(c, temp, rtemp, len, id, dist2, eb2, tnum);
                    }
                }
            }
        }
    }
    return added;
}

/* bbduk/BBDukIndexMod.java:351-373 addToMap */
static int64_t idx_add_to_map(bbo_ctx* c, const int64_t kmer, const int64_t rkmer, const int len, const int64_t extraBase,
                              const int id, const int64_t kmask0, const int hdist, const int edist, const int tnum) {
    /* asserts :354-355 */
    if (kmask0 != c->lengthMasks[len] || (kmer & kmask0) != 0) { fprintf(stderr, "oracle: addToMap assertion\n"); abort(); }
    if (hdist == 0) {
        const int64_t key = to_value(c, kmer, rkmer, kmask0);
        if (key % WAYS != tnum) return 0;
        return ha_set_if_not_present(&c->keySets[tnum], key, id);
    } else if (edist > 0) {
        return mutate(c, kmer, rkmer, len, id, edist, extraBase, tnum);
    } else {
        return mutate(c, kmer, rkmer, len, id, hdist, -1, tnum);
    }
}
/* bbduk/BBDukIndexMod.java:289-310 addToMapLeftShift */
static int64_t idx_add_left_shift(bbo_ctx* c, int64_t kmer, int64_t rkmer, const int64_t extraBase, const int id, const int tnum) {
    int64_t added = 0;
    for (int i = c->k - 1; i >= c->mink; i--) {
        kmer = kmer & c->rightMasks[i];
        rkmer = (int64_t)((uint64_t)rkmer >> 2);
        added += idx_add_to_map(c, kmer, rkmer, i, extraBase, id, c->lengthMasks[i], c->hammingDistance2, c->editDistance2, tnum);
    }
    return added;
}
/* bbduk/BBDukIndexMod.java:320-341 addToMapRightShift */
static int64_t idx_add_right_shift(bbo_ctx* c, int64_t kmer, int64_t rkmer, const int id, const int tnum) {
    int64_t added = 0;
    for (int i = c->k - 1; i >= c->mink; i--) {
        int64_t extraBase = kmer & c->symbolMask;
        kmer = (int64_t)((uint64_t)kmer >> 2);
        rkmer = rkmer & c->rightMasks[i];
        added += idx_add_to_map(c, kmer, rkmer, i, extraBase, id, c->lengthMasks[i], c->hammingDistance2, c->editDistance2, tnum);
    }
    return added;
}

/* bbduk/BBDukLoader.java:416-494 LoadThread.addToMap(Read,skip) for one way (tnum) */
static int64_t loader_add_to_map(bbo_ctx* c, const uint8_t* bases, int64_t blen, int id, int skip, int tnum) {
    skip = imax(c->minSkip, imin(c->maxSkip, skip));
    const int k = c->k, k2 = k - 1;
    int64_t kmer = 0, rkmer = 0, added = 0;
    int64_t len = 0;
    if (bases == NULL || blen < k) return 0;
    if (skip > 1) {
        for (int64_t i = 0; i < blen; i++) {
            uint8_t b = bases[i];
            int64_t x = num0(b), x2 = cnum0(b);
            kmer = (int64_t)((((uint64_t)kmer << 2) | (uint64_t)x) & (uint64_t)c->mask);
            rkmer = (int64_t)((((uint64_t)rkmer >> 2) | ((uint64_t)x2 << c->shift2)) & (uint64_t)c->mask);
            if (is_fully_defined(b)) len++; else { len = 0; rkmer = 0; }
            if (len >= k) {
                if (len % skip == 0) {
                    const int64_t extraBase = (i >= blen - 1 ? -1 : bbo_base_to_number(bases[i + 1]));
                    added += idx_add_to_map(c, kmer, rkmer, k, extraBase, id, c->kmask, c->hammingDistance, c->editDistance, tnum);
                    if (c->useShortKmers) {
                        if (i == k2) added += idx_add_right_shift(c, kmer, rkmer, id, tnum);
                        if (i == blen - 1) added += idx_add_left_shift(c, kmer, rkmer, extraBase, id, tnum);
                    }
                }
            }
        }
    } else {
        for (int64_t i = 0; i < blen; i++) {
            const uint8_t b = bases[i];
            const int64_t x = num0(b), x2 = cnum0(b);
            kmer = (int64_t)((((uint64_t)kmer << 2) | (uint64_t)x) & (uint64_t)c->mask);
            rkmer = (int64_t)((((uint64_t)rkmer >> 2) | ((uint64_t)x2 << c->shift2)) & (uint64_t)c->mask);
            if (is_fully_defined(b)) len++; else { len = 0; rkmer = 0; }
            if (len >= k) {
                const int64_t extraBase = (i >= blen - 1 ? -1 : bbo_base_to_number(bases[i + 1]));
                added += idx_add_to_map(c, kmer, rkmer, k, extraBase, id, c->kmask, c->hammingDistance, c->editDistance, tnum);
                if (c->useShortKmers) {
                    if (i == k2) added += idx_add_right_shift(c, kmer, rkmer, id, tnum);
                    if (i == blen - 1) added += idx_add_left_shift(c, kmer, rkmer, extraBase, id, tnum);
                }
            }
        }
    }
    return added;
}

/* bbduk/BBDukLoader.java:219-251, 388-404: every scaffold gets the next id; every LoadThread (one per way)
 * sees every scaffold in file order and inserts only keys it owns, so within a way insertion order is
 * file order and setIfNotPresent keeps the id of the first scaffold that produced the key. */
int64_t bbo_add_ref_sequence(bbo_ctx* c, const uint8_t* bases, int64_t len) {
    const int id = c->numScaffolds++;
    if (c->numScaffolds > c->scafCap) {
        int nc = c->scafCap * 2;
        c->scafReads = (int64_t*)realloc(c->scafReads, sizeof(int64_t) * (size_t)nc);
        c->scafBases = (int64_t*)realloc(c->scafBases, sizeof(int64_t) * (size_t)nc);
        memset(c->scafReads + c->scafCap, 0, sizeof(int64_t) * (size_t)(nc - c->scafCap));
        memset(c->scafBases + c->scafCap, 0, sizeof(int64_t) * (size_t)(nc - c->scafCap));
        c->scafCap = nc;
    }
    const int skip = len > 20000000 ? c->k : len > 5000000 ? 11 : len > 500000 ? 2 : 0;     /* :397 */
    int64_t added = 0;
    for (int tnum = 0; tnum < WAYS; tnum++) added += loader_add_to_map(c, bases, len, id, skip, tnum);
    c->storedKmers += added;
    return added;
}

int bbo_load_fasta(bbo_ctx* c, const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) return -1;
    size_t cap = 1 << 16, n = 0; uint8_t* seq = (uint8_t*)malloc(cap);
    int have = 0, nrec = 0, ch, bol = 1, inhdr = 0;
    while ((ch = fgetc(f)) != EOF) {
        if (inhdr) { if (ch == '\n') { inhdr = 0; bol = 1; } continue; }
        if (bol && ch == '>') {
            if (have && n > 0) { bbo_add_ref_sequence(c, seq, (int64_t)n); nrec++; }   /* stream/FastaReadInputStream minLen=1 */
            have = 1; n = 0; inhdr = 1; continue;
        }
        if (ch == '\n' || ch == '\r') { bol = 1; continue; }
        bol = 0;
        if (ch > '\r') {                                   /* FastaReadInputStream.java:362 buffer[x]>slashr */
            if (n == cap) { cap *= 2; seq = (uint8_t*)realloc(seq, cap); }
            seq[n++] = (uint8_t)ch;
        }
    }
    if (have && n > 0) { bbo_add_ref_sequence(c, seq, (int64_t)n); nrec++; }
    free(seq); fclose(f);
    return nrec;
}
int bbo_num_scaffolds(const bbo_ctx* c) { return c->numScaffolds; }
int64_t bbo_stored_kmers(const bbo_ctx* c) { return c->storedKmers; }

int bbo_table_get(const bbo_ctx* c, int64_t key) {
    if (key < 0) return NOT_PRESENT;
    return ha_get_value(&c->keySets[(int)(key % WAYS)], key);
}
int bbo_num_ways(const bbo_ctx* c) { (void)c; return WAYS; }
int bbo_way_image(const bbo_ctx* c, int way, int* prime, int64_t* ncells, const int64_t** keys, const int32_t** values,
                  int64_t* nvictims, const int64_t** vkeys, const int32_t** vvals) {
    if (way < 0 || way >= WAYS) return -1;
    const hash_array* h = &c->keySets[way];
    *prime = h->prime; *ncells = (int64_t)h->prime + HA_EXTRA; *keys = h->array; *values = h->values;
    *nvictims = h->vsize; *vkeys = h->vkeys; *vvals = h->vvals;
    return 0;
}
int64_t bbo_dump_pairs(const bbo_ctx* c, int64_t* keys, int32_t* values, int64_t cap) {
    int64_t n = 0;
    for (int w = 0; w < WAYS; w++) {
        const hash_array* h = &c->keySets[w];
        for (int64_t i = 0; i < (int64_t)h->prime + HA_EXTRA; i++) if (h->array[i] > NOT_PRESENT) {
            if (keys && n < cap) { keys[n] = h->array[i]; values[n] = h->values[i]; }
            n++;
        }
        for (int64_t i = 0; i < h->vsize; i++) {
            if (keys && n < cap) { keys[n] = h->vkeys[i]; values[n] = h->vvals[i]; }
            n++;
        }
    }
    return n;
}

/* ------------------------------------------------------------------------------------------ */
/* bbduk/BBDukIndexMod.java:492-520 getValueInner (qSkip=1, speed=0) */
static int get_value_inner(const bbo_ctx* c, const int64_t kmer, const int64_t rkmer, const int64_t lengthMask, const int qPos) {
    if (c->a.qSkip > 1 && (qPos % c->a.qSkip != 0)) return -1;                        /* :494 */
    const int64_t max = (c->rcomp ? (kmer > rkmer ? kmer : rkmer) : kmer);
    const int64_t key = (max & c->middleMask) | lengthMask;
    if (!(c->a.speed < 1 || ((key & INT64_MAX) % 17) >= c->a.speed)) return -1;     /* passesSpeed :506,562 */
    return ha_get_value(&c->keySets[(int)(key % WAYS)], key);
}
/* bbduk/BBDukIndexMod.java:462-481 getValue */
int bbo_get_value(const bbo_ctx* c, const int64_t kmer, const int64_t rkmer, const int64_t lengthMask, const int qPos, const int len, const int qHDist) {
    int id = get_value_inner(c, kmer, rkmer, lengthMask, qPos);
    if (id < 1 && qHDist > 0) {
        const int qHDist2 = qHDist - 1;
        for (int j = 0; j < SYMBOLS && id < 1; j++) {
            for (int i = 0; i < len && id < 1; i++) {
                const int64_t temp = (kmer & c->clearMasks[i]) | c->setMasks[j][i];
                if (temp != kmer) {
                    int64_t rtemp = bbo_rcomp(temp, len);
                    id = bbo_get_value(c, temp, rtemp, lengthMask, qPos, len, qHDist2);
                }
            }
        }
    }
    return id;
}

/* per-thread counter block */
typedef struct { int64_t counters[BBO_NCOUNTERS]; int64_t* scafReads; int64_t* scafBases;
                 int* countArray; int* idList;
                 int32_t* mN; int32_t* mIds; int32_t* mCnt; int mCap; int64_t mRead; } tcounters;             /* findBestMatch's per-thread state (:3170-3172) */

/* shared/TrimRead.java:304-345 trimByAmount, on lengths only (bases/quals are copied by the caller in Java).
 * Returns total trimmed; *newLen gets the resulting length. */
static int trim_by_amount(int len, int leftTrimAmount, int rightTrimAmount, int minResultingLength, int* newLen) {
    leftTrimAmount = imax(leftTrimAmount, 0);
    rightTrimAmount = imax(rightTrimAmount, 0);
    if (len < 1) { *newLen = len; return 0; }
    minResultingLength = imin(len, imax(minResultingLength, 0));
    if (leftTrimAmount + rightTrimAmount + minResultingLength > len) {
        rightTrimAmount = imax(1, len - minResultingLength);
        leftTrimAmount = 0;
    }
    const int total = leftTrimAmount + rightTrimAmount;
    *newLen = len - total;
    return total;
}
/* shared/TrimRead.java:273-276 trimToPosition */
static int trim_to_position(int len, int leftLoc, int rightLoc, int minResultingLength, int* newLen) {
    return trim_by_amount(len, leftLoc, len - rightLoc - 1, minResultingLength, newLen);
}

/* bbduk/BBDukProcessorS.java:1993-2140 ktrim(Read,start,stop).  *newLen = r.length() afterwards. */
static int ktrim_span(const bbo_ctx* c, tcounters* tc, const uint8_t* bases, const int blen, const int pairnum,
                      const int start, const int stop, const int ktrimLeft, const int ktrimRight, int* id0out, int* newLen) {
    const int k = c->k;
    *newLen = blen; *id0out = -1;
    if (blen < imax(1, (c->useShortKmers ? imin(k, c->mink) : k)) || c->storedKmers < 1) return 0;
    if ((c->a.skipR1 && pairnum == 0) || (c->a.skipR2 && pairnum == 1)) return 0;
    int64_t kmer = 0, rkmer = 0;
    int found = 0, len = 0, id0 = -1;
    int minLoc = 999999999, minLocExclusive = 999999999;
    int maxLoc = -1, maxLocExclusive = -1;

    for (int i = start; i < stop; i++) {                                            /* :2009-2029 */
        uint8_t b = bases[i];
        int64_t x = num0(b), x2 = cnum0(b);
        kmer = (int64_t)((((uint64_t)kmer << 2) | (uint64_t)x) & (uint64_t)c->mask);
        rkmer = (int64_t)((((uint64_t)rkmer >> 2) | ((uint64_t)x2 << c->shift2)) & (uint64_t)c->mask);
        if (c->forbidNs && !is_fully_defined(b)) { len = 0; rkmer = 0; } else { len++; }
        if (len >= c->minlen2 && i >= c->minlen) {
            const int id = bbo_get_value(c, kmer, rkmer, c->kmask, i, k, c->qHammingDistance);
            if (id > 0) {
                if (id0 < 0) id0 = id;
                minLoc = imin(minLoc, i - k + 1);
                maxLoc = i;
                found++;
            }
        }
    }
    if (minLoc != minLocExclusive) minLocExclusive = minLoc + k;                    /* :2031-2032 */
    if (maxLoc != maxLocExclusive) maxLocExclusive = maxLoc - k;

    if (c->useShortKmers && found == 0) {                                           /* :2034-2103 */
        if (ktrimLeft) {
            kmer = 0; rkmer = 0; len = 0;
            const int lim = imin(k, stop);
            for (int i = start; i < lim; i++) {
                uint8_t b = bases[i];
                int64_t x = num0(b), x2 = cnum0(b);
                kmer = (int64_t)((((uint64_t)kmer << 2) | (uint64_t)x) & (uint64_t)c->mask);
                rkmer = rkmer | (int64_t)((uint64_t)x2 << (2 * len));
                len++;
                if (len >= c->mink) {
                    const int id = bbo_get_value(c, kmer, rkmer, c->lengthMasks[len], i, len, c->qHammingDistance2);
                    if (id > 0) {
                        if (id0 < 0) id0 = id;
                        minLoc = 0;
                        minLocExclusive = imin(minLocExclusive, i + 1);
                        maxLoc = imax(maxLoc, i);
                        maxLocExclusive = imax(maxLocExclusive, 0);
                        found++;
                    }
                }
            }
        }
        if (ktrimRight) {
            kmer = 0; rkmer = 0; len = 0;
            const int lim = imax(-1, stop - k);
            for (int i = stop - 1; i > lim; i--) {
                uint8_t b = bases[i];
                int64_t x = num0(b), x2 = cnum0(b);
                kmer = kmer | (int64_t)((uint64_t)x << (2 * len));
                rkmer = (int64_t)((((uint64_t)rkmer << 2) | (uint64_t)x2) & (uint64_t)c->mask);
                len++;
                if (len >= c->mink) {
                    const int id = bbo_get_value(c, kmer, rkmer, c->lengthMasks[len], i, len, c->qHammingDistance2);
                    if (id > 0) {
                        if (id0 < 0) id0 = id;
                        minLoc = i;
                        minLocExclusive = imin(minLocExclusive, blen);
                        maxLoc = blen - 1;
                        maxLocExclusive = imax(maxLocExclusive, i - 1);
                        found++;
                    }
                }
            }
        }
    }
    if (found == 0) return 0;                                                        /* :2108 */
    tc->scafReads[id0]++;                                                            /* :2111-2119 */
    tc->scafBases[id0] += blen;
    *id0out = id0;
    if (c->a.trimPad != 0) {                                                         /* :2121-2126 */
        const int tp = c->a.trimPad;
        maxLoc = imid(0, maxLoc + tp, blen);
        minLoc = imid(0, minLoc - tp, blen);
        maxLocExclusive = imid(0, maxLocExclusive + tp, blen);
        minLocExclusive = imid(0, minLocExclusive - tp, blen);
    }
    if (ktrimLeft) {                                                                 /* :2128-2132 */
        return trim_to_position(blen, c->a.ktrimExclusive ? maxLocExclusive + 1 : maxLoc + 1, blen - 1, 1, newLen);
    } else {                                                                         /* :2133-2139 */
        return trim_to_position(blen, 0, c->a.ktrimExclusive ? minLocExclusive - 1 : minLoc - 1, 1, newLen);
    }
}
/* bbduk/BBDukProcessorS.java:1806-1811 ktrim(Read) */
static int ktrim_read(const bbo_ctx* c, tcounters* tc, const uint8_t* bases, int len, int pairnum, int* id0, int* newLen) {
    const int start = (c->a.restrictRight < 1 ? 0 : imax(0, len - c->a.restrictRight));
    const int stop  = (c->a.restrictLeft  < 1 ? len : imin(len, c->a.restrictLeft));
    return ktrim_span(c, tc, bases, len, pairnum, start, stop, c->a.ktrimLeft, c->a.ktrimRight, id0, newLen);
}

/* bbduk/BBDukProcessorS.java:1813-1828 ktrimTips: a right pass over [start,len), then a left pass over [0,stop) of the
 * read as the right pass left it (its first *newLen bases).  ktrimTip (:1832-1985) is ktrim(Read,start,stop) with the
 * side given explicitly.  *xLeft = what the left pass removed. */
static int ktrim_tips_read(const bbo_ctx* c, tcounters* tc, const uint8_t* bases, int len, int pairnum, int* id0, int* newLen, int* xLeft) {
    const int k = c->k;
    const int mid = len / 2 - (k - 1) / 2;
    int sum = 0, cur = len, idr = -1, idl = -1;
    *xLeft = 0;
    if (c->a.ktrimRight) {
        const int start = imax(0, (c->a.restrictRight < 1 ? mid : len - c->a.restrictRight));
        sum += ktrim_span(c, tc, bases, cur, pairnum, start, cur, 0, 1, &idr, &cur);
    }
    if (c->a.ktrimLeft) {
        const int stop = imin(cur, (c->a.restrictLeft < 1 ? mid + k - 1 : c->a.restrictLeft));
        const int x = ktrim_span(c, tc, bases, cur, pairnum, 0, stop, 1, 0, &idl, &cur);
        sum += x; *xLeft = x;
    }
    *id0 = idr >= 0 ? idr : idl;
    *newLen = cur;
    return sum;
}

/* bbduk/BBDukProcessorS.java:1534-1593 countSetKmers */
static int count_set_kmers(const bbo_ctx* c, tcounters* tc, const uint8_t* bases, const int blen, const int pairnum,
                           const int maxBadKmers, int* idout) {
    const int k = c->k;
    *idout = -1;
    if (blen < k || c->storedKmers < 1) return 0;
    if ((c->a.skipR1 && pairnum == 0) || (c->a.skipR2 && pairnum == 1)) return 0;
    int64_t kmer = 0, rkmer = 0;
    int found = 0, len = 0;
    const int start = (c->a.restrictRight < 1 ? 0 : imax(0, blen - c->a.restrictRight));
    const int stop  = (c->a.restrictLeft  < 1 ? blen : imin(blen, c->a.restrictLeft));
    for (int i = start; i < stop; i++) {
        uint8_t b = bases[i];
        int64_t x = num0(b), x2 = cnum0(b);
        kmer = (int64_t)((((uint64_t)kmer << 2) | (uint64_t)x) & (uint64_t)c->mask);
        rkmer = (int64_t)((((uint64_t)rkmer >> 2) | ((uint64_t)x2 << c->shift2)) & (uint64_t)c->mask);
        if (c->forbidNs && !is_fully_defined(b)) { len = 0; rkmer = 0; } else { len++; }
        if (len >= c->minlen2 && i >= c->minlen) {
            const int id = bbo_get_value(c, kmer, rkmer, c->kmask, i, k, c->qHammingDistance);
            if (id > 0) {
                if (found == maxBadKmers) {
                    tc->scafReads[id]++;
                    tc->scafBases[id] += blen;
                    *idout = id;
                    return (found = found + 1);
                }
                found++;
            }
        }
    }
    return found;
}

/* bbduk/BBDukProcessorS.java:1726-1804 countSetKmersBig */
static int count_set_kmers_big(const bbo_ctx* c, tcounters* tc, const uint8_t* bases, const int blen, const int pairnum,
                               const int maxBadKmers, int* idout) {
    const int k = c->k, kbig = c->kbig;
    *idout = -1;
    if (blen < kbig || c->storedKmers < 1) return 0;
    if ((c->a.skipR1 && pairnum == 0) || (c->a.skipR2 && pairnum == 1)) return 0;
    const int sub = kbig - k - 1;
    int64_t kmer = 0, rkmer = 0;
    int found = 0, len = 0;
    int bkStart = -1, bkStop = -1;
    int id = -1, lastId = -1;
    const int start = (c->a.restrictRight < 1 ? 0 : imax(0, blen - c->a.restrictRight));
    const int stop  = (c->a.restrictLeft  < 1 ? blen : imin(blen, c->a.restrictLeft));
    for (int i = start; i < stop; i++) {
        uint8_t b = bases[i];
        int64_t x = num0(b), x2 = cnum0(b);
        kmer = (int64_t)((((uint64_t)kmer << 2) | (uint64_t)x) & (uint64_t)c->mask);
        rkmer = (int64_t)((((uint64_t)rkmer >> 2) | ((uint64_t)x2 << c->shift2)) & (uint64_t)c->mask);
        if (c->forbidNs && !is_fully_defined(b)) { len = 0; rkmer = 0; } else { len++; }
        if (len >= c->minlen2 && i >= c->minlen) {
            id = bbo_get_value(c, kmer, rkmer, c->kmask, i, k, c->qHammingDistance);
            if (id > 0) {
                lastId = id;
                if (bkStart == -1) bkStart = i;
                bkStop = i;
            } else if (bkStart > -1) {
                const int dif = bkStop - bkStart - sub;
                bkStop = bkStart = -1;
                if (dif > 0) {
                    const int old = found;
                    found += dif;
                    if (found > maxBadKmers && old <= maxBadKmers) {
                        tc->scafReads[lastId]++; tc->scafBases[lastId] += blen;
                        *idout = lastId;
                        return found;                                                     /* :1773 early exit */
                    }
                }
            }
        }
    }
    if (bkStart > -1) {                                                                   /* :1783-1800 */
        const int dif = bkStop - bkStart - sub;
        if (dif > 0) {
            const int old = found;
            found += dif;
            if (found > maxBadKmers && old <= maxBadKmers) {
                tc->scafReads[lastId]++; tc->scafBases[lastId] += blen;
                *idout = lastId;
            }
        }
    }
    return found;
}

/* bbduk/BBDukProcessorS.java:1659-1719 findBestMatch.  Returns the id (or -1); *foundout = hits counted.  Restated for
 * maxBadKmers == 0 only: with found <= maxBadKmers the reference skips condenseLoose and leaves countArray dirty. */
static int find_best_match(const bbo_ctx* c, tcounters* tc, const uint8_t* bases, const int blen, const int pairnum,
                           const int maxBadKmers, int* foundout) {
    const int k = c->k;
    *foundout = 0;
    int nids = 0;
    if (tc->mN) tc->mN[tc->mRead + pairnum] = 0;                                           /* rename's lists (:1702, 2508-2522) */
    if (blen < k || c->storedKmers < 1) return -1;
    if ((c->a.skipR1 && pairnum == 0) || (c->a.skipR2 && pairnum == 1)) return -1;
    if (!tc->countArray) { tc->countArray = (int*)calloc((size_t)c->numScaffolds, sizeof(int)); tc->idList = (int*)calloc((size_t)c->numScaffolds, sizeof(int)); }
    int64_t kmer = 0, rkmer = 0;
    int found = 0, len = 0;
    const int start = (c->a.restrictRight < 1 ? 0 : imax(0, blen - c->a.restrictRight));
    const int stop  = (c->a.restrictLeft  < 1 ? blen : imin(blen, c->a.restrictLeft));
    for (int i = start; i < stop; i++) {
        uint8_t b = bases[i];
        int64_t x = num0(b), x2 = cnum0(b);
        kmer = (int64_t)((((uint64_t)kmer << 2) | (uint64_t)x) & (uint64_t)c->mask);
        rkmer = (int64_t)((((uint64_t)rkmer >> 2) | ((uint64_t)x2 << c->shift2)) & (uint64_t)c->mask);
        if (c->forbidNs && !is_fully_defined(b)) { len = 0; rkmer = 0; } else { len++; }
        if (len >= c->minlen2 && i >= c->minlen) {
            const int id = bbo_get_value(c, kmer, rkmer, c->kmask, i, k, c->qHammingDistance);
            if (id > 0) {
                tc->countArray[id]++;
                if (tc->countArray[id] == 1) tc->idList[nids++] = id;
                found++;
            }
        }
    }
    *foundout = found;
    int id = -1;
    if (found > maxBadKmers) {
        int max = 0;                                                                      /* condenseLoose :2531-2544 */
        for (int i = 0; i < nids; i++) { const int cnt = tc->countArray[tc->idList[i]]; if (cnt > max) max = cnt; }
        for (int i = 0; i < nids; i++) if (tc->countArray[tc->idList[i]] == max) { id = tc->idList[i]; break; }
        if (tc->mN) {                                                                     /* idList / countList as rename() walks them */
            const int64_t rd = tc->mRead + pairnum;
            tc->mN[rd] = nids;
            for (int i = 0; i < nids && i < tc->mCap; i++) { tc->mIds[rd * tc->mCap + i] = tc->idList[i]; tc->mCnt[rd * tc->mCap + i] = tc->countArray[tc->idList[i]]; }
        }
        for (int i = 0; i < nids; i++) tc->countArray[tc->idList[i]] = 0;
        tc->scafReads[id]++; tc->scafBases[id] += blen;
    }
    return id;
}

/* bbduk/BBDukProcessorS.java:2332-2506 ksplit, on lengths: returns 1 if the read is split; *trimmed = oldLen - new pairLength */
static int ksplit_read(const bbo_ctx* c, tcounters* tc, const uint8_t* bases, const int blen, int* id0out, int* leftOut, int* rightOut,
                       int* trimmed, int* newPairLen) {
    const int k = c->k;
    *id0out = -1; *leftOut = -1; *rightOut = -1; *trimmed = 0; *newPairLen = blen;
    if (blen < imax(1, (c->useShortKmers ? imin(k, c->mink) : k)) || c->storedKmers < 1) return 0;
    if (blen < k) return 0;
    int64_t kmer = 0, rkmer = 0;
    int64_t found = 0;
    int len = 0, id0 = -1;
    int leftmost = 2147483647, rightmost = -1;
    const int minus = k - 1 - c->a.trimPad, plus = c->a.trimPad;
    const int start = (c->a.restrictRight < 1 ? 0 : imax(0, blen - c->a.restrictRight));
    const int stop  = (c->a.restrictLeft  < 1 ? blen : imin(blen, c->a.restrictLeft));
    for (int i = start; i < stop; i++) {
        uint8_t b = bases[i];
        int64_t x = num0(b), x2 = cnum0(b);
        kmer = (int64_t)((((uint64_t)kmer << 2) | (uint64_t)x) & (uint64_t)c->mask);
        rkmer = (int64_t)((((uint64_t)rkmer >> 2) | ((uint64_t)x2 << c->shift2)) & (uint64_t)c->mask);
        if (c->forbidNs && !is_fully_defined(b)) { len = 0; rkmer = 0; } else { len++; }
        if (i >= c->minlen) {
            const int id = (len >= c->minlen2) ? bbo_get_value(c, kmer, rkmer, c->kmask, i, k, c->qHammingDistance) : -1;
            if (id > 0) {
                if (id0 < 0) id0 = id;
                leftmost = imin(leftmost, imax(0, i - minus));
                rightmost = imax(rightmost, i + plus);
                found++;
            }
        }
    }
    if (c->useShortKmers && id0 == -1) {
        {                                                                                 /* right side :2391-2431 */
            kmer = 0; rkmer = 0; len = 0;
            const int lim = imax(-1, stop - k);
            for (int i = stop - 1; i > lim; i--) {
                uint8_t b = bases[i];
                int64_t x = num0(b), x2 = cnum0(b);
                kmer = kmer | (int64_t)((uint64_t)x << (2 * len));
                rkmer = (int64_t)((((uint64_t)rkmer << 2) | (uint64_t)x2) & (uint64_t)c->mask);
                len++;
                if (len >= c->minminlen) {
                    const int id = (len >= c->mink) ? bbo_get_value(c, kmer, rkmer, c->lengthMasks[len], i, len, c->qHammingDistance2) : -1;
                    if (id > 0) {
                        if (id0 < 0) id0 = id;
                        leftmost = imin(leftmost, imax(0, i - c->a.trimPad));
                        rightmost = blen - 1;
                        found++;
                    }
                }
            }
        }
        if (id0 == -1) {                                                                  /* left side :2434-2473 */
            kmer = 0; rkmer = 0; len = 0;
            const int lim = imin(k, stop);
            for (int i = start; i < lim; i++) {
                uint8_t b = bases[i];
                int64_t x = num0(b), x2 = cnum0(b);
                kmer = (int64_t)((((uint64_t)kmer << 2) | (uint64_t)x) & (uint64_t)c->mask);
                rkmer = rkmer | (int64_t)((uint64_t)x2 << (2 * len));
                len++;
                if (len >= c->minminlen) {
                    const int id = (len >= c->mink) ? bbo_get_value(c, kmer, rkmer, c->lengthMasks[len], i, len, c->qHammingDistance2) : -1;
                    if (id > 0) {
                        if (id0 < 0) id0 = id;
                        leftmost = 0;
                        rightmost = imax(rightmost, i + c->a.trimPad);
                        found++;
                    }
                }
            }
        }
    }
    if (found == 0) return 0;
    tc->scafReads[id0]++; tc->scafBases[id0] += blen;
    *id0out = id0; *leftOut = leftmost; *rightOut = rightmost;
    int n1 = blen;
    if (leftmost == 0) {
        trim_to_position(blen, rightmost + 1, blen - 1, 1, &n1);
        *newPairLen = n1; *trimmed = blen - n1;
        return 0;
    } else if (rightmost == blen - 1) {
        trim_to_position(blen, 0, leftmost - 1, 1, &n1);
        *newPairLen = n1; *trimmed = blen - n1;
        return 0;
    }
    const int n2 = (blen - 1) - (rightmost + 1);            /* subRead(rightmost+1, length-1): copyOfRange's `to` is exclusive */
    if (n2 < 0) { *newPairLen = blen; return 0; }           /* rightmost beyond the end (trimPad>0): the reference throws; callers reject tp>0 */
    trim_to_position(blen, 0, leftmost - 1, 1, &n1);
    *newPairLen = n1 + n2; *trimmed = blen - (n1 + n2);
    return 1;
}

/* java.util.BitSet.set(from,to) on a word array */
static void bs_set(uint64_t* bs, int from, int to) {
    for (int i = from; i < to; i++) bs[i >> 6] |= 1ULL << (i & 63);
}

/* bbduk/BBDukProcessorS.java:2149-2323 kmask(Read), kmaskFullyCovered=false.  Returns bs.cardinality(); the masked
 * positions < blen are OR-ed into gmask at bit offset gbase (atomically: neighbouring reads share words). */
static void bs_clear(uint64_t* bs, int from, int to) {       /* java.util.BitSet.clear(from,to) */
    for (int i = from; i < to; i++) bs[i >> 6] &= ~(1ULL << (i & 63));
}
static int kmask_read(const bbo_ctx* c, tcounters* tc, const uint8_t* bases, const int blen, const int pairnum, int* id0out,
                      uint32_t* gmask, const int64_t gbase) {
    const int k = c->k;
    *id0out = -1;
    if (blen < imax(1, (c->useShortKmers ? imin(k, c->mink) : k)) || c->storedKmers < 1) return 0;
    if ((c->a.skipR1 && pairnum == 0) || (c->a.skipR2 && pairnum == 1)) return 0;
    if (blen < k) return 0;                                                          /* :2154 */
    int64_t kmer = 0, rkmer = 0;
    int found = 0, len = 0, id0 = -1;
    const int trimPad = c->a.trimPad;
    const int nbits = blen + (trimPad > 0 ? trimPad : 0) + 1;
    uint64_t* bs = (uint64_t*)calloc((size_t)(nbits + 64) / 64 + 1, sizeof(uint64_t));
    const int minus = k - 1 - trimPad, plus = trimPad + 1;
    const int mfc = c->a.kmaskFullyCovered;
    if (mfc) bs_set(bs, 0, blen);                                                    /* :2163 */
    const int start = (c->a.restrictRight < 1 ? 0 : imax(0, blen - c->a.restrictRight));
    const int stop  = (c->a.restrictLeft  < 1 ? blen : imin(blen, c->a.restrictLeft));
    for (int i = start; i < stop; i++) {                                             /* :2171-2200 */
        uint8_t b = bases[i];
        int64_t x = num0(b), x2 = cnum0(b);
        kmer = (int64_t)((((uint64_t)kmer << 2) | (uint64_t)x) & (uint64_t)c->mask);
        rkmer = (int64_t)((((uint64_t)rkmer >> 2) | ((uint64_t)x2 << c->shift2)) & (uint64_t)c->mask);
        if (c->forbidNs && !is_fully_defined(b)) { len = 0; rkmer = 0; } else { len++; }
        if (i >= c->minlen) {
            int id;
            if (len >= c->minlen2) id = bbo_get_value(c, kmer, rkmer, c->kmask, i, k, c->qHammingDistance);
            else id = -1;
            if (id > 0) {
                if (id0 < 0) id0 = id;
                if (!mfc) bs_set(bs, imax(0, i - minus), i + plus);
                found++;
            } else if (mfc) bs_clear(bs, imax(0, i - minus), imin(nbits, i + plus));
        }
    }
    if (c->useShortKmers) {                                                          /* :2203-2291: always, both sides */
        {
            kmer = 0; rkmer = 0; len = 0;
            const int lim = imin(k, stop);
            for (int i = start; i < lim; i++) {
                uint8_t b = bases[i];
                int64_t x = num0(b), x2 = cnum0(b);
                kmer = (int64_t)((((uint64_t)kmer << 2) | (uint64_t)x) & (uint64_t)c->mask);
                rkmer = rkmer | (int64_t)((uint64_t)x2 << (2 * len));
                len++;
                if (len >= c->minminlen) {                                             /* len2>=minminlen; looked up from mink on */
                    const int id = (len >= c->mink) ? bbo_get_value(c, kmer, rkmer, c->lengthMasks[len], i, len, c->qHammingDistance2) : -1;
                    if (id > 0) {
                        if (id0 < 0) id0 = id;
                        if (!mfc) bs_set(bs, 0, imin(blen, i + trimPad + 1));
                        found++;
                    } else if (mfc) bs_clear(bs, 0, imin(blen, i + trimPad + 1));
                }
            }
        }
        {
            kmer = 0; rkmer = 0; len = 0;
            const int lim = imax(-1, stop - k);
            for (int i = stop - 1; i > lim; i--) {
                uint8_t b = bases[i];
                int64_t x = num0(b), x2 = cnum0(b);
                kmer = kmer | (int64_t)((uint64_t)x << (2 * len));
                rkmer = (int64_t)((((uint64_t)rkmer << 2) | (uint64_t)x2) & (uint64_t)c->mask);
                len++;
                if (len >= c->minminlen) {
                    const int id = (len >= c->mink) ? bbo_get_value(c, kmer, rkmer, c->lengthMasks[len], i, len, c->qHammingDistance2) : -1;
                    if (id > 0) {
                        if (id0 < 0) id0 = id;
                        if (!mfc) bs_set(bs, imax(0, i - trimPad), blen);
                        found++;
                    } else if (mfc) bs_clear(bs, imax(0, i - trimPad), blen);
                }
            }
        }
    }
    if (found == 0) { free(bs); return 0; }
    tc->scafReads[id0]++;
    tc->scafBases[id0] += blen;
    *id0out = id0;
    int cardinality = 0;
    for (int w = 0; w <= (nbits + 63) / 64; w++) cardinality += __builtin_popcountll(bs[w]);
    if (gmask) {
        for (int i = 0; i < blen; i++) {                                             /* :2309-2320: the bases the caller replaces */
            if ((bs[i >> 6] >> (i & 63)) & 1ULL) __atomic_fetch_or(&gmask[(gbase + i) >> 5], 1u << ((gbase + i) & 31), __ATOMIC_RELAXED);
        }
    }
    free(bs);
    return cardinality;
}

/* stream/Read.java:1673-1683 numValidKmers */
static int num_valid_kmers(const uint8_t* bases, const int blen, const int k) {
    int len = 0, counted = 0;
    for (int i = 0; i < blen; i++) {
        if (!is_fully_defined(bases[i])) len = 0; else len++;
        if (len >= k) counted++;
    }
    return counted;
}

/* bbduk/BBDukProcessorS.java:1602-1651 countCoveredBases */
static int count_covered_bases(const bbo_ctx* c, tcounters* tc, const uint8_t* bases, const int blen, const int pairnum,
                               const int minCoveredBases, int* idout) {
    const int k = c->k;
    *idout = -1;
    if (blen < k || c->storedKmers < 1) return 0;
    if ((c->a.skipR1 && pairnum == 0) || (c->a.skipR2 && pairnum == 1)) return 0;
    int64_t kmer = 0, rkmer = 0;
    int found = 0, len = 0, lastFound = -1;
    const int start = (c->a.restrictRight < 1 ? 0 : imax(0, blen - c->a.restrictRight));
    const int stop  = (c->a.restrictLeft  < 1 ? blen : imin(blen, c->a.restrictLeft));
    for (int i = start; i < stop; i++) {
        uint8_t b = bases[i];
        int64_t x = num0(b), x2 = cnum0(b);
        kmer = (int64_t)((((uint64_t)kmer << 2) | (uint64_t)x) & (uint64_t)c->mask);
        rkmer = (int64_t)((((uint64_t)rkmer >> 2) | ((uint64_t)x2 << c->shift2)) & (uint64_t)c->mask);
        if (c->forbidNs && !is_fully_defined(b)) { len = 0; rkmer = 0; } else { len++; }
        if (len >= c->minlen2 && i >= c->minlen) {
            const int id = bbo_get_value(c, kmer, rkmer, c->kmask, i, k, c->qHammingDistance);
            if (id > 0) {
                const int extra = imin(k, i - lastFound);
                found += extra;
                lastFound = i;
                if (found >= minCoveredBases) {
                    tc->scafReads[id]++;
                    tc->scafBases[id] += blen;
                    *idout = id;
                    return found;
                }
            }
        }
    }
    return found;
}

static tcounters main_tc(bbo_ctx* c) { tcounters t; memset(&t, 0, sizeof t); t.scafReads = c->scafReads; t.scafBases = c->scafBases; return t; }

int bbo_ktrim_read(bbo_ctx* c, const uint8_t* bases, int len, int pairnum, int* id0) {
    tcounters t = main_tc(c); int nl;
    return ktrim_read(c, &t, bases, len, pairnum, id0, &nl);
}
int bbo_count_set_kmers(bbo_ctx* c, const uint8_t* bases, int len, int pairnum, int maxBadKmers, int* id) {
    tcounters t = main_tc(c);
    return count_set_kmers(c, &t, bases, len, pairnum, maxBadKmers, id);
}

/* bbduk/BBDukProcessorS.java:778-1093,1431-1443 -- one pair (r2len<0 => unpaired), k-mer stage only.
 * Everything before the k-mer stage (junk/chastity/GC/force-trim...) and after it (tbo/qtrim/...) is off
 * in every BASELINE config and stays in the Java host. */
/* setDiscarded / isDiscarded with trimFailuresTo1bp (:1464-1482): a read that was to be discarded is cut to one base (if it is longer),
 * and from then on "discarded" means "exactly one base long" -- which also holds for a read that is one base long for any other reason */
#define TF1BP(d, len) do { if (c->trimFailuresTo1bp) { if ((d) && (len) > 1) (len) = 1; (d) = ((len) == 1); } } while (0)

static void process_pair(const bbo_ctx* c, tcounters* tc, const uint8_t* b1, int l1, const uint8_t* b2, int l2, int has2,
                         int32_t* a, int32_t* ids, uint8_t* fl, uint32_t* gmask, int64_t g1, int64_t g2, int32_t* xleft, int32_t* xright) {
    const int initialLength1 = l1, initialLength2 = has2 ? l2 : 0;
    const int pairCount = has2 ? 2 : 1;
    const int minlen1 = (int)((float)initialLength1 * c->a.minLenFraction > (float)c->a.minReadLength ?
                              (float)initialLength1 * c->a.minLenFraction : (float)c->a.minReadLength);   /* :812 */
    const int minlen2 = (int)((float)initialLength2 * c->a.minLenFraction > (float)c->a.minReadLength ?
                              (float)initialLength2 * c->a.minLenFraction : (float)c->a.minReadLength);   /* :813 */
    tc->counters[BBO_READS_IN] += pairCount;                                          /* :817-818 */
    tc->counters[BBO_BASES_IN] += initialLength1 + initialLength2;
    int d1 = 0, d2 = 0, remove = 0;
    int newLen1 = l1, newLen2 = initialLength2;
    const int ktrimN = c->a.ktrimN && !(c->a.ktrimLeft || c->a.ktrimRight);
    const int doKmerTrimming = c->storedKmers > 0 && (c->a.ktrimLeft || c->a.ktrimRight || ktrimN || c->a.ksplit);   /* :772 */
    const int doKmerFiltering = c->storedKmers > 0 && !doKmerTrimming;                      /* :773 */
    a[0] = 0; ids[0] = -1; if (has2) { a[1] = 0; ids[1] = -1; }
    if (xleft && xright) { xleft[0] = -1; xright[0] = -1; }
    if (doKmerTrimming && c->a.ksplit && !(c->a.ktrimLeft || c->a.ktrimRight || ktrimN)) {     /* :999-1013, 1028-1029 */
        int id0, lm, rm, trimmed, npl;
        const int split = has2 ? 0 : ksplit_read(c, tc, b1, l1, &id0, &lm, &rm, &trimmed, &npl);   /* assert(r2==null) */
        if (!has2) {
            a[0] = trimmed; ids[0] = id0;
            if (xleft && xright) { xleft[0] = lm; xright[0] = rm; }
            tc->counters[BBO_BASES_KTRIMMED] += trimmed;
            tc->counters[BBO_READS_KTRIMMED] += (trimmed > 0 ? 1 : 0);
            newLen1 = npl;                                                            /* r1.pairLength() after the split */
            remove = split;                                                           /* remove=(r1.mate!=null) */
        }
    } else if (doKmerTrimming && ktrimN) {                                                   /* :984-998, 1009-1016, 1028-1029 */
        int xsum = 0, rktsum = 0, id0;
        int x = kmask_read(c, tc, b1, l1, 0, &id0, gmask, g1);
        xsum += x; rktsum += (x > 0 ? 1 : 0);
        if (l1 < minlen1) d1 = 1;
        a[0] = x; ids[0] = id0;
        if (has2) {
            x = kmask_read(c, tc, b2, l2, 1, &id0, gmask, g2);
            xsum += x; rktsum += (x > 0 ? 1 : 0);
            if (l2 < minlen2) d2 = 1;
            a[1] = x; ids[1] = id0;
        }
        TF1BP(d1, newLen1); if (has2) TF1BP(d2, newLen2);
        if ((c->removePairsIfEitherBad && (d1 || d2)) || (d1 && (!has2 || d2))) remove = 1;   /* ktrimN: xsum/rktsum unchanged (:1011) */
        tc->counters[BBO_BASES_KTRIMMED] += xsum;
        tc->counters[BBO_READS_KTRIMMED] += rktsum;
    } else if (doKmerTrimming) {                                                      /* :948-1033 */
        int rlen1 = 0, rlen2 = 0, xsum = 0, rktsum = 0;
        const int tips = c->a.ktrimLeft && c->a.ktrimRight;                           /* :771, 954-967 */
        int xl1 = 0, xl2 = 0;
        {
            int id0; int x = tips ? ktrim_tips_read(c, tc, b1, l1, 0, &id0, &newLen1, &xl1) : ktrim_read(c, tc, b1, l1, 0, &id0, &newLen1);
            xsum += x; rktsum += (x > 0 ? 1 : 0); rlen1 = newLen1;
            if (rlen1 < minlen1) d1 = 1;
            a[0] = x; ids[0] = id0;
        }
        if (has2) {
            int id0; int x = tips ? ktrim_tips_read(c, tc, b2, l2, 1, &id0, &newLen2, &xl2) : ktrim_read(c, tc, b2, l2, 1, &id0, &newLen2);
            xsum += x; rktsum += (x > 0 ? 1 : 0); rlen2 = newLen2;
            if (rlen2 < minlen2) d2 = 1;
            a[1] = x; ids[1] = id0;
        }
        TF1BP(d1, newLen1); if (has2) TF1BP(d2, newLen2);                             /* rlen1 / rlen2 keep the lengths ktrim left (:960, 966) */
        /* shouldRemove (:1489-1492) */
        if ((c->removePairsIfEitherBad && (d1 || d2)) || (d1 && (!has2 || d2))) {
            xsum += (rlen1 + rlen2);                                                  /* :1011-1014 (!ktrimN) */
            rktsum = pairCount;
            remove = 1;
        } else if (c->a.ktrimRight && c->a.trimPairsEvenly && xsum > 0 && has2 && newLen1 != newLen2) {   /* :1021-1031 */
            int x;
            if (newLen1 > newLen2) { x = trim_to_position(newLen1, 0, newLen2 - 1, 1, &newLen1); a[0] += x; }
            else { x = trim_to_position(newLen2, 0, newLen1 - 1, 1, &newLen2); a[1] += x; }
            if (rktsum < 2) rktsum++;
            xsum += x;
        }
        tc->counters[BBO_BASES_KTRIMMED] += xsum;                                     /* :1028-1029 */
        tc->counters[BBO_READS_KTRIMMED] += rktsum;
        if (xleft) { xleft[0] = xl1; if (has2) xleft[1] = xl2; }
    } else if (doKmerFiltering && c->a.minCoveredFraction > 0) {                      /* :1038-1049 */
        int id;
        const int mc1 = (int)ceil((double)(c->a.minCoveredFraction * (float)l1));
        const int cov1 = count_covered_bases(c, tc, b1, l1, 0, mc1, &id);
        a[0] = cov1; ids[0] = id;
        if (cov1 >= mc1) d1 = 1;
        if (has2) {
            const int mc2 = (int)ceil((double)(c->a.minCoveredFraction * (float)l2));
            const int cov2 = count_covered_bases(c, tc, b2, l2, 1, mc2, &id);
            a[1] = cov2; ids[1] = id;
            if (cov2 >= mc2) d2 = 1;
        }
        TF1BP(d1, newLen1); if (has2) TF1BP(d2, newLen2);                             /* (a one-base read is "discarded" before it is looked at, :1040) */
        if ((c->removePairsIfEitherBad && (d1 || d2)) || (d1 && (!has2 || d2))) {
            remove = 1;
            tc->counters[BBO_READS_KFILTERED]++; tc->counters[BBO_BASES_KFILTERED] += initialLength1;
            if (has2) { tc->counters[BBO_READS_KFILTERED]++; tc->counters[BBO_BASES_KFILTERED] += initialLength2; }
        }
    } else if (doKmerFiltering) {                                                     /* :1035-1093 */
        int maxBadKmersR1 = c->a.maxBadKmers0, maxBadKmersR2 = c->a.maxBadKmers0;     /* :1056-1057 */
        if (c->a.minKmerFraction != 0) {                                              /* :1058-1062 */
            const int vk1 = num_valid_kmers(b1, l1, c->keff), vk2 = has2 ? num_valid_kmers(b2, l2, c->keff) : 0;
            maxBadKmersR1 = imax(c->a.maxBadKmers0, (int)((float)(vk1 - 1) * c->a.minKmerFraction));
            maxBadKmersR2 = imax(c->a.maxBadKmers0, (int)((float)(vk2 - 1) * c->a.minKmerFraction));
        }
        if (c->a.findBestMatch) {                                                     /* :1072-1078 */
            int f; const int av = find_best_match(c, tc, b1, l1, 0, maxBadKmersR1, &f);
            a[0] = f; ids[0] = av;
            if (av > 0) d1 = 1;
            if (has2) {
                const int bv = find_best_match(c, tc, b2, l2, 1, maxBadKmersR2, &f);
                a[1] = f; ids[1] = bv;
                if (bv > 0) d2 = 1;
            }
        } else {                                                                      /* :1064-1070 */
            int id; const int av = (c->kbig <= c->k ? count_set_kmers(c, tc, b1, l1, 0, maxBadKmersR1, &id) : count_set_kmers_big(c, tc, b1, l1, 0, maxBadKmersR1, &id));
            a[0] = av; ids[0] = id;
            if (av > maxBadKmersR1) d1 = 1;
            if (has2) {
                const int bv = (c->kbig <= c->k ? count_set_kmers(c, tc, b2, l2, 1, maxBadKmersR2, &id) : count_set_kmers_big(c, tc, b2, l2, 1, maxBadKmersR2, &id));
                a[1] = bv; ids[1] = id;
                if (bv > maxBadKmersR2) d2 = 1;
            }
        }
        TF1BP(d1, newLen1); if (has2) TF1BP(d2, newLen2);
        if ((c->removePairsIfEitherBad && (d1 || d2)) || (d1 && (!has2 || d2))) {
            remove = 1;
            tc->counters[BBO_READS_KFILTERED]++; tc->counters[BBO_BASES_KFILTERED] += initialLength1;
            if (has2) { tc->counters[BBO_READS_KFILTERED]++; tc->counters[BBO_BASES_KFILTERED] += initialLength2; }
        }
    }
    if (remove && c->trimFailuresTo1bp) remove = 0;                                   /* :1431: flagged, counted above, but not evicted */
    if (remove) {                                                                     /* :1431-1443 */
        tc->counters[BBO_READS_OUTM] += pairCount;
        tc->counters[BBO_BASES_OUTM] += newLen1 + newLen2;
    } else {
        tc->counters[BBO_READS_OUTU] += pairCount;
        tc->counters[BBO_BASES_OUTU] += newLen1 + newLen2;
    }
    fl[0] = (uint8_t)((d1 ? BBO_FLAG_DISCARDED : 0) | (remove ? BBO_FLAG_REMOVED : 0));
    if (has2) fl[1] = (uint8_t)((d2 ? BBO_FLAG_DISCARDED : 0) | (remove ? BBO_FLAG_REMOVED : 0));
}

typedef struct {
    const bbo_ctx* c; const uint8_t* bases; const int64_t* offsets; int64_t n; int paired;
    int32_t* out_a; int32_t* out_id; uint8_t* out_flags; uint32_t* out_mask; int32_t* out_left; int32_t* out_right; int64_t u0, u1; tcounters tc;
} job;
typedef struct { int32_t* n; int32_t* ids; int32_t* counts; int cap; } match_out;

static void* job_run(void* p) {
    job* j = (job*)p;
    const int step = j->paired ? 2 : 1;
    for (int64_t u = j->u0; u < j->u1; u++) {
        int64_t r = u * step;
        const uint8_t* b1 = j->bases + j->offsets[r]; int l1 = (int)(j->offsets[r + 1] - j->offsets[r]);
        const uint8_t* b2 = NULL; int l2 = 0;
        if (j->paired) { b2 = j->bases + j->offsets[r + 1]; l2 = (int)(j->offsets[r + 2] - j->offsets[r + 1]); }
        j->tc.mRead = r;
        process_pair(j->c, &j->tc, b1, l1, b2, l2, j->paired, j->out_a + r, j->out_id + r, j->out_flags + r, j->out_mask,
                     j->offsets[r], j->paired ? j->offsets[r + 1] : 0, j->out_left ? j->out_left + r : NULL, j->out_right ? j->out_right + r : NULL);
    }
    return NULL;
}

int bbo_process_batch(bbo_ctx* c, const uint8_t* bases, const int64_t* offsets, int64_t n, int paired,
                      int32_t* out_a, int32_t* out_id, uint8_t* out_flags, int nthreads) {
    return bbo_process_batch_mask(c, bases, offsets, n, paired, out_a, out_id, out_flags, NULL, nthreads);
}
int bbo_process_batch_mask(bbo_ctx* c, const uint8_t* bases, const int64_t* offsets, int64_t n, int paired,
                           int32_t* out_a, int32_t* out_id, uint8_t* out_flags, uint32_t* out_mask, int nthreads) {
    return bbo_process_batch_ex(c, bases, offsets, n, paired, out_a, out_id, out_flags, out_mask, NULL, nthreads);
}
static int process_batch_all(bbo_ctx* c, const uint8_t* bases, const int64_t* offsets, int64_t n, int paired,
                             int32_t* out_a, int32_t* out_id, uint8_t* out_flags, uint32_t* out_mask, int32_t* out_left, int32_t* out_right, int nthreads,
                             const match_out* mo);
int bbo_process_batch_ex(bbo_ctx* c, const uint8_t* bases, const int64_t* offsets, int64_t n, int paired,
                         int32_t* out_a, int32_t* out_id, uint8_t* out_flags, uint32_t* out_mask, int32_t* out_left, int nthreads) {
    return process_batch_all(c, bases, offsets, n, paired, out_a, out_id, out_flags, out_mask, out_left, NULL, nthreads, NULL);
}
int bbo_process_batch_matches(bbo_ctx* c, const uint8_t* bases, const int64_t* offsets, int64_t n, int paired,
                              int32_t* out_a, int32_t* out_id, uint8_t* out_flags, int max_ids, int32_t* out_nids,
                              int32_t* out_match_ids, int32_t* out_match_counts, int nthreads) {
    const match_out mo = { out_nids, out_match_ids, out_match_counts, max_ids };
    if (!c->a.findBestMatch || max_ids < 1) return -1;
    for (int64_t i = 0; i < n; i++) out_nids[i] = 0;
    return process_batch_all(c, bases, offsets, n, paired, out_a, out_id, out_flags, NULL, NULL, NULL, nthreads, &mo);
}
int bbo_process_batch_split(bbo_ctx* c, const uint8_t* bases, const int64_t* offsets, int64_t n,
                            int32_t* out_a, int32_t* out_id, uint8_t* out_flags, int32_t* out_leftmost, int32_t* out_rightmost, int nthreads) {
    return process_batch_all(c, bases, offsets, n, 0, out_a, out_id, out_flags, NULL, out_leftmost, out_rightmost, nthreads, NULL);
}
static int process_batch_all(bbo_ctx* c, const uint8_t* bases, const int64_t* offsets, int64_t n, int paired,
                             int32_t* out_a, int32_t* out_id, uint8_t* out_flags, uint32_t* out_mask, int32_t* out_left, int32_t* out_right, int nthreads,
                             const match_out* mo) {
    if (paired && (n & 1)) return -1;
    if (nthreads < 1) nthreads = 1;
    const int64_t units = paired ? n / 2 : n;
    if (units < nthreads) nthreads = units > 0 ? (int)units : 1;
    job* jobs = (job*)calloc((size_t)nthreads, sizeof(job));
    pthread_t* th = (pthread_t*)calloc((size_t)nthreads, sizeof(pthread_t));
    const int ns = c->numScaffolds;
    for (int t = 0; t < nthreads; t++) {
        job* j = &jobs[t];
        j->c = c; j->bases = bases; j->offsets = offsets; j->n = n; j->paired = paired;
        j->out_a = out_a; j->out_id = out_id; j->out_flags = out_flags; j->out_mask = out_mask; j->out_left = out_left; j->out_right = out_right;
        j->u0 = units * t / nthreads; j->u1 = units * (t + 1) / nthreads;
        if (mo) { j->tc.mN = mo->n; j->tc.mIds = mo->ids; j->tc.mCnt = mo->counts; j->tc.mCap = mo->cap; }
        j->tc.scafReads = (int64_t*)calloc((size_t)ns, sizeof(int64_t));             /* thread-local copies (:272-277) */
        j->tc.scafBases = (int64_t*)calloc((size_t)ns, sizeof(int64_t));
        if (nthreads > 1) pthread_create(&th[t], NULL, job_run, j); else job_run(j);
    }
    for (int t = 0; t < nthreads; t++) {
        if (nthreads > 1) pthread_join(th[t], NULL);
        for (int i = 0; i < BBO_NCOUNTERS; i++) c->counters[i] += jobs[t].tc.counters[i];   /* BBDukProcessorS.add :300-342 */
        for (int i = 0; i < ns; i++) { c->scafReads[i] += jobs[t].tc.scafReads[i]; c->scafBases[i] += jobs[t].tc.scafBases[i]; }
        free(jobs[t].tc.scafReads); free(jobs[t].tc.scafBases); free(jobs[t].tc.countArray); free(jobs[t].tc.idList);
    }
    free(jobs); free(th);
    return 0;
}

int bbo_counters_len(const bbo_ctx* c) { return BBO_NCOUNTERS + 2 * c->numScaffolds; }
void bbo_get_counters(const bbo_ctx* c, int64_t* out) {
    memcpy(out, c->counters, sizeof c->counters);
    memcpy(out + BBO_NCOUNTERS, c->scafReads, sizeof(int64_t) * (size_t)c->numScaffolds);
    memcpy(out + BBO_NCOUNTERS + c->numScaffolds, c->scafBases, sizeof(int64_t) * (size_t)c->numScaffolds);
}
void bbo_reset_counters(bbo_ctx* c) {
    memset(c->counters, 0, sizeof c->counters);
    memset(c->scafReads, 0, sizeof(int64_t) * (size_t)c->scafCap);
    memset(c->scafBases, 0, sizeof(int64_t) * (size_t)c->scafCap);
}
