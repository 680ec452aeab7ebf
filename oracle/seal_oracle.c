/*
 * CPU restatement of jgi/Seal.java's k-mer path (TEST INFRASTRUCTURE: the checker for the HIP operator seal_batch*,
 * imported by tests/, never by the product).  Plain C, one thread, written to be read next to the reference:
 *
 *   loading     Seal.LoadThread.addToMap / mutate                         jgi/Seal.java:1760-1945
 *               values per k-mer: kmer/HashArrayHybridFast.insertValue + structures/IntList3.insertIntoList
 *               (ascending and unique because one loader thread sees the scaffolds in id order, Seal.java:108)
 *   per pair    Seal.ProcessThread.run, the length filter and the k-mer branch  jgi/Seal.java:2011-2290
 *   scan        findBestMatch(Read, sets, int[] hits, IntList idList)           jgi/Seal.java:2864-2909 (countArray form, the default)
 *   lists       condenseLoose(int[], IntList, IntList) :2654-2667, filterTopScaffolds_withClearzone :2697-2706
 *   assignment  assignTogether :2386-2452, assignIndependently :2462-2612, numKmers :1198-1207
 *
 * parity unpinned: the reference is Java and the image has no JVM, and the reference ships no test vectors for Seal; the
 * restatement follows the source line by line and the HIP path is compared with it.
 *
 * Not restated (the operator refuses them): qhdist>0, edist, processcontainedref, countvector=t (sorted id lists), rename,
 * taxonomy, barcodes, gene sets, quality trimming / filtering other than the length rule.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

typedef struct {
    int k, maskMiddle, midMaskLen, rcomp, forbidN, hdist, refSkip;
    int restrictLeft, restrictRight, qSkip, speed;
    int matchMode;            /* 0 all, 1 first, 2 unique   (Seal.MATCH_*) */
    int ambigMode;            /* 0 first, 1 all, 2 random, 3 toss (order of this restatement, see seal_gpu.h) */
    int keepPairsTogether, minKmerHits;
    float minKmerFraction;
    int clearzone;
    float clearzoneFraction;
    int minReadLength, maxReadLength;
    float minLenFraction;
    int requireBothBad;
} so_args;

enum { SO_READS_IN, SO_BASES_IN, SO_FRAGS_IN, SO_READS_MATCHED, SO_BASES_MATCHED, SO_READS_UNMATCHED, SO_BASES_UNMATCHED,
       SO_READS_QFILTERED, SO_BASES_QFILTERED, SO_READS_QTRIMMED, SO_NCOUNTERS };

typedef struct { int64_t key; int32_t id; } so_pair;

typedef struct {
    so_args a;
    int64_t middleMask, kmask, mask;
    int forbidNs;
    so_pair* pairs; int64_t npairs, cap; int sorted;
    int numScaffolds;                       /* scaffold 0 is the fake first one (Seal.java:134-139) */
    int* scafKmers;
    int64_t counters[SO_NCOUNTERS];
    int64_t *scafReads, *scafBases, *scafFrags, *scafAmbig;
    int* countArray;                        /* hits per scaffold, zero between reads */
} so_ctx;

static const int8_t* base_tab(void) {
    static int8_t t[256]; static int init = 0;
    if (!init) { memset(t, -1, sizeof t); t['A'] = t['a'] = 0; t['C'] = t['c'] = 1; t['G'] = t['g'] = 2; t['T'] = t['t'] = 3; t['U'] = t['u'] = 3; init = 1; }
    return t;
}
static inline int64_t num(uint8_t b)   { return base_tab()[b]; }                               /* AminoAcid.baseToNumber */
static inline int64_t num0(uint8_t b)  { int v = base_tab()[b]; return v < 0 ? 0 : v; }        /* baseToNumber0 */
static inline int64_t cnum(uint8_t b)  { int v = base_tab()[b]; return v < 0 ? -1 : 3 - v; }   /* baseToComplementNumber */
static inline int64_t cnum0(uint8_t b) { int v = base_tab()[b]; return v < 0 ? 0 : 3 - v; }    /* baseToComplementNumber0 (dna/AminoAcid.java:1306-1311: 0 for anything else) */

void so_default_args(so_args* a) {                                   /* Seal.java:104-131, 3088-3098 */
    memset(a, 0, sizeof *a);
    a->k = 31; a->maskMiddle = 1; a->rcomp = 1; a->qSkip = 1; a->matchMode = 0; a->ambigMode = 2; a->keepPairsTogether = 1;
    a->minKmerHits = 1; a->minReadLength = 10; a->maxReadLength = 0x7FFFFFFF;
}

so_ctx* so_create(const so_args* a) {
    if (a->k < 1 || a->k > 31) return NULL;
    so_ctx* c = (so_ctx*)calloc(1, sizeof *c);
    c->a = *a;
    if (c->a.maskMiddle) c->a.midMaskLen = c->a.midMaskLen > 0 ? c->a.midMaskLen : 2 - (c->a.k & 1);   /* :548-552 */
    else c->a.midMaskLen = 0;
    if (c->a.maskMiddle) {                                            /* :561-569 */
        if (!(c->a.k > c->a.midMaskLen + 1)) { free(c); return NULL; }
        const int bits = c->a.midMaskLen * 2, shift = ((c->a.k - c->a.midMaskLen) / 2) * 2;
        c->middleMask = (int64_t)~((~((uint64_t)-1 << bits)) << shift);
    } else c->middleMask = -1;
    c->forbidNs = (a->forbidN || a->hdist < 1);                       /* :492 */
    c->kmask = (int64_t)1 << (2 * a->k);
    c->mask = (int64_t)(~(~0ULL << (2 * a->k)));
    c->numScaffolds = 1;
    c->scafKmers = (int*)calloc(1, sizeof(int));
    return c;
}
void so_destroy(so_ctx* c) {
    if (!c) return;
    free(c->pairs); free(c->scafKmers); free(c->scafReads); free(c->scafBases); free(c->scafFrags); free(c->scafAmbig); free(c->countArray); free(c);
}

static int64_t rcomp_k(int64_t kmer, int len) {
    int64_t r = 0;
    for (int i = 0; i < len; i++) { r = (r << 2) | (3 - (kmer & 3)); kmer >>= 2; }
    return r;
}
static inline int64_t to_value(const so_ctx* c, int64_t kmer, int64_t rkmer, int64_t lengthMask) {   /* :2971-2975 */
    const int64_t v = c->a.rcomp ? (kmer > rkmer ? kmer : rkmer) : kmer;
    return (v & c->middleMask) | lengthMask;
}
static inline int passes_speed(const so_ctx* c, int64_t key) { return c->a.speed < 1 || ((key & INT64_MAX) % 17) >= c->a.speed; }

static void push_pair(so_ctx* c, int64_t key, int id) {              /* map.set(key, id): the set of ids of a key */
    if (c->npairs == c->cap) { c->cap = c->cap ? 2 * c->cap : 1024; c->pairs = (so_pair*)realloc(c->pairs, (size_t)c->cap * sizeof(so_pair)); }
    c->pairs[c->npairs].key = key; c->pairs[c->npairs].id = id; c->npairs++; c->sorted = 0;
}
static void mutate(so_ctx* c, int64_t kmer, int64_t rkmer, int len, int id, int dist) {     /* :1890-1945, substitutions */
    push_pair(c, to_value(c, kmer, rkmer, (int64_t)1 << (2 * len)), id);
    if (dist > 0) {
        for (int j = 0; j < 4; j++) for (int i = 0; i < len; i++) {
            const int64_t temp = (kmer & ~((int64_t)3 << (2 * i))) | ((int64_t)j << (2 * i));
            if (temp != kmer) mutate(c, temp, rcomp_k(temp, len), len, id, dist - 1);
        }
    }
}
/* LoadThread.addToMap(Read, skip) :1760-1826; returns the scaffold's id */
int so_add_ref_sequence(so_ctx* c, const uint8_t* bases, int64_t blen) {
    const int id = c->numScaffolds++;
    c->scafKmers = (int*)realloc(c->scafKmers, (size_t)c->numScaffolds * sizeof(int));
    c->scafKmers[id] = 0;
    const int k = c->a.k, shift2 = 2 * k - 2, skip = c->a.refSkip;
    if (!bases || blen < k) return id;
    int64_t kmer = 0, rkmer = 0; int len = 0, total = 0;
    for (int64_t i = 0; i < blen; i++) {
        const int64_t x = num(bases[i]), x2 = cnum(bases[i]);
        kmer = ((kmer << 2) | x) & c->mask;
        rkmer = (int64_t)(((uint64_t)rkmer >> 2) | ((uint64_t)x2 << shift2)) & c->mask;
        if (x < 0) { len = 0; rkmer = 0; } else len++;
        if (len >= k) {
            total++;
            if (skip > 1 && (len % skip) != 0) continue;
            if (c->a.hdist == 0) {
                const int64_t key = to_value(c, kmer, rkmer, c->kmask);
                if (!passes_speed(c, key)) continue;                  /* failsSpeed: only on this branch (:1848) */
                push_pair(c, key, id);
            } else mutate(c, kmer, rkmer, k, id, c->a.hdist);
        }
    }
    c->scafKmers[id] = total;
    return id;
}
static int cmp_pair(const void* x, const void* y) {
    const so_pair* a = (const so_pair*)x; const so_pair* b = (const so_pair*)y;
    if (a->key != b->key) return a->key < b->key ? -1 : 1;
    return a->id < b->id ? -1 : (a->id > b->id ? 1 : 0);
}
void so_finalize(so_ctx* c) {
    if (!c->sorted) {
        qsort(c->pairs, (size_t)c->npairs, sizeof(so_pair), cmp_pair);
        int64_t w = 0;
        for (int64_t i = 0; i < c->npairs; i++) if (w == 0 || cmp_pair(&c->pairs[i], &c->pairs[w - 1]) != 0) c->pairs[w++] = c->pairs[i];
        c->npairs = w; c->sorted = 1;
    }
    const size_t n = (size_t)c->numScaffolds;
    free(c->scafReads); free(c->scafBases); free(c->scafFrags); free(c->scafAmbig); free(c->countArray);
    c->scafReads = (int64_t*)calloc(n, 8); c->scafBases = (int64_t*)calloc(n, 8); c->scafFrags = (int64_t*)calloc(n, 8); c->scafAmbig = (int64_t*)calloc(n, 8);
    c->countArray = (int*)calloc(n, sizeof(int));
}
int so_num_scaffolds(const so_ctx* c) { return c->numScaffolds; }
int64_t so_num_pairs(const so_ctx* c) { return c->npairs; }
int64_t so_dump_pairs(const so_ctx* c, int64_t* keys, int32_t* ids, int64_t cap) {
    const int64_t n = c->npairs < cap ? c->npairs : cap;
    for (int64_t i = 0; i < n; i++) { keys[i] = c->pairs[i].key; ids[i] = c->pairs[i].id; }
    return c->npairs;
}
void so_counters(const so_ctx* c, int64_t* out) { memcpy(out, c->counters, sizeof c->counters); }
void so_scaffold_counts(const so_ctx* c, int which, int64_t* out) {
    const int64_t* s = which == 0 ? c->scafReads : which == 1 ? c->scafBases : which == 2 ? c->scafFrags : c->scafAmbig;
    memcpy(out, s, (size_t)c->numScaffolds * 8);
}
void so_reset_counters(so_ctx* c) {
    memset(c->counters, 0, sizeof c->counters);
    const size_t n = (size_t)c->numScaffolds * 8;
    memset(c->scafReads, 0, n); memset(c->scafBases, 0, n); memset(c->scafFrags, 0, n); memset(c->scafAmbig, 0, n);
}

/* the ids of a key: [*first, *first + return) in c->pairs, ascending */
static int get_values(const so_ctx* c, int64_t key, int64_t* first) {
    int64_t lo = 0, hi = c->npairs;
    while (lo < hi) { const int64_t m = (lo + hi) >> 1; if (c->pairs[m].key < key) lo = m + 1; else hi = m; }
    int n = 0;
    while (lo + n < c->npairs && c->pairs[lo + n].key == key) n++;
    *first = lo;
    return n;
}

typedef struct { int* v; int n, cap; } ilist;
static void il_add(ilist* l, int x) { if (l->n == l->cap) { l->cap = l->cap ? 2 * l->cap : 16; l->v = (int*)realloc(l->v, (size_t)l->cap * sizeof(int)); } l->v[l->n++] = x; }

/* findBestMatch(Read, sets, int[] hits, IntList idList) :2864-2909 */
static int find_best_match(so_ctx* c, const uint8_t* bases, int blen, ilist* idList) {
    if (!bases || c->npairs < 1) return 0;
    const int k = c->a.k, minlen = k - 1, minlen2 = c->a.maskMiddle ? (k - c->a.midMaskLen) / 2 : k, shift2 = 2 * k - 2;
    if (blen < k) return -1;
    const int start = c->a.restrictRight < 1 ? 0 : (blen - c->a.restrictRight > 0 ? blen - c->a.restrictRight : 0);
    const int stop = c->a.restrictLeft < 1 ? blen : (blen < c->a.restrictLeft ? blen : c->a.restrictLeft);
    int64_t kmer = 0, rkmer = 0; int found = 0, len = 0;
    for (int i = start; i < stop; i++) {
        const uint8_t b = bases[i];
        const int64_t x = num0(b), x2 = cnum0(b);
        kmer = ((kmer << 2) | x) & c->mask;
        rkmer = (int64_t)(((uint64_t)rkmer >> 2) | ((uint64_t)x2 << shift2)) & c->mask;
        if (b == 'N' && c->forbidNs) { len = 0; rkmer = 0; } else len++;
        if (len >= minlen2 && i >= minlen) {
            if (c->a.qSkip > 1 && (i % c->a.qSkip) != 0) continue;     /* getValuesInner :2792 */
            const int64_t key = to_value(c, kmer, rkmer, c->kmask);
            if (!passes_speed(c, key)) continue;
            int64_t first; const int nv = get_values(c, key, &first);
            if (nv > 0) {
                for (int q = 0; q < nv; q++) {
                    const int id = c->pairs[first + q].id;
                    if (++c->countArray[id] == 1) il_add(idList, id);
                }
                found++;
                if (c->a.matchMode == 1 || (c->a.matchMode == 2 && nv == 1)) break;
            }
        }
    }
    return found;
}
static int condense(so_ctx* c, const ilist* packed, ilist* counts) {  /* condenseLoose(int[], IntList, IntList) */
    counts->n = 0; int max = 0;
    for (int i = 0; i < packed->n; i++) { const int p = packed->v[i], cc = c->countArray[p]; il_add(counts, cc); c->countArray[p] = 0; if (cc > max) max = cc; }
    return max;
}
static void filter_top(const ilist* packed, const ilist* counts, ilist* out, int max, int cz) {   /* _withClearzone */
    out->n = 0;
    if (packed->n < 1) return;
    const int thresh = (max - cz) > 1 ? (max - cz) : 1;
    for (int i = 0; i < packed->n; i++) if (counts->v[i] >= thresh) il_add(out, packed->v[i]);
}
static int num_valid_kmers(const uint8_t* bases, int blen, int k) {   /* stream/Read.java numValidKmers */
    if (!bases || blen < k) return 0;
    int len = 0, counted = 0;
    for (int i = 0; i < blen; i++) { if (num(bases[i]) < 0) len = 0; else len++; if (len >= k) counted++; }
    return counted;
}
static int num_kmers(int l1, int l2, int has2, int k) { int x = l1 - k + 1 > 0 ? l1 - k + 1 : 0; if (has2) x += l2 - k + 1 > 0 ? l2 - k + 1 : 0; return x; }
static int cmp_int(const void* a, const void* b) { return *(const int*)a - *(const int*)b; }

/* the choice among `sites` final scaffolds (assignTogether / assignIndependently): [start, stop) of the (possibly sorted) list */
static void choose(const so_ctx* c, ilist* fin, int64_t numericID, int* start, int* stop) {
    const int sites = fin->n;
    if (sites < 2 || c->a.ambigMode == 1) { *start = 0; *stop = sites; }
    else if (c->a.ambigMode == 3) { *start = *stop = 0; }
    else if (c->a.ambigMode == 0) { qsort(fin->v, (size_t)fin->n, sizeof(int), cmp_int); *start = 0; *stop = 1; }
    else { *start = (int)(numericID % sites); *stop = *start + 1; }
}

/*
 * One pair (b2 == NULL: one read) through ProcessThread.run's length rule and k-mer branch.
 * out_ids[0..cap): the assigned scaffolds (read 1's, then read 2's when keepPairsTogether is off); out_info[0] = sites, [1] = assigned,
 * [2] = max (max1), [3] = max2, [4] = removed by the length rule, [5] / [6] = assigned to read 1 / read 2 and [7] / [8] = their
 * finalList sizes when keepPairsTogether is off.  Returns `assigned`.
 */
int so_process(so_ctx* c, const uint8_t* b1, int l1, const uint8_t* b2, int l2, int64_t numericID, int* out_ids, int cap, int* out_info) {
    const int has2 = b2 != NULL;
    const int pairCount = 1 + has2, pairLen = l1 + (has2 ? l2 : 0);
    c->counters[SO_FRAGS_IN]++; c->counters[SO_READS_IN] += pairCount; c->counters[SO_BASES_IN] += pairLen;
    const int rieb = !c->a.requireBothBad;
    const float f1 = (float)l1 * c->a.minLenFraction, f2 = (float)l2 * c->a.minLenFraction;
    const int minlen1 = (int)(f1 > (float)c->a.minReadLength ? f1 : (float)c->a.minReadLength);
    const int minlen2 = (int)(f2 > (float)c->a.minReadLength ? f2 : (float)c->a.minReadLength);
    const int d1 = l1 < minlen1 || l1 > c->a.maxReadLength, d2 = has2 && (l2 < minlen2 || l2 > c->a.maxReadLength);
    int sites = 0, assigned = 0, max1 = 0, max2 = 0, nout = 0, asgR[2] = {0, 0}, sitesR[2] = {0, 0};
    const int remove = (rieb && (d1 || d2)) || (d1 && (!has2 || d2));          /* :2125-2130 */
    if (remove) {
        c->counters[SO_BASES_QFILTERED] += pairLen; c->counters[SO_READS_QTRIMMED] += pairCount;      /* (sic) :2127-2128 */
        c->counters[SO_BASES_QFILTERED] += pairLen; c->counters[SO_READS_QFILTERED] += pairCount;     /* :2182-2190 */
    } else {
        ilist id1 = {0}, id2 = {0}, cnt1 = {0}, cnt2 = {0}, fin1 = {0}, fin2 = {0};
        if (c->a.keepPairsTogether) {
            find_best_match(c, b1, l1, &id1);
            if (has2) find_best_match(c, b2, l2, &id1);
            max1 = condense(c, &id1, &cnt1);
            int cz = c->a.clearzone;
            if (c->a.clearzoneFraction > 0) {
                const int v = (int)ceil((double)(c->a.clearzoneFraction * (float)(num_valid_kmers(b1, l1, c->a.k) + (has2 ? num_valid_kmers(b2, l2, c->a.k) : 0))));
                if (v > cz) cz = v;
            }
            filter_top(&id1, &cnt1, &fin1, max1, cz);
            sites = fin1.n;
            const int mh = (int)(c->a.minKmerFraction * (float)num_kmers(l1, l2, has2, c->a.k));
            const int minhits = c->a.minKmerHits > mh ? c->a.minKmerHits : mh;
            if (max1 >= minhits) {                                      /* assignTogether */
                int start, stop; choose(c, &fin1, numericID, &start, &stop);
                for (int j = start; j < stop; j++) {
                    const int id = fin1.v[j];
                    if (nout < cap) out_ids[nout] = id;
                    nout++;
                    c->scafReads[id] += pairCount; c->scafBases[id] += pairLen; c->scafFrags[id]++;
                    if (sites > 1) c->scafAmbig[id] += pairCount;
                }
                if (start < stop) { c->counters[SO_READS_MATCHED] += pairCount; c->counters[SO_BASES_MATCHED] += pairLen; }
                else { c->counters[SO_READS_UNMATCHED] += pairCount; c->counters[SO_BASES_UNMATCHED] += pairLen; }
                assigned = stop - start;
            } else { c->counters[SO_READS_UNMATCHED] += pairCount; c->counters[SO_BASES_UNMATCHED] += pairLen; assigned = 0; }
        } else {
            find_best_match(c, b1, l1, &id1);
            max1 = condense(c, &id1, &cnt1);
            int cz = c->a.clearzone;
            if (c->a.clearzoneFraction > 0) { const int v = (int)ceil((double)(c->a.clearzoneFraction * (float)num_valid_kmers(b1, l1, c->a.k))); if (v > cz) cz = v; }
            filter_top(&id1, &cnt1, &fin1, max1, cz);
            if (has2) {
                find_best_match(c, b2, l2, &id2);
                max2 = condense(c, &id2, &cnt2);
                cz = c->a.clearzone;
                if (c->a.clearzoneFraction > 0) { const int v = (int)ceil((double)(c->a.clearzoneFraction * (float)num_valid_kmers(b2, l2, c->a.k))); if (v > cz) cz = v; }
                filter_top(&id2, &cnt2, &fin2, max2, cz);
            }
            sites = fin1.n + fin2.n; sitesR[0] = fin1.n; sitesR[1] = fin2.n;
            for (int r = 0; r < 1 + has2; r++) {                        /* assignIndependently */
                ilist* fin = r ? &fin2 : &fin1;
                const int mx = r ? max2 : max1, L = r ? l2 : l1;
                const int mh = (int)(c->a.minKmerFraction * (float)num_kmers(L, 0, 0, c->a.k));
                const int minhits = c->a.minKmerHits > mh ? c->a.minKmerHits : mh;
                if (mx < minhits) continue;                             /* (neither matched nor unmatched is counted) */
                const int s = fin->n;
                int start, stop; choose(c, fin, numericID, &start, &stop);
                for (int j = start; j < stop; j++) {
                    const int id = fin->v[j];
                    if (nout < cap) out_ids[nout] = id;
                    nout++;
                    c->scafReads[id]++; c->scafBases[id] += L;
                    if (r ? (max2 > max1) : (max1 >= max2)) c->scafFrags[id]++;
                    if (s > 1) c->scafAmbig[id]++;
                }
                if (start < stop) { c->counters[SO_READS_MATCHED]++; c->counters[SO_BASES_MATCHED] += L; assigned += stop - start; asgR[r] = stop - start; }
                else { c->counters[SO_READS_UNMATCHED]++; c->counters[SO_BASES_UNMATCHED] += L; }
            }
        }
        free(id1.v); free(id2.v); free(cnt1.v); free(cnt2.v); free(fin1.v); free(fin2.v);
    }
    if (out_info) { out_info[0] = sites; out_info[1] = assigned; out_info[2] = max1; out_info[3] = max2; out_info[4] = remove;
                    out_info[5] = asgR[0]; out_info[6] = asgR[1]; out_info[7] = sitesR[0]; out_info[8] = sitesR[1]; }
    return assigned;
}
