/*
 * bbduk_oracle.h -- CPU ORACLE for the BBDuk k-mer matching path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference's algorithm (BBTools v40.02, Java) for the one
 * hot path this repo accelerates.  It exists so that the HIP path can be checked bit-for-bit.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product
 * (bbtools_amd/) never links, imports or calls anything in oracle/.
 *
 * PARITY UNPINNED: the reference is Java, there is no JVM in the build image, and the reference
 * ships no tests / golden vectors for bbduk/ or kmer/ (SURVEY.md §4, §8c).  The oracle is therefore
 * pinned only (a) against hand-derived known-answer values, (b) against the assertion identities
 * the reference itself checks under -ea, and (c) by differential testing against a second,
 * structurally different restatement (oracle/spec.py: string-based closed form).
 *
 * All citations are relative to /root/reference/current/ .
 */
#ifndef BBDUK_ORACLE_H
#define BBDUK_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* User-level arguments, i.e. the bbduk.sh key=value flags that reach this path
 * (bbduk/BBDukParser.java:448-874).  Fill with bbo_default_args() first. */
typedef struct bbo_args {
    int k;              /* k=            (default 27 when absent: BBDukParser.java:163) */
    int mink;           /* mink=         (-1 = off; :1230) */
    int hdist, hdist2;  /* hdist= hdist2= (hdist2=-1 -> hdist; :130) */
    int edist, edist2;  /* edist= edist2= */
    int qhdist, qhdist2;/* qhdist= qhdist2= */
    int maskMiddle;     /* mm=           (default t; :1091) */
    int midMaskLen;     /* mml=          (0 -> 2-(k&1); :232-236) */
    int rcomp;          /* rcomp=        (default t; :1208) */
    int forbidN;        /* forbidn=      (default f) */
    int ktrimRight;     /* ktrim=r */
    int ktrimLeft;      /* ktrim=l */
    int maxBadKmers0;   /* mbk= / mkh=-1 (default 0; :1232) */
    int minReadLength;  /* minlen=       (default 10; :437) */
    float minLenFraction;/* mlf=         (default 0;  :439) */
    int requireBothBad; /* rieb=f  ->  removePairsIfEitherBad=false (:109) */
    int trimPad;        /* tp= */
    int ktrimExclusive; /* ktrimexclusive= */
    int restrictLeft, restrictRight; /* restrictleft= restrictright= */
    int skipR1, skipR2; /* skipr1= skipr2= */
    int minSkip, maxSkip;/* rskip/minskip/maxskip (default 1,1) */
    int trimPairsEvenly;/* tpe            (BBDukProcessorS.java:1021-1031) */
    int qSkip;          /* qskip=         (default 1; BBDukIndexMod.java:494) */
    int speed;          /* speed=         (default 0; query-side gate, BBDukIndexMod.java:506,562) */
    float minKmerFraction;    /* mkf=     (default 0; BBDukProcessorS.java:1055-1062) */
    float minCoveredFraction; /* mcf=     (default 0; :1038-1049, countCoveredBases :1602-1651) */
    int ktrimN;         /* ktrim=n / kmask= (BBDukProcessorS.java:2149-2323 with kmaskFullyCovered=false) */
    int kbig;           /* the k of the command line when it exceeds 31 (then k=31; BBDukParser.java:164-165), else <= k */
    int findBestMatch;  /* findbestmatch / fbm (BBDukProcessorS.java:1659-1719; rename is not restated) */
    int ksplit;         /* ksplit=t (BBDukProcessorS.java:2332-2506; unpaired reads) */
    int kmaskFullyCovered; /* kmaskfullycovered / mfc (:2163, 2193-2195, 2243-2245, 2286-2288) */
    int trimFailuresTo1bp; /* trimfailures / trimfailuresto1bp (BBDukParser.java:105-109, 774; BBDukProcessorS.java:1431, 1464-1488): a discarded read
                              is cut to its first base instead, "discarded" means "one base long", and nothing is evicted */
} bbo_args;

#define BBO_NCOUNTERS 16
enum { BBO_READS_IN=0, BBO_BASES_IN, BBO_READS_KTRIMMED, BBO_BASES_KTRIMMED,
       BBO_READS_KFILTERED, BBO_BASES_KFILTERED, BBO_READS_OUTU, BBO_BASES_OUTU,
       BBO_READS_OUTM, BBO_BASES_OUTM };

#define BBO_FLAG_DISCARDED 1   /* this read was setDiscarded()                 */
#define BBO_FLAG_REMOVED   2   /* its pair was removed (shouldRemove -> remove) */

typedef struct bbo_ctx bbo_ctx;

void     bbo_default_args(bbo_args* a);
/* Create a context: derives all constants exactly as BBDukParser does. Returns NULL on bad args. */
bbo_ctx* bbo_create(const bbo_args* a);
void     bbo_destroy(bbo_ctx* c);

/* Derived constants, for known-answer tests.  name in {"k","mink","hdist","hdist2","forbidNs","minlen",
 * "minlen2","shift2","mask","kmask","middleMask","useShortKmers","maskMiddle","midMaskLen","kfilter"} */
int64_t  bbo_constant(const bbo_ctx* c, const char* name);

/* Primitive restatements, exported for known-answer tests. */
int64_t  bbo_rcomp(int64_t kmer, int k);                 /* dna/AminoAcid.java:585-601 */
int      bbo_base_to_number(int b);                      /* dna/AminoAcid.java:1284-1298 (-1 undefined) */
int      bbo_base_to_number0(int b);
int      bbo_base_to_complement_number0(int b);

/* Reference loading (BBDukLoader.spawnLoadThreads + LoadThread.addToMap).  Sequences are given in file
 * order; scaffold ids are assigned 1,2,3...  Returns number of keys added (sum of setIfNotPresent). */
int64_t  bbo_add_ref_sequence(bbo_ctx* c, const uint8_t* bases, int64_t len);
/* Convenience: parse a FASTA file (plain text) and add every record of length>=1. Returns #records or <0. */
int      bbo_load_fasta(bbo_ctx* c, const char* path);
int      bbo_num_scaffolds(const bbo_ctx* c);            /* scaffoldNames.size()  (ids are 1..n-1) */
int64_t  bbo_stored_kmers(const bbo_ctx* c);

/* Table access. */
/* tests only: walk a shorter probe window than the reference's 60 cells (process-wide; set before loading references) */
void     bbo_test_set_probe_window(int n);
int      bbo_table_get(const bbo_ctx* c, int64_t key);   /* AbstractKmerTable.getValue: -1 if absent */
/* Image of one HashArray1D way, the arrays a JVM caller would hand to bbduk_upload_table_way():
 * array() (kmer/HashArray.java:672), values() (kmer/HashArray1D.java:407), victims().toList(). */
int      bbo_num_ways(const bbo_ctx* c);
int      bbo_way_image(const bbo_ctx* c, int way, int* prime, int64_t* ncells,
                       const int64_t** keys, const int32_t** values,
                       int64_t* nvictims, const int64_t** vkeys, const int32_t** vvals);
/* Flat (key,value) dump of everything stored; returns count (call with NULLs to size). */
int64_t  bbo_dump_pairs(const bbo_ctx* c, int64_t* keys, int32_t* values, int64_t cap);

/* Index.getValue (BBDukIndexMod.java:462-520) incl. query-side Hamming expansion. */
int      bbo_get_value(const bbo_ctx* c, int64_t kmer, int64_t rkmer, int64_t lengthMask, int qPos, int len, int qHDist);

/* Per-read scans, exactly as the reference's private methods; they do NOT mutate anything but counters.
 * ktrim: returns bases trimmed x (BBDukProcessorS.java:1806-1811,1993-2140); *id0 = first hit id or -1.
 * pairnum = 0/1.  scaffold counters inside ctx (thread 0) are bumped.  */
int      bbo_ktrim_read(bbo_ctx* c, const uint8_t* bases, int len, int pairnum, int* id0);
/* countSetKmers (:1534-1593); *id = id at the exit hit or -1. */
int      bbo_count_set_kmers(bbo_ctx* c, const uint8_t* bases, int len, int pairnum, int maxBadKmers, int* id);

/* Batch = what processList does around the k-mer stage for every read/pair of a ListNum
 * (BBDukProcessorS.java:807-813, 948-1093, 1431-1443, 1464-1493).
 * bases: concatenated ASCII; offsets[n+1]; if paired, reads 2i and 2i+1 are mates (n even).
 * out_a[i]  = ktrim: bases trimmed x        | kfilter: found (countSetKmers return)
 * out_id[i] = ktrim: id0 (or -1)            | kfilter: id at exit hit (or -1)
 * out_flags[i] = BBO_FLAG_* bits.   Counters accumulate in ctx.   nthreads>=1 (pairs are sharded). */
/* ktrim=n: like bbo_process_batch; out_a = number of masked bases per read, out_mask = bit i set <=> base i of the
 * concatenated `bases` buffer is masked (caller zeroes the (offsets[n]+31)/32 words). */
int      bbo_process_batch_mask(bbo_ctx* c, const uint8_t* bases, const int64_t* offsets, int64_t n, int paired,
                                int32_t* out_a, int32_t* out_id, uint8_t* out_flags, uint32_t* out_mask, int nthreads);
/* everything at once: out_mask (ktrim=n) and out_left (ktrim=rl: what the left pass removed; out_a = right + left) may be NULL */
int      bbo_process_batch_ex(bbo_ctx* c, const uint8_t* bases, const int64_t* offsets, int64_t n, int paired,
                              int32_t* out_a, int32_t* out_id, uint8_t* out_flags, uint32_t* out_mask, int32_t* out_left, int nthreads);
int      bbo_process_batch(bbo_ctx* c, const uint8_t* bases, const int64_t* offsets, int64_t n, int paired,
                           int32_t* out_a, int32_t* out_id, uint8_t* out_flags, int nthreads);
/* findBestMatch with the lists rename() prints (:1702, 2508-2522): out_nids[i] = idList.size when read i matched, else 0;
 * out_match_ids / out_match_counts [i*max_ids + j], j < min(out_nids[i], max_ids) = idList / countList in first-hit order. */
int      bbo_process_batch_matches(bbo_ctx* c, const uint8_t* bases, const int64_t* offsets, int64_t n, int paired,
                                   int32_t* out_a, int32_t* out_id, uint8_t* out_flags, int max_ids, int32_t* out_nids,
                                   int32_t* out_match_ids, int32_t* out_match_counts, int nthreads);
/* ksplit (unpaired): out_a = bases removed, out_leftmost/out_rightmost = the span ksplit() computed (-1,-1: nothing found);
 * BBO_FLAG_REMOVED <=> the read was split in two (both pieces go to outm). */
int      bbo_process_batch_split(bbo_ctx* c, const uint8_t* bases, const int64_t* offsets, int64_t n,
                                 int32_t* out_a, int32_t* out_id, uint8_t* out_flags, int32_t* out_leftmost, int32_t* out_rightmost, int nthreads);

/* counters: out[0..15] as BBO_* then scaffoldReadCounts[0..nscaf) then scaffoldBaseCounts[0..nscaf). */
int      bbo_counters_len(const bbo_ctx* c);
void     bbo_get_counters(const bbo_ctx* c, int64_t* out);
void     bbo_reset_counters(bbo_ctx* c);

#ifdef __cplusplus
}
#endif
#endif
