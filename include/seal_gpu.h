/*
 * seal_gpu.h -- C ABI of the MI355X (gfx950) k-mer path of jgi/Seal.java, the next tool on the encode / lookup core of
 * bbduk_gpu.h ("beyond these four").  Same library (libbbduk_hip.so), same conventions: plain pointers and sizes,
 * 0 = OK, negative = BBDUK_ERR_* of bbduk_gpu.h, seal_last_error() for the text.
 *
 * What it replaces in the reference (a JNI shim would bind exactly these):
 *   seal_add_ref_sequence      Seal.LoadThread.addToMap(Read, skip) + mutate          jgi/Seal.java:1760-1945
 *   seal_upload_pairs          the (k-mer, scaffold) pairs of a table the JVM built   kmer/HashArrayHybridFast (Seal.java:108)
 *   seal_batch / _device       ProcessThread.run's length rule and k-mer branch       jgi/Seal.java:2034-2038, 2100-2130, 2180-2280
 *                              = findBestMatch :2864-2909, condenseLoose :2654, filterTopScaffolds_withClearzone :2697,
 *                                assignTogether :2386 / assignIndependently :2462
 *   seal_read_counters         the thread totals run() adds up                        jgi/Seal.java:1624-1660
 *
 * A k-mer maps to the SET of scaffolds that contain it (ascending ids, as the reference's loader leaves them).
 * Not served (seal_create / seal_params_from_args refuse them): qhdist>0, edist, processcontainedref,
 * countvector=t, rename, taxonomy / barcode / gene-set outputs, quality trimming and filtering other than the length rule.
 */
#ifndef SEAL_GPU_H
#define SEAL_GPU_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SEAL_MATCH_ALL    0          /* match=all  (default; Seal.java:124) */
#define SEAL_MATCH_FIRST  1          /* match=first: stop at the first matching k-mer */
#define SEAL_MATCH_UNIQUE 2          /* match=unique: stop at the first k-mer that belongs to one scaffold only */
#define SEAL_AMBIG_FIRST  0          /* ambig=first: the smallest id of the top scaffolds */
#define SEAL_AMBIG_ALL    1
#define SEAL_AMBIG_RANDOM 2          /* default: numericID % sites (deterministic, Seal.java:2401) */
#define SEAL_AMBIG_TOSS   3

#define SEAL_FLAG_REMOVED 1          /* the length rule removed the pair (minlength / maxlength, Seal.java:2100-2130) */
#define SEAL_FLAG_MATCHED 2          /* assigned >= 1: the pair goes to outm, else to outu (:2282-2290) */

/* counter vector: fixed slots, then four per-scaffold arrays of numScaffolds entries each:
 * scaffoldReadCounts, scaffoldBaseCounts, scaffoldFragCounts, scaffoldAmbigReadCounts (Seal.java:2425-2436) */
#define SEAL_NCOUNTERS 16
enum { SEAL_READS_IN = 0, SEAL_BASES_IN, SEAL_FRAGS_IN, SEAL_READS_MATCHED, SEAL_BASES_MATCHED, SEAL_READS_UNMATCHED,
       SEAL_BASES_UNMATCHED, SEAL_READS_QFILTERED, SEAL_BASES_QFILTERED, SEAL_READS_QTRIMMED, SEAL_CTR_STATUS = 15 };

typedef struct seal_params {
    int32_t k;                      /* 1..31 (default 31) */
    int32_t maskMiddle;             /* mm= (default on) */
    int32_t midMaskLen;             /* mm=<n>; 0 = 2-(k&1) (Seal.java:548-552) */
    int32_t rcomp;
    int32_t forbidNs;               /* forbidn= ; the operator applies  forbidNs || hdist<1  (:492) */
    int32_t hdist;                  /* reference-side Hamming distance of seal_add_ref_sequence (0..2) */
    int32_t refSkip;                /* rskip= */
    int32_t restrictLeft, restrictRight;
    int32_t qSkip;                  /* 0 or 1 = off */
    int32_t speed;                  /* 0..16 */
    int32_t matchMode;              /* SEAL_MATCH_* */
    int32_t ambigMode;              /* SEAL_AMBIG_* */
    int32_t keepPairsTogether;      /* kpt= (default on) */
    int32_t minKmerHits;            /* mkh= (>= 1) */
    float   minKmerFraction;        /* mkf= */
    int32_t clearzone;              /* cz= */
    float   clearzoneFraction;      /* czf=: the clear zone is at least ceil(czf * valid k-mers of the pair) (Seal.java:2213-2214) */
    int32_t minReadLength;          /* minlength= (default 10) */
    int32_t maxReadLength;          /* maxlength= */
    float   minLenFraction;         /* mlf= */
    int32_t requireBothBad;         /* rieb=f */
    int32_t maxScaffolds;           /* upper bound on the scaffold count incl. the fake scaffold 0 (sizes the counter vector) */
    int32_t device;
} seal_params;

typedef struct seal_handle seal_handle;

void seal_default_params(seal_params* p);                               /* Seal.java:104-131, 3088-3098 */
/* "k=25 hdist=1 ambig=toss mkh=2 ..." (blank-separated key=value, Seal's own names and synonyms for the parameters above); unknown or
 * unsupported keys -> BBDUK_ERR_ARG with the key in errbuf */
int  seal_params_from_args(const char* args, seal_params* p, char* errbuf, int errlen);
int  seal_create(const seal_params* p, seal_handle** out);
int  seal_destroy(seal_handle* h);
const char* seal_last_error(const seal_handle* h);

/* ---- table.  Scaffold ids start at 1 (scaffold 0 is the reference's fake first scaffold, Seal.java:134-139). */
int  seal_add_ref_sequence(seal_handle* h, const uint8_t* bases, int64_t len, int32_t* out_id);   /* ids in call order */
int  seal_upload_pairs(seal_handle* h, const int64_t* keys, const int32_t* ids, int64_t n);     /* any order, repeats allowed */
int  seal_finalize(seal_handle* h);                                                               /* sets of ids per key -> device */
int32_t seal_num_scaffolds(const seal_handle* h);          /* incl. scaffold 0 */
int64_t seal_table_keys(const seal_handle* h);             /* distinct k-mers */
int64_t seal_table_pairs(const seal_handle* h);            /* distinct (k-mer, scaffold) pairs */

/* ---- the operator.  Reads concatenated like bbduk_*_batch (offsets[n+1], paired: mates interleaved, n even).
 * first_numeric_id = Read.numericID of the batch's first pair (pairs, or reads when unpaired, are numbered consecutively from it).
 * Per READ i:  out_sites[i]    finalList.size (keepPairsTogether: the pair's, in both mates' slots)
 *              out_assigned[i] scaffolds assigned (ambig=all can assign several)
 *              out_max[i]      the highest hit count
 *              out_ids[i*max_ids .. ]  the first max_ids assigned scaffolds, in the reference's order
 *              out_flags[i]    SEAL_FLAG_*  (pair-level, in both mates' slots)
 * More than 64 distinct scaffolds hit by one pair is reported as BBDUK_ERR_ID_OVERFLOW (counter slot SEAL_CTR_STATUS), not answered wrongly. */
int  seal_batch_device(seal_handle* h, const uint8_t* d_bases, const int64_t* d_offsets, int64_t n, int64_t total_bases,
                       int32_t paired, int64_t first_numeric_id, int32_t max_ids,
                       int32_t* d_out_sites, int32_t* d_out_assigned, int32_t* d_out_max, int32_t* d_out_ids, uint8_t* d_out_flags,
                       int64_t* d_counters, void* stream);
int  seal_batch(seal_handle* h, const uint8_t* bases, const int64_t* offsets, int64_t n, int32_t paired, int64_t first_numeric_id,
                int32_t max_ids, int32_t* out_sites, int32_t* out_assigned, int32_t* out_max, int32_t* out_ids, uint8_t* out_flags);
int64_t seal_counters_len(const seal_handle* h);           /* SEAL_NCOUNTERS + 4 * maxScaffolds */
int  seal_read_counters(seal_handle* h, int64_t* out);     /* the handle's own vector (seal_batch adds to it) */
int  seal_reset_counters(seal_handle* h);
double seal_last_kernel_ms(seal_handle* h);                /* HIP events around the last seal kernel */

/* ---- multi-GPU: one process per GPU, the reads sharded over the ranks, the whole table on every rank; the only exchange is one RCCL
 * all-reduce (sum, int64) of the handles' counter vectors, as Seal adds up its ProcessThreads (jgi/Seal.java:1624-1660).  id128 from
 * bbduk_comm_unique_id (bbduk_gpu.h), broadcast by the caller's own launcher. */
int  seal_comm_create(seal_handle* h, int32_t nranks, int32_t rank, const uint8_t* id128);
int  seal_allreduce_counters(seal_handle* h);

#ifdef __cplusplus
}
#endif
#endif
