/*
 * bbduk_gpu.h -- C ABI of the MI355X-native BBDuk k-mer matching path (libbbduk_hip.so).
 *
 * This is the drop-in boundary SURVEY.md §8(b) describes: the reference (BBTools v40.02, Java) has no
 * plugin API for this path, so the boundary is introduced at *batch* granularity and follows the
 * reference's own JNI convention (jni/BBMergeOverlapper.c:505-519, jni/jgi_BBMergeOverlapper.h:22-39):
 * plain pointers and sizes, caller-owned buffers, an int status (0 = OK, negative = error; never throws,
 * never aborts), no callbacks, no pointer retained past return.  The JNI shim a maintainer would add
 * (Java_bbduk_BBDukGpu_*) is shown in INTEGRATION.md.
 *
 * What each entry point replaces in the reference (paths relative to /root/reference/current/):
 *
 *   bbduk_create            the derived constants of bbduk/BBDukParser.java:146-165,230-312 (handed over
 *                           already derived, exactly as the Java fields hold them) and the per-thread
 *                           processor state of bbduk/BBDukProcessorS.java:272-277.
 *   bbduk_upload_table_way  a verbatim image of one kmer.HashArray1D way: array() (kmer/HashArray.java:672),
 *                           values() (kmer/HashArray1D.java:407), victims().toList() (kmer/HashForest.java).
 *   bbduk_upload_pairs      the same key->id map as a flat list (the map.LongIntMapX default of BBDukS:
 *                           keys()/values(), map/LongIntMapX.java:589-596).
 *   bbduk_finalize_table    BBDukIndex.setKmersLoaded: the map becomes read-only and HBM-resident.
 *   bbduk_ktrim_batch       the ktrim(r1)/ktrim(r2) calls + the trimming branch of processList
 *                           (bbduk/BBDukProcessorS.java:948-1033, 1806-1811, 1993-2140) for a whole batch.
 *   bbduk_kfilter_batch     the countSetKmers calls + the filtering branch (:1035-1093, 1534-1593).
 *   bbduk_get_counters      BBDukProcessorS.add (:300-342): readsIn ... basesOutm + scaffold counters.
 *   *_device variants       the same operators on buffers that already live in HBM (no PCIe in the call);
 *                           this is what a device-side ingest stage and bench.py call.
 *
 * Batch layout: `bases` = the reads' ASCII bases concatenated (what Read.bases holds after
 * Read.validate), `offsets[n+1]` = int64 start of each read (offsets[n] = total bytes).  If `paired`,
 * reads 2i and 2i+1 are mates r1,r2 (n must be even); pairnum = index&1.
 */
#ifndef BBDUK_GPU_H
#define BBDUK_GPU_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BBDUK_ABI_VERSION 2

/* status codes */
#define BBDUK_OK                 0
#define BBDUK_ERR_ARG           -1   /* bad argument / unsupported parameter combination */
#define BBDUK_ERR_STATE         -2   /* call out of order (e.g. batch before finalize)    */
#define BBDUK_ERR_NOMEM         -3
#define BBDUK_ERR_DEVICE        -4   /* a HIP call failed; see bbduk_last_error           */
#define BBDUK_ERR_READ_TOO_LONG -5   /* a read exceeds BBDUK_MAX_READ_LEN                  */
#define BBDUK_ERR_ID_OVERFLOW   -6   /* findBestMatch: a read hit more than 64 distinct scaffolds */
#define BBDUK_ERR_UNSUPPORTED   -8   /* (rounds 3-4: trimfailuresto1bp on units beyond 2512 bases outside ktrim=r|l / kfilter; every operator serves it at
                                         any length since round 5 -- the code stays reserved) */
#define BBDUK_ERR_FORMAT        -7   /* FASTQ ingest: a record without '@' / '+' or with unequal base and quality lines */

#define BBDUK_MAX_READ_LEN   16384   /* reads up to this length go through the LDS-tiled kernels; longer ones (any length that
                                        fits an int) through the chunked long-read kernels of every operator */

/* operator selected at create time */
#define BBDUK_MODE_KFILTER   0       /* countSetKmers + filtering branch */
#define BBDUK_MODE_KTRIM_R   1       /* ktrim=r */
#define BBDUK_MODE_KTRIM_L   2       /* ktrim=l */
#define BBDUK_MODE_KMASK     3       /* ktrim=n / kmask=: mask matched bases instead of trimming (kmaskfullycovered=f) */
#define BBDUK_MODE_KTRIM_TIPS 4      /* ktrim=rl / ktrimtips=: a right pass, then a left pass on what is left */
#define BBDUK_MODE_KSPLIT    5       /* ksplit=t: cut the matched span out of an unpaired read (bbduk_ksplit_batch) */

/* per-read output flags */
#define BBDUK_FLAG_DISCARDED 1       /* setDiscarded(r)  (BBDukProcessorS.java:1464-1470)    */
#define BBDUK_FLAG_REMOVED   2       /* shouldRemove(r1,r2) -> remove (:1489-1492)           */

/* counter vector: [0..15] fixed slots, then scaffoldReadCounts[0..numScaffolds), then
 * scaffoldBaseCounts[0..numScaffolds)  (BBDukProcessorS.java:300-342). */
#define BBDUK_NCOUNTERS 16
enum { BBDUK_READS_IN = 0, BBDUK_BASES_IN, BBDUK_READS_KTRIMMED, BBDUK_BASES_KTRIMMED,
       BBDUK_READS_KFILTERED, BBDUK_BASES_KFILTERED, BBDUK_READS_OUTU, BBDUK_BASES_OUTU,
       BBDUK_READS_OUTM, BBDUK_BASES_OUTM, BBDUK_CTR_STATUS = 15 /* nonzero = device-side error code */ };

/* The a1 scalars of SURVEY §8a, as BBDukParser holds them AFTER its own derivations. */
typedef struct bbduk_params {
    int32_t abi_version;            /* BBDUK_ABI_VERSION */
    int32_t mode;                   /* BBDUK_MODE_* */
    int32_t k;                      /* 1..31 (BBDukParser.java:162-165) */
    int32_t mink;                   /* useShortKmers iff 0<mink<k (:289) */
    int32_t rcomp;                  /* :159 */
    int32_t forbidNs;               /* forbidNs || hdist<1 (:150) */
    int32_t minlen;                 /* k-1 (:274) */
    int32_t minlen2;                /* (maskMiddle?(k-midMaskLen)/2:k) (:276) */
    int64_t middleMask;             /* :303-312 (-1 when off) */
    int32_t qhdist, qhdist2;        /* qHammingDistance, qHammingDistance2 (query-side expansion) */
    int32_t maxBadKmers;            /* maxBadKmers0 (:1232) */
    int32_t minReadLength;          /* :437 */
    float   minLenFraction;         /* :439 */
    int32_t removePairsIfEitherBad; /* :109 */
    int32_t trimPad;                /* tp= */
    int32_t ktrimExclusive;
    int32_t restrictLeft, restrictRight;
    int32_t skipR1, skipR2;
    int32_t numScaffolds;           /* scaffoldNames.size(): ids are 1..numScaffolds-1 */
    int32_t device;                 /* HIP device ordinal */
    int32_t trimPairsEvenly;        /* tpe (BBDukProcessorS.java:1021-1031; ktrim=r pairs) */
    int32_t qSkip;                  /* qskip= (0 or 1 = off; BBDukIndexMod.java:494) */
    int32_t speed;                  /* speed= 0..16 (query-side gate, BBDukIndexMod.java:506,562) */
    float   minKmerFraction;        /* mkf= (kfilter; BBDukProcessorS.java:1055-1062) */
    float   minCoveredFraction;     /* mcf= (kfilter; :1038-1049, countCoveredBases :1602-1651; out_found = covered bases) */
    int32_t kbig;                   /* the command line's k when it exceeds 31 (then k=31 and maskMiddle is off, BBDukParser.java:164,
                                       237-243): kfilter counts runs of consecutive 31-mer hits (countSetKmersBig, :1726-1804); <= k = off */
    int32_t findBestMatch;          /* findbestmatch/fbm (kfilter; BBDukProcessorS.java:1659-1719): out_id = the scaffold with the most
                                       hits, out_found = hits counted; needs maxBadKmers == 0 and minKmerFraction == 0 */
    int32_t kmaskFullyCovered;      /* kmaskfullycovered / mfc (ktrim=n; BBDukProcessorS.java:2163, 2193-2195, 2243-2245, 2286-2288): only
                                       bases all of whose covering k-mers match stay masked */
    int32_t trimFailuresTo1bp;      /* trimfailures / trimfailuresto1bp (BBDukParser.java:105-109, 774; BBDukProcessorS.java:1431, 1464-1488): a read that
                                       would be discarded is cut to its FIRST base instead, "discarded" then means "one base long", pairs are
                                       removed only when both mates are, and nothing is evicted: BBDUK_FLAG_DISCARDED marks the reads to cut
                                       (after the k-trim the operator reports), BBDUK_FLAG_REMOVED is never set, readsOutm stays 0.  Every operator
                                       serves it at any read length */
    int32_t reserved0;              /* 0 */
} bbduk_params;

typedef struct bbduk_handle bbduk_handle;

int  bbduk_abi_version(void);
int  bbduk_create(const bbduk_params* p, bbduk_handle** out);
int  bbduk_destroy(bbduk_handle* h);
const char* bbduk_last_error(const bbduk_handle* h);      /* valid until the next call on h */

/* ---- table: key -> scaffold id (1..numScaffolds-1; anything else is BBDUK_ERR_ARG).  The upload calls may be repeated before
 * finalize; a key uploaded more than once keeps its SMALLEST id -- the first scaffold in file order, which is what
 * HashArray.setIfNotPresent leaves in the reference's own tables.  bbduk_finalize_table places the pairs on the device. */
int  bbduk_upload_table_way(bbduk_handle* h, int32_t way, int32_t prime,
                            const int64_t* keys, const int32_t* values, int64_t ncells,   /* keys[i]==-1: empty */
                            const int64_t* vkeys, const int32_t* vvals, int64_t nvictims);
int  bbduk_upload_pairs(bbduk_handle* h, const int64_t* keys, const int32_t* values, int64_t n);
int  bbduk_finalize_table(bbduk_handle* h);
/* Alternative to upload + finalize: build the map ON the device from the reference sequences themselves (SURVEY 8f-4;
 * BBDukLoader.addToMap bbduk/BBDukLoader.java:416-494, BBDukIndexMod.addToMap/mutate :289-445): refs = the scaffolds'
 * bases concatenated in file order (host pointer), ref_offsets[n_refs+1]; scaffold s gets id s+1 and the first
 * scaffold wins a shared key.  hdist/hdist2 (0..3) are the ref-side Hamming distances (BBDukParser.java:130-133);
 * k, mink, rcomp, middleMask come from bbduk_create.  Leaves the handle finalized. */
int  bbduk_build_table_device(bbduk_handle* h, const uint8_t* refs, const int64_t* ref_offsets, int32_t n_refs,
                              int32_t hdist, int32_t hdist2);
/* The same with reference-side EDIT distance (edist / edist2 <= 1: substitutions, deletions, insertions -- BBDukIndexMod.java:383-445 with
 * editDistance > 0; hdist / hdist2 as BBDukParser.java:146 leaves them, i.e. max(edist, hdist)).  Round 4: until then only the host builder. */
int  bbduk_build_table_device_edits(bbduk_handle* h, const uint8_t* refs, const int64_t* ref_offsets, int32_t n_refs,
                                    int32_t hdist, int32_t hdist2, int32_t edist, int32_t edist2);
/* The same build, streamed, for references that are produced or ingested on the device or do not fit one host buffer (the 10 GB
 * reference of BASELINE configs[3]): begin announces an upper bound on the number of keys (for hdist 0: the number of reference
 * bases) -- beyond 2^25 keys (streamed builds; bbduk_finalize_table and bbduk_build_table_device: beyond 2^20 keys for plain kfilter (maxbadkmers too), ktrim=r and ktrim=l configurations with k >= 16, 2^21 with hdist > 0, and they keep a cache-resident twin beside it up to 2^25 keys, which serves batches with units beyond 2 512 bases) the map takes the HBM-resident layout: 12-14 bytes per slot, ~0.3 keys per slot up to 2^31 keys (32-bit candidate values), ~0.6 beyond (wide values; BASELINE configs[3]: 10^10 keys in 239 GB), both scanned by bbduk_bigs_kernel / bbduk_bigs_every_kernel; built in place;
 * every add hands over WHOLE scaffolds already in HBM (d_refs device pointer, ref_offsets HOST array of n_refs+1 values starting at
 * 0; scaffold i of the call gets id first_id + i); end leaves the handle finalized.  An error ends the build and frees the map.
 * More keys than announced: bbduk_build_end returns BBDUK_ERR_NOMEM ("more keys than announced") -- every sink stops taking keys once it has
 * overflowed, so the calls return within milliseconds (round 6; a full scratch set used to be probed for ever). */
int  bbduk_build_begin(bbduk_handle* h, int64_t max_keys, int32_t hdist, int32_t hdist2);
int  bbduk_build_add_device(bbduk_handle* h, const uint8_t* d_refs, const int64_t* ref_offsets, int32_t n_refs, int32_t first_id);
int  bbduk_build_end(bbduk_handle* h);
int64_t bbduk_table_size(const bbduk_handle* h);          /* distinct keys resident, or <0.  (A large hdist=1 kfilter map built on the device
                                                             is stored as its PARENT windows -- bbduk_seed.inc, DESIGN 4.12: same key -> id
                                                             answers; with its cache-resident twin -- up to 2^25 keys -- this call answers
                                                             in the reference's keys, without one it counts records, four per window.) */
int64_t bbduk_table_bytes(const bbduk_handle* h);         /* HBM bytes held by the table image */
/* point lookups through the device table (test hook): out_ids[i] = id or -1 */
int  bbduk_table_lookup(bbduk_handle* h, const int64_t* keys, int64_t n, int32_t* out_ids);

/* ---- host-buffer operators (PCIe inside the call; staging is the library's) */
int  bbduk_ktrim_batch(bbduk_handle* h, const uint8_t* bases, const int64_t* offsets, int64_t n, int32_t paired,
                       int32_t* out_trimmed /* x: bases removed by ktrim */, int32_t* out_id0 /* or -1 */,
                       uint8_t* out_flags);
int  bbduk_kfilter_batch(bbduk_handle* h, const uint8_t* bases, const int64_t* offsets, int64_t n, int32_t paired,
                         int32_t* out_found /* countSetKmers return */, int32_t* out_id /* or -1 */,
                         uint8_t* out_flags);

/* ---- device-buffer operators: every pointer is HBM; d_bases must be 16-byte aligned; `stream` is a
 * hipStream_t (NULL = default stream).  Asynchronous: returns after enqueue.  One handle may be driven from several
 * host threads on several streams at once (at most 64 launches in flight per handle).  d_counters is an int64
 * vector of bbduk_counters_len(h) that the kernel accumulates into (caller zeroes it).                 */
int  bbduk_ktrim_batch_device(bbduk_handle* h, const uint8_t* d_bases, const int64_t* d_offsets, int64_t n,
                              int64_t total_bases, int32_t paired, int32_t* d_out_trimmed, int32_t* d_out_id0,
                              uint8_t* d_out_flags, int64_t* d_counters, void* stream);
int  bbduk_kfilter_batch_device(bbduk_handle* h, const uint8_t* d_bases, const int64_t* d_offsets, int64_t n,
                                int64_t total_bases, int32_t paired, int32_t* d_out_found, int32_t* d_out_id,
                                uint8_t* d_out_flags, int64_t* d_counters, void* stream);

/* findBestMatch with the per-read match lists that `rename=t` prints (BBDukProcessorS.findBestMatch :1659-1719 fills
 * idList / countList, rename() :2508-2522 appends "\tname=count" for each entry to the read's id).  The handle must have
 * been created with findBestMatch=1 (BBDukParser.java:153: rename implies it).  Besides bbduk_kfilter_batch's outputs:
 *   out_nids[i]                          idList.size of read i when it matched (found > maxBadKmers), else 0
 *   out_match_ids[i*max_ids + j], out_match_counts[...]   j < min(out_nids[i], max_ids): the j-th distinct scaffold id in
 *                                        first-hit order and its number of hits; entries past the list are not written
 * max_ids in 1..64 (a read that hits more than 64 scaffolds is BBDUK_ERR_ID_OVERFLOW as for plain findBestMatch). */
int  bbduk_kfilter_batch_matches(bbduk_handle* h, const uint8_t* bases, const int64_t* offsets, int64_t n, int32_t paired,
                                 int32_t* out_found, int32_t* out_id, uint8_t* out_flags, int32_t max_ids, int32_t* out_nids,
                                 int32_t* out_match_ids, int32_t* out_match_counts);
int  bbduk_kfilter_batch_matches_device(bbduk_handle* h, const uint8_t* d_bases, const int64_t* d_offsets, int64_t n,
                                        int64_t total_bases, int32_t paired, int32_t* d_out_found, int32_t* d_out_id,
                                        uint8_t* d_out_flags, int32_t max_ids, int32_t* d_out_nids, int32_t* d_out_match_ids,
                                        int32_t* d_out_match_counts, int64_t* d_counters, void* stream);

/* ---- packed boundary format.  The reference keeps reads as one ASCII byte per base (`Read.bases`, stream/Read.java) and
 * converts each base to its 2-bit code inside the rolling loops (AminoAcid.baseToNumber / baseToComplementNumber,
 * bbduk/BBDukProcessorS.java:1752-1757); these entry points take the same reads with that conversion already done by
 * the caller, which cuts the bytes over PCIe 2.7x and removes the character work from the kernel:
 *   codes : 2 bits per base, 16 bases per uint32, base b of the concatenated buffer in bits 2*(b%16).. of word b/16
 *           (A=0 C=1 G=2 T/U=3, dna/AminoAcid.java:1284-1298); undefined bases may hold any code; (total+15)/16 words
 *   undef : 1 bit per base, bit b%32 of word b/32 set <=> baseToNumber[base] < 0 (N, IUPAC, junk); (total+31)/32 words;
 *           bits past `total` in the last word are ignored
 *   offsets, outputs, flags, counters: exactly as for the ASCII operators (offsets count BASES).
 * bbduk_pack_bases_* produce the format from ASCII (host: plain C loop; device: one kernel).  Device buffers must be
 * 16-byte aligned. */
int  bbduk_pack_bases_host(const uint8_t* bases, int64_t total_bases, uint32_t* codes, uint32_t* undef);
int  bbduk_pack_bases_device(const uint8_t* d_bases, int64_t total_bases, uint32_t* d_codes, uint32_t* d_undef,
                             int32_t device, void* stream);
int  bbduk_ktrim_batch_packed(bbduk_handle* h, const uint32_t* codes, const uint32_t* undef, const int64_t* offsets,
                              int64_t n, int32_t paired, int32_t* out_trimmed, int32_t* out_id0, uint8_t* out_flags);
int  bbduk_kfilter_batch_packed(bbduk_handle* h, const uint32_t* codes, const uint32_t* undef, const int64_t* offsets,
                                int64_t n, int32_t paired, int32_t* out_found, int32_t* out_id, uint8_t* out_flags);
int  bbduk_ktrim_batch_packed_device(bbduk_handle* h, const uint32_t* d_codes, const uint32_t* d_undef,
                                     const int64_t* d_offsets, int64_t n, int64_t total_bases, int32_t paired,
                                     int32_t* d_out_trimmed, int32_t* d_out_id0, uint8_t* d_out_flags,
                                     int64_t* d_counters, void* stream);
int  bbduk_kfilter_batch_packed_device(bbduk_handle* h, const uint32_t* d_codes, const uint32_t* d_undef,
                                       const int64_t* d_offsets, int64_t n, int64_t total_bases, int32_t paired,
                                       int32_t* d_out_found, int32_t* d_out_id, uint8_t* d_out_flags,
                                       int64_t* d_counters, void* stream);

/* packed-input variants of the ktrim=n and ktrim=rl operators (same outputs as bbduk_kmask_batch_device / bbduk_ktrimtips_batch_device;
 * the output mask still has one bit per base of the batch) */
int  bbduk_kmask_batch_packed_device(bbduk_handle* h, const uint32_t* d_codes, const uint32_t* d_undef, const int64_t* d_offsets, int64_t n,
                                     int64_t total_bases, int32_t paired, int32_t* d_out_masked, int32_t* d_out_id0,
                                     uint8_t* d_out_flags, uint32_t* d_out_mask, int64_t* d_counters, void* stream);
int  bbduk_ktrimtips_batch_packed_device(bbduk_handle* h, const uint32_t* d_codes, const uint32_t* d_undef, const int64_t* d_offsets, int64_t n,
                                         int64_t total_bases, int32_t paired, int32_t* d_out_right, int32_t* d_out_left,
                                         int32_t* d_out_id0, uint8_t* d_out_flags, int64_t* d_counters, void* stream);

/* ---- memory helpers for callers without HIP bindings (the JNI shim, bbduk_cli): thin wrappers over hipMalloc / hipFree,
 * hipHostMalloc / hipHostFree (pinned staging) and hipMemcpyAsync + hipStreamSynchronize on the given stream (NULL = default). */
int  bbduk_device_malloc(int32_t device, int64_t bytes, void** out);
int  bbduk_device_free(int32_t device, void* p);
int  bbduk_pinned_malloc(int64_t bytes, void** out);
int  bbduk_pinned_free(void* p);
int  bbduk_copy_to_device(int32_t device, void* d_dst, const void* src, int64_t bytes, void* stream);     /* returns after the copy */
int  bbduk_copy_from_device(int32_t device, void* dst, const void* d_src, int64_t bytes, void* stream);
int  bbduk_device_memset(int32_t device, void* d_dst, int32_t value, int64_t bytes, void* stream);
/* asynchronous forms: a stream that does not synchronise with the default stream (hipStreamNonBlocking); bbduk_copy_async returns once the copy is
 * queued (kind 0 host to device, 1 device to host, 2 device to device; host memory from bbduk_pinned_malloc), bbduk_stream_synchronize waits. */
int  bbduk_stream_create(int32_t device, void** out);
int  bbduk_stream_destroy(int32_t device, void* stream);
int  bbduk_stream_synchronize(int32_t device, void* stream);
int  bbduk_copy_async(int32_t device, void* dst, const void* src, int64_t bytes, int32_t kind, void* stream);

/* ---- device-side FASTQ ingest: raw FASTQ text in HBM -> line offsets, base offsets and the packed
 * boundary format, without the host looking at a base.  Restates the record splitting of stream/FASTQ.java:778-853
 * (toReadList: every four lines that fileIO/ByteFile.nextLine returns -- a line ends at '\n', one preceding '\r' is
 * dropped -- are one read: '@' header, bases, '+' line, qualities; the '@' and '+' are asserted at :1047-1049).
 *   d_text1 (, d_text2)  the text (any alignment); with two texts read 2i comes from record i of text 1 and read 2i+1 from
 *                        record i of text 2 (the two-file pairing of the reference's input stream); one interleaved text
 *                        gives the same layout by itself
 *   is_final             nonzero: an unterminated last line counts as a line (end of file); zero: the text is a chunk, the
 *                        bytes behind out->consumed{1,2} (an incomplete record) are the caller's to resubmit
 *   d_lines1/2           int64[4*(max_reads/texts)+1]: byte offset of every line of the records taken; line 4r is the
 *                        header of record r, 4r+1 its bases, 4r+3 its qualities, lines[4*records] == consumed
 *   d_offsets            int64[max_reads+1] base offsets; d_codes / d_undef as for bbduk_*_batch_packed_device
 *                        (capacity max_bases bases; bases <= text bytes / 2 always holds)
 * Synchronous on `stream`.  BBDUK_ERR_FORMAT: out->first_bad_read names the first malformed read. */
typedef struct bbduk_fastq_result {
    int64_t n_reads, total_bases, consumed1, consumed2, first_bad_read;
} bbduk_fastq_result;
int  bbduk_fastq_ingest_device(const uint8_t* d_text1, int64_t nbytes1, const uint8_t* d_text2, int64_t nbytes2, int32_t is_final,
                               int64_t max_reads, int64_t max_bases, int64_t* d_lines1, int64_t* d_lines2,
                               int64_t* d_offsets, uint32_t* d_codes, uint32_t* d_undef,
                               int32_t device, void* stream, bbduk_fastq_result* out);

/* The other end: the reads of an ingested batch back to FASTQ text (stream/FASTQ.java:474-490 toFASTQ: '@' id, bases, a
 * bare '+', qualities), trimmed by d_left[i] / d_right[i] bases (TrimRead.trimByAmount with the operators' amounts; NULL =
 * 0), in input order.  want_removed == 0 writes the reads without BBDUK_FLAG_REMOVED (out=), != 0 those with it (outm=).
 * *out_bytes = text length; BBDUK_ERR_ARG if it exceeds cap_out (the input text length always suffices).  Synchronous. */
int  bbduk_fastq_write_device(const uint8_t* d_text1, const int64_t* d_lines1, const uint8_t* d_text2, const int64_t* d_lines2,
                              int64_t n, const int32_t* d_left, const int32_t* d_right, const uint8_t* d_flags, int32_t want_removed,
                              uint8_t* d_out, int64_t cap_out, int32_t device, void* stream, int64_t* out_bytes);

/* The same writer for ktrim=n: d_mask = the base mask bbduk_kmask_batch*_device returned for this batch, d_base_offsets = the batch's
 * base offsets; masked bases are written as `symbol` (qualities '!' when it is 'N') or, symbol < 0, in lower case
 * (BBDukProcessorS.java:2309-2320). */
int  bbduk_fastq_write_masked_device(const uint8_t* d_text1, const int64_t* d_lines1, const uint8_t* d_text2, const int64_t* d_lines2,
                                     int64_t n, const int32_t* d_left, const int32_t* d_right, const uint8_t* d_flags, int32_t want_removed,
                                     const int64_t* d_base_offsets, const uint32_t* d_mask, int32_t symbol,
                                     uint8_t* d_out, int64_t cap_out, int32_t device, void* stream, int64_t* out_bytes);

/* ---- ktrim=n (bbduk/BBDukProcessorS.java:2149-2323; bbduk_params.kmaskFullyCovered selects the fully-covered variant).  out_masked[i] = kmask(Read)'s return
 * (BitSet.cardinality()), out_mask = one bit per base of the concatenated `bases` buffer (bit b of word b/32 set <=> the
 * caller replaces base b by trimSymbol / lower-cases it, :2309-2320); (offsets[n]+31)/32 words, the device variant needs
 * two more words of slack and clears the buffer itself.  Pair flags as for ktrim (reads keep their length). */
int  bbduk_kmask_batch(bbduk_handle* h, const uint8_t* bases, const int64_t* offsets, int64_t n, int32_t paired,
                       int32_t* out_masked, int32_t* out_id0, uint8_t* out_flags, uint32_t* out_mask);
int  bbduk_kmask_batch_device(bbduk_handle* h, const uint8_t* d_bases, const int64_t* d_offsets, int64_t n,
                              int64_t total_bases, int32_t paired, int32_t* d_out_masked, int32_t* d_out_id0,
                              uint8_t* d_out_flags, uint32_t* d_out_mask, int64_t* d_counters, void* stream);

/* ---- ktrim=rl / ktrimtips (bbduk/BBDukProcessorS.java:1813-1985).  out_right[i] / out_left[i] = bases the right and the
 * left pass removed (ktrimTips(r) returns their sum; trimpairsevenly adds to the right amount), out_id0[i] = the scaffold
 * credited by the right pass, else by the left pass, else -1.  The caller applies TrimRead.trimByAmount(r, 0, right, 1)
 * and then (r, left, 0, 1). */
int  bbduk_ktrimtips_batch(bbduk_handle* h, const uint8_t* bases, const int64_t* offsets, int64_t n, int32_t paired,
                           int32_t* out_right, int32_t* out_left, int32_t* out_id0, uint8_t* out_flags);
int  bbduk_ktrimtips_batch_device(bbduk_handle* h, const uint8_t* d_bases, const int64_t* d_offsets, int64_t n,
                                  int64_t total_bases, int32_t paired, int32_t* d_out_right, int32_t* d_out_left,
                                  int32_t* d_out_id0, uint8_t* d_out_flags, int64_t* d_counters, void* stream);

/* ---- ksplit=t (bbduk/BBDukProcessorS.java:2332-2506; unpaired reads, trimPad <= 0).  out_leftmost/out_rightmost = the
 * span ksplit() computes (-1,-1: nothing matched), out_trimmed = oldLen - r.pairLength() after the cut (:1005).  The caller
 * applies :2485-2498: leftmost==0 -> trimToPosition(r, rightmost+1, len-1, 1); rightmost==len-1 -> trimToPosition(r, 0,
 * leftmost-1, 1); else r2=r.subRead(rightmost+1, len-1), trimToPosition(r, 0, leftmost-1, 1), and the two pieces leave as a
 * pair through outm (BBDUK_FLAG_REMOVED, :1011). */
int  bbduk_ksplit_batch(bbduk_handle* h, const uint8_t* bases, const int64_t* offsets, int64_t n,
                        int32_t* out_trimmed, int32_t* out_leftmost, int32_t* out_rightmost, int32_t* out_id0, uint8_t* out_flags);
int  bbduk_ksplit_batch_device(bbduk_handle* h, const uint8_t* d_bases, const int64_t* d_offsets, int64_t n, int64_t total_bases,
                               int32_t* d_out_trimmed, int32_t* d_out_leftmost, int32_t* d_out_rightmost, int32_t* d_out_id0,
                               uint8_t* d_out_flags, int64_t* d_counters, void* stream);

/* Average duration (HIP events on the launch stream) of the dominant kernel over the last `last_k` batch launches
 * of this handle (at most 64 are remembered).  Synchronises on those launches.  For roofline reporting. */
int  bbduk_kernel_time_ms(bbduk_handle* h, int32_t last_k, float* avg_ms);

/* ---- counters accumulated by the host-buffer operators */
int  bbduk_counters_len(const bbduk_handle* h);           /* 16 + 2*numScaffolds */
int  bbduk_get_counters(bbduk_handle* h, int64_t* out, int32_t n);
int  bbduk_reset_counters(bbduk_handle* h);

/* ---- multi-GPU (SURVEY 8b "allreduce_counters", 8e).  Reads shard across GPUs in contiguous blocks of whole pairs, the
 * k-mer map is replicated per GPU, and the only exchange is ONE sum of the counter vector at the end of a run: the
 * device-side form of BBDukProcessorS.add merging the per-thread processors (bbduk/BBDukProcessorS.java:300-342).  It is
 * one RCCL ncclAllReduce(ncclInt64, ncclSum) over xGMI.  librccl is dlopen'ed at the first of these calls.
 *   one process per GPU : rank 0 calls bbduk_comm_unique_id and its launcher hands the 128 bytes to every rank (MPI,
 *                         torch.distributed's store, a file ...); every rank calls bbduk_comm_create on its handle; then
 *                         bbduk_allreduce_counters(h) (the handle's own counters, blocking) or _device (any int64 vector of
 *                         bbduk_counters_len(h) on the handle's device; asynchronous on `stream`).  Every rank must call.
 *   one process, N GPUs : (the JVM host, bbduk_cli devices=) bbduk_comm_create_local(handles, n) once, then
 *                         bbduk_allreduce_counters_local(handles, n) from ONE thread: handles that share a device are summed
 *                         on it first, the device leaders run the all-reduce, every handle ends with the global sums.
 * The status slot (BBDUK_CTR_STATUS) is summed like the rest: nonzero anywhere stays nonzero. */
#define BBDUK_COMM_ID_BYTES 128
/* Optional, first thing in a host that will form a group over SEVERAL devices: starts the load of librccl (hundreds of megabytes whose code objects are
 * all registered at load: tens of seconds from a cold page cache) on a thread of its own, so that it overlaps the table build.  A group whose handles
 * share one device never loads it. */
int  bbduk_comm_preload(void);
int  bbduk_comm_unique_id(uint8_t* id128);
int  bbduk_comm_create(bbduk_handle* h, int32_t nranks, int32_t rank, const uint8_t* id128);
int  bbduk_comm_create_local(bbduk_handle** handles, int32_t n);
int  bbduk_comm_destroy(bbduk_handle* h);                 /* also done by bbduk_destroy; a local group ends for all its members */
int  bbduk_comm_size(const bbduk_handle* h);              /* ranks (distinct devices) of the handle's communicator, 0 = none */
int  bbduk_allreduce_counters(bbduk_handle* h);
int  bbduk_allreduce_counters_device(bbduk_handle* h, int64_t* d_counters, void* stream);
int  bbduk_allreduce_counters_local(bbduk_handle** handles, int32_t n);

/* ---- deterministic synthetic read generator, device side; the bit-identical host side is
 * bbduk_synth_generate_host.  Reads are fixed length; pair p = reads 2p, 2p+1.                        */
typedef struct bbduk_synth_params {
    uint64_t seed;
    int32_t  read_len;             /* 150 */
    int32_t  ins_min, ins_max;     /* insert size ~ U[ins_min, ins_max] */
    int32_t  adapter1_len, adapter2_len;
    const uint8_t* adapter1;       /* read-through sequence seen by r1 after the insert (host pointer) */
    const uint8_t* adapter2;       /* ... by r2 */
    uint32_t sub_rate_q32;         /* substitution probability inside adapters/contaminant, * 2^32 */
    uint32_t n_rate_q32;           /* probability of an N at any base, * 2^32 */
    uint32_t contam_frac_q32;      /* fraction of pairs drawn from the contaminant sequence, * 2^32 */
    int64_t  contam_len;           /* length of contaminant sequence (0 = none) */
    const uint8_t* contam;         /* contaminant sequence (host pointer) */
} bbduk_synth_params;

int  bbduk_synth_generate_device(const bbduk_synth_params* sp, int64_t first_pair, int64_t n_pairs,
                                 uint8_t* d_bases, int64_t* d_offsets /* 2*n_pairs+1 */, int32_t device, void* stream);
int  bbduk_synth_generate_host(const bbduk_synth_params* sp, int64_t first_pair, int64_t n_pairs,
                               uint8_t* bases, int64_t* offsets);
/* the truth behind the generator: out_insert[p] = insert size of pair first_pair + p (a read keeps min(read_len, insert) genome bases,
 * adapter read-through follows) -- what AddAdapters writes into read names as <initial>_<remaining> (jgi/AddAdapters.java:485) */
int  bbduk_synth_pair_inserts(const bbduk_synth_params* sp, int64_t first_pair, int64_t n_pairs, int32_t* out_insert);

#ifdef __cplusplus
}
#endif
#endif
