package bbduk;

import java.nio.ByteBuffer;
import java.nio.ByteOrder;

import shared.Shared;

/**
 * Batch k-mer stage of BBDuk on an MI355X, through libbbduk_jni.so -> libbbduk_hip.so (C ABI: include/bbduk_gpu.h).
 * Follows the JNI pattern of jgi.BBMergeOverlapper (static natives, Shared.loadJNI, int status, int[]/byte[] outputs).
 * NOT compiled in this repository (no JDK in the build image); see INTEGRATION.md.
 *
 * Usage from BBDukProcessorS.processList (split into pass A / device call / pass B, SURVEY.md section 8b):
 *   long h = BBDukGpu.create(parser, index.scaffoldNames.size(), deviceOrdinal);
 *   for each way w: BBDukGpu.uploadWay(h, w, table.prime(), table.array(), table.values(), victimKeys, victimVals);
 *   BBDukGpu.finalizeTable(h);            // or BBDukGpu.buildTable(h, refBases, refOffsets, hdist, hdist2)
 *   ... per >=1e5 reads: concatenate r.bases into one byte[] + long[] offsets, mates adjacent ...
 *   BBDukGpu.ktrimBatch(h, bases, offsets, n, paired, outTrimmed, outId0, outFlags);   // direct buffers from BBDukGpu.allocPinned
 *   ... TrimRead.trimByAmount(r, 0, outTrimmed[i], 1) for ktrim=r; discard/remove from outFlags[i] ...
 */
public final class BBDukGpu {

	static{
		// NOT Shared.loadJNI(name): that method is guarded by ONE global flag (shared/Shared.java:159, 732), so once
		// jgi.BBMergeOverlapper has loaded libbbtoolsjni (BBDukProcessorS imports it for tbo) a second call is a no-op and
		// every native below would throw UnsatisfiedLinkError.  System.loadLibrary keeps its own per-library record; the
		// library is found on java.library.path (bbduk.sh passes -Djava.library.path=<bbtools>/jni/, javasetup.sh).
		if(Shared.USE_JNI){
			try{
				System.loadLibrary("bbduk_jni");
			}catch(UnsatisfiedLinkError e){
				// same fallback Shared.loadJNI uses (shared/Shared.java:745-768): <classpath>/../jni/libbbduk_jni.so
				String cp=BBDukGpu.class.getProtectionDomain().getCodeSource().getLocation().getPath();
				System.load(new java.io.File(new java.io.File(cp).getParentFile(), "jni/libbbduk_jni.so").getAbsolutePath());
			}
		}
	}

	/** ip = {mode,k,mink,rcomp,forbidNs,minlen,minlen2,qhdist,qhdist2,maxBadKmers,minReadLength,
	 *  removePairsIfEitherBad,trimPad,ktrimExclusive,restrictLeft,restrictRight,skipR1,skipR2,numScaffolds,device,
	 *  trimPairsEvenly,qSkip,speed,kbig,findBestMatch,kmaskFullyCovered,trimFailuresTo1bp};  fp = {minLenFraction,minKmerFraction,minCoveredFraction} */
	private static native long createJNI(int[] ip, long middleMask, float[] fp);
	private static native int uploadWayJNI(long h, int way, int prime, long[] keys, int[] values, long[] vkeys, int[] vvals);
	private static native int uploadPairsJNI(long h, long[] keys, int[] values);
	private static native int finalizeJNI(long h);
	private static native int buildTableJNI(long h, ByteBuffer refs, ByteBuffer refOffsets, int nRefs, int hdist, int hdist2);
	private static native ByteBuffer allocPinnedJNI(long bytes);
	private static native void freePinnedJNI(ByteBuffer b);
	// batch operators: every buffer is DIRECT (see allocPinned): nothing is pinned or copied by the shim, and no JNI critical region
	// is held while the call waits for the GPU (integration/jni/BBDukGpu.c)
	private static native int ktrimBatchJNI(long h, ByteBuffer bases, ByteBuffer offsets, int n, boolean paired,
			ByteBuffer outTrimmed, ByteBuffer outId0, ByteBuffer outFlags);
	private static native int kfilterBatchJNI(long h, ByteBuffer bases, ByteBuffer offsets, int n, boolean paired,
			ByteBuffer outFound, ByteBuffer outId, ByteBuffer outFlags);
	private static native int ktrimBatchPackedJNI(long h, ByteBuffer codes, ByteBuffer undef, ByteBuffer offsets, int n, boolean paired,
			ByteBuffer outTrimmed, ByteBuffer outId0, ByteBuffer outFlags);
	private static native int kfilterBatchPackedJNI(long h, ByteBuffer codes, ByteBuffer undef, ByteBuffer offsets, int n, boolean paired,
			ByteBuffer outFound, ByteBuffer outId, ByteBuffer outFlags);
	private static native int kmaskBatchJNI(long h, ByteBuffer bases, ByteBuffer offsets, int n, boolean paired,
			ByteBuffer outMasked, ByteBuffer outId0, ByteBuffer outFlags, ByteBuffer outMask);
	private static native int ktrimTipsBatchJNI(long h, ByteBuffer bases, ByteBuffer offsets, int n, boolean paired,
			ByteBuffer outRight, ByteBuffer outLeft, ByteBuffer outId0, ByteBuffer outFlags);
	private static native int ksplitBatchJNI(long h, ByteBuffer bases, ByteBuffer offsets, int n,
			ByteBuffer outTrimmed, ByteBuffer outLeftmost, ByteBuffer outRightmost, ByteBuffer outId0, ByteBuffer outFlags);
	private static native int kfilterBatchMatchesJNI(long h, ByteBuffer bases, ByteBuffer offsets, int n, boolean paired,
			ByteBuffer outFound, ByteBuffer outId, ByteBuffer outFlags, int maxIds, ByteBuffer outNids, ByteBuffer outMatchIds, ByteBuffer outMatchCounts);
	private static native int countersJNI(long h, long[] out);
	/** One JVM driving several GPUs: handles[] (one per device, the map replicated) end with the summed counters (one RCCL all-reduce). */
	private static native int allreduceCountersLocalJNI(long[] handles);
	private static native void destroyJNI(long h);

	public static final int MODE_KFILTER=0, MODE_KTRIM_R=1, MODE_KTRIM_L=2, MODE_KMASK=3, MODE_KTRIM_TIPS=4, MODE_KSPLIT=5;
	public static final int FLAG_DISCARDED=1, FLAG_REMOVED=2;

	/** Copies BBDukParser's fields as they stand AFTER its own derivations (BBDukParser.java:130-312). */
	public static long create(BBDukParser p, int numScaffolds, int device){
		final int mode=(p.ksplit ? MODE_KSPLIT : (p.ktrimRight && p.ktrimLeft) ? MODE_KTRIM_TIPS : p.ktrimRight ? MODE_KTRIM_R :
			p.ktrimLeft ? MODE_KTRIM_L : p.ktrimN ? MODE_KMASK : MODE_KFILTER);
		final int[] ip={mode, p.k, p.mink, p.rcomp ? 1 : 0, p.forbidNs ? 1 : 0, p.minlen, p.minlen2,
				p.qHammingDistance, p.qHammingDistance2, p.maxBadKmers0, p.minReadLength,
				p.removePairsIfEitherBad ? 1 : 0, p.trimPad, p.ktrimExclusive ? 1 : 0,
				p.restrictLeft, p.restrictRight, p.skipR1 ? 1 : 0, p.skipR2 ? 1 : 0, numScaffolds, device,
				p.trimPairsEvenly ? 1 : 0, p.qSkip, p.speed, (p.kbig>p.k ? p.kbig : 0), (p.findBestMatch && mode==MODE_KFILTER) ? 1 : 0,
				(p.kmaskFullyCovered && mode==MODE_KMASK) ? 1 : 0, p.trimFailuresTo1bp ? 1 : 0};
		final float[] fp={p.minLenFraction, p.minKmerFraction, p.minCoveredFraction};
		final long h=createJNI(ip, p.middleMask, fp);
		if(h<=0){throw new RuntimeException("bbduk_create failed: "+h);}
		return h;
	}
	public static void uploadWay(long h, int way, int prime, long[] keys, int[] values, long[] vkeys, int[] vvals){
		check(uploadWayJNI(h, way, prime, keys, values, vkeys, vvals), "bbduk_upload_table_way");
	}
	public static void uploadPairs(long h, long[] keys, int[] values){check(uploadPairsJNI(h, keys, values), "bbduk_upload_pairs");}
	public static void finalizeTable(long h){check(finalizeJNI(h), "bbduk_finalize_table");}
	/** Page-locked host memory as a direct, little-endian ByteBuffer: batch buffers live here (asIntBuffer()/asLongBuffer() views for the
	 *  offsets and the int outputs).  The Java side writes r.bases into it while it walks the ListNum (bases.put(r.bases)); that is the
	 *  one copy a batch needs anyway, because every Read owns its own byte[] (stream/Read.java:3281). */
	public static ByteBuffer allocPinned(long bytes){
		final ByteBuffer b=allocPinnedJNI(bytes);
		if(b==null){throw new OutOfMemoryError("bbduk_pinned_malloc("+bytes+")");}
		return b.order(ByteOrder.LITTLE_ENDIAN);
	}
	public static void freePinned(ByteBuffer b){freePinnedJNI(b);}
	/** The GPU builds the map itself from the scaffolds' bases (hdist <= 3, no edist) instead of receiving the finished tables. */
	public static void buildTable(long h, ByteBuffer refs, ByteBuffer refOffsets, int nRefs, int hdist, int hdist2){
		check(buildTableJNI(h, refs, refOffsets, nRefs, hdist, hdist2), "bbduk_build_table_device");
	}
	public static void ktrimBatch(long h, ByteBuffer bases, ByteBuffer offsets, int n, boolean paired, ByteBuffer outTrimmed, ByteBuffer outId0, ByteBuffer outFlags){
		check(ktrimBatchJNI(h, bases, offsets, n, paired, outTrimmed, outId0, outFlags), "bbduk_ktrim_batch");
	}
	/** Also serves k>31 (countSetKmersBig) and findBestMatch when the parser set kbig / findBestMatch. */
	public static void kfilterBatch(long h, ByteBuffer bases, ByteBuffer offsets, int n, boolean paired, ByteBuffer outFound, ByteBuffer outId, ByteBuffer outFlags){
		check(kfilterBatchJNI(h, bases, offsets, n, paired, outFound, outId, outFlags), "bbduk_kfilter_batch");
	}
	/** findBestMatch plus idList/countList per read for rename() (BBDukProcessorS.java:1702, 2508-2522): read i matched outNids[i] scaffolds,
	 *  outMatchIds/outMatchCounts[i*maxIds+j] are the j-th of them in first-hit order (maxIds in 1..64). */
	public static void kfilterBatchMatches(long h, ByteBuffer bases, ByteBuffer offsets, int n, boolean paired, ByteBuffer outFound, ByteBuffer outId, ByteBuffer outFlags,
			int maxIds, ByteBuffer outNids, ByteBuffer outMatchIds, ByteBuffer outMatchCounts){
		check(kfilterBatchMatchesJNI(h, bases, offsets, n, paired, outFound, outId, outFlags, maxIds, outNids, outMatchIds, outMatchCounts), "bbduk_kfilter_batch_matches");
	}
	/** codes: 2 bits per base, 16 per int; undef: 1 bit per base (baseToNumber<0); offsets still count bases. */
	public static void ktrimBatchPacked(long h, ByteBuffer codes, ByteBuffer undef, ByteBuffer offsets, int n, boolean paired, ByteBuffer outTrimmed, ByteBuffer outId0, ByteBuffer outFlags){
		check(ktrimBatchPackedJNI(h, codes, undef, offsets, n, paired, outTrimmed, outId0, outFlags), "bbduk_ktrim_batch_packed");
	}
	public static void kfilterBatchPacked(long h, ByteBuffer codes, ByteBuffer undef, ByteBuffer offsets, int n, boolean paired, ByteBuffer outFound, ByteBuffer outId, ByteBuffer outFlags){
		check(kfilterBatchPackedJNI(h, codes, undef, offsets, n, paired, outFound, outId, outFlags), "bbduk_kfilter_batch_packed");
	}
	/** outMask: one bit per base of the concatenated batch, ((offsets[n]+31)/32+1)*4 bytes; the caller runs :2309-2320 over it. */
	public static void kmaskBatch(long h, ByteBuffer bases, ByteBuffer offsets, int n, boolean paired, ByteBuffer outMasked, ByteBuffer outId0, ByteBuffer outFlags, ByteBuffer outMask){
		check(kmaskBatchJNI(h, bases, offsets, n, paired, outMasked, outId0, outFlags, outMask), "bbduk_kmask_batch");
	}
	public static void ktrimTipsBatch(long h, ByteBuffer bases, ByteBuffer offsets, int n, boolean paired, ByteBuffer outRight, ByteBuffer outLeft, ByteBuffer outId0, ByteBuffer outFlags){
		check(ktrimTipsBatchJNI(h, bases, offsets, n, paired, outRight, outLeft, outId0, outFlags), "bbduk_ktrimtips_batch");
	}
	public static void ksplitBatch(long h, ByteBuffer bases, ByteBuffer offsets, int n, ByteBuffer outTrimmed, ByteBuffer outLeftmost, ByteBuffer outRightmost, ByteBuffer outId0, ByteBuffer outFlags){
		check(ksplitBatchJNI(h, bases, offsets, n, outTrimmed, outLeftmost, outRightmost, outId0, outFlags), "bbduk_ksplit_batch");
	}
	public static void counters(long h, long[] out){check(countersJNI(h, out), "bbduk_get_counters");}
	/** The device-side BBDukProcessorS.add (:300-342) over the per-GPU handles: call once, after the last batch, from one thread. */
	public static void allreduceCounters(long[] handles){check(allreduceCountersLocalJNI(handles), "bbduk_allreduce_counters_local");}
	public static void destroy(long h){destroyJNI(h);}

	private static void check(int rc, String what){
		if(rc!=0){throw new RuntimeException(what+" returned "+rc);}   // caller ORs this into errorState (BBDukS.java:199-202)
	}
}
