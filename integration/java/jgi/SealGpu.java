package jgi;

import java.nio.ByteBuffer;

import shared.Shared;

/**
 * Batch k-mer stage of Seal on an MI355X, through libbbduk_jni.so -> libbbduk_hip.so (C ABI: include/seal_gpu.h).
 * Same pattern as bbduk.BBDukGpu: static natives, int status, direct buffers for the batch (nothing pinned while the GPU works).
 * NOT compiled in this repository (no JDK in the build image); see INTEGRATION.md section 6.
 *
 * Usage from Seal (jgi/Seal.java):
 *   long h = SealGpu.create(k, maskMiddle, midMaskLen, rcomp, forbidNs, hammingDistance, refSkip, restrictLeft, restrictRight, qSkip, speed,
 *                           matchMode, ambigMode, keepPairsTogether, minKmerHits, minKmerFraction, clearzone, clearzoneFraction, minReadLength, maxReadLength,
 *                           minLenFraction, !removePairsIfEitherBad, scaffoldNames.size(), device);
 *   // table: either let the library load the references (what LoadThread.addToMap does, Seal.java:1760-1945) ...
 *   for each scaffold in file order: SealGpu.addRefSequence(h, bases);                 // ids 1, 2, ... as scaffoldNames assigns them
 *   // ... or hand over what the JVM's loader built: every (key, id) of keySets[w] (HashArrayHybridFast: values[cell] > 0 = one id,
 *   // < -1 = -(index into setList)), any order
 *   SealGpu.uploadPairs(h, keys, ids);
 *   SealGpu.finalizeTable(h);
 *   ... per >= 1e5 reads in ProcessThread.run (:2011): concatenate r.bases (mates adjacent) into a direct buffer + long offsets ...
 *   SealGpu.batch(h, bases, offsets, n, paired, firstNumericID, maxIds, outSites, outAssigned, outMax, outIds, outFlags);
 *   // outFlags[i]&2: the pair goes to rosm (mlist), else to rosu (:2282-2290); outIds[i*maxIds ..]: als.add(r1, scaffoldNames.get(id));
 *   // the counters (readsMatchedT, scaffoldReadCountsT, ...) come back once, at the end: SealGpu.readCounters(h, long[])
 */
public final class SealGpu {

	static{
		if(Shared.USE_JNI){System.loadLibrary("bbduk_jni");}   // (not Shared.loadJNI: see bbduk.BBDukGpu)
	}

	private static native long createJNI(int[] ip, float[] fp);
	private static native int addRefSequenceJNI(long h, byte[] bases);
	private static native int uploadPairsJNI(long h, long[] keys, int[] ids);
	private static native int finalizeJNI(long h);
	private static native int batchJNI(long h, ByteBuffer bases, ByteBuffer offsets, int n, boolean paired, long firstNumericID, int maxIds,
			ByteBuffer outSites, ByteBuffer outAssigned, ByteBuffer outMax, ByteBuffer outIds, ByteBuffer outFlags);
	private static native int readCountersJNI(long h, long[] out);
	private static native long countersLenJNI(long h);
	private static native String lastErrorJNI(long h);
	private static native void destroyJNI(long h);

	/** Scalars as Seal holds them after its constructor (jgi/Seal.java:480-570). */
	public static long create(int k, boolean maskMiddle, int midMaskLen, boolean rcomp, boolean forbidNs, int hdist, int refSkip,
			int restrictLeft, int restrictRight, int qSkip, int speed, int matchMode, int ambigMode, boolean keepPairsTogether,
			int minKmerHits, float minKmerFraction, int clearzone, float clearzoneFraction, int minReadLength, int maxReadLength, float minLenFraction,
			boolean requireBothBad, int maxScaffolds, int device){
		// Seal.AMBIG_* / MATCH_* -> SEAL_AMBIG_* / SEAL_MATCH_* of seal_gpu.h
		final int am=(ambigMode==Seal.AMBIG_FIRST ? 0 : ambigMode==Seal.AMBIG_ALL ? 1 : ambigMode==Seal.AMBIG_RANDOM ? 2 : 3);
		final int mm=(matchMode==Seal.MATCH_ALL ? 0 : matchMode==Seal.MATCH_FIRST ? 1 : 2);
		final int[] ip={k, maskMiddle ? 1 : 0, midMaskLen, rcomp ? 1 : 0, forbidNs ? 1 : 0, hdist, refSkip, restrictLeft, restrictRight, qSkip, speed,
				mm, am, keepPairsTogether ? 1 : 0, minKmerHits, clearzone, minReadLength, maxReadLength, requireBothBad ? 1 : 0, maxScaffolds, device};
		final long h=createJNI(ip, new float[] {minKmerFraction, minLenFraction, clearzoneFraction});
		if(h<=0){throw new RuntimeException("seal_create failed: "+h);}
		return h;
	}
	public static void addRefSequence(long h, byte[] bases){check(h, addRefSequenceJNI(h, bases));}
	public static void uploadPairs(long h, long[] keys, int[] ids){check(h, uploadPairsJNI(h, keys, ids));}
	public static void finalizeTable(long h){check(h, finalizeJNI(h));}
	/** All buffers direct (bbduk.BBDukGpu.allocPinned), little-endian views; outputs are per read. */
	public static void batch(long h, ByteBuffer bases, ByteBuffer offsets, int n, boolean paired, long firstNumericID, int maxIds,
			ByteBuffer outSites, ByteBuffer outAssigned, ByteBuffer outMax, ByteBuffer outIds, ByteBuffer outFlags){
		check(h, batchJNI(h, bases, offsets, n, paired, firstNumericID, maxIds, outSites, outAssigned, outMax, outIds, outFlags));
	}
	public static long[] readCounters(long h){
		final long[] out=new long[(int)countersLenJNI(h)];
		check(h, readCountersJNI(h, out));
		return out;
	}
	public static void destroy(long h){destroyJNI(h);}
	private static void check(long h, int rc){if(rc!=0){throw new RuntimeException("seal: rc="+rc+" "+lastErrorJNI(h));}}
}
