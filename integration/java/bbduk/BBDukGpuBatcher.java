package bbduk;

import java.nio.ByteBuffer;
import java.nio.IntBuffer;
import java.nio.LongBuffer;
import java.util.ArrayList;

import shared.TrimRead;
import stream.Read;
import structures.ListNum;

/**
 * The batch aggregator between pass A and pass B of BBDukProcessorS.processList (SURVEY.md section 8b, INTEGRATION.md section 2).
 *
 * processList is one loop per pair: stages that may mutate or discard a read (:778-946, "pass A"), the k-mer stage (:948-1093) and the stages
 * that depend on the trimmed read (:1096-1456, "pass B").  A device call per pair is pointless, so a ProcessThread that owns a batcher runs
 * pass A over a whole ListNum, hands the list over with {@link #offer}, and goes on to the next list; when enough reads have gathered (or the
 * input ends) {@link #run} makes ONE BBDukGpu call over every pair still alive, and {@link #drain} gives the lists back in the order they came
 * (ListNum.id ascending within a thread, which is all process() :740-766 needs: ros.add(list, ln.id) restores the global order) together with a
 * per-read view of the device's answers for pass B to use in place of ktrim() / ktrimTips() / kmask() / countSetKmers() / findBestMatch().
 *
 * Nothing here changes a decision: the device applies minlen1/minlen2, shouldRemove, trimPairsEvenly and the counter formulas of :948-1093 itself
 * (bit 0 of the flags = setDiscarded(r), bit 1 = the pair is removed); the batcher only carries bytes there and answers back.  What the device
 * does not see -- qualities, names, the Read objects -- never leaves the JVM: the trim is applied here with the reference's own
 * TrimRead.trimByAmount, so bases and qualities are cut together exactly as TrimRead.trimToPosition would have cut them.
 *
 * One batcher per ProcessThread, one BBDukGpu handle per GPU shared by all of them (the handle's two staging slots let two threads' calls
 * overlap).  NOT compiled in this repository (no JDK in the build image); tests/test_jni_shims.py checks every BBDukGpu call made here against
 * that class's declarations.
 */
public final class BBDukGpuBatcher {

	/** One list on its way through the device: the ListNum as pass A left it and where its pairs sit in the batch. */
	public static final class Slice {
		public final ListNum<Read> ln;
		/** slot[p] = batch index of pair p's r1 (its mate, when there is one, sits at slot[p]+1), or -1: the pair did not go to the
		 *  device (removed by pass A, or nothing to scan) and pass B treats it as the k-mer stage's "no hit". */
		final int[] slot;
		Slice(ListNum<Read> ln, int[] slot){this.ln=ln; this.slot=slot;}
	}

	private final long handle;
	private final int mode;
	private final boolean paired;
	private final boolean matches;   // findBestMatch with rename: idList / countList per read (BBDukProcessorS.java:1702, 2508-2522)
	private final int maxIds;
	private final int capReads;
	private final long capBases;

	private final ByteBuffer bases, offsetsB, outA, outB, outId, outFlags, outMask, outLeftmost, outRightmost, outNids, outMatchIds, outMatchCounts;
	private final LongBuffer offsets;
	private final IntBuffer a, b, id, leftmost, rightmost, nids, matchIds, matchCounts;
	private final ArrayList<Slice> pending=new ArrayList<Slice>();
	private int n=0;          // reads in the batch
	private long nb=0;        // bases in the batch
	private boolean ran=false;

	/**
	 * @param handle   BBDukGpu.create(parser, ...) with its table uploaded and finalized
	 * @param mode     BBDukGpu.MODE_* (the one create() derived from the parser)
	 * @param paired   cris.paired(): reads 2i and 2i+1 of a batch are mates
	 * @param capReads reads per device call (1e5 and up keeps the launch and the PCIe copies efficient; a 2x150 batch of 1e6 reads is 150 MB)
	 * @param capBases bases per device call
	 * @param maxIds   0, or with findBestMatch + rename the number of scaffold ids kept per read (1..64)
	 */
	public BBDukGpuBatcher(long handle, int mode, boolean paired, int capReads, long capBases, int maxIds){
		this.handle=handle; this.mode=mode; this.paired=paired; this.capReads=capReads; this.capBases=capBases;
		this.matches=(maxIds>0 && mode==BBDukGpu.MODE_KFILTER); this.maxIds=maxIds;
		bases=BBDukGpu.allocPinned(capBases+64);
		offsetsB=BBDukGpu.allocPinned(8L*(capReads+1)); offsets=offsetsB.asLongBuffer();
		outA=BBDukGpu.allocPinned(4L*capReads); a=outA.asIntBuffer();
		outId=BBDukGpu.allocPinned(4L*capReads); id=outId.asIntBuffer();
		outFlags=BBDukGpu.allocPinned(capReads);
		final boolean tips=(mode==BBDukGpu.MODE_KTRIM_TIPS), split=(mode==BBDukGpu.MODE_KSPLIT), mask=(mode==BBDukGpu.MODE_KMASK);
		outB=(tips ? BBDukGpu.allocPinned(4L*capReads) : null); b=(tips ? outB.asIntBuffer() : null);
		outMask=(mask ? BBDukGpu.allocPinned(((capBases+31)/32+1)*4) : null);
		outLeftmost=(split ? BBDukGpu.allocPinned(4L*capReads) : null); leftmost=(split ? outLeftmost.asIntBuffer() : null);
		outRightmost=(split ? BBDukGpu.allocPinned(4L*capReads) : null); rightmost=(split ? outRightmost.asIntBuffer() : null);
		outNids=(matches ? BBDukGpu.allocPinned(4L*capReads) : null); nids=(matches ? outNids.asIntBuffer() : null);
		outMatchIds=(matches ? BBDukGpu.allocPinned(4L*capReads*maxIds) : null); matchIds=(matches ? outMatchIds.asIntBuffer() : null);
		outMatchCounts=(matches ? BBDukGpu.allocPinned(4L*capReads*maxIds) : null); matchCounts=(matches ? outMatchCounts.asIntBuffer() : null);
		reset();
	}

	private void reset(){
		pending.clear(); n=0; nb=0; ran=false;
		bases.clear(); offsets.clear(); offsets.put(0, 0L);
	}

	/** Would this list still fit?  (A list is never split: its pairs stay together so that drain() hands whole lists back.) */
	public boolean fits(ListNum<Read> ln){
		long add=0; int reads=0;
		for(Read r1 : ln.list){
			if(r1==null){continue;}
			add+=r1.length(); reads++;
			if(r1.mate!=null){add+=r1.mate.length(); reads++;}
			else if(paired){reads++;}
		}
		return n+reads<=capReads && nb+add<=capBases;
	}

	/**
	 * Takes a list that has been through pass A.  alive[p] = pair p reaches the k-mer stage (processList's `remove` is still false for it).
	 * Mates go in adjacent; a pair that lost its mate in a paired run gets an empty second read so that the device's pairing by index holds.
	 * @return true when the batch is full enough to run (the caller then calls run() and drain())
	 */
	public boolean offer(ListNum<Read> ln, boolean[] alive){
		assert(!ran) : "drain() the finished batch first";
		final ArrayList<Read> reads=ln.list;
		final int[] slot=new int[reads.size()];
		for(int p=0; p<reads.size(); p++){
			final Read r1=reads.get(p);
			if(r1==null || !alive[p]){slot[p]=-1; continue;}
			final Read r2=r1.mate;
			slot[p]=n;
			put(r1.bases);
			if(paired){put(r2==null ? null : r2.bases);}
		}
		pending.add(new Slice(ln, slot));
		return n>=capReads*3/4 || nb>=capBases*3/4;
	}

	private void put(byte[] x){
		if(x!=null){bases.put(x); nb+=x.length;}
		n++;
		offsets.put(n, nb);
	}

	public boolean isEmpty(){return pending.isEmpty();}

	/** The device call over everything offered since the last drain().  Throws what BBDukGpu throws (the caller sets errorState, BBDukS.java:199-202). */
	public void run(){
		if(n>0){
			switch(mode){
			case BBDukGpu.MODE_KTRIM_R: case BBDukGpu.MODE_KTRIM_L:
				BBDukGpu.ktrimBatch(handle, bases, offsetsB, n, paired, outA, outId, outFlags); break;
			case BBDukGpu.MODE_KTRIM_TIPS:
				BBDukGpu.ktrimTipsBatch(handle, bases, offsetsB, n, paired, outA, outB, outId, outFlags); break;
			case BBDukGpu.MODE_KMASK:
				BBDukGpu.kmaskBatch(handle, bases, offsetsB, n, paired, outA, outId, outFlags, outMask); break;
			case BBDukGpu.MODE_KSPLIT:
				BBDukGpu.ksplitBatch(handle, bases, offsetsB, n, outA, outLeftmost, outRightmost, outId, outFlags); break;
			default:
				if(matches){BBDukGpu.kfilterBatchMatches(handle, bases, offsetsB, n, paired, outA, outId, outFlags, maxIds, outNids, outMatchIds, outMatchCounts);}
				else{BBDukGpu.kfilterBatch(handle, bases, offsetsB, n, paired, outA, outId, outFlags);}
			}
		}
		ran=true;
	}

	/** The lists of the finished batch in the order they were offered; the answers stay readable until the next offer(). */
	public ArrayList<Slice> drain(){
		assert(ran) : "run() first";
		final ArrayList<Slice> out=new ArrayList<Slice>(pending);
		pending.clear(); n=0; nb=0; ran=false;
		bases.clear(); offsets.put(0, 0L);
		return out;
	}

	/*--------------------------------------------------------------*/
	/*----------------   Answers, per read of a pair  ----------------*/
	/*--------------------------------------------------------------*/

	/** Batch index of mate `pairnum` (0 or 1) of pair p of a slice, or -1. */
	public int index(Slice s, int p, int pairnum){
		final int i=s.slot[p];
		return i<0 ? -1 : i+pairnum;
	}
	/** ktrim=r|l: bases trimmed (what ktrim() returns, :2108-2139); ktrim=rl: the right amount; ktrim=n: bases masked; kfilter: found (countSetKmers' return value,
	 *  or countCoveredBases' under mcf); ksplit: bases trimmed off the pair. */
	public int amount(int i){return a.get(i);}
	/** ktrim=rl: the left amount (ktrimTips trims both ends, :1813-1826). */
	public int leftAmount(int i){return b.get(i);}
	/** id0: the scaffold of the first hit in scan order (what the scaffold counters were bumped for), 0 = none. */
	public int id0(int i){return id.get(i);}
	public boolean discarded(int i){return (outFlags.get(i)&BBDukGpu.FLAG_DISCARDED)!=0;}
	/** processList's `remove` for the pair this read belongs to (shouldRemove, :1481-1484, or ksplit's r1.mate!=null). */
	public boolean removed(int i){return (outFlags.get(i)&BBDukGpu.FLAG_REMOVED)!=0;}
	/** ksplit: the span it found (leftmost, rightmost), -1 = none; the caller builds the two sub-reads (:2466-2504). */
	public int leftmost(int i){return leftmost.get(i);}
	public int rightmost(int i){return rightmost.get(i);}
	/** ktrim=n: is base j of read i masked?  (bit offsets[i]+j of the batch's mask) */
	public boolean masked(int i, int j){
		final long bit=offsets.get(i)+j;
		return ((outMask.getInt((int)(bit>>>5)*4)>>>(int)(bit&31))&1)!=0;
	}
	/** findBestMatch + rename: how many scaffolds read i matched (at most maxIds are listed), and the j-th of them in first-hit order. */
	public int numMatches(int i){return nids.get(i);}
	public int matchId(int i, int j){return matchIds.get(i*maxIds+j);}
	public int matchCount(int i, int j){return matchCounts.get(i*maxIds+j);}

	/**
	 * Pass B's first step for the trimming modes: cut the Read the way ktrim() / ktrimTips() / kmask() would have left it, with the reference's own
	 * TrimRead (bases and qualities together).  The decision (discard, remove, trimPairsEvenly's extra cut) is already inside the amounts and flags.
	 * @param trimFailuresTo1bp the parser's flag (BBDukParser.java:774): a failed read is cut to one base instead of being flagged
	 * @return what the replaced call would have returned (bases trimmed or masked)
	 */
	public int apply(Read r, int i, boolean kmaskLowercase, byte trimSymbol, boolean trimFailuresTo1bp){
		final int x=amount(i);
		switch(mode){
		case BBDukGpu.MODE_KTRIM_R: if(x>0){TrimRead.trimByAmount(r, 0, x, 1);} break;
		case BBDukGpu.MODE_KTRIM_L: if(x>0){TrimRead.trimByAmount(r, x, 0, 1);} break;
		case BBDukGpu.MODE_KTRIM_TIPS: {
			final int left=leftAmount(i);
			if(x>0 || left>0){TrimRead.trimByAmount(r, left, x, 1);}
			setDiscarded(r, i, trimFailuresTo1bp);
			return x+left;
		}
		case BBDukGpu.MODE_KMASK:
			if(x>0){   // :2309-2320
				final byte[] bs=r.bases, q=r.quality;
				for(int j=0; j<bs.length; j++){
					if(masked(i, j)){
						if(kmaskLowercase){bs[j]=(byte)Character.toLowerCase(bs[j]);}
						else{bs[j]=trimSymbol; if(q!=null && trimSymbol=='N'){q[j]=0;}}
					}
				}
			}
			break;
		default: break;   // kfilter and ksplit change no bases here
		}
		setDiscarded(r, i, trimFailuresTo1bp);
		return x;
	}

	/** BBDukProcessorS.setDiscarded (:1464-1470) for a read the device marked: with trimfailuresto1bp the mark means "cut to the first base
	 *  after the k-trim" (include/bbduk_gpu.h, bbduk_params.trimFailuresTo1bp) and nothing is flagged. */
	private void setDiscarded(Read r, int i, boolean trimFailuresTo1bp){
		if(!discarded(i)){return;}
		if(trimFailuresTo1bp){
			if(r.length()>1){TrimRead.trimByAmount(r, 0, r.length()-1, 1, false);}
		}else{
			r.setDiscarded(true);
		}
	}

	/** Frees the pinned buffers; the handle belongs to the caller. */
	public void close(){
		for(ByteBuffer x : new ByteBuffer[]{bases, offsetsB, outA, outB, outId, outFlags, outMask, outLeftmost, outRightmost, outNids, outMatchIds, outMatchCounts}){
			if(x!=null){BBDukGpu.freePinned(x);}
		}
	}
}
