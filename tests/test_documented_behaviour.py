"""Known answers taken from the reference's own DOCUMENTATION (docs/guides/BBDukGuide.txt, docs/guides/SealGuide.txt): where the guides
state what BBDuk / Seal do with a concrete input -- a count, a trimmed length, which scaffold gets the read -- the expected values below
are worked out BY HAND from that sentence (never by running a restatement), then asked of the CPU oracle (CPU tests) and of the HIP operators
through the C ABI (gpu tests).  This is not a substitute for replaying real BBDuk output (no JVM in the image: DESIGN.md section 2, "parity
unpinned"), but it ties both restatements and the kernels to statements the reference's author published about this path.

Every case cites the guide line it encodes.  Output conventions (include/bbduk_gpu.h): ktrim -> a = bases removed; kfilter -> a = k-mer
hits counted until the scan stops; flags bit 0 = read discarded, bit 1 = unit removed (goes to outm)."""
import random

import numpy as np
import pytest

from oracle.oracle_ffi import Oracle, pack_reads
from tests import util


def rnd(seed, n):
    return util.rand_seq(random.Random(seed), n)


def flip(base):
    return {65: 67, 67: 71, 71: 84, 84: 65}[base]          # A->C->G->T->A: always a different base


ADAPTER = b"CTGTCTCTTATACACATCTCCGAGCCCACGAGACTAAGGCGAATCTCGTATGCCGTCTTCTGCTTG"      # one 65-base adapter; any sequence would do


def _middle_error_read():
    """BBDukGuide.txt:60 -- 'a 100bp read that should match the reference, and a single error right in the middle ... the read has
    100-31+1=70 kmers; of those, 31 have an error, so they won't match the reference.  Thus, only 39/70 or 56% of the kmers match' and
    '99% of the bases would be covered by kmers matching the reference'."""
    ref = rnd(11, 400)
    read = bytearray(ref[150:250]); read[50] = flip(read[50])
    return ref, bytes(read)


def _mink_case():
    """BBDukGuide.txt:39 -- K=25: 'if the last 12bp of a read are adapter sequence, it will not match a reference 25-mer, because it is too
    short.  "mink=8" will additionally look for shorter kmers with lengths 24 to 8 ... with "ktrim=r", they will only be taken from the
    right end ... a single 24-mer of the last 24 bases, a single 23-mer, etc. down to a single 8-mer, the last 8 bases.  The reference short
    kmers will also come only from the ends of reference sequences, but they will come from both ends'."""
    genomic = rnd(12, 88)
    return genomic + ADAPTER[:12], genomic + ADAPTER[:7], genomic + ADAPTER[20:32]


# (name, bbduk.sh args, oracle kwargs, refs, reads, paired, expected a per read, expected flags per read)
def cases():
    out = []
    ref, read = _middle_error_read()
    # :60 the count itself (mkh beyond reach so the scan never stops early), maskmiddle off as in the guide's arithmetic
    out.append(("guide60_39_of_70_kmers", "k=31 mm=f mkh=100", dict(k=31, maskMiddle=0, maxBadKmers0=99), [ref], [read], False, [39], [0]))
    # :60 'By default, a read is considered to match the reference if they share a single kmer'
    out.append(("guide60_default_one_kmer", "k=31 mm=f", dict(k=31, maskMiddle=0), [ref], [read, rnd(13, 100)], False, [1, 0], [3, 0]))
    # :60 '"mkh=2" would require 2 kmer hits': a read with exactly ONE reference k-mer (31 reference bases inside random sequence) passes, the scan
    # of the 39-hit read stops at its second hit
    one = rnd(14, 40) + ref[10:41] + rnd(15, 29)
    out.append(("guide60_mkh2", "k=31 mm=f mkh=2", dict(k=31, maskMiddle=0, maxBadKmers0=1), [ref], [one, read], False, [1, 2], [0, 3]))
    # :60 '"mkf=0.5" would require 50% of the kmers to match': 39/70 = 56 % matches; at mkf=0.6 the same read does not
    # (a = hits counted until the scan stops: (int)(69 * 0.5) + 1 = 35 at mkf=0.5, all 39 at mkf=0.6)
    out.append(("guide60_mkf_half", "k=31 mm=f mkf=0.5", dict(k=31, maskMiddle=0, minKmerFraction=0.5), [ref], [read], False, [35], [3]))
    out.append(("guide60_mkf_0p6", "k=31 mm=f mkf=0.6", dict(k=31, maskMiddle=0, minKmerFraction=0.6), [ref], [read], False, [39], [0]))
    # :60 '"mcf=0.5" would require 50% of the bases to be covered' / '99% of the bases would be covered': a = covered bases
    out.append(("guide60_mcf_99_bases", "k=31 mm=f mcf=0.995", dict(k=31, maskMiddle=0, minCoveredFraction=0.995), [ref], [read], False, [99], [0]))
    # (at mcf=0.5 the guide only says the read matches; countCoveredBases stops counting once it has the 50 bases it needs, BBDukProcessorS.java:1635:
    # the count is left unchecked)
    out.append(("guide60_mcf_half", "k=31 mm=f mcf=0.5", dict(k=31, maskMiddle=0, minCoveredFraction=0.5), [ref], [read], False, None, [3]))
    # :39 mink
    r12, r7, rmid = _mink_case()
    out.append(("guide39_k25_too_short", "ktrim=r k=25 mm=f", dict(k=25, maskMiddle=0, ktrimRight=1), [ADAPTER], [r12], False, [0], [0]))
    out.append(("guide39_mink8_trims_12", "ktrim=r k=25 mink=8 mm=f", dict(k=25, mink=8, maskMiddle=0, ktrimRight=1), [ADAPTER], [r12], False, [12], [0]))
    # seven adapter bases are shorter than mink: untouched; twelve bases from the MIDDLE of the adapter are not an end of the reference: untouched
    out.append(("guide39_mink8_not_7", "ktrim=r k=25 mink=8 mm=f", dict(k=25, mink=8, maskMiddle=0, ktrimRight=1), [ADAPTER], [r7, rmid], False, [0, 0], [0, 0]))
    # 'they will come from both ends' of the reference: a read ending in the reverse complement of the adapter's LAST 12 bases is trimmed too
    rright = rnd(16, 88) + util.revcomp(ADAPTER[-12:])
    out.append(("guide39_both_reference_ends", "ktrim=r k=25 mink=8 mm=f", dict(k=25, mink=8, maskMiddle=0, ktrimRight=1), [ADAPTER], [rright], False, [12], [0]))
    # :118 '"ktrim=r" is for right-trimming (3' adapters), and "ktrim=l" is for left-trimming (5' adapters)'; '"hdist" means "hamming distance";
    # this allows one mismatch'
    ins = rnd(17, 60)
    r3 = ins + ADAPTER[:40]
    r5 = ADAPTER[-40:] + ins
    out.append(("guide118_ktrim_r", "ktrim=r k=25 mm=f", dict(k=25, maskMiddle=0, ktrimRight=1), [ADAPTER], [r3], False, [40], [0]))
    out.append(("guide118_ktrim_l", "ktrim=l k=25 mm=f", dict(k=25, maskMiddle=0, ktrimLeft=1), [ADAPTER], [r5], False, [40], [0]))
    bad = bytearray(ins + ADAPTER[:25]); bad[60 + 12] = flip(bad[60 + 12])          # the only 25-mer of adapter in the read, one mismatch
    out.append(("guide118_hdist0_misses", "ktrim=r k=25 mm=f hdist=0", dict(k=25, maskMiddle=0, ktrimRight=1), [ADAPTER], [bytes(bad)], False, [0], [0]))
    out.append(("guide118_hdist1_one_mismatch", "ktrim=r k=25 mm=f hdist=1", dict(k=25, maskMiddle=0, hdist=1, ktrimRight=1), [ADAPTER], [bytes(bad)], False, [25], [0]))
    # :70 'If two reference sequences contain the same kmer, the kmer will only be associated with the first reference'
    shared = rnd(18, 60)
    refs2 = [rnd(19, 50) + shared, shared + rnd(20, 50)]
    out.append(("guide70_first_reference_wins", "k=31 mm=f", dict(k=31, maskMiddle=0), refs2, [shared[5:55]], False, [1], [3]))
    # ---- bbduk.sh's own parameter descriptions
    g = rnd(51, 200)
    mid = bytearray(g[80:111]); mid[15] = flip(mid[15])
    # bbduk.sh:120 'maskmiddle=t (mm) Treat the middle base of a kmer as a wildcard'
    out.append(("sh120_maskmiddle_default", "k=31", dict(k=31), [g], [bytes(mid)], False, [1], [3]))
    out.append(("sh120_maskmiddle_off", "k=31 mm=f", dict(k=31, maskMiddle=0), [g], [bytes(mid)], False, [0], [0]))
    # bbduk.sh:118 'rcomp=t Look for reverse-complements of kmers in addition to forward kmers'
    rc = util.revcomp(g[40:140])
    out.append(("sh118_rcomp_default", "k=31 mm=f", dict(k=31, maskMiddle=0), [g], [rc], False, [1], [3]))
    out.append(("sh118_rcomp_off", "k=31 mm=f rcomp=f", dict(k=31, maskMiddle=0, rcomp=0), [g], [rc], False, [0], [0]))
    # bbduk.sh:142 'forbidn=f (fn) Forbids matching of read kmers containing N.  By default, these will match a reference 'A' if hdist>0'
    # (the window is one whose FORWARD strand is the canonical one: the N reads as A in the forward k-mer but as 0 instead of T's code in the
    # reverse one, BBDukProcessorS.java:1548-1552, so on a window stored by its reverse strand the key would be neither)
    code = {65: 0, 67: 1, 71: 2, 84: 3}
    num = lambda sq: sum(code[c] << (2 * (len(sq) - 1 - i)) for i, c in enumerate(sq))
    w0 = [p for p in range(60, 120) if num(g[p:p + 31]) > num(util.revcomp(g[p:p + 31])) and 65 in g[p:p + 31].replace(g[p + 15:p + 16], b"")][0]
    w = bytearray(g[w0:w0 + 31]); ia = [i for i in range(31) if w[i] == 65 and i != 15][0]; w[ia] = ord("N")
    out.append(("sh142_N_matches_A_hdist1", "k=31 mm=f hdist=1", dict(k=31, maskMiddle=0, hdist=1), [g], [bytes(w)], False, [1], [3]))
    out.append(("sh142_forbidn", "k=31 mm=f hdist=1 fn=t", dict(k=31, maskMiddle=0, hdist=1, forbidN=1), [g], [bytes(w)], False, [0], [0]))
    out.append(("sh142_hdist0_no_N", "k=31 mm=f hdist=0", dict(k=31, maskMiddle=0), [g], [bytes(w)], False, [0], [0]))
    # bbduk.sh:222 'minlength=10 (ml) Reads shorter than this after trimming will be discarded'
    s9, s10 = rnd(55, 9) + ADAPTER[:40], rnd(56, 10) + ADAPTER[:40]
    out.append(("sh222_minlength", "ktrim=r k=25 mm=f", dict(k=25, maskMiddle=0, ktrimRight=1), [ADAPTER], [s9, s10], False, [40, 40], [3, 0]))
    # bbduk.sh:145 'removeifeitherbad=t (rieb) Paired reads get sent to 'outmatch' if either is match (or either is trimmed shorter than minlen).
    # Set to false to require both.'
    mate = rnd(57, 60)
    out.append(("sh145_rieb_default", "ktrim=r k=25 mm=f", dict(k=25, maskMiddle=0, ktrimRight=1), [ADAPTER], [s9, mate], True, [40, 0], [3, 2]))
    out.append(("sh145_rieb_off", "ktrim=r k=25 mm=f rieb=f", dict(k=25, maskMiddle=0, ktrimRight=1, requireBothBad=1), [ADAPTER], [s9, mate], True, [40, 0], [1, 0]))
    # bbduk.sh:242 'tp=0 (trimpad) Trim this much extra around matching kmers'
    out.append(("sh242_trimpad", "ktrim=r k=25 mm=f tp=2", dict(k=25, maskMiddle=0, ktrimRight=1, trimPad=2), [ADAPTER], [r3], False, [42], [0]))
    # bbduk.sh:258 'restrictleft=0 If positive, only look for kmer matches in the leftmost X bases' (:260 restrictright: the rightmost)
    lone = rnd(53, 60) + g[10:41] + rnd(54, 9)                     # the read's only reference k-mer covers bases 60..90
    out.append(("sh258_restrictleft_50", "k=31 mm=f restrictleft=50", dict(k=31, maskMiddle=0, restrictLeft=50), [g], [lone], False, [0], [0]))
    out.append(("sh258_restrictleft_95", "k=31 mm=f restrictleft=95", dict(k=31, maskMiddle=0, restrictLeft=95), [g], [lone], False, [1], [3]))
    out.append(("sh260_restrictright_50", "k=31 mm=f restrictright=50", dict(k=31, maskMiddle=0, restrictRight=50), [g], [lone], False, [1], [3]))
    out.append(("sh260_restrictright_30", "k=31 mm=f restrictright=30", dict(k=31, maskMiddle=0, restrictRight=30), [g], [lone], False, [0], [0]))
    # bbduk.sh:205 'mink=0 Look for shorter kmers at read tips down to this length ... Enabling this will disable maskmiddle': the adapter 25-mer with
    # its middle base wrong is found by the default mm=t, and no longer once mink is set
    out.append(("sh205_mm_finds_middle_error", "ktrim=r k=25", dict(k=25, ktrimRight=1), [ADAPTER], [bytes(bad)], False, [25], [0]))
    out.append(("sh205_mink_disables_mm", "ktrim=r k=25 mink=8", dict(k=25, mink=8, ktrimRight=1), [ADAPTER], [bytes(bad)], False, [0], [0]))
    return out


CASES = cases()
IDS = {"guide70_first_reference_wins": [1]}                        # expected id0 where the guide states it


def run_oracle(okw, refs, reads, paired):
    o = Oracle(**okw)
    for r in refs:
        o.add_ref(r)
    b, off = pack_reads(reads)
    return o, o.process_batch(b, off, paired)


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_oracle_gives_the_documented_answer(case):
    name, args, okw, refs, reads, paired, want_a, want_fl = case
    o, (a, ids, fl) = run_oracle(okw, refs, reads, paired)
    assert (want_a is None or a.tolist() == want_a) and fl.tolist() == want_fl, (name, a.tolist(), fl.tolist())
    if name in IDS:
        assert ids.tolist() == IDS[name]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_second_restatement_gives_the_documented_answer(case):
    """oracle/spec.py, written independently of the C oracle, on the same documented cases."""
    from oracle.spec import Spec, Args
    name, args, okw, refs, reads, paired, want_a, want_fl = case
    bools = {f for f, t in Args.__annotations__.items() if t is bool}
    s = Spec(Args(**{k: (bool(v) if k in bools else v) for k, v in okw.items()}))
    for r in refs:
        s.add_ref(r)
    got = s.process_batch(reads, paired)
    assert (want_a is None or [g[0] for g in got] == want_a) and [g[2] for g in got] == want_fl, (name, got)
    if name in IDS:
        assert [g[1] for g in got] == IDS[name]


def test_oracle_stored_kmers_as_documented():
    """BBDukGuide.txt:34 -- 'If a hamming distance is used, such as hdist=1, then the number of kmers stored will be multiplied by
    1+(3*k)^hdist': one reference 31-mer -> 94 keys at hdist=1 (the guide's own E. coli figures: 427998710 / 4554207 = 93.98);
    :55 -- 'rskip=4 ... will only store every 4th reference kmer'."""
    kmer = rnd(21, 31)
    o0 = Oracle(k=31, maskMiddle=0); o0.add_ref(kmer)
    o1 = Oracle(k=31, maskMiddle=0, hdist=1); o1.add_ref(kmer)
    assert (o0.stored_kmers, o1.stored_kmers) == (1, 94)
    ref = rnd(22, 430)                                             # 400 windows
    o4 = Oracle(k=31, maskMiddle=0, minSkip=4, maxSkip=4); o4.add_ref(ref)
    assert o4.stored_kmers == 100


# ---------------------------------------------------------------------------------------------- Seal (docs/guides/SealGuide.txt)
def _seal_guide_read(k=15):
    """A read that shares 10 k-mers with scaffold A, 8 with B, 3 with C and none with D (SealGuide.txt:32): three reference pieces of
    k+9, k+7 and k+2 bases, separated by a base that matches none of them so that the windows across a junction belong to no scaffold."""
    A, Bs, Cs, Ds = rnd(31, 200), rnd(32, 200), rnd(33, 200), rnd(34, 200)
    pa, pb, pc = A[40:40 + k + 9], Bs[60:60 + k + 7], Cs[80:80 + k + 2]
    return [A, Bs, Cs, Ds], pa + pb + pc


def seal_oracle():
    from tests.test_seal import SealOracle
    return SealOracle


@pytest.mark.parametrize("cz,want", [(0, [1]), (2, [1, 2]), (7, [1, 2, 3]), (9999, [1, 2, 3])])
def test_seal_oracle_clearzone_example(cz, want):
    """SealGuide.txt:32 -- 'say a read R shares 10 kmers with ref A, 8 kmers with B, 3 kmers with C, and 0 kmers with D.  At clearzone=0, this
    read unambiguously matches A.  At clearzone=2, it ambiguously matches A and B.  At clearzone=7, it ambiguously matches A, B, and C.  At
    clearzone=9999, it still only matches A, B, and C, not D'."""
    refs, read = _seal_guide_read()
    o = seal_oracle()(refs, k=15, maskMiddle=0, ambigMode=1, clearzone=cz)         # ambig=all: every scaffold inside the clear zone
    got, info = o.process(read, None, 0)
    assert got == want and info[2] == 10, (cz, got, info)         # info = [sites, assigned, best count, ...]


@pytest.mark.parametrize("ambig,check", [(2, lambda g: len(g) == 1 and g[0] in (1, 2)), (0, lambda g: g == [1]), (1, lambda g: g == [1, 2]), (3, lambda g: g == [])])
def test_seal_oracle_ambig_example(ambig, check):
    """SealGuide.txt:27 -- 'if a read shares 2 kmers with reference A, 2 with reference B, and 1 with reference C, it will choose between A
    and B since they are equally good and both better than C' (random); first / all / toss as the same paragraph describes them."""
    k = 15
    A, Bs, Cs = rnd(41, 200), rnd(42, 200), rnd(43, 200)
    read = A[30:30 + k + 1] + Bs[50:50 + k + 1] + Cs[70:70 + k]
    o = seal_oracle()([A, Bs, Cs], k=k, maskMiddle=0, ambigMode=ambig)
    got, info = o.process(read, None, 0)
    assert check(got) and info[2] == 2, (ambig, got, info)


# ---------------------------------------------------------------------------------------------- the HIP operators
@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_hip_gives_the_documented_answer(case):
    from bbtools_amd import bbduk as B
    name, args, okw, refs, reads, paired, want_a, want_fl = case
    d = B.BBDuk(args, refs=refs)
    a, ids, fl = d.process_reads(reads, paired)
    assert (want_a is None or a.tolist() == want_a) and fl.tolist() == want_fl, (name, a.tolist(), fl.tolist())
    if name in IDS:
        assert ids.tolist() == IDS[name]
    o, (oa, oi, of) = run_oracle(okw, refs, reads, paired)        # and, as everywhere, the restatement bit for bit
    assert np.array_equal(a, oa) and np.array_equal(ids, oi) and np.array_equal(fl, of)
    d.close()


@pytest.mark.gpu
def test_hip_stored_kmers_as_documented():
    from bbtools_amd import bbduk as B
    kmer = rnd(21, 31)
    for args, want in (("k=31 mm=f", 1), ("k=31 mm=f hdist=1", 94)):
        d = B.BBDuk(args, refs=[kmer]); assert d.stored_kmers == d.gpu.table_size == want; d.close()
        d = B.BBDuk(args, refs=[kmer], build="device"); assert d.stored_kmers == want; d.close()
    d = B.BBDuk("k=31 mm=f rskip=4", refs=[rnd(22, 430)]); assert d.stored_kmers == 100; d.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cz,want", [(0, [1]), (2, [1, 2]), (7, [1, 2, 3]), (9999, [1, 2, 3])])
def test_seal_hip_clearzone_example(cz, want):
    from bbtools_amd.seal import Seal
    refs, read = _seal_guide_read()
    g = Seal("k=15 mm=f ambig=all clearzone=%d" % cz, refs=refs, max_scaffolds=8)
    sites, assigned, mx, ids, flags = g.process_reads([read], False, 0, max_ids=8)
    n = int(assigned[0])
    assert np.asarray(ids).reshape(1, -1)[0, :n].tolist() == want and int(mx[0]) == 10
    g.close()


# ---------------------------------------------------------------------------------------------- the other operators' descriptions (bbduk.sh)
def _bits(mask, lo, hi):
    return [int((mask[i >> 5] >> (i & 31)) & 1) for i in range(lo, hi)]


def _kmask_cases():
    """bbduk.sh:195 'kmask= Replace bases matching ref kmers with another symbol ... and processes short kmers on both ends if mink is set'."""
    mid = rnd(61, 40) + ADAPTER[:30] + rnd(62, 30)                  # six adapter 25-mers: bases 40..69 match, nothing else
    right = rnd(63, 88) + ADAPTER[:12]                              # the adapter's first 12 bases at the read's right end
    left = util.revcomp(ADAPTER[:12]) + rnd(64, 88)                 # and, reverse-complemented, at its left end ("both ends")
    return [("ktrim=n k=25 mm=f", dict(k=25, maskMiddle=0, ktrimN=1), mid, 30, (40, 70)),
            ("ktrim=n k=25 mm=f mink=8", dict(k=25, maskMiddle=0, ktrimN=1, mink=8), right, 12, (88, 100)),
            ("ktrim=n k=25 mm=f mink=8", dict(k=25, maskMiddle=0, ktrimN=1, mink=8), left, 12, (0, 12)),
            ("ktrim=n k=25 mm=f", dict(k=25, maskMiddle=0, ktrimN=1), right, 0, None)]      # without mink the 12 bases are too short (Guide :39)


def _ksplit_cases():
    """bbduk.sh:200 'ksplit=f For single-ended reads only.  Reads will be split into pairs around the kmer.  If the kmer is at the end of the read, it
    will be trimmed instead.  Singletons will go to out, and pairs will go to outm.'   (bases removed, leftmost, rightmost, sent to outm)"""
    # (the split read's count of removed bases is left unchecked: the reference cuts the second piece with Read.subRead(rightmost+1, length-1), whose end is
    # exclusive -- stream/Read.java:3731 -- so the piece loses its last base, 26 bases go instead of the k-mer's 25; both restatements carry the quirk)
    return [(rnd(65, 40) + ADAPTER[:25] + rnd(66, 35), (None, 40, 64, True)),      # split around the k-mer: the two sides leave as a pair
            (rnd(67, 75) + ADAPTER[:25], (25, 75, 99, False)),                     # the k-mer at the end: trimmed, a singleton
            (rnd(68, 100), (0, -1, -1, False))]


def _tips_cases():
    """bbduk.sh:193 'ktrimtips=0 Set this to a positive number to perform ktrim on both ends, examining only the outermost X bases.'   (right, left)"""
    both = ADAPTER[-25:] + rnd(69, 50) + ADAPTER[:25]
    middle = rnd(70, 40) + ADAPTER[:25] + rnd(71, 40)               # an adapter 25-mer 40 bases from either end: outside the outermost 30
    return [(both, (25, 25)), (middle, (0, 0))]


def test_oracle_kmask_ksplit_ktrimtips_as_documented():
    for args, okw, read, want_n, span in _kmask_cases():
        o = Oracle(**okw); o.add_ref(ADAPTER)
        b, off = pack_reads([read])
        a, ids, fl, mask = o.process_batch_mask(b, off, False)
        assert int(a[0]) == want_n and sum(_bits(mask, 0, len(read))) == want_n, (args, int(a[0]))
        if span:
            assert all(_bits(mask, *span))
    o = Oracle(k=25, maskMiddle=0, ksplit=1); o.add_ref(ADAPTER)
    for read, (x, lm, rm, outm) in _ksplit_cases():
        b, off = pack_reads([read])
        a, ids, fl, l, r = o.process_batch_split(b, off)
        assert (int(l[0]), int(r[0]), bool(fl[0] & 2)) == (lm, rm, outm) and (x is None or int(a[0]) == x), (read, a, l, r, fl)
    o = Oracle(k=25, maskMiddle=0, ktrimLeft=1, ktrimRight=1, restrictLeft=30, restrictRight=30); o.add_ref(ADAPTER)
    for read, (wr, wl) in _tips_cases():
        b, off = pack_reads([read])
        xr, xl, ids, fl = o.process_batch_tips(b, off, False)
        assert (int(xr[0]), int(xl[0])) == (wr, wl), (read, xr, xl)


@pytest.mark.gpu
def test_hip_kmask_ksplit_ktrimtips_as_documented():
    from bbtools_amd import bbduk as B
    for args, okw, read, want_n, span in _kmask_cases():
        d = B.BBDuk(args, refs=[ADAPTER])
        b, off = pack_reads([read])
        a, ids, fl, mask = d.gpu.kmask_batch(b, off, False)
        assert int(a[0]) == want_n and sum(_bits(mask, 0, len(read))) == want_n and (not span or all(_bits(mask, *span))), (args, int(a[0]))
        d.close()
    d = B.BBDuk("ksplit=t k=25 mm=f", refs=[ADAPTER])
    for read, (x, lm, rm, outm) in _ksplit_cases():
        b, off = pack_reads([read])
        a, ids, fl, l, r = d.gpu.ksplit_batch(b, off)
        assert (int(l[0]), int(r[0]), bool(fl[0] & 2)) == (lm, rm, outm) and (x is None or int(a[0]) == x)
    d.close()
    d = B.BBDuk("ktrimtips=30 k=25 mm=f", refs=[ADAPTER])
    for read, (wr, wl) in _tips_cases():
        b, off = pack_reads([read])
        xr, xl, ids, fl = d.gpu.ktrimtips_batch(b, off, False)
        assert (int(xr[0]), int(xl[0])) == (wr, wl)
    d.close()
