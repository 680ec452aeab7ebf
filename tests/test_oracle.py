"""CPU tests of the checker itself: known answers, the reference's own assertion identities, golden
fixtures (produced by oracle/spec.py) replayed through the C oracle, and a differential fuzz between the
two restatements.  No GPU."""
import random

import numpy as np
import pytest

from oracle.oracle_ffi import Oracle, lib, pack_reads
from oracle.spec import Spec, Args, rcomp_int
from tests import util


def enc(s):
    v = 0
    for ch in s:
        v = (v << 2) | "ACGT".index(ch)
    return v


# ---- hand-derived known answers
def test_kat_encode_and_rcomp():
    assert enc("ACGT") == 27
    assert lib().bbo_rcomp(27, 4) == 27                       # ACGT is its own reverse complement
    assert lib().bbo_rcomp(enc("AAAC"), 4) == enc("GTTT")
    for b, (x, x2, d) in {"A": (0, 3, 0), "C": (1, 2, 1), "G": (2, 1, 2), "T": (3, 0, 3), "U": (3, 0, 3),
                          "a": (0, 3, 0), "t": (3, 0, 3), "u": (3, 0, 3), "N": (0, 0, -1), ".": (0, 0, -1)}.items():
        assert lib().bbo_base_to_number0(ord(b)) == x
        assert lib().bbo_base_to_complement_number0(ord(b)) == x2
        assert lib().bbo_base_to_number(ord(b)) == d


def test_kat_derived_constants():
    o = Oracle(k=23, ktrimRight=1)                            # C1
    assert o.constant("kmask") == 1 << 46
    assert o.constant("middleMask") == ~(3 << 22)
    assert (o.constant("minlen"), o.constant("minlen2"), o.constant("forbidNs")) == (22, 11, 1)
    o = Oracle(k=23, mink=11, hdist=1, ktrimRight=1)          # C2: mink switches maskMiddle off AFTER minlen2
    assert o.constant("middleMask") == -1
    assert (o.constant("minlen2"), o.constant("forbidNs"), o.constant("useShortKmers")) == (11, 0, 1)
    assert o.constant("hdist2") == 1
    o = Oracle(k=31, hdist=1)                                 # C3
    assert o.constant("middleMask") == ~(3 << 30)
    assert (o.constant("minlen"), o.constant("minlen2"), o.constant("forbidNs")) == (30, 15, 0)
    o = Oracle(k=31)                                          # C4
    assert o.constant("forbidNs") == 1
    o = Oracle(k=24, maskMiddle=1)                            # even k masks two bases
    assert o.constant("midMaskLen") == 2 and o.constant("middleMask") == ~(15 << 22)
    o = Oracle(k=23, maskMiddle=0)
    assert o.constant("minlen2") == 23 and o.constant("middleMask") == -1
    o = Oracle()                                              # k defaults to 27 (BBDukParser.java:163)
    assert o.constant("k") == 27


def test_rcomp_identities():
    """The identities the reference asserts under -ea (BBDukProcessorS.java:1561-1568)."""
    rng = random.Random(5)
    for _ in range(2000):
        k = rng.randint(1, 31)
        v = rng.getrandbits(2 * k)
        r = lib().bbo_rcomp(v, k)
        assert r == rcomp_int(v, k)
        assert lib().bbo_rcomp(r, k) == v


def test_key_count_upper_bound():
    """docs/guides/BBDukGuide.txt:34: at most 1+(3k)^hdist keys per reference k-mer."""
    ref = util.rand_seq(random.Random(9), 200)
    o0 = Oracle(k=21); o0.add_ref(ref)
    o1 = Oracle(k=21, hdist=1); o1.add_ref(ref)
    n0, n1 = o0.stored_kmers, o1.stored_kmers
    assert n0 == 200 - 21 + 1
    assert n0 < n1 <= n0 * (1 + 3 * 21)


def test_whole_read_adapter_leaves_one_base():
    o = Oracle(k=23, ktrimRight=1); o.load_fasta(util.ADAPTERS)
    ad = b"AGATCGGAAGAGCACACGTCTGAACTCCAGTCACATCACGATCTCGTATGCCGTCTTCTGCTTG"
    x, id0 = o.ktrim(ad)
    assert (x, id0) == (len(ad) - 1, 1)                        # TrimRead clamps to one surviving base
    b, off = pack_reads([ad])
    a, ids, fl = o.process_batch(b, off, False)
    assert fl[0] == 3                                          # 1 < minlength=10 -> discarded, removed


def test_n_matches_a_under_hdist1():
    """with hdist>=1 forbidNs is false; N encodes as A forward and 0 in the reverse k-mer."""
    ref = b"GATTACAGATTACACCAGTAGCATCGA"          # 27 bases, k=27 -> one k-mer
    k = len(ref)
    read = bytearray(ref); read[5] = ord("N")      # ref[5] == 'C' -> reads as A: one substitution away
    for kw, expect_hit in ((dict(k=k, hdist=1, maskMiddle=0, rcomp=0), True), (dict(k=k, hdist=0, maskMiddle=0, rcomp=0), False)):
        o = Oracle(**kw); o.add_ref(ref)
        f, _ = o.count_set_kmers(bytes(read))
        assert (f > 0) == expect_hit


# ---- golden fixtures through the C oracle
@pytest.mark.parametrize("name", util.golden_names())
def test_golden_c_oracle(name):
    fx = util.load_golden(name)
    o = Oracle(**util.oracle_kwargs_from_spec(fx["spec_args"]))
    if "fasta" in fx["ref"]:
        o.load_fasta(util.ROOT + "/" + fx["ref"]["fasta"])
    else:
        for r in fx["ref"]["literal"]:
            o.add_ref(r.encode())
    assert o.stored_kmers == fx["stored_kmers"]
    assert o.num_scaffolds == fx["num_scaffolds"]
    b, off = pack_reads([r.encode() for r in fx["reads"]])
    a, ids, fl = o.process_batch(b, off, fx["paired"], nthreads=2)
    got = [list(t) for t in zip(a.tolist(), ids.tolist(), fl.tolist())]
    assert got == fx["expected"]
    assert o.counters().tolist() == fx["counters"]


# ---- differential fuzz: rolling C restatement vs string-based closed form
FUZZ = [dict(k=23, ktrimRight=1), dict(k=23, ktrimRight=1, mink=11, hdist=1), dict(k=31, hdist=1), dict(k=31),
        dict(k=13, ktrimLeft=1, mink=6, hdist=1, hdist2=0), dict(k=12, maskMiddle=0, qhdist=1),
        dict(k=11, ktrimRight=1, mink=5, qhdist=1, qhdist2=1, rcomp=0),
        dict(k=15, ktrimRight=1, restrictRight=40, forbidN=1, hdist=1), dict(k=15, ktrimLeft=1, restrictLeft=30, trimPad=2),
        dict(k=20, maxBadKmers0=2, requireBothBad=1, minLenFraction=0.5),
        dict(k=14, ktrimRight=1, ktrimExclusive=1, mink=8, minLenFraction=0.9), dict(k=9, hdist=2, maskMiddle=0),
        dict(k=16, ktrimRight=1, skipR2=1), dict(k=10, qhdist=2, maskMiddle=0),
        dict(k=13, ktrimRight=1, mink=7, hdist=1, trimPairsEvenly=1), dict(k=17, ktrimRight=1, trimPairsEvenly=1, requireBothBad=1, minReadLength=30),
        dict(k=12, qSkip=3), dict(k=14, ktrimRight=1, mink=6, qSkip=2), dict(k=11, speed=5, hdist=1), dict(k=15, ktrimLeft=1, mink=9, speed=9, qhdist=1),
        dict(k=11, minKmerFraction=0.3), dict(k=13, minKmerFraction=0.1, maxBadKmers0=3, requireBothBad=1, forbidN=1), dict(k=12, minCoveredFraction=0.4),
        dict(k=15, minCoveredFraction=0.9, hdist=1, restrictLeft=60), dict(k=9, minCoveredFraction=0.05, minKmerFraction=0.5, maskMiddle=0),
        dict(k=13, minSkip=3, maxSkip=3, ktrimRight=1, mink=6), dict(k=11, minSkip=2, maxSkip=5, hdist=1), dict(k=9, minSkip=4, maxSkip=4, ktrimLeft=1, mink=5, hdist=1, hdist2=0),
        # trimfailuresto1bp (BBDukParser.java:105-109, 774; BBDukProcessorS.java:1431, 1464-1488)
        dict(k=13, ktrimRight=1, mink=6, trimFailuresTo1bp=1), dict(k=15, ktrimRight=1, trimPairsEvenly=1, trimFailuresTo1bp=1, minReadLength=40),
        dict(k=11, ktrimLeft=1, trimFailuresTo1bp=1, minLenFraction=0.6), dict(k=12, trimFailuresTo1bp=1, hdist=1), dict(k=12, trimFailuresTo1bp=1, minCoveredFraction=0.3, minReadLength=1),
        dict(k=14, trimFailuresTo1bp=1, maxBadKmers0=2, requireBothBad=1)]
BOOLS = ("maskMiddle", "rcomp", "forbidN", "ktrimRight", "ktrimLeft", "requireBothBad", "ktrimExclusive", "skipR1", "skipR2", "trimPairsEvenly", "ktrimN", "kmaskFullyCovered", "trimFailuresTo1bp")
KMASK_FUZZ = [dict(k=11, ktrimN=1, trimFailuresTo1bp=1, minReadLength=30), dict(k=11, ktrimN=1, kmaskFullyCovered=1), dict(k=13, ktrimN=1, mink=6, hdist=1, kmaskFullyCovered=1),
              dict(k=15, ktrimN=1, trimPad=2, forbidN=1, restrictRight=50, kmaskFullyCovered=1), dict(k=12, ktrimN=1, mink=5, trimPad=-2, restrictLeft=70, kmaskFullyCovered=1, qhdist=1),
              dict(k=11, ktrimN=1), dict(k=13, ktrimN=1, mink=6, hdist=1), dict(k=15, ktrimN=1, trimPad=2, forbidN=1, restrictRight=50),
              dict(k=12, ktrimN=1, mink=5, trimPad=-2, qhdist=1, maskMiddle=0), dict(k=17, ktrimN=1, requireBothBad=1, minReadLength=40, skipR1=1)]


def spec_mask_words(reads, masks, nwords):
    out = np.zeros(nwords, np.uint32); pos = 0
    for r, m in zip(reads, masks):
        for b in range(len(r)):
            if (m >> b) & 1:
                out[(pos + b) >> 5] |= np.uint32(1 << ((pos + b) & 31))
        pos += len(r)
    return out


TIPS_FUZZ = [dict(k=11, ktrimLeft=1, ktrimRight=1), dict(k=13, ktrimLeft=1, ktrimRight=1, mink=6, hdist=1),
             dict(k=15, ktrimLeft=1, ktrimRight=1, restrictLeft=30, restrictRight=25, trimPad=1, forbidN=1),
             dict(k=12, ktrimLeft=1, ktrimRight=1, mink=5, trimPairsEvenly=1, ktrimExclusive=1, maskMiddle=0),
             dict(k=17, ktrimLeft=1, ktrimRight=1, requireBothBad=1, minReadLength=40, qhdist=1),
             dict(k=13, ktrimLeft=1, ktrimRight=1, mink=7, trimFailuresTo1bp=1, trimPairsEvenly=1, minReadLength=30)]


@pytest.mark.parametrize("ci", range(len(TIPS_FUZZ)))
def test_differential_fuzz_ktrimtips(ci):
    """ktrim=rl (tips): totals, ids, flags, counters and the (right, left) split of both restatements."""
    kw = TIPS_FUZZ[ci]
    for seed in range(5):
        rng = random.Random(9000 + 100 * ci + seed)
        refs = [util.rand_seq(rng, rng.randint(8, 80), 0.01) for _ in range(6)]
        o = Oracle(**kw)
        s = Spec(Args(**{k: (bool(v) if k in BOOLS else v) for k, v in kw.items()}))
        for r in refs:
            o.add_ref(r); s.add_ref(r)
        reads = util.fuzz_reads(rng, refs, 40, [0, 1, 5, 10, 11, 22, 23, 30, 31, 40, 75, 150], junk=True)
        for paired in (True, False):
            o.reset_counters()
            s.counters = [0] * 16; s.scafReads = [0] * s.nscaf; s.scafBases = [0] * s.nscaf; s.tips = []
            b, off = pack_reads(reads)
            xr, xl, ids, fl = o.process_batch_tips(b, off, paired, nthreads=3)
            want = s.process_batch(reads, paired)
            assert list(zip((xr + xl).tolist(), ids.tolist(), fl.tolist())) == want
            assert list(zip(xr.tolist(), xl.tolist())) == s.tips
            assert o.counters().tolist() == s.all_counters()
        assert any(l > 0 or r > 0 for r, l in s.tips)


@pytest.mark.parametrize("ci", range(len(KMASK_FUZZ)))
def test_differential_fuzz_kmask(ci):
    """ktrim=n (kmask): masked-base counts, ids, flags, counters and the per-base masks of both restatements."""
    kw = KMASK_FUZZ[ci]
    masked_total = 0
    for seed in range(5):
        rng = random.Random(7000 + 100 * ci + seed)
        refs = [util.rand_seq(rng, rng.randint(8, 80), 0.01) for _ in range(6)]
        o = Oracle(**kw)
        s = Spec(Args(**{k: (bool(v) if k in BOOLS else v) for k, v in kw.items()}))
        for r in refs:
            o.add_ref(r); s.add_ref(r)
        reads = util.fuzz_reads(rng, refs, 40, [0, 1, 5, 10, 11, 22, 23, 30, 31, 40, 75, 150], junk=True)
        for paired in (True, False):
            o.reset_counters()
            s.counters = [0] * 16; s.scafReads = [0] * s.nscaf; s.scafBases = [0] * s.nscaf; s.masks = []
            b, off = pack_reads(reads)
            a, ids, fl, mask = o.process_batch_mask(b, off, paired, nthreads=3)
            assert list(zip(a.tolist(), ids.tolist(), fl.tolist())) == s.process_batch(reads, paired)
            assert o.counters().tolist() == s.all_counters()
            assert np.array_equal(mask, spec_mask_words(reads, s.masks, len(mask)))
            masked_total += int(a.sum())
    assert masked_total > 0 or kw.get("kmaskFullyCovered")      # the fuzz reads really get masked


# kbig (k>31 emulated by runs of 31-mers), findBestMatch, ksplit: longer references so that runs and several scaffolds occur
BIG_FUZZ = [dict(k=31, kbig=40), dict(k=31, kbig=32, hdist=1), dict(k=31, kbig=50, maxBadKmers0=3, requireBothBad=1), dict(k=31, kbig=45, forbidN=1, restrictLeft=120),
            dict(k=31, kbig=36, minKmerFraction=0.2, qhdist=1), dict(k=25, findBestMatch=1), dict(k=13, findBestMatch=1, hdist=1, maskMiddle=0, requireBothBad=1),
            dict(k=19, findBestMatch=1, skipR1=1, restrictRight=90),
            dict(k=31, kbig=40, minCoveredFraction=0.3)]       # mcf wins over the run counting: countCoveredBases on the 31-mers (:1038)
SPLIT_FUZZ = [dict(k=23, ksplit=1), dict(k=15, ksplit=1, mink=7, hdist=1), dict(k=13, ksplit=1, trimPad=-2, forbidN=1),
              dict(k=17, ksplit=1, mink=9, restrictLeft=100, qhdist=1), dict(k=11, ksplit=1, mink=5, restrictRight=60, rcomp=0)]


@pytest.mark.parametrize("ci", range(len(BIG_FUZZ)))
def test_differential_fuzz_kbig_findbestmatch(ci):
    kw = BIG_FUZZ[ci]
    hits = 0
    for seed in range(4):
        rng = random.Random(11000 + 100 * ci + seed)
        refs, reads = util.long_refs_and_reads(rng)
        o = Oracle(**kw)
        s = Spec(Args(**{k: (bool(v) if k in BOOLS else v) for k, v in kw.items()}))
        for r in refs:
            o.add_ref(r); s.add_ref(r)
        for paired in (True, False):
            o.reset_counters()
            s.counters = [0] * 16; s.scafReads = [0] * s.nscaf; s.scafBases = [0] * s.nscaf
            b, off = pack_reads(reads)
            a, ids, fl = o.process_batch(b, off, paired, nthreads=3)
            assert list(zip(a.tolist(), ids.tolist(), fl.tolist())) == s.process_batch(reads, paired)
            assert o.counters().tolist() == s.all_counters()
            hits += int((ids > 0).sum())
            if kw.get("findBestMatch"):                          # the lists rename=t prints (:1702, 2508-2522), whole and truncated
                for cap in (64, 2):
                    o.reset_counters()
                    a2, ids2, fl2, nids, mi, mc = o.process_batch_matches(b, off, paired, cap, nthreads=3)
                    assert np.array_equal(a2, a) and np.array_equal(ids2, ids) and np.array_equal(fl2, fl)
                    for i, r in enumerate(reads):
                        found, _ = s.find_best_match(r, (i & 1) if paired else 0, 0)
                        want = s.last_matches if found > 0 else []
                        assert nids[i] == len(want), (i, r)
                        assert [(int(mi[i, j]), int(mc[i, j])) for j in range(min(cap, len(want)))] == want[:cap]
                        assert not mi[i, len(want):].any() and not mc[i, len(want):].any()
    assert hits > 20


@pytest.mark.parametrize("ci", range(len(SPLIT_FUZZ)))
def test_differential_fuzz_ksplit(ci):
    kw = SPLIT_FUZZ[ci]
    nsplit = 0
    for seed in range(4):
        rng = random.Random(12000 + 100 * ci + seed)
        refs, reads = util.long_refs_and_reads(rng)
        refs = [r[:rng.randint(12, 60)] for r in refs]           # short inserts so that reads keep flanks on both sides
        reads += [util.rand_seq(rng, rng.randint(20, 80)) + rng.choice(refs) + util.rand_seq(rng, rng.randint(20, 80)) for _ in range(10)]
        o = Oracle(**kw)
        s = Spec(Args(**{k: (bool(v) if k in BOOLS else v) for k, v in kw.items()}))
        for r in refs:
            o.add_ref(r); s.add_ref(r)
        o.reset_counters()
        b, off = pack_reads(reads)
        a, ids, fl, lm, rm = o.process_batch_split(b, off, nthreads=3)
        assert list(zip(a.tolist(), ids.tolist(), fl.tolist())) == s.process_batch(reads, False)
        assert list(zip(lm.tolist(), rm.tolist())) == s.splits
        assert o.counters().tolist() == s.all_counters()
        nsplit += int((fl & 2).astype(bool).sum())
    assert nsplit > 3


def test_kbig_known_answers():
    """A read holding 70 bases of a scaffold has 70-40+1 = 31 40-mers of it: bkStop-bkStart-(kbig-k-1) (:1758); shorter than
    kbig -> 0 (:1727); 39 shared bases -> nine 31-mer hits but no 40-mer."""
    rng = random.Random(3)
    ref = util.rand_seq(rng, 200)
    o = Oracle(k=31, kbig=40); o.add_ref(ref)
    assert o.constant("maskMiddle") == 0 and o.constant("minlen2") == 31
    flank = lambda n: bytes(rng.choice(b"AC") for _ in range(n))       # cannot extend a G/T-flanked match
    sub = lambda a, b: b"G" + ref[a:b] + b"G"
    for piece, want in ((70, 31), (40, 1), (39, 0)):
        a0 = 50
        while ref[a0 - 1:a0] in b"G" or ref[a0 + piece:a0 + piece + 1] in b"G":
            a0 += 1
        rd = flank(20) + sub(a0, a0 + piece) + flank(20)
        b, off = pack_reads([rd])
        a, ids, fl = o.process_batch(b, off, False)
        assert (int(a[0]), int(ids[0]), int(fl[0])) == ((want, 1, 3) if want else (0, -1, 0)), piece
    b, off = pack_reads([ref[:39]])
    assert o.process_batch(b, off, False)[0][0] == 0


@pytest.mark.parametrize("ci", range(len(FUZZ)))
def test_differential_fuzz(ci):
    kw = FUZZ[ci]
    for seed in range(6):
        rng = random.Random(1000 * ci + seed)
        refs = [util.rand_seq(rng, rng.randint(5, 80), 0.01) for _ in range(6)]
        o = Oracle(**kw)
        s = Spec(Args(**{k: (bool(v) if k in BOOLS else v) for k, v in kw.items()}))
        for r in refs:
            o.add_ref(r); s.add_ref(r)
        ks, vs = o.dump_pairs()
        assert dict(zip(ks.tolist(), vs.tolist())) == s.table
        reads = util.fuzz_reads(rng, refs, 40, [0, 1, 5, 10, 11, 22, 23, 30, 31, 40, 75, 150], junk=True)
        for paired in (True, False):
            o.reset_counters()
            s.counters = [0] * 16; s.scafReads = [0] * s.nscaf; s.scafBases = [0] * s.nscaf
            b, off = pack_reads(reads)
            a, ids, fl = o.process_batch(b, off, paired, nthreads=3)
            assert list(zip(a.tolist(), ids.tolist(), fl.tolist())) == s.process_batch(reads, paired)
            assert o.counters().tolist() == s.all_counters()


def test_way_images_cover_the_map():
    """The HashArray1D images handed across the boundary hold exactly the map (kmer/HashArray.java:672)."""
    o = Oracle(k=23, mink=11, hdist=1, ktrimRight=1); o.load_fasta(util.ADAPTERS)
    ks, vs = o.dump_pairs()
    seen = {}
    for w, (prime, keys, vals, vk, vv) in enumerate(o.way_images()):
        assert len(keys) == prime + 60
        m = keys >= 0
        assert np.all(keys[m] % 7 == w)                      # key%WAYS sharding (BBDukIndexMod.java:506)
        seen.update(zip(keys[m].tolist(), vals[m].tolist()))
        seen.update(zip(vk.tolist(), vv.tolist()))
    assert seen == dict(zip(ks.tolist(), vs.tolist()))
    for key in ks[:200].tolist():
        assert o.table_get(key) == seen[key]
    assert o.table_get(12345) == -1


def test_victim_path_of_the_hash_array_holds_the_same_map():
    """HashArray1D hands keys whose probe window is full to its victim forest (kmer/HashArray.java:221-239, 434-447;
    kmer/HashForest.java:229-233).  The reference's window of 60 cells is never full for real reference sets, so the restatement walks
    a window of 2 here: a few per cent of the keys become victims, and the map -- every key, every id -- has to stay what it was."""
    from oracle import oracle_ffi as O
    args, okw, ref = util.CONFIGS["c2"]
    plain = Oracle(**okw); plain.load_fasta(ref)
    assert sum(len(w[3]) for w in plain.way_images()) == 0
    O.set_probe_window(2)
    try:
        o = Oracle(**okw); o.load_fasta(ref)
        images = o.way_images()
        nv = sum(len(w[3]) for w in images)
        assert nv > 1000 and o.stored_kmers == plain.stored_kmers
        k0, v0 = plain.dump_pairs(); k1, v1 = o.dump_pairs()
        a, b = np.argsort(k0), np.argsort(k1)
        assert np.array_equal(k0[a], k1[b]) and np.array_equal(v0[a], v1[b])
        vk = np.concatenate([w[3] for w in images]); vv = np.concatenate([w[4] for w in images])
        for key, val in list(zip(vk.tolist(), vv.tolist()))[:500]:      # a victim is found through the forest, with its id
            assert o.table_get(key) == val
    finally:
        O.set_probe_window(60)

