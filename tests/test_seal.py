"""jgi/Seal.java's k-mer path: the C restatement (oracle/seal_oracle.c) on hand-checked cases, and the HIP operator (seal_batch, through
the C ABI of include/seal_gpu.h) against it on seeded random references, reads and parameter sets."""
import random

import numpy as np
import pytest

from oracle.seal_ffi import SealOracle

ACGT = b"ACGT"


def rnd_seq(rng, n):
    return bytes(rng.choice(ACGT) for _ in range(n))


def revcomp(s):
    return s.translate(bytes.maketrans(b"ACGTacgt", b"TGCAtgca"))[::-1]


def make_refs(rng, nrefs, length, share):
    """Scaffolds that share segments, so that many k-mers belong to several of them."""
    pool = [rnd_seq(rng, rng.randint(40, 120)) for _ in range(4)]
    refs = []
    for _ in range(nrefs):
        s = bytearray()
        while len(s) < length:
            if rng.random() < share:
                p = rng.choice(pool)
                s += p if rng.random() < 0.7 else revcomp(p)
            else:
                s += rnd_seq(rng, rng.randint(20, 90))
            if rng.random() < 0.1:
                s += b"N"
        refs.append(bytes(s[:length]))
    if rng.random() < 0.3:
        refs.append(b"ACGT")                         # shorter than k: a scaffold without k-mers
    if rng.random() < 0.3:
        refs.append(refs[0])                         # an exact duplicate: every k-mer shared
    return refs


def make_reads(rng, refs, n, paired, lengths=(0, 5, 9, 10, 30, 75, 150, 151, 260, 300)):
    reads = []
    for _ in range(n * (2 if paired else 1)):
        r = rng.random()
        L = rng.choice(lengths)
        if r < 0.15:
            s = bytearray(rnd_seq(rng, L))
        else:
            ref = rng.choice(refs)
            if len(ref) <= 4:
                ref = refs[0]
            a = rng.randrange(0, max(1, len(ref) - 1))
            s = bytearray(ref[a:a + L])
            if r < 0.45:                               # a chimera: two scaffolds in one read
                ref2 = rng.choice(refs)
                b = rng.randrange(0, max(1, len(ref2) - 1))
                s = s[:len(s) // 2] + bytearray(ref2[b:b + L - len(s) // 2])
            if rng.random() < 0.5:
                s = bytearray(revcomp(bytes(s)))
            for _m in range(rng.choice([0, 0, 1, 2, 5])):
                if s:
                    s[rng.randrange(len(s))] = rng.choice(b"ACGTNnRYacgt")
        reads.append(bytes(s))
    return reads


def rand_cfg(rng):
    k = rng.choice([9, 11, 15, 21, 25, 31])
    cfg = dict(k=k, mm=rng.random() < 0.6, hdist=rng.choice([0, 0, 0, 1]) if k <= 15 else rng.choice([0, 0, 0, 0, 1]),
               rcomp=rng.random() < 0.85, fn=rng.random() < 0.3, match=rng.choice(["all", "all", "first", "unique"]),
               ambig=rng.choice(["random", "first", "all", "toss"]), kpt=rng.random() < 0.6, mkh=rng.choice([1, 1, 2, 5]),
               mkf=rng.choice([0.0, 0.0, 0.05, 0.3]), cz=rng.choice([0, 0, 1, 3, 20]), czf=rng.choice([0.0, 0.0, 0.0, 0.02, 0.2]), qskip=rng.choice([1, 1, 1, 2, 3]),
               speed=rng.choice([0, 0, 0, 4]), rl=rng.choice([0, 0, 0, 40]), rr=rng.choice([0, 0, 0, 50]),
               minlength=rng.choice([10, 10, 1, 40]), maxlength=rng.choice([0, 0, 200]), mlf=rng.choice([0.0, 0.0, 0.5]),
               rieb=rng.random() < 0.7, rskip=rng.choice([0, 0, 0, 2, 3]))
    if cfg["mm"] and rng.random() < 0.3:
        cfg["mmlen"] = rng.choice([1, 2, 3])
    return cfg


def cfg_args(c):
    a = ["k=%d" % c["k"], "mm=%s" % (str(c["mmlen"]) if c.get("mmlen") else ("t" if c["mm"] else "f")), "hdist=%d" % c["hdist"],
         "rcomp=%s" % ("t" if c["rcomp"] else "f"), "fn=%s" % ("t" if c["fn"] else "f"), "match=%s" % c["match"], "ambig=%s" % c["ambig"],
         "kpt=%s" % ("t" if c["kpt"] else "f"), "mkh=%d" % c["mkh"], "mkf=%g" % c["mkf"], "cz=%d" % c["cz"], "czf=%g" % c["czf"], "qskip=%d" % c["qskip"],
         "speed=%d" % c["speed"], "restrictleft=%d" % c["rl"], "restrictright=%d" % c["rr"], "minlength=%d" % c["minlength"],
         "mlf=%g" % c["mlf"], "rieb=%s" % ("t" if c["rieb"] else "f"), "rskip=%d" % c["rskip"]]
    if c["maxlength"]:
        a.append("maxlength=%d" % c["maxlength"])
    return " ".join(a)


def cfg_oracle(c):
    return dict(k=c["k"], maskMiddle=1 if c["mm"] else 0, midMaskLen=c.get("mmlen", 0), hdist=c["hdist"], rcomp=1 if c["rcomp"] else 0,
                forbidN=1 if c["fn"] else 0, matchMode={"all": 0, "first": 1, "unique": 2}[c["match"]],
                ambigMode={"first": 0, "all": 1, "random": 2, "toss": 3}[c["ambig"]], keepPairsTogether=1 if c["kpt"] else 0,
                minKmerHits=c["mkh"], minKmerFraction=c["mkf"], clearzone=c["cz"], clearzoneFraction=c["czf"], qSkip=c["qskip"], speed=c["speed"], restrictLeft=c["rl"],
                restrictRight=c["rr"], minReadLength=c["minlength"], maxReadLength=c["maxlength"] or 0x7FFFFFFF, minLenFraction=c["mlf"],
                requireBothBad=0 if c["rieb"] else 1, refSkip=c["rskip"])


# ---------------------------------------------------------------------------------------------- the restatement, hand-checked
def test_oracle_counts_every_kmer_of_a_contained_read():
    rng = random.Random(5)
    refs = [rnd_seq(rng, 200), rnd_seq(rng, 200)]
    o = SealOracle(refs, k=21, maskMiddle=0)
    got, info = o.process(refs[1][30:130], None, 0)
    assert got == [2] and info[0] == 1 and info[1] == 1 and info[2] == 100 - 21 + 1 and info[4] == 0     # findBestMatch :2893-2897
    c, reads, bases, frags, ambig = o.counters()
    assert (c["readsMatched"], c["basesMatched"], c["readsUnmatched"], c["fragsIn"]) == (1, 100, 0, 1)
    assert reads.tolist() == [0, 0, 1] and bases.tolist() == [0, 0, 100] and frags.tolist() == [0, 0, 1] and ambig.tolist() == [0, 0, 0]


def test_oracle_ambiguous_modes_and_clearzone():
    rng = random.Random(6)
    shared = rnd_seq(rng, 80)
    refs = [rnd_seq(rng, 50) + shared, shared + rnd_seq(rng, 50), rnd_seq(rng, 130)]
    read = shared[5:75]                              # every k-mer belongs to scaffolds 1 and 2
    for ambig, want in ((2, lambda nid: [1 + nid % 2]), (0, lambda nid: [1]), (1, lambda nid: [1, 2]), (3, lambda nid: [])):
        o = SealOracle(refs, k=15, ambigMode=ambig)
        for nid in (0, 1, 7):
            got, info = o.process(read, None, nid)
            assert got == want(nid) and info[0] == 2 and info[2] == 70 - 15 + 1, (ambig, nid, got, info)
    # a read with 30 k-mers of scaffold 3 and 10 of scaffolds 1+2: only the top scaffold without a clear zone, all three with cz >= 20
    chim = refs[2][:44] + shared[:24]
    assert SealOracle(refs, k=15, ambigMode=1).process(chim, None, 0)[0] == [3]
    assert SealOracle(refs, k=15, ambigMode=1, clearzone=20).process(chim, None, 0)[0] == [3, 1, 2]        # list order = first seen (:2896)
    # minkmerhits above the best count: unmatched
    o = SealOracle(refs, k=15, minKmerHits=100)
    assert o.process(read, None, 0)[0] == [] and o.counters()[0]["readsUnmatched"] == 1


def test_oracle_match_modes_and_length_rule():
    rng = random.Random(7)
    shared = rnd_seq(rng, 60)
    refs = [shared + rnd_seq(rng, 60), shared + rnd_seq(rng, 60)]
    read = refs[1][40:100]                           # starts inside the shared part, ends in scaffold 2's own
    assert SealOracle(refs, k=15, matchMode=0).process(read, None, 0)[1][2] == 60 - 15 + 1
    got, info = SealOracle(refs, k=15, matchMode=1, ambigMode=1).process(read, None, 0)
    assert got == [1, 2] and info[2] == 1            # first k-mer only: it is shared
    got, info = SealOracle(refs, k=15, matchMode=2, ambigMode=1).process(read, None, 0)
    assert got == [2] and info[2] == 7               # six shared k-mers counted, then the first unique one ends the scan (:2901)
    o = SealOracle(refs, k=15)
    got, info = o.process(b"ACGTACG", read, 0)       # mate 1 shorter than minlength=10: the pair is removed (:2125-2130)
    c = o.counters()[0]
    assert got == [] and info[4] == 1 and c["readsQFiltered"] == 2 and c["basesQFiltered"] == 2 * 67 and c["readsQTrimmed"] == 2
    o = SealOracle(refs, k=15, requireBothBad=1)
    got, info = o.process(b"ACGTACG", read, 0)
    assert got == [2] and info[4] == 0


def test_oracle_only_the_letter_N_resets():
    rng = random.Random(8)
    ref = rnd_seq(rng, 120)
    o = SealOracle([ref], k=15, maskMiddle=0)
    clean = o.process(ref[10:80], None, 0)[1][2]
    withN = bytearray(ref[10:80]); withN[35] = ord("N")
    witha = bytearray(ref[10:80]); witha[35] = ord("n")
    assert clean == 56 and o.process(bytes(withN), None, 0)[1][2] == 56 - 15
    # 'n' is no reset (:2885): the windows over it are looked up with an A in its place
    assert o.process(bytes(witha), None, 0)[1][2] == (56 if ref[45] in b"Aa" else 56 - 15)


# ---------------------------------------------------------------------------------------------- HIP operator vs the restatement
def run_case(seed, nreads=160, lengths=None, ref_len=None):
    from bbtools_amd.seal import Seal
    rng = random.Random(seed)
    cfg = rand_cfg(rng)
    refs = make_refs(rng, rng.choice([2, 3, 6, 12]), ref_len or rng.choice([150, 300, 600]), rng.choice([0.0, 0.3, 0.7]))
    paired = rng.random() < 0.6
    reads = make_reads(rng, refs, nreads, paired, *([lengths] if lengths else []))
    first_id = rng.choice([0, 17, 1 << 33])
    o = SealOracle(refs, **cfg_oracle(cfg))
    g = Seal(cfg_args(cfg), refs=refs, max_scaffolds=len(refs) + 2)
    assert g.num_scaffolds == o.num_scaffolds and g.table_pairs == len(o.pairs()[0]), (cfg, "table")
    want = o.process_reads(reads, paired, first_id, max_ids=8)
    got = g.process_reads(reads, paired, first_id, max_ids=8)
    for name, w, h in zip(("sites", "assigned", "max", "ids", "flags"), want, got):
        assert np.array_equal(w, h), (seed, cfg, name, np.nonzero(np.asarray(w).reshape(len(reads), -1) != np.asarray(h).reshape(len(reads), -1))[0][:5])
    wc = o.counters(); gc = g.counters()
    assert wc[0] == gc[0], (seed, cfg, wc[0], gc[0])
    S = o.num_scaffolds
    for q in range(1, 5):
        assert np.array_equal(wc[q], gc[q][:S]) and not gc[q][S:].any(), (seed, cfg, q)
    g.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(40))
def test_seal_fuzz_parity(seed):
    run_case(1000 + seed)


@pytest.mark.gpu
@pytest.mark.parametrize("nscaf,ambig,kpt", [(3, "random", True), (70, "all", False), (300, "random", True), (300, "all", False)])
def test_seal_scaffold_counters_through_the_wave_cache(nscaf, ambig, kpt):
    """The per-scaffold counters (stats= / rpkm= columns) are summed in a 64-entry cache per wave (round 5: one atomic per assigned unit and column
    serialised the grid when a run had few scaffolds -- 4 scaffolds 17 -> 64 Gbases/s).  Few scaffolds (every unit hits the cache), and more scaffolds
    than a wave's cache holds with enough units per wave that it has to start over: every counter equal to the restatement's."""
    from bbtools_amd.seal import Seal
    rng = random.Random(900 + nscaf)
    shared = rnd_seq(rng, 60)
    refs = [rnd_seq(rng, 120) + (shared if i % 7 == 0 else b"") + rnd_seq(rng, 60) for i in range(nscaf)]
    n = 400_000 if nscaf > 64 else 60_000                      # ~100 units per wave on 256 CUs x 16 waves
    big = b"".join(refs)
    starts = np.cumsum([0] + [len(r) for r in refs])
    reads = []
    prng = np.random.default_rng(nscaf)
    which = prng.integers(0, nscaf, n); offs = prng.integers(0, 60, n); lens = prng.choice([60, 100, 150], n)
    for i in range(n):
        a = int(starts[which[i]] + offs[i])
        reads.append(big[a:a + int(lens[i])] if i % 50 else rnd_seq(rng, 80))      # (a read may run into the next scaffold: several sites)
    cfg = dict(k=21, mm=True, hdist=0, rcomp=True, fn=False, match="all", ambig=ambig, kpt=kpt, mkh=1, mkf=0.0, cz=0, czf=0.0, qskip=1, speed=0, rl=0, rr=0,
               minlength=10, maxlength=0, mlf=0.0, rieb=True, rskip=0)
    o = SealOracle(refs, **cfg_oracle(cfg))
    g = Seal(cfg_args(cfg), refs=refs, max_scaffolds=nscaf + 2)
    want = o.process_reads(reads, True, 5, max_ids=8)
    got = g.process_reads(reads, True, 5, max_ids=8)
    for name, w, h in zip(("sites", "assigned", "max", "ids", "flags"), want, got):
        assert np.array_equal(w, h), name
    wc = o.counters(); gc = g.counters()
    assert wc[0] == gc[0]
    for q in range(1, 5):
        assert np.array_equal(wc[q], gc[q][:o.num_scaffolds]) and not gc[q][o.num_scaffolds:].any(), q
    assert int((np.asarray(wc[1]) > 0).sum()) >= min(nscaf, 3)
    g.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(8))
def test_seal_units_beyond_the_wave_planes(seed):
    """bbduk_seal_wave_kernel holds units of up to ~690 bases in a wave's own planes and leaves longer ones to the tile kernel launched
    behind it: batches that mix both, in either order, with pairs whose mates fall on either side of the limit."""
    run_case(5000 + seed, nreads=120, lengths=(0, 30, 150, 300, 340, 352, 400, 700, 1500), ref_len=2500)


@pytest.mark.gpu
@pytest.mark.parametrize("k,rcomp", [(9, True), (11, False), (13, True)])
def test_seal_hdist2_table_is_the_reference_neighbourhood(k, rcomp):
    """hdist=2: the loader emits every neighbour once (ascending positions) where Seal.java:1890-1921 nests two rounds of single
    substitutions; the (k-mer, scaffold) SET and every read's answer have to be the restatement's."""
    from bbtools_amd.seal import Seal
    rng = random.Random(900 + k)
    refs = make_refs(rng, 3, 120, 0.3)
    o = SealOracle(refs, k=k, hdist=2, maskMiddle=0, rcomp=1 if rcomp else 0)
    g = Seal("k=%d hdist=2 mm=f rcomp=%s" % (k, "t" if rcomp else "f"), refs=refs, max_scaffolds=8)
    keys, ids = o.pairs()
    assert g.table_pairs == len(keys) and g.table_keys == len(np.unique(keys))
    reads = make_reads(rng, refs, 120, True)
    want = o.process_reads(reads, True, 3, max_ids=8); got = g.process_reads(reads, True, 3, max_ids=8)
    for w, h in zip(want, got):
        assert np.array_equal(w, h)
    assert o.counters()[0] == g.counters()[0]
    g.close()


@pytest.mark.gpu
def test_seal_uploaded_pairs_and_defaults():
    """A table handed over as (k-mer, scaffold) pairs (what a JVM-built HashArrayHybridFast holds) with Seal's defaults (k=31 mm=t ambig=random)."""
    from bbtools_amd.seal import Seal
    rng = random.Random(77)
    refs = make_refs(rng, 5, 500, 0.5)
    o = SealOracle(refs)
    keys, ids = o.pairs()
    perm = np.random.default_rng(3).permutation(len(keys))
    g = Seal("", pairs=(np.concatenate([keys[perm], keys[:50]]), np.concatenate([ids[perm], ids[:50]])), max_scaffolds=64)   # any order, repeats
    assert g.table_pairs == len(keys) and g.table_keys == len(np.unique(keys))
    reads = make_reads(rng, refs, 300, True)
    want = o.process_reads(reads, True, 5); got = g.process_reads(reads, True, 5)
    for w, h in zip(want, got):
        assert np.array_equal(w, h)
    assert o.counters()[0] == g.counters()[0]


@pytest.mark.gpu
def test_seal_more_than_64_scaffolds_in_one_read_is_an_error():
    from bbtools_amd.bbduk import BBDukError
    from bbtools_amd.seal import Seal
    rng = random.Random(9)
    core = rnd_seq(rng, 60)
    refs = [core + rnd_seq(rng, 20) for _ in range(70)]
    g = Seal("k=21 ambig=all", refs=refs, max_scaffolds=128)
    with pytest.raises(BBDukError, match="64 distinct"):
        g.process_reads([core], False)
    sites, assigned, mx, ids, flags = Seal("k=21 ambig=all", refs=refs[:60], max_scaffolds=128).process_reads([core], False, max_ids=64)
    assert sites[0] == 60 and assigned[0] == 60 and mx[0] == 40 and ids[0, :60].tolist() == list(range(1, 61))


def test_seal_args_are_parsed_like_the_reference_and_refusals_are_loud():
    """No GPU needed: seal_params_from_args (Seal.java:140-470 for the parameters of this path), the exports, the struct layout."""
    import ctypes as C
    import os
    import re
    from bbtools_amd import seal as S
    from tests import util
    L = S.lib()
    for sym in S.SYMBOLS:
        assert hasattr(L, sym), sym
    txt = open(os.path.join(util.ROOT, "include", "seal_gpu.h")).read()
    decl = set(re.findall(r"\b(seal_[a-z0-9_]+)\s*\(", txt)) - {"seal_params", "seal_handle"}
    assert decl == set(S.SYMBOLS)
    assert C.sizeof(S.SealParams) == 24 * 4
    p = S.parse_args("")
    assert (p.k, p.maskMiddle, p.rcomp, p.hdist, p.matchMode, p.ambigMode, p.keepPairsTogether, p.minKmerHits, p.minReadLength, p.qSkip) == \
        (31, 1, 1, 0, S.MATCH_ALL, S.AMBIG_RANDOM, 1, 1, 10, 1)
    p = S.parse_args("k=25 mm=f hammingdistance=1 ambiguous=discard mode=firstunique kpt=f minhits=3 mkf=0.25 cz=4 fn rieb=f minlen=20 mm=3")
    assert (p.k, p.maskMiddle, p.midMaskLen, p.hdist, p.ambigMode, p.matchMode, p.keepPairsTogether, p.minKmerHits, p.clearzone, p.forbidNs,
            p.requireBothBad, p.minReadLength) == (25, 1, 3, 1, S.AMBIG_TOSS, S.MATCH_UNIQUE, 0, 3, 4, 1, 1, 20)
    assert abs(p.minKmerFraction - 0.25) < 1e-7
    assert S.parse_args("fbm=f").matchMode == S.MATCH_FIRST and S.parse_args("fum").matchMode == S.MATCH_UNIQUE
    for bad in ("qhdist=1", "edist=1", "rename=t", "ambig=sometimes", "k=abc", "processcontainedref=t"):
        with pytest.raises(S.BBDukError):
            S.parse_args(bad)


# ---------------------------------------------------------------------------------------------- committed fixtures (tests/golden/seal_*)
GOLD = __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "golden")
SEAL_MODES = {"all": 1, "first": 0, "toss": 3, "random": 2}


def _seal_fixture():
    refs, names = [], []
    for line in open(GOLD + "/seal_refs.fa").read().split("\n"):
        if line.startswith(">"): names.append(line[1:])
        elif line: refs.append(line.encode())
    lines = open(GOLD + "/seal_reads.fq").read().split("\n")
    rnames = [lines[i][1:] for i in range(0, len(lines) - 1, 4)]
    reads = [lines[i + 1].encode() for i in range(0, len(lines) - 1, 4)]
    return refs, names, rnames, reads


def _seal_tsv(names, rnames, res):
    sites, assigned, mx, ids, flags = res
    return "".join("%s\t%d\t%d\t%d\t%s\t%d\n" % (rnames[u], sites[u], assigned[u], mx[u], ",".join(names[i - 1] for i in ids[u][:assigned[u]]), flags[u])
                   for u in range(0, len(rnames), 2))


@pytest.mark.parametrize("mode", list(SEAL_MODES))
def test_seal_restatement_reproduces_the_committed_fixture(mode):
    refs, names, rnames, reads = _seal_fixture()
    o = SealOracle(refs, ambigMode=SEAL_MODES[mode])
    assert _seal_tsv(names, rnames, o.process_reads(reads, True, 0, max_ids=8)) == open(GOLD + "/seal_%s.tsv" % mode).read()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", list(SEAL_MODES))
def test_seal_hip_reproduces_the_committed_fixture(mode):
    from bbtools_amd.seal import Seal
    refs, names, rnames, reads = _seal_fixture()
    g = Seal("ambig=%s k=31 mm=t" % mode, refs=refs, max_scaffolds=16)
    assert _seal_tsv(names, rnames, g.process_reads(reads, True, 0, max_ids=8)) == open(GOLD + "/seal_%s.tsv" % mode).read()
    c, reads_, bases_, frags_, ambig_ = g.counters()
    # the stats file of the fixture, from the device's counters (Seal.writeStats, statscolumns=5)
    rows = sorted(((names[i - 1], int(reads_[i]), int(bases_[i]), int(ambig_[i])) for i in range(1, len(names) + 1) if reads_[i] > 0), key=lambda r: (-r[2], -r[1], r[0]))
    rm, bm = 100.0 / c["readsIn"], 100.0 / c["basesIn"]
    txt = "#Total\t%d\t%d\n#Matched\t%d\t%.5f%%\t%d\n#Name\tReads\tReadsPct\tBases\tBasesPct\tAmbigReads\n" % (c["readsIn"], c["basesIn"], c["readsMatched"], rm * c["readsMatched"], c["basesMatched"])
    txt += "".join("%s\t%d\t%.5f%%\t%d\t%.5f%%\t%d\n" % (n, r, r * rm, b, b * bm, a) for n, r, b, a in rows)
    assert txt == open(GOLD + "/seal_%s_stats.txt" % mode).read()


@pytest.mark.gpu
def test_seal_counter_allreduce_world1():
    """seal_comm_create / seal_allreduce_counters with the only world a 1-GPU box can form: one rank's all-reduce leaves the vector as it is."""
    from bbtools_amd import bbduk as B
    from bbtools_amd.seal import Seal
    rng = random.Random(3)
    refs = [rnd_seq(rng, 300) for _ in range(3)]
    g = Seal("k=21", refs=refs, max_scaffolds=8)
    g.process_reads([refs[i % 3][10:110] for i in range(40)], True)
    before = g.counters()
    with pytest.raises(B.BBDukError):
        g.allreduce_counters()                         # no communicator yet
    g.comm_create(1, 0, B.comm_unique_id())
    g.allreduce_counters()
    after = g.counters()
    assert before[0] == after[0] and before[0]["readsMatched"] == 40 and all(np.array_equal(x, y) for x, y in zip(before[1:], after[1:]))
    g.close()
