"""The HBM-resident map layout (BASELINE configs[3]: a 10 GB reference) forced onto small maps with bbduk_test_hook, so that its
build (in place, on the device), its exact per-key lookup (every kernel's generic path) and its fast minimizer-sharing candidate
scan (bbduk_wave_kernel<.., BIG>) are compared with the oracle on inputs the oracle finishes in seconds.  The full-size run is
profiles/bench_c4_full.py."""
import random

import numpy as np
import pytest

from bbtools_amd import bbduk as B
from oracle.oracle_ffi import Oracle, pack_reads
from tests import util
from tests.test_gpu_parity import check_batch, check_kmask, check_tips, check_split, FUZZ

pytestmark = pytest.mark.gpu
BIG = {B.HOOK_BIG_LAYOUT: 1}
BIG52 = {B.HOOK_BIG_LAYOUT: 2}            # round 2's 52-bit line function and its pair scan (kept for A/B runs) at test sizes
BIGWIDE = {B.HOOK_BIG_LAYOUT: 3}          # the wide candidate values (gap_v52) that maps beyond 2^31 keys take, at test sizes


def big_pair(args, okw, refs=None, ref_fasta=None, build="host", hooks=BIG):
    o = Oracle(**okw)
    if refs is not None:
        d = B.BBDuk(args, refs=refs, hooks=hooks, build=build)
        for r in refs:
            o.add_ref(r)
    else:
        d = B.BBDuk(args, hooks=hooks, build=build)
        o.load_fasta(ref_fasta)
    assert d.gpu.table_size == o.stored_kmers
    return d, o


@pytest.mark.parametrize("build", ["host", "device", "device52", "devicewide", "hostwide"])
@pytest.mark.parametrize("cfg", ["c4s", "c3", "c2", "c1", "c5", "c3q", "c3mcf", "c3mkf"])
def test_big_layout_on_the_baseline_configs(cfg, build):
    args, okw, ref = util.CONFIGS[cfg]
    hooks = BIG
    if build == "device52":
        if cfg not in ("c4s", "c3"):
            pytest.skip("the 52-bit variant's own scan serves the plain kfilter")
        build, hooks = "device", BIG52
    if build in ("devicewide", "hostwide"):
        if cfg not in ("c4s", "c3", "c1", "c2"):
            pytest.skip("the wide values differ from the 32-bit ones in the scan's line function only: kfilter and ktrim=r cover it")
        build, hooks = build[:-4], BIGWIDE
    d, o = big_pair(args, okw, ref_fasta=ref, build=build, hooks=hooks)
    assert d.gpu.table_bytes >= 64 * (64 + 256 + 64)               # whole 32-slot lines
    ks, vs = o.dump_pairs()
    rng = np.random.default_rng(3)
    probe = np.concatenate([ks, rng.integers(0, 1 << 62, 50000, dtype=np.int64), np.array([-1, 0, 1 << 62], np.int64)])
    exp = np.concatenate([vs, np.array([o.table_get(int(x)) for x in probe[len(ks):]], np.int32)])
    assert np.array_equal(d.gpu.table_lookup(probe), exp)          # every key with its id, nothing else
    kw = dict(contam=B.read_fasta(util.PHIX)[0][1], contam_frac=0.05) if "phix" in args else {}
    b, off = B.synth_generate_host(B.synth_params(13, n_rate=0.004, **kw), 500, 20000)      # Ns in most reads' neighbourhoods
    for paired in (True, False):
        d.gpu.reset_counters(); o.reset_counters()
        ga, gi, gf = d.gpu.process_batch(b, off, paired)
        oa, oi, of = o.process_batch(b, off, paired, nthreads=8)
        bad = np.nonzero((ga != oa) | (gi != oi) | (gf != of))[0]
        assert len(bad) == 0, (cfg, build, paired, bad[:5], ga[bad[:5]], oa[bad[:5]])
        assert np.array_equal(d.gpu.counters(), o.counters())
    codes, undef = B.pack_bases_host(b)
    d.gpu.reset_counters()
    pa, pi, pf = d.gpu.process_batch_packed(codes, undef, off, True)
    oa, oi, of = o.process_batch(b, off, True, nthreads=8)
    assert np.array_equal(pa, oa) and np.array_equal(pi, oi) and np.array_equal(pf, of)
    assert int((oa > 0).sum()) > 100
    d.close()


@pytest.mark.parametrize("ci", [0, 1, 2, 3, 4, 7, 9, 15, 16, 17, 20, 22, 24, 26])
def test_big_layout_fuzz(ci):
    """Ragged, empty, junk-laden reads and the flag combinations of the general kernels on the big layout (k too small for its
    minimizer geometry falls back to the cache-resident layout, which is fine: the hook only asks)."""
    args, okw = FUZZ[ci]
    for seed in range(2):
        rng = random.Random(7000 + 100 * ci + seed)
        refs = [util.rand_seq(rng, rng.randint(5, 90), 0.01) for _ in range(8)]
        d, o = big_pair(args, okw, refs=refs, build="host", hooks=BIG52 if seed == 1 and ci in (0, 1, 2, 3) else (BIGWIDE if seed == 1 else BIG))
        reads = util.fuzz_reads(rng, refs, 600, [0, 1, 2, 5, 10, 11, 22, 23, 30, 31, 32, 40, 63, 64, 65, 75, 150, 151, 300], junk=True)
        check_batch(d, o, reads, True)
        check_batch(d, o, reads[:301], False)
        d.close()


def test_big_layout_long_reads_cross_the_block_seam():
    """Reads of several hundred bases: the fast scan advances 120 positions per block and hands the top lanes' windows to the next
    block; reference pieces are planted across those seams and at both ends, on both strands, with and without an N nearby."""
    rng = random.Random(5)
    ref = util.rand_seq(rng, 3000)
    for args, okw in (("k=31", dict(k=31)), ("k=31 hdist=1", dict(k=31, hdist=1)), ("k=27 mm=f", dict(k=27, maskMiddle=0)), ("k=24", dict(k=24))):
        d, o = big_pair(args, okw, refs=[ref], build="device")
        k = okw["k"]
        reads = []
        for L in (150, 151, 269, 270, 271, 300, 389, 390, 391, 520, 1000, 2400):
            for pos in (0, 1, L - k, L - k - 1, 119 + 30 - k + 1, 120 + 30 - k, 121 + 30 - k, 239 + 30 - k, 240 + 30 - k, 241 + 30 - k, L // 2):
                if pos < 0 or pos + k > L:
                    continue
                g = bytearray(util.rand_seq(rng, L))
                a = rng.randrange(len(ref) - k); piece = ref[a:a + k]
                g[pos:pos + k] = util.revcomp(piece) if rng.random() < 0.5 else piece
                if rng.random() < 0.4:
                    g[rng.randrange(L)] = ord("N")
                reads.append(bytes(g))
        reads += [util.rand_seq(rng, L, 0.003) for L in (150, 300, 1000) for _ in range(20)]
        check_batch(d, o, reads, True)
        check_batch(d, o, reads, False)
        d.close()


@pytest.mark.parametrize("mode", ["stream", "oneshot", "pairs"])
def test_big_layout_five_million_keys(mode):
    """A map the size at which the layout would be chosen without the hook's help is 2^25 keys; 4.6 M keys here (the oracle builds
    its own table), three ways in: the streaming device build (whole scaffolds per call), the one-shot device build from host
    memory, and uploaded pairs placed on the device."""
    import torch
    rng = np.random.default_rng(5)
    refs = [rng.choice(np.frombuffer(b"ACGT", np.uint8), n).tobytes() for n in (2_000_000, 1_500_000, 31, 30, 1_100_000)]
    o = Oracle(k=31)
    for r in refs:
        o.add_ref(r)
    if mode == "oneshot":
        d = B.BBDuk("k=31", refs=refs, hooks=BIG, build="device")
    elif mode == "pairs":
        d = B.BBDuk("k=31", refs=refs, hooks=BIG, build="host")
    else:
        d = B.BBDuk.__new__(B.BBDuk)
        d.host = B.HostIndex("k=31")
        for r in refs:
            d.host.add_ref(r)
        d.gpu = B.BBDukGpu(d.host.params(0))
        d.gpu.test_hook(B.HOOK_BIG_LAYOUT, 1)
        d.gpu.build_begin(sum(len(r) for r in refs), 0, 0)
        with pytest.raises(B.BBDukError):
            d.gpu.finalize_table()                                  # a build is in progress
        for first, group in ((1, refs[:2]), (3, refs[2:4]), (5, refs[4:])):
            cat = np.frombuffer(b"".join(group), np.uint8)
            offs = np.concatenate([[0], np.cumsum([len(r) for r in group])]).astype(np.int64)
            d.gpu.build_add_device(torch.from_numpy(cat.copy()).cuda(), offs, first)
        with pytest.raises(B.BBDukError):
            d.gpu.build_add_device(torch.zeros(40, dtype=torch.uint8).cuda(), np.array([0, 40], np.int64), 6)     # id beyond numScaffolds-1
        d.gpu.build_end()
        d.stored_kmers = d.gpu.table_size
    assert d.gpu.table_size == o.stored_kmers > (1 << 22)
    prng = random.Random(9)
    reads = []
    for i in range(20000):
        if i % 2:
            r = refs[prng.choice([0, 1, 4])]
            a = prng.randrange(len(r) - 150); s = r[a:a + 150]
            reads.append(util.revcomp(s) if prng.random() < 0.5 else s)
        else:
            reads.append(util.rand_seq(prng, 150, 0.002))
    check_batch(d, o, reads, True, nthreads=8)
    assert d.gpu.counters()[4] > 9000
    ks, vs = o.dump_pairs()
    sel = np.random.default_rng(1).choice(len(ks), 200000, replace=False)
    assert np.array_equal(d.gpu.table_lookup(ks[sel]), vs[sel])
    d.close()


def test_big_layout_secondary_operators():
    """ktrim=n, ktrim=rl, ksplit, k>31 and findbestmatch run the general scan functions: on the big layout their lookups take
    the exact per-key path."""
    rng = random.Random(11)
    refs, reads = util.long_refs_and_reads(rng)
    for args, okw in (("ktrim=n k=23 mink=11 hdist=1", dict(k=23, mink=11, hdist=1, ktrimN=1)), ("ktrim=rl k=23 mink=11", dict(k=23, mink=11, ktrimRight=1, ktrimLeft=1)),
                      ("k=40", dict(k=31, kbig=40, maskMiddle=0)), ("k=25 fbm=t", dict(k=25, findBestMatch=1)),
                      ("ksplit=t k=23 mink=11 hdist=1", dict(k=23, mink=11, hdist=1, ksplit=1))):
        o = Oracle(**okw)
        d = B.BBDuk(args, refs=refs, hooks=BIG)
        for r in refs:
            o.add_ref(r)
        assert d.gpu.table_size == o.stored_kmers
        rr = reads + reads[:1] * (len(reads) % 2)
        check_batch(d, o, rr, "ksplit" not in args)
        d.close()


def _is_big(d):
    try:
        d.gpu.line_histogram()
        return True
    except B.BBDukError:
        return False


@pytest.mark.parametrize("args,okw,genome,want_big", [
    ("k=31", dict(k=31), 1_150_000, True),                          # plain kfilter beyond 2^20 keys: minimizer lines without any hook
    ("k=31 hdist=1", dict(k=31, hdist=1), 14_000, False),           # 1.27 M keys of Hamming neighbourhoods: plain lines pay only beyond 2^21 keys
    ("k=31 hdist=1", dict(k=31, hdist=1), 27_000, True),            # 2.45 M of them: plain lines (+ the twin)
    ("k=31", dict(k=31), 900_000, False),                           # below the threshold: the cache-resident map
    ("k=21", dict(k=21), 1_150_000, True),                          # short k-mers too since bbduk_bigs_kernel (round 4: 320 against 202 Gbases/s)
    ("ktrim=r k=17 mink=9", dict(k=17, ktrimRight=1, mink=9), 1_150_000, True),
    ("k=31 mkf=0.1", dict(k=31, minKmerFraction=0.1), 1_150_000, True),       # round 5: the thresholds read the exact hit plane of bbduk_bigs_every_kernel
    ("k=31 mcf=0.3", dict(k=31, minCoveredFraction=0.3), 1_150_000, True),
    ("k=27 fbm=t", dict(k=27, findBestMatch=1), 1_150_000, True),             # ... and so do findbestmatch, ksplit, ktrim=n and ktrim=rl
    ("ksplit=t k=27 mink=12", dict(k=27, mink=12, ksplit=1), 1_150_000, True),
    ("ktrim=n k=27 mink=12", dict(k=27, mink=12, ktrimN=1), 1_150_000, True),
    ("ktrim=rl k=27 mink=12", dict(k=27, mink=12, ktrimRight=1, ktrimLeft=1), 1_150_000, True),
    ("k=31 restrictleft=100", dict(k=31, restrictLeft=100), 1_150_000, True),    # the GENERAL family's flags too (bbduk_bigs_general_kernel)
    ("ktrim=r k=27 mink=12 restrictright=80 qskip=2", dict(k=27, mink=12, ktrimRight=1, restrictRight=80, qSkip=2), 1_150_000, True),
    ("k=25 rcomp=f speed=3 skipr2=t", dict(k=25, rcomp=0, speed=3, skipR2=1), 1_150_000, True),
    ("k=31 qhdist=1", dict(k=31, qhdist=1), 1_150_000, False),                # query expansion: the tiled kernels over the cache-resident map
    ("k=40", dict(k=31, kbig=40, maskMiddle=0), 1_150_000, True),             # k > 31 (round 5, bbduk_bigs_kbig.hip): the run state machine over the exact hit plane
    ("ktrim=r k=27", dict(k=27, ktrimRight=1), 1_150_000, True),    # ktrim=r: minimizer lines too since round 4 (bbduk_bigs_kernel<KTRIM_R>)
    ("ktrim=l k=27", dict(k=27, ktrimLeft=1), 1_150_000, True),     # ktrim=l and kfilter with maxbadkmers > 0: bbduk_bigs_every_kernel (round 4)
    ("k=31 mbk=2", dict(k=31, maxBadKmers0=2), 1_150_000, True),
])
def test_layout_chosen_by_key_count(args, okw, genome, want_big):
    """The layouts as the library picks them by itself (round 3: the big layout from 2^20 keys on for the configurations that have its scan,
    half as many buckets for the others), each against the oracle: table, reads, counters."""
    g = np.frombuffer(b"ACGT", np.uint8)[np.random.default_rng(genome).integers(0, 4, genome)].tobytes()
    o = Oracle(**okw); o.add_ref(g)
    for build in ("host", "device"):
        d = B.BBDuk(args, refs=[g], build=build)
        assert d.gpu.table_size == o.stored_kmers and _is_big(d) == want_big
        ks, vs = o.dump_pairs()
        sel = np.random.default_rng(1).integers(0, len(ks), 200_000)
        probe = np.concatenate([ks[sel], np.random.default_rng(2).integers(0, 1 << 62, 50_000, dtype=np.int64)])
        got = d.gpu.table_lookup(probe)
        assert np.array_equal(got[:len(sel)], vs[sel]) and int((got[len(sel):] > 0).sum()) == 0
        b, off = B.synth_generate_host(B.synth_params(17, n_rate=0.002, contam=g[:200_000], contam_frac=0.05), 0, 20_000)
        pairings = (False,) if okw.get("ksplit") else (True, False)
        for paired in pairings:
            d.gpu.reset_counters(); o.reset_counters()
            ga, gi, gf = d.gpu.process_batch(b, off, paired)
            oa, oi, of = o.process_batch(b, off, paired, nthreads=8)
            assert np.array_equal(ga, oa) and np.array_equal(gi, oi) and np.array_equal(gf, of)
            assert np.array_equal(d.gpu.counters(), o.counters())
        assert int((oa > 0).sum()) > 100
        # units beyond a wave's planes (tile kernel) and beyond the tiles (long-read kernel): a big-layout map of this size answers them from its
        # cache-resident twin (build_both), the others from the one map they have
        rng = random.Random(genome)
        cut = lambda n: g[(p0 := rng.randrange(0, len(g) - n)):p0 + n]
        longs = [util.rand_seq(rng, 3000), util.rand_seq(rng, 1400) + cut(200) + util.rand_seq(rng, 1400), util.rand_seq(rng, 30000) + cut(64), util.rand_seq(rng, 150),
                 cut(5000), util.rand_seq(rng, 2600), util.rand_seq(rng, 64000), cut(40) + util.rand_seq(rng, 45000)]
        for reads in (longs[:2] + [util.rand_seq(rng, 150) for _ in range(600)], longs):
            bb, oo = pack_reads(reads)
            for paired in pairings:
                d.gpu.reset_counters(); o.reset_counters()
                ga, gi, gf = d.gpu.process_batch(bb, oo, paired)
                oa, oi, of = o.process_batch(bb, oo, paired, nthreads=4)
                assert np.array_equal(ga, oa) and np.array_equal(gi, oi) and np.array_equal(gf, of), (args, build, paired)
                assert np.array_equal(d.gpu.counters(), o.counters())
        d.close()


def test_big_layout_window_whose_first_fingerprint_match_is_an_impostor():
    """k=17 against 4.6 M keys: a fifth of the random reads hit the map by chance, mostly with ONE window.  The candidate scan reports a window's first
    fingerprint match; when that is another key (an impostor) the window's own key can still sit in a later way, in the alternate word or in the
    secondary map -- the verification has to ask for it before it moves on (before the fix: 25-37 of 2 M reads lost their only hit; found at the end of
    round 3 when the layout became the default from 2^20 keys on)."""
    G = 4_600_000
    g = np.frombuffer(b"ACGT", np.uint8)[np.random.default_rng(G).integers(0, 4, G)].tobytes()
    o = Oracle(k=17); o.add_ref(g)
    b, off = B.synth_generate_host(B.synth_params(3, contam=g, contam_frac=0.01), 0, 1_000_000)
    oa, oi, of = o.process_batch(b, off, True, nthreads=16)
    assert int((oa > 0).sum()) > 300_000
    for hooks in (BIG, None):
        d = B.BBDuk("k=17", refs=[g], build="device", hooks=hooks)
        ga, gi, gf = d.gpu.process_batch(b, off, True)
        bad = np.nonzero((ga != oa) | (gi != oi) | (gf != of))[0]
        assert len(bad) == 0, (hooks, len(bad), bad[:6])
        d.close()


@pytest.mark.parametrize("args,okw", [("ktrim=r k=31", dict(k=31, ktrimRight=1)),
                                      ("ktrim=r k=27 mink=12", dict(k=27, mink=12, ktrimRight=1)),
                                      ("ktrim=r k=25 mink=11 hdist=1", dict(k=25, mink=11, hdist=1, ktrimRight=1)),
                                      ("ktrim=r k=23 mm=f tpe", dict(k=23, maskMiddle=0, ktrimRight=1, trimPairsEvenly=1)),
                                      ("ktrim=r k=29 forbidn=t hdist=1 tp=2 minlen=30", dict(k=29, forbidN=1, hdist=1, ktrimRight=1, trimPad=2, minReadLength=30))])
def test_big_layout_ktrim_right(args, okw):
    """ktrim=r against a big-layout map (round 4: bbduk_bigs_kernel<KTRIM_R>): first hit by the stream scan over the minimizer lines, the short
    k-mers of mink out of the secondary map, units beyond a wave's planes through the tiled and the long-read kernels' exact lookups"""
    _planted_pieces_case(args, okw)


@pytest.mark.parametrize("args,okw", [("ktrim=l k=31", dict(k=31, ktrimLeft=1)),
                                      ("ktrim=l k=27 mink=12", dict(k=27, mink=12, ktrimLeft=1)),
                                      ("ktrim=l k=25 mink=11 hdist=1 tp=1", dict(k=25, mink=11, hdist=1, ktrimLeft=1, trimPad=1)),
                                      ("ktrim=l k=29 forbidn=t minlen=30", dict(k=29, forbidN=1, ktrimLeft=1, minReadLength=30)),
                                      ("k=31 mbk=1", dict(k=31, maxBadKmers0=1)),
                                      ("k=27 mbk=3 forbidn=t", dict(k=27, maxBadKmers0=3, forbidN=1)),
                                      ("k=25 mbk=2 hdist=1 mm=f", dict(k=25, maxBadKmers0=2, hdist=1, maskMiddle=0))])
def test_big_layout_every_hit_scans(args, okw):
    """ktrim=l and kfilter with maxbadkmers > 0 against a big-layout map (round 4: bbduk_bigs_every_kernel): every candidate of the stream scan
    is verified (lanes = candidates), the last hit / the hit count come out of the exact hit plane; same reads as the ktrim=r cases."""
    _planted_pieces_case(args, okw)


def _planted_pieces_case(args, okw):
    rng = random.Random(len(args))
    refs = [util.rand_seq(rng, 6000, 0.002), util.rand_seq(rng, 900), util.rand_seq(rng, 40), util.rand_seq(rng, 2500)]
    for build in ("device", "host"):
        d, o = big_pair(args, okw, refs=refs, build=build)
        assert d.gpu.table_bytes >= 64 * (64 + 256 + 64)
        ks, vs = o.dump_pairs()
        assert np.array_equal(d.gpu.table_lookup(ks), vs)          # the short k-mers too
        reads = util.fuzz_reads(rng, refs, 800, [0, 1, 10, 11, 12, 24, 25, 30, 31, 32, 75, 100, 150, 151, 250, 300], junk=True)
        for i in range(200):                                        # reads that run into a reference piece, cut at every distance from their end
            r = bytearray(util.rand_seq(rng, rng.choice([100, 150, 151])))
            src = refs[rng.choice([0, 0, 1, 3])]; a = rng.randrange(len(src) - 60); cut = rng.randint(5, 60)
            piece = bytearray(src[a:a + cut])
            if rng.random() < 0.3: piece[rng.randrange(len(piece))] = rng.choice(b"ACGTN")
            w = len(r) - cut if rng.random() < 0.6 else rng.randrange(len(r) - cut)
            r[w:w + cut] = piece
            reads.append(bytes(r))
        if len(reads) % 2: reads.pop()
        check_batch(d, o, reads, True)
        check_batch(d, o, reads[:401], False)
        long_reads = []
        for i in range(24):                                         # 4 kb pairs (tile kernel) and one 60 kb read (long-read kernel)
            n = 60000 if i == 7 else rng.choice([150, 4000])
            r = bytearray(util.rand_seq(rng, n))
            if rng.random() < 0.6:
                src = refs[0]; a = rng.randrange(len(src) - 80); w = rng.randrange(n - 80); r[w:w + 80] = src[a:a + 80]
            long_reads.append(bytes(r))
        # ... and in the same batch (the long-read kernel takes all of it) short reads whose only hit is an end's short k-mer, or the left end's
        # length-k lookup behind a junk symbol (found by the soak: the fallbacks' short scans asked the secondary map for a full-length key)
        mk = okw.get("mink", 0)
        for i in range(40):
            src = refs[rng.choice([0, 1, 3])]; a = rng.randrange(len(src) - 40)
            n = rng.choice([okw["k"], okw["k"] + 3, 60, 151])
            r = bytearray(util.rand_seq(rng, n))
            m = rng.randint(max(mk, 8), okw["k"])
            if okw.get("ktrimLeft"): r[:m] = src[a:a + m]
            else: r[n - m:] = src[a:a + m]
            if rng.random() < 0.5 and n > okw["k"]: r[rng.randrange(n)] = rng.choice(b".N")
            long_reads.append(bytes(r))
        check_batch(d, o, long_reads, True)
        d.close()


FAMILIES = [("ksplit=t k=23 mink=11 hdist=1", dict(k=23, mink=11, hdist=1, ksplit=1)),
            ("ksplit=t k=31", dict(k=31, ksplit=1)),
            ("ksplit=t k=27 mink=9 forbidn=t tp=-1", dict(k=27, mink=9, forbidN=1, ksplit=1, trimPad=-1)),
            ("ktrim=n k=23 mink=11 hdist=1", dict(k=23, mink=11, hdist=1, ktrimN=1)),
            ("ktrim=n k=31 tp=2", dict(k=31, ktrimN=1, trimPad=2)),
            ("ktrim=n k=25 mink=12 kmaskfullycovered=t", dict(k=25, mink=12, ktrimN=1, kmaskFullyCovered=1)),
            ("kmask=lc k=27 forbidn=t hdist=1 rieb=f minlen=40", dict(k=27, ktrimN=1, forbidN=1, hdist=1, requireBothBad=1, minReadLength=40)),
            ("ktrim=rl k=23 mink=11 hdist=1 tpe", dict(k=23, mink=11, hdist=1, ktrimLeft=1, ktrimRight=1, trimPairsEvenly=1)),
            ("ktrim=rl k=31", dict(k=31, ktrimLeft=1, ktrimRight=1)),
            ("ktrim=rl k=25 mink=10 forbidn=t hdist=1 tp=1", dict(k=25, mink=10, forbidN=1, hdist=1, ktrimLeft=1, ktrimRight=1, trimPad=1)),
            ("ktrim=rl k=24 mm=f ktrimexclusive=t", dict(k=24, maskMiddle=0, ktrimLeft=1, ktrimRight=1, ktrimExclusive=1)),
            ("k=25 fbm=t", dict(k=25, findBestMatch=1)),
            ("k=31 hdist=1 rename=t", dict(k=31, hdist=1, findBestMatch=1)),
            ("k=31 mkf=0.2", dict(k=31, minKmerFraction=0.2)),
            ("k=27 mkf=0.05 forbidn=t hdist=1", dict(k=27, minKmerFraction=0.05, forbidN=1, hdist=1)),
            ("k=31 mcf=0.5", dict(k=31, minCoveredFraction=0.5)),
            ("k=25 mcf=0.2 mm=f rieb=f", dict(k=25, minCoveredFraction=0.2, maskMiddle=0, requireBothBad=1)),
            ("k=31 mbk=7", dict(k=31, maxBadKmers0=7)),
            ("k=40", dict(k=31, kbig=40, maskMiddle=0)),                                    # k > 31 (bbduk_bigs_kbig.hip): runs of matching 31-mers over the exact hit plane
            ("k=33 mbk=2 hdist=1", dict(k=31, kbig=33, maskMiddle=0, maxBadKmers0=2, hdist=1)),
            ("k=55 mkf=0.1 forbidn=t", dict(k=31, kbig=55, maskMiddle=0, minKmerFraction=0.1, forbidN=1)),
            ("k=36 rieb=f minlen=40", dict(k=31, kbig=36, maskMiddle=0, requireBothBad=1, minReadLength=40))]


GENERALS = [("k=31 restrictright=70", dict(k=31, restrictRight=70)),
            ("k=27 restrictleft=60 restrictright=100 forbidn=t hdist=1", dict(k=27, restrictLeft=60, restrictRight=100, forbidN=1, hdist=1)),
            ("k=31 qskip=3 mbk=1", dict(k=31, qSkip=3, maxBadKmers0=1)),
            ("k=25 speed=5 mkf=0.1", dict(k=25, speed=5, minKmerFraction=0.1)),
            ("k=27 rcomp=f", dict(k=27, rcomp=0)),
            ("k=24 mm=f rcomp=f skipr2=t mcf=0.2", dict(k=24, maskMiddle=0, rcomp=0, skipR2=1, minCoveredFraction=0.2)),
            ("ktrim=r k=23 mink=11 hdist=1 restrictright=60", dict(k=23, mink=11, hdist=1, ktrimRight=1, restrictRight=60)),
            ("ktrim=r k=31 restrictright=20 qskip=2", dict(k=31, ktrimRight=1, restrictRight=20, qSkip=2)),          # a span shorter than k-1 bases
            ("ktrim=r k=25 mink=9 speed=4 rcomp=f tpe", dict(k=25, mink=9, ktrimRight=1, speed=4, rcomp=0, trimPairsEvenly=1)),
            ("ktrim=l k=27 mink=12 restrictleft=50 skipr1=t", dict(k=27, mink=12, ktrimLeft=1, restrictLeft=50, skipR1=1)),
            ("ktrim=l k=31 restrictright=90 qskip=2 forbidn=t", dict(k=31, ktrimLeft=1, restrictRight=90, qSkip=2, forbidN=1)),
            ("ksplit=t k=25 mink=10 restrictright=120 speed=2", dict(k=25, mink=10, ksplit=1, restrictRight=120, speed=2)),
            ("ksplit=t k=31 rcomp=f qskip=2", dict(k=31, ksplit=1, rcomp=0, qSkip=2)),
            ("ktrim=n k=27 mink=11 restrictleft=100 restrictright=100", dict(k=27, mink=11, ktrimN=1, restrictLeft=100, restrictRight=100)),
            ("ktrim=n k=31 kmaskfullycovered=t restrictright=80 qskip=2", dict(k=31, ktrimN=1, kmaskFullyCovered=1, restrictRight=80, qSkip=2)),
            ("kmask=lc k=25 speed=6 rcomp=f skipr1=t", dict(k=25, ktrimN=1, speed=6, rcomp=0, skipR1=1)),
            ("ktrimtips=40 k=23 mink=11 hdist=1", dict(k=23, mink=11, hdist=1, ktrimLeft=1, ktrimRight=1, restrictLeft=40, restrictRight=40)),
            ("ktrim=rl k=27 qskip=2 forbidn=t speed=2", dict(k=27, ktrimLeft=1, ktrimRight=1, qSkip=2, forbidN=1, speed=2)),
            ("ktrim=rl k=25 mink=12 rcomp=f skipr2=t", dict(k=25, mink=12, ktrimLeft=1, ktrimRight=1, rcomp=0, skipR2=1)),
            ("k=25 fbm=t restrictright=90 skipr1=t", dict(k=25, findBestMatch=1, restrictRight=90, skipR1=1)),
            ("k=27 fbm=t qskip=2 speed=3 rcomp=f", dict(k=27, findBestMatch=1, qSkip=2, speed=3, rcomp=0)),
            ("k=40 restrictright=100 skipr2=t", dict(k=31, kbig=40, maskMiddle=0, restrictRight=100, skipR2=1)),      # k > 31 with the flags it accepts (no speed, no qskip: BBDukParser reduces kbig there)
            ("k=34 rcomp=f restrictleft=120 mbk=1", dict(k=31, kbig=34, maskMiddle=0, rcomp=0, restrictLeft=120, maxBadKmers0=1))]


@pytest.mark.parametrize("ci", range(len(GENERALS)))
def test_big_layout_general_family(ci):
    """Round 5: the flags of the GENERAL family (restrictleft / restrictright, skipr1 / skipr2, qskip, speed, rcomp=f) in every operator over a
    big-layout map (bbduk_bigs_general_kernel): the plane of looked-up positions follows the span and qskip's grid, the windows that reach in
    front of a span start are looked up exactly by worker lanes, speed gates and rcomp=f picks the key in the exact lookups."""
    _families_case(*GENERALS[ci], seed=47000 + ci, builds=(("device", BIG), ("host", BIGWIDE)))


@pytest.mark.parametrize("ci", range(len(FAMILIES)))
def test_big_layout_operator_families(ci):
    """Round 5: ksplit, ktrim=n, ktrim=rl, findbestmatch and kfilter with mkf / mcf against a big-layout map (bbduk_bigs_every_kernel: the exact
    hit plane behind the stream scan over the minimizer lines; the mode's facts are read out of it).  Forced onto small references, so the map has
    no cache-resident twin and the units beyond a wave's planes run the big-layout instantiations of the tiled and the long-read kernels
    (bbduk_big_tiles.hip).  Every output the operator has is compared: mask bits, both trim amounts, split spans, match lists."""
    _families_case(*FAMILIES[ci], seed=31000 + ci, builds=(("device", BIG), ("host", BIG), ("device", BIGWIDE)))


def _families_case(args, okw, seed, builds):
    rng = random.Random(seed)
    refs = [util.rand_seq(rng, 6000, 0.002), util.rand_seq(rng, 900), util.rand_seq(rng, 40), util.rand_seq(rng, 2500)]
    refs.append(refs[0][100:180] + refs[1][-70:])               # a scaffold sharing k-mers with two others: first-writer-wins ids
    k = okw["k"]
    for build, hooks in builds:
        d, o = big_pair(args, okw, refs=refs, build=build, hooks=hooks)
        assert _is_big(d)
        ks, vs = o.dump_pairs()
        assert np.array_equal(d.gpu.table_lookup(ks), vs)
        reads = util.fuzz_reads(rng, refs, 700, [0, 1, 10, 11, 12, 24, 25, 30, 31, 32, 75, 100, 150, 151, 250, 300], junk=True)
        for i in range(300):                                    # reference pieces of every length at either end, in the middle, twice, with an N or junk nearby
            n = rng.choice([100, 150, 151, 250])
            r = bytearray(util.rand_seq(rng, n))
            for _ in range(rng.choice([1, 1, 2])):
                src = refs[rng.choice([0, 0, 1, 3, 4])]; cut = rng.randint(5, 90)
                a = rng.randrange(max(1, len(src) - cut)); piece = bytearray(src[a:a + cut])
                if rng.random() < 0.5: piece = bytearray(util.revcomp(bytes(piece)))
                if rng.random() < 0.3: piece[rng.randrange(len(piece))] = rng.choice(b"ACGTN.")
                cut = len(piece)
                w = rng.choice([0, n - cut, rng.randrange(n - cut + 1), n // 2 - cut // 2])
                r[w:w + cut] = piece
            if rng.random() < 0.2: r[rng.randrange(n)] = ord("N")
            reads.append(bytes(r))
        if len(reads) % 2: reads.pop()
        longs = []
        for i in range(20):                                     # 4 kb units (tiled kernel) and one 60 kb read (long-read kernel), beside short ones
            n = 60000 if i == 7 else rng.choice([150, 4000])
            r = bytearray(util.rand_seq(rng, n))
            if rng.random() < 0.7:
                src = refs[0]; a = rng.randrange(len(src) - 80); w = rng.choice([0, n - 80, rng.randrange(n - 80)]); r[w:w + 80] = src[a:a + 80]
            longs.append(bytes(r))
        if okw.get("ksplit"):
            check_split(d, o, reads); check_split(d, o, longs); check_split(d, o, [])
        elif okw.get("ktrimN"):
            for rr, paired in ((reads, True), (reads[:401], False), (longs, True)):
                check_kmask(d, o, rr, paired)
        elif okw.get("ktrimLeft") and okw.get("ktrimRight"):
            for rr, paired in ((reads, True), (reads[:401], False), (longs, True)):
                check_tips(d, o, rr, paired)
        else:
            check_batch(d, o, reads, True); check_batch(d, o, reads[:401], False); check_batch(d, o, longs, True)
        d.close()


@pytest.mark.parametrize("args,okw,hdist", [("k=31", dict(k=31), 0), ("ktrim=r k=27 mink=12", dict(k=27, mink=12, ktrimRight=1), 0),
                                            ("ksplit=t k=25 mink=11", dict(k=25, mink=11, ksplit=1), 0), ("k=31 hdist=1", dict(k=31, hdist=1), 1)])
def test_streamed_build_keeps_a_twin(args, okw, hdist):
    """Round 5: a streamed device build (bbduk_build_begin / _add_device / _end) of a map between 2^20 and 2^25 keys takes the big
    (or the seed) layout like the other builders and builds the cache-resident twin alongside -- every piece goes into both sinks while it is in HBM --
    so that units beyond a wave's planes run the faster fallbacks.  Table, short reads and 3 kb - 64 kb units against the oracle."""
    import torch
    G = 1_150_000 if hdist == 0 else 60_000
    rng0 = np.random.default_rng(G + hdist)
    refs = [np.frombuffer(b"ACGT", np.uint8)[rng0.integers(0, 4, n)].tobytes() for n in (G // 2, 40, G // 2)]
    o = Oracle(**okw)
    for r in refs: o.add_ref(r)
    d = B.BBDuk.__new__(B.BBDuk)
    d.host = B.HostIndex(args)
    for r in refs: d.host.add_ref(r)
    d.gpu = B.BBDukGpu(d.host.params(0))
    d.gpu.build_begin(sum(len(r) for r in refs) * (1 + 3 * okw["k"] * hdist) + 100000, hdist, hdist)
    for first, group in ((1, refs[:2]), (3, refs[2:])):
        cat = np.frombuffer(b"".join(group), np.uint8)
        offs = np.concatenate([[0], np.cumsum([len(r) for r in group])]).astype(np.int64)
        d.gpu.build_add_device(torch.from_numpy(cat.copy()).cuda(), offs, first)
    d.gpu.build_end()
    d.stored_kmers = d.gpu.table_size
    assert d.gpu.table_size == o.stored_kmers
    if hdist == 0: assert _is_big(d)
    big_alone = (64 + 256 + 64) * (d.gpu.table_size / (32 * 0.30))
    assert d.gpu.table_bytes > 8 * d.gpu.table_size                  # (more than one layout's worth: the twin is there)
    ks, vs = o.dump_pairs()
    sel = np.random.default_rng(1).integers(0, len(ks), 100_000)
    assert np.array_equal(d.gpu.table_lookup(ks[sel]), vs[sel])
    g = refs[0]
    rng = random.Random(G)
    cut = lambda n: g[(p0 := rng.randrange(0, len(g) - n)):p0 + n]
    reads = [util.rand_seq(rng, 150, 0.002) if i % 3 else cut(150) for i in range(4000)]
    longs = [util.rand_seq(rng, 3000), util.rand_seq(rng, 1400) + cut(200) + util.rand_seq(rng, 1400), util.rand_seq(rng, 30000) + cut(64), util.rand_seq(rng, 150),
             cut(5000), util.rand_seq(rng, 2600), util.rand_seq(rng, 64000), cut(40) + util.rand_seq(rng, 45000)]
    paired = not okw.get("ksplit")
    check_batch(d, o, reads, paired, nthreads=8)
    check_batch(d, o, longs[:2] + reads[:600], paired, nthreads=8)
    check_batch(d, o, longs, paired, nthreads=8)
    d.close()


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_under_announced_builds_end(tmp_path):
    """a caller that announces fewer keys to bbduk_build_begin than it then sends.  The scratch set of a cache-resident build stops at 7/8 full and the
    build fails with BBDUK_ERR_NOMEM (it used to probe a full set for ever); the twin of a streamed big-layout build that overflows is dropped and the map that is
    complete keeps serving -- table and reads against the oracle."""
    import torch
    rng0 = np.random.default_rng(5)
    # (1) cache-resident: 1 000 keys announced, 200 000 sent
    ref = np.frombuffer(b"ACGT", np.uint8)[rng0.integers(0, 4, 200_030)].tobytes()
    d = B.BBDuk.__new__(B.BBDuk)
    d.host = B.HostIndex("k=31 hdist=0")
    d.host.add_ref(ref)
    d.gpu = B.BBDukGpu(d.host.params(0))
    d.gpu.build_begin(1000, 0, 0)
    with pytest.raises(B.BBDukError):
        d.gpu.build_add_device(torch.from_numpy(np.frombuffer(ref, np.uint8).copy()).cuda(), np.array([0, len(ref)], np.int64), 1)
        d.gpu.build_end()
    d.gpu.close()
    # (2) a streamed big-layout build with a twin: 1.2 M keys announced (lines at 0.30 keys per slot: room for 4 M), 3.2 M sent -- the twin's scratch set
    #     (2^22 slots) passes 7/8 full and is dropped; the lines are complete
    G = 3_200_000
    ref = np.frombuffer(b"ACGT", np.uint8)[rng0.integers(0, 4, G + 30)].tobytes()
    o = Oracle(k=31, hdist=0)
    o.add_ref(ref)
    d = B.BBDuk.__new__(B.BBDuk)
    d.host = B.HostIndex("k=31 hdist=0")
    d.host.add_ref(ref)
    d.gpu = B.BBDukGpu(d.host.params(0))
    d.gpu.build_begin(1_200_000, 0, 0)
    try:
        d.gpu.build_add_device(torch.from_numpy(np.frombuffer(ref, np.uint8).copy()).cuda(), np.array([0, len(ref)], np.int64), 1)
        d.gpu.build_end()
    except B.BBDukError:
        d.gpu.close()
        return                                                   # (the lines overflowed first: a loud failure is an answer too)
    d.stored_kmers = d.gpu.table_size
    assert d.gpu.table_size == o.stored_kmers and _is_big(d)
    ks, vs = o.dump_pairs()
    sel = np.random.default_rng(1).integers(0, len(ks), 50_000)
    assert np.array_equal(d.gpu.table_lookup(ks[sel]), vs[sel])
    rng = random.Random(7)
    reads = [ref[(p0 := rng.randrange(0, G - 150)):p0 + 150] if rng.random() < 0.3 else util.rand_seq(rng, 150, 0.002) for _ in range(400)]
    reads += [util.rand_seq(rng, 3000) + ref[1000:1400] + util.rand_seq(rng, 2000), ref[5000:5000 + 70_000]]      # units beyond a wave's planes: without a twin, the big layout's tiled kernels
    if len(reads) % 2: reads.append(b"")
    check_batch(d, o, reads, True)
    d.close()
