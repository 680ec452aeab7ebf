"""The seed layout (bbtools_amd/csrc/bbduk_seed.inc): large reference-side hdist=1 maps stored as PARENTS under their halves instead of
1 + 3k neighbours per window.  Forced onto small references with bbduk_test_hook so that the oracle -- which enumerates the mutants as
BBDukIndexMod.mutate does -- can check every key, every id and every per-read decision, on the stream kernel, the tiled kernel and the
long-read kernel."""
import random

import numpy as np
import pytest

from bbtools_amd import bbduk as B
from oracle.oracle_ffi import Oracle, pack_reads
from tests import util
from tests.test_gpu_parity import check_batch

pytestmark = pytest.mark.gpu
SEED = {B.HOOK_SEED_LAYOUT: 1}
CFGS = [("k=31 hdist=1", dict(k=31, hdist=1)), ("k=25 hdist=1", dict(k=25, hdist=1)), ("k=31 hdist=1 mm=f", dict(k=31, hdist=1, maskMiddle=0)),
        ("k=24 hdist=1 mm=f rieb=f", dict(k=24, hdist=1, maskMiddle=0, requireBothBad=1)), ("k=27 hdist=1 minlen=40", dict(k=27, hdist=1, minReadLength=40)),
        # round 4: maxbadkmers > 0 on the seed layout (the per-read walk counts the exact hits)
        ("k=25 hdist=1 mbk=1", dict(k=25, hdist=1, maxBadKmers0=1)), ("k=31 hdist=1 mbk=3 rieb=f", dict(k=31, hdist=1, maxBadKmers0=3, requireBothBad=1)),
        # ... and forbidn=t (an undefined base resets rkmer: seed_window)
        ("k=27 hdist=1 forbidn=t", dict(k=27, hdist=1, forbidN=1)), ("k=25 hdist=1 forbidn=t mbk=2 mm=f", dict(k=25, hdist=1, forbidN=1, maxBadKmers0=2, maskMiddle=0))]


def seed_pair(args, okw, refs, expect_seed=True):
    o = Oracle(**okw)
    for r in refs:
        o.add_ref(r)
    d = B.BBDuk(args, refs=refs, hooks=SEED, build="device")
    assert d.gpu.table_bytes > 0
    # (a forced seed-layout map has no twin: bbduk_table_size then counts RECORDS -- four per reference window -- not the reference's keys)
    assert (d.gpu.table_size != o.stored_kmers) == expect_seed, (args, d.gpu.table_size, o.stored_kmers)
    return d, o


@pytest.mark.parametrize("ci", range(len(CFGS)))
def test_seed_layout_keys_and_ids(ci):
    args, okw = CFGS[ci]
    rng = random.Random(900 + ci)
    # scaffolds that share pieces (the smallest id has to win), reverse-complemented pieces, one with undefined bases, one shorter than k
    base = util.rand_seq(rng, 400)
    refs = [base, util.rand_seq(rng, 300) + base[100:180], util.revcomp(base[200:320]),
            util.rand_seq(rng, 150, 0.03), util.rand_seq(rng, 12), util.rand_seq(rng, 500)]
    d, o = seed_pair(args, okw, refs)
    ks, vs = o.dump_pairs()
    rs = np.random.default_rng(ci)
    near = ks[rs.integers(0, len(ks), 20000)] ^ (np.int64(1) << rs.integers(0, 2 * okw["k"], 20000).astype(np.int64))      # one bit off a key
    probe = np.concatenate([ks, near, rs.integers(0, 1 << (2 * okw["k"]), 20000, dtype=np.int64) | (np.int64(1) << (2 * okw["k"])), np.array([-1, 0], np.int64)])
    exp = np.array([o.table_get(int(x)) for x in probe], np.int32)
    got = d.gpu.table_lookup(probe)
    bad = np.nonzero(got != exp)[0]
    assert len(bad) == 0, (args, len(bad), [(hex(int(probe[i])), int(got[i]), int(exp[i])) for i in bad[:5]])
    d.close()


@pytest.mark.parametrize("ci", range(len(CFGS)))
def test_seed_layout_read_parity(ci):
    args, okw = CFGS[ci]
    for seed in range(2):
        rng = random.Random(7100 + 10 * ci + seed)
        refs = [util.rand_seq(rng, rng.randint(20, 400), 0.01) for _ in range(8)]
        d, o = seed_pair(args, okw, refs)
        reads = util.fuzz_reads(rng, refs, 600, [0, 1, 2, 23, 30, 31, 32, 40, 63, 64, 65, 75, 150, 151, 300], junk=True)
        # reads that carry reference pieces with ONE substitution (the neighbours the reference stores), and with two (which it does not)
        for _ in range(150):
            r = bytearray(util.rand_seq(rng, rng.choice([100, 150, 151, 250])))
            src = refs[rng.randrange(len(refs))]
            if len(src) >= 45:
                a = rng.randrange(len(src) - 44); piece = bytearray(src[a:a + 45])
                for _m in range(rng.choice([0, 1, 1, 2])):
                    j = rng.randrange(len(piece)); piece[j] = rng.choice(b"ACGT")
                if rng.random() < 0.5:
                    piece = bytearray(bytes(piece)[::-1].translate(bytes.maketrans(b"ACGT", b"TGCA")))
                w = rng.randrange(len(r) - 45); r[w:w + 45] = piece
            if rng.random() < 0.2:
                r[rng.randrange(len(r))] = ord("N")
            reads.append(bytes(r))
        if len(reads) % 2:
            reads.pop()
        check_batch(d, o, reads, True)
        check_batch(d, o, reads[:301], False)
        d.close()


def test_seed_layout_is_not_chosen_where_it_is_not_served():
    """hdist = 2, an off-centre mask: the hook only asks; such handles keep the mutants (cache-resident or big layout) and stay exact"""
    rng = random.Random(8)
    refs = [util.rand_seq(rng, 200, 0.01) for _ in range(4)]
    for args, okw in (("k=21 hdist=2", dict(k=21, hdist=2)),
                      # even k with an explicit one-base mask: the mask is off the mirror centre, seed_check is not exact there -> mutants kept
                      ("k=24 hdist=1 mm=1", dict(k=24, hdist=1, midMaskLen=1)), ("k=30 hdist=1 mm=1", dict(k=30, hdist=1, midMaskLen=1))):
        d, o = seed_pair(args, okw, refs, expect_seed=False)
        assert d.gpu.table_size == o.stored_kmers                  # the reference's keys, not records
        reads = util.fuzz_reads(rng, refs, 200, [0, 30, 31, 75, 150, 151], junk=True)
        check_batch(d, o, reads, True)
        d.close()


def test_seed_layout_long_units():
    """units beyond a wave's planes: the tiled kernel (5 kb pair) and the long-read kernel (45 kb read) look the seed layout up through lookup4"""
    for args, okw in (CFGS[0], CFGS[5]):
        rng = random.Random(31)
        refs = [util.rand_seq(rng, 300) for _ in range(4)]
        d, o = seed_pair(args, okw, refs)
        for big in (5000, 45000):
            reads = []
            for i in range(40):
                n = big if i in (4, 5, 22) else rng.choice([150, 151, 400])
                r = bytearray(util.rand_seq(rng, n))
                if rng.random() < 0.5:
                    src = refs[rng.randrange(4)]; a = rng.randrange(len(src) - 50); piece = bytearray(src[a:a + 50]); piece[rng.randrange(50)] = rng.choice(b"ACGT")
                    w = rng.randrange(n - 50); r[w:w + 50] = piece
                reads.append(bytes(r))
            check_batch(d, o, reads, True)
        d.close()


def test_seed_layout_chosen_from_2e22_keys_with_its_twin():
    """Round 4: a device build of k=31 hdist=1 beyond 2^22 keys (a 52 kbase reference: 4.7 M keys) takes the seed layout by itself -- joint halves,
    one gather per read position -- and keeps the cache-resident twin for units beyond a wave's planes; bbduk_table_size answers with the reference's
    key count.  Against the oracle: sampled keys, 2x150 reads with an N rate that puts undefined bases into most sub-tiles, long reads."""
    G = 52_000
    g = np.frombuffer(b"ACGT", np.uint8)[np.random.default_rng(G).integers(0, 4, G)].tobytes()
    o = Oracle(k=31, hdist=1); o.add_ref(g)
    d = B.BBDuk("k=31 hdist=1", refs=[g], build="device")
    assert d.gpu.table_size == o.stored_kmers
    assert d.gpu.table_bytes < 80 * o.stored_kmers                 # parents + the twin (counted since round 5: ~65 bytes per key), not 94 keys per window in a big-layout map BESIDE a twin (~105)
    ks, vs = o.dump_pairs()
    sel = np.random.default_rng(1).integers(0, len(ks), 100_000)
    probe = np.concatenate([ks[sel], np.random.default_rng(2).integers(0, 1 << 62, 20_000, dtype=np.int64) | (np.int64(1) << 62)])
    got = d.gpu.table_lookup(probe)
    assert np.array_equal(got[:len(sel)], vs[sel]) and int((got[len(sel):] > 0).sum()) == 0
    b, off = B.synth_generate_host(B.synth_params(23, n_rate=0.003, contam=g, contam_frac=0.08), 0, 20_000)
    for paired in (True, False):
        d.gpu.reset_counters(); o.reset_counters()
        ga, gi, gf = d.gpu.process_batch(b, off, paired)
        oa, oi, of = o.process_batch(b, off, paired, nthreads=8)
        bad = np.nonzero((ga != oa) | (gi != oi) | (gf != of))[0]
        assert len(bad) == 0, (paired, bad[:5], ga[bad[:5]], oa[bad[:5]])
        assert np.array_equal(d.gpu.counters(), o.counters())
    assert int((oa > 0).sum()) > 500
    rng = random.Random(5)
    cut = lambda n: g[(p0 := rng.randrange(0, len(g) - n)):p0 + n]
    longs = [util.rand_seq(rng, 3000), util.rand_seq(rng, 1400) + cut(200) + util.rand_seq(rng, 1400), util.rand_seq(rng, 30000) + cut(64), util.rand_seq(rng, 150),
             cut(5000), util.rand_seq(rng, 2600), util.rand_seq(rng, 64000), cut(40) + util.rand_seq(rng, 45000)]
    check_batch(d, o, longs, True)
    d.close()


@pytest.mark.parametrize("args,okw", [("k=31 hdist=1", dict(k=31, hdist=1)), ("k=25 hdist=1 mbk=1", dict(k=25, hdist=1, maxBadKmers0=1)),
                                      ("k=24 hdist=1 mm=f", dict(k=24, hdist=1, maskMiddle=0)), ("k=27 hdist=1 forbidn=t", dict(k=27, hdist=1, forbidN=1))])
def test_uploaded_hdist1_map_collapses_to_the_seed_layout(args, okw):
    """Round 5: the JVM builds the table (north_star), so an hdist=1 map arrives as (key, id) PAIRS -- 1 + 3k keys per window.
    bbduk_finalize_table finds the parents (windows whose whole neighbourhood is in the map, id = the largest id in it), builds the seed layout
    from them and checks every uploaded key against it.  Scaffolds that share pieces (the smallest id has to win), reverse-complemented pieces, a
    palindromic stretch, a scaffold with undefined bases; forced onto a small reference with the hook."""
    rng = random.Random(len(args))
    base = util.rand_seq(rng, 400)
    pal = util.rand_seq(rng, 40); pal = pal + util.revcomp(pal)
    refs = [util.rand_seq(rng, 300) + base[100:180], base, util.revcomp(base[200:320]), util.rand_seq(rng, 150, 0.03), util.rand_seq(rng, 12), pal + util.rand_seq(rng, 200), b"A" * 60 + util.rand_seq(rng, 100)]
    o = Oracle(**okw)
    for r in refs: o.add_ref(r)
    d = B.BBDuk(args, refs=refs, hooks=SEED, build="host")             # host index -> bbduk_upload_pairs -> bbduk_finalize_table
    assert d.gpu.table_layout == 2 + 4, d.gpu.table_layout              # seed, and the cache-resident map of the pairs as its twin
    assert d.gpu.table_size == o.stored_kmers
    ks, vs = o.dump_pairs()
    rs = np.random.default_rng(1)
    near = ks[rs.integers(0, len(ks), 20000)] ^ (np.int64(1) << rs.integers(0, 2 * okw["k"], 20000).astype(np.int64))
    probe = np.concatenate([ks, near, rs.integers(0, 1 << (2 * okw["k"]), 20000, dtype=np.int64) | (np.int64(1) << (2 * okw["k"]))])
    exp = np.array([o.table_get(int(x)) for x in probe], np.int32)
    got = d.gpu.table_lookup(probe)
    bad = np.nonzero(got != exp)[0]
    assert len(bad) == 0, (args, len(bad), [(hex(int(probe[i])), int(got[i]), int(exp[i])) for i in bad[:5]])
    reads = util.fuzz_reads(rng, refs, 1500, [0, 1, 30, 31, 32, 75, 100, 150, 151, 250, 300], junk=True)
    for i in range(300):
        r = bytearray(util.rand_seq(rng, 150)); src = refs[rng.choice([0, 1, 5, 6])]; a = rng.randrange(len(src) - 50); piece = bytearray(src[a:a + 50])
        for _ in range(rng.choice([0, 1, 1, 2])): piece[rng.randrange(50)] = rng.choice(b"ACGTN")
        w = rng.randrange(100); r[w:w + 50] = piece; reads.append(bytes(r))
    check_batch(d, o, reads, True)
    check_batch(d, o, reads[:401], False)
    check_batch(d, o, [util.rand_seq(rng, 5000), util.rand_seq(rng, 2000) + refs[1][50:150] + util.rand_seq(rng, 2000)] + reads[:100], True)      # the tiled kernel, over the twin
    d.close()


def test_uploaded_map_that_is_no_hdist1_neighbourhood_keeps_its_layout():
    """... and a map that is NOT a union of full 1-neighbourhoods (hdist=0 keys beside hdist=1 ones; an hdist=1 map with a key removed) stays what it was."""
    rng = random.Random(77)
    refs = [util.rand_seq(rng, 500), util.rand_seq(rng, 300)]
    o1 = Oracle(k=31, hdist=1); o1.add_ref(refs[0])
    o0 = Oracle(k=31); o0.add_ref(refs[1])
    k1, v1 = o1.dump_pairs(); k0, v0 = o0.dump_pairs()
    for keys, vals in ((np.concatenate([k1, k0]), np.concatenate([v1, v0 + 1])), (k1[1:], v1[1:])):
        keys, idx = np.unique(keys, return_index=True); vals = vals[idx]
        host = B.HostIndex("k=31 hdist=1"); host.add_ref(refs[0]); host.add_ref(refs[1])
        g = B.BBDukGpu(host.params(0)); g.test_hook(B.HOOK_SEED_LAYOUT, 1)
        g.upload_pairs(keys, vals.astype(np.int32)); g.finalize_table()
        assert g.table_layout in (0, 2 + 4)
        got = g.table_lookup(keys)
        assert np.array_equal(got, vals), int((got != vals).sum())
        if len(keys) == len(k1) - 1: assert g.table_layout == 0        # the hole cannot be covered by any parent
        g.close(); host.close()
