"""GPU parity tests (run on the MI355X box with -m gpu).  Every check goes through the C ABI
(libbbduk_hip.so) and compares bit-for-bit with the CPU oracle on the same seeded inputs, with the
committed golden fixtures, and -- at full workload sizes -- through size-independent properties."""
import random

import numpy as np
import pytest

from bbtools_amd import bbduk as B
from oracle.oracle_ffi import Oracle, pack_reads
from tests import util

pytestmark = pytest.mark.gpu


def make_pair(args, okw, refs=None, ref_fasta=None, hooks=None):
    if refs is not None:
        d = B.BBDuk(args, refs=refs, hooks=hooks)
        o = Oracle(**okw)
        for r in refs:
            o.add_ref(r)
    else:
        d = B.BBDuk(args)
        o = Oracle(**okw)
        o.load_fasta(ref_fasta)
    assert d.stored_kmers == o.stored_kmers == d.gpu.table_size
    return d, o


def check_batch(d, o, reads, paired, nthreads=4):
    b, off = pack_reads(reads)
    d.gpu.reset_counters(); o.reset_counters()
    ga, gi, gf = d.gpu.process_batch(b, off, paired)
    oa, oi, of = o.process_batch(b, off, paired, nthreads=nthreads)
    bad = np.nonzero((ga != oa) | (gi != oi) | (gf != of))[0]
    assert len(bad) == 0, [(int(i), reads[i], (int(ga[i]), int(gi[i]), int(gf[i])), (int(oa[i]), int(oi[i]), int(of[i]))) for i in bad[:3]]
    assert np.array_equal(d.gpu.counters(), o.counters())
    if d.gpu.params.mode == B.MODE_KFILTER and d.gpu.params.findBestMatch:
        # the per-read match lists rename=t prints (bbduk_kfilter_batch_matches), whole and truncated to three entries
        for cap in (64, 3):
            d.gpu.reset_counters(); o.reset_counters()
            ma, mi_, mf, nids, mids, mcnt = d.gpu.kfilter_batch_matches(b, off, paired, cap)
            oa2, oi2, of2, onids, omids, omcnt = o.process_batch_matches(b, off, paired, cap, nthreads=nthreads)
            assert np.array_equal(ma, oa) and np.array_equal(mi_, oi) and np.array_equal(mf, of)
            bad = np.nonzero((nids != onids) | (mids != omids).any(axis=1) | (mcnt != omcnt).any(axis=1))[0]
            assert len(bad) == 0, ("matches", cap, [(int(i), reads[i], int(nids[i]), mids[i].tolist(), mcnt[i].tolist(), int(onids[i]), omids[i].tolist(), omcnt[i].tolist()) for i in bad[:2]])
            assert np.array_equal(d.gpu.counters(), o.counters())
    if d.gpu.params.mode in (B.MODE_KTRIM_R, B.MODE_KTRIM_L, B.MODE_KFILTER):
        # the packed boundary format (2-bit codes + undefined bits) must give the same answers as the ASCII one
        codes, undef = B.pack_bases_host(b)
        d.gpu.reset_counters()
        pa, pi, pf = d.gpu.process_batch_packed(codes, undef, off, paired)
        bad = np.nonzero((pa != oa) | (pi != oi) | (pf != of))[0]
        assert len(bad) == 0, ("packed", [(int(i), reads[i], (int(pa[i]), int(pi[i]), int(pf[i])), (int(oa[i]), int(oi[i]), int(of[i]))) for i in bad[:3]])
        assert np.array_equal(d.gpu.counters(), o.counters())


# ---- golden fixtures
@pytest.mark.parametrize("name", util.golden_names())
def test_golden(name):
    fx = util.load_golden(name)
    if "fasta" in fx["ref"]:
        d = B.BBDuk(fx["cli"])
    else:
        d = B.BBDuk(fx["cli"], refs=[r.encode() for r in fx["ref"]["literal"]])
    assert d.stored_kmers == fx["stored_kmers"]
    b, off = pack_reads([r.encode() for r in fx["reads"]])
    a, ids, fl = d.gpu.process_batch(b, off, fx["paired"])
    got = [list(t) for t in zip(a.tolist(), ids.tolist(), fl.tolist())]
    assert got == fx["expected"]
    assert d.gpu.counters().tolist() == fx["counters"]


# ---- table
def test_table_lookup_and_way_images():
    args, okw, ref = util.CONFIGS["c2"]
    o = Oracle(**okw); o.load_fasta(ref)
    h = B.HostIndex(args); h.load_refs(); h.build_index()
    # the device map built from the HashArray1D images a JVM would hand over == built from the flat pairs
    g1 = B.BBDukGpu(h.params(0))
    for w, (prime, keys, vals, vk, vv) in enumerate(o.way_images()):
        g1.upload_table_way(w, prime, keys, vals, vk, vv)
    g1.finalize_table()
    ks, vs = o.dump_pairs()
    assert g1.table_size == len(ks)
    rng = np.random.default_rng(3)
    probe = np.concatenate([ks, rng.integers(0, 1 << 47, 50000, dtype=np.int64), np.array([-1, 0, 1 << 62], np.int64)])
    got = g1.table_lookup(probe)
    exp = np.array([o.table_get(int(x)) for x in probe], np.int32)
    assert np.array_equal(got, exp)
    with pytest.raises(B.BBDukError):
        g1.finalize_table()                      # read-only after finalize


def test_way_images_with_victims():
    """SURVEY a9 on the boundary: HashArray1D images whose victim lists are NOT empty (the restatement walks a probe window of 2 cells
    instead of 60, see tests/test_oracle.py) go through bbduk_upload_table_way; the device map has to hold every key of the arrays and
    of the victim lists with its id (HashForest.getValue semantics, kmer/HashForest.java:229-233), and a batch through it has to give
    what the plain oracle gives."""
    from oracle import oracle_ffi as O
    args, okw, ref = util.CONFIGS["c2"]
    plain = Oracle(**okw); plain.load_fasta(ref)
    h = B.HostIndex(args); h.load_refs(); h.build_index()
    O.set_probe_window(2)
    try:
        o = Oracle(**okw); o.load_fasta(ref)
        images = o.way_images()
        assert sum(len(w[3]) for w in images) > 1000
        g = B.BBDukGpu(h.params(0))
        for w, (prime, keys, vals, vk, vv) in enumerate(images):
            g.upload_table_way(w, prime, keys, vals, vk, vv)
        g.finalize_table()
        ks, vs = o.dump_pairs()
        assert g.table_size == len(ks) == plain.stored_kmers
        vk = np.concatenate([w[3] for w in images]); vv = np.concatenate([w[4] for w in images])
        assert np.array_equal(g.table_lookup(vk), vv)                    # the victims, each with its id
        rng = np.random.default_rng(5)
        probe = np.concatenate([ks, rng.integers(0, 1 << 47, 20000, dtype=np.int64)])
        assert np.array_equal(g.table_lookup(probe), np.array([o.table_get(int(x)) for x in probe], np.int32))
    finally:
        O.set_probe_window(60)
    b, off = B.synth_generate_host(B.synth_params(4), 0, 10000)
    ga, gi, gf = g.process_batch(b, off, True)
    oa, oi, of = plain.process_batch(b, off, True, nthreads=4)
    assert np.array_equal(ga, oa) and np.array_equal(gi, oi) and np.array_equal(gf, of)
    assert np.array_equal(g.counters(), plain.counters())


# ---- seeded synthetic inputs, the BASELINE configs at oracle-sized n
@pytest.mark.parametrize("cfg", list(util.CONFIGS))
def test_synthetic_config_parity(cfg):
    args, okw, ref = util.CONFIGS[cfg]
    d, o = make_pair(args, okw, ref_fasta=ref)
    kw = {}
    if "phix" in args:
        kw = dict(contam=B.read_fasta(util.PHIX)[0][1], contam_frac=0.05)
    sp = B.synth_params(11, **kw)
    b, off = B.synth_generate_host(sp, 1000, 20000)          # 40k reads
    d.gpu.reset_counters(); o.reset_counters()
    ga, gi, gf = d.gpu.process_batch(b, off, True)
    oa, oi, of = o.process_batch(b, off, True, nthreads=8)
    assert np.array_equal(ga, oa) and np.array_equal(gi, oi) and np.array_equal(gf, of)
    assert np.array_equal(d.gpu.counters(), o.counters())
    assert int((ga > 0).sum()) > 100                          # the workload really exercises hits


# ---- fuzz over parameter space, ragged / empty / junk reads
FUZZ = [("ktrim=r k=23", dict(k=23, ktrimRight=1)),
        ("ktrim=r k=23 mink=11 hdist=1", dict(k=23, ktrimRight=1, mink=11, hdist=1)),
        ("k=31 hdist=1", dict(k=31, hdist=1)), ("k=31", dict(k=31)),
        ("ktrim=l k=13 mink=6 hdist=1 hdist2=0", dict(k=13, ktrimLeft=1, mink=6, hdist=1, hdist2=0)),
        ("k=12 mm=f qhdist=1", dict(k=12, maskMiddle=0, qhdist=1)),
        ("ktrim=r k=11 mink=5 qhdist=1 qhdist2=1 rcomp=f", dict(k=11, ktrimRight=1, mink=5, qhdist=1, qhdist2=1, rcomp=0)),
        ("ktrim=r k=15 restrictright=40 forbidn=t hdist=1", dict(k=15, ktrimRight=1, restrictRight=40, forbidN=1, hdist=1)),
        ("ktrim=l k=15 restrictleft=30 tp=2", dict(k=15, ktrimLeft=1, restrictLeft=30, trimPad=2)),
        ("k=20 mbk=2 rieb=f mlf=0.5", dict(k=20, maxBadKmers0=2, requireBothBad=1, minLenFraction=0.5)),
        ("ktrim=r k=14 ktrimexclusive=t mink=8 mlf=0.9", dict(k=14, ktrimRight=1, ktrimExclusive=1, mink=8, minLenFraction=0.9)),
        ("k=9 hdist=2 mm=f", dict(k=9, hdist=2, maskMiddle=0)), ("ktrim=r k=16 skipr2=t", dict(k=16, ktrimRight=1, skipR2=1)),
        ("k=10 qhdist=2 mm=f", dict(k=10, qhdist=2, maskMiddle=0)), ("k=1 mm=f", dict(k=1, maskMiddle=0)),
        ("ktrim=l k=31 mink=1", dict(k=31, ktrimLeft=1, mink=1)), ("k=24", dict(k=24)), ("ktrim=r k=31 mink=30 tp=-3", dict(k=31, ktrimRight=1, mink=30, trimPad=-3)),
        ("ktrim=r k=23 mink=11 hdist=1 tpe", dict(k=23, ktrimRight=1, mink=11, hdist=1, trimPairsEvenly=1)),
        ("ktrim=r k=17 tpe=t rieb=f minlen=30", dict(k=17, ktrimRight=1, trimPairsEvenly=1, requireBothBad=1, minReadLength=30)),
        ("k=21 qskip=3", dict(k=21, qSkip=3)), ("ktrim=r k=14 mink=6 qskip=2", dict(k=14, ktrimRight=1, mink=6, qSkip=2)),
        ("k=25 speed=5 hdist=1", dict(k=25, speed=5, hdist=1)), ("ktrim=l k=15 mink=9 speed=9 qhdist=1", dict(k=15, ktrimLeft=1, mink=9, speed=9, qhdist=1)),
        ("ktrim=r k=27 speed=12 qskip=2 tpe", dict(k=27, ktrimRight=1, speed=12, qSkip=2, trimPairsEvenly=1)),
        ("k=21 mkf=0.3", dict(k=21, minKmerFraction=0.3)), ("k=13 mkf=0.1 mbk=3 rieb=f forbidn=t", dict(k=13, minKmerFraction=0.1, maxBadKmers0=3, requireBothBad=1, forbidN=1)),
        ("k=25 mcf=0.4", dict(k=25, minCoveredFraction=0.4)), ("k=15 mcf=0.9 hdist=1 restrictleft=60", dict(k=15, minCoveredFraction=0.9, hdist=1, restrictLeft=60)),
        ("k=19 mcf=0.05 mkf=0.5 mm=f", dict(k=19, minCoveredFraction=0.05, minKmerFraction=0.5, maskMiddle=0)),
        ("ktrim=r k=21 mink=9 rskip=3 hdist=1", dict(k=21, ktrimRight=1, mink=9, minSkip=3, maxSkip=3, hdist=1)),
        ("ktrim=r k=13 mink=7 edist=1", dict(k=13, ktrimRight=1, mink=7, edist=1)), ("k=15 edist=1 mm=f", dict(k=15, edist=1, maskMiddle=0)),
        # qhdist = 1 on otherwise plain first-hit configurations: bbduk_qh_kernel (round 4; the ones above with k < 16 or other general flags stay tiled)
        ("ktrim=r k=21 mink=11 hdist=1 qhdist=1", dict(k=21, ktrimRight=1, mink=11, hdist=1, qhdist=1)), ("k=23 qhdist=1", dict(k=23, qhdist=1)),
        ("ktrim=r k=25 mink=12 qhdist=1 qhdist2=0 forbidn=t", dict(k=25, ktrimRight=1, mink=12, qhdist=1, qhdist2=0, forbidN=1)),
        ("k=19 qhdist=1 mm=f forbidn=t", dict(k=19, qhdist=1, maskMiddle=0, forbidN=1)),
        ("ktrim=r k=31 qhdist=1 tpe minlen=20", dict(k=31, ktrimRight=1, qhdist=1, trimPairsEvenly=1, minReadLength=20))]


@pytest.mark.parametrize("ci", range(len(FUZZ)))
def test_fuzz_parity(ci):
    args, okw = FUZZ[ci]
    for seed in range(3):
        rng = random.Random(100 * ci + seed)
        refs = [util.rand_seq(rng, rng.randint(5, 90), 0.01) for _ in range(8)]
        d, o = make_pair(args, okw, refs=refs)
        reads = util.fuzz_reads(rng, refs, 600, [0, 1, 2, 5, 10, 11, 22, 23, 30, 31, 32, 40, 63, 64, 65, 75, 150, 151, 300], junk=True)
        check_batch(d, o, reads, True)
        check_batch(d, o, reads[:301], False)
        d.close()


def test_edge_shapes():
    args, okw, ref = util.CONFIGS["c2"]
    d, o = make_pair(args, okw, ref_fasta=ref)
    ad = B.TRUSEQ_R1
    rng = random.Random(4)
    # empty batch, single read, all-empty reads, reads of 1..70 bases, a tile boundary (129 reads), long reads
    check_batch(d, o, [], False)
    check_batch(d, o, [ad], False)
    check_batch(d, o, [b""] * 10, True)
    check_batch(d, o, [util.rand_seq(rng, n) + ad[:n % 40] for n in range(1, 71)], True)
    check_batch(d, o, [util.rand_seq(rng, 150)] * 129, False)
    longs = [util.rand_seq(rng, n, 0.001) + ad for n in (1000, 5000, 9000, 16000 - len(ad))]
    check_batch(d, o, longs, True)
    check_batch(d, o, longs + [util.rand_seq(rng, 40) + ad[:30]] * 301, False)
    # beyond BBDUK_MAX_READ_LEN the chunked kernels take over (test_long_reads_*, test_kmask_long_sequences, test_ktrimtips_long_reads, test_kscan_long_reads)
    check_batch(d, o, [util.rand_seq(rng, 50000) + ad], False)
    # wrong operator for the mode, odd paired batch
    b, off = pack_reads([ad, ad, ad])
    with pytest.raises(B.BBDukError):
        d.gpu.kfilter_batch(b, off, False)
    with pytest.raises(B.BBDukError):
        d.gpu.ktrim_batch(b, off, True)


def test_tile_kernel_fallback_matches_wave_kernel(monkeypatch):
    """The tile-synchronous kernel only runs when a pair exceeds a wave's LDS slice; force it on ordinary reads so
    that both kernels stay interchangeable (same results, same counters as the oracle)."""
    for cfg in ("c2", "c3", "c1", "c2tpe", "c3q", "c3mkf", "c3mcf"):
        args, okw, ref = util.CONFIGS[cfg]
        d, o = make_pair(args, okw, ref_fasta=ref)
        kw = dict(contam=B.read_fasta(util.PHIX)[0][1], contam_frac=0.05) if "phix" in args else {}
        b, off = B.synth_generate_host(B.synth_params(21, **kw), 0, 6000)
        o.reset_counters(); oa, oi, of = o.process_batch(b, off, True, nthreads=8)
        for force in (False, True):
            d.gpu.test_hook(B.HOOK_FORCE_TILE, int(force))
            d.gpu.reset_counters()
            ga, gi, gf = d.gpu.process_batch(b, off, True)
            assert np.array_equal(ga, oa) and np.array_equal(gi, oi) and np.array_equal(gf, of), (cfg, force)
            assert np.array_equal(d.gpu.counters(), o.counters()), (cfg, force)
        d.close()


def test_packed_format_device_pack_and_operators(monkeypatch):
    """bbduk_pack_bases_device == bbduk_pack_bases_host on arbitrary bytes; the device-buffer packed operators equal the
    ASCII ones (wave kernel and forced tile kernel), including undefined code bits holding garbage."""
    import torch
    rng = np.random.default_rng(9)
    for nb in (0, 1, 15, 16, 17, 31, 32, 33, 1000, 65537):
        raw = rng.integers(0, 256, nb, dtype=np.uint8)
        hc, hu = B.pack_bases_host(raw)
        d_raw = torch.from_numpy(np.concatenate([raw, np.zeros(16, np.uint8)])).cuda()[:nb]
        d_c = torch.full((len(hc) + 4,), -1, dtype=torch.int32, device="cuda"); d_u = torch.full((len(hu) + 4,), -1, dtype=torch.int32, device="cuda")
        B.pack_bases_device(d_raw, d_c, d_u)
        torch.cuda.synchronize()
        gc = d_c.cpu().numpy().view(np.uint32); gu = d_u.cpu().numpy().view(np.uint32)
        if nb:
            defined = np.ones(16 * len(hc), bool); defined[:nb] = ((hu.view(np.uint8)[:, None] >> np.arange(8)) & 1).reshape(-1)[:nb] == 0; defined[nb:] = False
            cm = np.repeat(defined, 2).reshape(-1, 32)
            cmask = (cm.astype(np.uint64) << np.arange(32, dtype=np.uint64)).sum(axis=1).astype(np.uint32)
            assert np.array_equal(gc[:len(hc)] & cmask, hc & cmask), nb
            tail = nb % 32
            um = np.full(len(hu), 0xFFFFFFFF, np.uint32)
            if tail:
                um[-1] = (1 << tail) - 1
            assert np.array_equal(gu[:len(hu)] & um, hu & um), nb
    for cfg in ("c2", "c3", "c2tpe", "c3mcf"):
        args, okw, ref = util.CONFIGS[cfg]
        d, o = make_pair(args, okw, ref_fasta=ref)
        kw = dict(contam=B.read_fasta(util.PHIX)[0][1], contam_frac=0.05) if "phix" in args else {}
        b, off = B.synth_generate_host(B.synth_params(33, **kw), 0, 5000)
        b = b.copy(); b[rng.integers(0, len(b), 3000)] = ord("N")
        o.reset_counters(); oa, oi, of = o.process_batch(b, off, True, nthreads=8)
        n = len(off) - 1
        codes, undef = B.pack_bases_host(b)
        und_bits = ((undef.view(np.uint8)[:, None] >> np.arange(8)) & 1).reshape(-1)[:16 * len(codes)]
        junk = np.repeat(und_bits.astype(np.uint32), 2).reshape(-1, 32)        # garbage in the code bits of undefined bases
        junk = (junk.astype(np.uint64) << np.arange(32, dtype=np.uint64)).sum(axis=1).astype(np.uint32)
        codes = codes | (junk & rng.integers(0, 2 ** 32, len(codes), dtype=np.uint64).astype(np.uint32))
        d_c = torch.from_numpy(codes.view(np.int32)).cuda(); d_u = torch.from_numpy(undef.view(np.int32)).cuda()
        d_off = torch.from_numpy(off).cuda()
        for force in (False, True):
            d.gpu.test_hook(B.HOOK_FORCE_TILE, int(force))
            d_a = torch.empty(n, dtype=torch.int32, device="cuda"); d_i = torch.empty(n, dtype=torch.int32, device="cuda")
            d_f = torch.empty(n, dtype=torch.uint8, device="cuda"); d_ctr = torch.zeros(d.gpu.counters_len, dtype=torch.int64, device="cuda")
            d.gpu.process_batch_packed_device(d_c, d_u, d_off, len(b), True, d_a, d_i, d_f, d_ctr)
            torch.cuda.synchronize()
            assert np.array_equal(d_a.cpu().numpy(), oa) and np.array_equal(d_i.cpu().numpy(), oi) and np.array_equal(d_f.cpu().numpy(), of), (cfg, force)
            assert np.array_equal(d_ctr.cpu().numpy()[:len(o.counters())], o.counters()), (cfg, force)
        d.close()


def test_generator_device_matches_host():
    import torch
    sp = B.synth_params(5, contam=B.read_fasta(util.PHIX)[0][1], contam_frac=0.1)
    n_pairs = 3000
    hb, hoff = B.synth_generate_host(sp, 123456, n_pairs)
    db = torch.empty(n_pairs * 300, dtype=torch.uint8, device="cuda")
    doff = torch.empty(2 * n_pairs + 1, dtype=torch.int64, device="cuda")
    B.synth_generate_device(sp, 123456, n_pairs, db, doff, 0)
    assert np.array_equal(db.cpu().numpy(), hb) and np.array_equal(doff.cpu().numpy(), hoff)
    # beyond 2^31 bytes (the 100 M-read workload is 15e9 bytes): check the tail of a 2.7e9-byte buffer
    n_big = 9_000_000
    db = torch.empty(n_big * 300, dtype=torch.uint8, device="cuda")
    doff = torch.empty(2 * n_big + 1, dtype=torch.int64, device="cuda")
    B.synth_generate_device(sp, 77, n_big, db, doff, 0)
    hb, hoff = B.synth_generate_host(sp, 77 + n_big - 500, 500)
    assert np.array_equal(db[-500 * 300:].cpu().numpy(), hb)
    assert int(doff[-1].item()) == n_big * 300


def test_device_operator_and_full_size_properties():
    """Device-resident operator at a workload the oracle cannot replay whole: (i) an oracle-checked sample,
    (ii) size-independent properties: counters are additive over shards and equal the per-read outputs'
    own sums; trimming is idempotent w.r.t. untouched reads; results do not depend on batch split."""
    import torch
    args, okw, ref = util.CONFIGS["c2"]
    d, o = make_pair(args, okw, ref_fasta=ref)
    sp = B.synth_params(2)
    n_pairs = 2_000_000                                       # 4M reads, 600 Mbases
    n = 2 * n_pairs
    db = torch.empty(n * 150, dtype=torch.uint8, device="cuda")
    doff = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    B.synth_generate_device(sp, 0, n_pairs, db, doff, 0)
    da = torch.empty(n, dtype=torch.int32, device="cuda"); di = torch.empty_like(da)
    df = torch.empty(n, dtype=torch.uint8, device="cuda")
    dc = torch.zeros(d.gpu.counters_len, dtype=torch.int64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    d.gpu.process_batch_device(db, doff, True, da, di, df, dc, st)
    torch.cuda.synchronize()
    a, ids, fl, c = da.cpu().numpy(), di.cpu().numpy(), df.cpu().numpy(), dc.cpu().numpy()
    assert c[B.NCOUNTERS - 1] == 0
    # (i) oracle on a slice in the middle
    s0, s1 = 1_000_000, 1_040_000
    hb, hoff = B.synth_generate_host(sp, s0 // 2, (s1 - s0) // 2)
    oa, oi, of = o.process_batch(hb, hoff, True, nthreads=8)
    assert np.array_equal(a[s0:s1], oa) and np.array_equal(ids[s0:s1], oi) and np.array_equal(fl[s0:s1], of)
    # (ii) properties
    assert c[0] == n and c[1] == n * 150
    removed = (fl & 2) != 0
    assert c[6] + c[8] == n and c[8] == int(removed.sum())
    assert np.array_equal(removed[0::2], removed[1::2])                          # mates leave together
    newlen = 150 - a.astype(np.int64)
    assert c[7] == int(newlen[~removed].sum()) and c[9] == int(newlen[removed].sum())
    assert c[2] == int(((a > 0) | removed).sum())                               # readsKTrimmed
    assert c[3] == int(a.sum() + newlen[removed].sum())                         # basesKTrimmed
    assert np.all((ids > 0) == (a > 0))
    nsc = d.gpu.params.numScaffolds
    assert c[16:16 + nsc].sum() == int((a > 0).sum()) and c[16 + nsc:].sum() == 150 * int((a > 0).sum())
    assert np.array_equal(np.bincount(ids[a > 0], minlength=nsc), c[16:16 + nsc])
    # split invariance: two half batches give the same per-read answers and counters that add up
    dc2 = torch.zeros_like(dc)
    half = n // 2
    d.gpu.process_batch_device(db[:half * 150], doff[:half + 1], True, da[:half], di[:half], df[:half], dc2, st)
    off2 = (doff[half:] - doff[half]).contiguous()
    d.gpu.process_batch_device(db[half * 150:], off2, True, da[half:], di[half:], df[half:], dc2, st)
    torch.cuda.synchronize()
    assert np.array_equal(da.cpu().numpy(), a) and np.array_equal(df.cpu().numpy(), fl) and np.array_equal(dc2.cpu().numpy(), c)


# ---- ktrim=n (kmask): counts, ids, flags, counters and the per-base masks
KMASK = [("ktrim=n k=11 mfc=t", dict(k=11, ktrimN=1, kmaskFullyCovered=1)), ("ktrim=n k=13 mink=6 hdist=1 kmaskfullycovered=t", dict(k=13, ktrimN=1, mink=6, hdist=1, kmaskFullyCovered=1)),
         ("ktrim=n k=15 tp=2 forbidn=t restrictright=50 mfc", dict(k=15, ktrimN=1, trimPad=2, forbidN=1, restrictRight=50, kmaskFullyCovered=1)),
         ("ktrim=n k=12 mink=5 tp=-2 restrictleft=70 mfc=t qhdist=1 qskip=2", dict(k=12, ktrimN=1, mink=5, trimPad=-2, restrictLeft=70, kmaskFullyCovered=1, qhdist=1, qSkip=2)),
         ("ktrim=n k=23", dict(k=23, ktrimN=1)), ("ktrim=N k=23 mink=11 hdist=1", dict(k=23, ktrimN=1, mink=11, hdist=1)),
         ("kmask=lc k=15 tp=2 forbidn=t restrictright=50", dict(k=15, ktrimN=1, trimPad=2, forbidN=1, restrictRight=50)),
         ("ktrim=n k=12 mink=5 tp=-2 qhdist=1 mm=f", dict(k=12, ktrimN=1, mink=5, trimPad=-2, qhdist=1, maskMiddle=0)),
         ("kmask=t k=17 rieb=f minlen=40 skipr1=t", dict(k=17, ktrimN=1, requireBothBad=1, minReadLength=40, skipR1=1)),
         ("ktrim=n k=31 mink=1 speed=3 qskip=2", dict(k=31, ktrimN=1, mink=1, speed=3, qSkip=2))]


def check_kmask(d, o, reads, paired):
    b, off = pack_reads(reads)
    d.gpu.reset_counters(); o.reset_counters()
    ga, gi, gf, gm = d.gpu.kmask_batch(b, off, paired)
    oa, oi, of, om = o.process_batch_mask(b, off, paired, nthreads=4)
    bad = np.nonzero((ga != oa) | (gi != oi) | (gf != of))[0]
    assert len(bad) == 0, [(int(i), reads[i], (int(ga[i]), int(gi[i]), int(gf[i])), (int(oa[i]), int(oi[i]), int(of[i]))) for i in bad[:3]]
    nw = (int(off[-1]) + 31) // 32
    assert np.array_equal(gm[:nw], om[:nw])
    assert np.array_equal(d.gpu.counters(), o.counters())
    if len(reads) and int(off[-1]) > 0:                           # the packed-input variant of the operator: same answers
        import torch
        codes, undef = B.pack_bases_host(b)
        n = len(off) - 1
        d_c = torch.from_numpy(codes.view(np.int32)).cuda(); d_u = torch.from_numpy(undef.view(np.int32)).cuda(); d_o = torch.from_numpy(off).cuda()
        d_a = torch.empty(n, dtype=torch.int32, device="cuda"); d_i = torch.empty_like(d_a); d_f = torch.empty(n, dtype=torch.uint8, device="cuda")
        d_m = torch.zeros(nw + 4, dtype=torch.int32, device="cuda"); d_ctr = torch.zeros(d.gpu.counters_len, dtype=torch.int64, device="cuda")
        rc = B.lib().bbduk_kmask_batch_packed_device(d.gpu.h, d_c.data_ptr(), d_u.data_ptr(), d_o.data_ptr(), n, int(off[-1]), int(paired),
                                                     d_a.data_ptr(), d_i.data_ptr(), d_f.data_ptr(), d_m.data_ptr(), d_ctr.data_ptr(), None)
        torch.cuda.synchronize()
        assert rc == 0
        assert np.array_equal(d_a.cpu().numpy(), oa) and np.array_equal(d_i.cpu().numpy(), oi) and np.array_equal(d_f.cpu().numpy(), of)
        assert np.array_equal(d_m.cpu().numpy().view(np.uint32)[:nw], om[:nw]) and np.array_equal(d_ctr.cpu().numpy(), o.counters())
    return int(ga.sum())


@pytest.mark.parametrize("ci", range(len(KMASK)))
def test_kmask_fuzz_parity(ci):
    args, okw = KMASK[ci]
    total = 0
    for seed in range(3):
        rng = random.Random(900 * ci + seed)
        refs = [util.rand_seq(rng, rng.randint(8, 90), 0.01) for _ in range(8)]
        d, o = make_pair(args, okw, refs=refs)
        reads = util.fuzz_reads(rng, refs, 600, [0, 1, 2, 5, 10, 11, 22, 23, 30, 31, 32, 40, 63, 64, 65, 75, 150, 151, 300], junk=True)
        total += check_kmask(d, o, reads, True)
        total += check_kmask(d, o, reads[:301], False)
        d.close()
    assert total > 0 or okw.get("kmaskFullyCovered")            # (with mink the fully-covered variant clears both ends of every read)


def test_kmask_adapters_and_long_reads():
    d, o = make_pair("ktrim=n k=23 mink=11 hdist=1 ref=adapters", dict(k=23, ktrimN=1, mink=11, hdist=1), ref_fasta=util.ADAPTERS)
    b, off = B.synth_generate_host(B.synth_params(31), 0, 5000)
    d.gpu.reset_counters(); o.reset_counters()
    ga, gi, gf, gm = d.gpu.kmask_batch(b, off, True)
    oa, oi, of, om = o.process_batch_mask(b, off, True, nthreads=8)
    assert np.array_equal(ga, oa) and np.array_equal(gi, oi) and np.array_equal(gf, of)
    assert np.array_equal(gm[:len(om) - 1], om[:-1]) and np.array_equal(d.gpu.counters(), o.counters())
    assert int((ga > 0).sum()) > 1000
    rng = random.Random(5)
    ad = B.TRUSEQ_R1
    longs = [util.rand_seq(rng, n, 0.001) + ad + util.rand_seq(rng, 40) for n in (1000, 5000, 16000 - len(ad) - 40)] + [ad, b"", ad[:12]]
    assert check_kmask(d, o, longs, False) > 0
    with pytest.raises(B.BBDukError):
        d.gpu.ktrim_batch(b, off, True)                        # the handle was created for ktrim=n
    d.close()


def test_concurrent_submitters_share_one_handle():
    """The boundary promises what the JNI convention needs: many host threads may call the batch operator
    on one handle at once; every caller gets its own batch's results and the counters add up."""
    import threading
    args, okw, ref = util.CONFIGS["c2"]
    d, o = make_pair(args, okw, ref_fasta=ref)
    batches = [B.synth_generate_host(B.synth_params(40 + t), 0, 3000 + 500 * t) for t in range(6)]
    want = []
    o.reset_counters()
    for b, off in batches:
        want.append(o.process_batch(b, off, True, nthreads=4))
    d.gpu.reset_counters()
    got = [None] * len(batches)

    def work(t):
        for _ in range(3):                                      # several calls per thread, interleaved with the others
            got[t] = d.gpu.process_batch(batches[t][0], batches[t][1], True)
    th = [threading.Thread(target=work, args=(t,)) for t in range(len(batches))]
    for x in th: x.start()
    for x in th: x.join()
    for t in range(len(batches)):
        for g, w in zip(got[t], want[t]):
            assert np.array_equal(g, w), t
    assert np.array_equal(d.gpu.counters(), 3 * o.counters())   # every batch was counted three times
    d.close()


# ---- ktrim=rl (ktrimtips): the (right, left) amounts, ids, flags and counters
TIPS = [("ktrim=rl k=23", dict(k=23, ktrimLeft=1, ktrimRight=1)), ("ktrim=lr k=23 mink=11 hdist=1 tpe", dict(k=23, ktrimLeft=1, ktrimRight=1, mink=11, hdist=1, trimPairsEvenly=1)),
        ("ktrimtips=30 k=15 tp=1 forbidn=t", dict(k=15, ktrimLeft=1, ktrimRight=1, restrictLeft=30, restrictRight=30, trimPad=1, forbidN=1)),
        ("ktrim=tips k=12 mink=5 ktrimexclusive=t mm=f", dict(k=12, ktrimLeft=1, ktrimRight=1, mink=5, ktrimExclusive=1, maskMiddle=0)),
        ("ktrim=rl k=17 rieb=f minlen=40 qhdist=1", dict(k=17, ktrimLeft=1, ktrimRight=1, requireBothBad=1, minReadLength=40, qhdist=1))]


def check_tips(d, o, reads, paired):
    b, off = pack_reads(reads)
    d.gpu.reset_counters(); o.reset_counters()
    gr, gl, gi, gf = d.gpu.ktrimtips_batch(b, off, paired)
    xr, xl, oi, of = o.process_batch_tips(b, off, paired, nthreads=4)
    bad = np.nonzero((gr != xr) | (gl != xl) | (gi != oi) | (gf != of))[0]
    assert len(bad) == 0, [(int(i), reads[i], (int(gr[i]), int(gl[i]), int(gi[i]), int(gf[i])), (int(xr[i]), int(xl[i]), int(oi[i]), int(of[i]))) for i in bad[:3]]
    assert np.array_equal(d.gpu.counters(), o.counters())
    if len(reads) and int(off[-1]) > 0:                           # the packed-input variant of the operator: same answers
        import torch
        codes, undef = B.pack_bases_host(b)
        n = len(off) - 1
        d_c = torch.from_numpy(codes.view(np.int32)).cuda(); d_u = torch.from_numpy(undef.view(np.int32)).cuda(); d_o = torch.from_numpy(off).cuda()
        d_r = torch.empty(n, dtype=torch.int32, device="cuda"); d_l = torch.empty_like(d_r); d_i = torch.empty_like(d_r); d_f = torch.empty(n, dtype=torch.uint8, device="cuda")
        d_ctr = torch.zeros(d.gpu.counters_len, dtype=torch.int64, device="cuda")
        rc = B.lib().bbduk_ktrimtips_batch_packed_device(d.gpu.h, d_c.data_ptr(), d_u.data_ptr(), d_o.data_ptr(), n, int(off[-1]), int(paired),
                                                         d_r.data_ptr(), d_l.data_ptr(), d_i.data_ptr(), d_f.data_ptr(), d_ctr.data_ptr(), None)
        torch.cuda.synchronize()
        assert rc == 0
        assert np.array_equal(d_r.cpu().numpy(), xr) and np.array_equal(d_l.cpu().numpy(), xl) and np.array_equal(d_i.cpu().numpy(), oi) and np.array_equal(d_f.cpu().numpy(), of)
        assert np.array_equal(d_ctr.cpu().numpy(), o.counters())
    return int(gr.sum()), int(gl.sum())


@pytest.mark.parametrize("ci", range(len(TIPS)))
def test_ktrimtips_fuzz_parity(ci):
    args, okw = TIPS[ci]
    tr = tl = 0
    for seed in range(3):
        rng = random.Random(1200 * ci + seed)
        refs = [util.rand_seq(rng, rng.randint(8, 90), 0.01) for _ in range(8)]
        d, o = make_pair(args, okw, refs=refs)
        reads = util.fuzz_reads(rng, refs, 600, [0, 1, 2, 5, 10, 11, 22, 23, 30, 31, 32, 40, 63, 64, 65, 75, 150, 151, 300], junk=True)
        for paired, rs in ((True, reads), (False, reads[:301])):
            a, b = check_tips(d, o, rs, paired); tr += a; tl += b
        d.close()
    assert tr > 0 and tl > 0


def test_ktrimtips_adapters_both_ends():
    d, o = make_pair("ktrim=rl k=23 mink=11 hdist=1 ref=adapters", dict(k=23, ktrimLeft=1, ktrimRight=1, mink=11, hdist=1), ref_fasta=util.ADAPTERS)
    rng = random.Random(8)
    ad = B.TRUSEQ_R1
    reads = [util.revcomp(ad)[-rng.randint(12, 33):] + util.rand_seq(rng, rng.randint(60, 120)) + ad[:rng.randint(0, 33)] for _ in range(2000)]
    r, l = check_tips(d, o, reads, True)
    assert r > 1000 and l > 1000
    longs = [util.revcomp(ad) + util.rand_seq(rng, n, 0.001) + ad for n in (900, 5000, 16000 - 2 * len(ad))]
    check_tips(d, o, longs, False)
    d.close()


# ---- random flag combinations (seeded): every draw is a legal BBDuk command line; device == oracle
def random_config(rng):
    k = rng.choice([9, 11, 13, 15, 16, 17, 20, 21, 23, 24, 25, 27, 31])
    mode = rng.choice(["r", "l", "f", "f"])
    okw, args = dict(k=k), ["k=%d" % k]
    if mode == "r": okw["ktrimRight"] = 1; args.append("ktrim=r")
    if mode == "l": okw["ktrimLeft"] = 1; args.append("ktrim=l")
    if mode != "f" and rng.random() < 0.5:
        mk = rng.randint(max(1, k - 14), k - 1); okw["mink"] = mk; args.append("mink=%d" % mk)
    hd = rng.choice([0, 0, 1, 1, 2 if k <= 11 else 1])
    if hd: okw["hdist"] = hd; args.append("hdist=%d" % hd)
    if "mink" in okw and rng.random() < 0.3: okw["hdist2"] = 0; args.append("hdist2=0")
    if k <= 13 and rng.random() < 0.3: okw["qhdist"] = 1; args.append("qhdist=1")
    if "mink" not in okw and rng.random() < 0.3: okw["maskMiddle"] = 0; args.append("mm=f")
    if rng.random() < 0.15: okw["rcomp"] = 0; args.append("rcomp=f")
    if rng.random() < 0.25: okw["forbidN"] = 1; args.append("forbidn=t")
    if rng.random() < 0.2: v = rng.randint(20, 120); okw["restrictLeft"] = v; args.append("restrictleft=%d" % v)
    if rng.random() < 0.2: v = rng.randint(20, 120); okw["restrictRight"] = v; args.append("restrictright=%d" % v)
    if rng.random() < 0.15: okw["requireBothBad"] = 1; args.append("rieb=f")
    if rng.random() < 0.3: v = rng.choice([1, 20, 40]); okw["minReadLength"] = v; args.append("minlen=%d" % v)
    if rng.random() < 0.2: okw["minLenFraction"] = 0.5; args.append("mlf=0.5")
    if rng.random() < 0.1: okw["skipR1"] = 1; args.append("skipr1=t")
    if rng.random() < 0.1: okw["qSkip"] = 2; args.append("qskip=2")
    if rng.random() < 0.1: v = rng.randint(1, 8); okw["speed"] = v; args.append("speed=%d" % v)
    if mode != "f":
        if rng.random() < 0.3: v = rng.choice([-2, 1, 3]); okw["trimPad"] = v; args.append("tp=%d" % v)
        if rng.random() < 0.2: okw["ktrimExclusive"] = 1; args.append("ktrimexclusive=t")
        if mode == "r" and rng.random() < 0.4: okw["trimPairsEvenly"] = 1; args.append("tpe")
    else:
        r = rng.random()
        if r < 0.25: v = rng.randint(1, 4); okw["maxBadKmers0"] = v; args.append("mbk=%d" % v)
        elif r < 0.4: okw["minKmerFraction"] = 0.2; args.append("mkf=0.2")
        elif r < 0.55: okw["minCoveredFraction"] = 0.3; args.append("mcf=0.3")
    return " ".join(args), okw


@pytest.mark.parametrize("block", range(6))
def test_random_flag_combinations(block):
    rng = random.Random(31337 + block)
    for _ in range(6):
        args, okw = random_config(rng)
        refs = [util.rand_seq(rng, rng.randint(10, 90), 0.01) for _ in range(8)]
        d, o = make_pair(args, okw, refs=refs)
        reads = util.fuzz_reads(rng, refs, 400, [0, 1, 5, 10, 22, 23, 31, 40, 64, 75, 150, 151, 300], junk=True)
        try:
            check_batch(d, o, reads, True)
            check_batch(d, o, reads[:201], False)
        except AssertionError as e:
            raise AssertionError("config %r: %s" % (args, e))
        d.close()


# ---- device-side table build (bbduk_build_table_device): same key -> id map as the oracle's loader, same results
DEVBUILD = [("ktrim=r k=23 mink=11 hdist=1", dict(k=23, ktrimRight=1, mink=11, hdist=1)), ("k=31 hdist=1", dict(k=31, hdist=1)),
            ("k=27", dict(k=27)), ("ktrim=l k=13 mink=6 hdist=1 hdist2=0", dict(k=13, ktrimLeft=1, mink=6, hdist=1, hdist2=0)),
            ("k=11 hdist=2 mm=f", dict(k=11, hdist=2, maskMiddle=0)), ("ktrim=r k=12 mink=5 hdist=2 hdist2=2 rcomp=f", dict(k=12, ktrimRight=1, mink=5, hdist=2, hdist2=2, rcomp=0)),
            ("k=20 hdist=1 mm=2", dict(k=20, hdist=1, midMaskLen=2)), ("ktrim=n k=16 mink=16 hdist=1", dict(k=16, ktrimN=1, mink=16, hdist=1)),
            # reference-side edit distance on the device (round 4; BBDukIndexMod.java:414-445): substitutions, deletions, insertions
            ("k=21 edist=1", dict(k=21, hdist=1, edist=1)), ("ktrim=r k=17 mink=8 edist=1 mm=f", dict(k=17, ktrimRight=1, mink=8, hdist=1, edist=1, maskMiddle=0)),
            ("ktrim=l k=15 mink=7 edist=1 edist2=0 rcomp=f", dict(k=15, ktrimLeft=1, mink=7, hdist=1, hdist2=0, edist=1, edist2=0, rcomp=0)),
            ("ktrim=r k=19 mink=9 hdist=1 edist2=1", dict(k=19, ktrimRight=1, mink=9, hdist=1, hdist2=1, edist=0, edist2=1))]


@pytest.mark.parametrize("ci", range(len(DEVBUILD)))
def test_device_built_table_equals_oracle_table(ci):
    args, okw = DEVBUILD[ci]
    rng = random.Random(500 + ci)
    refs = [util.rand_seq(rng, rng.choice([3, 10, 11, 12, 23, 31, 40, 90, 200]), 0.02) for _ in range(12)] + [b"", b"ACGTN" * 9]
    d = B.BBDuk(args, refs=refs, build="device")
    o = Oracle(**okw)
    for r in refs:
        o.add_ref(r)
    ks, vs = o.dump_pairs()
    assert d.gpu.table_size == o.stored_kmers == len(ks)
    assert np.array_equal(d.gpu.table_lookup(ks), vs)                    # every key, with the id of the first scaffold
    have = set(ks.tolist())
    absent = np.array([int(k) ^ 5 for k in ks[:2000].tolist() if (int(k) ^ 5) not in have], np.int64)
    assert np.all(d.gpu.table_lookup(absent) == -1)
    if okw.get("ktrimN"):
        reads = util.fuzz_reads(rng, refs, 300, [0, 5, 16, 31, 64, 150], junk=True)
        check_kmask(d, o, reads, True)
    else:
        reads = util.fuzz_reads(rng, refs, 400, [0, 1, 5, 10, 23, 31, 40, 64, 150, 300], junk=True)
        check_batch(d, o, reads, True)
    d.close()


def test_device_build_refuses_reference_side_skipping():
    with pytest.raises(B.BBDukError):
        B.BBDuk("k=21 rskip=3", refs=[b"ACGTTGCATGCATGCATGGGATCGATCGATTTACGAC"], build="device")


def test_device_built_adapter_and_phix_tables():
    for cfg in ("c2", "c3", "c1"):
        args, okw, ref = util.CONFIGS[cfg]
        d = B.BBDuk(args, build="device")
        o = Oracle(**okw); o.load_fasta(ref)
        ks, vs = o.dump_pairs()
        assert d.gpu.table_size == len(ks) and np.array_equal(d.gpu.table_lookup(ks), vs)
        kw = dict(contam=B.read_fasta(util.PHIX)[0][1], contam_frac=0.05) if "phix" in args else {}
        b, off = B.synth_generate_host(B.synth_params(51, **kw), 0, 5000)
        d.gpu.reset_counters(); o.reset_counters()
        ga, gi, gf = d.gpu.process_batch(b, off, True)
        oa, oi, of = o.process_batch(b, off, True, nthreads=8)
        assert np.array_equal(ga, oa) and np.array_equal(gi, oi) and np.array_equal(gf, of)
        assert np.array_equal(d.gpu.counters(), o.counters())
        d.close()


# ---- kbig (k>31 by runs of 31-mers), findBestMatch: behind the kfilter operators; ksplit: its own operator
BIG = [("k=40", dict(k=31, kbig=40)), ("k=32 hdist=1", dict(k=31, kbig=32, hdist=1)), ("k=50 mbk=3 rieb=f", dict(k=31, kbig=50, maxBadKmers0=3, requireBothBad=1)),
       ("k=45 forbidn=t restrictleft=120", dict(k=31, kbig=45, forbidN=1, restrictLeft=120)), ("k=36 mkf=0.2 qhdist=1", dict(k=31, kbig=36, minKmerFraction=0.2, qhdist=1)),
       ("k=25 fbm", dict(k=25, findBestMatch=1)), ("k=13 findbestmatch=t hdist=1 mm=f rieb=f", dict(k=13, findBestMatch=1, hdist=1, maskMiddle=0, requireBothBad=1)),
       ("k=19 fbm=t skipr1=t restrictright=90", dict(k=19, findBestMatch=1, skipR1=1, restrictRight=90)),
       ("k=40 mcf=0.3", dict(k=31, kbig=40, minCoveredFraction=0.3))]
SPLIT = [("k=23 ksplit=t", dict(k=23, ksplit=1)), ("k=15 ksplit mink=7 hdist=1", dict(k=15, ksplit=1, mink=7, hdist=1)),
         ("k=13 ksplit=t tp=-2 forbidn=t", dict(k=13, ksplit=1, trimPad=-2, forbidN=1)), ("k=17 ksplit mink=9 restrictleft=100 qhdist=1", dict(k=17, ksplit=1, mink=9, restrictLeft=100, qhdist=1)),
         ("k=11 ksplit mink=5 restrictright=60 rcomp=f", dict(k=11, ksplit=1, mink=5, restrictRight=60, rcomp=0)),
         # (soak seed 301004544: a left-end short k-mer of 2 bases with a negative trimpad leaves rightmost at the -1 it starts from, :2434)
         ("k=13 mink=2 hdist=1 forbidn=t restrictleft=68 tp=-3 ksplit=t", dict(k=13, mink=2, hdist=1, forbidN=1, restrictLeft=68, trimPad=-3, ksplit=1))]


@pytest.mark.parametrize("ci", range(len(BIG)))
def test_kbig_findbestmatch_fuzz_parity(ci):
    args, okw = BIG[ci]
    hits = 0
    for seed in range(4):
        rng = random.Random(21000 + 100 * ci + seed)
        refs, reads = util.long_refs_and_reads(rng)
        reads = reads * 8 + [rng.choice(refs)[:rng.randint(30, 200)] + util.rand_seq(rng, rng.randint(0, 300)) for _ in range(80)]
        if len(reads) % 2:
            reads.append(b"")
        d, o = make_pair(args, okw, refs=refs)
        check_batch(d, o, reads[:203], False)
        check_batch(d, o, reads, True)                     # ASCII and packed formats, per-read results and counters
        hits += int(d.gpu.counters()[B.NCOUNTERS:B.NCOUNTERS + d.gpu.params.numScaffolds].sum())
        d.close()
    assert hits > 20


def test_kbig_long_reads_and_phix():
    """Runs that span several 256-position blocks and many reads per tile: phiX pieces of 300..6000 bases, k=40 and k=100."""
    rng = random.Random(8)
    phix = B.read_fasta(util.PHIX)[0][1]
    for args, okw in (("k=40 ref=phix", dict(k=31, kbig=40)), ("k=100 hdist=1 ref=phix mbk=700", dict(k=31, kbig=100, hdist=1, maxBadKmers0=700))):
        d, o = make_pair(args, okw, ref_fasta=util.PHIX)
        reads = []
        for _ in range(300):
            L = rng.choice([300, 700, 1500, 4000]); a = rng.randrange(len(phix) - L)
            piece = bytearray(phix[a:a + L])
            for _ in range(rng.randint(0, 6)):             # substitutions / Ns break the runs
                piece[rng.randrange(L)] = rng.choice(b"ACGTN")
            reads.append(util.rand_seq(rng, rng.randint(0, 200)) + (util.revcomp(bytes(piece)) if rng.random() < 0.5 else bytes(piece)) + util.rand_seq(rng, rng.randint(0, 200)))
        check_batch(d, o, reads, True, nthreads=8)
        assert d.gpu.counters()[4] > 100                    # readsKFiltered
        d.close()


def test_findbestmatch_id_overflow_is_an_error():
    rng = random.Random(5)
    refs = [util.rand_seq(rng, 40) for _ in range(70)]     # one read made of 70 scaffolds: more distinct ids than a wave keeps
    d = B.BBDuk("k=21 fbm", refs=refs)
    b, off = pack_reads([b"".join(refs)])
    with pytest.raises(B.BBDukError):
        d.gpu.process_batch(b, off, False)
    b, off = pack_reads([b"".join(refs[:64])])             # exactly 64 still fits
    a, ids, fl = d.gpu.process_batch(b, off, False)
    assert ids[0] == 1 and a[0] == 64 * 20
    d.close()


def check_split(d, o, reads):
    b, off = pack_reads(reads)
    d.gpu.reset_counters(); o.reset_counters()
    gx, gi, gf, gl, gr = d.gpu.ksplit_batch(b, off)
    ox, oi, of, ol, orr = o.process_batch_split(b, off, nthreads=4)
    bad = np.nonzero((gx != ox) | (gi != oi) | (gf != of) | (gl != ol) | (gr != orr))[0]
    assert len(bad) == 0, [(int(i), reads[i], (int(gx[i]), int(gi[i]), int(gf[i]), int(gl[i]), int(gr[i])),
                            (int(ox[i]), int(oi[i]), int(of[i]), int(ol[i]), int(orr[i]))) for i in bad[:3]]
    assert np.array_equal(d.gpu.counters(), o.counters())
    return gf


@pytest.mark.parametrize("ci", range(len(SPLIT)))
def test_ksplit_fuzz_parity(ci):
    args, okw = SPLIT[ci]
    nsplit = 0
    for seed in range(4):
        rng = random.Random(22000 + 100 * ci + seed)
        refs, reads = util.long_refs_and_reads(rng)
        refs = [r[:rng.randint(12, 60)] for r in refs]
        reads = reads * 4 + [util.rand_seq(rng, rng.randint(0, 80)) + rng.choice(refs) + util.rand_seq(rng, rng.randint(0, 80)) for _ in range(200)]
        d, o = make_pair(args, okw, refs=refs)
        nsplit += int((check_split(d, o, reads) & 2).astype(bool).sum())
        check_split(d, o, [])
        with pytest.raises(B.BBDukError):
            d.gpu.process_batch(*pack_reads(reads[:4]), True)          # ksplit is for unpaired reads
        d.close()
    assert nsplit > 20


# ---- reads longer than the LDS tiles: bbduk_long_kernel (chunked scan, state carried across chunks)
LONG = [("ktrim=r k=23 mink=11 hdist=1", dict(k=23, ktrimRight=1, mink=11, hdist=1)), ("ktrim=l k=23 mink=11", dict(k=23, ktrimLeft=1, mink=11)),
        ("ktrim=r k=25 forbidn=t tp=2 ktrimexclusive=t", dict(k=25, ktrimRight=1, forbidN=1, trimPad=2, ktrimExclusive=1)),
        ("ktrim=l k=19 restrictleft=30000 qhdist=1", dict(k=19, ktrimLeft=1, restrictLeft=30000, qhdist=1)),
        ("k=31 hdist=1", dict(k=31, hdist=1)), ("k=27 mbk=40 rieb=f", dict(k=27, maxBadKmers0=40, requireBothBad=1)),
        ("k=25 mcf=0.01", dict(k=25, minCoveredFraction=0.01)), ("k=21 mkf=0.002 forbidn=t", dict(k=21, minKmerFraction=0.002, forbidN=1)),
        ("k=31 restrictright=45000 qskip=3", dict(k=31, restrictRight=45000, qSkip=3)), ("ktrim=r k=21 mink=8 tpe minlen=100", dict(k=21, ktrimRight=1, mink=8, trimPairsEvenly=1, minReadLength=100)),
        # scan spans that are empty (start > stop): the short k-mers of the end are still looked up next to `stop` (:2072-2076)
        ("ktrim=r k=13 mink=7 hdist=1 restrictleft=118 restrictright=60 tp=-2 ktrimexclusive=t", dict(k=13, ktrimRight=1, mink=7, hdist=1, restrictLeft=118, restrictRight=60, trimPad=-2, ktrimExclusive=1)),
        ("ktrim=l k=15 mink=6 restrictleft=40 restrictright=30", dict(k=15, ktrimLeft=1, mink=6, restrictLeft=40, restrictRight=30))]


@pytest.mark.parametrize("ci", range(len(LONG)))
def test_long_reads_chunked_kernel(ci):
    args, okw = LONG[ci]
    rng = random.Random(33000 + ci)
    refs = [util.rand_seq(rng, rng.randint(30, 120)) for _ in range(6)]
    d, o = make_pair(args, okw, refs=refs)
    reads = []
    for L in (41000, 50000, 16385 * 3, 120000, 2432 * 20 + 7, 90000, 45000, 70001):
        g = bytearray(util.rand_seq(rng, L, 0.0005))
        for _ in range(rng.randint(0, 4)):                      # reference pieces anywhere, also across chunk borders and at the ends
            r = rng.choice(refs); r = util.revcomp(r) if rng.random() < 0.5 else r
            pos = rng.choice([0, L - len(r), rng.randrange(L - len(r)), 2432 * rng.randint(1, 15) - rng.randint(0, 40)])
            g[pos:pos + len(r)] = r
        if rng.random() < 0.5:
            g[L - 14:] = rng.choice(refs)[:14]                   # a short k-mer at the very end
        if rng.random() < 0.5:
            g[:15] = rng.choice(refs)[-15:]
        reads.append(bytes(g))
    reads += util.fuzz_reads(rng, refs, 20, [0, 10, 150, 3000], junk=True)      # short mates ride along with long ones
    reads += [util.rand_seq(rng, 100) + r[:14] + util.rand_seq(rng, rng.randint(100, 900)) for r in refs]   # a short k-mer ending at base 114..118
    rng.shuffle(reads)
    check_batch(d, o, reads, True, nthreads=8)
    check_batch(d, o, reads[:15], False, nthreads=8)
    assert d.gpu.counters()[B.NCOUNTERS:].sum() > 0 or "restrictright=30" in args      # (that span is empty and its left end lies outside it)
    d.close()


def test_long_read_phix_genome_as_one_read():
    """The whole phiX genome (5386 bases) repeated 40 times = one 215 kb read, against ref=phix: every k-mer hits."""
    phix = B.read_fasta(util.PHIX)[0][1]
    d, o = make_pair("k=31 mbk=100000 ref=phix", dict(k=31, maxBadKmers0=100000), ref_fasta=util.PHIX)
    read = bytes(phix) * 40
    check_batch(d, o, [read, read[:100000]], False)
    a, ids, fl = d.gpu.process_batch(*pack_reads([read]), False)
    assert a[0] == 100001 and fl[0] == 3                       # the early exit at maxBadKmers+1 (BBDukProcessorS.java:1572-1590)
    d.close()



def test_concurrent_device_operators_on_streams():
    """Device-buffer operators of ONE handle issued from several host threads on their own streams: every launch has its own
    pre-pass flag, so a batch that needs the tile / long-read kernel cannot redirect a neighbour's launch."""
    import threading
    import torch
    args, okw, ref = util.CONFIGS["c2"]
    d, o = make_pair(args, okw, ref_fasta=ref)
    rng = random.Random(12)
    ad = B.TRUSEQ_R1
    plain = [util.rand_seq(rng, 150) if i % 3 else util.rand_seq(rng, 100) + ad[:50] for i in range(4000)]
    tile = plain[:2000] + [util.rand_seq(rng, 3000) + ad, util.rand_seq(rng, 100)]                  # one pair beyond a wave's planes
    longb = plain[:1000] + [util.rand_seq(rng, 60000) + ad, util.rand_seq(rng, 90)]                 # one read beyond the LDS tiles
    batches = [plain, tile, longb, plain[:1500], tile[100:], longb[200:]]
    want = []
    for rd in batches:
        b, off = pack_reads(rd)
        want.append(o.process_batch(b, off, True, nthreads=4))
    got = [None] * len(batches)
    errs = []

    def work(t):
        try:
            b, off = pack_reads(batches[t])
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                db = torch.from_numpy(np.concatenate([b, np.zeros(16, np.uint8)])).cuda()[:len(b)]; do = torch.from_numpy(off).cuda()
                n = len(off) - 1
                da = torch.empty(n, dtype=torch.int32, device="cuda"); di = torch.empty_like(da); df = torch.empty(n, dtype=torch.uint8, device="cuda")
                dc = torch.zeros(d.gpu.counters_len, dtype=torch.int64, device="cuda")
                for _ in range(20):
                    d.gpu.process_batch_device(db, do, True, da, di, df, dc, st.cuda_stream)
                st.synchronize()
                got[t] = (da.cpu().numpy(), di.cpu().numpy(), df.cpu().numpy())
        except Exception as e:                                   # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=work, args=(t,)) for t in range(len(batches))]
    for x in th: x.start()
    for x in th: x.join()
    assert not errs, errs
    for t in range(len(batches)):
        for g, w in zip(got[t], want[t]):
            assert np.array_equal(g, w), t
    d.close()


KMASK_LONG = [("ktrim=n k=21 mfc=t", dict(k=21, ktrimN=1, kmaskFullyCovered=1)), ("ktrim=n k=17 mink=8 tp=1 mfc=t hdist=1", dict(k=17, ktrimN=1, mink=8, trimPad=1, kmaskFullyCovered=1, hdist=1)),
              ("ktrim=n k=13 mink=7 restrictleft=118 restrictright=60", dict(k=13, ktrimN=1, mink=7, restrictLeft=118, restrictRight=60)), ("ktrim=n k=23", dict(k=23, ktrimN=1)), ("ktrim=N k=21 mink=9 hdist=1 tp=2", dict(k=21, ktrimN=1, mink=9, hdist=1, trimPad=2)),
              ("kmask=lc k=25 tp=-3 forbidn=t", dict(k=25, ktrimN=1, trimPad=-3, forbidN=1)), ("ktrim=n k=19 mink=7 restrictleft=60000 qhdist=1 skipr2=t", dict(k=19, ktrimN=1, mink=7, restrictLeft=60000, qhdist=1, skipR2=1)),
              ("ktrim=n k=27 restrictright=50000 minlen=45000 rieb=f", dict(k=27, ktrimN=1, restrictRight=50000, minReadLength=45000, requireBothBad=1))]


@pytest.mark.parametrize("ci", range(len(KMASK_LONG)))
def test_kmask_long_sequences(ci):
    """ktrim=n on sequences beyond the tiled kernel's planes (contigs): bbduk_kmask_long_kernel, mixed with short reads."""
    args, okw = KMASK_LONG[ci]
    rng = random.Random(44000 + ci)
    refs = [util.rand_seq(rng, rng.randint(30, 120)) for _ in range(6)]
    d, o = make_pair(args, okw, refs=refs)
    reads = []
    for L in (33000, 50000, 2432 * 20 + 7, 120000, 70001, 40000):
        g = bytearray(util.rand_seq(rng, L, 0.0005))
        for _ in range(rng.randint(0, 5)):
            r = rng.choice(refs); r = util.revcomp(r) if rng.random() < 0.5 else r
            pos = rng.choice([0, L - len(r), rng.randrange(L - len(r)), 2432 * rng.randint(1, 13) - rng.randint(0, 40)])
            g[pos:pos + len(r)] = r
        if rng.random() < 0.6:
            g[L - 14:] = rng.choice(refs)[:14]
        if rng.random() < 0.6:
            g[:15] = rng.choice(refs)[-15:]
        reads.append(bytes(g))
    reads += util.fuzz_reads(rng, refs, 30, [0, 10, 150, 3000], junk=True)
    rng.shuffle(reads)
    check_kmask(d, o, reads, True)
    check_kmask(d, o, reads[:17], False)
    assert d.gpu.counters()[3] > 0
    d.close()


TIPS_LONG = [("ktrim=rl k=23 mink=11 hdist=1", dict(k=23, ktrimLeft=1, ktrimRight=1, mink=11, hdist=1)),
             ("ktrimtips=150 k=21 mink=9", dict(k=21, ktrimLeft=1, ktrimRight=1, mink=9, restrictLeft=150, restrictRight=150)),
             ("ktrim=lr k=25 tp=1 forbidn=t ktrimexclusive=t tpe minlen=20000", dict(k=25, ktrimLeft=1, ktrimRight=1, trimPad=1, forbidN=1, ktrimExclusive=1, trimPairsEvenly=1, minReadLength=20000)),
             ("ktrimtips=40000 k=19 qhdist=1 rieb=f", dict(k=19, ktrimLeft=1, ktrimRight=1, restrictLeft=40000, restrictRight=40000, qhdist=1, requireBothBad=1))]


@pytest.mark.parametrize("ci", range(len(TIPS_LONG)))
def test_ktrimtips_long_reads(ci):
    """ktrim=rl / ktrimtips on reads beyond the tiled kernel's planes (long-read adapter trimming): bbduk_long_tips_kernel."""
    args, okw = TIPS_LONG[ci]
    rng = random.Random(55000 + ci)
    refs = [util.rand_seq(rng, rng.randint(30, 120)) for _ in range(6)]
    d, o = make_pair(args, okw, refs=refs)
    reads = []
    for L in (33000, 50000, 2432 * 20 + 7, 120000, 70001, 40000, 90011, 35000):
        g = bytearray(util.rand_seq(rng, L, 0.0005))
        for _ in range(rng.randint(0, 5)):
            r = rng.choice(refs); r = util.revcomp(r) if rng.random() < 0.5 else r
            pos = rng.choice([0, L - len(r), rng.randrange(L - len(r)), rng.randint(0, 120), L - len(r) - rng.randint(0, 120),
                              L // 2 + rng.randint(-60, 60), 2432 * rng.randint(1, 13) - rng.randint(0, 40)])
            g[pos:pos + len(r)] = r
        if rng.random() < 0.6:
            g[L - 14:] = rng.choice(refs)[:14]
        if rng.random() < 0.6:
            g[:15] = rng.choice(refs)[-15:]
        reads.append(bytes(g))
    reads += util.fuzz_reads(rng, refs, 30, [0, 10, 150, 3000], junk=True)
    rng.shuffle(reads)
    check_tips(d, o, reads, True)
    check_tips(d, o, reads[:17], False)
    assert d.gpu.counters()[3] > 0
    d.close()


KSCAN_LONG = [("k=40", dict(k=31, kbig=40)), ("k=45 mbk=30 rieb=f forbidn=t", dict(k=31, kbig=45, maxBadKmers0=30, requireBothBad=1, forbidN=1)),
              ("k=36 mkf=0.0005 hdist=1", dict(k=31, kbig=36, minKmerFraction=0.0005, hdist=1)),
              ("k=25 fbm", dict(k=25, findBestMatch=1)), ("k=21 fbm=t restrictright=60000 skipr1=t", dict(k=21, findBestMatch=1, restrictRight=60000, skipR1=1))]
KSPLIT_LONG = [("k=13 ksplit=t mink=7 restrictleft=118 restrictright=60", dict(k=13, ksplit=1, mink=7, restrictLeft=118, restrictRight=60)), ("k=23 ksplit=t", dict(k=23, ksplit=1)), ("k=21 ksplit mink=9 hdist=1", dict(k=21, ksplit=1, mink=9, hdist=1)),
               ("k=19 ksplit=t tp=-2 forbidn=t restrictleft=70000", dict(k=19, ksplit=1, trimPad=-2, forbidN=1, restrictLeft=70000))]


def _long_mix(rng, refs, short_lens):
    reads = []
    for L in (33000, 50000, 2432 * 20 + 7, 120000, 70001, 40000, 90011, 35000):
        g = bytearray(util.rand_seq(rng, L, 0.0005))
        for _ in range(rng.randint(0, 5)):
            r = rng.choice(refs); r = util.revcomp(r) if rng.random() < 0.5 else r
            r = r[:rng.randint(20, len(r))]
            pos = rng.choice([0, L - len(r), rng.randrange(L - len(r)), 2432 * rng.randint(1, 13) - rng.randint(0, 60)])
            g[pos:pos + len(r)] = r
        if rng.random() < 0.5:
            g[L - 14:] = rng.choice(refs)[:14]
        if rng.random() < 0.5:
            g[:15] = rng.choice(refs)[-15:]
        reads.append(bytes(g))
    reads += util.fuzz_reads(rng, refs, 30, short_lens, junk=True)
    rng.shuffle(reads)
    return reads


@pytest.mark.parametrize("ci", range(len(KSCAN_LONG)))
def test_kscan_long_reads(ci):
    """k>31 and findBestMatch on reads beyond the tiled kernel's planes: bbduk_kscan_long_kernel (runs across chunk borders)."""
    args, okw = KSCAN_LONG[ci]
    rng = random.Random(66000 + ci)
    refs = [util.rand_seq(rng, rng.randint(60, 400)) for _ in range(6)]
    d, o = make_pair(args, okw, refs=refs)
    reads = _long_mix(rng, refs, [0, 10, 150, 3000])
    check_batch(d, o, reads[:17], False, nthreads=8)
    check_batch(d, o, reads, True, nthreads=8)
    assert d.gpu.counters()[B.NCOUNTERS:].sum() > 0
    d.close()


@pytest.mark.parametrize("ci", range(len(KSPLIT_LONG)))
def test_ksplit_long_reads(ci):
    args, okw = KSPLIT_LONG[ci]
    rng = random.Random(77000 + ci)
    refs = [util.rand_seq(rng, rng.randint(30, 90)) for _ in range(6)]
    d, o = make_pair(args, okw, refs=refs)
    reads = _long_mix(rng, refs, [0, 10, 150, 3000])
    fl = check_split(d, o, reads)
    assert int((fl & 2).astype(bool).sum()) > 0
    d.close()



@pytest.mark.parametrize("lds", ["0", "10", None])
def test_crowded_map_and_reduced_filter(monkeypatch, lds):
    """The rarely taken lookup paths: a map squeezed to ~3.4 keys per 4-way bucket (long overflow chains, continuation flags
    nearly everywhere) with the LDS presence filter dropped, tiny or as usual (bbduk_test_hook BUCKET_BITS / LDS_BITS, honoured
    by bbduk_finalize_table)."""
    import math
    for cfg in ("c2", "c3", "c1"):
        args, okw, ref = util.CONFIGS[cfg]
        o = Oracle(**okw); o.load_fasta(ref)
        hooks = {B.HOOK_BUCKET_BITS: max(4, math.ceil(math.log2(o.stored_kmers / 3.4)))}
        if lds is not None:
            hooks[B.HOOK_LDS_BITS] = int(lds)
        d = B.BBDuk(args, hooks=hooks)
        assert d.stored_kmers == o.stored_kmers
        kw = dict(contam=B.read_fasta(util.PHIX)[0][1], contam_frac=0.05) if "phix" in args else {}
        b, off = B.synth_generate_host(B.synth_params(77, **kw), 0, 20000)
        o.reset_counters(); oa, oi, of = o.process_batch(b, off, True, nthreads=8)
        ga, gi, gf = d.gpu.process_batch(b, off, True)
        assert np.array_equal(ga, oa) and np.array_equal(gi, oi) and np.array_equal(gf, of), cfg
        assert np.array_equal(d.gpu.counters(), o.counters())
        ks, vs = o.dump_pairs()
        sel = np.random.default_rng(1).choice(len(ks), min(5000, len(ks)), replace=False)
        assert np.array_equal(d.gpu.table_lookup(ks[sel]), vs[sel])                     # every key is still found, with its id
        d.close()


@pytest.mark.parametrize("build", ["host", "device"])
def test_large_map_without_lds_filter(build):
    """More than 2^22 keys: the map gets no LDS presence filter (the HBM-resident regime of BASELINE config 4, in small).
    One 4.6 Mbase random scaffold, k=31, reads half drawn from it."""
    rng = np.random.default_rng(5)
    ref = rng.choice(np.frombuffer(b"ACGT", np.uint8), 4_600_000).tobytes()
    d, o = make_pair("k=31", dict(k=31), refs=[ref]) if build == "host" else (B.BBDuk("k=31", refs=[ref], build="device"), None)
    if o is None:
        o = Oracle(k=31); o.add_ref(ref)
    assert d.stored_kmers == o.stored_kmers > (1 << 22)
    prng = random.Random(9)
    reads = []
    for i in range(20000):
        if i % 2:
            a = prng.randrange(len(ref) - 150); r = ref[a:a + 150]
            reads.append(util.revcomp(r) if prng.random() < 0.5 else r)
        else:
            reads.append(util.rand_seq(prng, 150, 0.002))
    check_batch(d, o, reads, True, nthreads=8)
    assert d.gpu.counters()[4] > 9000
    d.close()


# ---- the C ABI's error behaviour (include/bbduk_gpu.h: BBDUK_ERR_*): a bad call returns its code, sets bbduk_last_error and
# leaves the handle usable; nothing is computed on the CPU instead.
def test_abi_error_codes():
    L = B.lib()
    host = B.HostIndex("ktrim=r k=23 mink=11 hdist=1")
    host.add_ref(b"AGATCGGAAGAGCACACGTCTGAACTCCAGTCAC")
    host.build_index()
    g = B.BBDukGpu(host.params(0))
    reads = [b"ACGTACGTACGTAGATCGGAAGAGCACACGTCTGAACTCCAGTCAC", b"TTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTT"]
    b, off = pack_reads(reads)
    a = np.empty(2, np.int32); ids = np.empty(2, np.int32); fl = np.empty(2, np.uint8)
    call = lambda fn, bases, offs, n, paired: fn(g.h, bases.ctypes.data, offs.ctypes.data, n, paired, a.ctypes.data, ids.ctypes.data, fl.ctypes.data)
    assert call(L.bbduk_ktrim_batch, b, off, 2, 0) == -2 and b"finalized" in L.bbduk_last_error(g.h)     # BBDUK_ERR_STATE: batch before the map
    g._check(L.bbduk_host_upload_index(host.h, g.h), "upload_index")
    assert call(L.bbduk_kfilter_batch, b, off, 2, 0) == -2                       # the other mode's operator
    m = np.zeros(8, np.uint32)
    assert L.bbduk_kmask_batch(g.h, b.ctypes.data, off.ctypes.data, 2, 0, a.ctypes.data, ids.ctypes.data, fl.ctypes.data, m.ctypes.data) == -2
    assert call(L.bbduk_ktrim_batch, b, off, -1, 0) == -1                        # BBDUK_ERR_ARG from here on
    assert L.bbduk_ktrim_batch(g.h, b.ctypes.data, None, 2, 0, a.ctypes.data, ids.ctypes.data, fl.ctypes.data) == -1
    assert L.bbduk_ktrim_batch(g.h, b.ctypes.data, off.ctypes.data, 2, 0, None, ids.ctypes.data, fl.ctypes.data) == -1
    assert L.bbduk_ktrim_batch(g.h, None, off.ctypes.data, 2, 0, a.ctypes.data, ids.ctypes.data, fl.ctypes.data) == -1
    assert L.bbduk_ktrim_batch(None, b.ctypes.data, off.ctypes.data, 2, 0, a.ctypes.data, ids.ctypes.data, fl.ctypes.data) == -1
    b3, off3 = pack_reads(reads + [b"ACGT"])
    assert call(L.bbduk_ktrim_batch, b3, off3, 3, 1) == -1                       # an odd number of reads cannot be pairs
    shifted = off + 1
    assert call(L.bbduk_ktrim_batch, b, shifted, 2, 0) == -1                     # offsets[0] != 0
    bad = off.copy(); bad[1] = bad[2] + 5                                         # not ascending: read 1 would get a negative length
    assert call(L.bbduk_ktrim_batch, b, bad, 2, 0) == -1 and b"ascend" in L.bbduk_last_error(g.h)
    bad = off.copy(); bad[1] = -3
    assert call(L.bbduk_ktrim_batch, b, bad, 2, 0) == -1
    nid = np.zeros(2, np.int32); mi = np.zeros((2, 4), np.int32); mc = np.zeros((2, 4), np.int32)
    assert L.bbduk_kfilter_batch_matches(g.h, b.ctypes.data, off.ctypes.data, 2, 0, a.ctypes.data, ids.ctypes.data, fl.ctypes.data, 4,
                                         nid.ctypes.data, mi.ctypes.data, mc.ctypes.data) == -2     # match lists need a findBestMatch handle
    assert L.bbduk_get_counters(g.h, None, 0) == -1
    assert L.bbduk_upload_pairs(g.h, None, None, 0) == -2                        # the map is final
    assert call(L.bbduk_ktrim_batch, b, off, 0, 0) == 0                          # an empty batch is fine
    assert call(L.bbduk_ktrim_batch, b, off, 2, 0) == 0 and a[0] == len(reads[0]) - 12 and a[1] == 0     # and the handle still works
    g.close(); g.close()                                                          # closing twice is harmless
    # scaffold ids index the counter vector on the device: anything outside 1..numScaffolds-1 is refused at upload
    g2 = B.BBDukGpu(host.params(0))
    ns = host.num_scaffolds
    k1 = np.array([5, 6], np.int64)
    for badv in (0, -1, ns, ns + 7):
        v = np.array([1, badv], np.int32)
        assert L.bbduk_upload_pairs(g2.h, k1.ctypes.data, v.ctypes.data, 2) == -1 and b"scaffold id" in L.bbduk_last_error(g2.h)
        assert L.bbduk_upload_table_way(g2.h, 0, 7, k1.ctypes.data, v.ctypes.data, 2, None, None, 0) == -1
        assert L.bbduk_upload_table_way(g2.h, 0, 7, None, None, 0, k1.ctypes.data, v.ctypes.data, 2) == -1
    v = np.array([1, ns - 1], np.int32)
    assert L.bbduk_upload_pairs(g2.h, k1.ctypes.data, v.ctypes.data, 2) == 0      # a refused call stored nothing
    g2.finalize_table(); assert g2.table_size == 2
    g2.close()
    host.close()
    d = B.BBDuk("k=23 rename=t", refs=[b"AGATCGGAAGAGCACACGTCTGAACTCCAGTCAC"])
    for cap in (0, 65):                                                           # max_ids outside 1..64
        assert L.bbduk_kfilter_batch_matches(d.gpu.h, b.ctypes.data, off.ctypes.data, 2, 0, a.ctypes.data, ids.ctypes.data, fl.ctypes.data, cap,
                                             nid.ctypes.data, mi.ctypes.data, mc.ctypes.data) == -1
    assert L.bbduk_kfilter_batch_matches(d.gpu.h, b.ctypes.data, off.ctypes.data, 2, 0, a.ctypes.data, ids.ctypes.data, fl.ctypes.data, 4,
                                         None, mi.ctypes.data, mc.ctypes.data) == -1
    assert L.bbduk_kfilter_batch_matches(d.gpu.h, b.ctypes.data, off.ctypes.data, 2, 0, a.ctypes.data, ids.ctypes.data, fl.ctypes.data, 4,
                                         nid.ctypes.data, mi.ctypes.data, mc.ctypes.data) == 0
    assert nid.tolist() == [1, 0] and mi[0, 0] == 1 and mc[0, 0] == 34 - 23 + 1 and ids[0] == 1
    d.close()


# ---------------------------------------------------------------------------------------------- trimfailuresto1bp (BBDukParser.java:105-109, 774)
def tf1bp_config(rng):
    """One random configuration of every operator family with trimfailures=t on top."""
    fam = rng.choice(["r", "r", "l", "rl", "n", "f", "f", "fbm", "mcf", "ksplit", "big"])
    k = rng.choice([11, 13, 17, 21, 23, 27])
    okw, args = dict(k=k, trimFailuresTo1bp=1), ["k=%d" % k, "trimfailures=t"]
    if fam == "big":                                               # k > 31: countSetKmersBig on 31-mers
        kb = rng.randint(32, 48); okw.update(k=31, kbig=kb); args[0] = "k=%d" % kb
    if fam in ("r", "l", "rl", "n", "ksplit") and rng.random() < 0.6:
        mk = rng.randint(max(1, k - 12), k - 1); okw["mink"] = mk; args.append("mink=%d" % mk)
    if fam == "r": okw["ktrimRight"] = 1; args.append("ktrim=r")
    elif fam == "l": okw["ktrimLeft"] = 1; args.append("ktrim=l")
    elif fam == "rl": okw["ktrimRight"] = okw["ktrimLeft"] = 1; args.append("ktrim=rl")
    elif fam == "n": okw["ktrimN"] = 1; args.append("ktrim=n")
    elif fam == "ksplit": okw["ksplit"] = 1; args.append("ksplit=t")
    elif fam == "fbm": okw["findBestMatch"] = 1; args.append("fbm=t")
    elif fam == "mcf": okw["minCoveredFraction"] = 0.3; args.append("mcf=0.3")
    elif rng.random() < 0.4: v = rng.randint(1, 3); okw["maxBadKmers0"] = v; args.append("mbk=%d" % v)
    if fam in ("r", "rl") and rng.random() < 0.6: okw["trimPairsEvenly"] = 1; args.append("tpe")
    if rng.random() < 0.5: okw["hdist"] = 1; args.append("hdist=1")
    if rng.random() < 0.3: okw["requireBothBad"] = 1; args.append("rieb=f")          # (off anyway: trimfailures turns rieb off, :109)
    v = rng.choice([1, 2, 10, 10, 40, 100]); okw["minReadLength"] = v; args.append("minlen=%d" % v)
    if rng.random() < 0.3: okw["minLenFraction"] = 0.6; args.append("mlf=0.6")
    if rng.random() < 0.2: okw["forbidN"] = 1; args.append("forbidn=t")
    if rng.random() < 0.15: okw["restrictRight"] = 60; args.append("restrictright=60")     # a GENERAL instantiation
    return fam, " ".join(args), okw


@pytest.mark.gpu
@pytest.mark.parametrize("block", range(8))
def test_trimfailuresto1bp_parity(block):
    rng = random.Random(4242 + block)
    ncut = 0
    for _ in range(6):
        fam, args, okw = tf1bp_config(rng)
        refs = [util.rand_seq(rng, rng.randint(12, 90), 0.01) for _ in range(6)]
        d, o = make_pair(args, okw, refs=refs)
        reads = util.fuzz_reads(rng, refs, 500, [0, 1, 2, 3, 9, 10, 12, 22, 23, 31, 40, 64, 75, 150, 151, 300], junk=True)
        try:
            if fam == "n":
                check_kmask(d, o, reads, True); check_kmask(d, o, reads[:201], False)
            elif fam == "rl":
                check_tips(d, o, reads, True); check_tips(d, o, reads[:201], False)
            elif fam == "ksplit":
                gf = check_split(d, o, reads); assert not (gf & B.FLAG_REMOVED).any()
            else:
                check_batch(d, o, reads, True); check_batch(d, o, reads[:201], False)
            c = d.gpu.counters()
            assert c[8] == 0 and c[9] == 0 and c[6] == c[0], (args, c[:10])           # nothing leaves through outm (:1431)
            ncut += int(c[1] - c[7])
        except AssertionError as e:
            raise AssertionError("config %r: %s" % (args, str(e)[:1500]))
        d.close()
    assert ncut > 0


@pytest.mark.gpu
@pytest.mark.parametrize("args,okw", [("k=21 ktrim=r trimfailures=t minlen=100 hdist=1", dict(k=21, ktrimRight=1, trimFailuresTo1bp=1, minReadLength=100, hdist=1)),
                                      ("k=21 ktrim=l trimfailures=t minlen=60 tpe", dict(k=21, ktrimLeft=1, trimFailuresTo1bp=1, minReadLength=60, trimPairsEvenly=1)),
                                      ("k=25 trimfailures=t hdist=1", dict(k=25, trimFailuresTo1bp=1, hdist=1)),
                                      ("k=23 ktrim=r mink=11 trimfailures=t minlen=2000 tpe", dict(k=23, ktrimRight=1, mink=11, trimFailuresTo1bp=1, minReadLength=2000, trimPairsEvenly=1))])
def test_trimfailuresto1bp_units_beyond_the_main_kernel(args, okw):
    """trimfailuresto1bp on units the wave kernel cannot hold: a 5 kb pair sends the batch to the tiled kernel, a 45 kb read to the chunked
    long-read kernel; both decide through record_stage, which cuts a failed read to its first base as BBDukProcessorS.java:1464-1488 does."""
    rng = random.Random(77)
    refs = [util.rand_seq(rng, 70) for _ in range(3)]
    d, o = make_pair(args, okw, refs=refs)
    def planted(n, where):
        r = bytearray(util.rand_seq(rng, n)); piece = refs[rng.randrange(3)][5:45]
        if where >= 0: r[where:where + len(piece)] = piece
        return bytes(r)
    for big in (5000, 45000):
        reads = []
        for i in range(60):
            n = big if i in (6, 7, 20, 33) else rng.choice([30, 150, 151, 400])
            reads.append(planted(n, rng.choice([-1, 0, 20, n // 2, max(0, n - 45)])))
        check_batch(d, o, reads, True)
        check_batch(d, o, reads[:31], False)
        c = d.gpu.counters()
        assert c[8] == 0 and c[9] == 0 and c[6] == c[0]                # nothing leaves through outm (:1431)
    d.close()


TF_LONG = [("n", "k=21 ktrim=n trimfailures=t minlen=100 hdist=1", dict(k=21, ktrimN=1, trimFailuresTo1bp=1, minReadLength=100, hdist=1)),
           ("n", "k=23 ktrim=n mink=11 trimfailures=t minlen=2000 mlf=0.5", dict(k=23, ktrimN=1, mink=11, trimFailuresTo1bp=1, minReadLength=2000, minLenFraction=0.5)),
           ("rl", "k=21 ktrim=rl mink=9 trimfailures=t minlen=100 tpe hdist=1", dict(k=21, ktrimRight=1, ktrimLeft=1, mink=9, trimFailuresTo1bp=1, minReadLength=100, trimPairsEvenly=1, hdist=1)),
           ("rl", "k=25 ktrim=rl trimfailures=t minlen=4000", dict(k=25, ktrimRight=1, ktrimLeft=1, trimFailuresTo1bp=1, minReadLength=4000)),
           ("ksplit", "k=23 ksplit=t mink=11 trimfailures=t minlen=50 hdist=1", dict(k=23, ksplit=1, mink=11, trimFailuresTo1bp=1, minReadLength=50, hdist=1)),
           ("f", "k=25 fbm=t trimfailures=t hdist=1", dict(k=25, findBestMatch=1, trimFailuresTo1bp=1, hdist=1)),
           ("f", "k=40 trimfailures=t", dict(k=31, kbig=40, trimFailuresTo1bp=1)),
           ("f", "k=36 trimfailures=t mbk=2 hdist=1", dict(k=31, kbig=36, trimFailuresTo1bp=1, maxBadKmers0=2, hdist=1))]


@pytest.mark.gpu
@pytest.mark.parametrize("ci", range(len(TF_LONG)))
def test_trimfailuresto1bp_other_modes_units_beyond_the_main_kernel(ci):
    """ktrim=n / ktrim=rl / ksplit / findbestmatch / k>31 with trimfailuresto1bp on units the wave kernel cannot hold (round 5; rounds 3-4 refused
    them): a 5 kb pair sends the batch to the tiled kernels, a 45 kb read to the long-unit kernels; their record stages cut a failed read to its first
    base and evict nothing, as BBDukProcessorS.java:1431, 1464-1488 do."""
    fam, args, okw = TF_LONG[ci]
    rng = random.Random(99 + ci)
    refs = [util.rand_seq(rng, 70) for _ in range(3)]
    d, o = make_pair(args, okw, refs=refs)
    def planted(n, where):
        r = bytearray(util.rand_seq(rng, n)); piece = refs[rng.randrange(3)][5:50]
        if where >= 0: r[where:where + len(piece)] = piece
        return bytes(r)
    ncut = 0
    for big in (5000, 45000):
        reads = []
        for i in range(60):
            n = big if i in (6, 7, 20, 33) else rng.choice([1, 30, 150, 151, 400])
            reads.append(planted(n, rng.choice([-1, 0, 20, n // 2, max(0, n - 50)])))
        for paired, sub in ((True, reads), (False, reads[:31])):
            if fam == "n": check_kmask(d, o, sub, paired)
            elif fam == "rl": check_tips(d, o, sub, paired)
            elif fam == "ksplit":
                if paired: continue
                gf = check_split(d, o, sub); assert not (gf & B.FLAG_REMOVED).any()
            else: check_batch(d, o, sub, paired)
            c = d.gpu.counters()
            assert c[8] == 0 and c[9] == 0 and c[6] == c[0], (args, c[:10])         # nothing leaves through outm (:1431)
            ncut += int(c[1] - c[7])
    assert ncut > 0 or fam == "ksplit"
    d.close()


# ---- read lengths around the block borders of the scans (the stream scan's 256-position blocks run across read borders; the pair scan of a
# handle the stream kernel does not take gives a read 128 positions per block): lengths around every such edge, reference pieces ending at
# the last position of a block, the first of the next and the read's end.
TAIL_CASES = [("ktrim=r k=23 mink=11 hdist=1", dict(k=23, ktrimRight=1, mink=11, hdist=1), 23),
              ("ktrim=r k=23", dict(k=23, ktrimRight=1), 23),                                                       # hdist=0: forbidNs, the undefined-base path
              ("ktrim=r k=23 mink=11 hdist=1 tpe minlen=20", dict(k=23, ktrimRight=1, mink=11, hdist=1, trimPairsEvenly=1, minReadLength=20), 23),
              ("k=31 hdist=1", dict(k=31, hdist=1), 31), ("k=27", dict(k=27), 27),
              ("ktrim=r k=21 hdist=1 restrictright=150", dict(k=21, ktrimRight=1, hdist=1, restrictRight=150), 21),        # GENERAL: the span starts inside the read
              ("k=25 restrictleft=160 forbidn=t", dict(k=25, restrictLeft=160, forbidN=1), 25),
              ("ktrim=r k=19 mink=9 speed=3", dict(k=19, ktrimRight=1, mink=9, speed=3), 19),
              ("ktrim=r k=23 mink=11 hdist=1 trimfailuresto1bp=t", dict(k=23, ktrimRight=1, mink=11, hdist=1, trimFailuresTo1bp=1), 23)]


@pytest.mark.parametrize("ci", range(len(TAIL_CASES)))
@pytest.mark.parametrize("paired", [True, False])
def test_tail_pass_parity(ci, paired):
    args, okw, k = TAIL_CASES[ci]
    rng = random.Random(7000 + ci)
    refs = [util.rand_seq(rng, rng.randint(40, 120), 0.0) for _ in range(6)]
    d, o = make_pair(args, okw, refs=refs)
    edge = [k - 1 + 128 + t for t in (0, 1, 2, 8, 9, 16, 17, 31, 32, 33)] + [k - 1 + 256 + t for t in (1, 5, 32, 33)]   # 0 / 33: just outside the rule
    lengths = edge * 3 + [100, 150, k - 1 + 128 - 1, 40, 0, 5]
    reads = util.fuzz_reads(rng, refs, 1200, lengths, p_insert=0.5)
    for L in edge:                                              # a piece of a reference ending exactly at chosen positions of a read of length L
        for end in (L, L - 1, L - 3, k - 1 + 128, k - 1 + 128 + 1, k - 1 + 128 + 2, k - 1 + 127, max(k, L - 33)):
            if end > L or end < k + 5:
                continue
            r = rng.choice(refs)
            piece = r[rng.randint(0, len(r) - k - 5):][:k + 5]
            g = bytearray(util.rand_seq(rng, L, 0.0))
            g[end - len(piece):end] = piece
            if rng.random() < 0.3:
                g[rng.randrange(L)] = ord("N")
            reads.append(bytes(g))
    rng.shuffle(reads)
    if paired and len(reads) & 1:
        reads.pop()
    check_batch(d, o, reads, paired)
    # the same edges diluted among plain lengths
    few = util.fuzz_reads(rng, refs, 400, [150, k - 1 + 100, k - 1 + 128], p_insert=0.5) + reads[:20]
    check_batch(d, o, few, paired)
    # short reads: many per block, pieces ending near their borders
    sl = [k - 1 + t for t in (84, 84, 83, 85, 44, 45, 43, 1, 2, 60, 70, 78)] + [k - 2, 5, 0, 150]
    shorts = util.fuzz_reads(rng, refs, 1500, sl, p_insert=0.6)
    for L in (k - 1 + 84, k - 1 + 83, k - 1 + 60):
        for end in (L, L - 1, k - 1 + 44, k - 1 + 45, k - 1 + 43, k + 1, k, k - 1 + 84, k - 1 + 2):
            if end > L or end < k:
                continue
            r = rng.choice(refs)
            piece = r[rng.randint(0, len(r) - k - 3):][:k + 3]
            g = bytearray(util.rand_seq(rng, L, 0.0))
            g[max(0, end - len(piece)):end] = piece[-min(len(piece), end):]
            for rep in range(3):
                shorts.insert(rng.randrange(len(shorts) + 1), bytes(g))
    if paired and len(shorts) & 1:
        shorts.pop()
    check_batch(d, o, shorts, paired)


@pytest.mark.parametrize("ci", [0, 1, 3, 5, 6, 7])
def test_pair_scan_candidate_form_behind_the_hook(ci):
    """BBDUK_HOOK_PAIR_SCAN: the first-hit scans on bbduk_wave_kernel's pair scan (what a table beyond 2^28 buckets runs, and the A/B
    reference of the stream kernel) give the oracle's answers at the same length edges."""
    args, okw, k = TAIL_CASES[ci]
    rng = random.Random(7300 + ci)
    refs = [util.rand_seq(rng, rng.randint(40, 120), 0.0) for _ in range(6)]
    d, o = make_pair(args, okw, refs=refs, hooks={B.HOOK_PAIR_SCAN: 1})
    lengths = [k - 1 + 128 + t for t in (0, 1, 2, 17, 32, 33)] + [k - 1 + 256 + 5, 100, 150, 151, 40, 0, 5, k - 1 + 84, k - 1 + 44]
    reads = util.fuzz_reads(rng, refs, 1500, lengths, p_insert=0.5)
    check_batch(d, o, reads, True)
    check_batch(d, o, reads[:701], False)
    d.close()


@pytest.mark.parametrize("kind,args,okw", [
    ("batch", "ktrim=r k=23 mink=11 hdist=1", dict(k=23, ktrimRight=1, mink=11, hdist=1)),          # three planes per wave: units up to 2512 bases
    ("batch", "ktrim=r k=23", dict(k=23, ktrimRight=1)),                                              # forbidNs: four planes, units up to 2384
    ("batch", "ktrim=l k=23 mink=11 hdist=1", dict(k=23, ktrimLeft=1, mink=11, hdist=1)),
    ("batch", "k=27 hdist=1 mbk=5", dict(k=27, hdist=1, maxBadKmers0=5)),                             # the exact plane
    ("batch", "k=25 hdist=1 fbm", dict(k=25, hdist=1, findBestMatch=1)),
    ("mask", "ktrim=n k=23 mink=11 hdist=1", dict(k=23, ktrimN=1, mink=11, hdist=1)),
    ("split", "ksplit=t k=23 mink=11 hdist=1", dict(k=23, mink=11, hdist=1, ksplit=1))])
def test_units_at_the_planes_capacity(kind, args, okw):
    """Units (pairs; ksplit: reads) whose length sits at and around what a wave's planes hold -- the stream kernels' candidate plane is
    full to its last word there, and one base more sends the batch to the tiled kernel."""
    rng = random.Random(7500 + len(args))
    refs = [util.rand_seq(rng, rng.randint(40, 120), 0.0) for _ in range(6)]
    d, o = make_pair(args, okw, refs=refs)
    for cap in (2384, 2512):                                        # WUNIT_MAX_KM, WUNIT_MAX
        for delta in (-3, -1, 0, 1):
            unit = cap + delta
            L1 = unit if kind == "split" else unit // 2
            L2 = unit - L1
            lens = [L1] if kind == "split" else [L1, L2]
            reads = []
            for i in range(40):                                     # long units among short ones, pieces of the references at both ends
                if i % 4 == 0:
                    for L in lens:
                        g = bytearray(util.rand_seq(rng, L, 0.001))
                        for at in (0, L - 60, rng.randrange(max(1, L - 60))):
                            r = rng.choice(refs); g[at:at + min(len(r), 50)] = r[:min(len(r), 50)]
                        reads.append(bytes(g))
                else:
                    reads += util.fuzz_reads(rng, refs, 1 if kind == "split" else 2, [150, 100, 31], p_insert=0.5)
            if kind == "batch":
                check_batch(d, o, reads, True)
            elif kind == "mask":
                check_kmask(d, o, reads, True)
            else:
                check_split(d, o, reads)
    d.close()


@pytest.mark.parametrize("kind,args,okw", [
    ("batch", "ktrim=l k=23 mink=11 hdist=1", dict(k=23, ktrimLeft=1, mink=11, hdist=1)),
    ("batch", "k=25 mbk=2", dict(k=25, maxBadKmers0=2)),
    ("batch", "k=21 hdist=1 mkf=0.1", dict(k=21, hdist=1, minKmerFraction=0.1)),
    ("mask", "ktrim=n k=23 mink=11 hdist=1", dict(k=23, ktrimN=1, mink=11, hdist=1)),
    ("split", "ksplit=t k=23 mink=11 hdist=1", dict(k=23, mink=11, hdist=1, ksplit=1))])
def test_pair_scan_every_hit_behind_the_hook(kind, args, okw):
    """BBDUK_HOOK_PAIR_SCAN for the every-hit scans: bbduk_wave_kernel's pair scan (their path for maps beyond 2^28 buckets and for qskip)."""
    rng = random.Random(7400 + len(args))
    k = okw["k"]
    refs = [util.rand_seq(rng, rng.randint(40, 120), 0.0) for _ in range(6)]
    d, o = make_pair(args, okw, refs=refs, hooks={B.HOOK_PAIR_SCAN: 1})
    lengths = [k - 1 + 128 + t for t in (0, 1, 2, 17, 32, 33)] + [k - 1 + 256 + 5, 100, 150, 151, 40, 0, 5]
    reads = util.fuzz_reads(rng, refs, 1200, lengths, p_insert=0.5)
    if kind == "batch":
        check_batch(d, o, reads, True)
        check_batch(d, o, reads[:501], False)
    elif kind == "mask":
        check_kmask(d, o, reads, True)
    else:
        check_split(d, o, reads)
    d.close()


# the every-hit scans at the block borders of the pair scan (round 2 had a tail pass there; they run the stream scan now)
TAIL_HITS_CASES = [("ktrim=l k=23 mink=11 hdist=1", dict(k=23, ktrimLeft=1, mink=11, hdist=1), 23, "batch"),
                   ("ktrim=l k=21", dict(k=21, ktrimLeft=1), 21, "batch"),
                   ("ktrim=l k=19 hdist=1 restrictleft=170 tp=1", dict(k=19, ktrimLeft=1, hdist=1, restrictLeft=170, trimPad=1), 19, "batch"),
                   ("ksplit=t k=23 mink=11 hdist=1", dict(k=23, mink=11, hdist=1, ksplit=1), 23, "split"),
                   ("ksplit=t k=25", dict(k=25, ksplit=1), 25, "split")]


@pytest.mark.parametrize("ci", range(len(TAIL_HITS_CASES)))
def test_tail_pass_every_hit_scans(ci):
    args, okw, k, kind = TAIL_HITS_CASES[ci]
    rng = random.Random(7100 + ci)
    refs = [util.rand_seq(rng, rng.randint(40, 120), 0.0) for _ in range(6)]
    d, o = make_pair(args, okw, refs=refs)
    edge = [k - 1 + 128 + t for t in (0, 1, 2, 8, 9, 16, 17, 31, 32, 33)] + [k - 1 + 256 + t for t in (1, 5, 32, 33)]
    reads = util.fuzz_reads(rng, refs, 1000, edge * 3 + [100, 150, 40, 0, 5], p_insert=0.6)
    for L in edge:
        for end in (L, L - 1, L - 3, k - 1 + 128, k - 1 + 128 + 1, k - 1 + 128 + 2, k - 1 + 127, max(k, L - 33)):
            if end > L or end < k + 5:
                continue
            r = rng.choice(refs)
            piece = r[rng.randint(0, len(r) - k - 5):][:k + 5]
            g = bytearray(util.rand_seq(rng, L, 0.0))
            g[end - len(piece):end] = piece
            if rng.random() < 0.5:                              # and a second piece far to the left: first / last hit in different passes
                g[30:30 + len(piece)] = piece
            if rng.random() < 0.3:
                g[rng.randrange(L)] = ord("N")
            reads.append(bytes(g))
    rng.shuffle(reads)
    if len(reads) & 1:
        reads.pop()
    if kind == "split":
        check_split(d, o, reads)
    else:
        check_batch(d, o, reads, True); check_batch(d, o, reads[:301], False)


@pytest.mark.parametrize("ci", range(4))
def test_tail_pass_kmask(ci):
    args, okw, k = [("ktrim=n k=23 mink=11 hdist=1", dict(k=23, ktrimN=1, mink=11, hdist=1), 23), ("ktrim=n k=21", dict(k=21, ktrimN=1), 21),
                    ("ktrim=n k=23 mink=11 hdist=1 mfc=t", dict(k=23, ktrimN=1, mink=11, hdist=1, kmaskFullyCovered=1), 23),
                    ("ktrim=n k=19 tp=2 restrictright=160 forbidn=t mfc=t", dict(k=19, ktrimN=1, trimPad=2, restrictRight=160, forbidN=1, kmaskFullyCovered=1), 19)][ci]
    rng = random.Random(7200 + ci)
    refs = [util.rand_seq(rng, rng.randint(40, 120), 0.0) for _ in range(6)]
    d, o = make_pair(args, okw, refs=refs)
    edge = [k - 1 + 128 + t for t in (0, 1, 2, 8, 9, 16, 17, 31, 32, 33)] + [k - 1 + 256 + t for t in (1, 5, 32, 33)]
    reads = util.fuzz_reads(rng, refs, 800, edge * 3 + [100, 150, 40, 0, 5], p_insert=0.6)
    for L in edge:
        for end in (L, L - 1, L - 3, k - 1 + 128, k - 1 + 128 + 1, k - 1 + 128 + 2, k - 1 + 127, max(k, L - 33)):
            if end > L or end < k + 5:
                continue
            r = rng.choice(refs)
            piece = r[rng.randint(0, len(r) - k - 5):][:k + 5]
            g = bytearray(util.rand_seq(rng, L, 0.0))
            g[end - len(piece):end] = piece
            if rng.random() < 0.3:
                g[rng.randrange(L)] = ord("N")
            reads.append(bytes(g))
    rng.shuffle(reads)
    if len(reads) & 1:
        reads.pop()
    check_kmask(d, o, reads, True); check_kmask(d, o, reads[:301], False)


# ---- query-side expansion tabulated at build time (round 5: KParams::qx, qx_rewrite)
QX = [("ktrim=r k=21 mink=11 hdist=1 qhdist=1", dict(k=21, mink=11, hdist=1, qhdist=1, ktrimRight=1)),
      ("ktrim=r k=23 mink=9 qhdist=1", dict(k=23, mink=9, qhdist=1, ktrimRight=1)),                         # hdist=0: forbidNs stays on (BBDukParser.java:150), the rolling rkmer restarts at an N
      ("ktrim=l k=19 mink=8 hdist=1 qhdist=1 tp=1", dict(k=19, mink=8, hdist=1, qhdist=1, ktrimLeft=1, trimPad=1)),
      ("k=25 mm=f qhdist=1", dict(k=25, maskMiddle=0, qhdist=1)),
      ("k=24 mm=f hdist=1 qhdist=1 mbk=2 rieb=f", dict(k=24, maskMiddle=0, hdist=1, qhdist=1, maxBadKmers0=2, requireBothBad=1)),
      ("ktrim=r k=21 mink=11 qhdist2=1", dict(k=21, mink=11, qhdist2=1, ktrimRight=1)),                     # the short k-mers alone are expanded
      ("ktrim=r k=21 mink=11 qhdist=1 qhdist2=0 restrictright=80", dict(k=21, mink=11, qhdist=1, qhdist2=0, ktrimRight=1, restrictRight=80)),
      ("ktrim=n k=17 mink=7 qhdist=1 hdist=1", dict(k=17, mink=7, qhdist=1, hdist=1, ktrimN=1)),
      ("ktrim=rl k=21 mink=10 qhdist=1", dict(k=21, mink=10, qhdist=1, ktrimLeft=1, ktrimRight=1)),
      ("ksplit=t k=21 mink=10 qhdist=1", dict(k=21, mink=10, qhdist=1, ksplit=1)),
      ("k=22 mm=f fbm=t qhdist=1", dict(k=22, maskMiddle=0, findBestMatch=1, qhdist=1)),
      # round 6: with the middle mask (the default kfilter): keyed by the masked forward k-mer, every filling of the masked base(s) evaluated
      ("k=31 qhdist=1", dict(k=31, qhdist=1)), ("k=24 hdist=1 qhdist=1 mbk=1", dict(k=24, hdist=1, qhdist=1, maxBadKmers0=1)),
      ("ktrim=l k=23 qhdist=1", dict(k=23, qhdist=1, ktrimLeft=1)), ("ktrim=n k=25 qhdist=1", dict(k=25, qhdist=1, ktrimN=1)),
      ("k=27 qhdist=1 mkf=0.05", dict(k=27, qhdist=1, minKmerFraction=0.05))]


@pytest.mark.gpu
@pytest.mark.parametrize("ci", range(len(QX)))
def test_query_expansion_tabulated(ci):
    """qhdist = 1 where the answer depends on kmer alone (no middle mask, rcomp=t, speed=0): bbduk finalizes such a handle by tabulating getValue over every
    forward k-mer within one substitution of a stored key and the kernels look THAT map up, one lookup per window; a window whose rolling rkmer is
    not kmer's reverse complement (an undefined base inside: N reads as A in kmer and as 0 in rkmer, and under forbidNs rkmer restarts) takes its direct
    lookup in the reference's own map.  Reads dense in Ns and junk, pieces with 0-2 substitutions, both builds, every mode; ids (the (j, i) order
    of the neighbour loop decides which scaffold is credited) and counters included."""
    args, okw = QX[ci]
    rng = random.Random(51000 + ci)
    base = util.rand_seq(rng, 300)
    refs = [base, util.rand_seq(rng, 200) + base[50:120], util.revcomp(base[150:260]), util.rand_seq(rng, 90, 0.02), util.rand_seq(rng, 14), util.rand_seq(rng, 400)]
    # two scaffolds one substitution apart: which one a neighbour lookup credits depends on the loop order
    twin = bytearray(refs[5][100:160]); twin[30] = ord("A") if twin[30] != ord("A") else ord("C"); refs.append(bytes(twin))
    for build in ("host", "device"):
        d = B.BBDuk(args, refs=refs, build=build)
        o = Oracle(**okw)
        for r in refs: o.add_ref(r)
        assert d.gpu.table_layout & 8, d.gpu.table_layout
        assert d.gpu.table_size == o.stored_kmers
        ks, vs = o.dump_pairs()
        assert np.array_equal(d.gpu.table_lookup(ks), vs)
        reads = util.fuzz_reads(rng, refs, 600, [0, 1, 7, 8, 9, 17, 19, 20, 21, 22, 25, 30, 40, 75, 150, 151, 300], junk=True)
        for i in range(500):
            n = rng.choice([60, 100, 150, 151])
            r = bytearray(util.rand_seq(rng, n))
            src = refs[rng.choice([0, 1, 2, 5, 6])]; cut = rng.randint(8, 50); a = rng.randrange(len(src) - cut); piece = bytearray(src[a:a + cut])
            if rng.random() < 0.5: piece = bytearray(util.revcomp(bytes(piece)))
            for _ in range(rng.choice([0, 1, 1, 2, 3])): piece[rng.randrange(len(piece))] = rng.choice(b"ACGTNN.")
            w = rng.choice([0, n - cut, rng.randrange(n - cut + 1)]); r[w:w + cut] = piece
            for _ in range(rng.choice([0, 0, 1, 2])): r[rng.randrange(n)] = ord("N")
            reads.append(bytes(r))
        if len(reads) % 2: reads.pop()
        longs = [util.rand_seq(rng, 3000) + refs[0][20:80], util.rand_seq(rng, 150), refs[5][10:70] + util.rand_seq(rng, 45000), util.rand_seq(rng, 100)]
        if okw.get("ksplit"):
            check_split(d, o, reads); check_split(d, o, longs)
        elif okw.get("ktrimN"):
            check_kmask(d, o, reads, True); check_kmask(d, o, reads[:301], False); check_kmask(d, o, longs, True)
        elif okw.get("ktrimLeft") and okw.get("ktrimRight"):
            check_tips(d, o, reads, True); check_tips(d, o, reads[:301], False); check_tips(d, o, longs, True)
        else:
            check_batch(d, o, reads, True); check_batch(d, o, reads[:301], False); check_batch(d, o, longs, True)
        d.close()


# ---- unit routing at scale (round 6).  Since round 5 every unit of ktrim=r|l and kfilter goes to the kernel that holds it
# (KParams::route: the wave kernel up to wunitMax bases, the tiled kernel up to CAP_BASES - 64, the chunked long-unit kernel beyond) instead of the whole
# batch following its longest unit.  A unit's result never depends on its neighbours (BBDukProcessorS.java:948-1093: one pair per iteration), so
# a batch of 2 M configs[1]-shaped reads with long units planted at the first pair, the last pair, tile and mini-tile borders and back to back must
# give, read for read and counter for counter, (i) what the same handle gives with routing off (BBDUK_HOOK_FORCE_TILE) and (ii) what the oracle gives.
ROUTE = [("ktrim=r k=23 mink=11 hdist=1 ref=adapters", dict(k=23, mink=11, hdist=1, ktrimRight=1), util.ADAPTERS),
         ("k=31 hdist=1 ref=phix", dict(k=31, hdist=1), util.PHIX),
         ("ktrim=l k=23 mink=11 ref=adapters", dict(k=23, mink=11, ktrimLeft=1), util.ADAPTERS)]


@pytest.mark.parametrize("ci", range(len(ROUTE)))
@pytest.mark.parametrize("paired", [True, False])
def test_unit_routing_at_scale(ci, paired):
    args, okw, ref = ROUTE[ci]
    d, o = make_pair(args, okw, ref_fasta=ref)
    phix = B.read_fasta(util.PHIX)[0][1]
    kw = dict(contam=phix, contam_frac=0.01) if "phix" in args else {}
    n_pairs = 1_000_000
    b, off = B.synth_generate_host(B.synth_params(2, **kw), 0, n_pairs)
    rng = random.Random(77 + ci)
    piece = (lambda: phix[rng.randrange(len(phix) - 200):][:rng.randint(31, 200)]) if "phix" in args else (lambda: rng.choice([B.TRUSEQ_R1, B.TRUSEQ_R2, util.revcomp(B.TRUSEQ_R1)]))

    def long_read(L):
        g = bytearray(util.rand_seq(rng, L, 0.0005))
        for _ in range(rng.randint(0, 3)):
            r = piece()[:max(12, L // 2)]
            pos = rng.choice([0, L - len(r), rng.randrange(L - len(r))])
            g[pos:pos + len(r)] = r
        if rng.random() < 0.4:
            g[L - 14:] = B.TRUSEQ_R1[:14]                          # a short k-mer at the very end (mink)
        return bytes(g)
    # where the long units go (pair indices): the first and the last pair, the borders of the 256-read tiles and of the 62-read mini-tiles (both sides), back to back
    spots = [0, 1, n_pairs - 1, n_pairs - 2, 127, 128, 129, 30, 31, 32, 61, 62, 63, 128 * 1000 - 1, 128 * 1000, 31 * 4001, 31 * 4001 + 1, 31 * 4001 + 2, 500_000, 500_001, 500_002, 500_003]
    spots += [rng.randrange(n_pairs) for _ in range(40 - len(spots))]
    spots = sorted(set(spots))
    lens = [2400, 2513, 2600, 3000, 5000, 9000, 20000, 40800, 40928, 41000, 45000, 60000]      # around wunitMax (2512) and CAP_BASES - 64 (40896), and well beyond
    planted = {}
    for i, p in enumerate(spots):
        L = lens[i % len(lens)]
        mate = long_read(rng.choice([100, 150, L])) if i % 3 else long_read(150)
        planted[p] = (long_read(L), mate) if i % 2 else (mate, long_read(L))
    # splice: the reads of the planted pairs are replaced, everything else stays where the generator put it
    parts, lengths, last = [], [], 0
    rl = np.diff(off)
    for p in spots:
        parts.append(b[off[last * 2]:off[2 * p]]); lengths.append(rl[2 * last:2 * p])
        for r in planted[p]:
            parts.append(np.frombuffer(r, np.uint8)); lengths.append(np.array([len(r)], np.int64))
        last = p + 1
    parts.append(b[off[2 * last]:]); lengths.append(rl[2 * last:])
    nb = np.concatenate(parts); noff = np.concatenate([[0], np.cumsum(np.concatenate(lengths))]).astype(np.int64)
    assert len(noff) == 2 * n_pairs + 1 and noff[-1] == len(nb)
    d.gpu.reset_counters(); o.reset_counters()
    ga, gi, gf = d.gpu.process_batch(nb, noff, paired)
    gc = d.gpu.counters().copy()
    # (ii) the oracle, every read
    oa, oi, of = o.process_batch(nb, noff, paired, nthreads=16)
    bad = np.nonzero((ga != oa) | (gi != oi) | (gf != of))[0]
    assert len(bad) == 0, [(int(i), int(noff[i + 1] - noff[i]), (int(ga[i]), int(gi[i]), int(gf[i])), (int(oa[i]), int(oi[i]), int(of[i]))) for i in bad[:5]]
    assert np.array_equal(gc, o.counters())
    longs = np.nonzero(np.diff(noff) > 2000)[0]
    assert len(longs) >= 40 and (ga[longs] > 0).sum() > 5          # the planted units are there, and some of them hit
    # ... and from page-locked buffers: a large call then goes through in pieces of whole pairs, each piece's upload under the piece before's kernel (host_batch)
    import torch
    pb = torch.from_numpy(nb).pin_memory().numpy(); po = torch.from_numpy(noff).pin_memory().numpy()
    d.gpu.reset_counters()
    pa, pi, pf = d.gpu.process_batch(pb, po, paired)
    assert np.array_equal(pa, ga) and np.array_equal(pi, gi) and np.array_equal(pf, gf)
    assert np.array_equal(d.gpu.counters(), gc)
    # (i) the same handle with routing off: every unit through the tiled / long-unit kernels
    d.gpu.test_hook(B.HOOK_FORCE_TILE, 1)
    d.gpu.reset_counters()
    ta, ti, tf = d.gpu.process_batch(nb, noff, paired)
    d.gpu.test_hook(B.HOOK_FORCE_TILE, 0)
    assert np.array_equal(ta, ga) and np.array_equal(ti, gi) and np.array_equal(tf, gf)
    assert np.array_equal(d.gpu.counters(), gc)
    d.close()


# ---- round 6: the read-aligned form of the stream scans (bbduk_stream_scan.inc: one read per slot pair where L > 128 and L - k + 1 <= 128) is taken only by
# batches whose sub-tiles hold ONE read length: every read of a batch is cut or padded to L (lengths on both sides of the form's limits, and short ones that keep
# the stream form), reference pieces land at the first window, the last window and anywhere between, with undefined bases.
def _uniform_reads(rng, refs, n, L):
    out = []
    for _ in range(n):
        g = bytearray(util.rand_seq(rng, L, 0.004))
        t = rng.random()
        if t < 0.6 and L >= 8:
            r = rng.choice(refs)
            piece = r[(a := rng.randrange(0, max(1, len(r) - 4))):a + rng.randint(4, max(4, min(len(r) - a, L)))]
            piece = piece[:L]
            pos = rng.choice([0, L - len(piece), rng.randrange(0, L - len(piece) + 1)])
            g[pos:pos + len(piece)] = piece
            if rng.random() < 0.2: g[rng.randrange(L)] = ord("N")
        out.append(bytes(g))
    return out


ALIGNED_LENGTHS = [36, 40, 50, 51, 64, 75, 76, 100, 125, 129, 140, 150, 151, 152, 153]
ALIGNED_CFGS = [("fuzz", c) for c in FUZZ[:6]] + [("kmask", KMASK[4]), ("kmask", KMASK[5]), ("tips", TIPS[0]), ("tips", TIPS[1]), ("split", SPLIT[0]), ("split", SPLIT[1])]


@pytest.mark.gpu
@pytest.mark.parametrize("ci", range(len(ALIGNED_CFGS)))
def test_read_aligned_forms(ci):
    kind, (args, okw) = ALIGNED_CFGS[ci]
    rng = random.Random(9000 + ci)
    refs = [util.rand_seq(rng, rng.randint(30, 200), 0.005) for _ in range(6)]
    d, o = make_pair(args, okw, refs=refs)
    for L in ALIGNED_LENGTHS:
        reads = _uniform_reads(rng, refs, 500, L)
        if kind == "kmask":
            check_kmask(d, o, reads, True)
        elif kind == "tips":
            check_tips(d, o, reads, True)
        elif kind == "split":
            check_split(d, o, reads)
        else:
            check_batch(d, o, reads, True)
            check_batch(d, o, reads[:251], False)
    d.close()
