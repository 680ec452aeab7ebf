"""CPU tests of the product's host side: the C-ABI library loads and exports every declared symbol, the
C++ mirror of BBDukParser derives the same constants as the oracle, its index builder yields the same
key->id map, and the synthetic generator is deterministic.  No compute calls that need a GPU."""
import ctypes as C
import os
import random
import re

import numpy as np
import pytest

from bbtools_amd import bbduk as B
from oracle.oracle_ffi import Oracle
from tests import util


def test_library_exports_every_declared_symbol():
    L = B.lib()
    assert L.bbduk_abi_version() == 2
    for s in B.GPU_SYMBOLS + B.HOST_SYMBOLS:
        assert hasattr(L, s), s
    # and the list itself matches the headers
    decl = set()
    for hdr in ("bbduk_gpu.h", "bbduk_host.h", "bbduk_test_hooks.h"):
        txt = open(os.path.join(util.ROOT, "include", hdr)).read()
        decl |= set(re.findall(r"\b(bbduk_[a-z0-9_]+)\s*\(", txt))
    decl -= {"bbduk_params", "bbduk_synth_params", "bbduk_handle", "bbduk_host"}
    assert decl == set(B.GPU_SYMBOLS + B.HOST_SYMBOLS)


def test_params_struct_layout_matches_header():
    assert C.sizeof(B.Params) == 136      # static_assert'ed on the C side (bbduk_hip.hip)
    assert B.Params.middleMask.offset == 32


@pytest.mark.parametrize("cfg", list(util.CONFIGS))
def test_host_mirror_matches_oracle(cfg):
    args, okw, ref = util.CONFIGS[cfg]
    h = B.HostIndex(args)
    h.load_refs()
    n = h.build_index()
    o = Oracle(**okw)
    o.load_fasta(ref)
    p = h.params(0)
    for name, field in (("k", "k"), ("mink", "mink"), ("forbidNs", "forbidNs"), ("minlen", "minlen"), ("minlen2", "minlen2"),
                        ("middleMask", "middleMask"), ("qhdist", "qhdist"), ("qhdist2", "qhdist2")):
        assert getattr(p, field) == o.constant(name), name
    assert p.numScaffolds == o.num_scaffolds == h.num_scaffolds
    hk, hv = h.pairs()
    ok, ov = o.dump_pairs()
    order = np.argsort(ok)
    assert n == len(ok) == o.stored_kmers
    assert np.array_equal(hk, ok[order]) and np.array_equal(hv, ov[order])


def test_host_index_random_refs_vs_oracle():
    rng = random.Random(77)
    for args, okw in (("k=9 hdist=2 mm=f", dict(k=9, hdist=2, maskMiddle=0)),
                      ("ktrim=l k=13 mink=6 hdist=1 hdist2=0", dict(k=13, mink=6, hdist=1, hdist2=0, ktrimLeft=1)),
                      ("k=24 hdist=1", dict(k=24, hdist=1)), ("k=12 rcomp=f", dict(k=12, rcomp=0)),
                      ("k=25 mm=3", dict(k=25, midMaskLen=3)), ("k=13 rskip=3", dict(k=13, minSkip=3, maxSkip=3)),
                      ("k=11 edist=1", dict(k=11, edist=1)), ("ktrim=r k=12 mink=6 edist=1 mm=f", dict(k=12, mink=6, edist=1, ktrimRight=1, maskMiddle=0)),
                      ("k=13 hdist=2 edist=1 rcomp=f", dict(k=13, hdist=2, edist=1, rcomp=0)), ("ktrim=l k=10 mink=5 hdist=1 edist2=1", dict(k=10, mink=5, hdist=1, edist2=1, ktrimLeft=1)),
                      ("ktrim=r k=13 mink=6 hdist=1 minskip=2 maxskip=4", dict(k=13, mink=6, hdist=1, ktrimRight=1, minSkip=2, maxSkip=4))):
        refs = [util.rand_seq(rng, rng.randint(3, 90), 0.02) for _ in range(8)] + [b"acgtunACGTUNacgtacgtacgtaacc"]
        h = B.HostIndex(args); o = Oracle(**okw)
        for r in refs:
            h.add_ref(r); o.add_ref(r)
        h.build_index()
        hk, hv = h.pairs(); ok, ov = o.dump_pairs(); order = np.argsort(ok)
        assert np.array_equal(hk, ok[order]) and np.array_equal(hv, ov[order]), args


def test_parser_errors_like_the_reference():
    with pytest.raises(B.BBDukError, match="Unknown parameter"):
        B.HostIndex("k=23 bogus=1")                       # BBDukParser.java:870-872
    with pytest.raises(B.BBDukError):
        B.HostIndex("k=23 mink=11")                       # mink needs a ktrim mode (:301)
    p = B.HostIndex("k=40").params()                      # k>31: k=31, kbig=40, maskMiddle off before minlen2 (:164-165, 237-243)
    assert (p.k, p.kbig, p.middleMask, p.minlen2, p.mode) == (31, 40, -1, 31, B.MODE_KFILTER)
    p = B.HostIndex("k=40 ktrim=r").params()              # trimming modes reduce K to 31 (:207-215): maskMiddle stays on
    assert (p.k, p.kbig, p.minlen2, p.mode) == (31, 0, 15, B.MODE_KTRIM_R) and p.middleMask == ~(3 << 30)
    p = B.HostIndex("k=35 speed=3").params()              # so do speed / qskip (:217-223)
    assert (p.k, p.kbig) == (31, 0)
    with pytest.raises(B.BBDukError, match="findBestMatch"):
        B.HostIndex("k=40 fbm=t")                         # :299
    with pytest.raises(B.BBDukError):
        B.HostIndex("k=25 fbm=t mbk=2")                   # thread-history dependent in the reference: refused
    assert B.HostIndex("k=25 rename=t").params().findBestMatch == 1        # rename implies findbestmatch (:153)
    with pytest.raises(B.BBDukError, match="rename"):
        B.HostIndex("k=25 rename=t mcf=0.5")                              # :1049-1052 re-runs findBestMatch on covered reads: refused
    p = B.HostIndex("k=25 fbm").params()
    assert (p.findBestMatch, p.mode) == (1, B.MODE_KFILTER)
    assert B.HostIndex("k=25 fbm ktrim=r").params().findBestMatch == 0     # only the filtering branch consults it (:1072)
    p = B.HostIndex("k=23 mink=11 ksplit=t").params()     # ksplit is a trimming mode: mink allowed (:301)
    assert p.mode == B.MODE_KSPLIT
    assert B.HostIndex("k=23 ksplit=t ktrim=r").params().mode == B.MODE_KTRIM_R    # a later ktrim= clears it (:604-620)
    with pytest.raises(B.BBDukError):
        B.HostIndex("k=23 ksplit tp=2")
    assert B.HostIndex("ktrim=n k=23 mfc=t").params().kmaskFullyCovered == 1
    assert B.HostIndex("ktrim=r k=23 kmaskfullycovered=t").params().kmaskFullyCovered == 0     # only kmask() consults it
    h = B.HostIndex("ktrim=r")                            # k defaults to 27 (:163)
    assert h.params().k == 27
    h = B.HostIndex("ktrim=r k=23 mkh=3 rieb=f mlf=0.5 minlen=20 tp=1")
    p = h.params()
    assert (p.maxBadKmers, p.removePairsIfEitherBad, p.minReadLength, p.trimPad) == (2, 0, 20, 1)
    assert abs(p.minLenFraction - 0.5) < 1e-7


def test_create_fails_loudly_without_gpu_or_bad_params():
    h = B.HostIndex("ktrim=r k=23")
    p = h.params(0)
    p.k = 40
    with pytest.raises(B.BBDukError):
        B.BBDukGpu(p)


def test_synth_generator_is_deterministic_and_shaped():
    sp = B.synth_params(2)
    b1, off1 = B.synth_generate_host(sp, 0, 50)
    b2, _ = B.synth_generate_host(sp, 10, 20)
    assert np.array_equal(b1[10 * 300:30 * 300], b2)      # any slice regenerates identically
    assert off1[-1] == len(b1) == 50 * 300 and np.all(np.diff(off1) == 150)
    assert set(np.unique(b1).tolist()) <= set(b"ACGTN")
    # pairs are reverse complements over the insert when nothing was mutated: check one with a long insert
    sp0 = B.synth_params(7, n_rate=0.0, sub_rate=0.0, ins_min=150, ins_max=150)
    b, _ = B.synth_generate_host(sp0, 0, 3)
    for p in range(3):
        r1, r2 = bytes(b[p * 300:p * 300 + 150]), bytes(b[p * 300 + 150:p * 300 + 300])
        assert r2 == util.revcomp(r1)
    # adapter read-through appears right after the insert
    sp1 = B.synth_params(7, n_rate=0.0, sub_rate=0.0, ins_min=100, ins_max=100)
    b, _ = B.synth_generate_host(sp1, 0, 2)
    assert bytes(b[100:150]) == B.TRUSEQ_R1[:50] and bytes(b[250:300]) == B.TRUSEQ_R2[:50]


def test_pack_bases_host_layout():
    """Packed boundary format: 2-bit codes in base order (16 per uint32), one undefined bit per base, tail of the last
    undefined word set.  Host-only C loop (no device call)."""
    rng = np.random.default_rng(5)
    for nb in (0, 1, 15, 16, 17, 33, 1000):
        raw = rng.choice(np.frombuffer(b"ACGTacgtUuNnRY.-*\x00\xff", np.uint8), nb)
        codes, undef = B.pack_bases_host(raw)
        assert len(codes) == (nb + 15) // 16 and len(undef) == (nb + 31) // 32
        lut = np.full(256, -1, np.int64)
        for ch, v in zip(b"ACGTUacgtu", [0, 1, 2, 3, 3] * 2):
            lut[ch] = v
        want = lut[raw]
        for b in range(nb):
            u = (int(undef[b >> 5]) >> (b & 31)) & 1
            assert u == (want[b] < 0)
            if not u:
                assert (int(codes[b >> 4]) >> (2 * (b & 15))) & 3 == want[b]
        for b in range(nb, 32 * len(undef)):
            assert (int(undef[b >> 5]) >> (b & 31)) & 1 == 1


def test_scaffold_names_and_lengths():
    """scaffoldNames / scaffoldLengths as BBDukLoader records them: FASTA headers, literals named by their id."""
    h = B.HostIndex("k=23 ref=adapters literal=ACGTACGTACGTACGTACGTACGTACGT,GGGGGGGGGGGGGGGGGGGGGGGGGGGGG")
    h.load_refs()
    fa = B.read_fasta(util.ADAPTERS)
    assert h.num_scaffolds == len(fa) + 2 + 1
    for i, (name, seq) in enumerate(fa):
        assert h.scaffold_info(i + 1) == (name, len(seq))
    assert h.scaffold_info(len(fa) + 1) == (str(len(fa) + 1), 28)
    assert h.scaffold_info(len(fa) + 2) == (str(len(fa) + 2), 29)
    with pytest.raises(B.BBDukError):
        h.scaffold_info(0)
