"""Shared helpers for the tests: config table, golden loader, random read factory."""
import json
import os
import random

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
GOLDEN = os.path.join(ROOT, "tests", "golden")
ADAPTERS = os.path.join(ROOT, "data", "adapters.fa")
PHIX = os.path.join(ROOT, "data", "phix2.fa.gz")

# (bbduk.sh-style args for the product, keyword args for the oracle, reference fasta)
CONFIGS = {
    "c1": ("ktrim=r k=23 hdist=0 ref=adapters", dict(k=23, ktrimRight=1), ADAPTERS),
    "c2": ("ktrim=r k=23 mink=11 hdist=1 ref=adapters", dict(k=23, mink=11, hdist=1, ktrimRight=1), ADAPTERS),
    "c3": ("k=31 hdist=1 ref=phix", dict(k=31, hdist=1), PHIX),
    "c4s": ("k=31 hdist=0 ref=phix", dict(k=31), PHIX),
    "c5": ("ktrim=r k=23 hdist=1 ref=adapters", dict(k=23, hdist=1, ktrimRight=1), ADAPTERS),
    "c2tpe": ("ktrim=r k=23 mink=11 hdist=1 tpe ref=adapters", dict(k=23, mink=11, hdist=1, ktrimRight=1, trimPairsEvenly=1), ADAPTERS),
    "c3q": ("k=31 hdist=1 qskip=2 speed=3 ref=phix", dict(k=31, hdist=1, qSkip=2, speed=3), PHIX),
    "c3mkf": ("k=31 hdist=1 mkf=0.2 ref=phix", dict(k=31, hdist=1, minKmerFraction=0.2), PHIX),
    "c3mcf": ("k=31 hdist=1 mcf=0.5 ref=phix", dict(k=31, hdist=1, minCoveredFraction=0.5), PHIX),
}


def golden_names():
    return sorted(f[:-5] for f in os.listdir(GOLDEN) if f.endswith(".json"))


def load_golden(name):
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        return json.load(f)


def oracle_kwargs_from_spec(spec_args):
    return {k: (int(v) if isinstance(v, bool) else v) for k, v in spec_args.items()}


def rand_seq(rng, n, pn=0.0):
    return bytes(rng.choice(b"ACGT") if rng.random() >= pn else ord("N") for _ in range(n))


def revcomp(s: bytes) -> bytes:
    return bytes(reversed(s.translate(bytes.maketrans(b"ACGTacgt", b"TGCAtgca"))))


def mutate(rng, s, p):
    s = bytearray(s)
    for i in range(len(s)):
        if rng.random() < p:
            s[i] = rng.choice(b"ACGT")
    return bytes(s)


def fuzz_reads(rng, refs, n, lengths, p_insert=0.7, p_n=0.02, junk=False):
    """Random reads, a fraction carrying (mutated, possibly reverse-complemented, possibly truncated) pieces
    of the references at random positions; occasional lower case / non-ACGTN bytes when junk=True."""
    reads = []
    for _ in range(n):
        L = rng.choice(lengths)
        g = bytearray(rand_seq(rng, L, p_n))
        if L > 12 and refs and rng.random() < p_insert:
            r = rng.choice(refs)
            if rng.random() < 0.3:
                r = revcomp(r)
            r = mutate(rng, r, 0.03)
            pos = rng.randint(0, L - 1)
            off = rng.randint(0, max(0, len(r) - 12)) if rng.random() < 0.3 else 0
            m = min(len(r) - off, L - pos)
            g[pos:pos + m] = r[off:off + m]
        if L > 0 and rng.random() < 0.2:
            g[rng.randrange(L)] = ord("N")
        if junk and L > 0 and rng.random() < 0.3:
            for _ in range(rng.randint(1, 3)):
                g[rng.randrange(L)] = rng.choice(b"acgtunRYKM.-*X\x00\x7f\x80\xff")
        reads.append(bytes(g))
    return reads


def long_refs_and_reads(rng):
    refs = [rand_seq(rng, rng.randint(40, 220), 0.005) for _ in range(5)]
    refs.append(refs[0][:60] + refs[1][-50:])                   # scaffolds sharing k-mers: ties and first-writer-wins ids
    reads = fuzz_reads(rng, refs, 30, [0, 1, 12, 30, 31, 32, 40, 41, 60, 100, 150, 151, 250, 300], junk=True)
    for _ in range(10):                                          # chimeras: pieces of two or three scaffolds in one read
        parts = []
        for _ in range(rng.randint(2, 3)):
            r = rng.choice(refs)
            a = rng.randrange(len(r)); parts.append(r[a:a + rng.randint(20, 90)]); parts.append(rand_seq(rng, rng.randint(0, 25)))
        reads.append(b"".join(parts))
    return refs, reads
