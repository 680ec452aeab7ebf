"""
Generates the Seal fixtures of this directory from oracle/seal_oracle.c (the C restatement of jgi/Seal.java's k-mer path):
  seal_refs.fa             six scaffolds; two share a 200-base segment (ambiguous k-mers), one is the reverse complement of a piece of another
  seal_reads.fq            interleaved pairs named <pair index>_<true scaffold> /1|/2, drawn from the scaffolds (1 % substitutions, some N),
                           chimeras and unrelated pairs
  seal_<mode>.tsv          per pair: name, sites, assigned, max hits, the assigned scaffold NAMES (comma separated), flags
  seal_<mode>_stats.txt    what `stats=` prints (Seal.writeStats, jgi/Seal.java:899-950, statscolumns=5) without the #File line
for ambig=all / first / toss / random (k=31 mm=t, Seal's defaults otherwise).  tools/verify_with_reference.sh runs the REAL seal.sh over the
same files on a machine with a JVM; tests/test_seal.py replays them through the HIP operator.
Run from the repo root:  python tests/golden/make_seal_golden.py
"""
import os
import random
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from oracle.seal_ffi import SealOracle  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
MODES = {"all": 1, "first": 0, "toss": 3, "random": 2}


def revcomp(s):
    return s.translate(bytes.maketrans(b"ACGT", b"TGCA"))[::-1]


def build():
    rng = random.Random(20260929)
    seq = lambda n: bytes(rng.choice(b"ACGT") for _ in range(n))
    shared = seq(200)
    refs = [seq(700), seq(300) + shared + seq(300), seq(250) + shared + seq(250), seq(900), seq(500), b""]
    refs[5] = seq(200) + revcomp(refs[3][100:400]) + seq(200)
    names = ["scafA", "scafB", "scafC", "scafD", "scafE", "scafF"]      # B and C share a segment; F holds the reverse complement of a piece of D
    reads, rnames = [], []
    for p in range(240):
        kind = rng.random()
        if kind < 0.12:
            r1, r2, src = seq(150), seq(150), "none"
        else:
            s = rng.randrange(6)
            ref = refs[s]
            ins = rng.randint(160, 400)
            a = rng.randrange(0, len(ref) - ins)
            frag = ref[a:a + ins]
            r1, r2, src = bytearray(frag[:150]), bytearray(revcomp(frag)[:150]), names[s].split()[0]
            if kind < 0.25:                                        # a chimera: mate 2 from another scaffold
                t = rng.randrange(6); b = rng.randrange(0, len(refs[t]) - 150)
                r2 = bytearray(refs[t][b:b + 150]); src += "+" + names[t].split()[0]
            for r in (r1, r2):
                for i in range(len(r)):
                    if rng.random() < 0.01:
                        r[i] = rng.choice(b"ACGTN")
            if kind > 0.9:
                r1 = r1[:rng.choice([8, 40, 90])]                  # short mates (minlength=10 removes the pair at 8)
            r1, r2 = bytes(r1), bytes(r2)
        reads += [r1, r2]
        rnames += ["%d_%s /1" % (p, src), "%d_%s /2" % (p, src)]
    return refs, names, reads, rnames


def stats_text(o, names, counters):
    c, reads, bases, frags, ambig = counters
    rows = [(names[i - 1], int(reads[i]), int(bases[i]), int(ambig[i])) for i in range(1, len(names) + 1) if reads[i] > 0]
    rows.sort(key=lambda r: (-r[2], -r[1], r[0]))                   # StringCount.compareTo: bases desc, reads desc, name asc
    rmult = 100.0 / (c["readsIn"] if c["readsIn"] > 0 else 1); bmult = 100.0 / (c["basesIn"] if c["basesIn"] > 0 else 1)
    out = ["#Total\t%d\t%d" % (c["readsIn"], c["basesIn"]),
           "#Matched\t%d\t%.5f%%\t%d" % (c["readsMatched"], rmult * c["readsMatched"], c["basesMatched"]),
           "#Name\tReads\tReadsPct\tBases\tBasesPct\tAmbigReads"]
    out += ["%s\t%d\t%.5f%%\t%d\t%.5f%%\t%d" % (n, r, r * rmult, b, b * bmult, a) for n, r, b, a in rows]
    return "\n".join(out) + "\n"


def main():
    refs, names, reads, rnames = build()
    with open(os.path.join(HERE, "seal_refs.fa"), "w") as f:
        for n, r in zip(names, refs):
            f.write(">%s\n%s\n" % (n, r.decode()))
    with open(os.path.join(HERE, "seal_reads.fq"), "w") as f:
        for n, r in zip(rnames, reads):
            f.write("@%s\n%s\n+\n%s\n" % (n, r.decode(), "I" * len(r)))
    for mode, am in MODES.items():
        o = SealOracle(refs, ambigMode=am)
        sites, assigned, mx, ids, flags = o.process_reads(reads, True, 0, max_ids=8)
        with open(os.path.join(HERE, "seal_%s.tsv" % mode), "w") as f:
            for u in range(0, len(reads), 2):
                who = ",".join(names[i - 1].split()[0] for i in ids[u][:assigned[u]])
                f.write("%s\t%d\t%d\t%d\t%s\t%d\n" % (rnames[u], sites[u], assigned[u], mx[u], who, flags[u]))
        with open(os.path.join(HERE, "seal_%s_stats.txt" % mode), "w") as f:
            f.write(stats_text(o, [n.split()[0] for n in names], o.counters()))
        c = o.counters()[0]
        print(mode, "pairs", len(reads) // 2, "matched reads", c["readsMatched"], "unmatched", c["readsUnmatched"], "qfiltered", c["readsQFiltered"],
              "ambiguous pairs", int((sites[0::2] > 1).sum()))


if __name__ == "__main__":
    main()
