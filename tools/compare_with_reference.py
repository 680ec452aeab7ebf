#!/usr/bin/env python3
"""
Compares what the REAL bbduk.sh wrote for one of this repository's fixture inputs with the expectation committed beside it
(tests/golden/cli_*.tsv: name, length, result, scaffold id, length after trimming, flags; flags bit 1 = the pair left through outm).
Plain Python 3, no dependencies; called by tools/verify_with_reference.sh on a machine that has a JDK and a BBTools checkout.

  compare_with_reference.py --tsv cli_c2.tsv --out o1.fq [o2.fq] --outm m1.fq [m2.fq] [--kfilter]

Checks, read by read (matched by the full FASTQ header, which the fixtures keep unique):
  * a read is in out= iff its expected flags say "kept", in outm= iff they say "removed";
  * its length in the reference's output equals the expected length after trimming (kfilter: its original length);
  * nothing is missing and nothing extra was written.
Exit status 0 = identical, 1 = differences (listed).
"""
import argparse
import sys


def read_fastq(paths):
    recs = {}
    for p in paths:
        with open(p) as f:
            lines = f.read().split("\n")
        for i in range(0, len(lines) - 1, 4):
            if not lines[i].startswith("@"):
                raise SystemExit("%s: line %d is not a FASTQ header" % (p, i + 1))
            name = lines[i][1:]
            if name in recs:
                raise SystemExit("%s: duplicate read name %r" % (p, name))
            recs[name] = len(lines[i + 1])
    return recs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tsv", required=True)
    ap.add_argument("--out", nargs="+", default=[])
    ap.add_argument("--outm", nargs="*", default=[])
    ap.add_argument("--kfilter", action="store_true", help="reads keep their length (the result column counts k-mer hits)")
    a = ap.parse_args()
    kept = read_fastq(a.out); removed = read_fastq(a.outm)
    bad = []
    n = 0
    with open(a.tsv) as f:
        for line in f:
            name, L, res, sid, newlen, flags = line.rstrip("\n").split("\t")
            n += 1
            want_len = int(L) if a.kfilter else int(newlen)
            want_removed = bool(int(flags) & 2)
            have = removed if want_removed else kept
            other = kept if want_removed else removed
            if name in other:
                bad.append("%s: expected in %s, the reference wrote it to %s" % (name, "outm" if want_removed else "out", "out" if want_removed else "outm"))
            elif name not in have:
                bad.append("%s: missing from the reference's output" % name)
            elif have[name] != want_len:
                bad.append("%s: length %d in the reference's output, %d expected" % (name, have[name], want_len))
    extra = (set(kept) | set(removed)) - {l.split("\t")[0] for l in open(a.tsv)}
    bad += ["%s: written by the reference, not in the fixture" % x for x in sorted(extra)]
    print("%s: %d reads, %d differences" % (a.tsv, n, len(bad)))
    for b in bad[:50]:
        print("  " + b)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
