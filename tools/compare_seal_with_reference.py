#!/usr/bin/env python3
"""
Compares what the REAL seal.sh wrote for tests/golden/seal_reads.fq with the expectation committed beside it (tests/golden/seal_<mode>.tsv:
pair name, sites, assigned, max hits, assigned scaffold names, flags).  Plain Python 3; called by tools/verify_with_reference.sh.

  compare_seal_with_reference.py --tsv seal_all.tsv --pattern 'work/seal_all_%.fq' --outu work/seal_all_u.fq --names scafA scafB ...

Checks: every pair sits in exactly the per-scaffold files (pattern=) its expectation names -- or in outu= when it names none -- and nothing
extra was written.  Exit status 0 = identical.
"""
import argparse
import os
import sys


def names_in(path):
    if not os.path.exists(path):
        return set()
    with open(path) as f:
        lines = f.read().split("\n")
    return {lines[i][1:] for i in range(0, len(lines) - 1, 4) if lines[i].endswith("/1")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tsv", required=True); ap.add_argument("--pattern", required=True); ap.add_argument("--outu", required=True)
    ap.add_argument("--names", nargs="+", required=True)
    a = ap.parse_args()
    got = {n: names_in(a.pattern.replace("%", n)) for n in a.names}
    unmatched = names_in(a.outu)
    bad = 0
    for line in open(a.tsv):
        name, sites, assigned, mx, who, flags = line.rstrip("\n").split("\t")
        want = set(who.split(",")) if who else set()
        for n in a.names:
            if (name in got[n]) != (n in want):
                print("pair %r: expected in %s, the reference wrote it to %s" % (name, sorted(want), sorted(m for m in a.names if name in got[m]))); bad += 1
                break
        if (name in unmatched) != (not want):
            print("pair %r: outu membership differs (expected %s)" % (name, "unmatched" if not want else "matched")); bad += 1
    print("%s: %s" % (os.path.basename(a.tsv), "identical" if not bad else "%d DIFFERENCES" % bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
