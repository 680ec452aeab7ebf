#!/usr/bin/env python3
"""The documented examples of tests/test_documented_behaviour.py (hand-derived from BBDukGuide.txt / bbduk.sh) as input files for the REAL
bbduk.sh, and the check of what it wrote (tools/verify_with_reference.sh, section 6).

  doc_cases.py export DIR            writes, per case: NAME.fa (references), NAME.fq (reads r0, r1, ...; interleaved when paired), NAME.args
                                     (the flags) and NAME.expect.tsv (read, kept|removed, length it must have or -)
  doc_cases.py check DIR NAME OUT.fq OUTM.fq      exit status 0 = the tool routed and trimmed every read as the guide says
"""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)


def expectations():
    from tests.test_documented_behaviour import CASES
    for name, args, okw, refs, reads, paired, want_a, want_fl in CASES:
        ktrim = "ktrim=" in args
        rows = []
        for i, rd in enumerate(reads):
            removed = bool(want_fl[i] & 2)
            if ktrim and want_a is not None:
                length = max(1, len(rd) - want_a[i]) if want_a[i] else len(rd)        # TrimRead leaves at least one base
            else:
                length = len(rd)                                                       # kfilter does not cut
            rows.append(("r%d" % i, "removed" if removed else "kept", length))
        yield name, args, refs, reads, paired, rows


def export(d):
    os.makedirs(d, exist_ok=True)
    names = []
    for name, args, refs, reads, paired, rows in expectations():
        with open(os.path.join(d, name + ".fa"), "w") as f:
            for i, r in enumerate(refs):
                f.write(">ref%d\n%s\n" % (i + 1, r.decode()))
        with open(os.path.join(d, name + ".fq"), "w") as f:
            for i, r in enumerate(reads):
                f.write("@r%d\n%s\n+\n%s\n" % (i, r.decode(), "I" * len(r)))
        with open(os.path.join(d, name + ".args"), "w") as f:
            f.write(args + (" int=t" if paired else " int=f") + "\n")
        with open(os.path.join(d, name + ".expect.tsv"), "w") as f:
            for row in rows:
                f.write("%s\t%s\t%d\n" % row)
        names.append(name)
    return names


def read_fq(path):
    out = {}
    if not os.path.exists(path):
        return out
    with open(path) as f:
        lines = f.read().split("\n")
    for i in range(0, len(lines) - 3, 4):
        out[lines[i][1:].split()[0]] = len(lines[i + 1])
    return out


def check(d, name, out_fq, outm_fq):
    kept, removed = read_fq(out_fq), read_fq(outm_fq)
    bad = []
    with open(os.path.join(d, name + ".expect.tsv")) as f:
        for line in f:
            rd, where, length = line.split("\t"); length = int(length)
            got = kept if where == "kept" else removed
            other = removed if where == "kept" else kept
            if rd in other or rd not in got:
                bad.append("%s: expected %s" % (rd, where))
            elif got[rd] != length:
                bad.append("%s: %d bases, expected %d" % (rd, got[rd], length))
    print("  doc case %-34s %s" % (name, "as documented" if not bad else "DIFFERS: " + "; ".join(bad)))
    return not bad


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "export":
        print("\n".join(export(sys.argv[2])))
    elif len(sys.argv) == 6 and sys.argv[1] == "check":
        sys.exit(0 if check(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5]) else 1)
    else:
        sys.exit(__doc__)
