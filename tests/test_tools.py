"""tools/compare_with_reference.py (the comparator of the JVM-side pinning kit, tools/verify_with_reference.sh) on outputs
synthesised from the fixtures themselves: what a BBDuk that agrees with the expectations would write passes, a single wrong
trim length, a misrouted pair or a missing read is reported."""
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
GOLD = os.path.join(ROOT, "tests", "golden")
TOOL = os.path.join(ROOT, "tools", "compare_with_reference.py")


def _fastq(path):
    with open(path) as f:
        lines = f.read().split("\n")
    return [(lines[i][1:], lines[i + 1], lines[i + 3]) for i in range(0, len(lines) - 1, 4)]


def _write(path, recs):
    with open(path, "w") as f:
        for n, b, q in recs:
            f.write("@%s\n%s\n+\n%s\n" % (n, b, q))


def _run(args):
    return subprocess.run([sys.executable, TOOL] + args, capture_output=True, text=True)


def _split(tsv, srcs, kfilter):
    src = {n: (b, q) for f in srcs for n, b, q in _fastq(os.path.join(GOLD, f))}
    rows = [l.rstrip("\n").split("\t") for l in open(os.path.join(GOLD, tsv))]
    kept, removed = [], []
    for name, L, a, sid, newlen, flags in rows:
        b, q = src[name]
        n = int(L) if kfilter else int(newlen)
        (removed if int(flags) & 2 else kept).append((name, b[:n], q[:n]))
    return src, rows, kept, removed


def test_comparator_accepts_agreeing_output_and_reports_differences(tmp_path):
    out, outm = str(tmp_path / "o.fq"), str(tmp_path / "m.fq")
    # ktrim fixture: lengths after trimming
    src, rows, kept, removed = _split("cli_c2.tsv", ("cli_c2_r1.fq", "cli_c2_r2.fq"), False)
    _write(out, kept); _write(outm, removed)
    tsv = os.path.join(GOLD, "cli_c2.tsv")
    r = _run(["--tsv", tsv, "--out", out, "--outm", outm])
    assert r.returncode == 0 and "0 differences" in r.stdout, r.stdout + r.stderr
    assert any(int(x[2]) > 0 for x in rows)                               # the fixture really trims
    i = next(j for j, x in enumerate(kept) if len(x[1]) < len(src[x[0]][0]))
    bad = list(kept); n, b, q = bad[i]; bad[i] = (n, src[n][0][:len(b) + 1], src[n][1][:len(b) + 1])   # one base too many
    _write(out, bad)
    r = _run(["--tsv", tsv, "--out", out, "--outm", outm])
    assert r.returncode == 1 and "1 differences" in r.stdout and n in r.stdout
    # kfilter fixture: routing between out= and outm=
    src, rows, kept, removed = _split("cli_c3.tsv", ("cli_c3_interleaved.fq",), True)
    assert len(removed) > 10 and len(kept) > 10
    tsv = os.path.join(GOLD, "cli_c3.tsv")
    _write(out, kept); _write(outm, removed)
    r = _run(["--kfilter", "--tsv", tsv, "--out", out, "--outm", outm])
    assert r.returncode == 0 and "0 differences" in r.stdout, r.stdout + r.stderr
    _write(out, kept + removed[:1]); _write(outm, removed[2:])            # one removed read routed to out=, one dropped altogether
    r = _run(["--kfilter", "--tsv", tsv, "--out", out, "--outm", outm])
    assert r.returncode == 1 and "expected in outm" in r.stdout and "missing" in r.stdout


def test_fixture_names_follow_the_addadapters_convention():
    """<initial>_<remaining> (jgi/AddAdapters.java:485) + a unique tail; `remaining` is the generator's truth, so a trimmed length
    that equals it is a correct trim -- the spec's ktrim=r answer lands on it, or up to a few bases short of it (with hdist=1 a k-mer that
    starts one to three genome bases early still matches the adapter), for nearly every read with an adapter long enough to see."""
    rows = [l.rstrip("\n").split("\t") for l in open(os.path.join(GOLD, "cli_c2.tsv"))]
    agree = total = 0
    for name, L, a, sid, newlen, flags in rows:
        initial, remaining = (int(x) for x in name.split(" ")[0].split("_"))
        assert initial == int(L) and 0 <= remaining <= initial
        if remaining < initial - 11:                                       # at least mink adapter bases in the read
            total += 1; agree += -5 <= int(newlen) - remaining <= 0 or (remaining == 0 and int(newlen) <= 1)
    assert total > 20 and agree >= 0.95 * total, (agree, total)


def test_seal_comparator_on_synthesised_output(tmp_path):
    """tools/compare_seal_with_reference.py: what a Seal that agrees with tests/golden/seal_all.tsv would write (pattern=, outu=) passes; a pair in
    the wrong scaffold's file or missing from outu is reported."""
    tool = os.path.join(ROOT, "tools", "compare_seal_with_reference.py")
    recs = _fastq(os.path.join(GOLD, "seal_reads.fq"))
    pair = {recs[i][0]: (recs[i], recs[i + 1]) for i in range(0, len(recs), 2)}
    names = ["scafA", "scafB", "scafC", "scafD", "scafE", "scafF"]
    files = {n: [] for n in names}; unmatched = []
    for line in open(os.path.join(GOLD, "seal_all.tsv")):
        name, sites, assigned, mx, who, flags = line.rstrip("\n").split("\t")
        if who:
            for n in who.split(","):
                files[n] += list(pair[name])
        else:
            unmatched += list(pair[name])
    for n in names:
        _write(str(tmp_path / ("s_%s.fq" % n)), files[n])
    _write(str(tmp_path / "u.fq"), unmatched)
    args = ["--tsv", os.path.join(GOLD, "seal_all.tsv"), "--pattern", str(tmp_path / "s_%.fq"), "--outu", str(tmp_path / "u.fq"), "--names"] + names
    r = subprocess.run([sys.executable, tool] + args, capture_output=True, text=True)
    assert r.returncode == 0 and "identical" in r.stdout, r.stdout + r.stderr
    assert any(len(v) for v in files.values()) and unmatched
    moved = files["scafA"][:2]                                   # one pair into the wrong file
    _write(str(tmp_path / "s_scafA.fq"), files["scafA"][2:]); _write(str(tmp_path / "s_scafE.fq"), files["scafE"] + moved)
    r = subprocess.run([sys.executable, tool] + args, capture_output=True, text=True)
    assert r.returncode == 1 and "DIFFERENCES" in r.stdout and moved[0][0] in r.stdout


def test_documented_cases_export_and_check(tmp_path):
    """tools/doc_cases.py (section 5b of the pinning kit): the exported inputs, and the checker on output synthesised from the C oracle's answers --
    what bbduk.sh would write if it behaves as its guide says -- and on a perturbed copy."""
    import importlib.util
    import numpy as np
    from oracle.oracle_ffi import Oracle, pack_reads
    from tests.test_documented_behaviour import CASES
    spec = importlib.util.spec_from_file_location("doc_cases", os.path.join(ROOT, "tools", "doc_cases.py"))
    dc = importlib.util.module_from_spec(spec); spec.loader.exec_module(dc)
    d = str(tmp_path)
    names = dc.export(d)
    assert names == [c[0] for c in CASES] and all(os.path.exists(os.path.join(d, n + ".args")) for n in names)
    for name, args, okw, refs, reads, paired, want_a, want_fl in CASES:
        o = Oracle(**okw)
        for r in refs:
            o.add_ref(r)
        b, off = pack_reads(reads)
        a, ids, fl = o.process_batch(b, off, paired)
        ktrim = bool(okw.get("ktrimRight") or okw.get("ktrimLeft"))
        out, outm = [], []
        for i, rd in enumerate(reads):
            n = max(1, len(rd) - int(a[i])) if (ktrim and a[i]) else len(rd)
            (outm if fl[i] & 2 else out).append("@r%d\n%s\n+\n%s\n" % (i, rd[:n].decode(), "I" * n))
        po, pm = os.path.join(d, name + ".o.fq"), os.path.join(d, name + ".m.fq")
        open(po, "w").write("".join(out)); open(pm, "w").write("".join(outm))
        assert dc.check(d, name, po, pm), name
        open(po, "w").write("".join(outm)); open(pm, "w").write("".join(out))          # routed the other way round: must be reported
        assert not dc.check(d, name, po, pm), name
