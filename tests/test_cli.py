"""bbduk_cli: the non-JVM caller of SURVEY 8(b), pinned by fixture FASTQs and the spec's per-read TSV
(tests/golden/cli_*.{fq,tsv}, written by tests/golden/make_golden.py from oracle/spec.py)."""
import json, os
import subprocess

import pytest

from bbtools_amd import _build

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
GOLD = os.path.join(ROOT, "tests", "golden")


def run_cli(args, cwd=ROOT):
    # A plain timeout that fails.  (Rounds 3-4 re-ran a devices=0,0,0 call that exceeded two minutes: "one run in ~150 hangs".  Round 5 looked inside with
    # bbduk_cli's watchdog= stack dump: the process sat in dlopen("librccl.so.1") -- 573 MB, cold page cache on a fresh box -- not in a deadlock; a group
    # on one device no longer loads the collective library at all, and 2 000 consecutive runs are clean: profiles/r05_hang_hunt.txt.)
    # (the tests name the path they mean: without deviceingest= the CLI would pick the device pipeline wherever it can, see test_cli_picks_the_device_pipeline_by_itself)
    if not any(a.startswith("deviceingest") or a == "auto" for a in args):
        args = args + ["deviceingest=f"]
    return subprocess.run([_build.cli_path()] + [a for a in args if a != "auto"], cwd=cwd, capture_output=True, text=True, timeout=300)


def test_cli_is_built_and_prints_usage():
    _build.build_cli()
    r = run_cli([])
    assert r.returncode == 1 and "usage: bbduk_cli" in r.stderr


def test_cli_rejects_unknown_flag_like_the_reference_parser():
    r = run_cli(["in=" + os.path.join(GOLD, "cli_c2_r1.fq"), "ktrim=r", "k=23", "notaflag=1", "ref=adapters"])
    assert r.returncode == 1 and "bad arguments" in r.stderr and "notaflag" in r.stderr


def _read_fastq(path):
    with open(path) as f:
        lines = f.read().split("\n")
    return [(lines[i][1:], lines[i + 1], lines[i + 3]) for i in range(0, len(lines) - 1, 4)]


@pytest.mark.gpu
def test_cli_ktrim_two_files_matches_spec_tsv(tmp_path):
    tsv, out, outm = (str(tmp_path / n) for n in ("o.tsv", "clean.fq", "removed.fq"))
    r = run_cli(["in=" + os.path.join(GOLD, "cli_c2_r1.fq"), "in2=" + os.path.join(GOLD, "cli_c2_r2.fq"), "tsv=" + tsv, "out=" + out,
                 "outm=" + outm, "batch=64", "ktrim=r", "k=23", "mink=11", "hdist=1", "ref=adapters"])
    assert r.returncode == 0, r.stderr
    assert open(tsv).read() == open(os.path.join(GOLD, "cli_c2.tsv")).read()
    rows = [l.split("\t") for l in open(tsv).read().splitlines()]
    src = {}
    for f in ("cli_c2_r1.fq", "cli_c2_r2.fq"):
        src.update({n: (b, q) for n, b, q in _read_fastq(os.path.join(GOLD, f))})
    kept = {n: (b, q) for n, b, q in _read_fastq(out)}
    removed = {n: (b, q) for n, b, q in _read_fastq(outm)}
    for name, L, a, _id, new_len, flags in rows:
        dst = removed if int(flags) & 2 else kept
        b, q = dst[name]
        assert b == src[name][0][:int(new_len)] and q == src[name][1][:int(new_len)] and int(new_len) == int(L) - int(a)
    assert len(kept) + len(removed) == len(rows)
    trimmed = sum(1 for x in rows if int(x[2]) > 0)
    assert "KTrimmed:" in r.stderr and "Input:" in r.stderr and trimmed > 0


@pytest.mark.gpu
def test_cli_kfilter_interleaved_matches_spec_tsv(tmp_path):
    tsv = str(tmp_path / "o.tsv")
    r = run_cli(["in=" + os.path.join(GOLD, "cli_c3_interleaved.fq"), "int=t", "tsv=" + tsv, "k=31", "hdist=1", "ref=phix", "devicebuild=t"])
    assert r.returncode == 0, r.stderr
    assert open(tsv).read() == open(os.path.join(GOLD, "cli_c3.tsv")).read()
    assert "Contaminants:" in r.stderr


@pytest.mark.gpu
def test_cli_kmask_masks_bases_like_the_spec(tmp_path):
    tsv, out = str(tmp_path / "o.tsv"), str(tmp_path / "masked.fq")
    r = run_cli(["in=" + os.path.join(GOLD, "cli_c2_r1.fq"), "in2=" + os.path.join(GOLD, "cli_c2_r2.fq"), "tsv=" + tsv, "out=" + out,
                 "batch=50", "ktrim=n", "k=23", "mink=11", "hdist=1", "ref=adapters"])
    assert r.returncode == 0, r.stderr
    assert open(tsv).read() == open(os.path.join(GOLD, "cli_kmask.tsv")).read()
    assert open(out).read() == open(os.path.join(GOLD, "cli_kmask_expected.fq")).read()
    assert "KMasked:" in r.stderr


@pytest.mark.gpu
def test_cli_k40_counts_runs_of_31mers(tmp_path):
    tsv = str(tmp_path / "o.tsv")
    r = run_cli(["in=" + os.path.join(GOLD, "cli_c3_interleaved.fq"), "int=t", "tsv=" + tsv, "k=40", "hdist=1", "ref=phix"])
    assert r.returncode == 0, r.stderr
    assert open(tsv).read() == open(os.path.join(GOLD, "cli_c3_k40.tsv")).read()


@pytest.mark.gpu
def test_cli_ksplit_trims_or_splits_like_the_spec(tmp_path):
    tsv, out, outm = (str(tmp_path / n) for n in ("o.tsv", "kept.fq", "split.fq"))
    r = run_cli(["in=" + os.path.join(GOLD, "cli_ksplit.fq"), "tsv=" + tsv, "out=" + out, "outm=" + outm, "batch=5",
                 "ksplit=t", "k=23", "mink=11", "hdist=1", "ref=adapters"])
    assert r.returncode == 0, r.stderr
    assert open(tsv).read() == open(os.path.join(GOLD, "cli_ksplit.tsv")).read()
    assert open(out).read() == open(os.path.join(GOLD, "cli_ksplit_expected_out.fq")).read()
    assert open(outm).read() == open(os.path.join(GOLD, "cli_ksplit_expected_outm.fq")).read()
    r = run_cli(["in=" + os.path.join(GOLD, "cli_c2_r1.fq"), "in2=" + os.path.join(GOLD, "cli_c2_r2.fq"), "ksplit=t", "k=23", "ref=adapters"])
    assert r.returncode == 1 and "unpaired" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("chunk", ["4096", "20000", "268435456"])
def test_cli_deviceingest_equals_the_host_path(tmp_path, chunk):
    """deviceingest=t: FASTQ text -> GPU (record splitting, packing, matching, writing).  Same TSV as the spec, same out= /
    outm= bytes as the host-parser path, also when records straddle chunk borders."""
    f1, f2 = os.path.join(GOLD, "cli_c2_r1.fq"), os.path.join(GOLD, "cli_c2_r2.fq")
    flags = ["ktrim=r", "k=23", "mink=11", "hdist=1", "ref=adapters"]
    base = {}
    for mode in ("host", "dev"):
        tsv, out, outm = (str(tmp_path / (mode + n)) for n in ("o.tsv", "clean.fq", "removed.fq"))
        extra = ["deviceingest=t", "chunk=" + chunk] if mode == "dev" else []
        r = run_cli(["in=" + f1, "in2=" + f2, "tsv=" + tsv, "out=" + out, "outm=" + outm] + flags + extra)
        assert r.returncode == 0, r.stderr
        base[mode] = (open(tsv).read(), open(out, "rb").read(), open(outm, "rb").read(), [l for l in r.stderr.splitlines() if "reads" in l])
    assert base["dev"] == base["host"]
    assert base["dev"][0] == open(os.path.join(GOLD, "cli_c2.tsv")).read()
    # interleaved kfilter, ktrim=l, k=40 through the same pipeline
    tsv = str(tmp_path / "c3.tsv")
    r = run_cli(["in=" + os.path.join(GOLD, "cli_c3_interleaved.fq"), "int=t", "tsv=" + tsv, "k=31", "hdist=1", "ref=phix", "deviceingest=t", "chunk=" + chunk])
    assert r.returncode == 0, r.stderr
    assert open(tsv).read() == open(os.path.join(GOLD, "cli_c3.tsv")).read()
    r = run_cli(["in=" + os.path.join(GOLD, "cli_c3_interleaved.fq"), "int=t", "tsv=" + tsv, "k=40", "hdist=1", "ref=phix", "deviceingest=t", "chunk=" + chunk])
    assert r.returncode == 0 and open(tsv).read() == open(os.path.join(GOLD, "cli_c3_k40.tsv")).read()
    outs = {}
    for mode in ("host", "dev"):
        out = str(tmp_path / (mode + "_l.fq"))
        r = run_cli(["in=" + f1, "out=" + out, "ktrim=l", "k=23", "mink=11", "ref=adapters"] + (["deviceingest=t", "chunk=" + chunk] if mode == "dev" else []))
        assert r.returncode == 0, r.stderr
        outs[mode] = open(out, "rb").read()
    assert outs["dev"] == outs["host"] and len(outs["dev"]) > 1000


@pytest.mark.gpu
@pytest.mark.parametrize("chunk", ["6000", "268435456"])
def test_cli_deviceingest_kmask_and_tips(tmp_path, chunk):
    """ktrim=n (symbol N, a custom symbol, lower case) and ktrim=rl through the device pipeline: the masked / trimmed FASTQ and the
    TSV equal what the host path writes."""
    f1, f2 = os.path.join(GOLD, "cli_c2_r1.fq"), os.path.join(GOLD, "cli_c2_r2.fq")
    for flags in (["ktrim=n", "k=23", "mink=11", "hdist=1"], ["ktrim=X", "k=23", "mink=11", "hdist=1"], ["kmask=lc", "k=21", "hdist=1"],
                  ["ktrim=n", "k=23", "mfc=t"], ["ktrim=rl", "k=23", "mink=11", "hdist=1", "tpe"], ["ktrimtips=60", "k=19", "mink=9"]):
        got = {}
        for mode in ("host", "dev"):
            tsv, out, outm = (str(tmp_path / (mode + n)) for n in ("o.tsv", "o.fq", "m.fq"))
            r = run_cli(["in=" + f1, "in2=" + f2, "tsv=" + tsv, "out=" + out, "outm=" + outm, "ref=adapters"] + flags + (["deviceingest=t", "chunk=" + chunk] if mode == "dev" else []))
            assert r.returncode == 0, (flags, r.stderr)
            got[mode] = (open(tsv).read(), open(out, "rb").read(), open(outm, "rb").read(), [l for l in r.stderr.splitlines() if "reads" in l])
        assert got["dev"] == got["host"], flags
    assert got["dev"][0] != "" and b"@" in got["dev"][1]
    r = run_cli(["in=" + f1, "in2=" + f2, "tsv=" + str(tmp_path / "k.tsv"), "out=" + str(tmp_path / "k.fq"), "ktrim=n", "k=23", "mink=11", "hdist=1", "ref=adapters", "deviceingest=t", "chunk=" + chunk])
    assert r.returncode == 0 and open(str(tmp_path / "k.fq")).read() == open(os.path.join(GOLD, "cli_kmask_expected.fq")).read()
    assert open(str(tmp_path / "k.tsv")).read() == open(os.path.join(GOLD, "cli_kmask.tsv")).read()


@pytest.mark.gpu
@pytest.mark.parametrize("chunk", ["4096", "5001", "20000", "3000000", ""])
def test_cli_deviceingest_pipelined(tmp_path, chunk):
    """One plain input file and no tsv=: the pipelined form (reader threads, uploads, operator and writer overlapped; pieces cut at fixed file
    offsets, tails carried on the device).  Its out= / outm= bytes and counter lines equal the serial form's (pipeline=f) and the host parser's,
    for piece sizes that cut records and pairs anywhere, and for a file of several pieces read by several threads each."""
    big = tmp_path / "big.fq"
    one = open(os.path.join(GOLD, "cli_c3_interleaved.fq"), "rb").read() + open(os.path.join(GOLD, "cli_c2_r1.fq"), "rb").read() + open(os.path.join(GOLD, "cli_c2_r2.fq"), "rb").read()
    big.write_bytes(one * (1 + 12_000_000 // len(one) if chunk in ("3000000", "") else 1))
    ck = ["chunk=" + chunk] if chunk else []
    for flags in (["ktrim=r", "k=23", "mink=11", "hdist=1", "ref=adapters", "int=t"], ["k=31", "hdist=1", "ref=phix", "int=t"], ["ktrim=n", "k=23", "mink=11", "hdist=1", "ref=adapters"],
                  ["ktrim=rl", "k=23", "mink=11", "hdist=1", "ref=adapters", "int=t", "tpe"], ["ktrim=l", "k=23", "mink=11", "ref=adapters", "readthreads=3", "writethreads=2"]):
        got = {}
        for mode, extra in (("host", []), ("serial", ["deviceingest=t", "pipeline=f"] + ck), ("piped", ["deviceingest=t", "timeline=t"] + ck)):
            if mode == "host" and len(one) * 2 < os.path.getsize(big) and flags[0] != "ktrim=r":
                continue                                          # (the host parser on the 12 MB file: once is enough)
            out, outm = (str(tmp_path / (mode + n)) for n in ("o.fq", "m.fq"))
            r = run_cli(["in=" + str(big), "out=" + out, "outm=" + outm] + flags + extra)
            assert r.returncode == 0, (flags, mode, r.stderr)
            got[mode] = (open(out, "rb").read(), open(outm, "rb").read(), [l for l in r.stderr.splitlines() if "reads" in l and not l.startswith("{")])
            if mode == "piped":
                tl = json.loads([l for l in r.stderr.splitlines() if l.startswith('{"timeline"')][0])["timeline"]
                assert tl["file_bytes"] == os.path.getsize(big) and tl["out_bytes"] == len(got[mode][0]) + len(got[mode][1])
                assert tl["pieces"] == -(-tl["file_bytes"] // tl["chunk"])
        assert got["piped"] == got["serial"], (flags, chunk)
        if "host" in got:
            assert got["piped"] == got["host"], (flags, chunk)
        assert len(got["piped"][0]) > 1000
    # what the serial form refuses, with the same words
    cut = tmp_path / "cut.fq"
    cut.write_bytes(one[:len(one) // 2 - 7])
    for extra in (["pipeline=f"], []):
        r = run_cli(["in=" + str(cut), "k=23", "ref=adapters", "deviceingest=t"] + ck + extra)
        assert r.returncode == 1 and ("truncated FASTQ record" in r.stderr or "malformed FASTQ record" in r.stderr), r.stderr
    odd = tmp_path / "odd.fq"
    odd.write_bytes(b"".join(b"@r%d\nACGTACGTACGTACGTACGTACGTACGTAC\n+\nIIIIIIIIIIIIIIIIIIIIIIIIIIIIII\n" % i for i in range(301)))
    for extra in (["pipeline=f"], []):
        r = run_cli(["in=" + str(odd), "int=t", "k=23", "ref=adapters", "deviceingest=t"] + ck + extra)
        assert r.returncode == 1 and "unpaired" in r.stderr, r.stderr
    empty = tmp_path / "empty.fq"
    empty.write_bytes(b"")
    r = run_cli(["in=" + str(empty), "k=23", "ref=adapters", "deviceingest=t", "out=" + str(tmp_path / "e.fq")] + ck)
    assert r.returncode == 0 and os.path.getsize(tmp_path / "e.fq") == 0 and "Input:" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("chunk", ["4096", "7001", "3000000", ""])
def test_cli_deviceingest_pipelined_two_files(tmp_path, chunk):
    """in= + in2= through the pipeline: each file has its own reader, uploader and tail; the file that is ahead in records waits a round.  The second
    file's records are a third longer than the first's here (longer names), so the two run out of step at every piece border."""
    def grow(path_in, path_out, reps, pad):
        recs = open(path_in, "rb").read().split(b"\n")
        lines = []
        for r in range(reps):
            for i in range(0, len(recs) - 1, 4):
                lines += [recs[i] + (b" %s%d" % (pad, r)), recs[i + 1], b"+" + (recs[i][1:] if pad else b""), recs[i + 3]]
        open(path_out, "wb").write(b"\n".join(lines) + b"\n")
    reps = 300 if chunk in ("3000000", "") else 3                # ~10 MB per file for the piece sizes in megabytes
    f1, f2 = str(tmp_path / "a1.fq"), str(tmp_path / "a2.fq")
    grow(os.path.join(GOLD, "cli_c2_r1.fq"), f1, reps, b"")
    grow(os.path.join(GOLD, "cli_c2_r2.fq"), f2, reps, b"a-much-longer-name-on-the-second-file-")
    ck = ["chunk=" + chunk] if chunk else []
    for flags in (["ktrim=r", "k=23", "mink=11", "hdist=1", "ref=adapters", "tpe"], ["ktrim=n", "k=23", "mink=11", "ref=adapters"], ["ktrim=rl", "k=23", "mink=11", "hdist=1", "ref=adapters"],
                  ["k=25", "hdist=1", "ref=adapters", "rieb=f"]):
        got = {}
        for mode, extra in (("host", ["deviceingest=f"]), ("serial", ["deviceingest=t", "pipeline=f"] + ck), ("piped", ["deviceingest=t", "timeline=t"] + ck)):
            if mode == "host" and reps > 100 and flags[0] != "ktrim=r":
                continue
            out, outm = (str(tmp_path / (mode + n)) for n in ("o.fq", "m.fq"))
            r = run_cli(["in=" + f1, "in2=" + f2, "out=" + out, "outm=" + outm] + flags + extra)
            assert r.returncode == 0, (flags, mode, r.stderr)
            got[mode] = (open(out, "rb").read(), open(outm, "rb").read(), [l for l in r.stderr.splitlines() if "reads" in l and not l.startswith("{")])
            if mode == "piped":
                tl = json.loads([l for l in r.stderr.splitlines() if l.startswith('{"timeline"')][0])["timeline"]
                assert tl["files"] == 2 and tl["file_bytes"] == os.path.getsize(f1) + os.path.getsize(f2) and tl["out_bytes"] == len(got[mode][0]) + len(got[mode][1])
        assert got["piped"] == got["serial"], (flags, chunk)
        if "host" in got:
            assert got["piped"] == got["host"], (flags, chunk)
        assert len(got["piped"][0]) > 1000
    short = str(tmp_path / "short.fq")
    open(short, "wb").write(b"\n".join(open(f2, "rb").read().split(b"\n")[:4 * 37]) + b"\n")
    for extra in (["pipeline=f"], []):
        r = run_cli(["in=" + f1, "in2=" + short, "k=23", "ref=adapters", "deviceingest=t"] + ck + extra)
        assert r.returncode == 1 and "different numbers of reads" in r.stderr, r.stderr


@pytest.mark.gpu
def test_cli_picks_the_device_pipeline_by_itself(tmp_path):
    """Without deviceingest= the CLI takes the device pipeline where it serves the run (timeline=t only prints there) and the host parser elsewhere
    (ksplit, trimfailuresto1bp, several devices); the outputs do not depend on the choice."""
    f1 = os.path.join(GOLD, "cli_c3_interleaved.fq")
    outs = {}
    for name, extra in (("auto", ["auto"]), ("host", ["deviceingest=f"])):
        out = str(tmp_path / (name + ".fq"))
        r = run_cli(["in=" + f1, "int=t", "out=" + out, "k=31", "hdist=1", "ref=phix", "timeline=t"] + extra)
        assert r.returncode == 0, r.stderr
        outs[name] = (open(out, "rb").read(), [l for l in r.stderr.splitlines() if "reads" in l and not l.startswith("{")])
        assert ('{"timeline"' in r.stderr) == (name == "auto")
    assert outs["auto"] == outs["host"]
    for flags in (["ksplit=t", "k=23", "ref=adapters"], ["ktrim=r", "k=23", "ref=adapters", "trimfailures=t"], ["k=31", "ref=phix", "devices=0,0"]):
        r = run_cli(["in=" + os.path.join(GOLD, "cli_c2_r1.fq"), "timeline=t", "auto"] + flags)
        assert r.returncode == 0 and '{"timeline"' not in r.stderr, (flags, r.stderr)


@pytest.mark.gpu
def test_cli_reads_a_named_pipe_to_its_end(tmp_path):
    """in= may be a FIFO, /dev/stdin or <(zcat ..): st_size is 0 there, and the pipelined ingest (the default) sizes its pieces from st_size -- such inputs
    take the serial form instead of reporting 'Input: 0 reads' with exit status 0."""
    import threading
    f1 = os.path.join(GOLD, "cli_c3_interleaved.fq")
    want = str(tmp_path / "file.fq")
    r0 = run_cli(["in=" + f1, "int=t", "out=" + want, "k=31", "hdist=1", "ref=phix", "auto"])
    assert r0.returncode == 0, r0.stderr
    fifo, got = str(tmp_path / "in.fifo"), str(tmp_path / "fifo.fq")
    os.mkfifo(fifo)
    def feed():
        with open(fifo, "wb") as w, open(f1, "rb") as src:
            w.write(src.read())
    t = threading.Thread(target=feed); t.start()
    r = run_cli(["in=" + fifo, "int=t", "out=" + got, "k=31", "hdist=1", "ref=phix", "auto"])
    t.join()
    assert r.returncode == 0, r.stderr
    assert open(got, "rb").read() == open(want, "rb").read()
    line = lambda s: [l for l in s.splitlines() if l.startswith("Input:")]
    assert line(r.stderr) == line(r0.stderr) and "\t0 reads" not in line(r.stderr)[0]


@pytest.mark.gpu
def test_cli_deviceingest_rejects_what_it_cannot_serve(tmp_path):
    f1 = os.path.join(GOLD, "cli_c2_r1.fq")
    r = run_cli(["in=" + f1, "ksplit=t", "k=23", "ref=adapters", "deviceingest=t"])
    assert r.returncode == 1 and "deviceingest" in r.stderr
    bad = tmp_path / "bad.fq"
    bad.write_text("@a\nACGT\n+\nIIII\n@b\nACGT\nIIII\n+\n")
    r = run_cli(["in=" + str(bad), "k=23", "ref=adapters", "deviceingest=t"])
    assert r.returncode == 1 and "malformed FASTQ record" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("ingest", [[], ["deviceingest=t"]])
def test_cli_stats_and_rpkm_files(tmp_path, ingest):
    """stats= (3 and 5 columns), rpkm= and refstats= as BBDukProcessorS.writeStats / writeRPKM (:572-655) and BBDukIndexMod.writeRefStats
    (:196-245) format the per-scaffold counters; expected files come from the spec's counters (tests/golden/make_golden.py write_stats_files)."""
    f1, f2 = os.path.join(GOLD, "cli_c2_r1.fq"), os.path.join(GOLD, "cli_c2_r2.fq")
    for cols, stem in (("3", "cli_c2_stats3.txt"), ("5", "cli_c2_stats5.txt")):
        st, rp, rs = str(tmp_path / "stats.txt"), str(tmp_path / "rpkm.txt"), str(tmp_path / "refstats.txt")
        r = run_cli(["in=" + f1, "in2=" + f2, "stats=" + st, "rpkm=" + rp, "refstats=" + rs, "statscolumns=" + cols, "ktrim=r", "k=23", "mink=11", "hdist=1", "ref=adapters"] + ingest)
        assert r.returncode == 0, r.stderr
        assert open(rs).read() == open(os.path.join(GOLD, "cli_c2_refstats.txt")).read().replace("IN1", f1).replace("IN2", f2)
        want = open(os.path.join(GOLD, stem)).read().replace("IN1", f1).replace("IN2", f2)
        assert open(st).read() == want
        assert open(rp).read() == open(os.path.join(GOLD, "cli_c2_rpkm.txt")).read().replace("IN1", f1).replace("IN2", f2)


@pytest.mark.gpu
def test_cli_rename_appends_scaffolds_and_hit_counts(tmp_path):
    """rename=t (implies findbestmatch, BBDukParser.java:153): a matched read's header gets "\\tscaffold=hits" for every
    scaffold it hit, in first-hit order (BBDukProcessorS.java:2508-2522).  Expected headers come from oracle/spec.py."""
    from oracle.spec import Spec, Args
    from tests import util
    names = [None]
    for line in open(util.ADAPTERS, "rb"):
        if line.startswith(b">"):
            names.append(line[1:].rstrip(b"\r\n").decode())
    s = Spec(Args(k=23, findBestMatch=True)); s.load_fasta(util.ADAPTERS)
    f1, f2 = os.path.join(GOLD, "cli_c2_r1.fq"), os.path.join(GOLD, "cli_c2_r2.fq")
    l1, l2 = open(f1).read().splitlines(), open(f2).read().splitlines()
    expect_m, expect_u, renamed = [], [], 0
    for i in range(0, len(l1), 4):
        pair = []
        for pairnum, l in ((0, l1), (1, l2)):
            found, vid = s.find_best_match(l[i + 1].encode(), pairnum, 0)
            hdr = l[i] + "".join("\t%s=%d" % (names[sid], c) for sid, c in (s.last_matches if found > 0 else []))
            renamed += found > 0
            pair.append((vid > 0, "\n".join([hdr, l[i + 1], "+", l[i + 3]]) + "\n"))
        (expect_m if (pair[0][0] or pair[1][0]) else expect_u).extend(p[1] for p in pair)     # rieb=t: either mate removes the pair
    out, outm = str(tmp_path / "clean.fq"), str(tmp_path / "matched.fq")
    r = run_cli(["in=" + f1, "in2=" + f2, "out=" + out, "outm=" + outm, "k=23", "rename=t", "ref=adapters", "batch=7"])
    assert r.returncode == 0, r.stderr
    assert renamed > 5 and any(h.count("\t") >= 2 for h in expect_m)                         # some reads hit several adapters
    assert open(outm).read() == "".join(expect_m)
    assert open(out).read() == "".join(expect_u)
    r = run_cli(["in=" + f1, "k=23", "rename=t", "mcf=0.5", "ref=adapters"])
    assert r.returncode == 1 and "rename" in r.stderr
    r = run_cli(["in=" + f1, "k=23", "rename=t", "ref=adapters", "deviceingest=t"])
    assert r.returncode == 1 and "rename" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["ktrim", "kmask", "kfilter"])
def test_cli_devices_shards_batches_and_allreduces_counters(tmp_path, mode):
    """devices=0,0,0: three handles, every batch cut into three blocks of whole pairs, one counter all-reduce at the end
    (bbduk_comm_create_local + bbduk_allreduce_counters_local).  Files, TSV, closing lines and stats equal the one-handle run."""
    flags = {"ktrim": ["ktrim=r", "k=23", "mink=11", "hdist=1", "ref=adapters"], "kmask": ["ktrim=n", "k=23", "mink=11", "hdist=1", "ref=adapters"],
             "kfilter": ["k=31", "hdist=1", "ref=phix"]}[mode]
    ins = ["in=" + os.path.join(GOLD, "cli_c3_interleaved.fq"), "int=t"] if mode == "kfilter" else \
          ["in=" + os.path.join(GOLD, "cli_c2_r1.fq"), "in2=" + os.path.join(GOLD, "cli_c2_r2.fq")]
    got = {}
    for tag, extra in (("one", ["batch=14"]), ("three", ["batch=14", "devices=0,0,0"])):
        tsv, out, outm, st = (str(tmp_path / (tag + n)) for n in (".tsv", ".fq", ".m.fq", ".stats"))
        r = run_cli(ins + ["tsv=" + tsv, "out=" + out, "outm=" + outm, "stats=" + st] + extra + flags)
        assert r.returncode == 0, r.stderr
        got[tag] = (open(tsv).read(), open(out).read(), open(outm).read(), open(st).read(), r.stderr)
    assert got["one"] == got["three"]
    assert "Input:" in got["three"][4] and len(got["three"][0]) > 0


def test_cli_refuses_a_quote_in_a_gz_path(tmp_path):
    r = run_cli(["in=" + str(tmp_path / "x';touch PWNED;'.fq.gz"), "ktrim=r", "k=23", "ref=adapters"])
    assert r.returncode == 1 and not os.path.exists(os.path.join(ROOT, "PWNED"))


@pytest.mark.gpu
def test_cli_trimfailuresto1bp_cuts_instead_of_removing(tmp_path):
    """trimfailures=t (BBDukParser.java:774): what the plain run sends to outm stays, cut to one base; nothing is evicted (:1431)."""
    tsv0, tsv1, out, outm = (str(tmp_path / n) for n in ("plain.tsv", "tf.tsv", "clean.fq", "removed.fq"))
    base = ["in=" + os.path.join(GOLD, "cli_c3_interleaved.fq"), "int=t", "k=31", "hdist=1", "ref=phix"]
    r0 = run_cli(base + ["tsv=" + tsv0, "rieb=f"])
    r1 = run_cli(base + ["tsv=" + tsv1, "out=" + out, "outm=" + outm, "trimfailures=t"])
    assert r0.returncode == 0 and r1.returncode == 0, r0.stderr + r1.stderr
    plain = [l.split("\t") for l in open(tsv0).read().splitlines()]
    tf = [l.split("\t") for l in open(tsv1).read().splitlines()]
    src = {n: (b, q) for n, b, q in _read_fastq(os.path.join(GOLD, "cli_c3_interleaved.fq"))}
    kept = _read_fastq(out)
    assert len(kept) == len(tf) and len(_read_fastq(outm)) == 0
    ncut = 0
    for (name, L, a, _id, new_len, flags), p, (kn, kb, kq) in zip(tf, plain, kept):
        assert name == p[0] == kn and a == p[2] and int(flags) & 2 == 0
        discarded = int(p[5]) & 1                    # the same k-mer verdict per read (rieb=f: the discard bit alone)
        assert (int(flags) & 1) == discarded
        want = 1 if (discarded and int(L) > 1) else int(L)
        assert int(new_len) == want and kb == src[name][0][:want] and kq == src[name][1][:want]
        ncut += discarded
    assert ncut > 0 and "Contaminants:" in r1.stderr
    r = run_cli(base + ["trimfailures=t", "deviceingest=t"])
    assert r.returncode != 0 and "trimfailuresto1bp" in r.stderr


# ---- the reference's own sample reads (resources/sample1.fq.gz, sample2.fq.gz of the BBTools tree, kept as data/): gzipped two-file
# input, real names and qualities; expectations from oracle/spec.py (tests/golden/make_sample_golden.py)
@pytest.mark.gpu
@pytest.mark.parametrize("stem,flags", [("cli_sample_ktrimr_k25_phix", ["ktrim=r", "k=25", "ref=phix"]),
                                        ("cli_sample_kfilter_k31_hdist1_phix", ["k=31", "hdist=1", "ref=phix"]),
                                        ("cli_sample_kfilter_k31_mkf_phix", ["k=31", "mkf=0.5", "ref=phix"])])
@pytest.mark.parametrize("ingest", ["deviceingest=f", "deviceingest=t", "auto"])      # the host parser; the device pipeline through `gzip -dc` (its serial form); what the CLI picks by itself
def test_cli_on_the_reference_sample_reads(tmp_path, stem, flags, ingest):
    tsv, out, outm = (str(tmp_path / n) for n in ("o.tsv", "clean.fq", "removed.fq"))
    r = run_cli(["in=" + os.path.join(ROOT, "data", "sample1.fq.gz"), "in2=" + os.path.join(ROOT, "data", "sample2.fq.gz"),
                 "tsv=" + tsv, "out=" + out, "outm=" + outm, ingest] + flags)
    assert r.returncode == 0, r.stderr
    assert open(tsv).read() == open(os.path.join(GOLD, stem + ".tsv")).read()
    rows = [l.split("\t") for l in open(tsv).read().splitlines()]
    kept, removed = _read_fastq(out), _read_fastq(outm)
    assert len(kept) + len(removed) == len(rows) == 200
    assert {n for n, _, _ in removed} == {x[0] for x in rows if int(x[5]) & 2}
    newlen = {x[0]: int(x[4]) for x in rows}
    assert all(len(b) == len(q) == newlen[n] for n, b, q in kept + removed)


@pytest.mark.gpu
def test_cli_2x151_reads_take_the_tail_pass(tmp_path):
    """2x151 with k=23: 129 k-mer end positions per read, one more than the pair scan's block holds (the tail pass of the main kernel)."""
    tsv = str(tmp_path / "o.tsv")
    r = run_cli(["in=" + os.path.join(GOLD, "cli_len151_interleaved.fq"), "int=t", "tsv=" + tsv, "ktrim=r", "k=23", "mink=11", "hdist=1", "ref=adapters"])
    assert r.returncode == 0, r.stderr
    assert open(tsv).read() == open(os.path.join(GOLD, "cli_len151.tsv")).read()
    rows = [l.split("\t") for l in open(tsv).read().splitlines()]
    assert sum(1 for x in rows if int(x[1]) == 151 and int(x[4]) == 128) >= 3               # first hit = the 129th position: found by the tail pass
