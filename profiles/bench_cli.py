#!/usr/bin/env python3
"""
Secondary measurement: bbduk_cli wall clock on an interleaved FASTQ file in /tmp (configs[1] reads and flags), host parser
path against deviceingest=t, with and without out=.  Whole process time (start-up, table build, file I/O on tmpfs/disk).
One JSON line.
"""
import argparse, json, os, subprocess, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "profiles"))


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--reads", type=int, default=4_000_000); ap.add_argument("--dir", default="/tmp"); ap.add_argument("--variants", default="", help="comma-separated subset of host_parse,host_parse_out,deviceingest_serial,deviceingest_serial_out,deviceingest,deviceingest_out,two_files_serial,two_files,two_files_out")
    a = ap.parse_args()
    from bbtools_amd import bbduk as B, _build
    from bench_ingest import fastq_text
    text, _, _ = fastq_text(B, a.reads // 2)
    path = os.path.join(a.dir, "bench_cli_in.fq"); outp = os.path.join(a.dir, "bench_cli_out.fq")
    want = a.variants.split(",") if a.variants else []
    one = not want or any(not v.startswith("two_files") for v in want)
    if one:                                                       # (the box's /tmp holds 79 GB: the interleaved file, the two-file form and out= do not fit together at 31 GB)
        text.tofile(path)
    p1, p2 = os.path.join(a.dir, "bench_cli_r1.fq"), os.path.join(a.dir, "bench_cli_r2.fq")
    two = not a.variants or "two_files" in a.variants
    if two:                                                       # the same reads as two files (records have one size here: the rows of a matrix)
        rows = text.reshape(a.reads // 2 * 2, -1)
        rows[0::2].tofile(p1); rows[1::2].tofile(p2)
    os.sync()                                                     # (an input file is not 31 GB of dirty page cache: without this the kernel's writeback of it competes with the run's own out= writes)
    flags = ["ktrim=r", "k=23", "mink=11", "hdist=1", "ref=adapters", "int=t", "devicebuild=t", "resources=" + os.path.join(ROOT, "data")]
    res = {}
    for name, extra in (("host_parse", ["deviceingest=f"]), ("host_parse_out", ["deviceingest=f", "out=" + outp]), ("deviceingest_serial", ["deviceingest=t", "pipeline=f"]), ("deviceingest_serial_out", ["deviceingest=t", "pipeline=f", "out=" + outp]),
                        ("deviceingest", ["deviceingest=t", "timeline=t"]), ("deviceingest_out", ["deviceingest=t", "timeline=t", "out=" + outp]),
                        ("two_files_serial", ["deviceingest=t", "pipeline=f"]), ("two_files", ["deviceingest=t", "timeline=t"]), ("two_files_out", ["deviceingest=t", "timeline=t", "out=" + outp])):
        if a.variants and name not in a.variants.split(","): continue
        best = None
        for _ in range(3 if "out" in name else 2):
            if os.path.exists(outp):
                os.remove(outp)                                   # (a fresh output file: truncating 28 GB of dirty page cache is seconds of its own)
            t0 = time.perf_counter()
            src = ["in=" + p1, "in2=" + p2] if name.startswith("two_files") else ["in=" + path]
            r = subprocess.run([_build.cli_path()] + src + [f for f in flags if not (f == "int=t" and name.startswith("two_files"))] + extra, capture_output=True, text=True)
            dt = time.perf_counter() - t0
            assert r.returncode == 0, r.stderr
            if best is None or dt < best:
                best = dt
                tl = [l for l in r.stderr.splitlines() if l.startswith('{"timeline"')]
        res[name] = {"s": round(best, 3), "Gbases_per_s": round(a.reads * 150 / best / 1e9, 3)}
        if tl:
            res[name]["timeline"] = json.loads(tl[0])["timeline"]
        if "out" in name:
            res[name]["out_bytes"] = os.path.getsize(outp)
    for q in (path, p1, p2):
        if os.path.exists(q):
            os.remove(q)
    if os.path.exists(outp):
        os.remove(outp)
    print(json.dumps({"workload": "bbduk_cli, interleaved FASTQ file of %d reads x 150 bp (%.2f GB) in %s, ktrim=r k=23 mink=11 hdist=1 ref=adapters"
                                  % (a.reads, len(text) / 1e9, a.dir), "wall": res}))


if __name__ == "__main__":
    main()
