#!/usr/bin/env python3
"""Secondary measurement: the reference's own sizing example -- E. coli, k=31 hdist=1 -> 427,998,710 keys
(docs/guides/BBDukGuide.txt:34) -- with a seeded random 4.6 Mbase genome in its place: kfilter k=31 hdist=1, the map built on the
device, 2x150 bp reads with 1 % of the pairs drawn from the genome.  Kernel Gbases/s, map bytes, and a sample checked against the
oracle on a PREFIX of the genome (the oracle builds hdist=1 maps at ~1e7 keys/s: --oracle-bases of it).  One JSON line."""
import argparse, json, os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np


def run(**kw):
    """bench.py's `configs` block calls this (defaults = the command line's)"""
    import types
    d = dict(genome_bases=4_600_000, reads=20_000_000, steps=3, check_reads=0, oracle_bases=100_000, args="k=31 hdist=1", bucket_bits=0, plain=False, m=0, build="device", lib="")
    d.update(kw)
    return _run(types.SimpleNamespace(**d))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-bases", type=int, default=4_600_000); ap.add_argument("--reads", type=int, default=20_000_000)
    ap.add_argument("--steps", type=int, default=3); ap.add_argument("--check-reads", type=int, default=0); ap.add_argument("--oracle-bases", type=int, default=100_000)
    ap.add_argument("--args", default="k=31 hdist=1"); ap.add_argument("--bucket-bits", type=int, default=0); ap.add_argument("--plain", action="store_true"); ap.add_argument("--m", type=int, default=0); ap.add_argument("--build", default="device", help="host: the map arrives as (key, id) pairs, as a JVM host uploads it"); ap.add_argument("--lib", default="", help="experiments: another build of the library under bbtools_amd/ (same-box A/B)")
    a = ap.parse_args()
    print(json.dumps(_run(a)))


def _run(a):
    import torch
    from bbtools_amd import bbduk as B
    if getattr(a, "lib", ""): B.use_library(os.path.join(ROOT, "bbtools_amd", a.lib + ".so"))
    rng = np.random.default_rng(77)
    genome = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, a.genome_bases)].tobytes()
    t0 = time.perf_counter()
    hooks = {}
    if a.bucket_bits: hooks[B.HOOK_BUCKET_BITS] = a.bucket_bits
    if a.plain: hooks[B.HOOK_LDS_BITS] = 0
    if a.m: hooks[B.HOOK_LDS_BITS] = a.m
    duk = B.BBDuk(a.args, refs=[genome], build=getattr(a, "build", "device"), hooks=hooks)
    build_s = time.perf_counter() - t0
    n = a.reads & ~1
    d_bases = torch.empty(n * 150, dtype=torch.uint8, device="cuda"); d_off = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    sp = B.synth_params(9, contam=genome[:a.oracle_bases], contam_frac=0.01)
    B.synth_generate_device(sp, 0, n // 2, d_bases, d_off, 0)
    d_a = torch.empty(n, dtype=torch.int32, device="cuda"); d_id = torch.empty_like(d_a); d_fl = torch.empty(n, dtype=torch.uint8, device="cuda")
    d_c = torch.zeros(duk.gpu.counters_len, dtype=torch.int64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(1 + a.steps):
        d_c.zero_(); duk.gpu.process_batch_device(d_bases, d_off, True, d_a, d_id, d_fl, d_c, st)
    torch.cuda.synchronize()
    ms = duk.gpu.kernel_time_ms(a.steps)
    c = d_c.cpu().tolist()
    out = {"workload": "%s vs a random %d-base genome (BBDukGuide.txt:34's E. coli example), 2x150 bp, 1 %% of the pairs from the genome" % (a.args if "ktrim" in a.args else "kfilter " + a.args, a.genome_bases),
           "keys": int(duk.stored_kmers), "table_bytes": int(duk.gpu.table_bytes), "build": getattr(a, "build", "device"), "layout": int(duk.gpu.table_layout), "spilled_keys": int(duk.gpu.table_spilled) if hasattr(duk.gpu, "table_spilled") else None,
           "build_s": round(build_s, 2), "reads": n, "kernel_ms": round(ms, 3), "Gbases_per_s": round(n * 150 / ms / 1e6, 1),
           "readsKFiltered": int(c[B.COUNTER_NAMES.index("readsKFiltered")]), "readsKTrimmed": int(c[B.COUNTER_NAMES.index("readsKTrimmed")])}
    if a.check_reads:
        from oracle.oracle_ffi import Oracle
        o = Oracle(k=31, hdist=1); o.add_ref(genome[:a.oracle_bases])
        m = a.check_reads & ~1
        hb, hoff = B.synth_generate_host(sp, 0, m // 2)
        oa, oi, of = o.process_batch(hb, hoff, True, nthreads=os.cpu_count())
        ga = d_a[:m].cpu().numpy(); gf = d_fl[:m].cpu().numpy()
        # the device map holds the WHOLE genome: a read the oracle's prefix map misses may still hit elsewhere (reported, ~0 expected)
        out["checked_reads"] = m; out["oracle_hits"] = int((oa > 0).sum()); out["device_hits"] = int((ga > 0).sum())
        out["oracle_hit_device_miss"] = int(((oa > 0) & (ga == 0)).sum()); out["device_hit_oracle_miss"] = int(((ga > 0) & (oa == 0)).sum())
    duk.close()
    return out


if __name__ == "__main__":
    main()
