#!/usr/bin/env python3
"""
BASELINE configs[3] at its real size: kfilter k=31 hdist=0 against a 10 Gbase synthetic contaminant reference (~10^10 distinct
31-mers), the map resident in HBM in the big layout, 2x150 bp reads with 1 % of the pairs drawn from the reference.

The reference never exists on the host: it is generated on the device chunk by chunk (seeded torch generator: chunk c = seed
40 + c, uniform ACGT, --scaffold-bases bases per scaffold) and handed to the streaming build (bbduk_build_begin / _add_device /
_end).  Checks at this size (no CPU can replay 10^10 keys in minutes):
  * table size = reference windows minus the handful of duplicate 31-mers a random 10 Gbase sequence holds;
  * sampled reference windows of EVERY chunk are found with their scaffold's id, random keys are absent (bbduk_table_lookup);
  * a sample of the reads equals the CPU oracle's answers, the oracle holding the part of the reference the contaminated reads are
    drawn from (a random read matching the rest of the reference has probability ~2e-8 per k-mer: reported, not expected).
One JSON line.  python profiles/bench_c4_full.py [--ref-bases 1e10] [--reads 20000000] [--check-reads 50000]
"""
import argparse, json, os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np


def keys_of(windows, k=31, mm=True):
    """canonical, middle-masked, length-tagged keys of ASCII k-mers (numpy uint8 [n, k]) -- BBDukIndexMod.toValue"""
    code = np.zeros(256, np.uint64); code[ord("C")] = 1; code[ord("G")] = 2; code[ord("T")] = 3
    c = code[windows]
    f = np.zeros(len(windows), np.uint64); r = np.zeros(len(windows), np.uint64)
    for j in range(k):
        f = (f << np.uint64(2)) | c[:, j]
        r = (r << np.uint64(2)) | (np.uint64(3) - c[:, k - 1 - j])
    v = np.maximum(f, r)
    if mm:
        v &= ~(np.uint64(3) << np.uint64(2 * ((k - 1) // 2)))
    return (v | (np.uint64(1) << np.uint64(2 * k))).astype(np.int64)


def run(ref_bases=1e10, scaffold_bases=10_000_000, chunk_scaffolds=50, reads=20_000_000, steps=3, check_reads=50_000, oracle_bases=50_000_000, fallback_reads=0):
    import types
    return _run(types.SimpleNamespace(ref_bases=ref_bases, scaffold_bases=scaffold_bases, chunk_scaffolds=chunk_scaffolds, reads=reads, steps=steps,
                                      check_reads=check_reads, oracle_bases=oracle_bases, pair_scan=False, big52=False, gm=0, load=0, lib="", mask="0", ktrim=False, fallback_reads=fallback_reads))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref-bases", type=float, default=1e10)
    ap.add_argument("--scaffold-bases", type=int, default=10_000_000)
    ap.add_argument("--chunk-scaffolds", type=int, default=50)
    ap.add_argument("--reads", type=int, default=20_000_000)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--check-reads", type=int, default=50_000)
    ap.add_argument("--pair-scan", action="store_true", help="experiments: the pair-scan kernel of round 2 instead of the stream kernel")
    ap.add_argument("--ktrim", action="store_true", help="ktrim=r k=31 against the same reference (round 4: the trimming mode's large-map path) instead of kfilter")
    ap.add_argument("--big52", action="store_true", help="experiments: the 52-bit line function and its pair scan below 2^31 keys too")
    ap.add_argument("--gm", type=int, default=0, help="experiments: minimizer length m of the big layout's lines (W = H - m + 1)")
    ap.add_argument("--load", type=int, default=0, help="experiments: keys per 100 slots the lines are sized for (BBDUK_HOOK_BIG_LOAD; 0 = the library's choice)")
    ap.add_argument("--lib", default="", help="experiments: another build of the library under bbtools_amd/ (ab_tsw: the timing switches)")
    ap.add_argument("--n-rate", type=float, default=0.001, help="experiments: share of undefined bases in the synthetic reads (the benchmark's: 0.001)")
    ap.add_argument("--mask", default="0", help="experiments: timing mask of the ab_tsw build (a+b = bits; 22: the stream form instead of the read-aligned one)")
    ap.add_argument("--oracle-bases", type=int, default=50_000_000, help="prefix of scaffold 1.. the contaminated reads are drawn from (and the oracle indexes)")
    a = ap.parse_args()
    print(json.dumps(_run(a)))


def _run(a):
    import torch
    from bbtools_amd import bbduk as B
    if getattr(a, "lib", ""): B.use_library(os.path.join(ROOT, "bbtools_amd", a.lib + ".so"))
    ref_bases = int(a.ref_bases)
    n_scaf = max(1, ref_bases // a.scaffold_bases)
    ref_bases = n_scaf * a.scaffold_bases
    host = B.HostIndex("ktrim=r k=31 hdist=0" if getattr(a, "ktrim", False) else "k=31 hdist=0")
    p = host.params(0); p.numScaffolds = n_scaf + 1
    gpu = B.BBDukGpu(p)
    if getattr(a, "pair_scan", False):
        gpu.test_hook(B.HOOK_PAIR_SCAN, 1)
    if getattr(a, "gm", 0):
        gpu.test_hook(B.HOOK_LDS_BITS, a.gm)
    if getattr(a, "load", 0):
        gpu.test_hook(B.HOOK_BIG_LOAD, a.load)
    if getattr(a, "mask", "0") != "0":
        gpu.test_hook(B.HOOK_TIMING_MASK, sum(1 << int(b) for b in a.mask.split("+")))
    if getattr(a, "big52", False):                                  # the 52-bit line function + the pair scan (what maps beyond 2^31 keys take) at any size
        gpu.test_hook(B.HOOK_BIG_LAYOUT, 2)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device="cuda")
    t0 = time.perf_counter()
    gpu.build_begin(ref_bases, 0, 0)
    t_alloc = time.perf_counter() - t0
    samples = []                                                     # (windows uint8[n,31], expected ids) from every chunk
    contam = None
    srng = np.random.default_rng(1)
    t_gen = t_add = 0.0
    for c0 in range(0, n_scaf, a.chunk_scaffolds):
        ns = min(a.chunk_scaffolds, n_scaf - c0)
        t = time.perf_counter()
        g = torch.Generator(device="cuda"); g.manual_seed(40 + c0)
        chunk = lut[torch.randint(0, 4, (ns * a.scaffold_bases,), generator=g, device="cuda", dtype=torch.int64)]
        torch.cuda.synchronize(); t_gen += time.perf_counter() - t
        offs = np.arange(ns + 1, dtype=np.int64) * a.scaffold_bases
        t = time.perf_counter(); gpu.build_add_device(chunk, offs, c0 + 1); t_add += time.perf_counter() - t
        pos = srng.integers(0, a.scaffold_bases - 31, 2000) + srng.integers(0, ns, 2000) * a.scaffold_bases
        idx = torch.from_numpy(pos[:, None] + np.arange(31)[None, :]).cuda()
        samples.append((chunk[idx].cpu().numpy(), (c0 + 1 + pos // a.scaffold_bases).astype(np.int32)))
        if contam is None:
            contam = chunk[:min(a.oracle_bases, a.scaffold_bases)].cpu().numpy().tobytes()
        del chunk, idx
    t = time.perf_counter(); gpu.build_end(); t_end = time.perf_counter() - t
    t_build = time.perf_counter() - t0
    nkeys = gpu.table_size
    out = {"workload": "BASELINE configs[3]: kfilter k=31 hdist=0 vs %d-base synthetic reference (%d scaffolds), map resident in HBM" % (ref_bases, n_scaf),
           "keys": int(nkeys), "windows": int(n_scaf * (a.scaffold_bases - 30)), "table_bytes": int(gpu.table_bytes),
           "bytes_per_key": round(gpu.table_bytes / max(1, nkeys), 2), "spilled_keys": int(gpu.table_spilled),
           "build_s": round(t_build, 2), "build_detail_s": {"alloc+clear": round(t_alloc, 2), "generate_reference": round(t_gen, 2), "insert": round(t_add, 2), "end": round(t_end, 2)}}
    hist = gpu.line_histogram()
    out["line_load_histogram"] = hist.tolist(); out["lines"] = int(hist.sum())
    out["line_load_mean"] = round(float((hist * np.arange(33)).sum() / hist.sum()), 3)      # (per 32 slots: round 6 counts the 64-byte HALVES of its 128-byte lines)
    out["gm"] = int(getattr(a, "gm", 0)); out["load_hook"] = int(getattr(a, "load", 0))
    # ---- table checks
    W = np.concatenate([w for w, _ in samples]); ids = np.concatenate([i for _, i in samples])
    got = gpu.table_lookup(keys_of(W))
    # a sampled window whose key also occurs in an earlier scaffold legitimately reports the smaller id
    out["sampled_reference_windows"] = int(len(W)); out["sampled_found_with_id"] = int((got == ids).sum()); out["sampled_found_smaller_id"] = int(((got > 0) & (got < ids)).sum())
    out["sampled_missing"] = int((got <= 0).sum())
    rk = np.random.default_rng(7).integers(0, 1 << 62, 1_000_000, dtype=np.int64) | (np.int64(1) << np.int64(62))
    rk &= ~(np.int64(3) << np.int64(30))
    out["random_keys_found"] = int((gpu.table_lookup(rk) > 0).sum())
    # ---- the operator
    sp = B.synth_params(4, contam=contam, contam_frac=0.01, n_rate=float(getattr(a, "n_rate", 0.001)))
    fb = int(getattr(a, "fallback_reads", 0) or 0)
    for reads in ([a.reads, fb] if 0 < fb < a.reads else [a.reads]):      # the batch beside the map: the requested size, else the smaller one
        n_pairs = reads // 2; n = 2 * n_pairs
        try:
            d_bases = torch.empty(n * 150, dtype=torch.uint8, device="cuda"); d_off = torch.empty(n + 1, dtype=torch.int64, device="cuda")
            d_a = torch.empty(n, dtype=torch.int32, device="cuda"); d_id = torch.empty_like(d_a); d_fl = torch.empty(n, dtype=torch.uint8, device="cuda")
            break
        except torch.OutOfMemoryError as e:
            if reads != a.reads or not fb:
                gpu.close(); host.close()
                raise
            out["reads_requested"] = int(a.reads); out["full_batch_error"] = repr(e)[:200]
            d_bases = d_off = d_a = d_id = d_fl = None
            torch.cuda.empty_cache()
    B.synth_generate_device(sp, 0, n_pairs, d_bases, d_off, 0)
    d_c = torch.zeros(gpu.counters_len, dtype=torch.int64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(1 + a.steps):
        d_c.zero_(); gpu.process_batch_device(d_bases, d_off, True, d_a, d_id, d_fl, d_c, st)
    torch.cuda.synchronize()
    ms = gpu.kernel_time_ms(a.steps)
    c = d_c.cpu().numpy()
    algo = n * (150 + 8 + 8 + 120 * 8) + 4 * int(c[4])
    out.update({"reads": n, "kernel_ms": round(ms, 3), "Gbases_per_s": round(n * 150 / (ms * 1e-3) / 1e9, 2),
                "algorithmic_bytes": int(algo), "algorithmic_GBps": round(algo / (ms * 1e-3) / 1e9, 1), "frac_of_8TBps": round(algo / (ms * 1e-3) / 8e12, 4),
                "readsKFiltered": int(c[4]), "device_status": int(c[15])})
    if a.check_reads > 0:
        from oracle.oracle_ffi import Oracle
        o = Oracle(k=31, ktrimRight=1) if getattr(a, "ktrim", False) else Oracle(k=31); o.add_ref(contam)
        m = min(n, a.check_reads) & ~1
        hb, hoff = B.synth_generate_host(sp, 0, m // 2)
        oa, oi, of = o.process_batch(hb, hoff, True, nthreads=os.cpu_count())
        ga = d_a[:m].cpu().numpy(); gi = d_id[:m].cpu().numpy(); gf = d_fl[:m].cpu().numpy()
        diff = np.nonzero((ga != oa) | (gi != oi) | (gf != of))[0]
        out["checked_reads"] = int(m); out["checked_reads_differ"] = int(len(diff)); out["checked_reads_hit"] = int((oa > 0).sum())
        out["readsKTrimmed"] = int(c[2])
        # a difference is legitimate only if the device found a k-mer of the reference OUTSIDE the oracle's part of it
        out["differ_device_hit_oracle_miss"] = int(((ga[diff] > 0) & (oa[diff] == 0)).sum()) if len(diff) else 0
    gpu.close(); host.close()
    del d_bases, d_off, d_a, d_id, d_fl
    torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    main()
