#!/usr/bin/env python3
"""Secondary measurement: Seal's k-mer path (seal_batch_device) on a binning workload, device-resident, kernel time by HIP events.
Workload: S random scaffolds of G bases (every tenth shares a 2 kb segment with its neighbour, so some k-mers belong to two), 2x150 bp
pairs sampled from them with 1 % substitutions, 10 % of the pairs unrelated; k=31 mm=t (Seal's defaults).  The first pairs are checked
against the C restatement.  One JSON line."""
import argparse, json, os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np


def run(pairs=2_000_000, scaffolds=64, scaffold_len=100_000, args="", check=2000, bucket_bits=0, layout=0, gm=0, big_load=0):
    import types
    a = types.SimpleNamespace(pairs=pairs, scaffolds=scaffolds, scaffold_len=scaffold_len, args=args, check=check, bucket_bits=bucket_bits, layout=layout, gm=gm, big_load=big_load)
    import torch
    from bbtools_amd.seal import Seal
    rng = np.random.default_rng(11)
    S, G, L = a.scaffolds, a.scaffold_len, 150
    refs = rng.integers(0, 4, size=(S, G), dtype=np.uint8)
    for s in range(0, S - 1, 10):
        refs[s + 1, 5000:7000] = refs[s, 5000:7000]
    lut = np.frombuffer(b"ACGT", np.uint8)
    ref_bytes = [lut[refs[s]].tobytes() for s in range(S)]
    n = a.pairs
    sc = rng.integers(0, S, n); ins = rng.integers(200, 500, n); pos = rng.integers(0, G - 500, n)
    idx = np.arange(L)
    r1 = refs[sc[:, None], pos[:, None] + idx[None, :]]
    r2 = 3 - refs[sc[:, None], (pos + ins)[:, None] - 1 - idx[None, :]]                   # reverse complement of the insert's far end
    junk = rng.random(n) < 0.10
    r1[junk] = rng.integers(0, 4, size=(int(junk.sum()), L), dtype=np.uint8); r2[junk] = rng.integers(0, 4, size=(int(junk.sum()), L), dtype=np.uint8)
    for r in (r1, r2):
        m = rng.random(r.shape) < 0.01
        r[m] = (r[m] + rng.integers(1, 4, int(m.sum()), dtype=np.uint8)) & 3
    reads = np.empty((2 * n, L), np.uint8); reads[0::2] = lut[r1]; reads[1::2] = lut[r2]
    t0 = time.time()
    g = Seal(a.args, refs=ref_bytes, max_scaffolds=S + 2, hooks={**({2: a.bucket_bits} if a.bucket_bits else {}), **({5: a.layout} if getattr(a, "layout", 0) else {}), **({3: a.gm} if getattr(a, "gm", 0) else {}), **({8: a.big_load} if getattr(a, "big_load", 0) else {})} or None)     # experiments: BBDUK_HOOK_BUCKET_BITS
    build_s = time.time() - t0
    nr = 2 * n
    d_bases = torch.from_numpy(reads.reshape(-1)).cuda(); d_off = (torch.arange(nr + 1, dtype=torch.int64) * L).cuda()
    cap = 4
    d_s = torch.empty(nr, dtype=torch.int32, device="cuda"); d_a = torch.empty_like(d_s); d_m = torch.empty_like(d_s)
    d_i = torch.empty(nr * cap, dtype=torch.int32, device="cuda"); d_f = torch.empty(nr, dtype=torch.uint8, device="cuda")
    d_c = torch.zeros(g.counters_len(), dtype=torch.int64, device="cuda")
    ts = []
    for it in range(4):
        d_c.zero_()
        g.process_device(d_bases, d_off, nr, nr * L, True, 0, cap, d_s, d_a, d_m, d_i, d_f, d_c)
        torch.cuda.synchronize()
        ts.append(g.kernel_ms())
    ms = min(ts[1:])
    c = d_c.cpu().numpy()
    ok = None
    if a.check:
        from oracle.seal_ffi import SealOracle
        o = SealOracle(ref_bytes)                       # defaults; --args is for experiments only
        if not a.args:
            k = 2 * a.check
            want = o.process_reads([reads[i].tobytes() for i in range(k)], True, 0, max_ids=cap)
            got = (d_s[:k].cpu().numpy(), d_a[:k].cpu().numpy(), d_m[:k].cpu().numpy(), d_i[:k * cap].cpu().numpy().reshape(k, cap), d_f[:k].cpu().numpy())
            ok = all(np.array_equal(w, h) for w, h in zip(want, got))
    return ({"workload": "seal %s: %d scaffolds x %d bases, %d pairs 2x150" % (a.args or "(defaults: k=31 mm=t ambig=random)", S, G, n),
                      "table_keys": int(g.table_keys), "table_pairs": int(g.table_pairs), "host_build_s": round(build_s, 2),
                      "kernel_ms": round(ms, 3), "Gbases_per_s": round(nr * L / ms / 1e6, 2), "readsMatched": int(c[3]), "readsUnmatched": int(c[5]),
                      "ambiguous_pairs": int((d_s[0::2] > 1).sum().item()), "sample_equals_restatement": ok, "status": int(c[15])})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=2_000_000); ap.add_argument("--scaffolds", type=int, default=64)
    ap.add_argument("--scaffold-len", type=int, default=100_000); ap.add_argument("--args", default="")
    ap.add_argument("--check", type=int, default=2000)
    ap.add_argument("--lib", default="", help="experiments: another build under bbtools_amd/ (ab_tsw: stage deletion through the SEAL_TSW environment variable)")
    ap.add_argument("--layout", type=int, default=0, help="BBDUK_HOOK_BIG_LAYOUT on the map: -1 cache-resident, 1 minimizer lines (default: by key count)")
    ap.add_argument("--gm", type=int, default=0, help="experiments: minimizer length m of the big layout's lines (BBDUK_HOOK_LDS_BITS)")
    ap.add_argument("--big-load", type=int, default=0, help="experiments: keys per 100 slots the big layout's lines are sized for (BBDUK_HOOK_BIG_LOAD)")
    ap.add_argument("--bucket-bits", type=int, default=0, help="experiments: log2 of the map's bucket count (default: one bucket per key, rounded up)")
    a = ap.parse_args()
    if a.lib:
        from bbtools_amd import bbduk as _B
        _B.use_library(os.path.join(ROOT, "bbtools_amd", a.lib + ".so"))
    r = run(a.pairs, a.scaffolds, a.scaffold_len, a.args, a.check, a.bucket_bits, a.layout, a.gm, a.big_load)
    if a.gm: r["gm"] = a.gm
    if a.big_load: r["big_load"] = a.big_load
    if a.bucket_bits:
        r["bucket_bits"] = a.bucket_bits
    print(json.dumps(r))


if __name__ == "__main__":
    main()
