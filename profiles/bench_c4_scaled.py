#!/usr/bin/env python3
"""
Secondary measurement (not the bench.py metric): the HBM-resident-map regime of BASELINE config 4, scaled to what
one host can index in a minute -- kfilter k=31 hdist=0 against a seeded uniform-random reference of --ref-bases
bases (default 1e8 => ~1e8 distinct 31-mers, a 9.7 GB device map: far beyond L2 + Infinity Cache), 1 % of the pairs
drawn from it.  With no LDS filter every k-mer probe is one random 8-byte gather into HBM.
Prints one JSON line: Gbases/s, kernel ms, achieved GB/s against the algorithmic 1126 B/read of SURVEY §8d.
  python profiles/bench_c4_scaled.py [--ref-bases N] [--reads R] [--check]
"""
import argparse, json, os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref-bases", type=int, default=100_000_000)
    ap.add_argument("--reads", type=int, default=20_000_000)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--check", action="store_true", help="compare 200k reads with the CPU oracle (builds its table too)")
    ap.add_argument("--build", default="host", choices=["host", "device"], help="device: bbduk_build_table_device (needed beyond ~2e8 keys)")
    a = ap.parse_args()
    import torch
    from bbtools_amd import bbduk as B
    rng = np.random.default_rng(40)
    ref = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, a.ref_bases, dtype=np.uint8)].tobytes()
    if a.build == "device":
        t = time.time(); duk = B.BBDuk("k=31 hdist=0", refs=[ref], build="device"); t_index = 0.0; t_upload = time.time() - t
        gpu = duk.gpu; nkeys = gpu.table_size
    else:
        t = time.time(); host = B.HostIndex("k=31 hdist=0"); host.add_ref(ref); nkeys = host.build_index(); t_index = time.time() - t
        t = time.time(); gpu = B.BBDukGpu(host.params(0))
        rc = B.lib().bbduk_host_upload_index(host.h, gpu.h); gpu._check(rc, "upload_index"); t_upload = time.time() - t
    n_pairs = a.reads // 2; n = 2 * n_pairs
    sp = B.synth_params(4, contam=ref, contam_frac=0.01)
    d_bases = torch.empty(n * 150, dtype=torch.uint8, device="cuda"); d_off = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    B.synth_generate_device(sp, 0, n_pairs, d_bases, d_off, 0)
    d_a = torch.empty(n, dtype=torch.int32, device="cuda"); d_id = torch.empty_like(d_a); d_fl = torch.empty(n, dtype=torch.uint8, device="cuda")
    d_c = torch.zeros(gpu.counters_len, dtype=torch.int64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(1 + a.steps):
        d_c.zero_(); gpu.process_batch_device(d_bases, d_off, True, d_a, d_id, d_fl, d_c, st)
    torch.cuda.synchronize()
    ms = gpu.kernel_time_ms(a.steps)
    c = d_c.cpu().numpy()
    algo = n * (150 + 8 + 8 + 120 * 8) + 4 * int(c[4])
    out = {"workload": "kfilter k=31 hdist=0 vs %d-base random reference (%d keys, map %.2f GB HBM), %d reads 2x150bp, 1%% contaminated"
                       % (a.ref_bases, nkeys, gpu.table_bytes / 1e9, n),
           "Gbases_per_s": round(n * 150 / (ms * 1e-3) / 1e9, 2), "kernel_ms": round(ms, 3),
           "algorithmic_GBps": round(algo / (ms * 1e-3) / 1e9, 1), "frac_of_8TBps": round(algo / (ms * 1e-3) / 8e12, 4),
           "readsKFiltered": int(c[4]), "table_build": a.build, "host_index_s": round(t_index, 1), "upload_finalize_or_device_build_s": round(t_upload, 1)}
    if a.check:
        from oracle.oracle_ffi import Oracle
        o = Oracle(k=31); o.add_ref(ref)
        m = 200_000
        hb, hoff = B.synth_generate_host(sp, 0, m // 2)
        oa, oi, of = o.process_batch(hb, hoff, True, nthreads=os.cpu_count())
        out["parity_sample_ok"] = bool(np.array_equal(d_a[:m].cpu().numpy(), oa) and np.array_equal(d_id[:m].cpu().numpy(), oi)
                                       and np.array_equal(d_fl[:m].cpu().numpy(), of))
        out["sample_hits"] = int((oa > 0).sum())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
