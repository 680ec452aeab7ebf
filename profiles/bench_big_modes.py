#!/usr/bin/env python3
"""Round 5: the operators that had no big-layout form -- ksplit, ktrim=n, ktrim=rl, findbestmatch, kfilter with mkf -- against a map
far beyond what the cache-resident layout indexes: a --ref-bases synthetic reference (default 2e9 bases = 2e9 keys, ~90 GB) generated on the device chunk
by chunk and streamed into bbduk_build_begin / _add_device / _end, as profiles/bench_c4_full.py does for configs[3].  Per operator: the map is accepted,
operator Gbases/s on --reads 2x150 reads (1 % of the pairs drawn from the reference), and the first --check-reads reads against the CPU oracle, which
indexes the part of the reference the contaminated pairs come from.  One JSON line per operator."""
import argparse, json, os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np

MODES = (("ksplit=t k=31", dict(k=31, ksplit=1), "ksplit"), ("ktrim=n k=31", dict(k=31, ktrimN=1), "kmask"), ("ktrim=rl k=31", dict(k=31, ktrimRight=1, ktrimLeft=1), "tips"),
         ("k=31 fbm", dict(k=31, findBestMatch=1), "fbm"), ("k=31 mkf=0.1", dict(k=31, minKmerFraction=0.1), "mkf"),
         ("k=40", dict(k=31, kbig=40, maskMiddle=0), "k40"))                   # k > 31 joined the big layout later in round 5 (bbduk_bigs_kbig.hip)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref-bases", type=float, default=2e9); ap.add_argument("--scaffold-bases", type=int, default=10_000_000); ap.add_argument("--chunk-scaffolds", type=int, default=50)
    ap.add_argument("--reads", type=int, default=20_000_000); ap.add_argument("--check-reads", type=int, default=50_000); ap.add_argument("--oracle-bases", type=int, default=10_000_000)
    ap.add_argument("--modes", default="ksplit,kmask")
    a = ap.parse_args()
    import torch
    from bbtools_amd import bbduk as B
    from oracle.oracle_ffi import Oracle
    L = B.lib()
    n_scaf = max(1, int(a.ref_bases) // a.scaffold_bases); ref_bases = n_scaf * a.scaffold_bases
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device="cuda")
    n = a.reads & ~1
    d_bases = torch.empty(n * 150, dtype=torch.uint8, device="cuda"); d_off = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    for args, okw, mode in MODES:
        if mode not in a.modes.split(","): continue
        host = B.HostIndex(args); p = host.params(0); p.numScaffolds = n_scaf + 1
        gpu = B.BBDukGpu(p)
        t0 = time.perf_counter()
        gpu.build_begin(ref_bases, 0, 0)
        contam = None
        for c0 in range(0, n_scaf, a.chunk_scaffolds):
            ns = min(a.chunk_scaffolds, n_scaf - c0)
            g = torch.Generator(device="cuda"); g.manual_seed(40 + c0)
            chunk = lut[torch.randint(0, 4, (ns * a.scaffold_bases,), generator=g, device="cuda", dtype=torch.int64)]
            gpu.build_add_device(chunk, np.arange(ns + 1, dtype=np.int64) * a.scaffold_bases, c0 + 1)
            if contam is None: contam = chunk[:min(a.oracle_bases, a.scaffold_bases)].cpu().numpy().tobytes()
            del chunk
        gpu.build_end()
        out = {"mode": mode, "args": args + " vs a %d-base synthetic reference (%d scaffolds), streamed device build" % (ref_bases, n_scaf), "keys": int(gpu.table_size),
               "table_bytes": int(gpu.table_bytes), "spilled_keys": int(gpu.table_spilled), "build_s": round(time.perf_counter() - t0, 2)}
        sp = B.synth_params(4, contam=contam, contam_frac=0.01)
        B.synth_generate_device(sp, 0, n // 2, d_bases, d_off, 0)
        d_a = torch.empty(n, dtype=torch.int32, device="cuda"); d_b = torch.empty_like(d_a); d_c2 = torch.empty_like(d_a); d_i = torch.empty_like(d_a); d_f = torch.empty(n, dtype=torch.uint8, device="cuda")
        d_m = torch.zeros(n * 150 // 32 + 8, dtype=torch.int32, device="cuda") if mode == "kmask" else None
        d_ctr = torch.zeros(gpu.counters_len, dtype=torch.int64, device="cuda")
        paired = 0 if mode == "ksplit" else 1
        ts = []
        for _ in range(4):
            d_ctr.zero_()
            if d_m is not None: d_m.zero_()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
            if mode == "kmask":
                rc = L.bbduk_kmask_batch_device(gpu.h, d_bases.data_ptr(), d_off.data_ptr(), n, n * 150, 1, d_a.data_ptr(), d_i.data_ptr(), d_f.data_ptr(), d_m.data_ptr(), d_ctr.data_ptr(), None)
            elif mode == "tips":
                rc = L.bbduk_ktrimtips_batch_device(gpu.h, d_bases.data_ptr(), d_off.data_ptr(), n, n * 150, 1, d_a.data_ptr(), d_b.data_ptr(), d_i.data_ptr(), d_f.data_ptr(), d_ctr.data_ptr(), None)
            elif mode == "ksplit":
                rc = L.bbduk_ksplit_batch_device(gpu.h, d_bases.data_ptr(), d_off.data_ptr(), n, n * 150, d_a.data_ptr(), d_b.data_ptr(), d_c2.data_ptr(), d_i.data_ptr(), d_f.data_ptr(), d_ctr.data_ptr(), None)
            else:
                rc = L.bbduk_kfilter_batch_device(gpu.h, d_bases.data_ptr(), d_off.data_ptr(), n, n * 150, 1, d_a.data_ptr(), d_i.data_ptr(), d_f.data_ptr(), d_ctr.data_ptr(), None)
            e1.record(); torch.cuda.synchronize()
            assert rc == 0, (rc, gpu.last_error() if hasattr(gpu, "last_error") else "")
            ts.append(e0.elapsed_time(e1))
        ms = min(ts[1:]); c = d_ctr.cpu().numpy()
        out.update({"reads": n, "operator_ms": round(ms, 2), "Gbases_per_s": round(n * 150 / ms / 1e6, 1), "readsKTrimmed": int(c[2]), "readsKFiltered": int(c[4]), "status": int(c[15])})
        m = min(n, a.check_reads) & ~1
        if m:
            o = Oracle(**okw); o.add_ref(contam)
            hb, hoff = B.synth_generate_host(sp, 0, m // 2)
            if mode == "ksplit":
                ox, oi, of, ol, orr = o.process_batch_split(hb, hoff, nthreads=os.cpu_count())
                got = [d_a[:m].cpu().numpy(), d_i[:m].cpu().numpy(), d_f[:m].cpu().numpy(), d_b[:m].cpu().numpy(), d_c2[:m].cpu().numpy()]; want = [ox, oi, of, ol, orr]
            elif mode == "kmask":
                oa, oi, of, om = o.process_batch_mask(hb, hoff, True, nthreads=os.cpu_count())
                nw = m * 150 // 32
                got = [d_a[:m].cpu().numpy(), d_i[:m].cpu().numpy(), d_f[:m].cpu().numpy()]; want = [oa, oi, of]
                out["checked_mask_words_differ"] = int((d_m[:nw].cpu().numpy().view(np.uint32) != om[:nw]).sum())
            elif mode == "tips":
                xr, xl, oi, of = o.process_batch_tips(hb, hoff, True, nthreads=os.cpu_count())
                got = [d_a[:m].cpu().numpy(), d_b[:m].cpu().numpy(), d_i[:m].cpu().numpy(), d_f[:m].cpu().numpy()]; want = [xr, xl, oi, of]
            else:
                oa, oi, of = o.process_batch(hb, hoff, True, nthreads=os.cpu_count())
                got = [d_a[:m].cpu().numpy(), d_i[:m].cpu().numpy(), d_f[:m].cpu().numpy()]; want = [oa, oi, of]
            bad = np.zeros(m, bool)
            for g_, w_ in zip(got, want): bad |= (g_ != w_)
            out["checked_reads"] = int(m); out["checked_reads_differ"] = int(bad.sum()); out["checked_reads_hit"] = int((want[0] > 0).sum())
        print(json.dumps(out), flush=True)
        gpu.close(); host.close()
        del d_a, d_b, d_c2, d_i, d_f, d_m
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
