#!/usr/bin/env python3
"""Workloads away from the benchmark's: does an operator fall off a cliff when every read hits, reads are very short, undefined bases are common, or a
contaminant run has one scaffold?  (Round 5: Seal lost 3.5x with four scaffolds instead of 64 -- atomics on a few counters -- and nobody had looked.)
Kernel Gbases/s by HIP events around the device operator, 4 M pairs each.  One JSON line per case."""
import json, os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from bbtools_amd import bbduk as B

PHIX = B.read_fasta(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "data", "phix2.fa.gz"))[0][1]
CASES = [
    ("ktrim=r k=23 mink=11 hdist=1 ref=adapters", dict(), "the benchmark's reads"),
    ("ktrim=r k=23 mink=11 hdist=1 ref=adapters", dict(ins_min=20, ins_max=100), "every read runs into its adapter"),
    ("ktrim=r k=23 mink=11 hdist=1 ref=adapters", dict(ins_min=400, ins_max=600), "no read holds an adapter"),
    ("ktrim=r k=23 mink=11 hdist=1 ref=adapters", dict(n_rate=0.05), "5 % undefined bases"),
    ("ktrim=r k=23 mink=11 hdist=1 ref=adapters", dict(read_len=36, ins_min=20, ins_max=80), "2x36"),
    ("ktrim=r k=23 mink=11 hdist=1 ref=adapters", dict(read_len=50, ins_min=30, ins_max=120), "2x50"),
    ("k=31 hdist=1 ref=phix", dict(contam=PHIX, contam_frac=0.01), "1 % of the pairs from phiX (configs[2])"),
    ("k=31 hdist=1 ref=phix", dict(contam=PHIX, contam_frac=1.0), "every pair from phiX (one scaffold takes every hit)"),
    ("k=31 hdist=1 ref=phix", dict(contam=PHIX, contam_frac=0.5), "half the pairs from phiX"),
    ("ktrim=n k=23 mink=11 hdist=1 ref=adapters", dict(ins_min=20, ins_max=100), "ktrim=n, every read runs into its adapter"),
    ("ktrim=rl k=23 mink=11 hdist=1 ref=adapters", dict(ins_min=20, ins_max=100), "ktrim=rl, every read runs into its adapter"),
    ("k=31 hdist=1 ref=phix fbm=t", dict(contam=PHIX, contam_frac=1.0), "findbestmatch, every pair from phiX"),
]
n = 8_000_000
L = B.lib()
ONLY = os.environ.get("BENCH_CLIFFS_ONLY", "")                   # substring of the case's description
for args, kw, what in CASES:
    if ONLY and ONLY not in what: continue
    RL = kw.get("read_len", 150)
    sp = B.synth_params(2, **kw)
    d_bases = torch.empty(n * RL, dtype=torch.uint8, device="cuda"); d_off = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    B.synth_generate_device(sp, 0, n // 2, d_bases, d_off, 0)
    duk = B.BBDuk(args)
    d_a = torch.empty(n, dtype=torch.int32, device="cuda"); d_b = torch.empty_like(d_a); d_i = torch.empty_like(d_a); d_f = torch.empty(n, dtype=torch.uint8, device="cuda")
    d_m = torch.zeros(n * RL // 32 + 8, dtype=torch.int32, device="cuda")
    d_c = torch.zeros(duk.gpu.counters_len, dtype=torch.int64, device="cuda")
    h = duk.gpu.h; ts = []
    for _ in range(4):
        d_c.zero_(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
        if "ktrim=n" in args: rc = L.bbduk_kmask_batch_device(h, d_bases.data_ptr(), d_off.data_ptr(), n, n * RL, 1, d_a.data_ptr(), d_i.data_ptr(), d_f.data_ptr(), d_m.data_ptr(), d_c.data_ptr(), None)
        elif "ktrim=rl" in args: rc = L.bbduk_ktrimtips_batch_device(h, d_bases.data_ptr(), d_off.data_ptr(), n, n * RL, 1, d_a.data_ptr(), d_b.data_ptr(), d_i.data_ptr(), d_f.data_ptr(), d_c.data_ptr(), None)
        elif "ktrim=r" in args: rc = L.bbduk_ktrim_batch_device(h, d_bases.data_ptr(), d_off.data_ptr(), n, n * RL, 1, d_a.data_ptr(), d_i.data_ptr(), d_f.data_ptr(), d_c.data_ptr(), None)
        else: rc = L.bbduk_kfilter_batch_device(h, d_bases.data_ptr(), d_off.data_ptr(), n, n * RL, 1, d_a.data_ptr(), d_i.data_ptr(), d_f.data_ptr(), d_c.data_ptr(), None)
        e1.record(); torch.cuda.synchronize(); assert rc == 0, rc
        ts.append(e0.elapsed_time(e1))
    ms = min(ts[1:]); c = d_c.cpu().numpy()
    print(json.dumps({"args": args, "reads": what, "read_len": RL, "operator_ms": round(ms, 3), "Gbases_per_s": round(n * RL / ms / 1e6, 1), "readsKTrimmed": int(c[2]), "readsKFiltered": int(c[4])}), flush=True)
    duk.close(); del d_bases, d_off
