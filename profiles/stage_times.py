#!/usr/bin/env python3
"""Stage deletion on the configs[1] launch: kernel ms per timing mask (a -DBBDUK_TIMING_SWITCHES build: bbtools_amd/ab_tsw.so, made by
bbtools_amd._build.build_timing_variant()).  Results of a masked run are wrong by construction; only the time is read.
Bits: 3 no main scan, 4 no short k-mers, 5 no staging loads, 7 no decide stage; stream kernel: 10 no resolve of flagged blocks, 11 no
candidate verification, 13 no fingerprint gathers (tag words faked from the hash), 14 no presence-filter reads.
usage: python profiles/stage_times.py [--reads N] [--read-len L] [--pair-scan] mask [mask ...]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from bbtools_amd import bbduk as B

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=20_000_000); ap.add_argument("--read-len", type=int, default=150)
ap.add_argument("--pair-scan", action="store_true"); ap.add_argument("--lib", default="ab_tsw")
ap.add_argument("--args", default="ktrim=r k=23 mink=11 hdist=1 ref=adapters")
ap.add_argument("--contam-phix", type=float, default=0.0, help="share of the pairs drawn from phiX (BASELINE configs[2]'s workload: 0.01)")
ap.add_argument("--seed", type=int, default=2); ap.add_argument("--bucket-bits", type=int, default=0); ap.add_argument("--lds-bits", type=int, default=-1)
ap.add_argument("--genome", type=int, default=0, help="bases of a random genome as the reference, built on the device (big-layout maps); 1 %% of the pairs drawn from it")
ap.add_argument("--big-layout", type=int, default=0, help="BBDUK_HOOK_BIG_LAYOUT (3: the wide candidate values of maps beyond 2^31 keys at any size)")
ap.add_argument("--n-rate", type=float, default=0.001)
ap.add_argument("masks", nargs="*", default=["0"])
a = ap.parse_args()
B.use_library(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bbtools_amd", a.lib + ".so"))
n, L = a.reads & ~1, a.read_len
d_bases = torch.empty(n * L, dtype=torch.uint8, device="cuda"); d_off = torch.empty(n + 1, dtype=torch.int64, device="cuda")
kw = dict(contam=B.read_fasta(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "data", "phix2.fa.gz"))[0][1], contam_frac=a.contam_phix) if a.contam_phix > 0 else {}
genome = None
if a.genome:
    import numpy as np
    genome = np.frombuffer(b"ACGT", np.uint8)[np.random.default_rng(77).integers(0, 4, a.genome)].tobytes()
    kw = dict(contam=genome[:100_000], contam_frac=0.01); a.seed = 9
kw["n_rate"] = a.n_rate
B.synth_generate_device(B.synth_params(a.seed, read_len=L, ins_min=L // 3, ins_max=L * 7 // 3, **kw) if not a.genome else B.synth_params(a.seed, **kw), 0, n // 2, d_bases, d_off, 0)
d_a = torch.empty(n, dtype=torch.int32, device="cuda"); d_id = torch.empty_like(d_a); d_fl = torch.empty(n, dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for m in a.masks:
    bits = sum(1 << int(b) for b in m.split("+")) if m != "0" else 0
    hooks = {B.HOOK_TIMING_MASK: bits} if bits else {}          # (mask 0 also runs on the product build)
    if a.pair_scan:
        hooks[B.HOOK_PAIR_SCAN] = 1
    if a.bucket_bits:
        hooks[B.HOOK_BUCKET_BITS] = a.bucket_bits
    if a.lds_bits >= 0:
        hooks[B.HOOK_LDS_BITS] = a.lds_bits
    if a.big_layout:
        hooks[B.HOOK_BIG_LAYOUT] = a.big_layout
    duk = B.BBDuk(a.args, hooks=hooks, refs=[genome], build="device") if genome is not None else B.BBDuk(a.args, hooks=hooks)
    d_c = torch.zeros(duk.gpu.counters_len, dtype=torch.int64, device="cuda")
    kmask = duk.gpu.params.mode == B.MODE_KMASK
    tips = "ktrim=rl" in a.args
    split = duk.gpu.params.mode == B.MODE_KSPLIT
    d_r2 = torch.empty(n, dtype=torch.int32, device="cuda") if split else None
    d_l = torch.empty(n, dtype=torch.int32, device="cuda") if (tips or split) else None
    if kmask:
        d_m = torch.zeros(n * L // 32 + 8, dtype=torch.int32, device="cuda")
    for _ in range(5):
        d_c.zero_()
        if kmask:
            rc = B.lib().bbduk_kmask_batch_device(duk.gpu.h, d_bases.data_ptr(), d_off.data_ptr(), n, n * L, 1, d_a.data_ptr(), d_id.data_ptr(), d_fl.data_ptr(), d_m.data_ptr(), d_c.data_ptr(), None)
            assert rc == 0
        elif split:
            rc = B.lib().bbduk_ksplit_batch_device(duk.gpu.h, d_bases.data_ptr(), d_off.data_ptr(), n, n * L, d_a.data_ptr(), d_l.data_ptr() if d_l is not None else d_r2.data_ptr(), d_r2.data_ptr(), d_id.data_ptr(), d_fl.data_ptr(), d_c.data_ptr(), None)
            assert rc == 0
        elif tips:
            rc = B.lib().bbduk_ktrimtips_batch_device(duk.gpu.h, d_bases.data_ptr(), d_off.data_ptr(), n, n * L, 1, d_a.data_ptr(), d_l.data_ptr(), d_id.data_ptr(), d_fl.data_ptr(), d_c.data_ptr(), None)
            assert rc == 0
        else:
            duk.gpu.process_batch_device(d_bases, d_off, True, d_a, d_id, d_fl, d_c, st)
    torch.cuda.synchronize()
    ms = duk.gpu.kernel_time_ms(4)
    print(json.dumps({"bits": m, "mask": bits, "pair_scan": a.pair_scan, "reads": n, "read_len": L, "kernel_ms": round(ms, 3), "Gbases_per_s": round(n * L / ms / 1e6, 1)}), flush=True)
    duk.close()
