#!/usr/bin/env python3
"""Experiment (GPU box): kfilter k=31 hdist=0 against random genomes of 1-30 Mbases (the reference's own sizing example is E. coli, 4.6 M
31-mers: docs/guides/BBDukGuide.txt:34): the cache-resident layout these sizes take by default vs the big layout's minimizer lines
(BBDUK_HOOK_BIG_LAYOUT), device-built maps, 2x150 bp reads with 1 % of the pairs drawn from the genome."""
import argparse, json, os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--reads", type=int, default=20_000_000)
    ap.add_argument("--genomes", default="500000,1000000,2000000,4600000,10000000,30000000"); ap.add_argument("--args", default="k=31 hdist=0"); ap.add_argument("--layouts", default="default,big"); ap.add_argument("--read-len", type=int, default=150); ap.add_argument("--build", default="device"); ap.add_argument("--lds-bits", type=int, default=-1, help="BBDUK_HOOK_LDS_BITS on top of the layout hook (0 = no LDS filter)"); ap.add_argument("--contam-frac", type=float, default=0.01)
    ap.add_argument("--big-load", type=int, default=0, help="BBDUK_HOOK_BIG_LOAD: keys per 100 slots"); ap.add_argument("--lib", default="", help="experiment build under bbtools_amd/ (ab_tsw: stage deletion)"); ap.add_argument("--masks", default="0", help="timing masks (a+b = bits), comma-separated: big layout stream scan: 3 no scan, 5 no staging, 7 no decide, 11 no walk, 12 no N pass, 13 no gathers, 16 no back half, 19 no minimizer")
    a = ap.parse_args()
    import torch
    from bbtools_amd import bbduk as B
    if a.lib: B.use_library(os.path.join(ROOT, "bbtools_amd", a.lib + ".so"))
    RL = a.read_len; n_pairs = a.reads * 150 // RL // 2; n = 2 * n_pairs
    d_bases = torch.empty(n * RL, dtype=torch.uint8, device="cuda"); d_off = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    d_a = torch.empty(n, dtype=torch.int32, device="cuda"); d_id = torch.empty_like(d_a); d_fl = torch.empty(n, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for G in [int(x) for x in a.genomes.split(",")]:
        genome = np.frombuffer(b"ACGT", np.uint8)[np.random.default_rng(G).integers(0, 4, G)].tobytes()
        B.synth_generate_device(B.synth_params(3, read_len=RL, ins_min=max(50, RL // 3), ins_max=max(350, 2 * RL), contam=genome, contam_frac=a.contam_frac), 0, n_pairs, d_bases, d_off, 0)
        ref = None
        for layout in a.layouts.split(","):
          for m in a.masks.split(","):
            bits = sum(1 << int(b) for b in m.split("+")) if m != "0" else 0
            try:
                duk = B.BBDuk(a.args, refs=[genome], build=a.build, hooks={**{"big": {B.HOOK_BIG_LAYOUT: 1}, "big52": {B.HOOK_BIG_LAYOUT: 2}, "bigpair": {B.HOOK_BIG_LAYOUT: 1, B.HOOK_PAIR_SCAN: 1}, "seed": {B.HOOK_SEED_LAYOUT: 1}, "cache": {B.HOOK_BIG_LAYOUT: -1}}.get(layout, {}), **({B.HOOK_LDS_BITS: a.lds_bits} if a.lds_bits >= 0 else {}), **({B.HOOK_TIMING_MASK: bits} if bits else {}), **({B.HOOK_BIG_LOAD: a.big_load} if a.big_load else {})} or None)
                d_c = torch.zeros(duk.gpu.counters_len, dtype=torch.int64, device="cuda")
                for _ in range(4):
                    d_c.zero_(); duk.gpu.process_batch_device(d_bases, d_off, True, d_a, d_id, d_fl, d_c, st)
                torch.cuda.synchronize()
                ms = duk.gpu.kernel_time_ms(3)
                sig = (int(d_a.sum().item()), int(d_id.sum().item()), int(d_fl.sum().item()), d_c.cpu().numpy()[:10].tolist())
                ref = ref or sig
                print(json.dumps({"genome_bases": G, "layout": layout, "mask": m, "build": a.build, "spilled": int(duk.gpu.table_spilled), "keys": int(duk.stored_kmers), "table_bytes": int(duk.gpu.table_bytes),
                                  "kernel_ms": round(ms, 3), "read_len": RL, "Gbases_per_s": round(n * RL / ms / 1e6, 1), "readsKFiltered": int(sig[3][4]), "same_answers": sig == ref}), flush=True)
                duk.close()
            except Exception as e:
                print(json.dumps({"genome_bases": G, "layout": layout, "mask": m, "error": repr(e)}), flush=True)


if __name__ == "__main__":
    main()
