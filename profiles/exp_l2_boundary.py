#!/usr/bin/env python3
"""Experiment (GPU box): kfilter k=31 hdist=1 against random genomes whose key counts sit just above a power of two -- the sizing rule
(one bucket per key, rounded up) then doubles the fingerprint array across the 4 MB of an XCD's L2.  Default sizing vs one bit less."""
import argparse, json, os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--reads", type=int, default=20_000_000)
    ap.add_argument("--genomes", default="5700,6186,7000,8000,11500,12500,14000"); ap.add_argument("--args", default="k=31 hdist=1")
    a = ap.parse_args()
    import torch
    from bbtools_amd import bbduk as B
    n_pairs = a.reads // 2; n = 2 * n_pairs
    d_bases = torch.empty(n * 150, dtype=torch.uint8, device="cuda"); d_off = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    d_a = torch.empty(n, dtype=torch.int32, device="cuda"); d_id = torch.empty_like(d_a); d_fl = torch.empty(n, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for G in [int(x) for x in a.genomes.split(",")]:
        genome = np.frombuffer(b"ACGT", np.uint8)[np.random.default_rng(G).integers(0, 4, G)].tobytes()
        B.synth_generate_device(B.synth_params(3, contam=genome, contam_frac=0.01), 0, n_pairs, d_bases, d_off, 0)
        ref = None
        for delta in (0, -1, -2):
            duk = B.BBDuk(a.args, refs=[genome])
            keys = int(duk.stored_kmers); duk.close()
            bits = max(10, int(np.ceil(np.log2(keys)))) + delta
            duk = B.BBDuk(a.args, refs=[genome], hooks={B.HOOK_BUCKET_BITS: bits} if delta else None)
            d_c = torch.zeros(duk.gpu.counters_len, dtype=torch.int64, device="cuda")
            for _ in range(4):
                d_c.zero_(); duk.gpu.process_batch_device(d_bases, d_off, True, d_a, d_id, d_fl, d_c, st)
            torch.cuda.synchronize()
            ms = duk.gpu.kernel_time_ms(3)
            sig = (int(d_a.sum().item()), int(d_id.sum().item()), int(d_fl.sum().item()))
            ref = ref or sig
            print(json.dumps({"genome_bases": G, "keys": keys, "bucket_bits": bits, "delta": delta, "keys_per_bucket": round(keys / 2 ** bits, 2), "tag_bytes": 8 << bits,
                              "kernel_ms": round(ms, 3), "Gbases_per_s": round(n * 150 / ms / 1e6, 1), "same_answers": sig == ref}), flush=True)
            duk.close()


if __name__ == "__main__":
    main()
