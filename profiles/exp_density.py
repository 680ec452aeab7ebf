#!/usr/bin/env python3
"""Experiment (GPU box): kernel time of configs[1] / configs[2] and of the Seal benchmark against the map's bucket count
(BBDUK_HOOK_BUCKET_BITS: fewer buckets = more keys per 4-way bucket = a smaller fingerprint array and record array).
One JSON line per point.  python profiles/exp_density.py [--reads 20000000]"""
import argparse, json, os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "profiles"))


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--reads", type=int, default=20_000_000); ap.add_argument("--seal-pairs", type=int, default=1_000_000)
    a = ap.parse_args()
    import torch
    from bbtools_amd import bbduk as B
    phix = B.read_fasta(os.path.join(ROOT, "data", "phix2.fa.gz"))[0][1]
    n_pairs = a.reads // 2; n = 2 * n_pairs
    d_bases = torch.empty(n * 150, dtype=torch.uint8, device="cuda"); d_off = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    d_a = torch.empty(n, dtype=torch.int32, device="cuda"); d_id = torch.empty_like(d_a); d_fl = torch.empty(n, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for name, args, seed, kw, bits in (("C2", "ktrim=r k=23 mink=11 hdist=1 ref=adapters", 2, {}, (0, 17, 19)),
                                       ("C3", "k=31 hdist=1 ref=phix", 3, dict(contam=phix, contam_frac=0.01), (0, 18, 20)),
                                       ("C5", "ktrim=r k=23 hdist=1 ref=adapters", 5, {}, (0, 17, 19))):
        ref = None
        for bb in bits:
            try:
                duk = B.BBDuk(args, hooks={B.HOOK_BUCKET_BITS: bb} if bb else None)
                B.synth_generate_device(B.synth_params(seed, **kw), 0, n_pairs, d_bases, d_off, 0)
                d_c = torch.zeros(duk.gpu.counters_len, dtype=torch.int64, device="cuda")
                for _ in range(4):
                    d_c.zero_(); duk.gpu.process_batch_device(d_bases, d_off, True, d_a, d_id, d_fl, d_c, st)
                torch.cuda.synchronize()
                ms = duk.gpu.kernel_time_ms(3)
                sig = (int(d_a.sum().item()), int(d_id.sum().item()), int(d_fl.sum().item()), d_c.cpu().numpy()[:10].tolist())
                ref = ref or sig
                print(json.dumps({"config": name, "bucket_bits": bb or "default", "keys": int(duk.stored_kmers), "table_bytes": int(duk.gpu.table_bytes),
                                  "kernel_ms": round(ms, 3), "Gbases_per_s": round(n * 150 / ms / 1e6, 1), "same_answers_as_default": sig == ref}), flush=True)
                duk.close()
            except Exception as e:
                print(json.dumps({"config": name, "bucket_bits": bb, "error": repr(e)}), flush=True)
    del d_bases, d_off, d_a, d_id, d_fl
    torch.cuda.empty_cache()
    import bench_seal
    for bb in (0, 22, 21, 24):
        try:
            r = bench_seal.run(pairs=a.seal_pairs, check=0, bucket_bits=bb)
            r["bucket_bits"] = bb or "default"
            print(json.dumps(r), flush=True)
        except Exception as e:
            print(json.dumps({"config": "seal", "bucket_bits": bb, "error": repr(e)}), flush=True)


if __name__ == "__main__":
    main()
