#!/usr/bin/env python3
"""Kernel time of every BASELINE configuration that fits the cache-resident regime, at --reads reads (secondary
measurement; bench.py reports configs[1] only).  One JSON line per config."""
import argparse, json, os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
CONFIGS = {
    "C1": ("ktrim=r k=23 hdist=0 ref=adapters", {}),
    "C2": ("ktrim=r k=23 mink=11 hdist=1 ref=adapters", {}),
    "C3": ("k=31 hdist=1 ref=phix", {"phix": 0.01}),
    "C5": ("ktrim=r k=23 hdist=1 ref=adapters", {}),
}


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--reads", type=int, default=20_000_000); ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--only", default=None, help="comma list of config names"); ap.add_argument("--args", default=None, help="override the BBDuk argument string (experiments)")
    a = ap.parse_args()
    import torch
    from bbtools_amd import bbduk as B
    phix = B.read_fasta(os.path.join(ROOT, "data", "phix2.fa.gz"))[0][1]
    n_pairs = a.reads // 2; n = 2 * n_pairs
    for name, (args, extra) in CONFIGS.items():
        if a.only and name not in a.only.split(","): continue
        if a.args: args = a.args
        duk = B.BBDuk(args)
        kw = dict(contam=phix, contam_frac=extra["phix"]) if "phix" in extra else {}
        sp = B.synth_params({"C1": 1, "C2": 2, "C3": 3, "C5": 5}[name], **kw)
        d_bases = torch.empty(n * 150, dtype=torch.uint8, device="cuda"); d_off = torch.empty(n + 1, dtype=torch.int64, device="cuda")
        B.synth_generate_device(sp, 0, n_pairs, d_bases, d_off, 0)
        d_a = torch.empty(n, dtype=torch.int32, device="cuda"); d_id = torch.empty_like(d_a); d_fl = torch.empty(n, dtype=torch.uint8, device="cuda")
        d_c = torch.zeros(duk.gpu.counters_len, dtype=torch.int64, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(1 + a.steps):
            d_c.zero_(); duk.gpu.process_batch_device(d_bases, d_off, paired=True, d_a=d_a, d_id=d_id, d_fl=d_fl, d_counters=d_c, stream_ptr=st)
        torch.cuda.synchronize()
        ms = duk.gpu.kernel_time_ms(a.steps); c = d_c.cpu().numpy()
        print(json.dumps({"config": name, "args": args, "reads": n, "keys": int(duk.stored_kmers), "kernel_ms": round(ms, 3),
                          "Gbases_per_s": round(n * 150 / (ms * 1e-3) / 1e9, 1), "readsKTrimmed": int(c[2]), "readsKFiltered": int(c[4])}))
        duk.close(); del d_bases, d_off, d_a, d_id, d_fl


if __name__ == "__main__":
    main()
