#!/usr/bin/env python3
"""Secondary measurement: bbduk_long_kernel (reads beyond BBDUK_MAX_READ_LEN, chunked scan) on random long reads with adapter
pieces, device-resident; kernel time by HIP events around the three launches of the operator.  One JSON line per case."""
import argparse, json, os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--reads", type=int, default=20000); ap.add_argument("--len", type=int, default=50000)
    a = ap.parse_args()
    import torch
    from bbtools_amd import bbduk as B
    rng = np.random.default_rng(3)
    n, L = a.reads, a.len
    hb = rng.choice(np.frombuffer(b"ACGT", np.uint8), n * L)
    ad = np.frombuffer(B.TRUSEQ_R1, np.uint8)
    for i in range(0, n, 7):                                      # an adapter somewhere in every seventh read
        p = int(rng.integers(0, L - len(ad))); hb[i * L + p:i * L + p + len(ad)] = ad
    off = np.arange(n + 1, dtype=np.int64) * L
    for args in ("ktrim=r k=23 mink=11 hdist=1 ref=adapters", "ktrim=l k=23 mink=11 hdist=1 ref=adapters", "k=31 hdist=1 ref=phix"):
        duk = B.BBDuk(args)
        d_b = torch.from_numpy(hb).cuda(); d_o = torch.from_numpy(off).cuda()
        d_a = torch.empty(n, dtype=torch.int32, device="cuda"); d_i = torch.empty_like(d_a); d_f = torch.empty(n, dtype=torch.uint8, device="cuda")
        d_c = torch.zeros(duk.gpu.counters_len, dtype=torch.int64, device="cuda")
        ts = []
        for _ in range(4):
            d_c.zero_(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); duk.gpu.process_batch_device(d_b, d_o, False, d_a, d_i, d_f, d_c); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ms = min(ts[1:])
        c = d_c.cpu().numpy()
        print(json.dumps({"args": args, "reads": n, "read_len": L, "operator_ms": round(ms, 2), "Gbases_per_s": round(n * L / ms / 1e6, 1),
                          "readsKTrimmed": int(c[2]), "readsKFiltered": int(c[4]), "status": int(c[15])}))
        duk.close()


if __name__ == "__main__":
    main()
