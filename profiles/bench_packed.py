#!/usr/bin/env python3
"""
Secondary measurement (never bench.py's `value`): the packed boundary format (2-bit codes + undefined bits,
include/bbduk_gpu.h) against the ASCII one on the configs[1] workload.
  (i)  device-resident: dominant-kernel time and HBM bytes read per read for both formats, results compared;
  (ii) host buffers (what a JNI caller sees, PCIe inside the call): bbduk_ktrim_batch vs bbduk_ktrim_batch_packed.
One JSON line.
"""
import argparse, json, os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=20_000_000); ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--host-reads", type=int, default=20_000_000)
    ap.add_argument("--args", default="ktrim=r k=23 mink=11 hdist=1 ref=adapters")
    a = ap.parse_args()
    import torch
    from bbtools_amd import bbduk as B
    duk = B.BBDuk(a.args)
    n_pairs = a.reads // 2; n = 2 * n_pairs
    sp = B.synth_params(2)
    d_bases = torch.empty(n * 150, dtype=torch.uint8, device="cuda"); d_off = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    B.synth_generate_device(sp, 0, n_pairs, d_bases, d_off, 0)
    d_codes = torch.empty((n * 150 + 15) // 16 + 4, dtype=torch.int32, device="cuda")
    d_undef = torch.empty((n * 150 + 31) // 32 + 4, dtype=torch.int32, device="cuda")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); B.pack_bases_device(d_bases, d_codes, d_undef); e1.record(); torch.cuda.synchronize()
    pack_ms = e0.elapsed_time(e1)
    outs = {}
    res = {}
    for fmt in ("ascii", "packed"):
        d_a = torch.empty(n, dtype=torch.int32, device="cuda"); d_i = torch.empty(n, dtype=torch.int32, device="cuda")
        d_f = torch.empty(n, dtype=torch.uint8, device="cuda"); d_c = torch.zeros(duk.gpu.counters_len, dtype=torch.int64, device="cuda")
        for it in range(1 + a.steps):
            d_c.zero_()
            if fmt == "ascii":
                duk.gpu.process_batch_device(d_bases, d_off, True, d_a, d_i, d_f, d_c)
            else:
                duk.gpu.process_batch_packed_device(d_codes, d_undef, d_off, n * 150, True, d_a, d_i, d_f, d_c)
        torch.cuda.synchronize()
        ms = duk.gpu.kernel_time_ms(a.steps)
        outs[fmt] = (d_a.cpu().numpy(), d_i.cpu().numpy(), d_f.cpu().numpy(), d_c.cpu().numpy())
        res[fmt] = {"kernel_ms": round(ms, 3), "Gbases_per_s": round(n * 150 / ms / 1e6, 1),
                    "input_bytes_per_read": 150 + 8 if fmt == "ascii" else round(150 / 4 + 150 / 8 + 8, 2)}
    same = all(np.array_equal(x, y) for x, y in zip(outs["ascii"], outs["packed"]))
    del d_bases, d_codes, d_undef
    # host buffers
    hp = a.host_reads // 2; hn = 2 * hp
    hb, hoff = B.synth_generate_host(sp, 0, hp)
    t = time.perf_counter(); codes, undef = B.pack_bases_host(hb); host_pack_s = time.perf_counter() - t
    host = {}
    for fmt in ("ascii", "packed"):
        ts = []
        for it in range(1 + 3):
            t = time.perf_counter()
            r = duk.gpu.process_batch(hb, hoff, True) if fmt == "ascii" else duk.gpu.process_batch_packed(codes, undef, hoff, True)
            ts.append(time.perf_counter() - t)
        dt = min(ts[1:])
        host[fmt] = {"ms_per_call": round(dt * 1e3, 2), "Gbases_per_s": round(hn * 150 / dt / 1e9, 2), "results": r}
    hsame = all(np.array_equal(x, y) for x, y in zip(host["ascii"].pop("results"), host["packed"].pop("results")))
    print(json.dumps({"workload": "configs[1] (%s), %d reads device-resident / %d reads per host call" % (a.args, n, hn),
                      "device": res, "device_results_equal": bool(same), "pack_kernel_ms": round(pack_ms, 3),
                      "host_operator": host, "host_results_equal": bool(hsame),
                      "host_pack_single_thread_Gbases_per_s": round(hn * 150 / host_pack_s / 1e9, 3)}))


if __name__ == "__main__":
    main()
