#!/usr/bin/env python3
"""
Secondary measurement (never bench.py's `value`): the PCIe-inclusive rate of the host-buffer operator
bbduk_ktrim_batch -- what a JNI caller sees: host bases+offsets in, H2D, kernel, D2H of the three
result arrays, all inside the call.  configs[1] workload, --reads reads per call.  One JSON line.
"""
import argparse, json, os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=20_000_000); ap.add_argument("--steps", type=int, default=3)
    a = ap.parse_args()
    from bbtools_amd import bbduk as B
    duk = B.BBDuk("ktrim=r k=23 mink=11 hdist=1 ref=adapters")
    n_pairs = a.reads // 2; n = 2 * n_pairs
    sp = B.synth_params(2)
    hb, hoff = B.synth_generate_host(sp, 0, n_pairs)
    times = []
    for i in range(1 + a.steps):
        t = time.perf_counter(); r = duk.gpu.process_batch(hb, hoff, True); times.append(time.perf_counter() - t)
    dt = min(times[1:])
    # the same call with the caller's buffers in pinned memory (what a JNI caller gets from bbduk_pinned_malloc / a registered
    # direct ByteBuffer): the DMA engine reads them directly, no on-the-fly page locking
    import torch
    pb = torch.from_numpy(hb).pin_memory().numpy(); po = torch.from_numpy(hoff).pin_memory().numpy()
    ptimes = []
    for i in range(1 + a.steps):
        t = time.perf_counter(); r2 = duk.gpu.process_batch(pb, po, True); ptimes.append(time.perf_counter() - t)
    pdt = min(ptimes[1:])
    c = duk.gpu.counters()
    print(json.dumps({"workload": "bbduk_ktrim_batch (host buffers, PCIe inside the call), configs[1], %d reads per call" % n,
                      "Gbases_per_s": round(n * 150 / dt / 1e9, 2), "ms_per_call": round(dt * 1e3, 2),
                      "bytes_in_per_read": 158, "bytes_out_per_read": 9, "host_to_device_GBps": round(n * 158 / dt / 1e9, 2),
                      "kernel_ms": round(duk.gpu.kernel_time_ms(1), 3), "readsKTrimmed_all_calls": int(c[2]),
                      "pinned_caller_buffers": {"ms_per_call": round(pdt * 1e3, 2), "Gbases_per_s": round(n * 150 / pdt / 1e9, 2),
                                                "host_to_device_GBps": round(n * 158 / pdt / 1e9, 2),
                                                "results_equal": bool(all((x == y).all() for x, y in zip(r, r2)))}}))


if __name__ == "__main__":
    main()
