#!/usr/bin/env python3
"""Secondary measurement: the configs[1] flags on reads of other lengths than 2x150 (2x100, 2x151, 2x250, 2x300), device-resident, kernel time
from the library's HIP events; ~3 Gbases per launch each.  One JSON line per read length, with the counters' trimmed-read share."""
import sys, json, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from bbtools_amd import bbduk as B
import argparse
ap = argparse.ArgumentParser(); ap.add_argument("--lib", default=None); ap.add_argument("--bases", type=float, default=3e9)
ap.add_argument("lens", nargs="*", type=int, default=[100, 150, 151, 250, 300])
a = ap.parse_args()
if a.lib: B.use_library(a.lib)
args = "ktrim=r k=23 mink=11 hdist=1 ref=adapters"
duk = B.BBDuk(args)
st = torch.cuda.current_stream().cuda_stream
for L in a.lens:
    n = (int(a.bases) // L) & ~1
    d_bases = torch.empty(n * L, dtype=torch.uint8, device="cuda"); d_off = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    B.synth_generate_device(B.synth_params(2, read_len=L, ins_min=L // 3, ins_max=L * 7 // 3), 0, n // 2, d_bases, d_off, 0)
    d_a = torch.empty(n, dtype=torch.int32, device="cuda"); d_id = torch.empty_like(d_a); d_fl = torch.empty(n, dtype=torch.uint8, device="cuda")
    d_c = torch.zeros(duk.gpu.counters_len, dtype=torch.int64, device="cuda")
    for _ in range(4):
        d_c.zero_(); duk.gpu.process_batch_device(d_bases, d_off, True, d_a, d_id, d_fl, d_c, st)
    torch.cuda.synchronize()
    ms = duk.gpu.kernel_time_ms(3)
    c = d_c.cpu().tolist()
    print(json.dumps({"args": args, "read_len": L, "reads": n, "kernel_ms": round(ms, 2), "Gbases_per_s": round(n * L / ms / 1e6, 1),
                      "readsKTrimmed_share": round(c[B.COUNTER_NAMES.index("readsKTrimmed")] / max(1, c[B.COUNTER_NAMES.index("readsIn")]), 3)}))
    del d_bases, d_off, d_a, d_id, d_fl
duk.close()
