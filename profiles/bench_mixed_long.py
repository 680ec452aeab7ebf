#!/usr/bin/env python3
"""One long unit in a batch of short pairs: what does the whole batch cost?  (A unit beyond a wave's planes sends the batch to the tiled kernel, one
beyond the tiles to the chunked long-read kernels.)  ktrim=r on the configs[1] reads, 8 M reads + one pair whose first mate is `long` bases."""
import json, os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from bbtools_amd import bbduk as B
n = 8_000_000
L = B.lib()
sp = B.synth_params(2)
d_b0 = torch.empty(n * 150, dtype=torch.uint8, device="cuda"); d_o0 = torch.empty(n + 1, dtype=torch.int64, device="cuda")
B.synth_generate_device(sp, 0, n // 2, d_b0, d_o0, 0)
duk = B.BBDuk("ktrim=r k=23 mink=11 hdist=1 ref=adapters")
for long in (0, 2400, 3000, 30000, 60000):
    if long:
        extra = torch.randint(0, 4, (long + 150,), device="cuda", dtype=torch.int64)
        lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device="cuda")
        d_bases = torch.cat([d_b0, lut[extra]]); d_off = torch.cat([d_o0, torch.tensor([n * 150 + long, n * 150 + long + 150], dtype=torch.int64, device="cuda")])
        m = n + 2
    else:
        d_bases, d_off, m = d_b0, d_o0, n
    tot = int(d_off[-1].item())
    d_a = torch.empty(m, dtype=torch.int32, device="cuda"); d_i = torch.empty_like(d_a); d_f = torch.empty(m, dtype=torch.uint8, device="cuda")
    d_c = torch.zeros(duk.gpu.counters_len, dtype=torch.int64, device="cuda")
    ts = []
    for _ in range(4):
        d_c.zero_(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
        rc = L.bbduk_ktrim_batch_device(duk.gpu.h, d_bases.data_ptr(), d_off.data_ptr(), m, tot, 1, d_a.data_ptr(), d_i.data_ptr(), d_f.data_ptr(), d_c.data_ptr(), None)
        e1.record(); torch.cuda.synchronize(); assert rc == 0, rc
        ts.append(e0.elapsed_time(e1))
    ms = min(ts[1:])
    print(json.dumps({"long_unit_bases": long, "reads": m, "operator_ms": round(ms, 3), "Gbases_per_s": round(tot / ms / 1e6, 1), "readsKTrimmed": int(d_c[2].item())}), flush=True)
