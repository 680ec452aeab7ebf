#!/usr/bin/env python3
"""Secondary measurement: configurations that run in the GENERAL kernel instantiations (and a few plain ones for scale) on the configs[1] reads,
device-resident, kernel time from the library's HIP events.  One JSON line per configuration."""
import sys, json, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from bbtools_amd import bbduk as B
n = 20_000_000
d_bases = torch.empty(n * 150, dtype=torch.uint8, device="cuda"); d_off = torch.empty(n + 1, dtype=torch.int64, device="cuda")
B.synth_generate_device(B.synth_params(2), 0, n // 2, d_bases, d_off, 0)
d_a = torch.empty(n, dtype=torch.int32, device="cuda"); d_id = torch.empty_like(d_a); d_fl = torch.empty(n, dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
ARGS = [a for a in os.environ.get("BENCH_GENERAL_ARGS", "").split(";") if a]
for args in ARGS or ("ktrim=r k=23 mink=11 hdist=1 ref=adapters", "ktrim=r k=23 mink=11 hdist=1 restrictright=100 ref=adapters", "ktrim=r k=23 mink=11 hdist=1 qskip=2 ref=adapters",
             "k=31 hdist=1 mkf=0.2 ref=phix", "k=31 hdist=1 mcf=0.5 ref=phix", "k=31 hdist=1 speed=4 ref=phix", "ktrim=r k=23 mink=11 hdist=1 rcomp=f ref=adapters",
             "ktrim=r k=21 mink=11 hdist=1 qhdist=1 ref=adapters", "ktrim=l k=23 mink=11 hdist=1 ref=adapters", "k=31 hdist=1 mbk=2 ref=phix"):
    duk = B.BBDuk(args, hooks={B.HOOK_FORCE_TILE: 1} if os.environ.get("BENCH_GENERAL_TILE") else None)
    d_c = torch.zeros(duk.gpu.counters_len, dtype=torch.int64, device="cuda")
    for _ in range(4):
        d_c.zero_(); duk.gpu.process_batch_device(d_bases, d_off, True, d_a, d_id, d_fl, d_c, st)
    torch.cuda.synchronize()
    ms = duk.gpu.kernel_time_ms(3)
    print(json.dumps({"args": args, "kernel_ms": round(ms, 2), "Gbases_per_s": round(n * 150 / ms / 1e6, 1)}))
    duk.close()
