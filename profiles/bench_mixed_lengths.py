#!/usr/bin/env python3
"""Secondary measurement: batches of mixed read lengths, as a quality-trimmed data set has them.  2x151 pairs from the generator; a random share
of the pairs is cut to 2x100 (both mates keep their first 100 bases).  (--pair-scan: the pair scan of rounds 1-2 behind BBDUK_HOOK_PAIR_SCAN.)
One JSON line per share."""
import json, os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from bbtools_amd import bbduk as B

n = 4_000_000
L = 151
duk = B.BBDuk("ktrim=r k=23 mink=11 hdist=1 ref=adapters", hooks={B.HOOK_PAIR_SCAN: 1} if "--pair-scan" in sys.argv else None)
st = torch.cuda.current_stream().cuda_stream
src = torch.empty(n * L, dtype=torch.uint8, device="cuda"); soff = torch.empty(n + 1, dtype=torch.int64, device="cuda")
B.synth_generate_device(B.synth_params(2, read_len=L, ins_min=L // 3, ins_max=L * 7 // 3), 0, n // 2, src, soff, 0)
g = torch.Generator(device="cuda"); g.manual_seed(5)
for share in (0.0, 0.05, 0.25, 0.5, 0.75, 0.95, 1.0):
    cut = (torch.rand(n // 2, device="cuda", generator=g) < share).repeat_interleave(2)          # per pair
    lens = torch.where(cut, torch.full((n,), 100, device="cuda", dtype=torch.int64), torch.full((n,), L, device="cuda", dtype=torch.int64))
    off = torch.zeros(n + 1, dtype=torch.int64, device="cuda"); off[1:] = torch.cumsum(lens, 0)
    total = int(off[-1].item())
    rid = torch.repeat_interleave(torch.arange(n, device="cuda"), lens)
    idx = torch.arange(total, device="cuda") - off[rid] + soff[rid]
    bases = src[idx].contiguous(); del rid, idx
    d_a = torch.empty(n, dtype=torch.int32, device="cuda"); d_id = torch.empty_like(d_a); d_fl = torch.empty(n, dtype=torch.uint8, device="cuda")
    d_c = torch.zeros(duk.gpu.counters_len, dtype=torch.int64, device="cuda")
    for _ in range(4):
        d_c.zero_(); duk.gpu.process_batch_device(bases, off, True, d_a, d_id, d_fl, d_c, st)
    torch.cuda.synchronize()
    ms = duk.gpu.kernel_time_ms(3)
    print(json.dumps({"share_cut_to_2x100": share, "reads": n, "bases": total, "kernel_ms": round(ms, 3), "Gbases_per_s": round(total / ms / 1e6, 1)}))
    del bases, off, lens
duk.close()
