import sys, json, os
sys.path.insert(0, "/root/repo")
import torch
from bbtools_amd import bbduk as B
if len(sys.argv) > 1: B.use_library(sys.argv[1])
duk = B.BBDuk("ktrim=r k=23 mink=11 hdist=1 ref=adapters")
st = torch.cuda.current_stream().cuda_stream
for n in (100_000, 1_000_000, 10_000_000):
    d_bases = torch.empty(n * 150, dtype=torch.uint8, device="cuda"); d_off = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    B.synth_generate_device(B.synth_params(2), 0, n // 2, d_bases, d_off, 0)
    d_a = torch.empty(n, dtype=torch.int32, device="cuda"); d_id = torch.empty_like(d_a); d_fl = torch.empty(n, dtype=torch.uint8, device="cuda")
    d_c = torch.zeros(duk.gpu.counters_len, dtype=torch.int64, device="cuda")
    for _ in range(6):
        d_c.zero_(); duk.gpu.process_batch_device(d_bases, d_off, True, d_a, d_id, d_fl, d_c, st)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        duk.gpu.process_batch_device(d_bases, d_off, True, d_a, d_id, d_fl, d_c, st)
    e1.record(); torch.cuda.synchronize()
    print(json.dumps({"lib": sys.argv[1] if len(sys.argv) > 1 else "product", "reads": n, "kernel_ms": round(duk.gpu.kernel_time_ms(3), 4), "call_ms_back_to_back": round(e0.elapsed_time(e1) / 20, 4)}))
