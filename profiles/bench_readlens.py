#!/usr/bin/env python3
"""Secondary measurement: one set of flags (default: configs[1]'s) on synthetic pairs of other read lengths than 2x150 (2x100, 2x151, 2x250,
2x300 ...), device-resident, kernel time from the library's HIP events; ~3 Gbases per launch each.  One JSON line per read length."""
import argparse, json, os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))


def run(args="ktrim=r k=23 mink=11 hdist=1 ref=adapters", lens=(100, 150, 151, 250, 300), bases=3e9, hooks=None):
    """-> one dict per read length (bench.py's `read_lengths` block calls this too)"""
    import torch
    from bbtools_amd import bbduk as B
    duk = B.BBDuk(args, hooks=hooks)
    st = torch.cuda.current_stream().cuda_stream
    out = []
    for L in lens:
        n = (int(bases) // L) & ~1
        d_bases = torch.empty(n * L, dtype=torch.uint8, device="cuda"); d_off = torch.empty(n + 1, dtype=torch.int64, device="cuda")
        B.synth_generate_device(B.synth_params(2, read_len=L, ins_min=L // 3, ins_max=L * 7 // 3), 0, n // 2, d_bases, d_off, 0)
        d_a = torch.empty(n, dtype=torch.int32, device="cuda"); d_id = torch.empty_like(d_a); d_fl = torch.empty(n, dtype=torch.uint8, device="cuda")
        d_c = torch.zeros(duk.gpu.counters_len, dtype=torch.int64, device="cuda")
        for _ in range(4):
            d_c.zero_(); duk.gpu.process_batch_device(d_bases, d_off, True, d_a, d_id, d_fl, d_c, st)
        torch.cuda.synchronize()
        ms = duk.gpu.kernel_time_ms(3)
        c = d_c.cpu().tolist()
        out.append({"args": args, "read_len": L, "reads": n, "kernel_ms": round(ms, 2), "Gbases_per_s": round(n * L / ms / 1e6, 1),
                    "readsKTrimmed_share": round(c[B.COUNTER_NAMES.index("readsKTrimmed")] / max(1, c[B.COUNTER_NAMES.index("readsIn")]), 3)})
        del d_bases, d_off, d_a, d_id, d_fl
    duk.close()
    return out


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--lib", default=None); ap.add_argument("--bases", type=float, default=3e9); ap.add_argument("--pair-scan", action="store_true")
    ap.add_argument("--args", default="ktrim=r k=23 mink=11 hdist=1 ref=adapters")
    ap.add_argument("lens", nargs="*", type=int, default=[100, 150, 151, 250, 300])
    a = ap.parse_args()
    if a.lib:
        from bbtools_amd import bbduk as B
        B.use_library(a.lib)
    from bbtools_amd import bbduk as B_
    for r in run(a.args, a.lens, a.bases, {B_.HOOK_PAIR_SCAN: 1} if a.pair_scan else None):
        print(json.dumps(r))


if __name__ == "__main__":
    main()
