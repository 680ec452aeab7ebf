#!/usr/bin/env python3
"""
bench.py -- BBDuk k-mer matching path on MI355X.

Metric (BASELINE.json): Gbases/s, BBDuk ktrim=r k=23 mink=11 hdist=1 ref=adapters on synthetic 2x150bp
reads (configs[1]: 100 M reads = 15 Gbases per GPU, resident in HBM before the timed region).
A "step" = one pass of the hot path (one kernel launch) over the rank's whole read set + the counter
all-reduce.  N>1: one process per GPU (torch.distributed, backend nccl = RCCL), reads sharded (weak scaling:
every rank owns the next 100 M reads of the counter-based synthetic stream), k-mer map replicated, no
data-path collective.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--reads R] [--no-cpu-baseline]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ARGS_C2 = "ktrim=r k=23 mink=11 hdist=1 ref=adapters"
ORACLE_C2 = dict(k=23, mink=11, hdist=1, ktrimRight=1)
READ_LEN = 150
ALGO_BYTES_PER_READ = READ_LEN + 8 + 8          # bases + int64 offset + (trimmed, id0)   
HBM_PEAK_GBS = 8000.0                           # MI355X_MICROARCH.md: 8 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=100_000_000, help="reads per GPU (pairs = reads/2)")
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--format", choices=["ascii", "packed"], default="ascii",
                    help="boundary format of the resident batch: ascii = the reference's byte per base (the headline), "
                         "packed = 2-bit codes + undefined bits (DESIGN.md 4.7; secondary)")
    a = ap.parse_args()

    import numpy as np
    import torch
    import bbtools_amd
    if not os.path.exists(bbtools_amd.lib_path()):   # the prebuilt .so normally travels with the tree; build it (hipcc, gfx950) if not
        bbtools_amd.build()
    from bbtools_amd import bbduk as B
    from bbtools_amd import dist as D

    rank, world, local = D.env_rank_world()
    distributed = world > 1
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # Dry-run hooks for a 1-GPU box (never set by the driver): every rank on GPU 0 and a gloo rendezvous, so that the
        # N>1 control flow (sharding, barrier, max-over-ranks clock, counter all-reduce) can be exercised without N GPUs.
        if os.environ.get("BBDUK_BENCH_SHARE_GPU"):
            local = 0
        torch.cuda.set_device(local)
        backend = os.environ.get("BBDUK_BENCH_BACKEND", "nccl")   # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    dev = local if distributed else 0
    torch.cuda.set_device(dev)
    assert a.gpus == world, "--gpus must equal WORLD_SIZE (launch with torch.distributed.run for N>1)"

    n_pairs = a.reads // 2
    n = 2 * n_pairs
    duk = B.BBDuk(ARGS_C2, device=dev)
    sp_kw = {}
    if os.environ.get("BBDUK_BENCH_INS"):            # experiments only: "min,max" insert range (default 50,350)
        lo_, hi_ = os.environ["BBDUK_BENCH_INS"].split(",")
        sp_kw = dict(ins_min=int(lo_), ins_max=int(hi_))
    sp = B.synth_params(a.seed, **sp_kw)
    lo, hi = D.weak_shard(n_pairs, rank)

    d_bases = torch.empty(n * READ_LEN, dtype=torch.uint8, device="cuda")
    d_off = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    B.synth_generate_device(sp, lo, n_pairs, d_bases, d_off, dev)
    d_a = torch.empty(n, dtype=torch.int32, device="cuda")
    d_id = torch.empty(n, dtype=torch.int32, device="cuda")
    d_fl = torch.empty(n, dtype=torch.uint8, device="cuda")
    d_ctr = torch.zeros(duk.gpu.counters_len, dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    packed = a.format == "packed"
    if packed:                                       # converted once, outside the timed region, like the generator
        d_codes = torch.empty((n * READ_LEN + 15) // 16 + 4, dtype=torch.int32, device="cuda")
        d_undef = torch.empty((n * READ_LEN + 31) // 32 + 4, dtype=torch.int32, device="cuda")
        B.pack_bases_device(d_bases, d_codes, d_undef, dev)
        torch.cuda.synchronize()

    def step(ev0=None, ev1=None):
        d_ctr.zero_()
        if ev0 is not None:
            ev0.record()
        if packed:
            duk.gpu.process_batch_packed_device(d_codes, d_undef, d_off, n * READ_LEN, True, d_a, d_id, d_fl, d_ctr, stream)
        else:
            duk.gpu.process_batch_device(d_bases, d_off, True, d_a, d_id, d_fl, d_ctr, stream)
        if ev1 is not None:
            ev1.record()
        total = d_ctr.clone()
        D.all_reduce_counters(total)             # the path's only collective: global hit/trim counters
        return total

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    barrier()
    t0 = time.perf_counter()
    for i in range(a.steps):
        totals = step(*evs[i])
    barrier()
    dt = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    launch_ms = sum(e0.elapsed_time(e1) for e0, e1 in evs) / max(1, a.steps)     # span pre-pass + wave kernel + no-op fallback
    kern_ms = duk.gpu.kernel_time_ms(a.steps)                                    # the dominant kernel alone (HIP events in the library)

    ctr = totals.cpu().numpy()
    status = int(d_ctr[B.NCOUNTERS - 1].item())
    total_bases = float(n) * READ_LEN * world
    gbases = total_bases * a.steps / dt / 1e9
    out = {
        "metric": "Gbases/s BBDuk k=23 ktrim on 2x150bp synthetic FASTQ",
        "value": round(gbases, 3), "unit": "Gbases/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "int64", "data": "synthetic",
        "config": {"workload": "BBDuk ktrim=r k=23 mink=11 hdist=1 ref=adapters, %d synthetic 2x150bp reads per GPU "
                               "(configs[1]), device-resident batch%s" % (n, ", packed 2-bit boundary format" if packed else ""),
                   "args": ARGS_C2, "reads_per_gpu": n, "read_len": READ_LEN, "seed": a.seed,
                   "table_keys": int(duk.stored_kmers), "table_bytes": int(duk.gpu.table_bytes),
                   "parallelism": "reads sharded x%d, table replicated, 1 counter all-reduce/step" % world},
        "counters": {k: int(v) for k, v in zip(B.COUNTER_NAMES, ctr[:10])},
        "device_status": status,
    }
    algo_bytes = (ALGO_BYTES_PER_READ if not packed else READ_LEN * 0.375 + 8 + 8) * n     # 0.375 B/base with 2-bit + mask
    ach = algo_bytes / (kern_ms * 1e-3) / 1e9
    # HBM-side traffic per launch: PMC counters cannot be read from inside this process, so the figure is the one
    # measured by profiles/run_profile.sh (separate rocprofv3 --pmc passes over this same command) and committed in
    # profiles/pmc_traffic.json; it is reported only for the workload size it was measured on, else null.
    traffic, traffic_src = None, None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            pt = json.load(fh)
        if int(pt["reads_per_launch"]) == n and not packed:
            traffic, traffic_src = int(pt["traffic_bytes"]), pt["source"]
    except Exception:
        pass
    out["roofline"] = {"bound": "hbm", "kernel": "bbduk_wave_kernel", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS,
                       "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                       "kernel_ms": round(kern_ms, 3), "launch_ms": round(launch_ms, 3), "algorithmic_bytes_per_launch": int(algo_bytes),
                       # the contract prices this path against HBM; what actually paces the kernel is instruction issue
                       # (131 VALU + 70 SALU wave-instructions per read at ~4.4 / ~2.6 SIMD cycles each)
                       "limiter": "instruction issue, not HBM: 131 VALU + 70 SALU per read (profiles/r01_final_summary.txt)"}

    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        try:
            # CPU baseline: the oracle (a C restatement of the reference's loops, NOT the Java reference: no JVM
            # here) on a bounded sample of the same workload, all host cores, plus a parity check of that sample.
            from oracle.oracle_ffi import Oracle
            cores = os.cpu_count() or 1
            o = Oracle(**ORACLE_C2)
            o.load_fasta(os.path.join(ROOT, "data", "adapters.fa"))
            hb, hoff = B.synth_generate_host(sp, lo, 100_000)
            t = time.perf_counter(); o.process_batch(hb, hoff, True, nthreads=cores); probe = time.perf_counter() - t
            sample_pairs = int(min(n_pairs, max(100_000, 100_000 * a.cpu_seconds / max(probe, 1e-3))))
            hb, hoff = B.synth_generate_host(sp, lo, sample_pairs)
            o.reset_counters()
            t = time.perf_counter(); oa, oi, of = o.process_batch(hb, hoff, True, nthreads=cores); cpu_dt = time.perf_counter() - t
            m = 2 * sample_pairs
            ok = bool(np.array_equal(d_a[:m].cpu().numpy(), oa) and np.array_equal(d_id[:m].cpu().numpy(), oi)
                      and np.array_equal(d_fl[:m].cpu().numpy(), of))
            out["cpu_baseline"] = {"value": round(m * READ_LEN / cpu_dt / 1e9, 4), "unit": "Gbases/s", "cores": cores,
                                   "kind": "port", "sample": "first %d reads of the same synthetic workload, %.1f s, "
                                   "oracle/bbduk_oracle.c with %d pthreads (match only, no I/O)" % (m, cpu_dt, cores)}
            out["parity_sample_ok"] = ok
        except Exception as e:                        # the bench line must not depend on the checker building / running
            out["cpu_baseline"] = {"value": None, "unit": "Gbases/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
    duk.close()
    if rank == 0:
        print(json.dumps(out))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
