#!/usr/bin/env python3
"""
bench.py -- BBDuk k-mer matching path on MI355X.

Metric (BASELINE.json): Gbases/s, BBDuk ktrim=r k=23 mink=11 hdist=1 ref=adapters on synthetic 2x150bp
reads (configs[1]: 100 M reads = 15 Gbases per GPU, resident in HBM before the timed region).
A "step" = one pass of the hot path (one kernel launch) over the rank's whole read set + the counter
all-reduce.  N>1: one process per GPU (torch.distributed, backend nccl = RCCL), reads sharded (weak scaling:
every rank owns the next 100 M reads of the counter-based synthetic stream), k-mer map replicated, no
data-path collective.

With --gpus N > 1 and no WORLD_SIZE in the environment the script launches its own N ranks (torch.distributed.run, one
process per GPU, 127.0.0.1 rendezvous); under an external launcher (RANK/LOCAL_RANK/WORLD_SIZE set) it is one of the ranks.
The counter all-reduce is the library's own (bbduk_allreduce_counters_device: RCCL ncclAllReduce behind the C ABI);
torch.distributed is used for the rendezvous (handing the 128-byte RCCL id to the ranks), the barrier and the max-over-ranks clock.

Besides the contract's line for configs[1] (device-resident, ASCII boundary), rank 0 at N=1 adds:
  "end_to_end": Gbases/s THROUGH the C ABI's host-buffer operators (H2D + kernel + D2H inside the call), ASCII and packed
  "operators":  operator Gbases/s of ktrim=n, ktrim=rl, ksplit, findbestmatch (+ match lists) on the configs[1] reads; "seal": Seal's k-mer path
  "read_lengths": kernel Gbases/s of the configs[1] flags on synthetic pairs of 2x100, 2x151, 2x250 bases (3 Gbases each)
  "configs":    kernel Gbases/s of the other BASELINE configurations (C1, C3, C5 at 20 M reads; C4 = the 10 Gbase reference, map in HBM)

  python bench.py [--gpus N] [--steps K] [--warmup W] [--reads R] [--no-cpu-baseline] [--no-extras]
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


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script, one per GPU, and pass rank 0's JSON line on."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def measure_end_to_end(B, duk, reads, seed):
    """Gbases/s through bbduk_ktrim_batch / bbduk_ktrim_batch_packed -- host buffers in, H2D + kernel + D2H inside
    the call.  Two submitting threads keep both staging slots of the handle busy (copies of one call under the kernel of the other)."""
    import threading
    import numpy as np
    import torch
    n_pairs = reads // 2; n = 2 * n_pairs
    sp = B.synth_params(seed)
    d_b = torch.empty(n * READ_LEN, dtype=torch.uint8, device="cuda"); d_o = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    B.synth_generate_device(sp, 0, n_pairs, d_b, d_o, duk.gpu.params.device)
    torch.cuda.synchronize()
    hb = d_b.cpu().numpy(); hoff = d_o.cpu().numpy()
    del d_b, d_o
    codes, undef = B.pack_bases_host(hb)
    res = {"reads_per_call": n, "note": "PCIe-inclusive; never the headline `value`"}
    for name, fn in (("ascii", lambda: duk.gpu.process_batch(hb, hoff, True)), ("packed", lambda: duk.gpu.process_batch_packed(codes, undef, hoff, True))):
        fn()                                                         # warm-up: staging buffers get allocated
        t = time.perf_counter(); fn(); one = time.perf_counter() - t
        calls = 3
        def worker():
            for _ in range(calls):
                fn()
        th = [threading.Thread(target=worker) for _ in range(2)]
        t = time.perf_counter()
        for x in th: x.start()
        for x in th: x.join()
        two = (time.perf_counter() - t) / (2 * calls)
        res[name] = {"Gbases_per_s_1_thread": round(n * READ_LEN / one / 1e9, 2), "Gbases_per_s_2_threads": round(n * READ_LEN / two / 1e9, 2),
                     "ms_per_call_1_thread": round(one * 1e3, 2), "ms_per_call_2_threads": round(two * 1e3, 2),
                     "bytes_in_per_read": 158 if name == "ascii" else round((len(codes) * 4 + len(undef) * 4) / n + 8, 2), "bytes_out_per_read": 9}
    # The same operators with every buffer in page-locked memory (bbduk_pinned_malloc: what BBDukGpu.allocPinned hands the Java batcher): the
    # copies then run at the link's rate instead of through the runtime's own staging of pageable memory (round 5: 34 GB/s of a 45-50 GB/s link)
    try:
        import ctypes as C
        L = B.lib()
        def pin(arr):
            p = C.c_void_p()
            assert L.bbduk_pinned_malloc(max(int(arr.nbytes), 16), C.byref(p)) == 0
            C.memmove(p, arr.ctypes.data, arr.nbytes)
            return p
        def pin_out(nbytes):
            p = C.c_void_p()
            assert L.bbduk_pinned_malloc(nbytes, C.byref(p)) == 0
            return p
        p_b, p_off, p_codes, p_undef = pin(hb), pin(hoff), pin(codes), pin(undef)
        outs = [(pin_out(4 * n), pin_out(4 * n), pin_out(n)) for _ in range(2)]                      # one set of outputs per submitting thread
        h = duk.gpu.h
        def call(name, t):
            a, i, f = outs[t]
            rc = L.bbduk_ktrim_batch(h, p_b, p_off, n, 1, a, i, f) if name == "ascii" else L.bbduk_ktrim_batch_packed(h, p_codes, p_undef, p_off, n, 1, a, i, f)
            assert rc == 0, rc
        for name in ("ascii", "packed"):
            call(name, 0)
            t = time.perf_counter(); call(name, 0); one = time.perf_counter() - t
            calls = 4
            th = [threading.Thread(target=lambda t=t: [call(name, t) for _ in range(calls)]) for t in range(2)]
            t = time.perf_counter()
            for x in th: x.start()
            for x in th: x.join()
            two = (time.perf_counter() - t) / (2 * calls)
            res[name + "_pinned"] = {"Gbases_per_s_1_thread": round(n * READ_LEN / one / 1e9, 2), "Gbases_per_s_2_threads": round(n * READ_LEN / two / 1e9, 2),
                                     "ms_per_call_1_thread": round(one * 1e3, 2), "ms_per_call_2_threads": round(two * 1e3, 2)}
        ga = np.ctypeslib.as_array(C.cast(outs[0][0], C.POINTER(C.c_int32)), shape=(n,))
        res["pinned_equals_pageable"] = bool(np.array_equal(ga, duk.gpu.process_batch(hb, hoff, True)[0]))
        for q in (p_b, p_off, p_codes, p_undef, *[x for o in outs for x in o]): L.bbduk_pinned_free(q)
    except Exception as e:
        res["pinned_error"] = repr(e)
    return res


# Counter-derived figures of the HBM-bound configurations, ECHOED from the committed profiles (like roofline.traffic: not measured in this run).
COUNTER_ROWS = {
    "C4": {"source": "profiles/r06_c4_rdreq.txt + profiles/r06_c4_counters.txt + profiles/r06_c4_aligned_ab.txt (rocprofv3 --pmc passes of profiles/bench_c4_full.py / stage_times.py, 20 M reads per launch, the library's own geometry: m = 9)",
           "fabric_requests_per_read": 28.6, "l2_requests_per_read": 31.3, "spilled_keys_frac": 0.0114, "valu_per_read": 242.0,
           "note": "round 5 (64-byte lines, m = 10): 35.4 fabric requests per read, 6.6 % of the keys spilled; every request moves 128 bytes whatever the load asks for "
                   "(profiles/r06_rand_gran_pmc.txt); 28.6 requests x 1e8 reads in 58.6 ms = 48.7 G requests/s: what a bare gather kernel reaches for random lines over this span "
                   "(48.6 G/s, profiles/r06_rand_gran.jsonl) -- the scan runs at the fabric's request rate (profiles/r06_c4_final.txt)"},
    "C6_hdist1_genome": {"source": "profiles/r04_seed_counters.txt, profiles/r06_seed_128B.txt", "fabric_requests_per_read": 66.0,
                         "note": "the kernel runs at the fabric's request rate for lines beyond an XCD's L2 (54.8 of ~55 G/s); 128-byte lines of 16 buckets change neither the count nor the rate"},
}


def measure_configs(B, reads, c4_ref_bases, c4_reads=0, c3_reads=0):
    """Kernel Gbases/s (HIP events inside the library) of the other BASELINE configurations, device-resident batches."""
    import numpy as np
    import torch
    phix = B.read_fasta(os.path.join(ROOT, "data", "phix2.fa.gz"))[0][1]
    n_pairs = reads // 2; n = 2 * n_pairs
    res = {}
    d_bases = torch.empty(n * READ_LEN, dtype=torch.uint8, device="cuda"); d_off = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    d_a = torch.empty(n, dtype=torch.int32, device="cuda"); d_id = torch.empty_like(d_a); d_fl = torch.empty(n, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream

    def run(duk_gpu, sp, steps=3):
        B.synth_generate_device(sp, 0, n_pairs, d_bases, d_off, 0)
        d_c = torch.zeros(duk_gpu.counters_len, dtype=torch.int64, device="cuda")
        for _ in range(1 + steps):
            d_c.zero_(); duk_gpu.process_batch_device(d_bases, d_off, True, d_a, d_id, d_fl, d_c, st)
        torch.cuda.synchronize()
        return duk_gpu.kernel_time_ms(steps), d_c.cpu().numpy()

    for name, args, seed, kw in (("C1", "ktrim=r k=23 hdist=0 ref=adapters", 1, {}),
                                 ("C5", "ktrim=r k=23 hdist=1 ref=adapters", 5, {})):
        duk = B.BBDuk(args)
        ms, c = run(duk.gpu, B.synth_params(seed, **kw))
        res[name] = {"args": args, "reads": n, "keys": int(duk.stored_kmers), "kernel_ms": round(ms, 3),
                     "Gbases_per_s": round(n * READ_LEN / (ms * 1e-3) / 1e9, 1), "readsKTrimmed": int(c[2]), "readsKFiltered": int(c[4])}
        duk.close()
    # BASELINE configs[2] at its own size (100 M reads, 1 % of the pairs from phiX): its own buffers (the row ran 20 M reads until round 6)
    try:
        n3 = (c3_reads or reads) // 2 * 2
        b3 = torch.empty(n3 * READ_LEN, dtype=torch.uint8, device="cuda"); o3 = torch.empty(n3 + 1, dtype=torch.int64, device="cuda")
        a3 = torch.empty(n3, dtype=torch.int32, device="cuda"); i3 = torch.empty_like(a3); f3 = torch.empty(n3, dtype=torch.uint8, device="cuda")
        args = "k=31 hdist=1 ref=phix"
        duk = B.BBDuk(args)
        B.synth_generate_device(B.synth_params(3, contam=phix, contam_frac=0.01), 0, n3 // 2, b3, o3, 0)
        d_c = torch.zeros(duk.gpu.counters_len, dtype=torch.int64, device="cuda")
        for _ in range(4):
            d_c.zero_(); duk.gpu.process_batch_device(b3, o3, True, a3, i3, f3, d_c, st)
        torch.cuda.synchronize()
        ms = duk.gpu.kernel_time_ms(3); c = d_c.cpu().numpy()
        res["C3"] = {"args": args, "reads": n3, "keys": int(duk.stored_kmers), "kernel_ms": round(ms, 3),
                     "Gbases_per_s": round(n3 * READ_LEN / (ms * 1e-3) / 1e9, 1), "readsKTrimmed": int(c[2]), "readsKFiltered": int(c[4])}
        duk.close()
        del b3, o3, a3, i3, f3
        torch.cuda.empty_cache()
    except Exception as e:
        res["C3"] = {"error": repr(e)}
    try:                                               # the reference's own sizing example (BBDukGuide.txt:34): k=31 hdist=1 vs a 4.6 Mbase genome -- the seed layout
        sys.path.insert(0, os.path.join(ROOT, "profiles"))
        import bench_hdist_big
        r = bench_hdist_big.run(reads=reads)
        res["C6_hdist1_genome"] = {k: r[k] for k in ("workload", "keys", "table_bytes", "build_s", "reads", "kernel_ms", "Gbases_per_s", "readsKFiltered")}
        res["C6_hdist1_genome"]["counters"] = COUNTER_ROWS["C6_hdist1_genome"]
    except Exception as e:
        res["C6_hdist1_genome"] = {"error": repr(e)}
    try:                                               # the same guide line's hdist=0 figure (E. coli: 4554207 keys): the big layout's minimizer lines since round 3 (was cache-resident: 67.6 Gbases/s)
        r = bench_hdist_big.run(reads=reads, args="k=31 hdist=0")
        res["C7_hdist0_genome"] = {k: r[k] for k in ("workload", "keys", "table_bytes", "build_s", "reads", "kernel_ms", "Gbases_per_s", "readsKFiltered")}
    except Exception as e:
        res["C7_hdist0_genome"] = {"error": repr(e)}
    # round 4: the same genome at BBDuk's default k = 27, and as a ktrim=r reference (the trimming mode's large-map path);
    # all three on bbduk_bigs_kernel, the stream scan over the minimizer lines (bbduk_bigs.inc)
    try:                                               # round 5: the same hdist=1 map as C6, but UPLOADED as (key, id) pairs -- what a JVM host hands over;
        # bbduk_finalize_table collapses it to its parents and the seed layout (64 Gbases/s and 11.1 GB as plain lines until round 5).  A 1 Mbase genome here:
        # the host-side index of the 4.6 Mbase one (4.2e8 pairs) takes a minute to build, profiles/r05_uploaded_hdist1.jsonl has it (121 Gbases/s)
        r = bench_hdist_big.run(reads=reads, genome_bases=1_000_000, build="host")
        res["C6u_hdist1_uploaded"] = {k: r[k] for k in ("workload", "keys", "table_bytes", "build", "layout", "build_s", "reads", "kernel_ms", "Gbases_per_s", "readsKFiltered")}
    except Exception as e:
        res["C6u_hdist1_uploaded"] = {"error": repr(e)}
    for name, args in (("C8_k27_genome", "k=27 hdist=0"), ("C9_ktrim_genome", "ktrim=r k=31 hdist=0")):
        try:
            r = bench_hdist_big.run(reads=reads, args=args)
            res[name] = {k: r[k] for k in ("workload", "keys", "table_bytes", "build_s", "reads", "kernel_ms", "Gbases_per_s", "readsKFiltered", "readsKTrimmed")}
        except Exception as e:
            res[name] = {"error": repr(e)}
    if c4_ref_bases > 0:
        # BASELINE configs[3]: kfilter k=31 hdist=0 against a seeded uniform-random reference (default: the full 10 Gbases, ~10^10 keys,
        # a 239 GB map resident in HBM in the big layout), generated and indexed on the device chunk by chunk, 1 % of the pairs
        # drawn from the reference.  profiles/bench_c4_full.py is the same measurement with its table checks and an oracle sample.
        del d_bases, d_off, d_a, d_id, d_fl
        torch.cuda.empty_cache()
        sys.path.insert(0, os.path.join(ROOT, "profiles"))
        import bench_c4_full
        # configs[3] names 100 M reads: 15.9 GB of batch + 0.9 GB of outputs beside the 239 GB map (288 GiB of HBM).  Its first 50 000 reads are
        # replayed by the CPU oracle, which indexes the part of the reference the contaminated pairs are drawn from.
        try:                                           # (bench_c4_full falls back to the 20 M-read batch by itself if the full one finds no room)
            r = bench_c4_full.run(ref_bases=c4_ref_bases, reads=c4_reads or reads, steps=3, check_reads=50_000, fallback_reads=reads)
        except Exception as e:
            res["C4"] = {"error": repr(e)}
            return res
        res["C4_counters"] = COUNTER_ROWS["C4"]
        res["C4"] = {k: r[k] for k in ("workload", "keys", "table_bytes", "bytes_per_key", "spilled_keys", "build_s", "reads", "kernel_ms", "Gbases_per_s",
                                       "algorithmic_GBps", "frac_of_8TBps", "readsKFiltered", "sampled_reference_windows", "sampled_found_with_id", "random_keys_found",
                                       "checked_reads", "checked_reads_differ", "checked_reads_hit", "differ_device_hit_oracle_miss", "reads_requested", "full_batch_error") if k in r}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=100_000_000, help="reads per GPU (pairs = reads/2): weak scaling, the default")
    ap.add_argument("--total-reads", type=int, default=0, help="strong scaling instead: this many reads in all, sharded over the GPUs in contiguous "
                    "blocks of whole pairs (BASELINE configs[4]: --gpus 8 --total-reads 1000000000 = 125 M reads per GPU)")
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--format", choices=["ascii", "packed"], default="ascii",
                    help="boundary format of the resident batch: ascii = the reference's byte per base (the headline), "
                         "packed = 2-bit codes + undefined bits (DESIGN.md 4.7; secondary)")
    ap.add_argument("--no-extras", action="store_true", help="skip the end_to_end and configs measurements")
    ap.add_argument("--extra-reads", type=int, default=20_000_000, help="reads per call / launch of the extra measurements")
    ap.add_argument("--c4-reads", type=int, default=100_000_000, help="reads of the configs[3] launch in `configs` (BASELINE: 100 M)")
    ap.add_argument("--c3-reads", type=int, default=100_000_000, help="reads of the configs[2] launch in `configs` (BASELINE: 100 M)")
    ap.add_argument("--c4-ref-bases", type=float, default=1e10, help="reference size of BASELINE configs[3] in `configs` (0 = skip; 1e10 needs ~245 GB of HBM)")
    ap.add_argument("--dry-run-share-gpu", action="store_true", help="N>1 control-flow rehearsal on one GPU (gloo; not a measurement)")
    ap.add_argument("--insert-range", default=None, help="experiments: 'min,max' insert size of the synthetic pairs")
    ap.add_argument("--pair-scan", action="store_true", help="experiments: the pair-scan kernel of rounds 1-2 instead of bbduk_stream_kernel (BBDUK_HOOK_PAIR_SCAN)")
    ap.add_argument("--lib", default=None, help="experiments: bind to another build of the library (bbtools_amd/<name>.so)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(a.gpus)

    import numpy as np
    import torch
    import bbtools_amd
    if not os.path.exists(bbtools_amd.lib_path()):   # the prebuilt .so normally travels with the tree; build it (hipcc, gfx950) if not
        bbtools_amd.build()
    from bbtools_amd import bbduk as B
    from bbtools_amd import dist as D
    if a.lib:
        B.use_library(a.lib)

    rank, world, local = D.env_rank_world()
    distributed = world > 1
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # --dry-run-share-gpu (never passed by the driver): every rank on GPU 0 with a gloo rendezvous, so that the N>1 control
        # flow (self-launch, sharding, barrier, max-over-ranks clock) can be exercised on a 1-GPU box.  RCCL refuses two ranks on
        # one device, so in that mode -- and only there -- the counters are summed by gloo and the line says "dry_run".
        if a.dry_run_share_gpu:
            local = 0
        torch.cuda.set_device(local)
        if a.dry_run_share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))       # "nccl" is RCCL on ROCm
    dev = local if distributed else 0
    torch.cuda.set_device(dev)
    if a.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    if torch.cuda.device_count() <= dev:
        raise SystemExit("bench.py: rank %d needs GPU %d but only %d visible" % (rank, dev, torch.cuda.device_count()))

    strong = a.total_reads > 0
    if strong:                                       # configs[4]'s shape: the read set is fixed, every rank takes its contiguous block of pairs
        lo, hi = D.shard_pairs(a.total_reads // 2, rank, world)
        n_pairs = hi - lo
    else:
        n_pairs = a.reads // 2
    n = 2 * n_pairs
    duk = B.BBDuk(ARGS_C2, device=dev, hooks={B.HOOK_PAIR_SCAN: 1} if a.pair_scan else None)
    if distributed and not a.dry_run_share_gpu:       # the RCCL communicator behind the C ABI: rank 0's id goes to every rank
        box = [B.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        duk.gpu.comm_create(world, rank, box[0])
    sp_kw = {}
    if a.insert_range:                               # experiments only: "min,max" insert range (default 50,350)
        lo_, hi_ = a.insert_range.split(",")
        sp_kw = dict(ins_min=int(lo_), ins_max=int(hi_))
    sp = B.synth_params(a.seed, **sp_kw)
    if not strong:
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
        if distributed and not a.dry_run_share_gpu:   # the path's only collective: global hit/trim counters (ncclAllReduce, int64 sum)
            duk.gpu.allreduce_counters_device(total, stream)
        elif distributed:
            t_ = total.cpu(); dist.all_reduce(t_); total = t_.cuda()
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
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if a.dry_run_share_gpu else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    launch_ms = sum(e0.elapsed_time(e1) for e0, e1 in evs) / max(1, a.steps)     # span pre-pass + wave kernel + no-op fallback
    kern_ms = duk.gpu.kernel_time_ms(a.steps)                                    # the dominant kernel alone (HIP events in the library)

    ctr = totals.cpu().numpy()
    status = int(d_ctr[B.NCOUNTERS - 1].item())
    if distributed:                                  # strong scaling: the shards differ by a pair at most, but count what was really done
        t = torch.tensor([float(n)], dtype=torch.float64, device="cpu" if a.dry_run_share_gpu else "cuda")
        dist.all_reduce(t)
        total_reads = float(t.item())
    else:
        total_reads = float(n)
    total_bases = total_reads * READ_LEN
    gbases = total_bases * a.steps / dt / 1e9
    out = {
        "metric": "Gbases/s BBDuk k=23 ktrim on 2x150bp synthetic FASTQ",
        "value": round(gbases, 3), "unit": "Gbases/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong" if strong else "weak",
        "vs_baseline": None, "dtype": "int64", "data": "synthetic",
        "config": {"workload": "BBDuk ktrim=r k=23 mink=11 hdist=1 ref=adapters, %d synthetic 2x150bp reads per GPU "
                               "(configs[1]), device-resident batch%s" % (n, ", packed 2-bit boundary format" if packed else ""),
                   "args": ARGS_C2, "reads_per_gpu": n, "read_len": READ_LEN, "seed": a.seed,
                   "table_keys": int(duk.stored_kmers), "table_bytes": int(duk.gpu.table_bytes),
                   "total_reads": int(total_reads),
                   "parallelism": ("reads sharded x%d (%s), table replicated, 1 counter all-reduce/step (RCCL via bbduk_allreduce_counters_device)"
                                   % (world, "contiguous blocks of a fixed read set" if strong else "every rank its own %d reads" % n)) if world > 1
                                  else "1 GPU: no collective in the step"},
        "counters": {k: int(v) for k, v in zip(B.COUNTER_NAMES, ctr[:10])},
        **({"dry_run": "all ranks share GPU 0, counters summed by gloo: control-flow rehearsal, not a measurement"} if a.dry_run_share_gpu else {}),
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
    issue = None                                     # what paces the kernel, from the committed counter pass of this workload (profiles/r03_pmc.sh)
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_issue.json")) as fh:
            issue = json.load(fh)
    except Exception:
        pass
    out["roofline"] = {"bound": "hbm", "kernel": "bbduk_wave_kernel" if a.pair_scan else "bbduk_stream_kernel", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS,
                       "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                       "kernel_ms": round(kern_ms, 3), "launch_ms": round(launch_ms, 3), "algorithmic_bytes_per_launch": int(algo_bytes),
                       # the contract prices this path against HBM; what paces the kernel is instruction issue: the
                       # measured wave-instructions per read, echoed from the committed counter pass (null if none is committed)
                       "issue_counters": issue}

    if rank == 0 and world == 1 and not packed and not a.no_extras:
        # the same launch on the packed boundary format (2-bit codes + undefined bits, what the device ingest produces anyway;
        # DESIGN 4.7) as one extra field -- never `value`, which stays on the reference's byte-per-base Read.bases
        try:
            d_codes = torch.empty((n * READ_LEN + 15) // 16 + 4, dtype=torch.int32, device="cuda")
            d_undef = torch.empty((n * READ_LEN + 31) // 32 + 4, dtype=torch.int32, device="cuda")
            B.pack_bases_device(d_bases, d_codes, d_undef, dev)
            d_a2 = torch.empty_like(d_a); d_id2 = torch.empty_like(d_id); d_fl2 = torch.empty_like(d_fl); d_c2 = torch.zeros_like(d_ctr)
            for _ in range(1 + 3):
                d_c2.zero_(); duk.gpu.process_batch_packed_device(d_codes, d_undef, d_off, n * READ_LEN, True, d_a2, d_id2, d_fl2, d_c2, stream)
            torch.cuda.synchronize()
            pk_ms = duk.gpu.kernel_time_ms(3)
            out["packed_format"] = {"kernel_ms": round(pk_ms, 3), "Gbases_per_s": round(n * READ_LEN / (pk_ms * 1e-3) / 1e9, 1),
                                    "same_results_as_ascii": bool(torch.equal(d_a2, d_a) and torch.equal(d_id2, d_id) and torch.equal(d_fl2, d_fl)),
                                    "note": "kernel rate of the same launch on 2-bit codes + undefined bits (bbduk_ktrim_batch_packed_device); secondary"}
            del d_codes, d_undef, d_a2, d_id2, d_fl2, d_c2
        except Exception as e:
            out["packed_format"] = {"error": repr(e)}
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        try:
            # CPU baseline: the oracle (a C restatement of the reference's loops, NOT the Java reference: no JVM
            # here) on a bounded sample of the same workload, all host cores, plus a parity check of that sample.
            from oracle.oracle_ffi import Oracle
            cores = os.cpu_count() or 1
            o = Oracle(**ORACLE_C2)
            o.load_fasta(os.path.join(ROOT, "data", "adapters.fa"))
            hb, hoff = B.synth_generate_host(sp, lo, 100_000)
            # the thread count the sample runs with is the one a 100k-pair probe is fastest at: a box can show more cores than its
            # cgroup lets the process use (r03: 256 visible, 16 threads 1.7x faster than 256), and the baseline should be the port's best
            avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else cores
            cand = {avail, 128, 64, 32, 16}
            try:                                      # a CPU quota joins the probe set (r03's box: "1600000 100000" = 16 CPUs of 256 visible)
                with open("/sys/fs/cgroup/cpu.max") as fh:
                    q_, p_ = fh.read().split()[:2]
                if q_ != "max" and 1 <= int(q_) // int(p_) < avail:
                    qc = max(1, int(q_) // int(p_)); cand |= {qc, 2 * qc, 4 * qc}
            except (OSError, ValueError):
                pass
            # Each candidate gets a probe of >= 2 s (round 5, VERDICT r4: 100k-pair probes of ~0.1 s ranked 128 threads above 16 on a box whose own
            # scaling table then showed 16 threads 18 % faster): the batch grows until the probe lasts that long; the rates are the scaling table.
            hbp, hoffp = B.synth_generate_host(sp, lo, min(n_pairs, 500_000))
            max_probe_pairs = (len(hoffp) - 1) // 2
            rates = {}
            for th in sorted(cand | {1}, reverse=True):
                if th > avail:
                    continue
                pairs_t = min(max_probe_pairs, 500_000 if th > 1 else 20_000)
                done_pairs = 0; dt_ = 0.0
                while dt_ < 2.0:                      # the same batch again until the probe has lasted 2 s
                    o.reset_counters()
                    t = time.perf_counter(); o.process_batch(hbp[:2 * pairs_t * READ_LEN], hoffp[:2 * pairs_t + 1], True, nthreads=th); dt_ += time.perf_counter() - t
                    done_pairs += pairs_t
                rates[th] = 2 * done_pairs * READ_LEN / dt_ / 1e9
            cores = max(rates, key=rates.get)
            probe = 100_000 * 2 * READ_LEN / 1e9 / rates[cores]      # seconds per 100k pairs at the chosen thread count
            del hbp, hoffp
            sample_pairs = int(min(n_pairs, max(100_000, 100_000 * a.cpu_seconds / max(probe, 1e-3))))
            hb, hoff = B.synth_generate_host(sp, lo, sample_pairs)
            o.reset_counters()
            t = time.perf_counter(); oa, oi, of = o.process_batch(hb, hoff, True, nthreads=cores); cpu_dt = time.perf_counter() - t
            m = 2 * sample_pairs
            ok = bool(np.array_equal(d_a[:m].cpu().numpy(), oa) and np.array_equal(d_id[:m].cpu().numpy(), oi)
                      and np.array_equal(d_fl[:m].cpu().numpy(), of))
            out["cpu_baseline"] = {"value": round(m * READ_LEN / cpu_dt / 1e9, 4), "unit": "Gbases/s", "cores": cores,
                                   "kind": "port", "sample": "first %d reads of the same synthetic workload, %.1f s, "
                                   "oracle/bbduk_oracle.c with %d pthreads (the fastest of %s in probes of >= 2 s each: threads_scaling; the cgroup's CPU quota and its multiples are in the set); os.cpu_count() = %d "
                                   "(match only, no I/O)" % (m, cpu_dt, cores, sorted(rates), os.cpu_count() or 1)}
            try:
                with open("/sys/fs/cgroup/cpu.max") as fh:
                    out["cpu_baseline"]["cgroup_cpu_max"] = fh.read().strip()
            except OSError:
                pass
            out["parity_sample_ok"] = ok
            if max(rates.values()) > out["cpu_baseline"]["value"]:      # (the baseline is the port's best measurement)
                out["cpu_baseline"]["value"] = round(max(rates.values()), 4); out["cpu_baseline"]["sample"] += "; value = the best >= 2 s probe (the long sample ran slower)"
            scal = {str(th): round(r, 4) for th, r in sorted(rates.items())}      # the probes above: >= 2 s each
            out["cpu_baseline"]["threads_scaling"] = scal
        except Exception as e:                        # the bench line must not depend on the checker building / running
            out["cpu_baseline"] = {"value": None, "unit": "Gbases/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
    if rank == 0 and world == 1 and not a.no_extras:
        del d_bases, d_off, d_a, d_id, d_fl
        if packed:
            del d_codes, d_undef
        torch.cuda.empty_cache()
        try:
            out["end_to_end"] = measure_end_to_end(B, duk, a.extra_reads, a.seed)
        except Exception as e:
            out["end_to_end"] = {"error": repr(e)}
        duk.close(); duk = None
        try:                                           # the other operators of the boundary on the configs[1] reads, and Seal's k-mer path
            sys.path.insert(0, os.path.join(ROOT, "profiles"))
            import bench_modes, bench_seal
            out["operators"] = {r["mode"]: {k: r[k] for k in ("args", "reads", "operator_ms", "Gbases_per_s")} for r in bench_modes.run(a.extra_reads)}
            # round 5: the same operator families against the guide's genome-size example (a random 4.6 Mbase genome), i.e. over a
            # big-layout map scanned by bbduk_bigs_every_kernel -- cache-resident at 60-75 Gbases/s until round 5
            out["operators_genome"] = {r["mode"]: {k: r[k] for k in ("args", "keys", "reads", "operator_ms", "Gbases_per_s")} for r in bench_modes.run_genome(a.extra_reads)}
            out["seal"] = bench_seal.run(pairs=1_000_000, check=2000)      # (the first 2 000 pairs replayed by oracle/seal_oracle.c: sample_equals_restatement)
            # the same number of keys in FOUR scaffolds: the tool's typical binning run (a handful of reference genomes).  Round 5: the per-scaffold
            # counters go through a cache in the wave's lanes -- one atomic per unit and column serialised the grid here (17 Gbases/s)
            out["seal_4_scaffolds"] = bench_seal.run(pairs=1_000_000, scaffolds=4, scaffold_len=1_600_000, check=2000)
        except Exception as e:
            out["operators"] = {"error": repr(e)}
        try:                                           # the configs[1] flags on pairs of other lengths (2x151: the tail pass of the main kernel)
            import bench_readlens
            out["read_lengths"] = {"2x%d" % r["read_len"]: {k: r[k] for k in ("reads", "kernel_ms", "Gbases_per_s")} for r in bench_readlens.run(lens=(100, 151, 250))}
        except Exception as e:
            out["read_lengths"] = {"error": repr(e)}
        try:
            out["configs"] = measure_configs(B, a.extra_reads, a.c4_ref_bases, a.c4_reads, a.c3_reads)
        except Exception as e:
            out["configs"] = {"error": repr(e)}
    if duk is not None:
        duk.close()
    if rank == 0:
        print(json.dumps(out))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
