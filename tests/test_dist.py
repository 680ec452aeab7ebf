"""world_size-2 gloo test of the N>1 path on CPU: shards are whole pairs, disjoint and covering; the
all-reduced counter vector of the shards equals the single-process counters (checked with the oracle as
the per-shard worker, since kernels cannot run here)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bbtools_amd import bbduk as B
from bbtools_amd import dist as D
from tests import util


def test_shards_cover_and_keep_pairs():
    for total in (0, 1, 7, 1000, 12345):
        for world in (1, 2, 3, 8):
            spans = [D.shard_pairs(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
    assert D.weak_shard(100, 3) == (300, 400)


def _worker(rank, world, port, n_pairs, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.oracle_ffi import Oracle
    args, okw, ref = util.CONFIGS["c2"]
    o = Oracle(**okw); o.load_fasta(ref)
    lo, hi = D.shard_pairs(n_pairs, rank, world)
    sp = B.synth_params(2)
    b, off = B.synth_generate_host(sp, lo, hi - lo)
    o.process_batch(b, off, True)
    c = torch.from_numpy(o.counters().copy())
    D.all_reduce_counters(c)
    if rank == 0:
        q.put(c.numpy().tolist())
    dist.barrier()
    dist.destroy_process_group()


def test_counter_allreduce_world2():
    n_pairs = 3001
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_pairs, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    from oracle.oracle_ffi import Oracle
    args, okw, ref = util.CONFIGS["c2"]
    o = Oracle(**okw); o.load_fasta(ref)
    b, off = B.synth_generate_host(B.synth_params(2), 0, n_pairs)
    o.process_batch(b, off, True, nthreads=4)
    assert got == o.counters().tolist()
