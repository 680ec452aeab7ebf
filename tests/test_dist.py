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


@pytest.mark.gpu
def test_two_shards_through_the_hip_path_and_the_abi_allreduce():
    """SURVEY 8e on the product path: contiguous blocks of whole pairs go through SEPARATE handles (two on device 0, plus one per
    further visible device), each accumulates its own counters, and one bbduk_allreduce_counters_local (same-device handles
    summed on the device, device leaders through RCCL) leaves every handle with the counters of a single-shot run -- which in
    turn equal the oracle's."""
    import torch
    from oracle.oracle_ffi import Oracle
    args, okw, ref = util.CONFIGS["c2"]
    n_pairs = 20000
    b, off = B.synth_generate_host(B.synth_params(2), 0, n_pairs)
    o = Oracle(**okw); o.load_fasta(ref)
    oa, oi, of = o.process_batch(b, off, True, nthreads=8)
    single = B.BBDuk(args)
    sa, si, sf = single.gpu.process_batch(b, off, True)
    want = single.gpu.counters()
    assert np.array_equal(want, o.counters()) and np.array_equal(sa, oa)
    devices = [0, 0] + list(range(1, torch.cuda.device_count()))
    duks = [B.BBDuk(args, device=d) for d in devices]
    world = len(duks)
    for r, d in enumerate(duks):
        lo, hi = D.shard_pairs(n_pairs, r, world)
        sb = b[off[2 * lo]:off[2 * hi]]; so = off[2 * lo:2 * hi + 1] - off[2 * lo]
        ga, gi, gf = d.gpu.process_batch(sb, so, True)
        assert np.array_equal(ga, oa[2 * lo:2 * hi]) and np.array_equal(gi, oi[2 * lo:2 * hi]) and np.array_equal(gf, of[2 * lo:2 * hi])
        assert not np.array_equal(d.gpu.counters(), want)
    gpus = [d.gpu for d in duks]
    B.comm_create_local(gpus)
    assert gpus[0].comm_size == len(set(devices))
    if torch.cuda.device_count() >= 2:                 # a multi-GPU lease: the device leaders went through RCCL, not just the same-device sum
        assert gpus[0].comm_size >= 2
    B.allreduce_counters_local(gpus)
    for g in gpus:
        assert np.array_equal(g.counters(), want)
    with pytest.raises(B.BBDukError):
        B.comm_create_local(gpus)                      # already in a group
    for d in duks:
        d.close()
    single.close()


@pytest.mark.gpu
def test_multi_process_form_of_the_communicator_world1():
    """The one-process-per-GPU entry points (bbduk_comm_unique_id / bbduk_comm_create / bbduk_allreduce_counters[_device]) with
    the only world a 1-GPU box can form: the all-reduce of one rank leaves the vector as it is."""
    import torch
    args, okw, ref = util.CONFIGS["c2"]
    d = B.BBDuk(args)
    b, off = B.synth_generate_host(B.synth_params(2), 0, 3000)
    d.gpu.process_batch(b, off, True)
    before = d.gpu.counters()
    with pytest.raises(B.BBDukError):
        d.gpu.allreduce_counters()                     # no communicator yet
    uid = B.comm_unique_id()
    assert len(uid) == 128
    d.gpu.comm_create(1, 0, uid)
    assert d.gpu.comm_size == 1
    d.gpu.allreduce_counters()
    assert np.array_equal(d.gpu.counters(), before) and before[0] == 6000
    t = torch.from_numpy(before.copy()).cuda()
    d.gpu.allreduce_counters_device(t, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(t.cpu().numpy(), before)
    with pytest.raises(B.BBDukError):
        d.gpu.comm_create(1, 0, uid)                   # one communicator per handle
    d.close()


def _rccl_worker(rank, world, uid, n_pairs, q):
    import torch
    torch.cuda.set_device(rank)
    args, okw, ref = util.CONFIGS["c2"]
    d = B.BBDuk(args, device=rank)
    d.gpu.comm_create(world, rank, uid)
    lo, hi = D.shard_pairs(n_pairs, rank, world)
    b, off = B.synth_generate_host(B.synth_params(2), lo, hi - lo)
    d.gpu.process_batch(b, off, True)
    d.gpu.allreduce_counters()                         # ncclAllReduce(int64, sum) over xGMI behind the C ABI
    q.put((rank, d.gpu.comm_size, d.gpu.counters().tolist()))
    d.close()


@pytest.mark.gpu
def test_two_rank_communicator_over_rccl():
    """One process per GPU, the form bench.py --gpus N runs (BASELINE configs[4]): two ranks on two DEVICES form the communicator
    from rank 0's id, each processes its contiguous block of pairs, and the library's all-reduce leaves both with the counters of
    the whole read set (= the oracle's).  Skipped on a 1-GPU box: the first multi-GPU lease exercises RCCL without a code change."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    from oracle.oracle_ffi import Oracle
    args, okw, ref = util.CONFIGS["c2"]
    n_pairs = 20001
    uid = B.comm_unique_id()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rccl_worker, args=(r, 2, uid, n_pairs, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=600) for _ in range(2)]
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    o = Oracle(**okw); o.load_fasta(ref)
    b, off = B.synth_generate_host(B.synth_params(2), 0, n_pairs)
    o.process_batch(b, off, True, nthreads=8)
    for rank, size, counters in got:
        assert size == 2 and counters == o.counters().tolist()
