"""
Multi-GPU form of the path: reads shard embarrassingly, the k-mer map is replicated per GPU,
and the only exchange is ONE all-reduce (sum, int64) of the counter vector at the end of the run -- the
device-side equivalent of BBDukProcessorS.add (bbduk/BBDukProcessorS.java:300-342) merging per-thread
counters.  One process per GPU; torch.distributed is plumbing only (backend "nccl" is RCCL on ROCm, "gloo"
in the CPU tests).
"""
import os


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def shard_pairs(total_pairs: int, rank: int, world: int):
    """Contiguous block of whole pairs for this rank (mates are never split; order inside a shard is input
    order, so `ordered` output can be restored by concatenating shards by rank)."""
    lo = total_pairs * rank // world
    hi = total_pairs * (rank + 1) // world
    return lo, hi


def weak_shard(pairs_per_gpu: int, rank: int):
    """Weak scaling: every rank owns the next `pairs_per_gpu` pairs of the (counter-based) synthetic stream."""
    return pairs_per_gpu * rank, pairs_per_gpu * (rank + 1)


def all_reduce_counters(counters):
    """In-place sum of the int64 counter vector over all ranks; no-op when not distributed."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(counters, op=dist.ReduceOp.SUM)
    return counters
