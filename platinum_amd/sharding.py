"""Independent-sample sharding across GPUs — host logic shared by bench.py and the gloo tests.

`frameIdx` is the only seed of the reference's sampler (samplers.metal:154-156), so giving rank g the sample indices
[g*spp_per_rank, (g+1)*spp_per_rank) makes the union over ranks exactly the sample set of one big render.  Each rank
keeps a private running mean; ONE all-reduce (sum) of the float accumulator followed by 1/N merges them (equal
shares).  Backend "nccl" is RCCL over xGMI on ROCm; the CPU tests use "gloo".
"""


def shard_samples(rank, world, spp_per_rank, base=0):
    """(first_sample, count) of `rank`."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return base + rank * spp_per_rank, spp_per_rank


def reduce_accumulator(acc, world, dist=None):
    """In-place: acc <- mean over ranks of the per-rank running means (alpha stays 1). `acc` is a torch tensor on the
    device the process group's backend reduces (cuda for nccl/RCCL, cpu for gloo)."""
    if world == 1 or dist is None:
        return acc
    dist.all_reduce(acc, op=dist.ReduceOp.SUM)
    acc.mul_(1.0 / world)
    return acc
