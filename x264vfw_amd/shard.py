"""Multi-GPU host logic: streams (independent closed-GOP sequences) shard one set per GPU, no collective in
the data path.  The only cross-rank traffic is the benchmark bookkeeping below (a barrier
and a MAX over ranks of the timed interval), which works on nccl(RCCL) and on gloo alike."""
import os


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def stream_ids(rank, world, streams_per_gpu):
    """global ids of the streams this rank encodes (weak scaling: every rank gets streams_per_gpu)"""
    assert 0 <= rank < world and streams_per_gpu >= 1
    return list(range(rank * streams_per_gpu, (rank + 1) * streams_per_gpu))


def split_groups(n_streams, groups):
    """sizes of the lock-step groups a rank's streams are split into (each group = one HIP stream)"""
    groups = max(1, min(groups, n_streams))
    return [n_streams // groups + (1 if i < n_streams % groups else 0) for i in range(groups)]


def stream_seed(base, global_stream_id):
    return base + global_stream_id


def max_over_ranks(value, dist, device="cpu"):
    """MAX-reduce a python float over all ranks (identity when torch.distributed is not initialised)"""
    if dist is None or not dist.is_initialized():
        return float(value)
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def aggregate_fps(streams_per_gpu, steps, world, max_seconds):
    """whole-job frames/s: every rank advanced streams_per_gpu streams by `steps` frames"""
    return streams_per_gpu * steps * world / max_seconds
