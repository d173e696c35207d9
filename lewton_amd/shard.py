"""Sharding of independent streams across the GPUs of a node: whole streams, `stream_id mod G`.

A stream's decode touches only its own PreviousWindowRight and the immutable headers (audio.rs:919), so streams never
exchange data: one process per GPU, one `lw_decoder` per process, no collective in the data path.  The only
communication is the benchmark's barrier and the MAX-reduction of the elapsed time.
"""


def shard_streams(n_streams, world_size, rank):
    """Stream ids owned by `rank`: stream_id mod world_size == rank."""
    if not (0 <= rank < world_size):
        raise ValueError("rank %d outside world of %d" % (rank, world_size))
    return list(range(rank, n_streams, world_size))


def owner_of(stream_id, world_size):
    return stream_id % world_size


def max_elapsed(elapsed_seconds, dist=None, device=None):
    """Whole-job time = the slowest rank's time (dist: an initialised torch.distributed module, or None for 1 process)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(elapsed_seconds)
    import torch
    t = torch.tensor([elapsed_seconds], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def job_throughput(units_per_rank, elapsed_seconds, dist=None, device=None):
    """Weak scaling: every rank processed `units_per_rank` units; value = all units / slowest rank's time."""
    world = 1 if dist is None or not dist.is_initialized() else dist.get_world_size()
    return units_per_rank * world / max_elapsed(elapsed_seconds, dist, device)
