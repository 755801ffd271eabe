"""Stream-parallel sharding of independent camera streams over the GPUs of one node.

Every frame of a stream depends only on that stream's previous frame, so streams are dealt to ranks
(one process per GPU) and never exchange activations.  The only collective is the global-shutter
barrier: an all-reduce(MIN) of each rank's finished frame-group number, so that the results of frame
group k are released on every GPU together.  The message is 4 bytes; it is latency-bound and is the
reason the backend (RCCL over xGMI for "nccl", gloo on CPU) hardly matters.  The reference is a
single-process, single-GPU program (feature/src/PPGExtractor.cpp:41) and has no counterpart.
"""
import torch


def shard_streams(n_streams: int, world: int, rank: int):
    """Stream ids owned by `rank`: round-robin, s -> GPU (s mod world)."""
    if not (0 <= rank < world):
        raise ValueError("rank outside world")
    return list(range(rank, n_streams, world))


class ShutterBarrier:
    """all-reduce(MIN) of the per-rank frame-group counter; returns the group every rank has finished."""

    def __init__(self, dist, device):
        self.dist = dist
        self.seq = torch.zeros(1, dtype=torch.int32, device=device)

    def release(self, group: int) -> int:
        if self.dist is None or not self.dist.is_initialized() or self.dist.get_world_size() == 1:
            return group
        self.seq.fill_(group)
        self.dist.all_reduce(self.seq, op=self.dist.ReduceOp.MIN)
        return int(self.seq.item())
