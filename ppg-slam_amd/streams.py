"""Stream-parallel sharding of independent camera streams over the GPUs of one node.

Every frame of a stream depends only on that stream's previous frame, so streams are dealt to ranks
(one process per GPU) and never exchange activations.  The only collective is the global-shutter
barrier: an all-reduce(MIN) of each rank's finished frame-group number, so that the results of frame
group k are released on every GPU together.  The message is 4 bytes; it is latency-bound and is the
reason the backend (RCCL over xGMI for "nccl", gloo on CPU) hardly matters.  The reference is a
single-process, single-GPU program (feature/src/PPGExtractor.cpp:41) and has no counterpart.

A stereo rig whose two cameras sit on two GPUs (ranks 2p and 2p+1) may also exchange the frame's
descriptors point to point (StereoExchange: one send and one receive of at most max_num x 256 floats
= 1 MiB, a few microseconds over one xGMI link) so that each side can match left against right with
the library's matcher.
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


class StereoExchange:
    """Point-to-point swap of a frame's descriptors between the two ranks of a stereo pair (2p, 2p+1).

    Buffers are fixed-size ([max_num, dim] plus a count), so the exchange is one send and one receive of
    constant shape per frame and nothing is allocated in steady state."""

    def __init__(self, dist, device, max_num, dim=256):
        self.dist = dist
        self.rank = dist.get_rank()
        world = dist.get_world_size()
        self.partner = self.rank ^ 1
        if self.partner >= world:
            raise ValueError("rank %d has no stereo partner in a world of %d" % (self.rank, world))
        self.max_num, self.dim = max_num, dim
        # row max_num carries the count in its first element
        self.out = torch.zeros(max_num + 1, dim, dtype=torch.float32, device=device)
        self.inp = torch.zeros_like(self.out)

    def swap(self, desc, n):
        """desc: [>= n, dim] float32 tensor on the exchange's device.  Returns (partner's descriptors [m, dim], m)."""
        if n > self.max_num:
            raise ValueError("more descriptors than max_num")
        self.out[:n].copy_(desc[:n])
        self.out[self.max_num, 0] = float(n)
        d = self.dist
        ops = [d.P2POp(d.isend, self.out, self.partner), d.P2POp(d.irecv, self.inp, self.partner)]
        if self.rank & 1:
            ops.reverse()  # the odd rank posts its receive first: send/recv pairs match up without a deadlock
        for r in d.batch_isend_irecv(ops):
            r.wait()
        m = int(self.inp[self.max_num, 0].item())
        return self.inp[:m], m
