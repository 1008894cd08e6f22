"""Channel sharding across GPUs and the one collective of the path: the audio gather.

Stations are independent (no state is shared between `Broadcast_FM_Demod` instances in the reference), so the batch
is partitioned into contiguous channel ranges, one per rank, and nothing is exchanged while demodulating.  The only
collective gathers the interleaved stereo audio of a block, `[C_local, n_audio, 2]` per rank, into `[C_total, n_audio, 2]`
(rank order == channel order).  On MI355X nodes the backend is RCCL ("nccl") over xGMI; the same code runs on gloo for
the CPU tests.
"""
from __future__ import annotations


def channel_range(total_channels: int, world_size: int, rank: int) -> tuple[int, int]:
    """Contiguous range [lo, hi) of global channel indices owned by `rank` (remainder spread over the first ranks)."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    base, rem = divmod(total_channels, world_size)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


class AudioGather:
    """Double-buffered all-gather of per-rank audio blocks; issue() is asynchronous, the returned handle's wait() (or
    `drain()`) completes it.  Equal channel counts per rank are required (all_gather_into_tensor)."""

    def __init__(self, dist, torch, c_local: int, n_audio: int, world_size: int, device, depth: int = 2):
        self.dist, self.torch = dist, torch
        self.stage = [torch.empty((c_local, n_audio, 2), dtype=torch.float32, device=device) for _ in range(depth)]
        self.out = [torch.empty((world_size * c_local, n_audio, 2), dtype=torch.float32, device=device) for _ in range(depth)]
        self.handles = [None] * depth
        self.depth = depth

    def issue(self, k: int, audio_local):
        s = k % self.depth
        if self.handles[s] is not None:
            self.handles[s].wait()
        self.stage[s].copy_(audio_local, non_blocking=True)
        self.handles[s] = self.dist.all_gather_into_tensor(self.out[s], self.stage[s], async_op=True)
        return s

    def drain(self):
        for i in range(self.depth):
            if self.handles[i] is not None:
                self.handles[i].wait()
                self.handles[i] = None

    def result(self, slot: int):
        if self.handles[slot] is not None:
            self.handles[slot].wait()
            self.handles[slot] = None
        return self.out[slot]
