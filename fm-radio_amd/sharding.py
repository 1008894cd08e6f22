"""Channel sharding across GPUs and the one collective of the path: the audio gather.

Stations are independent (no state is shared between `Broadcast_FM_Demod` instances in the reference), so the batch
is partitioned into contiguous channel ranges, one per rank, and nothing is exchanged while demodulating.  The only
collective gathers the interleaved stereo audio of a block, `[C_local, n_audio, 2]` per rank, into `[C_total, n_audio, 2]`
(rank order == channel order) on the collecting rank — in bench.py as the 16-bit PCM frames the reference's scraper writes.
On MI355X nodes the backend is RCCL ("nccl") over xGMI; the same code runs on gloo for the CPU tests.
"""
from __future__ import annotations


def channel_range(total_channels: int, world_size: int, rank: int) -> tuple[int, int]:
    """Contiguous range [lo, hi) of global channel indices owned by `rank` (remainder spread over the first ranks).
    AudioGather needs equal shards: use padded_shard() for the per-rank batch size when total % world != 0."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    base, rem = divmod(total_channels, world_size)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def padded_shard(total_channels: int, world_size: int) -> int:
    """Per-rank batch size that covers `total_channels` with EQUAL shards (the last ranks carry idle padding channels)."""
    return -(-total_channels // world_size)


def pcm16_frames(audio):
    """The reference scraper's float -> int16 conversion (fm_scraper.cpp:79-82): sample * (32767 * 0.95f), truncated toward
    zero.  `audio` is a torch float32 tensor (any device); the device path uses the library's own kernel instead."""
    import torch
    scale = torch.tensor(32767.0, dtype=torch.float32) * torch.tensor(0.95, dtype=torch.float32)   # the float product, as compiled
    return (audio * scale).to(torch.int32).to(torch.int16)


class AudioGather:
    """Double-buffered gather of per-rank audio blocks [c_local, n_audio, 2] into [world * c_local, n_audio, 2] (rank order ==
    channel order).  issue() is asynchronous; drain() / result() complete it.

    mode "root": every rank sends its block straight to rank `dst` (point-to-point over the direct xGMI link between the
    two GPUs) — the collector of a deployment; other ranks receive nothing.  mode "all": all_gather_into_tensor, every rank
    ends up with everything (N-1 times the traffic into every GPU).  dtype torch.int16 carries the 16-bit PCM frames the
    reference's scraper writes (half the bytes), torch.float32 the raw audio."""

    def __init__(self, dist, torch, c_local: int, n_audio: int, world_size: int, device, depth: int = 2, mode: str = "root",
                 dtype=None, dst: int = 0):
        if mode not in ("root", "all"):
            raise ValueError("mode must be 'root' or 'all'")
        if not (0 <= dst < world_size):
            raise ValueError("dst out of range")
        # every rank contributes the same number of channels (the gathered block is a plain concatenation in rank order):
        # checked once, collectively, so that an uneven shard fails on every rank alike instead of corrupting the gather
        sizes = [None] * world_size
        dist.all_gather_object(sizes, int(c_local))
        if len(set(sizes)) != 1:
            raise ValueError(f"AudioGather needs equal shards on every rank, got {sizes}: pad the last shard with idle channels")
        self.dist, self.torch, self.mode, self.dst = dist, torch, mode, dst
        self.dtype = dtype if dtype is not None else torch.float32
        self.rank = dist.get_rank()
        self.world = world_size
        self.stage = [torch.empty((c_local, n_audio, 2), dtype=self.dtype, device=device) for _ in range(depth)]
        holds_all = mode == "all" or self.rank == dst
        self.out = [torch.empty((world_size * c_local, n_audio, 2), dtype=self.dtype, device=device) if holds_all else None
                    for _ in range(depth)]
        self.handles = [None] * depth
        self.depth = depth

    def slot(self, k: int) -> int:
        """Staging slot of step k, free to be refilled (its previous gather has completed).  On RCCL ("nccl") Work.wait() orders the
        CURRENT STREAM behind the collective and returns at once — the host is not blocked; on gloo (the CPU plumbing tests) it
        blocks the calling thread, which is what a host-side collective is."""
        s = k % self.depth
        if self.handles[s] is not None:
            self.handles[s].wait()
            self.handles[s] = None
        return s

    def launch(self, s: int):
        """Start the gather of staging slot s (already filled on the current stream)."""
        # the collectives move bytes: neither RCCL nor gloo has a 16-bit integer type
        as_bytes = (lambda t: t.view(self.torch.uint8)) if self.dtype == self.torch.int16 else (lambda t: t)
        if self.mode == "root":   # the mode is fixed at construction: no per-rank fallback that could desynchronise the collective
            parts = [as_bytes(t) for t in self.out[s].chunk(self.world, dim=0)] if self.rank == self.dst else None
            self.handles[s] = self.dist.gather(as_bytes(self.stage[s]), parts, dst=self.dst, async_op=True)
            return s
        self.handles[s] = self.dist.all_gather_into_tensor(as_bytes(self.out[s]), as_bytes(self.stage[s]), async_op=True)
        return s

    def issue(self, k: int, audio_local):
        s = self.slot(k)
        self.stage[s].copy_(audio_local if audio_local.dtype == self.dtype else pcm16_frames(audio_local), non_blocking=True)
        return self.launch(s)

    def drain(self):
        for i in range(self.depth):
            if self.handles[i] is not None:
                self.handles[i].wait()
                self.handles[i] = None

    def result(self, slot: int):
        """The gathered block of a slot (None on ranks that do not collect)."""
        if self.handles[slot] is not None:
            self.handles[slot].wait()
            self.handles[slot] = None
        return self.out[slot]
