"""Channel sharding across GPUs and the one collective of the path: the gather of every block's outputs (audio, RDS bytes).

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
    two GPUs) — the collector of a deployment; other ranks receive nothing.  mode "rotate": the collector of step k is rank
    (dst + k) % world — at throughput-mode rates one collector would need more than its seven xGMI links (and its one PCIe link to the
    host) carry; rotating it puts 1 / world of every rank's output on each link and spreads the hand-over to the host over every GPU's
    PCIe link.  mode "all": all_gather_into_tensor, every rank ends up with everything (N-1 times the traffic into every GPU).
    dtype torch.int16 carries the 16-bit PCM frames the reference's scraper writes (half the bytes), torch.float32 the raw audio.
    rds_cap > 0: the per-station RDS byte buffers of the on-GPU Manchester decoder ([c_local, rds_cap] uint8) and their counts
    ([c_local] int32) travel with every block — what the reference hands its second observer per station (src/app.cpp:27-34)."""

    def __init__(self, dist, torch, c_local: int, n_audio: int, world_size: int, device, depth: int = 2, mode: str = "root",
                 dtype=None, dst: int = 0, rds_cap: int = 0):
        if mode not in ("root", "all", "rotate"):
            raise ValueError("mode must be 'root', 'rotate' or 'all'")
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
        holds_all = mode in ("all", "rotate") or self.rank == dst
        self.out = [torch.empty((world_size * c_local, n_audio, 2), dtype=self.dtype, device=device) if holds_all else None
                    for _ in range(depth)]
        # RDS payload of a rank and block: [c_local * rds_cap] bytes, then [c_local] int32 counts as bytes
        self.rds_cap, self.c_local = int(rds_cap), int(c_local)
        self.rds_bytes = c_local * (self.rds_cap + 4) if self.rds_cap > 0 else 0
        self.stage_rds = [torch.empty((self.rds_bytes,), dtype=torch.uint8, device=device) if self.rds_bytes else None for _ in range(depth)]
        self.out_rds = [torch.empty((world_size, self.rds_bytes), dtype=torch.uint8, device=device) if (self.rds_bytes and holds_all) else None
                        for _ in range(depth)]
        self.handles = [None] * depth
        self.collector = [dst] * depth           # which rank collected (collects) the slot's block
        self.depth = depth

    def collector_of(self, k: int) -> int:
        return (self.dst + k) % self.world if self.mode == "rotate" else self.dst

    def slot(self, k: int) -> int:
        """Staging slot of step k, free to be refilled (its previous gather has completed).  On RCCL ("nccl") Work.wait() orders the
        CURRENT STREAM behind the collective and returns at once — the host is not blocked; on gloo (the CPU plumbing tests) it
        blocks the calling thread, which is what a host-side collective is."""
        s = k % self.depth
        self._wait(s)
        return s

    def _wait(self, s: int):
        if self.handles[s] is not None:
            for w in self.handles[s]:
                w.wait()
            self.handles[s] = None

    def launch(self, s: int, k: int = 0):
        """Start the gather of staging slot s (already filled on the current stream); k = the step (mode "rotate": its collector)."""
        # the collectives move bytes: neither RCCL nor gloo has a 16-bit integer type
        as_bytes = (lambda t: t.view(self.torch.uint8)) if self.dtype == self.torch.int16 else (lambda t: t)
        if self.mode in ("root", "rotate"):   # the mode is fixed at construction: no per-rank fallback that could desynchronise the collective
            dst = self.collector_of(k)
            self.collector[s] = dst
            parts = [as_bytes(t) for t in self.out[s].chunk(self.world, dim=0)] if self.rank == dst else None
            works = [self.dist.gather(as_bytes(self.stage[s]), parts, dst=dst, async_op=True)]
            if self.rds_bytes:
                rparts = list(self.out_rds[s].unbind(0)) if self.rank == dst else None
                works.append(self.dist.gather(self.stage_rds[s], rparts, dst=dst, async_op=True))
            self.handles[s] = works
            return s
        works = [self.dist.all_gather_into_tensor(as_bytes(self.out[s]), as_bytes(self.stage[s]), async_op=True)]
        if self.rds_bytes:
            works.append(self.dist.all_gather_into_tensor(self.out_rds[s].view(-1), self.stage_rds[s], async_op=True))
        self.handles[s] = works
        return s

    def stage_rds_views(self, s: int):
        """(bytes [c_local, rds_cap] uint8, counts [c_local] int32) views of staging slot s, to be filled before launch()."""
        t = self.stage_rds[s]
        nb = self.c_local * self.rds_cap
        return t[:nb].view(self.c_local, self.rds_cap), t[nb:].view(self.torch.int32)

    def issue(self, k: int, audio_local, rds_bytes=None, rds_counts=None):
        s = self.slot(k)
        self.stage[s].copy_(audio_local if audio_local.dtype == self.dtype else pcm16_frames(audio_local), non_blocking=True)
        if self.rds_bytes:
            b, c = self.stage_rds_views(s)
            b.copy_(rds_bytes, non_blocking=True); c.copy_(rds_counts, non_blocking=True)
        return self.launch(s, k)

    def drain(self):
        for i in range(self.depth):
            self._wait(i)

    def holds(self, slot: int) -> bool:
        return self.mode == "all" or self.rank == self.collector[slot]

    def result(self, slot: int):
        """The gathered block of a slot (None on ranks that did not collect it)."""
        self._wait(slot)
        return self.out[slot] if self.holds(slot) else None

    def result_rds(self, slot: int):
        """(bytes [world * c_local, rds_cap] uint8, counts [world * c_local] int32) of a slot's block, None where result() is None."""
        self._wait(slot)
        if not self.rds_bytes or not self.holds(slot):
            return None
        t = self.out_rds[slot]
        nb = self.c_local * self.rds_cap
        return t[:, :nb].reshape(self.world * self.c_local, self.rds_cap), t[:, nb:].contiguous().view(self.torch.int32).reshape(-1)
