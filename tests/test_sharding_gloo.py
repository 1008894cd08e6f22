"""The N>1 path on CPU: two gloo ranks shard a channel batch, run a stand-in per-rank 'demodulation' (the oracle on each
rank's own channels — test infrastructure, CPU only) and gather the audio exactly the way bench.py does on RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import fmradio_loader


def test_channel_range_partitions():
    pkg = fmradio_loader.load()
    for total, world in ((65536, 8), (4096, 1), (10, 4), (7, 8)):
        seen = []
        for r in range(world):
            lo, hi = pkg.channel_range(total, world, r)
            assert 0 <= lo <= hi <= total
            seen += list(range(lo, hi))
        assert seen == list(range(total))
    with pytest.raises(ValueError):
        pkg.channel_range(8, 2, 2)


def _worker(rank: int, world: int, port: int, tmp: str, mode: str, pcm16: bool):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    for p in (root, root / "tests", root / "oracle"):
        sys.path.insert(0, str(p))
    import fmradio_loader as fl
    import oraclelib as O
    import synth
    pkg = fl.load()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    total, bs, nb = 6, 4096, 3
    lo, hi = pkg.channel_range(total, world, rank)
    c_local = hi - lo
    n_audio = bs // 32
    gather = pkg.AudioGather(dist, torch, c_local, n_audio, world, torch.device("cpu"), mode=mode,
                             dtype=torch.int16 if pcm16 else torch.float32)
    demods = [O.Demod(bs, 1_024_000) for _ in range(c_local)]
    caps = [synth.to_cf32(synth.fm_capture(bs * nb, seed=50, channel=lo + i)["iq"]) for i in range(c_local)]
    results = []
    for k in range(nb):
        local = np.stack([(d.process_cf32(c[k * bs:(k + 1) * bs]), d.get("audio"))[1].reshape(n_audio, 2) for d, c in zip(demods, caps)])
        slot = gather.issue(k, torch.from_numpy(local))
        got = gather.result(slot)
        assert (got is None) == (mode == "root" and rank != 0)     # only the collector holds the gathered block in root mode
        if got is not None:
            results.append(got.clone())
    gather.drain()
    if results:
        np.save(os.path.join(tmp, f"gathered{rank}.npy"), torch.stack(results).numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode,pcm16", [("root", True), ("root", False), ("all", False)])
def test_two_rank_gather_equals_single_process(tmp_path, mode, pcm16):
    import oraclelib as O
    import synth
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path), mode, pcm16), nprocs=2, join=True)
    collectors = [0] if mode == "root" else [0, 1]
    assert (tmp_path / "gathered1.npy").exists() == (mode == "all")
    total, bs, nb = 6, 4096, 3
    want = np.empty((nb, total, bs // 32, 2), np.float32)
    for c in range(total):
        d = O.Demod(bs, 1_024_000)
        cap = synth.to_cf32(synth.fm_capture(bs * nb, seed=50, channel=c)["iq"])
        for k in range(nb):
            d.process_cf32(cap[k * bs:(k + 1) * bs])
            want[k, c] = d.get("audio").reshape(-1, 2)
    if pcm16:   # the reference scraper's frames: sample * (32767 * 0.95f), truncated toward zero
        want = (want * (np.float32(32767.0) * np.float32(0.95))).astype(np.int32).astype(np.int16)
    for r in collectors:
        got = np.load(tmp_path / f"gathered{r}.npy")          # [blocks, channels, n_audio, 2]
        assert got.dtype == want.dtype and np.array_equal(got.view(np.uint8), want.view(np.uint8)), (mode, pcm16, r)
