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
    assert pkg.padded_shard(65536, 8) == 8192 and pkg.padded_shard(10, 4) == 3 and pkg.padded_shard(7, 8) == 1


def _uneven_worker(rank: int, world: int, port: int, tmp: str):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import fmradio_loader as fl
    pkg = fl.load()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = pkg.channel_range(7, world, rank)          # 4 + 3 channels: not a legal gather layout
    try:
        pkg.AudioGather(dist, torch, hi - lo, 16, world, torch.device("cpu"))
        outcome = "constructed"
    except ValueError as e:
        outcome = "ValueError" if "equal shards" in str(e) else f"other: {e}"
    # equal, padded shards are accepted on every rank
    pkg.AudioGather(dist, torch, pkg.padded_shard(7, world), 16, world, torch.device("cpu"))
    with open(os.path.join(tmp, f"uneven{rank}.txt"), "w") as f:
        f.write(outcome)
    dist.barrier()
    dist.destroy_process_group()


def test_uneven_shards_fail_on_every_rank_alike(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_uneven_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert [(tmp_path / f"uneven{r}.txt").read_text() for r in range(2)] == ["ValueError", "ValueError"]


def test_bench_refuses_a_world_size_mismatch_and_launches_its_own_ranks():
    """bench.py --gpus N: under a launcher WORLD_SIZE must equal N (a mismatch is an error, not silently ignored); without a
    launcher it starts `torch.distributed.run` as a child process itself and returns the child's exit code (here, without a
    GPU, the ranks fail: the parent must report that with a non-zero code and no result line)."""
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True)
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr
    if torch.cuda.is_available():
        return
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--preroll", "0", "--channels", "8",
                        "--backend", "gloo", "--share-gpu", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and '"metric"' not in r.stdout


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
    rds_cap = 16 if mode == "rotate" else 0          # (rotate: with the RDS byte buffers and counts, as bench.py gathers them)
    gather = pkg.AudioGather(dist, torch, c_local, n_audio, world, torch.device("cpu"), mode=mode,
                             dtype=torch.int16 if pcm16 else torch.float32, rds_cap=rds_cap)
    demods = [O.Demod(bs, 1_024_000) for _ in range(c_local)]
    caps = [synth.to_cf32(synth.fm_capture(bs * nb, seed=50, channel=lo + i)["iq"]) for i in range(c_local)]
    results = []
    for k in range(nb):
        local = np.stack([(d.process_cf32(c[k * bs:(k + 1) * bs]), d.get("audio"))[1].reshape(n_audio, 2) for d, c in zip(demods, caps)])
        # stand-in RDS payload: station (lo + i) of block k carries bytes k + 16 (lo + i) + j and the count 16
        rb = torch.tensor([[(k + 16 * (lo + i) + j) & 255 for j in range(16)] for i in range(c_local)], dtype=torch.uint8) if rds_cap else None
        rc = torch.full((c_local,), 16, dtype=torch.int32) if rds_cap else None
        slot = gather.issue(k, torch.from_numpy(local), rb, rc)
        got = gather.result(slot)
        collects = mode == "all" or rank == (k % world if mode == "rotate" else 0)
        assert (got is None) == (not collects)     # only the block's collector holds it (root: rank 0; rotate: rank k mod world)
        if got is not None:
            results.append((k, got.clone()))
            if rds_cap:
                gb, gc = gather.result_rds(slot)
                assert gb.shape == (total, 16) and torch.equal(gc, torch.full((total,), 16, dtype=torch.int32))
                assert all(int(gb[c, j]) == ((k + 16 * c + j) & 255) for c in range(total) for j in (0, 7, 15))
    gather.drain()
    if results:
        np.save(os.path.join(tmp, f"gathered{rank}.npy"), torch.stack([t for _, t in results]).numpy())
        np.save(os.path.join(tmp, f"blocks{rank}.npy"), np.array([k for k, _ in results]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode,pcm16", [("root", True), ("root", False), ("all", False), ("rotate", True)])
def test_two_rank_gather_equals_single_process(tmp_path, mode, pcm16):
    import oraclelib as O
    import synth
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path), mode, pcm16), nprocs=2, join=True)
    collectors = [0] if mode == "root" else [0, 1]
    assert (tmp_path / "gathered1.npy").exists() == (mode != "root")
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
        blocks = np.load(tmp_path / f"blocks{r}.npy")         # which blocks this rank collected (rotate: k mod world == r)
        assert list(blocks) == ([k for k in range(nb) if k % 2 == r] if mode == "rotate" else list(range(nb)))
        assert got.dtype == want.dtype and np.array_equal(got.view(np.uint8), want[blocks].view(np.uint8)), (mode, pcm16, r)
