"""Development tool (GPU box): how often the L-R phase estimate's sign decision (reference broadcast_fm_demod.cpp:500-510) falls
differently in the tolerance mode than in the oracle, per station-block, and what the whole-run RMS errors are."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (R, R + "/tests", R + "/oracle"): sys.path.insert(0, p)
import numpy as np
from concurrent.futures import ProcessPoolExecutor
import synth
fs, bs = 256000, 16384
nst = int(sys.argv[1]) if len(sys.argv) > 1 else 16
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 32
seed = 7700

def cap(c): return synth.to_cf32(synth.fm_capture(nb * bs, fs=float(fs), seed=seed, channel=c)["iq"])

def oracle(args):
    c, coeffs_bytes = args
    import oraclelib as O, ctypes as C
    k = O.Coeffs.from_buffer_copy(coeffs_bytes)
    o = O.run_chain(cap(c), bs, fs, u8=False, coeffs=k, streams=["lmr", "audio", "lmr_phase"])
    return c, o["lmr"], o["audio"], o["lmr_phase"]

if __name__ == "__main__":
    import fmradio_loader
    from gpu_parity import run_gpu
    import ctypes as C
    pkg = fmradio_loader.load()
    caps = np.stack([cap(c) for c in range(nst)])
    g = run_gpu(pkg, caps, bs, fs, fast_math=True)
    with ProcessPoolExecutor(min(nst, os.cpu_count() or 8)) as ex:
        res = list(ex.map(oracle, [(c, bytes(g["coeffs"][c])) for c in range(nst)]))
    flips = 0; per = []
    worst_lmr = worst_audio = 0.0
    for c, lmr, audio, ph in res:
        dl = (np.asarray(g["lmr"][c], np.float64) - lmr).reshape(nb, -1); da = (np.asarray(g["audio"][c], np.float64).reshape(-1) - audio.reshape(-1)).reshape(nb, -1)
        dph = np.asarray(g["lmr_phase"][c], np.float64).reshape(-1)[:nb] - ph.reshape(-1)[:nb]
        jumps = np.abs(np.diff(np.concatenate([[0.0], dph]))) > 7e-4       # one flipped estimate moves the offset by 0.1 pi / 205 = 1.5e-3
        flips += int(jumps.sum())
        worst_lmr = max(worst_lmr, float(np.sqrt((dl ** 2).mean()))); worst_audio = max(worst_audio, float(np.sqrt((da ** 2).mean())))
        per.append(np.sqrt((dl ** 2).mean(axis=1)))
    per = np.array(per)
    print("stations %d blocks %d: flipped estimates %d (%.2f %% of station-blocks); whole-run RMS worst station: lmr %.2e audio %.2e; median per-block lmr rms %.2e; blocks > 1e-4: %d"
          % (nst, nb, flips, 100.0 * flips / (nst * nb), worst_lmr, worst_audio, float(np.median(per)), int((per > 1e-4).sum())))
