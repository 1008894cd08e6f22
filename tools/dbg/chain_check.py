"""Development check of k_chain (fmd_debug_set_chain): the one-launch form against the three-launch form and against the oracle, 256 kSa/s cf32.
usage: chain_check.py [stations] [blocks]"""
import sys, numpy as np
sys.path[:0] = ["/root/repo", "/root/repo/tests", "/root/repo/oracle"]
import fmradio_loader, synth, oraclelib as O
from gpu_parity import lib_coeffs_to_oracle
pkg = fmradio_loader.load()
fs, bs = 256_000, 16384
n_ch, nb = int(sys.argv[1]) if len(sys.argv) > 1 else 11, int(sys.argv[2]) if len(sys.argv) > 2 else 10
caps = np.stack([synth.to_cf32(synth.fm_capture(nb * bs, fs=float(fs), seed=500, channel=c)["iq"]) for c in range(n_ch)])
import torch
def run(chain):
    dm = pkg.BatchDemod(n_ch, bs, fs, fast_math=True)
    dm.set_chain(chain)
    au, sy, by = [], [[] for _ in range(n_ch)], [b"" for _ in range(n_ch)]
    for b in range(nb):
        t = torch.from_numpy(np.ascontiguousarray(caps[:, b * bs:(b + 1) * bs])).cuda()
        assert dm.submit(t) == 0
        dm.synchronize()
        au.append(dm.audio().reshape(n_ch, -1).copy())
        s_, c_ = dm.rds_symbols()
        b_, bc = dm.rds_bytes()
        for c in range(n_ch):
            sy[c].append(s_[c, :c_[c]].copy()); by[c] += b_[c, :bc[c]].tobytes()
    k = [dm.get_coeffs(c) for c in range(n_ch)]
    nchain = dm.chain_blocks()
    dm.close()
    return np.concatenate(au, axis=1), [np.concatenate(x) for x in sy], by, k, nchain
a1, s1, b1, k, n1 = run(True)
a0, s0, b0, _, n0 = run(False)
print("chain blocks:", n1, n0)
na = a1.shape[1] // nb
for c in range(min(n_ch, 4)):
    o = O.run_chain(caps[c], bs, fs, u8=False, coeffs=lib_coeffs_to_oracle(k[c]), streams=["audio"])
    oa = o["audio"].reshape(-1)
    e1 = [float(np.sqrt(np.mean((a1[c, b * na:(b + 1) * na].astype(np.float64) - oa[b * na:(b + 1) * na]) ** 2))) for b in range(nb)]
    e0 = [float(np.sqrt(np.mean((a0[c, b * na:(b + 1) * na].astype(np.float64) - oa[b * na:(b + 1) * na]) ** 2))) for b in range(nb)]
    print(c, "chain vs oracle per block:", " ".join(f"{x:.1e}" for x in e1))
    print(c, "three vs oracle per block:", " ".join(f"{x:.1e}" for x in e0))
    print(c, "bytes equal (chain/three vs oracle):", np.array_equal(np.frombuffer(b1[c], np.uint8), o["rds_bytes"]), np.array_equal(np.frombuffer(b0[c], np.uint8), o["rds_bytes"]), "sym counts", s1[c].size, s0[c].size)
d = a1.astype(np.float64) - a0
print("chain vs three, audio rms per station:", " ".join(f"{x:.1e}" for x in np.sqrt((d ** 2).mean(axis=1))))
print("nan:", int(np.isnan(a1).sum()))
