"""Development: k_chain under repetition — fresh handles, a few blocks each, every block checked for NaNs in audio / RDS AGC state / symbol counts against the
three-launch form.  Prints the first block that differs.  usage: chain_stress.py [repetitions] [stations]"""
import sys, numpy as np
sys.path[:0] = ["/root/repo", "/root/repo/tests", "/root/repo/oracle"]
import fmradio_loader, synth, torch
pkg = fmradio_loader.load()
reps, n_ch = (int(sys.argv[1]) if len(sys.argv) > 1 else 30), (int(sys.argv[2]) if len(sys.argv) > 2 else 11)
fs, bs, nb = 256_000, 16384, 6
base = np.stack([synth.to_cf32(synth.fm_capture(nb * bs, fs=float(fs), seed=500, channel=c)["iq"]) for c in range(4)])
caps = base[np.arange(n_ch) % 4]
blocks = [torch.from_numpy(np.ascontiguousarray(caps[:, b * bs:(b + 1) * bs])).cuda() for b in range(nb)]
def run(chain):
    dm = pkg.BatchDemod(n_ch, bs, fs, fast_math=True)
    out = []
    for b in range(nb):
        dm.set_chain(chain)
        assert dm.submit(blocks[b]) == 0
        dm.synchronize()
        s_, c_ = dm.rds_symbols()
        out.append((dm.audio().reshape(n_ch, -1).copy(), dm.stream("agc_rds_gain").reshape(-1).copy(), c_.copy(), s_.copy()))
    dm.close()
    return out
ref = run(False)
bad = 0
for rep in range(reps):
    got = run(True)
    for b in range(nb):
        a, g, c, s = got[b]; ra, rg, rc, rs = ref[b]
        da = float(np.sqrt(np.mean((a.astype(np.float64) - ra) ** 2))) if not np.isnan(a).any() else float("nan")
        if np.isnan(a).any() or np.isnan(g).any() or not np.array_equal(c, rc) or not (da < 1e-5) or not np.allclose(g, rg, rtol=1e-3):
            st = [i for i in range(n_ch) if np.isnan(a[i]).any() or np.isnan(g[i]) or c[i] != rc[i] or abs(g[i] - rg[i]) > 1e-3 * abs(rg[i])]
            print(f"rep {rep} block {b}: audio nan {int(np.isnan(a).sum())} rms diff {da:.2e}; gain nan {int(np.isnan(g).sum())}; stations {st}; gain {g[st][:4]} ref {rg[st][:4]}; counts {c[st][:4]} ref {rc[st][:4]}")
            for i in st[:2]:
                idx = np.nonzero(np.isnan(a[i]))[0]
                if idx.size: print("     station", i, "NaN audio floats", idx.size, "first", int(idx[0]), "last", int(idx[-1]), "-> audio samples", int(idx[0]) // 2, "..", int(idx[-1]) // 2, "(tile", int(idx[0]) // 2 // 256, ")")
                sn = np.nonzero(np.isnan(s[i][:c[i]]))[0]
                print("     rds symbols NaN:", sn.size, "of", int(c[i]), "first", (int(sn[0]) if sn.size else None))
            bad += 1
            break
print("repetitions", reps, "bad", bad)
