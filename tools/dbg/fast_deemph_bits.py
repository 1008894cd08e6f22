import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, "."); sys.path.insert(0, "oracle")
import fmradio_loader; pkg = fmradio_loader.load()
import test_gpu_fast as T, oraclelib as O
from gpu_parity import lib_coeffs_to_oracle, run_gpu, oracle_controls
from fm_radio_amd.capi import default_controls
fs, bs = 1_024_000, 32768
def ctl(**kw):
    c = default_controls()
    for k, v in kw.items(): setattr(c, k, v)
    return c
nb = -(-fs * 12 // 10 // bs)
caps = T._caps(4, nb * bs, float(fs), seed=9500)
per = {0: ctl(use_deemphasis=1, deemphasis_tus=50), 1: ctl(use_deemphasis=1, deemphasis_tus=75), 3: ctl(use_deemphasis=1, deemphasis_tus=50, audio_out=1)}
g = run_gpu(pkg, caps, bs, fs, fast_math=True, per_channel_controls=per)
for c in range(4):
    o = O.run_chain(caps[c], bs, fs, u8=False, controls=oracle_controls(per[c]) if c in per else None, coeffs=lib_coeffs_to_oracle(g["coeffs"][c]), streams=["rds_sym", "audio", "fm_out_iq"])
    print("  audio rms err", T.rms(np.asarray(g["audio"][c], np.float64).reshape(-1) - o["audio"].reshape(-1)), "fm_out_iq", T.rms(np.asarray(g["fm_out_iq"][c], np.float64).reshape(-1) - o["fm_out_iq"].reshape(-1)))
    a, b = T.rds_bits(g["rds_bytes"][c]), T.rds_bits(o["rds_bytes"])
    n = min(a.size, b.size)
    best = None
    for sh in range(-24, 25):
        lo = max(0, -sh); m = n - abs(sh) - 24
        d = a[lo:lo + m] != b[lo + sh:lo + sh + m]
        last = np.nonzero(d)[0]
        k = (last[-1] if last.size else -1)
        if best is None or k < best[1]: best = (sh, k, int(d.sum()))
    print("  counts gpu", g["rds_count"][c][:12], "ora", o["rds_count"][:12]); print("channel", c, "bits", a.size, b.size, "best shift", best[0], "last differing bit", best[1], "n diff", best[2], "counts equal", np.array_equal(g["rds_count"][c], o["rds_count"]))
