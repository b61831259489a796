"""Phase timing of k_nw_ad on one realistic round (dev tool; DADA2HIP_AD_DEBUG skips phases -> wrong results).
Latency regime: the skip mask keeps ~4100 NW + ~2500 gapless comparisons, like a typical bench round."""
import os, sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import tperr1
from dada2_amd import api
from dada2_amd.opts import DadaOpts
from dada2_amd.synth import make_sample
d = make_sample(tperr1(), 100000, L=250, G=256, seed=20260925 + 2)
smp = api.Sample.from_derep(d)
os.environ['DADA2HIP_NW_KERNEL'] = 'coop'
os.environ['DADA2HIP_AD_DEBUG'] = '0'
lam, ham, cls, st = smp.compare(5, tperr1(), DadaOpts(), kdist_cutoff=0.42)
rng = np.random.default_rng(0)
for n_nw, n_gl in ((4100, 2500), (2000, 1000), (8000, 4000), (60060, 26640)):
    nw_idx = np.nonzero(cls == 3)[0]; gl_idx = np.nonzero(cls == 2)[0]
    keep = np.concatenate([rng.choice(nw_idx, min(n_nw, nw_idx.size), replace=False), rng.choice(gl_idx, min(n_gl, gl_idx.size), replace=False)])
    skip = np.ones(d.nraw, dtype=np.uint8); skip[keep] = 0
    for dbg in (0, 8, 12, 14, 15):   # never skip the DP alone: the traceback needs real pointers
        os.environ['DADA2HIP_AD_DEBUG'] = str(dbg)
        ts = []
        for rep in range(5):
            l2, h2, c2, st = smp.compare(5, tperr1(), DadaOpts(), kdist_cutoff=0.42, skip=skip)
            ts.append(st['nw_kernel_ms'])
        print(f"n_nw {st['nnw']:6d} gapless {st['ngapless']:6d} dbg {dbg:2d} nw_kernel_ms min {min(ts):.4f} median {sorted(ts)[2]:.4f}")
