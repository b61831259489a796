"""Phase timing of k_nw_ad on one realistic round (dev tool; DADA2HIP_AD_DEBUG skips phases -> wrong results)."""
import os, sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import tperr1
from dada2_amd import api
from dada2_amd.opts import DadaOpts
from dada2_amd.synth import make_sample
d = make_sample(tperr1(), 100000, L=250, G=256, seed=20260925 + 2)
smp = api.Sample.from_derep(d)
os.environ['DADA2HIP_NW_KERNEL'] = 'coop'
for centre in (0, 5):
    for dbg in (0, 8, 12, 14, 15):   # never skip the DP alone: the traceback needs real pointers
        os.environ['DADA2HIP_AD_DEBUG'] = str(dbg)
        ts = []
        for rep in range(4):
            lam, ham, cls, st = smp.compare(centre, tperr1(), DadaOpts(), kdist_cutoff=0.42)
            ts.append(st['nw_kernel_ms'])
        print(f"centre {centre} dbg {dbg:2d} n_nw {st['nnw']:6d} gapless {st['ngapless']:6d} nw_kernel_ms min {min(ts):.3f} screen_ms {st['screen_kernel_ms']:.3f}")
