import sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import tperr1
from dada2_amd import api
from dada2_amd.opts import DadaOpts
from dada2_amd.synth import make_sample
d = make_sample(tperr1(), 100000, L=250, G=256, seed=20260925 + 2)
for rep in range(3):
    t = time.perf_counter(); smp = api.Sample.from_derep(d); t1 = time.perf_counter() - t
    r = smp.run(tperr1(), DadaOpts()); print('create', round(t1*1e3,1), 'ms; ms_upload', round(r.stats['ms_upload'],1)); smp.close()
t = time.perf_counter(); r = api.dada_uniques(d.seqs, d.abundances, None, tperr1(), d.quals, DadaOpts()); print('one-shot dada_uniques', round((time.perf_counter()-t)*1e3,1), 'ms')
