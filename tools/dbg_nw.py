import os, sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import load_input, tperr1
from dada2_amd import api
from dada2_amd.opts import DadaOpts
d = load_input('sam1F')
smp = api.Sample.from_derep(d)
res = {}
for kern in ('lane', 'coop'):
    os.environ['DADA2HIP_NW_KERNEL'] = kern
    lam, ham, cls, st = smp.compare(0, tperr1(), DadaOpts(), kdist_cutoff=1.0)
    res[kern] = (lam.copy(), ham.copy(), cls.copy())
a, b = res['lane'], res['coop']
nw = a[2] == 3
print('n nw', nw.sum(), 'cls equal', np.array_equal(a[2], b[2]))
bad = np.nonzero(nw & ((a[0] != b[0]) | (a[1] != b[1])))[0]
print('mismatch', len(bad), 'of', nw.sum())
for r in bad[:12]:
    print(r, a[0][r], b[0][r], a[1][r], b[1][r], len(d.seqs[r]))
ok = np.nonzero(nw & (a[0] == b[0]) & (a[1] == b[1]))[0]
print('ok examples', ok[:10], [int(a[1][r]) for r in ok[:10]])
