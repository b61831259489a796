import sys, time, os
sys.path[:0]=['/root/repo','/root/repo/tests']
from dada2_amd import _lib
sys.path.insert(0, '/root/repo/tests/emu'); import build as emu_build; _lib.LIB_PATH = emu_build.build()
from helpers import case_inputs, assert_results_equal, WHOLE_PATH_CASES
from dada2_amd import api
import os
for name in [c for c in WHOLE_PATH_CASES if c != "sam2R_singletons" or os.environ.get("EMU_ALL")]:
    d, err, pri, opts, exp, meta = case_inputs(name)
    t0=time.time()
    try:
        got = api.dada_uniques(d.seqs, d.abundances, pri, err, d.quals, opts)
        assert_results_equal(got, exp, check_birth_from=pri is None)
        print(name, 'OK', round(time.time()-t0,1),'s', got.nclust, flush=True)
    except Exception as e:
        print(name, 'FAIL', repr(e)[:200], flush=True)
