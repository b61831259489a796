"""Golden vectors for mergePairs (R/paired.R:92-201), produced by the REFERENCE's own C code (C_nwalign / C_eval_pair /
C_pair_consensus compiled in place into oracle/_ref) under the R-level bookkeeping restated in oracle/merge.py:
    python tests/golden/make_merge_golden.py
writes tests/golden/merge_pairs.npz = the synthetic inputs (merge_cases()) + the expected rows of every option set, and
tests/golden/sam1_maps.npz = the derepFastq read -> unique maps of the reference's sam1F / sam1R fixtures (needed to
merge the two denoised fixtures on a box without /root/reference)."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from merge_cases import OPTION_SETS, make_case  # noqa: E402
from oracle import merge as omerge, ref  # noqa: E402


def main():
    out = {}
    for seed in (1, 2, 3):
        c = make_case(seed)
        for k in ("fwd", "rev", "n0F", "n0R"):
            out[f"c{seed}_{k}"] = np.asarray(c[k], dtype=np.int32)
        out[f"c{seed}_seqsF"] = np.array(c["seqsF"])
        out[f"c{seed}_seqsR"] = np.array(c["seqsR"])
        for name, kw in OPTION_SETS.items():
            rows = omerge.merge_pairs(c["fwd"], c["rev"], c["seqsF"], c["n0F"], c["seqsR"], c["n0R"], ref, return_rejects=True, **kw)
            out[f"c{seed}_{name}"] = np.array(json.dumps(rows))
    np.savez_compressed(os.path.join(HERE, "merge_pairs.npz"), **out)
    REF = "/root/reference/inst/extdata"
    if os.path.isdir(REF):
        from oracle.derep import derep_fastq
        maps = {}
        for fq in ("sam1F", "sam1R"):
            maps[fq] = derep_fastq(f"{REF}/{fq}.fastq.gz").map
        np.savez_compressed(os.path.join(HERE, "sam1_maps.npz"), **maps)
    print("written", len(out), "arrays")


if __name__ == "__main__":
    main()
