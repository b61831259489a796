"""CPU checks of the boundary: the C-ABI library loads and exports every symbol that
include/dada2hip.h declares, and (no GPU here) compute entry points fail loudly instead of
falling back to anything."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from dada2_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib.lib()


def test_header_symbols_all_exported(lib):
    from dada2_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "dada2hip.h")).read()
    declared = sorted(set(re.findall(r"\b(dada2hip_[a-zA-Z0-9_]+)\s*\(", hdr)))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/dada2hip.h but not exported"
    assert sorted(_lib.EXPORTS) == declared
    assert lib.dada2hip_version().startswith(b"dada2hip")


def test_opts_struct_layout():
    import ctypes as C
    from dada2_amd.opts import COpts, DadaOpts
    assert C.sizeof(COpts) == 112
    co = DadaOpts().to_c()
    assert (co.match, co.mismatch, co.gap, co.homo_gap, co.band_size) == (5, -4, -8, -8, 16)
    assert (co.use_kmers, co.use_quals, co.final_consensus, co.vectorized_alignment, co.SSE, co.gapless, co.greedy) == (
        1, 1, 0, 1, 2, 1, 1)
    assert co.kdist_cutoff == 0.42 and co.omegaA == 1e-40 and co.omegaC == 1e-40
    # R/dada.R:222-237 normalisation
    o = DadaOpts(GAP_PENALTY=8, HOMOPOLYMER_GAP_PENALTY=-1).to_c()
    assert o.gap == -8 and o.homo_gap == -1 and o.vectorized_alignment == 0
    assert DadaOpts(BAND_SIZE=0).to_c().vectorized_alignment == 0


def test_no_cpu_fallback_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from dada2_amd import api, _lib
    from helpers import tperr1
    with pytest.raises(_lib.Dada2HipError) as ei:
        api.dada_uniques(["ACGTACGTAC", "ACGTACGTAA"], [5, 1], None, tperr1(), np.full((2, 10), 30.0))
    assert ei.value.code == 2 and "no HIP device" in str(ei.value)


def test_input_validation_messages(lib):
    """Validation happens before any device work and keeps the reference's messages (Rmain.cpp:52-78)."""
    from dada2_amd import api, _lib
    from helpers import tperr1
    q = np.full((1, 4), 30.0)
    with pytest.raises(_lib.Dada2HipError, match="kmer-size"):
        api.dada_uniques(["ACGT"], [1], None, tperr1(), q)
    with pytest.raises(_lib.Dada2HipError, match="Zero input"):
        api.dada_uniques([], [], None, tperr1(), None)
    with pytest.raises(_lib.Dada2HipError, match="16 rows"):
        api.dada_uniques(["ACGTACGT"], [1], None, np.ones((4, 41)), np.full((1, 8), 30.0))
    with pytest.raises(_lib.Dada2HipError, match="associated qualities"):
        api.dada_uniques(["ACGTACGT", "ACGTACG"], [2, 1], None, tperr1(), np.full((2, 9), 30.0))
