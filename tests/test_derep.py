"""The dereplication front-end (dada2hip_derep_fastq, host-side C++; R/sequenceIO.R:45-124, :150-183) against the Python
restatement in dada2_amd/io.py and, where the reference's fixtures are on disk, against the committed golden inputs.
No GPU is involved: everything here runs in the CPU suite."""
import gzip
import os

import numpy as np
import pytest

from dada2_amd import api, io as dio
from dada2_amd._lib import Dada2HipError
from helpers import load_input

REF_EXT = "/root/reference/inst/extdata"


def write_fastq(path, seqs, quals, crlf=False, gz=False):
    nl = b"\r\n" if crlf else b"\n"
    op = gzip.open if gz else open
    with op(path, "wb") as fh:
        for i, (s, q) in enumerate(zip(seqs, quals)):
            fh.write(b"@r%d some description" % i + nl + s.encode() + nl + b"+" + nl + q + nl)


def random_reads(seed, nreads, nvar=40, L=(30, 60), qlo=35, qhi=74, zero_every=0):
    rng = np.random.default_rng(seed)
    variants = ["".join(rng.choice(list("ACGT"), size=int(rng.integers(L[0], L[1] + 1)))) for _ in range(nvar)]
    w = rng.dirichlet(np.full(nvar, 0.3))
    seqs, quals = [], []
    for i in range(nreads):
        s = variants[int(rng.choice(nvar, p=w))]
        if rng.random() < 0.3:   # a point error: more singletons and abundance ties
            p = int(rng.integers(len(s)))
            s = s[:p] + "ACGT"[int(rng.integers(4))] + s[p + 1:]
        if zero_every and i % zero_every == zero_every - 1:
            s = ""
        seqs.append(s)
        quals.append(bytes(rng.integers(qlo, qhi + 1, size=len(s)).astype(np.uint8)))
    return seqs, quals


def assert_same(got, want):
    assert got.seqs == want.seqs
    np.testing.assert_array_equal(got.abundances, want.abundances)
    np.testing.assert_array_equal(got.map, want.map)
    assert got.quals.shape == want.quals.shape
    np.testing.assert_array_equal(np.isnan(got.quals), np.isnan(want.quals))
    np.testing.assert_array_equal(np.nan_to_num(got.quals, nan=-1.0), np.nan_to_num(want.quals, nan=-1.0))   # bit-exact means


@pytest.mark.parametrize("gz,crlf", [(False, False), (True, False), (False, True)])
def test_derep_matches_the_restatement(tmp_path, gz, crlf):
    seqs, quals = random_reads(1, 3000, zero_every=97)
    p = tmp_path / ("r.fastq.gz" if gz else "r.fastq")
    write_fastq(p, seqs, quals, crlf=crlf, gz=gz)
    got = api.derep_fastq(str(p))
    want = dio.derep_from_reads(seqs, quals)
    assert_same(got, want)
    assert (got.map == -1).sum() == sum(1 for s in seqs if not s)
    assert np.all(np.diff(got.abundances) <= 0)
    # ties in abundance keep C-locale lexical order (one chunk)
    for a in np.unique(got.abundances):
        blk = [s for s, x in zip(got.seqs, got.abundances) if x == a]
        assert blk == sorted(blk)


@pytest.mark.parametrize("n", [1, 7, 500, 2999, 3000])
def test_chunked_reading_appends_new_uniques_after_the_earlier_ones(tmp_path, n):
    seqs, quals = random_reads(2, 3000, zero_every=211)
    p = tmp_path / "r.fastq.gz"
    write_fastq(p, seqs, quals, gz=True)
    got = api.derep_fastq(str(p), n=n)
    assert_same(got, dio.derep_from_reads(seqs, quals, n=n))
    one = api.derep_fastq(str(p))
    assert sorted(got.seqs) == sorted(one.seqs) and int(got.abundances.sum()) == int(one.abundances.sum())


def test_quality_encoding_auto_detection_and_explicit_offset(tmp_path):
    seqs, quals = random_reads(3, 500, qlo=66, qhi=104)    # Phred+64: nothing below ';'
    p = tmp_path / "old.fastq"
    write_fastq(p, seqs, quals)
    auto = api.derep_fastq(str(p))
    assert_same(auto, dio.derep_from_reads(seqs, quals, offset=64))
    forced = api.derep_fastq(str(p), qual_offset=33)
    assert_same(forced, dio.derep_from_reads(seqs, quals, offset=33))


def test_errors_of_the_reference_are_kept(tmp_path):
    with pytest.raises(Dada2HipError, match="Not all provided files exist"):
        api.derep_fastq(str(tmp_path / "missing.fastq.gz"))
    p = tmp_path / "empty_reads.fastq"
    write_fastq(p, ["", ""], [b"", b""])
    with pytest.raises(Dada2HipError, match="Only zero-length sequences detected during dereplication"):
        api.derep_fastq(str(p))
    bad = tmp_path / "bad.fastq"
    bad.write_bytes(b"@r0\nACGT\n+\nIII\n")
    with pytest.raises(Dada2HipError, match="malformed FASTQ"):
        api.derep_fastq(str(bad))


@pytest.mark.skipif(not os.path.isdir(REF_EXT), reason="reference fixtures not on this machine")
@pytest.mark.parametrize("name", ["sam1F", "sam1R", "sam2F", "sam2R"])
def test_reference_fixtures_give_the_golden_inputs(name):
    got = api.derep_fastq(f"{REF_EXT}/{name}.fastq.gz")
    want = load_input(name)
    assert got.seqs == list(want.seqs)
    np.testing.assert_array_equal(got.abundances, want.abundances)
    np.testing.assert_array_equal(np.nan_to_num(got.quals, nan=-1.0), np.nan_to_num(want.quals, nan=-1.0))
    assert len(got.map) == int(got.abundances.sum())


@pytest.mark.gpu
def test_native_derep_feeds_a_resident_sample_without_a_host_copy(tmp_path):
    from dada2_amd.synth import make_sample
    from helpers import assert_results_equal, tperr1
    from dada2_amd.io import extend_err
    from oracle import ref
    seqs, quals = random_reads(5, 6000, nvar=12, L=(120, 120), qlo=48, qhi=73)
    p = tmp_path / "s.fastq.gz"
    write_fastq(p, seqs, quals, gz=True)
    nd = api.NativeDerep(str(p))
    d = nd.to_derep()
    err = extend_err(tperr1(), 40)
    s = api.Sample.from_native(nd, device=0)
    got = s.run(err)
    want = ref.dada_uniques(d.seqs, d.abundances, None, err, d.quals, None)
    assert_results_equal(got, want)
    s.close()
    nd.close()
