"""The dereplication front-end (dada2hip_derep_fastq, host-side C++; R/sequenceIO.R:45-124, :150-183) against
  * the CPU restatement under oracle/derep.py (test infrastructure; cites the reference line by line),
  * the committed golden inputs (tests/golden/sam*.input.npz: sequences, abundances, INTEGER quality sums; sam1_maps.npz),
  * facts that do not come from this repo's author at all: the unique counts SURVEY.md records for the reference's
    fixtures, and coreutils `sort | uniq -c` run on the fixture files (sequences, abundances and the tie order).
No GPU is involved: everything here runs in the CPU suite."""
import gzip
import os

import numpy as np
import pytest

from dada2_amd import api
from oracle import derep as dio
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


FIXTURES = ["sam1F", "sam1R", "sam2F", "sam2R", "samPB"]
needs_fixtures = pytest.mark.skipif(not os.path.isdir(REF_EXT), reason="reference fixtures not on this machine")


@needs_fixtures
@pytest.mark.parametrize("name", FIXTURES)
def test_reference_fixtures_give_the_golden_inputs(name):
    """Byte for byte: sequences, abundances and the integer per-position quality sums the goldens hold (the means the
    library returns are sum / abundance in fp64: comparing sums is the stricter test)."""
    import numpy as np
    from helpers import GOLDEN
    got = api.derep_fastq(f"{REF_EXT}/{name}.fastq.gz")
    z = np.load(os.path.join(GOLDEN, name + ".input.npz"))
    assert got.seqs == [str(s) for s in z["seqs"]]
    np.testing.assert_array_equal(got.abundances, z["abundances"])
    qsum = z["qsum"]
    back = np.rint(np.nan_to_num(got.quals, nan=0.0) * got.abundances[:, None]).astype(np.int64)
    back[np.isnan(got.quals)] = -1
    want = qsum.astype(np.int64)
    want[qsum < 0] = -1
    np.testing.assert_array_equal(back, want)
    # ... and the means are exactly the quotient the reference forms (sequenceIO.R:95)
    want_mean = load_input(name).quals
    np.testing.assert_array_equal(np.nan_to_num(got.quals, nan=-1.0), np.nan_to_num(want_mean, nan=-1.0))
    assert len(got.map) == int(got.abundances.sum())
    if name in ("sam1F", "sam1R"):
        np.testing.assert_array_equal(got.map, np.load(os.path.join(GOLDEN, "sam1_maps.npz"))[name])


@needs_fixtures
def test_known_unique_counts_of_the_reference_fixtures():
    """SURVEY.md §4/§6 (from the reference's own documentation run): sam1F.fastq.gz holds 1 500 reads in 896 uniques."""
    d = api.derep_fastq(f"{REF_EXT}/sam1F.fastq.gz")
    assert d.nraw == 896 and int(d.abundances.sum()) == 1500 and len(d.map) == 1500


@needs_fixtures
@pytest.mark.parametrize("name", FIXTURES)
def test_uniques_and_order_against_coreutils(name, tmp_path):
    """An implementation nobody here wrote: `sort | uniq -c` in the C locale, then a stable sort by count, is exactly
    derepFastq's order for a one-chunk file (srsort order inside equal abundances, sequenceIO.R:98,161)."""
    import shutil
    import subprocess
    if not all(shutil.which(t) for t in ("sh", "zcat", "awk", "sort", "uniq")):
        pytest.skip("coreutils not available")
    cmd = (f"zcat {REF_EXT}/{name}.fastq.gz | awk 'NR % 4 == 2' | LC_ALL=C sort | LC_ALL=C uniq -c | "
           "LC_ALL=C sort -s -k1,1nr")
    out = subprocess.run(["sh", "-c", cmd], capture_output=True, text=True, check=True).stdout.split("\n")
    rows = [ln.split() for ln in out if ln.strip()]
    got = api.derep_fastq(f"{REF_EXT}/{name}.fastq.gz")
    assert got.seqs == [r[1] for r in rows]
    assert got.abundances.tolist() == [int(r[0]) for r in rows]


def test_chunk_boundaries_with_phred64_detection(tmp_path):
    """Phred+64 file read in chunks smaller than the file: the encoding is decided on the first chunk (derep.cpp) and the
    uniques of later chunks are appended behind the earlier ones (sequenceIO.R:85-88)."""
    seqs, quals = random_reads(11, 900, qlo=66, qhi=104, zero_every=53)
    p = tmp_path / "old64.fastq.gz"
    write_fastq(p, seqs, quals, gz=True)
    for n in (13, 250, 899):
        assert_same(api.derep_fastq(str(p), n=n), dio.derep_from_reads(seqs, quals, n=n, offset=64))


def test_corrupt_gzip_is_an_error_not_a_short_result(tmp_path):
    """A damaged .gz must fail loudly: never a silently truncated derep object, never a misleading 'zero-length' message."""
    seqs, quals = random_reads(12, 4000)
    p = tmp_path / "r.fastq.gz"
    write_fastq(p, seqs, quals, gz=True)
    raw = bytearray(p.read_bytes())
    cut = tmp_path / "cut.fastq.gz"
    cut.write_bytes(bytes(raw[: len(raw) // 2]))                 # truncated stream (no trailer)
    with pytest.raises(Dada2HipError, match="error reading"):
        api.derep_fastq(str(cut))
    flipped = bytearray(raw)
    for k in (len(raw) // 2, len(raw) // 2 + 1):
        flipped[k] ^= 0xFF
    bad = tmp_path / "flip.fastq.gz"
    bad.write_bytes(bytes(flipped))
    with pytest.raises(Dada2HipError):                            # data error or (if the damage decodes) a malformed record
        api.derep_fastq(str(bad))
    notrailer = tmp_path / "notrailer.fastq.gz"
    notrailer.write_bytes(bytes(raw[:-8]))                        # CRC32 + ISIZE trailer missing
    with pytest.raises(Dada2HipError, match="error reading"):
        api.derep_fastq(str(notrailer))


def test_dereplication_works_in_a_forked_child(tmp_path):
    """The marshalling pool spawns its threads once per process; a fork()ed child (multiprocessing's default start method,
    R's mclapply) has none of them and must rebuild the pool instead of waiting for workers that do not exist."""
    seqs, quals = random_reads(13, 20000, nvar=400, L=(200, 250))
    p = tmp_path / "r.fastq"
    write_fastq(p, seqs, quals)
    want = api.derep_fastq(str(p))                                # the parent's pool exists from here on
    rfd, wfd = os.pipe()
    pid = os.fork()
    if pid == 0:                                                  # child
        import signal
        signal.alarm(60)
        ok = b"0"
        try:
            got = api.derep_fastq(str(p))
            ok = b"1" if got.seqs == want.seqs and np.array_equal(got.abundances, want.abundances) else b"2"
        finally:
            os.write(wfd, ok)
            os._exit(0)
    os.close(wfd)
    _, status = os.waitpid(pid, 0)
    assert os.read(rfd, 1) == b"1" and status == 0
    os.close(rfd)


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


def test_large_chunks_take_the_pool_sorts_and_batched_hashing(tmp_path, monkeypatch):
    """Enough distinct reads in a chunk (> 65 536) that the chunk order and the abundance order go through the sample sort over the
    host pool, several batches of records per chunk, the table grown several times - against the restatement, with two chunk sizes
    (one chunk; a boundary inside the file), shared 20-nt prefixes (what a primer does to the leading bytes of every key) and reads
    that are prefixes of other reads."""
    monkeypatch.setenv("DADA2HIP_HOST_THREADS", "4")
    rng = np.random.default_rng(77)
    n = 90000
    prefix = "ACGTTGCAAGGCTTAACCGT"
    codes = rng.integers(0, 4, size=(n, 40), dtype=np.uint8)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    body = lut[codes]
    lens = rng.integers(25, 41, size=n)
    seqs = [prefix + body[i, : lens[i]].tobytes().decode() for i in range(n)]
    for i in range(0, n, 9):            # duplicates and prefixes of other reads
        seqs[i] = seqs[(i * 7 + 3) % n]
    for i in range(1, n, 1000):
        seqs[i] = seqs[i - 1][:30]
    quals = [bytes(rng.integers(40, 70, size=len(s)).astype(np.uint8)) for s in seqs]
    p = tmp_path / "big.fastq"
    write_fastq(p, seqs, quals)
    want = dio.derep_from_reads(seqs, quals)
    got = api.derep_fastq(str(p))
    assert len(got.seqs) > 65536
    assert_same(got, want)
    assert_same(api.derep_fastq(str(p), n=70001), dio.derep_from_reads(seqs, quals, n=70001))


def test_gzip_members_and_both_inflate_paths_agree(tmp_path, monkeypatch):
    """A .gz made of several members (cat a.gz b.gz) is read through, as gzread does; the one-shot libdeflate path (where the
    system has the library) and zlib's streaming inflate give the same object."""
    seqs, quals = random_reads(21, 5000, zero_every=211)
    a, b = tmp_path / "a.fastq.gz", tmp_path / "b.fastq.gz"
    write_fastq(a, seqs[:2000], quals[:2000], gz=True)
    write_fastq(b, seqs[2000:], quals[2000:], gz=True)
    both = tmp_path / "both.fastq.gz"
    both.write_bytes(a.read_bytes() + b.read_bytes())
    want = dio.derep_from_reads(seqs, quals)
    assert_same(api.derep_fastq(str(both)), want)
    monkeypatch.setenv("DADA2HIP_DEREP_INFLATE", "zlib")
    assert_same(api.derep_fastq(str(both)), want)
