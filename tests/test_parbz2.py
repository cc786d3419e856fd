"""A bzip2 FILE is decompressed on several threads (slacken_amd/host/parbz2.hpp): its blocks are found by their 48-bit magic at
any bit position, cut out, wrapped into one-block streams and handed to libbz2 in parallel.  Byte-for-byte equality with Python's
bz2 module for every level, concatenated streams, data that inflates fifty-fold, chunk borders everywhere; corrupt and cut files
fail.  CPU only."""
import bz2
import os
import subprocess

import numpy as np
import pytest

from test_pargz import CLI, fastq_text

pytestmark = pytest.mark.skipif(not os.path.exists(CLI), reason="slacken-amd not built")


def bunzip(path, threads=4, chunk=None, expect_fail=False):
    env = dict(os.environ, SLK_GZ_THREADS=str(threads))
    if chunk:
        env["SLK_BZ2_CHUNK"] = str(chunk)
    p = subprocess.run([CLI, "gunzip", path], env=env, capture_output=True, timeout=300)
    if expect_fail:
        assert p.returncode != 0, "corrupt input went unnoticed"
        return p.stderr.decode()
    assert p.returncode == 0, p.stderr.decode()
    return p.stdout


@pytest.fixture(scope="module")
def cases(tmp_path_factory):
    d = tmp_path_factory.mktemp("parbz2")
    rng = np.random.default_rng(8)
    text = fastq_text(rng, 9000)
    rnd = rng.integers(0, 256, 500_000, dtype=np.uint8).tobytes()
    runs = b"".join(bytes([int(rng.integers(65, 70))]) * int(rng.integers(1, 3000)) for _ in range(3000))   # inflates ~50-fold per block
    files = {}

    def put(name, blob, want):
        path = str(d / name)
        open(path, "wb").write(blob)
        files[name] = (path, want)

    for lvl in (1, 5, 9):
        put(f"text_l{lvl}.bz2", bz2.compress(text, lvl), text)
    put("random.bz2", bz2.compress(rnd, 9), rnd)
    put("runs.bz2", bz2.compress(runs, 9), runs)
    put("zeros.bz2", bz2.compress(bytes(20_000_000), 9), bytes(20_000_000))
    put("streams.bz2", b"".join(bz2.compress(text[a:a + 400_000], int(rng.integers(1, 10))) for a in range(0, len(text), 400_000)), text)
    put("streams_with_empty.bz2", bz2.compress(text[:300_000], 1) + bz2.compress(b"") + bz2.compress(text[300_000:], 2), text)
    put("empty.bz2", bz2.compress(b"") + bz2.compress(b"") + bz2.compress(b"x" * 70000, 1) * 3, b"x" * 210000)
    return files


@pytest.mark.parametrize("chunk", [0, 200_000, 30_000, 3_000])
def test_equals_libbz2(cases, chunk):
    for name, (path, want) in cases.items():
        got = bunzip(path, chunk=chunk or None)
        assert got == want, (name, chunk, len(got), len(want))
        assert bz2.decompress(open(path, "rb").read()) == want


def test_thread_counts_and_records(cases, tmp_path):
    path, want = cases["text_l9.bz2"]
    for threads in (1, 2, 16):
        assert bunzip(path, threads=threads, chunk=40_000) == want
    fq = str(tmp_path / "reads.fq.bz2")
    os.symlink(path, fq)
    outs = []
    for threads in (1, 6):
        env = dict(os.environ, SLK_GZ_THREADS=str(threads), SLK_BZ2_CHUNK="50000")
        p = subprocess.run([CLI, "parse", fq], env=env, capture_output=True, timeout=300)
        assert p.returncode == 0, p.stderr.decode()
        outs.append(p.stdout)
    assert outs[0] == outs[1] and outs[0].count(b"\n") == 9000


def test_corrupt_and_cut_files_fail(cases, tmp_path):
    path, _ = cases["text_l9.bz2"]
    blob = bytearray(open(path, "rb").read())
    cut = str(tmp_path / "cut.bz2")
    open(cut, "wb").write(blob[:len(blob) // 2])
    assert "read error" in bunzip(cut, chunk=50_000, expect_fail=True)
    flipped = str(tmp_path / "flipped.bz2")
    b2 = bytearray(blob)
    b2[len(b2) // 3] ^= 0x08
    open(flipped, "wb").write(b2)
    assert "read error" in bunzip(flipped, chunk=50_000, expect_fail=True)
    notbz = str(tmp_path / "not.bz2")
    open(notbz, "wb").write(b"@r\nACGT\n+\nIIII\n" * 100000)
    assert "read error" in bunzip(notbz, chunk=50_000, expect_fail=True)


def test_streaming_decoder_reads_every_stream(cases, tmp_path):
    """SLK_GZ_THREADS=0 (and pipes): libbz2's streaming decoder, restarted for every concatenated stream -- BZ2_bzread, which the
    input layer used before, stops after the first one without a word"""
    for name in ("streams.bz2", "streams_with_empty.bz2", "empty.bz2", "text_l9.bz2", "runs.bz2"):
        path, want = cases[name]
        assert bunzip(path, threads=0) == want, name
    path, want = cases["streams.bz2"]
    p = subprocess.run(f"cat {path} | {CLI} gunzip /dev/stdin.bz2", shell=True, capture_output=True, timeout=120)   # (name decides the codec)
    blob = bytearray(open(path, "rb").read())
    cut = str(tmp_path / "cut.bz2")
    open(cut, "wb").write(blob[:len(blob) // 2])
    assert "read error" in bunzip(cut, threads=0, expect_fail=True)
    bad = str(tmp_path / "bad.bz2")
    blob[len(blob) // 3] ^= 0x08
    open(bad, "wb").write(blob)
    assert "read error" in bunzip(bad, threads=0, expect_fail=True)


def _bits(blob):
    return "".join(format(b, "08b") for b in blob)


def _bytes_of(bits):
    bits = bits + "0" * (-len(bits) % 8)
    return int(bits, 2).to_bytes(len(bits) // 8, "big") if bits else b""


BLOCK = format(0x314159265359, "048b")
EOS = format(0x177245385090, "048b")


def test_stream_structure_is_checked_on_both_routes(tmp_path):
    """What a serial decoder checks BETWEEN the blocks, the parallel route must check too (block CRCs alone let these pass):
      * a file cut at a block boundary that falls on a byte boundary  -> every block decodes, the stream never ends
      * a stream with a whole block missing                          -> every block decodes, the combined CRC is wrong
      * bytes behind the last stream (zeros, text)                    -> no bzip2 data
      * an empty file
    Errors on the threaded route (SLK_GZ_THREADS=4) and on libbz2's streaming decoder (SLK_GZ_THREADS=0) alike."""
    rng = np.random.default_rng(12)
    text = fastq_text(rng, 30000)
    blob = bz2.compress(text, 1)                      # ~100 kB of text per block
    bits = _bits(blob)
    starts = []
    at = bits.find(BLOCK)
    while at >= 0:
        starts.append(at)
        at = bits.find(BLOCK, at + 48)
    eos = bits.rfind(EOS)
    assert len(starts) > 40 and eos > starts[-1]

    def both_fail(name, data):
        path = str(tmp_path / name)
        open(path, "wb").write(data)
        for threads in (4, 0):
            err = bunzip(path, threads=threads, chunk=100_000, expect_fail=True)
            assert "read error" in err, (name, threads, err)

    def both_ok(name, data, want):
        path = str(tmp_path / name)
        open(path, "wb").write(data)
        for threads in (4, 0):
            assert bunzip(path, threads=threads, chunk=100_000) == want, (name, threads)

    aligned = [s for s in starts[1:] if s % 8 == 0]
    assert aligned, "no block boundary of this file falls on a byte boundary: change the seed"
    both_fail("cut_at_block.bz2", blob[:aligned[0] // 8])
    # one block spliced out, the rest moved up bit by bit (the stream's CRC at its end now belongs to other blocks)
    i = len(starts) // 2
    both_fail("block_missing.bz2", _bytes_of(bits[:starts[i]] + bits[starts[i + 1]:eos + 80]))
    both_fail("zeros_behind.bz2", blob + bytes(1000))
    both_fail("text_behind.bz2", blob + b"@r\nACGT\n+\nIIII\n")
    both_fail("empty_file.bz2", b"")
    both_fail("header_only.bz2", b"BZh9")
    # and what must still pass: the file itself, two streams back to back, an empty stream in the middle
    both_ok("whole.bz2", blob, text)
    both_ok("two_streams.bz2", blob + bz2.compress(text[:50_000], 9), text + text[:50_000])
    both_ok("empty_stream_between.bz2", blob + bz2.compress(b"") + blob, text + text)
