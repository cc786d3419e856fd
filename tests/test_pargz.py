"""The host input layer inflates ONE gzip file on several threads (slacken_amd/host/pargz.hpp): chunks of the compressed file are
entered at deflate block boundaries found by search, decoded with the preceding 32 KiB unknown, and chained in file order.
Here: byte-for-byte equality with Python's gzip module (zlib) for every shape of file the scheme has a case for -- tiny chunks put
chunk borders inside blocks, headers and trailers -- and zlib's behaviour on corrupt and cut files.  CPU only."""
import gzip
import io
import os
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "slacken_amd", "bin", "slacken-amd")

pytestmark = pytest.mark.skipif(not os.path.exists(CLI), reason="slacken-amd not built")


def gunzip(path, threads=4, chunk=None, expect_fail=False):
    env = dict(os.environ, SLK_GZ_THREADS=str(threads))
    if chunk:
        env["SLK_GZ_CHUNK"] = str(chunk)
    p = subprocess.run([CLI, "gunzip", path], env=env, capture_output=True, timeout=300)
    if expect_fail:
        assert p.returncode != 0, "corrupt input went unnoticed"
        return p.stderr.decode()
    assert p.returncode == 0, p.stderr.decode()
    return p.stdout


def fastq_text(rng, n_reads, read_len=150):
    acgt = np.frombuffer(b"ACGT", np.uint8)
    out = io.BytesIO()
    quals = np.frombuffer(b"FFFFFFFFFFFF:FFF,FFFFFFFF:F,", np.uint8)
    for i in range(n_reads):
        seq = acgt[rng.integers(0, 4, read_len)].tobytes()
        q = quals[rng.integers(0, len(quals), read_len)].tobytes()
        out.write(b"@read%d/1 lane:%d\n%s\n+\n%s\n" % (i, i % 8, seq, q))
    return out.getvalue()


def bgzf(data, block=60000):
    """many small members with an extra field, as bgzip writes them"""
    out = io.BytesIO()
    for a in range(0, max(len(data), 1), block):
        piece = data[a:a + block]
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        body = c.compress(piece) + c.flush()
        bsize = 12 + 6 + len(body) + 8 - 1
        out.write(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + bsize.to_bytes(2, "little") + body +
                  zlib.crc32(piece).to_bytes(4, "little") + (len(piece) & 0xFFFFFFFF).to_bytes(4, "little"))
    out.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))   # bgzip's empty end-of-file member
    return out.getvalue()


def header_with_everything(data):
    c = zlib.compressobj(9, zlib.DEFLATED, -15)
    body = c.compress(data) + c.flush()
    head = b"\x1f\x8b\x08" + bytes([4 | 8 | 16 | 2]) + b"\0\0\0\0\x02\x03" + b"\x05\0hello" + b"name.fq\0" + b"a comment\0"
    head += (zlib.crc32(head) & 0xFFFF).to_bytes(2, "little")
    return head + body + zlib.crc32(data).to_bytes(4, "little") + (len(data) & 0xFFFFFFFF).to_bytes(4, "little")


@pytest.fixture(scope="module")
def cases(tmp_path_factory):
    d = tmp_path_factory.mktemp("pargz")
    rng = np.random.default_rng(5)
    text = fastq_text(rng, 12000)                                    # ~3.7 MB of FASTQ
    rnd = rng.integers(0, 256, 700_000, dtype=np.uint8).tobytes()     # incompressible: stored blocks
    rep = (b"ACGTTGCA" * 40 + b"\n") * 40000                          # long matches, a few huge blocks
    mixed = text[:400_000] + rnd[:300_000] + rep[:500_000] + bytes(200_000) + text[400_000:900_000]
    files = {}

    def put(name, blob, want):
        path = str(d / name)
        open(path, "wb").write(blob)
        files[name] = (path, want)

    for lvl in (1, 6, 9):
        put(f"text_l{lvl}.gz", gzip.compress(text, lvl), text)
    put("random.gz", gzip.compress(rnd, 6), rnd)
    put("repeats.gz", gzip.compress(rep, 6), rep)
    put("zeros.gz", gzip.compress(bytes(30_000_000), 6), bytes(30_000_000))
    put("mixed.gz", gzip.compress(mixed, 6), mixed)
    put("members.gz", b"".join(gzip.compress(text[a:a + 777_777], 6) for a in range(0, len(text), 777_777)), text)
    put("members_with_empty.gz", gzip.compress(text[:100_000]) + gzip.compress(b"") + gzip.compress(text[100_000:]) + gzip.compress(b""), text)
    put("bgzf.gz", bgzf(text), text)
    put("header_fields.gz", header_with_everything(text[:1_000_000]), text[:1_000_000])
    put("garbage_after.gz", gzip.compress(text[:500_000]) + b"this is not a gzip member" * 100, text[:500_000])
    put("fixed_huffman.gz", gzip.compress(b"abc" * 5) + gzip.compress(text[:300_000], 6), b"abc" * 5 + text[:300_000])
    c = zlib.compressobj(6, zlib.DEFLATED, 31)
    flushed = b"".join(c.compress(text[a:a + 50_000]) + c.flush(zlib.Z_FULL_FLUSH) for a in range(0, 1_500_000, 50_000)) + c.flush()
    put("full_flushes.gz", flushed, text[:1_500_000])                  # empty stored blocks between the others
    return files


@pytest.mark.parametrize("chunk", [0, 300_000, 40_000, 5_000, 700])
def test_equals_zlib(cases, chunk):
    for name, (path, want) in cases.items():
        if chunk and chunk < 5_000 and len(want) > 4_000_000:
            continue   # (30 MB of zeros through 700-byte chunks is only slow)
        if chunk == 0 and os.path.getsize(path) < 2 << 20:
            # default chunk size (1 MiB): small files stay with zlib -- still the same bytes
            assert gunzip(path) == want, name
            continue
        got = gunzip(path, chunk=chunk or None)
        assert got == want, (name, chunk, len(got), len(want))


@pytest.mark.parametrize("threads", [1, 2, 16])
def test_thread_counts(cases, threads):
    path, want = cases["text_l6.gz"]
    assert gunzip(path, threads=threads, chunk=20_000) == want
    path, want = cases["bgzf.gz"]
    assert gunzip(path, threads=threads, chunk=20_000) == want


def test_corrupt_and_cut_files_fail_as_with_zlib(cases, tmp_path):
    path, _ = cases["text_l6.gz"]
    blob = bytearray(open(path, "rb").read())
    cut = str(tmp_path / "cut.gz")
    open(cut, "wb").write(blob[:len(blob) // 2])
    assert "read error" in gunzip(cut, chunk=50_000, expect_fail=True)
    cut8 = str(tmp_path / "cut8.gz")
    open(cut8, "wb").write(blob[:-5])                                  # the trailer is incomplete
    assert "read error" in gunzip(cut8, chunk=50_000, expect_fail=True)
    bad_crc = str(tmp_path / "badcrc.gz")
    b2 = bytearray(blob)
    b2[-6] ^= 0x40
    open(bad_crc, "wb").write(b2)
    assert "read error" in gunzip(bad_crc, chunk=50_000, expect_fail=True)
    flipped = str(tmp_path / "flipped.gz")
    b3 = bytearray(blob)
    b3[len(b3) // 3] ^= 0x10                                           # somewhere in the deflate data
    open(flipped, "wb").write(b3)
    assert "read error" in gunzip(flipped, chunk=50_000, expect_fail=True)
    # the same files through zlib alone (SLK_GZ_THREADS=0) and through this decoder on one thread (1) fail too
    for f in (cut, cut8, bad_crc, flipped):
        assert "read error" in gunzip(f, threads=0, expect_fail=True)
        assert "read error" in gunzip(f, threads=1, chunk=50_000, expect_fail=True)


def test_reads_parse_the_same_through_both_inflaters(cases):
    """FASTQ records through the whole input layer: parallel inflate vs zlib"""
    path, _ = cases["text_l6.gz"]
    fq = path.replace(".gz", ".fq.gz")
    if not os.path.exists(fq):
        os.symlink(path, fq)
    outs = []
    for threads, chunk in ((0, None), (1, 30_000), (4, 30_000)):
        env = dict(os.environ, SLK_GZ_THREADS=str(threads))
        if chunk:
            env["SLK_GZ_CHUNK"] = str(chunk)
        p = subprocess.run([CLI, "parse", fq], env=env, capture_output=True, timeout=300)
        assert p.returncode == 0, p.stderr.decode()
        outs.append(p.stdout)
    assert outs[0] == outs[1] == outs[2] and outs[0].count(b"\n") == 12000


def fasta_text(rng, n_records, max_len, width=60, eol=b"\n"):
    acgt = np.frombuffer(b"ACGT", np.uint8)
    out = io.BytesIO()
    for i in range(n_records):
        L = int(rng.integers(1, max_len))
        seq = acgt[rng.integers(0, 4, L)].tobytes()
        out.write(b">seq%d some description" % i + eol)
        for a in range(0, L, width):
            out.write(seq[a:a + width] + eol)
    return out.getvalue()


@pytest.mark.parametrize("chunk", [1_000, 30_000, 250_000])
def test_records_of_files_inflated_in_place(tmp_path, chunk):
    """The readers of `classify` take a gzip file as ONE buffer that fills up while they parse it (pargz.hpp, region mode): records
    that cross segment borders, records longer than several segments, \\r\\n line ends -- against the same file through zlib."""
    rng = np.random.default_rng(chunk)
    files = {
        "reads.fq.gz": gzip.compress(fastq_text(rng, 6000), 6),
        "reads_crlf.fq.gz": gzip.compress(fastq_text(rng, 3000).replace(b"\n", b"\r\n"), 6),
        "long_reads.fq.gz": gzip.compress(fastq_text(rng, 150, read_len=20_000), 6),
        "genomes.fa.gz": gzip.compress(fasta_text(rng, 12, 400_000), 6),           # records of several segments' length
        "contigs_crlf.fa.gz": gzip.compress(fasta_text(rng, 800, 3000, eol=b"\r\n"), 6),
        "members.fq.gz": b"".join(gzip.compress(t, 6) for t in (fastq_text(rng, 1500), b"", fastq_text(rng, 2500))),
        "bgzf.fq.gz": bgzf(fastq_text(rng, 5000)),
    }
    for name, blob in files.items():
        path = str(tmp_path / name)
        open(path, "wb").write(blob)
        outs = []
        for threads in (0, 1, 4):
            env = dict(os.environ, SLK_GZ_THREADS=str(threads), SLK_GZ_CHUNK=str(chunk), SLK_PARSE_THREADS="3")
            p = subprocess.run([CLI, "parse", path], env=env, capture_output=True, timeout=300)
            assert p.returncode == 0, (name, p.stderr.decode())
            outs.append(p.stdout)
        assert outs[0] == outs[1] == outs[2], (name, chunk, outs[0].count(b"\n"), outs[1].count(b"\n"), outs[2].count(b"\n"))
        assert outs[0].count(b"\n") > 0


def test_corrupt_file_inflated_in_place_fails(cases, tmp_path):
    path, _ = cases["text_l6.gz"]
    blob = bytearray(open(path, "rb").read())
    for name, edit in (("cut.fq.gz", lambda b: b[:len(b) // 2]), ("crc.fq.gz", lambda b: b[:-6] + bytes([b[-6] ^ 1]) + b[-5:])):
        f = str(tmp_path / name)
        open(f, "wb").write(edit(blob))
        env = dict(os.environ, SLK_GZ_THREADS="4", SLK_GZ_CHUNK="50000")
        p = subprocess.run([CLI, "parse", f], env=env, capture_output=True, timeout=300)
        assert p.returncode != 0 and b"read error" in p.stderr, name


@pytest.mark.parametrize("seed", range(int(os.environ.get("SLK_GZ_SEEDS", 4))))   # (SLK_GZ_SEEDS: soak runs)
def test_random_streams(tmp_path, seed):
    """content, compression level and strategy (default, filtered, Huffman only, RLE, fixed codes), member cuts, chunk size and
    thread count drawn at random"""
    rng = np.random.default_rng(9000 + seed)
    pieces = []
    for _ in range(int(rng.integers(1, 6))):
        kind = int(rng.integers(0, 5))
        n = int(rng.integers(1, 600_000))
        if kind == 0:
            pieces.append(fastq_text(rng, n // 320 + 1, read_len=int(rng.integers(30, 300))))
        elif kind == 1:
            pieces.append(rng.integers(0, 256, n, dtype=np.uint8).tobytes())
        elif kind == 2:
            unit = rng.integers(65, 91, int(rng.integers(1, 40)), dtype=np.uint8).tobytes()
            pieces.append(unit * (n // len(unit) + 1))
        elif kind == 3:
            pieces.append(bytes(n))
        else:
            pieces.append(rng.integers(0, 4, n, dtype=np.uint8).tobytes())     # low entropy, no structure
    data = b"".join(pieces)
    members, a = [], 0
    while a < len(data) or not members:
        b = len(data) if rng.random() < 0.5 else min(len(data), a + int(rng.integers(0, max(2, len(data)))))
        c = zlib.compressobj(int(rng.integers(1, 10)), zlib.DEFLATED, 31, 8, int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED])))
        blob = c.compress(data[a:b])
        if rng.random() < 0.3:
            blob += c.flush(zlib.Z_SYNC_FLUSH)
        members.append(blob + c.flush())
        a = b
        if a >= len(data):
            break
    path = str(tmp_path / "r.gz")
    open(path, "wb").write(b"".join(members))
    chunk = int(rng.choice([300, 2_000, 17_000, 90_000, 400_000]))
    if os.path.getsize(path) < 2 * chunk:
        chunk = max(64, os.path.getsize(path) // 3)
    got = gunzip(path, threads=int(rng.integers(1, 9)), chunk=chunk)
    assert got == data, (seed, len(got), len(data), chunk)


def test_no_address_space_for_the_region_falls_back_to_zlib(tmp_path):
    """The in-place route reserves address space for the whole inflated file up front (1032 x the compressed size: all or nothing).
    Where that is refused -- ulimit -v, vm.overcommit_memory = 2; here SLK_GZ_RESERVE_LIMIT stands in -- the records must still be
    read, through zlib: a valid .gz is never an error because of the process's address-space limits."""
    import gzip as gz
    rng = np.random.default_rng(31)
    text = fastq_text(rng, 6000)
    path = str(tmp_path / "reads.fq.gz")
    open(path, "wb").write(gz.compress(text, 6))
    outs = []
    for limit in (None, "1000"):
        env = dict(os.environ, SLK_GZ_THREADS="4", SLK_GZ_CHUNK="40000", SLK_PARSE_THREADS="3")
        if limit:
            env["SLK_GZ_RESERVE_LIMIT"] = limit
        p = subprocess.run([CLI, "parse", path], env=env, capture_output=True, timeout=300)
        assert p.returncode == 0, p.stderr.decode()
        outs.append(p.stdout)
    assert outs[0] == outs[1] and outs[0].count(b"\n") == 6000
