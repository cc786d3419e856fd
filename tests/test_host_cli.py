"""C++ host layer (slacken_amd/host, `slacken-amd` binary) against the Python restatement in hostmodel.py: index parameters,
taxonomy + Kraken report, FASTA/FASTQ parsing and pairing.  These subcommands need no GPU."""
import gzip
import os
import subprocess

import numpy as np
import pytest

import hostmodel
import taxgen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "slacken_amd", "bin", "slacken-amd")


@pytest.fixture(scope="module", autouse=True)
def _built():
    if not os.path.exists(CLI):
        subprocess.check_call(["make", "-C", ROOT, "slacken_amd/bin/slacken-amd"])


def run(*args):
    return subprocess.run([CLI, *map(str, args)], check=True, capture_output=True, text=True).stdout


def write_taxonomy(d, parents, rng, ranked=0.5):
    """nodes.dmp / names.dmp in NCBI's `\t|\t` format; some nodes get a standard rank, others 'no rank' or 'subspecies'"""
    os.makedirs(d, exist_ok=True)
    titles = hostmodel.RANKS[2:] + ["no rank", "subspecies", "clade"]
    nodes, names = [], []
    for t in range(1, len(parents)):
        if t != 1 and parents[t] == 0:
            continue
        r = titles[int(rng.integers(len(titles)))] if rng.random() < ranked else "no rank"
        nodes.append((t, 1 if t == 1 else int(parents[t]), r))
        names.append((t, f"Taxon number {t}"))
    with open(os.path.join(d, "nodes.dmp"), "w") as f:
        for t, p, r in nodes:
            f.write(f"{t}\t|\t{p}\t|\t{r}\t|\tXX\t|\t0\t|\n")
    with open(os.path.join(d, "names.dmp"), "w") as f:
        for t, nm in names:
            f.write(f"{t}\t|\tsynonym of {t}\t|\t\t|\tsynonym\t|\n")
            f.write(f"{t}\t|\t{nm}\t|\t\t|\tscientific name\t|\n")
    return hostmodel.Taxonomy(nodes, names)


def test_props(tmp_path):
    loc = str(tmp_path / "lib")
    with open(loc + ".properties", "w") as f:   # as java.util.Properties.store writes it (HDFSUtil.writeProperties)
        f.write("#Properties for Slacken\n#Sun Oct 04 09:00:00 UTC 2026\nk=35\nm=31\nbuckets=2000\nversion=1\n"
                "splitter=randomXOR\nminimizerSpaces=7\nXORmask=-1905890618887226962\ncanonical=true\n")
    assert run("props", loc).strip() == "k=35 m=31 spaces=7 xorMask=-1905890618887226962 canonical=1"
    with open(loc + ".properties", "w") as f:
        f.write("k=31\nm=12\nversion=1\nsplitter=randomXOR\n")
    # defaults: no spaces, DEFAULT_TOGGLE_MASK (MinimizerPriorities.scala:150), canonical
    assert run("props", loc).strip() == f"k=31 m=12 spaces=0 xorMask={np.uint64(0xe37e28c4271b5a2d).astype(np.int64)} canonical=1"
    with open(loc + ".properties", "w") as f:
        f.write("k=31\nm=12\nversion=2\nsplitter=randomXOR\n")
    assert subprocess.run([CLI, "props", loc], capture_output=True).returncode != 0


def test_java_percent_format():
    # java.util.Formatter: HALF_UP on the shortest decimal digits (0.125 -> 0.13, 12.345 -> 12.35), unlike printf
    cases = ((0.125, "  0.13"), (0.375, "  0.38"), (2.5, "  2.50"), (99.995, "100.00"), (100.0, "100.00"), (0.0, "  0.00"),
             (12.345, " 12.35"), (0.004, "  0.00"), (0.005, "  0.01"), (1e-7, "  0.00"), (33.333333333333336, " 33.33"))
    for x, want in cases:
        assert hostmodel.fmt_6_2f(x) == want
    # through the binary: a two-leaf taxonomy whose clade fractions are the cases above (x% = a / b)
    for a, b, want in ((1, 800, "  0.13"), (2469, 20000, " 12.35"), (19999, 20000, "100.00"), (1, 3, " 33.33"), (1, 20000, "  0.01")):
        import tempfile
        with tempfile.TemporaryDirectory() as d:
            with open(os.path.join(d, "nodes.dmp"), "w") as f:
                f.write("1\t|\t1\t|\tno rank\t|\n2\t|\t1\t|\tgenus\t|\n3\t|\t1\t|\tgenus\t|\n")
            with open(os.path.join(d, "names.dmp"), "w") as f:
                f.write("1\t|\troot\t|\t\t|\tscientific name\t|\n")
            with open(os.path.join(d, "c.tsv"), "w") as f:
                f.write(f"2\t{a}\n3\t{b - a}\n")
            lines = run("report", d, os.path.join(d, "c.tsv")).split("\n")
            row = [l for l in lines if l.split("\t")[4:5] == ["2"]][0]
            assert row.split("\t")[0] == want, (a, b, row)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_report_matches_restatement(tmp_path, seed):
    rng = np.random.default_rng(seed)
    parents = taxgen.taxonomy(200, rng)
    tax = write_taxonomy(str(tmp_path / "tax"), parents, rng)
    defined = [t for t in range(1, len(parents)) if t == 1 or parents[t] != 0]
    chosen = rng.choice(defined, size=40, replace=False)
    counts = [(int(t), int(rng.integers(0, 100))) for t in chosen] + [(0, 8)]
    with open(tmp_path / "counts.tsv", "w") as f:
        for t, c in counts:
            f.write(f"{t}\t{c}\n")
    got = run("report", tmp_path / "tax", tmp_path / "counts.tsv").rstrip("\n").split("\n")
    want, own, clade = hostmodel.kraken_report(tax, counts)
    assert got == want
    # the reference's property spec (KrakenReportProps.scala:33-52) on what was printed
    rows = {int(l.split("\t")[4]): (int(l.split("\t")[1]), int(l.split("\t")[2])) for l in got[1:]}
    for t, c in counts:
        if t in rows:
            assert rows[t][1] == c and rows[t][0] >= c
    assert rows[1][0] == sum(c for t, c in counts if t != 0)
    assert rows[0] == (8, 8)


def test_report_ties_keep_child_order(tmp_path):
    # equal clade counts: the stable sort keeps Taxonomy.children's order (descending id, because children are prepended)
    nodes = [(1, 1, "no rank"), (2, 1, "genus"), (3, 1, "genus"), (4, 1, "genus")]
    names = [(t, f"n{t}") for t in (1, 2, 3, 4)]
    d = tmp_path / "tax"
    os.makedirs(d)
    with open(d / "nodes.dmp", "w") as f:
        for t, p, r in nodes:
            f.write(f"{t}\t|\t{p}\t|\t{r}\t|\n")
    with open(d / "names.dmp", "w") as f:
        for t, nm in names:
            f.write(f"{t}\t|\t{nm}\t|\t\t|\tscientific name\t|\n")
    with open(tmp_path / "c.tsv", "w") as f:
        f.write("2\t5\n3\t5\n4\t7\n")
    got = run("report", d, tmp_path / "c.tsv").rstrip("\n").split("\n")
    assert [int(l.split("\t")[4]) for l in got[1:]] == [1, 4, 3, 2]
    assert got == hostmodel.kraken_report(hostmodel.Taxonomy(nodes, names), [(2, 5), (3, 5), (4, 7)])[0]


FASTA = ">seq1 some description\nACGTACGT\nTTGGCC\n>seq2\r\nAAAA\r\nCCCC\r\n>empty_header_only\n>seq3 x y\nGATTACA\n"
FASTQ = ("@r1/1 extra\nACGTNACGT\n+\nIIIIIIIII\n@r2/1\nGGGGCCCC\n+r2\n@+@+@+@+\n@r3/1\nTTTT\n+\n@III\n")


def test_parse_fasta_fastq(tmp_path):
    fa = tmp_path / "a.fasta"
    fa.write_text(FASTA)
    got = [tuple(l.split("\t")) for l in run("parse", fa).rstrip("\n").split("\n")]
    assert got == hostmodel.parse_fasta(FASTA) == [("seq1", "ACGTACGTTTGGCC"), ("seq2", "AAAACCCC"), ("seq3", "GATTACA")]
    fq = tmp_path / "a.fastq"
    fq.write_text(FASTQ)
    got = [tuple(l.split("\t")) for l in run("parse", fq).rstrip("\n").split("\n")]
    assert got == hostmodel.parse_fastq(FASTQ) == [("r1/1", "ACGTNACGT"), ("r2/1", "GGGGCCCC"), ("r3/1", "TTTT")]
    gz = tmp_path / "b.fq.gz"
    with gzip.open(gz, "wt") as f:
        f.write(FASTQ)
    assert run("parse", gz) == run("parse", fq)


def test_parse_reference_testdata_shapes(tmp_path):
    # multi-line FASTA with > 2 lines per record and a first record preceded by nothing, as in the reference's testData/*.fasta
    rng = np.random.default_rng(5)
    recs = [(f"id{i}", "".join(rng.choice(list("ACGT"), size=int(rng.integers(1, 200))))) for i in range(50)]
    text = "".join(f">{h} len={len(s)}\n" + "\n".join(s[j:j + 60] for j in range(0, len(s), 60)) + "\n" for h, s in recs)
    p = tmp_path / "m.fa"
    p.write_text(text)
    got = [tuple(l.split("\t")) for l in run("parse", p).rstrip("\n").split("\n")]
    assert got == recs == hostmodel.parse_fasta(text)


def test_parse_paired(tmp_path):
    a = tmp_path / "x_1.fq"
    b = tmp_path / "x_2.fq"
    a.write_text("@p1/1\nAAAA\n+\nIIII\n@p2/1\nCCCC\n+\nIIII\n@lonely/1\nGG\n+\nII\n")
    b.write_text("@p2/2\nTTTT\n+\nIIII\n@p1/2\nGGGG\n+\nIIII\n")
    got = [tuple(l.split("\t")) for l in run("parse", a, b).rstrip("\n").split("\n")]
    # inner join on the header without /1 and /2 (InputReader.scala:105-131)
    assert got == [("p1", "AAAA", "GGGG"), ("p2", "CCCC", "TTTT")]


def test_classify_without_gpu_fails_loudly(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    loc = str(tmp_path / "lib")
    with open(loc + ".properties", "w") as f:
        f.write("k=35\nm=31\nversion=1\nsplitter=randomXOR\nminimizerSpaces=7\n")
    os.makedirs(loc + "_taxonomy")
    with open(loc + "_taxonomy/nodes.dmp", "w") as f:
        f.write("1\t|\t1\t|\tno rank\t|\n2\t|\t1\t|\tgenus\t|\n")
    with open(loc + "_taxonomy/names.dmp", "w") as f:
        f.write("1\t|\troot\t|\t\t|\tscientific name\t|\n")
    import struct
    with open(loc + ".slkrec", "wb") as f:
        f.write(b"SLKREC1\0" + struct.pack("<QII", 1, 1, 0) + struct.pack("<q", 12345) + struct.pack("<i", 2))
    (tmp_path / "r.fa").write_text(">a\nACGT\n")
    r = subprocess.run([CLI, "classify", "-i", loc, "-o", str(tmp_path / "out"), str(tmp_path / "r.fa")], capture_output=True, text=True)
    assert r.returncode != 0 and "slk_index_create" in r.stderr


@pytest.mark.parametrize("sep", ["\n", "\r\n", "\r"])
@pytest.mark.parametrize("comp", ["", ".gz", ".bz2"])
def test_input_round_trip_separators_and_compression(tmp_path, sep, comp):
    """T/kmers/InputReaderProps.scala:93-148: FASTA / FASTQ records written with each of the three line separators and with
    no / gzip / bzip2 compression read back as the same (id, nucleotides) fragments (mixed case, ambiguous codes kept)."""
    import bz2
    rng = np.random.default_rng(len(sep) * 10 + len(comp))
    alphabet = list("ACGTacgtNRK")
    recs = [("".join(rng.choice(list("abcXYZ019"), size=10)), "".join(rng.choice(alphabet, size=int(rng.integers(35, 200)))))
            for _ in range(int(rng.integers(1, 10)))]
    opener = {"": open, ".gz": gzip.open, ".bz2": bz2.open}[comp]
    fa = tmp_path / f"x.fasta{comp}"
    with opener(fa, "wb") as f:                                # the reference's generator: the sequence on one line
        f.write("".join(f">{h}{sep}{s}{sep}" for h, s in recs).encode())
    got = [tuple(l.split("\t")) for l in run("parse", fa).rstrip("\n").split("\n")]
    assert got == recs
    fq = tmp_path / f"x.fastq{comp}"
    with opener(fq, "wb") as f:
        f.write(sep.join(f"@{h}{sep}{s}{sep}+{sep}{'I' * len(s)}" for h, s in recs).encode())
    got = [tuple(l.split("\t")) for l in run("parse", fq).rstrip("\n").split("\n")]
    assert got == recs == hostmodel.parse_fastq(sep.join(f"@{h}{sep}{s}{sep}+{sep}{'I' * len(s)}" for h, s in recs))


@pytest.mark.parametrize("chunk", [1, 2, 3, 5, 17, 4096])
def test_streaming_parser_is_chunk_border_proof(tmp_path, chunk):
    """The reader works on fixed-size chunks; records, line ends (incl. \\r\\n) and the 3-line FASTQ look-ahead may straddle any
    chunk border.  SLK_IO_CHUNK forces tiny chunks."""
    env = dict(os.environ, SLK_IO_CHUNK=str(chunk))

    def parse(*files):
        out = subprocess.run([CLI, "parse", *map(str, files)], check=True, capture_output=True, text=True, env=env).stdout
        return [tuple(l.split("\t")) for l in out.rstrip("\n").split("\n")] if out.strip() else []

    fa = tmp_path / "a.fasta"
    fa.write_text(FASTA)
    assert parse(fa) == hostmodel.parse_fasta(FASTA)
    for sep in ("\n", "\r\n", "\r"):
        text = FASTQ.replace("\n", sep)
        fq = tmp_path / "a.fastq"
        fq.write_bytes(text.encode())
        assert parse(fq) == hostmodel.parse_fastq(text)
    # paired: the first records line up, then the mate file is in another order and lacks one mate
    a, b = tmp_path / "x_1.fq", tmp_path / "x_2.fq"
    ids = [f"p{i}" for i in range(12)]
    a.write_text("".join(f"@{i}/1\n{'ACGT' * 3}{n}\n+\nIIII\n" for n, i in enumerate(ids)))
    order = ids[:4] + ids[4:][::-1]
    order.remove("p7")
    b.write_text("".join(f"@{i}/2\n{'TTGA' * 2}{ids.index(i)}\n+\nIIII\n" for i in order))
    got = parse(a, b)
    assert got == [(i, f"{'ACGT' * 3}{n}", f"{'TTGA' * 2}{n}") for n, i in enumerate(ids) if i != "p7"]


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_dynamic_taxon_selection_matches_restatement(tmp_path, seed):
    """`slacken-amd taxonomy`: Taxonomy.depth (nearest ranked ancestor-or-self), CountFilter and taxaWithDescendants of the C++
    host against tests/hostmodel.py, on random taxonomies with unranked nodes (Dynamic.scala:174-185, Taxonomy.scala:217-224,
    304-311; the reference's TaxonomyProps.scala checks the same invariants on its own implementation)."""
    rng = np.random.default_rng(seed)
    parents = taxgen.taxonomy(300, rng)
    tax = write_taxonomy(str(tmp_path / "tax"), parents, rng, ranked=0.6)
    defined = [t for t in range(1, len(parents)) if t == 1 or parents[t] != 0]
    counts = sorted({int(t): int(rng.integers(1, 60)) for t in rng.choice(defined, size=60, replace=False)}.items())
    with open(tmp_path / "c.tsv", "w") as f:
        for t, c in counts:
            f.write(f"{t}\t{c}\n")
    for rank in ("species", "genus", "phylum"):
        rd = hostmodel.RANKS.index(rank) - 1
        for threshold in (1, 25, 100):
            out = run("taxonomy", tmp_path / "tax", rank, threshold, tmp_path / "c.tsv").split("\n")
            keep = [int(x) for x in out[0].split()]
            full = [int(x) for x in out[1].split()]
            want = hostmodel.count_filter(tax, counts, rd, threshold)
            assert keep == want
            assert full == sorted(hostmodel.with_descendants(tax, want))
            # TaxonomyProps-style invariants: every kept taxon is in the closure; the closure is closed under children
            assert set(keep) <= set(full)
            assert all(c in set(full) for t in full for c in tax.children[t])


@pytest.mark.parametrize("seed", range(int(os.environ.get("SLK_PARSE_SEEDS", 6))))
def test_segment_parallel_parser_on_adversarial_text(tmp_path, seed):
    """Plain files are mapped and cut into segments parsed on several threads (seqio.hpp, PlainSegmentParser): on text made of
    the characters the two record rules look at ('@', '+', '>', the three line ends, ' '), with segment borders at every
    granularity, the records are those of the restatement and of the serial reader (the same bytes gzip-compressed)."""
    rng = np.random.default_rng(100 + seed)
    alphabet = np.array(list("@+>\n\r ACx"))
    weights = np.array([3, 3, 2, 6, 3, 1, 4, 4, 2], float)
    text = "".join(rng.choice(alphabet, size=int(rng.integers(1, 300)), p=weights / weights.sum()))
    # ... followed by lines that often start with '@' or '+', so that FASTQ windows match, overlap and nest
    for _ in range(int(rng.integers(5, 60))):
        body = "".join(rng.choice(list("AC@+> x"), size=int(rng.integers(0, 6))))
        text += str(rng.choice(["@", "+", "", ">"], p=[0.4, 0.35, 0.2, 0.05])) + body + str(rng.choice(["\n", "\r\n", "\r", "\n\n"]))
    if rng.integers(0, 2):
        text = text.rstrip("\r\n")          # no line end after the last line

    def parse(path, chunk, threads):
        env = dict(os.environ, SLK_IO_CHUNK=str(chunk), SLK_PARSE_THREADS=str(threads))
        out = subprocess.run([CLI, "parse", str(path)], check=True, capture_output=True, env=env).stdout.decode()
        return [tuple(l.split("\t")) for l in out.split("\n")[:-1]]

    for ext, model in (("fastq", hostmodel.parse_fastq), ("fasta", hostmodel.parse_fasta)):
        p = tmp_path / f"t.{ext}"
        p.write_bytes(text.encode())
        gz = tmp_path / f"t.{ext}.gz"
        with gzip.open(gz, "wb") as f:
            f.write(text.encode())
        want = [tuple(r) for r in model(text)]
        assert parse(gz, 5, 1) == want
        for chunk, threads in ((1, 3), (2, 2), (3, 4), (7, 1), (64, 5), (1 << 20, 2)):
            assert parse(p, chunk, threads) == want, (ext, chunk, threads)


def test_input_from_a_pipe(tmp_path):
    """A named pipe (or /dev/stdin, or a shell's <(...)) can neither be mapped nor peeked into: it takes the streaming reader,
    with the format still decided by the name (FileInputs.forFile)."""
    fifo = tmp_path / "reads.fq"
    os.mkfifo(fifo)
    text = FASTQ * 50
    import threading
    w = threading.Thread(target=lambda: open(fifo, "w").write(text))
    w.start()
    out = run("parse", fifo)
    w.join()
    got = [tuple(l.split("\t")) for l in out.rstrip("\n").split("\n")]
    assert got == hostmodel.parse_fastq(text)
