"""End to end through the C++ host (`slacken-amd classify`): a library in Slacken's on-disk layout (<loc>/ *.parquet,
<loc>.properties, <loc>_taxonomy/*.dmp; KeyValueIndex.scala:125-139, IndexParams.scala:92-101), FASTA / FASTQ(.gz) inputs,
per-read output files and Kraken reports in the reference's directory layout (Classifier.scala:184-227,415-420).  Expected
lines are the golden vectors (tests/golden, produced by the CPU oracle); the report is checked against hostmodel.py."""
import glob
import gzip
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import hostmodel
from test_host_cli import CLI, ROOT, write_taxonomy

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, os.path.join(ROOT, "tools"))


def make_library(tmp_path, convert=True):
    import parquet_to_slkrec as conv
    g = json.load(open(os.path.join(GOLD, "golden_classify.json")))
    lib = np.load(os.path.join(GOLD, "library.npz"))
    loc = str(tmp_path / "golden_lib")
    conv.write_parquet_dir(loc, lib["keys"], lib["taxa"], buckets=7)
    mask = int(np.uint64(0xe37e28c4271b5a2d).astype(np.int64))
    with open(loc + ".properties", "w") as f:
        f.write(f"#Properties for Slacken\n#Sun Oct 04 09:00:00 UTC 2026\nk={g['k']}\nm={g['m']}\nbuckets=7\nversion=1\n"
                f"splitter=randomXOR\nminimizerSpaces={g['spaces']}\nXORmask={mask}\ncanonical=true\n")
    tax = write_taxonomy(loc + "_taxonomy", lib["parents"], np.random.default_rng(3))
    if convert:   # (otherwise the CLI reads the Parquet files themselves)
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "parquet_to_slkrec.py"), loc])
    reads = [line.rstrip("\n").split("\t") for line in open(os.path.join(GOLD, "reads.tsv"))]
    return g, loc, tax, reads


def read_out(d, sample="all"):
    files = sorted(glob.glob(os.path.join(d, f"sample={sample}", "part-*.txt.gz")))
    assert files, d
    return [l for fn in files for l in gzip.open(fn, "rt").read().split("\n") if l]


def classify(*args):
    r = subprocess.run([CLI, "classify", *map(str, args)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return r


@pytest.mark.gpu
@pytest.mark.parametrize("convert", [True, False], ids=["slkrec", "parquet"])
def test_cli_single_reads_golden(tmp_path, convert):
    g, loc, tax, reads = make_library(tmp_path, convert)
    fa = tmp_path / "reads.fasta"
    with open(fa, "w") as f:        # multi-line FASTA, descriptions after the id
        for t, s in reads[:300]:
            f.write(f">{t} length={len(s)}\n{s[:60]}\n{s[60:]}\n")
    fq = tmp_path / "reads.fq.gz"
    with gzip.open(fq, "wt") as f:
        for t, s in reads[300:]:
            f.write(f"@{t} x\n{s}\n+\n{'I' * len(s)}\n")
    out = tmp_path / "out" / "gold"
    classify("-i", loc, "-o", out, "-c", "0.0", "0.15", "0.5", str(fa), str(fq))
    by_title = {r["title"]: r for r in g["reads"]}
    for thr, suffix in ((0.0, "0.00"), (0.15, "0.15"), (0.5, "0.50")):   # max decimals of the list = 2
        lines = read_out(f"{out}_c{suffix}")
        want = [r for r in g["reads"] if r["hits"]]
        assert [l.split("\t")[1] for l in lines] == [r["title"] for r in want]
        for l, r in zip(lines, want):
            cu, title, taxon, lens, detail = l.split("\t")
            assert (int(taxon), cu) == (r[f"c{thr}"][0], "C" if r[f"c{thr}"][1] else "U")
            assert [lens, detail] == r["line"].split("\t")[3:]
            if thr == 0.0:
                assert l == r["line"]
        counts = {}
        for r in want:
            counts[r[f"c{thr}"][0]] = counts.get(r[f"c{thr}"][0], 0) + 1
        rep = open(f"{out}_c{suffix}/all_kreport.txt").read().rstrip("\n").split("\n")
        assert rep == hostmodel.kraken_report(tax, sorted(counts.items()))[0]


@pytest.mark.gpu
def test_cli_paired_samples_and_flags(tmp_path):
    g, loc, tax, reads = make_library(tmp_path)
    f1, f2 = tmp_path / "s_1.fq", tmp_path / "s_2.fq"
    with open(f1, "w") as a, open(f2, "w") as b:
        order = list(range(0, 300, 2))
        for i in order:
            a.write(f"@{reads[i][0]}/1\n{reads[i][1]}\n+\n{'I' * len(reads[i][1])}\n")
        for i in reversed(order):    # the mate file in another order: pairing is by id, not by position
            b.write(f"@{reads[i][0]}/2\n{reads[i + 1][1]}\n+\n{'I' * len(reads[i + 1][1])}\n")
    out = tmp_path / "paired"
    classify("-i", loc, "-o", out, "-p", f1, f2)
    lines = read_out(f"{out}_c0.0")
    assert lines == [p["line"] for p in g["pairs"] if p["line"]]
    # --nounclassified drops U rows from both the lines and the report; --nodetailed writes reports only
    out2 = tmp_path / "paired2"
    classify("-i", loc, "-o", out2, "-p", "--nounclassified", f1, f2)
    kept = read_out(f"{out2}_c0.0")
    assert kept == [l for l in lines if l.startswith("C")] and 0 < len(kept) < len(lines)
    rep2 = open(f"{out2}_c0.0/all_kreport.txt").read()
    assert "unclassified" not in rep2
    out3 = tmp_path / "paired3"
    classify("-i", loc, "-o", out3, "-p", "--nodetailed", f1, f2)
    assert not glob.glob(f"{out3}_c0.0/sample=*")
    assert open(f"{out3}_c0.0/all_kreport.txt").read() == open(f"{out}_c0.0/all_kreport.txt").read()
    # SLK_CLI_PACKED=1: the reports-only calls send the reads at 3 bits per base (slk_classify_batch_packed, pairs too): same report
    out3p = tmp_path / "paired3_packed"
    r = subprocess.run([CLI, "classify", "-i", loc, "-o", str(out3p), "-p", "--nodetailed", str(f1), str(f2)], capture_output=True, text=True,
                       env=dict(os.environ, SLK_CLI_PACKED="1"))
    assert r.returncode == 0, r.stderr
    assert open(f"{out3p}_c0.0/all_kreport.txt").read() == open(f"{out}_c0.0/all_kreport.txt").read()
    # --sample-regex: group 1 of the first match is the sample id (Classifier.scala:138-142); unmatched -> "other"
    out4 = tmp_path / "multi"
    classify("-i", loc, "-o", out4, "--sample-regex", r"sra\.(\d)", "-p", f1, f2)
    samples = sorted(os.path.basename(d)[len("sample="):] for d in glob.glob(f"{out4}_c0.0/sample=*"))
    assert len(samples) > 1 and all(len(s) == 1 and s.isdigit() for s in samples)
    merged = sorted(l for s in samples for l in read_out(f"{out4}_c0.0", s))
    assert merged == sorted(lines)
    for s in samples:
        assert all(l.split("\t")[1].split("sra.")[1][0] == s for l in read_out(f"{out4}_c0.0", s))
        assert os.path.exists(f"{out4}_c0.0/{s}_kreport.txt")
    # A match in which group 1 takes no part: Match.group(1) is null in the reference, and a null (or empty) partition value is
    # Spark's sample=__HIVE_DEFAULT_PARTITION__, from whose name the report's is read back (Classifier.scala:138-142,207-211,232-241);
    # characters a path cannot hold are %-escaped the way Spark's partition writer escapes them.  Reports only: regexp_extract's "".
    out5 = tmp_path / "nullgroup"
    classify("-i", loc, "-o", out5, "--sample-regex", r"(?:sra\.([0-4]))?\d", "-p", f1, f2)   # group 1 only for ids sra.0 .. sra.4
    samples5 = sorted(os.path.basename(d)[len("sample="):] for d in glob.glob(f"{out5}_c0.0/sample=*"))
    assert "__HIVE_DEFAULT_PARTITION__" in samples5 and set(samples5) - {"__HIVE_DEFAULT_PARTITION__"} <= set("01234")
    assert os.path.exists(f"{out5}_c0.0/__HIVE_DEFAULT_PARTITION___kreport.txt")
    assert sorted(l for s in samples5 for l in read_out(f"{out5}_c0.0", s)) == sorted(lines)
    out6 = tmp_path / "escaped"
    classify("-i", loc, "-o", out6, "--sample-regex", r"(a\.\d)", "-p", f1, f2)            # "a.3" needs no escape ...
    classify("-i", loc, "-o", tmp_path / "escaped2", "--sample-regex", r"(r)(a)", "-p", f1, f2)
    assert glob.glob(f"{out6}_c0.0/sample=a.*")
    out7 = tmp_path / "nullgroup_reports"
    classify("-i", loc, "-o", out7, "--nodetailed", "--sample-regex", r"(?:sra\.([0-4]))?\d", "-p", f1, f2)
    assert os.path.exists(f"{out7}_c0.0/_kreport.txt") and not os.path.exists(f"{out7}_c0.0/__HIVE_DEFAULT_PARTITION___kreport.txt")


@pytest.mark.parametrize("m", [31, 40, 100])
def test_native_parquet_reader_matches_converter(tmp_path, m):
    """`slacken-amd records`: the C++ Parquet reader (Arrow C++ from the pyarrow wheel) and the flat file written by the
    converter hold the same records; snappy-compressed bucket files as Spark writes them, several row groups; one id column
    per 32 nt of minimizer (id1..idN, KeyValueIndex.scala:49)."""
    import pyarrow as pa
    import pyarrow.parquet as pq
    rng = np.random.default_rng(2)
    n, W = 50_000, (m + 31) // 32
    keys = rng.integers(-2**62, 2**62, (n, W)).astype(np.int64)
    taxa = rng.integers(1, 3_000_000, n).astype(np.int32)
    loc = str(tmp_path / "lib")
    os.makedirs(loc)
    with open(loc + ".properties", "w") as f:
        f.write(f"k={m + 4}\nm={m}\nversion=1\nsplitter=randomXOR\n")
    for b in range(4):
        sel = np.arange(n) % 4 == b
        cols = {f"id{i + 1}": pa.array(keys[sel, i], pa.int64()) for i in range(W)}
        cols["taxon"] = pa.array(taxa[sel], pa.int32())
        pq.write_table(pa.table(cols), os.path.join(loc, f"part-00000-x_{b:05d}.c000.snappy.parquet"), compression="snappy",
                       row_group_size=3000)
    open(os.path.join(loc, "_SUCCESS"), "w").close()
    import parquet_to_slkrec as conv
    conv.convert(loc)
    k2, t2 = conv.read_parquet_dir(loc)
    assert np.array_equal(np.sort(t2), np.sort(taxa)) and k2.reshape(n, -1).shape == (n, W)
    out = subprocess.run([CLI, "records", loc], check=True, capture_output=True, text=True).stdout.strip().split("\n")
    if len(out) == 1:
        pytest.skip("CLI built without Parquet support")
    assert out[0].split(" ", 1)[1] == out[1].split(" ", 1)[1]
    assert f"n={n} " in out[0] and f"max_taxon={int(taxa.max())}" in out[0]
    assert f"taxon_sum={int(taxa.astype(np.int64).sum())}" in out[0]


def test_parquet_roundtrip(tmp_path):
    import parquet_to_slkrec as conv
    rng = np.random.default_rng(1)
    keys = rng.integers(-2**62, 2**62, 1000).astype(np.int64)
    taxa = rng.integers(1, 1000, 1000).astype(np.int32)
    loc = str(tmp_path / "lib")
    conv.write_parquet_dir(loc, keys, taxa, buckets=5)
    k2, t2 = conv.read_parquet_dir(loc)
    assert sorted(zip(k2.tolist(), t2.tolist())) == sorted(zip(keys.tolist(), taxa.tolist()))
    conv.write_slkrec(loc + ".slkrec", k2, t2)
    raw = open(loc + ".slkrec", "rb").read()
    assert raw[:8] == b"SLKREC1\0" and len(raw) == 24 + 12 * 1000
    assert np.array_equal(np.frombuffer(raw, np.int64, 1000, 24), k2)
    # the streaming converter (what the command line runs) writes the same file, whatever the batch size
    assert conv.convert(loc, batch_rows=77) == 1000
    assert open(loc + ".slkrec", "rb").read() == raw


@pytest.mark.gpu
def test_cli_many_reads_cross_slice_and_thread_borders(tmp_path, orc):
    """70 000 reads: several formatting slices (32 768 reads each) on several threads, gzip members appended in read order;
    every line equals the oracle's and the order is the input order."""
    import synth
    g, loc, tax, _ = make_library(tmp_path, convert=False)
    lib = np.load(os.path.join(GOLD, "library.npz"))
    p = orc.params(k=g["k"], m=g["m"], spaces=g["spaces"])
    oix = orc.Index(1, lib["keys"], lib["taxa"])
    base_reads = [line.rstrip("\n").split("\t")[1] for line in open(os.path.join(GOLD, "reads.tsv"))][:300]
    rng = np.random.default_rng(8)
    n = 70_000
    pick = rng.integers(0, len(base_reads), n)
    cut = rng.integers(40, 101, n)
    fq = tmp_path / "many.fq"
    with open(fq, "w") as f:
        for i in range(n):
            s = base_reads[pick[i]][:cut[i]]
            f.write(f"@q{i}\n{s}\n+\n{'I' * len(s)}\n")
    out = tmp_path / "many"
    classify("-i", loc, "-o", out, "-c", "0.05", fq)
    lines = read_out(f"{out}_c0.05")
    cache = {}
    want = []
    for i in range(n):
        key = (int(pick[i]), int(cut[i]))
        if key not in cache:
            res, hits = orc.classify_read(p, oix, lib["parents"], base_reads[key[0]][:key[1]], None, 2, 0.05)
            cache[key] = (res, hits)
        res, hits = cache[key]
        if hits:
            want.append(orc.output_line(res["classified"], f"q{i}", res["taxon"], hits, g["k"]))
    assert len(lines) == len(want)
    assert lines == want


@pytest.mark.gpu
def test_cli_several_input_files_side_by_side(tmp_path):
    """Several input files are read concurrently and their batches taken in turn: every read is classified exactly once, the
    output is the same on every run, and with SLK_INPUT_STREAMS=1 it is in file order."""
    g, loc, tax, reads = make_library(tmp_path, convert=False)
    files = []
    for i in range(5):
        fq = tmp_path / f"lane{i}.fq.gz"
        with gzip.open(fq, "wt") as f:
            for t, s in reads[i::5]:
                f.write(f"@{t}\n{s}\n+\n{'I' * len(s)}\n")
        files.append(str(fq))
    want = {r["title"]: r["line"] for r in g["reads"] if r["hits"]}
    outs = []
    for run_i, env in enumerate(({}, {}, {"SLK_INPUT_STREAMS": "1"})):
        out = tmp_path / f"multi{run_i}"
        r = subprocess.run([CLI, "classify", "-i", loc, "-o", str(out), *files], capture_output=True, text=True,
                           env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr
        outs.append(read_out(f"{out}_c0.0"))
    assert outs[0] == outs[1]
    assert sorted(outs[0]) == sorted(want.values()) == sorted(outs[2])
    in_file_order = [want[t] for i in range(5) for t, _ in reads[i::5] if t in want]
    assert outs[2] == in_file_order


@pytest.mark.gpu
def test_cli_long_reads_all_routes_agree(tmp_path, orc):
    """Long reads (the golden reads chained to 1.5-40 kbp, some with Ns) through the CLI: with per-read lines the deferred
    fragments take the wave kernel (hit lists), with --nodetailed the segment kernel; each line equals the oracle's and the two
    reports are the same file."""
    g, loc, tax, reads = make_library(tmp_path, convert=False)
    lib = np.load(os.path.join(GOLD, "library.npz"))
    p = orc.params(k=g["k"], m=g["m"], spaces=g["spaces"])
    oix = orc.Index(1, lib["keys"], lib["taxa"])
    rng = np.random.default_rng(21)
    seqs = [r[1] for r in reads]
    fq = tmp_path / "long.fq"
    longs = []
    with open(fq, "w") as f:
        for i in range(120):
            want_len = int(rng.choice([1500, 3000, 4500, 8000, 40000]))
            parts, have = [], 0
            while have < want_len:
                s = seqs[rng.integers(0, len(seqs))]
                if rng.random() < 0.1:
                    s = "N" * int(rng.choice([1, 10, 35, 80]))
                parts.append(s)
                have += len(s)
            s = "".join(parts)
            longs.append(s)
            f.write(f"@long{i}\n{s}\n+\n{'I' * len(s)}\n")
        for i in range(200):   # and short ones around them
            s = seqs[rng.integers(0, len(seqs))]
            longs.append(s)
            f.write(f"@short{i}\n{s}\n+\n{'I' * len(s)}\n")
    out = tmp_path / "long"
    classify("-i", loc, "-o", out, "-c", "0.1", fq)
    lines = read_out(f"{out}_c0.1")
    want = []
    for i, s in enumerate(longs):
        res, hits = orc.classify_read(p, oix, lib["parents"], s, None, 2, 0.1)
        if hits:
            want.append(orc.output_line(res["classified"], f"long{i}" if i < 120 else f"short{i - 120}", res["taxon"], hits, g["k"]))
    assert lines == want
    out2 = tmp_path / "long_reports"
    classify("-i", loc, "-o", out2, "-c", "0.1", "--nodetailed", fq)
    assert open(f"{out2}_c0.1/all_kreport.txt").read() == open(f"{out}_c0.1/all_kreport.txt").read()


@pytest.mark.gpu
@pytest.mark.parametrize("env", [dict(SLK_CLASSIFY_THREADS="1", SLK_PARSE_THREADS="1"),
                                 dict(SLK_CLASSIFY_THREADS="6", SLK_PARSE_THREADS="8", SLK_IO_CHUNK="4096", SLK_HOST_THREADS="3"),
                                 dict(SLK_CLASSIFY_THREADS="3", SLK_IO_CHUNK="777", SLK_GZIP_IMPL="zlib", SLK_GZIP_LEVEL="1"),
                                 dict(SLK_CLI_MERGED_HITS="0")],
                         ids=["serial", "many-small-batches", "tiny-segments-zlib", "un-merged-hit-lists"])
def test_cli_output_does_not_depend_on_the_host_pipeline_shape(tmp_path, env):
    """Segment size of the parallel parser, number of parsing / classifying / formatting threads, the gzip implementation, hit lists
    merged on the device (the default of the first pass) or by the formatter: the per-read lines (in input order) and the report are
    the same as with the defaults."""
    g, loc, tax, reads = make_library(tmp_path, convert=False)
    rng = np.random.default_rng(5)
    fq = tmp_path / "r.fq"
    with open(fq, "w") as f:
        for i in range(6000):
            t, s = reads[rng.integers(0, len(reads))]
            s = s[:int(rng.integers(min(40, len(s)), len(s) + 1))]
            f.write(f"@{t}.{i}\n{s}\n+\n{'I' * len(s)}\n")
    ref_out = tmp_path / "ref"
    classify("-i", loc, "-o", ref_out, "-c", "0.1", fq)
    out = tmp_path / "alt"
    r = subprocess.run([CLI, "classify", "-i", loc, "-o", str(out), "-c", "0.1", str(fq)], capture_output=True, text=True,
                       env=dict(os.environ, **env))
    assert r.returncode == 0, r.stderr
    assert read_out(f"{out}_c0.1") == read_out(f"{ref_out}_c0.1")
    assert open(f"{out}_c0.1/all_kreport.txt").read() == open(f"{ref_out}_c0.1/all_kreport.txt").read()


# ---- titles that occur more than once: Classifier.scala:92 (groupBy seqTitle), :136 (sort by ordinal); the paired reader's
# inner join on the header, InputReader.scala:104-119.  Expected rows come from the restatement in hostmodel.py
# (merge_by_title, paired_join) over the oracle's per-fragment hit lists and its Classifier.classify on the merged list.
# The reference holds no fixture for this: parity unpinned (restatement vs engine).
def _oracle_env():
    from oracle import oracle
    g = json.load(open(os.path.join(GOLD, "golden_classify.json")))
    lib = np.load(os.path.join(GOLD, "library.npz"))
    p = oracle.params(k=g["k"], m=g["m"], spaces=g["spaces"])
    return oracle, g, p, oracle.Index(1, lib["keys"], lib["taxa"]), lib["parents"]


def _expected_rows(fragments, thresholds=(0.0,), min_hit_groups=2):
    """fragments: [(title, seq1, seq2 or None)] in input order -> {threshold: [output lines]} in order of first appearance."""
    oracle, g, p, oix, parents = _oracle_env()
    per = []
    for title, s1, s2 in fragments:
        _, hits = oracle.classify_read(p, oix, parents, s1, s2)
        sp = oracle.spans(p, s1, s2)
        per.append((title, hits, [int(x["distinct"]) for x in sp]))
        assert len(sp) == len(hits)
    out = {}
    for thr in thresholds:
        rows = []
        for title, hits, distinct in hostmodel.merge_by_title(per):
            r = oracle.classify_hits(parents, hits, distinct, min_hit_groups, thr)
            rows.append(oracle.output_line(r["classified"], title, r["taxon"], hits, g["k"]))
        out[thr] = rows
    return out


def _report_of(tax, lines):
    counts = {}
    for l in lines:
        t = int(l.split("\t")[2])
        counts[t] = counts.get(t, 0) + 1
    return hostmodel.kraken_report(tax, sorted(counts.items()))[0]


@pytest.mark.gpu
def test_cli_repeated_titles_unpaired(tmp_path):
    """Both mate files given WITHOUT -p: every id occurs twice (the realistic way to get repeated titles).  The reference makes
    one row per id from the hits of both fragments; so must the CLI, in the lines, in the reports and with --nodetailed."""
    g, loc, tax, reads = make_library(tmp_path)
    f1, f2 = tmp_path / "s_1.fq", tmp_path / "s_2.fa"
    ids = list(range(0, 240, 2))
    with open(f1, "w") as a:
        for i in ids:
            a.write(f"@{reads[i][0]} first\n{reads[i][1]}\n+\n{'I' * len(reads[i][1])}\n")
        a.write(f"@{reads[300][0]}\n{reads[300][1]}\n+\n{'I' * len(reads[300][1])}\n")       # occurs once
        a.write(f"@{reads[0][0]}\n{reads[301][1]}\n+\n{'I' * len(reads[301][1])}\n")         # a third fragment of id 0
        a.write(f"@short\nACGT\n+\nIIII\n")                                                  # no span: no row
    with open(f2, "w") as b:
        for i in reversed(ids):
            b.write(f">{reads[i][0]} second\n{reads[i + 1][1]}\n")
        b.write(">short\nACGTACGT\n")                                                        # still no span
    frags = [(reads[i][0], reads[i][1], None) for i in ids] + [(reads[300][0], reads[300][1], None), (reads[0][0], reads[301][1], None),
                                                             ("short", "ACGT", None)]
    frags += [(reads[i][0], reads[i + 1][1], None) for i in reversed(ids)] + [("short", "ACGTACGT", None)]
    want = _expected_rows(frags, thresholds=(0.0, 0.5))
    out = tmp_path / "rep"
    r = classify("-i", loc, "-o", out, "-c", "0.0", "0.5", f1, f2)
    assert "occur more than once" in r.stderr
    for thr, suffix in ((0.0, "0.0"), (0.5, "0.5")):
        lines = read_out(f"{out}_c{suffix}")
        assert sorted(lines) == sorted(want[thr])          # (row order is not part of the contract: Spark's partition order)
        assert len(lines) == len(ids) + 1
        rep = open(f"{out}_c{suffix}/all_kreport.txt").read().rstrip("\n").split("\n")
        assert rep == _report_of(tax, want[thr])
    out2 = tmp_path / "rep_nd"
    classify("-i", loc, "-o", out2, "-c", "0.0", "0.5", "--nodetailed", f1, f2)
    for suffix in ("0.0", "0.5"):
        assert open(f"{out2}_c{suffix}/all_kreport.txt").read() == open(f"{out}_c{suffix}/all_kreport.txt").read()
    out3 = tmp_path / "rep_nu"
    classify("-i", loc, "-o", out3, "--nounclassified", f1, f2)
    assert sorted(read_out(f"{out3}_c0.0")) == sorted(l for l in want[0.0] if l.startswith("C"))
    # a merged row differs from its fragments' own rows: the regrouping is not a no-op on this input
    solo = _expected_rows(frags[:3])[0.0]
    assert not set(solo) <= set(want[0.0])


@pytest.mark.gpu
@pytest.mark.parametrize("shuffled", [False, True], ids=["lockstep", "shuffled"])
def test_cli_repeated_headers_paired(tmp_path, shuffled):
    """A header that repeats inside a file of a pair: the reference's join pairs every record of file 1 with every record of
    file 2 of that header (2 x 2, 2 x 1, 1 x 2 below), and the title grouping then merges the products into one row."""
    g, loc, tax, reads = make_library(tmp_path)
    t = [r[0] for r in reads]
    s = [r[1] for r in reads]
    rec1 = [(t[0], s[0]), (t[2], s[2]), (t[4], s[4]), (t[2], s[6]), (t[8], s[8]), (t[10], s[10]), (t[8], s[12]), (t[14], s[14])]
    rec2 = [(t[0], s[1]), (t[2], s[3]), (t[4], s[5]), (t[2], s[7]), (t[8], s[9]), (t[10], s[11]), (t[10], s[13]), (t[14], s[15])]
    if shuffled:
        rec2 = rec2[::-1]
    f1, f2 = tmp_path / "r_1.fq", tmp_path / "r_2.fq"
    with open(f1, "w") as a:
        for h, q in rec1:
            a.write(f"@{h}/1\n{q}\n+\n{'I' * len(q)}\n")
    with open(f2, "w") as b:
        for h, q in rec2:
            b.write(f"@{h}/2\n{q}\n+\n{'I' * len(q)}\n")
    frags = hostmodel.paired_join([(h + "/1", q) for h, q in rec1], [(h + "/2", q) for h, q in rec2])
    assert len(frags) == 3 + 4 + 2 + 2
    want = _expected_rows(frags)[0.0]
    out = tmp_path / "rp"
    classify("-i", loc, "-o", out, "-p", f1, f2)
    lines = read_out(f"{out}_c0.0")
    assert sorted(lines) == sorted(want) and len(lines) == 6
    assert open(f"{out}_c0.0/all_kreport.txt").read().rstrip("\n").split("\n") == _report_of(tax, want)
    out2 = tmp_path / "rp_nd"
    classify("-i", loc, "-o", out2, "-p", "--nodetailed", f1, f2)
    assert open(f"{out2}_c0.0/all_kreport.txt").read() == open(f"{out}_c0.0/all_kreport.txt").read()


@pytest.mark.gpu
def test_cli_devices_share_the_reads(tmp_path):
    """--devices: the table is replicated and the reads are shared out batch by batch.  The output does not depend on
    the device list -- byte for byte.  (On a one-GPU box the same device is listed twice: two tables, two sets of streams.)"""
    g, loc, tax, reads = make_library(tmp_path)
    fq = tmp_path / "reads.fq"
    with open(fq, "w") as f:
        for rep in range(40):     # enough batches for every worker to get some
            for t, s in reads:
                f.write(f"@{t}.{rep}\n{s}\n+\n{'I' * len(s)}\n")
    env = dict(os.environ, SLK_IO_CHUNK=str(1 << 18))   # small segments => many batches
    outs = []
    for name, devs in (("one", "0"), ("two", "0,0")):
        out = tmp_path / name
        r = subprocess.run([CLI, "classify", "-i", loc, "-o", str(out), "-c", "0.0", "0.15", "--devices", devs, str(fq)],
                           capture_output=True, text=True, env=dict(env, SLK_HOST_TIMING="1"))
        assert r.returncode == 0, r.stderr
        assert f"over {len(devs.split(','))} device table(s)" in r.stderr
        outs.append(out)
    for suffix in ("0.00", "0.15"):
        a, b = (gzip.open(f"{o}_c{suffix}/sample=all/part-00000.txt.gz", "rb").read() for o in outs)
        assert a == b and a.count(b"\n") >= 40 * 500
        assert open(f"{outs[0]}_c{suffix}/all_kreport.txt").read() == open(f"{outs[1]}_c{suffix}/all_kreport.txt").read()
    bad = subprocess.run([CLI, "classify", "-i", loc, "-o", str(tmp_path / "x"), "--devices", "0,99", str(fq)], capture_output=True, text=True)
    assert bad.returncode != 0 and "out of range" in bad.stderr


@pytest.mark.gpu
def test_cli_sample_whose_only_row_is_merged_away(tmp_path):
    """A sample whose only read is a title that occurs twice: one fragment classifies on its own, the other is all misses; merged
    (Classifier.scala:92) the read is unclassified -- its k-mer total has grown, its clade has not.  With --nounclassified the
    reference would write nothing for that sample: no part file (certainly not a 0-byte .gz, which is no gzip file), no
    directory, no report; with unclassified rows on, the one merged U row."""
    g, loc, tax, reads = make_library(tmp_path)
    rng = np.random.default_rng(8)
    fq = tmp_path / "reads.fq"
    probe = tmp_path / "probe.fq"
    with open(probe, "w") as f:
        for t, s in reads[:200]:
            f.write(f"@s1_{t}\n{s}\n+\n{'I' * len(s)}\n")
    classify("-i", loc, "-o", tmp_path / "probe", "-c", "0.5", "--sample-regex", "^(s[0-9]+)_", probe)
    first_c = next(l.split("\t")[1] for l in read_out(f"{tmp_path / 'probe'}_c0.5", "s1") if l.startswith("C"))
    seq_c = dict((f"s1_{t}", s) for t, s in reads[:200])[first_c]
    noise = "".join("ACGT"[i] for i in rng.integers(0, 4, 600))
    with open(fq, "w") as f:
        for t, s in reads[:200]:
            f.write(f"@s1_{t}\n{s}\n+\n{'I' * len(s)}\n")
        f.write(f"@s7_dup\n{seq_c}\n+\n{'I' * len(seq_c)}\n")
        f.write(f"@s7_dup\n{noise}\n+\n{'I' * len(noise)}\n")
    off, on = tmp_path / "off", tmp_path / "on"
    classify("-i", loc, "-o", off, "-c", "0.5", "--sample-regex", "^(s[0-9]+)_", "--nounclassified", fq)
    classify("-i", loc, "-o", on, "-c", "0.5", "--sample-regex", "^(s[0-9]+)_", fq)
    assert not os.path.exists(f"{off}_c0.5/sample=s7") and not os.path.exists(f"{off}_c0.5/s7_kreport.txt")
    assert os.path.exists(f"{off}_c0.5/sample=s1/part-00000.txt.gz") and not glob.glob(f"{off}_c0.5/sample=*/*.tmp")
    rows = read_out(f"{on}_c0.5", "s7")
    assert len(rows) == 1 and rows[0].startswith("U\ts7_dup\t0\t")
    assert read_out(f"{on}_c0.5", "s1") and all(l.startswith("C") for l in read_out(f"{off}_c0.5", "s1"))


@pytest.mark.gpu
def test_cli_shard_table_gives_the_same_files(tmp_path):
    """--shard-table: the library is SPREAD over the devices of --devices (each keeps the records whose minimizer falls to it) and
    the batches are classified in rounds through slk_shardset_classify -- minimizers to their owners, taxa back.  Per-read files
    and reports are byte for byte those of the replicated mode: single-end with long reads among them, pairs, a title that occurs
    twice (its merged row is classified from hit lists that come through the sharded route), reports only.  (One GPU: the device is
    listed two or three times, the exchange is device-to-device copies; the RCCL leg needs a device per table.)"""
    g, loc, tax, reads = make_library(tmp_path)
    rng = np.random.default_rng(3)
    fq = tmp_path / "reads.fq"
    with open(fq, "w") as f:
        for rep in range(12):
            for t, s in reads:
                f.write(f"@{t}.{rep}\n{s}\n+\n{'I' * len(s)}\n")
        long_read = "".join(s for _, s in reads[:40])           # > 1000 bases: the staged round
        f.write(f"@long\n{long_read}\n+\n{'I' * len(long_read)}\n")
        t0, s0 = reads[0]
        f.write(f"@{t0}.0\n{reads[1][1]}\n+\n{'I' * len(reads[1][1])}\n")      # a title for the second time
    f1, f2 = tmp_path / "r1.fq", tmp_path / "r2.fq"
    with open(f1, "w") as a, open(f2, "w") as b:
        for i in range(0, len(reads) - 1, 2):
            a.write(f"@p{i}/1\n{reads[i][1]}\n+\n{'I' * len(reads[i][1])}\n")
            b.write(f"@p{i}/2\n{reads[i + 1][1]}\n+\n{'I' * len(reads[i + 1][1])}\n")
    env = dict(os.environ, SLK_IO_CHUNK=str(1 << 17))   # small segments => many batches => several rounds
    runs = {}
    for name, extra in (("rep", []), ("sh2", ["--devices", "0,0", "--shard-table"]), ("sh3", ["--devices", "0,0,0", "--shard-table"])):
        for kind, args in (("se", [str(fq)]), ("pe", ["-p", str(f1), str(f2)]), ("nd", ["--nodetailed", str(fq)])):
            out = tmp_path / f"{name}_{kind}"
            r = subprocess.run([CLI, "classify", "-i", loc, "-o", str(out), "-c", "0.0", "0.15", *extra, *args], capture_output=True, text=True, env=env)
            assert r.returncode == 0, r.stderr
            if extra:
                assert f"table sharded over {len(extra[1].split(','))} device table(s), exchange by device-to-device copies" in r.stderr
            runs[(name, kind)] = out
    for kind in ("se", "pe", "nd"):
        for suffix in ("0.00", "0.15"):
            want_report = open(f"{runs[('rep', kind)]}_c{suffix}/all_kreport.txt").read()
            for name in ("sh2", "sh3"):
                assert open(f"{runs[(name, kind)]}_c{suffix}/all_kreport.txt").read() == want_report, (name, kind, suffix)
                if kind != "nd":
                    a, b = (gzip.open(f"{runs[(n, kind)]}_c{suffix}/sample=all/part-00000.txt.gz", "rb").read() for n in ("rep", name))
                    assert a == b and a.count(b"\n") > 100, (name, kind, suffix)
    bad = subprocess.run([CLI, "classify2", "-i", loc, "-o", str(tmp_path / "x"), "--library", str(tmp_path), "--shard-table", str(fq)],
                         capture_output=True, text=True)
    assert bad.returncode != 0 and "--shard-table is for classify" in bad.stderr
    # Batches that are all long reads (the send lists used to be sized from every base of a batch, long fragments
    # included, against a hard limit of 2^25 entries per list -- a batch of 512 MB of long reads aborted; the regions are now sized
    # below 2^32 entries and the long fragments emit nothing).  40 Mbp of 25-50 kbp reads among short ones, one big batch.
    lf = tmp_path / "long.fq"
    genome = "".join(s for _, s in reads)
    with open(lf, "w") as f:
        n = 0
        while n < 40_000_000:
            L = int(rng.integers(25_000, 50_000))
            reps = -(-L // len(genome))
            s = (genome * reps)[:L]
            f.write(f"@L{n}\n{s}\n+\n{'I' * L}\n")
            n += L
        for t, s in reads[:50]:
            f.write(f"@{t}.s\n{s}\n+\n{'I' * len(s)}\n")
    outs = {}
    for name, extra in (("rep", []), ("sh2", ["--devices", "0,0", "--shard-table"])):
        out = tmp_path / f"long_{name}"
        r = subprocess.run([CLI, "classify", "-i", loc, "-o", str(out), *extra, str(lf)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[name] = out
    a, b = (gzip.open(f"{outs[n]}_c0.0/sample=all/part-00000.txt.gz", "rb").read() for n in ("rep", "sh2"))
    assert a == b and a.count(b"\n") > 800


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(int(os.environ.get("SLK_TITLE_SEEDS", 4))))
def test_cli_repeated_titles_randomised(tmp_path, seed):
    """Random multiplicities of every title -- 0 to 3 records per file, any order, unpaired over two files or paired with the
    join's products -- against the restatement (hostmodel.merge_by_title / paired_join over the oracle's hit lists)."""
    import random
    g, loc, tax, reads = make_library(tmp_path)
    rnd = random.Random(1000 + seed)
    paired = seed % 2 == 1
    titles = [f"t{i}" for i in range(120)]
    pick = lambda: reads[rnd.randrange(len(reads))][1]
    r1 = [(t, pick()) for t in titles for _ in range(rnd.choice([0, 1, 1, 1, 1, 2, 3]))]
    r2 = [(t, pick()) for t in titles for _ in range(rnd.choice([0, 1, 1, 1, 1, 2, 3]))]
    if rnd.random() < 0.7:
        rnd.shuffle(r2)
    if rnd.random() < 0.3:
        rnd.shuffle(r1)
    f1, f2 = tmp_path / "x_1.fq", tmp_path / "x_2.fq"
    for f, recs, suf in ((f1, r1, "/1"), (f2, r2, "/2")):
        with open(f, "w") as out:
            for t, q in recs:
                out.write(f"@{t}{suf if paired else ''}\n{q}\n+\n{'I' * len(q)}\n")
    if paired:
        frags = hostmodel.paired_join([(t + "/1", q) for t, q in r1], [(t + "/2", q) for t, q in r2])
    else:
        frags = [(t, q, None) for t, q in r1 + r2]
    want = _expected_rows(frags, thresholds=(0.0, 0.3))
    out = tmp_path / "rnd"
    classify("-i", loc, "-o", out, "-c", "0.0", "0.3", *(["-p"] if paired else []), f1, f2)
    for thr, suffix in ((0.0, "0.0"), (0.3, "0.3")):
        lines = read_out(f"{out}_c{suffix}") if want[thr] else []
        assert sorted(lines) == sorted(want[thr]), (seed, thr)
        if want[thr]:
            assert open(f"{out}_c{suffix}/all_kreport.txt").read().rstrip("\n").split("\n") == _report_of(tax, want[thr])
    out2 = tmp_path / "rnd_nd"
    classify("-i", loc, "-o", out2, "-c", "0.0", "0.3", "--nodetailed", *(["-p"] if paired else []), f1, f2)
    for suffix in ("0.0", "0.3"):
        assert open(f"{out2}_c{suffix}/all_kreport.txt").read() == open(f"{out}_c{suffix}/all_kreport.txt").read()
