"""`slacken-amd classify2` (two-step classification with a dynamic library, BASELINE config 5) end to end against a Python
restatement of Dynamic.scala driven by the CPU oracle: detected taxon set, the records of the dynamic library (through the
classifications they produce) and the per-read output.  Library layout as the reference expects it: <lib>/library/**/*.fna
and <lib>/seqid2taxid.map (Slacken.scala:116-121)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import hostmodel
import synth
import taxgen
from test_host_cli import CLI, ROOT
from test_host_classify_gpu import read_out

sys.path.insert(0, os.path.join(ROOT, "tools"))
pytestmark = pytest.mark.gpu


def write_ranked_taxonomy(d, parents):
    """level L of taxgen's generator gets the L-th standard rank (superkingdom .. species); a few nodes are 'no rank'"""
    os.makedirs(d, exist_ok=True)
    ls = (len(parents) - 2) // taxgen.N_RANKS
    nodes, names = [(1, 1, "no rank")], [(1, "root")]
    for t in range(2, len(parents)):
        level = (t - 2) // ls + 1
        rank = hostmodel.RANKS[level + 1] if t % 7 else "no rank"
        nodes.append((t, int(parents[t]), rank))
        names.append((t, f"Taxon {t}"))
    with open(os.path.join(d, "nodes.dmp"), "w") as f:
        for t, p, r in nodes:
            f.write(f"{t}\t|\t{p}\t|\t{r}\t|\n")
    with open(os.path.join(d, "names.dmp"), "w") as f:
        for t, nm in names:
            f.write(f"{t}\t|\t{nm}\t|\t\t|\tscientific name\t|\n")
    return hostmodel.Taxonomy(nodes, names)


def setup(tmp_path, orc, seed=9):
    import parquet_to_slkrec as conv
    rng = np.random.default_rng(seed)
    p = orc.params()
    parents = taxgen.taxonomy(8 * 12, rng)
    ls = (len(parents) - 2) // taxgen.N_RANKS
    species = list(range(7 * ls + 2, 8 * ls + 2))
    genus = list(range(6 * ls + 2, 7 * ls + 2))
    g_taxa = [species[i] for i in (0, 1, 2, 3, 5, 8)] + [genus[1], genus[2]]
    genomes = [synth.random_dna(30000, rng) for _ in g_taxa]
    for g in range(1, len(genomes)):          # shared stretches => LCA records
        genomes[g][1000:2500] = genomes[g - 1][1000:2500]
    # each genome is three sequences with their own ids; sequences carry Ns and line breaks in the files
    seq_ids, seqs, seq_taxa = [], [], []
    for gi, (g, t) in enumerate(zip(genomes, g_taxa)):
        for part, (a, b) in enumerate(((0, 12000), (12000, 12020), (12020, 30000))):
            s = g[a:b].tobytes().decode()
            if part == 2:
                s = s[:500] + "N" * 30 + s[530:3000] + "R" + s[3001:]
            seq_ids.append(f"NC_{gi:03d}.{part}")
            seqs.append(s)
            seq_taxa.append(t)
    lib = tmp_path / "k2lib"
    for sub, idx in (("bacteria", range(0, 12)), ("archaea/deep", range(12, len(seqs)))):
        os.makedirs(lib / "library" / sub, exist_ok=True)
        with open(lib / "library" / sub / "library.fna", "w") as f:
            for i in idx:
                f.write(f">{seq_ids[i]} some organism\n")
                f.write("\n".join(seqs[i][j:j + 80] for j in range(0, len(seqs[i]), 80)) + "\n")
    (lib / "library" / "ignored.txt").write_text(">NC_000.0\nACGT\n")
    with open(lib / "seqid2taxid.map", "w") as f:
        for sid, t in zip(seq_ids, seq_taxa):
            f.write(f"{sid}\t{t}\n")
        f.write("NC_missing.1\t5\n")
    # base index = all genomes, in Slacken's on-disk layout
    bases = np.frombuffer("".join(seqs).encode(), np.uint8)
    offsets = np.zeros(len(seqs) + 1, np.uint64)
    np.cumsum([len(s) for s in seqs], out=offsets[1:])
    bk, bt = orc.build_records(p, parents, bases, offsets, seq_taxa)
    loc = str(tmp_path / "base")
    conv.write_parquet_dir(loc, bk, bt, buckets=3)
    conv.write_slkrec(loc + ".slkrec", bk, bt)
    with open(loc + ".properties", "w") as f:
        f.write("k=35\nm=31\nbuckets=3\nversion=1\nsplitter=randomXOR\nminimizerSpaces=7\ncanonical=true\n")
    tax = write_ranked_taxonomy(loc + "_taxonomy", parents)
    # sample: many reads from genomes 0, 1 and 6 (a genus-level label), a handful from genome 4, plus noise
    class L:
        pass
    reads = []
    for gi, n in ((0, 400), (1, 300), (6, 300), (4, 12)):
        L.genomes = [genomes[gi]]
        reads += synth.make_reads(L, n, rng, frac_random=0.0, short=0.0)
    L.genomes = []
    reads += synth.make_reads(L, 100, rng)
    reads = [(f"read{i}", r.tobytes().decode()) for i, r in enumerate(reads)]
    fq = tmp_path / "sample.fq"
    with open(fq, "w") as f:
        for t, s in reads:
            f.write(f"@{t}\n{s}\n+\n{'I' * len(s)}\n")
    return dict(p=p, parents=parents, tax=tax, lib=str(lib), loc=loc, fq=str(fq), reads=reads, seqs=seqs, seq_taxa=seq_taxa,
                seq_ids=seq_ids, base=(bk, bt))


def restate(orc, S, criterion, threshold, init_conf=0.15, rank_depth=8, thresholds=(0.0,)):
    p, parents, tax = S["p"], S["parents"], S["tax"]
    base = orc.Index(1, *S["base"])
    counts = {}
    if criterion == "reads":
        for _, s in S["reads"]:
            res, hits = orc.classify_read(p, base, parents, s, None, 2, init_conf)
            if hits and res["classified"]:
                counts[res["taxon"]] = counts.get(res["taxon"], 0) + 1
    else:
        seen = set()
        for _, s in S["reads"]:
            res, hits = orc.classify_read(p, base, parents, s, None, 2, 0.0)
            keys = [sp["key"][0] for sp in orc.spans(p, s)]
            assert len(keys) == len(hits)
            for (t, _), key in zip(hits, keys):
                if t in (-1, -2) or hostmodel.depth(tax, t) < rank_depth:
                    continue
                if criterion == "distinct":
                    if (t, key) in seen:
                        continue
                    seen.add((t, key))
                counts[t] = counts.get(t, 0) + 1
    keep = hostmodel.count_filter(tax, sorted(counts.items()), rank_depth, threshold)
    full = hostmodel.with_descendants(tax, keep)
    sel = [i for i, t in enumerate(S["seq_taxa"]) if t in full]
    b = np.frombuffer("".join(S["seqs"][i] for i in sel).encode(), np.uint8)
    o = np.zeros(len(sel) + 1, np.uint64)
    np.cumsum([len(S["seqs"][i]) for i in sel], out=o[1:])
    dk, dt = orc.build_records(p, parents, b, o, [S["seq_taxa"][i] for i in sel])
    dyn = orc.Index(1, dk, dt)
    lines = {thr: [] for thr in thresholds}
    for title, s in S["reads"]:
        for thr in thresholds:
            res, hits = orc.classify_read(p, dyn, parents, s, None, 2, thr)
            if hits:
                lines[thr].append(orc.output_line(res["classified"], title, res["taxon"], hits, 35))
    return keep, lines, len(dk)


def run2(*args):
    r = subprocess.run([CLI, "classify2", *map(str, args)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return r.stderr


def test_classify2_default_criterion(tmp_path, orc):
    S = setup(tmp_path, orc)
    out = tmp_path / "o" / "dyn"
    os.makedirs(tmp_path / "o")
    err = run2("-i", S["loc"], "-o", out, "--library", S["lib"], "-R", 50, "-c", "0.0", "0.1", S["fq"])
    keep, lines, nrec = restate(orc, S, "reads", 50, thresholds=(0.0, 0.1))
    assert [int(l) for l in open(f"{out}_taxonSet.txt").read().split()] == keep and len(keep) >= 2
    assert f"dynamic index: {nrec} records" in err
    assert read_out(f"{out}_c0.0") == lines[0.0]
    assert read_out(f"{out}_c0.1") == lines[0.1]
    # the genome with only 12 reads is not in the set: its reads are no longer classified to its species
    assert any(l.startswith("C") for l in lines[0.0]) and os.path.exists(f"{out}_c0.0/all_kreport.txt")
    # the same over two device tables (both passes and the dynamic library's construction on each): identical files
    out2 = tmp_path / "o" / "dyn2"
    err2 = run2("-i", S["loc"], "-o", out2, "--library", S["lib"], "-R", 50, "-c", "0.0", "0.1", "--devices", "0,0", S["fq"])
    assert f"dynamic index: {nrec} records" in err2
    assert open(f"{out2}_taxonSet.txt").read() == open(f"{out}_taxonSet.txt").read()
    for sfx in ("0.0", "0.1"):
        assert read_out(f"{out2}_c{sfx}") == read_out(f"{out}_c{sfx}")
        assert open(f"{out2}_c{sfx}/all_kreport.txt").read() == open(f"{out}_c{sfx}/all_kreport.txt").read()


@pytest.mark.parametrize("flag,criterion,threshold", [("-C", "total", 400), ("-D", "distinct", 300)])
def test_classify2_minimizer_criteria(tmp_path, orc, flag, criterion, threshold):
    S = setup(tmp_path, orc, seed=10)
    out = tmp_path / "dyn"
    run2("-i", S["loc"], "-o", out, "--library", S["lib"], flag, threshold, "--rank", "genus", S["fq"])
    keep, lines, _ = restate(orc, S, criterion, threshold, rank_depth=7)
    assert [int(l) for l in open(f"{out}_taxonSet.txt").read().split()] == keep and keep
    assert read_out(f"{out}_c0.0") == lines[0.0]


def test_classify2_repeated_titles_count_once_in_the_detection_pass(tmp_path, orc):
    """The taxon-set detection counts classified READS per taxon through Classifier.classify (Dynamic.scala:133-141), which regroups
    the hits by title (Classifier.scala:92): fragments that share a title are ONE read there.  A sample in which eleven fragments of
    one species' genome carry the same title: counted one by one that species has 30 classified reads, as the reference counts 20.
    With -R 25 the species is in the taxon set only under the wrong count -- so the set, the dynamic library and every output line
    depend on the regrouping in the FIRST pass as well as in the final one."""
    S = setup(tmp_path, orc, seed=13)
    p, parents, tax = S["p"], S["parents"], S["tax"]
    rng = np.random.default_rng(5)
    # a species genome that setup() did not sample gets 30 fragments, eleven of them under one title, placed apart in the file
    class L:
        pass
    import synth as _s
    reads = list(S["reads"])
    gi = next(g for g in (2, 3, 5) if hostmodel.depth(tax, S["seq_taxa"][3 * g]) >= 8)
    sp = S["seq_taxa"][3 * gi]
    L.genomes = [np.frombuffer(S["seqs"][3 * gi].encode(), np.uint8)[3000:]]   # part 0 of that genome, clear of the shared stretch
    extra = _s.make_reads(L, 30, rng, frac_random=0.0, short=0.0, n_single=0.0, n_run=0.0, lowercase=0.0)
    extra = [r.tobytes().decode() for r in extra]
    dup = [("dup_title", s) for s in extra[:11]]
    single = [(f"extra_{i}", s) for i, s in enumerate(extra[11:])]
    reads = reads[:300] + dup[:4] + reads[300:700] + single + dup[4:] + reads[700:]
    S["reads"] = reads
    fq = tmp_path / "sample_dup.fq"
    with open(fq, "w") as f:
        for t, s in reads:
            f.write(f"@{t}\n{s}\n+\n{'I' * len(s)}\n")
    base = orc.Index(1, *S["base"])

    def counts_of(grouped):
        frags = []
        for title, s in reads:
            res, hits = orc.classify_read(p, base, parents, s, None, 2, 0.15)
            distinct = [sp["distinct"] for sp in orc.spans(p, s)]
            frags.append((title, hits, distinct, res))
        counts = {}
        if not grouped:
            for _, hits, _, res in frags:
                if hits and res["classified"]:
                    counts[res["taxon"]] = counts.get(res["taxon"], 0) + 1
            return counts
        for title, hits, distinct in hostmodel.merge_by_title([(t, h, d) for t, h, d, _ in frags]):
            res = orc.classify_hits(parents, hits, distinct, 2, 0.15)
            if res["classified"]:
                counts[res["taxon"]] = counts.get(res["taxon"], 0) + 1
        return counts

    one_by_one, grouped = counts_of(False), counts_of(True)
    keep_wrong = hostmodel.count_filter(tax, sorted(one_by_one.items()), 8, 25)
    keep = hostmodel.count_filter(tax, sorted(grouped.items()), 8, 25)
    assert one_by_one[sp] == 30 and grouped[sp] == 20 and sp in keep_wrong and sp not in keep
    out = tmp_path / "dyn"
    err = run2("-i", S["loc"], "-o", out, "--library", S["lib"], "-R", 25, S["fq"].replace("sample.fq", "sample_dup.fq"))
    assert "1 read titles occur more than once (11 fragments)" in err
    assert [int(l) for l in open(f"{out}_taxonSet.txt").read().split()] == keep
    # the final classification against the dynamic library of THAT set, the repeated title as one row
    full = hostmodel.with_descendants(tax, keep)
    sel = [i for i, t in enumerate(S["seq_taxa"]) if t in full]
    b = np.frombuffer("".join(S["seqs"][i] for i in sel).encode(), np.uint8)
    o = np.zeros(len(sel) + 1, np.uint64)
    np.cumsum([len(S["seqs"][i]) for i in sel], out=o[1:])
    dyn = orc.Index(1, *orc.build_records(p, parents, b, o, [S["seq_taxa"][i] for i in sel]))
    frags = []
    for title, s in reads:
        _, hits = orc.classify_read(p, dyn, parents, s, None, 2, 0.0)
        frags.append((title, hits, [sp["distinct"] for sp in orc.spans(p, s)]))
    want = []
    for title, hits, distinct in hostmodel.merge_by_title(frags):
        res = orc.classify_hits(parents, hits, distinct, 2, 0.0)
        want.append(orc.output_line(res["classified"], title, res["taxon"], hits, 35))
    got = read_out(f"{out}_c0.0")
    assert sorted(got) == sorted(want) and sum(l.split("\t")[1] == "dup_title" for l in got) == 1


def test_classify2_gold_set(tmp_path, orc):
    """-g FILE: the detected set is compared with a gold set (Dynamic.findTaxonSet :262-274) -- read through merged.dmp's
    primaries, taxa without sequence in the library promoted to their nearest ancestor that has (readGoldSet :284-310),
    --promote-gold-set keeping those down to a rank -- and --classify-with-gold builds the dynamic library from the gold set
    instead of a detected one (makeRecords :362-374): no detection pass, no _taxonSet.txt.  Messages and files against a Python
    restatement driven by the oracle."""
    S = setup(tmp_path, orc, seed=12)
    tax, parents = S["tax"], S["parents"]
    ls = (len(parents) - 2) // taxgen.N_RANKS
    species = list(range(7 * ls + 2, 8 * ls + 2))
    genus = list(range(6 * ls + 2, 7 * ls + 2))
    lab = sorted(set(S["seq_taxa"]) | {5})                      # (seqid2taxid.map also labels NC_missing.1 with taxon 5)
    in_library = hostmodel.with_ancestors(tax, lab)
    # a merged.dmp: secondary id 900 -> the first labelled species
    sec = len(parents) + 50
    with open(S["loc"] + "_taxonomy/merged.dmp", "w") as f:
        f.write(f"{sec}\t|\t{species[0]}\t|\n")
    not_in_lib = [t for t in species if t not in in_library][:2]  # species without sequence: promoted to an ancestor that has
    gold_ids = [sec, species[1], genus[1]] + not_in_lib
    gold_file = tmp_path / "gold.txt"
    gold_file.write_text("".join(f"{t}\n" for t in gold_ids) + "\n")
    gold, st = hostmodel.read_gold_set(tax, [str(t) for t in gold_ids], {sec: species[0]}, in_library, 8)
    assert st["not_found"] == len(not_in_lib) > 0 and species[0] in gold and genus[1] not in gold
    # (1) compare only: same files as without -g, plus the comparison line
    out = tmp_path / "cmp"
    r = subprocess.run([CLI, "classify2", "-i", S["loc"], "-o", str(out), "--library", S["lib"], "-R", "50", "-g", str(gold_file), S["fq"]],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    keep, lines, _ = restate(orc, S, "reads", 50)
    assert [int(l) for l in open(f"{out}_taxonSet.txt").read().split()] == keep
    assert read_out(f"{out}_c0.0") == lines[0.0]
    tp = len(set(keep) & set(gold))
    fp, fn = len(keep) - tp, len(gold) - tp
    perc = lambda x: hostmodel.fmt_6_2f(100 * x).strip() + "%"
    assert f"Gold set contained {st['gold']} taxa" in r.stdout
    assert f"{st['not_found']} taxa from gold set not found in library, promoted to {st['promoted']} taxa." in r.stdout
    assert f"Initial adjusted gold set size {st['total']}, filtered at Species to {len(gold)}" in r.stdout
    assert (f"Comparing detected set with supplied gold set. True Positives: {tp}, False Positives: {fp}, False Negatives: {fn}, "
            f"Precision: {perc(tp / (tp + fp))}, Recall: {perc(tp / len(gold))}") in r.stdout
    # (2) --promote-gold-set genus: the promoted ancestors at genus and below stay in the set although they are above the species rank
    gold_p, st_p = hostmodel.read_gold_set(tax, [str(t) for t in gold_ids], {sec: species[0]}, in_library, 8, promote_depth=7)
    r = subprocess.run([CLI, "classify2", "-i", S["loc"], "-o", str(tmp_path / "cmp2"), "--library", S["lib"], "-R", "50", "-g", str(gold_file),
                        "--promote-gold-set", "genus", S["fq"]], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert f"Keeping {st_p['kept']} taxa at rank Genus and below from promoted set" in r.stdout
    assert f"filtered at Species to {len(gold_p)}" in r.stdout
    # (3) --classify-with-gold: the library of the gold set's clades, whatever the sample holds
    out3 = tmp_path / "withgold"
    r = subprocess.run([CLI, "classify2", "-i", S["loc"], "-o", str(out3), "--library", S["lib"], "-g", str(gold_file), "--classify-with-gold",
                        "--promote-gold-set", "genus", S["fq"]], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert not os.path.exists(f"{out3}_taxonSet.txt") and "Comparing detected set" not in r.stdout
    full = hostmodel.with_descendants(tax, gold_p)
    sel = [i for i, t in enumerate(S["seq_taxa"]) if t in full]
    b = np.frombuffer("".join(S["seqs"][i] for i in sel).encode(), np.uint8)
    o = np.zeros(len(sel) + 1, np.uint64)
    np.cumsum([len(S["seqs"][i]) for i in sel], out=o[1:])
    dk, dt = orc.build_records(S["p"], parents, b, o, [S["seq_taxa"][i] for i in sel])
    assert f"dynamic index: {len(dk)} records" in r.stderr
    dyn = orc.Index(1, dk, dt)
    want = []
    for title, sq in S["reads"]:
        res, hits = orc.classify_read(S["p"], dyn, parents, sq, None, 2, 0.0)
        if hits:
            want.append(orc.output_line(res["classified"], title, res["taxon"], hits, 35))
    assert read_out(f"{out3}_c0.0") == want
    # options that qualify a gold set need one
    r = subprocess.run([CLI, "classify2", "-i", S["loc"], "-o", str(out3), "--library", S["lib"], "--classify-with-gold", S["fq"]], capture_output=True, text=True)
    assert r.returncode != 0 and "gold set" in r.stderr
