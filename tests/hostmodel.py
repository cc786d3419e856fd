"""Independent Python restatement of the reference's host-side pieces around the hot path, used to check the C++ host layer
(slacken_amd/host): KrakenReport (S/slacken/KrakenReport.scala:26-116), Taxonomy.fromNodesAndNames (S/slacken/Taxonomy.scala:
81-109), FASTA/FASTQ text parsing (S/kmers/input/FileInputs.scala:155-221).  No reference output exists for these in the
repository (its only test is a property spec, KrakenReportProps.scala:33-52, restated in test_host_cli.py): parity unpinned."""
import re
from decimal import Decimal, ROUND_HALF_UP

RANKS = ["unclassified", "root", "superkingdom", "kingdom", "phylum", "class", "order", "family", "genus", "species"]
CODES = ["U", "R", "D", "K", "P", "C", "O", "F", "G", "S"]


class Taxonomy:
    def __init__(self, nodes, names, merged=()):
        n = max([t + 1 for t, _, _ in nodes] + [a + 1 for a, _ in merged] + [2])
        self.parents = [0] * n
        self.ranks = [None] * n
        self.names = [None] * n
        for t, nm in names:
            self.names[t] = nm
        self.names[0] = "unclassified"
        for t, p, r in nodes:
            self.parents[t] = p
            self.ranks[t] = RANKS.index(r) if r in RANKS else None
        self.parents[1] = 0
        self.ranks[0], self.ranks[1] = 0, 1
        self.children = [[] for _ in range(n)]
        for t in range(n):
            if self.parents[t] != 0 or t == 1:
                self.children[self.parents[t]].insert(0, t)


def fmt_6_2f(x):
    # java.util.Formatter rounds the shortest round-trip digits (repr) HALF_UP, not the exact binary value
    s = str(Decimal(repr(float(x))).quantize(Decimal("0.01"), rounding=ROUND_HALF_UP))
    return s.rjust(6)


def kraken_report(tax, counts):
    own, clade, total = {}, {}, 0
    for t, c in counts:
        own[t] = own.get(t, 0) + c
        total += c
        p = t
        while p != 0:
            clade[p] = clade.get(p, 0) + c
            p = tax.parents[p]
        if t == 0:
            clade[0] = c
    out = ["#Perc\tAggregate\tIn taxon\tRank\tTaxon\tName"]

    def line(t, rank, rd, depth):
        return "\t".join([fmt_6_2f(100.0 * clade.get(t, 0) / total), str(clade.get(t, 0)), str(own.get(t, 0)),
                          CODES[rank] + (str(rd) if rd else ""), str(t), "  " * depth + (tax.names[t] or "")])

    def dfs(t, rank, rd, depth):
        if tax.ranks[t] is not None:
            rank, rd = tax.ranks[t], 0
        else:
            rd += 1
        out.append(line(t, rank, rd, depth))
        kids = sorted(((c, clade.get(c, 0)) for c in tax.children[t]), key=lambda x: -x[1])   # stable
        for c, n in kids:
            if n > 0:
                dfs(c, rank, rd, depth + 1)

    if own.get(0, 0):
        out.append(line(0, 0, 0, 0))
    dfs(1, 1, 0, 0)
    return out, own, clade


def _jsplit(s, pattern):
    """java.lang.String.split: trailing empty strings removed"""
    parts = re.split(pattern, s)
    while parts and parts[-1] == "":
        parts.pop()
    return parts if parts else ([] if s else [""])


def parse_fasta(text):
    out = []
    for rec in text.split(">"):
        spl = _jsplit(rec, "[\n\r]+")
        if len(spl) >= 2:
            out.append((_jsplit(spl[0], " ")[0] if _jsplit(spl[0], " ") else "", "".join(spl[1:])))
    return out


def parse_fastq(text):
    lines = re.split("\r\n|\n|\r", text)      # Spark's text reader: \n, \r\n and \r all end a line
    if lines and lines[-1] == "":
        lines.pop()
    out = []
    for i in range(len(lines)):
        w = lines[i:i + 4]
        if len(w) >= 3 and w[0][:1] == "@" and w[2][:1] == "+":
            out.append((w[0].split(" ")[0][1:], w[1]))
    return out


# ---- Dynamic (two-step) library: S/slacken/Dynamic.scala:174-185,213-243,362-374; Taxonomy.scala:217-224,304-311 ----
def depth(tax, t):
    while t != 0:
        if tax.ranks[t] is not None:
            return tax.ranks[t] - 1
        t = tax.parents[t]
    return -1


def count_filter(tax, counts, rank_depth, threshold):
    """CountFilter.taxa: keys of the aggregator at depth >= rank whose clade total reaches the threshold (ascending)."""
    clade = {}
    for t, c in counts:
        p = t
        while p != 0:
            clade[p] = clade.get(p, 0) + c
            p = tax.parents[p]
    return sorted(t for t, _ in counts if depth(tax, t) >= rank_depth and clade.get(t, 0) >= threshold)


def with_descendants(tax, taxa):
    out, stack = set(), list(taxa)
    while stack:
        t = stack.pop()
        if t in out:
            continue
        out.add(t)
        stack.extend(tax.children[t])
    return out


def read_gold_set(tax, lines, primary, in_library, rank_depth, promote_depth=None):
    """Dynamic.readGoldSet (S/slacken/Dynamic.scala:284-310): the file's taxa mapped to their primaries; those without sequence in
    the library replaced by the nearest ancestor that has; filtered at the reclassification rank, the promoted ones kept down to
    promote_depth if given.  in_library: the labelled taxa with their ancestors (GenomeLibrary.taxonSet).  -> sorted list"""
    gold = {primary.get(int(l.split(",")[0]), int(l.split(",")[0])) for l in lines if l.strip()}
    not_found = {t for t in gold if t not in in_library}
    promoted = set()
    for t in not_found:
        p = t
        while p != 0:
            if p in in_library:
                promoted.add(p)
                break
            p = tax.parents[p]
    kept = {t for t in promoted if depth(tax, t) >= promote_depth} if promote_depth is not None else set()
    total = gold | promoted
    return sorted({t for t in total if depth(tax, t) >= rank_depth} | kept), dict(gold=len(gold), not_found=len(not_found),
                                                                                 promoted=len(promoted), kept=len(kept), total=len(total))


def with_ancestors(tax, taxa):
    out = set()
    for t in taxa:
        while t != 0 and t not in out:
            out.add(t)
            t = tax.parents[t]
    return out


# ---- regrouping by title: Classifier.spansToGroupedHits (S/slacken/Classifier.scala:77-96) + classifyHits (:124-147) ----
def merge_by_title(fragments):
    """fragments: [(title, hits, distinct)] in input order, hits = [(taxon, count)...] in ordinal order (a fragment without
    spans has no span rows and takes no part).  The reference groups the span rows of ALL fragments by title string
    (groupBy("seqTitle").agg(collect_list(hit)), :92) and sorts each group's hits by ordinal with a stable sort
    (java.util.Arrays.sort on objects, :136).  The order of equal ordinals is whatever collect_list delivered -- not
    defined by Spark; this host fixes it to input order.  Returns [(title, merged hits, merged distinct)] in order of first
    appearance."""
    groups, order = {}, []
    for title, hits, distinct in fragments:
        if not hits:
            continue
        if title not in groups:
            groups[title] = []
            order.append(title)
        groups[title].append((hits, distinct))
    out = []
    for title in order:
        rows = [(ordinal, seq, h, d) for seq, (hits, distinct) in enumerate(groups[title])
                for ordinal, (h, d) in enumerate(zip(hits, distinct))]
        rows.sort(key=lambda r: r[0])   # stable: equal ordinals keep fragment order
        out.append((title, [r[2] for r in rows], [r[3] for r in rows]))
    return out


def paired_join(records1, records2):
    """PairedInputReader.getFragments (S/kmers/input/InputReader.scala:104-119): inner join of the two files' records on the
    header with /1 and /2 removed -- every record of file 1 with EVERY record of file 2 that has the same header.
    records: [(header, seq)].  Returns [(header, seq1, seq2)], file-1 order then file-2 order (Spark's order is undefined)."""
    strip = lambda h, suf: h[:-len(suf)] if h.endswith(suf) else h
    by2 = {}
    for h, s in records2:
        by2.setdefault(strip(h, "/2"), []).append(s)
    return [(strip(h, "/1"), s, m) for h, s in records1 for m in by2.get(strip(h, "/1"), [])]
