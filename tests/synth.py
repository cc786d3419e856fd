"""Synthetic libraries and reads for the parity tests (the real standard-224 library is an S3
download and the reference's tiny genome file is missing from the mount).  numpy only; the oracle supplies the
minimizers of the synthetic genomes.  Test infrastructure."""
import numpy as np

import taxgen

ACGT = np.frombuffer(b"ACGT", np.uint8)
COMP = np.zeros(256, np.uint8)
for a, b in zip(b"ACGTacgtNn", b"TGCAtgcaNn"):
    COMP[a] = b


def random_dna(n, rng):
    return ACGT[rng.integers(0, 4, n)]


def revcomp(a):
    return COMP[a][::-1]


class Library:
    """genomes (one per chosen taxon) -> unique (key, LCA taxon) records (+ random padding records)."""

    def __init__(self, orc, p, parents, n_genomes=8, genome_len=20000, pad_records=0, seed=224):
        rng = np.random.default_rng(seed)
        self.parents = parents
        taxa = np.array(taxgen.defined_taxa(parents))
        leaves = np.setdiff1d(taxa, parents[taxa])  # taxa that are nobody's parent
        self.genome_taxa = rng.choice(leaves, size=min(n_genomes, len(leaves)), replace=False)
        self.genomes = [random_dna(genome_len, rng) for _ in self.genome_taxa]
        # shared segments make some minimizers occur in several genomes => LCA records above the leaves
        for g in range(1, len(self.genomes)):
            src = self.genomes[rng.integers(0, g)]
            a = rng.integers(0, genome_len - 600)
            self.genomes[g][a:a + 600] = src[a:a + 600]
        keys, tax = [], []
        for g, t in zip(self.genomes, self.genome_taxa):
            k = orc.minimizer_keys(p, g.tobytes())
            keys.append(k)
            tax.append(np.full(len(k), t, np.int32))
        keys, tax = np.concatenate(keys), np.concatenate(tax)
        order = np.argsort(keys, kind="stable")
        keys, tax = keys[order], tax[order]
        uniq, start = np.unique(keys, return_index=True)
        out_tax = tax[start].copy()
        ends = np.append(start[1:], len(keys))
        for i in np.nonzero(ends - start > 1)[0]:  # TaxonLCA aggregation, LowestCommonAncestor.scala:152-170
            t = 0
            for x in tax[start[i]:ends[i]]:
                t = orc.lca(parents, t, int(x))
            out_tax[i] = t
        if pad_records:
            space = np.uint64(p.space[0])
            pad = (rng.integers(0, 2**63, pad_records, dtype=np.uint64) * np.uint64(2) +
                   rng.integers(0, 2, pad_records, dtype=np.uint64)) & space
            pad = np.setdiff1d(pad.view(np.int64), uniq)
            uniq = np.concatenate([uniq, pad])
            out_tax = np.concatenate([out_tax, rng.choice(taxa, size=len(pad)).astype(np.int32)])
        self.keys, self.taxa = uniq.astype(np.int64), out_tax.astype(np.int32)


def make_reads(lib, n_reads, rng, length=150, frac_random=0.2, sub_rate=0.01, n_single=0.05, n_run=0.02,
               vary_length=False, lowercase=0.02, short=0.02):
    """-> list of uint8 arrays. Mix per SURVEY 8d plus the edge cases of SURVEY 3.3."""
    reads = []
    for _ in range(n_reads):
        L = int(rng.integers(20, 2 * length)) if vary_length else length
        if rng.random() < short:
            L = int(rng.integers(0, 40))
        if rng.random() < frac_random or not lib.genomes:
            r = random_dna(L, rng)
        else:
            g = lib.genomes[rng.integers(0, len(lib.genomes))]
            L = min(L, len(g))
            a = rng.integers(0, len(g) - L + 1)
            r = g[a:a + L].copy()
            if rng.random() < 0.5:
                r = revcomp(r).copy()
            subs = rng.random(L) < sub_rate
            r[subs] = random_dna(int(subs.sum()), rng)
        if L > 0 and rng.random() < n_single:
            r[rng.integers(0, L)] = ord("N")
        if L > 45 and rng.random() < n_run:
            a = rng.integers(0, L - 40)
            r[a:a + 40] = ord("N")
        if rng.random() < lowercase:
            r = np.frombuffer(r.tobytes().lower(), np.uint8).copy()
        reads.append(r)
    return reads


def pack(reads):
    """list of uint8 arrays -> (bases uint8[total], offsets uint64[R+1])"""
    lens = np.array([len(r) for r in reads], np.uint64)
    offsets = np.zeros(len(reads) + 1, np.uint64)
    np.cumsum(lens, out=offsets[1:])
    bases = np.concatenate(reads) if reads and offsets[-1] > 0 else np.zeros(0, np.uint8)
    return bases.astype(np.uint8), offsets
