#!/usr/bin/env python3
"""A host call of 1.3 M fragments with hit lists, single and paired, un-merged and merged: in one piece against the default sub-batches
(2^19 fragments: the sizes at which the pipelined route really runs; the test suite moves SLK_HOST_SUBBATCH down instead).  GPU box."""
import os, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import slacken_amd, taxgen
rng = np.random.default_rng(5)
parents = taxgen.taxonomy(8 * 64, rng)
taxa = np.array(taxgen.defined_taxa(parents))
G, L = 64, 1 << 20
acgt = np.frombuffer(b"ACGT", np.uint8)
bases = acgt[rng.integers(0, 4, G * L, dtype=np.uint8)]
ix = slacken_amd.Index(expected_records=G * L // 2, max_taxon=len(parents) - 1)
ix.set_taxonomy(parents)
ix.add_sequences(bases, np.arange(G + 1, dtype=np.uint64) * np.uint64(L), rng.choice(taxa[len(taxa) // 2:], G).astype(np.int32))
ix.finalize()
st = ix.stream()
R = 1_300_000
def reads(lo, hi):
    lens = rng.integers(lo, hi, R)
    offs = np.zeros(R + 1, np.uint64); np.cumsum(lens, out=offs[1:])
    starts = rng.integers(0, G * L - hi, R)
    idx = np.repeat(starts - offs[:-1].astype(np.int64), lens) + np.arange(int(offs[-1]))
    b = bases[idx].copy()
    b[rng.integers(0, len(b), len(b) // 500)] = ord("N")
    return b, offs
b1, o1 = reads(60, 200)
b2, o2 = reads(40, 160)
for paired in (False, True):
    mb, mo = (b2, o2) if paired else (None, None)
    out = {}
    for name, sub in (("one piece", "100000000"), ("sub-batches", "")):
        if sub: os.environ["SLK_HOST_SUBBATCH"] = sub
        else: os.environ.pop("SLK_HOST_SUBBATCH", None)
        for merged in (False, True):
            st.set_merged_hits(merged)
            out[name, merged] = st.classify_batch(b1, o1, mb, mo, thresholds=(0.0, 0.2), with_hits=True)
    st.set_merged_hits(False)
    for merged in (False, True):
        a, b = out["one piece", merged], out["sub-batches", merged]
        for k in ("taxon", "classified", "num_distinct", "total_kmers", "hit_offsets", "hits"):
            assert np.array_equal(a[k], b[k]), (paired, merged, k)
        print("paired" if paired else "single", "merged" if merged else "un-merged", "ok:", len(a["hits"]), "entries", flush=True)
