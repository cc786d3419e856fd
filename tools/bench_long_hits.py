#!/usr/bin/env python3
"""Long unpaired reads WITH hit lists (per-read output lines): kernel time of the host entry's launches (the engine's own events,
slk_stream_last_stage_ms) by read length.  Run twice -- default (hit lists from the wave-per-fragment kernel) and SLK_SEG_HITS=1 (from
5000 bases: the lane-per-segment kernel) -- to compare the two routes."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch  # noqa: F401  (PyTorch's HIP runtime first)
    import slacken_amd
    import taxgen
    rng = np.random.default_rng(3)
    parents = taxgen.taxonomy(8 * 64, rng)
    taxa = np.array(taxgen.defined_taxa(parents))
    G, L = 64, 1 << 22
    acgt = np.frombuffer(b"ACGT", np.uint8)
    bases = acgt[rng.integers(0, 4, G * L, dtype=np.uint8)]
    ix = slacken_amd.Index(expected_records=G * L // 2, max_taxon=len(parents) - 1)
    ix.set_taxonomy(parents)
    ix.add_sequences(bases, np.arange(G + 1, dtype=np.uint64) * np.uint64(L), rng.choice(taxa[len(taxa) // 2:], G).astype(np.int32))
    ix.finalize()
    st = ix.stream()
    out = dict(seg_min_len=os.environ.get("SLK_SEG_MIN_LEN", "default"), seg_hits=os.environ.get("SLK_SEG_HITS", "0"))
    with_hits = os.environ.get("SLK_BENCH_NO_HITS", "0") != "1"     # (=1: the same reads without hit lists, for the cost of the lists)
    out["with_hits"] = with_hits
    for L_read in [int(x) for x in os.environ.get('LENGTHS', '5000,10000,30000,100000').split(',')]:
        R = max(64, int(float(os.environ.get('SLK_BENCH_BASES', '2e8'))) // L_read)
        starts = rng.integers(0, G * L - L_read, R)
        rb = bases[(starts[:, None] + np.arange(L_read)[None, :]).reshape(-1)].copy()
        rb[np.arange(0, R, 100) * L_read + L_read // 3] = ord("N")      # one N in one read of a hundred
        ro = np.arange(0, (R + 1) * L_read, L_read, dtype=np.uint64)
        ms = []
        for _ in range(3):
            res = st.classify_batch(rb, ro, thresholds=(0.0,), with_hits=with_hits)
            ms.append(sum(st.last_stage_ms()))
        if not with_hits:
            out[f"{L_read}bp"] = dict(reads=R, kernel_ms=round(min(ms), 2), Gbp_per_s=round(R * L_read / min(ms) / 1e6, 1))
            continue
        out[f"{L_read}bp"] = dict(reads=R, kernel_ms=round(min(ms), 2), Gbp_per_s=round(R * L_read / min(ms) / 1e6, 1),
                                  hits=int(res["hit_offsets"][-1]), classified=round(float(res["classified"][0].mean()), 3),
                                  taxon_sum=int(res["taxon"].astype(np.int64).sum()), hit_sum=int(res["hits"]["count"].astype(np.int64).sum()))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
