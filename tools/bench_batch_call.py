#!/usr/bin/env python3
"""Wall clock of slk_classify_batch (host buffers in, host buffers out) by batch size, with and without hit lists.
Run on the GPU box with SLK_DEBUG_CALL_TIMING=1 to get the phases of each call on stderr."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch  # noqa: F401  (its HIP runtime first)
    import slacken_amd
    import taxgen
    rng = np.random.default_rng(3)
    parents = taxgen.taxonomy(8 * 64, rng)
    taxa = np.array(taxgen.defined_taxa(parents))
    G, L = 64, 1 << 20
    acgt = np.frombuffer(b"ACGT", np.uint8)
    genome = acgt[rng.integers(0, 4, G * L, dtype=np.uint8)]
    ix = slacken_amd.Index(expected_records=G * L // 2, max_taxon=len(parents) - 1)
    ix.set_taxonomy(parents)
    ix.add_sequences(genome, np.arange(G + 1, dtype=np.uint64) * np.uint64(L), rng.choice(taxa[len(taxa) // 2:], G).astype(np.int32))
    ix.finalize()
    st = ix.stream()
    out = {}
    for R in (50_000, 200_000, 1_000_000):
        starts = rng.integers(0, G * L - 150, R)
        bases = np.ascontiguousarray(genome[(starts[:, None] + np.arange(150)[None, :])].reshape(-1))
        offsets = np.arange(R + 1, dtype=np.uint64) * np.uint64(150)
        for hits in (False, True):
            for rep in range(4):
                t0 = time.perf_counter()
                st.classify_batch(bases, offsets, with_hits=hits)
                dt = time.perf_counter() - t0
            out[f"R={R} hits={hits}"] = round(dt * 1e3, 2)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
