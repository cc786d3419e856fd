#!/usr/bin/env python3
"""The lane kernel runs 64 fragments in lockstep, so a tile takes as long as its longest fragment: Gbp/s for reads of one
length against reads of mixed lengths with the same mean (device entry).  GPU box; prints one JSON object."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
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
    d_all = torch.from_numpy(bases).cuda()
    R = 4_000_000
    out = {}
    for name, lens in (("fixed_150", np.full(R, 150)), ("trimmed_mostly_150", np.where(rng.random(R) < 0.7, 150, rng.integers(40, 151, R))),
                       ("uniform_50_250", rng.integers(50, 251, R)), ("sorted_uniform_50_250", np.sort(rng.integers(50, 251, R)))):
        lens = lens.astype(np.int64)
        offs = np.zeros(R + 1, np.int64)
        np.cumsum(lens, out=offs[1:])
        total = int(offs[-1])
        starts = rng.integers(0, G * L - 256, R)
        # gather on the device: position p of the concatenation belongs to read r = searchsorted(offs, p)
        d_offs = torch.from_numpy(offs).cuda()
        pos = torch.arange(total, device="cuda")
        rid = torch.searchsorted(d_offs, pos, right=True) - 1
        src = torch.from_numpy(starts).cuda()[rid] + (pos - d_offs[rid])
        d_b = torch.cat([d_all[src], torch.zeros(64, dtype=torch.uint8, device="cuda")])
        del pos, rid, src
        d_t = torch.zeros(R, dtype=torch.int32, device="cuda")
        d_c = torch.zeros(R, dtype=torch.uint8, device="cuda")
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            st.classify_batch_device(d_b.data_ptr(), d_offs.data_ptr(), R, total, d_t.data_ptr(), d_c.data_ptr())
            st.synchronize()
            dt = time.perf_counter() - t0
        out[name] = dict(mean_len=round(float(lens.mean()), 1), ms=round(dt * 1e3, 2), M_reads_per_s=round(R / dt / 1e6, 1),
                         Gbp_per_s=round(total / dt / 1e9, 1))
        del d_b, d_offs
    print(json.dumps(out))


if __name__ == "__main__":
    main()
