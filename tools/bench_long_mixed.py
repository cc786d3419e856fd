#!/usr/bin/env python3
"""Long reads of MIXED lengths on the device entry: a log-normal length distribution like a nanopore run's (median ~3 kbp, a tail to
50 kbp; also one confined to 1-5 kbp, the band of the lane kernel's long variant), in random order and sorted by length -- the gap
between the two is what a length-ordered hand-on list could recover.  GPU box; prints one JSON object."""
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
    out = {}
    total_target = 1_000_000_000
    for name, lo, hi, med in (("nanopore_like_200_50000", 200, 50000, 3000), ("band_1001_4999", 1001, 4999, 2200)):
        lens = np.clip(rng.lognormal(np.log(med), 0.9, 2_000_000), lo, hi).astype(np.int64)
        lens = lens[:np.searchsorted(np.cumsum(lens), total_target)]
        for order in ("random", "sorted"):
            ll = np.sort(lens)[::-1].copy() if order == "sorted" else lens
            R = len(ll)
            offs = np.zeros(R + 1, np.int64)
            np.cumsum(ll, out=offs[1:])
            total = int(offs[-1])
            starts = rng.integers(0, G * L - hi, R)
            d_offs = torch.from_numpy(offs).cuda()
            pos = torch.arange(total, device="cuda")
            rid = torch.searchsorted(d_offs, pos, right=True) - 1
            d_b = d_all[torch.from_numpy(starts).cuda()[rid] + (pos - d_offs[rid])]
            del pos, rid
            d_t = torch.zeros(R, dtype=torch.int32, device="cuda")
            d_c = torch.zeros(R, dtype=torch.uint8, device="cuda")
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                st.classify_batch_device(d_b.data_ptr(), d_offs.data_ptr(), R, total, d_t.data_ptr(), d_c.data_ptr())
                st.synchronize()
                dt = time.perf_counter() - t0
            out[f"{name}_{order}"] = dict(reads=R, mean_len=round(float(ll.mean())), ms=round(dt * 1e3, 2), Gbp_per_s=round(total / dt / 1e9, 1),
                                          share_of_bases_1001_4999=round(float(ll[(ll > 1000) & (ll < 5000)].sum() / total), 2))
            del d_b, d_offs
    print(json.dumps(out))


if __name__ == "__main__":
    main()
