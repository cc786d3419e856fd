#!/usr/bin/env python3
"""Long reads on the device entry (bases resident in HBM): Gbp/s by read length, with 1 % of the reads carrying an N.
Run twice, with SLK_SEG_MIN_LEN=0 (wave kernel only) and without, to compare the two routes of the deferred fragments."""
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
    kms = [int(x) for x in os.environ.get("KMS", "35,31,7").split(",")]   # k, m, spaces of the splitter
    ix = slacken_amd.Index(k=kms[0], m=kms[1], spaces=kms[2], expected_records=int(os.environ.get('EXPECTED', G * L // 2)), max_taxon=len(parents) - 1,
                           load_factor=float(os.environ.get('LOAD', 0)))   # (EXPECTED=134217722 LOAD=0.5: a power of two of buckets)
    ix.set_taxonomy(parents)
    ix.add_sequences(bases, np.arange(G + 1, dtype=np.uint64) * np.uint64(L), rng.choice(taxa[len(taxa) // 2:], G).astype(np.int32))
    ix.finalize()
    st = ix.stream()
    d_all = torch.from_numpy(bases).cuda()
    out = dict(seg_min_len=os.environ.get("SLK_SEG_MIN_LEN", "default"))
    lengths = [int(x) for x in os.environ.get("LENGTHS", "1500,3000,5000,10000,30000,100000").split(",")]
    for L_read in lengths:
        R = max(64, 1_000_000_000 // L_read)
        for with_n in (False, True):
            starts = torch.from_numpy(rng.integers(0, G * L - L_read, R)).cuda()
            idx = starts[:, None] + torch.arange(L_read, device="cuda")[None, :]
            d_b = torch.cat([d_all[idx.reshape(-1)], torch.zeros(64, dtype=torch.uint8, device="cuda")])
            del idx
            if with_n:   # one N in one read of a hundred
                at = torch.arange(0, R, 100, device="cuda") * L_read + L_read // 3
                d_b[at] = ord("N")
            d_o = torch.arange(0, (R + 1) * L_read, L_read, dtype=torch.int64, device="cuda")
            d_t = torch.zeros(R, dtype=torch.int32, device="cuda")
            d_c = torch.zeros(R, dtype=torch.uint8, device="cuda")
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                st.classify_batch_device(d_b.data_ptr(), d_o.data_ptr(), R, R * L_read, d_t.data_ptr(), d_c.data_ptr())
                st.synchronize()
                dt = time.perf_counter() - t0
            out[f"{L_read}bp" + ("_1pct_with_N" if with_n else "")] = dict(reads=R, ms=round(dt * 1e3, 2), Gbp_per_s=round(R * L_read / dt / 1e9, 1),
                                                                    classified=round(float(d_c.float().mean().item()), 3),
                                                                    taxon_sum=int(d_t.long().sum().item()))
            del d_b, d_o
    print(json.dumps(out))


if __name__ == "__main__":
    main()
