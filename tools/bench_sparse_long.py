#!/usr/bin/env python3
"""Long reads that hit the library sparsely and in many places -- mostly novel sequence with short pieces of several genomes -- on a
deep taxonomy, with a confidence threshold that their best clade does not meet: resolveTree's confidence walk goes all the way up.
Segment kernel (classification only) and wave kernel (hit lists wanted / SLK_SEG_MIN_LEN=0).  GPU box; prints one JSON object."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import slacken_amd
    rng = np.random.default_rng(5)
    chain = int(os.environ.get("CHAIN", 3))
    parents, level = [0, 0], [1]
    for d in range(8):
        nxt = []
        for p in level:
            for _ in range(2):
                up = p
                for _ in range(chain):
                    parents.append(up)
                    up = len(parents) - 1
                parents.append(up)
                nxt.append(len(parents) - 1)
        level = nxt
    parents = np.array(parents, np.int32)
    leaves = level
    L = 1 << 16
    acgt = np.frombuffer(b"ACGT", np.uint8)
    genomes = [acgt[rng.integers(0, 4, L, dtype=np.uint8)] for _ in leaves]      # unrelated genomes: a hit names a leaf
    bases = np.concatenate(genomes)
    ix = slacken_amd.Index(expected_records=int(len(leaves) * L * 0.4), max_taxon=len(parents) - 1)
    ix.set_taxonomy(parents)
    ix.add_sequences(bases, np.arange(len(leaves) + 1, dtype=np.uint64) * np.uint64(L), np.array(leaves, np.int32))
    ix.finalize()
    st = ix.stream()
    R, LR = 20000, 10000
    reads = acgt[rng.integers(0, 4, (R, LR), dtype=np.uint8)]
    for r in range(R):                                 # 12 pieces of 60 bases from 12 random genomes: ~12 taxa, ~3 % of the k-mers
        for _ in range(12):
            g = genomes[int(rng.integers(0, len(genomes)))]
            a, b = int(rng.integers(0, L - 60)), int(rng.integers(0, LR - 60))
            reads[r, b:b + 60] = g[a:a + 60]
    d_b = torch.from_numpy(reads.reshape(-1)).cuda()
    d_o = torch.arange(0, (R + 1) * LR, LR, dtype=torch.int64, device="cuda")
    d_t = torch.zeros(2 * R, dtype=torch.int32, device="cuda")
    d_c = torch.zeros(2 * R, dtype=torch.uint8, device="cuda")
    out = dict(chain=chain, depth=int(8 * (chain + 1)), reads=R, read_len=LR)
    for name, env in (("segment_kernel", {}), ("wave_kernel", {"SLK_SEG_MIN_LEN": "0"})):
        os.environ.pop("SLK_SEG_MIN_LEN", None)
        os.environ.update(env)
        for thr in ((0.0,), (0.0, 0.15)):
            for _ in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                st.classify_batch_device(d_b.data_ptr(), d_o.data_ptr(), R, R * LR, d_t.data_ptr(), d_c.data_ptr(), thresholds=thr)
                st.synchronize()
                dt = time.perf_counter() - t0
            out[f"{name}_thresholds_{len(thr)}"] = dict(ms=round(dt * 1e3, 2), Gbp_per_s=round(R * LR / dt / 1e9, 1),
                                                        classified=[float(x) for x in d_c[:len(thr) * R].view(len(thr), R).float().mean(1).cpu()])
    print(json.dumps(out))


if __name__ == "__main__":
    main()
