#!/usr/bin/env python3
"""slk_classify_batch_packed / slk_classify_batch from pinned host memory, one call of R reads of 150 bp, over sub-batch sizes
(SLK_HOST_SUBBATCH): the PCIe-inclusive rate of the entry a JNI shim calls.  Prints one JSON object; run on the GPU box."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch  # noqa: F401
    import slacken_amd
    import taxgen
    from slacken_amd import capi
    rng = np.random.default_rng(5)
    parents = taxgen.taxonomy(8 * 1024, rng)
    taxa = np.array(taxgen.defined_taxa(parents))
    G, L = 256, 4 << 20
    bases = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, G * L, dtype=np.uint8)]
    ix = slacken_amd.Index(expected_records=int(G * L * 0.4), max_taxon=len(parents) - 1)
    ix.set_taxonomy(parents)
    ix.add_sequences(bases, np.arange(G + 1, dtype=np.uint64) * np.uint64(L), rng.choice(taxa[len(taxa) // 2:], G).astype(np.int32))
    ix.finalize()
    R = int(os.environ.get("R", 4_000_000))
    starts = rng.integers(0, G * L - 150, R)
    rb = bases[(starts[:, None] + np.arange(150)[None, :]).reshape(-1)]
    ro = np.arange(R + 1, dtype=np.uint64) * np.uint64(150)
    st = ix.stream()
    prb = capi.pinned_array(rb.shape, np.uint8); prb[:] = rb
    pro = capi.pinned_array(ro.shape, np.uint64); pro[:] = ro
    pout = dict(taxon=capi.pinned_array((1, R), np.int32), classified=capi.pinned_array((1, R), np.uint8),
                num_distinct=capi.pinned_array((R,), np.int32), total_kmers=capi.pinned_array((R,), np.int32))
    codes, valid = capi.pack_bases(rb, pinned=True)
    want = st.classify_batch(rb, ro, with_hits=False)

    def timed(fn, reps=5):
        best = 1e9
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            best = min(best, time.perf_counter() - t0)
        return best

    out = {"reads": R}
    for sub in (1 << 17, 1 << 18, 1 << 19, 1 << 20, 1 << 21):
        os.environ["SLK_HOST_SUBBATCH"] = str(sub)
        dt = timed(lambda: st.classify_batch(None, pro, with_hits=False, out=pout, packed=(codes, valid)))
        assert np.array_equal(pout["taxon"], want["taxon"])
        da = timed(lambda: st.classify_batch(prb, pro, with_hits=False, out=pout))
        assert np.array_equal(pout["taxon"], want["taxon"])
        out[f"sub_{sub}"] = dict(packed_M_reads_per_s=round(R / dt / 1e6, 1), packed_GB_per_s_up=round((codes.nbytes + valid.nbytes + ro.nbytes) / dt / 1e9, 1),
                                 ascii_M_reads_per_s=round(R / da / 1e6, 1), ascii_GB_per_s_up=round((rb.nbytes + ro.nbytes) / da / 1e9, 1))
    os.environ.pop("SLK_HOST_SUBBATCH")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
