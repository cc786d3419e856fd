#!/usr/bin/env python3
"""One CLI-sized batch (53 600 reads x 150 bp) through slk_classify_batch with hit lists, into a buffer that has been written before
(as the CLI's recycled result buffers have), alone on the machine: with SLK_DEBUG_CALL_TIMING=1 the library prints where a call's time
goes.  GPU box."""
import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import slacken_amd, taxgen
rng = np.random.default_rng(3)
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
R = int(os.environ.get("R", 53600))
starts = rng.integers(0, G * L - 150, R)
rb = bases[(starts[:, None] + np.arange(150)[None, :])].reshape(-1).copy()
offs = np.arange(0, (R + 1) * 150, 150, dtype=np.uint64)
import ctypes as C
from slacken_amd import capi
lib = capi.lib()
cap = len(rb) + R + 1
taxon = np.zeros((1, R), np.int32); cls = np.zeros((1, R), np.uint8); nd = np.zeros(R, np.int32); tk = np.zeros(R, np.int32)
hit_off = np.zeros(R + 1, np.uint64)
hits = np.ones(cap, capi.HIT_DTYPE)          # (touched: a recycled buffer of the CLI)
thr = (C.c_double * 1)(0.0)
if os.environ.get("MERGED") == "1":
    st.set_merged_hits(True)
for i in range(10):
    t = time.perf_counter()
    rc = lib.slk_classify_batch(ix.h, st.h, capi._ptr(rb), capi._ptr(offs), None, None, R, 2, thr, 1, capi._ptr(taxon), capi._ptr(cls),
                                capi._ptr(nd), capi._ptr(tk), capi._ptr(hit_off), capi._ptr(hits), cap)
    print("call", i, rc, round((time.perf_counter() - t) * 1e3, 2), "ms", int(hit_off[R]), file=sys.stderr)
