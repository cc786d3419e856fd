import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import slacken_amd
parents = np.array([0, 0, 1, 1, 2], np.int32)
seq = np.frombuffer(b"ACGTTGCAAGGCTTAACGGATCGATTACAGGCATCGATCGGATCGATCGTAGCTAGCTAGGATCGATCGATCGGGATTTACG" * 3, np.uint8)
off = np.array([0, len(seq)], np.uint64)
for i in range(16):
    ix = slacken_amd.Index(expected_records=int(3e9), max_taxon=4)     # 2^29 buckets = 32 GiB
    ix.set_taxonomy(parents)
    ix.add_sequences(seq, off, [3])
    ix.finalize()
    st = ix.stream()
    r = st.classify_batch(seq, off)
    assert r["classified"][0][0] == 1
    st.close(); ix.close()
    print(i, ix if False else "ok", flush=True)
# wide indexes too
for i in range(8):
    ix = slacken_amd.Index(k=50, m=40, expected_records=int(5e8), max_taxon=4)   # 2^30 slots * 20 B = 20 GiB
    ix.set_taxonomy(parents); ix.finalize(); ix.close()
print("no leak")
