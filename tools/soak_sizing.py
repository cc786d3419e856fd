import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import slacken_amd
bad = 0
rng = np.random.default_rng(11)
for case in range(400):
    n = int(10 ** rng.uniform(3, 6.7))
    mt = int(2 ** rng.uniform(8, 24))
    keys = np.unique(rng.integers(-2**62, 2**62, n, dtype=np.int64) & ~np.int64(0x33333333))
    taxa = rng.integers(1, mt + 1, len(keys)).astype(np.int32)
    ix = slacken_amd.Index(expected_records=len(keys), max_taxon=mt)
    try:
        ix.append(keys, taxa)
        info = ix.info()
        assert info.records == len(keys), (info.records, len(keys))
    except Exception as e:
        bad += 1
        print("case", case, "n", len(keys), "max_taxon", mt, "->", e)
    ix.close()
print("done, failures:", bad)
