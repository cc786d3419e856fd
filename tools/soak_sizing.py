"""Random table sizes, taxon ranges and load factors (default and given, up to 0.85): every record must find a cell, every key must
be found again with its taxon, absent keys must miss, and the records must come back out of the table (the range reduction onto
any number of buckets and its inverse, the displacement field, the buckets' overflow flag).  Every third case the library OUTGROWS
its table -- expected_records is a fraction of what is appended, in several calls, so that the load passes what the cells'
displacement field can count (loads of 0.8 to over 1 against 4- to 6-bit fields) -- and the table must grow instead of refusing
(capi.hip: grow_table); SLK_SOAK_HOST_GROW=1 makes the growth go through host memory.  Run on the GPU box."""
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import slacken_amd
bad = grown = 0
rng = np.random.default_rng(11)
CASES = int(sys.argv[1]) if len(sys.argv) > 1 else 400
for case in range(CASES):
    n = int(10 ** rng.uniform(3, 6.7))
    mt = int(2 ** rng.uniform(8, 24))
    lf = 0.0 if case % 2 == 0 else float(rng.uniform(0.3, 0.85))
    keys = np.unique(rng.integers(-2**62, 2**62, n, dtype=np.int64) & ~np.int64(0x33333333))
    taxa = rng.integers(1, mt + 1, len(keys)).astype(np.int32)
    outgrow = case % 3 == 2
    expected = max(16, int(len(keys) * rng.uniform(0.15, 0.7))) if outgrow else len(keys)
    ix = slacken_amd.Index(expected_records=expected, max_taxon=mt, load_factor=lf)
    b0 = ix.info().buckets
    try:
        if outgrow:
            cuts = np.sort(rng.integers(0, len(keys), 3))
            for a, b in zip(np.concatenate([[0], cuts]), np.concatenate([cuts, [len(keys)]])):
                ix.append(keys[a:b], taxa[a:b])
            grown += ix.info().buckets > b0
        else:
            ix.append(keys, taxa)
        info = ix.info()
        assert info.records == len(keys) and info.duplicate_keys == 0, (info.records, len(keys), info.duplicate_keys)
        if case % 4 < 2:
            parents = np.zeros(mt + 1, np.int32)
            parents[2:] = 1
            ix.set_taxonomy(parents)     # (ids beyond 22 bits: renumbered at finalize)
        ix.finalize()
        sample = rng.choice(len(keys), min(len(keys), 20000), replace=False)
        assert np.array_equal(ix.lookup(keys[sample]), taxa[sample]), "lookup of present keys"
        absent = np.setdiff1d(keys[sample] ^ np.int64(1 << 40), keys)
        assert not ix.lookup(absent).any(), "lookup of absent keys"
        if len(keys) < 300000:
            gk, gt = ix.export()
            assert np.array_equal(gk, keys) and np.array_equal(gt, taxa), "export"
    except Exception as e:
        bad += 1
        print("case", case, "n", len(keys), "max_taxon", mt, "load", lf, "buckets", ix.info().buckets, "->", e)
    ix.close()
print("done, failures:", bad, "tables that grew:", grown)
