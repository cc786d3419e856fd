#!/usr/bin/env python3
"""A table announced for 6.0e9 records that receives 9.0e9 (slk_index_append_device, 2^27 at a time): it has to grow on the way
(capi.hip: grow_table) -- on the device while both tables fit, through host memory with SLK_GROW_VIA_HOST=1 -- and every record must
be found afterwards (a sample of 2^27 keys from the first, a middle and the last chunk, regenerated from their seeds).  GPU box."""
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
    import bench
    dev = torch.device("cuda", 0)
    announced, total, CH = int(float(os.environ.get("ANNOUNCED", 6e9))), int(float(os.environ.get("TOTAL", 9e9))), 1 << 27
    parents, taxa, leaves = bench.build_taxonomy()
    ix = slacken_amd.Index(k=35, m=31, spaces=7, expected_records=announced, max_taxon=bench.TAX_EXTENT - 1)
    ix.set_taxonomy(parents)
    smask = ((2**62 - 1) & ~0x0CCCCCCC) << 2
    smask = smask - (1 << 64) if smask >= (1 << 63) else smask
    d_taxa = torch.from_numpy(taxa).to(dev)

    def chunk(i, n):
        gen = torch.Generator(device=dev)
        gen.manual_seed(1000 + i)
        keys = ((torch.randint(0, 2**32, (n,), generator=gen, device=dev, dtype=torch.int64) << 32) |
                torch.randint(0, 2**32, (n,), generator=gen, device=dev, dtype=torch.int64)) & smask
        tx = d_taxa[(keys >> 13) % len(taxa)]      # (a function of the key: duplicates among the random keys agree)
        return keys, tx

    b0 = int(ix.info().buckets)
    t0 = time.time()
    nchunks = (total + CH - 1) // CH
    for i in range(nchunks):
        n = min(CH, total - i * CH)
        keys, tx = chunk(i, n)
        torch.cuda.synchronize()
        ix.append_device(keys.data_ptr(), tx.data_ptr(), n)
        del keys, tx
        if i % 16 == 0:
            print(f"[growth] chunk {i}/{nchunks}: {int(ix.info().records)} records, {int(ix.info().buckets)} buckets ({time.time() - t0:.1f}s)", file=sys.stderr, flush=True)
    ix.finalize()
    info = ix.info()
    st = ix.stream()
    bad = 0
    for i in (0, nchunks // 2, nchunks - 1):
        n = min(CH, total - i * CH)
        keys, tx = chunk(i, n)
        out = torch.zeros(n, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        st.lookup_device(keys.data_ptr(), n, out.data_ptr())
        st.synchronize()
        torch.cuda.synchronize()
        bad += int((out != tx).sum().item())
    print(json.dumps(dict(announced=announced, appended=total, records=int(info.records), duplicate_keys=int(info.duplicate_keys),
                          buckets_at_creation=b0, buckets=int(info.buckets), table_GiB=round(info.table_bytes / 2**30, 1), grown=int(info.grown),
                          load=round(float(info.load_factor), 3), max_displacement=int(info.max_displacement), wrong_lookups=bad,
                          via_host=os.environ.get("SLK_GROW_VIA_HOST", "0"), seconds=round(time.time() - t0, 1))))
    assert bad == 0 and info.grown >= 1 and info.records + info.duplicate_keys == total


if __name__ == "__main__":
    main()
