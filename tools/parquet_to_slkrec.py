#!/usr/bin/env python3
"""Converts a Slacken library's Parquet record table (<idx>/part-*.parquet, columns id1: int64, taxon: int32;
KeyValueIndex.writeRecords, S/slacken/KeyValueIndex.scala:125-139) into the flat <idx>.slkrec file the C++ host reads,
and back (--to-parquet) for writing format-faithful test libraries.   usage: parquet_to_slkrec.py IDX [--to-parquet N]"""
import glob
import os
import struct
import sys

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq


def write_slkrec(path, keys, taxa):
    keys = np.ascontiguousarray(keys, np.int64)
    taxa = np.ascontiguousarray(taxa, np.int32)
    with open(path, "wb") as f:
        f.write(b"SLKREC1\0" + struct.pack("<QII", len(keys), 1, int(taxa.max()) if len(taxa) else 0))
        f.write(keys.tobytes())
        f.write(taxa.tobytes())


def read_parquet_dir(location):
    files = sorted(glob.glob(os.path.join(location, "*.parquet")))
    if not files:
        raise SystemExit(f"no *.parquet under {location}")
    keys, taxa = [], []
    for fn in files:
        t = pq.read_table(fn)
        if "id2" in t.column_names:
            raise SystemExit("this engine supports minimizers up to 32 nt (one id column)")
        keys.append(t.column("id1").to_numpy())
        taxa.append(t.column("taxon").to_numpy())
    return np.concatenate(keys), np.concatenate(taxa)


def write_parquet_dir(location, keys, taxa, buckets):
    """bucketed like Spark's bucketBy would name them (the assignment itself is a Spark internal and irrelevant for reading)"""
    os.makedirs(location, exist_ok=True)
    order = np.arange(len(keys)) % buckets
    for b in range(buckets):
        sel = order == b
        tab = pa.table({"id1": pa.array(keys[sel], pa.int64()), "taxon": pa.array(taxa[sel], pa.int32())})
        pq.write_table(tab, os.path.join(location, f"part-00000-test_{b:05d}.c000.snappy.parquet"), compression="snappy")


def convert(location, batch_rows=1 << 22):
    """Streams <location>/*.parquet into <location>.slkrec one record batch at a time (a standard library is ~1e10 rows)."""
    files = sorted(glob.glob(os.path.join(location, "*.parquet")))
    if not files:
        raise SystemExit(f"no *.parquet under {location}")
    n = 0
    for fn in files:
        pf = pq.ParquetFile(fn)
        if "id2" in pf.schema_arrow.names:
            raise SystemExit("this engine supports minimizers up to 32 nt (one id column)")
        n += pf.metadata.num_rows
    pos, max_taxon = 0, 0
    with open(location + ".slkrec", "wb") as f:
        f.write(b"SLKREC1\0" + struct.pack("<QII", n, 1, 0))
        for fn in files:
            for batch in pq.ParquetFile(fn).iter_batches(batch_size=batch_rows, columns=["id1", "taxon"]):
                k = np.ascontiguousarray(batch.column("id1").to_numpy(zero_copy_only=False), np.int64)
                t = np.ascontiguousarray(batch.column("taxon").to_numpy(zero_copy_only=False), np.int32)
                f.seek(24 + pos * 8)
                f.write(k.tobytes())
                f.seek(24 + n * 8 + pos * 4)
                f.write(t.tobytes())
                pos += len(k)
                if len(t):
                    max_taxon = max(max_taxon, int(t.max()))
        assert pos == n
        f.seek(20)
        f.write(struct.pack("<I", max_taxon))
    return n


if __name__ == "__main__":
    loc = sys.argv[1]
    print(f"{convert(loc)} records -> {loc}.slkrec")
