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
    """keys: int64 [n] (one id column) or [n, W] (id1..idW)"""
    keys = np.ascontiguousarray(keys, np.int64)
    taxa = np.ascontiguousarray(taxa, np.int32)
    W = 1 if keys.ndim == 1 else keys.shape[1]
    with open(path, "wb") as f:
        f.write(b"SLKREC1\0" + struct.pack("<QII", len(taxa), W, int(taxa.max()) if len(taxa) else 0))
        f.write(keys.tobytes())
        f.write(taxa.tobytes())


def read_parquet_dir(location):
    files = sorted(glob.glob(os.path.join(location, "*.parquet")))
    if not files:
        raise SystemExit(f"no *.parquet under {location}")
    keys, taxa = [], []
    for fn in files:
        t = pq.read_table(fn)
        ids = sorted((c for c in t.column_names if c.startswith("id")), key=lambda c: int(c[2:]))
        k = np.stack([t.column(c).to_numpy() for c in ids], axis=1)
        keys.append(k[:, 0] if len(ids) == 1 else k)
        taxa.append(t.column("taxon").to_numpy())
    return np.concatenate(keys), np.concatenate(taxa)


def write_parquet_dir(location, keys, taxa, buckets):
    """bucketed like Spark's bucketBy would name them (the assignment itself is a Spark internal and irrelevant for reading)"""
    os.makedirs(location, exist_ok=True)
    keys = np.asarray(keys, np.int64)
    k2 = keys.reshape(len(taxa), -1)
    order = np.arange(len(taxa)) % buckets
    for b in range(buckets):
        sel = order == b
        cols = {f"id{i + 1}": pa.array(k2[sel, i], pa.int64()) for i in range(k2.shape[1])}
        cols["taxon"] = pa.array(np.asarray(taxa)[sel], pa.int32())
        tab = pa.table(cols)
        pq.write_table(tab, os.path.join(location, f"part-00000-test_{b:05d}.c000.snappy.parquet"), compression="snappy")


def convert(location, batch_rows=1 << 22):
    """Streams <location>/*.parquet into <location>.slkrec one record batch at a time (a standard library is ~1e10 rows)."""
    files = sorted(glob.glob(os.path.join(location, "*.parquet")))
    if not files:
        raise SystemExit(f"no *.parquet under {location}")
    n = 0
    ids = None
    for fn in files:
        pf = pq.ParquetFile(fn)
        these = sorted((c for c in pf.schema_arrow.names if c.startswith("id")), key=lambda c: int(c[2:]))
        if ids is not None and these != ids:
            raise SystemExit(f"{fn}: id columns {these} differ from {ids}")
        ids = these
        n += pf.metadata.num_rows
    W = len(ids)
    pos, max_taxon = 0, 0
    with open(location + ".slkrec", "wb") as f:
        f.write(b"SLKREC1\0" + struct.pack("<QII", n, W, 0))
        for fn in files:
            for batch in pq.ParquetFile(fn).iter_batches(batch_size=batch_rows, columns=ids + ["taxon"]):
                k = np.ascontiguousarray(np.stack([batch.column(c).to_numpy(zero_copy_only=False) for c in ids], axis=1), np.int64)
                t = np.ascontiguousarray(batch.column("taxon").to_numpy(zero_copy_only=False), np.int32)
                f.seek(24 + pos * 8 * W)
                f.write(k.tobytes())
                f.seek(24 + n * 8 * W + pos * 4)
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
