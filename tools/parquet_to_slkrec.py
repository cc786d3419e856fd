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
        f.write(b"SLKREC1\0" + struct.pack("<QII", len(keys), 1, 0))
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


if __name__ == "__main__":
    loc = sys.argv[1]
    keys, taxa = read_parquet_dir(loc)
    write_slkrec(loc + ".slkrec", keys, taxa)
    print(f"{len(keys)} records -> {loc}.slkrec")
