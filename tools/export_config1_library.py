#!/usr/bin/env python3
"""Writes tests/golden/config1_library.npz as a library in Slacken's on-disk layout -- <loc>.properties, <loc>/part-*.parquet (id1:
int64, taxon: int32, snappy), <loc>_taxonomy/{nodes,names}.dmp under the reference's hard-coded test taxonomy
(T/slacken/Testing.scala:147-156) -- so that the REFERENCE can load it (KeyValueIndex.load) and classify the configs[0] read files:
the input of integration/GoldenExport.scala.

  python tools/export_config1_library.py /path/to/reference/testData/slacken/config1_library"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
GOLD = os.path.join(ROOT, "tests", "golden")
NAMES = {455631: ("Clostridioides difficile QCD-66c26", "strain"), 526997: ("Bacillus mycoides DSM 2048", "strain"),
         9606: ("Homo sapiens", "species")}


def main():
    import parquet_to_slkrec as conv
    loc = sys.argv[1]
    exp = json.load(open(os.path.join(GOLD, "config1_expected.json")))
    lib = np.load(os.path.join(GOLD, "config1_library.npz"))
    conv.write_parquet_dir(loc, lib["keys"], lib["taxa"], buckets=4)
    with open(loc + ".properties", "w") as f:
        f.write(f"k={exp['k']}\nm={exp['m']}\nbuckets=4\nversion=1\nsplitter=randomXOR\nminimizerSpaces={exp['spaces']}\ncanonical=true\n")
    os.makedirs(loc + "_taxonomy", exist_ok=True)
    with open(os.path.join(loc + "_taxonomy", "nodes.dmp"), "w") as f:
        f.write("1\t|\t1\t|\tno rank\t|\n")
        for t, (_, rank) in NAMES.items():
            f.write(f"{t}\t|\t1\t|\t{rank}\t|\n")
    with open(os.path.join(loc + "_taxonomy", "names.dmp"), "w") as f:
        f.write("1\t|\troot\t|\t\t|\tscientific name\t|\n")
        for t, (name, _) in NAMES.items():
            f.write(f"{t}\t|\t{name}\t|\t\t|\tscientific name\t|\n")
    print(f"wrote {loc}.properties, {loc}/ ({len(lib['keys'])} records), {loc}_taxonomy/")


if __name__ == "__main__":
    main()
