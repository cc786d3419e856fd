// parquet_source.hpp -- the library's record table read straight from Slacken's Parquet files (<idx>/*.parquet, columns
// id1: int64, taxon: int32; KeyValueIndex.writeRecords / loadRecords, S/slacken/KeyValueIndex.scala:125-159).
// Implemented in parquet_source.cpp against the Arrow C++ libraries that ship inside the pyarrow wheel (this image has no
// Arrow development package); when those are absent at build time the functions report "unavailable" and the host falls
// back to the flat <idx>.slkrec written by tools/parquet_to_slkrec.py.
#pragma once
#include <cstdint>
#include <functional>
#include <string>
#include <vector>

namespace slk_host {

bool parquet_available();
// Row count of all *.parquet files under dir (footers only) and the largest taxon according to the column statistics
// (-1 if some file carries none).  id_columns = number of id columns the index parameters imply (id1..idN, ceil(m / 32)).
// Throws std::runtime_error on unreadable files or a schema that does not match.
uint64_t parquet_count_rows(const std::string &dir, int id_columns, int64_t *max_taxon);
// The *.parquet files under dir, sorted; and the records of one of them (for decoding several files on several threads: a
// standard library is ~2000 bucket files, 120 GB, and snappy + dictionary decoding runs at about 1 GB/s per core).
std::vector<std::string> parquet_list_files(const std::string &dir);
void parquet_read_file(const std::string &file, int id_columns,
                       const std::function<void(const int64_t *, const int32_t *, uint64_t)> &fn);
// Streams the records in batches: keys are rows of id_columns words.
void parquet_for_each_batch(const std::string &dir, int id_columns,
                            const std::function<void(const int64_t *, const int32_t *, uint64_t)> &fn);

}  // namespace slk_host
