// parquet_source.cpp -- see parquet_source.hpp.  Built with -std=c++20 (Arrow 25's headers need std::span) and only when
// the Makefile finds pyarrow's bundled libparquet / libarrow; otherwise the stubs at the bottom are compiled.
#include "parquet_source.hpp"

#include <algorithm>
#include <filesystem>
#include <stdexcept>
#include <vector>

#ifdef SLK_HAVE_PARQUET
#include <parquet/api/reader.h>
#include <parquet/statistics.h>

namespace slk_host {

static std::vector<std::string> parquet_files(const std::string &dir) {
  std::vector<std::string> out;
  if (!std::filesystem::is_directory(dir)) return out;
  for (auto &e : std::filesystem::directory_iterator(dir)) {
    std::string p = e.path().string();
    if (e.is_regular_file() && p.size() > 8 && p.compare(p.size() - 8, 8, ".parquet") == 0) out.push_back(p);
  }
  std::sort(out.begin(), out.end());
  return out;
}

bool parquet_available() { return true; }

static void check_schema(const parquet::SchemaDescriptor *schema, const std::string &file, int W, int *c_id, int *c_tax) {
  if (schema->ColumnIndex("id" + std::to_string(W + 1)) >= 0)
    throw std::runtime_error(file + ": more id columns than the index parameters imply (" + std::to_string(W) + ")");
  for (int i = 0; i < W; i++) {
    c_id[i] = schema->ColumnIndex("id" + std::to_string(i + 1));
    if (c_id[i] < 0) throw std::runtime_error(file + ": expected columns id1..id" + std::to_string(W) + " and taxon");
    if (schema->Column(c_id[i])->physical_type() != parquet::Type::INT64) throw std::runtime_error(file + ": id columns must be int64");
  }
  *c_tax = schema->ColumnIndex("taxon");
  if (*c_tax < 0 || schema->Column(*c_tax)->physical_type() != parquet::Type::INT32) throw std::runtime_error(file + ": expected taxon: int32");
}

uint64_t parquet_count_rows(const std::string &dir, int W, int64_t *max_taxon) {
  uint64_t n = 0;
  int64_t mt = 0;
  bool have_stats = true;
  auto files = parquet_files(dir);
  if (files.empty()) throw std::runtime_error("no *.parquet under " + dir);
  for (auto &f : files) {
    auto reader = parquet::ParquetFileReader::OpenFile(f, false);
    auto md = reader->metadata();
    int c_id[4], c_tax;
    check_schema(md->schema(), f, W, c_id, &c_tax);
    n += (uint64_t)md->num_rows();
    for (int g = 0; g < md->num_row_groups(); g++) {
      auto cc = md->RowGroup(g)->ColumnChunk(c_tax);
      auto st = cc->is_stats_set() ? cc->statistics() : nullptr;
      if (st && st->HasMinMax()) mt = std::max<int64_t>(mt, static_cast<const parquet::Int32Statistics *>(st.get())->max());
      else if (md->RowGroup(g)->num_rows() > 0) have_stats = false;
    }
  }
  if (max_taxon) *max_taxon = have_stats ? mt : -1;
  return n;
}

template <class Reader, class T>
static int64_t read_values(Reader *r, int64_t want, std::vector<int16_t> &def, T *out) {
  int64_t values = 0;
  int64_t levels = r->ReadBatch(want, def.data(), nullptr, out, &values);
  if (levels != values) throw std::runtime_error("null values in the record table");
  return values;
}

std::vector<std::string> parquet_list_files(const std::string &dir) { return parquet_files(dir); }

void parquet_read_file(const std::string &f, int W, const std::function<void(const int64_t *, const int32_t *, uint64_t)> &fn) {
  const int64_t B = 1 << 20;
  std::vector<int64_t> col((size_t)B), keys((size_t)B * W);
  std::vector<int32_t> taxa((size_t)B);
  std::vector<int16_t> def((size_t)B);
  auto reader = parquet::ParquetFileReader::OpenFile(f, false);
  auto md = reader->metadata();
  int c_id[4], c_tax;
  check_schema(md->schema(), f, W, c_id, &c_tax);
  for (int g = 0; g < md->num_row_groups(); g++) {
    auto rg = reader->RowGroup(g);
    std::shared_ptr<parquet::ColumnReader> cols[4];
    for (int i = 0; i < W; i++) cols[i] = rg->Column(c_id[i]);
    auto col_tax = rg->Column(c_tax);
    auto *rtx = static_cast<parquet::Int32Reader *>(col_tax.get());
    while (rtx->HasNext()) {
      int64_t nt = 0;
      while (nt < B && rtx->HasNext()) nt += read_values(rtx, B - nt, def, taxa.data() + nt);
      for (int i = 0; i < W; i++) {
        auto *rid = static_cast<parquet::Int64Reader *>(cols[i].get());
        int64_t nk = 0;
        int64_t *dst = W == 1 ? keys.data() : col.data();
        while (nk < nt && rid->HasNext()) nk += read_values(rid, nt - nk, def, dst + nk);
        if (nk != nt) throw std::runtime_error(f + ": id and taxon columns differ in length");
        if (W > 1) for (int64_t r = 0; r < nt; r++) keys[(size_t)r * W + i] = col[(size_t)r];
      }
      fn(keys.data(), taxa.data(), (uint64_t)nt);
    }
  }
}

void parquet_for_each_batch(const std::string &dir, int W, const std::function<void(const int64_t *, const int32_t *, uint64_t)> &fn) {
  for (auto &f : parquet_files(dir)) parquet_read_file(f, W, fn);
}

}  // namespace slk_host

#else

namespace slk_host {
bool parquet_available() { return false; }
uint64_t parquet_count_rows(const std::string &, int, int64_t *) { throw std::runtime_error("built without Parquet support"); }
void parquet_for_each_batch(const std::string &, int, const std::function<void(const int64_t *, const int32_t *, uint64_t)> &) {
  throw std::runtime_error("built without Parquet support");
}
std::vector<std::string> parquet_list_files(const std::string &) { return {}; }
void parquet_read_file(const std::string &, int, const std::function<void(const int64_t *, const int32_t *, uint64_t)> &) {
  throw std::runtime_error("built without Parquet support");
}
}  // namespace slk_host

#endif
