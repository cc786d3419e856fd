// capi.hip -- implementation of the C ABI declared in include/slacken_amd.h on top of the kernels in kernels.hip.
// Host-side only: handle management, HBM table sizing, per-stream scratch, H2D/D2H for the host-pointer entry points.
// There is NO CPU fallback: without a gfx950 device every compute entry point fails with SLK_E_NO_GPU / SLK_E_HIP.
#include "hostside.h"
#include "../host/pack.hpp"

extern "C" {

const char *slk_last_error(void) { return g_err.c_str(); }
const char *slk_version(void) { return "slacken_amd 0.1 (gfx950)"; }

// Pinned host memory: buffers the host entry points can DMA from and to directly, without the staging copy.
int32_t slk_host_alloc(size_t bytes, void **out) {
  if (!out) return fail(SLK_E_INVALID, "null argument");
  *out = nullptr;
  void *p = nullptr;
  HIPCHK(hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault));
  pinned().add(p, bytes ? bytes : 1, true);
  *out = p;
  return SLK_OK;
}
int32_t slk_host_register(void *ptr, size_t bytes) {
  if (!ptr || !bytes) return fail(SLK_E_INVALID, "null argument");
  HIPCHK(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
  pinned().add(ptr, bytes, false);
  return SLK_OK;
}
int32_t slk_host_free(void *ptr) {  // memory of slk_host_alloc is freed, memory of slk_host_register is unpinned
  if (!ptr) return SLK_OK;
  bool owned = false;
  if (!pinned().remove(ptr, &owned)) return fail(SLK_E_INVALID, "not a pointer of slk_host_alloc / slk_host_register");
  if (owned) HIPCHK(hipHostFree(ptr));
  else HIPCHK(hipHostUnregister(ptr));
  return SLK_OK;
}

int32_t slk_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

static int ceil_log2_u64(uint64_t x) {
  int b = 0;
  while (b < 63 && (1ULL << b) < x) b++;
  return b;
}

// Geometry of the record table.  Any number of buckets (engine.h: the multiply-shift range reduction); a cell holds
//   [flag] remainder (64 - q, + 1 unless the count is a power of two) | displacement | taxon     in 64 bits,
// so the displacement field gets what the other fields leave (8 bits at most are used: 255 buckets of linear probing; a table
// filled to 0.8 has chains of over 63 full buckets), and the buckets' "a record went past" flag exists where a bit is left for it.
struct TableShape { uint64_t nb; int q, disp; bool flag; };
static const int DISP_MIN = CELLS == 16 ? 3 : 4;
static TableShape shape_of(uint64_t nb, int tb) {
  TableShape sh{std::max<uint64_t>(nb, 32), 0, 0, false};
  sh.q = ceil_log2_u64(sh.nb);
  const bool pow2 = sh.nb == (1ULL << sh.q);
  const int avail = 64 - tb - (64 - sh.q + (pow2 ? 0 : 1));
  static const bool no_flag = getenv("SLK_NO_BUCKET_FLAG") != nullptr && getenv("SLK_NO_BUCKET_FLAG")[0] == '1';   // (A/B switch)
  sh.flag = avail - 1 >= DISP_MIN && !no_flag;
  sh.disp = std::min(8, avail - (sh.flag ? 1 : 0));
  return sh;
}
static uint64_t grow_buckets(uint64_t nb) { const int q = ceil_log2_u64(nb); return nb == (1ULL << q) ? nb * 2 : (1ULL << q); }
// displacement bits a table filled to `load` needs: the chains of full buckets grow with the load (measured maxima at 1e5..1e10
// records: load 0.55: 14 buckets, 0.70: 32, 0.80: over 63)
static int need_disp_bits(double load) { return load <= 0.50 ? 4 : load <= 0.62 ? 5 : load <= 0.72 ? 6 : load <= 0.80 ? 7 : 8; }
// `records` records in at least `nb` buckets: the table is made larger (to the next power of two: one bit back from the remainder)
// until its cells leave a displacement field long enough for the load it will then have.  ok = false: no such table below 2^32 buckets.
static TableShape settle_shape(uint64_t nb, uint64_t records, int tb, bool *ok) {
  TableShape sh = shape_of(nb, tb);
  while (sh.disp < DISP_MIN && sh.nb < (1ULL << 33)) sh = shape_of(grow_buckets(sh.nb), tb);
  while (sh.nb < (1ULL << 32) && sh.disp < std::max(DISP_MIN, need_disp_bits((double)records / ((double)sh.nb * CELLS)))) sh = shape_of(grow_buckets(sh.nb), tb);
  *ok = !(sh.nb > (1ULL << 32) || sh.disp < DISP_MIN);
  return sh;
}

int32_t slk_index_create(const slk_params *p, const slk_table_config *cfg, int32_t device, slk_index **out) {
  if (!p || !cfg || !out) return fail(SLK_E_INVALID, "null argument");
  *out = nullptr;
  if (p->m < 1 || p->k < p->m || p->spaces < 0 || p->spaces > p->m / 2)
    return fail(SLK_E_INVALID, "invalid splitter parameters k=%d m=%d spaces=%d", p->k, p->m, p->spaces);
  const int W = (p->m + 31) / 32;
  if (W > WIDE_MAXW) return fail(SLK_E_UNSUPPORTED, "minimizer width m=%d: at most %d nt (%d id columns)", p->m, 32 * WIDE_MAXW, WIDE_MAXW);
  if (p->id_longs != W) return fail(SLK_E_INVALID, "id_longs=%d but m=%d needs %d id columns", p->id_longs, p->m, W);
  if (p->k - p->m + 1 > 512) return fail(SLK_E_UNSUPPORTED, "k - m + 1 = %d > 512 m-mers per window", p->k - p->m + 1);
  if (W > 1 && (p->k - p->m + 1) * W > 128)
    return fail(SLK_E_UNSUPPORTED, "k - m + 1 = %d m-mers per window with %d id columns: at most %d", p->k - p->m + 1, W, 128 / W);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(SLK_E_NO_GPU, "no HIP device available; this engine has no CPU fallback");
  if (device < 0 || device >= ndev) return fail(SLK_E_INVALID, "device %d out of range (%d devices)", device, ndev);
  HIPCHK(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(SLK_E_NO_GPU, "device %d is %s; this library holds gfx950 (MI355X) code objects only", device,
                prop.gcnArchName);

  slk_index *ix = new slk_index();
  ix->device = device;
  ix->params = *p;
  ScanParams &sp = ix->sp;
  sp.k = p->k; sp.m = p->m; sp.w = p->k - p->m + 1; sp.canonical = p->canonical ? 1 : 0;
  sp.sh = W == 1 ? (32 - p->m) * 2 : 0;   // (the one-word fields are unused with several id columns)
  sp.keep = (sp.sh == 0) ? ~0ULL : (~0ULL << sp.sh);
  // RandomXOR.mask (MinimizerPriorities.scala:146-160): one word; partial word => xorMask << (64 - (m%32)*2)
  sp.xmask = (p->m % 32 != 0) ? (p->xor_mask << (64 - (p->m % 32) * 2)) : p->xor_mask;
  // SpacedSeed.spaceMask (:285-300): fill(-1, m), then s times { <<= 4 ; |= 3 << (64 - (m%32)*2) }
  uint64_t sm = sp.keep;
  uint64_t finalBits = 3ULL << ((64 - (p->m % 32) * 2) & 63);
  for (int i = 0; i < p->spaces; i++) sm = (sm << 4) | finalBits;
  sp.smask = sm;

  if (W > 1) {
    // several id columns: the staged kernels of wide.hip over an open-addressing table of (W key words, taxon) slots
    ix->W = W;
    WideParams &wp = ix->wp;
    wp.k = p->k; wp.m = p->m; wp.w = p->k - p->m + 1; wp.canonical = p->canonical ? 1 : 0; wp.W = W;
    wp.last_sh = ((32 - p->m % 32) % 32) * 2;
    for (int i = 0; i < W; i++) {   // RandomXOR.mask :146-160; NTBitArray.fill(-1, m) for the space mask
      wp.xmask[i] = (i == W - 1 && p->m % 32 != 0) ? (p->xor_mask << (64 - (p->m % 32) * 2)) : p->xor_mask;
      wp.smask[i] = ~0ULL;
    }
    if (wp.last_sh) wp.smask[W - 1] = ~0ULL << wp.last_sh;
    const uint64_t fb = 3ULL << ((64 - (p->m % 32) * 2) & 63);
    for (int s = 0; s < p->spaces; s++) {   // SpacedSeed.spaceMask :285-300: s times { <<= 4 over all words ; |= finalBits }
      for (int i = 0; i < W; i++) wp.smask[i] = (wp.smask[i] << 4) | (i + 1 < W ? wp.smask[i + 1] >> 60 : 0);
      wp.smask[W - 1] |= fb;
    }
    uint64_t cap = 1ULL << ceil_log2_u64(std::max<uint64_t>(cfg->expected_records, 8) * 2);
    ix->wt.mask = cap - 1;
    ix->taxon_bits = 31;
    hipError_t e1 = hipMalloc((void **)&ix->wt.keys, cap * W * 8);
    hipError_t e2 = e1 == hipSuccess ? hipMalloc((void **)&ix->wt.taxa, cap * 4) : e1;
    if (e2 != hipSuccess) {
      (void)hipGetLastError();
      if (ix->wt.keys) (void)hipFree(ix->wt.keys);
      delete ix;
      return fail(SLK_E_HIP, "hipMalloc of the %llu-slot table failed: %s", (unsigned long long)cap, hipGetErrorString(e2));
    }
    HIPCHK(hipStreamCreate(&ix->build_stream));
    HIPCHK(hipMemsetAsync(ix->wt.taxa, 0, cap * 4, ix->build_stream));
    HIPCHK(hipMalloc((void **)&ix->d_max_disp, sizeof(int32_t)));
    HIPCHK(hipMalloc((void **)&ix->d_counters, 3 * sizeof(unsigned long long)));
    HIPCHK(hipMemsetAsync(ix->d_max_disp, 0, sizeof(int32_t), ix->build_stream));
    HIPCHK(hipMemsetAsync(ix->d_counters, 0, 3 * sizeof(unsigned long long), ix->build_stream));
    HIPCHK(hipStreamSynchronize(ix->build_stream));
    ix->nbuckets = cap;
    *out = ix;
    return SLK_OK;
  }
  int32_t max_taxon = cfg->max_taxon > 0 ? cfg->max_taxon : ((1 << 22) - 1);
  int tb = 1;
  while (tb < 31 && (1LL << tb) <= (long long)max_taxon) tb++;
  // Load factor.  Given: as given (at most 0.95).  Default: the table takes the memory the device has.  Filled to 0.55 while that
  // costs at most 55 % of the HBM; then fuller, up to 0.70, at that size; then 0.70 with a larger table, up to 80 % of the HBM
  // (2.0e10 records on a 288 GB part: 229 GB); beyond that fuller again, 0.85 at most.  Measured at 1.0e10 records, 64-byte
  // buckets (profiles/r03_bucket_geometry.txt): load 0.45 1 124 M reads/s, 0.55 1 118, 0.70 1 028 -- what a fuller table costs is
  // second-bucket probes.
  const bool default_lf = !(cfg->load_factor > 0);
  const uint64_t expected = std::max<uint64_t>(cfg->expected_records, 1);
  double lf = cfg->load_factor;
  if (default_lf) {
    // (the memory that is FREE now, not the part's total: several tables may share a device -- `--shard-table --devices 0,0`, a
    //  dynamic library beside its base -- and each then takes its share of what the earlier ones left)
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || total_b == 0) { (void)hipGetLastError(); free_b = total_b = (size_t)288 << 30; }
    const double cell_bytes = (double)expected * 8.0, t = (double)std::min(total_b, free_b + ((size_t)2 << 30));
    if (cell_bytes / 0.55 <= 0.55 * t) lf = 0.55;
    else if (cell_bytes / 0.70 <= 0.55 * t) lf = cell_bytes / (0.55 * t);
    else if (cell_bytes / 0.70 <= 0.80 * t) lf = 0.70;
    else lf = std::min(0.85, cell_bytes / (0.80 * t));
  }
  if (lf > 0.95) lf = 0.95;
  // (a record that finds no cell within reach of its displacement field all the same makes the table grow: grow_table)
  const uint64_t cells_needed = (uint64_t)((double)expected / lf) + CELLS;
  bool shape_ok = false;
  const TableShape sh = settle_shape((cells_needed + CELLS - 1) / CELLS, expected, tb, &shape_ok);
  if (!shape_ok) { delete ix; return fail(SLK_E_CAPACITY, "a table of %llu buckets is too large", (unsigned long long)sh.nb); }
  ix->load_target = (float)lf;
  ix->bucket_bits = sh.q;
  ix->taxon_bits = tb;
  ix->disp_bits = sh.disp;
  ix->bucket_flag = sh.flag;
  ix->nbuckets = sh.nb;
  size_t bytes = (size_t)ix->nbuckets * CELLS * 8;
  hipError_t e = hipMalloc((void **)&ix->cells, bytes);
  if (e != hipSuccess) {
    delete ix;
    return fail(SLK_E_HIP, "hipMalloc of %zu table bytes failed: %s", bytes, hipGetErrorString(e));
  }
  HIPCHK(hipStreamCreate(&ix->build_stream));
  HIPCHK(hipMemsetAsync(ix->cells, 0, bytes, ix->build_stream));
  HIPCHK(hipMalloc((void **)&ix->d_max_disp, sizeof(int32_t)));
  HIPCHK(hipMalloc((void **)&ix->d_counters, 3 * sizeof(unsigned long long)));
  HIPCHK(hipMemsetAsync(ix->d_max_disp, 0, sizeof(int32_t), ix->build_stream));
  HIPCHK(hipMemsetAsync(ix->d_counters, 0, 3 * sizeof(unsigned long long), ix->build_stream));
  HIPCHK(hipStreamSynchronize(ix->build_stream));
  *out = ix;
  return SLK_OK;
}

static int32_t read_build_counters(slk_index *ix) {
  unsigned long long c[3];
  int32_t md;
  HIPCHK(hipStreamSynchronize(ix->build_stream));
  HIPCHK(hipMemcpy(c, ix->d_counters, sizeof(c), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(&md, ix->d_max_disp, sizeof(md), hipMemcpyDeviceToHost));
  ix->records = c[0];
  ix->dups = c[1];
  ix->max_disp = md;
  ix->unplaced = c[2];
  if (c[2] != 0 && ix->W > 1) return fail(SLK_E_CAPACITY, "%llu records found no free slot: raise expected_records", c[2]);
  return SLK_OK;   // (one-word table: records that found no cell within reach are the caller's to settle -- insert_growing)
}

static TableBuild build_view(slk_index *ix) {
  TableBuild t;
  t.cells = ix->cells;
  t.g = ix->geom();
  t.disp_limit = (1 << ix->disp_bits) - 1;
  t.shard = ix->shard;
  t.n_shards = ix->n_shards;
  t.max_disp = ix->d_max_disp;
  t.n_inserted = ix->d_counters;
  t.n_duplicate = ix->d_counters + 1;
  t.n_overflow = ix->d_counters + 2;
  return t;
}

// A record that found no cell within reach of its cells' displacement field -- a chain of full buckets longer than the field can
// count; the sizing keeps that from happening at the loads it chooses, a load_factor given by the caller or a library that outgrew
// its expected_records may not -- does NOT fail the load (a library is hours of Parquet streaming by then): the table moves to
// the next larger geometry (twice the buckets: the load halves and the remainder gives a bit to the displacement), piece by piece
// through a bounded staging buffer -- on the device while both tables fit its memory, through host memory otherwise --, and the
// insert that hit the limit runs again (records it had placed are found again as duplicates of themselves: the caller corrects
// the count).  Replaces KeyValueIndex.loadRecords' "it is a table scan: any size works" (S/slacken/KeyValueIndex.scala:150-159).
static int32_t grow_table(slk_index *ix) {
  bool ok = false;
  const TableShape sh = settle_shape(grow_buckets(ix->nbuckets), std::max<uint64_t>(ix->records, 1), ix->taxon_bits, &ok);
  if (!ok) return fail(SLK_E_CAPACITY, "the table cannot grow beyond %llu buckets", (unsigned long long)ix->nbuckets);
  const size_t new_bytes = (size_t)sh.nb * CELLS * 8;
  const uint64_t CH = (uint64_t)1 << 24;   // buckets per piece (at most 2^27 records: 1.5 GB of staging)
  DevBuf dk, dt, dc;
  HIPCHK(dc.ensure(8));
  uint64_t *new_cells = nullptr;
  // (SLK_GROW_VIA_HOST=1: take the host route although both tables would fit the device -- how the tests reach it)
  const char *via_host = getenv("SLK_GROW_VIA_HOST");
  const bool on_device = !(via_host && via_host[0] == '1') && hipMalloc((void **)&new_cells, new_bytes) == hipSuccess;
  if (!on_device) (void)hipGetLastError();
  const uint64_t cap = std::min<uint64_t>(CH, ix->nbuckets) * CELLS;
  HIPCHK(dk.ensure(cap * 8));
  HIPCHK(dt.ensure(cap * 4));
  // scratch counters for the move (the index's own keep counting the caller's records); the new maximum displacement is the move's
  unsigned long long *d_scratch = nullptr;
  HIPCHK(hipMalloc((void **)&d_scratch, 3 * sizeof(unsigned long long)));
  HIPCHK(hipMemsetAsync(d_scratch, 0, 3 * sizeof(unsigned long long), ix->build_stream));
  std::vector<int64_t> h_keys;
  std::vector<int32_t> h_taxa;
  const TableView old_view = [&] { TableView v = ix->view(); v.to_orig = nullptr; return v; }();
  const uint64_t old_nb = ix->nbuckets;
  uint64_t *old_cells = ix->cells;
  auto adopt = [&](uint64_t *cells) {
    ix->cells = cells; ix->nbuckets = sh.nb; ix->bucket_bits = sh.q; ix->disp_bits = sh.disp; ix->bucket_flag = sh.flag;
  };
  auto insert_piece = [&](const int64_t *k, const int32_t *t, uint64_t n) -> int32_t {
    TableBuild nb = build_view(ix);
    nb.shard = 0; nb.n_shards = 0;   // (what is in the table is this shard's already)
    nb.n_inserted = d_scratch; nb.n_duplicate = d_scratch + 1; nb.n_overflow = d_scratch + 2;
    launch_table_insert(nb, k, t, n, ix->build_stream);
    HIPCHK(hipGetLastError());
    return SLK_OK;
  };
  if (on_device) {
    HIPCHK(hipMemsetAsync(new_cells, 0, new_bytes, ix->build_stream));
    HIPCHK(hipMemsetAsync(ix->d_max_disp, 0, sizeof(int32_t), ix->build_stream));
    adopt(new_cells);
    for (uint64_t b0 = 0; b0 < old_nb; b0 += CH) {
      const uint64_t b1 = std::min(old_nb, b0 + CH);
      unsigned long long n = 0;
      HIPCHK(hipMemsetAsync(dc.p, 0, 8, ix->build_stream));
      launch_export_range(old_view, b0, b1, dk.as<int64_t>(), dt.as<int32_t>(), cap, dc.as<unsigned long long>(), ix->build_stream);
      HIPCHK(hipGetLastError());
      HIPCHK(hipMemcpyAsync(&n, dc.p, 8, hipMemcpyDeviceToHost, ix->build_stream));
      HIPCHK(hipStreamSynchronize(ix->build_stream));
      int32_t rc = insert_piece(dk.as<int64_t>(), dt.as<int32_t>(), n);
      if (rc) return rc;
    }
    HIPCHK(hipStreamSynchronize(ix->build_stream));
    HIPCHK(hipFree(old_cells));
  } else {
    // both tables do not fit the device: the records wait in host memory (12 bytes each) while the old table makes room
    for (uint64_t b0 = 0; b0 < old_nb; b0 += CH) {
      const uint64_t b1 = std::min(old_nb, b0 + CH);
      unsigned long long n = 0;
      HIPCHK(hipMemsetAsync(dc.p, 0, 8, ix->build_stream));
      launch_export_range(old_view, b0, b1, dk.as<int64_t>(), dt.as<int32_t>(), cap, dc.as<unsigned long long>(), ix->build_stream);
      HIPCHK(hipGetLastError());
      HIPCHK(hipMemcpyAsync(&n, dc.p, 8, hipMemcpyDeviceToHost, ix->build_stream));
      HIPCHK(hipStreamSynchronize(ix->build_stream));
      const size_t at = h_keys.size();
      h_keys.resize(at + n); h_taxa.resize(at + n);
      if (n) {
        HIPCHK(hipMemcpy(h_keys.data() + at, dk.p, n * 8, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(h_taxa.data() + at, dt.p, n * 4, hipMemcpyDeviceToHost));
      }
    }
    HIPCHK(hipFree(old_cells));
    ix->cells = nullptr;
    if (hipMalloc((void **)&new_cells, new_bytes) != hipSuccess) {
      (void)hipGetLastError();
      (void)hipFree(d_scratch);
      return fail(SLK_E_HIP, "hipMalloc of %zu table bytes failed while the table was growing (the records are lost: load the library again "
                  "with a larger slk_table_config.expected_records)", new_bytes);
    }
    HIPCHK(hipMemsetAsync(new_cells, 0, new_bytes, ix->build_stream));
    HIPCHK(hipMemsetAsync(ix->d_max_disp, 0, sizeof(int32_t), ix->build_stream));
    adopt(new_cells);
    for (uint64_t o = 0; o < h_keys.size(); o += cap) {
      const uint64_t n = std::min<uint64_t>(cap, h_keys.size() - o);
      HIPCHK(hipMemcpyAsync(dk.p, h_keys.data() + o, n * 8, hipMemcpyHostToDevice, ix->build_stream));
      HIPCHK(hipMemcpyAsync(dt.p, h_taxa.data() + o, n * 4, hipMemcpyHostToDevice, ix->build_stream));
      int32_t rc = insert_piece(dk.as<int64_t>(), dt.as<int32_t>(), n);
      if (rc) return rc;
      HIPCHK(hipStreamSynchronize(ix->build_stream));
    }
  }
  unsigned long long c[3] = {0, 0, 0};
  HIPCHK(hipMemcpy(c, d_scratch, sizeof(c), hipMemcpyDeviceToHost));
  (void)hipFree(d_scratch);
  dk.release(); dt.release(); dc.release();
  if (c[2] != 0 || c[1] != 0) return fail(SLK_E_HIP, "moving the table to a larger one lost records (%llu unplaced, %llu collided)", c[2], c[1]);
  ix->grown++;
  static const bool verbose = getenv("SLK_DEBUG_GROW") != nullptr;
  if (verbose) fprintf(stderr, "[slk] table grown to %llu buckets (%d displacement bits), %llu records moved %s\n", (unsigned long long)sh.nb, sh.disp,
                       c[0], on_device ? "on the device" : "through host memory");
  return SLK_OK;
}

// Runs `insert` (which queues one batch of records on the build stream; re-runnable) until every record of the batch has a cell,
// moving the table to a larger one in between if need be.  counts_dups: the insert counts keys that are present already (the plain
// record insert; the library builder merges them instead): a re-run then counts the records the first run placed as duplicates of
// themselves, which is taken out again.
static int32_t insert_growing(slk_index *ix, bool counts_dups, const std::function<int32_t()> &insert) {
  int32_t rc = read_build_counters(ix);   // (the state before this batch)
  if (rc) return rc;
  const uint64_t ins0 = ix->records, dup0 = ix->dups;
  for (int attempt = 0;; attempt++) {
    const uint64_t ins_before = ix->records;
    rc = insert();
    if (rc) return rc;
    rc = read_build_counters(ix);
    if (rc) return rc;
    if (ix->unplaced == 0) {
      if (counts_dups && attempt > 0) {
        // this run saw every record of the batch: new ones it inserted, all others it counted -- among them the (ins_before - ins0)
        // records earlier runs had placed
        const uint64_t dups = dup0 + (ix->dups - dup0) - (ins_before - ins0);
        const unsigned long long v = dups;
        HIPCHK(hipMemcpy(ix->d_counters + 1, &v, sizeof(v), hipMemcpyHostToDevice));
        ix->dups = dups;
      }
      return SLK_OK;
    }
    if (ix->W > 1) return fail(SLK_E_CAPACITY, "%llu records found no free slot: raise expected_records", (unsigned long long)ix->unplaced);
    if (attempt >= 6) return fail(SLK_E_CAPACITY, "%llu records could not be placed after the table had grown %d times", (unsigned long long)ix->unplaced, attempt);
    rc = grow_table(ix);
    if (rc) return rc;
    // the next run starts from this batch's beginning: its duplicate count too
    const unsigned long long z[2] = {dup0, 0};
    HIPCHK(hipMemcpy(ix->d_counters + 1, &z[0], sizeof(unsigned long long), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ix->d_counters + 2, &z[1], sizeof(unsigned long long), hipMemcpyHostToDevice));
    ix->dups = dup0;
  }
}

// Table-sharded libraries (BASELINE configs[3]): the index keeps the records whose key falls to `shard` of `n_shards`
// (slk_shard_of) and drops the others where they arrive -- in slk_index_append[_device] and in slk_index_add_sequences[_device] --
// so that every rank can be handed the same record stream or the same genomes.  Before the first record.
int32_t slk_index_set_shard(slk_index *ix, uint32_t shard, uint32_t n_shards) {
  if (!ix) return fail(SLK_E_INVALID, "null argument");
  if (n_shards < 1 || n_shards > 64 || shard >= n_shards) return fail(SLK_E_INVALID, "shard %u of %u", shard, n_shards);
  if (ix->W > 1) return fail(SLK_E_UNSUPPORTED, "the sharded entry points support minimizers of up to 32 nt (one id column)");
  if (ix->finalized || ix->records != 0) return fail(SLK_E_STATE, "slk_index_set_shard must precede the first record");
  ix->shard = shard;
  ix->n_shards = n_shards;
  return SLK_OK;
}

int32_t slk_index_append_device(slk_index *ix, const int64_t *d_keys, const int32_t *d_taxa, uint64_t n) {
  if (!ix || (n && (!d_keys || !d_taxa))) return fail(SLK_E_INVALID, "null argument");
  if (ix->finalized) return fail(SLK_E_STATE, "index is finalized");
  int32_t rc = set_device(ix);
  if (rc) return rc;
  if (ix->W > 1) {
    launch_wide_insert(ix->wt, ix->W, d_keys, d_taxa, n, ix->d_counters, ix->build_stream);
    HIPCHK(hipGetLastError());
    return read_build_counters(ix);
  }
  return insert_growing(ix, true, [&]() -> int32_t {
    launch_table_insert(build_view(ix), d_keys, d_taxa, n, ix->build_stream);
    HIPCHK(hipGetLastError());
    return SLK_OK;
  });
}

int32_t slk_index_append(slk_index *ix, const int64_t *keys, const int32_t *taxa, uint64_t n) {
  if (!ix || (n && (!keys || !taxa))) return fail(SLK_E_INVALID, "null argument");
  if (ix->finalized) return fail(SLK_E_STATE, "index is finalized");
  int32_t rc = set_device(ix);
  if (rc) return rc;
  const uint64_t CH = 1ULL << 24;
  if (ix->W > 1) {  // keys: n rows of W words (id1..idW)
    const int W = ix->W;
    for (uint64_t o = 0; o < n; o += CH) {
      uint64_t c = std::min(CH, n - o);
      for (uint64_t i = 0; i < c; i++)
        if (taxa[o + i] < 0) return fail(SLK_E_INVALID, "record %llu: negative taxon %d", (unsigned long long)(o + i), taxa[o + i]);
      HIPCHK(ix->stage_keys.ensure(c * 8 * W));
      HIPCHK(ix->stage_taxa.ensure(c * 4));
      rc = copy_in(&ix->staging, ix->build_stream, ix->stage_keys.p, keys + o * W, c * 8 * W);
      if (!rc) rc = copy_in(&ix->staging, ix->build_stream, ix->stage_taxa.p, taxa + o, c * 4);
      if (rc) return rc;
      launch_wide_insert(ix->wt, W, ix->stage_keys.as<int64_t>(), ix->stage_taxa.as<int32_t>(), c, ix->d_counters, ix->build_stream);
      HIPCHK(hipGetLastError());
      HIPCHK(hipStreamSynchronize(ix->build_stream));
    }
    return read_build_counters(ix);
  }
  for (uint64_t o = 0; o < n; o += CH) {
    uint64_t c = std::min(CH, n - o);
    HIPCHK(ix->stage_keys.ensure(c * 8));
    HIPCHK(ix->stage_taxa.ensure(c * 4));
    int32_t max_t = (1 << ix->taxon_bits) - 1;
    for (uint64_t i = 0; i < c; i++)
      if (taxa[o + i] < 0 || taxa[o + i] > max_t)
        return fail(SLK_E_INVALID, "record %llu: taxon %d outside [0, %d] (slk_table_config.max_taxon)",
                    (unsigned long long)(o + i), taxa[o + i], max_t);
    rc = copy_in(&ix->staging, ix->build_stream, ix->stage_keys.p, keys + o, c * 8);
    if (!rc) rc = copy_in(&ix->staging, ix->build_stream, ix->stage_taxa.p, taxa + o, c * 4);
    if (rc) return rc;
    rc = insert_growing(ix, true, [&]() -> int32_t {
      launch_table_insert(build_view(ix), ix->stage_keys.as<int64_t>(), ix->stage_taxa.as<int32_t>(), c, ix->build_stream);
      HIPCHK(hipGetLastError());
      return SLK_OK;
    });
    if (rc) return rc;
  }
  return read_build_counters(ix);
}

// The taxonomy as the lane kernel reads it: per id {parent, tin, tout, 0}, tin / tout from a depth-first tour of the forest
// (every id whose parent is NONE is a root: ROOT, unused ids, the top of a detached subtree), so that "a is an ancestor-or-self
// of b" (Taxonomy.hasAncestor, Taxonomy.scala:236-244) is tin[a] <= tin[b] <= tout[a] -- two compares on values that are loaded
// once per taxon of a read's map -- instead of a walk of b's root path: NCBI lineages are 25-40 nodes deep, and resolveTree
// (LowestCommonAncestor.scala:101-146) asks it for every pair of map taxa and again at every step of the confidence walk.
static int32_t build_tax_nodes(const int32_t *parents, int32_t n, int32_t max_n, uint4 **out) {
  *out = nullptr;
  if (n < 2 || n > max_n) return SLK_OK;
  std::vector<uint32_t> first((size_t)n + 1, 0), kids;   // children of p: kids[first[p] .. first[p + 1]), in increasing id order
  for (int32_t t = 1; t < n; t++) if (parents[t] != 0) first[(size_t)parents[t] + 1]++;
  for (int32_t p = 0; p < n; p++) first[(size_t)p + 1] += first[p];
  kids.resize(first[n]);
  {
    std::vector<uint32_t> at(first.begin(), first.end() - 1);
    for (int32_t t = 1; t < n; t++) if (parents[t] != 0) kids[at[parents[t]]++] = (uint32_t)t;
  }
  std::vector<uint4> nodes((size_t)n, make_uint4(0, 0, 0, 0));
  std::vector<std::pair<uint32_t, uint32_t>> stack;   // (node, next child)
  uint32_t clock = 0;
  for (int32_t r = 1; r < n; r++) {
    if (parents[r] != 0) continue;
    stack.emplace_back((uint32_t)r, first[r]);
    nodes[r].y = ++clock;
    while (!stack.empty()) {
      auto &top = stack.back();
      if (top.second < first[(size_t)top.first + 1]) {
        const uint32_t c = kids[top.second++];
        nodes[c].x = top.first;
        nodes[c].y = ++clock;
        stack.emplace_back(c, first[c]);
      } else {
        nodes[top.first].z = clock;   // the largest tin of the subtree
        stack.pop_back();
      }
    }
  }
  HIPCHK(hipMalloc((void **)out, (size_t)n * sizeof(uint4)));
  HIPCHK(hipMemcpy(*out, nodes.data(), (size_t)n * sizeof(uint4), hipMemcpyHostToDevice));
  return SLK_OK;
}
static void free_tax_nodes(slk_index *ix) {
  if (ix->d_nodes_orig && ix->d_nodes_orig != ix->d_nodes) (void)hipFree(ix->d_nodes_orig);
  if (ix->d_nodes) (void)hipFree(ix->d_nodes);
  ix->d_nodes = ix->d_nodes_orig = nullptr;
}

int32_t slk_index_set_taxonomy(slk_index *ix, const int32_t *parents, int32_t T) {
  if (!ix || !parents || T < 2) return fail(SLK_E_INVALID, "taxonomy needs parents[] with at least ROOT (T >= 2)");
  int32_t rc = set_device(ix);
  if (rc) return rc;
  // The reference's parent walks terminate only on a forest (Taxonomy.scala:151-156); reject cycles up front.
  {
    std::vector<uint8_t> state((size_t)T, 0);  // 0 new, 1 on the current path, 2 done
    std::vector<int32_t> path;
    for (int32_t t = 1; t < T; t++) {
      int32_t x = t;
      path.clear();
      while (x != 0 && state[x] == 0) {
        if (parents[x] < 0 || parents[x] >= T) return fail(SLK_E_INVALID, "parents[%d] = %d out of range", x, parents[x]);
        state[x] = 1;
        path.push_back(x);
        x = parents[x];
      }
      if (x != 0 && state[x] == 1) return fail(SLK_E_INVALID, "taxonomy has a cycle through taxon %d", x);
      for (int32_t y : path) state[y] = 2;
    }
  }
  if (ix->D) return fail(SLK_E_STATE, "this finalized index stores dense taxon ids derived from its taxonomy: the taxonomy cannot be replaced");
  if (ix->d_parents) { HIPCHK(hipFree(ix->d_parents)); ix->d_parents = nullptr; }
  HIPCHK(hipMalloc((void **)&ix->d_parents, (size_t)T * sizeof(int32_t)));
  HIPCHK(hipMemcpy(ix->d_parents, parents, (size_t)T * sizeof(int32_t), hipMemcpyHostToDevice));
  ix->T = T;
  ix->h_parents.assign(parents, parents + T);
  // Euler tours: for the fused kernels (ids of at most 22 bits take the lane kernel; wider ones are renumbered at finalize, which
  // builds that tour then) and, in the caller's ids, for the staged classify kernel (up to 2^26 ids: 1 GiB of node records)
  free_tax_nodes(ix);
  int32_t rcn = build_tax_nodes(parents, T, (1 << 22) + 1, &ix->d_nodes);
  if (rcn) return rcn;
  if (ix->d_nodes) { ix->d_nodes_orig = ix->d_nodes; return SLK_OK; }
  return build_tax_nodes(parents, T, 1 << 26, &ix->d_nodes_orig);
}

// Library construction with several id columns (minimizers of 33..128 nt): the staged kernels of wide.hip.  Groups of about 64 MiB
// of sequence are cut into chunks of BUILD_CHUNK_WINDOWS k-mer windows (overlapping by k - 1 bases: the same minimizer SET), the
// chunks are scanned as a batch of fragments, and their SEQUENCE-flag spans are inserted / LCA-merged one lane per span.
static int32_t add_sequences_wide(slk_index *ix, const uint8_t *bases, const uint64_t *offsets, const int32_t *taxa, uint64_t S, bool bases_on_device) {
  const uint32_t k = (uint32_t)ix->wp.k, CW = BUILD_CHUNK_WINDOWS;
  for (uint64_t i = 0; i < S; i++) {
    if (offsets[i + 1] < offsets[i]) return fail(SLK_E_INVALID, "offsets must be non-decreasing (sequence %llu)", (unsigned long long)i);
    if (taxa[i] < 0) return fail(SLK_E_INVALID, "sequence %llu: negative taxon %d", (unsigned long long)i, taxa[i]);
  }
  std::vector<uint8_t> host_copy;
  if (bases_on_device && S) {   // (this path stages its chunks on the host)
    host_copy.resize(offsets[S]);
    HIPCHK(hipMemcpy(host_copy.data(), bases, offsets[S], hipMemcpyDeviceToHost));
    bases = host_copy.data();
  }
  const uint64_t GROUP = 64ULL << 20;
  DevBuf d_bases, d_off, d_tax, d_keys, d_meta, d_count;
  std::vector<uint8_t> cb;
  std::vector<uint64_t> coff;
  std::vector<int32_t> ctax;
  auto flush = [&]() -> int32_t {
    if (ctax.empty()) return SLK_OK;
    const uint64_t nc = ctax.size(), total = cb.size();
    HIPCHK(d_bases.ensure(total + 16));
    HIPCHK(d_off.ensure((nc + 1) * 8));
    HIPCHK(d_tax.ensure(nc * 4));
    HIPCHK(d_keys.ensure((total + 1) * 8 * ix->W));
    HIPCHK(d_meta.ensure((total + 1) * 4));
    HIPCHK(d_count.ensure((nc + 1) * 4));
    int32_t rc = copy_in(&ix->staging, ix->build_stream, d_bases.p, cb.data(), total);
    if (!rc) rc = copy_in(&ix->staging, ix->build_stream, d_off.p, coff.data(), (nc + 1) * 8);
    if (!rc) rc = copy_in(&ix->staging, ix->build_stream, d_tax.p, ctax.data(), nc * 4);
    if (rc) return rc;
    launch_wide_scan(ix->wp, d_bases.as<uint8_t>(), d_off.as<uint64_t>(), nullptr, nullptr, nc, d_keys.as<uint64_t>(), d_meta.as<int32_t>(),
                     d_count.as<int32_t>(), ix->build_stream);
    launch_wide_build_insert(ix->wt, ix->W, ix->d_parents, ix->T, d_off.as<uint64_t>(), nc, d_keys.as<uint64_t>(), d_meta.as<int32_t>(),
                             d_count.as<int32_t>(), d_tax.as<int32_t>(), ix->d_counters, ix->build_stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ix->build_stream));
    cb.clear(); coff.assign(1, 0); ctax.clear();
    return SLK_OK;
  };
  coff.assign(1, 0);
  for (uint64_t q = 0; q < S; q++) {
    const uint64_t len = offsets[q + 1] - offsets[q];
    if (taxa[q] == 0 || len < k) continue;
    const uint64_t windows = len - k + 1;
    for (uint64_t w0 = 0; w0 < windows; w0 += CW) {
      const uint64_t nw = std::min<uint64_t>(CW, windows - w0);
      const uint8_t *src = bases + offsets[q] + w0;
      cb.insert(cb.end(), src, src + nw + k - 1);
      coff.push_back(cb.size());
      ctax.push_back(taxa[q]);
      if (cb.size() >= GROUP) { int32_t rc = flush(); if (rc) return rc; }
    }
  }
  int32_t rc = flush();
  if (rc) return rc;
  d_bases.release(); d_off.release(); d_tax.release(); d_keys.release(); d_meta.release(); d_count.release();
  return read_build_counters(ix);
}

// bases_on_device: `bases` is resident on the index's GPU and is scanned where it lies
static int32_t add_sequences(slk_index *ix, const uint8_t *bases, const uint64_t *offsets, const int32_t *taxa, uint64_t S,
                             bool bases_on_device) {
  if (!ix || (S && (!bases || !offsets || !taxa))) return fail(SLK_E_INVALID, "null argument");
  if (ix->finalized) return fail(SLK_E_STATE, "index is finalized");
  if (!ix->d_parents) return fail(SLK_E_STATE, "slk_index_add_sequences needs the taxonomy (LCA merging): call slk_index_set_taxonomy first");
  if (ix->W == 1 && ix->sp.w > BUILD_MAX_W) return fail(SLK_E_UNSUPPORTED, "library construction supports windows of up to %d m-mers (k - m + 1 = %d)", BUILD_MAX_W, ix->sp.w);
  int32_t rc = set_device(ix);
  if (rc) return rc;
  if (ix->W > 1) return add_sequences_wide(ix, bases, offsets, taxa, S, bases_on_device);
  const int32_t max_t = (int32_t)((1LL << ix->taxon_bits) - 1);
  const uint32_t k = (uint32_t)ix->sp.k, CW = BUILD_CHUNK_WINDOWS;
  for (uint64_t i = 0; i < S; i++) {
    if (offsets[i + 1] < offsets[i]) return fail(SLK_E_INVALID, "offsets must be non-decreasing (sequence %llu)", (unsigned long long)i);
    if (taxa[i] < 0 || taxa[i] > max_t)
      return fail(SLK_E_INVALID, "sequence %llu: taxon %d outside [0, %d] (slk_table_config.max_taxon)", (unsigned long long)i, taxa[i], max_t);
  }
  // groups of whole sequences of about 1 GiB; each is cut into chunks of CW windows overlapping by k-1 bases
  const uint64_t GROUP = 1ULL << 30;
  DevBuf d_bases, d_start, d_len, d_tax;
  std::vector<uint64_t> cstart;
  std::vector<uint32_t> clen;
  std::vector<int32_t> ctax;
  uint64_t i = 0;
  while (i < S) {
    uint64_t j = i, g0 = offsets[i];
    while (j < S && (j == i || offsets[j + 1] - g0 <= GROUP)) j++;
    uint64_t gbytes = offsets[j] - g0;
    cstart.clear(); clen.clear(); ctax.clear();
    for (uint64_t q = i; q < j; q++) {
      uint64_t len = offsets[q + 1] - offsets[q];
      if (taxa[q] == 0 || len < k) continue;
      uint64_t windows = len - k + 1;
      for (uint64_t w0 = 0; w0 < windows; w0 += CW) {
        uint64_t nw = std::min<uint64_t>(CW, windows - w0);
        cstart.push_back(offsets[q] - g0 + w0);
        clen.push_back((uint32_t)(nw + k - 1));
        ctax.push_back(taxa[q]);
      }
    }
    if (!cstart.empty()) {
      uint64_t nc = cstart.size();
      HIPCHK(d_start.ensure(nc * 8));
      HIPCHK(d_len.ensure(nc * 4));
      HIPCHK(d_tax.ensure(nc * 4));
      const uint8_t *src = bases + g0;
      if (!bases_on_device) {
        HIPCHK(d_bases.ensure(gbytes));
        rc = copy_in(&ix->staging, ix->build_stream, d_bases.p, bases + g0, gbytes);
        if (rc) return rc;
        src = d_bases.as<uint8_t>();
      }
      rc = copy_in(&ix->staging, ix->build_stream, d_start.p, cstart.data(), nc * 8);
      if (!rc) rc = copy_in(&ix->staging, ix->build_stream, d_len.p, clen.data(), nc * 4);
      if (!rc) rc = copy_in(&ix->staging, ix->build_stream, d_tax.p, ctax.data(), nc * 4);
      if (rc) return rc;
      // (re-runnable: the merge by LCA is idempotent, so a group that ran into the table's limit is simply scanned again)
      rc = insert_growing(ix, false, [&]() -> int32_t {
        launch_build(ix->sp, build_view(ix), ix->d_parents, ix->T, src, gbytes, d_start.as<uint64_t>(),
                     d_len.as<uint32_t>(), d_tax.as<int32_t>(), nc, ix->build_stream);
        HIPCHK(hipGetLastError());
        return SLK_OK;
      });
      if (rc) return rc;
    }
    i = j;
  }
  d_bases.release(); d_start.release(); d_len.release(); d_tax.release();
  return read_build_counters(ix);
}

int32_t slk_index_add_sequences(slk_index *ix, const uint8_t *bases, const uint64_t *offsets, const int32_t *taxa,
                                uint64_t S) {
  return add_sequences(ix, bases, offsets, taxa, S, false);
}

int32_t slk_index_add_sequences_device(slk_index *ix, const uint8_t *d_bases, const uint64_t *offsets, const int32_t *taxa,
                                       uint64_t S) {
  return add_sequences(ix, d_bases, offsets, taxa, S, true);
}

int32_t slk_index_export(const slk_index *ix, int64_t *keys, int32_t *taxa, uint64_t capacity, uint64_t *n_records) {
  if (!ix || !n_records || (capacity && (!keys || !taxa))) return fail(SLK_E_INVALID, "null argument");
  (void)hipSetDevice(ix->device);
  DevBuf dk, dt, dc;
  HIPCHK(dk.ensure(std::max<uint64_t>(capacity, 1) * 8 * ix->W));
  HIPCHK(dt.ensure(std::max<uint64_t>(capacity, 1) * 4));
  HIPCHK(dc.ensure(8));
  HIPCHK(hipMemset(dc.p, 0, 8));
  if (ix->W > 1) {
    launch_wide_export(ix->wt, ix->W, dk.as<int64_t>(), dt.as<int32_t>(), capacity, dc.as<unsigned long long>(), ix->build_stream);
  } else {
    launch_export(ix->view(), ix->nbuckets, dk.as<int64_t>(), dt.as<int32_t>(), capacity, dc.as<unsigned long long>(), ix->build_stream);
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(ix->build_stream));
  unsigned long long n = 0;
  HIPCHK(hipMemcpy(&n, dc.p, 8, hipMemcpyDeviceToHost));
  *n_records = n;
  uint64_t got = std::min<uint64_t>(n, capacity);
  if (got) {
    HIPCHK(hipMemcpy(keys, dk.p, got * 8 * ix->W, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(taxa, dt.p, got * 4, hipMemcpyDeviceToHost));
  }
  dk.release(); dt.release(); dc.release();
  if (n > capacity && capacity) return fail(SLK_E_CAPACITY, "%llu records, capacity %llu", n, (unsigned long long)capacity);
  return SLK_OK;
}

// Dense taxon ids.  The lane-per-fragment kernel keeps a fragment's taxon -> count map as one LDS word per entry
// (taxon << 10 | count): taxon ids of up to 22 bits.  NCBI's ids pass 2^22 = 4 194 304 within a few releases, but the NODES of
// the taxonomy are far fewer than the id range; so when the caller's ids do not fit, the cells are rewritten once, here, to hold
// the rank of their taxon among the taxonomy's nodes (in increasing id order, ROOT = 1 stays 1), the fused kernels walk a
// parents array in those ranks, and ids are translated back where taxa leave the engine (engine.h: ext_taxon).  Needs the
// taxonomy to be set before finalize and every record's taxon to be one of its nodes; otherwise the ids stay as given and
// fragments take the wave-per-fragment kernel, as before.
static int32_t make_dense_taxa(slk_index *ix) {
  if (ix->W > 1 || ix->taxon_bits <= 22 || ix->h_parents.empty() || ix->D) return SLK_OK;
  const int32_t T = ix->T;
  std::vector<int32_t> to_dense((size_t)T, 0), to_orig(1, 0);
  for (int32_t t = 1; t < T; t++)
    if (t == 1 || ix->h_parents[t] != 0) { to_dense[t] = (int32_t)to_orig.size(); to_orig.push_back(t); }
  const int32_t D = (int32_t)to_orig.size() - 1;
  if (D < 1 || D >= (1 << 22)) return SLK_OK;
  std::vector<int32_t> pd((size_t)D + 1, 0);
  for (int32_t d = 1; d <= D; d++) pd[d] = to_dense[ix->h_parents[to_orig[d]]];  // (parent of ROOT is NONE = 0)
  int32_t *d_td = nullptr, *d_to = nullptr, *d_pd = nullptr;
  unsigned long long *d_bad = nullptr, bad = 0;
  HIPCHK(hipMalloc((void **)&d_td, (size_t)T * 4));
  HIPCHK(hipMalloc((void **)&d_bad, 8));
  HIPCHK(hipMemcpy(d_td, to_dense.data(), (size_t)T * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemset(d_bad, 0, 8));
  launch_remap_cells(ix->cells, ix->nbuckets * CELLS, ix->taxon_bits, d_td, T, d_bad, false, ix->build_stream);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(ix->build_stream));
  HIPCHK(hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost));
  if (bad != 0) {  // records whose taxon is not a node of this taxonomy: keep the ids as they are
    (void)hipFree(d_td); (void)hipFree(d_bad);
    return SLK_OK;
  }
  launch_remap_cells(ix->cells, ix->nbuckets * CELLS, ix->taxon_bits, d_td, T, d_bad, true, ix->build_stream);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(ix->build_stream));
  (void)hipFree(d_bad);
  HIPCHK(hipMalloc((void **)&d_to, ((size_t)D + 1) * 4));
  HIPCHK(hipMalloc((void **)&d_pd, ((size_t)D + 1) * 4));
  HIPCHK(hipMemcpy(d_to, to_orig.data(), ((size_t)D + 1) * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d_pd, pd.data(), ((size_t)D + 1) * 4, hipMemcpyHostToDevice));
  ix->d_to_dense = d_td; ix->d_to_orig = d_to; ix->d_parents_dense = d_pd; ix->D = D;
  if (ix->d_nodes && ix->d_nodes != ix->d_nodes_orig) (void)hipFree(ix->d_nodes);
  if (ix->d_nodes == ix->d_nodes_orig) ix->d_nodes = nullptr;   // (the tour of the ids as given stays with the staged kernel)
  return build_tax_nodes(pd.data(), D + 1, (1 << 22) + 1, &ix->d_nodes);
}

int32_t slk_index_finalize(slk_index *ix) {
  if (!ix) return fail(SLK_E_INVALID, "null argument");
  int32_t rc = set_device(ix);
  if (rc) return rc;
  rc = read_build_counters(ix);
  if (rc) return rc;
  ix->stage_keys.release();
  ix->stage_taxa.release();
  ix->staging.release();
  if (!ix->finalized) {
    rc = make_dense_taxa(ix);
    if (rc) return rc;
  }
  ix->finalized = true;
  return SLK_OK;
}

int32_t slk_index_get_info(const slk_index *ix, slk_index_info *out) {
  if (!ix || !out) return fail(SLK_E_INVALID, "null argument");
  memset(out, 0, sizeof(*out));
  out->records = ix->records;
  out->buckets = ix->nbuckets;
  out->table_bytes = ix->W > 1 ? ix->nbuckets * (8 * ix->W + 4) : ix->nbuckets * CELLS * 8;
  out->bucket_bits = ix->bucket_bits;
  out->taxon_bits = ix->taxon_bits;
  out->disp_bits = ix->disp_bits;
  out->max_displacement = ix->max_disp;
  out->duplicate_keys = ix->dups;
  out->taxonomy_size = ix->T;
  out->device = ix->device;
  out->dense_taxa = ix->D;
  out->bucket_cells = ix->W > 1 ? 1 : CELLS;
  out->load_factor = ix->load_target;
  out->grown = (int32_t)ix->grown;
  return SLK_OK;
}

int32_t slk_index_lookup(const slk_index *ix, const int64_t *keys, uint64_t n, int32_t *out_taxa) {
  if (!ix || (n && (!keys || !out_taxa))) return fail(SLK_E_INVALID, "null argument");
  if (!ix->finalized) return fail(SLK_E_STATE, "index is not finalized");
  int32_t rc = set_device(ix);
  if (rc) return rc;
  if (n == 0) return SLK_OK;
  DevBuf k, o;
  HIPCHK(k.ensure(n * 8 * ix->W));
  HIPCHK(o.ensure(n * 4));
  HIPCHK(hipMemcpy(k.p, keys, n * 8 * ix->W, hipMemcpyHostToDevice));
  if (ix->W > 1) launch_wide_lookup(ix->wt, ix->W, k.as<int64_t>(), n, o.as<int32_t>(), nullptr);
  else launch_table_lookup(ix->view(), k.as<int64_t>(), n, o.as<int32_t>(), nullptr);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(out_taxa, o.p, n * 4, hipMemcpyDeviceToHost));
  k.release();
  o.release();
  return SLK_OK;
}

void slk_index_destroy(slk_index *ix) {
  if (!ix) return;
  (void)hipSetDevice(ix->device);
  if (ix->cells) (void)hipFree(ix->cells);
  if (ix->wt.keys) (void)hipFree(ix->wt.keys);
  if (ix->wt.taxa) (void)hipFree(ix->wt.taxa);
  if (ix->d_max_disp) (void)hipFree(ix->d_max_disp);
  if (ix->d_counters) (void)hipFree(ix->d_counters);
  if (ix->d_parents) (void)hipFree(ix->d_parents);
  if (ix->d_parents_dense) (void)hipFree(ix->d_parents_dense);
  free_tax_nodes(ix);
  if (ix->d_to_orig) (void)hipFree(ix->d_to_orig);
  if (ix->d_to_dense) (void)hipFree(ix->d_to_dense);
  ix->stage_keys.release();
  ix->stage_taxa.release();
  ix->staging.release();
  if (ix->build_stream) (void)hipStreamDestroy(ix->build_stream);
  delete ix;
}

int32_t slk_stream_create(slk_index *ix, slk_stream **out) {
  if (!ix || !out) return fail(SLK_E_INVALID, "null argument");
  *out = nullptr;
  int32_t rc = set_device(ix);
  if (rc) return rc;
  slk_stream *st = new slk_stream();
  st->ix = ix;
  st->device = ix->device;
  HIPCHK(hipStreamCreate(&st->s));
  for (int i = 0; i < 4; i++) HIPCHK(hipEventCreate(&st->ev[i]));
  HIPCHK(hipMalloc((void **)&st->d_status, sizeof(int32_t)));
  HIPCHK(hipMemset(st->d_status, 0, sizeof(int32_t)));
  HIPCHK(hipHostMalloc((void **)&st->h_status, sizeof(int32_t), hipHostMallocDefault));
  *st->h_status = 0;
  *out = st;
  return SLK_OK;
}

int32_t slk_stream_synchronize(slk_stream *st) {
  if (!st) return fail(SLK_E_INVALID, "null argument");
  { int32_t rc_ = set_device(st->ix); if (rc_) return rc_; }
  HIPCHK(hipStreamSynchronize(st->s));
  return check_status(st);
}

int32_t slk_stream_set_merged_hits(slk_stream *st, int32_t on) {
  if (!st) return fail(SLK_E_INVALID, "null handle");
  st->merged_hits = on != 0;
  return SLK_OK;
}

void *slk_stream_hip_stream(slk_stream *st) { return st ? (void *)st->s : nullptr; }

void slk_stream_destroy(slk_stream *st) {
  if (!st) return;
  (void)hipSetDevice(st->device);
  (void)hipStreamSynchronize(st->s);
  DevBuf *bufs[] = {&st->span_keys, &st->span_meta, &st->span_taxon, &st->span_count, &st->bases, &st->offsets,
                    &st->mate_bases, &st->mate_offsets, &st->out_taxon, &st->out_cls, &st->out_nd, &st->out_tk,
                    &st->out_nh, &st->out_offsets, &st->out_items, &st->defer_list, &st->scan_tmp, &st->pk_codes, &st->pk_valid, &st->pk_mate_codes,
                    &st->pk_mate_valid};
  for (DevBuf *b : bufs) b->release();
  if (st->d_status) (void)hipFree(st->d_status);
  if (st->h_status) (void)hipHostFree(st->h_status);
  st->staging.release();
  st->staging_c.release();
  for (hipEvent_t e : st->up_ev) (void)hipEventDestroy(e);
  for (hipEvent_t e : st->dn_ev) (void)hipEventDestroy(e);
  if (st->cs) (void)hipStreamDestroy(st->cs);
  if (st->ds) { (void)hipStreamSynchronize(st->ds); (void)hipStreamDestroy(st->ds); }
  if (st->s2) { (void)hipStreamSynchronize(st->s2); (void)hipStreamDestroy(st->s2); }
  if (st->ev_unpack) (void)hipEventDestroy(st->ev_unpack);
  if (st->ev_fork) (void)hipEventDestroy(st->ev_fork);
  if (st->ev_join) (void)hipEventDestroy(st->ev_join);
  for (int i = 0; i < 4; i++) if (st->ev[i]) (void)hipEventDestroy(st->ev[i]);
  if (st->s) (void)hipStreamDestroy(st->s);
  delete st;
}

// span slots needed by a batch (see span_region in engine.h)
uint64_t span_slots(uint64_t total_bases, uint64_t total_mate_bases, uint64_t R, bool paired) {
  return total_bases + (paired ? total_mate_bases + R : 0) + 1;
}

int32_t ensure_scratch(slk_stream *st, uint64_t slots, uint64_t R) {
  HIPCHK(st->span_keys.ensure(slots * 8 * st->ix->W));
  HIPCHK(st->span_meta.ensure(slots * 4));
  HIPCHK(st->span_taxon.ensure(slots * 4));
  HIPCHK(st->span_count.ensure((R + 1) * 4));
  return SLK_OK;
}

int32_t check_ready(const slk_index *ix, const slk_stream *st, bool need_tax) {
  if (!ix || !st) return fail(SLK_E_INVALID, "null handle");
  if (st->ix != ix) return fail(SLK_E_INVALID, "stream belongs to a different index");
  if (!ix->finalized) return fail(SLK_E_STATE, "index is not finalized");
  if (need_tax && !ix->d_parents) return fail(SLK_E_STATE, "taxonomy not set");
  return SLK_OK;
}

// The fused wave-per-read kernels (fused.hip) cover windows of up to 32 m-mers; wider windows (and SLK_FORCE_V1=1, an
// A/B switch for tests) run the three separate lane-per-read kernels of kernels.hip.  Both are HIP: no CPU path.
static bool use_fused(const slk_index *ix) {
  static const bool force_v1 = getenv("SLK_FORCE_V1") != nullptr && getenv("SLK_FORCE_V1")[0] == '1';
  return !force_v1 && ix->W == 1 && ix->sp.w <= 32;
}

static bool force_wave() {  // SLK_FORCE_WAVE=1: A/B switch, classify with the wave-per-read kernel only
  static const bool v = getenv("SLK_FORCE_WAVE") != nullptr && getenv("SLK_FORCE_WAVE")[0] == '1';
  return v;
}


// The fused kernels keep a fragment's taxon -> count map in LDS (12 slots per lane, 128 per wave).  A fragment that hits more
// distinct taxa than that (long reads across conserved regions can) raises status bit 1; the batch is then classified again
// by the staged kernels, whose per-fragment map lives in HBM scratch and is unbounded -- the same three kernels that serve
// windows wider than 32 m-mers.  Slower (HBM intermediates), rare, and bit-identical for every other fragment.
static int32_t run_unbounded(slk_stream *st, const slk_stream::LastCall &L) {
  slk_index *ix = st->ix;
  const bool paired = L.mate_bases != nullptr;
  int32_t rc = ensure_scratch(st, span_slots(L.total, L.mate_total, L.R, paired) + L.span_shift, L.R);
  if (rc) return rc;
  uint64_t *const keys = st->span_keys.as<uint64_t>() + L.span_shift;   // (fused path only: one key word per span)
  int32_t *const meta = st->span_meta.as<int32_t>() + L.span_shift, *const taxa = st->span_taxon.as<int32_t>() + L.span_shift;
  launch_scan(ix->sp, L.bases, L.offsets, L.mate_bases, L.mate_offsets, L.R, keys, meta, st->span_count.as<int32_t>(), st->s);
  launch_probe(ix->view(), L.offsets, L.mate_offsets, L.R, keys, meta, st->span_count.as<int32_t>(), taxa, st->s);
  launch_classify(ix->d_parents, ix->d_nodes_orig, ix->T, L.offsets, L.mate_offsets, L.R, meta, taxa, st->span_count.as<int32_t>(), keys,
                  L.min_hit_groups, L.thr, L.C, L.out_stride, L.out_taxon, L.out_cls, L.out_nd, L.out_tk, L.out_nh, L.out_np, st->s);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(st->s));
  return SLK_OK;
}

int32_t check_status(slk_stream *st) {  // call after the stream has been synchronised
  int32_t v = *st->h_status;
  std::vector<slk_stream::LastCall> queued;
  queued.swap(st->queued);
  if (v != 0) {
    *st->h_status = 0;
    HIPCHK(hipMemsetAsync(st->d_status, 0, sizeof(int32_t), st->s));
    if (v == 1 && !queued.empty()) {
      // Some queued batch held a fragment with more distinct taxa than the LDS maps take.  The status word does not say
      // which, so every batch queued since the last synchronisation is classified again by the unbounded kernels, in
      // order (callers that reuse their output buffers from call to call end up with the last call's results, as before).
      for (const slk_stream::LastCall &L : queued) {
        if (!L.valid) return fail(SLK_E_CAPACITY, "a fragment hit more than %d distinct taxa; the per-read taxon map overflowed", 128);
        int32_t rc = run_unbounded(st, L);
        if (rc) return rc;
      }
      st->reran = true;
      return SLK_OK;
    }
    if (v & 2) return fail(SLK_E_CAPACITY, "a send region of slk_shard_step_device's EMIT job overflowed its capacity_per_owner");
    if (v & 1) return fail(SLK_E_CAPACITY, "a fragment hit more than %d distinct taxa; the per-read taxon map overflowed", 128);
    return fail(SLK_E_HIP, "device status %d", v);
  }
  return SLK_OK;
}

bool lane_path_ok(const slk_index *ix) {
  return use_fused(ix) && ix->sp.w <= 32 && ix->internal_taxon_bits() <= 22 && ix->d_nodes != nullptr;
}

static int32_t run_classify(slk_index *ix, slk_stream *st, const uint8_t *d_bases, const uint64_t *d_offsets,
                            const uint8_t *d_mate_bases, const uint64_t *d_mate_offsets, uint64_t R,
                            uint64_t total_bases, uint64_t total_mate_bases, int32_t min_hit_groups,
                            const double *thresholds, int32_t C, int32_t *d_out_taxon, uint8_t *d_out_classified,
                            int32_t *d_out_num_distinct, int32_t *d_out_total_kmers, int32_t *d_out_num_hits,
                            int32_t *d_out_num_probes, bool want_hits, uint64_t out_stride = 0, uint64_t span_shift = 0) {
  // span_shift (a sub-batch of a larger host call, hit lists wanted): its fragments' span regions are addressed by their ABSOLUTE
  // offsets but by the fragment's number INSIDE the sub-batch (span_region: offsets[r] + mate_offsets[r] + r for pairs), so the
  // arrays are handed over moved by the sub-batch's first fragment number -- the regions then are the ones the whole batch has,
  // and sub-batches do not overlap.  The caller has sized the scratch for the whole batch.
  if (out_stride == 0) out_stride = R;
  bool paired = d_mate_bases != nullptr;
  bool fused = use_fused(ix);
  int32_t rc;
  st->last_used_lane = false;
  if (!fused || want_hits) {
    rc = ensure_scratch(st, span_slots(total_bases, total_mate_bases, R, paired) + span_shift, R);
    if (rc) return rc;
  }
  Thresholds thr{};
  memcpy(thr.v, thresholds, C * sizeof(double));
  HIPCHK(hipEventRecord(st->ev[0], st->s));
  {
    if (st->queued.size() >= 4096) {  // (a caller that never synchronises: settle what is queued before taking more)
      HIPCHK(hipStreamSynchronize(st->s));
      rc = check_status(st);
      if (rc) return rc;
    }
    st->queued.emplace_back();
    slk_stream::LastCall &L = st->queued.back();
    L.valid = fused; L.want_hits = want_hits; L.thr = thr;
    L.bases = d_bases; L.offsets = d_offsets; L.mate_bases = d_mate_bases; L.mate_offsets = d_mate_offsets;
    L.R = R; L.total = total_bases; L.mate_total = total_mate_bases; L.min_hit_groups = min_hit_groups; L.C = C;
    L.out_stride = out_stride; L.span_shift = span_shift;
    L.out_taxon = d_out_taxon; L.out_cls = d_out_classified; L.out_nd = d_out_num_distinct; L.out_tk = d_out_total_kmers;
    L.out_nh = d_out_num_hits; L.out_np = d_out_num_probes;
  }
  if (fused) {
    FusedArgs A{};
    A.P = ix->sp; A.T = ix->view(); A.parents = ix->kernel_parents(); A.ntax = ix->kernel_ntax(); A.nodes = ix->kernel_nodes();
    A.bases = d_bases; A.offsets = d_offsets; A.mate_bases = d_mate_bases; A.mate_offsets = d_mate_offsets; A.R = R;
    A.out_stride = out_stride;
    A.min_hit_groups = min_hit_groups; A.thr = thr; A.C = C;
    A.out_taxon = d_out_taxon; A.out_classified = d_out_classified;
    A.out_nd = d_out_num_distinct; A.out_tk = d_out_total_kmers; A.out_nh = d_out_num_hits; A.out_np = d_out_num_probes;
    A.span_keys = nullptr;
    A.span_meta = want_hits ? st->span_meta.as<int32_t>() + span_shift : nullptr;
    A.span_taxon = want_hits ? st->span_taxon.as<int32_t>() + span_shift : nullptr;
    A.span_count = want_hits ? st->span_count.as<int32_t>() : nullptr;
    A.status = st->d_status;
    A.work_list = nullptr; A.work_count = nullptr; A.work_draw = nullptr;
    if (lane_path_ok(ix) && !force_wave() && R < 0xFFFFFFFFull) {  // (window of at most 32 m-mers, taxon ids of at most 22 bits)
      st->last_used_lane = true;
      // Hot path: one lane per fragment.  What that kernel does not take -- fragments over 1000 bases, taxon maps that overflow --
      // it appends to the hand-on list of the kernel that does (engine.h: FusedArgs.hand_*): four length classes for its own long
      // variant (1001 .. 4999 bases), the lane-per-segment kernel (unpaired, w = 5, the fragments that are long for their batch), the
      // wave-per-fragment kernel (the rest, and what the long variant hands on in turn).
      const size_t hdr_bytes = HandOn::WORDS * sizeof(uint64_t);
      const uint64_t long_cap = std::min<uint64_t>(R, (total_bases + total_mate_bases) / 1001 + 1);
      // SLK_LANE_LONG_MAX moves the long variant's limit (at most 8191: queue entries carry 13-bit k-mer counts; 0: no such pass)
      const char *long_env = getenv("SLK_LANE_LONG_MAX");
      const int long_max = std::min(long_env ? atoi(long_env) : 4999, 8191);
      // A batch whose fragments average more than 1000 bases gets a routing kernel instead of a first pass (engine.h:
      // FusedArgs.hand_short; SLK_ROUTE_FIRST=0 / 1 says so either way)
      const char *route_env = getenv("SLK_ROUTE_FIRST");
      const bool route_first = long_max > 1000 && (route_env ? route_env[0] == '1' : (total_bases + total_mate_bases) / 1000 > R);
      HIPCHK(st->defer_list.ensure(hdr_bytes + HandOn::entries(R, long_cap, route_first) * sizeof(uint32_t)));
      HIPCHK(hipMemsetAsync(st->defer_list.p, 0, hdr_bytes, st->s));
      A.hand_hdr = (unsigned long long *)st->defer_list.p;
      A.hand_lists = (uint32_t *)((char *)st->defer_list.p + hdr_bytes);
      A.hand_stride = R;
      A.hand_long_cap = long_cap;
      A.route_first = route_first ? 1 : 0;
      // SLK_SEG_MIN_LEN moves the segment kernel's limit (0: wave kernel only).  Read per call, like the others, so that tests can
      // move them.
      // Wave or segment kernel: on batches of ONE length the wave kernel is the faster one up to ~250 000 bases since round 4's diet
      // (115 against 101 Gbp/s at 15 kbp, 111 / 100 at 30 kbp, 91 / 92 at 100 kbp, 90 / 74 at 200 kbp, 70 / 75 at 300 kbp,
      // profiles/r04_long_routes.txt) -- but it takes a fragment per wave at ~15 Mbp/s, so a fragment that is long for its batch is
      // what the batch then waits for.  So the default follows the batch: the segment kernel takes what a single wave would need
      // about half the batch's time for -- fragments of more than 1/16384 of the batch's bases --, never under 16 000 bases (below
      // that its lanes have too little each) and always from 250 000; and the wave kernel starts its long fragments longest first
      // (engine.h: hand_hdr).  Nanopore-like mix, 200 .. 50 000 bases, 1 Gbp: 90-94 Gbp/s with the threshold at 12-16 000, 93-99 at
      // 30 000, 97-101 at 64 000 (none on the segment kernel).
      const char *seg_env = getenv("SLK_SEG_MIN_LEN");
      const uint64_t seg_auto = std::min<uint64_t>(250000, std::max<uint64_t>(16000, (total_bases + total_mate_bases) >> 14));
      const int seg_min = seg_env ? atoi(seg_env) : (int)seg_auto;
      // (hit lists: the segment kernel can put them together -- SLK_SEG_HITS=1 --, but the queues that take its spans to memory
      //  in order cost it half its resident waves, and it measured 51-53 Gbp/s against the wave kernel's 68-79 on the same reads:
      //  profiles/r03_long_hits_*.json; so per-read lines of long reads keep the wave kernel unless asked otherwise)
      const char *seg_hits_env = getenv("SLK_SEG_HITS");
      const bool seg_hits = seg_hits_env != nullptr && seg_hits_env[0] == '1';
      const bool seg_on = (!want_hits || seg_hits) && !paired && ix->sp.w == 5 && seg_min > 0;
      A.long_max = long_max > 1000 ? (uint32_t)long_max : 0;
      if (A.long_max) {  // class borders: a geometric ladder from 1000 to the limit (a tile's lanes then differ by at most ~1.5x)
        const double ratio = pow((double)A.long_max / 1000.0, 0.25);
        for (int i = 0; i < 3; i++) A.long_bound[i] = (uint32_t)(1000.0 * pow(ratio, i + 1));
      }
      A.seg_min_len = seg_on ? (uint32_t)std::max(seg_min, (int)std::max<uint32_t>(A.long_max, 1000) + 1) : 0;
      A.wave_min = std::max<uint32_t>(A.long_max, 1000) + 1;   // (the wave kernel's eight length classes: 1.75^7 = 50 times the shortest)
      A.wave_ratio_q10 = 1792;
      if (route_first) launch_route(A, st->s);
      else launch_lane(A, nullptr, 1000, st->s);  // (the one-word map entries carry 10-bit k-mer counts)
      // The passes over the hand-on lists depend on the first pass only, and the long variant runs BESIDE the other two (which
      // follow each other on a second stream): with a few hundred thousand long fragments in a batch the long variant is a handful
      // of waves per CU working through 5 000 lockstep steps, the segment pass not much more, and the wave kernel behind them on one
      // stream waited for both (nanopore-like mix: 1.1 + 4.9 + 5.8 ms one after the other, 77-84 Gbp/s; 95-101 this way;
      // profiles/r04_long_mixed_trace.txt).  Segment pass before wave pass: the wave kernel is bound by instruction issue and holds
      // every wave slot until it is through, the other two are chains of dependent steps that share a CU well.  What the long
      // variant hands on in turn (map overflows) goes to a list of its own that a second launch of the wave kernel takes when
      // both streams are through.
      if (!st->s2) {
        HIPCHK(hipStreamCreateWithFlags(&st->s2, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&st->ev_fork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&st->ev_join, hipEventDisableTiming));
      }
      HIPCHK(hipEventRecord(st->ev_fork, st->s));
      if (A.long_max) launch_lane_long(A, A.long_max, st->s);   // (first: its chain of steps is the longest, whoever comes first gets the CUs)
      HIPCHK(hipStreamWaitEvent(st->s2, st->ev_fork, 0));
      if (seg_on) {
        FusedArgs B = A;
        if (want_hits) B.span_keys = st->span_keys.as<uint64_t>() + span_shift;   // (scratch of the hit lists: the spans' places before the borders are settled)
        B.work_list = A.hand_lists + HandOn::list_at(HandOn::SEG, R, long_cap); B.work_count = A.hand_hdr + HandOn::SEG; B.work_draw = A.hand_hdr + HandOn::SEG_DRAW;
        launch_segments(B, st->s2);
      }
      {
        FusedArgs W = A;
        launch_order_wave_list(W, st->s2);
        W.work_list = A.hand_lists + HandOn::ordered_at(R, long_cap); W.work_count = A.hand_hdr + HandOn::ORDERED; W.work_draw = A.hand_hdr + HandOn::WAVE_DRAW;
        launch_fused(want_hits ? MODE_HITS : MODE_CLASSIFY, W, st->s2);
      }
      HIPCHK(hipEventRecord(st->ev_join, st->s2));
      HIPCHK(hipStreamWaitEvent(st->s, st->ev_join, 0));
      if (A.long_max) {
        A.work_list = A.hand_lists + HandOn::list_at(HandOn::LATE, R, long_cap); A.work_count = A.hand_hdr + HandOn::N_LATE; A.work_draw = A.hand_hdr + HandOn::LATE_DRAW;
        launch_fused(want_hits ? MODE_HITS : MODE_CLASSIFY, A, st->s);
      }
    } else {
      launch_fused(want_hits ? MODE_HITS : MODE_CLASSIFY, A, st->s);
    }
    HIPCHK(hipEventRecord(st->ev[1], st->s));
    HIPCHK(hipEventRecord(st->ev[2], st->s));
  } else {
    if (ix->W > 1) {
      launch_wide_scan(ix->wp, d_bases, d_offsets, d_mate_bases, d_mate_offsets, R, st->span_keys.as<uint64_t>(),
                       st->span_meta.as<int32_t>(), st->span_count.as<int32_t>(), st->s);
      HIPCHK(hipEventRecord(st->ev[1], st->s));
      launch_wide_probe(ix->wt, ix->W, d_offsets, d_mate_offsets, R, st->span_keys.as<uint64_t>(), st->span_meta.as<int32_t>(),
                        st->span_count.as<int32_t>(), st->span_taxon.as<int32_t>(), st->s);
    } else {
      launch_scan(ix->sp, d_bases, d_offsets, d_mate_bases, d_mate_offsets, R, st->span_keys.as<uint64_t>(),
                  st->span_meta.as<int32_t>(), st->span_count.as<int32_t>(), st->s);
      HIPCHK(hipEventRecord(st->ev[1], st->s));
      launch_probe(ix->view(), d_offsets, d_mate_offsets, R, st->span_keys.as<uint64_t>(), st->span_meta.as<int32_t>(),
                   st->span_count.as<int32_t>(), st->span_taxon.as<int32_t>(), st->s);
    }
    HIPCHK(hipEventRecord(st->ev[2], st->s));
    // the key slots are dead after the probe: the per-read taxon->count map reuses them
    launch_classify(ix->d_parents, ix->d_nodes_orig, ix->T, d_offsets, d_mate_offsets, R, st->span_meta.as<int32_t>(),
                    st->span_taxon.as<int32_t>(), st->span_count.as<int32_t>(), st->span_keys.as<uint64_t>(),
                    min_hit_groups, thr, C, out_stride, d_out_taxon, d_out_classified, d_out_num_distinct,
                    d_out_total_kmers, d_out_num_hits, d_out_num_probes, st->s);
  }
  HIPCHK(hipEventRecord(st->ev[3], st->s));
  HIPCHK(hipMemcpyAsync(st->h_status, st->d_status, sizeof(int32_t), hipMemcpyDeviceToHost, st->s));
  HIPCHK(hipGetLastError());
  st->timed = true;
  return SLK_OK;
}

int32_t slk_classify_batch_device(slk_index *ix, slk_stream *st, const uint8_t *d_bases, const uint64_t *d_offsets,
                                  const uint8_t *d_mate_bases, const uint64_t *d_mate_offsets, uint64_t R,
                                  uint64_t total_bases, uint64_t total_mate_bases, int32_t min_hit_groups,
                                  const double *thresholds, int32_t C, int32_t *d_out_taxon,
                                  uint8_t *d_out_classified, int32_t *d_out_num_distinct,
                                  int32_t *d_out_total_kmers, int32_t *d_out_num_hits,
                                  int32_t *d_out_num_probes) {
  int32_t rc = check_ready(ix, st, true);
  if (rc) return rc;
  if (C < 1 || C > MAX_THRESHOLDS || !thresholds) return fail(SLK_E_INVALID, "need 1..%d thresholds", MAX_THRESHOLDS);
  if (R && (!d_bases || !d_offsets || !d_out_taxon || !d_out_classified)) return fail(SLK_E_INVALID, "null argument");
  if ((d_mate_bases == nullptr) != (d_mate_offsets == nullptr))
    return fail(SLK_E_INVALID, "mate_bases and mate_offsets must be given together");
  rc = set_device(ix);
  if (rc) return rc;
  return run_classify(ix, st, d_bases, d_offsets, d_mate_bases, d_mate_offsets, R, total_bases, total_mate_bases,
                      min_hit_groups, thresholds, C, d_out_taxon, d_out_classified, d_out_num_distinct,
                      d_out_total_kmers, d_out_num_hits, d_out_num_probes, false);
}

int32_t slk_scan_device(slk_index *ix, slk_stream *st, const uint8_t *d_bases, const uint64_t *d_offsets,
                        const uint8_t *d_mate_bases, const uint64_t *d_mate_offsets, uint64_t R,
                        uint64_t *d_span_keys, int32_t *d_span_meta, int32_t *d_span_count) {
  int32_t rc = check_ready(ix, st, false);
  if (rc) return rc;
  if (ix->W > 1) return fail(SLK_E_UNSUPPORTED, "the staged and sharded entry points support minimizers of up to 32 nt (one id column)");
  if (R && (!d_bases || !d_offsets || !d_span_keys || !d_span_meta || !d_span_count)) return fail(SLK_E_INVALID, "null argument");
  if ((d_mate_bases == nullptr) != (d_mate_offsets == nullptr))
    return fail(SLK_E_INVALID, "mate_bases and mate_offsets must be given together");
  rc = set_device(ix);
  if (rc) return rc;
  if (use_fused(ix)) {
    FusedArgs A{};
    A.P = ix->sp; A.bases = d_bases; A.offsets = d_offsets; A.mate_bases = d_mate_bases; A.mate_offsets = d_mate_offsets;
    A.R = R; A.span_keys = d_span_keys; A.span_meta = d_span_meta; A.span_count = d_span_count; A.status = st->d_status;
    launch_fused(MODE_SPANS, A, st->s);
  } else {
    launch_scan(ix->sp, d_bases, d_offsets, d_mate_bases, d_mate_offsets, R, d_span_keys, d_span_meta, d_span_count, st->s);
  }
  HIPCHK(hipGetLastError());
  return SLK_OK;
}

int32_t slk_lookup_device(slk_index *ix, slk_stream *st, const int64_t *d_keys, uint64_t n, int32_t *d_out_taxa) {
  int32_t rc = check_ready(ix, st, false);
  if (rc) return rc;
  if (ix->W > 1) return fail(SLK_E_UNSUPPORTED, "the staged and sharded entry points support minimizers of up to 32 nt (one id column)");
  if (n && (!d_keys || !d_out_taxa)) return fail(SLK_E_INVALID, "null argument");
  rc = set_device(ix);
  if (rc) return rc;
  launch_lookup_coop(ix->view(), d_keys, n, d_out_taxa, st->s);
  HIPCHK(hipGetLastError());
  return SLK_OK;
}

// The table's range reduction and its inverse as plain host arithmetic (engine.h: table_slot / table_hash_of), for tests.
static TableGeom geom_for_tests(uint64_t nbuckets) {
  TableGeom g{};
  g.nbuckets = nbuckets;
  g.q = std::max(5, ceil_log2_u64(nbuckets));
  g.rem_mask = (1ULL << (64 - g.q)) - 1;
  return g;
}
int32_t slk_table_slot(uint64_t nbuckets, uint64_t hash, uint32_t *home, uint64_t *rem) {
  if (nbuckets < 32 || nbuckets > (1ULL << 32) || !home || !rem) return fail(SLK_E_INVALID, "32 <= nbuckets <= 2^32");
  table_slot(geom_for_tests(nbuckets), hash, *home, *rem);
  return SLK_OK;
}
int32_t slk_table_hash_of(uint64_t nbuckets, uint32_t home, uint64_t rem, uint64_t *hash) {
  if (nbuckets < 32 || nbuckets > (1ULL << 32) || home >= nbuckets || !hash) return fail(SLK_E_INVALID, "32 <= nbuckets <= 2^32, home < nbuckets");
  *hash = table_hash_of(geom_for_tests(nbuckets), home, rem);
  return SLK_OK;
}

uint32_t slk_shard_of(int64_t key, uint32_t n_shards) {
  return n_shards ? (uint32_t)(fmix64((uint64_t)key) % n_shards) : 0;
}


// rows of the batch log (engine.h: ShardIO.batch_base) a batch needs: tile t starts at row floor(span_region(64 t) / 64) + t
uint64_t slk_shard_batch_rows(uint64_t total_bases, uint64_t total_mate_bases, uint64_t R, int32_t paired) {
  return (span_slots(total_bases, total_mate_bases, R, paired != 0) >> 6) + (R + 63) / 64 + 2;
}

// entries a wave reserves at a time in an owner's send region (engine.h: ShardIO.chunk): one atomic per chunk and owner on ONE
// address per owner, so the fewer owners the larger the chunk (an owner's keys come 1 / n_shards as fast); what stays unwritten at
// the end of a launch is half a chunk per wave and owner -- 2 M entries of the 390 M of a 10 M-read batch whatever n_shards is
uint32_t slk_shard_chunk(uint32_t n_shards) {
  uint32_t c = 1024;
  while (c > 64 && c * n_shards > 1024) c >>= 1;
  return c;
}

// FusedArgs / ShardIO of a batch's EMIT job from its lists
static void fill_emit(const slk_index *ix, slk_stream *st, const slk_shard_lists &E, FusedArgs &A, ShardIO &S) {
  A.P = ix->sp; A.T = ix->view();
  A.bases = E.d_bases; A.offsets = E.d_offsets; A.mate_bases = E.d_mate_bases; A.mate_offsets = E.d_mate_offsets; A.R = E.R;
  A.span_meta = E.d_span_meta; A.span_taxon = E.d_span_taxon; A.span_count = E.d_span_count;
  A.status = st->d_status;
  S.n_shards = (int32_t)E.n_shards; S.chunk = slk_shard_chunk(E.n_shards); S.cap = E.capacity_per_owner;
  S.send_keys = E.d_send_keys; S.cursors = (unsigned long long *)E.d_cursors; S.send_meta = E.d_send_meta;
  S.batch_log = (uint4 *)E.d_batch_log; S.tile_rows = (uint2 *)E.d_tile_rows; S.read_info = (int2 *)E.d_read_info;
}
static int32_t check_lists(const slk_shard_lists &E, const char *what) {
  if (E.n_shards < 1 || E.n_shards > 64) return fail(SLK_E_INVALID, "%s: n_shards %u outside 1..64", what, E.n_shards);
  const uint32_t chunk = slk_shard_chunk(E.n_shards);
  if (E.capacity_per_owner < chunk || E.capacity_per_owner % chunk != 0 || E.capacity_per_owner >= (1ull << 32))
    return fail(SLK_E_INVALID, "%s: capacity_per_owner must be a multiple of slk_shard_chunk(n_shards) = %u below 2^32", what, chunk);
  if (!E.d_cursors || !E.d_defer || (E.R && (!E.d_offsets || !E.d_send_keys || !E.d_send_meta || !E.d_batch_log || !E.d_tile_rows || !E.d_read_info)))
    return fail(SLK_E_INVALID, "%s: null argument", what);
  if ((E.d_mate_bases == nullptr) != (E.d_mate_offsets == nullptr)) return fail(SLK_E_INVALID, "%s: mate_bases and mate_offsets must be given together", what);
  if ((E.d_span_meta == nullptr) != (E.d_span_taxon == nullptr) || (E.d_span_meta == nullptr) != (E.d_span_count == nullptr))
    return fail(SLK_E_INVALID, "%s: the span arrays of the hit lists must be given together", what);
  if (E.R >= 0xFFFFFFFFull) return fail(SLK_E_INVALID, "%s: a batch holds fewer than 2^32 fragments", what);
  return SLK_OK;
}

// One pipeline step of the table-sharded mode (engine.h: ShardIO): up to three jobs of three different batches in ONE kernel.
int32_t slk_shard_step_device(slk_index *ix, slk_stream *st, const slk_shard_lists *emit, const slk_shard_lookup *lookup,
                              const slk_shard_lists *apply_lists, const slk_shard_results *apply) {
  int32_t rc = check_ready(ix, st, apply != nullptr);
  if (rc) return rc;
  if (ix->W > 1) return fail(SLK_E_UNSUPPORTED, "the staged and sharded entry points support minimizers of up to 32 nt (one id column)");
  if (!lane_path_ok(ix)) return fail(SLK_E_UNSUPPORTED, "splitter outside the fused kernel's range: use the staged calls");
  if ((apply_lists == nullptr) != (apply == nullptr)) return fail(SLK_E_INVALID, "apply_lists and apply must be given together");
  if (emit && (rc = check_lists(*emit, "emit"))) return rc;
  if (emit && emit->R && !emit->d_bases) return fail(SLK_E_INVALID, "emit: null argument");
  if (apply_lists && (rc = check_lists(*apply_lists, "apply"))) return rc;
  if (lookup && lookup->n && (!lookup->d_keys || !lookup->d_out_taxa)) return fail(SLK_E_INVALID, "lookup: null argument");
  if (apply) {
    if (apply->C < 1 || apply->C > MAX_THRESHOLDS || !apply->thresholds) return fail(SLK_E_INVALID, "need 1..%d thresholds", MAX_THRESHOLDS);
    if (apply_lists->R && (!apply->d_taxa || !apply->d_out_taxon || !apply->d_out_classified)) return fail(SLK_E_INVALID, "apply: null argument");
    if (emit && emit->R && (apply_lists->d_span_meta == nullptr) != (emit->d_span_meta == nullptr))
      return fail(SLK_E_INVALID, "the batches of one step write hit lists or none does");
  }
  rc = set_device(ix);
  if (rc) return rc;
  const bool scans = emit && emit->R != 0;
  FusedArgs A{};
  ShardIO S{};
  A.P = ix->sp; A.status = st->d_status;
  if (scans) {
    fill_emit(ix, st, *emit, A, S);
    HIPCHK(hipMemsetAsync(emit->d_cursors, 0, ((size_t)emit->n_shards + 3) * sizeof(uint64_t), st->s));
    HIPCHK(hipMemsetAsync(emit->d_defer, 0, emit->R * sizeof(int32_t), st->s));
  } else if (emit) {
    HIPCHK(hipMemsetAsync(emit->d_cursors, 0, ((size_t)emit->n_shards + 3) * sizeof(uint64_t), st->s));
  }
  uint64_t *draw = scans ? emit->d_cursors + emit->n_shards : nullptr;
  bool lookup_beside = false;
  if (lookup && lookup->n) {
    // The lookups ride in the scan, their 64-key batches dealt out to its tiles -- unless the scan is far too short for them (a
    // tile sends off about 2 / (w + 1) keys per base; a tile handed several times as many lookups as that would finish them alone,
    // at its end, with the rest of the part idle): then they run as a kernel of their own, like those of a step without a scan.
    const uint64_t tiles = scans ? (emit->R + 63) / 64 : 0, batches = (lookup->n + 63) / 64;
    const double own = scans ? 2.0 / (ix->sp.w + 1) * (double)(emit->total_bases + emit->total_mate_bases) / 64.0 / (double)tiles : 0;
    const uint64_t per_tile = scans ? (batches + tiles - 1) / tiles : 0;
    if (scans && (double)per_tile <= 3.0 * own + 8.0) {
      S.side_keys = lookup->d_keys; S.side_n = lookup->n; S.side_out = lookup->d_out_taxa;
      S.side_per_tile = (uint32_t)per_tile;
    } else {
      // (beside the step's kernel when there is one -- the replay of a step without a scan, the pipeline's drain: the replay waits
      //  for two dependent loads per row, the lookups for the table; on one stream they took 2.1 + 9.6 ms, side by side ~10)
      const bool beside = scans || (apply && apply_lists->R != 0);
      if (beside) {
        if (!st->s2) {
          HIPCHK(hipStreamCreateWithFlags(&st->s2, hipStreamNonBlocking));
          HIPCHK(hipEventCreateWithFlags(&st->ev_fork, hipEventDisableTiming));
          HIPCHK(hipEventCreateWithFlags(&st->ev_join, hipEventDisableTiming));
        }
        HIPCHK(hipEventRecord(st->ev_fork, st->s));
        HIPCHK(hipStreamWaitEvent(st->s2, st->ev_fork, 0));
      }
      launch_lookup_coop(ix->view(), lookup->d_keys, lookup->n, lookup->d_out_taxa, beside ? st->s2 : st->s);
      HIPCHK(hipGetLastError());
      if (beside) { HIPCHK(hipEventRecord(st->ev_join, st->s2)); lookup_beside = true; }
    }
  }
  ApplyJob J{};
  const bool applies = apply && apply_lists->R != 0;
  if (applies) {
    Thresholds thr{};
    memcpy(thr.v, apply->thresholds, apply->C * sizeof(double));
    FusedArgs &B = J.A;
    B.P = ix->sp; B.T = ix->view(); B.parents = ix->kernel_parents(); B.ntax = ix->kernel_ntax(); B.nodes = ix->kernel_nodes();
    B.offsets = apply_lists->d_offsets; B.mate_offsets = apply_lists->d_mate_offsets; B.R = apply_lists->R; B.out_stride = apply_lists->R;
    B.min_hit_groups = apply->min_hit_groups; B.thr = thr; B.C = apply->C;
    B.out_taxon = apply->d_out_taxon; B.out_classified = apply->d_out_classified; B.out_nd = apply->d_out_num_distinct;
    B.out_tk = apply->d_out_total_kmers; B.out_nh = apply->d_out_num_hits;
    B.span_meta = apply_lists->d_span_meta; B.span_taxon = apply_lists->d_span_taxon; B.span_count = apply_lists->d_span_count;
    B.status = st->d_status;
    J.n_shards = (int32_t)apply_lists->n_shards; J.cap = apply_lists->capacity_per_owner;
    J.send_meta = apply_lists->d_send_meta; J.batch_log = (const uint4 *)apply_lists->d_batch_log;
    J.tile_rows = (const uint2 *)apply_lists->d_tile_rows; J.read_info = (const int2 *)apply_lists->d_read_info;
    J.taxa = apply->d_taxa; J.to_dense = ix->d_to_dense; J.n_to_dense = ix->T; J.defer = apply_lists->d_defer;
    J.n_deferred = (unsigned long long *)(apply_lists->d_cursors + apply_lists->n_shards + 2);
    if (!scans) {   // a step without a scan: the replay's waves draw their tiles from a counter of its own (the batch's spare word)
      draw = apply_lists->d_cursors + apply_lists->n_shards + 1;
      HIPCHK(hipMemsetAsync(draw, 0, sizeof(uint64_t), st->s));
      S.n_shards = 0;
    }
  }
  if (scans || applies) {
    // (S.cursors[S.n_shards] is the tile draw: with a scan the batch's own word behind its cursors, else the word chosen above)
    if (!scans) S.cursors = (unsigned long long *)draw;
    st->queued.emplace_back();   // (not re-runnable: a map overflow of the replay defers the fragment, a full region is an error)
    launch_lane_step(A, S, J, scans ? emit->d_defer : nullptr, 1000, st->s);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(st->h_status, st->d_status, sizeof(int32_t), hipMemcpyDeviceToHost, st->s));
  }
  if (lookup_beside) HIPCHK(hipStreamWaitEvent(st->s, st->ev_join, 0));
  return SLK_OK;
}

int32_t slk_classify_hits_device(slk_index *ix, slk_stream *st, const uint64_t *d_offsets,
                                 const uint64_t *d_mate_offsets, uint64_t R, const int32_t *d_span_meta,
                                 const int32_t *d_span_taxon, const int32_t *d_span_count, uint64_t *d_scratch,
                                 int32_t min_hit_groups, const double *thresholds, int32_t C, int32_t *d_out_taxon,
                                 uint8_t *d_out_classified, int32_t *d_out_num_distinct, int32_t *d_out_total_kmers,
                                 int32_t *d_out_num_hits) {
  int32_t rc = check_ready(ix, st, true);
  if (rc) return rc;
  if (ix->W > 1) return fail(SLK_E_UNSUPPORTED, "the staged and sharded entry points support minimizers of up to 32 nt (one id column)");
  if (C < 1 || C > MAX_THRESHOLDS || !thresholds) return fail(SLK_E_INVALID, "need 1..%d thresholds", MAX_THRESHOLDS);
  if (R && (!d_offsets || !d_span_meta || !d_span_taxon || !d_span_count || !d_scratch || !d_out_taxon || !d_out_classified))
    return fail(SLK_E_INVALID, "null argument");
  rc = set_device(ix);
  if (rc) return rc;
  Thresholds thr{};
  memcpy(thr.v, thresholds, C * sizeof(double));
  launch_classify(ix->d_parents, ix->d_nodes_orig, ix->T, d_offsets, d_mate_offsets, R, d_span_meta, d_span_taxon, d_span_count, d_scratch,
                  min_hit_groups, thr, C, R, d_out_taxon, d_out_classified, d_out_num_distinct,
                  d_out_total_kmers, d_out_num_hits, nullptr, st->s);
  HIPCHK(hipGetLastError());
  return SLK_OK;
}

// Classifier.classify (object, Classifier.scala:439-454) for hit lists the caller assembled itself: the host merges the
// hits of fragments that share a title (groupBy("seqTitle"), Classifier.scala:92, then sorted by ordinal :136) and has the
// merged lists classified here.  Host pointers; synchronous.
int32_t slk_classify_hits(slk_index *ix, slk_stream *st, uint64_t R, const uint64_t *hit_offsets, const slk_hit *hits,
                          const uint8_t *distinct, int32_t min_hit_groups, const double *thresholds, int32_t C,
                          int32_t *out_taxon, uint8_t *out_classified, int32_t *out_num_distinct, int32_t *out_total_kmers) {
  int32_t rc = check_ready(ix, st, true);
  if (rc) return rc;
  if (C < 1 || C > MAX_THRESHOLDS || !thresholds) return fail(SLK_E_INVALID, "need 1..%d thresholds", MAX_THRESHOLDS);
  if (!hit_offsets || (R && (!out_taxon || !out_classified))) return fail(SLK_E_INVALID, "null argument");
  for (uint64_t r = 0; r < R; r++)
    if (hit_offsets[r + 1] < hit_offsets[r] || hit_offsets[r + 1] - hit_offsets[r] > 0x7fffffffULL)
      return fail(SLK_E_INVALID, "hit_offsets must be non-decreasing (read %llu)", (unsigned long long)r);
  const uint64_t n = R ? hit_offsets[R] - hit_offsets[0] : 0;
  if (n && !hits) return fail(SLK_E_INVALID, "null argument");
  rc = set_device(ix);
  if (rc) return rc;
  if (R == 0) return SLK_OK;
  // the staged classify kernel's input: one slot per hit (fragment r's slots start at offsets[r]), meta = kmers|flag|distinct
  const uint64_t h0 = hit_offsets[0];
  std::vector<uint64_t> offs(R + 1);
  std::vector<int32_t> meta(n + 1), taxon(n + 1), count(R);
  for (uint64_t r = 0; r <= R; r++) offs[r] = hit_offsets[r] - h0;
  for (uint64_t r = 0; r < R; r++) count[r] = (int32_t)(offs[r + 1] - offs[r]);
  for (uint64_t i = 0; i < n; i++) {
    const slk_hit &h = hits[h0 + i];
    const int32_t flag = h.taxon == SLK_TAXON_AMBIGUOUS ? SLK_FLAG_AMBIGUOUS : h.taxon == SLK_TAXON_MATE_PAIR_BORDER ? SLK_FLAG_MATE_PAIR_BORDER : SLK_FLAG_SEQUENCE;
    if (h.taxon < SLK_TAXON_MATE_PAIR_BORDER) return fail(SLK_E_INVALID, "hit %llu: taxon %d", (unsigned long long)i, h.taxon);
    meta[i] = pack_meta(h.count, flag, (flag == SLK_FLAG_SEQUENCE && distinct && distinct[h0 + i]) ? 1 : 0);
    taxon[i] = h.taxon;
  }
  HIPCHK(st->offsets.ensure((R + 1) * 8));
  HIPCHK(st->span_meta.ensure((n + 1) * 4));
  HIPCHK(st->span_taxon.ensure((n + 1) * 4));
  HIPCHK(st->span_count.ensure((R + 1) * 4));
  HIPCHK(st->span_keys.ensure((n + 1) * 8));
  HIPCHK(st->out_taxon.ensure((size_t)C * R * 4));
  HIPCHK(st->out_cls.ensure((size_t)C * R));
  HIPCHK(st->out_nd.ensure(R * 4));
  HIPCHK(st->out_tk.ensure(R * 4));
  rc = copy_in(st, st->offsets.p, offs.data(), (R + 1) * 8);
  if (!rc) rc = copy_in(st, st->span_meta.p, meta.data(), (n + 1) * 4);
  if (!rc) rc = copy_in(st, st->span_taxon.p, taxon.data(), (n + 1) * 4);
  if (!rc) rc = copy_in(st, st->span_count.p, count.data(), R * 4);
  if (rc) return rc;
  Thresholds thr{};
  memcpy(thr.v, thresholds, C * sizeof(double));
  launch_classify(ix->d_parents, ix->d_nodes_orig, ix->T, st->offsets.as<uint64_t>(), nullptr, R, st->span_meta.as<int32_t>(),
                  st->span_taxon.as<int32_t>(), st->span_count.as<int32_t>(), st->span_keys.as<uint64_t>(), min_hit_groups,
                  thr, C, R, st->out_taxon.as<int32_t>(), st->out_cls.as<uint8_t>(), st->out_nd.as<int32_t>(),
                  st->out_tk.as<int32_t>(), nullptr, nullptr, st->s);
  HIPCHK(hipGetLastError());
  rc = copy_out(st, out_taxon, st->out_taxon.p, (size_t)C * R * 4);
  if (!rc) rc = copy_out(st, out_classified, st->out_cls.p, (size_t)C * R);
  if (!rc && out_num_distinct) rc = copy_out(st, out_num_distinct, st->out_nd.p, R * 4);
  if (!rc && out_total_kmers) rc = copy_out(st, out_total_kmers, st->out_tk.p, R * 4);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(st->s));
  return SLK_OK;
}

int32_t slk_stream_last_deferred(slk_stream *st, uint64_t *out_count) {
  if (!st || !out_count) return fail(SLK_E_INVALID, "null argument");
  { int32_t rc_ = set_device(st->ix); if (rc_) return rc_; }
  *out_count = 0;
  HIPCHK(hipStreamSynchronize(st->s));
  if (st->defer_list.p && st->last_used_lane)   // (word 9 of the hand-on header: what the first pass handed on)
    HIPCHK(hipMemcpy(out_count, (const uint64_t *)st->defer_list.p + HandOn::HANDED, sizeof(uint64_t), hipMemcpyDeviceToHost));
  return SLK_OK;
}

int32_t slk_stream_last_stage_ms(slk_stream *st, float out_ms[3]) {
  if (!st || !out_ms) return fail(SLK_E_INVALID, "null argument");
  if (!st->timed) return fail(SLK_E_STATE, "no classify call has been issued on this stream");
  { int32_t rc_ = set_device(st->ix); if (rc_) return rc_; }
  HIPCHK(hipEventSynchronize(st->ev[3]));
  HIPCHK(hipEventElapsedTime(&out_ms[0], st->ev[0], st->ev[1]));
  HIPCHK(hipEventElapsedTime(&out_ms[1], st->ev[1], st->ev[2]));
  HIPCHK(hipEventElapsedTime(&out_ms[2], st->ev[2], st->ev[3]));
  return SLK_OK;
}

static int32_t validate_reads(const uint64_t *offsets, const uint64_t *mate_offsets, uint64_t R) {
  // (4 M reads are 4 M compares per array: split over the copy threads)
  const uint64_t PART = 1 << 18;
  const uint64_t parts = (R + PART - 1) / PART;
  std::vector<uint64_t> bad(parts, ~0ULL);
  host_pool().parallel_for(parts, [&](size_t pi) {
    const uint64_t r1 = std::min<uint64_t>(R, (pi + 1) * PART);
    for (uint64_t r = pi * PART; r < r1; r++) {
      const bool ok = offsets[r + 1] >= offsets[r] && offsets[r + 1] - offsets[r] <= 0x7fffffffULL &&
                      (!mate_offsets || (mate_offsets[r + 1] >= mate_offsets[r] && mate_offsets[r + 1] - mate_offsets[r] <= 0x7fffffffULL));
      if (!ok) { bad[pi] = r; break; }
    }
  });
  for (uint64_t b : bad)
    if (b != ~0ULL)
      return fail(SLK_E_INVALID, "offsets (and mate_offsets) must be non-decreasing with reads shorter than 2^31 (read %llu)", (unsigned long long)b);
  return SLK_OK;
}

int32_t upload_reads(slk_stream *st, const uint8_t *bases, const uint64_t *offsets, const uint8_t *mate_bases,
                            const uint64_t *mate_offsets, uint64_t R, uint64_t *total, uint64_t *mate_total) {
  int32_t rc = validate_reads(offsets, mate_offsets, R);
  if (rc) return rc;
  *total = offsets[R];
  *mate_total = mate_offsets ? mate_offsets[R] : 0;
  HIPCHK(st->bases.ensure(*total));
  HIPCHK(st->offsets.ensure((R + 1) * 8));
  rc = copy_in(st, st->bases.p, bases, *total);
  if (!rc) rc = copy_in(st, st->offsets.p, offsets, (R + 1) * 8);
  if (rc) return rc;
  if (mate_offsets) {
    HIPCHK(st->mate_bases.ensure(*mate_total));
    HIPCHK(st->mate_offsets.ensure((R + 1) * 8));
    rc = copy_in(st, st->mate_bases.p, mate_bases, *mate_total);
    if (!rc) rc = copy_in(st, st->mate_offsets.p, mate_offsets, (R + 1) * 8);
    if (rc) return rc;
  }
  return SLK_OK;
}

// counts (device, int32[R]) -> out_offsets (host, u64[R+1]); uploads the offsets for a gather kernel
int32_t counts_to_offsets(slk_stream *st, const int32_t *d_counts, uint64_t R, uint64_t *out_offsets,
                                 uint64_t capacity) {
  out_offsets[0] = 0;
  if (R == 0) return SLK_OK;
  HIPCHK(st->out_offsets.ensure((R + 1) * 8));
  HIPCHK(st->scan_tmp.ensure((R / 2048 + 2) * 8));
  launch_counts_to_offsets(d_counts, R, st->out_offsets.as<uint64_t>(), st->scan_tmp.as<uint64_t>(), st->s);   // (kernels.hip)
  HIPCHK(hipGetLastError());
  int32_t rc = copy_out(st, out_offsets, st->out_offsets.p, (R + 1) * 8);
  if (rc) return rc;
  if (out_offsets[R] > capacity)
    return fail(SLK_E_CAPACITY, "output needs %llu entries, capacity is %llu", (unsigned long long)out_offsets[R],
                (unsigned long long)capacity);
  return SLK_OK;
}

// slk_spans_batch / slk_spans_batch_wide: out_keys (nullable) receives the spans' id1..idW rows
static int32_t spans_batch(slk_index *ix, slk_stream *st, const uint8_t *bases, const uint64_t *offsets, const uint8_t *mate_bases,
                           const uint64_t *mate_offsets, uint64_t R, uint64_t *out_span_offsets, slk_span *out_spans, int64_t *out_keys,
                           uint64_t spans_capacity) {
  int32_t rc = check_ready(ix, st, false);
  if (rc) return rc;
  if (ix->W > 1 && !out_keys) return fail(SLK_E_UNSUPPORTED, "slk_spans_batch returns one key word per span: minimizers of up to 32 nt (one id column); use slk_spans_batch_wide");
  if (!offsets || !out_span_offsets || (R && !bases)) return fail(SLK_E_INVALID, "null argument");
  if ((mate_bases == nullptr) != (mate_offsets == nullptr))
    return fail(SLK_E_INVALID, "mate_bases and mate_offsets must be given together");
  rc = set_device(ix);
  if (rc) return rc;
  out_span_offsets[0] = 0;
  if (R == 0) return SLK_OK;
  uint64_t total, mate_total;
  rc = upload_reads(st, bases, offsets, mate_bases, mate_offsets, R, &total, &mate_total);
  if (rc) return rc;
  bool paired = mate_offsets != nullptr;
  rc = ensure_scratch(st, span_slots(total, mate_total, R, paired), R);
  if (rc) return rc;
  const uint64_t *d_off = st->offsets.as<uint64_t>();
  const uint64_t *d_moff = paired ? st->mate_offsets.as<uint64_t>() : nullptr;
  if (ix->W > 1) {
    launch_wide_scan(ix->wp, st->bases.as<uint8_t>(), d_off, paired ? st->mate_bases.as<uint8_t>() : nullptr, d_moff, R,
                     st->span_keys.as<uint64_t>(), st->span_meta.as<int32_t>(), st->span_count.as<int32_t>(), st->s);
  } else if (use_fused(ix)) {
    FusedArgs A{};
    A.P = ix->sp; A.bases = st->bases.as<uint8_t>(); A.offsets = d_off;
    A.mate_bases = paired ? st->mate_bases.as<uint8_t>() : nullptr; A.mate_offsets = d_moff; A.R = R;
    A.span_keys = st->span_keys.as<uint64_t>(); A.span_meta = st->span_meta.as<int32_t>();
    A.span_count = st->span_count.as<int32_t>(); A.status = st->d_status;
    launch_fused(MODE_SPANS, A, st->s);
  } else {
    launch_scan(ix->sp, st->bases.as<uint8_t>(), d_off, paired ? st->mate_bases.as<uint8_t>() : nullptr, d_moff, R,
                st->span_keys.as<uint64_t>(), st->span_meta.as<int32_t>(), st->span_count.as<int32_t>(), st->s);
  }
  HIPCHK(hipGetLastError());
  rc = counts_to_offsets(st, st->span_count.as<int32_t>(), R, out_span_offsets, spans_capacity);
  if (rc) return rc;
  uint64_t n = out_span_offsets[R];
  if (n) {
    if (!out_spans) return fail(SLK_E_INVALID, "out_spans is null");
    HIPCHK(st->out_items.ensure(n * sizeof(slk_span)));
    if (ix->W > 1) {
      HIPCHK(st->out_taxon.ensure(n * 8 * ix->W));   // (free here: this entry classifies nothing)
      launch_wide_gather_spans(ix->W, d_off, d_moff, R, st->span_keys.as<uint64_t>(), st->span_meta.as<int32_t>(), st->out_offsets.as<uint64_t>(),
                               st->out_items.p, st->out_taxon.as<int64_t>(), st->s);
    } else {
      launch_gather_spans(d_off, d_moff, R, st->span_keys.as<uint64_t>(), st->span_meta.as<int32_t>(),
                          st->out_offsets.as<uint64_t>(), st->out_items.p, st->s);
    }
    HIPCHK(hipGetLastError());
    rc = copy_out(st, out_spans, st->out_items.p, n * sizeof(slk_span));
    if (!rc && ix->W > 1) rc = copy_out(st, out_keys, st->out_taxon.p, n * 8 * ix->W);
    if (rc) return rc;
    if (ix->W == 1 && out_keys)
      for (uint64_t i = 0; i < n; i++) out_keys[i] = out_spans[i].key;
  }
  HIPCHK(hipStreamSynchronize(st->s));
  return SLK_OK;
}

int32_t slk_spans_batch(slk_index *ix, slk_stream *st, const uint8_t *bases, const uint64_t *offsets,
                        const uint8_t *mate_bases, const uint64_t *mate_offsets, uint64_t R,
                        uint64_t *out_span_offsets, slk_span *out_spans, uint64_t spans_capacity) {
  return spans_batch(ix, st, bases, offsets, mate_bases, mate_offsets, R, out_span_offsets, out_spans, nullptr, spans_capacity);
}

int32_t slk_spans_batch_wide(slk_index *ix, slk_stream *st, const uint8_t *bases, const uint64_t *offsets,
                             const uint8_t *mate_bases, const uint64_t *mate_offsets, uint64_t R,
                             uint64_t *out_span_offsets, slk_span *out_spans, int64_t *out_keys, uint64_t spans_capacity) {
  if (!out_keys && spans_capacity) return fail(SLK_E_INVALID, "out_keys is null");
  return spans_batch(ix, st, bases, offsets, mate_bases, mate_offsets, R, out_span_offsets, out_spans, out_keys, spans_capacity);
}

// The reads of a host call: ASCII (bases / mate_bases) or the engine's 3-bit form (host/pack.hpp: 2-bit codes and validity bits,
// 16 bases per word, positions as in the ASCII concatenation).  Packed reads are unpacked on the device, behind their upload, into
// the stream's ASCII buffers -- 6 bytes over the link per 16 bases instead of 16 --, so every kernel of the path reads them as it
// reads text.
struct ReadSource {
  const uint8_t *bases = nullptr, *mate_bases = nullptr;
  const uint32_t *codes = nullptr, *mate_codes = nullptr;
  const uint16_t *valid = nullptr, *mate_valid = nullptr;
  bool packed() const { return codes != nullptr; }
};

// bases [p0, p1) of one mate from the caller's memory to dst (+ the device-side unpack on `run` for packed reads), ordered on `up`
static int32_t upload_range(slk_stream *st, Staging *g, hipStream_t up, hipStream_t run, hipEvent_t ev, bool packed, const uint8_t *bases,
                            const uint32_t *codes, const uint16_t *valid, DevBuf &d_codes, DevBuf &d_valid, uint8_t *dst, uint64_t p0, uint64_t p1) {
  if (p1 <= p0) return SLK_OK;
  if (!packed) return copy_in(g, up, dst + p0, bases + p0, p1 - p0);
  const uint64_t w0 = p0 / 16, w1 = (p1 + 15) / 16;
  int32_t rc = copy_in(g, up, d_codes.as<uint32_t>() + w0, codes + w0, (w1 - w0) * 4);
  if (!rc) rc = copy_in(g, up, d_valid.as<uint16_t>() + w0, valid + w0, (w1 - w0) * 2);
  if (rc) return rc;
  if (up != run) {
    HIPCHK(hipEventRecord(ev, up));
    HIPCHK(hipStreamWaitEvent(run, ev, 0));
  }
  // (whole words: a word that straddles two ranges is unpacked by both, to the same bytes, in stream order)
  launch_unpack_bases(d_codes.as<uint32_t>(), d_valid.as<uint16_t>(), w0, w1, dst, run);
  HIPCHK(hipGetLastError());
  return SLK_OK;
}

static int32_t classify_batch_host(slk_index *ix, slk_stream *st, const ReadSource &src, const uint64_t *offsets, const uint64_t *mate_offsets,
                                   uint64_t R, int32_t min_hit_groups, const double *thresholds, int32_t C, int32_t *out_taxon,
                                   uint8_t *out_classified, int32_t *out_num_distinct, int32_t *out_total_kmers, uint64_t *out_hit_offsets,
                                   slk_hit *out_hits, uint64_t hits_capacity) {
  int32_t rc = check_ready(ix, st, true);
  if (rc) return rc;
  const bool pk = src.packed();
  if (!offsets || (R && ((!pk && !src.bases) || (pk && !src.valid) || !out_taxon || !out_classified))) return fail(SLK_E_INVALID, "null argument");
  if (C < 1 || C > MAX_THRESHOLDS || !thresholds) return fail(SLK_E_INVALID, "need 1..%d thresholds", MAX_THRESHOLDS);
  const bool paired = mate_offsets != nullptr;
  if (paired != (pk ? (src.mate_codes != nullptr && src.mate_valid != nullptr) : src.mate_bases != nullptr) ||
      (!paired && (src.mate_codes || src.mate_valid || src.mate_bases)))
    return fail(SLK_E_INVALID, "the second mates' bases and mate_offsets must be given together");
  rc = set_device(ix);
  if (rc) return rc;
  if (out_hit_offsets) out_hit_offsets[0] = 0;
  if (R == 0) return SLK_OK;
  static const bool call_timing = getenv("SLK_DEBUG_CALL_TIMING") != nullptr;  // tuning aid: wall clock of the phases of a call
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double tp[6] = {now(), 0, 0, 0, 0, 0};
  uint64_t total = 0, mate_total = 0;
  const bool want_hits = out_hit_offsets != nullptr && out_hits != nullptr;
  bool early_down = false;
  HIPCHK(st->out_taxon.ensure((size_t)C * R * 4));
  HIPCHK(st->out_cls.ensure((size_t)C * R));
  HIPCHK(st->out_nd.ensure(R * 4));
  HIPCHK(st->out_tk.ensure(R * 4));
  HIPCHK(st->out_nh.ensure(R * 4));
  rc = validate_reads(offsets, mate_offsets, R);
  if (rc) return rc;
  total = offsets[R];
  mate_total = paired ? mate_offsets[R] : 0;
  // (packed reads are unpacked in whole words of 16 bases: the ASCII buffers hold the last word in full)
  HIPCHK(st->bases.ensure((total + 15) / 16 * 16));
  HIPCHK(st->offsets.ensure((R + 1) * 8));
  if (paired) {
    HIPCHK(st->mate_bases.ensure((mate_total + 15) / 16 * 16));
    HIPCHK(st->mate_offsets.ensure((R + 1) * 8));
  }
  if (pk) {
    HIPCHK(st->pk_codes.ensure((total + 15) / 16 * 4 + 4));
    HIPCHK(st->pk_valid.ensure((total + 15) / 16 * 2 + 2));
    if (paired) {
      HIPCHK(st->pk_mate_codes.ensure((mate_total + 15) / 16 * 4 + 4));
      HIPCHK(st->pk_mate_valid.ensure((mate_total + 15) / 16 * 2 + 2));
    }
    if (!st->ev_unpack) HIPCHK(hipEventCreateWithFlags(&st->ev_unpack, hipEventDisableTiming));
  }
  // A large call is cut into sub-batches: the reads of sub-batch i+1 go up (on a second stream) while the kernels of
  // sub-batch i run, so the call costs its upload plus ONE sub-batch of kernel time.  With hit lists too: the sub-batches leave
  // their spans in the batch's span arrays (run_classify: span_shift) and the lists are put together for the whole batch at the end.
  const char *sub_env = getenv("SLK_HOST_SUBBATCH");  // (read per call, so that tests can move it)
  // (2^19 reads: measured from pinned memory, 4 M reads of 150 bp -- packed 353 / 576 / 643 / 623 / 403 M reads/s at 2^17 .. 2^21, text
  //  313 / 324 / 327 / 315 / 246: smaller pieces pay per copy -- a sub-batch is five to nine DMAs --, larger ones leave the last
  //  piece's kernels exposed; profiles/r04_packed_entry.json)
  const uint64_t SUB = sub_env ? (uint64_t)std::max(1L, atol(sub_env)) : (uint64_t)1 << 19;
  if (use_fused(ix) && R >= 2 * SUB) {
    if (want_hits) {   // (once, for the whole batch: a sub-batch must not move the arrays under the kernels of the one before)
      rc = ensure_scratch(st, span_slots(total, mate_total, R, paired), R);
      if (rc) return rc;
    }
    if (!st->cs) HIPCHK(hipStreamCreateWithFlags(&st->cs, hipStreamNonBlocking));
    if (!st->ds) HIPCHK(hipStreamCreateWithFlags(&st->ds, hipStreamNonBlocking));
    const uint64_t nsub = (R + SUB - 1) / SUB;
    while (st->up_ev.size() < nsub) {
      hipEvent_t e;
      HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      st->up_ev.push_back(e);
    }
    while (st->dn_ev.size() < nsub) {
      hipEvent_t e;
      HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      st->dn_ev.push_back(e);
    }
    // Result buffers the library can DMA into take their rows sub-batch by sub-batch, beside the next sub-batch's kernels (the link
    // is full duplex: the rows come down while the reads go up); pageable ones are filled at the end, through the staging buffers.
    early_down = pinned().covers(out_taxon, (size_t)C * R * 4) && pinned().covers(out_classified, (size_t)C * R) &&
                 (!out_num_distinct || pinned().covers(out_num_distinct, R * 4)) && (!out_total_kmers || pinned().covers(out_total_kmers, R * 4));
    st->reran = false;
    for (uint64_t i = 0; i < nsub; i++) {
      const uint64_t r0 = i * SUB, r1 = std::min(R, r0 + SUB), n = r1 - r0;
      // (the offsets travel with their sub-batch: 8 bytes per read are a tenth of a packed batch, and the first kernels should not
      //  wait for all of them)
      rc = copy_in(&st->staging_c, st->cs, st->offsets.as<uint64_t>() + r0, offsets + r0, (n + 1) * 8);
      if (!rc && paired) rc = copy_in(&st->staging_c, st->cs, st->mate_offsets.as<uint64_t>() + r0, mate_offsets + r0, (n + 1) * 8);
      if (rc) return rc;
      rc = upload_range(st, &st->staging_c, st->cs, st->s, st->ev_unpack, pk, src.bases, src.codes, src.valid, st->pk_codes, st->pk_valid,
                        st->bases.as<uint8_t>(), offsets[r0], offsets[r1]);
      if (!rc && paired)
        rc = upload_range(st, &st->staging_c, st->cs, st->s, st->ev_unpack, pk, src.mate_bases, src.mate_codes, src.mate_valid, st->pk_mate_codes,
                          st->pk_mate_valid, st->mate_bases.as<uint8_t>(), mate_offsets[r0], mate_offsets[r1]);
      if (rc) return rc;
      HIPCHK(hipEventRecord(st->up_ev[i], st->cs));
      HIPCHK(hipStreamWaitEvent(st->s, st->up_ev[i], 0));
      rc = run_classify(ix, st, st->bases.as<uint8_t>(), st->offsets.as<uint64_t>() + r0, paired ? st->mate_bases.as<uint8_t>() : nullptr,
                        // (the offsets stay absolute, so "total bases" is where this sub-batch ENDS: it sizes the span scratch of
                        //  the unbounded re-run, whose regions are addressed by those offsets)
                        paired ? st->mate_offsets.as<uint64_t>() + r0 : nullptr, n, offsets[r1],
                        paired ? mate_offsets[r1] : 0, min_hit_groups, thresholds, C, st->out_taxon.as<int32_t>() + r0,
                        st->out_cls.as<uint8_t>() + r0, st->out_nd.as<int32_t>() + r0, st->out_tk.as<int32_t>() + r0,
                        st->out_nh.as<int32_t>() + r0, nullptr, want_hits, R, want_hits && paired ? r0 : 0);
      if (rc) return rc;
      if (early_down) {
        HIPCHK(hipEventRecord(st->dn_ev[i], st->s));
        HIPCHK(hipStreamWaitEvent(st->ds, st->dn_ev[i], 0));
        for (int32_t c = 0; c < C; c++) {
          HIPCHK(hipMemcpyAsync(out_taxon + (size_t)c * R + r0, st->out_taxon.as<int32_t>() + (size_t)c * R + r0, n * 4, hipMemcpyDeviceToHost, st->ds));
          HIPCHK(hipMemcpyAsync(out_classified + (size_t)c * R + r0, st->out_cls.as<uint8_t>() + (size_t)c * R + r0, n, hipMemcpyDeviceToHost, st->ds));
        }
        if (out_num_distinct) HIPCHK(hipMemcpyAsync(out_num_distinct + r0, st->out_nd.as<int32_t>() + r0, n * 4, hipMemcpyDeviceToHost, st->ds));
        if (out_total_kmers) HIPCHK(hipMemcpyAsync(out_total_kmers + r0, st->out_tk.as<int32_t>() + r0, n * 4, hipMemcpyDeviceToHost, st->ds));
      }
    }
    HIPCHK(hipStreamSynchronize(st->cs));  // (the caller's buffers are free from here on)
    if (call_timing) tp[1] = now();
  } else {
    rc = upload_range(st, &st->staging, st->s, st->s, st->ev_unpack, pk, src.bases, src.codes, src.valid, st->pk_codes, st->pk_valid,
                      st->bases.as<uint8_t>(), 0, total);
    if (!rc) rc = copy_in(st, st->offsets.p, offsets, (R + 1) * 8);
    if (!rc && paired) {
      rc = upload_range(st, &st->staging, st->s, st->s, st->ev_unpack, pk, src.mate_bases, src.mate_codes, src.mate_valid, st->pk_mate_codes,
                        st->pk_mate_valid, st->mate_bases.as<uint8_t>(), 0, mate_total);
      if (!rc) rc = copy_in(st, st->mate_offsets.p, mate_offsets, (R + 1) * 8);
    }
    if (rc) return rc;
    if (call_timing) { (void)hipStreamSynchronize(st->s); tp[1] = now(); }
    rc = run_classify(ix, st, st->bases.as<uint8_t>(), st->offsets.as<uint64_t>(), paired ? st->mate_bases.as<uint8_t>() : nullptr,
                      paired ? st->mate_offsets.as<uint64_t>() : nullptr, R, total, mate_total, min_hit_groups, thresholds, C,
                      st->out_taxon.as<int32_t>(), st->out_cls.as<uint8_t>(), st->out_nd.as<int32_t>(), st->out_tk.as<int32_t>(),
                      st->out_nh.as<int32_t>(), nullptr, want_hits);
    if (rc) return rc;
  }
  const uint64_t *d_off = st->offsets.as<uint64_t>();
  const uint64_t *d_moff = paired ? st->mate_offsets.as<uint64_t>() : nullptr;
  HIPCHK(hipStreamSynchronize(st->s));
  tp[2] = now();
  rc = check_status(st);  // (re-runs the batch through the unbounded path if a taxon map overflowed)
  if (rc) return rc;
  if (early_down) HIPCHK(hipStreamSynchronize(st->ds));
  if (!early_down || st->reran) {   // (rows that came down early are stale if the batch was classified again)
    rc = copy_out(st, out_taxon, st->out_taxon.p, (size_t)C * R * 4);
    if (!rc) rc = copy_out(st, out_classified, st->out_cls.p, (size_t)C * R);
    if (!rc && out_num_distinct) rc = copy_out(st, out_num_distinct, st->out_nd.p, R * 4);
    if (!rc && out_total_kmers) rc = copy_out(st, out_total_kmers, st->out_tk.p, R * 4);
    if (rc) return rc;
  }
  if (call_timing) { (void)hipStreamSynchronize(st->s); tp[3] = now(); }
  double th[3] = {0, 0, 0};
  if (out_hit_offsets) {
    const bool merged = st->merged_hits && want_hits;
    if (merged) {   // (the merged lists' lengths first: span_count is free once the kernels are through)
      launch_merged_hits(true, d_off, d_moff, R, st->span_meta.as<int32_t>(), st->span_taxon.as<int32_t>(), st->out_nh.as<int32_t>(), nullptr,
                         st->span_count.as<int32_t>(), nullptr, st->s);
      HIPCHK(hipGetLastError());
    }
    rc = counts_to_offsets(st, merged ? st->span_count.as<int32_t>() : st->out_nh.as<int32_t>(), R, out_hit_offsets, out_hits ? hits_capacity : ~0ULL);
    if (rc) return rc;
    if (call_timing) th[0] = now();
    uint64_t n = out_hit_offsets[R];
    if (n && out_hits) {
      HIPCHK(st->out_items.ensure(n * sizeof(slk_hit)));
      if (merged)
        launch_merged_hits(false, d_off, d_moff, R, st->span_meta.as<int32_t>(), st->span_taxon.as<int32_t>(), st->out_nh.as<int32_t>(),
                           st->out_offsets.as<uint64_t>(), nullptr, st->out_items.p, st->s);
      else
        launch_gather_hits(d_off, d_moff, R, st->span_meta.as<int32_t>(), st->span_taxon.as<int32_t>(),
                           st->out_offsets.as<uint64_t>(), st->out_items.p, st->s);
      HIPCHK(hipGetLastError());
      if (call_timing) { (void)hipStreamSynchronize(st->s); th[1] = now(); }
      rc = copy_out(st, out_hits, st->out_items.p, n * sizeof(slk_hit));
      if (rc) return rc;
      if (call_timing) th[2] = now();
    }
  }
  HIPCHK(hipStreamSynchronize(st->s));
  if (call_timing)
    fprintf(stderr, "slk_classify_batch%s R=%llu: upload %.2f ms, kernels %.2f, results %.2f, hit lists %.2f (offsets %.2f, gather %.2f, download %.2f)\n",
            pk ? "_packed" : "", (unsigned long long)R, tp[1] - tp[0], tp[2] - tp[1], tp[3] - tp[2], now() - tp[3], th[0] ? th[0] - tp[3] : 0.0,
            th[1] ? th[1] - th[0] : 0.0, th[2] ? th[2] - th[1] : 0.0);
  return check_status(st);
}

int32_t slk_classify_batch(slk_index *ix, slk_stream *st, const uint8_t *bases, const uint64_t *offsets,
                           const uint8_t *mate_bases, const uint64_t *mate_offsets, uint64_t R,
                           int32_t min_hit_groups, const double *thresholds, int32_t C, int32_t *out_taxon,
                           uint8_t *out_classified, int32_t *out_num_distinct, int32_t *out_total_kmers,
                           uint64_t *out_hit_offsets, slk_hit *out_hits, uint64_t hits_capacity) {
  if ((mate_bases == nullptr) != (mate_offsets == nullptr)) return fail(SLK_E_INVALID, "mate_bases and mate_offsets must be given together");
  ReadSource src;
  src.bases = bases; src.mate_bases = mate_bases;
  return classify_batch_host(ix, st, src, offsets, mate_offsets, R, min_hit_groups, thresholds, C, out_taxon, out_classified, out_num_distinct,
                             out_total_kmers, out_hit_offsets, out_hits, hits_capacity);
}

// The same call with the reads in the engine's 3-bit form (host/pack.hpp; slk_pack_bases makes it): 6 bytes per 16 bases over the
// link instead of 16.  InputFragment.nucleotides (S/kmers/minimizer/MinSplitter.scala:31-32) already encoded as
// BitRepresentation.charToTwobit would (S/kmers/util/BitRepresentation.scala:127-135), with the isValid test (:140-143) as a bit.
int32_t slk_classify_batch_packed(slk_index *ix, slk_stream *st, const uint32_t *codes, const uint16_t *valid, const uint64_t *offsets,
                                  const uint32_t *mate_codes, const uint16_t *mate_valid, const uint64_t *mate_offsets, uint64_t R,
                                  int32_t min_hit_groups, const double *thresholds, int32_t C, int32_t *out_taxon,
                                  uint8_t *out_classified, int32_t *out_num_distinct, int32_t *out_total_kmers,
                                  uint64_t *out_hit_offsets, slk_hit *out_hits, uint64_t hits_capacity) {
  if (R && (!codes || !valid)) return fail(SLK_E_INVALID, "null argument");
  if ((mate_codes == nullptr) != (mate_offsets == nullptr) || (mate_valid == nullptr) != (mate_offsets == nullptr))
    return fail(SLK_E_INVALID, "mate_codes, mate_valid and mate_offsets must be given together");
  ReadSource src;
  src.codes = codes; src.valid = valid; src.mate_codes = mate_codes; src.mate_valid = mate_valid;
  if (R == 0) { static const uint32_t z = 0; src.codes = &z; }   // (an empty batch is a packed one all the same)
  return classify_batch_host(ix, st, src, offsets, mate_offsets, R, min_hit_groups, thresholds, C, out_taxon, out_classified, out_num_distinct,
                             out_total_kmers, out_hit_offsets, out_hits, hits_capacity);
}

// n bases -> codes[ceil(n / 16)], valid[ceil(n / 16)] (host/pack.hpp), on the library's copy threads.  Host arithmetic: no GPU.
int32_t slk_pack_bases(const uint8_t *bases, uint64_t n, uint32_t *codes, uint16_t *valid) {
  if (n && (!bases || !codes || !valid)) return fail(SLK_E_INVALID, "null argument");
  const uint64_t SLICE = (uint64_t)1 << 22;   // (a multiple of 32 bases: slices start on word borders)
  const uint64_t parts = (n + SLICE - 1) / SLICE;
  if (parts <= 1) { pack_bases(bases, n, codes, valid); return SLK_OK; }
  host_pool().parallel_for(parts, [&](size_t i) {
    const uint64_t a = i * SLICE, b = std::min(n, a + SLICE);
    pack_bases(bases + a, b - a, codes + a / 16, valid + a / 16);
  });
  return SLK_OK;
}

}  // extern "C"
