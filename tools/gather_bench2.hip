// gather_bench2.hip -- round-2 experiments on the random-line ceiling of the table probe:
//   size    rate of uniformly random 64-B line reads against the table size (is 56 -> 48.5 G lines/s the Infinity Cache's hit
//           share or address translation?)
//   pages   the same number of random lines, but only the first F bytes of every 2 MiB page of a 128 GiB allocation are ever
//           touched: the data footprint is small (cache resident) while the translation footprint is the whole allocation --
//           separates the TLB from the DRAM
//   binned  probes pre-binned by bucket prefix: a workgroup takes one bin at a time and reads random lines inside it
//           (what a per-launch binning stage between queue and probe would present to the memory system)
//   wide    128-B and 256-B aligned requests (8 / 16 lanes x 16 B) against 64-B ones at equal request count
//   usage: gather_bench2 <table GiB> <experiment> [requests in millions]
// All patterns read with the probe's access shape (LPP lanes x 16 B per request, 64 / LPP requests per wave instruction).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x;
}

struct Pattern {
  int kind;             // 0 uniform, 1 pages, 2 binned
  uint64_t req_mask;    // number of request-sized units in the table - 1
  // pages: unit = (page << page_shift) | (rnd & in_page_mask)
  int page_shift;       // log2(units per page)
  uint64_t page_mask, in_page_mask;
  // binned: the table is cut into nbins = 2^bin_bits ranges of 2^unit_bits units; a block reads `per_bin` requests per thread
  // group from one bin, then moves on to bin + gridDim.x
  int bin_bits, unit_bits, per_bin;
};

template <int LPP, int UNROLL>
__global__ void __launch_bounds__(256) gather(const uint4 *__restrict__ t, Pattern P, int iters, uint64_t *out) {
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t grp = tid / LPP;
  const uint32_t c = (uint32_t)(tid % LPP);
  uint64_t acc = 0;
  for (int it = 0; it < iters; it += UNROLL) {
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      const uint64_t r = mix(grp * 1315423911ULL + (uint64_t)(it + u) * 0x9E3779B97F4A7C15ULL);
      uint64_t unit;
      if (P.kind == 0) {
        unit = r & P.req_mask;
      } else if (P.kind == 1) {
        unit = ((r & P.page_mask) << P.page_shift) | ((r >> 40) & P.in_page_mask);
      } else {
        const uint64_t bin = ((uint64_t)blockIdx.x + (uint64_t)((it + u) / P.per_bin) * gridDim.x) & ((1ULL << P.bin_bits) - 1);
        unit = (bin << P.unit_bits) | (r & ((1ULL << P.unit_bits) - 1));
      }
      uint4 a = t[unit * LPP + c];
      acc += a.x ^ a.y ^ a.z ^ a.w;
    }
  }
  if (acc == 0x1234567) out[0] = acc;
}

static const uint4 *g_t;
static uint64_t *g_out;
static double g_mreq = 2000.0;

template <int LPP, int UNROLL>
double run(const char *name, Pattern P, int blocks_per_cu = 8) {
  int blocks = 256 * blocks_per_cu;
  uint64_t threads = (uint64_t)blocks * 256;
  int iters = (int)(g_mreq * 1e6 * LPP / threads);
  iters = (iters / UNROLL + 1) * UNROLL;
  double reqs = (double)threads / LPP * iters;
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  hipLaunchKernelGGL((gather<LPP, UNROLL>), dim3(blocks), dim3(256), 0, 0, g_t, P, UNROLL, g_out);  // warm
  CHECK(hipEventRecord(a));
  hipLaunchKernelGGL((gather<LPP, UNROLL>), dim3(blocks), dim3(256), 0, 0, g_t, P, iters, g_out);
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms; CHECK(hipEventElapsedTime(&ms, a, b));
  double g = reqs / ms / 1e6;
  printf("  %-58s %3d B/request: %7.2f G requests/s = %6.2f TB/s  (%.1f ms)\n", name, LPP * 16, g, reqs * LPP * 16 / ms / 1e9, ms);
  fflush(stdout);
  CHECK(hipEventDestroy(a)); CHECK(hipEventDestroy(b));
  return g;
}

int main(int argc, char **argv) {
  double gib = argc > 1 ? atof(argv[1]) : 1.0;
  const char *exp = argc > 2 ? argv[2] : "size";
  if (argc > 3) g_mreq = atof(argv[3]);
  uint64_t nlines = 1; while ((double)(nlines * 2) * 64 <= gib * (1ULL << 30)) nlines *= 2;
  size_t bytes = nlines * 64;
  uint4 *t;
  CHECK(hipMalloc(&t, bytes)); CHECK(hipMalloc(&g_out, 64));
  CHECK(hipMemset(t, 1, bytes)); CHECK(hipDeviceSynchronize());
  g_t = t;
  printf("allocation %.3f GiB (%llu lines of 64 B), experiment %s\n", bytes / double(1ULL << 30), (unsigned long long)nlines, exp);
  int lb = 0; while ((1ULL << lb) < nlines) lb++;

  if (!strcmp(exp, "size")) {
    // uniform random 64-B lines over the first S bytes of the allocation
    for (int sb = 21; sb <= lb; sb++) {  // 128 MiB ...
      Pattern P{}; P.kind = 0; P.req_mask = (1ULL << sb) - 1;
      char nm[96]; snprintf(nm, sizeof nm, "uniform over %8.3f GiB", (double)(64ULL << sb) / (1ULL << 30));
      run<4, 8>(nm, P);
    }
  } else if (!strcmp(exp, "pages")) {
    // every 2 MiB page of the allocation is touched, but only its first F bytes
    const int page_lines_bits = 15;  // 2 MiB / 64 B
    for (int fb : {0, 2, 4, 6, 8, 10, 12, 15}) {  // F = 64 B ... 2 MiB
      Pattern P{}; P.kind = 1; P.page_shift = page_lines_bits; P.page_mask = (nlines >> page_lines_bits) - 1;
      P.in_page_mask = (1ULL << fb) - 1;
      double foot = (double)(nlines >> page_lines_bits) * (64ULL << fb) / (1ULL << 30);
      char nm[96]; snprintf(nm, sizeof nm, "first %7llu B of each 2 MiB page (data footprint %8.3f GiB)", 64ULL << fb, foot);
      run<4, 8>(nm, P);
    }
    // control: the same data footprints as one dense range (translation footprint = data footprint)
    for (int fb : {0, 2, 4, 6, 8, 10, 12, 15}) {
      uint64_t lines = (nlines >> page_lines_bits) << fb;
      Pattern P{}; P.kind = 0; P.req_mask = lines - 1;
      char nm[96]; snprintf(nm, sizeof nm, "dense range of %8.3f GiB", (double)lines * 64 / (1ULL << 30));
      run<4, 8>(nm, P);
    }
  } else if (!strcmp(exp, "binned")) {
    Pattern U{}; U.kind = 0; U.req_mask = nlines - 1;
    run<4, 8>("uniform (control)", U);
    // 2^bb bins; per visit a block reads per_bin * 64 requests from its bin (256 threads / 4 lanes per request)
    for (int bb : {12, 14, 16, 18, 20, 22, 24}) {
      if (bb >= lb) continue;
      for (int per_bin : {8, 64, 512}) {
        Pattern P{}; P.kind = 2; P.bin_bits = bb; P.unit_bits = lb - bb; P.per_bin = per_bin;
        char nm[96];
        snprintf(nm, sizeof nm, "2^%d bins of %9.1f KiB, %5d requests per bin visit", bb, (double)(64ULL << (lb - bb)) / 1024, per_bin * 64);
        run<4, 8>(nm, P);
      }
    }
  } else if (!strcmp(exp, "binned1")) {
    // What a binning stage would present for ONE launch of the classify kernel: `g_mreq` million probes spread evenly over the
    // bins, every bin visited exactly once by one workgroup (requests per visit = probes / bins, rounded up to the 64 request
    // groups of a block).  The probes of a launch touch a line at most once (4e8 probes, 2.1e9 lines), so L2 hits play no part.
    Pattern U{}; U.kind = 0; U.req_mask = nlines - 1;
    run<4, 1>("uniform (control)", U);
    for (int bb = 10; bb <= 22; bb += 2) {
      if (bb >= lb) continue;
      double per_visit = g_mreq * 1e6 / (double)(1ULL << bb);
      int per_bin = (int)(per_visit / 64 + 0.999);
      if (per_bin < 1) per_bin = 1;
      Pattern P{}; P.kind = 2; P.bin_bits = bb; P.unit_bits = lb - bb; P.per_bin = per_bin;
      char nm[96];
      snprintf(nm, sizeof nm, "2^%d bins of %9.1f KiB, %6d requests per bin, once", bb, (double)(64ULL << (lb - bb)) / 1024, per_bin * 64);
      // blocks x iters chosen so that blocks * (iters / per_bin) == number of bins
      int blocks = 2048;
      long visits_per_block = (long)((1ULL << bb) / blocks); if (visits_per_block < 1) { visits_per_block = 1; blocks = 1 << bb; }
      int iters = (int)(visits_per_block * per_bin);
      double reqs = (double)blocks * 64 * iters;
      hipEvent_t a, b;
      CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
      hipLaunchKernelGGL((gather<4, 1>), dim3(blocks), dim3(256), 0, 0, g_t, P, 1, g_out);
      CHECK(hipEventRecord(a));
      hipLaunchKernelGGL((gather<4, 1>), dim3(blocks), dim3(256), 0, 0, g_t, P, iters, g_out);
      CHECK(hipEventRecord(b));
      CHECK(hipEventSynchronize(b));
      float ms; CHECK(hipEventElapsedTime(&ms, a, b));
      printf("  %-58s  64 B/request: %7.2f G requests/s = %6.2f TB/s  (%.1f ms, %.0f M requests)\n", nm, reqs / ms / 1e6, reqs * 64 / ms / 1e9, ms, reqs / 1e6);
      fflush(stdout);
    }
  } else if (!strcmp(exp, "wide")) {
    Pattern P{}; P.kind = 0;
    P.req_mask = nlines - 1;      run<4, 8>("64-B aligned requests (4 lanes x 16 B)", P);
    P.req_mask = nlines / 2 - 1;  run<8, 8>("128-B aligned requests (8 lanes x 16 B)", P);
    P.req_mask = nlines / 4 - 1;  run<16, 8>("256-B aligned requests (16 lanes x 16 B)", P);
    P.req_mask = nlines / 8 - 1;  run<32, 8>("512-B aligned requests (32 lanes x 16 B)", P);
    P.req_mask = nlines / 16 - 1; run<64, 8>("1024-B aligned requests (64 lanes x 16 B)", P);
  } else {
    printf("unknown experiment\n");
    return 2;
  }
  return 0;
}
