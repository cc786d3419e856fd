// gather_bench.hip -- random 64-byte-bucket gather microbenchmark over an HBM-resident table (design input for the
// probe stage and the "practical ceiling" SURVEY.md 8d asks to report next to the 8 TB/s roofline).
//   usage: gather_bench <table GiB> [probes in millions]
// Patterns: A = one lane reads 8 B of a random bucket; B = one lane reads the whole 64-B bucket (4 x 16 B);
//           C = 8 lanes read one bucket cooperatively (8 B each); D = 4 lanes x 16 B.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x;
}

template <int MODE, int UNROLL>
__global__ void __launch_bounds__(256) gather(const uint64_t *__restrict__ t, uint64_t bucket_mask, int iters, uint64_t *out) {
  uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t acc = 0;
  for (int it = 0; it < iters; it += UNROLL) {
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      if (MODE == 0) {
        uint64_t b = mix(tid * 1315423911ULL + it + u) & bucket_mask;
        acc += t[b * 8 + (tid & 7)];
      } else if (MODE == 1) {
        uint64_t b = mix(tid * 1315423911ULL + it + u) & bucket_mask;
        const ulonglong2 *p = (const ulonglong2 *)(t + b * 8);
        ulonglong2 a0 = p[0], a1 = p[1], a2 = p[2], a3 = p[3];
        acc += a0.x ^ a0.y ^ a1.x ^ a1.y ^ a2.x ^ a2.y ^ a3.x ^ a3.y;
      } else if (MODE == 2) {
        uint64_t b = mix((tid >> 3) * 1315423911ULL + it + u) & bucket_mask;
        acc += t[b * 8 + (tid & 7)];
      } else {
        uint64_t b = mix((tid >> 2) * 1315423911ULL + it + u) & bucket_mask;
        ulonglong2 a = ((const ulonglong2 *)(t + b * 8))[tid & 3];
        acc += a.x ^ a.y;
      }
    }
  }
  if (acc == 0x1234567) out[0] = acc;
}

template <int MODE, int UNROLL>
void run(const char *name, const uint64_t *t, uint64_t nb, uint64_t *out, double mprobes, int blocks_per_cu) {
  int blocks = 256 * blocks_per_cu;
  uint64_t threads = (uint64_t)blocks * 256;
  int lanes_per_probe = MODE == 2 ? 8 : (MODE == 3 ? 4 : 1);
  int iters = (int)(mprobes * 1e6 * lanes_per_probe / threads);
  iters = (iters / UNROLL + 1) * UNROLL;
  double probes = (double)threads / lanes_per_probe * iters;
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  hipLaunchKernelGGL((gather<MODE, UNROLL>), dim3(blocks), dim3(256), 0, 0, t, nb - 1, UNROLL, out);  // warm
  CHECK(hipEventRecord(a));
  hipLaunchKernelGGL((gather<MODE, UNROLL>), dim3(blocks), dim3(256), 0, 0, t, nb - 1, iters, out);
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms; CHECK(hipEventElapsedTime(&ms, a, b));
  printf("  %-28s unroll %d  %2d blk/CU: %7.2f G probes/s  = %6.2f TB/s of 64-B buckets  (%.1f ms)\n", name, UNROLL,
         blocks_per_cu, probes / ms / 1e6, probes * 64 / ms / 1e9, ms);
  fflush(stdout);
}

int main(int argc, char **argv) {
  double gib = argc > 1 ? atof(argv[1]) : 1.0;
  double mprobes = argc > 2 ? atof(argv[2]) : 2000.0;
  uint64_t nb = 1; while ((double)(nb * 2) * 64 <= gib * (1ULL << 30)) nb *= 2;
  size_t bytes = nb * 64;
  uint64_t *t, *out;
  CHECK(hipMalloc(&t, bytes)); CHECK(hipMalloc(&out, 64));
  CHECK(hipMemset(t, 1, bytes)); CHECK(hipDeviceSynchronize());
  printf("table %.1f GiB (%llu buckets)\n", bytes / double(1ULL << 30), (unsigned long long)nb);
  for (int bpc : {4, 8}) {
    run<0, 1>("A lane x 8B", t, nb, out, mprobes, bpc);
    run<0, 4>("A lane x 8B", t, nb, out, mprobes, bpc);
    run<1, 1>("B lane x 64B", t, nb, out, mprobes, bpc);
    run<1, 4>("B lane x 64B", t, nb, out, mprobes, bpc);
    run<2, 4>("C 8 lanes x 8B", t, nb, out, mprobes, bpc);
    run<2, 8>("C 8 lanes x 8B", t, nb, out, mprobes, bpc);
    run<3, 4>("D 4 lanes x 16B", t, nb, out, mprobes, bpc);
    run<3, 8>("D 4 lanes x 16B", t, nb, out, mprobes, bpc);
  }
  return 0;
}
