// gather_rate.hip -- the part's random-request rate, measured in the bench run itself (libslk_gather.so; bench.py loads it after the
// timed region).  Measurement infrastructure, not product: nothing in slacken_amd/ links or loads it.
// The table probe of the classify kernel is one random 64-byte request per super-mer (4 lanes x 16 B, 16 requests per wave
// instruction), so what bounds it is the rate at which the memory system serves such requests over a table-sized footprint --
// 48.4 G/s in round 2's experiments (tools/gather_bench2.hip), a number that varies by a few per cent from box to box.  This
// helper repeats the `size` experiment's one relevant point on the caller's buffer: uniformly random aligned requests of
// lanes_per_request x 16 bytes over the whole buffer.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

__device__ __forceinline__ uint64_t mix(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x;
}

template <int LPP>
__global__ void __launch_bounds__(256) gather(const uint4 *__restrict__ t, uint64_t units, int iters, uint64_t *out) {
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t grp = tid / LPP;
  const uint32_t c = (uint32_t)(tid % LPP);
  uint64_t acc = 0;
  for (int it = 0; it < iters; it += 8) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const uint64_t r = mix(grp * 1315423911ULL + (uint64_t)(it + u) * 0x9E3779B97F4A7C15ULL);
      const uint64_t unit = __umul64hi(r, units);   // uniform over [0, units): any buffer size, not a power of two
      const uint4 a = t[unit * LPP + c];
      acc += a.x ^ a.y ^ a.z ^ a.w;
    }
  }
  if (acc == 0x1234567) out[0] = acc;
}

template <int LPP>
int run(const void *buf, uint64_t bytes, double mreq, double *out_g, float *out_ms) {
  const uint64_t units = bytes / ((uint64_t)LPP * 16);
  if (units == 0) return -1;
  const int blocks = 256 * 8;
  const uint64_t threads = (uint64_t)blocks * 256;
  int iters = (int)(mreq * 1e6 * LPP / (double)threads);
  iters = (iters / 8 + 1) * 8;
  const double reqs = (double)threads / LPP * iters;
  uint64_t *d_out = nullptr;
  hipEvent_t a = nullptr, b = nullptr;
  if (hipMalloc((void **)&d_out, 64) != hipSuccess) return -2;
  int rc = 0;
  float ms = 0;
  if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) rc = -2;
  if (!rc) {
    hipLaunchKernelGGL((gather<LPP>), dim3(blocks), dim3(256), 0, 0, (const uint4 *)buf, units, 8, d_out);   // warm
    if (hipEventRecord(a, 0) != hipSuccess) rc = -3;
    hipLaunchKernelGGL((gather<LPP>), dim3(blocks), dim3(256), 0, 0, (const uint4 *)buf, units, iters, d_out);
    if (hipEventRecord(b, 0) != hipSuccess || hipEventSynchronize(b) != hipSuccess || hipGetLastError() != hipSuccess) rc = -3;
    if (!rc && hipEventElapsedTime(&ms, a, b) != hipSuccess) rc = -3;
  }
  if (a) (void)hipEventDestroy(a);
  if (b) (void)hipEventDestroy(b);
  (void)hipFree(d_out);
  if (rc) return rc;
  *out_g = reqs / ms / 1e6;
  if (out_ms) *out_ms = ms;
  return 0;
}

}  // namespace

// d_buf: `bytes` of device memory on the current device (contents irrelevant); lanes_per_request: 4 (64-byte requests, the probe's
// shape) or 8 (128-byte requests); mreq: about how many million requests to issue.  -> G requests/s.  0 = ok.
extern "C" int slk_gather_rate(const void *d_buf, uint64_t bytes, int lanes_per_request, double mreq, double *out_grequests_per_s,
                               float *out_ms) {
  if (!d_buf || !out_grequests_per_s) return -1;
  if (lanes_per_request == 4) return run<4>(d_buf, bytes, mreq, out_grequests_per_s, out_ms);
  if (lanes_per_request == 8) return run<8>(d_buf, bytes, mreq, out_grequests_per_s, out_ms);
  return -1;
}
