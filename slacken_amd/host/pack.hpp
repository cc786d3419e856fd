// pack.hpp -- reads in the engine's 3-bit form, on the host: 16 bases per word, 2-bit codes (BitRepresentation.charToTwobit,
// S/kmers/util/BitRepresentation.scala:127-135: A = 0, C = 1, G = 2, T / U = 3, either case) and one validity bit per base
// (BitRepresentation.isValid :140-143); everything that is not a nucleotide is invalid, whatever it was.  This is what the lane
// kernel stages a tile as internally (lane.hip: pack16); shipped like this a read costs 3 bits per base on the PCIe link instead
// of 8 (slk_classify_batch_packed).  SIMD: 32 bases per step with AVX2 + BMI2 where the CPU has them (chosen at run time), a
// table-driven loop otherwise.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace slk {

// codes[i] bits 2j..2j+1 / valid[i] bit j describe base 16 i + j; words = ceil(n / 16); the unused high bits of the last word are 0
inline void pack_bases_scalar(const uint8_t *b, uint64_t n, uint32_t *codes, uint16_t *valid) {
  static const struct Lut {
    uint8_t v[256];
    Lut() {
      memset(v, 0x80, sizeof(v));   // bit 7: invalid
      const char *nt = "ACGT";
      for (int i = 0; i < 4; i++) { v[(uint8_t)nt[i]] = (uint8_t)i; v[(uint8_t)(nt[i] | 0x20)] = (uint8_t)i; }
      v[(uint8_t)'U'] = 3; v[(uint8_t)'u'] = 3;
    }
  } lut;
  const uint64_t words = (n + 15) / 16;
  for (uint64_t w = 0; w < words; w++) {
    uint32_t c = 0, v = 0;
    const uint64_t base = w * 16, m = n - base < 16 ? n - base : 16;
    for (uint64_t j = 0; j < m; j++) {
      const uint8_t x = lut.v[b[base + j]];
      if (!(x & 0x80)) { c |= (uint32_t)x << (2 * j); v |= 1u << j; }
    }
    codes[w] = c;
    valid[w] = (uint16_t)v;
  }
}

#if defined(__x86_64__)
__attribute__((target("avx2,bmi2"))) inline void pack_bases_avx2(const uint8_t *b, uint64_t n, uint32_t *codes, uint16_t *valid) {
  const __m256i fold = _mm256_set1_epi8((char)0xDF), three = _mm256_set1_epi8(3), one = _mm256_set1_epi8(1);
  const __m256i cA = _mm256_set1_epi8('A'), cC = _mm256_set1_epi8('C'), cG = _mm256_set1_epi8('G'), cT = _mm256_set1_epi8('T'), cU = _mm256_set1_epi8('U');
  const uint64_t full = n / 32;
  for (uint64_t i = 0; i < full; i++) {
    const __m256i x = _mm256_loadu_si256((const __m256i *)(b + 32 * i));
    const __m256i u = _mm256_and_si256(x, fold);                                   // upper case
    const __m256i ok = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(u, cA), _mm256_cmpeq_epi8(u, cC)),
                                       _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(u, cG), _mm256_cmpeq_epi8(u, cT)), _mm256_cmpeq_epi8(u, cU)));
    const uint32_t vm = (uint32_t)_mm256_movemask_epi8(ok);
    // (c >> 1) & 3: A, C, T / U, G -> 0, 1, 2, 3; ^= >> 1: -> A = 0, C = 1, G = 2, T = 3
    __m256i t = _mm256_and_si256(_mm256_srli_epi16(x, 1), three);
    t = _mm256_xor_si256(t, _mm256_and_si256(_mm256_srli_epi16(t, 1), one));
    t = _mm256_and_si256(t, ok);                                                   // (invalid bases: code 0, as the scalar route)
    alignas(32) uint64_t q[4];
    _mm256_store_si256((__m256i *)q, t);
    const uint64_t M = 0x0303030303030303ULL;
    codes[2 * i] = (uint32_t)(_pext_u64(q[0], M) | (_pext_u64(q[1], M) << 16));
    codes[2 * i + 1] = (uint32_t)(_pext_u64(q[2], M) | (_pext_u64(q[3], M) << 16));
    valid[2 * i] = (uint16_t)vm;
    valid[2 * i + 1] = (uint16_t)(vm >> 16);
  }
  if (n > full * 32) pack_bases_scalar(b + full * 32, n - full * 32, codes + 2 * full, valid + 2 * full);
}
#endif

inline void pack_bases(const uint8_t *b, uint64_t n, uint32_t *codes, uint16_t *valid) {
#if defined(__x86_64__)
  static const bool fast = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2");
  if (fast) { pack_bases_avx2(b, n, codes, valid); return; }
#endif
  pack_bases_scalar(b, n, codes, valid);
}

}  // namespace slk
