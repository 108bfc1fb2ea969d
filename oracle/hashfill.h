/* TEST INFRASTRUCTURE (oracle/): closed-form integer-hash fill shared by the
 * reference harness (ref_harness.cc), the C restatement (oracle.c) and, re-stated in
 * numpy, tests/hashfill.py.  Golden fixtures store only OUTPUTS; every input is
 * regenerated from (seed, index) with these functions on both sides.
 * Not part of the product path.
 */
#ifndef ALEPPO_ORACLE_HASHFILL_H
#define ALEPPO_ORACLE_HASHFILL_H
#include <stdint.h>

static inline uint32_t hf_u32(uint32_t seed, uint32_t idx) {
  uint32_t x = idx * 0x9E3779B1u + seed * 0x85EBCA77u + 0x165667B1u;
  x ^= x >> 16;
  x *= 0x7FEB352Du;
  x ^= x >> 15;
  x *= 0x846CA68Bu;
  x ^= x >> 16;
  return x;
}
/* uniform in [0,1) with 24 significant bits (exact in f32) */
static inline float hf_unit(uint32_t seed, uint32_t idx) {
  return (float)(hf_u32(seed, idx) >> 8) * (1.0f / 16777216.0f);
}
/* uniform in [lo,hi) */
static inline float hf_range(uint32_t seed, uint32_t idx, float lo, float hi) {
  return lo + (hi - lo) * hf_unit(seed, idx);
}
static inline uint8_t hf_byte(uint32_t seed, uint32_t idx) {
  return (uint8_t)(hf_u32(seed, idx) >> 24);
}
#endif
