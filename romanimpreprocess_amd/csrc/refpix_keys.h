// Order-preserving integer keys of f32 values and the digit layout of the exact radix selections (refpix.hip, refpix_one.hip).
#pragma once
#include <stdint.h>

__device__ __forceinline__ uint32_t f2key(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

// three levels: 11 + 11 + 10 key bits, most significant first
#define SEL_BINS 2048
__device__ __forceinline__ int sel_shift(int level) { return level == 0 ? 21 : (level == 1 ? 10 : 0); }
__device__ __forceinline__ int sel_bits(int level) { return level == 2 ? 10 : 11; }
