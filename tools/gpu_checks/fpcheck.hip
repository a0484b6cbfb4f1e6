// Checks of the short f32 sequences the fused kernel uses in place of the compiler's full IEEE expansions
// (all must reproduce the IEEE result bit for bit on the stated ranges):
//  A. reciprocal: r0 = v_rcp_f32(b); r = fma(fma(-b, r0, 1), r0, r0)            == 1.0f / b     (exhaustive, 2^-60 <= |b| <= 2^60)
//  B. division  : div_rcp(a, b, r) with that r                                   == a / b        (random pairs)
//  C. sqrt(x*x) == x for x >= 0 with x*x neither overflowing nor subnormal                        (exhaustive)
//  D. sqrt      : s = v_sqrt_f32(x) corrected by one ulp either way from the fma residuals == sqrtf(x)  (exhaustive, 2^-60..2^60)
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off fpcheck.hip -o fpcheck
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__device__ __forceinline__ float rcp_refined(float b) {
    const float r0 = __builtin_amdgcn_rcpf(b);
    const float e = fmaf(-b, r0, 1.0f);
    return fmaf(e, r0, r0);
}
__device__ __forceinline__ float div_rcp(float a, float b, float rb) {
    float q0 = a * rb;
    float r0 = fmaf(-b, q0, a);
    float q1 = fmaf(r0, rb, q0);
    float r1 = fmaf(-b, q1, a);
    return fmaf(r1, rb, q1);
}
__device__ __forceinline__ float sqrt_lean(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sd = __uint_as_float(__float_as_uint(s) - 1u), su = __uint_as_float(__float_as_uint(s) + 1u);
    const float rd = fmaf(-sd, s, x), ru = fmaf(-su, s, x);
    float r = (rd <= 0.0f) ? sd : s;
    r = (ru > 0.0f) ? su : r;
    return r;
}
__device__ uint64_t rng(uint64_t &s) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s;
}

// exhaustive over the 2^23 mantissas x exponents [elo, ehi]
__global__ void sweep(int elo, int ehi, unsigned long long *bad) {
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;  // 2^23 threads
    unsigned long long bA = 0, bC = 0, bD = 0;
    for (int e = elo; e <= ehi; ++e) {
        const float x = __uint_as_float(((uint32_t)e << 23) | m);
        if (__float_as_uint(rcp_refined(x)) != __float_as_uint(1.0f / x)) bA++;
        if (__float_as_uint(rcp_refined(-x)) != __float_as_uint(1.0f / -x)) bA++;
        if (__float_as_uint(sqrtf(x * x)) != __float_as_uint(x)) bC++;
        if (__float_as_uint(sqrt_lean(x)) != __float_as_uint(sqrtf(x))) bD++;
    }
    if (bA) atomicAdd(bad + 0, bA);
    if (bC) atomicAdd(bad + 2, bC);
    if (bD) atomicAdd(bad + 3, bD);
}
__global__ void pairs(uint64_t seed, int mode, unsigned long long *bad, unsigned long long *n) {
    uint64_t s = seed + 0x9E3779B97F4A7C15ull * (blockIdx.x * blockDim.x + threadIdx.x + 1);
    unsigned long long b = 0, cnt = 0;
    for (int it = 0; it < 4096; ++it) {
        uint32_t ua = (uint32_t)rng(s), ub = (uint32_t)rng(s);
        if (mode == 0) {
            ua = (ua & 0x807fffffu) | ((87u + (ua >> 23) % 81u) << 23);   // 2^-40 .. 2^40
            ub = (ub & 0x807fffffu) | ((87u + (ub >> 23) % 81u) << 23);
        } else {  // divisor mantissa near all-ones / all-zeros
            ua = (ua & 0x807fffffu) | ((100u + (ua >> 23) % 56u) << 23);
            uint32_t mm = (ub & 0xff);
            mm = (ub & 0x100) ? (0x7fffffu - mm) : mm;
            ub = (ub & 0x80000000u) | ((100u + (ub >> 23) % 56u) << 23) | mm;
        }
        const float a = __uint_as_float(ua), bb = __uint_as_float(ub);
        if (__float_as_uint(div_rcp(a, bb, rcp_refined(bb))) != __float_as_uint(a / bb)) b++;
        cnt++;
    }
    atomicAdd(bad + 1, b);
    atomicAdd(n, cnt);
}

int main() {
    unsigned long long *d, h[5];
    hipMalloc(&d, 40);
    hipMemset(d, 0, 40);
    // exponents 67..187 = 2^-60 .. 2^60  (x*x stays normal and finite for C on 2^-60..2^60: 2^-120 .. 2^120)
    hipLaunchKernelGGL(sweep, dim3((1u << 23) / 256), dim3(256), 0, 0, 67, 187, d);
    // wider sweep of A and D only: exponents 27..227 = 2^-100 .. 2^100
    unsigned long long *d2;
    hipMalloc(&d2, 40);
    hipMemset(d2, 0, 40);
    hipLaunchKernelGGL(sweep, dim3((1u << 23) / 256), dim3(256), 0, 0, 27, 227, d2);
    unsigned long long h2[5];
    hipDeviceSynchronize();
    hipMemcpy(h2, d2, 40, hipMemcpyDeviceToHost);
    printf("wide sweep 2^-100..2^100: reciprocal mismatches %llu, lean sqrt mismatches %llu\n", h2[0], h2[3]);
    for (int mode = 0; mode < 2; ++mode)
        for (int rep = 0; rep < 16; ++rep) hipLaunchKernelGGL(pairs, dim3(4096), dim3(256), 0, 0, 4242ull + rep * 7919ull + mode, mode, d, d + 4);
    hipDeviceSynchronize();
    hipMemcpy(h, d, 40, hipMemcpyDeviceToHost);
    printf("A reciprocal (exhaustive 2^-60..2^60, both signs): mismatches %llu\n", h[0]);
    printf("B division from the refined reciprocal: %llu pairs, mismatches %llu\n", h[4], h[1]);
    printf("C sqrt(x*x) == x (exhaustive): mismatches %llu\n", h[2]);
    printf("D lean sqrt (exhaustive): mismatches %llu\n", h[3]);
    return 0;
}
