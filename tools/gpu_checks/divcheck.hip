// Does the 5-operation reciprocal sequence reproduce IEEE f32 division bit for bit?
//   q0 = a*rb ; r0 = fma(-b,q0,a) ; q1 = fma(r0,rb,q0) ; r1 = fma(-b,q1,a) ; q = fma(r1,rb,q1),  rb = 1.0f/b (IEEE)
// Exhaustive over all f32 `a` in a few binades x many `b`, plus random pairs over wide ranges.
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off divcheck.hip -o divcheck
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__device__ __forceinline__ float div_rcp(float a, float b, float rb) {
    float q0 = a * rb;
    float r0 = fmaf(-b, q0, a);
    float q1 = fmaf(r0, rb, q0);
    float r1 = fmaf(-b, q1, a);
    return fmaf(r1, rb, q1);
}
__device__ __forceinline__ float div_rcp1(float a, float b, float rb) {  // single correction
    float q0 = a * rb;
    float r0 = fmaf(-b, q0, a);
    return fmaf(r0, rb, q0);
}

__device__ uint64_t rng(uint64_t &s) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s;
}

__global__ void check(uint64_t seed, int mode, unsigned long long *bad2, unsigned long long *bad1, unsigned long long *n) {
    uint64_t s = seed + 0x9E3779B97F4A7C15ull * (blockIdx.x * blockDim.x + threadIdx.x + 1);
    unsigned long long b2 = 0, b1 = 0, cnt = 0;
    for (int it = 0; it < 4096; ++it) {
        uint32_t ua = (uint32_t)rng(s), ub = (uint32_t)rng(s);
        float a, b;
        if (mode == 0) {  // a: any mantissa, exponent within +-20 binades of 1; b: likewise
            ua = (ua & 0x807fffffu) | ((107u + (ua >> 23) % 41u) << 23);
            ub = (ub & 0x807fffffu) | ((107u + (ub >> 23) % 41u) << 23);
        } else if (mode == 1) {  // b with mantissa near all-ones / all-zeros (hard cases for reciprocal steps)
            ua = (ua & 0x807fffffu) | ((120u + (ua >> 23) % 16u) << 23);
            uint32_t m = (ub & 0xff);
            m = (ub & 0x100) ? (0x7fffffu - m) : m;
            ub = (ub & 0x80000000u) | ((120u + (ub >> 23) % 16u) << 23) | m;
        } else {  // the linearity use: a = 2*(S-Smin) with S integer 0..65535 (+fractions), b = span ~ 5e4..7e4
            float S = (float)(ua & 0xffff) + (float)((ua >> 16) & 7) * 0.125f;
            float smin = 4000.0f + (float)(ub & 0x7ff) * 0.5f;
            a = 2.0f * (S - smin);
            b = 50000.0f + (float)((ub >> 11) & 0xffff) * 0.25f + (float)(ub >> 27) * 0.001f;
            ua = __float_as_uint(a);
            ub = __float_as_uint(b);
        }
        a = __uint_as_float(ua);
        b = __uint_as_float(ub);
        float rb = 1.0f / b;
        float ref = a / b;
        float f2 = div_rcp(a, b, rb), f1 = div_rcp1(a, b, rb);
        if (__float_as_uint(ref) != __float_as_uint(f2)) b2++;
        if (__float_as_uint(ref) != __float_as_uint(f1)) b1++;
        cnt++;
    }
    atomicAdd(bad2, b2);
    atomicAdd(bad1, b1);
    atomicAdd(n, cnt);
}

int main() {
    unsigned long long *d, h[3];
    hipMalloc(&d, 24);
    for (int mode = 0; mode < 3; ++mode) {
        hipMemset(d, 0, 24);
        for (int rep = 0; rep < 8; ++rep) hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, 12345ull + rep * 7919ull + mode, mode, d, d + 1, d + 2);
        hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("mode %d: %llu pairs, mismatches two-step %llu, one-step %llu\n", mode, h[2], h[0], h[1]);
    }
    return 0;
}
