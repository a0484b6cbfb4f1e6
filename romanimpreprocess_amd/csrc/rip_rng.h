// Counter-based random deviates for the simulation-side kernels (synth.hip): Philox-4x32-10 keyed by a 64-bit seed, the
// counter words chosen by the caller (pixel, read / plane, attempt, domain tag), so every deviate is a pure function of its
// coordinates: reproducible, order-independent, no generator state in memory.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace riprng {

__device__ __forceinline__ void philox(uint32_t (&c)[4], uint64_t seed) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        c[1] = (uint32_t)p1;
        c[3] = (uint32_t)p0;
        c[0] = n0;
        c[2] = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

// uniform in (0, 1) from two words (53 bits)
__device__ __forceinline__ double u53(uint32_t hi, uint32_t lo) {
    return ((double)(((uint64_t)hi << 21) | (lo >> 11)) + 0.5) * (1.0 / 9007199254740992.0);
}

// one f32 standard normal for (a, b, tag): Box-Muller on two 24-bit uniforms
__device__ __forceinline__ float normal_f32(uint64_t seed, uint32_t a, uint32_t b, uint32_t tag) {
    uint32_t c[4] = {a, b, tag, 0x6c317379u};
    philox(c, seed);
    const float u1 = ((float)(c[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u2 = ((float)(c[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    return sqrtf(-2.0f * logf(u1)) * cospif(2.0f * u2);
}

// log(k!) for integer-valued k >= 0: exact table below 8, Stirling's series above (truncation < 2e-12 at k + 1 = 9): one logarithm
// where lgamma costs several
__device__ __forceinline__ double log_factorial(double k) {
    if (k < 8.0) {
        const int i = (int)k;
        const double t[8] = {0.0, 0.0, 0.6931471805599453, 1.791759469228055, 3.1780538303479458, 4.787491742782046, 6.579251212010101,
                             8.525161361065415};
        double r = 0.0;
#pragma unroll
        for (int j = 0; j < 8; ++j) r = (i == j) ? t[j] : r;
        return r;
    }
    const double x = k + 1.0, ix = 1.0 / x, ix2 = ix * ix;
    const double corr = ix * (1.0 / 12.0 - ix2 * (1.0 / 360.0 - ix2 * (1.0 / 1260.0 - ix2 * (1.0 / 1680.0))));
    return (k + 0.5) * log(x) - x + 0.9189385332046727 + corr;
}

// Poisson deviates of ONE mean lam >= 10 drawn many times (the apportioning: the reads of one share of a pixel) by W. Hoermann's
// transformed rejection (PTRS, 1993): its constants once (PtrsPlan), and the acceptance test outside the squeeze -- which some lane of a wave needs on
// almost every draw, so every wave pays for it -- first in f32 in a form without the large cancelling terms:
//     -lam + k log lam - log k!  =  d - k log1p(d / lam) - log(2 pi k) / 2 - 1/(12 k) + 1/(360 k^3) - ...,   d = k - lam
// (Stirling; |d| is a few sqrt(lam), so the f32 error is ~4e-7 |d|, not 1e-7 k log lam), decided when the two sides differ by more
// than eps = 2e-6 + 1e-6 |d| and re-done in f64 otherwise (a few draws in 10^5: the distribution is that of the f64 test).
struct PtrsPlan {
    double lam, loglam, bb, aa, inv_alpha, vr;
    float bb32, aa32, ia32, inv_lam32;
};

__device__ __forceinline__ PtrsPlan ptrs_plan(double lam, double slam, double loglam) {
    PtrsPlan p;
    p.lam = lam, p.loglam = loglam;
    p.bb = 0.931 + 2.53 * slam;
    p.aa = -0.059 + 0.02483 * p.bb;
    p.inv_alpha = 1.1239 + 1.1328 / (p.bb - 3.4);
    p.vr = 0.9277 - 3.6224 / (p.bb - 2.0);
    p.bb32 = (float)p.bb, p.aa32 = (float)p.aa, p.ia32 = (float)p.inv_alpha, p.inv_lam32 = (float)(1.0 / lam);
    return p;
}

__device__ inline double poisson_ptrs(const PtrsPlan &p, uint64_t seed, uint32_t a, uint32_t b, uint32_t tag) {
    // one candidate (u, v): 1 accepted (k), 0 rejected
    auto candidate = [&](double u, double v, double &k) -> bool {
        const double us = 0.5 - fabs(u);
        double r = __builtin_amdgcn_rcp(us);   // 1 / us: hardware reciprocal + one Newton step (us >= 2^-34)
        r = __builtin_fma(__builtin_fma(-us, r, 1.0), r, r);
        k = floor((2.0 * p.aa * r + p.bb) * u + p.lam + 0.43);
        if (us >= 0.07 && v <= p.vr) return true;
        if (k < 0.0 || (us < 0.013 && v > us)) return false;
        if (k >= 16.0 && k < 8.0e6) {
            const float kf = (float)k, d = (float)(k - p.lam), r32 = (float)r;
            const float lhs = logf((float)v * p.ia32 * __builtin_amdgcn_rcpf(p.aa32 * (r32 * r32) + p.bb32));
            const float ik = __builtin_amdgcn_rcpf(kf);
            const float rhs = d - kf * log1pf(d * p.inv_lam32) - 0.5f * logf(6.2831853f * kf) - ik * (1.0f / 12.0f - ik * ik * (1.0f / 360.0f));
            const float eps = 2.0e-6f + 1.0e-6f * fabsf(d);
            if (rhs - lhs > eps) return true;
            if (rhs - lhs < -eps) return false;
        } else if (k < 16.0) {   // small counts (means of 10 .. 20): log k! from a table, terms below 50: f32 error below 1e-5
            const float lf[16] = {0.0f,       0.0f,       0.6931472f, 1.7917595f, 3.1780539f, 4.7874917f, 6.5792513f, 8.5251614f,
                                  10.604603f, 12.801827f, 15.104413f, 17.502308f, 19.987214f, 22.552164f, 25.191221f, 27.899271f};
            const int ki = (int)k;
            float lfk = 0.0f;
#pragma unroll
            for (int j = 0; j < 16; ++j) lfk = (ki == j) ? lf[j] : lfk;
            const float r32 = (float)r;
            const float lhs = logf((float)v * p.ia32 * __builtin_amdgcn_rcpf(p.aa32 * (r32 * r32) + p.bb32));
            const float rhs = (float)k * (float)p.loglam - (float)p.lam - lfk;
            if (rhs - lhs > 3.0e-5f) return true;
            if (rhs - lhs < -3.0e-5f) return false;
        }
        return log(v * p.inv_alpha / (p.aa * (r * r) + p.bb)) <= -p.lam + k * p.loglam - log_factorial(k);
    };
    // TWO candidates per Philox block (32-bit u and v each): a wave repeats the loop until its last lane has accepted -- with
    // one candidate per block three times on average (acceptance 0.86-0.9 per candidate, 64 lanes), with two 1.6 times
    for (uint32_t attempt = 0; attempt < 32; ++attempt) {
        uint32_t c[4] = {a, b, tag ^ (attempt << 24), 0x70747232u};
        philox(c, seed);
        double k;
        if (candidate(((double)c[0] + 0.5) * (1.0 / 4294967296.0) - 0.5, ((double)c[1] + 0.5) * (1.0 / 4294967296.0), k)) return k;
        if (candidate(((double)c[2] + 0.5) * (1.0 / 4294967296.0) - 0.5, ((double)c[3] + 0.5) * (1.0 / 4294967296.0), k)) return k;
    }
    return floor(p.lam + 0.5);   // not reached in practice
}

// Binomial(n, p) deviate: inversion (sequential search) where n min(p, 1-p) < 10, else W. Hoermann's transformed rejection
// with squeeze (BTRS, "The generation of binomial random variates", 1993); the acceptance test compares with the exact ratio
// of probabilities through lgamma.
__device__ inline int binomial(int n, double p, uint64_t seed, uint32_t a, uint32_t b, uint32_t tag) {
    if (n <= 0 || !(p > 0.0)) return 0;
    if (p >= 1.0) return n;
    const bool flip = p > 0.5;
    const double q = flip ? 1.0 - p : p;
    int k;
    if ((double)n * q < 10.0) {
        uint32_t c[4] = {a, b, tag, 0x62696e31u};
        philox(c, seed);
        const double u = u53(c[0], c[1]);
        const double odds = q / (1.0 - q);
        double pm = exp((double)n * log1p(-q)), cdf = pm;
        k = 0;
        while (u > cdf && k < n && k < 400) {
            ++k;
            pm *= odds * (double)(n - k + 1) / (double)k;
            cdf += pm;
        }
    } else {
        const double nd = (double)n, sd = sqrt(nd * q * (1.0 - q));
        const double bb = 1.15 + 2.53 * sd, aa = -0.0873 + 0.0248 * bb + 0.01 * q, cc = nd * q + 0.5, vr = 0.92 - 4.2 / bb;
        const double alpha = (2.83 + 5.1 / bb) * sd, lr = log(q / (1.0 - q));
        const double m = floor((nd + 1.0) * q);
        double lpm = 0.0;        // log pmf(m) up to the common terms: only the ~14 % of attempts outside the squeeze need it
        bool have_lpm = false;
        k = (int)m;
        for (uint32_t attempt = 0; attempt < 64; ++attempt) {
            uint32_t c[4] = {a, b, tag ^ (attempt << 24), 0x62747273u};
            philox(c, seed);
            const double u = u53(c[0], c[1]) - 0.5;
            double v = u53(c[2], c[3]);
            const double us = 0.5 - fabs(u);
            const double kd = floor((2.0 * aa / us + bb) * u + cc);
            if (kd < 0.0 || kd > nd) continue;
            if (us >= 0.07 && v <= vr) {
                k = (int)kd;
                break;
            }
            v = log(v * alpha / (aa / (us * us) + bb));
            if (!have_lpm) {
                lpm = -lgamma(m + 1.0) - lgamma(nd - m + 1.0) + m * lr;
                have_lpm = true;
            }
            if (v <= -lgamma(kd + 1.0) - lgamma(nd - kd + 1.0) + kd * lr - lpm) {
                k = (int)kd;
                break;
            }
        }
    }
    return flip ? n - k : k;
}

}   // namespace riprng
