// Per-pixel inverse linearity (ipc_linearity.invlinearity, ipc_linearity.py:347-394), shared by the stage kernel (invlin.hip)
// and the Level-1 synthesis (synth.hip).  24 bisection steps on z in (-1, 1); each evaluates the Legendre series of
// ipc_linearity._lin (:192-231) WITHOUT the linear extrapolation branch, in numpy's operation order and dtypes:
//     phi (f32) += coefs[L] (f32) * poly (ZT)          one rounding per operation; with ZT = f64 the sum is rounded back to f32
//     poly_next = c1_L * z * poly - c2_L * poly_prev    c1, c2 Python floats: cast to f32 when z is f32, exact f64 otherwise
//     z += phi < Slin ? 2^-j : -2^-j
// then S = Smin + (Smax - Smin) / 2 * (1 + z).  `ex` = |z| > 1 at the last evaluation (the reference's second return value).
#pragma once
#include <hip/hip_runtime.h>

template <typename ZT, int NP>
__device__ __forceinline__ ZT rip_invlin_pixel(ZT target, const float (&c)[NP], float smin, float smax, bool &ex) {
    ZT c1[NP], c2[NP];   // (2L+1)/(L+1) and L/(L+1): f64 division, then the cast numpy applies to a Python float operand
#pragma unroll
    for (int L = 1; L < NP; ++L) {
        c1[L] = (ZT)((double)(2 * L + 1) / (double)(L + 1));
        c2[L] = (ZT)((double)L / (double)(L + 1));
    }
    ZT z = (ZT)0;
    ZT step = (ZT)1;
    ex = false;
    for (int j = 1; j <= 24; ++j) {
        step = step * (ZT)0.5;
        ex = (z < (ZT)0 ? -z : z) > (ZT)1;
        float phi = c[0];
        ZT pp = (ZT)1, p = z;
#pragma unroll
        for (int L = 1; L < NP; ++L) {
            phi = (float)((ZT)phi + (ZT)c[L] * p);
            const ZT pn = (c1[L] * z) * p - c2[L] * pp;
            pp = p;
            p = pn;
        }
        z = z + (((ZT)phi < target) ? step : -step);
    }
    const float half = (smax - smin) / 2.0f;
    return (ZT)smin + (ZT)half * ((ZT)1 + z);
}

// The same bisection for a SEQUENCE of targets of one pixel (the reads of an exposure, synth.hip): the series values along the
// previous target's path are kept (phi_path[j], decisions in `path`), and while a target takes the same decisions as its
// predecessor it stands at the same z, so its series value IS the kept one -- no evaluation.  From the first differing
// decision on, z differs and every step is evaluated (and kept) as in rip_invlin_pixel.  Identical comparisons on identical
// values: bit-identical results, fewer f64 operations (consecutive reads differ by a few electrons out of ~1e5, so the first
// steps of their paths coincide).  `have` = a path has been kept; the wave evaluates a step when any of its lanes must.
// `cw`: the coefficients widened to ZT once per pixel by the caller (the conversion is exact and invariant over the 35 reads x
// 24 steps; the compiler does not hoist it out of the read loop by itself: 8 of the ~70 f64-rate instructions of an evaluation)
template <typename ZT, int NP>
__device__ __forceinline__ ZT rip_invlin_pixel_warm(ZT target, const float (&c)[NP], const ZT (&cw)[NP], float smin, float smax, bool &ex,
                                                    float (&phi_path)[24], uint32_t &path, bool &have) {
    ZT c1[NP], c2[NP];
#pragma unroll
    for (int L = 1; L < NP; ++L) {
        c1[L] = (ZT)((double)(2 * L + 1) / (double)(L + 1));
        c2[L] = (ZT)((double)L / (double)(L + 1));
    }
    ZT z = (ZT)0;
    ZT step = (ZT)1;
    ex = false;
    bool same = have;
    uint32_t newpath = 0;
#pragma unroll
    for (int j = 1; j <= 24; ++j) {
        step = step * (ZT)0.5;
        ex = (z < (ZT)0 ? -z : z) > (ZT)1;
        float phi = phi_path[j - 1];
        if (!same) {
            phi = c[0];
            ZT pp = (ZT)1, p = z;
#pragma unroll
            for (int L = 1; L < NP; ++L) {
                phi = (float)((ZT)phi + cw[L] * p);
                const ZT pn = (c1[L] * z) * p - c2[L] * pp;
                pp = p;
                p = pn;
            }
            phi_path[j - 1] = phi;
        }
        const bool up = (ZT)phi < target;
        same = same && (up == (((path >> (j - 1)) & 1u) != 0u));
        newpath |= (up ? 1u : 0u) << (j - 1);
        z = z + (up ? step : -step);
    }
    path = newpath;
    have = true;
    return (ZT)smin + (ZT)((smax - smin) / 2.0f) * ((ZT)1 + z);
}
