// Device functions of the ramp fit (shared by rampfit.hip and chain.hip).
//   utils/fitting.py:89-255   jump_detect -> fit_variant()      (one pass over groups [0,g))
//   utils/fitting.py:233-241  var_delta_slope in the reference's exact order -> exact_variance()
// Arithmetic recipe: oracle/rampfit.py.
#pragma once
#include "rip_common.h"

// Plan tables (header, variants, weights, difference tables) never change while a kernel runs.  Reading them
// through the constant address space lets the compiler use scalar loads (s_load, lgkmcnt) for these wave-uniform
// values even after the kernel has stored to global memory; as plain global pointers they become vector loads
// followed by s_waitcnt vmcnt(0), which also waits for every prefetched pixel load in flight.
#define RIP_K __attribute__((address_space(4)))
template <typename T>
__device__ __forceinline__ const RIP_K T *rip_k(const T *p) {
    return (const RIP_K T *)p;
}
#define KLD(x) (*rip_k(&(x)))
__device__ __forceinline__ RipVariant rip_load_variant(const RipVariant *vars, int i) {
    RipVariant v;
    v.g = KLD(vars[i].g);
    v.ndiff = KLD(vars[i].ndiff);
    v.diff_ofs = KLD(vars[i].diff_ofs);
    v.k_ofs = KLD(vars[i].k_ofs);
    v.coef = KLD(vars[i].coef);
    v.rfac = KLD(vars[i].rfac);
    return v;
}

__device__ __forceinline__ float log_f32(float x) {
    // np.log on an f32 array.  Evaluated in f64 and rounded: correctly rounded f32 logarithm.
    // (numpy's AVX512F/AVX2 f32 log differs from the correctly rounded value by <= 2 ulp on ~4% of
    // inputs; the result only feeds the jump threshold -- see DESIGN.md "log of the threshold".)
    return (float)log((double)x);
}

__device__ __forceinline__ float hypot_f32(float a, float b) {
    // np.hypot on f32 = libm hypotf = (float)sqrt((double)a*a + (double)b*b)  (glibc flt-32/e_hypotf.c)
    if (isinf(a) || isinf(b)) return INFINITY;
    double da = (double)a, db = (double)b;
    return (float)sqrt(da * da + db * db);
}

// Short forms of IEEE f32 reciprocal and square root, bit-identical to 1.0f/b and sqrtf(x) for 2^-100 <= |.| <= 2^100
// (tools/gpu_checks/fpcheck.hip: exhaustive over all such floats, 0 mismatches).  Callers vote on the range with
// rip_mid_range() over the wave and fall back to the compiler's full expansion otherwise.
__device__ __forceinline__ bool rip_mid_range(float x) {  // false for NaN, Inf, 0, subnormals, |x| outside 2^-59..2^59
    const float ax = fabsf(x);
    return ax > 1.8e-18f && ax < 5.7e17f;
}
__device__ __forceinline__ bool rip_mid36(float x) {  // 2^-36 < |x| < 2^36: sums / products / quotients of two such stay in range
    const float ax = fabsf(x);
    return ax > 1.5e-11f && ax < 6.8e10f;
}
__device__ __forceinline__ float rip_rcp_mid(float b) {
    const float r0 = __builtin_amdgcn_rcpf(b);
    const float e = fmaf(-b, r0, 1.0f);
    return fmaf(e, r0, r0);
}
__device__ __forceinline__ float div_rcp_(float a, float b, float rb) {  // a / b from the exact reciprocal (chain_common.h)
    const float q0 = a * rb;
    const float r0 = fmaf(-b, q0, a);
    const float q1 = fmaf(r0, rb, q0);
    const float r1 = fmaf(-b, q1, a);
    return fmaf(r1, rb, q1);
}
__device__ __forceinline__ float rip_sqrt_mid(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sd = __uint_as_float(__float_as_uint(s) - 1u), su = __uint_as_float(__float_as_uint(s) + 1u);
    const float rd = fmaf(-sd, s, x), ru = fmaf(-su, s, x);
    float r = (rd <= 0.0f) ? sd : s;
    r = (ru > 0.0f) ? su : r;
    return r;
}

template <typename T>
__device__ __forceinline__ T clip_lo(T x, T lo) {  // np.clip(x, lo, None): NaN stays NaN
    return x < lo ? lo : x;
}
template <typename T>
__device__ __forceinline__ T clip2(T x, T lo, T hi) {  // np.clip(x, lo, hi)
    return x < lo ? lo : (x > hi ? hi : x);
}

template <typename GT>
struct GainConst;
template <>
struct GainConst<float> {
    static __device__ __forceinline__ float lo() { return 1e-4f; }
    static __device__ __forceinline__ float hi() { return 1e4f; }
};
template <>
struct GainConst<double> {
    static __device__ __forceinline__ double lo() { return 1e-4; }
    static __device__ __forceinline__ double hi() { return 1e4; }
};

// var_delta_slope exactly as fitting.py:233-241 (order of accumulation and dtype of every term)
template <typename GT>
__device__ __noinline__ double exact_variance(const RipPlanHeader *__restrict__ h, const float *__restrict__ kv, int g,
                                              int di, int dj, float dt, GT dv, float s2) {
    const float inv = 1.0f / dt;
    double var = 0.0;
    for (int a = 0; a < g; ++a) {
        double wa = ((a == dj) ? (double)inv : (a == di) ? (double)(-inv) : 0.0) - (double)kv[a];
        double inner;
        if constexpr (sizeof(GT) == 4) {
            float t1 = dv * h->tau[a];
            float t2 = s2 / h->nreads[a];
            inner = (double)(t1 + t2);
        } else {
            inner = dv * (double)h->tau[a] + (double)(s2 / h->nreads[a]);
        }
        var += (wa * wa) * inner;
        double twa = 2.0 * wa;
        for (int b = 0; b < a; ++b) {
            double wb = ((b == dj) ? (double)inv : (b == di) ? (double)(-inv) : 0.0) - (double)kv[b];
            var += ((twa * wb) * (double)dv) * (double)h->tbar[b];
        }
    }
    return var;
}

// one jump_detect pass on the ramp D[t*STRIDE] (LDS column of this thread); ORs JUMP_DET into J[i*STRIDE] when `flag`
template <typename GT, int STRIDE>
__device__ __forceinline__ void fit_variant(const float *D, uint8_t *J, const RipPlanHeader *__restrict__ h,
                                            const RipVariant v, const float *__restrict__ kv,
                                            const RipDiff *__restrict__ df, GT gain, float rn, bool flag,
                                            double guard, float &s_out, float &er_out, float &ep_out) {
    const int g = v.g;
    const float d1 = D[STRIDE];
    float s = 0.0f;
    for (int t = 0; t < g; ++t) {
        float diff = D[t * STRIDE] - d1;
        float prod = kv[t] * diff;
        s = s + prod;
    }
    GT gc = clip2<GT>(gain, GainConst<GT>::lo(), GainConst<GT>::hi());
    GT dv = clip_lo<GT>((GT)s / gc, (GT)0);
    GT pv = clip_lo<GT>((GT)v.coef * dv, (GT)0);
    float ep;
    if constexpr (sizeof(GT) == 4)
        ep = sqrtf(pv);
    else
        ep = (float)sqrt(pv);
    s_out = s;
    er_out = rn * v.rfac;
    ep_out = ep;
    if (!flag) return;

    float xc = clip2<float>(s, h->ia, h->ib);
    float lx = log_f32(xc / h->ia);
    double sth = h->sa + h->dsb * ((double)lx / h->loglen);
    float sth32 = (float)sth;
    float band = (float)(guard * fabs(sth)) + 0.0f;
    const float s2 = rn * rn;
    const float dv32 = (float)dv;
    for (int k = 0; k < v.ndiff; ++k) {
        const RipDiff r = df[k];
        float num = D[r.j * STRIDE] - D[r.i * STRIDE];
        float delta = num / r.dt - s;
        float var32 = r.A * s2 + r.B * dv32;
        float sm = delta / sqrtf(var32);
        bool hit;
        if (fabsf(sm - sth32) > band) {
            hit = sm > sth32;
        } else {  // within the guard band of the threshold (or NaN): redo in the reference's exact order
            double var = exact_variance<GT>(h, kv, g, r.i, r.j, r.dt, dv, s2);
            float sme = delta / (float)sqrt(var);
            hit = (double)sme > sth;
        }
        if (hit) J[r.i * STRIDE] |= (uint8_t)DQ_JUMP_DET;
    }
}


// ---------------------------------------------------------------------------------------------
// fitting.py:258-355 (ramp_fit) for one pixel whose ramp D[g*STRIDE], input group flags Q[g*STRIDE] and
// (zeroed) new-flag accumulators J[g*STRIDE] sit in LDS.  Returns slope / errors of the selected fit
// variant and the pixel DQ after flag propagation; writes the updated group flags to gdq_out (global,
// element stride gstride) when it is not null.
template <typename GT, int STRIDE>
__device__ __forceinline__ void rampfit_pixel(const float *D, const uint8_t *Q, uint8_t *J, int G,
                                              const RipPlanHeader *__restrict__ h, const RipVariant *__restrict__ vars,
                                              const float *__restrict__ kvals, const RipDiff *__restrict__ diffs,
                                              GT gain, float rn, bool active, double guard, uint32_t pdq_in,
                                              uint8_t *gdq_out, size_t gstride, float &s, float &er, float &ep,
                                              uint32_t &pdq_out) {
    const int start = h->start;
    // full ramp (fitting.py:313-320): jump flags kept only where the last group is not saturated
    const bool unsat = (Q[(G - 1) * STRIDE] & DQ_SATURATED) == 0;
    fit_variant<GT, STRIDE>(D, J, h, vars[0], kvals + vars[0].k_ofs, diffs + vars[0].diff_ofs, gain, rn,
                            unsat && active, guard, s, er, ep);
    // ramps truncated at the group where the pixel first saturates (fitting.py:326-337)
    for (int vi = 1; vi < h->nvariants; ++vi) {
        const RipVariant v = vars[vi];
        const int iend = v.g;
        const bool layer = ((Q[iend * STRIDE] & ~Q[(iend - 1) * STRIDE]) & DQ_SATURATED) != 0;
        if (layer)
            fit_variant<GT, STRIDE>(D, J, h, v, kvals + v.k_ofs, diffs + v.diff_ofs, gain, rn, active, guard, s, er, ep);
    }
    // flag propagation (fitting.py:339-353)
    uint32_t or_unsat = 0, any_sat = 0;
    bool all_dnu = true;
    for (int g = 0; g < G; ++g) {
        const uint32_t r = (uint32_t)Q[g * STRIDE] | (uint32_t)J[g * STRIDE];
        if (gdq_out) gdq_out[(size_t)g * gstride] = (uint8_t)r;
        if ((r & DQ_SATURATED) == 0) or_unsat |= r;
        any_sat |= r & DQ_SATURATED;
        all_dnu = all_dnu && ((r & DQ_DO_NOT_USE) != 0);
    }
    uint32_t pdq2 = or_unsat & ~DQ_DO_NOT_USE;
    if (all_dnu) pdq2 |= DQ_DO_NOT_USE;
    if (Q[(1 + start) * STRIDE] & DQ_SATURATED) pdq2 |= DQ_DO_NOT_USE;
    pdq2 |= any_sat;
    pdq_out = (pdq_in & DQ_REFERENCE_PIXEL) ? pdq_in : (pdq_in | pdq2);
}

// gen_cal_image.py:458-475 (err = hypot, var_poisson, trim + zero border), :213-229 (dark rate on the active
// region), :607-613 (error split), :616-629 (flat flags + division).  Pointers may be null (step skipped).
__device__ __forceinline__ void finish_pixel(bool active, size_t p, const float *__restrict__ dark_rate,
                                             const uint32_t *__restrict__ dark_dq, const float *__restrict__ flat,
                                             const uint32_t *__restrict__ flat_flags, float &s, float &er, float &ep,
                                             uint32_t &pdq) {
    float err = hypot_f32(er, ep);
    float vp = ep * ep;
    if (!active) {
        s = 0.0f;
        err = 0.0f;
        vp = 0.0f;
    }
    if (active && dark_rate) s = s - dark_rate[p];
    if (active && dark_dq) pdq |= dark_dq[p];
    float ep2 = sqrtf(vp);
    float e2 = err * err;
    float p2 = ep2 * ep2;
    float er2 = sqrtf(clip_lo<float>(e2 - p2, 0.0f));
    if (flat) {
        if (flat_flags) pdq |= flat_flags[p];
        const float f = flat[p];
        s = s / f;
        er2 = er2 / f;
        ep2 = ep2 / f;
    }
    er = er2;
    ep = ep2;
}

// ---------------------------------------------------------------------------------------------
// Register-resident fast path of the FULL-ramp fit (variant 0) for a compile-time group count.
// Same results as fit_variant(): slope / errors are computed with the reference's exact operations;
// each jump significance is first evaluated approximately (reciprocal multiply, rsq) and compared
// with the threshold through a rigorous error band; only inside the band (or for NaN) is it
// re-evaluated in the reference's exact operation order, so the flags are identical.
//   d[G]    the pixel's ramp
//   jmask   bit i set <=> JUMP_DET on group i (OR-ed in)
template <int G>
__device__ __forceinline__ void fit_full_regs(const float (&d)[G], const RipPlanHeader *__restrict__ h,
                                              const RipVariant v, const float *__restrict__ kv,
                                              const RipDiff *__restrict__ df, float gain, float rn, bool flag,
                                              double guard, float &s_out, float &er_out, float &ep_out,
                                              uint32_t &jmask) {
    const float d1 = d[1];
    float s = 0.0f;
#pragma unroll
    for (int t = 0; t < G; ++t) {
        const float diff = d[t] - d1;
        const float prod = KLD(kv[t]) * diff;
        s = s + prod;
    }
    const float gc = clip2<float>(gain, 1e-4f, 1e4f);
    const float dv = clip_lo<float>(s / gc, 0.0f);
    const float pv = clip_lo<float>(v.coef * dv, 0.0f);
    s_out = s;
    er_out = rn * v.rfac;
    ep_out = sqrtf(pv);
    if (!__any(flag)) return;

    const float xc = clip2<float>(s, KLD(h->ia), KLD(h->ib));
    const bool need_log = xc != KLD(h->ia);  // log(1) = 0 exactly otherwise
    float lx = 0.0f;
    if (__any(need_log)) lx = need_log ? __logf(xc / KLD(h->ia)) : 0.0f;
    const float slope_th = (float)(KLD(h->dsb) / KLD(h->loglen));
    const float sth32 = (float)KLD(h->sa) + slope_th * lx;
    // error band of the approximate comparison (see DESIGN.md "jump significance fast path")
    const float band0 = 2e-6f * fabsf(sth32) + 4e-6f * fabsf(slope_th) + 1e-30f;
    const float s2 = rn * rn;
    const bool force_exact = !(guard < 1e300);
    const int start = KLD(h->start);
    int k = 0;
#pragma unroll
    for (int i = 0; i < G - 1; ++i) {
        if (i < start) continue;
#pragma unroll
        for (int di = 1; di <= 2; ++di) {
            const int j = i + di;
            if (j > G - 1) continue;
            if (di == 2 && (i == G - 2 || G - 1 - start == 2)) continue;
            RipDiff r;
            r.i = KLD(df[k].i), r.j = KLD(df[k].j), r.dt = KLD(df[k].dt), r.A = KLD(df[k].A), r.B = KLD(df[k].B);
            r.inv_dt = KLD(df[k].inv_dt), r.relerr = KLD(df[k].relerr);
            ++k;
            const float num = d[j] - d[i];
            const float q = num * r.inv_dt;
            const float delta = q - s;
            const float var32 = fmaf(r.B, dv, r.A * s2);
            const float rs = __frsqrt_rn(var32);
            const float sm = delta * rs;
            const float band = band0 + r.relerr * fabsf(sm) + 4e-7f * (fabsf(q) + fabsf(s)) * rs;
            bool hit = sm > sth32;
            const bool unsure = force_exact || !(fabsf(sm - sth32) > band);
            if (__any(unsure && flag)) {
                if (unsure) {
                    const float lxe = log_f32(xc / KLD(h->ia));
                    const double sth = KLD(h->sa) + KLD(h->dsb) * ((double)lxe / KLD(h->loglen));
                    const float de = num / r.dt - s;
                    const double var = exact_variance<float>(h, kv, G, r.i, r.j, r.dt, dv, s2);
                    const float sme = de / (float)sqrt(var);
                    hit = (double)sme > sth;
                }
            }
            if (hit && flag) jmask |= 1u << i;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Packed form of fit_full_regs: the ramp arrives as float2 pairs dA[p] = (d[2p], d[2p+1]) (as the IPC stage
// produces them); the jump differences (i, i+di), (i+1, i+1+di) of one pair slot are evaluated together
// with packed f32 arithmetic, from the dense per-plan table RipDense (compile-time offsets: every uniform
// table load can be issued up front).  Same outputs as fit_variant() on the full ramp.
typedef float rf2 __attribute__((ext_vector_type(2)));

// plan constants of the jump threshold, computed once per kernel (wave-uniform)
struct RipFitConst {
    float ia, ib;       // f32(IthreshA), f32(IthreshB)
    float sa32;         // f32(SthreshA)
    float slope_th;     // f32((SthreshB - SthreshA) / log(IthreshB / IthreshA))
    float inv_ia;       // f32(1 / IthreshA): argument of the APPROXIMATE log only (its 1e-7 relative error is far inside band0)
};
__device__ __forceinline__ RipFitConst rip_fit_const(const RipPlanHeader *h) {
    RipFitConst c;
    c.ia = KLD(h->ia);
    c.ib = KLD(h->ib);
    c.sa32 = (float)KLD(h->sa);
    c.slope_th = (float)(KLD(h->dsb) / KLD(h->loglen));
    c.inv_ia = 1.0f / c.ia;
    return c;
}

__device__ __forceinline__ void rip_load_pair(RipDensePair &r, const RipDense *dn, int ps) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        r.inv_dt[e] = KLD(dn->pairs[ps].inv_dt[e]);
        r.A[e] = KLD(dn->pairs[ps].A[e]);
        r.B[e] = KLD(dn->pairs[ps].B[e]);
        r.k1[e] = KLD(dn->pairs[ps].k1[e]);
    }
}

// the differences the full-ramp fit tests (fitting.py:225-229) as the bit mask RipDense::valid, at compile time
template <int G, int START>
constexpr uint32_t rip_full_valid() {
    uint32_t m = 0;
    for (int i = START; i < G - 1; ++i) {
        const int dimax = (i == G - 2 || G - 1 - START == 2) ? 1 : 2;
        for (int di = 1; di <= dimax; ++di) m |= 1u << (2 * (2 * (i / 2) + (di - 1)) + (i & 1));
    }
    return m;
}

// state of the packed full-ramp fit between its two halves (registers; the fused kernel puts a barrier between them)
struct RipFitState {
    float s, er, ep;           // slope, read-noise error, Poisson error of the full ramp
    float dv, s2, xc;          // operands of the exact pass
    uint32_t jfast, unsure;    // per difference (bit 2*ps+e): approximate decision, needs the exact pass
    bool live;                 // wave-uniform: some lane of the wave keeps jump flags
};

// where the first half reads the dense per-plan table from: the device copy through scalar loads
struct RipDenseK {
    const RipDense *dn;
    __device__ __forceinline__ float k2(int t) const { return KLD(dn->K2[t]); }
    __device__ __forceinline__ uint32_t valid() const { return KLD(dn->valid); }
    __device__ __forceinline__ float amin() const { return KLD(dn->amin); }
    __device__ __forceinline__ void pair(RipDensePair &r, int ps) const { rip_load_pair(r, dn, ps); }
};

// first half: slope, errors, threshold, approximate significance of every tested difference
// VALID != 0: the tested differences are known at compile time (no plan-uniform branches: the eight difference
// slots become one basic block the scheduler can interleave)
template <int G, uint32_t VALID, typename TAB>
__device__ __forceinline__ void fit_full_pk_a_t(const rf2 (&dA)[G / 2], const RipFitConst fc, const RipVariant v,
                                                const TAB tabsrc, float gain, float rn, bool flag, double guard,
                                                RipFitState &st) {
    constexpr int GP = G / 2;
    constexpr int NS = 2 * GP;  // pair slots
    // scalar loads of the weights and of the first difference slot are issued before the slope arithmetic; slot
    // ps+1 is requested while slot ps is evaluated (scalar loads return out of order: every wait is lgkmcnt(0))
    float k2[G];
#pragma unroll
    for (int t = 0; t < G; ++t) k2[t] = tabsrc.k2(t);
    const uint32_t valid = VALID ? VALID : tabsrc.valid();
    RipDensePair tab[2];
    tabsrc.pair(tab[0], 0);
    const float d1 = dA[0].y;
    const rf2 d11 = {d1, d1};
    float s = 0.0f;
#pragma unroll
    for (int p = 0; p < GP; ++p) {
        const rf2 diff = dA[p] - d11;
        const rf2 kk = {k2[2 * p], k2[2 * p + 1]};
        const rf2 prod = kk * diff;
        s = s + prod.x;
        s = s + prod.y;
    }
    const float gc = clip2<float>(gain, 1e-4f, 1e4f);
    // s / gc and sqrt(pv): short exact forms when every lane is in their validated range (see rip_rcp_mid)
    float dvq;
    if (__all(rip_mid_range(s) && rip_mid_range(gc)))
        dvq = div_rcp_(s, gc, rip_rcp_mid(gc));
    else
        dvq = s / gc;
    const float dv = clip_lo<float>(dvq, 0.0f);
    const float pv = clip_lo<float>(v.coef * dv, 0.0f);
    st.s = s;
    st.er = rn * v.rfac;
    if (__all(pv == 0.0f || rip_mid_range(pv)))  // rip_sqrt_mid(0) = 0
        st.ep = rip_sqrt_mid(pv);
    else
        st.ep = sqrtf(pv);
    st.dv = dv;
    st.s2 = rn * rn;
    st.jfast = 0;
    st.unsure = 0;
    st.xc = 0.0f;
    st.live = __any(flag);
    if (!st.live) return;

    const float xc = clip2<float>(s, fc.ia, fc.ib);
    st.xc = xc;
    const bool need_log = xc != fc.ia;  // log(1) = 0 exactly otherwise
    float lx = 0.0f;
    if (__any(need_log)) lx = need_log ? __logf(xc * fc.inv_ia) : 0.0f;
    const float slope_th = fc.slope_th;
    const float sth32 = fc.sa32 + slope_th * lx;
    // Acceptance test of the approximate significance sm (DESIGN.md "jump significance fast path"):
    //   |sm - sme| <= r |sm| + t,  r = relerr + 4.1e-7 (host, folded into k1 >= 1/(1-r); k2 = 2 - k1 <= 1/(1+r)),
    //   t = 8e-7 |s| rs  (rounding of num*inv_dt and of q - s, using |q| <= |delta| + |s|);  the threshold itself is
    //   known to +-band0.  With thp + t > 0 and thm - t > 0 (checked per pixel through rsmax >= every rs):
    //   sm > (thp + t) k1  => sure hit,   sm < (thm - t) k2  => surely no hit,   anything else (NaN included) -> exact pass.
    const float band0 = 2e-6f * fabsf(sth32) + 4.5e-6f * fabsf(slope_th) + 1e-30f;  // 4e-6: __logf; 0.5e-6: its argument
    const float s2 = st.s2;
    const float s8 = fabsf(s) * 8.1e-7f;
    const float thp = sth32 + band0, thm = sth32 - band0;
    const float rsmax = __frsqrt_rn(tabsrc.amin() * s2) * 1.000001f;
    const bool pass = fmaf(s8, rsmax, band0) <= 0.5f * sth32;  // false for NaN, for amin == 0 and for sth32 <= 0
    const bool lane_exact = !(guard < 1e300) || !pass;
    uint32_t jfast = 0, unsure_mask = 0;
    rf2 dB[GP];  // (d[2p+1], d[2p+2])
#pragma unroll
    for (int p = 0; p < GP; ++p) dB[p] = rf2{dA[p].y, (p + 1 < GP) ? dA[p + 1].x : 0.0f};
#pragma unroll
    for (int ps = 0; ps < NS; ++ps) {
        if (ps + 1 < NS) tabsrc.pair(tab[(ps + 1) & 1], ps + 1);
        const int ip = ps / 2, di = (ps & 1) + 1;
        const uint32_t vbits = (valid >> (2 * ps)) & 3u;
        if (vbits == 0) continue;  // plan-uniform
        const rf2 hi = (di == 1) ? dB[ip] : ((ip + 1 < GP) ? dA[ip + 1] : rf2{0.0f, 0.0f});
        const rf2 lo = dA[ip];
        const RipDensePair &r = tab[ps & 1];
        const rf2 num = hi - lo;
        const rf2 q = num * rf2{r.inv_dt[0], r.inv_dt[1]};
        const rf2 delta = q - rf2{s, s};
        const rf2 as2 = rf2{r.A[0], r.A[1]} * s2;
        const rf2 var = __builtin_elementwise_fma(rf2{r.B[0], r.B[1]}, rf2{dv, dv}, as2);
        const rf2 rs = {__frsqrt_rn(var.x), __frsqrt_rn(var.y)};
        const rf2 sm = delta * rs;
        const rf2 t = rs * s8;
        const rf2 k1 = {r.k1[0], r.k1[1]};
        const rf2 hiT = (t + thp) * k1;
        const rf2 loT = (rf2{thm, thm} - t) * (rf2{2.0f, 2.0f} - k1);  // 2 - k1 <= 1 - r <= 1/(1+r)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            if (!((vbits >> e) & 1u)) continue;  // plan-uniform
            const float sme = e ? sm.y : sm.x;
            const bool hit = sme > (e ? hiT.y : hiT.x);
            const bool nohit = sme < (e ? loT.y : loT.x);
            if (hit) jfast |= 1u << (2 * ps + e);
            if (lane_exact || !(hit || nohit)) unsure_mask |= 1u << (2 * ps + e);
        }
    }
    st.jfast = jfast;
    st.unsure = unsure_mask;
}

template <int G, uint32_t VALID = 0u>
__device__ __forceinline__ void fit_full_pk_a(const rf2 (&dA)[G / 2], const RipFitConst fc, const RipVariant v,
                                              const RipDense *__restrict__ dn, float gain, float rn, bool flag,
                                              double guard, RipFitState &st) {
    fit_full_pk_a_t<G, VALID, RipDenseK>(dA, fc, v, RipDenseK{dn}, gain, rn, flag, guard, st);
}

// second half: exact re-evaluation where the approximate significance was not decisive, jump mask
template <int G>
__device__ __forceinline__ void fit_full_pk_b(const rf2 (&dA)[G / 2], const RipPlanHeader *__restrict__ h,
                                              const RipFitConst fc, const RipDense *__restrict__ dn,
                                              const float *__restrict__ kv, const RipDiff *__restrict__ df, bool flag,
                                              const RipFitState &st, uint32_t &jmask) {
    constexpr int GP = G / 2;
    if (!st.live) return;
    const float s = st.s, dv = st.dv, s2 = st.s2, xc = st.xc;
    uint32_t jfast = st.jfast;
    const uint32_t unsure_mask = st.unsure;
    // differences whose approximate significance is within its error band of the threshold (or NaN): redo them in
    // the reference's exact operation order.  Rare; one wave-uniform test covers the whole pixel.
    if (__any(unsure_mask != 0 && flag)) {
        const float lxe = log_f32(xc / fc.ia);
        const double sth = KLD(h->sa) + KLD(h->dsb) * ((double)lxe / KLD(h->loglen));
        for (int bit = 0; bit < 4 * GP; ++bit) {
            if (!__any((unsure_mask >> bit) & 1u)) continue;
            if ((unsure_mask >> bit) & 1u) {
                const RipDiff rd = df[KLD(dn->kidx[bit])];
                // operands d[i], d[j] from the pairs (runtime indices only on this rare path)
                float di_ = 0.0f, dj_ = 0.0f;
#pragma unroll
                for (int p = 0; p < GP; ++p) {
                    di_ = (rd.i == 2 * p) ? dA[p].x : (rd.i == 2 * p + 1) ? dA[p].y : di_;
                    dj_ = (rd.j == 2 * p) ? dA[p].x : (rd.j == 2 * p + 1) ? dA[p].y : dj_;
                }
                const float de = (dj_ - di_) / rd.dt - s;
                const double varx = exact_variance<float>(h, kv, G, rd.i, rd.j, rd.dt, dv, s2);
                const float smx = de / (float)sqrt(varx);
                const bool hitx = (double)smx > sth;
                jfast = hitx ? (jfast | (1u << bit)) : (jfast & ~(1u << bit));
            }
        }
    }
    // bit 2*ps+e of jfast -> JUMP_DET on group i = 2*(ps/2) + e
    if (flag) {
#pragma unroll
        for (int ip = 0; ip < GP; ++ip) {
            const uint32_t four = (jfast >> (4 * ip)) & 15u;  // (di=1: e0,e1), (di=2: e0,e1)
            if ((four & 1u) || (four & 4u)) jmask |= 1u << (2 * ip);
            if ((four & 2u) || (four & 8u)) jmask |= 1u << (2 * ip + 1);
        }
    }
}

template <int G>
__device__ __forceinline__ void fit_full_pk(const rf2 (&dA)[G / 2], const RipPlanHeader *__restrict__ h,
                                            const RipFitConst fc, const RipVariant v, const RipDense *__restrict__ dn,
                                            const float *__restrict__ kv, const RipDiff *__restrict__ df, float gain,
                                            float rn, bool flag, double guard, float &s_out, float &er_out,
                                            float &ep_out, uint32_t &jmask) {
    RipFitState st;
    fit_full_pk_a<G>(dA, fc, v, dn, gain, rn, flag, guard, st);
    fit_full_pk_b<G>(dA, h, fc, dn, kv, df, flag, st, jmask);
    s_out = st.s;
    er_out = st.er;
    ep_out = st.ep;
}

// ---------------------------------------------------------------------------------------------
// Saturation-truncated refits (fitting.py:326-337) with the ramp in registers: compile-time recursion over the
// truncation length GV = G-1 ... 3.  A layer is evaluated only when some lane of the wave first saturates at
// group GV (wave-uniform test); results and jump flags replace the running ones for those lanes, in the
// reference's order (descending GV).
template <int G, int GV>
__device__ __forceinline__ void trunc_layers(const float (&d)[G], const uint32_t (&qe)[G],
                                             const RipPlanHeader *__restrict__ h, const RipVariant *__restrict__ vars,
                                             const float *__restrict__ kvals, const RipDiff *__restrict__ diffs,
                                             float gain, float rn, bool act, double guard, float &s, float &er,
                                             float &ep, uint32_t &jmask) {
    if constexpr (GV >= 3) {
        if (GV >= 3 + KLD(h->start)) {
            const bool layer = ((qe[GV] & ~qe[GV - 1]) & DQ_SATURATED) != 0;
            if (__any(layer)) {
                float dt[GV];
#pragma unroll
                for (int t = 0; t < GV; ++t) dt[t] = d[t];
                const RipVariant v = rip_load_variant(vars, G - GV);
                float s_, er_, ep_;
                uint32_t jm = 0;
                fit_full_regs<GV>(dt, h, v, kvals + v.k_ofs, diffs + v.diff_ofs, gain, rn, act, guard, s_, er_, ep_, jm);
                if (layer) {
                    s = s_;
                    er = er_;
                    ep = ep_;
                    jmask |= jm;
                }
            }
        }
        trunc_layers<G, GV - 1>(d, qe, h, vars, kvals, diffs, gain, rn, act, guard, s, er, ep, jmask);
    }
}

// flag propagation of fitting.py:339-353 from registers; writes the updated group flags when gdq_out != null
template <int G>
__device__ __forceinline__ uint32_t propagate_flags(const uint32_t (&qe)[G], uint32_t jmask, int start, uint32_t pdq_in,
                                                    uint8_t *gdq_out, unsigned gstride, unsigned lane_off = 0) {
    uint32_t or_unsat = 0, any_sat = 0;
    bool all_dnu = true;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const uint32_t rq = qe[g] | (((jmask >> g) & 1u) ? DQ_JUMP_DET : 0u);
        if (gdq_out) *(gdq_out + (size_t)g * gstride + lane_off) = (uint8_t)rq;  // uniform base + per-lane offset
        if ((rq & DQ_SATURATED) == 0) or_unsat |= rq;
        any_sat |= rq & DQ_SATURATED;
        all_dnu = all_dnu && ((rq & DQ_DO_NOT_USE) != 0);
    }
    uint32_t pdq2 = or_unsat & ~DQ_DO_NOT_USE;
    if (all_dnu) pdq2 |= DQ_DO_NOT_USE;
    const uint32_t q_early = start ? qe[(G > 2) ? 2 : G - 1] : qe[1];  // rdq[1 + start]
    if (q_early & DQ_SATURATED) pdq2 |= DQ_DO_NOT_USE;
    pdq2 |= any_sat;
    return (pdq_in & DQ_REFERENCE_PIXEL) ? pdq_in : (pdq_in | pdq2);
}

// ---------------------------------------------------------------------------------------------
// propagate_flags on the pixel's group flags PACKED four to a word (w0 = groups 0-3, w1 = groups 4-7, bytes of
// groups >= G are zero): the same results with byte-parallel logic instead of G unpack / test / select sequences.
//   gdq_row  uniform byte pointer to element (group 0, row, column 0) of the output group flags, or null
__device__ __forceinline__ uint32_t rip_spread_bit1(uint32_t t) {  // bytes holding 0x02 -> 0xFF, others (0x00) -> 0x00
    t |= t >> 1;
    t |= t << 2;
    t |= t << 4;
    return t;
}
template <int G>
__device__ __forceinline__ uint32_t propagate_flags_packed(const uint32_t (&w)[(G + 3) / 4], uint32_t jmask, int start,
                                                           uint32_t pdq_in, uint8_t *gdq_row, unsigned gstride,
                                                           unsigned lane_off, uint32_t *rq_out = nullptr) {
    static_assert(G > 4 && G <= 16, "flag words");
    constexpr int QW = (G + 3) / 4;
    // bytes of the last word beyond group G-1: missing groups count as DO_NOT_USE in the all-groups test
    constexpr uint32_t PADL = (G % 4 == 0) ? 0u : (0x01010101u << (8 * (G % 4)));
    uint32_t rq[QW];
#pragma unroll
    for (int i = 0; i < QW; ++i)  // bit g of jmask -> JUMP_DET (0x04) of byte g
        rq[i] = w[i] | ((__umul24((jmask >> (4 * i)) & 0xFu, 0x00204081u) & 0x01010101u) << 2);
    if (rq_out) {  // the updated group flags, still packed (the caller stores them later)
#pragma unroll
        for (int i = 0; i < QW; ++i) rq_out[i] = rq[i];
    }
    if (gdq_row) {  // uniform
        uint8_t *p = gdq_row;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            *(p + lane_off) = (uint8_t)(rq[g / 4] >> (8 * (g & 3)));
            p += gstride;
        }
    }
    uint32_t red = 0, alld = 0x01010101u;
#pragma unroll
    for (int i = 0; i < QW; ++i) {
        const uint32_t sat = rq[i] & 0x02020202u;                      // SATURATED bit of every group
        red |= (rq[i] & ~rip_spread_bit1(sat)) | sat;                  // flags of the unsaturated groups, plus the bit itself
        alld &= (i == QW - 1) ? (rq[i] | PADL) : rq[i];
    }
    red |= red >> 16;  // OR over groups: bit 1 = any saturated, other bits = OR of the unsaturated groups' flags
    red |= red >> 8;
    uint32_t pdq2 = red & 0xFEu;  // without DO_NOT_USE
    const bool all_dnu = (alld & 0x01010101u) == 0x01010101u;
    const bool early = ((w[0] >> (8 * (1 + start))) & DQ_SATURATED) != 0;  // rdq[1 + start]
    if (all_dnu || early) pdq2 |= DQ_DO_NOT_USE;
    return (pdq_in & DQ_REFERENCE_PIXEL) ? pdq_in : (pdq_in | pdq2);
}
