// Up-the-ramp slope fit, jump detection, saturation-truncated refits, flag propagation and the
// post-fit algebra (dark rate, error split, flat) -- one thread per pixel, the pixel's ramp staged
// in LDS ([group][thread], conflict-free) so that the group count is a run-time value.
//
// Replaces (reference file:line):
//   utils/fitting.py:89-255   jump_detect   -> fit_variant()
//   utils/fitting.py:258-355  ramp_fit      -> rampfit_kernel() (variant loop + flag propagation)
//   L1_to_L2/gen_cal_image.py:458-475, 213-229, 607-629 -> the `finish` block
// Arithmetic recipe: oracle/rampfit.py, oracle/finish.py (bit-for-bit pinned to the reference).
//
// Roofline: HBM.  Algorithmic bytes per pixel = G*(4+1+1) + 4 (pdq) + gain + 4 (read) + 16 (outputs)
// (+ dark_rate, flat, flat_flags when finishing).
#include "rip_common.h"

#define RF_THREADS 256

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

// one jump_detect pass on the ramp D[t*RF_THREADS] (LDS); ORs JUMP_DET into J[i*RF_THREADS] when `flag`
template <typename GT>
__device__ __forceinline__ void fit_variant(const float *D, uint8_t *J, const RipPlanHeader *__restrict__ h,
                                            const RipVariant v, const float *__restrict__ kv,
                                            const RipDiff *__restrict__ df, GT gain, float rn, bool flag,
                                            double guard, float &s_out, float &er_out, float &ep_out) {
    const int g = v.g;
    const float d1 = D[RF_THREADS];
    float s = 0.0f;
    for (int t = 0; t < g; ++t) {
        float diff = D[t * RF_THREADS] - d1;
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
        float num = D[r.j * RF_THREADS] - D[r.i * RF_THREADS];
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
        if (hit) J[r.i * RF_THREADS] |= (uint8_t)DQ_JUMP_DET;
    }
}

template <typename GT>
__global__ __launch_bounds__(RF_THREADS) void rampfit_kernel(RampFitArgs a, const RipPlanHeader *__restrict__ h,
                                                             const RipVariant *__restrict__ vars,
                                                             const float *__restrict__ kvals,
                                                             const RipDiff *__restrict__ diffs, double guard) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int G = a.ngrp;
    float *Dl = reinterpret_cast<float *>(lds_raw);                 // [G][RF_THREADS]
    uint8_t *Ql = lds_raw + (size_t)G * RF_THREADS * sizeof(float);  // [G][RF_THREADS] input group flags
    uint8_t *Jl = Ql + (size_t)G * RF_THREADS;                       // [G][RF_THREADS] new JUMP_DET flags

    const int tid = threadIdx.x;
    const size_t npix = (size_t)a.ny * a.nx;
    const size_t p = (size_t)blockIdx.x * RF_THREADS + tid;
    if (p >= npix) return;  // no barriers below: every thread only touches its own LDS column
    const int y = (int)(p / a.nx), x = (int)(p % a.nx);
    const bool active = (y >= a.nb) && (y < a.ny - a.nb) && (x >= a.nb) && (x < a.nx - a.nb);

    float *D = Dl + tid;
    uint8_t *Q = Ql + tid;
    uint8_t *J = Jl + tid;
    for (int g = 0; g < G; ++g) {
        D[g * RF_THREADS] = a.cube[(size_t)g * npix + p];
        Q[g * RF_THREADS] = a.gdq_in[(size_t)g * npix + p];
        J[g * RF_THREADS] = 0;
    }
    const GT gain = reinterpret_cast<const GT *>(a.gain)[p];
    const float rn = a.read_noise[p];
    const int start = h->start;

    // full ramp (fitting.py:313-320): jump flags kept only where the last group is not saturated
    float s, er, ep;
    const bool unsat = (Q[(G - 1) * RF_THREADS] & DQ_SATURATED) == 0;
    fit_variant<GT>(D, J, h, vars[0], kvals + vars[0].k_ofs, diffs + vars[0].diff_ofs, gain, rn, unsat && active, guard,
                    s, er, ep);

    // ramps truncated at the group where the pixel first saturates (fitting.py:326-337)
    for (int vi = 1; vi < h->nvariants; ++vi) {
        const RipVariant v = vars[vi];
        const int iend = v.g;
        const bool layer = ((Q[iend * RF_THREADS] & ~Q[(iend - 1) * RF_THREADS]) & DQ_SATURATED) != 0;
        if (layer) fit_variant<GT>(D, J, h, v, kvals + v.k_ofs, diffs + v.diff_ofs, gain, rn, active, guard, s, er, ep);
    }

    // flag propagation (fitting.py:339-353)
    uint32_t or_unsat = 0, any_sat = 0;
    bool all_dnu = true;
    for (int g = 0; g < G; ++g) {
        uint32_t r = (uint32_t)Q[g * RF_THREADS] | (uint32_t)J[g * RF_THREADS];
        if (a.gdq_out) a.gdq_out[(size_t)g * npix + p] = (uint8_t)r;
        if ((r & DQ_SATURATED) == 0) or_unsat |= r;
        any_sat |= r & DQ_SATURATED;
        all_dnu = all_dnu && ((r & DQ_DO_NOT_USE) != 0);
    }
    uint32_t pdq2 = or_unsat & ~DQ_DO_NOT_USE;
    if (all_dnu) pdq2 |= DQ_DO_NOT_USE;
    if (Q[(1 + start) * RF_THREADS] & DQ_SATURATED) pdq2 |= DQ_DO_NOT_USE;
    pdq2 |= any_sat;
    uint32_t pdq = a.pdq_in[p];
    if ((pdq & DQ_REFERENCE_PIXEL) == 0) pdq |= pdq2;

    if (a.finish) {
        // gen_cal_image.py:458-475: err = hypot, var_poisson = ep^2, trim + zero border
        float err = hypot_f32(er, ep);
        float vp = ep * ep;
        if (!active) {
            s = 0.0f;
            err = 0.0f;
            vp = 0.0f;
        }
        // :213-229 dark rate on the active region
        if (active && a.dark_rate) s = s - a.dark_rate[p];
        if (active && a.dark_dq) pdq |= a.dark_dq[p];
        // :607-613
        float ep2 = sqrtf(vp);
        float e2 = err * err;
        float p2 = ep2 * ep2;
        float er2 = sqrtf(clip_lo<float>(e2 - p2, 0.0f));
        // :616-629
        if (a.flat) {
            if (a.flat_flags) pdq |= a.flat_flags[p];
            const float f = a.flat[p];
            s = s / f;
            er2 = er2 / f;
            ep2 = ep2 / f;
        }
        er = er2;
        ep = ep2;
    }
    a.slope[p] = s;
    a.err_read[p] = er;
    a.err_poisson[p] = ep;
    a.pdq_out[p] = pdq;
}

int rip_launch_rampfit(rip_ctx *ctx, const RipPlan *plan, const RampFitArgs &a, int gain_dtype) {
    if (a.ngrp != plan->h.ngrp) return rip_fail(ctx, RIP_EINVAL, "ramp has %d groups, plan %d", a.ngrp, plan->h.ngrp);
    const size_t npix = (size_t)a.ny * a.nx;
    const unsigned blocks = (unsigned)((npix + RF_THREADS - 1) / RF_THREADS);
    const size_t lds = (size_t)a.ngrp * RF_THREADS * 6;
    const RipPlanHeader *h = reinterpret_cast<const RipPlanHeader *>(plan->dev);
    extern double rip_guard_band;
    if (gain_dtype == RIP_F64) {
        if (lds > 48 * 1024)
            RIP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(rampfit_kernel<double>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(rampfit_kernel<double>, dim3(blocks), dim3(RF_THREADS), lds, ctx->stream, a, h,
                           plan->d_variants, plan->d_k, plan->d_diffs, rip_guard_band);
    } else {
        if (lds > 48 * 1024)
            RIP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(rampfit_kernel<float>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(rampfit_kernel<float>, dim3(blocks), dim3(RF_THREADS), lds, ctx->stream, a, h,
                           plan->d_variants, plan->d_k, plan->d_diffs, rip_guard_band);
    }
    RIP_HIP(ctx, hipGetLastError());
    return RIP_OK;
}
