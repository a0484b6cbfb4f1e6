// Pseudo-Poisson noise layers ("O" directives of gen_noise_image.make_noise_cube, gen_noise_image.py:173-240): one deviate per
// pixel from the member of the Pearson family whose second, third and fourth moments match the given ones.
//
// Replaces L1_to_L2/GalPoisson/draw_with_tilnus.py: draw_from_Pearson (:12-135) and the parameter solvers / samplers it calls
//   classification (beta_1, beta_2, the type-1/3/4/5/6 regions)            :43-97
//   type 1 (beta):        solve_beta_parameters_vec / random_from_type1    :157-251
//   type 3 (gamma):       random_from_type3                                :264-289
//   type 4:               random_from_type4 (parameters :545-568)          :536-587
//   type 5 (inv. gamma):  solve_pearson5_parameters_vec / random_from_type5 :602-657
//   type 6 (beta prime):  solve_pearson6_params / random_from_type6        :671-722
// The PARAMETERS are the reference's formulas in f64, operation for operation (pinned by goldens made with the reference's own
// module: tests/golden/pearson_params.npz).  The DEVIATES come from a counter-based generator on the device (Philox-4x32-10
// keyed by seed, stream and element): the reference draws through scipy.stats on a numpy Generator, whose streams cannot be
// reproduced here -- the random part is statistically, not bit-wise, comparable (moments tested): PARITY UNPINNED for it.
// Samplers: gamma by Marsaglia & Tsang (shape < 1 by the U^(1/k) boost), beta and beta prime from two gammas, inverse gamma
// as scale / gamma, Pearson IV by the rejection method for log-concave densities on the angle variable (Heinrich 2004, sec. 7,
// the method the reference calls "Devroye"), here always in standard form (a = 1: acceptance ~ 1/4 for every parameter set, so
// the reference's second sampler for low acceptance is not needed); the normalisation needs Re lgamma(m + i nu/2), evaluated
// by the recurrence + Stirling series.
#include "rip_common.h"

namespace {

__device__ __forceinline__ void px_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    c[1] = (uint32_t)p1;
    c[3] = (uint32_t)p0;
    c[0] = n0;
    c[2] = n2;
}

// a stream of uniforms in (0, 1) with 52 random bits each, for one element: counter = (element, stream, block index)
struct PxRng {
    uint64_t seed;
    uint32_t elem_lo, elem_hi, stream, block;
    uint32_t buf[4];
    int have;
    __device__ PxRng(uint64_t s, uint64_t elem, uint32_t st) : seed(s), elem_lo((uint32_t)elem), elem_hi((uint32_t)(elem >> 32)), stream(st), block(0), have(0) {}
    __device__ void refill() {
        uint32_t c[4] = {elem_lo, elem_hi ^ (stream * 0x9E3779B1u), block++, 0x50656172u};
        uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
        for (int r = 0; r < 10; ++r) {
            px_round(c, k0, k1);
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        for (int i = 0; i < 4; ++i) buf[i] = c[i];
        have = 4;
    }
    __device__ double uniform() {
        if (have < 2) refill();
        const uint64_t hi = buf[--have], lo = buf[--have];
        const uint64_t bits = ((hi << 32) | lo) >> 12;                       // 52 bits
        return ((double)bits + 0.5) * (1.0 / 4503599627370496.0);          // (0, 1), never 0 or 1
    }
    __device__ double normal() {
        const double u1 = uniform(), u2 = uniform();
        return sqrt(-2.0 * log(u1)) * cospi(2.0 * u2);
    }
    // Gamma(k, 1): Marsaglia & Tsang (2000)
    __device__ double gamma(double k) {
        double boost = 1.0;
        if (k < 1.0) {
            boost = pow(uniform(), 1.0 / k);
            k += 1.0;
        }
        const double d = k - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
        for (int it = 0; it < 1000; ++it) {
            const double x = normal();
            double v = 1.0 + c * x;
            if (v <= 0.0) continue;
            v = v * v * v;
            const double u = uniform();
            if (u < 1.0 - 0.0331 * (x * x) * (x * x)) return boost * d * v;
            if (log(u) < 0.5 * x * x + d * (1.0 - v + log(v))) return boost * d * v;
        }
        return boost * d;  // (never reached in practice: acceptance > 95 %)
    }
};

// Re lgamma(x + i y), x > 0: recurrence up to Re z >= 12, then the Stirling series
__device__ double re_lgamma_complex(double x, double y) {
    double shift = 0.0;  // sum of log|z + k| over the recurrence steps
    while (x < 12.0) {
        shift += 0.5 * log(x * x + y * y);
        x += 1.0;
    }
    const double r2 = x * x + y * y, logr = 0.5 * log(r2), th = atan2(y, x);
    // (z - 1/2) log z - z + log(2 pi)/2, real part
    double re = (x - 0.5) * logr - y * th - x + 0.91893853320467274178;
    // sum B_2k / (2k (2k-1) z^(2k-1)): powers of 1/z
    const double ix = x / r2, iy = -y / r2;          // 1/z
    const double i2x = ix * ix - iy * iy, i2y = 2.0 * ix * iy;   // 1/z^2
    double px = ix, py = iy;                          // 1/z^(2k-1)
    const double coef[6] = {1.0 / 12.0, -1.0 / 360.0, 1.0 / 1260.0, -1.0 / 1680.0, 1.0 / 1188.0, -691.0 / 360360.0};
    for (int k = 0; k < 6; ++k) {
        re += coef[k] * px;
        const double nx = px * i2x - py * i2y, ny = px * i2y + py * i2x;
        px = nx;
        py = ny;
    }
    return re - shift;
}

// Pearson IV in standard form (a = 1, lambda = 0), density ~ (1 + x^2)^(-m) exp(-nu atan x): rejection on the angle
__device__ double pearson4_std(PxRng &g, double m, double nu) {
    const double b = 2.0 * m - 2.0;
    const double M = atan2(-nu, b);
    const double cosM = b / hypot(b, nu);
    const double r_const = b * log(cosM) - nu * M;
    const double logk = (2.0 * m - 2.0) * 0.69314718055994530942 + 2.0 * re_lgamma_complex(m, 0.5 * nu) -
                        (1.14472988584940017414 + lgamma(2.0 * m - 1.0));
    const double rc = exp(-r_const - logk);
    for (int it = 0; it < 10000; ++it) {
        double z = 0.0, x = 4.0 * g.uniform();
        int s = 0;
        if (x > 2.0) {
            x -= 2.0;
            s = 1;
        }
        if (x > 1.0) {
            const double l = log(x - 1.0);
            z = l;
            x = 1.0 - l;
        }
        x = s ? (M + rc * x) : (M - rc * x);
        if (fabs(x) >= 1.57079632679489661923) continue;
        if (z + log(g.uniform()) > b * log(cos(x)) - nu * x - r_const) continue;
        return tan(x);
    }
    return tan(M);
}

enum { PX_NONE = 0, PX_T1 = 1, PX_T3 = 3, PX_T4 = 4, PX_T5 = 5, PX_T6 = 6 };

// classification and parameters of one element (draw_with_tilnus.py:43-97 and the solve_* functions); par[0..3]:
//   type 1: a, b, mean, c          type 3: shape, scale, shift, sign        type 4: m, nu, a, lambda
//   type 5: a, b, mu, sign         type 6: alpha, beta, scale, shift  (sign = +1 when tilnu_31 >= 0)
__device__ int pearson_classify(double t21, double t31, double t41, double I_in, double (&par)[4]) {
    const double I = I_in < 0.01 ? 0.01 : I_in;   // np.clip(I, 0.01, None): NaN stays NaN
    const double t42 = 3.0 * (t21 * t21);
    const double b1 = (t31 * t31) / ((t21 * t21 * t21) * I);
    const double b2 = (t42 * I + t41) / ((t21 * t21) * I);
    par[0] = par[1] = par[2] = par[3] = 0.0;
    const bool base = (b2 > 0.0) && (b1 >= 0.0) && (b2 > b1 + 1.0) && (b2 > 0.75 * b1);
    if (!base) return PX_NONE;
    const double rhs1 = 1.5 * b1 + 3.0;
    const double rhs2 = (48.0 + 39.0 * b1 + 6.0 * pow(4.0 + b1, 1.5)) / (32.0 - b1);
    if (b2 < rhs1) {
        // ---- type 1: analytic_u_v_from_betas, ab_from_u_v, central_moments_beta
        const double u_denom = (b2 - 3.0) - 1.5 * b1;
        const double u = 3.0 * (b1 - b2 + 1.0) / u_denom;
        const double v = b1 * ((u + 2.0) * (u + 2.0)) / (4.0 * (u + 1.0));
        const double s = sqrt(v / (v + 4.0));
        const double ap = 0.5 * u * (1.0 + s), bp = 0.5 * u * (1.0 - s);
        const bool cond = (t31 < 0.0) ? (ap > bp) : (ap < bp);
        const double a = cond ? ap : bp, b = cond ? bp : ap;
        const double mean = a / (a + b);
        const double var = a * b / (((a + b) * (a + b)) * (a + b + 1.0));
        par[0] = a, par[1] = b, par[2] = mean, par[3] = sqrt((t21 * I) / var);
        return PX_T1;
    }
    if (b2 == rhs1) {
        const double scale = fabs(t31) / (2.0 * t21);
        const double shape = 4.0 * (t21 * t21 * t21) * I / (t31 * t31);
        par[0] = shape, par[1] = scale, par[2] = shape * scale, par[3] = t31 > 0.0 ? 1.0 : -1.0;
        return PX_T3;
    }
    if (b2 == rhs2) {
        const double sq = sqrt(4.0 + b1);
        const double pp = 4.0 * (1.0 + 2.0 / b1 + sq / b1), pm = 4.0 * (1.0 + 2.0 / b1 - sq / b1);
        const double p = pp > 4.0 ? pp : pm;
        const double sigma = sqrt(t21 * I);
        const double g5 = sigma * (p - 2.0) * sqrt(p - 3.0);
        par[0] = p - 1.0, par[1] = g5, par[2] = g5 / (p - 1.0 - 1.0), par[3] = t31 >= 0.0 ? 1.0 : -1.0;
        return PX_T5;
    }
    if (b2 > rhs1 && b2 < rhs2) {
        const double r = 6.0 * (b2 - b1 - 1.0) / (3.0 * b1 - 2.0 * b2 + 6.0);
        const double eps = (r * r) / (4.0 + (b1 / 4.0) * ((r + 2.0) * (r + 2.0)) / (r + 1.0));
        const double d = sqrt(r * r - 4.0 * eps);
        const double q1 = (2.0 - r + d) / 2.0, q2 = (r - 2.0 + d) / 2.0;
        const double alpha = q2 + 1.0, beta = q1 - q2 - 1.0;
        const double var1 = alpha * (alpha + beta - 1.0) / ((beta - 2.0) * ((beta - 1.0) * (beta - 1.0)));
        const double scale = sqrt(t21 * I / var1);
        par[0] = alpha, par[1] = beta, par[2] = scale, par[3] = scale * (alpha / (beta - 1.0));
        return PX_T6;
    }
    if (b2 > rhs2 && b1 < 32.0) {
        const double mu2 = t21 * I;
        const double denom = 2.0 * b2 - 3.0 * b1 - 6.0;
        const double r = 6.0 * (b2 - b1 - 1.0) / denom;
        const double inner = 16.0 * (r - 1.0) - b1 * ((r - 2.0) * (r - 2.0));
        if (!(r > 1.0) || !(inner > 0.0)) return PX_NONE;   // (the reference raises ValueError here)
        const double nu_mag = r * (r - 2.0) * sqrt(b1) / sqrt(inner);
        const double nu = (t31 >= 0.0 ? -1.0 : 1.0) * nu_mag;
        const double a = sqrt(mu2 * inner) / 4.0;
        const double m = r / 2.0 + 1.0;
        par[0] = m, par[1] = nu, par[2] = a, par[3] = a * nu / (2.0 * (m - 1.0));
        return PX_T4;
    }
    return PX_NONE;
}

__global__ __launch_bounds__(256) void pearson_kernel(const double *__restrict__ I, size_t n, double t21, double t31, double t41,
                                                      uint64_t seed, uint32_t stream, int draw, double *__restrict__ out,
                                                      int32_t *__restrict__ type_out, double *__restrict__ par_out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double par[4];
    const int type = pearson_classify(t21, t31, t41, I[i], par);
    if (type_out) type_out[i] = type;
    if (par_out) {
        for (int k = 0; k < 4; ++k) par_out[i * 4 + k] = par[k];
    }
    if (!draw || !out) return;
    PxRng g(seed, i, stream);
    double x = 0.0;
    const double sgn = t31 >= 0.0 ? 1.0 : -1.0;
    switch (type) {
        case PX_T1: {
            const double ga = g.gamma(par[0]), gb = g.gamma(par[1]);
            x = par[3] * (ga / (ga + gb) - par[2]);
            break;
        }
        case PX_T3:
            x = par[3] * (par[1] * g.gamma(par[0]) - par[2]);
            break;
        case PX_T4:
            x = par[2] * pearson4_std(g, par[0], par[1]) + par[3];
            break;
        case PX_T5:
            x = par[3] * (par[1] / g.gamma(par[0]) - par[2]);
            break;
        case PX_T6: {
            const double ga = g.gamma(par[0]), gb = g.gamma(par[1]);
            x = sgn * (par[2] * (ga / gb) - par[3]);
            break;
        }
        default:
            break;
    }
    out[i] = x;
}

}   // namespace

// host arrays in and out; draws / types / params may each be NULL
extern "C" int rip_stage_pearson(rip_ctx *ctx, size_t n, const double *I, double tilnu21, double tilnu31, double tilnu41,
                                 uint64_t seed, uint32_t stream, double *draws, int32_t *types, double *params) {
    if (!I || (!draws && !types && !params)) return rip_fail(ctx, RIP_EINVAL, "pearson: NULL argument");
    if (n == 0) return RIP_OK;
    RIP_HIP(ctx, hipSetDevice(ctx->device));
    double *d_I = nullptr, *d_o = nullptr, *d_p = nullptr;
    int32_t *d_t = nullptr;
    int rc = RIP_OK;
    auto done = [&]() {
        for (void *p : {(void *)d_I, (void *)d_o, (void *)d_p, (void *)d_t})
            if (p) (void)hipFree(p);
    };
#define PX_HIP(call)                                                                   \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) {                                                        \
            rc = rip_fail(ctx, RIP_EHIP, "%s: %s", #call, hipGetErrorString(e_));      \
            done();                                                                    \
            return rc;                                                                 \
        }                                                                              \
    } while (0)
    PX_HIP(hipMalloc((void **)&d_I, n * 8));
    if (draws) PX_HIP(hipMalloc((void **)&d_o, n * 8));
    if (types) PX_HIP(hipMalloc((void **)&d_t, n * 4));
    if (params) PX_HIP(hipMalloc((void **)&d_p, n * 32));
    PX_HIP(hipMemcpyAsync(d_I, I, n * 8, hipMemcpyDefault, ctx->stream));
    hipLaunchKernelGGL(pearson_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_I, n, tilnu21, tilnu31,
                       tilnu41, seed, stream, draws ? 1 : 0, d_o, d_t, d_p);
    PX_HIP(hipGetLastError());
    if (draws) PX_HIP(hipMemcpyAsync(draws, d_o, n * 8, hipMemcpyDefault, ctx->stream));
    if (types) PX_HIP(hipMemcpyAsync(types, d_t, n * 4, hipMemcpyDefault, ctx->stream));
    if (params) PX_HIP(hipMemcpyAsync(params, d_p, n * 32, hipMemcpyDefault, ctx->stream));
    PX_HIP(hipStreamSynchronize(ctx->stream));
#undef PX_HIP
    done();
    return RIP_OK;
}
