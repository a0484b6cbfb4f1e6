// Level-1 synthesis on the device: the per-pixel work of from_sim/sim_to_isim.py between an electron-count
// image and the raw u16 exposure --
//   rip_synth_apportion    romanisim.l1.apportion_counts_to_resultants, the sampling: binomial shares of the counts per read
//   rip_synth_resultants   make_l1_fullcal (sim_to_isim.py:163-262): reset noise, per-read IPC + inverse linearity
//                          (IL.apply, ipc_linearity.py:461-513), mean over the reads of a resultant, read noise, biascorr, rounding
//   rip_synth_fill         fill_in_refdata_and_1f (:306-403): reference pixels, correlated noise, reference output
//   rip_synth_extract_ref  the EXTRACT_REF block (:711-730)
// Everything is resident in HBM (device pointers in and out) and asynchronous on the context's stream: the many-realisations
// harness makes 256 exposures of 4096 x 4096 x 8 without a byte crossing PCIe.  Arithmetic in numpy's operation order and
// dtypes (numpy 2 promotion rules; contraction off), so that given the same deviates the results equal the reference's
// functions bit for bit (tests/golden/l1sim.npz, made by executing them).  One thread per pixel; these kernels stream each
// array once and are far from any roofline concern next to the 24 x nreads Legendre evaluations per pixel.
#include <cmath>
#include <cstring>

#include "rip_common.h"

#include "invlin_device.h"
#include "rip_rng.h"

namespace {

constexpr uint32_t TAG_TOTAL = 0x10, TAG_SHARE = 0x11, TAG_RESET = 0x12, TAG_READ = 0x13, TAG_FILL = 0x14, TAG_WHITE33 = 0x15;
constexpr int MAX_READS = 1024;

struct ShareTable {
    double p[MAX_READS];    // share of the electrons still to come that read r collects
    double w[MAX_READS];    // share of ALL the electrons that read r collects, its square root and its logarithm
    double sw[MAX_READS];
    double lw[MAX_READS];
};

constexpr unsigned DEFER_LISTS = 256, DEFER_HEAD = 2 * DEFER_LISTS * 32;
// capacity of one list: the pixels of the workgroups that feed it
__host__ __device__ inline size_t defer_cap(size_t npix) { return (((npix + 255) / 256 + DEFER_LISTS - 1) / DEFER_LISTS) * 256; }

// Poisson increments of one pixel (see apportion_kernel).  NT: entries of the table of cumulative probabilities (12 cover a mean of 4
// -- the first launch -- to one draw in a thousand; 28 for the deferred pixels, which would cover a mean of 10, measured slower:
// 3.4 against 2.8 ms for a frame at 4.7 electrons per read -- the comparisons cost more than the rare loop)
// BIG = false: every mean is below 10 (the first launch when the brighter pixels are deferred): no transformed rejection in the code,
// half the registers
template <int NT, bool BIG>
__device__ __forceinline__ void apportion_poisson_pixel(size_t i, double c, size_t npix, int nreads, const double *__restrict__ share,
                                                        uint64_t seed, int32_t *__restrict__ out) {
    const double sc = sqrt(c), lc = (c > 0.0) ? log(c) : 0.0;
    const double *w = share + MAX_READS, *sw = share + 2 * MAX_READS, *lw = share + 3 * MAX_READS;
    double got_d = 0.0;
    double w_prev = -1.0, lam = 0.0;
    float lam32 = 0.0f;
    bool small = false;
    riprng::PtrsPlan plan{};
    // small means: the cumulative probabilities of the first NT counts once per run of reads of one mean (registers), a deviate is
    // then NT comparisons without a loop -- the sequential search of the same sums (cdf_j = cdf_(j-1) + p_j, p_j = p_(j-1) lam / j,
    // stopped where p_j <= 1e-12 cannot move the sum any more: those entries are +inf), continued in a loop past the table
    float cdf_tab[NT], p_end = 0.0f, cdf_end = 0.0f;
#pragma unroll
    for (int j = 0; j < NT; ++j) cdf_tab[j] = 0.0f;
    for (int r0 = 0; r0 < nreads; r0 += 4) {
        uint32_t cw_[4] = {(uint32_t)i, (uint32_t)r0, TAG_TOTAL, 0x706f6934u};
        riprng::philox(cw_, seed);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = r0 + q;
            if (r >= nreads) break;
            if (fabs(w[r] - w_prev) > 1e-12 * w_prev) {
                w_prev = w[r];
                lam = c * w[r];
                small = lam > 0.0 && lam < 10.0;
                lam32 = (float)lam;
                if (small) {
                    float p_ = __expf(-lam32), cdf = p_;
                    bool live = true;
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        live = live && p_ > 1e-12f;
                        cdf_tab[j] = live ? cdf : __builtin_inff();
                        p_ *= lam32 * __builtin_amdgcn_rcpf((float)(j + 1));
                        cdf += p_;
                    }
                    p_end = p_, cdf_end = cdf;
                }
                if constexpr (BIG)
                    if (!small && lam > 0.0) plan = riprng::ptrs_plan(lam, sc * sw[r], lc + lw[r]);
            }
            double k = 0.0;
            if (small) {
                const float u = (float)(cw_[q] >> 9) * (1.0f / 8388608.0f) + (1.0f / 16777216.0f);   // in (0, 1), never 1
                int kk = 0;
#pragma unroll
                for (int j = 0; j < NT; ++j) kk += (u > cdf_tab[j]) ? 1 : 0;
                if (kk == NT) {
                    float p_ = p_end, cdf = cdf_end;
                    while (u > cdf && p_ > 1e-12f && kk < 200) {   // (p below 1e-12: the sum cannot grow any more)
                        ++kk;
                        p_ *= lam32 * __builtin_amdgcn_rcpf((float)kk);
                        cdf += p_;
                    }
                }
                k = (double)kk;
            } else if (lam > 0.0) {
                if constexpr (BIG) k = riprng::poisson_ptrs(plan, seed, (uint32_t)i, (uint32_t)r, TAG_TOTAL);
            }
            got_d += k;
            out[(size_t)r * npix + i] = (int)(got_d > 2.0e9 ? 2.0e9 : got_d);
        }
    }
}

// `defer` (Poisson only): pixels whose largest mean per read reaches defer_lam are not done here but appended to a list for
// apportion_deferred_kernel (layout: DEFER_HEAD counter words, then 2 x DEFER_LISTS lists of defer_cap(npix) pixels).  A frame's bright and hot pixels are few (2 % of the bench scene)
// and scattered: half of the waves hold one, and a wave takes as long as its slowest lane -- twenty steps of the sequential
// search, or the transformed rejection, where the sky's lanes need three (6.2 -> 3 ms per 4096^2 frame of 35 reads).  The
// deviates are functions of (seed, pixel, read): the same electrons whichever kernel draws them.
// MODE 0: given totals (binomial shares); 1: Poisson increments, every pixel here; 2: Poisson increments, the brighter pixels deferred
// (separate instantiations: the rejection sampler's and the binomial's registers halve the occupancy of the sky's straight-line code)
template <int MODE>
__global__ __launch_bounds__(256) void apportion_kernel(const float *__restrict__ counts, size_t npix, int nreads,
                                                        const double *__restrict__ share, uint64_t seed, int32_t *__restrict__ out,
                                                        uint32_t *__restrict__ defer, double defer_lam, double w_max) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npix) return;
    double c = (double)counts[i];
    if constexpr (MODE != 0) {
        // A Poisson total split multinomially over the reads IS a set of independent Poisson increments of mean counts * share:
        // same joint distribution as drawing the total first and then the binomial shares (romanisim's order), one cheap deviate
        // per read instead of an expensive one (the binomial's acceptance test costs four log-gammas).
        // Means below 10 (a few electrons per read: the sky) by inversion in f32 -- the sequential search for the first k whose
        // cumulative probability passes a 23-bit uniform deviate -- with the exponential once per run of reads of one share (equal read
        // spacings give equal shares up to the rounding of the time differences) and ONE Philox block for four reads; larger means
        // by transformed rejection, its constants once per run of reads too and its acceptance test in f32 first (riprng::
        // poisson_ptrs).  Device deviates are unpinned by nature (tests: mean, variance, third moment, histogram, P(0)).
        c = c < 0.0 ? 0.0 : (c > 2.0e9 ? 2.0e9 : c);
        if (MODE == 2 && c * w_max >= defer_lam) {
            // two classes of lists -- means below 10 (sequential search) and above (transformed rejection) -- of DEFER_LISTS lists
            // each, a workgroup's pixels into list blockIdx.x mod DEFER_LISTS: the counters are a cache line apart, and the atomics
            // have a wave-uniform address (the compiler makes them one per wave).  One counter for the frame serialises: half of
            // the waves hold a deferred pixel, 130 k atomics on one line took longer than the arithmetic.
            const size_t cap = defer_cap(npix);
            const unsigned q = blockIdx.x % DEFER_LISTS;
            if (c * w_max >= 10.0)
                defer[DEFER_HEAD + (DEFER_LISTS + q) * cap + atomicAdd(defer + (DEFER_LISTS + q) * 32, 1u)] = (uint32_t)i;
            else
                defer[DEFER_HEAD + q * cap + atomicAdd(defer + q * 32, 1u)] = (uint32_t)i;
            return;
        }
        apportion_poisson_pixel<12, MODE == 1>(i, c, npix, nreads, share, seed, out);   // (MODE 2: what is left has fewer than defer_lam <= 10 per read)
    } else {
        c = c < 0.0 ? 0.0 : (c > 2.0e9 ? 2.0e9 : c);   // np.clip(counts, 0, 2e9).astype(i4)
        const int total = (int)c;
        int got = 0;
        for (int r = 0; r < nreads; ++r) {
            got += riprng::binomial(total - got, share[r], seed, (uint32_t)i, (uint32_t)r, TAG_SHARE);
            out[(size_t)r * npix + i] = got;
        }
    }
}

// where each of the 2 * DEFER_LISTS lists starts in the concatenation of all of them (class 0 first): defer[l * 32 + 1], the total
// in defer[2]; one workgroup of 2 * DEFER_LISTS threads
__global__ __launch_bounds__(2 * DEFER_LISTS) void defer_scan_kernel(uint32_t *__restrict__ defer) {
    __shared__ uint32_t sh[2 * DEFER_LISTS];
    const unsigned l = threadIdx.x;
    const uint32_t mine = defer[l * 32];
    sh[l] = mine;
    __syncthreads();
    for (unsigned d = 1; d < 2 * DEFER_LISTS; d <<= 1) {
        const uint32_t add = l >= d ? sh[l - d] : 0u;
        __syncthreads();
        sh[l] += add;
        __syncthreads();
    }
    defer[l * 32 + 1] = sh[l] - mine;
    if (l == 2 * DEFER_LISTS - 1) defer[2] = sh[l];
}

// thread t takes entry t of the concatenated lists of class CLS (0: means of 4 .. 10 per read, sequential search only; 1: 10 and
// more): waves are dense whatever the lists hold (a single wave needs 250 us for its 35 dependent draws: the launch is bound by
// waves in flight, not by arithmetic -- hence one instantiation per class, each with the registers of its own path only).  The grid
// strides over the entries (it is sized for a few per cent of the frame, the usual case, in one pass)
template <int CLS>
__global__ __launch_bounds__(256) void apportion_deferred_kernel(const float *__restrict__ counts, size_t npix, int nreads,
                                                                 const double *__restrict__ share, uint64_t seed, int32_t *__restrict__ out,
                                                                 const uint32_t *__restrict__ defer) {
    const uint32_t first = CLS ? defer[DEFER_LISTS * 32 + 1] : 0u, last = CLS ? defer[2] : defer[DEFER_LISTS * 32 + 1];
    for (uint32_t t = first + blockIdx.x * 256 + threadIdx.x; t < last; t += gridDim.x * 256) {
        unsigned lo = CLS ? DEFER_LISTS : 0, hi = CLS ? 2 * DEFER_LISTS : DEFER_LISTS;   // the last list whose start is <= t
        while (hi - lo > 1) {
            const unsigned mid = (lo + hi) >> 1;
            if (defer[mid * 32 + 1] <= t) lo = mid; else hi = mid;
        }
        const size_t i = defer[DEFER_HEAD + lo * defer_cap(npix) + (t - defer[lo * 32 + 1])];
        double c = (double)counts[i];
        c = c < 0.0 ? 0.0 : (c > 2.0e9 ? 2.0e9 : c);
        apportion_poisson_pixel<12, CLS == 1>(i, c, npix, nreads, share, seed, out);
    }
}

// numpy: f32 array (op)= array of GT -- computed in promote(f32, GT), stored back as f32
template <typename GT>
struct Wide {
    using type = float;
};
template <>
struct Wide<double> {
    using type = double;
};

template <typename GT>
__global__ __launch_bounds__(256) void reset_kernel(const float *__restrict__ normals, const float *__restrict__ resetnoise,
                                                    const GT *__restrict__ gain, const float *__restrict__ dark_slope, int has_bias,
                                                    float tbias, int ny, int nx, int nb, uint64_t seed, float *__restrict__ start) {
    using P = typename Wide<GT>::type;
    const int xa = blockIdx.x * 256 + threadIdx.x, ya = blockIdx.y;
    const int nxa = nx - 2 * nb;
    if (xa >= nxa) return;
    const size_t ia = (size_t)ya * nxa + xa, i = (size_t)(ya + nb) * nx + (xa + nb);
    float v = normals ? normals[ia] : riprng::normal_f32(seed, (uint32_t)ia, 0u, TAG_RESET);
    v = v * resetnoise[i];
    v = (float)((P)v * (P)gain[i]);
    if (has_bias) {
        const float td = tbias * dark_slope[i];               // Python float * f32 array: f32
        v = (float)((P)v - (P)td / (P)gain[i]);
    }
    start[ia] = v;
}

__device__ __constant__ int8_t SY_DY[9] = {0, 1, -1, 0, 0, 1, 1, -1, -1};
__device__ __constant__ int8_t SY_DX[9] = {0, 0, 0, 1, -1, 1, -1, 1, -1};

struct ResArgs {
    const int32_t *reads_e;     // (nreads, nya, nxa)
    const float *start;         // (nya, nxa)
    const void *gain;           // (ny, nx)
    const void *kern;           // (3,3,nya,nxa) or null
    const float *coefs, *smin, *smax;   // full frame
    const float *read_noise;    // full frame
    const float *normals_read;  // (ngrp, nya, nxa) or null
    const float *biascorr;      // (ngrp, nya, nxa) or null
    float *resultants;          // (ngrp, nya, nxa) or null
    uint16_t *cube;             // (ngrp, ny, nx) or null
    int ny, nx, nb, ngrp;
    uint64_t seed;
    int count[RIP_MAX_GROUPS];
    double root[RIP_MAX_GROUPS];   // len ** 0.5
};

template <typename GT, typename KT, int NP>
__global__ __launch_bounds__(256) void resultants_kernel(ResArgs a) {
    const int nxa = a.nx - 2 * a.nb, nya = a.ny - 2 * a.nb;
    const int xa = blockIdx.x * 256 + threadIdx.x, ya = blockIdx.y;
    if (xa >= nxa) return;
    const size_t nact = (size_t)nya * nxa, npix = (size_t)a.ny * a.nx;
    const size_t ia = (size_t)ya * nxa + xa, i = (size_t)(ya + a.nb) * a.nx + (xa + a.nb);
    // loop invariants of the pixel: the nine coefficients (each at ITS source pixel), the sources' reset electrons, offsets
    double kq[9], sq[9];
    long off[9];
    const KT *kern = (const KT *)a.kern;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int dy = SY_DY[k], dx = SY_DX[k];
        const int sy = ya - dy, sx = xa - dx;
        const bool in = sy >= 0 && sy < nya && sx >= 0 && sx < nxa && (kern || k == 0);
        off[k] = in ? (long)sy * nxa + sx : -1;
        kq[k] = in && kern ? (double)kern[(size_t)(3 * (1 + dy) + (1 + dx)) * nact + off[k]] : 1.0;
        sq[k] = in ? (double)a.start[off[k]] : 0.0;
    }
    float c[NP];
    double cw[NP];   // widened once (exact): invariant over the reads and the bisection steps
#pragma unroll
    for (int L = 0; L < NP; ++L) {
        c[L] = a.coefs[(size_t)L * npix + i];
        cw[L] = (double)c[L];
    }
    const float smin = a.smin[i], smax = a.smax[i];
    const double g = (double)((const GT *)a.gain)[i];
    const float rn = a.read_noise[i];
    // the bisection of consecutive reads shares the first steps of its path (rip_invlin_pixel_warm)
    float phi_path[24];
    uint32_t path = 0;
    bool have = false;
#pragma unroll
    for (int q = 0; q < 24; ++q) phi_path[q] = 0.0f;
    int r = 0;
    // the nine electron counts of a read are requested one read ahead: they land while the 24 bisection steps of the read before
    // run (a read's loads are L2 hits ~1 k cycles away, its bisection ~5 k cycles: without the prefetch a fifth of the time is
    // spent waiting for them)
    int nreads_all = 0;
    for (int j = 0; j < a.ngrp; ++j) nreads_all += a.count[j];
    int32_t en[9];
    auto fetch_read = [&](int rr) {
        const int32_t *e = a.reads_e + (size_t)(rr < nreads_all ? rr : nreads_all - 1) * nact;
#pragma unroll
        for (int k = 0; k < 9; ++k) en[k] = (off[k] >= 0) ? e[off[k]] : 0;
    };
    fetch_read(0);
    for (int j = 0; j < a.ngrp; ++j) {
        float acc = 0.0f;
        for (int q = 0; q < a.count[j]; ++q, ++r) {
            int32_t ec[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) ec[k] = en[k];
            fetch_read(r + 1);
            // ipc_fwd of (electrons + reset) in f64: centre product first, then the neighbours in the reference's order
            double conv = ((double)ec[0] + sq[0]) * kq[0];
            if (kern) {
#pragma unroll
                for (int k = 1; k < 9; ++k)
                    if (off[k] >= 0) conv = conv + ((double)ec[k] + sq[k]) * kq[k];
            }
            bool ex;
            const double S = rip_invlin_pixel_warm<double, NP>(conv / g, c, cw, smin, smax, ex, phi_path, path, have);
            acc = (float)((double)acc + S);
        }
        float res = acc / (float)a.count[j];
        const float nrm = a.normals_read ? a.normals_read[(size_t)j * nact + ia] : riprng::normal_f32(a.seed, (uint32_t)ia, (uint32_t)j, TAG_READ);
        res = (float)((double)res + (double)(nrm * rn) / a.root[j]);
        if (a.biascorr) res = res + a.biascorr[(size_t)j * nact + ia];
        res = rintf(res);
        if (a.resultants) a.resultants[(size_t)j * nact + ia] = res;
        if (a.cube) a.cube[(size_t)j * npix + i] = (uint16_t)(res < 0.0f ? 0.0f : (res > 65535.0f ? 65535.0f : res));
    }
}

// zeroes the nb-pixel border of every group's plane (blockIdx.y): one thread per border pixel -- the 2 nb full rows, then the
// 2 nb edge columns of the other rows
__global__ __launch_bounds__(256) void zero_border_kernel(uint16_t *cube, int ny, int nx, int nb) {
    const int t = blockIdx.x * 256 + threadIdx.x, full = 2 * nb * nx;
    int x, y;
    if (t < full) {
        const int q = t / nx;
        x = t - q * nx;
        y = q < nb ? q : ny - 2 * nb + q;
    } else {
        const int u = t - full;
        if (u >= (ny - 2 * nb) * 2 * nb) return;
        const int j = u % (2 * nb);
        y = nb + u / (2 * nb);
        x = j < nb ? j : nx - 2 * nb + j;
    }
    cube[((size_t)blockIdx.y * ny + y) * nx + x] = 0;
}

struct FillArgs {
    const float *normals;   // (ngrp+1, ny, nx) or null
    const float *frames;    // (ngrp, nch+2, ny, cw) or null: no banding
    const float *white33;   // (ngrp, ny, cw) or null
    const float *read_noise, *resetnoise, *dark, *med, *std;
    uint16_t *cube, *amp33;
    int ny, nx, nb, cw, ngrp, nch;
    float u_pink, c_pink, ru_pink, m_pink;
    uint64_t seed;
    float root[RIP_MAX_GROUPS];   // f32(len ** 0.5): the divisor of an f32 array by a Python float
};

__global__ __launch_bounds__(256) void fill_kernel(FillArgs a) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, j = blockIdx.z;
    if (x >= a.nx) return;
    const size_t npix = (size_t)a.ny * a.nx, i = (size_t)y * a.nx + x;
    float v;
    if (y < a.nb || y >= a.ny - a.nb || x < a.nb || x >= a.nx - a.nb) {
        const float n_own = a.normals ? a.normals[(size_t)j * npix + i] : riprng::normal_f32(a.seed, (uint32_t)i, (uint32_t)j, TAG_FILL);
        const float n_reset = a.normals ? a.normals[(size_t)a.ngrp * npix + i] : riprng::normal_f32(a.seed, (uint32_t)i, (uint32_t)a.ngrp, TAG_FILL);
        v = (n_own * a.read_noise[i]) / a.root[j];
        v = v + n_reset * a.resetnoise[i];
        v = v + a.dark[(size_t)j * npix + i];
    } else {
        v = (float)a.cube[(size_t)j * npix + i];
    }
    if (a.frames) {
        const int ch = x / a.cw, xc = x % a.cw;
        const int xs = (ch & 1) ? a.cw - 1 - xc : xc;
        const size_t fsz = (size_t)a.ny * a.cw;
        const float *fj = a.frames + (size_t)j * (a.nch + 2) * fsz;
        const float common = fj[(size_t)y * a.cw + xs] * a.c_pink;
        const float stripe = fj[(size_t)(1 + ch) * fsz + (size_t)y * a.cw + xs] * a.u_pink + common;
        v = v + stripe / a.root[j];
    }
    v = rintf(v);
    a.cube[(size_t)j * npix + i] = (uint16_t)(v < 0.0f ? 0.0f : (v > 65535.0f ? 65535.0f : v));   // NaN: not defined in numpy either
}

__global__ __launch_bounds__(256) void amp33_kernel(FillArgs a) {
    const int xc = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, j = blockIdx.z;
    if (xc >= a.cw) return;
    const size_t fsz = (size_t)a.ny * a.cw, i = (size_t)y * a.cw + xc;
    const float *fj = a.frames + (size_t)j * (a.nch + 2) * fsz;
    const float nrm = a.white33 ? a.white33[(size_t)j * fsz + i] : riprng::normal_f32(a.seed, (uint32_t)i, (uint32_t)j, TAG_WHITE33);
    const float white = nrm * a.std[i];
    const float common = fj[i] * a.c_pink;
    const float pink = a.ru_pink * fj[(size_t)(a.nch + 1) * fsz + i] + a.m_pink * common;
    const float level = a.med[i] + (white + pink) / a.root[j];
    a.amp33[(size_t)j * fsz + i] = (uint16_t)(long long)level;   // the C cast numpy's astype performs: towards zero, modulo 2^16
}

__global__ __launch_bounds__(256) void extract_ref_kernel(uint16_t *data, int ngrp, size_t n, int offset, uint16_t *ref) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint16_t first = data[i];
    if (ref) ref[i] = first;
    const int shift = (int)first - offset;
    for (int k = 1; k < ngrp; ++k) {
        const int v = (int)data[(size_t)k * n + i] - shift;
        data[(size_t)k * n + i] = (uint16_t)(v < 0 ? 0 : (v > 65535 ? 65535 : v));
    }
}

int check_cal(rip_ctx *ctx, const rip_synth_cal *c, int ngrp, const int32_t *group_count, const char *who) {
    if (!c || !group_count || ngrp < 1 || ngrp > RIP_MAX_GROUPS) return rip_fail(ctx, RIP_EINVAL, "%s: bad arguments", who);
    if (c->ny < 1 || c->nx < 1 || c->nb < 0 || 2 * c->nb >= c->ny || 2 * c->nb >= c->nx)
        return rip_fail(ctx, RIP_EINVAL, "%s: frame %d x %d with border %d", who, c->ny, c->nx, c->nb);
    for (int j = 0; j < ngrp; ++j)
        if (group_count[j] < 1) return rip_fail(ctx, RIP_EINVAL, "%s: resultant %d has %d reads", who, j, group_count[j]);
    return RIP_OK;
}

template <typename GT, typename KT>
int launch_resultants(rip_ctx *ctx, int np_, const ResArgs &a) {
    const dim3 grid((unsigned)((a.nx - 2 * a.nb + 255) / 256), (unsigned)(a.ny - 2 * a.nb)), block(256);
#define SY_CASE(N)                                                                                        \
    case N:                                                                                               \
        hipLaunchKernelGGL((resultants_kernel<GT, KT, N>), grid, block, 0, ctx->stream, a);                \
        break;
    switch (np_) {
        SY_CASE(2) SY_CASE(3) SY_CASE(4) SY_CASE(5) SY_CASE(6) SY_CASE(7) SY_CASE(8) SY_CASE(9) SY_CASE(10) SY_CASE(11) SY_CASE(12)
        SY_CASE(13) SY_CASE(14) SY_CASE(15) SY_CASE(16) SY_CASE(17)
        default:
            return rip_fail(ctx, RIP_EINVAL, "synth_resultants: %d coefficient planes (2..17 supported)", np_);
    }
#undef SY_CASE
    RIP_HIP(ctx, hipGetLastError());
    return RIP_OK;
}

}   // namespace

extern "C" int rip_synth_apportion(rip_ctx *ctx, const float *counts, int nya, int nxa, int poisson, int nreads, const double *t_reads,
                                   uint64_t seed, int32_t *reads_e) {
    ctx->stream_dirty = true;
    if (!counts || !t_reads || !reads_e || nya < 1 || nxa < 1 || nreads < 1 || nreads > MAX_READS)
        return rip_fail(ctx, RIP_EINVAL, "synth_apportion: bad arguments (1..%d reads)", MAX_READS);
    RIP_HIP(ctx, hipSetDevice(ctx->device));
    ShareTable tab;
    double t_end = t_reads[0];
    for (int r = 1; r < nreads; ++r) {
        if (t_reads[r] < t_reads[r - 1]) return rip_fail(ctx, RIP_EINVAL, "synth_apportion: read times must ascend");
        t_end = t_reads[r];
    }
    double t_prev = 0.0;
    memset(&tab, 0, sizeof tab);
    for (int r = 0; r < nreads; ++r) {
        const double left = t_end - t_prev;
        double p = left > 0.0 ? (t_reads[r] - t_prev) / left : 1.0;
        tab.p[r] = p < 0.0 ? 0.0 : (p > 1.0 ? 1.0 : p);
        double w = t_end > 0.0 ? (t_reads[r] - t_prev) / t_end : (r == nreads - 1 ? 1.0 : 0.0);
        w = w < 0.0 ? 0.0 : w;
        tab.w[r] = w;
        tab.sw[r] = sqrt(w);
        tab.lw[r] = w > 0.0 ? log(w) : 0.0;
        t_prev = t_reads[r];
    }
    const void *had = ctx->ws[10];
    double *d_tab = (double *)rip_ws(ctx, 10, sizeof(ShareTable));
    if (!d_tab) return RIP_ENOMEM;
    // the device copy of the table is kept between calls: an exposure after exposure of one read pattern (the many-realisations
    // harness) uploads it once and the call stays asynchronous; a new table waits for the kernels still reading the old one
    const bool same = (const void *)d_tab == had && (int)ctx->share_tab.size() == nreads &&
                      memcmp(ctx->share_tab.data(), tab.p, sizeof(double) * nreads) == 0;
    if (!same) {
        RIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ctx->share_tab.assign(tab.p, tab.p + nreads);
        RIP_HIP(ctx, hipMemcpy(d_tab, &tab, sizeof(ShareTable), hipMemcpyHostToDevice));
    }
    const size_t npix = (size_t)nya * nxa;
    // Poisson increments: the pixels with more than 4 electrons per read in a second, dense launch (apportion_kernel's note).  The
    // lists are sized for every pixel; the launches that draw them stride over what they hold
    uint32_t *defer = nullptr;
    double w_max = 0.0;
    if (poisson && npix < 0xFFFFFFFFull) {
        defer = (uint32_t *)rip_ws(ctx, 16, (DEFER_HEAD + 2 * DEFER_LISTS * defer_cap(npix)) * sizeof(uint32_t));
        if (!defer) return RIP_ENOMEM;
        RIP_HIP(ctx, hipMemsetAsync(defer, 0, DEFER_HEAD * sizeof(uint32_t), ctx->stream));
        for (int r = 0; r < nreads; ++r) w_max = tab.w[r] > w_max ? tab.w[r] : w_max;
    }
    const dim3 ag((unsigned)((npix + 255) / 256));
    if (!poisson)
        hipLaunchKernelGGL(apportion_kernel<0>, ag, dim3(256), 0, ctx->stream, counts, npix, nreads, (const double *)d_tab, seed, reads_e, defer, 4.0, w_max);
    else if (!defer)
        hipLaunchKernelGGL(apportion_kernel<1>, ag, dim3(256), 0, ctx->stream, counts, npix, nreads, (const double *)d_tab, seed, reads_e, defer, 4.0, w_max);
    else
        hipLaunchKernelGGL(apportion_kernel<2>, ag, dim3(256), 0, ctx->stream, counts, npix, nreads, (const double *)d_tab, seed, reads_e, defer, 4.0, w_max);
    if (defer) {
        hipLaunchKernelGGL(defer_scan_kernel, dim3(1), dim3(2 * DEFER_LISTS), 0, ctx->stream, defer);
        const dim3 dg((unsigned)std::min<size_t>((npix + 255) / 256, 4096));
        hipLaunchKernelGGL(apportion_deferred_kernel<0>, dg, dim3(256), 0, ctx->stream, counts, npix, nreads, (const double *)d_tab, seed, reads_e,
                           (const uint32_t *)defer);
        hipLaunchKernelGGL(apportion_deferred_kernel<1>, dg, dim3(256), 0, ctx->stream, counts, npix, nreads, (const double *)d_tab, seed, reads_e,
                           (const uint32_t *)defer);
    }
    RIP_HIP(ctx, hipGetLastError());
    return RIP_OK;
}

extern "C" int rip_synth_resultants(rip_ctx *ctx, const rip_synth_cal *cal, int ngrp, const int32_t *group_count, const int32_t *reads_e,
                                    const float *normals_reset, const float *normals_read, uint64_t seed, float *start_e,
                                    float *resultants, uint16_t *cube) {
    ctx->stream_dirty = true;
    int rc = check_cal(ctx, cal, ngrp, group_count, "synth_resultants");
    if (rc) return rc;
    if (!reads_e || !cal->gain || !cal->read_noise || !cal->resetnoise || !cal->lin_coefs || !cal->smin || !cal->smax ||
        (cal->biascorr && !cal->dark_slope))
        return rip_fail(ctx, RIP_EINVAL, "synth_resultants: missing calibration array");
    if (!resultants && !cube) return rip_fail(ctx, RIP_EINVAL, "synth_resultants: no output requested");
    RIP_HIP(ctx, hipSetDevice(ctx->device));
    const int nya = cal->ny - 2 * cal->nb, nxa = cal->nx - 2 * cal->nb;
    float *start = start_e;
    if (!start) {
        start = (float *)rip_ws(ctx, 11, (size_t)nya * nxa * sizeof(float));
        if (!start) return RIP_ENOMEM;
    }
    const dim3 ga((unsigned)((nxa + 255) / 256), (unsigned)nya), block(256);
    const bool g64 = cal->gain_dtype == RIP_F64, k64 = cal->ipc_dtype == RIP_F64;
    if (g64)
        hipLaunchKernelGGL(reset_kernel<double>, ga, block, 0, ctx->stream, normals_reset, cal->resetnoise, (const double *)cal->gain,
                           cal->dark_slope, cal->biascorr ? 1 : 0, (float)cal->tbias, cal->ny, cal->nx, cal->nb, seed, start);
    else
        hipLaunchKernelGGL(reset_kernel<float>, ga, block, 0, ctx->stream, normals_reset, cal->resetnoise, (const float *)cal->gain,
                           cal->dark_slope, cal->biascorr ? 1 : 0, (float)cal->tbias, cal->ny, cal->nx, cal->nb, seed, start);
    RIP_HIP(ctx, hipGetLastError());
    ResArgs a{};
    a.reads_e = reads_e;
    a.start = start;
    a.gain = cal->gain;
    a.kern = cal->ipc4d;
    a.coefs = cal->lin_coefs;
    a.smin = cal->smin;
    a.smax = cal->smax;
    a.read_noise = cal->read_noise;
    a.normals_read = normals_read;
    a.biascorr = cal->biascorr;
    a.resultants = resultants;
    a.cube = cube;
    a.ny = cal->ny;
    a.nx = cal->nx;
    a.nb = cal->nb;
    a.ngrp = ngrp;
    a.seed = seed;
    for (int j = 0; j < ngrp; ++j) {
        a.count[j] = group_count[j];
        a.root[j] = std::pow((double)group_count[j], 0.5);   // len(x) ** 0.5
    }
    if (cube && cal->nb > 0)
        hipLaunchKernelGGL(zero_border_kernel, dim3((unsigned)((2 * cal->nb * (cal->nx + cal->ny - 2 * cal->nb) + 255) / 256), (unsigned)ngrp), block, 0,
                           ctx->stream, cube, cal->ny, cal->nx, cal->nb);
    if (g64 && k64) return launch_resultants<double, double>(ctx, cal->nplanes, a);
    if (g64) return launch_resultants<double, float>(ctx, cal->nplanes, a);
    if (k64) return launch_resultants<float, double>(ctx, cal->nplanes, a);
    return launch_resultants<float, float>(ctx, cal->nplanes, a);
}

extern "C" int rip_synth_fill(rip_ctx *ctx, const rip_synth_cal *cal, int ngrp, const int32_t *group_count, int banding, const float *normals,
                              const float *frames, const float *white33, uint64_t seed, uint16_t *cube, uint16_t *amp33) {
    ctx->stream_dirty = true;
    int rc = check_cal(ctx, cal, ngrp, group_count, "synth_fill");
    if (rc) return rc;
    if (!cube || !cal->read_noise || !cal->resetnoise || !cal->dark) return rip_fail(ctx, RIP_EINVAL, "synth_fill: missing array");
    if (cal->channelwidth < 1 || cal->nx % cal->channelwidth)
        return rip_fail(ctx, RIP_EINVAL, "synth_fill: %d columns are not whole channels of %d", cal->nx, cal->channelwidth);
    const int nch = cal->nx / cal->channelwidth;
    const bool do33 = banding && amp33 && cal->amp33_valid;
    if (do33 && (!cal->amp33_med || !cal->amp33_std)) return rip_fail(ctx, RIP_EINVAL, "synth_fill: amp33 statistics missing");
    RIP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t fsz = (size_t)cal->ny * cal->channelwidth;
    if (banding && !frames) {
        float *made = (float *)rip_ws(ctx, 12, (size_t)ngrp * (nch + 2) * fsz * sizeof(float));
        if (!made) return RIP_ENOMEM;
        if (ctx->frames_pending && ctx->frames_seed == seed && ctx->frames_geom[0] == cal->ny && ctx->frames_geom[1] == cal->channelwidth &&
            ctx->frames_geom[2] == ngrp * (nch + 2)) {
            // made ahead on the second stream by rip_synth_frames_ahead with this seed: wait for them, nothing to compute
            RIP_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_frames, 0));
        } else {
            rc = rip_synth_noise_1f(ctx, cal->ny, cal->channelwidth, ngrp * (nch + 2), seed, 0x31660000u, made);
            if (rc) return rc;
        }
        ctx->frames_pending = false;
        frames = made;
    }
    FillArgs a{};
    a.normals = normals;
    a.frames = banding ? frames : nullptr;
    a.white33 = white33;
    a.read_noise = cal->read_noise;
    a.resetnoise = cal->resetnoise;
    a.dark = cal->dark;
    a.med = cal->amp33_med;
    a.std = cal->amp33_std;
    a.cube = cube;
    a.amp33 = amp33;
    a.ny = cal->ny;
    a.nx = cal->nx;
    a.nb = cal->nb;
    a.cw = cal->channelwidth;
    a.nch = nch;
    a.ngrp = ngrp;
    a.u_pink = (float)cal->u_pink;
    a.c_pink = (float)cal->c_pink;
    a.ru_pink = (float)cal->ru_pink;
    a.m_pink = (float)cal->m_pink;
    a.seed = seed;
    for (int j = 0; j < ngrp; ++j) a.root[j] = (float)std::pow((double)group_count[j], 0.5);
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((cal->nx + 255) / 256), (unsigned)cal->ny, (unsigned)ngrp), dim3(256), 0, ctx->stream, a);
    if (do33)
        hipLaunchKernelGGL(amp33_kernel, dim3((unsigned)((cal->channelwidth + 255) / 256), (unsigned)cal->ny, (unsigned)ngrp), dim3(256), 0,
                           ctx->stream, a);
    RIP_HIP(ctx, hipGetLastError());
    // frames made ahead for the NEXT exposure (second stream) must not overwrite the workspace before these kernels have read it
    if (ctx->stream2) {
        if (!ctx->ev_fill) RIP_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_fill, hipEventDisableTiming));
        RIP_HIP(ctx, hipEventRecord(ctx->ev_fill, ctx->stream));
        ctx->ev_fill_valid = true;
    }
    return RIP_OK;
}

extern "C" int rip_synth_extract_ref(rip_ctx *ctx, uint16_t *data, int ngrp, size_t n, int offset, uint16_t *reference_read) {
    ctx->stream_dirty = true;
    if (!data || ngrp < 1 || n < 1) return rip_fail(ctx, RIP_EINVAL, "synth_extract_ref: bad arguments");
    RIP_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(extract_ref_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, data, ngrp, n, offset, reference_read);
    RIP_HIP(ctx, hipGetLastError());
    return RIP_OK;
}
