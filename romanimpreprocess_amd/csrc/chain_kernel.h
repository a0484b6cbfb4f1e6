// The fused L1->L2 kernel (template; instantiated in chain_np*.hip): reference-pixel apply + bias +
// Legendre linearity + IPC deconvolution + ramp fit / jump detection / flag propagation + dark rate +
// error split + flat, one launch per ramp, every CALDIR and ramp array read from HBM once, no
// intermediate cube written.
//
// Replaces gen_cal_image.py:533-629 between the reference-pixel tables (refpix.hip) and the L2
// planes; per-stage arithmetic and its reference lines are those of linearity.hip, ipc.hip,
// rampfit.hip (the unfused kernels, kept as the general path and as stage-level drop-ins).
//
// Geometry.  The frame is cut into strips of CH_OUTW = CH_BT-4 output columns; a workgroup has one
// thread per column of a strip plus 2 halo columns on each side (5x5 IPC footprint) and marches
// down the rows.  Per group it keeps rolling 3-row windows of x = gain*phi (linearised data) and of
// the first Neumann iterate in LDS, so the vertical halo costs nothing and the horizontal halo is
// 4/CH_BT of the linearity work.  All strips' rows are laid end to end and divided EQUALLY among the
// workgroups of a grid that is exactly resident: no tail wave.  Step r of the march:
//     P: issue the global loads of row r+3 (raw data of all groups, linearity planes), of the IPC
//        coefficients of destination row r+2 and of what the fit of row r needs, into registers
//     C: first iterate O1 of row r+1 (x rows r..r+2, coefficients loaded one step earlier)   -> LDS
//     E: second iterate of row r (O1 rows r-1..r+1) / gain -> the pixel's ramp in registers,
//        ramp fit + finish of pixel (r, c), results to HBM
//     A: refpix/bias/linearity of row r+3 from the registers filled in P                      -> LDS
// with a barrier after C and after A: the HBM latency of P hides behind C and E.
// The group count and the Legendre order are template parameters (all group loops unrolled).
//
// Roofline: HBM.  Algorithmic bytes per pixel: G*(2 + 4 + 4 + 1) in, G out (groupdq),
// 4*(NP+3) + 4 linearity, 36 ipc4d, 4 gain, 4 read, 4 dark rate, 4 flat, 4 flags, 4 pdq in, 16 out.
#pragma once
#include "rip_common.h"

#include "device_rampfit.h"

#define CH_BT 256

// Diagnostic build (-DCH_STAMP): per-phase cycle sums of every wave go to ChainArgs::dbg_buf (6 x u64 per wave).
#ifdef CH_STAMP
#define CH_T(i)                                                          \
    {                                                                    \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();      \
        __builtin_amdgcn_s_waitcnt(0xC07F);                              \
        st_[i] += t_ - tl_;                                              \
        tl_ = t_;                                                        \
    }
#else
#define CH_T(i)
#endif
#define CH_OUTW (CH_BT - 4)

template <typename A, typename B>
struct ChPromote {
    using type = float;
};
template <>
struct ChPromote<float, double> {
    using type = double;
};

// IEEE a/b from the correctly rounded reciprocal rb = 1.0f/b: two Newton corrections with exact (fma)
// residuals, i.e. the tail of the hardware division macro.  Bit-identical to a/b for finite normal
// operands and quotients (tools/gpu_checks/divcheck.hip: 1e11 pairs, 0 mismatches); callers fall back
// to the division operator when b is zero / subnormal / huge.
__device__ __forceinline__ float div_rcp(float a, float b, float rb) {
    const float q0 = a * rb;
    const float r0 = fmaf(-b, q0, a);
    const float q1 = fmaf(r0, rb, q0);
    const float r1 = fmaf(-b, q1, a);
    return fmaf(r1, rb, q1);
}
__device__ __forceinline__ bool rcp_safe(float b) {
    const float ab = fabsf(b);
    return ab > 1e-18f && ab < 1e18f;
}

// forward IPC operator at column `t` of three LDS rows (rm = row y-1, r0 = row y, rp = row y+1);
// term order and edge rule of ipc_linearity.py:69-94 (see ipc.hip).  ALL = every source is active
// (interior pixel): no per-term selects.
template <typename T, typename ST, typename KT, bool ALL>
__device__ __forceinline__ T fwd_rows(const ST *rm, const ST *r0, const ST *rp, int t, const KT (&kk)[9], unsigned valid) {
    T acc = (T)r0[t] * (T)kk[0];
    T p;
#define CH_TERM(k, src)      \
    p = (T)(src) * (T)kk[k]; \
    acc = (ALL || ((valid >> k) & 1u)) ? acc + p : acc;
    CH_TERM(1, rm[t])
    CH_TERM(2, rp[t])
    CH_TERM(3, r0[t - 1])
    CH_TERM(4, r0[t + 1])
    CH_TERM(5, rm[t - 1])
    CH_TERM(6, rm[t + 1])
    CH_TERM(7, rp[t - 1])
    CH_TERM(8, rp[t + 1])
#undef CH_TERM
    return acc;
}

template <typename KT>
__device__ __forceinline__ unsigned ch_load_coeffs(const KT *__restrict__ kern, unsigned plane, int nx, int y, int x, int y0,
                                                   int y1, int x0, int x1, KT (&kk)[9]) {
    unsigned valid = 0;
    const bool dest_ok = (y >= y0 && y < y1 && x >= x0 && x < x1);
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int dy = (k == 1 || k == 5 || k == 6) ? 1 : (k == 2 || k == 7 || k == 8) ? -1 : 0;
        const int dx = (k == 3 || k == 5 || k == 7) ? 1 : (k == 4 || k == 6 || k == 8) ? -1 : 0;
        const int sy = y - dy, sx = x - dx;
        const bool ok = dest_ok && sy >= y0 && sy < y1 && sx >= x0 && sx < x1;
        const unsigned off = (unsigned)(3 * (1 + dy) + (1 + dx)) * plane + (unsigned)(sy * nx + sx);
        kk[k] = ok ? kern[off] : (KT)0;
        valid |= ok ? (1u << k) : 0u;
    }
    return valid;
}

// load through a uniform base pointer + 32-bit per-lane BYTE offset (global_load ... v_off, s[base] form: no 64-bit
// address arithmetic per load); every array addressed this way is smaller than 4 GiB
template <typename T>
__device__ __forceinline__ T ldg(const void *base, unsigned byte_off) {
    return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + byte_off);
}

// raw bits of one row position, fetched one phase ahead of their use (no arithmetic on them in P)
template <int NP, int G>
struct RowRegs {
    uint32_t S[G];
    uint32_t q[G];
    float dk[G], bs[G];
    float cf[NP], smin, smax, sref, gain;
    uint32_t dq;
};

template <int NP, int G, typename KT>
__global__ __launch_bounds__(CH_BT) void chain_kernel(ChainArgs a, const RipPlanHeader *__restrict__ h,
                                                      const RipVariant *__restrict__ vars,
                                                      const float *__restrict__ kvals,
                                                      const RipDiff *__restrict__ diffs, double guard) {
    using T = typename ChPromote<float, KT>::type;  // gain is f32 on this path: x is f32, iterates are T
    extern __shared__ __align__(16) unsigned char lds_raw[];
    T *O1 = reinterpret_cast<T *>(lds_raw);                    // [G][3][CH_BT]
    float *X = reinterpret_cast<float *>(O1 + G * 3 * CH_BT);  // [G][3][CH_BT]
    float *R = X + G * 3 * CH_BT;                              // [G][CH_BT]  ramp (saturated pixels' refits only)
    uint8_t *Q = reinterpret_cast<uint8_t *>(R + G * CH_BT);   // [G][CH_BT]
    uint8_t *J = Q + G * CH_BT;                                // [G][CH_BT]
    double *LN = reinterpret_cast<double *>(J + G * CH_BT);    // [3][G][2]   channel lines of this strip
    double *RC = LN + 3 * G * 2;                               // [a.rc_rows][G] row corrections of this row range

    const int tid = threadIdx.x;
    const int ny = a.ny, nx = a.nx, nb = a.nb;
    const int ay0 = nb, ay1 = ny - nb, ax0 = nb, ax1 = nx - nb;
    const unsigned npix = (unsigned)ny * (unsigned)nx;
    const unsigned pl4 = npix * 4u;  // bytes per f32 plane
    const KT *__restrict__ kern = reinterpret_cast<const KT *>(a.kern);
    const int nch = nx / RIP_CW;
    const uint32_t bad = DQ_NO_LIN_CORR | DQ_REFERENCE_PIXEL;
    const float *__restrict__ planes = a.planes;
    const uint32_t *__restrict__ planes_u = reinterpret_cast<const uint32_t *>(a.planes);
    const uint16_t *__restrict__ d16 = reinterpret_cast<const uint16_t *>(a.data);
    const uint8_t *__restrict__ gdq = a.gdq;
    const float *__restrict__ dark = a.dark_data;
    const float *__restrict__ bias = a.bias;

    // recurrence constants of the Legendre series (linearity.hip); folded at compile time
    float c1[NP], c2[NP], chf[NP];
#pragma unroll
    for (int L = 1; L < NP; ++L) {
        c1[L] = (float)((double)(2 * L + 1) / (double)(L + 1));
        c2[L] = (float)((double)L / (double)(L + 1));
        chf[L] = (float)((double)(L * (L + 1)) / 2.0);
    }

    // equal share of (strip, row) work: global row index = strip * ny + row
    const int nstrips = (nx + CH_OUTW - 1) / CH_OUTW;
    const long total_rows = (long)nstrips * ny;
    const long per = (total_rows + gridDim.x - 1) / gridDim.x;
    long cur = (long)blockIdx.x * per;
    const long end = min(total_rows, cur + per);

    while (cur < end) {
        const int strip = (int)(cur / ny);
        const int R0 = (int)(cur % ny);
        const int R1 = (int)min((long)ny, (long)R0 + (end - cur));  // rows [R0, R1) of this strip
        cur += R1 - R0;
        const int c = strip * CH_OUTW - 2 + tid;  // column of this thread
        const bool col_ok = (c >= 0 && c < nx);
        const bool col_act = (c >= ax0 && c < ax1);
        const int cc = col_ok ? c : 0;  // clamped: out-of-frame lanes load valid addresses and discard
        const int ch0 = max(strip * CH_OUTW - 2, 0) / RIP_CW;  // first channel this strip touches
        const int chr = cc / RIP_CW - ch0;                     // 0..2
        // stage the reference-pixel tables of this (strip, row range): lines[g][ch0..ch0+2], rowcorr[g][R0-2..R1+1]
        __syncthreads();
        for (int i = tid; i < 3 * G * 2; i += CH_BT) {
            const int ch = i / (G * 2), g = (i / 2) % G, w = i & 1;
            LN[i] = (ch0 + ch < nch) ? a.lines[(g * nch + ch0 + ch) * 2 + w] : 0.0;
        }
        for (int i = tid; i < (R1 - R0 + 4) * G; i += CH_BT) {
            const int y = R0 - 2 + i / G, g = i % G;
            RC[i] = (y >= 0 && y < ny) ? a.rowcorr[g * ny + y] : 0.0;
        }
        __syncthreads();

        KT kA[9], kB[9], kC[9];
        unsigned vA = 0, vB = 0, vC = 0;
#pragma unroll
        for (int k = 0; k < 9; ++k) kA[k] = kB[k] = kC[k] = (KT)0;
        uint32_t d0 = 0, d1 = 0, d2 = 0;

#ifdef CH_STAMP
        unsigned long long st_[6] = {0, 0, 0, 0, 0, 0};
        unsigned long long tl_ = __builtin_amdgcn_s_memtime();
#endif
        // march: step r ingests row r+3, forms O1 of row r+1 and finishes row r.  The three steps before
        // R0-2 only ingest (rows R0-2, R0-1, R0).
        int s0 = 0;  // LDS slot of row r; rows r+1, r+2 (and r-1 for O1) follow cyclically
        for (int r = R0 - 5; r < R1; ++r) {
            const int s1 = (s0 == 2) ? 0 : s0 + 1, s2 = (s1 == 2) ? 0 : s1 + 1;
            const int yi = r + 3;  // row ingested in this step
            const bool do_ingest = (yi >= R0 - 2) && (yi <= R1 + 1);
            const bool row_in = do_ingest && yi >= 0 && yi < ny;
            const bool do_c = (r + 1 >= R0 - 1) && (r + 1 <= R1);
            const bool do_e = (r >= R0);

            // ---- P: issue loads (row yi raw inputs; IPC coefficients of row r+2; fit inputs of row r).
            // Straight-line: addresses are clamped into the frame and unused results are discarded later, so
            // that no branch separates the loads from the code that runs while they are in flight.
            const bool emit = do_e && tid >= 2 && tid < CH_BT - 2 && col_ok;
            const unsigned pe = (unsigned)(min(max(r, 0), ny - 1) * nx + cc);
            uint32_t qe[G];
#pragma unroll
            for (int g = 0; g < G; ++g) qe[g] = ldg<uint8_t>(gdq, (unsigned)g * npix + pe);
            const float e_gain = ldg<float>(planes, (unsigned)(NP + 4) * pl4 + pe * 4u);
            const float e_read = ldg<float>(planes, (unsigned)(NP + 5) * pl4 + pe * 4u);
            const float e_dark = ldg<float>(planes, (unsigned)(NP + 6) * pl4 + pe * 4u);
            const uint32_t e_ff = ldg<uint32_t>(planes, (unsigned)(NP + 8) * pl4 + pe * 4u);
            const uint32_t e_pdq = ldg<uint32_t>(a.pdq, pe * 4u);
            const float e_flat = a.flat ? ldg<float>(a.flat, pe * 4u) : 1.0f;

            {
                // coefficients of destination (r+2, c): raw loads at clamped source positions + validity mask
                const int y2 = r + 2;
                const bool want = tid >= 1 && tid < CH_BT - 1 && y2 >= R0 - 1 && y2 <= R1;
                const bool dest_ok = want && (y2 >= ay0 && y2 < ay1 && c >= ax0 && c < ax1);
                vC = 0;
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const int dy = (k == 1 || k == 5 || k == 6) ? 1 : (k == 2 || k == 7 || k == 8) ? -1 : 0;
                    const int dx = (k == 3 || k == 5 || k == 7) ? 1 : (k == 4 || k == 6 || k == 8) ? -1 : 0;
                    const int sy = y2 - dy, sx = c - dx;
                    const bool ok = dest_ok && sy >= ay0 && sy < ay1 && sx >= ax0 && sx < ax1;
                    const int syc = min(max(sy, 0), ny - 1), sxc = min(max(sx, 0), nx - 1);
                    kC[k] = ldg<KT>(kern, ((unsigned)(3 * (1 + dy) + (1 + dx)) * npix + (unsigned)(syc * nx + sxc)) * (unsigned)sizeof(KT));
                    vC |= ok ? (1u << k) : 0u;
                }
            }
            RowRegs<NP, G> rr;
            {
                const int yl = (a.dbg & 128) ? 0 : min(max(yi, 0), ny - 1);  // dbg 128: re-read row 0 (cache hits)
                const unsigned p = (unsigned)(yl * nx + cc);
                const unsigned p4 = p * 4u;
#pragma unroll
                for (int L = 0; L < NP; ++L) rr.cf[L] = ldg<float>(planes, (unsigned)L * pl4 + p4);
                rr.smin = ldg<float>(planes, (unsigned)(NP + 0) * pl4 + p4);
                rr.smax = ldg<float>(planes, (unsigned)(NP + 1) * pl4 + p4);
                rr.sref = ldg<float>(planes, (unsigned)(NP + 2) * pl4 + p4);
                rr.dq = ldg<uint32_t>(planes, (unsigned)(NP + 3) * pl4 + p4);
                rr.gain = ldg<float>(planes, (unsigned)(NP + 4) * pl4 + p4);
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    rr.S[g] = ldg<uint16_t>(d16, (unsigned)g * (pl4 >> 1) + (p4 >> 1));
                    rr.q[g] = ldg<uint8_t>(gdq, (unsigned)g * npix + p);
                    rr.dk[g] = ldg<float>(dark, (unsigned)g * pl4 + p4);
                    rr.bs[g] = ldg<float>(bias, (unsigned)g * pl4 + p4);
                }
            }

            CH_T(0)
            // ---- C: O1 of row r+1
            if (do_c && vB && !(a.dbg & 1)) {
                const bool all = __all(vB == 0x1ffu);  // wave-uniform: every active lane is an interior pixel
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const float *xb = X + g * 3 * CH_BT;
                    const float *xm = xb + s0 * CH_BT, *x0 = xb + s1 * CH_BT, *xp = xb + s2 * CH_BT;
                    const T f = all ? fwd_rows<T, float, KT, true>(xm, x0, xp, tid, kB, vB)
                                    : fwd_rows<T, float, KT, false>(xm, x0, xp, tid, kB, vB);
                    const float xc = x0[tid];
                    O1[(g * 3 + s1) * CH_BT + tid] = (T)(xc + xc) - f;
                }
            }
            CH_T(1)
            if (!(a.dbg & 64)) __syncthreads();
            CH_T(2)

            // ---- E: O2 of row r, ramp fit, outputs
            if (emit) {
                const bool act = col_act && r >= ay0 && r < ay1;
                const bool fastdiv = __all(rcp_safe(e_gain) || !act);  // wave-uniform
                const float rgain = 1.0f / e_gain;
                const bool all = __all(vA == 0x1ffu || !act);  // wave-uniform
                float d[G];
                uint32_t anyq = 0;
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const float xc = X[(g * 3 + s0) * CH_BT + tid];
                    float val = xc;
                    if (act && !(a.dbg & 2)) {
                        const T *ob = O1 + g * 3 * CH_BT;
                        const T *om = ob + s2 * CH_BT, *o0 = ob + s0 * CH_BT, *op = ob + s1 * CH_BT;
                        const T f = all ? fwd_rows<T, T, KT, true>(om, o0, op, tid, kA, vA)
                                        : fwd_rows<T, T, KT, false>(om, o0, op, tid, kA, vA);
                        const T o2 = (o0[tid] + (T)xc) - f;
                        if constexpr (sizeof(T) == 4) {
                            if (fastdiv)
                                val = div_rcp(o2, e_gain, rgain);
                            else
                                val = o2 / e_gain;
                        } else
                            val = (float)(o2 / (T)e_gain);
                    }
                    d[g] = val;
                    anyq |= qe[g];
                }
                if (a.dbg & 8) anyq = 0;
                if (a.cube_out) {
#pragma unroll
                    for (int g = 0; g < G; ++g) a.cube_out[(unsigned)g * npix + pe] = d[g];
                }
                float s, er, ep;
                uint32_t pdq;
                const uint32_t pdq_in = e_pdq | d0;
                if (anyq & DQ_SATURATED) {
                    // some group is saturated: general path with truncated refits (ramp staged in LDS)
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        R[g * CH_BT + tid] = d[g];
                        Q[g * CH_BT + tid] = qe[g];
                        J[g * CH_BT + tid] = 0;
                    }
                    rampfit_pixel<float, CH_BT>(R + tid, Q + tid, J + tid, G, h, vars, kvals, diffs, e_gain, e_read, act,
                                                guard, pdq_in, a.gdq_out ? a.gdq_out + pe : nullptr, npix, s, er, ep, pdq);
                } else {
                    uint32_t jmask = 0;
                    if (a.dbg & 4) {
                        s = d[G - 1] - d[1];
                        er = e_read;
                        ep = e_gain;
                    } else
                        fit_full_regs<G>(d, h, vars[0], kvals + vars[0].k_ofs, diffs + vars[0].diff_ofs, e_gain, e_read, act,
                                         guard, s, er, ep, jmask);
                    // flag propagation (fitting.py:339-353) without saturation
                    uint32_t orq = 0;
                    bool all_dnu = true;
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        const uint32_t rq = (uint32_t)qe[g] | (((jmask >> g) & 1u) ? DQ_JUMP_DET : 0u);
                        if (a.gdq_out) a.gdq_out[(unsigned)g * npix + pe] = (uint8_t)rq;
                        orq |= rq;
                        all_dnu = all_dnu && ((rq & DQ_DO_NOT_USE) != 0);
                    }
                    uint32_t pdq2 = orq & ~DQ_DO_NOT_USE;
                    if (all_dnu) pdq2 |= DQ_DO_NOT_USE;
                    pdq = (pdq_in & DQ_REFERENCE_PIXEL) ? pdq_in : (pdq_in | pdq2);
                }
                if (a.finish) {
                    // gen_cal_image.py:458-475, 213-229, 607-629 (see finish_pixel)
                    float err = hypot_f32(er, ep);
                    float vp = ep * ep;
                    if (!act) {
                        s = 0.0f;
                        err = 0.0f;
                        vp = 0.0f;
                    }
                    if (act && a.dark_rate) s = s - e_dark;
                    if (act && a.dark_dq) pdq |= a.dark_dq[pe];
                    float ep2 = sqrtf(vp);
                    const float e2 = err * err;
                    const float p2 = ep2 * ep2;
                    float er2 = sqrtf(clip_lo<float>(e2 - p2, 0.0f));
                    if (a.flat) {
                        pdq |= e_ff;
                        s = s / e_flat;
                        er2 = er2 / e_flat;
                        ep2 = ep2 / e_flat;
                    }
                    er = er2;
                    ep = ep2;
                }
                *reinterpret_cast<float *>(reinterpret_cast<char *>(a.slope) + pe * 4u) = s;
                *reinterpret_cast<float *>(reinterpret_cast<char *>(a.err_read) + pe * 4u) = er;
                *reinterpret_cast<float *>(reinterpret_cast<char *>(a.err_poisson) + pe * 4u) = ep;
                *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(a.pdq_out) + pe * 4u) = pdq;
            }

            CH_T(3)
            // ---- A: refpix apply + bias + linearity of row yi -> X slot of row r (its x was last read above,
            //         by this thread only); lin dq of the row enters the d-pipeline
            uint32_t d3 = 0;
            if (do_ingest) {
                float *xs = X + s0 * CH_BT + tid;
                if (!(row_in && col_ok)) {
#pragma unroll
                    for (int g = 0; g < G; ++g) xs[g * 3 * CH_BT] = 0.0f;
                } else {
                    const bool act = col_act && yi >= ay0 && yi < ay1;
                    const float smin = rr.smin;
                    const float span = rr.smax - smin;
                    const bool fastdiv = __all(rcp_safe(span));  // wave-uniform
                    const float rspan = 1.0f / span;
                    uint32_t dq = rr.dq;
                    const double yd = (double)yi;
                    bool any_ex = false;
                    float zz[G], SS[G];
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        // reference_subtraction.py:123 and :67-68 in f64, cast back to f32 after each step
                        float S = (float)rr.S[g];
                        const float dk = rr.dk[g];
                        float v = S - dk;
                        v = (float)((double)v - RC[(yi - (R0 - 2)) * G + g]);
                        const double *ln = LN + (chr * G + g) * 2;
                        const double iel = ln[0] * yd + ln[1];
                        v = (float)((double)v - iel);
                        S = v + dk;
                        if (act) S = S - rr.bs[g];
                        float t = S - smin;
                        t = 2.0f * t;
                        float quo;
                        if (fastdiv)
                            quo = div_rcp(t, span, rspan);
                        else
                            quo = t / span;
                        float z = -1.0f + quo;
                        if (g == 0 && a.do_not_flag_first) z = clip2<float>(z, -1.0f, 1.0f);
                        zz[g] = z;
                        SS[g] = S;
                        any_ex = any_ex || (fabsf(z) > 1.0f);
                    }
                    const bool slow = __any(any_ex) && !(a.dbg & 32);  // some sample extrapolates: series with the linear branch
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        const float z = zz[g];
                        float phi = rr.cf[0];
                        float pp = 1.0f, pc = z;
                        bool ex = false;
                        if (a.dbg & 16) {
                            phi = phi + z;
                        } else if (slow) {
                            const float az = fabsf(z);
                            ex = az > 1.0f;
                            const float exc = az - 1.0f;
                            const bool neg = z < 0.0f;
#pragma unroll
                            for (int L = 1; L < NP; ++L) {
                                float e = 1.0f + chf[L] * exc;
                                e = (neg && (L & 1)) ? -e : e;
                                const float sel = ex ? e : pc;
                                const float term = rr.cf[L] * sel;
                                phi = phi + term;
                                const float u = c1[L] * z;
                                const float pn = u * pc - c2[L] * pp;
                                pp = pc;
                                pc = pn;
                            }
                        } else {
#pragma unroll
                            for (int L = 1; L < NP; ++L) {
                                const float term = rr.cf[L] * pc;
                                phi = phi + term;
                                const float u = c1[L] * z;
                                const float pn = u * pc - c2[L] * pp;
                                pp = pc;
                                pc = pn;
                            }
                        }
                        const float val = ((dq & bad) == 0) ? phi : (SS[g] - rr.sref);
                        const bool first = (g == 0) && a.do_not_flag_first;
                        if (!first && ex && (rr.q[g] & DQ_SATURATED) == 0) dq |= DQ_NO_LIN_CORR;
                        // active pixels enter the IPC stage as gain*phi; border pixels keep phi (no IPC there)
                        xs[g * 3 * CH_BT] = act ? val * rr.gain : val;
                    }
                    d3 = dq;
                }
            }
            CH_T(4)
            if (!(a.dbg & 64)) __syncthreads();
            CH_T(5)
            d0 = d1;
            d1 = d2;
            d2 = d3;
            vA = vB;
            vB = vC;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                kA[k] = kB[k];
                kB[k] = kC[k];
            }
            s0 = s1;
        }
#ifdef CH_STAMP
        if ((tid & 63) == 0 && a.dbg_buf) {
            unsigned long long *o = a.dbg_buf + ((size_t)blockIdx.x * (CH_BT / 64) + (tid >> 6)) * 6;
            for (int i = 0; i < 6; ++i) o[i] += st_[i];
        }
#endif
    }
}

static inline size_t chain_lds_bytes(int G, int k_dtype) {
    const size_t t = k_dtype == RIP_F64 ? 8 : 4;
    return (size_t)G * CH_BT * (3 * t + 3 * 4 + 4 + 2);
}
static inline size_t chain_lds_tables(int G, int rc_rows) { return (size_t)(3 * G * 2 + rc_rows * G) * 8; }

template <int NP, int G, typename KT>
static int launch_chain(rip_ctx *ctx, const RipPlan *plan, const ChainArgs &a) {
    extern double rip_guard_band;
    const size_t lds0 = chain_lds_bytes(G, sizeof(KT) == 8 ? RIP_F64 : RIP_F32);
    static int ncu = 0;
    if (!ncu) {
        hipDeviceProp_t prop;
        RIP_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
        ncu = prop.multiProcessorCount;
    }
    const int per_cu = (int)((150 * 1024) / lds0) < 1 ? 1 : (int)((150 * 1024) / lds0);
    const int nstrips = (a.nx + CH_OUTW - 1) / CH_OUTW;
    long grid = (long)ncu * (per_cu > 4 ? 4 : per_cu);
    const long total_rows = (long)nstrips * a.ny;
    if (grid > (total_rows + 7) / 8) grid = (total_rows + 7) / 8;  // small frames: at least ~8 rows per workgroup
    if (grid < 1) grid = 1;
    const long per = (total_rows + grid - 1) / grid;  // rows per workgroup (a range never spans more than `per` rows)
    const size_t lds = lds0 + chain_lds_tables(G, (int)per + 4);
    if (lds > 160 * 1024) return rip_fail(ctx, RIP_EINVAL, "fused chain: LDS budget exceeded (%zu bytes)", lds);
    if (lds > 48 * 1024)
        RIP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(chain_kernel<NP, G, KT>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((chain_kernel<NP, G, KT>), dim3((unsigned)grid), dim3(CH_BT), lds, ctx->stream, a,
                       reinterpret_cast<const RipPlanHeader *>(plan->dev), plan->d_variants, plan->d_k, plan->d_diffs,
                       rip_guard_band);
    RIP_HIP(ctx, hipGetLastError());
    return RIP_OK;
}

template <int NP>
static int launch_chain_np(rip_ctx *ctx, const RipPlan *plan, const ChainArgs &a, int k_dtype) {
    const bool k64 = k_dtype == RIP_F64;
    switch (a.ngrp) {
        case 6:
            return k64 ? launch_chain<NP, 6, double>(ctx, plan, a) : launch_chain<NP, 6, float>(ctx, plan, a);
        case 8:
            return k64 ? launch_chain<NP, 8, double>(ctx, plan, a) : launch_chain<NP, 8, float>(ctx, plan, a);
        case 16:
            if (!k64) return launch_chain<NP, 16, float>(ctx, plan, a);
    }
    return rip_fail(ctx, RIP_EINVAL, "fused chain: %d groups (%s ipc4d) not instantiated", a.ngrp, k64 ? "f64" : "f32");
}
