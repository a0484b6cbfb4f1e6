// Fused L1->L2 kernel, wave-specialised form (f32 ipc4d, f32 gain, G <= 8): the arithmetic, the strip geometry and
// the packed-pair forms are those of chain_kernel.h; what changes is WHO does what.
//
// chain_kernel.h is bound by instruction issue at 2 waves/SIMD (one wave carries the registers of the linearity
// prefetch AND of the IPC/fit state, and LDS holds only two 256-column workgroups per CU).  Here a workgroup of
// 512 threads covers the same 256 columns with TWO ROLES of four waves each:
//     ingest waves (tid < 256)   P: raw loads of row r+4            A: refpix/bias/linearity of row r+3 -> x ring
//                                C: first IPC iterate of row r+2    (reads x rows r+1..r+3)          -> O1 ring
//     fit waves    (tid >= 256)  O2: second iterate of row r / gain (reads O1 rows r-1..r+1, x row r)
//                                F: ramp fit, flags, finish, stores of pixel (r, c)
// so each role needs <= 128 VGPRs and a CU holds 2 workgroups = 16 waves = 4 waves/SIMD.  Per step:
//     S1: ingest A(r+3)            | fit O2(r)            -- barrier --
//     S2: ingest C(r+2), P(r+4)    | fit F(r), loads of row r+1
// The x ring is 4 rows deep (rows r..r+3 are live during a step), the O1 ring 3 rows (C writes row r+2 into the
// slot of row r-1, which O2(r) finished reading before the barrier).  What the fit waves need from the ingest of
// the same pixel three steps earlier (linearity dq, the 8 groupdq bytes, gain) travels through small LDS rings.
// Saturated pixels are refitted from registers (trunc_layers), so no per-pixel ramp staging in LDS.
#pragma once
#include "chain_kernel.h"

#define C2_COLS 256
#define C2_THREADS 512
#define C2_OUTW (C2_COLS - 4)

template <int NP, int G>
__global__ __launch_bounds__(C2_THREADS, 4) void chain2_kernel(ChainArgs a, const RipPlanHeader *__restrict__ h,
                                                               const RipVariant *__restrict__ vars,
                                                               const float *__restrict__ kvals,
                                                               const RipDiff *__restrict__ diffs, double guard) {
    static_assert(G % 2 == 0 && G <= 8, "pairs of groups, at most 8 (groupdq bytes travel as one 64-bit word)");
    constexpr int GP = G / 2;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    f2 *X2 = reinterpret_cast<f2 *>(lds_raw);                       // [GP][4][C2_COLS]  x = gain*phi, pair-interleaved
    f2 *O12 = X2 + GP * 4 * C2_COLS;                                // [GP][3][C2_COLS]  first Neumann iterate
    uint32_t *DQ = reinterpret_cast<uint32_t *>(O12 + GP * 3 * C2_COLS);  // [4][C2_COLS] linearity dq of the row
    uint2 *QS = reinterpret_cast<uint2 *>(DQ + 4 * C2_COLS);       // [4][C2_COLS] groupdq bytes of the pixel
    float *GS = reinterpret_cast<float *>(QS + 4 * C2_COLS);        // [4][C2_COLS] gain
    double *LN = reinterpret_cast<double *>(GS + 4 * C2_COLS);      // [3][G][2] channel lines of this strip

    const int tid = threadIdx.x;
    const bool fit_role = tid >= C2_COLS;
    const int col = tid & (C2_COLS - 1);
    const int ny = a.ny, nx = a.nx, nb = a.nb;
    const int ay0 = nb, ay1 = ny - nb, ax0 = nb, ax1 = nx - nb;
    const unsigned npix = (unsigned)ny * (unsigned)nx;
    const unsigned pl4 = npix * 4u;
    const float *__restrict__ kern = reinterpret_cast<const float *>(a.kern);
    const int nch = nx / RIP_CW;
    const uint32_t bad = DQ_NO_LIN_CORR | DQ_REFERENCE_PIXEL;
    const float *__restrict__ planes = a.planes;
    const uint16_t *__restrict__ d16 = reinterpret_cast<const uint16_t *>(a.data);
    const uint8_t *__restrict__ gdq = a.gdq;
    const float *__restrict__ dark = a.dark_data;
    const float *__restrict__ bias = a.bias;

    float c1[NP], c2[NP], chf[NP];
#pragma unroll
    for (int L = 1; L < NP; ++L) {
        c1[L] = (float)((double)(2 * L + 1) / (double)(L + 1));
        c2[L] = (float)((double)L / (double)(L + 1));
        chf[L] = (float)((double)(L * (L + 1)) / 2.0);
    }

    const int nstrips = (nx + C2_OUTW - 1) / C2_OUTW;
    const int nranges = gridDim.x / nstrips;
    const int rows_per = (ny + nranges - 1) / nranges;
    const int strip = (int)blockIdx.x % nstrips;
    const int R0 = ((int)blockIdx.x / nstrips) * rows_per;
    const int R1 = min(ny, R0 + rows_per);
    if ((int)blockIdx.x >= nstrips * nranges || R0 >= ny) return;
    const int c = strip * C2_OUTW - 2 + col;
    const bool col_ok = (c >= 0 && c < nx);
    const bool col_act = (c >= ax0 && c < ax1);
    const int cc = col_ok ? c : 0;
    const int ch0 = max(strip * C2_OUTW - 2, 0) / RIP_CW;
    const int chr = cc / RIP_CW - ch0;
    for (int i = tid; i < 3 * G * 2; i += C2_THREADS) {
        const int ch = i / (G * 2), g = (i / 2) % G, w = i & 1;
        LN[i] = (ch0 + ch < nch) ? a.lines[(g * nch + ch0 + ch) * 2 + w] : 0.0;
    }
    __syncthreads();

    // coefficient loader: raw loads at clamped source positions + validity mask for destination (y, c)
    auto load_k = [&](int y, bool want, float (&kk)[9]) -> unsigned {
        const bool dest_ok = want && (y >= ay0 && y < ay1 && c >= ax0 && c < ax1);
        unsigned valid = 0;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int dy = (k == 1 || k == 5 || k == 6) ? 1 : (k == 2 || k == 7 || k == 8) ? -1 : 0;
            const int dx = (k == 3 || k == 5 || k == 7) ? 1 : (k == 4 || k == 6 || k == 8) ? -1 : 0;
            const int sy = y - dy, sx = c - dx;
            const bool ok = dest_ok && sy >= ay0 && sy < ay1 && sx >= ax0 && sx < ax1;
            const int syc = min(max(sy, 0), ny - 1), sxc = min(max(sx, 0), nx - 1);
            kk[k] = ldg<float>(kern, (unsigned)(3 * (1 + dy) + (1 + dx)) * pl4 + (unsigned)(syc * nx + sxc) * 4u);
            valid |= ok ? (1u << k) : 0u;
        }
        return valid;
    };

#ifdef CH_STAMP
    unsigned long long st_[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tl_ = __builtin_amdgcn_s_memtime();
#endif
    if (!fit_role) {
        // =========================================================================== ingest waves
        auto fetch_row = [&](int y, RowRegs<NP, G> &rr) {
            const int yl = min(max(y, 0), ny - 1);
            const unsigned p = (unsigned)(yl * nx + cc);
            // running byte offsets (one v_add per load, one SGPR stride) instead of per-plane constants: keeps the
            // scalar register file free of ~40 loop-invariant offsets
            unsigned o4 = p * 4u;
#pragma unroll
            for (int L = 0; L < NP; ++L) {
                rr.cf[L] = ldg<float>(planes, o4);
                o4 += pl4;
            }
            rr.smin = ldg<float>(planes, o4);
            o4 += pl4;
            rr.smax = ldg<float>(planes, o4);
            o4 += pl4;
            rr.sref = ldg<float>(planes, o4);
            o4 += pl4;
            rr.dq = ldg<uint32_t>(planes, o4);
            o4 += pl4;
            rr.gain = ldg<float>(planes, o4);
            unsigned g4 = p * 4u, g2 = p * 2u, g1 = p;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                rr.S[g] = ldg<uint16_t>(d16, g2);
                rr.q[g] = ldg<uint8_t>(gdq, g1);
                rr.dk[g] = ldg<float>(dark, g4);
                rr.bs[g] = ldg<float>(bias, g4);
                g4 += pl4;
                g2 += pl4 >> 1;
                g1 += npix;
            }
        };
        RowRegs<NP, G> rr;
        fetch_row(R0 - 2, rr);
        for (int r = R0 - 5; r < R1; ++r) {
            const int yi = r + 3, yc = r + 2;
            const bool do_a = (yi >= R0 - 2) && (yi <= R1 + 1);
            const bool do_c = (yc >= R0 - 1) && (yc <= R1);
            // ---- S1: coefficient loads for C, then A (linearity of row yi from rr)
            // per-row reference-pixel correction of the G groups: wave-uniform, scalar loads (constant address space)
            double rc[G];
            {
                const int yl = min(max(yi, 0), ny - 1);
#pragma unroll
                for (int g = 0; g < G; ++g) rc[g] = KLD(a.rowcorr[g * ny + yl]);
            }
            float kC[9];
            const unsigned vC = load_k(yc, do_c && col >= 1 && col < C2_COLS - 1, kC);
            if (do_a) {
                const int slot = yi & 3;
                f2 *xs = X2 + slot * C2_COLS + col;
                const bool row_in = yi >= 0 && yi < ny;
                if (!(row_in && col_ok)) {
#pragma unroll
                    for (int p = 0; p < GP; ++p) xs[p * 4 * C2_COLS] = f2{0.0f, 0.0f};
                    DQ[slot * C2_COLS + col] = 0u;
                    QS[slot * C2_COLS + col] = uint2{0u, 0u};
                    GS[slot * C2_COLS + col] = 1.0f;
                } else {
                    const bool act = col_act && yi >= ay0 && yi < ay1;
                    const float smin = rr.smin;
                    const float span = rr.smax - smin;
                    const bool fastdiv = __all(rcp_safe(span));
                    const float rspan = 1.0f / span;
                    uint32_t dq = rr.dq;
                    const double yd = (double)yi;
                    bool any_ex = false;
                    f2 zz[GP], SS[GP];
#pragma unroll
                    for (int p = 0; p < GP; ++p) {
                        float Sv[2];
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const int g = 2 * p + e;
                            float S = (float)rr.S[g];
                            const float dk = rr.dk[g];
                            float v = S - dk;
                            v = (float)((double)v - rc[g]);
                            const double *ln = LN + (chr * G + g) * 2;
                            const double iel = ln[0] * yd + ln[1];
                            v = (float)((double)v - iel);
                            S = v + dk;
                            if (act) S = S - rr.bs[g];
                            Sv[e] = S;
                        }
                        const f2 S2 = {Sv[0], Sv[1]};
                        f2 t = S2 - f2{smin, smin};
                        t = t * 2.0f;
                        f2 quo;
                        if (fastdiv)
                            quo = div_rcp2(t, span, rspan);
                        else
                            quo = f2{t.x / span, t.y / span};
                        f2 z = quo + (-1.0f);
                        if (p == 0 && a.do_not_flag_first) z.x = clip2<float>(z.x, -1.0f, 1.0f);
                        zz[p] = z;
                        SS[p] = S2;
                        any_ex = any_ex || (fabsf(z.x) > 1.0f) || (fabsf(z.y) > 1.0f);
                    }
                    const bool slow = __any(any_ex);
#pragma unroll
                    for (int p = 0; p < GP; ++p) {
                        const f2 z = zz[p];
                        f2 phi = {rr.cf[0], rr.cf[0]};
                        bool ex[2] = {false, false};
                        if (!slow) {
                            f2 pp = {1.0f, 1.0f}, pc = z;
#pragma unroll
                            for (int L = 1; L < NP; ++L) {
                                const f2 term = pc * rr.cf[L];
                                phi = phi + term;
                                const f2 u = z * c1[L];
                                const f2 pn = u * pc - pp * c2[L];
                                pp = pc;
                                pc = pn;
                            }
                        } else {
                            float ph[2];
#pragma unroll
                            for (int e = 0; e < 2; ++e) {
                                const float ze = e ? z.y : z.x;
                                const float az = fabsf(ze);
                                ex[e] = az > 1.0f;
                                const float exc = az - 1.0f;
                                const bool neg = ze < 0.0f;
                                float phs = rr.cf[0], pp = 1.0f, pc = ze;
#pragma unroll
                                for (int L = 1; L < NP; ++L) {
                                    float ee = 1.0f + chf[L] * exc;
                                    ee = (neg && (L & 1)) ? -ee : ee;
                                    const float sel = ex[e] ? ee : pc;
                                    const float term = rr.cf[L] * sel;
                                    phs = phs + term;
                                    const float u = c1[L] * ze;
                                    const float pn = u * pc - c2[L] * pp;
                                    pp = pc;
                                    pc = pn;
                                }
                                ph[e] = phs;
                            }
                            phi = f2{ph[0], ph[1]};
                        }
                        const f2 fb = SS[p] - f2{rr.sref, rr.sref};
                        float vout[2];
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const int g = 2 * p + e;
                            vout[e] = ((dq & bad) == 0) ? (e ? phi.y : phi.x) : (e ? fb.y : fb.x);
                            const bool first = (g == 0) && a.do_not_flag_first;
                            if (!first && ex[e] && (rr.q[g] & DQ_SATURATED) == 0) dq |= DQ_NO_LIN_CORR;
                        }
                        f2 xv = {vout[0], vout[1]};
                        if (act) xv = xv * rr.gain;
                        xs[p * 4 * C2_COLS] = xv;
                    }
                    DQ[slot * C2_COLS + col] = dq;
                    uint32_t w0 = 0, w1 = 0;
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        if (g < 4)
                            w0 |= (rr.q[g] & 0xffu) << (8 * g);
                        else
                            w1 |= (rr.q[g] & 0xffu) << (8 * (g - 4));
                    }
                    QS[slot * C2_COLS + col] = uint2{w0, w1};
                    GS[slot * C2_COLS + col] = rr.gain;
                }
            }
            CH_T(0)
            __syncthreads();
            CH_T(1)
            // ---- S2: issue the raw loads of row r+4 (consumed in S1 of the next step), then C of row yc
            fetch_row(r + 4, rr);
            if (do_c && vC) {
                const bool all = __all(vC == 0x1ffu);
                const int sm = (yc - 1) & 3, s0 = yc & 3, sp = (yc + 1) & 3;
                const int so = (yc + 3000) % 3;
                constexpr int NB = 1;  // register budget of the ingest role (the row prefetch is live here)
#pragma unroll
                for (int p0 = 0; p0 < GP; p0 += NB) {
                    const f2 *xm[NB], *x0[NB], *xp[NB];
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        const f2 *xb = X2 + (p0 + b) * 4 * C2_COLS;
                        xm[b] = xb + sm * C2_COLS, x0[b] = xb + s0 * C2_COLS, xp[b] = xb + sp * C2_COLS;
                    }
                    f2 f[NB], xc[NB];
                    if (all)
                        fwd_rows_batch<NB, true>(xm, x0, xp, col, kC, vC, f, xc);
                    else
                        fwd_rows_batch<NB, false>(xm, x0, xp, col, kC, vC, f, xc);
#pragma unroll
                    for (int b = 0; b < NB; ++b) O12[((p0 + b) * 3 + so) * C2_COLS + col] = (xc[b] + xc[b]) - f[b];
                }
            }
            CH_T(2)
            __syncthreads();
            CH_T(3)
        }
    } else {
        // =========================================================================== fit waves
        float kF[9];
        unsigned vF = 0;
#pragma unroll
        for (int k = 0; k < 9; ++k) kF[k] = 0.0f;
        const int start = KLD(h->start);
        const RipVariant v0 = rip_load_variant(vars, 0);
        for (int r = R0 - 5; r < R1; ++r) {
            const bool do_e = r >= R0;
            const bool emit = do_e && col >= 2 && col < C2_COLS - 2 && col_ok;
            const unsigned pe = (unsigned)(min(max(r, 0), ny - 1) * nx + cc);
            // ---- S1: loads the finish needs (latency hidden by O2), then the second IPC iterate of row r
            const float e_read = ldg<float>(planes, (unsigned)(NP + 5) * pl4 + pe * 4u);
            const float e_dark = ldg<float>(planes, (unsigned)(NP + 6) * pl4 + pe * 4u);
            const uint32_t e_ff = ldg<uint32_t>(planes, (unsigned)(NP + 8) * pl4 + pe * 4u);
            const uint32_t e_pdq = ldg<uint32_t>(a.pdq, pe * 4u);
            const float e_flat = a.flat ? ldg<float>(a.flat, pe * 4u) : 1.0f;
            float d[G];
            f2 dpair[GP];
            uint32_t qe[G];
            float e_gain = 1.0f;
            uint32_t lin_dq = 0;
            const bool act = emit && col_act && r >= ay0 && r < ay1;
            if (emit) {
                const int sx = r & 3;
                e_gain = GS[sx * C2_COLS + col];
                lin_dq = DQ[sx * C2_COLS + col];
                const uint2 qw = QS[sx * C2_COLS + col];
#pragma unroll
                for (int g = 0; g < G; ++g) qe[g] = ((g < 4 ? qw.x : qw.y) >> (8 * (g & 3))) & 0xffu;
                const bool fastdiv = __all(rcp_safe(e_gain) || !act);
                const float rgain = 1.0f / e_gain;
                const bool all = __all(vF == 0x1ffu || !act);
                const int om_ = (r - 1 + 3000) % 3, o0_ = (r + 3000) % 3, op_ = (r + 1 + 3000) % 3;
                constexpr int NB = (GP % 2 == 0) ? 2 : 1;
#pragma unroll
                for (int p0 = 0; p0 < GP; p0 += NB) {
                    f2 xc[NB], val[NB];
#pragma unroll
                    for (int b = 0; b < NB; ++b) val[b] = xc[b] = X2[((p0 + b) * 4 + sx) * C2_COLS + col];
                    if (act) {
                        const f2 *om[NB], *o0[NB], *op[NB];
#pragma unroll
                        for (int b = 0; b < NB; ++b) {
                            const f2 *ob = O12 + (p0 + b) * 3 * C2_COLS;
                            om[b] = ob + om_ * C2_COLS, o0[b] = ob + o0_ * C2_COLS, op[b] = ob + op_ * C2_COLS;
                        }
                        f2 f[NB], oc[NB];
                        if (all)
                            fwd_rows_batch<NB, true>(om, o0, op, col, kF, vF, f, oc);
                        else
                            fwd_rows_batch<NB, false>(om, o0, op, col, kF, vF, f, oc);
#pragma unroll
                        for (int b = 0; b < NB; ++b) {
                            const f2 o2 = (oc[b] + xc[b]) - f[b];
                            if (fastdiv)
                                val[b] = div_rcp2(o2, e_gain, rgain);
                            else
                                val[b] = f2{o2.x / e_gain, o2.y / e_gain};
                        }
                    }
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        d[2 * (p0 + b)] = val[b].x;
                        d[2 * (p0 + b) + 1] = val[b].y;
                        dpair[p0 + b] = val[b];
                    }
                }
            }
            CH_T(4)
            __syncthreads();
            CH_T(5)
            // ---- S2: coefficients of the next row's O2, then fit + finish + stores of pixel (r, c)
            float kN[9];
            const unsigned vN = load_k(r + 1, (r + 1 >= R0) && col >= 2 && col < C2_COLS - 2, kN);
            if (emit) {
                if (a.cube_out) {
#pragma unroll
                    for (int g = 0; g < G; ++g) a.cube_out[(unsigned)g * npix + pe] = d[g];
                }
                uint32_t anyq = 0;
#pragma unroll
                for (int g = 0; g < G; ++g) anyq |= qe[g];
                float s, er, ep;
                uint32_t jmask = 0;
                const bool unsat = (qe[G - 1] & DQ_SATURATED) == 0;
                fit_full_pk<G>(dpair, h, v0, a.dense, kvals + v0.k_ofs, diffs + v0.diff_ofs, e_gain, e_read,
                               unsat && act, guard, s, er, ep, jmask);
                if (__any((anyq & DQ_SATURATED) != 0))
                    trunc_layers<G, G - 1>(d, qe, h, vars, kvals, diffs, e_gain, e_read, act, guard, s, er, ep, jmask);
                uint32_t pdq = propagate_flags<G>(qe, jmask, start, e_pdq | lin_dq, a.gdq_out ? a.gdq_out + pe : nullptr, npix);
                if (a.finish) {
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
                        if (__all(rcp_safe(e_flat) && fabsf(s) < 1e18f && fabsf(er2) < 1e18f && fabsf(ep2) < 1e18f &&
                                  (s == 0.0f || fabsf(s) > 1e-18f) && (er2 == 0.0f || er2 > 1e-18f) &&
                                  (ep2 == 0.0f || ep2 > 1e-18f))) {
                            const float rflat = 1.0f / e_flat;
                            s = div_rcp(s, e_flat, rflat);
                            er2 = div_rcp(er2, e_flat, rflat);
                            ep2 = div_rcp(ep2, e_flat, rflat);
                        } else {
                            s = s / e_flat;
                            er2 = er2 / e_flat;
                            ep2 = ep2 / e_flat;
                        }
                    }
                    er = er2;
                    ep = ep2;
                }
                *reinterpret_cast<float *>(reinterpret_cast<char *>(a.slope) + pe * 4u) = s;
                *reinterpret_cast<float *>(reinterpret_cast<char *>(a.err_read) + pe * 4u) = er;
                *reinterpret_cast<float *>(reinterpret_cast<char *>(a.err_poisson) + pe * 4u) = ep;
                *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(a.pdq_out) + pe * 4u) = pdq;
            }
            CH_T(6)
            __syncthreads();
            CH_T(7)
            vF = vN;
#pragma unroll
            for (int k = 0; k < 9; ++k) kF[k] = kN[k];
        }
    }
#ifdef CH_STAMP
    if ((tid & 63) == 0 && a.dbg_buf) {
        unsigned long long *o = a.dbg_buf + ((size_t)blockIdx.x * (C2_THREADS / 64) + (tid >> 6)) * 9;
        for (int i = 0; i < 9; ++i) o[i] += st_[i];
    }
#endif
}

static inline size_t chain2_lds_bytes(int G) {
    return (size_t)(G / 2) * C2_COLS * 8 * (4 + 3) + (size_t)C2_COLS * 4 * (4 + 8 + 4) + (size_t)3 * G * 2 * 8;
}

template <int NP, int G>
static int launch_chain2(rip_ctx *ctx, const RipPlan *plan, const ChainArgs &a) {
    extern double rip_guard_band;
    const size_t lds = chain2_lds_bytes(G);
    static int ncu = 0;
    if (!ncu) {
        hipDeviceProp_t prop;
        RIP_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
        ncu = prop.multiProcessorCount;
    }
    int per_cu = (int)((160 * 1024) / lds);
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 2) per_cu = 2;  // 2 x 8 waves = 4 waves/SIMD at <= 128 VGPRs
    const int nstrips = (a.nx + C2_OUTW - 1) / C2_OUTW;
    int nranges = (int)(((long)ncu * per_cu) / nstrips);
    if (nranges > (a.ny + 7) / 8) nranges = (a.ny + 7) / 8;
    if (nranges < 1) nranges = 1;
    const long grid = (long)nranges * nstrips;
    if (lds > 48 * 1024)
        RIP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(chain2_kernel<NP, G>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((chain2_kernel<NP, G>), dim3((unsigned)grid), dim3(C2_THREADS), lds, ctx->stream, a,
                       reinterpret_cast<const RipPlanHeader *>(plan->dev), plan->d_variants, plan->d_k, plan->d_diffs,
                       rip_guard_band);
    RIP_HIP(ctx, hipGetLastError());
    return RIP_OK;
}
