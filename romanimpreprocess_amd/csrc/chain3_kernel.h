// Fused L1->L2 kernel, wave-private form: ONE WAVE owns a strip of 64 columns (60 outputs + 2 + 2 halo) and marches down a
// range of rows; nothing is shared between waves, so there are no barriers in the row loop.
//
//   * A lane owns its column.  The forward IPC operator (ipc_linearity.py:69-94) is evaluated in PRODUCT form: the pixel
//     at (y, x) multiplies its own value by its own nine coefficients K[:, :, y, x] and the rounded products travel to the
//     destination pixel -- one row up / down through the lane's own registers (the march keeps the last rows), one column
//     left / right through the DPP operand of the accumulating add (v_add_f32_dpp ... wave_shr:1 / wave_shl:1: the shift is
//     part of the add, no extra instruction).  Same products, same accumulation order as the reference, so bit-identical;
//     and every coefficient is loaded exactly once, by the lane that owns the pixel (the gather form of chain2_kernel.h
//     loads each plane at nine neighbouring positions in two roles).
//   * The rows the march keeps (x of the two previous rows, the first iterate of two rows) live in an LDS area PRIVATE to
//     the wave (16-byte units, lane-contiguous: conflict-free ds_read/write_b128): registers stay free for the prefetch of
//     the next raw row and for interleaving the groups' accumulation chains.  One wave's LDS operations execute in issue
//     order, so the ring needs no barrier and no wait between a read and the later overwrite of the same slot.
//   * Per step (new raw row yi): A = reference-pixel apply + bias + Legendre linearity of row yi -> x = gain * phi;
//     C = first Neumann iterate of row yi-1; O2 = second iterate of row r = yi-2 / gain -> the pixel's ramp in registers;
//     F/T = ramp fit with jump detection, saturated refits, flag propagation, finish, stores of pixel (r, c).
//     The raw loads of row yi+1 and the coefficients of the next step are issued between O2 and F.
//   * The arithmetic of every phase is that of chain2_kernel.h / chain_kernel.h (validated bit for bit against the oracle);
//     only the data movement differs.
//
// A workgroup is C3_NW waves on adjacent strips (same rows): they touch neighbouring cache lines at about the same time.
// The grid is exactly resident; the row ranges are equal.
#pragma once
#include "chain2_kernel.h"

#ifndef C3_NW
#define C3_NW 4
#endif
#ifndef C3_WPS   // waves per SIMD the 8-group f32 instantiation is compiled for (register budget 512 / C3_WPS)
#define C3_WPS 3
#endif
#define C3_OUTW 60
#define C3_THREADS (64 * C3_NW)
// One s_barrier per row step keeps the waves of a workgroup within a row of each other: neighbouring strips share cache lines
// (60-column pitch against 128-byte lines), and the second request for a line then finds it in L2 (measured: FETCH_SIZE
// -19 %, kernel time -8 %).  Nothing is exchanged at the barrier; -DC3_NOROWSYNC removes it.
#ifndef C3_NOROWSYNC
#define C3_SYNC() __builtin_amdgcn_s_barrier()
#else
#define C3_SYNC()
#endif

// wave shifts through the DPP operand: c3_shr(v)[lane] = v[lane - 1], c3_shl(v)[lane] = v[lane + 1]; lanes without a
// source read 0 (bound_ctrl).  The compiler folds the move into the consuming add (GCNDPPCombine).
__device__ __forceinline__ float c3_shr(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float c3_shl(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}
__device__ __forceinline__ double c3_shr(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, 0x138, 0xf, 0xf, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), 0x138, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double c3_shl(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, 0x130, 0xf, 0xf, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), 0x130, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// the nine coefficients one application of the forward operator at destination row y needs, each at the SOURCE pixel:
//   m[j] = plane 6+j at row y-1 (dy = +1),  z[j] = plane 3+j at row y (dy = 0),  p[j] = plane j at row y+1 (dy = -1)
// (plane = 3*(1+dy) + (1+dx) of the embedded ipc4d array)
template <typename T>
struct C3KSet {
    T m[3], z[3], p[3];
};

// forward IPC operator at the destination lane from the three source rows held by the lane and its neighbours.
// Term order and edge rule of ipc_linearity.py:69-94; bit k of `valid` = term k exists (ALL: every term does).
template <typename T, bool ALL>
__device__ __forceinline__ T c3_fwd(T vm, T v0, T vp, const C3KSet<T> &k, unsigned valid) {
    T acc = v0 * k.z[1];
#define C3_TERM(kk, expr)                                  \
    {                                                      \
        const T t_ = (expr);                               \
        acc = (ALL || ((valid >> kk) & 1u)) ? acc + t_ : acc; \
    }
    C3_TERM(1, vm * k.m[1])
    C3_TERM(2, vp * k.p[1])
    C3_TERM(3, c3_shr(v0 * k.z[2]))
    C3_TERM(4, c3_shl(v0 * k.z[0]))
    C3_TERM(5, c3_shr(vm * k.m[2]))
    C3_TERM(6, c3_shl(vm * k.m[0]))
    C3_TERM(7, c3_shr(vp * k.p[2]))
    C3_TERM(8, c3_shl(vp * k.p[0]))
#undef C3_TERM
    return acc;
}

template <int NP, int G, int START, typename KT, int WPS>
__global__ __launch_bounds__(C3_THREADS, WPS) void chain3_kernel(ChainArgs a, const RipPlanHeader *__restrict__ h,
                                                                  const RipVariant *__restrict__ vars,
                                                                  const float *__restrict__ kvals,
                                                                  const RipDiff *__restrict__ diffs, double guard) {
    static_assert(G % 2 == 0 && G > 4 && G <= 16, "pairs of groups; the groupdq bytes travel packed four to a word");
    constexpr int QW = (G + 3) / 4;
    constexpr int GP = G / 2;
    constexpr bool K64 = sizeof(KT) == 8;
    using T = KT;                                    // dtype of the Neumann iterates (numpy promotion f32 * KT)
    constexpr int NCH = (C3_NW * C3_OUTW + 4 + 126) / 128 + 1;  // channels a workgroup's columns can touch
    __shared__ double LN[NCH * G * 2];               // channel lines (m, c) of those channels, read-only after the fill
    constexpr int GQ = (G + 3) / 4;                  // 16-byte units of four f32 values per pixel
    constexpr int OQ = K64 ? G / 2 : GQ;             // 16-byte units of the first iterate (two doubles or four floats)
    typedef float c3_f4 __attribute__((ext_vector_type(4)));
    typedef double c3_d2 __attribute__((ext_vector_type(2)));
    __shared__ c3_f4 XR[C3_NW][2][GQ][64];           // x = gain * phi of the two previous rows, slot = row & 1
    __shared__ c3_f4 OR_[C3_NW][2][OQ][64];          // first iterate of the two previous rows (c3_d2 view for f64 ipc4d)

    const RIP_K C2KernArgs *kargs = (const RIP_K C2KernArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int ny = a.ny, nx = a.nx, nb = a.nb;
    const int ay0 = nb, ay1 = ny - nb, ax0 = nb, ax1 = nx - nb;
    const unsigned npix = (unsigned)ny * (unsigned)nx;
    const unsigned pl4 = npix * 4u;
    const int nch = nx / RIP_CW;
    const uint32_t bad = DQ_NO_LIN_CORR | DQ_REFERENCE_PIXEL;

    float c1[NP], c2[NP], chf[NP];
#pragma unroll
    for (int L = 1; L < NP; ++L) {
        c1[L] = (float)((double)(2 * L + 1) / (double)(L + 1));
        c2[L] = (float)((double)L / (double)(L + 1));
        chf[L] = (float)((double)(L * (L + 1)) / 2.0);
    }

    // work split: workgroup = C3_NW adjacent strips x one row range
    const int nstrips = (nx + C3_OUTW - 1) / C3_OUTW;
    const int nwgx = (nstrips + C3_NW - 1) / C3_NW;
    const int nranges = gridDim.x / nwgx;
    const int rows_per = (ny + nranges - 1) / nranges;
    const int wgx = (int)blockIdx.x % nwgx;
    const int R0 = ((int)blockIdx.x / nwgx) * rows_per;
    const int R1 = min(ny, R0 + rows_per);
    const int strip = wgx * C3_NW + wv;
    const int ch0 = max(wgx * C3_NW * C3_OUTW - 2, 0) / RIP_CW;
    for (int i = tid; i < NCH * G * 2; i += C3_THREADS) {
        const int ch = i / (G * 2), g = (i / 2) % G, w = i & 1;
        LN[i] = (ch0 + ch < nch && a.lines) ? a.lines[(g * nch + ch0 + ch) * 2 + w] : 0.0;
    }
    __syncthreads();
    if ((int)blockIdx.x >= nwgx * nranges || R0 >= ny || strip >= nstrips) return;

    const int c = strip * C3_OUTW - 2 + lane;
    const bool col_ok = (c >= 0 && c < nx);
    const bool col_act = (c >= ax0 && c < ax1);
    const int cc = col_ok ? c : (c < 0 ? 0 : nx - 1);
    const int chr = cc / RIP_CW - ch0;
    const bool emit_lane = lane >= 2 && lane < 62 && col_ok;
    const bool edge_wave = (strip * C3_OUTW - 2 < 0) || (strip * C3_OUTW + 62 > nx);  // wave-uniform: some lane is off the frame
    const unsigned cc4 = (unsigned)cc * 4u, cc2 = (unsigned)cc * 2u, cc1 = (unsigned)cc;
    const unsigned ccK = (unsigned)cc * (unsigned)sizeof(KT);
    unsigned colmask = 0;  // bit k: the source column of term k is in the active box (and so is c)
    {
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int dx = (k == 3 || k == 5 || k == 7) ? 1 : (k == 4 || k == 6 || k == 8) ? -1 : 0;
            const int sx = c - dx;
            if (sx >= ax0 && sx < ax1) colmask |= 1u << k;
        }
        if (!col_act) colmask = 0;
    }
    const unsigned lane_c = (lane >= 1 && lane < 63) ? colmask : 0u;  // lanes whose first iterate is read by somebody
    const unsigned lane_o = (lane >= 2 && lane < 62) ? colmask : 0u;  // lanes that emit
    const unsigned row4 = (unsigned)nx * 4u;
    auto rowbits = [&](int y) -> unsigned {  // wave-uniform: terms of destination row y whose source row is active
        if (y < ay0 || y >= ay1) return 0u;
        return ((y + 1 < ay1) ? 0x184u : 0u) | 0x019u | ((y - 1 >= ay0) ? 0x062u : 0u);
    };

    // ---- loaders (buffer loads: scalar base + plane/row scalar offset + loop-invariant per-lane column offset)
    auto fetch_groups = [&](const RIP_K ChainArgs *ka, int y, RowRegs<NP, G> &rr) {
        const unsigned yl = (unsigned)min(max(y, 0), ny - 1);
        const __amdgpu_buffer_rsrc_t rs = c2_rsrc(ka->data), rq = c2_rsrc(ka->gdq), rd = c2_rsrc(ka->dark_data),
                                     rb = c2_rsrc(ka->bias);
        unsigned o4 = yl * row4, o2 = yl * (row4 >> 1), o1 = yl * (row4 >> 2);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            rr.S[g] = c2_ld_u16<0>(rs, cc2, o2);
            rr.q[g] = c2_ld_u8<0>(rq, cc1, o1);
            rr.dk[g] = c2_ld_f32<0>(rd, cc4, o4);
            rr.bs[g] = c2_ld_f32<0>(rb, cc4, o4);
            o4 += pl4;
            o2 += pl4 >> 1;
            o1 += npix;
        }
    };
    auto fetch_coefs = [&](const RIP_K ChainArgs *ka, int y, RowRegs<NP, G> &rr) {
        const unsigned yl = (unsigned)min(max(y, 0), ny - 1);
        const __amdgpu_buffer_rsrc_t rp = c2_rsrc(ka->planes);
        unsigned o4 = yl * row4;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            rr.cf[i] = c2_ld_f32<0>(rp, cc4, o4);
            o4 += pl4;
        }
        rr.smin = c2_ld_f32<0>(rp, cc4, o4);
        rr.smax = c2_ld_f32<0>(rp, cc4, o4 + pl4);
        rr.sref = c2_ld_f32<0>(rp, cc4, o4 + 2u * pl4);
        rr.dq = c2_ld_u32<0>(rp, cc4, o4 + 3u * pl4);
        rr.gain = c2_ld_f32<0>(rp, cc4, o4 + 4u * pl4);
    };
    // coefficient set of destination row y: planes 6..8 at row y-1, 3..5 at row y, 0..2 at row y+1 (rows clamped into the
    // frame; terms whose source lies outside the active box are masked by the caller)
    auto fetch_kset = [&](const RIP_K ChainArgs *ka, int y, C3KSet<T> &k) {
        const __amdgpu_buffer_rsrc_t kr = c2_rsrc(ka->kern);
        const unsigned rowK = row4 * (unsigned)(sizeof(KT) / 4), plK = pl4 * (unsigned)(sizeof(KT) / 4);
        const unsigned om = (unsigned)min(max(y - 1, 0), ny - 1) * rowK, oz = (unsigned)min(max(y, 0), ny - 1) * rowK,
                       op = (unsigned)min(max(y + 1, 0), ny - 1) * rowK;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if constexpr (K64) {
                k.p[j] = c2_ld_f64<0>(kr, ccK, op + (unsigned)j * plK);
                k.z[j] = c2_ld_f64<0>(kr, ccK, oz + (unsigned)(3 + j) * plK);
                k.m[j] = c2_ld_f64<0>(kr, ccK, om + (unsigned)(6 + j) * plK);
            } else {
                k.p[j] = c2_ld_f32<0>(kr, ccK, op + (unsigned)j * plK);
                k.z[j] = c2_ld_f32<0>(kr, ccK, oz + (unsigned)(3 + j) * plK);
                k.m[j] = c2_ld_f32<0>(kr, ccK, om + (unsigned)(6 + j) * plK);
            }
        }
    };

    const RipVariant v0 = rip_load_variant(vars, 0);
    const RipFitConst fc0 = rip_fit_const(h);
    constexpr int start = START;

    // ---- rolling state of the march.  Registers: the coefficient sets and the small per-row words; LDS (private to the
    // wave): x of rows yi-2, yi-1 and the first iterate of rows yi-3, yi-2, slot = row & 1
    uint32_t dq_m = 0, dq_0 = 0;           // linearity dq of rows yi-2, yi-1
    uint32_t qw_m[QW], qw_0[QW];           // packed groupdq bytes of rows yi-2, yi-1
    float gain_m = 1.0f, gain_0 = 1.0f;    // gain of rows yi-2, yi-1
    C3KSet<T> kO, kC;                      // coefficient sets of destination rows yi-2 (second iterate), yi-1 (first)
#pragma unroll
    for (int i = 0; i < QW; ++i) qw_m[i] = qw_0[i] = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) kO.m[j] = kO.z[j] = kO.p[j] = kC.m[j] = kC.z[j] = kC.p[j] = (T)0;
    c3_f4 *const xr = &XR[wv][0][0][lane];    // [slot * GQ * 64 + q * 64]
    c3_f4 *const orr = &OR_[wv][0][0][lane];  // [slot * OQ * 64 + q * 64]
#pragma unroll
    for (int i = 0; i < 2 * GQ; ++i) xr[i * 64] = c3_f4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < 2 * OQ; ++i) orr[i * 64] = c3_f4{0.0f, 0.0f, 0.0f, 0.0f};
    // ring access (G values of one row; compile-time unrolled)
    auto ld_x = [&](int slot, float (&v)[G]) {
#pragma unroll
        for (int q = 0; q < GQ; ++q) {
            const c3_f4 t = xr[(slot * GQ + q) * 64];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * q + e < G) v[4 * q + e] = t[e];
        }
    };
    auto st_x = [&](int slot, const float (&v)[G]) {
#pragma unroll
        for (int q = 0; q < GQ; ++q) {
            c3_f4 t;
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] = (4 * q + e < G) ? v[4 * q + e] : 0.0f;
            xr[(slot * GQ + q) * 64] = t;
        }
    };
    auto ld_o = [&](int slot, T (&v)[G]) {
        if constexpr (K64) {
            const c3_d2 *od = reinterpret_cast<const c3_d2 *>(orr);
#pragma unroll
            for (int q = 0; q < OQ; ++q) {
                const c3_d2 t = od[(slot * OQ + q) * 64];
                v[2 * q] = t[0];
                v[2 * q + 1] = t[1];
            }
        } else {
#pragma unroll
            for (int q = 0; q < OQ; ++q) {
                const c3_f4 t = orr[(slot * OQ + q) * 64];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (4 * q + e < G) v[4 * q + e] = t[e];
            }
        }
    };
    auto st_o = [&](int slot, const T (&v)[G]) {
        if constexpr (K64) {
            c3_d2 *od = reinterpret_cast<c3_d2 *>(orr);
#pragma unroll
            for (int q = 0; q < OQ; ++q) od[(slot * OQ + q) * 64] = c3_d2{v[2 * q], v[2 * q + 1]};
        } else {
#pragma unroll
            for (int q = 0; q < OQ; ++q) {
                c3_f4 t;
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = (4 * q + e < G) ? v[4 * q + e] : 0.0f;
                orr[(slot * OQ + q) * 64] = t;
            }
        }
    };

    RowRegs<NP, G> rr;
    {
        const RIP_K ChainArgs *ka = &kargs->a;
        fetch_coefs(ka, R0 - 2, rr);
        fetch_groups(ka, R0 - 2, rr);
        fetch_kset(ka, R0 - 3, kC);  // set of the first step's C (destination row R0-3: not evaluated, loads stay in bounds)
    }

    for (int yi = R0 - 2; yi <= R1 + 1; ++yi) {
        const RIP_K C2KernArgs *kf = c2_args(kargs);
        const RIP_K ChainArgs *ka = &kf->a;
        const int yc = yi - 1, r = yi - 2;
        const bool do_c = (yc >= R0 - 1) && (yc <= R1);
        const bool do_e = (r >= R0) && (r < R1);
        // ---- loads of the finish of pixel (r, c): consumed at the end of the step
        const unsigned rc_ = (unsigned)min(max(r, 0), ny - 1);
        const unsigned t_row = rc_ * row4;
        const __amdgpu_buffer_rsrc_t rpl = c2_rsrc(ka->planes);
        const float e_read = c2_ld_f32<0>(rpl, cc4, (unsigned)(NP + 5) * pl4 + t_row);
        const float e_dark = c2_ld_f32<0>(rpl, cc4, (unsigned)(NP + 6) * pl4 + t_row);
        const uint32_t e_ff = c2_ld_u32<0>(rpl, cc4, (unsigned)(NP + 8) * pl4 + t_row);
        const uint32_t e_pdq = c2_ld_u32<0>(c2_rsrc(ka->pdq), cc4, t_row);
        const float e_flat_raw = c2_ld_f32<0>(c2_rsrc(ka->flat ? (const void *)ka->flat : (const void *)ka->planes), cc4, t_row);
        const uint32_t e_ddq_raw =
            c2_ld_u32<0>(c2_rsrc(ka->dark_dq ? (const void *)ka->dark_dq : (const void *)ka->planes), cc4, t_row);

#ifdef C3_EXP_LATE_RAW
        if (yi > R0 - 2) {
            fetch_coefs(ka, yi, rr);
            fetch_groups(ka, yi, rr);
        }
#endif
        // =========================================================== A: refpix apply + bias + linearity of row yi
        float xn[G];
        uint32_t dq_n = 0, qw_n[QW];
        float gain_n = 1.0f;
#pragma unroll
        for (int i = 0; i < QW; ++i) qw_n[i] = 0;
        const bool a_full = yi >= 0 && yi < ny;  // wave-uniform
        if (a_full) {
            double rc[G];
#pragma unroll
            for (int g = 0; g < G; ++g) rc[g] = KLD(ka->rowcorr[g * ny + yi]);
            const bool act = col_act && yi >= ay0 && yi < ay1;
            uint32_t dq = rr.dq;
            const float smin = rr.smin;
            const float span = rr.smax - smin;
            const bool fastdiv = __all(rcp_safe(span));
            const float rspan = rip_rcp_mid(span);
            const double yd = (double)yi;
            gain_n = rr.gain;
            const float gmul = act ? rr.gain : 1.0f;  // border pixels keep phi (x * 1 = x exactly)
            constexpr int PB = (GP % 2 == 0) ? 2 : 1;  // pairs per block: their recurrences interleave
#pragma unroll
            for (int pb = 0; pb < GP; pb += PB) {
                f2 zz[PB], SS[PB], tt[PB], quo[PB];
                bool any_ex = false;
#pragma unroll
                for (int b = 0; b < PB; ++b) {
                    const int p = pb + b;
                    float Sv[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int g = 2 * p + e;
                        // reference_subtraction.py:123 and :67-68 in f64, cast back to f32 after each step
                        float S = (float)rr.S[g];
                        const float dk = rr.dk[g];
                        float v = S - dk;
                        v = (float)((double)v - rc[g]);
                        const double *ln = LN + (chr * G + g) * 2;
                        const double iel = ln[0] * yd + ln[1];
                        v = (float)((double)v - iel);
                        S = v + dk;
                        S = S - rr.bs[g];  // the embedded bias planes have a zero border (x - 0 = x exactly)
                        Sv[e] = S;
                        qw_n[g / 4] |= (rr.q[g] & 0xffu) << (8 * (g & 3));
                    }
                    SS[b] = f2{Sv[0], Sv[1]};
                    const f2 t = SS[b] - f2{smin, smin};
                    tt[b] = t * 2.0f;
                }
                if (fastdiv) {
#pragma unroll
                    for (int b = 0; b < PB; ++b) quo[b] = div_rcp2(tt[b], span, rspan);
                } else {
#pragma unroll
                    for (int b = 0; b < PB; ++b) quo[b] = f2{tt[b].x / span, tt[b].y / span};
                }
#pragma unroll
                for (int b = 0; b < PB; ++b) {
                    f2 z = quo[b] + (-1.0f);
                    if (pb + b == 0 && a.do_not_flag_first) z.x = clip2<float>(z.x, -1.0f, 1.0f);
                    zz[b] = z;
                    any_ex = any_ex || (fabsf(z.x) > 1.0f) || (fabsf(z.y) > 1.0f);
                }
                const bool slow = __any(any_ex);
                // fallback S - Sref where the linearity file flags the pixel (running dq: a flag raised by group g switches
                // groups > g); one vote skips the selects when no lane of the wave can take it in this block
                const bool fallback = slow || __any((dq & bad) != 0);
                f2 phi[PB];
                bool ex[PB][2];
#pragma unroll
                for (int b = 0; b < PB; ++b) ex[b][0] = ex[b][1] = false;
                if (!slow) {
                    f2 pp[PB], pc[PB];
#pragma unroll
                    for (int b = 0; b < PB; ++b) {
                        phi[b] = f2{rr.cf[0], rr.cf[0]};
                        pp[b] = f2{1.0f, 1.0f};
                        pc[b] = zz[b];
                    }
#pragma unroll
                    for (int L = 1; L < NP; ++L) {
#pragma unroll
                        for (int b = 0; b < PB; ++b) {
                            const f2 term = pc[b] * rr.cf[L];
                            phi[b] = phi[b] + term;
                            const f2 u = zz[b] * c1[L];
                            const f2 pn = u * pc[b] - pp[b] * c2[L];
                            pp[b] = pc[b];
                            pc[b] = pn;
                        }
                    }
                } else {
#pragma unroll
                    for (int b = 0; b < PB; ++b) {
                        float ph[2];
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const float ze = e ? zz[b].y : zz[b].x;
                            const float az = fabsf(ze);
                            ex[b][e] = az > 1.0f;
                            const float exc = az - 1.0f;
                            const bool neg = ze < 0.0f;
                            float phs = rr.cf[0], pp = 1.0f, pc = ze;
#pragma unroll
                            for (int L = 1; L < NP; ++L) {
                                float ee = 1.0f + chf[L] * exc;
                                ee = (neg && (L & 1)) ? -ee : ee;
                                const float sel = ex[b][e] ? ee : pc;
                                const float term = rr.cf[L] * sel;
                                phs = phs + term;
                                const float u = c1[L] * ze;
                                const float pn = u * pc - c2[L] * pp;
                                pp = pc;
                                pc = pn;
                            }
                            ph[e] = phs;
                        }
                        phi[b] = f2{ph[0], ph[1]};
                    }
                }
#pragma unroll
                for (int b = 0; b < PB; ++b) {
                    const int p = pb + b;
                    float vout[2] = {phi[b].x, phi[b].y};
                    if (fallback) {
                        const f2 fb = SS[b] - f2{rr.sref, rr.sref};
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const int g = 2 * p + e;
                            vout[e] = ((dq & bad) == 0) ? (e ? phi[b].y : phi[b].x) : (e ? fb.y : fb.x);
                            const bool first = (g == 0) && a.do_not_flag_first;
                            const uint32_t qg = qw_n[g / 4] >> (8 * (g & 3));
                            if (!first && ex[b][e] && (qg & DQ_SATURATED) == 0) dq |= DQ_NO_LIN_CORR;
                        }
                    }
                    const f2 xv = f2{vout[0], vout[1]} * gmul;
                    xn[2 * p] = xv.x;
                    xn[2 * p + 1] = xv.y;
                }
            }
            dq_n = dq;
            if (edge_wave) {  // lanes beyond the frame edge worked on the clamped column: their row is zero
                dq_n = col_ok ? dq : 0u;
#pragma unroll
                for (int g = 0; g < G; ++g) xn[g] = col_ok ? xn[g] : 0.0f;
#pragma unroll
                for (int i = 0; i < QW; ++i) qw_n[i] = col_ok ? qw_n[i] : 0u;
            }
        } else {
#pragma unroll
            for (int g = 0; g < G; ++g) xn[g] = 0.0f;
        }

        // =========================================================== C: first Neumann iterate of row yc
        // slots: rows yi-2 and yi (same parity) share slot sA, rows yi-1 and yi-3 share slot sB
        const int sA = yi & 1, sB = sA ^ 1;
        float xm[G], x0[G];
        ld_x(sA, xm);
        ld_x(sB, x0);
        T o1n[G];
        {
            const unsigned vC = lane_c & rowbits(yc);
            const bool all = __all(vC == 0x1ffu || vC == 0u);
            if (do_c && all) {
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const T f = c3_fwd<T, true>((T)xm[g], (T)x0[g], (T)xn[g], kC, vC);
                    o1n[g] = (T)(x0[g] + x0[g]) - f;
                }
            } else if (do_c) {
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const T f = c3_fwd<T, false>((T)xm[g], (T)x0[g], (T)xn[g], kC, vC);
                    o1n[g] = (T)(x0[g] + x0[g]) - f;
                }
            } else {
#pragma unroll
                for (int g = 0; g < G; ++g) o1n[g] = (T)0;
            }
        }
        st_x(sA, xn);  // row yi takes the slot of row yi-2 (its values are in xm)

        // =========================================================== O2: second iterate of row r, division by the gain
        float d[G];
        f2 dpair[GP];
        const bool act = emit_lane && do_e && col_act && r >= ay0 && r < ay1;
        const float e_gain = gain_m;
        {
            T o1m[G], o10[G];
            ld_o(sB, o1m);  // row yi-3
            ld_o(sA, o10);  // row yi-2
            const unsigned vO = lane_o & rowbits(r);
            const bool all = __all(vO == 0x1ffu || !act);
            if constexpr (K64) {
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const double f = all ? c3_fwd<double, true>(o1m[g], o10[g], o1n[g], kO, vO)
                                         : c3_fwd<double, false>(o1m[g], o10[g], o1n[g], kO, vO);
                    const double o2 = (o10[g] + (double)xm[g]) - f;
                    d[g] = act ? (float)(o2 / (double)e_gain) : xm[g];
                }
            } else {
                const bool fastdiv = __all(rcp_safe(e_gain) || !act);
                const float rgain = rip_rcp_mid(e_gain);
                float o2[G];
                if (all) {
#pragma unroll
                    for (int g = 0; g < G; ++g) o2[g] = (o10[g] + xm[g]) - c3_fwd<float, true>(o1m[g], o10[g], o1n[g], kO, vO);
                } else {
#pragma unroll
                    for (int g = 0; g < G; ++g) o2[g] = (o10[g] + xm[g]) - c3_fwd<float, false>(o1m[g], o10[g], o1n[g], kO, vO);
                }
                if (fastdiv && __all(act || !emit_lane)) {  // interior wave: every lane that emits is active (the others are not read)
#pragma unroll
                    for (int p = 0; p < GP; ++p) {
                        const f2 q = div_rcp2(f2{o2[2 * p], o2[2 * p + 1]}, e_gain, rgain);
                        d[2 * p] = q.x;
                        d[2 * p + 1] = q.y;
                    }
                } else if (fastdiv) {
#pragma unroll
                    for (int p = 0; p < GP; ++p) {
                        const f2 q = div_rcp2(f2{o2[2 * p], o2[2 * p + 1]}, e_gain, rgain);
                        d[2 * p] = act ? q.x : xm[2 * p];
                        d[2 * p + 1] = act ? q.y : xm[2 * p + 1];
                    }
                } else {
#pragma unroll
                    for (int g = 0; g < G; ++g) d[g] = act ? o2[g] / e_gain : xm[g];
                }
            }
#pragma unroll
            for (int p = 0; p < GP; ++p) dpair[p] = f2{d[2 * p], d[2 * p + 1]};
            st_o(sB, o1n);  // row yi-1 takes the slot of row yi-3
        }
        const uint32_t lin_dq = dq_m;
        uint32_t qw[QW];
#pragma unroll
        for (int i = 0; i < QW; ++i) qw[i] = qw_m[i];

        // ---- rotate the per-row words, then request the raw values of row yi+1 and the coefficient set of the next step's C
        // (destination row yi); the set the second iterate just used is free
        dq_m = dq_0, dq_0 = dq_n;
#pragma unroll
        for (int i = 0; i < QW; ++i) qw_m[i] = qw_0[i], qw_0[i] = qw_n[i];
        gain_m = gain_0, gain_0 = gain_n;
        kO = kC;
        {
            const RIP_K ChainArgs *kb = &c2_args(kargs)->a;
            fetch_kset(kb, yi, kC);
#ifndef C3_EXP_LATE_RAW
            fetch_coefs(kb, yi + 1, rr);
            fetch_groups(kb, yi + 1, rr);
#endif
        }
        C3_SYNC();

        // =========================================================== F / T: ramp fit, flags, finish, stores of pixel (r, c)
        if (do_e && emit_lane) {
            const RIP_K C2KernArgs *kg = c2_args(kargs);
            const float e_flat = kg->a.flat ? e_flat_raw : 1.0f;
            const uint32_t e_ddq = kg->a.dark_dq ? e_ddq_raw : 0u;
            const unsigned pe = rc_ * (unsigned)nx + cc1;
            const size_t t_row4 = (size_t)t_row;
            const size_t pe_row = (size_t)(rc_ * (unsigned)nx);
            if (kg->a.cube_out) {
#pragma unroll
                for (int g = 0; g < G; ++g) kg->a.cube_out[(unsigned)g * npix + pe] = d[g];
            }
            RipFitState fs;
            const bool unsat = ((qw[(G - 1) / 4] >> (8 * ((G - 1) & 3))) & DQ_SATURATED) == 0;
            fit_full_pk_a<G, rip_full_valid<G, START>()>(dpair, fc0, v0, kg->a.dense, e_gain, e_read, unsat && act, kg->guard, fs);
            uint32_t qor = 0;
#pragma unroll
            for (int i = 0; i < QW; ++i) qor |= qw[i];
            const bool anysat = (qor & 0x02020202u) != 0u;
            uint32_t jmask = 0;
            fit_full_pk_b<G>(dpair, kg->h, fc0, kg->a.dense, kg->kvals + v0.k_ofs, kg->diffs + v0.diff_ofs, unsat && act, fs, jmask);
            float s = fs.s, er = fs.er, ep = fs.ep;
            if (__any(anysat)) {
                uint32_t qe[G];
#pragma unroll
                for (int g = 0; g < G; ++g) qe[g] = (qw[g / 4] >> (8 * (g & 3))) & 0xffu;
                trunc_layers<G, G - 1>(d, qe, kg->h, kg->vars, kg->kvals, kg->diffs, e_gain, e_read, act, kg->guard, s, er, ep, jmask);
            }
            // ---- T: flag propagation (fitting.py:339-353), finish (gen_cal_image.py:458-475, 213-229, 607-629), stores
            uint8_t *gq = kg->a.gdq_out ? kg->a.gdq_out + pe_row : nullptr;
            uint32_t pdq = propagate_flags_packed<G>(qw, jmask, start, e_pdq | lin_dq, gq, npix, c2_opaque(cc1));
            if (kg->a.finish) {
                const float sd = (act && kg->a.dark_rate) ? s - e_dark : s;
                const bool lean = kg->a.flat && __all(act && rip_mid36(sd) && (er == 0.0f || rip_mid36(er)) &&
                                                      (ep == 0.0f || rip_mid36(ep)) && e_flat > 0.0f && rip_mid36(e_flat));
                if (lean) {
                    const float err = hypot_f32(er, ep);
                    pdq |= e_ddq | e_ff;
                    const float ep2 = ep;  // sqrt(ep * ep)
                    const float e2 = err * err;
                    const float p2 = ep2 * ep2;
                    const float er2 = rip_sqrt_mid(clip_lo<float>(e2 - p2, 0.0f));
                    const float rflat = rip_rcp_mid(e_flat);
                    s = div_rcp(sd, e_flat, rflat);
                    er = div_rcp(er2, e_flat, rflat);
                    ep = div_rcp(ep2, e_flat, rflat);
                } else {
                    float err = hypot_f32(er, ep);
                    float vp = ep * ep;
                    if (!act) {
                        s = 0.0f;
                        err = 0.0f;
                        vp = 0.0f;
                    }
                    if (act && kg->a.dark_rate) s = s - e_dark;
                    if (act) pdq |= e_ddq;
                    float ep2 = sqrtf(vp);
                    const float e2 = err * err;
                    const float p2 = ep2 * ep2;
                    float er2 = sqrtf(clip_lo<float>(e2 - p2, 0.0f));
                    if (kg->a.flat) {
                        pdq |= e_ff;
                        s = s / e_flat;
                        er2 = er2 / e_flat;
                        ep2 = ep2 / e_flat;
                    }
                    er = er2;
                    ep = ep2;
                }
            }
            const unsigned w4 = c2_opaque(cc4);
            *reinterpret_cast<float *>(reinterpret_cast<char *>(kg->a.slope) + t_row4 + w4) = s;
            *reinterpret_cast<float *>(reinterpret_cast<char *>(kg->a.err_read) + t_row4 + w4) = er;
            *reinterpret_cast<float *>(reinterpret_cast<char *>(kg->a.err_poisson) + t_row4 + w4) = ep;
            *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(kg->a.pdq_out) + t_row4 + w4) = pdq;
        }
    }
}

// waves per SIMD the instantiation is compiled for (register budget 512 / WPS per lane)
template <int G, typename KT>
constexpr int chain3_wps() {
    return (G > 8) ? 2 : (sizeof(KT) == 8 ? 2 : C3_WPS);
}

template <int NP, int G, int START, typename KT = float>
static int launch_chain3_s(rip_ctx *ctx, const RipPlan *plan, const ChainArgs &a) {
    extern double rip_guard_band;
    constexpr int WPS = chain3_wps<G, KT>();
    static int ncu = 0;
    if (!ncu) {
        hipDeviceProp_t prop;
        RIP_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
        ncu = prop.multiProcessorCount;
    }
    // exactly resident grid: a second round of workgroups would start only when the first ends (measured: +40 %)
    const int wg_per_cu = (4 * WPS) / C3_NW;
    const int nstrips = (a.nx + C3_OUTW - 1) / C3_OUTW;
    const int nwgx = (nstrips + C3_NW - 1) / C3_NW;
    int nranges = (int)(((long)ncu * wg_per_cu) / nwgx);
    if (nranges > (a.ny + 7) / 8) nranges = (a.ny + 7) / 8;
    if (nranges < 1) nranges = 1;
    // equal ranges: with rows_per = ceil(ny / nranges) fewer ranges may cover the frame
    const int rows_per = (a.ny + nranges - 1) / nranges;
    nranges = (a.ny + rows_per - 1) / rows_per;
    const long grid = (long)nranges * nwgx;
    hipLaunchKernelGGL((chain3_kernel<NP, G, START, KT, WPS>), dim3((unsigned)grid), dim3(C3_THREADS), 0, ctx->stream, a,
                       reinterpret_cast<const RipPlanHeader *>(plan->dev), plan->d_variants, plan->d_k, plan->d_diffs,
                       rip_guard_band);
    RIP_HIP(ctx, hipGetLastError());
    return RIP_OK;
}

// returns the launch status, or 1 when the plan is not one the kernel was compiled for
template <int NP, int G, typename KT = float>
static int launch_chain3(rip_ctx *ctx, const RipPlan *plan, const ChainArgs &a) {
    if (plan->h.start == 0 && plan->dense.valid == rip_full_valid<G, 0>()) return launch_chain3_s<NP, G, 0, KT>(ctx, plan, a);
    if (plan->h.start == 1 && plan->dense.valid == rip_full_valid<G, 1>()) return launch_chain3_s<NP, G, 1, KT>(ctx, plan, a);
    return 1;
}
