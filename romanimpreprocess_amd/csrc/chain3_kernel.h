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
//     The raw loads of row yi+1 are issued right after A (their registers are free then), the coefficients of the next
//     step after O2.  The channel lines and the dense fit table are staged in LDS once per workgroup, the row
//     corrections of the next row come with one wide scalar load a step ahead: no serialised scalar-memory round trips
//     inside the row loop besides the kernel arguments.
//   * The arithmetic of every phase is that of chain2_kernel.h / chain_common.h (validated bit for bit against the oracle);
//     only the data movement differs.
//
// A workgroup is C3_NW waves on adjacent strips (same rows): they touch neighbouring cache lines at about the same time.
// The grid is exactly resident; the row ranges are equal.
#pragma once
#include <algorithm>

#include "chain2_kernel.h"

#ifndef C3_NW   // waves (adjacent strips) per workgroup.  Same-box A/B on the bench's non-periodic frame (f64 ipc4d): 4 waves 1.41 ms,
                // 2 waves 1.46-1.66 ms (on a tiled, homogeneous frame 2 waves are 3-4 % faster: 69 strips pack better; the real
                // frame's clustered saturated / jump pixels want the larger group), 3 waves 1.58 ms
#define C3_NW 4
#endif
#ifndef C3_WPS   // waves per SIMD the 8-group f32 instantiation is compiled for (register budget 512 / C3_WPS)
#define C3_WPS 3
#endif
#ifndef C3_NBATCH   // groups the forward operator evaluates in lockstep (2 or 4)
#define C3_NBATCH 4
#endif
#ifndef C3_OUTW   // columns a wave emits; -DC3_OUTW=64 -DC3_HALO=0 is a TIMING experiment (aligned windows, wrong strip edges)
#define C3_OUTW 60
#endif
#ifndef C3_HALO
#define C3_HALO 2
#endif
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

// forward IPC operator at the destination lane from the three source rows held by the lane and its neighbours, for NB
// groups in lockstep: the NB products of a term are formed first, then the NB accumulating adds (a DPP operand may not be
// read within two wait states of the instruction that wrote it; NB independent chains also hide the add latency).
// Term order and edge rule of ipc_linearity.py:69-94; bit k of `valid` = term k exists (ALL: every term does).
template <typename T, bool ALL, int NB>
__device__ __forceinline__ void c3_fwd(const T (&vm)[NB], const T (&v0)[NB], const T (&vp)[NB], const C3KSet<T> &k,
                                       unsigned valid, T (&acc)[NB]) {
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = v0[b] * k.z[1];
#define C3_TERM(kk, src, coef, SHIFT)                                                   \
    {                                                                                   \
        T t_[NB];                                                                       \
        _Pragma("unroll") for (int b = 0; b < NB; ++b) t_[b] = src[b] * (coef);         \
        _Pragma("unroll") for (int b = 0; b < NB; ++b) {                                \
            const T s_ = SHIFT(t_[b]);                                                  \
            acc[b] = (ALL || ((valid >> kk) & 1u)) ? acc[b] + s_ : acc[b];              \
        }                                                                               \
    }
#define C3_ID(x) (x)
    C3_TERM(1, vm, k.m[1], C3_ID)
    C3_TERM(2, vp, k.p[1], C3_ID)
    C3_TERM(3, v0, k.z[2], c3_shr)
    C3_TERM(4, v0, k.z[0], c3_shl)
    C3_TERM(5, vm, k.m[2], c3_shr)
    C3_TERM(6, vm, k.m[0], c3_shl)
    C3_TERM(7, vp, k.p[2], c3_shr)
    C3_TERM(8, vp, k.p[0], c3_shl)
#undef C3_ID
#undef C3_TERM
}

template <int N>
struct C3Int {
    static constexpr int value = N;
};

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
    constexpr int GQ = (G + 3) / 4;                  // 16-byte units of four f32 values per pixel
    constexpr int XQ = GQ;                           // x-ring units per row
    constexpr int WQ = 1 + (QW > 2 ? 1 : 0);         // units of a row's small words {gain, lin dq, groupdq bytes}
    constexpr int OQ = K64 ? G / 2 : GQ;             // 16-byte units of the first iterate (two doubles or four floats)
    typedef float c3_f4 __attribute__((ext_vector_type(4)));
    typedef double c3_d2 __attribute__((ext_vector_type(2)));
    // LDS (dynamic; chain3_lds_bytes):  XR  [C3_NW][3][XQ][64] x 16 B   x = gain * phi of rows yi-2, yi-1, yi, slot = row mod 3
    //                                   WR  [C3_NW][2][WQ][64] x 16 B   gain, linearity dq and packed groupdq bytes of rows yi-2 / yi (the
    //                                                                   slot is read, then rewritten), yi-1; slot = row & 1
    //                                   OR_ [C3_NW][2][OQ][64] x 16 B   first iterate of the two previous rows
    //                                   FT  C3FitTab                    dense fit table of the plan
    //                                   LN  [NCH][G][2] f64             channel lines (m, c) of the workgroup's channels
    // the last two are read-only after the fill (one barrier), the rings are private to their wave
    extern __shared__ __align__(16) unsigned char c3_lds[];
    c3_f4 *const XR = reinterpret_cast<c3_f4 *>(c3_lds);
    c3_f4 *const OR_ = XR + C3_NW * 3 * XQ * 64;
    c3_f4 *const WR = OR_ + C3_NW * 2 * OQ * 64;
    C3FitTab *const FT = reinterpret_cast<C3FitTab *>(WR + C3_NW * 2 * WQ * 64);
    double *const LN = reinterpret_cast<double *>(FT + 1);

    const RIP_K C2KernArgs *kargs = (const RIP_K C2KernArgs *)__builtin_amdgcn_kernarg_segment_ptr();
#ifdef C3_DBG
    const int dbg = a.dbg;  // timing experiments only (tools/gpu_checks/phase_timing.py): bits switch work off, results invalid
#else
    constexpr int dbg = 0;
#endif
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
    const int bid = c2_xcd_block((int)blockIdx.x, (int)gridDim.x);   // neighbouring workgroups of a row range on one XCD
    const int wgx = bid % nwgx;
    const int R0 = (bid / nwgx) * rows_per;
    const int R1 = min(ny, R0 + rows_per);
    const int strip = wgx * C3_NW + wv;
    const int ch0 = max(wgx * C3_NW * C3_OUTW - C3_HALO, 0) / RIP_CW;
    for (int i = tid; i < NCH * G * 2; i += C3_THREADS) {
        const int ch = i / (G * 2), g = (i / 2) % G, w = i & 1;
        LN[i] = (ch0 + ch < nch) ? a.lines[(g * nch + ch0 + ch) * 2 + w] : 0.0;
    }
    for (int i = tid; i < (int)(sizeof(C3FitTab) / 4); i += C3_THREADS) {
        uint32_t v = 0;
        constexpr int o_pairs = C3_MAXG, o_amin = o_pairs + C3_MAXG * 8;
        if (i < o_pairs)
            v = __float_as_uint(a.dense->K2[i]);
        else if (i < o_amin)
            v = reinterpret_cast<const uint32_t *>(a.dense->pairs)[i - o_pairs];
        else if (i == o_amin)
            v = __float_as_uint(a.dense->amin);
        else if (i == o_amin + 1)
            v = a.dense->valid;
        reinterpret_cast<uint32_t *>(FT)[i] = v;
    }
    __syncthreads();
    if (bid >= nwgx * nranges || R0 >= ny || strip >= nstrips) return;

    const int c = strip * C3_OUTW - C3_HALO + lane;
    const bool col_ok = (c >= 0 && c < nx);
    const bool col_act = (c >= ax0 && c < ax1);
    const int cc = col_ok ? c : (c < 0 ? 0 : nx - 1);
    const int chr = cc / RIP_CW - ch0;
    const bool emit_lane = lane >= 2 && lane < 62 && col_ok;
    const bool edge_wave = (strip * C3_OUTW - C3_HALO < 0) || (strip * C3_OUTW - C3_HALO + 64 > nx);  // wave-uniform: some lane is off the frame
    const unsigned cc4 = (unsigned)cc * 4u, cc2 = (unsigned)cc * 2u, cc1 = (unsigned)cc;
    const unsigned ccK = K64 ? cc4 * 2u : cc4;
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
    // rows this wave's range really uses: R0-2 .. R1+1; the loads of the drain steps are clamped into that band (lines the wave
    // reads anyway) instead of fetching a row of the next range's
    const int ylo = max(R0 - 2, 0), yhi = min(R1 + 1, ny - 1);
    auto fetch_groups = [&](const RIP_K ChainArgs *ka, int y, RowRegs<NP, G> &rr, int g_lo, int g_hi) {
        const unsigned yl = (unsigned)min(max(y, ylo), yhi);
        const __amdgpu_buffer_rsrc_t rs = c2_rsrc(ka->data), rq = c2_rsrc(ka->gdq), rd = c2_rsrc(ka->dark_data),
                                     rb = c2_rsrc(ka->bias);
        unsigned o4 = yl * row4 + (unsigned)g_lo * pl4, o2 = yl * (row4 >> 1) + (unsigned)g_lo * (pl4 >> 1),
                 o1 = yl * (row4 >> 2) + (unsigned)g_lo * npix;
#pragma unroll
        for (int g = g_lo; g < g_hi; ++g) {
            rr.S[g] = c2_ld_u16<0>(rs, cc2, o2);
            rr.q[g] = c2_ld_u8<0>(rq, cc1, o1);
            rr.dk[g] = c2_ld_f32<0>(rd, cc4, o4);
            rr.bs[g] = c2_ld_f32<0>(rb, cc4, o4);
            o4 += pl4;
            o2 += pl4 >> 1;
            o1 += npix;
        }
    };
    // planes i0..i1-1 of [cf[0..NP-1], Smin, Smax, Sref, dq, gain]
    auto fetch_coefs = [&](const RIP_K ChainArgs *ka, int y, RowRegs<NP, G> &rr, int i0, int i1) {
        const unsigned yl = (unsigned)min(max(y, ylo), yhi);
        const __amdgpu_buffer_rsrc_t rp = c2_rsrc(ka->planes);
        unsigned o4 = yl * row4 + (unsigned)i0 * pl4;
#pragma unroll
        for (int i = i0; i < i1; ++i) {
            if (i < NP)
                rr.cf[i] = c2_ld_f32<0>(rp, cc4, o4);
            else if (i == NP)
                rr.smin = c2_ld_f32<0>(rp, cc4, o4);
            else if (i == NP + 1)
                rr.smax = c2_ld_f32<0>(rp, cc4, o4);
            else if (i == NP + 2)
                rr.sref = c2_ld_f32<0>(rp, cc4, o4);
            else if (i == NP + 3)   // the flag word: linearity dq merged with the flat flags / dark dq this call applies (RipCal)
                rr.dq = c2_ld_u32<0>(rp, cc4, yl * row4 + (unsigned)(NP + ka->merged_dq) * pl4);
            else
                rr.gain = c2_ld_f32<0>(rp, cc4, o4);
            o4 += pl4;
        }
    };
    // coefficient set of destination row y: planes 6..8 at row y-1, 3..5 at row y, 0..2 at row y+1 (rows clamped into the
    // frame; terms whose source lies outside the active box are masked by the caller)
    auto fetch_kset = [&](const RIP_K ChainArgs *ka, int y, C3KSet<T> &k) {
        const __amdgpu_buffer_rsrc_t kr = c2_rsrc(ka->kern);
        const unsigned rowK = row4 * (unsigned)(sizeof(KT) / 4), plK = pl4 * (unsigned)(sizeof(KT) / 4);
        const unsigned om = (unsigned)min(max(y - 1, ylo), yhi) * rowK, oz = (unsigned)min(max(y, ylo), yhi) * rowK,
                       op = (unsigned)min(max(y + 1, ylo), yhi) * rowK;
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
    C3DenseLds<G> dtab;
    dtab.t = FT;
#pragma unroll
    for (int t = 0; t < G; ++t) dtab.k2v[t] = KLD(a.dense->K2[t]);

    // ---- rolling state of the march.  Registers: the two coefficient sets; LDS (private to the wave): x and the small
    // per-row words of rows yi-2, yi-1, yi, the first iterate of rows yi-3, yi-2
    C3KSet<T> kO, kC;                      // coefficient sets of destination rows yi-2 (second iterate), yi-1 (first)
#pragma unroll
    for (int j = 0; j < 3; ++j) kO.m[j] = kO.z[j] = kO.p[j] = kC.m[j] = kC.z[j] = kC.p[j] = (T)0;
    c3_f4 *const xr = XR + wv * 3 * XQ * 64 + lane;    // [slot * XQ * 64 + q * 64]
    c3_f4 *const orr = OR_ + wv * 2 * OQ * 64 + lane;  // [slot * OQ * 64 + q * 64]
    c3_f4 *const wr = WR + wv * 2 * WQ * 64 + lane;    // [slot * WQ * 64 + q * 64]
#pragma unroll
    for (int i = 0; i < 3 * XQ; ++i) xr[i * 64] = c3_f4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < 2 * OQ; ++i) orr[i * 64] = c3_f4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < 2 * WQ; ++i) wr[i * 64] = c3_f4{1.0f, 0.0f, 0.0f, 0.0f};
    // ring access, NB groups starting at group g0 (compile-time unrolled; NB * sizeof(value) is a multiple of 16 B except for
    // the last unit of a 6-group ramp, which is padded)
    auto ld_x = [&](int slot, int g0, auto &v, auto nb_tag) {
        constexpr int NBB = decltype(nb_tag)::value;
#pragma unroll
        for (int b = 0; b < NBB; ++b) {  // (one 16-byte read per unit: identical reads are merged)
            const c3_f4 t = xr[(slot * XQ + (g0 + b) / 4) * 64];
            v[b] = t[(g0 + b) & 3];
        }
    };
    auto st_x = [&](int slot, const float (&v)[G]) {
#pragma unroll
        for (int q = 0; q < GQ; ++q) {
            c3_f4 t;
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] = (4 * q + e < G) ? v[4 * q + e] : 0.0f;
            xr[(slot * XQ + q) * 64] = t;
        }
    };
    // the row's small words: {gain, linearity dq, groupdq bytes of groups 0-3, 4-7} (+ {groups 8-11, 12-15, -, -})
    auto st_w = [&](int slot, float gain, uint32_t dq, const uint32_t (&qw)[QW]) {
        wr[slot * WQ * 64] = c3_f4{gain, __uint_as_float(dq), __uint_as_float(qw[0]), __uint_as_float(qw[1])};
        if constexpr (QW > 2)
            wr[(slot * WQ + 1) * 64] = c3_f4{__uint_as_float(qw[2]), __uint_as_float(QW > 3 ? qw[QW - 1] : 0u), 0.0f, 0.0f};
    };
    auto ld_w = [&](int slot, float &gain, uint32_t &dq, uint32_t (&qw)[QW]) {
        const c3_f4 t = wr[slot * WQ * 64];
        gain = t[0], dq = __float_as_uint(t[1]), qw[0] = __float_as_uint(t[2]), qw[1] = __float_as_uint(t[3]);
        if constexpr (QW > 2) {
            const c3_f4 u = wr[(slot * WQ + 1) * 64];
            qw[2] = __float_as_uint(u[0]);
            if constexpr (QW > 3) qw[QW - 1] = __float_as_uint(u[1]);
        }
    };
    auto ld_o = [&](int slot, int g0, auto &v, auto nb_tag) {
        constexpr int NBB = decltype(nb_tag)::value;
        if constexpr (K64) {
            const c3_d2 *od = reinterpret_cast<const c3_d2 *>(orr);
#pragma unroll
            for (int b = 0; b < NBB; ++b) {
                const c3_d2 t = od[(slot * OQ + (g0 + b) / 2) * 64];
                v[b] = t[(g0 + b) & 1];
            }
        } else {
#pragma unroll
            for (int b = 0; b < NBB; ++b) {
                const c3_f4 t = orr[(slot * OQ + (g0 + b) / 4) * 64];
                v[b] = t[(g0 + b) & 3];
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
    // groups evaluated in lockstep by the forward operator: a multiple of the ring's 16-byte unit
    constexpr int NB = (G % C3_NBATCH == 0) ? C3_NBATCH : 2;
    static_assert(G % NB == 0, "batches");
    using NbTag = C3Int<NB>;

    constexpr int GQ1 = G / 4, GQ3 = (3 * G) / 4;  // quarters of the groups (load batches)
    RowRegs<NP, G> rr;
    {
        const RIP_K ChainArgs *ka = &kargs->a;
        fetch_coefs(ka, R0 - 2, rr, 0, NP + 5);
        fetch_groups(ka, R0 - 2, rr, 0, GQ3);
        fetch_kset(ka, R0 - 3, kC);  // set of the first step's C (destination row R0-3: not evaluated, loads stay in bounds)
    }

    int sx = 0;  // x-ring slot of row yi-2
    // results of the previous step's pixel, stored one step late: on gfx9 stores and loads complete in issue order (one vmcnt),
    // so stores issued at the end of a step would sit in front of the loads the next step's A waits for
    float pn_s = 0.0f, pn_er = 0.0f, pn_ep = 0.0f;
    uint32_t pn_pdq = 0, pn_rq[QW];
    bool pn_emit = false;
    unsigned pn_trow = 0;
#pragma unroll
    for (int i = 0; i < QW; ++i) pn_rq[i] = 0;
    auto flush_pending = [&]() {
        if (pn_emit) {
            const RIP_K C2KernArgs *kg = c2_args(kargs);
            const unsigned w4 = c2_opaque(cc4);
            const size_t t4 = (size_t)pn_trow;
            if (kg->a.gdq_out && !(dbg & 2)) {
                uint8_t *p = kg->a.gdq_out + (size_t)(pn_trow >> 2);
                const unsigned w1 = c2_opaque(cc1);
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    *(p + w1) = (uint8_t)(pn_rq[g / 4] >> (8 * (g & 3)));
                    p += npix;
                }
            }
            if (!(dbg & 4)) {
                *reinterpret_cast<float *>(reinterpret_cast<char *>(kg->a.slope) + t4 + w4) = pn_s;
                *reinterpret_cast<float *>(reinterpret_cast<char *>(kg->a.err_read) + t4 + w4) = pn_er;
                *reinterpret_cast<float *>(reinterpret_cast<char *>(kg->a.err_poisson) + t4 + w4) = pn_ep;
                *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(kg->a.pdq_out) + t4 + w4) = pn_pdq;
            }
        }
    };
    double rcn[G];  // row corrections of the row the next step ingests (wave-uniform: scalar registers)
    {
        const RIP_K double *rt = rip_k(a.rowcorr_t) + (size_t)min(max(R0 - 2, 0), ny - 1) * G;
#pragma unroll
        for (int g = 0; g < G; ++g) rcn[g] = rt[g];
    }
#ifdef CH_STAMP
    unsigned long long st_[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tl_ = __builtin_amdgcn_s_memtime();
#endif
    for (int yi = R0 - 2; yi <= R1 + 1; ++yi) {
        const RIP_K C2KernArgs *kf = c2_args(kargs);
        const RIP_K ChainArgs *ka = &kf->a;
        const int yc = yi - 1, r = yi - 2;
        const bool do_c = (yc >= R0 - 1) && (yc <= R1);
        const bool do_e = (r >= R0) && (r < R1);
        const unsigned rc_ = (unsigned)min(max(r, max(R0, 0)), yhi);   // (rows before R0: warm-up steps, nothing is emitted)
        const unsigned t_row = rc_ * row4;
        // Loads are requested in small batches spread over the step (a burst of 30-50 loads per wave stalls the issue of every
        // wave of the CU behind the vector-memory queue).  Row yi+1: Legendre planes after A; Smin..gain and the first quarter
        // of the groups after C; the second and third quarter, the IPC coefficients and the tail's planes after O2; the
        // last quarter here, at the top of the row's own step (consumed in the second half of A).
        fetch_groups(ka, yi, rr, GQ3, G);
#ifdef CH_STAMP
        if (a.dbg & 2048) __builtin_amdgcn_s_waitcnt(0);  // exposes what the step still waits for from the previous one
#endif
        CH_T(0)
        // =========================================================== A: refpix apply + bias + linearity of row yi
        float xn[G];
        uint32_t dq_n = 0, qw_n[QW];
        float gain_n = 1.0f;
#pragma unroll
        for (int i = 0; i < QW; ++i) qw_n[i] = 0;
        const bool a_full = yi >= 0 && yi < ny;  // wave-uniform
        if (a_full) {
            double rc[G];  // row corrections of this row (scalars, loaded during the previous step)
#pragma unroll
            for (int g = 0; g < G; ++g) rc[g] = rcn[g];
            {   // ... and those of the next row: one wide scalar load, consumed a whole step later
                const RIP_K double *rt = rip_k(ka->rowcorr_t) + (size_t)min(max(yi + 1, 0), ny - 1) * G;
#pragma unroll
                for (int g = 0; g < G; ++g) rcn[g] = rt[g];
            }
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

        CH_T(1)
        // ---- read noise of pixel (r, c) for the fit, Legendre planes of row yi+1 (this row's are consumed)
        flush_pending();  // (the previous step's results)
        pn_emit = false;
        float e_read;
        {
            const RIP_K ChainArgs *kb = &c2_args(kargs)->a;
            e_read = c2_ld_f32<0>(c2_rsrc(kb->planes), cc4, (unsigned)(NP + 5) * pl4 + t_row);
            fetch_coefs(kb, yi + 1, rr, 0, NP);
        }
        CH_T(2)
        // =========================================================== C: first Neumann iterate of row yc
        // x slots rotate with the row (row mod 3); the first iterate has two (row & 1)
        const int sxm = sx, sx0 = (sx == 2) ? 0 : sx + 1, sxn = (sx0 == 2) ? 0 : sx0 + 1;  // rows yi-2, yi-1, yi
        sx = sx0;
        const int sA = yi & 1, sB = sA ^ 1;  // first iterate: rows yi-2 (sA), yi-3 and yi-1 (sB)
        st_x(sxn, xn);
        float e_gain;
        uint32_t lin_dq, qw[QW];
        ld_w(yi & 1, e_gain, lin_dq, qw);  // row yi-2 ...
        st_w(yi & 1, gain_n, dq_n, qw_n);  // ... whose slot row yi takes (one wave's LDS operations execute in order)
        T o1n[G];
        {
            const unsigned vC = lane_c & rowbits(yc);
            const bool all = __all(vC == 0x1ffu || vC == 0u);
            auto c_pass = [&](auto allc) {
                constexpr bool ALLC = decltype(allc)::value;
#pragma unroll
                for (int g0 = 0; g0 < G; g0 += NB) {
                    float bm[NB], b0[NB];
                    ld_x(sxm, g0, bm, NbTag{});
                    ld_x(sx0, g0, b0, NbTag{});
                    T am[NB], a0[NB], ap[NB], f[NB];
#pragma unroll
                    for (int b = 0; b < NB; ++b) am[b] = (T)bm[b], a0[b] = (T)b0[b], ap[b] = (T)xn[g0 + b];
                    c3_fwd<T, ALLC, NB>(am, a0, ap, kC, vC, f);
#pragma unroll
                    for (int b = 0; b < NB; ++b) o1n[g0 + b] = (T)(b0[b] + b0[b]) - f[b];
                }
            };
            if (do_c && all) {
                c_pass(C2AllT{});
            } else if (do_c) {
                c_pass(C2SomeT{});
            } else {
#pragma unroll
                for (int g = 0; g < G; ++g) o1n[g] = (T)0;
            }
        }
        {
            const RIP_K ChainArgs *kb = &c2_args(kargs)->a;
            fetch_coefs(kb, yi + 1, rr, NP, NP + 5);
            fetch_groups(kb, yi + 1, rr, 0, GQ1);
        }
        CH_T(3)

        // =========================================================== O2: second iterate of row r, division by the gain
        float d[G];
        f2 dpair[GP];
        const bool act = emit_lane && do_e && col_act && r >= ay0 && r < ay1;
        {
            const unsigned vO = lane_o & rowbits(r);
            const bool all = __all(vO == 0x1ffu || !act);
            const bool fastdiv = __all(rcp_safe(e_gain) || !act);
            const bool allact = __all(act || !emit_lane);  // interior wave: every lane that emits is active (the others are not read)
            const float rgain = rip_rcp_mid(e_gain);
            auto o_pass = [&](auto allc) {
                constexpr bool ALLC = decltype(allc)::value;
#pragma unroll
                for (int g0 = 0; g0 < G; g0 += NB) {
                    float bx[NB];
                    T am[NB], a0[NB], ap[NB], f[NB], o2[NB];
                    ld_o(sB, g0, am, NbTag{});  // row yi-3
                    ld_o(sA, g0, a0, NbTag{});  // row yi-2
                    ld_x(sxm, g0, bx, NbTag{});
#pragma unroll
                    for (int b = 0; b < NB; ++b) ap[b] = o1n[g0 + b];
                    c3_fwd<T, ALLC, NB>(am, a0, ap, kO, vO, f);
#pragma unroll
                    for (int b = 0; b < NB; ++b) o2[b] = (a0[b] + (T)bx[b]) - f[b];
                    if constexpr (K64) {
                        float qf[NB];   // (c2_div64_shared: the divisor's half of the division expansion once per batch)
                        c2_div64_shared<NB>(o2, e_gain, act, qf);
#pragma unroll
                        for (int b = 0; b < NB; ++b) d[g0 + b] = act ? qf[b] : bx[b];
                    } else if (ALLC && fastdiv && allact) {  // interior wave
#pragma unroll
                        for (int b = 0; b < NB; b += 2) {
                            const f2 q = div_rcp2(f2{o2[b], o2[b + 1]}, e_gain, rgain);
                            d[g0 + b] = q.x;
                            d[g0 + b + 1] = q.y;
                        }
                    } else if (fastdiv) {
#pragma unroll
                        for (int b = 0; b < NB; b += 2) {
                            const f2 q = div_rcp2(f2{o2[b], o2[b + 1]}, e_gain, rgain);
                            d[g0 + b] = act ? q.x : bx[b];
                            d[g0 + b + 1] = act ? q.y : bx[b + 1];
                        }
                    } else {
#pragma unroll
                        for (int b = 0; b < NB; ++b) d[g0 + b] = act ? o2[b] / e_gain : bx[b];
                    }
                }
            };
            if (all && fastdiv && allact)
                o_pass(C2AllT{});
            else
                o_pass(C2SomeT{});
#pragma unroll
            for (int p = 0; p < GP; ++p) dpair[p] = f2{d[2 * p], d[2 * p + 1]};
            st_o(sB, o1n);  // row yi-1 takes the slot of row yi-3
        }
        CH_T(4)
        // ---- request the coefficient set of the next step's C (destination row yi); the set
        // the second iterate just used is free
        kO = kC;
        {
            const RIP_K ChainArgs *kb = &c2_args(kargs)->a;
            fetch_kset(kb, yi, kC);
            fetch_groups(kb, yi + 1, rr, GQ1, GQ3);
        }
        // what the tail of pixel (r, c) reads: lands while the fit runs
        float e_dark, e_flat_raw;
        uint32_t e_pdq;   // (flat flags and dark dq arrive with the linearity dq: ChainArgs::merged_dq)
        {
            const RIP_K ChainArgs *kb = &c2_args(kargs)->a;
            const __amdgpu_buffer_rsrc_t rpl = c2_rsrc(kb->planes);
            e_dark = c2_ld_f32<0>(rpl, cc4, (unsigned)(NP + 6) * pl4 + t_row);
            e_pdq = c2_ld_u32<0>(c2_rsrc(kb->pdq), cc4, t_row);
            e_flat_raw = c2_ld_f32<0>(c2_rsrc(kb->flat ? (const void *)kb->flat : (const void *)kb->planes), cc4, t_row);
        }
        CH_T(5)
        C3_SYNC();
        CH_T(6)

        // =========================================================== F / T: ramp fit, flags, finish, stores of pixel (r, c)
        if (do_e && emit_lane && !(dbg & 8)) {
            const RIP_K C2KernArgs *kg = c2_args(kargs);
            const float e_flat = kg->a.flat ? e_flat_raw : 1.0f;
            const unsigned pe = rc_ * (unsigned)nx + cc1;
            if (kg->a.cube_out) {
#pragma unroll
                for (int g = 0; g < G; ++g) kg->a.cube_out[(unsigned)g * npix + pe] = d[g];
            }
            RipFitState fs;
            const bool unsat = ((qw[(G - 1) / 4] >> (8 * ((G - 1) & 3))) & DQ_SATURATED) == 0;
            if (dbg & 1) {
                fs.s = d[0], fs.er = e_read, fs.ep = e_gain, fs.live = false;
            } else
                fit_full_pk_a_t<G, rip_full_valid<G, START>(), C3DenseLds<G>>(dpair, fc0, v0, dtab, e_gain, e_read, unsat && act,
                                                                              kg->guard, fs);
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
            CH_T(7)
            // ---- T: flag propagation (fitting.py:339-353), finish (gen_cal_image.py:458-475, 213-229, 607-629), stores
            uint32_t pdq = propagate_flags_packed<G>(qw, jmask, start, e_pdq | lin_dq, nullptr, npix, 0u, pn_rq);
            if (kg->a.finish) {
                const float sd = (act && kg->a.dark_rate) ? s - e_dark : s;
                const bool lean = kg->a.flat && __all(act && rip_mid36(sd) && (er == 0.0f || rip_mid36(er)) &&
                                                      (ep == 0.0f || rip_mid36(ep)) && e_flat > 0.0f && rip_mid36(e_flat));
                if (lean) {
                    const float err = hypot_f32(er, ep);
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
                    float ep2 = sqrtf(vp);
                    const float e2 = err * err;
                    const float p2 = ep2 * ep2;
                    float er2 = sqrtf(clip_lo<float>(e2 - p2, 0.0f));
                    if (kg->a.flat) {
                        s = s / e_flat;
                        er2 = er2 / e_flat;
                        ep2 = ep2 / e_flat;
                    }
                    er = er2;
                    ep = ep2;
                }
            }
            pn_s = s, pn_er = er, pn_ep = ep, pn_pdq = pdq;
            pn_emit = true;
        }
        pn_trow = t_row;
        CH_T(8)
    }
    flush_pending();
#ifdef CH_STAMP
    if (lane == 0 && a.dbg_buf) {
        unsigned long long *o = a.dbg_buf + (((size_t)blockIdx.x * C3_NW + wv) % 4096) * 9;
        for (int i = 0; i < 9; ++i) o[i] += st_[i];
    }
#endif
}

static inline size_t chain3_lds_bytes(int G, size_t ksize) {
    const size_t gq = (size_t)(G + 3) / 4, oq = ksize == 8 ? (size_t)G / 2 : gq;
    const size_t nch = (C3_NW * C3_OUTW + 4 + 126) / 128 + 1;
    const size_t wq = 1 + ((G + 3) / 4 > 2 ? 1 : 0);
    return (size_t)C3_NW * (3 * gq + 2 * oq + 2 * wq) * 64 * 16 + sizeof(C3FitTab) + nch * G * 2 * 8;
}

// waves per SIMD the instantiation is compiled for (register budget 512 / WPS per lane)
template <int G, typename KT>
constexpr int chain3_wps() {
    return (G > 8) ? 2 : (sizeof(KT) == 8 ? 2 : C3_WPS);
}

template <int NP, int G, int START, typename KT = float>
static int launch_chain3_s(rip_ctx *ctx, const RipPlan *plan, const ChainArgs &a) {
    constexpr int WPS = chain3_wps<G, KT>();
    const int ncu = ctx->ncu;
    // exactly resident grid: a second round of workgroups would start only when the first ends (measured: +40 %).
    // Registers allow (4 * WPS) / C3_NW workgroups per CU; LDS may allow fewer (asked from the runtime).
    const size_t lds = chain3_lds_bytes(G, sizeof(KT));
    int &wg_per_cu = ctx->wg_per_cu[reinterpret_cast<const void *>(chain3_kernel<NP, G, START, KT, WPS>)];   // per context
    if (!wg_per_cu) {
        if (lds > 48 * 1024)
            RIP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(chain3_kernel<NP, G, START, KT, WPS>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int nblk = 0;
        RIP_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, chain3_kernel<NP, G, START, KT, WPS>, C3_THREADS, lds));
        wg_per_cu = std::max(1, std::min(nblk, (4 * WPS) / C3_NW));
    }
    const int nstrips = (a.nx + C3_OUTW - 1) / C3_OUTW;
    const int nwgx = (nstrips + C3_NW - 1) / C3_NW;
    int nranges = (int)(((long)ncu * wg_per_cu) / nwgx);
    if (nranges > (a.ny + 7) / 8) nranges = (a.ny + 7) / 8;
    if (nranges < 1) nranges = 1;
    // equal ranges: with rows_per = ceil(ny / nranges) fewer ranges may cover the frame
    const int rows_per = (a.ny + nranges - 1) / nranges;
    nranges = (a.ny + rows_per - 1) / rows_per;
    const long grid = (long)nranges * nwgx;
    hipLaunchKernelGGL((chain3_kernel<NP, G, START, KT, WPS>), dim3((unsigned)grid), dim3(C3_THREADS), lds, ctx->stream, a,
                       reinterpret_cast<const RipPlanHeader *>(plan->dev), plan->d_variants, plan->d_k, plan->d_diffs,
                       ctx->guard_band);
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
