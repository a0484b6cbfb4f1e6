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
// down a range of rows.  Per group it keeps rolling 3-row windows of x = gain*phi (linearised data)
// and of the first Neumann iterate in LDS, so the vertical halo costs nothing and the horizontal
// halo is 4/CH_BT of the linearity work.  The grid is exactly resident (2 workgroups per CU): no
// tail.  Step r of the march:
//     P: issue the global loads of row r+3 (raw data of all groups, linearity planes), of the IPC
//        coefficients of destination row r+2 and of what the fit of row r needs, into registers
//     C: first iterate O1 of row r+1 (x rows r..r+2, coefficients loaded one step earlier)   -> LDS
//     E: second iterate of row r (O1 rows r-1..r+1) / gain -> the pixel's ramp in registers,
//        ramp fit + finish of pixel (r, c), results to HBM
//     A: refpix/bias/linearity of row r+3 from the registers filled in P                      -> LDS
// with a barrier after C and after A: the HBM latency of P hides behind C and E.
// The group count and the Legendre order are template parameters (all group loops unrolled).
//
// Instruction issue, not HBM, bounds this kernel at 2 waves/SIMD (LDS-limited occupancy), so groups are
// processed in PAIRS with packed f32 arithmetic (v_pk_mul_f32 / v_pk_add_f32 on float2: two IEEE
// operations per instruction, same rounding as the scalar forms): x and O1 are stored pair-interleaved
// in LDS (one ds_read_b64 per neighbour and pair).
//
// Roofline: HBM.  Algorithmic bytes per pixel: G*(2 + 4 + 4 + 1) in, G out (groupdq),
// 4*(NP+3) + 4 linearity, 36 ipc4d, 4 gain, 4 read, 4 dark rate, 4 flat, 4 flags, 4 pdq in, 16 out.
#pragma once
#include "rip_common.h"

#include "device_rampfit.h"

#define CH_BT 256
#define CH_OUTW (CH_BT - 4)

// Diagnostic build (-DCH_STAMP): per-phase cycle sums of every wave go to ChainArgs::dbg_buf (9 x u64 per wave).
#ifdef CH_STAMP
#define CH_T(i)                                                     \
    {                                                               \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_s_waitcnt(0xC07F);                         \
        st_[i] += t_ - tl_;                                         \
        tl_ = t_;                                                   \
    }
#else
#define CH_T(i)
#endif

typedef float f2 __attribute__((ext_vector_type(2)));

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
__device__ __forceinline__ f2 div_rcp2(f2 a, float b, float rb) {
    const f2 nb = {-b, -b}, rr = {rb, rb};
    const f2 q0 = a * rr;
    const f2 r0 = __builtin_elementwise_fma(nb, q0, a);
    const f2 q1 = __builtin_elementwise_fma(r0, rr, q0);
    const f2 r1 = __builtin_elementwise_fma(nb, q1, a);
    return __builtin_elementwise_fma(r1, rr, q1);
}
__device__ __forceinline__ bool rcp_safe(float b) {
    const float ab = fabsf(b);
    return ab > 1e-18f && ab < 1e18f;
}

// forward IPC operator at column `t` of three LDS rows (rm = row y-1, r0 = row y, rp = row y+1);
// term order and edge rule of ipc_linearity.py:69-94 (see ipc.hip).  ALL = every source is active
// (interior pixel): no per-term selects.  V is a scalar (one group) or a float2 (a pair of groups).
template <typename V, typename SV, typename KT, bool ALL>
__device__ __forceinline__ V fwd_rows(const SV *rm, const SV *r0, const SV *rp, int t, const KT (&kk)[9], unsigned valid) {
    V acc = V(r0[t]) * V(kk[0]);
    V p;
#define CH_TERM(k, src)       \
    p = V(src) * V(kk[k]);    \
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

// fwd_rows for NB pairs of groups at once: all 9*NB LDS reads are issued first (one exposed LDS latency per
// batch instead of one per term), then NB independent accumulation chains run interleaved (each chain keeps the
// reference's term order).  ctr[b] returns the centre value r0[b][t].
// The nine coefficients travel as five register pairs (k0,k1) (k2,k3) ... (k8,-): a packed multiply broadcasts
// either half of a pair through op_sel, so no per-coefficient copy into a (k,k) pair is needed.
#define RIP_KSPLAT(kk2, k) (((k) & 1) ? f2{(kk2)[(k) / 2].y, (kk2)[(k) / 2].y} : f2{(kk2)[(k) / 2].x, (kk2)[(k) / 2].x})
template <int NB, bool ALL>
__device__ __forceinline__ void fwd_rows_batch(const f2 *const (&rm)[NB], const f2 *const (&r0)[NB],
                                               const f2 *const (&rp)[NB], int t, const f2 (&kk2)[5], unsigned valid,
                                               f2 (&f)[NB], f2 (&ctr)[NB]) {
    f2 v[NB][9];
    __builtin_amdgcn_sched_barrier(0);  // batches do not overlap: the reads of the next batch stay behind this one's sums
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        v[b][5] = rm[b][t - 1];
        v[b][1] = rm[b][t];
        v[b][6] = rm[b][t + 1];
        v[b][3] = r0[b][t - 1];
        v[b][0] = r0[b][t];
        v[b][4] = r0[b][t + 1];
        v[b][7] = rp[b][t - 1];
        v[b][2] = rp[b][t];
        v[b][8] = rp[b][t + 1];
    }
    __builtin_amdgcn_sched_barrier(0);
    f2 acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = v[b][0] * RIP_KSPLAT(kk2, 0);
#pragma unroll
    for (int k = 1; k < 9; ++k) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const f2 p = v[b][k] * RIP_KSPLAT(kk2, k);
            acc[b] = (ALL || ((valid >> k) & 1u)) ? acc[b] + p : acc[b];
        }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        f[b] = acc[b];
        ctr[b] = v[b][0];
    }
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
    static_assert(G % 2 == 0, "groups are processed in pairs");
    using T = typename ChPromote<float, KT>::type;  // gain is f32 on this path: x is f32, iterates are T
    constexpr bool PK = sizeof(T) == 4;             // packed-pair arithmetic for the f32 IPC kernel
    constexpr int GP = G / 2;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    // x ring, pair-interleaved: X2[pair][slot][column] = (x of group 2*pair, x of group 2*pair+1)
    // O1 ring: same layout (f32) or planar doubles [G][3][CH_BT] (f64 ipc4d)
    T *O1 = reinterpret_cast<T *>(lds_raw);
    f2 *O12 = reinterpret_cast<f2 *>(lds_raw);
    f2 *X2 = reinterpret_cast<f2 *>(O1 + G * 3 * CH_BT);          // [GP][3][CH_BT]
    float *R = reinterpret_cast<float *>(X2 + GP * 3 * CH_BT);     // [G][CH_BT]  ramp (saturated pixels' refits only)
    uint8_t *Q = reinterpret_cast<uint8_t *>(R + G * CH_BT);       // [G][CH_BT]
    uint8_t *J = Q + G * CH_BT;                                    // [G][CH_BT]
    double *LN = reinterpret_cast<double *>(J + G * CH_BT);        // [3][G][2]   channel lines of this strip
    double *RC = LN + 3 * G * 2;                                   // [rows+4][G] row corrections of this row range

    const int tid = threadIdx.x;
    const int ny = a.ny, nx = a.nx, nb = a.nb;
    const int ay0 = nb, ay1 = ny - nb, ax0 = nb, ax1 = nx - nb;
    const unsigned npix = (unsigned)ny * (unsigned)nx;
    const unsigned pl4 = npix * 4u;  // bytes per f32 plane
    const KT *__restrict__ kern = reinterpret_cast<const KT *>(a.kern);
    const int nch = nx / RIP_CW;
    const uint32_t bad = DQ_NO_LIN_CORR | DQ_REFERENCE_PIXEL;
    const float *__restrict__ planes = a.planes;
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

    // Work split: the frame is cut into `nranges` row ranges; workgroup w handles strip (w % nstrips) of range
    // (w / nstrips): workgroups with consecutive ids march down adjacent strips of the same rows.
    const int nstrips = (nx + CH_OUTW - 1) / CH_OUTW;
    const int nranges = gridDim.x / nstrips;
    const int rows_per = (ny + nranges - 1) / nranges;
    const int strip = (int)blockIdx.x % nstrips;
    const int R0 = ((int)blockIdx.x / nstrips) * rows_per;
    const int R1 = min(ny, R0 + rows_per);  // rows [R0, R1) of this strip
    if ((int)blockIdx.x >= nstrips * nranges || R0 >= ny) return;
    const int c = strip * CH_OUTW - 2 + tid;  // column of this thread
    const bool col_ok = (c >= 0 && c < nx);
    const bool col_act = (c >= ax0 && c < ax1);
    const int cc = col_ok ? c : 0;  // clamped: out-of-frame lanes load valid addresses and discard
    const int ch0 = max(strip * CH_OUTW - 2, 0) / RIP_CW;  // first channel this strip touches
    const int chr = cc / RIP_CW - ch0;                     // 0..2
    // stage the reference-pixel tables of this (strip, row range): lines[g][ch0..ch0+2], rowcorr[g][R0-2..R1+1]
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

    const RipVariant v0 = rip_load_variant(vars, 0);
    const RipFitConst fc0 = rip_fit_const(h);
#ifdef CH_STAMP
    unsigned long long st_[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
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

        // ---- P: issue loads (fit inputs of row r; IPC coefficients of row r+2; row yi raw inputs).
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
            const int yl = min(max(yi, 0), ny - 1);
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
        // ---- C: O1 of row r+1 = (x + x) - fwd(x)
        if (do_c && vB) {
            const bool all = __all(vB == 0x1ffu);  // wave-uniform: every active lane is an interior pixel
            if constexpr (PK) {
#pragma unroll
                for (int p = 0; p < GP; ++p) {
                    const f2 *xb = X2 + p * 3 * CH_BT;
                    const f2 *xm = xb + s0 * CH_BT, *x0 = xb + s1 * CH_BT, *xp = xb + s2 * CH_BT;
                    const f2 f = all ? fwd_rows<f2, f2, KT, true>(xm, x0, xp, tid, kB, vB)
                                     : fwd_rows<f2, f2, KT, false>(xm, x0, xp, tid, kB, vB);
                    const f2 xc = x0[tid];
                    O12[(p * 3 + s1) * CH_BT + tid] = (xc + xc) - f;
                }
            } else {
                const float *Xf = reinterpret_cast<const float *>(X2);
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    // scalar view of the pair-interleaved ring: element (g&1) of pair g/2, stride 2 floats per column
                    const float *xb = Xf + (g / 2) * 3 * CH_BT * 2 + (g & 1);
                    auto X_at = [&](int slot, int col) { return xb[(slot * CH_BT + col) * 2]; };
                    T acc = (T)X_at(s1, tid) * (T)kB[0];
#define CH_TD(k, slot, col)                                   \
    {                                                         \
        const T p_ = (T)X_at(slot, col) * (T)kB[k];           \
        acc = (all || ((vB >> k) & 1u)) ? acc + p_ : acc;     \
    }
                    CH_TD(1, s0, tid)
                    CH_TD(2, s2, tid)
                    CH_TD(3, s1, tid - 1)
                    CH_TD(4, s1, tid + 1)
                    CH_TD(5, s0, tid - 1)
                    CH_TD(6, s0, tid + 1)
                    CH_TD(7, s2, tid - 1)
                    CH_TD(8, s2, tid + 1)
#undef CH_TD
                    const float xc = X_at(s1, tid);
                    O1[(g * 3 + s1) * CH_BT + tid] = (T)(xc + xc) - acc;
                }
            }
        }
        CH_T(1)
        __syncthreads();
        CH_T(2)

        // ---- E: O2 of row r, ramp fit, outputs
        if (emit) {
            const bool act = col_act && r >= ay0 && r < ay1;
            const bool fastdiv = __all(rcp_safe(e_gain) || !act);  // wave-uniform
            const float rgain = 1.0f / e_gain;
            const bool all = __all(vA == 0x1ffu || !act);  // wave-uniform
            float d[G];
            f2 dpair[GP];
            if constexpr (PK) {
#pragma unroll
                for (int p = 0; p < GP; ++p) {
                    const f2 xc = X2[(p * 3 + s0) * CH_BT + tid];
                    f2 val = xc;
                    if (act) {
                        const f2 *ob = O12 + p * 3 * CH_BT;
                        const f2 *om = ob + s2 * CH_BT, *o0 = ob + s0 * CH_BT, *op = ob + s1 * CH_BT;
                        const f2 f = all ? fwd_rows<f2, f2, KT, true>(om, o0, op, tid, kA, vA)
                                         : fwd_rows<f2, f2, KT, false>(om, o0, op, tid, kA, vA);
                        const f2 o2 = (o0[tid] + xc) - f;
                        if (fastdiv)
                            val = div_rcp2(o2, e_gain, rgain);
                        else
                            val = f2{o2.x / e_gain, o2.y / e_gain};
                    }
                    d[2 * p] = val.x;
                    d[2 * p + 1] = val.y;
                    dpair[p] = val;
                }
            } else {
                const float *Xf = reinterpret_cast<const float *>(X2);
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const float xc = Xf[(((g / 2) * 3 + s0) * CH_BT + tid) * 2 + (g & 1)];
                    float val = xc;
                    if (act) {
                        const T *ob = O1 + g * 3 * CH_BT;
                        const T *om = ob + s2 * CH_BT, *o0 = ob + s0 * CH_BT, *op = ob + s1 * CH_BT;
                        const T f = all ? fwd_rows<T, T, KT, true>(om, o0, op, tid, kA, vA)
                                        : fwd_rows<T, T, KT, false>(om, o0, op, tid, kA, vA);
                        const T o2 = (o0[tid] + (T)xc) - f;
                        val = (float)(o2 / (T)e_gain);
                    }
                    d[g] = val;
                }
#pragma unroll
                for (int p = 0; p < GP; ++p) dpair[p] = f2{d[2 * p], d[2 * p + 1]};
            }
            uint32_t anyq = 0;
#pragma unroll
            for (int g = 0; g < G; ++g) anyq |= qe[g];
            if (a.cube_out) {
#pragma unroll
                for (int g = 0; g < G; ++g) a.cube_out[(unsigned)g * npix + pe] = d[g];
            }
            CH_T(6)
            float s, er, ep;
            uint32_t pdq;
            const uint32_t pdq_in = e_pdq | d0;
            if (anyq & DQ_SATURATED) {
                // some group is saturated: general path with truncated refits (ramp staged in LDS)
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    R[g * CH_BT + tid] = d[g];
                    Q[g * CH_BT + tid] = (uint8_t)qe[g];
                    J[g * CH_BT + tid] = 0;
                }
                rampfit_pixel<float, CH_BT>(R + tid, Q + tid, J + tid, G, h, vars, kvals, diffs, e_gain, e_read, act,
                                            guard, pdq_in, a.gdq_out ? a.gdq_out + pe : nullptr, npix, s, er, ep, pdq);
            } else {
                uint32_t jmask = 0;
                fit_full_pk<G>(dpair, h, fc0, v0, a.dense, kvals + v0.k_ofs, diffs + v0.diff_ofs, e_gain, e_read,
                               act, guard, s, er, ep, jmask);
                // flag propagation (fitting.py:339-353) without saturation
                uint32_t orq = 0;
                bool all_dnu = true;
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const uint32_t rq = qe[g] | (((jmask >> g) & 1u) ? DQ_JUMP_DET : 0u);
                    if (a.gdq_out) a.gdq_out[(unsigned)g * npix + pe] = (uint8_t)rq;
                    orq |= rq;
                    all_dnu = all_dnu && ((rq & DQ_DO_NOT_USE) != 0);
                }
                uint32_t pdq2 = orq & ~DQ_DO_NOT_USE;
                if (all_dnu) pdq2 |= DQ_DO_NOT_USE;
                pdq = (pdq_in & DQ_REFERENCE_PIXEL) ? pdq_in : (pdq_in | pdq2);
            }
            CH_T(7)
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

        CH_T(3)
        // ---- A: refpix apply + bias + linearity of row yi -> x slot of row r (its x was last read above,
        //         by this thread only); lin dq of the row enters the d-pipeline
        uint32_t d3 = 0;
        if (do_ingest) {
            f2 *xs = X2 + s0 * CH_BT + tid;
            if (!(row_in && col_ok)) {
#pragma unroll
                for (int p = 0; p < GP; ++p) xs[p * 3 * CH_BT] = f2{0.0f, 0.0f};
            } else {
                const bool act = col_act && yi >= ay0 && yi < ay1;
                const float smin = rr.smin;
                const float span = rr.smax - smin;
                const bool fastdiv = __all(rcp_safe(span));  // wave-uniform
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
                const bool slow = __any(any_ex);  // some sample extrapolates: series with the linear branch
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
                    // running lin dq: a flag raised by group g switches groups > g to the S - Sref fallback
                    const f2 fb = SS[p] - f2{rr.sref, rr.sref};
                    float vout[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int g = 2 * p + e;
                        vout[e] = ((dq & bad) == 0) ? (e ? phi.y : phi.x) : (e ? fb.y : fb.x);
                        const bool first = (g == 0) && a.do_not_flag_first;
                        if (!first && ex[e] && (rr.q[g] & DQ_SATURATED) == 0) dq |= DQ_NO_LIN_CORR;
                    }
                    // active pixels enter the IPC stage as gain*phi; border pixels keep phi (no IPC there)
                    f2 xv = {vout[0], vout[1]};
                    if (act) xv = xv * rr.gain;
                    xs[p * 3 * CH_BT] = xv;
                }
                d3 = dq;
            }
        }
        CH_T(4)
        __syncthreads();
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
        unsigned long long *o = a.dbg_buf + ((size_t)blockIdx.x * (CH_BT / 64) + (tid >> 6)) * 9;
        for (int i = 0; i < 9; ++i) o[i] += st_[i];
    }
#endif
}

static inline size_t chain_lds_bytes(int G, int k_dtype) {
    const size_t t = k_dtype == RIP_F64 ? 8 : 4;
    return (size_t)G * CH_BT * (3 * t + 3 * 4 + 4 + 2);
}
static inline size_t chain_lds_tables(int G, int rc_rows) { return (size_t)(3 * G * 2 + rc_rows * G) * 8; }

template <int NP, int G, typename KT>
static int launch_chain(rip_ctx *ctx, const RipPlan *plan, const ChainArgs &a) {
    const size_t lds0 = chain_lds_bytes(G, sizeof(KT) == 8 ? RIP_F64 : RIP_F32);
    const int ncu = ctx->ncu;
    const int per_cu = (int)((150 * 1024) / lds0) < 1 ? 1 : (int)((150 * 1024) / lds0);
    const int nstrips = (a.nx + CH_OUTW - 1) / CH_OUTW;
    const long resident = (long)ncu * (per_cu > 4 ? 4 : per_cu);
    int nranges = (int)(resident / nstrips);  // as many row ranges as stay resident together
    if (nranges > (a.ny + 7) / 8) nranges = (a.ny + 7) / 8;  // small frames: at least ~8 rows per workgroup
    if (nranges < 1) nranges = 1;
    const long grid = (long)nranges * nstrips;
    const int per = (a.ny + nranges - 1) / nranges;  // rows per workgroup
    const size_t lds = lds0 + chain_lds_tables(G, per + 4);
    if (lds > 160 * 1024) return rip_fail(ctx, RIP_EINVAL, "fused chain: LDS budget exceeded (%zu bytes)", lds);
    if (lds > 48 * 1024)
        RIP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(chain_kernel<NP, G, KT>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((chain_kernel<NP, G, KT>), dim3((unsigned)grid), dim3(CH_BT), lds, ctx->stream, a,
                       reinterpret_cast<const RipPlanHeader *>(plan->dev), plan->d_variants, plan->d_k, plan->d_diffs,
                       ctx->guard_band);
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
