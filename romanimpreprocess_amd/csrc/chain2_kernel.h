// The fused L1->L2 kernel (f32 gain; f32 or f64 ipc4d; 6, 8 or 16 groups), wave-specialised: one launch per
// ramp does reference-pixel apply + bias + Legendre linearity + IPC deconvolution + ramp fit / jump detection / flag propagation
// + dark rate + error split + flat (gen_cal_image.py:533-629; stage arithmetic and reference lines as in linearity.hip, ipc.hip,
// rampfit.hip), every array read from HBM once.
//
// Geometry: the frame is cut into strips of 252 output columns (256-column windows: 2 + 2 halo columns for the two 3 x 3 IPC
// passes); a workgroup owns one strip and one row range and marches down its rows, keeping the rows the 3 x 3 stencils need in
// LDS rings.  The grid is exactly resident (the row ranges are equal, no tail).  A workgroup of 512 threads covers its 256
// columns with TWO ROLES of four waves each (one wave doing every phase needs > 200 registers: 2 waves/SIMD, issue bound):
//     ingest waves (tid < 256)   A: refpix/bias/linearity of row r+3 -> x ring      (raw loads of row r+4 issued between
//                                C: first IPC iterate of row r+2 -> O1 ring             its arithmetic blocks)
//     fit waves    (tid >= 256)  O2: second iterate of row r / gain (reads O1 rows r-1..r+1, x of its own column)
//                                F: ramp fit with jump detection and saturated refits of pixel (r, c)
//                                T: flag propagation, finish (dark rate, error split, flat), stores of pixel (r, c)
// so each role needs <= 128 VGPRs and a CU holds 2 workgroups = 16 waves = 4 waves/SIMD.  That is the 256-column form (f32 ipc4d
// with 6 / 8 groups, the bench path); 16 groups and f64 ipc4d, whose rings would leave one such workgroup per CU or none, run the
// NARROW forms (one wide workgroup per CU that drops rings; see the template below).  Per step:
//     S1: ingest A(r+3)   | fit O2(r), F(r) up to the half-step barrier (C2_BAR)       -- barrier --
//     S2: ingest C(r+2)   | fit: the rest of F(r) and T(r), the ring words of row r+1  -- barrier --
// Everything after O2 is register-only in a fit thread, so the half-step barrier can fall anywhere in it (C2_BAR, chosen per
// instantiation by same-box A/B).  Rings (3 rows each): x = gain*phi; the first iterate O1 (C writes row r+2 into the slot of
// row r-1, which O2(r) finished reading before the barrier; doubles with f64 ipc4d); the per-pixel words that travel from the
// ingest thread of a column to its fit thread (merged flag word, packed groupdq bytes, gain); and the K ring (2 rows): the nine
// IPC coefficients of a pixel are loaded ONCE, by its ingest thread, and handed to its fit thread.  Saturated pixels are
// refitted from registers (trunc_layers), so no per-pixel ramp staging in LDS.
#pragma once
#include "chain_common.h"

#ifndef C2_COLS_DEF
#define C2_COLS_DEF 256   // (128-column workgroups WITH every ring: 4 % slower, profiles/r03_summary.md; the narrow forms drop rings)
#endif
// columns of a workgroup's window: a constant `COLS` of the enclosing template (c2_cols: 256; 384 / 256 in the ring-dropping forms)
#define C2_COLS COLS
#define C2_THREADS (2 * C2_COLS)
// Diagnostic builds only (RIP_TIMING_BUILD: rip_version() then reports a timing build and the Python binding refuses the library
// unless told otherwise): -DC2_DBG switches phases off by ChainArgs::dbg (results invalid by construction), -DCH_STAMP records
// per-phase clock stamps.  Nothing else in this header changes what the kernel computes.
#if (defined(C2_DBG) || defined(CH_STAMP)) && !defined(RIP_TIMING_BUILD)
#error "C2_DBG / CH_STAMP are timing experiments: build them with -DRIP_TIMING_BUILD"
#endif
#ifdef CH_STAMP
#define C2_DRAIN()                                \
    if (a.dbg & 2048) {                           \
        __builtin_amdgcn_s_waitcnt(0);            \
    }
#else
#define C2_DRAIN()
#endif
#define C2_SYNC() __syncthreads()
// Strip geometry: the window of strip s starts at column s * C2_OUTW; its lanes 2 .. C2_COLS-3 emit, lanes 0, 1 and C2_COLS-2,
// C2_COLS-1 are the halo of the two 3 x 3 passes -- except at the frame's edge, where columns 0, 1 and nx-2, nx-1 are emitted
// by those lanes themselves (border pixels: no IPC, no neighbours needed; nb >= 2).  So n strips cover n * C2_OUTW + 4 columns:
// 33 strips of 128 columns cover 4096 exactly (34 with a uniform 2-column offset), 17 of 256.
#define C2_OUTW (C2_COLS - 4)
#define C2_NSTRIPS(nx) (((nx) - 4 + C2_OUTW - 1) / C2_OUTW < 1 ? 1 : ((nx) - 4 + C2_OUTW - 1) / C2_OUTW)
// Where the half-step barrier falls in the fit role: 0 after the first half of the fit, 1 after its second half (and the saturated
// refits), 2 after the flag propagation and the group-flag stores, 3 after the finish and the plane stores.  Same-box A/B on the
// bench frame (profiles/r03_summary.md): f32 ipc4d x 8 groups 0.884 / 0.887 / 0.895 ms for 0 / 1 / 2; 16 groups 2.048 / 2.003 /
// 2.024; f64 ipc4d 1.354 / 1.328 / 1.280 -- the forms whose IPC stages are longer want the barrier later.  -1: per instantiation.
#ifndef C2_BAR
#define C2_BAR -1
#endif
// Pairs per block of the linearity phase for the forms with a 256-register budget (16 groups: 2.048 -> 2.008 ms with 4; f64 ipc4d:
// 1.345 -> 1.360, stays 2)
#ifndef C2_PBW
#define C2_PBW -1
#endif
#ifndef C2_PBC   // f64 ipc4d: pairs the first iterate evaluates in lockstep (-1: per instantiation)
#define C2_PBC -1
#endif
// narrow forms: the fit role requests its coefficients at the top of the step (0), at the end of the step before (1), or two steps
// ahead -- half a step after the ingest role's read of the same lines (2: fabric traffic 1.44 -> 1.19 x (f64), 1.50 -> 1.39 x (16
// groups), 1.38 -> 1.27 x (both), and the kernels 3 % / 0 % / 10 % SLOWER: nine to eighteen more live registers; not bound by bytes)
#ifndef C2_KFIT_EARLY
#define C2_KFIT_EARLY 1
#endif
#ifndef C2_NBO   // f64 ipc4d: groups the second iterate evaluates in lockstep (2 or 4)
#define C2_NBO 2
#endif

// Returns x, opaque to the optimiser: used on loop-invariant per-lane offsets right before a global access so that
// the zero-extension stays next to the address add and instruction selection can use the
// `global_load v, v_off32, s[base:base+1]` form (with the offset hoisted out of the loop it falls back to a 64-bit
// vector add per access).  Emits no instruction.
__device__ __forceinline__ unsigned c2_opaque(unsigned x) {
    asm volatile("" : "+v"(x));
    return x;
}

// Buffer addressing for the global loads: descriptor (base, no stride, no bound) in four scalar registers, the
// loop-invariant column offset of the lane as the vector offset, plane + row as a 32-bit scalar offset -- one s_add per
// load where a 64-bit base costs two.  Every array addressed this way is smaller than 4 GiB.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t c2_rsrc(const void *p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, -1, 0x00020000);
}
// Cache policy of the ONCE-read arrays in the narrow forms (aux bits of the buffer loads; 2 = nt: stream through the caches).  The
// narrow forms' fit role reads the IPC coefficients (NARROW = 2: gain and groupdq bytes too) a second time one row step after the
// ingest role; a step of an XCD's 96 workgroups moves about 4 MB -- the size of its L2 -- so those lines have left L2 by then and
// come back over the fabric (1.4-1.5 x the algorithmic bytes at ~5 TB/s of fabric traffic).  With the hint on everything read once,
// same-box A/B (profiles/r04_summary.md): 16 groups 1.752 -> 1.730 / 1.780 -> 1.738 ms, f64 x 16 groups 2.414 -> 2.383 / 2.411 ->
// 2.399 ms, f64 x 8 groups (NARROW = 1) 1.179 -> 1.237 ms SLOWER -- so it is on for NARROW = 2 only.  Results are identical either
// way.  (256-column form, round 1: hints on the once-read arrays cost 11 %: no second read there.)
#ifndef C2_STREAM_AUX
#define C2_STREAM_AUX 2
#endif
#ifndef C2_STREAM_N1   // the same for NARROW = 1 (A/B)
#define C2_STREAM_N1 0
#endif
template <int AUX = 0>
__device__ __forceinline__ float c2_ld_f32(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, AUX));
}
__device__ __forceinline__ double c2_ld_f64(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
template <int AUX = 0>
__device__ __forceinline__ uint32_t c2_ld_u32(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, AUX);
}
template <int AUX = 0>
__device__ __forceinline__ uint32_t c2_ld_u16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(r, voff, soff, AUX);
}
template <int AUX = 0>
__device__ __forceinline__ uint32_t c2_ld_u8(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return (uint32_t)(uint8_t)__builtin_amdgcn_raw_buffer_load_b8(r, voff, soff, AUX);
}
// XCD-aware block order (the dispatcher deals consecutive block ids round the 8 XCDs, each with its own L2): block ids that
// share an XCD get CONSECUTIVE cells of the (row range, strip) grid, so that neighbouring strips -- whose 256-column windows at
// a 252-column pitch share cache lines and halo columns -- find each other's lines in their L2.  Bijective for any grid size;
// a speed choice only.
__device__ __forceinline__ int c2_xcd_block(int b, int n) {
    const int q = n >> 3, r = n & 7, x = b & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

// The kernel arguments re-read from the kernarg segment (scalar loads from the constant address space) through a
// pointer made opaque once per phase: the ~25 pointers and sizes of ChainArgs then live in scalar registers only
// from their first to their last use inside a phase, instead of all being loop invariants that the register
// allocator spills to VGPR lanes (v_readlane/v_writelane are vector-ALU instructions with scalar hazards).
struct C2KernArgs {  // the kernel's argument list as it lies in the kernarg segment
    ChainArgs a;
    const RipPlanHeader *h;
    const RipVariant *vars;
    const float *kvals;
    const RipDiff *diffs;
    double guard;
};
__device__ __forceinline__ const RIP_K C2KernArgs *c2_args(const RIP_K C2KernArgs *p) {
    asm volatile("" : "+s"(p));
    return p;
}

template <int N>
struct C2Int {
    static constexpr int value = N;
};

// forward IPC operator in f64 for NBB groups in lockstep (f64 ipc4d): v[b][k] = source value of term k of group b, terms in the
// reference's order (0 centre, 1 (y-1, x), 2 (y+1, x), 3 (y, x-1), 4 (y, x+1), 5 (y-1, x-1), 6 (y-1, x+1), 7 (y+1, x-1),
// 8 (y+1, x+1) as the rows / columns of the rings hold them); the NBB products of a term first, then the NBB accumulating adds --
// independent chains that hide the f64 latencies (one group at a time: 1.51 ms, batched: 1.34 ms for 8 groups, same box).
// Term order and edge rule of ipc_linearity.py:69-94.
template <bool ALL, int NBB, typename VT>
__device__ __forceinline__ void c2_ipc9_batch(const VT (&v)[NBB][9], const double (&kk)[9], unsigned valid, double (&acc)[NBB]) {
#pragma unroll
    for (int b = 0; b < NBB; ++b) acc[b] = (double)v[b][0] * kk[0];
#pragma unroll
    for (int k = 1; k < 9; ++k) {
        double p_[NBB];
#pragma unroll
        for (int b = 0; b < NBB; ++b) p_[b] = (double)v[b][k] * kk[k];
#pragma unroll
        for (int b = 0; b < NBB; ++b) acc[b] = (ALL || ((valid >> k) & 1u)) ? acc[b] + p_[b] : acc[b];
    }
}

// (float)(a[g] / (double)bf) for NG numerators and ONE divisor (the second iterate over the pixel's gain, f64 ipc4d), lanes with
// `use` only.  The compiler expands an f64 division into v_div_scale x 2, v_rcp_f64, two Newton steps on the reciprocal (four
// fma), q0 = a * y, r = fma(-b, q0, a), v_div_fmas (= fma(r, y, q0) when nothing was scaled), v_div_fixup (= its first operand for
// finite non-zero operands).  Everything up to y depends on the divisor alone, so it is computed once and each quotient costs a
// multiply and two fma: the same operations on the same operands as the expansion, hence the same bits, as long as v_div_scale
// scales nothing and v_div_fixup has nothing to fix -- which holds for 2^-60 < |b| < 2^60 and every quotient (rounded to f32)
// finite, non-zero and within 2^-59 .. 2^59 (then 2^-119 < |a| < 2^119: far from every scaling rule of the instruction).  One
// wave vote checks that; otherwise every lane takes the division operator.
template <int NG>
__device__ __forceinline__ void c2_div64_shared(const double (&a)[NG], float bf, bool use, float (&qf)[NG]) {
    const double b = (double)bf;
    double y = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    float asum = 0.0f, amin = 3.0e38f;   // NaN / Inf show in the sum, zeros and tiny values in the minimum
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const double q0 = a[g] * y;
        const double r = __builtin_fma(-b, q0, a[g]);
        qf[g] = (float)__builtin_fma(r, y, q0);
        asum = asum + fabsf(qf[g]);
        amin = fminf(amin, fabsf(qf[g]));
    }
    const bool ok = rcp_safe(bf) && asum < 5.7e17f && amin > 1.8e-18f;
    if (!__all(ok || !use)) {
#pragma unroll
        for (int g = 0; g < NG; ++g) qf[g] = (float)(a[g] / b);
    }
}

// Columns of a workgroup's window.  The ring-dropping (NARROW) forms were 128-column workgroups, three (two) per CU; what they
// pay is windows at a 124-column pitch: a window row of a byte plane is one 128-byte line, misaligned it touches two (u16: two ->
// three, f32: four -> five), and the lines shared with the neighbouring strip have left L2 by the time that strip wants them --
// a third of the algorithmic bytes fetched twice.  ONE workgroup of 384 (256) columns with the same rings dropped has the same
// waves per SIMD and LDS per CU and a third (half) of the seams.  C2_N*_COLS: A/B switches, results identical.
#ifndef C2_N1_COLS    // NARROW = 1 (f64 ipc4d x 6 / 8 groups): 43 KB per 128 columns
#define C2_N1_COLS 384
#endif
#ifndef C2_N2_COLS    // NARROW = 2, f32 ipc4d (16 groups): 51 KB per 128 columns
#define C2_N2_COLS 384
#endif
#ifndef C2_N2K_COLS   // NARROW = 2, f64 ipc4d (16 groups): 76 KB per 128 columns
#define C2_N2K_COLS 256
#endif
constexpr int c2_cols(bool k64, int narrow) {
    return narrow == 0 ? C2_COLS_DEF : (narrow == 1 ? C2_N1_COLS : (k64 ? C2_N2K_COLS : C2_N2_COLS));
}
// coefficients of the partial K ring of the NARROW = 1 forms: what fits beside the other rings in 160 KB (C2_N1_KRN: A/B switch)
#ifndef C2_N1_KRN
#define C2_N1_KRN 9
#endif
constexpr int c2_krn(int G) {
    // x ring + f64 O1 ring + word rings of C2_N1_COLS columns, lines: what is left over / (2 rows x C2_N1_COLS x 8 bytes)
    const long used = (long)(G / 2) * C2_N1_COLS * 8 * 3 + (long)G * C2_N1_COLS * 8 * 3 + (long)C2_N1_COLS * 4 * 3 * (2 + (G + 3) / 4) +
                      (long)(C2_N1_COLS / RIP_CW + 1) * G * 2 * 8;
    const long fit = (160 * 1024 - used) / (2L * C2_N1_COLS * 8);
    return fit < 0 ? 0 : (fit > C2_N1_KRN ? C2_N1_KRN : (int)fit);
}
// waves per SIMD an instantiation is compiled for (register budget 512 / waves) and launched with
constexpr int c2_wps(int G, bool k64, int narrow) {
    return !narrow ? ((G > 8 || k64) ? 2 : 4) : ((k64 && G > 8) ? 2 : ((k64 || G > 8) ? 3 : 4));
}

// NARROW forms (the ring-dropping forms): what does not fit the 256-column form's LDS budget twice per CU drops rings, and what a
// dropped ring carried the fit role loads itself, one step ahead (a second read of lines the ingest role fetched 1.5 steps earlier):
//   NARROW = 1  a PARTIAL K ring (what the LDS left   f64 ipc4d x 6 / 8 groups: 43 KB per 128 columns (with every ring: 120 KB per 256);
//               over holds: c2_krn, KRN below)         at 384 columns 5 (8 groups) / 9 (6 groups) of the nine f64 coefficients travel
//                                                      through LDS, the fit role reads the others again: traffic 1.33 -> 1.17 x
//   NARROW = 2  no K ring, no gain / groupdq rings     16 groups: 51 KB per 128 columns; f64 ipc4d x 16 groups: 76 KB
// Round 3 ran them as 128-column workgroups, three (two) per CU: 3 (2) waves per SIMD at <= 168 (256) VGPRs.  Round 4: ONE
// workgroup per CU of 384 columns (f64 x 16 groups: 256) -- the same waves per SIMD and LDS per CU, a third (half) of the seams
// between strips, whose misaligned, twice-fetched lines were a third of these forms' excess traffic (c2_cols above): 6-8 % faster,
// same bits.  Same arithmetic as the 256-column form.  For f32 ipc4d x 8 groups these forms are slower (same 16 waves per CU).
template <int NP, int G, int START, typename KT = float, int NARROW = 0>
__global__ __launch_bounds__(2 * c2_cols(sizeof(KT) == 8, NARROW), c2_wps(G, sizeof(KT) == 8, NARROW)) void chain2_kernel(ChainArgs a, const RipPlanHeader *__restrict__ h,
                                                               const RipVariant *__restrict__ vars,
                                                               const float *__restrict__ kvals,
                                                               const RipDiff *__restrict__ diffs, double guard) {
    static_assert(G % 2 == 0 && G > 4 && G <= 16, "pairs of groups; the groupdq bytes travel packed four to a word");
    constexpr int COLS = c2_cols(sizeof(KT) == 8, NARROW);
    constexpr bool KRING = !NARROW;
    // NARROW = 1 (f64 ipc4d x 6 / 8 groups; 129.5 KB of the 160 at 384 columns): the LDS left over holds a PARTIAL K ring -- the
    // first KRN of the nine f64 coefficients travel from the ingest thread to the fit thread through LDS (two rows, like the full ring
    // of the 256-column form), the fit role reads only the other 9 - KRN planes a second time: 32 instead of 72 bytes per pixel of
    // re-read (traffic 1.33 -> 1.18 x), five loads fewer per step in flight in the fit role
    constexpr int KRN = KRING ? 9 : ((NARROW == 1 && sizeof(KT) == 8) ? c2_krn(G) : 0);
    constexpr bool WRING = NARROW < 2;   // NARROW = 2 (16 groups): gain and packed groupdq bytes do not travel through LDS either --
                                         // the fit role loads them itself, like the coefficients (51 KB: three workgroups per CU)
    constexpr int QW = (G + 3) / 4;  // words of packed group flags per pixel
    constexpr int GP = G / 2;
    // f64 ipc4d (KT = double; the reference's production writer stores f64): x = gain*phi stays f32, the Neumann iterates and
    // the division by the gain are f64 (numpy promotion, ipc_linearity.py:95-142), so the O1 ring holds doubles, one plane per
    // group: 94 KB of LDS for 8 groups, one workgroup per CU (2 waves/SIMD, up to 256 VGPRs)
    constexpr bool K64 = sizeof(KT) == 8;
    constexpr int SA = NARROW == 2 ? C2_STREAM_AUX : (NARROW == 1 ? C2_STREAM_N1 : 0);   // cache policy of the once-read arrays
    extern __shared__ __align__(16) unsigned char lds_raw[];
    constexpr int XR = 3;  // rows of the x ring
    f2 *X2 = reinterpret_cast<f2 *>(lds_raw);                       // [GP][XR][C2_COLS]  x = gain*phi, pair-interleaved
    f2 *O12 = X2 + GP * XR * C2_COLS;                               // [GP][3][C2_COLS]  first Neumann iterate
    double *O1d = reinterpret_cast<double *>(O12);                  // f64 ipc4d: [G][3][C2_COLS] instead
    uint32_t *DQ = reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(O12) +
                                                (size_t)G * 3 * C2_COLS * sizeof(KT));  // [4][C2_COLS] linearity dq of the row
    // per-pixel words that travel from the ingest thread of a column to its fit thread (3-row rings, slots as the x ring: the fit
    // thread takes row r+1's at the end of step r): the flag word, the packed groupdq bytes, the gain
    uint32_t *QS = DQ + 3 * C2_COLS;                                // [3][QW][C2_COLS] groupdq bytes of the pixel, packed
    float *GN = reinterpret_cast<float *>(QS + (WRING ? 3 * QW * C2_COLS : 0));   // [3][C2_COLS] gain of the pixel (f32)
    double *LN = reinterpret_cast<double *>(GN + (WRING ? 3 * C2_COLS : 0));       // [NLC][G][2] channel lines of this strip
    // K ring: the nine IPC coefficients of destination (row, col), loaded ONCE by the ingest thread of the column and handed to
    // its fit thread (two rows live: C of row y runs two steps before O2 of row y)
    constexpr int NLC = C2_COLS / RIP_CW + 1;                       // channels a window can touch (a window is not channel-aligned)
    f2 *KR2 = reinterpret_cast<f2 *>(LN + NLC * G * 2);               // f32 ipc4d: [2][4][C2_COLS] pairs (k0,k1)..(k6,k7)
    float *KR1 = reinterpret_cast<float *>(KR2 + 2 * 4 * C2_COLS);  //            [2][C2_COLS] k8
    double *KRd = reinterpret_cast<double *>(LN + NLC * G * 2);       // f64 ipc4d: [2][KRN][C2_COLS]

    // ChainArgs is the first kernel argument: it sits at offset 0 of the kernarg segment
    const RIP_K C2KernArgs *kargs = (const RIP_K C2KernArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    const int tid = threadIdx.x;
#ifdef C2_DBG
    const int dbg = a.dbg;  // timing experiments only (tools/gpu_checks/phase_timing.py): bits switch phases off
#else
    constexpr int dbg = 0;
#endif
    const bool fit_role = tid >= C2_COLS;
    const int col = fit_role ? tid - C2_COLS : tid;
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

    const int nstrips = C2_NSTRIPS(nx);
    const int bid = c2_xcd_block((int)blockIdx.x, (int)gridDim.x);
    // (strip, row range) of this workgroup -- in QUAD mode (the last strip, when it has at most 64 live columns) of each of its four
    // (narrow forms: two) wave columns: 64-column windows (2 + 2 halo lanes each) of the same strip that march down four different row ranges, so that
    // a strip a quarter as wide takes a quarter of the workgroups (launcher: chain2_geometry)
    const int nfull = a.geo_rows_q ? nstrips - 1 : nstrips;
    const bool quad = bid >= nfull * a.geo_nr;
    const int rows_wg = quad ? a.geo_rows_q : a.geo_rows;   // steps of every wave of the workgroup (barriers inside)
    const int strip = quad ? nfull : bid % nfull;
    const int wcol = quad ? (col & 63) : col;               // column inside the wave column's window
    const int wl = quad ? 64 : C2_COLS;                     // ... and its width
    const int R0 = min(ny, quad ? ((bid - nfull * a.geo_nr) * (C2_COLS / 64) + __builtin_amdgcn_readfirstlane(col >> 6)) * rows_wg
                                : (bid / nfull) * rows_wg);
    const int R1 = min(ny, R0 + rows_wg);
    if (bid >= nfull * a.geo_nr + a.geo_nq) return;
    const int c = strip * C2_OUTW + wcol;
    const bool col_ok = (c >= 0 && c < nx);
    const bool col_act = (c >= ax0 && c < ax1);
    const int cc = col_ok ? c : 0;
    const int ch0 = (strip * C2_OUTW) / RIP_CW;
    const int chr = cc / RIP_CW - ch0;
    for (int i = tid; i < NLC * G * 2; i += C2_THREADS) {
        const int ch = i / (G * 2), g = (i / 2) % G, w = i & 1;
        LN[i] = (ch0 + ch < nch) ? a.lines[(g * nch + ch0 + ch) * 2 + w] : 0.0;
    }
    __syncthreads();

    // Addressing: every global access is (wave-uniform 64-bit base: array + plane + row, scalar ALU) + (per-lane
    // 32-bit column offset, loop-invariant VGPR), i.e. the saddr form of global_load/store with no vector
    // address arithmetic per access.
    const unsigned cc4 = (unsigned)cc * 4u, cc2 = (unsigned)cc * 2u, cc1 = (unsigned)cc;
    unsigned cx4[3];  // byte offset of the clamped source column c - dx, index dx + 1
    unsigned colmask = 0;  // bit k: source column of term k is in the active box (and so is c)
    {
        bool okx[3];
#pragma unroll
        for (int dxi = 0; dxi < 3; ++dxi) {
            const int sx = c - (dxi - 1);
            okx[dxi] = sx >= ax0 && sx < ax1;
            cx4[dxi] = (unsigned)min(max(sx, 0), nx - 1) * 4u;
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int dx = (k == 3 || k == 5 || k == 7) ? 1 : (k == 4 || k == 6 || k == 8) ? -1 : 0;
            if (okx[dx + 1]) colmask |= 1u << k;
        }
        if (!col_act) colmask = 0;
    }
    const unsigned lane_mask = fit_role ? ((wcol >= 2 && wcol < wl - 2) ? colmask : 0u)
                                        : ((wcol >= 1 && wcol < wl - 1) ? colmask : 0u);
    const unsigned row4 = (unsigned)nx * 4u;  // row pitch of an f32/u32 plane; offsets within a plane fit 32 bits
    // rows this (strip, row range) cell really uses: R0-2 .. R1+1 (two halo rows on each side).  The straight-line loads of the
    // warm-up and drain steps are clamped INTO that band, so that they touch lines the cell reads anyway instead of 3-5 rows of
    // its neighbours' (each such row costs HBM traffic and buys nothing)
    const int ylo = max(R0 - 2, 0), yhi = min(R1 + 1, ny - 1);

    // coefficient loader: raw loads at clamped source positions + validity mask for destination (y, c);
    // `want` is wave-uniform.  Planes are walked in memory order (plane = 3*(1+dy) + (1+dx)).
    auto load_k = [&](const void *kern_base, int y, bool want, f2 (&kk2)[5]) -> unsigned {
        if (dbg & 128) {
#pragma unroll
            for (int k = 0; k < 5; ++k) kk2[k] = f2{(k == 0) ? 1.0f : 0.001f, 0.001f};
            return want ? lane_mask : 0u;
        }
        unsigned rowoff[3];
        bool rok[3];
#pragma unroll
        for (int dyi = 0; dyi < 3; ++dyi) {
            const int sy = y - (dyi - 1);
            rok[dyi] = sy >= ay0 && sy < ay1;
            rowoff[dyi] = (unsigned)min(max(sy, ylo), yhi) * row4;   // rows outside the range's own are never used: keep to lines it reads anyway
        }
        // terms by source row: dy = -1 -> k in {2, 7, 8}, dy = 0 -> {0, 3, 4}, dy = +1 -> {1, 5, 6}
        const unsigned rowbits = (rok[0] ? 0x184u : 0u) | (rok[1] ? 0x019u : 0u) | (rok[2] ? 0x062u : 0u);
        const __amdgpu_buffer_rsrc_t kr = c2_rsrc(kern_base);
        unsigned pofs = 0;  // plane offset, planes walked in memory order (plane = 3*(1+dy) + (1+dx))
#pragma unroll
        for (int p = 0; p < 9; ++p) {
            const int dy = p / 3 - 1, dx = p % 3 - 1;
            // term index of (dy, dx) in the reference's order: 0 centre, 1 (1,0), 2 (-1,0), 3 (0,1), 4 (0,-1), 5 (1,1),
            // 6 (1,-1), 7 (-1,1), 8 (-1,-1)
            const int k = (dy == 0) ? (dx == 0 ? 0 : dx == 1 ? 3 : 4) : (dy == 1) ? (dx == 0 ? 1 : dx == 1 ? 5 : 6)
                                                                                  : (dx == 0 ? 2 : dx == 1 ? 7 : 8);
            const float kv_ = c2_ld_f32(kr, cx4[dx + 1], pofs + rowoff[dy + 1]);
            if (k & 1)
                kk2[k / 2].y = kv_;
            else
                kk2[k / 2].x = kv_;
            pofs += pl4;
        }
        const unsigned um = (want && y >= ay0 && y < ay1) ? rowbits : 0u;
        return lane_mask & um;
    };

    // f64 coefficients: the same walk with 8-byte loads, nine scalars in the reference's term order
    // (kmin: terms below it are not loaded -- the fit role of a form with a partial K ring gets them through LDS)
    auto load_kd = [&](const void *kern_base, int y, bool want, double (&kk)[9], auto kmin_c) -> unsigned {
        constexpr int KMIN = decltype(kmin_c)::value;
        unsigned rowoff[3];
        bool rok[3];
#pragma unroll
        for (int dyi = 0; dyi < 3; ++dyi) {
            const int sy = y - (dyi - 1);
            rok[dyi] = sy >= ay0 && sy < ay1;
            rowoff[dyi] = (unsigned)min(max(sy, ylo), yhi) * (row4 * 2u);
        }
        const unsigned rowbits = (rok[0] ? 0x184u : 0u) | (rok[1] ? 0x019u : 0u) | (rok[2] ? 0x062u : 0u);
        const __amdgpu_buffer_rsrc_t kr = c2_rsrc(kern_base);
        unsigned pofs = 0;
#pragma unroll
        for (int p = 0; p < 9; ++p) {
            const int dy = p / 3 - 1, dx = p % 3 - 1;
            const int k = (dy == 0) ? (dx == 0 ? 0 : dx == 1 ? 3 : 4) : (dy == 1) ? (dx == 0 ? 1 : dx == 1 ? 5 : 6)
                                                                                  : (dx == 0 ? 2 : dx == 1 ? 7 : 8);
            if (k >= KMIN) kk[k] = c2_ld_f64(kr, cx4[dx + 1] * 2u, pofs + rowoff[dy + 1]);
            pofs += pl4 * 2u;
        }
        const unsigned um = (want && y >= ay0 && y < ay1) ? rowbits : 0u;
        return lane_mask & um;
    };
    // validity mask of destination (y, c) alone, for the role that receives the coefficients through the K ring
    auto k_valid = [&](int y) -> unsigned {
        const bool r0 = (y + 1) >= ay0 && (y + 1) < ay1, r1 = y >= ay0 && y < ay1, r2 = (y - 1) >= ay0 && (y - 1) < ay1;
        const unsigned rowbits = (r0 ? 0x184u : 0u) | (r1 ? 0x019u : 0u) | (r2 ? 0x062u : 0u);
        return r1 ? (lane_mask & rowbits) : 0u;
    };
#ifdef CH_STAMP
    unsigned long long st_[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tl_ = __builtin_amdgcn_s_memtime();
#endif
    if (!fit_role) {
        // =========================================================================== ingest waves
        // plane p of the calibration slab at row yl: scalar offset p*pl4 + yl*row4; groups likewise in their arrays
        auto fetch_groups = [&](const RIP_K ChainArgs *ka, int y, int g0, int g1, RowRegs<NP, G> &rr) {
            if ((dbg & 64) && y > R0 - 2) return;   // timing experiment: every row works on the first row's (valid) values
            __builtin_amdgcn_sched_barrier(0);
            const unsigned yl = (unsigned)min(max(y, ylo), yhi);
            const __amdgpu_buffer_rsrc_t rs = c2_rsrc(ka->data), rq = c2_rsrc(ka->gdq), rd = c2_rsrc(ka->dark_data),
                                         rb = c2_rsrc(ka->bias);
            unsigned o4 = yl * row4 + (unsigned)g0 * pl4, o2 = yl * (row4 >> 1) + (unsigned)g0 * (pl4 >> 1),
                     o1 = yl * (row4 >> 2) + (unsigned)g0 * npix;
#pragma unroll
            for (int g = g0; g < g1; ++g) {
                rr.S[g] = c2_ld_u16<SA>(rs, cc2, o2);
                rr.q[g] = c2_ld_u8<WRING ? SA : 0>(rq, cc1, o1);   // (NARROW = 2: the fit role reads these bytes again)
                rr.dk[g] = c2_ld_f32<SA>(rd, cc4, o4);
                rr.bs[g] = c2_ld_f32<SA>(rb, cc4, o4);
                o4 += pl4;
                o2 += pl4 >> 1;
                o1 += npix;
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        // planes i0..i1-1 of [cf[0..NP-1], Smin, Smax, Sref, dq, gain]
        auto fetch_coefs = [&](const RIP_K ChainArgs *ka, int y, int i0, int i1, RowRegs<NP, G> &rr) {
            if ((dbg & 64) && y > R0 - 2) return;
            __builtin_amdgcn_sched_barrier(0);
            const unsigned yl = (unsigned)min(max(y, ylo), yhi);
            const __amdgpu_buffer_rsrc_t rp = c2_rsrc(ka->planes);
            unsigned o4 = yl * row4 + (unsigned)i0 * pl4;
#pragma unroll
            for (int i = i0; i < i1; ++i) {
                if (i < NP)
                    rr.cf[i] = c2_ld_f32<SA>(rp, cc4, o4);
                else if (i == NP)
                    rr.smin = c2_ld_f32<SA>(rp, cc4, o4);
                else if (i == NP + 1)
                    rr.smax = c2_ld_f32<SA>(rp, cc4, o4);
                else if (i == NP + 2)
                    rr.sref = c2_ld_f32<SA>(rp, cc4, o4);
                else if (i == NP + 3)   // the flag word: linearity dq merged with what the finish step ORs into pixeldq (RipCal)
                    rr.dq = c2_ld_u32<SA>(rp, cc4, yl * row4 + (unsigned)(NP + ka->merged_dq) * pl4);
                else
                    rr.gain = c2_ld_f32<WRING ? SA : 0>(rp, cc4, o4);
                o4 += pl4;
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        auto fetch_row = [&](int y, RowRegs<NP, G> &rr) {  // prologue: the whole first row at once
            const RIP_K ChainArgs *ka = &kargs->a;
            fetch_coefs(ka, y, 0, NP + 5, rr);
            fetch_groups(ka, y, 0, G, rr);
        };
        constexpr int NCO = NP + 5;                 // coefficient-type planes per pixel
        constexpr int CO_STEP = (NCO + GP - 1) / GP;  // issued per pair of C
        RowRegs<NP, G> rr;
        fetch_row(R0 - 2, rr);
        int so_c = (R0 - 5 + 2 + 3000) % 3;  // O1 ring slot of row yc = r + 2
        double rcn[G];  // row corrections of the row the next step ingests
        {
            const RIP_K double *rt = rip_k(kargs->a.rowcorr_t) + (size_t)min(max(R0 - 2, 0), ny - 1) * G;
#pragma unroll
            for (int g = 0; g < G; ++g) rcn[g] = rt[g];
        }
        for (int r = R0 - 5; r <= R0 + rows_wg; ++r, so_c = (so_c == 2) ? 0 : so_c + 1) {
            const RIP_K ChainArgs *ka = &c2_args(kargs)->a;  // S1 copy of the argument block
            const int yi = r + 3, yc = r + 2;
            const bool do_a = (yi >= R0 - 2) && (yi <= R1 + 1);
            const bool do_c = (yc >= R0 - 1) && (yc <= R1);
            // ---- S1: A (linearity of row yi from rr), then the loads S2 consumes
            // per-row reference-pixel correction of the G groups: wave-uniform, scalar loads (constant address space)
            double rc[G];  // loaded one step ahead: one wide scalar load from the row-major copy of the table
#pragma unroll
            for (int g = 0; g < G; ++g) rc[g] = rcn[g];
            {
                const RIP_K double *rt = rip_k(ka->rowcorr_t) + (size_t)min(max(yi + 1, 0), ny - 1) * G;
#pragma unroll
                for (int g = 0; g < G; ++g) rcn[g] = rt[g];
            }
            const bool a_full = do_a && yi >= 0 && yi < ny;  // wave-uniform
            // A: two pairs of groups at a time -- reference-pixel/bias arithmetic and z of both pairs, then their two
            // Legendre recurrences interleaved (independent chains), then the raw loads of the same groups of the next
            // row.  The loads are issued UNCONDITIONALLY between the blocks (the wait-count pass is path-insensitive: a
            // load that exists on one side of a branch only forces vmcnt(0) at later uses).  All lanes compute (lanes
            // beyond the frame edge work on the clamped column and store zeros).
            const int xslot = (so_c == 2) ? 0 : so_c + 1;    // 3-row rings (x and the per-pixel words): row yi = r + 3 takes the slot of row r
            f2 *xs = X2 + xslot * C2_COLS + col;
            const bool act = col_act && yi >= ay0 && yi < ay1;
            uint32_t dq = rr.dq;
            uint32_t w[QW];  // the pixel's groupdq bytes, packed
#pragma unroll
            for (int i = 0; i < QW; ++i) w[i] = 0;
            const float smin = rr.smin;
            const float span = rr.smax - smin;
            const bool fastdiv = __all(rcp_safe(span));
            const float rspan = rip_rcp_mid(span);  // used only when every lane passes rcp_safe (2^-59.8 .. 2^59.8)
            const double yd = (double)yi;
            // pairs per block (their recurrences interleave): 2 at 128 registers; per instantiation by same-box A/B at the wide forms
            // (profiles/r04_ab_runs.txt): 16 groups at 168 registers 1 (1.697 against 1.710 ms; 4 spills: 3.46), f64 ipc4d x 16 groups at
            // 256 registers 4 (2.24 against 2.27)
            constexpr int PBW = (C2_PBW > 0) ? C2_PBW : (NARROW == 2 ? (K64 ? 4 : 1) : 2);
            constexpr int PB = ((K64 || G > 8) && GP % PBW == 0) ? PBW : (GP % 2 == 0) ? 2 : 1;
#pragma unroll
            for (int pb = 0; pb < GP; pb += PB) {
                f2 zz[PB], SS[PB];
                bool any_ex = false;
                if (a_full && !(dbg & 32)) {
                    f2 tt[PB], quo[PB];
#pragma unroll
                    for (int b = 0; b < PB; ++b) {
                        const int p = pb + b;
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
                            w[g / 4] |= (rr.q[g] & 0xffu) << (8 * (g & 3));
                        }
                        SS[b] = f2{Sv[0], Sv[1]};
                        const f2 t = SS[b] - f2{smin, smin};
                        tt[b] = t * 2.0f;
                    }
                    if (fastdiv) {  // one block: the PB reciprocal-division chains interleave
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
                } else {
#pragma unroll
                    for (int b = 0; b < PB; ++b) zz[b] = SS[b] = f2{0.0f, 0.0f};
                }
                // raw values of these groups of the next row (this row's are consumed)
                fetch_groups(ka, r + 4, 2 * pb, 2 * (pb + PB), rr);
                if (a_full && (dbg & 16)) {
#pragma unroll
                    for (int b = 0; b < PB; ++b) xs[(pb + b) * XR * C2_COLS] = zz[b] + f2{1000.0f, 1100.0f};
                } else if (a_full) {
                    const bool slow = __any(any_ex);
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
                        const f2 fb = SS[b] - f2{rr.sref, rr.sref};
                        float vout[2];
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const int g = 2 * p + e;
                            vout[e] = ((dq & bad) == 0) ? (e ? phi[b].y : phi[b].x) : (e ? fb.y : fb.x);
                            const bool first = (g == 0) && a.do_not_flag_first;
                            const uint32_t qg = w[g / 4] >> (8 * (g & 3));
                            if (!first && ex[b][e] && (qg & DQ_SATURATED) == 0) dq |= DQ_NO_LIN_CORR;
                        }
                        f2 xv = {vout[0], vout[1]};
                        if (act) xv = xv * rr.gain;
                        xs[p * XR * C2_COLS] = col_ok ? xv : f2{0.0f, 0.0f};
                    }
                } else if (do_a) {
#pragma unroll
                    for (int b = 0; b < PB; ++b) xs[(pb + b) * XR * C2_COLS] = f2{0.0f, 0.0f};
                }
            }
            if (do_a) {
                const bool keep = a_full && col_ok;
                DQ[xslot * C2_COLS + col] = keep ? dq : 0u;
                if constexpr (WRING) {
#pragma unroll
                    for (int i = 0; i < QW; ++i) QS[(xslot * QW + i) * C2_COLS + col] = keep ? w[i] : 0u;
                    GN[xslot * C2_COLS + col] = rr.gain;
                }
            }
            // IPC coefficients of row yc, consumed by C after the barrier; issued here so that their registers are not live
            // during A (requested at the top of the step instead: 0.949 against 0.945 ms, the latency is not on the critical path)
            f2 kC[5];
            double kCd[9];
            kC[4].y = 0.0f;
            unsigned vC;
            if constexpr (K64)
                vC = load_kd(ka->kern, yc, do_c && wcol >= 1 && wcol < wl - 1, kCd, C2Int<0>{});
            else
                vC = load_k(ka->kern, yc, do_c && wcol >= 1 && wcol < wl - 1, kC);
            CH_T(2)
            C2_SYNC();
            CH_T(3)
            const RIP_K ChainArgs *kb2 = &c2_args(kargs)->a;  // S2 copy
            // ---- S2: issue the raw loads of row r+4 (consumed in S1 of the next step), then C of row yc
            CH_T(4)
            {
                // every lane evaluates (lanes without valid terms produce values nobody reads); the coefficient planes of
                // the next row are requested between the pairs, unconditionally (see above)
                const bool all = __all(vC == 0x1ffu || vC == 0u);
                const int so = so_c;  // slot of row yc in both 3-row rings
                const int sm = (so == 0) ? 2 : so - 1, s0 = so, sp = (so == 2) ? 0 : so + 1;
                // hand the coefficients of destination row yc to the fit thread of this column (O2 of row yc, two steps on)
                if constexpr (KRN > 0) {
                    const int ks = yc & 1;
                    if constexpr (K64) {
#pragma unroll
                        for (int k = 0; k < KRN; ++k) KRd[(ks * KRN + k) * C2_COLS + col] = kCd[k];
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) KR2[(ks * 4 + i) * C2_COLS + col] = kC[i];
                        KR1[ks * C2_COLS + col] = kC[4].x;
                    }
                }
                if constexpr (K64) {
                    // two pairs (four groups) in lockstep: their 18 ring reads first, then four interleaved f64 chains
                    constexpr bool BIG = !NARROW || (K64 && G > 8);   // 256-register budget (narrow f64 x 8 groups: 168 -- one pair at a time)
                    constexpr int PBC = (C2_PBC > 0) ? C2_PBC : ((GP % 2 == 0 && BIG) ? 2 : 1);   // (f64 x 16 groups with C2_PBW 4: 2.23 / 2.25 ms for 2 / 1)
#pragma unroll
                    for (int p0 = 0; p0 < GP; p0 += PBC) {
#pragma unroll
                        for (int q = 0; q < PBC; ++q)
                            fetch_coefs(kb2, r + 4, (p0 + q) * CO_STEP, ((p0 + q) * CO_STEP + CO_STEP < NCO) ? (p0 + q) * CO_STEP + CO_STEP : NCO, rr);
                        if (do_c) {
                            float v[2 * PBC][9];
#pragma unroll
                            for (int q = 0; q < PBC; ++q) {
                                const f2 *xb = X2 + (p0 + q) * XR * C2_COLS + col;
                                const f2 *xm_ = xb + sm * C2_COLS, *x0_ = xb + s0 * C2_COLS, *xp_ = xb + sp * C2_COLS;
                                const f2 tt[9] = {x0_[0], xm_[0], xp_[0], x0_[-1], x0_[1], xm_[-1], xm_[1], xp_[-1], xp_[1]};
#pragma unroll
                                for (int k = 0; k < 9; ++k) v[2 * q][k] = tt[k].x, v[2 * q + 1][k] = tt[k].y;
                            }
                            double f[2 * PBC];
                            if (all)
                                c2_ipc9_batch<true, 2 * PBC>(v, kCd, vC, f);
                            else
                                c2_ipc9_batch<false, 2 * PBC>(v, kCd, vC, f);
#pragma unroll
                            for (int b = 0; b < 2 * PBC; ++b) {
                                const float xc = v[b][0];
                                O1d[((2 * p0 + b) * 3 + so) * C2_COLS + col] = (double)(xc + xc) - f[b];
                            }
                        }
                    }
                } else if (do_c && all && !(dbg & 1)) {
                    // interior wave: one straight-line block (see O2)
#pragma unroll
                    for (int i = 0; i < 5; ++i) asm volatile("" : "+v"(kC[i]));
#pragma unroll
                    for (int p0 = 0; p0 < GP; ++p0) {
                        fetch_coefs(kb2, r + 4, p0 * CO_STEP, (p0 * CO_STEP + CO_STEP < NCO) ? p0 * CO_STEP + CO_STEP : NCO, rr);
                        const f2 *xb = X2 + p0 * XR * C2_COLS;
                        const f2 *xm[1] = {xb + sm * C2_COLS}, *x0[1] = {xb + s0 * C2_COLS}, *xp[1] = {xb + sp * C2_COLS};
                        f2 f[1], xc[1];
                        fwd_rows_batch<1, true>(xm, x0, xp, col, kC, vC, f, xc);
                        O12[(p0 * 3 + so) * C2_COLS + col] = (xc[0] + xc[0]) - f[0];
                    }
                } else {
#pragma unroll
                    for (int p0 = 0; p0 < GP; ++p0) {
                        fetch_coefs(kb2, r + 4, p0 * CO_STEP, (p0 * CO_STEP + CO_STEP < NCO) ? p0 * CO_STEP + CO_STEP : NCO, rr);
                        if (do_c && !(dbg & 1)) {
                            const f2 *xb = X2 + p0 * XR * C2_COLS;
                            const f2 *xm[1] = {xb + sm * C2_COLS}, *x0[1] = {xb + s0 * C2_COLS}, *xp[1] = {xb + sp * C2_COLS};
                            f2 f[1], xc[1];
                            fwd_rows_batch<1, false>(xm, x0, xp, col, kC, vC, f, xc);
                            O12[(p0 * 3 + so) * C2_COLS + col] = (xc[0] + xc[0]) - f[0];
                        }
                    }
                }
                if (GP * CO_STEP < NCO) fetch_coefs(kb2, r + 4, GP * CO_STEP, NCO, rr);
            }
            CH_T(5)
            CH_T(6)
            C2_SYNC();
            CH_T(7)
        }
    } else {
        // =========================================================================== fit waves
        f2 xnext[GP];  // x of (row r + 1, own column), read from the ring at the end of the step before (the ring holds 3 rows)
#pragma unroll
        for (int p0 = 0; p0 < GP; ++p0) xnext[p0] = f2{0.0f, 0.0f};
        const RipVariant v0 = rip_load_variant(vars, 0);
        const RipFitConst fc0 = rip_fit_const(h);
        constexpr int start = START;  // first group of the fit (exclude_first)
        float gain_next = 1.0f;   // gain, flag word and packed groupdq bytes of (row r + 1, own column): from the rings, like xnext
        uint32_t dq_next = 0, qw_next[QW];
#pragma unroll
        for (int i = 0; i < QW; ++i) qw_next[i] = 0;
        int o0_r = (R0 - 5 + 3000) % 3;  // O1 ring slot of row r
        uint32_t qb_next[WRING ? 1 : G];   // NARROW = 2: the raw groupdq bytes of the next step's pixel
#pragma unroll
        for (int g = 0; g < (WRING ? 1 : G); ++g) qb_next[g] = 0;
        f2 kn2[5];      // (C2_KFIT_EARLY == 2: those of the row after)
        double kn2_d[9];
#pragma unroll
        for (int i = 0; i < 5; ++i) kn2[i] = f2{0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < 9; ++k) kn2_d[k] = 0.0;
        f2 kn[5];       // narrow form: the coefficients of the next step's row (C2_KFIT_EARLY)
        double kn_d[9];
#pragma unroll
        for (int i = 0; i < 5; ++i) kn[i] = f2{0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < 9; ++k) kn_d[k] = 0.0;
        for (int r = R0 - 5; r <= R0 + rows_wg; ++r, o0_r = (o0_r == 2) ? 0 : o0_r + 1) {
            const bool emit = (r >= R0) && (r < R1) && col_ok && (wcol >= 2 || c < 2) && (wcol < wl - 2 || c >= nx - 2);
            const RIP_K C2KernArgs *kf = c2_args(kargs);  // S1 copy of the argument block
            const unsigned rc_ = (unsigned)min(max(r, R0), yhi);   // (rows before R0 are warm-up steps: nothing is emitted there)
            const unsigned pe = rc_ * (unsigned)nx + cc1;
            // ---- S1: read noise of the pixel (used by the fit), then the second IPC iterate of row r
            const __amdgpu_buffer_rsrc_t rpl = c2_rsrc(kf->a.planes);
            const unsigned t_row = rc_ * row4;  // byte offset of row r in an f32 plane (uniform)
            const unsigned t_ld = (dbg & 256) ? 0u : t_row;   // timing experiment: the fit role's loads all hit row 0 (cached)
            const float e_read = c2_ld_f32<SA>(rpl, cc4, (unsigned)(NP + 5) * pl4 + t_ld);
            const float e_gain = gain_next;
            // calibration planes of the tail (finish) of the same pixel, consumed after the barrier
            const size_t t_row4 = (size_t)t_row;
            const size_t pe_row = (size_t)(rc_ * (unsigned)nx);   // element offset of row r
            const float e_dark = c2_ld_f32<SA>(rpl, cc4, (unsigned)(NP + 6) * pl4 + t_ld);
            // (flat flags and dark dq arrive with the linearity dq of the pixel: ChainArgs::merged_dq)
            const uint32_t e_pdq = c2_ld_u32<SA>(c2_rsrc(kf->a.pdq), cc4, t_ld);
            // flat / dark_dq == null: read the first slab plane instead (value unused), keeps the loads in one block
            const float e_flat_raw = c2_ld_f32<SA>(c2_rsrc(kf->a.flat ? (const void *)kf->a.flat : (const void *)kf->a.planes), cc4, t_ld);
            const float e_flat = kf->a.flat ? e_flat_raw : 1.0f;
            float d[G];
            f2 dpair[GP];
            uint32_t qw[QW];  // the pixel's groupdq bytes, packed
#pragma unroll
            for (int i = 0; i < QW; ++i) qw[i] = 0;
            uint32_t lin_dq = 0;  // linearity dq of the pixel
            RipFitState fs;
            const bool act = emit && col_act && r >= ay0 && r < ay1;
            CH_T(0)
            C2_DRAIN()
            CH_T(1)
            // (round 4, the wide ring-dropping forms, same box: 0 / 1 / 2 = 16 groups 1.704 / 1.770 / 1.751 ms per ramp, f64 x 16
            // groups 2.251 / 2.278 / 2.302, f64 x 8 groups 1.138 / 1.145 / 1.153)
            constexpr int BAR = (C2_BAR >= 0) ? C2_BAR : (NARROW ? 0 : (K64 ? 2 : (G > 8 ? 1 : 0)));
            // The tail of pixel (r, c) in three parts; the half-step barrier falls between two of them (C2_BAR: everything after O2 is
            // register-only in a fit thread, so the barrier sits where both roles take about the same time in both halves).
            float s = 0.0f, er = 0.0f, ep = 0.0f;
            uint32_t jmask = 0, pdq = 0;
            // second half of the fit: exact pass where needed, jump mask; then the saturated refits
            auto part_fb = [&](const RIP_K C2KernArgs *kx) {
                if (kx->a.cube_out) {
#pragma unroll
                    for (int g = 0; g < G; ++g) kx->a.cube_out[(unsigned)g * npix + pe] = d[g];
                }
                uint32_t qor = 0;
#pragma unroll
                for (int i = 0; i < QW; ++i) qor |= qw[i];
                const bool anysat = (qor & 0x02020202u) != 0u;
                const bool unsat = ((qw[(G - 1) / 4] >> (8 * ((G - 1) & 3))) & DQ_SATURATED) == 0;
                if (dbg & 4) {
                    s = d[0], er = e_read, ep = e_gain;
                } else {
                    fit_full_pk_b<G>(dpair, kx->h, fc0, kx->a.dense, kx->kvals + v0.k_ofs, kx->diffs + v0.diff_ofs, unsat && act,
                                     fs, jmask);
                    s = fs.s, er = fs.er, ep = fs.ep;
                    if (__any(anysat)) {
                        uint32_t qe[G];
#pragma unroll
                        for (int g = 0; g < G; ++g) qe[g] = (qw[g / 4] >> (8 * (g & 3))) & 0xffu;
                        trunc_layers<G, G - 1>(d, qe, kx->h, kx->vars, kx->kvals, kx->diffs, e_gain, e_read, act, kx->guard, s, er, ep,
                                               jmask);
                    }
                }
            };
            // T, first part: flag propagation (fitting.py:339-353) and the stores of the group flags
            auto part_flags = [&](const RIP_K C2KernArgs *kx) {
                if (dbg & 8) return;
                uint8_t *gq = (kx->a.gdq_out && !(dbg & 512)) ? kx->a.gdq_out + pe_row : nullptr;
                pdq = propagate_flags_packed<G>(qw, jmask, start, e_pdq | lin_dq, gq, npix, c2_opaque(cc1));
            };
            // T, second part: finish and the stores of the four planes
            auto part_finish = [&](const RIP_K C2KernArgs *kx) {
                if (dbg & 8) return;
                if (kx->a.finish) {
                    // gen_cal_image.py:458-475, 213-229, 607-629.  One wave vote selects the straight-line form built
                    // from the short exact operations (rip_rcp_mid, rip_sqrt_mid, sqrt(x*x) = x: tools/gpu_checks/
                    // fpcheck.hip); every intermediate then lies in their validated range 2^-100 .. 2^100 or is +0.
                    const float sd = (act && kx->a.dark_rate) ? s - e_dark : s;
                    const bool lean = kx->a.flat && __all(act && rip_mid36(sd) && (er == 0.0f || rip_mid36(er)) &&
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
                        if (act && kx->a.dark_rate) s = s - e_dark;
                        float ep2 = sqrtf(vp);
                        const float e2 = err * err;
                        const float p2 = ep2 * ep2;
                        float er2 = sqrtf(clip_lo<float>(e2 - p2, 0.0f));
                        if (kx->a.flat) {
                            s = s / e_flat;
                            er2 = er2 / e_flat;
                            ep2 = ep2 / e_flat;
                        }
                        er = er2;
                        ep = ep2;
                    }
                }
                const unsigned w4 = c2_opaque(cc4);
                // (dbg & 512: timing experiment without the plane stores; the test keeps the four values live)
                if (!(dbg & 512) || (s + er + ep == 12345.678f && pdq == 0xdeadbeefu)) {
                    *reinterpret_cast<float *>(reinterpret_cast<char *>(kx->a.slope) + t_row4 + w4) = s;
                    *reinterpret_cast<float *>(reinterpret_cast<char *>(kx->a.err_read) + t_row4 + w4) = er;
                    *reinterpret_cast<float *>(reinterpret_cast<char *>(kx->a.err_poisson) + t_row4 + w4) = ep;
                    *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(kx->a.pdq_out) + t_row4 + w4) = pdq;
                }
            };
            // narrow form: the fit role loads the nine coefficients of destination (r, col) itself (all lanes: clamped addresses)
            f2 kF[5];
            double kFd[9];
            kF[4].y = 0.0f;
            if constexpr (!KRING) {
#if C2_KFIT_EARLY   // requested at the end of the step before (they land across the barrier); 2: two steps before
#pragma unroll
                for (int k = 0; k < 9; ++k) kFd[k] = kn_d[k];
#pragma unroll
                for (int i = 0; i < 5; ++i) kF[i] = kn[i];
#if C2_KFIT_EARLY == 2
#pragma unroll
                for (int k = 0; k < 9; ++k) kn_d[k] = kn2_d[k];
#pragma unroll
                for (int i = 0; i < 5; ++i) kn[i] = kn2[i];
#endif
#else
                if constexpr (K64)
                    (void)load_kd(kf->a.kern, r, true, kFd, C2Int<KRN>{});
                else
                    (void)load_k(kf->a.kern, r, true, kF);
#endif
            }
            if (emit) {
                // the nine coefficients of destination (r, col) from the ingest thread of this column
                if constexpr (KRN > 0) {
                    const int ks = r & 1;
                    if constexpr (K64) {
#pragma unroll
                        for (int k = 0; k < KRN; ++k) kFd[k] = KRd[(ks * KRN + k) * C2_COLS + col];
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) kF[i] = KR2[(ks * 4 + i) * C2_COLS + col];
                        kF[4].x = KR1[ks * C2_COLS + col];
                    }
                }
                const unsigned vF = k_valid(r);
#pragma unroll
                for (int i = 0; i < QW; ++i) qw[i] = qw_next[i];
                if constexpr (!WRING) {   // packed as the ingest role packs them
#pragma unroll
                    for (int i = 0; i < QW; ++i) qw[i] = 0;
#pragma unroll
                    for (int g = 0; g < G; ++g) qw[g / 4] |= (qb_next[g] & 0xffu) << (8 * (g & 3));
                }
                lin_dq = dq_next;
                const bool fastdiv = __all(rcp_safe(e_gain) || !act);
                const float rgain = rip_rcp_mid(e_gain);
                const bool all = __all(vF == 0x1ffu || !act);
                const int o0_ = o0_r, om_ = (o0_r == 0) ? 2 : o0_r - 1, op_ = (o0_r == 2) ? 0 : o0_r + 1;
                constexpr int NB = (GP % 2 == 0) ? 2 : 1;
                if constexpr (K64) {
                    // f64 iterate: (O1 + x) - fwd(O1) and the division by the gain in f64, one rounding to f32 at the end; C2_NBO groups in
                    // lockstep (their ring reads, then interleaved chains), the divisions together at the end.  Lanes that are not
                    // active (border pixels) evaluate on whatever the rings hold there and keep x.
                    constexpr int NBO = (G > 8 && C2_NBO == 2) ? 4 : ((G % C2_NBO == 0) ? C2_NBO : 2);   // (16 groups: 256 registers, four in lockstep)
                    double o2v[G];
#pragma unroll
                    for (int gb = 0; gb < G; gb += NBO) {
                        double v[NBO][9];
#pragma unroll
                        for (int b = 0; b < NBO; ++b) {
                            const double *ob = O1d + (size_t)(gb + b) * 3 * C2_COLS + col;
                            const double *om = ob + om_ * C2_COLS, *o0 = ob + o0_ * C2_COLS, *op = ob + op_ * C2_COLS;
                            v[b][0] = o0[0], v[b][1] = om[0], v[b][2] = op[0], v[b][3] = o0[-1], v[b][4] = o0[1], v[b][5] = om[-1],
                            v[b][6] = om[1], v[b][7] = op[-1], v[b][8] = op[1];
                        }
                        double f[NBO];
                        if (all)
                            c2_ipc9_batch<true, NBO>(v, kFd, vF, f);
                        else
                            c2_ipc9_batch<false, NBO>(v, kFd, vF, f);
#pragma unroll
                        for (int b = 0; b < NBO; ++b) {
                            const float xc = ((gb + b) & 1) ? xnext[(gb + b) / 2].y : xnext[(gb + b) / 2].x;
                            o2v[gb + b] = (v[b][0] + (double)xc) - f[b];
                        }
                    }
                    float qf[G];
                    c2_div64_shared<G>(o2v, e_gain, act, qf);
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        const float xc = (g & 1) ? xnext[g / 2].y : xnext[g / 2].x;
                        d[g] = act ? qf[g] : xc;
                    }
#pragma unroll
                    for (int p0 = 0; p0 < GP; ++p0) dpair[p0] = f2{d[2 * p0], d[2 * p0 + 1]};
                } else if (all && fastdiv && __all(act) && !(dbg & 2)) {
                    // interior wave: one straight-line block (the coefficient pairs stay 64-bit registers whose halves the
                    // packed multiplies broadcast through op_sel)
#pragma unroll
                    for (int i = 0; i < 5; ++i) asm volatile("" : "+v"(kF[i]));
#pragma unroll
                    for (int p0 = 0; p0 < GP; p0 += NB) {
                        f2 xc[NB];
                        const f2 *om[NB], *o0[NB], *op[NB];
#pragma unroll
                        for (int b = 0; b < NB; ++b) {
                            xc[b] = xnext[p0 + b];
                            const f2 *ob = O12 + (p0 + b) * 3 * C2_COLS;
                            om[b] = ob + om_ * C2_COLS, o0[b] = ob + o0_ * C2_COLS, op[b] = ob + op_ * C2_COLS;
                        }
                        f2 f[NB], oc[NB];
                        fwd_rows_batch<NB, true>(om, o0, op, col, kF, vF, f, oc);
#pragma unroll
                        for (int b = 0; b < NB; ++b) {
                            const f2 val = div_rcp2((oc[b] + xc[b]) - f[b], e_gain, rgain);
                            d[2 * (p0 + b)] = val.x;
                            d[2 * (p0 + b) + 1] = val.y;
                            dpair[p0 + b] = val;
                        }
                    }
                } else {
                    constexpr int NBG = 1;  // boundary waves (rare): one pair at a time keeps this path's register demand low
#pragma unroll
                    for (int p0 = 0; p0 < GP; p0 += NBG) {
                        f2 xc[NBG], val[NBG];
#pragma unroll
                        for (int b = 0; b < NBG; ++b) val[b] = xc[b] = xnext[p0 + b];
                        if (act && !(dbg & 2)) {
                            const f2 *om[NBG], *o0[NBG], *op[NBG];
#pragma unroll
                            for (int b = 0; b < NBG; ++b) {
                                const f2 *ob = O12 + (p0 + b) * 3 * C2_COLS;
                                om[b] = ob + om_ * C2_COLS, o0[b] = ob + o0_ * C2_COLS, op[b] = ob + op_ * C2_COLS;
                            }
                            f2 f[NBG], oc[NBG];
                            if (all)
                                fwd_rows_batch<NBG, true>(om, o0, op, col, kF, vF, f, oc);
                            else
                                fwd_rows_batch<NBG, false>(om, o0, op, col, kF, vF, f, oc);
                            f2 o2[NBG];
#pragma unroll
                            for (int b = 0; b < NBG; ++b) o2[b] = (oc[b] + xc[b]) - f[b];
                            if (fastdiv) {
#pragma unroll
                                for (int b = 0; b < NBG; ++b) val[b] = div_rcp2(o2[b], e_gain, rgain);
                            } else {
#pragma unroll
                                for (int b = 0; b < NBG; ++b) val[b] = f2{o2[b].x / e_gain, o2[b].y / e_gain};
                            }
                        }
#pragma unroll
                        for (int b = 0; b < NBG; ++b) {
                            d[2 * (p0 + b)] = val[b].x;
                            d[2 * (p0 + b) + 1] = val[b].y;
                            dpair[p0 + b] = val[b];
                        }
                    }
                }
                // first half of the ramp fit (registers only): slope, errors, approximate jump significances
                const bool unsat = ((qw[(G - 1) / 4] >> (8 * ((G - 1) & 3))) & DQ_SATURATED) == 0;
                if (!(dbg & 4))
                    fit_full_pk_a<G, rip_full_valid<G, START>()>(dpair, fc0, v0, kf->a.dense, e_gain, e_read, unsat && act, kf->guard, fs);
                if (BAR >= 1) part_fb(kf);
                if (BAR >= 2) part_flags(kf);
                if (BAR >= 3) part_finish(kf);
            }
            CH_T(2)
            C2_SYNC();
            CH_T(3)
            const RIP_K C2KernArgs *kg = c2_args(kargs);  // S2 copy
            // ---- S2: the rest of pixel (r, c) (C2_BAR); at its end the per-pixel words of row r + 1 from the rings
            CH_T(4)
            C2_DRAIN()
            CH_T(5)
            if (emit) {
                if (BAR < 1) part_fb(kg);
                if (BAR < 2) part_flags(kg);
                if (BAR < 3) part_finish(kg);
            }
            // x of (r + 1, own column) for the next step's O2: its ring slot is overwritten in S1 of that step (row r + 4)
            {
                const int sn = (o0_r == 2) ? 0 : o0_r + 1;
#pragma unroll
                for (int p0 = 0; p0 < GP; ++p0) xnext[p0] = X2[(p0 * XR + sn) * C2_COLS + col];
                dq_next = DQ[sn * C2_COLS + col];
                if constexpr (WRING) {
#pragma unroll
                    for (int i = 0; i < QW; ++i) qw_next[i] = QS[(sn * QW + i) * C2_COLS + col];
                    gain_next = GN[sn * C2_COLS + col];
                } else {
                    // the fit role's own loads of row r + 1 (second read of lines its ingest role fetched three steps earlier):
                    // packed as the ingest role packs them; rows / columns outside the frame are never emitted
                    const unsigned yl = (unsigned)min(max((dbg & 1024) ? R0 : r + 1, ylo), yhi);   // (dbg 1024: timing, the re-read hits cache)
                    const __amdgpu_buffer_rsrc_t rq = c2_rsrc(kg->a.gdq);
                    unsigned o1 = yl * (row4 >> 2);
#pragma unroll
                    for (int g = 0; g < G; ++g) {   // (raw: packed at the top of the next step, when they have landed)
                        qb_next[g] = c2_ld_u8(rq, cc1, o1);
                        o1 += npix;
                    }
                    gain_next = c2_ld_f32(c2_rsrc(kg->a.planes), cc4, (unsigned)(NP + 4) * pl4 + yl * row4);
                }
            }
#if C2_KFIT_EARLY == 2   // (row r + 2: the lines the ingest role fetched half a step ago)
            if constexpr (!KRING) {
                if constexpr (K64)
                    (void)load_kd(kg->a.kern, r + 2, true, kn2_d, C2Int<KRN>{});
                else
                    (void)load_k(kg->a.kern, r + 2, true, kn2);
            }
#elif C2_KFIT_EARLY
            if constexpr (!KRING) {
                if constexpr (K64)
                    (void)load_kd(kg->a.kern, (dbg & 1024) ? R0 : r + 1, true, kn_d, C2Int<KRN>{});
                else
                    (void)load_k(kg->a.kern, (dbg & 1024) ? R0 : r + 1, true, kn);
            }
#endif
            CH_T(6)
            C2_SYNC();
            CH_T(7)
        }
    }
#ifdef CH_STAMP
    if ((tid & 63) == 0 && a.dbg_buf) {
        unsigned long long *o = a.dbg_buf + ((size_t)blockIdx.x * (C2_THREADS / 64) + (tid >> 6)) * 9;
        for (int i = 0; i < 9; ++i) o[i] += st_[i];
    }
#endif
}

static inline size_t chain2_lds_bytes(int G, size_t ksize = 4, int cols = C2_COLS_DEF, int krn = 9, bool wring = true) {
    // x ring (3 rows) + O1 ring (3 rows) + flag word (+ packed groupdq / gain) rings (3 rows) + channel lines + K ring (2 rows)
    return (size_t)(G / 2) * cols * 8 * 3 + (size_t)G * cols * ksize * 3 + (size_t)cols * 4 * 3 * (wring ? 2 + (G + 3) / 4 : 1) +
           (size_t)(cols / RIP_CW + 1) * G * 2 * 8 + (size_t)2 * krn * cols * ksize;
}

// Launch geometry on `slots` co-resident workgroups, `reserve` of them left free where that costs nothing (the pre-pass of the NEXT
// ramp runs in them beside this kernel): every strip gets the same number of row ranges -- except a last strip of at most 64 live
// columns (nx = 4096 in the 256-column form: 16 strips of 252 + 64), which is covered in QUAD mode: nq workgroups whose four wave
// columns take a row range each.  4096 x 4096: 16 x 31 ranges of 133 rows + 8 quad workgroups (32 ranges of 128 rows) = 504 workgroups
// of 139 steps; before (17 x 30 ranges of 137 rows): 510 of 143.  Returns the grid size.
static inline long chain2_geometry(ChainArgs &a, int nstrips, int live_last, int slots, int reserve, int wc = 4, bool quad_ok = true) {
    const int maxr = (a.ny + 7) / 8;   // at least 8 rows per range
    auto cdiv = [](int x, int y) { return (x + y - 1) / y; };
    int best_nr = 0, best_nq = 0, best_steps = 1 << 30;
    if (quad_ok && nstrips > 1 && live_last <= 64) {
        const int nfull = nstrips - 1;
        for (int pass = 0; pass < 2 && !best_nr; ++pass) {   // (second pass: without the reserve, when it leaves no room)
            const int avail = slots - (pass ? 0 : reserve);
            for (int nq = 1; wc * nq <= maxr && nq < avail; ++nq) {
                int nr = (avail - nq) / nfull;
                if (nr > maxr) nr = maxr;
                if (nr < 1) break;
                const int steps = cdiv(a.ny, nr) > cdiv(a.ny, wc * nq) ? cdiv(a.ny, nr) : cdiv(a.ny, wc * nq);
                if (steps < best_steps) best_steps = steps, best_nr = nr, best_nq = nq;
            }
        }
    }
    int nr_u = (slots - reserve) / nstrips;   // every strip alike
    if (nr_u < 1) nr_u = slots / nstrips;
    if (nr_u > maxr) nr_u = maxr;
    if (nr_u < 1) nr_u = 1;
    if (best_nr && best_steps < cdiv(a.ny, nr_u)) {
        a.geo_nr = best_nr, a.geo_rows = cdiv(a.ny, best_nr), a.geo_nq = best_nq, a.geo_rows_q = cdiv(a.ny, wc * best_nq);
        return (long)best_nr * (nstrips - 1) + best_nq;
    }
    a.geo_nr = nr_u, a.geo_rows = cdiv(a.ny, nr_u), a.geo_nq = 0, a.geo_rows_q = 0;
    return (long)nr_u * nstrips;
}

template <int NP, int G, int START, typename KT = float, int NARROW = 0>
static int launch_chain2_s(rip_ctx *ctx, const RipPlan *plan, const ChainArgs &a) {
    constexpr int COLS = c2_cols(sizeof(KT) == 8, NARROW);
    constexpr int KRN = !NARROW ? 9 : ((NARROW == 1 && sizeof(KT) == 8) ? c2_krn(G) : 0);
    const size_t lds = chain2_lds_bytes(G, sizeof(KT), COLS, KRN, NARROW < 2);
    const int ncu = ctx->ncu;
    int per_cu = (int)((160 * 1024) / lds);
    if (per_cu < 1) per_cu = 1;
    // 4 waves/SIMD at <= 128 VGPRs (2 at 256 for G = 16 / f64; 3 at 168 in the narrow form)
    const int max_wg = 4 * c2_wps(G, sizeof(KT) == 8, NARROW) / (C2_THREADS / 64);
    if (per_cu > max_wg) per_cu = max_wg;
    if (a.nb < 2) return 1;   // (the frame-edge lanes of the first / last strip emit without neighbours: border pixels)
    ChainArgs ag = a;
    const long grid = chain2_geometry(ag, C2_NSTRIPS(a.nx), a.nx - (C2_NSTRIPS(a.nx) - 1) * C2_OUTW, ncu * per_cu,
                                      NARROW ? 0 : ctx->chain_reserve, COLS / 64, ctx->chain_quad);
    static bool lds_set[64] = {};   // per device, once per instantiation (contexts are used from one thread each)
    if (lds > 48 * 1024 && !lds_set[ctx->device & 63]) {
        RIP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(chain2_kernel<NP, G, START, KT, NARROW>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        lds_set[ctx->device & 63] = true;
    }
    hipLaunchKernelGGL((chain2_kernel<NP, G, START, KT, NARROW>), dim3((unsigned)grid), dim3(C2_THREADS), lds, ctx->stream, ag,
                       reinterpret_cast<const RipPlanHeader *>(plan->dev), plan->d_variants, plan->d_k, plan->d_diffs,
                       ctx->guard_band);
    RIP_HIP(ctx, hipGetLastError());
    return RIP_OK;
}

// returns the launch status, or 1 when the plan is not one the specialised kernel was compiled for
template <int NP, int G, typename KT = float, int NARROW = 0>
static int launch_chain2(rip_ctx *ctx, const RipPlan *plan, const ChainArgs &a) {
    if (plan->h.start == 0 && plan->dense.valid == rip_full_valid<G, 0>()) return launch_chain2_s<NP, G, 0, KT, NARROW>(ctx, plan, a);
    if (plan->h.start == 1 && plan->dense.valid == rip_full_valid<G, 1>()) return launch_chain2_s<NP, G, 1, KT, NARROW>(ctx, plan, a);
    return 1;
}
