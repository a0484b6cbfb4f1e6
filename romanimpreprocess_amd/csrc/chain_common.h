// Shared pieces of the fused L1->L2 kernel (chain2_kernel.h): packed-pair arithmetic, the exact short forms of
// division, the forward IPC operator on LDS rows, the registers of a prefetched row, the diagnostic stamps.
//
// The fused kernels replace gen_cal_image.py:533-629 between the reference-pixel tables (refpix.hip) and the L2 planes: reference-
// pixel apply + bias + Legendre linearity + IPC deconvolution + ramp fit / jump detection / flag propagation + dark rate + error
// split + flat in one launch per ramp, every CALDIR and ramp array read from HBM once, no intermediate cube written; per-stage
// arithmetic and its reference lines are those of linearity.hip, ipc.hip, rampfit.hip (the unfused kernels: the general path for
// any group count, Legendre order, gain dtype or sub-chain, and the stage-level drop-ins).  Round 1's general fused kernel (one
// wave doing every phase, 2 waves/SIMD) lived here; it was dropped in round 3 -- the configurations it alone covered (a CALDIR
// set whose flag words cannot be merged, plans outside the compile-time difference masks) take the stage kernels.
//
// Groups are processed in PAIRS (float2: two IEEE operations side by side, same rounding as the scalar forms): x and the first
// Neumann iterate are stored pair-interleaved in LDS (one ds_read_b64 per neighbour and pair).
//
// Roofline: HBM.  Algorithmic bytes per pixel: G*(2 + 4 + 4 + 1) in, G out (groupdq),
// 4*(NP+3) + 4 linearity, 36 ipc4d, 4 gain, 4 read, 4 dark rate, 4 flat, 4 flags, 4 pdq in, 16 out.
#pragma once
#include "rip_common.h"

#include "device_rampfit.h"

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

// Forward IPC operator (ipc_linearity.py:69-94: term order and edge rule, see ipc.hip) at column `t` of three LDS rows (rm = row
// y-1, r0 = row y, rp = row y+1) for NB pairs of groups at once; ALL = every source is active (interior pixel): no per-term
// selects.  All 9*NB LDS reads are issued first (one exposed LDS latency per
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

// raw bits of one row position, fetched one phase ahead of their use (no arithmetic on them in P)
template <int NP, int G>
struct RowRegs {
    uint32_t S[G];
    uint32_t q[G];
    float dk[G], bs[G];
    float cf[NP], smin, smax, sref, gain;
    uint32_t dq;
};

