// Reference-pixel correction tables in ONE launch (the pre-pass in front of the fused kernel).
//
// Replaces (reference file:line), like the multi-launch form in refpix.hip whose results it reproduces bit for bit:
//   L1_to_L2/gen_cal_image.py:536-539        amp33 block = amp33 - med, minus its own np.median
//   utils/reference_subtraction.py:104-123   ref_subtraction_row: row medians of the reference output, their median `ctr`,
//                                            per-row correction slope * (med - ctr)
//   utils/reference_subtraction.py:50-60     ref_subtraction_channel: medians of the bottom / top 4 reference rows of every
//                                            128-column channel, line through them
// (all medians exact: np.median of an even count = f32 mean of the two middle elements).
//
// Where it is used: a pre-pass that runs in front of its own ramp on the same stream (single calls, host arrays, "overlap" off):
// 0.068 ms against the 0.094 ms of the nine launches of refpix.hip.  A pre-pass that OVERLAPS the previous ramp's fused kernel
// (device-resident ramps back to back: the bench) keeps the nine small launches: they slip into the fused kernel's tail and cost
// 0.03 ms of wall time per ramp, which neither this kernel in-stream (+0.07 ms) nor a one-workgroup-per-group variant of it
// running beside the fused kernel in the workgroup slots that kernel leaves free (0.58 ms alone on its 8 CUs, 0.93 ms beside
// the fused kernel: longer than the kernel it hides behind) beats -- same-box A/B in profiles/r04_summary.md.  Here the
// phases of one group (= one resultant of the ramp) are separated by barriers among the workgroups of THAT group only; groups are
// independent.  Workgroups are numbered group-major, and a workgroup only ever waits for workgroups of its own group, all of
// which precede every later group's in each XCD's dispatch queue -- so a partially resident grid (a busy or shared device) cannot
// deadlock: the lowest unfinished group always becomes resident.
//
//   phase A   every workgroup: 128 rows of the reference output of its group; a wave sorts a row's 128 values (two registers
//             per lane, bitonic network on order-preserving integer keys: DPP / swizzle lane exchanges + v_med3_u32) -> the row's
//             two middle elements; the sorted keys stay in registers.  Some waves also sort one of the 8 x nch rows of
//             (data - dark) the channel step needs (the row correction x -> f32(f64(x) - rc) is monotone, so the order survives it).
//   S1        global median M of the group's ny * 128 values: 3-level radix selection (11 + 11 + 10 key bits); histograms from the
//             register-resident sorted keys with run-length aggregation (one LDS atomic per distinct bin of a sorted slot: no
//             contention whatever the distribution), merged with global atomics, one group barrier per level, every workgroup
//             scans the merged histogram itself.
//   S2, S3    workgroup 0 of the group: row medians -> their median ctr (the same selection inside the workgroup) -> rowcorr;
//             channel medians of 4 x 128 presorted values by merge networks in one wave each -> lines.
//
// Cross-workgroup visibility without fences (MI355X_MICROARCH.md, "Hand-offs measured with sc1 loads"): histograms and counters
// are agent-scope atomics, everything else that crosses workgroups is written with sc1 (write-through) stores of whole 128-byte
// lines by one wave instruction, each storing wave waits vmcnt(0) before its workgroup's lane 0 arrives on the counter, consumers
// poll with sc1 loads, pass a workgroup barrier and read with sc1 loads.
#include "rip_common.h"
#include "refpix_keys.h"
#include <string.h>

namespace {

constexpr int R1_THREADS = 1024;
constexpr int R1_WAVES = R1_THREADS / 64;
constexpr int R1_RPW = 8;                     // reference-output rows per wave (two key registers each)
constexpr int R1_ROWS = R1_WAVES * R1_RPW;    // rows of a workgroup
constexpr int R1_NV = 4;                      // row medians per thread in S2: ny <= 4096
constexpr unsigned R1_SPIN_LIMIT = 1u << 22;  // a barrier that never completes ends with an error word, not a hung device

// per-group control words (zero between launches: the last workgroup of a group to leave restores them)
struct R1Ctrl {
    uint32_t arrive[3];
    uint32_t exitc;
};

struct R1Args {
    const void *data;
    const float *dark;
    const uint16_t *amp33;
    const float *med;
    const double *lines_override;
    double *rowcorr, *rowcorr_t, *lines;
    R1Ctrl *ctrl;        // [G]
    uint32_t *ghist;     // [G][3][2][SEL_BINS]
    uint32_t *lo, *hi;   // [G][ny] keys of the two middle elements of every row
    uint32_t *chsort;    // [G][nch][8][128] presorted keys of the channel rows (odd rows descending)
    uint32_t *status;    // != 0: a barrier timed out
    unsigned long long *stamps;   // diagnostic (tools/gpu_checks/prepass_stamps.py): 16 clock stamps per workgroup, or null
    double slope;
    int ny, nx, G, B;
};

__device__ __forceinline__ uint32_t ld_sc1(const uint32_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1(uint32_t *p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// median of three (v_med3_u32): med3(v, p, 0) = min(v, p), med3(v, p, ~0) = max(v, p)
__device__ __forceinline__ uint32_t med3u(uint32_t a, uint32_t b, uint32_t c) {
    const uint32_t mn = a < b ? a : b, mx = a < b ? b : a;
    const uint32_t t = mx < c ? mx : c;
    return mn > t ? mn : t;
}
__device__ __forceinline__ uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint32_t umax(uint32_t a, uint32_t b) { return a < b ? b : a; }

// value of lane (lane ^ J), vector-ALU instructions only (ds_swizzle / ds_bpermute go through the LDS pipe, one per CU: with 16
// waves exchanging lanes all the time it, not the four vector pipes, set the pace -- measured 2.5 x): DPP inside a row of 16
// lanes, the gfx950 row / half swaps across rows
template <int J>
__device__ __forceinline__ uint32_t lane_xor(uint32_t v, int lane) {
    if constexpr (J == 1) {
        return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
    } else if constexpr (J == 2) {
        return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
    } else if constexpr (J == 4) {
        // lanes with bit 2 clear (banks 0, 2) take lane + 4 (row_shl:4), the others lane - 4 (row_shr:4)
        const int p = __builtin_amdgcn_update_dpp(0, (int)v, 0x104, 0xf, 0x5, false);
        return (uint32_t)__builtin_amdgcn_update_dpp(p, (int)v, 0x114, 0xf, 0xA, false);
    } else if constexpr (J == 8) {
        return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, true);  // row_ror:8
    } else if constexpr (J == 16) {
        // v_permlane16_swap: odd rows of the first operand <-> even rows of the second: {[r0 r0 r2 r2], [r1 r1 r3 r3]}
        const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        return (lane & 16) ? r[0] : r[1];
    } else {
        // v_permlane32_swap: upper half of the first operand <-> lower half of the second: {[lo lo], [hi hi]}
        const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
        return (lane & 32) ? r[0] : r[1];
    }
}
// One compare-exchange stage (distance J) of the bitonic network on NS independent slots of 64 keys (a key per lane and slot),
// slot loop innermost so that the lane exchanges of one slot fill the wait states of the next.  Blocks of k lanes alternate
// between ascending and descending (k = 64: one direction, `desc`).  Distances 16 and 32 take the slots in PAIRS: one row / half
// swap puts the partners of both slots side by side ({[X.r0 Y.r0 X.r2 Y.r2], [X.r1 Y.r1 X.r3 Y.r3]}), a minimum and a maximum
// (direction per lane: v_med3 against 0 / ~0) and the same swap puts them back -- 2 instructions per slot, no copies.
template <int J, int NS>
__device__ __forceinline__ void stageN(uint32_t (&v)[NS], int lane, int k, bool desc) {
    bool up = (k == 64) ? true : ((lane & k) == 0);
    if (desc) up = !up;
    if constexpr (J >= 16 && NS % 2 == 0) {
        const uint32_t cA = up ? 0u : 0xffffffffu;
#pragma unroll
        for (int s = 0; s < NS; s += 2) {
            uint32_t P, Q;
            if constexpr (J == 16) {
                const auto r = __builtin_amdgcn_permlane16_swap(v[s], v[s + 1], false, false);
                P = med3u(r[0], r[1], cA), Q = med3u(r[0], r[1], ~cA);
                const auto r2 = __builtin_amdgcn_permlane16_swap(P, Q, false, false);
                v[s] = r2[0], v[s + 1] = r2[1];
            } else {
                const auto r = __builtin_amdgcn_permlane32_swap(v[s], v[s + 1], false, false);
                P = med3u(r[0], r[1], cA), Q = med3u(r[0], r[1], ~cA);
                const auto r2 = __builtin_amdgcn_permlane32_swap(P, Q, false, false);
                v[s] = r2[0], v[s + 1] = r2[1];
            }
        }
    } else {
        const uint32_t c = (((lane & J) == 0) == up) ? 0u : 0xffffffffu;
        uint32_t p[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) p[s] = lane_xor<J>(v[s], lane);
#pragma unroll
        for (int s = 0; s < NS; ++s) v[s] = med3u(v[s], p[s], c);
    }
}
// NS slots of 64 keys, each sorted across the lanes: bitonic network, 21 compare-exchange stages
template <int NS>
__device__ __forceinline__ void sortN(uint32_t (&v)[NS], int lane, bool desc = false) {
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
        if (k >= 64) stageN<32, NS>(v, lane, k, desc);
        if (k >= 32) stageN<16, NS>(v, lane, k, desc);
        if (k >= 16) stageN<8, NS>(v, lane, k, desc);
        if (k >= 8) stageN<4, NS>(v, lane, k, desc);
        if (k >= 4) stageN<2, NS>(v, lane, k, desc);
        stageN<1, NS>(v, lane, k, desc);
    }
}
// NS bitonic sequences of 64 keys -> sorted (the last merge of the network alone)
template <int NS>
__device__ __forceinline__ void mergeN(uint32_t (&v)[NS], int lane, bool desc = false) {
    stageN<32, NS>(v, lane, 64, desc);
    stageN<16, NS>(v, lane, 64, desc);
    stageN<8, NS>(v, lane, 64, desc);
    stageN<4, NS>(v, lane, 64, desc);
    stageN<2, NS>(v, lane, 64, desc);
    stageN<1, NS>(v, lane, 64, desc);
}
__device__ __forceinline__ uint32_t sort64(uint32_t v, int lane, bool desc = false) {
    uint32_t a[1] = {v};
    sortN<1>(a, lane, desc);
    return a[0];
}
__device__ __forceinline__ uint32_t merge64(uint32_t v, int lane, bool desc = false) {
    uint32_t a[1] = {v};
    mergeN<1>(a, lane, desc);
    return a[0];
}
__device__ __forceinline__ uint32_t rev64(uint32_t v, int lane) {
    return (uint32_t)__builtin_amdgcn_ds_bpermute((63 - lane) << 2, (int)v);
}
// The two middle elements of 128 keys held as k0 (64 keys ascending across the lanes) and k1r (the other 64, descending): the 64
// smallest are min(k0[i], k1r[i]) -- k0 up to the lane x where the sequences cross, k1r from there -- so their maximum sits at the
// crossing, and so does the minimum of the 64 largest.  One wave vote and four lane reads.
__device__ __forceinline__ void middle_pair(uint32_t k0, uint32_t k1r, uint32_t &lo, uint32_t &hi) {
    const int x = __popcll(__ballot(k0 <= k1r));   // lanes [0, x): k0 <= k1r (monotone: k0 rises, k1r falls)
    const uint32_t a0 = (uint32_t)__builtin_amdgcn_readlane((int)k0, x > 0 ? x - 1 : 0);
    const uint32_t a1 = (uint32_t)__builtin_amdgcn_readlane((int)k0, x < 64 ? x : 63);
    const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)k1r, x > 0 ? x - 1 : 0);
    const uint32_t b1 = (uint32_t)__builtin_amdgcn_readlane((int)k1r, x < 64 ? x : 63);
    lo = x == 0 ? b1 : (x == 64 ? a0 : umax(a0, b1));
    hi = x == 0 ? a1 : (x == 64 ? b0 : umin(b0, a1));
}

// maximum / minimum over the wave (returned to every lane): running maximum along each row of 16 lanes by DPP shifts (lanes
// without a source keep their own value), the row results passed on by the two row broadcasts; lane 63 holds the result
template <bool MAX>
__device__ __forceinline__ uint32_t wave_reduce(uint32_t v, int lane) {
    auto op = [](uint32_t x, int y) { return MAX ? umax(x, (uint32_t)y) : umin(x, (uint32_t)y); };
    v = op(v, __builtin_amdgcn_update_dpp((int)v, (int)v, 0x111, 0xf, 0xf, false));   // row_shr:1
    v = op(v, __builtin_amdgcn_update_dpp((int)v, (int)v, 0x112, 0xf, 0xf, false));   // row_shr:2
    v = op(v, __builtin_amdgcn_update_dpp((int)v, (int)v, 0x114, 0xf, 0xf, false));   // row_shr:4
    v = op(v, __builtin_amdgcn_update_dpp((int)v, (int)v, 0x118, 0xf, 0xf, false));   // row_shr:8: lane 15 of a row = its result
    v = op(v, __builtin_amdgcn_update_dpp((int)v, (int)v, 0x142, 0xa, 0xf, false));   // row_bcast:15 into rows 1, 3
    v = op(v, __builtin_amdgcn_update_dpp((int)v, (int)v, 0x143, 0xc, 0xf, false));   // row_bcast:31 into rows 2, 3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t wave_umax(uint32_t v, int lane) { return wave_reduce<true>(v, lane); }
__device__ __forceinline__ uint32_t wave_umin(uint32_t v, int lane) { return wave_reduce<false>(v, lane); }

// One monotone (sorted either way) slot of 64 keys into the LDS histogram of a selection level: keys whose bits above the
// level's digit equal the prefix's; lanes with the same digit are neighbours, the first of a run adds the run's length.
// `live`: the lane holds a value.
__device__ __forceinline__ void hist_slot(uint32_t key, bool live, int lv, uint32_t prefix, uint32_t *h, int lane) {
    const int shift = sel_shift(lv), bits = sel_bits(lv), above = shift + bits;
    const bool inr = live && (above >= 32 || ((key ^ prefix) >> above) == 0);
    const uint32_t bin = inr ? ((key >> shift) & ((1u << bits) - 1u)) : 0xffffu;
    // (the first lane of every row of 16 starts a run of its own: a DPP shift does not cross rows)
    const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp((int)~bin, (int)bin, 0x111, 0xf, 0xf, false);   // row_shr:1
    const bool head = inr && prev != bin;
    const unsigned long long stop = __ballot(head || !inr);
    if (head) {
        const unsigned long long m = (stop >> 1) >> lane;   // bit t: lane + 1 + t ends the run
        const int run = m ? __ffsll((long long)m) : 64 - lane;
        atomicAdd(&h[bin], (uint32_t)run);
    }
}

// Among SEL_BINS counts (four per thread: c[0..3] = bins 4t .. 4t+3 for t = tid & 511; the halves of the workgroup scan one
// histogram each, q = tid >> 9) the bin holding rank[q]; result in sel[q] = {bin, rank inside the bin}.  All threads call.
__device__ __forceinline__ void scan_find(const uint32_t (&c)[4], uint32_t rank, uint32_t (*wtot)[8], uint32_t (*sel)[2], int tid) {
    const int lane = tid & 63, q = tid >> 9, t = tid & 511, wq = (tid >> 6) & 7;
    const uint32_t own = c[0] + c[1] + c[2] + c[3];
    uint32_t incl = own;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t y = (uint32_t)__shfl_up((int)incl, off, 64);
        if (lane >= off) incl += y;
    }
    if (lane == 63) wtot[q][wq] = incl;
    __syncthreads();
    for (int k = 0; k < wq; ++k) incl += wtot[q][k];
    const uint32_t excl = incl - own;
    if ((excl <= rank && rank < incl) || (t == 511 && rank >= incl)) {
        uint32_t cum = excl;
        int b = 0;
        while (b < 3 && cum + c[b] <= rank) cum += c[b++];
        sel[q][0] = (uint32_t)(4 * t + b);
        sel[q][1] = rank - cum;
    }
    __syncthreads();
}

// barrier among the B workgroups of a group on a counter that counts arrivals (zero at launch)
__device__ __forceinline__ void group_barrier(uint32_t *ctr, uint32_t target, uint32_t *status) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's atomics and sc1 stores have been performed
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (ld_sc1(ctr) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > R1_SPIN_LIMIT) {
                st_sc1(status, 1u);
                break;
            }
        }
    }
    __syncthreads();
}

// a row of 128 keys (k0: elements 0..63, k1: 64..127 of the row, one per lane) sorted: ascending -- lo holds ranks 0..63 in lane
// order, hi 64..127 -- or descending (lo: the 64 largest, largest first)
__device__ __forceinline__ void sort128(uint32_t k0, uint32_t k1, int lane, uint32_t &lo, uint32_t &hi, bool desc = false) {
    uint32_t k[2] = {k0, k1};
    sortN<2>(k, lane);
    const uint32_t k1r = rev64(k[1], lane);
    uint32_t m[2] = {desc ? umax(k[0], k1r) : umin(k[0], k1r), desc ? umin(k[0], k1r) : umax(k[0], k1r)};
    mergeN<2>(m, lane, desc);
    lo = m[0], hi = m[1];
}

// np.median of 4 x 128 keys: x[row][slot], rows 0 and 2 sorted ascending (slot 0: ranks 0..63 in lane order), rows 1 and 3
// descending.  Rows (0, 1) -> P: 256 keys ascending in four slots; rows (2, 3) -> Q: descending.  Flip stage of the bitonic merge:
// X[i] against Y[127 - i] (Y is held descending), the minima are the 128 smallest: a bitonic sequence over two slots, sorted by the
// half-cleaner between the slots and the 64-lane merges.  Then P[i] against Q[255 - i]: the maximum of the minima is rank 255, the
// minimum of the maxima rank 256.
__device__ __forceinline__ void merge_rows(const uint32_t (&xa)[2], const uint32_t (&yd)[2], bool desc, uint32_t (&out)[4], int lane) {
    const uint32_t l0 = umin(xa[0], yd[0]), l1 = umin(xa[1], yd[1]);
    const uint32_t h0 = umax(xa[0], yd[0]), h1 = umax(xa[1], yd[1]);
    const uint32_t la = umin(l0, l1), lb = umax(l0, l1), ha = umin(h0, h1), hb = umax(h0, h1);
    if (!desc)
        out[0] = la, out[1] = lb, out[2] = ha, out[3] = hb;
    else
        out[0] = hb, out[1] = ha, out[2] = lb, out[3] = la;
    mergeN<4>(out, lane, desc);
}
__device__ __forceinline__ float chan_median(const uint32_t (&x)[4][2], int lane) {
    uint32_t P[4], Q[4];
    merge_rows(x[0], x[1], false, P, lane);
    merge_rows(x[2], x[3], true, Q, lane);
    uint32_t lo = umin(P[0], Q[0]), hi = umax(P[0], Q[0]);
#pragma unroll
    for (int s = 1; s < 4; ++s) {
        lo = umax(lo, umin(P[s], Q[s]));
        hi = umin(hi, umax(P[s], Q[s]));
    }
    lo = wave_umax(lo, lane);
    hi = wave_umin(hi, lane);
    return (key2f(lo) + key2f(hi)) * 0.5f;
}
// the line through (1.5, b), (ny - 2.5, t) (reference_subtraction.py:57-60; DESIGN.md "channel line fit")
__device__ __forceinline__ void store_line(const R1Args &a, int g, int nch, int ch, float b, float t) {
    const double m = ((double)t - (double)b) / (double)(a.ny - 4);
    const double c = (double)b - 1.5 * m;
    a.lines[((size_t)g * nch + ch) * 2] = m;
    a.lines[((size_t)g * nch + ch) * 2 + 1] = c;
}

// Row tables of one group from its row medians (refmed[k]: row tid + 1024 k, already minus the block's median): ctr = their
// np.median by the three-level selection inside the workgroup (histograms in LDS, zero on entry and on return),
// rowcorr = slope * f64(f32(refmed - ctr)) (reference_subtraction.py:115-123), and the corrections of the 4 + 4 rows the channel
// step reads in rc8.  All threads of the workgroup call; ends with a barrier.
__device__ __forceinline__ void row_tables(const float (&refmed)[R1_NV], const R1Args &a, int g, uint32_t (*hist)[SEL_BINS],
                                           uint32_t (*wtot)[8], uint32_t (*sel)[2], double *rc8) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, ny = a.ny;
    uint32_t rk[R1_NV];
    int nlive[R1_NV];   // live lanes of the slot in this wave (rows tid + 1024 k: a slot's rows are consecutive)
#pragma unroll
    for (int k = 0; k < R1_NV; ++k) {
        const int r = tid + R1_THREADS * k;
        const int first = (w << 6) + R1_THREADS * k;   // row of lane 0
        nlive[k] = min(max(ny - first, 0), 64);
        // (sorted ascending below: the lanes without a row at the top -- any key equal to the filler is interchangeable with it)
        rk[k] = r < ny ? f2key(refmed[k]) : 0xffffffffu;
    }
    sortN<R1_NV>(rk, lane);
    uint32_t pre2[2] = {0u, 0u}, rank2[2] = {(uint32_t)((ny & 1) ? ny / 2 : ny / 2 - 1), (uint32_t)(ny / 2)};
#pragma unroll 1
    for (int lv = 0; lv < 3; ++lv) {
        const bool same = pre2[0] == pre2[1];
#pragma unroll
        for (int k = 0; k < R1_NV; ++k) {
            if (nlive[k] == 0) continue;
            hist_slot(rk[k], lane < nlive[k], lv, pre2[0], hist[0], lane);
            if (!same) hist_slot(rk[k], lane < nlive[k], lv, pre2[1], hist[1], lane);
        }
        __syncthreads();
        {
            const int q = tid >> 9, t = tid & 511;
            uint32_t *src = hist[same ? 0 : q] + 4 * t;
            uint32_t c[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) c[k] = src[k];
            scan_find(c, rank2[q], wtot, sel, tid);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            pre2[q] |= sel[q][0] << sel_shift(lv);
            rank2[q] = sel[q][1];
        }
        for (int i = tid; i < 2 * SEL_BINS; i += R1_THREADS) (&hist[0][0])[i] = 0;
        __syncthreads();
    }
    const float ctr = (key2f(pre2[0]) + key2f(pre2[1])) * 0.5f;
#pragma unroll
    for (int k = 0; k < R1_NV; ++k) {
        const int r = tid + R1_THREADS * k;
        if (r < ny) {
            const double v = a.slope * (double)(refmed[k] - ctr);
            a.rowcorr[(size_t)g * ny + r] = v;
            if (a.rowcorr_t) a.rowcorr_t[(size_t)r * a.G + g] = v;   // [row][group]: one scalar load per row in the fused kernel
            if (r < 4) rc8[r] = v;
            if (r >= ny - 4) rc8[4 + r - (ny - 4)] = v;
        }
    }
    __syncthreads();
}


#define R1_STAMP(k)                                                                                    \
    if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memtime();

template <typename DT>
__global__ __launch_bounds__(R1_THREADS) void refpix_one_kernel(R1Args a) {
    __shared__ uint32_t hist[2][SEL_BINS];
    __shared__ uint32_t wtot[2][8];
    __shared__ uint32_t sel[2][2];
    __shared__ uint32_t lohi_s[2][R1_ROWS];
    __shared__ double rc8[8];
    __shared__ uint32_t flag_s;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g = (int)blockIdx.x / a.B, b = (int)blockIdx.x % a.B;
    const int ny = a.ny, nx = a.nx, nch = nx / RIP_CW;
    R1Ctrl *ctrl = a.ctrl + g;
    uint32_t *gh = a.ghist + (size_t)g * 3 * 2 * SEL_BINS;
    for (int i = tid; i < 2 * SEL_BINS; i += R1_THREADS) (&hist[0][0])[i] = 0;
    R1_STAMP(0)

    // ---- phase A: the rows of this workgroup.  Row i of wave w: r0 + i * 16 + w
    const int r0 = b * R1_ROWS;
    uint32_t key[R1_RPW][2];
    bool rvalid[R1_RPW];
    {
        uint32_t raw[R1_RPW][2];
        float md[R1_RPW][2];
#pragma unroll
        for (int i = 0; i < R1_RPW; ++i) {
            const int r = r0 + i * R1_WAVES + w;
            rvalid[i] = r < ny;
            const int rr = rvalid[i] ? r : ny - 1;
            const uint16_t *ap = a.amp33 + ((size_t)g * ny + rr) * RIP_CW;
            const float *mp = a.med + (size_t)rr * RIP_CW;
            raw[i][0] = ap[lane], raw[i][1] = ap[lane + 64];
            md[i][0] = mp[lane], md[i][1] = mp[lane + 64];
        }
        // channel rows of (data - dark): item it = (ch, j), j = 0..3 the bottom rows, 4..7 the top rows; a wave takes items
        // b * 16 + w, + B * 16, ...  Sorted ascending; the odd rows are stored descending (what the merge in S3 wants).
        if (!a.lines_override) {
            for (int it = b * R1_WAVES + w; it < nch * 8; it += a.B * R1_WAVES) {
                const int ch = it >> 3, j = it & 7;
                const int row = j < 4 ? j : ny - 8 + j;
                const size_t idx = ((size_t)g * ny + row) * nx + (size_t)ch * RIP_CW + lane;
                const DT *dp = (const DT *)a.data;
                const uint32_t k0 = f2key((float)dp[idx] - a.dark[idx]);
                const uint32_t k1 = f2key((float)dp[idx + 64] - a.dark[idx + 64]);
                uint32_t lo, hi;
                sort128(k0, k1, lane, lo, hi);
                uint32_t *o = a.chsort + ((size_t)(g * nch + ch) * 8 + j) * RIP_CW;
                if (j & 1) {
                    st_sc1(o + 127 - lane, lo);
                    st_sc1(o + 63 - lane, hi);
                } else {
                    st_sc1(o + lane, lo);
                    st_sc1(o + 64 + lane, hi);
                }
            }
        }
        uint32_t ks[2 * R1_RPW];
#pragma unroll
        for (int i = 0; i < R1_RPW; ++i) {
            ks[2 * i] = f2key((float)raw[i][0] - md[i][0]);
            ks[2 * i + 1] = f2key((float)raw[i][1] - md[i][1]);
        }
        sortN<2 * R1_RPW>(ks, lane);
#pragma unroll
        for (int i = 0; i < R1_RPW; ++i) {
            uint32_t lo, hi;
            middle_pair(ks[2 * i], rev64(ks[2 * i + 1], lane), lo, hi);
            if (lane == 0) {
                lohi_s[0][i * R1_WAVES + w] = lo;
                lohi_s[1][i * R1_WAVES + w] = hi;
            }
            key[i][0] = ks[2 * i], key[i][1] = ks[2 * i + 1];
        }
    }
    __syncthreads();
    R1_STAMP(1)
    // the middle elements of this workgroup's rows, whole lines per wave instruction
    if (tid < 2 * R1_ROWS) {
        const int p = tid / R1_ROWS, i = tid % R1_ROWS;
        if (r0 + i < ny) st_sc1((p ? a.hi : a.lo) + (size_t)g * ny + r0 + i, lohi_s[p][i]);
    }

    // ---- S1: the two middle elements of the group's ny * 128 values
    const uint32_t n = (uint32_t)ny * RIP_CW;
    uint32_t prefix[2] = {0u, 0u}, rank[2] = {n / 2 - 1, n / 2};
#pragma unroll 1
    for (int lv = 0; lv < 3; ++lv) {
        const bool same = prefix[0] == prefix[1];   // (level 0: no prefix yet)
#pragma unroll
        for (int i = 0; i < R1_RPW; ++i) {
            if (!rvalid[i]) continue;
            hist_slot(key[i][0], true, lv, prefix[0], hist[0], lane);
            hist_slot(key[i][1], true, lv, prefix[0], hist[0], lane);
            if (!same) {
                hist_slot(key[i][0], true, lv, prefix[1], hist[1], lane);
                hist_slot(key[i][1], true, lv, prefix[1], hist[1], lane);
            }
        }
        __syncthreads();
        R1_STAMP(2 + 3 * lv)
        uint32_t *ghl = gh + (size_t)lv * 2 * SEL_BINS;
        for (int i = tid; i < (same ? 1 : 2) * SEL_BINS; i += R1_THREADS) {
            const uint32_t c = (&hist[0][0])[i];
            if (c) {
                __hip_atomic_fetch_add(ghl + i, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                (&hist[0][0])[i] = 0;
            }
        }
        group_barrier(&ctrl->arrive[lv], (uint32_t)a.B, a.status);
        R1_STAMP(3 + 3 * lv)
        {
            const int q = tid >> 9, t = tid & 511;
            const uint32_t *src = ghl + (size_t)(same ? 0 : q) * SEL_BINS + 4 * t;
            uint32_t c[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) c[k] = ld_sc1(src + k);
            scan_find(c, rank[q], wtot, sel, tid);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            prefix[q] |= sel[q][0] << sel_shift(lv);
            rank[q] = sel[q][1];
        }
        __syncthreads();   // sel is rewritten by the next level
        R1_STAMP(4 + 3 * lv)
    }
    const float M = (key2f(prefix[0]) + key2f(prefix[1])) * 0.5f;   // np.median of the block

    // this workgroup is done with the group's histograms and counters: the last one to say so restores their zero state
    if (tid == 0) flag_s = __hip_atomic_fetch_add(&ctrl->exitc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (flag_s == (uint32_t)a.B - 1u) {
        for (int i = tid; i < 3 * 2 * SEL_BINS; i += R1_THREADS) st_sc1(gh + i, 0u);
        if (tid < 4) st_sc1(&ctrl->arrive[0] + tid, 0u);
    }
    R1_STAMP(11)
    if (b != 0) return;

    // ---- S2 (workgroup 0 of the group): row medians minus M, their median, the row table
    float refmed[R1_NV];
#pragma unroll
    for (int k = 0; k < R1_NV; ++k) {
        const int r = tid + R1_THREADS * k;
        const int rr = r < ny ? r : ny - 1;
        const float lo = key2f(ld_sc1(a.lo + (size_t)g * ny + rr)) - M;
        const float hi = key2f(ld_sc1(a.hi + (size_t)g * ny + rr)) - M;
        refmed[k] = (lo + hi) * 0.5f;
    }
    row_tables(refmed, a, g, hist, wtot, sel, rc8);
    R1_STAMP(13)

    // ---- S3: channel lines.  A wave per channel: medians of the 4 x 128 row-corrected values of its bottom and top rows
    if (a.lines_override) {
        for (int i = tid; i < nch * 2; i += R1_THREADS) a.lines[(size_t)g * nch * 2 + i] = a.lines_override[(size_t)g * nch * 2 + i];
        return;
    }
    for (int ch = w; ch < nch; ch += R1_WAVES) {
        float bt[2];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            // rows j = 4 half + {0, 1, 2, 3}; x[row][slot]: even rows ascending (slot 0: ranks 0..63), odd rows descending
            uint32_t x[4][2];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const uint32_t *o = a.chsort + ((size_t)(g * nch + ch) * 8 + half * 4 + jj) * RIP_CW;
                const double rc = rc8[half * 4 + jj];
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const float v = key2f(ld_sc1(o + s * 64 + lane));
                    x[jj][s] = f2key((float)((double)v - rc));
                }
            }
            bt[half] = chan_median(x, lane);
        }
        if (lane == 0) store_line(a, g, nch, ch, bt[0], bt[1]);
    }
    R1_STAMP(14)
}


}  // namespace

// whether the single-launch form covers a frame (otherwise: the multi-launch form of refpix.hip)
bool rip_refpix_one_supported(const RefpixArgs &a) {
    if (!a.amp33 || a.nx % RIP_CW || a.ny < 8 || a.ny > R1_NV * R1_THREADS) return false;
    const int B = (a.ny + R1_ROWS - 1) / R1_ROWS;
    return a.ngrp >= 1 && a.ngrp <= RIP_MAX_GROUPS && (long)a.ngrp * B <= 4096;
}

int rip_launch_refpix_one(rip_ctx *ctx, const RefpixArgs &a) {
    hipStream_t st = a.stream ? a.stream : ctx->stream;
    if (!rip_refpix_one_supported(a)) return rip_fail(ctx, RIP_EINVAL, "refpix: frame not covered by the single-launch pre-pass");
    const int G = a.ngrp, ny = a.ny, nch = a.nx / RIP_CW;
    const int B = (ny + R1_ROWS - 1) / R1_ROWS;
    // scratch: control words + histograms (zero between launches; cleared when the slot is (re)allocated) | status | lo, hi | chsort
    const size_t ctrl_b = ((size_t)RIP_MAX_GROUPS * sizeof(R1Ctrl) + 255) / 256 * 256;
    const size_t gh_b = (size_t)RIP_MAX_GROUPS * 3 * 2 * SEL_BINS * sizeof(uint32_t);
    const size_t zero_b = ctrl_b + gh_b + 256;
    const void *had = ctx->ws[14];
    char *z = (char *)rip_ws(ctx, 14, zero_b);
    if (!z) return RIP_ENOMEM;
    if ((const void *)z != had) RIP_HIP(ctx, hipMemsetAsync(z, 0, zero_b, st));
    const size_t lohi_b = ((size_t)G * ny * 4 + 255) / 256 * 256;
    const size_t chs_b = (size_t)G * nch * 8 * RIP_CW * 4;
    char *s = (char *)rip_ws(ctx, 15, 2 * lohi_b + chs_b);
    if (!s) return RIP_ENOMEM;
    R1Args r;
    r.data = a.data;
    r.dark = a.dark_data;
    r.amp33 = a.amp33;
    r.med = a.amp33_med;
    r.lines_override = a.lines_override;
    r.rowcorr = a.rowcorr;
    r.rowcorr_t = a.rowcorr_t;
    r.lines = a.lines;
    r.ctrl = (R1Ctrl *)z;
    r.ghist = (uint32_t *)(z + ctrl_b);
    r.status = (uint32_t *)(z + ctrl_b + gh_b);
    r.lo = (uint32_t *)s;
    r.hi = (uint32_t *)(s + lohi_b);
    r.chsort = (uint32_t *)(s + 2 * lohi_b);
    r.slope = a.slope;
    r.ny = ny;
    r.nx = a.nx;
    r.G = G;
    r.B = B;
    r.stamps = (unsigned long long *)ctx->prepass_stamps;
    if (a.data_dtype == RIP_U16)
        hipLaunchKernelGGL(refpix_one_kernel<uint16_t>, dim3((unsigned)(G * B)), dim3(R1_THREADS), 0, st, r);
    else
        hipLaunchKernelGGL(refpix_one_kernel<float>, dim3((unsigned)(G * B)), dim3(R1_THREADS), 0, st, r);
    RIP_HIP(ctx, hipGetLastError());
    return RIP_OK;
}

// != 0 after a pre-pass whose group barrier timed out (diagnostic; synchronises the stream)
int rip_refpix_one_status(rip_ctx *ctx, int *status) {
    *status = 0;
    if (!ctx->ws[14]) return RIP_OK;
    const size_t ctrl_b = ((size_t)RIP_MAX_GROUPS * sizeof(R1Ctrl) + 255) / 256 * 256;
    const size_t gh_b = (size_t)RIP_MAX_GROUPS * 3 * 2 * SEL_BINS * sizeof(uint32_t);
    uint32_t v = 0;
    RIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->stream2) RIP_HIP(ctx, hipStreamSynchronize(ctx->stream2));
    RIP_HIP(ctx, hipMemcpy(&v, (char *)ctx->ws[14] + ctrl_b + gh_b, 4, hipMemcpyDeviceToHost));
    *status = (int)v;
    return RIP_OK;
}
