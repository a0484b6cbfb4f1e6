// Reference-pixel correction tables (exact medians) -- a small pre-pass in front of the cube kernel.
//
// Replaces (reference file:line):
//   L1_to_L2/gen_cal_image.py:536-539        amp33 block = amp33 - med, minus its own np.median
//   utils/reference_subtraction.py:104-123   ref_subtraction_row  (row medians of the reference output,
//                                            their median `ctr`, per-row correction slope*(med-ctr))
//   utils/reference_subtraction.py:50-60     ref_subtraction_channel (medians of the bottom/top 4 reference
//                                            rows of every 128-column channel, line through them)
// Only the tables are produced here: rowcorr[g][r] (f64) and lines[g][ch] = (m, c) (f64).  The
// per-pixel application is fused into the cube kernel (linearity.hip).  What makes this possible:
//   * with `slope` given (always, when read.amp33 exists) the science-row medians and the polyfit
//     of reference_subtraction.py:107,114 are dead code;
//   * x -> f32(x - M) is monotone, so the rank-63/64 elements of a row of (v - M) are the
//     rank-63/64 elements of v, minus M: row medians and the global median M come from one pass;
//   * the channel step only needs rows 0:4 and ny-4:ny of the row-corrected image.
// All medians are exact selections (np.median: mean of the two middle elements in f32).
// Arithmetic recipe: oracle/refpix.py.
//
// The line through (1.5, b), (ny-2.5, t) is m = (t-b)/(ny-4), c = b - 1.5 m in f64; the reference gets
// it from LAPACK gelsd, which agrees to ~1e-13 relative but not bit for bit -- the caller may pass
// LAPACK's (m, c) instead (`lines_override`).  See DESIGN.md "channel line fit".
#include "rip_common.h"
#include "refpix_keys.h"

// np.median of vals[0..n) held in LDS, by rank counting; result broadcast through slot[0..1].
// All threads of the block must call it.  Ties are ordered by index, so ranks are a permutation.
__device__ float block_median(const float *vals, int n, float *slot) {
    const int k_hi = n / 2, k_lo = (n & 1) ? n / 2 : n / 2 - 1;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float v = vals[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const float w = vals[j];
            rank += (w < v || (w == v && j < i)) ? 1 : 0;
        }
        if (rank == k_lo) slot[0] = v;
        if (rank == k_hi) slot[1] = v;
    }
    __syncthreads();
    const float m = (slot[0] + slot[1]) * 0.5f;
    __syncthreads();
    return m;
}

// np.median of vals[0..n) in LDS by an in-place bitonic sort; the buffer must hold npow2 >= n floats
// (entries n.. are overwritten with +inf).  All threads of the block must call it.
__device__ float block_median_sorted(float *vals, int n, int npow2) {
    for (int i = n + threadIdx.x; i < npow2; i += blockDim.x) vals[i] = INFINITY;
    __syncthreads();
    for (int k = 2; k <= npow2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < npow2; i += blockDim.x) {
                const int p = i ^ j;
                if (p > i) {
                    const float x = vals[i], y = vals[p];
                    const bool up = (i & k) == 0;
                    if ((x > y) == up) {
                        vals[i] = y;
                        vals[p] = x;
                    }
                }
            }
            __syncthreads();
        }
    const int k_hi = n / 2, k_lo = (n & 1) ? n / 2 : n / 2 - 1;
    return (vals[k_lo] + vals[k_hi]) * 0.5f;
}

// ---- 1. per (group,row): rank-63 / rank-64 elements of v = f32(amp33) - med -----------------------
// one wave per row: the 128 values live two per lane (elements lane and lane+64) and are sorted by a
// bitonic network of cross-lane exchanges; rank 63 ends in (lane 63, slot 0), rank 64 in (lane 0, slot 1)
__global__ __launch_bounds__(256) void amp33_rows_kernel(const uint16_t *__restrict__ amp33,
                                                         const float *__restrict__ med, float *__restrict__ lohi,
                                                         int ny, int nrows_total) {
    const int lane = threadIdx.x & 63;
    const int row_id = blockIdx.x * 4 + (threadIdx.x >> 6);  // flattened (g, r)
    if (row_id >= nrows_total) return;
    const int r = row_id % ny;
    const uint16_t *a = amp33 + (size_t)row_id * RIP_CW;
    const float *m = med + (size_t)r * RIP_CW;
    float v0 = (float)a[lane] - m[lane];
    float v1 = (float)a[lane + 64] - m[lane + 64];
#pragma unroll
    for (int k = 2; k <= 128; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j == 64) {  // partner = the other slot of this lane; k = 128: ascending everywhere
                const float lo = fminf(v0, v1), hi = fmaxf(v0, v1);
                v0 = lo;
                v1 = hi;
            } else {
                const float p0 = __shfl_xor(v0, j, 64), p1 = __shfl_xor(v1, j, 64);
                const bool lower = (lane & j) == 0;
                const bool up0 = (lane & k) == 0, up1 = ((lane + 64) & k) == 0;
                v0 = (lower == up0) ? fminf(v0, p0) : fmaxf(v0, p0);
                v1 = (lower == up1) ? fminf(v1, p1) : fmaxf(v1, p1);
            }
        }
    }
    float *o = lohi + (size_t)row_id * 2;
    if (lane == 63) o[0] = v0;
    if (lane == 0) o[1] = v1;
}

// ---- 2. exact selection of two ranks among the ny*128 values of each group (3-level radix) -------
struct SelState {
    uint32_t prefix[2];
    uint32_t rank[2];
};


__global__ __launch_bounds__(256) void sel_hist_kernel(const uint16_t *__restrict__ amp33, const float *__restrict__ med,
                                                       const SelState *__restrict__ st, uint32_t *__restrict__ ghist,
                                                       int ny, int level, int chunk) {
    __shared__ uint32_t h[2][SEL_BINS];
    const int g = blockIdx.y;
    for (int i = threadIdx.x; i < 2 * SEL_BINS; i += blockDim.x) (&h[0][0])[i] = 0;
    __syncthreads();
    const size_t n = (size_t)ny * RIP_CW;
    const size_t lo = (size_t)blockIdx.x * chunk;
    const size_t hi = (lo + chunk < n) ? lo + chunk : n;
    const int shift = sel_shift(level);
    const uint32_t mask = (1u << sel_bits(level)) - 1u;
    const int above = shift + sel_bits(level);  // bits above this level's field
    const uint32_t p0 = level ? st[g].prefix[0] : 0u, p1 = level ? st[g].prefix[1] : 0u;   // level 0: no prefix yet (st not read)
    for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const uint32_t key = f2key((float)amp33[(size_t)g * n + i] - med[i]);
        const uint32_t bin = (key >> shift) & mask;
        const bool m0 = (above >= 32) || (((key ^ p0) >> above) == 0);
        const bool m1 = (above >= 32) || (((key ^ p1) >> above) == 0);
        if (m0) atomicAdd(&h[0][bin], 1u);
        if (m1) atomicAdd(&h[1][bin], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * SEL_BINS; i += blockDim.x) {
        const uint32_t c = (&h[0][0])[i];
        if (c) atomicAdd(&ghist[(size_t)g * 2 * SEL_BINS + i], c);
    }
}

__global__ __launch_bounds__(256) void sel_scan_kernel(SelState *__restrict__ st, uint32_t *__restrict__ ghist, int level, uint32_t n) {
    // bin holding the wanted rank: 8 bins per thread, wave prefix sums over the 256 partial sums, the owner walks its 8 bins
    __shared__ uint32_t part[4];
    const int g = blockIdx.x, t = blockIdx.y;
    uint32_t *h = ghist + ((size_t)g * 2 + t) * SEL_BINS;
    const int per = SEL_BINS / 256;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // level 0 starts the selection: the two middle elements of n values, empty prefix (no separate initialisation launch)
    const uint32_t rank = level ? st[g].rank[t] : (t ? n / 2 : n / 2 - 1);
    const uint32_t before = level ? st[g].prefix[t] : 0u;
    uint32_t own = 0;
    for (int k = 0; k < per; ++k) own += h[tid * per + k];
    uint32_t incl = own;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t y = __shfl_up(incl, off, 64);
        if (lane >= off) incl += y;
    }
    if (lane == 63) part[w] = incl;
    __syncthreads();
    for (int k = 0; k < w; ++k) incl += part[k];
    const uint32_t excl = incl - own;
    if ((excl <= rank && rank < incl) || (tid == 255 && rank >= incl)) {
        uint32_t cum = excl;
        int b = tid * per;
        while (b < tid * per + per - 1 && cum + h[b] <= rank) cum += h[b++];
        st[g].prefix[t] = before | ((uint32_t)b << sel_shift(level));
        st[g].rank[t] = rank - cum;
    }
    __syncthreads();
    for (int k = 0; k < per; ++k) h[tid * per + k] = 0;  // ready for the next level
}

// np.median of vals[0..n) in LDS by a three-level radix selection inside one block (11 + 11 + 10 key bits, two LDS histograms:
// one per middle element): 3 x (count, scan) instead of the 78 compare-exchange passes of a bitonic sort of 4096 values.
// The median of an even count is the f32 mean of the two middle elements, as np.median forms it.  All threads of the block
// call it; h: 2 x SEL_BINS words, tmp: 8 words of LDS.
__device__ float block_median_select(const float *vals, int n, uint32_t (*h)[SEL_BINS], uint32_t *tmp) {
    uint32_t prefix[2] = {0u, 0u};
    uint32_t rank[2] = {(uint32_t)((n & 1) ? n / 2 : n / 2 - 1), (uint32_t)(n / 2)};
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    for (int lv = 0; lv < 3; ++lv) {
        for (int i = t; i < 2 * SEL_BINS; i += blockDim.x) (&h[0][0])[i] = 0;
        __syncthreads();
        const int shift = sel_shift(lv), above = shift + sel_bits(lv);
        const uint32_t mask = (1u << sel_bits(lv)) - 1u;
        for (int i = t; i < n; i += blockDim.x) {
            const uint32_t key = f2key(vals[i]);
            const uint32_t bin = (key >> shift) & mask;
            if (above >= 32 || ((key ^ prefix[0]) >> above) == 0) atomicAdd(&h[0][bin], 1u);
            if (above >= 32 || ((key ^ prefix[1]) >> above) == 0) atomicAdd(&h[1][bin], 1u);
        }
        __syncthreads();
        for (int q = 0; q < 2; ++q) {
            // parallel scan of the 2048 bins: 8 per thread for the first 256 threads, wave prefix sums, then the owner walks its 8
            const int per = SEL_BINS / 256;
            uint32_t own = 0, incl = 0;
            if (t < 256) {
                for (int k = 0; k < per; ++k) own += h[q][t * per + k];
                incl = own;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const uint32_t y = __shfl_up(incl, off, 64);
                    if (lane >= off) incl += y;
                }
                if (lane == 63) tmp[w] = incl;
            }
            __syncthreads();
            if (t < 256) {
                for (int k = 0; k < w; ++k) incl += tmp[k];
                const uint32_t excl = incl - own;
                if ((excl <= rank[q] && rank[q] < incl) || (t == 255 && rank[q] >= incl)) {
                    uint32_t cum = excl;
                    int b = t * per;
                    while (b < t * per + per - 1 && cum + h[q][b] <= rank[q]) cum += h[q][b++];
                    tmp[4] = (uint32_t)b;
                    tmp[5] = rank[q] - cum;
                }
            }
            __syncthreads();
            prefix[q] |= tmp[4] << shift;
            rank[q] = tmp[5];
            __syncthreads();
        }
    }
    return (key2f(prefix[0]) + key2f(prefix[1])) * 0.5f;
}

// ---- 3. per group: global median M, row medians, ctr, rowcorr ------------------------------------
__global__ __launch_bounds__(1024) void rowcorr_kernel(const SelState *__restrict__ st, const float *__restrict__ lohi,
                                                       double slope, double *__restrict__ rowcorr,
                                                       double *__restrict__ rowcorr_t, float *__restrict__ dbg_refmed,
                                                       float *__restrict__ dbg_scal, int ny, int npow2) {
    extern __shared__ float rm[];  // [npow2] sort buffer
    const int g = blockIdx.x;
    const float M = (key2f(st[g].prefix[0]) + key2f(st[g].prefix[1])) * 0.5f;  // np.median of the block
    auto refmed = [&](int r) {
        const float a = lohi[((size_t)g * ny + r) * 2] - M;
        const float b = lohi[((size_t)g * ny + r) * 2 + 1] - M;
        return (a + b) * 0.5f;
    };
    __shared__ uint32_t hsel[2][SEL_BINS];
    __shared__ uint32_t tmp[8];
    for (int r = threadIdx.x; r < ny; r += blockDim.x) rm[r] = refmed(r);
    __syncthreads();
    const float ctr = block_median_select(rm, ny, hsel, tmp);
    for (int r = threadIdx.x; r < ny; r += blockDim.x) {
        const float v = refmed(r);
        rowcorr[(size_t)g * ny + r] = slope * (double)(v - ctr);
        if (rowcorr_t) rowcorr_t[(size_t)r * gridDim.x + g] = slope * (double)(v - ctr);  // [row][group]: one scalar load per row
        if (dbg_refmed) dbg_refmed[(size_t)g * ny + r] = v;
    }
    if (dbg_scal && threadIdx.x == 0) {
        dbg_scal[g * 2] = M;
        dbg_scal[g * 2 + 1] = ctr;
    }
}

// ---- 4. per (group, channel): bottom/top medians of the row-corrected image, line fit ---------
template <typename DT>
__global__ __launch_bounds__(1024) void chan_kernel(const DT *__restrict__ data, const float *__restrict__ dark,
                                                    const double *__restrict__ rowcorr,
                                                    const double *__restrict__ lines_override,
                                                    double *__restrict__ lines, float *__restrict__ dbg_bt, int ny,
                                                    int nx) {
    __shared__ float v[1024];
    const int ch = blockIdx.x, g = blockIdx.y, nch = gridDim.x;
    const int e = threadIdx.x & 511, half = threadIdx.x >> 9;
    const int row = (half ? ny - 4 : 0) + e / RIP_CW;
    const size_t idx = ((size_t)g * ny + row) * nx + (size_t)ch * RIP_CW + e % RIP_CW;
    float val = (float)data[idx] - dark[idx];
    val = (float)((double)val - rowcorr[(size_t)g * ny + row]);
    v[threadIdx.x] = val;
    __syncthreads();
    // both halves at once: the bitonic network over the 1024 values stopped at runs of 512 leaves the bottom rows' values
    // ascending in v[0:512] and the top rows' descending in v[512:1024]; the two middle elements sit at 255, 256 either way
    // (45 compare-exchange steps per thread instead of 512 rank comparisons: a third of the kernel's time)
    for (int k = 2; k <= 512; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int i = threadIdx.x, p = i ^ j;
            if (p > i) {
                const float x = v[i], y = v[p];
                const bool up = (i & k) == 0;
                if ((x > y) == up) {
                    v[i] = y;
                    v[p] = x;
                }
            }
            __syncthreads();
        }
    if (threadIdx.x == 0) {
        const float b = (v[255] + v[256]) * 0.5f;
        const float t = (v[512 + 255] + v[512 + 256]) * 0.5f;
        double m, c;
        if (lines_override) {
            m = lines_override[((size_t)g * nch + ch) * 2];
            c = lines_override[((size_t)g * nch + ch) * 2 + 1];
        } else {
            m = ((double)t - (double)b) / (double)(ny - 4);
            c = (double)b - 1.5 * m;
        }
        lines[((size_t)g * nch + ch) * 2] = m;
        lines[((size_t)g * nch + ch) * 2 + 1] = c;
        if (dbg_bt) {
            dbg_bt[((size_t)g * nch + ch) * 2] = b;
            dbg_bt[((size_t)g * nch + ch) * 2 + 1] = t;
        }
    }
}

int rip_launch_refpix_prepass(rip_ctx *ctx, const RefpixArgs &a) {
    if (a.nx % RIP_CW) return rip_fail(ctx, RIP_EINVAL, "refpix: nx=%d is not a multiple of 128", a.nx);
    if (a.ny < 8) return rip_fail(ctx, RIP_EINVAL, "refpix: ny=%d too small", a.ny);
    const int G = a.ngrp, ny = a.ny, nch = a.nx / RIP_CW;
    hipStream_t strm = a.stream ? a.stream : ctx->stream;   // (the overlapped pre-pass: the context's second stream)
    if ((ctx->prepass_form == 1 || (ctx->prepass_form < 0 && !a.background)) && rip_refpix_one_supported(a))
        return rip_launch_refpix_one(ctx, a);
    if (a.amp33) {
        // scratch: lohi (G,ny,2) f32 | SelState[G]
        const size_t lohi_b = (size_t)G * ny * 2 * sizeof(float);
        const size_t st_b = ((size_t)G * sizeof(SelState) + 255) / 256 * 256;
        char *ws = (char *)rip_ws(ctx, 4, lohi_b + st_b);
        if (!ws) return RIP_ENOMEM;
        float *lohi = (float *)ws;
        SelState *st = (SelState *)(ws + lohi_b);
        // the selection histograms have a workspace slot of their own, of a fixed size: every level's scan leaves them zero, so
        // they are cleared only when the slot is first allocated (no initialisation launch per call)
        const size_t gh_b = (size_t)RIP_MAX_GROUPS * 2 * SEL_BINS * sizeof(uint32_t);
        const void *had = ctx->ws[13];
        uint32_t *ghist = (uint32_t *)rip_ws(ctx, 13, gh_b);
        if (!ghist) return RIP_ENOMEM;
        if ((const void *)ghist != had) RIP_HIP(ctx, hipMemsetAsync(ghist, 0, gh_b, strm));
        const uint32_t n = (uint32_t)ny * RIP_CW;
        hipLaunchKernelGGL(amp33_rows_kernel, dim3((unsigned)((G * ny + 3) / 4)), dim3(256), 0, strm, a.amp33,
                           a.amp33_med, lohi, ny, G * ny);
        const int chunk = 8192;
        const unsigned nblk = (unsigned)((n + chunk - 1) / chunk);
        for (int level = 0; level < 3; ++level) {
            hipLaunchKernelGGL(sel_hist_kernel, dim3(nblk, G), dim3(256), 0, strm, a.amp33, a.amp33_med, st,
                               ghist, ny, level, chunk);
            hipLaunchKernelGGL(sel_scan_kernel, dim3(G, 2), dim3(256), 0, strm, st, ghist, level, n);
        }
        int npow2 = 1;
        while (npow2 < ny) npow2 <<= 1;
        const size_t lds = (size_t)npow2 * sizeof(float);
        if (lds > 48 * 1024)
            RIP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(rowcorr_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(rowcorr_kernel, dim3(G), dim3(1024), lds, strm, st, lohi, a.slope, a.rowcorr,
                           a.rowcorr_t, (float *)nullptr, (float *)nullptr, ny, npow2);
    } else {
        // no reference output in the read file: the row step is the identity
        RIP_HIP(ctx, hipMemsetAsync(a.rowcorr, 0, (size_t)G * ny * sizeof(double), strm));
        if (a.rowcorr_t) RIP_HIP(ctx, hipMemsetAsync(a.rowcorr_t, 0, (size_t)G * ny * sizeof(double), strm));
    }
    if (a.data_dtype == RIP_U16)
        hipLaunchKernelGGL(chan_kernel<uint16_t>, dim3(nch, G), dim3(1024), 0, strm, (const uint16_t *)a.data,
                           a.dark_data, a.rowcorr, a.lines_override, a.lines, (float *)nullptr, ny, a.nx);
    else
        hipLaunchKernelGGL(chan_kernel<float>, dim3(nch, G), dim3(1024), 0, strm, (const float *)a.data,
                           a.dark_data, a.rowcorr, a.lines_override, a.lines, (float *)nullptr, ny, a.nx);
    RIP_HIP(ctx, hipGetLastError());
    return RIP_OK;
}

// ------------------------------------------------------------------ image-level drop-ins
// ref_subtraction_row(image, use_ref_channel=True, slope) / ref_subtraction_channel(image,
// use_ref_channel=True) on one (ny, nx+128) f32 image resident on the device.

__global__ __launch_bounds__(RIP_CW) void img_rowmed_kernel(const float *__restrict__ image, float *__restrict__ ref_med,
                                                            int nx) {
    __shared__ float v[RIP_CW];
    __shared__ float slot[2];
    const int r = blockIdx.x;
    v[threadIdx.x] = image[(size_t)r * (nx + RIP_CW) + nx + threadIdx.x];
    __syncthreads();
    const float m = block_median(v, RIP_CW, slot);
    if (threadIdx.x == 0) ref_med[r] = m;
}

__global__ __launch_bounds__(1024) void img_ctr_kernel(const float *__restrict__ ref_med, double slope,
                                                       double *__restrict__ rowcorr, float *__restrict__ ctr_out, int ny,
                                                       int npow2) {
    extern __shared__ float rm[];
    for (int r = threadIdx.x; r < ny; r += blockDim.x) rm[r] = ref_med[r];
    const float ctr = block_median_sorted(rm, ny, npow2);
    for (int r = threadIdx.x; r < ny; r += blockDim.x) rowcorr[r] = slope * (double)(ref_med[r] - ctr);
    if (threadIdx.x == 0 && ctr_out) *ctr_out = ctr;
}

__global__ void img_rowapply_kernel(float *__restrict__ image, const double *__restrict__ rowcorr, int ny, int w) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)ny * w) return;
    image[i] = (float)((double)image[i] - rowcorr[i / w]);
}

__global__ __launch_bounds__(1024) void img_chan_kernel(const float *__restrict__ image,
                                                        const double *__restrict__ lines_override,
                                                        double *__restrict__ lines, float *__restrict__ bt_out, int ny,
                                                        int w) {
    __shared__ float v[1024];
    __shared__ float lh[2][2];
    const int ch = blockIdx.x;
    const int e = threadIdx.x & 511, half = threadIdx.x >> 9;
    const int row = (half ? ny - 4 : 0) + e / RIP_CW;
    const float val = image[(size_t)row * w + (size_t)ch * RIP_CW + e % RIP_CW];
    v[threadIdx.x] = val;
    __syncthreads();
    const float *mine = v + half * 512;
    int rank = 0;
    for (int j = 0; j < 512; ++j) {
        const float q = mine[j];
        rank += (q < val || (q == val && j < e)) ? 1 : 0;
    }
    if (rank == 255) lh[half][0] = val;
    if (rank == 256) lh[half][1] = val;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float b = (lh[0][0] + lh[0][1]) * 0.5f;
        const float t = (lh[1][0] + lh[1][1]) * 0.5f;
        double m, c;
        if (lines_override) {
            m = lines_override[ch * 2];
            c = lines_override[ch * 2 + 1];
        } else {
            m = ((double)t - (double)b) / (double)(ny - 4);
            c = (double)b - 1.5 * m;
        }
        lines[ch * 2] = m;
        lines[ch * 2 + 1] = c;
        if (bt_out) {
            bt_out[ch * 2] = b;
            bt_out[ch * 2 + 1] = t;
        }
    }
}

__global__ void img_chanapply_kernel(float *__restrict__ image, const double *__restrict__ lines, int ny, int w) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)ny * w) return;
    const int r = (int)(i / w), c = (int)(i % w);
    const double *ln = lines + (size_t)(c / RIP_CW) * 2;
    const double iel = ln[0] * (double)r + ln[1];
    image[i] = (float)((double)image[i] - iel);
}

int rip_refpix_image(rip_ctx *ctx, float *d_image, int ny, int nx, double slope, int do_row, int do_channel,
                     const double *d_lines, float *d_ref_med, float *d_ctr, float *d_bottom_top) {
    if (nx % RIP_CW) return rip_fail(ctx, RIP_EINVAL, "refpix: nx=%d is not a multiple of 128", nx);
    const int w = nx + RIP_CW, nch = w / RIP_CW;
    const size_t n = (size_t)ny * w;
    char *ws = (char *)rip_ws(ctx, 4, (size_t)ny * (sizeof(double) + sizeof(float)) + (size_t)nch * 2 * sizeof(double) + 256);
    if (!ws) return RIP_ENOMEM;
    double *rowcorr = (double *)ws;
    double *lines = rowcorr + ny;
    float *refmed = (float *)(lines + nch * 2);
    if (do_row) {
        hipLaunchKernelGGL(img_rowmed_kernel, dim3(ny), dim3(RIP_CW), 0, ctx->stream, d_image, refmed, nx);
        int npow2 = 1;
        while (npow2 < ny) npow2 <<= 1;
        const size_t lds = (size_t)npow2 * sizeof(float);
        if (lds > 48 * 1024)
            RIP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(img_ctr_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(img_ctr_kernel, dim3(1), dim3(1024), lds, ctx->stream, refmed, slope, rowcorr, d_ctr, ny,
                           npow2);
        hipLaunchKernelGGL(img_rowapply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_image,
                           rowcorr, ny, w);
        if (d_ref_med)
            RIP_HIP(ctx, hipMemcpyAsync(d_ref_med, refmed, (size_t)ny * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    }
    if (do_channel) {
        hipLaunchKernelGGL(img_chan_kernel, dim3(nch), dim3(1024), 0, ctx->stream, d_image, d_lines, lines, d_bottom_top,
                           ny, w);
        hipLaunchKernelGGL(img_chanapply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_image,
                           lines, ny, w);
    }
    RIP_HIP(ctx, hipGetLastError());
    return RIP_OK;
}

// ------------------------------------------------------------------ general forms of the image-level drop-ins
// reference_subtraction.py with ANY of its arguments (the forms above are the configuration of calibrateimage):
//   ref_subtraction_row(image, use_ref_channel=False, slope=None)      :77-125  row medians of the 4+4 border pixels or of
//       the reference output, of the science pixels (for the np.polyfit of slope=None, done by the host mirror), update
//       in the dtype numpy's promotion gives: f64 for a numpy f64 slope, f32 for a Python float / numpy f32 slope
//   ref_subtraction_channel(image, channel_start, channel_end, use_ref_channel) :16-74  windows [start+128k, end+128k),
//       one after the other as the reference's loop runs them (windows wider than 128 columns overlap)

// median of the values of `nr` rows x (cols [c0, c0+n0) u [c1, c1+n1)) per block: block b starts at row row0 + b * rstep.
// Dynamic LDS: npow2 floats.
__global__ __launch_bounds__(1024) void img_window_median_kernel(const float *__restrict__ image, int w, int row0, int rstep,
                                                                 int nr, int c0, int n0, int c1, int n1,
                                                                 float *__restrict__ out, int npow2) {
    extern __shared__ float mv[];
    const int per = n0 + n1, n = nr * per;
    const int rbase = row0 + (int)blockIdx.x * rstep;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int rr = i / per, cc = i % per;
        const int col = cc < n0 ? c0 + cc : c1 + (cc - n0);
        mv[i] = image[(size_t)(rbase + rr) * w + col];
    }
    __syncthreads();
    const float m = block_median_sorted(mv, n, npow2);
    if (threadIdx.x == 0) out[blockIdx.x] = m;
}

// ctr = median of the row medians; corr[r] = slope * f64(f32(ref_med[r] - ctr)) (f64 form) or f32(slope) * (ref_med[r] - ctr)
// rounded to f32 (f32 form)
__global__ __launch_bounds__(1024) void img_ctr2_kernel(const float *__restrict__ ref_med, double slope, int f32_form,
                                                        double *__restrict__ rowcorr, float *__restrict__ ctr_out, int ny,
                                                        int npow2) {
    extern __shared__ float rm[];
    for (int r = threadIdx.x; r < ny; r += blockDim.x) rm[r] = ref_med[r];
    const float ctr = block_median_sorted(rm, ny, npow2);
    const float s32 = (float)slope;
    for (int r = threadIdx.x; r < ny; r += blockDim.x) {
        const float d = ref_med[r] - ctr;
        rowcorr[r] = f32_form ? (double)(s32 * d) : slope * (double)d;
    }
    if (threadIdx.x == 0 && ctr_out) *ctr_out = ctr;
}

__global__ void img_rowapply_f32_kernel(float *__restrict__ image, const double *__restrict__ rowcorr, int ny, int w) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)ny * w) return;
    image[i] = image[i] - (float)rowcorr[i / w];
}

// line through (1.5, b), (ny - 2.5, t) of one window (or the caller's), then the update of its columns
__global__ void img_line1_kernel(const float *__restrict__ bt, const double *__restrict__ line_override,
                                 double *__restrict__ line, float *__restrict__ bt_out, int ny) {
    const float b = bt[0], t = bt[1];
    double m, c;
    if (line_override) {
        m = line_override[0];
        c = line_override[1];
    } else {
        m = ((double)t - (double)b) / (double)(ny - 4);
        c = (double)b - 1.5 * m;
    }
    line[0] = m;
    line[1] = c;
    if (bt_out) {
        bt_out[0] = b;
        bt_out[1] = t;
    }
}

__global__ void img_winapply_kernel(float *__restrict__ image, const double *__restrict__ line, int ny, int w, int c0,
                                    int ncols) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)ny * ncols) return;
    const int r = (int)(i / ncols), c = c0 + (int)(i % ncols);
    const double iel = line[0] * (double)r + line[1];
    float *p = image + (size_t)r * w + c;
    *p = (float)((double)*p - iel);
}

static int pow2_at_least(int n) {
    int p = 1;
    while (p < n) p <<= 1;
    return p;
}

template <typename K>
static int with_lds(rip_ctx *ctx, K kernel, size_t lds) {
    if (lds > 48 * 1024)
        RIP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    return RIP_OK;
}

int rip_refpix_row_general(rip_ctx *ctx, float *d_image, int ny, int width, int nside, int use_ref_channel, int mode,
                           double slope, float *d_ref_med, float *d_sci_med, float *d_ctr) {
    if (nside < 16 || nside > width || (use_ref_channel && nside + RIP_CW > width))
        return rip_fail(ctx, RIP_EINVAL, "refpix row: nside=%d does not fit an image %d wide", nside, width);
    if (nside - 8 > 32768 || ny > 32768) return rip_fail(ctx, RIP_EINVAL, "refpix row: frame too large (%d x %d)", ny, nside);
    char *ws = (char *)rip_ws(ctx, 4, (size_t)ny * (sizeof(double) + 2 * sizeof(float)) + 256);
    if (!ws) return RIP_ENOMEM;
    double *rowcorr = (double *)ws;
    float *refmed = (float *)(rowcorr + ny), *scimed = refmed + ny;
    int rc;
    {   // reference medians: the reference output, or the 4 + 4 border pixels of the row (reference_subtraction.py:108-111)
        const int n = use_ref_channel ? RIP_CW : 8, np2 = pow2_at_least(n);
        if ((rc = with_lds(ctx, img_window_median_kernel, (size_t)np2 * 4))) return rc;
        if (use_ref_channel)
            hipLaunchKernelGGL(img_window_median_kernel, dim3(ny), dim3(1024), (size_t)np2 * 4, ctx->stream, d_image, width, 0, 1, 1,
                               nside, RIP_CW, 0, 0, refmed, np2);
        else
            hipLaunchKernelGGL(img_window_median_kernel, dim3(ny), dim3(1024), (size_t)np2 * 4, ctx->stream, d_image, width, 0, 1, 1,
                               0, 4, nside - 4, 4, refmed, np2);
    }
    if (d_sci_med) {   // science medians (:107), only needed for the polyfit of slope=None
        const int n = nside - 8, np2 = pow2_at_least(n);
        if ((rc = with_lds(ctx, img_window_median_kernel, (size_t)np2 * 4))) return rc;
        hipLaunchKernelGGL(img_window_median_kernel, dim3(ny), dim3(1024), (size_t)np2 * 4, ctx->stream, d_image, width, 0, 1, 1, 4,
                           n, 0, 0, scimed, np2);
        RIP_HIP(ctx, hipMemcpyAsync(d_sci_med, scimed, (size_t)ny * 4, hipMemcpyDeviceToDevice, ctx->stream));
    }
    {
        const int np2 = pow2_at_least(ny);
        if ((rc = with_lds(ctx, img_ctr2_kernel, (size_t)np2 * 4))) return rc;
        hipLaunchKernelGGL(img_ctr2_kernel, dim3(1), dim3(1024), (size_t)np2 * 4, ctx->stream, refmed, slope, mode == 2 ? 1 : 0,
                           rowcorr, d_ctr, ny, np2);
    }
    if (d_ref_med) RIP_HIP(ctx, hipMemcpyAsync(d_ref_med, refmed, (size_t)ny * 4, hipMemcpyDeviceToDevice, ctx->stream));
    const size_t n = (size_t)ny * width;
    if (mode == 1)
        hipLaunchKernelGGL(img_rowapply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_image, rowcorr, ny,
                           width);
    else if (mode == 2)
        hipLaunchKernelGGL(img_rowapply_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_image, rowcorr,
                           ny, width);
    RIP_HIP(ctx, hipGetLastError());
    return RIP_OK;
}

int rip_refpix_channel_general(rip_ctx *ctx, float *d_image, int ny, int width, int channel_start, int channel_end, int nchan,
                               const double *d_lines, float *d_bottom_top) {
    const int cw = channel_end - channel_start;
    if (ny < 8 || cw < 1 || channel_start < 0 || nchan < 1 || channel_end + (nchan - 1) * RIP_CW > width)
        return rip_fail(ctx, RIP_EINVAL, "refpix channel: windows [%d,%d) + 128 k, k < %d do not fit an image %d wide", channel_start,
                        channel_end, nchan, width);
    if (4 * cw > 32768) return rip_fail(ctx, RIP_EINVAL, "refpix channel: window of %d columns is too wide", cw);
    char *ws = (char *)rip_ws(ctx, 4, 64);
    if (!ws) return RIP_ENOMEM;
    double *line = (double *)ws;
    float *bt = (float *)(line + 2);
    const int np2 = pow2_at_least(4 * cw);
    int rc;
    if ((rc = with_lds(ctx, img_window_median_kernel, (size_t)np2 * 4))) return rc;
    const size_t n = (size_t)ny * cw;
    for (int k = 0; k < nchan; ++k) {   // in the reference's order: a window sees the updates of the windows before it
        const int c0 = channel_start + k * RIP_CW;
        // two blocks: rows 0:4 and ny-4:ny
        hipLaunchKernelGGL(img_window_median_kernel, dim3(2), dim3(1024), (size_t)np2 * 4, ctx->stream, d_image, width, 0, ny - 4, 4,
                           c0, cw, 0, 0, bt, np2);
        hipLaunchKernelGGL(img_line1_kernel, dim3(1), dim3(1), 0, ctx->stream, bt, d_lines ? d_lines + 2 * k : nullptr, line,
                           d_bottom_top ? d_bottom_top + 2 * k : nullptr, ny);
        hipLaunchKernelGGL(img_winapply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_image, line, ny,
                           width, c0, cw);
    }
    RIP_HIP(ctx, hipGetLastError());
    return RIP_OK;
}
