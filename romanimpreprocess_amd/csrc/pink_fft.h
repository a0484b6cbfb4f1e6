// The complex-to-real transform of the 1/f frames (pink.hip) for power-of-two frame lengths, hand-written: two passes over the data
// where the library's plan for 2^20 points makes seven (a pre-processing kernel, three transposes, two row transforms; 4.0 ms for
// 64 frames, 3.3 of them in the transposes -- profiles/r04_config5_kernels_start_of_round.txt: beside the other kernels of a realisation; alone 1.06 ms, r04_pink_form_ab.txt).
//
//   x[n], n = 0 .. L-1, real, from the folded coefficients S_0 .. S_N (N = L/2; pink.hip's head):  with y[m] = x[2m] + i x[2m+1]
//       y[m] = sum_{j<N} W_j e^{+2 pi i j m / N},   W_j = (S_j + conj S_{N-j}) + i w^j (S_j - conj S_{N-j}),  w = e^{2 pi i / L}
//   (the even samples transform X_j + X_{j+N}, the odd ones (X_j - X_{j+N}) w^j, X the Hermitian extension of S), and only
//   x[0 : N] is kept by the caller: y[0 : N/2].  The N-point transform is cut N = n1 * n2 (512 x 1024 for the 4096 x 128 frame):
//       j = j1 n2 + j2,  m = k1 + n1 k2:   y[m] = sum_{j2} [ w_N^{j2 k1} sum_{j1} W_j w_n1^{j1 k1} ] w_n2^{j2 k2}
//   pass 1 (pf_cols_kernel): a workgroup takes 8 adjacent columns j2 of a frame (128-byte segments of every row j1) into LDS,
//       transforms them along j1 (radix-4 decimation in frequency, in place), multiplies by w_N^{j2 k1} and writes A[k1][j2] back
//       over the columns it read -- the transpose of the library's plan is the LDS tile;
//   pass 2 (pf_rows_kernel): a workgroup takes 8 adjacent rows k1 (contiguous), transforms them along j2 and writes y[k1 + n1 k2]
//       for k2 < n2 / 2: 8 consecutive k1 are 128 contiguous bytes of x.
//   The folded coefficients never exist: pf_fill_kernel forms W_j and W_{N-j} from the four deviate pairs they share.
// HBM traffic per frame of 2^20 points: 8.4 MB written by the fill, 8.4 + 8.4 by pass 1, 8.4 + 4.2 by pass 2 (the library's plan:
// 8.4 + 6 x 16.8).  f64 throughout, twiddle factors from tables made on the host in long double; the operations are not the
// library's (nor numpy's pocketfft): results agree to rounding, as before.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <vector>

namespace pf {

constexpr int TW = 8;   // columns (pass 1) / rows (pass 2) of a workgroup's tile: 128-byte segments

struct Dev {
    int n1, n2;                 // N = n1 * n2
    int lg1, lg2;               // their logarithms
    int split_lg;               // w_N^q = twa[q >> split_lg] * twb[q & ((1 << split_lg) - 1)]
    const double2 *tw1, *tw2;   // e^{2 pi i k / n1}, e^{2 pi i k / n2}
    const double2 *twa, *twb;
    const int *rev1, *rev2;     // output index held by position p after the in-place transform of n1 / n2 points
};

__device__ __forceinline__ double2 cmul(double2 a, double2 b) {   // (fused: four operations instead of six, one rounding fewer)
    return make_double2(__builtin_fma(a.x, b.x, -(a.y * b.y)), __builtin_fma(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ double2 cadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 csub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ double2 cmuli(double2 a) { return make_double2(-a.y, a.x); }   // i * a

// In-place decimation-in-frequency transform (sign +, unnormalised) of `lines` sequences of n points held in LDS: element p of
// line l at buf[l * ls + p * ps].  Radix 4 while four points remain, then one radix-2 stage when log2 n is odd.  Afterwards
// position p holds the output of index rev[p] (the digits of p reversed; table from the host).  LINE_FAST: how the butterflies
// of a stage are dealt to the threads -- lines fastest (pass 1: the line index is the contiguous one) or butterflies fastest.
template <bool LINE_FAST, int LINES>
__device__ __forceinline__ void fft_lds(double2 *buf, const double2 *tw, int lgn, int ls, int ps, int tid, int nthreads) {
    // (every size is a power of two: shifts and masks, no integer division in the loops)
    const int n = 1 << lgn;
    int lglen = lgn;
    for (; lglen >= 2; lglen -= 2) {
        const int lgq = lglen - 2, q = 1 << lgq, lgnb = lgn - 2, lgt = lgn - lglen;
        for (int t = tid; t < (LINES << lgnb); t += nthreads) {
            const int l = LINE_FAST ? t % LINES : t >> lgnb, b = LINE_FAST ? t / LINES : t & ((1 << lgnb) - 1);
            const int k = b & (q - 1), base = (b >> lgq) << lglen;
            double2 *p0 = buf + l * ls + (base + k) * ps;
            const double2 a0 = p0[0], a1 = p0[q * ps], a2 = p0[2 * q * ps], a3 = p0[3 * q * ps];
            const double2 t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = cmuli(csub(a1, a3));
            p0[0] = cadd(t0, t2);
            if (q > 1) {
                p0[q * ps] = cmul(cadd(t1, t3), tw[k << lgt]);
                p0[2 * q * ps] = cmul(csub(t0, t2), tw[(2 * k) << lgt]);
                p0[3 * q * ps] = cmul(csub(t1, t3), tw[(3 * k) << lgt]);
            } else {
                p0[q * ps] = cadd(t1, t3);
                p0[2 * q * ps] = csub(t0, t2);
                p0[3 * q * ps] = csub(t1, t3);
            }
        }
        __syncthreads();
    }
    if (lglen == 1) {
        const int lgnb = lgn - 1;
        for (int t = tid; t < (LINES << lgnb); t += nthreads) {
            const int l = LINE_FAST ? t % LINES : t >> lgnb, b = LINE_FAST ? t / LINES : t & ((1 << lgnb) - 1);
            double2 *p0 = buf + l * ls + (2 * b) * ps;
            const double2 a0 = p0[0], a1 = p0[ps];
            p0[0] = cadd(a0, a1);
            p0[ps] = csub(a0, a1);
        }
        __syncthreads();
    }
    (void)n;
}

constexpr int NT1 = 512, NT2 = 1024;   // threads per workgroup: two workgroups per CU in pass 1 (72 KB each), one in pass 2 (144 KB)

// pass 1: grid (n2 / TW, frames); dynamic LDS (n1 * TW + n1) complex
__global__ __launch_bounds__(NT1) void pf_cols_kernel(double2 *__restrict__ data, Dev d) {
    extern __shared__ double2 pf_lds[];
    const int n1 = d.n1, n2 = d.n2, tid = threadIdx.x;
    double2 *tile = pf_lds, *tw = pf_lds + (size_t)n1 * TW;
    double2 *frame = data + (size_t)blockIdx.y * n1 * n2;
    const int c0 = blockIdx.x * TW;
    for (int i = tid; i < n1 * TW; i += NT1) tile[i] = frame[((size_t)(i / TW) << d.lg2) + c0 + i % TW];
    for (int i = tid; i < n1; i += NT1) tw[i] = d.tw1[i];
    __syncthreads();
    fft_lds<true, TW>(tile, tw, d.lg1, 1, TW, tid, NT1);
    const int smask = (1 << d.split_lg) - 1;
    for (int i = tid; i < n1 * TW; i += NT1) {
        const int k1 = d.rev1[i / TW], j2 = c0 + i % TW;
        const long q = (long)j2 * k1;   // < N
        const double2 w = cmul(d.twa[q >> d.split_lg], d.twb[q & smask]);
        frame[((size_t)k1 << d.lg2) + j2] = cmul(tile[i], w);
    }
}

// pass 2: grid (n1 / TW, frames); dynamic LDS (TW * (n2 + 1) + n2) complex.  x: frames of 2 N doubles, the first N written
__global__ __launch_bounds__(NT2) void pf_rows_kernel(const double2 *__restrict__ data, double *__restrict__ x, Dev d) {
    extern __shared__ double2 pf_lds[];
    const int n1 = d.n1, n2 = d.n2, tid = threadIdx.x, rs = n2 + 1;   // (padded rows: the stores below walk down the rows)
    double2 *tile = pf_lds, *tw = pf_lds + (size_t)TW * rs;
    const double2 *frame = data + (size_t)blockIdx.y * n1 * n2;
    const int r0 = blockIdx.x * TW;
    for (int i = tid; i < n2 * TW; i += NT2) tile[(i >> d.lg2) * rs + (i & (n2 - 1))] = frame[((size_t)r0 << d.lg2) + i];
    for (int i = tid; i < n2; i += NT2) tw[i] = d.tw2[i];
    __syncthreads();
    fft_lds<false, TW>(tile, tw, d.lg2, rs, 1, tid, NT2);
    double2 *out = reinterpret_cast<double2 *>(x + (size_t)blockIdx.y * 2 * n1 * n2);
    for (int i = tid; i < n2 * TW; i += NT2) {
        const int p = i / TW, rr = i % TW, k2 = d.rev2[p];
        if (k2 < n2 / 2) out[((size_t)k2 << d.lg1) + r0 + rr] = tile[rr * rs + p];
    }
}

// host side: tables of one frame length
struct Tables {
    Dev dev{};
    void *mem = nullptr;
    size_t lds1 = 0, lds2 = 0;
};

inline bool supported(size_t L) {
    if (L < 256 || (L & (L - 1))) return false;
    int lg = 0;
    while (((size_t)1 << lg) < L / 2) ++lg;
    return lg >= 6 && lg <= 20;   // n1, n2 in 8 .. 1024
}

inline std::vector<int> digit_reverse(int n) {
    // position p = sum_s d_s * n / (r_1 .. r_s) holds index sum_s d_s * (r_1 .. r_{s-1}): radix 4 first, a final radix 2 if needed
    std::vector<int> radices;
    for (int len = n; len >= 2;) {
        const int r = len >= 4 ? 4 : 2;
        radices.push_back(r);
        len /= r;
    }
    std::vector<int> rev(n);
    for (int p = 0; p < n; ++p) {
        int rest = p, span = n, weight = 1, idx = 0;
        for (int r : radices) {
            span /= r;
            const int dgt = rest / span;
            rest -= dgt * span;
            idx += dgt * weight;
            weight *= r;
        }
        rev[p] = idx;
    }
    return rev;
}

// fills `t` (device tables in one allocation); returns hipSuccess or the error
inline hipError_t make_tables(size_t L, Tables &t) {
    const size_t N = L / 2;
    int lg = 0;
    while (((size_t)1 << lg) < N) ++lg;
    const int lg1 = lg / 2, lg2 = lg - lg1;
    const int n1 = 1 << lg1, n2 = 1 << lg2;
    const int split_lg = lg < 10 ? lg : 10;
    const size_t nb = (size_t)1 << split_lg, na = N >> split_lg;
    std::vector<double2> h(n1 + n2 + na + nb);
    const long double two_pi = 6.283185307179586476925286766559005768L;
    auto unit = [&](long double frac) { return make_double2((double)cosl(two_pi * frac), (double)sinl(two_pi * frac)); };
    size_t o = 0;
    for (int k = 0; k < n1; ++k) h[o++] = unit((long double)k / n1);
    for (int k = 0; k < n2; ++k) h[o++] = unit((long double)k / n2);
    for (size_t k = 0; k < na; ++k) h[o++] = unit((long double)(k << split_lg) / (long double)N);
    for (size_t k = 0; k < nb; ++k) h[o++] = unit((long double)k / (long double)N);
    std::vector<int> r1 = digit_reverse(n1), r2 = digit_reverse(n2);
    const size_t bytes_c = h.size() * sizeof(double2), bytes = bytes_c + (size_t)(n1 + n2) * sizeof(int);
    hipError_t e = hipMalloc(&t.mem, bytes);
    if (e != hipSuccess) return e;
    char *base = (char *)t.mem;
    if ((e = hipMemcpy(base, h.data(), bytes_c, hipMemcpyHostToDevice)) != hipSuccess) return e;
    if ((e = hipMemcpy(base + bytes_c, r1.data(), n1 * sizeof(int), hipMemcpyHostToDevice)) != hipSuccess) return e;
    if ((e = hipMemcpy(base + bytes_c + n1 * sizeof(int), r2.data(), n2 * sizeof(int), hipMemcpyHostToDevice)) != hipSuccess) return e;
    const double2 *c = (const double2 *)base;
    t.dev.n1 = n1, t.dev.n2 = n2, t.dev.lg1 = lg1, t.dev.lg2 = lg2, t.dev.split_lg = split_lg;
    t.dev.tw1 = c, t.dev.tw2 = c + n1, t.dev.twa = c + n1 + n2, t.dev.twb = c + n1 + n2 + na;
    t.dev.rev1 = (const int *)(base + bytes_c);
    t.dev.rev2 = t.dev.rev1 + n1;
    t.lds1 = ((size_t)n1 * TW + n1) * sizeof(double2);
    t.lds2 = ((size_t)TW * (n2 + 1) + n2) * sizeof(double2);
    if (t.lds1 > 48 * 1024)
        if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(pf_cols_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)t.lds1)) != hipSuccess)
            return e;
    if (t.lds2 > 48 * 1024)
        if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(pf_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)t.lds2)) != hipSuccess)
            return e;
    return hipSuccess;
}

}   // namespace pf
