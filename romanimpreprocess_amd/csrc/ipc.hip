// Inter-pixel-capacitance operators.
//
// Replaces (reference file:line):
//   utils/ipc_linearity.py:37-99    ipc_fwd       -> fwd_at() / ipc_fwd_image_kernel
//   utils/ipc_linearity.py:102-142  ipc_rev       -> ipc_cube_kernel (order 2, LDS tiled) / image kernels (any order)
//   utils/ipc_linearity.py:145-186  correct_cube  -> ipc_cube_kernel over all groups of a cube
// Arithmetic recipe: oracle/ipc.py.  out[y,x] = sum in[y-dy,x-dx] * K[1+dy,1+dx,y-dy,x-dx], products
// rounded then added in the order centre,(1,0),(-1,0),(0,1),(0,-1),(1,1),(1,-1),(-1,1),(-1,-1); a source
// outside the active region contributes no term.  Working type = promote(f32, kernel dtype, gain dtype).
//
// The 4-D kernel has 9 coefficients that are unique to every pixel, so there is no operand reuse
// to exploit with matrix cores: this is a 5x5-footprint stencil bound by HBM (9 coefficient planes
// + the cube), tiled through LDS.  Per destination pixel the 9 coefficients are loaded ONCE into
// registers and reused for both Neumann iterations and for all groups of the cube.
#include "rip_common.h"

#define IPC_TW 64
#define IPC_TH 16
#define IPC_THREADS 256
#define IPC_XW (IPC_TW + 4)
#define IPC_XH (IPC_TH + 4)
#define IPC_OW (IPC_TW + 2)
#define IPC_OH (IPC_TH + 2)
#define IPC_RING (2 * IPC_OW + 2 * IPC_TH)  // 164 positions of the halo-1 ring

// (dy,dx) of term k (k = 0 is the centre); plane index in the (3,3) kernel = 3*(1+dy)+(1+dx)
__device__ __constant__ int8_t IPC_DY[9] = {0, 1, -1, 0, 0, 1, 1, -1, -1};
__device__ __constant__ int8_t IPC_DX[9] = {0, 0, 0, 1, -1, 1, -1, 1, -1};

template <typename A, typename B>
struct Promote {
    using type = float;
};
template <>
struct Promote<float, double> {
    using type = double;
};
template <>
struct Promote<double, float> {
    using type = double;
};
template <>
struct Promote<double, double> {
    using type = double;
};

// forward operator at one destination, reading the source image from an LDS tile `S` with row
// stride `stride`, centred at S[0]; kk[k] = coefficient of term k, valid bit k = source k is active
template <typename T, typename ST, typename KT>
__device__ __forceinline__ T fwd_at(const ST *S, int stride, const KT (&kk)[9], unsigned valid) {
    T acc = (T)S[0] * (T)kk[0];
#pragma unroll
    for (int k = 1; k < 9; ++k) {
        const int dy = (k == 1 || k == 5 || k == 6) ? 1 : (k == 2 || k == 7 || k == 8) ? -1 : 0;
        const int dx = (k == 3 || k == 5 || k == 7) ? 1 : (k == 4 || k == 6 || k == 8) ? -1 : 0;
        T prod = (T)S[-dy * stride - dx] * (T)kk[k];
        T sum = acc + prod;
        acc = (valid >> k) & 1u ? sum : acc;
    }
    return acc;
}

template <typename KT>
__device__ __forceinline__ unsigned load_coeffs(const KT *__restrict__ kern, size_t plane, int nx, int y, int x, int y0,
                                                int y1, int x0, int x1, KT (&kk)[9]) {
    // destination (y,x) full-frame coordinates; active region rows [y0,y1) cols [x0,x1)
    unsigned valid = 0;
    const bool dest_ok = (y >= y0 && y < y1 && x >= x0 && x < x1);
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int dy = (k == 1 || k == 5 || k == 6) ? 1 : (k == 2 || k == 7 || k == 8) ? -1 : 0;
        const int dx = (k == 3 || k == 5 || k == 7) ? 1 : (k == 4 || k == 6 || k == 8) ? -1 : 0;
        const int sy = y - dy, sx = x - dx;
        const bool ok = dest_ok && sy >= y0 && sy < y1 && sx >= x0 && sx < x1;
        kk[k] = ok ? kern[(size_t)(3 * (1 + dy) + (1 + dx)) * plane + (size_t)sy * nx + sx] : (KT)0;
        valid |= ok ? (1u << k) : 0u;
    }
    return valid;
}

template <typename KT, typename GT, bool HAS_GAIN>
__global__ __launch_bounds__(IPC_THREADS) void ipc_cube_kernel(IpcArgs a) {
    // dtype walk of ipc_linearity.py:186,134-142: x = data*g in XT = promote(f32, gain); the first
    // "output + image2" is still in XT; everything that touches the kernel is in T = promote(XT, kernel)
    using XT = typename Promote<float, GT>::type;
    using T = typename Promote<XT, KT>::type;
    __shared__ XT X[IPC_XH * IPC_XW];  // gain * data on tile + halo 2
    __shared__ T O1[IPC_OH * IPC_OW];  // first Neumann iterate on tile + halo 1
    __shared__ GT GL[IPC_XH * IPC_XW];

    const int tid = threadIdx.x;
    const int tx = tid % IPC_TW, tyb = tid / IPC_TW;
    const int X0 = blockIdx.x * IPC_TW, Y0 = blockIdx.y * IPC_TH;  // tile origin, full-frame coordinates
    const int ny = a.ny, nx = a.nx, nb = a.nb;
    const int ay0 = nb, ay1 = ny - nb, ax0 = nb, ax1 = nx - nb;
    const size_t plane = (size_t)ny * nx;
    const KT *__restrict__ kern = reinterpret_cast<const KT *>(a.kern);
    const GT *__restrict__ gain = reinterpret_cast<const GT *>(a.gain);

    // coefficients of the four tile pixels this thread owns (rows tyb, tyb+4, tyb+8, tyb+12)
    KT kt[4][9];
    unsigned vt[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
        vt[q] = load_coeffs<KT>(kern, plane, nx, Y0 + tyb + 4 * q, X0 + tx, ay0, ay1, ax0, ax1, kt[q]);
    // one position of the halo-1 ring for the first IPC_RING threads
    KT kr[9];
    unsigned vr = 0;
    int rly = 0, rlx = 0;  // ring position in O1 coordinates
    if (tid < IPC_RING) {
        if (tid < IPC_OW) {
            rly = 0;
            rlx = tid;
        } else if (tid < 2 * IPC_OW) {
            rly = IPC_OH - 1;
            rlx = tid - IPC_OW;
        } else if (tid < 2 * IPC_OW + IPC_TH) {
            rly = 1 + (tid - 2 * IPC_OW);
            rlx = 0;
        } else {
            rly = 1 + (tid - 2 * IPC_OW - IPC_TH);
            rlx = IPC_OW - 1;
        }
        vr = load_coeffs<KT>(kern, plane, nx, Y0 - 1 + rly, X0 - 1 + rlx, ay0, ay1, ax0, ax1, kr);
    }
    // gain on tile + halo 2 (1 outside the active region: never used there)
    for (int idx = tid; idx < IPC_XH * IPC_XW; idx += IPC_THREADS) {
        const int y = Y0 - 2 + idx / IPC_XW, x = X0 - 2 + idx % IPC_XW;
        const bool ok = (y >= ay0 && y < ay1 && x >= ax0 && x < ax1);
        GL[idx] = (HAS_GAIN && ok) ? gain[(size_t)y * nx + x] : (GT)1;
    }
    __syncthreads();

    for (int g = 0; g < a.ngrp; ++g) {
        const float *__restrict__ in = a.in + (size_t)g * plane;
        float *__restrict__ out = a.out + (size_t)g * plane;
        for (int idx = tid; idx < IPC_XH * IPC_XW; idx += IPC_THREADS) {
            const int y = Y0 - 2 + idx / IPC_XW, x = X0 - 2 + idx % IPC_XW;
            const bool ok = (y >= ay0 && y < ay1 && x >= ax0 && x < ax1);
            XT v = (XT)0;
            if (ok) {
                v = (XT)in[(size_t)y * nx + x];
                if (HAS_GAIN) v = v * (XT)GL[idx];
            }
            X[idx] = v;
        }
        __syncthreads();
        // first iterate: out1 = (x + x) - fwd(x)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ly = tyb + 4 * q, lx = tx;  // tile coordinates
            const XT *c = &X[(ly + 2) * IPC_XW + (lx + 2)];
            T f = fwd_at<T, XT, KT>(c, IPC_XW, kt[q], vt[q]);
            O1[(ly + 1) * IPC_OW + (lx + 1)] = (T)(c[0] + c[0]) - f;
        }
        if (tid < IPC_RING) {
            const XT *c = &X[(rly + 1) * IPC_XW + (rlx + 1)];
            T f = fwd_at<T, XT, KT>(c, IPC_XW, kr, vr);
            O1[rly * IPC_OW + rlx] = (T)(c[0] + c[0]) - f;
        }
        __syncthreads();
        // second iterate: out2 = (out1 + x) - fwd(out1); result / gain
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ly = tyb + 4 * q, lx = tx;
            const int y = Y0 + ly, x = X0 + lx;
            if (y < ny && x < nx) {
                const bool act = (y >= ay0 && y < ay1 && x >= ax0 && x < ax1);
                float res;
                if (act) {
                    const T *c1 = &O1[(ly + 1) * IPC_OW + (lx + 1)];
                    const T xc = (T)X[(ly + 2) * IPC_XW + (lx + 2)];
                    T f = fwd_at<T, T, KT>(c1, IPC_OW, kt[q], vt[q]);
                    T o2 = (c1[0] + xc) - f;
                    if (HAS_GAIN) o2 = o2 / (T)GL[(ly + 2) * IPC_XW + (lx + 2)];
                    res = (float)o2;
                } else {
                    res = in[(size_t)y * nx + x];
                }
                out[(size_t)y * nx + x] = res;
            }
        }
        __syncthreads();
    }
}

template <typename KT, typename GT, bool HG>
static void launch_cube(rip_ctx *ctx, const IpcArgs &a) {
    dim3 grid((a.nx + IPC_TW - 1) / IPC_TW, (a.ny + IPC_TH - 1) / IPC_TH);
    hipLaunchKernelGGL((ipc_cube_kernel<KT, GT, HG>), grid, dim3(IPC_THREADS), 0, ctx->stream, a);
}

int rip_launch_ipc_cube(rip_ctx *ctx, const IpcArgs &a) {
    const bool k64 = a.k_dtype == RIP_F64, g64 = a.g_dtype == RIP_F64;
    if (!a.gain) {
        if (k64)
            launch_cube<double, float, false>(ctx, a);
        else
            launch_cube<float, float, false>(ctx, a);
    } else if (k64 && g64)
        launch_cube<double, double, true>(ctx, a);
    else if (k64)
        launch_cube<double, float, true>(ctx, a);
    else if (g64)
        launch_cube<float, double, true>(ctx, a);
    else
        launch_cube<float, float, true>(ctx, a);
    RIP_HIP(ctx, hipGetLastError());
    return RIP_OK;
}

// ------------------------------------------------------------------ generic image operators
// (function-level drop-ins for ipc_fwd / ipc_rev on an arbitrary image: any dtype mix, any order;
//  straightforward global-memory kernels, not on the throughput path)

template <typename XT, typename IT, typename GT>
__global__ void ipc_scale_kernel(const IT *img, const GT *gain, XT *out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = gain ? (XT)gain[i] * (XT)img[i] : (XT)img[i];
}

// dst = fwd(src) (mode 0) or dst = (src + x) - fwd(src) (mode 1); the sum is formed in the type of src
template <typename T, typename ST, typename XT, typename KT>
__global__ void ipc_fwd_image_kernel(const ST *src, const XT *x, const KT *kern, T *dst, int ny, int nx, int mode) {
    const int xx = blockIdx.x * blockDim.x + threadIdx.x, yy = blockIdx.y;
    if (xx >= nx || yy >= ny) return;
    const size_t plane = (size_t)ny * nx;
    const size_t c = (size_t)yy * nx + xx;
    T acc = (T)src[c] * (T)kern[4 * plane + c];
#pragma unroll
    for (int k = 1; k < 9; ++k) {
        const int dy = IPC_DY[k], dx = IPC_DX[k];
        const int sy = yy - dy, sx = xx - dx;
        if (sy >= 0 && sy < ny && sx >= 0 && sx < nx) {
            const size_t s = (size_t)sy * nx + sx;
            T prod = (T)src[s] * (T)kern[(size_t)(3 * (1 + dy) + (1 + dx)) * plane + s];
            acc = acc + prod;
        }
    }
    dst[c] = mode ? (T)(src[c] + (ST)x[c]) - acc : acc;
}

template <typename T, typename GT>
__global__ void ipc_unscale_kernel(T *buf, const GT *gain, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && gain) buf[i] = buf[i] / (T)gain[i];
}

template <typename T, typename XT, typename IT, typename KT, typename GT>
static int ipc_image_typed(rip_ctx *ctx, int reverse, int order, const void *img, int ny, int nx, const void *kern,
                           const void *gain, void *out) {
    // XT = dtype of gain*image (image alone without gain); T = promote(XT, kernel) = output dtype
    const size_t n = (size_t)ny * nx;
    XT *xbuf = (XT *)rip_ws(ctx, 5, n * sizeof(XT));
    T *abuf = (T *)rip_ws(ctx, 6, n * sizeof(T));
    T *bbuf = (T *)out;
    if (!xbuf || !abuf) return RIP_ENOMEM;
    const unsigned nb1 = (unsigned)((n + 255) / 256);
    dim3 g2((nx + 255) / 256, ny);
    hipLaunchKernelGGL((ipc_scale_kernel<XT, IT, GT>), dim3(nb1), dim3(256), 0, ctx->stream, (const IT *)img,
                       (const GT *)gain, xbuf, n);
    if (!reverse) {
        hipLaunchKernelGGL((ipc_fwd_image_kernel<T, XT, XT, KT>), g2, dim3(256), 0, ctx->stream, (const XT *)xbuf,
                           (const XT *)xbuf, (const KT *)kern, bbuf, ny, nx, 0);
    } else {
        // out_0 = x ; out_{n+1} = (out_n + x) - fwd(out_n); out_0 + x is still in the dtype of x
        if (order == 0) return rip_fail(ctx, RIP_EINVAL, "ipc_rev: order must be >= 1");
        T *dst = (order % 2) ? bbuf : abuf;
        hipLaunchKernelGGL((ipc_fwd_image_kernel<T, XT, XT, KT>), g2, dim3(256), 0, ctx->stream, (const XT *)xbuf,
                           (const XT *)xbuf, (const KT *)kern, dst, ny, nx, 1);
        const T *cur = dst;
        dst = (dst == abuf) ? bbuf : abuf;
        for (int it = 1; it < order; ++it) {
            hipLaunchKernelGGL((ipc_fwd_image_kernel<T, T, XT, KT>), g2, dim3(256), 0, ctx->stream, cur,
                               (const XT *)xbuf, (const KT *)kern, dst, ny, nx, 1);
            cur = dst;
            dst = (dst == abuf) ? bbuf : abuf;
        }
    }
    hipLaunchKernelGGL((ipc_unscale_kernel<T, GT>), dim3(nb1), dim3(256), 0, ctx->stream, bbuf, (const GT *)gain, n);
    RIP_HIP(ctx, hipGetLastError());
    return RIP_OK;
}

int rip_launch_ipc_image(rip_ctx *ctx, int reverse, int order, const void *img, int img_dtype, int ny, int nx,
                         const void *kern, int k_dtype, const void *gain, int g_dtype, void *out, int) {
    const bool i64 = img_dtype == RIP_F64, k64 = k_dtype == RIP_F64, g64 = gain && g_dtype == RIP_F64;
#define RIP_IPC_CASE(TI, TK, TG)                                                                      \
    if (i64 == (sizeof(TI) == 8) && k64 == (sizeof(TK) == 8) && g64 == (sizeof(TG) == 8)) {           \
        using XT_ = typename Promote<TI, TG>::type;                                                   \
        using T_ = typename Promote<XT_, TK>::type;                                                   \
        return ipc_image_typed<T_, XT_, TI, TK, TG>(ctx, reverse, order, img, ny, nx, kern, gain, out); \
    }
    RIP_IPC_CASE(float, float, float)
    RIP_IPC_CASE(float, float, double)
    RIP_IPC_CASE(float, double, float)
    RIP_IPC_CASE(float, double, double)
    RIP_IPC_CASE(double, float, float)
    RIP_IPC_CASE(double, float, double)
    RIP_IPC_CASE(double, double, float)
    RIP_IPC_CASE(double, double, double)
#undef RIP_IPC_CASE
    return rip_fail(ctx, RIP_EINVAL, "ipc image: unsupported dtype combination");
}
