// 1/f noise frames: sim_to_isim.noise_1f_frame (sim_to_isim.py:265-303), the generator behind the
// correlated part of a read-noise layer (fill_in_refdata_and_1f :376-399: one common and 32 per-channel frames plus one for
// the reference output, per group).  One frame of rows x width samples:
//     L = 2*rows*width ;  z_k = (n_k + i n_{L+k}) * a_k ,  a_k = |k|^-1/2 for the signed frequency index k (a_0 = 0)
//     block = Re(FFT(z))[0 : L/2] / sqrt(2) ;  block -= mean(block) ;  reshaped (rows, width), cast to f32
// in f64 as the reference (complex128 FFT of 2^20 points for the 4096 x 128 frame).  Only the REAL part of the transform is kept,
// and Re(sum_j z_j e^{-i t_j}) pairs the terms j and L-j (cos t_{L-j} = cos t_j, sin t_{L-j} = -sin t_j):
//     Re Z_k = s_0 + s_{L/2} (-1)^k + sum_{0<j<L/2} Re(s_j e^{+i t_j}),   s_j = (Re z_j + Re z_{L-j}) - i (Im z_j - Im z_{L-j})
// which is a complex-to-real transform of length L on the L/2+1 folded coefficients (hipFFT Z2D, batched: half the arithmetic and
// half the traffic of the complex transform of round 2; the same products a_k n_k as the reference forms, added in another order).
// The 2L standard normal deviates per frame come from the caller or, normals == NULL, from the device (Philox + Box-Muller).  The
// FFT is not the reference's pocketfft: results agree to rounding (~1e-12 relative), not bit for bit.
#include <hipfft/hipfft.h>

#include <algorithm>

#include "pink_fft.h"
#include "rip_common.h"

namespace {

__device__ __forceinline__ void philox10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        c[1] = (uint32_t)p1;
        c[3] = (uint32_t)p0;
        c[0] = n0;
        c[2] = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

// a_k for k = 0 .. L-1, once per frame length (the f64 power is the most expensive operation of the fill, and the same for every
// frame): signed frequency index as the reference builds it -- linspace(0, 1 - 1/L, L), upper half minus one, times L
__global__ __launch_bounds__(256) void pink_amp_kernel(double *__restrict__ amp, size_t L) {
    const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= L) return;
    const double step = (1.0 - 1.0 / (double)L) / (double)(L - 1);
    double freq = (double)k * step;
    if (k >= L / 2) freq -= 1.0;
    amp[k] = (k == 0) ? 0.0 : pow(1.0e-99 + fabs(freq * (double)L), -0.5);
}

// z_j and z_{L-j} of frame f, j <= L/2: the deviate pairs (n_k, n_{L+k}) times the amplitude a_k, from the caller's array or from
// the device generator -- ONE Philox block per pair of terms (every user of z_j uses z_{L-j} too): words 0, 1 make z_j, words 2, 3
// z_{L-j}, each by Box-Muller on a 40-bit and a 24-bit uniform deviate with the logarithm and the circular functions in f32 (as
// every other device deviate of the library, rip_rng.h: the frames are f32, the deviates need no more -- the f64 versions and a
// block per term were two thirds of the generator's instructions, which run beside rip_synth_resultants' f64 arithmetic,
// profiles/r04_summary.md).  j = 0 and j = L/2: the second term is the first one's index (or none): not to be used.
__device__ __forceinline__ void box_muller_64(uint32_t w0, uint32_t w1, double &a, double &b) {
    const double u1 = ((double)(((uint64_t)w0 << 8) | (w1 >> 24)) + 0.5) * (1.0 / 1099511627776.0);   // 40 bits, (0, 1)
    const float u2 = ((float)(w1 & 0xFFFFFFu) + 0.5f) * (1.0f / 16777216.0f);                          // 24 bits, (0, 1)
    const float r = sqrtf(-2.0f * logf((float)u1));   // ((float)u1 keeps the small values: the tail is the 40-bit one, 7.4 sigma)
    float sn, cs;
    sincospif(2.0f * u2, &sn, &cs);
    a = (double)(r * cs);
    b = (double)(r * sn);
}

__device__ __forceinline__ void pink_pair(const double *__restrict__ normals, const double *__restrict__ amps, size_t L, int f, size_t j,
                                          uint64_t seed, uint32_t stream_id, double &re, double &im, double &re2, double &im2) {
    const size_t j2 = (j == 0) ? 0 : L - j;
    double a, b, a2, b2;
    if (normals) {
        a = normals[(size_t)f * 2 * L + j];
        b = normals[(size_t)f * 2 * L + L + j];
        a2 = normals[(size_t)f * 2 * L + j2];
        b2 = normals[(size_t)f * 2 * L + L + j2];
    } else {
        uint32_t c[4] = {(uint32_t)j, (uint32_t)(j >> 32) ^ (uint32_t)f, stream_id, 0x70696e6bu};
        philox10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
        box_muller_64(c[0], c[1], a, b);
        box_muller_64(c[2], c[3], a2, b2);
    }
    const double amp = amps[j], amp2 = amps[j2];
    re = a * amp;
    im = b * amp;
    re2 = a2 * amp2;
    im2 = b2 * amp2;
}

// folded coefficients S[f*(L/2+1) + j], j = 0 .. L/2, of the complex-to-real transform (see the head of the file): S_0 = Re z_0,
// S_{L/2} = Re z_{L/2}, S_j = ((Re z_j + Re z_{L-j}) - i (Im z_j - Im z_{L-j})) / 2 (the transform counts those terms twice)
__global__ __launch_bounds__(256) void pink_fill_kernel(const double *__restrict__ normals, const double *__restrict__ amps,
                                                        hipfftDoubleComplex *__restrict__ S, size_t L, int nframes, uint64_t seed,
                                                        uint32_t stream_id, int f_first, int f_block) {
    const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int f = blockIdx.y;
    const size_t half = L / 2;
    if (j > half) return;
    // generator coordinates of frame F = f_first + f of the call: (F mod f_block, stream_id + F - F mod f_block) -- what the frames
    // get when they are made in blocks of f_block, whatever the batch of the transform is; the caller's deviates are indexed by f
    const int fg = normals ? f : (f_first + f) % f_block;
    stream_id += (uint32_t)((f_first + f) - (f_first + f) % f_block);
    double re, im, re2, im2;
    pink_pair(normals, amps, L, fg, j, seed, stream_id, re, im, re2, im2);
    hipfftDoubleComplex v;
    if (j == 0 || j == half) {
        v.x = re;
        v.y = 0.0;
    } else {
        v.x = (re + re2) * 0.5;
        v.y = -(im - im2) * 0.5;
    }
    S[(size_t)f * (half + 1) + j] = v;
}

// The coefficients W_j of the hand-written transform (pink_fft.h), j = 0 .. N-1, N = L/2: thread j <= N/2 forms W_j and W_{N-j}
// from the four deviate pairs (j, L-j, N-j, N+j) they share -- the folded S_j, S_{N-j} of pink_fill_kernel, never stored:
//     E = S_j + conj S_{N-j},  D = S_j - conj S_{N-j},  w = e^{2 pi i j / L}:   W_j = E + i w D,   W_{N-j} = conj(E - i w D)
//     W_0 = (S_0 + S_N) + i (S_0 - S_N)  (both real),   W_{N/2} = 2 conj S_{N/2}
__global__ __launch_bounds__(256) void pink_fill_w_kernel(const double *__restrict__ normals, const double *__restrict__ amps,
                                                          double2 *__restrict__ W, size_t L, uint64_t seed, uint32_t stream_id,
                                                          int f_first, int f_block) {
    const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int f = blockIdx.y;
    const size_t N = L / 2;
    if (j > N / 2) return;
    const int fg = normals ? f : (f_first + f) % f_block;
    stream_id += (uint32_t)((f_first + f) - (f_first + f) % f_block);
    double2 *Wf = W + (size_t)f * N;
    double ra, ia, rb, ib;
    if (j == 0) {
        double r0, rn, unused[3];
        pink_pair(normals, amps, L, fg, 0, seed, stream_id, r0, ia, unused[0], unused[1]);
        pink_pair(normals, amps, L, fg, N, seed, stream_id, rn, ib, unused[0], unused[2]);
        Wf[0] = make_double2(r0 + rn, r0 - rn);
        return;
    }
    pink_pair(normals, amps, L, fg, j, seed, stream_id, ra, ia, rb, ib);
    const double2 sj = make_double2((ra + rb) * 0.5, -(ia - ib) * 0.5);
    const size_t m = N - j;
    if (m == j) {
        Wf[j] = make_double2(2.0 * sj.x, -2.0 * sj.y);
        return;
    }
    pink_pair(normals, amps, L, fg, m, seed, stream_id, ra, ia, rb, ib);
    const double2 sm = make_double2((ra + rb) * 0.5, -(ia - ib) * 0.5);
    const double2 E = make_double2(sj.x + sm.x, sj.y - sm.y), D = make_double2(sj.x - sm.x, sj.y + sm.y);
    double ws, wc;
    sincospi((double)j / (double)N, &ws, &wc);   // e^{i pi j / N}
    const double2 wd = make_double2(wc * D.x - ws * D.y, wc * D.y + ws * D.x), iwd = make_double2(-wd.y, wd.x);
    Wf[j] = make_double2(E.x + iwd.x, E.y + iwd.y);
    Wf[m] = make_double2(E.x - iwd.x, -(E.y - iwd.y));
}

// sum of x[0 : L/2] / sqrt(2) per frame (f64): 256 block partials per frame, added in a FIXED order by pink_out_kernel
// (an atomicAdd across blocks would make the subtracted mean, hence the frame, differ in the last bit from run to run)
__global__ __launch_bounds__(256) void pink_sum_kernel(const double *__restrict__ x, size_t L, double *__restrict__ sums) {
    __shared__ double sh[256];
    const int f = blockIdx.y;
    const size_t half = L / 2;
    double acc = 0.0;
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < half; k += (size_t)gridDim.x * 256) acc += x[(size_t)f * L + k] / sqrt(2.0);
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) sums[(size_t)f * gridDim.x + blockIdx.x] = sh[0];
}

// the frame's mean from the 256 block sums, added in index order (same bits in every run); written over the frame's first sum
__global__ __launch_bounds__(64) void pink_mean_kernel(double *__restrict__ sums, size_t half, int nframes) {
    const int f = blockIdx.x * 64 + threadIdx.x;
    if (f >= nframes) return;
    double tot = 0.0;
    for (int b = 0; b < 256; ++b) tot += sums[(size_t)f * 256 + b];
    sums[(size_t)f * 256] = tot / (double)half;
}

__global__ __launch_bounds__(256) void pink_out_kernel(const double *__restrict__ x, size_t L, const double *__restrict__ sums,
                                                       float *__restrict__ out) {
    const size_t k = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;   // two samples per thread
    const int f = blockIdx.y;
    const size_t half = L / 2;
    if (k >= half) return;
    const double mean = sums[(size_t)f * 256];
    const double *xf = x + (size_t)f * L;
    float *of = out + (size_t)f * half;
    if ((half & 1) == 0) {   // (an odd frame size: the pairs of the odd frames are not aligned)
        const double2 v = *reinterpret_cast<const double2 *>(xf + k);
        *reinterpret_cast<float2 *>(of + k) = make_float2((float)(v.x / sqrt(2.0) - mean), (float)(v.y / sqrt(2.0) - mean));
    } else {
        of[k] = (float)(xf[k] / sqrt(2.0) - mean);
        if (k + 1 < half) of[k + 1] = (float)(xf[k + 1] / sqrt(2.0) - mean);
    }
}

// out: host memory (frames copied back chunk by chunk, call synchronous) or, out_dev, device memory (asynchronous: the transform
// plan and buffers are kept with the context between calls of the same frame length and batch)
// `st`: the stream of this call (the context's main stream, or its second one for frames made ahead).  Every call shares the
// transform plan and the pink_z / pink_s buffers: a call that runs on another stream than the call before waits for it first
// (ev_pink, recorded at the end of every call on the stream that ran it).
int noise_1f_impl(rip_ctx *ctx, hipStream_t st, int rows, int width, int nframes, const double *normals, uint64_t seed,
                  uint32_t stream_id, float *out, bool out_dev) {
    if (rows < 1 || width < 1 || nframes < 1 || !out) return rip_fail(ctx, RIP_EINVAL, "noise_1f: bad arguments");
    RIP_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->ev_pink_valid && ctx->pink_stream != st) RIP_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_pink, 0));
    const size_t L = (size_t)2 * rows * width, half = L / 2;
    double *d_n = nullptr;
    float *d_o = nullptr;
    int rc = RIP_OK;
    auto done = [&]() {   // per-call buffers; the plan and the transform buffers stay with the context
        for (void *p : {(void *)d_n, (void *)d_o})
            if (p) (void)hipFree(p);
    };
#define PK_HIP(call)                                                                   \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) {                                                        \
            rc = rip_fail(ctx, RIP_EHIP, "%s: %s", #call, hipGetErrorString(e_));      \
            done();                                                                    \
            return rc;                                                                 \
        }                                                                              \
    } while (0)
    // frames are transformed in chunks so that the complex buffer stays at about 1 GB: equal chunks (272 frames of 2^20 points:
    // 5 x 55, not 4 x 64 + 16 padded to 64 -- the transform runs its full batch every time).  The device generator's streams are
    // laid out in blocks of `fblock` frames whatever the chunk is (pink_fill_kernel)
    const int fblock = (int)std::max<size_t>(1, ((size_t)1 << 30) / (L * sizeof(hipfftDoubleComplex)));
    const int nchunks = (nframes + fblock - 1) / fblock;
    const int chunk = (nframes + nchunks - 1) / nchunks;
    // power-of-two frames: the hand-written two-pass transform (pink_fft.h); otherwise (and with option "pink_form" = 0) the library's
    const bool own = ctx->pink_form != 0 && pf::supported(L);
    if (ctx->pink_L != L || ctx->pink_chunk != chunk || ctx->pink_own != own) {   // another frame length or batch: new plan and buffers
        PK_HIP(hipStreamSynchronize(ctx->stream));       // (either stream may still be using the old ones)
        if (ctx->stream2) PK_HIP(hipStreamSynchronize(ctx->stream2));
        rip_pink_release(ctx);
        // one buffer: the folded coefficients (chunk x (L/2+1) complex), then the real series (chunk x L), then the block sums
        PK_HIP(hipMalloc(&ctx->pink_z, (size_t)chunk * (half + 1) * sizeof(hipfftDoubleComplex) + (size_t)chunk * L * sizeof(double)));
        PK_HIP(hipMalloc(&ctx->pink_s, ((size_t)chunk * 256 + L) * sizeof(double)));   // the block sums, then the amplitudes a_k
        hipLaunchKernelGGL(pink_amp_kernel, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, st, (double *)ctx->pink_s + (size_t)chunk * 256, L);
        PK_HIP(hipStreamSynchronize(st));   // (once per frame length: later calls may come on the context's other stream)
        if (own) {
            pf::Tables *t = new pf::Tables();
            ctx->pink_tab = t;
            const hipError_t e = pf::make_tables(L, *t);
            if (e != hipSuccess) {
                rc = rip_fail(ctx, RIP_EHIP, "noise_1f: transform tables for %zu points: %s", L, hipGetErrorString(e));
                rip_pink_release(ctx);
                done();
                return rc;
            }
        } else {
            int n1 = (int)L;
            hipfftHandle made = 0;
            if (hipfftPlanMany(&made, 1, &n1, nullptr, 1, (int)(half + 1), nullptr, 1, n1, HIPFFT_Z2D, chunk) != HIPFFT_SUCCESS) {
                rc = rip_fail(ctx, RIP_EHIP, "noise_1f: hipfftPlanMany(%zu points x %d) failed", L, chunk);
                rip_pink_release(ctx);
                done();
                return rc;
            }
            ctx->pink_plan = (void *)made;
        }
        ctx->pink_L = L;
        ctx->pink_chunk = chunk;
        ctx->pink_own = own;
    }
    hipfftDoubleComplex *z = (hipfftDoubleComplex *)ctx->pink_z;
    double *x = reinterpret_cast<double *>(z + (size_t)chunk * (half + 1));
    double *d_s = (double *)ctx->pink_s;
    const double *d_amp = d_s + (size_t)chunk * 256;
    hipfftHandle plan = (hipfftHandle)ctx->pink_plan;
    if (!out_dev) PK_HIP(hipMalloc((void **)&d_o, (size_t)chunk * half * sizeof(float)));
    if (normals) PK_HIP(hipMalloc((void **)&d_n, (size_t)chunk * 2 * L * sizeof(double)));
    if (!own && hipfftSetStream(plan, st) != HIPFFT_SUCCESS) {
        rc = rip_fail(ctx, RIP_EHIP, "noise_1f: hipfftSetStream failed");
        done();
        return rc;
    }
    for (int f0 = 0; f0 < nframes; f0 += chunk) {
        const int nf = std::min(chunk, nframes - f0);
        if (normals)
            PK_HIP(hipMemcpyAsync(d_n, normals + (size_t)f0 * 2 * L, (size_t)nf * 2 * L * sizeof(double), hipMemcpyHostToDevice, st));
        if (own) {
            const pf::Tables &t = *(const pf::Tables *)ctx->pink_tab;
            double2 *w = reinterpret_cast<double2 *>(z);
            hipLaunchKernelGGL(pink_fill_w_kernel, dim3((unsigned)((half / 2 + 1 + 255) / 256), nf), dim3(256), 0, st, (const double *)d_n, d_amp,
                               w, L, seed, stream_id, f0, fblock);
            hipLaunchKernelGGL(pf::pf_cols_kernel, dim3(t.dev.n2 / pf::TW, nf), dim3(pf::NT1), t.lds1, st, w, t.dev);
            hipLaunchKernelGGL(pf::pf_rows_kernel, dim3(t.dev.n1 / pf::TW, nf), dim3(pf::NT2), t.lds2, st, (const double2 *)w, x, t.dev);
        } else {
            hipLaunchKernelGGL(pink_fill_kernel, dim3((unsigned)((half + 1 + 255) / 256), nf), dim3(256), 0, st, (const double *)d_n, d_amp, z, L,
                               nf, seed, stream_id, f0, fblock);
            if (nf < chunk)
                PK_HIP(hipMemsetAsync(z + (size_t)nf * (half + 1), 0, (size_t)(chunk - nf) * (half + 1) * sizeof(hipfftDoubleComplex), st));
            if (hipfftExecZ2D(plan, z, x) != HIPFFT_SUCCESS) {
                rc = rip_fail(ctx, RIP_EHIP, "noise_1f: hipfftExecZ2D failed");
                done();
                return rc;
            }
        }
        hipLaunchKernelGGL(pink_sum_kernel, dim3(256, nf), dim3(256), 0, st, (const double *)x, L, d_s);
        float *dst = out_dev ? out + (size_t)f0 * half : d_o;
        hipLaunchKernelGGL(pink_mean_kernel, dim3((unsigned)((nf + 63) / 64)), dim3(64), 0, st, d_s, half, nf);
        hipLaunchKernelGGL(pink_out_kernel, dim3((unsigned)(((half + 1) / 2 + 255) / 256), nf), dim3(256), 0, st, (const double *)x, L,
                           (const double *)d_s, dst);
        PK_HIP(hipGetLastError());
        if (!out_dev) PK_HIP(hipMemcpyAsync(out + (size_t)f0 * half, d_o, (size_t)nf * half * sizeof(float), hipMemcpyDeviceToHost, st));
        if (!out_dev) PK_HIP(hipStreamSynchronize(st));   // device output: the next chunk follows in stream order
    }
    if (!ctx->ev_pink) PK_HIP(hipEventCreateWithFlags(&ctx->ev_pink, hipEventDisableTiming));
    PK_HIP(hipEventRecord(ctx->ev_pink, st));
    ctx->pink_stream = st;
    ctx->ev_pink_valid = true;
#undef PK_HIP
    done();
    return RIP_OK;
}

}   // namespace

void rip_pink_release(rip_ctx *ctx) {
    if (ctx->pink_plan) (void)hipfftDestroy((hipfftHandle)ctx->pink_plan);
    if (ctx->pink_tab) {
        pf::Tables *t = (pf::Tables *)ctx->pink_tab;
        if (t->mem) (void)hipFree(t->mem);
        delete t;
        ctx->pink_tab = nullptr;
    }
    if (ctx->pink_z) (void)hipFree(ctx->pink_z);
    if (ctx->pink_s) (void)hipFree(ctx->pink_s);
    ctx->pink_plan = ctx->pink_z = ctx->pink_s = nullptr;
    ctx->pink_L = 0;
    ctx->pink_chunk = 0;
}

extern "C" int rip_stage_noise_1f(rip_ctx *ctx, int rows, int width, int nframes, const double *normals, uint64_t seed,
                                  uint32_t stream_id, float *out) {
    if (ctx->frames_pending && ctx->ev_frames) (void)hipStreamWaitEvent(ctx->stream, ctx->ev_frames, 0);
    return noise_1f_impl(ctx, ctx->stream, rows, width, nframes, normals, seed, stream_id, out, false);
}

// frames made ahead on the second stream share the transform buffers with every other 1/f call: those wait for them first
static int frames_drain(rip_ctx *ctx) {
    if (ctx->frames_pending && ctx->ev_frames) RIP_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_frames, 0));
    return RIP_OK;
}

extern "C" int rip_synth_noise_1f(rip_ctx *ctx, int rows, int width, int nframes, uint64_t seed, uint32_t stream_id, float *out) {
    ctx->stream_dirty = true;
    int rc = frames_drain(ctx);
    if (rc) return rc;
    return noise_1f_impl(ctx, ctx->stream, rows, width, nframes, nullptr, seed, stream_id, out, true);
}

extern "C" int rip_synth_frames_ahead(rip_ctx *ctx, int rows, int width, int nframes, uint64_t seed) {
    if (!ctx->stream2) return RIP_OK;   // no second stream: rip_synth_fill makes the frames itself, in stream order
    if (rows < 1 || width < 1 || nframes < 1) return rip_fail(ctx, RIP_EINVAL, "synth_frames_ahead: bad geometry");
    RIP_HIP(ctx, hipSetDevice(ctx->device));
    float *made = (float *)rip_ws(ctx, 12, (size_t)nframes * rows * width * sizeof(float));
    if (!made) return RIP_ENOMEM;
    if (!ctx->ev_frames) RIP_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_frames, hipEventDisableTiming));
    // behind what the main stream holds at the time of the call: the previous exposure's fill kernels (they read the frames this
    // call overwrites), and whatever the caller wants out of the way first -- the transforms take HBM bandwidth from a
    // bandwidth-bound neighbour (rip_synth_apportion beside rocFFT's transposes: 2.7 -> 9.3 ms), not from rip_synth_resultants
    if (!ctx->ev_ahead) RIP_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_ahead, hipEventDisableTiming));
    RIP_HIP(ctx, hipEventRecord(ctx->ev_ahead, ctx->stream));
    RIP_HIP(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_ahead, 0));
    const int rc = noise_1f_impl(ctx, ctx->stream2, rows, width, nframes, nullptr, seed, 0x31660000u, made, true);
    if (rc) return rc;
    RIP_HIP(ctx, hipEventRecord(ctx->ev_frames, ctx->stream2));
    ctx->frames_pending = true;
    ctx->frames_seed = seed;
    ctx->frames_geom[0] = rows, ctx->frames_geom[1] = width, ctx->frames_geom[2] = nframes;
    return RIP_OK;
}
