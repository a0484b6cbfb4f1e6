// Micro-benchmark: how fast can the byte mix of the fused chain be streamed at all?  Reads, per pixel of a 4096 x 4096 frame,
// the arrays the fused kernel reads (8 groups x {u16 cube, u8 groupdq, f32 dark, f32 bias}, 15 + 9 + 5 f32/u32 planes) and
// writes what it writes (4 planes + 8 group-flag bytes), with a trivial reduction in between.  Two access shapes:
//   W = 1: one pixel per lane (the fused kernel's shape: 64-, 128- and 256-byte wave accesses)
//   W = 4: four consecutive pixels per lane (dword / dwordx2 / dwordx4 accesses)
// and R = rows marched per thread with all of a row's loads issued before the sums (the fused kernel's row prefetch).
//   hipcc --offload-arch=gfx950 -O3 -o stream_mix stream_mix.hip && ./stream_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int N = 4096, G = 8, NPL = 29;  // 15 slab planes + 9 ipc planes + read, dark rate, flat, flat flags, pdq
struct Args {
    const uint16_t *cube; const uint8_t *gdq; const float *dark, *bias, *planes;
    float *o0, *o1, *o2, *o3; uint8_t *gout;
};

// L = 0: every array planar (group / plane, row, column), as the caller's arrays and today's CALDIR copies are laid out;
// L = 1: the CALDIR arrays (dark, bias, the 29 planes) row-interleaved (row, group / plane, column): a row step of a strip then
//        touches ~45 pieces within one 0.7 MB region instead of 45 regions 67 MB apart (address-translation reach, DRAM pages)
template <int W, int L>
__global__ __launch_bounds__(256) void mix_kernel(Args a, int rows_per) {
    const size_t npix = (size_t)N * N;
    const int col = (blockIdx.x % (N / (256 * W))) * 256 * W + threadIdx.x * W;
    const int r0 = (blockIdx.x / (N / (256 * W))) * rows_per;
    for (int r = r0; r < r0 + rows_per && r < N; ++r) {
        const size_t p = (size_t)r * N + col;
        float acc[W];
        unsigned q[W];
#pragma unroll
        for (int w = 0; w < W; ++w) acc[w] = 0.f, q[w] = 0;
        if constexpr (W == 1) {
            unsigned s[G], qq[G]; float dk[G], bs[G], pl[NPL];
            // L = 3: the CALDIR words TILED (row, strip of 256 columns, word, column): one contiguous 45 KB piece per row step of a strip
            const size_t tb = (((size_t)r * (N / 256) + col / 256) * 45) * 256 + (col & 255);
#pragma unroll
            for (int g = 0; g < G; ++g) { s[g] = a.cube[g * npix + p]; qq[g] = a.gdq[g * npix + p]; dk[g] = (L == 3) ? a.planes[tb + g * 256] : a.dark[L ? ((size_t)r * G + g) * N + col : g * npix + p]; bs[g] = (L == 3) ? a.planes[tb + (8 + g) * 256] : a.bias[L ? ((size_t)r * G + g) * N + col : g * npix + p]; }
#pragma unroll
            for (int i = 0; i < NPL; ++i) pl[i] = (L == 3) ? a.planes[tb + (16 + i) * 256] : a.planes[L ? ((size_t)r * NPL + i) * N + col : i * npix + p];
#pragma unroll
            for (int g = 0; g < G; ++g) { acc[0] += (float)s[g] - dk[g] + bs[g]; q[0] |= qq[g] << (g & 3); }
#pragma unroll
            for (int i = 0; i < NPL; ++i) acc[0] += pl[i];
        } else {
            uint2 s[G]; unsigned qq[G]; float4 dk[G], bs[G], pl[NPL];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                s[g] = *reinterpret_cast<const uint2 *>(a.cube + g * npix + p);
                qq[g] = *reinterpret_cast<const unsigned *>(a.gdq + g * npix + p);
                dk[g] = *reinterpret_cast<const float4 *>(a.dark + (L ? ((size_t)r * G + g) * N + col : g * npix + p));
                bs[g] = *reinterpret_cast<const float4 *>(a.bias + (L ? ((size_t)r * G + g) * N + col : g * npix + p));
            }
#pragma unroll
            for (int i = 0; i < NPL; ++i) pl[i] = *reinterpret_cast<const float4 *>(a.planes + (L ? ((size_t)r * NPL + i) * N + col : i * npix + p));
#pragma unroll
            for (int g = 0; g < G; ++g) {
                acc[0] += (float)(s[g].x & 0xffff) - dk[g].x + bs[g].x; acc[1] += (float)(s[g].x >> 16) - dk[g].y + bs[g].y;
                acc[2] += (float)(s[g].y & 0xffff) - dk[g].z + bs[g].z; acc[3] += (float)(s[g].y >> 16) - dk[g].w + bs[g].w;
                q[0] |= (qq[g] & 0xff) << (g & 3); q[1] |= ((qq[g] >> 8) & 0xff) << (g & 3);
                q[2] |= ((qq[g] >> 16) & 0xff) << (g & 3); q[3] |= (qq[g] >> 24) << (g & 3);
            }
#pragma unroll
            for (int i = 0; i < NPL; ++i) { acc[0] += pl[i].x; acc[1] += pl[i].y; acc[2] += pl[i].z; acc[3] += pl[i].w; }
        }
        if constexpr (W == 1) {
            a.o0[p] = acc[0]; a.o1[p] = acc[0] * 2.f; a.o2[p] = acc[0] * 3.f; a.o3[p] = __uint_as_float(q[0]);
#pragma unroll
            for (int g = 0; g < G; ++g) a.gout[g * npix + p] = (uint8_t)(q[0] >> g);
        } else {
            *reinterpret_cast<float4 *>(a.o0 + p) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            *reinterpret_cast<float4 *>(a.o1 + p) = make_float4(acc[0] * 2.f, acc[1] * 2.f, acc[2] * 2.f, acc[3] * 2.f);
            *reinterpret_cast<float4 *>(a.o2 + p) = make_float4(acc[0] * 3.f, acc[1] * 3.f, acc[2] * 3.f, acc[3] * 3.f);
            *reinterpret_cast<float4 *>(a.o3 + p) = make_float4(__uint_as_float(q[0]), __uint_as_float(q[1]), __uint_as_float(q[2]), __uint_as_float(q[3]));
#pragma unroll
            for (int g = 0; g < G; ++g)
                *reinterpret_cast<unsigned *>(a.gout + g * npix + p) = ((q[0] >> g) & 0xff) | (((q[1] >> g) & 0xff) << 8) | (((q[2] >> g) & 0xff) << 16) | ((q[3] >> g) << 24);
        }
    }
}

// L = 2: the CALDIR arrays as PLANES OF FLOAT4 (four scalar planes interleaved per pixel): 45 dword streams become 11 streams of
// 16 bytes per lane -- one contiguous KiB per wave instruction, the float4-copy access shape; the caller's cube / groupdq and
// the outputs stay planar.  One pixel per lane.
__global__ __launch_bounds__(256) void quad_kernel(Args a, int rows_per) {
    const size_t npix = (size_t)N * N;
    constexpr int NQ = 11;   // 16 dark / bias + 29 planes = 45 words -> 11 float4 planes (44 words: main() allocates exactly that)
    const int col = (blockIdx.x % (N / 256)) * 256 + threadIdx.x;
    const int r0 = (blockIdx.x / (N / 256)) * rows_per;
    const float4 *qp = reinterpret_cast<const float4 *>(a.planes);
    for (int r = r0; r < r0 + rows_per && r < N; ++r) {
        const size_t p = (size_t)r * N + col;
        unsigned s[G], qq[G];
        float4 q4[NQ];
#pragma unroll
        for (int g = 0; g < G; ++g) { s[g] = a.cube[g * npix + p]; qq[g] = a.gdq[g * npix + p]; }
#pragma unroll
        for (int i = 0; i < NQ; ++i) q4[i] = qp[i * npix + p];
        float acc = 0.f;
        unsigned q = 0;
#pragma unroll
        for (int g = 0; g < G; ++g) { acc += (float)s[g]; q |= qq[g] << (g & 3); }
#pragma unroll
        for (int i = 0; i < NQ; ++i) acc += q4[i].x + q4[i].y + q4[i].z + q4[i].w;
        a.o0[p] = acc; a.o1[p] = acc * 2.f; a.o2[p] = acc * 3.f; a.o3[p] = __uint_as_float(q);
#pragma unroll
        for (int g = 0; g < G; ++g) a.gout[g * npix + p] = (uint8_t)(q >> g);
    }
}

int main() {
    const size_t npix = (size_t)N * N;
    Args a;
    void *p;
    auto mk = [&](size_t bytes) { CK(hipMalloc(&p, bytes)); CK(hipMemset(p, 1, bytes)); return p; };
    a.cube = (const uint16_t *)mk(G * npix * 2); a.gdq = (const uint8_t *)mk(G * npix); a.dark = (const float *)mk(G * npix * 4);
    a.bias = (const float *)mk(G * npix * 4);
    a.planes = (const float *)mk((size_t)45 * npix * 4);   // 29 scalar planes, or the 11 float4 planes of quad_kernel (44 words per pixel)
    a.o0 = (float *)mk(npix * 4); a.o1 = (float *)mk(npix * 4); a.o2 = (float *)mk(npix * 4); a.o3 = (float *)mk(npix * 4);
    a.gout = (uint8_t *)mk(G * npix);
    const double bytes = (double)npix * (G * (2 + 1 + 4 + 4) + NPL * 4 + 16 + G);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rows_per : {8, 128}) {
        for (int WL : {10, 11, 13, 40, 41}) {
            const int W = WL / 10, L = WL % 10;
            if (W == 4 && rows_per == 128) continue;   // 128 blocks: not enough parallelism, measured once (2.3 ms)
            const int strips = N / (256 * W), ranges = (N + rows_per - 1) / rows_per;
            auto launch = [&]() {
                if (WL == 10) hipLaunchKernelGGL((mix_kernel<1, 0>), dim3(strips * ranges), dim3(256), 0, 0, a, rows_per);
                if (WL == 11) hipLaunchKernelGGL((mix_kernel<1, 1>), dim3(strips * ranges), dim3(256), 0, 0, a, rows_per);
                if (WL == 13) hipLaunchKernelGGL((mix_kernel<1, 3>), dim3(strips * ranges), dim3(256), 0, 0, a, rows_per);
                if (WL == 40) hipLaunchKernelGGL((mix_kernel<4, 0>), dim3(strips * ranges), dim3(256), 0, 0, a, rows_per);
                if (WL == 41) hipLaunchKernelGGL((mix_kernel<4, 1>), dim3(strips * ranges), dim3(256), 0, 0, a, rows_per);
            };
            for (int i = 0; i < 300; ++i) launch();   // warm-up incl. the clock ramp
            CK(hipEventRecord(e0, 0));
            const int reps = 200;
            for (int i = 0; i < reps; ++i) launch();
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("W=%d L=%d rows_per=%3d blocks=%6d : %.4f ms per pass, %.2f TB/s (%.0f B/pixel)\n", W, L, rows_per, strips * ranges, ms / reps,
                   bytes / (ms / reps * 1e-3) / 1e12, bytes / npix);
        }
    }
    for (int rows_per : {8, 128}) {
        const int strips = N / 256, ranges = (N + rows_per - 1) / rows_per;
        auto launch = [&]() { hipLaunchKernelGGL(quad_kernel, dim3(strips * ranges), dim3(256), 0, 0, a, rows_per); };
        for (int i = 0; i < 300; ++i) launch();
        CK(hipEventRecord(e0, 0));
        const int reps = 200;
        for (int i = 0; i < reps; ++i) launch();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double qbytes = (double)npix * (G * 3 + 11 * 16 + 16 + G);
        printf("quad planes rows_per=%3d blocks=%6d : %.4f ms per pass, %.2f TB/s (%.0f B/pixel) -> %.4f ms at 228 B/pixel\n", rows_per,
               strips * ranges, ms / reps, qbytes / (ms / reps * 1e-3) / 1e12, qbytes / npix, ms / reps * 228.0 / (qbytes / npix));
    }
    return 0;
}
