// Micro-benchmark: what separates the streaming rate of the fused chain's byte mix (0.70 - 0.80 ms for 3.8 GB, stream_mix.hip) from
// the float4-copy rate of the same chip (6.3 TB/s, MI355X_MICROARCH.md)?  One variable at a time:
//   copy      float4 copy, half reads half writes (the guide's figure)
//   read      float4 reads only (+ 1/16 of the bytes written): the chain is 89 % reads
//   streams   the same bytes read as S separate planes (S = 1 .. 60), one dword per lane and plane, 1/8 written
//   width     one plane set read as bytes / shorts / dwords / dwordx4 per lane (same total bytes)
//   mix       the chain's mix: 8 x {u16, u8, f32, f32} + 29 f32 planes in, 4 f32 + 8 u8 out, one pixel per lane -- and with the u16 / u8
//             streams read as one DWORD per lane by the first 32 / 16 lanes of a wave (the other lanes get their piece by a lane
//             read: here v_readlane-free ds_bpermute), i.e. the same bytes in fewer, wider requests
// Every kernel: 256 threads, 8 rows of 256 columns per block unless noted; all loads of a row issued before its sums.
//   hipcc --offload-arch=gfx950 -O3 -o stream_bound stream_bound.hip && ./stream_bound
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int N = 4096;
constexpr size_t NPIX = (size_t)N * N;

__global__ __launch_bounds__(256) void copy4_kernel(const float4 *__restrict__ in, float4 *__restrict__ out, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) out[i] = in[i];
}
__global__ __launch_bounds__(256) void read4_kernel(const float4 *__restrict__ in, float *__restrict__ out, size_t n4) {
    // 16 x float4 per thread in flight, one float written per 4 float4 read
    const size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
    if (base + 16 > n4) return;
    float4 v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = in[(size_t)blockIdx.x * 256 * 16 + k * 256 + threadIdx.x];
    float s[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 16; ++k) s[k & 3] += v[k].x + v[k].y + v[k].z + v[k].w;
#pragma unroll
    for (int k = 0; k < 4; ++k) out[(size_t)blockIdx.x * 1024 + k * 256 + threadIdx.x] = s[k];
}

// S planes of f32, one pixel per lane, rows_per rows per block; writes one plane per 8 read
template <int S>
__global__ __launch_bounds__(256) void streams_kernel(const float *__restrict__ in, float *__restrict__ out, int rows_per) {
    const int col = (blockIdx.x % (N / 256)) * 256 + threadIdx.x;
    const int r0 = (blockIdx.x / (N / 256)) * rows_per;
    for (int r = r0; r < r0 + rows_per; ++r) {
        const size_t p = (size_t)r * N + col;
        float v[S];
#pragma unroll
        for (int i = 0; i < S; ++i) v[i] = in[(size_t)i * NPIX + p];
        constexpr int NO = (S + 7) / 8;
        float s[NO];
#pragma unroll
        for (int i = 0; i < NO; ++i) s[i] = 0.f;
#pragma unroll
        for (int i = 0; i < S; ++i) s[i % NO] += v[i];
#pragma unroll
        for (int i = 0; i < NO; ++i) out[(size_t)i * NPIX + p] = s[i];
    }
}

// 32 planes' worth of bytes (32 * 4 B per pixel) read with T-sized elements: planes of T, 128 / sizeof(T) planes; one element per lane
template <typename T>
__global__ __launch_bounds__(256) void width_kernel(const T *__restrict__ in, float *__restrict__ out, int rows_per) {
    constexpr int S = 64 / sizeof(T) > 64 ? 64 : 64 / sizeof(T);   // 64 B per pixel: 64 byte planes, 32 short planes, 16 dword planes
    const int col = (blockIdx.x % (N / 256)) * 256 + threadIdx.x;
    const int r0 = (blockIdx.x / (N / 256)) * rows_per;
    for (int r = r0; r < r0 + rows_per; ++r) {
        const size_t p = (size_t)r * N + col;
        T v[S];
#pragma unroll
        for (int i = 0; i < S; ++i) v[i] = in[(size_t)i * NPIX + p];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < S; ++i) s += (float)v[i];
        out[p] = s;
    }
}
__global__ __launch_bounds__(256) void width16_kernel(const float4 *__restrict__ in, float *__restrict__ out, int rows_per) {
    // 4 float4 planes = 64 B per pixel
    const int col = (blockIdx.x % (N / 256)) * 256 + threadIdx.x;
    const int r0 = (blockIdx.x / (N / 256)) * rows_per;
    for (int r = r0; r < r0 + rows_per; ++r) {
        const size_t p = (size_t)r * N + col;
        float4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = in[(size_t)i * NPIX + p];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) s += v[i].x + v[i].y + v[i].z + v[i].w;
        out[p] = s;
    }
}

struct MixArgs {
    const uint16_t *cube; const uint8_t *gdq; const float *dark, *bias, *planes;
    float *o0, *o1, *o2, *o3; uint8_t *gout;
};
constexpr int G = 8, NPL = 29;
// WIDE = 0: one pixel per lane for every array (the fused kernel's loads).  WIDE = 1: the u16 cube and the u8 groupdq of a wave's 64
// pixels are fetched as dwords by its first 32 / 16 lanes and handed round with ds_bpermute (same bytes, a half / a quarter of the
// lanes per request); the byte stores likewise gathered into dwords by 16 lanes.
template <int WIDE>
__global__ __launch_bounds__(256) void mix_kernel(MixArgs a, int rows_per) {
    const int lane = threadIdx.x & 63;
    const int col = (blockIdx.x % (N / 256)) * 256 + threadIdx.x;
    const int wcol = col - lane;   // first column of the wave
    const int r0 = (blockIdx.x / (N / 256)) * rows_per;
    for (int r = r0; r < r0 + rows_per; ++r) {
        const size_t p = (size_t)r * N + col, pw = (size_t)r * N + wcol;
        unsigned s[G], qq[G];
        float dk[G], bs[G], pl[NPL];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (WIDE) {
                const unsigned ws = (lane < 32) ? reinterpret_cast<const unsigned *>(a.cube + g * NPIX + pw)[lane] : 0u;
                const unsigned wq = (lane < 16) ? reinterpret_cast<const unsigned *>(a.gdq + g * NPIX + pw)[lane] : 0u;
                const unsigned ts = (unsigned)__builtin_amdgcn_ds_bpermute((lane >> 1) << 2, (int)ws);
                const unsigned tq = (unsigned)__builtin_amdgcn_ds_bpermute((lane >> 2) << 2, (int)wq);
                s[g] = (ts >> (16 * (lane & 1))) & 0xffffu;
                qq[g] = (tq >> (8 * (lane & 3))) & 0xffu;
            } else {
                s[g] = a.cube[g * NPIX + p];
                qq[g] = a.gdq[g * NPIX + p];
            }
            dk[g] = a.dark[g * NPIX + p];
            bs[g] = a.bias[g * NPIX + p];
        }
#pragma unroll
        for (int i = 0; i < NPL; ++i) pl[i] = a.planes[(size_t)i * NPIX + p];
        float acc = 0.f;
        unsigned q = 0;
#pragma unroll
        for (int g = 0; g < G; ++g) { acc += (float)s[g] - dk[g] + bs[g]; q |= qq[g] << (g & 3); }
#pragma unroll
        for (int i = 0; i < NPL; ++i) acc += pl[i];
        a.o0[p] = acc; a.o1[p] = acc * 2.f; a.o2[p] = acc * 3.f; a.o3[p] = __uint_as_float(q);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const unsigned b = (q >> g) & 0xffu;
            if (WIDE) {
                // lanes 4k .. 4k+3 -> one dword in lane k
                const unsigned b1 = (unsigned)__builtin_amdgcn_ds_bpermute(((lane * 4 + 1) & 63) << 2, (int)b);
                const unsigned b2 = (unsigned)__builtin_amdgcn_ds_bpermute(((lane * 4 + 2) & 63) << 2, (int)b);
                const unsigned b3 = (unsigned)__builtin_amdgcn_ds_bpermute(((lane * 4 + 3) & 63) << 2, (int)b);
                const unsigned b0 = (unsigned)__builtin_amdgcn_ds_bpermute(((lane * 4) & 63) << 2, (int)b);
                if (lane < 16) reinterpret_cast<unsigned *>(a.gout + g * NPIX + pw)[lane] = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
            } else {
                a.gout[g * NPIX + p] = (uint8_t)b;
            }
        }
    }
}

template <typename F>
static double time_ms(F launch, int reps = 100) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 150; ++i) launch();
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    return ms / reps;
}

int main() {
    void *big, *outb;
    const size_t in_bytes = (size_t)64 * NPIX * 4;   // 4.3 GB: up to 64 f32 planes
    CK(hipMalloc(&big, in_bytes)); CK(hipMemset(big, 1, in_bytes));
    CK(hipMalloc(&outb, (size_t)16 * NPIX * 4)); CK(hipMemset(outb, 0, (size_t)16 * NPIX * 4));
    auto report = [](const char *name, double bytes, double ms) {
        printf("%-46s %8.4f ms  %6.2f TB/s  (%.2f GB)\n", name, ms, bytes / (ms * 1e-3) / 1e12, bytes / 1e9);
        fflush(stdout);
    };
    {   // copy: 2 GB in, 2 GB out
        const size_t n4 = (size_t)32 * NPIX / 4;
        double ms = time_ms([&]() { hipLaunchKernelGGL(copy4_kernel, dim3(8192), dim3(256), 0, 0, (const float4 *)big, (float4 *)outb, n4 / 2); });
        report("copy float4 (1.07 GB in + 1.07 GB out)", (double)n4 / 2 * 32, ms);
        const size_t n4r = (size_t)56 * NPIX / 4;   // 3.76 GB read
        ms = time_ms([&]() { hipLaunchKernelGGL(read4_kernel, dim3((unsigned)(n4r / (256 * 16))), dim3(256), 0, 0, (const float4 *)big, (float *)outb, n4r); });
        report("read float4, 1/16 written", (double)n4r * 16 * (1.0 + 1.0 / 16), ms);
    }
    for (int rows_per : {8, 128}) {
        const unsigned blocks = (N / 256) * (N / rows_per);
        char nm[96];
#define STREAMS(S)                                                                                                                \
    {                                                                                                                             \
        double ms = time_ms([&]() { hipLaunchKernelGGL(streams_kernel<S>, dim3(blocks), dim3(256), 0, 0, (const float *)big, (float *)outb, rows_per); }); \
        snprintf(nm, sizeof nm, "streams S=%2d dword/lane rows_per=%d", S, rows_per);                                              \
        report(nm, (double)NPIX * 4 * (S + (S + 7) / 8), ms);                                                                      \
    }
        STREAMS(4) STREAMS(8) STREAMS(15) STREAMS(30) STREAMS(60)
    }
    {
        const int rows_per = 8;
        const unsigned blocks = (N / 256) * (N / rows_per);
        double ms = time_ms([&]() { hipLaunchKernelGGL(width_kernel<uint8_t>, dim3(blocks), dim3(256), 0, 0, (const uint8_t *)big, (float *)outb, rows_per); });
        report("width: 64 planes of u8 per lane (+4 B out)", (double)NPIX * 68, ms);
        ms = time_ms([&]() { hipLaunchKernelGGL(width_kernel<uint16_t>, dim3(blocks), dim3(256), 0, 0, (const uint16_t *)big, (float *)outb, rows_per); });
        report("width: 32 planes of u16 per lane", (double)NPIX * 68, ms);
        ms = time_ms([&]() { hipLaunchKernelGGL(width_kernel<float>, dim3(blocks), dim3(256), 0, 0, (const float *)big, (float *)outb, rows_per); });
        report("width: 16 planes of dword per lane", (double)NPIX * 68, ms);
        ms = time_ms([&]() { hipLaunchKernelGGL(width16_kernel, dim3(blocks), dim3(256), 0, 0, (const float4 *)big, (float *)outb, rows_per); });
        report("width: 4 planes of dwordx4 per lane", (double)NPIX * 68, ms);
    }
    {
        MixArgs a;
        char *b = (char *)big;
        a.cube = (const uint16_t *)b; b += G * NPIX * 2;
        a.gdq = (const uint8_t *)b; b += G * NPIX;
        a.dark = (const float *)b; b += G * NPIX * 4;
        a.bias = (const float *)b; b += G * NPIX * 4;
        a.planes = (const float *)b;
        char *o = (char *)outb;
        a.o0 = (float *)o; o += NPIX * 4; a.o1 = (float *)o; o += NPIX * 4; a.o2 = (float *)o; o += NPIX * 4; a.o3 = (float *)o; o += NPIX * 4;
        a.gout = (uint8_t *)o;
        const double bytes = (double)NPIX * (G * (2 + 1 + 4 + 4) + NPL * 4 + 16 + G);
        for (int rows_per : {8, 128}) {
            const unsigned blocks = (N / 256) * (N / rows_per);
            char nm[96];
            double ms = time_ms([&]() { hipLaunchKernelGGL(mix_kernel<0>, dim3(blocks), dim3(256), 0, 0, a, rows_per); });
            snprintf(nm, sizeof nm, "mix, one pixel per lane, rows_per=%d", rows_per);
            report(nm, bytes, ms);
            ms = time_ms([&]() { hipLaunchKernelGGL(mix_kernel<1>, dim3(blocks), dim3(256), 0, 0, a, rows_per); });
            snprintf(nm, sizeof nm, "mix, u16 / u8 as dwords of 32 / 16 lanes, rows_per=%d", rows_per);
            report(nm, bytes, ms);
        }
    }
    return 0;
}
