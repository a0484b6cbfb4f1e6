// Micro-benchmark: sustained VALU issue on MI355X at LOW occupancy (what the fused kernel runs at).
// For W waves per SIMD and C independent dependency chains per wave, time a long loop of
// (a) v_mul_f32 + v_add_f32 pairs, (b) the same on float2 (v_pk_mul_f32 / v_pk_add_f32), (c) f64 mul+add.
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int C, typename T>
__global__ void k(T *out, T a, T b, int iters) {
    T x[C];
#pragma unroll
    for (int c = 0; c < C; ++c) x[c] = a + (T)(threadIdx.x + c);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int c = 0; c < C; ++c) x[c] = x[c] * a + b;  // mul, add (not fused: -ffp-contract=off)
        }
    }
    T s = x[0];
#pragma unroll
    for (int c = 1; c < C; ++c) s = s + x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int C, typename T>
double run(int waves_per_simd, int iters, T a, T b, int ops_per_elem) {
    T *out;
    const int ncu = 256;
    const int blocks = ncu * waves_per_simd;  // 256 threads = 4 waves = 1 per SIMD
    hipMalloc(&out, sizeof(T) * blocks * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<C, T>), dim3(blocks), dim3(256), 0, 0, out, a, b, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<C, T>), dim3(blocks), dim3(256), 0, 0, out, a, b, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipFree(out);
    // wave-instructions per SIMD
    const double instr = (double)iters * 8 * C * 2 * waves_per_simd;
    return ms * 1e-3 * 2.4e9 / instr;  // cycles per wave-instruction per SIMD (at 2.4 GHz nominal)
}

int main() {
    f2 a2 = {1.0000001f, 0.9999999f}, b2 = {1e-7f, -1e-7f};
    printf("cycles per VALU wave-instruction per SIMD (2.4 GHz nominal)\n");
    printf("%-10s %6s %8s %8s %8s %8s\n", "type", "w/SIMD", "C=1", "C=2", "C=4", "C=8");
    for (int w : {1, 2, 3, 4}) {
        printf("%-10s %6d %8.2f %8.2f %8.2f %8.2f\n", "f32", w, run<1, float>(w, 20000, 1.0000001f, 1e-7f, 1), run<2, float>(w, 20000, 1.0000001f, 1e-7f, 1),
               run<4, float>(w, 20000, 1.0000001f, 1e-7f, 1), run<8, float>(w, 20000, 1.0000001f, 1e-7f, 1));
        printf("%-10s %6d %8.2f %8.2f %8.2f %8.2f\n", "f32x2(pk)", w, run<1, f2>(w, 20000, a2, b2, 2), run<2, f2>(w, 20000, a2, b2, 2),
               run<4, f2>(w, 20000, a2, b2, 2), run<8, f2>(w, 20000, a2, b2, 2));
        printf("%-10s %6d %8.2f %8.2f %8.2f %8.2f\n", "f64", w, run<1, double>(w, 20000, 1.0000001, 1e-7, 1), run<2, double>(w, 20000, 1.0000001, 1e-7, 1),
               run<4, double>(w, 20000, 1.0000001, 1e-7, 1), run<8, double>(w, 20000, 1.0000001, 1e-7, 1));
    }
    return 0;
}
