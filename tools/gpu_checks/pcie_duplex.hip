// PCIe copy rates with page-locked host memory: H2D alone, D2H alone, both at once on two streams -- free-running and as
// a pipeline (upload -> kernel -> download chained by events, two buffer sets), which is what a batched host boundary does.
// build: hipcc -O2 --offload-arch=gfx950 pcie_duplex.hip -o pcie_duplex
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void touch(float *p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] += 1.0f;
}
int main() {
    const size_t n = 448ull << 20;
    void *h1[2], *h2[2], *d1[2], *d2[2];
    for (int k = 0; k < 2; ++k) {
        hipHostMalloc(&h1[k], n, hipHostMallocDefault);
        hipHostMalloc(&h2[k], n, hipHostMallocDefault);
        hipMalloc(&d1[k], n);
        hipMalloc(&d2[k], n);
    }
    hipStream_t a, b, c;
    hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    hipStreamCreate(&c);
    auto run = [&](int mode) {
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 4; ++i) {
            if (mode & 1) hipMemcpyAsync(d1[0], h1[0], n, hipMemcpyHostToDevice, a);
            if (mode & 2) hipMemcpyAsync(h2[0], d2[0], n, hipMemcpyDeviceToHost, b);
        }
        hipDeviceSynchronize();
        double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return 4.0 * n / s / 1e9;
    };
    run(3);
    printf("H2D alone %.1f GB/s\n", run(1));
    printf("D2H alone %.1f GB/s\n", run(2));
    printf("both, free-running: %.1f GB/s each direction\n", run(3));
    // pipeline with events
    hipEvent_t ein[2], edone[2], eout[2];
    for (int k = 0; k < 2; ++k) {
        hipEventCreateWithFlags(&ein[k], hipEventDisableTiming);
        hipEventCreateWithFlags(&edone[k], hipEventDisableTiming);
        hipEventCreateWithFlags(&eout[k], hipEventDisableTiming);
    }
    for (int rep = 0; rep < 2; ++rep) {
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        const int N = 8;
        for (int i = 0; i < N; ++i) {
            const int k = i & 1;
            if (i >= 2) hipStreamWaitEvent(a, edone[k], 0);
            hipMemcpyAsync(d1[k], h1[k], n, hipMemcpyHostToDevice, a);
            hipEventRecord(ein[k], a);
            hipStreamWaitEvent(c, ein[k], 0);
            if (i >= 2) hipStreamWaitEvent(c, eout[k], 0);
            hipLaunchKernelGGL(touch, dim3(4096), dim3(256), 0, c, (float *)d2[k], (size_t)1 << 20);
            hipEventRecord(edone[k], c);
            hipStreamWaitEvent(b, edone[k], 0);
            hipMemcpyAsync(h2[k], d2[k], n, hipMemcpyDeviceToHost, b);
            hipEventRecord(eout[k], b);
        }
        hipDeviceSynchronize();
        double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("pipeline with events: %.1f ms per item (%.1f GB/s each direction)\n", 1e3 * s / N, N * n / s / 1e9);
    }
    return 0;
}
