// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access widths the fused kernels use (MI355X_MICROARCH.md: the counter
// reports half the bytes of wide coalesced reads; other widths are uncalibrated).  Each kernel streams a buffer of known size
// once, one element per lane per load, rows of 64 lanes at a 240-byte-per-dword-row pitch like the kernels' strips do not matter
// here: plain contiguous streaming.  Run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and compare with the sizes printed.
//   hipcc -O3 --offload-arch=gfx950 tools/gpu_checks/fetch_calib.hip -o tools/gpu_checks/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <typename T>
__global__ void stream_read(const T *__restrict__ p, size_t n, unsigned long long *out) {
    unsigned long long acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += (unsigned long long)p[i];
    if (acc == 0x1234567887654321ull) *out = acc;
}
__global__ void stream_read16(const uint4 *__restrict__ p, size_t n, unsigned long long *out) {
    unsigned long long acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = p[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 0x1234567887654321ull) *out = acc;
}
// the strips of the wave-private kernel: 64 lanes read 64 consecutive dwords, consecutive waves start 60 dwords apart
__global__ void strips_read(const uint32_t *__restrict__ p, size_t nrows, size_t rowlen, unsigned long long *out) {
    unsigned long long acc = 0;
    const size_t nstrip = rowlen / 60;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / 64, lane = threadIdx.x & 63;
    const size_t nw = (size_t)gridDim.x * blockDim.x / 64;
    for (size_t s = wave; s < nstrip * nrows; s += nw) {
        const size_t row = s / nstrip, st = s % nstrip;
        const size_t c = st * 60 + lane;
        if (c < rowlen) acc += p[row * rowlen + c];
    }
    if (acc == 0x1234567887654321ull) *out = acc;
}

int main() {
    const size_t bytes = (size_t)1 << 30;
    void *buf;
    unsigned long long *out;
    hipMalloc(&buf, bytes);
    hipMalloc(&out, 8);
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    const int grid = 256 * 16, block = 256;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(stream_read<uint8_t>, dim3(grid), dim3(block), 0, 0, (const uint8_t *)buf, bytes / 4, out);      // 256 MiB of u8
        hipLaunchKernelGGL(stream_read<uint16_t>, dim3(grid), dim3(block), 0, 0, (const uint16_t *)buf, bytes / 4, out);    // 512 MiB of u16
        hipLaunchKernelGGL(stream_read<uint32_t>, dim3(grid), dim3(block), 0, 0, (const uint32_t *)buf, bytes / 4, out);    // 1 GiB of u32
        hipLaunchKernelGGL(stream_read<unsigned long long>, dim3(grid), dim3(block), 0, 0, (const unsigned long long *)buf, bytes / 8, out);  // 1 GiB of u64
        hipLaunchKernelGGL(stream_read16, dim3(grid), dim3(block), 0, 0, (const uint4 *)buf, bytes / 16, out);              // 1 GiB of 16-byte loads
        hipLaunchKernelGGL(strips_read, dim3(grid), dim3(block), 0, 0, (const uint32_t *)buf, (size_t)65536, (size_t)4096, out);  // 1 GiB, strips
    }
    hipDeviceSynchronize();
    printf("expected bytes: u8 %zu  u16 %zu  u32 %zu  u64 %zu  b128 %zu  strips(u32, 64 lanes at a 60-dword pitch) %zu useful\n", bytes / 4, bytes / 2, bytes,
           bytes, bytes, bytes);
    return 0;
}
