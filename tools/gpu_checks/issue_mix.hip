// Micro-benchmark: do VALU and SALU (and LDS) instructions of DIFFERENT waves on one SIMD issue in parallel on
// MI355X?  Each wave runs a straight-line block of 256 instructions of a given mix, looped; W waves per SIMD.
// Reported: SIMD cycles per instruction-of-the-wave (2.4 GHz nominal) for VALU-only, SALU-only and mixes.
// build: hipcc -O3 --offload-arch=gfx950 issue_mix.hip -o issue_mix
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

template <int MODE>
__global__ void k(float *out, int iters) {
    float v0 = threadIdx.x, v1 = 1.5f, v2 = 2.5f, v3 = 3.5f, v4 = 0.5f, v5 = 5.5f, v6 = 6.5f, v7 = 7.5f;
    int s0 = iters, s1 = 1, s2 = 2, s3 = 3;
    __shared__ __attribute__((aligned(16))) float lds[1024];
    lds[threadIdx.x] = v0;
    __syncthreads();
    float l0 = 0, l1 = 0;
    const unsigned la = (threadIdx.x & 255) * 4, la8 = (threadIdx.x & 255) * 8;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {  // 128 VALU (4 independent chains)
            asm volatile(REP16("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n"
                               "v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n")
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4));
        } else if (MODE == 1) {  // 128 SALU (4 chains)
            asm volatile(REP16("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 3\n s_add_u32 %2, %2, 5\n s_add_u32 %3, %3, 7\n"
                               "s_xor_b32 %0, %0, %1\n s_xor_b32 %1, %1, %2\n s_xor_b32 %2, %2, %3\n s_xor_b32 %3, %3, %0\n")
                         : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) :: "scc");
        } else if (MODE == 2) {  // 128 VALU + 128 SALU interleaved 1:1
            asm volatile(REP16("v_mul_f32 %0, %0, %8\n s_add_u32 %4, %4, 1\n v_mul_f32 %1, %1, %8\n s_add_u32 %5, %5, 3\n"
                               "v_mul_f32 %2, %2, %8\n s_add_u32 %6, %6, 5\n v_mul_f32 %3, %3, %8\n s_add_u32 %7, %7, 7\n"
                               "v_add_f32 %0, %0, %8\n s_xor_b32 %4, %4, %5\n v_add_f32 %1, %1, %8\n s_xor_b32 %5, %5, %6\n"
                               "v_add_f32 %2, %2, %8\n s_xor_b32 %6, %6, %7\n v_add_f32 %3, %3, %8\n s_xor_b32 %7, %7, %4\n")
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(v4) : "scc");
        } else if (MODE == 3) {  // 128 VALU + 32 SALU (4:1)
            asm volatile(REP16("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n s_add_u32 %4, %4, 1\n"
                               "v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n s_xor_b32 %5, %5, %4\n")
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(v4) : "scc");
        } else if (MODE == 4) {  // 128 packed f32 (4 chains)
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 a = {v0, v1}, b = {v2, v3}, c = {v5, v6}, d = {v7, v0}, m = {v4, v4};
            asm volatile(REP16("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                               "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            v0 = a.x + a.y, v1 = b.x + b.y, v2 = c.x + c.y, v3 = d.x + d.y;
        } else if (MODE == 5) {  // 128 VALU + 16 ds_read_b32
            asm volatile(REP16("v_mul_f32 %[a], %[a], %[m]\n v_mul_f32 %[b], %[b], %[m]\n v_mul_f32 %[c], %[c], %[m]\n v_mul_f32 %[d], %[d], %[m]\n"
                               "ds_read_b32 %[l], %[la]\n"
                               "v_add_f32 %[a], %[a], %[m]\n v_add_f32 %[b], %[b], %[m]\n v_add_f32 %[c], %[c], %[m]\n v_add_f32 %[d], %[d], %[m]\n")
                         : [a] "+v"(v0), [b] "+v"(v1), [c] "+v"(v2), [d] "+v"(v3), [l] "+v"(l0) : [m] "v"(v4), [la] "v"(la));
            asm volatile("s_waitcnt lgkmcnt(0)");
        } else if (MODE == 6) {  // 128 dependent VALU, single chain
            asm volatile(REP64("v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n") : "+v"(v0) : "v"(v4));
        } else if (MODE == 8) {  // 128 pk + 128 SALU interleaved
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 a = {v0, v1}, b = {v2, v3}, c = {v5, v6}, d = {v7, v0}, m = {v4, v4};
            asm volatile(REP16("v_pk_mul_f32 %0, %0, %8\n s_add_u32 %4, %4, 1\n v_pk_mul_f32 %1, %1, %8\n s_add_u32 %5, %5, 3\n"
                               "v_pk_mul_f32 %2, %2, %8\n s_add_u32 %6, %6, 5\n v_pk_mul_f32 %3, %3, %8\n s_add_u32 %7, %7, 7\n"
                               "v_pk_add_f32 %0, %0, %8\n s_xor_b32 %4, %4, %5\n v_pk_add_f32 %1, %1, %8\n s_xor_b32 %5, %5, %6\n"
                               "v_pk_add_f32 %2, %2, %8\n s_xor_b32 %6, %6, %7\n v_pk_add_f32 %3, %3, %8\n s_xor_b32 %7, %7, %4\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(m) : "scc");
            v0 = a.x + a.y, v1 = b.x + b.y, v2 = c.x + c.y, v3 = d.x + d.y;
        } else if (MODE == 9) {  // 128 pk + 32 ds_read_b64
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 a = {v0, v1}, b = {v2, v3}, c = {v5, v6}, d = {v7, v0}, m = {v4, v4}, l = {l0, l1};
            asm volatile(REP16("v_pk_mul_f32 %[a], %[a], %[m]\n v_pk_mul_f32 %[b], %[b], %[m]\n ds_read_b64 %[l], %[la]\n v_pk_mul_f32 %[c], %[c], %[m]\n v_pk_mul_f32 %[d], %[d], %[m]\n"
                               "v_pk_add_f32 %[a], %[a], %[m]\n v_pk_add_f32 %[b], %[b], %[m]\n ds_read_b64 %[l], %[la] offset:8\n v_pk_add_f32 %[c], %[c], %[m]\n v_pk_add_f32 %[d], %[d], %[m]\n")
                         : [a] "+v"(a), [b] "+v"(b), [c] "+v"(c), [d] "+v"(d), [l] "+v"(l) : [m] "v"(m), [la] "v"(la8));
            asm volatile("s_waitcnt lgkmcnt(0)");
            v0 = a.x + a.y, v1 = b.x + b.y, v2 = c.x + c.y, v3 = d.x + d.y, l0 = l.x, l1 = l.y;
        } else if (MODE == 10) {  // 128 f32 + 32 global_load_dword (cache hits; `out` holds >= 64 KB)
            asm volatile(REP16("v_mul_f32 %[a], %[a], %[m]\n v_mul_f32 %[b], %[b], %[m]\n global_load_dword %[l], %[la], %[p]\n v_mul_f32 %[c], %[c], %[m]\n v_mul_f32 %[d], %[d], %[m]\n"
                               "v_add_f32 %[a], %[a], %[m]\n v_add_f32 %[b], %[b], %[m]\n global_load_dword %[l], %[la], %[p] offset:1024\n v_add_f32 %[c], %[c], %[m]\n v_add_f32 %[d], %[d], %[m]\n")
                         : [a] "+v"(v0), [b] "+v"(v1), [c] "+v"(v2), [d] "+v"(v3), [l] "+v"(l0) : [m] "v"(v4), [la] "v"(la), [p] "s"(out));
            asm volatile("s_waitcnt vmcnt(0)");
        } else if (MODE == 11) {  // 64 pk + 64 f32 + 64 SALU + 16 ds_read_b64 (kernel-like mix)
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 a = {v0, v1}, b = {v2, v3}, m = {v4, v4}, l = {l0, l1};
            asm volatile(REP16("v_pk_mul_f32 %[a], %[a], %[m]\n s_add_u32 %[s0], %[s0], 1\n v_mul_f32 %[c], %[c], %[n]\n v_pk_add_f32 %[b], %[b], %[m]\n s_add_u32 %[s1], %[s1], 3\n v_add_f32 %[d], %[d], %[n]\n ds_read_b64 %[l], %[la]\n"
                               "v_pk_add_f32 %[a], %[a], %[m]\n s_xor_b32 %[s0], %[s0], %[s1]\n v_mul_f32 %[d], %[d], %[n]\n v_pk_mul_f32 %[b], %[b], %[m]\n s_xor_b32 %[s1], %[s1], %[s2]\n v_add_f32 %[c], %[c], %[n]\n")
                         : [a] "+v"(a), [b] "+v"(b), [c] "+v"(v5), [d] "+v"(v6), [s0] "+s"(s0), [s1] "+s"(s1), [s2] "+s"(s2), [l] "+v"(l)
                         : [m] "v"(m), [n] "v"(v4), [la] "v"(la8) : "scc");
            asm volatile("s_waitcnt lgkmcnt(0)");
            v0 = a.x + a.y, v1 = b.x + b.y, l0 = l.x, l1 = l.y;
        } else if (MODE == 7) {  // 64 v_cndmask + 64 v_mul (4 chains)
            asm volatile(REP16("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n"
                               "v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n")
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4) : "vcc");
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = v0 + v1 + v2 + v3 + (float)(s0 + s1 + s2 + s3) + l0 + l1;
}

template <int MODE>
double run(int w, int iters, double instr_per_iter) {
    float *out;
    const int blocks = 256 * w;
    hipMalloc(&out, sizeof(float) * blocks * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipFree(out);
    return ms * 1e-3 * 2.4e9 / ((double)iters * w);  // SIMD cycles per block-iteration of one wave
}

int main() {
    printf("SIMD cycles per loop iteration per resident wave (2.4 GHz nominal)\n");
    printf("%-44s %8s %8s %8s %8s\n", "mix per iteration", "w=1", "w=2", "w=4", "w=8");
#define ROW(M, name) printf("%-44s %8.0f %8.0f %8.0f %8.0f\n", name, run<M>(1, 4000, 0), run<M>(2, 4000, 0), run<M>(4, 4000, 0), run<M>(8, 4000, 0));
    ROW(0, "128 VALU f32 (4 chains)")
    ROW(6, "128 VALU f32 (1 chain)")
    ROW(4, "128 VALU pk_f32 (4 chains)")
    ROW(7, "64 v_mul + 64 v_cndmask (4 chains)")
    ROW(1, "128 SALU")
    ROW(2, "128 VALU + 128 SALU interleaved")
    ROW(3, "128 VALU + 32 SALU")
    ROW(5, "128 VALU + 16 ds_read_b32 (+wait)")
    ROW(8, "128 pk + 128 SALU interleaved")
    ROW(9, "128 pk + 32 ds_read_b64")
    ROW(10, "128 f32 + 32 global_load_dword")
    ROW(11, "64 pk + 64 f32 + 64 SALU + 16 ds_read_b64")
    return 0;
}
