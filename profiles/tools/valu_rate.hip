// VALU issue rates on gfx950: wave64 v_fma_f32, v_pk_fma_f32, v_pk_mul_f32 / v_pk_add_f32, v_mul_f32 + v_add_f32, with 1 / 2 / 4 waves per SIMD.
// Prints lane-instructions per second of the whole chip.  hipcc --offload-arch=gfx950 -O3 profiles/tools/valu_rate.hip -o gpurun_out/valu_rate && gpurun_out/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP 64
template <int KIND> __global__ void k(float* out, int iters) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a2}, p5 = {a3, a4}, p6 = {a5, a6}, p7 = {a7, a0};
    const float c = 1.0000001f, d = 1e-9f; const f2 c2 = {c, c}, d2 = {d, d};
    for (int i = 0; i < iters; i++) {
        #pragma unroll
        for (int r = 0; r < REP / 8; r++) {
            if (KIND == 0) { asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d)); }
            if (KIND == 1) { asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9"
                                          : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c2), "v"(d2)); }
            if (KIND == 2) { asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %9\n v_pk_mul_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %9\n v_pk_mul_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %9\n v_pk_mul_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %9"
                                          : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c2), "v"(d2)); }
            if (KIND == 3) { asm volatile("v_mul_f32 %0, %0, %8\n v_add_f32 %1, %1, %9\n v_mul_f32 %2, %2, %8\n v_add_f32 %3, %3, %9\n v_mul_f32 %4, %4, %8\n v_add_f32 %5, %5, %9\n v_mul_f32 %6, %6, %8\n v_add_f32 %7, %7, %9"
                                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d)); }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}
template <int KIND> double run(float* out, int waves_per_simd, int iters) {
    const int threads = 256 * waves_per_simd, blocks = 256;        // one block per CU, waves_per_simd waves on each of the 4 SIMDs
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, out, 10);
    hipEventRecord(a); hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, out, iters); hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return (double)blocks * threads * (double)iters * REP / (ms * 1e-3);
}
int main() {
    float* out; hipMalloc(&out, 256 * 1024 * sizeof(float));
    const char* names[4] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32 / v_pk_add_f32", "v_mul_f32 / v_add_f32"};
    for (int w : {1, 2, 4}) {
        const double r[4] = {run<0>(out, w, 20000), run<1>(out, w, 20000), run<2>(out, w, 20000), run<3>(out, w, 20000)};
        for (int i = 0; i < 4; i++) printf("%d wave(s) per SIMD  %-30s %7.2f T lane-instructions/s (%s)\n", w, names[i], r[i] / 1e12, (i == 1 || i == 2) ? "two floats per lane-instruction" : "one float per lane-instruction");
    }
    return 0;
}
