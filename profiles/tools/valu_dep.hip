// Dependent-chain VALU issue on gfx950: one wave's instruction cadence with 1 / 2 / 4 independent chains, and with a whole-wave DPP shift in the chain, at 1 / 2 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 profiles/tools/valu_dep.hip -o /tmp/valu_dep && /tmp/valu_dep        (prints cycles per instruction per wave, assuming 2.4 GHz)
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
template <int KIND> __global__ void k(float* out, int iters) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;
    const float c = 1.0000001f, d = 1e-9f;
    for (int i = 0; i < iters; i++) {
        #pragma unroll
        for (int r = 0; r < REP / 8; r++) {
            if (KIND == 0) asm volatile("v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %2\n v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %2\n v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %2\n v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %2" : "+v"(a0) : "v"(c), "v"(d));
            if (KIND == 1) asm volatile("v_mul_f32 %0, %0, %2\n v_mul_f32 %1, %1, %2\n v_add_f32 %0, %0, %3\n v_add_f32 %1, %1, %3\n v_mul_f32 %0, %0, %2\n v_mul_f32 %1, %1, %2\n v_add_f32 %0, %0, %3\n v_add_f32 %1, %1, %3" : "+v"(a0), "+v"(a1) : "v"(c), "v"(d));
            if (KIND == 2) asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n v_add_f32 %0, %0, %5\n v_add_f32 %1, %1, %5\n v_add_f32 %2, %2, %5\n v_add_f32 %3, %3, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c), "v"(d));
            if (KIND == 3) asm volatile("v_mul_f32 %0, %0, %1\n s_nop 1\n v_add_f32_dpp %0, %0, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %2\n v_mul_f32 %0, %0, %1\n s_nop 1\n v_add_f32_dpp %0, %0, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %2" : "+v"(a0) : "v"(c), "v"(d));
            if (KIND == 4) asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(c), "v"(d));
            if (KIND == 5) asm volatile("v_mul_f32 %0, %0, %1\n s_mov_b32 s20, 0\n v_add_f32 %0, %0, %2\n s_mov_b32 s20, 0\n v_mul_f32 %0, %0, %1\n s_mov_b32 s20, 0\n v_add_f32 %0, %0, %2\n s_mov_b32 s20, 0\n v_mul_f32 %0, %0, %1\n s_mov_b32 s20, 0\n v_add_f32 %0, %0, %2\n s_mov_b32 s20, 0\n v_mul_f32 %0, %0, %1\n s_mov_b32 s20, 0\n v_add_f32 %0, %0, %2\n s_mov_b32 s20, 0" : "+v"(a0) : "v"(c), "v"(d) : "s20");
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}
template <int KIND> double run(float* out, int waves_per_simd, int iters) {
    const int threads = 256 * waves_per_simd, blocks = 256;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, out, 10);
    hipEventRecord(a); hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, out, iters); hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return ms * 1e-3 * 2.4e9 / ((double)iters * REP);      // cycles per VALU instruction of one wave
}
int main() {
    float* out; hipMalloc(&out, 256 * 1024 * sizeof(float));
    const char* names[6] = {"1 chain mul/add", "2 chains", "4 chains", "1 chain, DPP wave_shr every 4th (s_nop 1 before)", "1 chain fma", "1 chain mul/add + an s_mov after every VALU"};
    for (int w : {1, 2, 3}) {
        const double r[6] = {run<0>(out, w, 20000), run<1>(out, w, 20000), run<2>(out, w, 20000), run<3>(out, w, 20000), run<4>(out, w, 20000), run<5>(out, w, 20000)};
        for (int i = 0; i < 6; i++) printf("%d wave(s) per SIMD  %-52s %6.2f cycles per VALU instruction per wave (at 2.4 GHz)\n", w, names[i], r[i]);
    }
    return 0;
}
