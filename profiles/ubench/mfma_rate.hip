// Microbenchmark (round 6): how many cycles does one v_mfma_f64_16x16x4_f64 occupy a SIMD's matrix pipe, and what does a DEPENDENT one
// cost?  One workgroup; W waves per SIMD (1, 2), C independent accumulator chains per wave (1, 2, 4), N MFMAs per chain; cycles from
// s_memtime (shader clock) around the loop of wave 0.  hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate && ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int C>
__global__ void k(double* out, unsigned long long* cyc, int n) {
    d4 acc[C];
    for (int c = 0; c < C; ++c) acc[c] = d4{0, 0, 0, 0};
    const double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a + c, b - c, acc[c], 0, 0, 0);      // (distinct operands: the chains must not be merged)
    }
    double s = 0;
    for (int c = 0; c < C; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
    double* out; unsigned long long* cyc;
    hipMalloc(&out, 8 * 1024 * 256); hipMalloc(&cyc, 8);
    const int n = 2000;
    for (int waves = 4; waves <= 16; waves *= 2)            // 4 waves = 1 per SIMD, 8 = 2 per SIMD, 16 = 4 per SIMD
        for (int C = 1; C <= 8; C *= 2) {
            for (int rep = 0; rep < 2; ++rep) {
                if (C == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, n);
                if (C == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, n);
                if (C == 4) hipLaunchKernelGGL(k<4>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, n);
                if (C == 8) hipLaunchKernelGGL(k<8>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, n);
            }
            unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
            const double per_simd = (double)h / ((double)n * C * (waves / 4));
            printf("waves/SIMD %d, chains/wave %d: %.1f cycles per MFMA of a wave, %.1f cycles of SIMD time per MFMA\n", waves / 4, C, (double)h / (n * C), per_simd);
        }
    return 0;
}
