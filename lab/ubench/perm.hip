// probe: semantics of v_permlane32_swap / v_permlane16_swap for a 4-row (stride-16) lane sum (scratch)
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double row4_sum(double x) {
    unsigned lo = __double2loint(x), hi = __double2hiint(x);
    auto r0 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto r1 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    double a = __hiloint2double(r1[0], r0[0]), b = __hiloint2double(r1[1], r0[1]);
    x = a + b;
    lo = __double2loint(x); hi = __double2hiint(x);
    auto s0 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto s1 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    a = __hiloint2double(s1[0], s0[0]); b = __hiloint2double(s1[1], s0[1]);
    return a + b;
}
__global__ void probe(double* out) {
    const int l = threadIdx.x;
    double x = (l & 15) * 1000.0 + (1 << (l >> 4));     // position*1000 + 2^row
    out[l] = row4_sum(x);
}
int main() {
    double* out; hipMalloc(&out, 64 * 8);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out);
    double h[64]; hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    int ok = 1;
    for (int l = 0; l < 64; ++l) if (h[l] != (l & 15) * 4000.0 + 15.0) ok = 0;
    printf("row4_sum %s; lane0 %.0f lane17 %.0f lane63 %.0f\n", ok ? "OK" : "WRONG", h[0], h[17], h[63]);
}
