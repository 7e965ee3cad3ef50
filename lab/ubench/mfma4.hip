// probe: lane layout of v_mfma_f64_4x4x4_4b_f64 and its use as a 64-lane sum (scratch)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(double* out) {
    const int l = threadIdx.x;
    const double a = (double)(1 << (l & 15)) + 65536.0 * (l >> 4) ;   // bit per lane-in-block, block id in high part
    double d1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, 1.0, 0.0, 0, 0, 0);    // sum over k of A[i][k]
    double d2 = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, d1, 0.0, 0, 0, 0);   // sum over k of B[k][j]
    double e1 = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, a, 0.0, 0, 0, 0);    // sum over k of B[k][j]
    double e2 = __builtin_amdgcn_mfma_f64_4x4x4f64(e1, 1.0, 0.0, 0, 0, 0);
    out[l] = d1; out[64 + l] = d2; out[128 + l] = e1; out[192 + l] = e2;
}
int main() {
    double* out; hipMalloc(&out, 256 * 8);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out);
    double h[256]; hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    const char* nm[4] = {"d1 = mfma(a,1)", "d2 = mfma(1,d1)", "e1 = mfma(1,a)", "e2 = mfma(e1,1)"};
    for (int t = 0; t < 4; ++t) {
        printf("%s\n", nm[t]);
        for (int l = 0; l < 20; ++l) { double v = h[64 * t + l]; long long lo = (long long)v % 65536; long long hi = (long long)v / 65536; printf("  lane %2d: lanes-mask %04llx blk-part %lld\n", l, lo, hi); }
    }
}
