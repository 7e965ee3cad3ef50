// probe: the helper chain of k_eig_tri in isolation with dependency-ordered stamps (scratch)
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../mpstime.jl_amd/csrc/mpst_internal.h"
using namespace mpst;
__device__ __forceinline__ unsigned long long stamp(double& x) {
    unsigned long long t;
    asm volatile("s_nop 0" : "+v"(x));
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    asm volatile("s_nop 0" : "+v"(x));
    return t;
}
__device__ __forceinline__ double frcp(double b) {
    double r = __builtin_amdgcn_rcp(b);
    r = fma(fma(-b, r, 1.0), r, r);
    r = fma(fma(-b, r, 1.0), r, r);
    return r;
}
__global__ void probe(double* out, unsigned long long* cyc, const double* in, int j) {
    __shared__ double p[128], vb[128], xr[128], vbn[128], misc[8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 128; i += blockDim.x) { p[i] = in[i]; vb[i] = in[128 + i]; xr[i] = in[256 + i]; }
    __syncthreads();
    if (wave != 0) return;
    const int c0 = lane, c1 = lane + 64;
    unsigned long long T[12];
    double dep = in[0];
    for (int rep = 0; rep < 3; ++rep) {
    int k = 0;
    T[k++] = stamp(dep);
    const double tau = misc[0] + dep * 0 + 0.5;
    double p0 = p[c0], p1 = p[c1], v0 = vb[c0], v1 = vb[c1], a0 = xr[c0], a1 = xr[c1];
    double pj = p[j], vj = vb[j];
    p0 += dep * 0;
    dep = p0 + p1 + v0 + v1 + a0 + a1 + pj + vj;
    T[k++] = stamp(dep);                         // 1: loads done
    double dot = wave_sum_fast(p0 * v0 + p1 * v1);
    dep = dot;
    T[k++] = stamp(dep);                         // 2: dot
    dot = dep;
    const double a2 = -0.5 * tau * dot;
    const double wj = pj + a2 * vj;
    const double g = a2 * vj + wj;
    double x0 = fma(-g, v0, fma(-vj, p0, a0));
    double x1 = fma(-g, v1, fma(-vj, p1, a1));
    dep = x0 + x1;
    T[k++] = stamp(dep);                         // 3: x'
    x0 += dep * 0;
    double s = wave_sum_fast((c0 >= j + 2 ? x0 * x0 : 0.0) + (c1 >= j + 2 ? x1 * x1 : 0.0));
    dep = s;
    T[k++] = stamp(dep);                         // 4: norm sum
    s = dep;
    const double di = j < 64 ? readlane_f64(x0, j) : readlane_f64(x1, j - 64);
    const double al = j + 1 < 64 ? readlane_f64(x0, j + 1) : readlane_f64(x1, j + 1 - 64);
    dep = di + al;
    T[k++] = stamp(dep);                         // 5: readlanes
    const double xx = fma(al, al, s) + dep * 0;
    const bool nz = s != 0.0 && xx > 1e-280;
    const double rs = __builtin_amdgcn_rsq(nz ? xx : 1.0);
    double nrm = xx * rs;
    const double hrs = 0.5 * rs;
    nrm = fma(fma(-nrm, nrm, xx), hrs, nrm);
    nrm = fma(fma(-nrm, nrm, xx), hrs, nrm);
    dep = nrm;
    T[k++] = stamp(dep);                         // 6: sqrt
    nrm = dep;
    const double bneg = copysign(nrm, al);
    const double ib = frcp(bneg), is = frcp(al + bneg);
    const double beta = nz ? -bneg : al;
    const double tauN = nz ? (bneg + al) * ib : 0.0;
    const double scale = nz ? is : 0.0;
    dep = scale + tauN + beta;
    T[k++] = stamp(dep);                         // 7: rcp
    const double sc2 = scale + dep * 0;
    const double w0 = (c0 == j + 1) ? 1.0 : (c0 > j + 1 ? x0 * sc2 : 0.0);
    const double w1 = (c1 == j + 1) ? 1.0 : (c1 > j + 1 ? x1 * sc2 : 0.0);
    vbn[c0] = w0;
    vbn[c1] = w1;
    if (lane == 0) { misc[1] = di; misc[2] = beta; misc[3] = tauN; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    dep = w0 + w1;
    T[k++] = stamp(dep);                         // 8: writes acked
    if (lane == 0 && rep == 2) for (int q = 0; q < k; ++q) cyc[q] = T[q] - T[0];
    }
    out[threadIdx.x] = dep + vbn[lane];
}
int main() {
    double *out, *in; unsigned long long* cyc;
    hipMalloc(&out, 1024 * 8); hipMalloc(&cyc, 64 * 8); hipMalloc(&in, 384 * 8);
    double h[384];
    for (int i = 0; i < 384; ++i) h[i] = 0.01 * ((i * 37) % 101) - 0.3;
    hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
    const char* names[] = {"start", "loads", "dot", "x'", "norm-sum", "readlanes", "sqrt", "rcp", "writes"};
    for (int threads : {64, 1024}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(threads), 0, 0, out, cyc, in, 61);
        hipDeviceSynchronize();
        unsigned long long c[16];
        hipMemcpy(c, cyc, 16 * 8, hipMemcpyDeviceToHost);
        printf("threads=%d (stamp overhead ~56 each)\n", threads);
        for (int i = 1; i < 9; ++i) printf("  %-10s +%llu\n", names[i], c[i] - c[i - 1]);
    }
}
