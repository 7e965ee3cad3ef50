// probe: cycles of one Sturm count (128 rows) in the k_eig_vec configuration (scratch)
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ int hi32(double x) { return __double2hiint(x); }
__device__ __forceinline__ unsigned long long stamp(double& x) {
    unsigned long long t;
    asm volatile("s_nop 0" : "+v"(x));
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    asm volatile("s_nop 0" : "+v"(x));
    return t;
}
template <int MODE>
__device__ __forceinline__ int sturm(const double* __restrict__ de, int n, double x) {
    const double2* __restrict__ de2 = (const double2*)de;
    double pp = 1.0, p = de[0] - x;
    unsigned sb = ((unsigned)hi32(p)) >> 31;
    int cnt = (int)sb;
    for (int j = 1; j < n; j += 8) {
        double2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = de2[j + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const double t = v[u].x - x;
            const double pn = fma(t, p, -v[u].y * pp);
            if (MODE != 1) sb = __builtin_amdgcn_alignbit(sb, (unsigned)hi32(pn), 31);
            pp = p;
            p = pn;
        }
        if (MODE != 1) cnt += __popc((sb ^ (sb >> 1)) & 0xffu);
        if (MODE != 2) {
            int e = __builtin_amdgcn_frexp_exp(p);
            if (p == 0.0) e = __builtin_amdgcn_frexp_exp(pp);
            p = __builtin_amdgcn_ldexp(p, -e);
            pp = __builtin_amdgcn_ldexp(pp, -e);
        }
    }
    if (MODE == 1) cnt = hi32(p);
    return cnt;
}
__device__ __forceinline__ int sturm_pipe(const double* __restrict__ de, int n, double x) {
    const double2* __restrict__ de2 = (const double2*)de;
    double p = de[0] - x;
    double q = de[3];                 // e_0^2 * P_0
    unsigned sb = ((unsigned)hi32(p)) >> 31;
    int cnt = (int)sb;
    for (int j = 1; j < n; j += 8) {
        double2 v[8];
        double en[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = de2[j + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) en[u] = de[2 * (j + u + 1) + 1];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const double t = v[u].x - x;
            const double pn = fma(t, p, -q);
            q = en[u] * p;            // e_{j}^2 * P_j for the next row: independent of pn
            sb = __builtin_amdgcn_alignbit(sb, (unsigned)hi32(pn), 31);
            p = pn;
        }
        cnt += __popc((sb ^ (sb >> 1)) & 0xffu);
        int e = __builtin_amdgcn_frexp_exp(p);
        if (p == 0.0) e = __builtin_amdgcn_frexp_exp(q);
        p = __builtin_amdgcn_ldexp(p, -e);
        q = __builtin_amdgcn_ldexp(q, -e);
    }
    return cnt;
}
__global__ void probe(const double* in, unsigned long long* cyc, int* out, int n) {
    __shared__ double de[272];
    for (int i = threadIdx.x; i < 272; i += blockDim.x) de[i] = in[i];
    __syncthreads();
    double x = -1.0 + 2.0 * threadIdx.x / blockDim.x;
    unsigned long long t0, t1;
    int acc = 0;
    t0 = stamp(x); acc += sturm<0>(de, n, x); x += acc * 1e-30; t1 = stamp(x);
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    t0 = stamp(x); acc += sturm<0>(de, n, x); x += acc * 1e-30; t1 = stamp(x);
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[1] = t1 - t0;
    t0 = stamp(x); acc += sturm<1>(de, n, x); x += acc * 1e-30; t1 = stamp(x);
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[2] = t1 - t0;
    t0 = stamp(x); acc += sturm<2>(de, n, x); x += acc * 1e-30; t1 = stamp(x);
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[3] = t1 - t0;
    t0 = stamp(x); acc += sturm_pipe(de, n, x); x += acc * 1e-30; t1 = stamp(x);
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[4] = t1 - t0;
    int c0 = sturm<0>(de, n, 0.137), c1 = sturm_pipe(de, n, 0.137);
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[5] = (unsigned long long)(c0 * 1000 + c1);
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
    double h[272];
    for (int i = 0; i < 136; ++i) { h[2 * i] = 0.3 * ((i * 37) % 11 - 5) / 5.0; h[2 * i + 1] = 0.01 * ((i * 13) % 7 + 1); }
    double* in; unsigned long long* cyc; int* out;
    hipMalloc(&in, sizeof h); hipMalloc(&cyc, 128); hipMalloc(&out, 32 * 1024 * 4);
    hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep)
    for (int threads : {64, 256, 1024}) for (int blocks : {1}) {
        hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), 0, 0, in, cyc, out, 128);
        hipDeviceSynchronize();
        unsigned long long c[6]; hipMemcpy(c, cyc, 48, hipMemcpyDeviceToHost);
        if (rep) printf("threads=%4d blocks=%2d: count %llu / %llu cyc; no-sign %llu; no-rescale %llu  (per step %.1f) pipe %llu check %llu\n", threads, blocks, c[0], c[1], c[2], c[3], c[1] / 127.0, c[4], c[5]);
    }
}
