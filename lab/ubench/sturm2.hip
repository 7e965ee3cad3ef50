// probe: throughput of Sturm counts vs waves per SIMD (scratch)
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ int hi32(double x) { return __double2hiint(x); }
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
            sb = __builtin_amdgcn_alignbit(sb, (unsigned)hi32(pn), 31);
            pp = p;
            p = pn;
        }
        cnt += __popc((sb ^ (sb >> 1)) & 0xffu);
        int e = __builtin_amdgcn_frexp_exp(p);
        if (p == 0.0) e = __builtin_amdgcn_frexp_exp(pp);
        p = __builtin_amdgcn_ldexp(p, -e);
        pp = __builtin_amdgcn_ldexp(pp, -e);
    }
    return cnt;
}
template <int PTS>
__global__ void probe(const double* in, int* out, int n, int reps) {
    __shared__ double de[272];
    for (int i = threadIdx.x; i < 272; i += blockDim.x) de[i] = in[i];
    __syncthreads();
    int acc = 0;
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int q = 0; q < PTS; ++q) {
            double x = -1.0 + 2.0 * (threadIdx.x * PTS + q) / (blockDim.x * PTS) + acc * 1e-30;
            acc += sturm(de, n, x);
        }
        __syncthreads();
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int PTS>
__global__ void probe_ilp(const double* in, int* out, int n, int reps) {   // PTS points interleaved in one loop
    __shared__ double de[272];
    for (int i = threadIdx.x; i < 272; i += blockDim.x) de[i] = in[i];
    __syncthreads();
    const double2* de2 = (const double2*)de;
    int acc = 0;
    for (int r = 0; r < reps; ++r) {
        double x[PTS], p[PTS], pp[PTS]; unsigned sb[PTS]; int cnt[PTS];
#pragma unroll
        for (int q = 0; q < PTS; ++q) { x[q] = -1.0 + 2.0 * (threadIdx.x * PTS + q) / (blockDim.x * PTS) + acc * 1e-30; pp[q] = 1.0; p[q] = de[0] - x[q]; sb[q] = ((unsigned)hi32(p[q])) >> 31; cnt[q] = sb[q]; }
        for (int j = 1; j < n; j += 8) {
            double2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = de2[j + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#pragma unroll
                for (int q = 0; q < PTS; ++q) {
                    const double t = v[u].x - x[q];
                    const double pn = fma(t, p[q], -v[u].y * pp[q]);
                    sb[q] = __builtin_amdgcn_alignbit(sb[q], (unsigned)hi32(pn), 31);
                    pp[q] = p[q]; p[q] = pn;
                }
            }
#pragma unroll
            for (int q = 0; q < PTS; ++q) {
                cnt[q] += __popc((sb[q] ^ (sb[q] >> 1)) & 0xffu);
                int e = __builtin_amdgcn_frexp_exp(p[q]);
                if (p[q] == 0.0) e = __builtin_amdgcn_frexp_exp(pp[q]);
                p[q] = __builtin_amdgcn_ldexp(p[q], -e); pp[q] = __builtin_amdgcn_ldexp(pp[q], -e);
            }
        }
#pragma unroll
        for (int q = 0; q < PTS; ++q) acc += cnt[q];
        __syncthreads();
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <typename F> float timeit(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms * 1e3f;
}
int main() {
    double h[272];
    for (int i = 0; i < 136; ++i) { h[2 * i] = 0.3 * ((i * 37) % 11 - 5) / 5.0; h[2 * i + 1] = 0.01 * ((i * 13) % 7 + 1); }
    double* in; int* out;
    hipMalloc(&in, sizeof h); hipMalloc(&out, 64 * 1024 * 4);
    hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
    const int reps = 70;
    for (int threads : {64, 256, 512, 1024}) {
        float us = timeit([&] { hipLaunchKernelGGL(probe<1>, dim3(32), dim3(threads), 0, 0, in, out, 128, reps); });
        printf("threads=%4d 1 pt/thread: %.2f us per count-round (%.0f cycles)\n", threads, us / reps, us / reps * 2400);
    }
    for (int threads : {256, 512}) {
        float us = timeit([&] { hipLaunchKernelGGL(probe_ilp<2>, dim3(32), dim3(threads), 0, 0, in, out, 128, reps); });
        printf("threads=%4d 2 pts interleaved: %.2f us per round\n", threads, us / reps);
        us = timeit([&] { hipLaunchKernelGGL(probe_ilp<4>, dim3(32), dim3(threads), 0, 0, in, out, 128, reps); });
        printf("threads=%4d 4 pts interleaved: %.2f us per round\n", threads, us / reps);
    }
}
