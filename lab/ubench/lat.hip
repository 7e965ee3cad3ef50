// micro-latency probe for the single-wave chains of k_eig_tri (scratch, not shipped)
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../mpstime.jl_amd/csrc/mpst_internal.h"
using namespace mpst;
__device__ __forceinline__ unsigned long long clk_(double& x) {
    unsigned long long t;
    asm volatile("s_nop 0" : "+v"(x));
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    asm volatile("s_nop 0" : "+v"(x));
    return t;
}
#define clk() clk_(x)
#define REP 64
__global__ void probe(double* out, unsigned long long* cyc, double seed) {
    __shared__ double sh[1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double x = seed + lane * 1e-3, y = seed * 0.5;
    unsigned long long t0, t1;
    int k = 0;
    __syncthreads();
    // 0: dependent v_add_f64
    __syncthreads();
    t0 = clk();
#pragma unroll
    for (int i = 0; i < REP; ++i) x = x + y;
    t1 = clk(); if (threadIdx.x == 0) cyc[k] = t1 - t0; k++;
    // 1: dependent fma
    __syncthreads();
    t0 = clk();
#pragma unroll
    for (int i = 0; i < REP; ++i) x = fma(x, y, y);
    t1 = clk(); if (threadIdx.x == 0) cyc[k] = t1 - t0; k++;
    // 2: wave_sum (readlane version)
    __syncthreads();
    t0 = clk();
#pragma unroll
    for (int i = 0; i < 8; ++i) x = wave_sum(x) * 1e-2 + lane;
    t1 = clk(); if (threadIdx.x == 0) cyc[k] = t1 - t0; k++;
    // 3: wave_sum_fast
    __syncthreads();
    t0 = clk();
#pragma unroll
    for (int i = 0; i < 8; ++i) x = wave_sum_fast(x) * 1e-2 + lane;
    t1 = clk(); if (threadIdx.x == 0) cyc[k] = t1 - t0; k++;
    // 4: rcp chain
    __syncthreads();
    t0 = clk();
#pragma unroll
    for (int i = 0; i < 8; ++i) x = __builtin_amdgcn_rcp(x) + 1.5;
    t1 = clk(); if (threadIdx.x == 0) cyc[k] = t1 - t0; k++;
    // 5: rsq chain
    __syncthreads();
    t0 = clk();
#pragma unroll
    for (int i = 0; i < 8; ++i) x = __builtin_amdgcn_rsq(x) + 1.5;
    t1 = clk(); if (threadIdx.x == 0) cyc[k] = t1 - t0; k++;
    // 6: LDS write -> read same wave round trip
    __syncthreads();
    t0 = clk();
#pragma unroll
    for (int i = 0; i < 8; ++i) { sh[threadIdx.x] = x; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); x = sh[threadIdx.x ^ 1] + 1.0; }
    t1 = clk(); if (threadIdx.x == 0) cyc[k] = t1 - t0; k++;
    // 7: barrier round trip
    __syncthreads();
    t0 = clk();
#pragma unroll
    for (int i = 0; i < 8; ++i) { __syncthreads(); }
    t1 = clk(); if (threadIdx.x == 0) cyc[k] = t1 - t0; k++;
    // 8: LDS write + barrier + read
    __syncthreads();
    t0 = clk();
#pragma unroll
    for (int i = 0; i < 8; ++i) { sh[threadIdx.x] = x; __syncthreads(); x = sh[(threadIdx.x + 64) % blockDim.x] + 1.0; __syncthreads(); }
    t1 = clk(); if (threadIdx.x == 0) cyc[k] = t1 - t0; k++;
    // 9: independent fma x4 chains (ILP)
    double a = x, b = x + 1, c = x + 2, d = x + 3;
    __syncthreads();
    t0 = clk(); a += x; 
#pragma unroll
    for (int i = 0; i < REP; ++i) { a = fma(a, y, y); b = fma(b, y, y); c = fma(c, y, y); d = fma(d, y, y); }
    t1 = clk(); if (threadIdx.x == 0) cyc[k] = t1 - t0; k++;
    x = a + b + c + d;
    // 10: dpp level only (sum16) x8
    __syncthreads();
    t0 = clk();
#pragma unroll
    for (int i = 0; i < 8; ++i) x = sum16(x) * 1e-2 + lane;
    t1 = clk(); if (threadIdx.x == 0) cyc[k] = t1 - t0; k++;
    // 11: readlane pair x 8 dependent
    __syncthreads();
    t0 = clk();
#pragma unroll
    for (int i = 0; i < 8; ++i) x = readlane_f64(x, 63) + lane;
    t1 = clk(); if (threadIdx.x == 0) cyc[k] = t1 - t0; k++;
    // 12: clk overhead
    t0 = clk(); t1 = clk(); if (threadIdx.x == 0) cyc[k] = t1 - t0; k++;
    // 13: ds_read_b128 x8 batch
    {
        t0 = clk();
        double2 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = *(const double2*)&sh[2 * ((lane & 7) + 8 * i)];
        double s = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += v[i].x + v[i].y;
        t1 = clk(); if (threadIdx.x == 0) cyc[k] = t1 - t0; k++;
        x += s;
    }
    out[threadIdx.x] = x;
    (void)wave;
}
int main(int argc, char** argv) {
    double* out; unsigned long long* cyc;
    hipMalloc(&out, 1024 * 8); hipMalloc(&cyc, 64 * 8);
    const char* names[] = {"64 dep add_f64", "64 dep fma_f64", "8 wave_sum(readlane)", "8 wave_sum_fast", "8 rcp+add", "8 rsq+add",
                           "8 lds wr->rd (wave)", "8 barriers", "8 (wr,bar,rd,bar)", "64x4 indep fma", "8 sum16", "8 readlane pair+add", "clk overhead", "8 b128 batch"};
    for (int threads : {64, 256, 1024}) {
        hipMemset(cyc, 0, 64 * 8);
        hipLaunchKernelGGL(probe, dim3(1), dim3(threads), 0, 0, out, cyc, 1.25);
        hipDeviceSynchronize();
        unsigned long long h[64];
        hipMemcpy(h, cyc, 64 * 8, hipMemcpyDeviceToHost);
        printf("threads=%d\n", threads);
        for (int i = 0; i < 14; ++i) printf("  %-24s %6llu\n", names[i], h[i]);
    }
    return 0;
}
