// probe: dependent-kernel hand-over latency: (a) one stream, implicit ordering; (b) two streams alternating,
// ordering by a device flag (kernel k+1 is dispatched while k runs and spins on the flag) (scratch)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void work_plain(double* buf, int iters) {
    double x = buf[threadIdx.x];
    for (int i = 0; i < iters; ++i) x = fma(x, 1.0000001, 1e-9);
    buf[threadIdx.x] = x;
}
__global__ void work_flag(double* buf, int iters, unsigned long long* progress, unsigned long long* counter, unsigned long long seq, unsigned nblocks) {
    if (threadIdx.x == 0) {
        long spins = 0;
        while (__hip_atomic_load(progress, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < seq - 1) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1L << 24)) break;
        }
    }
    __syncthreads();
    __builtin_amdgcn_s_dcache_inv();
    double x = buf[blockIdx.x * blockDim.x + threadIdx.x];
    for (int i = 0; i < iters; ++i) x = fma(x, 1.0000001, 1e-9);
    buf[blockIdx.x * blockDim.x + threadIdx.x] = x;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long prev = atomicAdd(counter, 1ULL);
        if (prev == (unsigned long long)nblocks * seq - 1) __hip_atomic_store(progress, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}
int main() {
    double* buf; unsigned long long *progress, *counter;
    hipMalloc(&buf, 65536 * 8); hipMemset(buf, 0, 65536 * 8);
    hipMalloc(&progress, 8); hipMalloc(&counter, 8);
    hipStream_t s[2]; hipStreamCreate(&s[0]); hipStreamCreate(&s[1]);
    const int K = 2000;
    for (int iters : {10, 2000}) for (int nb : {1, 64}) {
        // (a)
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < K; ++k) hipLaunchKernelGGL(work_plain, dim3(nb), dim3(256), 0, s[0], buf, iters);
        hipStreamSynchronize(s[0]);
        auto t1 = std::chrono::steady_clock::now();
        double a = std::chrono::duration<double, std::micro>(t1 - t0).count() / K;
        // (b)
        hipMemset(progress, 0, 8); hipMemset(counter, 0, 8);
        hipDeviceSynchronize();
        t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < K; ++k) hipLaunchKernelGGL(work_flag, dim3(nb), dim3(256), 0, s[k & 1], buf, iters, progress, counter, (unsigned long long)(k + 1), (unsigned)nb);
        hipStreamSynchronize(s[0]); hipStreamSynchronize(s[1]);
        t1 = std::chrono::steady_clock::now();
        double b = std::chrono::duration<double, std::micro>(t1 - t0).count() / K;
        unsigned long long pr; hipMemcpy(&pr, progress, 8, hipMemcpyDeviceToHost);
        printf("iters=%5d blocks=%3d: one stream %.2f us/kernel; two streams + flag %.2f us/kernel (progress %llu/%d)\n", iters, nb, a, b, pr, K);
    }
}
