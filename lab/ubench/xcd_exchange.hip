// The exchange pattern of k_bt_coop between workgroups that sit on ONE XCD (every 8th workgroup of the launch), with
// L2-local operations: each of G workgroups (NT threads) writes its share of an n-vector, waits for its stores, barrier,
// one L2 atomic; polls the counter; then every thread reads the whole vector and checks it.  Variants of the store / load
// flavour tell which combination is coherent through the XCD's L2 and what a round costs.
// hipcc --offload-arch=gfx950 -O3 xcd_exchange.hip -o /tmp/xcd_exchange && /tmp/xcd_exchange
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xF; }
template <int LK> __device__ __forceinline__ double ld(const double* p) {
    double v;
    if (LK == 0) asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (LK == 1) asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (LK == 2) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dwordx2 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int SK> __device__ __forceinline__ void st(double* p, double v) {
    if (SK == 0) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    else if (SK == 1) asm volatile("global_store_dwordx2 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
    else if (SK == 2) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
template <int LK> __device__ __forceinline__ unsigned ldu(const unsigned* p) {
    unsigned v;
    if (LK == 0) asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (LK == 1) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (LK == 2) asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dword %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int SK, int LK, int INV>
__global__ void k_x(double* buf, unsigned* counter, unsigned* xcc_seen, int G, int stride, int n, int iters, int* bad) {
    if ((int)blockIdx.x % stride != 0) return;
    const int g = blockIdx.x / stride, tid = threadIdx.x, NT = blockDim.x;
    __shared__ int ok;
    if (tid == 0) xcc_seen[g] = xcc_id();
    double acc = 0.0;
    for (int it = 0; it < iters; ++it) {
        double* b = buf + (it & 1) * n;
        for (int i = g + G * tid; i < n; i += G * NT) st<SK>(b + i, (double)(it * 4096 + i));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            asm volatile("global_atomic_add %0, %1, off" ::"v"(counter), "v"(1u) : "memory");
            int sp = 0;
            for (; sp < 400000; ++sp) {
                if (INV) asm volatile("buffer_inv sc0" ::: "memory");
                if (ldu<LK>(counter) >= (unsigned)G * (it + 1)) break;
                __builtin_amdgcn_s_sleep(1);
            }
            ok = sp < 400000;
        }
        __syncthreads();
        if (!ok) { if (tid == 0) atomicAdd(bad, 1 << 20); return; }
        if (INV) asm volatile("buffer_inv sc0" ::: "memory");
        for (int i = tid; i < n; i += NT) {
            const double v = ld<LK>(b + i);
            if (v != (double)(it * 4096 + i)) atomicAdd(bad, 1);
            acc += v;
        }
    }
    if (acc == -1.0) buf[0] = acc;
}
template <int SK, int LK, int INV>
void run(const char* name, int G, int stride, int NT, int n) {
    double* buf; unsigned *counter, *xcc; int* bad;
    hipMalloc(&buf, 2 * n * sizeof(double)); hipMalloc(&counter, 4); hipMalloc(&xcc, 256 * 4); hipMalloc(&bad, 4);
    hipMemset(counter, 0, 4); hipMemset(bad, 0, 4); hipMemset(buf, 0, 2 * n * sizeof(double));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 1000;
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_x<SK, LK, INV>), dim3(G * stride), dim3(NT), 0, 0, buf, counter, xcc, G, stride, n, iters, bad);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned> hx(G); int hb; hipMemcpy(hx.data(), xcc, G * 4, hipMemcpyDeviceToHost); hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
    unsigned lo = 99, hi = 0; for (unsigned x : hx) { lo = x < lo ? x : lo; hi = x > hi ? x : hi; }
    printf("%-34s G=%2d stride=%d NT=%d n=%d: %6.2f us/round, xcc %u..%u, bad %d\n", name, G, stride, NT, n, 1e3 * ms / iters, lo, hi, hb); fflush(stdout);
    hipFree(buf); hipFree(counter); hipFree(xcc); hipFree(bad);
}
int main() {
    for (int G : {16, 32}) {
        run<0, 0, 0>("st plain, ld sc0", G, 8, 512, 512);
        run<1, 0, 0>("st sc0,   ld sc0", G, 8, 512, 512);
        run<1, 0, 1>("st sc0,   ld sc0 + buffer_inv sc0", G, 8, 512, 512);
        run<0, 0, 1>("st plain, ld sc0 + buffer_inv sc0", G, 8, 512, 512);
        run<2, 0, 0>("st sc1,   ld sc0", G, 8, 512, 512);
        run<2, 1, 0>("st sc1,   ld sc1", G, 8, 512, 512);
        run<2, 1, 0>("st sc1,   ld sc1 (all XCDs)", G, 1, 512, 512);
        run<3, 2, 0>("st sc0sc1, ld sc0sc1", G, 8, 512, 512);
        run<0, 3, 0>("st plain, ld nt", G, 8, 512, 512);
    }
    run<2, 1, 0>("st sc1,   ld sc1 (all XCDs)", 64, 1, 512, 512);
    return 0;
}
