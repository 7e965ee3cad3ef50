// Cost of one "publish, counting barrier, read the others" round between workgroups (a) spread over all XCDs with
// agent-scope operations and (b) confined to ONE XCD (workgroup ids = 0 mod 8, checked with HW_REG_XCC_ID) with
// L1-bypassing (sc0) loads and L2 atomics only.  hipcc --offload-arch=gfx950 -O3 xcd_barrier.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xF; }   // HW_REG_XCC_ID[3:0]
__device__ __forceinline__ double ld_sc0(const double* p) {
    double v;
    asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ unsigned ld_sc0_u32(const unsigned* p) {
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void atomic_inc_l2(unsigned* p) {
    asm volatile("global_atomic_add %0, %1, off" ::"v"(p), "v"(1u) : "memory");
}
// mode 0: all XCDs, agent scope, counting barrier.  mode 1: one XCD, L2 scope by hand.  mode 2: all XCDs, one flag per
// workgroup (a store each, no read-modify-write on a shared word), polled by the first wave with one coalesced load.
__global__ void k_round(double* buf, unsigned* counter, unsigned* xcc_seen, int G, int stride, int iters, int mode, int* bad, unsigned* flags) {
    if ((int)blockIdx.x % stride != 0) return;
    const int g = blockIdx.x / stride;
    if (threadIdx.x == 0) xcc_seen[g] = xcc_id();
    double acc = 0.0;
    for (int it = 0; it < iters; ++it) {
        // publish
        if (threadIdx.x == 0) {
            if (mode != 1) __hip_atomic_store(buf + (it & 1) * G + g, (double)(it * 1000 + g), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else buf[(it & 1) * G + g] = (double)(it * 1000 + g);
        }
        __syncthreads();
        if (mode == 2) {
            if (threadIdx.x == 0) __hip_atomic_store(flags + g, (unsigned)(it + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (threadIdx.x < 64) {
                int sp = 0;
                for (; sp < 2000000; ++sp) {
                    bool ok = true;
                    for (int f = threadIdx.x; f < G; f += 64) ok = ok && __hip_atomic_load(flags + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)(it + 1);
                    if (__all(ok)) break;
                }
            }
        } else if (threadIdx.x == 0) {
            if (mode == 0) {
                __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                for (int sp = 0; sp < 2000000 && __hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)G * (it + 1); ++sp) __builtin_amdgcn_s_sleep(1);
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                atomic_inc_l2(counter);
                int sp = 0;
                for (; sp < 200000 && ld_sc0_u32(counter) < (unsigned)G * (it + 1); ++sp) __builtin_amdgcn_s_sleep(1);
                if (sp >= 200000) { atomicAdd(bad, 1000000); it = iters; }
            }
        }
        __syncthreads();
        // read what the others published
        for (int f = threadIdx.x; f < G; f += 64) {
            const double v = mode != 1 ? __hip_atomic_load(buf + (it & 1) * G + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                       : ld_sc0(buf + (it & 1) * G + f);
            if (v != (double)(it * 1000 + f)) atomicAdd(bad, 1);
            acc += v;
        }
    }
    if (acc == -1.0) buf[0] = acc;
}
int main() {
    double* buf; unsigned *counter, *xcc, *flags; int* bad;
    hipMalloc(&flags, 256 * 4);
    hipMalloc(&buf, 2 * 256 * sizeof(double)); hipMalloc(&counter, 4); hipMalloc(&xcc, 256 * 4); hipMalloc(&bad, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int mode = 0; mode < 3; ++mode) {
        for (int G : {8, 16, 32, 64, 128}) {
            if (mode == 1 && G > 16) continue;
            const int stride = mode == 1 ? 8 : 1;
            hipMemset(flags, 0, 256 * 4);
            hipMemset(counter, 0, 4); hipMemset(bad, 0, 4); hipMemset(buf, 0, 2 * 256 * sizeof(double));
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_round, dim3(G * stride), dim3(64), 0, 0, buf, counter, xcc, G, stride, iters, mode, bad, flags);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned> hx(G); int hb; hipMemcpy(hx.data(), xcc, G * 4, hipMemcpyDeviceToHost); hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
            unsigned lo = 99, hi = 0; for (unsigned x : hx) { lo = x < lo ? x : lo; hi = x > hi ? x : hi; }
            printf("mode %d (%s) G=%2d: %.2f us per round, xcc ids %u..%u, stale reads %d\n", mode, mode == 1 ? "one XCD, L2" : (mode == 2 ? "all XCDs, flags" : "all XCDs, counter"), G, 1e3 * ms / iters, lo, hi, hb); fflush(stdout);
        }
    }
    return 0;
}
