// One-shot direct-write all-reduce of the bond gradient over xGMI: the message is small (2 + C (d chi)^2
// doubles = 262 KB at the headline shape) and sits on the critical path of every bond, so it is latency-, not
// bandwidth-bound.  Every rank owns an INBOX with one slot per rank; a rank pushes its contribution into the slot it
// owns in every peer's inbox (7 posted writes streams over the 7 point-to-point links, no read round trips, no ring
// steps), publishes a flag per peer behind a system-scope release, waits for the flags in its OWN memory (local polls),
// and sums the slots in rank order - the same order on every rank, so all replicas hold the same bits.
//
// Inboxes are double-buffered by the parity of a per-call epoch: a rank can only start pushing epoch e+2 after it has
// seen every peer's flag of epoch e+1, which a peer raises after it has finished reading epoch e.  One flag per peer
// and call is therefore enough.  Every spin is bounded (ArParams.spin_limit shader cycles: 10 s unless MPST_AR_TIMEOUT_S
// says otherwise - first-use peer mapping, a redone eigensolve on one rank or host jitter must not trip it); on time-out
// the device-side status word is set, the call returns MPST_ERR_DEVICE instead of hanging the node, and the host retires
// the one-shot path: flags, epochs and slots are then in an undefined cross-rank state.
#include "mpst_internal.h"
#include <cstdlib>

namespace mpst {

// Workgroups of the all-reduce kernel.  They must all be resident at once: every workgroup waits for the flags of ALL ranks,
// its own included, and a rank's flag is raised by the last of its workgroups to have pushed.  64 << 256 CUs on a GPU a rank has
// to itself.  When several ranks SHARE one GPU (bench / test mode on 1-GPU boxes) the spinning workgroups of all of them
// together can cover every CU - 8 ranks x 64 = 512 workgroups - and a peer's next compute kernel that needs an empty CU
// (k_eig_trivec: 1024 threads, 157 KB of LDS) is never placed: everybody then waits for that peer's flag until the time-out.
// That is what lost `sharded_n32768` in round 3's 8-rank run (reproduced in profiles/r04_oneshot_shared_gpu.txt);
// MPST_AR_WG caps the workgroups, bench.py sets it to 4 when MPST_BENCH_SHARE_GPU=1.
constexpr int AR_WG = 64;

__global__ __launch_bounds__(256) void k_allreduce_oneshot(ArParams p) {
    __shared__ int last;
    if (*p.status != 0) return;      // an earlier call timed out (or the decomposition failed): do not wait for peers again
    const int par = (int)(p.epoch & 1ull);
    int64_t n = p.n_fixed;
    if (p.lid >= 0) n = 2 + (int64_t)p.C * ((int64_t)p.d * p.chi[p.lid]) * ((int64_t)p.d * p.chi[p.lid + 2]);
    const int64_t base = ((int64_t)par * p.nranks + p.rank) * p.slot;
    const int64_t stride = (int64_t)gridDim.x * 256;
    // ---- push: my message into my slot of every inbox ---------------------------------------------------------
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const double x = p.buf[i];
        for (int r = 0; r < p.nranks; ++r) __builtin_nontemporal_store(x, p.inbox[r] + base + i);
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int t = atomicAdd(p.counter + par, 1u);
        last = (t == gridDim.x - 1);
    }
    __syncthreads();
    if (last) {
        // every workgroup of this rank has pushed and fenced: raise my flag in every peer's flag block
        __threadfence_system();
        if (threadIdx.x == 0) p.counter[par] = 0;
        if (threadIdx.x < (unsigned)p.nranks)
            __hip_atomic_store(p.flags[threadIdx.x] + par * 8 + p.rank, p.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // ---- wait for every rank's flag in my own flag block ------------------------------------------------------
    if (threadIdx.x < (unsigned)p.nranks) {
        const unsigned long long* f = p.flags[p.rank] + par * 8 + threadIdx.x;
        const long long t0 = __builtin_readcyclecounter();
        while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != p.epoch) {
            __builtin_amdgcn_s_sleep(2);
            if (__builtin_readcyclecounter() - t0 > p.spin_limit) {
                *p.status = MPST_ERR_DEVICE;
                if (p.dbg) {            // who was missing, what its flag held, which call this was: the host puts it in the error text
                    p.dbg[0] = (int32_t)threadIdx.x;
                    p.dbg[1] = (int32_t)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) & 0x7fffffffull);
                }
                break;
            }
        }
    }
    __syncthreads();
    // ---- sum the slots in rank order --------------------------------------------------------------------------
    const double* in = p.inbox[p.rank] + (int64_t)par * p.nranks * p.slot;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        // system-scope loads: the slots were written by other devices and the same addresses are reused every other call
        double s = __hip_atomic_load(in + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        for (int r = 1; r < p.nranks; ++r) s += __hip_atomic_load(in + (int64_t)r * p.slot + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        p.buf[i] = s;
    }
}

void launch_allreduce_oneshot(const ArParams& p, hipStream_t s) {
    static const int wg = [] {
        const char* e = getenv("MPST_AR_WG");
        const int v = e ? atoi(e) : AR_WG;
        return v < 1 ? 1 : (v > AR_WG ? AR_WG : v);
    }();
    hipLaunchKernelGGL(k_allreduce_oneshot, dim3(wg), dim3(256), 0, s, p);
}

}  // namespace mpst
