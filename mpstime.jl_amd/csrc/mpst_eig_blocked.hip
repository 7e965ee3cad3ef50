// Symmetric eigensolver for Gram matrices beyond 128 x 128 (n = d*chi up to DIM_LIMIT): hand-written blocked Householder
// tridiagonalisation spread over the chip, bisection + twisted factorisation per wanted eigenvalue, reflector
// back-transformation, verification.  Stand-in for ITensors.svd -> LAPACK gesdd in decomposeBT
// (src/Training/RealRealHighDimension.jl:166-169,185-188) at the sizes of the reference's documented runs
// (d = 8..12, chi_max = 37..64: docs/src/hyperparameters.md:65,127,241-245).  A multi-workgroup one-sided Jacobi
// iteration (mpst_eig.hip: slow, robust, hand-written like the rest - no vendor solver is linked) stays as the fallback when the on-device
// verification rejects the result (clustered kept eigenvalues).
//
// One dependent launch per Householder step instead of LAPACK's symv / dot / axpy / syr2 sequence (dsytd2):
//   k_bt_step  (G workgroups) step j.  Every workgroup redundantly finishes the previous step's w (the one global
//              reduction a step needs, alpha = -tau/2 y^T v, is deferred to here), applies that pending rank-2 update to
//              row j and forms the reflector; then it updates ITS rows of the trailing matrix with the previous step's
//              (v, w) and takes their products with the new v in the same pass: the matrix is read and written once per
//              step, coalesced by rows, and no workgroup waits for another inside a launch.
//   k_bt_vec   one workgroup per wanted eigenvalue: 256-way multisection Sturm bisection (sturm_count of the small
//              solver), twisted factorisation (forward / backward pivots on two waves), back-transformation through the
//              n-2 reflectors.
//   k_bt_gram / k_bt_decide / k_bt_polish: Z^T Z - I on the MFMA, truncation rule + verdict (or the fallback flag),
//              up to two Loewdin rounds E = Z (I - D/2) on the MFMA, publication.
// The live size n is read on the device (bond dimensions never visit the host); launches are laid out for the capacity.
#include <atomic>
#include "mpst_internal.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace mpst {

constexpr int BT_G = 128;          // workgroups of a Householder step
constexpr int BT_T = 256;
constexpr int BT_NMAX = DIM_LIMIT; // 1024
constexpr int BT_TAIL = 128;       // once the trailing matrix is this small the one-workgroup reduction of mpst_eig.hip finishes it (k_eig_tail)
// first step the tail kernel takes over, or a value no step reaches (no hand-over: small n, or MPST_BT_NO_TAIL)
__device__ __forceinline__ int bt_tail_start(int n, int use_tail) { return (use_tail && n > BT_TAIL) ? n - BT_TAIL : (1 << 30); }

struct BtProblem {
    const double* G;
    int n, rows, nspec, K0;
    bool pair;      // real embedding of a complex Hermitian matrix (View::zw == 2): all counts are the doubled ones, the vector
                    // kernel computes one eigenvector per eigenvalue pair and stores its partner J u next to it
};
__device__ __forceinline__ int bt_nvec(const BtProblem& p) { return p.pair ? p.K0 >> 1 : p.K0; }
__device__ __forceinline__ BtProblem bt_resolve(const View& v, int lid, int going_left, const double* rawG, int rawn) {
    BtProblem p;
    if (rawn > 0) {
        p.G = rawG;
        p.n = rawn;
        p.rows = rawn;
        p.nspec = rawn;
        p.K0 = rawn < CAP_LIMIT ? rawn : CAP_LIMIT;
        p.pair = v.zw == 2 && (rawn & 1) == 0;                    // test hook (mpst_selftest_eig, alg bit 2)
        if (p.pair) p.K0 &= ~1;
    } else {
        const int zw = view_zw(v);
        const int Dl = v.chi[lid], Dr = v.chi[lid + 2];
        const int X = Dl * v.d, Y = v.d * Dr;
        p.G = v.gram;
        p.n = zw * (going_left ? Y : X);
        p.rows = zw * v.C * (going_left ? X : Y);
        p.nspec = p.rows < p.n ? p.rows : p.n;
        p.K0 = p.nspec < v.chi_max ? p.nspec : v.chi_max;         // pair mode: v.chi_max is the doubled count too
        p.pair = zw == 2;
    }
    return p;
}

struct BtBufs {
    double* A;       // [ncap][ncap] working copy of G, full symmetric storage
    double* Y;       // [2][ncap] y of the column just processed, indexed by the parity of the step
    double* Vall;    // [ncap][ncap] row j = reflector j
    double* dd;      // [ncap] diagonal of T
    double* ee;      // [ncap] off-diagonal of T
    double* tau;     // [ncap]
    double* Z;       // [CAP_LIMIT][ncap] eigenvectors of G (row k = vector k)
    double* lam;     // [CAP_LIMIT]
    double* res;     // [CAP_LIMIT]
    double* D;       // [CAP_LIMIT][CAP_LIMIT] Z^T Z - I
    double* Tfac;    // [ncap/16][16][16] compact-WY factors of the reflector blocks (k_bt_larft)
    int32_t* flag;   // [1] 0 ok, 1 = verification failed (host falls back to the library)
    int32_t* ctl;    // [4] see k_bt_decide
    int32_t* sticky; // [1] set when any solve since the last reset ended with flag != 0 (sweeps that do not read the verdict per bond), or null
    double* tailG;   // [BT_TAIL][BT_TAIL] the trailing block handed to k_eig_tail (all earlier reflectors applied)
    int use_tail;    // 1: the last BT_TAIL steps run in k_eig_tail
    int ncap;
    int cnative;     // 1: pair mode through the NATIVE Hermitian reduction (k_bt_coop_c): dd / ee describe the real tridiagonal matrix of
                     // order n/2, Vall / tau hold complex reflectors (double2, row stride ncap/2), k_bt_back_c back-transforms
    const int32_t* abort;   // persistent tridiagonalisation's abort word (null on the launch-per-step path): when it is set,
                            // dd / ee / Vall are stale or partial and every kernel after it must leave without publishing
    // subspace solver in front of the exact one (mpst_eig_subspace.inl):
    const int32_t* skip;    // exact phase: word that says "the subspace result stands" - every kernel of the exact solve leaves at once
    const int32_t* gate;    // subspace phase: word that says "this bond is attempted" - 0: the shared verification kernels leave at once
    int ss;                 // 1: k_bt_gram / k_bt_decide / k_bt_polish run for the subspace phase (Z, lam, res, dd filled by k_ss_*)
};
__device__ __forceinline__ bool bt_aborted(const BtBufs& b) { return b.abort && *(const volatile int32_t*)b.abort != 0; }
// the launch has nothing to do: the subspace result stands (exact phase), or the bond is not attempted (subspace phase)
__device__ __forceinline__ bool bt_idle(const BtBufs& b) {
    return (b.skip && *(const volatile int32_t*)b.skip != 0) || (b.gate && *(const volatile int32_t*)b.gate == 0);
}

// the sum a 512-thread workgroup (8 waves, added in wave order) forms, from 256 threads that each carry the partial sums of
// two of its threads (t and t + 256)
__device__ __forceinline__ double bt_block_sum512(double sA, double sB, double* red8) {
    sA = wave_sum_fast(sA);
    sB = wave_sum_fast(sB);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        red8[threadIdx.x >> 6] = sA;
        red8[4 + (threadIdx.x >> 6)] = sB;
    }
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < 8; ++w) s += red8[w];
    return s;
}
__device__ __forceinline__ double bt_block_sum(double x, double* red) {
    x = wave_sum(x);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(BT_T) void k_bt_prep(View v, int lid, int going_left, const double* rawG, int rawn, BtBufs b) {
    if (bt_idle(b)) return;
    const BtProblem pb = bt_resolve(v, lid, going_left, rawG, rawn);
    const int n = pb.n, ld = b.ncap;
    for (int64_t i = (int64_t)blockIdx.x * BT_T + threadIdx.x; i < (int64_t)n * n; i += (int64_t)gridDim.x * BT_T) {
        const int r = (int)(i / n), c = (int)(i - (int64_t)r * n);
        b.A[(int64_t)r * ld + c] = pb.G[i];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *b.flag = 0;
        if (n == 1) {
            b.dd[0] = pb.G[0];
            b.ee[0] = 0.0;
        }
    }
}

// beta, tau and 1/(alpha - beta) of dlarfg from alpha = a0 and the squared norm s of the rest of the column, for both the
// launch-per-step and the persistent kernels.  The chain sits on every step's critical path in
// every workgroup, so it is the branch-free form of k_eig_tri: v_rsq_f64 + two Heron steps instead of the IEEE square
// root, hardware reciprocal seed + Newton steps (frcp) instead of the two IEEE divisions.  A column whose tail is zero
// or below 1e-140 in norm is treated as already reduced (tau = 0), as dlarfg's s == 0 branch.
__device__ __forceinline__ void bt_reflector_scalars(double a0, double s, double& beta, double& tau, double& scale) {
    const double xx = fma(a0, a0, s);
    const bool nz = s > 0.0 && xx > 1e-280;
    const double rs = __builtin_amdgcn_rsq(nz ? xx : 1.0);
    double nrm = xx * rs;
    const double hrs = 0.5 * rs;
    nrm = fma(fma(-nrm, nrm, xx), hrs, nrm);
    nrm = fma(fma(-nrm, nrm, xx), hrs, nrm);
    const double bneg = copysign(nrm, a0);                 // -beta
    const double ib = frcp(bneg), is = frcp(a0 + bneg);
    beta = nz ? -bneg : a0;
    tau = nz ? (bneg + a0) * ib : 0.0;                     // (beta - alpha) / beta
    scale = nz ? is : 0.0;                                 // 1 / (alpha - beta)
}

// One Householder step per launch (dsytd2 with the rank-2 update of step j-1 folded into step j's pass over the
// trailing matrix, so the matrix is read and written once per step and the only global reductions are two block sums
// that every workgroup repeats for itself):
//   (a) every workgroup: alpha_{j-1} = -tau/2 y^T v, w = y + alpha v of the previous step (y, v complete in memory);
//   (b) every workgroup: row j of the matrix with that pending update applied, d_j, the reflector v_j, tau_j, e_j;
//   (c) its own rows r > j: A[r][:] -= v_prev[r] w^T + w[r] v_prev^T (full rows: both triangles stay consistent, and a
//       row's product with v_j needs nothing from other workgroups), then y_r = tau_j A[r][:] v_j.
__global__ __launch_bounds__(BT_T) void k_bt_step(View v, int lid, int going_left, int rawn, BtBufs b, int j) {
    if (bt_idle(b)) return;
    __shared__ double xs[BT_NMAX];      // row j with the pending update applied, then the reflector v_j
    __shared__ double vl[BT_NMAX];      // v_{j-1}
    __shared__ double wl[BT_NMAX];      // w_{j-1}
    __shared__ double red[8];
    const BtProblem pb = bt_resolve(v, lid, going_left, nullptr, rawn);
    const int n = pb.n, ld = b.ncap, tid = threadIdx.x;
    if (j > n - 2 || j >= bt_tail_start(n, b.use_tail)) return;
    // (a)
    if (j > 0) {
        const double* vprev = b.Vall + (int64_t)(j - 1) * ld;
        // y is double-buffered by the parity of the step: a fast workgroup writes this step's y in (c) while a slow one is
        // still reading the previous step's here
        const double* yprev = b.Y + (int64_t)((j - 1) & 1) * ld;
        // the two block sums of a step are formed exactly as the persistent kernel forms them with 512 threads (thread t
        // there owns the entries j + t + 512 q; thread t here plays t and t + 256): the paths then agree to the bit, and a
        // bond that falls back to this path mid-sweep does not perturb the result
        double sA = 0.0, sB = 0.0;
        for (int r = j + tid; r < n; r += BT_T) vl[r] = vprev[r];
        for (int r = j + tid; r < n; r += 2 * BT_T) sA = fma(yprev[r], vprev[r], sA);
        for (int r = j + tid + BT_T; r < n; r += 2 * BT_T) sB = fma(yprev[r], vprev[r], sB);
        const double s = bt_block_sum512(sA, sB, red);
        const double alpha = -0.5 * b.tau[j - 1] * s;
        for (int r = j + tid; r < n; r += BT_T) wl[r] = fma(alpha, vl[r], yprev[r]);
    } else {
        for (int r = tid; r < n; r += BT_T) vl[r] = wl[r] = 0.0;
    }
    __syncthreads();
    // (b) row j (= column j: the trailing matrix is kept symmetric in full storage)
    {
        const double vj = vl[j], wj = wl[j];
        const double* arow = b.A + (int64_t)j * ld;
        for (int c = j + tid; c < n; c += BT_T) xs[c] = fma(-wj, vl[c], fma(-vj, wl[c], arow[c]));
    }
    __syncthreads();
    const double dj = xs[j];
    double tau = 0.0, beta;
    {
        double sA = 0.0, sB = 0.0;
        for (int r = j + tid; r < n; r += 2 * BT_T)
            if (r >= j + 2) sA = fma(xs[r], xs[r], sA);
        for (int r = j + tid + BT_T; r < n; r += 2 * BT_T) sB = fma(xs[r], xs[r], sB);
        const double s = bt_block_sum512(sA, sB, red);
        const double a0 = xs[j + 1];
        beta = a0;
        double scale = 0.0;
        bt_reflector_scalars(a0, s, beta, tau, scale);
        __syncthreads();
        for (int r = j + tid; r < n; r += BT_T) {
            const double vr = r <= j ? 0.0 : (r == j + 1 ? 1.0 : xs[r] * scale);
            xs[r] = vr;
            if (blockIdx.x == 0) b.Vall[(int64_t)j * ld + r] = vr;
        }
        if (blockIdx.x == 0 && tid == 0) {
            b.tau[j] = tau;
            b.dd[j] = dj;
            b.ee[j] = beta;
            if (j == n - 2) {
                // the last diagonal entry, with the pending update of step n-3 applied
                b.dd[n - 1] = b.A[(int64_t)(n - 1) * ld + (n - 1)] - 2.0 * vl[n - 1] * wl[n - 1];
                b.ee[n - 1] = 0.0;
            }
        }
        __syncthreads();
    }
    if (j == n - 2) return;
    // (c) own rows: one wave per row, lanes along the row, 4 x 64 columns per round trip
    {
        const int wave = tid >> 6, lane = tid & 63;
        const int m = n - (j + 1);
        const int per = (m + gridDim.x - 1) / gridDim.x;
        const int r0 = j + 1 + blockIdx.x * per, r1 = min(n, r0 + per);
        for (int r = r0 + wave; r < r1; r += 4) {
            double* arow = b.A + (int64_t)r * ld;
            const double vr = vl[r], wr = wl[r];
            double s = 0.0;
            for (int c0 = j + 1 + lane; c0 < n; c0 += 64 * 4) {
                double a4[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) a4[q] = (c0 + 64 * q < n) ? arow[c0 + 64 * q] : 0.0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = c0 + 64 * q;
                    if (c < n) {
                        const double a = fma(-wr, vl[c], fma(-vr, wl[c], a4[q]));
                        arow[c] = a;
                        s = fma(a, xs[c], s);
                    }
                }
            }
            s = wave_sum(s);
            if (lane == 0) b.Y[(int64_t)(j & 1) * ld + r] = tau * s;
        }
    }
}

// launch-per-step path, before k_eig_tail: the trailing block with the pending update of step m-1 applied - the same
// alpha, w and element-wise update as step m would form (and as the persistent kernel forms at its hand-over), so both
// paths hand identical bits to the tail kernel
__global__ __launch_bounds__(BT_T) void k_bt_tail_prep(View v, int lid, int going_left, int rawn, BtBufs b) {
    if (bt_idle(b)) return;
    __shared__ double vl[BT_NMAX];
    __shared__ double wl[BT_NMAX];
    __shared__ double red[8];
    const BtProblem pb = bt_resolve(v, lid, going_left, nullptr, rawn);
    const int n = pb.n, ld = b.ncap, tid = threadIdx.x;
    const int j = bt_tail_start(n, b.use_tail);
    if (j > n - 2) return;
    {
        const double* vprev = b.Vall + (int64_t)(j - 1) * ld;
        const double* yprev = b.Y + (int64_t)((j - 1) & 1) * ld;
        double sA = 0.0, sB = 0.0;
        for (int r = j + tid; r < n; r += BT_T) vl[r] = vprev[r];
        for (int r = j + tid; r < n; r += 2 * BT_T) sA = fma(yprev[r], vprev[r], sA);
        for (int r = j + tid + BT_T; r < n; r += 2 * BT_T) sB = fma(yprev[r], vprev[r], sB);
        const double s = bt_block_sum512(sA, sB, red);
        const double alpha = -0.5 * b.tau[j - 1] * s;
        for (int r = j + tid; r < n; r += BT_T) wl[r] = fma(alpha, vl[r], yprev[r]);
    }
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63, nt = n - j;
    for (int r = j + (int)blockIdx.x * 4 + wave; r < n; r += (int)gridDim.x * 4) {
        const double* arow = b.A + (int64_t)r * ld;
        const double vr = vl[r], wr = wl[r];
        for (int c = j + lane; c < n; c += 64) b.tailG[(int64_t)(r - j) * nt + (c - j)] = fma(-wr, vl[c], fma(-vr, wl[c], arow[c]));
    }
}

// ---- the same tridiagonalisation as ONE persistent launch -----------------------------------------------------------------
// A Householder step per launch costs 6.4 us at n = 300-500: 1.5 us of launch and the rest because every
// step rewrites the trailing matrix through memory (the eight XCDs have separate L2s).  Here the matrix never leaves the
// chip's local memories: workgroup g keeps rows g, g+G, g+2G, ... in LDS for the whole factorisation and a step exchanges
// only two length-n vectors through memory - y (every owner writes its entries) and the row the next step starts from -
// with agent-scope loads / stores and one counting barrier (1.8 us per round across the XCDs, scratch/ubench/xcd_barrier.hip).
// Every workgroup still repeats the cheap part (alpha, w, the reflector) for itself as k_bt_step does - the same arithmetic
// in the same association (bt_block_sum512) - so all paths give identical bits (tests/probes/path_hash.py).  The
// launch-per-step path remains the fallback when a workgroup's wait runs out of patience (its peers not resident: other
// work holding the CUs).
struct BtCoop {
    double* ybuf;            // [2][ncap] y of a step, by parity
    double* rowbuf;          // [2][ncap] the published row, by parity
    unsigned int* counter;   // arrivals, monotonic within a solve
    int32_t* abort_flag;     // set by a workgroup that gave up waiting: everybody leaves, the host falls back
    unsigned long long* roll;  // roll call of the XCD-local variant: low byte = workgroups present, 7 bits per XCD above it
    int xsel;                // which workgroups of a strided launch take part (blockIdx % stride): the context's ordinal, so that
                             // concurrent fits on one GPU confine their solves to different XCDs
};
// Two variants of the exchange.  SC_AGENT: relaxed agent-scope atomics (sc1: served by memory, correct wherever the
// workgroups run) and a counting barrier.  SC_XCD: plain stores (the L1 writes through to the L2) and NON-TEMPORAL loads
// (never kept in the reader's L1, so always served by the L2): coherent exactly when every participant sits on the SAME
// XCD, which the kernel checks with a roll call (HW_REG_XCC_ID) before the first such operation - it gives up
// otherwise.  scratch/ubench/xcd_exchange.hip: 1.33 us per counted round of 32 workgroups against 2.1-2.4 us with agent
// scope (and what does NOT work: sc0 loads with 32 workgroups, with or without buffer_inv sc0).  On top of that the
// XCD-local variant drops the counter: its entries validate themselves (below).
constexpr int SC_AGENT = __HIP_MEMORY_SCOPE_AGENT, SC_XCD = __HIP_MEMORY_SCOPE_WORKGROUP;
constexpr int ABORT_PATIENCE = 1, ABORT_PLACEMENT = 2;
template <int SC> __device__ __forceinline__ void st_sc(double* p, double x) { __hip_atomic_store(p, x, __ATOMIC_RELAXED, SC); }
// ---- XCD-local exchange: self-validating entries, no counter ----------------------------------------------------------
// An entry is 16 bytes {value, key ^ bits(value)}; key = (solve sequence number, kind, step).  The producer just stores
// it - no s_waitcnt, no barrier, no atomic on its side - and every consumer polls the entries it needs until they carry
// the key of the step: the poll IS the load.  A torn or stale read cannot validate (an old value under a new tag word
// fails the xor unless the value is the same anyway), so nothing depends on the two halves arriving together.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned long long tag_key(unsigned int seq, int kind, int step) {
    return ((unsigned long long)seq << 32) | ((unsigned long long)kind << 20) | (unsigned long long)(step + 1);
}
__device__ __forceinline__ void st_tag(double* base, int idx, double x, unsigned long long key) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(x), t = key ^ bits;
    const u32x4 q = {(unsigned int)bits, (unsigned int)(bits >> 32), (unsigned int)t, (unsigned int)(t >> 32)};
    // s_nop: a VMEM store of more than 64 bits reads its data registers over two cycles and the next VALU write of those registers
    // needs a wait state (CDNA3 ISA, data hazards).  The compiler's hazard recogniser does not look into inline asm: two of these
    // back to back (the complex reduction stores re and im) reused the registers of q and tore the first entry's tag.
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(base + 2 * idx), "v"(q) : "memory");
}
__device__ __forceinline__ bool tag_ok(const u32x4 q, unsigned long long key, double& x) {
    const unsigned long long bits = (unsigned long long)q.x | ((unsigned long long)q.y << 32);
    const unsigned long long t = (unsigned long long)q.z | ((unsigned long long)q.w << 32);
    x = __longlong_as_double((long long)bits);
    return (t ^ bits) == key;
}
// the five entries a step needs from the others, requested together with non-temporal loads; true if all carry their key
__device__ __forceinline__ bool ld5_tag(const double* p0, const double* p1, const double* p2, const double* p3, const double* p4,
                                        unsigned long long k0, unsigned long long k1, unsigned long long k2, unsigned long long k3,
                                        unsigned long long k4, double& v0, double& v1, double& v2, double& v3, double& v4) {
    u32x4 q0, q1, q2, q3, q4;
    asm volatile("global_load_dwordx4 %0, %5, off nt\n\t"
                 "global_load_dwordx4 %1, %6, off nt\n\t"
                 "global_load_dwordx4 %2, %7, off nt\n\t"
                 "global_load_dwordx4 %3, %8, off nt\n\t"
                 "global_load_dwordx4 %4, %9, off nt\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "=&v"(q4)
                 : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4)
                 : "memory");
    bool ok = tag_ok(q0, k0, v0);
    ok = tag_ok(q1, k1, v1) && ok;
    ok = tag_ok(q2, k2, v2) && ok;
    ok = tag_ok(q3, k3, v3) && ok;
    ok = tag_ok(q4, k4, v4) && ok;
    return ok;
}
// six entries (three complex numbers) of the native Hermitian reduction, same protocol
__device__ __forceinline__ bool ld6_tag(const double* p0, const double* p1, const double* p2, const double* p3, const double* p4, const double* p5,
                                        unsigned long long ka, unsigned long long kb, unsigned long long kc, double2& v0, double2& v1, double2& v2) {
    u32x4 q0, q1, q2, q3, q4, q5;
    asm volatile("global_load_dwordx4 %0, %6, off nt\n\t"
                 "global_load_dwordx4 %1, %7, off nt\n\t"
                 "global_load_dwordx4 %2, %8, off nt\n\t"
                 "global_load_dwordx4 %3, %9, off nt\n\t"
                 "global_load_dwordx4 %4, %10, off nt\n\t"
                 "global_load_dwordx4 %5, %11, off nt\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "=&v"(q4), "=&v"(q5)
                 : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(p5)
                 : "memory");
    bool ok = tag_ok(q0, ka, v0.x);
    ok = tag_ok(q1, ka, v0.y) && ok;
    ok = tag_ok(q2, kb, v1.x) && ok;
    ok = tag_ok(q3, kb, v1.y) && ok;
    ok = tag_ok(q4, kc, v2.x) && ok;
    ok = tag_ok(q5, kc, v2.y) && ok;
    return ok;
}
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xF; }   // HW_REG_XCC_ID[3:0]
// every workgroup has arrived `target` times in total; false if the wait was abandoned
template <int SC>
__device__ __forceinline__ bool bt_coop_wait(const BtCoop& cp, unsigned int target, int* sh_ok) {
    if (threadIdx.x == 0) {
        int ok = 1;
        int spins = 0;
        // relaxed polling: everything exchanged between workgroups is itself read and written with operations of the
        // same scope, so no cache needs invalidating when the count is reached
        while (__hip_atomic_load(cp.counter, __ATOMIC_RELAXED, SC) < target) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 1023) == 0 &&
                (spins > (1 << 21) || __hip_atomic_load(cp.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                __hip_atomic_fetch_or(cp.abort_flag, ABORT_PATIENCE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
        }
        *sh_ok = ok;
    }
    __syncthreads();
    return *sh_ok != 0;
}
template <int SC>
__device__ __forceinline__ void bt_coop_arrive(const BtCoop& cp) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's (write-through) stores of the step have left the CU
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(cp.counter, 1u, __ATOMIC_RELAXED, SC);
}
// XCD-local variant only: everybody announces itself and its XCD (agent scope, one word), waits for the G participants
// and checks that exactly one XCD field is populated
__device__ __forceinline__ bool bt_coop_roll_call(const BtCoop& cp, int G, int* sh_ok) {
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(cp.roll, 1ull | (1ull << (8 + 7 * (xcc_id() & 7))), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int ok = 1, spins = 0;
        unsigned long long r;
        while (((r = __hip_atomic_load(cp.roll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & 0xffull) < (unsigned long long)G) {
            __builtin_amdgcn_s_sleep(1);
            // (about 30 ms: peers that are not resident by then are waiting for CUs other work holds)
            if ((++spins & 1023) == 0 &&
                (spins > (1 << 15) || __hip_atomic_load(cp.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                __hip_atomic_fetch_or(cp.abort_flag, ABORT_PATIENCE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
        }
        if (ok) {
            int fields = 0;
            for (int x = 0; x < 8; ++x) fields += ((r >> (8 + 7 * x)) & 0x7full) != 0;
            if (fields != 1) {                    // every participant reads the same final word: all leave together
                __hip_atomic_fetch_or(cp.abort_flag, ABORT_PLACEMENT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
            }
        }
        *sh_ok = ok;
    }
    __syncthreads();
    return *sh_ok != 0;
}

// stride: only the workgroups blockIdx.x = cp.xsel mod stride take part (the dispatcher deals consecutive workgroups round-robin
// over the 8 XCDs, so stride 8 puts the participants on one XCD - the roll call verifies it); the others leave at once.
template <int NT, int SC>
__global__ __launch_bounds__(NT) void k_bt_coop(View v, int lid, int going_left, BtBufs b, BtCoop cp, int stride, unsigned int seq) {
    if (bt_idle(b)) return;
    if ((int)blockIdx.x % stride != cp.xsel % stride) return;
    constexpr int NW = NT / 64;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double red_a[NW], red_b[NW], bc[2];
    __shared__ int sh_ok;
    constexpr int QV = BT_NMAX / NT;
    const BtProblem pb = bt_resolve(v, lid, going_left, nullptr, 0);
    const int n = pb.n, ld = b.ncap, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int G = ((int)gridDim.x + stride - 1) / stride, g = (int)blockIdx.x / stride;
    double* xs = smem;                   // [ncap] the reflector v_j (v_{j-1} when a step starts)
    double* vl = xs + ld;                // [ncap] v_{j-1}
    double* wl = vl + ld;                // [ncap] w_{j-1}
    double* rows = wl + ld;              // [ceil(ncap / G)][ncap] this workgroup's rows g, g+G, ...
    const int nown = g < n ? (n - 1 - g) / G + 1 : 0;
    // sweeps that read the verdict once at their end (launch_eig_blocked_nosync): a solve that failed earlier in the sweep has
    // condemned it - the sweep is redone bond by bond anyway - so later bonds neither spin for their peers nor compute on the
    // unspecified state: every workgroup sees the same sticky word (written by an earlier kernel of the stream) and leaves
    if (b.sticky && *(const volatile int32_t*)b.sticky != 0) {
        if (blockIdx.x == 0 && tid == 0) *cp.abort_flag = ABORT_PATIENCE;
        return;
    }
    if (g == 0 && tid == 0) *b.flag = 0;
    for (int k = 0; k < nown; ++k) {
        const int r = g + k * G;
        for (int c = tid; c < n; c += NT) rows[k * ld + c] = pb.G[(int64_t)r * n + c];
    }
    for (int c = tid; c < n; c += NT) xs[c] = vl[c] = wl[c] = 0.0;
    __syncthreads();
    if (n == 1) {
        if (g == 0 && tid == 0) {
            b.dd[0] = pb.G[0];
            b.ee[0] = 0.0;
        }
        return;
    }
    constexpr bool TAGGED = SC == SC_XCD;
    if (TAGGED && !bt_coop_roll_call(cp, G, &sh_ok)) return;          // leaves sh_ok = 1
    if (g == 0) {
        for (int c = tid; c < n; c += NT) {
            if constexpr (TAGGED) st_tag(cp.rowbuf, c, rows[c], tag_key(seq, 2, 0));
            else st_sc<SC>(cp.rowbuf + c, rows[c]);
        }
    }
    if constexpr (!TAGGED) bt_coop_arrive<SC>(cp);
    double tau_prev = 0.0;
#ifdef MPST_COOP_PROF
    unsigned long long pt[6] = {0, 0, 0, 0, 0, 0}, t0 = __builtin_amdgcn_s_memrealtime(), t1;
#define PSTAMP(i) do { t1 = __builtin_amdgcn_s_memrealtime(); pt[i] += t1 - t0; t0 = t1; } while (0)
#else
#define PSTAMP(i) do { } while (0)
#endif
    for (int j = 0; j <= n - 2; ++j) {
        PSTAMP(5);
        double vp[QV], yp[QV], aj[QV], yj;
        if constexpr (TAGGED) {
            static_assert(QV == 2 || !TAGGED, "the XCD-local exchange is written for 512 threads");
            // entries are 16 bytes; y of step j-1 sits in the buffer of parity (j+1)&1, the published row j in j&1
            const double* yprev = cp.ybuf + (int64_t)((j + 1) & 1) * 2 * ld;
            const double* rowj = cp.rowbuf + (int64_t)(j & 1) * 2 * ld;
            const unsigned long long kr = tag_key(seq, 2, j), ky = j > 0 ? tag_key(seq, 1, j - 1) : kr;
            const double* ysrc = j > 0 ? yprev : rowj;            // step 0 has no y: poll the row entries twice
            const int i0 = j + tid, i1 = j + tid + NT;
            const int c0 = i0 < n ? i0 : j, c1 = i1 < n ? i1 : j;        // out-of-range lanes poll entry j and drop it
            int spins = 0;
            while (!ld5_tag(ysrc + 2 * c0, rowj + 2 * c0, ysrc + 2 * c1, rowj + 2 * c1, ysrc + 2 * j, ky, kr, ky, kr, ky,
                            yp[0], aj[0], yp[1], aj[1], yj)) {
                // (no s_sleep: a poll is an L2 round trip, there is nothing else for the wave to do)
                if ((++spins & 255) == 0 &&
                    (spins > (1 << 19) || __hip_atomic_load(cp.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                    __hip_atomic_fetch_or(cp.abort_flag, ABORT_PATIENCE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    sh_ok = 0;                                     // seen by everybody after the next barrier
                    break;
                }
            }
            yp[0] = (i0 < n && j > 0) ? yp[0] : 0.0;
            yp[1] = (i1 < n && j > 0) ? yp[1] : 0.0;
            aj[0] = i0 < n ? aj[0] : 0.0;
            aj[1] = i1 < n ? aj[1] : 0.0;
            yj = j > 0 ? yj : 0.0;
            vp[0] = (i0 < n && j > 0) ? xs[i0] : 0.0;
            vp[1] = (i1 < n && j > 0) ? xs[i1] : 0.0;
        } else {
            if (!bt_coop_wait<SC>(cp, (unsigned int)G * (unsigned int)(j + 1), &sh_ok)) return;
            PSTAMP(0);
            // everything this step needs from the others, requested at once
            const double* yprev = cp.ybuf + (int64_t)((j + 1) & 1) * ld;
            const double* rowj = cp.rowbuf + (int64_t)(j & 1) * ld;
#pragma unroll
            for (int q = 0; q < QV; ++q) {
                const int idx = j + tid + NT * q;
                const bool ok = idx < n;
                yp[q] = (ok && j > 0) ? __hip_atomic_load(yprev + idx, __ATOMIC_RELAXED, SC) : 0.0;
                aj[q] = ok ? __hip_atomic_load(rowj + idx, __ATOMIC_RELAXED, SC) : 0.0;
                vp[q] = (ok && j > 0) ? xs[idx] : 0.0;
            }
            yj = j > 0 ? __hip_atomic_load(yprev + j, __ATOMIC_RELAXED, SC) : 0.0;
        }
        PSTAMP(1);
        // (a) alpha_{j-1}
        double alpha;
        {
            double s = 0.0;
#pragma unroll
            for (int q = 0; q < QV; ++q) s = fma(yp[q], vp[q], s);
            s = wave_sum_fast(s);
            if (lane == 0) red_a[wave] = s;
            __syncthreads();
            if (TAGGED && sh_ok == 0) return;          // somebody's poll ran out of patience: all leave together
            double sa = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) sa += red_a[w];
            alpha = -0.5 * tau_prev * sa;
        }
        PSTAMP(2);
        // (b) row j with the pending update applied (v_{j-1}[j] is the leading one of reflector j-1, w_{j-1}[j] = y_j + alpha)
        const double vj = j > 0 ? 1.0 : 0.0, wj = j > 0 ? yj + alpha : 0.0;
        double xr[QV];
        double tau = 0.0, beta, dj;
        {
            double s = 0.0;
#pragma unroll
            for (int q = 0; q < QV; ++q) {
                const int idx = j + tid + NT * q;
                yp[q] = j > 0 ? fma(alpha, vp[q], yp[q]) : 0.0;        // w_{j-1} from here on
                xr[q] = fma(-wj, vp[q], fma(-vj, yp[q], aj[q]));
                if (idx >= j + 2 && idx < n) s = fma(xr[q], xr[q], s);
                if (idx < n) {
                    vl[idx] = vp[q];
                    wl[idx] = yp[q];
                }
            }
            if (tid == 0) bc[0] = xr[0];               // d_j
            if (tid == 1) bc[1] = xr[0];               // the leading entry of the column below the diagonal
            s = wave_sum_fast(s);
            if (lane == 0) red_b[wave] = s;
            __syncthreads();
            if (j == bt_tail_start(n, b.use_tail)) {
                // hand-over to k_eig_tail: v_{j-1}, w_{j-1} are complete in LDS; every workgroup applies that pending update to
                // its rows of the trailing block and writes them out - the reduction of the last BT_TAIL columns runs on
                // one CU from here (0.9 us per step instead of this kernel's 2.6)
                const int nt = n - j;
                for (int k = wave; k < nown; k += NW) {
                    const int r = g + k * G;
                    if (r < j) continue;
                    const double* arow = rows + k * ld;
                    const double vr = vl[r], wr = wl[r];
                    for (int c = j + lane; c < n; c += 64) b.tailG[(int64_t)(r - j) * nt + (c - j)] = fma(-wr, vl[c], fma(-vr, wl[c], arow[c]));
                }
                return;
            }
            s = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) s += red_b[w];
            const double a0 = bc[1];
            dj = bc[0];
            beta = a0;
            double scale = 0.0;
            bt_reflector_scalars(a0, s, beta, tau, scale);
#pragma unroll
            for (int q = 0; q < QV; ++q) {
                const int idx = j + tid + NT * q;
                if (idx < n) xs[idx] = idx <= j ? 0.0 : (idx == j + 1 ? 1.0 : xr[q] * scale);
            }
            __syncthreads();
        }
        // what only the later kernels read (the stored reflector, tau, d_j, e_j) is written off the critical path, after the
        // step's exchange has been issued, every workgroup a share
        auto bookkeeping = [&]() {
            const int per = (n + G - 1) / G, idx = g * per + tid;      // every workgroup a contiguous share of the reflector
            if (tid < per && idx >= j && idx < n) b.Vall[(int64_t)j * ld + idx] = xs[idx];
            if (g == j % G && tid == 0) {
                b.tau[j] = tau;
                b.dd[j] = dj;
                b.ee[j] = beta;
            }
        };
        tau_prev = tau;
        PSTAMP(3);
        if (j == n - 2) {
            bookkeeping();
            // the last diagonal entry, with the pending update of step n-3 applied, from the row's owner
            if ((n - 1) % G == g && tid == 0) {
                const int k = (n - 1 - g) / G;
                b.dd[n - 1] = rows[k * ld + (n - 1)] - 2.0 * vl[n - 1] * wl[n - 1];
                b.ee[n - 1] = 0.0;
            }
            break;
        }
        // (c) own rows r > j: pending update, product with the new reflector; the owner of row j+1 publishes it
        double* ynew = cp.ybuf + (int64_t)(j & 1) * (TAGGED ? 2 : 1) * ld;
        double* rownext = cp.rowbuf + (int64_t)((j + 1) & 1) * (TAGGED ? 2 : 1) * ld;
        const unsigned long long key_y = tag_key(seq, 1, j), key_r = tag_key(seq, 2, j + 1);
        auto put_row = [&](int c, double x) {
            if constexpr (TAGGED) st_tag(rownext, c, x, key_r);
            else st_sc<SC>(rownext + c, x);
        };
        auto put_y = [&](int r, double x) {
            if constexpr (TAGGED) st_tag(ynew, r, x, key_y);
            else st_sc<SC>(ynew + r, x);
        };
        // two rows per pass (k and k + NW): they share the LDS reads of v, w and the new reflector, and their two
        // reductions overlap; per row the arithmetic is that of the one-row loop
        for (int k = wave; k < nown; k += 2 * NW) {
            const int k2 = k + NW;
            const int r1 = g + k * G, r2 = g + k2 * G;
            const bool live1 = r1 > j, live2 = k2 < nown && r2 > j;
            if (!live2) {
                if (!live1) continue;
                double* arow = rows + k * ld;
                const double vr = vl[r1], wr = wl[r1];
                const bool pub = r1 == j + 1;
                double s = 0.0;
                for (int c = j + 1 + lane; c < n; c += 64) {
                    const double a_ = fma(-wr, vl[c], fma(-vr, wl[c], arow[c]));
                    arow[c] = a_;
                    s = fma(a_, xs[c], s);
                    if (pub) put_row(c, a_);
                }
                s = wave_sum_fast(s);
                if (lane == 0) put_y(r1, tau * s);
                continue;
            }
            double* arow1 = rows + k * ld;
            double* arow2 = rows + k2 * ld;
            const double vr1 = vl[r1], wr1 = wl[r1], vr2 = vl[r2], wr2 = wl[r2];
            const bool pub1 = live1 && r1 == j + 1, pub2 = r2 == j + 1;
            double s1 = 0.0, s2 = 0.0;
            // 4 x 64 columns per pass, every LDS operand requested before the first use: the loop is bound by the
            // LDS latency of a pass, not by its bytes (per lane the columns are still accumulated in ascending order)
            for (int cb = j + 1 + lane; cb < n + lane; cb += 256) {
                double vc[4], wc[4], xc[4], r1v[4], r2v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int c = cb + 64 * u, cc = c < n ? c : n - 1;
                    vc[u] = vl[cc];
                    wc[u] = wl[cc];
                    xc[u] = xs[cc];
                    r1v[u] = arow1[cc];
                    r2v[u] = arow2[cc];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int c = cb + 64 * u;
                    if (c < n) {
                        if (live1) {
                            const double a1 = fma(-wr1, vc[u], fma(-vr1, wc[u], r1v[u]));
                            arow1[c] = a1;
                            s1 = fma(a1, xc[u], s1);
                            if (pub1) put_row(c, a1);
                        }
                        const double a2 = fma(-wr2, vc[u], fma(-vr2, wc[u], r2v[u]));
                        arow2[c] = a2;
                        s2 = fma(a2, xc[u], s2);
                        if (pub2) put_row(c, a2);
                    }
                }
            }
            PSTAMP(0);
            s1 = wave_sum_fast(s1);
            s2 = wave_sum_fast(s2);
            if (lane == 0) {
                if (live1) put_y(r1, tau * s1);
                put_y(r2, tau * s2);
            }
        }
        PSTAMP(4);
        if constexpr (!TAGGED) bt_coop_arrive<SC>(cp);
        bookkeeping();
    }
#ifdef MPST_COOP_PROF
    if (g == 0 && tid == 0) {
        unsigned long long* st = v.sc->eig_stamps;
        st[0] = 0; st[1] = pt[0]; st[2] = 0; st[3] = pt[1]; st[4] = pt[1] + pt[2]; st[5] = pt[1] + pt[2] + pt[3]; st[8] = 0; st[9] = pt[4]; st[6] = 0; st[7] = pt[5];
    }
#endif
}

// ---- pair mode, native: Hermitian tridiagonalisation with complex reflectors ----------------------------------------------
// What the embedding costs: the 2n x 2n real problem takes 2n - 2 dependent Householder steps, each exchanging 2n-vectors.
// The Hermitian matrix G = Gr + i Gi itself (read out of the embedding k_tgram wrote: Gr = M[0:n, 0:n], Gi = M[n:2n, 0:n]) is
// reduced here by n - 1 steps of zhetd2 (LAPACK, UPLO = 'L') in the data flow of k_bt_coop - workgroup g keeps rows g, g+G, ...
// in LDS for the whole factorisation, the rank-2 update of step j-1 is applied during step j, a step exchanges y and the next
// row - with complex arithmetic in these places (conjugates as in the NumPy model the kernel was transcribed from):
//   alpha_{j-1} = -tau/2 (y^H v),  w = y + alpha v;   column j = conj(row j) - v conj(w_j) - w conj(v_j)
//   zlarfg: beta = -sign(Re a0) sqrt(|a0|^2 + |x|^2) REAL, tau = (beta - a0) / beta COMPLEX, v = x / (a0 - beta)
//   A[r][c] -= v_r conj(w_c) + w_r conj(v_c),   y_r = tau sum_c A[r][c] v_c
// T is real symmetric tridiagonal of order n with a SIMPLE spectrum: k_bt_vec works on it as on any real problem (no pairs),
// k_bt_back_c applies Q = H_0 ... H_{n-2} (the last reflector is a pure phase, tau != 0) and writes the eigenvector of the
// embedding (Re u, Im u) with its partner J u next to it, so k_bt_gram / k_bt_decide / k_bt_polish run unchanged.
__device__ __forceinline__ double2 c_mul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ double2 c_mulc(double2 a, double2 b) { return make_double2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }   // a conj(b)
__device__ __forceinline__ double2 ld_c(const double2* p) {
    return make_double2(__hip_atomic_load(&p->x, __ATOMIC_RELAXED, SC_AGENT), __hip_atomic_load(&p->y, __ATOMIC_RELAXED, SC_AGENT));
}
__device__ __forceinline__ void st_c(double2* p, double2 v) {
    __hip_atomic_store(&p->x, v.x, __ATOMIC_RELAXED, SC_AGENT);
    __hip_atomic_store(&p->y, v.y, __ATOMIC_RELAXED, SC_AGENT);
}
// the same two through ONE XCD's L2 (all participants on that XCD - roll call): plain 128-bit store (the L1 writes through),
// non-temporal load (never kept in the reader's L1, so always served by the L2); a line of these buffers is never touched any other way
__device__ __forceinline__ double2 ld_x(const double2* p) {
    u32x4 q;
    asm volatile("global_load_dwordx4 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=&v"(q) : "v"(p) : "memory");
    return make_double2(__longlong_as_double((long long)((unsigned long long)q.x | ((unsigned long long)q.y << 32))),
                        __longlong_as_double((long long)((unsigned long long)q.z | ((unsigned long long)q.w << 32))));
}
__device__ __forceinline__ void ld_x3(const double2* p0, const double2* p1, const double2* p2, double2& v0, double2& v1, double2& v2) {
    u32x4 q0, q1, q2;
    asm volatile("global_load_dwordx4 %0, %3, off nt\n\t"
                 "global_load_dwordx4 %1, %4, off nt\n\t"
                 "global_load_dwordx4 %2, %5, off nt\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(q0), "=&v"(q1), "=&v"(q2)
                 : "v"(p0), "v"(p1), "v"(p2)
                 : "memory");
    auto cv = [](const u32x4 q) {
        return make_double2(__longlong_as_double((long long)((unsigned long long)q.x | ((unsigned long long)q.y << 32))),
                            __longlong_as_double((long long)((unsigned long long)q.z | ((unsigned long long)q.w << 32))));
    };
    v0 = cv(q0);
    v1 = cv(q1);
    v2 = cv(q2);
}
__device__ __forceinline__ void st_x(double2* p, double2 v) {
    const unsigned long long a = (unsigned long long)__double_as_longlong(v.x), b = (unsigned long long)__double_as_longlong(v.y);
    const u32x4 q = {(unsigned int)a, (unsigned int)(a >> 32), (unsigned int)b, (unsigned int)(b >> 32)};
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(q) : "memory");
}
// counting barrier inside one XCD: the arrivals are L2 atomics (no scope bits), the poll a non-temporal load
__device__ __forceinline__ bool xcd_wait(const BtCoop& cp, unsigned int target, int* sh_ok) {
    if (threadIdx.x == 0) {
        int ok = 1, spins = 0;
        for (;;) {
            unsigned int c;
            asm volatile("global_load_dword %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=&v"(c) : "v"(cp.counter) : "memory");
            if (c >= target) break;
            if ((++spins & 1023) == 0 &&
                (spins > (1 << 21) || __hip_atomic_load(cp.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                __hip_atomic_fetch_or(cp.abort_flag, ABORT_PATIENCE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
        }
        *sh_ok = ok;
    }
    __syncthreads();
    return *sh_ok != 0;
}
__device__ __forceinline__ void xcd_arrive(const BtCoop& cp) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's stores have been taken by the L2
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(cp.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
constexpr int BTC_NT = 512, BTC_NMAX = BT_NMAX / 2;
// TAGGED: the XCD-local exchange of k_bt_coop<512, SC_XCD> (self-validating 16-byte entries - two per complex number -, polls through
// the XCD's L2, no counter; every `stride`-th workgroup of the launch takes part, a roll call checks that they share an XCD), else
// agent-scope atomics and the counting barrier.
// XMODE 0: agent-scope atomics and the counting barrier (workgroups anywhere); 1: tagged entries inside one XCD; 3: the counting
// barrier and plain data inside one XCD (st_x / ld_x / xcd_wait / xcd_arrive)
template <int XMODE>
__global__ __launch_bounds__(BTC_NT) void k_bt_coop_c(View v, int lid, int going_left, BtBufs b, BtCoop cp, int stride, unsigned int seq,
                                                       const double* rawG, int rawn) {
    if (bt_idle(b)) return;
    if ((int)blockIdx.x % stride != cp.xsel % stride) return;
    constexpr int NT = BTC_NT, NW = NT / 64;
    constexpr bool TAGGED = XMODE == 1, XCNT = XMODE == 3;
    constexpr bool SOLO = XMODE == 4;         // ONE workgroup owns every row (order <= 96): y and the next row go through LDS, a step costs
                                              // barriers instead of two round trips through the L2 (the Rayleigh-Ritz problem of the subspace solver)
    static_assert(BTC_NMAX <= NT, "one element of every length-n vector per thread");
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double red_a[NW][2], red_b[NW];
    __shared__ double2 bc[2];
    __shared__ int sh_ok;
    const BtProblem pb = bt_resolve(v, lid, going_left, rawG, rawn);
    const int nf = pb.n, n = pb.n >> 1, ld = b.ncap >> 1, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int G = ((int)gridDim.x + stride - 1) / stride, g = (int)blockIdx.x / stride;
    double2* xs = (double2*)smem;        // [ld] the reflector v_j (v_{j-1} when a step starts)
    double2* vl = xs + ld;               // [ld] v_{j-1}
    double2* wl = vl + ld;               // [ld] w_{j-1}
    double2* rows = wl + ld;             // [ceil(ld / G)][ld] this workgroup's rows g, g+G, ...
    const int nown = g < n ? (n - 1 - g) / G + 1 : 0;
    double2* yl = rows + (size_t)(SOLO ? ld : 0) * ld;      // SOLO: [2][ld] y of the step before / at hand ...
    double2* rl = yl + 2 * ld;                               // ... and [2][ld] the row the next step starts from
    if (b.sticky && *(const volatile int32_t*)b.sticky != 0) {       // the sweep is condemned already (see k_bt_coop)
        if (blockIdx.x == 0 && tid == 0) *cp.abort_flag = ABORT_PATIENCE;
        return;
    }
    if (g == 0 && tid == 0) *b.flag = 0;
    for (int k = 0; k < nown; ++k) {
        const int r = g + k * G;
        for (int c = tid; c < n; c += NT) rows[k * ld + c] = make_double2(pb.G[(int64_t)r * nf + c], pb.G[(int64_t)(n + r) * nf + c]);
    }
    for (int c = tid; c < ld; c += NT) xs[c] = vl[c] = wl[c] = make_double2(0.0, 0.0);
    __syncthreads();
    double2* Vc = (double2*)b.Vall;
    double2* tauc = (double2*)b.tau;
    if (n == 1) {
        if (g == 0 && tid == 0) {
            b.dd[0] = pb.G[0];
            b.ee[0] = 0.0;
        }
        return;
    }
    double2* ybuf = (double2*)cp.ybuf;
    double2* rowbuf = (double2*)cp.rowbuf;
    // TAGGED: a parity buffer holds 2 ld entries of 16 bytes (re, im of element e at entries 2e, 2e + 1) = 4 ld doubles
    auto put_c = [&](double* base, int e, double2 x, unsigned long long key) {
        st_tag(base, 2 * e, x.x, key);
        st_tag(base, 2 * e + 1, x.y, key);
    };
    if ((TAGGED || XCNT) && !bt_coop_roll_call(cp, G, &sh_ok)) return;          // leaves sh_ok = 1
    if (g == 0) {
        for (int c = tid; c < n; c += NT) {
            if constexpr (SOLO) rl[c] = rows[c];
            else if constexpr (TAGGED) put_c(cp.rowbuf, c, rows[c], tag_key(seq, 2, 0));
            else if constexpr (XCNT) st_x(rowbuf + c, rows[c]);
            else st_c(rowbuf + c, rows[c]);
        }
    }
    if constexpr (SOLO) __syncthreads();
    else if constexpr (XCNT) xcd_arrive(cp);
    else if constexpr (!TAGGED) bt_coop_arrive<SC_AGENT>(cp);
    double2 tau_prev = make_double2(0.0, 0.0);
    for (int j = 0; j <= n - 2; ++j) {
        const int idx = j + tid;
        const bool ok = idx < n;
        const double2 zero = make_double2(0.0, 0.0);
        double2 yp, aj, yj;
        if constexpr (SOLO) {
            const double2* yprev = yl + ((j + 1) & 1) * ld;
            const double2* rowj = rl + (j & 1) * ld;
            yp = (ok && j > 0) ? yprev[idx] : zero;
            aj = ok ? rowj[idx] : zero;
            yj = j > 0 ? yprev[j] : zero;
        } else if constexpr (TAGGED) {
            const double* yprev = cp.ybuf + (int64_t)((j + 1) & 1) * 4 * ld;
            const double* rowj = cp.rowbuf + (int64_t)(j & 1) * 4 * ld;
            const unsigned long long kr = tag_key(seq, 2, j), ky = j > 0 ? tag_key(seq, 1, j - 1) : kr;
            const double* ysrc = j > 0 ? yprev : rowj;            // step 0 has no y: poll the row entries twice
            const int c0 = ok ? idx : j;                           // out-of-range lanes poll entry j and drop it
            int spins = 0;
            while (!ld6_tag(ysrc + 4 * c0, ysrc + 4 * c0 + 2, rowj + 4 * c0, rowj + 4 * c0 + 2, ysrc + 4 * j, ysrc + 4 * j + 2, ky, kr, ky, yp, aj, yj)) {
                if ((++spins & 255) == 0 &&
                    (spins > (1 << 19) || __hip_atomic_load(cp.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                    __hip_atomic_fetch_or(cp.abort_flag, ABORT_PATIENCE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    sh_ok = 0;                                     // seen by everybody after the next barrier
                    break;
                }
            }
            // (component by component: a ternary between two double2 LVALUES selects an address, and both operands then live
            // in scratch - 80 bytes of it and a memory round trip per use made both XCD-local variants slower than the agent-scope one)
            const bool oky = ok && j > 0;
            yp = make_double2(oky ? yp.x : 0.0, oky ? yp.y : 0.0);
            aj = make_double2(ok ? aj.x : 0.0, ok ? aj.y : 0.0);
            yj = make_double2(j > 0 ? yj.x : 0.0, j > 0 ? yj.y : 0.0);
        } else if constexpr (XCNT) {
            if (!xcd_wait(cp, (unsigned int)G * (unsigned int)(j + 1), &sh_ok)) return;
            const double2* yprev = ybuf + (int64_t)((j + 1) & 1) * ld;
            const double2* rowj = rowbuf + (int64_t)(j & 1) * ld;
            const int c0 = ok ? idx : j;
            ld_x3(yprev + c0, rowj + c0, yprev + j, yp, aj, yj);       // (step 0 reads parity buffers nobody has written: dropped below)
            const bool oky = ok && j > 0;
            yp = make_double2(oky ? yp.x : 0.0, oky ? yp.y : 0.0);
            aj = make_double2(ok ? aj.x : 0.0, ok ? aj.y : 0.0);
            yj = make_double2(j > 0 ? yj.x : 0.0, j > 0 ? yj.y : 0.0);
        } else {
            if (!bt_coop_wait<SC_AGENT>(cp, (unsigned int)G * (unsigned int)(j + 1), &sh_ok)) return;
            const double2* yprev = ybuf + (int64_t)((j + 1) & 1) * ld;
            const double2* rowj = rowbuf + (int64_t)(j & 1) * ld;
            yp = (ok && j > 0) ? ld_c(yprev + idx) : zero;
            aj = ok ? ld_c(rowj + idx) : zero;
            yj = j > 0 ? ld_c(yprev + j) : zero;
        }
        const double2 vp = (ok && j > 0) ? xs[idx] : zero;
        // (a) alpha_{j-1} = -tau/2 (y^H v)
        double2 alpha;
        {
            double sr = yp.x * vp.x + yp.y * vp.y, si = yp.x * vp.y - yp.y * vp.x;        // conj(y) v
            sr = wave_sum_fast(sr);
            si = wave_sum_fast(si);
            if (lane == 0) {
                red_a[wave][0] = sr;
                red_a[wave][1] = si;
            }
            __syncthreads();
            if (TAGGED && sh_ok == 0) return;          // somebody's poll ran out of patience: all leave together
            double ar = 0.0, ai = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                ar += red_a[w][0];
                ai += red_a[w][1];
            }
            alpha = c_mul(make_double2(-0.5 * tau_prev.x, -0.5 * tau_prev.y), make_double2(ar, ai));
        }
        // (b) column j with the pending update: conj(row j) - v conj(w_j) - w conj(v_j); v_{j-1}[j] is the leading one
        const double2 wj = j > 0 ? make_double2(yj.x + alpha.x, yj.y + alpha.y) : zero;
        double2 col;
        double2 tau = zero;
        double beta, dj;
        {
            if (j > 0) {
                const double2 av = c_mul(alpha, vp);
                yp = make_double2(yp.x + av.x, yp.y + av.y);                                   // w_{j-1} from here on
            } else {
                yp = zero;
            }
            const double2 t1 = c_mulc(vp, wj);
            col = make_double2(aj.x - t1.x - (j > 0 ? yp.x : 0.0), -aj.y - t1.y - (j > 0 ? yp.y : 0.0));
            double s = (idx >= j + 2 && ok) ? col.x * col.x + col.y * col.y : 0.0;
            if (ok) {
                vl[idx] = vp;
                wl[idx] = yp;
            }
            if (tid == 0) bc[0] = col;                 // d_j (its real part)
            if (tid == 1) bc[1] = col;                 // the leading entry of the column below the diagonal
            s = wave_sum_fast(s);
            if (lane == 0) red_b[wave] = s;
            __syncthreads();
            s = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) s += red_b[w];
            const double2 a0 = bc[1];
            dj = bc[0].x;
            // zlarfg
            double2 scale = zero;
            const double xx = a0.x * a0.x + a0.y * a0.y + s;
            if ((s > 0.0 || a0.y != 0.0) && xx > 1e-280) {
                // (reciprocal square root + two Heron steps and Newton reciprocals, as bt_reflector_scalars: the divisions and the
                // square root of the library forms sit on every workgroup's critical path, 511 times per solve)
                const double rs = __builtin_amdgcn_rsq(xx), hrs = 0.5 * rs;
                double nrm = xx * rs;
                nrm = fma(fma(-nrm, nrm, xx), hrs, nrm);
                nrm = fma(fma(-nrm, nrm, xx), hrs, nrm);
                beta = -copysign(nrm, a0.x);
                const double rb = frcp(beta);
                tau = make_double2((beta - a0.x) * rb, -a0.y * rb);
                const double dr = a0.x - beta, di = a0.y, dn = frcp(dr * dr + di * di);
                scale = make_double2(dr * dn, -di * dn);                                        // 1 / (a0 - beta)
            } else {
                beta = a0.x;
            }
            if (ok) xs[idx] = idx <= j ? zero : (idx == j + 1 ? make_double2(1.0, 0.0) : c_mul(col, scale));
            __syncthreads();
        }
        auto bookkeeping = [&]() {
            const int per = (n + G - 1) / G, i2 = g * per + tid;
            if (tid < per && i2 >= j && i2 < n) Vc[(int64_t)j * ld + i2] = xs[i2];
            if (g == j % G && tid == 0) {
                tauc[j] = tau;
                b.dd[j] = dj;
                b.ee[j] = beta;
            }
        };
        tau_prev = tau;
        if (j == n - 2) {
            bookkeeping();
            if ((n - 1) % G == g && tid == 0) {
                const int k = (n - 1 - g) / G;
                const double2 vv = vl[n - 1], ww = wl[n - 1];
                b.dd[n - 1] = rows[k * ld + (n - 1)].x - 2.0 * (vv.x * ww.x + vv.y * ww.y);      // - 2 Re(v conj(w))
                b.ee[n - 1] = 0.0;
            }
            break;
        }
        // (c) own rows r > j: pending update, product with the new reflector; the owner of row j+1 publishes it
        double2* ynew = ybuf + (int64_t)(j & 1) * ld;
        double2* rownext = rowbuf + (int64_t)((j + 1) & 1) * ld;
        double* ynew_t = cp.ybuf + (int64_t)(j & 1) * 4 * ld;
        double* rownext_t = cp.rowbuf + (int64_t)((j + 1) & 1) * 4 * ld;
        const unsigned long long key_y = tag_key(seq, 1, j), key_r = tag_key(seq, 2, j + 1);
        for (int k = wave; k < nown; k += NW) {
            const int r = g + k * G;
            if (r <= j) continue;
            double2* arow = rows + k * ld;
            const double2 vr = vl[r], wr = wl[r];
            const bool pub = r == j + 1;
            double sr = 0.0, si = 0.0;
            for (int c = j + 1 + lane; c < n; c += 64) {
                const double2 t1 = c_mulc(vr, wl[c]), t2 = c_mulc(wr, vl[c]);
                const double2 a_ = make_double2(arow[c].x - t1.x - t2.x, arow[c].y - t1.y - t2.y);
                arow[c] = a_;
                const double2 pr = c_mul(a_, xs[c]);
                sr += pr.x;
                si += pr.y;
                if (pub) {
                    if constexpr (SOLO) rl[((j + 1) & 1) * ld + c] = a_;
                    else if constexpr (TAGGED) put_c(rownext_t, c, a_, key_r);
                    else if constexpr (XCNT) st_x(rownext + c, a_);
                    else st_c(rownext + c, a_);
                }
            }
            sr = wave_sum_fast(sr);
            si = wave_sum_fast(si);
            if (lane == 0) {
                if constexpr (SOLO) yl[(j & 1) * ld + r] = c_mul(tau, make_double2(sr, si));
                else if constexpr (TAGGED) put_c(ynew_t, r, c_mul(tau, make_double2(sr, si)), key_y);
                else if constexpr (XCNT) st_x(ynew + r, c_mul(tau, make_double2(sr, si)));
                else st_c(ynew + r, c_mul(tau, make_double2(sr, si)));
            }
        }
        if constexpr (SOLO) __syncthreads();
        else if constexpr (XCNT) xcd_arrive(cp);
        else if constexpr (!TAGGED) bt_coop_arrive<SC_AGENT>(cp);
        bookkeeping();
    }
}

// u = H_0 H_1 ... H_{n-2} z for the eigenvectors z of T that k_bt_vec left in rows 2k of Z (first n entries, real): one WAVE per
// vector (8 elements per lane at n = 512, reductions on DPP: no workgroup barrier), reflectors from L2 (the four waves of a
// workgroup walk the same rows).  Writes (Re u, Im u) into row 2k and J u = (-Im u, Re u) into row 2k + 1.
__global__ __launch_bounds__(256) void k_bt_back_c(View v, int lid, int going_left, int rawn, BtBufs b) {
    if (bt_idle(b)) return;
    const BtProblem pb = bt_resolve(v, lid, going_left, nullptr, rawn);
    if (bt_aborted(b)) return;
    const int n = pb.n >> 1, ld = b.ncap, ldc = b.ncap >> 1, lane = threadIdx.x & 63;
    const int k = (int)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= bt_nvec(pb)) return;
    constexpr int QE = BTC_NMAX / 64;
    const int nq = (n + 63) >> 6;            // live 64-element slots (2 of 8 at n = 96: the Rayleigh-Ritz problem of the subspace solver)
    double* zrow = b.Z + (int64_t)(2 * k) * ld;
    double2 z[QE];
#pragma unroll
    for (int q = 0; q < QE; ++q) {
        const int i = lane + 64 * q;
        z[q] = make_double2(i < n ? zrow[i] : 0.0, 0.0);
    }
    const double2* Vc = (const double2*)b.Vall;
    const double2* tauc = (const double2*)b.tau;
    // reflectors (and their tau) are requested ahead of their application: the loop is a chain of L2 round trips otherwise
    // (0.43 ms for 511 reflectors at n = 512; one ahead 0.34)
    auto fetch = [&](int j, double2 (&dst)[QE], double2& tj) {
        const double2* vj = Vc + (int64_t)(j >= 0 ? j : 0) * ldc;
#pragma unroll
        for (int q = 0; q < QE; ++q) {
            if (q < nq) {
                const int i = lane + 64 * q;
                dst[q] = (j >= 0 && i > j && i < n) ? vj[i] : make_double2(0.0, 0.0);
            }
        }
        tj = j >= 0 ? tauc[j] : make_double2(0.0, 0.0);
    };
    auto apply = [&](const double2 (&vv)[QE], double2 tj) {
        double sr = 0.0, si = 0.0;
#pragma unroll
        for (int q = 0; q < QE; ++q) {
            if (q < nq) {
                sr += vv[q].x * z[q].x + vv[q].y * z[q].y;             // conj(v) z
                si += vv[q].x * z[q].y - vv[q].y * z[q].x;
            }
        }
        sr = wave_sum_fast(sr);
        si = wave_sum_fast(si);
        const double2 t = c_mul(tj, make_double2(sr, si));
#pragma unroll
        for (int q = 0; q < QE; ++q) {
            if (q < nq) {
                const double2 u = c_mul(t, vv[q]);
                z[q].x -= u.x;
                z[q].y -= u.y;
            }
        }
    };
    // four reflectors in flight (a ring of register buffers): with one ahead the loop waited an L2 round trip every other step
    double2 va[QE], vb[QE], vc[QE], vd[QE], ta, tb, tc, td;
    fetch(n - 2, va, ta);
    fetch(n - 3, vb, tb);
    fetch(n - 4, vc, tc);
    for (int j = n - 2; j >= 0; j -= 4) {
        fetch(j - 3, vd, td);
        apply(va, ta);
        if (j - 1 < 0) break;
        fetch(j - 4, va, ta);
        apply(vb, tb);
        if (j - 2 < 0) break;
        fetch(j - 5, vb, tb);
        apply(vc, tc);
        if (j - 3 < 0) break;
        fetch(j - 6, vc, tc);
        apply(vd, td);
    }
    double* zp = zrow + ld;
#pragma unroll
    for (int q = 0; q < QE; ++q) {
        const int i = lane + 64 * q;
        if (i < n) {
            zrow[i] = z[q].x;
            zrow[n + i] = z[q].y;
            zp[i] = -z[q].y;
            zp[n + i] = z[q].x;
        }
    }
    if (lane == 0) {
        b.lam[2 * k + 1] = b.lam[2 * k];
        b.res[2 * k + 1] = b.res[2 * k];
    }
}

// ---- compact-WY factors of the reflector blocks ------------------------------------------------------------------------
// Block bk = reflectors j0 .. j0+15 (j0 = 16 bk): H_{j0} H_{j0+1} ... H_{j0+15} = I - V T V^T with V = [v_j0 ... v_j0+15] and T
// upper triangular (LAPACK dlarft, forward / columnwise): T_ii = tau_i, T_{0:i,i} = -tau_i T_{0:i,0:i} (V_{0:i}^T v_i).
// One workgroup per block: S = V^T V on the fp64 MFMA (the contraction over the rows split over the 4 waves), then one wave
// runs the 16-step recurrence, lane r holding row r of T.  Reflectors beyond n-3 (the tail of the last block) count as
// identities (tau = 0).
constexpr int BT_NB = 16;
__global__ __launch_bounds__(BT_T) void k_bt_larft(View v, int lid, int going_left, int rawn, BtBufs b) {
    if (bt_idle(b)) return;
    __shared__ double part[4][256];
    __shared__ double S[16][17];
    const BtProblem pb = bt_resolve(v, lid, going_left, nullptr, rawn);
    if (bt_aborted(b) || b.cnative) return;
    const int n = pb.n, ld = b.ncap, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, kq = lane >> 4;
    const int j0 = BT_NB * (int)blockIdx.x, nref = n - 2;          // reflectors 0 .. n-3
    if (j0 >= nref) return;
    // rows n .. ncap-1 of this block's reflectors may hold entries of an earlier solve at a larger n (bond dimensions differ
    // near the ends of the chain): cleared here, so that the back-transformation can fetch whole rows without a per-lane test
    for (int e = tid; e < BT_NB * (ld - n); e += BT_T) {
        const int m = e / (ld - n), r = n + e - m * (ld - n);
        if (j0 + m < nref) b.Vall[(int64_t)(j0 + m) * ld + r] = 0.0;
    }
    const int jm = j0 + i16;                                        // this lane's reflector (as A row and as B column)
    const bool jv = jm < nref;
    const double* vrow = b.Vall + (int64_t)jm * ld;
    // rows j0+1 .. n-1 carry the block (row j+1 is the leading one of reflector j); 4 rows per k-step
    const int rbeg = j0 + 1, nrows = n - rbeg;
    const int ks = (((nrows + 3) >> 2) + 3) >> 2;                   // k-steps per wave
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    for (int u0 = wave * ks; u0 < (wave + 1) * ks; u0 += 8) {
        double x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int row = rbeg + 4 * (u0 + u) + kq;
            x[u] = (jv && u0 + u < (wave + 1) * ks && row < n) ? vrow[row] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            acc = mfma_f64(x[u], x[u], acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) part[wave][r * 64 + lane] = acc[r];
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            S[kq + 4 * r][i16] = (part[0][r * 64 + lane] + part[1][r * 64 + lane]) + (part[2][r * 64 + lane] + part[3][r * 64 + lane]);
    }
    __syncthreads();
    if (wave == 0 && lane < 16) {
        const int r = lane;
        double T[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) T[m] = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const double tau = (j0 + i < nref) ? b.tau[j0 + i] : 0.0;
            double s = 0.0;
#pragma unroll
            for (int m = 0; m < 16; ++m)
                if (m < i) s = fma(T[m], S[m][i], s);               // T[r][m] is zero for m < r
            T[i] = r < i ? -tau * s : (r == i ? tau : 0.0);
        }
        double* out = b.Tfac + ((int64_t)blockIdx.x * 16 + r) * 16;
#pragma unroll
        for (int m = 0; m < 16; ++m) out[m] = T[m];
    }
}

__global__ __launch_bounds__(BT_T) void k_bt_vec(View v, int lid, int going_left, int rawn, BtBufs b) {
    if (bt_idle(b)) return;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double red[4];
    __shared__ int cnt_s[4];
    __shared__ int arg_s[4];
    const BtProblem pb = bt_resolve(v, lid, going_left, nullptr, rawn);
    // native pair mode (k_bt_coop_c): T is the real tridiagonal matrix of the Hermitian problem itself - order n/2, simple spectrum;
    // this kernel then stops at the eigenvector of T (k_bt_back_c applies the complex reflectors)
    const bool nat = b.cnative != 0;
    const int n = nat ? pb.n >> 1 : pb.n, ld = b.ncap, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k = blockIdx.x;
    if (k >= bt_nvec(pb) || bt_aborted(b)) return;
    double* de = smem;                  // [n + 8][2] (d_j, e_{j-1}^2), padded for sturm_count's groups of 8 rows
    double* es = de + 2 * (BT_NMAX + 8);      // [n]
    double* Dp = es + BT_NMAX;          // [n] forward pivots
    double* Dm = Dp + BT_NMAX;          // [n] backward pivots
    double* z = Dm + BT_NMAX;           // [n]
    for (int j = tid; j < n; j += BT_T) {
        const double e = j > 0 ? b.ee[j - 1] : 0.0;
        de[2 * j] = b.dd[j];
        de[2 * j + 1] = e * e;
        es[j] = b.ee[j];
    }
    __syncthreads();
    // Gershgorin bounds
    double gl = 1e300, gu = -1e300;
    for (int j = tid; j < n; j += BT_T) {
        const double a = j > 0 ? fabs(es[j - 1]) : 0.0, c = j < n - 1 ? fabs(es[j]) : 0.0;
        gl = fmin(gl, de[2 * j] - a - c);
        gu = fmax(gu, de[2 * j] + a + c);
    }
    gl = -wave_max(-gl);
    gu = wave_max(gu);
    if (lane == 0) {
        red[wave] = gl;
    }
    __syncthreads();
    gl = fmin(fmin(red[0], red[1]), fmin(red[2], red[3]));
    __syncthreads();
    if (lane == 0) red[wave] = gu;
    __syncthreads();
    gu = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    __syncthreads();
    const double tnorm = fmax(fabs(gl), fabs(gu));
    const double pad = 2.0 * n * 2.3e-16 * tnorm + 1e-300;
    double lo = gl - pad, hi = gu + pad;
    // pad T to 1 + a multiple of 8 rows with decoupled rows (e^2 = 0) whose diagonal lies above every abscissa: they add no
    // sign change and let sturm_count run in whole groups of 8
    {
        const double dpad = hi + (hi - lo) + 1.0;
        if (tid < 16) {
            de[2 * n + tid] = (tid & 1) ? 0.0 : dpad;
        }
    }
    __syncthreads();
    // multisection for the k-th largest eigenvalue (pair mode: the upper one of the k-th pair)
    const int target = n - 1 - ((pb.pair && !nat) ? 2 * k : k);
    if (n >= 24) {
        // two-sided Sturm count as in k_eig_vec: a pair of lanes runs the recurrence top-down over rows 0..kk-1 and
        // bottom-up over rows n-1..kk+1, the sign of the twisted pivot at row kk completes the inertia; 128 abscissae x 8
        // rounds of half the length instead of 256 x 7
        double* deR = Dp;                               // [n] pairs of the reversed matrix (Dp, Dm are free until the bisection is over)
        const int kk = ((n >> 1) & ~7) + 1;
        for (int m = tid; m < n; m += BT_T) {
            deR[2 * m] = de[2 * (n - 1 - m)];
            deR[2 * m + 1] = m >= 1 ? de[2 * (n - m) + 1] : 0.0;
        }
        __syncthreads();
        const bool bottom = tid & 1;
        const double* arr = bottom ? deR : de;
        const int rows = bottom ? n - 1 - kk : kk;
        const double dk = de[2 * kk], e2a = de[2 * kk + 1], e2b = de[2 * (kk + 1) + 1];
        constexpr int NPT = BT_T / 2;
        for (int it = 0; it < 8; ++it) {
            const double h = (hi - lo) * (1.0 / (NPT + 1));
            const double xq = lo + h * ((tid >> 1) + 1);
            double p1, p2;
            int cnt = sturm_half(arr, rows, xq, p1, p2);
            const int ocnt = __builtin_amdgcn_update_dpp(0, cnt, 0xB1, 0xF, 0xF, true);     // the partner lane (lane ^ 1)
            const double o1 = dpp_mov<0xB1>(p1), o2 = dpp_mov<0xB1>(p2);
            const double P1 = bottom ? o1 : p1, P2 = bottom ? o2 : p2, Q1 = bottom ? p1 : o1, Q2 = bottom ? p2 : o2;
            const double gq = fma(dk - xq, P1 * Q1, -fma(e2a * P2, Q1, e2b * Q2 * P1));
            cnt += ocnt + (((hi32(gq) ^ hi32(P1) ^ hi32(Q1)) >> 31) & 1);
            const unsigned long long bal = __ballot(cnt <= target) & 0x5555555555555555ull;
            if (lane == 0) cnt_s[wave] = __popcll(bal);
            __syncthreads();
            const int jj = cnt_s[0] + cnt_s[1] + cnt_s[2] + cnt_s[3];
            __syncthreads();
            const double nlo = jj > 0 ? lo + h * jj : lo;
            const double nhi = jj < NPT ? lo + h * (jj + 1) : hi;
            lo = nlo;
            hi = nhi;
        }
        __syncthreads();
    } else {
        for (int it = 0; it < 7; ++it) {
            const double h = (hi - lo) * (1.0 / (BT_T + 1));
            const double xq = lo + h * (tid + 1);
            const int cnt = sturm_count(de, n, xq);
            const unsigned long long bal = __ballot(cnt <= target);
            if (lane == 0) cnt_s[wave] = __popcll(bal);
            __syncthreads();
            const int jj = cnt_s[0] + cnt_s[1] + cnt_s[2] + cnt_s[3];
            __syncthreads();
            const double nlo = jj > 0 ? lo + h * jj : lo;
            const double nhi = jj < BT_T ? lo + h * (jj + 1) : hi;
            lo = nlo;
            hi = nhi;
        }
    }
    const double lamk = 0.5 * (lo + hi);
    // twisted factorisation (dlar1v): forward pivots on wave 0, backward pivots on wave 1, one lane each
    const double pivmin = 1e-290 + 1e-30 * tnorm;
    if (tid == 0) {
        double dcur = de[0] - lamk;
        if (!(fabs(dcur) >= pivmin)) dcur = -pivmin;
        Dp[0] = dcur;
        for (int j = 1; j < n; ++j) {
            dcur = (de[2 * j] - lamk) - de[2 * j + 1] * frcp(dcur);
            if (!(fabs(dcur) >= pivmin)) dcur = -pivmin;
            Dp[j] = dcur;
        }
    } else if (tid == 64) {
        double dcur = de[2 * (n - 1)] - lamk;
        if (!(fabs(dcur) >= pivmin)) dcur = -pivmin;
        Dm[n - 1] = dcur;
        for (int j = n - 2; j >= 0; --j) {
            dcur = (de[2 * j] - lamk) - de[2 * (j + 1) + 1] * frcp(dcur);
            if (!(fabs(dcur) >= pivmin)) dcur = -pivmin;
            Dm[j] = dcur;
        }
    }
    __syncthreads();
    // twist index: argmin |gamma_j|, gamma_j = D+_j + D-_j - (d_j - lambda)
    double g = 1e300;
    int gi = 0;
    for (int j = tid; j < n; j += BT_T) {
        const double gj = fabs(Dp[j] + Dm[j] - (de[2 * j] - lamk));
        if (gj < g) {
            g = gj;
            gi = j;
        }
    }
    {
        const double gmin = -wave_max(-g);
        const unsigned long long msk = __ballot(g == gmin);
        const int first = __ffsll((long long)msk) - 1;
        const int idx = __builtin_amdgcn_readlane(gi, first);
        if (lane == 0) {
            red[wave] = gmin;
            arg_s[wave] = idx;
        }
    }
    __syncthreads();
    int rb = arg_s[0];
    {
        double gm = red[0];
        for (int w = 1; w < 4; ++w)
            if (red[w] < gm) {
                gm = red[w];
                rb = arg_s[w];
            }
    }
    __syncthreads();
    // the multipliers of the two substitutions do not depend on the running product: all threads form them first (the
    // reciprocals leave the serial chains, which are then one multiply per row), in place of the pivots they replace
    for (int j = tid; j < n; j += BT_T) {
        if (j < rb) Dp[j] = -(es[j] * frcp(Dp[j]));
        else if (j > rb) Dm[j] = -(es[j - 1] * frcp(Dm[j]));
    }
    __syncthreads();
    if (tid == 0) {
        double zc = 1.0;
        z[rb] = 1.0;
        for (int j = rb - 1; j >= 0; --j) {
            zc = Dp[j] * zc;
            z[j] = zc;
        }
    } else if (tid == 64) {
        double zc = 1.0;
        for (int j = rb + 1; j < n; ++j) {
            zc = Dm[j] * zc;
            z[j] = zc;
        }
    }
    __syncthreads();
    double nrm = 0.0;
    for (int j = tid; j < n; j += BT_T) nrm = fma(z[j], z[j], nrm);
    nrm = bt_block_sum(nrm, red);
    const double sc = 1.0 / sqrt(nrm);
    double ri = 0.0;
    for (int j = tid; j < n; j += BT_T) {
        const double zj = z[j] * sc, zp = j > 0 ? z[j - 1] * sc : 0.0, zn = j < n - 1 ? z[j + 1] * sc : 0.0;
        ri = fmax(ri, fabs((de[2 * j] - lamk) * zj + (j > 0 ? es[j - 1] * zp : 0.0) + (j < n - 1 ? es[j] * zn : 0.0)));
    }
    ri = wave_max(ri);
    __syncthreads();
    if (lane == 0) red[wave] = ri;
    __syncthreads();
    ri = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    __syncthreads();
    for (int j = tid; j < n; j += BT_T) z[j] *= sc;
    __syncthreads();
    if (nat) {
        for (int j = tid; j < n; j += BT_T) b.Z[(int64_t)(2 * k) * ld + j] = z[j];
        if (tid == 0) {
            b.lam[2 * k] = lamk;
            b.res[2 * k] = tnorm > 0.0 ? ri / tnorm : ri;
        }
        return;
    }
    // back-transformation: z <- H_0 H_1 ... H_{n-3} z in blocks of 16 reflectors, last block first:
    //     z <- z - V (T (V^T z)),   T from k_bt_larft.
    // Every thread keeps its rows of z (r = tid + 256 q) in registers and fetches its rows of the block's 16 reflectors (for
    // n <= 512 one block ahead); the 16 partial products of a thread meet through an LDS transpose (16 values x 256 threads ->
    // 16 sums), T is applied by 16 lanes, and the update is 16 FMAs per row.  Three barriers per 16 reflectors instead of
    // eight rounds of one barrier and three wave sums each (0.18 -> 0.05 ms at n = 512).
    {
        constexpr int QV = BT_NMAX / BT_T;
        // [16][257] partial products: over T, e^2 and the pivots at the head of the dynamic LDS (2064 + 1024 + 1024 doubles,
        // all consumed by now); z behind them stays
        static_assert(2 * (BT_NMAX + 8) + 2 * BT_NMAX >= 16 * (BT_T + 1), "the partial products fit over de, es, Dp");
        double (*Ps)[BT_T + 1] = (double (*)[BT_T + 1])smem;
        __shared__ double Ws[16], W2[16], Ts[16][17];
        __syncthreads();
        const int nref = n - 2, nblk = (nref + BT_NB - 1) / BT_NB;
        const int nq = (n + BT_T - 1) / BT_T;                     // live row slots per thread
        const bool ahead = nq <= 2;
        double zr[QV];
#pragma unroll
        for (int q = 0; q < QV; ++q) {
            const int r = tid + BT_T * q;
            zr[q] = r < n ? z[r] : 0.0;
        }
        // rows tid, tid + 256 of block bk: entries at rows <= j are zero in Vall (never written); the tail of the last block
        // is a wave-uniform test, rows >= n a per-thread constant
        const bool rv0 = tid < n, rv1 = tid + BT_T < n;          // (the row stride may equal n: a row >= n would alias the next reflector)
        auto fetch_block = [&](int bk, double (&dst)[16][2]) {
            const double* base = b.Vall + (int64_t)(BT_NB * (bk > 0 ? bk : 0)) * ld + tid;
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                const bool live = bk >= 0 && BT_NB * bk + m < nref;
                dst[m][0] = (live && rv0) ? base[0] : 0.0;
                dst[m][1] = (live && rv1) ? base[BT_T] : 0.0;
                base += ld;
            }
        };
        // one block: partial products -> LDS transpose -> 16 sums -> T -> update; V rows in vb (nq <= 2) or fetched here
        auto apply_block = [&](int bk, double (&vb2)[16][2], double tcur) {
            double vw[16][QV - 2];                                     // row slots 2, 3 (n > 512): fetched per block, no prefetch
            if (!ahead) {
                fetch_block(bk, vb2);
                const double* base = b.Vall + (int64_t)(BT_NB * bk) * ld + tid + 2 * BT_T;
#pragma unroll
                for (int m = 0; m < 16; ++m) {
                    const bool live = BT_NB * bk + m < nref;
#pragma unroll
                    for (int q = 0; q < QV - 2; ++q) vw[m][q] = (live && tid + BT_T * (q + 2) < n) ? base[BT_T * q] : 0.0;
                    base += ld;
                }
            }
            Ts[tid >> 4][tid & 15] = tcur;
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                double sm = fma(vb2[m][1], zr[1], vb2[m][0] * zr[0]);
                if (!ahead) {
#pragma unroll
                    for (int q = 0; q < QV - 2; ++q) sm = fma(vw[m][q], zr[q + 2], sm);
                }
                Ps[m][tid] = sm;
            }
            __syncthreads();
            {
                const int m = tid >> 4, c = tid & 15;
                double sm = 0.0;
#pragma unroll
                for (int u = 0; u < 16; ++u) sm += Ps[m][u * 16 + c];       // lanes c read consecutive words: no bank conflict
                sm = sum16(sm);
                if (c == 0) Ws[m] = sm;
            }
            __syncthreads();
            if (tid < 16) {
                double t2 = 0.0;
#pragma unroll
                for (int m = 0; m < 16; ++m) t2 = fma(Ts[tid][m], Ws[m], t2);       // T is upper triangular: zeros below
                W2[tid] = t2;
            }
            __syncthreads();
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                const double wm = W2[m];
                zr[0] = fma(-wm, vb2[m][0], zr[0]);
                zr[1] = fma(-wm, vb2[m][1], zr[1]);
                if (!ahead) {
#pragma unroll
                    for (int q = 0; q < QV - 2; ++q) zr[q + 2] = fma(-wm, vw[m][q], zr[q + 2]);
                }
            }
        };
        double va[16][2], vbb[16][2];
        if (ahead && nblk > 0) fetch_block(nblk - 1, va);
        double ta = nblk > 0 ? b.Tfac[(int64_t)(nblk - 1) * 256 + tid] : 0.0, tb = 0.0;      // T one block ahead as well
        // two blocks per pass, the buffers alternating: the next block's rows and T fly during a block's work
        for (int bk = nblk - 1; bk >= 0; bk -= 2) {
            if (bk > 0) {
                if (ahead) fetch_block(bk - 1, vbb);
                tb = b.Tfac[(int64_t)(bk - 1) * 256 + tid];
            }
            apply_block(bk, va, ta);
            if (bk > 0) {
                if (bk > 1) {
                    if (ahead) fetch_block(bk - 2, va);
                    ta = b.Tfac[(int64_t)(bk - 2) * 256 + tid];
                }
                apply_block(bk - 1, vbb, tb);
            }
        }
#pragma unroll
        for (int q = 0; q < QV; ++q) {
            const int r = tid + BT_T * q;
            if (r < n) z[r] = zr[q];
        }
        __syncthreads();
    }
    const int kc = pb.pair ? 2 * k : k;
    for (int j = tid; j < n; j += BT_T) b.Z[(int64_t)kc * ld + j] = z[j];
    if (tid == 0) {
        b.lam[kc] = lamk;
        b.res[kc] = tnorm > 0.0 ? ri / tnorm : ri;
    }
    if (pb.pair) {
        // the partner in the double eigenspace: J u = (-u_im, u_re) - i times the complex eigenvector
        const int nc = n >> 1;
        for (int j = tid; j < n; j += BT_T) b.Z[(int64_t)(kc + 1) * ld + j] = j < nc ? -z[j + nc] : z[j - nc];
        if (tid == 0) {
            b.lam[kc + 1] = lamk;
            b.res[kc + 1] = tnorm > 0.0 ? ri / tnorm : ri;
        }
    }
}

// ---- verification, re-orthonormalisation, publication -------------------------------------------------------------------
// D = Z^T Z - I for all K0 candidate vectors: one 16 x 16 tile per workgroup, the n rows split over its 4 waves.
__global__ __launch_bounds__(BT_T) void k_bt_gram(View v, int lid, int going_left, int rawn, BtBufs b, int second) {
    if (bt_idle(b)) return;
    __shared__ double part[4][256];
    const BtProblem pb = bt_resolve(v, lid, going_left, nullptr, rawn);
    if (bt_aborted(b) || (second && !b.ctl[2])) return;
    const int n = pb.n, ld = b.ncap, K0 = pb.K0;
    const int tk = (K0 + 15) >> 4;
    if ((int)blockIdx.x >= tk * tk) return;
    const int a0 = ((int)blockIdx.x / tk) * 16, c0 = ((int)blockIdx.x % tk) * 16;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i16 = lane & 15, kq = lane >> 4;
    const int per = ((((n + 3) >> 2) + 3) >> 2) * 4;            // rows per wave, a multiple of 4
    const int rbeg = wave * per, rend = min(n, rbeg + per);
    const bool av = a0 + i16 < K0, cv = c0 + i16 < K0;
    const double* za = b.Z + (int64_t)(a0 + i16) * ld;
    const double* zc = b.Z + (int64_t)(c0 + i16) * ld;
    d4 acc = {0, 0, 0, 0};
    for (int r0 = rbeg; r0 < rend; r0 += 64) {
        double x[16], y[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int r = r0 + 4 * u + kq;
            x[u] = (av && r < rend) ? za[r] : 0.0;
            y[u] = (cv && r < rend) ? zc[r] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u)
            acc = mfma_f64(x[u], y[u], acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) part[wave][r * 64 + lane] = acc[r];
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int a = a0 + kq + 4 * r, c = c0 + i16;
            const double sum = (part[0][r * 64 + lane] + part[1][r * 64 + lane]) + (part[2][r * 64 + lane] + part[3][r * 64 + lane]);
            if (a < K0 && c < K0) b.D[(int64_t)a * CAP_LIMIT + c] = sum - (a == c ? 1.0 : 0.0);
        }
    }
}

// truncation rule (as k_eig_fin / k_big_fin), verdict on the kept vectors, publication of the scalars.
// ctl: [0] vectors to publish, [1] verdict ok, [2] a second Loewdin round is needed, [3] first round needed at all
__global__ __launch_bounds__(512) void k_bt_decide(View v, int lid, int going_left, const double* rawG, int rawn, BtBufs b,
                                                   double* rawlam, int32_t* rawinfo, int second) {
    if (bt_idle(b)) return;
    __shared__ double lam_s[CAP_LIMIT + 2];
    __shared__ double red[16];
    const BtProblem pb = bt_resolve(v, lid, going_left, rawG, rawn);
    const bool raw = rawn > 0;
    const int tid = threadIdx.x;
    const int n = pb.n, nspec = pb.nspec, K0 = pb.K0;
    if (bt_aborted(b)) {                // nothing is published; the host redoes the bond (it reads the abort word first)
        if (tid == 0) *b.flag = 1;
        return;
    }
    if (second) {
        // after the first polish: is the deviation of the polished vectors already at rounding?
        if (!b.ctl[2]) return;
        const int kout = b.ctl[0];
        double emax = 0.0;
        for (int e = tid; e < kout * kout; e += 512) emax = fmax(emax, fabs(b.D[(int64_t)(e / kout) * CAP_LIMIT + (e % kout)]));
        emax = wave_max(emax);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = emax;
        __syncthreads();
        emax = 0.0;
        for (int i = 0; i < 8; ++i) emax = fmax(emax, red[i]);
        if (tid == 0 && !(emax < 1e-6)) {          // the first round should have squared it
            b.ctl[1] = 0;
            b.ctl[2] = 0;
            *b.flag = 1;
        }
        return;
    }
    double part = 0.0, partT = 0.0;
    const int nT = b.cnative ? n >> 1 : n;          // native pair mode: T is the tridiagonal matrix of the Hermitian problem (order n/2)
    for (int i = tid; i < n; i += 512) {
        part += pb.G[(size_t)i * n + i];
        if (i < nT) partT += b.dd[i];
    }
    double tr = wave_sum(part), trT = wave_sum(partT);
    if ((tid & 63) == 0) {
        red[tid >> 6] = tr;
        red[8 + (tid >> 6)] = trT;
    }
    __syncthreads();
    tr = 0.0;
    trT = 0.0;
    for (int i = 0; i < 8; ++i) {
        tr += red[i];
        trT += red[8 + i];
    }
    // the one check of T against G itself (residuals and orthonormality below are relative to T): an orthogonal
    // similarity keeps the trace, so a tridiagonal matrix that belongs to another bond does not pass
    const double trG = b.cnative ? 0.5 * tr : tr;       // trace of the complex matrix = half the embedding's
    const bool trace_ok = fabs(trT - trG) <= 1e-9 * fabs(trG) + 1e-300;
    if (tid < K0) lam_s[tid] = fmax(b.lam[tid], 0.0);
    __syncthreads();
    if (pb.pair) tr *= 0.5;             // trace of the complex matrix
    const double inv = (!raw && v.rescale_after) ? 1.0 / sqrt(tr) : 1.0;
    const double cutoff = raw ? -1.0 : v.cutoff;
    const double inv2 = inv * inv;
    const int st = pb.pair ? 2 : 1;
    int nk = st * truncate_rule(lam_s, K0, nspec, tr, inv2, cutoff, pb.pair);      // real vectors to verify and publish
    const int kout = raw ? K0 : nk;
    double rmax = tid < kout ? b.res[tid] : 0.0;
    rmax = wave_max(rmax);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = rmax;
    __syncthreads();
    rmax = 0.0;
    for (int i = 0; i < 8; ++i) rmax = fmax(rmax, red[i]);
    __syncthreads();
    // subspace phase: res[k] = |G v_k - theta_k v_k| / sqrt(theta_k tr); the root of the sum of their squares bounds the error of the
    // projected two-site tensor, ||M (P~ - P)||_F / ||M||_F (mpst_eig_subspace.inl); 1e300 marks a failed Frobenius certificate
    double r2 = (b.ss && tid < kout) ? b.res[tid] * b.res[tid] : 0.0;
    r2 = wave_sum(r2);
    if ((tid & 63) == 0) red[tid >> 6] = r2;
    __syncthreads();
    r2 = 0.0;
    for (int i = 0; i < 8; ++i) r2 += red[i];
    __syncthreads();
    const bool res_ok = b.ss ? (sqrt(r2) < 1e-9) : (rmax < 1e-8 && rmax == rmax);
    double emax = 0.0;
    for (int e = tid; e < kout * kout; e += 512) emax = fmax(emax, fabs(b.D[(int64_t)(e / kout) * CAP_LIMIT + (e % kout)]));
    emax = wave_max(emax);
    if ((tid & 63) == 0) red[tid >> 6] = emax;
    __syncthreads();
    emax = 0.0;
    for (int i = 0; i < 8; ++i) emax = fmax(emax, red[i]);
    // vectors of eigenvalues close to the cutoff come out of the twisted factorisation orthogonal to ~1e-6 only (two
    // Loewdin rounds take them to rounding); |D| > 1e-3 means a genuine cluster -> library solver
    // subspace phase, fp32 tensors (kept values no smaller than their rounding noise, theta_k >= 1e-13 theta_1): V = Z W Theta^-1/2 is
    // orthonormal to ~1e-9 and ONE Loewdin round takes that to rounding - no second round is enqueued there, so a larger deviation
    // counts as a failure (exact path).  fp64 tensors keep values down to the cutoff (theta_k = 1e-10 theta_1: the errors of W are
    // amplified by theta_1 / theta_k) and both rounds.
    const bool ok = res_ok && emax < ((b.ss && v.ss_f32) ? 1e-8 : 1e-3) && emax == emax && trace_ok;
    if (tid == 0) {
        b.ctl[0] = kout;
        b.ctl[1] = ok ? 1 : 0;
        b.ctl[2] = (ok && emax >= 1e-8) ? 1 : 0;
        b.ctl[3] = (ok && emax > 1e-14) ? 1 : 0;
        *b.flag = ok ? 0 : 1;
    }
    if (raw) {
        if (tid < K0) rawlam[tid] = lam_s[tid];
        if (tid == 0) *rawinfo = ok ? -3 : -4;
    } else if (ok) {
        if (tid < K0 / st) v.lam[tid] = lam_s[st * tid];
        if (tid == 0) {
            bool bad = !(tr == tr) || tr > 1e300;
            for (int i = 0; i < K0; ++i) {
                const double P = lam_s[i] * inv2;
                if (!(P == P) || P > 1e300) bad = true;
            }
            nk /= st;
            v.sc->n_keep = nk;
            v.sc->n_spec = K0 / st;
            v.sc->bt_norm2 = tr;
            v.sc->inv_norm = inv;
            v.sc->eig_sweeps = 0;
            if (bad) v.sc->status = MPST_ERR_SVD;
            v.chi[lid + 1] = nk;
        }
    }
}

// E = Z (I - D/2) over the kept vectors (Loewdin: squares the deviation from orthonormality without leaving the subspace):
// 16 x 16 tiles of E on the MFMA, A operand = Z^T, B operand = -D/2 + I.  second = 1: the input is the first round's E.
__global__ __launch_bounds__(BT_T) void k_bt_polish(View v, int lid, int going_left, int rawn, BtBufs b, double* rawE, int second) {
    if (bt_idle(b)) return;
    const BtProblem pb = bt_resolve(v, lid, going_left, nullptr, rawn);
    // the last launch of a solve: remember a failed verdict (verification, or the persistent kernel gave up) for callers that
    // look at it once per sweep instead of once per bond
    if (second && b.sticky && blockIdx.x == 0 && threadIdx.x == 0 && *b.flag != 0) *b.sticky = 1;
    if (bt_aborted(b) || !b.ctl[1] || (second && !b.ctl[2])) return;
    const bool raw = rawn > 0;
    const int n = pb.n, ld = b.ncap, kout = b.ctl[0];
    const bool corr = second ? true : (b.ctl[3] != 0);
    double* Eout = raw ? rawE : v.E;
    const int ldE = raw ? n : v.cap;
    const int tn = (n + 15) >> 4, tk = (kout + 15) >> 4;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i16 = lane & 15, kq = lane >> 4;
    for (int t = blockIdx.x * 4 + wave; t < tn * tk; t += gridDim.x * 4) {
        const int c0 = (t / tk) * 16, k0 = (t % tk) * 16;
        const int c = c0 + i16, kk = k0 + i16;
        d4 acc = {0, 0, 0, 0};
        if (corr) {
            for (int q0 = 0; q0 < kout; q0 += 64) {
                double x[16], y[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int k2 = q0 + 4 * u + kq;
                    x[u] = (c < n && k2 < kout) ? b.Z[(int64_t)k2 * ld + c] : 0.0;
                    y[u] = (kk < kout && k2 < kout) ? -0.5 * b.D[(int64_t)k2 * CAP_LIMIT + kk] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    acc = mfma_f64(x[u], y[u], acc);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int cr = c0 + kq + 4 * r, kc = k0 + i16;
            if (cr < n && kc < kout) Eout[(size_t)cr * ldE + kc] = b.Z[(int64_t)kc * ld + cr] + acc[r];
        }
    }
}

// second round: the polished vectors become the input (Z <- E^T)
__global__ __launch_bounds__(BT_T) void k_bt_copyback(View v, int lid, int going_left, int rawn, BtBufs b, const double* rawE) {
    if (bt_idle(b)) return;
    const BtProblem pb = bt_resolve(v, lid, going_left, nullptr, rawn);
    if (bt_aborted(b) || !b.ctl[2]) return;
    const bool raw = rawn > 0;
    const int n = pb.n, ld = b.ncap, kout = b.ctl[0];
    const double* E = raw ? rawE : v.E;
    const int ldE = raw ? n : v.cap;
    for (int64_t e = (int64_t)blockIdx.x * BT_T + threadIdx.x; e < (int64_t)n * kout; e += (int64_t)gridDim.x * BT_T) {
        const int kk = (int)(e / n), c = (int)(e - (int64_t)kk * n);
        b.Z[(int64_t)kk * ld + c] = E[(size_t)c * ldE + kk];
    }
}

#include "mpst_eig_subspace.inl"

// ---- host side ----------------------------------------------------------------------------------------------------------
struct BlockedEig {
    BtBufs b{};
    BtCoop cp{};
    int32_t* host_flag = nullptr;      // pinned: [0] verdict, [1] the persistent kernel gave up
    int coop_aborts = 0;
    unsigned int seq = 0;              // solve sequence number: part of the key of the XCD-local exchange's entries
    int xcd_misplaced = 0;             // solves whose XCD-local attempt found its workgroups on more than one XCD
    int cooldown = 0;                  // solves left that skip the persistent kernels after one of them gave up waiting
    int32_t* sticky = nullptr;         // device [1]: a solve enqueued by launch_eig_blocked_nosync failed since the last reset
    SsBufs ss{};                       // subspace solver in front of the exact one (mpst_eig_subspace.inl); ss.Mw == nullptr: off
    int32_t* host_st = nullptr;        // pinned [4]: copy of ss.st
    BlockedEig* rr = nullptr;          // complex: workspace of the Hermitian Rayleigh-Ritz solve (order 2 pc, pair mode, raw)
};
static int coop_threads() {
    static const int nt = [] { const char* e = getenv("MPST_BT_COOP_T"); return e ? atoi(e) : 512; }();
    return nt == 256 ? 256 : 512;
}
static int coop_grid(int ncap) {
    static const int gmax = [] { const char* e = getenv("MPST_BT_COOP_G"); return e ? std::max(1, atoi(e)) : 0; }();
    // measured: 512 threads and one row per wave (8 rows per workgroup): 37 workgroups at n = 296, 64 at n = 512 - fewer
    // workgroups make the barrier cheaper, more threads keep the row work off the critical path
    return std::max(1, std::min(gmax ? gmax : 128, (ncap + 7) / 8));
}
static size_t coop_lds(int ncap) {
    const int G = coop_grid(ncap);
    return (size_t)(3 + (ncap + G - 1) / G) * ncap * sizeof(double);
}
// XCD-local variant: one workgroup per CU of one XCD (32), every 8th workgroup of the launch
constexpr int XCD_G = 32, XCD_STRIDE = 8;
static size_t xcd_lds(int ncap) { return (size_t)(3 + (ncap + XCD_G - 1) / XCD_G) * ncap * sizeof(double); }
static bool xcd_usable(int ncap) {
    static const bool off = getenv("MPST_BT_NO_XCD") != nullptr;
    return !off && coop_threads() == 512 && xcd_lds(ncap) <= 150 * 1024;
}

static size_t bt_vec_lds() { return (size_t)(6 * BT_NMAX + 16) * sizeof(double); }
// native Hermitian reduction (pair mode): 8 complex rows per workgroup; MPST_BT_NO_CNATIVE=1 keeps the embedded real reduction
static int coopc_grid(int ncap) { return std::max(1, std::min(128, (ncap / 2 + 7) / 8)); }
static size_t coopc_lds(int ncap) {
    const int nc = ncap / 2, G = coopc_grid(ncap);
    return (size_t)(3 + (nc + G - 1) / G) * nc * sizeof(double2);
}
// XCD-local variants: 32 workgroups on one XCD, ceil(n / 32) complex rows each
static size_t coopc_xcd_lds(int ncap) {
    const int nc = ncap / 2;
    return (size_t)(3 + (nc + XCD_G - 1) / XCD_G) * nc * sizeof(double2);
}
// 0: across the XCDs; 1: one XCD, tagged entries (default where the rows fit); 3: one XCD, counting barrier.  Same box, c64, per bond
// at n_c = 256 / 512: across the XCDs 1.23 / 3.41 ms, counted 1.10 / 3.32, tagged 1.01 / 3.07 (MPST_BT_C_XCD=0|counted|tagged)
static int cnative_xcd_mode(int ncap) {
    static const int pick = [] {
        if (getenv("MPST_BT_NO_XCD") != nullptr) return 0;
        const char* e = getenv("MPST_BT_C_XCD");
        if (e && !strcmp(e, "0")) return 0;
        return (e && !strcmp(e, "counted")) ? 3 : 1;
    }();
    return coopc_xcd_lds(ncap) <= 156 * 1024 ? pick : 0;
}
static bool cnative_xcd_usable(int ncap) { return cnative_xcd_mode(ncap) != 0; }
static bool cnative_usable(const View& v, int ncap) {
    static const bool off = getenv("MPST_BT_NO_CNATIVE") != nullptr;
    return !off && v.zw == 2 && (ncap & 1) == 0 && ncap / 2 <= BTC_NMAX && coopc_lds(ncap) <= 150 * 1024;
}

int blocked_eig_create(BlockedEig** out, int ncap, std::string* err) {
    BlockedEig* e = new BlockedEig();
    e->b.ncap = ncap;
    e->b.use_tail = getenv("MPST_BT_NO_TAIL") == nullptr ? 1 : 0;
    auto al = [&](double** p, size_t n) { return hipMalloc((void**)p, n * sizeof(double)) == hipSuccess; };
    const size_t n1 = ncap, n2 = (size_t)ncap * ncap;
    bool ok = al(&e->b.A, n2) && al(&e->b.D, (size_t)CAP_LIMIT * CAP_LIMIT) && al(&e->b.Y, 2 * n1) && al(&e->b.Vall, n2) &&
              al(&e->b.dd, n1) && al(&e->b.ee, n1) && al(&e->b.tau, n1) && al(&e->b.Z, (size_t)CAP_LIMIT * n1) && al(&e->b.lam, CAP_LIMIT) &&
              al(&e->b.res, CAP_LIMIT) && al(&e->b.tailG, (size_t)BT_TAIL * BT_TAIL) && al(&e->b.Tfac, (size_t)((ncap + 15) / 16) * 256) && hipMalloc((void**)&e->b.flag, sizeof(int32_t)) == hipSuccess && hipMalloc((void**)&e->b.ctl, 4 * sizeof(int32_t)) == hipSuccess && hipMalloc((void**)&e->sticky, sizeof(int32_t)) == hipSuccess &&
              hipHostMalloc((void**)&e->host_flag, 2 * sizeof(int32_t)) == hipSuccess && al(&e->cp.ybuf, 4 * n1) && al(&e->cp.rowbuf, 4 * n1) &&
              hipMalloc((void**)&e->cp.counter, 16) == hipSuccess;
    if (ok) {               // one 16-byte control block, cleared by one memset per solve: counter | abort flag | roll call
        static std::atomic<int> ordinal{0};
        e->cp.abort_flag = (int32_t*)(e->cp.counter + 1);
        e->cp.roll = (unsigned long long*)(e->cp.counter + 2);
        e->cp.xsel = ordinal.fetch_add(1) & 7;
    }
    if (ok) ok = hipMemset(e->b.Vall, 0, n2 * sizeof(double)) == hipSuccess && hipMemset(e->b.ctl, 0, 4 * sizeof(int32_t)) == hipSuccess && hipMemset(e->sticky, 0, sizeof(int32_t)) == hipSuccess && hipMemset(e->b.Y, 0, 2 * n1 * sizeof(double)) == hipSuccess;
    if (ok) ok = hipFuncSetAttribute((const void*)k_bt_vec, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bt_vec_lds()) == hipSuccess &&
                 hipFuncSetAttribute((const void*)k_bt_coop<256, SC_AGENT>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) == hipSuccess &&
                 hipFuncSetAttribute((const void*)k_bt_coop<512, SC_AGENT>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) == hipSuccess &&
                 hipFuncSetAttribute((const void*)k_bt_coop<512, SC_XCD>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) == hipSuccess &&
                 hipFuncSetAttribute((const void*)k_bt_coop_c<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) == hipSuccess &&
                 hipFuncSetAttribute((const void*)k_bt_coop_c<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) == hipSuccess &&
                 hipFuncSetAttribute((const void*)k_bt_coop_c<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) == hipSuccess &&
                 hipFuncSetAttribute((const void*)k_bt_coop_c<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024) == hipSuccess;
    if (!ok) {
        if (err) *err = "allocation of the blocked eigensolver's workspace failed";
        blocked_eig_destroy(e);
        return MPST_ERR_NOMEM;
    }
    *out = e;
    return 0;
}
void blocked_eig_destroy(BlockedEig* e) {
    if (!e) return;
    double* ps[] = {e->b.A, e->b.D, e->b.Y, e->b.Vall, e->b.dd, e->b.ee, e->b.tau, e->b.Z, e->b.lam, e->b.res, e->b.Tfac, e->b.tailG};
    for (double* p : ps)
        if (p) (void)hipFree(p);
    if (e->b.flag) (void)hipFree(e->b.flag);
    if (e->b.ctl) (void)hipFree(e->b.ctl);
    if (e->sticky) (void)hipFree(e->sticky);
    if (e->host_flag) (void)hipHostFree(e->host_flag);
    if (e->cp.ybuf) (void)hipFree(e->cp.ybuf);
    if (e->cp.rowbuf) (void)hipFree(e->cp.rowbuf);
    if (e->cp.counter) (void)hipFree(e->cp.counter);
    {
        SsBufs& q = e->ss;
        double* qs[] = {q.Mw, q.Lb[0], q.Lb[1], q.Rb[0], q.Rb[1], q.Sp, q.Tm, q.H, q.lamH, q.WH, q.wsH, q.part};
        for (double* p : qs)
            if (p) (void)hipFree(p);
        if (q.infoH) (void)hipFree(q.infoH);
        if (q.st) (void)hipFree(q.st);
        if (e->host_st) (void)hipHostFree(e->host_st);
        if (e->rr) blocked_eig_destroy(e->rr);
    }
    delete e;
}

// Switch the subspace solver on for this workspace: mcap rows (C * d * cap), kcap = chi_max.  Off (and harmless) when the shape
// leaves no room for it (block wider than SS_PMAX, or the matrix not at least twice as wide as the block) or MPST_NO_SUBSPACE is set.
int blocked_eig_enable_subspace(BlockedEig* e, int mcap, int kcap, int C, int cx, std::string* err) {
    if (!e || e->ss.Mw) return 0;
    static const bool off = getenv("MPST_NO_SUBSPACE") != nullptr;
    static const bool off_c = getenv("MPST_NO_SUBSPACE_C") != nullptr;
    // complex: the workspace of the exact solver holds the 2n embedding; counts below are those of the complex matrix
    const int ncap = cx ? e->b.ncap / 2 : e->b.ncap;
    if (cx) mcap /= 2;
    if (off || (cx && off_c) || kcap > 64 || C * kcap > SS_PMAX) return 0;      // the block holds rank(M0) <= C chi columns
    const int pc = std::min(SS_PMAX, (C * kcap + SS_EXTRA + 15) & ~15);
    if (ncap < 2 * pc || (cx ? 2 : 1) * ncap <= MAX_DIM) return 0;
    if (cx && pc > 96) return 0;                                                 // the complex elimination keeps a 3 x 12 register tile
    SsBufs& q = e->ss;
    q.pc = pc;
    q.capped = (C * kcap + SS_EXTRA > pc) ? 1 : 0;
    q.cx = cx ? 1 : 0;
    q.dbg = getenv("MPST_SS_DBG") ? atoi(getenv("MPST_SS_DBG")) : 0;
    q.mcap = mcap;
    q.ncap = ncap;
    const size_t z = cx ? 2 : 1;          // doubles per element
    auto al = [&](double** p, size_t n) { return hipMalloc((void**)p, n * sizeof(double)) == hipSuccess && hipMemset(*p, 0, n * sizeof(double)) == hipSuccess; };
    const size_t tn = (size_t)(z * ncap + 15) / 16;
    bool ok = al(&q.Mw, z * mcap * ncap) && al(&q.Lb[0], z * mcap * pc) && al(&q.Lb[1], z * mcap * pc) && al(&q.Rb[0], z * ncap * pc) &&
              al(&q.Rb[1], z * ncap * pc) && al(&q.Sp, z * SS_KS * pc * pc) && al(&q.Tm, z * pc * pc) && al(&q.H, z * z * pc * pc) &&
              al(&q.lamH, SS_PMAX + 16) && al(&q.WH, z * z * pc * pc) && al(&q.wsH, eig_workspace_doubles()) && al(&q.part, tn * CAP_LIMIT + tn + 16) &&
              hipMalloc((void**)&q.infoH, sizeof(int32_t)) == hipSuccess && hipMalloc((void**)&q.st, 4 * sizeof(int32_t)) == hipSuccess &&
              hipMemset(q.st, 0, 4 * sizeof(int32_t)) == hipSuccess && hipHostMalloc((void**)&e->host_st, 4 * sizeof(int32_t)) == hipSuccess;
    if (ok) ok = hipFuncSetAttribute((const void*)k_ss_cholb, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) == hipSuccess;
    if (ok && cx) {
        std::string e2;
        ok = blocked_eig_create(&e->rr, 2 * pc, &e2) == 0;
        if (ok) q.rrflag = e->rr->b.flag;
    }
    if (!ok) {
        if (err) *err = "allocation of the subspace eigensolver's workspace failed";
        return MPST_ERR_NOMEM;
    }
    return 0;
}
bool blocked_eig_subspace_on(const BlockedEig* e) { return e && e->ss.Mw != nullptr; }
// bonds the subspace solver attempted / whose result was accepted since creation (synchronises the stream)
int blocked_eig_subspace_counts(BlockedEig* e, hipStream_t s, int32_t* attempted, int32_t* accepted) {
    *attempted = *accepted = 0;
    if (!blocked_eig_subspace_on(e)) return 0;
    if (hipMemcpyAsync(e->host_st, e->ss.st, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, s) != hipSuccess) return MPST_ERR_DEVICE;
    if (hipStreamSynchronize(s) != hipSuccess) return MPST_ERR_DEVICE;
    *accepted = e->host_st[2];
    *attempted = e->host_st[3];
    return 0;
}

// The subspace phase of a solve: ~30 launches, all leave at once on a bond that is not attempted.  Afterwards ss.st[0] says whether
// E / lam / chi are published.
static void enqueue_after_tridiag(const View& v, int lid, int going_left, const double* rawG, int rawn, double* rawlam, double* rawE,
                                  int32_t* rawinfo, const BtBufs& b, hipStream_t s);
static int cnative_xcd_mode(int ncap);
static size_t coopc_xcd_lds(int ncap);
static size_t coopc_lds(int ncap);
static int coopc_grid(int ncap);

// complex element types: the same sequence with the complex kernels; the Hermitian Rayleigh-Ritz problem goes through the native
// pair-mode chain (k_bt_coop_c ...) on the small workspace e->rr, raw mode, gated by st[1]
static void enqueue_subspace_c(const View& v, int lid, int going_left, BlockedEig* e, hipStream_t s) {
    const SsBufs& q = e->ss;
    const int pc = q.pc, tp = pc / 16;
    const int tm = (q.mcap + 15) / 16, tn = (q.ncap + 15) / 16;
    const int kmax = std::min(v.chi_max / 2, 64), tk = (kmax + 15) / 16;
    double2 *L0 = (double2*)q.Lb[0], *L1 = (double2*)q.Lb[1], *R0 = (double2*)q.Rb[0], *R1 = (double2*)q.Rb[1];
    const int short_start = v.ss_f32 ? 1 : 0;
    hipLaunchKernelGGL(k_ss_load_c, dim3(256), dim3(256), 0, s, v, lid, going_left, q, short_start);
    auto orth = [&](double2* raw, double2* out, int left) {
        hipLaunchKernelGGL(k_ss_gram_c, dim3(tp * tp * SS_KS), dim3(256), 0, s, v, lid, going_left, q, (const double2*)raw, left);
        hipLaunchKernelGGL(k_ss_chol_c<0>, dim3(1), dim3(CH_T), 0, s, v, lid, going_left, q);
        hipLaunchKernelGGL(k_ss_apply_c, dim3(((left ? tm : tn) * tp + 3) / 4), dim3(256), 0, s, v, lid, going_left, q, (const double2*)raw, out, left);
    };
    if (!short_start) {
        hipLaunchKernelGGL(k_ss_mm_c<1>, dim3(tn * tp), dim3(256), 0, s, v, lid, going_left, q, (const double2*)L0, R0, 0);
        orth(R0, R1, 0);
    }
    hipLaunchKernelGGL(k_ss_mm_c<0>, dim3(tm * tp), dim3(256), 0, s, v, lid, going_left, q, (const double2*)R1, L0, 0);
    orth(L0, L1, 1);
    hipLaunchKernelGGL(k_ss_mm_c<1>, dim3(tn * tp), dim3(256), 0, s, v, lid, going_left, q, (const double2*)L1, R0, 0);
    orth(R0, R1, 0);
    hipLaunchKernelGGL(k_ss_mm_c<0>, dim3(tm * tp), dim3(256), 0, s, v, lid, going_left, q, (const double2*)R1, L0, 0);
    orth(L0, L1, 1);
    hipLaunchKernelGGL(k_ss_mm_c<1>, dim3(tn * tp), dim3(256), 0, s, v, lid, going_left, q, (const double2*)L1, R0, 0);
    hipLaunchKernelGGL(k_ss_gram_c, dim3(tp * tp * SS_KS), dim3(256), 0, s, v, lid, going_left, q, (const double2*)R0, 0);
    hipLaunchKernelGGL(k_ss_chol_c<1>, dim3((pc * pc + CH_T - 1) / CH_T), dim3(CH_T), 0, s, v, lid, going_left, q);
    {   // Hermitian Rayleigh-Ritz: eigenpairs of the 2 pc x 2 pc embedding in pair mode -> lamH (each value twice), WH (vector k in column 2k)
        BlockedEig* r = e->rr;
        const int rn = 2 * pc;
        (void)hipMemsetAsync(r->cp.counter, 0, 16, s);
        BtBufs bt = r->b;
        bt.abort = r->cp.abort_flag;
        bt.sticky = nullptr;
        bt.skip = nullptr;
        bt.gate = q.st + 1;
        bt.ss = 0;
        bt.cnative = 1;
        const int xm = cnative_xcd_mode(rn);
        // MPST_RR_SOLO=1: ONE workgroup with the exchange through LDS (XMODE 4).  Measured and rejected as the default: 95 steps of
        // barriers instead of L2 round trips, but the row updates of a 96 x 96 complex matrix on one CU take 515 us against the 223 us
        // of 32 workgroups (c64 at configs[4]'s shape: 1.30 against 1.53 sweeps/s)
        static const bool solo = getenv("MPST_RR_SOLO") != nullptr;
        const size_t solo_lds = ((size_t)(3 + pc) * pc + 4 * (size_t)pc) * sizeof(double2);
        if (solo && solo_lds <= 158 * 1024)
            hipLaunchKernelGGL(k_bt_coop_c<4>, dim3(1), dim3(BTC_NT), solo_lds, s, v, 0, 0, bt, r->cp, 1, 0u, (const double*)q.H, rn);
        else if (xm == 3)
            hipLaunchKernelGGL(k_bt_coop_c<3>, dim3(XCD_G * XCD_STRIDE), dim3(BTC_NT), coopc_xcd_lds(rn), s, v, 0, 0, bt, r->cp, XCD_STRIDE, 0u, (const double*)q.H, rn);
        else if (xm == 1)
            hipLaunchKernelGGL(k_bt_coop_c<1>, dim3(XCD_G * XCD_STRIDE), dim3(BTC_NT), coopc_xcd_lds(rn), s, v, 0, 0, bt, r->cp, XCD_STRIDE, ++r->seq, (const double*)q.H, rn);
        else
            hipLaunchKernelGGL(k_bt_coop_c<0>, dim3(coopc_grid(rn)), dim3(BTC_NT), coopc_lds(rn), s, v, 0, 0, bt, r->cp, 1, 0u, (const double*)q.H, rn);
        enqueue_after_tridiag(v, 0, 0, q.H, rn, q.lamH, q.WH, q.infoH, bt, s);
    }
    BtBufs b = e->b;
    b.abort = nullptr;
    b.sticky = nullptr;
    b.skip = nullptr;
    b.gate = q.st + 1;
    b.ss = 1;
    b.cnative = 0;
    hipLaunchKernelGGL(k_ss_ritz_c, dim3((tn * tk + 3) / 4), dim3(256), 0, s, v, lid, going_left, q, b, (const double2*)R0, 0);
    const int tne = (2 * q.ncap + 15) / 16;
    hipLaunchKernelGGL(k_ss_resid, dim3(tne * tk), dim3(256), 0, s, v, lid, going_left, q, b);
    hipLaunchKernelGGL(k_ss_collect, dim3(1), dim3(1024), 0, s, v, lid, going_left, q, b);
    const int ncap = b.ncap, tkk = (std::min(v.chi_max, CAP_LIMIT) + 15) / 16, tnn = (ncap + 15) / 16;
    // verification / Loewdin rounds: one for fp32 tensors (k_bt_decide fails the bond on a deviation from orthonormality above 1e-8), two for fp64
    for (int second = 0; second < (v.ss_f32 ? 1 : 2); ++second) {
        if (second) hipLaunchKernelGGL(k_bt_copyback, dim3(64), dim3(BT_T), 0, s, v, lid, going_left, 0, b, (const double*)nullptr);
        hipLaunchKernelGGL(k_bt_gram, dim3(tkk * tkk), dim3(BT_T), 0, s, v, lid, going_left, 0, b, second);
        hipLaunchKernelGGL(k_bt_decide, dim3(1), dim3(512), 0, s, v, lid, going_left, (const double*)nullptr, 0, b, (double*)nullptr, (int32_t*)nullptr, second);
        hipLaunchKernelGGL(k_bt_polish, dim3(std::max(1, std::min(256, (tnn * tkk + 3) / 4))), dim3(BT_T), 0, s, v, lid, going_left, 0, b, (double*)nullptr, second);
    }
    hipLaunchKernelGGL(k_ss_verdict, dim3(1), dim3(64), 0, s, q, b);
}

static void enqueue_subspace(const View& v, int lid, int going_left, BlockedEig* e, hipStream_t s) {
    if (e->ss.cx) {
        enqueue_subspace_c(v, lid, going_left, e, s);
        return;
    }
    const SsBufs& q = e->ss;
    const int pc = q.pc, tp = pc / 16;
    const int tm = (q.mcap + 15) / 16, tn = (q.ncap + 15) / 16;
    const size_t chol_lds = 0;
    const int kmax = std::min(v.chi_max, 64), tk = (kmax + 15) / 16;
    const int short_start = v.ss_f32 ? 1 : 0;
    hipLaunchKernelGGL(k_ss_load, dim3(256), dim3(256), 0, s, v, lid, going_left, q, short_start);
    static const bool pivotwise = getenv("MPST_SS_PIVOT") != nullptr;     // the pivot-by-pivot elimination instead of the blocked factorisation
    const size_t cb_lds = ((size_t)pc * (pc + CB_LDPAD) + pc + (size_t)(pc / 16) * 256) * sizeof(double);
    auto orth = [&](double* raw, double* out, int left) {          // out = cholqr(raw)
        hipLaunchKernelGGL(k_ss_gram, dim3(tp * tp * SS_KS), dim3(256), 0, s, v, lid, going_left, q, (const double*)raw, left);
        if (!pivotwise) hipLaunchKernelGGL(k_ss_cholb, dim3(1), dim3(CB_T), cb_lds, s, v, lid, going_left, q);
        else if (pc <= 96) hipLaunchKernelGGL(k_ss_chol<2>, dim3(1), dim3(CH_T), chol_lds, s, v, lid, going_left, q);
        else hipLaunchKernelGGL(k_ss_chol<0>, dim3(1), dim3(CH_T), chol_lds, s, v, lid, going_left, q);
        hipLaunchKernelGGL(k_ss_apply, dim3(((left ? tm : tn) * tp + 3) / 4), dim3(256), 0, s, v, lid, going_left, q, (const double*)raw, out, left);
    };
    // X = cholqr(M^T Omega)  (fp32: X = Omega / sqrt(n), written by k_ss_load)
    if (!short_start) {
        hipLaunchKernelGGL(k_ss_mm<1>, dim3(tn * tp), dim3(256), 0, s, v, lid, going_left, q, (const double*)q.Lb[0], q.Rb[0]);
        orth(q.Rb[0], q.Rb[1], 0);
    }
    // Q = cholqr(M X); X = cholqr(M^T Q); Q = cholqr(M X)   (twice where the capacity of the Rayleigh-Ritz solver leaves the block no
    // oversampling - two classes at chi_max = 64: rank(M0) = 128 = the block -: without it 15-20 % of such bonds end at 1e-9 ... 1e-8
    // and go back to the exact solver)
    for (int round = 0; round < (q.capped ? 2 : 1); ++round) {
        hipLaunchKernelGGL(k_ss_mm<0>, dim3(tm * tp), dim3(256), 0, s, v, lid, going_left, q, (const double*)q.Rb[1], q.Lb[0]);
        orth(q.Lb[0], q.Lb[1], 1);
        hipLaunchKernelGGL(k_ss_mm<1>, dim3(tn * tp), dim3(256), 0, s, v, lid, going_left, q, (const double*)q.Lb[1], q.Rb[0]);
        orth(q.Rb[0], q.Rb[1], 0);
    }
    hipLaunchKernelGGL(k_ss_mm<0>, dim3(tm * tp), dim3(256), 0, s, v, lid, going_left, q, (const double*)q.Rb[1], q.Lb[0]);
    orth(q.Lb[0], q.Lb[1], 1);
    // Z = M^T Q; H = Z^T Z; Rayleigh-Ritz
    hipLaunchKernelGGL(k_ss_mm<1>, dim3(tn * tp), dim3(256), 0, s, v, lid, going_left, q, (const double*)q.Lb[1], q.Rb[0]);
    hipLaunchKernelGGL(k_ss_gram, dim3(tp * tp * SS_KS), dim3(256), 0, s, v, lid, going_left, q, (const double*)q.Rb[0], 0);
    hipLaunchKernelGGL(k_ss_chol<1>, dim3((pc * pc + CH_T - 1) / CH_T), dim3(CH_T), chol_lds, s, v, lid, going_left, q);
    launch_eig_raw_gated(q.H, pc, q.lamH, q.WH, q.infoH, q.wsH, q.st + 1, s);
    BtBufs b = e->b;
    b.abort = nullptr;
    b.sticky = nullptr;
    b.skip = nullptr;
    b.gate = q.st + 1;
    b.ss = 1;
    b.cnative = 0;
    hipLaunchKernelGGL(k_ss_ritz, dim3((tn * tk + 3) / 4), dim3(256), 0, s, v, lid, going_left, q, b, (const double*)q.Rb[0]);
    hipLaunchKernelGGL(k_ss_resid, dim3(tn * tk), dim3(256), 0, s, v, lid, going_left, q, b);
    hipLaunchKernelGGL(k_ss_collect, dim3(1), dim3(1024), 0, s, v, lid, going_left, q, b);
    const int ncap = b.ncap, tkk = (std::min(v.chi_max, CAP_LIMIT) + 15) / 16, tnn = (ncap + 15) / 16;
    // verification / Loewdin rounds: one for fp32 tensors (k_bt_decide fails the bond on a deviation from orthonormality above 1e-8), two for fp64
    for (int second = 0; second < (v.ss_f32 ? 1 : 2); ++second) {
        if (second) hipLaunchKernelGGL(k_bt_copyback, dim3(64), dim3(BT_T), 0, s, v, lid, going_left, 0, b, (const double*)nullptr);
        hipLaunchKernelGGL(k_bt_gram, dim3(tkk * tkk), dim3(BT_T), 0, s, v, lid, going_left, 0, b, second);
        hipLaunchKernelGGL(k_bt_decide, dim3(1), dim3(512), 0, s, v, lid, going_left, (const double*)nullptr, 0, b, (double*)nullptr, (int32_t*)nullptr, second);
        hipLaunchKernelGGL(k_bt_polish, dim3(std::max(1, std::min(256, (tnn * tkk + 3) / 4))), dim3(BT_T), 0, s, v, lid, going_left, 0, b, (double*)nullptr, second);
    }
    hipLaunchKernelGGL(k_ss_verdict, dim3(1), dim3(64), 0, s, q, b);
}

// the launches that follow the tridiagonalisation: eigenvectors, verification, publication
static void enqueue_after_tridiag(const View& v, int lid, int going_left, const double* rawG, int rawn, double* rawlam, double* rawE,
                                  int32_t* rawinfo, const BtBufs& b, hipStream_t s) {
    const int ncap = rawn > 0 ? rawn : b.ncap;
    const int kmax = rawn > 0 ? std::min(rawn, CAP_LIMIT) : std::min(v.chi_max, CAP_LIMIT);
    hipLaunchKernelGGL(k_bt_larft, dim3((ncap + BT_NB - 1) / BT_NB), dim3(BT_T), 0, s, v, lid, going_left, rawn, b);
    hipLaunchKernelGGL(k_bt_vec, dim3(kmax), dim3(BT_T), bt_vec_lds(), s, v, lid, going_left, rawn, b);
    if (b.cnative) hipLaunchKernelGGL(k_bt_back_c, dim3((kmax / 2 + 3) / 4), dim3(256), 0, s, v, lid, going_left, rawn, b);
    const int tk = (kmax + 15) / 16, tn = (ncap + 15) / 16;
    for (int second = 0; second < 2; ++second) {
        if (second) hipLaunchKernelGGL(k_bt_copyback, dim3(64), dim3(BT_T), 0, s, v, lid, going_left, rawn, b, (const double*)rawE);
        hipLaunchKernelGGL(k_bt_gram, dim3(tk * tk), dim3(BT_T), 0, s, v, lid, going_left, rawn, b, second);
        hipLaunchKernelGGL(k_bt_decide, dim3(1), dim3(512), 0, s, v, lid, going_left, rawG, rawn, b, rawlam, rawinfo, second);
        hipLaunchKernelGGL(k_bt_polish, dim3(std::max(1, std::min(256, (tn * tk + 3) / 4))), dim3(BT_T), 0, s, v, lid, going_left, rawn, b, rawE, second);
    }
}

// enqueue the whole solve; returns 1 if the on-device verification asks for the library fallback, 0 if E / lam / chi are
// published, < 0 on a runtime error.  Synchronises the stream once (the verdict is read by the host).  Inside a sweep the
// tridiagonalisation is the persistent kernel; if its workgroups could not all become resident within its patience it
// gives up cleanly and the same bond is redone one launch per step.
int launch_eig_blocked(const View& v, int lid, int going_left, const double* rawG, int rawn, double* rawlam, double* rawE,
                       int32_t* rawinfo, BlockedEig* e, hipStream_t s) {
    const BtBufs& b = e->b;
    const int ncap = rawn > 0 ? rawn : b.ncap;
    static const bool no_coop = getenv("MPST_BT_NO_COOP") != nullptr;
    if (rawn == 0 && blocked_eig_subspace_on(e)) {
        // the subspace solver first; its verdict is read here (this path synchronises per bond anyway), so the exact solve is
        // enqueued only when it is needed
        enqueue_subspace(v, lid, going_left, e, s);
        if (hipMemcpyAsync(e->host_st, e->ss.st, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, s) != hipSuccess) return MPST_ERR_DEVICE;
        if (hipStreamSynchronize(s) != hipSuccess) return MPST_ERR_DEVICE;
        if (hipGetLastError() != hipSuccess) return MPST_ERR_DEVICE;
        static const bool dbg = getenv("MPST_SS_DEBUG") != nullptr;
        if (dbg && e->host_st[1]) {
            double res[CAP_LIMIT], lam[CAP_LIMIT], D[CAP_LIMIT];
            int32_t ctl[4], flag, info;
            (void)hipMemcpy(res, b.res, sizeof res, hipMemcpyDeviceToHost);
            (void)hipMemcpy(lam, b.lam, sizeof lam, hipMemcpyDeviceToHost);
            (void)hipMemcpy(D, b.D, sizeof D, hipMemcpyDeviceToHost);
            (void)hipMemcpy(ctl, b.ctl, sizeof ctl, hipMemcpyDeviceToHost);
            (void)hipMemcpy(&flag, b.flag, sizeof flag, hipMemcpyDeviceToHost);
            (void)hipMemcpy(&info, e->ss.infoH, sizeof info, hipMemcpyDeviceToHost);
            double r2 = 0, rmax = 0; int bad = -1;
            for (int i = 0; i < ctl[0] && i < CAP_LIMIT; ++i) { r2 += res[i] * res[i]; if (res[i] > rmax) { rmax = res[i]; bad = i; } }
            double stamp[8];
            (void)hipMemcpy(stamp, e->ss.lamH + SS_PMAX, sizeof stamp, hipMemcpyDeviceToHost);
            fprintf(stderr, "[ss] chol phases: load %.2f us, elimination %.2f us | blocked: load %.2f, updates %.2f, diagonal blocks %.2f, panels %.2f, inverse %.2f, store %.2f us\n",
                    stamp[0] * 0.01, stamp[1] * 0.01, stamp[2] * 0.01, stamp[3] * 0.01, stamp[4] * 0.01, stamp[5] * 0.01, stamp[6] * 0.01, stamp[7] * 0.01);
            fprintf(stderr, "[ss] lid %d gl %d st %d %d ctl %d %d %d %d flag %d eiginfo %d est %.3e rmax %.3e at %d lam0 %.3e lam[k-1] %.3e D00 %.3e D01 %.3e\n", lid, going_left,
                    e->host_st[0], e->host_st[1], ctl[0], ctl[1], ctl[2], ctl[3], flag, info, sqrt(r2), rmax, bad, lam[0], lam[ctl[0] > 0 ? ctl[0] - 1 : 0], D[0], D[1]);
        }
        if (e->host_st[0]) return 0;
    }
    // mode 2: persistent kernel confined to one XCD; 1: persistent kernel across the XCDs; 0: one launch per step
    int mode = (rawn == 0 && !no_coop) ? (xcd_usable(ncap) ? 2 : 1) : 0;
    // pair mode: the Hermitian problem itself (k_bt_coop_c) - 4: confined to one XCD, 3: across the XCDs
    if (rawn == 0 && !no_coop && cnative_usable(v, ncap)) mode = cnative_xcd_usable(ncap) ? 4 : 3;
    // after a persistent kernel ran out of patience (its workgroups could not all become resident: the GPU is shared with
    // other work) the next solves go straight to the launch-per-step path instead of each paying the wait again
    if (mode && e->cooldown > 0) {
        e->cooldown--;
        mode = 0;
    }
    for (int attempt = 0; attempt < 3; ++attempt) {
        if (mode) {
            if (hipMemsetAsync(e->cp.counter, 0, 16, s) != hipSuccess) return MPST_ERR_DEVICE;
            if (mode == 4 && cnative_xcd_mode(ncap) == 3)
                hipLaunchKernelGGL(k_bt_coop_c<3>, dim3(XCD_G * XCD_STRIDE), dim3(BTC_NT), coopc_xcd_lds(ncap), s, v, lid, going_left, b, e->cp, XCD_STRIDE, 0u, (const double*)nullptr, 0);
            else if (mode == 4)
                hipLaunchKernelGGL(k_bt_coop_c<1>, dim3(XCD_G * XCD_STRIDE), dim3(BTC_NT), coopc_xcd_lds(ncap), s, v, lid, going_left, b, e->cp, XCD_STRIDE, ++e->seq, (const double*)nullptr, 0);
            else if (mode == 3)
                hipLaunchKernelGGL(k_bt_coop_c<0>, dim3(coopc_grid(ncap)), dim3(BTC_NT), coopc_lds(ncap), s, v, lid, going_left, b, e->cp, 1, 0u, (const double*)nullptr, 0);
            else if (mode == 2)
                hipLaunchKernelGGL((k_bt_coop<512, SC_XCD>), dim3(XCD_G * XCD_STRIDE), dim3(512), xcd_lds(ncap), s, v, lid, going_left, b, e->cp, XCD_STRIDE, ++e->seq);
            else if (coop_threads() == 512)
                hipLaunchKernelGGL((k_bt_coop<512, SC_AGENT>), dim3(coop_grid(ncap)), dim3(512), coop_lds(ncap), s, v, lid, going_left, b, e->cp, 1, 0u);
            else
                hipLaunchKernelGGL((k_bt_coop<256, SC_AGENT>), dim3(coop_grid(ncap)), dim3(256), coop_lds(ncap), s, v, lid, going_left, b, e->cp, 1, 0u);
        } else {
            hipLaunchKernelGGL(k_bt_prep, dim3(256), dim3(BT_T), 0, s, v, lid, going_left, rawG, rawn, b);
            static const int bt_g = [] { const char* e = getenv("MPST_BT_G"); return e ? std::max(1, atoi(e)) : BT_G; }();
            for (int j = 0; j <= ncap - 2; ++j) {
                // fewer workgroups once the trailing matrix is small: the redundant prologue is paid per workgroup
                const int m = ncap - 1 - j;
                const int g = std::max(1, std::min(bt_g, (m + 3) / 4));
                hipLaunchKernelGGL(k_bt_step, dim3(g), dim3(BT_T), 0, s, v, lid, going_left, rawn, b, j);
            }
        }
        BtBufs bt = b;
        bt.abort = mode ? e->cp.abort_flag : nullptr;
        bt.cnative = mode >= 3 ? 1 : 0;
        if (b.use_tail && mode < 3) {
            // the last BT_TAIL steps on one CU (the kernels above stopped at step n - BT_TAIL; nothing to do for n <= BT_TAIL)
            if (!mode) hipLaunchKernelGGL(k_bt_tail_prep, dim3(32), dim3(BT_T), 0, s, v, lid, going_left, rawn, b);
            launch_eig_tail(v, lid, going_left, rawn, b.tailG, b.ncap, b.Vall, b.dd, b.ee, b.tau, bt.abort, bt.skip, s);
        }
        enqueue_after_tridiag(v, lid, going_left, rawG, rawn, rawlam, rawE, rawinfo, bt, s);
        e->host_flag[1] = 0;
        if (hipMemcpyAsync(e->host_flag, b.flag, sizeof(int32_t), hipMemcpyDeviceToHost, s) != hipSuccess) return MPST_ERR_DEVICE;
        if (mode && hipMemcpyAsync(e->host_flag + 1, e->cp.abort_flag, sizeof(int32_t), hipMemcpyDeviceToHost, s) != hipSuccess) return MPST_ERR_DEVICE;
        if (hipStreamSynchronize(s) != hipSuccess) return MPST_ERR_DEVICE;
        if (hipGetLastError() != hipSuccess) return MPST_ERR_DEVICE;
        if (!(mode && e->host_flag[1])) break;
        // the persistent kernel gave up: the same bond again - across the XCDs if only the placement was wrong (its
        // workgroups were all there), else one launch per step
        if (mode == 2 && e->host_flag[1] == ABORT_PLACEMENT) {
            e->xcd_misplaced++;
            mode = 1;
        } else if (mode == 4 && e->host_flag[1] == ABORT_PLACEMENT) {
            e->xcd_misplaced++;
            mode = 3;           // the same reduction with the cross-XCD exchange
        } else if (mode >= 3) {
            e->coop_aborts++;
            mode = 1;           // the embedded real reduction, across the XCDs
        } else {
            e->coop_aborts++;
            e->cooldown = 32;
            mode = 0;
        }
    }
    return *e->host_flag ? 1 : 0;
}
// The same solve without reading the verdict: for sweeps that check ONCE at their end whether any bond failed (and then
// redo the sweep bond by bond).  No host synchronisation, no device-to-host copies; a failed verification or a persistent
// kernel that gave up leaves its mark in e->sticky, and the bonds after it run on whatever the failed one left - the
// caller discards that sweep.  Returns 0 or a negative error.
int launch_eig_blocked_nosync(const View& v, int lid, int going_left, BlockedEig* e, hipStream_t s) {
    const BtBufs& b = e->b;
    const int ncap = b.ncap;
    static const bool no_coop = getenv("MPST_BT_NO_COOP") != nullptr;
    if (no_coop) return MPST_ERR_UNSUPPORTED;
    const bool ss = blocked_eig_subspace_on(e);
    if (ss) enqueue_subspace(v, lid, going_left, e, s);      // the exact solve below leaves at once where its result stands
    if (hipMemsetAsync(e->cp.counter, 0, 16, s) != hipSuccess) return MPST_ERR_DEVICE;
    BtBufs bt = b;
    bt.abort = e->cp.abort_flag;
    bt.sticky = e->sticky;
    bt.skip = ss ? e->ss.st : nullptr;
    const bool nat = cnative_usable(v, ncap);
    bt.cnative = nat ? 1 : 0;
    if (nat && cnative_xcd_mode(ncap) == 3)
        hipLaunchKernelGGL(k_bt_coop_c<3>, dim3(XCD_G * XCD_STRIDE), dim3(BTC_NT), coopc_xcd_lds(ncap), s, v, lid, going_left, bt, e->cp, XCD_STRIDE, 0u, (const double*)nullptr, 0);
    else if (nat && cnative_xcd_mode(ncap) == 1)
        hipLaunchKernelGGL(k_bt_coop_c<1>, dim3(XCD_G * XCD_STRIDE), dim3(BTC_NT), coopc_xcd_lds(ncap), s, v, lid, going_left, bt, e->cp, XCD_STRIDE, ++e->seq, (const double*)nullptr, 0);
    else if (nat)
        hipLaunchKernelGGL(k_bt_coop_c<0>, dim3(coopc_grid(ncap)), dim3(BTC_NT), coopc_lds(ncap), s, v, lid, going_left, bt, e->cp, 1, 0u, (const double*)nullptr, 0);
    else if (xcd_usable(ncap))
        hipLaunchKernelGGL((k_bt_coop<512, SC_XCD>), dim3(XCD_G * XCD_STRIDE), dim3(512), xcd_lds(ncap), s, v, lid, going_left, bt, e->cp, XCD_STRIDE, ++e->seq);
    else if (coop_threads() == 512)
        hipLaunchKernelGGL((k_bt_coop<512, SC_AGENT>), dim3(coop_grid(ncap)), dim3(512), coop_lds(ncap), s, v, lid, going_left, bt, e->cp, 1, 0u);
    else
        hipLaunchKernelGGL((k_bt_coop<256, SC_AGENT>), dim3(coop_grid(ncap)), dim3(256), coop_lds(ncap), s, v, lid, going_left, bt, e->cp, 1, 0u);
    if (b.use_tail && !nat) launch_eig_tail(v, lid, going_left, 0, b.tailG, b.ncap, b.Vall, b.dd, b.ee, b.tau, bt.abort, bt.skip, s);
    enqueue_after_tridiag(v, lid, going_left, nullptr, 0, nullptr, nullptr, nullptr, bt, s);
    return 0;
}
// 1 if a solve enqueued by launch_eig_blocked_nosync failed since the last call (synchronises the stream), 0 if none, < 0 on error
int blocked_eig_take_sticky(BlockedEig* e, hipStream_t s) {
    int32_t h = 0;
    if (hipMemcpyAsync(&h, e->sticky, sizeof h, hipMemcpyDeviceToHost, s) != hipSuccess) return MPST_ERR_DEVICE;
    if (hipStreamSynchronize(s) != hipSuccess) return MPST_ERR_DEVICE;
    if (h && hipMemsetAsync(e->sticky, 0, sizeof(int32_t), s) != hipSuccess) return MPST_ERR_DEVICE;
    return h ? 1 : 0;
}
void blocked_eig_force_sticky(BlockedEig* e, hipStream_t s) { (void)hipMemsetAsync(e->sticky, 1, sizeof(int32_t), s); }   // test hook

int blocked_eig_coop_aborts(const BlockedEig* e) { return e ? e->coop_aborts : 0; }
int blocked_eig_xcd_misplaced(const BlockedEig* e) { return e ? e->xcd_misplaced : 0; }

}  // namespace mpst
