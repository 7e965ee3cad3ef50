// Stage 2 of a two-stage tridiagonalisation, built to be timed against the engine's one-stage k_eig_tri
// (VERDICT r02 item 1b): symmetric band matrix (half bandwidth B = 2, 4, 8) -> tridiagonal by Householder bulge
// chasing, one workgroup, the matrix in LDS.  Sweep s annihilates column s below the first subdiagonal and chases
// the bulge to the end of the matrix in steps of B rows; only the first column of each bulge is annihilated, so the
// working band is 2B - 1 wide.  Sweep s + 1 may run step k as soon as sweep s has finished step k + 2 (stagger 3:
// lab/ubench/band_ref.py shows the result is bit-identical to the sequential order, and wrong with stagger 2),
// so up to n / (3B) ... 16 sweeps are in flight, one wave each; they synchronise through progress counters in LDS
// (DS operations of one wave are executed in order, the LDS unit serves the CU's waves from one queue, so a wave
// that has seen the counter sees the data written before it) - no workgroup barrier inside the reduction.
//
// One step: the 3 blocks it touches (left block B x B, diagonal block B x B, next block B x B) are spread over
// 3B lanes as B-vectors (a column of the left block / a row of the next block / a column of the diagonal block),
// so applying H = I - tau v v^T from the left / right / both sides is lane-local except for one group sum and the
// broadcast of w inside the B lanes of the diagonal block (DPP for B <= 4).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 lab/ubench/band2tri.hip -o lab/ubench/band2tri.bin
//   lab/ubench/band2tri.bin 128 200   -> gpurun_out/band2tri_T.txt, checked by lab/ubench/band_check.py
#include <hip/hip_runtime.h>
#include "../../mpstime.jl_amd/csrc/mpst_internal.h"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
using namespace mpst;
#define CK(x) do { auto e_ = (x); if (e_ != 0) { printf("fail %s -> %d line %d\n", #x, (int)e_, __LINE__); exit(1); } } while (0)

constexpr int B2T_WAVES = 16;
constexpr int B2T_NMAX = 128;
constexpr int B2T_DONE = 1 << 20;
#ifndef B2T_SLEEP
#define B2T_SLEEP 2
#endif

template <int B>
__device__ __forceinline__ double group_sum(double x) {
    if constexpr (B == 2) return x + dpp_mov<DPP_XOR1>(x);
    else if constexpr (B == 4) return sum4(x);
    else return sum8(x);
}
template <int B, int E>
__device__ __forceinline__ double group_bcast(double x) {
    if constexpr (B == 2) return dpp_mov<(E == 0 ? 0xA0 : 0xF5)>(x);
    else if constexpr (B == 4) return dpp_mov<E * 0x55>(x);
    else return __shfl(x, E, 8);
}

template <int B>
__global__ __launch_bounds__(B2T_WAVES * 64) void k_band2tri(const double* __restrict__ band, int n, double* __restrict__ de,
                                                            unsigned long long* stamps) {
    constexpr int LD = 2 * B;
    // Ab[j][d] = A[j + d][j], d = 0 .. 2B-1; columns n .. n + 2B hold zeros so that no step needs a bounds test
    __shared__ double Ab[(B2T_NMAX + 2 * B + 1) * LD];
    __shared__ int prog[B2T_NMAX];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (stamps && tid == 0) {
        stamps[0] = __builtin_amdgcn_s_memrealtime();
        stamps[6] = __builtin_readcyclecounter();
    }
    for (int i = tid; i < (B2T_NMAX + 2 * B + 1) * LD; i += B2T_WAVES * 64) {
        const int j = i / LD, d = i % LD;
        Ab[i] = (j < n && d <= B && j + d < n) ? band[j * (B + 1) + d] : 0.0;
    }
    if (tid < B2T_NMAX) prog[tid] = 0;
    __syncthreads();
    const int role = lane / B, idx = lane % B;            // 0: left block column, 1: next block row, 2: diagonal block column
    // LDS offsets of the lane's B elements relative to column r0 of the band storage (steps k >= 1)
    int ce[B];
#pragma unroll
    for (int e = 0; e < B; ++e) {
        if (role == 0) ce[e] = (idx - B) * LD + (B - idx + e);                // A[r0+e][r0-B+idx]
        else if (role == 1) ce[e] = e * LD + (B + idx - e);                  // A[r0+B+idx][r0+e]
        else ce[e] = e >= idx ? idx * LD + (e - idx) : e * LD + (idx - e);   // A[r0+e][r0+idx] (either triangle)
    }
    volatile int* vprog = prog;
    const bool lane_used = role < 3;
    // One step on the B rows starting at r0.  The lane's vector y and the defining column x (read by every lane, so the
    // reflector is formed lane-locally and nothing has to be broadcast) are requested right behind the progress
    // counter of the previous sweep: the LDS unit serves a wave's requests in order, so if the counter read first
    // already shows the needed value the data behind it is valid, otherwise the whole group is asked again.
    auto step = [&](const int s, const int k, const int r0, const int (&off)[B], const int offx0, const bool on) {
        const double* base = Ab + r0 * LD;
        double y[B], x[B];
#ifdef B2T_STEPPROF
        const bool pr = stamps && lane == 0 && k == 5 && (s == 0 || s == 40);
        const int ps = s == 0 ? 16 : 24;
#define SP(j) do { if (pr) stamps[ps + (j)] = __builtin_readcyclecounter(); } while (0)
#else
#define SP(j) do { } while (0)
#endif
        SP(0);
        if (s > 0) {
            // poll with one LDS read per round and back off: 15 spinning waves otherwise saturate the LDS queue the
            // working wave needs (first version: data requested behind the counter on every round - 450-cycle loads)
            for (;;) {
                const int f = vprog[s - 1];
                if (f >= k + 3) break;
                __builtin_amdgcn_s_sleep(B2T_SLEEP);
            }
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int e = 0; e < B; ++e) {
            x[e] = base[offx0 + e];
            y[e] = on ? base[off[e]] : 0.0;
        }
        double s2 = 0.0;
#pragma unroll
        for (int e = 1; e < B; ++e) s2 = fma(x[e], x[e], s2);
        if (s2 == 12345.0) return;      // consume the loads before the stamp
        SP(1);
        const double al = x[0];
        const double xx = fma(al, al, s2);
        const bool nz = s2 != 0.0 && xx > 1e-280;
        const double rs = __builtin_amdgcn_rsq(nz ? xx : 1.0);
        double nrm = xx * rs;
        const double hrs = 0.5 * rs;
        nrm = fma(fma(-nrm, nrm, xx), hrs, nrm);
        nrm = fma(fma(-nrm, nrm, xx), hrs, nrm);
        const double bneg = copysign(nrm, al);
        const double ib = frcp(bneg), is = frcp(al + bneg);
        const double beta = nz ? -bneg : al;
        const double tau = nz ? (bneg + al) * ib : 0.0;
        const double scale = nz ? is : 0.0;
        if (tau == 12345.0 || scale == 12345.0) return;
        SP(2);
        double v[B];
        v[0] = 1.0;
#pragma unroll
        for (int e = 1; e < B; ++e) v[e] = x[e] * scale;
        // y <- (I - tau v v^T) y for the left / next block vectors; the diagonal block needs both sides
        double dot = y[0];
#pragma unroll
        for (int e = 1; e < B; ++e) dot = fma(v[e], y[e], dot);
        double vidx = 1.0;
#pragma unroll
        for (int e = 1; e < B; ++e) vidx = idx == e ? v[e] : vidx;
        const double vp = group_sum<B>(vidx * dot);              // v^T D v (diagonal-block lanes)
        const double t = tau * dot;
        const double w = tau * fma(-0.5 * tau * vp, vidx, dot);   // w = tau (p - tau/2 (v^T p) v)
        double we[B];
        we[0] = group_bcast<B, 0>(w);
        if constexpr (B >= 2) we[1] = group_bcast<B, 1>(w);
        if constexpr (B >= 4) {
            we[2] = group_bcast<B, 2>(w);
            we[3] = group_bcast<B, 3>(w);
        }
        if constexpr (B >= 8) {
            we[4] = group_bcast<B, 4>(w);
            we[5] = group_bcast<B, 5>(w);
            we[6] = group_bcast<B, 6>(w);
            we[7] = group_bcast<B, 7>(w);
        }
        const bool dg = role == 2;
        const double c1 = dg ? w : t;                  // multiplies v_e
        const double c2 = dg ? vidx : 0.0;             // multiplies w_e
#pragma unroll
        for (int e = 0; e < B; ++e) y[e] = fma(-c2, we[e], fma(-c1, v[e], y[e]));
        if (lane == 0 && nz) {
            y[0] = beta;
#pragma unroll
            for (int e = 1; e < B; ++e) y[e] = 0.0;
        }
        if (y[0] == 12345.0) return;
        SP(3);
        double* wbase = Ab + r0 * LD;
        if (on) {
#pragma unroll
            for (int e = 0; e < B; ++e)
                if (!dg || e >= idx) wbase[off[e]] = y[e];
        }
        asm volatile("" ::: "memory");
        if (lane == 0) vprog[s] = k + 1;
        SP(4);
    };
    int ce0[B];                                        // step 0: the "left block" is the single column s = r0 - 1
#pragma unroll
    for (int e = 0; e < B; ++e) ce0[e] = role == 0 ? -LD + 1 + e : ce[e];
    const bool on0 = lane_used && (role != 0 || idx == 0);
    for (int s = wave; s < n - 2; s += B2T_WAVES) {
#ifdef B2T_SWEEPPROF
        if (stamps && lane == 0) stamps[64 + s] = __builtin_readcyclecounter();
#endif
        step(s, 0, s + 1, ce0, -LD + 1, on0);
        for (int k = 1;; ++k) {
            const int r0 = s + k * B + 1;
            if (n - r0 < 2) break;
            step(s, k, r0, ce, -B * LD + B, lane_used);
        }
        asm volatile("" ::: "memory");
        if (lane == 0) vprog[s] = B2T_DONE;
    }
    __syncthreads();
    if (tid < n) {
        de[2 * tid] = Ab[tid * LD];
        de[2 * tid + 1] = tid < n - 1 ? Ab[tid * LD + 1] : 0.0;
    }
    if (stamps && tid == 0) {
        stamps[1] = __builtin_amdgcn_s_memrealtime();
        stamps[7] = __builtin_readcyclecounter();
    }
}

template <int B>
static void run_case(int n, int reps, FILE* f, hipStream_t s, unsigned long long* st) {
    std::vector<double> band((size_t)n * (B + 1), 0.0);
    srand(11 + B);
    for (int j = 0; j < n; ++j)
        for (int d = 0; d <= B && j + d < n; ++d) band[(size_t)j * (B + 1) + d] = rand() / (double)RAND_MAX - 0.5;
    double *dband, *dde;
    CK(hipMalloc(&dband, sizeof(double) * band.size()));
    CK(hipMalloc(&dde, sizeof(double) * 2 * n));
    CK(hipMemcpy(dband, band.data(), sizeof(double) * band.size(), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k_band2tri<B>, dim3(1), dim3(B2T_WAVES * 64), 0, s, dband, n, dde, nullptr);
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_band2tri<B>, dim3(1), dim3(B2T_WAVES * 64), 0, s, dband, n, dde, nullptr);
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    CK(hipGetLastError());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemset(st, 0, 8 * 1024));
    hipLaunchKernelGGL(k_band2tri<B>, dim3(1), dim3(B2T_WAVES * 64), 0, s, dband, n, dde, st);
    CK(hipStreamSynchronize(s));
    std::vector<unsigned long long> hs(1024);
    CK(hipMemcpy(hs.data(), st, 8 * 1024, hipMemcpyDeviceToHost));
    std::vector<double> de(2 * n);
    CK(hipMemcpy(de.data(), dde, sizeof(double) * 2 * n, hipMemcpyDeviceToHost));
    int nsteps = 0;
    for (int sw = 0; sw < n - 2; ++sw)
        for (int k = 0;; ++k) {
            const int r0 = k == 0 ? sw + 1 : sw + k * B + 1;
            if (n - r0 < 2) break;
            ++nsteps;
        }
    printf("band -> tridiagonal, n = %d, b = %d: %.2f us per launch (%d back-to-back), in-kernel %.2f us = %llu cycles; %d sweeps, %d steps, "
           "%.0f cycles per sweep on the critical path\n",
           n, B, 1e3 * ms / reps, reps, 0.01 * (double)(hs[1] - hs[0]), hs[7] - hs[6], n - 2, nsteps, (double)(hs[7] - hs[6]) / (n - 2));
#ifdef B2T_STEPPROF
    for (int q = 0; q < 2; ++q)
        printf("  sweep %d step 5: loads %lld | reflector chain %lld | apply %lld | store + counter %lld cycles\n", q ? 40 : 0,
               (long long)(hs[16 + 8 * q + 1] - hs[16 + 8 * q]), (long long)(hs[16 + 8 * q + 2] - hs[16 + 8 * q + 1]),
               (long long)(hs[16 + 8 * q + 3] - hs[16 + 8 * q + 2]), (long long)(hs[16 + 8 * q + 4] - hs[16 + 8 * q + 3]));
#endif
#ifdef B2T_SWEEPPROF
    printf("  sweep start, cycles after the previous sweep's start:");
    for (int sw = 1; sw < n - 2; ++sw) printf(" %lld", (long long)(hs[64 + sw] - hs[64 + sw - 1]));
    printf("\n");
#endif
    fprintf(f, "case %d %d\n", n, B);
    for (int j = 0; j < n; ++j) {
        for (int d = 0; d <= B; ++d) fprintf(f, "%.17g ", band[(size_t)j * (B + 1) + d]);
        fprintf(f, "\n");
    }
    for (int j = 0; j < n; ++j) fprintf(f, "%.17g %.17g\n", de[2 * j], de[2 * j + 1]);
    CK(hipFree(dband));
    CK(hipFree(dde));
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 128;
    const int reps = argc > 2 ? atoi(argv[2]) : 200;
    hipStream_t s;
    CK(hipStreamCreate(&s));
    unsigned long long* st;
    CK(hipMalloc(&st, 8 * 1024));
    FILE* f = fopen("gpurun_out/band2tri_T.txt", "w");
    if (!f) { printf("cannot open gpurun_out/band2tri_T.txt\n"); return 1; }
    run_case<2>(n, reps, f, s, st);
    run_case<4>(n, reps, f, s, st);
    run_case<8>(n, reps, f, s, st);
    fclose(f);
    return 0;
}
