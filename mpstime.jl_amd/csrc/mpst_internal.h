// Internal declarations shared by the translation units of libmpstime_hip.so.
// gfx950 only.  The fp64 real sweep (View) and the element-typed sweep (TView: fp32 / complex, mpst_typed.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <string>
#include <vector>
#include "../../include/mpstime_hip.h"

namespace mpst {

typedef double d4 __attribute__((ext_vector_type(4)));

// v_mfma_f64_16x16x4_f64: D(16x16) += A(16x4) * B(4x16).
// lane l supplies A[l&15][l>>4] and B[l>>4][l&15]; it owns D[(l>>4) + 4*r][l&15], r = 0..3.
__device__ __forceinline__ d4 mfma_f64(double a, double b, d4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ int hi32(double x) { return __double2hiint(x); }

// The two MFMA forms the engine computes in, behind one interface: v_mfma_f64_16x16x4_f64 and v_mfma_f32_16x16x4_f32.  A / B
// operands are laid out alike (lane l: A[l&15][l>>4], B[l>>4][l&15]); the C/D maps differ: f64 row = (l>>4) + 4*reg,
// f32 row = 4*(l>>4) + reg.
typedef float f4 __attribute__((ext_vector_type(4)));
template <typename R> struct Mx;
template <> struct Mx<double> {
    using acc_t = d4;
    using vec2 = double2;
    static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) { return mfma_f64(a, b, c); }
    static __device__ __forceinline__ int row(int kq, int r) { return kq + 4 * r; }
};
template <> struct Mx<float> {
    using acc_t = f4;
    using vec2 = float2;
    static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row(int kq, int r) { return 4 * kq + r; }
};

// # eigenvalues of T smaller than x: sign changes of the Sturm sequence, division-free, rescaled by
// a power of two every 8 steps.  (d_j, e_{j-1}^2) pairs are fetched 8 at a time so the LDS latency
// is paid once per 8 steps.
__device__ __forceinline__ int sturm_count(const double* __restrict__ de, int n, double x) {
    // One wave per SIMD runs this loop, so what counts is the number of instructions per step: 3 fp64
    // ops (d - x, e^2 * p_{j-2}, fma) + 1 integer op that shifts the sign of p_j into a bit queue; the
    // sign changes of 8 steps are counted with one popcount, the rescaling is a frexp/ldexp pair, and
    // the 8 (d, e^2) pairs of the next group are requested from LDS before the current 8 are consumed.
    // (The scalar data path was tried for T - it is wave-uniform - and lost: s_load latency per group.)
    const double2* __restrict__ de2 = (const double2*)de;
    double pp = 1.0, p = de[0] - x;
    unsigned sb = ((unsigned)hi32(p)) >> 31;      // bit queue of signs, newest in bit 0
    int cnt = (int)sb;
    int j = 1;
    double2 bufA[8], bufB[8];                     // ping-pong: no register copies
    auto load8 = [&](double2 (&buf)[8], int j0) {
#pragma unroll
        for (int u = 0; u < 8; ++u) buf[u] = de2[j0 + u];
    };
    auto run8 = [&](const double2 (&buf)[8]) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const double t = buf[u].x - x;
            const double pn = fma(t, p, -buf[u].y * pp);
            sb = __builtin_amdgcn_alignbit(sb, (unsigned)hi32(pn), 31);   // (sb << 1) | sign(pn)
            pp = p;
            p = pn;
        }
        cnt += __popc((sb ^ (sb >> 1)) & 0xffu);
        int e = __builtin_amdgcn_frexp_exp(p);
        if (p == 0.0) e = __builtin_amdgcn_frexp_exp(pp);
        p = __builtin_amdgcn_ldexp(p, -e);
        pp = __builtin_amdgcn_ldexp(pp, -e);
    };
    // rows n .. n+7 are padding (k_eig_vec, k_bt_vec: T is padded with decoupled rows): whole groups only
    if (j < n) load8(bufA, j);
    while (j < n) {
        if (j + 8 < n) load8(bufB, j + 8);
        run8(bufA);
        j += 8;
        if (j >= n) break;
        if (j + 8 < n) load8(bufA, j + 8);
        run8(bufB);
        j += 8;
    }
    return cnt;
}


// One half of a two-sided Sturm count: the recurrence of sturm_count over `rows` rows of a (d, e^2) pair array that may be
// the reversed matrix, without padding (whole groups of 8, then the remainder row by row).  Returns the number of negative
// pivots among these rows and leaves the last two values of the sequence (rescaled together) in p, pp - what the caller
// needs to evaluate the twisted pivot where the two halves meet.
__device__ __forceinline__ int sturm_half(const double* __restrict__ arr, int rows, double x, double& p_out, double& pp_out) {
    const double2* __restrict__ a2 = (const double2*)arr;
    double pp = 1.0, p = arr[0] - x;
    unsigned sb = ((unsigned)hi32(p)) >> 31;
    int cnt = (int)sb;
    int j = 1;
    while (j + 8 <= rows) {               // (a ping-pong prefetch of the next group, as in sturm_count, measured slower here)
        double2 buf[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) buf[u] = a2[j + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const double t = buf[u].x - x;
            const double pn = fma(t, p, -buf[u].y * pp);
            sb = __builtin_amdgcn_alignbit(sb, (unsigned)hi32(pn), 31);
            pp = p;
            p = pn;
        }
        cnt += __popc((sb ^ (sb >> 1)) & 0xffu);
        int e = __builtin_amdgcn_frexp_exp(p);
        if (p == 0.0) e = __builtin_amdgcn_frexp_exp(pp);
        p = __builtin_amdgcn_ldexp(p, -e);
        pp = __builtin_amdgcn_ldexp(pp, -e);
        j += 8;
    }
    for (; j < rows; ++j) {
        const double2 b = a2[j];
        const double pn = fma(b.x - x, p, -b.y * pp);
        sb = __builtin_amdgcn_alignbit(sb, (unsigned)hi32(pn), 31);
        cnt += (int)((sb ^ (sb >> 1)) & 1u);
        pp = p;
        p = pn;
    }
    {
        // hand the pair over at unit scale: the caller multiplies values of the two halves
        int e = __builtin_amdgcn_frexp_exp(p);
        if (p == 0.0) e = __builtin_amdgcn_frexp_exp(pp);
        p = __builtin_amdgcn_ldexp(p, -e);
        pp = __builtin_amdgcn_ldexp(pp, -e);
    }
    p_out = p;
    pp_out = pp;
    return cnt;
}

// reciprocal to full double precision from the hardware seed (2 Newton steps)
__device__ __forceinline__ double frcp(double b) {
    double r = __builtin_amdgcn_rcp(b);
    r = fma(fma(-b, r, 1.0), r, r);
    r = fma(fma(-b, r, 1.0), r, r);
    return r;
}

// ---- cross-lane reductions on the VALU (DPP) instead of ds_bpermute ------------------------
// hipcc lowers __shfl_xor to ds_bpermute_b32 (LDS crossbar, >100 cycles of dependent latency
// per step); the reductions here sit on the critical path of single-workgroup kernels, so they
// use DPP row operations (a VALU modifier) and v_readlane for the cross-row step.
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
// Workgroup barrier that orders LDS traffic only: unlike __syncthreads() it does not drain vmcnt, so
// direct global->LDS loads (global_load_lds) issued earlier stay in flight across it.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// 16 bytes per lane global -> LDS without a VGPR stop-over; the LDS destination of a wave is
// lds_wave_base + lane * 16 (contiguous 1 KB), the global source is per lane.  Issued from inline asm
// on purpose: hipcc drains vmcnt(0) before the first ds_read after a global_load_lds it can see (it
// cannot prove the LDS ranges disjoint), which would serialise the copy with the work it is meant to
// overlap.  The caller owns the wait: s_waitcnt vmcnt(0) + a barrier before anybody reads the data,
// and no compiler-visible global load may be consumed while these are in flight.
__device__ __forceinline__ void glds16(const double* gsrc_lane, double* lds_wave_base) {
    unsigned keep;
    const unsigned lds_dst = __builtin_amdgcn_readfirstlane(
        (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)lds_wave_base);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc_lane), "s"(lds_dst)
                 : "memory");
}
// sum over the 4 lanes l, l^16, l^32, l^48 (same position in each 16-lane row), result in all 4:
// v_permlane32_swap(a, b) exchanges lanes 32-63 of a with lanes 0-31 of b; v_permlane16_swap the odd
// rows of a with the even rows of b (gfx950).  With a = b = x the two halves add up lane-wise.
__device__ __forceinline__ double row4_sum(double x) {
    unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
    auto r0 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto r1 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    x = __hiloint2double((int)r1[0], (int)r0[0]) + __hiloint2double((int)r1[1], (int)r0[1]);
    lo = (unsigned)__double2loint(x);
    hi = (unsigned)__double2hiint(x);
    auto s0 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto s1 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double((int)s1[0], (int)s0[0]) + __hiloint2double((int)s1[1], (int)s0[1]);
}
constexpr int DPP_XOR1 = 0xB1;         // quad_perm [1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;         // quad_perm [2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141; // reverse within each 8 lanes
constexpr int DPP_MIRROR = 0x140;      // reverse within each 16 lanes
// all-reduce (sum) over aligned groups of 4 / 8 / 16 lanes
__device__ __forceinline__ double sum4(double x) {
    x += dpp_mov<DPP_XOR1>(x);
    x += dpp_mov<DPP_XOR2>(x);
    return x;
}
__device__ __forceinline__ double sum8(double x) {
    x = sum4(x);
    x += dpp_mov<DPP_HALF_MIRROR>(x);
    return x;
}
__device__ __forceinline__ double sum16(double x) {
    x = sum8(x);
    x += dpp_mov<DPP_MIRROR>(x);
    return x;
}
__device__ __forceinline__ double readlane_f64(double x, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
    return __hiloint2double(hi, lo);
}
// all-reduce over the 64 lanes of a wave (fixed order, wave-uniform result)
__device__ __forceinline__ double wave_sum(double x) {
    x = sum16(x);
    return (readlane_f64(x, 0) + readlane_f64(x, 16)) + (readlane_f64(x, 32) + readlane_f64(x, 48));
}
// same sum, same association ((r0+r1)+(r2+r3)), with the cross-row part as two DPP row broadcasts
// (row_bcast:15 into rows 1,3; row_bcast:31 into rows 2,3) and a single read of lane 63: 8 dependent
// instructions instead of 15 - for latency-critical single-wave chains.
__device__ __forceinline__ double dpp_bcast_add(double x, const int ctrl15_or_31) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    if (ctrl15_or_31 == 15) {
        lo = __builtin_amdgcn_update_dpp(0, lo, 0x142, 0xA, 0xF, false);
        hi = __builtin_amdgcn_update_dpp(0, hi, 0x142, 0xA, 0xF, false);
    } else {
        lo = __builtin_amdgcn_update_dpp(0, lo, 0x143, 0xC, 0xF, false);
        hi = __builtin_amdgcn_update_dpp(0, hi, 0x143, 0xC, 0xF, false);
    }
    return x + __hiloint2double(hi, lo);
}
// inclusive prefix sum over the 64 lanes: Hillis-Steele inside the 16-lane rows (row_shr 1, 2, 4, 8 with zero fill), then
// the last lane of row 0 / 2 into rows 1 / 3 (row_bcast:15) and of row 1 into rows 2 and 3 (row_bcast:31)
__device__ __forceinline__ double wave_incl_scan(double x) {
    x += dpp_mov<0x111>(x);
    x += dpp_mov<0x112>(x);
    x += dpp_mov<0x114>(x);
    x += dpp_mov<0x118>(x);
    x = dpp_bcast_add(x, 15);
    x = dpp_bcast_add(x, 31);
    return x;
}
// 64-lane sum on the matrix pipe: v_mfma_f64_4x4x4 (4 independent 4x4x4 products; A lane = 16k + 4b + i,
// B lane = 16k + 4b + j, D lane = 16i + 4b + j - measured) with an all-ones partner first adds the four
// 16-lane rows (k), then the four lanes of a quad group (i <-> k); two DPP rotations add the four groups b.
// 2 MFMA + 6 VALU instructions instead of 30, result in every lane; fixed order, so every wave that calls
// it on the same data gets the same bits.
__device__ __forceinline__ double wave_sum_mfma(double x) {
    x = __builtin_amdgcn_mfma_f64_4x4x4f64(x, 1.0, 0.0, 0, 0, 0);
    x = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, x, 0.0, 0, 0, 0);
    x += dpp_mov<0x128>(x);       // row_ror:8
    x += dpp_mov<0x124>(x);       // row_ror:4
    return x;
}
// x(lane) + x(lane ^ 16): one v_permlane16_swap per dword (odd 16-lane rows of a <-> even rows of b)
__device__ __forceinline__ double pair16_sum(double x) {
    const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
    auto s0 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto s1 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double((int)s1[0], (int)s0[0]) + __hiloint2double((int)s1[1], (int)s0[1]);
}
__device__ __forceinline__ double wave_sum_fast(double x) {
    x = sum16(x);
    x = dpp_bcast_add(x, 15);
    x = dpp_bcast_add(x, 31);
    return readlane_f64(x, 63);
}
__device__ __forceinline__ double max16(double x) {
    x = fmax(x, dpp_mov<DPP_XOR1>(x));
    x = fmax(x, dpp_mov<DPP_XOR2>(x));
    x = fmax(x, dpp_mov<DPP_HALF_MIRROR>(x));
    x = fmax(x, dpp_mov<DPP_MIRROR>(x));
    return x;
}
__device__ __forceinline__ double wave_max(double x) {
    x = max16(x);
    return fmax(fmax(readlane_f64(x, 0), readlane_f64(x, 16)), fmax(readlane_f64(x, 32), readlane_f64(x, 48)));
}

// A class-pure run of consecutive series (<= 16 for tiles, <= 64 for chunks).
struct Span {
    int32_t start;
    int32_t count;
    int32_t cls;
    int32_t pad;
};

// A part = the series one persistent workgroup of k_bond_fused walks: a class-pure run [start, start+count) of class
// `own`, contracted with the bond tensor of class `cls` (KLD: cls == own; MSE: every class).  Parts are ordered by
// `cls`, so the partial gradients of one class are consecutive.
struct Part {
    int32_t start, count, own, cls;
    int32_t first_of_cls, pad0, pad1, pad2;
};
constexpr int PARTS_TARGET = 128;   // persistent workgroups of the fused gradient kernel (256 for >= 512 tiles of 16 series)

constexpr int TILE_S = 16;    // series per yhat/env tile (one MFMA M-tile)
constexpr int CHUNK_S = 64;   // series per gradient chunk (the GEMM K extent of one partial)
constexpr int GB = 64;        // gradient output block edge per workgroup
// k_grad (unfused path): workgroups that share the chunks of one (class, output block); each walks its share with the
// accumulators in registers and writes one partial block.  Enough workgroups to fill the chip (GRAD_WG_TARGET over all
// blocks and classes), never more than there are chunks: the partial count is independent of N.
constexpr int GRAD_WG_TARGET = 512;
inline int grad_nsplit(int nchunks, int nblocks, int C) {
    const int per = (GRAD_WG_TARGET + nblocks * C - 1) / (nblocks * C);
    return per < 1 ? 1 : (per > nchunks ? (nchunks > 0 ? nchunks : 1) : per);
}
constexpr int MAX_DIM = 128;  // d*chi_max limit of the register/LDS-resident eigensolver and of the single-pass bond kernels
constexpr int DIM_LIMIT = 1024;  // d*chi_max limit of the engine (LDS tile of 16 series x d*chi doubles)
constexpr int CAP_LIMIT = 128;   // largest bond dimension (environment rows are staged 16 lanes x 8 values)
constexpr int MAX_C = 16;

// Device-resident scalars of the running sweep (never read by the host inside a sweep).
struct DevScalars {
    double loss;        // loss of the first update iteration of the last bond
    double grad_norm;
    double bt_norm2;    // trace of the Gram matrix = ||bt_new||_F^2
    double inv_norm;    // 1/||bt_new|| when rescale_after else 1
    int32_t n_keep;
    int32_t n_spec;
    int32_t eig_sweeps;
    int32_t status;     // sticky error flag (non-finite spectrum ...)
    int32_t eig_sweeps_total;
    int32_t eig_fallbacks;      // bonds on which the tridiagonal path failed its check (Jacobi used)
    int32_t pad[2];             // one-shot all-reduce: time-out diagnostics (OneShotParams.dbg)
    int32_t xref;               // typed MSE: largest overlap exponent of the bond's series (k_tbt_assemble resets, k_tyhat raises)
    int32_t redo;               // k_bond_tail: 1 + the position in the sweep of a bond whose on-device verification failed - every later tail launch of the
                                // sweep leaves at once and the host finishes the sweep from that bond on the chain that has the Jacobi fallback (sticky
                                // until the host clears it)
    unsigned long long eig_stamps[64];  // s_memrealtime (100 MHz) at the phase boundaries of the last eigensolve
};

// One encoded data set on the device.
struct DataSet {
    int64_t N = 0;
    double* phi = nullptr;       // [T][N][d] site-major
    int32_t* label = nullptr;    // [N]
    Span* tiles = nullptr;       // class-pure tiles of <= TILE_S series
    int32_t ntiles = 0;
    Span* chunks = nullptr;      // class-pure chunks of <= CHUNK_S series
    int32_t nchunks = 0;
    int32_t* cls_chunk_off = nullptr;  // [C+1] first chunk of each class
    int32_t* cls_off = nullptr;        // [C+1] first series of each class
    double* inv_count = nullptr;       // [C] 1 / (global series count of the class)
    Part* parts[2] = {nullptr, nullptr};          // [0] KLD, [1] MSE
    int32_t nparts[2] = {0, 0};
    int32_t* part_off[2] = {nullptr, nullptr};    // [C+1] first part of each bond-tensor class
    std::vector<int64_t> counts;       // local per-class counts
    std::vector<int64_t> gcounts;      // global per-class counts
    int64_t Nglobal = 0;
};

// Everything a kernel needs, passed by value.
struct View {
    int32_t T, d, C, chi_max;  // chi_max = truncation maxdim
    int32_t cap;               // capacity bond dimension: row stride of LE/RE/E, slot sizing
    int64_t N;          // series of the data set the launch works on
    double invN;        // 1 / global N
    const double* phi;
    const int32_t* label;
    const Span* tiles;
    const Span* chunks;
    const int32_t* cls_chunk_off;
    const double* inv_count;
    int32_t ntiles, nchunks;
    int32_t* chi;       // [T+1] device bond dimensions
    int32_t* label_site;
    double* sites;      // T slots of site_stride doubles, layout [c][l][s][r] compact
    int64_t site_stride;
    double* LE;         // [T][N][chi_max]
    double* RE;
    double* bt;         // [C][X][Y] compact
    double* yhat;       // [C][N]
    double* tile_loss;  // [C][ntiles]
    double* partial;    // gradient partials
    const Part* parts;  // fused path: persistent workgroups of k_bond_fused
    const int32_t* part_off;
    int32_t nparts, n_norm_part;
    double* norm_part;  // [n_norm_part] pieces of ||grad||^2
    double* btn;        // bt_new, written by k_gram_upd
    double* btnT;       // going right on the four-launch chain: bt_new once more as [c][y][x] (k_bond_tail contracts it along y), else null
    // sliced bond GEMMs (k_yhat_s / k_grad_s, mpst_fused.hip)
    const int32_t* cls_off;   // [C+1] first series of every class
    int32_t kcls_off[MAX_C + 1];    // the same table, and the first 16-series tile of every class, in the kernel arguments:
    int32_t kcls_tile[MAX_C + 1];   // no dependent global load between a workgroup's id and its series
    double* ypart;      // [passes][N][8 slices] contributions of the 16-column slices of B_c to yhat
    double* lossp;      // [C][ksplit] loss pieces
    unsigned int* tick; // [C * blocks_cap + 1] arrival tickets of the gradient blocks, + the launch-wide one
    int32_t b2_ksplit;  // shares the series of a pass are split into
    int32_t b2_nw;      // waves per workgroup of k_grad_s: 8, or 4 for a context that is advanced in batches (mpst_fused.hip: k_grad_s)
    int32_t n_lossp;    // > 0: the bond's loss is still in lossp's pieces ([C][n_lossp]); 0: it is gradbuf[0]
    unsigned long long* dbg;   // bring-up stamps (-DMPST_B2_DEBUG), else null
    double* trace;      // track_cost: this bond's row of the loss trace ([update_iters + 1]) or null
    int32_t trace_it;   // which entry the launch at hand fills
    int32_t yhat_scaled; // k_yhat: multiply yhat by sc->inv_norm (loss at the normalised bt_new)
    double* gradbuf;    // [2 + C*Lmax]: loss, pad, grad[c][x][y]
    double* gram;       // [MAX_DIM*MAX_DIM]
    double* lam;        // [MAX_DIM]
    double* E;          // [MAX_DIM][ldE = cap] eigenvectors of the kept subspace
    double* eig_ws;     // eigensolver workspace (tridiagonal matrix, reflectors, eigenpairs)
    DevScalars* sc;
    // options
    int32_t loss, optimiser, rescale_before, rescale_after, train_sep, svd_alg;
    double eta, cutoff;
    // 2: `gram` is the real 2n x 2n embedding [[Gr, -Gi], [Gi, Gr]] of a complex Hermitian Gram matrix (element-typed sweep,
    // mpst_typed.hip): every eigenvalue is double, the eigensolvers compute ONE vector per pair and store its partner
    // J u = (-u_im, u_re) next to it (columns 2k, 2k+1), the truncation rule runs on the de-duplicated spectrum and
    // chi_max / cap of this view are the doubled (real) counts.  0 / 1: plain real symmetric problem.
    int32_t zw;
    // large bonds: the bond tensor as the subspace eigensolver reads it (mpst_eig_subspace.inl) - [C][X][Y] in fp64 (ss_f32 = 0) or
    // fp32 (1); null: no subspace attempt (complex element types, small bonds).  Raw mode of the LDS-resident solver
    // (launch_eig_raw_gated): label_site doubles as the gate word (0 = leave at once).
    const void* ss_bt;
    int32_t ss_f32;
};
__device__ __forceinline__ int view_zw(const View& v) { return v.zw == 2 ? 2 : 1; }

// NDTensors truncate! (relative cutoff) as the eigensolvers' final kernels apply it: lam[0..K0) are the largest
// eigenvalues of the Gram matrix (descending), tr its trace, nspec = min(rows, cols) of the decomposed matrix, inv2 the
// rescale factor.  pair (complex problem in its real embedding): lam holds every eigenvalue twice and K0 / nspec are the
// doubled counts (tr is the trace of the COMPLEX matrix, half the embedding's); the rule runs on the de-duplicated spectrum
// and the complex count is returned.
__device__ __forceinline__ int truncate_rule(const double* lam, int K0, int nspec, double tr, double inv2, double cutoff, bool pair) {
    const int st = pair ? 2 : 1;
    const int K = K0 / st, ns = nspec / st;
    const double scale0 = tr * inv2;
    const double scale = scale0 == 0.0 ? 1.0 : scale0;
    double kept = 0.0;
    for (int i = 0; i < K; ++i) kept += lam[st * i] * inv2;
    int nk = K;
    double truncerr = scale0 - kept;
    if (truncerr < 0.0 || ns <= K) truncerr = 0.0;
    if (ns > 1) {
        while (nk > 1 && truncerr + lam[st * (nk - 1)] * inv2 <= cutoff * scale) {
            truncerr += lam[st * (nk - 1)] * inv2;
            --nk;
        }
    }
    return nk;
}

// The environment products out_i = Z_i M (update_caches!, construct_caches) are FOUR chains of MFMAs over consecutive quarters of the
// contraction, added as (q0 + q1) + (q2 + q3): one wave runs the four chains side by side (k_env, k_env_split: four independent
// accumulators instead of one dependent chain of 32), or four waves run one each (k_bond_tail) - the same bits either way, which is
// what keeps a sweep with the reference's cache rebuilds identical to one without.  k-steps (of 4 entries) per quarter:
__device__ __forceinline__ int env_ks4(int nsteps) { return (nsteps + 3) >> 2; }

// loss of the current bond: gradbuf[0], or - after k_grad_s on a single rank - the sum of its pieces (fixed order)
__device__ __forceinline__ double bond_loss(const View& v) {
    if (v.n_lossp <= 0) return v.gradbuf[0];
    const bool mse = v.loss == MPST_LOSS_MSE;
    double l = 0.0;
    for (int cc = 0; cc < v.C; ++cc) {
        const double w = mse ? v.invN : (v.train_sep ? v.inv_count[cc] : v.invN);     // loss_functions.jl:612 / :423,:371
        double lc = 0.0;
        for (int k = 0; k < v.n_lossp; ++k) lc += v.lossp[cc * v.n_lossp + k];
        l += lc * w;
    }
    return l;
}

// ---- element-typed sweep (mpst_typed.hip): fp32 / complex MPSs and encodings ------------------------------------------------
// Buffers marked (E) hold elements of the context's type: float / double, or interleaved (re, im) pairs of them.
struct TView {
    int32_t T, d, C, chi_max, cap;
    int32_t cx, f32;                // element type
    int64_t N;
    double invN;
    const void* phi;                // (E) [T][N][d] product states (NOT conjugated: the kernels conjugate)
    const int32_t* label;
    const Span* tiles;
    const Span* chunks;
    const int32_t* cls_chunk_off;
    const double* inv_count;
    int32_t ntiles, nchunks;
    int32_t* chi;
    int32_t* label_site;
    void* sites;                    // (E) T slots of site_stride elements, [c][l][s][r] compact
    int64_t site_stride;
    void* LE;                       // (E) [T][N][cap], every row scaled by 2^-x (largest component in [0.5, 1)) ...
    void* RE;
    int32_t* xLE;                   // ... [T][N] the exponents x (k_tenv)
    int32_t* xRE;
    void* bt;                       // (E) [C][X][Y] compact
    double* yhat;                   // [C][N][2] (re, im), fp64: the overlap computed from the SCALED environments
    int32_t* yexp;                  // [N] its exponent: yhat_true = yhat 2^yexp
    double* tile_loss;              // [C][ntiles]
    void* partial;                  // (E) gradient partials [C][nsplit][X][Y]
    double* gradbuf;                // fp64 [2 + C*L*(1|2)]: loss, pad, grad - the all-reduce message
    double* norm_part;              // pieces of ||grad||^2
    int32_t n_norm_part;
    double* gram;                   // fp64 Gram matrix (complex: the 2n x 2n real embedding)
    double* E;                      // fp64 eigenvectors from the eigensolver, row stride ldE (complex: rows [0,n) Re, [n,2n) Im; vector k in column 2k)
    int32_t ldE;
    DevScalars* sc;
    int32_t loss, optimiser, rescale_before, rescale_after, train_sep;
    double eta, cutoff;
    double* trace;
    int32_t trace_it, yhat_scaled;
};
void launch_tbt_assemble(const TView& v, int lid, hipStream_t s);
void launch_tbt_prescale(const TView& v, int lid, hipStream_t s);
void launch_tenv(const TView& v, int site, int left_side, const void* prev, const int32_t* xprev, int prev_bond, int mode, int out_bond, void* out,
                 int32_t* xout, hipStream_t s);
void launch_tyhat(const TView& v, int lid, hipStream_t s);
void launch_tgrad(const TView& v, int lid, hipStream_t s);
void launch_tgrad_reduce(const TView& v, int lid, hipStream_t s);
void launch_tgrad_norm(const TView& v, int lid, hipStream_t s);
void launch_tupdate(const TView& v, int lid, int first_iter, hipStream_t s);
void launch_tgram(const TView& v, int lid, int going_left, hipStream_t s);
void launch_tsplit(const TView& v, int lid, int going_left, hipStream_t s);
void launch_teval_final(const TView& v, const void* Lc, const int32_t* Lx, const void* Rc, const int32_t* Rx, double* yout, hipStream_t s);
void launch_teval_reduce(const TView& v, const double* yin, double* out3, int64_t* conf, int32_t* pred, hipStream_t s);
void launch_tnorm2(const TView& v, double* out_norm2, double* gscratch /* 6*cap*cap doubles */, hipStream_t s);
void launch_tscale_sites(const TView& v, const double* norm2, hipStream_t s);
void launch_tcast(const double* src, int src_cx, void* dst, int dst_cx, int dst_f32, int64_t n, hipStream_t s);
int typed_grad_nsplit(const TView& v, int nchunks);
int typed_norm_parts(const TView& v);
size_t typed_max_lds(const TView& v);
hipError_t typed_init_attrs(int device);

// fused path for bond tensors up to MAX_DIM x MAX_DIM (mpst_fused.hip)
void launch_bond_fused(const View& v, int lid, int assemble, hipStream_t s);
void launch_fused_reduce(const View& v, int lid, hipStream_t s);
void launch_grad_norm(const View& v, int lid, hipStream_t s);
// sliced bond GEMMs (k_yhat_s + k_grad_s): the pair that replaces k_bond_fused + k_fused_reduce
void launch_yhat_s(const View& v, int lid, hipStream_t s);
void launch_grad_s(const View& v, int lid, hipStream_t s);
void launch_loss_sum(const View& v, hipStream_t s);      // gradbuf[0..1] from k_grad_s's loss pieces (before an all-reduce)
int b2_blocks_cap(const View& v);                          // gradient blocks per class at the capacity bond dimension
int b2_ksplit(const View& v, int64_t max_pass);            // shares per block; max_pass = most series any pass walks
int64_t b2_partial_elems(const View& v, int64_t max_pass);
hipError_t b2_init_attrs(int device);
void launch_gram_upd(const View& v, int lid, int going_left, int first_iter, hipStream_t s);
void launch_env_split(const View& v, int lid, int going_left, int site, int left_side, const double* prev, int prev_bond,
                      int out_bond, double* out, int chain /* also assemble the next bond's tensor */, hipStream_t s);
// the four-launch chain's last launch (k_bond_tail): k_eig_fin's verification + polish, environment update, back-split, the next
// bond's tensor (chain) and the next bond's overlaps (want_next) in one
void launch_bond_tail(const View& v, int lid, int going_left, int chain, int want_next, unsigned long long* span /* [2 * 2048] or null */, hipStream_t s);
bool bond_tail_supported(const View& v);
// construct_caches in one launch: a workgroup walks the whole chain with its 16 series (k_env_walk); left_side: LE, else RE
void launch_env_walk(const View& v, int left_side, int nstep /* sites, from the chain end */, hipStream_t s);
bool env_walk_supported(const View& v);
// the headline chain for K independent fits of one shape per launch (blockIdx.z selects the fit's View in the device array vs)
void launch_yhat_s_b(const View& v, const View* vs, int K, int lid, hipStream_t s);
void launch_grad_s_b(const View& v, const View* vs, int K, int lid, hipStream_t s);
void launch_gram_upd_b(const View& v, const View* vs, int K, int lid, int going_left, int first_iter, hipStream_t s);
void launch_env_split_b(const View& v, const View* vs, int K, int lid, int going_left, int site, int left_side, int64_t prev_off, int prev_bond,
                        int out_bond, int64_t out_off, int chain, hipStream_t s);
void launch_eig_b(const View& v, const View* vs, int K, int lid, int going_left, int stage, hipStream_t s);
void launch_bt_assemble_b(const View& v, const View* vs, int K, int lid, hipStream_t s);
// one-shot direct-write all-reduce over peer-mapped inboxes (mpst_allreduce.hip)
constexpr int AR_MAX_RANKS = 8;
struct ArParams {
    double* inbox[AR_MAX_RANKS];             // base of every rank's inbox (own one included), as mapped in THIS process
    unsigned long long* flags[AR_MAX_RANKS]; // base of every rank's flag block: [2 parities][8 ranks]
    double* buf;                // local message, replaced by the sum
    int32_t* status;            // device status word (DevScalars.status), set on time-out
    int32_t* dbg;               // [2] on time-out: the rank whose flag was missing, the epoch its flag held (DevScalars.pad)
    unsigned int* counter;      // [2] local arrival counters
    int nranks, rank;
    int64_t slot;               // doubles per inbox slot
    unsigned long long epoch;   // 1, 2, 3, ... identical on every rank
    long long spin_limit;       // shader cycles a rank waits for its peers' flags before it gives up
    const int32_t* chi;         // live message length 2 + C * d*chi[lid] * d*chi[lid+2] when lid >= 0
    int lid, C, d;
    int64_t n_fixed;            // message length when lid < 0
};
void launch_allreduce_oneshot(const ArParams& p, hipStream_t s);
// launchers (mpst_kernels.hip)
void launch_bt_assemble(const View& v, int lid, hipStream_t s);
void launch_bt_prescale(const View& v, int lid, hipStream_t s);
void launch_yhat(const View& v, int lid, hipStream_t s);
void launch_grad(const View& v, int lid, hipStream_t s);
void launch_grad_reduce(const View& v, int lid, hipStream_t s);
void launch_update(const View& v, int lid, int first_iter, hipStream_t s);
void launch_trace_loss(const View& v, hipStream_t s);   // track_cost: tile losses of the last launch_yhat -> v.trace[v.trace_it]
void launch_gram(const View& v, int lid, int going_left, hipStream_t s);
void launch_split(const View& v, int lid, int going_left, hipStream_t s);
// env step: out[i][k] = sum_z Z_i[z] M[z][k], Z = (prev (x) phi_site) on the left
// (z = a*d + s) or (phi_site (x) prev) on the right (z = s*D + b).
// mode 0: M = kept eigenvectors v.E;  mode 1: M = site tensor as [(a,s)][k];
// mode 2: M = site tensor transposed, M[(s,b)][k] = W[k][(s,b)].
enum { ENV_M_E = 0, ENV_M_SITE = 1, ENV_M_SITE_T = 2 };
void launch_env(const View& v, int site, int left_side, const double* prev, int prev_bond,
                int mode, int out_bond, double* out, hipStream_t s, int bt_lid = -1);   // bt_lid: also assemble that bond's tensor
// device-side preprocessing + encoding (mpst_encode.hip)
struct EncDev {
    int64_t N;
    int32_t T, d;
    int32_t norm, sigmoid, minmax, is_test;
    int32_t fourier;        // 1: complex basis (Fourier, Stoudenmire, Sahand), phi holds (re, im) pairs
    int32_t basis;          // MPST_BASIS_*
    double med, s;          // robust sigmoid: 1 / (1 + exp(-(x - med) / s)), s = iqr / 1.35
    double lb, ub, a, b;    // data_bounds and the basis' input range
    double nrm;             // legendre: sqrt(Pl(1, d; normalized) * d)
    const double* lohi;     // device [2] min, max of the (sigmoid-transformed) training data
    const double* fix;      // device [N][2] per-series (shift, scale) of the out-of-bounds rescale, or null
};
size_t order_stats_temp_bytes(int64_t n);
hipError_t launch_order_stats(const double* X, double* sorted, void* temp, size_t temp_bytes, int64_t n, double* out3, hipStream_t s);
void launch_encode(const EncDev& e, const double* X, double* phi, double* part, double* lohi, double* fix, int fit_range,
                   hipStream_t s);
// per-device opt-in to large dynamic LDS (hipFuncSetAttribute applies to the current device only)
hipError_t init_kernel_attrs(int device);
hipError_t eig_init_attrs(int device);
void launch_eval_final(const View& v, const double* Lc, const double* Rc, double* yhat_out, hipStream_t s);
void launch_eval_reduce(const View& v, const double* yhat_in, double* out3, int64_t* conf, int32_t* pred, hipStream_t s);
void launch_norm2(const View& v, double* out_norm2, double* gscratch /* 3*cap*cap doubles */, hipStream_t s);
void launch_scale_sites(const View& v, const double* norm2, hipStream_t s);
void launch_selftest_mfma(const double* A, const double* B, int K, double* C, hipStream_t s);

// hand-written blocked eigensolver for d*chi_max > MAX_DIM (mpst_eig_blocked.hip); returns 1 when its on-device
// verification asks for the library fallback
struct BlockedEig;
int launch_eig_blocked_nosync(const View& v, int lid, int going_left, BlockedEig* e, hipStream_t s);   // no verdict read: see blocked_eig_take_sticky
int blocked_eig_take_sticky(BlockedEig* e, hipStream_t s);
void blocked_eig_force_sticky(BlockedEig* e, hipStream_t s);
int blocked_eig_coop_aborts(const BlockedEig* e);
int blocked_eig_xcd_misplaced(const BlockedEig* e);
int blocked_eig_create(BlockedEig** out, int ncap, std::string* err);
int blocked_eig_enable_subspace(BlockedEig* e, int mcap, int kcap, int C, int cx, std::string* err);
bool blocked_eig_subspace_on(const BlockedEig* e);
int blocked_eig_subspace_counts(BlockedEig* e, hipStream_t s, int32_t* attempted, int32_t* accepted);
void blocked_eig_destroy(BlockedEig* e);
int launch_eig_blocked(const View& v, int lid, int going_left, const double* rawG, int rawn, double* rawlam, double* rawE,
                       int32_t* rawinfo, BlockedEig* e, hipStream_t s);
// imputation engine (mpst_impute.hip)
hipError_t impute_init_attrs(int device);
// what the engine reads of a model: site tensors [a][s][b] (label site: one block per class) at a uniform stride, live bond
// dimensions, the encoded known values [T][N][d] and the class of every instance.  Elements are doubles / floats
// (compute_f32) or interleaved (re, im) pairs of them (is_complex).
struct ImpModel {
    const void* sites;
    int64_t site_stride;        // elements
    const int32_t* chi;         // [T+1]
    const int32_t* label_site;
    const void* phi;
    const int32_t* label;
    int64_t N;
    int32_t T, d, cap, is_complex, compute_f32;
};
struct ImputeParams {
    const uint8_t* missing;     // device [N][T]
    void* Rbuf;                 // device [chunk][max_missing][cap*cap] elements
    void* work;                 // device [chunk][4][cap*cap] elements when the bond dimension exceeds the LDS kernel's, else null
    const double *grid_x, *grid_phi, *u;    // grid_phi: [ngrid][d] doubles or (re, im) pairs
    double *pbuf, *sbuf, *x_out, *err_out;
    int max_missing, ngrid, method, get_wmad, rev, ntrial, mean_basis;
    double reject_thr;
    int trig;                   // the grid states are the Fourier basis on the uniform grid x0 + k dxu: densities in closed form
    double x0, dxu;
    const int32_t* order;       // device [N]: the order in which the instances are dealt out to workgroups
    const double* lin;          // device [2d-1][d][d]: Legendre linearisation table (trig, real models), else null
};
int impute_chi_limit(bool cx, bool f32);
int64_t impute_work_elems(int cap, bool cx, bool f32);     // per-instance scratch elements of the large-chi environment kernel
int launch_impute(const ImpModel& v, const ImputeParams& q, int64_t i0, int64_t count, hipStream_t s, hipEvent_t mid = nullptr);   // 1: the sweep ran sixteen instances per workgroup (k_imp_leftb)
// mpst_eig.hip
void launch_eig(const View& v, int lid, int going_left, int stage, hipStream_t s);   // stage 0 tri (or tri + vec merged), 1 vec, 2 fin
void launch_eig_tail(const View& v, int lid, int going_left, int rawn, const double* Gt, int ld, double* Vall, double* dd, double* ee,
                     double* tau, const int32_t* abort_flag, const int32_t* skip, hipStream_t s);   // last 128 steps of a larger reduction on one CU
bool eig_merged();   // k_eig_trivec instead of k_eig_tri + k_eig_vec (default; MPST_EIG_SPLIT=1 restores the three-kernel chain)
void launch_eig_raw(const double* G, int n, int alg, double* lam, double* E, int32_t* info, double* ws, hipStream_t s);
void launch_eig_raw_gated(const double* G, int n, double* lam, double* E, int32_t* info, double* ws, const int32_t* gate, hipStream_t s);
size_t eig_workspace_doubles();
// robust slow path for d*chi_max > MAX_DIM (multi-workgroup one-sided Jacobi at the capacity size, see mpst_eig.hip; no vendor solver is linked)
struct BigEig;
int big_eig_create(BigEig** out, int ncap, hipStream_t s, std::string* err);
void big_eig_destroy(BigEig* b);
int launch_eig_big(const View& v, int lid, int going_left, BigEig* b, hipStream_t s);
int launch_eig_big_raw(const double* G, int n, double* lam, double* E, int32_t* info, BigEig* b, hipStream_t s);

}  // namespace mpst
