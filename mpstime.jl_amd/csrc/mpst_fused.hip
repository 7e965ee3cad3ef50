// Fused per-bond kernels for bond tensors up to 128 x 128 (the headline shapes): 7 dependent launches per bond
// instead of 10, a gradient workspace whose size does not depend on N, and the bond tensor held in registers.
//
//   k_bond_fused   flatten_bt (RealRealHighDimension.jl:221-238) + the fused phi-tilde / yhat / gradient loop
//                  (loss_functions.jl:248-262,322-379 / :435-531,561-619).  One persistent workgroup per PART - a
//                  class-pure run of series - keeps its class' B_c in registers (column tile per wave), walks its
//                  series 16 at a time (yhat on the MFMA, weights w_i, rank-16 update of the 128 x 128 gradient
//                  accumulators in registers) and writes ONE partial gradient.  #parts is fixed (<= PARTS_TARGET).
//   k_fused_reduce grad[c] = scale_c * sum of the class' partials, loss, and per-workgroup pieces of ||grad||^2
//   k_gram_upd     TSGO / GD step (loss_functions.jl:49,79) applied on the fly to the operands of the Gram matrix of
//                  decomposeBT's matrix (RealRealHighDimension.jl:166-169,185-188); the diagonal tiles write bt_new
//   k_env_split    update_caches! (:107-144) and the back-split of decomposeBT (:172-176,190-194) in one launch:
//                  both depend only on the kept eigenvectors E and bt_new
#include "mpst_internal.h"
#include "mpst_eig_common.inl"
#include <algorithm>
#include <cstring>

namespace mpst {

struct BondDimsF {
    int Dl, Dm, Dr, X, Y, L;
};
__device__ __forceinline__ BondDimsF bond_dims_f(const View& v, int lid) {
    BondDimsF b;
    b.Dl = v.chi[lid];
    b.Dm = v.chi[lid + 1];
    b.Dr = v.chi[lid + 2];
    b.X = b.Dl * v.d;
    b.Y = v.d * b.Dr;
    b.L = b.X * b.Y;
    return b;
}

constexpr int FUSED_T = 512;          // 8 waves: wave w owns the 16 columns [16w, 16w+16) of B_c and of the gradient
constexpr int FXS = MAX_DIM + 2;      // LDS row stride of the Khatri-Rao tiles

// Khatri-Rao tile of 16 series by 256 threads (16 per series), as stage_tile16 of mpst_kernels.hip, split into the
// global loads (issued early, results parked in registers) and the products written to LDS; `t` = thread index
// within the 256.
template <int DM>
struct StageRegs {
    double pa[8], pp[DM];
};
template <int DM>
__device__ __forceinline__ void stage16_load(StageRegs<DM>& r, const int32_t start, const int32_t count, const double* __restrict__ prev,
                                             int Dp, const double* __restrict__ ph, int d, int cap, int t) {
    const int i = t >> 4, j = t & 15;
    const bool valid = i < count;
    const int64_t smp = start + (valid ? i : 0);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int a = j + 16 * m;
        r.pa[m] = (valid && a < Dp) ? (prev ? prev[smp * cap + a] : 1.0) : 0.0;
    }
#pragma unroll
    for (int s = 0; s < DM; ++s) r.pp[s] = (valid && s < d) ? ph[smp * d + s] : 0.0;
}
template <int DM>
__device__ __forceinline__ void stage16_store(const StageRegs<DM>& r, double* __restrict__ Zs, const int32_t start, const int32_t count,
                                              int Dp, const double* __restrict__ ph, int d, bool left, int t) {
    const int i = t >> 4, j = t & 15;
    const bool valid = i < count;
    const int64_t smp = start + (valid ? i : 0);
    double* row = Zs + i * FXS;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int a = j + 16 * m;
        if (a < Dp) {
#pragma unroll
            for (int s = 0; s < DM; ++s)
                if (s < d) row[left ? a * d + s : s * Dp + a] = r.pa[m] * r.pp[s];
            for (int s = DM; s < d; ++s) row[left ? a * d + s : s * Dp + a] = valid ? r.pa[m] * ph[smp * d + s] : 0.0;
        }
    }
    for (int z = Dp * d + j; z < MAX_DIM; z += 16) row[z] = 0.0;
}
__device__ __forceinline__ void stage16(double* __restrict__ Zs, const int32_t start, const int32_t count,
                                        const double* __restrict__ prev, int Dp, const double* __restrict__ ph, int d, int cap,
                                        bool left, int t) {
    StageRegs<8> r;
    stage16_load(r, start, count, prev, Dp, ph, d, cap, t);
    stage16_store(r, Zs, start, count, Dp, ph, d, left, t);
}

constexpr int WLS = 34;     // LDS row stride of the staged 32-column slice of W[lid]

template <int DM>
__global__ __launch_bounds__(FUSED_T) void k_bond_fused(View v, int lid, int assemble) {
    __shared__ __attribute__((aligned(16))) double Xs[16 * FXS];
    __shared__ __attribute__((aligned(16))) double Ys[16 * FXS];
    __shared__ __attribute__((aligned(16))) double Wls[MAX_DIM * WLS];
    __shared__ double red[8 * 16];
    const BondDimsF b = bond_dims_f(v, lid);
    const Part pt = v.parts[blockIdx.x];
    const int d = v.d, rid = lid + 1;
    const bool mse = v.loss == MPST_LOSS_MSE;
    const int c = pt.cls;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, kq = lane >> 4;
    const int col = 16 * wave + i16;
    const bool cv = col < b.Y;
#ifdef MPST_FUSED_DEBUG
    // bring-up only: 100 MHz stamps of workgroup 0 (read with mpst_debug_stamps, slots 32..)
    unsigned long long* dbg = (blockIdx.x == 0 && tid == 0) ? v.sc->eig_stamps + 32 : nullptr;
    int dbgi = 0;
#define FSTAMP() do { if (dbg && dbgi < 30) dbg[dbgi++] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define FSTAMP() do { } while (0)
#endif
    FSTAMP();
    const double* LEp = lid > 0 ? v.LE + (int64_t)(lid - 1) * v.N * v.cap : nullptr;
    const double* REn = rid < v.T - 1 ? v.RE + (int64_t)(rid + 1) * v.N * v.cap : nullptr;
    const double* phl = v.phi + (int64_t)lid * v.N * d;
    const double* phr = v.phi + (int64_t)rid * v.N * d;
    const bool xhalf = tid < 256;                 // threads 0..255 stage the X tile, 256..511 the Y tile
    const int st = xhalf ? tid : tid - 256;
    const double* s_prev = xhalf ? LEp : REn;
    const double* s_ph = xhalf ? phl : phr;
    const int s_D = xhalf ? b.Dl : b.Dr;
    double* s_dst = xhalf ? Xs : Ys;
    // the first tile's environment rows and site vectors are requested before anything else: their latency is
    // covered by the assembly of the bond tensor
    StageRegs<DM> sr;
    stage16_load(sr, pt.start, min(16, pt.count), s_prev, s_D, s_ph, d, v.cap, st);

    // ---- B_c column tile of this wave: bq[mt][r] = B_c[16mt + kq + 4r][col] (k-step u = 4mt + r of the yhat product) ----
    d4 bq[8];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) bq[mt] = d4{0.0, 0.0, 0.0, 0.0};
    if (assemble) {
        // flatten_bt: B_c = W[lid] W[rid] (the label index sits on one of the two sites); W[lid] goes through LDS
        // 32 columns of the shared bond at a time, this wave's 16 columns of W[rid] through registers
        const int ls = *v.label_site;
        const int cl = (ls == lid) ? c : 0, cr = (ls == rid) ? c : 0;
        const double* Wl = v.sites + (int64_t)lid * v.site_stride + (int64_t)cl * b.X * b.Dm;   // [x][m]
        const double* Wr = v.sites + (int64_t)rid * v.site_stride + (int64_t)cr * b.Dm * b.Y;   // [m][y]
        for (int m0 = 0; m0 < b.Dm; m0 += 32) {
            double wr[8], wlv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int idx = tid + FUSED_T * j, x = idx >> 5, m = m0 + (idx & 31);
                wlv[j] = (x < b.X && m < b.Dm) ? Wl[(int64_t)x * b.Dm + m] : 0.0;
            }
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int m = m0 + 4 * s + kq;
                wr[s] = (cv && m < b.Dm) ? Wr[(int64_t)m * b.Y + col] : 0.0;
            }
            if (m0) __syncthreads();
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int idx = tid + FUSED_T * j;
                Wls[(idx >> 5) * WLS + (idx & 31)] = wlv[j];
            }
            __syncthreads();
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) {
                if (16 * mt < b.X) {
#pragma unroll
                    for (int s = 0; s < 8; ++s)
                        if (m0 + 4 * s < b.Dm) bq[mt] = mfma_f64(Wls[(16 * mt + i16) * WLS + 4 * s + kq], wr[s], bq[mt]);
                }
            }
        }
        if (pt.first_of_cls) {
            double* out = v.bt + (int64_t)c * b.L;
#pragma unroll
            for (int mt = 0; mt < 8; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int x = 16 * mt + kq + 4 * r;
                    if (cv && x < b.X) out[(int64_t)x * b.Y + col] = bq[mt][r];
                }
        }
    } else {
        const double* Bc = v.bt + (int64_t)c * b.L;
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int x = 16 * mt + kq + 4 * r;
                bq[mt][r] = (cv && x < b.X) ? Bc[(int64_t)x * b.Y + col] : 0.0;
            }
    }
    FSTAMP();
    const int XP = (b.X + 3) & ~3;
    const int nmt = (b.X + 15) >> 4;

    d4 gacc[8];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) gacc[mt] = d4{0.0, 0.0, 0.0, 0.0};
    double loss = 0.0;
    const double delta = (pt.own == c) ? 1.0 : 0.0;

    for (int t0 = 0; t0 < pt.count; t0 += 16) {
        const int cnt = min(16, pt.count - t0);
        __syncthreads();                                   // the previous tile's operands are consumed
        stage16_store(sr, s_dst, pt.start + t0, cnt, s_D, s_ph, d, xhalf, st);
        __syncthreads();
        // the next tile's loads fly during this tile's matrix work
        if (t0 + 16 < pt.count) stage16_load(sr, pt.start + t0 + 16, min(16, pt.count - t0 - 16), s_prev, s_D, s_ph, d, v.cap, st);
        FSTAMP();
        // ---- yhat_i = X_i^T B_c Y_i: this wave's 16 columns (two accumulation chains) ---------------------------
        d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int u = 0; u < 32; u += 2) {
            // (unconditional: the staged rows are zero beyond the live extent - an exact +0; a predicate per MFMA costs a copy of the
            // accumulator or a branch that keeps the operand reads from being batched)
            acc0 = mfma_f64(Xs[i16 * FXS + 4 * u + kq], bq[u >> 2][u & 3], acc0);
            acc1 = mfma_f64(Xs[i16 * FXS + 4 * u + 4 + kq], bq[(u + 1) >> 2][(u + 1) & 3], acc1);
            // at most 8 operand reads ahead of the matrix pipe: the whole tile's worth (64 VGPRs) would spill
            if ((u & 7) == 6) asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double x = sum16((acc0[r] + acc1[r]) * Ys[(kq + 4 * r) * FXS + col]);
            if (i16 == 0) red[wave * 16 + kq + 4 * r] = x;
        }
        FSTAMP();
        __syncthreads();
        FSTAMP();
        // every wave forms the 16 weights itself (same fixed-order sum -> same bits), lane j < 16 holds series j
        double wj = 0.0;
        {
            const int j = lane & 15;
            const double yh = ((red[j] + red[16 + j]) + (red[32 + j] + red[48 + j])) + ((red[64 + j] + red[80 + j]) + (red[96 + j] + red[112 + j]));
            if (j < cnt) {
                wj = mse ? (yh - delta) : 1.0 / yh;                                  // :489,:608 / :258,:367
                if (wave == 0 && lane < 16) loss += mse ? 0.5 * (yh - delta) * (yh - delta) : -log(yh * yh);   // :554 / :318
            }
        }
        // ---- G_c += sum_i w_i X_i Y_i^T: 8 row tiles x this wave's column tile, K = 16 series -------------------
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = 4 * u + kq;
            const double wi = __shfl(wj, i, 64);
            const double bop = wi * Ys[i * FXS + col];
#pragma unroll
            for (int mt = 0; mt < 8; ++mt)
                gacc[mt] = mfma_f64(Xs[i * FXS + 16 * mt + i16], bop, gacc[mt]);      // (rows beyond the live tiles are zero)
            asm volatile("" ::: "memory");
        }
        FSTAMP();
    }
    // ---- one partial per part ------------------------------------------------------------------------------------
    double* out = v.partial + (int64_t)blockIdx.x * b.L;
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * mt + kq + 4 * r;
            if (row < b.X && cv) out[(int64_t)row * b.Y + col] = gacc[mt][r];
        }
    }
    if (wave == 0) {
        loss = sum16(loss);
        if (lane == 0) v.tile_loss[blockIdx.x] = loss;
    }
    FSTAMP();
}

// Deterministic block-wide sum (fixed tree) for 256 threads.
__device__ __forceinline__ double block_sum256(double x, double* red) {
    x = wave_sum(x);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = x;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// grad[c] = scale_c * sum of the partials of class c (fixed order); gradbuf = [loss, 0, grad...] is what the multi-GPU
// all-reduce sums.  64 gradient entries per workgroup, wave g sums the partials k = k0 + g, k0 + g + 4, ... (16 loads
// in flight), the four wave sums meet in LDS: (g0 + g1) + (g2 + g3).  norm_part[block] = sum of squares of the block's
// 64 entries (single GPU: pieces of the norm of the complete gradient; with a communicator k_grad_norm recomputes
// them after the all-reduce).
constexpr int RED_E = 64;
__global__ __launch_bounds__(256) void k_fused_reduce(View v, int lid) {
    __shared__ double red[4];
    __shared__ double part[4][RED_E];
    const BondDimsF b = bond_dims_f(v, lid);
    const bool mse = v.loss == MPST_LOSS_MSE;
    const int64_t total = (int64_t)v.C * b.L;
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int64_t idx = (int64_t)blockIdx.x * RED_E + lane;
    double s = 0.0;
    int c = 0;
    if (idx < total) {
        c = (int)(idx / b.L);
        const int64_t e = idx - (int64_t)c * b.L;
        const int k0 = v.part_off[c], k1 = v.part_off[c + 1];
        const double* p = v.partial + e;
        double acc[4] = {0, 0, 0, 0};
        for (int k = k0 + g; k < k1; k += 64) {
            double t[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) t[u] = (k + 4 * u < k1) ? p[(int64_t)(k + 4 * u) * b.L] : 0.0;
#pragma unroll
            for (int u = 0; u < 16; ++u) acc[u & 3] += t[u];
        }
        s = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    }
    part[g][lane] = s;
    __syncthreads();
    double gsum = 0.0;
    if (g == 0) {
        if (idx < total) {
            const double scale = mse ? v.invN : -(v.train_sep ? v.inv_count[c] : v.invN);     // :608 / :367,:424
            gsum = ((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane])) * scale;
            v.gradbuf[2 + idx] = gsum;
        }
        const double n2 = wave_sum(gsum * gsum);
        if (lane == 0) v.norm_part[blockIdx.x] = n2;
    }
    if (blockIdx.x == 0) {
        double l = 0.0;
        for (int i = threadIdx.x; i < v.nparts; i += 256) {
            const Part pt = v.parts[i];
            const double w = mse ? v.invN : (v.train_sep ? v.inv_count[pt.own] : v.invN);   // :612 / :423,:371
            l += v.tile_loss[i] * w;
        }
        const double tot = block_sum256(l, red);
        if (threadIdx.x == 0) {
            v.gradbuf[0] = tot;
            v.gradbuf[1] = 0.0;
        }
    }
}

// ||grad||^2 pieces of the (all-reduced) gradient, same slicing as k_fused_reduce
__global__ __launch_bounds__(64) void k_grad_norm(View v, int lid) {
    const BondDimsF b = bond_dims_f(v, lid);
    const int64_t total = (int64_t)v.C * b.L;
    const int64_t idx = (int64_t)blockIdx.x * RED_E + threadIdx.x;
    const double g = idx < total ? v.gradbuf[2 + idx] : 0.0;
    const double n2 = wave_sum(g * g);
    if (threadIdx.x == 0) v.norm_part[blockIdx.x] = n2;
}

// step of TSGO / GD from the norm pieces: every caller sums them in the same order, so all see the same bits
// Workgroups are dealt round-robin over the 8 XCDs in launch order (observed, speed only): the p-th workgroup of the launch lands on
// XCD p % 8.  The blocks of one (fit, class, share) read the SAME series, a different 64-byte piece of every environment row each: dealt
// in launch order they sit on all 8 XCDs and every XCD's L2 pulls every series from memory (K = 8: 17x the algorithmic bytes).  The
// batched launch therefore gives XCD x the x-th CONTIGUOUS eighth of the logical order (blocks fastest, then shares, classes, fits): the
// blocks of a share run side by side on one XCD and meet in its L2.  Which workgroup computes what does not enter any sum.
__device__ __forceinline__ void xcd_contiguous(int& bx, int& by, int& bz) {
    const int gx = (int)gridDim.x, gy = (int)gridDim.y, gz = (int)gridDim.z;
    const int G = gx * gy * gz;
    if (G & 7) return;
    const int p = (int)blockIdx.x + gx * ((int)blockIdx.y + gy * (int)blockIdx.z);
    const int l = (p & 7) * (G >> 3) + (p >> 3);
    bx = l % gx;
    by = (l / gx) % gy;
    bz = l / (gx * gy);
}
__device__ __forceinline__ double step_from_parts(const View& v, double* red, double* nrm_out) {
    double s = 0.0;
    for (int i = threadIdx.x; i < v.n_norm_part; i += 256) s += v.norm_part[i];
    const double nrm2 = block_sum256(s, red);
    const double nrm = sqrt(nrm2);
    if (nrm_out) *nrm_out = nrm;
    return (v.optimiser == MPST_OPT_TSGO) ? v.eta / nrm : v.eta;
}

// Gram matrix of bt_new = bt - step*grad viewed as the matrix decomposeBT hands to svd, one 16 x 16 tile per
// workgroup, K split over the 4 waves.  The diagonal tiles see every entry of bt_new exactly once (their A-operand
// panel) and write it to v.btn - a second buffer, the other tiles are still reading the old one.
__device__ __forceinline__ void gram_upd_body(const View& v, int lid, int going_left, int first_iter, const int tile) {
    __shared__ double part[4][256];
    __shared__ double red[4];
    const BondDimsF b = bond_dims_f(v, lid);
    const int n = going_left ? b.Y : b.X;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tn = (n + 15) >> 4;
    const bool live = tile < tn * tn;
    const int m0 = (tile / tn) * 16, n0 = (tile % tn) * 16;
    const bool diag = m0 == n0;
    const int i = lane & 15, kq = lane >> 4;
    const int m = m0 + i, nn = n0 + i;
    const bool mv = live && m < n, nv = live && nn < n;
    const double* g = v.gradbuf + 2;
    // element (k, p) of the matrix: going left  k = (c, x) jointly, p = y:  bt[k*Y + p]
    //                               going right k = y, p = x, per class:    bt[c*L + p*Y + k]
    const int nc = going_left ? 1 : v.C;
    const int K = going_left ? v.C * b.X : b.Y;
    const int kq4 = (((K + 3) >> 2) + 3) & ~3;
    const int kbeg = wave * kq4, kend = min(K, (wave + 1) * kq4);
    const int64_t sa_m = going_left ? 1 : b.Y, sa_k = going_left ? b.Y : 1;
    // The first batch of operands (all of them for the headline shapes) is requested BEFORE the step size is formed:
    // the norm reduction (a dependent global round trip + two barriers) then overlaps the operand latency.
    double a_bt[16], a_g[16], b_bt[16], b_g[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int k = kbeg + 4 * u + kq;
        const bool kv = k < kend;
        const int64_t ia = (int64_t)m * sa_m + (int64_t)k * sa_k, ib = (int64_t)nn * sa_m + (int64_t)k * sa_k;
        a_bt[u] = (mv && kv) ? v.bt[ia] : 0.0;
        a_g[u] = (mv && kv) ? g[ia] : 0.0;
        b_bt[u] = (nv && kv) ? v.bt[ib] : 0.0;
        b_g[u] = (nv && kv) ? g[ib] : 0.0;
    }
    double nrm;
    const double step = step_from_parts(v, red, &nrm);
    if (tile == 0 && threadIdx.x == 0) {
        if (first_iter) {
            v.sc->loss = bond_loss(v);
            v.sc->grad_norm = nrm;
        }
        if (v.trace) v.trace[v.trace_it] = bond_loss(v);      // "Loss before step i" of the last iteration
    }
    if (!live) return;
    d4 acc = {0, 0, 0, 0};
    for (int c = 0; c < nc; ++c) {
        const int64_t base = (int64_t)c * b.L;
        for (int k0 = kbeg; k0 < kend; k0 += 64) {
            double a[16], bb[16];
            int64_t ia[16];
            const bool first = c == 0 && k0 == kbeg;
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int k = k0 + 4 * u + kq;
                const bool kv = k < kend;
                ia[u] = base + (int64_t)m * sa_m + (int64_t)k * sa_k;
                const int64_t ib = base + (int64_t)nn * sa_m + (int64_t)k * sa_k;
                if (first) {
                    a[u] = fma(-step, a_g[u], a_bt[u]);
                    bb[u] = fma(-step, b_g[u], b_bt[u]);
                } else {
                    a[u] = (mv && kv) ? fma(-step, g[ia[u]], v.bt[ia[u]]) : 0.0;
                    bb[u] = (nv && kv) ? fma(-step, g[ib], v.bt[ib]) : 0.0;
                }
            }
            if (diag) {
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    if (mv && k0 + 4 * u + kq < kend) v.btn[ia[u]] = a[u];
                if (v.btnT && !going_left) {        // (m = x, k = y): the same entries as [c][y][x] for k_bond_tail
#pragma unroll
                    for (int u = 0; u < 16; ++u)
                        if (mv && k0 + 4 * u + kq < kend) v.btnT[base + (int64_t)(k0 + 4 * u + kq) * b.X + m] = a[u];
                }
            }
#pragma unroll
            for (int u = 0; u < 16; ++u)
                acc = mfma_f64(a[u], bb[u], acc);           // (operands beyond the contraction are zero: an exact +0, no predicate)
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) part[wave][r * 64 + lane] = acc[r];
    __syncthreads();
    if (wave == 0) {
        const int col = n0 + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + (lane >> 4) + 4 * r;
            const double sum = (part[0][r * 64 + lane] + part[1][r * 64 + lane]) + (part[2][r * 64 + lane] + part[3][r * 64 + lane]);
            if (row < n && col < n) v.gram[(int64_t)row * n + col] = sum;
        }
    }
}

// ---- environment update + back-split in one launch ---------------------------------------------------------------
// wave-level 16 x 16 tile with operands from global / L2 (as wave_gemm_tile of mpst_kernels.hip)
__device__ __forceinline__ d4 gemm_tile_g(const double* __restrict__ A, int64_t sam, int64_t sak, int M,
                                          const double* __restrict__ B, int64_t sbk, int64_t sbn, int N, int K, int m0, int n0) {
    const int lane = threadIdx.x & 63;
    const int i = lane & 15, kq = lane >> 4;
    const int m = m0 + i, n = n0 + i;
    const bool mv = m < M, nv = n < N;
    const double* ap = A + (int64_t)m * sam;
    const double* bp = B + (int64_t)n * sbn;
    d4 acc = {0, 0, 0, 0};
    for (int k0 = 0; k0 < K; k0 += 64) {
        double a[16], b[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int k = k0 + 4 * u + kq;
            const bool kv = k < K;
            a[u] = (mv && kv) ? ap[(int64_t)k * sak] : 0.0;
            b[u] = (nv && kv) ? bp[(int64_t)k * sbk] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u)
            acc = mfma_f64(a[u], b[u], acc);
    }
    return acc;
}

// The same tile with the contraction cut into the four ranges chain_bt_block gives its four waves, the partial tiles
// added as ((p0 + p1) + (p2 + p3)): the site tensor the back-split stores is then bit for bit the T that
// chain_bt_block feeds into the next bond tensor, and a sweep replayed from the graph (chained tensors) equals the same
// sweep stepped bond by bond (tensor re-assembled from the stored sites) in every bit.
__device__ __forceinline__ d4 gemm_tile_g4(const double* __restrict__ A, int64_t sam, int64_t sak, int M,
                                           const double* __restrict__ B, int64_t sbk, int64_t sbn, int N, int K, int m0, int n0) {
    const int lane = threadIdx.x & 63;
    const int i = lane & 15, kq = lane >> 4;
    const int m = m0 + i, n = n0 + i;
    const bool mv = m < M, nv = n < N;
    const double* ap = A + (int64_t)m * sam;
    const double* bp = B + (int64_t)n * sbn;
    const int ks4 = (((K + 3) >> 2) + 3) >> 2;         // k-steps per range, as in chain_bt_block
    d4 p[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        p[w] = d4{0, 0, 0, 0};
        const int kbeg = 4 * ks4 * w, kend = min(K, 4 * ks4 * (w + 1));
        for (int q0 = kbeg; q0 < kend; q0 += 32) {
            double a[8], b[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = q0 + 4 * u + kq;
                const bool kv = k < kend;
                a[u] = (mv && kv) ? ap[(int64_t)k * sak] : 0.0;
                b[u] = (nv && kv) ? bp[(int64_t)k * sbk] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                p[w] = mfma_f64(a[u], b[u], p[w]);
        }
    }
    d4 acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = (p[0][r] + p[1][r]) + (p[2][r] + p[3][r]);
    return acc;
}

// decomposeBT back-split from the kept eigenvectors E (as k_split, reading bt_new from v.btn); `blk` of `nblk` workgroups
// (Ev / ldE / nk / inv: the kept eigenvectors, their row stride, how many were kept and 1/||bt_new|| - from memory as k_eig_fin left them,
// or from the workgroup's own LDS copy in k_bond_tail)
__device__ __forceinline__ void split_block(const View& v, int lid, int going_left, int blk, int nblk, const double* __restrict__ Ev, const int ldE,
                                            const int nk, const double inv) {
    const BondDimsF b = bond_dims_f(v, lid);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double* Wl = v.sites + (int64_t)lid * v.site_stride;
    double* Wr = v.sites + (int64_t)(lid + 1) * v.site_stride;
    const int tk = (nk + 15) >> 4;
    const int col = lane & 15, rq = lane >> 4;
    if (going_left) {
        const int tx = (b.X + 15) >> 4;
        const int ntile = v.C * tx * tk;
        for (int tile = blk * 4 + wave; tile < ntile; tile += nblk * 4) {
            const int c = tile / (tx * tk), rem = tile - c * tx * tk;
            const int m0 = (rem / tk) * 16, n0 = (rem % tk) * 16;
            const double* Bc = v.btn + (int64_t)c * b.L;
            const d4 acc = gemm_tile_g4(Bc, b.Y, 1, b.X, Ev, ldE, 1, nk, b.Y, m0, n0);
            double* out = Wl + (int64_t)c * b.X * nk;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + rq + 4 * r;
                if (row < b.X && n0 + col < nk) out[(int64_t)row * nk + n0 + col] = acc[r] * inv;
            }
        }
        for (int i = blk * 256 + threadIdx.x; i < nk * b.Y; i += nblk * 256) {
            const int k = i / b.Y, y = i - k * b.Y;
            Wr[i] = Ev[(int64_t)y * ldE + k];
        }
        if (blk == 0 && threadIdx.x == 0) *v.label_site = lid;
    } else {
        const int ty = (b.Y + 15) >> 4;
        const int ntile = v.C * tk * ty;
        for (int tile = blk * 4 + wave; tile < ntile; tile += nblk * 4) {
            const int c = tile / (tk * ty), rem = tile - c * tk * ty;
            const int m0 = (rem / ty) * 16, n0 = (rem % ty) * 16;
            const double* Bc = v.btn + (int64_t)c * b.L;
            const d4 acc = gemm_tile_g4(Ev, 1, ldE, nk, Bc, b.Y, 1, b.Y, b.X, m0, n0);
            double* out = Wr + (int64_t)c * nk * b.Y;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + rq + 4 * r;
                if (row < nk && n0 + col < b.Y) out[(int64_t)row * b.Y + n0 + col] = acc[r] * inv;
            }
        }
        for (int i = blk * 256 + threadIdx.x; i < b.X * nk; i += nblk * 256) {
            const int x = i / nk, k = i - x * nk;
            Wl[i] = Ev[(int64_t)x * ldE + k];
        }
        if (blk == 0 && threadIdx.x == 0) *v.label_site = lid + 1;
    }
}

// flatten_bt of the NEXT bond (RealRealHighDimension.jl:221-238) without waiting for the back-split to land in
// memory: with T = bt_new * E (the site that keeps the label, up to 1/||bt_new||) the next bond tensor is
//   going left : bt'[c][(a',s')][(s,k)]  = sum_a W[lid-1][(a',s')][a] * T_c[(a,s)][k]
//   going right: bt'[c][(k,s)][(s'',b'')] = sum_b T_c[k][(s,b)] * W[lid+2][b][(s'',b'')]
// One workgroup per (class c, site value s, 16 kept states k): each wave forms the 16 x 16 tiles of T for that (s, k)
// block on the MFMA and feeds them straight back as operand of the second product - register r of a 16 x 16 fp64
// accumulator tile holds rows 4r .. 4r+3 (lane>>4) of the tile, which is exactly the B (or, transposed, the A) operand
// of k-step r - then the waves share out the row (column) tiles of bt'.
constexpr int CHAIN_J = 4;      // 16-row blocks of the contracted bond: chi <= 64 whenever d*chi <= 128 and d >= 2
__device__ __forceinline__ void chain_bt_block(const View& v, int lid, int going_left, int job, double* __restrict__ cpart /* [4][CHAIN_J][256] */,
                                               const double* __restrict__ Ev, const int ldE, const int nk, const double inv) {
    const BondDimsF b = bond_dims_f(v, lid);
    const int d = v.d;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i16 = lane & 15, kq = lane >> 4;
    const int ktc = (v.cap + 15) >> 4;                 // jobs are laid out for the capacity
    const int kt = job % ktc, s = (job / ktc) % d, c = job / (ktc * d);
    const int k0 = 16 * kt;
    if (k0 >= nk || c >= v.C) return;
    const int kcol = k0 + i16;
    const bool kv = kcol < nk;
    const double* Bc = v.btn + (int64_t)c * b.L;
    // ---- T tiles: the contraction index (y going left, x going right) is split over the 4 waves, partial tiles meet in LDS
    const int Kc = going_left ? b.Y : b.X;             // contraction length of the first product
    const int Dc = going_left ? b.Dl : b.Dr;           // rows of T: the bond shared with the neighbouring site
    const int nj = (Dc + 15) >> 4;
    // The neighbouring site's tensor - the other operand of the SECOND product - does not depend on T: it is requested now,
    // so that the launch pays one round trip to memory for its operands instead of two in a row.  A wave owns the row
    // (column) tiles wave and wave + 4 of bt' (d*chi <= 128: at most 8 tiles).
    const int Dnb = going_left ? v.chi[lid - 1] : v.chi[lid + 3];           // the neighbouring site's outer bond
    double wpre[2][CHAIN_J][4];
    {
        const double* Wn = v.sites + (int64_t)(going_left ? lid - 1 : lid + 2) * v.site_stride;
        const int Xn = Dnb * d;                        // rows of bt' going left (x') / columns going right (y'')
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int tl = 16 * (wave + 4 * it) + i16;
#pragma unroll
            for (int j = 0; j < CHAIN_J; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int a = 16 * j + 4 * r + kq;
                    const bool ok = j < nj && tl < Xn && a < Dc;
                    wpre[it][j][r] = ok ? (going_left ? Wn[(int64_t)tl * b.Dl + a] : Wn[(int64_t)a * Xn + tl]) : 0.0;
                }
        }
    }
    const int ks4 = (((Kc + 3) >> 2) + 3) >> 2;         // k-steps per wave
    const int kbeg = 4 * ks4 * wave, kend = min(Kc, 4 * ks4 * (wave + 1));
#pragma unroll
    for (int j = 0; j < CHAIN_J; ++j) {
        d4 t = {0, 0, 0, 0};
        if (j < nj) {
            const int row = 16 * j + i16;                // a (going left) or b (going right)
            const double* ap = going_left ? Bc + (int64_t)(row * d + s) * b.Y : Bc + s * b.Dr + row;
            const int64_t astr = going_left ? 1 : b.Y;
            for (int q0 = kbeg; q0 < kend; q0 += 32) {
                double av[8], bv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int q = q0 + 4 * u + kq;
                    av[u] = (row < Dc && q < kend) ? ap[(int64_t)q * astr] : 0.0;
                    bv[u] = (kv && q < kend) ? Ev[(int64_t)q * ldE + kcol] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    t = mfma_f64(av[u], bv[u], t);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) cpart[(wave * CHAIN_J + j) * 256 + r * 64 + lane] = t[r];
    }
    __syncthreads();
    d4 T[CHAIN_J];
#pragma unroll
    for (int j = 0; j < CHAIN_J; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = j * 256 + r * 64 + lane;
            T[j][r] = ((cpart[o] + cpart[CHAIN_J * 256 + o]) + (cpart[2 * CHAIN_J * 256 + o] + cpart[3 * CHAIN_J * 256 + o])) * inv;
        }
    if (going_left) {
        const int Xp = Dnb * d, Yp = d * nk;
        const int64_t Lp = (int64_t)Xp * Yp;
        double* out = v.bt + (int64_t)c * Lp;
        const int ntx = (Xp + 15) >> 4;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int mt = wave + 4 * it;
            if (mt >= ntx) break;
            d4 acc = {0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < CHAIN_J; ++j)
                if (j < nj) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc = mfma_f64(wpre[it][j][r], T[j][r], acc);
                }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int xr = 16 * mt + kq + 4 * r;
                if (xr < Xp && kv) out[(int64_t)xr * Yp + s * nk + kcol] = acc[r];
            }
        }
    } else {
        const int Ypp = d * Dnb, Xp = nk * d;
        const int64_t Lp = (int64_t)Xp * Ypp;
        double* out = v.bt + (int64_t)c * Lp;
        const int nty = (Ypp + 15) >> 4;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int nt = wave + 4 * it;
            if (nt >= nty) break;
            const int y = 16 * nt + i16;
            d4 acc = {0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < CHAIN_J; ++j)
                if (j < nj) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc = mfma_f64(T[j][r], wpre[it][j][r], acc);    // A[m = k][kk = b] = T^T tile, register r
                }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = k0 + kq + 4 * r;
                if (k < nk && y < Ypp) out[(int64_t)(k * d + s) * Ypp + y] = acc[r];
            }
        }
    }
}

// blocks [0, ntiles): new environment rows out_i = Z_i E (update_caches!); [ntiles, ntiles + nsplit): the back-split;
// beyond: the next bond's tensor (chain_bt_block)
__device__ __forceinline__ void env_split_body(const View& v, int lid, int going_left, int site, int left_side,
                                               const double* __restrict__ prev, int prev_bond, int out_bond,
                                               double* __restrict__ out, int nsplit, int ntb, int tp, const int bid) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    if (bid >= ntb + nsplit) {
        chain_bt_block(v, lid, going_left, bid - ntb - nsplit, smem, v.E, v.cap, v.sc->n_keep, v.sc->inv_norm);
        return;
    }
    if (bid >= ntb) {
        split_block(v, lid, going_left, bid - ntb, nsplit, v.E, v.cap, v.sc->n_keep, v.sc->inv_norm);
        return;
    }
    // new environment rows out_i = Z_i E.  A wave owns one 16-column tile of the output; with fewer than four column tiles
    // (capacity <= 32) and enough tiles to go round, the workgroup takes two 16-series tiles at a time (tp, chosen by the
    // launcher) so that no wave idles, and it keeps walking
    // tiles (stride = number of tile blocks) with its E operand - 32 doubles per lane - loaded once: at N = 32768 one
    // workgroup per tile spent its time launching and re-reading E (33 us for 17 MB of environment traffic).
    const int d = v.d;
    const int Dp = prev ? v.chi[prev_bond] : 1;
    const int Dout = v.chi[out_bond];
    const int Z = Dp * d;
    const double* __restrict__ M = v.E;
    const int64_t sz = v.cap;
    const int ZP = (Z + 3) & ~3;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, kq = lane >> 4;
    const double* ph = v.phi + (int64_t)site * v.N * d;
    const int nt_out = (Dout + 15) >> 4;                    // <= 4 on this path (chi <= 64)
    const int slot = wave / (4 / tp), nt = wave % (4 / tp);
    const int col = nt * 16 + i16;
    const bool cv = nt < nt_out && col < Dout;
    const int nsteps = ZP >> 2, ks4 = env_ks4(nsteps);          // four chains over the quarters of the contraction (mpst_internal.h)
    double bv[4][8];                                            // quarter-major: d*chi <= 128 here, at most 8 k-steps per quarter
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int z = 4 * (q * ks4 + u) + kq;
            bv[q][u] = (cv && u < ks4 && z < Z) ? M[(int64_t)z * sz + col] : 0.0;
        }
    for (int t0 = bid * tp; t0 < v.ntiles; t0 += ntb * tp) {
        if (t0 != bid * tp) __syncthreads();          // the previous pass has been consumed
        for (int sl = 0; sl < tp; ++sl) {
            if (t0 + sl < v.ntiles) {
                const Span tl = v.tiles[t0 + sl];
                stage16(smem + sl * 16 * FXS, tl.start, tl.count, prev, Dp, ph, d, v.cap, left_side != 0, tid);
            }
        }
        __syncthreads();
        if (t0 + slot < v.ntiles && nt < nt_out) {
            const Span tl = v.tiles[t0 + slot];
            const double* Xs = smem + slot * 16 * FXS;
            d4 p[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) p[q] = d4{0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int step = q * ks4 + u;
                    // (unconditional: the B operand of a step beyond the contraction is zero and its A operand is read from the last live
                    // step, so the chain gains an exact +0 - a predicated MFMA costs a copy of the accumulator, and the copies of 32 of
                    // them took the kernel past 256 registers: one workgroup per CU instead of two, 2.7x the time with 32 fits per launch)
                    p[q] = mfma_f64(Xs[i16 * FXS + 4 * min(step, nsteps - 1) + kq], bv[q][u], p[q]);
                }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = kq + 4 * r;
                if (i < tl.count && cv) out[(int64_t)(tl.start + i) * v.cap + col] = (p[0][r] + p[1][r]) + (p[2][r] + p[3][r]);
            }
        }
    }
}

// =====================================================================================================================
// Sliced bond GEMMs (round 3): k_yhat_s + k_grad_s replace k_bond_fused + k_fused_reduce.
//
// k_bond_fused keeps a whole B_c (128 KB) per workgroup and writes a whole partial gradient (128 KB) per workgroup: at
// N = 4096 both transfers cost more than the matrix work between them, and the partials (16.8 MB written, then re-read by
// k_fused_reduce) are 14x the algorithmic bytes of the bond.  The pair below moves tens of KB per workgroup:
//   k_yhat_s   yhat_i = X_i^T B_c Y_i split over 16-column slices of B_c: one workgroup = 128 series (one 16-series tile
//              per wave) x one slice, so the B_c operand is 16 KB per workgroup and is shared by its 8 waves; X_i is formed
//              on the fly from its factors (environment row, site vector) staged in LDS.  The slices' contributions go to
//              ypart[series][slice] and are added in slice order by the reader.
//   k_grad_s   output-stationary gradient: one workgroup = one (class, 16 x 16 block of G_c, share of the class' series).
//              A block is (a range of the left bond) x (all left site states) by (all right site states) x (a range of
//              the right bond), so a series contributes 2 (16/d + d) numbers to it instead of two environment rows.  The
//              8 waves split the series and meet in LDS.  With ONE share per block (the headline shape) the block is final:
//              no partial gradients exist at all.  With more shares (large N) they meet through memory: write-through
//              partials (2 KB each), a ticket per block, and the LAST arriver adds them in share order - a fixed order, so
//              the result does not depend on who was last.  loss_functions.jl:248-262,353-369 (KLD), :489-531,600-612 (MSE).
// =====================================================================================================================
constexpr int YS_W = 16;         // columns of B_c per slice
constexpr int YS_MAXSL = 8;      // slices at the capacity of this path (d*chi <= 128)
struct B2 {
    int Dl, Dr, X, Y;
    int64_t L;
    int aw, bw;         // bond entries per x / y block: aw*d <= 32 rows, d*bw <= 32 columns
    int nbx, nby;       // live blocks
};
__device__ __forceinline__ B2 b2_dims(const View& v, int lid) {
    B2 b;
    b.Dl = v.chi[lid];
    b.Dr = v.chi[lid + 2];
    b.X = b.Dl * v.d;
    b.Y = v.d * b.Dr;
    b.L = (int64_t)b.X * b.Y;
    b.aw = max(1, 32 / v.d);
    b.bw = b.aw;
    b.nbx = (b.Dl + b.aw - 1) / b.aw;
    b.nby = (b.Dr + b.bw - 1) / b.bw;
    return b;
}
__device__ __forceinline__ void st_agent(double* p, double x) { __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_agent(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

constexpr int YS_T = 512;        // 8 waves, one 16-series tile each

// class-pure tile t of the data set from the tables in the kernel arguments (no dependent global load)
__device__ __forceinline__ Span tile_span_k(const View& v, int t) {
    int c = 0;
#pragma unroll 1
    for (int k = 1; k < v.C; ++k) c += (t >= v.kcls_tile[k]) ? 1 : 0;
    Span s;
    s.cls = c;
    s.start = v.kcls_off[c] + TILE_S * (t - v.kcls_tile[c]);
    s.count = t < v.ntiles ? min(TILE_S, v.kcls_off[c + 1] - s.start) : 0;
    s.pad = 0;
    return s;
}

// grid.x = nslc * ngw (see launch_yhat_s): workgroup id -> (group walker = id % ngw, slice = id / ngw); grid.y = passes (MSE: C)
// Every wave stages, consumes and stores its own tile: no workgroup barrier, one round trip to memory per group.
// LM: 16-entry pieces of an environment row a lane group fetches (capacity <= 16 LM); D4: d == 4 (the headline shapes).
// V2 (d == 4, even capacity <= 32): the rows of a tile are fetched with 16-byte loads - 10 load instructions per lane and group instead
// of 20.  Measured (profiles/r06_batched_gemm_ab.txt): of the 6.3 us a group takes, 2.7 are spent ISSUING the next group's 20 loads (about
// 40 cycles of the CU's address path per wave-wide 8-byte load, 160 of them per group and CU), 0.25 writing the rows to LDS, 2.9 on the
// 32 MFMAs, the row dots and the stores.  The right-hand factors (RE row, phi_r) then go through LDS like the left-hand ones: the
// 16-byte loads fetch whole rows, the lane that needs Y_i[col] = phi_r[i][sp] RE_i[bb] picks its two numbers there.
template <int LM, bool D4, bool V2>
__device__ __forceinline__ void yhat_s_body(const View& v, int lid, int nslc, int ngw, const int bid_x, const int bid_y) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const B2 b = b2_dims(v, lid);
    const int d = v.d, rid = lid + 1;
    const bool mse = v.loss == MPST_LOSS_MSE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, kq = lane >> 4;
    const int gw = bid_x % ngw, sl = bid_x / ngw;       // consecutive ids (XCDs) walk different series groups
    if (sl * YS_W >= b.Y) return;
    const int pass = mse ? bid_y : 0;
    const int col = sl * YS_W + i16;
    const bool cv = col < b.Y;
    const int sp = cv ? col / b.Dr : 0, bb = cv ? col - sp * b.Dr : 0;       // Y_i[col] = phi_r[i][sp] * RE_i[bb]
    const double* LEp = lid > 0 ? v.LE + (int64_t)(lid - 1) * v.N * v.cap : nullptr;
    const double* REn = rid < v.T - 1 ? v.RE + (int64_t)(rid + 1) * v.N * v.cap : nullptr;
    const double* phl = v.phi + (int64_t)lid * v.N * d;
    const double* phr = v.phi + (int64_t)rid * v.N * d;
    const int ngroups = (v.ntiles + 7) >> 3;
    const int ls = v.cap + 1, ps = d + 1;                  // LDS row strides (odd: the 16 rows a wave reads hit 16 banks)
    const int wstride = (V2 ? 2 : 1) * 16 * (ls + ps);     // V2: [LE rows | phi_l | RE rows | phi_r] per wave
    double* LEs = smem + wave * wstride;                   // [16][ls] this wave's environment rows, zero beyond Dl
    double* PHs = LEs + 16 * ls;                           // [16][ps] site vectors
    double* REs = PHs + 16 * ps;                           // V2: [16][ls] right environment rows, zero beyond Dr
    double* PRs = REs + 16 * ls;                           // V2: [16][ps]
    const int XP = (b.X + 3) & ~3;
    const unsigned magic = (65536u + (unsigned)d - 1u) / (unsigned)d;        // x / d = (x * magic) >> 16 for x < 1024
    double* ypart = v.ypart + (int64_t)pass * v.N * YS_MAXSL;
#ifdef MPST_B2_DEBUG
    unsigned long long* dbg = (v.dbg && tid == 0) ? v.dbg + (4096 + (int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 : nullptr;
    int dbi = 0;
#define YSTAMP() do { if (dbg && dbi < 8) dbg[dbi++] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define YSTAMP() do { } while (0)
#endif
    YSTAMP();
    d4 bq[8];
    int cur_cls = -1;
#ifdef MPST_B2_DEBUG
    if (b.Y > 0) YSTAMP();              // bond dimensions have arrived
#endif
    // The slice of B_c the first group needs goes through LDS once per workgroup (its 8 waves would otherwise each pull the
    // same 16 KB through the CU's L1); a wave whose tile belongs to another class (class boundaries, later groups of a
    // persistent workgroup) reads its fragment from global memory instead.  The first group's own rows are requested in the
    // same breath: one round trip to memory before the first MFMA, not two.
    const int cls0 = mse ? pass : tile_span_k(v, 8 * gw).cls;
    double* Bsh = smem + 8 * wstride;                      // [128][17]
    double t4[4];
    {
        const double* Bc = v.bt + (int64_t)cls0 * b.L;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int idx = tid + YS_T * k, x = idx >> 4, cc = sl * YS_W + (idx & 15);
            t4[k] = (x < b.X && cc < b.Y) ? Bc[(int64_t)x * b.Y + cc] : 0.0;
        }
    }
    // this wave's tile of a group.  Scalar form: 4 rows x 16 consecutive bond entries per load instruction, and the 4 Y values per lane.
    // V2: lane (kq, i16) fetches entries 2 i16, 2 i16 + 1 of rows kq + 4 q of both environments (4 + 4 loads), lanes < 32 the site
    // vectors of row lane / 2 (two entries each: 1 + 1 loads)
    double lev[4][LM], phv[4], yv[4];
    double2 le2[4], re2[4], pl2 = make_double2(0.0, 0.0), pr2 = make_double2(0.0, 0.0);
    Span tl{0, 0, 0, 0};
    auto load_tile = [&](int g) {
        tl = tile_span_k(v, 8 * g + wave);
        if constexpr (V2) {
            const int a = 2 * i16;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = 4 * q + kq;
                const bool valid = row < tl.count;
                const int64_t smp = tl.start + (valid ? row : 0);
                le2[q] = make_double2(0.0, 0.0);
                re2[q] = make_double2(0.0, 0.0);
                if (valid && a < v.cap) {
                    if (LEp) le2[q] = *(const double2*)(LEp + smp * v.cap + a);
                    else le2[q] = make_double2(1.0, 1.0);
                    if (REn) re2[q] = *(const double2*)(REn + smp * v.cap + a);
                    else re2[q] = make_double2(1.0, 1.0);
                }
                if (a >= b.Dl) le2[q].x = 0.0;
                if (a + 1 >= b.Dl) le2[q].y = 0.0;
                if (a >= b.Dr) re2[q].x = 0.0;
                if (a + 1 >= b.Dr) re2[q].y = 0.0;
            }
            const int prow = lane >> 1, half = lane & 1;
            const bool pvalid = lane < 32 && prow < tl.count;
            const int64_t psmp = tl.start + (pvalid ? prow : 0);
            pl2 = pvalid ? *(const double2*)(phl + psmp * 4 + 2 * half) : make_double2(0.0, 0.0);
            pr2 = pvalid ? *(const double2*)(phr + psmp * 4 + 2 * half) : make_double2(0.0, 0.0);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = kq + 4 * r;
                const int64_t smp = tl.start + (i < tl.count ? i : 0);
                yv[r] = (cv && i < tl.count) ? phr[smp * d + sp] * (REn ? REn[smp * v.cap + bb] : 1.0) : 0.0;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = 4 * q + kq;
                const bool valid = row < tl.count;
                const int64_t smp = tl.start + (valid ? row : 0);
#pragma unroll
                for (int m = 0; m < LM; ++m) {
                    const int a = i16 + 16 * m;
                    lev[q][m] = (valid && a < b.Dl) ? (LEp ? LEp[smp * v.cap + a] : 1.0) : 0.0;
                }
                phv[q] = (valid && i16 < d) ? phl[smp * d + i16] : 0.0;
            }
        }
    };
    if (gw < ngroups) load_tile(gw);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int idx = tid + YS_T * k;
        Bsh[(idx >> 4) * 17 + (idx & 15)] = t4[k];
    }
    __syncthreads();
    for (int g = gw; g < ngroups; g += ngw) {
        const Span tc = tl;                                // the tile whose rows are in the registers
#ifdef MPST_B2_DEBUG
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        YSTAMP();
        double yc[4];
        if (tc.count > 0) {
            if constexpr (V2) {
                const int a = 2 * i16;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = 4 * q + kq;
                    if (a < v.cap) {
                        LEs[row * ls + a] = le2[q].x;
                        LEs[row * ls + a + 1] = le2[q].y;
                        REs[row * ls + a] = re2[q].x;
                        REs[row * ls + a + 1] = re2[q].y;
                    }
                    if (i16 == 0) LEs[row * ls + v.cap] = 0.0;      // the entry a = Dl = capacity the padded K extent can touch
                }
                if (lane < 32) {
                    const int prow = lane >> 1, half = lane & 1;
                    PHs[prow * ps + 2 * half] = pl2.x;
                    PHs[prow * ps + 2 * half + 1] = pl2.y;
                    PRs[prow * ps + 2 * half] = pr2.x;
                    PRs[prow * ps + 2 * half + 1] = pr2.y;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {                       // (same wave wrote them: program order, no barrier)
                    const int i = kq + 4 * r;
                    yc[r] = cv ? PRs[i * ps + sp] * REs[i * ls + bb] : 0.0;
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = 4 * q + kq;
#pragma unroll
                    for (int m = 0; m < LM; ++m) {
                        const int a = i16 + 16 * m;
                        if (a < v.cap) LEs[row * ls + a] = lev[q][m];
                    }
                    if (i16 == 0) LEs[row * ls + v.cap] = 0.0;      // the entry a = Dl = capacity the padded K extent can touch
                    if (i16 < d) PHs[row * ps + i16] = phv[q];
                }
            }
        }
        if constexpr (!V2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) yc[r] = yv[r];
        }
        if (g + ngw < ngroups) load_tile(g + ngw);         // the next group's rows fly during this group's matrix work
        if (tc.count <= 0) continue;                       // (wave-uniform; nothing below synchronises across waves)
        const int cls = mse ? pass : tc.cls;
        if (cls != cur_cls) {                              // B_c fragment: bq[mt][r] = B_c[16 mt + kq + 4 r][col]
            if (cls == cls0) {
#pragma unroll
                for (int mt = 0; mt < 8; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) bq[mt][r] = Bsh[(16 * mt + kq + 4 * r) * 17 + i16];
            } else {
                const double* Bc = v.bt + (int64_t)cls * b.L;
#pragma unroll
                for (int mt = 0; mt < 8; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int x = 16 * mt + kq + 4 * r;
                        bq[mt][r] = (cv && x < b.X) ? Bc[(int64_t)x * b.Y + col] : 0.0;
                    }
            }
            cur_cls = cls;
        }
        YSTAMP();
        {
            const double* ler = LEs + i16 * ls;
            const double* phx = PHs + i16 * ps;
            d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
            if (D4) {
                // x = 4 u + kq = a d + s with a = u, s = kq: the site factor is one register, the row entries sit at
                // immediate offsets
                const double ph = phx[kq];
                if (v.cap == 32) {
                    // every row has its 32 entries (+ the zero behind them) and is zero beyond the live bond: all 32 k-steps, no predicate
                    // (a predicate per MFMA is a branch that keeps the compiler from batching the operand reads)
#pragma unroll
                    for (int u = 0; u < 32; u += 2) {
                        acc0 = mfma_f64(ler[u] * ph, bq[u >> 2][u & 3], acc0);
                        acc1 = mfma_f64(ler[u + 1] * ph, bq[(u + 1) >> 2][(u + 1) & 3], acc1);
                        if ((u & 7) == 6) asm volatile("" ::: "memory");  // at most 8 operands ahead of the matrix pipe
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < 32; u += 2) {
                        if (4 * u < XP) acc0 = mfma_f64(ler[u] * ph, bq[u >> 2][u & 3], acc0);
                        if (4 * u + 4 < XP) acc1 = mfma_f64(ler[u + 1] * ph, bq[(u + 1) >> 2][(u + 1) & 3], acc1);
                        if ((u & 7) == 6) asm volatile("" ::: "memory");  // at most 8 operands ahead of the matrix pipe
                    }
                }
            } else {
#pragma unroll
                for (int u = 0; u < 32; u += 2) {
                    if (4 * u < XP) {
                        const unsigned x = 4u * u + kq, a = (x * magic) >> 16, s = x - a * d;
                        acc0 = mfma_f64(ler[a] * phx[s], bq[u >> 2][u & 3], acc0);
                    }
                    if (4 * u + 4 < XP) {
                        const unsigned x = 4u * u + 4u + kq, a = (x * magic) >> 16, s = x - a * d;
                        acc1 = mfma_f64(ler[a] * phx[s], bq[(u + 1) >> 2][(u + 1) & 3], acc1);
                    }
                    if ((u & 3) == 2) asm volatile("" ::: "memory");      // at most 4 operand pairs ahead of the matrix pipe
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double x = sum16((acc0[r] + acc1[r]) * yc[r]);
                const int i = kq + 4 * r;
                if (i16 == 0 && i < tc.count) ypart[(int64_t)(tc.start + i) * YS_MAXSL + sl] = x;
            }
        }
        YSTAMP();
    }
}

constexpr int GS_T = 512;        // 8 waves: wave = (MFMA tile 0..3 of the 32 x 32 block, half of a stage's series)
constexpr int GS_KC = 256;       // series per stage: one round trip to memory each
constexpr int GS_MAXKS = 64;

// grid.x = ksplit * nbc * nbc (capacity), grid.y = C.  Workgroup id -> (ks = id % ksplit, block = id / ksplit).
// AW2: compile-time bound of ceil(aw / 8), D2 of ceil(d / 8) (register arrays of the loader role).
template <int AW2, int D2, int FS, int KC, int NW>
__device__ __forceinline__ void grad_s_body(const View& v, int lid, int ksplit, int nbc, const int bid_x, const int bid_y) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double redl[8];
    __shared__ int last_s;
    const B2 b = b2_dims(v, lid);
    const int d = v.d, rid = lid + 1;
    const bool mse = v.loss == MPST_LOSS_MSE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, kq = lane >> 4;
    const int c = bid_y;
    const int ks = bid_x % ksplit, blk = bid_x / ksplit;
    const int bx = blk / nbc, by = blk % nbc;
    const int grp = c * nbc * nbc + blk;                    // ticket / norm-piece slot (capacity layout)
    if (bx >= b.nbx || by >= b.nby) {
        if (ks == 0 && tid == 0) v.norm_part[grp] = 0.0;    // a block that is not live at this bond contributes nothing
        return;
    }
    const int nsl = (b.Y + YS_W - 1) / YS_W;
    // the series of this pass and this workgroup's share of them
    const int p0 = mse ? 0 : v.kcls_off[c], p1 = mse ? (int)v.N : v.kcls_off[c + 1];
    int len = (p1 - p0 + ksplit - 1) / ksplit;
    len = (len + 3) & ~3;
    const int s0 = min(p1, p0 + ks * len), s1 = min(p1, s0 + len);
    const int a0 = bx * b.aw, b0 = by * b.bw;
    const double* LEp = lid > 0 ? v.LE + (int64_t)(lid - 1) * v.N * v.cap : nullptr;
    const double* REn = rid < v.T - 1 ? v.RE + (int64_t)(rid + 1) * v.N * v.cap : nullptr;
    const double* phl = v.phi + (int64_t)lid * v.N * d;
    const double* phr = v.phi + (int64_t)rid * v.N * d;
    const double2* ypart2 = (const double2*)(v.ypart + (int64_t)(mse ? c : 0) * v.N * YS_MAXSL);
    // record of a series in LDS: [LE slice (aw) | phi_l (d) | RE slice (bw) | phi_r (d) | 0], stride fs (odd); then the weights
    const int o_pl = b.aw, o_re = b.aw + d, o_pr = b.aw + d + b.bw, zc = 2 * b.aw + 2 * d;
    const int fs = FS > 0 ? FS : ((zc + 1) | 1);            // FS: the stride at compile time (d = 4: 25) - every LDS address of the matrix loop an immediate
    double* wv = smem + KC * fs;                         // [KC] w_i of the stage's series (0 beyond the share)
    // loader role 1: 8 lanes per series (entries j, j + 8 of every factor), 4 passes of 64 series per stage
    const int lsm = tid >> 3, j = tid & 7;
    constexpr int NP = KC / (8 * NW);    // passes of 8 NW series per stage (8 lanes per series)
    double r_le[NP][AW2], r_re[NP][AW2], r_pl[NP][D2], r_pr[NP][D2];
    // loader role 2 (threads < KC): the slices' contributions to yhat of series base + tid, 64 contiguous bytes
    double2 r_y[4];
    double r_dl = 0.0;
    bool r_ok = false;
    auto load_stage = [&](int base) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int smp = base + 8 * NW * p + lsm;
            const bool ok = smp < s1;
            const int64_t sm = ok ? smp : s0;
#pragma unroll
            for (int h = 0; h < AW2; ++h) {
                const int jj = j + 8 * h;
                r_le[p][h] = (ok && jj < b.aw && a0 + jj < b.Dl) ? (LEp ? LEp[sm * v.cap + a0 + jj] : 1.0) : 0.0;
                r_re[p][h] = (ok && jj < b.bw && b0 + jj < b.Dr) ? (REn ? REn[sm * v.cap + b0 + jj] : 1.0) : 0.0;
            }
#pragma unroll
            for (int h = 0; h < D2; ++h) {
                const int jj = j + 8 * h;
                r_pl[p][h] = (ok && jj < d) ? phl[sm * d + jj] : 0.0;
                r_pr[p][h] = (ok && jj < d) ? phr[sm * d + jj] : 0.0;
            }
        }
        if (tid < KC) {
            const int smp = base + tid;
            r_ok = smp < s1;
            const int64_t sm = r_ok ? smp : s0;
#pragma unroll
            for (int q = 0; q < 4; ++q) r_y[q] = ypart2[sm * (YS_MAXSL / 2) + q];
            r_dl = (mse && r_ok && v.label[sm] == c) ? 1.0 : 0.0;
        }
    };
    // consumer role (round 6): wave w takes the series [32 w, 32 w + 32) of a stage - an eighth of the contraction - and ALL four 16 x 16
    // tiles of the block: two row operands and two column operands per k-step feed four MFMAs (9 LDS reads and 6 products per four
    // MFMAs; one tile per wave cost 5 reads and 3 products per MFMA and the LDS pipe, not the matrix pipe, set the pace).  The eight
    // partial blocks meet in LDS after the last stage, in wave order.
    const int tile = wave & 3, kh = wave >> 2;              // epilogue: waves 0..3 own tile (tx, ty) of the finished block
    const int tx = tile >> 1, ty = tile & 1;
    int ia[2], ipl[2], ib[2], ipr[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int xr = 16 * h + i16, yc = 16 * h + i16;         // block-local row / column this lane feeds
        const int al = xr / d, slx = xr - al * d;               // X_i[row] = LE_i[a0 + al] * phi_l[i][slx]
        const int spl = yc / b.bw, bl = yc - spl * b.bw;        // Y_i[col] = phi_r[i][spl] * RE_i[b0 + bl]
        const bool xv = al < b.aw && a0 + al < b.Dl, yvld = spl < d && b0 + bl < b.Dr;
        // operands of lanes outside the live block come from the record's zero cell: no select in the loop
        ia[h] = xv ? al : zc;
        ipl[h] = xv ? o_pl + slx : zc;
        ib[h] = yvld ? o_re + bl : zc;
        ipr[h] = yvld ? o_pr + spl : zc;
    }
    d4 acc[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) acc[h][0] = acc[h][1] = d4{0.0, 0.0, 0.0, 0.0};
    double loss = 0.0;
    const bool do_loss = bx == 0 && by == 0;
#ifdef MPST_B2_DEBUG
    unsigned long long* dbg = (v.dbg && tid == 0) ? v.dbg + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 : nullptr;
#define GSTAMP(i) do { if (dbg) dbg[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define GWAITSTAMP(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); GSTAMP(i); } while (0)
#else
#define GSTAMP(i) do { } while (0)
#define GWAITSTAMP(i) do { } while (0)
#endif
    GSTAMP(0);
    if (s0 < s1) load_stage(s0);
    GWAITSTAMP(6);
#ifdef MPST_B2_DEBUG
    unsigned long long g_wait = 0, g_stage = 0, g_n = 0;
#endif
    for (int base = s0; base < s1; base += KC) {
#ifdef MPST_B2_DEBUG
        const unsigned long long g_t0 = __builtin_amdgcn_s_memrealtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        g_wait += __builtin_amdgcn_s_memrealtime() - g_t0;
#endif
        __syncthreads();                                   // the previous stage is consumed
        if (tid < KC) {
            double w = 0.0;
            if (r_ok) {
                const double ys[8] = {r_y[0].x, r_y[0].y, r_y[1].x, r_y[1].y, r_y[2].x, r_y[2].y, r_y[3].x, r_y[3].y};
                double yh = 0.0;
#pragma unroll
                for (int q = 0; q < YS_MAXSL; ++q)
                    if (q < nsl) yh += ys[q];                   // slice order
                w = mse ? (yh - r_dl) : 1.0 / yh;                                              // :489,:608 / :258,:367
                if (do_loss) loss += mse ? 0.5 * (yh - r_dl) * (yh - r_dl) : -log(yh * yh);      // :554 / :318
            }
            wv[tid] = w;
        }
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            double* rec = smem + (8 * NW * p + lsm) * fs;
#pragma unroll
            for (int h = 0; h < AW2; ++h) {
                const int jj = j + 8 * h;
                if (jj < b.aw) {
                    rec[jj] = r_le[p][h];
                    rec[o_re + jj] = r_re[p][h];
                }
            }
#pragma unroll
            for (int h = 0; h < D2; ++h) {
                const int jj = j + 8 * h;
                if (jj < d) {
                    rec[o_pl + jj] = r_pl[p][h];
                    rec[o_pr + jj] = r_pr[p][h];
                }
            }
            if (j == 0) rec[zc] = 0.0;
        }
        __syncthreads();
#ifdef MPST_B2_DEBUG
        g_stage += __builtin_amdgcn_s_memrealtime() - g_t0;
        g_n += 1;
        if (dbg) { dbg[7] = g_wait; dbg[5] = g_stage | (g_n << 48); }
#endif
        if (base + KC < s1) load_stage(base + KC);   // the next stage's loads fly during the matrix work
        const double* rbase = smem + (wave * (KC / NW) + kq) * fs;
        const double* wb = wv + wave * (KC / NW) + kq;
#pragma unroll
        for (int u = 0; u < KC / (4 * NW); ++u) {         // series beyond the share are zero records with zero weight
            const double* rec = rbase + 4 * u * fs;
            const double w = wb[4 * u];
            const double x0 = rec[ia[0]] * rec[ipl[0]], x1 = rec[ia[1]] * rec[ipl[1]];
            const double y0 = rec[ib[0]] * rec[ipr[0]] * w, y1 = rec[ib[1]] * rec[ipr[1]] * w;
            acc[0][0] = mfma_f64(x0, y0, acc[0][0]);
            acc[0][1] = mfma_f64(x0, y1, acc[0][1]);
            acc[1][0] = mfma_f64(x1, y0, acc[1][0]);
            acc[1][1] = mfma_f64(x1, y1, acc[1][1]);
            if (u & 1) asm volatile("" ::: "memory");       // operands of at most two k-steps ahead of the matrix pipe
        }
    }
    GSTAMP(1);
    // the eight partial blocks meet in LDS (64 KB, over the stage records): tile t of wave w at [(w * 4 + t) * 256]
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) smem[(wave * 4 + 2 * h + g) * 256 + r * 64 + lane] = acc[h][g][r];
    if (do_loss) {                                          // threads < KC carry the terms
        loss = wave_sum(loss);
        if (lane == 0) redl[wave] = loss;
    }
    __syncthreads();
    const double scale = mse ? v.invN : -(v.train_sep ? v.inv_count[c] : v.invN);           // :608 / :367,:424
    // waves 0..3 (kh == 0) own the block's entries: lane holds rows kq + 4 r of tile (tx, ty), column i16
    double tot[4] = {0.0, 0.0, 0.0, 0.0};
    int64_t gidx[4];
    bool ev[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 16 * tx + kq + 4 * r, colb = 16 * ty + i16;
        const int aa = row / d, sx = row - aa * d, sy = colb / b.bw, bb2 = colb - sy * b.bw;
        ev[r] = kh == 0 && aa < b.aw && a0 + aa < b.Dl && sy < d && b0 + bb2 < b.Dr;
        gidx[r] = 2 + (int64_t)c * b.L + (int64_t)((a0 + aa) * d + sx) * b.Y + sy * b.Dr + b0 + bb2;
        if (kh == 0) {                                      // wave order, pairwise: ((0 + 1) + (2 + 3)) + ((4 + 5) + (6 + 7))
            const double* q = smem + tile * 256 + r * 64 + lane;
            tot[r] = (q[0] + q[1024]) + (q[2048] + q[3072]);
            if (NW == 8) tot[r] += (q[4096] + q[5120]) + (q[6144] + q[7168]);
        }
    }
    if (do_loss && tid == 0) v.lossp[c * ksplit + ks] = (redl[0] + redl[1]) + (redl[2] + redl[3]);    // waves 4..7 carry none
    double n2 = 0.0;
    if (ksplit == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (ev[r]) {
                const double gval = tot[r] * scale;
                v.gradbuf[gidx[r]] = gval;
                n2 = fma(gval, gval, n2);
            }
    } else {
        double* part = v.partial + ((int64_t)grp * ksplit + ks) * 1024 + tile * 256;
        if (kh == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) st_agent(part + r * 64 + lane, tot[r]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the write-through stores have left before the ticket is taken
        __syncthreads();
        GSTAMP(2);
        if (tid == 0) {
            const unsigned t = __hip_atomic_fetch_add(v.tick + grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last_s = (t == (unsigned)ksplit - 1) ? 1 : 0;
        }
        __syncthreads();
        GSTAMP(3);
        if (!last_s) return;
        // last arriver: the shares in share order
        const double* pbase = v.partial + (int64_t)grp * ksplit * 1024 + tile * 256;
        if (kh == 0) {
            double s[4] = {0.0, 0.0, 0.0, 0.0};
            for (int k0 = 0; k0 < ksplit; k0 += 8) {           // 32 loads in flight: one round trip for up to 8 shares
                double t[8][4];
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        t[u][r] = (k0 + u < ksplit) ? ld_agent(pbase + (int64_t)(k0 + u) * 1024 + r * 64 + lane) : 0.0;
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[r] += t[u][r];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (ev[r]) {
                    const double gval = s[r] * scale;
                    v.gradbuf[gidx[r]] = gval;
                    n2 = fma(gval, gval, n2);
                }
        }
        if (tid == 0) __hip_atomic_store(v.tick + grp, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    n2 = wave_sum(n2);
    __syncthreads();
    if (lane == 0) redl[wave] = n2;
    __syncthreads();
    if (tid == 0) v.norm_part[grp] = (redl[0] + redl[1]) + (redl[2] + redl[3]);      // waves 4..7 carry no entries
    GSTAMP(4);
}

// =====================================================================================================================
// k_bond_tail (round 6): everything of a bond that follows the eigenvectors, in ONE launch - the four-launch chain
//   k_grad_s -> k_gram_upd -> k_eig_trivec -> k_bond_tail.
// What used to be three dependent launches (k_eig_fin, k_env_split, the next bond's k_yhat_s: 33 us, of which about 12 are
// three cold starts and first round trips to memory) is one:
//  * every workgroup repeats k_eig_fin's work for itself - truncation rule (NDTensors truncate!), verification of the candidate
//    vectors, Loewdin polish (verify_and_polish, the same code: the same bits in every workgroup) - and keeps the kept
//    eigenvectors E in ITS LDS: 512 MFMAs of redundant work per CU (1.7 us) against a launch boundary and a trip through memory;
//  * tile workgroups (16 series each): the new environment rows env' = S E (update_caches!, the very sums of k_env_split / k_env:
//    a sweep with the reference's cache rebuilds stays bit-identical), and the NEXT bond's overlaps
//        yhat_i = <B'_c, X'_i (x) Y'_i>,  B' = W[neighbour] * T_c,  T_c = bt_new_c E / ||bt_new||   (flatten_bt, RealRealHighDimension.jl:221-238)
//               = O_i^T bt_new_c (E (E^T S_i)) / ||bt_new||
//    with S_i / O_i the Khatri-Rao vectors of this bond on the side the eigenvectors live on / the other side (the environment of
//    the other side is the next bond's outer environment contracted with the neighbouring site: it exists in the cache).  The
//    product P = O bt_new (256 MFMAs per tile) needs neither E nor the new environment and runs before the polish; a wave owns 16
//    columns of it and forms z = E env' for the same 16 columns in the accumulator layout, so the row dot needs no exchange;
//  * split / chain workgroups: the back-split and the next bond's tensor as in k_env_split, with E from LDS.
// A failed verification (genuinely clustered kept eigenvalues: k_eig_fin would fall through to its Jacobi solver) sets the sticky
// DevScalars::redo (= 1 + the bond's position in the sweep) and leaves the MPS, the caches and the chained tensor alone; every later tail
// launch of the sweep leaves at once, and the host finishes the sweep FROM THAT BOND on the six-launch chain (mpst_sweep; no snapshot).  loss_functions.jl:248-262 (yhat), RealRealHighDimension.jl:107-203.
// =====================================================================================================================
constexpr int BT_T = 512;                    // 8 waves
constexpr int BT_ZS = 36;                    // LDS row stride of the candidate / kept eigenvectors: rows 16 apart in 16 different bank pairs
                                             // (with the 32 of k_eig_fin's layout the A operand of Z D is a 16-way conflict: 3 us per polish)
constexpr int BT_ENVS = 34;                  // LDS row stride of the 16 x 32 tile of new environment rows
constexpr int BT_ELS = 37, BT_PLS = 21;      // LDS row strides of the staged factors (odd: the 16 rows a wave reads hit 16 bank pairs): 32 bond
                                             // entries / 16 site states and zeros behind them (the padded K extent reads up to 4 beyond the live ones)
constexpr int BT_FAC = 16 * (BT_ELS + BT_PLS);
constexpr int BT_SS = 129;                   // row stride of the dense S tile (over D / Dh / misc once the polish is done)
constexpr int BT_LDS_DOUBLES = 128 * BT_ZS + 32 * 32 + 1024 + 128 + 2048 + 16 * BT_ENVS + 128;     // 76 KB
static_assert(2 * BT_FAC <= 2048, "the factors live where the pieces of env' go later");
static_assert(16 * BT_SS <= 32 * 32 + 1024 + 128, "the dense S tile lives in the polish scratch");

// One side's Khatri-Rao vectors of a 16-series tile, kept as their FACTORS in LDS (environment row, site vector); entry z of row i
// is formed where it is wanted: left  z = a d + s: prev_i[a] phi_i[s];  right  z = s Dp + b: phi_i[s] prev_i[b]
// (the one product stage16 forms: the same bits).  Division by d / Dp with a multiply (z < 1024).
struct KrSide {
    const double* env;      // [16][BT_ELS]
    const double* ph;       // [16][BT_PLS]
    unsigned magic, div;
    bool left;
};
__device__ __forceinline__ KrSide kr_side(const double* fac, int Dp, int d, bool left) {
    KrSide k;
    k.env = fac;
    k.ph = fac + 16 * BT_ELS;
    k.div = (unsigned)(left ? d : Dp);
    k.magic = (65536u + k.div - 1u) / k.div;
    k.left = left;
    return k;
}
__device__ __forceinline__ double kr_at(const KrSide& k, int row, unsigned z) {
    const unsigned q = (z * k.magic) >> 16, r = z - q * k.div;
    // (entries the padded extent of a product asks for beyond the live ones: on the right the site index reaches 21 at d = 11 .. 16 with a
    // bond of 3 and 31 with a bond of 1 - tests/fuzz_chain4.py found the first; it is clamped onto the row's zero pad, as is the bond index on the left)
    const unsigned ie = min(k.left ? q : r, (unsigned)(BT_ELS - 1)), ip = min(k.left ? r : q, (unsigned)(BT_PLS - 1));
    return k.env[row * BT_ELS + ie] * k.ph[row * BT_PLS + ip];
}

// =====================================================================================================================
// k_env_walk (round 6): construct_caches (RealRealHighDimension.jl:45-103) in ONE launch.  The environments of different series never
// meet: a workgroup takes a tile of 16 series and walks the whole chain with it - T - 1 dependent steps, no grid-wide dependency -
// where enqueue_caches() issues one k_env launch per site (99 launches of ~7 us at the headline shape, twice per sweep when the
// reference's cache rebuilds are on: 1.4 ms of a 40 ms sweep).  A step is the product of k_env, bit for bit (the Khatri-Rao entries
// prev_i[a] * phi_i[s], four MFMA chains over the quarters of the contraction added as (q0 + q1) + (q2 + q3): mpst_internal.h), so the
// caches it leaves are those of the per-site launches and of the incremental update_caches! of a sweep.  8 waves = (column tile,
// quarter); the row just computed stays in LDS as the next step's `prev`; the next site's vectors and this wave's share of the next
// site tensor are requested while the current step computes.  Bond dimensions up to 32, d * chi <= 128.
// =====================================================================================================================
constexpr int EW_T = 512;
constexpr int EW_ZS = 130;                   // row stride of the dense Khatri-Rao tile
constexpr int EW_PS = 34;                    // row stride of the kept environment rows
constexpr int EW_TMAX = 1023;                // longest chain (the bond dimensions are staged in LDS)
__global__ __launch_bounds__(EW_T) void k_env_walk(View v, int left_side, int nstep) {
    __shared__ __attribute__((aligned(16))) double Zt[16 * EW_ZS];
    __shared__ __attribute__((aligned(16))) double prevs[16 * EW_PS];
    __shared__ __attribute__((aligned(16))) double part[8 * 256];
    __shared__ int chis[EW_TMAX + 1];             // the bond dimensions: a scalar load from memory per step would be a dependent microsecond each
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, kq = lane >> 4;
    const int d = v.d, T = v.T;
    for (int i = tid; i <= T; i += EW_T) chis[i] = v.chi[i];
    const Span tl = tile_span_k(v, (int)blockIdx.x);
    __syncthreads();
    const int nt = wave & 1, q = wave >> 1;
    const int col = nt * 16 + i16;
    const int64_t cs = (int64_t)v.N * v.cap;
    double* base = left_side ? v.LE : v.RE;
    // sites in walking order: left 0 .. T-2 (LE[j] from LE[j-1] and site j), right T-1 .. 1 (RE[j] from RE[j+1] and site j); nstep of them
    auto site_of = [&](int st) { return left_side ? st : T - 1 - st; };
    // the sites' vectors come from HBM (2+ us away, more than a step takes): they are fetched a CHUNK of sites ahead - thread = (site of
    // the chunk, row, s) - and parked in LDS, two chunks deep
    const int per = 16 * d;
    const int CH = min(8, EW_T / per);              // sites per chunk (d <= 16: at least 2)
    __shared__ double phs[2][8 * 16 * 17];
    const int cs_site = tid / per, cs_rem = tid - cs_site * per;
    const int prow = cs_rem / d, ps = cs_rem - prow * d;
    auto load_phi_chunk = [&](int c0) -> double {   // this thread's value of the chunk that starts at step c0
        const int st = c0 + cs_site;
        const bool ok = cs_site < CH && st < nstep && prow < tl.count;
        return ok ? v.phi[((int64_t)site_of(st) * v.N + tl.start + prow) * d + ps] : 0.0;
    };
    // this wave's share of a site tensor as the B operand of its quarter: M[z][col], z = 4 (q ks4 + u) + kq
    auto load_b = [&](int st, double (&bv)[8]) {
        const int j = site_of(st);
        const int Dp = left_side ? __builtin_amdgcn_readfirstlane(chis[j]) : __builtin_amdgcn_readfirstlane(chis[j + 1]);
        const int Dout = left_side ? __builtin_amdgcn_readfirstlane(chis[j + 1]) : __builtin_amdgcn_readfirstlane(chis[j]);
        const int Z = Dp * d, nsteps = ((Z + 3) & ~3) >> 2, ks4 = env_ks4(nsteps);
        const double* M = v.sites + (int64_t)j * v.site_stride;
        // left: M[(a,s)][k], row stride chi[j+1];  right: M[k][(s,b)] read transposed, column stride d * chi[j+1]
        const int64_t sz = left_side ? Dout : 1, sk = left_side ? 1 : (int64_t)d * __builtin_amdgcn_readfirstlane(chis[j + 1]);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int z = 4 * (q * ks4 + u) + kq;
            bv[u] = (col < Dout && u < ks4 && z < Z) ? M[(int64_t)z * sz + (int64_t)col * sk] : 0.0;
        }
    };
    double bv[8], bvn[8];                           // this step's and the next step's B operand; the one after is requested inside the step
    double phq = load_phi_chunk(0);                 // chunk 0 ...
    load_b(0, bv);
#pragma unroll
    for (int u = 0; u < 8; ++u) bvn[u] = 0.0;
    if (nstep > 1) load_b(1, bvn);
    if (cs_site < CH) phs[0][cs_site * 272 + prow * 17 + ps] = phq;
    phq = load_phi_chunk(CH);                       // ... and chunk 1 in flight
    for (int st = 0; st < nstep; ++st) {
        const int j = site_of(st);
        const int Dp = st == 0 ? 1 : (left_side ? __builtin_amdgcn_readfirstlane(chis[j]) : __builtin_amdgcn_readfirstlane(chis[j + 1]));
        const int Dout = left_side ? __builtin_amdgcn_readfirstlane(chis[j + 1]) : __builtin_amdgcn_readfirstlane(chis[j]);
        const int Z = Dp * d, ZP = (Z + 3) & ~3, nsteps = ZP >> 2, ks4 = env_ks4(nsteps);
        const int ch = st / CH, cin = st - ch * CH;
        if (cin == 0 && st > 0) {
            // a new chunk begins: the one fetched during the last chunk goes to LDS (the buffer of the chunk before the last is free),
            // the next one is requested
            if (cs_site < CH) phs[ch & 1][cs_site * 272 + prow * 17 + ps] = phq;
            phq = load_phi_chunk((ch + 1) * CH);
        }
        // (LDS-only barriers: __syncthreads() would also wait for the acknowledgement of the last step's stores and for the operands
        // requested two steps ahead - the very round trips the walk is built to hide)
        lds_barrier();                              // (also: the previous step's rows are in prevs, its Zt is consumed)
        // ---- the Khatri-Rao tile of this step: Zt[i][z] = prev_i[a] phi_i[s], left z = a d + s, right z = s Dp + a ----
        {
            const double* ph = phs[ch & 1] + cin * 272;
            const int row = tid >> 5, a = tid & 31;  // 16 rows x 32 bond entries
            const double pa = (a < Dp && row < tl.count) ? (st == 0 ? 1.0 : prevs[row * EW_PS + a]) : 0.0;
            if (a < Dp) {
                for (int s = 0; s < d; ++s) Zt[row * EW_ZS + (left_side ? a * d + s : s * Dp + a)] = pa * ph[row * 17 + s];
            }
            for (int z = Z + a; z < ZP; z += 32) Zt[row * EW_ZS + z] = 0.0;
        }
        lds_barrier();
        double bvnn[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) bvnn[u] = 0.0;
        if (st + 2 < nstep) load_b(st + 2, bvnn);   // two steps ahead: the site tensors come from the L2, about a step away
        d4 acc = {0.0, 0.0, 0.0, 0.0};
        if (nt * 16 < Dout) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int step = q * ks4 + u;
                acc = mfma_f64(Zt[i16 * EW_ZS + 4 * min(step, nsteps - 1) + kq], bv[u], acc);      // (bv is zero beyond the contraction)
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) part[(q * 2 + nt) * 256 + r * 64 + lane] = acc[r];
        lds_barrier();
        {
            const int tnt = tid >> 8, e = tid & 255;
            const double sum = (part[tnt * 256 + e] + part[(2 + tnt) * 256 + e]) + (part[(4 + tnt) * 256 + e] + part[(6 + tnt) * 256 + e]);
            const int i = ((e >> 4) & 3) + 4 * (e >> 6), c = tnt * 16 + (e & 15);
            if (i < tl.count && c < Dout) base[(int64_t)j * cs + (int64_t)(tl.start + i) * v.cap + c] = sum;
            prevs[i * EW_PS + c] = sum;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            bv[u] = bvn[u];
            bvn[u] = bvnn[u];
        }
    }
}
void launch_env_walk(const View& v, int left_side, int nstep, hipStream_t s) {
    if (nstep > 0) hipLaunchKernelGGL(k_env_walk, dim3(v.ntiles), dim3(EW_T), 0, s, v, left_side, std::min(nstep, v.T - 1));
}
bool env_walk_supported(const View& v) { return v.zw != 2 && v.cap <= 32 && v.d * v.cap <= MAX_DIM && v.d >= 2 && v.d <= 16 && v.T >= 2 && v.T <= EW_TMAX; }

// ---- the back-split and the next bond's tensor inside k_bond_tail: split_block / chain_bt_block (the same products, the same order
// of operations: the same bits), done by tile workgroups AFTER their tile work, all 8 waves, a 16-register share of the operands per
// lane requested as soon as the overlap product has released its registers.  A unit of work is small - 8 MFMAs before and 8 after one
// exchange through LDS - so that its hosts finish not long after the plain tile workgroups (as 4-wave jobs with 32 operands per lane
// the hosts ended 6 us after the others; as workgroups of their own, or as extra waves, worse: see k_bond_tail).  Capacity <= 32 on this
// path: the shared bond has at most two 16-row blocks.
//
// Chain job (class c, site state s, 16 kept vectors kt) = workgroup; wave w = (contraction range w & 3, row block j = w >> 2) of
// T = bt_new_c[.., s, ..] E, then output tile w of bt' = W[neighbour] T.  R[0..7]: bt_new operand, R[8..15]: wpre[j'][r] of tile w.
__device__ __forceinline__ void tail_chain_request(const View& v, const BondDimsF& b, int lid, int going_left, int job, double (&R)[16], const int Dnb) {
    const int d = v.d;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i16 = lane & 15, kq = lane >> 4;
    const int ktc = (v.cap + 15) >> 4;
    const int s = (job / ktc) % d, c = job / (ktc * d);
    const double* Bc = v.btn + (int64_t)c * b.L;
    const int Kc = going_left ? b.Y : b.X, Dc = going_left ? b.Dl : b.Dr;
    const int nj = (Dc + 15) >> 4;
    const int ks4 = (((Kc + 3) >> 2) + 3) >> 2;
    {
        const int rg = wave & 3, j = wave >> 2;
        const int kbeg = 4 * ks4 * rg, kend = min(Kc, 4 * ks4 * (rg + 1));
        // (every load unconditional, from a clamped address, the value dropped afterwards: a load under a lane predicate becomes a branch
        // around it whose join waits for the data - sixteen such loads were sixteen L2 round trips one after the other, 1.8 us stamped)
        const int row = 16 * j + i16, rowc = min(row, Dc - 1);
        const double* ap = going_left ? Bc + (int64_t)(rowc * d + s) * b.Y : Bc + s * b.Dr + rowc;
        const int64_t astr = going_left ? 1 : b.Y;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int q = kbeg + 4 * u + kq;
            const double x = ap[(int64_t)min(q, Kc - 1) * astr];
            R[u] = (j < nj && row < Dc && q < kend) ? x : 0.0;
        }
    }
    {
        const double* Wn = v.sites + (int64_t)(going_left ? lid - 1 : lid + 2) * v.site_stride;
        const int Xn = Dnb * d;
        const int tl = 16 * wave + i16, tlc = min(tl, Xn - 1);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int a = 16 * j + 4 * r + kq, ac = min(a, Dc - 1);
                const bool ok = j < nj && tl < Xn && a < Dc;
                const double x = going_left ? Wn[(int64_t)tlc * b.Dl + ac] : Wn[(int64_t)ac * Xn + tlc];
                R[8 + 4 * j + r] = ok ? x : 0.0;
            }
    }
}
__device__ __forceinline__ void tail_chain_job(const View& v, const BondDimsF& b, int lid, int going_left, int job, const double (&R)[16],
                                               double* __restrict__ cpart /* [4 ranges][2 row blocks][256] */, const double* __restrict__ Ev, const int ldE,
                                               const int nk, const double inv, const int Dnb) {
    const int d = v.d;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i16 = lane & 15, kq = lane >> 4;
    const int ktc = (v.cap + 15) >> 4;
    const int kt = job % ktc, s = (job / ktc) % d, c = job / (ktc * d);
    const int k0 = 16 * kt;
    const bool live = k0 < nk;                      // (wave-uniform; the barrier below is reached either way)
    const int kcol = k0 + i16;
    const bool kv = kcol < nk;
    const int Kc = going_left ? b.Y : b.X, Dc = going_left ? b.Dl : b.Dr;
    const int nj = (Dc + 15) >> 4;
    const int ks4 = (((Kc + 3) >> 2) + 3) >> 2;
    {
        const int rg = wave & 3, j = wave >> 2;
        const int kbeg = 4 * ks4 * rg, kend = min(Kc, 4 * ks4 * (rg + 1));
        d4 t = {0, 0, 0, 0};
        if (live && j < nj) {
            double bv[8];                           // the eigenvector operand first, then eight MFMAs back to back
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int q = kbeg + 4 * u + kq;
                bv[u] = (kv && q < kend) ? Ev[(int64_t)q * ldE + kcol] : 0.0;
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int u = 0; u < 8; ++u)
                t = mfma_f64(R[u], bv[u], t);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) cpart[(rg * 2 + j) * 256 + r * 64 + lane] = t[r];
    }
    lds_barrier();
    if (!live) return;
    d4 T[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = j * 256 + r * 64 + lane;
            T[j][r] = ((cpart[o] + cpart[2 * 256 + o]) + (cpart[4 * 256 + o] + cpart[6 * 256 + o])) * inv;
        }
    const int mt = wave;                            // this wave's output tile
    if (going_left) {
        const int Xp = Dnb * d, Yp = d * nk;
        double* out = v.bt + (int64_t)c * Xp * Yp;
        if (16 * mt < Xp) {
            d4 acc = {0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < 2; ++j)
                if (j < nj) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc = mfma_f64(R[8 + 4 * j + r], T[j][r], acc);
                }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int xr = 16 * mt + kq + 4 * r;
                if (xr < Xp && kv) out[(int64_t)xr * Yp + s * nk + kcol] = acc[r];
            }
        }
    } else {
        const int Ypp = d * Dnb, Xp = nk * d;
        double* out = v.bt + (int64_t)c * Xp * Ypp;
        if (16 * mt < Ypp) {
            const int y = 16 * mt + i16;
            d4 acc = {0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < 2; ++j)
                if (j < nj) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc = mfma_f64(T[j][r], R[8 + 4 * j + r], acc);    // A[m = k][kk = b] = T^T tile, register r
                }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = k0 + kq + 4 * r;
                if (k < nk && y < Ypp) out[(int64_t)(k * d + s) * Ypp + y] = acc[r];
            }
        }
    }
}
// back-split: tile = (class, row / column tile of the site that keeps the label, kept-column tile) laid out for the CAPACITY (how many
// vectors are kept is not known when the operands are requested).  A host takes two tiles: wave w = (tile w >> 2, contraction range
// w & 3); the ranges meet in LDS as (p0 + p1) + (p2 + p3).  R[0..7]: the bt_new operand of the wave's range.
__device__ __forceinline__ void tail_split_tile(const View& v, const BondDimsF& b, int going_left, int tile, int& c, int& m0, int& n0) {
    const int tkc = (v.cap + 15) >> 4;
    if (going_left) {
        const int tx = (b.X + 15) >> 4;
        c = tile / (tx * tkc);
        const int rem = tile - c * tx * tkc;
        m0 = (rem / tkc) * 16;      // row tile of T (x)
        n0 = (rem % tkc) * 16;      // kept columns
    } else {
        const int ty = (b.Y + 15) >> 4;
        c = tile / (tkc * ty);
        const int rem = tile - c * tkc * ty;
        m0 = (rem / ty) * 16;       // kept rows
        n0 = (rem % ty) * 16;       // column tile of T (y)
    }
}
__device__ __forceinline__ void tail_split_request(const View& v, const BondDimsF& b, int going_left, int host, double (&R)[16]) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i16 = lane & 15, kq = lane >> 4;
    const int tile = 2 * host + (wave >> 2), rg = wave & 3;
    int c, m0, n0;
    tail_split_tile(v, b, going_left, tile, c, m0, n0);
    const bool live = c < v.C;
    const double* Bc = v.btn + (int64_t)(live ? c : 0) * b.L;
    // going left  T[x][k] = sum_y bt_new[x][y] E[y][k]: A operand bt_new[m = x][kk = y];  going right T[k][y] = sum_x E[x][k] bt_new[x][y]: B operand bt_new[kk = x][n = y]
    const int K = going_left ? b.Y : b.X;
    const int ks4 = (((K + 3) >> 2) + 3) >> 2;
    const int mm = (going_left ? m0 : n0) + i16, mmc = min(mm, (going_left ? b.X : b.Y) - 1);
    const bool mv = live && mm < (going_left ? b.X : b.Y);
    const int kbeg = 4 * ks4 * rg, kend = min(K, 4 * ks4 * (rg + 1));
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int k = kbeg + 4 * u + kq, kc = min(k, K - 1);
        const double x = going_left ? Bc[(int64_t)mmc * b.Y + kc] : Bc[(int64_t)kc * b.Y + mmc];      // (unconditional: see tail_chain_request)
        R[u] = (mv && k < kend) ? x : 0.0;
    }
}
__device__ __forceinline__ void tail_split_job(const View& v, const BondDimsF& b, int lid, int going_left, int host, int nhost, const double (&R)[16],
                                               double* __restrict__ spart /* [8 waves][256] */, const double* __restrict__ Ev, const int ldE, const int nk,
                                               const double inv) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int i16 = lane & 15, kq = lane >> 4;
    double* Wl = v.sites + (int64_t)lid * v.site_stride;
    double* Wr = v.sites + (int64_t)(lid + 1) * v.site_stride;
    const int tkc = (v.cap + 15) >> 4;
    const int K = going_left ? b.Y : b.X;
    const int ks4 = (((K + 3) >> 2) + 3) >> 2;
    const int ntile = v.C * tkc * (((going_left ? b.X : b.Y) + 15) >> 4);
    const int tile = 2 * host + (wave >> 2), rg = wave & 3;
    int c, m0, n0;
    tail_split_tile(v, b, going_left, tile, c, m0, n0);
    const int kn0 = going_left ? n0 : m0;           // first kept vector of the tile
    const bool live = tile < ntile && kn0 < nk;     // (wave-uniform)
    const int kcol = kn0 + i16;
    {
        const int kbeg = 4 * ks4 * rg, kend = min(K, 4 * ks4 * (rg + 1));
        d4 p = {0, 0, 0, 0};
        if (live) {
            double ev[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = kbeg + 4 * u + kq;
                ev[u] = (kcol < nk && k < kend) ? Ev[(int64_t)k * ldE + kcol] : 0.0;
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (kbeg + 4 * u < kend) p = going_left ? mfma_f64(R[u], ev[u], p) : mfma_f64(ev[u], R[u], p);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) spart[wave * 256 + r * 64 + lane] = p[r];
    }
    lds_barrier();
    if (live && rg == 0) {
        const double* sp = spart + (wave & 4) * 256;
        if (going_left) {
            double* out = Wl + (int64_t)c * b.X * nk;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + kq + 4 * r, o = r * 64 + lane;
                if (row < b.X && n0 + i16 < nk) out[(int64_t)row * nk + n0 + i16] = ((sp[o] + sp[256 + o]) + (sp[512 + o] + sp[768 + o])) * inv;
            }
        } else {
            double* out = Wr + (int64_t)c * nk * b.Y;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + kq + 4 * r, o = r * 64 + lane;
                if (row < nk && n0 + i16 < b.Y) out[(int64_t)row * b.Y + n0 + i16] = ((sp[o] + sp[256 + o]) + (sp[512 + o] + sp[768 + o])) * inv;
            }
        }
    }
    // the other site: the kept eigenvectors themselves, a share per host (two-dimensional walks: an integer division per element costs more
    // than the copy)
    if (going_left) {
        for (int k = host; k < nk; k += nhost)
            for (int y = tid; y < b.Y; y += BT_T) Wr[(int64_t)k * b.Y + y] = Ev[(int64_t)y * ldE + k];
        if (host == 0 && tid == 0) *v.label_site = lid;
    } else {
        const int k = tid & 31;
        for (int x = host * 16 + (tid >> 5); x < b.X; x += nhost * 16)
            if (k < nk) Wl[(int64_t)x * nk + k] = Ev[(int64_t)x * ldE + k];
        if (host == 0 && tid == 0) *v.label_site = lid + 1;
    }
}

// what the host knows of a tail launch: no pointer arithmetic on the device's scalar unit in front of the first request
struct TailArgs {
    const double *Sprev, *Oprev;    // environment rows of the side the eigenvectors live on / of the other side (null: chain end)
    const double *phS, *phO;        // the two sites' encoded series
    const double* M;                // bt_new as [c][k = O index][n = S index]: btn going left, btnT going right
    double* out;                    // the new environment rows
    int32_t lid, going_left, nsplit, nchain, flags;     // flags: 1 the next bond's overlaps are wanted, 2 leave phase stamps
    unsigned long long* span;       // stamped launch: (start, end) of every workgroup, or null
};

// Verification + re-orthonormalisation of the K candidate vectors Z ([c * BT_ZS + k], zero beyond the live rows / columns), all 8 waves:
// D = Z^T Z - I on the MFMA; |D| >= 1e-4 or a residual ||T z - lambda z|| above 1e-8 ||T|| (rres: this thread's, threads < K): the
// caller's fallback.  |D| < 1e-13: nothing to do.  |D| < 1e-8: Z <- Z (I - D/2) (Loewdin, first order: the deviation is squared).
// Else Z <- Z (I - D/2 + 3 D^2 / 8): the deviation becomes 5/8 |D|^3 <= 1.7e-14 for |D| <= 3e-5 - ONE pass over Z where
// k_eig_fin's verify_and_polish forms D twice (D^2 is a 32^3 product); beyond 3e-5 a second, first-order pass follows.
__device__ __forceinline__ bool tail_polish(double* __restrict__ Z, double* __restrict__ D, double* __restrict__ Dh, double* __restrict__ misc,
                                            const int n, const int K, const double rres, double& emax0, const bool worker) {
    // worker: one of the 8 waves that do the work; the workgroup's other waves (the role waves of k_bond_tail) only keep the barriers company
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int jl = lane & 15, q4 = lane >> 4;
    for (int pass = 0; pass < 2; ++pass) {
        // ---- D: wave w owns the 16 x 16 tile (w & 3) over rows [64 (w >> 2), +64) ----
        const int a0 = 16 * ((wave & 3) >> 1), b0 = 16 * (wave & 1), kb = 64 * (wave >> 2);
        d4 acc = {0.0, 0.0, 0.0, 0.0};
        if (worker) {
            // operands a batch of 8 k-steps ahead of the MFMAs that consume them (see k_bond_tail's overlap product)
            double da[2][8], db[2][8];
            auto fetch = [&](int bt) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const double* zr = Z + (kb + 4 * (8 * bt + u) + q4) * BT_ZS;
                    da[bt][u] = zr[a0 + jl];
                    db[bt][u] = zr[b0 + jl];
                }
            };
            fetch(0);
            fetch(1);
            asm volatile("" ::: "memory");
#pragma unroll
            for (int bt = 0; bt < 2; ++bt)
#pragma unroll
                for (int u = 0; u < 8; ++u) acc = mfma_f64(da[bt][u], db[bt][u], acc);
        }
        if (worker && wave >= 4) {
#pragma unroll
            for (int r = 0; r < 4; ++r) Dh[(wave & 3) * 256 + (q4 + 4 * r) * 16 + jl] = acc[r];
        }
        __syncthreads();
        double err = 0.0;
        if (wave < 4) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int aa = a0 + q4 + 4 * r, bb = b0 + jl;
                double dv = 0.0;
                if (aa < K && bb < K) {
                    dv = (acc[r] + Dh[wave * 256 + (q4 + 4 * r) * 16 + jl]) - (aa == bb ? 1.0 : 0.0);
                    err = fmax(err, fabs(dv));
                }
                D[aa * 32 + bb] = dv;
            }
        }
        err = wave_max(err);
        const double rr = wave_max(rres);
        if (worker && lane == 0) {
            misc[8 + wave] = err;
            misc[16 + wave] = rr;
        }
        __syncthreads();
        double emax = 0.0, rmax = 0.0;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            emax = fmax(emax, misc[8 + w]);
            rmax = fmax(rmax, misc[16 + w]);
        }
        if (pass == 0) emax0 = emax;
        if (!(rmax < 1e-8) || !(emax < 1e-4)) return false;        // (NaN fails both)
        if (emax < 1e-13) return true;
        if (pass == 1 && !(emax < 1e-8)) return false;             // the second pass starts below 1e-12: anything else is not a rounding effect
        const bool second = pass == 0 && !(emax < 1e-8);
        const double* Cm = D;
        if (second) {
            // Cm = D - 3/4 D^2 (the update below takes half of it): four tiles, waves 0..3
            if (wave < 4) {             // (workers)
                d4 c2 = {0.0, 0.0, 0.0, 0.0};
                double ca_[8], cb_[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    ca_[u] = D[(a0 + jl) * 32 + 4 * u + q4];
                    cb_[u] = D[(4 * u + q4) * 32 + b0 + jl];
                }
                asm volatile("" ::: "memory");
#pragma unroll
                for (int u = 0; u < 8; ++u) c2 = mfma_f64(ca_[u], cb_[u], c2);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int aa = a0 + q4 + 4 * r, bb = b0 + jl;
                    Dh[aa * 32 + bb] = fma(-0.75, c2[r], D[aa * 32 + bb]);
                }
            }
            Cm = Dh;
            __syncthreads();
        }
        // ---- Z <- Z - (Z Cm) / 2: 8 row tiles x 2 column tiles, two per wave ----
        d4 upd[2];
        upd[0] = d4{0.0, 0.0, 0.0, 0.0};
        upd[1] = d4{0.0, 0.0, 0.0, 0.0};
        if (worker) {
            // the two tiles of a wave share the row tile of Z: its operand once, both column tiles of Cm
            const int c0 = 16 * wave;
            double ua[8], ub0[8], ub1[8];
#pragma unroll
            for (int s4 = 0; s4 < 8; ++s4) {
                ua[s4] = Z[(c0 + jl) * BT_ZS + 4 * s4 + q4];
                ub0[s4] = Cm[(4 * s4 + q4) * 32 + jl];
                ub1[s4] = Cm[(4 * s4 + q4) * 32 + 16 + jl];
            }
            asm volatile("" ::: "memory");
            d4 a0_ = {0.0, 0.0, 0.0, 0.0}, a1_ = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s4 = 0; s4 < 8; ++s4) {
                a0_ = mfma_f64(ua[s4], ub0[s4], a0_);
                a1_ = mfma_f64(ua[s4], ub1[s4], a1_);
            }
            upd[0] = a0_;
            upd[1] = a1_;
        }
        __syncthreads();
        if (worker) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int tile = wave * 2 + h;
                const int c0 = 16 * (tile >> 1), ca = 16 * (tile & 1);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = c0 + q4 + 4 * r, aa = ca + jl;
                    if (c < n && aa < K) Z[c * BT_ZS + aa] -= 0.5 * upd[h][r];
                }
            }
        }
        __syncthreads();
        if (emax < 3e-5) return true;
    }
    return true;
}

// NDTensors truncate! (relative cutoff, mindim 1) with the K <= 32 largest eigenvalues in the lanes of the calling wave (lane k:
// lam_k, zero beyond K) instead of an array in LDS every thread walks (1.3 us of dependent reads).  The rule drops values from the small end
// while the discarded weight stays within cutoff * scale; the weight behind value m is base + sum_{i >= m} P_i (base: what lies beyond the K
// values, trace - sum), so the answer is the smallest m >= 1 whose tail qualifies - one prefix scan and one ballot.  The sums are taken in
// another order than truncate_rule()'s loop: the two can differ where a tail equals the threshold to the last bit.
__device__ __forceinline__ int truncate_rule_lanes(const double lam_lane, const int K, const int ns, const double tr, const double inv2, const double cutoff) {
    const int lane = threadIdx.x & 63;
    const double scale0 = tr * inv2;
    const double scale = scale0 == 0.0 ? 1.0 : scale0;
    const double P = lane < K ? lam_lane * inv2 : 0.0;
    const double pre = wave_incl_scan(P);
    const double kept = readlane_f64(pre, 63);
    double base = scale0 - kept;
    if (base < 0.0 || ns <= K) base = 0.0;
    const double tail = base + ((kept - pre) + P);
    const unsigned long long ok = __ballot(lane >= 1 && lane < K && tail <= cutoff * scale);
    if (ns <= 1 || ok == 0ull) return K;
    return __ffsll((long long)ok) - 1;
}

// D4: d == 4 (the headline shapes): on the left side a = u, s = kq are immediates
template <bool D4>
__device__ __forceinline__ void bond_tail_body(const View& v, const TailArgs& ta) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double red[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, kq = lane >> 4;
    const int lid = ta.lid, going_left = ta.going_left, nsplit = ta.nsplit, nchain = ta.nchain;
    const int want_next = ta.flags & 1;
    const int d = v.d;
    // One 16-series tile per workgroup.  The first nchain workgroups ALSO form the next bond's tensor (one job each), the next nsplit the
    // back-split (two tiles each) - after their tile work, from operands requested as soon as the overlap product has released its
    // registers.  (As workgroups of their own - 24 beside 256 tile workgroups at the headline shape - the roles shared CUs with
    // tiles and the launch took 22 us instead of 15, stamped; as extra waves of the tile workgroups they held their hosts' barriers up: 21 us.)
    const int bid = (int)blockIdx.x;
    const int role = bid < nchain ? 2 : (bid < nchain + nsplit ? 1 : 0);
    // phase stamps (100 MHz): DevScalars::eig_stamps[16..] workgroup 0 (hosts a job of the next bond's tensor when the sweep goes on),
    // [32..] the last workgroup (no role)
    unsigned long long* stp = nullptr;
    if ((ta.flags & 2) && tid == 0) {
        if (bid == 0) stp = v.sc->eig_stamps + 16;
        else if (bid == (int)gridDim.x - 1) stp = v.sc->eig_stamps + 32;
    }
    int sti = 0;
#define TSTAMP() do { if (stp) stp[sti++] = __builtin_amdgcn_s_memrealtime(); } while (0)
    TSTAMP();
    // ... and every workgroup of a stamped launch leaves its own start and end: the launch's span, and how the workgroups are spread over it
    unsigned long long* spanp = ((ta.flags & 2) && tid == 0 && ta.span && bid < 2048) ? ta.span + 2 * bid : nullptr;
    if (spanp) spanp[0] = __builtin_amdgcn_s_memrealtime();
    if ((ta.flags & 2) && tid == 0 && bid == 0) v.sc->eig_stamps[60] = gridDim.x;
#define TEND() do { if (spanp) spanp[1] = __builtin_amdgcn_s_memrealtime(); } while (0)
    double* Zl = smem;                             // [128][BT_ZS] candidates, then the kept eigenvectors E (zero beyond the live rows / kept columns)
    double* Dl = Zl + 128 * BT_ZS;                 // [32][32]
    double* Dh = Dl + 1024;                        // [1024]
    double* misc = Dh + 1024;                      // [128]
    double* Sf = misc + 128;                       // factors of the S side, then of the O side ...
    double* Of = Sf + BT_FAC;
    double* part = Sf;                             // ... and, once both are consumed, [4 quarters][2 column tiles][256] pieces of env'
    double* envs = part + 2048;                    // [16][BT_ENVS] new environment rows of the tile
    double* redy = envs + 16 * BT_ENVS;            // [8][16] the waves' pieces of yhat
    double* cpart = Dl;                            // [4][2][256] the chain role's partial tiles (the polish scratch and the S tile are long consumed)
    double* St = Dl;                               // [16][BT_SS] the dense S tile, once the polish is done
    const double* __restrict__ ws = v.eig_ws;
    // the loader role of a thread: threads [0, 256) the S side, [256, 512) the O side; 16 threads per series row, two bond entries and
    // one site state each
    const bool lower = tid < 256;
    const int lrow = (tid & 255) >> 4, lj = tid & 15;
    // ONE tile per tile workgroup, no loop around any of this: a loop invites the compiler to hoist the address arithmetic of every
    // phase - polish, roles, products - in front of it, and the kernel then lives in scratch memory (400 bytes per lane, measured)
    // ---- requests, first those that need nothing but the kernel arguments: the bond dimensions are a dependent (scalar) load from memory,
    // about a microsecond on a cold start, and everything asked for before their first use is in flight by the time they arrive ----
    double zin[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) zin[m] = ws[WS_Z + tid + m * BT_T];
    const double triflag = ws[WS_MISC + 3], tnorm_in = ws[WS_MISC + 2];
    double lam_in = ws[WS_LAM + (lane & 31)];                            // every wave: the truncation rule runs in its lanes (masked below)
    const double res_in = ws[WS_RES + (tid & 31)];                       // (used by threads < nk only)
    const int redo_in = v.sc->redo;
    TSTAMP();      // [1] candidates requested
    Span tl{0, 0, 0, 0};
    double fe0 = 0.0, fe1 = 0.0, fp = 0.0;
    {
        tl = tile_span_k(v, bid);                   // (count 0 beyond the last tile: a workgroup that is there for its role only)
        if (lower || want_next) {
            // whole rows (the capacity is a kernel argument); what lies beyond the live bond is dropped when the dimensions are known
            const double* prev = lower ? ta.Sprev : ta.Oprev;
            const double* ph = lower ? ta.phS : ta.phO;
            const bool valid = lrow < tl.count;
            const int64_t smp = tl.count > 0 ? tl.start + (valid ? lrow : 0) : 0;
            // (unconditional loads from clamped addresses: see tail_chain_request; a tile beyond the data set has count 0 and start 0)
            const double x0 = prev ? prev[smp * v.cap + min(lj, v.cap - 1)] : 1.0;
            const double x1 = prev ? prev[smp * v.cap + min(lj + 16, v.cap - 1)] : 1.0;
            const double xp = ph[smp * d + min(lj, d - 1)];
            fe0 = (valid && lj < v.cap) ? x0 : 0.0;
            fe1 = (valid && lj + 16 < v.cap) ? x1 : 0.0;
            fp = (valid && lj < d) ? xp : 0.0;
        }
    }
    TSTAMP();      // [2] factors requested
    // ---- ... then those that need the bond dimensions -----------------------------------------------------------------------
    const EigProblem pb = resolve(v, lid, going_left, nullptr, 0, 0);
    const BondDimsF b = bond_dims_f(v, lid);
    // (the outer bond of the neighbouring site, for the chain role: asked for with the other dimensions, not a dependent round trip later)
    const int Dnb = role == 2 ? (going_left ? v.chi[max(lid - 1, 0)] : v.chi[min(lid + 3, v.T)]) : 1;
    const int n = pb.n, K0 = pb.K0, nspec = pb.nspec;
    // S: the side the kept eigenvectors live on (Y going left, X going right); O: the other side
    const int DS = ta.Sprev ? (going_left ? b.Dr : b.Dl) : 1, DO = ta.Oprev ? (going_left ? b.Dl : b.Dr) : 1;
    const KrSide ks = kr_side(Sf, DS, d, !going_left), ko = kr_side(Of, DO, d, going_left != 0);
    const int KO = going_left ? b.X : b.Y, NS = going_left ? b.Y : b.X;      // bt_new as [k = O index][n = S index]
    const int KP = (KO + 3) & ~3, ZS = DS * d, ZP = (ZS + 3) & ~3;
    if (stp) { asm volatile("" :: "s"(KO), "s"(NS)); }
    TSTAMP();      // [3] bond dimensions known
    d4 pacc0 = {0.0, 0.0, 0.0, 0.0}, pacc1 = {0.0, 0.0, 0.0, 0.0};
    double bm[32];          // requested operands of the workgroup's role: the tile's slice of bt_new / see tail_chain_request, tail_split_load
    if (want_next) {
        // this wave's 16 columns of bt_new, every k-step: no predicates (sixteen exec-masked loads in a row keep the memory
        // pipeline from ever holding a tile's worth of requests).  Rows beyond the live ones meet zeros of the O side, columns
        // beyond them zeros of z: their (finite) values are read from clamped addresses and do not matter.
        const unsigned col = (unsigned)min(16 * wave + i16, NS - 1);
        const double* __restrict__ M = ta.M + (int64_t)tl.cls * b.L;        // (uniform base + 32-bit lane offsets)
#pragma unroll
        for (int u = 0; u < 32; ++u) bm[u] = M[(unsigned)min(4 * u + kq, KO - 1) * (unsigned)NS + col];
    }
    const double gd_ = pb.G[(size_t)min(tid, n - 1) * (n + 1)];
    const double gdiag = tid < n ? gd_ : 0.0;
    lam_in = lane < K0 ? lam_in : 0.0;
    {
        const int Dp = lower ? DS : DO;
        const bool bnd = (lower ? ta.Sprev : ta.Oprev) == nullptr;      // chain end: the one live entry is 1
        fe0 = lj < Dp ? fe0 : 0.0;
        fe1 = (lj + 16 < Dp && !bnd) ? fe1 : 0.0;
    }
    TSTAMP();      // [4] everything requested
    {
        double* fac = lower ? Sf : Of;
        fac[lrow * BT_ELS + lj] = fe0;
        fac[lrow * BT_ELS + lj + 16] = fe1;
        if (lj < BT_ELS - 32) fac[lrow * BT_ELS + 32 + lj] = 0.0;
        fac[16 * BT_ELS + lrow * BT_PLS + lj] = fp;
        if (lj < BT_PLS - 16) fac[16 * BT_ELS + lrow * BT_PLS + 16 + lj] = 0.0;
    }
    {
        const double trw = wave_sum(gdiag);         // pieces of the trace: they meet at the barrier the factors need anyway
        if (lane == 0) red[wave] = trw;
    }
    __syncthreads();
    TSTAMP();      // [5] factors in LDS, trace pieces published
    // ---- k_eig_fin's work, by every workgroup for itself: trace, truncation rule, verification, polish ---------------------------------
    double tr = 0.0;
    for (int i = 0; i < BT_T / 64; ++i) tr += red[i];
    const double inv = v.rescale_after ? 1.0 / sqrt(tr) : 1.0;
    const int nk = truncate_rule_lanes(lam_in, K0, nspec, tr, inv * inv, v.cutoff);
    TSTAMP();      // [4] trace, truncation rule
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int i = tid + m * BT_T;
        const int c = i >> 5, kk = i & 31;
        Zl[c * BT_ZS + kk] = (c < n && kk < nk) ? zin[m] : 0.0;
    }
    __syncthreads();
    TSTAMP();      // [5] candidates in LDS
    bool ok = triflag == 1.0 && redo_in == 0 && !(ta.flags & 4);
    double emax0 = 0.0;
    if (ok) ok = tail_polish(Zl, Dl, Dh, misc, n, nk, (tid < nk ? res_in : 0.0) / (tnorm_in > 0.0 ? tnorm_in : 1.0), emax0, true);
    TSTAMP();      // [6] verified + polished
    if (!ok) {
        if (bid == 0 && tid == 0 && redo_in == 0) {
            v.sc->redo = 1 + (going_left ? v.T - 2 - lid : v.T - 1 + lid);      // 1 + the bond's position in the sweep
            v.sc->eig_fallbacks += 1;
        }
        return;
    }
    // P = O bt_new AFTER the polish: its operand (128 KB of bt_new per workgroup: the CU's L1 moves 64 bytes a cycle) has arrived by now,
    // the candidates came first
    {
        if (want_next) {
            // P = O bt_new, this wave's 16 columns.  A wave issues one MFMA per 64 cycles whatever their dependencies (profiles/ubench/
            // mfma_rate.hip), so what a product costs is 64 cycles per MFMA PLUS every operand latency the wave waits out in between:
            // operands are fetched a batch of 8 k-steps ahead (left to itself the compiler reads, waits, multiplies, reads ...).
            // Whole batches: beyond the live extent the O side is zero.
            double ab[4][8];
            auto fetch = [&](int bt) {
                if (D4 && ko.left) {
                    const double* er = ko.env + i16 * BT_ELS + 8 * bt;
#pragma unroll
                    for (int u = 0; u < 8; ++u) ab[bt][u] = er[u];
                } else {
#pragma unroll
                    for (int u = 0; u < 8; ++u) ab[bt][u] = kr_at(ko, i16, 4u * (8 * bt + u) + kq);
                }
            };
            const double phf = (D4 && ko.left) ? ko.ph[i16 * BT_PLS + kq] : 1.0;
            fetch(0);
#pragma unroll
            for (int bt = 0; bt < 4; ++bt) {
                if (bt < 3) fetch(bt + 1);
                asm volatile("" ::: "memory");
                if (32 * bt < KP) {
#pragma unroll
                    for (int u = 0; u < 8; u += 2) {
                        pacc0 = mfma_f64(ab[bt][u] * phf, bm[8 * bt + u], pacc0);
                        pacc1 = mfma_f64(ab[bt][u + 1] * phf, bm[8 * bt + u + 1], pacc1);
                    }
                }
            }
        }
    }
    TSTAMP();      // [10] P issued
    // the role's operands into the registers the overlap product has just released: they arrive under the rest of the tile's work
    // (registers of their own: the overlap product's MFMAs may still be reading bm)
    double rr[16];
    if (role == 2) tail_chain_request(v, b, lid, going_left, bid, rr, Dnb);
    else if (role == 1) tail_split_request(v, b, going_left, bid - nchain, rr);
    TSTAMP();      // role operands requested
    if (bid == 0 && wave == 0) {                    // publication (fin_body)
        if (lane < K0) v.lam[lane] = lam_in;
        bool bad = !(tr == tr) || tr > 1e300;
        const double P = lam_in * inv * inv;
        if (__ballot(lane < K0 && (!(P == P) || P > 1e300))) bad = true;
        if (lane == 0) {
            // how far from orthonormal the candidates were, by class (diagnostics: mpst_get_tail_phases)
            v.sc->eig_stamps[56 + (emax0 < 1e-13 ? 0 : emax0 < 1e-8 ? 1 : emax0 < 3e-5 ? 2 : 3)] += 1ull;
            v.sc->n_keep = nk;
            v.sc->n_spec = K0;
            v.sc->bt_norm2 = tr;
            v.sc->inv_norm = inv;
            v.sc->eig_sweeps = 0;
            if (bad) v.sc->status = MPST_ERR_SVD;
            v.chi[lid + 1] = nk;
        }
    }
    const double* __restrict__ Ef = Zl;
    // ---- the dense S tile, by all waves (env' then pays one LDS read per MFMA, whichever side S is).  It goes over the polish scratch:
    // the barrier keeps it off the error pieces a slower wave may still be reading (tail_polish returns without one when nothing is to do)
    lds_barrier();                                  // B1 (LDS-only barriers from here on: __syncthreads() would also wait for the role's operands)
    TSTAMP();      // B1 passed
    {
        const int row = tid >> 5, z0 = tid & 31;    // 4 entries per thread: z0, z0 + 32, ...
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int z = z0 + 32 * m;
            St[row * BT_SS + z] = z < ZS ? kr_at(ks, row, (unsigned)z) : 0.0;
        }
    }
    lds_barrier();                                  // B2
    TSTAMP();      // [7] S tile formed
    // env' = S E, the sums of k_env / k_env_split (mpst_internal.h: four chains over the quarters of the contraction): wave = (column tile,
    // quarter), the quarters meet in LDS as (q0 + q1) + (q2 + q3).  At most 32 vectors are kept here: two column tiles.
    {
        const int nsteps = ZP >> 2, ks4 = env_ks4(nsteps);
        const int nt = wave & 1, q = wave >> 1;
        const int col = nt * 16 + i16;
        d4 acc = {0.0, 0.0, 0.0, 0.0};
        if (nt * 16 < nk) {
            double sa[8], sb[8];                    // both operands of the quarter first, then eight MFMAs back to back
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int st0 = q * ks4 + u, step = min(st0, nsteps - 1);
                sa[u] = St[i16 * BT_SS + 4 * step + kq];
                const double e = Ef[(4 * step + kq) * BT_ZS + col];
                sb[u] = (u < ks4 && st0 < nsteps) ? e : 0.0;        // (a step beyond the quarter adds an exact +0: no predicate on the MFMA)
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = mfma_f64(sa[u], sb[u], acc);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) part[(q * 2 + nt) * 256 + r * 64 + lane] = acc[r];      // (both factor sets are consumed: P is issued, S is dense)
    }
    lds_barrier();                                  // B3
    {
        const int nt = tid >> 8, e = tid & 255;
        const double sum = (part[nt * 256 + e] + part[(2 + nt) * 256 + e]) + (part[(4 + nt) * 256 + e] + part[(6 + nt) * 256 + e]);
        const int i = ((e >> 4) & 3) + 4 * (e >> 6), col = nt * 16 + (e & 15);
        if (i < tl.count && col < nk) ta.out[(int64_t)(tl.start + i) * v.cap + col] = sum;
        envs[i * BT_ENVS + col] = sum;
    }
    TSTAMP();      // [8] new environment rows
    if (want_next) {
        lds_barrier();                              // B4 (LDS only: __syncthreads() would wait for the acknowledgement of the rows just stored)
        {
            // z = E env'^T for this wave's 16 columns, in the accumulator layout of P; yhat piece = sum over the columns of P .* z
            d4 zacc = {0.0, 0.0, 0.0, 0.0};
            double za[8], zb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                za[u] = envs[i16 * BT_ENVS + 4 * u + kq];
                zb[u] = Ef[(16 * wave + i16) * BT_ZS + 4 * u + kq];
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int u = 0; u < 8; ++u) zacc = mfma_f64(za[u], zb[u], zacc);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double x = sum16((pacc0[r] + pacc1[r]) * zacc[r]);
                if (i16 == 0) redy[wave * 16 + kq + 4 * r] = x;
            }
        }
        lds_barrier();                              // B5
        TSTAMP();  // [9] z, row dot
        if (tid < 16 && tid < tl.count) {
            const double y = (((redy[tid] + redy[16 + tid]) + (redy[32 + tid] + redy[48 + tid])) +
                              ((redy[64 + tid] + redy[80 + tid]) + (redy[96 + tid] + redy[112 + tid]))) * inv;
            // the reader (k_grad_s) adds the eight slice slots of a series in order: the overlap in slot 0, zeros behind it
            double2* yp = (double2*)(v.ypart + (int64_t)(tl.start + tid) * YS_MAXSL);
            yp[0] = make_double2(y, 0.0);
            yp[1] = make_double2(0.0, 0.0);
            yp[2] = make_double2(0.0, 0.0);
            yp[3] = make_double2(0.0, 0.0);
        }
    }
    TSTAMP();      // [10] tile done
    // ---- the workgroup's role, if it has one: all waves, operands long in registers, E still in LDS ------------------------------------
    if (role == 2) tail_chain_job(v, b, lid, going_left, bid, rr, cpart, Ef, BT_ZS, nk, inv, Dnb);
    else if (role == 1) tail_split_job(v, b, lid, going_left, bid - nchain, nsplit, rr, cpart, Ef, BT_ZS, nk, inv);
    if (role != 0) TSTAMP();  // [11] role done (stores in flight)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TSTAMP();      // [12] stores drained
    TEND();
#undef TSTAMP
#undef TEND
}
template <bool D4>
__global__ __launch_bounds__(BT_T) void k_bond_tail(View v, TailArgs ta) {
    bond_tail_body<D4>(v, ta);
}

// ---- the kernels proper: one fit per launch (View in the kernel arguments), or K independent fits of the same shape per
// launch (blockIdx.z picks the fit's View from a device array - mpst_sweep_batch): the command processor dispatches about
// 70 k kernels a second however many queues feed it, which caps K concurrent single-fit chains at 2.3x one chain; one
// chain of K-fold launches keeps the kernel count of ONE fit.
__global__ __launch_bounds__(256) void k_gram_upd(View v, int lid, int going_left, int first_iter) { gram_upd_body(v, lid, going_left, first_iter, (int)blockIdx.x); }
__global__ __launch_bounds__(256) void k_gram_upd_b(const View* __restrict__ vs, int lid, int going_left, int first_iter) {
    // the tiles of one fit read the same two matrices (bt, the gradient): side by side on one XCD, they meet in its L2 (xcd_contiguous)
    int bx = (int)blockIdx.x, by = 0, bz = (int)blockIdx.z;
    xcd_contiguous(bx, by, bz);
    const View& v = vs[bz];       // by reference: a local copy would live in per-lane scratch (the class tables are indexed dynamically)
    gram_upd_body(v, lid, going_left, first_iter, bx);
}
__global__ __launch_bounds__(256, 2) void k_env_split(View v, int lid, int going_left, int site, int left_side, const double* __restrict__ prev,
                                                   int prev_bond, int out_bond, double* __restrict__ out, int nsplit, int ntb, int tp) {
    env_split_body(v, lid, going_left, site, left_side, prev, prev_bond, out_bond, out, nsplit, ntb, tp, (int)blockIdx.x);
}
// batched: the environment rows are addressed by element offsets into the fit's own LE / RE (prev_off < 0: boundary)
__global__ __launch_bounds__(256, 2) void k_env_split_b(const View* __restrict__ vs, int lid, int going_left, int site, int left_side, int64_t prev_off,
                                                     int prev_bond, int out_bond, int64_t out_off, int nsplit, int ntb, int tp) {
    // a fit's workgroups all read its eigenvectors E (32 KB each) and, the split / chain ones, bt_new: one XCD per fit (xcd_contiguous)
    int bx = (int)blockIdx.x, by = 0, bz = (int)blockIdx.z;
    xcd_contiguous(bx, by, bz);
    const View& v = vs[bz];       // by reference: a local copy would live in per-lane scratch (the class tables are indexed dynamically)
    double* base = left_side ? v.LE : v.RE;
    env_split_body(v, lid, going_left, site, left_side, prev_off >= 0 ? base + prev_off : nullptr, prev_bond, out_bond, base + out_off, nsplit, ntb, tp, bx);
}
template <int LM, bool D4, bool V2> __global__ __launch_bounds__(YS_T) void k_yhat_s(View v, int lid, int nslc, int ngw) {
    yhat_s_body<LM, D4, V2>(v, lid, nslc, ngw, (int)blockIdx.x, (int)blockIdx.y);
}
template <int LM, bool D4, bool V2> __global__ __launch_bounds__(YS_T) void k_yhat_s_b(const View* __restrict__ vs, int lid, int nslc, int ngw) {
    // the slices of one group walker read the same series (a full environment row each): side by side on one XCD (xcd_contiguous)
    int bx = (int)blockIdx.x, by = (int)blockIdx.y, bz = (int)blockIdx.z;
    xcd_contiguous(bx, by, bz);
    const View& v = vs[bz];       // by reference: a local copy would live in per-lane scratch (the class tables are indexed dynamically)
    const int sl = bx % nslc, gw = bx / nslc;           // logical x: slices fastest (the body wants gw = x % ngw)
    yhat_s_body<LM, D4, V2>(v, lid, nslc, ngw, sl * ngw + gw, by);
}
// KC = series per stage.  Measured with KC = 128 under __launch_bounds__(GS_T, 4) (128 VGPRs, 38 spilled; two workgroups per CU so that one
// stages while the other multiplies - with one workgroup per CU the staging of a stage, 1.3 us of 4.2, leaves the matrix pipe idle): it
// LOSES, 40.5 -> 63.9 us with 8 fits per launch, 140 -> 161 us with 32 (profiles/r06_batched_gemm_ab.txt); 256 everywhere.
// NW = waves per workgroup.  8 (512 threads, one workgroup per CU at 174+ registers) where a launch has about one workgroup per CU and
// latency counts (a single fit); 4 (256 threads, two workgroups per CU, each wave a quarter of a stage) for contexts advanced in batches:
// one workgroup stages its next 256 series (1.3 us of the 4.2 a stage takes, all eight waves idle on the matrix pipe meanwhile -
// profiles/r06_batched_gemm_ab.txt) while the other multiplies.  The wave count fixes which wave sums which series, so it belongs to the
// context (View::b2_nw): its solo and its batched sweeps agree bit for bit.
template <int AW2, int D2, int FS, int KC, int NW> __global__ __launch_bounds__(64 * NW, 2) void k_grad_s(View v, int lid, int ksplit, int nbc) {
    grad_s_body<AW2, D2, FS, KC, NW>(v, lid, ksplit, nbc, (int)blockIdx.x, (int)blockIdx.y);
}
template <int AW2, int D2, int FS, int KC, int NW> __global__ __launch_bounds__(64 * NW, 2) void k_grad_s_b(const View* __restrict__ vs, int lid, int ksplit, int nbc) {
    int bx = (int)blockIdx.x, by = (int)blockIdx.y, bz = (int)blockIdx.z;
    xcd_contiguous(bx, by, bz);
    const View& v = vs[bz];       // by reference: a local copy would live in per-lane scratch (the class tables are indexed dynamically)
    // logical x: blocks fastest, then shares (the body wants ks = x % ksplit)
    const int nblk = nbc * nbc;
    const int blk = bx % nblk, ks = bx / nblk;
    grad_s_body<AW2, D2, FS, KC, NW>(v, lid, ksplit, nbc, blk * ksplit + ks, by);
}

// the loss of the bond from the pieces of k_grad_s (same order wherever it is formed): gradbuf[0..1] for the all-reduce
__global__ __launch_bounds__(64) void k_loss_sum(View v) {
    if (threadIdx.x == 0) {
        v.gradbuf[0] = bond_loss(v);
        v.gradbuf[1] = 0.0;
    }
}

// ---- launchers ------------------------------------------------------------------------------------------------------
static inline int cdivf(int a, int b) { return (a + b - 1) / b; }

void launch_bond_fused(const View& v, int lid, int assemble, hipStream_t s) {
    if (v.d <= 4) hipLaunchKernelGGL(k_bond_fused<4>, dim3(v.nparts), dim3(FUSED_T), 0, s, v, lid, assemble);
    else hipLaunchKernelGGL(k_bond_fused<8>, dim3(v.nparts), dim3(FUSED_T), 0, s, v, lid, assemble);
}
void launch_fused_reduce(const View& v, int lid, hipStream_t s) {
    hipLaunchKernelGGL(k_fused_reduce, dim3(v.n_norm_part), dim3(256), 0, s, v, lid);
}
// ---- sliced bond GEMMs: host-side geometry (depends on capacity, d, C and the data set only) -------------------------------
static int b2_aw(const View& v) { return std::max(1, 32 / v.d); }
int b2_blocks_cap(const View& v) {
    const int nbc = cdivf(v.cap, b2_aw(v));
    return nbc * nbc;
}
// shares per gradient block: enough workgroups to fill the chip (~256 in all), at least one stage of series each
int b2_ksplit(const View& v, int64_t max_pass) {
    if (const char* e = getenv("MPST_B2_KSPLIT")) return std::max(1, std::min(atoi(e), GS_MAXKS));
    const int groups = std::max(1, v.C * b2_blocks_cap(v));
    int64_t ks = std::max(1, 256 / groups);
    ks = std::min<int64_t>(ks, std::max<int64_t>(1, (max_pass + GS_KC - 1) / GS_KC));
    return (int)std::max<int64_t>(1, std::min<int64_t>(ks, GS_MAXKS));
}
int64_t b2_partial_elems(const View& v, int64_t max_pass) { return (int64_t)v.C * b2_blocks_cap(v) * b2_ksplit(v, max_pass) * 1024; }
static bool yhat_s_v2(const View& v) { return v.d == 4 && v.cap <= 32 && !(v.cap & 1) && getenv("MPST_YS_V1") == nullptr; }      // 16-byte row loads
static size_t yhat_s_lds(const View& v) { return (size_t)128 * ((yhat_s_v2(v) ? 2 : 1) * (v.cap + 1 + v.d + 1) + 17) * sizeof(double); }
static size_t grad_s_lds(const View& v) { return std::max((size_t)GS_KC * (((2 * b2_aw(v) + 2 * v.d + 1) | 1) + 1), (size_t)32 * 256) * sizeof(double); }
hipError_t b2_init_attrs(int device) {
    static std::atomic<unsigned long long> done{0};
    if (device >= 0 && device < 64 && (done.load(std::memory_order_acquire) >> device) & 1ull) return hipSuccess;
    hipError_t e;
    if ((e = hipFuncSetAttribute((const void*)k_yhat_s<2, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)k_yhat_s<2, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)k_yhat_s<2, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)k_yhat_s<4, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)k_grad_s<1, 1, 0, 256, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)k_grad_s<1, 1, 25, 256, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)k_grad_s<1, 1, 25, 256, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)k_grad_s<2, 1, 0, 256, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)k_grad_s<1, 2, 0, 256, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)k_yhat_s_b<2, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)k_yhat_s_b<2, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)k_yhat_s_b<2, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)k_yhat_s_b<4, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)k_grad_s_b<1, 1, 0, 256, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)k_grad_s_b<1, 1, 25, 256, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)k_grad_s_b<1, 1, 25, 256, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)k_grad_s_b<2, 1, 0, 256, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)k_grad_s_b<1, 2, 0, 256, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)k_bond_tail<true>, hipFuncAttributeMaxDynamicSharedMemorySize, BT_LDS_DOUBLES * (int)sizeof(double))) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)k_bond_tail<false>, hipFuncAttributeMaxDynamicSharedMemorySize, BT_LDS_DOUBLES * (int)sizeof(double))) != hipSuccess) return e;
    if (device >= 0 && device < 64) done.fetch_or(1ull << device, std::memory_order_release);
    return hipSuccess;
}
void launch_yhat_s(const View& v, int lid, hipStream_t s) {
    const int nslc = cdivf(v.d * v.cap, YS_W);
    const int ngroups = cdivf(v.ntiles, 8);
    const int ngw = std::max(1, std::min(ngroups, std::max(1, 512 / nslc)));      // group walkers per slice
    const dim3 grid(nslc * ngw, v.loss == MPST_LOSS_MSE ? v.C : 1);
    if (yhat_s_v2(v)) hipLaunchKernelGGL((k_yhat_s<2, true, true>), grid, dim3(YS_T), yhat_s_lds(v), s, v, lid, nslc, ngw);
    else if (v.cap <= 32 && v.d == 4) hipLaunchKernelGGL((k_yhat_s<2, true, false>), grid, dim3(YS_T), yhat_s_lds(v), s, v, lid, nslc, ngw);
    else if (v.cap <= 32) hipLaunchKernelGGL((k_yhat_s<2, false, false>), grid, dim3(YS_T), yhat_s_lds(v), s, v, lid, nslc, ngw);
    else hipLaunchKernelGGL((k_yhat_s<4, false, false>), grid, dim3(YS_T), yhat_s_lds(v), s, v, lid, nslc, ngw);
}
void launch_loss_sum(const View& v, hipStream_t s) { hipLaunchKernelGGL(k_loss_sum, dim3(1), dim3(64), 0, s, v); }

// ---- batched launchers: v = the shape every fit of the batch shares, vs = the K Views on the device ----
void launch_yhat_s_b(const View& v, const View* vs, int K, int lid, hipStream_t s) {
    const int nslc = cdivf(v.d * v.cap, YS_W);
    const int ngroups = cdivf(v.ntiles, 8);
    // group walkers per slice: about 512 workgroups over the whole batch - fewer, longer walks per fit amortise a workgroup's
    // start-up and its slice of B_c over more series (the series a walker takes do not change any sum)
    const int ngw = std::max(1, std::min(ngroups, std::max(1, 512 / (nslc * K))));
    const dim3 grid(nslc * ngw, v.loss == MPST_LOSS_MSE ? v.C : 1, K);
    if (yhat_s_v2(v)) hipLaunchKernelGGL((k_yhat_s_b<2, true, true>), grid, dim3(YS_T), yhat_s_lds(v), s, vs, lid, nslc, ngw);
    else if (v.cap <= 32 && v.d == 4) hipLaunchKernelGGL((k_yhat_s_b<2, true, false>), grid, dim3(YS_T), yhat_s_lds(v), s, vs, lid, nslc, ngw);
    else if (v.cap <= 32) hipLaunchKernelGGL((k_yhat_s_b<2, false, false>), grid, dim3(YS_T), yhat_s_lds(v), s, vs, lid, nslc, ngw);
    else hipLaunchKernelGGL((k_yhat_s_b<4, false, false>), grid, dim3(YS_T), yhat_s_lds(v), s, vs, lid, nslc, ngw);
}
void launch_grad_s_b(const View& v, const View* vs, int K, int lid, hipStream_t s) {
    const int aw = b2_aw(v), nbc = cdivf(v.cap, aw);
    const dim3 grid(v.b2_ksplit * nbc * nbc, v.C, K);
    if (aw > 8) hipLaunchKernelGGL((k_grad_s_b<2, 1, 0, 256, 8>), grid, dim3(GS_T), grad_s_lds(v), s, vs, lid, v.b2_ksplit, nbc);
    else if (v.d > 8) hipLaunchKernelGGL((k_grad_s_b<1, 2, 0, 256, 8>), grid, dim3(GS_T), grad_s_lds(v), s, vs, lid, v.b2_ksplit, nbc);
    else if (v.d == 4 && v.b2_nw == 4) hipLaunchKernelGGL((k_grad_s_b<1, 1, 25, 256, 4>), grid, dim3(256), grad_s_lds(v), s, vs, lid, v.b2_ksplit, nbc);
    else if (v.d == 4) hipLaunchKernelGGL((k_grad_s_b<1, 1, 25, 256, 8>), grid, dim3(GS_T), grad_s_lds(v), s, vs, lid, v.b2_ksplit, nbc);
    else hipLaunchKernelGGL((k_grad_s_b<1, 1, 0, 256, 8>), grid, dim3(GS_T), grad_s_lds(v), s, vs, lid, v.b2_ksplit, nbc);
}
void launch_gram_upd_b(const View& v, const View* vs, int K, int lid, int going_left, int first_iter, hipStream_t s) {
    const int dm = v.d * v.cap;
    hipLaunchKernelGGL(k_gram_upd_b, dim3(cdivf(dm, 16) * cdivf(dm, 16), 1, K), dim3(256), 0, s, vs, lid, going_left, first_iter);
}
void launch_env_split_b(const View& v, const View* vs, int K, int lid, int going_left, int site, int left_side, int64_t prev_off, int prev_bond,
                        int out_bond, int64_t out_off, int chain, hipStream_t s) {
    const int dm = v.d * v.cap;
    const int nsplit = cdivf(v.C * cdivf(dm, 16) * cdivf(v.cap, 16), 4);
    const int nchain = chain ? v.C * v.d * cdivf(v.cap, 16) : 0;
    // tile blocks: about 512 over the whole batch (as one fit of K times the series would get), each walking its fit's tiles
    // (K = 8 at the headline shape: 66.9 -> 64.7 ms per batched sweep; walkers of k_yhat_s_b and shares of k_grad_s_b scanned flat)
    constexpr int envb = 512;
    const int tp = (v.cap <= 32 && v.ntiles * K >= 512 && v.ntiles >= 2 * std::max(1, envb / K)) ? 2 : 1;
    const size_t lds = std::max((size_t)tp * 16 * FXS, chain ? (size_t)4 * CHAIN_J * 256 : (size_t)0) * sizeof(double);
    const int ntb = std::max(1, std::min(v.ntiles, std::max(1, envb / K)));
    hipLaunchKernelGGL(k_env_split_b, dim3(ntb + nsplit + nchain, 1, K), dim3(256), lds, s, vs, lid, going_left, site, left_side, prev_off, prev_bond,
                       out_bond, out_off, nsplit, ntb, tp);
}
void launch_grad_s(const View& v, int lid, hipStream_t s) {
    const int aw = b2_aw(v), nbc = cdivf(v.cap, aw);
    const dim3 grid(v.b2_ksplit * nbc * nbc, v.C);
    if (aw > 8) hipLaunchKernelGGL((k_grad_s<2, 1, 0, 256, 8>), grid, dim3(GS_T), grad_s_lds(v), s, v, lid, v.b2_ksplit, nbc);         // d = 2, 3
    else if (v.d > 8) hipLaunchKernelGGL((k_grad_s<1, 2, 0, 256, 8>), grid, dim3(GS_T), grad_s_lds(v), s, v, lid, v.b2_ksplit, nbc);   // d = 9..16
    else if (v.d == 4 && v.b2_nw == 4) hipLaunchKernelGGL((k_grad_s<1, 1, 25, 256, 4>), grid, dim3(256), grad_s_lds(v), s, v, lid, v.b2_ksplit, nbc);
    else if (v.d == 4) hipLaunchKernelGGL((k_grad_s<1, 1, 25, 256, 8>), grid, dim3(GS_T), grad_s_lds(v), s, v, lid, v.b2_ksplit, nbc);
    else hipLaunchKernelGGL((k_grad_s<1, 1, 0, 256, 8>), grid, dim3(GS_T), grad_s_lds(v), s, v, lid, v.b2_ksplit, nbc);
}
// the four-launch chain: real fp64, KLD, at most 32 kept vectors (the 32-column layout of the eigenvector block), tridiagonal solver
bool bond_tail_supported(const View& v) {
    return v.zw != 2 && v.loss == MPST_LOSS_KLD && v.chi_max <= 32 && v.cap <= 32 && v.d * v.cap <= MAX_DIM && v.svd_alg != MPST_SVD_JACOBI && v.d >= 2 && v.d <= 16;
}
void launch_bond_tail(const View& v, int lid, int going_left, int chain, int want_next /* bit 0; bit 2: forced failure (test hook) */, unsigned long long* span, hipStream_t s) {
    const int dm = v.d * v.cap, rid = lid + 1;
    TailArgs ta;
    ta.lid = lid;
    ta.going_left = going_left;
    ta.nsplit = cdivf(v.C * cdivf(dm, 16) * cdivf(v.cap, 16), 2);        // hosts of the back-split: two tiles (of the capacity layout) each
    ta.nchain = chain ? v.C * v.d * cdivf(v.cap, 16) : 0;
    ta.flags = (want_next & 1) | (want_next & 4);      // bit 2: test hook - this launch reports a failed verification
    ta.span = span;
    const int64_t cs = (int64_t)v.N * v.cap;
    const double* LEp = lid > 0 ? v.LE + (int64_t)(lid - 1) * cs : nullptr;
    const double* REn = rid < v.T - 1 ? v.RE + (int64_t)(rid + 1) * cs : nullptr;
    const double* phl = v.phi + (int64_t)lid * v.N * v.d;
    const double* phr = v.phi + (int64_t)rid * v.N * v.d;
    ta.Sprev = going_left ? REn : LEp;
    ta.Oprev = going_left ? LEp : REn;
    ta.phS = going_left ? phr : phl;
    ta.phO = going_left ? phl : phr;
    ta.M = going_left ? v.btn : v.btnT;
    ta.out = going_left ? v.RE + (int64_t)rid * cs : v.LE + (int64_t)lid * cs;
    // which bond's tail leaves its phase stamps (mpst_get_tail_phases): MPST_TAIL_STAMP="lid,going_left", default the middle bond going left
    static const int stamp_lid = [] { const char* e = getenv("MPST_TAIL_STAMP"); return e ? atoi(e) : -1; }();
    static const int stamp_dir = [] { const char* e = getenv("MPST_TAIL_STAMP"); const char* q = e ? strchr(e, ',') : nullptr; return q ? atoi(q + 1) : 1; }();
    if (lid == (stamp_lid >= 0 ? stamp_lid : (v.T - 1) / 2) && (going_left != 0) == (stamp_dir != 0)) ta.flags |= 2;
    const size_t lds = (size_t)BT_LDS_DOUBLES * sizeof(double);
    const dim3 grid(std::max(v.ntiles, ta.nchain + ta.nsplit));           // one 16-series tile per workgroup; the first ones carry a role as well
    if (v.d == 4) hipLaunchKernelGGL(k_bond_tail<true>, grid, dim3(BT_T), lds, s, v, ta);
    else hipLaunchKernelGGL(k_bond_tail<false>, grid, dim3(BT_T), lds, s, v, ta);
}
void launch_grad_norm(const View& v, int lid, hipStream_t s) {
    hipLaunchKernelGGL(k_grad_norm, dim3(v.n_norm_part), dim3(64), 0, s, v, lid);
}
void launch_gram_upd(const View& v, int lid, int going_left, int first_iter, hipStream_t s) {
    const int dm = v.d * v.cap;
    hipLaunchKernelGGL(k_gram_upd, dim3(cdivf(dm, 16) * cdivf(dm, 16)), dim3(256), 0, s, v, lid, going_left, first_iter);
}
void launch_env_split(const View& v, int lid, int going_left, int site, int left_side, const double* prev, int prev_bond,
                      int out_bond, double* out, int chain, hipStream_t s) {
    const int dm = v.d * v.cap;
    const int nsplit = cdivf(v.C * cdivf(dm, 16) * cdivf(v.cap, 16), 4);
    const int nchain = chain ? v.C * v.d * cdivf(v.cap, 16) : 0;
    const int tp = (v.cap <= 32 && v.ntiles >= 512) ? 2 : 1;      // tiles a workgroup stages per pass
    const size_t lds = std::max((size_t)tp * 16 * FXS, chain ? (size_t)4 * CHAIN_J * 256 : (size_t)0) * sizeof(double);
    // tile blocks: at most two per CU; each walks the tiles with that stride (one pass covers one or two tiles, see the kernel)
    const int ntb = std::max(1, std::min(v.ntiles, 512));
    hipLaunchKernelGGL(k_env_split, dim3(ntb + nsplit + nchain), dim3(256), lds, s, v, lid, going_left,
                       site, left_side, prev, prev_bond, out_bond, out, nsplit, ntb, tp);
}

}  // namespace mpst
