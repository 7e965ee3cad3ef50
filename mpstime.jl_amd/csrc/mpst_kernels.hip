// Hot-path kernels of the sweep engine (gfx950, fp64).
//
// Per bond (lid, rid = lid+1) the reference runs, one series at a time,
//   phi~_i = ps_i[lid] (x) LE_i (x) ps_i[rid] (x) RE_i,   yhat_i = <bt[:,c_i], phi~_i>,
//   grad[:,c] += phi~_i / yhat_i          (src/Training/loss_functions.jl:248-262,322-379)
// Here the same sums are three batched GEMMs over Khatri-Rao operands that are
// never materialised in HBM:
//   X_i[x] = LE_i[a] * ps_i[lid][s],   x = a*d + s      (dim X = chi_l * d)
//   Y_i[y] = ps_i[rid][s] * RE_i[b],   y = s*chi_r + b  (dim Y = d * chi_r)
//   yhat_i = X_i^T B_c Y_i,   G_c = sum_i w_i X_i Y_i^T,   env_new = Y E  /  X E.
// X/Y tiles are formed in LDS from coalesced rows of LE/RE and the (d) site
// vectors, and fed to v_mfma_f64_16x16x4_f64.
#include "mpst_internal.h"

namespace mpst {

// ---------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------
struct BondDims {
    int Dl, Dm, Dr, X, Y, L;
};
__device__ __forceinline__ BondDims bond_dims(const View& v, int lid) {
    BondDims b;
    b.Dl = v.chi[lid];
    b.Dm = v.chi[lid + 1];
    b.Dr = v.chi[lid + 2];
    b.X = b.Dl * v.d;
    b.Y = v.d * b.Dr;
    b.L = b.X * b.Y;
    return b;
}

// Deterministic block-wide sum (fixed tree), result broadcast to all threads.
// `red` must hold blockDim.x/64 doubles.
__device__ __forceinline__ double block_sum(double x, double* red) {
    x = wave_sum(x);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = x;
    __syncthreads();
    double t = 0.0;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}

// One wave computes a 16x16 tile  acc += sum_k A(m0+i, k) * B(k, n0+j)  with operands read
// straight from global/L2 (the matrices here are <= 128 KB and MFMA f64 issues once per 64 cycles,
// so operand delivery is not the limiter).
__device__ __forceinline__ d4 wave_gemm_tile(const double* __restrict__ A, int64_t sam, int64_t sak, int M,
                                             const double* __restrict__ B, int64_t sbk, int64_t sbn, int N,
                                             int K, int m0, int n0, d4 acc, int kbeg = 0, int kend = -1) {
    const int lane = threadIdx.x & 63;
    const int i = lane & 15, kq = lane >> 4;
    const int m = m0 + i, n = n0 + i;
    const bool mv = m < M, nv = n < N;
    const double* ap = A + (int64_t)m * sam;
    const double* bp = B + (int64_t)n * sbn;
    if (kend < 0) kend = K;
    // 16 k-steps (64 columns of K) per batch: 32 independent loads are in flight before the MFMAs, so a
    // K = 128 product costs two round trips to L2 (the callers are latency-, not throughput-bound)
    for (int k0 = kbeg; k0 < kend; k0 += 64) {
        double a[16], b[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int k = k0 + 4 * u + kq;
            const bool kv = k < kend;
            a[u] = (mv && kv) ? ap[(int64_t)k * sak] : 0.0;
            b[u] = (nv && kv) ? bp[(int64_t)k * sbk] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u)
            acc = mfma_f64(a[u], b[u], acc);        // (operands beyond the contraction are zero: an exact +0, no predicate)
    }
    return acc;
}

// Khatri-Rao tile of 16 series in LDS: Zs[i][z], z = a*d + s (left: prev_i[a]*phi_i[s]) or
// z = s*Dp + a (right: phi_i[s]*prev_i[a]); columns [Z, ZS) are zeroed.  256 threads: 16 per series.
// All global loads are issued before the first use (no per-element integer division).
__device__ __forceinline__ void stage_tile16(double* __restrict__ Zs, int ZS, const Span tl,
                                             const double* __restrict__ prev, int Dp, const double* __restrict__ ph,
                                             int d, int cap, bool left) {
    const int i = threadIdx.x >> 4, j = threadIdx.x & 15;
    const bool valid = i < tl.count;
    const int64_t smp = tl.start + (valid ? i : 0);
    double pa[8], pp[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int a = j + 16 * m;
        pa[m] = (valid && a < Dp) ? (prev ? prev[smp * cap + a] : 1.0) : 0.0;
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) pp[s] = (valid && s < d) ? ph[smp * d + s] : 0.0;
    double* row = Zs + i * ZS;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int a = j + 16 * m;
        if (a < Dp) {
#pragma unroll
            for (int s = 0; s < 8; ++s)
                if (s < d) row[left ? a * d + s : s * Dp + a] = pa[m] * pp[s];
            for (int s = 8; s < d; ++s)
                row[left ? a * d + s : s * Dp + a] = valid ? pa[m] * ph[smp * d + s] : 0.0;
        }
    }
    for (int z = Dp * d + j; z < ZS; z += 16) row[z] = 0.0;
}

// ---------------------------------------------------------------------------------------
// flatten_bt  (RealRealHighDimension.jl:221-238):  B_c = W[lid] * W[rid] per class
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void bt_assemble_block(const View& v, int lid, int bx, int c) {
    const BondDims b = bond_dims(v, lid);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tx = (b.X + 15) >> 4, ty = (b.Y + 15) >> 4;
    const int tile = bx * 4 + wave;
    if (tile >= tx * ty) return;
    const int m0 = (tile / ty) * 16, n0 = (tile % ty) * 16;
    const int ls = *v.label_site;
    const int cl = (ls == lid) ? c : 0, cr = (ls == lid + 1) ? c : 0;
    const double* Wl = v.sites + (int64_t)lid * v.site_stride + (int64_t)cl * b.X * b.Dm;
    const double* Wr = v.sites + (int64_t)(lid + 1) * v.site_stride + (int64_t)cr * b.Dm * b.Y;
    d4 acc = {0, 0, 0, 0};
    acc = wave_gemm_tile(Wl, b.Dm, 1, b.X, Wr, b.Y, 1, b.Y, b.Dm, m0, n0, acc);
    double* out = v.bt + (int64_t)c * b.L;
    const int col = n0 + (lane & 15);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = m0 + (lane >> 4) + 4 * r;
        if (row < b.X && col < b.Y) out[(int64_t)row * b.Y + col] = acc[r];
    }
}
__global__ __launch_bounds__(256) void k_bt_assemble(View v, int lid) { bt_assemble_block(v, lid, blockIdx.x, blockIdx.y); }
__global__ __launch_bounds__(256) void k_bt_assemble_b(const View* __restrict__ vs, int lid) {
    const View& v = vs[blockIdx.z];       // by reference: a local copy would live in per-lane scratch (the class tables are indexed dynamically)
    bt_assemble_block(v, lid, blockIdx.x, blockIdx.y);
}

// normalize!(BT_init) when rescale[1] (loss_functions.jl:109-111): one workgroup, C*L <= 2*16384.
__global__ __launch_bounds__(1024) void k_bt_prescale(View v, int lid) {
    __shared__ double red[16];
    const BondDims b = bond_dims(v, lid);
    const int n = v.C * b.L;
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 1024) s += v.bt[i] * v.bt[i];
    const double tot = block_sum(s, red);
    const double inv = 1.0 / sqrt(tot);
    for (int i = threadIdx.x; i < n; i += 1024) v.bt[i] *= inv;
}

// ---------------------------------------------------------------------------------------
// yhat_i = X_i^T B_c Y_i for a tile of 16 series  (+ per-tile loss terms)
//   KLD: own class only (loss_functions.jl:302-320);  MSE: every class (blockIdx.y) (:535-558)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_yhat(View v, int lid) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const BondDims b = bond_dims(v, lid);
    const int d = v.d, rid = lid + 1;
    const Span tl = v.tiles[blockIdx.x];
    const bool mse = v.loss == MPST_LOSS_MSE;
    const int c = mse ? (int)blockIdx.y : tl.cls;
    const int XP = (b.X + 3) & ~3, XS = XP + 2;
    const int YP = (b.Y + 15) & ~15, YS = YP + 2;
    double* Xs = smem;                 // [16][XS]
    double* Ys = Xs + 16 * XS;         // [16][YS]
    double* red = Ys + 16 * YS;        // [4][16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    const double* LEp = lid > 0 ? v.LE + (int64_t)(lid - 1) * v.N * v.cap : nullptr;
    const double* REn = rid < v.T - 1 ? v.RE + (int64_t)(rid + 1) * v.N * v.cap : nullptr;
    const double* phl = v.phi + (int64_t)lid * v.N * d;
    const double* phr = v.phi + (int64_t)rid * v.N * d;

    stage_tile16(Xs, XS, tl, LEp, b.Dl, phl, d, v.cap, true);
    stage_tile16(Ys, YS, tl, REn, b.Dr, phr, d, v.cap, false);
    __syncthreads();

    const double* Bc = v.bt + (int64_t)c * b.L;
    const int i16 = lane & 15, kq = lane >> 4;
    double p[4] = {0, 0, 0, 0};
    const int nty = YP >> 4;
    // Each wave owns the column tiles nt = wave and wave + 4 (d*chi <= 128: at most 8 tiles).  The whole B_c
    // column slice a lane needs for BOTH tiles (<= 2 x 32 k-steps) is requested from L2 up front - one round
    // trip - and the MFMAs then run back to back with the A operand from LDS.
    static_assert(MAX_DIM <= 128, "k_yhat keeps the B operand of two 16-column tiles (32 k-steps each) in registers");
    {
        double bv[2][32];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int col = (wave + 4 * h) * 16 + i16;
            const bool cv = wave + 4 * h < nty && col < b.Y;
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                const int kx = 4 * u + kq;
                bv[h][u] = (cv && kx < b.X) ? Bc[(int64_t)kx * b.Y + col] : 0.0;
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int nt = wave + 4 * h;
            if (nt < nty) {
                const int col = nt * 16 + i16;
                d4 acc = {0, 0, 0, 0};
#pragma unroll
                for (int u = 0; u < 32; ++u) {
                    const int k0 = 4 * u;
                    if (k0 < XP) acc = mfma_f64(Xs[i16 * XS + k0 + kq], bv[h][u], acc);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) p[r] += acc[r] * Ys[(kq + 4 * r) * YS + col];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const double x = sum16(p[r]);
        if (i16 == 0) red[wave * 16 + kq + 4 * r] = x;
    }
    __syncthreads();
    if (tid < 64) {
        double term = 0.0;
        if (tid < 16) {
            double yh = red[tid] + red[16 + tid] + red[32 + tid] + red[48 + tid];
            if (v.yhat_scaled) yh *= v.sc->inv_norm;           // track_cost: loss at the normalised bt_new (loss_functions.jl:177-184)
            if (tid < tl.count) {
                v.yhat[(int64_t)c * v.N + tl.start + tid] = yh;
                if (mse) {
                    const double m = (tl.cls == c) ? 1.0 : 0.0;
                    term = 0.5 * (yh - m) * (yh - m);          // MSE_iter! :554
                } else {
                    term = -log(yh * yh);                      // KLD_iter! :318
                }
            }
        }
        term = sum16(term);
        if (tid == 0) v.tile_loss[(int64_t)(mse ? c : 0) * v.ntiles + blockIdx.x] = term;
    }
}

// Same contraction for bond tensors beyond 128 x 128 (d*chi_max up to DIM_LIMIT).  A workgroup of 16 waves takes MT tiles
// of 16 series: every B_c fragment a wave pulls from L2 feeds 2*MT MFMAs (MT tiles x a pair of adjacent 16-column tiles),
// and the 16 waves split the (column pair, K range) space, so 4 waves per SIMD hide the L2 latency of the streamed
// panel.  Neither Khatri-Rao tile is staged: X_i[x] = LE_i[a] * phi_i[s] is formed from its factors in LDS on the way
// into the MFMA (one multiply per 2 MFMAs), Y_i[y] = phi_i[s] * RE_i[b] is applied to the accumulators.  The partial
// sums of the 16 waves meet in LDS in a fixed order.
template <int MT>
__global__ __launch_bounds__(1024) void k_yhat_gen(View v, int lid) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const BondDims b = bond_dims(v, lid);
    const int d = v.d, rid = lid + 1;
    const bool mse = v.loss == MPST_LOSS_MSE;
    const int t0 = blockIdx.x * MT;
    const int ntl = min(MT, v.ntiles - t0);
    const int LS = (b.Dl + 2) | 1, PS = d | 1, RS = (b.Dr + 1) | 1;
    double* Lf = smem;                      // [MT][16][LS]  left environment rows (zero beyond Dl)
    double* Pl = Lf + MT * 16 * LS;         // [MT][16][PS]  phi of the left site
    double* Pr = Pl + MT * 16 * PS;         // [MT][16][PS]  phi of the right site
    double* Rr = Pr + MT * 16 * PS;         // [MT][16][RS]  right environment rows
    double* red = Rr + MT * 16 * RS;        // [16 waves][MT][16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double* LEp = lid > 0 ? v.LE + (int64_t)(lid - 1) * v.N * v.cap : nullptr;
    const double* REn = rid < v.T - 1 ? v.RE + (int64_t)(rid + 1) * v.N * v.cap : nullptr;
    const double* phl = v.phi + (int64_t)lid * v.N * d;
    const double* phr = v.phi + (int64_t)rid * v.N * d;
    {
        const int m = tid >> 8, i = (tid >> 4) & 15, j = tid & 15;
        if (m < MT) {
            bool valid = false;
            int64_t smp = 0;
            if (m < ntl) {
                const Span tl = v.tiles[t0 + m];
                valid = i < tl.count;
                smp = tl.start + (valid ? i : 0);
            }
            double* lf = Lf + (m * 16 + i) * LS;
            for (int a = j; a < LS; a += 16) lf[a] = (valid && a < b.Dl) ? (LEp ? LEp[smp * v.cap + a] : 1.0) : 0.0;
            double* rr = Rr + (m * 16 + i) * RS;
            for (int a = j; a < b.Dr; a += 16) rr[a] = valid ? (REn ? REn[smp * v.cap + a] : 1.0) : 0.0;
            for (int s = j; s < d; s += 16) {
                Pl[(m * 16 + i) * PS + s] = valid ? phl[smp * d + s] : 0.0;
                Pr[(m * 16 + i) * PS + s] = valid ? phr[smp * d + s] : 0.0;
            }
        }
    }
    __syncthreads();
    const int i16 = lane & 15, kq = lane >> 4;
    const int XP = (b.X + 3) & ~3;
    const int nty = (b.Y + 15) >> 4, npair = (nty + 1) >> 1;
    const int KS = npair >= 16 ? 1 : 16 / npair;                   // K ranges per column pair: 16 wave tasks or more
    constexpr int UB = MT == 4 ? 4 : 8;                            // k-steps whose B fragments are in flight together (register budget: 128)
    const int Kc = ((XP + KS - 1) / KS + 4 * UB - 1) / (4 * UB) * (4 * UB);
    const int da = 4 / d, ds = 4 - da * d;
    double p[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) p[m][r] = 0.0;
    // KLD: the tiles of a group may belong to two classes at a class boundary; each run of equal class is one pass
    for (int rs = 0, re; rs < ntl; rs = re) {
        int c;
        if (mse) {
            c = blockIdx.y;
            re = ntl;
        } else {
            c = v.tiles[t0 + rs].cls;
            re = rs + 1;
            while (re < ntl && v.tiles[t0 + re].cls == c) ++re;
        }
        const double* Bc = v.bt + (int64_t)c * b.L;
        for (int task = wave; task < npair * KS; task += 16) {
            const int pr2 = task % npair, ks = task / npair;
            const int kbeg = ks * Kc, kend = min(XP, kbeg + Kc);
            if (kbeg >= kend) continue;
            const int col0 = (2 * pr2) * 16 + i16, col1 = col0 + 16;
            d4 acc[MT][2];
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m][0] = acc[m][1] = d4{0, 0, 0, 0};
            int ka = (kbeg + kq) / d, ksx = (kbeg + kq) - ka * d;     // x = ka * d + ksx: the A operand's factor indices
            for (int kb = kbeg; kb < kend; kb += 4 * UB) {
                double bv[2][UB];
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int kx = kb + 4 * u + kq;
                    const double* row = Bc + (int64_t)kx * b.Y;
                    bv[0][u] = (col0 < b.Y && kx < b.X) ? row[col0] : 0.0;
                    bv[1][u] = (col1 < b.Y && kx < b.X) ? row[col1] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    if (kb + 4 * u < kend) {
#pragma unroll
                        for (int m = 0; m < MT; ++m) {
                            const double am = Lf[(m * 16 + i16) * LS + ka] * Pl[(m * 16 + i16) * PS + ksx];
                            acc[m][0] = mfma_f64(am, bv[0][u], acc[m][0]);
                            acc[m][1] = mfma_f64(am, bv[1][u], acc[m][1]);
                        }
                    }
                    ksx += ds;
                    ka += da;
                    if (ksx >= d) {
                        ksx -= d;
                        ++ka;
                    }
                }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int col = h ? col1 : col0;
                if (col < b.Y) {
                    const int sy = col / b.Dr, bb = col - sy * b.Dr;
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        if (m >= rs && m < re) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int i = m * 16 + kq + 4 * r;
                                p[m][r] += acc[m][h][r] * (Pr[i * PS + sy] * Rr[i * RS + bb]);
                            }
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double x = sum16(p[m][r]);
            if (i16 == 0) red[(wave * MT + m) * 16 + kq + 4 * r] = x;
        }
    __syncthreads();
    if (tid < 64) {
        const int m = tid >> 4, i = tid & 15;
        double term = 0.0;
        int c = 0;
        if (m < ntl) {
            const Span tl = v.tiles[t0 + m];
            c = mse ? (int)blockIdx.y : tl.cls;
            double yh = 0.0;
            for (int w = 0; w < 16; ++w) yh += red[(w * MT + m) * 16 + i];
            if (v.yhat_scaled) yh *= v.sc->inv_norm;
            if (i < tl.count) {
                v.yhat[(int64_t)c * v.N + tl.start + i] = yh;
                if (mse) {
                    const double mm = (tl.cls == c) ? 1.0 : 0.0;
                    term = 0.5 * (yh - mm) * (yh - mm);
                } else {
                    term = -log(yh * yh);
                }
            }
        }
        term = sum16(term);
        if (i == 0 && m < ntl) v.tile_loss[(int64_t)(mse ? c : 0) * v.ntiles + t0 + m] = term;
    }
}

// ---------------------------------------------------------------------------------------
// gradient partials: P[c][j] (64x64 block) = sum_{i in the j-th share of class c's chunks} w_i X_i Y_i^T
//   KLD: w_i = 1/yhat_i over the series of class c      (loss_functions.jl:258, :367)
//   MSE: w_i = yhat_i^c - [label_i == c] over all series (:489, :608)
// A workgroup walks its share of the chunks with the accumulators in registers and writes ONE partial block, so the
// partial count (C * nsplit) and the workspace do not grow with N; the next chunk's operands are in flight (registers)
// while the MFMAs of the current one run out of LDS.
// ---------------------------------------------------------------------------------------
struct GradStage {
    double le[16], pl[16], re[16], pr[16];
};
__global__ __launch_bounds__(256, 2) void k_grad(View v, int lid, int nsplit) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    // [64 series][64 columns], unpadded (two workgroups per CU): odd rows hold their 16-column groups pairwise swapped, so
    // the rows k and k + 1 a half-wave reads together land on different halves of the banks
    constexpr int XS = GB;
    double* Xs = smem;
    double* Ys = smem + CHUNK_S * XS;
    const BondDims b = bond_dims(v, lid);
    const int d = v.d, rid = lid + 1;
    const int nby = (b.Y + GB - 1) / GB, nbx = (b.X + GB - 1) / GB;
    const int blk = blockIdx.y;
    if (blk >= nbx * nby) return;
    const int bx = blk / nby, by = blk - bx * nby;
    const bool mse = v.loss == MPST_LOSS_MSE;
    const int c = blockIdx.z;
    const int kc0 = mse ? 0 : v.cls_chunk_off[c], kc1 = mse ? v.nchunks : v.cls_chunk_off[c + 1];
    const int ka = kc0 + (int)(((int64_t)(kc1 - kc0) * blockIdx.x) / nsplit);
    const int kb = kc0 + (int)(((int64_t)(kc1 - kc0) * (blockIdx.x + 1)) / nsplit);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    const double* LEp = lid > 0 ? v.LE + (int64_t)(lid - 1) * v.N * v.cap : nullptr;
    const double* REn = rid < v.T - 1 ? v.RE + (int64_t)(rid + 1) * v.N * v.cap : nullptr;
    const double* phl = v.phi + (int64_t)lid * v.N * d;
    const double* phr = v.phi + (int64_t)rid * v.N * d;
    const double* yh = v.yhat + (int64_t)c * v.N;

    // index tables: one integer division per column of the block instead of one per element
    __shared__ int xa_[GB], xs_[GB], yb_[GB], ys_[GB];
    __shared__ double w_[2][CHUNK_S];    // per-series weight: one IEEE division per series, not one per thread
    if (tid < GB) {
        const int x = bx * GB + tid;
        const int a = x / d;
        xa_[tid] = x < b.X ? a : -1;
        xs_[tid] = x - a * d;
    } else if (tid < 2 * GB) {
        const int y = by * GB + tid - GB;
        const int sy = y / b.Dr;
        yb_[tid - GB] = y < b.Y ? y - sy * b.Dr : -1;
        ys_[tid - GB] = sy;
    }
    __syncthreads();
    // thread -> column xx = tid & 63 of the block, series i = (tid >> 6) + 4m
    const int xx = tid & 63, i0 = tid >> 6;
    const int a_ = xa_[xx], sx = xs_[xx], bb = yb_[xx], sy = ys_[xx];
    GradStage g;
    Span nxt = v.chunks[ka < kb ? ka : 0];   // the chunk descriptor is read one chunk ahead of its use
    int cnt_cur = 0, cls_cur = 0;
    double yv = 1.0;
    auto fetch = [&](int k) {
        const Span ch = nxt;
        cnt_cur = ch.count;
        cls_cur = ch.cls;
        if (k + 1 < kb) nxt = v.chunks[k + 1];
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const int i = i0 + 4 * m;
            const bool ok = i < ch.count;
            const int64_t smp = ch.start + (ok ? i : 0);
            g.le[m] = (ok && a_ >= 0) ? (LEp ? LEp[smp * v.cap + a_] : 1.0) : 0.0;
            g.pl[m] = (ok && a_ >= 0) ? phl[smp * d + sx] : 0.0;
            g.re[m] = (ok && bb >= 0) ? (REn ? REn[smp * v.cap + bb] : 1.0) : 0.0;
            g.pr[m] = (ok && bb >= 0) ? phr[smp * d + sy] : 0.0;
        }
        if (tid < CHUNK_S) yv = tid < ch.count ? yh[ch.start + tid] : 1.0;
    };
    // the weights of the fetched chunk: after the MFMAs, so that nobody waits on the yhat load before them
    auto publish_w = [&](int k) {
        if (tid < CHUNK_S) w_[k & 1][tid] = mse ? (yv - ((cls_cur == c) ? 1.0 : 0.0)) : 1.0 / yv;
    };
    const int i16 = lane & 15, kq = lane >> 4;
    const int mt = wave;  // 16 x-rows of the block per wave
    d4 acc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[nt] = d4{0, 0, 0, 0};
    const bool rows_live = bx * GB + mt * 16 < b.X;
    if (ka < kb) {
        fetch(ka);
        publish_w(ka);
    }
    for (int k = ka; k < kb; ++k) {
        const int kmax = (cnt_cur + 3) & ~3;
        __syncthreads();               // w_[k & 1] written; the previous chunk's MFMAs are done with Xs / Ys
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const int i = i0 + 4 * m;
            const int xw = xx ^ ((i & 1) << 4);
            Xs[i * XS + xw] = g.le[m] * g.pl[m];
            Ys[i * XS + xw] = w_[k & 1][i] * g.pr[m] * g.re[m];
        }
        __syncthreads();
        if (k + 1 < kb) fetch(k + 1);
        if (rows_live) {
            const int sw = (kq & 1) << 4;                       // row k0 + kq is odd iff kq is
            const double* xr = Xs + kq * XS + ((mt * 16) ^ sw) + i16;
            const double* yr = Ys + kq * XS + i16;
            if (kmax == CHUNK_S) {                              // full chunk: straight-line, LDS reads run ahead of the MFMAs
#pragma unroll
                for (int k0 = 0; k0 < CHUNK_S; k0 += 4) {
                    const double a = xr[k0 * XS];
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) acc[nt] = mfma_f64(a, yr[k0 * XS + ((nt * 16) ^ sw)], acc[nt]);
                }
            } else {
                for (int k0 = 0; k0 < kmax; k0 += 4) {
                    const double a = xr[k0 * XS];
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) acc[nt] = mfma_f64(a, yr[k0 * XS + ((nt * 16) ^ sw)], acc[nt]);
                }
            }
        }
        if (k + 1 < kb) publish_w(k + 1);
    }
    double* out = v.partial + ((int64_t)c * nsplit + blockIdx.x) * (int64_t)b.L;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int col = by * GB + nt * 16 + i16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = bx * GB + mt * 16 + kq + 4 * r;
            if (row < b.X && col < b.Y) out[(int64_t)row * b.Y + col] = acc[nt][r];
        }
    }
}

// grad[c] = scale_c * sum_chunks P ; loss = scaled sum of the tile terms.
// gradbuf = [loss, 0, grad[c][x][y] ...] is the buffer the multi-GPU all-reduce sums.
__global__ __launch_bounds__(256) void k_grad_reduce(View v, int lid, int nsplit) {
    __shared__ double red[4];
    const BondDims b = bond_dims(v, lid);
    const bool mse = v.loss == MPST_LOSS_MSE;
    const int64_t total = (int64_t)v.C * b.L;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx < total) {
        const int c = (int)(idx / b.L);
        const int64_t e = idx - (int64_t)c * b.L;
        // fixed association order: 8 interleaved partial sums, then a fixed tree
        const double* p = v.partial + (int64_t)c * nsplit * b.L + e;
        const double scale = mse ? v.invN                                       // :608
                                 : -(v.train_sep ? v.inv_count[c] : v.invN);    // :367 / :424
        double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int k = 0; k < nsplit; k += 16) {                    // 16 loads in flight; same association as 8 + 8
            double t[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) t[u] = (k + u < nsplit) ? p[(int64_t)(k + u) * b.L] : 0.0;
#pragma unroll
            for (int u = 0; u < 16; ++u) acc[u & 7] += t[u];
        }
        double s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
        s *= scale;
        v.gradbuf[2 + idx] = s;
    }
    if (blockIdx.x == 0) {
        double s = 0.0;
        if (mse) {
            for (int i = threadIdx.x; i < v.C * v.ntiles; i += 256) s += v.tile_loss[i];
            s *= v.invN;                                           // :612
        } else {
            for (int i = threadIdx.x; i < v.ntiles; i += 256) {
                const double w = v.train_sep ? v.inv_count[v.tiles[i].cls] : v.invN;   // :423 / :371
                s += v.tile_loss[i] * w;
            }
        }
        const double tot = block_sum(s, red);
        if (threadIdx.x == 0) {
            v.gradbuf[0] = tot;
            v.gradbuf[1] = 0.0;
        }
    }
}

// TSGO: bt -= eta * grad/||grad||  (loss_functions.jl:79);  GD: bt -= eta * grad (:49).
// ||grad|| comes from the pieces k_grad_norm wrote, summed by every workgroup in the same fixed order.
__global__ __launch_bounds__(256) void k_update(View v, int lid, int first_iter) {
    __shared__ double red[4];
    const BondDims b = bond_dims(v, lid);
    const int n = v.C * b.L;
    const double* g = v.gradbuf + 2;
    // ||grad||^2 from the pieces k_grad_norm wrote (64 entries each): every workgroup sums them in the same order, so
    // all see the same bits - and none re-reads the whole gradient (4 MB at d*chi = 512)
    double s = 0.0;
    for (int i = threadIdx.x; i < v.n_norm_part; i += 256) s += v.norm_part[i];
    const double nrm2 = block_sum(s, red);
    const double nrm = sqrt(nrm2);
    const double step = (v.optimiser == MPST_OPT_TSGO) ? v.eta / nrm : v.eta;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) v.bt[i] -= step * g[i];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (first_iter) {
            v.sc->loss = bond_loss(v);
            v.sc->grad_norm = nrm;
        }
        if (v.trace) v.trace[v.trace_it] = bond_loss(v);      // "Loss before step i", loss_functions.jl:50-52 / :80-82
    }
}

// track_cost: loss from the per-tile terms k_yhat just wrote (same weighting as k_grad_reduce), one workgroup
__global__ __launch_bounds__(256) void k_trace_loss(View v) {
    __shared__ double red[4];
    const bool mse = v.loss == MPST_LOSS_MSE;
    double s = 0.0;
    if (mse) {
        for (int i = threadIdx.x; i < v.C * v.ntiles; i += 256) s += v.tile_loss[i];
        s *= v.invN;
    } else {
        for (int i = threadIdx.x; i < v.ntiles; i += 256) s += v.tile_loss[i] * (v.train_sep ? v.inv_count[v.tiles[i].cls] : v.invN);
    }
    const double tot = block_sum(s, red);
    if (threadIdx.x == 0 && v.trace) v.trace[v.trace_it] = tot;
}

// ---------------------------------------------------------------------------------------
// Gram matrix of the bond tensor viewed as the matrix decomposeBT hands to svd
// (RealRealHighDimension.jl:166-169 / :185-188):
//   going left : rows (a, c, s_l), cols (s_r, b)  -> G = sum_c B_c^T B_c   (Y x Y)
//   going right: rows (b, c, s_r), cols (s_l, a)  -> G = sum_c B_c B_c^T   (X x X)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gram(View v, int lid, int going_left) {
    __shared__ double part[4][256];
    const BondDims b = bond_dims(v, lid);
    const int n = going_left ? b.Y : b.X;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tn = (n + 15) >> 4;
    const int tile = blockIdx.x;          // one 16x16 tile per workgroup, K split over its 4 waves
    if (tile >= tn * tn) return;
    const int m0 = (tile / tn) * 16, n0 = (tile % tn) * 16;
    d4 acc = {0, 0, 0, 0};
    if (going_left) {
        // k runs over (c, x) jointly: element (k, p) at k*Y + p
        const int K = v.C * b.X;
        const int kq4 = (((K + 3) >> 2) + 3) & ~3;
        acc = wave_gemm_tile(v.bt, 1, b.Y, n, v.bt, b.Y, 1, n, K, m0, n0, acc, wave * kq4, min(K, (wave + 1) * kq4));
    } else {
        const int K = b.Y;
        const int kq4 = (((K + 3) >> 2) + 3) & ~3;
        for (int c = 0; c < v.C; ++c) {
            const double* Bc = v.bt + (int64_t)c * b.L;
            acc = wave_gemm_tile(Bc, b.Y, 1, n, Bc, 1, b.Y, n, K, m0, n0, acc, wave * kq4, min(K, (wave + 1) * kq4));
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

// ---------------------------------------------------------------------------------------
// decomposeBT back-split (RealRealHighDimension.jl:172-176 / :190-194) from the kept
// eigenvectors E (dim x n_keep):
//   going left : W[rid] = E^T (right-orthonormal V),   W[lid][c] = B_c E * inv_norm  (= U*S, label)
//   going right: W[lid] = E   (left-orthonormal  U),   W[rid][c] = E^T B_c * inv_norm (= V*S, label)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_split(View v, int lid, int going_left) {
    const BondDims b = bond_dims(v, lid);
    const int nk = v.sc->n_keep;
    const int ldE = v.cap;
    const double inv = v.sc->inv_norm;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double* Wl = v.sites + (int64_t)lid * v.site_stride;
    double* Wr = v.sites + (int64_t)(lid + 1) * v.site_stride;
    const int tk = (nk + 15) >> 4;
    const int col = lane & 15, rq = lane >> 4;
    if (going_left) {
        const int tx = (b.X + 15) >> 4;
        const int ntile = v.C * tx * tk;
        for (int tile = blockIdx.x * 4 + wave; tile < ntile; tile += gridDim.x * 4) {
            const int c = tile / (tx * tk), rem = tile - c * tx * tk;
            const int m0 = (rem / tk) * 16, n0 = (rem % tk) * 16;
            const double* Bc = v.bt + (int64_t)c * b.L;
            d4 acc = {0, 0, 0, 0};
            acc = wave_gemm_tile(Bc, b.Y, 1, b.X, v.E, ldE, 1, nk, b.Y, m0, n0, acc);
            double* out = Wl + (int64_t)c * b.X * nk;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + rq + 4 * r;
                if (row < b.X && n0 + col < nk) out[(int64_t)row * nk + n0 + col] = acc[r] * inv;
            }
        }
        // W[rid][k][y] = E[y][k]
        for (int i = blockIdx.x * 256 + threadIdx.x; i < nk * b.Y; i += gridDim.x * 256) {
            const int k = i / b.Y, y = i - k * b.Y;
            Wr[i] = v.E[(int64_t)y * ldE + k];
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) *v.label_site = lid;
    } else {
        const int ty = (b.Y + 15) >> 4;
        const int ntile = v.C * tk * ty;
        for (int tile = blockIdx.x * 4 + wave; tile < ntile; tile += gridDim.x * 4) {
            const int c = tile / (tk * ty), rem = tile - c * tk * ty;
            const int m0 = (rem / ty) * 16, n0 = (rem % ty) * 16;
            const double* Bc = v.bt + (int64_t)c * b.L;
            d4 acc = {0, 0, 0, 0};
            // (E^T B_c)[k][y] : A(m=k, kk=x) = E[x*ldE + k], B(kk=x, n=y) = Bc[x*Y + y]
            acc = wave_gemm_tile(v.E, 1, ldE, nk, Bc, b.Y, 1, b.Y, b.X, m0, n0, acc);
            double* out = Wr + (int64_t)c * nk * b.Y;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + rq + 4 * r;
                if (row < nk && n0 + col < b.Y) out[(int64_t)row * b.Y + n0 + col] = acc[r] * inv;
            }
        }
        // W[lid][x][k] = E[x][k]
        for (int i = blockIdx.x * 256 + threadIdx.x; i < b.X * nk; i += gridDim.x * 256) {
            const int x = i / nk, k = i - x * nk;
            Wl[i] = v.E[(int64_t)x * ldE + k];
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) *v.label_site = lid + 1;
    }
}

// ---------------------------------------------------------------------------------------
// environment step (construct_caches :72-77,:90-98 and update_caches! :124-141):
//   out[i][k] = sum_z Z_i[z] * M[z*sz + k*sk]
// left side : Z_i[a*d+s] = prev_i[a] * phi_i[s]     (LE),  dimension Dp*d
// right side: Z_i[s*Dp+b] = phi_i[s] * prev_i[b]    (RE)
// prev == nullptr means the boundary (Dp = 1, value 1).
// ---------------------------------------------------------------------------------------
// Blocks [ntiles, gridDim.x) do not belong to the environment update: they assemble the bond tensor of
// the NEXT bond (bt_lid >= 0), which depends on the same inputs (the site tensors k_split just wrote) -
// one dependent kernel hand-over (~5 us) less per bond.
__global__ __launch_bounds__(256, 2) void k_env(View v, int site, int left_side, const double* __restrict__ prev,
                                             int prev_bond, int mode, int out_bond, double* __restrict__ out,
                                             int bt_lid, int bt_bx) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    if ((int)blockIdx.x >= v.ntiles) {
        const int idx = (int)blockIdx.x - v.ntiles;
        bt_assemble_block(v, bt_lid, idx % bt_bx, idx / bt_bx);
        return;
    }
    const int d = v.d;
    const int Dp = prev ? v.chi[prev_bond] : 1;
    const int Dout = v.chi[out_bond];
    const int Z = Dp * d;
    const double* __restrict__ M;
    int64_t sz, sk;
    if (mode == ENV_M_E) {
        M = v.E;
        sz = v.cap;
        sk = 1;
    } else if (mode == ENV_M_SITE) {
        M = v.sites + (int64_t)site * v.site_stride;   // [(a,s)][k], k = right bond
        sz = v.chi[site + 1];
        sk = 1;
    } else {
        M = v.sites + (int64_t)site * v.site_stride;   // [k][(s,b)], k = left bond
        sz = 1;
        sk = (int64_t)d * v.chi[site + 1];
    }
    const int ZP = (Z + 3) & ~3, ZS = ZP + 2;
    const Span tl = v.tiles[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double* ph = v.phi + (int64_t)site * v.N * d;
    stage_tile16(smem, ZS, tl, prev, Dp, ph, d, v.cap, left_side != 0);
    __syncthreads();
    const int i16 = lane & 15, kq = lane >> 4;
    const int nt_out = (Dout + 15) >> 4;
    const int nsteps = ZP >> 2, ks4 = env_ks4(nsteps);          // four chains over the quarters of the contraction (mpst_internal.h)
    for (int nt = wave; nt < nt_out; nt += 4) {
        const int col = nt * 16 + i16;
        const bool cv = col < Dout;
        d4 p[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) p[q] = d4{0, 0, 0, 0};
        for (int s0 = 0; s0 < ks4; s0 += 8) {
            double bv[4][8];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int step = q * ks4 + s0 + u, z = 4 * step + kq;
                    bv[q][u] = (cv && s0 + u < ks4 && z < Z) ? M[(int64_t)z * sz + (int64_t)col * sk] : 0.0;
                }
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int step = q * ks4 + s0 + u;
                    // (unconditional: beyond the contraction the B operand is zero and the A operand is read from the last live step - an
                    // exact +0; predicated MFMAs cost accumulator copies and a workgroup per CU)
                    p[q] = mfma_f64(smem[i16 * ZS + 4 * min(step, nsteps - 1) + kq], bv[q][u], p[q]);
                }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = kq + 4 * r;
            if (i < tl.count && cv) out[(int64_t)(tl.start + i) * v.cap + col] = (p[0][r] + p[1][r]) + (p[2][r] + p[3][r]);
        }
    }
}

// ---------------------------------------------------------------------------------------
// evaluation (src/summary.jl:4-136): yhat_i[c] = sum_{a,s,b} L_i[a] phi_i[s] R_i[b] W_p[c][a][s][b]
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_eval_final(View v, const double* __restrict__ Lc,
                                                    const double* __restrict__ Rc, double* __restrict__ yout) {
    const int p = *v.label_site;
    const int Dl = v.chi[p], Dr = v.chi[p + 1], d = v.d;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= v.N) return;
    const double* W = v.sites + (int64_t)p * v.site_stride;
    const double* ph = v.phi + ((int64_t)p * v.N + i) * d;
    const double* L = (p > 0) ? Lc + i * v.cap : nullptr;
    const double* R = (p < v.T - 1) ? Rc + i * v.cap : nullptr;
    for (int c = 0; c < v.C; ++c) {
        const double* Wc = W + (int64_t)c * Dl * d * Dr;
        double acc = 0.0;
        for (int a = 0; a < Dl; ++a) {
            const double la = L ? L[a] : 1.0;
            for (int s = 0; s < d; ++s) {
                const double ls = la * ph[s];
                const double* w = Wc + ((int64_t)a * d + s) * Dr;
                double t = 0.0;
                for (int bb = 0; bb < Dr; ++bb) t += w[bb] * (R ? R[bb] : 1.0);
                acc += ls * t;
            }
        }
        yout[i * v.C + c] = acc;
    }
}

// MSE_loss_acc_iter (summary.jl:33-58): one workgroup, fixed-order sums.
// out3 = {mean mse, mean kld, accuracy}; conf[truth][pred]; pred[i] = argmax |yhat| (first max).
__global__ __launch_bounds__(1024) void k_eval_reduce(View v, const double* __restrict__ yin, double* out3,
                                                      int64_t* conf, int32_t* pred) {
    __shared__ double red[16];
    __shared__ int cm[MAX_C * MAX_C];
    const int C = v.C;
    for (int i = threadIdx.x; i < C * C; i += 1024) cm[i] = 0;
    __syncthreads();
    double mse = 0.0, kld = 0.0, acc = 0.0;
    for (int64_t i = threadIdx.x; i < v.N; i += 1024) {
        const double* y = yin + i * C;
        const int lab = v.label[i];
        double s = 0.0, best = -1.0;
        int arg = 0;
        for (int c = 0; c < C; ++c) {
            const double t = y[c] - (c == lab ? 1.0 : 0.0);
            s += t * t;
            const double ab = fabs(y[c]);
            if (ab > best) {
                best = ab;
                arg = c;
            }
        }
        mse += 0.5 * s;
        kld += -log(y[lab] * y[lab]);
        acc += (arg == lab) ? 1.0 : 0.0;
        if (pred) pred[i] = arg;
        atomicAdd(&cm[lab * C + arg], 1);
    }
    const double tm = block_sum(mse, red);
    const double tk = block_sum(kld, red);
    const double ta = block_sum(acc, red);
    __syncthreads();
    if (threadIdx.x == 0) {
        out3[0] = tm;
        out3[1] = tk;
        out3[2] = ta;
    }
    if (conf)
        for (int i = threadIdx.x; i < C * C; i += 1024) conf[i] = cm[i];
}

// ---------------------------------------------------------------------------------------
// normalize!(W) (RealRealHighDimension.jl:852; ITensors: every site / exp(lognorm/T)).
// <W|W> by transfer matrices in one workgroup: E <- sum_{s(,c)} A_s^T E A_s.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_norm2(View v, double* out_norm2, double* gscratch) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int cm = v.cap;
    // three cm x cm matrices: in LDS while they fit, else in the global scratch the caller provides (a single workgroup:
    // its own global writes are visible to it after a barrier)
    double* E = gscratch ? gscratch : smem;   // [cm][cm]
    double* Tm = E + cm * cm;     // [cm][cm]  T = E * A_s
    double* En = Tm + cm * cm;    // [cm][cm]
    const int tid = threadIdx.x;
    const int ls = *v.label_site;
    for (int i = tid; i < cm * cm; i += 1024) E[i] = 0.0;
    __syncthreads();
    if (tid == 0) E[0] = 1.0;
    __syncthreads();
    for (int j = 0; j < v.T; ++j) {
        const int Dl = v.chi[j], Dr = v.chi[j + 1], d = v.d;
        const int Cj = (j == ls) ? v.C : 1;
        const double* W = v.sites + (int64_t)j * v.site_stride;
        for (int i = tid; i < Dr * Dr; i += 1024) En[(i / Dr) * cm + (i % Dr)] = 0.0;
        __syncthreads();
        for (int c = 0; c < Cj; ++c)
            for (int s = 0; s < d; ++s) {
                const double* A = W + (int64_t)c * Dl * d * Dr + (int64_t)s * Dr;  // A[a][r] at a*d*Dr + r
                for (int i = tid; i < Dl * Dr; i += 1024) {
                    const int a = i / Dr, r = i - a * Dr;
                    double t = 0.0;
                    for (int a2 = 0; a2 < Dl; ++a2) t += E[a * cm + a2] * A[(int64_t)a2 * d * Dr + r];
                    Tm[a * cm + r] = t;
                }
                __syncthreads();
                for (int i = tid; i < Dr * Dr; i += 1024) {
                    const int r1 = i / Dr, r2 = i - r1 * Dr;
                    double t = 0.0;
                    for (int a = 0; a < Dl; ++a) t += A[(int64_t)a * d * Dr + r1] * Tm[a * cm + r2];
                    En[r1 * cm + r2] += t;
                }
                __syncthreads();
            }
        for (int i = tid; i < Dr * Dr; i += 1024) {
            const int r1 = i / Dr, r2 = i - r1 * Dr;
            E[r1 * cm + r2] = En[r1 * cm + r2];
        }
        __syncthreads();
    }
    if (tid == 0) *out_norm2 = E[0];
}

__global__ __launch_bounds__(256) void k_scale_sites(View v, const double* norm2) {
    const int j = blockIdx.x;
    const int ls = *v.label_site;
    const int n = ((j == ls) ? v.C : 1) * v.chi[j] * v.d * v.chi[j + 1];
    const double z = exp(0.5 * log(*norm2) / (double)v.T);
    double* W = v.sites + (int64_t)j * v.site_stride;
    for (int i = threadIdx.x; i < n; i += 256) W[i] /= z;
}

__global__ void k_selftest_mfma(const double* A, const double* B, int K, double* C) {
    const int lane = threadIdx.x & 63;
    d4 acc = {0, 0, 0, 0};
    acc = wave_gemm_tile(A, K, 1, 16, B, 16, 1, 16, K, 0, 0, acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) C[((lane >> 4) + 4 * r) * 16 + (lane & 15)] = acc[r];
}

// ---------------------------------------------------------------------------------------
// launchers.  Grids are sized for the maximum bond dimension; kernels read the live
// dimensions from device memory and idle the surplus workgroups, so no launch depends on
// a host read-back and a whole sweep can be enqueued (or graph-captured) at once.
// ---------------------------------------------------------------------------------------
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

void launch_bt_assemble(const View& v, int lid, hipStream_t s) {
    const int dm = v.d * v.cap;
    const int tiles = cdiv(dm, 16) * cdiv(dm, 16);
    hipLaunchKernelGGL(k_bt_assemble, dim3(cdiv(tiles, 4), v.C), dim3(256), 0, s, v, lid);
}
void launch_bt_assemble_b(const View& v, const View* vs, int K, int lid, hipStream_t s) {
    const int dm = v.d * v.cap;
    const int tiles = cdiv(dm, 16) * cdiv(dm, 16);
    hipLaunchKernelGGL(k_bt_assemble_b, dim3(cdiv(tiles, 4), v.C, K), dim3(256), 0, s, vs, lid);
}
void launch_bt_prescale(const View& v, int lid, hipStream_t s) {
    hipLaunchKernelGGL(k_bt_prescale, dim3(1), dim3(1024), 0, s, v, lid);
}
void launch_yhat(const View& v, int lid, hipStream_t s) {
    const int dm = v.d * v.cap;
    const int gy = v.loss == MPST_LOSS_MSE ? v.C : 1;
    if (dm <= MAX_DIM) {
        const size_t lds = (size_t)(16 * (((dm + 3) & ~3) + 2) + 16 * (((dm + 15) & ~15) + 2) + 64) * sizeof(double);
        hipLaunchKernelGGL(k_yhat, dim3(v.ntiles, gy), dim3(256), lds, s, v, lid);
    } else {
        // tiles per workgroup: as many as still leave a workgroup per CU
        const int mt = (v.ntiles >= 1024 && v.cap <= 64) ? 4 : (v.ntiles >= 512 ? 2 : 1);
        const size_t lds = (size_t)(mt * 16 * (((v.cap + 2) | 1) + 2 * (v.d | 1) + ((v.cap + 1) | 1)) + 16 * mt * 16) * sizeof(double);
        const dim3 grid(cdiv(v.ntiles, mt), gy);
        if (mt == 4) hipLaunchKernelGGL(k_yhat_gen<4>, grid, dim3(1024), lds, s, v, lid);
        else if (mt == 2) hipLaunchKernelGGL(k_yhat_gen<2>, grid, dim3(1024), lds, s, v, lid);
        else hipLaunchKernelGGL(k_yhat_gen<1>, grid, dim3(1024), lds, s, v, lid);
    }
}
void launch_grad(const View& v, int lid, hipStream_t s) {
    const int dm = v.d * v.cap;
    const int nb = cdiv(dm, GB) * cdiv(dm, GB);
    const size_t lds = (size_t)2 * CHUNK_S * GB * sizeof(double);
    const int nsplit = grad_nsplit(v.nchunks, nb, v.C);
    hipLaunchKernelGGL(k_grad, dim3(nsplit, nb, v.C), dim3(256), lds, s, v, lid, nsplit);
}
void launch_grad_reduce(const View& v, int lid, hipStream_t s) {
    const int dm = v.d * v.cap;
    const int64_t tot = (int64_t)v.C * dm * dm;
    const int nsplit = grad_nsplit(v.nchunks, cdiv(dm, GB) * cdiv(dm, GB), v.C);
    hipLaunchKernelGGL(k_grad_reduce, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, v, lid, nsplit);
}
void launch_update(const View& v, int lid, int first_iter, hipStream_t s) {
    launch_grad_norm(v, lid, s);
    const int64_t tot = (int64_t)v.C * v.d * v.cap * v.d * v.cap;
    const int grid = (int)std::min<int64_t>(512, std::max<int64_t>(32, (tot + 1023) / 1024));
    hipLaunchKernelGGL(k_update, dim3(grid), dim3(256), 0, s, v, lid, first_iter);
}
void launch_trace_loss(const View& v, hipStream_t s) { hipLaunchKernelGGL(k_trace_loss, dim3(1), dim3(256), 0, s, v); }
void launch_gram(const View& v, int lid, int going_left, hipStream_t s) {
    const int dm = v.d * v.cap;
    const int tiles = cdiv(dm, 16) * cdiv(dm, 16);
    hipLaunchKernelGGL(k_gram, dim3(tiles), dim3(256), 0, s, v, lid, going_left);
}
void launch_split(const View& v, int lid, int going_left, hipStream_t s) {
    const int dm = v.d * v.cap;
    const int tiles = v.C * cdiv(dm, 16) * cdiv(v.cap, 16);
    hipLaunchKernelGGL(k_split, dim3(cdiv(tiles, 4)), dim3(256), 0, s, v, lid, going_left);
}
void launch_env(const View& v, int site, int left_side, const double* prev, int prev_bond, int mode,
                int out_bond, double* out, hipStream_t s, int bt_lid) {
    const int dm = v.d * v.cap;
    const size_t lds = (size_t)16 * (((dm + 3) & ~3) + 2) * sizeof(double);
    const int bt_bx = cdiv(cdiv(dm, 16) * cdiv(dm, 16), 4);
    const int extra = bt_lid >= 0 ? bt_bx * v.C : 0;
    hipLaunchKernelGGL(k_env, dim3(v.ntiles + extra), dim3(256), lds, s, v, site, left_side, prev, prev_bond, mode,
                       out_bond, out, bt_lid, bt_bx);
}
hipError_t init_kernel_attrs(int device) {
    static std::atomic<unsigned long long> done{0};      // one bit per device: the attribute is per device and per process
    if (device >= 0 && device < 64 && (done.load(std::memory_order_acquire) >> device) & 1ull) return hipSuccess;
    hipError_t e;
    if ((e = hipFuncSetAttribute((const void*)k_grad, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)(2 * CHUNK_S * GB * sizeof(double)))) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)k_yhat, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)k_env, hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)k_yhat_gen<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)k_yhat_gen<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)k_yhat_gen<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)k_norm2, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024)) != hipSuccess) return e;
    if (device >= 0 && device < 64) done.fetch_or(1ull << device, std::memory_order_release);
    return hipSuccess;
}
void launch_eval_final(const View& v, const double* Lc, const double* Rc, double* yout, hipStream_t s) {
    hipLaunchKernelGGL(k_eval_final, dim3((unsigned)((v.N + 255) / 256)), dim3(256), 0, s, v, Lc, Rc, yout);
}
void launch_eval_reduce(const View& v, const double* yin, double* out3, int64_t* conf, int32_t* pred,
                        hipStream_t s) {
    hipLaunchKernelGGL(k_eval_reduce, dim3(1), dim3(1024), 0, s, v, yin, out3, conf, pred);
}
void launch_norm2(const View& v, double* out_norm2, double* gscratch, hipStream_t s) {
    const size_t lds = (size_t)3 * v.cap * v.cap * sizeof(double);
    const bool fits = lds <= 128 * 1024;
    hipLaunchKernelGGL(k_norm2, dim3(1), dim3(1024), fits ? lds : 0, s, v, out_norm2, fits ? (double*)nullptr : gscratch);
}
void launch_scale_sites(const View& v, const double* norm2, hipStream_t s) {
    hipLaunchKernelGGL(k_scale_sites, dim3(v.T), dim3(256), 0, s, v, norm2);
}
void launch_selftest_mfma(const double* A, const double* B, int K, double* C, hipStream_t s) {
    hipLaunchKernelGGL(k_selftest_mfma, dim3(1), dim3(64), 0, s, A, B, K, C);
}

}  // namespace mpst
