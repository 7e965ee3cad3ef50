// Element-typed sweep: the same bond update as mpst_kernels.hip / mpst_fused.hip for MPSs and encodings that are not real
// fp64 - real fp32 (opts.dtype = Float32), complex fp64 / fp32 (the reference's Fourier, Sahand and Stoudenmire bases).
//
// Semantics.  The reference trains complex encodings only through its legacy ITensor engine (the array engine raises,
// src/Training/RealRealHighDimension.jl:461-466), so the formulas are those of
//   src/legacy_itensor/loss_functions.jl:433-462   phi~ = conj(ps_l) (x) LE (x) conj(ps_r) (x) RE, yhat = BT * phi~ (no conj on BT)
//   src/legacy_itensor/loss_functions.jl:468-596   KLD: loss = -log|yhat|^2, grad[c] = -conj(sum_i phi~_i / yhat_i) / N
//   src/legacy_itensor/loss_functions.jl:599-640   MSE: 0.5 sum_c |yhat_c - y_c|^2, grad = (yhat - y) conj(phi~) / N
//   src/legacy_itensor/RealRealLegacyITensor.jl:2-142   caches (conj on the product states only), decomposeBT_IT
// and they reduce to the array engine's (src/Training/loss_functions.jl:193-619) for real element types.
//
// As in the fp64 path the sample loop is three GEMMs over Khatri-Rao operands that never exist in HBM:
//   X_i[(a,s)] = LE_i[a] conj(ps_i[lid][s]),  Y_i[(s,b)] = conj(ps_i[rid][s]) RE_i[b]
//   yhat_i = X_i^T B_c Y_i                                        k_tyhat   (MFMA in the element's precision)
//   G_c = sum_i u_i conj(X_i) conj(Y_i)^T, u_i = conj(1/yhat_i) | yhat_i - delta     k_tgrad
//   env_new = Y conj(E) | X conj(E)                              k_tenv    (the new site tensor IS conj(E) / E^T)
// Complex numbers are interleaved (re, im) pairs everywhere; a complex product is four real MFMA chains (v_mfma_f32_16x16x4_f32
// / v_mfma_f64_16x16x4_f64).  What decides the truncation stays in fp64 whatever the element type: yhat and the loss, the
// reduced gradient and its norm, the optimiser step, the Gram matrix A^H A (complex: its real embedding, see View::zw),
// the eigensolver (mpst_eig*.hip) and the back-split A E - an fp32 Gram matrix cannot resolve a relative cutoff of 1e-10.
#include "mpst_internal.h"
#include <algorithm>
#include <type_traits>

namespace mpst {

namespace {

template <typename R> struct Cp {
    R re, im;
};
template <typename R, bool CX> struct Et {
    using vec2 = typename Mx<R>::vec2;
    using type = std::conditional_t<CX, vec2, R>;
};
template <typename R, bool CX> __device__ __forceinline__ Cp<R> eld(const void* __restrict__ p, int64_t e) {
    Cp<R> o;
    if constexpr (CX) {
        const typename Mx<R>::vec2 t = reinterpret_cast<const typename Mx<R>::vec2*>(p)[e];
        o.re = t.x;
        o.im = t.y;
    } else {
        o.re = reinterpret_cast<const R*>(p)[e];
        o.im = R(0);
    }
    return o;
}
template <typename R, bool CX> __device__ __forceinline__ void est(void* __restrict__ p, int64_t e, Cp<R> v) {
    if constexpr (CX) {
        typename Mx<R>::vec2 t;
        t.x = v.re;
        t.y = v.im;
        reinterpret_cast<typename Mx<R>::vec2*>(p)[e] = t;
    } else {
        reinterpret_cast<R*>(p)[e] = v.re;
    }
}
template <typename R> __device__ __forceinline__ Cp<R> cmul(Cp<R> a, Cp<R> b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
template <typename R> __device__ __forceinline__ Cp<R> cmulc(Cp<R> a, Cp<R> b) { return {a.re * b.re + a.im * b.im, a.im * b.re - a.re * b.im}; }   // a conj(b)
template <typename R> __device__ __forceinline__ Cp<double> todbl(Cp<R> a) { return {(double)a.re, (double)a.im}; }

struct TDims {
    int Dl, Dm, Dr, X, Y, L;
};
__device__ __forceinline__ TDims tdims(const TView& v, int lid) {
    TDims b;
    b.Dl = v.chi[lid];
    b.Dm = v.chi[lid + 1];
    b.Dr = v.chi[lid + 2];
    b.X = b.Dl * v.d;
    b.Y = v.d * b.Dr;
    b.L = b.X * b.Y;
    return b;
}

__device__ __forceinline__ double tblock_sum(double x, double* red) {      // fixed tree, result in every thread
    x = wave_sum(x);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = x;
    __syncthreads();
    double t = 0.0;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}

// One MFMA step of a (complex) product on 16 x 16 tiles: acc += a * b.
template <typename R, bool CX>
__device__ __forceinline__ void mstep(Cp<R> a, Cp<R> b, typename Mx<R>::acc_t& ar, typename Mx<R>::acc_t& ai) {
    ar = Mx<R>::mma(a.re, b.re, ar);
    if constexpr (CX) {
        ar = Mx<R>::mma(-a.im, b.im, ar);
        ai = Mx<R>::mma(a.re, b.im, ai);
        ai = Mx<R>::mma(a.im, b.re, ai);
    }
}

// The same step for k_tyhat in the 3M form (k_tgrad is bound by forming its operand tiles, 3M bought it nothing): three real MFMAs
// per complex step instead of four,
//   K1 += (a.re + a.im) b.re,  K2 += a.re (b.im - b.re),  K3 += a.im (b.re + b.im);   re = K1 - K3,  im = K1 + K2
// (the derived operands are three VALU adds beside three 32-cycle MFMAs; the accumulators are combined once, after the k-loops).
template <typename R, bool CX>
__device__ __forceinline__ void mstep3(Cp<R> a, Cp<R> b, typename Mx<R>::acc_t& k1, typename Mx<R>::acc_t& k2, typename Mx<R>::acc_t& k3) {
    if constexpr (CX) {
        k1 = Mx<R>::mma(a.re + a.im, b.re, k1);
        k2 = Mx<R>::mma(a.re, b.im - b.re, k2);
        k3 = Mx<R>::mma(a.im, b.re + b.im, k3);
    } else {
        k1 = Mx<R>::mma(a.re, b.re, k1);
    }
}
template <typename A> __device__ __forceinline__ void combine3(A& k1_re, A& k2_im, const A& k3) {
    const A k1 = k1_re;
    k1_re = k1 - k3;
    k2_im = k1 + k2_im;
}

// 16 x 16 tile of a product whose operands come from global memory through loaders (lane-local lambdas: fa(k) = A[m0 + i16][k],
// fb(k) = B[k][n0 + i16], zero outside the matrix), accumulated on the fp64 MFMA whatever the storage type.
template <bool CX, typename FA, typename FB>
__device__ __forceinline__ void wave_tile64(FA fa, FB fb, int kbeg, int kend, d4& accR, d4& accI) {
    const int kq = (threadIdx.x & 63) >> 4;
    for (int k0 = kbeg; k0 < kend; k0 += 32) {
        Cp<double> a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = k0 + 4 * u + kq;
            const bool kv = k < kend;
            a[u] = kv ? fa(k) : Cp<double>{0.0, 0.0};
            b[u] = kv ? fb(k) : Cp<double>{0.0, 0.0};
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (k0 + 4 * u < kend) mstep<double, CX>(a[u], b[u], accR, accI);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// flatten_bt: B_c = W[lid] W[rid]  (RealRealHighDimension.jl:221-238)
// ---------------------------------------------------------------------------------------------------------------------------
template <typename R, bool CX>
__global__ __launch_bounds__(256) void k_tbt_assemble(TView v, int lid) {
    const TDims b = tdims(v, lid);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i16 = lane & 15, kq = lane >> 4;
    const int ty = (b.Y + 15) >> 4, tx = (b.X + 15) >> 4;
    const int tile = blockIdx.x * 4 + wave, c = blockIdx.y;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) v.sc->xref = INT32_MIN;   // first launch of a bond's chain
    if (tile >= tx * ty) return;
    const int m0 = (tile / ty) * 16, n0 = (tile % ty) * 16;
    const int ls = *v.label_site;
    const int cl = (ls == lid) ? c : 0, cr = (ls == lid + 1) ? c : 0;
    const int64_t ol = (int64_t)lid * v.site_stride + (int64_t)cl * b.X * b.Dm, orr = (int64_t)(lid + 1) * v.site_stride + (int64_t)cr * b.Dm * b.Y;
    const int m = m0 + i16, n = n0 + i16;
    d4 aR = {0, 0, 0, 0}, aI = {0, 0, 0, 0};
    wave_tile64<CX>([&](int k) { return m < b.X ? todbl(eld<R, CX>(v.sites, ol + (int64_t)m * b.Dm + k)) : Cp<double>{0, 0}; },
                    [&](int k) { return n < b.Y ? todbl(eld<R, CX>(v.sites, orr + (int64_t)k * b.Y + n)) : Cp<double>{0, 0}; }, 0, b.Dm, aR, aI);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = m0 + kq + 4 * r;
        if (row < b.X && n < b.Y) est<R, CX>(v.bt, (int64_t)c * b.L + (int64_t)row * b.Y + n, Cp<R>{(R)aR[r], (R)aI[r]});
    }
}

// normalize!(BT_init) when rescale[1] (loss_functions.jl:109-111): one workgroup
template <typename R, bool CX>
__global__ __launch_bounds__(1024) void k_tbt_prescale(TView v, int lid) {
    __shared__ double red[16];
    const TDims b = tdims(v, lid);
    const int n = v.C * b.L * (CX ? 2 : 1);
    R* p = (R*)v.bt;
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 1024) s += (double)p[i] * (double)p[i];
    const double inv = 1.0 / sqrt(tblock_sum(s, red));
    for (int i = threadIdx.x; i < n; i += 1024) p[i] = (R)((double)p[i] * inv);
}

// ---------------------------------------------------------------------------------------------------------------------------
// environment step (construct_caches / update_caches!): out_i[k] = sum_z Z_i[z] M[z][k],
//   left : Z_i[a*d + s]  = prev_i[a] conj(phi_i[s]),  right: Z_i[s*Dp + b] = conj(phi_i[s]) prev_i[b]
//   mode ENV_M_SITE: M[(a,s)][k] = W[site][a][s][k];  ENV_M_SITE_T: M[(s,b)][k] = W[site][k][s][b].
// One wave per tile of 16 series (wave-local LDS, no workgroup barrier); the Khatri-Rao operand is formed from its factors on
// the way into the MFMA; NT column tiles share every A fragment.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int TENV_NT = 4, TENV_G = 2;        // column tiles per group, groups held in registers (cap <= 128: 8 tiles)
// Every environment row leaves this kernel scaled by a power of two so that its largest component lies in [0.5, 1), the
// exponent accumulating in a per-series int32 (xout = xprev + e): products of T site contractions shrink like d^(-T/2)
// (1e-95 after 200 Fourier sites) and leave the range of fp32 after ~80 sites, of fp64 after ~600.  Powers of two are exact,
// so nothing else changes; the KLD gradient phi~/yhat does not see the scale at all, the loss gets -2 (xL + xR) ln 2 back.
template <typename R, bool CX>
__global__ __launch_bounds__(256) void k_tenv(TView v, int site, int left_side, const void* __restrict__ prev, const int32_t* __restrict__ xprev,
                                               int prev_bond, int mode, int out_bond, void* __restrict__ out, int32_t* __restrict__ xout) {
    using E = typename Et<R, CX>::type;
    using acc_t = typename Mx<R>::acc_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int d = v.d;
    const int Dp = prev ? v.chi[prev_bond] : 1;
    const int Dout = v.chi[out_bond];
    const int Z = Dp * d;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i16 = lane & 15, kq = lane >> 4;
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= v.ntiles) return;
    const Span tl = v.tiles[tile];
    const int PS = (v.cap + 1) | 1, DS = d | 1;
    E* Pv = reinterpret_cast<E*>(smem_raw) + (size_t)wave * 16 * (PS + DS);      // [16][PS] previous environment rows
    E* Ph = Pv + 16 * PS;                                                        // [16][DS] product states of `site`
    for (int i = 0; i < 16; ++i) {
        const bool ok = i < tl.count;
        const int64_t smp = tl.start + (ok ? i : 0);
        for (int a = lane; a < Dp; a += 64) {
            Cp<R> x = prev ? eld<R, CX>(prev, smp * v.cap + a) : Cp<R>{R(1), R(0)};
            if (!ok) x = Cp<R>{R(0), R(0)};
            est<R, CX>(Pv, i * PS + a, x);
        }
        if (lane < d) {
            Cp<R> x = eld<R, CX>(v.phi, ((int64_t)site * v.N + smp) * d + lane);
            if (!ok) x = Cp<R>{R(0), R(0)};
            est<R, CX>(Ph, i * DS + lane, x);
        }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    int64_t sz, sk;
    const int64_t so = (int64_t)site * v.site_stride;
    if (mode == ENV_M_SITE) {
        sz = v.chi[site + 1];
        sk = 1;
    } else {
        sz = 1;
        sk = (int64_t)d * v.chi[site + 1];
    }
    const int LO = left_side ? d : Dp;           // z = hi * LO + lo
    const int dhi = 4 / LO, dlo = 4 - dhi * LO;
    const int ZP = (Z + 3) & ~3;
    const int ntout = (Dout + 15) >> 4;
    acc_t aR[TENV_G][TENV_NT], aI[TENV_G][TENV_NT];
#pragma unroll
    for (int g = 0; g < TENV_G; ++g)
#pragma unroll
        for (int t = 0; t < TENV_NT; ++t) aR[g][t] = aI[g][t] = acc_t{0, 0, 0, 0};
#pragma unroll
    for (int g = 0; g < TENV_G; ++g) {
        const int g0 = g * TENV_NT;
        if (g0 >= ntout) break;
        int hi = kq / LO, lo = kq - hi * LO;
        constexpr int UB = 4;
        for (int kb = 0; kb < ZP; kb += 4 * UB) {
            Cp<R> bv[TENV_NT][UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int z = kb + 4 * u + kq;
#pragma unroll
                for (int t = 0; t < TENV_NT; ++t) {
                    const int col = (g0 + t) * 16 + i16;
                    bv[t][u] = (z < Z && col < Dout) ? eld<R, CX>(v.sites, so + (int64_t)z * sz + (int64_t)col * sk) : Cp<R>{R(0), R(0)};
                }
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                if (kb + 4 * u < ZP) {
                    const int z = kb + 4 * u + kq;
                    const int ia = left_side ? hi : lo, is = left_side ? lo : hi;
                    Cp<R> a = {R(0), R(0)};
                    if (z < Z) a = cmulc(eld<R, CX>(Pv, i16 * PS + ia), eld<R, CX>(Ph, i16 * DS + is));
#pragma unroll
                    for (int t = 0; t < TENV_NT; ++t) mstep<R, CX>(a, bv[t][u], aR[g][t], aI[g][t]);
                }
                lo += dlo;
                hi += dhi;
                if (lo >= LO) {
                    lo -= LO;
                    ++hi;
                }
            }
        }
    }
    // largest component of every row (16 lanes x all tiles), its binary exponent, the rescale, the store
    int er[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        double m = 0.0;
#pragma unroll
        for (int g = 0; g < TENV_G; ++g)
#pragma unroll
            for (int t = 0; t < TENV_NT; ++t) m = fmax(m, fmax(fabs((double)aR[g][t][r]), fabs((double)aI[g][t][r])));
        m = max16(m);
        er[r] = (m > 0.0 && m < 1e300) ? __builtin_amdgcn_frexp_exp(m) : 0;
    }
#pragma unroll
    for (int g = 0; g < TENV_G; ++g) {
        if (g * TENV_NT >= ntout) break;
#pragma unroll
        for (int t = 0; t < TENV_NT; ++t) {
            const int col = (g * TENV_NT + t) * 16 + i16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = Mx<R>::row(kq, r);
                if (i < tl.count && col < Dout)
                    est<R, CX>(out, (int64_t)(tl.start + i) * v.cap + col,
                               Cp<R>{(R)__builtin_amdgcn_ldexp((double)aR[g][t][r], -er[r]), (R)__builtin_amdgcn_ldexp((double)aI[g][t][r], -er[r])});
            }
        }
    }
    if (i16 == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = Mx<R>::row(kq, r);
            if (i < tl.count) xout[tl.start + i] = (xprev ? xprev[tl.start + i] : 0) + er[r];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// yhat_i = X_i^T B_c Y_i for MT tiles of 16 series per workgroup (+ the per-tile loss terms).
//   KLD: the tile's own class (legacy loss_functions.jl:468-491);  MSE: every class, blockIdx.y (:599-622).
// 8 waves share the staged factors (LE, RE rows and the two product states of every series) and split the (column group,
// K range) space; a wave's B fragments (from L2) feed MT * NT MFMA chains, its A fragments NT.  The partial sums meet in LDS
// in a fixed order; yhat and the loss are fp64.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int TY_MT = 2, TY_NT = 4, TY_W = 8;
template <typename R, bool CX>
__global__ __launch_bounds__(64 * TY_W) void k_tyhat(TView v, int lid) {
    using E = typename Et<R, CX>::type;
    using acc_t = typename Mx<R>::acc_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const TDims b = tdims(v, lid);
    const int d = v.d, rid = lid + 1;
    const bool mse = v.loss == MPST_LOSS_MSE;
    const int t0 = blockIdx.x * TY_MT;
    const int ntl = min(TY_MT, v.ntiles - t0);
    const int LS = (v.cap + 1) | 1, DS = d | 1;
    constexpr int NS = 16 * TY_MT;
    E* Lf = reinterpret_cast<E*>(smem_raw);          // [NS][LS] left environment rows
    E* Rr = Lf + NS * LS;                            // [NS][LS] right environment rows
    E* Pl = Rr + NS * LS;                            // [NS][DS]
    E* Pr = Pl + NS * DS;                            // [NS][DS]
    double* red = reinterpret_cast<double*>(Pr + NS * DS + 2);     // [TY_W][NS][2]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, kq = lane >> 4;
    const void* LEp = lid > 0 ? (const char*)v.LE + (size_t)(lid - 1) * v.N * v.cap * sizeof(E) : nullptr;
    const void* REn = rid < v.T - 1 ? (const char*)v.RE + (size_t)(rid + 1) * v.N * v.cap * sizeof(E) : nullptr;
    int cls[TY_MT];
    for (int m = 0; m < TY_MT; ++m) cls[m] = m < ntl ? v.tiles[t0 + m].cls : -1;
    for (int i = wave; i < NS; i += TY_W) {
        const int m = i >> 4;
        bool ok = false;
        int64_t smp = 0;
        if (m < ntl) {
            const Span tl = v.tiles[t0 + m];
            ok = (i & 15) < tl.count;
            smp = tl.start + (ok ? (i & 15) : 0);
        }
        const Cp<R> zero = {R(0), R(0)}, one = {R(1), R(0)};
        for (int a = lane; a < LS; a += 64) {
            est<R, CX>(Lf, i * LS + a, (ok && a < b.Dl) ? (LEp ? eld<R, CX>(LEp, smp * v.cap + a) : one) : zero);
            est<R, CX>(Rr, i * LS + a, (ok && a < b.Dr) ? (REn ? eld<R, CX>(REn, smp * v.cap + a) : one) : zero);
        }
        if (lane < d) {
            est<R, CX>(Pl, i * DS + lane, ok ? eld<R, CX>(v.phi, ((int64_t)lid * v.N + smp) * d + lane) : zero);
            est<R, CX>(Pr, i * DS + lane, ok ? eld<R, CX>(v.phi, ((int64_t)rid * v.N + smp) * d + lane) : zero);
        }
    }
    __syncthreads();
    const int XP = (b.X + 3) & ~3;
    const int nty = (b.Y + 15) >> 4, ncg = (nty + TY_NT - 1) / TY_NT;
    const int KSP = ncg >= TY_W ? 1 : TY_W / ncg;                       // K ranges per column group: at least TY_W wave tasks
    constexpr int UB = 4;
    const int Kc = ((XP + KSP - 1) / KSP + 4 * UB - 1) / (4 * UB) * (4 * UB);
    const int dhi = 4 / d, dlo = 4 - dhi * d;
    double pr_[TY_MT][4], pi_[TY_MT][4];
#pragma unroll
    for (int m = 0; m < TY_MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) pr_[m][r] = pi_[m][r] = 0.0;
    // KLD: the tiles of a workgroup may belong to two classes at a class boundary: one pass per distinct class
    const int npass = (!mse && ntl == 2 && cls[1] != cls[0]) ? 2 : 1;
    for (int pass = 0; pass < npass; ++pass) {
        const int c = mse ? (int)blockIdx.y : cls[pass];
        const int64_t bo = (int64_t)c * b.L;
        for (int task = wave; task < ncg * KSP; task += TY_W) {
            const int cg = task % ncg, ks = task / ncg;
            const int kbeg = ks * Kc, kend = min(XP, kbeg + Kc);
            if (kbeg >= kend) continue;
            acc_t aR[TY_MT][TY_NT], aI[TY_MT][TY_NT], a3[TY_MT][CX ? TY_NT : 1];
#pragma unroll
            for (int m = 0; m < TY_MT; ++m)
#pragma unroll
                for (int t = 0; t < TY_NT; ++t) aR[m][t] = aI[m][t] = a3[m][CX ? t : 0] = acc_t{0, 0, 0, 0};
            int ka = (kbeg + kq) / d, ksx = (kbeg + kq) - ka * d;        // x = ka * d + ksx
            for (int kb = kbeg; kb < kend; kb += 4 * UB) {
                Cp<R> bv[TY_NT][UB];
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int kx = kb + 4 * u + kq;
#pragma unroll
                    for (int t = 0; t < TY_NT; ++t) {
                        const int col = (cg * TY_NT + t) * 16 + i16;
                        bv[t][u] = (kx < b.X && col < b.Y) ? eld<R, CX>(v.bt, bo + (int64_t)kx * b.Y + col) : Cp<R>{R(0), R(0)};
                    }
                }
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    if (kb + 4 * u < kend) {
#pragma unroll
                        for (int m = 0; m < TY_MT; ++m) {
                            // rows beyond X: ka >= Dl reads a zero of the padded row (LS > cap >= Dl)
                            const Cp<R> a = cmulc(eld<R, CX>(Lf, (m * 16 + i16) * LS + min(ka, LS - 1)), eld<R, CX>(Pl, (m * 16 + i16) * DS + ksx));
#pragma unroll
                            for (int t = 0; t < TY_NT; ++t) mstep3<R, CX>(a, bv[t][u], aR[m][t], aI[m][t], a3[m][CX ? t : 0]);
                        }
                    }
                    ksx += dlo;
                    ka += dhi;
                    if (ksx >= d) {
                        ksx -= d;
                        ++ka;
                    }
                }
            }
            if constexpr (CX) {
#pragma unroll
                for (int m = 0; m < TY_MT; ++m)
#pragma unroll
                    for (int t = 0; t < TY_NT; ++t) combine3(aR[m][t], aI[m][t], a3[m][t]);
            }
#pragma unroll
            for (int t = 0; t < TY_NT; ++t) {
                const int col = (cg * TY_NT + t) * 16 + i16;
                if (col < b.Y) {
                    const int sy = col / b.Dr, bb = col - sy * b.Dr;
#pragma unroll
                    for (int m = 0; m < TY_MT; ++m) {
                        if (mse || cls[m] == c) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int i = m * 16 + Mx<R>::row(kq, r);
                                const Cp<R> y = cmulc(eld<R, CX>(Rr, i * LS + bb), eld<R, CX>(Pr, i * DS + sy));     // conj(ps_r) RE
                                pr_[m][r] += (double)aR[m][t][r] * (double)y.re - (double)aI[m][t][r] * (double)y.im;
                                if constexpr (CX) pi_[m][r] += (double)aR[m][t][r] * (double)y.im + (double)aI[m][t][r] * (double)y.re;
                            }
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int m = 0; m < TY_MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double xr = sum16(pr_[m][r]), xi = sum16(pi_[m][r]);
            if (i16 == 0) {
                const int i = m * 16 + Mx<R>::row(kq, r);
                red[(wave * NS + i) * 2 + 0] = xr;
                red[(wave * NS + i) * 2 + 1] = xi;
            }
        }
    __syncthreads();
    if (tid < 64) {
        const int m = tid >> 4, i = tid & 15;
        double term = 0.0;
        int c = 0, xmax = INT32_MIN;
        if (m < ntl) {
            const Span tl = v.tiles[t0 + m];
            c = mse ? (int)blockIdx.y : tl.cls;
            double yr = 0.0, yi = 0.0;
            for (int w = 0; w < TY_W; ++w) {
                yr += red[(w * NS + m * 16 + i) * 2 + 0];
                yi += red[(w * NS + m * 16 + i) * 2 + 1];
            }
            if (v.yhat_scaled) {                    // track_cost: loss at the normalised bt_new (loss_functions.jl:177-184)
                yr *= v.sc->inv_norm;
                yi *= v.sc->inv_norm;
            }
            if (i < tl.count) {
                // the environments are scaled by 2^-xL, 2^-xR (k_tenv): yhat = (yr, yi) 2^xs.  What is stored is the SCALED
                // overlap and its exponent - all the KLD gradient needs (phi~/yhat does not see the scale)
                const int64_t smp = tl.start + i;
                const int xs = (lid > 0 ? v.xLE[(int64_t)(lid - 1) * v.N + smp] : 0) + (rid < v.T - 1 ? v.xRE[(int64_t)(rid + 1) * v.N + smp] : 0);
                double* yo = v.yhat + ((int64_t)c * v.N + smp) * 2;
                yo[0] = yr;
                yo[1] = yi;
                if (c == (mse ? 0 : tl.cls)) v.yexp[smp] = xs;
                if (mse && c == 0) xmax = xs;
                if (mse) {
                    const double mm = (tl.cls == c) ? 1.0 : 0.0;
                    const double tr_ = ldexp(yr, xs), ti_ = ldexp(yi, xs);
                    term = 0.5 * ((tr_ - mm) * (tr_ - mm) + ti_ * ti_);
                } else {
                    term = -log(yr * yr + yi * yi) - 2.0 * 0.6931471805599453 * (double)xs;
                }
            }
        }
        term = sum16(term);
        if (i == 0 && m < ntl) v.tile_loss[(int64_t)(mse ? c : 0) * v.ntiles + t0 + m] = term;
        if (mse && blockIdx.y == 0) {        // MSE weights are staged relative to the largest exponent (k_tgrad)
            for (int o = 32; o; o >>= 1) xmax = max(xmax, __shfl_xor(xmax, o));
            if (tid == 0 && xmax != INT32_MIN) atomicMax(&v.sc->xref, xmax);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// gradient partials.  Output-stationary blocks of (AB left-bond values x all d site states) x (all d site states x BB
// right-bond values), AB = BB = max(1, 64 / d): a series contributes AB + BB + 2 d + 1 numbers to a block instead of two
// environment rows.  A workgroup = (share of the class' 64-series chunks, block, class); it stages the factors of a chunk,
// forms the two operand tiles in LDS
//     A_i[(a,s)] = conj(LE_i[a]) ps_i[lid][s] u_i,   B_i[(s,b)] = ps_i[rid][s] conj(RE_i[b])       (= conj(X_i) u_i, conj(Y_i))
// and runs the MFMA chains over the series with the accumulators in registers; ONE partial block per workgroup.
//   KLD: u_i = conj(1 / yhat_i) over the series of class c;  MSE: u_i = yhat_i^c - [label_i == c] over all series.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int TG_B = 64;            // block edge (rows / columns of the output block, and the LDS tile width)
constexpr int TG_LD = 80;           // LDS row stride of the operand tiles: rows k, k+1, k+2, k+3 of one MFMA step on different banks
__host__ __device__ inline int tg_ab(int d) { return d >= TG_B ? 1 : TG_B / d; }
template <typename R, bool CX>
__global__ __launch_bounds__(256) void k_tgrad(TView v, int lid, int nsplit) {
    using E = typename Et<R, CX>::type;
    using acc_t = typename Mx<R>::acc_t;
    constexpr int SS = sizeof(E) == 16 ? 32 : 64;        // series per stage
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const TDims b = tdims(v, lid);
    const int d = v.d, rid = lid + 1;
    const int AB = tg_ab(d);
    const int nbx = (b.Dl + AB - 1) / AB, nby = (b.Dr + AB - 1) / AB;
    const int nby_cap = (v.cap + AB - 1) / AB;
    const int bx = blockIdx.y / nby_cap, by = blockIdx.y - bx * nby_cap;
    if (bx >= nbx || by >= nby) return;
    const int a0 = bx * AB, b0 = by * AB;
    const bool mse = v.loss == MPST_LOSS_MSE;
    const int c = blockIdx.z;
    const int kc0 = mse ? 0 : v.cls_chunk_off[c], kc1 = mse ? v.nchunks : v.cls_chunk_off[c + 1];
    const int ka = kc0 + (int)(((int64_t)(kc1 - kc0) * blockIdx.x) / nsplit);
    const int kb = kc0 + (int)(((int64_t)(kc1 - kc0) * (blockIdx.x + 1)) / nsplit);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, kq = lane >> 4;
    const int FS = 2 * AB + 2 * d + 1;                   // factor record of a series: LE slice, RE slice, ps_l, ps_r, u
    E* As = reinterpret_cast<E*>(smem_raw);              // [SS][TG_LD]
    E* Bs = As + SS * TG_LD;                             // [SS][TG_LD]
    E* Fs = Bs + SS * TG_LD;                             // [SS][FS | 1]
    const int FSP = FS | 1;
    const void* LEp = lid > 0 ? (const char*)v.LE + (size_t)(lid - 1) * v.N * v.cap * sizeof(E) : nullptr;
    const void* REn = rid < v.T - 1 ? (const char*)v.RE + (size_t)(rid + 1) * v.N * v.cap * sizeof(E) : nullptr;
    // operand-tile column of this thread: row r of the block = (a0 + r / d, r % d); column q = (q / AB, b0 + q % AB)
    const int col = tid & 63, i0 = tid >> 6;
    const int ra = col / d, rs = col - ra * d;
    const bool rowv = col < AB * d && a0 + ra < b.Dl;
    const int qs = col / AB, qb = col - qs * AB;
    const bool colv = qs < d && b0 + qb < b.Dr;
    acc_t aR[4], aI[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) aR[t] = aI[t] = acc_t{0, 0, 0, 0};
    const Cp<R> zero = {R(0), R(0)}, one = {R(1), R(0)};
    for (int k = ka; k < kb; ++k) {
        const Span ch = v.chunks[k];
        for (int h0 = 0; h0 < ch.count; h0 += SS) {
            const int cnt = min(SS, ch.count - h0);
            __syncthreads();                 // the previous stage's MFMAs are done with the tiles
            // ---- stage the factors: 4 threads per series ----
            {
                const int i = tid >> 2, j = tid & 3;
                if (i < SS) {
                    const bool ok = i < cnt;
                    const int64_t smp = ch.start + h0 + (ok ? i : 0);
                    E* f = Fs + i * FSP;
                    for (int a = j; a < AB; a += 4) {
                        est<R, CX>(f, a, (ok && a0 + a < b.Dl) ? (LEp ? eld<R, CX>(LEp, smp * v.cap + a0 + a) : one) : zero);
                        est<R, CX>(f, AB + a, (ok && b0 + a < b.Dr) ? (REn ? eld<R, CX>(REn, smp * v.cap + b0 + a) : one) : zero);
                    }
                    for (int s = j; s < d; s += 4) {
                        est<R, CX>(f, 2 * AB + s, ok ? eld<R, CX>(v.phi, ((int64_t)lid * v.N + smp) * d + s) : zero);
                        est<R, CX>(f, 2 * AB + d + s, ok ? eld<R, CX>(v.phi, ((int64_t)rid * v.N + smp) * d + s) : zero);
                    }
                    if (j == 0) {
                        Cp<R> u = zero;
                        if (ok) {
                            const double* yp = v.yhat + ((int64_t)c * v.N + smp) * 2;
                            const double yr = yp[0], yi = yp[1];
                            if (mse) {
                                // u 2^xs with the true overlap yhat = (yr, yi) 2^xs: the operands below carry 2^-xs.  Staged
                                // relative to the bond's largest exponent (2^xs leaves fp32's range after ~85 Fourier sites);
                                // k_tgrad_reduce puts 2^xref back in fp64
                                const int xs = v.yexp[smp], xr = xs - v.sc->xref;
                                u = Cp<R>{(R)ldexp(ldexp(yr, xs) - ((ch.cls == c) ? 1.0 : 0.0), xr), (R)ldexp(yi, xs + xr)};
                            } else {
                                const double q = 1.0 / (yr * yr + yi * yi);       // conj(1 / yhat) = yhat / |yhat|^2
                                u = Cp<R>{(R)(yr * q), (R)(yi * q)};
                            }
                        }
                        est<R, CX>(f, 2 * AB + 2 * d, u);
                    }
                }
            }
            __syncthreads();
            // ---- operand tiles ----
#pragma unroll 4
            for (int m = 0; m < SS / 4; ++m) {
                const int i = i0 + 4 * m;
                const E* f = Fs + i * FSP;
                Cp<R> av = zero, bv = zero;
                if (rowv) {
                    const Cp<R> le = eld<R, CX>(f, ra), pl = eld<R, CX>(f, 2 * AB + rs), u = eld<R, CX>(f, 2 * AB + 2 * d);
                    av = cmul(cmulc(pl, le), u);                                   // conj(LE) ps_l u
                }
                if (colv) {
                    const Cp<R> re = eld<R, CX>(f, AB + qb), pr = eld<R, CX>(f, 2 * AB + d + qs);
                    bv = cmulc(pr, re);                                            // ps_r conj(RE)
                }
                est<R, CX>(As, i * TG_LD + col, av);
                est<R, CX>(Bs, i * TG_LD + col, bv);
            }
            __syncthreads();
            const int kmax = (cnt + 3) & ~3;
            for (int k0 = 0; k0 < kmax; k0 += 4) {
                const Cp<R> a = eld<R, CX>(As, (k0 + kq) * TG_LD + wave * 16 + i16);
#pragma unroll
                for (int t = 0; t < 4; ++t) mstep<R, CX>(a, eld<R, CX>(Bs, (k0 + kq) * TG_LD + t * 16 + i16), aR[t], aI[t]);
            }
        }
    }
    const int64_t po = ((int64_t)c * nsplit + blockIdx.x) * (int64_t)b.L;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int q = t * 16 + i16;
        const int s2 = q / AB, bb = b0 + (q - s2 * AB);
        if (s2 >= d || bb >= b.Dr) continue;
        const int y = s2 * b.Dr + bb;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rr = wave * 16 + Mx<R>::row(kq, r);
            const int aa = a0 + rr / d;
            if (rr < AB * d && aa < b.Dl) est<R, CX>(v.partial, po + (int64_t)(a0 * d + rr) * b.Y + y, Cp<R>{aR[t][r], aI[t][r]});
        }
    }
}

// grad[c] = scale_c * sum over the shares, in fp64; gradbuf = [loss, 0, grad (interleaved pairs when complex)] is the buffer the
// multi-GPU all-reduce sums.  Every workgroup also leaves its piece of ||grad||^2 (k_tupdate adds the pieces in a fixed order).
template <typename R, bool CX>
__global__ __launch_bounds__(256) void k_tgrad_reduce(TView v, int lid, int nsplit) {
    __shared__ double red[4];
    constexpr int ZW = CX ? 2 : 1;
    const TDims b = tdims(v, lid);
    const bool mse = v.loss == MPST_LOSS_MSE;
    const int64_t total = (int64_t)v.C * b.L;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double g2 = 0.0;
    if (idx < total) {
        const int c = (int)(idx / b.L);
        const int64_t e = idx - (int64_t)c * b.L;
        const double scale = mse ? ldexp(v.invN, v.sc->xref) : -(v.train_sep ? v.inv_count[c] : v.invN);
        double sr = 0.0, si = 0.0;
        for (int k = 0; k < nsplit; ++k) {
            const Cp<R> p = eld<R, CX>(v.partial, ((int64_t)c * nsplit + k) * b.L + e);
            sr += (double)p.re;
            si += (double)p.im;
        }
        sr *= scale;
        si *= scale;
        v.gradbuf[2 + idx * ZW] = sr;
        if constexpr (CX) v.gradbuf[2 + idx * ZW + 1] = si;
        g2 = sr * sr + si * si;
    }
    const double piece = tblock_sum(g2, red);
    if (threadIdx.x == 0) v.norm_part[blockIdx.x] = piece;
    if (blockIdx.x == 0) {
        double s = 0.0;
        if (mse) {
            for (int i = threadIdx.x; i < v.C * v.ntiles; i += 256) s += v.tile_loss[i];
            s *= v.invN;
        } else {
            for (int i = threadIdx.x; i < v.ntiles; i += 256) s += v.tile_loss[i] * (v.train_sep ? v.inv_count[v.tiles[i].cls] : v.invN);
        }
        const double tot = tblock_sum(s, red);
        if (threadIdx.x == 0) {
            v.gradbuf[0] = tot;
            v.gradbuf[1] = 0.0;
        }
    }
}

// after an all-reduce: the pieces of ||grad||^2 again, from the summed gradient
template <bool CX>
__global__ __launch_bounds__(256) void k_tgrad_norm(TView v, int lid) {
    __shared__ double red[4];
    constexpr int ZW = CX ? 2 : 1;
    const TDims b = tdims(v, lid);
    const int64_t total = (int64_t)v.C * b.L;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double g2 = 0.0;
    if (idx < total) {
        const double gr = v.gradbuf[2 + idx * ZW], gi = CX ? v.gradbuf[2 + idx * ZW + 1] : 0.0;
        g2 = gr * gr + gi * gi;
    }
    const double piece = tblock_sum(g2, red);
    if (threadIdx.x == 0) v.norm_part[blockIdx.x] = piece;
}

// TSGO: bt -= eta grad / ||grad|| (legacy loss_functions.jl:151);  GD: bt -= eta grad (:123), evaluated in fp64
template <typename R, bool CX>
__global__ __launch_bounds__(256) void k_tupdate(TView v, int lid, int first_iter) {
    __shared__ double red[4];
    constexpr int ZW = CX ? 2 : 1;
    const TDims b = tdims(v, lid);
    const int64_t n = (int64_t)v.C * b.L * ZW;
    double s = 0.0;
    for (int i = threadIdx.x; i < v.n_norm_part; i += 256) s += v.norm_part[i];
    const double nrm = sqrt(tblock_sum(s, red));
    const double step = (v.optimiser == MPST_OPT_TSGO) ? (nrm > 0.0 ? v.eta / nrm : 0.0) : v.eta;   // a vanished gradient: no step
    R* p = (R*)v.bt;
    const double* g = v.gradbuf + 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = (R)((double)p[i] - step * g[i]);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (first_iter) {
            v.sc->loss = v.gradbuf[0];
            v.sc->grad_norm = nrm;
        }
        if (v.trace) v.trace[v.trace_it] = v.gradbuf[0];
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Gram matrix of the bond tensor as decomposeBT hands it to svd (RealRealLegacyITensor.jl:103-107 / :122-126), fp64:
//   going left : rows (a, c, s_l), cols (s_r, b)  -> G = sum_c B_c^H B_c   (Y x Y)
//   going right: rows (b, c, s_r), cols (s_l, a)  -> G = sum_c conj(B_c) B_c^T   (X x X)
// Complex: G = Gr + i Gi is Hermitian; what the real eigensolvers get is its embedding [[Gr, -Gi], [Gi, Gr]] (2n x 2n).
// One 16 x 16 tile per workgroup, K split over its 4 waves.
// ---------------------------------------------------------------------------------------------------------------------------
template <typename R, bool CX>
__global__ __launch_bounds__(256) void k_tgram(TView v, int lid, int going_left) {
    __shared__ double part[2][4][256];
    const TDims b = tdims(v, lid);
    const int n = going_left ? b.Y : b.X;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i16 = lane & 15, kq = lane >> 4;
    const int tn = (n + 15) >> 4;
    const int tile = blockIdx.x;
    if (tile >= tn * tn) return;
    const int m0 = (tile / tn) * 16, n0 = (tile % tn) * 16;
    const int m = m0 + i16, nn = n0 + i16;
    d4 aR = {0, 0, 0, 0}, aI = {0, 0, 0, 0};
    // G[m][n] = sum_k conj(A[k][m]) A[k][n]
    if (going_left) {
        const int K = v.C * b.X;             // k = (c, x): element (k, p) at k*Y + p
        const int kq4 = (((K + 3) >> 2) + 3) & ~3;
        wave_tile64<CX>([&](int k) { Cp<double> a = m < n ? todbl(eld<R, CX>(v.bt, (int64_t)k * b.Y + m)) : Cp<double>{0, 0}; a.im = -a.im; return a; },
                        [&](int k) { return nn < n ? todbl(eld<R, CX>(v.bt, (int64_t)k * b.Y + nn)) : Cp<double>{0, 0}; },
                        wave * kq4, min(K, (wave + 1) * kq4), aR, aI);
    } else {
        const int K = b.Y;
        const int kq4 = (((K + 3) >> 2) + 3) & ~3;
        for (int c = 0; c < v.C; ++c) {
            const int64_t bo = (int64_t)c * b.L;
            wave_tile64<CX>([&](int k) { Cp<double> a = m < n ? todbl(eld<R, CX>(v.bt, bo + (int64_t)m * b.Y + k)) : Cp<double>{0, 0}; a.im = -a.im; return a; },
                            [&](int k) { return nn < n ? todbl(eld<R, CX>(v.bt, bo + (int64_t)nn * b.Y + k)) : Cp<double>{0, 0}; },
                            wave * kq4, min(K, (wave + 1) * kq4), aR, aI);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        part[0][wave][r * 64 + lane] = aR[r];
        if constexpr (CX) part[1][wave][r * 64 + lane] = aI[r];
    }
    __syncthreads();
    if (wave == 0) {
        const int ne = CX ? 2 * n : n;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + kq + 4 * r, col = nn;
            const double gr = (part[0][0][r * 64 + lane] + part[0][1][r * 64 + lane]) + (part[0][2][r * 64 + lane] + part[0][3][r * 64 + lane]);
            if (row < n && col < n) {
                v.gram[(int64_t)row * ne + col] = gr;
                if constexpr (CX) {
                    const double gi = (part[1][0][r * 64 + lane] + part[1][1][r * 64 + lane]) + (part[1][2][r * 64 + lane] + part[1][3][r * 64 + lane]);
                    v.gram[(int64_t)(n + row) * ne + n + col] = gr;
                    v.gram[(int64_t)row * ne + n + col] = -gi;
                    v.gram[(int64_t)(n + row) * ne + col] = gi;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// decomposeBT back-split from the kept eigenvectors (RealRealLegacyITensor.jl:109-110 / :128-129).  With E the matrix of
// right singular vectors (columns; complex: column k of the embedding's eigenvector block is (Re E_k, Im E_k), see View::zw):
//   going left : W[lid][c] = B_c E * inv_norm (= U S, keeps the label),  W[rid][k][y] = conj(E[y][k])   (ITensors' V = V^H)
//   going right: W[rid][c] = E^T B_c * inv_norm (= V S),                 W[lid][x][k] = conj(E[x][k])
// The products run on the fp64 MFMA from the stored element type.
// ---------------------------------------------------------------------------------------------------------------------------
template <typename R, bool CX>
__global__ __launch_bounds__(256) void k_tsplit(TView v, int lid, int going_left) {
    const TDims b = tdims(v, lid);
    const int nk = v.sc->n_keep;
    const int ldE = v.ldE, kst = CX ? 2 : 1;
    const double inv = v.sc->inv_norm;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i16 = lane & 15, kq = lane >> 4;
    const int64_t ol = (int64_t)lid * v.site_stride, orr = (int64_t)(lid + 1) * v.site_stride;
    const int nc = going_left ? b.Y : b.X;
    auto Ec = [&](int z, int k) -> Cp<double> {          // E[z][k], z < nc, k < nk
        Cp<double> e;
        e.re = v.E[(int64_t)z * ldE + kst * k];
        e.im = CX ? v.E[(int64_t)(nc + z) * ldE + kst * k] : 0.0;
        return e;
    };
    const int tk = (nk + 15) >> 4;
    if (going_left) {
        const int tx = (b.X + 15) >> 4;
        const int ntile = v.C * tx * tk;
        for (int tile = blockIdx.x * 4 + wave; tile < ntile; tile += gridDim.x * 4) {
            const int c = tile / (tx * tk), rem = tile - c * tx * tk;
            const int m0 = (rem / tk) * 16, n0 = (rem % tk) * 16;
            const int m = m0 + i16, n = n0 + i16;
            const int64_t bo = (int64_t)c * b.L;
            d4 aR = {0, 0, 0, 0}, aI = {0, 0, 0, 0};
            wave_tile64<CX>([&](int k) { return m < b.X ? todbl(eld<R, CX>(v.bt, bo + (int64_t)m * b.Y + k)) : Cp<double>{0, 0}; },
                            [&](int k) { return n < nk ? Ec(k, n) : Cp<double>{0, 0}; }, 0, b.Y, aR, aI);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + kq + 4 * r;
                if (row < b.X && n < nk) est<R, CX>(v.sites, ol + (int64_t)c * b.X * nk + (int64_t)row * nk + n, Cp<R>{(R)(aR[r] * inv), (R)(aI[r] * inv)});
            }
        }
        for (int i = blockIdx.x * 256 + threadIdx.x; i < nk * b.Y; i += gridDim.x * 256) {
            const int k = i / b.Y, y = i - k * b.Y;
            const Cp<double> e = Ec(y, k);
            est<R, CX>(v.sites, orr + i, Cp<R>{(R)e.re, (R)(-e.im)});
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) *v.label_site = lid;
    } else {
        const int ty = (b.Y + 15) >> 4;
        const int ntile = v.C * tk * ty;
        for (int tile = blockIdx.x * 4 + wave; tile < ntile; tile += gridDim.x * 4) {
            const int c = tile / (tk * ty), rem = tile - c * tk * ty;
            const int m0 = (rem / ty) * 16, n0 = (rem % ty) * 16;
            const int m = m0 + i16, n = n0 + i16;
            const int64_t bo = (int64_t)c * b.L;
            d4 aR = {0, 0, 0, 0}, aI = {0, 0, 0, 0};
            wave_tile64<CX>([&](int k) { return m < nk ? Ec(k, m) : Cp<double>{0, 0}; },
                            [&](int k) { return n < b.Y ? todbl(eld<R, CX>(v.bt, bo + (int64_t)k * b.Y + n)) : Cp<double>{0, 0}; }, 0, b.X, aR, aI);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + kq + 4 * r;
                if (row < nk && n < b.Y) est<R, CX>(v.sites, orr + (int64_t)c * nk * b.Y + (int64_t)row * b.Y + n, Cp<R>{(R)(aR[r] * inv), (R)(aI[r] * inv)});
            }
        }
        for (int i = blockIdx.x * 256 + threadIdx.x; i < b.X * nk; i += gridDim.x * 256) {
            const int x = i / nk, k = i - x * nk;
            const Cp<double> e = Ec(x, k);
            est<R, CX>(v.sites, ol + i, Cp<R>{(R)e.re, (R)(-e.im)});
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) *v.label_site = lid + 1;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// evaluation (src/summary.jl:4-136): yhat_i[c] = sum L_i[a] conj(ps_i[s]) R_i[b] W_p[c][a][s][b], fp64 accumulation
// ---------------------------------------------------------------------------------------------------------------------------
template <typename R, bool CX>
__global__ __launch_bounds__(256) void k_teval_final(TView v, const void* __restrict__ Lc, const int32_t* __restrict__ Lx, const void* __restrict__ Rc,
                                                     const int32_t* __restrict__ Rx, double* __restrict__ yout) {
    const int p = *v.label_site;
    const int Dl = v.chi[p], Dr = v.chi[p + 1], d = v.d;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= v.N) return;
    const int64_t so = (int64_t)p * v.site_stride;
    for (int c = 0; c < v.C; ++c) {
        const int64_t wo = so + (int64_t)c * Dl * d * Dr;
        double ar = 0.0, ai = 0.0;
        for (int a = 0; a < Dl; ++a) {
            const Cp<double> la = (p > 0) ? todbl(eld<R, CX>(Lc, i * v.cap + a)) : Cp<double>{1.0, 0.0};
            for (int s = 0; s < d; ++s) {
                const Cp<double> ls = cmulc(la, todbl(eld<R, CX>(v.phi, ((int64_t)p * v.N + i) * d + s)));
                double tr = 0.0, ti = 0.0;
                for (int bb = 0; bb < Dr; ++bb) {
                    const Cp<double> w = todbl(eld<R, CX>(v.sites, wo + ((int64_t)a * d + s) * Dr + bb));
                    const Cp<double> rr = (p < v.T - 1) ? todbl(eld<R, CX>(Rc, i * v.cap + bb)) : Cp<double>{1.0, 0.0};
                    const Cp<double> t = cmul(w, rr);
                    tr += t.re;
                    ti += t.im;
                }
                ar += ls.re * tr - ls.im * ti;
                ai += ls.re * ti + ls.im * tr;
            }
        }
        const int xs = (p > 0 ? Lx[i] : 0) + (p < v.T - 1 ? Rx[i] : 0);       // the chains are scaled by powers of two (k_tenv)
        yout[(i * v.C + c) * 2 + 0] = ldexp(ar, xs);
        yout[(i * v.C + c) * 2 + 1] = ldexp(ai, xs);
    }
}

// MSE_loss_acc_iter (summary.jl:33-58) on (re, im) overlaps: one workgroup, fixed-order sums
__global__ __launch_bounds__(1024) void k_teval_reduce(TView v, const double* __restrict__ yin, double* out3, int64_t* conf, int32_t* pred) {
    __shared__ double red[16];
    __shared__ int cm[MAX_C * MAX_C];
    const int C = v.C;
    for (int i = threadIdx.x; i < C * C; i += 1024) cm[i] = 0;
    __syncthreads();
    double mse = 0.0, kld = 0.0, acc = 0.0;
    for (int64_t i = threadIdx.x; i < v.N; i += 1024) {
        const double* y = yin + i * C * 2;
        const int lab = v.label[i];
        double s = 0.0, best = -1.0;
        int arg = 0;
        for (int c = 0; c < C; ++c) {
            const double tr = y[2 * c] - (c == lab ? 1.0 : 0.0), ti = y[2 * c + 1];
            s += tr * tr + ti * ti;                                    // abs2.(yhat - y)  :45
            const double ab = y[2 * c] * y[2 * c] + y[2 * c + 1] * y[2 * c + 1];      // argmax(abs.(yhat)) :52 - same order as abs2
            if (ab > best) {
                best = ab;
                arg = c;
            }
        }
        mse += 0.5 * s;
        kld += -log(y[2 * lab] * y[2 * lab] + y[2 * lab + 1] * y[2 * lab + 1]);       // -log(abs2(yhat[label]))  :48
        acc += (arg == lab) ? 1.0 : 0.0;
        if (pred) pred[i] = arg;
        atomicAdd(&cm[lab * C + arg], 1);
    }
    const double tm = tblock_sum(mse, red);
    const double tk = tblock_sum(kld, red);
    const double ta = tblock_sum(acc, red);
    __syncthreads();
    if (threadIdx.x == 0) {
        out3[0] = tm;
        out3[1] = tk;
        out3[2] = ta;
    }
    if (conf)
        for (int i = threadIdx.x; i < C * C; i += 1024) conf[i] = cm[i];
}

// ---------------------------------------------------------------------------------------------------------------------------
// normalize!(W) (RealRealHighDimension.jl:852): <W|W> by transfer matrices in one workgroup, fp64 complex in global scratch:
//   E'[r1][r2] = sum_{s(,c)} sum_{a1,a2} conj(A_s[a1][r1]) E[a1][a2] A_s[a2][r2].
// ---------------------------------------------------------------------------------------------------------------------------
template <typename R, bool CX>
__global__ __launch_bounds__(1024) void k_tnorm2(TView v, double* out_norm2, double* gs) {
    const int cm = v.cap;
    double* E = gs;                    // [cm][cm][2]
    double* Tm = E + 2 * cm * cm;      // T = E A_s
    double* En = Tm + 2 * cm * cm;
    const int tid = threadIdx.x;
    const int ls = *v.label_site;
    for (int i = tid; i < 2 * cm * cm; i += 1024) E[i] = 0.0;
    __syncthreads();
    if (tid == 0) E[0] = 1.0;
    __syncthreads();
    for (int j = 0; j < v.T; ++j) {
        const int Dl = v.chi[j], Dr = v.chi[j + 1], d = v.d;
        const int Cj = (j == ls) ? v.C : 1;
        const int64_t so = (int64_t)j * v.site_stride;
        for (int i = tid; i < Dr * Dr; i += 1024) En[2 * ((i / Dr) * cm + (i % Dr))] = En[2 * ((i / Dr) * cm + (i % Dr)) + 1] = 0.0;
        __syncthreads();
        for (int c = 0; c < Cj; ++c)
            for (int s = 0; s < d; ++s) {
                const int64_t ao = so + (int64_t)c * Dl * d * Dr + (int64_t)s * Dr;      // A[a][r] at a*d*Dr + r
                for (int i = tid; i < Dl * Dr; i += 1024) {
                    const int a = i / Dr, r = i - a * Dr;
                    double tr = 0.0, ti = 0.0;
                    for (int a2 = 0; a2 < Dl; ++a2) {
                        const Cp<double> w = todbl(eld<R, CX>(v.sites, ao + (int64_t)a2 * d * Dr + r));
                        const double er = E[2 * (a * cm + a2)], ei = E[2 * (a * cm + a2) + 1];
                        tr += er * w.re - ei * w.im;
                        ti += er * w.im + ei * w.re;
                    }
                    Tm[2 * (a * cm + r)] = tr;
                    Tm[2 * (a * cm + r) + 1] = ti;
                }
                __syncthreads();
                for (int i = tid; i < Dr * Dr; i += 1024) {
                    const int r1 = i / Dr, r2 = i - r1 * Dr;
                    double tr = 0.0, ti = 0.0;
                    for (int a = 0; a < Dl; ++a) {
                        const Cp<double> w = todbl(eld<R, CX>(v.sites, ao + (int64_t)a * d * Dr + r1));
                        const double xr = Tm[2 * (a * cm + r2)], xi = Tm[2 * (a * cm + r2) + 1];
                        tr += w.re * xr + w.im * xi;             // conj(w) * x
                        ti += w.re * xi - w.im * xr;
                    }
                    En[2 * (r1 * cm + r2)] += tr;
                    En[2 * (r1 * cm + r2) + 1] += ti;
                }
                __syncthreads();
            }
        for (int i = tid; i < Dr * Dr; i += 1024) {
            const int r1 = i / Dr, r2 = i - r1 * Dr;
            E[2 * (r1 * cm + r2)] = En[2 * (r1 * cm + r2)];
            E[2 * (r1 * cm + r2) + 1] = En[2 * (r1 * cm + r2) + 1];
        }
        __syncthreads();
    }
    if (tid == 0) *out_norm2 = E[0];
}

template <typename R, bool CX>
__global__ __launch_bounds__(256) void k_tscale_sites(TView v, const double* norm2) {
    const int j = blockIdx.x;
    const int ls = *v.label_site;
    const int n = ((j == ls) ? v.C : 1) * v.chi[j] * v.d * v.chi[j + 1] * (CX ? 2 : 1);
    const double z = exp(0.5 * log(*norm2) / (double)v.T);
    R* W = (R*)v.sites + (int64_t)j * v.site_stride * (CX ? 2 : 1);
    for (int i = threadIdx.x; i < n; i += 256) W[i] = (R)((double)W[i] / z);
}

// product states encoded in fp64 (mpst_encode.hip: doubles or (re, im) pairs) into the context's element type
__global__ __launch_bounds__(256) void k_tcast(const double* __restrict__ src, int src_cx, void* __restrict__ dst, int dst_cx, int dst_f32, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double re = src_cx ? src[2 * i] : src[i], im = src_cx ? src[2 * i + 1] : 0.0;
        if (dst_f32) {
            if (dst_cx) ((float2*)dst)[i] = make_float2((float)re, (float)im);
            else ((float*)dst)[i] = (float)re;
        } else {
            if (dst_cx) ((double2*)dst)[i] = make_double2(re, im);
            else ((double*)dst)[i] = re;
        }
    }
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline size_t tesz(const TView& v) { return (v.f32 ? 4 : 8) * (v.cx ? 2 : 1); }

}  // namespace

#define TLAUNCH(v, K, grid, block, lds, s, ...)                                                              \
    do {                                                                                                     \
        if ((v).f32) {                                                                                       \
            if ((v).cx) hipLaunchKernelGGL((K<float, true>), grid, block, lds, s, __VA_ARGS__);              \
            else hipLaunchKernelGGL((K<float, false>), grid, block, lds, s, __VA_ARGS__);                    \
        } else {                                                                                             \
            if ((v).cx) hipLaunchKernelGGL((K<double, true>), grid, block, lds, s, __VA_ARGS__);             \
            else hipLaunchKernelGGL((K<double, false>), grid, block, lds, s, __VA_ARGS__);                   \
        }                                                                                                    \
    } while (0)

int typed_grad_nsplit(const TView& v, int nchunks) {
    const int dm_b = cdiv(v.cap, tg_ab(v.d));
    return grad_nsplit(nchunks, dm_b * dm_b, v.C);
}
int typed_norm_parts(const TView& v) {
    const int64_t tot = (int64_t)v.C * v.d * v.cap * v.d * v.cap;
    return (int)((tot + 255) / 256);
}
static size_t tyhat_lds(const TView& v) {
    const size_t e = tesz(v);
    const int LS = (v.cap + 1) | 1, DS = v.d | 1, NS = 16 * TY_MT;
    return (size_t)(2 * NS * LS + 2 * NS * DS + 2) * e + (size_t)TY_W * NS * 2 * sizeof(double) + 16;
}
static size_t tenv_lds(const TView& v) { return (size_t)4 * 16 * (((v.cap + 1) | 1) + (v.d | 1)) * tesz(v); }
static size_t tgrad_lds(const TView& v) {
    const size_t e = tesz(v);
    const int SS = e == 16 ? 32 : 64;
    return (size_t)(2 * SS * TG_LD + SS * ((2 * tg_ab(v.d) + 2 * v.d + 1) | 1)) * e;
}

void launch_tcast(const double* src, int src_cx, void* dst, int dst_cx, int dst_f32, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_tcast, dim3((unsigned)std::min<int64_t>(4096, (n + 255) / 256)), dim3(256), 0, s, src, src_cx, dst, dst_cx, dst_f32, n);
}
void launch_tbt_assemble(const TView& v, int lid, hipStream_t s) {
    const int dm = v.d * v.cap;
    const int tiles = cdiv(dm, 16) * cdiv(dm, 16);
    TLAUNCH(v, k_tbt_assemble, dim3(cdiv(tiles, 4), v.C), dim3(256), 0, s, v, lid);
}
void launch_tbt_prescale(const TView& v, int lid, hipStream_t s) { TLAUNCH(v, k_tbt_prescale, dim3(1), dim3(1024), 0, s, v, lid); }
void launch_tenv(const TView& v, int site, int left_side, const void* prev, const int32_t* xprev, int prev_bond, int mode, int out_bond, void* out,
                 int32_t* xout, hipStream_t s) {
    TLAUNCH(v, k_tenv, dim3(cdiv(v.ntiles, 4)), dim3(256), tenv_lds(v), s, v, site, left_side, prev, xprev, prev_bond, mode, out_bond, out, xout);
}
void launch_tyhat(const TView& v, int lid, hipStream_t s) {
    const int gy = v.loss == MPST_LOSS_MSE ? v.C : 1;
    TLAUNCH(v, k_tyhat, dim3(cdiv(v.ntiles, TY_MT), gy), dim3(64 * TY_W), tyhat_lds(v), s, v, lid);
}
void launch_tgrad(const TView& v, int lid, hipStream_t s) {
    const int nb1 = cdiv(v.cap, tg_ab(v.d));
    const int nsplit = typed_grad_nsplit(v, v.nchunks);
    TLAUNCH(v, k_tgrad, dim3(nsplit, nb1 * nb1, v.C), dim3(256), tgrad_lds(v), s, v, lid, nsplit);
}
void launch_tgrad_reduce(const TView& v, int lid, hipStream_t s) {
    TLAUNCH(v, k_tgrad_reduce, dim3(typed_norm_parts(v)), dim3(256), 0, s, v, lid, typed_grad_nsplit(v, v.nchunks));
}
void launch_tgrad_norm(const TView& v, int lid, hipStream_t s) {
    if (v.cx) hipLaunchKernelGGL(k_tgrad_norm<true>, dim3(typed_norm_parts(v)), dim3(256), 0, s, v, lid);
    else hipLaunchKernelGGL(k_tgrad_norm<false>, dim3(typed_norm_parts(v)), dim3(256), 0, s, v, lid);
}
void launch_tupdate(const TView& v, int lid, int first_iter, hipStream_t s) {
    const int64_t tot = (int64_t)v.C * v.d * v.cap * v.d * v.cap * (v.cx ? 2 : 1);
    const int grid = (int)std::min<int64_t>(512, std::max<int64_t>(32, (tot + 1023) / 1024));
    TLAUNCH(v, k_tupdate, dim3(grid), dim3(256), 0, s, v, lid, first_iter);
}
void launch_tgram(const TView& v, int lid, int going_left, hipStream_t s) {
    const int dm = v.d * v.cap;
    TLAUNCH(v, k_tgram, dim3(cdiv(dm, 16) * cdiv(dm, 16)), dim3(256), 0, s, v, lid, going_left);
}
void launch_tsplit(const TView& v, int lid, int going_left, hipStream_t s) {
    const int dm = v.d * v.cap;
    const int tiles = v.C * cdiv(dm, 16) * cdiv(v.cap, 16);
    TLAUNCH(v, k_tsplit, dim3(cdiv(tiles, 4)), dim3(256), 0, s, v, lid, going_left);
}
void launch_teval_final(const TView& v, const void* Lc, const int32_t* Lx, const void* Rc, const int32_t* Rx, double* yout, hipStream_t s) {
    TLAUNCH(v, k_teval_final, dim3((unsigned)((v.N + 255) / 256)), dim3(256), 0, s, v, Lc, Lx, Rc, Rx, yout);
}
void launch_teval_reduce(const TView& v, const double* yin, double* out3, int64_t* conf, int32_t* pred, hipStream_t s) {
    hipLaunchKernelGGL(k_teval_reduce, dim3(1), dim3(1024), 0, s, v, yin, out3, conf, pred);
}
void launch_tnorm2(const TView& v, double* out_norm2, double* gscratch, hipStream_t s) {
    TLAUNCH(v, k_tnorm2, dim3(1), dim3(1024), 0, s, v, out_norm2, gscratch);
}
void launch_tscale_sites(const TView& v, const double* norm2, hipStream_t s) { TLAUNCH(v, k_tscale_sites, dim3(v.T), dim3(256), 0, s, v, norm2); }

#define TATTR(K, bytes)                                                                                                             \
    do {                                                                                                                            \
        if ((e = hipFuncSetAttribute((const void*)K<float, false>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes)) != hipSuccess) return e;  \
        if ((e = hipFuncSetAttribute((const void*)K<float, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes)) != hipSuccess) return e;   \
        if ((e = hipFuncSetAttribute((const void*)K<double, false>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes)) != hipSuccess) return e; \
        if ((e = hipFuncSetAttribute((const void*)K<double, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes)) != hipSuccess) return e;  \
    } while (0)
hipError_t typed_init_attrs(int device) {
    static std::atomic<unsigned long long> done{0};
    if (device >= 0 && device < 64 && (done.load(std::memory_order_acquire) >> device) & 1ull) return hipSuccess;
    hipError_t e;
    TATTR(k_tenv, 144 * 1024);
    TATTR(k_tyhat, 144 * 1024);
    TATTR(k_tgrad, 144 * 1024);
    if (device >= 0 && device < 64) done.fetch_or(1ull << device, std::memory_order_release);
    return hipSuccess;
}
// LDS the three staged kernels ask for at this shape: the caller rejects shapes beyond the 144 KB opted into above
size_t typed_max_lds(const TView& v) { return std::max(tyhat_lds(v), std::max(tenv_lds(v), tgrad_lds(v))); }

}  // namespace mpst
