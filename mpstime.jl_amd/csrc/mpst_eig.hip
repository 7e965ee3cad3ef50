// Symmetric eigensolver for the Gram matrix of the bond tensor (n = d*chi <= 128), and the
// NDTensors truncation rule.  Device-side stand-in for ITensors.svd -> LAPACK gesdd + truncate!
// in decomposeBT (src/Training/RealRealHighDimension.jl:166-169,185-188).
//
// With A the (chi*C*d) x (d*chi) matrix the reference decomposes, G = A^T A = V S^2 V^T: the
// right singular vectors are the eigenvectors of G, S = sqrt(lambda), and U*S = A V (k_split).
// Only the chi_max largest eigenpairs and the trace are needed: the truncation rule discards the
// rest, and the discarded weight is trace - sum(kept).
//
// Three launches per bond, two algorithms:
//
//  * default:
//    k_eig_tri  (1 workgroup)  Householder tridiagonalisation G = Q T Q^T with the matrix held in
//               registers (32 doubles per thread), 2 barriers per reflector, look-ahead reflector
//               construction; T, the reflectors and Gershgorin bounds go to a 100 KB workspace.
//    k_eig_vec  (one workgroup PER EIGENVALUE, K <= 32 of them run concurrently on 32 CUs)
//               256-way multisection Sturm bisection (7 steps, division-free recurrence), eigenvector
//               of T by twisted factorisation with the two factorisations and the two substitution
//               sweeps on different waves, back-transformation through the reflectors two at a time.
//    k_eig_fin  (1 workgroup)  NDTensors truncation rule on the eigenvalues, on-device verification
//               (residual in T, orthonormality) of the kept vectors, Loewdin re-orthonormalisation,
//               publication of n_keep / chi / E.  If verification fails (genuinely clustered kept
//               eigenvalues) it falls through to
//  * MPST_SVD_JACOBI: one-sided (Hestenes) Jacobi on the columns of G in LDS - slow (ms) but
//    unconditionally robust; column k converges to lambda_k v_k.
#include "mpst_internal.h"
#include "mpst_eig_common.inl"
#include <type_traits>

namespace mpst {

constexpr int EIG_MAX_SWEEPS = 40;
constexpr int VEC_THREADS = 256;  // k_eig_vec: one workgroup per eigenvalue
constexpr int TRI_NSTEP = 7;      // 257^-7 = 1.4e-17 of the Gershgorin interval
constexpr int EIG_LDS_DOUBLES = 17664;  // 138 KB: max over both algorithms


// =====================================================================================
// Jacobi path
// =====================================================================================
struct EigShared {
    double* Gs;     // [np][np] column-major
    double* nrm;    // [np]
    int* rank;      // [np]
    int* flag;      // [2]
};

__device__ __forceinline__ void pair_of(int r, int k, int np, int& p, int& q) {
    const int m = np - 1;
    if (k == 0) {
        p = m;
        q = r;
    } else {
        p = (r + k) % m;
        q = (r - k + m) % m;
    }
    if (p > q) {
        const int t = p;
        p = q;
        q = t;
    }
}

// Returns the number of sweeps used.  On exit Gs columns are orthogonal; nrm[k] = ||col k||.
__device__ int jacobi_core(EigShared sh, int np) {
    const int tid = threadIdx.x;
    const int k = tid >> 4, sub = tid & 15;
    const int npairs = np >> 1;
    const int nrow_it = (np + 15) >> 4;
    const double tol = 2.5e-15;
    int sweeps = 0;
    for (int sweep = 0; sweep < EIG_MAX_SWEEPS; ++sweep) {
        if (tid == 0) sh.flag[0] = 0;
        __syncthreads();
        int rotated = 0;
        for (int r = 0; r < np - 1; ++r) {
            for (int kk = k; kk < npairs; kk += EIG_THREADS >> 4) {
                int p, q;
                pair_of(r, kk, np, p, q);
                double* cp = sh.Gs + (size_t)p * np;
                double* cq = sh.Gs + (size_t)q * np;
                double vp[8], vq[8];
                double app = 0.0, aqq = 0.0, apq = 0.0;
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const int row = sub + 16 * m;
                    const bool ok = m < nrow_it && row < np;
                    vp[m] = ok ? cp[row] : 0.0;
                    vq[m] = ok ? cq[row] : 0.0;
                    app += vp[m] * vp[m];
                    aqq += vq[m] * vq[m];
                    apq += vp[m] * vq[m];
                }
                app = sum16(app);
                aqq = sum16(aqq);
                apq = sum16(apq);
                if (fabs(apq) > tol * sqrt(app * aqq) && app * aqq > 0.0) {
                    const double zeta = (aqq - app) / (2.0 * apq);
                    const double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                    const double c = 1.0 / sqrt(1.0 + t * t);
                    const double s = c * t;
#pragma unroll
                    for (int m = 0; m < 8; ++m) {
                        const int row = sub + 16 * m;
                        if (m < nrow_it && row < np) {
                            cp[row] = c * vp[m] - s * vq[m];
                            cq[row] = s * vp[m] + c * vq[m];
                        }
                    }
                    rotated = 1;
                }
            }
            __syncthreads();
        }
        if (rotated && sub == 0) sh.flag[0] = 1;
        __syncthreads();
        ++sweeps;
        const int any = sh.flag[0];
        __syncthreads();
        if (!any) break;
    }
    // column norms (= eigenvalues) and descending rank
    for (int col = tid >> 4; col < np; col += EIG_THREADS >> 4) {
        const double* cp = sh.Gs + (size_t)col * np;
        double s = 0.0;
        for (int row = sub; row < np; row += 16) s += cp[row] * cp[row];
        s = sum16(s);
        if (sub == 0) sh.nrm[col] = sqrt(s);
    }
    __syncthreads();
    if (tid < np) {
        const double me = sh.nrm[tid];
        int rk = 0;
        for (int j = 0; j < np; ++j) {
            const double o = sh.nrm[j];
            rk += (o > me || (o == me && j < tid)) ? 1 : 0;
        }
        sh.rank[tid] = rk;
    }
    __syncthreads();
    return sweeps;
}

__device__ __forceinline__ EigShared carve(double* smem, int np) {
    EigShared sh;
    sh.Gs = smem;
    sh.nrm = smem + (size_t)np * np;
    sh.rank = (int*)(sh.nrm + np);
    sh.flag = sh.rank + np;
    return sh;
}

// Full Jacobi solve of G (global, n x n): lam_out[0..n) descending, eigenvectors
// E[row*ldE + rank] for rank < kcols.  Returns sweeps.
__device__ int jacobi_solve(const double* __restrict__ G, int n, double* smem, double* lam_out, double* E, int ldE,
                            int kcols) {
    const int tid = threadIdx.x;
    const int np = (n + 1) & ~1;
    EigShared sh = carve(smem, np);
    __syncthreads();
    for (int i = tid; i < np * np; i += EIG_THREADS) {
        const int col = i / np, row = i - col * np;
        sh.Gs[i] = (row < n && col < n) ? G[(size_t)row * n + col] : 0.0;
    }
    __syncthreads();
    const int sweeps = jacobi_core(sh, np);
    if (tid < np) {
        const int rk = sh.rank[tid];
        if (rk < n) lam_out[rk] = sh.nrm[tid];
    }
    for (int i = tid; i < np * np; i += EIG_THREADS) {
        const int col = i / np, row = i - col * np;
        const int rk = sh.rank[col];
        if (rk < kcols && row < n) {
            const double nr = sh.nrm[col];
            E[(size_t)row * ldE + rk] = nr > 0.0 ? sh.Gs[i] / nr : 0.0;
        }
    }
    __syncthreads();
    return sweeps;
}

// =====================================================================================
// Tridiagonal path
// =====================================================================================
struct TriShared {
    double* Vs;    // packed Householder vectors: v_i[c], c > i, at voff(i) + c - i - 1   (<= 8128)
    double* xs;    // [2][128]
    double* ps;    // [2][128]
    double* de;    // [128][2]: (d_j, e_{j-1}^2)
    double* es;    // [128]
    double* taus;  // [128]
    double* lam;   // [32]
    double* Z;     // [128][32]  k_eig_tri: published rows (k_eig_fin carves its own layout, see FinShared)
    double* Ub;    // [128][32]
    double* misc;  // [64] scalars + [128] raw row
};
__device__ __forceinline__ TriShared tri_carve(double* smem) {
    TriShared t;
    t.Vs = smem;
    t.xs = t.Vs + 8128;
    t.ps = t.xs + 256;
    t.de = t.ps + 256;
    t.es = t.de + 256;
    t.taus = t.es + 128;
    t.lam = t.taus + 128;
    t.Z = t.lam + 32;
    t.Ub = t.Z + 128 * 32;
    t.misc = t.Ub + 128 * 32;         // total 17440 + 192 <= EIG_LDS_DOUBLES
    return t;
}
__device__ __forceinline__ int voff(int i, int n) { return i * (n - 1) - (i * (i - 1)) / 2; }



// =====================================================================================
// k_eig_tri
// =====================================================================================
// Layout constants of the register-resident matrix: TRI_T threads, thread (r, q) owns NE elements of
// row r as column pairs c = 2q + 2*QN*k + {0,1}; a wave holds RPW consecutive rows.
constexpr int TRI_T = 1024;
constexpr int EIG_TAIL_N = 128;       // trailing block the blocked solver hands to the one-workgroup reduction (= BT_TAIL there)
constexpr int QN = TRI_T / 128;          // threads per row
constexpr int NE = 16384 / TRI_T;        // elements per thread
constexpr int NP = NE / 2;               // column pairs per thread
constexpr int RPW = 64 / QN;             // rows per wave
__device__ __forceinline__ double sum_q(double x) { return QN == 8 ? sum8(x) : sum4(x); }

// position of reflector column j (0..15 inside its block of 16) in row c of a dense reflector block kept in LDS by the
// merged kernel: pairs of columns are rotated by the row, so that the 64 rows a wave writes for ONE reflector land in 16
// bank pairs instead of one (row stride = 16 doubles = all 64 lanes on the same bank otherwise)
__device__ __forceinline__ int vsw(int c, int j) { return c * 16 + ((j + 2 * c) & 15); }

// The tridiagonalisation proper.  DENSE = false: reflectors packed in t.Vs (k_eig_tri publishes them afterwards);
// DENSE = true: written straight into the dense 16-column blocks Vd[8][128][16] (rotated, vsw) the back-transformation
// of the same workgroup reads - the merged kernel k_eig_trivec.
template <bool DENSE>
__device__ __forceinline__ void tri_core(const double* __restrict__ G, const int n, const TriShared t, double* __restrict__ Vd,
                                         unsigned long long* stamps) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (stamps && tid == 0) {
        stamps[0] = __builtin_amdgcn_s_memrealtime();
        stamps[6] = __builtin_readcyclecounter();
    }
    const int r = tid / QN, q = tid % QN;
    // ---- load: thread (r, q) owns the column pairs c = 2q + 8k + {0,1}, k = 0..15 of row r ----
    // (A[2k+h], 32 doubles); its LDS operands are 16-byte reads, broadcast across the 16 rows of a wave.
    double A[NE];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = 2 * q + 2 * QN * k + h;
            A[2 * k + h] = (r < n && c < n) ? G[(size_t)r * n + c] : 0.0;
        }
    }
    if (tid < 256) {
        t.xs[tid] = 0.0;
        t.ps[tid] = 0.0;
    }
    if constexpr (DENSE) {
        // what the standalone kernel's publication step guarantees: nothing but zeros beyond the problem
        if (tid < 128) {
            t.es[tid] = 0.0;
            t.taus[tid] = 0.0;
        }
        for (int i = tid; i < (129 - n) * 128; i += TRI_T) {        // reflector columns n-1 .. 127 are never written
            const int jr = n - 1 + (i >> 7), c = i & 127;
            Vd[(jr >> 4) * 2048 + vsw(c, jr & 15)] = 0.0;
        }
    }
    __syncthreads();
    // ---- Householder tridiagonalisation (dsytd2, full storage) -----------------------------
    // Step i, reflector v_i and tau_i published in LDS:
    //  (a) every live wave forms its rows of p = tau*A*v.                                  | barrier
    //  (b) live waves: the scalar v^T p (redundantly per wave) and the rank-2 update
    //          A[r][c] -= v_r p_c + (a2 v_r + w_r) v_c,      w = p + a2 v,
    //      after which the lanes that own row i+2 publish it for the next step, while ONE helper
    //      wave - a retired one as soon as there is one - rebuilds the updated row i+1 from LDS
    //      operands (row i+1 as published one step earlier, p, v) with the very same two FMAs
    //      (64 lanes x 2 columns), and runs the norm -> sqrt -> reciprocal chain of reflector i+1:
    //      that chain no longer waits for anybody's register update.                       | barrier
    // Both phases are bound by LDS operand traffic (every lane reads v and p at its 16 columns), so
    // the 16-column groups that are already entirely in the finished part (columns <= i) are
    // skipped: a wave-uniform switch with fall-through keeps the register indices static.
    double* xrb = t.Z;                             // [2][128] row j before the update of step j-1, parity j&1
    auto publish_row = [&](int row) {
        if (r == row) {
            double* x = xrb + (row & 1) * 128;
#pragma unroll
            for (int k = 0; k < NP; ++k) *(double2*)&x[2 * q + 2 * QN * k] = make_double2(A[2 * k], A[2 * k + 1]);
        }
    };
    [[maybe_unused]] int dslot = -1;
#ifdef MPST_TRI_DEBUG
#define DBG(j) do { if (dslot >= 0) stamps[dslot + (j)] = __builtin_readcyclecounter(); } while (0)
#else
#define DBG(j) do { } while (0)
#endif
    int pend_i = -1;                               // reflector this wave still has to write out (wave-uniform)
    double pend_v0 = 0.0, pend_v1 = 0.0, pend_d = 0.0, pend_e = 0.0;
    auto finish_reflector = [&](int i, double x0, double x1, double di, double al) {
        // all 64 lanes of one wave; lane holds columns c0 = lane, c1 = lane + 64 of row i; di = x[i] and
        // al = x[i+1] arrive as wave-uniform values (formed from LDS scalars, no cross-lane reads on the chain)
        double* vbn = t.xs + (i & 1) * 128;
        const int c0 = lane, c1 = lane + 64;
        const double xm0 = c0 >= i + 2 ? x0 : 0.0, xm1 = c1 >= i + 2 ? x1 : 0.0;     // the part that is scaled
        const double e0 = c0 == i + 1 ? 1.0 : 0.0, e1 = c1 == i + 1 ? 1.0 : 0.0;     // the unit entry of v
        const double s = wave_sum_mfma(xm0 * xm0 + xm1 * xm1);
        // Branch-free: rsq + two Heron steps instead of the IEEE sqrt chain, reciprocals from the hardware
        // seed + 2 Newton steps.  A row whose tail is below 1e-140 in norm is treated as already reduced
        // (s == 0 path of dlarfg); squared norms above 1e280 would need rescaling and are left to the
        // verification in k_eig_fin (-> Jacobi fallback).
        const double xx = fma(al, al, s);
        const bool nz = s != 0.0 && xx > 1e-280;
        const double rs = __builtin_amdgcn_rsq(nz ? xx : 1.0);
        double nrm = xx * rs;
        const double hrs = 0.5 * rs;
        nrm = fma(fma(-nrm, nrm, xx), hrs, nrm);
        nrm = fma(fma(-nrm, nrm, xx), hrs, nrm);
        const double bneg = copysign(nrm, al);                 // -beta
        const double ib = frcp(bneg), is = frcp(al + bneg);
        const double beta = nz ? -bneg : al;
        const double tau = nz ? (bneg + al) * ib : 0.0;        // (beta - al) / beta
        const double scale = nz ? is : 0.0;                    // 1 / (al - beta)
        const double v0 = fma(xm0, scale, e0), v1 = fma(xm1, scale, e1);
        vbn[c0] = v0;
        vbn[c1] = v1;
        if (lane == 0) t.taus[i] = tau;
        // what only the later kernels read (the stored reflector, d_i, e_i) is written by flush_reflector()
        // while this wave idles in the next step's first phase - off the chain the barrier waits for
        pend_i = i;
        pend_v0 = v0;
        pend_v1 = v1;
        pend_d = di;
        pend_e = beta;
    };
    auto flush_reflector = [&]() {
        if (pend_i >= 0) {
            const int i = pend_i, c0 = lane, c1 = lane + 64;
            if constexpr (DENSE) {
                double* vb = Vd + (i >> 4) * 2048;                     // all 128 rows: zeros above the start and beyond n
                vb[vsw(c0, i & 15)] = pend_v0;
                vb[vsw(c1, i & 15)] = pend_v1;
            } else {
                const int off = voff(i, n) - i - 1;
                if (c0 > i && c0 < n) t.Vs[off + c0] = pend_v0;
                if (c1 > i && c1 < n) t.Vs[off + c1] = pend_v1;
            }
            if (lane == 0) {
                t.de[2 * i] = pend_d;
                t.es[i] = pend_e;
            }
            pend_i = -1;
        }
    };
    publish_row(0);
    publish_row(1);
    __syncthreads();
    if (wave == 0) finish_reflector(0, xrb[lane], xrb[lane + 64], xrb[0], xrb[1]);
    __syncthreads();
    // One step, with the number K0 of finished 16-column groups as a compile-time constant: the main
    // loop below is cut into 8 "eras" of 16 steps, each running its own branch-free copy of the body.
    auto step = [&](auto K0c, const int i_) {
        constexpr int K0 = decltype(K0c)::value;
        const int i = __builtin_amdgcn_readfirstlane(i_);      // keep the step index (and all it feeds) scalar
        const double* vb = t.xs + (i & 1) * 128;
        double* p = t.ps + (i & 1) * 128;
        const bool live = (wave * RPW + RPW - 1) > i;    // this wave still owns trailing rows
        // the most recently retired wave: its SIMD has just lost a live wave
        const int helper = i >= RPW - 1 ? (i + 1) / RPW - 1 : (TRI_T / 64 - 1);
        const double tau = t.taus[i];
        double2 vv[NP];
#ifdef MPST_TRI_DEBUG
        dslot = (stamps && i == 60 && lane == 0) ? (wave == 12 ? 16 : wave == 6 ? 24 : wave == 7 ? 32 : -1) : -1;
#endif
        DBG(0);
#ifdef MPST_TRI_STEPPROF
        if (stamps && tid == 0) stamps[64 + i] = __builtin_readcyclecounter();
#endif
        if (!live) flush_reflector();              // a helper wave is idle in this phase
        if (live) {
#pragma unroll
            for (int k = K0; k < NP; ++k) vv[k] = *(const double2*)&vb[2 * q + 2 * QN * k];
            double accx[2] = {0.0, 0.0}, accy[2] = {0.0, 0.0};
#pragma unroll
            for (int k = K0; k < NP; ++k) {
                accx[k & 1] = fma(A[2 * k], vv[k].x, accx[k & 1]);
                accy[k & 1] = fma(A[2 * k + 1], vv[k].y, accy[k & 1]);
            }
            const double acc = sum_q((accx[0] + accy[0]) + (accx[1] + accy[1]));
            if (q == 0) p[r] = (r > i) ? tau * acc : 0.0;
        } else if (wave * RPW + RPW - 1 == i) {
            // this wave's rows have just retired: clear their p entries in both buffers for good
            if (q == 0) {
                t.ps[r] = 0.0;
                t.ps[128 + r] = 0.0;
            }
        }
        DBG(1);
        __syncthreads();
        DBG(2);
        const int c0 = lane, c1 = lane + 64;
        if (wave == helper && i + 1 < n - 1) {
            __builtin_amdgcn_s_setprio(3);
            const int j = i + 1;
            const double* xr = xrb + (j & 1) * 128;
            const double p0 = p[c0], p1 = p[c1], v0 = vb[c0], v1 = vb[c1], a0 = xr[c0], a1 = xr[c1];
            // wave-uniform operands of the two leading entries of the updated row (v_j = 1 exactly)
            const double pj = p[j], pj1 = p[j + 1], vj1 = vb[j + 1], aj = xr[j], aj1 = xr[j + 1];
            const double dot = wave_sum_mfma(p0 * v0 + p1 * v1);
            const double a2 = -0.5 * tau * dot;
            const double g = a2 + (pj + a2);                    // a2 v_j + (p_j + a2 v_j) with v_j = 1
            // same two roundings per entry as the register update: (a - 1 p) first, then the fma with g
            const double x0 = fma(-g, v0, a0 - p0);
            const double x1 = fma(-g, v1, a1 - p1);
            const double di = (aj - pj) - g;
            const double al = fma(-g, vj1, aj1 - pj1);
            DBG(6);
            finish_reflector(j, x0, x1, di, al);
            __builtin_amdgcn_s_setprio(0);
            DBG(7);
        }
        if (live) {
            const double pr = p[r], vr = vb[r];
            const double dot = wave_sum_mfma(p[c0] * vb[c0] + p[c1] * vb[c1]);
            const double a2 = -0.5 * tau * dot;
            const double wr = pr + a2 * vr;
            const double g = a2 * vr + wr;
            DBG(3);
            double2 pv[NP];
#pragma unroll
            for (int k = K0; k < NP; ++k) pv[k] = *(const double2*)&p[2 * q + 2 * QN * k];
#pragma unroll
            for (int k = K0; k < NP; ++k) {
                A[2 * k] = fma(-vr, pv[k].x, A[2 * k]);
                A[2 * k + 1] = fma(-vr, pv[k].y, A[2 * k + 1]);
                A[2 * k] = fma(-g, vv[k].x, A[2 * k]);
                A[2 * k + 1] = fma(-g, vv[k].y, A[2 * k + 1]);
            }
            if (i + 2 < n - 1) publish_row(i + 2);
            flush_reflector();                     // early steps: the helper is still a live wave
            DBG(4);
        }
        __syncthreads();
        DBG(5);
    };
    static_assert(NP == 8 && QN == 8, "the era loops assume 8 column groups of 16 columns per thread");
    {
        int i = 0;
#define TRI_ERA(K) for (; i < n - 1 && ((i + 1) >> 4) == K; ++i) step(std::integral_constant<int, K>{}, i);
        TRI_ERA(0) TRI_ERA(1) TRI_ERA(2) TRI_ERA(3) TRI_ERA(4) TRI_ERA(5) TRI_ERA(6) TRI_ERA(7)
#undef TRI_ERA
    }
    flush_reflector();
    {   // last diagonal element
        double* x = t.xs + ((n - 1) & 1) * 128;
        __syncthreads();
        if (r == n - 1) {
#pragma unroll
            for (int k = 0; k < NP; ++k) *(double2*)&x[2 * q + 2 * QN * k] = make_double2(A[2 * k], A[2 * k + 1]);
        }
        __syncthreads();
        if (tid == 0) {
            t.de[2 * (n - 1)] = x[n - 1];
            t.es[n - 1] = 0.0;
        }
        __syncthreads();
    }
    if (stamps && tid == 0) {
        stamps[1] = __builtin_amdgcn_s_memrealtime();
        stamps[7] = __builtin_readcyclecounter();
    }
    // ---- e^2 and Gershgorin bounds ----------------------------------------------------------
    if (tid < n) t.de[2 * tid + 1] = tid > 0 ? t.es[tid - 1] * t.es[tid - 1] : 0.0;
    if (wave == 0) {
        double gl = 1e300, gu = -1e300;
        for (int j = lane; j < n; j += 64) {
            const double a = j > 0 ? fabs(t.es[j - 1]) : 0.0, b = j < n - 1 ? fabs(t.es[j]) : 0.0;
            const double dj = t.de[2 * j];
            gl = fmin(gl, dj - a - b);
            gu = fmax(gu, dj + a + b);
        }
        gl = -wave_max(-gl);
        gu = wave_max(gu);
        if (lane == 0) {
            const double w = fmax(fabs(gl), fabs(gu));
            const double pad = 2.0 * n * 2.3e-16 * w + 1e-300;
            t.misc[0] = gl - pad;
            t.misc[1] = gu + pad;
            t.misc[2] = w;
        }
    }
    __syncthreads();
}

// The last EIG_TAIL_N steps of a LARGER tridiagonalisation (bond tensors beyond 128 x 128, mpst_eig_blocked.hip): once the
// trailing matrix of the multi-workgroup reduction is 128 x 128 it fits one CU's registers, where a Householder step costs
// 0.9 us instead of the 2.6 us of a step that exchanges vectors between 32 workgroups.  Gt is that trailing block with
// every earlier reflector applied; the reflectors, tau, d and e of its reduction are written at offset m = n - nt into the
// arrays the blocked path's later kernels read (Vall row j = reflector j, entries j .. n-1).
__global__ __launch_bounds__(TRI_T) void k_eig_tail(View v, int lid, int going_left, int rawn, const double* __restrict__ Gt, int ld,
                                                          double* __restrict__ Vall, double* __restrict__ dd, double* __restrict__ ee,
                                                          double* __restrict__ tau, const int32_t* abort_flag, const int32_t* skip) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    if (abort_flag && *(const volatile int32_t*)abort_flag != 0) return;
    if (skip && *(const volatile int32_t*)skip != 0) return;          // the subspace solver's result stands (mpst_eig_subspace.inl)
    const EigProblem pb = resolve(v, lid, going_left, nullptr, rawn, 0);
    const int n = pb.n, tid = threadIdx.x;
    if (n <= EIG_TAIL_N) return;                   // the multi-workgroup reduction did all the steps itself
    const int nt = EIG_TAIL_N, m = n - nt;
    TriShared t = tri_carve(smem);
    tri_core<false>(Gt, nt, t, nullptr, nullptr);
    if (tid < nt) {
        dd[m + tid] = t.de[2 * tid];
        ee[m + tid] = tid < nt - 1 ? t.es[tid] : 0.0;
        tau[m + tid] = tid < nt - 1 ? t.taus[tid] : 0.0;
    }
    for (int idx = tid; idx < (nt - 1) * nt; idx += TRI_T) {
        const int i = idx / nt, c = idx - i * nt;
        if (c >= i) Vall[(int64_t)(m + i) * ld + m + c] = c == i ? 0.0 : t.Vs[voff(i, nt) + c - i - 1];
    }
}

__global__ __launch_bounds__(TRI_T) void k_eig_tri(View v, int lid, int going_left, const double* rawG, int rawn,
                                                         int rawalg, double* __restrict__ ws,
                                                         unsigned long long* stamps) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const EigProblem pb = resolve(v, lid, going_left, rawG, rawn, rawalg);
    const int tid = threadIdx.x;
    if (!pb.tri) {
        if (tid == 0) ws[WS_MISC + 3] = 0.0;
        return;
    }
    const int n = pb.n;
    TriShared t = tri_carve(smem);
    tri_core<false>(pb.G, n, t, nullptr, stamps);
    // ---- publish T, the reflectors and the bounds ------------------------------------------------
    if (tid < 256) ws[WS_DE + tid] = tid < 2 * n ? t.de[tid] : 0.0;
    if (tid < 128) {
        ws[WS_ES + tid] = tid < n ? t.es[tid] : 0.0;
        ws[WS_TAU + tid] = tid < n - 1 ? t.taus[tid] : 0.0;
    }
    // reflectors in the blocked dense layout the back-transformation wants (k_eig_vec): 16 per block,
    // zero above the start of each reflector and beyond n
    for (int i = tid; i < 8 * 128 * 16; i += TRI_T) {
        const int j = i & 15, c = (i >> 4) & 127, jr = (i >> 11) * 16 + j;
        ws[WS_VS + i] = (jr < n - 1 && c > jr && c < n) ? t.Vs[voff(jr, n) + c - jr - 1] : 0.0;
    }
    if (tid == 0) {
        ws[WS_MISC + 0] = t.misc[0];
        ws[WS_MISC + 1] = t.misc[1];
        ws[WS_MISC + 2] = t.misc[2];
        ws[WS_MISC + 3] = 1.0;
    }
}

// =====================================================================================
// k_eig_vec: one workgroup per eigenvalue
// =====================================================================================
// Everything k_eig_vec does for eigenvalue k.  MERGED = false: T and the reflectors come from the global workspace
// (written by k_eig_tri); MERGED = true: they are already in this workgroup's LDS (tri_core<true>, rotated rows).
// HALVES (k_eig_trivec_bm): TWO eigenpairs at a time, threads [0, 256) one, [256, 512) the other - the same code on a scratch set each
// (T, the reflectors and their compact-WY factors are shared and read-only; what both halves write there they write with the same
// values), every barrier workgroup-wide and met by both halves in step (no barrier sits under a data-dependent branch).  Each pair's
// arithmetic is exactly that of the one-pair kernels: the same bits.
constexpr int VEC_SCRATCH = 128 * 4 + 8 + 128 + 8 + 24 + 24;      // Dm .. Bb
constexpr int VEC_LDS_DOUBLES = 16384 + 272 + 128 * 6 + 144 + 48 + 2048 + 16;
template <bool MERGED, bool HALVES = false>
__device__ __forceinline__ void vec_core(const EigProblem& pb, const int k, double* smem, int* cnt_s_, double* red_s_, int* arg_s_,
                                         double* __restrict__ ws, unsigned long long* stamps, const double lo_in,
                                         const double hi_in, const double tnorm_in) {
    auto vix = [](int c, int j) { return MERGED ? vsw(c, j) : c * 16 + j; };
    const int n = pb.n;
    const int half = HALVES ? ((int)threadIdx.x >> 8) : 0;
    const int tid = HALVES ? ((int)threadIdx.x & (VEC_THREADS - 1)) : (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int* cnt_s = cnt_s_ + 8 * half;
    double* red_s = red_s_ + 8 * half;
    int* arg_s = arg_s_ + 8 * half;
    double* Vd = smem;              // [8][128][16] reflectors (dense blocks of 16)
    double* de = Vd + 16384;        // [272]: 128 (d, e^2) pairs + padding pairs for the 8-step Sturm groups
    double* es = de + 272;          // [128]
    double* taus = es + 128;        // [128]
    double* Dm = half ? smem + VEC_LDS_DOUBLES : taus + 128;        // [128] D^-_i  (bottom-up pivots)
    double* Ub = Dm + 128;          // [128] U_i
    double* Dp = Ub + 128;          // [128] D_i    (top-down pivots)
    double* Lb = Dp + 128;          // [128] L_i
    double* z = Lb + 128 + 8;       // [128] with 8 slack entries before and after
    double* Fb = z + 128 + 8;       // [24] rescaled group-boundary minors, top-down
    double* Bb = Fb + 24;           // [24] bottom-up
    double* Tb = taus + 128 + VEC_SCRATCH;      // [8][16][16] triangular factors of the blocked reflectors (behind the first scratch set)
    const bool st = stamps && k == 0 && tid == 0;
#ifdef MPST_TRI_DEBUG
#define VDBG(j) do { if (st) stamps[40 + (j)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define VDBG(j) do { } while (0)
#endif
    if (st) stamps[2] = __builtin_amdgcn_s_memrealtime();
    // ---- stage T (needed now) and request the reflectors (needed last) ----------------------------
    // T goes through registers and is consumed at once; the 128 KB of reflectors are then requested
    // as direct global->LDS loads that stay in flight during the whole bisection (every barrier up to
    // the point where they are needed is an LDS-only barrier, and no ordinary global load result is
    // consumed in between - either would drain them).
    double lo = lo_in, hi = hi_in;
    double tnorm = tnorm_in;
    if constexpr (!MERGED) {
        lo = ws[WS_MISC + 0];
        hi = ws[WS_MISC + 1];
        tnorm = ws[WS_MISC + 2];
        if ((tid >> 1) < n) de[tid] = ws[WS_DE + tid];
        if (tid < 128) {
            es[tid] = ws[WS_ES + tid];
            taus[tid] = ws[WS_TAU + tid];
        }
    }
    // pad T to 1 + a multiple of 8 rows with decoupled rows (e^2 = 0) whose diagonal lies above every
    // abscissa: they add no sign change and let the Sturm loop run in whole groups of 8
    {
        const double dpad = hi + (hi - lo) + 1.0;
        if ((tid >> 1) >= n) de[tid] = (tid & 1) ? 0.0 : dpad;
        if (tid < 16) de[256 + tid] = (tid & 1) ? 0.0 : dpad;
    }
    if constexpr (!MERGED) {
#pragma unroll
        for (int m = 0; m < 32; ++m)
            glds16(ws + WS_VS + 2 * (tid + m * VEC_THREADS), Vd + 2 * (wave * 64 + m * VEC_THREADS));
    }
    lds_barrier();
    VDBG(0);
    // ---- multisection for the k-th largest eigenvalue ---------------------------------------------------
    const int target = n - 1 - (pb.pair ? 2 * k : k);       // ascending index (pair mode: the upper one of the k-th pair)
    if (n >= 24) {
        // Two-sided Sturm count: the inertia of T - x is that of its twisted factorisation - negative top-down pivots
        // of rows 0..kk-1, negative bottom-up pivots of rows n-1..kk+1, and the sign of the twisted pivot at row kk.
        // A pair of lanes runs the two halves of one abscissa (the same division-free recurrence on T and on the
        // reversed T), so a round walks n/2 rows instead of n: 128 abscissae x 8 rounds cost 60 % of 256 x 7.
        double* deR = Dm;                               // [n] pairs of the reversed matrix (the pivot arrays are free until the bisection is over)
        const int kk = ((n >> 1) & ~7) + 1;             // top half: 1 + a multiple of 8 rows
        for (int m = tid; m < n; m += VEC_THREADS) {
            deR[2 * m] = de[2 * (n - 1 - m)];
            deR[2 * m + 1] = m >= 1 ? de[2 * (n - m) + 1] : 0.0;
        }
        lds_barrier();
        const bool bottom = tid & 1;
        const double* arr = bottom ? deR : de;
        const int rows = bottom ? n - 1 - kk : kk;
        const double dk = de[2 * kk], e2a = de[2 * kk + 1], e2b = de[2 * (kk + 1) + 1];     // d_kk, e^2_{kk-1}, e^2_kk
        constexpr int NPT = VEC_THREADS / 2;
        for (int it = 0; it < 8; ++it) {
            const double h = (hi - lo) * (1.0 / (NPT + 1));
            const double xq = lo + h * ((tid >> 1) + 1);
            double p1, p2;
            int cnt = sturm_half(arr, rows, xq, p1, p2);
            // the partner's count and end values (lane ^ 1): quad_perm [1, 0, 3, 2]
            const int ocnt = __builtin_amdgcn_update_dpp(0, cnt, 0xB1, 0xF, 0xF, true);
            const double o1 = dpp_mov<0xB1>(p1), o2 = dpp_mov<0xB1>(p2);
            // top lane: (p1, p2) = (P_{kk-1}, P_{kk-2}), partner's = (Q_{kk+1}, Q_{kk+2}); bottom lane the other way round
            const double P1 = bottom ? o1 : p1, P2 = bottom ? o2 : p2, Q1 = bottom ? p1 : o1, Q2 = bottom ? p2 : o2;
            // gamma_kk = (d - x) - e2a P2 / P1 - e2b Q2 / Q1, signed through P1 Q1 (the two pairs carry their own scales)
            const double g = fma(dk - xq, P1 * Q1, -fma(e2a * P2, Q1, e2b * Q2 * P1));
            const int sg = ((hi32(g) ^ hi32(P1) ^ hi32(Q1)) >> 31) & 1;
            cnt += ocnt + sg;
            const unsigned long long bal = __ballot(cnt <= target) & 0x5555555555555555ull;
            int* slot = cnt_s + 4 * (it & 1);                 // parity-indexed: one barrier per round
            if (lane == 0) slot[wave] = __popcll(bal);
            lds_barrier();
            const int jj = slot[0] + slot[1] + slot[2] + slot[3];
            const double nlo = jj > 0 ? lo + h * jj : lo;
            const double nhi = jj < NPT ? lo + h * (jj + 1) : hi;
            lo = nlo;
            hi = nhi;
        }
    } else {
        for (int it = 0; it < TRI_NSTEP; ++it) {
            const double h = (hi - lo) * (1.0 / (VEC_THREADS + 1));
            const double xq = lo + h * (tid + 1);
            const int cnt = sturm_count(de, n, xq);
            const unsigned long long b = __ballot(cnt <= target);
            if (lane == 0) cnt_s[wave] = __popcll(b);
            lds_barrier();
            const int jj = cnt_s[0] + cnt_s[1] + cnt_s[2] + cnt_s[3];
            lds_barrier();
            const double nlo = jj > 0 ? lo + h * jj : lo;
            const double nhi = jj < VEC_THREADS ? lo + h * (jj + 1) : hi;
            lo = nlo;
            hi = nhi;
        }
    }
    const double lamk = 0.5 * (lo + hi);
    VDBG(1);
    // the reflectors have landed by now; make them visible to every wave
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    if (st) stamps[3] = __builtin_amdgcn_s_memrealtime();
    // ---- eigenvector of T: twisted factorisation --------------------------------------------------
    // The pivots of the top-down and bottom-up factorisations of T - lambda are ratios of consecutive
    // leading / trailing principal minors, D_j = P_{j+1}/P_j and D^-_j = Q_j/Q_{j+1}.  The minors obey
    // the division-free three-term recurrence of the Sturm sequence (3 dependent fp64 ops per row instead
    // of a reciprocal + Newton chain), so the serial part runs on two waves (forward / backward, every
    // lane redundantly, lane 0 stores the (minor, previous minor) pairs, rescaled together every 8
    // rows) and the 2n divisions, the safeguard and L_i = e_i/D_i, U_i = e_i/D^-_{i+1} are done in
    // parallel afterwards.
    const double pivmin = 1e-290 + 1e-30 * tnorm;
    double* F1 = Dp;                // [<= 137] P_j at F1[j]          (aliases Dp, Lb until the quotients are taken)
    double* B1 = Dm + 8;            // [-7 .. 128] Q_j at B1[j]       (aliases Dm, Ub)
    {
        const double2* de2 = (const double2*)de;
        if (wave == 1) {
            double pm = 1.0, pc = de[0] - lamk;
            if (lane == 0) {
                F1[0] = pm;
                F1[1] = pc;
            }
            int b = 0;
            for (int j = 1; j < n; j += 8, ++b) {     // rows n.. are the padding rows: harmless
                double2 v[8];
                double o[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = de2[j + u];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const double t = v[u].x - lamk;
                    const double pn = fma(t, pc, -v[u].y * pm);
                    o[u] = pn;
                    pm = pc;
                    pc = pn;
                }
                int e = __builtin_amdgcn_frexp_exp(pc);
                if (pc == 0.0) e = __builtin_amdgcn_frexp_exp(pm);
                pc = __builtin_amdgcn_ldexp(pc, -e);
                pm = __builtin_amdgcn_ldexp(pm, -e);
                if (lane == 0) {
#pragma unroll
                    for (int u = 0; u < 8; u += 2) *(double2*)&F1[j + 1 + u] = make_double2(o[u], o[u + 1]);
                    Fb[b] = pc;                       // P_{j+8} on the scale of the next group
                }
            }
        } else if (wave == 0) {
            double qm = 1.0, qc = de[2 * (n - 1)] - lamk;          // Q_n, Q_{n-1}
            if (lane == 0) {
                B1[n] = qm;
                B1[n - 1] = qc;
            }
            double e2 = de[2 * (n - 1) + 1];                        // e_{n-2}^2 couples rows n-2, n-1
            int b = 0;
            for (int j = n - 2; j >= 0; j -= 8, ++b) {
                double2 v[8];
                double o[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = de2[j - u];        // j - u < 0 reads the tail of Vs: finite, unused
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const double t = v[u].x - lamk;
                    const double qn = fma(t, qc, -e2 * qm);         // Q_j = (d_j - l) Q_{j+1} - e_j^2 Q_{j+2}
                    o[u] = qn;
                    e2 = v[u].y;                                    // e_{j-1}^2 for the next row up
                    qm = qc;
                    qc = qn;
                }
                int e = __builtin_amdgcn_frexp_exp(qc);
                if (qc == 0.0) e = __builtin_amdgcn_frexp_exp(qm);
                qc = __builtin_amdgcn_ldexp(qc, -e);
                qm = __builtin_amdgcn_ldexp(qm, -e);
                if (lane == 0) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) B1[j - u] = o[u];
                    Bb[b] = qc;                       // Q_{j-7} on the scale of the next group
                }
            }
        } else if (half == 0) {
            // (HALVES: the factors are shared and formed IN PLACE - Gram matrix, then its triangular inverse over it - so one half forms them)
            // Waves 2, 3 (idle otherwise): compact-WY factors of the reflector blocks for the
            // back-transformation.  Q_b = H(16b) ... H(16b+15) = I - V_b T_b V_b^T with
            // T_b^-1 = diag(1/tau) + striu(V_b^T V_b); the Gram matrix comes from the fp64 MFMA
            // (A and B operand are the same register: lane l feeds row 4s + (l>>4), reflector l&15),
            // the inverse of the triangular matrix column by column (lane = column, tau-multiplied
            // form, so tau = 0 - an identity reflector - needs no special case).
            const int jl = lane & 15, q4 = lane >> 4;
            for (int bi = 0; bi < 4; ++bi) {
                const int b = 2 * bi + (wave - 2);
                d4 acc = {0.0, 0.0, 0.0, 0.0};
                const double* vb = Vd + (size_t)b * 2048;
                for (int c = 16 * b + q4; c < 128; c += 16) {
                    const double a0 = vb[vix(c + 0, jl)], a1 = vb[vix(c + 4, jl)], a2 = vb[vix(c + 8, jl)], a3 = vb[vix(c + 12, jl)];
                    acc = mfma_f64(a0, a0, acc);
                    acc = mfma_f64(a1, a1, acc);
                    acc = mfma_f64(a2, a2, acc);
                    acc = mfma_f64(a3, a3, acc);
                }
                double* S = Tb + b * 256;               // S[i][j'] at i*16 + j', i = q4 + 4r
#pragma unroll
                for (int r = 0; r < 4; ++r) S[(q4 + 4 * r) * 16 + jl] = acc[r];
            }
            // all 4 blocks of this wave at once: lane (column jl, block 2*q4 + wave-2)
            {
                const int b = 2 * q4 + (wave - 2);
                const double* S = Tb + b * 256;
                const double* tb = taus + 16 * b;
                double tcol[16];
#pragma unroll
                for (int i = 15; i >= 0; --i) {
                    double sum = 0.0;
#pragma unroll
                    for (int kk = i + 1; kk < 16; ++kk) sum = fma(S[i * 16 + kk], tcol[kk], sum);
                    const double ti = tb[i];
                    tcol[i] = (i == jl) ? ti : -ti * sum;
                }
                double* T = Tb + b * 256;
#pragma unroll
                for (int i = 0; i < 16; ++i) T[i * 16 + jl] = tcol[i];
            }
        }
    }
    VDBG(2);
    __syncthreads();
    VDBG(3);
    // quotients + safeguard (mirrors the sequential rule: a pivot below pivmin is replaced by -pivmin and
    // the next pivot of that recurrence is recomputed from the replaced value).  A minor that is the
    // last of its group of 8 was rescaled before the next group used it: the rescaled copy is in Fb/Bb.
    auto quotF = [&](int j) {      // D_j = P_{j+1} / P_j
        const double den = (j >= 9 && (j & 7) == 1) ? Fb[(j - 9) >> 3] : F1[j];
        return F1[j + 1] / den;
    };
    auto quotB = [&](int j) {      // D^-_j = Q_j / Q_{j+1}
        const int m = n - 2 - j;   // position in the bottom-up order
        const double den = (m >= 8 && (m & 7) == 0) ? Bb[(m >> 3) - 1] : B1[j + 1];
        return B1[j] / den;
    };
    double dp_ = 0.0, dm_ = 0.0, dsh = 0.0;
    if (tid < n) {
        dsh = de[2 * tid] - lamk;
        dp_ = quotF(tid);
        dm_ = tid < n - 1 ? quotB(tid) : dsh;
        if (tid > 0 && !(fabs(quotF(tid - 1)) >= pivmin)) dp_ = dsh + de[2 * tid + 1] / pivmin;
        if (tid < n - 1) {
            const double rnext = tid + 1 < n - 1 ? quotB(tid + 1) : de[2 * (n - 1)] - lamk;
            if (!(fabs(rnext) >= pivmin)) dm_ = dsh + de[2 * (tid + 1) + 1] / pivmin;
        }
        if (!(fabs(dp_) >= pivmin)) dp_ = -pivmin;
        if (!(fabs(dm_) >= pivmin)) dm_ = -pivmin;
    }
    __syncthreads();                 // all minors consumed: Dm/Ub/Dp/Lb may be overwritten
    // twist index r = argmin |gamma_i|, gamma_i = D_i + D^-_i - (d_i - lambda)
    {
        double g = 1e300;
        int gi = 0;
        if (tid < n) {
            Dm[tid] = dm_;
            g = fabs(dp_ + dm_ - dsh);
            gi = tid;
        }
        // wave-level argmin (ties -> smallest index), then across the 4 waves
        const double gmin = -wave_max(-g);
        const unsigned long long m = __ballot(g == gmin);
        const int first = __ffsll((long long)m) - 1;
        const int idx = __builtin_amdgcn_readlane(gi, first);
        if (lane == 0) {
            red_s[wave] = gmin;
            arg_s[wave] = idx;
        }
    }
    __syncthreads();
    int rb = arg_s[0];
    {
        double gm = red_s[0];
        for (int w = 1; w < VEC_THREADS / 64; ++w)
            if (red_s[w] < gm) {
                gm = red_s[w];
                rb = arg_s[w];
            }
    }
    rb = __builtin_amdgcn_readfirstlane(rb);
    if (tid < 128) {
        // L_i = e_i / D_i, U_i = e_i / D^-_{i+1}; zero beyond n-2 and in the (now dead) Dp array, so the
        // substitution below can fetch whole groups of 8 on either side without masks
        const bool in = tid < n - 1;
        const double e = in ? es[tid] : 0.0;
        Lb[tid] = in ? e / dp_ : 0.0;
        Ub[tid] = in ? e / Dm[tid + 1] : 0.0;
        Dp[tid] = 0.0;
    }
    __syncthreads();
    VDBG(4);
    // substitution: downwards from the twist on wave 0, upwards on wave 1 (serial products; operands are
    // fetched 8 at a time, every lane computes, lane 0 stores - z has 8 slack entries on both sides)
    if (wave == 0) {
        double zc = 1.0, nrm = 1.0;
        if (lane == 0) z[rb] = 1.0;
        for (int i = rb - 1; i >= 0; i -= 8) {
            double l[8], o[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) l[u] = Lb[i - u];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                zc = -l[u] * zc;
                o[u] = zc;
                nrm = fma(zc, zc, nrm);
            }
            if (lane == 0) {
#pragma unroll
                for (int u = 0; u < 8; ++u) z[i - u] = o[u];
            }
        }
        if (lane == 0) red_s[4] = nrm;
    } else if (wave == 1) {
        double zc = 1.0, nrm = 0.0;
        for (int i = rb; i < n - 1; i += 8) {
            double l[8], o[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) l[u] = Ub[i + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                zc = -l[u] * zc;
                o[u] = zc;
                nrm = fma(zc, zc, nrm);
            }
            if (lane == 0) {
#pragma unroll
                for (int u = 0; u < 8; ++u) z[i + u + 1] = o[u];
            }
        }
        if (lane == 0) red_s[5] = nrm;
    }
    __syncthreads();
    VDBG(5);
    const double sc = 1.0 / sqrt(red_s[4] + red_s[5]);
    // normalise; residual ||T z - lambda z||_inf (verification)
    double zi = 0.0, ri = 0.0;
    if (tid < n) {
        zi = z[tid] * sc;
        const double zp = tid > 0 ? z[tid - 1] * sc : 0.0, zn = tid < n - 1 ? z[tid + 1] * sc : 0.0;
        ri = fabs((de[2 * tid] - lamk) * zi + (tid > 0 ? es[tid - 1] * zp : 0.0) + (tid < n - 1 ? es[tid] * zn : 0.0));
    }
    ri = wave_max(ri);
    __syncthreads();
    if (tid < 128) z[tid] = zi;                    // zi = 0 for rows >= n: the blocked back-transformation reads all 128
    if (lane == 0) red_s[wave] = ri;
    __syncthreads();
    if (st) stamps[4] = __builtin_amdgcn_s_memrealtime();
    // ---- back-transformation z <- H(0) H(1) ... H(n-2) z on wave 0, 16 reflectors per step ----------
    // z <- z - V_b (T_b (V_b^T z)), b = last block .. 0.  Three short phases per block, each bound by
    // its instruction count on one wave: (i) lane (j, h) sums V[c][j] z[c] over the rows c = 16b+h+4m,
    // the 4 partial sums of a column meet through two permlane swaps; (ii) u = T_b y, 4 terms per lane
    // + the same swap-sum; (iii) every lane updates its two rows with the 16 u's.
    if (wave == 0) {
        const int jl = lane & 15, q4 = lane >> 4;
        const int c0 = lane, c1 = lane + 64;
        double z0 = z[c0], z1 = z[c1];                 // rows >= n hold zeros
        double* yb = Fb;                               // [16]  (Fb, Bb are free now)
        double* ub = Bb;                               // [16]
        for (int b = (n - 2) >> 4; b >= 0; --b) {
            const double* vb = Vd + (size_t)b * 2048;
            // (i) y = V_b^T z
            double part = 0.0;
            for (int c = 16 * b + q4; c < 128; c += 16) {
                const double w0 = vb[vix(c + 0, jl)], w1 = vb[vix(c + 4, jl)], w2 = vb[vix(c + 8, jl)], w3 = vb[vix(c + 12, jl)];
                const double x0 = z[c], x1 = z[c + 4], x2 = z[c + 8], x3 = z[c + 12];
                part = fma(w0, x0, part);
                part = fma(w1, x1, part);
                part = fma(w2, x2, part);
                part = fma(w3, x3, part);
            }
            const double yj = row4_sum(part);
            if (q4 == 0) yb[jl] = yj;
            // (ii) u = T_b y
            const double2 ya = *(const double2*)&yb[4 * q4], yc = *(const double2*)&yb[4 * q4 + 2];
            const double2 ta = *(const double2*)&Tb[b * 256 + jl * 16 + 4 * q4], tc = *(const double2*)&Tb[b * 256 + jl * 16 + 4 * q4 + 2];
            const double ui = row4_sum(fma(ta.x, ya.x, ta.y * ya.y) + fma(tc.x, yc.x, tc.y * yc.y));
            if (q4 == 0) ub[jl] = ui;
            // (iii) z -= V_b u
            double2 u2[8];
#pragma unroll
            for (int m = 0; m < 8; ++m) u2[m] = *(const double2*)&ub[MERGED ? 2 * ((m - c0) & 7) : 2 * m];   // rotated rows: pair slot m of row c holds reflector pair (m - c) & 7 (c1 = c0 + 64: the same)
            if (b < 4) {                                // rows 0..63 are above every reflector of blocks 4..7
                const double2* r0 = (const double2*)&vb[c0 * 16];
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const double2 w = r0[m];
                    z0 = fma(-w.x, u2[m].x, z0);
                    z0 = fma(-w.y, u2[m].y, z0);
                }
                z[c0] = z0;
            }
            const double2* r1 = (const double2*)&vb[c1 * 16];
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const double2 w = r1[m];
                z1 = fma(-w.x, u2[m].x, z1);
                z1 = fma(-w.y, u2[m].y, z1);
            }
            z[c1] = z1;
        }
        const int kc = pb.pair ? 2 * k : k;
        double* Zk = ws + WS_Z + kc;                   // [c][KS] layout: k_eig_fin reads it linearly
        const int ks = kstride(pb.K0);
        if (c0 < n) Zk[c0 * ks] = z0;
        if (c1 < n) Zk[c1 * ks] = z1;
        const double resk = fmax(fmax(red_s[0], red_s[1]), fmax(red_s[2], red_s[3]));
        if (lane == 0) {
            ws[WS_LAM + kc] = lamk;
            ws[WS_RES + kc] = resk;
        }
        if (pb.pair) {
            // the partner of u = (u_re, u_im) in the double eigenspace: J u = (-u_im, u_re), i.e. i times the complex vector
            const int nc = n >> 1;
            if (c0 < n) Zk[c0 * ks + 1] = c0 < nc ? -z[c0 + nc] : z[c0 - nc];
            if (c1 < n) Zk[c1 * ks + 1] = c1 < nc ? -z[c1 + nc] : z[c1 - nc];
            if (lane == 0) {
                ws[WS_LAM + kc + 1] = lamk;
                ws[WS_RES + kc + 1] = resk;
            }
        }
    }
    if (st) stamps[5] = __builtin_amdgcn_s_memrealtime();
}

__global__ __launch_bounds__(VEC_THREADS) void k_eig_vec(View v, int lid, int going_left, int rawn, int rawalg,
                                                         double* __restrict__ ws, unsigned long long* stamps) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ int cnt_s[8];
    __shared__ double red_s[8];
    __shared__ int arg_s[8];
    const EigProblem pb = resolve(v, lid, going_left, nullptr, rawn, rawalg);
    const int k = blockIdx.x;
    if (!pb.tri || k >= eig_nvec(pb)) return;
    vec_core<false>(pb, k, smem, cnt_s, red_s, arg_s, ws, stamps, 0.0, 0.0, 0.0);
}

// =====================================================================================
// k_eig_trivec: tridiagonalisation + one eigenpair in ONE launch, no exchange between workgroups
// =====================================================================================
// Every eigenvector workgroup repeats the whole tridiagonalisation for itself (same code, same data, same order of
// operations: the same bits in all of them) and goes on with its own eigenvalue: T and the reflectors never leave LDS.
// That removes a dependent launch (its cold start and first round trip to memory), the publication of 128 KB of
// reflectors by one workgroup and their reload by 32, at the price of 31 CUs doing redundant work on a chip that has
// nothing else to do at this point of the chain.  LDS layout = k_eig_vec's; the tridiagonalisation's scratch rows
// live where the compact-WY factors (Tb) go later.  After the reduction the 12 waves the eigenvector part does not
// need retire (a terminated wave no longer counts at s_barrier).
__device__ __forceinline__ void trivec_body(const View& v, int lid, int going_left, const double* rawG, int rawn,
                                            int rawalg, double* __restrict__ ws, unsigned long long* stamps) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ int cnt_s[8];
    __shared__ double red_s[8];
    __shared__ int arg_s[8];
    const EigProblem pb = resolve(v, lid, going_left, rawG, rawn, rawalg);
    const int k = blockIdx.x, tid = threadIdx.x;
    if (!pb.tri) {
        if (k == 0 && tid == 0) ws[WS_MISC + 3] = 0.0;
        return;
    }
    if (k >= eig_nvec(pb)) return;
    double* Vd = smem;
    double* de = Vd + 16384;
    double* Tb = de + 272 + 128 * 6 + 8 + 128 + 8 + 24 + 24;      // k_eig_vec's carve: see vec_core
    TriShared t{};
    t.de = de;
    t.es = de + 272;
    t.taus = t.es + 128;
    t.xs = Tb;
    t.ps = Tb + 256;
    t.Z = Tb + 512;             // [2][128] published rows
    t.misc = Tb + 768;
    unsigned long long* st = k == 0 ? stamps : nullptr;
    tri_core<true>(pb.G, pb.n, t, Vd, st);
    const double lo = t.misc[0], hi = t.misc[1], tnorm = t.misc[2];
    if (k == 0 && tid == 0) {
        ws[WS_MISC + 0] = lo;
        ws[WS_MISC + 1] = hi;
        ws[WS_MISC + 2] = tnorm;
        ws[WS_MISC + 3] = 1.0;
    }
    if (tid >= VEC_THREADS) return;
    vec_core<true>(pb, k, smem, cnt_s, red_s, arg_s, ws, st, lo, hi, tnorm);
}
__global__ __launch_bounds__(TRI_T) void k_eig_trivec(View v, int lid, int going_left, const double* rawG, int rawn,
                                                            int rawalg, double* __restrict__ ws, unsigned long long* stamps) {
    trivec_body(v, lid, going_left, rawG, rawn, rawalg, ws, stamps);
}
__device__ __noinline__ void vec_core_call(const EigProblem& pb, const int k, double* smem, int* cnt_s, double* red_s, int* arg_s,
                                           double* __restrict__ ws, unsigned long long* stamps, const double lo, const double hi, const double tnorm) {
    vec_core<true, true>(pb, k, smem, cnt_s, red_s, arg_s, ws, stamps, lo, hi, tnorm);
}
// The same for SEVERAL eigenpairs per workgroup: blockIdx.x, + gridDim.x, ... - one reduction, then the vector phase once per pair
// (30 us each).  For mpst_sweep_batch beyond 8 fits: 32 workgroups per fit repeat the 114 us reduction 32 times, K x 32 > 256
// workgroups run in rounds; 16 (8) workgroups per fit with 2 (4) pairs each keep 16 (32) fits in one round.  A body of its own:
// the single-fit kernels keep their register allocation.  Round 6: the vector phase runs for TWO pairs at a time, on the two halves
// of the 512 threads that stay (vec_core<.., HALVES>): 4 pairs cost two vector phases, not four (32 fits: 247 -> 190 us per bond).
__device__ __forceinline__ void trivec_body_multi(const View& v, int lid, int going_left, const double* rawG, int rawn, int rawalg,
                                                  double* __restrict__ ws, unsigned long long* stamps) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ int cnt_s[16];
    __shared__ double red_s[16];
    __shared__ int arg_s[16];
    const EigProblem pb = resolve(v, lid, going_left, rawG, rawn, rawalg);
    const int k = blockIdx.x, tid = threadIdx.x;
    if (!pb.tri) {
        if (k == 0 && tid == 0) ws[WS_MISC + 3] = 0.0;
        return;
    }
    const int nvec = eig_nvec(pb);
    if (k >= nvec) return;
    double* Vd = smem;
    double* de = Vd + 16384;
    double* Tb = de + 272 + 128 * 6 + 8 + 128 + 8 + 24 + 24;
    TriShared t{};
    t.de = de;
    t.es = de + 272;
    t.taus = t.es + 128;
    t.xs = Tb;
    t.ps = Tb + 256;
    t.Z = Tb + 512;
    t.misc = Tb + 768;
    unsigned long long* st = k == 0 ? stamps : nullptr;
    tri_core<true>(pb.G, pb.n, t, Vd, st);
    const double lo = t.misc[0], hi = t.misc[1], tnorm = t.misc[2];
    if (k == 0 && tid == 0) {
        ws[WS_MISC + 0] = lo;
        ws[WS_MISC + 1] = hi;
        ws[WS_MISC + 2] = tnorm;
        ws[WS_MISC + 3] = 1.0;
    }
    if (tid >= 2 * VEC_THREADS) return;
    const int half = tid >> 8;
    for (int kk = k; kk < nvec; kk += 2 * (int)gridDim.x) {
        // this half's pair; a half without one repeats its partner's (the same values to the same places) and keeps the barriers in step
        const int mine = kk + half * (int)gridDim.x < nvec ? kk + half * (int)gridDim.x : kk;
        vec_core_call(pb, mine, smem, cnt_s, red_s, arg_s, ws, mine == 0 && half == 0 ? st : nullptr, lo, hi, tnorm);
        lds_barrier();                  // (the scratch of the vector phase is reused by the next pairs)
    }
}
// raw problem behind a gate word (v.label_site, 0 = leave at once): the Rayleigh-Ritz problem of the subspace solver
// (mpst_eig_subspace.inl), which is enqueued whether or not the bond at hand is attempted.  A kernel of its own: the headline
// kernel above keeps its register allocation.
__global__ __launch_bounds__(TRI_T) void k_eig_trivec_g(View v, const double* rawG, int rawn, int rawalg, double* __restrict__ ws) {
    if (*(const volatile int32_t*)v.label_site == 0) return;
    trivec_body(v, 0, 0, rawG, rawn, rawalg, ws, nullptr);
}
// K independent fits per launch (blockIdx.z): see mpst_fused.hip.  The eigensolver reads a dozen scalar fields of the View: they
// are copied into a local one (scalar registers after SROA) - through a reference into global memory the register-starved
// reflector loop of trivec_body spilled 60 bytes per lane more and ran 55 us longer.
__device__ __forceinline__ View eig_fields(const View& s) {
    View v{};
    v.T = s.T; v.d = s.d; v.C = s.C; v.chi_max = s.chi_max; v.cap = s.cap;
    v.chi = s.chi; v.label_site = s.label_site;
    v.gram = s.gram; v.lam = s.lam; v.E = s.E; v.eig_ws = s.eig_ws; v.sc = s.sc;
    v.rescale_after = s.rescale_after; v.svd_alg = s.svd_alg; v.cutoff = s.cutoff; v.zw = s.zw;
    return v;
}
// (rawG / rawn / rawalg stay run-time arguments - always null / 0 here: with them folded away the register allocation of the
// reflector loop changes and spills 60 bytes per lane more)
__global__ __launch_bounds__(TRI_T) void k_eig_trivec_b(const View* __restrict__ vs, int lid, int going_left, const double* rawG, int rawn, int rawalg) {
    const View v = eig_fields(vs[blockIdx.z]);
    trivec_body(v, lid, going_left, rawG, rawn, rawalg, v.eig_ws, v.sc->eig_stamps);
}

__global__ __launch_bounds__(TRI_T) void k_eig_trivec_bm(const View* __restrict__ vs, int lid, int going_left, const double* rawG, int rawn, int rawalg) {
    const View v = eig_fields(vs[blockIdx.z]);
    trivec_body_multi(v, lid, going_left, rawG, rawn, rawalg, v.eig_ws, v.sc->eig_stamps);
}

// =====================================================================================
// k_eig_fin: truncation, verification, re-orthonormalisation, publication (or Jacobi)
// =====================================================================================

// tridiagonal branch of k_eig_fin: candidates (already in registers) -> LDS, verification, polish, publication
template <int KS>
__device__ bool fin_tri(double* smem, const double* zin, int n, int nk, double res_in, double tnorm_in, double* Eout, int ldE) {
    const int tid = threadIdx.x;
    FinShared t = fin_carve<KS>(smem);
    // eigenvectors of values the truncation rule discards are never looked at: those are the
    // clustered, noise-level ones for which the twisted factorisation loses orthogonality
#pragma unroll
    for (int m = 0; m < 128 * KS / EIG_THREADS; ++m) {
        const int i = tid + m * EIG_THREADS;
        const int c = i / KS, kk = i % KS;
        t.Z[i] = (c < n && kk < nk) ? zin[m] : 0.0;
    }
    if (tid < KS) t.misc[32 + tid] = tid < nk ? res_in : 0.0;
    if (tid == 0) t.misc[2] = tnorm_in;
    __syncthreads();
    const bool done = verify_and_polish<KS>(t, n, nk);
    if (done) {
        for (int i = tid; i < n * nk; i += EIG_THREADS) {
            const int c = i / nk, kk = i - c * nk;
            Eout[(size_t)c * ldE + kk] = t.Z[c * KS + kk];
        }
    }
    __syncthreads();
    return done;
}

__device__ __forceinline__ void fin_body(const View& v, int lid, int going_left, const double* rawG, int rawn,
                                         int rawalg, double* __restrict__ ws, double* rawlam,
                                         double* rawE, int32_t* rawinfo, unsigned long long* stamps) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double lam_s[MAX_DIM + 2];
    __shared__ double red[16];
    const EigProblem pb = resolve(v, lid, going_left, rawG, rawn, rawalg);
    const int tid = threadIdx.x;
    const int n = pb.n, K0 = pb.K0, nspec = pb.nspec;
    const bool raw = rawn > 0;
    double* Eout = raw ? rawE : v.E;
    const int ldE = raw ? n : v.cap;
    if (stamps && tid == 0) stamps[8] = __builtin_amdgcn_s_memrealtime();
    // everything this kernel reads from global memory is requested up front (one round trip instead of three
    // dependent ones): the diagonal of G, the flag / eigenvalues / residuals and the K0 candidate vectors
    const double gdiag = tid < n ? pb.G[(size_t)tid * n + tid] : 0.0;
    const bool tri = pb.tri && ws[WS_MISC + 3] == 1.0;
    double lam_in = 0.0, res_in = 0.0, tnorm_in = 0.0;
    constexpr int NZL = 128 * TRI_KMAX / EIG_THREADS;
    double zin[NZL];
    const int ks = kstride(K0);
    if (pb.tri) {
        lam_in = tid < K0 ? ws[WS_LAM + tid] : 0.0;
        res_in = tid < TRI_KMAX ? ws[WS_RES + tid] : 0.0;
        tnorm_in = ws[WS_MISC + 2];
#pragma unroll
        for (int m = 0; m < NZL; ++m) zin[m] = (m < NZL / 2 || ks == 64) ? ws[WS_Z + tid + m * EIG_THREADS] : 0.0;
    }
    // trace = ||bt_new||_F^2 (fixed order)
    double tr = wave_sum(gdiag);
    if ((tid & 63) == 0) red[tid >> 6] = tr;
    __syncthreads();
    tr = 0.0;
    for (int i = 0; i < EIG_THREADS / 64; ++i) tr += red[i];
    __syncthreads();
    if (pb.pair) tr *= 0.5;             // trace of the complex matrix
    const double inv = (!raw && v.rescale_after) ? 1.0 / sqrt(tr) : 1.0;
    const double cutoff = raw ? -1.0 : v.cutoff;
    // NDTensors truncate! (relative cutoff) on P = lambda*inv^2.  The weight beyond the first K values is
    // trace - sum(first K).  Evaluated redundantly by every thread.  Returns the number of (real) vectors to keep: in pair
    // mode both members of every kept pair.
    auto truncate = [&](const double* lam, int K) -> int {
        return (pb.pair ? 2 : 1) * truncate_rule(lam, K, nspec, tr, inv * inv, cutoff, pb.pair);
    };
    int sweeps = 0, nk = K0;
    bool done = false;
    if (tri) {
        if (tid < K0) lam_s[tid] = lam_in;
        __syncthreads();
        nk = truncate(lam_s, K0);
        if (raw && (rawalg & 8)) {
            // Rayleigh-Ritz problem of the subspace solver: a rank-deficient block (growth phase of a fit) has a cluster of zero
            // eigenvalues whose vectors nobody reads - only the pairs above 1e-13 of the largest are verified and delivered
            int cnt = 1;
            while (cnt < K0 && lam_s[cnt] > 1e-13 * lam_s[0]) ++cnt;
            nk = cnt;
        }
        done = ks == 32 ? fin_tri<32>(smem, zin, n, nk, res_in, tnorm_in, Eout, ldE)
                        : fin_tri<64>(smem, zin, n, nk, res_in, tnorm_in, Eout, ldE);
    }
    const bool fell_back = !done && pb.tri;
    if (!done) {
        const int kc = raw ? n : K0;
        sweeps = jacobi_solve(pb.G, n, smem, lam_s, Eout, ldE, kc);
        if (sweeps == 0) sweeps = 1;
        __syncthreads();
        nk = truncate(lam_s, K0);
    }
    __syncthreads();
    if (raw) {
        const int nl = done ? K0 : n;
        if (tid < nl) rawlam[tid] = lam_s[tid];
        if (tid == 0) *rawinfo = done ? -1 : (pb.tri ? 0 : sweeps);
    } else {
        const int st = pb.pair ? 2 : 1;       // the spectrum and the counts the caller sees are those of the complex matrix
        if (tid < K0 / st) v.lam[tid] = lam_s[st * tid];
        if (tid == 0) {
            bool bad = !(tr == tr) || tr > 1e300;
            for (int i = 0; i < K0; ++i) {
                const double P = lam_s[i] * inv * inv;
                if (!(P == P) || P > 1e300) bad = true;
            }
            nk /= st;
            v.sc->n_keep = nk;
            v.sc->n_spec = K0 / st;
            v.sc->bt_norm2 = tr;
            v.sc->inv_norm = inv;
            v.sc->eig_sweeps = sweeps;
            v.sc->eig_sweeps_total += sweeps;
            if (fell_back) v.sc->eig_fallbacks += 1;
            if (bad || sweeps >= EIG_MAX_SWEEPS) v.sc->status = MPST_ERR_SVD;
            v.chi[lid + 1] = nk;
        }
    }
    if (stamps && tid == 0) stamps[9] = __builtin_amdgcn_s_memrealtime();
}
__global__ __launch_bounds__(EIG_THREADS) void k_eig_fin(View v, int lid, int going_left, const double* rawG, int rawn,
                                                         int rawalg, double* __restrict__ ws, double* rawlam,
                                                         double* rawE, int32_t* rawinfo, unsigned long long* stamps) {
    fin_body(v, lid, going_left, rawG, rawn, rawalg, ws, rawlam, rawE, rawinfo, stamps);
}
__global__ __launch_bounds__(EIG_THREADS) void k_eig_fin_b(const View* __restrict__ vs, int lid, int going_left) {
    const View v = eig_fields(vs[blockIdx.z]);
    fin_body(v, lid, going_left, nullptr, 0, 0, v.eig_ws, nullptr, nullptr, nullptr, v.sc->eig_stamps);
}

__global__ __launch_bounds__(EIG_THREADS) void k_eig_fin_g(View v, const double* rawG, int rawn, int rawalg, double* __restrict__ ws, double* rawlam,
                                                           double* rawE, int32_t* rawinfo) {
    if (*(const volatile int32_t*)v.label_site == 0) return;
    fin_body(v, 0, 0, rawG, rawn, rawalg, ws, rawlam, rawE, rawinfo, nullptr);
}
// raw-mode helper: clear the outputs of the test entry point
__global__ void k_eig_clear(double* lam, double* E, int n) {
    for (int i = threadIdx.x; i < n * n; i += blockDim.x) E[i] = 0.0;
    if (threadIdx.x < (unsigned)n) lam[threadIdx.x] = 0.0;
}

static size_t eig_lds_bytes() { return (size_t)EIG_LDS_DOUBLES * sizeof(double); }
static size_t tri_lds_bytes() { return (size_t)(8128 + 256 * 3 + 128 * 2 + 32 + 192 + 64) * sizeof(double); }
static size_t vec_lds_bytes() { return (size_t)VEC_LDS_DOUBLES * sizeof(double); }

size_t eig_workspace_doubles() { return WS_TOTAL; }
hipError_t eig_init_attrs(int device) {
    static std::atomic<unsigned long long> done{0};
    if (device >= 0 && device < 64 && (done.load(std::memory_order_acquire) >> device) & 1ull) return hipSuccess;
    hipError_t e;
    if ((e = hipFuncSetAttribute((const void*)k_eig_fin, hipFuncAttributeMaxDynamicSharedMemorySize, (int)eig_lds_bytes())) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)k_eig_tri, hipFuncAttributeMaxDynamicSharedMemorySize, (int)eig_lds_bytes())) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)k_eig_tail, hipFuncAttributeMaxDynamicSharedMemorySize, (int)eig_lds_bytes())) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)k_eig_vec, hipFuncAttributeMaxDynamicSharedMemorySize, (int)vec_lds_bytes())) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)k_eig_trivec, hipFuncAttributeMaxDynamicSharedMemorySize, (int)vec_lds_bytes())) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)k_eig_trivec_b, hipFuncAttributeMaxDynamicSharedMemorySize, (int)vec_lds_bytes())) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)k_eig_trivec_bm, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(vec_lds_bytes() + VEC_SCRATCH * sizeof(double)))) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)k_eig_fin_b, hipFuncAttributeMaxDynamicSharedMemorySize, (int)eig_lds_bytes())) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)k_eig_trivec_g, hipFuncAttributeMaxDynamicSharedMemorySize, (int)vec_lds_bytes())) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)k_eig_fin_g, hipFuncAttributeMaxDynamicSharedMemorySize, (int)eig_lds_bytes())) != hipSuccess) return e;
    if (device >= 0 && device < 64) done.fetch_or(1ull << device, std::memory_order_release);
    return hipSuccess;
}

// MPST_EIG_SPLIT=1: the three-kernel chain (k_eig_tri, k_eig_vec, k_eig_fin) instead of k_eig_trivec + k_eig_fin
bool eig_merged() {
    static const bool m = [] {
        const char* e = getenv("MPST_EIG_SPLIT");
        return !(e && e[0] && e[0] != '0');
    }();
    return m;
}

void launch_eig(const View& v, int lid, int going_left, int stage, hipStream_t s) {
    unsigned long long* st = v.sc ? v.sc->eig_stamps : nullptr;
    const dim3 gvec(v.chi_max < TRI_KMAX ? (v.chi_max < 32 ? 32 : v.chi_max) : TRI_KMAX);
    if (stage == 0 && eig_merged())
        hipLaunchKernelGGL(k_eig_trivec, gvec, dim3(TRI_T), vec_lds_bytes(), s, v, lid, going_left, (const double*)nullptr, 0, 0,
                           v.eig_ws, st);
    else if (stage == 0)
        hipLaunchKernelGGL(k_eig_tri, dim3(1), dim3(TRI_T), eig_lds_bytes(), s, v, lid, going_left,
                           (const double*)nullptr, 0, 0, v.eig_ws, st);
    else if (stage == 1) {
        if (!eig_merged())
            hipLaunchKernelGGL(k_eig_vec, gvec, dim3(VEC_THREADS), vec_lds_bytes(), s, v, lid, going_left, 0, 0, v.eig_ws, st);
    } else
        hipLaunchKernelGGL(k_eig_fin, dim3(1), dim3(EIG_THREADS), eig_lds_bytes(), s, v, lid, going_left,
                           (const double*)nullptr, 0, 0, v.eig_ws, (double*)nullptr, (double*)nullptr, (int32_t*)nullptr, st);
}

// the merged chain for K fits of one shape (mpst_sweep_batch): stage 0 k_eig_trivec_b, stage 2 k_eig_fin_b
void launch_eig_b(const View& v, const View* vs, int K, int lid, int going_left, int stage, hipStream_t s) {
    const dim3 gvec(v.chi_max < TRI_KMAX ? (v.chi_max < 32 ? 32 : v.chi_max) : TRI_KMAX, 1, K);
    // more workgroups than CUs (one workgroup per CU: 157 KB of LDS): several eigenpairs per workgroup instead of rounds of workgroups
    // that repeat the reduction (MPST_EIG_BM=0: never)
    static const bool bm_on = [] { const char* e = getenv("MPST_EIG_BM"); return !(e && e[0] == '0'); }();
    const int rounds = ((int)gvec.x * K + 255) / 256;
    if (stage == 0 && bm_on && rounds > 1) {
        const int nb = std::max(8, ((int)gvec.x + rounds - 1) / rounds);
        hipLaunchKernelGGL(k_eig_trivec_bm, dim3(nb, 1, K), dim3(TRI_T), vec_lds_bytes() + VEC_SCRATCH * sizeof(double), s, vs, lid, going_left, (const double*)nullptr, 0, 0);
        return;
    }
    if (stage == 0) hipLaunchKernelGGL(k_eig_trivec_b, gvec, dim3(TRI_T), vec_lds_bytes(), s, vs, lid, going_left, (const double*)nullptr, 0, 0);
    else hipLaunchKernelGGL(k_eig_fin_b, dim3(1, 1, K), dim3(EIG_THREADS), eig_lds_bytes(), s, vs, lid, going_left);
}

void launch_eig_tail(const View& v, int lid, int going_left, int rawn, const double* Gt, int ld, double* Vall, double* dd, double* ee,
                     double* tau, const int32_t* abort_flag, const int32_t* skip, hipStream_t s) {
    hipLaunchKernelGGL(k_eig_tail, dim3(1), dim3(TRI_T), eig_lds_bytes(), s, v, lid, going_left, rawn, Gt, ld, Vall, dd, ee, tau, abort_flag, skip);
}

// the TRI_KMAX largest eigenpairs of a symmetric n x n matrix (n <= 128) on the device, no host synchronisation; gate: device word,
// 0 = every launch leaves at once.  E: n x n row-major, column k = vector k; lam descending.
void launch_eig_raw_gated(const double* G, int n, double* lam, double* E, int32_t* info, double* ws, const int32_t* gate, hipStream_t s) {
    View v{};
    v.label_site = const_cast<int32_t*>(gate);
    const int alg = 8;
    const int K = n < TRI_KMAX ? (n < 32 ? 32 : n) : TRI_KMAX;
    hipLaunchKernelGGL(k_eig_trivec_g, dim3(K), dim3(TRI_T), vec_lds_bytes(), s, v, G, n, alg, ws);
    hipLaunchKernelGGL(k_eig_fin_g, dim3(1), dim3(EIG_THREADS), eig_lds_bytes(), s, v, G, n, alg, ws, lam, E, info);
}

void launch_eig_raw(const double* G, int n, int alg, double* lam, double* E, int32_t* info, double* ws, hipStream_t s) {
    View v{};
    hipLaunchKernelGGL(k_eig_clear, dim3(1), dim3(256), 0, s, lam, E, n);
    if (eig_merged())
        hipLaunchKernelGGL(k_eig_trivec, dim3(RAW_KMAX), dim3(TRI_T), vec_lds_bytes(), s, v, 0, 0, G, n, alg, ws,
                           (unsigned long long*)nullptr);
    else {
        hipLaunchKernelGGL(k_eig_tri, dim3(1), dim3(TRI_T), eig_lds_bytes(), s, v, 0, 0, G, n, alg, ws,
                           (unsigned long long*)nullptr);
        hipLaunchKernelGGL(k_eig_vec, dim3(RAW_KMAX), dim3(VEC_THREADS), vec_lds_bytes(), s, v, 0, 0, n, alg, ws,
                           (unsigned long long*)nullptr);
    }
    hipLaunchKernelGGL(k_eig_fin, dim3(1), dim3(EIG_THREADS), eig_lds_bytes(), s, v, 0, 0, G, n, alg, ws, lam, E, info,
                       (unsigned long long*)nullptr);
}


// =====================================================================================
// Bond tensors beyond 128 x 128 (d*chi_max in (128, DIM_LIMIT]): the robust slow path
// =====================================================================================
// The register/LDS-resident solver above holds n <= 128; larger Gram matrices (the reference's documented d = 8..12, chi_max = 37..64
// runs, docs/src/hyperparameters.md:65,127,241-245) are solved by the blocked tridiagonal path (mpst_eig_blocked.hip) or the subspace
// solver in front of it.  What stands BEHIND those - a bond whose on-device verification fails, MPST_BIG_EIG=jacobi - is this: a
// one-sided (Hestenes) Jacobi iteration on the columns of G, the algorithm of jacobi_core above spread over workgroups: a round of the
// round-robin tournament is one launch (one workgroup per column pair, columns in global memory), a sweep is np - 1 rounds, and a
// small kernel after every sweep records whether anything was rotated; later launches leave at once when the matrix has converged.
// Slow (milliseconds) and unconditionally robust: it is the reference's gesdd -> gesvd -> recursive chain's last resort
// (RealRealHighDimension.jl:146-203), and the reason no vendor solver is linked.  The problem is always solved at the CAPACITY size
// ncap = d*cap with the live n x n Gram matrix zero-padded (bond dimensions live on the device): the padding adds exact zero eigenvalues
// below the spectrum of the positive semi-definite G and leaves the eigenvectors of its non-zero eigenvalues untouched.
constexpr int BIGJ_MAX_SWEEPS = 160;     // (graded spectra converge slowly: 67 sweeps for the cluster test of tests/test_gpu_bigbond.py)
struct BigEig {
    int ncap = 0;
    hipStream_t stream = nullptr;
    double *W = nullptr;        // [ncap][ncap] working columns (column-major: column p at W + p * ncap)
    double *A = nullptr;        // [ncap][ncap] eigenvectors, column i = vector of D[i] (ascending), as dsyevd leaves them
    double *D = nullptr;        // [ncap] eigenvalues, ascending
    double *nrm = nullptr;      // [ncap] column norms
    int32_t* info = nullptr;    // [4]: 0 = converged flag (1: stop), 1 = rotations in the current sweep, 2 = sweeps used, 3 = result (0 ok)
};

__global__ __launch_bounds__(256) void k_big_prep(View v, int lid, int going_left, const double* rawG, int rawn, double* A, int ncap, int32_t* info) {
    const EigProblem pb = resolve(v, lid, going_left, rawG, rawn, 0);
    const int n = pb.n;
    if (blockIdx.x == 0 && threadIdx.x < 4) info[threadIdx.x] = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)ncap * ncap; i += (int64_t)gridDim.x * 256) {
        const int r = (int)(i / ncap), c = (int)(i - (int64_t)r * ncap);
        A[i] = (r < n && c < n) ? pb.G[(size_t)r * n + c] : 0.0;       // symmetric: row- and column-major coincide
    }
}
// deterministic block-wide sums of three values (256 threads)
__device__ __forceinline__ void block_sum3(double& a, double& b, double& c, double* red /* [12] */) {
    a = wave_sum(a);
    b = wave_sum(b);
    c = wave_sum(c);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        red[w] = a;
        red[4 + w] = b;
        red[8 + w] = c;
    }
    __syncthreads();
    a = (red[0] + red[1]) + (red[2] + red[3]);
    b = (red[4] + red[5]) + (red[6] + red[7]);
    c = (red[8] + red[9]) + (red[10] + red[11]);
}
// one round of the tournament: workgroup kk rotates the column pair pair_of(r, kk)
// (scale: the largest column norm of G.  Two columns that are both below 1e-14 of it hold nothing but the rounding of the large ones - their
// eigenvalues are beyond what a double-precision G resolves - and are left alone: rotating noise against noise never converges)
__global__ __launch_bounds__(256) void k_bigjac_round(double* __restrict__ W, int ncap, int r, int32_t* info, const double* __restrict__ scale) {
    __shared__ double red[12];
    if (*(const volatile int32_t*)info != 0) return;          // converged in an earlier sweep
    const double tiny2 = 1e-28 * scale[0] * scale[0];
    int p, q;
    pair_of(r, (int)blockIdx.x, ncap, p, q);
    double* cp = W + (size_t)p * ncap;
    double* cq = W + (size_t)q * ncap;
    double vp[4], vq[4];
    double app = 0.0, aqq = 0.0, apq = 0.0;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int row = threadIdx.x + 256 * m;
        vp[m] = row < ncap ? cp[row] : 0.0;
        vq[m] = row < ncap ? cq[row] : 0.0;
        app += vp[m] * vp[m];
        aqq += vq[m] * vq[m];
        apq += vp[m] * vq[m];
    }
    block_sum3(app, aqq, apq, red);
    if (fabs(apq) > 2.5e-15 * sqrt(app * aqq) && app * aqq > 0.0 && (app > tiny2 || aqq > tiny2)) {
        const double zeta = (aqq - app) / (2.0 * apq);
        const double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / sqrt(1.0 + t * t);
        const double sn = c * t;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int row = threadIdx.x + 256 * m;
            if (row < ncap) {
                cp[row] = c * vp[m] - sn * vq[m];
                cq[row] = sn * vp[m] + c * vq[m];
            }
        }
        if (threadIdx.x == 0) info[1] = 1;                     // (benign race: every writer writes 1)
    }
}
__global__ __launch_bounds__(256) void k_bigjac_scale(const double* __restrict__ nrm, int ncap, double* __restrict__ scale) {
    __shared__ double red[4];
    double m = 0.0;
    for (int j = threadIdx.x; j < ncap; j += 256) m = fmax(m, nrm[j]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) scale[0] = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}
__global__ void k_bigjac_check(int32_t* info) {
    if (threadIdx.x == 0 && info[0] == 0) {
        info[2] += 1;
        if (info[1] == 0) info[0] = 1;
        info[1] = 0;
    }
}
// column norms (= eigenvalues), ascending order, normalised vectors: D and A as dsyevd would leave them
__global__ __launch_bounds__(256) void k_bigjac_norms(const double* __restrict__ W, int ncap, double* __restrict__ nrm) {
    __shared__ double red[12];
    const double* cp = W + (size_t)blockIdx.x * ncap;
    double s = 0.0, z0 = 0.0, z1 = 0.0;
    for (int row = threadIdx.x; row < ncap; row += 256) s += cp[row] * cp[row];
    block_sum3(s, z0, z1, red);
    if (threadIdx.x == 0) nrm[blockIdx.x] = sqrt(s);
}
__global__ __launch_bounds__(256) void k_bigjac_sort(const double* __restrict__ W, const double* __restrict__ nrm, int ncap, double* __restrict__ A, double* __restrict__ D,
                                                     int32_t* info) {
    const int col = blockIdx.x;
    const double me = nrm[col];
    int below = 0;                                  // rank in ASCENDING order (ties by index)
    for (int j = threadIdx.x; j < ncap; j += 256) {
        const double o = nrm[j];
        below += (o < me || (o == me && j < col)) ? 1 : 0;
    }
    __shared__ int cnt[4];
    for (int off = 32; off > 0; off >>= 1) below += __shfl_down(below, off, 64);
    if ((threadIdx.x & 63) == 0) cnt[threadIdx.x >> 6] = below;
    __syncthreads();
    const int rk = cnt[0] + cnt[1] + cnt[2] + cnt[3];
    if (threadIdx.x == 0) D[rk] = me;
    const double inv = me > 0.0 ? 1.0 / me : 0.0;
    for (int row = threadIdx.x; row < ncap; row += 256) A[(size_t)rk * ncap + row] = W[(size_t)col * ncap + row] * inv;
    if (col == 0 && threadIdx.x == 0) info[3] = info[0] == 1 ? 0 : 1;      // not converged within BIGJ_MAX_SWEEPS: the caller's MPST_ERR_SVD
}

// eigenvalues D (ascending, ncap of them), eigenvectors in the columns of A (column-major): truncation rule and
// publication exactly as in k_eig_fin
__global__ __launch_bounds__(EIG_THREADS) void k_big_fin(View v, int lid, int going_left, const double* rawG, int rawn,
                                                         const double* __restrict__ A, const double* __restrict__ D,
                                                         const int32_t* info, int ncap, double* rawlam, double* rawE,
                                                         int32_t* rawinfo) {
    __shared__ double lam_s[CAP_LIMIT + 2];
    __shared__ double red[16];
    EigProblem pb = resolve(v, lid, going_left, rawG, rawn, 0);
    const bool raw = rawn > 0;
    const int tid = threadIdx.x;
    const int n = pb.n, nspec = pb.nspec;
    const int K0 = raw ? (n < CAP_LIMIT ? n : CAP_LIMIT) : (nspec < v.chi_max ? nspec : v.chi_max);
    double* Eout = raw ? rawE : v.E;
    const int ldE = raw ? n : v.cap;
    double part = 0.0;
    for (int i = tid; i < n; i += EIG_THREADS) part += pb.G[(size_t)i * n + i];
    double tr = wave_sum(part);
    if ((tid & 63) == 0) red[tid >> 6] = tr;
    __syncthreads();
    tr = 0.0;
    for (int i = 0; i < EIG_THREADS / 64; ++i) tr += red[i];
    if (tid < K0) lam_s[tid] = fmax(D[ncap - 1 - tid], 0.0);
    __syncthreads();
    if (pb.pair) tr *= 0.5;             // trace of the complex matrix
    const double inv = (!raw && v.rescale_after) ? 1.0 / sqrt(tr) : 1.0;
    const double cutoff = raw ? -1.0 : v.cutoff;
    const double inv2 = inv * inv;
    const int st = pb.pair ? 2 : 1;
    int nk = st * truncate_rule(lam_s, K0, nspec, tr, inv2, cutoff, pb.pair);      // real vectors to publish
    const int kout = raw ? K0 : nk;
    for (int i = tid; i < n * kout; i += EIG_THREADS) {
        const int k = i / n, c = i - k * n;
        Eout[(size_t)c * ldE + k] = A[(size_t)(ncap - 1 - k) * ncap + c];
    }
    const int failed = info[3];
    if (raw) {
        if (tid < K0) rawlam[tid] = lam_s[tid];
        if (tid == 0) *rawinfo = failed ? 1000 + failed : -2;
    } else {
        if (tid < K0 / st) v.lam[tid] = lam_s[st * tid];
        if (tid == 0) {
            bool bad = !(tr == tr) || tr > 1e300 || failed != 0;
            for (int i = 0; i < K0; ++i) {
                const double P = lam_s[i] * inv2;
                if (!(P == P) || P > 1e300) bad = true;
            }
            nk /= st;
            v.sc->n_keep = nk;
            v.sc->n_spec = K0 / st;
            v.sc->bt_norm2 = tr;
            v.sc->inv_norm = inv;
            v.sc->eig_sweeps = info[2];
            if (bad) v.sc->status = MPST_ERR_SVD;
            v.chi[lid + 1] = nk;
        }
    }
}

int big_eig_create(BigEig** out, int ncap, hipStream_t s, std::string* err) {
    BigEig* b = new BigEig();
    b->ncap = (ncap + 1) & ~1;          // the tournament pairs columns: an even number of them (one more zero column changes nothing)
    auto bail = [&](const char* what) {
        if (err) *err = what;
        big_eig_destroy(b);
        return MPST_ERR_DEVICE;
    };
    b->stream = s;
    const size_t nn = (size_t)b->ncap * b->ncap;
    if (hipMalloc((void**)&b->W, sizeof(double) * nn) != hipSuccess || hipMalloc((void**)&b->A, sizeof(double) * nn) != hipSuccess ||
        hipMalloc((void**)&b->D, sizeof(double) * b->ncap) != hipSuccess || hipMalloc((void**)&b->nrm, sizeof(double) * b->ncap) != hipSuccess ||
        hipMalloc((void**)&b->info, sizeof(int32_t) * 4) != hipSuccess)
        return bail("hipMalloc of the large-bond Jacobi buffers failed");
    *out = b;
    return 0;
}
void big_eig_destroy(BigEig* b) {
    if (!b) return;
    if (b->W) (void)hipFree(b->W);
    if (b->A) (void)hipFree(b->A);
    if (b->D) (void)hipFree(b->D);
    if (b->nrm) (void)hipFree(b->nrm);
    if (b->info) (void)hipFree(b->info);
    delete b;
}
// the iteration proper on b->W (filled by k_big_prep): leaves b->D, b->A, b->info[3]
static void enqueue_big_jacobi(BigEig* b, hipStream_t s) {
    const int np = b->ncap;
    hipLaunchKernelGGL(k_bigjac_norms, dim3(np), dim3(256), 0, s, b->W, np, b->nrm);
    hipLaunchKernelGGL(k_bigjac_scale, dim3(1), dim3(256), 0, s, b->nrm, np, b->D);        // (D[0] is scratch until the sort writes D)
    // sweeps in chunks of eight, the verdict read after each chunk: this is the cold path of a bond whose fast solver has failed its check
    // (the plain stream, never a captured graph), and a converged matrix should not cost the launches of the sweeps it no longer needs
    for (int sweep = 0; sweep < BIGJ_MAX_SWEEPS; ++sweep) {
        for (int r = 0; r < np - 1; ++r) hipLaunchKernelGGL(k_bigjac_round, dim3(np / 2), dim3(256), 0, s, b->W, np, r, b->info, b->D);
        hipLaunchKernelGGL(k_bigjac_check, dim3(1), dim3(64), 0, s, b->info);
        if ((sweep & 7) == 7) {
            int32_t done = 0;
            if (hipMemcpyAsync(&done, b->info, sizeof done, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) break;
            if (done) break;
        }
    }
    hipLaunchKernelGGL(k_bigjac_norms, dim3(np), dim3(256), 0, s, b->W, np, b->nrm);
    hipLaunchKernelGGL(k_bigjac_sort, dim3(np), dim3(256), 0, s, b->W, b->nrm, np, b->A, b->D, b->info);
}
int launch_eig_big(const View& v, int lid, int going_left, BigEig* b, hipStream_t s) {
    const int ncap = b->ncap;
    hipLaunchKernelGGL(k_big_prep, dim3(256), dim3(256), 0, s, v, lid, going_left, (const double*)nullptr, 0, b->W, ncap, b->info);
    enqueue_big_jacobi(b, s);
    hipLaunchKernelGGL(k_big_fin, dim3(1), dim3(EIG_THREADS), 0, s, v, lid, going_left, (const double*)nullptr, 0, b->A, b->D, b->info, ncap,
                       (double*)nullptr, (double*)nullptr, (int32_t*)nullptr);
    return hipGetLastError() == hipSuccess ? 0 : MPST_ERR_DEVICE;
}
int launch_eig_big_raw(const double* G, int n, double* lam, double* E, int32_t* info, BigEig* b, hipStream_t s) {
    View v{};
    hipLaunchKernelGGL(k_eig_clear, dim3(1), dim3(256), 0, s, lam, E, n);
    hipLaunchKernelGGL(k_big_prep, dim3(256), dim3(256), 0, s, v, 0, 0, G, n, b->W, b->ncap, b->info);
    enqueue_big_jacobi(b, s);
    hipLaunchKernelGGL(k_big_fin, dim3(1), dim3(EIG_THREADS), 0, s, v, 0, 0, G, n, b->A, b->D, b->info, b->ncap, lam, E, info);
    return hipGetLastError() == hipSuccess ? 0 : MPST_ERR_DEVICE;
}

}  // namespace mpst
