// Symmetric eigensolver for the Gram matrix of the bond tensor (n = d*chi <= 128), and the
// NDTensors truncation rule.  Device-side stand-in for ITensors.svd -> LAPACK gesdd + truncate!
// in decomposeBT (src/Training/RealRealHighDimension.jl:166-169,185-188).
//
// With A the (chi*C*d) x (d*chi) matrix the reference decomposes, G = A^T A = V S^2 V^T: the
// right singular vectors are the eigenvectors of G, S = sqrt(lambda), and U*S = A V (k_split).
// Only the chi_max largest eigenpairs and the trace are needed: the truncation rule discards the
// rest, and the discarded weight is trace - sum(kept).
//
// One workgroup (the problem is a latency chain, not a throughput problem), two algorithms:
//
//  * default: Householder tridiagonalisation with the matrix held in registers (16 doubles per
//    thread, thread (r, q) owns G[r][q + 8k]), 2 barriers per reflector; multisection Sturm
//    bisection for the K largest eigenvalues (16 lanes per eigenvalue, 17-fold interval shrink
//    per step, division-free recurrence); eigenvectors of the tridiagonal by twisted factorisation
//    (one lane per eigenvalue); back-transformation through the stored reflectors (16 lanes per
//    vector).  The result is verified on the device (residual in T, orthonormality of the output)
//    and, if the check fails (clustered eigenvalues), the kernel falls through to
//  * MPST_SVD_JACOBI: one-sided (Hestenes) Jacobi on the columns of G in LDS - slow (ms) but
//    unconditionally robust; column k converges to lambda_k v_k.
#include "mpst_internal.h"

namespace mpst {

constexpr int EIG_THREADS = 512;   // 8 waves = 2 per SIMD: the phases are VALU-issue bound, fewer fatter waves win
constexpr int EIG_MAX_SWEEPS = 40;
constexpr int TRI_KMAX = 32;      // eigenpairs the tridiagonal path can deliver
constexpr int TRI_NSTEP = 14;     // 17^-14 = 6e-18 of the Gershgorin interval
constexpr int EIG_LDS_DOUBLES = 17664;  // 138 KB: max over both algorithms

__device__ __forceinline__ int hi32(double x) { return __double2hiint(x); }

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
    double* Z;     // [128][32]  eigenvectors of T, later of G
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
    t.Ub = t.Z + 128 * TRI_KMAX;
    t.misc = t.Ub + 128 * TRI_KMAX;   // total 17440 + 192 <= EIG_LDS_DOUBLES
    return t;
}
__device__ __forceinline__ int voff(int i, int n) { return i * (n - 1) - (i * (i - 1)) / 2; }

// reciprocal to full double precision from the hardware seed (2 Newton steps)
__device__ __forceinline__ double frcp(double b) {
    double r = __builtin_amdgcn_rcp(b);
    r = fma(fma(-b, r, 1.0), r, r);
    r = fma(fma(-b, r, 1.0), r, r);
    return r;
}

// # eigenvalues of T smaller than x: sign changes of the Sturm sequence, division-free with
// power-of-two rescaling every 8 steps.
__device__ __forceinline__ int sturm_count(const double* __restrict__ de, int n, double x) {
    double pp = 1.0, p = de[0] - x;
    int cnt = ((unsigned)hi32(p)) >> 31;
    for (int j = 1; j < n; ++j) {
        const double t = de[2 * j] - x;
        const double pn = fma(t, p, -de[2 * j + 1] * pp);
        cnt += ((unsigned)(hi32(pn) ^ hi32(p))) >> 31;
        pp = p;
        p = pn;
        if ((j & 7) == 0) {
            int e = (hi32(p) >> 20) & 0x7ff;
            if (e == 0) e = (hi32(pp) >> 20) & 0x7ff;
            e = e < 2 ? 2 : (e > 2044 ? 2044 : e);
            const double sc = __hiloint2double((2046 - e) << 20, 0);
            p *= sc;
            pp *= sc;
        }
    }
    return cnt;
}

// Top-K eigenpairs of the symmetric G (global, n x n, 2 <= n <= 128, K <= 32).
// Outputs: t.lam[0..K) descending (LDS); eigenvectors in t.Z[c*32 + k].  Returns true if the
// on-device verification passed.
// `select(lam, K)` is called (by every thread, after a barrier) once the K largest eigenvalues are in
// t.lam and returns how many eigenvectors are wanted (<= K): eigenvectors of values the truncation
// rule is going to discard are never formed - those are the clustered, noise-level ones that make
// the twisted factorisation lose orthogonality.
template <typename Select>
__device__ bool tri_solve(const double* __restrict__ G, int n, int K, TriShared t, unsigned long long* stamps,
                          Select select) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#define TRI_STAMP(i) do { if (stamps && tid == 0) stamps[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
    TRI_STAMP(0);
    if (stamps && tid == 0) stamps[6] = __builtin_readcyclecounter();
    const int r = tid >> 2, q = tid & 3;
    // ---- load: thread (r, q) owns the column pairs c = 2q + 8k + {0,1}, k = 0..15 of row r ----
    // (A[2k+h], 32 doubles); its LDS operands are 16-byte reads, broadcast across the 16 rows of a wave.
    double A[32];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = 2 * q + 8 * k + h;
            A[2 * k + h] = (r < n && c < n) ? G[(size_t)r * n + c] : 0.0;
        }
    }
    if (tid < 256) {
        t.xs[tid] = 0.0;
        t.ps[tid] = 0.0;
    }
    __syncthreads();
    // ---- Householder tridiagonalisation (dsytd2, full storage) -----------------------------
    // Step i: (a) every live wave forms its rows of p = tau*A*v from the published reflector v_i
    // (barrier), (b) the scalar v^T p and the rank-2 update A -= v w^T + w v^T on its register
    // block, with w = p + a2 v folded into two FMAs per element:
    //     A[r][c] -= v_r p_c + (a2 v_r + w_r) v_c.
    // (c) Look-ahead: the wave that owns row i+1 runs at raised priority, so it finishes its part of
    // the update first and builds reflector i+1 (row -> LDS, norm, beta, tau, v) while the other
    // waves are still updating.  2 barriers per step.
    auto build_reflector = [&](int i) {
        // executed by all 64 lanes of the wave that owns row i
        double* vbn = t.xs + (i & 1) * 128;
        double* xr = t.misc + 64;                  // [128] raw row i (only this wave touches it)
        if (r == i) {
#pragma unroll
            for (int k = 0; k < 16; ++k) *(double2*)&xr[2 * q + 8 * k] = make_double2(A[2 * k], A[2 * k + 1]);
        }
        const int c0 = lane, c1 = lane + 64;
        const double x0 = xr[c0], x1 = xr[c1];     // same wave: LDS ops complete in order
        const double s = wave_sum((c0 >= i + 2 ? x0 * x0 : 0.0) + (c1 >= i + 2 ? x1 * x1 : 0.0));
        const double al = xr[i + 1], di = xr[i];
        double beta = al, tau = 0.0, scale = 0.0;
        if (s != 0.0) {
            beta = -copysign(sqrt(al * al + s), al);
            tau = (beta - al) * frcp(beta);
            scale = frcp(al - beta);
        }
        const double v0 = (c0 == i + 1) ? 1.0 : (c0 > i + 1 ? x0 * scale : 0.0);
        const double v1 = (c1 == i + 1) ? 1.0 : (c1 > i + 1 ? x1 * scale : 0.0);
        vbn[c0] = v0;
        vbn[c1] = v1;
        const int off = voff(i, n) - i - 1;
        if (c0 > i && c0 < n) t.Vs[off + c0] = v0;
        if (c1 > i && c1 < n) t.Vs[off + c1] = v1;
        if (lane == 0) {
            t.de[2 * i] = di;
            t.es[i] = beta;
            t.taus[i] = tau;
        }
    };
    if (wave == 0) build_reflector(0);
    __syncthreads();
    for (int i = 0; i < n - 1; ++i) {
        const double* vb = t.xs + (i & 1) * 128;
        double* p = t.ps + (i & 1) * 128;
        const bool live = (wave * 16 + 15) > i;    // this wave still owns trailing rows
        const bool next_owner = (i + 1 < n - 1) && wave == ((i + 1) >> 4);
        const double tau = t.taus[i];
        double2 vv[16];
        if (live) {
            if (next_owner) __builtin_amdgcn_s_setprio(3);
#pragma unroll
            for (int k = 0; k < 16; ++k) vv[k] = *(const double2*)&vb[2 * q + 8 * k];
            double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
#pragma unroll
            for (int k = 0; k < 16; k += 2) {
                acc0 = fma(A[2 * k], vv[k].x, acc0);
                acc1 = fma(A[2 * k + 1], vv[k].y, acc1);
                acc2 = fma(A[2 * k + 2], vv[k + 1].x, acc2);
                acc3 = fma(A[2 * k + 3], vv[k + 1].y, acc3);
            }
            const double acc = sum4((acc0 + acc1) + (acc2 + acc3));
            if (q == 0) p[r] = (r > i) ? tau * acc : 0.0;
        } else if (wave * 16 + 15 == i) {
            // this wave's rows have just retired: clear their p entries in both buffers for good
            if (q == 0) {
                t.ps[r] = 0.0;
                t.ps[128 + r] = 0.0;
            }
        }
        __syncthreads();
        if (live) {
            const int c0 = lane, c1 = lane + 64;
            const double pr = p[r], vr = vb[r];
            const double dot = wave_sum(p[c0] * vb[c0] + p[c1] * vb[c1]);
            const double a2 = -0.5 * tau * dot;
            const double wr = pr + a2 * vr;
            const double g = a2 * vr + wr;
#pragma unroll
            for (int kb = 0; kb < 16; kb += 8) {
                double2 pv[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) pv[k] = *(const double2*)&p[2 * q + 8 * (kb + k)];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int kk = kb + k;
                    A[2 * kk] = fma(-vr, pv[k].x, A[2 * kk]);
                    A[2 * kk + 1] = fma(-vr, pv[k].y, A[2 * kk + 1]);
                    A[2 * kk] = fma(-g, vv[kk].x, A[2 * kk]);
                    A[2 * kk + 1] = fma(-g, vv[kk].y, A[2 * kk + 1]);
                }
            }
            if (next_owner) {
                build_reflector(i + 1);
                __builtin_amdgcn_s_setprio(0);
            }
        }
        __syncthreads();
    }
    {   // last diagonal element
        double* x = t.xs + ((n - 1) & 1) * 128;
        __syncthreads();
        if (r == n - 1) {
#pragma unroll
            for (int k = 0; k < 16; ++k) *(double2*)&x[2 * q + 8 * k] = make_double2(A[2 * k], A[2 * k + 1]);
        }
        __syncthreads();
        if (tid == 0) {
            t.de[2 * (n - 1)] = x[n - 1];
            t.es[n - 1] = 0.0;
        }
        __syncthreads();
    }
    TRI_STAMP(1);
    if (stamps && tid == 0) stamps[7] = __builtin_readcyclecounter();
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
    // ---- multisection bisection for the K largest eigenvalues --------------------------------
    {
        const int grp = tid >> 4, sec = tid & 15;
        if (grp < K) {
            double lo = t.misc[0], hi = t.misc[1];
            const int target = n - 1 - grp;   // ascending index of the grp-th largest
            for (int it = 0; it < TRI_NSTEP; ++it) {
                const double h = (hi - lo) * (1.0 / 17.0);
                const double xq = lo + h * (sec + 1);
                const int cnt = sturm_count(t.de, n, xq);
                const unsigned long long b = __ballot(cnt <= target);
                const int jj = __popc((unsigned)((b >> (lane & 48)) & 0xFFFFull));
                const double nlo = jj > 0 ? lo + h * jj : lo;
                const double nhi = jj < 16 ? lo + h * (jj + 1) : hi;
                lo = nlo;
                hi = nhi;
            }
            if (sec == 0) t.lam[grp] = 0.5 * (lo + hi);
        }
    }
    __syncthreads();
    K = select(t.lam, K);
    TRI_STAMP(2);
    // ---- eigenvectors of T: twisted factorisation, one lane per eigenvalue ------------------------
    if (tid < K) {
        const int k = tid;
        const double lamk = t.lam[k];
        const double pivmin = 1e-290 + 1e-30 * t.misc[2];
        double* z = t.Z + k;     // stride 32
        double* ub = t.Ub + k;
        double dm = t.de[2 * (n - 1)] - lamk;
        if (fabs(dm) < pivmin) dm = -pivmin;
        z[(n - 1) * 32] = dm;                              // D^-_{n-1}
        for (int i = n - 2; i >= 0; --i) {
            const double e = t.es[i];
            const double u = e * frcp(dm);                 // U_i = e_i / D^-_{i+1}
            ub[i * 32] = u;
            dm = (t.de[2 * i] - lamk) - u * e;
            if (fabs(dm) < pivmin) dm = -pivmin;
            z[i * 32] = dm;
        }
        double dp = t.de[0] - lamk;
        if (fabs(dp) < pivmin) dp = -pivmin;
        double gmin = fabs(z[0]);                          // gamma_0 = D^-_0
        int rb = 0;
        for (int i = 0; i < n - 1; ++i) {
            const double e = t.es[i];
            const double l = e * frcp(dp);                 // L_i = e_i / D_i
            const double sh = t.de[2 * (i + 1)] - lamk;
            double dn = sh - l * e;
            if (fabs(dn) < pivmin) dn = -pivmin;
            const double gam = dn + z[(i + 1) * 32] - sh;  // gamma_{i+1}
            z[i * 32] = l;
            if (fabs(gam) < gmin) {
                gmin = fabs(gam);
                rb = i + 1;
            }
            dp = dn;
        }
        z[rb * 32] = 1.0;
        double nrm = 1.0;
        for (int i = rb - 1; i >= 0; --i) {
            const double zi = -z[i * 32] * z[(i + 1) * 32];
            z[i * 32] = zi;
            nrm += zi * zi;
        }
        for (int i = rb; i < n - 1; ++i) {
            const double zi = -ub[i * 32] * z[i * 32];
            z[(i + 1) * 32] = zi;
            nrm += zi * zi;
        }
        const double sc = 1.0 / sqrt(nrm);
        // normalise + residual ||T z - lam z||_inf (verification)
        double res = 0.0, zprev = 0.0, zc = z[0] * sc;
        for (int i = 0; i < n; ++i) {
            const double zn = (i < n - 1) ? z[(i + 1) * 32] * sc : 0.0;
            const double ri = (t.de[2 * i] - lamk) * zc + (i > 0 ? t.es[i - 1] * zprev : 0.0) +
                              (i < n - 1 ? t.es[i] * zn : 0.0);
            res = fmax(res, fabs(ri));
            z[i * 32] = zc;
            zprev = zc;
            zc = zn;
        }
        t.misc[32 + k] = res;
    }
    __syncthreads();
    TRI_STAMP(3);
    // ---- back-transformation z <- H(0) H(1) ... H(n-2) z, 16 lanes per vector --------------------
    {
        const int k = tid >> 4, j = tid & 15;
        if (k < K) {
            double zz[8];
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int c = j + 16 * m;
                zz[m] = c < n ? t.Z[c * 32 + k] : 0.0;
            }
            for (int i = n - 2; i >= 0; --i) {
                const double tau = t.taus[i];
                const double* vi = t.Vs + voff(i, n) - i - 1;    // vi[c] valid for i < c < n
                double vv[8];
                double dot = 0.0;
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const int c = j + 16 * m;
                    vv[m] = (c > i && c < n) ? vi[c] : 0.0;
                    dot += vv[m] * zz[m];
                }
                dot = sum16(dot);
                const double f = tau * dot;
#pragma unroll
                for (int m = 0; m < 8; ++m) zz[m] -= f * vv[m];
            }
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int c = j + 16 * m;
                if (c < n) t.Z[c * 32 + k] = zz[m];
            }
        }
    }
    __syncthreads();
    TRI_STAMP(4);
    // ---- verification + symmetric (Loewdin) re-orthonormalisation ---------------------------------
    // D = Z^T Z - I; Z <- Z (I - D/2) squares the deviation without leaving the subspace.  Up to two
    // rounds: close (but separated) eigenvalues leave |D| ~ 1e-6, genuine clusters leave |D| ~ 1 and
    // are handed to the Jacobi path, as is a residual ||T z - lambda z|| above 1e-8 ||T||.
    double* D = t.Ub;      // [32][32], Ub is free now
    bool ok = true;
    {
        double rres = 0.0;
        if (tid < K) rres = t.misc[32 + tid] / (t.misc[2] > 0.0 ? t.misc[2] : 1.0);
        rres = wave_max(rres);
        __syncthreads();
        if (lane == 0) t.misc[8 + wave] = rres;
        __syncthreads();
        double rmax = 0.0;
        for (int w = 0; w < EIG_THREADS / 64; ++w) rmax = fmax(rmax, t.misc[8 + w]);
        ok = rmax < 1e-8 && rmax == rmax;
    }
    for (int round = 0; round < 2 && ok; ++round) {
        double err = 0.0;
        const int b = tid & 31;
        for (int a = tid >> 5; a < 32; a += EIG_THREADS >> 5) {
            double dv = 0.0;
            if (a < K && b < K) {
                double dot = 0.0;
                for (int c = 0; c < n; ++c) dot += t.Z[c * 32 + a] * t.Z[c * 32 + b];
                dv = dot - (a == b ? 1.0 : 0.0);
                err = fmax(err, fabs(dv));
            }
            D[a * 32 + b] = dv;
        }
        err = wave_max(err);
        __syncthreads();
        if (lane == 0) t.misc[8 + wave] = err;
        __syncthreads();
        double emax = 0.0;
        for (int w = 0; w < EIG_THREADS / 64; ++w) emax = fmax(emax, t.misc[8 + w]);
        if (!(emax < 1e-4)) {
            ok = false;
            break;
        }
        if (round == 1 && emax < 1e-13) break;       // already orthonormal to rounding
        constexpr int NZ = 4096 / EIG_THREADS;
        double zn[NZ];
#pragma unroll
        for (int m = 0; m < NZ; ++m) {
            const int idx = tid + m * EIG_THREADS;        // over [128][32]
            const int c = idx >> 5, a = idx & 31;
            double acc = 0.0;
            if (c < n && a < K) {
                for (int bb = 0; bb < K; ++bb) acc += t.Z[c * 32 + bb] * D[bb * 32 + a];
                acc = t.Z[c * 32 + a] - 0.5 * acc;
            }
            zn[m] = acc;
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < NZ; ++m) {
            const int idx = tid + m * EIG_THREADS;
            const int c = idx >> 5, a = idx & 31;
            if (c < n && a < K) t.Z[idx] = zn[m];
        }
        __syncthreads();
        if (emax < 1e-8) break;                       // one round suffices: residual |D|^2 < 1e-16
    }
    TRI_STAMP(5);
#undef TRI_STAMP
    return ok;
}

// =====================================================================================
// Engine kernel: eigen-decompose v.gram, apply the NDTensors truncation rule, publish
// n_keep / chi / inv_norm / spectrum and the kept eigenvectors E[dim][ldE].
// =====================================================================================
__global__ __launch_bounds__(EIG_THREADS) void k_eig(View v, int lid, int going_left) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double lam_s[MAX_DIM + 2];
    __shared__ double red[16];
    const int tid = threadIdx.x;
    const int Dl = v.chi[lid], Dr = v.chi[lid + 2];
    const int X = Dl * v.d, Y = v.d * Dr;
    const int n = going_left ? Y : X;
    const int rows = v.C * (going_left ? X : Y);
    const int nspec = rows < n ? rows : n;                    // LAPACK's min(m, n)
    const int K0 = nspec < v.chi_max ? nspec : v.chi_max;     // eigenpairs that can survive maxdim
    const int ldE = v.cap;
    // trace = ||bt_new||_F^2 (fixed order)
    double tr = 0.0;
    for (int i = tid; i < n; i += EIG_THREADS) tr += v.gram[(size_t)i * n + i];
    tr = wave_sum(tr);
    if ((tid & 63) == 0) red[tid >> 6] = tr;
    __syncthreads();
    tr = 0.0;
    for (int i = 0; i < EIG_THREADS / 64; ++i) tr += red[i];
    __syncthreads();

    const double inv = v.rescale_after ? 1.0 / sqrt(tr) : 1.0;
    // NDTensors truncate! (relative cutoff) on P = lambda*inv^2.  The weight beyond the
    // first K0 values is trace - sum(first K0).  Evaluated redundantly by every thread.
    auto truncate = [&](const double* lam, int K) -> int {
        const double inv2 = inv * inv;
        const double scale0 = tr * inv2;
        const double scale = scale0 == 0.0 ? 1.0 : scale0;
        double kept = 0.0;
        for (int i = 0; i < K; ++i) kept += lam[i] * inv2;
        int nk = K;
        double truncerr = scale0 - kept;
        if (truncerr < 0.0 || nspec <= K) truncerr = 0.0;
        if (nspec > 1) {
            while (nk > 1 && truncerr + lam[nk - 1] * inv2 <= v.cutoff * scale) {
                truncerr += lam[nk - 1] * inv2;
                --nk;
            }
        }
        return nk;
    };
    int sweeps = 0;
    bool done = false;
    int nk = K0;
    if (v.svd_alg != MPST_SVD_JACOBI && K0 <= TRI_KMAX && n >= 2) {
        TriShared t = tri_carve(smem);
        done = tri_solve(v.gram, n, K0, t, v.sc->eig_stamps, [&](const double* lam, int K) {
            nk = truncate(lam, K);
            return nk;
        });
        if (done) {
            if (tid < K0) lam_s[tid] = t.lam[tid];
            for (int i = tid; i < n * nk; i += EIG_THREADS) {
                const int c = i / nk, k = i - c * nk;
                v.E[(size_t)c * ldE + k] = t.Z[c * 32 + k];
            }
        }
        __syncthreads();
    }
    const bool fell_back = !done && v.svd_alg != MPST_SVD_JACOBI && K0 <= TRI_KMAX && n >= 2;
    if (!done) {
        sweeps = jacobi_solve(v.gram, n, smem, lam_s, v.E, ldE, K0);
        if (sweeps == 0) sweeps = 1;
        __syncthreads();
        nk = truncate(lam_s, K0);
    }
    __syncthreads();
    if (tid < K0) v.lam[tid] = lam_s[tid];
    if (tid == 0) {
        bool bad = !(tr == tr) || tr > 1e300;
        for (int i = 0; i < K0; ++i) {
            const double P = lam_s[i] * inv * inv;
            if (!(P == P) || P > 1e300) bad = true;
        }
        v.sc->n_keep = nk;
        v.sc->n_spec = K0;
        v.sc->bt_norm2 = tr;
        v.sc->inv_norm = inv;
        v.sc->eig_sweeps = sweeps;
        v.sc->eig_sweeps_total += sweeps;
        if (fell_back) v.sc->eig_fallbacks += 1;
        if (bad || sweeps >= EIG_MAX_SWEEPS) v.sc->status = MPST_ERR_SVD;
        v.chi[lid + 1] = nk;
    }
}

// Raw variant for tests.  alg 1: Jacobi, full spectrum, *info = sweeps.  alg 0/2: tridiagonal
// path, top K = min(n, 32) eigenpairs (the remaining outputs are zero), *info = -1; if the
// device-side verification failed Jacobi is used instead and *info = 0.
__global__ __launch_bounds__(EIG_THREADS) void k_eig_raw(const double* G, int n, int alg, double* lam, double* E,
                                                         int32_t* info) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double lam_s[MAX_DIM + 2];
    const int tid = threadIdx.x;
    for (int i = tid; i < n * n; i += EIG_THREADS) E[i] = 0.0;
    if (tid < n) lam[tid] = 0.0;
    __syncthreads();
    if (alg != MPST_SVD_JACOBI && n >= 2) {
        const int K = n < TRI_KMAX ? n : TRI_KMAX;
        TriShared t = tri_carve(smem);
        const bool ok = tri_solve(G, n, K, t, nullptr, [](const double*, int K_) { return K_; });
        if (ok) {
            if (tid < K) lam[tid] = t.lam[tid];
            for (int i = tid; i < n * K; i += EIG_THREADS) {
                const int c = i / K, k = i - c * K;
                E[(size_t)c * n + k] = t.Z[c * 32 + k];
            }
            if (tid == 0) *info = -1;
            return;
        }
        __syncthreads();
    }
    const int sweeps = jacobi_solve(G, n, smem, lam_s, E, n, n);
    if (tid < n) lam[tid] = lam_s[tid];
    if (tid == 0) *info = (alg != MPST_SVD_JACOBI && n >= 2) ? 0 : sweeps;
}

static size_t eig_lds_bytes() { return (size_t)EIG_LDS_DOUBLES * sizeof(double); }

static bool g_attr_set = false;
static void ensure_attrs() {
    if (g_attr_set) return;
    (void)hipFuncSetAttribute((const void*)k_eig, hipFuncAttributeMaxDynamicSharedMemorySize, (int)eig_lds_bytes());
    (void)hipFuncSetAttribute((const void*)k_eig_raw, hipFuncAttributeMaxDynamicSharedMemorySize, (int)eig_lds_bytes());
    g_attr_set = true;
}

void launch_eig(const View& v, int lid, int going_left, hipStream_t s) {
    ensure_attrs();
    hipLaunchKernelGGL(k_eig, dim3(1), dim3(EIG_THREADS), eig_lds_bytes(), s, v, lid, going_left);
}

void launch_eig_raw(const double* G, int n, int alg, double* lam, double* E, int32_t* sweeps, hipStream_t s) {
    ensure_attrs();
    hipLaunchKernelGGL(k_eig_raw, dim3(1), dim3(EIG_THREADS), eig_lds_bytes(), s, G, n, alg, lam, E, sweeps);
}

}  // namespace mpst
