// Symmetric eigensolver for the Gram matrix of the bond tensor (n = d*chi <= 128), and the
// NDTensors truncation rule.  This is the device-side stand-in for ITensors.svd -> LAPACK gesdd
// + truncate! in decomposeBT (src/Training/RealRealHighDimension.jl:166-169,185-188).
//
// With A the (chi*C*d) x (d*chi) matrix the reference decomposes, G = A^T A = V S^2 V^T, so the
// right singular vectors are the eigenvectors of G, S = sqrt(lambda) and U*S = A V (k_split).
//
// Algorithm (MPST_SVD_JACOBI and, for now, the default): one-sided (Hestenes) Jacobi on the
// columns of G held in LDS (128 KB for n = 128): rotating column pairs until all columns are
// mutually orthogonal gives G J = V Lambda, i.e. column k converges to lambda_k v_k; the
// eigenvector is the normalised column and lambda_k its norm.  Orthogonality of the output is
// at the level of the (relative) rotation threshold, ~1e-15.  64 disjoint pairs are rotated
// concurrently (round-robin tournament ordering), 16 lanes per pair.
#include "mpst_internal.h"

namespace mpst {

constexpr int EIG_THREADS = 1024;
constexpr int EIG_MAX_SWEEPS = 40;

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
            if (k < npairs) {
                int p, q;
                pair_of(r, k, np, p, q);
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
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
                    app += __shfl_xor(app, o, 64);
                    aqq += __shfl_xor(aqq, o, 64);
                    apq += __shfl_xor(apq, o, 64);
                }
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
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o, 64);
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

// Engine kernel: eigen-decompose v.gram, apply the NDTensors truncation rule, publish
// n_keep / chi / inv_norm / spectrum and the kept eigenvectors E[dim][ldE].
__global__ __launch_bounds__(EIG_THREADS) void k_eig(View v, int lid, int going_left) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double red[16];
    const int tid = threadIdx.x;
    const int Dl = v.chi[lid], Dr = v.chi[lid + 2];
    const int X = Dl * v.d, Y = v.d * Dr;
    const int n = going_left ? Y : X;
    const int rows = v.C * (going_left ? X : Y);
    const int np = (n + 1) & ~1;
    EigShared sh = carve(smem, np);
    for (int i = tid; i < np * np; i += EIG_THREADS) {
        const int col = i / np, row = i - col * np;
        sh.Gs[i] = (row < n && col < n) ? v.gram[(size_t)row * n + col] : 0.0;
    }
    __syncthreads();
    // trace = ||bt_new||_F^2 (fixed order)
    double tr = 0.0;
    for (int i = tid; i < n; i += EIG_THREADS) tr += sh.Gs[(size_t)i * np + i];
    {
        // block_sum inline (deterministic)
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) tr += __shfl_xor(tr, o, 64);
        if ((tid & 63) == 0) red[tid >> 6] = tr;
        __syncthreads();
        tr = 0.0;
        for (int i = 0; i < EIG_THREADS / 64; ++i) tr += red[i];
        __syncthreads();
    }
    const int sweeps = jacobi_core(sh, np);

    const double inv = v.rescale_after ? 1.0 / sqrt(tr) : 1.0;
    const int ldE = v.cap;
    // eigenvalues in descending order
    __shared__ double lam_s[MAX_DIM + 2];
    if (tid < np) lam_s[sh.rank[tid]] = sh.nrm[tid];
    __syncthreads();
    if (tid < np) v.lam[tid] = lam_s[tid];
    // truncation (NDTensors truncate!, relative cutoff) by one thread
    if (tid == 0) {
        int nspec = rows < n ? rows : n;
        const double inv2 = inv * inv;
        int nk = nspec;
        double truncerr = 0.0, scale = 0.0;
        bool bad = false;
        for (int i = 0; i < nspec; ++i) {
            const double P = lam_s[i] * inv2;
            scale += P;
            if (!(P == P) || P > 1e300) bad = true;
        }
        if (scale == 0.0) scale = 1.0;
        if (nspec > 1) {
            while (nk > v.chi_max) {
                truncerr += lam_s[nk - 1] * inv2;
                --nk;
            }
            while (nk > 1 && truncerr + lam_s[nk - 1] * inv2 <= v.cutoff * scale) {
                truncerr += lam_s[nk - 1] * inv2;
                --nk;
            }
        }
        if (nk > v.chi_max) nk = v.chi_max;
        v.sc->n_keep = nk;
        v.sc->n_spec = nspec;
        v.sc->bt_norm2 = tr;
        v.sc->inv_norm = inv;
        v.sc->eig_sweeps = sweeps;
        v.sc->eig_sweeps_total += sweeps;
        if (bad || sweeps >= EIG_MAX_SWEEPS) v.sc->status = MPST_ERR_SVD;
        v.chi[lid + 1] = nk;
    }
    // kept eigenvectors: E[row][rank] = col/||col||
    for (int i = tid; i < np * np; i += EIG_THREADS) {
        const int col = i / np, row = i - col * np;
        const int rk = sh.rank[col];
        if (rk < ldE && row < n) {
            const double nr = sh.nrm[col];
            v.E[(size_t)row * ldE + rk] = nr > 0.0 ? sh.Gs[i] / nr : 0.0;
        }
    }
}

// Raw variant for tests: full spectrum + all eigenvectors E[i][k] (n x n row-major).
__global__ __launch_bounds__(EIG_THREADS) void k_eig_raw(const double* G, int n, double* lam, double* E,
                                                         int32_t* sweeps_out) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x;
    const int np = (n + 1) & ~1;
    EigShared sh = carve(smem, np);
    for (int i = tid; i < np * np; i += EIG_THREADS) {
        const int col = i / np, row = i - col * np;
        sh.Gs[i] = (row < n && col < n) ? G[(size_t)row * n + col] : 0.0;
    }
    __syncthreads();
    const int sweeps = jacobi_core(sh, np);
    if (tid < np) {
        const int rk = sh.rank[tid];
        if (rk < n) lam[rk] = sh.nrm[tid];
    }
    for (int i = tid; i < np * np; i += EIG_THREADS) {
        const int col = i / np, row = i - col * np;
        const int rk = sh.rank[col];
        if (rk < n && row < n) {
            const double nr = sh.nrm[col];
            E[(size_t)row * n + rk] = nr > 0.0 ? sh.Gs[i] / nr : 0.0;
        }
    }
    if (tid == 0) *sweeps_out = sweeps;
}

static size_t eig_lds_bytes(int np) { return ((size_t)np * np + np) * sizeof(double) + (np + 4) * sizeof(int); }

static bool g_attr_set = false;
static void ensure_attrs() {
    if (g_attr_set) return;
    const int maxb = (int)eig_lds_bytes(MAX_DIM);
    (void)hipFuncSetAttribute((const void*)k_eig, hipFuncAttributeMaxDynamicSharedMemorySize, maxb);
    (void)hipFuncSetAttribute((const void*)k_eig_raw, hipFuncAttributeMaxDynamicSharedMemorySize, maxb);
    g_attr_set = true;
}

void launch_eig(const View& v, int lid, int going_left, hipStream_t s) {
    ensure_attrs();
    int dm = v.d * v.cap;
    if (dm > MAX_DIM) dm = MAX_DIM;
    const int np = (dm + 1) & ~1;
    hipLaunchKernelGGL(k_eig, dim3(1), dim3(EIG_THREADS), eig_lds_bytes(np), s, v, lid, going_left);
}

void launch_eig_raw(const double* G, int n, int alg, double* lam, double* E, int32_t* sweeps, hipStream_t s) {
    (void)alg;
    ensure_attrs();
    const int np = (n + 1) & ~1;
    hipLaunchKernelGGL(k_eig_raw, dim3(1), dim3(EIG_THREADS), eig_lds_bytes(np), s, G, n, lam, E, sweeps);
}

}  // namespace mpst
