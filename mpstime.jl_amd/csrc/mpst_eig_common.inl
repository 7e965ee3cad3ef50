// Pieces of the n <= 128 eigensolver that more than one translation unit compiles: the layout of the global workspace the
// vector workgroups leave their results in, the problem a bond poses, and the verification + Loewdin polish of the candidate
// vectors (k_eig_fin of mpst_eig.hip; k_bond_tail of mpst_fused.hip runs the same code at the head of the bond's last launch).
#pragma once
#include "mpst_internal.h"

namespace mpst {

constexpr int EIG_THREADS = 512;   // 8 waves = 2 per SIMD: the phases are VALU-issue bound, fewer fatter waves win
constexpr int TRI_KMAX = 64;      // eigenpairs the tridiagonal path can deliver (n <= 128 and d >= 2 bound chi_max by 64)
constexpr int RAW_KMAX = 32;      // eigenpairs mpst_selftest_eig asks of it

// ---- global workspace shared by the three kernels (doubles) ----------------------------------
constexpr int WS_DE = 0;        // [128][2]  (d_j, e_{j-1}^2)
constexpr int WS_ES = 256;      // [128]
constexpr int WS_TAU = 384;     // [128]
constexpr int WS_MISC = 512;    // [0] lo  [1] hi  [2] ||T||  [3] 1.0 if the tridiagonal path is active
constexpr int WS_VS = 592;      // [8][128][16] reflectors, dense: block b, row c, reflector 16b+j (0 above its start)
constexpr int WS_Z = 16976;     // [128][KS] eigenvectors of G: component c of vector k at c*KS + k, KS = 32 or 64 (kstride)
constexpr int WS_LAM = 25168;   // [64]
constexpr int WS_RES = 25232;   // [64]  ||T z - lambda z||_inf
constexpr int WS_TOTAL = 25296;
// row stride of the eigenvector block: 32 columns while at most 32 eigenpairs are wanted (the headline shapes), else 64
__device__ __forceinline__ int kstride(int K0) { return K0 <= 32 ? 32 : 64; }

struct EigProblem {
    const double* G;
    int n, rows, nspec, K0;
    bool tri;       // tridiagonal path applicable
    bool pair;      // real embedding of a complex Hermitian matrix (View::zw): n, rows, nspec, K0 are the doubled counts
};
// eigenvectors the vector kernels compute: one per eigenvalue pair in pair mode
__device__ __forceinline__ int eig_nvec(const EigProblem& p) { return p.pair ? p.K0 >> 1 : p.K0; }
__device__ __forceinline__ EigProblem resolve(const View& v, int lid, int going_left, const double* rawG, int rawn,
                                              int rawalg) {
    EigProblem p;
    if (rawn > 0) {
        p.G = rawG;
        p.n = rawn;
        p.rows = rawn;
        p.nspec = rawn;
        const int kmax = (rawalg & 8) ? TRI_KMAX : RAW_KMAX;      // bit 3: all TRI_KMAX pairs (Rayleigh-Ritz of the subspace solver)
        p.K0 = rawn < kmax ? rawn : kmax;
        p.pair = (rawalg & 4) != 0 && (rawn & 1) == 0;            // test hook: rawG is an embedding, one vector per pair
        if (p.pair) p.K0 &= ~1;
        p.tri = (rawalg & 3) != MPST_SVD_JACOBI && rawn >= 2;
    } else {
        const int zw = view_zw(v);
        const int Dl = v.chi[lid], Dr = v.chi[lid + 2];
        const int X = Dl * v.d, Y = v.d * Dr;
        p.G = v.gram;
        p.n = zw * (going_left ? Y : X);
        p.rows = zw * v.C * (going_left ? X : Y);
        p.nspec = p.rows < p.n ? p.rows : p.n;                    // LAPACK's min(m, n)
        p.K0 = p.nspec < v.chi_max ? p.nspec : v.chi_max;         // eigenpairs that can survive maxdim (pair mode: v.chi_max is doubled too)
        p.pair = zw == 2;
        p.tri = v.svd_alg != MPST_SVD_JACOBI && p.K0 <= TRI_KMAX && p.n >= 2;
    }
    return p;
}

// LDS layout of k_eig_fin's tridiagonal branch: KS = 32 (at most 32 eigenpairs wanted) or 64 columns per row.
struct FinShared {
    double* Z;     // [128][KS] candidate eigenvectors of G
    double* D;     // [KS][KS]  Z^T Z - I
    double* Dh;    // KS = 32 only: [4][256] second-half partial tiles of the Gram matrix
    double* misc;  // [2] ||T||, [8..16) per-wave scratch, [32..32+KS) residuals
};
template <int KS>
__device__ __forceinline__ FinShared fin_carve(double* smem) {
    FinShared f;
    f.Z = smem;
    f.D = f.Z + 128 * KS;
    f.Dh = f.D + KS * KS;
    f.misc = f.Dh + (KS == 32 ? 1024 : 0);      // KS = 64: 8192 + 4096 + 128 doubles
    return f;
}

// Verifies the K vectors in t.Z ([c*KS + k]) against the residuals in t.misc[32..] and
// re-orthonormalises them; returns false if the Jacobi path has to take over.
template <int KS>
__device__ bool verify_and_polish(FinShared t, int n, int K) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // ---- verification + symmetric (Loewdin) re-orthonormalisation ---------------------------------
    // D = Z^T Z - I; Z <- Z (I - D/2) squares the deviation without leaving the subspace.  Up to two
    // rounds: close (but separated) eigenvalues leave |D| ~ 1e-6, genuine clusters leave |D| ~ 1 and
    // are handed to the Jacobi path, as is a residual ||T z - lambda z|| above 1e-8 ||T||.
    double* D = t.D;
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
    static_assert(EIG_THREADS == 512, "the MFMA tiling below is written for 8 waves");
    const int jl = lane & 15, q4 = lane >> 4;
    for (int round = 0; round < 2 && ok; ++round) {
        double err = 0.0;
        if constexpr (KS == 32) {
            // ---- D = Z^T Z - I on the fp64 MFMA: wave w owns the 16x16 tile (w&3) over rows [64(w>>2), +64) ----
            double* Dh = t.Dh;
            const int a0 = 16 * ((wave & 3) >> 1), b0 = 16 * (wave & 1), kb = 64 * (wave >> 2);
            d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
            for (int s4 = 0; s4 < 16; ++s4) {
                const double* zr = t.Z + (kb + 4 * s4 + q4) * 32;
                acc = mfma_f64(zr[a0 + jl], zr[b0 + jl], acc);
            }
            if (wave >= 4) {
#pragma unroll
                for (int r = 0; r < 4; ++r) Dh[(wave & 3) * 256 + (q4 + 4 * r) * 16 + jl] = acc[r];
            }
            __syncthreads();
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
        } else {
            // 4 x 4 tiles of 16 x 16, two per wave, each over all 128 rows
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int tile = wave + 8 * h;
                const int a0 = 16 * (tile >> 2), b0 = 16 * (tile & 3);
                d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
                for (int s4 = 0; s4 < 32; ++s4) {
                    const double* zr = t.Z + (4 * s4 + q4) * KS;
                    acc = mfma_f64(zr[a0 + jl], zr[b0 + jl], acc);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int aa = a0 + q4 + 4 * r, bb = b0 + jl;
                    double dv = 0.0;
                    if (aa < K && bb < K) {
                        dv = acc[r] - (aa == bb ? 1.0 : 0.0);
                        err = fmax(err, fabs(dv));
                    }
                    D[aa * KS + bb] = dv;
                }
            }
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
        // ---- Z <- Z - (Z D)/2: 8 row tiles x KS/16 column tiles, KS/16 per wave ----
        constexpr int CT = KS / 16, PER = CT;         // 8*CT tiles over 8 waves
        d4 upd[PER];
#pragma unroll
        for (int h = 0; h < PER; ++h) {
            const int tile = wave * PER + h;
            const int c0 = 16 * (tile / CT), a0 = 16 * (tile % CT);
            d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s4 = 0; s4 < KS / 4; ++s4) acc = mfma_f64(t.Z[(c0 + jl) * KS + 4 * s4 + q4], D[(4 * s4 + q4) * KS + a0 + jl], acc);
            upd[h] = acc;
        }
        __syncthreads();
#pragma unroll
        for (int h = 0; h < PER; ++h) {
            const int tile = wave * PER + h;
            const int c0 = 16 * (tile / CT), a0 = 16 * (tile % CT);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = c0 + q4 + 4 * r, aa = a0 + jl;
                if (c < n && aa < K) t.Z[c * KS + aa] -= 0.5 * upd[h][r];
            }
        }
        __syncthreads();
        if (emax < 1e-8) break;                       // one round suffices: residual |D|^2 < 1e-16
    }
    return ok;
}

}  // namespace mpst
