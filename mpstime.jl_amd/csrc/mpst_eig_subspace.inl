// Top-chi eigenpairs of the Gram matrix of a LARGE bond tensor (n = d*chi > 128) without tridiagonalising it - part of the
// translation unit of mpst_eig_blocked.hip (BtBufs, BtProblem, the verification / publication kernels are shared).
//
// What decomposeBT needs (RealRealHighDimension.jl:146-203) is the chi_max largest singular triplets of the m x n matrix M it
// hands to svd, and the trace.  The exact route squares M and reduces the n x n Gram matrix with n - 2 DEPENDENT Householder
// steps (2.6-3.4 us each across the chip: 1.45 ms at n = 512, 60-80 % of a bond).  Measured on the bond matrices of real sweeps
// (profiles/r05_subspace_feasibility.json, lab/subspace/): behind the kept values the spectrum of M falls off fast - the
// update M = M0 - eta g adds a quickly decaying tail to a matrix of rank <= C chi - so a randomised subspace iteration with
// p = chi_max + 32 columns captures the kept triplets to 1e-10 of sigma_1 after FIVE applications of M / M^H, from a random start,
// on every one of the dumped bonds.  Every step is a GEMM on the fp64 MFMA:
//     X = cholqr(M^H Omega);   Q = cholqr(M X);   X = cholqr(M^H Q);   Q = cholqr(M X);   Z = M^H Q;
//     Z^H Z = W Theta W^H (p x p, the LDS-resident solver of mpst_eig.hip);   V = Z W Theta^-1/2,  sigma^2 = Theta.
// (fp32 bond tensors, whose entries carry 6e-8 of rounding noise and whose bar is 3e-8: FOUR applications, X = Omega directly.)
// The half-steps go through M, not G = M^H M: a Cholesky-QR of G X sees the SQUARED spectrum (condition 1e10 and more).
// cholqr = Gram matrix (partials over row slices, summed in a fixed order), the factor L^-H of it, and the product with it.  The
// factorisation is one workgroup: real types blocked (k_ss_cholb: 16 x 16 diagonal blocks factored and inverted by one wave in
// registers, everything else on the MFMA from LDS; 45 us at p = 96), complex types pivot by pivot with the matrix in registers
// (k_ss_chol_c; 94 us; the real pivot-wise kernel k_ss_chol stays selectable, MPST_SS_PIVOT=1).  Pivots below 2e-15 of the column's own
// squared norm drop the column (rank-deficient blocks of the growth phase).
// The result is CERTIFIED on the device before it is used: residuals r_k = G v_k - theta_k v_k of the kept pairs against G itself,
//     sqrt(sum_k |r_k|^2 / theta_k) <= 1e-9 ||M||_F      (the first-order bound of ||M (P~ - P)||_F: a residual lies outside the
//     iterated block, where the spectrum is far below theta_k; checked against the true error on every dumped bond),
// orthonormality |V^H V - I| (k_bt_gram / k_bt_decide, then the Loewdin rounds of k_bt_polish as for the exact solver) and the
// Frobenius certificate ||G||_F^2 - ||H||_F^2 >= (what the block missed)^2.  RESOLUTION OF THAT CERTIFICATE: the subtraction resolves
// 64 eps ||G||_F^2, i.e. a missed eigenvalue above ~1.2e-7 lambda_1; the fp64 cutoff keeps values down to 1e-10 of the trace.  For kept
// values in that window the certificate proves nothing and nothing is claimed: there the residual test says the kept pairs ARE
// eigenpairs, and that they are the LARGEST ones rests on the iteration itself (a direction of M with a larger singular value than a
// kept one is amplified MORE than the kept one by every application; to be absent after five it must be orthogonal to the start block
// to ~1e-16) - an argument, not a check.  The linear deficit tr G - tr H (resolution eps tr) was considered and not used: it sums
// lambda_j (1 - |Q^H u_j|^2) over ALL directions, so the unconverged oversampling columns and the discarded tail, which are legitimate,
// exceed a kept value of 1e-9 tr routinely and the bond would be sent to the exact path for no reason.
// A bond that fails any of them is solved by the exact path, whose launches follow in the same stream and leave at once when
// the word st[0] says the result stands.  Deterministic: Omega is a hash of (row, column), all sums have a fixed order.
// Real and complex element types (the complex kernels are at the end of the file); not attempted: d chi <= 128 (measured: no gain,
// profiles/r05_subspace_feasibility.json), C chi_max > 128, complex blocks wider than 96, fp32 tensors whose bond is still growing.

constexpr int SS_EXTRA = 32;      // oversampling: columns beyond chi_max
constexpr int SS_PMAX = 128;      // block width limit = order limit of the LDS-resident Rayleigh-Ritz solver
constexpr int SS_KS = 4;          // row slices of the Gram partials

struct SsBufs {
    double* Mw;          // [mcap][ncap] fp64 copy of the decomposed matrix, row stride = live n
    double* Lb[2];       // left blocks  [mcap][pc]  (Omega, M X, Q)
    double* Rb[2];       // right blocks [ncap][pc]  (M^H Q, X, Z)
    double* Sp;          // [SS_KS][pc*pc] Gram partials
    double* Tm;          // [pc][pc] L^-H (upper triangular, dead columns zero)
    double* H;           // [pc][pc] Z^H Z
    double* lamH;        // [pc]
    double* WH;          // [pc][pc] eigenvectors of H (column k)
    int32_t* infoH;
    double* wsH;         // workspace of the LDS-resident solver
    double* part;        // [ncap/16][CAP_LIMIT] squared residual pieces, then [ncap/16] pieces of ||G||_F^2
    int32_t* st;         // [0] this bond's subspace result stands (the exact path's kernels leave), [1] attempted on this bond,
                         // [2] bonds accepted, [3] bonds attempted (since creation)
    int pc;              // block width at capacity: multiple of 16, <= SS_PMAX
    int mcap, ncap;
    int capped;          // 1: C chi_max + 32 exceeds the block capacity: no oversampling at full bond dimension, one more round of half-steps
    int cx;              // 1: complex element type - Mw, the blocks, Sp, Tm are double2 arrays of the same element counts, H is the real
                         //    embedding [[Hr, -Hi], [Hi, Hr]] (2 pc x 2 pc) the pair-mode Hermitian solver reads
    const int32_t* rrflag;   // complex: verdict word of the Rayleigh-Ritz solve (0 = delivered)
    int dbg;             // bring-up: MPST_SS_DBG bits switch parts of the elimination step off (timing only, results are wrong)
};

struct SsProblem {
    int n, m, K0, p;     // columns, rows, wanted pairs, block width - of the COMPLEX matrix in pair mode
    int st;              // 2: the Gram matrix / Z / lam / res are those of the real embedding (pair mode), else 1
    bool active;
};
__device__ __forceinline__ SsProblem ss_resolve(const View& v, int lid, int going_left, const SsBufs& s) {
    const BtProblem pb = bt_resolve(v, lid, going_left, nullptr, 0);
    SsProblem q;
    q.st = pb.pair ? 2 : 1;
    q.n = pb.n / q.st;
    q.m = pb.rows / q.st;
    q.K0 = pb.K0 / q.st;
    // the block has to hold what the matrix holds before the update - M0 = W[lid] W[rid] has rank <= C chi(lid, rid) - and the kept
    // values, plus the oversampling; where the capacity of the Rayleigh-Ritz solver cuts it short the oversampling goes first
    const int r0 = min(pb.nspec / q.st, v.C * v.chi[lid + 1]);
    const int want = max(q.K0, r0);
    const int extra = (s.dbg & 16) ? 16 : SS_EXTRA;          // bring-up: MPST_SS_DBG=16 halves the oversampling
    q.p = min(s.pc, (want + extra + 15) & ~15);
    q.active = (pb.pair ? s.cx != 0 : s.cx == 0) && v.ss_bt != nullptr && want <= q.p && q.n >= 2 * q.p && q.st * q.n > MAX_DIM && q.m >= 1 &&
               q.K0 >= 1 && q.K0 <= 64;
    // fp32 tensors while the bond between the two sites is still growing: the kept values reach down to the cutoff, i.e. into the
    // rounding noise of the entries, whose spectrum does not decay - such a bond never passes (measured: every one of the first
    // half-sweep), so it is not attempted
    if (v.ss_f32 && q.st * v.chi[lid + 1] < v.chi_max && v.chi[lid + 1] < pb.nspec / q.st) q.active = false;
    return q;
}
__device__ __forceinline__ double ss_omega(int r, int j) {
    unsigned long long h = (unsigned long long)r * 0x9E3779B97F4A7C15ull + (unsigned long long)j * 0xC2B2AE3D27D4EB4Full;
    h ^= h >> 29;
    h *= 0xBF58476D1CE4E5B9ull;
    h ^= h >> 32;
    return (h & 1ull) ? 1.0 : -1.0;
}

// M as decomposeBT sees it, in fp64: going left rows (c, x) | cols y; going right rows (c, y) | cols x  (bt = [C][X][Y]).
// Omega into the first left block.  First launch of a solve: it also publishes whether the bond is attempted at all.
// start_right = 1 (fp32 bond tensors: one application of M fewer is enough for their 3e-8 bar): the start block is X = Omega / sqrt(n)
// itself - independent +-1 columns are orthonormal to 1 / sqrt(n), which is all the first product needs - written to the right block.
__global__ __launch_bounds__(256) void k_ss_load(View v, int lid, int going_left, SsBufs s, int start_right) {
    const SsProblem q = ss_resolve(v, lid, going_left, s);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        s.st[0] = 0;
        s.st[1] = q.active ? 1 : 0;
        if (q.active) s.st[3] += 1;
    }
    if (!q.active) return;
    const int n = q.n, m = q.m;
    const int Dl = v.chi[lid], Dr = v.chi[lid + 2];
    const int X = Dl * v.d, Y = v.d * Dr;
    const int64_t total = (int64_t)m * n;
    const float* bf = (const float*)v.ss_bt;
    const double* bd = (const double*)v.ss_bt;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        int64_t src;
        if (going_left) src = e;                      // (c, x, y) is the row-major order of M already
        else {
            const int row = (int)(e / n), x = (int)(e - (int64_t)row * n);
            const int c = row / Y, y = row - c * Y;
            src = ((int64_t)c * X + x) * Y + y;
        }
        s.Mw[e] = v.ss_f32 ? (double)bf[src] : bd[src];
    }
    const int64_t tot2 = (int64_t)(start_right ? n : m) * q.p;
    double* __restrict__ O = start_right ? s.Rb[1] : s.Lb[0];
    const double sc = start_right ? 1.0 / sqrt((double)n) : 1.0;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < tot2; e += (int64_t)gridDim.x * 256) {
        const int r = (int)(e / q.p), j = (int)(e - (int64_t)r * q.p);
        O[(int64_t)r * s.pc + j] = sc * ss_omega(r, j);
    }
}

// dst[R x p] = op(M) src:   TA = 0: R = m, K = n, op(M) = M;   TA = 1: R = n, K = m, op(M) = M^H (real: M^T).
// One 16 x 16 tile per workgroup, the contraction split over its 4 waves (k_tgram's scheme), partial tiles meet in LDS.
template <int TA>
__global__ __launch_bounds__(256) void k_ss_mm(View v, int lid, int going_left, SsBufs s, const double* __restrict__ src, double* __restrict__ dst) {
    __shared__ double part[4][256];
    if (!s.st[1]) return;
    const SsProblem q = ss_resolve(v, lid, going_left, s);
    const int n = q.n, m = q.m, p = q.p, ld = s.pc;
    const int R = TA ? n : m, K = TA ? m : n;
    const int tp = p >> 4;
    const int rt = (int)blockIdx.x / tp, ct = (int)blockIdx.x - rt * tp;
    if (rt * 16 >= R) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i16 = lane & 15, kq = lane >> 4;
    const int row = rt * 16 + i16, col = ct * 16 + i16;
    const bool rv = row < R;
    const int per = ((((K + 3) >> 2) + 3) >> 2) << 2;            // contraction range of a wave, a multiple of 4
    const int kbeg = wave * per, kend = min(K, kbeg + per);
    const double* __restrict__ Mw = s.Mw;
    d4 acc = {0, 0, 0, 0};
    for (int k0 = kbeg; k0 < kend; k0 += 64) {
        double a[16], b[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int k = k0 + 4 * u + kq;
            const bool kv = k < kend;
            a[u] = (kv && rv) ? (TA ? Mw[(int64_t)k * n + row] : Mw[(int64_t)row * n + k]) : 0.0;
            b[u] = kv ? src[(int64_t)k * ld + col] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u)
            acc = mfma_f64(a[u], b[u], acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) part[wave][r * 64 + lane] = acc[r];
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int orow = rt * 16 + kq + 4 * r;
            const double sum = (part[0][r * 64 + lane] + part[1][r * 64 + lane]) + (part[2][r * 64 + lane] + part[3][r * 64 + lane]);
            if (orow < R) dst[(int64_t)orow * ld + col] = sum;
        }
    }
}

// Gram partials of a block A [R x p]: Sp[ks][i][j] = sum over the rows of slice ks of A[r][i] A[r][j].
// left = 1: A is a left block (R = m), else a right block (R = n).  grid = (p/16)^2 * SS_KS.
__global__ __launch_bounds__(256) void k_ss_gram(View v, int lid, int going_left, SsBufs s, const double* __restrict__ A, int left) {
    __shared__ double part[4][256];
    if (!s.st[1]) return;
    const SsProblem q = ss_resolve(v, lid, going_left, s);
    const int p = q.p, ld = s.pc, R = left ? q.m : q.n;
    const int tp = p >> 4;
    const int ks = (int)blockIdx.x / (tp * tp), t = (int)blockIdx.x - ks * tp * tp;
    if (ks >= SS_KS || tp == 0) return;
    const int ti = t / tp, tj = t - ti * tp;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i16 = lane & 15, kq = lane >> 4;
    const int per = ((((R + 4 * SS_KS - 1) / (4 * SS_KS)) + 3) >> 2) << 2;        // rows per wave of a slice, a multiple of 4
    const int kbeg = (ks * 4 + wave) * per, kend = min(R, kbeg + per);
    d4 acc = {0, 0, 0, 0};
    for (int k0 = kbeg; k0 < kend; k0 += 64) {
        double a[16], b[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int k = k0 + 4 * u + kq;
            const bool kv = k < kend;
            a[u] = kv ? A[(int64_t)k * ld + ti * 16 + i16] : 0.0;
            b[u] = kv ? A[(int64_t)k * ld + tj * 16 + i16] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u)
            acc = mfma_f64(a[u], b[u], acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) part[wave][r * 64 + lane] = acc[r];
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = ti * 16 + kq + 4 * r, j = tj * 16 + i16;
            s.Sp[(int64_t)ks * ld * ld + (int64_t)i * ld + j] = (part[0][r * 64 + lane] + part[1][r * 64 + lane]) + (part[2][r * 64 + lane] + part[3][r * 64 + lane]);
        }
    }
}

// S = sum of the partials, S = L L^T by a Gauss-Jordan elimination that leaves the unit-lower inverse below the diagonal and the
// pivots on it; Tm = L^-T (Tm[l][j] = Linv[j][l], l <= j); a column whose pivot is below 2e-15 of its own squared norm is dropped
// (Tm[:, j] = 0).  One workgroup of FOUR waves - a step is a handful of fp64 instructions per element plus a fixed overhead per
// wave, and fp64 issues at a quarter rate: 16 waves spent 1900 cycles per step on the overhead alone (72 us), 4 waves with a
// 4 x 16 register tile each spend a few hundred.  Thread (rg = tid >> 3, cg = tid & 7) owns rows rg + 32 u (u = 0..3) of columns
// cg + 8 t (t = 0..15).  A step exchanges the pivot row and the pivot column through LDS (double-buffered: one barrier per step);
// the loop is cut into pieces of 8 steps with a compile-time pivot-column register, and finished rows cost no instructions.
constexpr int CH_T = 256, CH_RU = 4, CH_CT = 16;      // register tile at the capacity p = 128; RU x CT live (3 x 12 for p <= 96)
struct CholShared {
    double rowb[2][SS_PMAX], colb[2][SS_PMAX], pivd[SS_PMAX], d0[SS_PMAX];
    int dead[SS_PMAX];
};
template <int E, int S, int RU, int CT>      // steps 32 E + 8 S .. + 7: rows with u < E are finished; the pivot column sits in register t = 4 E + S
__device__ __forceinline__ void chol_piece(double (&val)[CH_RU][CH_CT], CholShared& sh, const int p, const int rg, const int cg, const int dbg) {
    constexpr int TP = 4 * E + S;
    const int j0 = 32 * E + 8 * S;
    const int jend = min(p, j0 + 8);
    for (int j = j0; j < jend; ++j) {
        const int b = j & 1;
        // everything a step reads from LDS is requested up front (one round trip, not a chain of them)
        const double piv = sh.rowb[b][j], dj = sh.d0[j];
        double fc[RU], r[CT];
#pragma unroll
        for (int u = E; u < RU; ++u) fc[u] = sh.colb[b][rg + 32 * u];
#pragma unroll
        for (int t = 0; t < CT; ++t) r[t] = (dbg & 4) ? 1e-3 : sh.rowb[b][cg + 8 * t];
        const bool isdead = !(piv > 2e-15 * dj) || !(dj > 0.0);       // uniform
        if (rg == 0 && cg == 0) {
            sh.pivd[j] = piv;
            sh.dead[j] = isdead ? 1 : 0;
        }
        double rc = __builtin_amdgcn_rcp(piv);
        rc = rc * (2.0 - piv * rc);
        rc = rc * (2.0 - piv * rc);
        if (isdead) rc = 0.0;           // dropped column: nothing is eliminated with it ...
        // the pivot column itself: val - f (piv + 1) = -f, the entry of the unit-lower inverse
        r[TP] += (cg == (j & 7)) ? 1.0 : 0.0;
#pragma unroll
        for (int u = E; u < RU; ++u) {
            const int i = rg + 32 * u;
            const double f = (u > E || i > j) ? fc[u] * rc : 0.0;
            if (!(dbg & 1)) {
#pragma unroll
                for (int t = 0; t < CT; ++t) val[u][t] -= f * r[t];
            } else val[u][TP] -= f * r[TP];
        }
        if (isdead && cg == (j & 7)) {  // ... and nothing of it survives below the diagonal
#pragma unroll
            for (int u = E; u < RU; ++u)
                if (rg + 32 * u > j) val[u][TP] = 0.0;
        }
        // publish row j + 1 and column j + 1 for the next step
        // (tried: the next step's operands first, its LDS reads under the bulk of this step's update - 44 against 40 us per
        // factorisation at p = 96: the second copy of the step's operands costs more registers than the overlap wins)
        const int jn = j + 1;
        if (jn < p && !(dbg & 2)) {
            if (rg == (jn & 31)) {
                const bool same = (jn >> 5) == E;
#pragma unroll
                for (int t = 0; t < CT; ++t) {
                    double a = val[E][t];
                    if constexpr (E + 1 < RU) {
                        if (!same) a = val[E + 1][t];
                    }
                    sh.rowb[b ^ 1][cg + 8 * t] = a;
                }
            }
            if (cg == (jn & 7)) {
                const bool same = (jn >> 3) == TP;
#pragma unroll
                for (int u = E; u < RU; ++u) {
                    double a = val[u][TP];
                    if constexpr (TP + 1 < CT) {
                        if (!same) a = val[u][TP + 1];
                    }
                    sh.colb[b ^ 1][rg + 32 * u] = a;
                }
            }
        }
        if (!(dbg & 8)) __syncthreads();
    }
}
template <int E, int RU, int CT>
__device__ __forceinline__ void chol_era(double (&val)[CH_RU][CH_CT], CholShared& sh, const int p, const int rg, const int cg, const int dbg) {
    chol_piece<E, 0, RU, CT>(val, sh, p, rg, cg, dbg);
    if (p > 32 * E + 8) chol_piece<E, 1, RU, CT>(val, sh, p, rg, cg, dbg);
    if (p > 32 * E + 16) chol_piece<E, 2, RU, CT>(val, sh, p, rg, cg, dbg);
    if (p > 32 * E + 24) chol_piece<E, 3, RU, CT>(val, sh, p, rg, cg, dbg);
}

// MODE 0 / 2: the factorisation above -> Tm (2: block capacity <= 96).  MODE 1: H = sum of the partials, zero padded to pc x pc (the Rayleigh-Ritz matrix).
template <int MODE>
__global__ __launch_bounds__(CH_T) void k_ss_chol(View v, int lid, int going_left, SsBufs s) {
    __shared__ CholShared sh;
    if (!s.st[1]) return;
    const SsProblem q = ss_resolve(v, lid, going_left, s);
    const int p = q.p, ld = s.pc, tid = threadIdx.x;
    const int64_t pp = (int64_t)ld * ld;
    if (MODE == 1) {
        // grid of (ld*ld / 256) workgroups; the partials of (i, j) and (j, i) are the same products in the same order: H is symmetric bit for bit
        const int e = (int)blockIdx.x * CH_T + tid;
        if (e < ld * ld) {
            const int i = e / ld, j = e - i * ld;
            double x = 0.0;
            if (i < p && j < p) {
#pragma unroll
                for (int k = 0; k < SS_KS; ++k) x += s.Sp[k * pp + e];
            }
            s.H[e] = x;
        }
        return;
    }
    const int rg = tid >> 3, cg = tid & 7;
    unsigned long long t0 = 0, t1 = 0, t2 = 0;
    if (tid == 0) t0 = __builtin_amdgcn_s_memrealtime();
    double val[CH_RU][CH_CT];
#pragma unroll
    for (int u = 0; u < CH_RU; ++u) {
        const int i = rg + 32 * u;
#pragma unroll
        for (int t = 0; t < CH_CT; ++t) {
            const int c = cg + 8 * t;
            double x = 0.0;
            if (i < p && c < p) {
#pragma unroll
                for (int k = 0; k < SS_KS; ++k) x += s.Sp[k * pp + (int64_t)i * ld + c];
            }
            val[u][t] = x;
            if (i == c) sh.d0[i] = x;
            if (i == 0) sh.rowb[0][c] = x;
            if (c == 0) sh.colb[0][i] = x;
        }
    }
    __syncthreads();
    if (tid == 0) t1 = __builtin_amdgcn_s_memrealtime();
    if constexpr (MODE == 2) {          // block capacity <= 96: 3 x 12 of the 4 x 16 registers are live
        chol_era<0, 3, 12>(val, sh, p, rg, cg, s.dbg);
        if (p > 32) chol_era<1, 3, 12>(val, sh, p, rg, cg, s.dbg);
        if (p > 64) chol_era<2, 3, 12>(val, sh, p, rg, cg, s.dbg);
    } else {
        chol_era<0, 4, 16>(val, sh, p, rg, cg, s.dbg);
        if (p > 32) chol_era<1, 4, 16>(val, sh, p, rg, cg, s.dbg);
        if (p > 64) chol_era<2, 4, 16>(val, sh, p, rg, cg, s.dbg);
        if (p > 96) chol_era<3, 4, 16>(val, sh, p, rg, cg, s.dbg);
    }
    if (tid == 0) {         // phase stamps (100 MHz) of the last call: load, elimination
        t2 = __builtin_amdgcn_s_memrealtime();
        s.lamH[SS_PMAX] = (double)(t1 - t0);
        s.lamH[SS_PMAX + 1] = (double)(t2 - t1);
    }
    // Tm[c][i] = Linv[i][c] = rsqrt(pivot_i) * (c == i ? 1 : c < i ? unit-lower inverse [i][c] : 0); dropped rows / columns are zero
#pragma unroll
    for (int u = 0; u < CH_RU; ++u) {
        const int i = rg + 32 * u;
        const bool live = i < p && !sh.dead[i];
        const double sc = live ? 1.0 / sqrt(sh.pivd[i]) : 0.0;
#pragma unroll
        for (int t = 0; t < CH_CT; ++t) {
            const int c = cg + 8 * t;
            if (i < ld && c < ld) s.Tm[(int64_t)c * ld + i] = (live && c <= i && !sh.dead[c]) ? (c == i ? sc : sc * val[u][t]) : 0.0;
        }
    }
}

// dst = src Tm  (rows x p times p x p): a 16 x 16 tile per wave.  left: the blocks are left blocks (R = m).
__global__ __launch_bounds__(256) void k_ss_apply(View v, int lid, int going_left, SsBufs s, const double* __restrict__ src, double* __restrict__ dst, int left) {
    if (!s.st[1]) return;
    const SsProblem q = ss_resolve(v, lid, going_left, s);
    const int p = q.p, ld = s.pc, R = left ? q.m : q.n;
    const int tp = p >> 4, tr = (R + 15) >> 4;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i16 = lane & 15, kq = lane >> 4;
    const int t = (int)blockIdx.x * 4 + wave;
    if (t >= tr * tp) return;
    const int rt = t / tp, ct = t - rt * tp;
    const int row = rt * 16 + i16, col = ct * 16 + i16;
    d4 acc = {0, 0, 0, 0};
    for (int k0 = 0; k0 < p; k0 += 64) {
        double a[16], b[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int k = k0 + 4 * u + kq;
            const bool kv = k < p;
            a[u] = (kv && row < R) ? src[(int64_t)row * ld + k] : 0.0;
            b[u] = kv ? s.Tm[(int64_t)k * ld + col] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u)
            acc = mfma_f64(a[u], b[u], acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int orow = rt * 16 + kq + 4 * r;
        if (orow < R) dst[(int64_t)orow * ld + col] = acc[r];
    }
}

// Ritz vectors: b.Z[k][c] = sum_l Zb[c][l] W[l][k] / sqrt(theta_k) for the K0 largest eigenpairs (theta, W) of H = Zb^T Zb;
// b.lam = theta; a pair whose theta is not positive, or that the small solver did not deliver, gets a zero vector (it is never kept).
__global__ __launch_bounds__(256) void k_ss_ritz(View v, int lid, int going_left, SsBufs s, BtBufs b, const double* __restrict__ Zb) {
    if (!s.st[1]) return;
    const SsProblem q = ss_resolve(v, lid, going_left, s);
    const int p = q.p, ld = s.pc, n = q.n, K0 = q.K0;
    const int tk = (K0 + 15) >> 4, tn = (n + 15) >> 4;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i16 = lane & 15, kq = lane >> 4;
    if (blockIdx.x == 0 && threadIdx.x < K0) b.lam[threadIdx.x] = fmax(s.lamH[threadIdx.x], 0.0);
    const int t = (int)blockIdx.x * 4 + wave;
    if (t >= tn * tk) return;
    const int rt = t / tk, kt = t - rt * tk;
    const int row = rt * 16 + i16, kc = kt * 16 + i16;
    d4 acc = {0, 0, 0, 0};
    for (int k0 = 0; k0 < p; k0 += 64) {
        double a[16], bb[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int k = k0 + 4 * u + kq;
            const bool kv = k < p;
            a[u] = (kv && row < n) ? Zb[(int64_t)row * ld + k] : 0.0;
            bb[u] = (kv && kc < K0) ? s.WH[(int64_t)k * ld + kc] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u)
            acc = mfma_f64(a[u], bb[u], acc);
    }
    const double th = kc < K0 ? s.lamH[kc] : 0.0;
    const double sc = th > 1e-13 * s.lamH[0] ? 1.0 / sqrt(th) : 0.0;      // below: the small solver delivered no vector (and nobody keeps one)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int c = rt * 16 + kq + 4 * r;
        if (c < n && kc < K0) b.Z[(int64_t)kc * b.ncap + c] = acc[r] * sc;
    }
}

// Residuals of the Ritz pairs against G itself: part[rt][k] = sum over the 16 rows of row tile rt of (G v_k - theta_k v_k)^2, and
// (k tile 0 only) part[tn*CAP_LIMIT + rt] = the rows' share of ||G||_F^2.  One tile per workgroup, contraction over its 4 waves.
// Pair mode: G is the 2n x 2n embedding, vector k is row 2k of Z (its partner J u has the same residual).
__global__ __launch_bounds__(256) void k_ss_resid(View v, int lid, int going_left, SsBufs s, BtBufs b) {
    __shared__ double part[4][256];
    __shared__ double fro[4];
    if (!s.st[1]) return;
    const SsProblem q = ss_resolve(v, lid, going_left, s);
    const int st = q.st, n = st * q.n, K0 = q.K0;
    const int tk = (K0 + 15) >> 4, tn = (n + 15) >> 4;
    const int rt = (int)blockIdx.x / tk, kt = (int)blockIdx.x - rt * tk;
    if (rt >= tn) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i16 = lane & 15, kq = lane >> 4;
    const int row = rt * 16 + i16, kc = kt * 16 + i16;
    const int per = ((((n + 3) >> 2) + 3) >> 2) << 2;
    const int kbeg = wave * per, kend = min(n, kbeg + per);
    const double* __restrict__ G = v.gram;
    d4 acc = {0, 0, 0, 0};
    double f2 = 0.0;
    for (int k0 = kbeg; k0 < kend; k0 += 64) {
        double a[16], bb[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int k = k0 + 4 * u + kq;
            const bool kv = k < kend;
            a[u] = (kv && row < n) ? G[(int64_t)row * n + k] : 0.0;
            bb[u] = (kv && kc < K0) ? b.Z[(int64_t)(st * kc) * b.ncap + k] : 0.0;
            f2 += a[u] * a[u];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u)
            acc = mfma_f64(a[u], bb[u], acc);
    }
    f2 = wave_sum(f2);
#pragma unroll
    for (int r = 0; r < 4; ++r) part[wave][r * 64 + lane] = acc[r];
    if (lane == 0) fro[wave] = f2;
    __syncthreads();
    if (wave == 0) {
        const double th = kc < K0 ? b.lam[st * kc] : 0.0;
        double col2 = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = rt * 16 + kq + 4 * r;
            const double gv = (part[0][r * 64 + lane] + part[1][r * 64 + lane]) + (part[2][r * 64 + lane] + part[3][r * 64 + lane]);
            const double vk = (c < n && kc < K0) ? b.Z[(int64_t)(st * kc) * b.ncap + c] : 0.0;
            const double rr = (c < n && kc < K0) ? gv - th * vk : 0.0;
            col2 += rr * rr;
        }
        // the 4 lane groups (kq) of column i16 hold its 16 rows: fixed-order sum
        col2 += __shfl_xor(col2, 16);
        col2 += __shfl_xor(col2, 32);
        if (kq == 0 && kc < K0) s.part[(int64_t)rt * CAP_LIMIT + kc] = col2;
        if (kt == 0 && lane == 0) s.part[(int64_t)tn * CAP_LIMIT + rt] = (fro[0] + fro[1]) + (fro[2] + fro[3]);
    }
}

// b.res[k] = |r_k| / sqrt(theta_k tr) (what k_bt_decide's subspace criterion sums), b.dd <- diag(G) (its trace check then compares
// G with itself), and the Frobenius certificate: res[CAP_LIMIT - 1] slot is not used - the verdict word ctl is left to k_bt_decide;
// a failed certificate is reported as an infinite residual.
__global__ __launch_bounds__(1024) void k_ss_collect(View v, int lid, int going_left, SsBufs s, BtBufs b) {
    __shared__ double red[3][16];
    if (!s.st[1]) return;
    const SsProblem q = ss_resolve(v, lid, going_left, s);
    const int st = q.st, n = st * q.n, K0 = q.K0, tid = threadIdx.x;
    const int tn = (n + 15) >> 4;
    const double* __restrict__ G = v.gram;
    // three sums in one pass: trace of G, ||G||_F^2 (row-tile pieces of k_ss_resid), ||H||_F^2 (pair mode: all three of the embeddings,
    // i.e. twice those of the complex matrices)
    double tr = 0.0, fr = 0.0, h2 = 0.0;
    for (int i = tid; i < n; i += 1024) {
        const double g = G[(int64_t)i * n + i];
        b.dd[i] = g;
        tr += g;
    }
    for (int i = tid; i < tn; i += 1024) fr += s.part[(int64_t)tn * CAP_LIMIT + i];
    const int hn = st * s.pc;
    for (int e = tid; e < hn * hn; e += 1024) h2 += s.H[e] * s.H[e];
    double r2 = 0.0, th = 0.0;
    if (tid < K0) {
        for (int t = 0; t < tn; ++t) r2 += s.part[(int64_t)t * CAP_LIMIT + tid];
        th = b.lam[st * tid];
    }
    tr = wave_sum(tr);
    fr = wave_sum(fr);
    h2 = wave_sum(h2);
    if ((tid & 63) == 0) {
        red[0][tid >> 6] = tr;
        red[1][tid >> 6] = fr;
        red[2][tid >> 6] = h2;
    }
    __syncthreads();
    tr = fr = h2 = 0.0;
    for (int w = 0; w < 16; ++w) {
        tr += red[0][w];
        fr += red[1][w];
        h2 += red[2][w];
    }
    const double inv_st = 1.0 / (double)st;
    tr *= inv_st;
    fr *= inv_st;
    h2 *= inv_st;
    // unseen part of the spectrum: ||G||_F^2 - ||H||_F^2 >= sum of the squares of what the block missed (H is a compression of
    // M M^H, whose non-zero spectrum is that of G); below the resolution of the subtraction (64 eps ||G||_F^2) nothing can be said
    // and nothing is claimed
    const double unseen2 = fmax(fr - h2, 0.0);
    const double floor2 = 64.0 * 2.2e-16 * fr;
    const bool rr_failed = s.rrflag && *(const volatile int32_t*)s.rrflag != 0;      // complex: the Hermitian Rayleigh-Ritz solve gave up
    if (tid < K0) {
        // fp32 bond tensors carry 6e-8 of rounding noise per entry: 3e-8 of the tensor's norm is the bar there, 1e-9 in fp64
        // (k_bt_decide compares the root of the sum of squares with 1e-9)
        double rel = (th > 0.0 && tr > 0.0) ? sqrt(r2 / (th * tr)) * (v.ss_f32 ? 1.0 / 30.0 : 1.0) : 0.0;
        // an eigenvalue the block missed and that is larger than this kept one: the certificate fails for it
        if (unseen2 > floor2 && unseen2 > th * th) rel = th > 1e-10 * tr ? 1e300 : rel;
        if (rr_failed) rel = 1e300;
        b.res[st * tid] = rel;
        if (st == 2) b.res[2 * tid + 1] = 0.0;          // the partner J u: its residual is the same one, counted once
    }
}

// the subspace phase's verdict, after its k_bt_polish rounds: st[0] = 1 if the result was published (the exact path leaves)
__global__ void k_ss_verdict(SsBufs s, BtBufs b) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (!s.st[1]) return;
    const int ok = (*b.flag == 0) ? 1 : 0;
    s.st[0] = ok;
    if (ok) s.st[2] += 1;
}


// =====================================================================================================================================
// Blocked form of the orthonormalisation's factorisation (real): S = L L^T with 16 x 16 blocks.  The pivot-by-pivot elimination above
// pays a barrier, an LDS round trip and the reciprocal chain for each of p pivots (0.4 us each); here the dependent chain is the nb = p / 16
// diagonal blocks, each factored AND inverted by ONE wave in registers (lane = row; v_readlane broadcasts, no LDS, no barrier:
// ~1000 instructions), everything else - the updates of a block column and its scaling by L_JJ^-T - on the MFMA with operands from LDS.
// What leaves is Tm = L^-T, formed by a block forward substitution on the MFMA as well (measured on the way: leaving L and doing the
// substitution per row tile in the apply kernel made that kernel latency-bound, 15 against 7 us).
// A pivot below 2e-15 of its column's own squared norm drops the column: its l_kk counts as infinite (inverse diagonal 0), its
// sub-column as 0 - Q gets a zero column, as with the pivot-by-pivot kernel.
// =====================================================================================================================================
constexpr int CB_LDPAD = 4;        // LDS row stride = pc + 4 doubles: (4 row + k) mod 32 distinct over a wave's operand read

// 16 x 16 block at LDS address T (row stride ldl): Cholesky factor into its lower triangle, inverse of the factor to Li (row stride 16,
// lower triangle, zeros above).  One wave; lanes 0..15 own a row each.  d0: the block's original diagonal of S (drop criterion).
__device__ __forceinline__ void cb_diag_block(double* __restrict__ T, int ldl, double* __restrict__ Li, const double* __restrict__ d0, int lane) {
    const int r = lane & 15;
    double a[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) a[c] = T[r * ldl + c];
    const double dr = d0[r];
    double inv[16];                                          // 1 / l_kk (0: dropped)
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const double piv = readlane_f64(a[k], k);
        const double dk = readlane_f64(dr, k);
        const bool dead = !(piv > 2e-15 * dk) || !(dk > 0.0);
        double rs = __builtin_amdgcn_rsq(piv);                  // v_rsq_f64 + two Newton steps: the IEEE sqrt and division cost 60 instructions a pivot
        rs = rs * (1.5 - 0.5 * piv * rs * rs);
        rs = rs * (1.5 - 0.5 * piv * rs * rs);
        if (dead) rs = 0.0;
        inv[k] = rs;
        const double lk = a[k] * rs;                         // l_rk for r >= k (l_kk = sqrt(piv)); a dropped column is zero
        a[k] = lk;
#pragma unroll
        for (int c = k + 1; c < 16; ++c) a[c] -= lk * readlane_f64(lk, c);
    }
    // inverse of the lower-triangular factor, column by column: lane c owns column c of X = L^-1
    double x[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) x[q] = 0.0;
    const int c = r;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        // row q of X: x_qc = (delta_qc - sum_{j < q} l_qj x_jc) / l_qq
        double sacc = (q == c) ? 1.0 : 0.0;
#pragma unroll
        for (int j = 0; j < q; ++j) sacc -= readlane_f64(a[j], q) * x[j];
        x[q] = (q >= c) ? sacc * inv[q] : 0.0;
    }
    if (lane < 16) {
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) T[r * ldl + cc] = cc <= r ? a[cc] : 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) Li[q * 16 + c] = x[q];
    }
}

// Tm = L^-T (as k_ss_chol leaves it: Tm[l][j] = Linv[j][l], l <= j, dropped columns zero) from the Gram partials.  One workgroup of
// 16 waves (all of them load and store, up to nb of them work on a block column), the matrix in LDS.  After the factorisation the
// inverse is formed block column by block column, X_IJ = -Linv_II sum_{K=J}^{I-1} L_IK X_KJ (a wave per block column, MFMA, operands
// from LDS), and kept TRANSPOSED in the unused upper triangle - which is exactly the layout of Tm.
constexpr int CB_T = 512;
__global__ __launch_bounds__(CB_T) void k_ss_cholb(View v, int lid, int going_left, SsBufs s) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    if (!s.st[1]) return;
    const SsProblem q = ss_resolve(v, lid, going_left, s);
    const int p = q.p, ld = s.pc, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i16 = lane & 15, kq = lane >> 4;
    const int ldl = ld + CB_LDPAD, nb = p >> 4;
    constexpr int NW = CB_T / 64;
    double* A = sm;                                 // [pc][ldl]
    double* d0 = A + (size_t)ld * ldl;              // [pc]
    double* Ld = d0 + ld;                           // [nb][16][16] inverses of the diagonal blocks
    const int64_t pp = (int64_t)ld * ld;
    unsigned long long tq[6] = {0, 0, 0, 0, 0, 0}, tl = 0;
    auto stamp = [&](int k) {
        if (tid == 0) {
            const unsigned long long t = __builtin_amdgcn_s_memrealtime();
            tq[k] += t - tl;
            tl = t;
        }
    };
    if (tid == 0) tl = __builtin_amdgcn_s_memrealtime();
    {
        const int j = tid & (SS_PMAX - 1);
        for (int i = tid >> 7; i < p; i += CB_T / SS_PMAX) {
            if (j < p) {
                double x = 0.0;
#pragma unroll
                for (int k = 0; k < SS_KS; ++k) x += s.Sp[k * pp + (int64_t)i * ld + j];
                A[i * ldl + j] = x;
                if (i == j) d0[i] = x;
            }
        }
    }
    __syncthreads();
    stamp(0);
    for (int J = 0; J < nb; ++J) {
        // (1) block column J, rows I >= J: T_IJ = S_IJ - sum_{K < J} L_IK L_JK^T   (left-looking; the waves share out the block rows)
        for (int I = J + wave; I < nb; I += NW) {
            d4 acc;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = A[(16 * I + kq + 4 * r) * ldl + 16 * J + i16];
            for (int K = 0; K < J; ++K) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const double a = A[(16 * I + i16) * ldl + 16 * K + 4 * u + kq];
                    const double b = A[(16 * J + i16) * ldl + 16 * K + 4 * u + kq];
                    acc = mfma_f64(-a, b, acc);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) A[(16 * I + kq + 4 * r) * ldl + 16 * J + i16] = acc[r];
        }
        __syncthreads();
        stamp(1);
        // (2) the diagonal block: factor + inverse, one wave
        if (wave == 0) cb_diag_block(A + (16 * J) * ldl + 16 * J, ldl, Ld + J * 256, d0 + 16 * J, lane);
        __syncthreads();
        stamp(2);
        // (3) L_IJ = T_IJ Linv_JJ^T for I > J
        for (int I = J + 1 + wave; I < nb; I += NW) {
            d4 acc = {0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double a = A[(16 * I + i16) * ldl + 16 * J + 4 * u + kq];
                const double b = Ld[J * 256 + i16 * 16 + 4 * u + kq];              // (Linv^T)[k][c] = Linv[c][k]
                acc = mfma_f64(a, b, acc);
            }
            // (in place: a wave's LDS operations execute in program order and the stores depend on every load through the MFMAs)
#pragma unroll
            for (int r = 0; r < 4; ++r) A[(16 * I + kq + 4 * r) * ldl + 16 * J + i16] = acc[r];
        }
        __syncthreads();
        stamp(3);
    }
    // (4) X = L^-1, block column J on wave J: X_IJ^T goes to block (J, I) of the upper triangle
    if (wave < nb) {
        const int J = wave;
        for (int I = J + 1; I < nb; ++I) {
            d4 acc = {0, 0, 0, 0};
            for (int K = J; K < I; ++K) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const double a = A[(16 * I + i16) * ldl + 16 * K + 4 * u + kq];                        // L_IK[i][k]
                    const double b = (K == J) ? Ld[J * 256 + (4 * u + kq) * 16 + i16]                       // X_JJ[k][c]
                                              : A[(16 * J + i16) * ldl + 16 * K + 4 * u + kq];              // X_KJ[k][c], stored transposed
                    acc = mfma_f64(a, b, acc);
                }
            }
            d4 out = {0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < 4; ++u) out = mfma_f64(-Ld[I * 256 + i16 * 16 + 4 * u + kq], acc[u], out);    // -Linv_II (.)
#pragma unroll
            for (int r = 0; r < 4; ++r) A[(16 * J + i16) * ldl + 16 * I + kq + 4 * r] = out[r];               // X_IJ[i = kq + 4r][c = i16], transposed
        }
    }
    __syncthreads();
    stamp(4);
    // Tm[l][j] = Linv[j][l]: the upper triangle as it stands, the diagonal blocks from Ld (transposed), zeros below
    {
        const int j = tid & (SS_PMAX - 1);
        for (int l = tid >> 7; l < ld; l += CB_T / SS_PMAX) {
            if (j < ld) {
                double x = 0.0;
                if (l < p && j < p && l <= j) x = ((l >> 4) == (j >> 4)) ? Ld[(j >> 4) * 256 + (j & 15) * 16 + (l & 15)] : A[l * ldl + j];
                s.Tm[(int64_t)l * ld + j] = x;
            }
        }
    }
    stamp(5);
    if (tid == 0)
        for (int k = 0; k < 6; ++k) s.lamH[SS_PMAX + 2 + k] = (double)tq[k];       // phase stamps (100 MHz) of the last call
}

// =====================================================================================================================================
// Complex element types (Fourier / Sahand / Stoudenmire encodings, BASELINE configs[4]): the same sequence on the complex m x n
// matrix itself.  Blocks, Gram partials and the triangular factor are double2 arrays; a complex product is four real MFMAs.  The
// Rayleigh-Ritz problem is Hermitian (p x p): its eigenpairs come from the pair-mode machinery of this file - k_bt_coop_c reduces
// it with complex reflectors to a real tridiagonal matrix, k_bt_vec / k_bt_back_c deliver the vectors in the embedding layout -
// on a small workspace of its own (raw mode, gated by st[1]).  What leaves this phase is what the exact pair-mode path leaves:
// Z rows 2k = (Re v, Im v), 2k + 1 = J v; lam twice; the shared verification kernels run unchanged.
// =====================================================================================================================================
__device__ __forceinline__ void cmma(double ar, double ai, double br, double bi, d4& cR, d4& cI) {
    cR = mfma_f64(ar, br, cR);
    cR = mfma_f64(-ai, bi, cR);
    cI = mfma_f64(ar, bi, cI);
    cI = mfma_f64(ai, br, cI);
}

__global__ __launch_bounds__(256) void k_ss_load_c(View v, int lid, int going_left, SsBufs s, int start_right) {
    const SsProblem q = ss_resolve(v, lid, going_left, s);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        s.st[0] = 0;
        s.st[1] = q.active ? 1 : 0;
        if (q.active) s.st[3] += 1;
    }
    if (!q.active) return;
    const int n = q.n, m = q.m;
    const int Dl = v.chi[lid], Dr = v.chi[lid + 2];
    const int X = Dl * v.d, Y = v.d * Dr;
    const int64_t total = (int64_t)m * n;
    const float2* bf = (const float2*)v.ss_bt;
    const double2* bd = (const double2*)v.ss_bt;
    double2* __restrict__ Mw = (double2*)s.Mw;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        int64_t src;
        if (going_left) src = e;
        else {
            const int row = (int)(e / n), x = (int)(e - (int64_t)row * n);
            const int c = row / Y, y = row - c * Y;
            src = ((int64_t)c * X + x) * Y + y;
        }
        // going right the decomposed matrix is the plain transpose of the [X][Y] slices (rows (c, y), columns x), no conjugate:
        // M^H M = sum_c conj(B_c) B_c^T is the Gram matrix k_tgram forms in that direction
        Mw[e] = v.ss_f32 ? make_double2((double)bf[src].x, (double)bf[src].y) : bd[src];
    }
    const int64_t tot2 = (int64_t)(start_right ? n : m) * q.p;
    double2* __restrict__ O = (double2*)(start_right ? s.Rb[1] : s.Lb[0]);
    const double sc = start_right ? 1.0 / sqrt((double)n) : 1.0;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < tot2; e += (int64_t)gridDim.x * 256) {
        const int r = (int)(e / q.p), j = (int)(e - (int64_t)r * q.p);
        O[(int64_t)r * s.pc + j] = make_double2(sc * ss_omega(r, j), 0.0);
    }
}

// dst = op(M) src.  TA = 0: M src;  TA = 1: M^H src.  cj = 1: M stands for conj(Mw) (going right, see k_ss_load_c).
template <int TA>
__global__ __launch_bounds__(256) void k_ss_mm_c(View v, int lid, int going_left, SsBufs s, const double2* __restrict__ src, double2* __restrict__ dst, int cj) {
    __shared__ double part[2][4][256];
    if (!s.st[1]) return;
    const SsProblem q = ss_resolve(v, lid, going_left, s);
    const int n = q.n, m = q.m, p = q.p, ld = s.pc;
    const int R = TA ? n : m, K = TA ? m : n;
    const int tp = p >> 4;
    const int rt = (int)blockIdx.x / tp, ct = (int)blockIdx.x - rt * tp;
    if (rt * 16 >= R) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i16 = lane & 15, kq = lane >> 4;
    const int row = rt * 16 + i16, col = ct * 16 + i16;
    const bool rv = row < R;
    const int per = ((((K + 3) >> 2) + 3) >> 2) << 2;
    const int kbeg = wave * per, kend = min(K, kbeg + per);
    const double2* __restrict__ Mw = (const double2*)s.Mw;
    const double sg = ((TA != 0) != (cj != 0)) ? -1.0 : 1.0;         // conjugate of the stored entry: M^H of M, or M of conj(M)
    d4 cR = {0, 0, 0, 0}, cI = {0, 0, 0, 0};
    for (int k0 = kbeg; k0 < kend; k0 += 32) {
        double2 a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = k0 + 4 * u + kq;
            const bool kv = k < kend;
            a[u] = (kv && rv) ? (TA ? Mw[(int64_t)k * n + row] : Mw[(int64_t)row * n + k]) : make_double2(0.0, 0.0);
            b[u] = kv ? src[(int64_t)k * ld + col] : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (k0 + 4 * u < kend) cmma(a[u].x, sg * a[u].y, b[u].x, b[u].y, cR, cI);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        part[0][wave][r * 64 + lane] = cR[r];
        part[1][wave][r * 64 + lane] = cI[r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int orow = rt * 16 + kq + 4 * r;
            const double xr = (part[0][0][r * 64 + lane] + part[0][1][r * 64 + lane]) + (part[0][2][r * 64 + lane] + part[0][3][r * 64 + lane]);
            const double xi = (part[1][0][r * 64 + lane] + part[1][1][r * 64 + lane]) + (part[1][2][r * 64 + lane] + part[1][3][r * 64 + lane]);
            if (orow < R) dst[(int64_t)orow * ld + col] = make_double2(xr, xi);
        }
    }
}

// Gram partials S = A^H A of a complex block
__global__ __launch_bounds__(256) void k_ss_gram_c(View v, int lid, int going_left, SsBufs s, const double2* __restrict__ A, int left) {
    __shared__ double part[2][4][256];
    if (!s.st[1]) return;
    const SsProblem q = ss_resolve(v, lid, going_left, s);
    const int p = q.p, ld = s.pc, R = left ? q.m : q.n;
    const int tp = p >> 4;
    const int ks = (int)blockIdx.x / (tp * tp), t = (int)blockIdx.x - ks * tp * tp;
    if (ks >= SS_KS || tp == 0) return;
    const int ti = t / tp, tj = t - ti * tp;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i16 = lane & 15, kq = lane >> 4;
    const int per = ((((R + 4 * SS_KS - 1) / (4 * SS_KS)) + 3) >> 2) << 2;
    const int kbeg = (ks * 4 + wave) * per, kend = min(R, kbeg + per);
    d4 cR = {0, 0, 0, 0}, cI = {0, 0, 0, 0};
    for (int k0 = kbeg; k0 < kend; k0 += 32) {
        double2 a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = k0 + 4 * u + kq;
            const bool kv = k < kend;
            a[u] = kv ? A[(int64_t)k * ld + ti * 16 + i16] : make_double2(0.0, 0.0);
            b[u] = kv ? A[(int64_t)k * ld + tj * 16 + i16] : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (k0 + 4 * u < kend) cmma(a[u].x, -a[u].y, b[u].x, b[u].y, cR, cI);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        part[0][wave][r * 64 + lane] = cR[r];
        part[1][wave][r * 64 + lane] = cI[r];
    }
    __syncthreads();
    if (wave == 0) {
        double2* __restrict__ Sp = (double2*)s.Sp;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = ti * 16 + kq + 4 * r, j = tj * 16 + i16;
            const double xr = (part[0][0][r * 64 + lane] + part[0][1][r * 64 + lane]) + (part[0][2][r * 64 + lane] + part[0][3][r * 64 + lane]);
            const double xi = (part[1][0][r * 64 + lane] + part[1][1][r * 64 + lane]) + (part[1][2][r * 64 + lane] + part[1][3][r * 64 + lane]);
            Sp[(int64_t)ks * ld * ld + (int64_t)i * ld + j] = make_double2(xr, xi);
        }
    }
}

// The elimination for a Hermitian positive definite S (pivots real): 3 x 12 complex register tile per thread, block capacity <= 96.
struct CholSharedC {
    double2 rowb[2][96], colb[2][96];
    double pivd[96], d0[96];
    int dead[96];
};
template <int E, int S>
__device__ __forceinline__ void chol_piece_c(double2 (&val)[3][12], CholSharedC& sh, const int p, const int rg, const int cg) {
    constexpr int TP = 4 * E + S, RU = 3, CT = 12;
    const int j0 = 32 * E + 8 * S;
    const int jend = min(p, j0 + 8);
    for (int j = j0; j < jend; ++j) {
        const int b = j & 1;
        const double piv = sh.rowb[b][j].x, dj = sh.d0[j];
        double2 fc[RU], r[CT];
#pragma unroll
        for (int u = E; u < RU; ++u) fc[u] = sh.colb[b][rg + 32 * u];
#pragma unroll
        for (int t = 0; t < CT; ++t) r[t] = sh.rowb[b][cg + 8 * t];
        const bool isdead = !(piv > 2e-15 * dj) || !(dj > 0.0);
        if (rg == 0 && cg == 0) {
            sh.pivd[j] = piv;
            sh.dead[j] = isdead ? 1 : 0;
        }
        double rc = __builtin_amdgcn_rcp(piv);
        rc = rc * (2.0 - piv * rc);
        rc = rc * (2.0 - piv * rc);
        if (isdead) rc = 0.0;
        if (cg == (j & 7)) r[TP] = make_double2(piv + 1.0, 0.0);      // val - f (piv + 1) = -f (the pivot itself is real: its rounding-level imaginary part goes)
#pragma unroll
        for (int u = E; u < RU; ++u) {
            const int i = rg + 32 * u;
            const bool on = u > E || i > j;
            const double fr = on ? fc[u].x * rc : 0.0, fi = on ? fc[u].y * rc : 0.0;
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                val[u][t].x -= fr * r[t].x - fi * r[t].y;
                val[u][t].y -= fr * r[t].y + fi * r[t].x;
            }
        }
        if (isdead && cg == (j & 7)) {
#pragma unroll
            for (int u = E; u < RU; ++u)
                if (rg + 32 * u > j) val[u][TP] = make_double2(0.0, 0.0);
        }
        const int jn = j + 1;
        if (jn < p) {
            if (rg == (jn & 31)) {
                const bool same = (jn >> 5) == E;
#pragma unroll
                for (int t = 0; t < CT; ++t) {
                    double2 a = val[E][t];
                    if constexpr (E + 1 < RU) {
                        if (!same) a = make_double2(val[E + 1][t].x, val[E + 1][t].y);
                    }
                    sh.rowb[b ^ 1][cg + 8 * t] = a;
                }
            }
            if (cg == (jn & 7)) {
                const bool same = (jn >> 3) == TP;
#pragma unroll
                for (int u = E; u < RU; ++u) {
                    double2 a = val[u][TP];
                    if constexpr (TP + 1 < CT) {
                        if (!same) a = make_double2(val[u][TP + 1].x, val[u][TP + 1].y);
                    }
                    sh.colb[b ^ 1][rg + 32 * u] = a;
                }
            }
        }
        __syncthreads();
    }
}
template <int E>
__device__ __forceinline__ void chol_era_c(double2 (&val)[3][12], CholSharedC& sh, const int p, const int rg, const int cg) {
    chol_piece_c<E, 0>(val, sh, p, rg, cg);
    if (p > 32 * E + 8) chol_piece_c<E, 1>(val, sh, p, rg, cg);
    if (p > 32 * E + 16) chol_piece_c<E, 2>(val, sh, p, rg, cg);
    if (p > 32 * E + 24) chol_piece_c<E, 3>(val, sh, p, rg, cg);
}
// MODE 0: Tm = L^-H of S = sum of the partials = L L^H.  MODE 1: H = sum of the partials, written as the real embedding
// [[Hr, -Hi], [Hi, Hr]] of order 2 pc (zero padded) - what the pair-mode Hermitian solver reads.
template <int MODE>
__global__ __launch_bounds__(CH_T) void k_ss_chol_c(View v, int lid, int going_left, SsBufs s) {
    __shared__ CholSharedC sh;
    if (!s.st[1]) return;
    const SsProblem q = ss_resolve(v, lid, going_left, s);
    const int p = q.p, ld = s.pc, tid = threadIdx.x;
    const int64_t pp = (int64_t)ld * ld;
    const double2* __restrict__ Sp = (const double2*)s.Sp;
    if (MODE == 1) {
        const int e = (int)blockIdx.x * CH_T + tid;
        if (e < ld * ld) {
            const int i = e / ld, j = e - i * ld;
            double xr = 0.0, xi = 0.0;
            if (i < p && j < p) {
#pragma unroll
                for (int k = 0; k < SS_KS; ++k) {
                    xr += Sp[k * pp + e].x;
                    xi += Sp[k * pp + e].y;
                }
                if (i == j) xi = 0.0;
            }
            const int hn = 2 * ld;
            s.H[(int64_t)i * hn + j] = xr;
            s.H[(int64_t)i * hn + ld + j] = -xi;
            s.H[(int64_t)(ld + i) * hn + j] = xi;
            s.H[(int64_t)(ld + i) * hn + ld + j] = xr;
        }
        return;
    }
    const int rg = tid >> 3, cg = tid & 7;
    double2 val[3][12];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int i = rg + 32 * u;
#pragma unroll
        for (int t = 0; t < 12; ++t) {
            const int c = cg + 8 * t;
            double xr = 0.0, xi = 0.0;
            if (i < p && c < p) {
#pragma unroll
                for (int k = 0; k < SS_KS; ++k) {
                    xr += Sp[k * pp + (int64_t)i * ld + c].x;
                    xi += Sp[k * pp + (int64_t)i * ld + c].y;
                }
            }
            if (i == c) xi = 0.0;
            val[u][t] = make_double2(xr, xi);
            if (i == c) sh.d0[i] = xr;
            if (i == 0) sh.rowb[0][c] = make_double2(xr, xi);
            if (c == 0) sh.colb[0][i] = make_double2(xr, xi);
        }
    }
    __syncthreads();
    chol_era_c<0>(val, sh, p, rg, cg);
    if (p > 32) chol_era_c<1>(val, sh, p, rg, cg);
    if (p > 64) chol_era_c<2>(val, sh, p, rg, cg);
    // Tm[c][i] = conj(Linv[i][c]),  Linv[i][c] = rsqrt(pivot_i) * (c == i ? 1 : c < i ? unit-lower inverse [i][c] : 0)
    double2* __restrict__ Tm = (double2*)s.Tm;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int i = rg + 32 * u;
        const bool live = i < p && !sh.dead[i];
        const double sc = live ? 1.0 / sqrt(sh.pivd[i]) : 0.0;
#pragma unroll
        for (int t = 0; t < 12; ++t) {
            const int c = cg + 8 * t;
            if (i < ld && c < ld) {
                const bool on = live && c <= i && !sh.dead[c];
                Tm[(int64_t)c * ld + i] = on ? (c == i ? make_double2(sc, 0.0) : make_double2(sc * val[u][t].x, -sc * val[u][t].y)) : make_double2(0.0, 0.0);
            }
        }
    }
}

// dst = src Tm (complex)
__global__ __launch_bounds__(256) void k_ss_apply_c(View v, int lid, int going_left, SsBufs s, const double2* __restrict__ src, double2* __restrict__ dst, int left) {
    if (!s.st[1]) return;
    const SsProblem q = ss_resolve(v, lid, going_left, s);
    const int p = q.p, ld = s.pc, R = left ? q.m : q.n;
    const int tp = p >> 4, tr = (R + 15) >> 4;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i16 = lane & 15, kq = lane >> 4;
    const int t = (int)blockIdx.x * 4 + wave;
    if (t >= tr * tp) return;
    const int rt = t / tp, ct = t - rt * tp;
    const int row = rt * 16 + i16, col = ct * 16 + i16;
    const double2* __restrict__ Tm = (const double2*)s.Tm;
    d4 cR = {0, 0, 0, 0}, cI = {0, 0, 0, 0};
    for (int k0 = 0; k0 < p; k0 += 32) {
        double2 a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = k0 + 4 * u + kq;
            const bool kv = k < p;
            a[u] = (kv && row < R) ? src[(int64_t)row * ld + k] : make_double2(0.0, 0.0);
            b[u] = kv ? Tm[(int64_t)k * ld + col] : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (k0 + 4 * u < p) cmma(a[u].x, a[u].y, b[u].x, b[u].y, cR, cI);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int orow = rt * 16 + kq + 4 * r;
        if (orow < R) dst[(int64_t)orow * ld + col] = make_double2(cR[r], cI[r]);
    }
}

// Ritz vectors of the complex problem: v_k = Zb W[:, k] / sqrt(theta_k); (theta, W) from the pair-mode solve of the embedding of H
// (rawlam[2k], vector k in column 2k of rawE: rows [0, pc) real parts, [pc, 2 pc) imaginary parts).  cj: the iteration ran on conj(M)
// (going right): the eigenvectors of G = conj(M^H M) are the conjugates.  Z rows 2k = (Re v, Im v), 2k + 1 = (-Im v, Re v); lam twice.
__global__ __launch_bounds__(256) void k_ss_ritz_c(View v, int lid, int going_left, SsBufs s, BtBufs b, const double2* __restrict__ Zb, int cj) {
    if (!s.st[1]) return;
    const SsProblem q = ss_resolve(v, lid, going_left, s);
    const int p = q.p, ld = s.pc, n = q.n, K0 = q.K0, hn = 2 * s.pc;
    const int tk = (K0 + 15) >> 4, tn = (n + 15) >> 4;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i16 = lane & 15, kq = lane >> 4;
    if (blockIdx.x == 0 && threadIdx.x < K0) {
        const double th = fmax(s.lamH[2 * threadIdx.x], 0.0);
        b.lam[2 * threadIdx.x] = th;
        b.lam[2 * threadIdx.x + 1] = th;
    }
    const int t = (int)blockIdx.x * 4 + wave;
    if (t >= tn * tk) return;
    const int rt = t / tk, kt = t - rt * tk;
    const int row = rt * 16 + i16, kc = kt * 16 + i16;
    d4 cR = {0, 0, 0, 0}, cI = {0, 0, 0, 0};
    for (int k0 = 0; k0 < p; k0 += 32) {
        double2 a[8], w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = k0 + 4 * u + kq;
            const bool kv = k < p;
            a[u] = (kv && row < n) ? Zb[(int64_t)row * ld + k] : make_double2(0.0, 0.0);
            w[u] = (kv && kc < K0) ? make_double2(s.WH[(int64_t)k * hn + 2 * kc], s.WH[(int64_t)(ld + k) * hn + 2 * kc]) : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (k0 + 4 * u < p) cmma(a[u].x, a[u].y, w[u].x, w[u].y, cR, cI);
    }
    const double th = kc < K0 ? s.lamH[2 * kc] : 0.0;
    const double sc = th > 1e-13 * s.lamH[0] ? 1.0 / sqrt(th) : 0.0;
    const double sgn = cj ? -1.0 : 1.0;
    const int64_t ldz = b.ncap;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int c = rt * 16 + kq + 4 * r;
        if (c < n && kc < K0) {
            const double xr = cR[r] * sc, xi = sgn * cI[r] * sc;
            b.Z[(int64_t)(2 * kc) * ldz + c] = xr;
            b.Z[(int64_t)(2 * kc) * ldz + n + c] = xi;
            b.Z[(int64_t)(2 * kc + 1) * ldz + c] = -xi;
            b.Z[(int64_t)(2 * kc + 1) * ldz + n + c] = xr;
        }
    }
}
