// Prototype: the product tridiagonalisation with TWO rows per thread (thread = rows 2rp, 2rp+1 x 8 columns instead of one row
// x 16 columns): every LDS operand (v and p at a thread's columns) serves two rows, which halves the operand traffic that
// bounds the first ~48 steps; the row sums are 16-lane instead of 8-lane reductions.  Everything else - helper wave,
// look-ahead reflector, deferred stores - is the product kernel's (generated from mpst_eig.hip by the substitutions in
// lab/ubench/make_tri_rows2.py).
#pragma once
namespace mpst {
constexpr int QN2 = 16;        // threads per row pair
constexpr int NP2 = 4;         // column pairs per thread and row
__global__ __launch_bounds__(TRI_T) void k_eig_tri2(View v, int lid, int going_left, const double* rawG, int rawn,
                                                         int rawalg, double* __restrict__ ws,
                                                         unsigned long long* stamps) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const EigProblem pb = resolve(v, lid, going_left, rawG, rawn, rawalg);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (!pb.tri) {
        if (tid == 0) ws[WS_MISC + 3] = 0.0;
        return;
    }
    const double* __restrict__ G = pb.G;
    const int n = pb.n;
    TriShared t = tri_carve(smem);
    if (stamps && tid == 0) {
        stamps[0] = __builtin_amdgcn_s_memrealtime();
        stamps[6] = __builtin_readcyclecounter();
    }
    // ---- load: thread (rp, q) owns rows r0 = 2 rp, r1 = r0 + 1 at the column pairs c = 2q + 32k + {0,1}, k = 0..3:
    // two rows share every LDS operand, which halves the operand traffic that bounds the early steps
    const int rp = tid / QN2, q = tid % QN2;
    const int r0 = 2 * rp, r1 = r0 + 1;
    double A0[2 * NP2], A1[2 * NP2];
#pragma unroll
    for (int k = 0; k < NP2; ++k) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = 2 * q + 2 * QN2 * k + h;
            A0[2 * k + h] = (r0 < n && c < n) ? G[(size_t)r0 * n + c] : 0.0;
            A1[2 * k + h] = (r1 < n && c < n) ? G[(size_t)r1 * n + c] : 0.0;
        }
    }
    if (tid < 256) {
        t.xs[tid] = 0.0;
        t.ps[tid] = 0.0;
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
        double* x = xrb + (row & 1) * 128;
        if (r0 == row) {
#pragma unroll
            for (int k = 0; k < NP2; ++k) *(double2*)&x[2 * q + 2 * QN2 * k] = make_double2(A0[2 * k], A0[2 * k + 1]);
        }
        if (r1 == row) {
#pragma unroll
            for (int k = 0; k < NP2; ++k) *(double2*)&x[2 * q + 2 * QN2 * k] = make_double2(A1[2 * k], A1[2 * k + 1]);
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
            const int off = voff(i, n) - i - 1;
            if (c0 > i && c0 < n) t.Vs[off + c0] = pend_v0;
            if (c1 > i && c1 < n) t.Vs[off + c1] = pend_v1;
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
        double2 vv[NP2];
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
            for (int k = K0; k < NP2; ++k) vv[k] = *(const double2*)&vb[2 * q + 2 * QN2 * k];
            double ax0 = 0.0, ay0 = 0.0, ax1 = 0.0, ay1 = 0.0;
#pragma unroll
            for (int k = K0; k < NP2; ++k) {
                ax0 = fma(A0[2 * k], vv[k].x, ax0);
                ay0 = fma(A0[2 * k + 1], vv[k].y, ay0);
                ax1 = fma(A1[2 * k], vv[k].x, ax1);
                ay1 = fma(A1[2 * k + 1], vv[k].y, ay1);
            }
            const double acc0 = sum16(ax0 + ay0), acc1 = sum16(ax1 + ay1);
            if (q == 0) {
                p[r0] = (r0 > i) ? tau * acc0 : 0.0;
                p[r1] = (r1 > i) ? tau * acc1 : 0.0;
            }
        } else if (wave * RPW + RPW - 1 == i) {
            // this wave's rows have just retired: clear their p entries in both buffers for good
            if (q == 0) {
                t.ps[r0] = 0.0;
                t.ps[128 + r0] = 0.0;
                t.ps[r1] = 0.0;
                t.ps[128 + r1] = 0.0;
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
            const double2 prr = *(const double2*)&p[r0], vrr = *(const double2*)&vb[r0];      // rows r0, r0 + 1 (r0 is even)
            const double dot = wave_sum_mfma(p[c0] * vb[c0] + p[c1] * vb[c1]);
            const double a2 = -0.5 * tau * dot;
            const double wr0 = prr.x + a2 * vrr.x, wr1 = prr.y + a2 * vrr.y;
            const double g0 = a2 * vrr.x + wr0, g1 = a2 * vrr.y + wr1;
            DBG(3);
            double2 pv[NP2];
#pragma unroll
            for (int k = K0; k < NP2; ++k) pv[k] = *(const double2*)&p[2 * q + 2 * QN2 * k];
#pragma unroll
            for (int k = K0; k < NP2; ++k) {
                A0[2 * k] = fma(-vrr.x, pv[k].x, A0[2 * k]);
                A0[2 * k + 1] = fma(-vrr.x, pv[k].y, A0[2 * k + 1]);
                A0[2 * k] = fma(-g0, vv[k].x, A0[2 * k]);
                A0[2 * k + 1] = fma(-g0, vv[k].y, A0[2 * k + 1]);
                A1[2 * k] = fma(-vrr.y, pv[k].x, A1[2 * k]);
                A1[2 * k + 1] = fma(-vrr.y, pv[k].y, A1[2 * k + 1]);
                A1[2 * k] = fma(-g1, vv[k].x, A1[2 * k]);
                A1[2 * k + 1] = fma(-g1, vv[k].y, A1[2 * k + 1]);
            }
            if (i + 2 < n - 1) publish_row(i + 2);
            flush_reflector();                     // early steps: the helper is still a live wave
            DBG(4);
        }
        __syncthreads();
        DBG(5);
    };
    static_assert(NP2 == 4 && QN2 == 16, "the era loops assume 4 column groups of 32 columns per thread");
    {
        int i = 0;
#define TRI_ERA(K) for (; i < n - 1 && ((i + 1) >> 5) == K; ++i) step(std::integral_constant<int, K>{}, i);
        TRI_ERA(0) TRI_ERA(1) TRI_ERA(2) TRI_ERA(3)
#undef TRI_ERA
    }
    flush_reflector();
    {   // last diagonal element
        double* x = t.xs + ((n - 1) & 1) * 128;
        __syncthreads();
        if (r0 == n - 1) {
#pragma unroll
            for (int k = 0; k < NP2; ++k) *(double2*)&x[2 * q + 2 * QN2 * k] = make_double2(A0[2 * k], A0[2 * k + 1]);
        }
        if (r1 == n - 1) {
#pragma unroll
            for (int k = 0; k < NP2; ++k) *(double2*)&x[2 * q + 2 * QN2 * k] = make_double2(A1[2 * k], A1[2 * k + 1]);
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


static void launch_tri_rows2(const double* G, int n, double* ws, unsigned long long* stamps, hipStream_t s) {
    View v{};
    static bool init = false;
    if (!init) {
        (void)hipFuncSetAttribute((const void*)k_eig_tri2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)eig_lds_bytes());
        init = true;
    }
    hipLaunchKernelGGL(k_eig_tri2, dim3(1), dim3(TRI_T), eig_lds_bytes(), s, v, 0, 0, G, n, 0, ws, stamps);
}
}  // namespace mpst
