// Prototype: Householder tridiagonalisation with ONE workgroup barrier per reflector.
// Every 8-lane row group repeats the cheap part of a step for itself at its own 16 columns (the pending rank-2 update's
// scalars, the updated next row, its norm, the reflector), then updates its row and multiplies it with the NEW reflector
// in the same pass; the only exchange of a step is y = A v (one entry per row) plus the raw next row.
#pragma once
namespace mpst {
constexpr int TRI_PROTO_VARIANTS = 1;

__global__ __launch_bounds__(TRI_T) void k_eig_tri1(View v, int lid, int going_left, const double* rawG, int rawn,
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
    const int r = tid / QN, q = tid % QN;
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
    double* xrb = t.Z;                             // [2][128] raw row j (all updates < j-1 applied), parity j & 1
    if (r == 0) {
#pragma unroll
        for (int k = 0; k < NP; ++k) *(double2*)&xrb[2 * q + 2 * QN * k] = make_double2(A[2 * k], A[2 * k + 1]);
    }
    __syncthreads();
    double2 vv[NP];                                // the current reflector at this thread's columns
#pragma unroll
    for (int k = 0; k < NP; ++k) vv[k] = make_double2(0.0, 0.0);

    // Step i (i = -1 .. n-3): apply the rank-2 update of reflector i (none for i = -1), build reflector j = i+1 from the
    // updated row j, y_j = A v_j.  K0 = finished 16-column groups (all their columns <= i).
    auto step = [&](auto K0c, const int i_) {
        constexpr int K0 = decltype(K0c)::value;
        const int i = __builtin_amdgcn_readfirstlane(i_);
        const int j = i + 1;
#ifdef MPST_TRI_STEPPROF
        if (stamps && tid == 0) stamps[64 + j] = __builtin_readcyclecounter();
#endif
        const bool live = (wave * RPW + RPW - 1) > i;          // this wave still owns rows >= j
        if (live) {
            const double* yb = t.ps + (i & 1) * 128;
            const double* vb = t.xs + (i & 1) * 128;
            const double* xr = xrb + (j & 1) * 128;
            double2 yv[NP], xv[NP];
#pragma unroll
            for (int k = K0; k < NP; ++k) {
                yv[k] = *(const double2*)&yb[2 * q + 2 * QN * k];
                xv[k] = *(const double2*)&xr[2 * q + 2 * QN * k];
            }
            const double tau = i >= 0 ? t.taus[i] : 0.0;
            const double y_r = yb[r], v_r = vb[r];
            const double y_j = yb[j], y_j1 = yb[j + 1], v_j1 = vb[j + 1], a_jj = xr[j], a_jj1 = xr[j + 1];
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int k = K0; k < NP; ++k) {
                s0 = fma(yv[k].x, vv[k].x, s0);
                s1 = fma(yv[k].y, vv[k].y, s1);
            }
            const double doty = sum_q(s0 + s1);
            const double a2 = -0.5 * tau * (tau * doty);
            const double p_r = tau * y_r;
            const double w_r = p_r + a2 * v_r;
            const double g_r = a2 * v_r + w_r;
            const double p_j = tau * y_j, p_j1 = tau * y_j1;
            const double g_j = a2 + (p_j + a2);                  // v_i[j] = 1
            const double di = (a_jj - p_j) - g_j;
            const double al = fma(-g_j, v_j1, a_jj1 - p_j1);
            double2 xm[NP];
            double n0 = 0.0, n1 = 0.0;
#pragma unroll
            for (int k = K0; k < NP; ++k) {
                const double pcx = tau * yv[k].x, pcy = tau * yv[k].y;
                A[2 * k] = fma(-v_r, pcx, A[2 * k]);
                A[2 * k + 1] = fma(-v_r, pcy, A[2 * k + 1]);
                A[2 * k] = fma(-g_r, vv[k].x, A[2 * k]);
                A[2 * k + 1] = fma(-g_r, vv[k].y, A[2 * k + 1]);
                double x0 = fma(-g_j, vv[k].x, xv[k].x - pcx);
                double x1 = fma(-g_j, vv[k].y, xv[k].y - pcy);
                if (k < K0 + 2) {                                // columns <= j+1 only occur in the first two live groups
                    const int c = 2 * q + 2 * QN * k;
                    x0 = c >= j + 2 ? x0 : 0.0;
                    x1 = c + 1 >= j + 2 ? x1 : 0.0;
                }
                xm[k] = make_double2(x0, x1);
                n0 = fma(x0, x0, n0);
                n1 = fma(x1, x1, n1);
            }
            const double s = sum_q(n0 + n1);
            // reflector j (as finish_reflector of k_eig_tri)
            const double xx = fma(al, al, s);
            const bool nz = s != 0.0 && xx > 1e-280;
            const double rs = __builtin_amdgcn_rsq(nz ? xx : 1.0);
            double nrm = xx * rs;
            const double hrs = 0.5 * rs;
            nrm = fma(fma(-nrm, nrm, xx), hrs, nrm);
            nrm = fma(fma(-nrm, nrm, xx), hrs, nrm);
            const double bneg = copysign(nrm, al);
            const double ib = frcp(bneg), is = frcp(al + bneg);
            const double beta = nz ? -bneg : al;
            const double taun = nz ? (bneg + al) * ib : 0.0;
            const double scale = nz ? is : 0.0;
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int k = K0; k < NP; ++k) {
                double v0 = xm[k].x * scale, v1 = xm[k].y * scale;
                if (k < K0 + 2) {
                    const int c = 2 * q + 2 * QN * k;
                    v0 = c == j + 1 ? 1.0 : v0;
                    v1 = c + 1 == j + 1 ? 1.0 : v1;
                }
                vv[k] = make_double2(v0, v1);
                a0 = fma(A[2 * k], v0, a0);
                a1 = fma(A[2 * k + 1], v1, a1);
            }
            const double yn = sum_q(a0 + a1);
            double* ybn = t.ps + (j & 1) * 128;
            if (q == 0) ybn[r] = r > j ? yn : 0.0;
            if (r == j + 1) {                                    // the raw row the next step starts from
                double* x = xrb + ((j + 1) & 1) * 128;
#pragma unroll
                for (int k = K0; k < NP; ++k) *(double2*)&x[2 * q + 2 * QN * k] = make_double2(A[2 * k], A[2 * k + 1]);
            }
            if (r == j) {                                        // this row group publishes the reflector and T's entries
                double* vbn = t.xs + (j & 1) * 128;
                const int off = voff(j, n) - j - 1;
#pragma unroll
                for (int k = K0; k < NP; ++k) {
                    const int c = 2 * q + 2 * QN * k;
                    *(double2*)&vbn[c] = vv[k];
                    if (c > j && c < n) t.Vs[off + c] = vv[k].x;
                    if (c + 1 > j && c + 1 < n) t.Vs[off + c + 1] = vv[k].y;
                }
                if (q == 0) {
                    t.taus[j] = taun;
                    t.de[2 * j] = di;
                    t.es[j] = beta;
                }
            }
        } else if (wave * RPW + RPW - 1 == i) {
            if (q == 0) {
                t.ps[r] = 0.0;
                t.ps[128 + r] = 0.0;
            }
        }
        __syncthreads();
    };
    static_assert(NP == 8 && QN == 8, "the era loops assume 8 column groups of 16 columns per thread");
    {
        int i = -1;
#define TRI_ERA(K) for (; i < n - 2 && ((i + 1) >> 4) == K; ++i) step(std::integral_constant<int, K>{}, i);
        TRI_ERA(0) TRI_ERA(1) TRI_ERA(2) TRI_ERA(3) TRI_ERA(4) TRI_ERA(5) TRI_ERA(6) TRI_ERA(7)
#undef TRI_ERA
    }
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
    if (tid < n) t.de[2 * tid + 1] = tid > 0 ? t.es[tid - 1] * t.es[tid - 1] : 0.0;
    if (wave == 0) {
        double gl = 1e300, gu = -1e300;
        for (int jj = lane; jj < n; jj += 64) {
            const double a = jj > 0 ? fabs(t.es[jj - 1]) : 0.0, b = jj < n - 1 ? fabs(t.es[jj]) : 0.0;
            const double dj = t.de[2 * jj];
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
    if (tid < 256) ws[WS_DE + tid] = tid < 2 * n ? t.de[tid] : 0.0;
    if (tid < 128) {
        ws[WS_ES + tid] = tid < n ? t.es[tid] : 0.0;
        ws[WS_TAU + tid] = tid < n - 1 ? t.taus[tid] : 0.0;
    }
    for (int ii = tid; ii < 8 * 128 * 16; ii += TRI_T) {
        const int jj = ii & 15, c = (ii >> 4) & 127, jr = (ii >> 11) * 16 + jj;
        ws[WS_VS + ii] = (jr < n - 1 && c > jr && c < n) ? t.Vs[voff(jr, n) + c - jr - 1] : 0.0;
    }
    if (tid == 0) {
        ws[WS_MISC + 0] = t.misc[0];
        ws[WS_MISC + 1] = t.misc[1];
        ws[WS_MISC + 2] = t.misc[2];
        ws[WS_MISC + 3] = 1.0;
    }
}

static void launch_tri_proto(int which, const double* G, int n, double* ws, unsigned long long* stamps, hipStream_t s) {
    View v{};
    static bool init = false;
    if (!init) {
        (void)hipFuncSetAttribute((const void*)k_eig_tri1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)eig_lds_bytes());
        init = true;
    }
    hipLaunchKernelGGL(k_eig_tri1, dim3(1), dim3(TRI_T), eig_lds_bytes(), s, v, 0, 0, G, n, 0, ws, stamps);
}
}  // namespace mpst
