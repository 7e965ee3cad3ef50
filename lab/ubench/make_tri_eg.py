"""Generates lab/ubench/tri_eg.hip: the product k_eig_tri (mpst_eig.hip) + a single-wave endgame for the last 32 x 32
trailing block.  Run from the repository root."""
src = open('mpstime.jl_amd/csrc/mpst_eig.hip').read()
a = src.index('__global__ __launch_bounds__(TRI_T) void k_eig_tri(')
b = src.index('// =====================================================================================\n// k_eig_vec')
k = src[a:b].replace('void k_eig_tri(', 'void k_eig_tri_eg(')
old_loop = k[k.index('    static_assert(NP == 8 && QN == 8'):k.index('    if (stamps && tid == 0) {\n        stamps[1] = __builtin_amdgcn_s_memrealtime();')]
new_loop = r'''    static_assert(NP == 8 && QN == 8, "the era loops assume 8 column groups of 16 columns per thread");
    int i = 0;
    const int i_sw = n >= 64 ? n - 33 : (1 << 30);             // step at which the single-wave endgame takes over
#define TRI_ERA(K) for (; i < n - 1 && i < i_sw && ((i + 1) >> 4) == K; ++i) step(std::integral_constant<int, K>{}, i);
    TRI_ERA(0) TRI_ERA(1) TRI_ERA(2) TRI_ERA(3) TRI_ERA(4) TRI_ERA(5) TRI_ERA(6) TRI_ERA(7)
#undef TRI_ERA
    flush_reflector();
    if (i == i_sw) {
        // ---- single-wave endgame -------------------------------------------------------------------------------------------
        // Top of step i: v_i is published (t.xs), tau_i in taus, the registers hold A^(i).  The trailing block (rows and
        // columns >= R0 = i + 1, 32 x 32) goes through LDS to wave 0: lane (rl = lane >> 1, ql = lane & 1) keeps the columns
        // cl = 2 k + ql, k = 0..15, of row rl.  Vectors live in LDS de-interleaved ([parity][16]) so that a lane reads its
        // 16 entries as 8 x 16 bytes.  Nothing below synchronises with another wave.
        const int R0 = i + 1;
        double* Ablk = t.Ub;                     // [32][2][16]
        double* vloc = t.Ub + 1024;              // [2][16] current reflector (local columns)
        double* pbuf = vloc + 32;                // [2][16]
        double* xrow = pbuf + 32;                // [2][16]
        if (r >= R0 && r < n) {
#pragma unroll
            for (int k = 0; k < NP; ++k)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int c = 2 * q + 2 * QN * k + h;
                    if (c >= R0 && c < n) {
                        const int cl = c - R0;
                        Ablk[((r - R0) * 2 + (cl & 1)) * 16 + (cl >> 1)] = A[2 * k + h];
                    }
                }
        }
        __syncthreads();
        if (wave == 0) {
            const int rl = lane >> 1, ql = lane & 1;
            double Ae[16], ve[16], tv[16];
#pragma unroll
            for (int k = 0; k < 16; k += 2) {
                const double2 a2v = *(const double2*)&Ablk[(rl * 2 + ql) * 16 + k];
                Ae[k] = a2v.x;
                Ae[k + 1] = a2v.y;
            }
            {
                const double* vb = t.xs + (i & 1) * 128;
#pragma unroll
                for (int k = 0; k < 16; ++k) ve[k] = vb[R0 + 2 * k + ql];
                if (rl == 0) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) vloc[ql * 16 + k] = ve[k];
                }
            }
#ifdef TRI_EG_DEBUG
            double* dbgd = (double*)(stamps + 512);
            auto dump = [&](int slot) {
                if (stamps) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) dbgd[slot * 1024 + rl * 32 + 2 * k + ql] = Ae[k];
                }
            };
            dump(0);
            if (stamps && rl == 0) {
#pragma unroll
                for (int k = 0; k < 16; ++k) dbgd[4 * 1024 + 2 * k + ql] = ve[k];
                if (lane == 0) dbgd[4 * 1024 + 32] = t.taus[i];
            }
#endif
            // a lane reads what ANOTHER lane of the same wave has just written to LDS: the LDS executes a wave's instructions in
            // order, but the compiler must not move the (per-thread independent) load above the masked store
#define EG_HANDOVER() asm volatile("" ::: "memory")
            EG_HANDOVER();
            auto eg_step = [&](auto K0c, const int i_) {
                constexpr int K0 = decltype(K0c)::value;           // finished column pairs: 0 or 8
                const int ii = __builtin_amdgcn_readfirstlane(i_);
                const int li = ii - R0, lj = li + 1, j = ii + 1;
#ifdef MPST_TRI_STEPPROF
                if (stamps && lane == 0) stamps[64 + ii] = __builtin_readcyclecounter();
#endif
                const double tau = t.taus[ii];
                // (a) p = tau A v: this lane's half of the row, the partner's through one DPP move
                double a0 = 0.0, a1 = 0.0;
#pragma unroll
                for (int k = K0; k < 16; k += 2) {
                    a0 = fma(Ae[k], ve[k], a0);
                    a1 = fma(Ae[k + 1], ve[k + 1], a1);
                }
                double acc = a0 + a1;
                acc += dpp_mov<DPP_XOR1>(acc);
                const double p_r = rl > li ? tau * acc : 0.0;
                if (ql == 0) pbuf[(rl & 1) * 16 + (rl >> 1)] = p_r;
                EG_HANDOVER();
                // (b) v^T p (every lane pair forms it for itself: same bits), rank-2 update
#pragma unroll
                for (int k = K0; k < 16; k += 2) {
                    const double2 t2 = *(const double2*)&pbuf[ql * 16 + k];
                    tv[k] = t2.x;
                    tv[k + 1] = t2.y;
                }
                const double v_r = vloc[(rl & 1) * 16 + (rl >> 1)];
                double d0 = 0.0, d1 = 0.0;
#pragma unroll
                for (int k = K0; k < 16; k += 2) {
                    d0 = fma(tv[k], ve[k], d0);
                    d1 = fma(tv[k + 1], ve[k + 1], d1);
                }
                double dot = d0 + d1;
                dot += dpp_mov<DPP_XOR1>(dot);
                const double a2 = -0.5 * tau * dot;
                const double w_r = p_r + a2 * v_r;
                const double g_r = a2 * v_r + w_r;
#pragma unroll
                for (int k = K0; k < 16; ++k) {
                    Ae[k] = fma(-v_r, tv[k], Ae[k]);
                    Ae[k] = fma(-g_r, ve[k], Ae[k]);
                }
                // (c) reflector j = i + 1 from the updated row j: its owner pair publishes it, everybody builds v_j
                if (rl == lj) {
#pragma unroll
                    for (int k = K0; k < 16; k += 2) *(double2*)&xrow[ql * 16 + k] = make_double2(Ae[k], Ae[k + 1]);
                }
                EG_HANDOVER();
#pragma unroll
                for (int k = K0; k < 16; k += 2) {
                    const double2 t2 = *(const double2*)&xrow[ql * 16 + k];
                    tv[k] = t2.x;
                    tv[k + 1] = t2.y;
                }
                const double di = xrow[(lj & 1) * 16 + (lj >> 1)];
                const double al = xrow[((lj + 1) & 1) * 16 + ((lj + 1) >> 1)];
                double n0 = 0.0, n1 = 0.0;
#pragma unroll
                for (int k = K0; k < 16; ++k) {
                    const int cl = 2 * k + ql;
                    tv[k] = cl >= lj + 2 ? tv[k] : 0.0;
                    if (k & 1) n1 = fma(tv[k], tv[k], n1);
                    else n0 = fma(tv[k], tv[k], n0);
                }
                double ssum = n0 + n1;
                ssum += dpp_mov<DPP_XOR1>(ssum);
                const double xx = fma(al, al, ssum);
                const bool nz = ssum != 0.0 && xx > 1e-280;
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
#pragma unroll
                for (int k = K0; k < 16; ++k) {
                    const int cl = 2 * k + ql;
                    ve[k] = cl == lj + 1 ? 1.0 : tv[k] * scale;
                }
                if (rl == 0) {
                    const int off = voff(j, n) - j - 1;
#pragma unroll
                    for (int k = K0; k < 16; ++k) {
                        const int c = R0 + 2 * k + ql;
                        vloc[ql * 16 + k] = ve[k];
                        if (c > j && c < n) t.Vs[off + c] = ve[k];
                    }
                    if (ql == 0) {
                        t.taus[j] = taun;
                        t.de[2 * j] = di;
                        t.es[j] = beta;
                    }
                }
                EG_HANDOVER();
            };
            int ie = i;
            for (; ie < n - 2 && ie - R0 < 15; ++ie) {
                eg_step(std::integral_constant<int, 0>{}, ie);
#ifdef TRI_EG_DEBUG
                if (ie - i < 2) dump(1 + ie - i);
#endif
            }
            for (; ie < n - 2; ++ie) eg_step(std::integral_constant<int, 8>{}, ie);
            if (lane == 63) {                                  // row 31, column 31: the last diagonal element
                t.de[2 * (n - 1)] = Ae[15];
                t.es[n - 1] = 0.0;
            }
        }
        __syncthreads();
    } else {   // last diagonal element
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
'''
k = k.replace(old_loop, new_loop)
hdr = '''// Prototype (generated by lab/ubench/make_tri_eg.py from mpst_eig.hip): the product tridiagonalisation with a
// single-wave endgame - the last 32 x 32 trailing block is handed to ONE wave (two lanes per row, 16 interleaved columns
// per lane) that finishes the factorisation without a workgroup barrier.
#pragma once
namespace mpst {
'''
open('lab/ubench/tri_eg.hip', 'w').write(hdr + k + '''
static void launch_tri_eg(const double* G, int n, double* ws, unsigned long long* stamps, hipStream_t s) {
    View v{};
    static bool init = false;
    if (!init) {
        (void)hipFuncSetAttribute((const void*)k_eig_tri_eg, hipFuncAttributeMaxDynamicSharedMemorySize, (int)eig_lds_bytes());
        init = true;
    }
    hipLaunchKernelGGL(k_eig_tri_eg, dim3(1), dim3(TRI_T), eig_lds_bytes(), s, v, 0, 0, G, n, 0, ws, stamps);
}
}  // namespace mpst
''')
