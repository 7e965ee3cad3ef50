// Same-box A/B harness for the n <= 128 tridiagonalisation: runs the engine's k_eig_tri (included from the product
// source) and the prototypes of tri_proto.hip on the same Gram matrix, times them with HIP events, dumps per-step cycle
// stamps and writes (d, e) of each so that a Python check can compare the spectra.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMPST_TRI_STEPPROF lab/ubench/tri_ab.hip -o lab/ubench/tri_ab.bin -lrocsolver -lrocblas -ldl
#include "../../mpstime.jl_amd/csrc/mpst_eig.hip"
#ifdef HAVE_PROTO
#include "tri_proto_gen.hip"
#include "tri_rows2.hip"
#include "tri_par.hip"
#include "tri_eg.hip"
#endif
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>
using namespace mpst;
#define CK(x) do { auto e_ = (x); if (e_ != 0) { printf("fail %s -> %d line %d\n", #x, (int)e_, __LINE__); exit(1); } } while (0)

static std::vector<double> make_gram(int m, int n, int rank, double eta, unsigned seed) {
    // A = L R (rank `rank`, graded) + eta * noise: the shape of an updated bond tensor
    std::vector<double> A((size_t)m * n, 0.0), L((size_t)m * rank), R((size_t)rank * n), G((size_t)n * n);
    srand(seed);
    auto rnd = []() { return rand() / (double)RAND_MAX - 0.5; };
    for (auto& x : L) x = rnd();
    for (auto& x : R) x = rnd();
    for (int k = 0; k < rank; ++k) {
        const double g = std::pow(0.7, k);
        for (int i = 0; i < m; ++i) L[(size_t)i * rank + k] *= g;
    }
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < n; ++j) {
            double t = 0;
            for (int k = 0; k < rank; ++k) t += L[(size_t)i * rank + k] * R[(size_t)k * n + j];
            A[(size_t)i * n + j] = t + eta * rnd();
        }
    double nrm = 0;
    for (auto x : A) nrm += x * x;
    nrm = 1.0 / std::sqrt(nrm);
    for (auto& x : A) x *= nrm;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) {
            double t = 0;
            for (int k = 0; k < m; ++k) t += A[(size_t)k * n + i] * A[(size_t)k * n + j];
            G[(size_t)i * n + j] = G[(size_t)j * n + i] = t;
        }
    return G;
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 128;
    const int reps = argc > 2 ? atoi(argv[2]) : 200;
    CK(eig_init_attrs(0));
    std::vector<double> G = make_gram(2 * n, n, n / 4, 0.01, 7);
    double *dG, *ws, *ws2;
    unsigned long long* st;
    CK(hipMalloc(&dG, sizeof(double) * n * n));
    CK(hipMalloc(&ws, sizeof(double) * WS_TOTAL));
    CK(hipMalloc(&ws2, sizeof(double) * WS_TOTAL));
    CK(hipMalloc(&st, 8 * 8192));
    CK(hipMemset(st, 0, 8 * 8192));
    CK(hipMemcpy(dG, G.data(), sizeof(double) * n * n, hipMemcpyHostToDevice));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    View v{};
    auto run = [&](int which, unsigned long long* stp, double* w) {
        if (which == 0)
            hipLaunchKernelGGL(k_eig_tri, dim3(1), dim3(TRI_T), eig_lds_bytes(), s, v, 0, 0, (const double*)dG, n, 0, w, stp);
#ifdef HAVE_PROTO
        else if (which <= TRI_PROTO_VARIANTS)
            launch_tri_proto(which, dG, n, w, stp, s);
        else if (which == TRI_PROTO_VARIANTS + 1)
            launch_tri_rows2(dG, n, w, stp, s);
        else if (which == TRI_PROTO_VARIANTS + 2)
            launch_tri_par(dG, n, w, stp, s);
        else
            launch_tri_eg(dG, n, w, stp, s);
#endif
    };
    const int nvar =
#ifdef HAVE_PROTO
        4 + TRI_PROTO_VARIANTS;
#else
        1;
#endif
    FILE* f = fopen("gpurun_out/tri_ab_T.txt", "w");
    for (int which = 0; which < nvar; ++which) {
        double* w = which == 0 ? ws : ws2;
        CK(hipMemset(w, 0, sizeof(double) * WS_TOTAL));
        for (int i = 0; i < 5; ++i) run(which, nullptr, w);
        CK(hipStreamSynchronize(s));
        // back-to-back launches (dependent on the stream): average
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < reps; ++i) run(which, nullptr, w);
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        CK(hipGetLastError());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        run(which, st, w);
        CK(hipStreamSynchronize(s));
        std::vector<unsigned long long> hs(8192);
        CK(hipMemcpy(hs.data(), st, 8 * 8192, hipMemcpyDeviceToHost));
        std::vector<double> hw(WS_TOTAL);
        CK(hipMemcpy(hw.data(), w, sizeof(double) * WS_TOTAL, hipMemcpyDeviceToHost));
        double tr = 0, trG = 0;
        for (int i = 0; i < n; ++i) { tr += hw[WS_DE + 2 * i]; trG += G[(size_t)i * n + i]; }
        printf("variant %d [%s]: %.2f us per launch (%d back-to-back), in-kernel %.2f us = %llu cycles, trace err %.2e\n", which,
#ifdef HAVE_PROTO
               which <= TRI_PROTO_VARIANTS ? tri_proto_name(which) : which == TRI_PROTO_VARIANTS + 1 ? "two rows per thread (two-barrier steps, 32-column eras)" : which == TRI_PROTO_VARIANTS + 2 ? "product step with compile-time buffer parity" : "product + single-wave endgame (last 32 steps)",
#else
               "product k_eig_tri",
#endif
               1e3 * ms / reps, reps, 0.01 * (double)(hs[1] - hs[0]), hs[7] - hs[6], std::fabs(tr - trG));
        printf("  step cycles:");
        for (int i = 0; i + 1 < n - 1; ++i)
            if (hs[64 + i] && hs[65 + i]) printf(" %llu", hs[65 + i] - hs[64 + i]);
        printf("\n");
#ifdef TRI_EG_DEBUG
        if (which == nvar - 1) {
            FILE* fd = fopen("gpurun_out/tri_eg_dbg.txt", "w");
            const double* dd = (const double*)(hs.data() + 512);
            for (int i = 0; i < 4 * 1024 + 33; ++i) fprintf(fd, "%.17g\n", dd[i]);
            fclose(fd);
        }
#endif
        if (f) {
            fprintf(f, "variant %d\n", which);
            for (int i = 0; i < n; ++i) fprintf(f, "%.17g %.17g\n", hw[WS_DE + 2 * i], hw[WS_ES + i]);
        }
    }
    if (f) {
        fprintf(f, "G %d\n", n);
        for (size_t i = 0; i < (size_t)n * n; ++i) fprintf(f, "%.17g\n", G[i]);
        fclose(f);
    }
    return 0;
}
