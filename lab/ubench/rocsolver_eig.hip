// Times rocSOLVER's symmetric eigensolvers on Gram matrices of the sizes the large-bond path meets (n = d*chi_max > 128).
// hipcc --offload-arch=gfx950 -O2 rocsolver_eig.hip -o rocsolver_eig -lrocsolver -lrocblas
#include <hip/hip_runtime.h>
#include <rocsolver/rocsolver.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#define CK(x) do { auto e_ = (x); if (e_ != 0) { printf("fail %s -> %d line %d\n", #x, (int)e_, __LINE__); exit(1); } } while (0)
int main() {
    rocblas_handle h;
    CK(rocblas_create_handle(&h));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    CK(rocblas_set_stream(h, s));
    for (int n : {160, 256, 296, 384, 512, 564}) {
        const int m = 2 * n, K = 64;
        std::vector<double> A((size_t)m * n), G((size_t)n * n);
        srand(1);
        for (auto& x : A) x = rand() / (double)RAND_MAX - 0.5;
        // graded spectrum: scale columns
        for (int j = 0; j < n; ++j) for (int i = 0; i < m; ++i) A[(size_t)i * n + j] *= std::pow(0.93, j);
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double t = 0; for (int k = 0; k < m; ++k) t += A[(size_t)k * n + i] * A[(size_t)k * n + j]; G[(size_t)i * n + j] = t; }
        double *dG, *dA, *dD, *dE, *dZ; rocblas_int *dinfo, *dnev;
        CK(hipMalloc(&dG, sizeof(double) * n * n)); CK(hipMalloc(&dA, sizeof(double) * n * n)); CK(hipMalloc(&dZ, sizeof(double) * n * n));
        CK(hipMalloc(&dD, sizeof(double) * n)); CK(hipMalloc(&dE, sizeof(double) * n)); CK(hipMalloc(&dinfo, 4)); CK(hipMalloc(&dnev, 4));
        CK(hipMemcpy(dG, G.data(), sizeof(double) * n * n, hipMemcpyHostToDevice));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int alg = 0; alg < 4; ++alg) {
            float best = 1e9;
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipMemcpyAsync(dA, dG, sizeof(double) * n * n, hipMemcpyDeviceToDevice, s));
                CK(hipEventRecord(e0, s));
                rocblas_status st = rocblas_status_success;
                if (alg == 0) st = rocsolver_dsyevd(h, rocblas_evect_original, rocblas_fill_lower, n, dA, n, dD, dE, dinfo);
                if (alg == 1) st = rocsolver_dsyev(h, rocblas_evect_original, rocblas_fill_lower, n, dA, n, dD, dE, dinfo);
                if (alg == 2) { double res; rocblas_int sw; double* dres; rocblas_int* dsw; CK(hipMalloc(&dres, 8)); CK(hipMalloc(&dsw, 4));
                    st = rocsolver_dsyevj(h, rocblas_esort_ascending, rocblas_evect_original, rocblas_fill_lower, n, dA, n, 1e-14, dres, 50, dsw, dD, dinfo); (void)res; (void)sw; }
                if (alg == 3) st = rocsolver_dsyevdx(h, rocblas_evect_original, rocblas_erange_index, rocblas_fill_lower, n, dA, n, 0.0, 0.0, n - K + 1, n, dnev, dD, dZ, n, dinfo);
                CK(hipEventRecord(e1, s));
                CK(hipStreamSynchronize(s));
                if (st != rocblas_status_success) { printf("n=%d alg=%d status %d\n", n, alg, (int)st); break; }
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            std::vector<double> D(n);
            CK(hipMemcpy(D.data(), dD, sizeof(double) * n, hipMemcpyDeviceToHost));
            int info; CK(hipMemcpy(&info, dinfo, 4, hipMemcpyDeviceToHost));
            printf("n=%4d %-8s best %8.3f ms  info=%d  lam[last]=%.6e lam[0]=%.3e\n", n, alg == 0 ? "syevd" : alg == 1 ? "syev" : alg == 2 ? "syevj" : "syevdx", best, info,
                   alg == 3 ? D[K - 1] : D[n - 1], D[0]);
        }
        hipFree(dG); hipFree(dA); hipFree(dD); hipFree(dE); hipFree(dZ); hipFree(dinfo); hipFree(dnev);
    }
    return 0;
}
