// Floor of a chain of dependent launches on this box: empty kernel, a kernel that does one global load + store, and a
// kernel with 3 block barriers - plain stream vs hipGraph replay.  hipcc --offload-arch=gfx950 -O3 launchfloor.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_empty(double* p) {}
__global__ void k_touch(double* p) { if (threadIdx.x == 0) p[blockIdx.x] += 1.0; }
__global__ __launch_bounds__(256) void k_sync(double* p) {
    __shared__ double s[256];
    s[threadIdx.x] = p[blockIdx.x * 256 + threadIdx.x];
    __syncthreads();
    double a = s[(threadIdx.x + 1) & 255];
    __syncthreads();
    s[threadIdx.x] = a;
    __syncthreads();
    p[blockIdx.x * 256 + threadIdx.x] = s[(threadIdx.x + 7) & 255];
}
template <typename F> double run(F launch, hipStream_t s, int n, bool graph) {
    hipGraphExec_t ex = nullptr;
    if (graph) {
        hipGraph_t g;
        hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        for (int i = 0; i < n; ++i) launch();
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
        hipGraphDestroy(g);
        hipGraphLaunch(ex, s);
        hipStreamSynchronize(s);
    } else {
        for (int i = 0; i < n; ++i) launch();
        hipStreamSynchronize(s);
    }
    auto t0 = std::chrono::steady_clock::now();
    if (graph) hipGraphLaunch(ex, s);
    else for (int i = 0; i < n; ++i) launch();
    hipStreamSynchronize(s);
    auto t1 = std::chrono::steady_clock::now();
    if (ex) hipGraphExecDestroy(ex);
    return std::chrono::duration<double, std::micro>(t1 - t0).count() / n;
}
int main() {
    hipStream_t s;
    hipStreamCreate(&s);
    double* p;
    hipMalloc(&p, 256 * 256 * sizeof(double));
    hipMemset(p, 0, 256 * 256 * sizeof(double));
    const int n = 2000;
    for (int g : {1, 64, 128}) {
        for (int gr = 0; gr < 2; ++gr) {
            double a = run([&] { hipLaunchKernelGGL(k_empty, dim3(g), dim3(256), 0, s, p); }, s, n, gr);
            double b = run([&] { hipLaunchKernelGGL(k_touch, dim3(g), dim3(256), 0, s, p); }, s, n, gr);
            double c = run([&] { hipLaunchKernelGGL(k_sync, dim3(g), dim3(256), 0, s, p); }, s, n, gr);
            printf("grid %3d %s: empty %.2f us, touch %.2f us, load+3 barriers+store %.2f us per dependent launch\n", g, gr ? "graph " : "stream", a, b, c);
        }
    }
    return 0;
}
