// mpst_encode.hip - device-side preprocessing + Legendre encoding.
//
// Replaces, for the real bases the array sweep can train on, the host pipeline
//   transform_train_data / transform_test_data  (src/utils.jl:161-275)
//   legendre / legendre_no_norm                  (src/Encodings/bases.jl:70-108)
//   the per-value encode loop of encode_dataset  (src/Encodings/encodings.jl:120-150)
// so that a fit uploads the N x T raw matrix instead of the d times larger product states.  The order
// statistics of the RobustSigmoid fit (median and quartiles of the training set, utils.jl:171-176) come from a
// device radix sort of the N T values (hipCUB - a plain library sort, not a hot kernel) when asked for.
// Every formula keeps the reference's order of operations; FMA contraction is off in this file so the
// Bonnet recursion rounds like the host restatement (differences come from exp() only, ~1 ulp).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include "mpst_internal.h"

#pragma clang fp contract(off)

namespace mpst {

__device__ __forceinline__ double enc_stage1(const EncDev& e, double x) {
    if (e.sigmoid) x = 1.0 / (1.0 + exp(-(x - e.med) / e.s));
    return x;
}
// value after sigmoid, min-max and data_bounds (what the test path inspects per series)
__device__ __forceinline__ double enc_stage2(const EncDev& e, double x) {
    x = enc_stage1(e, x);
    if (e.minmax) {
        const double lo = e.lohi[0], hi = e.lohi[1];
        x = (x - lo) / (hi - lo);
        x = x * (e.ub - e.lb) + e.lb;
    }
    return x;
}

// min / max of the sigmoid-transformed training data: per-block partials, then one block
__global__ __launch_bounds__(256) void k_enc_range(EncDev e, const double* __restrict__ X, double* __restrict__ part) {
    __shared__ double smin[4], smax[4];
    const int64_t total = e.N * e.T;
    double mn = 1e300, mx = -1e300;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const double y = enc_stage1(e, X[i]);
        mn = fmin(mn, y);
        mx = fmax(mx, y);
    }
    mn = -wave_max(-mn);
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) {
        smin[threadIdx.x >> 6] = mn;
        smax[threadIdx.x >> 6] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = fmin(fmin(smin[0], smin[1]), fmin(smin[2], smin[3]));
        part[2 * blockIdx.x + 1] = fmax(fmax(smax[0], smax[1]), fmax(smax[2], smax[3]));
    }
}
__global__ __launch_bounds__(64) void k_enc_range_final(const double* __restrict__ part, int nblocks, double* __restrict__ lohi) {
    double mn = 1e300, mx = -1e300;
    for (int i = threadIdx.x; i < nblocks; i += 64) {
        mn = fmin(mn, part[2 * i]);
        mx = fmax(mx, part[2 * i + 1]);
    }
    mn = -wave_max(-mn);
    mx = wave_max(mx);
    if (threadIdx.x == 0) {
        lohi[0] = mn;
        lohi[1] = mx;
    }
}

// test data: per-series out-of-bounds rescale (utils.jl:243-266): shift by the minimum if it is
// negative, then divide by the maximum if it exceeds 1.  One wave per series.
__global__ __launch_bounds__(256) void k_enc_series_fix(EncDev e, const double* __restrict__ X, double* __restrict__ fix) {
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= e.N) return;
    double mn = 1e300, mx = -1e300;
    for (int t = lane; t < e.T; t += 64) {
        const double u = enc_stage2(e, X[i * e.T + t]);
        mn = fmin(mn, u);
        mx = fmax(mx, u);
    }
    mn = -wave_max(-mn);
    mx = wave_max(mx);
    const double shift = mn < 0.0 ? mn : 0.0;
    const double hi = mx - shift;             // maximum after the shift (same subtraction the elements get)
    const double scale = hi > 1.0 ? hi : 1.0;
    if (lane == 0) {
        fix[2 * i] = shift;
        fix[2 * i + 1] = scale;
    }
}

// one thread per (site, series); consecutive threads = consecutive series, so the d values every
// thread writes are contiguous across the wave ([T][N][d] site-major, the sweep's layout)
__global__ __launch_bounds__(256) void k_encode(EncDev e, const double* __restrict__ X, double* __restrict__ phi) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= e.N * e.T) return;
    const int64_t t = idx / e.N, i = idx - t * e.N;
    double x = enc_stage2(e, X[i * e.T + t]);
    if (e.fix) {
        const double shift = e.fix[2 * i], scale = e.fix[2 * i + 1];
        if (shift != 0.0) x -= shift;
        if (scale != 1.0) x /= scale;
    }
    x = (e.b - e.a) * x + e.a;
    if (e.basis == MPST_BASIS_STOUDENMIRE) {
        // angle_encode (bases.jl:7-21, periods = 1/4): cispi(3x/2) cospi(x/2), cispi(-3x/2) sinpi(x/2)
        double* outc = phi + idx * 4;
        double s3, c3, sh, ch;
        sincospi(1.5 * x, &s3, &c3);
        sincospi(0.5 * x, &sh, &ch);
        outc[0] = c3 * ch;
        outc[1] = s3 * ch;
        outc[2] = c3 * sh;
        outc[3] = -s3 * sh;
        return;
    }
    if (e.basis == MPST_BASIS_SAHAND) {
        // sahand_encode (bases.jl:45-68): d/2 intervals of width 2/d, two states each, zero outside their interval
        double* outc = phi + idx * e.d * 2;
        const double dxs = 2.0 / e.d;
        for (int k = 0; k < e.d; ++k) {
            const int interval = k / 2 + 1;
            const double startx = (interval - 1) * dxs;
            const bool inside = startx <= x && x <= interval * dxs;
            double s3, c3, sh, ch;
            sincospi(1.5 * x / dxs, &s3, &c3);
            sincospi(0.5 * (x - startx) / dxs, &sh, &ch);
            const bool odd = (k & 1) == 0;
            outc[2 * k] = inside ? (odd ? c3 * ch : c3 * sh) : 0.0;
            outc[2 * k + 1] = inside ? (odd ? s3 * ch : -s3 * sh) : 0.0;
        }
        return;
    }
    if (e.basis == MPST_BASIS_UNIFORM) {
        double* outu = phi + idx * e.d;
        for (int k = 0; k < e.d; ++k) outu[k] = 1.0 / e.d;        // uniform_encode (bases.jl:2-4)
        return;
    }
    if (e.fourier) {
        // fourier_encode (bases.jl:23-42): cispi(f x) / sqrt(d) with f = 0, 1, -1, 2, -2, ...
        double* outc = phi + idx * e.d * 2;
        const double inv = 1.0 / sqrt((double)e.d);
        for (int k = 0; k < e.d; ++k) {
            const int f = (k + 1) / 2 * ((k & 1) ? 1 : -1);
            double sn, cs;
            sincospi((double)f * x, &sn, &cs);
            outc[2 * k] = cs * inv;
            outc[2 * k + 1] = sn * inv;
        }
        return;
    }
    double* out = phi + idx * e.d;
    // Bonnet recursion, then sqrt((2k+1)/2) (normalised Legendre), then the optional 1/nrm (bases.jl:77-92)
    double p0 = 1.0, p1 = x;
    for (int k = 0; k < e.d; ++k) {
        double p;
        if (k == 0) p = 1.0;
        else if (k == 1) p = x;
        else {
            const int m = k - 1;
            p = ((2 * m + 1) * x * p1 - m * p0) / (m + 1);
            p0 = p1;
            p1 = p;
        }
        double v = p * sqrt((2.0 * k + 1.0) / 2.0);
        if (e.norm) v = v / e.nrm;
        out[k] = v;
    }
}

// ---- RobustSigmoid fit: median and inter-quartile range of all training values ---------------------------------
// type-7 quantiles (Julia's / NumPy's default) of the sorted values; the interpolation is written as NumPy's _lerp so that
// the host restatement and this kernel agree to the bit (Julia's a + g (b - a) differs from it by at most one ulp)
__global__ void k_enc_quantiles(const double* __restrict__ xs, int64_t n, double* __restrict__ out /* median, q25, q75 */) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    auto lerp = [](double a, double b, double t) {
        const double dba = b - a;
        return t >= 0.5 ? b - dba * (1.0 - t) : a + dba * t;
    };
    auto quant = [&](double q) {
        const double pos = q * (double)(n - 1);
        const int64_t lo = (int64_t)floor(pos);
        const int64_t hi = lo + 1 < n ? lo + 1 : n - 1;
        return lerp(xs[lo], xs[hi], pos - (double)lo);
    };
    out[0] = (n & 1) ? xs[n / 2] : 0.5 * (xs[n / 2 - 1] + xs[n / 2]);       // np.median / Statistics.median: mean of the middle pair
    out[1] = quant(0.25);
    out[2] = quant(0.75);
}
size_t order_stats_temp_bytes(int64_t n) {
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortKeys(nullptr, bytes, (const double*)nullptr, (double*)nullptr, (int)n);
    return bytes;
}
hipError_t launch_order_stats(const double* X, double* sorted, void* temp, size_t temp_bytes, int64_t n, double* out3, hipStream_t s) {
    hipError_t e = hipcub::DeviceRadixSort::SortKeys(temp, temp_bytes, X, sorted, (int)n, 0, 64, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_enc_quantiles, dim3(1), dim3(1), 0, s, (const double*)sorted, n, out3);
    return hipGetLastError();
}

void launch_encode(const EncDev& e, const double* X, double* phi, double* part, double* lohi, double* fix, int fit_range,
                   hipStream_t s) {
    if (fit_range) {
        const int nb = 256;
        hipLaunchKernelGGL(k_enc_range, dim3(nb), dim3(256), 0, s, e, X, part);
        hipLaunchKernelGGL(k_enc_range_final, dim3(1), dim3(64), 0, s, (const double*)part, nb, lohi);
    }
    if (fix) hipLaunchKernelGGL(k_enc_series_fix, dim3((unsigned)((e.N + 3) / 4)), dim3(256), 0, s, e, X, fix);
    hipLaunchKernelGGL(k_encode, dim3((unsigned)((e.N * e.T + 255) / 256)), dim3(256), 0, s, e, X, phi);
}

}  // namespace mpst
