// Accuracy of v_rsq_f64 and of its Newton refinements (the pivot chain of potrf16): max relative error against a long-double reference.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double* x, double* y0, double* y1, double* y2, double* y1g, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double d = x[i];
    double y = __builtin_amdgcn_rsq(d);
    y0[i] = y;
    const double h = 0.5 * d;
    double a = y * (1.5 - h * y * y);
    y1[i] = a;
    a = a * (1.5 - h * a * a);
    y2[i] = a;
    // one coupled (Goldschmidt) step: e = 1 - d y^2 ; y' = y + (y/2) e
    const double t = d * y, e = fma(-t, y, 1.0);
    y1g[i] = fma(0.5 * y, e, y);
}
int main() {
    const int n = 1 << 22;
    std::vector<double> x(n), a(n), b(n), c(n), g(n);
    unsigned long long s = 88172645463325252ULL;
    for (int i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double u = (s >> 11) * (1.0 / 9007199254740992.0);
        x[i] = std::ldexp(1.0 + 3.0 * u, (int)(s % 41) - 20);
    }
    double *dx, *d0, *d1, *d2, *d3;
    hipMalloc(&dx, 8 * n); hipMalloc(&d0, 8 * n); hipMalloc(&d1, 8 * n); hipMalloc(&d2, 8 * n); hipMalloc(&d3, 8 * n);
    hipMemcpy(dx, x.data(), 8 * n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, d3, n);
    hipMemcpy(a.data(), d0, 8 * n, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d1, 8 * n, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), d2, 8 * n, hipMemcpyDeviceToHost); hipMemcpy(g.data(), d3, 8 * n, hipMemcpyDeviceToHost);
    long double e0 = 0, e1 = 0, e2 = 0, e3 = 0;
    for (int i = 0; i < n; ++i) {
        const long double r = 1.0L / sqrtl((long double)x[i]);
        e0 = std::max(e0, fabsl(a[i] - r) / r); e1 = std::max(e1, fabsl(b[i] - r) / r);
        e2 = std::max(e2, fabsl(c[i] - r) / r); e3 = std::max(e3, fabsl(g[i] - r) / r);
    }
    printf("max rel err: v_rsq_f64 %.3Le (2^%.1Lf) | one Newton step %.3Le | two %.3Le | one coupled step %.3Le   (2^-53 = 1.11e-16)\n", e0, log2l(e0), e1, e2, e3);
    return 0;
}
