// Measures the raw accuracy of v_rcp_f64 / v_rsq_f64 and of one and two Newton refinements,
// on the GPU, against the host's correctly rounded results.  Diagnostic only.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void k(const double* x, double* o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double d = x[i];
    double r0 = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r0, 1.0);
    double r1 = __builtin_fma(e, r0, r0);
    e = __builtin_fma(-d, r1, 1.0);
    double r2 = __builtin_fma(e, r1, r1);
    double e0 = __builtin_fma(-d, r0, 1.0);
    double r3 = __builtin_fma(__builtin_fma(e0, e0, e0), r0, r0);  // the form the kernels use
    double y = __builtin_amdgcn_rsq(d);
    double g = d * y, h = 0.5 * y;
    double r = __builtin_fma(-h, g, 0.5);
    double g1 = __builtin_fma(g, r, g), h1 = __builtin_fma(h, r, h);
    double dd = __builtin_fma(-g1, g1, d);
    double g2 = __builtin_fma(dd, h1, g1);
    o[6 * i + 0] = r0; o[6 * i + 1] = r3; o[6 * i + 2] = r2;
    o[6 * i + 3] = g;  o[6 * i + 4] = g1; o[6 * i + 5] = g2;
}
int main() {
    const int n = 1 << 20;
    double *hx = (double*)malloc(n * 8), *ho = (double*)malloc(n * 48);
    srand(1);
    for (int i = 0; i < n; ++i) hx[i] = exp(((double)rand() / RAND_MAX) * 40.0 - 20.0) * (1.0 + (double)rand() / RAND_MAX);
    double *dx, *dox;
    hipMalloc(&dx, n * 8); hipMalloc(&dox, n * 48);
    hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, dox, n);
    hipMemcpy(ho, dox, n * 48, hipMemcpyDeviceToHost);
    double m[6] = {0};
    for (int i = 0; i < n; ++i) {
        double rr = 1.0 / hx[i], sq = sqrt(hx[i]);
        for (int j = 0; j < 3; ++j) { double e = fabs(ho[6 * i + j] - rr) / rr; if (e > m[j]) m[j] = e; }
        for (int j = 3; j < 6; ++j) { double e = fabs(ho[6 * i + j] - sq) / sq; if (e > m[j]) m[j] = e; }
    }
    printf("max rel err  rcp raw %.3e  r0(1+e+e^2) %.3e  +2NR %.3e | sqrt via rsq raw %.3e  +iter %.3e  +corr %.3e\n",
           m[0], m[1], m[2], m[3], m[4], m[5]);
    return 0;
}
