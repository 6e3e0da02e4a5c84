// micro-benchmark: rocSOLVER dense symmetric eigensolvers (dsyevd / dsyevdj / dsyevj / dsyevdx(value range)),
// fp64, random symmetric matrices.  build: hipcc -O2 --offload-arch=gfx950 eigbench.cpp -lrocsolver -lrocblas
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    rocblas_handle h; rocblas_create_handle(&h);
    for (int ai = 1; ai < argc; ++ai) {
        const int n = atoi(argv[ai]);
        std::vector<double> A((size_t)n * n);
        std::mt19937_64 g(1); std::normal_distribution<double> nd;
        for (int j = 0; j < n; ++j) for (int i = 0; i <= j; ++i) { double v = nd(g); A[(size_t)j * n + i] = v; A[(size_t)i * n + j] = v; }
        double *dA, *dA0, *dD, *dE, *dZ; rocblas_int *info, *nev, *nsweeps; double* resid;
        hipMalloc(&dA, sizeof(double) * n * n); hipMalloc(&dA0, sizeof(double) * n * n); hipMalloc(&dZ, sizeof(double) * n * n);
        hipMalloc(&dD, sizeof(double) * n); hipMalloc(&dE, sizeof(double) * n); hipMalloc(&info, 4); hipMalloc(&nev, 4);
        hipMalloc(&nsweeps, 4); hipMalloc(&resid, 8);
        hipMemcpy(dA0, A.data(), sizeof(double) * n * n, hipMemcpyHostToDevice);
        for (int which = 0; which < 4; ++which) {
            if (which == 2 && n > 1200) continue;            // plain Jacobi: too slow to bother
            double best = 1e30;
            for (int rep = 0; rep < 3; ++rep) {
                hipMemcpy(dA, dA0, sizeof(double) * n * n, hipMemcpyDeviceToDevice);
                hipDeviceSynchronize();
                double t0 = now();
                rocblas_status s = rocblas_status_success;
                if (which == 0) s = rocsolver_dsyevd(h, rocblas_evect_original, rocblas_fill_upper, n, dA, n, dD, dE, info);
                if (which == 1) s = rocsolver_dsyevdj(h, rocblas_evect_original, rocblas_fill_upper, n, dA, n, dD, info);
                if (which == 2) s = rocsolver_dsyevj(h, rocblas_esort_ascending, rocblas_evect_original, rocblas_fill_upper, n, dA, n, 1e-14, resid, 100, nsweeps, dD, info);
                if (which == 3) s = rocsolver_dsyevdx(h, rocblas_evect_original, rocblas_erange_value, rocblas_fill_upper, n, dA, n, 0.0, 1e300, 0, 0, nev, dD, dZ, n, info);
                hipDeviceSynchronize();
                double t = now() - t0;
                if (s != rocblas_status_success) { printf("n=%d which=%d status %d\n", n, which, (int)s); break; }
                if (rep > 0 && t < best) best = t;
            }
            const char* nm[] = {"dsyevd", "dsyevdj", "dsyevj", "dsyevdx(0,inf]"};
            printf("n=%5d %-16s %9.3f ms  (%.2f TFLOP/s at (10/3) n^3)\n", n, nm[which], best * 1e3, (10.0 / 3.0) * n * (double)n * n / best / 1e12);
            fflush(stdout);
        }
        hipFree(dA); hipFree(dA0); hipFree(dZ); hipFree(dD); hipFree(dE); hipFree(info); hipFree(nev); hipFree(nsweeps); hipFree(resid);
    }
    return 0;
}
