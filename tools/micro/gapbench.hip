// What a dependent launch costs on MI355X as a function of its shape: chains of L dependent launches in ONE captured hipGraph
// (so that the host's enqueue rate is out of the picture), per-launch time by events.  Variables: workgroups per launch,
// static LDS per workgroup, kernel-argument bytes, one kernel vs two alternating kernels, and the in-kernel work of a
// "load 64 KB per workgroup from memory another launch wrote" step (the shape of the Lanczos step kernels).
// hipcc --offload-arch=gfx950 -O2 -o tools/micro/gapbench tools/micro/gapbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Pad { double p[32]; };                      // 256 bytes of extra kernel arguments

template <int LDS_KB>
__global__ void __launch_bounds__(256) k_small(double* __restrict__ v, const double* __restrict__ u, int n, double a) {
    __shared__ double s[LDS_KB > 0 ? LDS_KB * 128 : 1];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (LDS_KB > 0) s[threadIdx.x] = a;
    if (i < n) v[i] = a * u[i] + v[i] + (LDS_KB > 0 ? s[threadIdx.x] * 0.0 : 0.0);
}
__global__ void __launch_bounds__(256) k_small_b(double* __restrict__ v, const double* __restrict__ u, int n, double a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) v[i] = a * u[i] - v[i] * 1e-30;
}
__global__ void __launch_bounds__(256) k_args(double* __restrict__ v, const double* __restrict__ u, int n, double a, Pad p) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) v[i] = a * u[i] + v[i] + p.p[31] * 0.0;
}
// every workgroup streams `cols` columns of 64 rows (512 bytes per wave-load) written by the previous launch, then writes 64 values
__global__ void __launch_bounds__(256) k_tile(double* __restrict__ out, const double* __restrict__ in, int ld, int cols) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    double acc = 0.0;
    for (int c = wv; c < cols; c += 4) acc += in[(long long)c * ld + i];
    __shared__ double s[4][64];
    s[wv][lane] = acc;
    __syncthreads();
    if (wv == 0) out[(long long)(blockIdx.x & 1) * ld + i] = (s[0][lane] + s[1][lane]) + (s[2][lane] + s[3][lane]);
}

template <typename F>
static int timeit(const char* name, hipStream_t s, F chain, int L) {
    const int REP = 30;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    chain();
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    float ms = 0;
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < REP; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s)); CK(hipEventElapsedTime(&ms, e0, e1));
    std::printf("%-72s %.3f us per launch\n", name, 1e3 * ms / (REP * L));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return 0;
}

int main() {
    const int L = 200;
    const int ld = 4096;
    double *a, *b, *big;
    CK(hipMalloc(&a, 256 * 256 * sizeof(double))); CK(hipMalloc(&b, 256 * 256 * sizeof(double)));
    CK(hipMalloc(&big, (size_t)ld * 260 * sizeof(double)));
    CK(hipMemset(a, 0, 256 * 256 * sizeof(double))); CK(hipMemset(b, 0, 256 * 256 * sizeof(double)));
    CK(hipMemset(big, 0, (size_t)ld * 260 * sizeof(double)));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    Pad pad{};
    for (int wg : {1, 16, 63, 126, 256, 512}) {
        char nm[128]; std::snprintf(nm, sizeof nm, "trivial kernel, %d workgroups", wg);
        const int n = wg * 256;
        if (timeit(nm, s, [&]() { for (int k = 0; k < L; ++k) hipLaunchKernelGGL(k_small<0>, dim3(wg), dim3(256), 0, s, (k & 1) ? a : b, (k & 1) ? b : a, n, 1e-9); }, L)) return 1;
    }
    if (timeit("trivial kernel, 63 workgroups, 32 KB static LDS", s, [&]() { for (int k = 0; k < L; ++k) hipLaunchKernelGGL(k_small<32>, dim3(63), dim3(256), 0, s, (k & 1) ? a : b, (k & 1) ? b : a, 63 * 256, 1e-9); }, L)) return 1;
    if (timeit("trivial kernel, 63 workgroups, + 256 bytes of kernel arguments", s, [&]() { for (int k = 0; k < L; ++k) hipLaunchKernelGGL(k_args, dim3(63), dim3(256), 0, s, (k & 1) ? a : b, (k & 1) ? b : a, 63 * 256, 1e-9, pad); }, L)) return 1;
    if (timeit("two ALTERNATING trivial kernels, 63 workgroups", s, [&]() { for (int k = 0; k < L; ++k) { if (k & 1) hipLaunchKernelGGL(k_small<0>, dim3(63), dim3(256), 0, s, a, b, 63 * 256, 1e-9); else hipLaunchKernelGGL(k_small_b, dim3(63), dim3(256), 0, s, b, a, 63 * 256, 1e-9); } }, L)) return 1;
    for (int cols : {16, 64, 128, 256}) {
        char nm[128]; std::snprintf(nm, sizeof nm, "63 workgroups each streaming %d x 512 B columns (%d KB) the previous launch touched", cols, cols / 2);
        if (timeit(nm, s, [&]() { for (int k = 0; k < L; ++k) hipLaunchKernelGGL(k_tile, dim3(63), dim3(256), 0, s, big + (size_t)256 * ld, (const double*)big, ld, cols); }, L)) return 1;
    }
    for (int wg : {32, 126, 252}) {
        char nm[128]; std::snprintf(nm, sizeof nm, "%d workgroups each streaming 128 columns (64 KB)", wg);
        if (timeit(nm, s, [&]() { for (int k = 0; k < L; ++k) hipLaunchKernelGGL(k_tile, dim3(wg), dim3(256), 0, s, big + (size_t)256 * ld, (const double*)big, ld, 128); }, L)) return 1;
    }
    return 0;
}
