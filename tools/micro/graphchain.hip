// Dependent-launch floor, stream launches vs one captured hipGraph: a chain of L small dependent kernels (63 workgroups
// of 256 threads each, the shape of a Lanczos step launch at n = 4000), timed with events on the launching stream.
// hipcc --offload-arch=gfx950 -O2 -o tools/micro/graphchain tools/micro/graphchain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void __launch_bounds__(256) k_step(double* __restrict__ v, const double* __restrict__ u, int n, double a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) v[i] = a * u[i] + v[i];
}

int main() {
    const int n = 63 * 256, L = 254, REP = 50;
    double *a, *b;
    CK(hipMalloc(&a, n * sizeof(double))); CK(hipMalloc(&b, n * sizeof(double)));
    CK(hipMemset(a, 0, n * sizeof(double))); CK(hipMemset(b, 0, n * sizeof(double)));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto chain = [&]() { for (int k = 0; k < L; ++k) hipLaunchKernelGGL(k_step, dim3(63), dim3(256), 0, s, (k & 1) ? a : b, (k & 1) ? b : a, n, 1e-9); };
    // stream launches
    chain(); CK(hipStreamSynchronize(s));
    float ms_stream = 0;
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < REP; ++r) chain();
    CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s)); CK(hipEventElapsedTime(&ms_stream, e0, e1));
    // one graph of the same chain
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    chain();
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    float ms_graph = 0;
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < REP; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s)); CK(hipEventElapsedTime(&ms_graph, e0, e1));
    std::printf("chain of %d dependent launches (63 x 256 threads): stream %.3f us per launch, hipGraph %.3f us per launch\n",
                L, 1e3 * ms_stream / (REP * L), 1e3 * ms_graph / (REP * L));
    return 0;
}
