// Microbenchmark: cost of a grid-wide barrier inside a cooperative kernel on gfx950, with a
// cross-workgroup data exchange per phase (each workgroup writes 64 doubles, reads its neighbour's).
// build: hipcc -O3 --offload-arch=gfx950 gridsync.hip -o gridsync
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
#include <vector>
namespace cg = cooperative_groups;

__global__ void __launch_bounds__(256) k_sync(double* buf, int iters, double* out) {
    cg::grid_group grid = cg::this_grid();
    const int g = blockIdx.x, G = gridDim.x, t = threadIdx.x;
    double acc = 0.0;
    for (int it = 0; it < iters; ++it) {
        if (t < 64) buf[(size_t)(it & 1) * G * 64 + g * 64 + t] = acc + g + it;
        grid.sync();
        if (t < 64) acc += buf[(size_t)(it & 1) * G * 64 + ((g + 1) % G) * 64 + t];
    }
    if (t < 64) out[g * 64 + t] = acc;
}

// hand-rolled barrier: one atomic counter, sense by iteration number
__global__ void __launch_bounds__(256) k_sync_atomic(double* buf, int iters, double* out, unsigned* ctr) {
    const int g = blockIdx.x, G = gridDim.x, t = threadIdx.x;
    double acc = 0.0;
    for (int it = 0; it < iters; ++it) {
        if (t < 64) __hip_atomic_store(&buf[(size_t)(it & 1) * G * 64 + g * 64 + t], acc + g + it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (t == 0) {
            __atomic_thread_fence(__ATOMIC_RELEASE);
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(it + 1) * G;
            long spins = 0;
            while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < 20000000) {}
        }
        __syncthreads();
        if (t < 64) acc += __hip_atomic_load(&buf[(size_t)(it & 1) * G * 64 + ((g + 1) % G) * 64 + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (t < 64) out[g * 64 + t] = acc;
}

int main() {
    for (int G : {63, 126, 252}) {
        const int iters = 2000;
        double *buf, *out; unsigned* ctr;
        hipMalloc(&buf, sizeof(double) * 2 * G * 64); hipMalloc(&out, sizeof(double) * G * 64); hipMalloc(&ctr, 4);
        hipMemset(buf, 0, sizeof(double) * 2 * G * 64);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        int it = iters; void* args[] = {&buf, &it, &out};
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a);
            hipError_t e = hipLaunchCooperativeKernel((void*)k_sync, dim3(G), dim3(256), args, 0, 0);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            std::vector<double> h(G * 64); hipMemcpy(h.data(), out, sizeof(double) * G * 64, hipMemcpyDeviceToHost);
            if (rep) printf("cg grid.sync  G=%3d: %.2f us per phase (err %d) check %.0f\n", G, 1e3 * ms / iters, (int)e, h[0]);
        }
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(ctr, 0, 4);
            hipEventRecord(a);
            void* args2[] = {&buf, &it, &out, &ctr};
            hipError_t e = hipLaunchCooperativeKernel((void*)k_sync_atomic, dim3(G), dim3(256), args2, 0, 0);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            std::vector<double> h(G * 64); hipMemcpy(h.data(), out, sizeof(double) * G * 64, hipMemcpyDeviceToHost);
            if (rep) printf("atomic barrier G=%3d: %.2f us per phase (err %d) check %.0f\n", G, 1e3 * ms / iters, (int)e, h[0]);
        }
    }
    return 0;
}
