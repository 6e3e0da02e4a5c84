// Microbenchmark (round 5): can a HAND-ROLLED grid-wide flag wait between the two halves of a Lanczos
// step beat the 1.4 us a dependent launch costs on gfx950?  tools/micro/gridsync.hip measured cooperative_groups::grid.sync()
// (8.1 us at 63 workgroups) and one agent-scope counter polled by every workgroup (3.5 us).  Variants here, all with the
// step's data pattern (every workgroup writes a 1 KB record, waits, reads EVERY workgroup's record -- the partial-dot
// records of k_fop_finish -> k_lz_orth):
//   poll      : one counter, every workgroup's thread 0 polls it (the old measurement, for reference on this box)
//   sleep     : the same with s_sleep between polls (fewer requests hammering the counter's line)
//   bcast     : the LAST arriver (fetch_add returns G-1) writes a per-workgroup "go" word; each workgroup polls its OWN line
//   wave      : a whole wave arrives with ONE atomic per workgroup but the records are written/read with plain
//               stores/loads + explicit __threadfence (buffer_wbl2 / inv on gfx950) instead of agent-scope atomics per element
// and the reference point: the same exchange across a KERNEL BOUNDARY (two launches per phase pair, as the library does).
// build: hipcc -O3 --offload-arch=gfx950 flagsync.hip -o flagsync
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int REC = 128;                        // doubles per workgroup record (1 KB)

__device__ __forceinline__ void put_rec(double* buf, int slot, int g, int G, int t, double v) {
    if (t < REC) __hip_atomic_store(&buf[((size_t)slot * G + g) * REC + t], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double get_all(const double* buf, int slot, int G, int t) {
    double a = 0.0;
    for (int q = t; q < G * REC; q += 256) a += __hip_atomic_load(&buf[(size_t)slot * G * REC + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return a;
}

// the same read with 16 loads IN FLIGHT per thread (get_all above adds every value to one accumulator in a loop: the compiler
// waits for each relaxed atomic load before it issues the next -- 32 dependent round trips at G = 63, which is most of the
// 10 us the first variants show; this is the variant that says what the exchange itself costs)
__device__ __forceinline__ double get_all_batched(const double* buf, int slot, int G, int t) {
    double a = 0.0;
    const double* bp = buf + (size_t)slot * G * REC;
    const int tot = G * REC;
    for (int q0 = t; q0 < tot; q0 += 256 * 16) {
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int q = q0 + 256 * u;
            v[u] = __hip_atomic_load(&bp[q < tot ? q : t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) if (q0 + 256 * u < tot) a += v[u];
    }
    return a;
}

template <int MODE>
__global__ void __launch_bounds__(256) k_flag(double* buf, int iters, double* out, unsigned* ctr, unsigned* go) {
    const int g = blockIdx.x, G = gridDim.x, t = threadIdx.x;
    double acc = 0.0;
    for (int it = 0; it < iters; ++it) {
        const int slot = it & 1;
        if (MODE == 3 || MODE == 4) { if (t < REC) buf[((size_t)slot * G + g) * REC + t] = acc * 1e-30 + g + it; }
        else put_rec(buf, slot, g, G, t, acc * 1e-30 + g + it);
        if (MODE == 3) __threadfence();
        if (MODE == 4) __atomic_thread_fence(__ATOMIC_RELEASE);      // (every storing wave: L2 write-back of its lines)
        __syncthreads();
        if (t == 0) {
            const unsigned target = (unsigned)(it + 1) * G;
            if (MODE == 2) {
                const unsigned old = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                if (old + 1 == target) {
                    for (int q = 0; q < G; ++q) __hip_atomic_store(&go[q * 32], (unsigned)(it + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    long spins = 0;
                    while (__hip_atomic_load(&go[g * 32], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(it + 1) && ++spins < 20000000) {}
                }
            } else {
                __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                long spins = 0;
                while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < 20000000) {
                    if (MODE == 1) __builtin_amdgcn_s_sleep(2);
                }
            }
        }
        __syncthreads();
        if (MODE == 4) {
            __atomic_thread_fence(__ATOMIC_ACQUIRE);                 // buffer_inv: the plain loads below miss to memory once
            const double* bp = buf + (size_t)slot * G * REC;
            double a = 0.0;
            for (int q = t; q < G * REC; q += 256) a += bp[q];
            acc += a;
        } else if (MODE == 3) {
            __threadfence();
            double a = 0.0;
            for (int q = t; q < G * REC; q += 256) a += __builtin_nontemporal_load(&buf[(size_t)slot * G * REC + q]);
            acc += a;
        } else if (MODE == 5) acc += get_all_batched(buf, slot, G, t);
        else acc += get_all(buf, slot, G, t);
    }
    out[g * 256 + t] = acc;
}

// the same exchange through kernel boundaries: write records | read all records
__global__ void __launch_bounds__(256) k_write(double* buf, int slot, int it, const double* acc_in) {
    const int g = blockIdx.x, G = gridDim.x, t = threadIdx.x;
    if (t < REC) buf[((size_t)slot * G + g) * REC + t] = acc_in[g * 256 + t] * 1e-30 + g + it;
}
__global__ void __launch_bounds__(256) k_read(const double* buf, int slot, double* acc) {
    const int g = blockIdx.x, G = gridDim.x, t = threadIdx.x;
    double a = 0.0;
    for (int q = t; q < G * REC; q += 256) a += buf[(size_t)slot * G * REC + q];
    acc[g * 256 + t] += a;
}

int main() {
    const char* names[6] = {"poll (one counter)", "poll + s_sleep", "last arriver broadcasts per-WG go words", "plain stores + __threadfence, nontemporal loads",
                            "plain stores + release | acquire + plain loads", "poll (one counter), 16 record loads in flight"};
    for (int G : {63, 126}) {
        for (int pass = 0; pass < 1; ++pass) {}
        const int iters = 4000;
        double *buf, *out; unsigned *ctr, *go;
        hipMalloc(&buf, sizeof(double) * 2 * G * REC); hipMalloc(&out, sizeof(double) * G * 256); hipMalloc(&ctr, 4); hipMalloc(&go, 4 * 32 * G);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (int mode = 0; mode < 6; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                hipMemset(ctr, 0, 4); hipMemset(go, 0, 4 * 32 * G); hipMemset(buf, 0, sizeof(double) * 2 * G * REC);
                int it = iters;
                void* args[] = {&buf, &it, &out, &ctr, &go};
                void* fn = mode == 0 ? (void*)k_flag<0> : mode == 1 ? (void*)k_flag<1> : mode == 2 ? (void*)k_flag<2> : mode == 3 ? (void*)k_flag<3> : mode == 4 ? (void*)k_flag<4> : (void*)k_flag<5>;
                hipEventRecord(a);
                hipError_t e = hipLaunchCooperativeKernel(fn, dim3(G), dim3(256), args, 0, 0);     // (cooperative only for co-residency)
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                std::vector<double> h(G * 256); hipMemcpy(h.data(), out, sizeof(double) * G * 256, hipMemcpyDeviceToHost);
                double expect = 0.0;                   // sum over iterations of sum_g REC * (g + it)  (acc feedback is 1e-30-scaled)
                for (int q = 0; q < iters; ++q) expect += (double)REC * (G * (double)q + 0.5 * G * (G - 1.0));
                double got = 0.0; for (int t = 0; t < 256; ++t) got += h[t];
                if (rep) printf("G=%3d %-46s %.2f us per exchange (err %d, check rel %.1e)\n", G, names[mode], 1e3 * ms / iters, (int)e, (got - expect) / expect);
            }
        }
        {   // kernel boundaries
            double* acc; hipMalloc(&acc, sizeof(double) * G * 256); hipMemset(acc, 0, sizeof(double) * G * 256);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(a);
                for (int it = 0; it < iters; ++it) {
                    hipLaunchKernelGGL(k_write, dim3(G), dim3(256), 0, 0, buf, it & 1, it, (const double*)acc);
                    hipLaunchKernelGGL(k_read, dim3(G), dim3(256), 0, 0, (const double*)buf, it & 1, acc);
                }
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (rep) printf("G=%3d %-46s %.2f us per exchange = TWO dependent launches (the flag variants replace ONE of them)\n", G, "kernel boundaries (write | read)", 1e3 * ms / iters);
            }
            hipFree(acc);
        }
        hipFree(buf); hipFree(out); hipFree(ctr); hipFree(go);
    }
    return 0;
}
