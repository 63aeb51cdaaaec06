// How fast is ONE returning atomicAdd per workgroup on ONE address (block_alloc in k_shade: the pool's tail)?
// hipcc --offload-arch=gfx950 -O3 -o atomic_rate atomic_rate.hip && ./atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
template <int MODE>   // 0: no atomic; 1: returning atomic on one address; 2: returning, 64 addresses 256 B apart; 3: one address, result unused
__global__ void __launch_bounds__(256) k(uint32_t *counter, uint32_t *out, int work) {
    __shared__ uint32_t s_base;
    double x = (double)threadIdx.x;
    for (int i = 0; i < work; ++i) x = x * 1.0000001 + 0.5;   // some VALU work before the allocation
    if (threadIdx.x == 0) {
        uint32_t b = 0;
        if (MODE == 1) b = atomicAdd(counter, 3u);
        if (MODE == 2) b = atomicAdd(counter + 64 * (blockIdx.x & 63), 3u);
        if (MODE == 3) atomicAdd(counter, 3u);
        s_base = b;
    }
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = s_base + (uint32_t)x;
}
int main() {
    const int N = 340000;
    uint32_t *counter, *out;
    CK(hipMalloc(&counter, 64 * 64 * 4));
    CK(hipMalloc(&out, (size_t)N * 256 * 4));
    CK(hipMemset(counter, 0, 64 * 64 * 4));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int work : {0, 200, 2000}) {
        for (int mode = 0; mode < 4; ++mode) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(a));
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(N), dim3(256), 0, 0, counter, out, work);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(N), dim3(256), 0, 0, counter, out, work);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(N), dim3(256), 0, 0, counter, out, work);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(N), dim3(256), 0, 0, counter, out, work);
                CK(hipEventRecord(b));
                CK(hipEventSynchronize(b));
                float ms; CK(hipEventElapsedTime(&ms, a, b));
                if (ms < best) best = ms;
            }
            printf("work %4d mode %d: %.3f ms for %d workgroups = %.1f ns per workgroup\n", work, mode, best, N, best * 1e6 / N);
        }
    }
    return 0;
}
