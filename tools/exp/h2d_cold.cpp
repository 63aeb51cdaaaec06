// Experiment: what does the first large pageable host-to-device copy of a process cost, and what avoids it?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t MB = 1 << 20, N = 100 * MB;
    double t = now();
    hipSetDevice(0); hipFree(nullptr);
    printf("init %.1f ms\n", now() - t);
    void *d = nullptr, *d2 = nullptr;
    t = now(); hipMalloc(&d, N); hipMalloc(&d2, N); printf("2 x hipMalloc 100MB %.1f ms\n", now() - t);
    char *a = (char *)malloc(N); memset(a, 1, N);
    t = now(); hipMemcpy(d, a, 4096, hipMemcpyHostToDevice); printf("first 4 KB copy %.2f ms\n", now() - t);
    t = now(); hipMemcpy(d, a, N, hipMemcpyHostToDevice); printf("first 100MB pageable copy %.1f ms\n", now() - t);
    t = now(); hipMemcpy(d, a, N, hipMemcpyHostToDevice); printf("same buffer again %.1f ms\n", now() - t);
    char *b = (char *)malloc(N); memset(b, 2, N);
    t = now(); hipMemcpy(d2, b, N, hipMemcpyHostToDevice); printf("another fresh 100MB buffer %.1f ms\n", now() - t);
    t = now(); for (size_t o = 0; o < N; o += 4 * MB) hipMemcpy((char *)d2 + o, b + o, 4 * MB, hipMemcpyHostToDevice); printf("same in 4MB pieces %.1f ms\n", now() - t);
    char *c = (char *)malloc(N); memset(c, 3, N);
    t = now(); for (size_t o = 0; o < N; o += 1 * MB) hipMemcpy((char *)d2 + o, c + o, 1 * MB, hipMemcpyHostToDevice); printf("fresh buffer in 1MB pieces %.1f ms\n", now() - t);
    void *h = nullptr;
    t = now(); hipHostMalloc(&h, N, hipHostMallocDefault); printf("hipHostMalloc 100MB %.1f ms\n", now() - t);
    t = now(); memcpy(h, a, N); printf("memcpy into pinned %.1f ms\n", now() - t);
    t = now(); hipMemcpy(d, h, N, hipMemcpyHostToDevice); printf("pinned copy %.1f ms\n", now() - t);
    char *e = (char *)malloc(N); memset(e, 4, N);
    t = now(); hipHostRegister(e, N, hipHostRegisterDefault); printf("hipHostRegister fresh 100MB %.1f ms\n", now() - t);
    t = now(); hipMemcpy(d, e, N, hipMemcpyHostToDevice); printf("registered copy %.1f ms\n", now() - t);
    return 0;
}
