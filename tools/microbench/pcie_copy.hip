// GPU box: device->host copy rates into hipHostMalloc memory (context for vsmpc_solve_batch's pinned path).
// hipcc --offload-arch=gfx950 -O2 pcie_copy.hip -o pcie_copy && ./pcie_copy
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t bytes = 4096ull * 588 * 8;
    void *d, *h0, *h1;
    hipMalloc(&d, bytes);
    hipHostMalloc(&h0, bytes, hipHostMallocDefault);
    hipHostMalloc(&h1, bytes, hipHostMallocNonCoherent);
    void* hp = malloc(bytes); memset(hp, 1, bytes);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    struct { const char* name; void* p; hipStream_t st; } cases[] = {
        {"default flags, null stream", h0, nullptr}, {"default flags, own stream", h0, s},
        {"non-coherent, own stream", h1, s}, {"pageable, null stream", hp, nullptr}};
    for (auto& c : cases) {
        for (int touch = 0; touch < 2; ++touch) {
            if (touch) memset(c.p, 0, bytes);   // CPU has touched every page
            for (int i = 0; i < 3; ++i) { hipMemcpyAsync(c.p, d, bytes, hipMemcpyDeviceToHost, c.st); hipStreamSynchronize(c.st); }
            const double t = now();
            for (int i = 0; i < 10; ++i) { hipMemcpyAsync(c.p, d, bytes, hipMemcpyDeviceToHost, c.st); hipStreamSynchronize(c.st); }
            printf("D2H %-30s %s: %.1f GB/s\n", c.name, touch ? "(CPU-touched)" : "(fresh)      ", bytes * 10 / (now() - t) / 1e9);
        }
    }
    return 0;
}
