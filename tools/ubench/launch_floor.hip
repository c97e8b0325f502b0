// Micro-benchmark (development aid): what an EMPTY kernel costs by grid shape and LDS footprint (event-timed, like
// bench.py's kernel timer) -- the floor under every launch of the per-step loop.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/ubench/launch_floor.hip -o /tmp/lf && /tmp/lf
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int LDS_BYTES>
__global__ void empty(float *sink) {
    __shared__ char lds[LDS_BYTES > 0 ? LDS_BYTES : 4];
    if (LDS_BYTES > 0) lds[threadIdx.x] = (char)threadIdx.x;
    if (sink == (float *)1) sink[0] = lds[0];
}

template <int LDS_BYTES>
static void run(const char *name, int grid, int block, float *sink, hipEvent_t e0, hipEvent_t e1) {
    float best = 1e9f, sum = 0.f;
    const int iters = 60;
    for (int i = 0; i < iters; ++i) {
        hipExtLaunchKernelGGL((empty<LDS_BYTES>), dim3(grid), dim3(block), 0, 0, e0, e1, 0, sink);
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
        sum += ms;
    }
    printf("%-34s avg %.2f us  min %.2f us\n", name, sum / iters * 1e3f, best * 1e3f);
}

int main() {
    float *sink;
    CHECK(hipMalloc(&sink, 64));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("1 x 64", 1, 64, sink, e0, e1);
        run<0>("256 x 256", 256, 256, sink, e0, e1);
        run<0>("1536 x 256", 1536, 256, sink, e0, e1);
        run<26000>("1536 x 256, 26 KB LDS", 1536, 256, sink, e0, e1);
        run<26000>("768 x 512, 26 KB LDS", 768, 512, sink, e0, e1);
        run<26000>("384 x 1024, 26 KB LDS", 384, 1024, sink, e0, e1);
        run<0>("6144 x 64", 6144, 64, sink, e0, e1);
        run<0>("393 x 256 (end-of-step kernel)", 393, 256, sink, e0, e1);
        run<26000>("10880 x 256, 26 KB LDS (2040x1356)", 10880, 256, sink, e0, e1);
    }
    return 0;
}
