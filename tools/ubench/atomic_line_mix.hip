// Micro-benchmark (development aid): are device-scope atomics safe next to plain stores / plain loads of OTHER words of
// the same cache line, from workgroups on all eight XCDs, in one launch and across consecutive launches?  (The status
// words of the tile pass: status[0] = 1 by a plain store from every non-empty tile, atomicOr on status[1] by a tile whose
// row overflowed -- csrc/gi2d_fused_core.h; and a counter that one kernel bumps with atomicAdd and the next reads with a
// plain load while both kernels read and write neighbouring words of the line.)
//   test A: per launch, every workgroup stores s[0] = 1 (plain) at its end; workgroups b % 61 == 0 do atomicOr(&s[1], bit)
//           at their start or end; the host checks that no bit of s[1] was lost.
//   test B: kernel 1: workgroup 0 stores s[0] = launch (plain), every workgroup reads s[2] (plain) and workgroups
//           b % 7 == 0 atomicAdd(&s[17], 1); kernel 2: every workgroup reads s[17] with a plain load and writes what it
//           saw to out[b]; the host checks that all saw the full count.  Then the same with the counter at word 32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void test_a(int *s, int late, float *sink) {
    const int b = blockIdx.x;
    float x = threadIdx.x;
    if (!late && b % 61 == 0 && threadIdx.x == 0) atomicOr(&s[1], 1 << ((b / 61) % 31));
    for (int i = 0; i < 200 + (b % 13) * 40; ++i) x = x * 1.0001f + 0.5f;  // some work, uneven
    if (late && b % 61 == 0 && threadIdx.x == 0) atomicOr(&s[1], 1 << ((b / 61) % 31));
    if (threadIdx.x == 0) s[0] = 1;
    if (x == 12345.f) sink[0] = x;
}
__global__ __launch_bounds__(256) void test_b1(int *s, int word, int launch, int *sink) {
    const int b = blockIdx.x;
    if (b == 0 && threadIdx.x == 0) s[0] = launch;
    const int v = s[2];
    if (b % 7 == 0 && threadIdx.x == 0) atomicAdd(&s[word], 1);
    if (v == -77) sink[0] = v;
}
__global__ __launch_bounds__(256) void test_b2(const int *s, int word, int *out) {
    if (threadIdx.x == 0) out[blockIdx.x] = s[word];
}
__global__ void reset(int *s, int word) { s[word] = 0; }

int main() {
    int *s, *out; float *sink; int *isink;
    CHECK(hipMalloc(&s, 4096)); CHECK(hipMalloc(&out, 4096 * 4)); CHECK(hipMalloc(&sink, 64)); CHECK(hipMalloc(&isink, 64));
    const int G = 1536;
    for (int late = 0; late < 2; ++late) {
        int lost = 0;
        for (int it = 0; it < 2000; ++it) {
            CHECK(hipMemsetAsync(s, 0, 4096, 0));
            hipLaunchKernelGGL(test_a, dim3(G), dim3(256), 0, 0, s, late, sink);
            int h[4];
            CHECK(hipMemcpy(h, s, 16, hipMemcpyDeviceToHost));
            unsigned want = 0;
            for (int b = 0; b < G; b += 61) want |= 1u << ((b / 61) % 31);
            if ((unsigned)h[1] != want || h[0] != 1) ++lost;
        }
        printf("test A (atomicOr %s the plain stores): %d of 2000 launches lost a bit\n", late ? "after" : "before", lost);
    }
    for (int word : {17, 32}) {
        int bad = 0;
        std::vector<int> h(G);
        CHECK(hipMemset(s, 0, 4096));
        for (int it = 0; it < 2000; ++it) {
            hipLaunchKernelGGL(test_b1, dim3(G), dim3(256), 0, 0, s, word, it, isink);
            hipLaunchKernelGGL(test_b2, dim3(G), dim3(256), 0, 0, s, word, out);
            hipLaunchKernelGGL(reset, dim3(1), dim3(1), 0, 0, s, word);
            CHECK(hipMemcpy(h.data(), out, G * 4, hipMemcpyDeviceToHost));
            const int want = (G + 6) / 7;
            int wrong = 0;
            for (int b = 0; b < G; ++b) wrong += h[b] != want;
            if (wrong) ++bad;
        }
        printf("test B (counter at word %d, plain store to word 0 and plain loads of word 2 in the same launch): %d of 2000 "
               "launches had a workgroup that read a wrong count in the next launch\n", word, bad);
    }
    return 0;
}
