// Micro-benchmark (development aid): the second load round of the tile pass's head -- 1536 workgroups x ~73 entries
// each gather one 64-byte record per entry by id (ids of spatially close gaussians, as tile lists hold) into LDS.
//   per_lane : lane = entry, four 16-byte loads of ITS record (what tile_list_head does): one wave instruction touches
//              64 different cache lines, 16 bytes of each
//   quad     : four lanes per entry, each loads one quarter of the record: one wave instruction covers 16 whole lines
//   ids_only : the first round alone (header + ids), for scale
//   empty    : no load at all (dispatch + kernel arguments + exit)
// Build it a second time with -mllvm -amdgpu-kernarg-preload-count=8 to see what the kernel-argument fetch costs.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/ubench/record_gather.hip -o /tmp/rg && /tmp/rg
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int ROW = 16 + 1024;

template <int MODE>
__global__ __launch_bounds__(256) void gather(const int *__restrict__ rows, const float4 *__restrict__ recs, float *sink) {
    __shared__ float4 st[4][256];
    const int tid = threadIdx.x;
    const int *row = rows + (size_t)blockIdx.x * ROW;
    const int count = MODE == 3 ? 0 : row[0];
    float acc = 0.f;
    if (MODE == 0) {
        const int id = row[16 + tid];
        if (tid < count) {
            const float4 *p = recs + 4 * (size_t)id;
            st[0][tid] = p[0], st[1][tid] = p[1], st[2][tid] = p[2], st[3][tid] = p[3];
        }
    } else if (MODE == 1) {
        // entry e = 64 * round + tid / 4, quarter = tid & 3; ids of the 64 entries of a round via LDS
        __shared__ int ids[256];
        ids[tid] = row[16 + tid];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = 64 * r + (tid >> 2);
            if (e < count) st[tid & 3][e] = recs[4 * (size_t)ids[e] + (tid & 3)];
        }
    } else if (MODE == 2) {
        const int id = row[16 + tid];
        if (tid < count) acc = (float)id;
    }
    __syncthreads();
    if (MODE < 2) acc = st[0][tid].x + st[1][tid].y + st[2][tid].z + st[3][tid].w;
    if (acc == 12345.678f) sink[0] = acc;
}

int main() {
    const int n = 50000, tiles_x = 48, tiles_y = 32, T = tiles_x * tiles_y;
    std::vector<float> gx(n), gy(n);
    srand(1);
    for (int g = 0; g < n; ++g) gx[g] = (rand() / (float)RAND_MAX) * 768, gy[g] = (rand() / (float)RAND_MAX) * 512;
    std::vector<int> rows((size_t)T * ROW, 0);
    long total = 0;
    for (int g = 0; g < n; ++g) {
        const int x0 = (int)((gx[g] - 4.5f) / 16), x1 = (int)((gx[g] + 4.5f) / 16), y0 = (int)((gy[g] - 4.5f) / 16),
                  y1 = (int)((gy[g] + 4.5f) / 16);
        for (int yy = y0; yy <= y1; ++yy)
            for (int xx = x0; xx <= x1; ++xx)
                if (xx >= 0 && yy >= 0 && xx < tiles_x && yy < tiles_y) {
                    int *row = rows.data() + (size_t)(yy * tiles_x + xx) * ROW;
                    if (row[0] < 256) row[16 + row[0]++] = g, ++total;
                }
    }
    int *d_rows;
    float4 *d_recs;
    float *d_sink;
    CHECK(hipMalloc(&d_rows, rows.size() * 4));
    CHECK(hipMalloc(&d_recs, (size_t)n * 64));
    CHECK(hipMalloc(&d_sink, 64));
    CHECK(hipMemcpy(d_rows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemset(d_recs, 0, (size_t)n * 64));
    // something that evicts caches between timed launches, as the end-of-step kernel does in the real loop
    float *d_flush;
    const size_t flush_n = 128u << 20;
    CHECK(hipMalloc(&d_flush, flush_n));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("entries %ld over %d tiles\n", total, T);
    const char *names[4] = {"per_lane", "quad", "ids_only", "empty"};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 4; ++mode) {
            float best = 1e9f, sum = 0.f;
            const int iters = 40;
            for (int i = 0; i < iters; ++i) {
                CHECK(hipMemsetAsync(d_flush, i, flush_n, 0));
                CHECK(hipMemsetAsync(d_recs, 0, (size_t)n * 64, 0));  // records freshly written, as in the loop
                if (mode == 0)
                    hipExtLaunchKernelGGL((gather<0>), dim3(T), dim3(256), 0, 0, e0, e1, 0, d_rows, d_recs, d_sink);
                else if (mode == 1)
                    hipExtLaunchKernelGGL((gather<1>), dim3(T), dim3(256), 0, 0, e0, e1, 0, d_rows, d_recs, d_sink);
                else if (mode == 2)
                    hipExtLaunchKernelGGL((gather<2>), dim3(T), dim3(256), 0, 0, e0, e1, 0, d_rows, d_recs, d_sink);
                else
                    hipExtLaunchKernelGGL((gather<3>), dim3(T), dim3(256), 0, 0, e0, e1, 0, d_rows, d_recs, d_sink);
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
                sum += ms;
            }
            printf("%-9s avg %.2f us  min %.2f us\n", names[mode], sum / iters * 1e3f, best * 1e3f);
        }
    return 0;
}
