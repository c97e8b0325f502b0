// Micro-benchmark (development aid): is a kernel boundary between the tile pass and the update kernel
// cheaper than a device-side hand-off inside ONE launch?  A model of the two kernels' shapes, not their arithmetic:
//   "tile" workgroups   1536 x 256 lanes, 26 KB of LDS (six per CU: one residency round), a dependent load, ~12 us of
//                       arithmetic of uneven length, then 72 rows of 64 bytes stored per workgroup (write-through);
//   "update" workgroups 197 x 256 lanes (same LDS: same kernel in the fused form), a first round of loads that does not
//                       depend on the tiles, the rows' loads, ~1.5 us of arithmetic, 96 bytes stored per lane.
// two:   the two as two launches on one stream (what the product does), a chain of ITER such pairs;
// fused: ONE launch of 1536 + 197 workgroups per iteration: a tile workgroup signals its rows (s_waitcnt vmcnt(0) behind
//        its write-through stores, then an agent-scope atomic add on one of 32 counter shards); an update workgroup
//        issues its independent loads, then polls the 32 shards (sc1 loads, s_sleep between polls) until all 1536 have
//        arrived, then loads the rows (sc1).  Workgroups are dispatched in index order, so every tile workgroup is
//        resident before an update workgroup can start (it needs the LDS a finished tile workgroup frees): the wait
//        cannot deadlock -- and is bounded anyway (a poll budget; a workgroup that runs out of it flags the run).
// Prints microseconds per iteration of either form and the flag.   hipcc --offload-arch=gfx950 -O3 fused_boundary.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int TILES = 1536, UPD = 197, SHARDS = 32, ROWS = 72, N = 50000;
struct Args {
    float4 *rows;        // [TILES][ROWS][4]
    const float4 *recs;  // [N][4]
    float *state;        // [N][24]
    float *out;          // [N][24]
    unsigned *done;      // [SHARDS * 32] (one 128-byte line per shard)
    unsigned *flag;
    unsigned epoch;      // arrivals expected per shard = epoch * TILES / SHARDS
    int work;
};

__device__ __forceinline__ float spin_work(float x, int n) {
    for (int i = 0; i < n; ++i) x = __builtin_fmaf(x, 1.0001f, 0.5f);
    return x;
}
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store16_wt(float4 *base, unsigned off, float4 v) {
    const u4 r = {(unsigned)__float_as_int(v.x), (unsigned)__float_as_int(v.y), (unsigned)__float_as_int(v.z), (unsigned)__float_as_int(v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(r, __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000), (int)off, 0, 16);
}

__device__ void tile_part(const Args &a, int tile, bool signal) {
    __shared__ float lds[26 * 1024 / 4];
    const int tid = threadIdx.x;
    const float4 r = a.recs[4 * ((tile * 67 + tid * 13) % N)];  // a dependent gather
    lds[tid] = r.x;
    __syncthreads();
    float x = spin_work(lds[(tid + 1) & 255] + r.y, a.work + (tile % 13) * (a.work / 40));
    __syncthreads();
    if (tid < ROWS) {
        const unsigned off = (unsigned)(((size_t)tile * ROWS + tid) * 64);
        store16_wt(a.rows, off, make_float4(x, x + 1.f, x + 2.f, x + 3.f));
        store16_wt(a.rows, off + 16, make_float4(x, x, x, x));
        store16_wt(a.rows, off + 32, make_float4(x, 0.f, 0.f, (float)tid));
    }
    if (signal) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(&a.done[(tile % SHARDS) * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ void update_part(const Args &a, int block, bool wait) {
    const int tid = threadIdx.x, g = block * 256 + tid;
    float s[24];
    const bool in = g < N;
    if (in) {
#pragma unroll
        for (int q = 0; q < 24; ++q) s[q] = a.state[(size_t)g * 24 + q];
    }
    if (wait) {
        __shared__ unsigned ok;
        if (tid < 64) {  // one wave polls: lane l reads shard l
            unsigned budget = 20000;  // x ~0.1 us: a bounded wait
            for (;;) {
                unsigned v = tid < SHARDS ? __hip_atomic_load(&a.done[tid * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
                unsigned sum = v;
                for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, 64);
                if (sum >= a.epoch * TILES) { if (tid == 0) ok = 1; break; }
                if (--budget == 0) { if (tid == 0) { ok = 0; atomicOr(a.flag, 1u); } break; }
                __builtin_amdgcn_s_sleep(8);
            }
        }
        __syncthreads();
        if (!ok) return;
    }
    if (!in) return;
    // the rows of "its tiles": two dependent 64-byte rows per lane, read past the L2 (they were written through)
    float acc = 0.f;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const size_t row = ((size_t)(g * 31 + q * 977) % ((size_t)TILES * ROWS));
        const float *p = reinterpret_cast<const float *>(a.rows + 4 * row);
        acc += __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) +
               __hip_atomic_load(p + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const float x = spin_work(acc + s[0], a.work / 8);
#pragma unroll
    for (int q = 0; q < 24; ++q) a.out[(size_t)g * 24 + q] = s[q] + x;
}

__global__ __launch_bounds__(256, 6) void tile_kernel(Args a) { tile_part(a, blockIdx.x, false); }
__global__ __launch_bounds__(256, 6) void update_kernel(Args a) { update_part(a, blockIdx.x, false); }
__global__ __launch_bounds__(256, 6) void fused_kernel(Args a) {
    if (blockIdx.x < TILES)
        tile_part(a, blockIdx.x, true);
    else
        update_part(a, blockIdx.x - TILES, true);
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 300, work = argc > 2 ? atoi(argv[2]) : 3000;
    Args a;
    CHECK(hipMalloc(&a.rows, (size_t)TILES * ROWS * 64));
    CHECK(hipMalloc((void **)&a.recs, (size_t)N * 64));
    CHECK(hipMalloc(&a.state, (size_t)N * 96));
    CHECK(hipMalloc(&a.out, (size_t)N * 96));
    CHECK(hipMalloc(&a.done, SHARDS * 128));
    CHECK(hipMalloc(&a.flag, 4));
    CHECK(hipMemset((void *)a.recs, 0, (size_t)N * 64));
    CHECK(hipMemset(a.state, 0, (size_t)N * 96));
    CHECK(hipMemset(a.flag, 0, 4));
    a.work = work;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        float ms_two = 0.f, ms_fused = 0.f, ms_tile = 0.f;
        a.epoch = 0;
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(tile_kernel, dim3(TILES), dim3(256), 0, 0, a);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms_tile, e0, e1));
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < iters; ++i) {
            hipLaunchKernelGGL(tile_kernel, dim3(TILES), dim3(256), 0, 0, a);
            hipLaunchKernelGGL(update_kernel, dim3(UPD), dim3(256), 0, 0, a);
        }
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms_two, e0, e1));
        CHECK(hipMemset(a.done, 0, SHARDS * 128));
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < iters; ++i) {
            a.epoch = (unsigned)(i + 1);
            hipLaunchKernelGGL(fused_kernel, dim3(TILES + UPD), dim3(256), 0, 0, a);
        }
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms_fused, e0, e1));
        unsigned flag = 0;
        CHECK(hipMemcpy(&flag, a.flag, 4, hipMemcpyDeviceToHost));
        printf("work %d: tile kernel alone %.2f us, tile + update as two launches %.2f us per iteration, fused in one launch %.2f us "
               "(timed-out waits: %s)\n", work, 1e3f * ms_tile / iters, 1e3f * ms_two / iters, 1e3f * ms_fused / iters, flag ? "YES" : "none");
    }
    return 0;
}
