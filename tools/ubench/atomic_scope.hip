// Micro-benchmark (development aid): cost of the bucket fill's returning int atomics by scope.
//   agent scope     -- what atomicAdd() emits: coherent across the 8 XCDs, executed at the memory side
//   workgroup scope -- executed in the issuing XCD's L2; usable across workgroups only if every workgroup that
//                      touches a counter sits on the same XCD (counters replicated per XCD, selected by XCC_ID)
// 111k increments on 6144 counters (one 64-byte line each) from 50k lanes, as in the N=50 000 / 768x512 step.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int SCOPE, bool STORE, int EVERY = 1>
__global__ __launch_bounds__(256) void fill(int n, const int4 *__restrict__ targets, int *cursors, int *buckets,
                                            int sets_stride) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    if (EVERY == 0) { if (targets[g].x == -77) buckets[0] = g; return; }
    if (EVERY > 1 && (g % EVERY) != 0) { if (targets[g].x == -77) buckets[0] = g; return; }
    int xcc = 0;
    if (SCOPE == __HIP_MEMORY_SCOPE_WORKGROUP) xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 7;  // HW_REG_XCC_ID
    int *cur = cursors + (size_t)xcc * sets_stride * 16;
    int *buk = buckets + (size_t)xcc * sets_stride * 256;
    const int4 t = targets[g];
    const int c[4] = {t.x, t.y, t.z, t.w};
    int p[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
        p[q] = c[q] >= 0 ? __hip_atomic_fetch_add(&cur[c[q] * 16], 1, __ATOMIC_RELAXED, SCOPE) : 256;
    if (STORE) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (p[q] < 256) buk[c[q] * 256 + p[q]] = g;
    } else {
        if (p[0] + p[1] + p[2] + p[3] == -12345) buk[0] = g;
    }
}

__global__ void reset(int n, int *cursors) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) cursors[i * 16] = 0;
}

int main() {
    const int n = 50000, tiles_x = 48, tiles_y = 32, T = tiles_x * tiles_y, C = T * 4, SETS = 8;
    std::vector<int4> tg(n);
    srand(1);
    long total = 0;
    for (int g = 0; g < n; ++g) {
        const float x = (rand() / (float)RAND_MAX) * 768, y = (rand() / (float)RAND_MAX) * 512;
        const int x0 = (int)((x - 4.5f) / 16), x1 = (int)((x + 4.5f) / 16), y0 = (int)((y - 4.5f) / 16), y1 = (int)((y + 4.5f) / 16);
        int k = 0, c[4] = {-1, -1, -1, -1};
        for (int yy = y0; yy <= y1; ++yy)
            for (int xx = x0; xx <= x1; ++xx)
                if (xx >= 0 && yy >= 0 && xx < tiles_x && yy < tiles_y && k < 4) c[k++] = (yy * tiles_x + xx) * 4 + (g & 3);
        total += k;
        tg[g] = make_int4(c[0], c[1], c[2], c[3]);
    }
    int4 *d_t;
    int *d_c, *d_b;
    CHECK(hipMalloc(&d_t, n * sizeof(int4)));
    CHECK(hipMalloc(&d_c, (size_t)SETS * C * 16 * sizeof(int)));
    CHECK(hipMalloc(&d_b, (size_t)SETS * C * 256 * sizeof(int)));
    CHECK(hipMemcpy(d_t, tg.data(), n * sizeof(int4), hipMemcpyHostToDevice));
    CHECK(hipMemset(d_c, 0, (size_t)SETS * C * 16 * sizeof(int)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("%ld increments on %d counters\n", total, C);
    auto run = [&](const char *name, auto kern, int bs) {
        float best = 1e9, sum = 0;
        const int reps = 50;
        for (int r = 0; r < reps + 5; ++r) {
            hipLaunchKernelGGL(reset, dim3((SETS * C + 255) / 256), dim3(256), 0, 0, SETS * C, d_c);
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3((n + bs - 1) / bs), dim3(bs), 0, 0, n, d_t, d_c, d_b, C);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 5) { sum += ms; best = ms < best ? ms : best; }
        }
        printf("%-46s block %3d: mean %.2f us, best %.2f us\n", name, bs, sum / reps * 1e3, best * 1e3);
    };
    run("no atomics (launch + one load)", fill<__HIP_MEMORY_SCOPE_AGENT, false, 0>, 256);
    run("1 lane in 4 fills", fill<__HIP_MEMORY_SCOPE_AGENT, true, 4>, 256);
    run("1 lane in 8 fills", fill<__HIP_MEMORY_SCOPE_AGENT, true, 8>, 256);
    run("1 lane in 16 fills", fill<__HIP_MEMORY_SCOPE_AGENT, true, 16>, 256);
    for (int bs : {256}) {
        run("agent scope, atomics + id stores", fill<__HIP_MEMORY_SCOPE_AGENT, true>, bs);
        run("agent scope, atomics only", fill<__HIP_MEMORY_SCOPE_AGENT, false>, bs);
        run("workgroup scope (per-XCD sets), atomics + stores", fill<__HIP_MEMORY_SCOPE_WORKGROUP, true>, bs);
        run("workgroup scope (per-XCD sets), atomics only", fill<__HIP_MEMORY_SCOPE_WORKGROUP, false>, bs);
    }
    // correctness of the per-XCD sets: counters summed over the sets must equal the agent-scope counts
    std::vector<int> a((size_t)C * 16), b((size_t)SETS * C * 16);
    hipLaunchKernelGGL(reset, dim3((SETS * C + 255) / 256), dim3(256), 0, 0, SETS * C, d_c);
    hipLaunchKernelGGL((fill<__HIP_MEMORY_SCOPE_AGENT, false>), dim3((n + 255) / 256), dim3(256), 0, 0, n, d_t, d_c, d_b, C);
    CHECK(hipMemcpy(a.data(), d_c, a.size() * sizeof(int), hipMemcpyDeviceToHost));
    hipLaunchKernelGGL(reset, dim3((SETS * C + 255) / 256), dim3(256), 0, 0, SETS * C, d_c);
    hipLaunchKernelGGL((fill<__HIP_MEMORY_SCOPE_WORKGROUP, false>), dim3((n + 255) / 256), dim3(256), 0, 0, n, d_t, d_c, d_b, C);
    CHECK(hipMemcpy(b.data(), d_c, b.size() * sizeof(int), hipMemcpyDeviceToHost));
    long bad = 0, used[8] = {0};
    for (int c = 0; c < C; ++c) {
        int s = 0;
        for (int k = 0; k < SETS; ++k) { s += b[((size_t)k * C + c) * 16]; used[k] += b[((size_t)k * C + c) * 16]; }
        bad += s != a[(size_t)c * 16];
    }
    printf("per-XCD sets vs agent scope: %ld counters differ; increments per set:", bad);
    for (int k = 0; k < 8; ++k) printf(" %ld", used[k]);
    printf("\n");
    return 0;
}
