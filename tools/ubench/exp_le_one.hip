// Check (development aid) behind gi2d_common.h::AlphaRule::clamp: v_exp_f32 of a NON-POSITIVE argument never exceeds
// 1.0 -- every float x <= 0 (all 2^31 bit patterns with the sign bit set, -0.0 and the denormals included) is tried --
// so with 0 <= opac <= 1 a landing pair (sigma' >= 0) has opac * exp2(-sigma') <= 1 and min(1, .) is the identity.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/ubench/exp_le_one.hip -o /tmp/e1 && /tmp/e1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void sweep(unsigned *max_bits, unsigned long long *above_one) {
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;  // 2^20 threads x 2^11 patterns
    unsigned mx = 0;
    unsigned long long bad = 0;
    for (unsigned i = 0; i < 2048u; ++i) {
        const unsigned bits = 0x80000000u | (t * 2048u + i);  // x <= 0 (NaNs give NaN: skipped)
        const float x = __uint_as_float(bits);
        if (x != x) continue;
        const float v = __builtin_amdgcn_exp2f(x);     // what pair_vis() issues: exp2(-sigma'), sigma' >= 0
        const float w = __builtin_amdgcn_exp2f(-(-x)); // the negated-operand form the compiler emits (v_exp_f32 -v)
        const unsigned vb = __float_as_uint(v > w ? v : w);
        mx = vb > mx ? vb : mx;
        bad += (v > 1.f) + (w > 1.f);
    }
    atomicMax(max_bits, mx);
    if (bad) atomicAdd(above_one, bad);
}

int main() {
    unsigned *mx;
    unsigned long long *bad;
    CHECK(hipMalloc(&mx, 4));
    CHECK(hipMalloc(&bad, 8));
    CHECK(hipMemset(mx, 0, 4));
    CHECK(hipMemset(bad, 0, 8));
    hipLaunchKernelGGL(sweep, dim3(4096), dim3(256), 0, 0, mx, bad);
    unsigned h_mx;
    unsigned long long h_bad;
    CHECK(hipMemcpy(&h_mx, mx, 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(&h_bad, bad, 8, hipMemcpyDeviceToHost));
    float f;
    __builtin_memcpy(&f, &h_mx, 4);
    printf("max v_exp_f32(x <= 0) = %.9g (bits 0x%08x), results above 1.0: %llu of 2^31 arguments\n", f, h_mx, h_bad);
    return h_bad ? 1 : 0;
}
