// Population changes of a fit without leaving the device: non-positive-definite pruning
// (GaussianImage_Covariance.non_semi_definite_prune, models/gaussianimage_covariance.py:352-382, every `prune_iter`
// iterations, train.py:147-148) and error-driven growth (SimpleTrainer2d.add_sample_positions, train.py:85-118, with
// densification_postfix, models/gaussianimage_covariance.py:307-350, every `grow_iter` iterations).
//
// The reference does both on the host side of torch: boolean indexing (a device->host count for every output shape),
// torch.topk, optimizer-state surgery with new nn.Parameters, `.item()`s.  Here the number of live gaussians is a
// 32-bit word in HBM (gi2d_train_state::num_points_dev): the fitting kernels read it, these two operations update it,
// and the host only keeps an UPPER BOUND for launch sizes -- nothing in the loop waits for the device.
//
//   prune   flag (covariance + bound positive definite: det > 0, both diagonal entries > 0, :372-379) -> exclusive
//           scan -> every per-gaussian array (parameters, optimizer moments, bound, opacity) is compacted in order,
//           as boolean indexing does, through a scratch copy; nothing moves when nothing is pruned (the usual case)
//   grow    per-pixel error sum |clamp(render) - gt| over the channels -> the k largest (k = the growth budget of
//           train.py:91-97, from the live count) by a radix select on the float bits, ties to the lower pixel index ->
//           ordered by (error descending, index ascending) = torch.topk's sorted output with a stable tie rule ->
//           centre = (index mod W, index div W), covariance = rand + (0.5, 0, 0.5), colour 0; non-definite draws are
//           dropped in order (densification_postfix :317-320); the survivors are appended with zero optimizer moments
//           and the low-pass bound of the NEW population size (:343-348).
//           The reference normalises the errors by their sum before top-k (train.py:87); dividing by a positive constant
//           keeps the order, so the selection works on the raw sums.
// Both end with the caller re-initialising the fast workspace (gaussian ids were renumbered / appended).
#include "gi2d_common.h"

namespace gi2d {
int launch_workspace_init(void *ws, int n, int tiles_x, int tiles_y, const int32_t *only_if_moved, gi2d_stream_t st);  // gi2d_fast.hip
}

namespace gi2d {

#define GI2D_DENSIFY_MAX_ARRAYS 24
struct RowArrays {  // the per-gaussian arrays of a fit that move together; width = floats per row
    int count;
    float *ptr[GI2D_DENSIFY_MAX_ARRAYS];
    int width[GI2D_DENSIFY_MAX_ARRAYS];
    int offset[GI2D_DENSIFY_MAX_ARRAYS + 1];  // float offset of the array's row inside a scratch row
};

// block-wide exclusive scan of one int per lane (1024 lanes), returns the exclusive prefix and the block total
__device__ __forceinline__ int block_exclusive_scan(int v, int *wsum /* [17] LDS */, int &total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, waves = (int)blockDim.x >> 6;
    const int incl = wave_inclusive_scan(v);
    __syncthreads();  // wsum may still be read from the previous use
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    int base = 0;
    total = 0;
    for (int k = 0; k < waves; ++k) {
        const int s = wsum[k];
        if (k < wv) base += s;
        total += s;
    }
    return base + incl - v;
}

// models/gaussianimage_covariance.py:372-379
__device__ __forceinline__ bool positive_definite(float a, float b, float c) {
#pragma clang fp contract(off)
    return (a * c - b * b > 0.f) && (a > 0.f) && (c > 0.f);
}

// ------------------------------------------------------------------------------------------------ prune
// One workgroup of 1024 lanes: flags + scan in one sweep.  pos[g] = new row of gaussian g (-1: pruned);
// counts = {new population, old population}.
__global__ __launch_bounds__(1024) void prune_scan_kernel(const int32_t *__restrict__ n_dev, int n_bound,
                                                          const float *__restrict__ chol,
                                                          const float *__restrict__ bound, int bound_stride,
                                                          int32_t *__restrict__ pos, int32_t *__restrict__ counts) {
    __shared__ int wsum[17];
    const int n = min(*n_dev, n_bound);
    int run = 0;
    for (int base = 0; base < n; base += 1024) {
        const int g = base + threadIdx.x;
        int keep = 0;
        if (g < n) {
            const float *bd = bound + (size_t)bound_stride * g;
            keep = positive_definite(chol[3 * g] + bd[0], chol[3 * g + 1] + bd[1], chol[3 * g + 2] + bd[2]) ? 1 : 0;
        }
        int total;
        const int ex = block_exclusive_scan(keep, wsum, total);
        if (g < n) pos[g] = keep ? run + ex : -1;
        run += total;
    }
    if (threadIdx.x == 0) {
        counts[0] = run;
        counts[1] = n;
    }
}

// every kept row -> its new position in the scratch copy (nothing to do when nothing was pruned, or when pruning
// would leave no gaussian at all: the reference's guard `if to_prune and n - to_prune > 0` of trainer.py keeps them)
__global__ __launch_bounds__(256) void prune_move_kernel(RowArrays rows, const int32_t *__restrict__ pos,
                                                         const int32_t *__restrict__ counts,
                                                         float *__restrict__ scratch) {
    const int new_n = counts[0], old_n = counts[1];
    if (new_n == old_n || new_n == 0) return;
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= old_n) return;
    const int p = pos[g];
    if (p < 0) return;
    const int stride = rows.offset[rows.count];
    float *dst = scratch + (size_t)p * stride;
    for (int a = 0; a < rows.count; ++a) {
        const int w = rows.width[a];
        const float *src = rows.ptr[a] + (size_t)g * w;
        for (int q = 0; q < w; ++q) dst[rows.offset[a] + q] = src[q];
    }
}
__global__ __launch_bounds__(256) void prune_copyback_kernel(RowArrays rows, const int32_t *__restrict__ counts,
                                                             const float *__restrict__ scratch,
                                                             int32_t *__restrict__ n_dev,
                                                             int32_t *__restrict__ pruned_total) {
    const int new_n = counts[0], old_n = counts[1];
    if (new_n == old_n || new_n == 0) return;
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g == 0) {
        *n_dev = new_n;
        if (pruned_total) *pruned_total += old_n - new_n;
    }
    if (g >= new_n) return;
    const int stride = rows.offset[rows.count];
    const float *src = scratch + (size_t)g * stride;
    for (int a = 0; a < rows.count; ++a) {
        const int w = rows.width[a];
        float *dst = rows.ptr[a] + (size_t)g * w;
        for (int q = 0; q < w; ++q) dst[q] = src[rows.offset[a] + q];
    }
}

// ------------------------------------------------------------------------------------------------- grow
// train.py:86: errors = |render - gt| summed over the channels (render = clamp(out_img, 0, 1)); keys = the float bits
// (non-negative floats order like their bit patterns).
__global__ __launch_bounds__(256) void grow_error_kernel(int npix, const float *__restrict__ out_img,
                                                         const float *__restrict__ gt, uint32_t *__restrict__ key,
                                                         int32_t *__restrict__ sel_state, int state_words) {
#pragma clang fp contract(off)
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < state_words) sel_state[p] = 0;  // the select's ticket, histogram and counts (grow_hist_kernel)
    if (p >= npix) return;
    float e = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float r = fminf(fmaxf(out_img[3 * (size_t)p + c], 0.f), 1.f);
        e = e + fabsf(r - gt[3 * (size_t)p + c]);
    }
    key[p] = e == e ? __float_as_uint(e) : 0u;  // a NaN error never wins
}

// The k = growth budget largest keys, as pixel indices in ascending order: a radix select of the k-th largest key (four
// 8-bit digits), then every key above that threshold and the first `want` keys equal to it.  One workgroup reading every
// key six times was 0.47 ms per growth step and image (nine steps per fit); now GI2D_SEL_WGS workgroups of 1024 lanes
// each take a contiguous run of pixels, in six launches:
//   grow_hist_kernel<24 / 16 / 8 / 0>  LDS histogram of the digit over the workgroup's run (keys that match the
//                                      threshold's prefix so far), added to the pass's global histogram
//   grow_count_kernel                  per (workgroup, wave): keys above / equal to the threshold in its run
//   grow_write_kernel                  every workgroup scans those 1024 pairs itself and writes its run's share
// Nobody combines the histograms in a step of its own: every workgroup of the NEXT launch reads the finished histogram
// (256 words, one kernel boundary behind it) and picks the digit itself -- a suffix sum and a ballot, the same answer
// everywhere; workgroup 0 also leaves the threshold so far in the state words for the launch after.  (A "last
// workgroup in combines" ticket needs a release fence per workgroup, and on this 8-XCD part that is an L2 write-back:
// 21 ... 25 us per pass against 5 for reading the keys.)
// state words (`sel_state`, zeroed by grow_error_kernel): [0..3) and [4..7) two slots of (prefix, mask, want) -- the
// threshold as resolved by the last launch, slots alternating; [8 + 256 p ..) histogram of pass p; [8 + 1024 ..) the
// 2 x 16 x GI2D_SEL_WGS counts.
#define GI2D_SEL_WGS 64
#define GI2D_SEL_STATE_WORDS (8 + 4 * 256 + 2 * 16 * GI2D_SEL_WGS)
__device__ __forceinline__ int grow_budget(const int32_t *__restrict__ n_dev, int n_bound, int max_points, int budget_cap,
                                           int kmax, int npix, int &n) {
    n = min(*n_dev, n_bound);
    const int k = max(0, min(budget_cap, max_points - n));  // train.py:91-97
    return min(min(k, kmax), npix);
}
// pixels [w0, w1) of wave `wave` of workgroup `b`: runs ascend with (b, wave), 64-aligned
__device__ __forceinline__ void sel_run(int npix, int b, int wave, int &w0, int &w1) {
    const int per_wave = (((npix + 16 * GI2D_SEL_WGS - 1) / (16 * GI2D_SEL_WGS)) + 63) & ~63;
    w0 = min(npix, (b * 16 + wave) * per_wave);
    w1 = min(npix, w0 + per_wave);
}
// The digit at `shift` of the threshold, from the finished histogram `gh` of the keys that match (prefix, mask): the
// largest d >= 1 with (keys of digit >= d) >= want, else 0 -- walking the bins from the top until `want` keys are
// covered -- as a suffix sum over the 256 bins by the first four waves (lane i holds bin 255 - i).  Whole workgroup
// (>= 256 lanes); on return prefix / mask include the digit and want is the rank still wanted among the keys equal so far.
__device__ __forceinline__ void resolve_digit(const int32_t *__restrict__ gh, int shift, unsigned &prefix, unsigned &mask,
                                              int &want) {
    __shared__ int suffix[257];
    __shared__ int wtot[4], first_hit[4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int incl = 0;
    if (tid < 256) {
        incl = wave_inclusive_scan(gh[255 - tid]);
        if (lane == 63) wtot[wave] = incl;
    }
    if (tid == 0) suffix[256] = 0;
    __syncthreads();
    if (tid < 256) {
        int before = 0;
        for (int w = 0; w < wave; ++w) before += wtot[w];
        const int sfx = before + incl;  // keys whose digit is >= 255 - tid
        suffix[255 - tid] = sfx;
        const unsigned long long hit = __ballot(sfx >= want && tid < 255);
        if (lane == 0) first_hit[wave] = hit ? wave * 64 + (__ffsll((long long)hit) - 1) : 256;
    }
    __syncthreads();
    const int i = min(min(first_hit[0], first_hit[1]), min(first_hit[2], first_hit[3]));
    const int d = i < 256 ? 255 - i : 0;
    prefix |= (unsigned)d << shift;
    mask |= 255u << shift;
    want -= suffix[d + 1];
    __syncthreads();  // the arrays above are reused by the next call
}

template <int SHIFT>
__global__ __launch_bounds__(1024) void grow_hist_kernel(int npix, const uint32_t *__restrict__ key,
                                                         const int32_t *__restrict__ n_dev, int n_bound, int max_points,
                                                         int budget_cap, int kmax, int32_t *__restrict__ state,
                                                         int32_t *__restrict__ info) {
    __shared__ int hist[256];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int n;
    const int k = grow_budget(n_dev, n_bound, max_points, budget_cap, kmax, npix, n);
    if (SHIFT == 24 && blockIdx.x == 0 && tid == 0) {
        info[0] = k;  // info = {k, live n at entry}
        info[1] = n;
    }
    if (k == 0) return;
    // the threshold so far: its digits above SHIFT, and the rank (from the top) still wanted among the keys matching them
    unsigned prefix = 0, mask = 0;
    int want = k;
    if (SHIFT < 24) {
        // two slots of (prefix, mask, want), read and written alternately: a launch never writes the slot its own
        // workgroups -- which start at different times -- are still reading
        const int32_t *in = state + (SHIFT == 8 ? 0 : 4);
        int32_t *out = state + (SHIFT == 8 ? 4 : 0);
        if (SHIFT < 16) prefix = (unsigned)in[0], mask = (unsigned)in[1], want = in[2];
        resolve_digit(state + 8 + 256 * ((24 - SHIFT) / 8 - 1), SHIFT + 8, prefix, mask, want);
        if (blockIdx.x == 0 && tid == 0) out[0] = (int32_t)prefix, out[1] = (int32_t)mask, out[2] = want;
    }
    int32_t *ghist = state + 8 + 256 * ((24 - SHIFT) / 8);
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    int w0, w1;
    sel_run(npix, (int)blockIdx.x, wave, w0, w1);
    for (int base = w0; base < w1; base += 64 * 8) {  // wave-uniform trip count (ballots below); eight loads in flight
        unsigned v8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int p = base + 64 * u + lane;
            v8[u] = p < w1 ? key[p] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned v = v8[u];
            const bool act = base + 64 * u + lane < w1 && (v & mask) == prefix;
            const unsigned d = (v >> SHIFT) & 255u;
            if (SHIFT == 24) {
                // the top byte of an error in [0, 3] takes a handful of values: one LDS atomic per distinct value and
                // wave instead of one per key (64 lanes on one address serialise)
                unsigned long long todo = __ballot(act);
                while (todo) {
                    const int src = __ffsll((long long)todo) - 1;
                    const unsigned d0 = (unsigned)__shfl((int)d, src, 64);
                    const unsigned long long same = __ballot(act && d == d0);
                    if (lane == src) atomicAdd(&hist[d0], __popcll(same));
                    todo &= ~same;
                }
            } else if (act) {
                atomicAdd(&hist[d], 1);
            }
        }
    }
    __syncthreads();
    if (tid < 256 && hist[tid]) atomicAdd(&ghist[tid], hist[tid]);
}

__global__ __launch_bounds__(1024) void grow_count_kernel(int npix, const uint32_t *__restrict__ key,
                                                          const int32_t *__restrict__ info,
                                                          int32_t *__restrict__ state) {
    if (info[0] == 0) return;
    unsigned prefix = (unsigned)state[0], mask = (unsigned)state[1];  // slot 0: left by grow_hist_kernel<0>
    int want = state[2];
    resolve_digit(state + 8 + 256 * 3, 0, prefix, mask, want);
    if (blockIdx.x == 0 && threadIdx.x == 0)  // slot 1, for grow_write_kernel
        state[4] = (int32_t)prefix, state[5] = (int32_t)mask, state[6] = want;
    const unsigned thr = prefix;  // the k-th largest key
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int w0, w1;
    sel_run(npix, (int)blockIdx.x, wave, w0, w1);
    int above = 0, equal = 0;  // wave-uniform
    for (int base = w0; base < w1; base += 64 * 8) {
        unsigned v8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int p = base + 64 * u + lane;
            v8[u] = p < w1 ? key[p] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool in = base + 64 * u + lane < w1;
            above += __popcll(__ballot(in && v8[u] > thr));
            equal += __popcll(__ballot(in && v8[u] == thr));
        }
    }
    if (lane == 0) {
        int32_t *cnt = state + 8 + 4 * 256 + 2 * ((int)blockIdx.x * 16 + wave);
        cnt[0] = above;
        cnt[1] = equal;
    }
}

// The selected pixels in ascending index order: a selected pixel's position is everything selected in front of it -- the
// runs before its wave's (a scan of the 16 x GI2D_SEL_WGS count pairs, done by every workgroup for itself) and, inside a
// 64-key group, a ballot and a popcount.  `want` of the keys equal to the threshold are taken, lowest index first.
__global__ __launch_bounds__(1024) void grow_write_kernel(int npix, const uint32_t *__restrict__ key,
                                                          const int32_t *__restrict__ info,
                                                          const int32_t *__restrict__ state, int32_t *__restrict__ sel) {
    static_assert(16 * GI2D_SEL_WGS == 1024, "one count pair per lane");
    __shared__ int wsum[17];
    __shared__ int a_base[16], e_base[16];
    if (info[0] == 0) return;
    const unsigned thr = (unsigned)state[4];  // slot 1: left by grow_count_kernel
    const int want = state[6];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int32_t *cnt = state + 8 + 4 * 256;
    int total;
    const int a_ex = block_exclusive_scan(cnt[2 * tid], wsum, total);
    const int e_ex = block_exclusive_scan(cnt[2 * tid + 1], wsum, total);
    if ((tid >> 4) == (int)blockIdx.x) a_base[tid & 15] = a_ex, e_base[tid & 15] = e_ex;
    __syncthreads();
    int a_before = a_base[wave], e_before = e_base[wave];
    int w0, w1;
    sel_run(npix, (int)blockIdx.x, wave, w0, w1);
    const unsigned long long lt = lanemask_lt();
    for (int base = w0; base < w1; base += 64 * 8) {
        unsigned v8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int p = base + 64 * u + lane;
            v8[u] = p < w1 ? key[p] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int p = base + 64 * u + lane;
            const bool in = p < w1, is_a = in && v8[u] > thr, is_e = in && v8[u] == thr;
            const unsigned long long ba = __ballot(is_a), be = __ballot(is_e);
            const int a_cnt = a_before + __popcll(ba & lt), e_cnt = e_before + __popcll(be & lt);
            if (is_a) sel[a_cnt + min(e_cnt, want)] = p;
            if (is_e && e_cnt < want) sel[a_cnt + e_cnt] = p;
            a_before += __popcll(ba);
            e_before += __popcll(be);
        }
    }
}

// rank of every selected pixel in (error descending, pixel index ascending) order -- torch.topk's sorted output
// with a stable tie rule -- by counting: k is a few thousand at most times in a fit, k^2 compares are microseconds
__global__ __launch_bounds__(256) void grow_rank_kernel(const uint32_t *__restrict__ key,
                                                        const int32_t *__restrict__ sel,
                                                        const int32_t *__restrict__ info,
                                                        int32_t *__restrict__ ordered) {
    __shared__ unsigned tk[256];
    __shared__ int ti[256];
    const int k = info[0];
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x * blockDim.x >= k) return;
    const int pj = j < k ? sel[j] : 0;
    const unsigned kj = j < k ? key[pj] : 0u;
    int rank = 0;
    for (int base = 0; base < k; base += 256) {
        const int i = base + threadIdx.x;
        __syncthreads();
        if (i < k) {
            ti[threadIdx.x] = sel[i];
            tk[threadIdx.x] = key[sel[i]];
        }
        __syncthreads();
        const int m = min(256, k - base);
        for (int q = 0; q < m; ++q) rank += (tk[q] > kj || (tk[q] == kj && ti[q] < pj)) ? 1 : 0;
    }
    if (j < k) ordered[rank] = pj;
}

// One workgroup of 1024 lanes: covariance draws, drop the non-definite ones in order, append the survivors.
__global__ __launch_bounds__(1024) void grow_append_kernel(RowArrays rows, int img_w, int img_h,
                                                           const int32_t *__restrict__ ordered,
                                                           const int32_t *__restrict__ info,
                                                           const float *__restrict__ rand3, float *xyz, float *chol,
                                                           float *feat, float *opacity, float *bound, int bound_stride,
                                                           int32_t *__restrict__ n_dev, int32_t *__restrict__ added) {
#pragma clang fp contract(off)
    __shared__ int wsum[17];
    const int k = info[0], n0 = info[1];
    int run = 0;
    for (int base = 0; base < k; base += 1024) {
        const int r = base + threadIdx.x;
        float c0 = 0.f, c1 = 0.f, c2 = 0.f;
        int keep = 0, pix = 0;
        if (r < k) {
            pix = ordered[r];
            c0 = rand3[3 * r] + 0.5f, c1 = rand3[3 * r + 1] + 0.f, c2 = rand3[3 * r + 2] + 0.5f;  // train.py:109-111
            keep = positive_definite(c0, c1, c2) ? 1 : 0;  // densification_postfix: check_non_semi_definite(new_cov2d)
        }
        int total;
        const int ex = block_exclusive_scan(keep, wsum, total);
        if (keep) {
            const size_t g = (size_t)n0 + run + ex;
            for (int a = 0; a < rows.count; ++a) {  // optimizer moments (and everything else) of the new row: zero
                float *dst = rows.ptr[a] + g * rows.width[a];
                for (int q = 0; q < rows.width[a]; ++q) dst[q] = 0.f;
            }
            xyz[2 * g] = (float)(pix % img_w);
            xyz[2 * g + 1] = (float)(pix / img_w);
            chol[3 * g] = c0, chol[3 * g + 1] = c1, chol[3 * g + 2] = c2;
            feat[3 * g] = feat[3 * g + 1] = feat[3 * g + 2] = 0.f;
            opacity[g] = 1.f;
        }
        run += total;
    }
    __syncthreads();
    const int n1 = n0 + run;
    if (bound_stride) {  // SLV: the new rows get the low-pass bound of the new population size (:343-348)
        const float low = (float)fmin((double)img_h * (double)img_w / (9.0 * 3.141592653589793 * (double)n1), 300.0);
        for (int g = n0 + threadIdx.x; g < n1; g += 1024) {
            bound[3 * (size_t)g] = low;
            bound[3 * (size_t)g + 1] = 0.f;
            bound[3 * (size_t)g + 2] = low;
        }
    }
    if (threadIdx.x == 0) {
        *n_dev = n1;
        if (added) *added += run;
    }
}

static RowArrays rows_of(const gi2d_train_state *s) {
    RowArrays r;
    r.count = 0;
    auto add = [&](float *p, int w) {
        if (!p) return;
        r.ptr[r.count] = p;
        r.width[r.count] = w;
        ++r.count;
    };
    add(s->xyz, 2), add(s->chol, 3), add(s->feat, 3), add((float *)s->opacity, 1);
    add(s->m_xyz, 2), add(s->v_xyz, 2), add(s->m_chol, 3), add(s->v_chol, 3), add(s->m_feat, 3), add(s->v_feat, 3);
    if (s->optimizer == 1) {
        add(s->d_xyz, 2), add(s->d_chol, 3), add(s->d_feat, 3), add(s->pg_xyz, 2), add(s->pg_chol, 3),
            add(s->pg_feat, 3);
    }
    if (s->bound_stride == 3) add((float *)s->bound, 3);
    r.offset[0] = 0;
    for (int a = 0; a < r.count; ++a) r.offset[a + 1] = r.offset[a] + r.width[a];
    return r;
}

}  // namespace gi2d

using namespace gi2d;

extern "C" {

size_t gi2d_densify_scratch_bytes(const gi2d_train_state *s, int max_points) {
    if (!s || max_points < 0) return 0;
    const RowArrays r = rows_of(s);
    const size_t npix = (size_t)s->img_height * (size_t)s->img_width;
    const size_t cap = (size_t)(max_points > s->num_points ? max_points : s->num_points);
    // pos[cap] + counts[8] | keys[npix] + sel[cap] + ordered[cap] + info[8] + select state | row scratch
    return 256 * 9 + sizeof(int32_t) * (3 * cap + 16 + GI2D_SEL_STATE_WORDS) + sizeof(uint32_t) * npix +
           sizeof(float) * cap * r.offset[r.count];
}

struct DensifyWs {
    int32_t *pos, *counts, *sel, *ordered, *info, *sel_state;
    uint32_t *key;
    float *rowbuf;
};
static DensifyWs carve_densify(void *base, size_t cap, size_t npix) {
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    char *b = (char *)base;
    size_t off = 0;
    DensifyWs w;
    w.counts = (int32_t *)(b + off), off += up(8 * sizeof(int32_t));
    w.info = (int32_t *)(b + off), off += up(8 * sizeof(int32_t));
    w.sel_state = (int32_t *)(b + off), off += up(GI2D_SEL_STATE_WORDS * sizeof(int32_t));
    w.pos = (int32_t *)(b + off), off += up(cap * sizeof(int32_t));
    w.sel = (int32_t *)(b + off), off += up(cap * sizeof(int32_t));
    w.ordered = (int32_t *)(b + off), off += up(cap * sizeof(int32_t));
    w.key = (uint32_t *)(b + off), off += up(npix * sizeof(uint32_t));
    w.rowbuf = (float *)(b + off);
    return w;
}

static int densify_check(const gi2d_train_state *s, void *scratch, size_t scratch_bytes, int max_points) {
    if (!s || !s->num_points_dev || !scratch) {
        set_error("densify: needs a train state with num_points_dev (device-resident population) and scratch");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    if (s->kind != 1) {
        set_error("densify: pruning / growth are the covariance model's (train.py: adaptive_add / prune are switched "
                  "off for the other models)");
        return GI2D_ERR_UNSUPPORTED;
    }
    if (scratch_bytes < gi2d_densify_scratch_bytes(s, max_points)) {
        set_error("densify: scratch too small");
        return GI2D_ERR_WORKSPACE_TOO_SMALL;
    }
    return GI2D_OK;
}

int gi2d_train_prune(const gi2d_train_state *s, void *scratch, size_t scratch_bytes, int32_t *pruned_total,
                     gi2d_stream_t st_) {
    int rc = densify_check(s, scratch, scratch_bytes, s ? s->num_points : 0);
    if (rc != GI2D_OK) return rc;
    const int n = s->num_points;
    if (n == 0) return GI2D_OK;
    hipStream_t st = (hipStream_t)st_;
    const RowArrays rows = rows_of(s);
    const DensifyWs w = carve_densify(scratch, (size_t)n, (size_t)s->img_height * s->img_width);
    hipLaunchKernelGGL(prune_scan_kernel, dim3(1), dim3(1024), 0, st, (const int32_t *)s->num_points_dev, n,
                       (const float *)s->chol, s->bound, s->bound_stride, w.pos, w.counts);
    const dim3 grid((n + 255) / 256), block(256);
    hipLaunchKernelGGL(prune_move_kernel, grid, block, 0, st, rows, (const int32_t *)w.pos, (const int32_t *)w.counts,
                       w.rowbuf);
    hipLaunchKernelGGL(prune_copyback_kernel, grid, block, 0, st, rows, (const int32_t *)w.counts,
                       (const float *)w.rowbuf, s->num_points_dev, pruned_total);
    rc = check_launch("train prune");
    if (rc != GI2D_OK) return rc;
    // rows that moved were renumbered: the workspace's persistent tile lists (they hold gaussian ids) start over --
    // decided on the device, like the move itself; a check that drops nothing (nearly all of them) leaves the lists alone
    if (s->workspace)
        rc = launch_workspace_init(s->workspace, n, (s->img_width + 15) / 16, (s->img_height + 15) / 16,
                                   (const int32_t *)w.counts, st_);
    return rc;
}

int gi2d_train_grow(const gi2d_train_state *s, int max_points, int budget_cap, const float *rand3, int rand_rows,
                    void *scratch, size_t scratch_bytes, int32_t *added, gi2d_stream_t st_) {
    int rc = densify_check(s, scratch, scratch_bytes, max_points);
    if (rc != GI2D_OK) return rc;
    if (max_points < 0 || budget_cap < 0 || rand_rows < 0 || (rand_rows > 0 && !rand3)) {
        set_error("train grow: bad argument");
        return GI2D_ERR_INVALID_ARGUMENT;
    }
    const int kmax = budget_cap < rand_rows ? budget_cap : rand_rows;
    if (kmax == 0) return GI2D_OK;
    hipStream_t st = (hipStream_t)st_;
    const int npix = s->img_height * s->img_width;
    const size_t cap = (size_t)(max_points > s->num_points ? max_points : s->num_points);
    const DensifyWs w = carve_densify(scratch, cap, (size_t)npix);
    RowArrays moments = rows_of(s);  // every array of a new row starts from zero; parameters are then written over it
    const int err_threads = npix > GI2D_SEL_STATE_WORDS ? npix : GI2D_SEL_STATE_WORDS;
    hipLaunchKernelGGL(grow_error_kernel, dim3((err_threads + 255) / 256), dim3(256), 0, st, npix,
                       (const float *)s->out_img, s->gt, w.key, w.sel_state, GI2D_SEL_STATE_WORDS);
    const dim3 sg(GI2D_SEL_WGS), sb(1024);
    const uint32_t *key = (const uint32_t *)w.key;
    const int32_t *n_dev = (const int32_t *)s->num_points_dev;
    hipLaunchKernelGGL(grow_hist_kernel<24>, sg, sb, 0, st, npix, key, n_dev, s->num_points, max_points, budget_cap, kmax,
                       w.sel_state, w.info);
    hipLaunchKernelGGL(grow_hist_kernel<16>, sg, sb, 0, st, npix, key, n_dev, s->num_points, max_points, budget_cap, kmax,
                       w.sel_state, w.info);
    hipLaunchKernelGGL(grow_hist_kernel<8>, sg, sb, 0, st, npix, key, n_dev, s->num_points, max_points, budget_cap, kmax,
                       w.sel_state, w.info);
    hipLaunchKernelGGL(grow_hist_kernel<0>, sg, sb, 0, st, npix, key, n_dev, s->num_points, max_points, budget_cap, kmax,
                       w.sel_state, w.info);
    hipLaunchKernelGGL(grow_count_kernel, sg, sb, 0, st, npix, key, (const int32_t *)w.info, w.sel_state);
    hipLaunchKernelGGL(grow_write_kernel, sg, sb, 0, st, npix, key, (const int32_t *)w.info,
                       (const int32_t *)w.sel_state, w.sel);
    hipLaunchKernelGGL(grow_rank_kernel, dim3((kmax + 255) / 256), dim3(256), 0, st, (const uint32_t *)w.key,
                       (const int32_t *)w.sel, (const int32_t *)w.info, w.ordered);
    hipLaunchKernelGGL(grow_append_kernel, dim3(1), dim3(1024), 0, st, moments, s->img_width, s->img_height,
                       (const int32_t *)w.ordered, (const int32_t *)w.info, rand3, s->xyz, s->chol, s->feat,
                       (float *)s->opacity, (float *)s->bound, s->bound_stride, s->num_points_dev, added);
    return check_launch("train grow");
}

}  // extern "C"
