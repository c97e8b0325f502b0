// Several images per launch (BASELINE config 3 is a 24-image batch; train.py:294-308 fits the images one after the
// other): per-image argument blocks of the tile pass and of the fitting kernels, shared by the single-image kernels
// (which take one block by value as their kernel argument) and the batched ones (which pick theirs from a table in HBM
// by workgroup index).  Why batches: one 768x512 image is 1536 tile workgroups -- exactly one residency round of the
// chip (6 per CU x 256 CUs), all started in the same cycle and walking the same phases in lockstep, so the
// latency-bound head of the tile pass is fully exposed and its compute phases contend for the VALUs; and the
// per-gaussian kernels of one image are less than one wave per SIMD of pure dependent latency.  K images in one launch
// are K residency rounds whose workgroups overlap each other's phases.
#pragma once
#include "gi2d_fast_internal.h"

namespace gi2d {

struct TrainParams {
    // model state (updated in place)
    float *xyz;       // [N,2] raw (pre-tanh) for kind 0, pixel coordinates for kind 1
    float *chol;      // [N,3] raw cholesky (kind 0) / covariance (kind 1), before the additive bound
    float *feat;      // [N,3] colours
    const float *opacity;  // [N]   (a buffer of ones in the reference models; not optimised)
    const float *bound;    // [3] or [N,3]: additive bound (cholesky_bound / cov bound)
    int bound_stride;      // 0 or 3
    const int32_t *n_dev;  // live population on the device, or null (gi2d_train_state::num_points_dev)
    // Adam / Adan state: first moment, second moment; Adan only: moment of the gradient difference, previous gradient
    float *m_xyz, *v_xyz, *m_chol, *v_chol, *m_feat, *v_feat;
    float *d_xyz, *d_chol, *d_feat, *pg_xyz, *pg_chol, *pg_feat;
};

// Best-model snapshot kept on the device (train.py:133-139 deep-copies the state dict on the host whenever the
// PSNR improves): best_sse[2] ping-pongs between steps so every workgroup of a launch reads the same value.
struct BestSnap {
    float *xyz, *chol, *feat, *bound;  // [N,2], [N,3], [N,3], [N,3] or null (bound_stride 0)
    float *sse;                        // [2]
    int32_t *info;                     // [2]: num_points, step of the snapshot
    const float *tile_sse;
    int num_tiles, step;
};

// What the update kernel needs to start the NEXT iteration itself (FILL_NEXT): the freshly updated parameters
// are still in registers, so their activation + projection + bucket fill ride along and the next iteration
// begins with its tile pass -- one launch and one parameter round trip less per iteration.
struct NextFill {
    float clip_coe;
    int32_t *num_tiles_hit, *lists, *status, *tile_order;
    PrevBox *prev_box;
    RecSets recs;
    float4 *inbox;  // the tiles' inboxes (gi2d_train_state::inbox; nullptr: none)
};

struct AdamStep {
    float step_size;       // lr / (1 - beta1^t)
    float bc2_sqrt;        // sqrt(1 - beta2^t)            (Adan: sqrt(1 - beta3^t))
    float one_minus_b1, b2, one_minus_b2, eps;
    // Adan only
    float b1, b3, one_minus_b3, step_size_diff;  // lr * beta2 / (1 - beta2^t)
    int first;                                     // Adam step count == 1: the previous gradient is this one
};

// One image's arguments of the single-pass tile kernel (gi2d_fused_core.h::fused_tile).
struct TilePassArgs {
    int tiles_x, tiles_y, img_w, img_h;
    RecSets rs;
    int32_t *lists;
    int2 *tile_bins;
    float4 *partial_g, *partial_big;
    int32_t *status;
    float *out_img;
    const float *vsrc;  // MODE 0: gradient image; MODE 1: target image
    float grad_scale;
    float *tile_sse;
    const int32_t *tile_order;
    int32_t *big_tile;  // two-phase tile pass: which tiles the small form left to the general one (FastWs::big_tile)
    float4 *inbox;      // the tiles' inboxes (gi2d_train_state::inbox; nullptr: none)
    int write_through;  // single-image launches of one residency round: rows and image leave as sc1 stores (store16)
};

// One image's arguments of the per-gaussian fitting kernels (project+fill, reduce+update).
struct UpdateArgs {
    int n;  // gaussians (an upper bound when P.n_dev is set)
    TrainParams P;
    float2 *xys;
    int32_t *radii;
    float *conics;
    int tiles_x, tiles_y;
    float radius_clip;
    const int32_t *gids_sorted;
    const int2 *tile_bins;
    const float4 *partial_g, *partial_big;
    float img_w, img_h;
    float *dbg_grads;
    BestSnap best;  // .step is set per launch
    NextFill next;  // also what the stand-alone project+fill kernel bins into (lists, prev_box, recs, status)
};

static inline TilePassArgs tile_pass_args(const FastWs &w, int n, int tiles_x, int tiles_y, int img_w, int img_h,
                                          int32_t *status, float *out_img, const float *vsrc, float grad_scale,
                                          float *tile_sse) {
    TilePassArgs a;
    a.tiles_x = tiles_x, a.tiles_y = tiles_y, a.img_w = img_w, a.img_h = img_h;
    a.rs = rec_sets(w, n);
    a.lists = w.lists;
    a.inbox = w.inbox_recs;
    a.write_through = 0;
    a.tile_bins = (int2 *)w.tile_bins;
    a.partial_g = w.partial_g;
    a.partial_big = w.partial_big;
    a.status = status;
    a.out_img = out_img;
    a.vsrc = vsrc;
    a.grad_scale = grad_scale;
    a.tile_sse = tile_sse;
    a.tile_order = w.tile_order;
    a.big_tile = w.big_tile;
    return a;
}

// Quantisation-aware iterations (gi2d_train.hip): the quantisers' state of one image.
struct QuantTrain {
    float qmax_xy, qmax_cov, qmax_col;  // unsigned quantisers: qmin = 0
    float qmin_rot, qmax_rot;           // rotation-scale model: the SIGNED rotation quantiser
    float *qparams;                     // [12] xy scale[2], xy beta[2], cov scale, cov beta, colour scale[3], colour beta[3]
    float *qm, *qv;                     // [12] Adam moments of qparams
    float *range;                       // [4] min log, max log, #elements at the min, #at the max (variance channels)
    float *qfeat;                       // [N,3] dequantised colours
    float *partial;                     // [blocks][GI2D_QT_ROW]
    int32_t *defer;                     // [8 + 8*defer_cap]: count, then 32-byte entries (flat index into chol,
                                        // gradient, parameter, Adam moments, bound)
    int defer_cap;
    float *best_q, *dbg_q;              // [12] snapshot of qparams / [16] gradients (tests), or null
};
#define GI2D_QT_ROW 24  // 12 LSQ sums, 2 log sums, next range (min, #min, max, #max), padding

struct BatchImage {
    TilePassArgs t;
    UpdateArgs u;
    QuantTrain q;  // quantisation-aware batches only
};

#define GI2D_BATCH_MAX 64 /* images per launch: one ballot finds a workgroup's image */
// Device layout of a batch table: [0, 65) tile-pass workgroup index at which image k starts (entry K = total),
// [128, 193) the same for the per-gaussian kernels, then the K argument blocks.
struct BatchHead {
    int tile_start[96];  // GI2D_BATCH_MAX + 1 used
    // How many tiles of the batch had a row above GI2D_SMALL_CAP candidates in the last tile pass of the current call
    // (counted from the passes' marks when the call ends; the table is rewritten, and this word cleared, when a call
    // starts): read back behind the call's kernels, it tells the NEXT call on this table whether the two-phase tile
    // pass pays (gi2d_fast.hip::batch_pass_begin).  A hint: never a correctness input.
    int big_seen;
    int reserved[31];
    int pg_start[128];
};
struct BatchTable {
    BatchHead *head;
    BatchImage *img;
    size_t bytes;
};
static inline BatchTable carve_batch(void *base, int k) {
    BatchTable b;
    b.head = (BatchHead *)base;
    b.img = (BatchImage *)((char *)base + align_up(sizeof(BatchHead)));
    b.bytes = align_up(sizeof(BatchHead)) + align_up((size_t)(k > 0 ? k : 1) * sizeof(BatchImage));
    return b;
}
// Which image does workgroup `block` belong to?  starts[0..K]: ascending, starts[K] = number of workgroups.
// Wave-uniform result; every wave of the workgroup computes it for itself (one 65-word load, one ballot).
__device__ __forceinline__ int batch_find(const int *__restrict__ starts, int k_images, int block) {
    const int lane = threadIdx.x & 63;
    const int v = lane < k_images ? starts[lane] : 0x7fffffff;
    const unsigned long long m = __ballot(v <= block);
    return __builtin_amdgcn_readfirstlane(__popcll(m) - 1);
}

// gi2d_train.hip: write `imgs` (host) and `head` into the table with kernels that carry them as kernel arguments --
// stream-ordered, no host buffer whose lifetime anybody has to think about, capturable in a graph.
void write_batch_table(const BatchTable &table, const BatchImage *imgs, int k_images, const BatchHead &head,
                       hipStream_t st);
// gi2d_fast.hip: the batched single-pass tile kernel (MODE 1: L2-loss gradient against t.vsrc = target) over
// `total_blocks` = head->tile_start[K] workgroups; uniform_tiles > 0: every image has that many tiles.
// `form` (batch_pass_begin says which): 0 one launch of the general form; 1 / 2 the small form on every tile it can serve,
// then the general form on the rest -- 2: the last pass that reported found such tiles (the second launch is shaped for work).
int launch_tile_pass_batched(int mode, const BatchTable &b, int k_images, int total_blocks, int uniform_tiles,
                             int form, hipStream_t st);
// Bracket of one C-ABI call's batched tile passes on table `batch`: _begin answers "which form?" (as above) from what the
// previous call on the same table reported (without waiting for anything), _end queues the read-back of this call's
// report.  (The same bracket for a single image of more than one residency round of tiles, keyed by its workspace;
// _begin is 0 for smaller images, which never run as two launches.)
int single_pass_begin(const void *ws, long long tiles, hipStream_t st);
void single_pass_end(const void *ws, const FastWs &w, long long tiles, hipStream_t st);
// gi2d_fast_rasterize_forward_backward with the form given: 1 / 2 two launches, 0 one, -1 the C entry's own rule
int fast_forward_backward_form(int n, int tiles_x, int tiles_y, unsigned img_w, unsigned img_h, const float *background,
                               const float *v_output, const float *target, float grad_scale, float *tile_sse, void *ws,
                               size_t ws_bytes, int32_t *status, float *out_img, gi2d_stream_t st, int form,
                               float4 *inbox = nullptr);  // inbox: the tiles' inboxes to take entrants out of (fast_fwdbwd_kernel)
int batch_pass_begin(const void *batch, int total_blocks, hipStream_t st);
void batch_pass_end(const void *batch, const BatchTable &b, int k_images, int total_blocks, hipStream_t st);

}  // namespace gi2d
