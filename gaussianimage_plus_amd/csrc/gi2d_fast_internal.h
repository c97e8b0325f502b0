// Internals of the fused fast path shared by gi2d_fast.hip and gi2d_train.hip: workspace layout, bucket
// fill, partial-row addressing and the ordered per-gaussian reduction.
#pragma once
#include <atomic>
#include <type_traits>

#include "gi2d_project_core.h"
#include "gi2d_raster_core.h"

namespace gi2d {


// Tile lists of the fused fast path are PERSISTENT: row t of `lists` holds
//     { count, sorted_len, 14 words of padding | ids[GI2D_FAST_C] }        (the header is one 64-byte line)
// ids[0 .. sorted_len) are the tile's members as of the last tile pass, ascending; ids[sorted_len .. count) were
// appended since (any order).  `prev_box[g]` is the tile box gaussian g was binned with.  A binning step
// (fill_diff) compares each gaussian's new box with that box and appends the gaussian only to the tiles it has
// ENTERED (returning atomics on the row header): from one iteration of a fit to the next nearly every box is unchanged,
// so a step issues a few hundred atomics instead of one per (gaussian, tile) -- 111 k at N = 50 000, which cost
// ~6 us per step: the device-scope atomic rate, not their latency, was the bound (tools/ubench/atomic_scope.hip).  The
// tile pass tests every entry against the CURRENT box of its gaussian (exactly the membership rule of
// map_gaussian_to_intersects, forward.cu:161-166), drops those that left, merges the appended ones into the
// ascending order and writes the row back only where it changed.  Invariant between a binning step and the tile
// pass that follows it: row t holds exactly {g : t in prev_box[g]} plus entries of gaussians that have left t, no id
// twice.  Results are those of a from-scratch binning for ANY change of the inputs (a workspace that saw other inputs
// before only costs more appends); gi2d_fast_workspace_init empties all rows and boxes, and must be called when
// gaussian ids are renumbered (pruning / compaction) or after an overflow.
#define GI2D_FAST_C 1024                             /* list slots per tile */
#define GI2D_FAST_HDR 16                             /* header words in front of a row's ids */
#define GI2D_INBOX_WORDS 64                          /* bitmap of the tile's inbox behind its ids (below) */
#define GI2D_INBOX_SLOTS (32 * GI2D_INBOX_WORDS)     /* 8 neighbour directions x 256 ranks */
#define GI2D_INBOX_MAX_TILES 1536                    /* images of more tiles do without (Inbox below) */
#define GI2D_FAST_LROW (GI2D_FAST_C + GI2D_FAST_HDR + GI2D_INBOX_WORDS) /* words per row */
#define GI2D_FAST_EPT (GI2D_FAST_C / 256)            /* list entries per lane of the 256-lane tile workgroup */
#define GI2D_FAST_S 32                               /* gaussian-major partial rows per gaussian: a gaussian on <= 32
                                                        tiles finds its rows by position in its box.  16 left every
                                                        gaussian on 17..32 tiles (a few per cent of a TRAINED scene:
                                                        flat regions end up under gaussians 50..100 px wide) to a binary
                                                        search per tile in the tile's id list -- 8 dependent loads each,
                                                        32 searches per lane: the training update kernel of a Kodak fit
                                                        took 166 us per 24-image launch, 71 us with 32 rows */
#ifndef GI2D_FILL_BATCH
#define GI2D_FILL_BATCH 16                           /* row-header atomics a lane keeps in flight */
#endif
#ifndef GI2D_REDUCE_BATCH
#define GI2D_REDUCE_BATCH 8                          /* partial rows a lane loads before it starts adding */
#endif
#define GI2D_BIG_TILES_F 32
#ifndef GI2D_UPDATE_ROWS_AHEAD
#define GI2D_UPDATE_ROWS_AHEAD 2                     /* gradient rows the update kernel requests before it knows the box (reduce_one) */
#endif
#define GI2D_FAST_ROW 4 /* float4 per partial row: 48 bytes of data padded to one 64-byte line */

// What the binning step remembers per gaussian: (x, y) the packed tile box it is binned with (pack_box; 0/0 = none),
// z = first row of its run in the row pool (-1: none), w = rows of that run.  Only a gaussian on more than GI2D_FAST_S
// tiles owns a run: GI2D_FAST_S rows per gaussian are reserved in the gaussian-major buffer, anything larger (a per
// cent of a trained scene's gaussians, 50...100 px wide) gets box-many consecutive rows from the pool the first time its
// box needs more than it holds -- a bump allocator (one returning atomic on the cursor word, in the header of tile
// row 0; emptied with the workspace).  A run is addressed like the gaussian-major rows, by position in the box, so the
// per-gaussian sum reads consecutive rows instead of searching every tile's id list for its rank (8 dependent loads per
// tile: on a trained Kodak scene those searches were 23 of the end-of-step kernel's 38 us).  A pool that runs out hands
// out rows past its end: the tile pass sees the row index, raises the overflow status and stores nothing; the sum skips
// such a run; the caller falls back as for an overflowing tile row.
typedef int4 PrevBox;
#define GI2D_POOL_CURSOR 8 /* word of tile row 0's header (words 2..15 of a header are padding) */
#define GI2D_NO_ROW ((int)0x80000000) /* partial-row code of an entry whose pool row lies past the pool's end */
/* bits of the status words 1 (this pass) and 2 (sticky): 1 a tile row overflowed, 2 the log quantiser's parking list
   (gi2d_train.hip), 4 the row pool ran out */
#define GI2D_STATUS_POOL 4
__host__ __device__ __forceinline__ PrevBox no_box() { return make_int4(0, 0, -1, 0); }
static inline size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }
// THE INBOX OF A TILE -- how a gaussian enters a tile without waiting for an atomic's answer.  The update kernel of a
// fit ends with the binning step of the next iteration, and a gaussian that has ENTERED a tile used to reserve its slot
// in the tile's row with a returning atomic: a device-scope round trip at the very end of every lane's dependency
// chain, 1.2 us of a 10 us kernel, paid by the whole launch because a launch ends with its slowest
// wave and one wave in five holds a gaussian whose tile box changes.  A slot
// that needs no answer is one nobody else can want: a gaussian entering tile t is, one pass earlier, at a known rank
// r < 256 in the staged list of some NEIGHBOUR s of t that it is in (the tile pass hands every staged entry its rank in
// the spare word of the gradient row it writes anyway) -- slot (t, direction of s, r) is its alone.  It stores its
// 64-byte record there and sets the slot's bit in the tile's 2048-bit bitmap with a NON-returning atomicOr; the tile's
// workgroup -- the only writer of its row during a pass -- reads the bitmap with its row, the marked records with its
// entries' records, appends what it finds and clears the bits (tile_list_head).  Entrants with no staged neighbour
// (a new scene, a gaussian from afar, rank beyond the 256-entry cap) and boxes of more than eight tiles keep the
// returning atomic; binning kernels that do not know ranks (everything but the update kernel of a fit) likewise.
// Only where it pays: the update kernel of ONE image of at most GI2D_INBOX_MAX_TILES tiles -- the launches whose tile
// pass is the general form throughout.  A batch hides the round trip behind its other images' workgroups, and a larger
// image's tile pass runs the small form first (gi2d_fast.hip), which has no registers to spare for entrants.
// WHO MAY FIND AN INBOX NON-EMPTY: only the tile pass that gi2d_train_steps issues right behind such an update kernel,
// in the same call -- the update kernel only bins when another iteration follows in its call (FILL_NEXT), so every call
// returns with all inboxes empty, and no other kernel ever writes to one.  That pass (fast_fwdbwd_kernel<1, 0, true>,
// picked by the launch code like the update kernel's INBOX instantiation) is the only kernel built with the code that
// takes entrants in; nobody else loads a bitmap word.  The ranks are good for
// the same reason: the update kernel runs right behind the tile pass whose gradient rows it reduces.
struct Inbox {
    float4 *recs;  // [T][GI2D_INBOX_SLOTS][4]: a record as write_record leaves it, its last word the gaussian's id
};
// Access rule of this workspace's shared words, which the bitmap, the row headers and the status words obey:
//   * within ONE launch a WORD is touched either by atomics only or by plain accesses only -- the bitmap: atomicOr in the
//     update kernel, plain load + plain store of 0 (behind the head's barrier) in the tile pass, the two kinds of launch
//     alternating on one stream; a row header: atomicAdd in binning launches, plain in the tile pass; status[1..3]:
//     atomicOr / atomicMax in tile passes, plain reset by lane 0 of a binning launch;
//   * no word is reset, or read for a decision, by "whoever comes last" inside the launch that still bumps it: every
//     reset belongs to a LATER launch (or sits behind a workgroup barrier of the only workgroup that touches the word);
//   * DIFFERENT words of one 64-byte granule may mix the two kinds in one launch (an id stored plainly next to a bitmap
//     word that takes an atomicOr: rows are 69 granules long, the bitmap starts on a granule of its own but shares its
//     128-byte L2 line with the row's last ids) -- measured safe in both directions by tools/ubench/atomic_line_mix.hip.
static_assert((GI2D_FAST_LROW * 4) % 64 == 0 && ((GI2D_FAST_HDR + GI2D_FAST_C) * 4) % 64 == 0,
              "rows and their inbox bitmaps start on 64-byte granules");
__device__ __forceinline__ int32_t *inbox_bits_of(int32_t *lists, int tile) {
    return lists + (size_t)tile * GI2D_FAST_LROW + GI2D_FAST_HDR + GI2D_FAST_C;
}
// What an entering gaussian knows of its old tiles (the update kernel of a fit): the ranks it was staged at in the
// (<= 8) tiles of its old box.
struct InboxSrc {
    // rank + 1 as the tile pass left it in the gradient row of tile i of the old box (0: none); kept as loaded -- only the
    // few lanes whose box changes ever look at one.  (Eight named words, not an array: the compiler turns a select over
    // an array's elements into an indexed load, i.e. puts the array -- and the struct around it -- into scratch memory.)
    unsigned t0, t1, t2, t3, t4, t5, t6, t7;
    bool on;  // lanes that may use the inboxes at all
    __device__ __forceinline__ void clear() { t0 = t1 = t2 = t3 = t4 = t5 = t6 = t7 = 0u; }
    __device__ __forceinline__ void set(int q, unsigned v) {
        switch (q) {
            case 0: t0 = v; break;
            case 1: t1 = v; break;
            case 2: t2 = v; break;
            case 3: t3 = v; break;
            case 4: t4 = v; break;
            case 5: t5 = v; break;
            case 6: t6 = v; break;
            default: t7 = v; break;
        }
    }
    __device__ __forceinline__ unsigned get(int k) const {
        unsigned a0 = t0, a1 = t1, a2 = t2, a3 = t3, a4 = t4, a5 = t5, a6 = t6, a7 = t7;
        // (values, not loads: a select between loads of one object becomes a load at a selected address)
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        unsigned v = a0;
        v = k == 1 ? a1 : v;
        v = k == 2 ? a2 : v;
        v = k == 3 ? a3 : v;
        v = k == 4 ? a4 : v;
        v = k == 5 ? a5 : v;
        v = k == 6 ? a6 : v;
        v = k == 7 ? a7 : v;
        return v;
    }
};
// Workgroup size of the one-lane-per-gaussian kernels (project+fill, reduce+project backward, optimizer update):
// single waves spread a small population over many CUs and shorten the dependent load chains
// (N=2500: project+fill 10.9 -> 5.5 us, reduce 7.4 -> 4.7 us; neutral to slightly worse beyond ~30k gaussians).
#ifndef GI2D_PG_BIG
#define GI2D_PG_BIG 256
#endif
static inline int per_gaussian_block(int n) { return n <= 32768 ? 64 : GI2D_PG_BIG; }
struct FastWs {
    int32_t *lists;        // [T * LROW]         persistent tile lists (see above), each row with its inbox bitmap
    float4 *inbox_recs;    // [T * 2048 * 4]     the tiles' inboxes (Inbox): NOT part of the workspace -- a buffer of its own
                           //                    that only the caller of gi2d_train_steps on ONE image may bring
                           //                    (gi2d_train_state::inbox, gi2d_train_inbox_bytes); nullptr everywhere else
    int32_t *gids_sorted;  // == lists: tile_bins hold absolute word positions into it
    int32_t *tile_bins;    // [T * 2]            [row base + HDR, row base + HDR + len)
    GaussRec *packed;      // [T * 256]          tile-sorted records of the first 256 entries
    PrevBox *prev_box;     // [N]                tile box each gaussian was last binned with (packed, 0/0 = none) and
                           //                    its rows in the pool (gaussians on > GI2D_FAST_S tiles; see PrevBox)
    float4 *recs;          // [2][N * 4]         one 64-byte record per gaussian as of the last binning step: what a tile
                           //                    pass needs of it, in ONE cache line (see write_record); two sets, see RecSets
    int32_t *ver;          // [2]                which record set is current (RecSets)
    float4 *partial_g;     // [N * S * 4]        gaussian-major partial rows (64 B each)
    float4 *partial_big;   // [T * 256 * 4]      the row POOL: partial rows of gaussians on > S tiles, one run of rows
                           //                    per gaussian (its box in row-major order), allocated by the binning step
    int32_t *tile_order;   // [T]                tile handled by workgroup b of the single-pass tile kernel: a
                           //                    permutation that balances tile populations over the CUs
    int32_t *big_tile;     // [T]                two-phase tile pass (gi2d_fast.hip): 1 = the small form passed this tile
                           //                    over (its row holds more candidates than that form stages), 0 = done
    size_t bytes;
};
static FastWs carve_fast(void *base, int n, int num_tiles) {
    FastWs w;
    char *b = (char *)base;
    size_t off = 0;
    const size_t t = (size_t)(num_tiles > 0 ? num_tiles : 1), nn = (size_t)(n > 0 ? n : 1);
    // (the inboxes' records -- 128 KB of sparsely used address space per tile -- were part of every workspace in round 5:
    // batches, the drop-in wrappers' pooled workspaces, evaluation renders and every multi-GPU rank paid 192 MiB at
    // 768x512 for something only a single-image fit touches.  They are the fit's own buffer now: inbox_bytes below.)
    w.inbox_recs = nullptr;
    w.lists = (int32_t *)(b + off);
    w.gids_sorted = w.lists;
    off += align_up(t * GI2D_FAST_LROW * sizeof(int32_t));
    w.tile_bins = (int32_t *)(b + off);
    off += align_up(t * 2 * sizeof(int32_t));
    w.packed = (GaussRec *)(b + off);
    off += align_up(t * GI2D_TILE_LIST_CAP * sizeof(GaussRec));
    // lists, tile_order and prev_box carry state from call to call: they sit in front of every region whose offset
    // depends on the gaussian count, so a workspace initialised for a capacity can be used with any smaller population
    w.tile_order = (int32_t *)(b + off);
    off += align_up(t * sizeof(int32_t));
    w.big_tile = (int32_t *)(b + off);
    off += align_up(t * sizeof(int32_t));
    w.prev_box = (PrevBox *)(b + off);
    off += align_up(nn * sizeof(PrevBox));
    w.ver = (int32_t *)(b + off);
    off += align_up(64 * sizeof(int32_t));
    w.recs = (float4 *)(b + off);
    off += 2 * align_up(nn * 4 * sizeof(float4));
    w.partial_g = (float4 *)(b + off);
    off += align_up(nn * GI2D_FAST_S * GI2D_FAST_ROW * sizeof(float4));
    w.partial_big = (float4 *)(b + off);
    off += align_up(t * GI2D_TILE_LIST_CAP * GI2D_FAST_ROW * sizeof(float4));
    w.bytes = off;
    return w;
}
// May a launch on this workspace store write-through (gi2d_raster_core.h::store16)?  Its buffer stores address the
// gradient rows -- gaussian-major rows and the row pool behind them -- by a 32-bit offset from `partial_g`, the image and
// the packed records from their own bases: every one of those spans must stay below 4 GB (2 million gaussians' rows).
static inline bool wt_fits(const FastWs &w, int num_tiles, size_t image_bytes) {
    const size_t rows_span = (size_t)((const char *)w.partial_big - (const char *)w.partial_g) +
                             (size_t)(num_tiles > 0 ? num_tiles : 1) * GI2D_TILE_LIST_CAP * GI2D_FAST_ROW * sizeof(float4);
    const size_t packed_span = (size_t)(num_tiles > 0 ? num_tiles : 1) * GI2D_TILE_LIST_CAP * sizeof(GaussRec);
    return rows_span < 0xffff0000ull && packed_span < 0xffff0000ull && image_bytes < 0xffff0000ull;
}
// Bytes of the inbox buffer of an image of `num_tiles` tiles (0: such an image does without, Inbox above).
static inline size_t inbox_bytes(long long num_tiles) {
    if (num_tiles < 1 || num_tiles > GI2D_INBOX_MAX_TILES) return 0;
    return align_up((size_t)num_tiles * GI2D_INBOX_SLOTS * 4 * sizeof(float4));
}
__host__ __device__ __forceinline__ int list_base(int tile) { return tile * GI2D_FAST_LROW + GI2D_FAST_HDR; }

// ----------------------------------------------------------------------------------------- fill
// Tile boxes are kept as (min | max << 16) per axis; 0/0 is the empty box of a gaussian that is in no list.
__device__ __forceinline__ int2 pack_box(int mnx, int mny, int mxx, int mxy) {
    return make_int2(mnx | (mxx << 16), mny | (mxy << 16));
}
// Bin gaussian g with tile box [mnx, mxx) x [mny, mxy) (`member` false: it is in no tile -- culled, clipped or off
// screen): append it to the rows of the tiles of the new box that were not in the box it was binned with before.
// `old`: prev_box[g], loaded by the caller together with its other inputs -- read here, after the caller's stores, it
// would be one more dependent memory round trip at the end of a latency-bound kernel.
// In two halves: fill_diff_begin stores the new box and issues the FIRST trip's returning atomics, fill_diff_end stores
// the ids into the slots they returned (and runs any further trips: boxes of more than GI2D_FILL_BATCH tiles).  A caller
// puts its own stores between the two: they are then issued while the atomics are in flight instead of after their
// round trip (a returning atomic is waited for in issue order, behind every store issued before it) -- the binning step
// sits at the very end of the per-gaussian kernels' dependency chain, where this round trip cost the training update
// kernel 4.8 of its 13.9 us.
struct FillPending {
    int c[GI2D_FILL_BATCH], p[GI2D_FILL_BATCH];
    int nt, w, mnx, mny, omnx, omxx, omny, omxy, di, dj;  // nt == 0: nothing to append
    int pool;                                             // first pool row of the gaussian's run (-1: none)
    bool live;                                            // the last trip issued atomics (c / p are set)
};
// What a binning step that may use the inboxes hands to fill_diff_begin: where they are, what the gaussian knows of its
// old tiles, and its record (the words write_record stores; the pool word is filled in once the run is known).
#ifdef GI2D_INBOX_STATS /* development aid (tools/inbox_stats.py): counters in the padding of tile row 0's header */
#define GI2D_INBOX_STAT(word) atomicAdd(&lists[word], 1)
#else
#define GI2D_INBOX_STAT(word) \
    do {                      \
    } while (0)
#endif
struct InboxFill {
    Inbox ib;
    InboxSrc src;
    float4 q[4];
};
// Tile (ti, tj) has been entered by a gaussian whose old box is f.o*: through the inbox if the nearest tile of the old
// box is a neighbour that handed a rank on (true), through the row's header otherwise.
__device__ __forceinline__ bool inbox_enter(const FillPending &f, const InboxFill &in, int ti, int tj, int tiles_x,
                                            int32_t *__restrict__ lists) {
    const int sj = min(max(tj, f.omnx), f.omxx - 1), si = min(max(ti, f.omny), f.omxy - 1);
    const int ddx = sj - tj, ddy = si - ti;  // not both zero: the tile is outside the old box
    if (ddx < -1 || ddx > 1 || ddy < -1 || ddy > 1) {
        GI2D_INBOX_STAT(11);
        return false;
    }
    const int k = (si - f.omny) * (f.omxx - f.omnx) + (sj - f.omnx);  // < 8: the caller only comes with such boxes
    const int r = (int)in.src.get(k) - 1;
    if ((unsigned)r >= (unsigned)GI2D_TILE_LIST_CAP) {
        GI2D_INBOX_STAT(12);
        return false;
    }
    GI2D_INBOX_STAT(9);
    int d = (ddy + 1) * 3 + (ddx + 1);
    d -= d > 4 ? 1 : 0;
    const int slot = d * GI2D_TILE_LIST_CAP + r, tile = ti * tiles_x + tj;
    float4 *dst = in.ib.recs + 4 * ((size_t)tile * GI2D_INBOX_SLOTS + slot);
    dst[0] = in.q[0], dst[1] = in.q[1], dst[2] = in.q[2], dst[3] = in.q[3];
    atomicOr(reinterpret_cast<unsigned *>(inbox_bits_of(lists, tile)) + (slot >> 5), 1u << (slot & 31));  // no answer needed
    return true;
}
template <bool INBOX>
__device__ __forceinline__ void fill_trip_issue(FillPending &f, int base, int tiles_x, int32_t *__restrict__ lists,
                                                const InboxFill *in = nullptr) {
    // the entered tiles that take the gaussian through their inbox (a ROLLED loop over the trip's tiles: one copy of the
    // record's stores, no indexing of register arrays; a lane has one or two entered tiles, rarely more)
    unsigned handled = 0u;
    if constexpr (INBOX) {
#ifdef GI2D_INBOX_STATS
        if (base == 0) {
            GI2D_INBOX_STAT(13);                       // lanes whose box changed
            if (!in->src.on) GI2D_INBOX_STAT(14);      // ... that may not use the inboxes
            if ((threadIdx.x & 63) == __builtin_ctzll(__ballot(true))) GI2D_INBOX_STAT(15);  // waves
        }
#endif
        bool left = false;  // an entered tile that is left to the header's atomic
        int di = f.di, dj = f.dj;
        const int trips = min(GI2D_FILL_BATCH, f.nt - base);
        for (int q = 0; q < trips; ++q) {
            const int ti = f.mny + di, tj = f.mnx + dj;
            const bool was = tj >= f.omnx && tj < f.omxx && ti >= f.omny && ti < f.omxy;
            if (!was) {
                if (in->src.on && inbox_enter(f, *in, ti, tj, tiles_x, lists))
                    handled |= 1u << q;
                else
                    left = true;
            }
            if (++dj == f.w) dj = 0, ++di;
        }
        if (__ballot(left) == 0ull) {
            // nothing for the headers (the steady state of a fit): the trip's sixteen-fold address arithmetic, its
            // atomics and the stores behind them are not even looked at
            f.di = di, f.dj = dj;
            f.live = false;
            if (f.nt - base <= GI2D_FILL_BATCH) f.nt = 0;  // the box's last trip: fill_diff_end has nothing to do
            return;
        }
    }
    // GI2D_FILL_BATCH tiles per trip: the returning atomics of a trip are issued back to back, then the stores, so a
    // lane pays one round trip per trip instead of one per tile
#pragma unroll
    for (int q = 0; q < GI2D_FILL_BATCH; ++q) {
        const int ti = f.mny + f.di, tj = f.mnx + f.dj;
        const bool was = tj >= f.omnx && tj < f.omxx && ti >= f.omny && ti < f.omxy;  // still listed from before
        f.c[q] = (base + q < f.nt && !was && !((handled >> q) & 1u)) ? (ti * tiles_x + tj) * GI2D_FAST_LROW : -1;
        if (++f.dj == f.w) f.dj = 0, ++f.di;
    }
#ifdef GI2D_INBOX_STATS
    if constexpr (INBOX) {
        bool any = false;
#pragma unroll
        for (int q = 0; q < GI2D_FILL_BATCH; ++q)
            if (f.c[q] >= 0) {
                any = true;
                GI2D_INBOX_STAT(10);  // entered tiles that took the header's atomic
            }
        const unsigned long long wm = __ballot(any);
        if (wm != 0ull && (threadIdx.x & 63) == __builtin_ctzll(wm)) GI2D_INBOX_STAT(7);  // waves that wait for one
    }
#endif
#pragma unroll
    for (int q = 0; q < GI2D_FILL_BATCH; ++q) f.p[q] = f.c[q] >= 0 ? atomicAdd(&lists[f.c[q]], 1) : GI2D_FAST_C;
    f.live = true;
}
__device__ __forceinline__ void fill_trip_store(const FillPending &f, int g, int32_t *__restrict__ lists) {
    if (!f.live) return;  // (the trip issued nothing: c / p are not even set)
#pragma unroll
    for (int q = 0; q < GI2D_FILL_BATCH; ++q)
        if (f.p[q] < GI2D_FAST_C) lists[f.c[q] + GI2D_FAST_HDR + f.p[q]] = g;  // a fuller row is flagged by the tile pass
}
template <bool INBOX = false>
__device__ __forceinline__ FillPending fill_diff_begin(int g, bool member, int mnx, int mny, int mxx, int mxy,
                                                       int tiles_x, PrevBox old, PrevBox *__restrict__ prev_box,
                                                       int32_t *__restrict__ lists, InboxFill *in = nullptr) {
    FillPending f;
    f.nt = 0;
    f.pool = old.z;
    member = member && mxx > mnx && mxy > mny;
    const int2 nw = member ? pack_box(mnx, mny, mxx, mxy) : make_int2(0, 0);
    if (old.x == nw.x && old.y == nw.y) return f;  // the usual case: same tiles as last time, nothing to do
    const int ntiles = member ? (mxx - mnx) * (mxy - mny) : 0;
    int cap = old.w;
    if (ntiles > GI2D_FAST_S && ntiles > cap) {  // its run of pool rows (see PrevBox): rare, and rarer still after the first time
        // A run that has to grow grows by at least half: a gaussian swelling tile by tile (36, 42, 49, ... tiles: the
        // large gaussians of a fixed-population fit, whose workspace is never emptied) would otherwise abandon a run per
        // step -- the pool is a bump allocator, nothing is handed back before gi2d_fast_workspace_init -- and leak many
        // times its live size; with geometric growth the abandoned runs sum to less than twice the live one.
        cap = max(ntiles, cap + (cap >> 1));
        f.pool = atomicAdd(&lists[GI2D_POOL_CURSOR], cap);
    }
    prev_box[g] = make_int4(nw.x, nw.y, f.pool, cap);
    if (!member) return f;  // its old entries are dropped by the tile pass (they fail the membership test)
    f.omnx = old.x & 0xffff, f.omxx = (int)((unsigned)old.x >> 16), f.omny = old.y & 0xffff,
    f.omxy = (int)((unsigned)old.y >> 16);
    f.mnx = mnx, f.mny = mny;
    f.w = mxx - mnx, f.nt = f.w * (mxy - mny);
    f.di = f.dj = 0;
    if constexpr (INBOX) in->q[3].z = __int_as_float(f.pool);
    fill_trip_issue<INBOX>(f, 0, tiles_x, lists, in);
    return f;
}
template <bool INBOX = false>
__device__ __forceinline__ void fill_diff_end(int g, FillPending &f, int tiles_x, int32_t *__restrict__ lists,
                                              const InboxFill *in = nullptr) {
    if (f.nt == 0) return;
    fill_trip_store(f, g, lists);
    for (int base = GI2D_FILL_BATCH; base < f.nt; base += GI2D_FILL_BATCH) {
        fill_trip_issue<INBOX>(f, base, tiles_x, lists, in);
        fill_trip_store(f, g, lists);
    }
}
__device__ __forceinline__ void fill_diff(int g, bool member, int mnx, int mny, int mxx, int mxy, int tiles_x,
                                          PrevBox old, PrevBox *__restrict__ prev_box, int32_t *__restrict__ lists) {
    FillPending f = fill_diff_begin(g, member, mnx, mny, mxx, mxy, tiles_x, old, prev_box, lists);
    fill_diff_end(g, f, tiles_x, lists);
}
// The tile box a gaussian with centre xy and INT radius rad is binned with -- the rule of map_gaussian_to_intersects
// (forward.cu:161-166): radius > 0, not below the clip, box of the int radius.  False: it is in no tile.
__device__ __forceinline__ bool bin_box(const float2 xy, int rad, float radius_clip, int tiles_x, int tiles_y, int &mnx,
                                        int &mny, int &mxx, int &mxy) {
    mnx = mny = mxx = mxy = 0;
    if (rad <= 0 || (float)rad < radius_clip) return false;
    tile_bbox(xy.x, xy.y, (float)rad, tiles_x, tiles_y, mnx, mny, mxx, mxy);
    return mxx > mnx && mxy > mny;
}
// Is tile (tx, ty) in that box?  (What the tile pass asks of every entry of its row.)
__device__ __forceinline__ bool tile_member(const float2 xy, int rad, float radius_clip, int tiles_x, int tiles_y,
                                            int tx, int ty, int &mnx, int &mny, int &mxx, int &mxy) {
    return bin_box(xy, rad, radius_clip, tiles_x, tiles_y, mnx, mny, mxx, mxy) && tx >= mnx && tx < mxx && ty >= mny &&
           ty < mxy;
}
// The records exist in two sets so that ONE launch can hold lanes that still read the records the last tile pass used
// (gradient reduce + projection backward of step i) next to lanes that already write those of step i + 1 (projection +
// binning): the end-of-step kernel of a loop over given inputs runs the two as separate workgroups.  Which set is
// which follows from two counters that are only ever written with values derived from the OTHER one, so every lane of a
// launch computes the same answer whenever it looks (kernels of one stream do not overlap):
//   a binning kernel   reads ver[0] = c, writes set (c + 1) & 1, leaves ver[1] = c + 1;
//   a tile pass        reads ver[1] = b, reads set b & 1, leaves ver[0] = b;
//   "the records the last tile pass used" are set ver[0] & 1.
struct RecSets {
    float4 *base;
    int32_t *ver;
    unsigned stride;  // float4 per set
    int seq;          // a number no other C-ABI call of this process carries (rec_sets): part of a binning call's stamp
};
// What a binning kernel gets: the record set it writes, and the word in which it notes that SOME gaussian is in a tile
// -- stamped, so nobody ever has to reset it: a forward that must render the background when there is not a single
// intersection (rasterize_sum_plus.py:110-118) compares the word with the stamp of the LATEST binning call
// (tile_pass_has_members) instead of waiting for a second launch behind the tile pass.  The stamp is the record-set
// version the call leaves (which only a tile pass advances) AND the call's own number (two words): two binning calls with
// no tile pass between them -- the C ABI allows it -- write the same version, and a second one WITHOUT a member must
// not inherit the first one's word (round 5's stamp was the version alone: bin, bin(empty), forward rendered zeros where
// the reference returns the background).
// Layout rule of this line (`ver`, 64 words of their own): every word is written with PLAIN stores only, by lanes that
// all store the same value in one launch, and read by plain loads of LATER launches -- never an atomic next to them,
// never a read-modify-write (the dropped ticket counters).
#define GI2D_VER_ANY 2     /* word of RecSets::ver: version left by the last binning call that put a gaussian into a tile */
#define GI2D_VER_ANY_SEQ 3 /* ... and that call's number */
#define GI2D_VER_LATEST 4  /* number of the last binning call */
#define GI2D_VER_WORDS 5
static_assert(GI2D_VER_WORDS <= 16, "the stamps share the version words' 64 bytes; word 16 onwards has other owners");
struct BinRecs {
    float4 *recs;
    int32_t *any;
    int stamp, seq;
};
__device__ __forceinline__ BinRecs recs_for_binning(const RecSets &rs, bool writer) {
    const int c = rs.ver[0];
    BinRecs b;
    b.recs = rs.base + (size_t)((c + 1) & 1) * rs.stride;
    b.any = rs.ver + GI2D_VER_ANY;
    b.stamp = c + 1;
    b.seq = rs.seq;
    if (writer) rs.ver[1] = c + 1, rs.ver[GI2D_VER_LATEST] = rs.seq;
    return b;
}
// (a wave that holds a member: one lane of it)
__device__ __forceinline__ void note_member(const BinRecs &b) { b.any[0] = b.stamp, b.any[GI2D_VER_ANY_SEQ - GI2D_VER_ANY] = b.seq; }
__device__ __forceinline__ bool tile_pass_has_members(const RecSets &rs) {
    return rs.ver[GI2D_VER_ANY] == rs.ver[1] && rs.ver[GI2D_VER_ANY_SEQ] == rs.ver[GI2D_VER_LATEST];
}
__device__ __forceinline__ const float4 *recs_for_tile_pass(const RecSets &rs, bool writer) {
    const int b = rs.ver[1];
    if (writer) rs.ver[0] = b;
    return rs.base + (size_t)(b & 1) * rs.stride;
}
__device__ __forceinline__ const float4 *recs_of_last_pass(const RecSets &rs) {
    return rs.base + (size_t)(rs.ver[0] & 1) * rs.stride;
}
inline std::atomic<unsigned> g_call_seq{1};
static inline RecSets rec_sets(const FastWs &w, int n) {
    RecSets rs;
    rs.seq = (int)(g_call_seq.fetch_add(1, std::memory_order_relaxed) & 0x7fffffffu);
    rs.base = w.recs;
    rs.stride = (unsigned)(align_up((size_t)(n > 0 ? n : 1) * 4 * sizeof(float4)) / sizeof(float4));
    rs.ver = w.ver;
    return rs;
}
// What a binning step needs besides the projection: colour / opacity for the records, the workspace's state.
struct BinTarget {
    const float *colors, *opacities;
    PrevBox *prev_box;
    int32_t *lists;
    RecSets recs;
    int32_t *status;
};

// First thing of every binning kernel (lane of gaussian 0): reset the per-call status words (status[2] is the sticky
// copy of the overflow flag; status[3] the fullest tile row above half the capacity the following tile pass sees).
__device__ __forceinline__ void begin_binning(int g, int32_t *__restrict__ status) {
    if (g == 0) {
        status[0] = 0;
        status[1] = 0;
        status[3] = 0;
    }
}
// The record a binning step leaves per gaussian -- everything a tile pass needs of it, gathered with four 16-byte loads
// of ONE line instead of nine dwords from five arrays (each of the ~72 entries of a tile used to cost 5-6 line requests,
// all 1536 tiles asking at once):
//   (gx, gy, a, b) (c, opacity, r, g) (b, hx, hy, box.x) (box.y, radius, pool, id)
// a, b, c: conic; (hx, hy): half extents of the alpha >= 1/255 box (gi2d_common.h::cull_extent), computed here once per
// gaussian instead of once per (tile, gaussian); box: the tile box it is binned with (0/0: in no tile), which is what
// the tile pass tests membership against and derives the partial-row slot from; pool: first row of its run in the row
// pool (PrevBox).
__device__ __forceinline__ void make_record(float4 (&q)[4], int g, float2 xy, float a, float b, float c, float opac,
                                            float cr, float cg, float cb, int2 box, int radius, int pool) {
    float hx, hy;
    cull_extent(xy.x, xy.y, a, b, c, opac, hx, hy);
    q[0] = make_float4(xy.x, xy.y, a, b);
    q[1] = make_float4(c, opac, cr, cg);
    q[2] = make_float4(cb, hx, hy, __int_as_float(box.x));
    q[3] = make_float4(__int_as_float(box.y), __int_as_float(radius), __int_as_float(pool), __int_as_float(g));
}
// (Written through -- gi2d_raster_core.h::store16 -- the records cost the single-image update kernel +0.38 us and bought
// the tile pass behind it nothing: these stores sit at the very end of every wave, where nothing is left to overlap.)
__device__ __forceinline__ void store_record(float4 *__restrict__ recs, int g, const float4 (&q)[4]) {
    float4 *r = recs + 4 * (size_t)g;
    r[0] = q[0], r[1] = q[1], r[2] = q[2], r[3] = q[3];
}
__device__ __forceinline__ void write_record(float4 *__restrict__ recs, int g, float2 xy, float a, float b, float c,
                                             float opac, float cr, float cg, float cb, int2 box, int radius, int pool) {
    float4 q[4];
    make_record(q, g, xy, a, b, c, opac, cr, cg, cb, box, radius, pool);
    store_record(recs, g, q);
}
struct BinRec {
    GaussRec r;
    float hx, hy;
    int2 box;
    int pool;
};
__device__ __forceinline__ BinRec record_of(const float4 q0, const float4 q1, const float4 q2, const float4 q3, int g) {
    BinRec o;
    o.r.gx = q0.x, o.r.gy = q0.y, o.r.a = q0.z, o.r.b = q0.w;
    o.r.c = q1.x, o.r.opac = q1.y, o.r.cr = q1.z, o.r.cg = q1.w;
    o.r.cb = q2.x, o.r.slot = -1, o.r.gid = g, o.r.pad = 0;
    o.hx = q2.y, o.hy = q2.z;
    o.box = make_int2(__float_as_int(q2.w), __float_as_int(q3.x));
    o.pool = __float_as_int(q3.z);
    return o;
}
__device__ __forceinline__ BinRec load_record(const float4 *__restrict__ recs, int g) {
    const float4 *p = recs + 4 * (size_t)g;
    const float4 q0 = p[0], q1 = p[1], q2 = p[2];
    const float4 q3 = p[3];
    return record_of(q0, q1, q2, q3, g);
}
// a record waiting in an inbox slot: the same words, the gaussian's id in the last one
__device__ __forceinline__ BinRec load_inbox_record(const float4 *__restrict__ p) {
    const float4 q0 = p[0], q1 = p[1], q2 = p[2];
    const float4 q3 = p[3];
    return record_of(q0, q1, q2, q3, __float_as_int(q3.w));
}
__device__ __forceinline__ void unpack_box(int2 box, int &mnx, int &mny, int &mxx, int &mxy) {
    mnx = box.x & 0xffff, mxx = (int)((unsigned)box.x >> 16), mny = box.y & 0xffff, mxy = (int)((unsigned)box.y >> 16);
}
// Binning step of one gaussian: its tile box (the rasterizer's radius_clip equals the projection's on this path), the
// appends to the rows of tiles it has entered, and its record.
// `between`: the caller's own stores, issued after the binning step's atomics and before the stores that wait for them
// (fill_diff_begin / _end above).
// INBOX (the update kernel of a fit): `in` = the inboxes and what the gaussian knows of its old tiles (InboxFill; its
// record words are filled in here).
template <bool INBOX = false, class Between>
__device__ __forceinline__ void bin_one(int g, float2 xy, int radius, bool has_tiles, float ka, float kb, float kc,
                                        float opac, float cr, float cg, float cb, int tiles_x, int tiles_y,
                                        float radius_clip, PrevBox old_box, PrevBox *__restrict__ prev_box,
                                        int32_t *__restrict__ lists, const BinRecs &br, Between between,
                                        InboxFill *in = nullptr) {
    int mnx, mny, mxx, mxy;
    const bool member = bin_box(xy, radius, radius_clip, tiles_x, tiles_y, mnx, mny, mxx, mxy) && has_tiles;
    {  // one store per wave that holds a member (by its first such lane)
        const unsigned long long mm = __ballot(member);
        if (mm != 0ull && (int)(threadIdx.x & 63) == __builtin_ctzll(mm)) note_member(br);
    }
    float4 *recs = br.recs;
    const int2 box = member ? pack_box(mnx, mny, mxx, mxy) : make_int2(0, 0);
    if constexpr (INBOX) {
        // the record first (registers): an entered tile's inbox gets a copy
        make_record(in->q, g, xy, ka, kb, kc, opac, cr, cg, cb, box, radius, old_box.z);
        FillPending f = fill_diff_begin<true>(g, member, mnx, mny, mxx, mxy, tiles_x, old_box, prev_box, lists, in);
        in->q[3].z = __int_as_float(f.pool);
        store_record(recs, g, in->q);
        between();
        fill_diff_end<true>(g, f, tiles_x, lists, in);
    } else {
        FillPending f = fill_diff_begin(g, member, mnx, mny, mxx, mxy, tiles_x, old_box, prev_box, lists);
        write_record(recs, g, xy, ka, kb, kc, opac, cr, cg, cb, box, radius, f.pool);
        between();
        fill_diff_end(g, f, tiles_x, lists);
    }
}
__device__ __forceinline__ void bin_one(int g, float2 xy, int radius, bool has_tiles, float ka, float kb, float kc,
                                        float opac, float cr, float cg, float cb, int tiles_x, int tiles_y,
                                        float radius_clip, PrevBox old_box, PrevBox *__restrict__ prev_box,
                                        int32_t *__restrict__ lists, const BinRecs &recs) {
    bin_one(g, xy, radius, has_tiles, ka, kb, kc, opac, cr, cg, cb, tiles_x, tiles_y, radius_clip, old_box, prev_box, lists,
            recs, [] {});
}
template <bool INBOX = false, class Between>
__device__ __forceinline__ void bin_projected(int g, const ProjOut &o, float opac, float cr, float cg, float cb,
                                              int tiles_x, int tiles_y, float radius_clip, PrevBox old_box,
                                              PrevBox *__restrict__ prev_box, int32_t *__restrict__ lists,
                                              const BinRecs &recs, Between between, InboxFill *in = nullptr) {
    bin_one<INBOX>(g, o.xy, o.radius, o.tiles_hit > 0, o.k0, o.k1, o.k2, opac, cr, cg, cb, tiles_x, tiles_y, radius_clip,
                   old_box, prev_box, lists, recs, between, in);
}
__device__ __forceinline__ void bin_projected(int g, const ProjOut &o, float opac, float cr, float cg, float cb,
                                              int tiles_x, int tiles_y, float radius_clip, PrevBox old_box,
                                              PrevBox *__restrict__ prev_box, int32_t *__restrict__ lists,
                                              const BinRecs &recs) {
    bin_projected(g, o, opac, cr, cg, cb, tiles_x, tiles_y, radius_clip, old_box, prev_box, lists, recs, [] {});
}

// partial-row code of gaussian g in tile (tx, ty) of its box: >= 0 gaussian-major row, < 0: -(pool row) - 1
// (`pool`: first row of its run, from its record)
__device__ __forceinline__ int partial_slot(int g, int2 box, int tx, int ty, int pool) {
    int mnx, mny, mxx, mxy;
    unpack_box(box, mnx, mny, mxx, mxy);
    const int w = mxx - mnx, ntiles = w * (mxy - mny), at = (ty - mny) * w + (tx - mnx);
    if (ntiles <= GI2D_FAST_S) return g * GI2D_FAST_S + at;
    return -(pool + at) - 1;
}
// Where the partial row with that code lives; nullptr (and the overflow status raised) for a pool row past the pool's end.
__device__ __forceinline__ float4 *partial_row(int slot, float4 *__restrict__ partial_g, float4 *__restrict__ partial_big,
                                               int pool_rows, int32_t *__restrict__ status) {
    if (slot >= 0) return partial_g + GI2D_FAST_ROW * (size_t)slot;
    const int row = -slot - 1;
    if (row < 0 || row >= pool_rows) {  // the pool ran out (PrevBox): the caller falls back as for a tile row that
        atomicOr(&status[1], GI2D_STATUS_POOL);  // overflowed, but is told which of the two it was
        atomicOr(&status[2], GI2D_STATUS_POOL);
        return nullptr;
    }
    return partial_big + GI2D_FAST_ROW * (size_t)row;
}

#ifndef GI2D_HEAD_TRACE /* gi2d_fused_core.h defines it for its phase trace (development aid) */
#define GI2D_HEAD_TRACE(i) \
    do {                   \
    } while (0)
#endif
// ------------------------------------------------------------------------------- head of a tile pass
// What every tile kernel of the fast path does first (all 256 lanes of the tile's workgroup): read the tile's row,
// test every entry against the box its gaussian is binned with NOW (its record), order the survivors by ascending id
// (the stable key sort of the reference pipeline: bin_and_sort_gaussians with depth == 0), write the row / header /
// tile_bins back where they changed, and hand every survivor to the caller in two steps: `prep(g, record)` forms what
// the caller stages of it (registers; ONCE per entry), `put(rank, g, staged)` stores it (rank = position in the ascending
// list; the caller stages rank < 256).  Entries [0, sorted_len) are already ascending, so a survivor among
// them only needs the number of survivors in front of it (a ballot scan) plus the number of smaller APPENDED ids; an
// appended entry is ranked against everything.  With no append since the last pass -- the steady state of a fit --
// there is no loop at all, and nothing is stored.  The row header and the first 256 ids are ONE round of loads (the ids
// are read before the count is known; slots past the count hold stale ids that are ignored), the records the second.
// `ids`: GI2D_FAST_C ints of LDS, `grp`: 32 ints of LDS; both are free again after the caller's next workgroup
// barrier.  Returns the number of survivors.
// OPTIMISTIC: a survivor of the ascending part is `put` at rank = its position BEFORE the workgroup barrier -- right
// when its record arrives -- which is where it ends up whenever nothing was dropped or appended; only otherwise
// (tile-uniform, known after the barrier) everything is `put` again at its true rank -- the stores only: what was
// prepared is kept (on a scene whose gaussians move, most tiles take this way every step).  `put` must then be
// idempotent LDS staging for rank < GI2D_TILE_LIST_CAP (entries past the cap are only ever put once, at the end).  With
// OPTIMISTIC the staging is complete and visible to the whole workgroup on return (the usual case costs ONE barrier in
// all); otherwise the caller's barrier after the call closes it.
// The first round of loads of a tile's row (header + this lane's first id), separable from the rest so that a caller
// whose tile index is itself the result of a load (the tile order of the single-pass kernel) can request the row of the
// tile it EXPECTS -- the identity order of an evenly populated scene -- in the same round as that index, and only
// re-request on a miss (head_row_for): one dependent memory round trip less at the top of every tile.
struct HeadRow {
    int hdr_count, hdr_sorted, id0;
    unsigned inbox_w;  // word (lane) of the tile's inbox bitmap: every wave holds all 64 words
};
// `inbox`: the pass takes entrants out of the tiles' inboxes (compile-time at every call site).
__device__ __forceinline__ HeadRow head_row_load(const int32_t *__restrict__ lists, int tile, bool inbox) {
    const int32_t *row = lists + (size_t)tile * GI2D_FAST_LROW;
    HeadRow h;
    h.hdr_count = row[0], h.hdr_sorted = row[1], h.id0 = row[GI2D_FAST_HDR + threadIdx.x];
    h.inbox_w = 0u;
    if (inbox) h.inbox_w = (unsigned)row[GI2D_FAST_HDR + GI2D_FAST_C + (threadIdx.x & 63)];
    return h;
}
// `order[slot]` is the tile this workgroup handles; the row of tile `slot` is requested alongside.
__device__ __forceinline__ HeadRow head_row_for(const int32_t *__restrict__ lists, const int32_t *__restrict__ order,
                                                int slot, int &tile, bool inbox) {
    HeadRow h = head_row_load(lists, slot, inbox);
    tile = __builtin_amdgcn_readfirstlane(order[slot]);
    if (tile != slot) h = head_row_load(lists, tile, inbox);  // workgroup-uniform
    return h;
}
// The general form of the head for a row of more than 64 candidates: EPT = list entries per lane.  A row of at most 256
// candidates -- every tile of the bench scene, nearly every tile of a trained one -- is served by the EPT = 1
// instantiation, whose per-lane state is scalars and which has none of the `u < rounds` bookkeeping of the four-entry
// form (a third of the head's scalar instructions on such rows).
template <bool OPTIMISTIC, int EPT, bool INBOX, class Prep, class Put>
__device__ __forceinline__ int tile_list_head_rows(int *ids, int *grp, int tile, int tx, int ty,
                                                   const float4 *__restrict__ recs, int32_t *__restrict__ row,
                                                   int2 *__restrict__ tile_bins, int32_t *__restrict__ status, Prep prep,
                                                   Put put, int id0, int hdr_count, int hdr_sorted, int count, int sorted,
                                                   unsigned inbox_w, const float4 *__restrict__ inbox) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int my_id[EPT];
    my_id[0] = id0;
    // ---- the tile's inbox (Inbox): lane tid looks at byte tid of the bitmap, i.e. at slots 8 tid .. 8 tid + 7.  The
    // FIRST entrant of a byte is staged like a row entry -- its record (the inbox's copy) arrives in the same round as
    // the row entries' records; further ones of the same byte (two gaussians of neighbouring ranks entering from the
    // same side: rare) only leave their ids and fetch their records when they are ranked.  `inbox` = this tile's slots.
    int k_in = 0, ent_id = -1, ent_at = 0;
    unsigned ent_more = 0u;
    decltype(prep(0, BinRec())) st_in;
    const bool any_in = INBOX && __ballot(inbox_w != 0u) != 0ull;  // workgroup-uniform
    if (any_in) {
        const int pc = __popc(inbox_w), incl_w = wave_inclusive_scan(pc);
        k_in = wave_read_lane(incl_w, 63);
        const unsigned wd = (unsigned)__shfl((int)inbox_w, tid >> 2, 64);
        const int sh = 8 * (tid & 3);
        const unsigned byte = (wd >> sh) & 0xffu;
        ent_at = count + __shfl(incl_w - pc, tid >> 2, 64) + __popc(wd & ((1u << sh) - 1u));
        if (byte != 0u) {
            const BinRec rin = load_inbox_record(inbox + 4 * (size_t)(tid * 8 + __ffs((int)byte) - 1));
            ent_id = rin.r.gid;
            st_in = prep(ent_id, rin);
            if (ent_at < GI2D_FAST_C) ids[ent_at] = ent_id;
            ent_more = byte & (byte - 1u);
#ifdef GI2D_INBOX_STATS /* development aid: entrants taken in / of them second and later ones of a lane's byte */
            atomicAdd(&(row - (size_t)tile * GI2D_FAST_LROW)[5], 1);
            if (ent_more) atomicAdd(&(row - (size_t)tile * GI2D_FAST_LROW)[6], __popc(ent_more));
#endif
            int at = ent_at + 1;
            for (unsigned m = ent_more; m; m &= m - 1u, ++at)
                if (at < GI2D_FAST_C) ids[at] = __float_as_int(inbox[4 * (size_t)(tid * 8 + __ffs((int)m) - 1) + 3].w);
        }
    }
    const int count_all = min(count + k_in, GI2D_FAST_C);  // candidates in `ids` once the barrier is passed
    if (tid == 0 && (hdr_count > GI2D_FAST_C || count + k_in > GI2D_FAST_C)) {  // more candidates than a row holds:
        atomicOr(&status[1], 1);                                                  // the caller must fall back
        atomicOr(&status[2], 1);
    }
    // high-water mark for callers that read the status one call late (the autograd wrappers): as long as no row was
    // more than half full, an overflow cannot be one slowly moving step away
    if (tid == 0 && hdr_count + k_in > GI2D_FAST_C / 2) atomicMax(&status[3], hdr_count + k_in);
    if (tid >= count) my_id[0] = -1;
    GI2D_HEAD_TRACE(11);
    // the first entry's record is kept in registers; entries past 256 (rare) fetch theirs again when they are staged
    BinRec r0;
    bool keep[EPT];
    int pos[EPT];
#pragma unroll
    for (int u = 0; u < EPT; ++u) keep[u] = false, pos[u] = 0;
    if constexpr (EPT > 1) {
#pragma unroll
        for (int u = 1; u < EPT; ++u) my_id[u] = -1;
    }
    const auto member = [&](int2 box) {
        int mnx, mny, mxx, mxy;
        unpack_box(box, mnx, mny, mxx, mxy);
        return tx >= mnx && tx < mxx && ty >= mny && ty < mxy;  // the empty box 0/0 contains no tile
    };
    const int rounds = EPT == 1 ? 1 : (count + 255) >> 8;  // tile-uniform
    const int wv_s = __builtin_amdgcn_readfirstlane(wv);
    decltype(prep(0, BinRec())) st0;  // what entry 0 stages, formed ONCE (a re-ranked entry only repeats the stores)
    if ((wv_s << 6) >= count) {
        // a wave whose 64 slots lie past the row's end (two of four at 72 candidates per tile) has no entry to load,
        // test, count or stage: it reports empty groups and waits
        if (lane == 0) {
#pragma unroll
            for (int u = 0; u < EPT; ++u) grp[wv + 4 * u] = 0, grp[16 + wv + 4 * u] = 0;
        }
    } else {
        if (my_id[0] >= 0) {
            r0 = load_record(recs, my_id[0]);
            keep[0] = member(r0.box);
            if (keep[0]) st0 = prep(my_id[0], r0);
        }
        if constexpr (EPT > 1) {
#pragma unroll
            for (int u = 1; u < EPT; ++u) {
                if (u < rounds) {
                    const int e = tid + 256 * u;
                    if (e < count) {
                        my_id[u] = row[GI2D_FAST_HDR + e];
                        const float4 *p = recs + 4 * (size_t)my_id[u];
                        keep[u] = member(make_int2(__float_as_int(p[2].w), __float_as_int(p[3].x)));
                    }
                }
            }
        }
        GI2D_HEAD_TRACE(12);
#pragma unroll
        for (int u = 0; u < EPT; ++u) {
            if (u < rounds) {
                const int e = tid + 256 * u;
                if (e < count) ids[e] = keep[u] ? my_id[u] : -1;
                const unsigned long long kp = __ballot(keep[u] && e < sorted), ka = __ballot(keep[u] && e >= sorted);
                pos[u] = __popcll(kp & lanemask_lt());
                if (lane == 0) {
                    grp[wv + 4 * u] = __popcll(kp);       // survivors of the ascending part in entries [64 i, 64 i + 64)
                    grp[16 + wv + 4 * u] = __popcll(ka);  // survivors of the appended part
                }
            } else if (lane == 0) {
                grp[wv + 4 * u] = 0;
                grp[16 + wv + 4 * u] = 0;
            }
        }
    }
    if (OPTIMISTIC && keep[0] && tid < sorted) put(tid, my_id[0], st0);  // tid < 256 = GI2D_TILE_LIST_CAP
    __syncthreads();
    // the inbox is taken: every wave has read the bitmap (before the barrier), wave 0 clears it
    if (any_in && wv == 0 && inbox_w != 0u) row[GI2D_FAST_HDR + GI2D_FAST_C + lane] = 0;
    GI2D_HEAD_TRACE(13);
    // one wave-level scan gives every lane what it needs: lane i < 16 holds group i's ascending survivors, lanes
    // 16..31 the appended ones
    const int cnt = (lane < 32 && (lane & 15) < 4 * EPT) ? grp[lane] : 0;  // (the groups this form wrote)
    const int incl = wave_inclusive_scan(cnt);
    const int len = wave_read_lane(incl, 31) + (count_all - count);  // (entrants are members: their tile took them in)
    const int wv_u = __builtin_amdgcn_readfirstlane(wv);
    int before[EPT];
#pragma unroll
    for (int u = 0; u < EPT; ++u) before[u] = wave_read_lane(incl - cnt, wv_u + 4 * u);
    const bool clean = len == count && sorted == count && !any_in;  // nothing dropped, nothing appended: rank == position
    if (tid == 0) {
        // an overflowed row keeps its count: it has lost entries, so every pass flags it until the workspace is emptied
        if (hdr_count <= GI2D_FAST_C && (hdr_count != len || hdr_sorted != len)) {
            row[0] = len;
            row[1] = len;
        }
        tile_bins[tile] = make_int2(list_base(tile), list_base(tile) + len);
    }
    if (OPTIMISTIC && clean) {
        if constexpr (EPT > 1) {
#pragma unroll
            for (int u = 1; u < EPT; ++u)  // entries past the cap: put once, here (rank == position)
                if (u < rounds && keep[u]) put(tid + 256 * u, my_id[u], prep(my_id[u], load_record(recs, my_id[u])));
        }
        return len;
    }
    // (the optimistic staging was complete at the barrier above: what follows overwrites it in program order)
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
        if (u >= rounds || !keep[u]) continue;
        const int e = tid + 256 * u, g = my_id[u];
        // ids are unique within a row, so "smaller" needs no tie rule; dropped entries read as -1 = 0xffffffff
        int rank, lo, hi;
        if (e < sorted)
            rank = before[u] + pos[u], lo = sorted, hi = count_all;
        else
            rank = 0, lo = 0, hi = count_all;
        for (int q = lo; q < hi; ++q) rank += ((unsigned)ids[q] < (unsigned)g) ? 1 : 0;
        const bool in_place = e < sorted && rank == e;  // an entry of the ascending part that nothing in front of it moved
        if (!in_place) row[GI2D_FAST_HDR + rank] = g;
        // ... and whose optimistic staging at rank = position (before the barrier) therefore already is the final one
        if (!(OPTIMISTIC && u == 0 && in_place)) put(rank, g, u == 0 ? st0 : prep(g, load_record(recs, g)));
    }
    if (ent_id >= 0 && ent_at < GI2D_FAST_C) {  // the entrants: ranked against everything
        int rank = 0;
        for (int q = 0; q < count_all; ++q) rank += ((unsigned)ids[q] < (unsigned)ent_id) ? 1 : 0;
        row[GI2D_FAST_HDR + rank] = ent_id;
        put(rank, ent_id, st_in);
        int at = ent_at + 1;
        for (unsigned m = ent_more; m; m &= m - 1u, ++at) {
            if (at >= GI2D_FAST_C) break;
            const int g = ids[at];
            rank = 0;
            for (int q = 0; q < count_all; ++q) rank += ((unsigned)ids[q] < (unsigned)g) ? 1 : 0;
            row[GI2D_FAST_HDR + rank] = g;
            put(rank, g, prep(g, load_record(recs, g)));  // (the records are as the inbox's copies: no binning since)
        }
    }
    if (OPTIMISTIC) __syncthreads();  // OPTIMISTIC callers need no barrier of their own after the call
    return len;
}

// INBOX: the head takes the entrants out of the tile's inbox (Inbox) -- the tile pass that follows the one kernel that
// puts any in; every other caller is built without that code (and without the bitmap's load).
template <bool OPTIMISTIC, bool INBOX = false, class Prep, class Put>
__device__ __forceinline__ int tile_list_head(int *ids, int *grp, int tile, int tx, int ty,
                                              const float4 *__restrict__ recs, int32_t *__restrict__ lists,
                                              int2 *__restrict__ tile_bins, int32_t *__restrict__ status, Prep prep, Put put,
                                              const Inbox &ib, const HeadRow hr) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int32_t *row = lists + (size_t)tile * GI2D_FAST_LROW;
    const bool any_in = INBOX && __ballot(hr.inbox_w != 0u) != 0ull;  // entrants in the tile's inbox (workgroup-uniform)
    const float4 *inbox = ib.recs + 4 * ((size_t)tile * GI2D_INBOX_SLOTS);
    const int hdr_count_v = hr.hdr_count, hdr_sorted_v = hr.hdr_sorted;
    int my_id[1];
    my_id[0] = hr.id0;
    // the header is the same for the whole workgroup: say so (vector loads leave it, and everything derived from it --
    // count, sorted length, rounds -- in vector registers; the single-pass tile kernel sits at its 80-register budget)
    const int hdr_count = __builtin_amdgcn_readfirstlane(hdr_count_v);
    const int hdr_sorted = __builtin_amdgcn_readfirstlane(hdr_sorted_v);
    const int count = min(max(hdr_count, 0), GI2D_FAST_C), sorted = min(max(hdr_sorted, 0), count);
    if (count <= 64 && !any_in) {
        // A row of at most 64 candidates -- every tile of a 2040x1356 image at 50 000 gaussians, every tile of a fit's
        // first 45 000 iterations (5 000 ... 14 000 large gaussians: ~24 per tile) -- is ONE wave's work: membership,
        // ranks (ballot / compare loop over the wave's own LDS copy), row write-back and staging need nothing from the
        // other three waves, which used to run the general path's loads, ballots, group counts and wave scan for
        // nothing (~80 instructions each, ~250 of a sparse tile's ~3 200).  They wait at the barrier and read the length.
        if (wv == 0) {
            const int g = lane < count ? my_id[0] : -1;
            BinRec r0;
            bool keep = false;
            if (g >= 0) {
                r0 = load_record(recs, g);
                int mnx, mny, mxx, mxy;
                unpack_box(r0.box, mnx, mny, mxx, mxy);
                keep = tx >= mnx && tx < mxx && ty >= mny && ty < mxy;  // the empty box 0/0 contains no tile
            }
            const unsigned long long km = __ballot(keep);
            const int len = __popcll(km);
            int rank = lane;
            if (!(len == count && sorted == count)) {  // something was dropped or appended: rank = smaller survivors
                ids[lane] = keep ? g : -1;             // (ids are unique; dropped entries read as 0xffffffff)
                __builtin_amdgcn_wave_barrier();
                rank = 0;
                for (int q = 0; q < count; ++q) rank += ((unsigned)ids[q] < (unsigned)g) ? 1 : 0;
                const bool in_place = lane < sorted && rank == lane;
                if (keep && !in_place) row[GI2D_FAST_HDR + rank] = g;
            }
            if (keep) put(rank, g, prep(g, r0));
            if (lane == 0) {
                if (hdr_count != len || hdr_sorted != len) {
                    row[0] = len;
                    row[1] = len;
                }
                tile_bins[tile] = make_int2(list_base(tile), list_base(tile) + len);
                grp[0] = len;
            }
        }
        __syncthreads();  // staging complete and visible (OPTIMISTIC or not: a caller's own barrier after this is harmless)
        return grp[0];
    }
    if (count <= 256)  // workgroup-uniform
        return tile_list_head_rows<OPTIMISTIC, 1, INBOX>(ids, grp, tile, tx, ty, recs, row, tile_bins, status, prep, put, hr.id0,
                                                  hdr_count, hdr_sorted, count, sorted, hr.inbox_w, inbox);
    return tile_list_head_rows<OPTIMISTIC, GI2D_FAST_EPT, INBOX>(ids, grp, tile, tx, ty, recs, row, tile_bins, status, prep, put,
                                                          hr.id0, hdr_count, hdr_sorted, count, sorted, hr.inbox_w, inbox);
}

// ------------------------------------------------------------------------------- tile order
// All workgroups of the single-pass tile kernel are resident at once on a 768x512 image (6 per CU), and the
// dispatcher was observed to put workgroups b, b + 256, b + 512, ... on the same CU; a CU is done when its six
// tiles are, and tile populations differ (40...108 gaussians at N=50 000), so the slowest CU finished 5 us after
// the fastest.  This routine -- one extra workgroup of the training update kernel, where it hides behind the other
// workgroups -- sorts the tiles by the population the tile pass has just seen (counting sort, descending) and deals
// them to workgroup indices in snake order over the 256 CU slots, for the NEXT iteration's tile pass (populations
// drift slowly during training).  Any permutation gives the same results; placement is a pure speed choice
// (dispatch order is undefined by contract).  Measured: training iteration at N=50 000 50.2 -> 44.6 us; hot-path
// step 38.3 -> 37.0 us.  Its LDS histogram atomics (many lanes per bin) take ~5 us, so it only rides on kernels that
// are longer than that (the training update kernel; the end-of-step kernel above 32k gaussians).
#define GI2D_ORDER_BINS (GI2D_FAST_C + 1) /* populations 0 .. GI2D_FAST_C */
#define GI2D_CU_SLOTS 256
#define GI2D_ORDER_MAX_TILES 2048 /* larger grids run in several rounds of resident workgroups and balance themselves */
// `always`: skip the "worth it?" test below (the training update kernel, which only asks every 16th step).
__device__ __forceinline__ void compute_tile_order(const int2 *__restrict__ tile_bins, int num_tiles,
                                                   int32_t *__restrict__ tile_order, bool always = false) {
    if (num_tiles <= GI2D_CU_SLOTS || num_tiles > GI2D_ORDER_MAX_TILES) return;  // order stays the identity
    __shared__ int hist[GI2D_ORDER_BINS + 1];
    const int tid = threadIdx.x, bs = blockDim.x;  // 64 or 256 lanes
    for (int b = tid; b <= GI2D_ORDER_BINS; b += bs) hist[b] = 0;
    // every lane's (<= 32) tile populations: all loads in flight together, kept in registers for both passes (a
    // lane-serial chain of dependent loads would put this workgroup on the kernel's critical path)
    constexpr int PL = GI2D_ORDER_MAX_TILES / 64;
    int pop[PL];
    // (unconditional loads at a clamped index, the bound applied afterwards: a conditional load each sits in a block of
    // its own behind its own s_waitcnt vmcnt(0) -- thirty-two round trips one after the other)
    int2 rr[PL];
#pragma unroll
    for (int q = 0; q < PL; ++q) rr[q] = tile_bins[min(tid + q * bs, num_tiles - 1)];
#pragma unroll
    for (int q = 0; q < PL; ++q) pop[q] = tid + q * bs < num_tiles ? min(max(rr[q].y - rr[q].x, 0), GI2D_ORDER_BINS - 1) : 0;
    // Worth it?  With populations as even as a uniform scene's (fullest tile < 1.75 x the mean; 1.49 at 50 000 uniform
    // gaussians) dealing them out buys the tile pass 0.1-0.2 us and this workgroup, the longest of the end-of-step
    // kernel, costs that kernel 1.1 us; on a trained scene (gaussians crowd where the detail is) the same ordering is
    // worth 3.8 us per iteration.  The order in place stays (any permutation is correct).
    {
        __shared__ int red_max[4], red_sum[4];
        int mx = 0, sm = 0;
#pragma unroll
        for (int q = 0; q < PL; ++q) mx = max(mx, pop[q]), sm += pop[q];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) mx = max(mx, __shfl_xor(mx, d, 64)), sm += __shfl_xor(sm, d, 64);
        if ((tid & 63) == 0) red_max[tid >> 6] = mx, red_sum[tid >> 6] = sm;
        __syncthreads();
        mx = 0, sm = 0;
        for (int k = 0; k < (bs >> 6); ++k) mx = max(mx, red_max[k]), sm += red_sum[k];
        if (!always && 4 * mx * num_tiles <= 7 * sm) return;  // workgroup-uniform
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < PL; ++q)
        if (tid + q * bs < num_tiles) atomicAdd(&hist[GI2D_ORDER_BINS - 1 - pop[q]], 1);  // bin 0 = fullest tiles
    __syncthreads();
    // exclusive scan of the bins by the first wave: 17 bins per lane in registers, one wave scan of the lane sums
    if (tid < 64) {
        constexpr int PER = (GI2D_ORDER_BINS + 63) / 64;
        int v[PER], sum = 0;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int b = tid * PER + q;
            v[q] = b < GI2D_ORDER_BINS ? hist[b] : 0;
            sum += v[q];
        }
        int run = wave_inclusive_scan(sum) - sum;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int b = tid * PER + q;
            if (b < GI2D_ORDER_BINS) hist[b] = run;
            run += v[q];
        }
    }
    __syncthreads();
    const int full_rounds = num_tiles / GI2D_CU_SLOTS;
#pragma unroll
    for (int q = 0; q < PL; ++q) {
        const int t = tid + q * bs;
        if (t >= num_tiles) continue;
        const int rank = atomicAdd(&hist[GI2D_ORDER_BINS - 1 - pop[q]], 1);  // ties: any order
        const int round = rank / GI2D_CU_SLOTS, pos = rank % GI2D_CU_SLOTS;
        const int slot = (round < full_rounds && (round & 1)) ? GI2D_CU_SLOTS - 1 - pos : pos;
        tile_order[round * GI2D_CU_SLOTS + slot] = t;
    }
}

// acc[11] <- ordered sum of gaussian g's partial rows; `box`: the tile box g was binned with for the tile pass that
// wrote them and `pool` its run of pool rows (PrevBox / the fields of its record; 0/0 for lanes without a gaussian).
// Must be called by whole waves.  Order of the sum (the reference-shaped ops use the same, gi2d_raster.hip): ascending
// tile id up to GI2D_BIG_TILES_F tiles; beyond, lane l of the wave adds the box's tiles l, l + 64, ... and the 64 sums meet
// in a fixed butterfly.
static_assert(GI2D_FAST_S == GI2D_BIG_TILES_F, "gaussian-major rows up to the size where the order of the sum changes");
static_assert(GI2D_REDUCE_BATCH >= 8 && GI2D_TILE_LIST_CAP == 256, "reduce_one reads the ranks of a box of <= 8 tiles in its first trip, eight bits each");
// `src` (optional): the ranks the tile pass left in the spare word of the rows (store_partial_row), for a box of at most
// eight tiles -- what the gaussian needs to enter a neighbouring tile through its inbox (InboxSrc).
// `ahead` (AHEAD > 0): the first AHEAD gaussian-major rows of g, requested by rows_ahead() together with the caller's
// first round of loads -- their addresses depend on g alone, so they need not wait for the box that says how many of
// them count (the bench scene's gaussians lie on 1, 2 or 4 tiles: the box -> rows round trip disappears for every wave
// without a larger one).  What is added, and in which order, is the same with or without.
template <int AHEAD>
struct RowsAhead {
    float4 r[AHEAD > 0 ? AHEAD : 1][3];
};
template <int AHEAD>
__device__ __forceinline__ RowsAhead<AHEAD> rows_ahead(const float4 *__restrict__ partial_g, int g_in_rows) {
    static_assert(AHEAD <= GI2D_REDUCE_BATCH && AHEAD <= GI2D_FAST_S, "rows of the first trip only");
    RowsAhead<AHEAD> a;
    const float4 *rows = partial_g + GI2D_FAST_ROW * ((size_t)g_in_rows * GI2D_FAST_S);
#pragma unroll
    for (int q = 0; q < AHEAD; ++q) {
        a.r[q][0] = rows[GI2D_FAST_ROW * q];
        a.r[q][1] = rows[GI2D_FAST_ROW * q + 1];
        a.r[q][2] = rows[GI2D_FAST_ROW * q + 2];
    }
    return a;
}
template <int AHEAD = 0>
__device__ __forceinline__ void reduce_one(int g, int2 box, int pool, int pool_rows, const float4 *__restrict__ partial_g,
                                           const float4 *__restrict__ partial_big, float (&acc)[11],
                                           InboxSrc *src = nullptr, const RowsAhead<AHEAD> *ahead = nullptr) {
    const int lane = threadIdx.x & 63;
    if (src) src->clear();
#pragma unroll
    for (int q = 0; q < 11; ++q) acc[q] = 0.f;
    int mnx, mny, mxx, mxy;
    unpack_box(box, mnx, mny, mxx, mxy);
    const bool mapped = mxx > mnx && mxy > mny;
    const int ntiles = mapped ? (mxx - mnx) * (mxy - mny) : 0;
    if (mapped && ntiles <= GI2D_FAST_S) {
        // GI2D_REDUCE_BATCH rows per trip: their loads are in flight together, the additions stay in ascending
        // tile order
        const float4 *rows = partial_g + GI2D_FAST_ROW * ((size_t)g * GI2D_FAST_S);
        // (the first trip is written out: rows [0, AHEAD) are in registers already)
        const auto trip = [&](int k0, auto have) {
            constexpr int HAVE = decltype(have)::value;
            float4 r[GI2D_REDUCE_BATCH][3];
#pragma unroll
            for (int q = 0; q < GI2D_REDUCE_BATCH; ++q) {
                if (q < HAVE) {
                    r[q][0] = ahead->r[q][0], r[q][1] = ahead->r[q][1], r[q][2] = ahead->r[q][2];
                } else if (k0 + q < ntiles) {
                    r[q][0] = rows[GI2D_FAST_ROW * (k0 + q)];
                    r[q][1] = rows[GI2D_FAST_ROW * (k0 + q) + 1];
                    r[q][2] = rows[GI2D_FAST_ROW * (k0 + q) + 2];
                }
            }
#pragma unroll
            for (int q = 0; q < GI2D_REDUCE_BATCH; ++q)
                if (k0 + q < ntiles) add_partial_row(acc, r[q][0], r[q][1], r[q][2]);
            if (src && k0 == 0) {  // (the first eight tiles' tags; whoever uses them looks at boxes of <= 8 tiles only)
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (q < ntiles) src->set(q, (unsigned)__float_as_int(r[q][2].w));
            }
        };
        trip(0, std::integral_constant<int, AHEAD>());
        for (int k0 = GI2D_REDUCE_BATCH; k0 < ntiles; k0 += GI2D_REDUCE_BATCH) trip(k0, std::integral_constant<int, 0>());
    }
    // a run that does not lie inside the pool was never written (partial_row): the overflow status is up
    unsigned long long big = __ballot(mapped && ntiles > GI2D_BIG_TILES_F && pool >= 0 && pool + ntiles <= pool_rows);
    while (big) {  // a gaussian on > 32 tiles: the whole wave strides over its run of rows
        const int src = __ffsll((long long)big) - 1;
        big &= big - 1;
        const int bn = __shfl(ntiles, src, 64), bpool = __shfl(pool, src, 64);
        float part[11];
#pragma unroll
        for (int q = 0; q < 11; ++q) part[q] = 0.f;
        for (int t = lane; t < bn; t += 64) add_partial<GI2D_FAST_ROW>(part, partial_big, (size_t)bpool + t);
#pragma unroll
        for (int q = 0; q < 11; ++q) {
            float v = part[q];
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
            if (lane == src) acc[q] = v;
        }
    }
}


}  // namespace gi2d
