"""Core of the two additive-rasterizer autograd Functions (rasterize_sum.py / rasterize_sum_plus.py).

Reference orchestration (rasterize_sum_plus.py:98-172): cumsum -> `.item()` (host sync in the middle of
the forward) -> map -> torch.sort -> gather -> bin edges -> rasterize.  Here one forward is two native calls
on a pooled workspace of the fused fast path (csrc/gi2d_fast.hip):
    gi2d_fast_bin (binning step on the workspace's persistent tile lists)  ->  gi2d_fast_rasterize_forward
and one backward is gi2d_fast_rasterize_backward_tiles + _reduce on the same workspace: no float atomics,
bitwise reproducible.  "No intersection at all" gives the background image (rasterize_sum_plus.py:110-118) on the
device.  A tile row overflow (more than 1024 candidates in one tile) re-runs the forward on the capacity-free ops
(gi2d_bin_gaussians + plain rasterizer), so results are exact -- but the host does not wait for the status words of
every forward (rounds 1-3 did: one drain of the GPU queue per training iteration, the stall SURVEY 8f-1 exists to
remove).  A workspace is checked synchronously on its first use and whenever the fullest tile row of its last checked
pass was above half the row capacity; otherwise the 16 status bytes travel to pinned memory behind the kernels and are
read when the host next touches the workspace (the backward, or the next forward): by then they have arrived, and the
GPU never idles.  An overflow found that way -- a row going from <= 512 to > 1024 candidates between two consecutive
calls -- is repaired where the call it belongs to is still open: the BACKWARD of that forward re-runs the forward on the
capacity-free ops, writes the exact image into the tensor the forward handed out, and differentiates the exact lists
(_repair_late_overflow; the reference has no capacity and never raises on population, rasterize_sum_plus.py:98-172).
Only a forward whose graph is gone -- a no-grad render, found by the next call on its workspace -- can still only be
reported (see _settle); GI2D_WRAPPER_SYNC=1 restores the synchronous check.

No forward stays unchecked: a workspace with a posted status sits in `_pending` until it is settled -- by its own next
use, by ANY later forward / backward of the wrappers once its copy has landed (`_settle_landed`: an event query, no
wait), by `settle_all()` (blocking; launch.fit_image calls it behind its last evaluation render), or by the
interpreter-exit hook, which can only report.  And the "slowly moving rows" premise is only trusted for ONE scene: the
pool hands a workspace to whichever scene of its shape comes next (train.py's next image, a checkpoint just loaded, a
model after prune / growth), so a forward whose opacity tensor / shapes are not the ones the workspace's last forward
saw -- the opacity tensor's storage, the shapes -- is checked synchronously again (_scene_identity; a model's `_opacity`
keeps its storage from iteration to iteration and changes it exactly at those events).  No wall clock is involved: a host stall does not force a check, a fast hand-over does not skip one."""
from __future__ import annotations

import atexit
import os
import sys
import warnings
import weakref

import torch

from . import cuda as _C

BLOCK = 16
# GI2D_WRAPPER_SYNC=1: wait for the status words of every forward before its image is handed on (the behaviour of
# rounds 1-3: one GPU queue drain per iteration).  Default: see FastWorkspace.must_check_now / _settle.
SYNC_EVERY_FORWARD = os.environ.get("GI2D_WRAPPER_SYNC", "0") == "1"
_capacity = {}   # exact path: (device index, N, H, W) -> intersection capacity
_pool = {}       # fast path: (device index, N, tiles_x, tiles_y) -> idle FastWorkspace objects
_pending = []    # workspaces whose last forward posted its status words and has not been looked at yet
_captured = []   # workspaces a captured forward ran on (check_captured), pooled or still leased


def tile_bounds_of(img_height: int, img_width: int, block_h: int, block_w: int):
    return ((img_width + block_w - 1) // block_w, (img_height + block_h - 1) // block_h, 1)


class _Lease:
    """Holds a pooled workspace from the forward until the autograd graph that needs it is released."""

    def __init__(self, key, ws):
        self.key, self.ws = key, ws

    def __del__(self):
        try:
            _pool.setdefault(self.key, []).append(self.ws)
        except Exception:  # interpreter shutdown
            pass


def _acquire(xys, num_points, tile_bounds) -> _Lease:
    key = (xys.device.index, num_points, tile_bounds[0], tile_bounds[1])
    free = _pool.setdefault(key, [])
    ws = free.pop() if free else _C.FastWorkspace(num_points, tile_bounds, xys)
    return _Lease(key, ws)


def _scene_identity(colors, opacity):
    """What tells one scene from the next on a pooled workspace: the storage of the model's opacity tensor and the
    shapes of colours and opacity.  The opacity is a parameter / buffer of every reference model (`_opacity`: it keeps its
    storage across the iterations of a fit; a new image's model, prune and growth all allocate a new one), while the
    colours a model hands over may be computed afresh every iteration (sigmoid of the features, de-quantised features):
    their address says nothing.  A scene that changes without any of this changing (a checkpoint copied into the same
    parameters) is caught late at worst, and a late overflow is repaired (_repair_late_overflow)."""
    return (opacity.data_ptr(), tuple(opacity.shape), tuple(colors.shape))


def _settle(ws, what: str, repairable: bool = False) -> bool:
    """Look at the status words of the last forward on `ws` that has not been checked yet.  A tile row that overflowed
    there (more than 1024 candidate gaussians in one 16x16 tile, on a workspace whose fullest row was at most half that
    one call earlier) means an image has already been handed on that was rendered from a truncated tile list: the
    workspace is emptied and every later call on it is checked before its result is used (exact fallback).  Returns True
    for such an overflow when the caller can still repair it (`repairable`: the backward of that very forward);
    otherwise the caller is told by an exception -- the image is gone, nothing can be re-rendered into it."""
    if ws in _pending:
        _pending.remove(ws)
    st = ws.settle()
    if st is not None and st[1]:
        ws.reset()
        if repairable:
            return True
        raise RuntimeError(
            f"gsplat drop-in: a tile row overflowed (> {_C.fast_tile_capacity()} candidate gaussians in one tile) in {what}, "
            "whose status was read one call late; that call's image was rendered from a truncated tile list. The "
            "workspace has been emptied and checks every call from now on (exact fallback). Re-run the step, or set "
            "GI2D_WRAPPER_SYNC=1 to check every forward before its result is used.")
    return False


def _settle_landed(skip=None) -> None:
    """Look at every posted status whose copy has already landed (no waiting): a forward that nothing on its own
    workspace follows -- the last evaluation render of a loop -- is checked by whatever wrapper call comes next."""
    for ws in list(_pending):
        if ws is not skip and ws.event is not None and ws.event.query():
            _settle(ws, "an earlier forward (its workspace was not used again)")


def settle_all() -> None:
    """Blocking: look at the status words of every forward that has not been checked yet (raises as _settle does).  For
    callers whose last wrapper call is a forward: an evaluation render followed by nothing."""
    for ws in list(_pending):
        _settle(ws, "an earlier forward")


def _report_at_exit() -> None:  # the last resort: nothing can be raised to anyone any more
    try:
        settle_all()
    except RuntimeError as e:
        print(f"ERROR at interpreter exit: {e}", file=sys.stderr)
    except Exception:  # the device may already be gone
        pass


atexit.register(_report_at_exit)


def check_captured() -> None:
    """For loops replayed from a captured graph: look (synchronously) at the status words of every workspace a captured
    forward ran on -- pooled or still leased (a caller that keeps the captured loss or output alive, or uses
    retain_graph, holds the lease): `_captured` is a registry of its own.  A tile row that overflowed in some replay
    since the last look (more than 1024 candidate gaussians in one 16x16 tile) means images were rendered from
    truncated tile lists -- the sticky word remembers it -- and raises; the exact fallback of the eager path cannot run
    inside a graph, so such a scene needs the eager loop."""
    for ws in list(_captured):
        if not getattr(ws, "captured", False):
            _captured.remove(ws)
            continue
        if int(ws.status[2].item()):
            ws.captured = False
            _captured.remove(ws)
            ws.reset()
            raise RuntimeError(
                f"gsplat drop-in: a tile row overflowed (> {_C.fast_tile_capacity()} candidate gaussians in one "
                "tile) in a replay of a captured iteration; its images were rendered from truncated tile lists. "
                "Run this scene through the eager loop (which falls back to the capacity-free ops).")


def _exact_forward(plus, xys, radii, conics, colors, opacity, img_height, img_width, tile_bounds, block, img_size,
                   background, radius_clip, isprint):
    """Capacity-free ops (any tile population): gi2d_bin_gaussians + gi2d_rasterize_sum[_plus]_forward."""
    num_points = xys.size(0)
    num_tiles = tile_bounds[0] * tile_bounds[1]
    key = (xys.device.index, num_points, img_height, img_width)
    capacity = _capacity.get(key) or max(4 * num_points, num_tiles, 1024)
    fwd = _C.rasterize_sum_plus_forward if plus else _C.rasterize_sum_forward
    while True:
        gaussian_ids_sorted, tile_bins, status = _C.bin_gaussians(xys, radii, tile_bounds, radius_clip, capacity)
        res = fwd(tile_bounds, block, img_size, gaussian_ids_sorted, tile_bins, xys, conics, colors, opacity,
                  background, isprint, num_intersects_dev=status)
        num_intersects, overflow = status[:2].tolist()
        capacity = _capacity[key] = max(int(1.25 * num_intersects) + 1024, num_tiles)
        if not overflow:
            return res[0], res[2], gaussian_ids_sorted, tile_bins, num_intersects


def forward_impl(ctx, plus: bool, xys, depths, radii, conics, num_tiles_hit, colors, opacity, img_height,
                 img_width, BLOCK_H, BLOCK_W, background, radius_clip, isprint):
    num_points = xys.size(0)
    if BLOCK_H != BLOCK or BLOCK_W != BLOCK:
        raise RuntimeError(f"only {BLOCK}x{BLOCK} tiles are supported (csrc/config.h BLOCK_X/BLOCK_Y)")
    tile_bounds = tile_bounds_of(img_height, img_width, BLOCK_H, BLOCK_W)
    block = (BLOCK_W, BLOCK_H, 1)
    img_size = (img_width, img_height, 1)
    if not plus and colors.shape[-1] != 3:  # rasterize_sum.py:170-171 would pick nd_rasterize_sum_forward
        raise NotImplementedError("N-channel rasterization is outside this build (RGB only)")
    radii = radii if radii.dtype == torch.int32 else radii.to(torch.int32)

    lease = _acquire(xys, num_points, tile_bounds)
    ws = lease.ws
    # Inside a stream capture (torch.cuda.graph around a whole training iteration: launch.fit_image(graph=True)) nothing may
    # wait and no event may be queried: the pass is recorded as it is, and the status words of every REPLAY are the
    # caller's to look at -- check_captured() below, between replays.
    capturing = torch.cuda.is_current_stream_capturing()
    if not capturing:
        _settle(ws, "the previous forward on this workspace")
        _settle_landed(skip=ws)
        scene = _scene_identity(colors, opacity)
        if getattr(ws, "scene", scene) != scene:
            ws.fullest = None  # serves another scene now: checked before its image is handed on
        ws.scene = scene
    # "not a single intersection -> background image" (rasterize_sum_plus.py:110-118) is decided on the device
    out_img = _C.fast_forward(ws, xys, radii, conics, colors, opacity, img_height, img_width, radius_clip,
                              background=background)
    ctx.exact = None
    if capturing:
        ws.captured = True
        if ws not in _captured:
            _captured.append(ws)
    elif SYNC_EVERY_FORWARD or ws.must_check_now:
        _, overflow = ws.read_now()
        if overflow:
            ws.reset()  # the overflowing row lost entries: its workspace starts from empty lists next time
            out_img, final_idx, gids, bins, m = _exact_forward(plus, xys, radii, conics, colors, opacity, img_height,
                                                               img_width, tile_bounds, block, img_size, background,
                                                               radius_clip, isprint)
            if m < 1:  # no intersection at all: the exact ops wrote the background; nothing to differentiate
                gids = None
            ctx.exact = (gids, bins, final_idx)
            lease = None
    else:
        ws.post()  # looked at when the host next needs this workspace (the backward, or the next forward) ...
        _pending.append(ws)  # ... or by whichever wrapper call comes next once the copy has landed (_settle_landed)
    # rasterize_sum.py returns these; the plus wrapper drops them (final_T is never updated: forward.cu:558)
    final_Ts = None if plus else torch.ones(img_height, img_width, device=xys.device)
    cnt_gs_counts = None if plus else torch.zeros(img_height, img_width, dtype=torch.int32, device=xys.device)

    ctx.set_materialize_grads(False)  # the unused outputs' gradients as None, not as zero-filled tensors
    ctx.img_width, ctx.img_height = img_width, img_height
    ctx.BLOCK_H, ctx.BLOCK_W = BLOCK_H, BLOCK_W
    ctx.radius_clip = float(radius_clip)
    ctx.lease = lease
    ctx.plus, ctx.background = plus, background
    # (a weak reference: the autograd node must not keep its own output alive)
    ctx.out_ref = weakref.ref(out_img) if (lease is not None and not capturing) else None
    ctx.save_for_backward(xys, radii, conics, colors, opacity)
    return out_img, final_Ts, cnt_gs_counts


def _repair_late_overflow(ctx, xys, radii, conics, colors, opacity):
    """The forward this backward belongs to overflowed a tile row and was found one call late: run it again on the
    capacity-free ops (as the synchronous check of forward_impl would have), put the exact image into the tensor that
    was handed out, and leave the exact lists for the backward.  What cannot be repaired is a loss VALUE the caller has
    already formed from the truncated image -- the incoming gradient belongs to that image; for a loss that is linear in
    the image the result is exact, otherwise the step is one optimizer step off a truncated tile -- hence the warning."""
    tile_bounds = tile_bounds_of(ctx.img_height, ctx.img_width, ctx.BLOCK_H, ctx.BLOCK_W)
    out_img, final_idx, gids, bins, m = _exact_forward(
        ctx.plus, xys, radii, conics, colors, opacity, ctx.img_height, ctx.img_width, tile_bounds,
        (ctx.BLOCK_W, ctx.BLOCK_H, 1), (ctx.img_width, ctx.img_height, 1), ctx.background, ctx.radius_clip, False)
    handed = ctx.out_ref() if ctx.out_ref is not None else None
    if handed is not None:
        handed.data.copy_(out_img)
    ctx.exact = (gids if m >= 1 else None, bins, final_idx)
    ctx.lease = None
    warnings.warn(
        f"gsplat drop-in: a tile row overflowed (> {_C.fast_tile_capacity()} candidate gaussians in one tile) in this "
        "backward's forward, found one call late: the forward was re-run on the capacity-free ops, its image tensor now "
        "holds the exact render and the gradients are those of the exact tile lists. A loss value already computed from "
        "the first render saw a truncated tile. The workspace checks every call from now on; GI2D_WRAPPER_SYNC=1 checks "
        "every forward before its result is used.", RuntimeWarning, stacklevel=3)


def backward_impl(ctx, plus: bool, v_out_img):
    xys, radii, conics, colors, opacity = ctx.saved_tensors
    if v_out_img is None:  # (set_materialize_grads(False): the image took no part in the loss)
        v_out_img = torch.zeros(ctx.img_height, ctx.img_width, 3, device=xys.device)
    v_out_img = v_out_img.contiguous()
    if ctx.exact is None and not torch.cuda.is_current_stream_capturing():
        if _settle(ctx.lease.ws, "this backward's forward", repairable=True):
            _repair_late_overflow(ctx, xys, radii, conics, colors, opacity)
        _settle_landed(skip=ctx.lease.ws if ctx.lease is not None else None)
    if ctx.exact is not None:
        gids, bins, final_idx = ctx.exact
        if gids is None:  # rasterize_sum_plus.py:198-202: no intersection, zero gradients
            v_abs = None if plus else torch.zeros(xys.size(0), 4, device=xys.device)
            return (torch.zeros_like(xys), torch.zeros_like(conics), torch.zeros_like(colors),
                    torch.zeros_like(opacity), v_abs)
        v_xy, v_conic, v_colors, v_opacity, v_abs = _C.rasterize_backward_fast(
            ctx.img_height, ctx.img_width, gids, bins, xys, radii, conics, colors, opacity, final_idx, v_out_img,
            ctx.radius_clip, with_abs=not plus)
    else:
        # without a single intersection the tile pass finds empty rows and the per-gaussian sums are zeros
        v_xy, v_conic, v_colors, v_opacity, v_abs = _C.fast_backward(
            ctx.lease.ws, xys, radii, v_out_img, ctx.img_height, ctx.img_width, ctx.radius_clip, with_abs=not plus)
    return v_xy, v_conic, v_colors, v_opacity.view_as(opacity), v_abs
