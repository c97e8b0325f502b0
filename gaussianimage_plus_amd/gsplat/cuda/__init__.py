"""Native op table: what `import gsplat.cuda as _C` resolves to.

Same names, argument order and return arity as the reference's pybind table
(/root/reference/gsplat/gsplat/cuda/csrc/ext.cpp:16-66, signatures bindings.h), but every op
is a thin torch<->pointer shim over the gfx950 C ABI (include/gi2d.h): torch only provides the
HBM allocations and the current HIP stream.  Inputs must be device tensors and contiguous
(`CHECK_INPUT`, bindings.h:9-14 -> RuntimeError).  Nothing here computes on the CPU.

Where the reference's Python wrappers and its C++ bindings disagree the op
follows the WRAPPER's intent: rasterize_sum_forward returns 4 tensors (with the never-filled
cnt_gs_counts, bindings.cu:506-508) and rasterize_sum_backward returns 5 (with v_abs_xys).
"""
from __future__ import annotations

import os

import torch

from ... import _lib

_TILE = 16

# The compiled op table (csrc/torch_ext/gi2d_torch_ext.cpp, built by __graft_entry__.build(): the pybind module of
# ext.cpp:16-66 over the same C ABI).  The ctypes functions below are the fallback OF THE BINDING -- no C++ compiler at
# hand, GI2D_BINDING=ctypes, or a development variant of the library selected with GI2D_LIB (the compiled module is
# linked against the product library) -- never of the kernels.
_ext = None
if os.environ.get("GI2D_BINDING", "compiled") != "ctypes" and not os.environ.get("GI2D_LIB"):
    try:
        _lib.load()  # torch's HIP runtime and the (checked) product library first
        from ... import _gi2d_torch as _ext
    except ImportError:
        _ext = None
BINDING = "compiled" if _ext is not None else "ctypes"


def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def _chk(t: torch.Tensor, name: str, dtype=None) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise RuntimeError(f"{name} must be a tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f"{name} must be {dtype}, got {t.dtype}")
    return t


def _ptr(t):
    return None if t is None else t.data_ptr()


def _f32(*shape, like):
    return torch.empty(shape, dtype=torch.float32, device=like.device)


def _i32(*shape, like):
    return torch.empty(shape, dtype=torch.int32, device=like.device)


def _workspace(nbytes: int, like) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=like.device)


# ------------------------------------------------------------------------------- projection
def _project_fwd(cname, num_points, clip_coe, means2d, params, img_height, img_width, tile_bounds,
                 clip_thresh, radius_clip):
    _chk(means2d, "means2d", torch.float32)
    n = int(num_points)
    xys, depths, radii = _f32(n, 2, like=means2d), _f32(n, like=means2d), _i32(n, like=means2d)
    conics, nth = _f32(n, 3, like=means2d), _i32(n, like=means2d)
    with torch.cuda.device(means2d.device):
        _lib.call(cname, n, float(clip_coe), means2d.data_ptr(), *[p.data_ptr() for p in params],
                  int(img_height), int(img_width), int(tile_bounds[0]), int(tile_bounds[1]),
                  float(clip_thresh), float(radius_clip), xys.data_ptr(), depths.data_ptr(),
                  radii.data_ptr(), conics.data_ptr(), nth.data_ptr(), _stream(means2d))
    return xys, depths, radii, conics, nth


def project_gaussians_2d_forward(num_points, clip_coe, means2d, L_elements, img_height, img_width,
                                 tile_bounds, clip_thresh, radius_clip, isprint=False):
    """bindings.cu:1317-1381 -> (xys, depths, radii, conics, num_tiles_hit)"""
    _chk(L_elements, "L_elements", torch.float32)
    return _project_fwd("gi2d_project_gaussians_2d_forward", num_points, clip_coe, means2d, [L_elements],
                        img_height, img_width, tile_bounds, clip_thresh, radius_clip)


def project_gaussians_2d_covariance_forward(num_points, clip_coe, means2d, L_elements, img_height,
                                            img_width, tile_bounds, clip_thresh, radius_clip,
                                            isprint=False):
    """bindings.cu:1449-1513"""
    _chk(L_elements, "L_elements", torch.float32)
    return _project_fwd("gi2d_project_gaussians_2d_covariance_forward", num_points, clip_coe, means2d,
                        [L_elements], img_height, img_width, tile_bounds, clip_thresh, radius_clip)


def project_gaussians_2d_scale_rot_forward(num_points, clip_coe, means2d, scales2d, rotation, img_height,
                                           img_width, tile_bounds, clip_thresh, radius_clip,
                                           isprint=False):
    """bindings.cu:1384-1448"""
    _chk(scales2d, "scales2d", torch.float32)
    _chk(rotation, "rotation", torch.float32)
    return _project_fwd("gi2d_project_gaussians_2d_scale_rot_forward", num_points, clip_coe, means2d,
                        [scales2d, rotation], img_height, img_width, tile_bounds, clip_thresh, radius_clip)


def _project_bwd(cname, num_points, means2d, params, img_height, img_width, radii, conics, v_xy, v_depth,
                 v_conic, out_shapes):
    for t, nm in ((means2d, "means2d"), (conics, "conics"), (v_xy, "v_xy"), (v_conic, "v_conic")):
        _chk(t, nm, torch.float32)
    _chk(radii, "radii", torch.int32)
    n = int(num_points)
    v_cov2d, v_mean2d = _f32(n, 3, like=means2d), _f32(n, 2, like=means2d)
    outs = [_f32(*s, like=means2d) for s in out_shapes]
    with torch.cuda.device(means2d.device):
        _lib.call(cname, n, means2d.data_ptr(), *[p.data_ptr() for p in params], int(img_height),
                  int(img_width), radii.data_ptr(), conics.data_ptr(), v_xy.data_ptr(), _ptr(v_depth),
                  v_conic.data_ptr(), v_cov2d.data_ptr(), v_mean2d.data_ptr(),
                  *[o.data_ptr() for o in outs], _stream(means2d))
    return (v_cov2d, v_mean2d, *outs)


def project_gaussians_2d_backward(num_points, means2d, L_elements, img_height, img_width, radii, conics,
                                  v_xy, v_depth, v_conic):
    """bindings.cu:1517-1564 -> (v_cov2d, v_mean2d, v_L_elements)"""
    _chk(L_elements, "L_elements", torch.float32)
    return _project_bwd("gi2d_project_gaussians_2d_backward", num_points, means2d, [L_elements], img_height,
                        img_width, radii, conics, v_xy, v_depth, v_conic, [(int(num_points), 3)])


def project_gaussians_2d_covariance_backward(num_points, means2d, L_elements, img_height, img_width, radii,
                                             conics, v_xy, v_depth, v_conic):
    """bindings.cu:1565-1612"""
    _chk(L_elements, "L_elements", torch.float32)
    return _project_bwd("gi2d_project_gaussians_2d_covariance_backward", num_points, means2d, [L_elements],
                        img_height, img_width, radii, conics, v_xy, v_depth, v_conic, [(int(num_points), 3)])


def project_gaussians_2d_scale_rot_backward(num_points, means2d, scales2d, rotation, img_height, img_width,
                                            radii, conics, v_xy, v_depth, v_conic):
    """bindings.cu:1614-1668 -> (v_cov2d, v_mean2d, v_scale[N,2], v_rot[N,1])"""
    _chk(scales2d, "scales2d", torch.float32)
    _chk(rotation, "rotation", torch.float32)
    n = int(num_points)
    return _project_bwd("gi2d_project_gaussians_2d_scale_rot_backward", num_points, means2d,
                        [scales2d, rotation], img_height, img_width, radii, conics, v_xy, v_depth, v_conic,
                        [(n, 2), (n, 1)])


def compute_cov2d_bounds(num_pts, clip_coe, covs2d):
    """bindings.cu:44-63 -> (conics[N,3], radii[N,1])"""
    _chk(covs2d, "covs2d", torch.float32)
    n = int(num_pts)
    conics, radii = _f32(n, 3, like=covs2d), _f32(n, 1, like=covs2d)
    with torch.cuda.device(covs2d.device):
        _lib.call("gi2d_compute_cov2d_bounds", n, float(clip_coe), covs2d.data_ptr(), conics.data_ptr(),
                  radii.data_ptr(), _stream(covs2d))
    return conics, radii


compute_cov2d_bounds_xy = compute_cov2d_bounds  # ext.cpp:55 binds both names to the same function


# ------------------------------------------------------------------------------- binning
def cumsum_tiles_hit(num_tiles_hit):
    """Device-side replacement of torch.cumsum (utils.py:248) -> (cum i32[N], total i32[1] on device)."""
    _chk(num_tiles_hit, "num_tiles_hit", torch.int32)
    n = num_tiles_hit.numel()
    cum, total = _i32(n, like=num_tiles_hit), _i32(1, like=num_tiles_hit)
    with torch.cuda.device(num_tiles_hit.device):
        _lib.call("gi2d_cumsum_tiles_hit", n, num_tiles_hit.data_ptr(), cum.data_ptr(), total.data_ptr(),
                  _stream(num_tiles_hit))
    return cum, total


def map_gaussian_to_intersects(num_points, num_intersects, xys, depths, radii, cum_tiles_hit, tile_bounds,
                               radius_clip=1.0, isprint=False):
    """bindings.cu:283-365 -> (isect_ids i64[M], gaussian_ids i32[M])"""
    _chk(xys, "xys", torch.float32)
    _chk(depths, "depths", torch.float32)
    _chk(radii, "radii", torch.int32)
    _chk(cum_tiles_hit, "cum_tiles_hit", torch.int32)
    m = int(num_intersects)
    isect = torch.empty(m, dtype=torch.int64, device=xys.device)
    gids = _i32(m, like=xys)
    with torch.cuda.device(xys.device):
        _lib.call("gi2d_map_gaussian_to_intersects", int(num_points), m, xys.data_ptr(), depths.data_ptr(),
                  radii.data_ptr(), cum_tiles_hit.data_ptr(), int(tile_bounds[0]), int(tile_bounds[1]),
                  float(radius_clip), isect.data_ptr(), gids.data_ptr(), _stream(xys))
    return isect, gids


def sort_intersects(isect_ids, gaussian_ids, num_tiles, want_perm=False, want_inv_perm=False,
                    want_bins=False, want_keys=True):
    """Stable sort of (key, gaussian id) pairs by key: the native stand-in for torch.sort + torch.gather
    (utils.py:301-302).  -> dict(isect_ids_sorted, gaussian_ids_sorted, perm, inv_perm, tile_bins)"""
    _chk(isect_ids, "isect_ids", torch.int64)
    _chk(gaussian_ids, "gaussian_ids", torch.int32)
    m, t = isect_ids.numel(), int(num_tiles)
    dev = isect_ids
    keys = torch.empty(m, dtype=torch.int64, device=dev.device) if want_keys else None
    gids = _i32(m, like=dev)
    perm = _i32(m, like=dev) if want_perm else None
    inv = _i32(m, like=dev) if want_inv_perm else None
    bins = _i32(t, 2, like=dev) if want_bins else None
    nbytes = _lib.load().gi2d_sort_workspace_bytes(m, t)
    ws = _workspace(nbytes, dev)
    with torch.cuda.device(dev.device):
        _lib.call("gi2d_sort_intersects", m, t, isect_ids.data_ptr(), gaussian_ids.data_ptr(), _ptr(keys),
                  gids.data_ptr(), _ptr(perm), _ptr(inv), _ptr(bins), ws.data_ptr(), ws.numel(), _stream(dev))
    return dict(isect_ids_sorted=keys, gaussian_ids_sorted=gids, perm=perm, inv_perm=inv, tile_bins=bins,
                status=ws[:16].view(torch.int32))


def get_tile_bin_edges(num_intersects, isect_ids_sorted, rows=None):
    """bindings.cu:368-383 -> tile_bins i32[rows,2] (rows = num_intersects as in the reference)."""
    _chk(isect_ids_sorted, "isect_ids_sorted", torch.int64)
    m = int(num_intersects)
    rows = m if rows is None else int(rows)
    bins = _i32(rows, 2, like=isect_ids_sorted)
    with torch.cuda.device(isect_ids_sorted.device):
        _lib.call("gi2d_get_tile_bin_edges", m, isect_ids_sorted.data_ptr(), rows, bins.data_ptr(),
                  _stream(isect_ids_sorted))
    return bins


def bin_gaussians(xys, radii, tile_bounds, radius_clip, capacity):
    """Sync-free binning (include/gi2d.h gi2d_bin_gaussians): the native replacement of
    compute_cumulative_intersects + bin_and_sort_gaussians for depth == 0.
    -> (gaussian_ids_sorted i32[capacity], tile_bins i32[T,2], status i32[4] = {M, overflow, 0, 0})"""
    _chk(xys, "xys", torch.float32)
    _chk(radii, "radii", torch.int32)
    n, cap = xys.size(0), int(capacity)
    t = int(tile_bounds[0]) * int(tile_bounds[1])
    gids, bins, status = _i32(cap, like=xys), _i32(t, 2, like=xys), _i32(4, like=xys)
    ws = _workspace(_lib.load().gi2d_bin_workspace_bytes(cap, t), xys)
    with torch.cuda.device(xys.device):
        _lib.call("gi2d_bin_gaussians", n, cap, xys.data_ptr(), radii.data_ptr(), int(tile_bounds[0]),
                  int(tile_bounds[1]), float(radius_clip), gids.data_ptr(), bins.data_ptr(), status.data_ptr(),
                  ws.data_ptr(), ws.numel(), _stream(xys))
    return gids, bins, status


def rasterize_backward_fast(img_height, img_width, gaussian_ids_sorted, tile_bins, xys, radii, conics, colors,
                            opacities, final_idx, v_output, radius_clip, with_abs=False):
    """gi2d_rasterize_backward_tiles + gi2d_rasterize_backward_reduce (box form: the per-gaussian sum
    re-derives the tile box from xys/radii).  -> (v_xy, v_conic, v_colors, v_opacity[N,1], v_abs_xys|None)"""
    for t, nm in ((gaussian_ids_sorted, "gaussian_ids_sorted"), (tile_bins, "tile_bins"), (final_idx, "final_idx"),
                  (radii, "radii")):
        _chk(t, nm, torch.int32)
    for t, nm in ((xys, "xys"), (conics, "conics"), (colors, "colors"), (opacities, "opacities"),
                  (v_output, "v_output")):
        _chk(t, nm, torch.float32)
    n, cap = xys.size(0), gaussian_ids_sorted.numel()
    h, w = int(img_height), int(img_width)
    tx, ty = (w + _TILE - 1) // _TILE, (h + _TILE - 1) // _TILE
    partials = _f32(max(cap, 1), 12, like=xys)
    v_xy, v_conic = _f32(n, 2, like=xys), _f32(n, 3, like=xys)
    v_colors, v_opacity = _f32(n, 3, like=xys), _f32(n, 1, like=xys)
    v_abs = _f32(n, 4, like=xys) if with_abs else None
    with torch.cuda.device(xys.device):
        st = _stream(xys)
        _lib.call("gi2d_rasterize_backward_tiles", h, w, gaussian_ids_sorted.data_ptr(), tile_bins.data_ptr(),
                  tile_bins.size(0), xys.data_ptr(), conics.data_ptr(), colors.data_ptr(), opacities.data_ptr(),
                  final_idx.data_ptr(), v_output.data_ptr(), 1 if with_abs else 0, partials.data_ptr(), st)
        _lib.call("gi2d_rasterize_backward_reduce", n, xys.data_ptr(), radii.data_ptr(), tx, ty, float(radius_clip),
                  gaussian_ids_sorted.data_ptr(), tile_bins.data_ptr(), tile_bins.size(0), partials.data_ptr(),
                  v_xy.data_ptr(), v_conic.data_ptr(), v_colors.data_ptr(), v_opacity.data_ptr(), _ptr(v_abs), st)
    return v_xy, v_conic, v_colors, v_opacity, v_abs


# ------------------------------------------------------------------------------- fused fast path
def fast_tile_capacity() -> int:
    """Candidate gaussians a tile row of the fused fast path holds (include/gi2d.h gi2d_fast_tile_capacity)."""
    return int(_lib.load().gi2d_fast_tile_capacity())


class FastWorkspace:
    """A workspace of the fused fast path (include/gi2d.h "fused fast path"): allocated and initialised once,
    then reused by every forward/backward pair of the same problem shape.

    status (device int32[4]) = {any intersection, overflow of this pass, sticky overflow, fullest tile row seen above
    half the row capacity}.  The autograd wrappers do not wait for it on every forward (that would drain the GPU queue
    once per iteration): `post()` starts an asynchronous copy into pinned host memory behind the kernels just enqueued,
    `settle()` -- at the next forward or backward that uses the workspace -- reads what has arrived by then."""

    def __init__(self, num_points, tile_bounds, like):
        self.n, self.tx, self.ty = int(num_points), int(tile_bounds[0]), int(tile_bounds[1])
        nbytes = _lib.load().gi2d_fast_workspace_bytes(self.n, self.tx, self.ty)
        self.buf = torch.empty(nbytes, dtype=torch.uint8, device=like.device)
        self.status = torch.zeros(4, dtype=torch.int32, device=like.device)
        self.host = torch.zeros(4, dtype=torch.int32).pin_memory()
        self._event = torch.cuda.Event()
        self.event = None        # self._event once it is recorded behind the status copy of an unchecked forward
        self.fullest = None      # fullest tile row the last CHECKED pass saw (0: at most half the capacity); None: unknown
        self.reset(like)

    def reset(self, like=None):
        """Empty tile lists: at creation, and after an overflow (its lost entries leave lists and boxes inconsistent)."""
        t = self.buf if like is None else like
        with torch.cuda.device(t.device):
            if _ext is not None:
                _ext.fast_workspace_init(self.buf, self.n, self.tx, self.ty)
            else:
                _lib.call("gi2d_fast_workspace_init", self.buf.data_ptr(), self.buf.numel(), self.n, self.tx, self.ty,
                          _stream(t))
            self.status.zero_()  # incl. the sticky word: an overflow that has been dealt with is not reported again
        self.event, self.fullest = None, None

    def post(self):
        """Asynchronous copy of the status words behind everything enqueued so far on the current stream."""
        with torch.cuda.device(self.buf.device):
            self.host.copy_(self.status, non_blocking=True)
            self._event.record()
            self.event = self._event

    def settle(self):
        """(any_hit, overflow) of the last forward whose status has not been looked at yet, or None if there is none.
        Waits for the copy only if it has not landed yet (by the time the host comes back -- at the backward, or at the
        next iteration's forward -- it has)."""
        if self.event is None:
            return None
        self.event.synchronize()
        self.event = None
        any_hit, overflow, _, fullest = self.host.tolist()
        self.fullest = fullest
        return any_hit, overflow

    def read_now(self):
        """Blocking read of the status of the pass just enqueued."""
        any_hit, overflow, _, fullest = self.status.tolist()
        self.event, self.fullest = None, fullest
        return any_hit, overflow

    @property
    def must_check_now(self) -> bool:
        """A pass on a workspace whose fullest tile row is unknown (first use, or after a reset) or was above half the
        row capacity is checked before its result is handed on; otherwise the check trails by one call."""
        return self.fullest is None or self.fullest > 0


def fast_forward(ws, xys, radii, conics, colors, opacities, img_height, img_width, radius_clip, background=None):
    """gi2d_fast_bin (binning step + records on the workspace's persistent lists) + gi2d_fast_rasterize_forward
    -> out_img[H,W,3]; ws.status = {any intersection, overflow, sticky overflow, fullest row above half capacity}.
    With `background` the device itself writes the background image when not a single gaussian lands
    (rasterize_sum_plus.py:110-118)."""
    if _ext is not None:
        return _ext.fast_forward(ws.buf, ws.status, ws.n, ws.tx, ws.ty, xys, radii, conics, colors, opacities,
                                 int(img_height), int(img_width), float(radius_clip), background)
    if background is not None:
        _chk(background, "background", torch.float32)
    for t, nm in ((xys, "xys"), (conics, "conics"), (colors, "colors"), (opacities, "opacities")):
        _chk(t, nm, torch.float32)
    _chk(radii, "radii", torch.int32)
    h, w = int(img_height), int(img_width)
    out_img = _f32(h, w, 3, like=xys)
    with torch.cuda.device(xys.device):
        st = _stream(xys)
        _lib.call("gi2d_fast_bin", ws.n, xys.data_ptr(), radii.data_ptr(), conics.data_ptr(), colors.data_ptr(),
                  opacities.data_ptr(), ws.tx, ws.ty, float(radius_clip), ws.buf.data_ptr(), ws.buf.numel(),
                  ws.status.data_ptr(), st)
        _lib.call("gi2d_fast_rasterize_forward", ws.n, ws.tx, ws.ty, w, h, _ptr(background), ws.buf.data_ptr(),
                  ws.buf.numel(), ws.status.data_ptr(), None, None, out_img.data_ptr(), st)
    return out_img


def fast_backward(ws, xys, radii, v_output, img_height, img_width, radius_clip, with_abs=False):
    """gi2d_fast_rasterize_backward_tiles + _reduce on the workspace the forward filled.
    -> (v_xy, v_conic, v_colors, v_opacity[N,1], v_abs_xys|None)"""
    if _ext is not None:
        return _ext.fast_backward(ws.buf, ws.n, ws.tx, ws.ty, v_output, int(img_height), int(img_width), bool(with_abs))
    _chk(xys, "xys", torch.float32)
    _chk(radii, "radii", torch.int32)
    _chk(v_output, "v_output", torch.float32)
    n = ws.n
    v_xy, v_conic = _f32(n, 2, like=xys), _f32(n, 3, like=xys)
    v_colors, v_opacity = _f32(n, 3, like=xys), _f32(n, 1, like=xys)
    v_abs = _f32(n, 4, like=xys) if with_abs else None
    with torch.cuda.device(xys.device):
        st = _stream(xys)
        _lib.call("gi2d_fast_rasterize_backward_tiles", n, ws.tx, ws.ty, int(img_width), int(img_height), None,
                  v_output.data_ptr(), 1 if with_abs else 0, ws.buf.data_ptr(), ws.buf.numel(), st)
        _lib.call("gi2d_fast_rasterize_backward_reduce", n, ws.tx, ws.ty, ws.buf.data_ptr(), ws.buf.numel(),
                  v_xy.data_ptr(), v_conic.data_ptr(), v_colors.data_ptr(), v_opacity.data_ptr(), _ptr(v_abs), st)
    return v_xy, v_conic, v_colors, v_opacity, v_abs


# ------------------------------------------------------------------------------- rasterizer
def _check_block(block):
    if int(block[0]) != _TILE or int(block[1]) != _TILE:
        raise RuntimeError(f"only {_TILE}x{_TILE} tiles are supported (csrc/config.h BLOCK_X/BLOCK_Y), got {block}")


def _raster_fwd(cname, tile_bounds, block, img_size, gaussian_ids_sorted, tile_bins, xys, conics, colors,
                opacities, background, num_intersects_dev=None):
    for t, nm in ((gaussian_ids_sorted, "gaussian_ids_sorted"), (tile_bins, "tile_bins")):
        _chk(t, nm, torch.int32)
    for t, nm in ((xys, "xys"), (conics, "conics"), (colors, "colors"), (opacities, "opacities"),
                  (background, "background")):
        _chk(t, nm, torch.float32)
    _check_block(block)
    if colors.dim() != 2 or colors.size(1) != 3:
        raise RuntimeError("colors must have dimensions (num_points, 3)")
    w, h = int(img_size[0]), int(img_size[1])
    out_img, final_Ts, final_idx = _f32(h, w, 3, like=xys), _f32(h, w, like=xys), _i32(h, w, like=xys)
    with torch.cuda.device(xys.device):
        _lib.call(cname, int(tile_bounds[0]), int(tile_bounds[1]), w, h, gaussian_ids_sorted.data_ptr(),
                  tile_bins.data_ptr(), tile_bins.size(0), xys.data_ptr(), conics.data_ptr(),
                  colors.data_ptr(), opacities.data_ptr(), background.data_ptr(), _ptr(num_intersects_dev),
                  final_Ts.data_ptr(), final_idx.data_ptr(), out_img.data_ptr(), _stream(xys))
    return out_img, final_Ts, final_idx


def rasterize_sum_forward(tile_bounds, block, img_size, gaussian_ids_sorted, tile_bins, xys, conics, colors,
                          opacities, background, isprint=False, num_intersects_dev=None):
    """bindings.cu:453-526 (+ the 4th result rasterize_sum.py:157 unpacks)
    -> (out_img, final_Ts, final_idx, cnt_gs_counts)"""
    out_img, final_Ts, final_idx = _raster_fwd("gi2d_rasterize_sum_forward", tile_bounds, block, img_size,
                                               gaussian_ids_sorted, tile_bins, xys, conics, colors, opacities,
                                               background, num_intersects_dev)
    cnt_gs_counts = torch.zeros_like(final_idx)  # allocated, never filled (bindings.cu:506-508)
    return out_img, final_Ts, final_idx, cnt_gs_counts


def rasterize_sum_plus_forward(tile_bounds, block, img_size, gaussian_ids_sorted, tile_bins, xys, conics,
                               colors, opacities, background, isprint=False, num_intersects_dev=None):
    """bindings.cu:529-610 -> (out_img, final_Ts, final_idx)"""
    return _raster_fwd("gi2d_rasterize_sum_plus_forward", tile_bounds, block, img_size, gaussian_ids_sorted,
                       tile_bins, xys, conics, colors, opacities, background, num_intersects_dev)


def _raster_bwd(cname, with_abs, img_height, img_width, BLOCK_H, BLOCK_W, gaussian_ids_sorted, tile_bins, xys,
                conics, colors, opacities, final_idx, v_output, cum_tiles_hit, inv_perm):
    _chk(xys, "xys", torch.float32)
    _chk(colors, "colors", torch.float32)
    if xys.dim() != 2 or xys.size(1) != 2:
        raise RuntimeError("xys must have dimensions (num_points, 2)")  # bindings.cu:1193-1195
    if colors.dim() != 2 or colors.size(1) != 3:
        raise RuntimeError("colors must have 2 dimensions")  # bindings.cu:1197-1199
    _check_block((BLOCK_W, BLOCK_H))
    for t, nm in ((gaussian_ids_sorted, "gaussian_ids_sorted"), (tile_bins, "tile_bins"),
                  (final_idx, "final_idx")):
        _chk(t, nm, torch.int32)
    for t, nm in ((conics, "conics"), (opacities, "opacities"), (v_output, "v_output")):
        _chk(t, nm, torch.float32)
    n, m = xys.size(0), gaussian_ids_sorted.numel()
    v_xy, v_conic = _f32(n, 2, like=xys), _f32(n, 3, like=xys)
    v_colors, v_opacity = _f32(n, 3, like=xys), _f32(n, 1, like=xys)
    v_abs = _f32(n, 4, like=xys) if with_abs else None
    ws = _workspace(_lib.load().gi2d_rasterize_backward_workspace_bytes(n, m), xys)
    args = [n, m, int(img_height), int(img_width), gaussian_ids_sorted.data_ptr(), tile_bins.data_ptr(),
            tile_bins.size(0), xys.data_ptr(), conics.data_ptr(), colors.data_ptr(), opacities.data_ptr(),
            final_idx.data_ptr(), v_output.data_ptr(), _ptr(cum_tiles_hit), _ptr(inv_perm), v_xy.data_ptr(),
            v_conic.data_ptr(), v_colors.data_ptr(), v_opacity.data_ptr()]
    if with_abs:
        args.append(v_abs.data_ptr())
    with torch.cuda.device(xys.device):
        _lib.call(cname, *args, ws.data_ptr(), ws.numel(), _stream(xys))
    return v_xy, v_conic, v_colors, v_opacity, v_abs


def rasterize_sum_backward(img_height, img_width, BLOCK_H, BLOCK_W, gaussian_ids_sorted, tile_bins, xys,
                           conics, colors, opacities, background, final_Ts, final_idx, v_output,
                           v_output_alpha=None, cum_tiles_hit=None, inv_perm=None):
    """bindings.cu:1166-1240 (+ v_abs_xys, rasterize_sum.py:308)
    -> (v_xy, v_conic, v_colors, v_opacity[N,1], v_abs_xys[N,4])"""
    return _raster_bwd("gi2d_rasterize_sum_backward", True, img_height, img_width, BLOCK_H, BLOCK_W,
                       gaussian_ids_sorted, tile_bins, xys, conics, colors, opacities, final_idx, v_output,
                       cum_tiles_hit, inv_perm)


def rasterize_sum_plus_backward(img_height, img_width, BLOCK_H, BLOCK_W, gaussian_ids_sorted, tile_bins, xys,
                                conics, colors, opacities, background, final_Ts, final_idx, v_output,
                                v_output_alpha=None, cum_tiles_hit=None, inv_perm=None):
    """bindings.cu:1241-1314 -> (v_xy, v_conic, v_colors, v_opacity[N,1])"""
    return _raster_bwd("gi2d_rasterize_sum_plus_backward", False, img_height, img_width, BLOCK_H, BLOCK_W,
                       gaussian_ids_sorted, tile_bins, xys, conics, colors, opacities, final_idx, v_output,
                       cum_tiles_hit, inv_perm)[:4]


# ------------------------------------------------------------------------------- out of scope
def _unsupported(name):
    def f(*a, **k):
        raise NotImplementedError(f"gsplat.cuda.{name}: the 3D / N-channel paths are outside this build "
                                  "(SURVEY.md section 2, items 16-17)")
    f.__name__ = name
    return f


for _n in ("nd_rasterize_forward", "nd_rasterize_backward", "nd_rasterize_sum_forward",
           "nd_rasterize_sum_backward", "nd_rasterize_gs_sum_forward", "nd_rasterize_gs_sum_backward",
           "rasterize_forward", "rasterize_backward", "project_gaussians_forward",
           "project_gaussians_backward", "compute_sh_forward", "compute_sh_backward"):
    globals()[_n] = _unsupported(_n)


# ------------------------------------------------------------------------------- the compiled table takes over
_COMPILED_NAMES = ("project_gaussians_2d_forward", "project_gaussians_2d_backward",
                   "project_gaussians_2d_covariance_forward", "project_gaussians_2d_covariance_backward",
                   "project_gaussians_2d_scale_rot_forward", "project_gaussians_2d_scale_rot_backward",
                   "compute_cov2d_bounds", "compute_cov2d_bounds_xy", "map_gaussian_to_intersects", "get_tile_bin_edges",
                   "rasterize_sum_forward", "rasterize_sum_backward", "rasterize_sum_plus_forward",
                   "rasterize_sum_plus_backward")
CTYPES_TABLE = {n: globals()[n] for n in _COMPILED_NAMES}  # kept reachable: tests hold the two bindings to each other
if _ext is not None:
    for _n in _COMPILED_NAMES:
        globals()[_n] = getattr(_ext, _n)
