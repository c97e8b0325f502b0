"""Persistent-buffer driver of the hot path over the C ABI (include/gi2d.h).

`HotPath` owns every HBM buffer one image's fitting loop needs (sized once: gaussians N, image HxW) and
issues the C-ABI calls with pre-built argument lists, so one step costs four native calls of host work and
nothing else: no allocation, no host read-back, nothing but kernel launches on the current HIP stream
(so a step can be captured into a hipGraph: `capture_graph()`).

  mode "fused" (default) -- the fast path (csrc/gi2d_fast.hip).  step() in a loop is TWO launches:
      gi2d_fast_rasterize_forward_backward (one tile pass for both directions)
      -> gi2d_fast_reduce_project_backward_project_bin (ends this step AND projects + bins for the next one,
         from the inputs as they are at that moment -- the shape of a training loop, where the optimizer update
         sits in the same place: trainer.py);
    the first step after construction / set_inputs() is preceded by a stand-alone gi2d_fast_project_bin.
    (forward() / backward() called separately use gi2d_fast_rasterize_forward / _backward_tiles; step(pipelined=
    False) issues the three calls project+bin, tile pass, reduce+project backward.)
  mode "exact" -- the capacity-free ops (any tile population):
      gi2d_project_*_forward -> gi2d_bin_gaussians -> gi2d_rasterize_sum_forward
      -> gi2d_rasterize_backward_tiles -> gi2d_rasterize_backward_reduce -> gi2d_project_*_backward
`check_status()` raises if a fused step overflowed a tile row (and empties the workspace); `step_safe()` re-runs such a step in
"exact" mode.  Both modes produce the same numbers (tests/test_hotpath_gpu.py).

The gradient image of a step is either given (`set_v_out`) or -- `set_target(gt)` -- the L2-loss gradient of the
image the step itself renders, 2/(3HW) * (clamp(out,0,1) - gt) as in the reference's training loop
(models/gaussianimage_cholesky.py:302-317 with loss_type "L2"); the fused tile pass forms it per pixel in
registers, the exact mode with torch ops between its forward and backward calls.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib

_KINDS = {"cholesky": 0, "covariance": 1, "scale_rot": 2}
_PROJ = {
    "cholesky": ("gi2d_project_gaussians_2d_forward", "gi2d_project_gaussians_2d_backward"),
    "covariance": ("gi2d_project_gaussians_2d_covariance_forward", "gi2d_project_gaussians_2d_covariance_backward"),
    "scale_rot": ("gi2d_project_gaussians_2d_scale_rot_forward", "gi2d_project_gaussians_2d_scale_rot_backward"),
}


class HotPath:
    def __init__(self, num_points: int, height: int, width: int, device, kind: str = "cholesky",
                 mode: str = "fused", clip_coe: float = 3.0, radius_clip: float = 1.0):
        assert kind in _PROJ
        assert mode in ("fused", "exact")
        self.lib = _lib.load()
        self.n, self.h, self.w, self.kind, self.mode = int(num_points), int(height), int(width), kind, mode
        self.dev = torch.device(device)
        self.tx, self.ty = (self.w + 15) // 16, (self.h + 15) // 16
        self.T = self.tx * self.ty
        self.clip_coe, self.radius_clip = float(clip_coe), float(radius_clip)
        n, h, w, dev = self.n, self.h, self.w, self.dev
        f32 = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)
        i32 = lambda *s: torch.zeros(s, dtype=torch.int32, device=dev)
        self.means, self.params = f32(n, 2), f32(n, 2 if kind == "scale_rot" else 3)
        self.rot, self.v_rot = f32(n, 1), f32(n, 1)  # scale_rot only
        self.colors, self.opac = f32(n, 3), f32(n, 1)
        self.background = torch.ones(3, dtype=torch.float32, device=dev)
        self.xys, self.depths, self.radii, self.conics, self.nth = f32(n, 2), f32(n), i32(n), f32(n, 3), i32(n)
        self.status = i32(4)
        self.out_img, self.final_idx = f32(h, w, 3), i32(h, w)
        self.v_out = f32(h, w, 3)
        self.target = None            # set_target(): L2 loss against this image instead of a given v_out
        self.tile_sse = f32(self.T)
        self.grad_scale = 2.0 / (3.0 * h * w)
        self.v_xy, self.v_conic, self.v_rgb, self.v_opac = f32(n, 2), f32(n, 3), f32(n, 3), f32(n, 1)
        self.v_cov2d, self.v_mean2d, self.v_params = f32(n, 3), f32(n, 2), f32(n, 2 if kind == "scale_rot" else 3)
        # fused-path workspace (persistent tile lists, emptied once here)
        nbytes = self.lib.gi2d_fast_workspace_bytes(n, self.tx, self.ty)
        self.ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        self._stream_ptr = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            _lib.call("gi2d_fast_workspace_init", self.ws.data_ptr(), nbytes, n, self.tx, self.ty, self._stream_ptr)
        # exact-path buffers are created on first use
        self.capacity = 0
        self._exact_ready = False
        self._graph = None
        self._binned = False  # lists / records of the workspace are those of the current inputs' projection
        self._build_fused_calls()

    # ------------------------------------------------------------------ call lists
    def _build_fused_calls(self):
        p = lambda t: t.data_ptr()
        n, h, w, tx, ty, k = self.n, self.h, self.w, self.tx, self.ty, _KINDS[self.kind]
        L = self.lib
        ws, wsb = p(self.ws), self.ws.numel()
        rot = p(self.rot) if self.kind == "scale_rot" else None
        v_rot = p(self.v_rot) if self.kind == "scale_rot" else None
        self._f_bin = (L.gi2d_fast_project_bin, "fast project+bin",
                       [k, n, self.clip_coe, p(self.means), p(self.params), rot, p(self.colors), p(self.opac), h, w, tx, ty,
                        self.radius_clip, p(self.xys), p(self.depths), p(self.radii), p(self.conics), p(self.nth), ws, wsb,
                        p(self.status)])
        self._f_fwd = (L.gi2d_fast_rasterize_forward, "fast rasterize forward",
                       [n, tx, ty, w, h, None, ws, wsb, p(self.status), None, None, p(self.out_img)])
        tgt = p(self.target) if self.target is not None else None
        self._f_both = (L.gi2d_fast_rasterize_forward_backward, "fast rasterize forward+backward",
                        [n, tx, ty, w, h, None, None if tgt else p(self.v_out), tgt, self.grad_scale, p(self.tile_sse), ws,
                         wsb, p(self.status), p(self.out_img)])
        self._f_tiles = (L.gi2d_fast_rasterize_backward_tiles, "fast rasterize backward tiles",
                         [n, tx, ty, w, h, None, p(self.v_out), 0, ws, wsb])
        self._f_red_next = (L.gi2d_fast_reduce_project_backward_project_bin, "fast reduce+project backward + project+bin",
                            [k, n, self.clip_coe, p(self.means), p(self.params), rot, p(self.colors), p(self.opac), h, w,
                             p(self.xys), p(self.depths), p(self.radii), p(self.conics), p(self.nth), tx, ty,
                             self.radius_clip, ws, wsb, p(self.status), p(self.v_xy), p(self.v_conic), p(self.v_rgb),
                             p(self.v_opac), None, p(self.v_cov2d), p(self.v_mean2d), p(self.v_params), v_rot])
        self._f_red = (L.gi2d_fast_reduce_project_backward, "fast reduce+project backward",
                       [k, n, p(self.params), rot, h, w, p(self.xys), p(self.radii), p(self.conics), tx, ty,
                        self.radius_clip, ws, wsb, p(self.v_xy), p(self.v_conic), p(self.v_rgb), p(self.v_opac), None,
                        p(self.v_cov2d), p(self.v_mean2d), p(self.v_params), v_rot])

    def _build_exact_calls(self, capacity: int):
        p = lambda t: t.data_ptr()
        n, h, w, tx, ty = self.n, self.h, self.w, self.tx, self.ty
        L = self.lib
        dev = self.dev
        self.capacity = int(capacity)
        self.gids_sorted = torch.zeros(self.capacity, dtype=torch.int32, device=dev)
        self.partials = torch.zeros(self.capacity, 12, dtype=torch.float32, device=dev)
        self.tile_bins = torch.zeros(self.T, 2, dtype=torch.int32, device=dev)
        self.final_Ts = torch.zeros(h, w, dtype=torch.float32, device=dev)
        self.ws_bin = torch.zeros(L.gi2d_bin_workspace_bytes(self.capacity, self.T), dtype=torch.uint8, device=dev)
        fwd_name, bwd_name = _PROJ[self.kind]
        sr = self.kind == "scale_rot"
        self._e_fwd = [
            (getattr(L, fwd_name), "project forward",
             [n, self.clip_coe, p(self.means), p(self.params)] + ([p(self.rot)] if sr else []) +
             [h, w, tx, ty, 0.01, self.radius_clip, p(self.xys), p(self.depths), p(self.radii), p(self.conics),
              p(self.nth)]),
            (L.gi2d_bin_gaussians, "bin_gaussians",
             [n, self.capacity, p(self.xys), p(self.radii), tx, ty, self.radius_clip, p(self.gids_sorted),
              p(self.tile_bins), p(self.status), p(self.ws_bin), self.ws_bin.numel()]),
            (L.gi2d_rasterize_sum_forward, "rasterize forward",
             [tx, ty, w, h, p(self.gids_sorted), p(self.tile_bins), self.T, p(self.xys), p(self.conics),
              p(self.colors), p(self.opac), p(self.background), p(self.status), p(self.final_Ts),
              p(self.final_idx), p(self.out_img)]),
        ]
        self._e_tiles = (L.gi2d_rasterize_backward_tiles, "rasterize backward tiles",
                         [h, w, p(self.gids_sorted), p(self.tile_bins), self.T, p(self.xys), p(self.conics),
                          p(self.colors), p(self.opac), p(self.final_idx), p(self.v_out), 0, p(self.partials)])
        self._e_rest = [
            (L.gi2d_rasterize_backward_reduce, "rasterize backward reduce",
             [n, p(self.xys), p(self.radii), tx, ty, self.radius_clip, p(self.gids_sorted), p(self.tile_bins),
              self.T, p(self.partials), p(self.v_xy), p(self.v_conic), p(self.v_rgb), p(self.v_opac), None]),
            (getattr(L, bwd_name), "project backward",
             [n, p(self.means), p(self.params)] + ([p(self.rot)] if sr else []) +
             [h, w, p(self.radii), p(self.conics), p(self.v_xy), None, p(self.v_conic), p(self.v_cov2d),
              p(self.v_mean2d), p(self.v_params)] + ([p(self.v_rot)] if sr else [])),
        ]
        self._exact_ready = True

    def _run(self, call, stream):
        fn, what, args = call
        rc = fn(*args, stream)
        if rc != 0:
            raise _lib.Gi2dError(f"{what} failed (status {rc}): {self.lib.gi2d_last_error_string().decode()}")

    def _stream(self):
        return torch.cuda.current_stream(self.dev).cuda_stream

    # ------------------------------------------------------------------ inputs
    def set_inputs(self, means, params, colors, opac, rot=None):
        """means: tanh(xyz) in (-1,1) for "cholesky" / pixel coordinates otherwise; params: the activated
        Cholesky or covariance triple, or the [N,2] scales (+ rot [N,1]) for "scale_rot"; colors [N,3];
        opacity [N,1]."""
        pairs = [(self.means, means), (self.params, params), (self.colors, colors), (self.opac, opac)]
        if rot is not None:
            pairs.append((self.rot, rot))
        for dst, src in pairs:
            dst.copy_(torch.as_tensor(np.ascontiguousarray(src) if isinstance(src, np.ndarray) else src).to(self.dev))
        self._drop_bins()

    def _drop_bins(self):
        """The inputs changed: what was binned ahead for the old ones is not the next step's binning.  The tile lists in
        the workspace stay as they are -- the next project+bin call diffs every gaussian's tile box against them."""
        self._graph = None  # a captured step assumes the binning it left behind
        self._binned = False

    def _reset_workspace(self):
        """Empty tile lists (after an overflow, whose entries are lost, the lists no longer match the boxes)."""
        with torch.cuda.device(self.dev):
            _lib.call("gi2d_fast_workspace_init", self.ws.data_ptr(), self.ws.numel(), self.n, self.tx, self.ty,
                      self._stream())
        self._binned = False

    def set_v_out(self, v_out: torch.Tensor):
        """Use a fixed gradient image dL/d(out_img) [H,W,3]."""
        self.v_out.copy_(v_out)
        if self.target is not None:
            self.target = None
            self._build_fused_calls()

    def set_target(self, gt: torch.Tensor):
        """Use the L2 loss mean((clamp(out,0,1) - gt)^2) of each step's own render: gt [H,W,3] in [0,1]."""
        self.target = gt.to(self.dev, torch.float32).contiguous().clone()
        self._build_fused_calls()

    def _l2_grad_from_render(self):
        """v_out <- d mean((clamp(out,0,1)-gt)^2) / d out, tile_sse-equivalent loss in self.loss_value (torch ops)."""
        o = self.out_img
        d = o.clamp(0, 1) - self.target
        self.v_out.copy_(torch.where((o >= 0) & (o <= 1), self.grad_scale * d, torch.zeros_like(d)))

    def loss(self) -> float:
        """L2 loss of the last fused step with a target (sum of the per-tile squared errors / (3HW))."""
        return float(self.tile_sse.double().sum().item()) / (3.0 * self.h * self.w)

    # ------------------------------------------------------------------ the path
    def forward(self) -> torch.Tensor:
        st = self._stream()
        with torch.cuda.device(self.dev):
            if self.mode == "fused":
                if not self._binned:
                    self._run(self._f_bin, st)
                self._run(self._f_fwd, st)
                self._binned = True  # lists and records stay valid for these inputs (another pass may follow)
            else:
                self._exact_forward(st)
        return self.out_img

    def _exact_forward(self, st):
        if not self._exact_ready:
            self._build_exact_calls(max(4 * self.n, self.T, 1024))
        for c in self._e_fwd:
            self._run(c, st)
        m, overflow = self.status[:2].tolist()  # (re)size the lists; only on the exact path
        if overflow or self.capacity > 2 * max(m, self.T) + 4096:
            self._build_exact_calls(int(1.3 * m) + 1024)
            for c in self._e_fwd:
                self._run(c, st)

    def backward(self, timer=None, index: int = 0):
        st = self._stream()
        with torch.cuda.device(self.dev):
            if self.target is not None:
                self._l2_grad_from_render()
            if timer is not None:
                timer["bwd0"][index].record()
            self._run(self._f_tiles if self.mode == "fused" else self._e_tiles, st)
            if timer is not None:
                timer["bwd1"][index].record()
            if self.mode == "fused":
                self._run(self._f_red, st)
            else:
                for c in self._e_rest:
                    self._run(c, st)

    def step(self, timer=None, index: int = 0, pipelined: bool = True):
        st = self._stream()
        with torch.cuda.device(self.dev):
            if self.mode == "fused":
                if not self._binned:
                    self._run(self._f_bin, st)
                if timer is not None:  # start/stop events ride on the tile-pass dispatch itself
                    self.lib.gi2d_timer_arm(timer["rast"][index])
                self._run(self._f_both, st)
                if pipelined:  # finish this step and project + bin for the next one in the same launch
                    self._run(self._f_red_next, st)
                    self._binned = True
                else:
                    self._run(self._f_red, st)
                    self._binned = False
                return
            else:
                if not self._exact_ready:
                    self._exact_forward(st)
                self._run(self._e_fwd[0], st)
                self._run(self._e_fwd[1], st)
                if timer is not None:
                    timer["fwd0"][index].record()
                self._run(self._e_fwd[2], st)
                if timer is not None:
                    timer["fwd1"][index].record()
        self.backward(timer, index)

    def step_safe(self):
        """One step with the overflow check the fused path needs (host read-back of 8 bytes); not pipelined, so
        the flag of this step's tile pass is still there to read."""
        self.step(pipelined=False)
        if self.mode == "fused" and self.status[1].item():
            self._reset_workspace()
            self.mode = "exact"
            try:
                self.step()
            finally:
                self.mode = "fused"

    # ------------------------------------------------------------------ hipGraph
    def capture_graph(self):
        """Capture one step (all launches, no host work) into a hipGraph; replay() re-issues it."""
        assert self.mode == "fused"
        side = torch.cuda.Stream(device=self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):
            self.step()  # warm-up on the capture stream
        torch.cuda.current_stream(self.dev).wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            self.step()
        self._graph = g
        return g

    def replay(self):
        self._graph.replay()

    def tile_lists(self):
        """(ids, tile_bins) of the last fused tile pass: `ids` is an int32 view of the workspace, tile_bins[t] = [start,
        end) word positions of tile t's ascending id list in it (gi2d_fast_workspace_views; tests, debugging)."""
        import ctypes
        gp, bp = ctypes.c_void_p(), ctypes.c_void_p()
        _lib.call("gi2d_fast_workspace_views", self.ws.data_ptr(), self.ws.numel(), self.n, self.tx, self.ty,
                  ctypes.byref(gp), ctypes.byref(bp))
        base = self.ws.data_ptr()
        ids = self.ws[gp.value - base:].view(torch.int32)
        bins = self.ws[bp.value - base:bp.value - base + 8 * self.T].view(torch.int32).view(self.T, 2)
        return ids, bins

    # ------------------------------------------------------------------ bookkeeping for bench.py
    def num_intersects(self) -> int:
        """Intersection count M = sum(num_tiles_hit) of the last projection (a setup/teardown query)."""
        return int(self.nth.sum().item())

    def check_status(self):
        # status[2] is the sticky copy of the overflow flag: status[1] is reset by every project+bin, which in the
        # pipelined step runs BEHIND the tile pass that may have raised it
        m = self.num_intersects()
        now, sticky = self.status[1:3].tolist()
        self.status[2] = 0
        overflow = int(bool(now or sticky))
        if overflow:
            if self.mode == "fused":
                self._reset_workspace()  # the overflowed rows lost entries: start over from empty lists
            what = "a tile row" if self.mode == "fused" else f"the intersection capacity {self.capacity}"
            if self.mode == "fused" and (now | sticky) & 4:  # GI2D_STATUS_POOL (csrc/gi2d_fast_internal.h)
                what = "the row pool (gradient rows of gaussians on more than 32 tiles)"
            raise RuntimeError(f"{what} overflowed (M={m}); results of the last step are invalid")

    def kernel_timers(self, steps: int):
        mk = lambda: [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
        if self.mode == "fused":
            import ctypes
            handles = []
            for _ in range(steps):
                h = ctypes.c_void_p()
                _lib.call("gi2d_timer_create", ctypes.byref(h))
                handles.append(h)
            return {"rast": handles}
        return {"fwd0": mk(), "fwd1": mk(), "bwd0": mk(), "bwd1": mk()}

    def _timer_us(self, ev):
        import ctypes
        ts = []
        for h in ev["rast"]:
            us = ctypes.c_float()
            _lib.call("gi2d_timer_elapsed_us", h, ctypes.byref(us))
            ts.append(us.value)
        return float(np.mean(ts)), float(np.min(ts))

    def _avg_us(self, ev, a, b):
        ts = [x.elapsed_time(y) * 1e3 for x, y in zip(ev[a], ev[b])]
        return float(np.mean(ts)), float(np.min(ts))

    def dominant_kernel_stats(self, ev):
        """The dominant kernel of a step (HIP-event spans of this run) with its algorithmic bytes per launch
        (SURVEY 8d): forward 40*M + 20*H*W, backward tiles 40*M + 16*H*W + 36*N; the fused tile pass does both."""
        m = self.num_intersects()
        if self.mode == "fused":
            avg, mn = self._timer_us(ev)
            return {"name": "gi2d::fast_fwdbwd_kernel<%d>" % (1 if self.target is not None else 0),
                    "avg_us": avg, "min_us": mn, "bytes": 80 * m + 36 * self.h * self.w + 36 * self.n}
        f_avg, f_min = self._avg_us(ev, "fwd0", "fwd1")
        b_avg, b_min = self._avg_us(ev, "bwd0", "bwd1")
        if f_avg >= b_avg:
            return {"name": "gi2d::raster_fwd_kernel", "avg_us": f_avg, "min_us": f_min,
                    "bytes": 40 * m + 20 * self.h * self.w}
        return {"name": "gi2d::raster_bwd_kernel<false>", "avg_us": b_avg, "min_us": b_min,
                "bytes": 40 * m + 16 * self.h * self.w + 36 * self.n}

    def pair_stats(self, ev, pair_bytes):
        m = self.num_intersects()
        if self.mode == "fused":
            t_avg = self._timer_us(ev)[0]
            extra = {"fwdbwd_kernel_us": t_avg}
        else:
            f_avg, _ = self._avg_us(ev, "fwd0", "fwd1")
            b_avg, _ = self._avg_us(ev, "bwd0", "bwd1")
            t_avg = f_avg + b_avg
            extra = {"fwd_kernel_us": f_avg, "bwd_tile_kernel_us": b_avg}
        extra.update({"algorithmic_bytes": pair_bytes, "achieved_GBps": pair_bytes / (t_avg * 1e-6) / 1e9,
                      # every staged (tile, gaussian) entry against every pixel of its tile, forward + backward: the
                      # NOMINAL pair count of the reference's loops (forward.cu:650, backward.cu:1258); the kernels
                      # skip most of these pairs through their cull boxes, so this is not an evaluated-work figure
                      "nominal_pairs_per_s": 2 * 256.0 * m / (t_avg * 1e-6),
                      "note": "HIP start/stop events of the rasterizer kernel(s) inside the timed loop"})
        return extra

    def describe(self) -> str:
        if self.mode == "fused":
            return ("HotPath[fused]: 2 C-ABI calls/step in a loop (rasterize fwd+bwd tile pass; reduce+project bwd of this "
                    "step fused with project+bin of the next) on "
                    "persistent HBM buffers, eager launches on the current HIP stream")
        return f"HotPath[exact]: 6 C-ABI calls/step, intersection capacity {self.capacity}"
