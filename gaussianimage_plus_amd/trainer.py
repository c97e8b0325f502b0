"""Native fitting loop: one GaussianImage++ training iteration = one C-ABI call (`gi2d_train_step`, four kernel
launches; csrc/gi2d_train.hip).  Mirrors `GaussianImage_Cholesky.train_iter` / `GaussianImage_Covariance.train_iter`
with L2 loss and torch.optim.Adam + StepLR (models/gaussianimage_cholesky.py:123-130,302-317;
models/gaussianimage_covariance.py:234-259), without the ~25 small PyTorch kernels and the two host syncs per
iteration of the reference loop.  Parameters live in ordinary torch tensors (`xyz`, `chol`, `feat`), so checkpoints,
densification or any other host logic can read and modify them between calls.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional

import torch

from . import _lib

_KINDS = {"cholesky": 0, "covariance": 1}


class _TrainState(C.Structure):
    """struct gi2d_train_state (include/gi2d.h), field for field."""
    _fields_ = [
        ("kind", C.c_int), ("num_points", C.c_int), ("img_height", C.c_int), ("img_width", C.c_int),
        ("clip_coe", C.c_float), ("radius_clip", C.c_float),
        ("xyz", C.c_void_p), ("chol", C.c_void_p), ("feat", C.c_void_p),
        ("opacity", C.c_void_p), ("bound", C.c_void_p),
        ("bound_stride", C.c_int), ("pad0", C.c_int),
        ("m_xyz", C.c_void_p), ("v_xyz", C.c_void_p), ("m_chol", C.c_void_p), ("v_chol", C.c_void_p),
        ("m_feat", C.c_void_p), ("v_feat", C.c_void_p),
        ("gt", C.c_void_p),
        ("xys", C.c_void_p), ("conics", C.c_void_p),
        ("radii", C.c_void_p), ("num_tiles_hit", C.c_void_p),
        ("out_img", C.c_void_p), ("tile_sse", C.c_void_p),
        ("status", C.c_void_p),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
        ("dbg_grads", C.c_void_p),
    ]


class NativeFitter:
    def __init__(self, gt_hwc: torch.Tensor, num_points: int, kind: str = "cholesky", lr: float = 1e-3,
                 betas=(0.9, 0.999), eps: float = 1e-8, lr_step: int = 20000, lr_gamma: float = 0.5,
                 seed: int = 3047, clip_coe: float = 3.0, radius_clip: float = 1.0,
                 init: Optional[dict] = None, debug_grads: bool = False):
        assert kind in _KINDS and gt_hwc.is_cuda and gt_hwc.dim() == 3 and gt_hwc.size(2) == 3
        self.lib = _lib.load()
        self.kind, self.dev = kind, gt_hwc.device
        self.h, self.w, self.n = int(gt_hwc.shape[0]), int(gt_hwc.shape[1]), int(num_points)
        self.tx, self.ty = (self.w + 15) // 16, (self.h + 15) // 16
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.lr_step, self.lr_gamma = int(lr_step), float(lr_gamma)
        self.iteration = 0
        n, h, w, dev = self.n, self.h, self.w, self.dev
        self.gt = gt_hwc.contiguous().float()
        if init is None:  # models/gaussianimage_cholesky.py:57-58,99 / gaussianimage_covariance.py:52-57
            g = torch.Generator(device="cpu").manual_seed(seed)
            if kind == "cholesky":
                xyz = torch.atanh(2 * (torch.rand(n, 2, generator=g) - 0.5))
            else:
                xyz = torch.rand(n, 2, generator=g) * torch.tensor([float(w), float(h)])
            init = {"xyz": xyz, "chol": torch.rand(n, 3, generator=g), "feat": torch.zeros(n, 3)}
        self.xyz = init["xyz"].detach().to(dev, torch.float32).contiguous().clone()
        self.chol = init["chol"].detach().to(dev, torch.float32).contiguous().clone()
        self.feat = init["feat"].detach().to(dev, torch.float32).contiguous().clone()
        self.opacity = init.get("opacity", torch.ones(n, 1)).detach().to(dev, torch.float32).contiguous().clone()
        low_pass = min(h * w / (9 * math.pi * n), 300)  # SLV bound, models/gaussianimage_cholesky.py:80-82
        self.bound = init.get("bound", torch.tensor([low_pass, 0.0, low_pass])).detach().to(dev, torch.float32).contiguous()
        f32 = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)
        i32 = lambda *s: torch.zeros(s, dtype=torch.int32, device=dev)
        self.m_xyz, self.v_xyz = f32(n, 2), f32(n, 2)
        self.m_chol, self.v_chol = f32(n, 3), f32(n, 3)
        self.m_feat, self.v_feat = f32(n, 3), f32(n, 3)
        self.xys, self.conics, self.radii, self.nth = f32(n, 2), f32(n, 3), i32(n), i32(n)
        self.out_img, self.tile_sse, self.status = f32(h, w, 3), f32(self.tx * self.ty), i32(4)
        self.dbg_grads = f32(n, 8) if debug_grads else None
        nbytes = self.lib.gi2d_fast_workspace_bytes(n, self.tx, self.ty)
        self.ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.call("gi2d_fast_workspace_init", self.ws.data_ptr(), nbytes, n, self.tx, self.ty,
                      torch.cuda.current_stream(dev).cuda_stream)
        p = lambda t: t.data_ptr()
        self.state = _TrainState(
            _KINDS[kind], n, h, w, float(clip_coe), float(radius_clip), p(self.xyz), p(self.chol), p(self.feat),
            p(self.opacity), p(self.bound), 3 if self.bound.numel() == 3 * n and n > 1 else 0, 0,
            p(self.m_xyz), p(self.v_xyz), p(self.m_chol), p(self.v_chol), p(self.m_feat), p(self.v_feat), p(self.gt),
            p(self.xys), p(self.conics), p(self.radii), p(self.nth), p(self.out_img), p(self.tile_sse), p(self.status),
            p(self.ws), nbytes, p(self.dbg_grads) if debug_grads else None)
        self._state_ref = C.byref(self.state)
        self._lr3 = (C.c_float * 3)()
        self._step_fn = self.lib.gi2d_train_step
        self._step_fn.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_int, C.c_void_p]
        self._step_fn.restype = C.c_int
        self._render_fn = self.lib.gi2d_train_render
        self._render_fn.argtypes = [C.c_void_p, C.c_void_p]
        self._render_fn.restype = C.c_int

    # ------------------------------------------------------------------
    def _check(self, rc, what):
        if rc != 0:
            raise _lib.Gi2dError(f"{what} failed (status {rc}): {self.lib.gi2d_last_error_string().decode()}")

    def current_lr(self) -> float:
        """StepLR(step_size=lr_step, gamma=lr_gamma), stepped once per iteration after the optimizer."""
        return self.lr * self.lr_gamma ** (self.iteration // self.lr_step)

    def train(self, iterations: int) -> None:
        """Run `iterations` training iterations (asynchronous: only kernel launches)."""
        st = torch.cuda.current_stream(self.dev).cuda_stream
        b1, b2 = self.betas
        state, lr3, fn = self._state_ref, self._lr3, self._step_fn
        with torch.cuda.device(self.dev):
            for _ in range(int(iterations)):
                lr = self.current_lr()
                lr3[0] = lr3[1] = lr3[2] = lr
                self.iteration += 1
                rc = fn(state, lr3, b1, b2, self.eps, self.iteration, st)
                if rc != 0:
                    self._check(rc, "gi2d_train_step")

    def render(self) -> torch.Tensor:
        """Rasterize the current parameters; returns clamp(out_img, 0, 1) as [H, W, 3]."""
        with torch.cuda.device(self.dev):
            self._check(self._render_fn(self._state_ref, torch.cuda.current_stream(self.dev).cuda_stream),
                        "gi2d_train_render")
        return self.out_img.clamp(0, 1)

    def last_step_psnr(self) -> float:
        """PSNR of the render made inside the last training step (from the per-tile squared errors)."""
        mse = float(self.tile_sse.sum().item()) / (3.0 * self.h * self.w)
        return 10 * math.log10(1.0 / max(mse, 1e-12))

    def psnr(self) -> float:
        mse = torch.nn.functional.mse_loss(self.render(), self.gt).item()
        return 10 * math.log10(1.0 / max(mse, 1e-12))

    def check_status(self):
        if int(self.status[1].item()):
            raise RuntimeError("a tile bucket overflowed (> 128 gaussians per (tile, id mod 4)); results invalid")
