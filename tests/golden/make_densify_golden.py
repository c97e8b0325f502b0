#!/usr/bin/env python3
"""Dev-container only: drive the REFERENCE's own prune / growth code on CPU and commit inputs + outputs as a fixture
(tests/golden/densify_reference.npz).  What runs here, as written in the reference, imported and not copied:

  models/gaussianimage_covariance.py   GaussianImage_Covariance.__init__ (Adam groups), check_non_semi_definite,
                                       non_semi_definite_prune -> _prune_optimizer, densification_postfix ->
                                       cat_tensors_to_optimizer                                     (:261-382)
  train.py                             SimpleTrainer2d.add_sample_positions                          (:85-118)

The fixture pins tests/test_densify_cpu.py's statement of those lines and, on the GPU, gi2d_train_prune /
gi2d_train_grow (csrc/gi2d_densify.hip).  Nothing of the reference travels: the fixture is arrays.

Both files import, at module level, packages this image does not have (torchvision, constriction, pytorch_msssim, cv2,
wandb, vector_quantize_pytorch, jaxtyping behind the bundled gsplat) -- none of them is touched by the methods driven
below (plain torch: boolean masks, torch.cat, torch.topk, optimizer-state surgery), so the generator registers empty
placeholder modules under those names before the import.  SimpleTrainer2d.__init__ hard-codes cuda:0 and reads an
image file (train.py:39-40): the trainer object is made with object.__new__ and given exactly the attributes
add_sample_positions reads."""
import argparse
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


class _Anything(types.ModuleType):
    """A placeholder module: every attribute is another placeholder (never called by the code under test)."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        m = _Anything(self.__name__ + "." + name)
        setattr(self, name, m)
        return m


def _import_reference():
    names = ["torchvision", "torchvision.transforms", "constriction", "pytorch_msssim", "cv2", "wandb",
             "vector_quantize_pytorch", "gsplat", "gsplat.project_gaussians_2d_covariance", "gsplat.rasterize_sum_plus"]
    saved = {k: sys.modules.get(k) for k in names + ["utils", "quantize", "optimizer", "models", "train"]}
    for k in names:
        sys.modules[k] = _Anything(k)
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.path.insert(0, REF)
    try:
        import train  # noqa: E402  the reference's own file: pulls in utils, models.utils, models.gaussianimage_covariance
        from models.gaussianimage_covariance import GaussianImage_Covariance  # noqa: E402
    finally:
        sys.path.pop(0)
    return train, GaussianImage_Covariance, saved


class _Log:
    def write(self, *a, **k):
        pass


def _args(iterations, grow_iter, max_points):
    return argparse.Namespace(SLV_init=True, color_norm=False, coords_norm=False, coords_act="none",
                              iterations=iterations, clip_coe=3.0, radius_clip=1.0, cov_quant="lsq", color_quant="lsq",
                              xy_quant="lsq", xy_bit=12, cov_bit=10, color_bit=6, grow_iter=grow_iter,
                              max_num_points=max_points)


def _state(model):
    """Parameters, Adam moments and the per-point bound, as numpy copies."""
    out = {}
    for group in model.optimizer.param_groups:
        p = group["params"][0]
        st = model.optimizer.state[p]
        nm = group["name"]
        out[nm] = p.detach().numpy().copy()
        out["m_" + nm] = st["exp_avg"].numpy().copy()
        out["v_" + nm] = st["exp_avg_sq"].numpy().copy()
        out["step_" + nm] = np.float32(float(st["step"]))
    out["bound"] = model.cholesky_bound.numpy().copy()
    out["opacity"] = model._opacity.detach().numpy().copy()
    return out


def _put(out, prefix, state):
    for k, v in state.items():
        out[f"{prefix}_{k}"] = v


def main():
    train, Model, _ = _import_reference()
    out = {}
    h, w, n0, max_points = 96, 144, 3000, 5200
    iterations, grow_iter = 400, 40
    args = _args(iterations, grow_iter, max_points)
    torch.manual_seed(3047)
    model = Model(loss_type="L2", opt_type="adam", num_points=n0, H=h, W=w, BLOCK_H=16, BLOCK_W=16,
                  device=torch.device("cpu"), lr=0.018, quantize=False, args=args, logwriter=_Log())
    # a few optimizer steps on seeded gradients: non-zero moments in every row
    g = torch.Generator().manual_seed(11)
    for _ in range(3):
        for group in model.optimizer.param_groups:
            p = group["params"][0]
            p.grad = torch.randn(p.shape, generator=g) * 1e-2
        model.optimizer.step()
    # make a seeded set of rows non positive definite in (cov2d + bound): indefinite, negative diagonal, singular
    rng = np.random.default_rng(5)
    bad = np.sort(rng.choice(n0, 137, replace=False))
    with torch.no_grad():
        bound = model.cholesky_bound
        for j, i in enumerate(bad):
            kind = j % 3
            tgt = ([0.2, 5.0, 0.3], [-1.0, 0.0, -2.0], [1.0, 1.0, 1.0])[kind]  # det < 0; diag < 0; det == 0 exactly
            model._cov2d[i] = torch.tensor(tgt) - bound[i]
    out["dims"] = np.array([h, w, n0, max_points, iterations, grow_iter], np.int32)
    _put(out, "p0", _state(model))

    # ---- prune (models/gaussianimage_covariance.py:354-382)
    to_prune, valid = model.check_non_semi_definite()
    out["prune_valid_mask"] = valid.numpy().copy()
    pruned, n1 = model.non_semi_definite_prune(h, w)
    assert pruned == to_prune and n1 == n0 - pruned and pruned >= 100
    out["prune_counts"] = np.array([pruned, n1], np.int32)
    _put(out, "p1", _state(model))
    # a second check finds nothing
    assert model.non_semi_definite_prune(h, w)[0] == 0

    # ---- growth steps (train.py:85-118 + densification_postfix): an ordinary step (1000), a step the cap clips, the
    # last step (whole remaining budget)
    tr = object.__new__(train.SimpleTrainer2d)
    tr.device = torch.device("cpu")
    tr.args, tr.iterations, tr.max_num_points = args, iterations, max_points
    tr.H, tr.W, tr.gaussian_model, tr.logwriter = h, w, model, _Log()
    gi = torch.Generator().manual_seed(23)
    tr.gt_image = torch.rand(1, 3, h, w, generator=gi)
    out["gt"] = tr.gt_image[0].permute(1, 2, 0).contiguous().numpy().copy()  # [H,W,3]
    steps = [("g1", grow_iter, None), ("g2", 2 * grow_iter, 700), ("g3", iterations - grow_iter, None)]
    for tag, it, cap in steps:
        if cap is not None:  # clip this step: max_num_points - cur < 1000
            tr.max_num_points = model.cur_num_points + cap
        cur = model.cur_num_points
        k_want = max(0, tr.max_num_points - cur) if it == iterations - grow_iter else max(0, min(1000, tr.max_num_points - cur))
        # no topk ties (torch.topk's order among equal values is unspecified): draw renders until the k + 1 largest
        # normalised errors are distinct AND stay distinct without the normalisation (the port ranks the plain sums)
        for _ in range(200):
            render = torch.rand(1, 3, h, w, generator=gi)  # a clamped render lies in [0, 1]
            err = torch.abs(render - tr.gt_image).sum(dim=1)
            p = (err / err.sum()).view(-1)
            top = torch.topk(p, k_want + 1)
            if len(torch.unique(top.values)) == k_want + 1 and len(torch.unique(err.view(-1)[top.indices])) == k_want + 1:
                break
        else:
            raise SystemExit("no tie-free render found")
        seed = 1000 + it
        torch.manual_seed(seed)
        rand3 = torch.rand(k_want, 3)  # what add_sample_positions will draw (train.py:111)
        torch.manual_seed(seed)
        added = tr.add_sample_positions(render, iter=it)
        assert added == k_want
        out[f"{tag}_render"] = render[0].permute(1, 2, 0).contiguous().numpy().copy()
        out[f"{tag}_rand3"] = rand3.numpy().copy()
        out[f"{tag}_args"] = np.array([it, tr.max_num_points, cur, k_want, model.cur_num_points], np.int32)
        _put(out, tag, _state(model))
        tr.max_num_points = max_points
    assert model.cur_num_points <= max_points
    path = os.path.join(HERE, "densify_reference.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: out[k].tolist() for k in out if k.endswith("_args") or k == "prune_counts"},
          os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
