"""GPU: the persistent tile lists of the fused fast path (csrc/gi2d_fast_internal.h).  A binning step appends a
gaussian only to tiles it has entered and the tile pass drops entries that left; whatever the inputs did between two
steps -- nothing, a small drift, a jump across the image, a collapse to radius 0 -- every step's lists, image and
gradients must equal those of the capacity-free ops, which bin from scratch (bit for bit), and the lists must equal
the oracle's cumsum + map + sort + bin edges."""
import numpy as np
import pytest
import torch

from helpers import check_close, synth_cholesky

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _lists(hp):
    ids, bins = hp.tile_lists()
    ids, bins = ids.cpu().numpy(), bins.cpu().numpy()
    return [ids[a:b].tolist() for a, b in bins]


def _oracle_lists(oracle, hp, n, h, w):
    tb = oracle.tile_bounds(h, w)
    nth = hp.nth.cpu().numpy()
    m, cum = oracle.compute_cumulative_intersects(nth)
    _, _, so, go, bins = oracle.bin_and_sort_gaussians(n, m, hp.xys.cpu().numpy(), np.zeros(n, np.float32),
                                                       hp.radii.cpu().numpy(), cum, tb, 1.0)
    return [go[a:b].tolist() for a, b in bins[:tb[0] * tb[1]]]


def _moves(rng, xyz, L, step):
    """One scripted change per step."""
    xyz, L = xyz.copy(), L.copy()
    kind = step % 6
    n = len(xyz)
    if kind == 1:    # everything drifts a little (a fit's usual step)
        xyz += rng.normal(size=xyz.shape).astype(np.float32) * 2e-3
    elif kind == 2:  # a tenth of the gaussians jump anywhere
        idx = rng.choice(n, n // 10, replace=False)
        xyz[idx] = (rng.random((len(idx), 2)) * 2 - 1).astype(np.float32) * 0.98
    elif kind == 3:  # some collapse (det == 0: culled), some swell over many tiles
        L[rng.choice(n, n // 20, replace=False)] = 0.0
        big = rng.choice(n, 5, replace=False)
        L[big] = [[14.0, 2.0, 11.0]]
    elif kind == 4:  # the collapsed ones come back, the big ones shrink
        dead = np.nonzero((L == 0).all(1))[0]
        L[dead] = (rng.random((len(dead), 3)) + np.array([0.7, 0, 0.7])).astype(np.float32)
        L[L[:, 0] > 10] = [[0.9, 0.1, 0.8]]
    elif kind == 5:  # a slide of the whole scene by a bit more than one tile
        xyz[:, 0] += 0.11
    return np.clip(xyz, -1.3, 1.3).astype(np.float32), L.astype(np.float32)


@pytest.mark.parametrize("n,h,w,pipelined", [(3000, 96, 160, True), (3000, 96, 160, False), (700, 50, 70, True)])
def test_lists_follow_any_change_of_the_inputs(oracle, n, h, w, pipelined):
    from gaussianimage_plus_amd.hotpath import HotPath
    rng = np.random.default_rng(11)
    xyz, L, col, op = synth_cholesky(n, h, w, 21)
    fused = HotPath(n, h, w, device=DEV, mode="fused")
    exact = HotPath(n, h, w, device=DEV, mode="exact")
    v = torch.from_numpy(rng.normal(size=(h, w, 3)).astype(np.float32) * 1e-3).to(DEV)
    for hp in (fused, exact):
        hp.set_v_out(v)
    for step in range(13):
        xyz, L = _moves(rng, xyz, L, step)
        for hp in (fused, exact):
            hp.set_inputs(xyz, L, col, op)
        # the pipelined call also bins AHEAD for the (unchanged) inputs: the next set_inputs must cope with that
        fused.step(pipelined=pipelined)
        exact.step()
        fused.check_status()
        for a, b in ((fused.out_img, exact.out_img), (fused.v_xy, exact.v_xy), (fused.v_conic, exact.v_conic),
                     (fused.v_rgb, exact.v_rgb), (fused.v_params, exact.v_params), (fused.v_mean2d, exact.v_mean2d)):
            assert torch.equal(a, b), f"step {step} (move {step % 6})"
        if not pipelined:  # xys / radii of the step are still there: compare the lists with the oracle's
            assert _lists(fused) == _oracle_lists(oracle, fused, n, h, w), f"step {step}"


def test_steady_state_issues_no_appends_and_keeps_rows_untouched():
    """Unchanged inputs: the second step neither appends nor rewrites (row headers and ids bitwise as before)."""
    from gaussianimage_plus_amd.hotpath import HotPath
    n, h, w = 5000, 128, 192
    xyz, L, col, op = synth_cholesky(n, h, w, 3)
    hp = HotPath(n, h, w, device=DEV, mode="fused")
    hp.set_inputs(xyz, L, col, op)
    hp.set_v_out(torch.zeros(h, w, 3, device=DEV))
    hp.step()
    hp.step()
    torch.cuda.synchronize()
    before = hp.ws.clone()
    ids0, bins0 = (t.clone() for t in hp.tile_lists())
    img0 = hp.out_img.clone()
    hp.step()
    torch.cuda.synchronize()
    ids1, bins1 = hp.tile_lists()
    assert torch.equal(bins0, bins1) and torch.equal(img0, hp.out_img)
    for t in range(hp.T):  # headers {count, sorted_len} and the ids of every row
        a, b = int(bins0[t, 0]), int(bins0[t, 1])
        assert torch.equal(ids0[a - 16:b], ids1[a - 16:b])
        assert int(ids1[a - 16]) == b - a and int(ids1[a - 15]) == b - a


def test_workspace_shared_by_unrelated_scenes(oracle):
    """The autograd wrappers pool workspaces by shape: consecutive forwards on one workspace with unrelated inputs."""
    import gaussianimage_plus_amd.gsplat.cuda as C
    n, h, w = 2000, 80, 112
    tb = oracle.tile_bounds(h, w)
    ws = None
    for seed in (1, 2, 3, 2):
        xyz, L, col, op = synth_cholesky(n, h, w, seed)
        t = lambda a: torch.from_numpy(a).to(DEV)
        xys, depths, radii, conics, nth = C.project_gaussians_2d_forward(n, 3.0, t(xyz), t(L), h, w, tb, 0.01, 1.0, False)
        ws = ws or C.FastWorkspace(n, tb, xys)
        out = C.fast_forward(ws, xys, radii, conics, t(col), t(op), h, w, 1.0)
        assert ws.status[:2].tolist() == [1, 0]
        gids, bins, st = C.bin_gaussians(xys, radii, tb, 1.0, int(nth.sum()) + 8)
        ref, _, _ = C.rasterize_sum_plus_forward(tb, (16, 16, 1), (w, h, 1), gids[:int(st[0])].contiguous(), bins, xys,
                                                 conics, t(col), t(op), torch.ones(3, device=DEV), False)
        assert torch.equal(out, ref), seed


# ------------------------------------------------------------------ the kernels of a single-image fit, against the oracle
def _fitter_lists(fit):
    """Tile rows of a NativeFitter's workspace after its last tile pass (gi2d_fast_workspace_views)."""
    import ctypes
    from gaussianimage_plus_amd import _lib
    gp, bp = ctypes.c_void_p(), ctypes.c_void_p()
    _lib.call("gi2d_fast_workspace_views", fit.ws.data_ptr(), fit.ws.numel(), fit.cap, fit.tx, fit.ty,
              ctypes.byref(gp), ctypes.byref(bp))
    torch.cuda.synchronize()
    base, tiles = fit.ws.data_ptr(), fit.tx * fit.ty
    ids = fit.ws[gp.value - base:].view(torch.int32).cpu().numpy()
    bins = fit.ws[bp.value - base:bp.value - base + 8 * tiles].view(torch.int32).view(tiles, 2).cpu().numpy()
    return [ids[a:b].tolist() for a, b in bins]


def _bench_like_fitter(**kw):
    """The headline workload: Cholesky model, 50 000 gaussians, 768x512, lr 1e-3 (bench.py)."""
    from gaussianimage_plus_amd.launch import synthetic_image
    from gaussianimage_plus_amd.trainer import NativeFitter
    n, h, w = 50000, 512, 768
    xyz, L, col, _ = synth_cholesky(n, h, w, 3047)
    lp = min(h * w / (9 * np.pi * n), 300)
    init = {"xyz": torch.from_numpy(np.arctanh(xyz.astype(np.float64)).astype(np.float32)),
            "chol": torch.from_numpy(L - np.array([lp, 0, lp], np.float32)), "feat": torch.from_numpy(col)}
    return NativeFitter(synthetic_image(h, w, 1).to(DEV), n, kind="cholesky", lr=1e-3, seed=3047, init=init, **kw)


def _wild(**kw):
    from test_batched_gpu import _wild_fitter
    return _wild_fitter(**kw)


@pytest.mark.parametrize("make,calls", [(_wild, (1, 2, 5)), (_bench_like_fitter, (1, 40))])
def test_single_image_fit_kernels_against_the_oracle(oracle, make, calls):
    """`gi2d_train_steps` on ONE image of at most 1 536 tiles runs `fast_fwdbwd_kernel<1, 0, true>` behind
    `train_reduce_update_kernel<.., true, .., true>` from its second iteration on: entering gaussians travel through the
    tiles' inboxes (csrc/gi2d_fast_internal.h::Inbox) -- the pair `bench.py`'s headline times.  After calls of 1, 2, 5
    (resp. 1, 40) iterations each, fitter A's state is held to the ORACLE directly:
      (i)   its tile rows == compute_cumulative_intersects + bin_and_sort_gaussians (the membership rule of
            forward.cu:161-166, stable order) on the xys / radii / num_tiles_hit its last binning step projected;
      (ii)  its last render == oracle.rasterize_sum_forward on those lists with the colours that iteration used;
      (iii) the gradients of that iteration w.r.t. the raw parameters == the oracle's rasterize backward -> projection
            backward -> tanh' chain, 1e-5 of the column maximum.
    The parameters an iteration STARTED from are not observable after the call (its update has run), so a twin fitter B
    takes the same steps one iteration behind and is read just before; A == B bit for bit is asserted when B catches up."""
    a, b = make(debug_grads=True), make(debug_grads=True)
    n, h, w = a.n, a.h, a.w
    tb = oracle.tile_bounds(h, w)
    gt = a.gt.cpu().numpy()
    done = 0
    for count in calls:
        if count > 1:
            b.train(count - 1)
        raw_xyz, raw_chol, feat = (t.cpu().numpy().copy() for t in (b.xyz, b.chol, b.feat))
        bound = b.bound.cpu().numpy()
        a.train(count)
        a.check_status()
        done += count
        xys, radii, nth = a.xys[:n].cpu().numpy(), a.radii[:n].cpu().numpy(), a.nth[:n].cpu().numpy()
        conics = a.conics[:n].cpu().numpy()
        # the projection belongs to the parameters B holds (tanh by numpy: a few ulp of the device's)
        want_xy = (0.5 * np.array([w, h]) * np.tanh(raw_xyz.astype(np.float64)) + 0.5 * np.array([w, h]))
        assert np.abs(xys - want_xy).max() < 1e-3, f"iteration {done}: projection of other parameters"
        # (i) lists
        m, cum = oracle.compute_cumulative_intersects(nth)
        _, _, so, go, bins = oracle.bin_and_sort_gaussians(n, m, xys, np.zeros(n, np.float32), radii, cum, tb, 1.0)
        want_lists = [go[s:e].tolist() for s, e in bins[:tb[0] * tb[1]]]
        got_lists = _fitter_lists(a)
        bad = [t for t in range(len(want_lists)) if got_lists[t] != want_lists[t]]
        assert not bad, f"iteration {done}: {len(bad)} tile rows differ from the oracle's, first tile {bad[0]}"
        # (ii) render
        op = np.ones((n, 1), np.float32)
        out_o, fT, fidx, amb, absimg = oracle.rasterize_sum_forward(tb, (16, 16, 1), (w, h, 1), go, bins, xys, conics,
                                                                    feat, op, with_aux=True)
        ok = np.repeat((amb == 0)[..., None], 3, -1)
        check_close(f"render of iteration {done}", a.out_img.cpu().numpy(), out_o, absimg, mask=ok)
        # (iii) gradients: L2 of the clamped render (models/gaussianimage_cholesky.py:307-310)
        oc = np.clip(out_o, 0.0, 1.0)
        v_out = np.where(oc == out_o, (2.0 / (3 * h * w)) * (oc - gt), 0.0).astype(np.float32)
        g = oracle.rasterize_sum_backward(h, w, 16, 16, go, bins, xys, conics, feat, op, None, fT, fidx, v_out,
                                          with_aux=True)
        L = (raw_chol + bound).astype(np.float32)
        means = np.tanh(raw_xyz).astype(np.float32)
        pb = oracle.project_gaussians_2d_backward(n, means, L, h, w, radii, conics, g[0], np.zeros(n, np.float32), g[1])
        v_mean2d, v_L = pb[1], pb[2]
        v_raw = v_mean2d.astype(np.float64) * (1.0 - np.tanh(raw_xyz.astype(np.float64)) ** 2)
        want = np.concatenate([v_raw, v_L, g[2]], 1)
        got = a.dbg_grads[:n].cpu().numpy().astype(np.float64)
        clear = g[4] == 0                                         # gaussians no cut-off pair touches
        scale = np.abs(want[clear]).max(axis=0, keepdims=True) + 1e-30
        err = (np.abs(got - want)[clear] / scale).max()
        print(f"[oracle] iteration {done}: {int(m)} intersections, {int((~clear).sum())} gaussians set aside, "
              f"gradient error {err:.2e} of the column maximum")
        assert err < 1e-5, f"iteration {done}: gradients off by {err:.3e}"
        b.train(1)
        for nm in ("xyz", "chol", "feat", "m_xyz", "v_chol"):
            assert torch.equal(getattr(a, nm), getattr(b, nm)), (done, nm)
