"""GPU: the drop-in `gsplat` surface end to end (autograd Functions), including the legacy call
shapes the Cholesky / RS model files use, against the oracle run stage by stage on the same tensors."""
import numpy as np
import pytest
import torch

from helpers import check_close, synth_cholesky

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def gs():
    assert torch.cuda.is_available()
    import gaussianimage_plus_amd
    gaussianimage_plus_amd.install_as_gsplat()
    import gsplat
    return gsplat


def _stage_check(oracle, h, w, d_xys, d_depths, d_radii, d_conics, d_nth, col, op, img, v_img, grads, rclip=1.0):
    tb = oracle.tile_bounds(h, w)
    npts = d_xys.shape[0]
    m, cum = oracle.compute_cumulative_intersects(d_nth)
    _, _, so, go, bins = oracle.bin_and_sort_gaussians(npts, m, d_xys, d_depths, d_radii, cum, tb, rclip)
    out_o, fT, fidx, amb, absimg = oracle.rasterize_sum_forward(tb, (16, 16, 1), (w, h, 1), go, bins, d_xys, d_conics,
                                                                col, op, with_aux=True)
    ok = np.repeat((amb == 0)[..., None], 3, -1)
    check_close("wrapper out_img", img, out_o, absimg, mask=ok)
    want = oracle.rasterize_sum_backward(h, w, 16, 16, go, bins, d_xys, d_conics, col, op, None, fT, fidx, v_img,
                                         with_aux=True)
    okg = want[4] == 0
    check_close("wrapper v_colors", grads["colors"], want[2], want[5][:, 5:8], mask=np.repeat(okg[:, None], 3, 1),
                atol=1e-12)
    return want, okg


def test_cholesky_model_call_shape_end_to_end(gs, oracle):
    """models/gaussianimage_cholesky.py:206-221 verbatim call shapes (legacy 6-return projection +
    rasterize_gaussians_sum 3-return)."""
    from gsplat.project_gaussians_2d import project_gaussians_2d
    from gsplat.rasterize_sum import rasterize_gaussians_sum
    npts, h, w = 3000, 112, 176
    xyz, L, col, op = synth_cholesky(npts, h, w, 21)
    tb = oracle.tile_bounds(h, w)
    x_t = torch.from_numpy(xyz).to(DEV).requires_grad_(True)
    L_t = torch.from_numpy(L).to(DEV).requires_grad_(True)
    c_t = torch.from_numpy(col).to(DEV).requires_grad_(True)
    o_t = torch.from_numpy(op).to(DEV)
    screen0 = torch.zeros((npts, 4), dtype=torch.float32, requires_grad=True, device=DEV)
    screen = screen0 + 0
    xys, screen, depths, radii, conics, nth = project_gaussians_2d(x_t, screen, L_t, h, w, tb, isprint=False)
    out_img, per_pix, screen = rasterize_gaussians_sum(xys, screen, depths, radii, conics, nth, c_t, o_t, h, w, 16, 16,
                                                       background=torch.ones(3, device=DEV), return_alpha=False)
    assert out_img.shape == (h, w, 3) and per_pix.shape == (h, w) and per_pix.dtype == torch.int32
    xys.retain_grad()
    conics.retain_grad()
    v_img = torch.from_numpy(np.random.default_rng(0).normal(size=(h, w, 3)).astype(np.float32) * 1e-3).to(DEV)
    (out_img * v_img).sum().backward()
    want, okg = _stage_check(oracle, h, w, xys.detach().cpu().numpy(), depths.detach().cpu().numpy(),
                             radii.cpu().numpy(), conics.detach().cpu().numpy(), nth.cpu().numpy(), col, op,
                             out_img.detach().cpu().numpy(), v_img.cpu().numpy(),
                             {"colors": c_t.grad.cpu().numpy()})
    # gradient handed back for screenspace_points = v_abs_xys (rasterize_sum.py:308,328)
    assert screen0.grad is not None and screen0.grad.shape == (npts, 4)
    sg = screen0.grad.cpu().numpy()
    assert np.array_equal(sg[:, :2], xys.grad.cpu().numpy())
    check_close("v_abs_xy", sg[:, 2:], want[6][:, 2:], want[6][:, 2:], mask=np.repeat(okg[:, None], 2, 1), atol=1e-12)
    # projection backward (reference-faithful Cholesky VJP) on the rasterizer's own gradients
    pb = oracle.project_gaussians_2d_backward(npts, xyz, L, h, w, radii.cpu().numpy(), conics.detach().cpu().numpy(),
                                              xys.grad.cpu().numpy(), None, conics.grad.cpu().numpy())
    # same operations in the same order on the same numbers: bit for bit
    assert np.array_equal(L_t.grad.cpu().numpy(), pb[2]) and np.array_equal(x_t.grad.cpu().numpy(), pb[1])


def test_covariance_model_call_shape_end_to_end(gs, oracle):
    """models/gaussianimage_covariance.py:194-208: project_gaussians_2d_covariance + rasterize_gaussians_plus."""
    npts, h, w = 2500, 90, 130
    rng = np.random.default_rng(4)
    mean_px = (rng.random((npts, 2)) * np.array([w, h])).astype(np.float32)
    cov = (rng.random((npts, 3)) * np.array([1, 0.5, 1]) + np.array([3, -0.25, 3])).astype(np.float32)
    col = rng.random((npts, 3)).astype(np.float32)
    op = (0.3 + 0.7 * rng.random((npts, 1))).astype(np.float32)
    tb = oracle.tile_bounds(h, w)
    m_t = torch.from_numpy(mean_px).to(DEV).requires_grad_(True)
    c_t = torch.from_numpy(cov).to(DEV).requires_grad_(True)
    col_t = torch.from_numpy(col).to(DEV).requires_grad_(True)
    op_t = torch.from_numpy(op).to(DEV).requires_grad_(True)
    xys, depths, radii, conics, nth = gs.project_gaussians_2d_covariance(m_t, c_t, h, w, tb, coords_norm=False,
                                                                        clip_coe=2.5, radius_clip=2.0, isprint=False)
    img = gs.rasterize_gaussians_plus(xys, depths, radii, conics, nth, col_t, op_t, h, w, 16, 16,
                                      background=torch.ones(3, device=DEV), isprint=False, radius_clip=2.0)
    v_img = torch.from_numpy(rng.normal(size=(h, w, 3)).astype(np.float32) * 1e-3).to(DEV)
    (img * v_img).sum().backward()
    want, okg = _stage_check(oracle, h, w, xys.detach().cpu().numpy(), depths.detach().cpu().numpy(),
                             radii.cpu().numpy(), conics.detach().cpu().numpy(), nth.cpu().numpy(), col, op,
                             img.detach().cpu().numpy(), v_img.cpu().numpy(), {"colors": col_t.grad.cpu().numpy()},
                             rclip=2.0)
    check_close("v_opacity", op_t.grad.cpu().numpy(), want[3], want[5][:, 8:9], mask=okg[:, None], atol=1e-12)
    assert m_t.grad is not None and c_t.grad is not None and torch.isfinite(c_t.grad).all()


def test_zero_intersections_gives_background_and_zero_grads(gs):
    npts, h, w = 16, 40, 40
    tb = ((w + 15) // 16, (h + 15) // 16, 1)
    m_t = torch.full((npts, 2), 20.0, device=DEV, requires_grad=True)
    c_t = torch.zeros(npts, 3, device=DEV, requires_grad=True)  # det == 0 -> every gaussian culled
    col = torch.rand(npts, 3, device=DEV, requires_grad=True)
    op = torch.ones(npts, 1, device=DEV)
    bg = torch.tensor([0.1, 0.2, 0.3], device=DEV)
    xys, depths, radii, conics, nth = gs.project_gaussians_2d_covariance(m_t, c_t, h, w, tb)
    img = gs.rasterize_gaussians_plus(xys, depths, radii, conics, nth, col, op, h, w, background=bg)
    assert torch.equal(img, bg.expand(h, w, 3))
    img.sum().backward()
    assert float(col.grad.abs().max()) == 0.0
    # "not a single intersection" is noted by the binning step (a word stamped with its version: nothing ever resets
    # it) and acted on by the forward's own tile pass -- so the same pooled workspace must get it right in any order:
    # a scene with intersections, none again, and a scene again
    c_ok = torch.tensor([[9.0, 0.0, 9.0]], device=DEV).repeat(npts, 1)
    for cov, empty in ((c_ok, False), (c_t.detach(), True), (c_t.detach(), True), (c_ok, False), (c_t.detach(), True)):
        with torch.no_grad():
            xys, depths, radii, conics, nth = gs.project_gaussians_2d_covariance(m_t.detach(), cov, h, w, tb)
            img = gs.rasterize_gaussians_plus(xys, depths, radii, conics, nth, col.detach() + 0.5, op, h, w, background=bg)
        assert torch.equal(img, bg.expand(h, w, 3)) == empty, empty


def test_two_binning_calls_without_a_tile_pass_between_them_and_the_background(gs):
    """The C ABI allows gi2d_fast_bin twice on one workspace with no tile pass in between.  The first call has members,
    the second has none: the forward must render the background (rasterize_sum_plus.py:110-118), not an all-zero image
    from the first call's any-member word (csrc/gi2d_fast_internal.h: the word carries the binning call's own number
    next to the record-set version, which only a tile pass advances)."""
    from gaussianimage_plus_amd import _lib
    from gaussianimage_plus_amd.gsplat import cuda as table
    npts, h, w = 16, 40, 40
    tb = ((w + 15) // 16, (h + 15) // 16, 1)
    m_t = torch.full((npts, 2), 20.0, device=DEV)
    c_ok = torch.tensor([[9.0, 0.0, 9.0]], device=DEV).repeat(npts, 1)
    c_none = torch.zeros(npts, 3, device=DEV)  # det == 0 -> every gaussian culled
    col, op = torch.rand(npts, 3, device=DEV) + 0.5, torch.ones(npts, 1, device=DEV)
    bg = torch.tensor([0.1, 0.2, 0.3], device=DEV)
    ws = None
    st = torch.cuda.current_stream().cuda_stream
    for covs, empty in (((c_ok, c_none), True), ((c_none, c_ok), False), ((c_ok, c_none, c_none), True), ((c_ok, c_ok), False)):
        for cov in covs:
            xys, depths, radii, conics, nth = gs.project_gaussians_2d_covariance(m_t, cov, h, w, tb)
            ws = ws or table.FastWorkspace(npts, tb, xys)
            _lib.call("gi2d_fast_bin", npts, xys.data_ptr(), radii.data_ptr(), conics.data_ptr(), col.data_ptr(),
                      op.data_ptr(), tb[0], tb[1], 1.0, ws.buf.data_ptr(), ws.buf.numel(), ws.status.data_ptr(), st)
        out = torch.empty(h, w, 3, device=DEV)
        _lib.call("gi2d_fast_rasterize_forward", npts, tb[0], tb[1], w, h, bg.data_ptr(), ws.buf.data_ptr(),
                  ws.buf.numel(), ws.status.data_ptr(), None, None, out.data_ptr(), st)
        torch.cuda.synchronize()
        assert torch.equal(out, bg.expand(h, w, 3)) == empty, (len(covs), empty)
        assert bool(out.abs().sum() > 0)


def test_wrapper_errors(gs):
    with pytest.raises(ValueError):
        gs.rasterize_gaussians_plus(torch.zeros(4, 3, device=DEV), None, None, None, None, torch.zeros(4, 3, device=DEV),
                                    None, 16, 16)
    with pytest.raises(ValueError):
        gs.project_gaussians_2d_scale_rot(torch.zeros(0, 2, device=DEV), torch.zeros(0, 2, device=DEV),
                                          torch.zeros(0, 1, device=DEV), 16, 16, (1, 1, 1))
    with pytest.raises(AssertionError):
        gs.rasterize_gaussians_plus(torch.zeros(4, 2, device=DEV), None, None, None, None, torch.zeros(4, 3, device=DEV),
                                    None, 16, 16, background=torch.ones(4, device=DEV))
    with pytest.raises(NotImplementedError):
        gs.spherical_harmonics(3, None, None)


def test_bucket_overflow_falls_back_to_the_capacity_free_ops(gs, oracle):
    """700 gaussians in two tiles overflow the fast path's buckets; the wrapper must notice and redo the work on
    the exact ops, forward and backward, with the reference's 256-entry cap semantics intact."""
    g = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "case_chol_crowded.npz"))
    h, w, npts = int(g["h"]), int(g["w"]), int(g["n"])
    tb = oracle.tile_bounds(h, w)
    x_t = torch.from_numpy(g["in_means"]).to(DEV).requires_grad_(True)
    L_t = torch.from_numpy(g["in_L"]).to(DEV).requires_grad_(True)
    c_t = torch.from_numpy(g["colors"]).to(DEV).requires_grad_(True)
    o_t = torch.from_numpy(g["opacity"]).to(DEV)
    xys, depths, radii, conics, nth = gs.project_gaussians_2d(x_t, L_t, h, w, tb)
    img = gs.rasterize_gaussians_plus(xys, depths, radii, conics, nth, c_t, o_t, h, w, background=torch.ones(3, device=DEV))
    ok = np.repeat((g["pix_ambig"] == 0)[..., None], 3, -1)
    check_close("overflow out_img", img.detach().cpu().numpy(), g["out_img"], g["pix_abs"], mask=ok)  # measured: 0.12 of the 1e-5 bar
    (img * torch.from_numpy(g["v_out"]).to(DEV)).sum().backward()
    okg = np.repeat((g["g_ambig"] == 0)[:, None], 3, 1)
    check_close("overflow v_rgb", c_t.grad.cpu().numpy(), g["v_rgb"], g["g_abs9"][:, 5:8], mask=okg, atol=1e-12)


def test_no_grad_render_and_training_forward_can_interleave(gs, oracle):
    """A no-grad render between a training forward and its backward must not disturb the saved workspace."""
    npts, h, w = 1200, 64, 96
    xyz, L, col, op = synth_cholesky(npts, h, w, 31)
    tb = oracle.tile_bounds(h, w)
    t = lambda a, g=False: torch.from_numpy(a).to(DEV).requires_grad_(g)
    x_t, L_t, c_t, o_t = t(xyz), t(L), t(col, True), t(op)
    bg = torch.ones(3, device=DEV)

    def fwd(colors):
        xys, depths, radii, conics, nth = gs.project_gaussians_2d(x_t, L_t, h, w, tb)
        return gs.rasterize_gaussians_plus(xys, depths, radii, conics, nth, colors, o_t, h, w, background=bg)

    img = fwd(c_t)
    with torch.no_grad():
        other = fwd(torch.rand_like(c_t))  # different colours: would corrupt the packed records if shared
    img.sum().backward()
    g1 = c_t.grad.clone()
    c_t.grad = None
    fwd(c_t).sum().backward()
    assert torch.equal(g1, c_t.grad) and not torch.equal(other, img)


def _crowded_scene(npts, h, w, seed):
    """`npts` small gaussians whose centres all lie inside ONE 16x16 tile of an h x w image (cholesky parameters)."""
    rng = np.random.default_rng(seed)
    cx, cy = 24.0 + rng.uniform(-5, 5, npts), 24.0 + rng.uniform(-5, 5, npts)  # tile (1, 1), away from its edges
    xyz = np.stack([cx / (0.5 * w) - 1.0, cy / (0.5 * h) - 1.0], 1).astype(np.float32)
    L = np.stack([rng.uniform(0.25, 0.4, npts), rng.uniform(-0.05, 0.05, npts), rng.uniform(0.25, 0.4, npts)], 1)
    col = rng.uniform(0, 1, (npts, 3)).astype(np.float32) * 0.01
    return xyz, L.astype(np.float32), col, np.ones((npts, 1), np.float32)


def _render_and_grad(gs, oracle, xyz, L, col, op, h, w):
    tb = oracle.tile_bounds(h, w)
    t = lambda a, g=False: torch.from_numpy(a).to(DEV).requires_grad_(g)
    x_t, L_t, c_t, o_t = t(xyz), t(L), t(col, True), t(op)
    xys, depths, radii, conics, nth = gs.project_gaussians_2d(x_t, L_t, h, w, tb)
    img = gs.rasterize_gaussians_plus(xys, depths, radii, conics, nth, c_t, o_t, h, w, background=torch.ones(3, device=DEV))
    v = torch.from_numpy(np.random.default_rng(5).normal(size=(h, w, 3)).astype(np.float32) * 1e-3).to(DEV)
    (img * v).sum().backward()
    return img, c_t.grad, (xys, depths, radii, conics, nth), v


def test_a_tile_row_beyond_its_capacity_falls_back_to_the_capacity_free_ops(gs, oracle):
    """1300 gaussians in ONE tile: more candidates than a tile row of the fused fast path holds (1024).  A workspace's
    first forward is checked before its image is handed on, finds the overflow and redoes the work on the exact ops --
    with the reference's rule intact that only the 256 lowest ids of the tile are rasterized (forward.cu:553)."""
    from gaussianimage_plus_amd.gsplat import cuda as table
    npts, h, w = 1300, 48, 64
    assert npts > table.fast_tile_capacity()
    xyz, L, col, op = _crowded_scene(npts, h, w, 1)
    img, g_col, proj, v = _render_and_grad(gs, oracle, xyz, L, col, op, h, w)
    d = [p.detach().cpu().numpy() for p in proj]
    want, okg = _stage_check(oracle, h, w, d[0], d[1], d[2], d[3], d[4], col, op, img.detach().cpu().numpy(),
                             v.cpu().numpy(), {"colors": g_col.cpu().numpy()})
    assert float(g_col[256 + 50:].abs().max()) == 0.0  # ids beyond the cap received nothing


class _Model:
    """Parameters that keep their storage from call to call, as a model's nn.Parameters do across the iterations of a
    fit (what the wrappers tell one scene on a pooled workspace from the next by)."""

    def __init__(self, scene):
        t = lambda a, g=False: torch.from_numpy(a).to(DEV).requires_grad_(g)
        self.x, self.L, self.c, self.o = t(scene[0]), t(scene[1]), t(scene[2], True), t(scene[3])

    def load(self, scene):  # an optimizer step, however wild: the same tensors, new values
        with torch.no_grad():
            for dst, src in zip((self.x, self.L, self.c, self.o), scene):
                dst.copy_(torch.from_numpy(src))
        self.c.grad = None

    def render_and_grad(self, gs, oracle, h, w):
        tb = oracle.tile_bounds(h, w)
        xys, depths, radii, conics, nth = gs.project_gaussians_2d(self.x, self.L, h, w, tb)
        img = gs.rasterize_gaussians_plus(xys, depths, radii, conics, nth, self.c, self.o, h, w,
                                          background=torch.ones(3, device=DEV))
        v = torch.from_numpy(np.random.default_rng(5).normal(size=(h, w, 3)).astype(np.float32) * 1e-3).to(DEV)
        (img * v).sum().backward()
        return img, self.c.grad, (xys, depths, radii, conics, nth), v


def _calm_scene(npts, seed):
    rng = np.random.default_rng(seed)
    return (rng.uniform(-0.9, 0.9, (npts, 2)).astype(np.float32),
            np.stack([rng.uniform(0.3, 0.6, npts), np.zeros(npts), rng.uniform(0.3, 0.6, npts)], 1).astype(np.float32),
            rng.uniform(0, 1, (npts, 3)).astype(np.float32), np.ones((npts, 1), np.float32))


def test_an_overflow_found_one_call_late_is_repaired_by_the_backward(gs, oracle):
    """The status words of a forward are read when the host next touches the workspace (no GPU queue drain per
    iteration).  A tile row that goes from at most half its capacity to beyond it between two consecutive iterations of
    ONE model is therefore found late -- by the backward of that forward, which re-runs it on the capacity-free ops:
    the image tensor that was handed out then holds the exact render, the gradients are those of the exact lists (the
    reference never raises on population: rasterize_sum_plus.py:98-172), and a warning says what happened."""
    from gaussianimage_plus_amd.gsplat import _raster_common
    if _raster_common.SYNC_EVERY_FORWARD:
        pytest.skip("GI2D_WRAPPER_SYNC=1: every forward is checked before its image is used")
    npts, h, w = 1300, 48, 64
    model = _Model(_calm_scene(npts, 2))
    model.render_and_grad(gs, oracle, h, w)   # first use of the workspace: checked at once, rows far below half full
    model.c.grad = None
    model.render_and_grad(gs, oracle, h, w)   # from now on the check trails by one call
    crowded = _crowded_scene(npts, h, w, 3)
    model.load(crowded)
    with pytest.warns(RuntimeWarning, match="overflowed"):
        img, g_col, proj, v = model.render_and_grad(gs, oracle, h, w)
    for _ in range(2):  # the repaired call, then the same step again (the emptied workspace checks at once)
        d = [p.detach().cpu().numpy() for p in proj]
        _stage_check(oracle, h, w, d[0], d[1], d[2], d[3], d[4], crowded[2], crowded[3], img.detach().cpu().numpy(),
                     v.cpu().numpy(), {"colors": g_col.cpu().numpy()})
        assert float(g_col[256 + 50:].abs().max()) == 0.0  # the 256-lowest-ids rule (forward.cu:553) held
        model.c.grad = None
        img, g_col, proj, v = model.render_and_grad(gs, oracle, h, w)


def test_a_workspace_handed_to_another_scene_is_checked_at_once_again(gs, oracle):
    """The pool hands a workspace to whichever scene of its shape comes next (train.py's next image, a checkpoint just
    loaded, a model after prune / growth): "its rows were at most half full one call ago" says nothing about THAT scene.
    A forward whose colour / opacity tensors are not the ones the workspace last saw is checked before its image is
    handed on -- the crowded scene falls back to the capacity-free ops at once, however quickly it follows."""
    import warnings
    from gaussianimage_plus_amd.gsplat import _raster_common
    if _raster_common.SYNC_EVERY_FORWARD:
        pytest.skip("GI2D_WRAPPER_SYNC=1: every forward is checked before its image is used")
    npts, h, w = 1300, 48, 64
    first = _Model(_calm_scene(npts, 4))
    first.render_and_grad(gs, oracle, h, w)
    first.c.grad = None
    first.render_and_grad(gs, oracle, h, w)   # the check trails by one call from here on
    crowded = _crowded_scene(npts, h, w, 6)
    second = _Model(crowded)                   # (`first` stays alive: the second model's tensors are other storage)
    with warnings.catch_warnings():
        warnings.simplefilter("error")         # no late repair: checked at once, exact fallback
        img, g_col, proj, v = second.render_and_grad(gs, oracle, h, w)
    d = [p.detach().cpu().numpy() for p in proj]
    _stage_check(oracle, h, w, d[0], d[1], d[2], d[3], d[4], crowded[2], crowded[3], img.detach().cpu().numpy(),
                 v.cpu().numpy(), {"colors": g_col.cpu().numpy()})


def test_the_last_forward_of_a_loop_does_not_stay_unchecked(gs, oracle):
    """A no-grad render that nothing on its workspace follows (the evaluation render at the end of a fit) posts its
    status words like any other forward; `settle_all()` -- launch.fit_image calls it, and an interpreter-exit hook
    reports what is left -- looks at them.  Here that render overflows a tile row one call after a calm one; its graph
    does not exist, so nothing can be repaired: it is reported."""
    from gaussianimage_plus_amd.gsplat import _raster_common
    if _raster_common.SYNC_EVERY_FORWARD:
        pytest.skip("GI2D_WRAPPER_SYNC=1: every forward is checked before its image is used")
    npts, h, w = 1300, 48, 64
    model = _Model(_calm_scene(npts, 8))
    model.render_and_grad(gs, oracle, h, w)
    model.c.grad = None
    model.render_and_grad(gs, oracle, h, w)
    _raster_common.settle_all()  # nothing to report
    tb = oracle.tile_bounds(h, w)
    model.load(_crowded_scene(npts, h, w, 9))
    with torch.no_grad():
        xys, depths, radii, conics, nth = gs.project_gaussians_2d(model.x, model.L, h, w, tb)
        gs.rasterize_gaussians_plus(xys, depths, radii, conics, nth, model.c, model.o, h, w,
                                    background=torch.ones(3, device=DEV))
    with pytest.raises(RuntimeError, match="overflowed"):
        _raster_common.settle_all()
    _raster_common.settle_all()  # reported once


def test_compiled_and_ctypes_op_tables_agree_bit_for_bit(oracle):
    """csrc/torch_ext (the pybind module of ext.cpp:16-66) and the ctypes table are two bindings of ONE C ABI."""
    from gaussianimage_plus_amd.gsplat import cuda as table
    if table.BINDING != "compiled":
        pytest.skip("the compiled op table is not built (no C++ compiler?)")
    npts, h, w = 3000, 96, 160
    xyz, L, col, op = synth_cholesky(npts, h, w, 17)
    tb = oracle.tile_bounds(h, w)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    a = (npts, 3.0, t(xyz), t(L), h, w, tb, 0.01, 1.0, False)
    got, ref = table.project_gaussians_2d_forward(*a), table.CTYPES_TABLE["project_gaussians_2d_forward"](*a)
    assert len(got) == len(ref) == 5 and all(torch.equal(x, y) for x, y in zip(got, ref))
    xys, depths, radii, conics, nth = got
    v_xy, v_conic = torch.randn_like(xys), torch.randn_like(conics)
    b = (npts, t(xyz), t(L), h, w, radii, conics, v_xy, None, v_conic)
    got, ref = table.project_gaussians_2d_backward(*b), table.CTYPES_TABLE["project_gaussians_2d_backward"](*b)
    assert len(got) == len(ref) == 3 and all(torch.equal(x, y) for x, y in zip(got, ref))
    gids, bins, status = table.bin_gaussians(xys, radii, tb, 1.0, 8 * npts)
    gids = gids[:int(status[0])].contiguous()  # the reference's ops take exactly num_intersects entries
    bg = torch.ones(3, device=DEV)
    c = (tb, (16, 16, 1), (w, h, 1), gids, bins, xys, conics, t(col), t(op), bg, False)
    for name, n_out in (("rasterize_sum_forward", 4), ("rasterize_sum_plus_forward", 3)):
        got, ref = getattr(table, name)(*c), table.CTYPES_TABLE[name](*c)
        assert len(got) == len(ref) == n_out and all(torch.equal(x, y) for x, y in zip(got, ref)), name
    out_img, final_Ts, final_idx = got
    v_out = torch.randn_like(out_img) * 1e-3
    d = (h, w, 16, 16, gids, bins, xys, conics, t(col), t(op), bg, final_Ts, final_idx, v_out, None)
    for name, n_out in (("rasterize_sum_backward", 5), ("rasterize_sum_plus_backward", 4)):
        got, ref = getattr(table, name)(*d), table.CTYPES_TABLE[name](*d)
        assert len(got) == len(ref) == n_out and all(torch.equal(x, y) for x, y in zip(got, ref)), name
    with pytest.raises(RuntimeError):
        table.project_gaussians_2d_forward(npts, 3.0, t(xyz).cpu(), t(L), h, w, tb, 0.01, 1.0, False)  # CHECK_INPUT


def test_an_iteration_replayed_from_a_hip_graph_follows_the_eager_loop():
    """launch.fit_image(graph=True): render + loss + backward + Adam step captured once (torch.cuda.graph) and replayed.
    The wrappers record their passes without the host-side status protocol; the optimizer is Adam with its step count
    on the device.  Same loop, same model: after 40 iterations the fit's error agrees with the eager loop's to 1e-3."""
    from gaussianimage_plus_amd import launch
    gt = launch.synthetic_image(96, 144, 8).to(DEV)
    eager = launch.fit_image(gt, 1500, 40, eval_renders=1)
    replay = launch.fit_image(gt, 1500, 40, eval_renders=1, graph=True)
    assert abs(replay["mse"] - eager["mse"]) <= 1e-3 * eager["mse"], (eager["mse"], replay["mse"])
    start = launch.fit_image(gt, 1500, 4, eval_renders=1)
    assert replay["mse"] < 0.98 * start["mse"]  # the replays did train


def test_a_tile_row_overflow_inside_a_replay_is_reported_by_check_captured(gs, oracle):
    """Inside a captured graph the exact fallback of the eager path cannot run; _raster_common.check_captured(), called
    between replays, reads the sticky status word of every workspace a captured forward used and raises."""
    from gaussianimage_plus_amd.gsplat import _raster_common
    npts, h, w = 1300, 48, 64
    tb = oracle.tile_bounds(h, w)
    rng = np.random.default_rng(2)
    calm_xyz = rng.uniform(-0.9, 0.9, (npts, 2)).astype(np.float32)
    L = np.stack([rng.uniform(0.3, 0.6, npts), np.zeros(npts), rng.uniform(0.3, 0.6, npts)], 1).astype(np.float32)
    x_t = torch.from_numpy(calm_xyz).to(DEV)
    L_t = torch.from_numpy(L).to(DEV)
    c_t = torch.rand(npts, 3, device=DEV)
    o_t = torch.ones(npts, 1, device=DEV)
    bg = torch.ones(3, device=DEV)

    def render():
        xys, depths, radii, conics, nth = gs.project_gaussians_2d(x_t, L_t, h, w, tb)
        return gs.rasterize_gaussians_plus(xys, depths, radii, conics, nth, c_t, o_t, h, w, background=bg)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), torch.no_grad():
        for _ in range(3):
            eager = render()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.graph(graph):
        img = render()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(img, eager)
    _raster_common.check_captured()  # nothing to report
    x_t.copy_(torch.from_numpy(_crowded_scene(npts, h, w, 3)[0]).to(DEV))  # 1300 centres into one tile, in place
    graph.replay()
    with pytest.raises(RuntimeError, match="overflowed"):
        _raster_common.check_captured()
    _raster_common.check_captured()  # reported once; the workspace was emptied
