"""GPU: the fused fast path (gi2d_fast_*; two-kernel and single-pass tile forms) against the capacity-free ops and
the oracle."""
import numpy as np
import pytest
import torch

from helpers import check_close, rs_term_magnitudes, synth_cholesky, synth_gt

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _mk(mode, n, h, w, seed=3, kind="cholesky"):
    from gaussianimage_plus_amd.hotpath import HotPath
    xyz, L, col, op = synth_cholesky(n, h, w, seed)
    hp = HotPath(n, h, w, device=DEV, mode=mode, kind=kind)
    hp.set_inputs(xyz, L, col, op)
    return hp, (xyz, L, col, op)


def _v_out(h, w, seed=0):
    return torch.from_numpy(np.random.default_rng(seed).normal(size=(h, w, 3)).astype(np.float32) * 1e-3).to(DEV)


@pytest.mark.parametrize("n,h,w", [(6000, 256, 384), (900, 70, 100), (50000, 512, 768)])
def test_fused_equals_exact_bitwise_and_matches_oracle(oracle, n, h, w):
    fused, (xyz, L, col, op) = _mk("fused", n, h, w)
    exact, _ = _mk("exact", n, h, w)
    v = _v_out(h, w)
    outs = []
    for hp in (fused, exact):
        img = hp.forward().clone()
        hp.set_v_out(v)
        hp.backward()
        hp.check_status()
        outs.append([img] + [t.clone() for t in (hp.v_xy, hp.v_conic, hp.v_rgb, hp.v_opac, hp.v_mean2d, hp.v_params,
                                                 hp.xys, hp.conics, hp.radii, hp.nth)])
    assert fused.num_intersects() == exact.num_intersects()
    for a, b in zip(*outs):  # same per-pair arithmetic, same list order, same summation order
        assert torch.equal(a, b)
    # and against the oracle, stage by stage on the device's own projection
    tb = oracle.tile_bounds(h, w)
    d_xys, d_conics = fused.xys.cpu().numpy(), fused.conics.cpu().numpy()
    d_radii, d_nth = fused.radii.cpu().numpy(), fused.nth.cpu().numpy()
    m, cum = oracle.compute_cumulative_intersects(d_nth)
    assert m == fused.num_intersects()
    _, _, so, go, bins = oracle.bin_and_sort_gaussians(n, m, d_xys, np.zeros(n, np.float32), d_radii, cum, tb, 1.0)
    out_o, fT, fidx, amb, absimg = oracle.rasterize_sum_forward(tb, (16, 16, 1), (w, h, 1), go, bins, d_xys, d_conics,
                                                                col, op, with_aux=True)
    ok = np.repeat((amb == 0)[..., None], 3, -1)
    check_close("fused out_img", outs[0][0].cpu().numpy(), out_o, absimg, mask=ok)
    want = oracle.rasterize_sum_backward(h, w, 16, 16, go, bins, d_xys, d_conics, col, op, None, fT, fidx,
                                         v.cpu().numpy(), with_aux=True)
    okg = want[4] == 0
    for got, wv, sl, nm in ((outs[0][1], want[0], slice(0, 2), "v_xy"), (outs[0][2], want[1], slice(2, 5), "v_conic"),
                            (outs[0][3], want[2], slice(5, 8), "v_rgb"), (outs[0][4], want[3], slice(8, 9), "v_opacity")):
        g = got.cpu().numpy().reshape(wv.shape)
        check_close("fused " + nm, g, wv, want[5][:, sl], mask=np.repeat(okg[:, None], g.shape[1], 1), atol=1e-12)


def test_covariance_kind_fused_equals_exact():
    from gaussianimage_plus_amd.hotpath import HotPath
    n, h, w = 3000, 96, 144
    rng = np.random.default_rng(8)
    mean_px = (rng.random((n, 2)) * np.array([w, h])).astype(np.float32)
    cov = (rng.random((n, 3)) * np.array([1, 0.5, 1]) + np.array([3, -0.25, 3])).astype(np.float32)
    col = rng.random((n, 3)).astype(np.float32)
    op = (0.3 + 0.7 * rng.random((n, 1))).astype(np.float32)
    v = _v_out(h, w, 1)
    res = []
    for mode in ("fused", "exact"):
        hp = HotPath(n, h, w, device=DEV, mode=mode, kind="covariance", radius_clip=2.0, clip_coe=2.5)
        hp.set_inputs(mean_px, cov, col, op)
        img = hp.forward().clone()
        hp.set_v_out(v)
        hp.backward()
        hp.check_status()
        res.append((img, hp.v_mean2d.clone(), hp.v_params.clone(), hp.v_rgb.clone(), hp.v_opac.clone()))
    for a, b in zip(*res):
        assert torch.equal(a, b)


def test_bucket_overflow_is_flagged_and_step_safe_recovers(oracle):
    """> 1024 candidates in one tile row: status[1] is raised and step_safe() redoes the step
    on the capacity-free ops."""
    from gaussianimage_plus_amd.hotpath import HotPath
    n, h, w = 1400, 32, 48
    xyz, L, col, op = synth_cholesky(n, h, w, 4)
    xyz = (xyz * 0.12).astype(np.float32)  # everything into the two middle tiles
    fused = HotPath(n, h, w, device=DEV, mode="fused")
    exact = HotPath(n, h, w, device=DEV, mode="exact")
    v = _v_out(h, w, 2)
    for hp in (fused, exact):
        hp.set_inputs(xyz, L, col, op)
        hp.set_v_out(v)
    fused.step()
    with pytest.raises(RuntimeError):
        fused.check_status()
    fused.step_safe()
    exact.step()
    exact.check_status()
    assert torch.equal(fused.out_img, exact.out_img) and torch.equal(fused.v_params, exact.v_params)
    # the workspace was emptied: a later, non-overflowing problem on the same object still works
    xyz2, L2, col2, op2 = synth_cholesky(n, h, w, 5)
    fused.set_inputs(xyz2, L2, col2, op2)
    exact.set_inputs(xyz2, L2, col2, op2)
    fused.step()
    fused.check_status()
    exact.step()
    assert torch.equal(fused.out_img, exact.out_img) and torch.equal(fused.v_params, exact.v_params)


@pytest.mark.parametrize("n", [380, 900])
def test_more_than_256_per_tile_keeps_lowest_ids(oracle, n):
    """256 < population <= 1024 in a tile: the fused forward ranks all ids and rasterizes the 256 lowest
    (the stable-sort order the oracle uses); gradients of the others are exactly zero."""
    h, w = 16, 16
    rng = np.random.default_rng(12)
    xyz = ((rng.random((n, 2)) - 0.5) * 0.8).astype(np.float32)
    L = (rng.random((n, 3)) * np.array([1.5, 0.3, 1.5]) + np.array([0.6, 0, 0.6])).astype(np.float32)
    col = rng.random((n, 3)).astype(np.float32)
    op = np.ones((n, 1), np.float32)
    from gaussianimage_plus_amd.hotpath import HotPath
    hp = HotPath(n, h, w, device=DEV, mode="fused")
    hp.set_inputs(xyz, L, col, op)
    img = hp.forward().clone()
    hp.check_status()
    ref = oracle.render_cholesky(xyz, L, col, op, h, w, with_aux=True)
    assert ref["M"] > 256
    out_o, fT, fidx, amb, absimg = ref["ras"]
    # sums of up to 256 terms in a different fp32 order than the oracle's: measured 0.06 of the 1e-5 bar
    check_close("capped out_img", img.cpu().numpy(), out_o, absimg, mask=np.repeat((amb == 0)[..., None], 3, -1))
    hp.set_v_out(_v_out(h, w, 3))
    hp.backward()
    dropped = ref["gids_sorted"][256:]
    assert float(hp.v_rgb[torch.from_numpy(dropped).long().to(DEV)].abs().max()) == 0.0


def test_graph_replay_matches_eager():
    hp, _ = _mk("fused", 8000, 256, 384, seed=9)
    hp.set_v_out(_v_out(256, 384, 4))
    hp.step()
    torch.cuda.synchronize()
    want = [t.clone() for t in (hp.out_img, hp.v_params, hp.v_mean2d, hp.v_rgb)]
    hp.capture_graph()
    for t in (hp.out_img, hp.v_params, hp.v_mean2d, hp.v_rgb):
        t.zero_()
    hp.replay()
    hp.replay()
    torch.cuda.synchronize()
    hp.check_status()
    for a, b in zip(want, (hp.out_img, hp.v_params, hp.v_mean2d, hp.v_rgb)):
        assert torch.equal(a, b)


def test_scale_rot_kind_at_config5_size(oracle):
    """BASELINE.json config 5 geometry: rotation-scale model, 30 000 gaussians, 768x512 (activations as in
    models/gaussianimage_rs.py:167,172: scale = |s + 0.5|, rot = sigmoid(r) * 2 pi)."""
    from gaussianimage_plus_amd.hotpath import HotPath
    n, h, w = 30000, 512, 768
    rng = np.random.default_rng(30)
    mean_px = (rng.random((n, 2)) * np.array([w, h])).astype(np.float32)
    scales = np.abs(rng.random((n, 2)) + 0.5).astype(np.float32)
    rot = (1 / (1 + np.exp(-rng.random((n, 1)))) * 2 * np.pi).astype(np.float32)
    col = rng.random((n, 3)).astype(np.float32)
    op = np.ones((n, 1), np.float32)
    v = _v_out(h, w, 5)
    res = []
    for mode in ("fused", "exact"):
        hp = HotPath(n, h, w, device=DEV, mode=mode, kind="scale_rot")
        hp.set_inputs(mean_px, scales, col, op, rot=rot)
        img = hp.forward().clone()
        hp.set_v_out(v)
        hp.backward()
        hp.check_status()
        res.append((img, hp.v_mean2d.clone(), hp.v_params.clone(), hp.v_rot.clone(), hp.v_rgb.clone(), hp.xys.clone(),
                    hp.conics.clone(), hp.radii.clone(), hp.nth.clone(), hp.v_xy.clone(), hp.v_conic.clone()))
    for a, b in zip(*res):
        assert torch.equal(a, b)
    # against the oracle: projection, then rasterizer + projection backward on the device's projection
    tb = oracle.tile_bounds(h, w)
    po = oracle.project_gaussians_2d_scale_rot_forward(n, 3.0, mean_px, scales, rot, h, w, tb, 0.01, 1.0)
    d_xys, d_conics, d_radii, d_nth = [t.cpu().numpy() for t in (res[0][5], res[0][6], res[0][7], res[0][8])]
    same = (d_radii == po[2]) & (d_nth == po[4])
    print(f"scale-rot N={n}: radii / num_tiles_hit differ on {int((~same).sum())} gaussians (device sin/cos vs libm)")
    assert int((~same).sum()) <= n // 1000
    cs = np.abs(po[3][same]).max(axis=-1, keepdims=True)
    check_close("rs conics", d_conics[same], po[3][same], cs, rtol=1e-5)
    m, cum = oracle.compute_cumulative_intersects(d_nth)
    _, _, so, go, bins = oracle.bin_and_sort_gaussians(n, m, d_xys, np.zeros(n, np.float32), d_radii, cum, tb, 1.0)
    out_o, fT, fidx, amb, absimg = oracle.rasterize_sum_forward(tb, (16, 16, 1), (w, h, 1), go, bins, d_xys, d_conics,
                                                                col, op, with_aux=True)
    check_close("rs out_img", res[0][0].cpu().numpy(), out_o, absimg, mask=np.repeat((amb == 0)[..., None], 3, -1))
    pb = oracle.project_gaussians_2d_scale_rot_backward(n, mean_px, scales, rot, h, w, d_radii, d_conics,
                                                        res[0][9].cpu().numpy(), None, res[0][10].cpu().numpy())
    v_xy_d, v_conic_d = res[0][9].cpu().numpy(), res[0][10].cpu().numpy()
    for got, want, nm in ((res[0][2], pb[2], "v_scale"), (res[0][3], pb[3], "v_rot"), (res[0][1], pb[1], "v_mean2d")):
        # 1e-5 of the size of the terms each entry sums (they cancel, v_rot most of all; device sin / cos vs libm's)
        sc = rs_term_magnitudes(d_conics, v_conic_d, v_xy_d, scales, rot, nm)
        check_close("rs " + nm, got.cpu().numpy().reshape(want.shape), want, sc, rtol=1e-5)


def test_2k_image_config4_against_the_oracle(oracle):
    """BASELINE config 4 (2040x1356, N = 50 000): image and gradients of the fused path against the CPU oracle on the
    device's own projection (the oracle needs about a second at this size)."""
    from gaussianimage_plus_amd.hotpath import HotPath
    n, h, w = 50000, 1356, 2040
    xyz, L, col, op = synth_cholesky(n, h, w, 44)
    v = _v_out(h, w, 6)
    hp = HotPath(n, h, w, device=DEV, mode="fused")
    hp.set_inputs(xyz, L, col, op)
    hp.set_v_out(v)
    hp.step(pipelined=False)
    hp.check_status()
    tb = oracle.tile_bounds(h, w)
    po = oracle.project_gaussians_2d_forward(n, 3.0, xyz, L, h, w, tb, 0.01, 1.0)
    d_xys, d_conics, d_radii, d_nth = (t.cpu().numpy() for t in (hp.xys, hp.conics, hp.radii, hp.nth))
    assert np.array_equal(d_radii, po[2]) and np.array_equal(d_nth, po[4])   # integer outputs: bit-exact
    assert np.array_equal(d_xys, po[0]) and np.array_equal(d_conics, po[3])  # same operations, same rounding
    m, cum = oracle.compute_cumulative_intersects(d_nth)
    _, _, so, go, bins = oracle.bin_and_sort_gaussians(n, m, d_xys, np.zeros(n, np.float32), d_radii, cum, tb, 1.0)
    ids, tbins = hp.tile_lists()
    ids, tbins = ids.cpu().numpy(), tbins.cpu().numpy()
    T = tb[0] * tb[1]
    assert np.array_equal(tbins[:, 1] - tbins[:, 0], bins[:T, 1] - bins[:T, 0])
    assert np.array_equal(np.concatenate([ids[a:b] for a, b in tbins]), go)   # every tile's ascending id list
    out_o, fT, fidx, amb, absimg = oracle.rasterize_sum_forward(tb, (16, 16, 1), (w, h, 1), go, bins, d_xys, d_conics,
                                                                col, op, with_aux=True)
    check_close("2K out_img", hp.out_img.cpu().numpy(), out_o, absimg, mask=np.repeat((amb == 0)[..., None], 3, -1))
    want = oracle.rasterize_sum_backward(h, w, 16, 16, go, bins, d_xys, d_conics, col, op, None, fT, fidx,
                                         v.cpu().numpy(), with_aux=True, with_amb9=True)
    okg = want[4] == 0
    for got, wv, sl, nm in ((hp.v_xy, want[0], slice(0, 2), "v_xy"), (hp.v_conic, want[1], slice(2, 5), "v_conic"),
                            (hp.v_rgb, want[2], slice(5, 8), "v_rgb"), (hp.v_opac, want[3], slice(8, 9), "v_opacity")):
        g = got.cpu().numpy().reshape(wv.shape)
        # 1e-5 of the summed absolute contributions, every element of every unflagged gaussian.  (Until round 3 this
        # comparison allowed 5e-5 of the elements to be off: gaussian 48340 of this scene has a pair whose alpha is
        # 1/255 - 7e-10 at pixel (575, 1261) -- below the cut-off in the oracle's fp32 form, above it in the device's --
        # and the oracle's backward skipped that pair at its final_idx gate BEFORE looking at its alpha, so the gaussian
        # was not flagged ambiguous.  The flag is now raised ahead of the gate, oracle/gi2d_oracle.c.)
        check_close("2K " + nm, g, wv, want[5][:, sl], mask=np.repeat(okg[:, None], g.shape[1], 1), atol=1e-12)
        # the flagged gaussians are not let go either: they may differ by what their flagged pairs add, no more
        bound = 1.001 * want[7][~okg][:, sl] + 1e-5 * want[5][~okg][:, sl] + 1e-12
        diff = np.abs(g[~okg].astype(np.float64) - wv[~okg].astype(np.float64))
        assert (diff <= bound).all(), (nm, float((diff / bound).max()))
    assert okg.mean() > 0.98  # the flags stay rare (measured: 1.05 % of the gaussians have a pair inside the band)


def test_2k_image_config4_size_properties():
    """BASELINE.json config 4 geometry: 50 000 gaussians on a 2040x1356 image (10 880 tiles, ragged edges):
    fused == exact bitwise, linear in colour, <v_out, out(c)> == <v_rgb, c>."""
    from gaussianimage_plus_amd.hotpath import HotPath
    n, h, w = 50000, 1356, 2040
    xyz, L, col, op = synth_cholesky(n, h, w, 44)
    v = _v_out(h, w, 6)
    res = []
    for mode in ("fused", "exact"):
        hp = HotPath(n, h, w, device=DEV, mode=mode)
        hp.set_inputs(xyz, L, col, op)
        img = hp.forward().clone()
        hp.set_v_out(v)
        hp.backward()
        hp.check_status()
        res.append((img, hp.v_rgb.clone(), hp.v_params.clone(), hp.v_mean2d.clone()))
    for a, b in zip(*res):
        assert torch.equal(a, b)
    hp.set_inputs(xyz, L, 2 * col, op)
    img2 = hp.forward()
    assert torch.allclose(img2, 2 * res[0][0], rtol=1e-6, atol=1e-7)
    lhs = float((v.double() * res[0][0].double()).sum())
    rhs = float((res[0][1].double() * torch.from_numpy(col).to(DEV).double()).sum())
    assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), abs(rhs)) + 1e-9


@pytest.mark.parametrize("n,h,w", [(6000, 256, 384), (900, 70, 100), (50000, 512, 768), (380, 16, 16)])
def test_single_pass_step_equals_separate_forward_backward(n, h, w):
    """step() runs forward and backward of a tile in ONE kernel (gi2d_fast_rasterize_forward_backward); the result
    must equal forward() + backward() (two kernels) and the capacity-free ops bit for bit."""
    one, _ = _mk("fused", n, h, w, seed=21)
    two, _ = _mk("fused", n, h, w, seed=21)
    exact, _ = _mk("exact", n, h, w, seed=21)
    v = _v_out(h, w, 6)
    for hp in (one, two, exact):
        hp.set_v_out(v)
    one.step()
    one.step()  # twice: the bucket cursors must come back clean
    one.check_status()
    two.forward()
    two.backward()
    exact.step()
    for name in ("out_img", "v_xy", "v_conic", "v_rgb", "v_opac", "v_mean2d", "v_params"):
        a, b, c = getattr(one, name), getattr(two, name), getattr(exact, name)
        assert torch.equal(a, b), name
        assert torch.equal(a, c), name


@pytest.mark.parametrize("n,h,w", [(6000, 256, 384), (900, 70, 100)])
def test_single_pass_l2_target_matches_torch_loss_gradient(n, h, w):
    """set_target(): the tile pass forms the L2-loss gradient from its own pixel; exact mode does the same with
    torch ops between its forward and backward calls (same fp32 operations -> same bits)."""
    fused, _ = _mk("fused", n, h, w, seed=22)
    exact, _ = _mk("exact", n, h, w, seed=22)
    gt = torch.from_numpy(synth_gt(h, w, 5)).to(DEV)
    # push some pixels outside [0,1] so the clamp's zero-gradient branch is exercised
    fused.colors.mul_(1.8)
    exact.colors.mul_(1.8)
    for hp in (fused, exact):
        hp.set_target(gt)
        hp.step()
        hp.check_status()
    assert float((fused.out_img > 1).float().mean()) > 0.001
    for name in ("out_img", "v_xy", "v_conic", "v_rgb", "v_opac", "v_mean2d", "v_params"):
        assert torch.equal(getattr(fused, name), getattr(exact, name)), name
    want = torch.nn.functional.mse_loss(fused.out_img.clamp(0, 1), gt).item()
    assert abs(fused.loss() - want) <= 1e-5 * want
    # and autograd agrees with the hand-written gradient image
    o = fused.out_img.clone().requires_grad_(True)
    torch.nn.functional.mse_loss(o.clamp(0, 1), gt).backward()
    exact._l2_grad_from_render()
    inner = (fused.out_img > 0) & (fused.out_img < 1)
    assert torch.allclose(exact.v_out[inner], o.grad[inner], rtol=1e-6, atol=0)


def test_single_pass_edge_geometries(oracle):
    """The one-kernel tile pass on awkward inputs: an image smaller than a tile, gaussians covering hundreds of
    tiles (wave-cooperative reduce, (tile, rank) partial rows), gaussians far outside, zero opacity, a single
    gaussian -- against the capacity-free ops bit for bit and against the oracle."""
    from gaussianimage_plus_amd.hotpath import HotPath
    cases = []
    rng = np.random.default_rng(77)
    # (a) 9x7 image, 3 gaussians
    cases.append((7, 9, (rng.random((3, 2)).astype(np.float32) - 0.5), np.array([[2, 0.1, 2]] * 3, np.float32),
                  rng.random((3, 3)).astype(np.float32), np.ones((3, 1), np.float32)))
    # (b) one gaussian in a 40x56 image
    cases.append((40, 56, np.array([[0.1, -0.2]], np.float32), np.array([[3, 0.5, 2]], np.float32),
                  np.array([[0.3, 0.6, 0.9]], np.float32), np.ones((1, 1), np.float32)))
    # (c) 300 gaussians, some huge (cover the whole 160x240 image = 150 tiles), some outside, some transparent
    n = 300
    xyz = ((rng.random((n, 2)) - 0.5) * 1.9).astype(np.float32)
    L = (rng.random((n, 3)) * np.array([3, 0.5, 3]) + np.array([1.5, 0, 1.5])).astype(np.float32)
    L[:6] = [60, 4, 55]
    xyz[6:10] = [4.0, -3.0]
    op = np.ones((n, 1), np.float32)
    op[10:14] = 0.0
    op[14:20] = 2.5  # alpha clamps to 1 near the centre
    cases.append((160, 240, xyz, L, rng.random((n, 3)).astype(np.float32), op))
    for h, w, xyz, L, col, op in cases:
        n = xyz.shape[0]
        one = HotPath(n, h, w, device=DEV, mode="fused")
        exact = HotPath(n, h, w, device=DEV, mode="exact")
        gt = torch.from_numpy(synth_gt(h, w, 9)).to(DEV)
        for hp in (one, exact):
            hp.set_inputs(xyz, L, col, op)
            hp.set_target(gt)
            hp.step()
            hp.check_status()
        for name in ("out_img", "v_xy", "v_conic", "v_rgb", "v_opac", "v_mean2d", "v_params"):
            assert torch.equal(getattr(one, name), getattr(exact, name)), (h, w, name)
        ref = oracle.render_cholesky(xyz, L, col, op, h, w, with_aux=True)
        out_o, fT, fidx, amb, absimg = ref["ras"]
        check_close("edge out_img", one.out_img.cpu().numpy(), out_o, absimg,
                    mask=np.repeat((amb == 0)[..., None], 3, -1))  # measured: 0.06 of the 1e-5 bar


def test_pipelined_step_equals_unpipelined_and_survives_input_changes():
    """step() ends with one launch that also projects + bins for the next step; the results must not depend on that,
    nor on inputs being replaced between steps (the buckets filled ahead are dropped)."""
    n, h, w = 5000, 128, 192
    a, (xyz, L, col, op) = _mk("fused", n, h, w, seed=31)
    b, _ = _mk("fused", n, h, w, seed=31)
    gt = torch.from_numpy(synth_gt(h, w, 3)).to(DEV)
    a.set_target(gt)
    b.set_target(gt)
    for k in range(3):
        a.step()                 # pipelined
        b.step(pipelined=False)  # project+bin, tile pass, reduce+project backward as three calls
        for name in ("out_img", "v_rgb", "v_mean2d", "v_params", "xys", "conics", "radii", "nth"):
            assert torch.equal(getattr(a, name), getattr(b, name)), (k, name)
    a.check_status()
    xyz2, L2, col2, op2 = synth_cholesky(n, h, w, 32)
    a.set_inputs(xyz2, L2, col2, op2)   # buckets were filled ahead for the old inputs
    b.set_inputs(xyz2, L2, col2, op2)
    a.step()
    img = a.forward().clone()           # a separate forward right after a pipelined step uses the bins it left
    b.step(pipelined=False)
    a.check_status()
    for name in ("v_rgb", "v_mean2d", "v_params"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    assert torch.equal(img, b.out_img)
    a.step()                            # and the loop goes on
    assert torch.equal(a.v_params, b.v_params)


def _big_and_small(n_big, n_small, h, w, seed, sigma=(18.0, 60.0)):
    """n_big gaussians tens of pixels wide (tile boxes of far more than 32 tiles) in front of n_small ordinary ones."""
    rng = np.random.default_rng(seed)
    xyz_s, L_s, col_s, op_s = synth_cholesky(n_small, h, w, seed)
    xyz_b = ((rng.random((n_big, 2)) - 0.5) * 1.6).astype(np.float32)
    s = rng.uniform(sigma[0], sigma[1], (n_big, 2))
    L_b = np.stack([s[:, 0], rng.uniform(-5, 5, n_big), s[:, 1]], 1).astype(np.float32)
    col_b = (rng.random((n_big, 3)) * 0.05).astype(np.float32)
    xyz, L = np.concatenate([xyz_b, xyz_s]), np.concatenate([L_b, L_s])
    col, op = np.concatenate([col_b, col_s]), np.ones((n_big + n_small, 1), np.float32)
    return xyz, L, col, op


def test_gaussians_on_more_than_32_tiles_use_the_row_pool_and_equal_the_exact_path():
    """Gaussians on more than GI2D_FAST_S = 32 tiles keep their partial rows in a pooled run (csrc/gi2d_fast_internal.h:
    PrevBox).  Over steps in which they grow (a larger run is allocated), shrink and move (the run is reused), every
    step's image and gradients equal the capacity-free ops bit for bit."""
    from gaussianimage_plus_amd.hotpath import HotPath
    n_big, n_small, h, w = 40, 3000, 256, 384
    xyz, L, col, op = _big_and_small(n_big, n_small, h, w, 21)
    n = n_big + n_small
    fused = HotPath(n, h, w, device=DEV, mode="fused")
    exact = HotPath(n, h, w, device=DEV, mode="exact")
    v = _v_out(h, w, 3)
    rng = np.random.default_rng(5)
    for step in range(6):
        if step:
            scale = (1.35, 0.5, 1.0, 1.6, 0.8)[step - 1]   # grow, shrink below 32 tiles for some, move only, grow, shrink
            L[:n_big] *= np.float32(scale)
            xyz[:n_big] = np.clip(xyz[:n_big] + rng.normal(size=(n_big, 2)).astype(np.float32) * 0.05, -0.95, 0.95)
        for hp in (fused, exact):
            hp.set_inputs(xyz, L, col, op)
            hp.set_v_out(v)
            hp.step(pipelined=False)
            hp.check_status()
        assert int((fused.nth[:n_big] > 32).sum()) >= (0 if step == 2 else 10), "the scene must exercise the pool"
        for a, b in ((fused.out_img, exact.out_img), (fused.v_params, exact.v_params), (fused.v_mean2d, exact.v_mean2d),
                     (fused.v_rgb, exact.v_rgb), (fused.v_opac, exact.v_opac)):
            assert torch.equal(a, b), step


def test_row_pool_overflow_is_flagged_and_step_safe_recovers():
    """The pool holds tiles x 256 rows; 300 gaussians that each cover all 64 tiles of a 128x128 image ask for 19 200 of
    its 16 384: the tile pass must raise the overflow status (and write nothing out of bounds), step_safe() redoes the
    step on the capacity-free ops, and the emptied workspace serves a later problem."""
    from gaussianimage_plus_amd.hotpath import HotPath
    n_big, n_small, h, w = 300, 200, 128, 128
    xyz, L, col, op = _big_and_small(n_big, n_small, h, w, 8, sigma=(70.0, 90.0))
    xyz[:n_big] *= 0.1
    n = n_big + n_small
    fused = HotPath(n, h, w, device=DEV, mode="fused")
    exact = HotPath(n, h, w, device=DEV, mode="exact")
    v = _v_out(h, w, 4)
    for hp in (fused, exact):
        hp.set_inputs(xyz, L, col, op)
        hp.set_v_out(v)
    fused.step(pipelined=False)
    assert int(fused.nth[:n_big].min()) == 64
    with pytest.raises(RuntimeError, match="row pool"):  # told apart from a tile row that overflowed
        fused.check_status()
    fused.step_safe()
    exact.step()
    exact.check_status()
    assert torch.equal(fused.out_img, exact.out_img) and torch.equal(fused.v_params, exact.v_params)
    xyz2, L2, col2, op2 = _big_and_small(20, n - 20, h, w, 9)
    for hp in (fused, exact):
        hp.set_inputs(xyz2, L2, col2, op2)
        hp.step(pipelined=False)
        hp.check_status()
    assert torch.equal(fused.out_img, exact.out_img) and torch.equal(fused.v_params, exact.v_params)


def test_slowly_growing_gaussians_do_not_leak_the_row_pool():
    """The pool is a bump allocator that is only emptied with the workspace (fixed-population fits never do that): a run
    that has to grow is re-allocated with at least half as much again, so the runs a gaussian swelling tile by tile
    leaves behind sum to at most 3 x its last capacity = 4.5 x its live rows (measured here: 5.8 x the peak of the live
    sum over 40 steps, the peaks not being simultaneous); a new run per step -- the round-3 behaviour -- hands out
    ~25 x, more than this pool holds, and ends in a fit whose gradient rows are dropped."""
    from gaussianimage_plus_amd.hotpath import HotPath
    n_big, n_small, h, w = 30, 500, 512, 768
    xyz, L, col, op = _big_and_small(n_big, n_small, h, w, 33, sigma=(28.0, 34.0))
    n = n_big + n_small
    fused = HotPath(n, h, w, device=DEV, mode="fused")
    fused.set_v_out(_v_out(h, w, 6))
    steps, peak_live = 40, 0
    for step in range(steps):
        L[:n_big] *= np.float32(1.025)  # +2.5 % per step: the tile box gains a row or a column every few steps
        fused.set_inputs(xyz, L, col, op)
        fused.step(pipelined=False)
        fused.check_status()
        nth = fused.nth[:n_big]
        peak_live = max(peak_live, int(nth[nth > 32].sum()))
    cursor = int(fused.tile_lists()[0][8].item())   # GI2D_POOL_CURSOR (a word of tile row 0's header): rows handed out since the last init
    pool_rows = fused.T * 256
    assert int((fused.nth[:n_big] > 32).sum()) >= 20 and peak_live > 3000, "the scene must exercise the pool"
    print(f"row pool: {cursor} rows handed out over {steps} steps for {peak_live} live rows (pool {pool_rows})")
    assert cursor <= 7 * peak_live, (cursor, peak_live)
    assert cursor < 0.5 * pool_rows


def test_two_phase_tile_pass_on_a_large_image_with_crowded_tiles_equals_the_exact_path():
    """An image of more tiles than the chip holds at once (here 64 x 40 = 2560 > 1536) runs its single-pass tile kernel in
    two phases (csrc/gi2d_fast.hip): the small form (128 staged entries, eight workgroups per CU) on the tiles whose
    row holds at most 128 candidates, the general form on the ones it passed over.  A scene with both kinds of tile --
    a sparse background and three crowded regions with 130 ... 700 gaussians per tile, above and below the 256-entry
    cap -- must come out bit for bit as the capacity-free ops compute it, in the given-gradient and the L2-target mode,
    and again after the gaussians have moved (rows with appended and departed entries)."""
    from gaussianimage_plus_amd.hotpath import HotPath
    h, w = 640, 1024
    n_bg, n_cl = 9000, 6000
    rng = np.random.default_rng(77)
    xyz_b, L_b, col_b, op_b = synth_cholesky(n_bg, h, w, 78)
    centres = np.array([[-0.5, -0.4], [0.3, 0.5], [0.7, -0.6]], np.float32)
    which = rng.integers(0, 3, n_cl)
    spread = np.array([0.06, 0.03, 0.025], np.float32)[which][:, None]
    xyz_c = (centres[which] + rng.normal(size=(n_cl, 2)).astype(np.float32) * spread).clip(-0.98, 0.98)
    L_c = np.stack([rng.uniform(0.4, 1.2, n_cl), rng.uniform(-0.2, 0.2, n_cl), rng.uniform(0.4, 1.2, n_cl)], 1).astype(np.float32)
    col_c = rng.uniform(0, 0.02, (n_cl, 3)).astype(np.float32)
    xyz, L = np.concatenate([xyz_b, xyz_c]), np.concatenate([L_b, L_c])
    col, op = np.concatenate([col_b, col_c]), np.ones((n_bg + n_cl, 1), np.float32)
    n = n_bg + n_cl
    fused = HotPath(n, h, w, device=DEV, mode="fused")
    exact = HotPath(n, h, w, device=DEV, mode="exact")
    assert fused.T > 1536
    v = _v_out(h, w, 5)
    for step in range(3):
        if step:
            xyz = (xyz + rng.normal(size=xyz.shape).astype(np.float32) * 0.004).clip(-0.98, 0.98)
        for hp in (fused, exact):
            hp.set_inputs(xyz, L, col, op)
            hp.set_v_out(v)
            hp.step(pipelined=False)
            hp.check_status()
        ids, tbins = fused.tile_lists()
        pop = (tbins[:, 1] - tbins[:, 0]).cpu().numpy()
        assert (pop > 256).sum() >= 3 and ((pop > 128) & (pop <= 256)).sum() >= 5 and (pop <= 128).mean() > 0.9, \
            "the scene must have tiles on both sides of the small form's capacity and of the 256-entry cap"
        for a, b in ((fused.out_img, exact.out_img), (fused.v_params, exact.v_params), (fused.v_mean2d, exact.v_mean2d),
                     (fused.v_rgb, exact.v_rgb), (fused.v_opac, exact.v_opac)):
            assert torch.equal(a, b), step


@pytest.mark.parametrize("h,w,n_bg,corner", [(640, 1024, 9000, True), (544, 768, 33500, False)])
def test_pipelined_steps_of_a_two_phase_tile_pass_whose_first_slot_is_a_crowded_tile(h, w, n_bg, corner):
    """The record-set bookkeeping of a tile pass (RecSets: the launch's first workgroup leaves ver[0] = the set it read)
    must not depend on WHICH tile that workgroup handles.  In the two-launch form of a large image the small form's
    first workgroup returns early when the tile at slot 0 holds more than 128 candidates; the pipelined end-of-step
    kernel (reduce + projection backward of step i next to projection + binning of step i + 1) then reads `ver[0]`.
    Two ways to get a crowded tile to slot 0: tile 0 itself crowded on a grid too large for the tile ordering
    (identity order above 2048 tiles), and the FULLEST tile of a 1537 ... 2048-tile grid with more than 32 768
    gaussians, which the ordering workgroup deals to slot 0.  Pipelined steps over frozen inputs, fused against the
    capacity-free ops, bit for bit."""
    from gaussianimage_plus_amd.hotpath import HotPath
    rng = np.random.default_rng(5)
    xyz_b, L_b, col_b, op_b = synth_cholesky(n_bg, h, w, 12)
    n_cl = 400
    centre_px = np.array([7.5, 8.5] if corner else [0.63 * w, 0.41 * h], np.float32)
    pix = centre_px + rng.normal(size=(n_cl, 2)).astype(np.float32) * 2.5
    xyz_c = (pix / np.array([0.5 * w, 0.5 * h], np.float32) - 1.0).clip(-0.995, 0.995).astype(np.float32)
    L_c = np.stack([rng.uniform(0.4, 1.0, n_cl), rng.uniform(-0.2, 0.2, n_cl), rng.uniform(0.4, 1.0, n_cl)], 1).astype(np.float32)
    col_c = rng.uniform(0, 0.02, (n_cl, 3)).astype(np.float32)
    xyz, L = np.concatenate([xyz_b, xyz_c]), np.concatenate([L_b, L_c])
    n = n_bg + n_cl
    col, op = np.concatenate([col_b, col_c]), np.ones((n, 1), np.float32)
    fused = HotPath(n, h, w, device=DEV, mode="fused")
    exact = HotPath(n, h, w, device=DEV, mode="exact")
    assert fused.T > 1536 and (corner == (fused.T > 2048)) and (corner or n > 32768)
    v = _v_out(h, w, 6)
    for hp in (fused, exact):
        hp.set_inputs(xyz, L, col, op)
        hp.set_v_out(v)
    for step in range(4):  # (the ordering workgroup deals the fullest tile to slot 0 from the second step on)
        fused.step(pipelined=True)
        exact.step()
        fused.check_status()
        for a, b in ((fused.out_img, exact.out_img), (fused.v_params, exact.v_params), (fused.v_mean2d, exact.v_mean2d),
                     (fused.v_rgb, exact.v_rgb), (fused.v_opac, exact.v_opac)):
            assert torch.equal(a, b), step
    ids, tbins = fused.tile_lists()
    pop = (tbins[:, 1] - tbins[:, 0]).cpu().numpy()
    assert pop.max() > 128 and (pop[0] > 128 if corner else True), "the scene must put a crowded tile at slot 0"
